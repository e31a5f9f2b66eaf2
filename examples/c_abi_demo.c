/* The drop-in boundary from plain C: no Python, no torch -- include/azgym.h + libazgym_hip.so.
 *
 *   gcc -std=c11 -O2 -Iinclude examples/c_abi_demo.c -o /tmp/c_abi_demo \
 *       -Lalphazero_gym_amd/csrc -lazgym_hip -Wl,-rpath,$PWD/alphazero_gym_amd/csrc -lm
 *   /tmp/c_abi_demo            (needs an MI355X)
 *
 * What it does: MCTSContinuous(n_rollouts=200, c_uct=0.05, c_pw=1, kappa=0.5, gamma=1) (alphazero/search/mcts.py:537-549) for 64
 * Pendulum-v1 roots with a 3 -> 64 -> {1, 2} ELU policy/value network given as a torch-layout weight blob, then
 * MCTS.return_results (mcts.py:269-307) for every tree.  Exit code 0 iff every tree's root counts add up to n_rollouts. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "azgym.h"

int main(void) {
    enum { B = 64, NS = 200, H = 64 };
    azg_config cfg;
    memset(&cfg, 0, sizeof cfg);
    cfg.struct_size = (int32_t)sizeof cfg;
    cfg.env_id = AZG_ENV_PENDULUM_V1; cfg.mode = AZG_MODE_CONTINUOUS;
    cfg.n_trees = B; cfg.n_sims = NS;
    cfg.c_uct = 0.05; cfg.gamma = 1.0; cfg.c_pw = 1.0; cfg.kappa = 0.5;
    cfg.reward_scale = 16.2736044; cfg.action_bound = 2.0; cfg.seed = 34;
    azg_engine* e = NULL;
    if (azg_engine_create(&cfg, &e) != AZG_OK) { fprintf(stderr, "create: %s\n", azg_last_error(NULL)); return 2; }

    azg_mlp_desc d;
    memset(&d, 0, sizeof d);
    d.struct_size = (int32_t)sizeof d;
    d.in_dim = 3; d.n_hidden = 1; d.hidden[0] = H; d.n_dist = 2; d.activation = AZG_ACT_ELU;
    d.log_std_min = -5.0f; d.log_std_max = 2.0f;
    /* state_dict order: trunk W[H][3], b[H]; value_head W[1][H], b[1]; dist_head W[2][H], b[2] */
    size_t n = (size_t)H * 3 + H + H + 1 + 2 * H + 2;
    float* blob = (float*)malloc(n * sizeof(float));
    unsigned s = 12345u;
    for (size_t i = 0; i < n; ++i) { s = s * 1664525u + 1013904223u; blob[i] = ((float)(s >> 8) / 16777216.0f - 0.5f) * 0.4f; }
    if (azg_set_weights(e, &d, blob, n) != AZG_OK) { fprintf(stderr, "weights: %s\n", azg_last_error(e)); return 2; }

    double* roots = (double*)malloc(sizeof(double) * B * 2);
    azg_synthetic_roots(e, roots);                                      /* fixed-seed Env.reset() states */
    if (azg_search(e, roots, NULL) != AZG_OK) { fprintf(stderr, "search: %s\n", azg_last_error(e)); return 2; }

    int K = azg_max_children(e);
    float* actions = (float*)malloc(sizeof(float) * B * K);
    int32_t* counts = (int32_t*)malloc(sizeof(int32_t) * B * K);
    double* Q = (double*)malloc(sizeof(double) * B * K);
    double* vt = (double*)malloc(sizeof(double) * B);
    int32_t* nch = (int32_t*)malloc(sizeof(int32_t) * B);
    if (azg_results(e, actions, counts, Q, vt, nch) != AZG_OK) { fprintf(stderr, "results: %s\n", azg_last_error(e)); return 2; }
    int bad = 0;
    for (int t = 0; t < B; ++t) {
        int sum = 0, best = 0;
        for (int a = 0; a < nch[t]; ++a) { sum += counts[t * K + a]; if (counts[t * K + a] > counts[t * K + best]) best = a; }
        if (sum != NS) bad++;
        if (t < 3) printf("tree %d: root (%.3f, %.3f)  children %d  action %.4f (visits %d, Q %.5f)  V_target %.5f\n", t, roots[2 * t],
                          roots[2 * t + 1], nch[t], actions[t * K + best], counts[t * K + best], Q[t * K + best], vt[t]);
    }
    float ms = 0.0f;
    azg_last_search_ms(e, &ms);
    printf("%d trees x %d simulations in %.3f ms; %d trees with a wrong visit total\n", B, NS, ms, bad);
    azg_engine_destroy(e);
    free(blob); free(roots); free(actions); free(counts); free(Q); free(vt); free(nch);
    return bad ? 1 : 0;
}
