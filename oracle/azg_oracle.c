/*
 * azg_oracle.c -- CPU ORACLE.  TEST INFRASTRUCTURE ONLY.
 *
 * A plain-C, one-tree-at-a-time restatement of the reference's MCTS hot path
 * (timoklein/alphazero-gym, /root/reference).  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load this library; the product path
 * (alphazero_gym_amd + libazgym_hip.so) never does.
 *
 * Parity status: PINNED for the tree logic against golden vectors captured by importing the
 * reference's alphazero.search.mcts in the build container (tests/golden/gen_golden.py ->
 * tests/golden/ npz files).  UNPINNED for the gym classic-control dynamics: the `gym` package
 * (pinned gym==0.19.0, requirements.txt:10) is not vendored in the reference nor installed; the
 * dynamics below restate the published closed forms and are checked against the build's own
 * numpy restatement (alphazero_gym_amd/envs.py), see SURVEY.md 8c.
 *
 * Reference lines followed (alphazero/...):
 *   search/mcts.py:418-462   MCTSDiscrete.search           -> search_tree() discrete branch
 *   search/mcts.py:656-702   MCTSContinuous.search         -> search_tree() continuous branch
 *   search/mcts.py:464-493   MCTSDiscrete.selectionUCT     -> select_discrete()
 *   search/mcts.py:704-741   MCTSContinuous.selectionUCT   -> select_continuous()
 *   search/states.py:252-275 NodeContinuous.check_pw       -> pw_need[] table
 *   search/mcts.py:175-195   MCTS.epsilon_greedy           -> eps_greedy()
 *   helpers.py:30-52         argmax (ties: lowest index here, random in the reference)
 *   search/mcts.py:385-416   MCTSDiscrete.evaluation       -> evaluate_node()
 *   search/mcts.py:602-654   add_value_estimate / add_pw_action -> evaluate_node() / widen()
 *   search/mcts.py:241-267   MCTS.backprop, states.py:97-112 Action.update -> backup()
 *   search/mcts.py:269-307   MCTS.return_results + value targets 92-173 -> azo_results()
 *   network/policies.py:101-120, 150-160, 238-259, 340-352, 436-464, 488-499 -> mlp_forward()
 */
#include <math.h>
#ifdef _OPENMP
#include <omp.h>
#endif
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../include/azg_math.h"
#include "../include/azgym.h"

#define FLAG_EXPANDED 1
#define FLAG_TERMINAL 2

typedef struct {
    int n_layers;              /* hidden layers */
    int in_dim, in_pad;
    int hid[AZG_MAX_HIDDEN_LAYERS];   /* true widths */
    int hidp[AZG_MAX_HIDDEN_LAYERS];  /* padded to a multiple of 64 */
    int n_out;                 /* 1 + n_dist */
    int act;
    float ls_min, ls_max;
    int ncomp;                 /* mixture components (0: squashed Normal) */
    int layernorm;
    float* lng[AZG_MAX_HIDDEN_LAYERS];   /* LayerNorm weight / bias, zero padded to hidp */
    float* lnb[AZG_MAX_HIDDEN_LAYERS];
    float* W[AZG_MAX_HIDDEN_LAYERS];  /* [hidp][kp] zero padded; kp = in_dim (layer 0) or hidp[l-1] */
    float* Wp[AZG_MAX_HIDDEN_LAYERS]; /* the same numbers as [hidp/8][kp][8] in the order mlp_forward consumes them (k permuted) */
    float* b[AZG_MAX_HIDDEN_LAYERS];
    float* Wh;                 /* [n_out][hidp_last] */
    float* bh;
    int ready;
} mlp_t;

typedef struct {
    int n_rec;
    int32_t* parent;   /* record of the parent node; -1 root */
    int32_t* edge_n;
    double* edge_W;
    double* edge_Q;
    float* edge_action;
    int32_t* node_n;
    double* node_r;
    float* node_V;
    uint8_t* flags;
    double* state;     /* [R][S_env] */
    float* dist;       /* [R][n_dist]: continuous mu,sigma ; discrete priors */
    int32_t* n_child;
    int32_t* child;    /* [R][Kmax] record ids */
} tree_t;

struct azg_engine {
    azg_config cfg;
    mlp_t mlp;
    int S_env, S_obs, Kmax, R, n_dist;
    int32_t* pw_need;  /* [n_sims + 2] */
    tree_t* trees;
    double* roots;
    int32_t* carry;
    uint32_t search_idx;
    int searched;
    /* azo_trace_enable: per tree and trace, the record the trace ended in and its tightest arg-max (T3 mismatch attribution) */
    int32_t* trace_leaf; double* trace_margin;
    /* self-play */
    int sp_on, sp_max_len, sp_det, sp_cap, sp_steps, sp_row;
    int sp_insert, sp_fs, sp_ring;
    long long sp_total;
    double sp_temperature, sp_agent_eps;
    float* res_actions; int32_t* res_counts; double* res_Q; double* res_vt; int32_t* res_nch;   /* azo_results_resident */
    uint32_t sp_step_idx;
    int32_t* sp_t; int32_t* sp_episode; int32_t* sp_fcnt; double* sp_ret; double* sp_fsum; float* sp_rows;
    char err[256];
};

static char g_create_err[256];

static int fail(azg_engine* e, int code, const char* msg) {
    snprintf(e ? e->err : g_create_err, 256, "%s", msg);
    return code;
}

int azo_abi_version(void) { return AZG_ABI_VERSION; }

/* worker threads of the OpenMP loop over trees (bench.py's cpu_baseline: one per physical core, or 1) */
int azo_set_threads(int n) {
#ifdef _OPENMP
    if (n >= 1) omp_set_num_threads(n);
    return omp_get_max_threads();
#else
    (void)n;
    return 1;
#endif
}
const char* azo_last_error(const azg_engine* e) { return e ? e->err : g_create_err; }

/* ------------------------------------------------------------------ environments (float64) */

static int env_state_dim(int env) { return (env == AZG_ENV_CARTPOLE || env == AZG_ENV_ACROBOT) ? 4 : 2; }
static int env_obs_dim(int env) {
    if (env == AZG_ENV_ACROBOT) return 6;
    return env == AZG_ENV_CARTPOLE ? 4 : ((env == AZG_ENV_MOUNTAINCAR || env == AZG_ENV_MOUNTAINCAR_CONT) ? 2 : 3);
}
static int env_is_discrete(int env) { return env == AZG_ENV_CARTPOLE || env == AZG_ENV_MOUNTAINCAR || env == AZG_ENV_ACROBOT; }
static int env_num_actions(int env) { return env == AZG_ENV_CARTPOLE ? 2 : ((env == AZG_ENV_MOUNTAINCAR || env == AZG_ENV_ACROBOT) ? 3 : 0); }

static void env_obs(int env, const double* s, float* obs) {
    if (env == AZG_ENV_ACROBOT) {
        azg_acrobot_obs(s, obs);   /* (cos, sin of both angles, both velocities: include/azg_math.h) */
    } else if (env == AZG_ENV_CARTPOLE) {
        for (int i = 0; i < 4; ++i) obs[i] = (float)s[i];
    } else if (env == AZG_ENV_MOUNTAINCAR || env == AZG_ENV_MOUNTAINCAR_CONT) {
        obs[0] = (float)s[0]; obs[1] = (float)s[1];
    } else {
        double sn, cs;
        azg_sincos(s[0], &sn, &cs);
        obs[0] = (float)cs; obs[1] = (float)sn; obs[2] = (float)s[1];
    }
}

/* gym CartPoleEnv.step: explicit Euler, reward 1.0 */
static void cartpole_step(const double* s, int action, double* o, double* reward, int* done) {
    const double gravity = 9.8, masspole = 0.1, total_mass = 0.1 + 1.0, length = 0.5;
    const double polemass_length = 0.1 * 0.5, force_mag = 10.0, tau = 0.02;
    const double theta_thr = 12.0 * 2.0 * 3.141592653589793 / 360.0, x_thr = 2.4;
    double x = s[0], x_dot = s[1], theta = s[2], theta_dot = s[3];
    double force = action == 1 ? force_mag : -force_mag;
    double sintheta, costheta;
    azg_sincos(theta, &sintheta, &costheta);
    double temp = (force + (polemass_length * (theta_dot * theta_dot)) * sintheta) / total_mass;
    double thetaacc = (gravity * sintheta - costheta * temp) /
                      (length * (4.0 / 3.0 - (masspole * (costheta * costheta)) / total_mass));
    double xacc = temp - ((polemass_length * thetaacc) * costheta) / total_mass;
    x = x + tau * x_dot;
    x_dot = x_dot + tau * xacc;
    theta = theta + tau * theta_dot;
    theta_dot = theta_dot + tau * thetaacc;
    o[0] = x; o[1] = x_dot; o[2] = theta; o[3] = theta_dot;
    *done = (x < -x_thr) || (x > x_thr) || (theta < -theta_thr) || (theta > theta_thr);
    *reward = 1.0;
}

/* gym MountainCarEnv.step (MountainCar-v0: three actions, push left / none / right): velocity += (a - 1) * force + cos(3 x) * (-gravity),
 * clipped to +-max_speed; x += velocity, clipped to [-1.2, 0.6]; an inelastic wall on the left; reward -1 per step; done at the flag */
static void mountaincar_step(const double* s, int action, double* o, double* reward, int* done) {
    const double min_position = -1.2, max_position = 0.6, max_speed = 0.07, goal_position = 0.5, goal_velocity = 0.0;
    const double force = 0.001, gravity = 0.0025;
    double position = s[0], velocity = s[1];
    double sn, cs;
    azg_sincos(3.0 * position, &sn, &cs);
    velocity = velocity + ((double)(action - 1) * force + cs * (-gravity));
    velocity = velocity < -max_speed ? -max_speed : (velocity > max_speed ? max_speed : velocity);
    position = position + velocity;
    position = position < min_position ? min_position : (position > max_position ? max_position : position);
    if (position == min_position && velocity < 0.0) velocity = 0.0;
    o[0] = position; o[1] = velocity;
    *done = (position >= goal_position) && (velocity >= goal_velocity);
    *reward = -1.0;
}

/* gym Continuous_MountainCarEnv.step (MountainCarContinuous-v0; third-party gym, restated like the others -- envs.py
 * MountainCarContinuousEnv is the definition): force = clip(action, -1, 1); velocity += force * power - 0.0025 * cos(3 x), clipped;
 * x += velocity, clipped to [-1.2, 0.6]; inelastic wall on the left; done at x >= 0.45 with velocity >= 0;
 * reward = 100 * done - 0.1 * action^2 (the action as it came).  The float32 action is widened to float64. */
static void mountaincar_cont_step(const double* s, float action, double* o, double* reward, int* done) {
    const double min_position = -1.2, max_position = 0.6, max_speed = 0.07, goal_position = 0.45, goal_velocity = 0.0, power = 0.0015;
    double position = s[0], velocity = s[1];
    const double a = (double)action;
    const double force = a < -1.0 ? -1.0 : (a > 1.0 ? 1.0 : a);
    double sn, cs;
    azg_sincos(3.0 * position, &sn, &cs);
    velocity = velocity + (force * power - 0.0025 * cs);
    velocity = velocity > max_speed ? max_speed : velocity;
    velocity = velocity < -max_speed ? -max_speed : velocity;
    position = position + velocity;
    position = position > max_position ? max_position : position;
    position = position < min_position ? min_position : position;
    if (position == min_position && velocity < 0.0) velocity = 0.0;
    const int d = (position >= goal_position) && (velocity >= goal_velocity);
    o[0] = position; o[1] = velocity;
    *done = d;
    *reward = (d ? 100.0 : 0.0) - (a * a) * 0.1;
}

/* gym PendulumEnv.step (v0: clip after integrating theta; v1: clip before); the float32 action is widened to float64 */
static void pendulum_step(int v1, const double* s, float action, double* o, double* reward, int* done) {
    const double max_speed = 8.0, dt = 0.05, pi = 3.141592653589793;
    const float max_torque = 2.0f;
    double th = s[0], thdot = s[1];
    float uc = action < -max_torque ? -max_torque : (action > max_torque ? max_torque : action);
    double u = (double)uc;
    double an = azg_pymod(th + pi, 2.0 * pi, 0.15915494309189535) - pi;
    double costs = (an * an + 0.1 * (thdot * thdot)) + 0.001 * (u * u);
    double newth, newthdot, sn, cs;
    if (v1) {
        azg_sincos(th, &sn, &cs);
        newthdot = thdot + (15.0 * sn + 3.0 * u) * dt;
        newthdot = newthdot < -max_speed ? -max_speed : (newthdot > max_speed ? max_speed : newthdot);
        newth = th + newthdot * dt;
    } else {
        azg_sincos(th + pi, &sn, &cs);
        newthdot = thdot + (-15.0 * sn + 3.0 * u) * dt;
        newth = th + newthdot * dt;
        newthdot = newthdot < -max_speed ? -max_speed : (newthdot > max_speed ? max_speed : newthdot);
    }
    o[0] = newth; o[1] = newthdot;
    *reward = -costs;
    *done = 0;
}

static int env_root_terminal(int env, const double* s) {
    if (env == AZG_ENV_MOUNTAINCAR) return s[0] >= 0.5 && s[1] >= 0.0;
    if (env == AZG_ENV_MOUNTAINCAR_CONT) return s[0] >= 0.45 && s[1] >= 0.0;
    if (env == AZG_ENV_ACROBOT) return azg_acrobot_terminal(s);
    if (env != AZG_ENV_CARTPOLE) return 0;
    const double theta_thr = 12.0 * 2.0 * 3.141592653589793 / 360.0, x_thr = 2.4;
    return (s[0] < -x_thr) || (s[0] > x_thr) || (s[2] < -theta_thr) || (s[2] > theta_thr);
}

/* ------------------------------------------------------------------ MLP (the engine's summation-order spec) */

static int pad64(int n) { return (n + 63) / 64 * 64; }

/* canonical accumulation order over a padded hidden vector: i -> k (see DESIGN.md "MLP arithmetic") */
static inline int perm_k(int i) {
    int t = i >> 4, r = (i >> 2) & 3, g = i & 3;
    return 16 * t + 4 * g + r;
}

static float act_fn(int act, float x) { return azg_activation(act, x); }

/* nn.LayerNorm (eps 1e-5) over the H true units of a layer, in the engine's summation order: the 256 threads of a workgroup
 * each hold 16 units of one tree (wave w, lane group g: units 16(w*NTW+i)+4g+r); a thread sums its own, then the 4 lane groups
 * of a wave are added in order, then the 4 waves.  y = ((h - mean) * (1/sqrt(var + eps))) * weight + bias, unfused. */
static float ln_sum(const float* v, int hp) {
    int ntw = hp / 64;
    float total = 0.0f;
    for (int w = 0; w < 4; ++w) {
        float wave = 0.0f;
        for (int g = 0; g < 4; ++g) {
            float s = 0.0f;
            for (int i = 0; i < ntw; ++i)
                for (int r = 0; r < 4; ++r) s = s + v[16 * (w * ntw + i) + 4 * g + r];
            wave = wave + s;
        }
        total = total + wave;
    }
    return total;
}

static void layer_norm(float* h, int hp, int H, const float* gamma, const float* beta) {
    float tmp[4096];
    float mean = ln_sum(h, hp) / (float)H;
    for (int n = 0; n < hp; ++n) { float d = n < H ? h[n] - mean : 0.0f; tmp[n] = d * d; }
    float var = ln_sum(tmp, hp) / (float)H;
    float inv = 1.0f / __builtin_sqrtf(var + 1e-5f);
    for (int n = 0; n < hp; ++n) h[n] = n < H ? ((h[n] - mean) * inv) * gamma[n] + beta[n] : 0.0f;
}

/* out[0] = V, out[1..] = raw distribution head */
static void mlp_forward(const mlp_t* m, const float* obs, float* out) {
    float bufa[4096], bufb[4096];
    float* x = bufa;
    float* h = bufb;
    int kp = m->in_pad;
    for (int i = 0; i < kp; ++i) x[i] = i < m->in_dim ? obs[i] : 0.0f;
    for (int l = 0; l < m->n_layers; ++l) {
        int hp = m->hidp[l];
        /* eight units at a time: each unit's sum is still its own k-ordered fma chain (same bits as one unit after the other);
         * the eight independent chains keep the CPU's fma pipes busy (one 8-lane vfmadd per k when the compiler vectorises:
         * per-lane IEEE fma, same bits).  hp is a multiple of 64. */
        float xp[4096];
        for (int i = 0; i < kp; ++i) xp[i] = x[l == 0 ? i : perm_k(i)];
        for (int n0 = 0; n0 < hp; n0 += 8) {
            float acc[8];
            const float* w = m->Wp[l] + (size_t)n0 * kp;
            for (int j = 0; j < 8; ++j) acc[j] = m->b[l][n0 + j];
            for (int i = 0; i < kp; ++i) {
                const float xv = xp[i];
#pragma omp simd
                for (int j = 0; j < 8; ++j) acc[j] = AZG_FMAF(w[(size_t)i * 8 + j], xv, acc[j]);
            }
            for (int j = 0; j < 8; ++j) h[n0 + j] = act_fn(m->act, acc[j]);
        }
        if (m->layernorm) layer_norm(h, hp, m->hid[l], m->lng[l], m->lnb[l]);
        float* t = x; x = h; h = t;
        kp = hp;
    }
    /* head outputs: chunks of the (padded) hidden vector are chains from 0, the chunk partials are then added in order.
     * HP <= 256: 8 chunks of HP/8 positions; wider layers: chunks of 64 positions */
    int q = kp <= 256 ? kp / 8 : 64;
    int nch = kp / q;
    for (int o = 0; o < m->n_out; ++o) {
        const float* w = m->Wh + (size_t)o * kp;
        float total = m->bh[o];
        for (int c = 0; c < nch; ++c) {
            float p = 0.0f;
            for (int i = c * q; i < (c + 1) * q; ++i) { int k = perm_k(i); p = AZG_FMAF(w[k], x[k], p); }
            total = total + p;
        }
        out[o] = total;
    }
}

/* V + cached distribution parameters of a node */
static void evaluate_obs(const azg_engine* e, const float* obs, float* V, float* dist) {
    float out[1 + 64];
    mlp_forward(&e->mlp, obs, out);
    *V = out[0];
    if (e->cfg.mode == AZG_MODE_CONTINUOUS && e->mlp.ncomp >= 2) {
        /* DiagonalGMMPolicy.forward (policies.py:544-560): [mu_c], [log_std_c], [log_coeff_c]; cached as mu, sigma and the
         * cumulative mixture probabilities (softmax of log_coeff, summed in component order) */
        int C = e->mlp.ncomp;
        float mx = out[1 + 2 * C];
        for (int c = 1; c < C; ++c) mx = out[1 + 2 * C + c] > mx ? out[1 + 2 * C + c] : mx;
        float ex[8], sum = 0.0f, cum = 0.0f;
        for (int c = 0; c < C; ++c) { ex[c] = azg_expf(out[1 + 2 * C + c] - mx); sum = sum + ex[c]; }
        for (int c = 0; c < C; ++c) {
            float ls = out[1 + C + c];
            ls = ls < e->mlp.ls_min ? e->mlp.ls_min : (ls > e->mlp.ls_max ? e->mlp.ls_max : ls);
            dist[c] = out[1 + c];
            dist[C + c] = azg_expf(ls);
            cum = cum + ex[c] / sum;
            dist[2 * C + c] = cum;
        }
    } else if (e->cfg.mode == AZG_MODE_CONTINUOUS) {
        /* DiagonalNormalPolicy.forward (policies.py:436-464) */
        float ls = out[2];
        ls = ls < e->mlp.ls_min ? e->mlp.ls_min : (ls > e->mlp.ls_max ? e->mlp.ls_max : ls);
        dist[0] = out[1];
        dist[1] = azg_expf(ls);
    } else {
        int A = e->n_dist;
        float mx = out[1];
        for (int a = 1; a < A; ++a) mx = out[1 + a] > mx ? out[1 + a] : mx;
        float sum = 0.0f;
        for (int a = 0; a < A; ++a) { dist[a] = azg_expf(out[1 + a] - mx); sum = sum + dist[a]; }
        for (int a = 0; a < A; ++a) dist[a] = dist[a] / sum;
    }
}

/* ------------------------------------------------------------------ engine */

static void alloc_tree(const azg_engine* e, tree_t* t) {
    int R = e->R;
    t->parent = (int32_t*)calloc(R, 4); t->edge_n = (int32_t*)calloc(R, 4);
    t->edge_W = (double*)calloc(R, 8); t->edge_Q = (double*)calloc(R, 8);
    t->edge_action = (float*)calloc(R, 4); t->node_n = (int32_t*)calloc(R, 4);
    t->node_r = (double*)calloc(R, 8); t->node_V = (float*)calloc(R, 4);
    t->flags = (uint8_t*)calloc(R, 1); t->state = (double*)calloc((size_t)R * e->S_env, 8);
    t->dist = (float*)calloc((size_t)R * e->n_dist, 4); t->n_child = (int32_t*)calloc(R, 4);
    t->child = (int32_t*)calloc((size_t)R * e->Kmax, 4);
}

static void free_tree(tree_t* t) {
    free(t->parent); free(t->edge_n); free(t->edge_W); free(t->edge_Q); free(t->edge_action);
    free(t->node_n); free(t->node_r); free(t->node_V); free(t->flags); free(t->state); free(t->dist);
    free(t->n_child); free(t->child);
}

void azo_engine_destroy(azg_engine* e) {
    if (!e) return;
    if (e->trees) { for (int i = 0; i < e->cfg.n_trees; ++i) free_tree(&e->trees[i]); free(e->trees); }
    for (int l = 0; l < AZG_MAX_HIDDEN_LAYERS; ++l) { free(e->mlp.W[l]); free(e->mlp.Wp[l]); free(e->mlp.b[l]); free(e->mlp.lng[l]); free(e->mlp.lnb[l]); }
    free(e->mlp.Wh); free(e->mlp.bh); free(e->pw_need); free(e->roots); free(e->carry);
    free(e->sp_t); free(e->sp_episode); free(e->sp_fcnt); free(e->sp_ret); free(e->sp_fsum); free(e->sp_rows);
    free(e->res_actions); free(e->res_counts); free(e->res_Q); free(e->res_vt); free(e->res_nch);
    free(e->trace_leaf); free(e->trace_margin);
    free(e);
}

int azo_engine_create(const azg_config* cfg, azg_engine** out) {
    if (!cfg || !out) return fail(NULL, AZG_E_INVALID, "null argument");
    if (cfg->struct_size != (int32_t)sizeof(azg_config)) return fail(NULL, AZG_E_INVALID, "azg_config size mismatch");
    if (cfg->n_trees < 1 || cfg->n_sims < 1) return fail(NULL, AZG_E_INVALID, "n_trees and n_sims must be >= 1");
    if (cfg->env_id < 0 || cfg->env_id > AZG_ENV_ACROBOT) return fail(NULL, AZG_E_INVALID, "unknown env_id");
    if (cfg->mode == AZG_MODE_DISCRETE && !env_is_discrete(cfg->env_id))
        return fail(NULL, AZG_E_UNSUPPORTED, "discrete mode requires a discrete-action env (CartPole)");
    if (cfg->mode == AZG_MODE_CONTINUOUS && env_is_discrete(cfg->env_id))
        return fail(NULL, AZG_E_UNSUPPORTED, "continuous mode requires a continuous-action env (Pendulum)");
    if (cfg->tie_break != AZG_TIE_FIRST && cfg->tie_break != AZG_TIE_RANDOM) return fail(NULL, AZG_E_INVALID, "unknown tie_break");
    /* states.py:271-275: with c_pw <= 0 no node is ever entitled to a child and the reference's first selection takes the arg-max of
     * an empty list (helpers.py:30-52 raises) */
    if (cfg->mode == AZG_MODE_CONTINUOUS && !(cfg->c_pw > 0.0 && cfg->c_pw < 1e6 && cfg->kappa >= 0.0 && cfg->kappa <= 8.0))
        return fail(NULL, AZG_E_INVALID, "c_pw must be > 0 and kappa >= 0 (progressive widening: ceil(c_pw (n + 1)^kappa) children)");
    if (cfg->mode == AZG_MODE_DISCRETE && cfg->num_actions != env_num_actions(cfg->env_id))
        return fail(NULL, AZG_E_INVALID, "num_actions does not match the env (CartPole 2, MountainCar 3, Acrobot 3)");
    azg_engine* e = (azg_engine*)calloc(1, sizeof(azg_engine));
    e->cfg = *cfg;
    e->S_env = env_state_dim(cfg->env_id);
    e->S_obs = env_obs_dim(cfg->env_id);
    int ns = cfg->n_sims;
    if (cfg->mode == AZG_MODE_CONTINUOUS) {
        /* NodeContinuous.check_pw (states.py:271-273): ceil(c_pw * (n+1)**kappa), python float pow + math.ceil */
        e->pw_need = (int32_t*)malloc(sizeof(int32_t) * (ns + 2));
        int kmax = 1;
        for (int n = 0; n < ns + 2; ++n) {
            double v = ceil(cfg->c_pw * pow((double)(n + 1), cfg->kappa));
            if (v > 1e6) v = 1e6;
            e->pw_need[n] = (int32_t)v;
            if (n < ns && e->pw_need[n] > kmax) kmax = e->pw_need[n];
        }
        e->Kmax = kmax;
        e->R = ns + 2;
        e->n_dist = 2;
    } else {
        e->Kmax = cfg->num_actions;
        e->R = 1 + cfg->num_actions * (ns + 1);
        e->n_dist = cfg->num_actions;
    }
    if (cfg->tie_break == AZG_TIE_RANDOM && e->Kmax > 16) {   /* the same limit and error as the device engine */
        free(e->pw_need); free(e);
        return fail(NULL, AZG_E_UNSUPPORTED, "tie_break random supports at most 16 children per node");
    }
    /* per-tree arrays are allocated by the thread that first searches the tree (first touch: pages land on its NUMA node) */
    e->trees = (tree_t*)calloc(cfg->n_trees, sizeof(tree_t));
    e->roots = (double*)calloc((size_t)cfg->n_trees * e->S_env, 8);
    e->carry = (int32_t*)calloc(cfg->n_trees, 4);
    *out = e;
    return AZG_OK;
}

int azo_set_weights(azg_engine* e, const azg_mlp_desc* d, const float* blob, size_t n_floats) {
    if (!e || !d || !blob) return AZG_E_INVALID;
    if (d->struct_size != (int32_t)sizeof(azg_mlp_desc)) return fail(e, AZG_E_INVALID, "azg_mlp_desc size mismatch");
    if (d->n_hidden < 1 || d->n_hidden > AZG_MAX_HIDDEN_LAYERS) return fail(e, AZG_E_INVALID, "n_hidden out of range");
    if (d->activation < 0 || d->activation > AZG_ACT_HARDSWISH) return fail(e, AZG_E_INVALID, "unknown activation");
    if (d->in_dim != e->S_obs) return fail(e, AZG_E_INVALID, "in_dim does not match the env observation");
    if (e->cfg.mode == AZG_MODE_CONTINUOUS) {
        int C = d->num_components >= 2 ? d->num_components : 0;
        if (C > 5) return fail(e, AZG_E_UNSUPPORTED, "at most 5 mixture components");
        if (d->n_dist != (C ? 3 * C : 2)) return fail(e, AZG_E_INVALID, "n_dist does not match num_components");
        if (d->n_dist != e->n_dist) {
            /* the per-node distribution cache is sized by n_dist */
            for (int i = 0; i < e->cfg.n_trees; ++i)
                if (e->trees[i].parent) { free(e->trees[i].dist); e->trees[i].dist = (float*)calloc((size_t)e->R * d->n_dist, 4); }
            e->n_dist = d->n_dist;
        }
        e->mlp.ncomp = C;
    } else if (d->n_dist != e->n_dist) return fail(e, AZG_E_INVALID, "n_dist does not match the engine mode");
    size_t need = 0;
    int k = d->in_dim;
    for (int l = 0; l < d->n_hidden; ++l) {
        if (d->hidden[l] < 1 || d->hidden[l] > 4096) return fail(e, AZG_E_INVALID, "hidden width out of range");
        need += (size_t)d->hidden[l] * k + d->hidden[l] + (d->layernorm ? 2 * (size_t)d->hidden[l] : 0);
        k = d->hidden[l];
    }
    need += (size_t)(1 + d->n_dist) * k + (1 + d->n_dist);
    if (need != n_floats) return fail(e, AZG_E_INVALID, "weight blob size mismatch");
    mlp_t* m = &e->mlp;
    m->layernorm = d->layernorm ? 1 : 0;
    for (int l = 0; l < AZG_MAX_HIDDEN_LAYERS; ++l) { free(m->W[l]); free(m->Wp[l]); free(m->b[l]); m->W[l] = NULL; m->Wp[l] = NULL; m->b[l] = NULL; }
    free(m->Wh); free(m->bh);
    m->n_layers = d->n_hidden; m->in_dim = d->in_dim; m->n_out = 1 + d->n_dist; m->act = d->activation;
    m->ls_min = d->log_std_min; m->ls_max = d->log_std_max;
    const float* p = blob;
    /* every hidden layer is zero-padded to one common width HP (multiple of 64); the input to 4 slots */
    int hmax = 0;
    for (int l = 0; l < d->n_hidden; ++l) if (d->hidden[l] > hmax) hmax = d->hidden[l];
    const int HP = pad64(hmax);
    int kt = d->in_dim, kp = (d->in_dim + 3) / 4 * 4;
    m->in_pad = kp;
    for (int l = 0; l < d->n_hidden; ++l) {
        int h = d->hidden[l], hp = HP;
        m->hid[l] = h; m->hidp[l] = hp;
        m->W[l] = (float*)calloc((size_t)hp * kp, 4);
        m->b[l] = (float*)calloc(hp, 4);
        for (int n = 0; n < h; ++n) memcpy(m->W[l] + (size_t)n * kp, p + (size_t)n * kt, sizeof(float) * kt);
        m->Wp[l] = (float*)calloc((size_t)hp * kp, 4);
        for (int n = 0; n < hp; ++n)
            for (int i = 0; i < kp; ++i)
                m->Wp[l][((size_t)(n / 8) * kp + i) * 8 + n % 8] = m->W[l][(size_t)n * kp + (l == 0 ? i : perm_k(i))];
        p += (size_t)h * kt;
        memcpy(m->b[l], p, sizeof(float) * h);
        p += h;
        free(m->lng[l]); free(m->lnb[l]);
        m->lng[l] = (float*)calloc(hp, 4); m->lnb[l] = (float*)calloc(hp, 4);
        if (d->layernorm) {
            memcpy(m->lng[l], p, sizeof(float) * h); p += h;
            memcpy(m->lnb[l], p, sizeof(float) * h); p += h;
        }
        kt = h; kp = hp;
    }
    m->Wh = (float*)calloc((size_t)m->n_out * kp, 4);
    m->bh = (float*)calloc(m->n_out, 4);
    /* value head then dist head */
    memcpy(m->Wh, p, sizeof(float) * kt); p += kt;
    m->bh[0] = *p++;
    for (int o = 0; o < d->n_dist; ++o) memcpy(m->Wh + (size_t)(1 + o) * kp, p + (size_t)o * kt, sizeof(float) * kt);
    p += (size_t)d->n_dist * kt;
    memcpy(m->bh + 1, p, sizeof(float) * d->n_dist);
    m->ready = 1;
    return AZG_OK;
}

/* (the oracle's "device" is the host: the test-suite binds both engines through one table of entry points) */
int azo_set_weights_device(azg_engine* e, const azg_mlp_desc* d, const float* blob, size_t n_floats) { return azo_set_weights(e, d, blob, n_floats); }

int azo_set_search_index(azg_engine* e, uint32_t idx) { if (!e) return AZG_E_INVALID; e->search_idx = idx; return AZG_OK; }

/* ------------------------------------------------------------------ one tree */

typedef struct {
    const azg_engine* e;
    tree_t* t;
    uint32_t gtree;   /* global tree id */
    uint32_t search;
    uint32_t eps_draws;
    double margin;    /* smallest best-minus-runner-up score gap among the arg-max selections of the current trace */
} ctx_t;

/* fill the node half of record j (MCTS.expansion, mcts.py:216-238) and evaluate it */
static void make_node(ctx_t* c, int j, const double* state, double r, int terminal) {
    const azg_engine* e = c->e;
    tree_t* t = c->t;
    memcpy(t->state + (size_t)j * e->S_env, state, sizeof(double) * e->S_env);
    t->node_r[j] = r;
    t->node_n[j] = 0;
    t->n_child[j] = 0;
    t->flags[j] = (uint8_t)(FLAG_EXPANDED | (terminal ? FLAG_TERMINAL : 0));
    if (terminal) {
        /* V = 0 for terminal nodes (mcts.py:406-410, 619-623); their priors/edges are never used */
        t->node_V[j] = 0.0f;
        return;
    }
    float obs[8];
    env_obs(e->cfg.env_id, state, obs);
    evaluate_obs(e, obs, &t->node_V[j], t->dist + (size_t)j * e->n_dist);
    if (e->cfg.mode == AZG_MODE_DISCRETE) {
        /* MCTSDiscrete.evaluation (mcts.py:412-415): all num_actions edges, Q_init = V */
        int A = e->cfg.num_actions;
        for (int a = 0; a < A; ++a) {
            int k = t->n_rec++;
            t->parent[k] = j; t->edge_n[k] = 0; t->edge_W[k] = 0.0; t->edge_Q[k] = (double)t->node_V[j];
            t->edge_action[k] = (float)a; t->flags[k] = 0; t->node_n[k] = 0; t->n_child[k] = 0;
            t->child[(size_t)j * e->Kmax + a] = k;
        }
        t->n_child[j] = A;
    }
}

/* MCTSContinuous.add_pw_action (mcts.py:625-654): sample a = bound*tanh(mu + sigma*eps), Q_init = node.V */
static int widen(ctx_t* c, int p) {
    const azg_engine* e = c->e;
    tree_t* t = c->t;
    int k = t->n_rec++;
    const float* d = t->dist + (size_t)p * e->n_dist;
    float mu = d[0], sigma = d[1];
    float eps = azg_normal(e->cfg.seed, c->gtree, c->search, (uint32_t)k);
    if (e->mlp.ncomp >= 2) {
        /* MixtureSameFamily.sample (policies.py:656-668): pick a component by inverse CDF, then sample it */
        int C = e->mlp.ncomp, comp = C - 1;
        azg_u32x4 b = azg_draw(e->cfg.seed, c->gtree, c->search, (uint32_t)k, AZG_STREAM_PW);
        float u = azg_u01(b.v[2]);
        for (int i = 0; i < C; ++i) if (u < d[2 * C + i]) { comp = i; break; }
        mu = d[comp]; sigma = d[C + comp];
    }
    float z = mu + sigma * eps;
    float a = (float)e->cfg.action_bound * azg_tanhf(z);
    t->parent[k] = p; t->edge_n[k] = 0; t->edge_W[k] = 0.0; t->edge_Q[k] = (double)t->node_V[p];
    t->edge_action[k] = a; t->flags[k] = 0; t->node_n[k] = 0; t->n_child[k] = 0;
    t->child[(size_t)p * e->Kmax + t->n_child[p]] = k;
    t->n_child[p] += 1;
    return k;
}

/* MCTS.epsilon_greedy (mcts.py:190-195) with the engine's Philox draws; returns -1 when greedy */
static int eps_greedy(ctx_t* c, int K) {
    azg_u32x4 b = azg_draw(c->e->cfg.seed, c->gtree, c->search, c->eps_draws++, AZG_STREAM_EPS);
    if ((double)azg_u01(b.v[0]) < c->e->cfg.epsilon) return (int)(b.v[1] % (uint32_t)K);
    return -1;
}

static int select_child(ctx_t* c, int p) {
    const azg_engine* e = c->e;
    tree_t* t = c->t;
    int K = t->n_child[p];
    if (e->cfg.mode == AZG_MODE_CONTINUOUS) {
        int n = t->node_n[p];
        int need = e->pw_need[n < e->cfg.n_sims + 1 ? n : e->cfg.n_sims + 1];
        if (need - K > 0) return widen(c, p);
    }
    if (e->cfg.epsilon != 0.0) {
        int r = eps_greedy(c, K);
        if (r >= 0) return t->child[(size_t)p * e->Kmax + r];
    }
    double sq = sqrt((double)(t->node_n[p] + 1));
    int best = -1;
    double bestU = 0.0, secondU = 0.0;
    double Us[64];
    float cf = (float)e->cfg.c_uct;
    for (int i = 0; i < K; ++i) {
        int k = t->child[(size_t)p * e->Kmax + i];
        double ratio = sq / (double)(t->edge_n[k] + 1);
        double U;
        if (e->cfg.mode == AZG_MODE_DISCRETE) {
            /* NumPy>=2 (NEP 50): float32 prior * python float c_uct is a float32 product */
            float pc = t->dist[(size_t)p * e->n_dist + i] * cf;
            U = t->edge_Q[k] + (double)pc * ratio;
        } else {
            U = t->edge_Q[k] + e->cfg.c_uct * ratio;
        }
        if (best < 0) { best = k; bestU = U; secondU = -INFINITY; }
        else if (U > bestU) { secondU = bestU; best = k; bestU = U; }
        else if (U > secondU) secondU = U;
        Us[i < 64 ? i : 63] = U;
    }
    if (K >= 2 && bestU - secondU < c->margin) c->margin = bestU - secondU;
    if (e->cfg.tie_break == AZG_TIE_RANDOM) {   /* (K <= Kmax <= 16: azo_engine_create) */
        /* helpers.argmax (helpers.py:46-52): uniform among the children that hold the maximum; the draw is keyed by the node and
         * its visit count, so it does not matter when between two visits of the node the selection is taken */
        int cnt = 0;
        for (int i = 0; i < K; ++i) cnt += (Us[i] == bestU);
        if (cnt > 1) {
            azg_u32x4 b = azg_draw(e->cfg.seed, c->gtree, c->search, ((uint32_t)t->node_n[p] << 16) ^ (uint32_t)p, AZG_STREAM_TIE);
            int kth = (int)(b.v[0] % (uint32_t)cnt);
            for (int i = 0; i < K; ++i)
                if (Us[i] == bestU && kth-- == 0) { best = t->child[(size_t)p * e->Kmax + i]; break; }
        }
    }
    return best;
}

/* MCTS.backprop (mcts.py:260-267) */
static void backup(ctx_t* c, int leaf) {
    const azg_engine* e = c->e;
    tree_t* t = c->t;
    int j = leaf;
    int first = 1;
    double R = 0.0;
    while (t->parent[j] >= 0) {
        double gR;
        if (first) {
            /* continuous: V is a float32 0-d array, gamma a python scalar -> float32 product (NEP 50);
             * discrete: V is a python float -> float64 product */
            if (e->cfg.mode == AZG_MODE_CONTINUOUS) gR = (double)((float)e->cfg.gamma * t->node_V[j]);
            else gR = e->cfg.gamma * (double)t->node_V[j];
            first = 0;
        } else {
            gR = e->cfg.gamma * R;
        }
        R = t->node_r[j] + gR;
        t->edge_n[j] += 1;
        t->edge_W[j] += R;
        t->edge_Q[j] = t->edge_W[j] / (double)t->edge_n[j];
        j = t->parent[j];
        t->node_n[j] += 1;
    }
}

static void search_tree(ctx_t* c, const double* root, int carry) {
    const azg_engine* e = c->e;
    tree_t* t = c->t;
    t->n_rec = 1;
    t->parent[0] = -1; t->edge_n[0] = 0; t->edge_W[0] = 0.0; t->edge_Q[0] = 0.0; t->edge_action[0] = 0.0f;
    make_node(c, 0, root, 0.0, 0);
    t->node_n[0] = carry;
    if (e->cfg.mode == AZG_MODE_CONTINUOUS) widen(c, 0);   /* mcts.py:673 */
    for (int sim = 0; sim < e->cfg.n_sims; ++sim) {
        int node = 0;
        c->margin = INFINITY;
        while (!(t->flags[node] & FLAG_TERMINAL)) {
            int k = select_child(c, node);
            if (t->flags[k] & FLAG_EXPANDED) { node = k; continue; }
            double ns[4], r;
            int done;
            const double* s = t->state + (size_t)node * e->S_env;
            if (e->cfg.env_id == AZG_ENV_CARTPOLE) cartpole_step(s, (int)t->edge_action[k], ns, &r, &done);
            else if (e->cfg.env_id == AZG_ENV_MOUNTAINCAR) mountaincar_step(s, (int)t->edge_action[k], ns, &r, &done);
            else if (e->cfg.env_id == AZG_ENV_ACROBOT) azg_acrobot_step(s, (int)t->edge_action[k], ns, &r, &done);
            else if (e->cfg.env_id == AZG_ENV_MOUNTAINCAR_CONT) mountaincar_cont_step(s, t->edge_action[k], ns, &r, &done);
            else pendulum_step(e->cfg.env_id == AZG_ENV_PENDULUM_V1, s, t->edge_action[k], ns, &r, &done);
            if (e->cfg.mode == AZG_MODE_CONTINUOUS) r = r / e->cfg.reward_scale;   /* mcts.py:687 */
            make_node(c, k, ns, r, done);
            node = k;
            break;
        }
        backup(c, node);
        if (e->trace_leaf) {
            size_t ti = (size_t)(t - e->trees) * e->cfg.n_sims + sim;
            e->trace_leaf[ti] = node | ((t->parent[node] < 0 ? 0 : t->parent[node]) << 16); e->trace_margin[ti] = c->margin;
        }
    }
}

int azo_upload_roots(azg_engine* e, const double* roots, const int32_t* carry) {
    if (!e || !roots) return AZG_E_INVALID;
    for (int i = 0; i < e->cfg.n_trees; ++i) {
        if (env_root_terminal(e->cfg.env_id, roots + (size_t)i * e->S_env))
            return fail(e, AZG_E_TERMINAL_ROOT, "Can't do tree search from a terminal node");
        if (carry && (carry[i] < 0 || carry[i] > (1 << 30))) return fail(e, AZG_E_INVALID, "root_n_carry out of range");
        /* MCTSContinuous never reuses a tree (no forward(): mcts.py:589-600 builds a fresh root for every search) */
        if (carry && carry[i] != 0 && e->cfg.mode == AZG_MODE_CONTINUOUS)
            return fail(e, AZG_E_INVALID, "root_n_carry: continuous searches start from a fresh root (no carried count)");
    }
    memcpy(e->roots, roots, sizeof(double) * (size_t)e->cfg.n_trees * e->S_env);
    if (carry) memcpy(e->carry, carry, 4 * (size_t)e->cfg.n_trees); else memset(e->carry, 0, 4 * (size_t)e->cfg.n_trees);
    return AZG_OK;
}

int azo_search_resident(azg_engine* e) {
    if (!e) return AZG_E_INVALID;
    if (!e->mlp.ready) return fail(e, AZG_E_STATE, "azo_set_weights has not been called");
    int B = e->cfg.n_trees;
    uint32_t sidx = e->search_idx;
    /* static schedule: a tree is searched by the thread that allocated it (see alloc_tree) */
#pragma omp parallel for schedule(static)
    for (int i = 0; i < B; ++i) {
        ctx_t c;
        if (!e->trees[i].parent) alloc_tree(e, &e->trees[i]);
        c.e = e; c.t = &e->trees[i]; c.gtree = (uint32_t)(e->cfg.tree_id_base + i); c.search = sidx; c.eps_draws = 0;
        search_tree(&c, e->roots + (size_t)i * e->S_env, e->carry[i]);
    }
    e->search_idx += 1;
    e->searched = 1;
    return AZG_OK;
}

int azo_search(azg_engine* e, const double* roots, const int32_t* carry) {
    int rc = azo_upload_roots(e, roots, carry);
    if (rc) return rc;
    return azo_search_resident(e);
}

/* Checker-only diagnostics (no azg_ counterpart): from the next search on, keep for every tree and trace the record the trace
 * ended in (low 16 bits; with its parent node's record in the high 16 bits the pair pins the trace's whole path) and the smallest gap between the best and the second-best selectionUCT score met on the way down.  The end-to-end
 * tier (T3: reference with its torch MLP vs this arithmetic) uses them to show that a tree whose visit counts differ from the
 * reference's left the reference's sequence of traces at an arg-max that a ~1e-7 network difference can flip. */
int azo_trace_enable(azg_engine* e) {
    if (!e) return AZG_E_INVALID;
    size_t n = (size_t)e->cfg.n_trees * (size_t)e->cfg.n_sims;
    if (!e->trace_leaf) { e->trace_leaf = (int32_t*)calloc(n, 4); e->trace_margin = (double*)calloc(n, 8); }
    return (e->trace_leaf && e->trace_margin) ? AZG_OK : fail(e, AZG_E_STATE, "out of memory for the trace buffers");
}
int azo_trace_get(azg_engine* e, int32_t* leaf, double* margin) {
    if (!e || !e->trace_leaf || !e->searched) return AZG_E_STATE;
    size_t n = (size_t)e->cfg.n_trees * (size_t)e->cfg.n_sims;
    if (leaf) memcpy(leaf, e->trace_leaf, n * 4);
    if (margin) memcpy(margin, e->trace_margin, n * 8);
    return AZG_OK;
}

int azo_sync(azg_engine* e) { (void)e; return AZG_OK; }
int azo_last_search_ms(azg_engine* e, float* ms) { (void)e; if (ms) *ms = 0.0f; return AZG_OK; }

int azo_results(azg_engine* e, float* actions, int32_t* counts, double* Q, double* v_target, int32_t* n_children) {
    if (!e) return AZG_E_INVALID;
    if (!e->searched) return fail(e, AZG_E_STATE, "no search has run");
    int K = e->Kmax;
    for (int i = 0; i < e->cfg.n_trees; ++i) {
        tree_t* t = &e->trees[i];
        int nc = t->n_child[0];
        double qmax = 0.0, onp = 0.0;
        long tot = 0;
        for (int a = 0; a < nc; ++a) tot += t->edge_n[t->child[a]];
        for (int a = 0; a < K; ++a) {
            int k = a < nc ? t->child[a] : -1;
            if (actions) actions[(size_t)i * K + a] = k >= 0 ? t->edge_action[k] : 0.0f;
            if (counts) counts[(size_t)i * K + a] = k >= 0 ? t->edge_n[k] : 0;
            if (Q) Q[(size_t)i * K + a] = k >= 0 ? t->edge_Q[k] : 0.0;
            if (k >= 0) {
                if (a == 0 || t->edge_Q[k] > qmax) qmax = t->edge_Q[k];
                if (e->cfg.mode == AZG_MODE_DISCRETE) onp += ((double)t->edge_n[k] / (double)tot) * t->edge_Q[k];
            }
        }
        if (e->cfg.mode == AZG_MODE_CONTINUOUS) {
            /* reference quirk: in continuous mode Q has shape (K,1) (the env reward is a length-1 array), so
             * (counts/sum)[K] * Q[K,1] broadcasts to a K x K matrix and np.sum adds all of it (mcts.py:111) */
            for (int a = 0; a < nc; ++a)
                for (int b = 0; b < nc; ++b)
                    onp += ((double)t->edge_n[t->child[b]] / (double)tot) * t->edge_Q[t->child[a]];
        }
        /* off_policy: Q.max() (mcts.py:131); on_policy (mcts.py:111); greedy == Q.max() at the root because the
         * reference's loop guard `node.terminal and node.has_children` (mcts.py:155) is false there */
        if (v_target) v_target[i] = e->cfg.v_target == AZG_VT_ON_POLICY ? onp : qmax;
        if (n_children) n_children[i] = nc;
    }
    return AZG_OK;
}

int azo_root_children(azg_engine* e, int32_t* child_n, double* child_state) {
    if (!e) return AZG_E_INVALID;
    if (!e->searched) return fail(e, AZG_E_STATE, "no search has run");
    int K = e->Kmax, S = e->S_env;
    for (int i = 0; i < e->cfg.n_trees; ++i) {
        tree_t* t = &e->trees[i];
        for (int a = 0; a < K; ++a) {
            int k = a < t->n_child[0] ? t->child[a] : -1;
            int ex = k >= 0 && (t->flags[k] & FLAG_EXPANDED);
            if (child_n) child_n[(size_t)i * K + a] = ex ? t->node_n[k] : -1;
            if (child_state) for (int s = 0; s < S; ++s) child_state[((size_t)i * K + a) * S + s] = ex ? t->state[(size_t)k * S + s] : 0.0;
        }
    }
    return AZG_OK;
}

int azo_root_eval(azg_engine* e, float* value, float* dist) {
    if (!e) return AZG_E_INVALID;
    if (!e->searched) return fail(e, AZG_E_STATE, "no search has run");
    for (int i = 0; i < e->cfg.n_trees; ++i) {
        if (value) value[i] = e->trees[i].node_V[0];
        if (dist) memcpy(dist + (size_t)i * e->n_dist, e->trees[i].dist, 4 * (size_t)e->n_dist);
    }
    return AZG_OK;
}

int azo_dump_tree(azg_engine* e, int32_t* n_records, int32_t* parent, int32_t* edge_n, double* edge_W, double* edge_Q,
                  float* edge_action, int32_t* node_n, double* node_r, float* node_V, uint8_t* node_flags) {
    if (!e) return AZG_E_INVALID;
    if (!e->searched) return fail(e, AZG_E_STATE, "no search has run");
    size_t R = (size_t)e->R;
    for (int i = 0; i < e->cfg.n_trees; ++i) {
        tree_t* t = &e->trees[i];
        size_t n = (size_t)t->n_rec, o = (size_t)i * R;
        if (n_records) n_records[i] = t->n_rec;
        if (parent) { memset(parent + o, 0, 4 * R); memcpy(parent + o, t->parent, 4 * n); }
        if (edge_n) { memset(edge_n + o, 0, 4 * R); memcpy(edge_n + o, t->edge_n, 4 * n); }
        if (edge_W) { memset(edge_W + o, 0, 8 * R); memcpy(edge_W + o, t->edge_W, 8 * n); }
        if (edge_Q) { memset(edge_Q + o, 0, 8 * R); memcpy(edge_Q + o, t->edge_Q, 8 * n); }
        if (edge_action) { memset(edge_action + o, 0, 4 * R); memcpy(edge_action + o, t->edge_action, 4 * n); }
        if (node_n) { memset(node_n + o, 0, 4 * R); memcpy(node_n + o, t->node_n, 4 * n); }
        if (node_r) { memset(node_r + o, 0, 8 * R); for (size_t j = 0; j < n; ++j) node_r[o + j] = (t->flags[j] & FLAG_EXPANDED) ? t->node_r[j] : 0.0; }
        if (node_V) { memset(node_V + o, 0, 4 * R); for (size_t j = 0; j < n; ++j) node_V[o + j] = (t->flags[j] & FLAG_EXPANDED) ? t->node_V[j] : 0.0f; }
        if (node_flags) { memset(node_flags + o, 0, R); memcpy(node_flags + o, t->flags, n); }
    }
    return AZG_OK;
}

int azo_max_children(const azg_engine* e) { return e ? e->Kmax : AZG_E_INVALID; }
int azo_max_records(const azg_engine* e) { return e ? e->R : AZG_E_INVALID; }
int azo_env_state_dim(const azg_engine* e) { return e ? e->S_env : AZG_E_INVALID; }
int azo_obs_dim(const azg_engine* e) { return e ? e->S_obs : AZG_E_INVALID; }

/* synthetic fixed-seed roots */
int azo_synthetic_roots(azg_engine* e, double* roots) {
    if (!e || !roots) return AZG_E_INVALID;
    for (int i = 0; i < e->cfg.n_trees; ++i)
        azg_reset_state(e->cfg.seed, (uint32_t)(e->cfg.tree_id_base + i), 0u, azg_reset_kind(e->cfg.env_id), roots + (size_t)i * e->S_env);
    return AZG_OK;
}

/* ------------------------------------------------------------------ self-play (run_continuous.py:111-142, run_discrete.py:94-122) */

int azo_selfplay_row_len(const azg_engine* e) { return e ? e->S_obs + 3 * e->Kmax + 1 : AZG_E_INVALID; }

int azo_selfplay_begin_ex(azg_engine* e, const azg_selfplay_config* c) {
    if (!e || !c) return AZG_E_INVALID;
    if (c->struct_size != (int32_t)sizeof(azg_selfplay_config)) return fail(e, AZG_E_INVALID, "azg_selfplay_config size mismatch");
    if (c->max_episode_length < 1 || c->capacity_steps < 1) return fail(e, AZG_E_INVALID, "max_episode_length and capacity_steps must be >= 1");
    if (c->final_selection != AZG_FS_MAX_VISIT && c->final_selection != AZG_FS_MAX_VALUE) return fail(e, AZG_E_INVALID, "unknown final_selection");
    if (c->ring_mode != AZG_RING_STOP && c->ring_mode != AZG_RING_FIFO) return fail(e, AZG_E_INVALID, "unknown ring_mode");
    if (!(c->temperature > 0.0)) return fail(e, AZG_E_INVALID, "temperature must be > 0");
    if (c->agent_epsilon < 0.0 || c->agent_epsilon > 1.0) return fail(e, AZG_E_INVALID, "agent_epsilon must be in [0, 1]");
    if (e->cfg.mode == AZG_MODE_DISCRETE && c->final_selection == AZG_FS_MAX_VALUE && c->temperature != 1.0)
        return fail(e, AZG_E_UNSUPPORTED, "final_selection max_value on the device supports temperature 1 only");
    int B = e->cfg.n_trees;
    free(e->sp_t); free(e->sp_episode); free(e->sp_fcnt); free(e->sp_ret); free(e->sp_fsum); free(e->sp_rows);
    e->sp_row = e->S_obs + 3 * e->Kmax + 1;
    e->sp_t = (int32_t*)calloc(B, 4); e->sp_episode = (int32_t*)calloc(B, 4); e->sp_fcnt = (int32_t*)calloc(B, 4);
    e->sp_ret = (double*)calloc(B, 8); e->sp_fsum = (double*)calloc(B, 8);
    e->sp_rows = (float*)calloc((size_t)c->capacity_steps * B * e->sp_row, 4);
    e->sp_on = 1; e->sp_max_len = c->max_episode_length; e->sp_det = c->deterministic; e->sp_cap = c->capacity_steps; e->sp_steps = 0;
    e->sp_insert = 0; e->sp_total = 0; e->sp_fs = c->final_selection; e->sp_ring = c->ring_mode;
    e->sp_temperature = c->temperature; e->sp_agent_eps = c->agent_epsilon;
    e->sp_step_idx = 0;
    for (int i = 0; i < B; ++i)
        azg_reset_state(e->cfg.seed, (uint32_t)(e->cfg.tree_id_base + i), 0u, azg_reset_kind(e->cfg.env_id), e->roots + (size_t)i * e->S_env);
    memset(e->carry, 0, 4 * (size_t)B);
    return AZG_OK;
}

int azo_selfplay_begin(azg_engine* e, int32_t max_episode_length, int32_t deterministic, int32_t capacity_steps) {
    azg_selfplay_config c;
    memset(&c, 0, sizeof(c));
    c.struct_size = (int32_t)sizeof(c);
    c.max_episode_length = max_episode_length; c.deterministic = deterministic; c.capacity_steps = capacity_steps;
    c.final_selection = AZG_FS_MAX_VISIT; c.ring_mode = AZG_RING_STOP; c.temperature = 1.0; c.agent_epsilon = 0.0;
    return azo_selfplay_begin_ex(e, &c);
}

/* One step of the run loops (run_continuous.py:111-142, run_discrete.py:94-122) for every game: act (search + final action,
 * agents.py:257-303, 492-537), buffer.store, Env.step, then reset_mcts / mcts_forward or the episode's end. */
int azo_selfplay_step(azg_engine* e) {
    if (!e || !e->sp_on) return e ? fail(e, AZG_E_STATE, "azo_selfplay_begin has not been called") : AZG_E_INVALID;
    if (e->sp_ring == AZG_RING_STOP && e->sp_steps >= e->sp_cap) return fail(e, AZG_E_STATE, "replay ring is full: download and clear the rows");
    int rc = azo_search_resident(e);
    if (rc) return rc;
    const int B = e->cfg.n_trees, K = e->Kmax, S = e->S_env, So = e->S_obs, RL = e->sp_row;
    const int cont = e->cfg.mode == AZG_MODE_CONTINUOUS;
    /* ReplayBuffer.store (buffers.py:75-82), one step block of B rows at a time */
    int slot;
    if (e->sp_steps < e->sp_cap) { slot = e->sp_steps; e->sp_steps += 1; }
    else { slot = e->sp_insert; e->sp_insert += 1; if (e->sp_insert >= e->sp_steps) e->sp_insert = 0; }
    float* rows = e->sp_rows + (size_t)slot * B * RL;
    for (int i = 0; i < B; ++i) {
        tree_t* t = &e->trees[i];
        const uint32_t gtree = (uint32_t)(e->cfg.tree_id_base + i);
        double* root = e->roots + (size_t)i * S;
        float* row = rows + (size_t)i * RL;
        int nc = t->n_child[0];
        /* replay row (s, actions, counts, Qs, V) of buffer.store */
        env_obs(e->cfg.env_id, root, row);
        double qmax = 0.0, onp = 0.0;
        long tot = 0;
        int cmax = 0, amax = 0, qarg = 0;
        for (int a = 0; a < nc; ++a) tot += t->edge_n[t->child[a]];
        for (int a = 0; a < K; ++a) {
            int k = a < nc ? t->child[a] : -1;
            row[So + a] = k >= 0 ? t->edge_action[k] : 0.0f;
            row[So + K + a] = k >= 0 ? (float)t->edge_n[k] : 0.0f;
            row[So + 2 * K + a] = k >= 0 ? (float)t->edge_Q[k] : 0.0f;
            if (k >= 0) {
                if (a == 0 || t->edge_Q[k] > qmax) { qmax = t->edge_Q[k]; qarg = a; }   /* first index on ties */
                if (!cont) onp += ((double)t->edge_n[k] / (double)tot) * t->edge_Q[k];
                if (a == 0 || t->edge_n[k] > cmax) { cmax = t->edge_n[k]; amax = a; }
            }
        }
        if (cont)
            for (int a = 0; a < nc; ++a)
                for (int b = 0; b < nc; ++b) onp += ((double)t->edge_n[t->child[b]] / (double)tot) * t->edge_Q[t->child[a]];
        row[So + 3 * K] = (float)(e->cfg.v_target == AZG_VT_ON_POLICY ? onp : qmax);
        /* final action */
        int pick;
        if (cont) {
            /* ContinuousAgent.act (agents.py:524-535) */
            pick = e->sp_fs == AZG_FS_MAX_VALUE ? qarg : amax;
            if (e->sp_agent_eps != 0.0) {
                /* epsilon_greedy (agents.py:471-490) */
                azg_u32x4 b = azg_draw(e->cfg.seed, gtree, e->sp_step_idx, 0u, AZG_STREAM_ACT);
                if ((double)azg_u01(b.v[0]) < e->sp_agent_eps) pick = (int)(b.v[1] % (uint32_t)nc);
            }
        } else {
            /* DiscreteAgent.act (agents.py:294-301) with stable_normalizer (helpers.py:26-27): y = (x / max x) ** temp,
             * pi = |y / sum(y)|; then pi.argmax() or numpy's choice(len(pi), p=pi): cdf = cumsum(pi); cdf /= cdf[-1];
             * index = searchsorted(cdf, u, side="right") */
            double pi[64], sum = 0.0;
            for (int a = 0; a < nc; ++a) {
                int k = t->child[a];
                double y;
                if (e->sp_fs == AZG_FS_MAX_VALUE) y = t->edge_Q[k] / qmax;   /* temperature 1 */
                else if (e->sp_temperature == 1.0) y = (double)t->edge_n[k] / (double)cmax;
                else y = pow((double)t->edge_n[k] / (double)cmax, e->sp_temperature);   /* stable_normalizer: (x / max(x)) ** temp, helpers.py:10-27 */
                pi[a] = y;
                sum = sum + y;
            }
            double best = 0.0, last = 0.0;
            pick = 0;
            for (int a = 0; a < nc; ++a) {
                pi[a] = fabs(pi[a] / sum);
                if (a == 0 || pi[a] > best) { best = pi[a]; pick = a; }
                last = last + pi[a];
            }
            if (!e->sp_det) {
                azg_u32x4 b = azg_draw(e->cfg.seed, gtree, e->sp_step_idx, 0u, AZG_STREAM_ACT);
                double u = ((double)b.v[0] + 0.5) * (1.0 / 4294967296.0);
                double cum = 0.0;
                pick = nc - 1;
                for (int a = 0; a < nc; ++a) {
                    cum = cum + pi[a];
                    if (u < cum / last) { pick = a; break; }
                }
            }
        }
        int krec = t->child[pick];
        /* real env step */
        double ns[4], r;
        int done;
        if (e->cfg.env_id == AZG_ENV_CARTPOLE) cartpole_step(root, pick, ns, &r, &done);
        else if (e->cfg.env_id == AZG_ENV_MOUNTAINCAR) mountaincar_step(root, pick, ns, &r, &done);
        else if (e->cfg.env_id == AZG_ENV_ACROBOT) azg_acrobot_step(root, pick, ns, &r, &done);
        else if (e->cfg.env_id == AZG_ENV_MOUNTAINCAR_CONT) mountaincar_cont_step(root, t->edge_action[krec], ns, &r, &done);
        else pendulum_step(e->cfg.env_id == AZG_ENV_PENDULUM_V1, root, t->edge_action[krec], ns, &r, &done);
        e->sp_ret[i] = e->sp_ret[i] + r;
        e->sp_t[i] += 1;
        if (done || e->sp_t[i] >= e->sp_max_len) {
            e->sp_fsum[i] = e->sp_fsum[i] + e->sp_ret[i];
            e->sp_fcnt[i] += 1;
            e->sp_ret[i] = 0.0;
            e->sp_t[i] = 0;
            e->sp_episode[i] += 1;
            azg_reset_state(e->cfg.seed, gtree, (uint32_t)e->sp_episode[i], azg_reset_kind(e->cfg.env_id), root);
            e->carry[i] = 0;
        } else {
            memcpy(root, ns, sizeof(double) * S);
            /* MCTSDiscrete.forward (mcts.py:495-526): a reused root keeps its visit count; continuous trees are rebuilt */
            e->carry[i] = (!cont && (t->flags[krec] & FLAG_EXPANDED)) ? t->node_n[krec] : 0;
        }
    }
    e->sp_total += 1;
    e->sp_step_idx += 1;
    return AZG_OK;
}

int azo_selfplay_rows(azg_engine* e, float* rows, size_t max_rows, int32_t clear) {
    if (!e || !e->sp_on) return AZG_E_STATE;
    size_t n = (size_t)e->sp_steps * e->cfg.n_trees;
    if (n > max_rows) n = max_rows;
    if (rows) memcpy(rows, e->sp_rows, n * e->sp_row * 4);
    if (clear) { e->sp_steps = 0; e->sp_insert = 0; }
    return (int)n;
}

int azo_selfplay_ring(azg_engine* e, int32_t* size_steps, int32_t* insert_step, int64_t* total_steps) {
    if (!e || !e->sp_on) return AZG_E_STATE;
    if (size_steps) *size_steps = e->sp_steps;
    if (insert_step) *insert_step = e->sp_insert;
    if (total_steps) *total_steps = e->sp_total;
    return AZG_OK;
}

/* (host memory here: the oracle has no device) */
/* azg_results_resident's counterpart: the "device" buffers are host memory owned by the engine */
int azo_results_resident(azg_engine* e, const float** actions, const int32_t** counts, const double** Q, const double** v_target,
                         const int32_t** n_children) {
    if (!e) return AZG_E_INVALID;
    size_t B = (size_t)e->cfg.n_trees, K = (size_t)e->Kmax;
    if (!e->res_actions) {
        e->res_actions = (float*)calloc(B * K, 4); e->res_counts = (int32_t*)calloc(B * K, 4); e->res_Q = (double*)calloc(B * K, 8);
        e->res_vt = (double*)calloc(B, 8); e->res_nch = (int32_t*)calloc(B, 4);
    }
    int rc = azo_results(e, e->res_actions, e->res_counts, e->res_Q, e->res_vt, e->res_nch);
    if (rc) return rc;
    if (actions) *actions = e->res_actions;
    if (counts) *counts = e->res_counts;
    if (Q) *Q = e->res_Q;
    if (v_target) *v_target = e->res_vt;
    if (n_children) *n_children = e->res_nch;
    return AZG_OK;
}

int azo_selfplay_rows_device(azg_engine* e, void** ptr, size_t* capacity_rows, size_t* row_len) {
    if (!e || !ptr || !e->sp_on) return AZG_E_STATE;
    *ptr = e->sp_rows;
    if (capacity_rows) *capacity_rows = (size_t)e->sp_cap * e->cfg.n_trees;
    if (row_len) *row_len = (size_t)e->sp_row;
    return AZG_OK;
}

int azo_selfplay_stats(azg_engine* e, double* fsum, int32_t* fcnt, double* env_state) {
    if (!e || !e->sp_on) return AZG_E_STATE;
    int B = e->cfg.n_trees;
    if (fsum) memcpy(fsum, e->sp_fsum, 8 * (size_t)B);
    if (fcnt) memcpy(fcnt, e->sp_fcnt, 4 * (size_t)B);
    if (env_state) memcpy(env_state, e->roots, 8 * (size_t)B * e->S_env);
    return AZG_OK;
}

/* ------------------------------------------------------------------ scalar hooks for tests and the golden generator */

/* fn ids shared with azg_math_selftest on the device */
int azo_math_eval(int fn_id, const double* in, double* out, size_t n) {
    for (size_t i = 0; i < n; ++i) {
        double x = in[i], s, c;
        switch (fn_id) {
            case 0: out[i] = (double)azg_expf((float)x); break;
            case 1: out[i] = (double)azg_expm1f((float)x); break;
            case 2: out[i] = (double)azg_tanhf((float)x); break;
            case 3: out[i] = (double)azg_logf((float)x); break;
            case 4: out[i] = (double)azg_cos2pif((float)x); break;
            case 5: azg_sincos(x, &s, &c); out[i] = s; break;
            case 6: azg_sincos(x, &s, &c); out[i] = c; break;
            case 7: out[i] = azg_pymod(x, 2.0 * 3.141592653589793, 0.15915494309189535); break;
            case 8: out[i] = (double)azg_normal(34u, (uint32_t)x, 0u, (uint32_t)(x * 7.0)); break;
            case 9: out[i] = (double)((float)x / 3.0f); break;
            case 10: out[i] = (double)__builtin_sqrtf((float)x); break;
            case 11: out[i] = x / 3.0; break;
            case 12: out[i] = sqrt(x); break;
            default: return AZG_E_INVALID;
        }
    }
    return AZG_OK;
}

float azo_normal(uint64_t seed, uint32_t tree, uint32_t search, uint32_t draw) { return azg_normal(seed, tree, search, draw); }

void azo_eps_draw(uint64_t seed, uint32_t tree, uint32_t search, uint32_t draw, float* u, uint32_t* r) {
    azg_u32x4 b = azg_draw(seed, tree, search, draw, AZG_STREAM_EPS);
    *u = azg_u01(b.v[0]);
    *r = b.v[1];
}

/* the self-play draws: a game's reset state of an episode, and the final-action draw of a step (u01 float for
 * `random.random() < epsilon`, the double uniform of the inverse-CDF sample, the raw word of the uniform index) */
void azo_reset_state(uint64_t seed, uint32_t tree, uint32_t episode, int kind, double* s) {
    azg_reset_state(seed, tree, episode, kind, s);
}

void azo_act_draw(uint64_t seed, uint32_t tree, uint32_t step, float* u01, double* u, uint32_t* word1) {
    azg_u32x4 b = azg_draw(seed, tree, step, 0u, AZG_STREAM_ACT);
    *u01 = azg_u01(b.v[0]);
    *u = ((double)b.v[0] + 0.5) * (1.0 / 4294967296.0);
    *word1 = b.v[1];
}

float azo_sample_action(float mu, float sigma, float eps, float bound) { return bound * azg_tanhf(mu + sigma * eps); }

/* the uniform that picks the mixture component of widening draw `draw` (third word of the draw's Philox block) */
float azo_gmm_u(uint64_t seed, uint32_t tree, uint32_t search, uint32_t draw) {
    azg_u32x4 b = azg_draw(seed, tree, search, draw, AZG_STREAM_PW);
    return azg_u01(b.v[2]);
}

/* batched evaluator for tests: obs [n][S_obs] -> value [n], dist [n][n_dist], raw [n][1+n_dist] */
int azo_mlp_eval(azg_engine* e, const float* obs, size_t n, float* value, float* dist, float* raw) {
    if (!e || !e->mlp.ready) return AZG_E_STATE;
    for (size_t i = 0; i < n; ++i) {
        float v, d[64], out[65];
        evaluate_obs(e, obs + i * e->S_obs, &v, d);
        if (value) value[i] = v;
        if (dist) memcpy(dist + i * e->n_dist, d, 4 * (size_t)e->n_dist);
        if (raw) { mlp_forward(&e->mlp, obs + i * e->S_obs, out); memcpy(raw + i * (1 + e->n_dist), out, 4 * (size_t)(1 + e->n_dist)); }
    }
    return AZG_OK;
}

/* env step for tests: state [S_env], action -> next [S_env], reward, done, obs [S_obs] */
int azo_env_step(int env_id, const double* state, float action, double* next, double* reward, int32_t* done, float* obs) {
    int d = 0;
    if (env_id == AZG_ENV_CARTPOLE) cartpole_step(state, (int)action, next, reward, &d);
    else if (env_id == AZG_ENV_MOUNTAINCAR) mountaincar_step(state, (int)action, next, reward, &d);
    else if (env_id == AZG_ENV_ACROBOT) azg_acrobot_step(state, (int)action, next, reward, &d);
    else if (env_id == AZG_ENV_MOUNTAINCAR_CONT) mountaincar_cont_step(state, action, next, reward, &d);
    else if (env_id == AZG_ENV_PENDULUM_V0 || env_id == AZG_ENV_PENDULUM_V1) pendulum_step(env_id == AZG_ENV_PENDULUM_V1, state, action, next, reward, &d);
    else return AZG_E_INVALID;
    *done = d;
    if (obs) env_obs(env_id, next, obs);
    return AZG_OK;
}

int azo_env_obs(int env_id, const double* state, float* obs) { env_obs(env_id, state, obs); return AZG_OK; }

void azo_philox(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, uint32_t* out) {
    azg_u32x4 r = azg_philox4x32(c0, c1, c2, c3, k0, k1);
    for (int i = 0; i < 4; ++i) out[i] = r.v[i];
}
