// search_kernel.cuh -- the persistent search kernel: one launch = all n_sims traces of B trees (16 trees per workgroup).
#pragma once
#include "records.h"
#include "env.cuh"
#include "mlp.cuh"
#include "tree.cuh"

template <int ENV, int HP, int NREG, bool TLDS, bool GMM>
__global__ __launch_bounds__(256, 1) void search_kernel(KParams P) {
    constexpr bool CONT = (ENV != AZG_ENV_CARTPOLE);
    constexpr int S = CONT ? 2 : 4;
    typedef typename TreeStore<TLDS>::Rec Rec;
    typedef typename TreeStore<TLDS>::Id Id;
    __shared__ f32x4 s_parts[4 * 64];
    __shared__ float s_obsT[4 * 16];
    __shared__ float s_bhead[16];
    __shared__ float s_ln[2 * 64];
    extern __shared__ double s_dyn[];   // sqrt_tab [tab_n], pw_need [n_sims+2] ints, two activation buffers, (TLDS) the 16 trees' hot records

    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;
    const int sub = lane & 15;
    const int tl = wave * 4 + (lane >> 4);          // tree within the workgroup
    const int tree = blockIdx.x * TREES_PER_WG + tl;
    const bool live = tree < P.B;
    const unsigned gtree = (unsigned)(P.tree_base + tree);

    double* s_sqrt = s_dyn;
    int* s_pw = (int*)(s_dyn + P.tab_n);
    const size_t act_off = ((size_t)P.tab_n * 8 + (size_t)(P.n_sims + 2) * 4 + 15) / 16 * 16;
    f32x4* s_actA = (f32x4*)((char*)s_dyn + act_off);
    f32x4* s_actB = s_actA + HP / 16 * 64;
    for (int i = tid; i < P.tab_n; i += 256) s_sqrt[i] = P.sqrt_tab[i];
    if (CONT) for (int i = tid; i < P.n_sims + 2; i += 256) s_pw[i] = P.pw_need[i];
    if (tid < 16) s_bhead[tid] = P.bhead[tid];

    // register-resident weights
    WRegs<HP, NREG> wr;
    if constexpr (HP <= 256) {
        constexpr int NTW0 = HP / 64;
#pragma unroll
        for (int i = 0; i < NTW0; ++i) {
            wr.w0[i] = P.W0[(wave * NTW0 + i) * 64 + lane];
            wr.b0[i] = P.b0[(wave * NTW0 + i) * 64 + lane];
        }
    }
    if (NREG > 0) {
        constexpr int NTW = HP / 64, S4 = HP / 16;
#pragma unroll
        for (int l = 0; l < NREG; ++l) {
#pragma unroll
            for (int i = 0; i < NTW; ++i) {
                wr.b[l][i] = P.bl[l][(wave * NTW + i) * 64 + lane];
#pragma unroll
                for (int s4 = 0; s4 < S4; ++s4) wr.w[l][i][s4] = P.Wl[l][((wave * NTW + i) * S4 + s4) * 64 + lane];
            }
        }
#pragma unroll
        for (int i = 0; i < NTW; ++i) wr.wh[i] = P.Whead[(wave * NTW + i) * 64 + lane];
    }

    const size_t tb = (size_t)(live ? tree : 0) * P.R;
    Cold* cold = P.cold + tb;
    double* edge_W = P.edge_W + tb;
    float* action = P.action + tb;
    TreeStore<TLDS> ts;
    if (TLDS) {
        // per tree: R records of 16 B, then (continuous) R x Kp child ids or (discrete) R priors
        size_t off = act_off + (size_t)2 * HP * 64;
        size_t per = (size_t)P.R * 16 + (CONT ? (size_t)P.R * P.Kp : (size_t)P.R * 4);
        per = (per + 15) / 16 * 16;
        char* base = (char*)s_dyn + off + per * tl;
        ts.hot = (Rec*)base;
        ts.child = (Id*)(base + (size_t)P.R * 16);
        ts.prior = (float*)(base + (size_t)P.R * 16);
    } else {
        ts.hot = (Rec*)(P.hot + tb);
        ts.child = (Id*)(P.child + tb * P.Kp);
        ts.prior = P.prior + tb;
    }

#ifdef AZG_STAMPS
    unsigned long long st_acc[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#endif
    int nrec = 1;
    unsigned eps_draws = 0;
    int leaf = 0;
    bool need_eval = live;
    // the current trace's path, one record per lane (slot = depth & 15): id, reward, W -- consumed by backup_path
    int path_D = 0, my_depth = -1, pid = 0;
    double pr = 0.0, pW = 0.0;
    // progressive-widening noise: lane `sub` holds the N(0,1) draw for record kbase + sub
    int kbase = 1;
    float eps_c = 0.0f;
    if (CONT && live) eps_c = azg_normal(P.seed, gtree, P.search_idx, (unsigned)(kbase + sub));

    // ---- root (initialize_search + evaluation / add_value_estimate: mcts.py:364-383, 437; 589-600, 672)
    {
        double rs[S], sn;
#pragma unroll
        for (int k = 0; k < S; ++k) rs[k] = live ? P.roots[(size_t)tree * S + k] : 0.0;
        float obs[4];
        env_obs<ENV>(rs, obs, &sn);
        if (live && sub == 0) {
            Rec h = make_edge<Rec>(0.0, 0);
            h.node_n = (decltype(h.node_n))P.carry[tree];
            h.flags = FLAG_EXPANDED;
            clear_pad(h);
            ts.hot[0] = h;
            Cold c;
#pragma unroll
            for (int k = 0; k < 4; ++k) c.s[k] = k < S ? rs[k] : 0.0;
            if (CONT) c.s[2] = sn;
            c.r = 0.0; c.V = 0.0f; c.mu = 0.0f; c.sg = 0.0f; c.pad = 0.0f;
            cold[0] = c;
            edge_W[0] = 0.0;
            if (CONT) action[0] = 0.0f;
        }
        if (sub < 4) s_obsT[sub * 16 + tl] = live ? obs[sub] : 0.0f;
    }
    __syncthreads();

    for (int sim = -1; sim < P.n_sims; ++sim) {
        // ================= network phase: evaluate the 16 pending leaves =================
        STAMP(t_a);
        int any = __syncthreads_or(need_eval ? 1 : 0);
        STAMP(t_b);
#ifdef AZG_STAMPS
        if (any) mlp_forward<HP, NREG>(P, wr, s_obsT, s_actA, s_actB, s_parts, s_ln, wave, lane, st_acc);
#else
        if (any) mlp_forward<HP, NREG>(P, wr, s_obsT, s_actA, s_actB, s_parts, s_ln, wave, lane);
#endif
        STAMP(t_c);

        // ================= tree phase A: finish the evaluated leaf, back up =================
        if (live) {
            float V = 0.0f;
            if (need_eval) {
                V = head_output(s_parts, s_bhead, tl, 0);
                if (CONT) {
                    float mu, sg;
                    float gd[15];
                    if constexpr (GMM) {
                        gmm_params(s_parts, s_bhead, tl, P.ncomp, P.ls_min, P.ls_max, gd);
                        mu = gd[0]; sg = gd[GMM_MAXC];
                        float* g = P.gmm + (tb + leaf) * 3 * GMM_MAXC;
                        if (sub == 0) {
#pragma unroll
                            for (int i = 0; i < 3 * GMM_MAXC; ++i) g[i] = gd[i];
                        }
                    } else {
                        mu = head_output(s_parts, s_bhead, tl, 1);
                        float ls = head_output(s_parts, s_bhead, tl, 2);
                        ls = ls < P.ls_min ? P.ls_min : (ls > P.ls_max ? P.ls_max : ls);
                        sg = azg_expf(ls);
                    }
                    if (sub == 0) { cold[leaf].V = V; cold[leaf].mu = mu; cold[leaf].sg = sg; }
                    if (sim < 0) {
                        // add_pw_action(root) before the first trace (mcts.py:673)
                        int k = nrec++;
                        if constexpr (GMM) gmm_pick(gd, P.ncomp, P.seed, gtree, P.search_idx, (unsigned)k, &mu, &sg);
                        float eps = __shfl(eps_c, k - kbase, 16);
                        float a = P.bound_f * azg_tanhf(mu + sg * eps);
                        if (sub == 0) {
                            Rec h = make_edge<Rec>((double)V, 0);
                            clear_pad(h);
                            ts.hot[k] = h;
                            edge_W[k] = 0.0;
                            action[k] = a;
                            ts.child[0] = (Id)k;
                            ts.hot[0].n_child = 1;
                        }
                    }
                } else {
                    // softmax priors + all num_actions edges with Q_init = V (MCTSDiscrete.evaluation, mcts.py:412-416)
                    const int A = P.A;
                    float mx = head_output(s_parts, s_bhead, tl, 1);
                    for (int a = 1; a < A; ++a) { float v = head_output(s_parts, s_bhead, tl, 1 + a); mx = v > mx ? v : mx; }
                    float sum = 0.0f;
                    for (int a = 0; a < A; ++a) sum = sum + azg_expf(head_output(s_parts, s_bhead, tl, 1 + a) - mx);
                    int k0 = nrec;
                    nrec += A;
                    if (sub < A) {
                        float pr = azg_expf(head_output(s_parts, s_bhead, tl, 1 + sub) - mx) / sum;
                        Rec h = make_edge<Rec>((double)V, leaf);
                        clear_pad(h);
                        ts.hot[k0 + sub] = h;
                        ts.prior[k0 + sub] = pr;
                        edge_W[k0 + sub] = 0.0;
                    }
                    if (sub == 0) {
                        cold[leaf].V = V;
                        ts.hot[leaf].n_child = (decltype(ts.hot[leaf].n_child))A;
                        ts.hot[leaf].first = (decltype(ts.hot[leaf].first))k0;
                    }
                }
            }
            STAMP(t_c2);
            STAMP_ADD(8, t_c, t_c2);    // finish leaf (before backup)
            if (sim >= 0) {
                if (!TLDS) __threadfence_block();   // lane 0's partial record stores above must land before the path is re-read
                backup_path<CONT, TLDS>(ts, cold, edge_W, V, sub, P.gamma_f, P.gamma, path_D, my_depth, pid, pr, pW);
            }
        }
        if (sim == P.n_sims - 1) break;
        __threadfence_block();
        STAMP(t_d);

        // ================= tree phase B: next trace: select down, step the env, expand =================
        // The descent loop contains only UCT levels, so the four trees of a wave run the same code and differ only in
        // trip count; widening and expansion happen once, after the loop, for all four trees together.
        need_eval = false;
        if (live) {
            if (CONT && nrec >= kbase + 16) {
                kbase = nrec;
                eps_c = azg_normal(P.seed, gtree, P.search_idx, (unsigned)(kbase + sub));
            }
            int p = 0;
            Rec hp = ts.hot[0];
            Cold cp = cold[0];   // cold part of the current node, prefetched one level ahead
            path_D = 0; my_depth = sub == 0 ? 0 : -1; pid = 0; pr = 0.0; pW = 0.0;
            int chosen = 0;
            bool widen = false, hit_terminal = false;
            while (true) {
                const int K = hp.n_child;
                if (CONT) {
                    int nn = (int)hp.node_n < P.n_sims + 1 ? (int)hp.node_n : P.n_sims + 1;
                    widen = s_pw[nn] - K > 0;   // NodeContinuous.check_pw (states.py:271-275)
                    if (widen) break;
                }
                STAMP(t_l0);
                int pick = -1;
                if (P.epsilon != 0.0) {
                    // MCTS.epsilon_greedy (mcts.py:190-195)
                    azg_u32x4 b = azg_draw(P.seed, gtree, P.search_idx, eps_draws++, AZG_STREAM_EPS);
                    if ((double)azg_u01(b.v[0]) < P.epsilon) pick = (int)(b.v[1] % (unsigned)K);
                }
                const double sq = s_sqrt[hp.node_n];
                int win_c = 0;
                if (K <= 16) {
                    // the common case: all children fit one 16-lane row
                    const bool valid = sub < K;
                    int c = 0;
                    double U = 0.0;
                    if (valid) {
                        c = CONT ? (int)ts.child[p * P.Kp + sub] : (int)hp.first + sub;
                        Rec h = ts.hot[c];
                        double ratio = sq / (double)((int)h.edge_n + 1);
                        if (CONT) {
                            U = h.Q + P.c_uct * ratio;
                        } else {
                            float pc = ts.prior[c] * P.c_uct_f;   // float32 product (NumPy >= 2 promotion)
                            U = h.Q + (double)pc * ratio;
                        }
                    }
                    if (pick >= 0) win_c = __shfl(c, pick, 16);
                    else win_c = argmax16_payload(U, valid, sub, c);
                } else {
                    double win_u = 0.0;
                    bool have = false;
                    for (int base = 0; base < K; base += 16) {   // children are scanned 16 at a time
                        const int i = base + sub;
                        const bool valid = i < K;
                        int c = 0;
                        double U = 0.0;
                        if (valid) {
                            c = CONT ? (int)ts.child[p * P.Kp + i] : (int)hp.first + i;
                            Rec h = ts.hot[c];
                            double ratio = sq / (double)((int)h.edge_n + 1);
                            if (CONT) {
                                U = h.Q + P.c_uct * ratio;
                            } else {
                                float pc = ts.prior[c] * P.c_uct_f;
                                U = h.Q + (double)pc * ratio;
                            }
                        }
                        int w;
                        if (pick >= 0) w = (pick >= base && pick < base + 16) ? pick - base : -1;
                        else w = argmax16(U, valid, sub);
                        if (w >= 0) {
                            int wc = __shfl(c, w, 16);
                            double wu = __shfl(U, w, 16);
                            if (pick >= 0 || !have || wu > win_u) { win_c = wc; win_u = wu; have = true; }
                        }
                    }
                }
                chosen = win_c;
                Rec hc = ts.hot[chosen];
                STAMP(t_l1);
#ifdef AZG_STAMPS
                st_acc[11] += t_l1 - t_l0; st_acc[12] += 1;
#endif
                if (!(hc.flags & FLAG_EXPANDED)) break;   // an edge without a child node: expand it
                path_D += 1;
                p = chosen;
                hp = hc;
                if (sub == (path_D & 15)) {   // only the slot's lane fetches the level's reward and W (used by backup_path)
                    my_depth = path_D; pid = chosen;
                    pr = cold[chosen].r; pW = edge_W[chosen];
                }
                if (hc.flags & FLAG_TERMINAL) { hit_terminal = true; break; }
                cp = cold[p];
            }
            STAMP(t_x);
            STAMP_ADD(9, t_d, t_x);    // descent until the expansion point
            if (hit_terminal) {
                leaf = p;
            } else {
                float cact = 0.0f;
                if (widen) {
                    // MCTSContinuous.add_pw_action (mcts.py:625-654)
                    const int K = hp.n_child;
                    chosen = nrec++;
                    float eps = __shfl(eps_c, chosen - kbase, 16);
                    float wmu = cp.mu, wsg = cp.sg;
                    if constexpr (GMM) {
                        float gd[15];
                        const float* g = P.gmm + (tb + p) * 3 * GMM_MAXC;
#pragma unroll
                        for (int i = 0; i < 3 * GMM_MAXC; ++i) gd[i] = g[i];
                        gmm_pick(gd, P.ncomp, P.seed, gtree, P.search_idx, (unsigned)chosen, &wmu, &wsg);
                    }
                    cact = P.bound_f * azg_tanhf(wmu + wsg * eps);
                    if (sub == 0) {
                        Rec h = make_edge<Rec>((double)cp.V, p);
                        clear_pad(h);
                        ts.hot[chosen] = h;
                        edge_W[chosen] = 0.0;
                        action[chosen] = cact;
                        ts.child[p * P.Kp + K] = (Id)chosen;
                        ts.hot[p].n_child = (decltype(hp.n_child))(K + 1);
                    }
                }
                // MCTS.expansion (mcts.py:216-238): step the env from the parent's cached state
                STAMP(t_w1);
                STAMP_ADD(13, t_x, t_w1);  // widening (waits for the parent's cold record)
                path_D += 1;
                double ns[S], r, sn;
                int done;
                if (CONT) {
                    if (!widen) cact = action[chosen];
                    pendulum_step(P.v1, cp.s, cp.s[2], cact, ns, &r, &done);
                    r = r / P.reward_scale;   // mcts.py:687
                } else {
                    cartpole_step(cp.s, chosen - (int)hp.first, ns, &r, &done);
                }
                float obs[4];
                env_obs<ENV>(ns, obs, &sn);
                STAMP(t_w2);
                STAMP_ADD(14, t_w1, t_w2);  // env step + observation
                if (sub == 0) {
                    Cold c;
#pragma unroll
                    for (int k = 0; k < 4; ++k) c.s[k] = k < S ? ns[k] : 0.0;
                    if (CONT) c.s[2] = sn;
                    c.r = r; c.V = 0.0f; c.mu = 0.0f; c.sg = 0.0f; c.pad = 0.0f;
                    cold[chosen] = c;
                    ts.hot[chosen].flags = (unsigned char)(FLAG_EXPANDED | (done ? FLAG_TERMINAL : 0));
                }
                if (sub == (path_D & 15)) { my_depth = path_D; pid = chosen; pr = r; pW = 0.0; }
                leaf = chosen;
                need_eval = !done;
                if (sub < 4) s_obsT[sub * 16 + tl] = done ? 0.0f : obs[sub];
            }
            STAMP(t_y);
            STAMP_ADD(10, t_x, t_y);   // widen + env step + node creation
        }
        STAMP(t_f);
        __threadfence_block();
        STAMP(t_e);
        STAMP_ADD(15, t_f, t_e);   // store drain at the end of the tree phase
        STAMP_ADD(0, t_a, t_b);   // wait at the barrier in front of the network phase
        STAMP_ADD(1, t_b, t_c);   // network phase
        STAMP_ADD(2, t_c, t_d);   // finish leaf + backup
        STAMP_ADD(3, t_d, t_e);   // select / step / expand
    }
#ifdef AZG_STAMPS
    if (lane == 0) for (int i = 0; i < 16; ++i) P.stamps[((size_t)blockIdx.x * 4 + wave) * 16 + i] = st_acc[i];
#endif
    if (live) {
        if (sub == 0) P.n_rec[tree] = nrec;
        if (TLDS) {
            // publish the LDS-resident tree in the global format
            RecL* gh = P.hot + tb;
            for (int j = sub; j < nrec; j += 16) {
                Rec h = ts.hot[j];
                RecL o;
                o.Q = h.Q; o.edge_n = h.edge_n; o.node_n = h.node_n; o.parent = (short)h.parent; o.n_child = h.n_child;
                o.first = h.first; o.flags = h.flags; o.pad = 0;
                gh[j] = o;
                if (CONT) {
                    for (int i = 0; i < (int)h.n_child; ++i) P.child[(tb + j) * P.Kp + i] = ts.child[j * P.Kp + i];
                } else {
                    P.prior[tb + j] = ts.prior[j];
                }
            }
        }
    }
}
