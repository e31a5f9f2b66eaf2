// records.h -- tree record layouts, kernel parameter block, diagnostic stamp macros (device + host).
#pragma once
#include <hip/hip_runtime.h>
#include "../../include/azg_math.h"
#include "../../include/azgym.h"

#define FLAG_EXPANDED 1
#define FLAG_TERMINAL 2
#define TREES_PER_WG 16
#define MAX_STREAM_LAYERS 8

typedef float f32x4 __attribute__((ext_vector_type(4)));

// Record j of a tree = edge j + (once expanded) the node that edge leads to; record 0 is the root.
// "Hot" part: everything selection and the count/Q side of backup touch.  Two encodings:
//   RecS (16 B) lives in LDS for the whole search when the tree fits (<= 255 records, counts < 65536, <= 16 children),
//   RecL (24 B) lives in global memory (any size); it is also the format trees are published in at the end of a search.
// Child lists of RecS nodes (continuous mode): child 0 is `first`; from the second child on the list lives in a per-tree
// byte pool in LDS, in blocks of 4 / 8 / 16 ids at offset 4*cbase (a full block is copied into a fresh one of twice the size).
// A node that ends with c children has used 0 (c <= 1), 1 (c <= 4), 3 (c <= 8) or 7 (c <= 16) four-byte units, i.e. at most
// 7/9 of a unit per child, and a tree of R records has R-1 children in total: POOL_UNITS(R) units always suffice.
struct __attribute__((aligned(16))) RecS {
    double Q;                // edge action value (Q_init = parent V)
    unsigned short edge_n;   // edge visit count
    unsigned short node_n;   // node visit count
    unsigned char parent;    // record of the parent node
    unsigned char n_child : 5;   // node: number of child edges (<= 16)
    unsigned char wnext : 1;     // continuous mode: the node widens at its next visit (tree.cuh: set_wnext)
    unsigned char flags : 2;     // FLAG_EXPANDED | FLAG_TERMINAL
    unsigned char first;     // record of child edge 0 (discrete: the children are contiguous)
    unsigned char cbase;     // continuous, n_child >= 2: the child list starts at pool[4 * cbase];
                             // discrete: index of the node's cached selection among its children (tree_phases.cuh: rec_best)
};
#define POOL_UNITS(R) ((7 * ((R) - 1) + 8) / 9 + 1)
// The same record for trees of 256..511 records (9-bit ids, counts < 2048): everything but Q packed into one 64-bit word.
// Its child-list pool holds 16-bit ids (units of 4 ids = 8 bytes).
struct __attribute__((aligned(16))) RecM {
    double Q;
    unsigned long long edge_n : 11;
    unsigned long long node_n : 11;
    unsigned long long parent : 9;
    unsigned long long first : 9;
    unsigned long long cbase : 9;
    unsigned long long n_child : 5;
    unsigned long long flags : 2;
    unsigned long long wnext : 1;   // (as in RecS)
};
static_assert(sizeof(RecM) == 16, "RecM must be 16 bytes");
// tree storage of a kernel variant: global memory (RecL), LDS with 8-bit ids (RecS), LDS with 9-bit ids (RecM)
#define TS_GLOBAL 0
#define TS_LDS8 1
#define TS_LDS9 2
struct __attribute__((aligned(8))) RecL {
    double Q;
    int edge_n;
    int node_n;
    short parent;
    unsigned short n_child;
    unsigned short first;
    unsigned char flags;
    unsigned char pad;       // discrete: index of the node's cached selection (rec_best); continuous: bit 0 = the node widens at its next visit
};
static_assert(sizeof(RecS) == 16, "RecS must be 16 bytes");
static_assert(sizeof(RecL) == 24, "RecL must be 24 bytes");

// "Cold" part of a node (global memory): read once when a child is expanded from it or an action is sampled at it
struct __attribute__((aligned(16))) Cold {
    double s[4];   // env state (Pendulum: theta, theta_dot, sin(theta) cached, unused)
    double r;      // reward on arriving here (already divided by reward_scale in continuous mode)
    float V;       // value estimate
    float mu;      // continuous: cached squashed-Normal mean
    float sg;      //             and standard deviation
    float pad;
};
static_assert(sizeof(Cold) == 64, "Cold must be 64 bytes");

struct KParams {
    int B, n_sims, R, Kp, A, nd, n_out, n_hidden, act, v1, tree_base, mode;
    double c_uct, gamma, epsilon, reward_scale;
    float c_uct_f, gamma_f, bound_f, ls_min, ls_max;
    unsigned long long seed;
    unsigned search_idx;
    int S;                   // env state dim
    int tab_n;               // entries in sqrt_tab
    const double* roots;     // [B][S]
    const int* carry;        // [B]
    RecL* hot;               // [B][R]      published trees (and working storage when the tree does not fit LDS)
    Cold* cold;              // [B][R]
    double* edge_W;          // [B][R]      edge cumulative return
    float* action;           // [B][R]      continuous: edge action
    float* prior;            // [B][R]      discrete: edge prior
    float* gmm;              // [B][R][15]  continuous mixture head: mu[5] | sigma[5] | cumulative mixture probability[5]
    int ncomp;               // C (0: squashed Normal)
    unsigned short* child;   // [B][R][Kp]  continuous: child record ids of a node, in creation order
    int* n_rec;              // [B]
    const int* pw_need;      // [n_sims+2]
    const double* sqrt_tab;  // [tab_n]  sqrt(n+1)
    const f32x4* W0u;        // [HP] first layer per unit: (w[u][0], w[u][1], w[u][2], w[u][3]) -- the VALU form of the first layer
    const f32x4* b0u;        // [HP/4] first-layer bias, four consecutive units per entry
    const float* W0;         // [HP/16][64]   first layer, inputs 0..3 (one MFMA k-step: lane l = unit l & 15, input l >> 4)
    const float* W0b;        // [HP/16][64]   inputs 4..7 (networks with more than four inputs: Acrobot's six observations)
    int in8;                 // the network has more than four inputs: the first layer takes a second k-step (W0b, observation rows 4..7)
    const f32x4* b0;         // [HP/16][64]
    const f32x4* Wl[MAX_STREAM_LAYERS]; // hidden->hidden layer l (1-based index l-1): [HP/16 tiles][HP/16 s4][64]
    const f32x4* bl[MAX_STREAM_LAYERS]; // [HP/16][64]
    const f32x4* Whead;      // [HP/16 s4][64]
    const float* bhead;      // [16]
    int layernorm;           // nn.LayerNorm after every trunk activation
    int Htrue[MAX_STREAM_LAYERS];          // true (unpadded) width of trunk layer l
    const f32x4* lng[MAX_STREAM_LAYERS];   // LayerNorm weight of trunk layer l, [HP/16][64] (D-register layout, zero padded)
    const f32x4* lnb[MAX_STREAM_LAYERS];   // LayerNorm bias
    // MCTS.return_results of every tree (aux_kernels.cuh: results_for_tree), written by the search kernel's epilogue or results_kernel
    float* res_actions; int* res_counts; double* res_Q; double* res_vt; int* res_nch; int* res_child_n; double* res_child_state;
    float* res_root_V; float* res_root_dist;
    int res_Kmax, res_v_target;
    int tie_random;             // AZG_TIE_RANDOM: exactly equal selection scores are broken by a Philox draw instead of by lowest index
    int env_id;                 // AZG_ENV_* (the discrete family's kernels serve CartPole and MountainCar: env step chosen at run time)
    int trace_cap;              // discrete mode: traces a tree may run per simulation step (search_kernel.cuh; >= 1)
    int lds_state;              // discrete LDS trees: the env states of expanded nodes live in LDS too (set by the launch planning)
    int pw0;                    // continuous mode: pw_need[0], the children a node without visits is entitled to (a new node widens at its first visit iff > 0)
    int publish;                // LDS trees: write them out in the global RecL format after the last trace (azg_dump_tree asks for it;
                                // the product path's results come from the search kernel's epilogue and need no published tree)
    unsigned long long* stamps; // diagnostic build only (-DAZG_STAMPS): [grid][8] cycle sums per phase
};

// ---- state of the lock-step path (lockstep.cuh) between its launches
struct LsLane { int my_depth, pid; double pr, pW; float eps_c; float pad; };
struct LsTree { int nrec; unsigned eps_draws; int leaf, need_eval, path_D, kbase; };

struct LockStep {
    float* obsT;        // [G][4][16]
    f32x4* act[2];      // [G][HP/16][64]   ping-pong activations (D-register layout)
    f32x4* parts;       // [G][HP/64][64]   partial head sums, one per 64-unit chunk
    LsTree* tree;       // [B]
    LsLane* lane;       // [B][16]
};


#ifdef AZG_STAMPS
/* -DAZG_STAMP_ONLY=<slot>: ONE stamp pair only, the one that feeds that slot, so that the build runs within a few per cent of the
   product's time (the full set of stamps costs the lean kernels 20 %: read shares from the full set, cycles from the single pairs).
   STAMP2(var, s1, s2): a time stamp that feeds slots s1 and s2 (-1: none). */
#ifdef AZG_STAMP_ONLY
#define STAMP_ON(s1, s2) ((s1) == (AZG_STAMP_ONLY) || (s2) == (AZG_STAMP_ONLY))
#else
#define STAMP_ON(s1, s2) 1
#endif
#define STAMP3(var, s1, s2, s3) [[maybe_unused]] unsigned long long var = 0; if constexpr (STAMP_ON(s1, s2) || STAMP_ON(s3, s3)) { var = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); }
#define STAMP2(var, s1, s2) STAMP3(var, s1, s2, -1)
#define STAMP(var) STAMP2(var, -2, -2)
#define STAMP_ADD(slot, t0, t1) do { if constexpr (STAMP_ON(slot, slot)) st_acc[slot] += (t1) - (t0); } while (0)
#define STAMP_PARAM , unsigned long long* st_acc
#define STAMP_ARG , st_acc
#ifdef AZG_STAMPS_A   /* slots 4..6 = phase A's parts (finish leaf | backup | re-scoring) instead of the network's */
#define STAMP_A(var, s1, s2) STAMP2(var, s1, s2)
#define STAMP_A_ADD(slot, t0, t1) do { if constexpr (STAMP_ON(slot, slot)) { if (st_acc) st_acc[slot] += (t1) - (t0); } } while (0)
#define STAMP_M(var, s1, s2) [[maybe_unused]] unsigned long long var = 0
#define STAMP_M_ADD(slot, t0, t1)
#else
#define STAMP_A(var, s1, s2)
#define STAMP_A_ADD(slot, t0, t1)
#define STAMP_M(var, s1, s2) STAMP2(var, s1, s2)
#define STAMP_M_ADD(slot, t0, t1) STAMP_ADD(slot, t0, t1)
#endif
#define STAMP_PARAM_OPT , unsigned long long* st_acc = nullptr
#else
#define STAMP_PARAM
#define STAMP_ARG
#define STAMP(var)
#define STAMP2(var, s1, s2)
#define STAMP3(var, s1, s2, s3)
#define STAMP_ADD(slot, t0, t1)
#define STAMP_A(var, s1, s2)
#define STAMP_A_ADD(slot, t0, t1)
#define STAMP_M(var, s1, s2)
#define STAMP_M_ADD(slot, t0, t1)
#define STAMP_PARAM_OPT
#endif
