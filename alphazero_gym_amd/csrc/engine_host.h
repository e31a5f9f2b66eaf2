// engine_host.h -- the engine object behind the C ABI and the entry points of the kernel translation units.
// The persistent search kernels are compiled in several translation units (dispatch_*.hip: one family of template
// instantiations each) so that the library builds in parallel; azg_engine.hip holds the C ABI and the small kernels.
#pragma once
#include <hip/hip_runtime.h>

#include <string>
#include <vector>

#include "records.h"

// diagnostic switches (environment variables, read once when the engine is created): force the other code paths in tests
struct EngineOptions {
    int force_persistent;      // AZG_FORCE_PERSISTENT=1: wide networks on the one-launch kernel instead of the lock-step path
    int force_stream_weights;  // AZG_FORCE_STREAM_WEIGHTS=1: hidden->hidden weights streamed from L2 instead of register-resident
    int force_global_tree;     // AZG_FORCE_GLOBAL_TREE=1: trees in global memory instead of LDS
    int waves;                 // AZG_WAVES=4|8 (0: automatic)
    int groups;                // AZG_GROUPS=1|2 (0: automatic)
    int trace_cap;             // AZG_TRACE_CAP=n: traces a discrete tree may run per simulation step (0: automatic)
    int no_spec;               // AZG_NO_SPEC=1: the general kernels where a compile-time specialised one exists (dispatch.cuh)
    int tile_trees;            // AZG_TILE_TREES=16|8: trees per 16-column MFMA tile of the small-network kernels (0: automatic)
    int ls_team;               // AZG_LS_TEAM=0: the per-layer launches instead of the persistent team kernel (team.cuh)
    int team_wide;             // AZG_TEAM_WIDE=0: only the first form of the team kernel, 32-tree teams at two workgroups per CU (batches beyond that
                               // then take the per-layer launches)
    int team_tt;               // AZG_TEAM_TT=32 / 64: only teams of that many trees (default 0: 32, and 64 for batches beyond two 32-tree workgroups per CU)
    long team_spin_limit;      // AZG_TEAM_SPIN_LIMIT=n: polls a team hand-off may wait before the launch gives up (tests: 0)
};
#define AZG_MAX_DEVICES 64    // per-device caches of kernel attributes (host side)

// Where every element of the engine's weight buffer comes from, for one network shape (azg_engine.hip: build_weight_map)
struct WeightMap {
    bool valid = false;
    azg_mlp_desc desc;
    int HP = 0;
    std::vector<unsigned> src;   // per output float: 1 + index into the caller's blob, 0 = padding zero
    size_t oW0, oW0b, ob0, oW0u, ob0u, oWl[MAX_STREAM_LAYERS], obl[MAX_STREAM_LAYERS], oWh, obh, olg[MAX_STREAM_LAYERS], olb[MAX_STREAM_LAYERS];
};

struct azg_engine {
    azg_config cfg;
    EngineOptions opt;
    int carry_max;           // largest carried root visit count of the uploaded roots
    int S_env, S_obs, Kmax, Kp, R, nd, tab_n;
    int mlp_ready, HP, n_hidden, n_out, act, nreg;
    int tree_lds;            // tree storage of the last launch: TS_GLOBAL, TS_LDS8, TS_LDS9 (records.h)
    int waves, groups, n_cus; // waves / tree groups per workgroup of the last launch; compute units of the device
    int tile_trees;          // trees per group (= per 16-column MFMA tile) of the last launch: 16, or 8 / 4 (half-filled tiles)
    int spec;                // the last launch ran a compile-time specialised kernel (search_kernel's SPEC argument)
    size_t dyn_lds;          // dynamic LDS bytes per workgroup
    float ls_min, ls_max;
    hipStream_t stream;
    hipEvent_t ev0, ev1;
    KParams P;
    std::vector<void*> dev_allocs;
    float* d_wblob;          // every re-laid-out weight tensor of the current network in one buffer (reused while the shape stays)
    size_t w_floats;
    std::vector<float> w_stage;      // host staging of that buffer
    WeightMap wmap;                  // the re-layout as an index map, rebuilt when the network shape changes
    unsigned* d_wmap;                // its device copy (azg_set_weights_device)
    std::vector<void*> dist_allocs;  // continuous mode: per-node mixture cache + root distribution staging, sized by the head
    int dist_nd, dist_ncomp;
    // results staging
    float* d_actions; int* d_counts; double* d_Q; double* d_vt; int* d_nch; int* d_child_n; double* d_child_state;
    float* d_rootV; float* d_rootdist;
    char* d_res_block; void* h_res_block; size_t res_bytes;   // d_Q | d_vt | d_actions | d_counts | d_nch in one block + its pinned host mirror
    double* d_roots; int* d_carry;
    uint32_t search_idx;
    int sp_on, sp_max_len, sp_det, sp_cap, sp_steps, sp_row;   // sp_steps = ReplayBuffer.size in steps
    int sp_insert, sp_fs, sp_ring;   // ReplayBuffer.insert_index in steps; final selection; ring mode
    long long sp_total;              // steps played since begin
    double sp_agent_eps;
    double* d_sp_ctab;
    uint32_t sp_step_idx;
    int* d_sp_t; int* d_sp_episode; int* d_sp_fcnt; double* d_sp_ret; double* d_sp_fsum; float* d_sp_rows;
    std::vector<void*> sp_allocs;
    unsigned* d_team_cnt; size_t team_cnt_bytes;   // team kernel: hand-off counters + abort word
    int team_pending;        // a team kernel has been launched since its abort word was last read
    int team_fallbacks;      // searches it gave up on (redone by the per-layer launches)
    uint32_t team_search_idx;
    int team_tt;             // trees per team of the last team launch (32 or 64)
    int team_parts;          // launches the last team search was cut into (batches beyond the widest form)
    int team_kc, team_minb;  // the team kernel form of the last launch (chunk length, workgroups per CU)
    int launch_timed;        // the last search's launch recorded ev0 / ev1 itself (hipExtLaunchKernelGGL)
    int kernel_form;         // what the last search ran as: 0 search_kernel, 1 lock-step launches, 2 team kernel
    int lds_exit;            // AZG_LDS_*: why the last search's trees were not LDS-resident (dispatch.cuh: launch)
    int lds_warned;          // the one stderr line about it has been printed
    LockStep ls;             // lock-step path for wide networks (lockstep.cuh)
    std::vector<void*> ls_allocs;
    int ls_hp;
    float* d_eval; size_t eval_floats;   // scratch of azg_mlp_eval (grow-only)
    int searched, results_valid;
    int publish_always;      // AZG_PUBLISH_TREES=1: every search writes its LDS trees out in the global format (diagnostic tools)
    int publish_once;        // set by azg_dump_tree around its re-run of the last search
    int published;           // the last search's trees are in global memory (global-tree / lock-step / team forms always are)
    int redo_ok;             // roots, carried counts and weights are still the ones the last search ran on: azg_dump_tree may re-run it
    float last_ms;
    uint32_t last_search_idx;   // the search index the last search ran under (azg_dump_tree re-runs it with this one)
    float ms_kept; int ms_kept_valid;   // kernel time of the last search, kept across azg_dump_tree's re-run of it
    std::string err;
};

// one search of all trees on e->stream with the kernel variant that fits (dispatch.cuh); hipErrorInvalidValue: no such variant
hipError_t azg_dispatch_cartpole(azg_engine* e);
hipError_t azg_dispatch_pendulum_small(azg_engine* e);   // hidden width (padded) <= 128
hipError_t azg_dispatch_pendulum_large(azg_engine* e);   // 256 and wider
hipError_t azg_dispatch_acrobot(azg_engine* e);          // Acrobot-v1 (discrete MCTS, six network inputs), all widths, one-launch kernels only
hipError_t azg_dispatch_mcc(azg_engine* e);              // MountainCarContinuous (continuous MCTS with terminal nodes), all widths
hipError_t azg_ls_dispatch_mcc(azg_engine* e);
hipError_t azg_team_dispatch_mcc(azg_engine* e);
hipError_t azg_ls_dispatch_cartpole(azg_engine* e);      // lock-step path (lockstep.cuh), buffers prepared by the caller
hipError_t azg_ls_dispatch_pendulum(azg_engine* e);
// the same search as ONE persistent launch (team.cuh); hipErrorNotReady: its workgroups cannot all be resident, use the launches
hipError_t azg_team_dispatch_cartpole(azg_engine* e);
hipError_t azg_team_dispatch_pendulum(azg_engine* e);
// the team kernel's forms for more than two 32-tree workgroups per CU (config E's network; team_dispatch.cuh); dry: residency check only
hipError_t azg_team_wide_forms(azg_engine* e, int g_base, int G, bool dry, bool common, bool t32, bool t64);
// batched network inference of n observations (device pointers) on e->stream (mlp_eval.cuh)
hipError_t azg_dispatch_mlp_eval(azg_engine* e, const float* obs, int n, float* value, float* dist, float* raw);
