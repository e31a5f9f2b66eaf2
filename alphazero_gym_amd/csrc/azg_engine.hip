// azg_engine.hip -- MI355X (gfx950) batched MCTS engine behind the C ABI of include/azgym.h.
//
// One persistent kernel launch runs a whole MCTS search (all n_sims simulations) for B independent trees.
//   * workgroup = 256 threads = 4 waves = one group of 16 trees (= one 16-row MFMA tile of leaf evaluations);
//     a workgroup never talks to another one, so there is no grid-wide synchronisation anywhere.
//   * tree phase: a 16-lane sub-wave owns one tree.  Lanes scan the <=16 children of a node in parallel
//     (PUCT / progressive-widening UCT, float64), arg-max by a 4-step butterfly, step the closed-form
//     environment, expand, and back the return up the path.
//   * network phase: the policy/value MLP for the 16 new leaves on v_mfma_f32_16x16x4_f32.  Activations are
//     kept transposed ([unit][tree]) so that an MFMA's D registers are the next layer's B operand as they
//     stand; hidden->hidden weights can live in the 512-entry VGPR/AGPR file for the whole search (NREG>0).
// Reference semantics: alphazero/search/mcts.py (search 418-462 / 656-702, selectionUCT 464-493 / 704-741,
// backprop 241-267, return_results 269-307), alphazero/search/states.py, alphazero/network/policies.py.
// The arithmetic (operation order, float32/float64 placement) is specified by oracle/azg_oracle.c, which is pinned
// to the reference by tests/golden; this file must agree with it bit for bit.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "engine_host.h"
#include "env.cuh"
#include "mlp.cuh"
#include "tree.cuh"
#include "aux_kernels.cuh"

// ------------------------------------------------------------------------------------------------ host side

static std::string g_create_err;

static int fail(azg_engine* e, int code, const std::string& msg) {
    if (e) e->err = msg; else g_create_err = msg;
    return code;
}

#define HIPCHK(e, call)                                                                                  \
    do {                                                                                                 \
        hipError_t _rc = (call);                                                                         \
        if (_rc != hipSuccess) return fail(e, AZG_E_DEVICE, std::string(#call) + ": " + hipGetErrorString(_rc)); \
    } while (0)

// Every entry point works on the engine's device and leaves the caller's current HIP device as it found it (PyTorch and
// other engines in the same process keep theirs).
struct DeviceScope {
    int prev;
    bool ok;
    explicit DeviceScope(int dev) : prev(-1) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        ok = (prev == dev) || hipSetDevice(dev) == hipSuccess;
        if (prev == dev) prev = -1;
    }
    ~DeviceScope() { if (prev >= 0) (void)hipSetDevice(prev); }
};
#define ON_DEVICE(e)                                     \
    DeviceScope _scope((e)->cfg.device_id);              \
    if (!_scope.ok) return fail(e, AZG_E_DEVICE, "hipSetDevice failed")

template <typename T>
static int dalloc(azg_engine* e, T** p, size_t n, std::vector<void*>& reg) {
    void* q = nullptr;
    hipError_t rc = hipMalloc(&q, n * sizeof(T) > 0 ? n * sizeof(T) : 16);
    if (rc != hipSuccess) return fail(e, AZG_E_DEVICE, std::string("hipMalloc: ") + hipGetErrorString(rc));
    reg.push_back(q);
    *p = (T*)q;
    return AZG_OK;
}

// ---- lock-step path (wide networks): a few grid-wide launches per simulation step
static int ls_prepare(azg_engine* e) {
    if (e->ls_hp == e->HP) return AZG_OK;
    for (void* p : e->ls_allocs) (void)hipFree(p);
    e->ls_allocs.clear();
    // tree groups padded to a multiple of 4: the tiled layer kernel works on 4 groups per workgroup
    const size_t B = e->cfg.n_trees, G = ((B + TREES_PER_WG - 1) / TREES_PER_WG + 3) / 4 * 4, HP = e->HP;
    float* obsT; float *a0, *a1, *parts; LsTree* tr; LsLane* ln;
    e->team_cnt_bytes = (((B + 31) / 32) * 8 * 32 + 32) * sizeof(unsigned);   // per team 8 counters 128 B apart, + the abort word
    if (dalloc(e, &e->d_team_cnt, e->team_cnt_bytes / 4, e->ls_allocs) ||
        dalloc(e, &obsT, G * 64, e->ls_allocs) || dalloc(e, &a0, G * HP * 16, e->ls_allocs) || dalloc(e, &a1, G * HP * 16, e->ls_allocs) ||
        dalloc(e, &parts, G * (HP / 64) * 64 * 4, e->ls_allocs) || dalloc(e, &tr, B, e->ls_allocs) ||
        dalloc(e, &ln, B * 16, e->ls_allocs))
        return AZG_E_DEVICE;
    e->ls.obsT = obsT; e->ls.act[0] = (f32x4*)a0; e->ls.act[1] = (f32x4*)a1; e->ls.parts = (f32x4*)parts;
    e->ls.tree = tr; e->ls.lane = ln;
    // the padding groups are computed like the others (their columns never mix with real ones): give them defined inputs
    if (hipMemset(obsT, 0, G * 64 * sizeof(float)) != hipSuccess || hipMemset(a0, 0, G * HP * 16 * sizeof(float)) != hipSuccess || hipMemset(a1, 0, G * HP * 16 * sizeof(float)) != hipSuccess) return AZG_E_DEVICE;
    e->ls_hp = e->HP;
    return AZG_OK;
}

// Rows (16 counters each) of the diagnostic stamp buffer: one per wave of the search kernel -- four waves per 4 trees at the least
// filled tile shape, eight waves per 16-tree workgroup (also when the batch has fewer than 16 trees) -- or eight counters per
// team-kernel workgroup (16 workgroups per 32 trees, at least one team).
static size_t stamp_rows(size_t B) {
    size_t r = ((B + 3) / 4) * 4;
    const size_t r8 = ((B + 15) / 16) * 8, team = ((B + 31) / 32) * 16 * 3 / 2;   // (team kernel: 8 + 16 counters per workgroup)
    if (r8 > r) r = r8;
    if (team > r) r = team;
    return r < 16 ? 16 : r;
}

static int env_digit(const char* name, int dflt) {
    const char* v = getenv(name);
    return (v && v[0] >= '0' && v[0] <= '9') ? v[0] - '0' : dflt;
}

static bool use_lockstep(const azg_engine* e) {
    if (e->opt.force_persistent) return false;
    if (e->P.in8) return false;   // (more than four network inputs: the lock-step / team kernels' first layer takes one k-step only)
    return e->HP >= 512 && e->n_hidden >= 2 && !e->P.layernorm;
}

// The persistent team kernel leaves instead of hanging when one of its waits times out (its workgroups were not all resident,
// e.g. another process holds part of the GPU): it raises a word that is read here, after the stream has been synchronised.
// The search is then run again, with the same search index, as per-layer launches -- which the engine uses from then on.
extern "C" int azg_search_resident(azg_engine* e);
static int team_check(azg_engine* e) {
    if (!e->team_pending) return AZG_OK;
    unsigned flag = 0;
    e->team_pending = 0;
    if (hipMemcpy(&flag, e->d_team_cnt + (e->team_cnt_bytes / 4 - 1), 4, hipMemcpyDeviceToHost) != hipSuccess)
        return fail(e, AZG_E_DEVICE, "reading the team kernel's status failed");
    if (flag) e->opt.ls_team = 0;
    if (flag == 0) return AZG_OK;
    e->team_fallbacks += 1;
    e->search_idx = e->team_search_idx;
    int rc = azg_search_resident(e);
    if (rc) return rc;
    if (hipStreamSynchronize(e->stream) != hipSuccess) return fail(e, AZG_E_DEVICE, "hipStreamSynchronize failed");
    return AZG_OK;
}

extern "C" {

int azg_abi_version(void) { return AZG_ABI_VERSION; }

const char* azg_last_error(const azg_engine* e) { return e ? e->err.c_str() : g_create_err.c_str(); }

void azg_engine_destroy(azg_engine* e) {
    if (!e) return;
    DeviceScope scope(e->cfg.device_id);
    if (e->stream) (void)hipStreamSynchronize(e->stream);
    for (void* p : e->dev_allocs) (void)hipFree(p);
    for (void* p : e->dist_allocs) (void)hipFree(p);
    if (e->d_wblob) (void)hipFree(e->d_wblob);
    if (e->d_wmap) (void)hipFree(e->d_wmap);
    if (e->h_res_block) (void)hipHostFree(e->h_res_block);
    if (e->d_eval) (void)hipFree(e->d_eval);
    for (void* p : e->sp_allocs) (void)hipFree(p);
    for (void* p : e->ls_allocs) (void)hipFree(p);
    if (e->ev0) (void)hipEventDestroy(e->ev0);
    if (e->ev1) (void)hipEventDestroy(e->ev1);
    if (e->stream) (void)hipStreamDestroy(e->stream);
    delete e;
}

// Two HIP runtimes in one process (PyTorch-ROCm wheels bundle their own libamdhip64 under the system library's SONAME; whichever is
// loaded first serves every library that comes later -- unless the other one was pulled in by path, as `import torch` does):
// the second one to initialise then finds no GPU or hangs.  Seen from here as two different libamdhip64 files in the process's
// memory map; reported instead of risking either.  Returns the number of distinct files and their paths.
static int mapped_hip_runtimes(std::string& paths) {
    // scanned once per process: the answer can only change by loading yet another runtime behind this library's back, and a process
    // that creates engines in a loop should not re-parse its memory map every time
    static int cached_n = -1;
    static std::string cached_paths;
    if (cached_n >= 0) { paths = cached_paths; return cached_n; }
    std::vector<std::string> seen;
    FILE* f = fopen("/proc/self/maps", "r");
    if (!f) return 0;
    char line[1024];
    while (fgets(line, sizeof line, f)) {
        if (!strstr(line, "libamdhip64")) continue;
        const char* p = strchr(line, '/');
        if (!p) continue;
        std::string path(p);
        while (!path.empty() && (path.back() == '\n' || path.back() == ' ')) path.pop_back();
        bool dup = false;
        for (const auto& q : seen) dup = dup || q == path;
        if (!dup) seen.push_back(path);
    }
    fclose(f);
    paths.clear();
    for (const auto& q : seen) paths += (paths.empty() ? "" : ", ") + q;
    cached_paths = paths;
    cached_n = (int)seen.size();
    return cached_n;
}

int azg_engine_create(const azg_config* cfg, azg_engine** out) {
    if (!cfg || !out) return fail(nullptr, AZG_E_INVALID, "null argument");
    {
        std::string rts;
        const char* allow = getenv("AZG_ALLOW_MULTI_HIP");
        if (mapped_hip_runtimes(rts) > 1 && !(allow && allow[0] == '1'))
            return fail(nullptr, AZG_E_DEVICE, ("two HIP runtimes are mapped in this process (" + rts + "): load PyTorch (import torch) BEFORE "
                                               "libazgym_hip.so so that both use PyTorch's copy; if they are meant to coexist (differing "
                                               "SONAMEs, a deliberate second copy), set AZG_ALLOW_MULTI_HIP=1").c_str());
    }
    if (cfg->struct_size != (int32_t)sizeof(azg_config)) return fail(nullptr, AZG_E_INVALID, "azg_config size mismatch");
    if (cfg->n_trees < 1 || cfg->n_sims < 1) return fail(nullptr, AZG_E_INVALID, "n_trees and n_sims must be >= 1");
    if (cfg->env_id < 0 || cfg->env_id > AZG_ENV_ACROBOT) return fail(nullptr, AZG_E_INVALID, "unknown env_id");
    const bool discrete_env = cfg->env_id == AZG_ENV_CARTPOLE || cfg->env_id == AZG_ENV_MOUNTAINCAR || cfg->env_id == AZG_ENV_ACROBOT;
    if (cfg->mode == AZG_MODE_DISCRETE && !discrete_env)
        return fail(nullptr, AZG_E_UNSUPPORTED, "discrete mode requires a discrete-action env (CartPole, MountainCar, Acrobot)");
    if (cfg->mode == AZG_MODE_CONTINUOUS && discrete_env)
        return fail(nullptr, AZG_E_UNSUPPORTED, "continuous mode requires a continuous-action env (Pendulum, MountainCarContinuous)");
    if (cfg->mode == AZG_MODE_DISCRETE && cfg->num_actions != (cfg->env_id == AZG_ENV_CARTPOLE ? 2 : 3))
        return fail(nullptr, AZG_E_INVALID, "num_actions does not match the env (CartPole 2, MountainCar 3, Acrobot 3)");
    if (cfg->tie_break != AZG_TIE_FIRST && cfg->tie_break != AZG_TIE_RANDOM) return fail(nullptr, AZG_E_INVALID, "unknown tie_break");
    // progressive widening (states.py:271-275): ceil(c_pw (n + 1)^kappa) children; with c_pw <= 0 no node is ever entitled to a child and the
    // reference's first selection takes the arg-max of an empty list (helpers.py:30-52 raises)
    if (cfg->mode == AZG_MODE_CONTINUOUS && !(cfg->c_pw > 0.0 && cfg->c_pw < 1e6 && cfg->kappa >= 0.0 && cfg->kappa <= 8.0))
        return fail(nullptr, AZG_E_INVALID, "c_pw must be > 0 and kappa >= 0 (progressive widening: ceil(c_pw (n + 1)^kappa) children)");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) return fail(nullptr, AZG_E_DEVICE, "no HIP device available");
    if (cfg->device_id < 0 || cfg->device_id >= ndev) return fail(nullptr, AZG_E_DEVICE, "device_id out of range");
    azg_engine* e = new azg_engine();
    e->cfg = *cfg;
    e->opt.force_persistent = env_digit("AZG_FORCE_PERSISTENT", 0) == 1;
    e->opt.force_stream_weights = env_digit("AZG_FORCE_STREAM_WEIGHTS", 0) == 1;
    e->opt.force_global_tree = env_digit("AZG_FORCE_GLOBAL_TREE", 0) == 1;
    e->opt.no_spec = env_digit("AZG_NO_SPEC", 0) == 1;
    e->opt.waves = env_digit("AZG_WAVES", 0);
    e->opt.groups = env_digit("AZG_GROUPS", 0);
    { const char* v = getenv("AZG_TRACE_CAP"); const int t = v ? atoi(v) : 0; e->opt.trace_cap = t > 0 ? t : 0; }
    { const char* v = getenv("AZG_TILE_TREES"); const int t = v ? atoi(v) : 0; e->opt.tile_trees = (t == 16 || t == 8) ? t : 0; }
    e->opt.ls_team = env_digit("AZG_LS_TEAM", 1);
    e->opt.team_wide = env_digit("AZG_TEAM_WIDE", 1);
    e->team_kc = 0; e->team_minb = 0; e->team_tt = 32;
    { const char* v = getenv("AZG_TEAM_TT"); e->opt.team_tt = v ? atoi(v) : 0; }
    { const char* v = getenv("AZG_TEAM_SPIN_LIMIT"); e->opt.team_spin_limit = v ? atol(v) : (1L << 23); }
    e->d_team_cnt = nullptr; e->team_cnt_bytes = 0; e->team_pending = 0; e->team_fallbacks = 0; e->team_search_idx = 0;
    e->kernel_form = -1; e->lds_exit = 0; e->lds_warned = 0; e->last_search_idx = 0; e->ms_kept = 0.0f; e->ms_kept_valid = 0;
    e->carry_max = 0;
    e->h_res_block = nullptr; e->d_res_block = nullptr; e->res_bytes = 0;
    e->d_wblob = nullptr; e->d_wmap = nullptr; e->w_floats = 0; e->dist_nd = -1; e->dist_ncomp = -1;
    e->d_eval = nullptr; e->eval_floats = 0;
    e->stream = nullptr; e->ev0 = e->ev1 = nullptr;
    e->publish_always = env_digit("AZG_PUBLISH_TREES", 0) == 1; e->publish_once = 0; e->published = 0; e->redo_ok = 0;
    e->mlp_ready = 0; e->searched = 0; e->results_valid = 0; e->search_idx = 0; e->last_ms = 0.0f; e->sp_on = 0; e->ls_hp = 0;
    e->S_env = (cfg->env_id == AZG_ENV_CARTPOLE || cfg->env_id == AZG_ENV_ACROBOT) ? 4 : 2;
    e->S_obs = cfg->env_id == AZG_ENV_ACROBOT ? 6 : (cfg->env_id == AZG_ENV_CARTPOLE ? 4 : ((cfg->env_id == AZG_ENV_MOUNTAINCAR || cfg->env_id == AZG_ENV_MOUNTAINCAR_CONT) ? 2 : 3));
    const int ns = cfg->n_sims;
    std::vector<int> pw(ns + 2, 0);
    if (cfg->mode == AZG_MODE_CONTINUOUS) {
        int kmax = 1;
        for (int n = 0; n < ns + 2; ++n) {
            // NodeContinuous.check_pw (states.py:271-273): python float pow + math.ceil, evaluated on the host with libm
            double v = std::ceil(cfg->c_pw * std::pow((double)(n + 1), cfg->kappa));
            if (v > 1e6) v = 1e6;
            pw[n] = (int)v;
            if (n < ns && pw[n] > kmax) kmax = pw[n];
        }
        e->Kmax = kmax;
        e->R = ns + 2;
        e->nd = 2;
    } else {
        e->Kmax = cfg->num_actions;
        e->R = 1 + cfg->num_actions * (ns + 1);
        e->nd = cfg->num_actions;
    }
    if (e->R > 32767) { delete e; return fail(nullptr, AZG_E_UNSUPPORTED, "tree too large: records per tree must be < 32768"); }
    if (cfg->tie_break == AZG_TIE_RANDOM && e->Kmax > 16) { delete e; return fail(nullptr, AZG_E_UNSUPPORTED, "tie_break random supports at most 16 children per node"); }
    e->Kp = (e->Kmax + 15) / 16 * 16;
    // sqrt(n+1) table: node visit counts reach n_sims (+ the carried root count in discrete mode; beyond 3 n_sims the kernel
    // computes the root's square root in place)
    e->tab_n = cfg->mode == AZG_MODE_CONTINUOUS ? ns + 2 : 4 * ns + 4;
    e->tree_lds = 0;
    e->dyn_lds = 0;
    DeviceScope scope(cfg->device_id);
    if (!scope.ok) { delete e; return fail(nullptr, AZG_E_DEVICE, "hipSetDevice failed"); }
    e->waves = 4; e->groups = 1; e->tile_trees = 16; e->spec = 0; e->n_cus = 256;
    (void)hipDeviceGetAttribute(&e->n_cus, hipDeviceAttributeMultiprocessorCount, cfg->device_id);
#define CK(x) do { int _r = (x); if (_r != AZG_OK) { g_create_err = e->err; azg_engine_destroy(e); return _r; } } while (0)
#define HK(call) do { hipError_t _rc = (call); if (_rc != hipSuccess) { g_create_err = std::string(#call) + ": " + hipGetErrorString(_rc); azg_engine_destroy(e); return AZG_E_DEVICE; } } while (0)
    HK(hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking));
    HK(hipEventCreate(&e->ev0));
    HK(hipEventCreate(&e->ev1));
    KParams& P = e->P;
    memset(&P, 0, sizeof(P));
    const size_t B = (size_t)cfg->n_trees, R = (size_t)e->R;
    RecL* hot; Cold* cold; double* edge_W; float* action; float* prior; unsigned short* child; int* n_rec;
    int* d_pw; double* d_sq;
    CK(dalloc(e, &hot, B * R, e->dev_allocs));
    CK(dalloc(e, &cold, B * R, e->dev_allocs));
    CK(dalloc(e, &edge_W, B * R, e->dev_allocs));
    CK(dalloc(e, &action, B * R, e->dev_allocs));
    CK(dalloc(e, &prior, B * R, e->dev_allocs));
    CK(dalloc(e, &child, B * R * e->Kp, e->dev_allocs));
    CK(dalloc(e, &n_rec, B, e->dev_allocs));
    CK(dalloc(e, &d_pw, (size_t)ns + 2, e->dev_allocs));
    CK(dalloc(e, &d_sq, (size_t)e->tab_n, e->dev_allocs));
    CK(dalloc(e, &e->d_roots, B * e->S_env, e->dev_allocs));
    CK(dalloc(e, &e->d_carry, B, e->dev_allocs));
    const size_t K = (size_t)e->Kmax;
    {
        // return_results' five arrays in ONE device block (float64 parts first) with a pinned host mirror: azg_results is one
        // device-to-host copy, not five
        e->res_bytes = B * K * 8 + B * 8 + B * K * 4 + B * K * 4 + B * 4;
        char* blk;
        CK(dalloc(e, &blk, e->res_bytes, e->dev_allocs));
        e->d_res_block = blk;
        e->d_Q = (double*)blk; e->d_vt = (double*)(blk + B * K * 8);
        e->d_actions = (float*)(blk + B * K * 8 + B * 8); e->d_counts = (int*)(blk + B * K * 8 + B * 8 + B * K * 4);
        e->d_nch = (int*)(blk + B * K * 8 + B * 8 + B * K * 8);
        HK(hipHostMalloc(&e->h_res_block, e->res_bytes, hipHostMallocDefault));
    }
    CK(dalloc(e, &e->d_child_n, B * K, e->dev_allocs));
    CK(dalloc(e, &e->d_child_state, B * K * e->S_env, e->dev_allocs));
    CK(dalloc(e, &e->d_rootV, B, e->dev_allocs));
    CK(dalloc(e, &e->d_rootdist, B * e->nd, e->dev_allocs));
    {
        unsigned long long* st;
        CK(dalloc(e, &st, stamp_rows(B) * 16, e->dev_allocs));   // (diagnostic builds: one row of 16 counters per wave)
        e->P.stamps = st;
    }
    std::vector<double> sq(e->tab_n);
    for (int n = 0; n < e->tab_n; ++n) sq[n] = std::sqrt((double)(n + 1));
    HK(hipMemcpy(d_pw, pw.data(), sizeof(int) * (ns + 2), hipMemcpyHostToDevice));
    HK(hipMemcpy(d_sq, sq.data(), sizeof(double) * e->tab_n, hipMemcpyHostToDevice));
    HK(hipMemset(hot, 0, B * R * sizeof(RecL)));
    HK(hipMemset(e->d_carry, 0, B * sizeof(int)));
    P.B = cfg->n_trees; P.n_sims = ns; P.R = e->R; P.Kp = e->Kp; P.A = cfg->num_actions; P.nd = e->nd;
    P.trace_cap = e->opt.trace_cap > 0 ? e->opt.trace_cap : 5;   // (config B on MI355X: 0.484 ms with 1, 0.388 with 3, 0.371 with 5 or 6, 0.43 with 8)
    P.tie_random = cfg->tie_break == AZG_TIE_RANDOM; P.env_id = cfg->env_id; P.v1 = cfg->env_id == AZG_ENV_PENDULUM_V1; P.tree_base = cfg->tree_id_base; P.mode = cfg->mode;
    P.c_uct = cfg->c_uct; P.gamma = cfg->gamma; P.epsilon = cfg->epsilon; P.reward_scale = cfg->reward_scale;
    P.c_uct_f = (float)cfg->c_uct; P.gamma_f = (float)cfg->gamma; P.bound_f = (float)cfg->action_bound;
    P.seed = cfg->seed; P.S = e->S_env; P.tab_n = e->tab_n; P.pw0 = cfg->mode == AZG_MODE_CONTINUOUS ? pw[0] : 0;
    P.roots = e->d_roots; P.carry = e->d_carry;
    P.hot = hot; P.cold = cold; P.edge_W = edge_W; P.action = action; P.prior = prior; P.child = child;
    P.n_rec = n_rec; P.pw_need = d_pw; P.sqrt_tab = d_sq;
    P.res_actions = e->d_actions; P.res_counts = e->d_counts; P.res_Q = e->d_Q; P.res_vt = e->d_vt; P.res_nch = e->d_nch;
    P.res_child_n = e->d_child_n; P.res_child_state = e->d_child_state; P.res_root_V = e->d_rootV; P.res_root_dist = e->d_rootdist;
    P.res_Kmax = e->Kmax; P.res_v_target = cfg->v_target;
    *out = e;
    return AZG_OK;
#undef CK
#undef HK
}

static int pad64(int n) { return (n + 63) / 64 * 64; }
static inline int unit_of(int i) { int t = i >> 4, r = (i >> 2) & 3, g = i & 3; return 16 * t + 4 * g + r; }

// ---- weights: from the caller's torch-layout blob to the kernels' operand layouts.
// The re-layout is a pure gather (every element of the engine's weight buffer is one element of the blob or a padding zero), so
// it is described ONCE per network shape by an index map (WeightMap::src: 1 + blob index, 0 = zero) and then applied either on the
// host (azg_set_weights: blob in host memory, one H2D copy of the result) or by a gather kernel (azg_set_weights_device: blob in
// device memory, e.g. the parameters PyTorch just updated or an RCCL broadcast buffer -- no host round trip).  Same map, same
// numbers, whichever side applies it.
static bool same_desc(const azg_mlp_desc& a, const azg_mlp_desc& b) {
    if (a.in_dim != b.in_dim || a.n_hidden != b.n_hidden || a.n_dist != b.n_dist || a.layernorm != b.layernorm) return false;
    for (int l = 0; l < a.n_hidden; ++l) if (a.hidden[l] != b.hidden[l]) return false;
    return true;
}

// validation shared by both entry points; HP_out = common padded hidden width, ncomp_out = mixture components (0: none)
static int check_desc(azg_engine* e, const azg_mlp_desc* d, size_t n_floats, int* HP_out, int* ncomp_out) {
    if (d->struct_size != (int32_t)sizeof(azg_mlp_desc)) return fail(e, AZG_E_INVALID, "azg_mlp_desc size mismatch");
    if (d->n_hidden < 1 || d->n_hidden > AZG_MAX_HIDDEN_LAYERS) return fail(e, AZG_E_INVALID, "n_hidden out of range");
    if (d->activation < 0 || d->activation > AZG_ACT_HARDSWISH) return fail(e, AZG_E_INVALID, "unknown activation");
    if (d->in_dim != e->S_obs) return fail(e, AZG_E_INVALID, "in_dim does not match the env observation");
    int ncomp = 0;
    if (e->cfg.mode == AZG_MODE_CONTINUOUS) {
        ncomp = d->num_components >= 2 ? d->num_components : 0;
        if (ncomp > 5) return fail(e, AZG_E_UNSUPPORTED, "at most 5 mixture components");
        if (d->n_dist != (ncomp ? 3 * ncomp : 2)) return fail(e, AZG_E_INVALID, "n_dist does not match num_components");
    } else if (d->n_dist != e->nd) return fail(e, AZG_E_INVALID, "n_dist does not match the engine mode");
    if (1 + d->n_dist > 16) return fail(e, AZG_E_UNSUPPORTED, "at most 15 distribution outputs");
    size_t need = 0;
    int k = d->in_dim, hmax = 0;
    for (int l = 0; l < d->n_hidden; ++l) {
        if (d->hidden[l] < 1 || d->hidden[l] > 4096) return fail(e, AZG_E_INVALID, "hidden width out of range");
        need += (size_t)d->hidden[l] * k + d->hidden[l] + (d->layernorm ? 2 * (size_t)d->hidden[l] : 0);
        k = d->hidden[l];
        if (k > hmax) hmax = k;
    }
    need += (size_t)(1 + d->n_dist) * k + (1 + d->n_dist);
    if (need != n_floats) return fail(e, AZG_E_INVALID, "weight blob size mismatch");
    const int HP = pad64(hmax);
    if (HP != 64 && HP != 128 && HP != 256 && HP != 512 && HP != 1024)
        return fail(e, AZG_E_UNSUPPORTED, "hidden width (padded to a multiple of 64) must be one of 64,128,256,512,1024");
    *HP_out = HP;
    *ncomp_out = ncomp;
    return AZG_OK;
}

// the index map of a network shape: where every element of the engine's weight buffer comes from
static void build_weight_map(const azg_mlp_desc* d, int HP, WeightMap& m) {
    typedef unsigned idx_t;   // 1 + index into the blob; 0: padding
    const int NT = HP / 16, S4 = HP / 16;
    // the blob's tensors as zero-padded [HP][Kp] index matrices (blob order = state_dict order: per layer weight, bias
    // (, LayerNorm weight, bias), then value head, distribution head)
    std::vector<std::vector<idx_t>> Wd(d->n_hidden), bd(d->n_hidden), gd(d->n_hidden), ed(d->n_hidden);
    idx_t p = 1;
    const int kp0 = d->in_dim > 4 ? 8 : 4;   // the first layer's input slots: one MFMA k-step, or two (five to eight inputs)
    int kt = d->in_dim, kp = kp0;
    for (int l = 0; l < d->n_hidden; ++l) {
        const int h = d->hidden[l];
        Wd[l].assign((size_t)HP * kp, 0);
        bd[l].assign(HP, 0);
        for (int n = 0; n < h; ++n)
            for (int kk = 0; kk < kt; ++kk) Wd[l][(size_t)n * kp + kk] = p + (idx_t)((size_t)n * kt + kk);
        p += (idx_t)((size_t)h * kt);
        for (int n = 0; n < h; ++n) bd[l][n] = p + n;
        p += h;
        gd[l].assign(HP, 0);
        ed[l].assign(HP, 0);
        if (d->layernorm) {
            for (int n = 0; n < h; ++n) gd[l][n] = p + n;
            p += h;
            for (int n = 0; n < h; ++n) ed[l][n] = p + n;
            p += h;
        }
        kt = h; kp = HP;
    }
    std::vector<idx_t> Wh((size_t)16 * HP, 0), bh(16, 0);
    for (int kk = 0; kk < kt; ++kk) Wh[kk] = p + kk;
    p += kt;
    bh[0] = p++;
    for (int o = 0; o < d->n_dist; ++o)
        for (int kk = 0; kk < kt; ++kk) Wh[(size_t)(1 + o) * HP + kk] = p + (idx_t)((size_t)o * kt + kk);
    p += (idx_t)((size_t)d->n_dist * kt);
    for (int o = 0; o < d->n_dist; ++o) bh[1 + o] = p + o;
    // MFMA operand layouts (lane l: row/col = l & 15, k-slot g = l >> 4; D register r of tile t = unit 16t + 4g + r); every
    // tensor starts 256-byte aligned in ONE buffer (one H2D copy / one gather per weight sync)
    std::vector<idx_t>& st = m.src;
    st.clear();
    auto reserve = [&](size_t n) { size_t off = st.size(); st.resize(off + (n + 63) / 64 * 64, 0); return off; };
    m.oW0 = reserve((size_t)NT * 64); m.ob0 = reserve((size_t)NT * 64 * 4);
    m.oW0b = reserve((size_t)NT * 64);        // inputs 4..7 (zeros for networks of at most four inputs)
    for (int t = 0; t < NT; ++t)
        for (int l = 0; l < 64; ++l) {
            const int row = 16 * t + (l & 15), g = l >> 4;
            st[m.oW0 + (size_t)t * 64 + l] = Wd[0][(size_t)row * kp0 + g];
            st[m.oW0b + (size_t)t * 64 + l] = kp0 == 8 ? Wd[0][(size_t)row * kp0 + 4 + g] : 0;
            for (int r = 0; r < 4; ++r) st[m.ob0 + ((size_t)t * 64 + l) * 4 + r] = bd[0][16 * t + 4 * g + r];
        }
    m.oW0u = reserve((size_t)HP * 4); m.ob0u = reserve((size_t)HP);
    for (int u = 0; u < HP; ++u) {   // (the VALU form of the first layer, team kernel: networks of at most four inputs only)
        for (int kk = 0; kk < 4; ++kk) st[m.oW0u + (size_t)u * 4 + kk] = Wd[0][(size_t)u * kp0 + kk];
        st[m.ob0u + u] = bd[0][u];
    }
    for (int l = 0; l < MAX_STREAM_LAYERS; ++l) { m.oWl[l] = m.obl[l] = m.olg[l] = m.olb[l] = 0; }
    for (int l = 1; l < d->n_hidden; ++l) {
        const size_t oW = reserve((size_t)NT * S4 * 64 * 4), ob = reserve((size_t)NT * 64 * 4);
        m.oWl[l - 1] = oW; m.obl[l - 1] = ob;
        for (int t = 0; t < NT; ++t)
            for (int l64 = 0; l64 < 64; ++l64) {
                const int row = 16 * t + (l64 & 15), g = l64 >> 4;
                for (int s4 = 0; s4 < S4; ++s4)
                    for (int j = 0; j < 4; ++j) {
                        const int i = 4 * (4 * s4 + j) + g;   // canonical position consumed by k-slot g of step 4*s4+j
                        st[oW + (((size_t)t * S4 + s4) * 64 + l64) * 4 + j] = Wd[l][(size_t)row * HP + unit_of(i)];
                    }
                for (int r = 0; r < 4; ++r) st[ob + ((size_t)t * 64 + l64) * 4 + r] = bd[l][16 * t + 4 * g + r];
            }
    }
    m.oWh = reserve((size_t)S4 * 64 * 4); m.obh = reserve(16);
    for (int s4 = 0; s4 < S4; ++s4)
        for (int l64 = 0; l64 < 64; ++l64) {
            const int o = l64 & 15, g = l64 >> 4;
            for (int j = 0; j < 4; ++j) {
                const int i = 4 * (4 * s4 + j) + g;
                st[m.oWh + ((size_t)s4 * 64 + l64) * 4 + j] = Wh[(size_t)o * HP + unit_of(i)];
            }
        }
    for (int o = 0; o < 16; ++o) st[m.obh + o] = bh[o];
    if (d->layernorm)
        for (int l = 0; l < d->n_hidden; ++l) {
            m.olg[l] = reserve((size_t)NT * 64 * 4); m.olb[l] = reserve((size_t)NT * 64 * 4);
            for (int t = 0; t < NT; ++t)
                for (int l64 = 0; l64 < 64; ++l64)
                    for (int r = 0; r < 4; ++r) {
                        st[m.olg[l] + ((size_t)t * 64 + l64) * 4 + r] = gd[l][16 * t + 4 * (l64 >> 4) + r];
                        st[m.olb[l] + ((size_t)t * 64 + l64) * 4 + r] = ed[l][16 * t + 4 * (l64 >> 4) + r];
                    }
        }
    m.desc = *d;
    m.HP = HP;
    m.valid = true;
}

__global__ __launch_bounds__(256) void weight_gather_kernel(const unsigned* __restrict__ src, const float* __restrict__ blob,
                                                            float* __restrict__ out, size_t n) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) { const unsigned s = src[i]; out[i] = s ? blob[s - 1] : 0.0f; }
}

// Common body of azg_set_weights / azg_set_weights_device: `blob` is a host pointer (on_device false) or a device pointer whose
// contents are complete (its producer's stream synchronised or otherwise ordered before this call).
static int set_weights_impl(azg_engine* e, const azg_mlp_desc* d, const float* blob, size_t n_floats, bool on_device) {
    if (!e || !d || !blob) return AZG_E_INVALID;
    int HP = 0, ncomp = 0;
    { int rc = check_desc(e, d, n_floats, &HP, &ncomp); if (rc) return rc; }
    ON_DEVICE(e);
    HIPCHK(e, hipStreamSynchronize(e->stream));
    { int trc = team_check(e); if (trc) return trc; }   // an abandoned team search is redone with the weights it was started with
    // nothing below may leave a half-updated weight set usable: the flag goes up again only on success
    e->mlp_ready = 0;
    e->results_valid = 0;
    e->redo_ok = 0;
    WeightMap& m = e->wmap;
    if (!m.valid || m.HP != HP || !same_desc(m.desc, *d)) {
        m.valid = false;
        build_weight_map(d, HP, m);
        if (e->d_wmap) { (void)hipFree(e->d_wmap); e->d_wmap = nullptr; }   // (uploaded when the device path first needs it)
    }
    const size_t n_out_f = m.src.size();
    if (n_out_f != e->w_floats || !e->d_wblob) {
        if (e->d_wblob) (void)hipFree(e->d_wblob);
        e->d_wblob = nullptr; e->w_floats = 0;
        void* q = nullptr;
        HIPCHK(e, hipMalloc(&q, n_out_f * sizeof(float)));
        e->d_wblob = (float*)q; e->w_floats = n_out_f;
    }
    if (on_device) {
        if (!e->d_wmap) {
            void* q = nullptr;
            HIPCHK(e, hipMalloc(&q, n_out_f * sizeof(unsigned)));
            e->d_wmap = (unsigned*)q;
            HIPCHK(e, hipMemcpy(e->d_wmap, m.src.data(), n_out_f * sizeof(unsigned), hipMemcpyHostToDevice));
        }
        hipLaunchKernelGGL(weight_gather_kernel, dim3((unsigned)((n_out_f + 255) / 256)), dim3(256), 0, e->stream, e->d_wmap, blob, e->d_wblob, n_out_f);
        HIPCHK(e, hipGetLastError());
        HIPCHK(e, hipStreamSynchronize(e->stream));   // the caller may overwrite its blob as soon as this returns
    } else {
        std::vector<float>& st = e->w_stage;
        st.resize(n_out_f);
        const unsigned* src = m.src.data();
        for (size_t i = 0; i < n_out_f; ++i) st[i] = src[i] ? blob[src[i] - 1] : 0.0f;
        HIPCHK(e, hipMemcpy(e->d_wblob, st.data(), n_out_f * sizeof(float), hipMemcpyHostToDevice));
    }
    const float* wb = e->d_wblob;
    e->P.W0u = (const f32x4*)(wb + m.oW0u);
    e->P.b0u = (const f32x4*)(wb + m.ob0u);
    e->P.W0 = wb + m.oW0;
    e->P.W0b = wb + m.oW0b;
    e->P.in8 = d->in_dim > 4 ? 1 : 0;
    e->P.b0 = (const f32x4*)(wb + m.ob0);
    for (int l = 0; l < MAX_STREAM_LAYERS; ++l) {
        const bool on = l + 1 < d->n_hidden;
        e->P.Wl[l] = on ? (const f32x4*)(wb + m.oWl[l]) : nullptr;
        e->P.bl[l] = on ? (const f32x4*)(wb + m.obl[l]) : nullptr;
    }
    e->P.Whead = (const f32x4*)(wb + m.oWh);
    e->P.bhead = wb + m.obh;
    e->P.layernorm = d->layernorm ? 1 : 0;
    for (int l = 0; l < MAX_STREAM_LAYERS; ++l) {
        e->P.Htrue[l] = l < d->n_hidden ? d->hidden[l] : 0;
        e->P.lng[l] = (d->layernorm && l < d->n_hidden) ? (const f32x4*)(wb + m.olg[l]) : nullptr;
        e->P.lnb[l] = (d->layernorm && l < d->n_hidden) ? (const f32x4*)(wb + m.olb[l]) : nullptr;
    }
    if (e->cfg.mode == AZG_MODE_CONTINUOUS && (d->n_dist != e->dist_nd || ncomp != e->dist_ncomp)) {
        // the per-node mixture cache and the root-distribution staging buffer are sized by the head: rebuilt only when it changes
        // (the last search's cached distributions go with them)
        for (void* q : e->dist_allocs) (void)hipFree(q);
        e->dist_allocs.clear();
        e->dist_nd = -1; e->searched = 0;
        float* g = nullptr;
        if (ncomp) { if (dalloc(e, &g, (size_t)e->cfg.n_trees * e->R * 3 * GMM_MAXC, e->dist_allocs)) return AZG_E_DEVICE; }
        float* rd = nullptr;
        if (dalloc(e, &rd, (size_t)e->cfg.n_trees * d->n_dist, e->dist_allocs)) return AZG_E_DEVICE;
        e->P.gmm = g; e->P.ncomp = ncomp; e->d_rootdist = rd; e->P.res_root_dist = rd; e->nd = d->n_dist; e->P.nd = d->n_dist;
        e->dist_nd = d->n_dist; e->dist_ncomp = ncomp;
    }
    const int n_out = 1 + d->n_dist;
    e->HP = HP; e->n_hidden = d->n_hidden; e->n_out = n_out; e->act = d->activation;
    e->P.n_hidden = d->n_hidden; e->P.n_out = n_out; e->P.act = d->activation; e->P.ls_min = d->log_std_min; e->P.ls_max = d->log_std_max;
    // hidden->hidden layers that fit the register file stay there for the whole search
    int nhh = d->n_hidden - 1;
    int regs = nhh * (HP * HP / 256);   // VGPRs per lane: each of the 4 waves holds a quarter of every layer
    e->nreg = (nhh >= 1 && nhh <= 2 && regs <= 288) ? nhh : 0;   // (three and more hidden->hidden layers: streamed, any depth)
    // LayerNorm and the rare activations live in the weight-streaming kernels only (keeps the register-resident kernels lean)
    if (d->layernorm || (d->activation != AZG_ACT_RELU && d->activation != AZG_ACT_ELU)) e->nreg = 0;
    if (e->opt.force_stream_weights) e->nreg = 0;
    e->mlp_ready = 1;
    return AZG_OK;
}

int azg_set_weights(azg_engine* e, const azg_mlp_desc* d, const float* blob, size_t n_floats) {
    return set_weights_impl(e, d, blob, n_floats, false);
}

int azg_set_weights_device(azg_engine* e, const azg_mlp_desc* d, const float* device_blob, size_t n_floats) {
    return set_weights_impl(e, d, device_blob, n_floats, true);
}

int azg_set_search_index(azg_engine* e, uint32_t idx) { if (!e) return AZG_E_INVALID; e->search_idx = idx; return AZG_OK; }

int azg_upload_roots(azg_engine* e, const double* roots, const int32_t* carry) {
    if (!e || !roots) return AZG_E_INVALID;
    const int B = e->cfg.n_trees, S = e->S_env;
    if (e->cfg.env_id == AZG_ENV_CARTPOLE) {
        const double theta_thr = 12.0 * 2.0 * 3.141592653589793 / 360.0, x_thr = 2.4;
        for (int i = 0; i < B; ++i) {
            const double* s = roots + (size_t)i * S;
            if ((s[0] < -x_thr) || (s[0] > x_thr) || (s[2] < -theta_thr) || (s[2] > theta_thr))
                return fail(e, AZG_E_TERMINAL_ROOT, "Can't do tree search from a terminal node");
        }
    } else if (e->cfg.env_id == AZG_ENV_ACROBOT) {
        for (int i = 0; i < B; ++i)
            if (azg_acrobot_terminal(roots + (size_t)i * S)) return fail(e, AZG_E_TERMINAL_ROOT, "Can't do tree search from a terminal node");
    } else if (e->cfg.env_id == AZG_ENV_MOUNTAINCAR || e->cfg.env_id == AZG_ENV_MOUNTAINCAR_CONT) {
        // (mcts.py:382-383, 599-600; the flag is at 0.5 in MountainCar-v0, at 0.45 in MountainCarContinuous-v0)
        const double goal = e->cfg.env_id == AZG_ENV_MOUNTAINCAR ? 0.5 : 0.45;
        for (int i = 0; i < B; ++i) {
            const double* s = roots + (size_t)i * S;
            if (s[0] >= goal && s[1] >= 0.0) return fail(e, AZG_E_TERMINAL_ROOT, "Can't do tree search from a terminal node");
        }
    }
    int cmax = 0;
    if (carry)
        for (int i = 0; i < B; ++i) {
            if (carry[i] < 0 || carry[i] > (1 << 30)) return fail(e, AZG_E_INVALID, "root_n_carry out of range");
            // MCTSContinuous never reuses a tree (no forward(): mcts.py:589-600 builds a fresh root for every search); its kernels' sqrt
            // table ends at n_sims + 1
            if (carry[i] != 0 && e->cfg.mode == AZG_MODE_CONTINUOUS)
                return fail(e, AZG_E_INVALID, "root_n_carry: continuous searches start from a fresh root (no carried count)");
            if (carry[i] > cmax) cmax = carry[i];
        }
    ON_DEVICE(e);
    if (e->team_pending) {   // an abandoned team search is redone on the roots it was started with
        HIPCHK(e, hipStreamSynchronize(e->stream));
        int trc = team_check(e);
        if (trc) return trc;
    }
    e->carry_max = cmax;
    e->redo_ok = 0;
    HIPCHK(e, hipMemcpyAsync(e->d_roots, roots, sizeof(double) * (size_t)B * S, hipMemcpyHostToDevice, e->stream));
    if (carry) HIPCHK(e, hipMemcpyAsync(e->d_carry, carry, sizeof(int) * (size_t)B, hipMemcpyHostToDevice, e->stream));
    else HIPCHK(e, hipMemsetAsync(e->d_carry, 0, sizeof(int) * (size_t)B, e->stream));
    HIPCHK(e, hipStreamSynchronize(e->stream));
    return AZG_OK;
}

int azg_search_resident(azg_engine* e) {
    if (!e) return AZG_E_INVALID;
    if (!e->mlp_ready) return fail(e, AZG_E_STATE, "azg_set_weights has not been called");
    ON_DEVICE(e);
    e->P.search_idx = e->search_idx;
    e->last_search_idx = e->search_idx;
    e->ms_kept_valid = 0;
    e->P.publish = (e->publish_always || e->publish_once) ? 1 : 0;
    e->team_search_idx = e->search_idx;
    const bool lockstep = use_lockstep(e);
    if (lockstep) { int prc = ls_prepare(e); if (prc) return prc; }
    e->launch_timed = 0;
    if (lockstep) HIPCHK(e, hipEventRecord(e->ev0, e->stream));   // (several launches; the one-launch search kernel stamps ev0 / ev1 itself)
    hipError_t rc;
    const bool cartpole = e->cfg.mode == AZG_MODE_DISCRETE;   // the discrete family's kernels (CartPole, MountainCar)
    const bool mcc = e->cfg.env_id == AZG_ENV_MOUNTAINCAR_CONT;   // the continuous family whose episodes end (env.cuh: EnvFamily)
    if (lockstep) rc = cartpole ? azg_ls_dispatch_cartpole(e) : (mcc ? azg_ls_dispatch_mcc(e) : azg_ls_dispatch_pendulum(e));
    else if (e->cfg.env_id == AZG_ENV_ACROBOT) rc = azg_dispatch_acrobot(e);
    else if (cartpole) rc = azg_dispatch_cartpole(e);
    else if (mcc) rc = azg_dispatch_mcc(e);
    else rc = e->HP <= 128 ? azg_dispatch_pendulum_small(e) : azg_dispatch_pendulum_large(e);
    if (rc != hipSuccess) return fail(e, AZG_E_DEVICE, std::string("search kernel launch: ") + hipGetErrorString(rc));
    if (!e->launch_timed) HIPCHK(e, hipEventRecord(e->ev1, e->stream));
    e->search_idx += 1;
    e->searched = 1;
    e->results_valid = e->kernel_form == 0 ? 1 : 0;   // the one-launch search kernel writes return_results in its epilogue
    e->published = (e->kernel_form != 0 || e->tree_lds == TS_GLOBAL || e->P.publish) ? 1 : 0;
    e->redo_ok = 1;
    if (e->kernel_form == 0 && e->tree_lds == TS_GLOBAL && e->lds_exit != AZG_LDS_EXIT_FORCED && !e->lds_warned) {
        e->lds_warned = 1;
        static const char* why[] = {"", "more than 511 records per tree (n_sims + 2)", "more than 16 children per node (c_pw / kappa)",
                                    "the workgroup's LDS plan exceeds the CU's 160 KB", ""};
        if (!getenv("AZG_QUIET"))
            fprintf(stderr, "azgym: the trees of this search do not fit LDS residency (%s): they are kept in global memory -- same results, "
                            "slower tree walk (azg_search_info; once per engine)\n", why[e->lds_exit & 3]);
    }
    return AZG_OK;
}

int azg_sync(azg_engine* e) {
    if (!e) return AZG_E_INVALID;
    ON_DEVICE(e);
    HIPCHK(e, hipStreamSynchronize(e->stream));
    return team_check(e);
}

int azg_search(azg_engine* e, const double* roots, const int32_t* carry) {
    int rc = azg_upload_roots(e, roots, carry);
    if (rc) return rc;
    rc = azg_search_resident(e);
    if (rc) return rc;
    return azg_sync(e);
}

int azg_last_search_ms(azg_engine* e, float* ms) {
    if (!e || !ms) return AZG_E_INVALID;
    if (!e->searched) return fail(e, AZG_E_STATE, "no search has run");
    ON_DEVICE(e);
    HIPCHK(e, hipEventSynchronize(e->ev1));
    if (e->team_pending) {   // (the time of the launches that redid an abandoned team search, not of the abandoned launch)
        HIPCHK(e, hipStreamSynchronize(e->stream));
        int trc = team_check(e);
        if (trc) return trc;
    }
    if (e->ms_kept_valid) { *ms = e->ms_kept; return AZG_OK; }   // (azg_dump_tree re-ran the search since: the time of the search itself)
    HIPCHK(e, hipEventElapsedTime(ms, e->ev0, e->ev1));
    return AZG_OK;
}

static int launch_results(azg_engine* e) {
    if (!e->searched) return fail(e, AZG_E_STATE, "no search has run");
    if (e->results_valid) return AZG_OK;
    if (e->team_pending) {
        HIPCHK(e, hipStreamSynchronize(e->stream));
        int trc = team_check(e);
        if (trc) return trc;
    }
    int B = e->cfg.n_trees;
    hipLaunchKernelGGL(results_kernel, dim3((B + RS_TREES - 1) / RS_TREES), dim3(16 * RS_TREES), 0, e->stream, e->P);
    HIPCHK(e, hipGetLastError());
    e->results_valid = 1;   // (in stream order: whatever reads the buffers is ordered after this launch)
    return AZG_OK;
}

static int gather_results(azg_engine* e) {
    ON_DEVICE(e);
    int rc = launch_results(e);
    if (rc) return rc;
    HIPCHK(e, hipStreamSynchronize(e->stream));
    return AZG_OK;
}

#define D2H(dst, src, bytes) do { if (dst) HIPCHK(e, hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost)); } while (0)

int azg_results(azg_engine* e, float* actions, int32_t* counts, double* Q, double* v_target, int32_t* n_children) {
    if (!e) return AZG_E_INVALID;
    ON_DEVICE(e);
    int rc = launch_results(e);
    if (rc) return rc;
    const size_t B = e->cfg.n_trees, K = e->Kmax;
    // one copy of the whole block into pinned memory behind the search (and results) on the engine's stream, then host copies
    HIPCHK(e, hipMemcpyAsync(e->h_res_block, e->d_res_block, e->res_bytes, hipMemcpyDeviceToHost, e->stream));
    HIPCHK(e, hipStreamSynchronize(e->stream));
    const char* h = (const char*)e->h_res_block;
    if (Q) memcpy(Q, h, B * K * 8);
    if (v_target) memcpy(v_target, h + B * K * 8, B * 8);
    if (actions) memcpy(actions, h + B * K * 8 + B * 8, B * K * 4);
    if (counts) memcpy(counts, h + B * K * 8 + B * 8 + B * K * 4, B * K * 4);
    if (n_children) memcpy(n_children, h + B * K * 8 + B * 8 + B * K * 8, B * 4);
    return AZG_OK;
}

int azg_results_resident(azg_engine* e, const float** actions, const int32_t** counts, const double** Q, const double** v_target,
                         const int32_t** n_children) {
    if (!e) return AZG_E_INVALID;
    ON_DEVICE(e);
    int rc = launch_results(e);
    if (rc) return rc;
    if (actions) *actions = e->d_actions;
    if (counts) *counts = e->d_counts;
    if (Q) *Q = e->d_Q;
    if (v_target) *v_target = e->d_vt;
    if (n_children) *n_children = e->d_nch;
    return AZG_OK;
}

int azg_root_children(azg_engine* e, int32_t* child_n, double* child_state) {
    if (!e) return AZG_E_INVALID;
    int rc = gather_results(e);
    if (rc) return rc;
    size_t B = e->cfg.n_trees, K = e->Kmax;
    D2H(child_n, e->d_child_n, B * K * 4);
    D2H(child_state, e->d_child_state, B * K * e->S_env * 8);
    return AZG_OK;
}

int azg_root_eval(azg_engine* e, float* value, float* dist) {
    if (!e) return AZG_E_INVALID;
    int rc = gather_results(e);
    if (rc) return rc;
    size_t B = e->cfg.n_trees;
    D2H(value, e->d_rootV, B * 4);
    D2H(dist, e->d_rootdist, B * e->nd * 4);
    return AZG_OK;
}

int azg_mlp_eval(azg_engine* e, const float* obs, size_t n, float* value, float* dist, float* raw) {
    if (!e || !obs) return AZG_E_INVALID;
    if (!e->mlp_ready) return fail(e, AZG_E_STATE, "azg_set_weights has not been called");
    if (n == 0) return AZG_OK;
    if (n > (size_t)1 << 24) return fail(e, AZG_E_INVALID, "too many observations in one call");
    ON_DEVICE(e);
    const size_t So = e->S_obs, nd = e->nd;
    const size_t need = n * (So + 1 + nd + 1 + nd);
    if (need > e->eval_floats) {
        if (e->d_eval) (void)hipFree(e->d_eval);
        e->d_eval = nullptr; e->eval_floats = 0;
        void* q = nullptr;
        HIPCHK(e, hipMalloc(&q, need * sizeof(float)));
        e->d_eval = (float*)q; e->eval_floats = need;
    }
    float* d_obs = e->d_eval;
    float* d_v = d_obs + n * So;
    float* d_d = d_v + n;
    float* d_r = d_d + n * nd;
    HIPCHK(e, hipMemcpyAsync(d_obs, obs, n * So * 4, hipMemcpyHostToDevice, e->stream));
    hipError_t rc = azg_dispatch_mlp_eval(e, d_obs, (int)n, d_v, d_d, d_r);
    if (rc != hipSuccess) return fail(e, AZG_E_DEVICE, std::string("mlp_eval kernel launch: ") + hipGetErrorString(rc));
    HIPCHK(e, hipStreamSynchronize(e->stream));
    D2H(value, d_v, n * 4);
    D2H(dist, d_d, n * nd * 4);
    D2H(raw, d_r, n * (nd + 1) * 4);
    return AZG_OK;
}

int azg_dump_tree(azg_engine* e, int32_t* n_records, int32_t* parent, int32_t* edge_n, double* edge_W, double* edge_Q,
                  float* edge_action, int32_t* node_n, double* node_r, float* node_V, uint8_t* node_flags) {
    if (!e) return AZG_E_INVALID;
    if (!e->searched) return fail(e, AZG_E_STATE, "no search has run");
    ON_DEVICE(e);
    HIPCHK(e, hipStreamSynchronize(e->stream));
    { int trc = team_check(e); if (trc) return trc; }
    if (!e->published) {
        // The search kernel keeps its trees in LDS and, on the product path, writes out only return_results.  A dump re-runs the
        // last search -- same roots, weights and search index: the same trees, bit for bit -- with the trees published this once.
        if (!e->redo_ok)
            return fail(e, AZG_E_STATE, "the last search's trees were not written out and its inputs have changed since (self-play step, new "
                                        "roots or weights): dump right after the search, or set AZG_PUBLISH_TREES=1");
        // (with the index THAT search ran under -- azg_set_search_index may have moved the counter since -- and without disturbing
        // the counter or the timing of the search being inspected: azg_last_search_ms keeps reporting that search, not the re-run)
        float ms_before = 0.0f;
        HIPCHK(e, hipEventElapsedTime(&ms_before, e->ev0, e->ev1));
        const uint32_t idx_now = e->search_idx;
        e->publish_once = 1;
        e->search_idx = e->last_search_idx;
        int rc = azg_search_resident(e);
        e->publish_once = 0;
        e->search_idx = idx_now;
        if (rc) return rc;
        HIPCHK(e, hipStreamSynchronize(e->stream));
        { int trc = team_check(e); if (trc) return trc; }
        e->ms_kept = ms_before; e->ms_kept_valid = 1;
    }
    size_t B = e->cfg.n_trees, R = e->R;
    std::vector<RecL> hot(B * R);
    std::vector<Cold> cold(B * R);
    std::vector<int> nrec(B);
    std::vector<double> ew(B * R);
    std::vector<float> ac(B * R);
    HIPCHK(e, hipMemcpy(hot.data(), e->P.hot, B * R * sizeof(RecL), hipMemcpyDeviceToHost));
    HIPCHK(e, hipMemcpy(cold.data(), e->P.cold, B * R * sizeof(Cold), hipMemcpyDeviceToHost));
    HIPCHK(e, hipMemcpy(nrec.data(), e->P.n_rec, B * 4, hipMemcpyDeviceToHost));
    HIPCHK(e, hipMemcpy(ew.data(), e->P.edge_W, B * R * 8, hipMemcpyDeviceToHost));
    HIPCHK(e, hipMemcpy(ac.data(), e->P.action, B * R * 4, hipMemcpyDeviceToHost));
    const bool cont = e->cfg.mode == AZG_MODE_CONTINUOUS;
    const int A = e->cfg.num_actions;
    for (size_t i = 0; i < B; ++i) {
        if (n_records) n_records[i] = nrec[i];
        for (size_t j = 0; j < R; ++j) {
            size_t o = i * R + j;
            bool in = (int)j < nrec[i];
            const RecL& h = hot[o];
            bool ex = in && (h.flags & FLAG_EXPANDED);
            if (parent) parent[o] = in ? (j == 0 ? -1 : h.parent) : 0;
            if (edge_n) edge_n[o] = in ? h.edge_n : 0;
            if (edge_W) edge_W[o] = in ? ew[o] : 0.0;
            if (edge_Q) edge_Q[o] = in ? h.Q : 0.0;
            if (edge_action) edge_action[o] = in ? (cont ? ac[o] : (j == 0 ? 0.0f : (float)((j - 1) % A))) : 0.0f;
            if (node_n) node_n[o] = in ? h.node_n : 0;
            if (node_r) node_r[o] = ex ? cold[o].r : 0.0;
            if (node_V) node_V[o] = ex ? cold[o].V : 0.0f;
            if (node_flags) node_flags[o] = in ? h.flags : 0;
        }
    }
    return AZG_OK;
}

// the kernel(s) of the last search as rocprofv3 names them (template arguments: ENV, HP, NREG, tree storage, mixture head, waves, tree
// groups, trees per group, compile-time specialisation; team kernel: ..., staging chunk length, workgroups per CU -- see search_kernel.cuh / team.cuh)
static int kernel_name(const azg_engine* e, char* buf, size_t n) {
    // (ENV = 0: the CartPole / MountainCar family, 5: Acrobot, 2: both Pendulum versions, 4: MountainCarContinuous)
    const int env = e->cfg.mode == AZG_MODE_DISCRETE ? (e->cfg.env_id == AZG_ENV_ACROBOT ? 5 : 0) : (e->cfg.env_id == AZG_ENV_MOUNTAINCAR_CONT ? 4 : 2);
    const char* gmm = (env != 0 && e->P.ncomp >= 2) ? "true" : "false";
    switch (e->kernel_form) {
        case 0: return snprintf(buf, n, "search_kernel<%d, %d, %d, %d, %s, %d, %d, %d, %d>", env, e->HP, e->nreg, e->tree_lds, gmm, e->waves, e->groups, e->tile_trees, e->spec);
        case 1: return snprintf(buf, n, "ls_tree_kernel<%d, ...> + ls_layer0_kernel + ls_hidden_tiled_kernel<%d, ...> per simulation step", env, e->HP);
        case 2: return snprintf(buf, n, "ls_team_kernel<%d, %d, %s, %d, %d, %d, %d, %d>", env, e->HP, gmm, e->tree_lds, e->team_kc, e->team_minb, e->spec, e->team_tt);
        default: return snprintf(buf, n, "(no search yet)");
    }
}

// include/azgym.h: what the last search ran as (kernel form, tree residency and why, team fall-backs)
int azg_search_info(azg_engine* e, azg_search_report* info) {
    if (!e || !info) return AZG_E_INVALID;
    if (info->struct_size != (int32_t)sizeof(azg_search_report)) return fail(e, AZG_E_INVALID, "azg_search_info: struct_size mismatch");
    if (e->searched && e->team_pending) {   // (an abandoned team search is redone before its form is reported)
        ON_DEVICE(e);
        HIPCHK(e, hipStreamSynchronize(e->stream));
        int trc = team_check(e);
        if (trc) return trc;
    }
    memset(info, 0, sizeof(*info));
    info->struct_size = (int32_t)sizeof(azg_search_report);
    info->kernel_form = e->searched ? e->kernel_form : AZG_FORM_NONE;
    info->max_records = e->R; info->max_children = e->Kmax;
    info->team_fallbacks = e->team_fallbacks;
    if (!e->searched) { kernel_name(e, info->kernel_name, sizeof(info->kernel_name)); info->lds_exit = AZG_LDS_RESIDENT; return AZG_OK; }
    const bool persistent = e->kernel_form == 0;
    info->tree_storage = persistent ? e->tree_lds : AZG_TREES_GLOBAL;
    info->lds_exit = !persistent ? AZG_LDS_NOT_APPLICABLE : (e->tree_lds != TS_GLOBAL ? AZG_LDS_RESIDENT : e->lds_exit);
    info->spec = e->spec;
    if (persistent) { info->waves = e->waves; info->groups = e->groups; info->tile_trees = e->tile_trees; }
    if (e->kernel_form == 2) { info->team_trees = e->team_tt; info->team_per_cu = e->team_minb; info->team_parts = e->team_parts; }
    float ms = 0.0f;
    int rc = azg_last_search_ms(e, &ms);
    if (rc) return rc;
    info->last_ms = ms;
    kernel_name(e, info->kernel_name, sizeof(info->kernel_name));
    return AZG_OK;
}

// diagnostic (-DAZG_STAMPS builds): per-wave cycle sums [n_workgroups*4][16]; returns the number of rows
int azg_debug_stamps(azg_engine* e, unsigned long long* out, size_t max_rows) {
    if (!e || !out) return AZG_E_INVALID;
    size_t rows = stamp_rows((size_t)e->cfg.n_trees);
    if (rows > max_rows) rows = max_rows;
    ON_DEVICE(e);
    HIPCHK(e, hipStreamSynchronize(e->stream));
    HIPCHK(e, hipMemcpy(out, e->P.stamps, rows * 16 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    return (int)rows;
}

int azg_max_children(const azg_engine* e) { return e ? e->Kmax : AZG_E_INVALID; }
int azg_max_records(const azg_engine* e) { return e ? e->R : AZG_E_INVALID; }
int azg_env_state_dim(const azg_engine* e) { return e ? e->S_env : AZG_E_INVALID; }
int azg_obs_dim(const azg_engine* e) { return e ? e->S_obs : AZG_E_INVALID; }

int azg_synthetic_roots(azg_engine* e, double* roots) {
    if (!e || !roots) return AZG_E_INVALID;
    for (int i = 0; i < e->cfg.n_trees; ++i)
        azg_reset_state(e->cfg.seed, (uint32_t)(e->cfg.tree_id_base + i), 0u, azg_reset_kind(e->cfg.env_id), roots + (size_t)i * e->S_env);
    return AZG_OK;
}

int azg_selfplay_row_len(const azg_engine* e) { return e ? e->S_obs + 3 * e->Kmax + 1 : AZG_E_INVALID; }

int azg_selfplay_begin_ex(azg_engine* e, const azg_selfplay_config* c) {
    if (!e || !c) return AZG_E_INVALID;
    if (c->struct_size != (int32_t)sizeof(azg_selfplay_config)) return fail(e, AZG_E_INVALID, "azg_selfplay_config size mismatch");
    if (c->max_episode_length < 1 || c->capacity_steps < 1) return fail(e, AZG_E_INVALID, "max_episode_length and capacity_steps must be >= 1");
    if (c->final_selection != AZG_FS_MAX_VISIT && c->final_selection != AZG_FS_MAX_VALUE) return fail(e, AZG_E_INVALID, "unknown final_selection");
    if (c->ring_mode != AZG_RING_STOP && c->ring_mode != AZG_RING_FIFO) return fail(e, AZG_E_INVALID, "unknown ring_mode");
    if (!(c->temperature > 0.0)) return fail(e, AZG_E_INVALID, "temperature must be > 0");
    if (c->agent_epsilon < 0.0 || c->agent_epsilon > 1.0) return fail(e, AZG_E_INVALID, "agent_epsilon must be in [0, 1]");
    const bool discrete = e->cfg.mode == AZG_MODE_DISCRETE;
    if (discrete && c->final_selection == AZG_FS_MAX_VALUE && c->temperature != 1.0)
        return fail(e, AZG_E_UNSUPPORTED, "final_selection max_value on the device supports temperature 1 only");
    ON_DEVICE(e);
    HIPCHK(e, hipStreamSynchronize(e->stream));
    for (void* p : e->sp_allocs) (void)hipFree(p);
    e->sp_allocs.clear();
    e->sp_on = 0;
    const size_t B = e->cfg.n_trees;
    e->sp_row = e->S_obs + 3 * e->Kmax + 1;
    if (dalloc(e, &e->d_sp_t, B, e->sp_allocs) || dalloc(e, &e->d_sp_episode, B, e->sp_allocs) || dalloc(e, &e->d_sp_fcnt, B, e->sp_allocs) ||
        dalloc(e, &e->d_sp_ret, B, e->sp_allocs) || dalloc(e, &e->d_sp_fsum, B, e->sp_allocs) ||
        dalloc(e, &e->d_sp_rows, (size_t)c->capacity_steps * B * e->sp_row, e->sp_allocs))
        return AZG_E_DEVICE;
    e->d_sp_ctab = nullptr;
    if (discrete && c->temperature != 1.0) {
        // stable_normalizer (helpers.py:10-27) raises x / max(x) to the temperature: (c / m)^t for every pair of a root edge count c
        // and the root's largest count m that can occur, python float pow = libm pow on the host (like check_pw's table)
        const size_t ns = (size_t)e->cfg.n_sims;
        if (ns > 2048) return fail(e, AZG_E_UNSUPPORTED, "temperature != 1 on the device supports n_sims <= 2048");
        std::vector<double> tab((ns + 1) * (ns + 2) / 2, 0.0);
        for (size_t m = 1; m <= ns; ++m)
            for (size_t k = 0; k <= m; ++k) tab[m * (m + 1) / 2 + k] = std::pow((double)k / (double)m, c->temperature);
        if (dalloc(e, &e->d_sp_ctab, tab.size(), e->sp_allocs)) return AZG_E_DEVICE;
        HIPCHK(e, hipMemcpy(e->d_sp_ctab, tab.data(), tab.size() * 8, hipMemcpyHostToDevice));
    }
    HIPCHK(e, hipMemset(e->d_sp_t, 0, B * 4));
    HIPCHK(e, hipMemset(e->d_sp_episode, 0, B * 4));
    HIPCHK(e, hipMemset(e->d_sp_fcnt, 0, B * 4));
    HIPCHK(e, hipMemset(e->d_sp_ret, 0, B * 8));
    HIPCHK(e, hipMemset(e->d_sp_fsum, 0, B * 8));
    std::vector<double> roots(B * e->S_env);
    azg_synthetic_roots(e, roots.data());
    HIPCHK(e, hipMemcpy(e->d_roots, roots.data(), roots.size() * 8, hipMemcpyHostToDevice));
    HIPCHK(e, hipMemset(e->d_carry, 0, B * 4));
    e->carry_max = discrete ? e->cfg.n_sims : 0;   // a reused root carries its node count as a child: at most n_sims
    e->sp_on = 1; e->sp_max_len = c->max_episode_length; e->sp_det = c->deterministic; e->sp_cap = c->capacity_steps; e->sp_steps = 0;
    e->sp_insert = 0; e->sp_total = 0; e->sp_fs = c->final_selection; e->sp_ring = c->ring_mode; e->sp_agent_eps = c->agent_epsilon;
    e->sp_step_idx = 0;
    return AZG_OK;
}

int azg_selfplay_begin(azg_engine* e, int32_t max_episode_length, int32_t deterministic, int32_t capacity_steps) {
    azg_selfplay_config c;
    memset(&c, 0, sizeof(c));
    c.struct_size = (int32_t)sizeof(c);
    c.max_episode_length = max_episode_length; c.deterministic = deterministic; c.capacity_steps = capacity_steps;
    c.final_selection = AZG_FS_MAX_VISIT; c.ring_mode = AZG_RING_STOP; c.temperature = 1.0; c.agent_epsilon = 0.0;
    return azg_selfplay_begin_ex(e, &c);
}

int azg_selfplay_step(azg_engine* e) {
    if (!e) return AZG_E_INVALID;
    if (!e->sp_on) return fail(e, AZG_E_STATE, "azg_selfplay_begin has not been called");
    if (e->sp_ring == AZG_RING_STOP && e->sp_steps >= e->sp_cap) return fail(e, AZG_E_STATE, "replay ring is full: download and clear the rows");
    int rc = azg_search_resident(e);
    if (rc) return rc;
    ON_DEVICE(e);
    if (e->team_pending) {   // (wide networks: the persistent team kernel may have given up -- see team_check)
        HIPCHK(e, hipStreamSynchronize(e->stream));
        rc = team_check(e);
        if (rc) return rc;
    }
    // ReplayBuffer.store (buffers.py:75-82) for this step's block of n_trees rows
    int slot;
    if (e->sp_steps < e->sp_cap) { slot = e->sp_steps; e->sp_steps += 1; }
    else { slot = e->sp_insert; e->sp_insert = (e->sp_insert + 1) % e->sp_steps; }
    SelfPlay sp;
    sp.max_len = e->sp_max_len; sp.deterministic = e->sp_det; sp.step_idx = e->sp_step_idx;
    sp.final_selection = e->sp_fs; sp.agent_eps = e->sp_agent_eps; sp.ctab = e->d_sp_ctab;
    sp.t = e->d_sp_t; sp.episode = e->d_sp_episode; sp.fcnt = e->d_sp_fcnt; sp.ret = e->d_sp_ret; sp.fsum = e->d_sp_fsum;
    sp.rows = e->d_sp_rows + (size_t)slot * e->cfg.n_trees * e->sp_row;
    sp.roots = e->d_roots; sp.carry = e->d_carry;
    const int B = e->cfg.n_trees;
    e->redo_ok = 0;   // (the step moves the roots on: the search that just ran cannot be re-run for a dump)
    if (e->Kmax <= 16) {
        rc = launch_results(e);   // return_results of this search (a launch only after the lock-step / team kernels)
        if (rc) return rc;
    }
    if (e->Kmax <= 16)
        hipLaunchKernelGGL(selfplay_kernel16, dim3((B + SP_TREES - 1) / SP_TREES), dim3(16 * SP_TREES), 0, e->stream, e->P, sp, e->Kmax, e->cfg.v_target, e->cfg.env_id, e->S_obs);
    else
        hipLaunchKernelGGL(selfplay_kernel, dim3((B + RK_THREADS - 1) / RK_THREADS), dim3(RK_THREADS), 0, e->stream, e->P, sp, e->Kmax, e->cfg.v_target, e->cfg.env_id, e->S_obs);
    HIPCHK(e, hipGetLastError());
    e->sp_total += 1;
    e->sp_step_idx += 1;
    return AZG_OK;
}

int azg_selfplay_rows(azg_engine* e, float* rows, size_t max_rows, int32_t clear) {
    if (!e) return AZG_E_INVALID;
    if (!e->sp_on) return fail(e, AZG_E_STATE, "azg_selfplay_begin has not been called");
    ON_DEVICE(e);
    HIPCHK(e, hipStreamSynchronize(e->stream));
    size_t n = (size_t)e->sp_steps * e->cfg.n_trees;
    if (n > max_rows) n = max_rows;
    if (rows && n) HIPCHK(e, hipMemcpy(rows, e->d_sp_rows, n * e->sp_row * 4, hipMemcpyDeviceToHost));
    if (clear) { e->sp_steps = 0; e->sp_insert = 0; }   // ReplayBuffer.clear (buffers.py:56-60)
    return (int)n;
}

int azg_selfplay_ring(azg_engine* e, int32_t* size_steps, int32_t* insert_step, int64_t* total_steps) {
    if (!e) return AZG_E_INVALID;
    if (!e->sp_on) return fail(e, AZG_E_STATE, "azg_selfplay_begin has not been called");
    if (size_steps) *size_steps = e->sp_steps;
    if (insert_step) *insert_step = e->sp_insert;
    if (total_steps) *total_steps = e->sp_total;
    return AZG_OK;
}

int azg_selfplay_rows_device(azg_engine* e, void** device_ptr, size_t* capacity_rows, size_t* row_len) {
    if (!e || !device_ptr) return AZG_E_INVALID;
    if (!e->sp_on) return fail(e, AZG_E_STATE, "azg_selfplay_begin has not been called");
    *device_ptr = e->d_sp_rows;
    if (capacity_rows) *capacity_rows = (size_t)e->sp_cap * e->cfg.n_trees;
    if (row_len) *row_len = (size_t)e->sp_row;
    return AZG_OK;
}

int azg_selfplay_stats(azg_engine* e, double* fsum, int32_t* fcnt, double* env_state) {
    if (!e) return AZG_E_INVALID;
    if (!e->sp_on) return fail(e, AZG_E_STATE, "azg_selfplay_begin has not been called");
    ON_DEVICE(e);
    HIPCHK(e, hipStreamSynchronize(e->stream));
    size_t B = e->cfg.n_trees;
    D2H(fsum, e->d_sp_fsum, B * 8);
    D2H(fcnt, e->d_sp_fcnt, B * 4);
    D2H(env_state, e->d_roots, B * e->S_env * 8);
    return AZG_OK;
}

int azg_math_selftest(int device_id, int fn_id, const double* in, double* out, size_t n) {
    if (!in || !out || n == 0) return AZG_E_INVALID;
    DeviceScope scope(device_id);
    if (!scope.ok) return AZG_E_DEVICE;
    double *di = nullptr, *dout = nullptr;
    if (hipMalloc((void**)&di, n * 8) != hipSuccess) return AZG_E_DEVICE;
    if (hipMalloc((void**)&dout, n * 8) != hipSuccess) { (void)hipFree(di); return AZG_E_DEVICE; }
    int rc = AZG_OK;
    if (hipMemcpy(di, in, n * 8, hipMemcpyHostToDevice) != hipSuccess) rc = AZG_E_DEVICE;
    if (rc == AZG_OK) {
        if (fn_id == 100) {
            int K = (int)((n - 1) / 2);
            hipLaunchKernelGGL(mfma_probe_kernel, dim3(1), dim3(64), 0, 0, di, dout, K);
        } else if (fn_id == 102 && n >= 16) {
            // out[0..11] = shader cycles per dependent operation (see latency_probe_kernel)
            int* chase = nullptr;
            std::vector<int> hc(1 << 16);
            for (size_t i = 0; i < hc.size(); ++i) hc[i] = (int)((i * 4099 + 77) & (hc.size() - 1));
            if (hipMalloc((void**)&chase, hc.size() * 4) == hipSuccess) {
                (void)hipMemcpy(chase, hc.data(), hc.size() * 4, hipMemcpyHostToDevice);
                hipLaunchKernelGGL(latency_probe_kernel, dim3(1), dim3(64), 0, 0, dout, chase, 4096);
                (void)hipDeviceSynchronize();
                (void)hipFree(chase);
            }
        } else if (fn_id == 101 && n >= 8) {
            // in = [workgroups, iterations, launches]; out = [cycles, 100 MHz ticks, -, ms per launch, TFLOP/s]
            const int wgs = (int)in[0], iters = (int)in[1], reps = (int)in[2] > 0 ? (int)in[2] : 1;
            hipEvent_t e0, e1;
            (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
            hipLaunchKernelGGL(mfma_rate_kernel, dim3(wgs), dim3(256), 0, 0, dout, iters);
            (void)hipEventRecord(e0, 0);
            for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(mfma_rate_kernel, dim3(wgs), dim3(256), 0, 0, dout, iters);
            (void)hipEventRecord(e1, 0);
            (void)hipEventSynchronize(e1);
            float ms = 0.0f;
            (void)hipEventElapsedTime(&ms, e0, e1);
            double res[2] = {ms / reps, (double)wgs * 4 * iters * 16 * 2048.0 / (ms / reps * 1e-3) / 1e12};
            (void)hipMemcpy(dout + 3, res, sizeof(res), hipMemcpyHostToDevice);
            (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
        } else {
            hipLaunchKernelGGL(math_selftest_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, fn_id, di, dout, n);
        }
        if (hipGetLastError() != hipSuccess || hipDeviceSynchronize() != hipSuccess) rc = AZG_E_DEVICE;
    }
    if (rc == AZG_OK && hipMemcpy(out, dout, n * 8, hipMemcpyDeviceToHost) != hipSuccess) rc = AZG_E_DEVICE;
    (void)hipFree(di);
    (void)hipFree(dout);
    return rc;
}

}  // extern "C"
