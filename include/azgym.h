/*
 * azgym.h -- C ABI of the MI355X batched MCTS engine (libazgym_hip.so).
 *
 * The reference (timoklein/alphazero-gym, pure Python) has no FFI; its boundary for the
 * hot path is the Python class surface MCTSDiscrete / MCTSContinuous (alphazero/search/mcts.py)
 * called from DiscreteAgent.act / ContinuousAgent.act (alphazero/agent/agents.py:257-303, 492-537).
 * Each entry point below replaces one piece of that surface for B independent trees at once; the
 * ctypes facade in alphazero_gym_amd/search/mcts.py binds exactly these symbols.
 *
 * Conventions: plain pointers and sizes only; the caller owns every in/out buffer (host memory),
 * the engine copies in/out and never retains caller pointers; the engine owns all device memory.
 * Return 0 on success, a negative AZG_E_* code otherwise, message via azg_last_error().
 * An engine is single-owner and not re-entrant: one engine per GPU per process.
 *
 * The CPU oracle (oracle/azg_oracle.c, test infrastructure only) exports the same functions with
 * the prefix azo_ instead of azg_ and the same structs, so tests drive both with identical inputs.
 */
#ifndef AZGYM_H
#define AZGYM_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AZG_ABI_VERSION 1

enum { AZG_OK = 0, AZG_E_INVALID = -1, AZG_E_TERMINAL_ROOT = -2, AZG_E_DEVICE = -3, AZG_E_STATE = -4, AZG_E_UNSUPPORTED = -5 };

/* closed-form environments (gym classic control; call sites alphazero/search/mcts.py:443-449, 680-687) */
enum { AZG_ENV_CARTPOLE = 0, AZG_ENV_PENDULUM_V0 = 1, AZG_ENV_PENDULUM_V1 = 2, AZG_ENV_MOUNTAINCAR = 3,
       AZG_ENV_MOUNTAINCAR_CONT = 4, /* gym MountainCarContinuous-v0: one continuous action, TERMINATES at the flag -- the
                                        continuous search's terminal-node surface (mcts.py:599-600, 619-623, 682) */
       AZG_ENV_ACROBOT = 5 /* gym Acrobot-v1: three discrete torques, SIX observations (the network's first layer takes two MFMA
                              k-steps), reward -1 per step and 0 on the step that ends the episode */ };
/* MCTSDiscrete (mcts.py:310-526) / MCTSContinuous (mcts.py:529-741) */
enum { AZG_MODE_DISCRETE = 0, AZG_MODE_CONTINUOUS = 1 };
/* V_target_policy (mcts.py:299-304) */
enum { AZG_VT_OFF_POLICY = 0, AZG_VT_ON_POLICY = 1, AZG_VT_GREEDY = 2 };
/* trunk nonlinearity (alphazero/network/utils.py:5-14; the configs use relu and elu; swish == silu) */
enum { AZG_ACT_RELU = 0, AZG_ACT_ELU = 1, AZG_ACT_LEAKYRELU = 2, AZG_ACT_RELU6 = 3, AZG_ACT_SILU = 4, AZG_ACT_HARDSWISH = 5 };

#define AZG_MAX_HIDDEN_LAYERS 8
/* Ties in selectionUCT's arg-max.  The reference picks one of the tied children with random.choice (helpers.py:46-52), which
 * no fixed-seed comparison can follow; AZG_TIE_FIRST (default, the parity setting: the goldens assert that no tie occurred)
 * takes the lowest index, AZG_TIE_RANDOM picks uniformly among the tied children with a Philox draw keyed by (tree, search,
 * node record, node visit count) -- the reference's distribution, reproducible, same on the CPU oracle and the GPU. */
enum { AZG_TIE_FIRST = 0, AZG_TIE_RANDOM = 1 };

/* Constructor kwargs of MCTSDiscrete.__init__ (mcts.py:316-362) / MCTSContinuous.__init__ (mcts.py:537-587),
 * plus the batch geometry. */
typedef struct azg_config {
    int32_t struct_size;   /* sizeof(azg_config), checked */
    int32_t device_id;     /* HIP device ordinal */
    int32_t env_id;        /* AZG_ENV_* */
    int32_t mode;          /* AZG_MODE_* */
    int32_t n_trees;       /* B: independent trees searched by one azg_search call */
    int32_t n_sims;        /* n_rollouts */
    int32_t num_actions;   /* discrete only: the env's own action count (CartPole 2, MountainCar 3, Acrobot 3) */
    int32_t v_target;      /* AZG_VT_* */
    int32_t tree_id_base;  /* global id of local tree 0 (multi-GPU sharding; keys the RNG streams) */
    int32_t tie_break;     /* AZG_TIE_* : what helpers.argmax (helpers.py:30-52) does with exactly equal scores */
    double c_uct;
    double gamma;
    double epsilon;
    double c_pw;           /* continuous only */
    double kappa;          /* continuous only */
    double reward_scale;   /* continuous only: PENDULUM_R_SCALE (mcts.py:20, 687); reward /= reward_scale */
    double action_bound;   /* continuous only: squashed-Normal bound (policies.py:493-499) */
    uint64_t seed;
} azg_config;

/* Policy/value MLP: trunk of n_hidden Linear+activation layers, value head Linear(H,1), distribution head
 * Linear(H, n_dist) with n_dist = num_actions (DiscretePolicy, policies.py:238-259), 2*action_dim
 * (DiagonalNormalPolicy, policies.py:434) or num_components*(2*action_dim+1) (DiagonalGMMPolicy, policies.py:541-542).  Weight blob = torch state_dict order and layout:
 * for each trunk layer W[out][in] then b[out]; value_head W[1][H], b[1]; dist_head W[n_dist][H], b[n_dist]. */
typedef struct azg_mlp_desc {
    int32_t struct_size;
    int32_t in_dim;
    int32_t n_hidden;
    int32_t hidden[AZG_MAX_HIDDEN_LAYERS];
    int32_t n_dist;
    int32_t activation;    /* AZG_ACT_* */
    float log_std_min;     /* clamp of log_std (policies.py:456-460); continuous only */
    float log_std_max;
    int32_t num_components; /* continuous only: 0/1 = squashed Normal (n_dist = 2); C >= 2 = Gaussian mixture,
                             * DiagonalGMMPolicy (policies.py:502-669), n_dist = 3C laid out [mu_0..mu_C-1, log_std_0.., log_coeff_0..] */
    int32_t layernorm;      /* 1: nn.LayerNorm (eps 1e-5, affine) after every trunk activation (policies.py:105-118, 242-255);
                             * the blob then holds, per trunk layer, W, b, ln_weight[out], ln_bias[out] */
} azg_mlp_desc;

typedef struct azg_engine azg_engine;

int azg_abi_version(void);

/* MCTS*.__init__ */
int azg_engine_create(const azg_config* cfg, azg_engine** out);
void azg_engine_destroy(azg_engine* e);
const char* azg_last_error(const azg_engine* e); /* e may be NULL: last create error */

/* model=self.nn (agents.py:82): copies and re-lays-out the weights; call again after every optimiser step */
int azg_set_weights(azg_engine* e, const azg_mlp_desc* desc, const float* blob, size_t n_floats);
/* The same with the blob in DEVICE memory on the engine's GPU (the parameters PyTorch-ROCm just updated, flattened in state_dict
 * order, or the buffer an RCCL broadcast delivered: run_continuous.py:144-155's train step scaled out).  The re-layout runs as a
 * gather kernel on the engine's stream: no device -> host -> device hop.  The blob's contents must be complete when this is
 * called (producer stream synchronised); it may be reused as soon as the call returns.  Same blob, same network as azg_set_weights. */
int azg_set_weights_device(azg_engine* e, const azg_mlp_desc* desc, const float* device_blob, size_t n_floats);

/* index mixed into the RNG counter; auto-incremented by azg_search */
int azg_set_search_index(azg_engine* e, uint32_t idx);

/* MCTS*.search(Env) (mcts.py:418-462, 656-702) for B trees.
 *   root_env_state [B][S_env] float64: CartPole (x, x_dot, theta, theta_dot); Pendulum (theta, theta_dot); both MountainCars
 *                  (position, velocity); Acrobot (theta1, theta2, theta1_dot, theta2_dot)
 *   root_n_carry   [B] or NULL: visit count carried by a reused root (MCTSDiscrete.forward, mcts.py:495-526)
 * Terminal roots -> AZG_E_TERMINAL_ROOT (ValueError at mcts.py:382-383, 599-600). */
int azg_search(azg_engine* e, const double* root_env_state, const int32_t* root_n_carry);

/* MCTS.return_results (mcts.py:269-307): root statistics, rows padded to azg_max_children().
 *   actions [B][Kmax] float32 (discrete: the action index as float), counts [B][Kmax] int32,
 *   Q [B][Kmax] float64, v_target [B] float64, n_children [B] int32.  Any pointer may be NULL. */
int azg_results(azg_engine* e, float* actions, int32_t* counts, double* Q, double* v_target, int32_t* n_children);
/* The same for consumers on the engine's GPU: return_results of the last search is computed into the engine's own device buffers
 * (asynchronous: a launch on the engine's stream, no copy, no wait) and their addresses are handed out -- same shapes and dtypes
 * as azg_results; valid until the engine is destroyed, contents until the next search's results.  Any pointer may be NULL.
 * Order consumers after the engine's stream (azg_sync) before reading. */
int azg_results_resident(azg_engine* e, const float** actions, const int32_t** counts, const double** Q, const double** v_target,
                         const int32_t** n_children);

/* What MCTSDiscrete.forward (mcts.py:495-526) inspects: per root child the child node's visit count
 * (-1: edge has no child node) and its environment state.  child_n [B][Kmax], child_state [B][Kmax][S_env]. */
int azg_root_children(azg_engine* e, int32_t* child_n, double* child_state);

/* root network outputs cached by the search: value [B] float32; dist [B][n_dist] float32
 * (continuous: mu.., sigma..; discrete: softmax priors). For MLP parity tests. */
int azg_root_eval(azg_engine* e, float* value, float* dist);

/* Batched network inference without a search: Policy.predict_V / DiscretePolicy.predict_pi / DiagonalNormalPolicy.forward /
 * DiagonalGMMPolicy.forward (policies.py:150-160, 340-352, 436-464, 544-560) for n observations at once, with exactly the
 * arithmetic a search uses for its leaves.  obs [n][obs_dim] float32 (host); value [n]; dist [n][n_dist] (discrete: softmax
 * priors; Normal: mu, sigma; mixture: mu_c.., sigma_c.., cumulative mixture probabilities); raw [n][1 + n_dist] = the
 * value head and the untransformed distribution head (logits / mu, log_std).  Output pointers may be NULL. */
int azg_mlp_eval(azg_engine* e, const float* obs, size_t n, float* value, float* dist, float* raw);

/* whole-tree dump for bit-exact parity tests: per tree the node/edge records in creation order.
 * See DESIGN.md "record layout". rec_* arrays are [B][azg_max_records()]; n_records [B]. */
int azg_dump_tree(azg_engine* e, int32_t* n_records, int32_t* parent, int32_t* edge_n, double* edge_W, double* edge_Q,
                  float* edge_action, int32_t* node_n, double* node_r, float* node_V, uint8_t* node_flags);

int azg_max_children(const azg_engine* e);
int azg_max_records(const azg_engine* e);
int azg_env_state_dim(const azg_engine* e);
int azg_obs_dim(const azg_engine* e);

/* synthetic fixed-seed root states for benchmarks and self-play resets:
 * Pendulum theta~U(-pi,pi), theta_dot~U(-1,1); CartPole ~U(-0.05,0.05)^4, keyed by global tree id. */
int azg_synthetic_roots(azg_engine* e, double* root_env_state /*[B][S_env]*/);

/* timing of the last azg_search measured with HIP events on the engine stream (kernel only), milliseconds */
int azg_last_search_ms(azg_engine* e, float* ms);

/* What the last search ran as.  The engine picks a kernel form and a tree residency per search from the engine's parameters, and
 * two of those choices are performance cliffs a caller should be able to see (nothing about the RESULTS changes):
 *   - trees that do not fit LDS residency (more than 511 records = n_sims + 2, more than 16 children per node -- a large c_pw --,
 *     or a network / batch whose LDS plan exceeds the CU's 160 KB) are kept in global memory: every tree-walk access becomes an
 *     L2 / HBM round trip instead of an LDS one (lds_exit says which limit);
 *   - a wide-network search whose persistent team kernel could not keep all its workgroups resident (a GPU shared with other work)
 *     is redone by the per-layer launches (team_fallbacks counts them).
 * The first search of an engine that leaves LDS residency prints one line to stderr (AZG_QUIET=1 silences it).
 * Replaces the reference's nothing: its Python tree has one form (alphazero/search/mcts.py:418-462, 656-702). */
enum { AZG_FORM_NONE = -1, AZG_FORM_PERSISTENT = 0 /* one launch = a whole search (search_kernel) */,
       AZG_FORM_PER_LAYER = 1 /* wide networks: one launch per layer and tree phase of every simulation step */,
       AZG_FORM_TEAM = 2 /* wide networks: one persistent launch, the batch cut into teams of workgroups (ls_team_kernel) */ };
enum { AZG_TREES_GLOBAL = 0, AZG_TREES_LDS8 = 1 /* <= 255 records, 8-bit ids */, AZG_TREES_LDS9 = 2 /* <= 511 records */ };
enum { AZG_LDS_RESIDENT = 0, AZG_LDS_EXIT_RECORDS = 1, AZG_LDS_EXIT_CHILDREN = 2, AZG_LDS_EXIT_SIZE = 3, AZG_LDS_EXIT_FORCED = 4 /* AZG_FORCE_GLOBAL_TREE=1 (tests) */,
       AZG_LDS_NOT_APPLICABLE = 5 /* wide-network forms: the trees live in HBM by design (the team kernel stages its workgroup's two or four trees in LDS) */ };
typedef struct azg_search_report {
    int32_t struct_size;      /* sizeof(azg_search_report), set by the caller, checked */
    int32_t kernel_form;      /* AZG_FORM_* */
    int32_t tree_storage;     /* AZG_TREES_*: where the trees' hot records lived during the search */
    int32_t lds_exit;         /* AZG_LDS_*: why they were not LDS-resident */
    int32_t spec;             /* 1: the kernel compiled for epsilon 0 / lowest-index ties / the env's own action count (same results as 0) */
    int32_t waves, groups, tile_trees;            /* persistent form: waves per workgroup, 16-tree groups per workgroup, trees per group */
    int32_t team_trees, team_per_cu, team_parts;  /* team form: trees per team, workgroups per CU, launches the batch was cut into */
    int32_t team_fallbacks;   /* searches of this engine that the team kernel gave up on and the per-layer launches redid (cumulative) */
    int32_t max_records, max_children;            /* per tree: n_sims + 2 records; children per node the search can create */
    float last_ms;            /* azg_last_search_ms */
    char kernel_name[192];    /* the kernel of the last search as rocprofv3 names it */
} azg_search_report;
int azg_search_info(azg_engine* e, azg_search_report* info);

/* diagnostic builds only (-DAZG_STAMPS, tools/phase_profile.py): per-wave cycle sums of the in-kernel phase stamps, [rows][16] */
int azg_debug_stamps(azg_engine* e, unsigned long long* out, size_t max_rows);

/* device-resident variant used by bench.py: roots already uploaded by a previous azg_search/azg_upload_roots;
 * runs the search kernel only (no host<->device copies) */
int azg_upload_roots(azg_engine* e, const double* root_env_state, const int32_t* root_n_carry);
int azg_search_resident(azg_engine* e);
int azg_sync(azg_engine* e);

/* Device-resident self-play (the run loop of run_continuous.py:111-142 / run_discrete.py:94-122 for B games in lock step):
 * games live on the device; one azg_selfplay_step = one search from the current roots, then per game: the agent's final
 * action rule (continuous: most visited root action, first index on ties, agents.py:533; discrete: sampled in proportion
 * to counts/max(counts), temperature 1, agents.py:300-301, or arg-max when deterministic), one replay row
 * [obs | actions[Kmax] | counts[Kmax] | Q[Kmax] | V_target] (float32) appended to a device ring buffer, the real env
 * step, episode bookkeeping (reset on termination or after max_episode_length steps) and the next search's root
 * (discrete: with the reused root's carried visit count, MCTSDiscrete.forward mcts.py:495-526). */
int azg_selfplay_begin(azg_engine* e, int32_t max_episode_length, int32_t deterministic, int32_t capacity_steps);

/* The agents' whole final-action surface and the replay buffer's overwrite rule:
 *   final_selection  "max_visit(s)" / "max_value" (agents.py:294-301, 524-535): the rule works on the root's counts or its Qs
 *   discrete   pi = stable_normalizer(x, temperature) = |y / sum(y)|, y = (x / max(x))^temperature (helpers.py:9-27);
 *              action = pi.argmax() (deterministic) or sampled from pi by inverse CDF as numpy's choice() does
 *              (cdf = cumsum(pi) / cdf[-1], first index with u < cdf) with u from the engine's Philox stream.
 *              Counts: any temperature (c^t from a host-built libm table, (c / max)^t = c^t / max^t); Qs: temperature 1 only.
 *   continuous actions[x.argmax()] (first index on ties), or with probability agent_epsilon a uniformly random root action
 *              (ContinuousAgent.epsilon_greedy, agents.py:471-490)
 *   ring_mode  AZG_RING_STOP: azg_selfplay_step fails once capacity_steps steps are stored (download and clear);
 *              AZG_RING_FIFO: ReplayBuffer.store (buffers.py:75-82) in units of one step (n_trees rows): append until full,
 *              then overwrite slot insert_index and advance it cyclically. */
enum { AZG_FS_MAX_VISIT = 0, AZG_FS_MAX_VALUE = 1 };
enum { AZG_RING_STOP = 0, AZG_RING_FIFO = 1 };
typedef struct azg_selfplay_config {
    int32_t struct_size;        /* sizeof(azg_selfplay_config), checked */
    int32_t max_episode_length; /* cfg.max_episode_length (run_*.py) */
    int32_t deterministic;      /* DiscreteAgent.act(deterministic=...) */
    int32_t capacity_steps;     /* replay ring capacity in steps: ReplayBuffer.max_size = capacity_steps * n_trees rows */
    int32_t final_selection;    /* AZG_FS_* */
    int32_t ring_mode;          /* AZG_RING_* */
    double temperature;         /* DiscreteAgent.temperature */
    double agent_epsilon;       /* ContinuousAgent.epsilon */
} azg_selfplay_config;
int azg_selfplay_begin_ex(azg_engine* e, const azg_selfplay_config* cfg);
/* ReplayBuffer.size and .insert_index in steps, and the number of steps played since begin */
int azg_selfplay_ring(azg_engine* e, int32_t* size_steps, int32_t* insert_step, int64_t* total_steps);
/* the ring itself, for consumers on the same device (training batches gathered and all-gathered HBM -> HBM, no PCIe hop):
 * device pointer to [capacity_steps][n_trees][row_len] float32, valid until the next azg_selfplay_begin* / destroy; rows of the
 * first size_steps slots are defined once the engine's stream is synchronised (azg_sync). */
int azg_selfplay_rows_device(azg_engine* e, void** device_ptr, size_t* capacity_rows, size_t* row_len);
int azg_selfplay_step(azg_engine* e);
int azg_selfplay_row_len(const azg_engine* e);
/* rows of the steps played since the last clear, ordered [step][tree]; returns the number of rows copied */
int azg_selfplay_rows(azg_engine* e, float* rows, size_t max_rows, int32_t clear);
/* per game: sum of the returns of finished episodes, number of finished episodes, current env state [B][S_env] */
int azg_selfplay_stats(azg_engine* e, double* finished_return_sum, int32_t* finished_episodes, double* env_state);

/* bit-exactness self test of azg_math.h on the device: evaluates fn_id on n inputs */
int azg_math_selftest(int device_id, int fn_id, const double* in, double* out, size_t n);

#ifdef __cplusplus
}
#endif
#endif /* AZGYM_H */
