/*
 * ipp_engine.h -- C-ABI of the MI355X-native batched IPP environment-step engine.
 *
 * The reference (dmar-bonn/ipp-rl) has no FFI boundary: its hot path is plain Python/NumPy
 * behind the Mapping / Sensor / Simulation classes.  This header declares the entry points a
 * ctypes binding of that path would bind (INTEGRATION.md shows the stub); every function names
 * the reference interface it replaces (file:line relative to the reference root).
 *
 * Conventions
 *   - plain C, no exceptions; every function returns int: 0 = OK, < 0 = error
 *     (ipp_last_error() returns a thread-local message).
 *   - every pointer marked [dev] is a DEVICE pointer (e.g. torch.Tensor.data_ptr()); the caller
 *     owns all buffers including the state arena.  [host] marks host pointers.
 *   - `stream` is a hipStream_t passed as void* (torch.cuda.current_stream().cuda_stream);
 *     all work is enqueued on it and nothing synchronises unless stated.
 *   - grid is H x W = y_dim x x_dim, N = H*W cells, flat cell = x_dim*row + col (C order),
 *     position = [x (col axis), y (row axis), altitude] in metres (float64, like the reference).
 *   - one engine per device; calls on one engine are not re-entrant.
 */
#ifndef IPP_ENGINE_H
#define IPP_ENGINE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define IPP_ABI_VERSION 14

/* covariance state representation */
#define IPP_DENSE  0 /* P[N][N] fp32 per env, updated in place (mapping/grid_maps.py:10-11)         */
#define IPP_FACTOR 1 /* P = P0(sigma^2, l) - U U^T, U[N][r] fp32 grows by m columns per step         */

/* ipp_step flags */
#define IPP_COV_ONLY        1u /* no observation / mean update     (mappings.py:114 cov_only=True)        */
#define IPP_PREDICT_ONLY    2u /* reward only, nothing is written  (mappings.py:114 predict_only=True)    */
#define IPP_ADAPTIVE        4u /* masked reward                    (planning/common/rewards.py:8-12)      */
#define IPP_USE_FLIGHT_TIME 8u /* cost = flight time, else distance (planning/common/actions.py:8-12)     */
#define IPP_GIVEN_OBSERVATION 16u /* meas_noise holds the observation z itself (update_grid_map(pos, z),  */
                                  /* mapping/mappings.py:114-121): no crop / noise / clip is applied       */
#define IPP_UPDATE_PREV     32u /* after the cost is taken, prev_action[item] <- action[item]: the driver's    */
                                /* "previous_action = action" (planning/mcts_zero/episode_generators.py:146)  */
                                /* without a copy kernel; prev_action must then be writable device memory     */

/* per-item status written by ipp_step */
#define IPP_STATUS_OK            0
#define IPP_STATUS_CHOL_FALLBACK 1 /* S not PD: inverse formula used (mappings.py:200-215), dense only  */
#define IPP_STATUS_NOT_PD        2 /* S not PD in factor form: step not applied, reward = NaN           */
#define IPP_STATUS_RANK_FULL     3 /* factor rank_cap exceeded: reward valid, state not written         */
#define IPP_STATUS_BAD_FOOTPRINT 4 /* m or f above the engine's compiled caps / unsupported resize      */

#define IPP_MAX_MEAS 25 /* largest supported measurement count m per step */

typedef struct ipp_config {
    int32_t x_dim;            /* W [cells]   environment.x_dim    (mapping/grid_maps.py:13-24)   */
    int32_t y_dim;            /* H [cells]   environment.y_dim    (mapping/grid_maps.py:26-37)   */
    double  resolution;       /* [m/cell]    environment.resolution (grid_maps.py:39-50)         */
    double  tan_half_fov_x;   /* tan(0.5*radians(angle_x)) computed by the host in fp64 (sensors/cameras.py:44) */
    double  tan_half_fov_y;   /* tan(0.5*radians(angle_y))                               (sensors/cameras.py:45) */
    double  rf_altitude;      /* resolution factor is 2 strictly above this altitude: 10.0 (cameras.py:125)    */
    double  coeff_a;          /* sensor.model.coeff_a (sensors/models/sensor_models.py:27-30) */
    double  coeff_b;          /* sensor.model.coeff_b                                          */
    double  signal_variance;  /* mapping.signal_variance: nominal Matern sigma^2 (mapping/mappings.py:235-244) */
    double  length_scale;     /* mapping.length_scale:   nominal Matern l                                      */
    double  max_v;            /* experiment.uav.max_v (planning/common/actions.py:32-41) */
    double  max_a;            /* experiment.uav.max_a                                     */
    double  value_threshold;  /* experiment.scenario.value_threshold (rewards.py:11)     */
    double  interval_factor;  /* experiment.scenario.interval_factor (rewards.py:11)     */
    double  cluster_radius;   /* sensor.simulation.cluster_radius (simulations/simulations.py:43-47) */
    int32_t state_repr;       /* IPP_DENSE or IPP_FACTOR */
    int32_t capacity;         /* number of env state slots */
    int32_t rank_cap;         /* IPP_FACTOR: max columns of U per env (>= steps per episode * max m) */
    int32_t max_batch;        /* largest n of one ipp_step / ipp_reset call */
    int32_t max_measurements; /* compile-time cap on m: 9 (default config) or 25 */
    int32_t tile_threads;     /* 0 = auto; threads per streaming workgroup (multiple of 64, <= 640) */
    int32_t window_rows;      /* IPP_FACTOR: 0 = exact full columns; R > 0 = a new column of U is kept only on the
                                 cells within R grid rows AND R grid columns of its footprint, zero elsewhere
                                 (|Wc| < 3e-8 beyond 12 cells for the example prior); stored columns that
                                 cannot reach the current footprint are not streamed; bytes are counted as moved.
                                 The dropped entries scale like exp(-sqrt(3) R resolution / length_scale): choose R
                                 from the prior (12 = 13 length scales for the example config), 0 when in doubt,
                                 or ask ipp_min_window_rows.  ipp_engine_create refuses an R that drops prior
                                 covariances above 1e-6 for the largest length scale a reset may install */
    int32_t score_scratch;    /* 1 = reserve the scratch of ipp_score_actions in the arena: band of G = P M P
                                 (N x 190 doubles) and, for IPP_FACTOR, one dense P (N x Npad floats) */
    int32_t node_capacity;    /* tree nodes of ipp_tree_step (each (max_measurements + 1) x Npad floats); 0 = none */
    int32_t fixed_prior;      /* windowed IPP_FACTOR engines: 1 = no reset installs a length scale above
                                 cfg.length_scale (no shuffle_prior_cov), 0 = up to 1.2 x (mappings.py:238-240).
                                 Sets the length scale the window bound is checked for; a reset beyond it
                                 poisons the env (prior, mean, diag = NaN) instead of losing accuracy silently */
} ipp_config;

typedef struct ipp_info {
    int32_t abi_version;
    int32_t n_cells;      /* N */
    int32_t n_pad;        /* padded row length of every per-cell array [floats] */
    int32_t tile_threads; /* threads per streaming workgroup */
    int32_t n_tiles;      /* streaming workgroups per env */
    int32_t meas_cap;     /* compiled m cap in use (9 or 25) */
    int32_t fp_cap;       /* compiled footprint-cell cap in use (4 * meas_cap) */
    int32_t window_rows;
    uint64_t arena_bytes;     /* total bytes the engine carves from the caller's arena */
    uint64_t cov_slot_bytes;  /* bytes of covariance state per env slot */
    uint64_t step_lds_bytes;  /* dynamic LDS per workgroup of the streaming step kernel (occupancy: 160 KiB per CU) */
    int32_t fused_step;       /* 1: ipp_step is ONE fused kernel per launch (ipp_step_autoreset folds the resets into it) */
    int32_t patch_layout;     /* 1: factor columns stored as compact patches of their rectangles (k_step_patch.h) */
    int32_t patch_waves;      /* patch layout: waves per item of the step kernel (workgroup = 64 x this many threads), else 0 */
    int32_t patch_big_min_items; /* patch layout, launches that run two waves per item (IPP_PATCH_WAVES=2, or patch_two_wave_min_items): launches of at least this many items take
                                 the instantiation with four rows per request group (k_step_patch<2, 4, 6>); 0: one instantiation for every
                                 launch size (the default: k_step_patch<3, 8, 6>) */
    int32_t patch_split_min_items; /* patch layout: launches of at least this many items run the SPLIT step -- an item-parallel prologue
                                 kernel and a unit-parallel streaming kernel (csrc/k_step_split.h) -- instead of the fused kernel; results are
                                 bit-identical; 0: never */
    int32_t patch_two_wave_min_items; /* patch layout, three-wave engines (the default): launches of at least this many items run two waves
                                 per item (k_step_patch<2>: 12 items per CU instead of 8; from patch_big_min_items items its
                                 four-rows-per-group form) -- a launch of many rounds of workgroup slots has no tail of heavy items to
                                 shorten and is paid in items in flight; bit-identical results; 0: never */
} ipp_info;

/* Debug / test view of the last ipp_step's per-item prologue (host struct, filled by ipp_debug_step_item). */
typedef struct ipp_step_item {
    int32_t env, dst, rank_before, status;
    int32_t xl, xr, yu, yd;   /* sensors/cameras.py:49-75 */
    int32_t rf, m, f, pad;    /* cameras.py:122-125, mappings.py:125-126 */
    double  cost;             /* actions.py:8-41 */
    double  noise_var;        /* sensor_models.py:27-30 */
    double  S[IPP_MAX_MEAS * IPP_MAX_MEAS];     /* row-major m x m (mappings.py:182-183) */
    double  Linv[IPP_MAX_MEAS * IPP_MAX_MEAS];  /* upper-triangular inverse (mappings.py:185-186), or S^-1 on fallback */
    double  z[IPP_MAX_MEAS];  /* observation after noise + clip (simulations/sensor_manipulations.py:44-57) */
    double  y[IPP_MAX_MEAS];  /* L^-T (z - H x)  (mappings.py:189,195-196) */
} ipp_step_item;

int         ipp_abi_version(void);
const char* ipp_last_error(void);

/*
 * Smallest window_rows > 0 ipp_engine_create accepts for `cfg` (prior, resolution, fixed_prior): the smallest R
 * with  sigma^2 (1 + a) exp(-a) <= 1e-6,  a = sqrt(3) R resolution / l_max.  10 for the example config with
 * fixed_prior = 1, 12 with the 1.2 x headroom of shuffle_prior_cov.
 */
int ipp_min_window_rows(const ipp_config* cfg, int32_t* rows /*[host]*/);

/*
 * Size of the device arena the engine needs for `cfg` (state slabs for `capacity` envs + per-call
 * scratch for `max_batch` items).  Replaces: the per-object NumPy allocations of GridMap.mean /
 * GridMap.cov_matrix (mapping/grid_maps.py:10-11, mapping/mappings.py:259-261).
 */
int ipp_engine_arena_bytes(const ipp_config* cfg, uint64_t* bytes /*[host]*/);

/*
 * Create an engine on HIP device `device` over a caller-owned arena (256-byte aligned).
 * Replaces: GridMap(params) + Mapping(grid_map, sensor) construction (mapping/mappings.py:16-21),
 * Camera / AltitudeSensorModel parameter capture (sensors/cameras.py:13-32, sensor_models.py:14-25).
 */
int ipp_engine_create(const ipp_config* cfg, int device, void* arena /*[dev]*/, uint64_t arena_bytes,
                      void** engine /*[host] out*/);
int ipp_engine_destroy(void* engine);
int ipp_engine_info(void* engine, ipp_info* out /*[host]*/);

/*
 * Episode reset of `n` env slots: mean <- 0.5, covariance <- prior, rank <- 0, ground truth <- gt / GRF.
 * Replaces: Mapping.init_priors GP branch (mapping/mappings.py:235-261) incl. shuffle_prior_cov draws
 * passed in as prior_scale; GaussianRandomField.create_ground_truth_map -> gaussian_random_field
 * (simulations/simulations.py:37-47, simulations/ground_truths.py:14-33).
 *   env_ids     [dev] int32[n] or NULL (= 0..n-1)
 *   prior_scale [dev] double[n][2] = (sigma^2, l) per env, or NULL (= nominal)
 *   gt          [dev] float[n][H][W] ground truth to install, or NULL
 *   white_noise [dev] float[n][H][W] standard normals -> device GRF (used when gt == NULL), or NULL
 *   (both NULL: ground truth left unchanged)
 */
int ipp_reset(void* engine, const int32_t* env_ids, int32_t n, const double* prior_scale, const float* gt,
              const float* white_noise, void* stream);

/*
 * ipp_reset plus the UAV's return to the mission start: prev_action[env][0..2] <- init_action for every reset
 * env (Mission.init_action, planning/missions.py:69), in the same kernel.
 *   prev_action [dev] double[capacity][3] indexed by ENV id (the batched driver's previous-waypoint table), or NULL
 *   init_action [host] double[3]
 */
int ipp_reset_episode(void* engine, const int32_t* env_ids, int32_t n, const double* prior_scale, const float* gt,
                      const float* white_noise, double* prev_action, const double* init_action, void* stream);

/*
 * ipp_step (in place, factor engines) with the episode resets that fall on this step folded into the same launch:
 * item i steps its env and, when reset_src[i] >= 0, then resets it exactly like ipp_reset_episode with
 * gt = reset_gt[reset_src[i]], the engine's default prior and prev_action[env] <- init_action.  The batched driver's
 * auto-reset: what Mission.execute's episode loop (planning/missions.py:69-118) does between two episodes, without a
 * launch between two step kernels.  The windowed 256-thread engine resets inside the step kernel, the other factor
 * paths by a second launch behind it; results are the same as ipp_step followed by ipp_reset_episode.
 *   prev_action [dev] double[..][3]: read per ITEM by the step (and written with IPP_UPDATE_PREV), written per ENV by
 *               the reset: use it with env_ids == NULL (item == env) unless the two index spaces agree
 *   reset_src   [dev] int32[n]: index into reset_gt, or -1 (no reset); NULL: plain ipp_step
 *   reset_gt    [dev] float[..][H][W] ground-truth fields (e.g. from ipp_generate_grf); NULL: every resetting env takes the field staged
 *               in its alternate plane (ipp_generate_grf_groups with gt_out == NULL) -- a flip, no copy.  The engine keeps one
 *               "staged" flag per env (set by that generator, taken by the flip): a flip of an env whose alternate plane holds no
 *               field generated since its last flip would observe a stale plane, so it poisons the env instead (prior, mean,
 *               variance NaN: every later step of the episode returns IPP_STATUS_NOT_PD and a NaN reward)
 *   init_action [host] double[3]
 */
int ipp_step_autoreset(void* engine, const int32_t* env_ids, int32_t n, const double* action, double* prev_action,
                       const float* meas_noise, uint32_t flags, float* reward, int32_t* status, const int32_t* reset_src,
                       const float* reset_gt, const double* init_action, void* stream);

/*
 * ipp_step_autoreset of the whole batch as n_parts launches, one per PART of the dispatch order, each on its own
 * stream: part p runs the items at positions [part_begin[p], part_begin[p + 1]) of the order installed by
 * ipp_set_item_order (n entries; the per-item arrays keep their batch indexing, env_ids == NULL: item == env).
 * Envs are independent (no exchange between envs; an episode is a chain of update_grid_map calls on ONE
 * map, mapping/mappings.py:114-153), so only an env's own step t + 1 has to follow its step t: with a FIXED partition of the
 * envs into parts and one stream per part, the next launch of part A starts in the slots that the slowest items of part B
 * still leave empty -- the double-buffered ("async") form of a vectorised env, where the policy of one half runs while
 * the other half steps.  The caller orders the streams against everything else (inputs ready on streams[p], results
 * consumed behind streams[p]); nothing in here joins them.  Fused engines only (ipp_info.fused_step == 1).
 *   part_begin  [host] int32[n_parts + 1], increasing, part_begin[0] == 0, part_begin[n_parts] == n
 *   streams     [host] hipStream_t[n_parts]
 * Other arguments: ipp_step_autoreset.  Results are bit-identical to the single launch (same kernel, same items).
 * With IPP_PREDICT_ONLY (rewards of candidate actions, Mapping.simulate_prediction_step without the state write: no resets, no
 * IPP_UPDATE_PREV) consecutive calls do not depend on each other at all; the parts still keep two launches in flight.
 */
int ipp_step_parts(void* engine, int32_t n, const double* action, double* prev_action, const float* meas_noise,
                   uint32_t flags, float* reward, int32_t* status, const int32_t* reset_src, const float* reset_gt,
                   const double* init_action, int32_t n_parts, const int32_t* part_begin, void* const* streams);

/*
 * Priors of the episodes that the following ipp_step_autoreset launches start: prior [dev] double[..][2] = (sigma^2, l)
 * of the episode that takes ground truth reset_gt[k] -- Mapping.init_priors with shuffle_prior_cov
 * (mapping/mappings.py:235-240: 0.8 .. 1.2 x the config's values per episode), the same numbers ipp_reset_episode takes
 * as prior_scale.  NULL (default): the config's prior.  The pointer is kept, not copied: it has to stay valid until the
 * launches that use it have run.  A length scale beyond what the column window was sized for poisons the env with NaN,
 * as in ipp_reset_episode.
 */
int ipp_set_reset_prior(void* engine, const double* prior);

/*
 * Reward of n candidate actions from the CURRENT state of ONE env slot; nothing is written.  The call of
 * greedy_search and of the rollout policy: simulate_prediction_step for every reachable action from the same
 * state (planning/common/optimization.py:33-104, planning/mcts_mission.py:232-246).
 * Same rewards as ipp_step(IPP_COV_ONLY | IPP_PREDICT_ONLY) with the env id repeated, but the state is read once
 * instead of once per candidate: reward = tr(S^-1 H (P M P)[F,F] H^T) / (cost + 1), M = adaptive mask
 * (csrc/k_score.h).  Needs ipp_config.score_scratch = 1.
 *   actions     [dev]  double[n][3]
 *   prev_action [host] double[3]      the one previous waypoint all candidates start from
 *   flags              IPP_ADAPTIVE | IPP_USE_FLIGHT_TIME
 *   reward      [dev]  float[n]; status [dev] int32[n] or NULL (IPP_STATUS_BAD_FOOTPRINT: footprint above the
 *                      compiled caps or wider than 10 cells, reward 0)
 */
int ipp_score_actions(void* engine, int32_t env_id, const double* actions, int32_t n, const double* prev_action,
                      uint32_t flags, float* reward, int32_t* status, void* stream);

#define IPP_TREE_DEPTH 6 /* nodes on a path of ipp_tree_step */

/*
 * Tree-search step on path-local factor columns: item i starts from the state of env slot root_ids[i] followed
 * by the covariance-only steps recorded in the nodes path_ids[i][0..5] (root side first, -1 = unused), takes
 * the covariance-only predict step for action[i] and returns its reward -- simulate_prediction_step at a tree node
 * (planning/mcts_zero/mcts.py:166-265, planning/mcts_mission.py:167-246).  With new_ids[i] >= 0 the step is also
 * recorded as node new_ids[i] (a child of the last path node): the node keeps only the <= max_measurements columns
 * the step appended and the state's diagonal -- the reference stores a whole N x N covariance per node
 * (mcts.py:16-21).  The root env slot is never written; the map mean (adaptive mask) is the root's.
 * Needs IPP_FACTOR with window_rows > 0, default tile_threads, node_capacity > 0; root rank + path columns +
 * new columns <= rank_cap (else IPP_STATUS_RANK_FULL: reward valid, no node written).
 *   root_ids [dev] int32[n]; path_ids [dev] int32[n][IPP_TREE_DEPTH]; new_ids [dev] int32[n] or NULL
 *   flags    IPP_ADAPTIVE | IPP_USE_FLIGHT_TIME | IPP_PREDICT_ONLY (ignore new_ids)
 */
int ipp_tree_step(void* engine, const int32_t* root_ids, const int32_t* path_ids, const int32_t* new_ids, int32_t n,
                  const double* action, const double* prev_action, uint32_t flags, float* reward, int32_t* status,
                  void* stream);
/*
 * ipp_score_actions from the state of a tree node: root env slot + path (host int32[IPP_TREE_DEPTH], -1 padded) --
 * the rollout policy's "score every reachable action" at a search node (planning/mcts_mission.py:232-246).
 */
int ipp_tree_score_actions(void* engine, int32_t root_id, const int32_t* path_ids, const double* actions, int32_t n,
                           const double* prev_action, uint32_t flags, float* reward, int32_t* status, void* stream);
/* diag of a node's state, float[N] */
int ipp_tree_read_diag(void* engine, int32_t node_id, float* out, void* stream);

/*
 * Device-side tree search (csrc/k_mcts.h).  Replaces the per-simulation Python of planning/mcts_zero/mcts.py:
 * simulate (:166-265), compute_uct with forced playouts (:280-296), get_next_actions_mask (:148-158),
 * add_exploration_noise (:160-164) and the value backup (:255-265) for MANY roots at once; get_policy (:83-143) reads
 * the root rows back.  The caller owns every buffer ([dev] pointers below; the host side in
 * planning/mcts_zero/device_mcts.py allocates them as torch tensors) and keeps them alive for the search.
 * One wave owns one root: root j uses the node ids [j nodes_per_root, (j+1) nodes_per_root) (the first is the root node),
 * the device-node ids [j dev_per_root, (j+1) dev_per_root) of the engine's node pool and its own hash table.
 * (This is an ORDERING invariant too: between a root's descents the select kernel only waits for its own stores -- a
 * workgroup-scope fence, csrc/k_mcts.h -- which is correct exactly because no table row is shared between two waves.)
 * Before a search: n_flags = 0 except the root nodes (2 = stored), n_value = 0, n_devpath = -1, n_hash[root node] = any
 * distinct non-zero key, root_count = 1, dev_count = 0, h_keys = 0, err = 0; before every wave: pend_count = rq_count = 0.
 */
typedef struct ipp_mcts_tables {
    int32_t roots;            /* R */
    int32_t kmax;             /* width of the padded valid-action rows */
    int32_t nodes_per_root;
    int32_t dev_per_root;
    int32_t table_size;       /* hash slots per root, a power of two >= 2 nodes_per_root */
    int32_t max_depth;        /* path steps recorded per descent: >= episode_horizon + 1 - depth */
    int32_t wave;             /* most simulations in flight per root (buffer dimension W) */
    int32_t horizon;          /* episode_horizon (mcts.py:36) */
    int32_t grid_w, grid_h;   /* cells along the first / second waypoint coordinate */
    int32_t n_levels, n_off;  /* altitude levels; candidate cell offsets around a position */
    int32_t num_actions;      /* A = n_levels grid_w grid_h */
    int32_t use_flight_time;  /* cost = trapezoidal flight time (actions.py:32-41) instead of the distance */
    int32_t tie_break;        /* 0 = lowest action index among equal PUCT scores, 1 = counter-based uniform draw */
    int32_t device;
    double res, max_dist;     /* cell size; max_valid_action_distance */
    double gamma, puct_init, puct_base, fpf;  /* mcts.py:25-33 */
    double vmax, amax;
    /* geometry [dev] */
    const double* actions;       /* [A][3] waypoints in the reference's enumeration (actions.py:73-82) */
    const int32_t* cell_action;  /* [grid_w][grid_h] level-0 action index of a cell */
    const int32_t* off_x;        /* [n_off] cell offsets, ordered so that the action index ascends within a level */
    const int32_t* off_y;
    const uint64_t* zkey;        /* [A] per-action hash keys (node key = root key + sum over the measurements taken) */
    const double* uniform_ps;    /* [kmax + 1] prior of each of K equally likely valid actions (summed like NumPy sums) */
    /* edge tables [R nodes_per_root][kmax] */
    int32_t* t_idx; double* t_ps; double* t_nsa; double* t_qsa; double* t_num; int32_t* t_child;
    /* node tables [R nodes_per_root] */
    int32_t* n_k; double* n_ns; uint8_t* n_flags; uint64_t* n_hash; double* n_value;
    int32_t* n_devpath;          /* [..][6] device nodes from the root env slot to this node's state, -1 padded */
    int32_t* root_count; int32_t* dev_count;  /* [R] ids used */
    uint64_t* h_keys; int32_t* h_vals;        /* [R][table_size] */
    /* per wave: recorded descents [W][R][max_depth] / [W][R], pending leaves [R][W] */
    int32_t* p_node; int32_t* p_k; double* p_cost; int32_t* p_len; int32_t* leaf;
    int32_t* pend_node; int32_t* pend_depth; int32_t* pend_sim; double* pend_prev; double* pend_budget; int32_t* pend_count;
    /* covariance steps requested by a wave of simulations [max_depth R W] (one list; a descent asks for at most max_depth): these
     * are the argument arrays of ipp_tree_step */
    int32_t* rq_root; int32_t* rq_parent; int32_t* rq_k; int32_t* rq_child; int32_t* rq_newdev;
    double* rq_cost; double* rq_prev; double* rq_action; int32_t* rq_count;  /* rq_count [1] */
    int32_t* ts_paths; float* ts_reward; int32_t* ts_status;
    int32_t* err;                /* [4] node range / device-node range exhausted, tree-step status, kmax too small */
    /* optional (NULL / 0: evaluated in the kernel): the two factors of the PUCT prior that depend on a node's visit count only,
     * for Ns = 0 .. ns_table_n - 1 -- puct_c[Ns] = puct_init + log((Ns + puct_base + 1) / puct_base), sqrt_ns1[Ns] = sqrt(Ns + 1)
     * (mcts.py:280-296), tabulated by the host: an fp64 log and a square root per tree level of every descent otherwise */
    const double* puct_c; const double* sqrt_ns1; int64_t ns_table_n;
    /* A search split into GROUPS of roots, one table set per group (DeviceMCTS(groups = 2): the groups' waves of simulations alternate on
     * two streams, so that one group's selection -- one wave per SIMD, latency-bound -- runs beside the other's tree steps).  All 0 for a
     * search in one piece.  Results do not depend on the split: the counter-based draws (tie breaks, Dirichlet noise) are keyed on
     * root_base + j, root j of this set uses the device nodes [dev_base + j dev_per_root, ..) of the engine's node pool, and the tree
     * steps of this set use the engine's per-item scratch slots from scratch_base (sets stepped at the same time: disjoint ranges;
     * scratch_base + roots x wave <= ipp_config.max_batch; patch-layout engines only). */
    int32_t root_base; int32_t dev_base; int32_t scratch_base; int32_t reserved0;
} ipp_mcts_tables;

/* W descents per root (virtual visits between them), from waypoint prev0[j] with budget0[j] at tree depth `depth`;
 * sim0 = index of the first of these simulations in the search.  Fills p_*, leaf, pend_*, rq_*, the path arguments ts_paths of the
 * requested steps and the device paths (n_devpath) of the children that get a device node. */
int ipp_mcts_select(const ipp_mcts_tables* t, const int32_t* root_env /*[dev] R env slots*/, const double* prev0 /*[dev] R x 3*/,
                    const double* budget0 /*[dev] R*/, int32_t depth, int32_t sim0, int32_t wave, uint64_t seed, void* stream);
/* The covariance steps [first, first + n) of the wave's request list: ipp_tree_step on them and the edge numerators
 * t_num = reward (cost + 1).  The requests of a wave of simulations do not depend on each other (a node that gets its device
 * state in this wave is a leaf until ipp_mcts_expand), so one launch takes them all; n <= ipp_config.max_batch per call.
 * flags as for ipp_tree_step.  n < 0 (first = 0, engines with ipp_info.patch_layout = 1): the count stays on the device
 * (t->rq_count[0], written by ipp_mcts_select) and the launch is sized for min(max_batch, roots x wave) items -- the driver queues it
 * behind the selection without a read-back and launches requests beyond that size, if any, when it has read the count. */
int ipp_mcts_steps(void* engine, const ipp_mcts_tables* t, int32_t first, int32_t n, uint32_t flags, void* stream);
/* Pending leaves: valid-action sets (sets_only = 1: only those, so that the caller can ask its network with them) and
 * priors / value: prior [dev] [R W][kmax] on the valid sets or NULL = uniform; value [dev] [R W] or NULL = value_const;
 * Dirichlet(alpha) noise of weight eps on the root of simulation 0. */
int ipp_mcts_expand(const ipp_mcts_tables* t, const double* prior, const double* value, double value_const, int32_t sets_only,
                    double alpha, double eps, uint64_t seed, void* stream);
/* Values back along the `wave` recorded descents of every root; clears pend_count and rq_count for the next ipp_mcts_select
 * (the caller clears them once before the first). */
int ipp_mcts_backup(const ipp_mcts_tables* t, int32_t wave, void* stream);
/* The search policies of all roots from their visit counts (get_policy's read-out, planning/mcts_zero/mcts.py:83-143, for
 * temperature > 0): deploy_time = 0 takes the forced playouts back (:109-131; tie_uniform [dev] R numbers in [0, 1) pick the kept
 * action among the most visited ones like the reference's rng.choice, NULL = the first), then visits^(1/temperature) normalised.
 * policy [dev] R x kmax float64 on the roots' valid sets (0 on the padding), valid_idx [dev] R x kmax int32 = those sets (-1 padded;
 * may be NULL), ok [dev] R = 0 where the reference returns None (root not expanded / no visits kept). */
int ipp_mcts_policy(const ipp_mcts_tables* t, const double* tie_uniform, double temperature, int32_t deploy_time, double* policy,
                    int32_t* valid_idx, int32_t* ok, void* stream);

/*
 * NN input state plane of one env slot: the N x N covariance with the rows / columns outside the adaptive mask
 * zeroed, min-max normalised (planning/common/features.py:91-101,74-81), fp32, row-major [N][N].
 *   mean_for_mask [dev] float[N] = adaptive_info["mean"] (the reference masks every state of the history with the
 *                 CURRENT map mean), or NULL = the slot's own mean; the diagonal is always the slot's
 *   flags         IPP_ADAPTIVE (0 = no masking)
 * IPP_FACTOR engines need ipp_config.score_scratch = 1 (the slot is densified into that scratch).
 */
int ipp_state_plane(void* engine, int32_t env_id, const float* mean_for_mask, uint32_t flags, float* out, void* stream);

/*
 * Ground-truth generation only: white noise [n][H][W] -> min-max normalised Gaussian random field into the
 * caller buffer gt_out [n][H][W] (no env slot is touched).  Lets the host prepare the next episodes' ground
 * truths on a side stream while ipp_step runs, then install them with ipp_reset(gt = ...).
 * Replaces gaussian_random_field (simulations/ground_truths.py:14-33).  Uses its own scratch, so it may
 * overlap ipp_step / ipp_reset of the same engine issued on another stream.
 */
int ipp_generate_grf(void* engine, int32_t n, const float* white_noise /*[dev]*/, float* gt_out /*[dev]*/, void* stream);
/*
 * The same with the white noise drawn inside the generator: field i gets exactly the numbers
 * ipp_fill_normal_rows(out, 1, n, H * W, row_ids, row_offset, seed, subsequence) would have put into row i (so a ground truth does
 * not depend on which of the two routes made it), without the H * W floats per field written and read back.  Only where
 * the engine has such a generator (50x50 and 100x100 grids: k_grf_fft.h); returns -3 elsewhere (use the two calls).
 *   row_ids [dev] int32[n] or NULL (row i)
 */
int ipp_generate_grf_rows(void* engine, int32_t n, const int32_t* row_ids, int64_t row_offset, uint64_t seed, uint64_t subsequence,
                          float* gt_out /*[dev]*/, void* stream);
/* ... for fields of SEVERAL episodes in one launch (the batched driver stages the ground truths of a block of steps at once):
 * field i belongs to group i / group_rows and draws from subsequence + group_subsequence[group] ([host] int64, at most 16 groups,
 * passed by value: nothing is uploaded in front of the launch); a negative row_ids[i] skips field i (its gt_out row stays as it is).
 * group_rows = 0: one group (= ipp_generate_grf_rows).
 * gt_out == NULL: field i is written into the ALTERNATE ground-truth plane of env slot row_ids[i] -- every env owns two planes, the
 * current one and the one for its next episode; a reset folded into a step launch with reset_gt == NULL (ipp_step_autoreset,
 * ipp_step_parts) then only flips the env to it instead of copying H * W floats in and out (at 100x100 and 2048 resets per step the
 * copies were 164 MB of a step's traffic).  ipp_read_gt / ipp_write_gt / ipp_reset always address the CURRENT plane.
 * simulations/ground_truths.py:14-33 (the field), mapping/mappings.py:217-261 (the reset that installs it). */
int ipp_generate_grf_groups(void* engine, int32_t n, int32_t group_rows, const int64_t* group_subsequence /*[host] or NULL*/,
                            const int32_t* row_ids /*[dev] or NULL*/, int64_t row_offset, uint64_t seed, uint64_t subsequence,
                            float* gt_out /*[dev]*/, void* stream);

/*
 * One fused environment step for `n` items.  Replaces, per item:
 *   simulate_prediction_step            (planning/common/optimization.py:14-30)
 *     compute_adaptive_msk              (planning/common/rewards.py:8-12)
 *     Mapping.update_grid_map predict   (mapping/mappings.py:114-153, 156-215)
 *       project_field_of_view           (sensors/cameras.py:49-75), get_resolution_factor (:122-125)
 *       measurement_variance_matrix / measurement_model_matrix (sensors/models/sensor_models.py:32-81)
 *     compute_reward / action_costs     (planning/common/rewards.py:15-31, planning/common/actions.py:8-41)
 *   Sensor.take_measurement             (sensors/cameras.py:108-116 -> simulations/simulations.py:26-34,
 *                                        simulations/sensor_manipulations.py:7-57)
 *   Mapping.update_grid_map execute     (mapping/mappings.py:114-153: mean + covariance commit)
 *
 *   env_ids     [dev] int32[n] source slots (may repeat when IPP_PREDICT_ONLY), NULL = 0..n-1
 *   dst_ids     [dev] int32[n] slots that receive the updated state (tree-search expansion: the
 *               source slot is left untouched), or NULL = update in place
 *   action      [dev] double[n][3]   measurement position
 *   prev_action [dev] double[n][3]   previous waypoint (cost term)
 *   meas_noise  [dev] float[n][max_measurements] standard normals, C order of the downsampled
 *               observation, scaled in-kernel by the noise "variance" used as std
 *               (sensor_manipulations.py:56-57); NULL = noise-free.  Ignored with IPP_COV_ONLY.
 *   reward      [dev] float[n] out
 *   status      [dev] int32[n] out (IPP_STATUS_*), may be NULL
 */
int ipp_step(void* engine, const int32_t* env_ids, const int32_t* dst_ids, int32_t n, const double* action,
             const double* prev_action, const float* meas_noise, uint32_t flags, float* reward, int32_t* status,
             void* stream);

/*
 * Observation only: z = clip(area_downsample(gt[F]) + noise_var * eps, 0, 1) for `n` items, no state access
 * beyond the ground truth.  Replaces Sensor.take_measurement -> ScalarFieldSimulation.take_measurement
 * (sensors/cameras.py:108-116, simulations/simulations.py:26-34, simulations/sensor_manipulations.py:7-57).
 *   z_out [dev] float[n][max_measurements] (C order of the downsampled observation), m_out [dev] int32[n],
 *   shape_out [dev] int32[n][2] = (rows, cols) of the observation as the reference returns it (may be NULL)
 */
int ipp_observe(void* engine, const int32_t* env_ids, int32_t n, const double* action, const float* meas_noise,
                float* z_out, int32_t* m_out, int32_t* shape_out, void* stream);

/* UAV limits of the flight-time cost (uav_specifications argument, planning/common/actions.py:8-41). */
int ipp_set_uav(void* engine, double max_v, double max_a);

/*
 * Dispatch order of the items of the following ipp_step / ipp_step_autoreset launches of exactly n items: workgroup b
 * takes item order[b] ([dev] int32[n], a permutation of 0..n-1 that stays alive until replaced; NULL = the default
 * XCD-balanced order).  Results do not depend on it.  The batched driver sorts by descending steps since the env's reset
 * (more stored columns = a longer item), so that a launch ends on its short items: no reference counterpart
 * (scheduling only).
 */
int ipp_set_item_order(void* engine, const int32_t* order, int32_t n);

/* Change the adaptive-mask parameters used by subsequent steps.  The reference passes them per call in
 * adaptive_info = {"mean", "value_threshold", "interval_factor"} (planning/common/optimization.py:22-25). */
int ipp_set_adaptive(void* engine, double value_threshold, double interval_factor);

/* Copy state slots src[i] -> dst[i] (tree-search children keep their parent's belief: the reference
 * keeps one dense P per node, planning/mcts_mission.py:25-31, planning/mcts_zero/mcts.py:16-21). */
int ipp_fork(void* engine, const int32_t* src_ids /*[dev]*/, const int32_t* dst_ids /*[dev]*/, int32_t n,
             void* stream);

/* Materialise per-env state for callers / tests (reference: grid_map.mean, np.diag(cov_matrix),
 * cov_matrix, sensor_simulation.ground_truth_map; planning/missions.py:176-203). out is [dev]. */
int ipp_read_mean(void* engine, int32_t env_id, float* out /*[N]*/, void* stream);
int ipp_read_diag(void* engine, int32_t env_id, float* out /*[N]*/, void* stream);
int ipp_read_gt(void* engine, int32_t env_id, float* out /*[N]*/, void* stream);
int ipp_read_cov_dense(void* engine, int32_t env_id, float* out /*[N][N]*/, void* stream);
int ipp_read_rank(void* engine, int32_t env_id, int32_t* rank /*[host] out*/, void* stream); /* synchronises */
int ipp_read_ranks(void* engine, int32_t* out /*[dev] int32[capacity]*/, void* stream);      /* all slots, async */

/* Inject state (reference callers assign grid_map.mean / grid_map.cov_matrix directly:
 * planning/mcts_mission.py:395,413; adaptive_info["mean"], planning/common/optimization.py:22-25).
 * mean / P are [dev] float arrays; P (N x N, row-major) is accepted only by IPP_DENSE engines. */
int ipp_write_mean(void* engine, int32_t env_id, const float* mean /*[N]*/, void* stream);
int ipp_write_gt(void* engine, int32_t env_id, const float* gt /*[N]*/, void* stream);
int ipp_write_cov_dense(void* engine, int32_t env_id, const float* P /*[N][N]*/, void* stream);

/*
 * Per-env evaluation metrics after a step (planning/evaluation_metrics.py:4-58 as called from
 * planning/missions.py:176-203): out[i][0..7] = rmse, masked rmse, wrmse, mll, wmll, trace,
 * masked trace, uncertainty difference (mask = ground truth >= value_threshold).
 */
int ipp_metrics(void* engine, const int32_t* env_ids /*[dev]*/, int32_t n, float* out /*[dev] [n][8]*/, void* stream);

/* Philox4x32-10 + Box-Muller standard normals into a caller buffer (throughput runs; parity runs feed
 * NumPy legacy-stream normals instead).  out[i] depends only on (seed, subsequence, i). */
int ipp_fill_normal(void* engine, float* out /*[dev]*/, uint64_t count, uint64_t seed, uint64_t subsequence,
                    void* stream);

/* Row-keyed variant for sharded runs: out[p][j][c] (planes x rows x row_len) depends only on (seed,
 * subsequence + p, row id, c) with row id = (row_ids ? row_ids[j] : j) + row_offset.  With a row = one env (its
 * ground-truth white noise of one episode: simulations/ground_truths.py:19; its measurement noise of one step:
 * simulations/sensor_manipulations.py:56) and row_offset = the shard's first GLOBAL env id, every env draws the same
 * numbers however the batch is split over GPUs ("seeds derived from the global env id").
 *   row_ids [dev] int32[rows] or NULL */
int ipp_fill_normal_rows(void* engine, float* out /*[dev]*/, int32_t planes, int32_t rows, int32_t row_len,
                         const int32_t* row_ids, int64_t row_offset, uint64_t seed, uint64_t subsequence, void* stream);

/* Scheduling probe for ipp_step_parts: `launches` dependent 30-us kernels on each of the two streams, issued alternately; ms = the time
 * until both chains have run (launches x 0.03 ms when the two hardware queues behind the streams dispatch independently, ~a third more
 * when a dependent launch waits ~10 us for its queue's turn -- which depends on the queues the runtime happened to give the streams).
 * The batched driver times a few candidate pairs and keeps the best.  Synchronises both streams. */
int ipp_probe_stream_pair(void* engine, void* stream_a, void* stream_b, int32_t launches, double* ms /*[host]*/);

/* Arena placement (round 6).  The engine lives in ONE caller-owned arena (ipp_engine_create).  How that arena is MAPPED decides 15-20 % of a
 * large batch's step rate (configs[3] share 0.60 -> 0.51 ms per step, profiles/r06_arena_modes.txt): the driver covers a mapping with large
 * address-translation fragments only where the virtual address and the physical blocks are mutually aligned, which hipMalloc and a
 * framework's caching allocator (2-MiB-aligned addresses, whatever blocks the allocator's history left) achieve by chance.  ipp_arena_alloc:
 * IPP_ARENA_HIPMALLOC = plain hipMalloc; IPP_ARENA_VMM = physical chunks of `chunk_bytes` (0 = 1 GiB; rounded up to the device's
 * granularity; one tail chunk of the remainder) from hipMemCreate, mapped at an address aligned to `align_bytes` (a power of two, 0 =
 * chunk_bytes; the reservation's own alignment argument is not honoured by this runtime: one `align_bytes` more is reserved and the chunks
 * go to the aligned address inside).  chunk = align = 1 GiB is the mapping IPPEngine(arena = "auto") uses from 256 MiB.
 * ipp_arena_probe: the step kernel's row stream without its arithmetic (512-byte slices of 656-float patches spread over `items` equal
 * slots of the arena, `rows` stored columns per unit), ms = average duration of one of `launches` launches on `stream` (synchronises).
 * (Neither this probe nor ipp_arena_latency tells a fast mapping from a slow one -- the fragment size only shows on the step kernel's own
 * access pattern -- they are the instruments that ruled the memory system's bandwidth and idle latency out.)
 * No reference counterpart: the reference's state is NumPy arrays in host memory (mapping/grid_maps.py:10-11). */
#define IPP_ARENA_HIPMALLOC 0
#define IPP_ARENA_VMM       1
int ipp_arena_alloc(int device, uint64_t bytes, int32_t kind, uint64_t chunk_bytes, uint64_t align_bytes,
                    void** arena /*[host] out: [dev] pointer*/);
int ipp_arena_free(void* arena /*[dev], from ipp_arena_alloc*/);
/* ipp_arena_free of an IPP_ARENA_VMM arena returns its physical memory but keeps its ADDRESS range reserved (a range that is
 * mapped again after a free has been seen to fault on this runtime); this is how many bytes of address space are retired so far. */
int ipp_arena_retired_bytes(uint64_t* bytes /*[host]*/);
/* ... and keeps its whole physical chunks in a per-process pool (at most IPP_ARENA_POOL_GIB GiB, default 48; 0: none) for the next VMM arena
 * of the same chunk size on that device: a chunk that has served a fast arena serves the next one too, while memory the driver has just taken
 * back comes out in pieces for a while (the second large workload of a process ran 5-25 % slower than in a fresh process).  ipp_arena_trim
 * hands the pooled chunks of `device` (< 0: every device) back to the driver; released [host] (may be NULL) receives the bytes. */
int ipp_arena_trim(int device, uint64_t* released /*[host]*/);
int ipp_arena_probe(int device, const void* arena /*[dev]*/, uint64_t bytes, int32_t items, int32_t rows, int32_t launches,
                    void* stream, double* ms /*[host]*/);
/* ... and its latency: `waves` single-wave workgroups each follow a chain of `hops` DEPENDENT 512-byte requests to pseudo-random
 * patches of the arena; ns_per_hop = duration of the launch / hops (synchronises). */
int ipp_arena_latency(int device, const void* arena /*[dev]*/, uint64_t bytes, int32_t waves, int32_t hops, void* stream,
                      double* ns_per_hop /*[host]*/);

/* Copy the prologue record of item `idx` of the most recent ipp_step to the host (synchronises).  The fp64 copies of S, L^-1, z
 * and y are written by the step only after ipp_debug_capture(engine, 1) (they cost 1.4 KB of stores per item); footprint, m, cost
 * and status are always there. */
int ipp_debug_capture(void* engine, int32_t enable);
int ipp_debug_step_item(void* engine, int32_t idx, ipp_step_item* out /*[host]*/, void* stream);

/* Average duration [ms] of the streaming kernels of ipp_step launched since the last call with
 * reset != 0 (hipEvent pairs recorded on the caller's stream; bench.py's roofline leg).
 * kind: 0 = gain kernel, 1 = dense downdate kernel, 2 = prologue kernel.  Synchronises. */
int ipp_profile_enable(void* engine, int32_t enable);
int ipp_profile_read(void* engine, int32_t kind, double* avg_ms /*[host]*/, int64_t* launches /*[host]*/, int32_t reset);
/* Same events, other question: the time [ms] during which AT LEAST ONE kernel of that kind was running (union of the
 * dispatches' [start, stop] intervals) since the last reset of this figure -- what the launches of ipp_step_parts cost when
 * they overlap on the device (their individual durations then add up to more than the wall time).  Synchronises. */
int ipp_profile_read_busy(void* engine, int32_t kind, double* busy_ms /*[host]*/, int64_t* launches /*[host]*/, int32_t reset);
/* Bytes the gain kernel actually streamed / wrote since the last reset of the counter (rows x valid cells x 4 +
 * 4 floats of mean / diag traffic per touched cell), counted on the device per workgroup: the numerator of roofline.achieved when
 * window_rows > 0.  Synchronises. */
int ipp_streamed_bytes(void* engine, uint64_t* bytes /*[host]*/, int32_t reset, void* stream);
/* Same, plus the bytes the fused kernel reads a second time (mean / diag of the touched tiles: once for the adaptive
 * mask, rewards.py:11, once more by the update): traffic beyond SURVEY 8(d)'s per-cell count r + m + 4, which
 * `bytes` follows.  mask_reread_bytes may be NULL.  Synchronises. */
int ipp_streamed_bytes_detail(void* engine, uint64_t* bytes /*[host]*/, uint64_t* mask_reread_bytes /*[host]*/, int32_t reset,
                              void* stream);
/* Patch-layout engines, belonging to the last ipp_streamed_bytes(_detail) read: the same count with every stored row taken only on
 * the cells INSIDE that column's own rectangle (the lanes outside are masked requests that fetch nothing) -- the bytes that have
 * to move, a lower bound of `bytes` (which charges a stored row on every cell of the unit that meets it).  0 for other engines. */
int ipp_streamed_bytes_needed(void* engine, uint64_t* bytes /*[host]*/);

#ifdef __cplusplus
}
#endif
#endif /* IPP_ENGINE_H */
