/*
 * marl_amd.h -- C ABI of libmarl_amd.so: batched Overcooked step / reset / featurize, the GAE
 * pass, and three small passes of the rollout / update loop (reward-event bookkeeping, action
 * sampling, the PPO minibatch loss) for MI355X (gfx950).
 *
 * The reference (loganwoudstra/MARL) is pure Python and has no FFI of its own; the boundary
 * it does have is the duck-typed vector env returned by main.py:66-96 `make_vector_env`
 * and driven at agent_environment.py:34,50,88, plus `compute_gae`
 * (agents/MAPPO.py:82-114, agents/CentralizedMAPPO.py:48-82).  Each entry point below names
 * the reference call it stands behind.  All pointers named d_* are DEVICE pointers owned by
 * the caller (16-byte aligned where noted); `stream` is a hipStream_t passed as void*
 * (NULL = default stream).  Nothing here synchronises the device or allocates in the
 * per-step calls; marl_env_create / marl_env_destroy allocate and free the small per-layout
 * lookup tables.  Every function returns 0 on success and a negative code on failure;
 * marl_last_error() describes the last failure on the calling thread.
 *
 * Threads: different env handles may be created, stepped and destroyed from different host
 * threads at the same time (the library's own shared bookkeeping is locked).  One handle belongs to
 * one host thread and one stream at a time: its state buffer is updated in place and
 * marl_env_set_auto_reset changes the handle without locking.  A handle belongs to the HIP
 * device that was current when it was created; every call on it must be made with that device
 * current (MARL_ERR_INVALID otherwise).
 */
#ifndef MARL_AMD_H
#define MARL_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct marl_env marl_env_t;

/* feature sets: names registered at overcooked_config.py:14-28 (+ README.md:65) */
enum {
    MARL_FEAT_GLOBAL_OBS = 0,             /* "global_obs"   overcooked_features.py:11-130 */
    MARL_FEAT_LOCAL_OBS = 1,              /* "local_obs"    overcooked_features.py:132-229 */
    MARL_FEAT_MINIMAL_SPATIAL = 2,        /* "Minimal_spatial" :284-312 */
    MARL_FEAT_MINIMAL_SPATIAL_OTHER = 3,  /* "Minimal_spatial_other_agent_aware" :231-281 */
    MARL_FEAT_ONLY_DIRECTION = 4          /* "only_direction" README.md:65,112 */
};

enum {
    MARL_OK = 0,
    MARL_ERR_INVALID = -1,   /* bad argument (shape, alignment, unsupported agent count ...) */
    MARL_ERR_LAYOUT = -2,    /* malformed ASCII layout */
    MARL_ERR_HIP = -3        /* a HIP runtime call failed */
};

const char *marl_last_error(void);

/*
 * Build an env family on the current HIP device.
 * Stands behind main.py:29-64 `make_env` + main.py:66-96 `make_vector_env`:
 *   layout_rows/height  the ASCII grid (overcooked_config.py:34-62 legend; '+' = spawn floor)
 *   num_agents          1 to 4 (config["num_agents"], main.py:48)
 *   feature             MARL_FEAT_* (config["features"], main.py:50)
 *   max_steps           episode length before truncation (overcooked_config.py:108)
 *   num_envs            E, envs held by THIS process (rows = E * num_agents, env-major)
 *   seed, env_index_base  counter-RNG key: env e draws from (seed, env_index_base + e), so
 *                       a shard of a larger job reproduces the same envs as the whole job
 *   dist_metric         0 = L1, 1 = squared L2 ordering for the "closest" features
 */
int marl_env_create(marl_env_t **out, const char *const *layout_rows, int height, int num_agents,
                    int feature, int max_steps, int num_envs, uint64_t seed, int64_t env_index_base,
                    int dist_metric);
void marl_env_destroy(marl_env_t *env);

/*
 * The reference steps cogrid through a vector wrapper that restarts a finished env inside
 * step() (default here, enabled = 1).  A bare cogrid env (tests/test_load.py:104-111) does not:
 * with enabled = 0 a truncated env keeps its final state and observation, t keeps counting (a
 * 32-bit counter: `truncated` stays set however long the caller goes on stepping), and the
 * caller resets.
 */
int marl_env_set_auto_reset(marl_env_t *env, int enabled);

/* shape queries */
int marl_env_obs_dim(const marl_env_t *env);       /* D: single_observation_space.shape[0], main.py:92 */
int marl_env_num_rows(const marl_env_t *env);      /* E * A: vec_env.num_envs, main.py:186 */
int marl_env_num_pots(const marl_env_t *env);
int marl_env_grid_cells(const marl_env_t *env);    /* H * W */
int marl_env_state_words(const marl_env_t *env);   /* uint32 words per env of packed state */
size_t marl_env_state_bytes(const marl_env_t *env);/* bytes the caller must allocate for d_state */

/*
 * vec_env.reset() (agent_environment.py:34,88): every env restarts; d_obs receives
 * float32 [E*A][D] (16-byte aligned).  d_spawn_override / d_dir_override: uint8 [E][A]
 * (cell index row*W+col / direction 0-3) or NULL to draw from the counter RNG.  The caller
 * names floor cells, a different one per agent of an env; the kernel does not read them back
 * for checking, it only keeps them inside its tables (a cell that is not floor becomes the
 * first floor cell, a direction is taken modulo 4).
 * d_state must have been zero-filled before the first reset (the RNG draw counter lives
 * in it and survives resets).
 */
int marl_env_reset(marl_env_t *env, void *d_state, float *d_obs, const uint8_t *d_spawn_override,
                   const uint8_t *d_dir_override, void *stream);

/*
 * vec_env.step(actions) (agent_environment.py:50): d_actions holds E*A action indices in
 * [0,7) with element size action_elem_size in {1, 4, 8} bytes (8 = the int64 that
 * `actions.cpu().numpy()` yields).  Outputs: d_obs float32 [E*A][D] (16-byte aligned),
 * d_rewards float32 [E*A], d_terminated / d_truncated uint8 [E*A].  An env whose agents are
 * all done restarts inside the call and reports the new episode's first observation
 * together with the done flags.  d_prio_override: uint8 [E][A] rank per agent (lower moves
 * first) replacing the RNG shuffle of movers, or NULL.
 */
int marl_env_step(marl_env_t *env, void *d_state, const void *d_actions, int action_elem_size,
                  float *d_obs, float *d_rewards, uint8_t *d_terminated, uint8_t *d_truncated,
                  const uint8_t *d_prio_override, void *stream);

/*
 * num_steps consecutive env.step calls (agent_environment.py:50) in one launch, for callers that
 * know the actions of all of them up front (random-action rollouts, replaying a recorded action
 * sequence): step k reads d_actions[k][N] and writes d_obs[k][N][D], d_rewards[k][N],
 * d_terminated[k][N], d_truncated[k][N] (d_prio_override[k][N] if given), N = num_envs * num_agents;
 * results are those of num_steps marl_env_step calls, bit for bit.  The env stays in registers
 * between steps and the simulation of a step overlaps the previous step's row stores.
 * N * D must be a multiple of 4 when num_steps > 1 (each step's rows start 16-byte aligned).
 */
int marl_env_step_many(marl_env_t *env, void *d_state, const void *d_actions, int action_elem_size, int num_steps,
                       float *d_obs, float *d_rewards, uint8_t *d_terminated, uint8_t *d_truncated,
                       const uint8_t *d_prio_override, void *stream);

/* feature_space.generate over the stored state (overcooked_features.py:107-121) */
int marl_env_featurize(marl_env_t *env, void *d_state, float *d_obs, void *stream);

/*
 * Canonical snapshot <-> packed state, for parity harnesses and state injection:
 *   d_agents int32 [E][A][4] = (row, col, dir, inv)      inv: 0 none 1 onion 2 plate 3 soup
 *   d_pots   int32 [E][NP][2] = (onion count, cooking timer)
 *   d_items  uint8 [E][H*W]  item lying on each floor / counter cell
 *   d_misc   int32 [E][2]    = (t, RNG draw counter)
 * Import does not judge the snapshot, but whatever the arrays hold it stays inside the library's
 * tables: codes are masked to their field widths and an agent placed off the grid (each coordinate
 * is tested against its own bound) or off the floor goes to the first floor cell.  A snapshot that
 * is to mean something is validated by the caller beforehand (the Python host does:
 * marl_amd/env.py, _validate_snapshot).
 */
int marl_env_export_state(marl_env_t *env, const void *d_state, int32_t *d_agents, int32_t *d_pots,
                          uint8_t *d_items, int32_t *d_misc, void *stream);
int marl_env_import_state(marl_env_t *env, void *d_state, const int32_t *d_agents, const int32_t *d_pots,
                          const uint8_t *d_items, const int32_t *d_misc, void *stream);

/*
 * compute_gae + value targets.
 * mode 0: agents/MAPPO.py:82-114 and :144.  d_next_values float32 [N].  d_workspace unused.
 * mode 1: agents/CentralizedMAPPO.py:48-82: the reward of step t is the mean over all N
 *         rows; d_next_values float32 [1].  d_workspace: marl_gae_workspace_bytes(T, N) bytes,
 *         8-byte aligned; its first T floats are the per-step reward means.  With
 *         reward_mean_ready == 0 the call computes them from d_rewards; a multi-GPU caller
 *         all-reduces its own sums, writes the means there and passes reward_mean_ready == 1.
 * d_rewards, d_dones, d_values, d_adv, d_ret: float32 [T][N] (row t contiguous).
 * d_ret may be NULL.  Arithmetic is float32 with every product and sum rounded
 * separately, in the reference's evaluation order.
 */
int marl_gae(const float *d_rewards, const float *d_dones, const float *d_values,
             const float *d_next_values, int T, int N, double gamma, double lam, float *d_adv,
             float *d_ret, int mode, void *d_workspace, int reward_mean_ready, void *stream);
size_t marl_gae_workspace_bytes(int T, int N);

/* per-step sums over rows, double [T] (fixed summation order); lets a sharded caller form the
 * global reward mean.  d_workspace as for marl_gae. */
int marl_row_sums(const float *d_x, int T, int N, double *d_sums, void *d_workspace, void *stream);

/*
 * Reward-event bookkeeping of the rollout driver, agent_environment.py:66-76: a step counts as a
 * delivery when any reward is >= 1, as a plating when any is == 0.3, as an onion-in-pot when any
 * is == 0.1 (float32 comparisons), and the episode return adds the mean reward.  The reference
 * looks at rows 0..1 only; this call keeps that (d_first_rows) and adds the same for every env.
 * Each call ADDS one step's rewards (float32 [num_envs * num_agents], env-major) to running
 * totals the caller zeroes at the start of an episode:
 *   d_totals      uint64[4]: sum over all rows of round(10 * reward) (exact: rewards are sums of
 *                 0.1, 0.3 and whole numbers), number of envs with a delivery / a plating / an
 *                 onion-in-pot this step.  Integer, so independent of summation order.
 *   d_first_rows  double[4] or NULL: float32 mean of rewards[0:2] (added as a double, like the
 *                 reference's Python float) and 0/1 flags for the three events among rows 0..1.
 *   d_workspace   marl_reward_events_workspace_bytes() bytes, 8-byte aligned.
 */
int marl_reward_events(const float *d_rewards, int num_envs, int num_agents, uint64_t *d_totals,
                       double *d_first_rows, void *d_workspace, void *stream);
size_t marl_reward_events_workspace_bytes(int num_envs, int num_agents);

/*
 * The same bookkeeping for num_steps steps at once: d_rewards float32 [num_steps][num_envs * num_agents]
 * (e.g. the rows a rollout has just stored, agent_environment.py:63-76 applied to each in turn).
 * Results equal num_steps marl_reward_events calls in step order, bit for bit (the rows-0..1 return is
 * accumulated step after step).  d_workspace: marl_reward_events_many_workspace_bytes() bytes, 8-byte aligned.
 * At most MARL_REWARD_EVENTS_MANY_MAX_STEPS steps per call (a rollout's worth; the finishing pass is one
 * workgroup that walks the steps in order); longer runs are passed in pieces.
 */
#define MARL_REWARD_EVENTS_MANY_MAX_STEPS 4096
int marl_reward_events_many(const float *d_rewards, int num_steps, int num_envs, int num_agents,
                            uint64_t *d_totals, double *d_first_rows, void *d_workspace, void *stream);
size_t marl_reward_events_many_workspace_bytes(int num_steps, int num_envs, int num_agents);

/*
 * marl_reward_events for callers that cannot allocate device memory or carry a workspace along:
 * code under hipGraph capture, tight serving loops, harnesses that forbid hipMalloc.  No
 * counterpart in the reference.  Same arguments and bit-identical results, minus d_workspace:
 * the per-chunk partial sums go to an 8 KiB array that is part of the library's device image
 * (every GPU the library is used on holds its own copy until the process ends).  The call
 * allocates nothing and frees nothing.  At most MARL_REWARD_EVENTS_NOALLOC_MAX_ROWS rows
 * (num_envs * num_agents) per call; more is refused with MARL_ERR_INVALID.  All calls on one GPU
 * share that array: calls that may overlap on one GPU (different streams, different host
 * threads) race on it and must be serialised by the caller; calls on one stream are ordered and
 * safe.  The first call on each device looks the array's address up (no stream work, no
 * allocation); make that call outside graph capture.
 */
#define MARL_REWARD_EVENTS_NOALLOC_MAX_ROWS (1 << 24)
int marl_reward_events_noalloc(const float *d_rewards, int num_envs, int num_agents, uint64_t *d_totals,
                               double *d_first_rows, void *stream);

/*
 * The sampling half of MAPPO.act (agents/MAPPO.py:58-77 -> model.py:52-56): for every row of
 * d_logits (float32, row-major, row_stride floats from one row to the next; 0 = num_actions)
 * draw an action from Categorical(logits=row[0:num_actions]) and report its log-probability
 * and the distribution's entropy.
 *   d_actions  int64 [num_rows];  d_logprob, d_entropy  float32 [num_rows] or NULL.
 *   d_next_column  float32 [num_rows] or NULL: receives row[num_actions] of every row (needs
 *   row_stride > num_actions).  For callers that compute the critic's value in the same GEMM
 *   as the logits (model.py:49-61 evaluates actor and critic on the same hidden layer): the
 *   value comes out contiguous without a strided copy of its own.
 *   seed, d_counter, row_index_base: the uniform of a row comes from a counter RNG keyed by
 *   (seed, *d_counter, row_index_base + row).  d_counter is a uint64 in DEVICE memory (NULL = 0)
 *   that the caller advances between calls (it is read, not written, here: a launch captured
 *   in a hipGraph draws fresh numbers on every replay as long as the counter moves).
 * log_prob and entropy are the float32 formulas of torch.distributions.Categorical
 * (logits - logsumexp; -sum p * log p); they agree with PyTorch's to the last bits, not bit
 * for bit (another exp / log implementation).
 */
int marl_sample_categorical(const float *d_logits, int num_rows, int num_actions, int row_stride, uint64_t seed,
                            const uint64_t *d_counter, int64_t row_index_base, int64_t *d_actions,
                            float *d_logprob, float *d_entropy, float *d_next_column, void *stream);

/*
 * Loss and head gradients of one PPO minibatch, agents/MAPPO.py:167-200 (decentralised critic;
 * value loss agents/MAPPO.py:243-248, distribution model.py:52-56):
 *   loss = mean(max(-A r, -A clamp(r, 1-clip, 1+clip))) - entropy_coef mean(H) + value_coef mean(0.5 (v - target)^2)
 * with r = exp(log_prob(action) - old_log_prob), H the entropy of Categorical(logits=row).
 * Sample n takes its logits from row n of d_logits (float32, row_stride floats per row, 0 =
 * num_actions) and its value from d_values[n]; action (stored as float32 like buffer.py:12),
 * old log-probability, advantage and value target come from entry d_index[n] of the rollout
 * store's flat arrays (d_index NULL: entry n).
 *   d_grad_logits float32 [num_samples][num_actions], d_grad_values float32 [num_samples]:
 *                 d loss / d logits and d loss / d values (the loss is a mean, so no second pass)
 *   d_stats       double[7]: loss, then the means the learner logs (MAPPO.py:211-218): value loss,
 *                 policy loss, entropy, approx_kl = mean((r - 1) - log r), old_approx_kl =
 *                 mean(-log r), clip fraction = mean(|r - 1| > clip)
 *   d_workspace   marl_ppo_loss_workspace_bytes(num_samples) bytes, 8-byte aligned.
 * Agrees with the autograd evaluation of the same formulas to float32 rounding.
 */
int marl_ppo_loss(const float *d_logits, int num_samples, int num_actions, int row_stride, const float *d_values,
                  const int64_t *d_index, const float *d_actions, const float *d_old_logprob,
                  const float *d_advantages, const float *d_value_target, double clip, double entropy_coef,
                  double value_coef, float *d_grad_logits, float *d_grad_values, double *d_stats, void *d_workspace,
                  void *stream);
size_t marl_ppo_loss_workspace_bytes(int num_samples);

#ifdef __cplusplus
}
#endif
#endif
