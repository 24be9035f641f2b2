/*
 * TEST INFRASTRUCTURE ONLY -- CPU oracle for the batched Overcooked hot path.
 *
 * Plain-C restatement of the algorithm that the reference drives through
 * `env.step` / `env.reset` (agent_environment.py:34,50,88), the feature
 * composition of overcooked_features.py:38-78,107-130 and the GAE pass of
 * agents/MAPPO.py:82-114 / agents/CentralizedMAPPO.py:48-82.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * link or load this file's library; the product (marl_amd/) never does.
 *
 * PARITY STATUS
 *  - GAE: pinned against golden vectors produced by importing the reference's
 *    agents/MAPPO.py and agents/CentralizedMAPPO.py (tests/golden/).
 *  - Featurizer: pinned against the one golden vector the reference ships,
 *    states/agent_0_dropping_onion_state.npy (202 floats).
 *  - Env dynamics (move / collide / interact / pot timers / rewards) and the
 *    leaf features' internals live in the third-party package `cogrid`
 *    (unpinned in requirements.txt:4; README.md:63 links commit f1beb729),
 *    which is absent from the reference tree and from this image:
 *    PARITY UNPINNED for those.  They are restated from cogrid's published
 *    behaviour and constrained by the fixture above and by the reference's
 *    call sites (agent_environment.py:66-72 reward magnitudes,
 *    tests/test_load.py:37-43 action indices, overcooked_config.py:68-110).
 */
#ifndef OVERCOOKED_ORACLE_H
#define OVERCOOKED_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define OC_MAX_AGENTS 8
#define OC_MAX_CELLS 256
#define OC_MAX_POTS 8

/* static cell types */
enum { OC_WALL = 0, OC_FLOOR = 1, OC_COUNTER = 2, OC_POT = 3,
       OC_ONION_STACK = 4, OC_PLATE_STACK = 5, OC_DELIVERY = 6 };
/* loose / held / placed items */
enum { OC_NONE = 0, OC_ONION = 1, OC_PLATE = 2, OC_SOUP = 3 };
/* feature sets (overcooked_config.py:14-28; only_direction: README.md:65) */
enum { OC_FEAT_GLOBAL = 0, OC_FEAT_LOCAL = 1, OC_FEAT_MINIMAL = 2,
       OC_FEAT_MINIMAL_OTHER = 3, OC_FEAT_ONLY_DIRECTION = 4 };
/* cardinal action set: (MoveUp, MoveDown, MoveLeft, MoveRight, PickupDrop, Toggle, Noop) */
enum { OC_ACT_UP = 0, OC_ACT_DOWN = 1, OC_ACT_LEFT = 2, OC_ACT_RIGHT = 3,
       OC_ACT_PICKUP_DROP = 4, OC_ACT_TOGGLE = 5, OC_ACT_NOOP = 6 };
/* directions (README.md:112: [RIGHT, DOWN, LEFT, UP]) */
enum { OC_DIR_RIGHT = 0, OC_DIR_DOWN = 1, OC_DIR_LEFT = 2, OC_DIR_UP = 3 };

#define OC_POT_CAPACITY 3
#define OC_COOK_TIME 30

typedef struct oc_layout {
    int H, W, A;
    int max_steps;
    int feature;      /* OC_FEAT_* */
    int dist_metric;  /* 0 = L1 (|dy|+|dx|), 1 = squared L2 */
    int auto_reset;   /* 1 (default): the vector wrapper restarts a finished env inside step */
    unsigned char type[OC_MAX_CELLS];
    int nspawn;
    int spawn[OC_MAX_CELLS]; /* '+' cells in scan order */
    int npots;
    int pot_cell[OC_MAX_POTS]; /* scan order */
    int obs_dim;
    int block_dim;
} oc_layout;

/* ---- layout ---- */
/* rows: H strings of equal width.  Returns 0 on success, <0 on error. */
int oc_layout_init(oc_layout *L, const char *const *rows, int H, int num_agents,
                   int feature, int max_steps, int dist_metric);
int oc_obs_dim(const oc_layout *L);

/*
 * Canonical state snapshot (one env), also the wire format of the product's
 * export/import entry points:
 *   agents int32[A][4]  = (row, col, dir, inv)
 *   pots   int32[NP][2] = (onion count, cooking timer)
 *   items  uint8[H*W]   = OC_NONE/ONION/PLATE/SOUP lying on a floor or counter cell
 *   misc   int32[2]     = (t, rng draw counter)
 */

/* ---- batched entry points (E envs, arrays laid out [E][...]) ---- */

/* Reset all envs.  spawn_override / dir_override (uint8[E][A], cell index /
 * direction) may be NULL: then spawn cells and directions come from the
 * counter RNG keyed by (seed, env_index_base + e). */
void oc_reset_batch(const oc_layout *L, int E, uint64_t seed, int64_t env_index_base,
                    int32_t *agents, int32_t *pots, uint8_t *items, int32_t *misc,
                    const uint8_t *spawn_override, const uint8_t *dir_override,
                    float *obs /* [E*A][D] or NULL */, int nthreads);

/* One vectorised step with per-env auto-reset (truncation at t >= max_steps
 * resets the env and returns the reset observation with the done flags).
 * actions int64[E*A]; prio_override uint8[E][A] (rank per agent) or NULL. */
void oc_step_batch(const oc_layout *L, int E, uint64_t seed, int64_t env_index_base,
                   int32_t *agents, int32_t *pots, uint8_t *items, int32_t *misc,
                   const int64_t *actions, const uint8_t *prio_override,
                   float *obs, float *rewards, uint8_t *terminated, uint8_t *truncated,
                   int nthreads);

void oc_featurize_batch(const oc_layout *L, int E, const int32_t *agents, const int32_t *pots,
                        const uint8_t *items, const int32_t *misc, float *obs, int nthreads);

/* ---- GAE (agents/MAPPO.py:82-114,144; agents/CentralizedMAPPO.py:48-82) ----
 * rewards, dones, values: float32[T][N]; next_values: float32[N] (mode 0) or
 * float32[1] (mode 1, centralised).  adv, ret: float32[T][N].  mode 0 = MAPPO,
 * mode 1 = CMAPPO (reward is the mean over all N rows per t). */
void oc_gae(const float *rewards, const float *dones, const float *values,
            const float *next_values, int T, int N, double gamma, double lam,
            float *adv, float *ret, int mode);

/* counter RNG shared by the spec (not cogrid's): exposed for tests */
uint32_t oc_rng_u32(uint64_t seed, int64_t env_index, uint32_t counter);

#ifdef __cplusplus
}
#endif
#endif
