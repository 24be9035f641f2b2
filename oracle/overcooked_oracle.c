/*
 * TEST INFRASTRUCTURE ONLY -- see overcooked_oracle.h for scope and parity status.
 *
 * The code below deliberately follows the reference stack's object-grid style
 * (scan the whole grid per feature, per-agent Python-like loops) rather than
 * the packed tables of the HIP kernels, so that agreement between the two is
 * evidence and not a tautology.
 */
#include "overcooked_oracle.h"

#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------ */
/* counter RNG (part of this build's spec; cogrid uses numpy's PCG64, */
/* whose draws are replaced by explicit override arrays in replay)    */
/* ------------------------------------------------------------------ */
static uint32_t mix32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352du;
    x ^= x >> 15; x *= 0x846ca68bu;
    x ^= x >> 16;
    return x;
}

uint32_t oc_rng_u32(uint64_t seed, int64_t env_index, uint32_t counter) {
    uint32_t x = mix32((uint32_t)seed + 0x9E3779B9u);
    x = mix32(x ^ (uint32_t)(seed >> 32));
    x = mix32(x ^ (uint32_t)((uint64_t)env_index));
    x = mix32(x ^ (uint32_t)((uint64_t)env_index >> 32));
    x = mix32(x ^ counter);
    return x;
}

static int rng_below(uint64_t seed, int64_t env_index, uint32_t *ctr, int n) {
    uint32_t r = oc_rng_u32(seed, env_index, (*ctr)++);
    return (int)(((uint64_t)r * (uint64_t)n) >> 32);
}

/* ------------------------------------------------------------------ */
/* layout                                                             */
/* ------------------------------------------------------------------ */
static int block_dim_for(int feature, int A) {
    switch (feature) {
    case OC_FEAT_GLOBAL:
    case OC_FEAT_LOCAL: return 99 + 2 * (A - 1);           /* overcooked_features.py:38-78 */
    case OC_FEAT_MINIMAL: return 4 + 3 + 4 + 16 + 2 + 4;   /* :291-298 */
    case OC_FEAT_MINIMAL_OTHER: return 33 + 2 * (A - 1);   /* :239-254 */
    case OC_FEAT_ONLY_DIRECTION: return 4;                 /* README.md:65,112 */
    }
    return -1;
}

int oc_layout_init(oc_layout *L, const char *const *rows, int H, int num_agents,
                   int feature, int max_steps, int dist_metric) {
    memset(L, 0, sizeof(*L));
    if (H <= 0 || num_agents <= 0 || num_agents > OC_MAX_AGENTS) return -1;
    int W = (int)strlen(rows[0]);
    if (W <= 0 || H * W > OC_MAX_CELLS) return -2;
    L->H = H; L->W = W; L->A = num_agents;
    L->max_steps = max_steps; L->feature = feature; L->dist_metric = dist_metric; L->auto_reset = 1;
    for (int r = 0; r < H; ++r) {
        if ((int)strlen(rows[r]) != W) return -3;
        for (int c = 0; c < W; ++c) {
            int cell = r * W + c;
            switch (rows[r][c]) {
            case '#': L->type[cell] = OC_WALL; break;
            case ' ': L->type[cell] = OC_FLOOR; break;
            case '+': L->type[cell] = OC_FLOOR; L->spawn[L->nspawn++] = cell; break;
            case 'C': L->type[cell] = OC_COUNTER; break;
            case 'U':
                L->type[cell] = OC_POT;
                if (L->npots >= OC_MAX_POTS) return -4;
                L->pot_cell[L->npots++] = cell;
                break;
            case 'O': L->type[cell] = OC_ONION_STACK; break;
            case '=': L->type[cell] = OC_PLATE_STACK; break;
            case '@': L->type[cell] = OC_DELIVERY; break;
            default: return -5;
            }
        }
    }
    /* every agent needs a start cell of its own: a '+' cell each when the layout names spawn points,
     * any floor cell otherwise */
    int nfloor = 0;
    for (int c = 0; c < H * W; ++c) nfloor += L->type[c] == OC_FLOOR;
    /* the grid must be walled in: no floor on its border (every registered layout is) */
    for (int r = 0; r < H; ++r)
        for (int c = 0; c < W; ++c)
            if ((r == 0 || r == H - 1 || c == 0 || c == W - 1) && L->type[r * W + c] == OC_FLOOR) return -9;
    if (L->nspawn > 0 && L->nspawn < num_agents) return -7;
    if (nfloor < num_agents) return -8;
    L->block_dim = block_dim_for(feature, num_agents);
    if (L->block_dim < 0) return -6;
    L->obs_dim = (feature == OC_FEAT_GLOBAL) ? num_agents * L->block_dim : L->block_dim;
    return 0;
}

int oc_obs_dim(const oc_layout *L) { return L->obs_dim; }

/* ------------------------------------------------------------------ */
/* per-env working state                                              */
/* ------------------------------------------------------------------ */
typedef struct {
    int t;
    uint32_t ctr;
    int row[OC_MAX_AGENTS], col[OC_MAX_AGENTS], dir[OC_MAX_AGENTS], inv[OC_MAX_AGENTS];
    int pot_count[OC_MAX_POTS], pot_timer[OC_MAX_POTS];
    uint8_t item[OC_MAX_CELLS];
} env_t;

static void env_load(const oc_layout *L, env_t *s, const int32_t *agents, const int32_t *pots,
                     const uint8_t *items, const int32_t *misc) {
    for (int a = 0; a < L->A; ++a) {
        s->row[a] = agents[a * 4 + 0]; s->col[a] = agents[a * 4 + 1];
        s->dir[a] = agents[a * 4 + 2]; s->inv[a] = agents[a * 4 + 3];
    }
    for (int p = 0; p < L->npots; ++p) { s->pot_count[p] = pots[p * 2]; s->pot_timer[p] = pots[p * 2 + 1]; }
    memcpy(s->item, items, (size_t)(L->H * L->W));
    s->t = misc[0];
    s->ctr = (uint32_t)misc[1];
}

static void env_store(const oc_layout *L, const env_t *s, int32_t *agents, int32_t *pots,
                      uint8_t *items, int32_t *misc) {
    for (int a = 0; a < L->A; ++a) {
        agents[a * 4 + 0] = s->row[a]; agents[a * 4 + 1] = s->col[a];
        agents[a * 4 + 2] = s->dir[a]; agents[a * 4 + 3] = s->inv[a];
    }
    for (int p = 0; p < L->npots; ++p) { pots[p * 2] = s->pot_count[p]; pots[p * 2 + 1] = s->pot_timer[p]; }
    memcpy(items, s->item, (size_t)(L->H * L->W));
    misc[0] = s->t;
    misc[1] = (int32_t)s->ctr;
}

static const int DROW[4] = {0, 1, 0, -1}; /* Right, Down, Left, Up */
static const int DCOL[4] = {1, 0, -1, 0};

static int pot_index(const oc_layout *L, int cell) {
    for (int p = 0; p < L->npots; ++p) if (L->pot_cell[p] == cell) return p;
    return -1;
}

/* "fwd_cell is None or fwd_cell.can_overlap()": a bare floor tile with nothing lying on it */
static int cell_walkable(const oc_layout *L, const env_t *s, int r, int c) {
    if (r < 0 || c < 0 || r >= L->H || c >= L->W) return 0;
    int cell = r * L->W + c;
    return L->type[cell] == OC_FLOOR && s->item[cell] == OC_NONE;
}

/* ------------------------------------------------------------------ */
/* reset                                                              */
/* ------------------------------------------------------------------ */
static void env_reset(const oc_layout *L, env_t *s, uint64_t seed, int64_t env_index,
                      const uint8_t *spawn_override, const uint8_t *dir_override) {
    /* the draw counter survives resets so successive episodes differ */
    s->t = 0;
    memset(s->item, 0, sizeof(s->item));
    for (int p = 0; p < L->npots; ++p) { s->pot_count[p] = 0; s->pot_timer[p] = OC_COOK_TIME; }
    int used_spawn[OC_MAX_CELLS];
    memset(used_spawn, 0, sizeof(used_spawn));
    for (int a = 0; a < L->A; ++a) {
        int cell;
        if (spawn_override) {
            cell = spawn_override[a];
        } else if (L->nspawn > 0) {
            /* uniformly among the '+' cells not taken yet, in scan order */
            int avail[OC_MAX_CELLS], n = 0;
            for (int k = 0; k < L->nspawn; ++k) if (!used_spawn[k]) avail[n++] = k;
            int k = avail[rng_below(seed, env_index, &s->ctr, n)];
            used_spawn[k] = 1;
            cell = L->spawn[k];
        } else {
            /* uniformly among free floor cells not occupied by earlier agents, scan order */
            int avail[OC_MAX_CELLS], n = 0;
            for (int c = 0; c < L->H * L->W; ++c) {
                if (L->type[c] != OC_FLOOR) continue;
                int taken = 0;
                for (int b = 0; b < a; ++b) if (s->row[b] * L->W + s->col[b] == c) taken = 1;
                if (!taken) avail[n++] = c;
            }
            cell = avail[rng_below(seed, env_index, &s->ctr, n)];
        }
        s->row[a] = cell / L->W; s->col[a] = cell % L->W;
        s->dir[a] = dir_override ? dir_override[a] : rng_below(seed, env_index, &s->ctr, 4);
        s->inv[a] = OC_NONE;
    }
}

/* ------------------------------------------------------------------ */
/* step                                                               */
/* ------------------------------------------------------------------ */
static void env_step(const oc_layout *L, env_t *s, uint64_t seed, int64_t env_index,
                     const int64_t *actions, const uint8_t *prio_override,
                     float *rewards, uint8_t *terminated, uint8_t *truncated) {
    const int A = L->A, W = L->W;
    /* previous state, kept for the reward functions (they look at state before the step) */
    env_t prev = *s;

    s->t += 1;

    /* grid.tick(): pots that are full keep cooking */
    for (int p = 0; p < L->npots; ++p)
        if (s->pot_count[p] == OC_POT_CAPACITY && s->pot_timer[p] > 0) s->pot_timer[p] -= 1;

    /* ---- move_agents ---- */
    int old_r[OC_MAX_AGENTS], old_c[OC_MAX_AGENTS];
    int want_r[OC_MAX_AGENTS], want_c[OC_MAX_AGENTS];
    int new_r[OC_MAX_AGENTS], new_c[OC_MAX_AGENTS], decided[OC_MAX_AGENTS];
    for (int a = 0; a < A; ++a) {
        old_r[a] = s->row[a]; old_c[a] = s->col[a];
        want_r[a] = old_r[a]; want_c[a] = old_c[a];
        int act = (int)actions[a];
        int d = -1;
        if (act == OC_ACT_UP) d = OC_DIR_UP;
        else if (act == OC_ACT_DOWN) d = OC_DIR_DOWN;
        else if (act == OC_ACT_LEFT) d = OC_DIR_LEFT;
        else if (act == OC_ACT_RIGHT) d = OC_DIR_RIGHT;
        if (d >= 0) {
            s->dir[a] = d; /* a move action always turns the agent, even when blocked */
            int fr = old_r[a] + DROW[d], fc = old_c[a] + DCOL[d];
            if (cell_walkable(L, s, fr, fc)) { want_r[a] = fr; want_c[a] = fc; }
        }
    }
    /* agents that stay where they are keep their cell unconditionally */
    int movers[OC_MAX_AGENTS], m = 0;
    for (int a = 0; a < A; ++a) {
        decided[a] = 0;
        if (want_r[a] == old_r[a] && want_c[a] == old_c[a]) {
            new_r[a] = old_r[a]; new_c[a] = old_c[a]; decided[a] = 1;
        } else {
            movers[m++] = a;
        }
    }
    /* the others are served in a random priority order */
    if (prio_override) {
        for (int i = 1; i < m; ++i) { /* insertion sort by (rank, id) */
            int x = movers[i], j = i - 1;
            while (j >= 0 && prio_override[movers[j]] > prio_override[x]) { movers[j + 1] = movers[j]; --j; }
            movers[j + 1] = x;
        }
    } else {
        for (int i = m - 1; i >= 1; --i) { /* Fisher-Yates */
            int j = rng_below(seed, env_index, &s->ctr, i + 1);
            int tmp = movers[i]; movers[i] = movers[j]; movers[j] = tmp;
        }
    }
    for (int k = 0; k < m; ++k) {
        int a = movers[k], taken = 0;
        for (int b = 0; b < A; ++b)
            if (decided[b] && new_r[b] == want_r[a] && new_c[b] == want_c[a]) taken = 1;
        if (taken) { new_r[a] = old_r[a]; new_c[a] = old_c[a]; }
        else { new_r[a] = want_r[a]; new_c[a] = want_c[a]; }
        decided[a] = 1;
    }
    /* two agents may not pass through each other */
    for (int a = 0; a < A; ++a)
        for (int b = a + 1; b < A; ++b)
            if (new_r[a] == old_r[b] && new_c[a] == old_c[b] && new_r[b] == old_r[a] && new_c[b] == old_c[a] &&
                !(new_r[a] == old_r[a] && new_c[a] == old_c[a])) {
                new_r[a] = old_r[a]; new_c[a] = old_c[a];
                new_r[b] = old_r[b]; new_c[b] = old_c[b];
            }
    /* with three or more agents a refused mover can be left on a cell that a
     * higher-priority agent has claimed; whoever moved onto an occupied cell goes back
     * (repeat: the retreat can expose the next overlap) */
    for (int pass = 0; pass < A; ++pass) {
        int changed = 0;
        for (int a = 0; a < A; ++a) {
            int moved = !(new_r[a] == old_r[a] && new_c[a] == old_c[a]);
            if (!moved) continue;
            for (int b = 0; b < A; ++b) {
                if (b == a) continue;
                if (new_r[b] == new_r[a] && new_c[b] == new_c[a]) {
                    new_r[a] = old_r[a]; new_c[a] = old_c[a]; changed = 1; break;
                }
            }
        }
        if (!changed) break;
    }
    for (int a = 0; a < A; ++a) { s->row[a] = new_r[a]; s->col[a] = new_c[a]; }

    /* ---- interact, in agent-id order ---- */
    for (int a = 0; a < A; ++a) {
        if ((int)actions[a] != OC_ACT_PICKUP_DROP) continue; /* Toggle does nothing to Overcooked objects */
        int fr = s->row[a] + DROW[s->dir[a]], fc = s->col[a] + DCOL[s->dir[a]];
        if (fr < 0 || fc < 0 || fr >= L->H || fc >= W) continue;
        int cell = fr * W + fc;
        int type = L->type[cell];
        int agent_ahead = 0;
        for (int b = 0; b < A; ++b) if (b != a && s->row[b] == fr && s->col[b] == fc) agent_ahead = 1;
        int p = (type == OC_POT) ? pot_index(L, cell) : -1;

        if (type == OC_FLOOR && s->item[cell] != OC_NONE && s->inv[a] == OC_NONE) {
            /* loose object in front: pick it up */
            s->inv[a] = s->item[cell]; s->item[cell] = OC_NONE;
        } else if (type == OC_ONION_STACK && s->inv[a] == OC_NONE) {
            s->inv[a] = OC_ONION;
        } else if (type == OC_PLATE_STACK && s->inv[a] == OC_NONE) {
            s->inv[a] = OC_PLATE;
        } else if (type == OC_COUNTER && s->item[cell] != OC_NONE && s->inv[a] == OC_NONE) {
            s->inv[a] = s->item[cell]; s->item[cell] = OC_NONE;
        } else if (type == OC_POT && s->inv[a] == OC_PLATE &&
                   s->pot_count[p] == OC_POT_CAPACITY && s->pot_timer[p] == 0) {
            /* plate + finished pot -> soup in hand, pot starts over */
            s->inv[a] = OC_SOUP; s->pot_count[p] = 0; s->pot_timer[p] = OC_COOK_TIME;
        } else if (s->inv[a] != OC_NONE) {
            if (type == OC_FLOOR && s->item[cell] == OC_NONE) {
                if (!agent_ahead) { s->item[cell] = (uint8_t)s->inv[a]; s->inv[a] = OC_NONE; }
            } else if (type == OC_COUNTER && s->item[cell] == OC_NONE) {
                s->item[cell] = (uint8_t)s->inv[a]; s->inv[a] = OC_NONE;
            } else if (type == OC_POT && s->inv[a] == OC_ONION && s->pot_count[p] < OC_POT_CAPACITY) {
                s->pot_count[p] += 1; s->inv[a] = OC_NONE;
            } else if (type == OC_DELIVERY && s->inv[a] == OC_SOUP) {
                s->inv[a] = OC_NONE;
            }
        }
    }

    /* ---- rewards: functions of (previous state, actions), accumulated in the
     * config's list order [onion_in_pot, soup_in_dish, delivery]
     * (overcooked_config.py:96) as Python floats, returned as float32 ---- */
    double rew[OC_MAX_AGENTS];
    for (int a = 0; a < A; ++a) rew[a] = 0.0;
    for (int pass = 0; pass < 3; ++pass) {
        for (int a = 0; a < A; ++a) {
            if ((int)actions[a] != OC_ACT_PICKUP_DROP) continue;
            int fr = prev.row[a] + DROW[prev.dir[a]], fc = prev.col[a] + DCOL[prev.dir[a]];
            if (fr < 0 || fc < 0 || fr >= L->H || fc >= W) continue;
            int cell = fr * W + fc;
            int type = L->type[cell];
            int p = (type == OC_POT) ? pot_index(L, cell) : -1;
            if (pass == 0) {
                if (prev.inv[a] == OC_ONION && type == OC_POT && prev.pot_count[p] < OC_POT_CAPACITY) rew[a] += 0.1;
            } else if (pass == 1) {
                if (prev.inv[a] == OC_PLATE && type == OC_POT &&
                    prev.pot_count[p] == OC_POT_CAPACITY && prev.pot_timer[p] == 0) rew[a] += 0.3;
            } else {
                if (prev.inv[a] == OC_SOUP && type == OC_DELIVERY)
                    for (int b = 0; b < A; ++b) rew[b] += 1.0; /* shared */
            }
        }
    }
    for (int a = 0; a < A; ++a) {
        rewards[a] = (float)rew[a];
        terminated[a] = 0;
        truncated[a] = (uint8_t)(s->t >= L->max_steps);
    }
}

/* ------------------------------------------------------------------ */
/* features                                                           */
/* ------------------------------------------------------------------ */
/* adjacency enumeration used by NextToCounter / NextToPot / CanMoveDirection:
 * index 0 = right, 1 = left, 2 = down, 3 = up (pinned by the golden vector) */
static const int ADJ_DR[4] = {0, 0, 1, -1};
static const int ADJ_DC[4] = {1, -1, 0, 0};

/* pot condition classes */
enum { POT_EMPTY = 0, POT_PARTIAL = 1, POT_COOKING = 2, POT_READY = 3 };
static int pot_class(const env_t *s, int p) {
    if (s->pot_count[p] == 0) return POT_EMPTY;
    if (s->pot_count[p] < OC_POT_CAPACITY) return POT_PARTIAL;
    return s->pot_timer[p] > 0 ? POT_COOKING : POT_READY;
}
/* NextToPot: 4 status rows x 4 adjacency columns; an empty pot is row 0 (golden vector) */
static const int NEXT_TO_POT_ROW[4] = {0, 1, 2, 3};
/* pot feature status one-hot; an empty pot sets index 1 (golden vector), cooking 2 and
 * ready 3 follow the docstring at overcooked_features.py:23 */
static const int POT_STATUS_INDEX[4] = {1, 0, 2, 3};
/* OvercookedInventory one-hot: onion is index 0 (golden vector) */
static int inventory_index(int inv) { return inv == OC_ONION ? 0 : inv == OC_SOUP ? 1 : 2; }

static int dist_of(const oc_layout *L, int dy, int dx) {
    if (L->dist_metric == 1) return dy * dy + dx * dx;
    return abs(dy) + abs(dx);
}

static float *feat_agent_dir(const env_t *s, int a, float *o) {
    for (int k = 0; k < 4; ++k) o[k] = 0.f;
    o[s->dir[a]] = 1.f;
    return o + 4;
}
static float *feat_inventory(const env_t *s, int a, float *o) {
    for (int k = 0; k < 3; ++k) o[k] = 0.f;
    if (s->inv[a] != OC_NONE) o[inventory_index(s->inv[a])] = 1.f;
    return o + 3;
}
static float *feat_next_to_counter(const oc_layout *L, const env_t *s, int a, float *o) {
    for (int k = 0; k < 4; ++k) {
        int r = s->row[a] + ADJ_DR[k], c = s->col[a] + ADJ_DC[k];
        o[k] = (r >= 0 && c >= 0 && r < L->H && c < L->W && L->type[r * L->W + c] == OC_COUNTER) ? 1.f : 0.f;
    }
    return o + 4;
}
static float *feat_next_to_pot(const oc_layout *L, const env_t *s, int a, float *o) {
    for (int k = 0; k < 16; ++k) o[k] = 0.f;
    for (int k = 0; k < 4; ++k) {
        int r = s->row[a] + ADJ_DR[k], c = s->col[a] + ADJ_DC[k];
        if (r < 0 || c < 0 || r >= L->H || c >= L->W) continue;
        int cell = r * L->W + c;
        if (L->type[cell] != OC_POT) continue;
        o[NEXT_TO_POT_ROW[pot_class(s, pot_index(L, cell))] * 4 + k] = 1.f;
    }
    return o + 16;
}
/* ClosestObj(kind, n): scan the grid in row-major order, collect every cell that IS the
 * object or carries it, sort by distance (stable => scan order on ties), emit
 * (agent - object) as (dy, dx) for the n nearest, zero padded.
 * is_item: kind is OC_ONION/PLATE/SOUP (looked up in the item layer), otherwise a cell type. */
static float *feat_closest(const oc_layout *L, const env_t *s, int a, int is_item, int kind, int n, float *o) {
    int cand_d[OC_MAX_CELLS], cand_dy[OC_MAX_CELLS], cand_dx[OC_MAX_CELLS], nc = 0;
    for (int cell = 0; cell < L->H * L->W; ++cell) {
        int match = is_item ? (s->item[cell] == kind) : (L->type[cell] == kind);
        if (!match) continue;
        int r = cell / L->W, c = cell % L->W;
        if (r == s->row[a] && c == s->col[a]) continue; /* never the agent's own cell */
        int dy = s->row[a] - r, dx = s->col[a] - c;
        cand_d[nc] = dist_of(L, dy, dx); cand_dy[nc] = dy; cand_dx[nc] = dx; ++nc;
    }
    for (int i = 1; i < nc; ++i) { /* stable insertion sort */
        int d = cand_d[i], dy = cand_dy[i], dx = cand_dx[i], j = i - 1;
        while (j >= 0 && cand_d[j] > d) {
            cand_d[j + 1] = cand_d[j]; cand_dy[j + 1] = cand_dy[j]; cand_dx[j + 1] = cand_dx[j]; --j;
        }
        cand_d[j + 1] = d; cand_dy[j + 1] = dy; cand_dx[j + 1] = dx;
    }
    for (int k = 0; k < n; ++k) {
        o[2 * k] = k < nc ? (float)cand_dy[k] : 0.f;
        o[2 * k + 1] = k < nc ? (float)cand_dx[k] : 0.f;
    }
    return o + 2 * n;
}
/* NClosestPotFeatures(num_pots): per pot
 * [reachable, status(4), contents, timer, dy, dx, row, col]; missing pots are all zero */
static float *feat_pots(const oc_layout *L, const env_t *s, int a, int num_pots, float *o) {
    int order[OC_MAX_POTS], dist[OC_MAX_POTS];
    for (int p = 0; p < L->npots; ++p) {
        int r = L->pot_cell[p] / L->W, c = L->pot_cell[p] % L->W;
        order[p] = p; dist[p] = dist_of(L, s->row[a] - r, s->col[a] - c);
    }
    for (int i = 1; i < L->npots; ++i) { /* stable */
        int x = order[i], j = i - 1;
        while (j >= 0 && dist[order[j]] > dist[x]) { order[j + 1] = order[j]; --j; }
        order[j + 1] = x;
    }
    for (int k = 0; k < num_pots; ++k) {
        float *q = o + 11 * k;
        for (int i = 0; i < 11; ++i) q[i] = 0.f;
        if (k >= L->npots) continue;
        int p = order[k];
        int r = L->pot_cell[p] / L->W, c = L->pot_cell[p] % L->W;
        q[0] = 1.f;
        q[1 + POT_STATUS_INDEX[pot_class(s, p)]] = 1.f;
        q[5] = (float)s->pot_count[p];
        q[6] = (s->pot_count[p] == OC_POT_CAPACITY) ? (float)s->pot_timer[p] : -1.f;
        q[7] = (float)(s->row[a] - r); q[8] = (float)(s->col[a] - c);
        q[9] = (float)r; q[10] = (float)c;
    }
    return o + 11 * num_pots;
}
static float *feat_dist_to_others(const oc_layout *L, const env_t *s, int a, float *o) {
    int k = 0;
    for (int b = 0; b < L->A; ++b) {
        if (b == a) continue;
        o[2 * k] = (float)(s->row[a] - s->row[b]);
        o[2 * k + 1] = (float)(s->col[a] - s->col[b]);
        ++k;
    }
    return o + 2 * (L->A - 1);
}
static float *feat_position(const env_t *s, int a, float *o) {
    o[0] = (float)s->row[a]; o[1] = (float)s->col[a];
    return o + 2;
}
static float *feat_can_move(const oc_layout *L, const env_t *s, int a, float *o) {
    for (int k = 0; k < 4; ++k)
        o[k] = cell_walkable(L, s, s->row[a] + ADJ_DR[k], s->col[a] + ADJ_DC[k]) ? 1.f : 0.f;
    return o + 4;
}

/* one agent's block, overcooked_features.py:38-78 / :239-254 / :291-298 */
static float *player_encoding(const oc_layout *L, const env_t *s, int a, float *o) {
    switch (L->feature) {
    case OC_FEAT_GLOBAL:
    case OC_FEAT_LOCAL:
        o = feat_agent_dir(s, a, o);
        o = feat_inventory(s, a, o);
        o = feat_next_to_counter(L, s, a, o);
        o = feat_next_to_pot(L, s, a, o);
        o = feat_closest(L, s, a, 1, OC_ONION, 4, o);
        o = feat_closest(L, s, a, 1, OC_PLATE, 4, o);
        o = feat_closest(L, s, a, 0, OC_PLATE_STACK, 2, o);
        o = feat_closest(L, s, a, 0, OC_ONION_STACK, 2, o);
        o = feat_closest(L, s, a, 1, OC_SOUP, 4, o);
        o = feat_closest(L, s, a, 0, OC_DELIVERY, 2, o);
        o = feat_closest(L, s, a, 0, OC_COUNTER, 4, o);
        o = feat_pots(L, s, a, 2, o);
        o = feat_dist_to_others(L, s, a, o);
        o = feat_position(s, a, o);
        o = feat_can_move(L, s, a, o);
        break;
    case OC_FEAT_MINIMAL_OTHER:
        o = feat_agent_dir(s, a, o);
        o = feat_inventory(s, a, o);
        o = feat_next_to_counter(L, s, a, o);
        o = feat_next_to_pot(L, s, a, o);
        o = feat_dist_to_others(L, s, a, o);
        o = feat_position(s, a, o);
        o = feat_can_move(L, s, a, o);
        break;
    case OC_FEAT_MINIMAL:
        o = feat_agent_dir(s, a, o);
        o = feat_inventory(s, a, o);
        o = feat_next_to_counter(L, s, a, o);
        o = feat_next_to_pot(L, s, a, o);
        o = feat_position(s, a, o);
        o = feat_can_move(L, s, a, o);
        break;
    case OC_FEAT_ONLY_DIRECTION:
        o = feat_agent_dir(s, a, o);
        break;
    }
    return o;
}

/* observation row of agent a: own block first, then the others in agent-id order
 * (overcooked_features.py:110-117); the non-global sets are the own block only */
static void featurize_agent(const oc_layout *L, const env_t *s, int a, float *o) {
    o = player_encoding(L, s, a, o);
    if (L->feature == OC_FEAT_GLOBAL)
        for (int b = 0; b < L->A; ++b)
            if (b != a) o = player_encoding(L, s, b, o);
}

/* ------------------------------------------------------------------ */
/* batched drivers                                                    */
/* ------------------------------------------------------------------ */
static void set_threads(int nthreads) {
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#else
    (void)nthreads;
#endif
}

void oc_reset_batch(const oc_layout *L, int E, uint64_t seed, int64_t env_index_base,
                    int32_t *agents, int32_t *pots, uint8_t *items, int32_t *misc,
                    const uint8_t *spawn_override, const uint8_t *dir_override,
                    float *obs, int nthreads) {
    const int A = L->A, NP = L->npots, NC = L->H * L->W, D = L->obs_dim;
    set_threads(nthreads);
#pragma omp parallel for schedule(static)
    for (int e = 0; e < E; ++e) {
        env_t s;
        memset(&s, 0, sizeof(s));
        s.ctr = (uint32_t)misc[e * 2 + 1];
        env_reset(L, &s, seed, env_index_base + e,
                  spawn_override ? spawn_override + (size_t)e * A : NULL,
                  dir_override ? dir_override + (size_t)e * A : NULL);
        env_store(L, &s, agents + (size_t)e * A * 4, pots + (size_t)e * NP * 2, items + (size_t)e * NC, misc + (size_t)e * 2);
        if (obs)
            for (int a = 0; a < A; ++a) featurize_agent(L, &s, a, obs + ((size_t)e * A + a) * D);
    }
}

void oc_step_batch(const oc_layout *L, int E, uint64_t seed, int64_t env_index_base,
                   int32_t *agents, int32_t *pots, uint8_t *items, int32_t *misc,
                   const int64_t *actions, const uint8_t *prio_override,
                   float *obs, float *rewards, uint8_t *terminated, uint8_t *truncated,
                   int nthreads) {
    const int A = L->A, NP = L->npots, NC = L->H * L->W, D = L->obs_dim;
    set_threads(nthreads);
#pragma omp parallel for schedule(static)
    for (int e = 0; e < E; ++e) {
        env_t s;
        env_load(L, &s, agents + (size_t)e * A * 4, pots + (size_t)e * NP * 2, items + (size_t)e * NC, misc + (size_t)e * 2);
        env_step(L, &s, seed, env_index_base + e, actions + (size_t)e * A,
                 prio_override ? prio_override + (size_t)e * A : NULL,
                 rewards + (size_t)e * A, terminated + (size_t)e * A, truncated + (size_t)e * A);
        /* vector wrapper: when every agent of the env is done, reset it at once and
         * hand back the first observation of the new episode with the done flags */
        int all_done = 1;
        for (int a = 0; a < A; ++a)
            if (!(terminated[(size_t)e * A + a] || truncated[(size_t)e * A + a])) all_done = 0;
        if (all_done && L->auto_reset) env_reset(L, &s, seed, env_index_base + e, NULL, NULL);
        env_store(L, &s, agents + (size_t)e * A * 4, pots + (size_t)e * NP * 2, items + (size_t)e * NC, misc + (size_t)e * 2);
        if (obs)
            for (int a = 0; a < A; ++a) featurize_agent(L, &s, a, obs + ((size_t)e * A + a) * D);
    }
}

void oc_featurize_batch(const oc_layout *L, int E, const int32_t *agents, const int32_t *pots,
                        const uint8_t *items, const int32_t *misc, float *obs, int nthreads) {
    const int A = L->A, NP = L->npots, NC = L->H * L->W, D = L->obs_dim;
    set_threads(nthreads);
#pragma omp parallel for schedule(static)
    for (int e = 0; e < E; ++e) {
        env_t s;
        env_load(L, &s, agents + (size_t)e * A * 4, pots + (size_t)e * NP * 2, items + (size_t)e * NC, misc + (size_t)e * 2);
        for (int a = 0; a < A; ++a) featurize_agent(L, &s, a, obs + ((size_t)e * A + a) * D);
    }
}

/* ------------------------------------------------------------------ */
/* GAE: every product and sum rounded to float32 in the order the      */
/* reference's tensor expression evaluates them                        */
/* ------------------------------------------------------------------ */
void oc_gae(const float *rewards, const float *dones, const float *values,
            const float *next_values, int T, int N, double gamma_d, double lam_d,
            float *adv, float *ret, int mode) {
    /* Python holds gamma and lam as doubles: gamma * lam is a double product, and each
     * scalar is rounded to float32 only when it meets a float32 tensor */
    const float gamma = (float)gamma_d;
    const float gl = (float)(gamma_d * lam_d);
    float *last = (float *)calloc((size_t)N, sizeof(float));
    for (int t = T - 1; t >= 0; --t) {
        /* MAPPO.py:102-107: the last row uses its OWN done flag (dones[-1]) */
        const float *drow = dones + (size_t)(t == T - 1 ? T - 1 : t + 1) * N;
        volatile float rmean = 0.f;
        if (mode == 1) {
            /* CentralizedMAPPO.py:77: rewards[t].mean() over all rows */
            double acc = 0.0;
            for (int n = 0; n < N; ++n) acc += (double)rewards[(size_t)t * N + n];
            rmean = (float)(acc / (double)N);
        }
        for (int n = 0; n < N; ++n) {
            volatile float mask = 1.0f - drow[n];
            float nv;
            if (t == T - 1) nv = (mode == 1) ? next_values[0] : next_values[n];
            else nv = values[(size_t)(t + 1) * N + n];
            float r = (mode == 1) ? rmean : rewards[(size_t)t * N + n];
            volatile float a1 = gamma * nv;
            volatile float a2 = a1 * mask;
            volatile float a3 = r + a2;
            volatile float delta = a3 - values[(size_t)t * N + n];
            volatile float b1 = gl * mask;
            volatile float b2 = b1 * last[n];
            volatile float out = delta + b2;
            last[n] = out;
            adv[(size_t)t * N + n] = out;
            if (ret) {
                volatile float rr = out + values[(size_t)t * N + n]; /* MAPPO.py:144 */
                ret[(size_t)t * N + n] = rr;
            }
        }
    }
    free(last);
}
