// Fused Overcooked step + featurize kernel for gfx950 (MI355X).
//
// Work decomposition: one lane per observation row, i.e. per (env, agent).  The A lanes of
// an env each replay the env's (tiny, integer) dynamics in registers from the packed
// struct-of-arrays state, so no lane ever waits for another; lane 0 of the env writes the
// state back.  Each lane then builds its own agent's feature block as packed int8 in
// registers (static part from a per-floor-cell template staged in LDS, dynamic part OR-ed
// in at compile-time byte offsets) and drops it into the wave's int8 image of the output
// rows in LDS.  Finally the wave streams that image out: one aligned LDS dword (4 features)
// per lane per iteration -> 4 floats -> one 16-byte store, so every store instruction writes
// 1 KiB of consecutive HBM bytes.  The float32 observation write is >90 % of the algorithmic
// bytes of this path; everything else is sized to stay out of its way.
//
// Replaces, for the reference's drivers, the third-party env.step()/reset() reached at
// agent_environment.py:34,50,88 and the feature composition of
// overcooked_features.py:38-78,107-130 (one block per agent; own block first, then the
// others in agent-id order).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "oc_common.h"

namespace oc {

// ---------------------------------------------------------------------------------------
// small helpers
// ---------------------------------------------------------------------------------------
// Compile-time loop.  Register arrays below are only ever touched through it: a run-time
// `for` over a small array (even under #pragma unroll) lets the optimiser fold
// "if (k == j) a[k] = x" into the indexed store a[j] = x, which parks the array in scratch.
template <int B, int E, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (B < E) {
        f(std::integral_constant<int, B>{});
        static_for<B + 1, E>(f);
    }
}
__device__ __forceinline__ uint32_t mix32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352du;
    x ^= x >> 15; x *= 0x846ca68bu;
    x ^= x >> 16;
    return x;
}

struct Rng {
    uint32_t key;
    uint32_t ctr;
    __device__ __forceinline__ uint32_t below(uint32_t n) {
        uint32_t r = mix32(key ^ ctr);
        ++ctr;
        return __umulhi(r, n);
    }
};

template <int NIW>
__device__ __forceinline__ uint32_t item_get(const uint32_t (&it)[NIW], uint32_t slot) {
    uint32_t w = it[0];
    static_for<1, NIW>([&](auto K) { w = ((slot >> 4) == (uint32_t)K.value) ? it[K.value] : w; });
    return (w >> ((slot & 15u) * 2u)) & 3u;
}

template <int NIW>
__device__ __forceinline__ void item_set(uint32_t (&it)[NIW], uint32_t slot, uint32_t v) {
    const uint32_t sh = (slot & 15u) * 2u;
    const uint32_t keep = ~(3u << sh);
    const uint32_t val = v << sh;
    static_for<0, NIW>([&](auto K) {
        const uint32_t hit = (slot >> 4) == (uint32_t)K.value ? 0xFFFFFFFFu : 0u;
        it[K.value] = (it[K.value] & (keep | ~hit)) | (val & hit);
    });
}

// runtime-indexed read of a small register array
template <int A>
__device__ __forceinline__ uint32_t pick(const uint32_t (&v)[A], uint32_t i) {
    uint32_t r = v[0];
    static_for<1, A>([&](auto K) { r = (i == (uint32_t)K.value) ? v[K.value] : r; });
    return r;
}

__device__ __forceinline__ int dir_delta(uint32_t d, int W) {
    int s = (d & 1u) ? W : 1;  // right +1, down +W, left -1, up -W
    return (d & 2u) ? -s : s;
}

// adjacency enumeration shared by NextToCounter / NextToPot / CanMoveDirection:
// 0 = right, 1 = left, 2 = down, 3 = up (pinned by the reference's golden vector)
__device__ __forceinline__ int adj_delta(int k, int W) {
    return k == 0 ? 1 : k == 1 ? -1 : k == 2 ? W : -W;
}

__device__ __forceinline__ uint32_t pot_count(uint64_t pots, uint32_t p) { return (uint32_t)(pots >> (8u * p)) & 3u; }
__device__ __forceinline__ uint32_t pot_timer(uint64_t pots, uint32_t p) { return (uint32_t)(pots >> (8u * p + 2u)) & 31u; }
__device__ __forceinline__ uint32_t pot_class(uint64_t pots, uint32_t p) {
    const uint32_t c = pot_count(pots, p), t = pot_timer(pots, p);
    return c == 0 ? POT_EMPTY : c < (uint32_t)POT_CAPACITY ? POT_PARTIAL : t > 0 ? POT_COOKING : POT_READY;
}

struct Lds {
    const uint8_t* b;
    __device__ __forceinline__ uint32_t celltype(uint32_t c) const { return b[TAB_CELLTYPE + c]; }
    __device__ __forceinline__ uint32_t slot(uint32_t c) const { return b[TAB_SLOT_OF_CELL + c]; }
    __device__ __forceinline__ uint32_t pot(uint32_t c) const { return b[TAB_POT_OF_CELL + c]; }
    __device__ __forceinline__ uint32_t fidx(uint32_t c) const { return b[TAB_FIDX_OF_CELL + c]; }
    __device__ __forceinline__ uint32_t rc(uint32_t c) const {
        return reinterpret_cast<const uint16_t*>(b + TAB_CELL_RC)[c];
    }
    __device__ __forceinline__ uint32_t cell_of_slot(uint32_t s) const { return b[TAB_CELL_OF_SLOT + s]; }
    __device__ __forceinline__ uint32_t floor_cell(uint32_t i) const { return b[TAB_FLOOR_CELL + i]; }
    __device__ __forceinline__ uint32_t spawn_cell(uint32_t i) const { return b[TAB_SPAWN_CELL + i]; }
    __device__ __forceinline__ uint32_t pot_order(uint32_t c, int k) const { return b[TAB_POT_ORDER + 2 * c + k]; }
};

// a bare floor tile with nothing lying on it
template <int NIW>
__device__ __forceinline__ bool walkable(const Lds& T, const uint32_t (&items)[NIW], uint32_t c) {
    return T.celltype(c) == FLOOR && item_get<NIW>(items, T.slot(c)) == NONE;
}

// ---------------------------------------------------------------------------------------
// env state in registers
// ---------------------------------------------------------------------------------------
template <int A, int NIW>
struct EnvState {
    uint32_t t;
    Rng rng;
    uint64_t pots;
    uint32_t cell[A], dir[A], inv[A];
    uint32_t items[NIW];
};

template <int A, int NIW>
__device__ __forceinline__ void load_state(EnvState<A, NIW>& s, const uint32_t* __restrict__ st, int E, int e, int NP) {
    const uint32_t* p = st + e;
    const size_t stride = (size_t)E;
    s.t = p[0] & 0xFFFFu;
    s.rng.ctr = p[stride];
    s.pots = p[2 * stride];
    const int base = NP > 4 ? 4 : 3;
    if (NP > 4) s.pots |= (uint64_t)p[3 * stride] << 32;
    static_for<0, (A + 1) / 2>([&](auto K) {
        constexpr int k = K.value;
        const uint32_t aw = p[(size_t)(base + k) * stride];
        s.cell[2 * k] = aw & 0xFFu; s.dir[2 * k] = (aw >> 8) & 3u; s.inv[2 * k] = (aw >> 10) & 3u;
        if constexpr (2 * k + 1 < A) {
            s.cell[2 * k + 1] = (aw >> 16) & 0xFFu; s.dir[2 * k + 1] = (aw >> 24) & 3u; s.inv[2 * k + 1] = (aw >> 26) & 3u;
        }
    });
    static_for<0, NIW>([&](auto K) { s.items[K.value] = p[(size_t)(base + (A + 1) / 2 + K.value) * stride]; });
}

template <int A, int NIW>
__device__ __forceinline__ void store_state(const EnvState<A, NIW>& s, uint32_t* __restrict__ st, int E, int e, int NP) {
    uint32_t* p = st + e;
    const size_t stride = (size_t)E;
    p[0] = s.t;
    p[stride] = s.rng.ctr;
    p[2 * stride] = (uint32_t)s.pots;
    const int base = NP > 4 ? 4 : 3;
    if (NP > 4) p[3 * stride] = (uint32_t)(s.pots >> 32);
    static_for<0, (A + 1) / 2>([&](auto K) {
        constexpr int k = K.value;
        uint32_t aw = s.cell[2 * k] | (s.dir[2 * k] << 8) | (s.inv[2 * k] << 10);
        if constexpr (2 * k + 1 < A) aw |= (s.cell[2 * k + 1] << 16) | (s.dir[2 * k + 1] << 24) | (s.inv[2 * k + 1] << 26);
        p[(size_t)(base + k) * stride] = aw;
    });
    static_for<0, NIW>([&](auto K) { p[(size_t)(base + (A + 1) / 2 + K.value) * stride] = s.items[K.value]; });
}

// ---------------------------------------------------------------------------------------
// reset: empty kitchen, agents on random free floor cells (or on the '+' cells when the
// layout has them), random facing.  Draws come from the counter RNG unless overridden.
// ---------------------------------------------------------------------------------------
template <int A, int NIW>
__device__ __forceinline__ void reset_env(EnvState<A, NIW>& s, const Lds& T, const EnvParams& P,
                                          const uint8_t* spawn_ov, const uint8_t* dir_ov, int e) {
    s.t = 0;
    static_for<0, NIW>([&](auto K) { s.items[K.value] = 0; });
    s.pots = 0;
    for (int p = 0; p < P.NP; ++p) s.pots |= (uint64_t)(COOK_TIME << 2) << (8 * p);
    // indices (into the spawn or floor list) already handed out, kept sorted ascending
    uint32_t taken[A];
    static_for<0, A>([&](auto K) { taken[K.value] = 0xFFFFFFFFu; });
    static_for<0, A>([&](auto AI) {
        constexpr int a = AI.value;
        uint32_t c;
        if (spawn_ov) {
            c = spawn_ov[(size_t)e * A + a];
        } else {
            const bool use_spawn = P.nspawn > 0;
            const uint32_t n = (use_spawn ? P.nspawn : P.nfloor) - a;
            uint32_t idx = s.rng.below(n);
            // idx-th entry of the list once the taken ones are skipped
            static_for<0, a>([&](auto B) { idx += (taken[B.value] <= idx) ? 1u : 0u; });
            c = use_spawn ? T.spawn_cell(idx) : T.floor_cell(idx);
            // insert idx, keeping taken[0..a] sorted
            uint32_t x = idx;
            static_for<0, a + 1>([&](auto B) {
                const uint32_t lo = min(taken[B.value], x), hi = max(taken[B.value], x);
                taken[B.value] = lo; x = hi;
            });
        }
        s.cell[a] = c;
        s.dir[a] = dir_ov ? (uint32_t)dir_ov[(size_t)e * A + a] : s.rng.below(4);
        s.inv[a] = NONE;
    });
}

// ---------------------------------------------------------------------------------------
// one env step; the three masks say which agents met each reward condition
// ---------------------------------------------------------------------------------------
template <int A, int NIW>
__device__ __forceinline__ void step_env(EnvState<A, NIW>& s, const Lds& T, const EnvParams& P,
                                         const uint32_t (&act)[A], const uint8_t* prio, int e,
                                         uint32_t& onion_mask, uint32_t& dish_mask, uint32_t& deliver_mask) {
    const int W = P.W;
    // what the reward functions look at: holdings and pots BEFORE this step
    uint32_t prev_inv[A];
    static_for<0, A>([&](auto K) { prev_inv[K.value] = s.inv[K.value]; });
    const uint64_t prev_pots = s.pots;

    s.t += 1;
    // pots that are full keep cooking
    for (int p = 0; p < P.NP; ++p)
        if (pot_count(s.pots, p) == POT_CAPACITY && pot_timer(s.pots, p) > 0) s.pots -= (uint64_t)4 << (8 * p);

    // ---- movement ----
    uint32_t want[A], npos[A];
    uint32_t movers = 0;  // packed 4-bit agent ids
    uint32_t m = 0;
    uint32_t decided = 0;
    static_for<0, A>([&](auto AI) {
        constexpr int a = AI.value;
        want[a] = s.cell[a];
        if (act[a] < 4u) {
            const uint32_t d = (0x0213u >> (4u * act[a])) & 3u;  // up,down,left,right -> 3,1,2,0
            s.dir[a] = d;  // a move action always turns the agent
            const uint32_t f = (uint32_t)((int)s.cell[a] + dir_delta(d, W));
            if (walkable<NIW>(T, s.items, f)) want[a] = f;
        }
        npos[a] = s.cell[a];
        if (want[a] == s.cell[a]) {
            decided |= 1u << a;  // stayers keep their cell unconditionally
        } else {
            movers |= (uint32_t)a << (4u * m);
            ++m;
        }
    });
    if (prio) {
        // replay mode: serve movers by ascending (rank, id)
        uint32_t sorted = 0, left = 0;
        static_for<0, A>([&](auto K) { if ((uint32_t)K.value < m) left |= 1u << ((movers >> (4 * K.value)) & 15u); });
        static_for<0, A>([&](auto K) {
            if ((uint32_t)K.value < m) {
                uint32_t best = 0xFFFFFFFFu;
                static_for<0, A>([&](auto AI) {
                    if (left & (1u << AI.value))
                        best = min(best, ((uint32_t)prio[(size_t)e * A + AI.value] << 4) | (uint32_t)AI.value);
                });
                sorted |= (best & 15u) << (4 * K.value);
                left &= ~(1u << (best & 15u));
            }
        });
        movers = sorted;
    } else {
        // Fisher-Yates on the packed list, i = m-1 .. 1
        static_for<1, A>([&](auto K) {
            constexpr int i = A - K.value;  // A-1 .. 1
            if ((uint32_t)i < m) {
                const uint32_t j = s.rng.below(i + 1);
                const uint32_t x = ((movers >> (4 * i)) ^ (movers >> (4 * j))) & 15u;
                movers ^= (x << (4 * i)) | (x << (4 * j));
            }
        });
    }
    static_for<0, A>([&](auto K) {
        if ((uint32_t)K.value < m) {
            const uint32_t a = (movers >> (4 * K.value)) & 15u;
            const uint32_t wa = pick<A>(want, a);
            bool taken = false;
            static_for<0, A>([&](auto B) { taken |= ((decided >> B.value) & 1u) && npos[B.value] == wa; });
            static_for<0, A>([&](auto B) {
                npos[B.value] = (a == (uint32_t)B.value && !taken) ? wa : npos[B.value];
            });
            decided |= 1u << a;
        }
    });
    // no passing through each other
    static_for<0, A>([&](auto AI) {
        static_for<AI.value + 1, A>([&](auto BI) {
            constexpr int a = AI.value, b = BI.value;
            if (npos[a] == s.cell[b] && npos[b] == s.cell[a] && npos[a] != s.cell[a]) {
                npos[a] = s.cell[a]; npos[b] = s.cell[b];
            }
        });
    });
    // three or more agents: whoever moved onto a cell somebody was left standing on goes back
    if constexpr (A > 2) {
        static_for<0, A>([&](auto) {
            static_for<0, A>([&](auto AI) {
                constexpr int a = AI.value;
                bool clash = false;
                static_for<0, A>([&](auto BI) { if (BI.value != a) clash |= npos[BI.value] == npos[a]; });
                if (clash && npos[a] != s.cell[a]) npos[a] = s.cell[a];
            });
        });
    }
    static_for<0, A>([&](auto K) { s.cell[K.value] = npos[K.value]; });

    // ---- interact (agent-id order) and reward conditions (previous state) ----
    onion_mask = dish_mask = deliver_mask = 0;
    static_for<0, A>([&](auto AI) {
        constexpr int a = AI.value;
        if (act[a] == ACT_PICKUP_DROP) {
            const uint32_t f = (uint32_t)((int)s.cell[a] + dir_delta(s.dir[a], W));
            const uint32_t ct = T.celltype(f);
            const uint32_t slot = T.slot(f);
            const bool holds_items = (ct == FLOOR || ct == COUNTER);
            const uint32_t lying = holds_items ? item_get<NIW>(s.items, slot) : (uint32_t)NONE;
            const uint32_t p = T.pot(f);
            // a PickupDrop agent neither moved nor turned, so f is also its previous front cell
            if (ct == POT) {
                if (prev_inv[a] == ONION && pot_count(prev_pots, p) < (uint32_t)POT_CAPACITY) onion_mask |= 1u << a;
                if (prev_inv[a] == PLATE && pot_count(prev_pots, p) == (uint32_t)POT_CAPACITY && pot_timer(prev_pots, p) == 0)
                    dish_mask |= 1u << a;
            }
            if (ct == DELIVERY && prev_inv[a] == SOUP) deliver_mask |= 1u << a;

            bool ahead = false;
            static_for<0, A>([&](auto BI) { if (BI.value != a) ahead |= s.cell[BI.value] == f; });

            const uint32_t inv = s.inv[a];
            if (holds_items && lying != NONE && inv == NONE) {
                s.inv[a] = lying; item_set<NIW>(s.items, slot, NONE);
            } else if (ct == ONION_STACK && inv == NONE) {
                s.inv[a] = ONION;
            } else if (ct == PLATE_STACK && inv == NONE) {
                s.inv[a] = PLATE;
            } else if (ct == POT && inv == PLATE && pot_count(s.pots, p) == (uint32_t)POT_CAPACITY && pot_timer(s.pots, p) == 0) {
                s.inv[a] = SOUP;
                s.pots = (s.pots & ~((uint64_t)0xFF << (8 * p))) | ((uint64_t)(COOK_TIME << 2) << (8 * p));
            } else if (inv != NONE) {
                if (ct == FLOOR && lying == NONE) {
                    if (!ahead) { item_set<NIW>(s.items, slot, inv); s.inv[a] = NONE; }
                } else if (ct == COUNTER && lying == NONE) {
                    item_set<NIW>(s.items, slot, inv); s.inv[a] = NONE;
                } else if (ct == POT && inv == ONION && pot_count(s.pots, p) < (uint32_t)POT_CAPACITY) {
                    s.pots += (uint64_t)1 << (8 * p); s.inv[a] = NONE;
                } else if (ct == DELIVERY && inv == SOUP) {
                    s.inv[a] = NONE;
                }
            }
        }
    });
}

// ---------------------------------------------------------------------------------------
// feature block of agent i, packed int8 in NW dwords
// ---------------------------------------------------------------------------------------
template <int NW>
struct Block {
    uint32_t w[NW];
    template <int OFF>
    __device__ __forceinline__ void put(int v) {
        static_assert(OFF / 4 < NW, "offset outside block");
        w[OFF / 4] |= ((uint32_t)v & 0xFFu) << ((OFF % 4) * 8);
    }
    // byte OFF0 + j, j known only at run time, j in [0, RANGE)
    template <int OFF0, int RANGE>
    __device__ __forceinline__ void put_dyn(uint32_t j, int v) {
        const uint32_t off = OFF0 + j;
        const uint32_t val = ((uint32_t)v & 0xFFu) << ((off & 3u) * 8u);
        static_for<OFF0 / 4, (OFF0 + RANGE - 1) / 4 + 1>([&](auto K) {
            w[K.value] |= ((off >> 2) == (uint32_t)K.value) ? val : 0u;
        });
    }
};

// sorted (ascending) top-4 of 32-bit keys
struct Top4 {
    uint32_t k0, k1, k2, k3;
    __device__ __forceinline__ void init() { k0 = k1 = k2 = k3 = 0xFFFFFFFFu; }
    __device__ __forceinline__ void insert(uint32_t x) {
        uint32_t lo;
        lo = min(k0, x); x = max(k0, x); k0 = lo;
        lo = min(k1, x); x = max(k1, x); k1 = lo;
        lo = min(k2, x); x = max(k2, x); k2 = lo;
        k3 = min(k3, x);
    }
};

template <int OFF, int NW>
__device__ __forceinline__ void emit_closest(Block<NW>& blk, const Lds& T, uint32_t key, int ar, int ac) {
    if (key != 0xFFFFFFFFu) {
        const uint32_t rc = T.rc(key & 0xFFu);
        blk.template put<OFF>(ar - (int)(rc >> 8));
        blk.template put<OFF + 1>(ac - (int)(rc & 0xFFu));
    }
}

template <int OFF, int NW>
__device__ __forceinline__ void emit_pot(Block<NW>& blk, uint32_t p, uint64_t pots) {
    if (p != 0xFFu) {
        const uint32_t cls = pot_class(pots, p);
        blk.template put_dyn<OFF + 1, 4>((POT_STATUS_INDEX_PACKED >> (4 * cls)) & 15u, 1);
        const uint32_t cnt = pot_count(pots, p);
        blk.template put<OFF + 5>((int)cnt);
        blk.template put<OFF + 6>(cnt == (uint32_t)POT_CAPACITY ? (int)pot_timer(pots, p) : -1);
    }
}

template <int A, int FEAT, int NIW>
__device__ __forceinline__ void build_block(Block<FeatGeom<A, FEAT>::NW>& blk, const EnvState<A, NIW>& s,
                                            const Lds& T, const EnvParams& P, uint32_t i) {
    using G = FeatGeom<A, FEAT>;
    constexpr int NW = G::NW;
    const uint32_t mycell = pick<A>(s.cell, i);
    const uint32_t mydir = pick<A>(s.dir, i);
    const uint32_t myinv = pick<A>(s.inv, i);

    // static part of the block for an agent standing on this floor cell
    const uint32_t* tmpl = reinterpret_cast<const uint32_t*>(T.b + TAB_POSTAB) + T.fidx(mycell) * NW;
    static_for<0, NW>([&](auto K) { blk.w[K.value] = tmpl[K.value]; });

    blk.template put_dyn<G::O_DIR, 4>(mydir, 1);
    if constexpr (G::kSpatial) {
        if (myinv != NONE) blk.template put_dyn<G::O_INV, 3>((INVENTORY_INDEX_PACKED >> (4 * myinv)) & 15u, 1);

        const uint32_t rc = T.rc(mycell);
        const int ar = (int)(rc >> 8), ac = (int)(rc & 0xFFu);

        // NextToPot (4 status rows x 4 adjacency columns) and CanMoveDirection
        static_for<0, 4>([&](auto K) {
            constexpr int k = K.value;
            const uint32_t nb = (uint32_t)((int)mycell + adj_delta(k, P.W));
            const uint32_t ct = T.celltype(nb);
            if (ct == POT) {
                const uint32_t row = (NEXT_TO_POT_ROW_PACKED >> (4 * pot_class(s.pots, T.pot(nb)))) & 15u;
                blk.template put_dyn<G::O_NEXT_POT, 16>(row * 4 + k, 1);
            }
            if (ct == FLOOR && item_get<NIW>(s.items, T.slot(nb)) == NONE) blk.template put<G::O_CAN_MOVE + k>(1);
        });

        if constexpr (G::kOthers && A > 1) {
            // (self - other) for the other agents in id order
            static_for<0, A>([&](auto BI) {
                constexpr int b = BI.value;
                if ((uint32_t)b != i) {
                    const uint32_t j = (uint32_t)b < i ? (uint32_t)b : (uint32_t)b - 1u;
                    const uint32_t orc = T.rc(s.cell[b]);
                    blk.template put_dyn<G::O_OTHERS, 2 * (A - 1)>(2 * j, ar - (int)(orc >> 8));
                    blk.template put_dyn<G::O_OTHERS, 2 * (A - 1)>(2 * j + 1, ac - (int)(orc & 0xFFu));
                }
            });
        }

        if constexpr (G::kFull) {
            // ClosestObj for the movable things: walk the occupied item slots (slot order is
            // row-major cell order), keep the four nearest per kind; ties go to the earlier cell
            Top4 onion, plate, soup;
            onion.init(); plate.init(); soup.init();
            static_for<0, NIW>([&](auto K) {
                const uint32_t word = s.items[K.value];
                uint32_t nz = (word | (word >> 1)) & 0x55555555u;
                while (nz) {
                    const uint32_t bit = __builtin_ctz(nz);
                    nz &= nz - 1;
                    const uint32_t kind = (word >> bit) & 3u;
                    const uint32_t c = T.cell_of_slot(K.value * 16 + (bit >> 1));
                    const uint32_t orc = T.rc(c);
                    const int dy = ar - (int)(orc >> 8), dx = ac - (int)(orc & 0xFFu);
                    const uint32_t d = P.dist_metric ? (uint32_t)(dy * dy + dx * dx) : (uint32_t)(abs(dy) + abs(dx));
                    const uint32_t key = (d << 8) | c;
                    onion.insert(kind == ONION ? key : 0xFFFFFFFFu);
                    plate.insert(kind == PLATE ? key : 0xFFFFFFFFu);
                    soup.insert(kind == SOUP ? key : 0xFFFFFFFFu);
                }
            });
            emit_closest<G::O_ONION + 0>(blk, T, onion.k0, ar, ac);
            emit_closest<G::O_ONION + 2>(blk, T, onion.k1, ar, ac);
            emit_closest<G::O_ONION + 4>(blk, T, onion.k2, ar, ac);
            emit_closest<G::O_ONION + 6>(blk, T, onion.k3, ar, ac);
            emit_closest<G::O_PLATE + 0>(blk, T, plate.k0, ar, ac);
            emit_closest<G::O_PLATE + 2>(blk, T, plate.k1, ar, ac);
            emit_closest<G::O_PLATE + 4>(blk, T, plate.k2, ar, ac);
            emit_closest<G::O_PLATE + 6>(blk, T, plate.k3, ar, ac);
            emit_closest<G::O_SOUP + 0>(blk, T, soup.k0, ar, ac);
            emit_closest<G::O_SOUP + 2>(blk, T, soup.k1, ar, ac);
            emit_closest<G::O_SOUP + 4>(blk, T, soup.k2, ar, ac);
            emit_closest<G::O_SOUP + 6>(blk, T, soup.k3, ar, ac);

            // status / contents / timer of the two nearest pots (their static fields are in the template)
            emit_pot<G::O_POT0>(blk, T.pot_order(mycell, 0), s.pots);
            emit_pot<G::O_POT1>(blk, T.pot_order(mycell, 1), s.pots);
        }
    }
}

// copy a BLK-byte block (NW dwords in registers) to an arbitrary byte address in LDS
template <int BLK, int NW>
__device__ __forceinline__ void write_block(uint8_t* img, uint32_t dst, const uint32_t (&w)[NW]) {
    const uint32_t sh = dst & 3u;
    uint32_t* base = reinterpret_cast<uint32_t*>(img + (dst - sh));
    const uint32_t end = sh + BLK;         // first byte past the block, relative to base
    const uint32_t kt = end >> 2;          // first dword not fully covered
    const uint32_t tc = end & 3u;          // bytes of the block in dword kt
    static_for<0, NW + 1>([&](auto K) {
        constexpr int k = K.value;
        uint32_t hi = 0u, lo = 0u;
        if constexpr (k < NW) hi = w[k];
        if constexpr (k > 0) lo = w[k - 1];
        // bytes [4k - sh, 4k - sh + 4) of the block: a funnel shift of the register pair
        const uint32_t v = sh ? __builtin_amdgcn_alignbit(hi, lo, 32u - 8u * sh) : hi;
        uint8_t* bp = reinterpret_cast<uint8_t*>(base + k);
        if constexpr (k == 0) {
            if (sh == 0) base[0] = v;
            else {
                if (sh <= 1) bp[1] = (uint8_t)(v >> 8);
                if (sh <= 2) bp[2] = (uint8_t)(v >> 16);
                bp[3] = (uint8_t)(v >> 24);
            }
        } else {
            if ((uint32_t)k < kt) {
                base[k] = v;
            } else if ((uint32_t)k == kt) {
                if (tc >= 1) bp[0] = (uint8_t)v;
                if (tc >= 2) bp[1] = (uint8_t)(v >> 8);
                if (tc >= 3) bp[2] = (uint8_t)(v >> 16);
            }
        }
    });
}

// ---------------------------------------------------------------------------------------
// the kernel.  mode: 0 = step (+ auto-reset on truncation), 1 = reset every env,
// 2 = featurize the stored state only.
// ---------------------------------------------------------------------------------------
template <int A, int FEAT, int NIW>
__global__ void __launch_bounds__(256)
env_kernel(EnvParams P, const uint8_t* __restrict__ tab_g, uint32_t* __restrict__ state,
           const void* __restrict__ actions, int act_size, const uint8_t* __restrict__ prio,
           const uint8_t* __restrict__ spawn_ov, const uint8_t* __restrict__ dir_ov,
           float* __restrict__ obs, float* __restrict__ rewards, uint8_t* __restrict__ terminated,
           uint8_t* __restrict__ truncated, int mode) {
    using G = FeatGeom<A, FEAT>;
    constexpr int D = G::D, BLK = G::BLK, NW = G::NW;
    extern __shared__ __align__(16) uint8_t smem[];

    // stage the lookup blob
    {
        const uint4* src = reinterpret_cast<const uint4*>(tab_g);
        uint4* dst = reinterpret_cast<uint4*>(smem);
        for (int k = threadIdx.x; k < P.tab_bytes / 16; k += blockDim.x) dst[k] = src[k];
    }
    __syncthreads();
    Lds T{smem};

    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    uint8_t* img = smem + P.tab_bytes + wave * (64 * D);

    const size_t N = (size_t)P.E * A;
    const size_t row0 = ((size_t)blockIdx.x * (blockDim.x >> 6) + wave) * 64;  // first row of this wave
    const size_t row = row0 + lane;
    const bool active = row < N;

    const int ablate = mode >> 8;  // profiling aid: 1 = skip the store stream, 2 = skip the simulation
    mode &= 0xFF;
    if (active && !(ablate & 2)) {
        const int e = (int)(row / A);
        const uint32_t i = (uint32_t)(row % A);
        EnvState<A, NIW> s;
        load_state<A, NIW>(s, state, P.E, e, P.NP);
        {
            const uint64_t g = (((uint64_t)P.env_base_hi << 32) | P.env_base_lo) + (uint64_t)e;
            uint32_t k = mix32(P.seed_key ^ (uint32_t)g);
            s.rng.key = mix32(k ^ (uint32_t)(g >> 32));
        }
        if (mode == 0) {
            uint32_t act[A];
            static_for<0, A>([&](auto K) {
                const size_t r = (size_t)e * A + K.value;
                act[K.value] = act_size == 8 ? (uint32_t) reinterpret_cast<const long long*>(actions)[r]
                             : act_size == 4 ? (uint32_t) reinterpret_cast<const int*>(actions)[r]
                                             : (uint32_t) reinterpret_cast<const uint8_t*>(actions)[r];
            });
            uint32_t onion_mask, dish_mask, deliver_mask;
            onion_mask = dish_mask = deliver_mask = 0;
            if (!(ablate & 4)) step_env<A, NIW>(s, T, P, act, prio, e, onion_mask, dish_mask, deliver_mask);
            // rewards accumulate as Python floats in the config's list order
            // [onion_in_pot 0.1, soup_in_dish 0.3, delivery 1.0 shared] (overcooked_config.py:96)
            double r = 0.0;
            if ((onion_mask >> i) & 1u) r += 0.1;
            if ((dish_mask >> i) & 1u) r += 0.3;
            static_for<0, A>([&](auto K) { if ((deliver_mask >> K.value) & 1u) r += 1.0; });
            const bool trunc = s.t >= (uint32_t)P.max_steps;
            rewards[row] = (float)r;
            terminated[row] = 0;
            truncated[row] = trunc ? 1 : 0;
            // vector-env semantics: a finished env restarts at once and reports the first
            // observation of the new episode together with the done flags
            if (trunc) reset_env<A, NIW>(s, T, P, nullptr, nullptr, e);
            if (i == 0) store_state<A, NIW>(s, state, P.E, e, P.NP);
        } else if (mode == 1) {
            reset_env<A, NIW>(s, T, P, spawn_ov, dir_ov, e);
            if (i == 0) store_state<A, NIW>(s, state, P.E, e, P.NP);
        }

        Block<NW> blk;
        if (!(ablate & 8)) build_block<A, FEAT, NIW>(blk, s, T, P, i);
        else static_for<0, NW>([&](auto K) { blk.w[K.value] = s.items[0]; });
        if (ablate & 16) {
            if (blk.w[0] == 0x12345678u) img[lane] = 1;
        } else if constexpr (FEAT == FEAT_GLOBAL) {
            // my block goes into every row of my env: first in my own, after the lower ids elsewhere
            const uint32_t env_lane0 = (uint32_t)lane - i;
            static_for<0, A>([&](auto J) {
                constexpr uint32_t j = J.value;
                const uint32_t slot = (j == i) ? 0u : (i < j ? i + 1u : i);
                write_block<BLK, NW>(img, (env_lane0 + j) * D + slot * BLK, blk.w);
            });
        } else {
            write_block<BLK, NW>(img, (uint32_t)lane * D, blk.w);
        }
    }
    __syncthreads();

    // stream the wave's rows out: 4 int8 -> 4 floats -> one 16-byte store per lane
    if (row0 < N && !(ablate & 1)) {
        const size_t rows_here = (N - row0) < 64 ? (N - row0) : 64;
        const uint32_t nchunk = (uint32_t)(rows_here * D) / 4;  // 64*D is a multiple of 4; tails handled below
        const uint32_t* img32 = reinterpret_cast<const uint32_t*>(img);
        float4* out = reinterpret_cast<float4*>(obs + row0 * D);
        for (uint32_t q = lane; q < nchunk; q += 64) {
            const uint32_t v = img32[q];
            float4 f;
            f.x = (float)(int)(int8_t)(v);
            f.y = (float)(int)(int8_t)(v >> 8);
            f.z = (float)(int)(int8_t)(v >> 16);
            f.w = (float)(int)(int8_t)(v >> 24);
            out[q] = f;
        }
        // a short last group can leave up to 3 floats past the last full chunk
        const uint32_t rem = (uint32_t)(rows_here * D) & 3u;
        if (rem && lane < (int)rem) {
            const uint32_t idx = nchunk * 4 + lane;
            obs[row0 * D + idx] = (float)(int)(int8_t)img[idx];
        }
    }
}

// ---------------------------------------------------------------------------------------
// dispatch
// ---------------------------------------------------------------------------------------
template <int A, int FEAT, int NIW>
static hipError_t launch_one(const EnvParams& P, const uint8_t* tab, uint32_t* state, const void* actions,
                             int act_size, const uint8_t* prio, const uint8_t* spawn_ov, const uint8_t* dir_ov,
                             float* obs, float* rewards, uint8_t* term, uint8_t* trunc, int mode,
                             hipStream_t stream) {
    using G = FeatGeom<A, FEAT>;
    // 4 waves per workgroup unless their row images would not leave room for two workgroups per CU
    int waves = 4;
    while (waves > 1 && (size_t)P.tab_bytes + (size_t)waves * 64 * G::D > 64 * 1024) waves >>= 1;
    const size_t lds = (size_t)P.tab_bytes + (size_t)waves * 64 * G::D;
    static size_t configured[64] = {0};  // per instantiation and device
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return hipErrorInvalidDevice;
    if (lds > configured[dev]) {
        hipError_t err = hipFuncSetAttribute(reinterpret_cast<const void*>(&env_kernel<A, FEAT, NIW>),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (err != hipSuccess) return err;
        configured[dev] = lds;
    }
    const size_t N = (size_t)P.E * A;
    const size_t rows_per_block = (size_t)waves * 64;
    const unsigned grid = (unsigned)((N + rows_per_block - 1) / rows_per_block);
    hipLaunchKernelGGL((env_kernel<A, FEAT, NIW>), dim3(grid), dim3(waves * 64), lds, stream, P, tab, state,
                       actions, act_size, prio, spawn_ov, dir_ov, obs, rewards, term, trunc, mode);
    return hipGetLastError();
}

template <int A, int FEAT>
static hipError_t launch_niw(int niw, const EnvParams& P, const uint8_t* tab, uint32_t* state,
                             const void* actions, int act_size, const uint8_t* prio, const uint8_t* spawn_ov,
                             const uint8_t* dir_ov, float* obs, float* rewards, uint8_t* term, uint8_t* trunc,
                             int mode, hipStream_t stream) {
#define OC_GO(NIW_) return launch_one<A, FEAT, NIW_>(P, tab, state, actions, act_size, prio, spawn_ov, dir_ov, obs, rewards, term, trunc, mode, stream)
    switch (niw) {
        case 1: OC_GO(1);
        case 2: OC_GO(2);
        case 4: OC_GO(4);
        case 8: OC_GO(8);
    }
#undef OC_GO
    return hipErrorInvalidValue;
}

template <int A>
hipError_t launch_env_kernel_A(int feat, int niw, const EnvParams& P, const uint8_t* tab, uint32_t* state,
                               const void* actions, int act_size, const uint8_t* prio, const uint8_t* spawn_ov,
                               const uint8_t* dir_ov, float* obs, float* rewards, uint8_t* term, uint8_t* trunc,
                               int mode, hipStream_t stream) {
#define OC_GO(FEAT_) return launch_niw<A, FEAT_>(niw, P, tab, state, actions, act_size, prio, spawn_ov, dir_ov, obs, rewards, term, trunc, mode, stream)
    switch (feat) {
        case FEAT_GLOBAL: OC_GO(FEAT_GLOBAL);
        case FEAT_LOCAL: OC_GO(FEAT_LOCAL);
        case FEAT_MINIMAL: OC_GO(FEAT_MINIMAL);
        case FEAT_MINIMAL_OTHER: OC_GO(FEAT_MINIMAL_OTHER);
        case FEAT_ONLY_DIRECTION: OC_GO(FEAT_ONLY_DIRECTION);
    }
#undef OC_GO
    return hipErrorInvalidValue;
}

#ifndef OC_AGENTS
#error "compile with -DOC_AGENTS=<1|2|4>: one translation unit per agent count"
#endif
template hipError_t launch_env_kernel_A<OC_AGENTS>(int, int, const EnvParams&, const uint8_t*, uint32_t*, const void*,
                                                   int, const uint8_t*, const uint8_t*, const uint8_t*, float*, float*,
                                                   uint8_t*, uint8_t*, int, hipStream_t);

}  // namespace oc
