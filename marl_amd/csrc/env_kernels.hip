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
// bytes of this path, and at 65,536 envs the whole launch is a single round of waves: the
// simulation does not hide behind the store stream, it delays it, and what it costs is the
// length of the dependent chain a wave runs before its first store (about one VALU
// instruction per 8 cycles, plus every LDS round trip on the way).  Hence the style below: everything an agent needs to know
// about the cells around it comes from 16-byte LDS reads of per-floor-cell records built on the
// host, rule chains are host-built decision tables, and the dynamics are selects, not branches.
// An env takes 1, 2 or 4 adjacent lanes (with three agents the fourth has no row), so its lanes
// share a wave and the wave's row image is private to it.  With three or four agents replaying
// everything in every lane costs too much: there the env's four lanes split the step and
// trade results through DPP moves (step_env_quad).
//
// Replaces, for the reference's drivers, the third-party env.step()/reset() reached at
// agent_environment.py:34,50,88 and the feature composition of
// overcooked_features.py:38-78,107-130 (one block per agent; own block first, then the
// others in agent-id order).
#include <hip/hip_runtime.h>
// Profiling aid: with -DOC_STAMPS the priority-override argument is taken as a uint64 buffer
// [waves][16] that receives each wave's cycle counter after every phase, the 100 MHz wall clock at
// start / simulation done / end, and where the wave ran (no override then).
#ifdef OC_STAMPS
#define OC_STAMP(k) stamp[k] = __builtin_readcyclecounter()
#else
#define OC_STAMP(k)
#endif

#include <stdint.h>
#include <stdlib.h>

#include <algorithm>
#include <mutex>
#include <type_traits>

#include "oc_common.h"

#ifndef OC_DEFER_SMALL_STORES
#define OC_DEFER_SMALL_STORES 1  // three and four agents: rewards / flags / state stored after the row stream
#endif

// Cache policy of the observation stores (buffer-instruction aux bits on gfx950: 1 = sc0,
// 2 = nt, 16 = sc1).  sc1 = write-through: what a launch writes does not stay dirty in the
// XCD L2s, so the end-of-kernel write-back has nothing left to flush.  Measured per launch on
// cramped_room, 2 agents, global_obs: 65,536 envs (106 MB of rows) plain 20.3 us, nt 23.8,
// sc1 19.3, sc1+nt 23.6; 262,144 envs (424 MB) plain 85.3 us, nt 82.7, sc1 83.3, sc1+nt 79.5.
constexpr int STORE_NT = 2, STORE_SC1 = 16;

namespace oc {
// State, rewards and done flags are stored write-through as well (a relaxed system-scope atomic
// store compiles to a plain store with the sc0 sc1 bits): nothing of a launch is left dirty in
// the XCD L2s for the end-of-kernel write-back (18.92 -> 18.74 us on the headline config).
template <class T> __device__ __forceinline__ void wt_store(T* p, T v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }

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
    // draw in [0, n) if `take`, else leave the counter alone (the value is then unused)
    __device__ __forceinline__ uint32_t below(uint32_t n, bool take = true) {
        const uint32_t r = mix32(key ^ ctr);
        ctr += take ? 1u : 0u;
        return __umulhi(r, n);
    }
};

template <int NIW>
__device__ __forceinline__ uint32_t item_get(const uint32_t (&it)[NIW], uint32_t slot) {
    uint32_t w = it[0];
    static_for<1, NIW>([&](auto K) { w = ((slot >> 4) == (uint32_t)K.value) ? it[K.value] : w; });
    return (w >> ((slot & 15u) * 2u)) & 3u;
}

// items[slot] = v where `on`; nothing otherwise
template <int NIW>
__device__ __forceinline__ void item_set(uint32_t (&it)[NIW], uint32_t slot, uint32_t v, bool on) {
    const uint32_t sh = (slot & 15u) * 2u;
    const uint32_t field = 3u << sh;
    const uint32_t val = v << sh;
    static_for<0, NIW>([&](auto K) {
        const uint32_t m = (on && (slot >> 4) == (uint32_t)K.value) ? field : 0u;
        it[K.value] = (it[K.value] & ~m) | (val & m);
    });
}

// runtime-indexed read of a small register array
template <int A>
__device__ __forceinline__ uint32_t pick(const uint32_t (&v)[A], uint32_t i) {
    uint32_t r = v[0];
    static_for<1, A>([&](auto K) { r = (i == (uint32_t)K.value) ? v[K.value] : r; });
    return r;
}

__device__ __forceinline__ uint32_t sel4(const uint4& v, uint32_t d) {
    const uint32_t lo = (d & 1u) ? v.y : v.x;
    const uint32_t hi = (d & 1u) ? v.w : v.z;
    return (d & 2u) ? hi : lo;
}

__device__ __forceinline__ uint32_t nb_type(uint32_t w) { return w & 7u; }
__device__ __forceinline__ uint32_t nb_slot(uint32_t w) { return (w >> 3) & 127u; }
__device__ __forceinline__ uint32_t nb_pot(uint32_t w) { return (w >> 10) & 3u; }  // meaningful when the type is POT
__device__ __forceinline__ uint32_t nb_fidx(uint32_t w) { return (w >> 16) & 255u; }

__device__ __forceinline__ uint32_t pot_count(uint32_t pots, uint32_t p) { return (pots >> (8u * p)) & 3u; }
__device__ __forceinline__ uint32_t pot_timer(uint32_t pots, uint32_t p) { return (pots >> (8u * p + 2u)) & 31u; }
__device__ __forceinline__ uint32_t pot_class(uint32_t pots, uint32_t p) {
    const uint32_t c = pot_count(pots, p), t = pot_timer(pots, p);
    // branch-free: the nested conditional compiles to divergent control flow otherwise
    static_assert(POT_EMPTY == 0 && POT_PARTIAL == 1 && POT_COOKING == 2 && POT_READY == 3, "class codes");
    const uint32_t filling = min(c, 1u);            // empty 0, partial 1
    const uint32_t full = 3u - min(t, 1u);          // cooking 2, ready 3
    return c == (uint32_t)POT_CAPACITY ? full : filling;
}

// One cooking tick for all pots at once (a byte per pot: count in bits 0-1, timer in bits 2-6;
// bytes of pots the layout does not have are zero and stay zero): a full pot (count 3) whose
// timer is above zero loses one timer step.
__device__ __forceinline__ uint32_t tick_pots(uint32_t pots) {
    static_assert(POT_CAPACITY == 3, "full means both count bits set");
    const uint32_t full = pots & (pots >> 1) & 0x01010101u;
    const uint32_t timer = (pots >> 2) & 0x1F1F1F1Fu;
    const uint32_t running = ((timer + 0x1F1F1F1Fu) >> 5) & 0x01010101u;  // timer + 31 reaches 32 unless it is 0
    return pots - ((full & running) << 2);
}

// pot condition as the decision tables see it
__device__ __forceinline__ uint32_t pot_state(uint32_t pots, uint32_t p) {
    const uint32_t c = pot_count(pots, p), t = pot_timer(pots, p);
    static_assert(POTST_ROOM == 0 && POTST_COOKING == 1 && POTST_READY == 2, "state codes");
    return c == (uint32_t)POT_CAPACITY ? 2u - min(t, 1u) : 0u;
}

struct Lds {
    const uint8_t* b;
    const EnvParams& P;
    __device__ __forceinline__ uint32_t int_entry(uint32_t idx) const { return b[P.off_inttab + idx]; }
    __device__ __forceinline__ uint32_t rew_entry(uint32_t idx) const { return b[P.off_rewtab + idx]; }
    __device__ __forceinline__ float rew_value(uint32_t idx) const { return reinterpret_cast<const float*>(b + P.off_rewval)[idx]; }
    __device__ __forceinline__ uint4 nb(uint32_t f) const { return reinterpret_cast<const uint4*>(b + P.off_nb)[f]; }
    __device__ __forceinline__ uint4 pot_adj(uint32_t f) const { return reinterpret_cast<const uint4*>(b + P.off_adj)[2 * f]; }
    __device__ __forceinline__ uint4 adj_info(uint32_t f) const { return reinterpret_cast<const uint4*>(b + P.off_adj)[2 * f + 1]; }
    __device__ __forceinline__ uint32_t frc(uint32_t f) const {
        return reinterpret_cast<const uint16_t*>(b + P.off_frc)[f];
    }
    __device__ __forceinline__ uint32_t slotrc(uint32_t s) const {
        return reinterpret_cast<const uint16_t*>(b + P.off_slotrc)[s];
    }
    __device__ __forceinline__ uint32_t pot_order(uint32_t f) const {  // two pot ids, low byte = nearest
        return reinterpret_cast<const uint16_t*>(b + P.off_potorder)[f];
    }
    __device__ __forceinline__ uint32_t spawnf(uint32_t i) const { return b[P.off_spawnf + i]; }
};

// ---------------------------------------------------------------------------------------
// env state in registers
// ---------------------------------------------------------------------------------------
template <int A, int NIW>
struct EnvState {
    uint32_t t;
    Rng rng;
    uint32_t pots;
    uint32_t f[A], dir[A], inv[A];  // floor index, facing, holding
    uint32_t items[NIW];
};

constexpr int state_words_ct(int A, int NIW) { return 3 + (A + 1) / 2 + NIW; }

template <int A, int NIW>
__device__ __forceinline__ void load_raw(uint32_t (&raw)[3 + (A + 1) / 2 + NIW], const uint32_t* __restrict__ st, int E, int e) {
    const uint32_t* p = st + e;
    static_for<0, 3 + (A + 1) / 2 + NIW>([&](auto K) { raw[K.value] = p[(size_t)K.value * (size_t)E]; });
}

template <int A, int NIW>
__device__ __forceinline__ void unpack_state(EnvState<A, NIW>& s, const uint32_t (&raw)[3 + (A + 1) / 2 + NIW]) {
    s.t = raw[0];  // the whole word: without auto-reset the counter runs on past max_steps
    s.rng.ctr = raw[1];
    s.pots = raw[2];
    static_for<0, (A + 1) / 2>([&](auto K) {
        constexpr int k = K.value;
        const uint32_t aw = raw[3 + k];
        s.f[2 * k] = aw & 0xFFu; s.dir[2 * k] = (aw >> 8) & 3u; s.inv[2 * k] = (aw >> 10) & 3u;
        if constexpr (2 * k + 1 < A) {
            s.f[2 * k + 1] = (aw >> 16) & 0xFFu; s.dir[2 * k + 1] = (aw >> 24) & 3u; s.inv[2 * k + 1] = (aw >> 26) & 3u;
        }
    });
    static_for<0, NIW>([&](auto K) { s.items[K.value] = raw[3 + (A + 1) / 2 + K.value]; });
}

template <int A, int NIW>
__device__ __forceinline__ void store_state(const EnvState<A, NIW>& s, uint32_t* __restrict__ st, int E, int e) {
    uint32_t* p = st + e;
    const size_t stride = (size_t)E;
    wt_store(p, s.t);
    wt_store(p + stride, s.rng.ctr);
    wt_store(p + 2 * stride, s.pots);
    static_for<0, (A + 1) / 2>([&](auto K) {
        constexpr int k = K.value;
        uint32_t aw = s.f[2 * k] | (s.dir[2 * k] << 8) | (s.inv[2 * k] << 10);
        if constexpr (2 * k + 1 < A) aw |= (s.f[2 * k + 1] << 16) | (s.dir[2 * k + 1] << 24) | (s.inv[2 * k + 1] << 26);
        wt_store(p + (size_t)(3 + k) * stride, aw);
    });
    static_for<0, NIW>([&](auto K) { wt_store(p + (size_t)(3 + (A + 1) / 2 + K.value) * stride, s.items[K.value]); });
}

// the same in two halves, for a caller that stores later than it packs
template <int A, int NIW>
__device__ __forceinline__ void pack_state(uint32_t (&raw)[3 + (A + 1) / 2 + NIW], const EnvState<A, NIW>& s) {
    raw[0] = s.t; raw[1] = s.rng.ctr; raw[2] = s.pots;
    static_for<0, (A + 1) / 2>([&](auto K) {
        constexpr int k = K.value;
        uint32_t aw = s.f[2 * k] | (s.dir[2 * k] << 8) | (s.inv[2 * k] << 10);
        if constexpr (2 * k + 1 < A) aw |= (s.f[2 * k + 1] << 16) | (s.dir[2 * k + 1] << 24) | (s.inv[2 * k + 1] << 26);
        raw[3 + k] = aw;
    });
    static_for<0, NIW>([&](auto K) { raw[3 + (A + 1) / 2 + K.value] = s.items[K.value]; });
}
template <int A, int NIW>
__device__ __forceinline__ void store_raw(const uint32_t (&raw)[3 + (A + 1) / 2 + NIW], uint32_t* __restrict__ st, int E, int e) {
    uint32_t* p = st + e;
    static_for<0, 3 + (A + 1) / 2 + NIW>([&](auto K) { wt_store(p + (size_t)K.value * (size_t)E, raw[K.value]); });
}

// ---------------------------------------------------------------------------------------
// reset: empty kitchen, agents on random free floor cells (or on the '+' cells when the
// layout has them), random facing.  Draws come from the counter RNG unless overridden
// (spawn_ov names cells; the cell -> floor index table is read from global memory, rare path).
// ---------------------------------------------------------------------------------------
template <int A, int NIW>
__device__ __forceinline__ void reset_env(EnvState<A, NIW>& s, const Lds& T, const EnvParams& P,
                                          const uint8_t* spawn_ov, const uint8_t* dir_ov, const uint8_t* aux_g, int e) {
    s.t = 0;
    static_for<0, NIW>([&](auto K) { s.items[K.value] = 0; });
    s.pots = 0;
    for (int p = 0; p < P.NP; ++p) s.pots |= (uint32_t)(COOK_TIME << 2) << (8 * p);
    // indices (into the spawn list or the floor list) already handed out, kept sorted ascending
    uint32_t taken[A];
    static_for<0, A>([&](auto K) { taken[K.value] = 0xFFFFFFFFu; });
    static_for<0, A>([&](auto AI) {
        constexpr int a = AI.value;
        uint32_t fi;
        if (spawn_ov) {
            fi = aux_g[AUX_FIDX_OF_CELL + spawn_ov[(size_t)e * A + a]];  // overrides name cells
            fi = fi == 0xFFu ? 0u : fi;  // not a floor cell: the first floor cell, so that every lookup stays inside the tables
        } else {
            const bool use_spawn = P.nspawn > 0;
            const uint32_t n = (use_spawn ? P.nspawn : P.nfloor) - a;
            uint32_t idx = s.rng.below(n);
            // idx-th entry of the list once the taken ones are skipped
            static_for<0, a>([&](auto B) { idx += (taken[B.value] <= idx) ? 1u : 0u; });
            fi = use_spawn ? T.spawnf(idx) : idx;
            // insert idx, keeping taken[0..a] sorted
            uint32_t x = idx;
            static_for<0, a + 1>([&](auto B) {
                const uint32_t lo = min(taken[B.value], x), hi = max(taken[B.value], x);
                taken[B.value] = lo; x = hi;
            });
        }
        s.f[a] = fi;
        s.dir[a] = dir_ov ? ((uint32_t)dir_ov[(size_t)e * A + a] & 3u) : s.rng.below(4);
        s.inv[a] = NONE;
    });
}

// ---------------------------------------------------------------------------------------
// one env step, every lane of the env computing all of it (one or two agents); returns the
// reward-table index of agent i: own onion-in-pot | own soup-plated << 1 | deliveries in the env << 2
// ---------------------------------------------------------------------------------------
template <int A, int NIW>
__device__ __forceinline__ uint32_t step_env(EnvState<A, NIW>& s, const Lds& T, const EnvParams& P,
                                             const uint32_t (&act)[A], const uint8_t* prio, int e, uint32_t i) {
    uint32_t onion_mask, dish_mask, deliver_mask;
    // the four neighbour words of every agent's cell: all the map knowledge this step needs
    uint4 nb[A];
    static_for<0, A>([&](auto K) { nb[K.value] = T.nb(s.f[K.value]); });

    // what the reward functions look at: holdings and pots BEFORE this step
    uint32_t prev_inv[A];
    static_for<0, A>([&](auto K) { prev_inv[K.value] = s.inv[K.value]; });
    const uint32_t prev_pots = s.pots;

    s.t += 1;
    // pots that are full keep cooking
    s.pots = tick_pots(s.pots);

    // ---- movement ----
    uint32_t want[A];
    static_for<0, A>([&](auto AI) {
        constexpr int a = AI.value;
        const bool is_move = act[a] < 4u;
        const uint32_t d = (0x0213u >> (4u * (act[a] & 3u))) & 3u;  // up,down,left,right -> 3,1,2,0
        s.dir[a] = is_move ? d : s.dir[a];                           // a move action always turns the agent
        const uint32_t w = sel4(nb[a], d);
        const bool open = nb_type(w) == FLOOR && item_get<NIW>(s.items, nb_slot(w)) == NONE;
        want[a] = (is_move && open) ? nb_fidx(w) : s.f[a];
    });
    if constexpr (A == 2) {
        // two agents: the general procedure below in closed form
        const bool m0 = want[0] != s.f[0], m1 = want[1] != s.f[1];
        const bool both = m0 && m1;
        bool first1;  // agent 1 is served before agent 0
        if (prio) {
            first1 = both && prio[(size_t)e * 2 + 1] < prio[(size_t)e * 2];
        } else {
            const uint32_t j = s.rng.below(2, both);  // Fisher-Yates step i = 1: swap when j == 0
            first1 = both && j == 0;
        }
        uint32_t n0 = want[0], n1 = want[1];
        // a lone mover may not enter the cell its partner keeps
        n0 = (m0 && !m1 && want[0] == s.f[1]) ? s.f[0] : n0;
        n1 = (m1 && !m0 && want[1] == s.f[0]) ? s.f[1] : n1;
        // same target: whoever is served second stays
        const bool same = both && want[0] == want[1];
        n0 = (same && first1) ? s.f[0] : n0;
        n1 = (same && !first1) ? s.f[1] : n1;
        // no passing through each other
        const bool swap = both && want[0] == s.f[1] && want[1] == s.f[0];
        s.f[0] = swap ? s.f[0] : n0;
        s.f[1] = swap ? s.f[1] : n1;
    } else {
        uint32_t npos[A];
        uint32_t movers = 0;  // packed 4-bit agent ids
        uint32_t m = 0;
        uint32_t decided = 0;
        static_for<0, A>([&](auto AI) {
            constexpr int a = AI.value;
            npos[a] = s.f[a];
            const bool stays = want[a] == s.f[a];
            decided |= stays ? (1u << a) : 0u;  // stayers keep their cell unconditionally
            movers |= stays ? 0u : ((uint32_t)a << (4u * m));
            m += stays ? 0u : 1u;
        });
        if constexpr (A > 1) {
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
                    const bool on = (uint32_t)i < m;
                    const uint32_t j = s.rng.below(i + 1, on);
                    const uint32_t x = on ? (((movers >> (4 * i)) ^ (movers >> (4 * j))) & 15u) : 0u;
                    movers ^= (x << (4 * i)) | (x << (4 * j));
                });
            }
        }
        static_for<0, A>([&](auto K) {
            const bool on = (uint32_t)K.value < m;
            const uint32_t a = (movers >> (4 * K.value)) & 15u;
            const uint32_t wa = pick<A>(want, a);
            bool taken = false;
            static_for<0, A>([&](auto B) { taken |= ((decided >> B.value) & 1u) && npos[B.value] == wa; });
            static_for<0, A>([&](auto B) {
                npos[B.value] = (on && a == (uint32_t)B.value && !taken) ? wa : npos[B.value];
            });
            decided |= on ? (1u << a) : 0u;
        });
        // no passing through each other
        static_for<0, A>([&](auto AI) {
            static_for<AI.value + 1, A>([&](auto BI) {
                constexpr int a = AI.value, b = BI.value;
                const bool swap = npos[a] == s.f[b] && npos[b] == s.f[a] && npos[a] != s.f[a];
                npos[a] = swap ? s.f[a] : npos[a];
                npos[b] = swap ? s.f[b] : npos[b];
            });
        });
        // three or more agents: whoever moved onto a cell somebody was left standing on goes back
        if constexpr (A > 2) {
            static_for<0, A>([&](auto) {
                static_for<0, A>([&](auto AI) {
                    constexpr int a = AI.value;
                    bool clash = false;
                    static_for<0, A>([&](auto BI) { if (BI.value != a) clash |= npos[BI.value] == npos[a]; });
                    npos[a] = (clash && npos[a] != s.f[a]) ? s.f[a] : npos[a];
                });
            });
        }
        static_for<0, A>([&](auto K) { s.f[K.value] = npos[K.value]; });
    }

    // ---- interact (agent-id order) and reward conditions (previous state) ----
    // A PickupDrop agent neither moved nor turned, so the neighbour words loaded above are
    // still those of its cell and its front cell is also its previous front cell.  What happens
    // is a pure function of a handful of small codes -- front cell type, what lies there, what
    // the agent holds, the pot's condition, whether another agent stands there -- so the rule
    // chain is evaluated once on the host into two byte tables (csrc/capi.hip) and each agent's
    // interaction here is one LDS lookup.  Codes that do not apply to a cell type (the pot
    // condition of a counter, ...) index identical entries, so they need no masking.
    onion_mask = dish_mask = deliver_mask = 0;
    static_for<0, A>([&](auto AI) {
        constexpr int a = AI.value;
        const bool on = act[a] == ACT_PICKUP_DROP;
        const uint32_t w = sel4(nb[a], s.dir[a]);
        const uint32_t ct = nb_type(w), slot = nb_slot(w), p = nb_pot(w), nf = nb_fidx(w);
        const uint32_t lying = item_get<NIW>(s.items, slot);
        const uint32_t inv = s.inv[a];

        const uint32_t rent = T.rew_entry(ct | (prev_inv[a] << 3) | (pot_state(prev_pots, p) << 5));
        onion_mask |= on ? ((rent & 1u) << a) : 0u;
        dish_mask |= on ? (((rent >> 1) & 1u) << a) : 0u;
        deliver_mask |= on ? (((rent >> 2) & 1u) << a) : 0u;

        uint32_t ahead = 0;
        static_for<0, A>([&](auto BI) { if (BI.value != a) ahead |= (s.f[BI.value] == nf) ? 1u : 0u; });

        const uint32_t looked = T.int_entry(ct | (lying << 3) | (inv << 5) | (pot_state(s.pots, p) << 7) | (ahead << 9));
        const uint32_t ent = on ? looked : inv;  // new holding | item op << 2 | pot op << 4
        const uint32_t iop = (ent >> 2) & 3u, pop = (ent >> 4) & 3u;
        s.inv[a] = ent & 3u;
        item_set<NIW>(s.items, slot, iop == INT_ITEM_CLEAR ? (uint32_t)NONE : inv, iop != INT_ITEM_KEEP);
        const uint32_t pbyte = 0xFFu << (8u * p);
        s.pots = pop == INT_POT_RESET ? ((s.pots & ~pbyte) | ((uint32_t)(COOK_TIME << 2) << (8u * p))) : s.pots;
        s.pots += pop == INT_POT_ADD ? (1u << (8u * p)) : 0u;
    });
    return ((onion_mask >> i) & 1u) | (((dish_mask >> i) & 1u) << 1) | ((uint32_t)__builtin_popcount(deliver_mask) << 2);
}

// ---------------------------------------------------------------------------------------
// one env step for three or four agents.  The env's four lanes (a quad) share the work: lane i
// looks up its own agent's cell, target and interaction, the quad trades the results through
// DPP moves, and every lane applies them to its copy of the env in the same order, so all copies
// stay identical.  Same rules and same results as step_env; about 40 % fewer instructions than
// every lane working through every agent.  Returns the reward-table index of agent i.
// ---------------------------------------------------------------------------------------
// value held by lane K of the caller's quad (a switched-off lane reads as zero)
template <int K>
__device__ __forceinline__ uint32_t quad_bcast(uint32_t x) {
    return (uint32_t)__builtin_amdgcn_mov_dpp((int)x, K * 0x55, 0xF, 0xF, true);
}
// OR / sum over the lanes of the quad, as a two-step butterfly: all four lanes must be running,
// partial results pass through each of them
__device__ __forceinline__ uint32_t quad_or(uint32_t x) {
    x |= (uint32_t)__builtin_amdgcn_mov_dpp((int)x, 0xB1, 0xF, 0xF, true);  // quad_perm [1, 0, 3, 2]
    x |= (uint32_t)__builtin_amdgcn_mov_dpp((int)x, 0x4E, 0xF, 0xF, true);  // quad_perm [2, 3, 0, 1]
    return x;
}
__device__ __forceinline__ uint32_t quad_sum(uint32_t x) {
    x += (uint32_t)__builtin_amdgcn_mov_dpp((int)x, 0xB1, 0xF, 0xF, true);
    x += (uint32_t)__builtin_amdgcn_mov_dpp((int)x, 0x4E, 0xF, 0xF, true);
    return x;
}

template <int A, int NIW>
__device__ __forceinline__ uint32_t step_env_quad(EnvState<A, NIW>& s, const Lds& T, const EnvParams& P,
                                                  uint32_t my_act, const uint8_t* prio, int e, uint32_t i) {
    static_assert(A > 2 && lanes_per_env(A) == 4, "an env of three or four agents takes one quad");
    // With three agents the quad's last lane (i == 3) runs along without an agent: nothing it
    // computes is read by the others except through the quad-wide OR / sum, where it adds nothing.
    const bool real = i < (uint32_t)A;
    const uint32_t my_f = pick<A>(s.f, i), my_inv0 = pick<A>(s.inv, i);
    const uint4 nbm = T.nb(my_f);  // the four neighbour words of my cell
    const uint32_t prev_pots = s.pots;

    s.t += 1;
    // pots that are full keep cooking
    s.pots = tick_pots(s.pots);

    // ---- movement: my own turn and target, then everybody's ----
    uint32_t want[A];
    {
        const bool is_move = my_act < 4u;
        const uint32_t d = (0x0213u >> (4u * (my_act & 3u))) & 3u;  // up,down,left,right -> 3,1,2,0
        const uint32_t my_dir = is_move ? d : pick<A>(s.dir, i);     // a move action always turns the agent
        const uint32_t w = sel4(nbm, d);
        const bool open = nb_type(w) == FLOOR && item_get<NIW>(s.items, nb_slot(w)) == NONE;
        const uint32_t mine = ((is_move && open) ? nb_fidx(w) : my_f) | (my_dir << 8);
        static_for<0, A>([&](auto K) {
            const uint32_t q = quad_bcast<K.value>(mine);
            want[K.value] = q & 0xFFu;
            s.dir[K.value] = q >> 8;
        });
    }
    // the order in which movers are served (every lane draws the same numbers)
    uint32_t movers = 0;  // packed 4-bit agent ids
    uint32_t m = 0;
    static_for<0, A>([&](auto AI) {
        constexpr int a = AI.value;
        const bool stays = want[a] == s.f[a];
        movers |= stays ? 0u : ((uint32_t)a << (4u * m));
        m += stays ? 0u : 1u;
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
            constexpr int k = A - K.value;  // A-1 .. 1
            const bool on = (uint32_t)k < m;
            const uint32_t j = s.rng.below(k + 1, on);
            const uint32_t x = on ? (((movers >> (4 * k)) ^ (movers >> (4 * j))) & 15u) : 0u;
            movers ^= (x << (4 * k)) | (x << (4 * j));
        });
    }
    // first claim wins: a lane follows its own agent only and asks the quad whether the target
    // of the agent being served is held by somebody whose place is settled
    const uint32_t my_want = pick<A>(want, i);
    uint32_t my_npos = my_f;
    bool my_decided = my_want == my_f;  // stayers keep their cell unconditionally
    static_for<0, A>([&](auto K) {
        const bool on = (uint32_t)K.value < m;
        const uint32_t a = (movers >> (4 * K.value)) & 15u;
        const uint32_t wa = pick<A>(want, a);
        const bool taken = quad_or((real && my_decided && my_npos == wa) ? 1u : 0u) != 0u;
        const bool me = on && a == i;
        my_npos = (me && !taken) ? wa : my_npos;
        my_decided = my_decided || me;
    });
    uint32_t npos[A];
    static_for<0, A>([&](auto K) { npos[K.value] = quad_bcast<K.value>(my_npos); });
    // no passing through each other.  At most one agent stands where mine is going, so the
    // pairs can be looked at independently of each other.
    {
        bool swap = false;
        static_for<0, A>([&](auto B) { swap = swap || (my_npos == s.f[B.value] && npos[B.value] == my_f); });
        my_npos = (swap && my_npos != my_f) ? my_f : my_npos;
    }
    // whoever moved onto a cell somebody was left standing on goes back; the retreat can expose
    // the next overlap, a chain of at most A-1 retreats behind one agent that stood still
    static_for<0, A - 1>([&](auto) {
        static_for<0, A>([&](auto K) { npos[K.value] = quad_bcast<K.value>(my_npos); });
        uint32_t same = 0;  // agents on my cell, myself included
        static_for<0, A>([&](auto B) { same += (npos[B.value] == my_npos) ? 1u : 0u; });
        my_npos = (same > 1u && my_npos != my_f) ? my_f : my_npos;
    });
    static_for<0, A>([&](auto K) { s.f[K.value] = quad_bcast<K.value>(my_npos); });

    // ---- interact (agent-id order) and reward conditions (previous state) ----
    // A PickupDrop agent neither moved nor turned, so nbm still describes its surroundings.  Each
    // lane condenses its own agent's side of the interaction into one word (front cell type,
    // counter slot, pot, whether another agent stands in front, whether it interacts at all);
    // what depends on the running state (what lies there, holdings, pot condition) is looked up
    // by every lane for every agent in turn.
    const bool on_me = my_act == (uint32_t)ACT_PICKUP_DROP;
    uint32_t my_desc;
    uint32_t rew_bits;
    {
        const uint32_t w = sel4(nbm, pick<A>(s.dir, i));
        const uint32_t ct = nb_type(w), slot = nb_slot(w), p = nb_pot(w), nf = nb_fidx(w);
        const uint32_t rent = T.rew_entry(ct | (my_inv0 << 3) | (pot_state(prev_pots, p) << 5));
        rew_bits = (on_me && real) ? rent : 0u;  // onion in pot | soup plated << 1 | delivery << 2
        uint32_t ahead = 0;
        static_for<0, A>([&](auto B) { ahead |= (s.f[B.value] == nf && (uint32_t)B.value != i) ? 1u : 0u; });
        // ct and ahead sit where the decision table's index wants them
        my_desc = ct | (ahead << 9) | (slot << 10) | (p << 17) | ((on_me ? 1u : 0u) << 19);
    }
    static_for<0, A>([&](auto AI) {
        constexpr int a = AI.value;
        const uint32_t dsc = quad_bcast<a>(my_desc);
        const bool on = (dsc >> 19) & 1u;
        const uint32_t slot = (dsc >> 10) & 127u, p = (dsc >> 17) & 3u;
        const uint32_t lying = item_get<NIW>(s.items, slot);
        const uint32_t inv = s.inv[a];
        const uint32_t looked = T.int_entry((dsc & 0x207u) | (lying << 3) | (inv << 5) | (pot_state(s.pots, p) << 7));
        const uint32_t ent = on ? looked : inv;  // new holding | item op << 2 | pot op << 4
        const uint32_t iop = (ent >> 2) & 3u, pop = (ent >> 4) & 3u;
        s.inv[a] = ent & 3u;
        item_set<NIW>(s.items, slot, iop == INT_ITEM_CLEAR ? (uint32_t)NONE : inv, iop != INT_ITEM_KEEP);
        const uint32_t pbyte = 0xFFu << (8u * p);
        s.pots = pop == INT_POT_RESET ? ((s.pots & ~pbyte) | ((uint32_t)(COOK_TIME << 2) << (8u * p))) : s.pots;
        s.pots += pop == INT_POT_ADD ? (1u << (8u * p)) : 0u;
    });
    return (rew_bits & 3u) | (quad_sum((rew_bits >> 2) & 1u) << 2);
}

// ---------------------------------------------------------------------------------------
// feature block of agent i, packed int8 in NW dwords
// ---------------------------------------------------------------------------------------
template <int NW>
struct Block {
    uint32_t w[NW];
    // one byte at compile-time offset OFF
    template <int OFF>
    __device__ __forceinline__ void put(int v) {
        static_assert(OFF / 4 < NW, "offset outside block");
        w[OFF / 4] |= ((uint32_t)v & 0xFFu) << ((OFF % 4) * 8);
    }
    // four bytes (a dword whose byte k lands at OFF + k) at compile-time offset OFF
    template <int OFF>
    __device__ __forceinline__ void put4(uint32_t v) {
        static_assert(OFF / 4 < NW, "offset outside block");
        constexpr int sh = (OFF % 4) * 8;
        w[OFF / 4] |= v << sh;
        if constexpr (sh != 0 && OFF / 4 + 1 < NW) w[OFF / 4 + 1] |= v >> (32 - sh);
    }
};

// sorted (ascending) top-4 of 32-bit keys.
// Keys are (distance << 23) | (slot << 16) | (dx & 0xFF) << 8 | (dy & 0xFF): ordered by distance,
// then by slot (= row-major cell order); the low 16 bits are the two feature bytes themselves.
// An empty place holds TOP4_NONE: above every real key (distances stay below 2^9 - 1, so a real
// key is below 0xFF800000) and with a zero payload, so emitting needs no "is it real" test.
constexpr uint32_t TOP4_NONE = 0xFFFF0000u;
struct Top4 {
    uint32_t k0, k1, k2, k3;
    __device__ __forceinline__ void init() { k0 = k1 = k2 = k3 = TOP4_NONE; }
    __device__ __forceinline__ void insert(uint32_t x) {
        uint32_t lo;
        lo = min(k0, x); x = max(k0, x); k0 = lo;
        lo = min(k1, x); x = max(k1, x); k1 = lo;
        lo = min(k2, x); x = max(k2, x); k2 = lo;
        k3 = min(k3, x);
    }
};

template <int OFF, int NW>
__device__ __forceinline__ void emit_top4(Block<NW>& blk, const Top4& t) {
    blk.template put4<OFF>((t.k0 & 0xFFFFu) | (t.k1 << 16));
    blk.template put4<OFF + 4>((t.k2 & 0xFFFFu) | (t.k3 << 16));
}

template <int OFF, int NW>
__device__ __forceinline__ void emit_pot(Block<NW>& blk, uint32_t p, uint32_t pots) {
    const bool have = p != 0xFFu;
    const uint32_t pp = p & 3u;
    const uint32_t cls = pot_class(pots, pp);
    const uint32_t cnt = pot_count(pots, pp);
    const uint32_t onehot = 1u << (8u * ((POT_STATUS_INDEX_PACKED >> (4 * cls)) & 3u));
    const uint32_t timer = cnt == (uint32_t)POT_CAPACITY ? pot_timer(pots, pp) : 0xFFu;  // -1 when not cooking
    blk.template put4<OFF + 1>(have ? onehot : 0u);
    blk.template put4<OFF + 5>(have ? (cnt | (timer << 8)) : 0u);
}

template <int A, int FEAT, int NIW>
__device__ __forceinline__ void build_block(Block<FeatGeom<A, FEAT>::NW>& blk, const EnvState<A, NIW>& s,
                                            const Lds& T, const EnvParams& P, uint32_t i) {
    using G = FeatGeom<A, FEAT>;
    constexpr int NW = G::NW, NWP = G::NWP;
    const uint32_t myf = pick<A>(s.f, i);
    const uint32_t mydir = pick<A>(s.dir, i);
    const uint32_t myinv = pick<A>(s.inv, i);

    // static part of the block for an agent standing on this floor cell, 16 bytes at a time
    const uint4* tmpl = reinterpret_cast<const uint4*>(T.b + P.off_postab) + myf * (NWP / 4);
    static_for<0, NWP / 4>([&](auto K) {
        constexpr int k = K.value;
        const uint4 v = tmpl[k];
        blk.w[4 * k] = v.x;
        if constexpr (4 * k + 1 < NW) blk.w[4 * k + 1] = v.y;
        if constexpr (4 * k + 2 < NW) blk.w[4 * k + 2] = v.z;
        if constexpr (4 * k + 3 < NW) blk.w[4 * k + 3] = v.w;
    });

    blk.template put4<G::O_DIR>(1u << (8u * mydir));
    if constexpr (G::kSpatial) {
        blk.template put4<G::O_INV>(myinv != NONE ? 1u << (8u * ((INVENTORY_INDEX_PACKED >> (4 * myinv)) & 3u)) : 0u);

        const uint32_t rc = T.frc(myf);
        const int ar = (int)(rc >> 8), ac = (int)(rc & 0xFFu);

        // NextToPot (4 status rows x 4 adjacency columns; adjacency order right, left, down, up,
        // pinned by the reference's golden vector).  Which neighbours are which pot is static:
        // the table holds, per floor cell and pot, the four adjacency bytes; a pot's condition
        // only picks the row they go to.
        const uint4 pa = T.pot_adj(myf);
        uint32_t ntp[4] = {0u, 0u, 0u, 0u};
        static_for<0, MAX_POTS>([&](auto PI) {
            constexpr int p = PI.value;
            if (p < P.NP) {
                static_assert(NEXT_TO_POT_ROW_PACKED == 0x3210, "row of a class is the class code itself");
                const uint32_t row = pot_class(s.pots, p);
                const uint32_t m = p == 0 ? pa.x : p == 1 ? pa.y : p == 2 ? pa.z : pa.w;
                static_for<0, 4>([&](auto R) { ntp[R.value] |= row == (uint32_t)R.value ? m : 0u; });
            }
        });
        blk.template put4<G::O_NEXT_POT + 0>(ntp[0]);
        blk.template put4<G::O_NEXT_POT + 4>(ntp[1]);
        blk.template put4<G::O_NEXT_POT + 8>(ntp[2]);
        blk.template put4<G::O_NEXT_POT + 12>(ntp[3]);

        // CanMoveDirection: the neighbour is floor (static byte mask) and nothing lies on it
        const uint4 ad = T.adj_info(myf);  // x = floor mask, y = item slot of each neighbour
        uint32_t occ[NIW];
        static_for<0, NIW>([&](auto K) { occ[K.value] = (s.items[K.value] | (s.items[K.value] >> 1)) & 0x55555555u; });
        uint32_t busy = 0;
        static_for<0, 4>([&](auto K) {
            constexpr int k = K.value;
            const uint32_t sl = (ad.y >> (8 * k)) & 0xFFu;
            const uint32_t word = pick<NIW>(occ, sl >> 4);
            busy |= ((word >> ((sl & 15u) * 2u)) & 1u) << (8 * k);
        });
        blk.template put4<G::O_CAN_MOVE>(ad.x & ~busy);

        if constexpr (G::kOthers && A == 2) {
            const uint32_t orc = T.frc(i ? s.f[0] : s.f[1]);
            blk.template put4<G::O_OTHERS>(((uint32_t)(ar - (int)(orc >> 8)) & 0xFFu) |
                                           (((uint32_t)(ac - (int)(orc & 0xFFu)) & 0xFFu) << 8));
        } else if constexpr (G::kOthers && A > 2) {
            // (self - other) for the other agents in id order: pair j sits at byte 2j
            uint32_t lo = 0, hi = 0;
            static_for<0, A>([&](auto BI) {
                constexpr uint32_t b = BI.value;
                const uint32_t orc = T.frc(s.f[b]);
                const uint32_t pair = ((uint32_t)(ar - (int)(orc >> 8)) & 0xFFu) | (((uint32_t)(ac - (int)(orc & 0xFFu)) & 0xFFu) << 8);
                const uint32_t j = b < i ? b : b - 1u;  // unused when b == i
                const bool other = b != i;
                lo |= (other && j == 0u) ? pair : (other && j == 1u) ? (pair << 16) : 0u;
                hi |= (other && j == 2u) ? pair : 0u;
            });
            blk.template put4<G::O_OTHERS>(lo);
            if constexpr (A > 3) blk.template put4<G::O_OTHERS + 4>(hi);
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
                    const uint32_t slot = K.value * 16 + (bit >> 1);
                    const uint32_t orc = T.slotrc(slot);
                    const int dy = ar - (int)(orc >> 8), dx = ac - (int)(orc & 0xFFu);
                    const uint32_t d = P.dist_metric ? (uint32_t)(dy * dy + dx * dx) : (uint32_t)(abs(dy) + abs(dx));
                    const uint32_t key = (d << 23) | (slot << 16) | (((uint32_t)dx & 0xFFu) << 8) | ((uint32_t)dy & 0xFFu);
                    onion.insert(kind == ONION ? key : TOP4_NONE);
                    plate.insert(kind == PLATE ? key : TOP4_NONE);
                    soup.insert(kind == SOUP ? key : TOP4_NONE);
                }
            });
            emit_top4<G::O_ONION>(blk, onion);
            emit_top4<G::O_PLATE>(blk, plate);
            emit_top4<G::O_SOUP>(blk, soup);

            // status / contents / timer of the two nearest pots (their static fields are in the template)
            const uint32_t po = T.pot_order(myf);
            emit_pot<G::O_POT0>(blk, po & 0xFFu, s.pots);
            emit_pot<G::O_POT1>(blk, po >> 8, s.pots);
        }
    }
}

// Copy a BLK-byte block (NW dwords in registers) to an arbitrary byte address in LDS.
// With sh = distance of dst above the aligned dword BELOW it counted as 1..4 (4 when dst is
// itself aligned), aligned dword k >= 1 of that frame holds block bytes [4k - sh, 4k - sh + 4):
// one funnel shift of the register pair (w[k], w[k-1]), the same instruction for every sh.
// Dwords 1..KFULL lie inside the block for every sh and are stored unconditionally; the few
// head and tail bytes that some alignment leaves uncovered are stored as single bytes, also
// unconditionally (a byte written twice gets the same value twice).  (gfx950 LDS also takes
// dword and wider stores at odd addresses, which would make this a plain copy, but measured
// slower: 20.5 us per launch with wide unaligned stores, 28.5 us with unaligned dwords,
// against 19.4 us for this form at the time.)
template <int BLK, int NW>
__device__ __forceinline__ void write_block(uint8_t* img, uint32_t dst, const uint32_t (&w)[NW]) {
    if constexpr (BLK % 4 == 0) {
        // callers only produce multiples of BLK here: dword copies
        uint32_t* base = reinterpret_cast<uint32_t*>(img + dst);
        static_for<0, NW>([&](auto K) { base[K.value] = w[K.value]; });
    } else {
        constexpr int KFULL = (BLK + 1) / 4 - 1;
        static_assert(KFULL >= 1 && KFULL < NW, "block too small for the funnel copy");
        const uint32_t sh = ((dst + 3u) & 3u) + 1u;
        uint32_t* base = reinterpret_cast<uint32_t*>(img + (dst - sh));
        const uint32_t rot = (32u - 8u * sh) & 31u;
        static_for<1, KFULL + 1>([&](auto K) {
            constexpr int k = K.value;
            base[k] = __builtin_amdgcn_alignbit(w[k], w[k - 1], rot);
        });
        uint8_t* bp = img + dst;
        static_for<0, 3>([&](auto J) { bp[J.value] = (uint8_t)(w[0] >> (8 * J.value)); });
        static_for<4 * KFULL, BLK>([&](auto J) {
            constexpr int j = J.value;
            bp[j] = (uint8_t)(w[j / 4] >> (8 * (j % 4)));
        });
    }
}

// Two agents, global_obs: a row is [own block | partner's block].  The lane takes the partner's
// block from the neighbouring lane with 26 cross-lane moves, appends it to its own with
// constant funnel shifts (BLK % 4 == 1), and stores its whole row once: even lanes' rows start
// dword-aligned, odd lanes' rows two bytes later, so one funnel by 0 or 16 bits serves both and
// the two bytes the dwords leave over (the row's last two or first two) go out as one
// halfword.  Versus copying the block into both rows of the env this trades a few more VALU
// instructions for 27 instead of 68 LDS stores per lane (18.68 -> 18.55 us per launch).
template <int BLK, int NW>
__device__ __forceinline__ void write_row_pair(uint8_t* img, uint32_t lane, const uint32_t (&w)[NW]) {
    static_assert(BLK % 4 == 1 && NW == BLK / 4 + 1, "block must end one byte into its last dword");
    constexpr int KB = BLK / 4;            // whole dwords in a block
    constexpr int NR = (2 * BLK + 3) / 4;  // dwords of a row, the last one half used
    uint32_t p[NW];
    static_for<0, NW>([&](auto K) {
        // quad_perm [1, 0, 3, 2]: swap with the other lane of the pair
        p[K.value] = (uint32_t)__builtin_amdgcn_mov_dpp((int)w[K.value], 0xB1, 0xF, 0xF, true);
    });
    uint32_t R[NR];
    static_for<0, KB>([&](auto K) { R[K.value] = w[K.value]; });
    R[KB] = (w[KB] & 0xFFu) | (p[0] << 8);
    static_for<1, NW>([&](auto K) { R[KB + K.value] = __builtin_amdgcn_alignbit(p[K.value], p[K.value - 1], 24); });
    const uint32_t odd = lane & 1u;
    const uint32_t rot = odd * 16u;
    uint8_t* row = img + lane * (uint32_t)(2 * BLK);
    uint32_t* base = reinterpret_cast<uint32_t*>(row + odd * 2u);
    static_for<0, NR - 1>([&](auto M) {
        constexpr int m = M.value;
        base[m] = __builtin_amdgcn_alignbit(R[m + 1], R[m], rot);
    });
    uint16_t* half = reinterpret_cast<uint16_t*>(row + (odd ? 0u : (uint32_t)(4 * (NR - 1))));
    *half = (uint16_t)(odd ? R[0] : R[NR - 1]);
}

// ---------------------------------------------------------------------------------------
// the kernel.  mode: 0 = step (+ auto-reset on truncation), 1 = reset every env,
// 2 = featurize the stored state only.  Bits 8+ of mode are a profiling aid.
// ---------------------------------------------------------------------------------------
#define OC_KERNEL_NAME env_kernel
#define OC_MANY 0
#include "env_kernel_body.inc"
#undef OC_KERNEL_NAME
#undef OC_MANY
#define OC_KERNEL_NAME env_kernel_many
#define OC_MANY 1
#include "env_kernel_body.inc"
#undef OC_KERNEL_NAME
#undef OC_MANY

// ---------------------------------------------------------------------------------------
// dispatch
// ---------------------------------------------------------------------------------------
template <int A, int FEAT, int NIW>
static hipError_t launch_one(const EnvParams& P, const uint8_t* tab, const uint8_t* aux, uint32_t* state, const void* actions,
                             int act_size, const uint8_t* prio, const uint8_t* spawn_ov, const uint8_t* dir_ov,
                             float* obs, float* rewards, uint8_t* term, uint8_t* trunc, int mode,
                             hipStream_t stream) {
    using G = FeatGeom<A, FEAT>;
    // 4 waves per workgroup (the kernel is bounded to 256 threads; the lookup blob is staged once per
    // workgroup) unless their row images would not fit 64 KiB of LDS: a CU's 160 KiB then hold two
    // such workgroups, and narrower ones of the wide-row cases fill it equally well.
    static const int waves_req = [] { const char* v = getenv("MARL_AMD_WAVES"); return v ? atoi(v) : 4; }();  // tuning aid
    int waves = (waves_req == 1 || waves_req == 2 || waves_req == 4) ? waves_req : 4;
    constexpr int RPW = 64 / lanes_per_env(A) * A;  // rows per wave
    constexpr int IMG = RPW * G::D;                 // bytes of a wave's row image
    static const size_t wg_cap = [] { const char* v = getenv("MARL_AMD_LDS_CAP_KB"); return (size_t)(v ? atoi(v) : 64) * 1024; }();  // tuning aid
    while (waves > 1 && (size_t)P.tab_bytes + (size_t)waves * IMG > wg_cap) waves >>= 1;
    const size_t lds = (size_t)P.tab_bytes + (size_t)waves * IMG;
    // dynamic-LDS limit already granted to this instantiation's kernels, per device; host threads
    // that step different handles of one family at the same time meet here
    static size_t configured[64] = {0};
    static std::mutex configured_mutex;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return hipErrorInvalidDevice;
    std::unique_lock<std::mutex> configured_lock(configured_mutex);
    if (lds > configured[dev]) {
        hipError_t err = hipFuncSetAttribute(reinterpret_cast<const void*>(&env_kernel<A, FEAT, NIW>),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (err == hipSuccess)
            err = hipFuncSetAttribute(reinterpret_cast<const void*>(&env_kernel_many<A, FEAT, NIW>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (err != hipSuccess) return err;
        configured[dev] = lds;
    }
    configured_lock.unlock();
    const size_t N = (size_t)P.E * A;
    const size_t rows_per_block = (size_t)waves * RPW;
    const unsigned grid = (unsigned)((N + rows_per_block - 1) / rows_per_block);
    if (P.nsteps > 1 && (mode & 0xFF) == 0)
        hipLaunchKernelGGL((env_kernel_many<A, FEAT, NIW>), dim3(grid), dim3(waves * 64), lds, stream, P, tab, aux, state,
                           actions, act_size, prio, spawn_ov, dir_ov, obs, rewards, term, trunc, mode);
    else
        hipLaunchKernelGGL((env_kernel<A, FEAT, NIW>), dim3(grid), dim3(waves * 64), lds, stream, P, tab, aux, state,
                           actions, act_size, prio, spawn_ov, dir_ov, obs, rewards, term, trunc, mode);
    return hipGetLastError();
}

template <int A, int FEAT>
static hipError_t launch_niw(int niw, const EnvParams& P, const uint8_t* tab, const uint8_t* aux, uint32_t* state,
                             const void* actions, int act_size, const uint8_t* prio, const uint8_t* spawn_ov,
                             const uint8_t* dir_ov, float* obs, float* rewards, uint8_t* term, uint8_t* trunc,
                             int mode, hipStream_t stream) {
#define OC_GO(NIW_) return launch_one<A, FEAT, NIW_>(P, tab, aux, state, actions, act_size, prio, spawn_ov, dir_ov, obs, rewards, term, trunc, mode, stream)
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
hipError_t launch_env_kernel_A(int feat, int niw, const EnvParams& P, const uint8_t* tab, const uint8_t* aux,
                               uint32_t* state,
                               const void* actions, int act_size, const uint8_t* prio, const uint8_t* spawn_ov,
                               const uint8_t* dir_ov, float* obs, float* rewards, uint8_t* term, uint8_t* trunc,
                               int mode, hipStream_t stream) {
#define OC_GO(FEAT_) return launch_niw<A, FEAT_>(niw, P, tab, aux, state, actions, act_size, prio, spawn_ov, dir_ov, obs, rewards, term, trunc, mode, stream)
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
template hipError_t launch_env_kernel_A<OC_AGENTS>(int, int, const EnvParams&, const uint8_t*, const uint8_t*, uint32_t*, const void*,
                                                   int, const uint8_t*, const uint8_t*, const uint8_t*, float*, float*,
                                                   uint8_t*, uint8_t*, int, hipStream_t);

}  // namespace oc
