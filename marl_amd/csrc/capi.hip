// Host side of libmarl_amd.so: the extern "C" entry points of include/marl_amd.h, the
// per-layout lookup-table builder, and the small pack/unpack kernels behind
// marl_env_export_state / marl_env_import_state.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <string>
#include <vector>

#include "../../include/marl_amd.h"
#include "oc_common.h"

namespace oc {
template <int A>
hipError_t launch_env_kernel_A(int feat, int niw, const EnvParams& P, const uint8_t* tab, const uint8_t* aux,
                               uint32_t* state, const void* actions, int act_size, const uint8_t* prio, const uint8_t* spawn_ov,
                               const uint8_t* dir_ov, float* obs, float* rewards, uint8_t* term, uint8_t* trunc,
                               int mode, hipStream_t stream);
extern template hipError_t launch_env_kernel_A<1>(int, int, const EnvParams&, const uint8_t*, const uint8_t*, uint32_t*, const void*, int,
                                                  const uint8_t*, const uint8_t*, const uint8_t*, float*, float*, uint8_t*,
                                                  uint8_t*, int, hipStream_t);
extern template hipError_t launch_env_kernel_A<2>(int, int, const EnvParams&, const uint8_t*, const uint8_t*, uint32_t*, const void*, int,
                                                  const uint8_t*, const uint8_t*, const uint8_t*, float*, float*, uint8_t*,
                                                  uint8_t*, int, hipStream_t);
extern template hipError_t launch_env_kernel_A<3>(int, int, const EnvParams&, const uint8_t*, const uint8_t*, uint32_t*, const void*, int,
                                                  const uint8_t*, const uint8_t*, const uint8_t*, float*, float*, uint8_t*,
                                                  uint8_t*, int, hipStream_t);
extern template hipError_t launch_env_kernel_A<4>(int, int, const EnvParams&, const uint8_t*, const uint8_t*, uint32_t*, const void*, int,
                                                  const uint8_t*, const uint8_t*, const uint8_t*, float*, float*, uint8_t*,
                                                  uint8_t*, int, hipStream_t);
hipError_t launch_gae(const float* rewards, const float* dones, const float* values, const float* next_values,
                      int T, int N, float gamma, float gl, float* adv, float* ret, int mode, const float* rmean,
                      hipStream_t stream);
hipError_t launch_row_sums(const float* x, int T, int N, double* partial, double* sums, float* mean_out,
                           hipStream_t stream);
int row_sum_chunks(int N);
int reward_event_chunks(int N);
hipError_t launch_sample_categorical(const float* logits, int N, int K, int ld, uint32_t key, const uint64_t* counter,
                                     uint64_t row_base, long long* actions, float* logprob, float* entropy,
                                     float* next_col, hipStream_t stream);
int ppo_loss_chunks(int M);
hipError_t launch_ppo_loss(const float* logits, int M, int K, int ld, const float* values, const long long* index,
                           const float* actions_f, const float* old_logprob, const float* advantages,
                           const float* value_target, float clip, float entropy_coef, float value_coef,
                           float* grad_logits, float* grad_values, double* stats, double* partial, hipStream_t stream);
hipError_t launch_reward_events(const float* rewards, int E, int A, uint64_t* partial, uint64_t* totals,
                                double* first_rows, hipStream_t stream);
hipError_t launch_reward_events_many(const float* rewards, int K, int E, int A, uint64_t* partial, uint64_t* totals,
                                     double* first_rows, hipStream_t stream);
}  // namespace oc

using namespace oc;

static thread_local std::string g_err;
static int fail(int code, const std::string& msg) {
    g_err = msg;
    return code;
}
namespace oc {
// for the entry points that live next to their kernels (csrc/stats_kernels.hip)
int set_error(int code, const std::string& msg) { return fail(code, msg); }
}  // namespace oc
static int fail_hip(hipError_t e, const char* what) {
    g_err = std::string(what) + ": " + hipGetErrorString(e);
    return MARL_ERR_HIP;
}

struct marl_env {
    EnvParams P;
    int A, feat, niw, H, W, ncell, NP, D, BLK, words;
    int device;                // HIP device that holds the tables; every call must run with it current
    int stream_nt_forced;      // MARL_AMD_STREAM_NT as read when the handle was made: 0 / 1, -1 = not set
    std::vector<uint8_t> tab;  // host copy of the hot lookup blob
    std::vector<uint8_t> aux;  // host copy of the cell-indexed tables of the snapshot kernels
    uint8_t* d_tab;
    uint8_t* d_aux;
};

// ---------------------------------------------------------------------------------------
// lookup tables
// ---------------------------------------------------------------------------------------
static uint32_t mix32_host(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352du;
    x ^= x >> 15; x *= 0x846ca68bu;
    x ^= x >> 16;
    return x;
}

static int dist_of(int metric, int dy, int dx) {
    return metric ? dy * dy + dx * dx : std::abs(dy) + std::abs(dx);
}

static bool by_distance(const std::pair<int, int>& a, const std::pair<int, int>& b) { return a.first < b.first; }

// (agent - object) for the n nearest cells of one static type; stable on ties => scan order
static void static_closest(const std::vector<uint8_t>& type, int W, int ncell, uint32_t kind, int n, int ar, int ac,
                           int metric, int8_t* out) {
    std::vector<std::pair<int, int>> cand;  // (distance, cell)
    for (int c = 0; c < ncell; ++c)
        if (type[c] == kind) cand.push_back({dist_of(metric, ar - c / W, ac - c % W), c});
    std::stable_sort(cand.begin(), cand.end(), by_distance);
    for (int k = 0; k < n && k < (int)cand.size(); ++k) {
        out[2 * k] = (int8_t)(ar - cand[k].second / W);
        out[2 * k + 1] = (int8_t)(ac - cand[k].second % W);
    }
}

static int align16(int x) { return (x + 15) & ~15; }

// The PickupDrop rule chain, evaluated for every combination of the codes the kernel can meet:
// pick up what lies loose on the floor / take from a stack / take from a counter / plate a
// finished soup, otherwise, with something in hand: put it on an empty floor cell nobody
// stands on, on an empty counter, an onion into a pot with room, a soup into the delivery
// zone.  Rewards look at the state before the step: onion into a pot with room (0.1),
// plate at a finished pot (0.3), soup at the delivery zone (1.0, shared).
static void build_decision_tables(uint8_t* itab, uint8_t* rtab) {
    for (uint32_t idx = 0; idx < (uint32_t)INT_TABLE_BYTES; ++idx) {
        const uint32_t ct = idx & 7u, lying = (idx >> 3) & 3u, inv = (idx >> 5) & 3u, pst = (idx >> 7) & 3u;
        const bool ahead = (idx >> 9) & 1u;
        uint32_t ninv = inv, iop = INT_ITEM_KEEP, pop = INT_POT_KEEP;
        const bool holder = ct == FLOOR || ct == COUNTER;
        if (holder && lying != NONE && inv == NONE) { ninv = lying; iop = INT_ITEM_CLEAR; }
        else if (ct == ONION_STACK && inv == NONE) ninv = ONION;
        else if (ct == PLATE_STACK && inv == NONE) ninv = PLATE;
        else if (ct == POT && inv == PLATE && pst == POTST_READY) { ninv = SOUP; pop = INT_POT_RESET; }
        else if (inv != NONE) {
            if (ct == FLOOR && lying == NONE) { if (!ahead) { ninv = NONE; iop = INT_ITEM_PUT; } }
            else if (ct == COUNTER && lying == NONE) { ninv = NONE; iop = INT_ITEM_PUT; }
            else if (ct == POT && inv == ONION && pst == POTST_ROOM) { ninv = NONE; pop = INT_POT_ADD; }
            else if (ct == DELIVERY && inv == SOUP) ninv = NONE;
        }
        itab[idx] = (uint8_t)(ninv | (iop << 2) | (pop << 4));
    }
    for (uint32_t idx = 0; idx < (uint32_t)REW_TABLE_BYTES; ++idx) {
        const uint32_t ct = idx & 7u, inv = (idx >> 3) & 3u, pst = (idx >> 5) & 3u;
        uint32_t r = 0;
        if (ct == POT && inv == ONION && pst == POTST_ROOM) r |= 1u;
        if (ct == POT && inv == PLATE && pst == POTST_READY) r |= 2u;
        if (ct == DELIVERY && inv == SOUP) r |= 4u;
        rtab[idx] = (uint8_t)r;
    }
}

static int build_tables(marl_env* env, const char* const* rows, int height, int dist_metric) {
    const int H = height;
    if (H <= 0 || rows == nullptr || rows[0] == nullptr) return fail(MARL_ERR_LAYOUT, "layout has no rows");
    const int W = (int)strlen(rows[0]);
    if (W < 3 || H < 3) return fail(MARL_ERR_LAYOUT, "layout smaller than 3x3");
    if (H * W > MAX_CELLS) return fail(MARL_ERR_LAYOUT, "layout has more than 256 cells");
    // the kernel orders "closest" candidates by 9-bit distance keys
    if (dist_of(dist_metric, H - 1, W - 1) > 510)
        return fail(MARL_ERR_LAYOUT, "layout too elongated for the squared-distance ordering (distance must stay below 511)");
    const int ncell = H * W;
    std::vector<uint8_t> type(ncell, WALL);
    std::vector<int> spawn, floor_cells, pot_cells, slots;
    for (int r = 0; r < H; ++r) {
        if (rows[r] == nullptr || (int)strlen(rows[r]) != W) return fail(MARL_ERR_LAYOUT, "layout rows differ in width");
        for (int c = 0; c < W; ++c) {
            const int cell = r * W + c;
            const bool border = (r == 0 || c == 0 || r == H - 1 || c == W - 1);
            switch (rows[r][c]) {
                case '#': type[cell] = WALL; break;
                case ' ': type[cell] = FLOOR; break;
                case '+': type[cell] = FLOOR; spawn.push_back(cell); break;
                case 'C': type[cell] = COUNTER; break;
                case 'U': type[cell] = POT; pot_cells.push_back(cell); break;
                case 'O': type[cell] = ONION_STACK; break;
                case '=': type[cell] = PLATE_STACK; break;
                case '@': type[cell] = DELIVERY; break;
                default: return fail(MARL_ERR_LAYOUT, std::string("unknown layout character '") + rows[r][c] + "'");
            }
            if (border && type[cell] == FLOOR) return fail(MARL_ERR_LAYOUT, "floor on the border: the grid must be enclosed");
            if (type[cell] == FLOOR) floor_cells.push_back(cell);
            if (type[cell] == FLOOR || type[cell] == COUNTER) slots.push_back(cell);
        }
    }
    const int A = env->A;
    if ((int)pot_cells.size() > MAX_POTS) return fail(MARL_ERR_LAYOUT, "more than 4 pots");
    if ((int)floor_cells.size() < A) return fail(MARL_ERR_LAYOUT, "fewer floor cells than agents");
    if (!spawn.empty() && (int)spawn.size() < A) return fail(MARL_ERR_LAYOUT, "fewer '+' spawn cells than agents");
    if ((int)slots.size() > MAX_SLOTS) return fail(MARL_ERR_LAYOUT, "more than 126 item-capable cells");

    const int nfloor = (int)floor_cells.size(), nslots = (int)slots.size(), NP = (int)pot_cells.size();
    env->H = H; env->W = W; env->ncell = ncell; env->NP = NP;
    env->BLK = block_dim(A, env->feat);
    env->D = env->feat == FEAT_GLOBAL ? A * env->BLK : env->BLK;
    const int NW = (env->BLK + 3) / 4, NWP = (NW + 3) / 4 * 4;
    env->niw = item_words_padded(nslots);
    env->words = state_words(A, nslots);

    // cell-indexed helper tables (also the aux blob of the snapshot kernels)
    env->aux.assign(AUX_BYTES, 0xFF);
    uint8_t* aux = env->aux.data();
    for (int k = 0; k < nfloor; ++k) { aux[AUX_FLOOR_CELL + k] = (uint8_t)floor_cells[k]; aux[AUX_FIDX_OF_CELL + floor_cells[k]] = (uint8_t)k; }
    for (int k = 0; k < nslots; ++k) { aux[AUX_CELL_OF_SLOT + k] = (uint8_t)slots[k]; aux[AUX_SLOT_OF_CELL + slots[k]] = (uint8_t)k; }
    std::vector<int> pot_of_cell(ncell, -1);
    for (int k = 0; k < NP; ++k) pot_of_cell[pot_cells[k]] = k;

    EnvParams& P = env->P;
    P.NP = NP; P.nslots = nslots; P.nfloor = nfloor; P.nspawn = (int)spawn.size(); P.dist_metric = dist_metric;
    int off = 0;
    P.off_nb = off; off = align16(off + nfloor * 16);
    P.off_postab = off; off = align16(off + nfloor * NWP * 4);
    P.off_adj = off; off = align16(off + nfloor * 32);
    P.off_frc = off; off = align16(off + nfloor * 2);
    P.off_slotrc = off; off = align16(off + 128 * 2);  // full size: the "no slot" code may be looked up (result unused)
    P.off_potorder = off; off = align16(off + nfloor * 2);
    P.off_spawnf = off; off = align16(off + (int)spawn.size() + 1);
    P.off_inttab = off; off = align16(off + INT_TABLE_BYTES);
    P.off_rewtab = off; off = align16(off + REW_TABLE_BYTES);
    P.off_rewval = off; off = align16(off + REW_VALUE_BYTES);
    P.tab_bytes = off;
    env->tab.assign((size_t)off, 0);
    uint8_t* t = env->tab.data();

    build_decision_tables(t + P.off_inttab, t + P.off_rewtab);
    {
        // rewards accumulate as Python floats in the config's list order
        // [onion_in_pot 0.1, soup_in_dish 0.3, delivery 1.0 shared] (overcooked_config.py:96),
        // then become float32 (agent_environment.py:63)
        float* rv = reinterpret_cast<float*>(t + P.off_rewval);
        for (int k = 0; k <= MAX_AGENTS; ++k)
            for (int own = 0; own < 4; ++own) {
                double r = 0.0;
                r += (own & 1) ? 0.1 : 0.0;
                r += (own & 2) ? 0.3 : 0.0;
                for (int j = 0; j < k; ++j) r += 1.0;
                rv[k * 4 + own] = (float)r;
            }
    }
    uint32_t* nb = reinterpret_cast<uint32_t*>(t + P.off_nb);
    uint16_t* frc = reinterpret_cast<uint16_t*>(t + P.off_frc);
    uint16_t* slotrc = reinterpret_cast<uint16_t*>(t + P.off_slotrc);
    for (int k = 0; k < nslots; ++k) slotrc[k] = (uint16_t)(((slots[k] / W) << 8) | (slots[k] % W));
    for (int k = 0; k < (int)spawn.size(); ++k) t[P.off_spawnf + k] = aux[AUX_FIDX_OF_CELL + spawn[k]];
    memset(t + P.off_potorder, 0xFF, (size_t)nfloor * 2);

    const bool full = env->feat == FEAT_GLOBAL || env->feat == FEAT_LOCAL;
    const bool others = full || env->feat == FEAT_MINIMAL_OTHER;
    const bool spatial = env->feat != FEAT_ONLY_DIRECTION;
    const int o_pos = (full ? 93 : 27) + (others ? 2 * (A - 1) : 0);
    static const int dir_dr[4] = {0, 1, 0, -1}, dir_dc[4] = {1, 0, -1, 0};  // right, down, left, up
    static const int adj_dr[4] = {0, 0, 1, -1}, adj_dc[4] = {1, -1, 0, 0};  // right, left, down, up
    for (int k = 0; k < nfloor; ++k) {
        const int cell = floor_cells[k], ar = cell / W, ac = cell % W;
        frc[k] = (uint16_t)((ar << 8) | ac);
        for (int d = 0; d < 4; ++d) {
            const int c = (ar + dir_dr[d]) * W + ac + dir_dc[d];
            const uint32_t slot = aux[AUX_SLOT_OF_CELL + c] == 0xFF ? NB_NO_SLOT : aux[AUX_SLOT_OF_CELL + c];
            const uint32_t pot = pot_of_cell[c] < 0 ? NB_NO_POT : (uint32_t)pot_of_cell[c];
            const uint32_t fi = type[c] == FLOOR ? aux[AUX_FIDX_OF_CELL + c] : 0u;
            nb[4 * k + d] = nb_pack(type[c], slot, pot, fi);
        }
        uint32_t* adj = reinterpret_cast<uint32_t*>(t + P.off_adj) + 8 * k;
        for (int j = 0; j < 4; ++j) {
            const int c = (ar + adj_dr[j]) * W + ac + adj_dc[j];
            if (pot_of_cell[c] >= 0) adj[pot_of_cell[c]] |= 1u << (8 * j);
            if (type[c] == FLOOR) adj[4] |= 1u << (8 * j);
            adj[5] |= (aux[AUX_SLOT_OF_CELL + c] == 0xFF ? NB_NO_SLOT : (uint32_t)aux[AUX_SLOT_OF_CELL + c]) << (8 * j);
        }
        int8_t* blk = reinterpret_cast<int8_t*>(t + P.off_postab) + (size_t)k * NWP * 4;
        // nearest pots, used by the kernel for the dynamic pot fields as well
        std::vector<std::pair<int, int>> pots;  // (distance, pot id)
        for (int p = 0; p < NP; ++p) pots.push_back({dist_of(dist_metric, ar - pot_cells[p] / W, ac - pot_cells[p] % W), p});
        std::stable_sort(pots.begin(), pots.end(), by_distance);
        for (int j = 0; j < 2 && j < NP; ++j) t[P.off_potorder + 2 * k + j] = (uint8_t)pots[j].second;
        if (!spatial) continue;
        for (int j = 0; j < 4; ++j)
            if (type[(ar + adj_dr[j]) * W + ac + adj_dc[j]] == COUNTER) blk[7 + j] = 1;
        if (full) {
            static_closest(type, W, ncell, PLATE_STACK, 2, ar, ac, dist_metric, blk + 43);
            static_closest(type, W, ncell, ONION_STACK, 2, ar, ac, dist_metric, blk + 47);
            static_closest(type, W, ncell, DELIVERY, 2, ar, ac, dist_metric, blk + 59);
            static_closest(type, W, ncell, COUNTER, 4, ar, ac, dist_metric, blk + 63);
            for (int j = 0; j < 2 && j < NP; ++j) {
                const int pc = pot_cells[pots[j].second];
                int8_t* q = blk + 71 + 11 * j;
                q[0] = 1;  // "reachable", a constant in the feature this stands in for
                q[7] = (int8_t)(ar - pc / W); q[8] = (int8_t)(ac - pc % W);
                q[9] = (int8_t)(pc / W); q[10] = (int8_t)(pc % W);
            }
        }
        blk[o_pos] = (int8_t)ar; blk[o_pos + 1] = (int8_t)ac;
    }
    return MARL_OK;
}

// ---------------------------------------------------------------------------------------
// snapshot kernels (one thread per env; setup / test path, not the hot path)
// ---------------------------------------------------------------------------------------
__global__ void export_kernel(EnvParams P, int A, int niw, int ncell, int W, const uint8_t* __restrict__ aux,
                              const uint32_t* __restrict__ st, int32_t* agents, int32_t* pots, uint8_t* items,
                              int32_t* misc) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= P.E) return;
    const size_t E = (size_t)P.E;
    int w = 0;
    misc[2 * e] = (int32_t)st[(w++) * E + e];
    misc[2 * e + 1] = (int32_t)st[(w++) * E + e];
    const uint32_t pw = st[(w++) * E + e];
    for (int p = 0; p < P.NP; ++p) {
        pots[((size_t)e * P.NP + p) * 2] = (int32_t)((pw >> (8 * p)) & 3u);
        pots[((size_t)e * P.NP + p) * 2 + 1] = (int32_t)((pw >> (8 * p + 2)) & 31u);
    }
    for (int k = 0; k < (A + 1) / 2; ++k) {
        const uint32_t aw = st[(w++) * E + e];
        for (int h = 0; h < 2 && 2 * k + h < A; ++h) {
            const uint32_t v = aw >> (16 * h);
            int32_t* o = agents + ((size_t)e * A + 2 * k + h) * 4;
            const uint32_t cell = aux[AUX_FLOOR_CELL + (v & 0xFFu)];
            o[0] = (int32_t)(cell / W); o[1] = (int32_t)(cell % W); o[2] = (v >> 8) & 3u; o[3] = (v >> 10) & 3u;
        }
    }
    for (int c = 0; c < ncell; ++c) {
        const uint32_t slot = aux[AUX_SLOT_OF_CELL + c];
        uint8_t it = 0;
        if (slot != 0xFFu) it = (uint8_t)((st[(size_t)(w + (slot >> 4)) * E + e] >> ((slot & 15u) * 2u)) & 3u);
        items[(size_t)e * ncell + c] = it;
    }
}

__global__ void import_kernel(EnvParams P, int A, int niw, int ncell, int W, const uint8_t* __restrict__ aux,
                              uint32_t* st, const int32_t* __restrict__ agents, const int32_t* __restrict__ pots,
                              const uint8_t* __restrict__ items, const int32_t* __restrict__ misc) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= P.E) return;
    const size_t E = (size_t)P.E;
    int w = 0;
    st[(w++) * E + e] = (uint32_t)misc[2 * e];
    st[(w++) * E + e] = (uint32_t)misc[2 * e + 1];
    uint32_t pw = 0;
    for (int p = 0; p < P.NP; ++p) {
        const uint32_t b = ((uint32_t)pots[((size_t)e * P.NP + p) * 2] & 3u) |
                           (((uint32_t)pots[((size_t)e * P.NP + p) * 2 + 1] & 31u) << 2);
        pw |= b << (8 * p);
    }
    st[(w++) * E + e] = pw;
    for (int k = 0; k < (A + 1) / 2; ++k) {
        uint32_t aw = 0;
        for (int h = 0; h < 2 && 2 * k + h < A; ++h) {
            const int32_t* o = agents + ((size_t)e * A + 2 * k + h) * 4;
            // a position off the grid or off the floor is replaced by the first floor cell: the
            // lookup must stay inside the table whatever the caller uploaded
            // (each coordinate against its own bound: a product of hostile values could wrap into range)
            const bool inside = o[0] >= 0 && o[0] < ncell / W && o[1] >= 0 && o[1] < W;
            uint32_t fi = inside ? aux[AUX_FIDX_OF_CELL + (uint32_t)(o[0] * W + o[1])] : 0xFFu;
            fi = fi == 0xFFu ? 0u : fi;
            const uint32_t v = fi | (((uint32_t)o[2] & 3u) << 8) | (((uint32_t)o[3] & 3u) << 10);
            aw |= v << (16 * h);
        }
        st[(w++) * E + e] = aw;
    }
    for (int k = 0; k < niw; ++k) {
        uint32_t word = 0;
        for (int j = 0; j < 16; ++j) {
            const int slot = k * 16 + j;
            if (slot < P.nslots) word |= ((uint32_t)items[(size_t)e * ncell + aux[AUX_CELL_OF_SLOT + slot]] & 3u) << (2 * j);
        }
        st[(size_t)(w + k) * E + e] = word;
    }
}

// ---------------------------------------------------------------------------------------
// entry points
// ---------------------------------------------------------------------------------------
// Observation bytes per launch above which the stores carry the non-temporal hint.  Measured
// on MI355X (cramped_room, 2 agents, global_obs): at 106 MB per launch the hint costs 22 %, at
// 159 MB it gains 1 %, from 212 MB on 3-5 %: the crossover is where the rows stop fitting the
// 256 MiB last-level cache together with everything else the step touches.
static constexpr size_t STREAM_NT_THRESHOLD_BYTES = (size_t)128 << 20;

extern "C" const char* marl_last_error(void) { return g_err.c_str(); }

extern "C" int marl_env_create(marl_env_t** out, const char* const* layout_rows, int height, int num_agents,
                               int feature, int max_steps, int num_envs, uint64_t seed, int64_t env_index_base,
                               int dist_metric) {
    if (!out) return fail(MARL_ERR_INVALID, "out is NULL");
    *out = nullptr;
    if (num_agents < 1 || num_agents > MAX_AGENTS)
        return fail(MARL_ERR_INVALID, "num_agents must be 1, 2, 3 or 4");
    if (feature < 0 || feature > FEAT_ONLY_DIRECTION) return fail(MARL_ERR_INVALID, "unknown feature id");
    if (max_steps <= 0 || max_steps > 65535) return fail(MARL_ERR_INVALID, "max_steps must be in [1, 65535]");
    if (num_envs <= 0) return fail(MARL_ERR_INVALID, "num_envs must be positive");
    if ((int64_t)num_envs * num_agents > 0x7FFFFFFF) return fail(MARL_ERR_INVALID, "num_envs * num_agents exceeds 2^31-1");
    if (dist_metric != 0 && dist_metric != 1) return fail(MARL_ERR_INVALID, "dist_metric must be 0 or 1");
    marl_env* env = new marl_env();
    env->A = num_agents; env->feat = feature; env->d_tab = nullptr; env->d_aux = nullptr;
    int rc = build_tables(env, layout_rows, height, dist_metric);
    if (rc != MARL_OK) { delete env; return rc; }
    EnvParams& P = env->P;
    P.E = num_envs; P.max_steps = max_steps; P.auto_reset = 1; P.nsteps = 1;
    {
        // MARL_AMD_STREAM_NT=0/1 forces the choice (tuning aid)
        const size_t obs_bytes = (size_t)num_envs * num_agents * env->D * sizeof(float);
        const char* v = getenv("MARL_AMD_STREAM_NT");
        env->stream_nt_forced = v ? (atoi(v) != 0) : -1;
        P.stream_nt = v ? env->stream_nt_forced : (obs_bytes > STREAM_NT_THRESHOLD_BYTES);
    }
    uint32_t k = mix32_host((uint32_t)seed + 0x9E3779B9u);
    P.seed_key = mix32_host(k ^ (uint32_t)(seed >> 32));
    P.env_base_lo = (uint32_t)(uint64_t)env_index_base;
    P.env_base_hi = (uint32_t)((uint64_t)env_index_base >> 32);
    hipError_t e = hipGetDevice(&env->device);
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&env->d_tab), env->tab.size());
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&env->d_aux), env->aux.size());
    if (e == hipSuccess) e = hipMemcpy(env->d_tab, env->tab.data(), env->tab.size(), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(env->d_aux, env->aux.data(), env->aux.size(), hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        if (env->d_tab) (void)hipFree(env->d_tab);
        if (env->d_aux) (void)hipFree(env->d_aux);
        delete env;
        return fail_hip(e, "device allocation / upload of the lookup tables");
    }
    *out = env;
    return MARL_OK;
}

extern "C" void marl_env_destroy(marl_env_t* env) {
    if (!env) return;
    if (env->d_tab) (void)hipFree(env->d_tab);
    if (env->d_aux) (void)hipFree(env->d_aux);
    delete env;
}

extern "C" int marl_env_set_auto_reset(marl_env_t* env, int enabled) {
    if (!env) return fail(MARL_ERR_INVALID, "env is NULL");
    env->P.auto_reset = enabled ? 1 : 0;
    return MARL_OK;
}

extern "C" int marl_env_obs_dim(const marl_env_t* env) { return env ? env->D : MARL_ERR_INVALID; }
extern "C" int marl_env_num_rows(const marl_env_t* env) { return env ? env->P.E * env->A : MARL_ERR_INVALID; }
extern "C" int marl_env_num_pots(const marl_env_t* env) { return env ? env->NP : MARL_ERR_INVALID; }
extern "C" int marl_env_grid_cells(const marl_env_t* env) { return env ? env->ncell : MARL_ERR_INVALID; }
extern "C" int marl_env_state_words(const marl_env_t* env) { return env ? env->words : MARL_ERR_INVALID; }
extern "C" size_t marl_env_state_bytes(const marl_env_t* env) {
    return env ? (size_t)env->words * (size_t)env->P.E * sizeof(uint32_t) : 0;
}

// every call on a handle must run with the handle's device current: its tables live there
static int check_device(const marl_env* env) {
    int dev = -1;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return fail_hip(e, "hipGetDevice");
    if (dev != env->device)
        return fail(MARL_ERR_INVALID, "the env was created on HIP device " + std::to_string(env->device) +
                                          " but device " + std::to_string(dev) + " is current");
    return MARL_OK;
}

static int run_env_kernel(marl_env* env, void* d_state, const void* d_actions, int act_size, const uint8_t* prio,
                          const uint8_t* spawn_ov, const uint8_t* dir_ov, float* d_obs, float* d_rewards,
                          uint8_t* d_term, uint8_t* d_trunc, int mode, void* stream, int nsteps = 1) {
    if (!env) return fail(MARL_ERR_INVALID, "env is NULL");
    if (!d_state) return fail(MARL_ERR_INVALID, "d_state is NULL");
    if (!d_obs) return fail(MARL_ERR_INVALID, "d_obs is NULL");
    if (((uintptr_t)d_obs & 15u) != 0) return fail(MARL_ERR_INVALID, "d_obs must be 16-byte aligned");
    if (((uintptr_t)d_state & 3u) != 0) return fail(MARL_ERR_INVALID, "d_state must be 4-byte aligned");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    hipError_t e;
    if (const int rc = check_device(env); rc != MARL_OK) return rc;
    uint32_t* st = reinterpret_cast<uint32_t*>(d_state);
    EnvParams P = env->P;
    P.nsteps = nsteps;
    if (nsteps > 1 && env->stream_nt_forced < 0) {
        // the cache policy follows what the launch writes: all steps' rows
        const size_t obs_bytes = (size_t)nsteps * (size_t)P.E * env->A * env->D * sizeof(float);
        P.stream_nt = obs_bytes > STREAM_NT_THRESHOLD_BYTES;
    }
    switch (env->A) {
        case 1: e = launch_env_kernel_A<1>(env->feat, env->niw, P, env->d_tab, env->d_aux, st, d_actions, act_size, prio, spawn_ov, dir_ov, d_obs, d_rewards, d_term, d_trunc, mode, s); break;
        case 2: e = launch_env_kernel_A<2>(env->feat, env->niw, P, env->d_tab, env->d_aux, st, d_actions, act_size, prio, spawn_ov, dir_ov, d_obs, d_rewards, d_term, d_trunc, mode, s); break;
        case 3: e = launch_env_kernel_A<3>(env->feat, env->niw, P, env->d_tab, env->d_aux, st, d_actions, act_size, prio, spawn_ov, dir_ov, d_obs, d_rewards, d_term, d_trunc, mode, s); break;
        case 4: e = launch_env_kernel_A<4>(env->feat, env->niw, P, env->d_tab, env->d_aux, st, d_actions, act_size, prio, spawn_ov, dir_ov, d_obs, d_rewards, d_term, d_trunc, mode, s); break;
        default: return fail(MARL_ERR_INVALID, "unsupported agent count");
    }
    if (e != hipSuccess) return fail_hip(e, "env kernel launch");
    return MARL_OK;
}

extern "C" int marl_env_reset(marl_env_t* env, void* d_state, float* d_obs, const uint8_t* d_spawn_override,
                              const uint8_t* d_dir_override, void* stream) {
    return run_env_kernel(env, d_state, nullptr, 0, nullptr, d_spawn_override, d_dir_override, d_obs, nullptr, nullptr,
                          nullptr, 1, stream);
}

extern "C" int marl_env_step(marl_env_t* env, void* d_state, const void* d_actions, int action_elem_size, float* d_obs,
                             float* d_rewards, uint8_t* d_terminated, uint8_t* d_truncated,
                             const uint8_t* d_prio_override, void* stream) {
    if (!d_actions || !d_rewards || !d_terminated || !d_truncated)
        return fail(MARL_ERR_INVALID, "actions / rewards / terminated / truncated must not be NULL");
    if (action_elem_size != 1 && action_elem_size != 4 && action_elem_size != 8)
        return fail(MARL_ERR_INVALID, "action_elem_size must be 1, 4 or 8");
    if (((uintptr_t)d_actions & (uintptr_t)(action_elem_size - 1)) != 0)
        return fail(MARL_ERR_INVALID, "d_actions is not aligned to its element size");
    if (((uintptr_t)d_rewards & 3u) != 0) return fail(MARL_ERR_INVALID, "d_rewards must be 4-byte aligned");
#ifdef OC_ABLATE
    // diagnostic build only (profiles/diag/build_variants.sh ablate:-DOC_ABLATE): MARL_AMD_ABLATE=1|2
    // times the simulation without the store stream, or the store stream without the simulation;
    // results are then meaningless.  The product library does not read the variable.
    static const int ablate = [] { const char* v = getenv("MARL_AMD_ABLATE"); return v ? atoi(v) & 31 : 0; }();
#else
    constexpr int ablate = 0;
#endif
    return run_env_kernel(env, d_state, d_actions, action_elem_size, d_prio_override, nullptr, nullptr, d_obs,
                          d_rewards, d_terminated, d_truncated, ablate << 8, stream);
}


extern "C" int marl_env_step_many(marl_env_t* env, void* d_state, const void* d_actions, int action_elem_size,
                                  int num_steps, float* d_obs, float* d_rewards, uint8_t* d_terminated,
                                  uint8_t* d_truncated, const uint8_t* d_prio_override, void* stream) {
    if (!d_actions || !d_rewards || !d_terminated || !d_truncated)
        return fail(MARL_ERR_INVALID, "actions / rewards / terminated / truncated must not be NULL");
    if (action_elem_size != 1 && action_elem_size != 4 && action_elem_size != 8)
        return fail(MARL_ERR_INVALID, "action_elem_size must be 1, 4 or 8");
    if (((uintptr_t)d_actions & (uintptr_t)(action_elem_size - 1)) != 0)
        return fail(MARL_ERR_INVALID, "d_actions is not aligned to its element size");
    if (((uintptr_t)d_rewards & 3u) != 0) return fail(MARL_ERR_INVALID, "d_rewards must be 4-byte aligned");
    if (num_steps < 1 || num_steps > 65536) return fail(MARL_ERR_INVALID, "num_steps must be in [1, 65536]");
    if (env && (((size_t)env->P.E * env->A * env->D) & 3u) != 0 && num_steps > 1)
        return fail(MARL_ERR_INVALID, "rows of one step must be a multiple of 16 bytes for a multi-step launch "
                                      "(every step's rows start 16-byte aligned)");
    return run_env_kernel(env, d_state, d_actions, action_elem_size, d_prio_override, nullptr, nullptr, d_obs,
                          d_rewards, d_terminated, d_truncated, 0, stream, num_steps);
}

extern "C" int marl_env_featurize(marl_env_t* env, void* d_state, float* d_obs, void* stream) {
    return run_env_kernel(env, d_state, nullptr, 0, nullptr, nullptr, nullptr, d_obs, nullptr, nullptr, nullptr, 2, stream);
}

extern "C" int marl_env_export_state(marl_env_t* env, const void* d_state, int32_t* d_agents, int32_t* d_pots,
                                     uint8_t* d_items, int32_t* d_misc, void* stream) {
    if (!env || !d_state || !d_agents || !d_items || !d_misc || (env->NP > 0 && !d_pots))
        return fail(MARL_ERR_INVALID, "NULL argument to marl_env_export_state");
    if (const int rc = check_device(env); rc != MARL_OK) return rc;
    const int block = 128, grid = (env->P.E + block - 1) / block;
    hipLaunchKernelGGL(export_kernel, dim3(grid), dim3(block), 0, reinterpret_cast<hipStream_t>(stream), env->P, env->A,
                       env->niw, env->ncell, env->W, env->d_aux, reinterpret_cast<const uint32_t*>(d_state), d_agents, d_pots,
                       d_items, d_misc);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? MARL_OK : fail_hip(e, "export kernel launch");
}

extern "C" int marl_env_import_state(marl_env_t* env, void* d_state, const int32_t* d_agents, const int32_t* d_pots,
                                     const uint8_t* d_items, const int32_t* d_misc, void* stream) {
    if (!env || !d_state || !d_agents || !d_items || !d_misc || (env->NP > 0 && !d_pots))
        return fail(MARL_ERR_INVALID, "NULL argument to marl_env_import_state");
    if (const int rc = check_device(env); rc != MARL_OK) return rc;
    const int block = 128, grid = (env->P.E + block - 1) / block;
    hipLaunchKernelGGL(import_kernel, dim3(grid), dim3(block), 0, reinterpret_cast<hipStream_t>(stream), env->P, env->A,
                       env->niw, env->ncell, env->W, env->d_aux, reinterpret_cast<uint32_t*>(d_state), d_agents, d_pots, d_items,
                       d_misc);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? MARL_OK : fail_hip(e, "import kernel launch");
}

// workspace: [T floats: per-step reward mean, padded to 8 bytes][T * chunks doubles: partial sums]
static size_t ws_mean_bytes(int T) { return ((size_t)T * sizeof(float) + 7) & ~(size_t)7; }

extern "C" size_t marl_gae_workspace_bytes(int T, int N) {
    if (T <= 0 || N <= 0) return 0;
    return ws_mean_bytes(T) + (size_t)T * (size_t)row_sum_chunks(N) * sizeof(double);
}

extern "C" int marl_row_sums(const float* d_x, int T, int N, double* d_sums, void* d_workspace, void* stream) {
    if (!d_x || !d_sums || !d_workspace || T <= 0 || N <= 0) return fail(MARL_ERR_INVALID, "bad argument to marl_row_sums");
    if (((uintptr_t)d_workspace & 7u) != 0) return fail(MARL_ERR_INVALID, "d_workspace must be 8-byte aligned");
    double* partial = reinterpret_cast<double*>(reinterpret_cast<char*>(d_workspace) + ws_mean_bytes(T));
    hipError_t e = launch_row_sums(d_x, T, N, partial, d_sums, nullptr, reinterpret_cast<hipStream_t>(stream));
    return e == hipSuccess ? MARL_OK : fail_hip(e, "row sum kernel launch");
}

extern "C" int marl_gae(const float* d_rewards, const float* d_dones, const float* d_values,
                        const float* d_next_values, int T, int N, double gamma, double lam, float* d_adv, float* d_ret,
                        int mode, void* d_workspace, int reward_mean_ready, void* stream) {
    if (!d_rewards || !d_dones || !d_values || !d_next_values || !d_adv)
        return fail(MARL_ERR_INVALID, "NULL argument to marl_gae");
    if (T <= 0 || N <= 0) return fail(MARL_ERR_INVALID, "T and N must be positive");
    if (mode != 0 && mode != 1) return fail(MARL_ERR_INVALID, "mode must be 0 (MAPPO) or 1 (CMAPPO)");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    float* mean = nullptr;
    if (mode == 1) {
        if (!d_workspace) return fail(MARL_ERR_INVALID, "mode 1 needs d_workspace (marl_gae_workspace_bytes(T, N) bytes)");
        if (((uintptr_t)d_workspace & 7u) != 0) return fail(MARL_ERR_INVALID, "d_workspace must be 8-byte aligned");
        mean = reinterpret_cast<float*>(d_workspace);
        if (!reward_mean_ready) {
            double* partial = reinterpret_cast<double*>(reinterpret_cast<char*>(d_workspace) + ws_mean_bytes(T));
            hipError_t e = launch_row_sums(d_rewards, T, N, partial, nullptr, mean, s);
            if (e != hipSuccess) return fail_hip(e, "row sum kernel launch");
        }
    }
    // Python forms gamma * lam in double; each scalar is rounded to float32 when it meets a tensor
    hipError_t e = launch_gae(d_rewards, d_dones, d_values, d_next_values, T, N, (float)gamma, (float)(gamma * lam),
                              d_adv, d_ret, mode, mean, s);
    return e == hipSuccess ? MARL_OK : fail_hip(e, "gae kernel launch");
}

extern "C" size_t marl_reward_events_workspace_bytes(int num_envs, int num_agents) {
    if (num_envs <= 0 || num_agents <= 0) return 0;
    return (size_t)reward_event_chunks(num_envs * num_agents) * 4 * sizeof(uint64_t);
}

extern "C" int marl_reward_events(const float* d_rewards, int num_envs, int num_agents, uint64_t* d_totals,
                                  double* d_first_rows, void* d_workspace, void* stream) {
    if (!d_rewards || !d_totals || !d_workspace) return fail(MARL_ERR_INVALID, "NULL argument to marl_reward_events");
    if (num_envs <= 0 || num_agents <= 0 || (int64_t)num_envs * num_agents > 0x7FFFFFFF)
        return fail(MARL_ERR_INVALID, "num_envs and num_agents must be positive and their product below 2^31");
    if (((uintptr_t)d_workspace & 7u) != 0 || ((uintptr_t)d_totals & 7u) != 0 || ((uintptr_t)d_first_rows & 7u) != 0)
        return fail(MARL_ERR_INVALID, "d_totals, d_first_rows and d_workspace must be 8-byte aligned");
    hipError_t e = launch_reward_events(d_rewards, num_envs, num_agents, reinterpret_cast<uint64_t*>(d_workspace), d_totals,
                                        d_first_rows, reinterpret_cast<hipStream_t>(stream));
    return e == hipSuccess ? MARL_OK : fail_hip(e, "reward event kernel launch");
}

extern "C" size_t marl_reward_events_many_workspace_bytes(int num_steps, int num_envs, int num_agents) {
    if (num_steps <= 0 || num_envs <= 0 || num_agents <= 0) return 0;
    return (size_t)num_steps * (size_t)reward_event_chunks(num_envs * num_agents) * 4 * sizeof(uint64_t);
}

extern "C" int marl_reward_events_many(const float* d_rewards, int num_steps, int num_envs, int num_agents,
                                       uint64_t* d_totals, double* d_first_rows, void* d_workspace, void* stream) {
    if (!d_rewards || !d_totals || !d_workspace) return fail(MARL_ERR_INVALID, "NULL argument to marl_reward_events_many");
    // (the finishing pass is one workgroup that walks the steps in order: kept to rollout lengths)
    if (num_steps <= 0 || num_steps > MARL_REWARD_EVENTS_MANY_MAX_STEPS)
        return fail(MARL_ERR_INVALID, "num_steps must be in [1, " + std::to_string(MARL_REWARD_EVENTS_MANY_MAX_STEPS) + "]");
    if (num_envs <= 0 || num_agents <= 0 || (int64_t)num_envs * num_agents > 0x7FFFFFFF)
        return fail(MARL_ERR_INVALID, "num_envs and num_agents must be positive and their product below 2^31");
    if (((uintptr_t)d_workspace & 7u) != 0 || ((uintptr_t)d_totals & 7u) != 0 || ((uintptr_t)d_first_rows & 7u) != 0)
        return fail(MARL_ERR_INVALID, "d_totals, d_first_rows and d_workspace must be 8-byte aligned");
    hipError_t e = launch_reward_events_many(d_rewards, num_steps, num_envs, num_agents,
                                             reinterpret_cast<uint64_t*>(d_workspace), d_totals, d_first_rows,
                                             reinterpret_cast<hipStream_t>(stream));
    return e == hipSuccess ? MARL_OK : fail_hip(e, "reward event kernel launch");
}

extern "C" int marl_sample_categorical(const float* d_logits, int num_rows, int num_actions, int row_stride,
                                       uint64_t seed, const uint64_t* d_counter, int64_t row_index_base,
                                       int64_t* d_actions, float* d_logprob, float* d_entropy, float* d_next_column,
                                       void* stream) {
    if (!d_logits || !d_actions) return fail(MARL_ERR_INVALID, "NULL argument to marl_sample_categorical");
    if (num_rows <= 0 || num_actions <= 0 || num_actions > 64)
        return fail(MARL_ERR_INVALID, "num_rows must be positive and num_actions in [1, 64]");
    if (row_stride == 0) row_stride = num_actions;
    if (row_stride < num_actions) return fail(MARL_ERR_INVALID, "row_stride is smaller than num_actions");
    if (d_next_column && row_stride <= num_actions)
        return fail(MARL_ERR_INVALID, "d_next_column needs rows of more than num_actions floats (row_stride)");
    if (((uintptr_t)d_actions & 7u) != 0 || ((uintptr_t)d_counter & 7u) != 0 || ((uintptr_t)d_logits & 3u) != 0)
        return fail(MARL_ERR_INVALID, "d_actions and d_counter must be 8-byte aligned, d_logits 4-byte aligned");
    uint32_t k = mix32_host((uint32_t)seed + 0x9E3779B9u);
    k = mix32_host(k ^ (uint32_t)(seed >> 32));
    hipError_t e = launch_sample_categorical(d_logits, num_rows, num_actions, row_stride, k, d_counter,
                                             (uint64_t)row_index_base, reinterpret_cast<long long*>(d_actions), d_logprob,
                                             d_entropy, d_next_column, reinterpret_cast<hipStream_t>(stream));
    return e == hipSuccess ? MARL_OK : fail_hip(e, "sampling kernel launch");
}

extern "C" size_t marl_ppo_loss_workspace_bytes(int num_samples) {
    if (num_samples <= 0) return 0;
    return (size_t)ppo_loss_chunks(num_samples) * 6 * sizeof(double);
}

extern "C" int marl_ppo_loss(const float* d_logits, int num_samples, int num_actions, int row_stride,
                             const float* d_values, const int64_t* d_index, const float* d_actions,
                             const float* d_old_logprob, const float* d_advantages, const float* d_value_target,
                             double clip, double entropy_coef, double value_coef, float* d_grad_logits,
                             float* d_grad_values, double* d_stats, void* d_workspace, void* stream) {
    if (!d_logits || !d_values || !d_actions || !d_old_logprob || !d_advantages || !d_value_target || !d_grad_logits ||
        !d_grad_values || !d_stats || !d_workspace)
        return fail(MARL_ERR_INVALID, "NULL argument to marl_ppo_loss");
    if (num_samples <= 0 || num_actions < 2 || num_actions > 8)
        return fail(MARL_ERR_INVALID, "num_samples must be positive and num_actions in [2, 8]");
    if (row_stride == 0) row_stride = num_actions;
    if (row_stride < num_actions) return fail(MARL_ERR_INVALID, "row_stride is smaller than num_actions");
    if (((uintptr_t)d_stats & 7u) != 0 || ((uintptr_t)d_workspace & 7u) != 0 || ((uintptr_t)d_index & 7u) != 0)
        return fail(MARL_ERR_INVALID, "d_stats, d_workspace and d_index must be 8-byte aligned");
    hipError_t e = launch_ppo_loss(d_logits, num_samples, num_actions, row_stride, d_values,
                                   reinterpret_cast<const long long*>(d_index), d_actions, d_old_logprob, d_advantages,
                                   d_value_target, (float)clip, (float)entropy_coef, (float)value_coef, d_grad_logits,
                                   d_grad_values, d_stats, reinterpret_cast<double*>(d_workspace),
                                   reinterpret_cast<hipStream_t>(stream));
    return e == hipSuccess ? MARL_OK : fail_hip(e, "PPO loss kernel launch");
}
