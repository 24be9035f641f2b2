// Shared between host table builder and device kernels: constants, feature-block
// geometry, the packed per-env state layout and the layout of the lookup blob that every
// workgroup stages into LDS.
#pragma once

#include <stdint.h>

namespace oc {

// static cell types
enum : uint32_t { WALL = 0, FLOOR = 1, COUNTER = 2, POT = 3, ONION_STACK = 4, PLATE_STACK = 5, DELIVERY = 6 };
// items (held, lying on the floor, or placed on a counter)
enum : uint32_t { NONE = 0, ONION = 1, PLATE = 2, SOUP = 3 };
// feature sets: registered names at overcooked_config.py:14-28; only_direction: README.md:65
enum : int { FEAT_GLOBAL = 0, FEAT_LOCAL = 1, FEAT_MINIMAL = 2, FEAT_MINIMAL_OTHER = 3, FEAT_ONLY_DIRECTION = 4 };
// cardinal action set, index 6 = no-op, 2 = move left (tests/test_load.py:37-43)
enum : uint32_t { ACT_UP = 0, ACT_DOWN = 1, ACT_LEFT = 2, ACT_RIGHT = 3, ACT_PICKUP_DROP = 4, ACT_TOGGLE = 5, ACT_NOOP = 6 };
// directions, README.md:112
enum : uint32_t { DIR_RIGHT = 0, DIR_DOWN = 1, DIR_LEFT = 2, DIR_UP = 3 };

constexpr int POT_CAPACITY = 3;
constexpr int COOK_TIME = 30;
constexpr int MAX_CELLS = 256;
constexpr int MAX_POTS = 4;     // one byte per pot in a single state word
constexpr int MAX_SLOTS = 126;  // item-capable cells; 127 is the "no slot" code
constexpr int MAX_AGENTS = 4;
// lanes one env takes in a wave: the agent count rounded up to a power of two
constexpr int lanes_per_env(int A) { return A <= 1 ? 1 : A <= 2 ? 2 : 4; }

// pot condition classes and where each lands in the two pot encodings.
// An empty pot is row 0 of NextToPot and index 1 of the pot-feature status one-hot: both
// pinned by states/agent_0_dropping_onion_state.npy; cooking = 2 and ready = 3 follow the
// docstring at overcooked_features.py:23.
enum : uint32_t { POT_EMPTY = 0, POT_PARTIAL = 1, POT_COOKING = 2, POT_READY = 3 };
constexpr uint32_t NEXT_TO_POT_ROW_PACKED = 0x3210;    // nibble k = row of class k
constexpr uint32_t POT_STATUS_INDEX_PACKED = 0x3201;   // nibble k = one-hot index of class k
// OvercookedInventory one-hot index per item (onion = 0 pinned by the golden vector)
constexpr uint32_t INVENTORY_INDEX_PACKED = 0x1200;    // nibble item -> index: onion 0, plate 2, soup 1

// Geometry of one agent's feature block (overcooked_features.py:38-78, :239-254, :291-298).
template <int A, int FEAT>
struct FeatGeom {
    static constexpr bool kFull = (FEAT == FEAT_GLOBAL || FEAT == FEAT_LOCAL);
    static constexpr bool kOthers = kFull || FEAT == FEAT_MINIMAL_OTHER;
    static constexpr bool kSpatial = FEAT != FEAT_ONLY_DIRECTION;
    static constexpr int O_DIR = 0;
    static constexpr int O_INV = 4;
    static constexpr int O_NEXT_COUNTER = 7;
    static constexpr int O_NEXT_POT = 11;
    // full sets only
    static constexpr int O_ONION = 27;
    static constexpr int O_PLATE = 35;
    static constexpr int O_PLATE_STACK = 43;
    static constexpr int O_ONION_STACK = 47;
    static constexpr int O_SOUP = 51;
    static constexpr int O_DELIVERY = 59;
    static constexpr int O_COUNTER = 63;
    static constexpr int O_POT0 = 71;
    static constexpr int O_POT1 = 82;
    static constexpr int O_OTHERS = kFull ? 93 : 27;
    static constexpr int O_POS = O_OTHERS + (kOthers ? 2 * (A - 1) : 0);
    static constexpr int O_CAN_MOVE = O_POS + 2;
    static constexpr int BLK = kSpatial ? O_CAN_MOVE + 4 : 4;
    static constexpr int NW = (BLK + 3) / 4;      // block padded to dwords
    static constexpr int NWP = (NW + 3) / 4 * 4;  // template row stride in dwords (16-byte rows)
    static constexpr int D = (FEAT == FEAT_GLOBAL) ? A * BLK : BLK;
};

inline int block_dim(int A, int feat) {
    switch (feat) {
        case FEAT_GLOBAL:
        case FEAT_LOCAL: return 99 + 2 * (A - 1);
        case FEAT_MINIMAL: return 33;
        case FEAT_MINIMAL_OTHER: return 33 + 2 * (A - 1);
        case FEAT_ONLY_DIRECTION: return 4;
    }
    return -1;
}

// ---- neighbour word: what lies one step from a floor cell in one direction ----
//   bits 0-2   cell type
//   bits 3-9   item slot of that cell (127 = cannot hold an item)
//   bits 10-12 pot id (7 = not a pot)
//   bits 16-23 floor index of that cell (valid when the type is FLOOR)
constexpr uint32_t NB_NO_SLOT = 127;
constexpr uint32_t NB_NO_POT = 7;
inline uint32_t nb_pack(uint32_t type, uint32_t slot, uint32_t pot, uint32_t fidx) {
    return type | (slot << 3) | (pot << 10) | (fidx << 16);
}

// ---- decision tables of the interact rule chain (built once on the host) ----
// interact table, 1024 bytes, index = type | lying << 3 | holding << 5 | pot state << 7 | agent ahead << 9
//   entry = new holding | item op << 2 | pot op << 4
// reward table, 128 bytes, index = type | holding before the step << 3 | pot state before the step << 5
//   entry = onion_in_pot | soup_in_dish << 1 | delivery << 2
enum : uint32_t { POTST_ROOM = 0, POTST_COOKING = 1, POTST_READY = 2 };
enum : uint32_t { INT_ITEM_KEEP = 0, INT_ITEM_CLEAR = 1, INT_ITEM_PUT = 2 };
enum : uint32_t { INT_POT_KEEP = 0, INT_POT_ADD = 1, INT_POT_RESET = 2 };
constexpr int INT_TABLE_BYTES = 1024;
constexpr int REW_TABLE_BYTES = 128;
// reward values, float32[(MAX_AGENTS + 1) * 4], index = onion_in_pot | soup_in_dish << 1 | deliveries in the env << 2:
// the Python-float sum 0.1 + 0.3 + 1.0 + ... in the config's list order, rounded to float32 once
constexpr int REW_VALUE_BYTES = (4 + 1) * 4 * 4;

// ---- hot lookup blob, indexed by FLOOR INDEX (agents only ever stand on floor cells) ----
// All offsets are multiples of 16 bytes and live in EnvParams.
//   nb       uint32[nfloor][4]     neighbour words in direction order (right, down, left, up)
//   postab   uint32[nfloor][NWP]   static part of the feature block of an agent standing there
//   adj      uint32[nfloor][8]     [0..3] per pot: byte k = 1 where neighbour k is that pot; [4] byte k = 1
//                                  where neighbour k is floor; [5] byte k = item slot of neighbour k
//                                  (neighbour order right, left, down, up)
//   frc      uint16[nfloor]        (row << 8) | col of the floor cell
//   slotrc   uint16[nslots]        (row << 8) | col of each item slot (slots are in row-major cell order)
//   potorder uint8[nfloor][2]      the two nearest pots (0xFF = none)
//   spawnf   uint8[nspawn]         floor indices of the '+' cells
//   inttab   uint8[1024], rewtab uint8[128]   the decision tables above
//   rewval   float[20]             reward of an agent-step by (own events, deliveries in the env)
struct EnvParams {
    int E;            // envs in this shard
    int NP;           // pots
    int nslots;       // cells that can hold an item
    int nfloor;
    int nspawn;
    int max_steps;
    int tab_bytes;    // size of the hot blob, multiple of 16
    int dist_metric;  // 0 = L1, 1 = squared L2
    int auto_reset;   // 1: a truncated env restarts inside the step (vector-env semantics)
    int stream_nt;    // 1: observation rows exceed the last-level cache, store them non-temporally
    int nsteps;       // consecutive steps worked off by one launch (1 for the plain step entry point)
    int off_nb, off_postab, off_adj, off_frc, off_slotrc, off_potorder, off_spawnf, off_inttab, off_rewtab, off_rewval;
    uint32_t seed_key;                  // mix of the 64-bit seed
    uint32_t env_base_lo, env_base_hi;  // global index of env 0 of this shard
};

// ---- auxiliary tables for the snapshot kernels (global memory, cell-indexed) ----
constexpr int AUX_FLOOR_CELL = 0;      // u8[256] floor index -> cell
constexpr int AUX_FIDX_OF_CELL = 256;  // u8[256] cell -> floor index
constexpr int AUX_SLOT_OF_CELL = 512;  // u8[256] cell -> item slot (0xFF none)
constexpr int AUX_CELL_OF_SLOT = 768;  // u8[256] item slot -> cell
constexpr int AUX_BYTES = 1024;

// per-env packed state, struct-of-arrays: word w of env e lives at state[w * E + e]
//   word 0: t, steps since the episode began (all 32 bits; at most max_steps <= 65535 while envs auto-reset)
//   word 1: RNG draw counter
//   word 2: pots, one byte each: count (bits 0-1) | timer (bits 2-6)
//   then (A+1)/2 agent words, 16 bits per agent: floor index (0-7) | dir (8-9) | inv (10-11)
//   then NIW item words, 2 bits per item slot
inline int agent_words(int A) { return (A + 1) / 2; }
inline int item_words_exact(int nslots) { return (nslots + 15) / 16; }
inline int item_words_padded(int nslots) {
    int w = item_words_exact(nslots);
    return w <= 1 ? 1 : w <= 2 ? 2 : w <= 4 ? 4 : 8;
}
inline int state_words(int A, int nslots) { return 3 + agent_words(A) + item_words_padded(nslots); }

}  // namespace oc
