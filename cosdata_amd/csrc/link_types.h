// link_types.h — POD arguments of the builder's device-side link phase (kernels_link.hip).
#pragma once
#include <stdint.h>
#include "device_common.h"

namespace cosdev {

// Mutable link state of one level: the adjacency the walk kernel reads (adj_node / adj_vec) plus what
// ProbNode::add_neighbor keeps next to every neighbour pointer — the slot's similarity, as a MetricResult order key — and
// the node's cached (lowest_index, lowest similarity) of prob_node.rs:108.
struct LinkLevelDev {
    u32 *adj_node;       // [n][M] neighbour's node index within the level, 0xFFFFFFFF = null slot (level 0: same array as adj_vec)
    u32 *adj_vec;        // [n][M] neighbour's vector row
    const u32 *node_vec; // [n] node index -> vector row (levels >= 1)
    int32_t *key;        // [n][M] order key of the slot's similarity; INT32_MIN for a null slot
    uint8_t *low_idx;    // [n]
    int32_t *low_key;    // [n]
    u32 *owner;          // [n] claim tag of the round that last claimed the row: (round << 13) | (8191 - batch position)
    const uint8_t *kind; // pseudo-root component only: ReplicaNodeKind of every node (0 Base, 1 Pseudo, 2 Metadata); nullptr on base graphs
    u32 M;
};

struct LinkArgs {
    LinkLevelDev lv[MAX_LEVELS];
    u32 L1;              // number of levels
    const u32 *z_nodes;  // [B][L1][64] walk results per (batch item, level slot): node index within the level, best first
    const float *z_sims; // [B][L1][64]
    const u32 *z_counts; // [B][L1]
    const u32 *me;       // [B][L1] node index of batch item b on level l (undefined above its max level)
    u32 metric;
    int32_t kmin, kmax;  // order keys of MetricResult::min / ::max (types.rs:435-457)
};

constexpr u32 LINK_MAX_BATCH = 8192;       // batch position must fit the 13-bit field of the claim tag
constexpr u32 LINK_MAX_ROUND = (1u << 19) - 1;

} // namespace cosdev
