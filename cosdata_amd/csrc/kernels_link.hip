// kernels_link.hip — the builder's link phase on the device.
//   create_node_edges                 vector_store.rs:976-1074
//   ProbNode::add_neighbor            models/prob_node.rs:210-283   (replace-lowest slot, cached lowest index/similarity)
//   remove_neighbor_by_id             models/prob_node.rs:285-306   (evictee drops its back edge, cache NOT refreshed)
//   remove_neighbor_by_index_and_id   vector_store.rs:1050-1067     (roll the forward half back when the back edge is refused)
//
// Schedule: ROUND-SYNCHRONOUS (oracle/cosdata_oracle_hnsw.c:coso_index_build_rounds, ordered variant, is the CPU statement;
// the tests assert identical graphs).  The nodes a batch adds to one level are linked in rounds.  Every pending node claims
// {itself} + its walk candidates; the first claimer of a row in batch order owns it for the round — on the device that is ONE
// atomicMax per (node, row) on a tag (round << 13 | 8191 - batch position), no scan — and a node runs iff it owns every row it
// claimed.  Runnable nodes touch disjoint rows, so one wavefront per node executes the literal sequential create_node_edges
// on its rows (a row's M slots <-> the lanes) with no synchronisation between waves.  The only rows touched outside a node's
// own claim are those of nodes it EVICTS from a candidate's row (the evictee must drop its back edge): those removals are
// queued and applied by evict_kernel after every runnable node of the round has finished, exactly like the oracle applies
// them at the end of the round (they commute: an (evictee, target) pair exists at most once).  Blocked nodes go to the next
// round.  All levels of a batch share the rounds (levels are independent graphs).
//
// HBM-bound bookkeeping (integer compares, 256 B rows); no MFMA.  Rows of the link state are read and written with
// agent-scope relaxed atomics so a wave re-reading a row it modified earlier in the same launch sees its own stores.
#include <hip/hip_runtime.h>

#include <algorithm>

#include "engine_types.h"
#include "link_types.h"

using namespace cosdev;

namespace {

constexpr u32 NONE = 0xFFFFFFFFu;
constexpr int32_t EMPTY_KEY = INT32_MIN; // below every real key: the first-minimum scan finds the first empty slot

template <typename T>
__device__ __forceinline__ T ld(const T *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
template <typename T>
__device__ __forceinline__ void st(T *p, T v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// a value loaded by lane 0 and broadcast: per-node scalars (lowest cache) are always read and written by lane 0
template <typename T>
__device__ __forceinline__ u32 ld_uniform_u32(const T *p, int lane) {
    u32 v = 0;
    if (lane == 0) v = (u32)ld(p);
    return readlane_u32(v, 0);
}

// MetricResult order as a signed key (models/types.rs:401-411): a < b  <=>  order_key(a) < order_key(b)
__device__ __forceinline__ int32_t order_key(u32 metric, float v) {
    int32_t b = (int32_t)__float_as_uint(v);
    b ^= (int32_t)(((u32)(b >> 31)) >> 1);
    return (metric == 1u || metric == 2u) ? ~b : b;
}

// minimum over the wave of a u32, in the VALU (DPP) like group_reduce_add_u32: no ds_bpermute round trips.  The candidate loop of
// link_kernel is a dependent chain per wave, and its two `lowest` scans per candidate used to be 12 ds_bpermute each (~60 clocks a
// piece): more than the row loads they sit between.
__device__ __forceinline__ u32 wave_min_u32(u32 v) {
    u32 o;
    o = dpp_mov<0xB1>(0xFFFFFFFFu, v); v = o < v ? o : v;   // quad_perm [1,0,3,2]
    o = dpp_mov<0x4E>(0xFFFFFFFFu, v); v = o < v ? o : v;   // quad_perm [2,3,0,1]
    o = dpp_mov<0x141>(0xFFFFFFFFu, v); v = o < v ? o : v;  // row_half_mirror
    o = dpp_mov<0x140>(0xFFFFFFFFu, v); v = o < v ? o : v;  // row_mirror: every lane of a 16-lane row holds the row's minimum
    const u32 a = readlane_u32(v, 0), b = readlane_u32(v, 16), c = readlane_u32(v, 32), d = readlane_u32(v, 48);
    const u32 ab = a < b ? a : b, cd = c < d ? c : d;
    return ab < cd ? ab : cd;
}
// minimum of (hi, lo) pairs in lexicographic order = two u32 reductions (the second over the lanes that hold the minimal hi)
__device__ __forceinline__ u64 wave_min_u64(u64 v) {
    const u32 hi = (u32)(v >> 32), lo = (u32)v;
    const u32 mh = wave_min_u32(hi);
    const u32 ml = wave_min_u32(hi == mh ? lo : 0xFFFFFFFFu);
    return ((u64)mh << 32) | ml;
}

// One row of a level (M slots: slot j lives in lane j % 64, register j / 64) — the wave-level mirror of ProbNode.neighbors.
template <int SL>
struct Row {
    u32 nb[SL];
    int32_t nk[SL];
    __device__ __forceinline__ void load(const LinkLevelDev &lv, u32 node, int lane) {
#pragma unroll
        for (int s = 0; s < SL; s++) {
            const u32 j = (u32)lane + 64u * s;
            nb[s] = NONE;
            nk[s] = EMPTY_KEY;
            if (j < lv.M) {
                nb[s] = ld(&lv.adj_node[(u64)node * lv.M + j]);
                nk[s] = ld(&lv.key[(u64)node * lv.M + j]);
            }
        }
    }
    __device__ __forceinline__ void store_slot(const LinkLevelDev &lv, u32 node, u32 slot, int lane) const {
#pragma unroll
        for (int s = 0; s < SL; s++) {
            const u32 j = (u32)lane + 64u * s;
            if (j == slot) {
                const u64 o = (u64)node * lv.M + j;
                st(&lv.adj_node[o], nb[s]);
                if (lv.adj_vec != lv.adj_node) st(&lv.adj_vec[o], nb[s] == NONE ? NONE : lv.node_vec[nb[s]]);
                st(&lv.key[o], nk[s]);
            }
        }
    }
    __device__ __forceinline__ void store_all(const LinkLevelDev &lv, u32 node, int lane) const {
#pragma unroll
        for (int s = 0; s < SL; s++) {
            const u32 j = (u32)lane + 64u * s;
            if (j < lv.M) {
                const u64 o = (u64)node * lv.M + j;
                st(&lv.adj_node[o], nb[s]);
                if (lv.adj_vec != lv.adj_node) st(&lv.adj_vec[o], nb[s] == NONE ? NONE : lv.node_vec[nb[s]]);
                st(&lv.key[o], nk[s]);
            }
        }
    }
    // (neighbour, key) of one slot, wave-uniform
    __device__ __forceinline__ void get(u32 slot, u32 &n_out, int32_t &k_out) const {
        u32 n = NONE;
        int32_t k = EMPTY_KEY;
#pragma unroll
        for (int s = 0; s < SL; s++)
            if ((slot >> 6) == (u32)s) { n = nb[s]; k = nk[s]; }
        n_out = readlane_u32(n, (int)(slot & 63u)); // `slot` is wave-uniform (a cached lowest index / a returned slot)
        k_out = (int32_t)readlane_u32((u32)k, (int)(slot & 63u));
    }
    __device__ __forceinline__ void set(u32 slot, u32 n, int32_t k, int lane) {
#pragma unroll
        for (int s = 0; s < SL; s++)
            if ((u32)lane + 64u * s == slot) { nb[s] = n; nk[s] = k; }
    }
    // prob_node.rs:245-259: the scan starts from MetricResult::max; the first empty slot wins outright (its key is below every
    // real key), otherwise the first strictly-smallest similarity; no slot below max -> (0, max)
    __device__ __forceinline__ void lowest(u32 M, int32_t kmin, int32_t kmax, int lane, u32 &idx_out, int32_t &key_out) const {
        u64 best = ~0ull;
#pragma unroll
        for (int s = 0; s < SL; s++) {
            const u32 j = (u32)lane + 64u * s;
            if (j < M) {
                const u64 v = ((u64)((u32)nk[s] ^ 0x80000000u) << 32) | j; // signed order -> unsigned order; ties: smaller index
                best = v < best ? v : best;
            }
        }
        best = wave_min_u64(best);
        const int32_t mn = (int32_t)((u32)(best >> 32) ^ 0x80000000u);
        if (mn < kmax) { idx_out = (u32)best; key_out = mn == EMPTY_KEY ? kmin : mn; }
        else { idx_out = 0; key_out = kmax; }
    }
    // first slot holding `target` (remove_neighbor_by_id scans in slot order), or NONE
    __device__ __forceinline__ u32 find(u32 target, u32 M, int lane) const {
        u32 found = NONE;
#pragma unroll
        for (int s = SL - 1; s >= 0; s--) {
            const u32 j = (u32)lane + 64u * s;
            const u64 m = __ballot(j < M && nb[s] == target);
            if (m) found = (u32)(__ffsll((long long)m) - 1) + 64u * s;
        }
        return found;
    }
};

// ProbNode::add_neighbor on a row held in registers.  Returns the slot (or -1); `evicted` = the neighbour that was replaced.
// The caller owns the row's cached (lowest index, lowest key) and gets them updated.
template <int SL>
__device__ __forceinline__ int add_neighbor(Row<SL> &row, u32 M, int32_t kmin, int32_t kmax, u32 &low_idx, int32_t &low_key, u32 nbr, int32_t dist,
                                            int lane, u32 &evicted) {
    evicted = NONE;
    if (dist <= low_key) return -1;                                     // :222-225
    const u32 slot = low_idx;
    u32 old_n;
    int32_t old_k;
    row.get(slot, old_n, old_k);
    const bool ok = old_n == NONE || dist > old_k;                      // :227-243
    if (ok) row.set(slot, nbr, dist, lane);
    row.lowest(M, kmin, kmax, lane, low_idx, low_key);                  // :245-262 (refreshed whether or not the slot was taken)
    if (!ok) return -1;
    evicted = old_n;
    return (int)slot;
}

// ---- phase 1 of a round: claims ---------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void claim_kernel(const LinkArgs a, const u32 *__restrict__ pend, const u32 *__restrict__ pcount, u32 *next_count,
                                                    u32 *evq_count, u32 round) {
    const u32 count = *pcount;
    if (blockIdx.x == 0 && threadIdx.x == 0) { *next_count = 0; *evq_count = 0; } // consumed by the previous round; refilled by link_kernel
    const u32 per = KEEP_INDEX + 1;
    const u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    const u32 e_idx = (u32)(t / per), i = (u32)(t % per);
    if (e_idx >= count) return;
    const u32 e = pend[e_idx], level = e >> 16, b = e & 0xFFFFu;
    const LinkLevelDev &lv = a.lv[level];
    const u32 slot = a.L1 - 1 - level;
    u32 r;
    if (i == 0) r = a.me[(u64)b * a.L1 + level];
    else {
        if (i - 1 >= a.z_counts[(u64)b * a.L1 + slot]) return;
        r = a.z_nodes[((u64)b * a.L1 + slot) * KEEP_INDEX + (i - 1)];
    }
    atomicMax(&lv.owner[r], (round << 13) | (8191u - b));
}

// ---- phase 2: one wave per pending node; runnable nodes execute create_node_edges on their rows ---------------------
template <int SL>
__global__ __launch_bounds__(64) void link_kernel(const LinkArgs a, const u32 *__restrict__ pend, const u32 *__restrict__ pcount, u32 *__restrict__ next,
                                                  u32 *next_count, u32 *__restrict__ evq, u32 *evq_count, u32 round) {
    const int lane = threadIdx.x;
    if (blockIdx.x >= *pcount) return;
    const u32 e = pend[blockIdx.x], level = e >> 16, b = e & 0xFFFFu;
    const LinkLevelDev lv = a.lv[level];
    const u32 M = lv.M, slotz = a.L1 - 1 - level;
    const u32 node = a.me[(u64)b * a.L1 + level];
    const u32 cnt = a.z_counts[(u64)b * a.L1 + slotz];
    const u64 zb = ((u64)b * a.L1 + slotz) * KEEP_INDEX;
    const u32 cz = (u32)lane < cnt ? a.z_nodes[zb + lane] : NONE;
    const float cs = (u32)lane < cnt ? a.z_sims[zb + lane] : 0.0f;
    const u32 tag = (round << 13) | (8191u - b);
    bool mine = true;
    if ((u32)lane < cnt) mine = ld(&lv.owner[cz]) == tag;
    const bool runnable = __all(mine) && ld_uniform_u32(&lv.owner[node], lane) == tag;
    if (!runnable) { // blocked by an earlier node of the batch: next round
        if (lane == 0) next[atomicAdd(next_count, 1u)] = e;
        return;
    }

    // The candidate loop below is sequential by nature (create_node_edges), and every iteration used to start with DEPENDENT HBM
    // round trips: the candidate's cached (lowest index, lowest key), then its row — 64 candidates x 2 trips x ~900 cycles is what a
    // round cost (72-145 us per link_kernel call, rounds 1-2).  Everything those trips fetch is known up front:
    //   * lane i reads candidate i's cached lowest entry NOW (all 64 in flight together) and the loop takes it from the lane's
    //     registers — exact, because this wave owns row c for the round, a candidate appears once in the walk result, and the only
    //     writer of c's cache is the iteration that processes c (evictions do not refresh caches, prob_node.rs:285-306);
    //   * lane i touches both 128-byte lines of candidate i's adjacency row and key row, so the loop's row loads hit L2.
    u32 pre_low_idx = 0, warm = 0, pre_kind = 0;
    int32_t pre_low_key = 0;
    const u32 self_kind = lv.kind ? (u32)lv.kind[node] : 0u;
    if ((u32)lane < cnt) {
        if (lv.kind) pre_kind = (u32)lv.kind[cz];
        pre_low_idx = (u32)ld(&lv.low_idx[cz]);
        pre_low_key = (int32_t)ld(&lv.low_key[cz]);
        const u64 o = (u64)cz * M;
        warm = ld(&lv.adj_node[o]) ^ (u32)ld(&lv.key[o]);
        if (M > 32) warm ^= ld(&lv.adj_node[o + 32]) ^ (u32)ld(&lv.key[o + 32]);
    }
    Row<SL> self;
    self.load(lv, node, lane);
    u32 s_low_idx = ld_uniform_u32(&lv.low_idx[node], lane);
    int32_t s_low_key = (int32_t)ld_uniform_u32(&lv.low_key[node], lane);
    const u32 node_vec = lv.adj_vec != lv.adj_node ? lv.node_vec[node] : node;

    // evictee `old` drops its back edge to `target`: now if the row is in this node's claim, else at the end of the round.
    // Whether the row is ours needs no memory access: a runnable node owns exactly the rows it claimed — itself and its candidates —
    // so the test is a ballot over the candidate registers (it used to be a dependent load of owner[old]: an HBM miss per eviction,
    // and in a mature graph nearly every accepted edge evicts somebody).  Evictions outside the claim are parked in the lane of the
    // candidate that caused them (one for the node's own row, one for the candidate's row) and queued with ONE atomic at the end
    // (the queue counter used to be bumped, and waited for, once per eviction).
    u32 q_old[2] = {NONE, NONE}, q_tgt[2] = {NONE, NONE}; // lane i: evictions caused while processing candidate i
    auto drop_back_edge = [&](u32 old, u32 target, u32 i, int which) {
        const bool own = old == node || __any((u32)lane < cnt && cz == old);
        if (own) {
            Row<SL> r;
            r.load(lv, old, lane);
            const u32 j = r.find(target, M, lane);
            if (j != NONE) { r.set(j, NONE, EMPTY_KEY, lane); r.store_slot(lv, old, j, lane); }
        } else if ((u32)lane == i) {
            q_old[which] = old;
            q_tgt[which] = target;
        }
    };

    u32 succ = 0;
    for (u32 i = 0; i < cnt; i++) { // walk results in descending order until M edges succeeded (vector_store.rs:995-1074)
        if (succ >= M) break;
        const u32 c = readlane_u32(cz, (int)i);
        const float csim = __uint_as_float(readlane_u32(__float_as_uint(cs), (int)i));
        const int32_t dk = order_key(a.metric, csim);
        if (lv.kind && a.metric == 0u) { // pseudo-root component, cosine: edges the reference refuses outright (vector_store.rs:1017-1041)
            const u32 nk = readlane_u32(pre_kind, (int)i);
            if (nk == 1u && self_kind == 2u && csim != 1.0f) continue;  // a Metadata replica links to a pseudo node only on an exact match
            if (nk == 2u && self_kind == 2u && csim == -1.0f) continue; // two Metadata replicas whose metadata dimensions disagree
        }
        u32 ev;
        const int r = add_neighbor<SL>(self, M, a.kmin, a.kmax, s_low_idx, s_low_key, c, dk, lane, ev);
        if (ev != NONE) drop_back_edge(ev, node, i, 0);
        if (r < 0) continue;
        // the back edge on the candidate's row (its cached lowest entry was read up front, see above)
        u32 c_low_idx = readlane_u32(pre_low_idx, (int)i);
        int32_t c_low_key = (int32_t)readlane_u32((u32)pre_low_key, (int)i);
        int r2 = -1;
        if (dk > c_low_key) {
            Row<SL> cr;
            cr.load(lv, c, lane);
            r2 = add_neighbor<SL>(cr, M, a.kmin, a.kmax, c_low_idx, c_low_key, node, dk, lane, ev);
            if (r2 >= 0) { // the slot written holds `node`: derive its vector row here (node_vec of a level >= 1)
#pragma unroll
                for (int s = 0; s < SL; s++) {
                    const u32 j = (u32)lane + 64u * s;
                    if (j == (u32)r2) {
                        const u64 o = (u64)c * M + j;
                        st(&lv.adj_node[o], node);
                        if (lv.adj_vec != lv.adj_node) st(&lv.adj_vec[o], node_vec);
                        st(&lv.key[o], dk);
                    }
                }
            }
            if (lane == 0) { st(&lv.low_idx[c], (uint8_t)c_low_idx); st(&lv.low_key[c], c_low_key); }
            if (ev != NONE) drop_back_edge(ev, c, i, 1);
        }
        if (r2 >= 0) succ++;
        else { // remove_neighbor_by_index_and_id: the forward half is rolled back if the slot still holds the candidate
            u32 cur_n;
            int32_t cur_k;
            self.get((u32)r, cur_n, cur_k);
            if (cur_n == c) self.set((u32)r, NONE, EMPTY_KEY, lane);
        }
    }
    self.store_all(lv, node, lane);
    if (lane == 0) { st(&lv.low_idx[node], (uint8_t)s_low_idx); st(&lv.low_key[node], s_low_key); }
    { // queue the parked evictions: one counter bump per wave
        const u64 m0 = __ballot(q_old[0] != NONE), m1 = __ballot(q_old[1] != NONE);
        const u32 n0 = (u32)__popcll(m0), n1 = (u32)__popcll(m1);
        if (n0 + n1) {
            u32 base = 0;
            if (lane == 0) base = atomicAdd(evq_count, n0 + n1);
            base = readlane_u32(base, 0);
            const u64 lt = (1ull << lane) - 1ull;
            if (q_old[0] != NONE) {
                const u64 q = (u64)base + (u32)__popcll(m0 & lt);
                evq[3 * q + 0] = level; evq[3 * q + 1] = q_old[0]; evq[3 * q + 2] = q_tgt[0];
            }
            if (q_old[1] != NONE) {
                const u64 q = (u64)base + n0 + (u32)__popcll(m1 & lt);
                evq[3 * q + 0] = level; evq[3 * q + 1] = q_old[1]; evq[3 * q + 2] = q_tgt[1];
            }
        }
    }
    if (warm == 0x9E3779B9u && round == 0xFFFFFFFFu) next[0] = warm; // never true (rounds stay below 2^19): keeps the warm-up loads alive
}

// ---- phase 3: evictees outside the evictor's claim drop their back edges ---------------------------------------------
__global__ __launch_bounds__(256) void evict_kernel(const LinkArgs a, const u32 *__restrict__ evq, const u32 *__restrict__ evq_count) {
    const u32 n = *evq_count;
    for (u64 q = (u64)blockIdx.x * blockDim.x + threadIdx.x; q < n; q += (u64)gridDim.x * blockDim.x) {
        const u32 level = evq[3 * q], old = evq[3 * q + 1], target = evq[3 * q + 2];
        const LinkLevelDev &lv = a.lv[level];
        for (u32 j = 0; j < lv.M; j++) {
            const u64 o = (u64)old * lv.M + j;
            if (lv.adj_node[o] == target) {
                lv.adj_node[o] = NONE;
                if (lv.adj_vec != lv.adj_node) lv.adj_vec[o] = NONE;
                lv.key[o] = EMPTY_KEY;
                break;
            }
        }
    }
}

__global__ void fill_i32_kernel(int32_t *p, u64 n, int32_t v) {
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (u64)gridDim.x * blockDim.x) p[i] = v;
}

// cos_index_delete (builder.hip; delete_embedding, vector_store.rs:1277-1369): one wave per level; dn[level] = the deleted vector's node on
// that level (NONE: the walk did not reach it / it has no node there).  Every neighbour of the node drops its back edge
// (remove_neighbor_by_id, prob_node.rs:285-306: the first slot holding it; the slot's key goes, the neighbour's lowest cache stays) and
// the node's own slots are emptied.  The removals touch different rows and commute — unless a neighbour would be left with NO
// neighbour at all: the reference links such a node again, in slot order, between the removals (:1305-1357).  The kernel looks
// first and, if it finds one, changes nothing and says so (status 1): the host then runs the level in the reference's order.
__global__ __launch_bounds__(64) void unlink_kernel(const LinkArgs a, const u32 *__restrict__ dn, u32 *__restrict__ status) {
    const u32 level = blockIdx.x;
    const int lane = threadIdx.x;
    const u32 node = dn[level];
    if (node == NONE) { if (lane == 0) status[level] = 0; return; }
    const LinkLevelDev &lv = a.lv[level];
    const u32 M = lv.M;
    bool orphan = false;
    for (u32 s = lane; s < M; s += 64) {
        const u32 x = ld(&lv.adj_node[(u64)node * M + s]);
        if (x == NONE) continue;
        bool found = false, other = false;
        for (u32 j = 0; j < M; j++) {
            const u32 v = ld(&lv.adj_node[(u64)x * M + j]);
            if (v == node && !found) found = true;
            else if (v != NONE) other = true;
        }
        orphan = orphan || (found && !other);
    }
    if (__builtin_amdgcn_ballot_w64(orphan)) { if (lane == 0) status[level] = 1; return; }
    for (u32 s = lane; s < M; s += 64) {
        const u64 o = (u64)node * M + s;
        const u32 x = ld(&lv.adj_node[o]);
        if (x == NONE) continue;
        for (u32 j = 0; j < M; j++) {
            const u64 ox = (u64)x * M + j;
            if (ld(&lv.adj_node[ox]) == node) {
                st(&lv.adj_node[ox], NONE);
                if (lv.adj_vec != lv.adj_node) st(&lv.adj_vec[ox], NONE);
                st(&lv.key[ox], EMPTY_KEY);
                break;
            }
        }
        st(&lv.adj_node[o], NONE);
        if (lv.adj_vec != lv.adj_node) st(&lv.adj_vec[o], NONE);
        st(&lv.key[o], EMPTY_KEY);
    }
    if (lane == 0) status[level] = 2;
}

// cos_index_restore_link_state (builder.hip): the link state of an UPLOADED graph, as the reference has it after a reload.
//   edge_pairs_kernel: (node's vector row, neighbour's vector row) of every slot of a chunk of nodes — an empty slot pairs the node with
//                      itself (a row that is hot anyway; its result is never looked at) — for the distance operator's kernel;
//   edge_keys_kernel:  slot keys from the similarities (empty: EMPTY_KEY); a failed distance of a real slot raises the flag;
//   low_cache_kernel:  ProbNode::new_with_neighbors_and_versions (prob_node.rs:145-181): the first empty slot with MetricResult::min,
//                      else the first strictly smallest similarity.
__global__ void edge_pairs_kernel(const u32 *__restrict__ adj_vec, const u32 *__restrict__ node_vec, u32 node0, u64 total, u32 M, u32 *__restrict__ px, u32 *__restrict__ py) {
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (u64)gridDim.x * blockDim.x) {
        const u32 node = node0 + (u32)(i / M);
        const u32 self = node_vec ? node_vec[node] : node;
        const u32 v = adj_vec[(u64)node0 * M + i];
        px[i] = self;
        py[i] = v == NONE ? self : v;
    }
}
__global__ void edge_keys_kernel(const u32 *__restrict__ adj_vec, const float *__restrict__ sims, const int32_t *__restrict__ st_in, u32 node0, u64 total, u32 M, u32 metric,
                                 int32_t *__restrict__ key, int32_t *__restrict__ fail) {
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (u64)gridDim.x * blockDim.x) {
        const u64 o = (u64)node0 * M + i;
        const bool empty = adj_vec[o] == NONE;
        if (!empty && st_in[i] != 0) atomicMax(fail, st_in[i]);
        key[o] = empty ? EMPTY_KEY : order_key(metric, sims[i]);
    }
}
__global__ void low_cache_kernel(const u32 *__restrict__ adj_vec, const int32_t *__restrict__ key, u32 n, u32 M, int32_t kmin, int32_t kmax, uint8_t *__restrict__ low_idx,
                                 int32_t *__restrict__ low_key) {
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (u64)gridDim.x * blockDim.x) {
        u32 nl = 0;
        int32_t nk = kmax;
        for (u32 j = 0; j < M; j++) {
            if (adj_vec[i * M + j] == NONE) { nl = j; nk = kmin; break; }
            const int32_t k = key[i * M + j];
            if (k < nk) { nk = k; nl = j; }
        }
        low_idx[i] = (uint8_t)nl;
        low_key[i] = nk;
    }
}

// cos_index_append (builder.hip): a level's [n][M] array grows.  Rows [0, old_n - 1) keep their place, the LAST old row (the root) becomes
// the last new row, the rows between (the new nodes) are filled; a value equal to remap_from (a reference to the root) becomes remap_to.
__global__ void grow_rows_kernel(const u32 *__restrict__ src, u32 *__restrict__ dst, u32 old_n, u32 new_n, u32 M, u32 remap_from, u32 remap_to, u32 fill) {
    const u64 total = (u64)new_n * M;
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (u64)gridDim.x * blockDim.x) {
        const u64 row = i / M;
        const u32 col = (u32)(i - row * M);
        u32 v = fill;
        if (row + 1 < old_n) v = src[row * M + col];
        else if (row + 1 == new_n) v = src[(u64)(old_n - 1) * M + col];
        else { dst[i] = fill; continue; }
        dst[i] = v == remap_from ? remap_to : v;
    }
}
__global__ void grow_bytes_kernel(const uint8_t *__restrict__ src, uint8_t *__restrict__ dst, u32 old_n, u32 new_n) {
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < new_n; i += (u64)gridDim.x * blockDim.x)
        dst[i] = i + 1 < old_n ? src[i] : (i + 1 == new_n ? src[old_n - 1] : (uint8_t)0);
}

// LevelDev::adj_mag (engine_types.h): the norm of every scanned neighbour slot, next to the adjacency.  slots is a power of two
// (min of two powers of two) or the shortlist size: plain division.
__global__ void fill_adj_mag_kernel(const u32 *__restrict__ adj_vec, const float *__restrict__ mags, float *__restrict__ adj_mag, u64 total, u32 M, u32 slots) {
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (u64)gridDim.x * blockDim.x) {
        const u64 node = i / slots;
        const u32 slot = (u32)(i - node * slots);
        const u32 v = adj_vec[node * M + slot];
        adj_mag[i] = v == NONE ? 1.0f : mags[v];
    }
}

} // namespace

namespace cosdev {

hipError_t launch_fill_i32(int32_t *p, u64 n, int32_t v, hipStream_t st) {
    if (n == 0) return hipSuccess;
    const u32 blocks = (u32)std::min<u64>((n + 255) / 256, 65535);
    hipLaunchKernelGGL(fill_i32_kernel, dim3(blocks), dim3(256), 0, st, p, n, v);
    return hipGetLastError();
}

hipError_t launch_edge_pairs(const u32 *adj_vec, const u32 *node_vec, u32 node0, u32 cn, u32 M, u32 *px, u32 *py, hipStream_t st) {
    const u64 total = (u64)cn * M;
    if (!total) return hipSuccess;
    hipLaunchKernelGGL(edge_pairs_kernel, dim3((u32)std::min<u64>((total + 255) / 256, 1u << 20)), dim3(256), 0, st, adj_vec, node_vec, node0, total, M, px, py);
    return hipGetLastError();
}
hipError_t launch_edge_keys(const u32 *adj_vec, const float *sims, const int32_t *st_in, u32 node0, u32 cn, u32 M, u32 metric, int32_t *key, int32_t *fail, hipStream_t st) {
    const u64 total = (u64)cn * M;
    if (!total) return hipSuccess;
    hipLaunchKernelGGL(edge_keys_kernel, dim3((u32)std::min<u64>((total + 255) / 256, 1u << 20)), dim3(256), 0, st, adj_vec, sims, st_in, node0, total, M, metric, key, fail);
    return hipGetLastError();
}
hipError_t launch_low_cache(const u32 *adj_vec, const int32_t *key, u32 n, u32 M, int32_t kmin, int32_t kmax, uint8_t *low_idx, int32_t *low_key, hipStream_t st) {
    if (!n) return hipSuccess;
    hipLaunchKernelGGL(low_cache_kernel, dim3((u32)std::min<u64>(((u64)n + 255) / 256, 1u << 20)), dim3(256), 0, st, adj_vec, key, n, M, kmin, kmax, low_idx, low_key);
    return hipGetLastError();
}
hipError_t launch_unlink(const LinkArgs &a, const u32 *dn, u32 *status, hipStream_t st) {
    hipLaunchKernelGGL(unlink_kernel, dim3(a.L1), dim3(64), 0, st, a, dn, status);
    return hipGetLastError();
}
hipError_t launch_grow_rows(const u32 *src, u32 *dst, u32 old_n, u32 new_n, u32 M, u32 remap_from, u32 remap_to, u32 fill, hipStream_t st) {
    const u64 total = (u64)new_n * M;
    if (total == 0 || old_n == 0) return hipErrorInvalidValue;
    const u32 blocks = (u32)std::min<u64>((total + 255) / 256, 1u << 20);
    hipLaunchKernelGGL(grow_rows_kernel, dim3(blocks), dim3(256), 0, st, src, dst, old_n, new_n, M, remap_from, remap_to, fill);
    return hipGetLastError();
}
hipError_t launch_grow_bytes(const uint8_t *src, uint8_t *dst, u32 old_n, u32 new_n, hipStream_t st) {
    if (new_n == 0 || old_n == 0) return hipErrorInvalidValue;
    const u32 blocks = (u32)std::min<u64>(((u64)new_n + 255) / 256, 1u << 20);
    hipLaunchKernelGGL(grow_bytes_kernel, dim3(blocks), dim3(256), 0, st, src, dst, old_n, new_n);
    return hipGetLastError();
}

hipError_t launch_fill_adj_mag(const u32 *adj_vec, const float *mags, float *adj_mag, u32 n, u32 M, u32 slots, hipStream_t st) {
    const u64 total = (u64)n * slots;
    if (total == 0) return hipSuccess;
    const u32 blocks = (u32)std::min<u64>((total + 255) / 256, 1u << 20);
    hipLaunchKernelGGL(fill_adj_mag_kernel, dim3(blocks), dim3(256), 0, st, adj_vec, mags, adj_mag, total, M, slots);
    return hipGetLastError();
}

// one round over at most `count_ub` pending entries (the true count is read on the device from pcount)
hipError_t launch_link_round(const LinkArgs &a, u32 maxM, const u32 *pend, const u32 *pcount, u32 *next, u32 *next_count, u32 *evq, u32 *evq_count,
                             u32 count_ub, u32 round, hipStream_t st) {
    if (count_ub == 0) return hipSuccess;
    const u64 threads = (u64)count_ub * (KEEP_INDEX + 1);
    hipLaunchKernelGGL(claim_kernel, dim3((u32)((threads + 255) / 256)), dim3(256), 0, st, a, pend, pcount, next_count, evq_count, round);
    if (maxM <= 64) hipLaunchKernelGGL(link_kernel<1>, dim3(count_ub), dim3(64), 0, st, a, pend, pcount, next, next_count, evq, evq_count, round);
    else if (maxM <= 128) hipLaunchKernelGGL(link_kernel<2>, dim3(count_ub), dim3(64), 0, st, a, pend, pcount, next, next_count, evq, evq_count, round);
    else hipLaunchKernelGGL(link_kernel<4>, dim3(count_ub), dim3(64), 0, st, a, pend, pcount, next, next_count, evq, evq_count, round);
    const u32 eblocks = std::min<u32>(1024u, std::max<u32>(1u, count_ub / 2u));
    hipLaunchKernelGGL(evict_kernel, dim3(eblocks), dim3(256), 0, st, a, evq, evq_count);
    return hipGetLastError();
}

} // namespace cosdev
