// kernels_meta.hip — metadata-filtered search on the device.
//   search_internal with a Filter: start at the pseudo root             indexes/hnsw/mod.rs:413-423
//   ann_search with query_filter_dims                                   vector_store.rs:256-402
//        per level: ONE PerformantFixedSet shared by the walks of all QueryFilterDimensions (:266-271), one
//        traverse_find_nearest per filter (:278-291), results with cosine exactly -1.0 dropped (:293-303), sorted descending,
//        first 100 kept (:307-311); empty -> the entry node with its strongest match over the filters (:329-367)
//   CosineSimilarity::calculate, (node kind, query kind) arms           distance/cosine.rs:36-102
//   cosine_similarity_mdims                                             distance/cosine.rs:243-262
//   VectorData::replica_node_kind                                       models/types.rs:219-241
// The component's nodes are REPLICAS: node -> (internal id, vector row, metadata row) through LevelDev::node_id / node_vec /
// node_meta.  One wavefront owns one query; 64 lanes <-> the <= 64 neighbour slots of an expansion, like walk_kernel, but
// without its lookahead window and scalar-row fast path: this path is about parity (HBM-bound gather like the plain walk; the
// filter arithmetic is a few dozen integer compares per neighbour).  oracle/cosdata_oracle_hnsw.c (ann_search_filtered) is the
// CPU statement; tests/test_gpu_meta.py asserts identical per-level lists and final results.
#include <hip/hip_runtime.h>

#include <algorithm>

#include "dot_engines.h"
#include "engine_types.h"

using namespace cosdev;

#define COS_OK 0
#define COS_ERR_CALCULATION 2
#define COS_ERR_UNIMPLEMENTED 4
#define COS_QUERY_ID 0xFFFFFFFEu

namespace {

constexpr u32 PSEUDO_LO = 0xFFFFFEFEu, PSEUDO_HI = 0xFFFFFFFDu; // u32::MAX - 257 ..= u32::MAX - 2 (types.rs:229)
constexpr int PBM = 4;                                          // code rows in flight per lane group
enum : int { KIND_BASE = 0, KIND_PSEUDO = 1, KIND_METADATA = 2 };

// dot_product_f32 in the order of dot_product_f32_simd (x86_64.rs:418-444) by ONE lane: 8 FMA chains over chunks of 8,
// ((s0+s1)+(s2+s3)) + ((s4+s5)+(s6+s7)), non-fused scalar tail.  a: query metadata dims (LDS), b: node mbits (global).
__device__ __forceinline__ float mdims_dot_ref(const float *__restrict__ a, const int32_t *__restrict__ b, u32 n) {
    float s[8];
#pragma unroll
    for (int i = 0; i < 8; i++) s[i] = 0.0f;
    const u32 chunks = n >> 3;
    for (u32 c = 0; c < chunks; c++)
#pragma unroll
        for (int i = 0; i < 8; i++) s[i] = __fmaf_rn(a[c * 8 + i], (float)b[c * 8 + i], s[i]);
    float r = __fadd_rn(__fadd_rn(__fadd_rn(s[0], s[1]), __fadd_rn(s[2], s[3])), __fadd_rn(__fadd_rn(s[4], s[5]), __fadd_rn(s[6], s[7])));
    for (u32 i = chunks * 8; i < n; i++) r = __fadd_rn(r, __fmul_rn(a[i], (float)b[i]));
    return r;
}

struct MetaSmem {
    u32 *vis;     // PerformantFixedSet replica of the level (shared by the level's filter walks)
    u64 *res;     // popped (key, node) list of the current walk, ef entries
    u64 *acc;     // merged candidates of the level so far (<= 100, sorted descending)
    u32 *wl_vec;  // winners that need the vector cosine: rows
    u32 *wl_node;
    float *fq;    // current filter's dimensions as f32 [mdim]
    int32_t *fqi; //                             as i32 [mdim]
    float *qf;    // float engines: the query vector
    u32 *win;     // lookahead window: [5][LA_M][64] neighbour node | its id | metadata row | vector row | metadata norm (as bits)
};
constexpr int LA_M = 4; // pool entries whose adjacency rows (and what they name) are fetched together

// INDEXING = the walk of index_embedding for a node of the component itself (vector_store.rs:484-640): the "query" is a pseudo node
// or a Metadata replica — its vector row is wa.q_rows[b], its id wa.self_ids[b] (pre-inserted in the visited filter, :807; the id
// also says whether it is a pseudo node), its own metadata dimensions are the ONE filter of the query — the list is cut to wa.keep
// (64), nothing is dropped for being -1.0, and the node indices are returned for the link kernels (wa.out_nodes).
template <int ENG, int CH, int R, bool INDEXING>
__global__ __launch_bounds__(64) void walk_meta_kernel(const IndexDev ix, const WalkArgs wa) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int lane = threadIdx.x;
    const u32 qi = blockIdx.x;
    if (qi >= wa.B) return;
    const u32 L = ix.num_layers, metric = ix.metric, md = ix.mdim;
    const u32 Mmax = ix.lv[0].M > ix.lv[L].M ? ix.lv[0].M : ix.lv[L].M;
    MetaSmem sm;
    {
        unsigned char *p = smem_raw;
        sm.vis = (u32 *)p;       p += (size_t)Mmax * 8;
        sm.res = (u64 *)p;       p += (size_t)(wa.ef < 128u ? 128u : wa.ef) * 8; // also the 128-entry staging list of the merge
        sm.acc = (u64 *)p;       p += 128 * 8;
        sm.wl_vec = (u32 *)p;    p += 64 * 4;
        sm.wl_node = (u32 *)p;   p += 64 * 4;
        sm.fq = (float *)p;      p += 64 * 4;
        sm.fqi = (int32_t *)p;   p += 64 * 4;
        sm.win = (u32 *)p;       p += (size_t)5 * LA_M * 64 * 4;
        p = (unsigned char *)(((size_t)p + 15) & ~(size_t)15);
        sm.qf = (float *)p;
    }
    const u32 qrow = (INDEXING && wa.q_rows) ? wa.q_rows[qi] : qi;
    const u32 self_id = INDEXING ? wa.self_ids[qi] : COS_QUERY_ID;
    const bool self_pseudo = INDEXING && self_id >= PSEUDO_LO && self_id <= PSEUDO_HI;
    const u32 keep_n = INDEXING ? wa.keep : (u32)KEEP_SEARCH;
    const uint8_t *qcode = wa.qcodes + (u64)qrow * ix.row_stride;
    const float qmag = wa.qmags[qrow];
    const u32 f0 = wa.f_off[qi], f1 = wa.f_off[qi + 1];

    constexpr bool FLOAT_ENG = ENG == ENG_F32 || ENG == ENG_F16;
    const int G = (ENG == ENG_F32) ? 8 : (ENG == ENG_F16 ? 1 : (int)ix.G); // f32: eight lanes per row (f32_oct_dot)
    const int lig = lane & (G - 1), grp = lane / G, RP = 64 / G;
    uint4 qreg[CH];
    if constexpr (!FLOAT_ENG) {
#pragma unroll
        for (int c = 0; c < CH; c++) {
            const u32 chunk = (u32)lig + (u32)c * (u32)G;
            qreg[c] = chunk < ix.nchunks ? *(const uint4 *)(qcode + (u64)chunk * 16) : make_uint4(0, 0, 0, 0);
        }
    } else if constexpr (ENG == ENG_F16) {
        const __half *qh = (const __half *)qcode;
        for (u32 i = lane; i < ix.dim; i += 64) sm.qf[i] = __half2float(qh[i]);
#pragma unroll
        for (int c = 0; c < CH; c++) qreg[c] = make_uint4(0, 0, 0, 0);
    } else {
        const float *qg = (const float *)qcode;
        for (u32 i = lane; i < (u32)(ix.row_stride / 4); i += 64) sm.qf[i] = qg[i];
#pragma unroll
        for (int c = 0; c < CH; c++) qreg[c] = make_uint4(0, 0, 0, 0);
    }
    __builtin_amdgcn_s_waitcnt(0);

    int32_t status = COS_OK;
    u32 entry = ix.lv[L].root_idx;

    // vector cosine / dot of ONE row by lane group 0 (entry node); valid in every lane
    auto vec_distance_single = [&](u32 row, float &sim_out) -> bool {
        float dotf;
        if constexpr (!FLOAT_ENG) {
            u32 a = 0;
            if (grp == 0) {
#pragma unroll
                for (int c = 0; c < CH; c++) {
                    const u32 chunk = (u32)lig + (u32)c * (u32)G;
                    if (chunk < ix.nchunks) a = chunk_dot<ENG>(qreg[c], *(const uint4 *)(ix.codes + (u64)row * ix.row_stride + (u64)chunk * 16), a);
                }
            }
            a = group_reduce_add_u32(a, G);
            dotf = (float)readlane_u32(a, 0);
        } else if constexpr (ENG == ENG_F16) {
            dotf = __uint_as_float(readlane_u32(__float_as_uint(f16_lane_dot(ix.codes + (u64)row * ix.row_stride, sm.qf, ix.dim)), 0));
        } else {
            dotf = __uint_as_float(readlane_u32(__float_as_uint(f32_oct_dot((const float *)(ix.codes + (u64)row * ix.row_stride), sm.qf, ix.dim, lane & 7)), 0));
        }
        if (metric == 0u) {
            const float den = __fmul_rn(qmag, ix.mags[row]);
            if (den == 0.0f) return false;
            sim_out = __fdiv_rn(dotf, den);
        } else
            sim_out = dotf;
        return true;
    };

    // CosineSimilarity::calculate's dispatch for (stored node, query with the filter in sm.fq / sm.fqi):
    // returns 0 = the similarity is `sim`, 1 = the vector cosine decides, < 0 = -status.  Executed per LANE (its own node).
    // (k = the node's metadata row, ymag = its norm: loaded by the caller, TOGETHER with the node's id and vector row — until round 6
    // they were loaded here, one dependent trip to memory after the other: seven per expansion)
    auto meta_decide = [&](u32 k, float ymag, u32 nid, float fmag, int xkind, float &sim) -> int {
        if (metric != 0u) return 1; // the other metrics never look at node kinds
        const int ykind = ymag == 0.0f ? KIND_BASE : ((nid >= PSEUDO_LO && nid <= PSEUDO_HI) ? KIND_PSEUDO : KIND_METADATA);
        const int32_t *yb = ix.mbits + (u64)k * md;
        if (ykind == KIND_PSEUDO && xkind == KIND_METADATA) { // a metadata query strongly matches / mismatches a pseudo node
            bool eq = true;
            for (u32 j = 0; j < md; j++) eq &= yb[j] == sm.fqi[j];
            sim = eq ? 1.0f : -1.0f;
            return 0;
        }
        if (ykind == KIND_PSEUDO && xkind == KIND_PSEUDO) { // two pseudo nodes (index time only): cosine_similarity_mdims (cosine.rs:243-262)
            const float dp = mdims_dot_ref(sm.fq, yb, md);
            const float den = __fmul_rn(fmag, ymag);
            if (den == 0.0f) return -COS_ERR_CALCULATION;
            sim = __fdiv_rn(dp, den);
            return 0;
        }
        if (ykind == KIND_BASE && xkind == KIND_BASE) return 1;
        if (ykind == KIND_METADATA && xkind == KIND_METADATA) {
            const float dp = mdims_dot_ref(sm.fq, yb, md);
            const float den = __fmul_rn(fmag, ymag);
            if (den == 0.0f) return -COS_ERR_CALCULATION;
            if (__fdiv_rn(dp, den) > 0.99f) return 1;
            sim = -1.0f;
            return 0;
        }
        if (ykind == KIND_BASE && xkind == KIND_METADATA) { sim = 0.0f; return 0; }
        return -COS_ERR_UNIMPLEMENTED; // the reference's unreachable!() arms (a Base-kind query under the pseudo root, ...)
    };

    for (int level = (int)L; level >= 0 && status == COS_OK; level--) {
        const LevelDev lv = ix.lv[level];
        const u32 M = lv.M;
        const u32 slots = M < ix.shortlist ? M : ix.shortlist;
        const u32 bitmask = 64u * M - 1u;
        const u32 out_slot = L - (u32)level;
        for (u32 w = lane; w < 2 * M; w += 64) sm.vis[w] = 0;
        if (lane == 0) { const u32 b = self_id & bitmask; sm.vis[b >> 5] |= 1u << (b & 31); }
        u32 nacc = 0; // merged candidates so far (sm.acc)

        for (u32 f = f0; f < f1 && status == COS_OK; f++) {
            // this filter's dimensions -> LDS
            for (u32 j = lane; j < md; j += 64) {
                const int32_t v = wa.f_dims[(u64)f * md + j];
                sm.fqi[j] = v;
                sm.fq[j] = (float)v;
            }
            __builtin_amdgcn_s_waitcnt(0);
            __builtin_amdgcn_wave_barrier();
            const float fmag = wa.f_mags[f];
            // a query has no id (types.rs:227-236); a node being indexed has one: VectorData::replica_node_kind (types.rs:219-241)
            const int xkind = fmag == 0.0f ? KIND_BASE : (self_pseudo ? KIND_PSEUDO : KIND_METADATA);

            Pool<R> pool;
            pool.clear();
            u32 npool = 0, npop = 0;
            { // start node: evaluated and pushed whatever the visited filter says (vector_store.rs:1144-1148)
                const u32 eid = lv.node_id[entry], ek = lv.node_meta[entry];
                const float eymag = ix.mmags[ek];
                float s0 = 0.0f;
                int dec = 0;
                if (lane == 0) dec = meta_decide(ek, eymag, eid, fmag, xkind, s0);
                dec = (int)readlane_u32((u32)dec, 0);
                s0 = __uint_as_float(readlane_u32(__float_as_uint(s0), 0));
                if (dec < 0) { status = -dec; break; }
                if (dec == 1 && !vec_distance_single(lv.node_vec[entry], s0)) { status = COS_ERR_CALCULATION; break; }
                if (lane == 0) { const u32 b = eid & bitmask; sm.vis[b >> 5] |= 1u << (b & 31); }
                pool.insert_at(pack_key(metric_key(metric, s0), entry), 0, lane);
                npool = 1;
            }
            bool failed = false;
            // Lookahead window (round 6, as in walk_kernel.inc): the adjacency rows of the next LA_M pool entries are fetched together,
            // then — together again — what every neighbour they name needs (id, metadata row, vector row; then the metadata norm): three
            // trips to memory per WINDOW where an expansion made them one after the other.  Entry i + 1 of the window is consumed only
            // while it is still provably the next pop, i.e. while no candidate has been inserted ahead of a waiting entry.  A pool
            // position past the window holds a real node or the empty key (node 0): its row is fetched and never looked at.
            while (npool > 0 && npop < wa.ef && !failed) {
                u32 kwin = npool < (u32)LA_M ? npool : (u32)LA_M;
                if (kwin > wa.ef - npop) kwin = wa.ef - npop;
                {
                    const u32 slot_l = (u32)lane < slots ? (u32)lane : slots - 1u;
                    u32 wnb[LA_M], wid[LA_M], wk[LA_M], wvr[LA_M];
                    float wym[LA_M];
                    static_for<0, LA_M>([&](auto ic) {
                        constexpr int i = decltype(ic)::value;
                        const u32 wnd = pool.template peek_node<i>();
                        wnb[i] = lv.adj_node[(u64)wnd * M + slot_l];
                    });
                    static_for<0, LA_M>([&](auto ic) {
                        constexpr int i = decltype(ic)::value;
                        if ((u32)lane >= slots) wnb[i] = ROW_EMPTY;
                        const u32 nbs = wnb[i] != ROW_EMPTY ? wnb[i] : 0u; // no lane masked off: an empty slot reads node 0
                        wid[i] = lv.node_id[nbs];
                        wk[i] = lv.node_meta[nbs];
                        wvr[i] = lv.node_vec[nbs];
                    });
                    static_for<0, LA_M>([&](auto ic) {
                        constexpr int i = decltype(ic)::value;
                        wym[i] = ix.mmags[wk[i]];
                    });
                    static_for<0, LA_M>([&](auto ic) {
                        constexpr int i = decltype(ic)::value;
                        sm.win[(0 * LA_M + i) * 64 + lane] = wnb[i];
                        sm.win[(1 * LA_M + i) * 64 + lane] = wid[i];
                        sm.win[(2 * LA_M + i) * 64 + lane] = wk[i];
                        sm.win[(3 * LA_M + i) * 64 + lane] = wvr[i];
                        sm.win[(4 * LA_M + i) * 64 + lane] = __float_as_uint(wym[i]);
                    });
                }
                bool window_ok = true;
                for (u32 wi = 0; wi < kwin && window_ok && !failed; wi++) {
                const u64 cur = pool.head(); // the window entry being consumed IS the pool's head
                pool.pop_head(lane);
                npool--;
                if (lane == 0) sm.res[npop] = cur;
                npop++;
                const int limit = (int)wa.ef - (int)npop;
                const int ahead = (int)kwin - 1 - (int)wi; // window entries still waiting at pool positions 0 .. ahead - 1
                const u32 nb_node = sm.win[(0 * LA_M + wi) * 64 + lane];
                const bool valid = nb_node != ROW_EMPTY;
                const u32 nid_l = sm.win[(1 * LA_M + wi) * 64 + lane], nk = sm.win[(2 * LA_M + wi) * 64 + lane], nvrow = sm.win[(3 * LA_M + wi) * 64 + lane];
                const float nymag = __uint_as_float(sm.win[(4 * LA_M + wi) * 64 + lane]);
                const u32 nid = valid ? nid_l : 0u;
                const u32 bit = nid & bitmask, word = bit >> 5, msk = 1u << (bit & 31);
                const bool pre = valid && (sm.vis[word] & msk);
                const bool cand = valid && !pre;
                if (!__any(cand)) continue;
                u32 old = 0;
                if (cand) old = atomicOr(&sm.vis[word], msk);
                const bool lost = cand && (old & msk);
                bool win = cand && !lost;
                u64 lostmask = __ballot(lost);
                while (lostmask) { // two slots of this expansion alias the same residue: the LOWER slot wins (sequential scan order)
                    const int l = __ffsll((long long)lostmask) - 1;
                    const u32 b = readlane_u32(bit, l);
                    const u64 g = __ballot(cand && bit == b);
                    const int w = __ffsll((long long)g) - 1;
                    if (cand && bit == b) win = (lane == w);
                    lostmask &= ~g;
                }
                // node-kind dispatch per winner
                float csim = 0.0f;
                int dec = 0;
                if (win) dec = meta_decide(nk, nymag, nid, fmag, xkind, csim);
                const u64 errm = __ballot(win && dec < 0);
                if (errm) { status = -(int)readlane_u32((u32)dec, __ffsll((long long)errm) - 1); failed = true; break; }
                // winners whose similarity is already known
                // (round 6 also tried walk_kernel.inc's ranked merge for them — under a metadata filter most of a row is a strong mismatch or a
                // Base replica: parity green, 2.17 -> 2.14 ms per 256-query batch and 3.08 -> 3.28 ms per 4 096 (2 KB more LDS per wave): not kept)
                u64 cm = __ballot(win && dec == 0);
                const u32 ckey = metric_key(metric, csim);
                while (cm) {
                    const int l = __ffsll((long long)cm) - 1;
                    cm &= cm - 1;
                    const u64 kk = pack_key(readlane_u32(ckey, l), readlane_u32(nb_node, l));
                    const int pos = pool.rank_of(kk);
                    if (pos < limit) { pool.insert_at(kk, pos, lane); if (npool < (u32)(64 * R)) npool++; if (pos < ahead) window_ok = false; }
                }
                // winners that need the vector cosine: compact (slot order) and evaluate RP rows per pass
                const bool vwin = win && dec == 1;
                const u64 vmask = __ballot(vwin);
                const int W = __popcll(vmask);
                if (vwin) {
                    const int rank = __popcll(vmask & ((1ull << lane) - 1ull));
                    sm.wl_vec[rank] = nvrow;
                    sm.wl_node[rank] = nb_node;
                }
                __builtin_amdgcn_wave_barrier();
                for (int base = 0; base < W && !failed; base += RP * PBM) {
                    u32 prow[PBM];
                    float pmag[PBM];
                    uint4 buf[PBM][CH];
                    float fdot[PBM];
#pragma unroll
                    for (int p = 0; p < PBM; p++) {
                        if (base + p * RP >= W) break;
                        const int my = base + p * RP + grp;
                        const bool v = my < W;
                        prow[p] = v ? sm.wl_vec[my] : 0u;
                        pmag[p] = v ? ix.mags[prow[p]] : 1.0f;
                        if constexpr (!FLOAT_ENG) {
#pragma unroll
                            for (int c = 0; c < CH; c++) {
                                const u32 chunk = (u32)lig + (u32)c * (u32)G;
                                buf[p][c] = make_uint4(0, 0, 0, 0);
                                if (v && chunk < ix.nchunks) buf[p][c] = *(const uint4 *)(ix.codes + (u64)prow[p] * ix.row_stride + (u64)chunk * 16);
                            }
                        } else if constexpr (ENG == ENG_F32)
                            fdot[p] = f32_oct_dot((const float *)(ix.codes + (u64)prow[p] * ix.row_stride), sm.qf, ix.dim, lane & 7);
                        else
                            fdot[p] = f16_lane_dot(ix.codes + (u64)prow[p] * ix.row_stride, sm.qf, ix.dim);
                    }
#pragma unroll
                    for (int p = 0; p < PBM; p++) {
                        if (base + p * RP >= W) break;
                        float dotf;
                        if constexpr (!FLOAT_ENG) {
                            u32 a = 0;
#pragma unroll
                            for (int c = 0; c < CH; c++) a = chunk_dot<ENG>(qreg[c], buf[p][c], a);
                            a = group_reduce_add_u32(a, G);
                            dotf = (float)a;
                        } else
                            dotf = fdot[p];
                        float sim = dotf;
                        bool bad = false;
                        if (metric == 0u) {
                            const float den = __fmul_rn(qmag, pmag[p]);
                            bad = (base + p * RP + grp < W) && den == 0.0f;
                            sim = __fdiv_rn(dotf, den);
                        }
                        if (__any(bad)) { status = COS_ERR_CALCULATION; failed = true; break; }
                        const u32 key = metric_key(metric, sim);
                        for (int g = 0; g < RP; g++) {
                            const int my = base + p * RP + g;
                            if (my >= W) break;
                            const u64 kk = pack_key(readlane_u32(key, g * G), sm.wl_node[my]);
                            const int pos = pool.rank_of(kk);
                            if (pos < limit) { pool.insert_at(kk, pos, lane); if (npool < (u32)(64 * R)) npool++; if (pos < ahead) window_ok = false; }
                        }
                    }
                }
                } // window entries
            }
            if (failed || status != COS_OK) break;
            // this walk's result: popped list sorted descending, first 100; entries with cosine exactly -1.0 are dropped;
            // merged into the level's candidate list (sorted descending, first 100 kept)
            u64 rk[R];
#pragma unroll
            for (int r = 0; r < R; r++) {
                const u32 e = (u32)lane * R + r;
                rk[r] = e < npop ? sm.res[e] : 0ull;
            }
            bitonic_sort_desc<R>(rk, lane);
            const u32 keepn = npop < keep_n ? npop : keep_n;
            const u32 minus1 = metric_key(metric, -1.0f);
            u64 mk[4]; // 256 >= 100 (so far) + 100 (this walk)
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const u32 e = (u32)lane * 4 + r;
                mk[r] = e < nacc ? sm.acc[e] : 0ull;
            }
            __builtin_amdgcn_wave_barrier();
            // the walk's survivors, compacted (they stay in descending order) into the staging list, then appended behind the
            // nacc entries already there and sorted together
            bool okf[R];
            u32 mine = 0;
#pragma unroll
            for (int r = 0; r < R; r++) {
                const u32 e = (u32)lane * R + r;
                okf[r] = e < keepn && (INDEXING || !(metric == 0u && (u32)(rk[r] >> 32) == minus1));
                mine += okf[r] ? 1u : 0u;
            }
            u32 incl = mine;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const u32 t = (u32)__shfl_up((int)incl, d, 64);
                if (lane >= d) incl += t;
            }
            const u32 nsurv = readlane_u32(incl, 63); // <= 100
            u32 pos = incl - mine;
#pragma unroll
            for (int r = 0; r < R; r++)
                if (okf[r]) sm.res[pos++] = rk[r]; // the popped list is dead: its LDS doubles as the staging list (>= 128 entries)
            __builtin_amdgcn_s_waitcnt(0);
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const u32 e = (u32)lane * 4 + r;
                if (e >= nacc && e < nacc + nsurv) mk[r] = sm.res[e - nacc];
            }
            bitonic_sort_desc<4>(mk, lane);
            nacc = nacc + nsurv < keep_n ? nacc + nsurv : keep_n;
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const u32 e = (u32)lane * 4 + r;
                if (e < nacc) sm.acc[e] = mk[r];
            }
            __builtin_amdgcn_s_waitcnt(0);
            __builtin_amdgcn_wave_barrier();
        }
        if (status != COS_OK) break;

        if (nacc == 0) { // no candidate survived: the entry node with its strongest match over the filters
            u64 best = 0ull;
            for (u32 f = f0; f < f1 && status == COS_OK; f++) {
                for (u32 j = lane; j < md; j += 64) { const int32_t v = wa.f_dims[(u64)f * md + j]; sm.fqi[j] = v; sm.fq[j] = (float)v; }
                __builtin_amdgcn_s_waitcnt(0);
                __builtin_amdgcn_wave_barrier();
                const float fmag = wa.f_mags[f];
                const int xkind = fmag == 0.0f ? KIND_BASE : KIND_METADATA; // (vector_store.rs:329-367: the fallback builds the query without an id)
                const u32 eid = lv.node_id[entry], ek = lv.node_meta[entry];
                const float eymag = ix.mmags[ek];
                float s0 = 0.0f;
                int dec = 0;
                if (lane == 0) dec = meta_decide(ek, eymag, eid, fmag, xkind, s0);
                dec = (int)readlane_u32((u32)dec, 0);
                s0 = __uint_as_float(readlane_u32(__float_as_uint(s0), 0));
                if (dec < 0) { status = -dec; break; }
                if (dec == 1 && !vec_distance_single(lv.node_vec[entry], s0)) { status = COS_ERR_CALCULATION; break; }
                const u64 kk = pack_key(metric_key(metric, s0), entry);
                if (best == 0ull || (u32)(kk >> 32) > (u32)(best >> 32)) best = kk;
                __builtin_amdgcn_wave_barrier();
            }
            if (status != COS_OK) break;
            if (best == 0ull) { status = COS_ERR_UNIMPLEMENTED; break; } // a query without any filter does not come here
            if (lane == 0) sm.acc[0] = best;
            nacc = 1;
            __builtin_amdgcn_s_waitcnt(0);
            __builtin_amdgcn_wave_barrier();
        }
        const u64 obase = ((u64)qi * (L + 1) + out_slot) * wa.keep;
        for (u32 e = lane; e < nacc; e += 64) {
            const u64 kv = sm.acc[e];
            wa.out_ids[obase + e] = lv.node_id[(u32)kv];
            wa.out_sims[obase + e] = metric_key_inv(metric, (u32)(kv >> 32));
            if (INDEXING && wa.out_nodes) wa.out_nodes[obase + e] = (u32)kv;
        }
        if (lane == 0) wa.out_counts[(u64)qi * (L + 1) + out_slot] = nacc;
        if (level > 0) entry = lv.child[(u32)sm.acc[0]];
        __builtin_amdgcn_wave_barrier();
    }
    if (lane == 0) wa.out_status[qi] = status;
}

} // namespace

namespace cosdev {

size_t walk_meta_smem_bytes(const IndexDev &ix, u32 ef, int eng) {
    const u32 Mmax = ix.lv[0].M > ix.lv[ix.num_layers].M ? ix.lv[0].M : ix.lv[ix.num_layers].M;
    const u32 res = ef < 128u ? 128u : ef; // the popped list doubles as the 128-entry staging list of the merge
    size_t b = (size_t)Mmax * 8 + (size_t)res * 8 + 128 * 8 + 64 * 4 * 4 + (size_t)5 * LA_M * 64 * 4;
    b = (b + 15) & ~(size_t)15;
    if (eng == ENG_F32) b += (size_t)ix.row_stride;
    if (eng == ENG_F16) b += ((size_t)ix.dim * 4 + 15) & ~(size_t)15;
    return b + 16;
}

template <int ENG, int CH, bool INDEXING>
static hipError_t launch_meta_r(const IndexDev &ix, const WalkArgs &wa, hipStream_t st) {
    const size_t smem = walk_meta_smem_bytes(ix, wa.ef, ENG);
    dim3 grid(wa.B), block(64);
    if (wa.ef <= 64) hipLaunchKernelGGL((walk_meta_kernel<ENG, CH, 1, INDEXING>), grid, block, smem, st, ix, wa);
    else if (wa.ef <= 256) hipLaunchKernelGGL((walk_meta_kernel<ENG, CH, 4, INDEXING>), grid, block, smem, st, ix, wa);
    else if (wa.ef <= 512) hipLaunchKernelGGL((walk_meta_kernel<ENG, CH, 8, INDEXING>), grid, block, smem, st, ix, wa);
    else if (wa.ef <= 1024) hipLaunchKernelGGL((walk_meta_kernel<ENG, CH, 16, INDEXING>), grid, block, smem, st, ix, wa);
    else return hipErrorInvalidValue;
    return hipGetLastError();
}

template <bool INDEXING>
static hipError_t launch_walk_meta_t(int eng, const IndexDev &ix, const WalkArgs &wa, hipStream_t st) {
    if (wa.B == 0) return hipSuccess;
    const u32 ch = (eng == ENG_F32 || eng == ENG_F16) ? 1 : (ix.nchunks + ix.G - 1) / ix.G;
    switch (eng) {
    case ENG_U8: return ch == 1 ? launch_meta_r<ENG_U8, 1, INDEXING>(ix, wa, st) : (ch == 2 ? launch_meta_r<ENG_U8, 2, INDEXING>(ix, wa, st) : hipErrorInvalidValue);
    case ENG_Q2: return ch == 1 ? launch_meta_r<ENG_Q2, 1, INDEXING>(ix, wa, st) : hipErrorInvalidValue;
    case ENG_Q1: return ch == 1 ? launch_meta_r<ENG_Q1, 1, INDEXING>(ix, wa, st) : hipErrorInvalidValue;
    case ENG_Q3: return ch == 1 ? launch_meta_r<ENG_Q3, 1, INDEXING>(ix, wa, st) : hipErrorInvalidValue;
    case ENG_F32: return launch_meta_r<ENG_F32, 1, INDEXING>(ix, wa, st);
    case ENG_F16: return launch_meta_r<ENG_F16, 1, INDEXING>(ix, wa, st);
    default: return hipErrorInvalidValue;
    }
}

hipError_t launch_walk_meta(int eng, const IndexDev &ix, const WalkArgs &wa, hipStream_t st) { return launch_walk_meta_t<false>(eng, ix, wa, st); }
// the walks of the component's builder (cos_index_build_meta): wa.q_rows / self_ids / keep / out_nodes as described at the kernel
hipError_t launch_walk_meta_index(int eng, const IndexDev &ix, const WalkArgs &wa, hipStream_t st) { return launch_walk_meta_t<true>(eng, ix, wa, st); }

} // namespace cosdev
