// kernels_walk_lat4.hip — FOUR WAVES PER QUERY: the walk for the smallest launches (one client batch of 256 queries on a
// 256-CU chip), same ann_search / traverse_find_nearest (vector_store.rs:256-402, 1112-1204), same per-level lists bit for bit.
//
// The one-wave latency kernel (kernels_walk_lat.hip) spends half of a lone batch issuing instructions — 229 k per query at ~4.5
// clocks each (profiles/archive/r03_single_batch_sq_counters_one_wave_latency_walk.txt) — and the other half parked on two dependent HBM
// round trips per round.  A workgroup of four waves (one per SIMD of the query's CU) attacks both:
//   * the window is NW x E entries (E = 1: four, the default; E = 2: eight), wave w owns entries w (, w + 4): its adjacency rows,
//     its candidates under the filter as it stands at the start of the round, their similarities — the whole speculative half of
//     a round splits four ways with no exchange;
//   * the COMMIT is data-parallel over the window instead of sequential per entry (model-checked in
//     tests/test_commit_equivalence.py: same pops, same filter, same pool wherever it can still be popped):
//       - a candidate wins iff no EARLIER (entry, slot) of the round holds its filter bit: one LDS atomicMin of the rank
//         entry * 64 + slot per bit, all entries at once (the lower slot wins an alias inside an entry, the earlier entry across
//         entries — the sequential scan order);
//       - entry i makes the window stale iff one of its winners beats the LAST waiting window entry (an insert lands ahead of a
//         waiting entry iff it beats the smallest of them), so the commit horizon j = the first such entry; entries after j are
//         not committed (their claims are dropped, they are looked at again next round);
//       - pool := top-CAP of (pool minus the j + 1 popped heads) U (winners of entries <= j): every pool entry and every winner
//         computes its own position (binary search + a count over the round's winners) and is scattered into the other pool buffer.
//         The sequential kernels' `rank < limit` test only ever rejects keys that can never be popped, so the plain merge is
//         equivalent.
// Four workgroup barriers per round.  Integer engines ENG_U8 / ENG_Q2 with <= 64 chunks per row, ef <= 256, reference visited
// filter — the launches the one-wave latency kernel takes; launch_walk picks this one up to cos_index_set_latency_waves' size.
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "engine_types.h"
#include "tuning.h"
#include "dot_engines.h"

using namespace cosdev;

#define COS_OK 0
#define COS_ERR_CALCULATION 2
#define COS_QUERY_ID 0xFFFFFFFEu
#define COS_ROOT_ID 0xFFFFFFFFu

namespace {

constexpr int NW = 4;   // waves per query
constexpr int GL4 = 16; // lanes per code row
constexpr int RPL4 = 64 / GL4;
constexpr int PBL4 = 8; // passes in flight before the dots are consumed (32 rows per wave)
constexpr u32 FREE = 0xFFFFFFFFu;

template <int ENG, int CH, int R, int E>
__global__ __launch_bounds__(256) void walk_lat4_kernel(const IndexDev ix, const WalkArgs wa) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    constexpr u32 CAP = 64u * R;
    constexpr u32 LA = (u32)(NW * E);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6); // wave index in an SGPR: its tests are scalar branches
    const u32 qi = blockIdx.x;
    if (qi >= wa.B) return;

    const u32 Mmax = ix.lv[0].M > ix.lv[ix.num_layers].M ? ix.lv[0].M : ix.lv[ix.num_layers].M;
    u64 *s_pool, *s_res, *s_win, *s_cl, *s_spec;
    u32 *s_vis, *s_first, *s_flag, *s_misc;
    {
        unsigned char *p = smem_raw;
        s_pool = (u64 *)p;  p += (size_t)2 * CAP * 8;           // sorted (descending) pool, double-buffered
        s_res = (u64 *)p;   p += (size_t)wa.ef * 8;             // popped (key, node) list
        s_win = (u64 *)p;   p += (size_t)LA * 64 * 8;           // the winners of the round's committed entries (keys), entry by entry
        s_cl = (u64 *)p;    p += (size_t)NW * E * 64 * 8;       // per wave: compacted candidates, vector row | (i * 64 + slot) << 32
        s_spec = (u64 *)p;  p += (size_t)NW * E * 64 * 8;       // per wave: per (own entry i, slot) similarity key | zero-denominator << 32
        s_first = (u32 *)p; p += (size_t)64 * Mmax * 4;         // per filter bit: smallest rank (entry * 64 + slot) claiming it this round
        s_vis = (u32 *)p;   p += (size_t)2 * Mmax * 4;          // visited filter words
        s_flag = (u32 *)p;  p += (size_t)LA * 4;                // per entry: stale | fail << 1 | winners << 8
        s_misc = (u32 *)p;                                      // [0] level entry hand-over, [1] status
    }
    u64 *my_cl = s_cl + (size_t)wave * E * 64, *my_spec = s_spec + (size_t)wave * E * 64;

    const u32 qrow = wa.q_rows ? wa.q_rows[qi] : qi;
    const u32 self_id = wa.self_ids ? wa.self_ids[qi] * ix.id_stride : COS_QUERY_ID;
    const uint8_t *qcode = wa.qcodes + (u64)qrow * ix.row_stride;
    const float qmag = wa.qmags[qrow];
    const u32 N = ix.n;
    const u32 L = ix.num_layers;
    const u32 metric = ix.metric;

    const int lig = lane & (GL4 - 1);
    const int grp = lane / GL4;
    const u64 lt_mask = (1ull << lane) - 1ull;
    uint4 qreg[CH];
#pragma unroll
    for (int c = 0; c < CH; c++) {
        const u32 chunk = (u32)lig + (u32)c * (u32)GL4;
        qreg[c] = chunk < ix.nchunks ? *(const uint4 *)(qcode + (u64)chunk * 16) : make_uint4(0, 0, 0, 0);
    }

    for (u32 i = tid; i < 64u * Mmax; i += 256) s_first[i] = FREE;

    u64 n_evals = 0, n_exp = 0, adj_bytes = 0, n_rounds = 0; // identical in every thread
    int32_t status = COS_OK;
    u32 entry = ix.lv[L].root_idx;

    // similarity of ONE row, computed by lane group 0 of the calling wave; result in every lane of that wave
    auto single_distance = [&](u32 row, float &sim_out) -> bool {
        u32 acc = 0;
        if (grp == 0) {
#pragma unroll
            for (int c = 0; c < CH; c++) {
                const u32 chunk = (u32)lig + (u32)c * (u32)GL4;
                if (chunk < ix.nchunks) acc = chunk_dot<ENG>(qreg[c], *(const uint4 *)(ix.codes + (u64)row * ix.row_stride + (u64)chunk * 16), acc);
            }
        }
        acc = group_reduce_add_u32(acc, GL4);
        acc = readlane_u32(acc, 0);
        const float dotf = (float)acc; // integer dot `as f32` (RNE)
        if (metric == 0u) {            // cosine_similarity_from_dot_product (cosine.rs:223-235)
            const float den = uniform_f32(__fmul_rn(qmag, ix.mags[row]));
            if (den == 0.0f) return false;
            sim_out = __fdiv_rn(dotf, den);
        } else {
            sim_out = dotf; // DotProductDistance (dotproduct.rs:14-64)
        }
        return true;
    };

    for (int level = (int)L; level >= 0; level--) {
        const LevelDev lv = ix.lv[level];
        const u32 M = lv.M;
        const u32 slots = M < ix.shortlist ? M : ix.shortlist;
        const u32 bitmask = 64u * M - 1u;
        const u32 out_slot = L - (u32)level;
        u32 cur = 0; // pool buffer in use
        // table level (WalkArgs::tab, u8 codes): the integer dot (`as f32`) of the query with every node of this level was computed
        // ahead of the walk; the cosine is formed here with the product and quotient of the row levels (walk_kernel.inc)
        const bool tab_level = ENG == ENG_U8 && wa.tab != nullptr && (u32)level >= wa.tab_level_min;
        const float *tabq = wa.tab + (u64)qi * wa.tab_stride + wa.tab_col0[level];
        const float *tabm = wa.tab_mags + wa.tab_col0[level];

        // fresh visited filter, pre-seeded with the query / new-node id (vector_store.rs:266-271, :807)
        for (u32 w = tid; w < 2 * M; w += 256) s_vis[w] = 0;
        __syncthreads();
        u32 npool = 0, npop = 0;
        if (wave == 0) { // start node (vector_store.rs:1144-1148)
            const u32 erow = uniform_u32(lv.node_vec ? lv.node_vec[entry] : entry);
            float s0 = 0.0f;
            bool ok;
            if (tab_level) {
                s0 = uniform_f32(tabq[entry]);
                ok = true;
                if (metric == 0u) {
                    const float den = uniform_f32(__fmul_rn(qmag, tabm[entry]));
                    ok = den != 0.0f;
                    s0 = __fdiv_rn(s0, den); // (0 / 0 when !ok: never used)
                }
            } else
                ok = single_distance(erow, s0);
            if (lane == 0) {
                const u32 b = self_id & bitmask;
                s_vis[b >> 5] |= 1u << (b & 31);
                const u32 eid = erow == N ? COS_ROOT_ID : erow * ix.id_stride;
                const u32 b2 = eid & bitmask;
                s_vis[b2 >> 5] |= 1u << (b2 & 31);
                s_pool[0] = pack_key(metric_key(metric, s0), entry);
                s_misc[1] = ok ? 0u : 1u;
            }
        }
        n_evals++;
        __syncthreads();
        if (uniform_u32(s_misc[1])) { status = COS_ERR_CALCULATION; break; } // the same word in every lane: a scalar branch
        npool = 1;

        bool failed = false;
        while (npool > 0 && npop < wa.ef) {
            n_rounds++;
            u32 kwin = npool < LA ? npool : LA;
            if (kwin > wa.ef - npop) kwin = wa.ef - npop;
            const u64 *pool = s_pool + (size_t)cur * CAP;
            u64 *pool_nx = s_pool + (size_t)(cur ^ 1u) * CAP;

            // ---- A. this wave's window entries: adjacency rows (all loads first, none predicated), candidates, claims, similarities ----
            u32 av[E], an[E], wnode[E];
            u64 wkey[E];
            const u32 slot_l = (u32)lane < slots ? (u32)lane : slots - 1u;
#pragma unroll
            for (int i = 0; i < E; i++) {
                const u32 e = (u32)(i * NW + wave);
                wkey[i] = pool[e < kwin ? e : 0u];
                wnode[i] = uniform_u32((u32)wkey[i]); // the same LDS word in every lane: a scalar row base for the adjacency loads
                av[i] = lv.adj_vec[(u64)wnode[i] * M + slot_l];
            }
            if (level != 0) {
#pragma unroll
                for (int i = 0; i < E; i++) an[i] = lv.adj_node[(u64)wnode[i] * M + slot_l];
            }
            const u64 last_key = pool[kwin - 1u];
            bool cnd[E];
            u32 bitv[E];
            u32 T = 0;
#pragma unroll
            for (int i = 0; i < E; i++) {
                const u32 e = (u32)(i * NW + wave);
                const bool live = e < kwin && (u32)lane < slots;
                av[i] = live ? av[i] : ROW_EMPTY;
                an[i] = live ? (level == 0 ? av[i] : an[i]) : ROW_EMPTY;
                // PerformantFixedSet: bucket=(id>>6)&(M-1), bit=id&63  <=> linear bit id & (64M-1)
                const u32 id = av[i] == N ? COS_ROOT_ID : av[i] * ix.id_stride;
                bitv[i] = id & bitmask;
                const u32 vword = s_vis[bitv[i] >> 5];
                cnd[i] = av[i] != ROW_EMPTY && !(vword & (1u << (bitv[i] & 31u)));
                if (cnd[i]) atomicMin(&s_first[bitv[i]], e * 64u + (u32)lane);
                const u64 cm = ballot64(cnd[i]);
                if (cnd[i]) my_cl[T + (u32)__popcll(cm & lt_mask)] = (u64)av[i] | ((u64)(u32)(i * 64 + lane) << 32);
                T += (u32)__popcll(cm);
            }
            __builtin_amdgcn_wave_barrier();
            if (tab_level) {
                // table level: every candidate's dot and norm are two 4-byte gathers by the lane that holds the slot (no code row, no dot)
#pragma unroll
                for (int i = 0; i < E; i++) {
                    float sim = 0.0f, magv = 1.0f;
                    if (cnd[i]) { sim = tabq[an[i]]; magv = tabm[an[i]]; }
                    bool bad = false;
                    if (metric == 0u) {
                        const float den = __fmul_rn(qmag, magv);
                        bad = den == 0.0f;
                        sim = __fdiv_rn(sim, den);
                    }
                    if (cnd[i]) my_spec[i * 64 + lane] = (u64)metric_key(metric, sim) | (bad ? (1ull << 32) : 0ull);
                }
                T = 0; // nothing left for the row loop below
            }
            // similarities of this wave's candidates: 4 rows per pass, PBL4 passes in flight; no lane is ever masked off (see
            // kernels_walk_lat.hip: a lane group without a candidate re-reads the block's first row, a lane past the row's last
            // chunk re-reads that chunk against a zero query chunk)
            for (u32 b0 = 0; b0 < T; b0 += RPL4 * PBL4) {
                uint4 buf[PBL4][CH];
                float pmag[PBL4];
                u32 ppos[PBL4], prow[PBL4];
#pragma unroll
                for (int p = 0; p < PBL4; p++) {
                    if (b0 + (u32)(p * RPL4) >= T) break; // wave-uniform
                    const u32 my = b0 + (u32)(p * RPL4 + grp);
                    const bool v = my < T;
                    const u64 ce = my_cl[v ? my : b0];
                    prow[p] = (u32)ce;
                    ppos[p] = v ? (u32)(ce >> 32) : 0xFFFFFFFFu;
                }
#pragma unroll
                for (int p = 0; p < PBL4; p++) {
                    if (b0 + (u32)(p * RPL4) >= T) break; // wave-uniform
                    pmag[p] = ix.mags[prow[p]];
                    const uint8_t *rp = ix.codes + (u64)prow[p] * ix.row_stride;
#pragma unroll
                    for (int c = 0; c < CH; c++) {
                        u32 chunk = (u32)lig + (u32)c * (u32)GL4;
                        if (c == CH - 1) chunk = chunk < ix.nchunks ? chunk : ix.nchunks - 1u;
                        buf[p][c] = *(const uint4 *)(rp + (u64)chunk * 16);
                    }
                }
#pragma unroll
                for (int p = 0; p < PBL4; p++) {
                    if (b0 + (u32)(p * RPL4) >= T) break; // wave-uniform
                    u32 part[CH];
#pragma unroll
                    for (int c = 0; c < CH; c++) part[c] = chunk_dot<ENG>(qreg[c], buf[p][c], 0u);
                    u32 acc = part[0];
#pragma unroll
                    for (int c = 1; c < CH; c++) acc += part[c];
                    acc = group_reduce_add_u32(acc, GL4);
                    const float dotf = (float)acc; // integer dot `as f32` (RNE)
                    float sim = dotf;
                    bool bad = false;
                    if (metric == 0u) { // cosine_similarity_from_dot_product (cosine.rs:223-235)
                        const float den = __fmul_rn(qmag, pmag[p]);
                        bad = den == 0.0f;
                        sim = __fdiv_rn(dotf, den);
                    }
                    if (lig == 0 && ppos[p] != 0xFFFFFFFFu) my_spec[ppos[p]] = (u64)metric_key(metric, sim) | (bad ? (1ull << 32) : 0ull);
                }
            }
            __syncthreads(); // B1: every claim of the round is in s_first (and this wave's similarities are in my_spec)

            // ---- B. winners: first occurrence per filter bit in (entry, slot) order; stale / fail flags per entry ----------------------
            bool win[E];
            u64 key[E];
            u64 wmask[E];
#pragma unroll
            for (int i = 0; i < E; i++) {
                const u32 e = (u32)(i * NW + wave);
                win[i] = cnd[i] && s_first[bitv[i]] == e * 64u + (u32)lane;
                const u64 sp = win[i] ? my_spec[i * 64 + lane] : 0ull;
                key[i] = pack_key((u32)sp, an[i]);
                wmask[i] = ballot64(win[i]);
                if (e < kwin) { // wave-uniform
                    const bool stale = e + 1u < kwin && (wmask[i] & ballot64(key[i] > last_key)) != 0ull;
                    const bool bad = (wmask[i] & ballot64((sp >> 32) != 0ull)) != 0ull;
                    if (lane == 0) s_flag[e] = (stale ? 1u : 0u) | (bad ? 2u : 0u) | ((u32)__popcll(wmask[i]) << 8);
                }
            }
            __syncthreads(); // B2: the flags of every entry

            // ---- C. commit horizon; the winners of the committed entries are appended; filter bits; popped list --------------------
            // (the flags are read by one lane each and combined with ballots: a loop of dependent LDS reads costs a lone wave
            // ~130 clocks per iteration)
            u32 j = kwin - 1u, n_valid = 0, my_base[E];
            {   // the <= LA flags are read by one lane each, broadcast with v_readlane and combined in scalar registers (a scan with
                // __shfl_up is five ds_bpermute round trips, ~300 clocks for a lone wave)
                const u32 f = (u32)lane < kwin ? s_flag[lane] : 0u;
                bool found = false;
#pragma unroll
                for (int i = 0; i < E; i++) my_base[i] = 0;
#pragma unroll
                for (int e = 0; e < (int)LA; e++) {
                    const u32 fe = readlane_u32(f, e);
                    if ((u32)e < kwin && !found) {
#pragma unroll
                        for (int i = 0; i < E; i++)
                            if ((u32)(i * NW + wave) == (u32)e) my_base[i] = n_valid; // where this wave's committed entry puts its winners
                        n_valid += fe >> 8;
                        if (fe & 2u) failed = true; // a committed entry has a winner with a zero denominator -> CalculationError
                        if (fe & 1u) { j = (u32)e; found = true; }
                    }
                }
            }
            if (failed) break; // uniform
#pragma unroll
            for (int i = 0; i < E; i++) {
                const u32 e = (u32)(i * NW + wave);
                if (win[i] && e <= j) {
                    atomicOr(&s_vis[bitv[i] >> 5], 1u << (bitv[i] & 31u));
                    s_win[my_base[i] + (u32)__popcll(wmask[i] & lt_mask)] = key[i];
                }
                if (cnd[i]) s_first[bitv[i]] = FREE; // every claim of the round is released (losers and uncommitted entries too)
            }
            if ((u32)tid <= j) s_res[npop + (u32)tid] = pool[tid];
            const u64 pk = (u32)tid < npool ? pool[tid] : 0ull; // this thread's pool entry (read before the barrier: one LDS round trip)
            if (n_valid) __syncthreads(); // B3: the winner list (uniform: nobody wrote one if no committed entry has a winner)

            // ---- D. merge: pool := top-CAP of (pool minus the popped heads) U winners ---------------------------------------------------
            if (n_valid == 0u) {
                // nothing was inserted (most pops discover nothing new): the pool just loses its heads
                if ((u32)tid > j && (u32)tid < npool) pool_nx[(u32)tid - (j + 1u)] = pk;
            } else if (n_valid <= 64u && npool <= 64u) {
                // the usual case at ef <= 64: pool and winners both fit ONE wave's lanes, so wave 0 merges alone with ballots and
                // v_readlane — no LDS round trip inside the loop, no binary search (8 dependent LDS reads), the other waves wait at B4
                if (wave == 0) {
                    const u64 kw_l = (u32)lane < n_valid ? s_win[lane] : 0ull; // lane l = winner l
                    const bool live_p = (u32)lane > j && (u32)lane < npool;    // lane l = pool entry l
                    u32 pos_p = (u32)lane - (j + 1u), pos_w = 0;
                    for (u32 w = 0; w < n_valid; w++) {
                        const u64 kw = readlane_u64(kw_l, (int)w);
                        pos_p += kw > pk ? 1u : 0u;                              // a pool entry moves up by the winners above it
                        pos_w += kw > kw_l ? 1u : 0u;                            // a winner lands behind the winners above it ...
                        const u32 above = (u32)__popcll(ballot64(live_p && pk > kw));
                        if ((u32)lane == w) pos_w += above;                      // ... and behind the pool entries above it
                    }
                    if (live_p && pos_p < CAP) pool_nx[pos_p] = pk;
                    if ((u32)lane < n_valid && pos_w < CAP) pool_nx[pos_w] = kw_l;
                }
            } else {
            // general case: every wave holds the winners in registers (lane l of register r = winner r * 64 + l) and broadcasts them with
            // v_readlane; pool entries are spread over the workgroup's threads, winners' registers over the waves
            constexpr int WR = (int)((LA * 64u + 63u) / 64u); // registers for up to LA * 64 winners
            u64 wreg[WR];
#pragma unroll
            for (int r = 0; r < WR; r++) wreg[r] = ((u32)(r * 64 + lane) < n_valid) ? s_win[r * 64 + lane] : 0ull;
            // a pool entry behind the popped heads moves down by the heads and up by the winners above it
            u32 pos_p = (u32)tid - (j + 1u);
            // a winner lands behind the pool entries above it (binary search, all winners of a wave at once) and the winners above it
            u32 pos_w[WR];
#pragma unroll
            for (int r = 0; r < WR; r++) {
                pos_w[r] = 0;
                if ((u32)(r * 64) < n_valid && wave == (r & (NW - 1))) { // wave-uniform: winners' registers are spread over the waves
                    u32 lo = j + 1u, hi = npool; // pool entries (descending) greater than the winner: [j + 1, lo)
                    const u64 kw = wreg[r];
                    while (ballot64(lo < hi)) {
                        const u32 mid = (lo + hi) >> 1;
                        const bool go = lo < hi;
                        const u64 pm = pool[go ? mid : j + 1u];
                        if (go) { if (pm > kw) lo = mid + 1u; else hi = mid; }
                    }
                    pos_w[r] = lo - (j + 1u);
                }
            }
            if (n_valid <= 64u) { // the usual case: one register of winners
                for (u32 w = 0; w < n_valid; w++) {
                    const u64 kw = readlane_u64(wreg[0], (int)w);
                    pos_p += kw > pk ? 1u : 0u;
                    pos_w[0] += kw > wreg[0] ? 1u : 0u;
                }
            } else {
                for (u32 w = 0; w < n_valid; w++) { // winner w, broadcast from its register
                    u64 kw = 0ull;
#pragma unroll
                    for (int r = 0; r < WR; r++)
                        if ((w >> 6) == (u32)r) kw = readlane_u64(wreg[r], (int)(w & 63u));
                    pos_p += kw > pk ? 1u : 0u;
#pragma unroll
                    for (int r = 0; r < WR; r++)
                        if ((u32)(r * 64) < n_valid) pos_w[r] += kw > wreg[r] ? 1u : 0u;
                }
            }
            if ((u32)tid > j && (u32)tid < npool && pos_p < CAP) pool_nx[pos_p] = pk;
#pragma unroll
            for (int r = 0; r < WR; r++)
                if ((u32)(r * 64 + lane) < n_valid && wave == (r & (NW - 1)) && pos_w[r] < CAP) pool_nx[pos_w[r]] = wreg[r];
            }
            n_evals += n_valid;
            n_exp += j + 1u;
            adj_bytes += (u64)(j + 1u) * slots * 4;
            npop += j + 1u;
            npool = npool - (j + 1u) + n_valid;
            if (npool > CAP) npool = CAP;
            cur ^= 1u;
            __syncthreads(); // B4: the new pool, the filter and the released claims are in place
        }
        if (failed) { status = COS_ERR_CALCULATION; break; }

        // keep the best `keep`, sorted descending (vector_store.rs:1194-1201): wave 0 sorts the popped list in registers
        __syncthreads();
        if (wave == 0) {
            u64 rk[R];
#pragma unroll
            for (int r = 0; r < R; r++) {
                const u32 e = (u32)lane * R + r;
                rk[r] = e < npop ? s_res[e] : 0ull;
            }
            bitonic_sort_desc<R>(rk, lane);
            u32 cnt = npop < wa.keep ? npop : wa.keep;
            bool ok = true;
            if (npop == 0) { // only if ef == 0: the entry node's own distance (vector_store.rs:329-380)
                const u32 erow = lv.node_vec ? lv.node_vec[entry] : entry;
                float s0 = 0.0f;
                ok = single_distance(erow, s0);
                rk[0] = lane == 0 ? pack_key(metric_key(metric, s0), entry) : 0ull;
                cnt = 1;
            }
            const u64 obase = ((u64)qi * (L + 1) + out_slot) * wa.keep;
#pragma unroll
            for (int r = 0; r < R; r++) {
                const u32 e = (u32)lane * R + r;
                if (ok && e < cnt) {
                    const u32 nd = (u32)rk[r];
                    const u32 vrow = lv.node_vec ? lv.node_vec[nd] : nd;
                    wa.out_ids[obase + e] = vrow == N ? COS_ROOT_ID : vrow * ix.id_stride;
                    wa.out_sims[obase + e] = metric_key_inv(metric, (u32)(rk[r] >> 32));
                    if (wa.out_nodes) wa.out_nodes[obase + e] = nd;
                }
            }
            if (lane == 0) {
                wa.out_counts[(u64)qi * (L + 1) + out_slot] = ok ? cnt : 0u;
                s_misc[1] = ok ? 0u : 1u;
                if (level > 0 && ok) s_misc[0] = lv.child[(u32)readlane_u64(rk[0], 0)]; // descend through the best hit's child link (vector_store.rs:382-385)
            }
        }
        __syncthreads();
        if (uniform_u32(s_misc[1])) { status = COS_ERR_CALCULATION; break; }
        if (level > 0) entry = uniform_u32(s_misc[0]);
        __syncthreads(); // s_misc is rewritten by the next level's start node
    }

    if (tid == 0) {
        wa.out_status[qi] = status;
        if (wa.out_stats) {
            wa.out_stats[(u64)qi * 4 + 0] = n_evals;
            wa.out_stats[(u64)qi * 4 + 1] = n_exp;
            wa.out_stats[(u64)qi * 4 + 2] = adj_bytes;
            wa.out_stats[(u64)qi * 4 + 3] = n_rounds;
        }
    }
}

template <int E>
size_t walk_lat4_smem_bytes(const IndexDev &ix, u32 ef, u32 cap) {
    const u32 Mmax = ix.lv[0].M > ix.lv[ix.num_layers].M ? ix.lv[0].M : ix.lv[ix.num_layers].M;
    const size_t la = (size_t)NW * E;
    return (size_t)2 * cap * 8 + (size_t)ef * 8 + la * 64 * 8 + (size_t)NW * E * 64 * 8 * 2 + (size_t)64 * Mmax * 4 + (size_t)2 * Mmax * 4 + la * 4 + 16;
}

template <int ENG, int CH, int E>
hipError_t launch_lat4_r(const IndexDev &ix, const WalkArgs &wa, hipStream_t st) {
    dim3 grid(wa.B), block(256);
    if (wa.ef <= 64) hipLaunchKernelGGL((walk_lat4_kernel<ENG, CH, 1, E>), grid, block, walk_lat4_smem_bytes<E>(ix, wa.ef, 64), st, ix, wa);
    else hipLaunchKernelGGL((walk_lat4_kernel<ENG, CH, 4, E>), grid, block, walk_lat4_smem_bytes<E>(ix, wa.ef, 256), st, ix, wa);
    return hipGetLastError();
}

template <int ENG, int E>
hipError_t launch_lat4_ch(const IndexDev &ix, const WalkArgs &wa, u32 ch, hipStream_t st) {
    switch (ch) {
    case 1: return launch_lat4_r<ENG, 1, E>(ix, wa, st);
    case 2: return launch_lat4_r<ENG, 2, E>(ix, wa, st);
    case 3: return launch_lat4_r<ENG, 3, E>(ix, wa, st);
    case 4: return launch_lat4_r<ENG, 4, E>(ix, wa, st);
    default: return hipErrorInvalidValue;
    }
}

} // namespace

namespace cosdev {

// same launches as the one-wave latency kernel (walk_lat_applicable), up to max_B queries
bool walk_lat4_applicable(int eng, const IndexDev &ix, const WalkArgs &wa, u32 max_B) {
    if (max_B == 0 || wa.B > max_B) return false;
    if (ix.visited_mode != 0) return false;
    if (eng != ENG_U8 && eng != ENG_Q2) return false;
    if (ix.nchunks == 0 || ix.nchunks > (u32)(4 * GL4)) return false;
    if (wa.ef == 0 || wa.ef > 256) return false;
    const u32 Mmax = ix.lv[0].M > ix.lv[ix.num_layers].M ? ix.lv[0].M : ix.lv[ix.num_layers].M;
    if (Mmax > 64) return false; // s_first is sized 64 * Mmax words: keep the workgroup's LDS small
    return true;
}

hipError_t launch_walk_lat4(int eng, const IndexDev &ix, const WalkArgs &wa, hipStream_t st) {
    // window: 4 entries (one per wave).  8 (two per wave, COS_WALK_LAT4_E=2) needs 24 % fewer rounds but measured 2.6x slower per
    // round (profiles/archive/r03_latency_sweep_*): the wasted evaluations and the wider merge cost every wave more than the rounds save
    const int e_env = (int)tune_or(TUNE_WALK_LAT4_E, 1);
    const u32 ch = (ix.nchunks + GL4 - 1) / GL4;
    if (e_env == 1) {
        if (eng == ENG_U8) return launch_lat4_ch<ENG_U8, 1>(ix, wa, ch, st);
        if (eng == ENG_Q2) return launch_lat4_ch<ENG_Q2, 1>(ix, wa, ch, st);
    } else {
        if (eng == ENG_U8) return launch_lat4_ch<ENG_U8, 2>(ix, wa, ch, st);
        if (eng == ENG_Q2) return launch_lat4_ch<ENG_Q2, 2>(ix, wa, ch, st);
    }
    return hipErrorInvalidValue;
}

} // namespace cosdev
