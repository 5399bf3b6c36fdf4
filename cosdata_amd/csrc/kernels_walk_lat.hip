// kernels_walk_lat.hip — the LATENCY variant of walk_kernel (kernels_walk.hip) for small launches: same walk
// (ann_search / traverse_find_nearest, vector_store.rs:256-402, 1112-1204), same per-level lists bit for bit.
//
// A lone 256-query batch (the reference's `query-batch = 256`) gives the throughput kernel one wave per CU: that wave spends
// 42 % of its cycles issuing (77 instructions per distance evaluation with one code row per wave pass) and 58 % parked on a
// chain of dependent HBM round trips — adjacency row, then the code rows of every window entry one entry after the other.
// Here, still one wave per query:
//   * SPECULATION across the lookahead window.  The adjacency rows of the next `la` (<= 4) pool entries arrive together; every
//     neighbour that is unvisited *under the filter as it stands at the start of the round* is a candidate, the code rows of ALL
//     candidates of ALL window entries are fetched together (up to 32 rows in flight) and their similarities parked in LDS.  The
//     entries are then COMMITTED in pop order exactly like the throughput kernel: visited claims, inserts, the "still provably
//     the next pop" rule.  The filter only gains bits during a round, so a slot that wins at commit time was a candidate at
//     the start of the round: its similarity is there.  A similarity is a pure function of (query, row), so using the parked
//     value changes nothing; candidates that lose (claimed by an earlier entry of the round, or the window went stale) are
//     wasted bandwidth — on a chip that a small launch leaves idle.  A round is two dependent HBM round trips whatever the
//     window holds.
//   * 16 lanes per code row (4 rows per wave pass, 48-64 B per lane at 768-1024 u8 dims), no predicated load, one v_dot4 chain per
//     chunk; the commit is unrolled over the window (rows and keys stay in registers) and screens an expansion's winners with one
//     vector compare before the insertion loop.  A lone wave pays ~4 clocks for EVERY instruction, scalar ones included, so the
//     kernel is written for instruction count (DESIGN.md 4.1 has the step-by-step measurements).
// Integer engines ENG_U8 / ENG_Q2 with <= 64 chunks per row, ef <= 256, reference visited filter; everything else stays on
// walk_kernel (launch_walk decides; cos_index_set_latency_mode).
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

constexpr int LAL = 4;  // window capacity (adjacency rows prefetched per round); the launch picks la <= LAL.  8 measured slower (see launch_walk_lat)
constexpr int GL = 16;  // lanes per code row
constexpr int RPL = 64 / GL; // rows per wave pass
constexpr int PBL = 8;  // passes in flight before the dots are consumed (32 rows)

template <int ENG, int CH, int R, bool PREFETCH_ADJ>
__global__ __launch_bounds__(64) void walk_lat_kernel(const IndexDev ix, const WalkArgs wa, const u32 la) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int lane = threadIdx.x;
    const u32 qi = blockIdx.x;
    if (qi >= wa.B) return;

    const u32 Mmax = ix.lv[0].M > ix.lv[ix.num_layers].M ? ix.lv[0].M : ix.lv[ix.num_layers].M;
    u64 *s_spec, *s_cl, *s_res;
    u32 *s_vis;
    {
        unsigned char *p = smem_raw;
        s_spec = (u64 *)p;     p += (size_t)LAL * 64 * 8; // per (window entry, slot): similarity key | zero-denominator flag << 32
        s_cl = (u64 *)p;       p += (size_t)LAL * 64 * 8; // compacted candidates: vector row | (entry * 64 + slot) << 32
        s_res = (u64 *)p;      p += (size_t)wa.ef * 8;    // popped (key, node) list
        s_vis = (u32 *)p;      // visited filter words, Mmax * 2
    }

    const u32 qrow = wa.q_rows ? wa.q_rows[qi] : qi;
    const u32 self_id = wa.self_ids ? wa.self_ids[qi] * ix.id_stride : COS_QUERY_ID;
    const uint8_t *qcode = wa.qcodes + (u64)qrow * ix.row_stride;
    const float qmag = wa.qmags[qrow];
    const u32 N = ix.n;
    const u32 L = ix.num_layers;
    const u32 metric = ix.metric;

    const int lig = lane & (GL - 1);
    const int grp = lane / GL;
    const u64 lt_mask = (1ull << lane) - 1ull;
    uint4 qreg[CH];
#pragma unroll
    for (int c = 0; c < CH; c++) {
        const u32 chunk = (u32)lig + (u32)c * (u32)GL;
        qreg[c] = chunk < ix.nchunks ? *(const uint4 *)(qcode + (u64)chunk * 16) : make_uint4(0, 0, 0, 0);
    }

    u64 n_evals = 0, n_exp = 0, adj_bytes = 0, n_rounds = 0;
    int32_t status = COS_OK;
    u32 entry = ix.lv[L].root_idx;
    u32 warm = 0; // see the L2 warm-up in step 2

    // similarity of ONE row, computed by lane group 0; result in every lane
    auto single_distance = [&](u32 row, float &sim_out) -> bool {
        u32 acc = 0;
        if (grp == 0) {
#pragma unroll
            for (int c = 0; c < CH; c++) {
                const u32 chunk = (u32)lig + (u32)c * (u32)GL;
                if (chunk < ix.nchunks) acc = chunk_dot<ENG>(qreg[c], *(const uint4 *)(ix.codes + (u64)row * ix.row_stride + (u64)chunk * 16), acc);
            }
        }
        acc = group_reduce_add_u32(acc, GL);
        acc = readlane_u32(acc, 0);
        const float dotf = (float)acc; // integer dot `as f32` (RNE)
        if (metric == 0u) {            // cosine_similarity_from_dot_product (cosine.rs:223-235)
            const float den = __fmul_rn(qmag, ix.mags[row]);
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

        // fresh visited filter, pre-seeded with the query / new-node id (vector_store.rs:266-271, :807)
        for (u32 w = lane; w < 2 * M; w += 64) s_vis[w] = 0;
        __builtin_amdgcn_wave_barrier();
        if (lane == 0) {
            const u32 b = self_id & bitmask;
            s_vis[b >> 5] |= 1u << (b & 31);
        }

        Pool<R> pool;
        pool.clear();
        u32 npool = 0, npop = 0;

        { // start node (vector_store.rs:1144-1148)
            const u32 erow = lv.node_vec ? lv.node_vec[entry] : entry;
            float s0;
            n_evals++;
            if (!single_distance(erow, s0)) { status = COS_ERR_CALCULATION; break; }
            const u32 eid = erow == N ? COS_ROOT_ID : erow * ix.id_stride;
            if (lane == 0) {
                const u32 b = eid & bitmask;
                s_vis[b >> 5] |= 1u << (b & 31);
            }
            pool.insert_at(pack_key(metric_key(metric, s0), entry), 0, lane);
            npool = 1;
        }
        __builtin_amdgcn_wave_barrier();

        bool failed = false;
        while (npool > 0 && npop < wa.ef) {
            n_rounds++;
            u32 kwin = npool < la ? npool : la;
            if (kwin > wa.ef - npop) kwin = wa.ef - npop;

            // ---- 1. adjacency rows of the window entries: independent loads, one latency -----------------------------------
            u32 av[LAL], an[LAL];
            u64 wkey[LAL]; // wave-uniform: the window entries' (key, node); entries past kwin are never looked at
            // Every load is issued before the first value is looked at, and none is predicated: entries past kwin re-read node 0's
            // row, lanes past `slots` re-read the last slot (both replaced by ROW_EMPTY afterwards).  The first version wrote
            // `an[i] = level == 0 ? av[i] : ...` inside the per-entry `if`: the copy made the compiler wait for av[i] (s_waitcnt
            // vmcnt(0)) before it issued entry i+1's load — four DEPENDENT round trips per round instead of one.
            const u32 slot_l = (u32)lane < slots ? (u32)lane : slots - 1u;
            static_for<0, LAL>([&](auto ic) {
                constexpr int i = decltype(ic)::value;
                wkey[i] = pool.template peek<i>();
                const u32 nd = (u32)i < kwin ? (u32)wkey[i] : 0u;
                av[i] = lv.adj_vec[(u64)nd * M + slot_l];
            });
            if (level != 0) {
                static_for<0, LAL>([&](auto ic) {
                    constexpr int i = decltype(ic)::value;
                    const u32 nd = (u32)i < kwin ? (u32)wkey[i] : 0u;
                    an[i] = lv.adj_node[(u64)nd * M + slot_l];
                });
            }
            static_for<0, LAL>([&](auto ic) {
                constexpr int i = decltype(ic)::value;
                const bool live = (u32)i < kwin && (u32)lane < slots;
                av[i] = live ? av[i] : ROW_EMPTY;
                an[i] = live ? (level == 0 ? av[i] : an[i]) : ROW_EMPTY;
            });

            // ---- 2. candidates: unvisited under the filter as it stands now (read only), compacted entry by entry ----------
            u32 T = 0;
            u32 wv[2 * LAL]; // L2 warm-up loads (below)
            u32 vword[LAL]; // the filter words first (one LDS round trip for the window), then the ballots
            static_for<0, LAL>([&](auto ic) {
                constexpr int i = decltype(ic)::value;
                const u32 id = av[i] == N ? COS_ROOT_ID : av[i] * ix.id_stride; // an empty slot maps somewhere inside the filter too
                vword[i] = s_vis[(id & bitmask) >> 5];
            });
            static_for<0, LAL>([&](auto ic) {
                constexpr int i = decltype(ic)::value;
                const u32 id = av[i] == N ? COS_ROOT_ID : av[i] * ix.id_stride;
                const bool c = av[i] != ROW_EMPTY && !(vword[i] & (1u << (id & bitmask & 31u))); // entries past kwin hold only empty slots
                const u64 cm = __ballot(c);
                if (c) s_cl[T + (u32)__popcll(cm & lt_mask)] = (u64)av[i] | ((u64)(u32)(i * 64 + lane) << 32);
                T += (u32)__popcll(cm);
                // L2 warm-up: a candidate that is evaluated now is expanded a few rounds later, and its adjacency row is then the
                // FIRST of the round's two dependent HBM round trips.  One dword per 128-byte line of that row, requested now (a lane
                // per candidate: two load instructions per window entry, on a chip a small launch leaves idle), turns that trip
                // into an L2 / Infinity Cache hit.  The values are folded into `warm` AFTER step 3 (loads complete in order, so by
                // then they have arrived with the code rows and nothing waits for them on their own); `warm` is only ever written
                // out under a condition that cannot hold, so the loads survive the optimizer.
                if (PREFETCH_ADJ) {
                    const u32 *arow = lv.adj_vec + (u64)(c ? an[i] : 0u) * M;
                    wv[i] = arow[0];
                    wv[LAL + i] = arow[M > 32 ? 32 : 0];
                }
            });
            __builtin_amdgcn_wave_barrier();

            // ---- 3. similarities of every candidate: 4 rows per pass, PBL passes in flight ---------------------------------
            // No lane is ever masked off: a lane group without a candidate re-reads the block's first row and a lane past the
            // row's last chunk re-reads that chunk against a zero query chunk (both dropped / worth 0) — every predicated load
            // was three scalar instructions of exec bookkeeping, 40 % of this kernel's instructions were scalar
            // (profiles/archive/r02_single_batch_latency_walk_sq_counters.txt).
            for (u32 b0 = 0; b0 < T; b0 += RPL * PBL) {
                uint4 buf[PBL][CH];
                float pmag[PBL];
                u32 ppos[PBL], prow[PBL];
#pragma unroll
                for (int p = 0; p < PBL; p++) { // the candidates' rows first: one LDS round trip for the whole block
                    if (b0 + (u32)(p * RPL) >= T) break; // wave-uniform
                    const u32 my = b0 + (u32)(p * RPL + grp);
                    const bool v = my < T;
                    const u64 e = s_cl[v ? my : b0];
                    prow[p] = (u32)e;
                    ppos[p] = v ? (u32)(e >> 32) : 0xFFFFFFFFu;
                }
#pragma unroll
                for (int p = 0; p < PBL; p++) {
                    if (b0 + (u32)(p * RPL) >= T) break; // wave-uniform
                    pmag[p] = ix.mags[prow[p]];
                    const uint8_t *rp = ix.codes + (u64)prow[p] * ix.row_stride;
#pragma unroll
                    for (int c = 0; c < CH; c++) {
                        u32 chunk = (u32)lig + (u32)c * (u32)GL;
                        if (c == CH - 1) chunk = chunk < ix.nchunks ? chunk : ix.nchunks - 1u; // only the last round of chunks can overshoot
                        buf[p][c] = *(const uint4 *)(rp + (u64)chunk * 16);
                    }
                }
#pragma unroll
                for (int p = 0; p < PBL; p++) {
                    if (b0 + (u32)(p * RPL) >= T) break; // wave-uniform
                    u32 part[CH]; // one chain per chunk: independent dot4 chains interleave instead of waiting on each other
#pragma unroll
                    for (int c = 0; c < CH; c++) part[c] = chunk_dot<ENG>(qreg[c], buf[p][c], 0u);
                    u32 acc = part[0];
#pragma unroll
                    for (int c = 1; c < CH; c++) acc += part[c];
                    acc = group_reduce_add_u32(acc, GL);
                    const float dotf = (float)acc; // integer dot `as f32` (RNE)
                    float sim = dotf;
                    bool bad = false;
                    if (metric == 0u) { // cosine_similarity_from_dot_product (cosine.rs:223-235)
                        const float den = __fmul_rn(qmag, pmag[p]);
                        bad = den == 0.0f;
                        sim = __fdiv_rn(dotf, den);
                    }
                    if (lig == 0 && ppos[p] != 0xFFFFFFFFu) s_spec[ppos[p]] = (u64)metric_key(metric, sim) | (bad ? (1ull << 32) : 0ull);
                }
            }
            if (PREFETCH_ADJ) {
#pragma unroll
                for (int i = 0; i < 2 * LAL; i++) warm ^= wv[i];
            }
            __builtin_amdgcn_wave_barrier();

            // ---- 4. commit in pop order: walk_kernel's consume loop with the similarities read from LDS --------------------
            // Unrolled over the window so that an entry's adjacency row and key stay in the registers step 1 loaded them into
            // (the runtime loop read them back from LDS: three more dependent LDS round trips per pop for a lone wave).
            bool window_ok = true;
            static_for<0, LAL>([&](auto ic) {
                constexpr int wi = decltype(ic)::value;
                if ((u32)wi < kwin && window_ok && !failed) { // wave-uniform
                    const u64 sp_all = s_spec[wi * 64 + lane]; // issued with the filter word below: one LDS wait for both
                    pool.pop_head(lane);
                    npool--;
                    if (lane == 0) s_res[npop] = wkey[wi];
                    npop++;
                    n_exp++;
                    adj_bytes += (u64)slots * 4;
                    const int limit = (int)wa.ef - (int)npop;     // future pops still allowed
                    const int ahead = (int)kwin - 1 - wi;         // window entries still waiting at pool positions 0..ahead-1

                    const u32 nb_vec = av[wi], nb_node = an[wi];
                    const bool valid = nb_vec != ROW_EMPTY;
                    // PerformantFixedSet: bucket=(id>>6)&(M-1), bit=id&63  <=> linear bit id & (64M-1)
                    const u32 id = nb_vec == N ? COS_ROOT_ID : nb_vec * ix.id_stride;
                    const u32 bit = id & bitmask;
                    const u32 word = bit >> 5, msk = 1u << (bit & 31);
                    const bool pre = valid && (s_vis[word] & msk);
                    const bool cand = valid && !pre;
                    if (__any(cand)) { // else nothing new: the next window entry is certainly the next pop
                        u32 old = 0;
                        if (cand) old = atomicOr(&s_vis[word], msk);
                        const bool lost = cand && (old & msk);
                        bool win = cand && !lost;
                        u64 lostmask = __ballot(lost);
                        // two slots of this expansion alias the same residue: the LOWER slot wins (sequential scan order)
                        while (lostmask) {
                            const int l = __ffsll((long long)lostmask) - 1;
                            const u32 b = readlane_u32(bit, l);
                            const u64 g = __ballot(cand && bit == b);
                            const int w = __ffsll((long long)g) - 1;
                            if (cand && bit == b) win = (lane == w);
                            lostmask &= ~g;
                        }
                        u64 m = __ballot(win);
                        n_evals += (u64)__popcll(m);
                        if (__any(win && (sp_all >> 32) != 0ull)) failed = true; // zero denominator -> CalculationError
                        else {
                            const u32 keyv = (u32)sp_all;
                            // one vector compare screens every winner against the entry that closes the poppable part of the pool
                            // (position limit - 1): below it a key has at least `limit` entries above it and the loop would only
                            // rank it to reject it.  The bar only rises while winners go in, so the screen is conservative.
                            const u64 bar = limit > 0 ? pool.peek_dyn((u32)(limit - 1)) : ~0ull;
                            m = __ballot(win && pack_key(keyv, nb_node) > bar);
                            while (m) { // winners in slot order (vector_store.rs:1161-1171)
                                const int l = __ffsll((long long)m) - 1;
                                m &= m - 1;
                                const u64 kk = pack_key(readlane_u32(keyv, l), readlane_u32(nb_node, l));
                                const int pos = pool.rank_of(kk);
                                if (pos < limit) {
                                    pool.insert_at(kk, pos, lane);
                                    if (npool < (u32)(64 * R)) npool++;
                                    if (pos < ahead) window_ok = false; // landed ahead of a prefetched entry: the window is stale
                                }
                            }
                        }
                    }
                }
            });
            if (failed) break;
            __builtin_amdgcn_wave_barrier();
        }
        if (failed) { status = COS_ERR_CALCULATION; break; }

        // keep the best `keep`, sorted descending (vector_store.rs:1194-1201)
        __builtin_amdgcn_wave_barrier();
        u64 rk[R];
#pragma unroll
        for (int r = 0; r < R; r++) {
            const u32 e = (u32)lane * R + r;
            rk[r] = e < npop ? s_res[e] : 0ull;
        }
        bitonic_sort_desc<R>(rk, lane);
        u32 cnt = npop < wa.keep ? npop : wa.keep;
        if (npop == 0) { // only if ef == 0: the entry node's own distance (vector_store.rs:329-380)
            const u32 erow = lv.node_vec ? lv.node_vec[entry] : entry;
            float s0;
            if (!single_distance(erow, s0)) { status = COS_ERR_CALCULATION; break; }
            rk[0] = lane == 0 ? pack_key(metric_key(metric, s0), entry) : 0ull;
            cnt = 1;
        }
        const u64 obase = ((u64)qi * (L + 1) + out_slot) * wa.keep;
#pragma unroll
        for (int r = 0; r < R; r++) {
            const u32 e = (u32)lane * R + r;
            if (e < cnt) {
                const u32 nd = (u32)rk[r];
                const u32 vrow = lv.node_vec ? lv.node_vec[nd] : nd;
                wa.out_ids[obase + e] = vrow == N ? COS_ROOT_ID : vrow * ix.id_stride;
                wa.out_sims[obase + e] = metric_key_inv(metric, (u32)(rk[r] >> 32));
                if (wa.out_nodes) wa.out_nodes[obase + e] = nd;
            }
        }
        if (lane == 0) wa.out_counts[(u64)qi * (L + 1) + out_slot] = cnt;
        if (level > 0) { // descend through the best hit's child link (vector_store.rs:382-385)
            const u32 best = (u32)readlane_u64(rk[0], 0);
            entry = lv.child[best];
        }
    }

    if (warm == 0x9E3779B9u && wa.keep == 0xFFFFFFFFu) wa.out_status[qi] = (int32_t)warm; // keep is 100 / 64: never true, but not provably — keeps the warm-up loads alive
    if (lane == 0) {
        wa.out_status[qi] = status;
        if (wa.out_stats) {
            wa.out_stats[(u64)qi * 4 + 0] = n_evals;
            wa.out_stats[(u64)qi * 4 + 1] = n_exp;
            wa.out_stats[(u64)qi * 4 + 2] = adj_bytes;
            wa.out_stats[(u64)qi * 4 + 3] = n_rounds;
        }
    }
}

size_t walk_lat_smem_bytes(const IndexDev &ix, u32 ef) {
    const u32 Mmax = ix.lv[0].M > ix.lv[ix.num_layers].M ? ix.lv[0].M : ix.lv[ix.num_layers].M;
    return (size_t)LAL * 64 * 8 * 2 + (size_t)ef * 8 + (size_t)Mmax * 8 + 16;
}

template <int ENG, int CH>
hipError_t launch_lat_r(const IndexDev &ix, const WalkArgs &wa, u32 la, hipStream_t st) {
    const size_t smem = walk_lat_smem_bytes(ix, wa.ef);
    dim3 grid(wa.B), block(64);
    const bool warm = tune_or(TUNE_WALK_LAT_WARM, 1) != 0; // experiments: 0 = off
    if (warm) {
        if (wa.ef <= 64) hipLaunchKernelGGL((walk_lat_kernel<ENG, CH, 1, true>), grid, block, smem, st, ix, wa, la);
        else hipLaunchKernelGGL((walk_lat_kernel<ENG, CH, 4, true>), grid, block, smem, st, ix, wa, la);
    } else {
        if (wa.ef <= 64) hipLaunchKernelGGL((walk_lat_kernel<ENG, CH, 1, false>), grid, block, smem, st, ix, wa, la);
        else hipLaunchKernelGGL((walk_lat_kernel<ENG, CH, 4, false>), grid, block, smem, st, ix, wa, la);
    }
    return hipGetLastError();
}

template <int ENG>
hipError_t launch_lat_ch(const IndexDev &ix, const WalkArgs &wa, u32 ch, u32 la, hipStream_t st) {
    switch (ch) {
    case 1: return launch_lat_r<ENG, 1>(ix, wa, la, st);
    case 2: return launch_lat_r<ENG, 2>(ix, wa, la, st);
    case 3: return launch_lat_r<ENG, 3>(ix, wa, la, st);
    case 4: return launch_lat_r<ENG, 4>(ix, wa, la, st);
    default: return hipErrorInvalidValue;
    }
}

} // namespace

namespace cosdev {

// which launches take the latency kernel: reference filter, u8 / quaternary codes of <= 64 chunks, ef <= 256, at most max_B queries
bool walk_lat_applicable(int eng, const IndexDev &ix, const WalkArgs &wa, u32 max_B) {
    if (max_B == 0 || wa.B > max_B) return false;
    if (ix.visited_mode != 0) return false;
    if (eng != ENG_U8 && eng != ENG_Q2) return false;
    if (ix.nchunks == 0 || ix.nchunks > (u32)(4 * GL)) return false;
    if (wa.ef == 0 || wa.ef > 256) return false;
    return true;
}

hipError_t launch_walk_lat(int eng, const IndexDev &ix, const WalkArgs &wa, hipStream_t st) {
    // window size: 4 (profiles/archive/r02_latency_walk_sweep_first_version_window4_vs_8.jsonl: an 8-entry window needs 24 % fewer rounds but only 3.9
    // of its 8 entries are consumed before it goes stale, and the wasted evaluations cost more issue time than the rounds save; the
    // kernel has been unrolled for at most LAL = 4 since); tuning knob walk_lat_la = 1..4 narrows it per launch (experiments)
    const u32 la_env = (u32)std::max<long long>(0, tune_or(TUNE_WALK_LAT_LA, 0));
    u32 la = la_env ? la_env : 4u;
    if (la > (u32)LAL) la = LAL;
    const u32 ch = (ix.nchunks + GL - 1) / GL;
    if (eng == ENG_U8) return launch_lat_ch<ENG_U8>(ix, wa, ch, la, st);
    if (eng == ENG_Q2) return launch_lat_ch<ENG_Q2>(ix, wa, ch, la, st);
    return hipErrorInvalidValue;
}

} // namespace cosdev
