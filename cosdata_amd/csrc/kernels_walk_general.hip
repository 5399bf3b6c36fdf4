// kernels_walk_general.hip — the walk OUTSIDE the fast kernels' parameter domain: the same ann_search / traverse_find_nearest
// (vector_store.rs:256-402, 1112-1204), the same per-level lists bit for bit, for
//   * beams wider than the widest register pool (ef_search / ef_construction > 1024; hnsw/types.rs:10-17 accepts any u32), and
//   * more than 64 scanned neighbour slots per node (min(neighbors_count, shortlist_size) > 64; config.toml:32 is a free `usize`,
//     the reference's own gRPC test config sets 100: grpc/vectors/tests.rs:47), and
//   * code rows wider than walk_kernel's chunk passes (u8 above 4096 dimensions, SubByte above 64 chunks of 16 bytes: binary 8192,
//     quaternary 4096, octal 2048 dimensions).
// Rounds 1-5 refused all three at cos_index_create (COS_ERR_UNIMPLEMENTED).  This kernel is the plain statement of the loop — one wave per
// query, nothing speculative:
//   * the candidates of a level live in ONE array of ef keys in LDS, sorted descending.  The reference's BinaryHeap only ever hands out
//     its best ef - popped entries, so a sorted array truncated at `limit` = ef - popped reproduces the pop sequence (device_common.h
//     Pool); the popped entries simply stay where they are: positions [0, popped) ARE the popped list, [popped, popped + live) the
//     candidates, and live <= limit keeps the two inside the ef slots;
//   * an expansion scans its neighbour slots 64 at a time, in slot order (vector_store.rs:1161-1171); a pass tests and sets the
//     visited filter for its 64 slots exactly as walk_kernel.inc does (the lower slot wins an alias), evaluates the winners one row at a
//     time with the whole wave — any number of 16-byte chunks per row — and inserts them in slot order (binary search + a shift
//     through LDS);
//   * the level's result: the best `keep` of the popped entries by repeated selection.
// Every storage (u8, SubByte 1-3, f16, f32), cosine and dot product, the reference filter and COS_VISITED_EXACT, self ids, row
// indirection, delete_embedding's unseeded filter.  No level table, no locality order, no latency variants: launch_walk sends a launch
// here only when the index asks for what the fast kernels cannot hold, and run_search then skips the table GEMM and the split.
// A few times slower per query than walk_kernel at ef 1024 — a domain, not a fast path.
#include <hip/hip_runtime.h>

#include <algorithm>

#include "engine_types.h"
#include "dot_engines.h"

using namespace cosdev;

#define COS_OK 0
#define COS_ERR_CALCULATION 2
#define COS_QUERY_ID 0xFFFFFFFEu
#define COS_ROOT_ID 0xFFFFFFFFu

namespace {

__device__ __forceinline__ void wave_lds_sync() { // one wave's LDS writes before its (other lanes') LDS reads
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

template <int ENG, bool EXACT>
__global__ __launch_bounds__(64) void walk_general_kernel(const IndexDev ix, const WalkArgs wa) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int lane = threadIdx.x;
    const u32 qi = blockIdx.x;
    if (qi >= wa.B) return;
    constexpr bool FLOAT_ENG = ENG == ENG_F32 || ENG == ENG_F16;

    u64 *arr;      // [ef] popped list | candidates
    u32 *vis_lds;  // [2 * Mmax] visited filter words (reference filter)
    unsigned char *qstage; // the query's code row (integer engines) or its f32 values (float engines)
    {
        unsigned char *p = smem_raw;
        arr = (u64 *)p;      p += (size_t)(wa.ef ? wa.ef : 1u) * 8;
        vis_lds = (u32 *)p;  p += (size_t)wa.smem_mmax * 8;
        p = (unsigned char *)(((size_t)p + 15) & ~(size_t)15);
        qstage = p;
    }
    const u32 qrow = wa.q_rows ? wa.q_rows[qi] : qi;
    const u32 self_id = wa.self_ids ? wa.self_ids[qi] * ix.id_stride : COS_QUERY_ID; // self_ids are vector rows
    const uint8_t *qcode = wa.qcodes + (u64)qrow * ix.row_stride;
    const float qmag = wa.qmags[qrow];
    const u32 N = ix.n, L = ix.num_layers, metric = ix.metric;
    u32 *vis = EXACT ? wa.vis_bits + (u64)qi * wa.vis_words_per_query : vis_lds;
    u32 *vlog = EXACT ? wa.vis_log + (u64)qi * wa.vis_log_cap : nullptr;

    float *qf = (float *)qstage;
    if constexpr (ENG == ENG_F16) {
        const __half *qh = (const __half *)qcode;
        for (u32 i = lane; i < ix.dim; i += 64) qf[i] = __half2float(qh[i]);
    } else if constexpr (ENG == ENG_F32) {
        const float *qg = (const float *)qcode;
        for (u32 i = lane; i < (u32)(ix.row_stride / 4); i += 64) qf[i] = qg[i];
    } else {
        for (u32 c = lane; c < ix.nchunks; c += 64) *(uint4 *)(qstage + (size_t)c * 16) = *(const uint4 *)(qcode + (u64)c * 16);
    }
    wave_lds_sync();

    // similarity of ONE row by the whole wave; the same value in every lane.  false = zero denominator (CalculationError)
    auto row_similarity = [&](u32 row, float &sim_out) -> bool {
        float dotf;
        const uint8_t *rp = ix.codes + (u64)row * ix.row_stride;
        if constexpr (!FLOAT_ENG) {
            u32 acc = 0;
            for (u32 c = lane; c < ix.nchunks; c += 64) acc = chunk_dot<ENG>(*(const uint4 *)(qstage + (size_t)c * 16), *(const uint4 *)(rp + (u64)c * 16), acc);
            acc = group_reduce_add_u32(acc, 64);
            dotf = (float)readlane_u32(acc, 0); // integer dot `as f32` (RNE)
        } else if constexpr (ENG == ENG_F16) {
            const float d = f16_lane_dot(rp, qf, ix.dim);
            dotf = __uint_as_float(readlane_u32(__float_as_uint(d), 0));
        } else {
            const float d = f32_oct_dot((const float *)rp, qf, ix.dim, lane & 7);
            dotf = __uint_as_float(readlane_u32(__float_as_uint(d), 0));
        }
        if (metric == 0u) { // cosine_similarity_from_dot_product (cosine.rs:223-235)
            const float den = uniform_f32(__fmul_rn(qmag, ix.mags[row]));
            if (den == 0.0f) return false;
            sim_out = __fdiv_rn(dotf, den);
        } else {
            sim_out = dotf; // DotProductDistance (dotproduct.rs:14-64)
        }
        return true;
    };

    u64 n_evals = 0, n_exp = 0, adj_bytes = 0;
    int32_t status = COS_OK;
    u32 entry = uniform_u32(ix.lv[L].root_idx);

    for (int level = (int)L; level >= 0; level--) {
        const LevelDev lv = ix.lv[level];
        const u32 M = lv.M;
        const u32 slots = M < ix.shortlist ? M : ix.shortlist;
        const u32 bitmask = 64u * M - 1u;
        const u32 out_slot = L - (u32)level;
        u32 nlog = 0; // EXACT: entries of the undo log

        // fresh visited filter, pre-seeded with the query / new-node id (vector_store.rs:266-271, :807)
        if (!EXACT) {
            for (u32 w = lane; w < 2 * M; w += 64) vis_lds[w] = 0;
            wave_lds_sync();
            if (lane == 0 && wa.no_self_seed == 0u) {
                const u32 b = self_id & bitmask;
                vis_lds[b >> 5] |= 1u << (b & 31);
            }
        }
        u32 npool = 0, npop = 0, failed = 0;
        u32 lev_evals = 1;
        {   // start node (vector_store.rs:1144-1148)
            const u32 erow = uniform_u32(lv.node_vec ? lv.node_vec[entry] : entry);
            float s0;
            if (!row_similarity(erow, s0)) { status = COS_ERR_CALCULATION; break; }
            const u32 eid = erow == N ? COS_ROOT_ID : erow * ix.id_stride;
            if (lane == 0) {
                if (!EXACT) { const u32 b = eid & bitmask; vis_lds[b >> 5] |= 1u << (b & 31); }
                else {
                    atomicOr(&vis[entry >> 5], 1u << (entry & 31));
                    vlog[0] = entry >> 5;
                }
                if (wa.ef) arr[0] = pack_key(metric_key(metric, s0), entry);
            }
            if (EXACT) nlog = 1;
            npool = wa.ef ? 1u : 0u;
            wave_lds_sync();
        }

        while (npool > 0 && npop < wa.ef && !failed) {
            const u32 node = uniform_u32((u32)arr[npop]); // the best candidate: popped, it stays where it is
            npop++;
            npool--;
            const u32 limit = wa.ef - npop; // future pops still allowed = the candidates worth keeping
            for (u32 s0 = 0; s0 < slots && !failed; s0 += 64) {
                // neighbour slots s0 .. s0 + 63 in slot order, one per lane (vector_store.rs:1161-1171)
                const u32 slot = s0 + (u32)lane;
                const bool in = slot < slots;
                const u64 off = (u64)node * M + (in ? slot : slots - 1u);
                u32 nb_vec = lv.adj_vec[off];
                u32 nb_node = level == 0 ? nb_vec : lv.adj_node[off];
                nb_vec = in ? nb_vec : ROW_EMPTY;
                const u64 vmask = ballot64(nb_vec != ROW_EMPTY);
                u64 wmask;
                if (!EXACT) {
                    // PerformantFixedSet: bucket=(id>>6)&(M-1), bit=id&63  <=> linear bit id & (64M-1)
                    const u32 id = nb_vec == N ? COS_ROOT_ID : nb_vec * ix.id_stride;
                    const u32 bit = id & bitmask;
                    const u32 word = bit >> 5, msk = 1u << (bit & 31);
                    const u32 seen = vis_lds[word];
                    const u64 cmask = vmask & ballot64((seen & msk) == 0u);
                    u32 old = 0;
                    if (__builtin_amdgcn_inverse_ballot_w64(cmask)) old = atomicOr(&vis_lds[word], msk);
                    u64 lostmask = cmask & ballot64((old & msk) != 0u);
                    wmask = cmask & ~lostmask;
                    // two slots of this pass alias the same residue: the LOWER slot wins (sequential scan order)
                    while (lostmask) {
                        const int l = __ffsll((long long)lostmask) - 1;
                        const u64 g = cmask & ballot64(bit == readlane_u32(bit, l));
                        wmask = (wmask & ~g) | (g & (0ull - g));
                        lostmask &= ~g;
                    }
                    wave_lds_sync();
                } else {
                    u32 w = 0;
                    if (__builtin_amdgcn_inverse_ballot_w64(vmask)) w = __hip_atomic_load(&vis[nb_node >> 5], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    wmask = vmask & ballot64(((w >> (nb_node & 31)) & 1u) == 0u);
                    if (__builtin_amdgcn_inverse_ballot_w64(wmask)) {
                        atomicOr(&vis[nb_node >> 5], 1u << (nb_node & 31));
                        vlog[nlog + (u32)__popcll(wmask & ((1ull << lane) - 1ull))] = nb_node >> 5; // undo log, in slot order
                    }
                    nlog += (u32)__popcll(wmask);
                }
                lev_evals += (u32)__popcll(wmask);
                // the pass's winners in slot order: evaluate, insert
                while (wmask) {
                    const int l = __ffsll((long long)wmask) - 1;
                    wmask &= wmask - 1ull;
                    const u32 row = readlane_u32(nb_vec, l), nd = readlane_u32(nb_node, l);
                    float sim;
                    if (!row_similarity(row, sim)) { failed = 1; break; }
                    if (limit == 0u) continue;
                    const u64 key = pack_key(metric_key(metric, sim), nd);
                    const u64 *live = arr + npop;
                    u32 lo = 0, hi = npool; // candidates greater than the key: [0, lo)
                    while (lo < hi) {
                        const u32 mid = (lo + hi) >> 1;
                        const u64 pm = live[mid]; // (the same LDS word in every lane)
                        if (pm > key) lo = mid + 1u; else hi = mid;
                    }
                    const u32 pos = uniform_u32(lo);
                    if (pos >= limit) continue; // `limit` better candidates already wait: it could never be popped
                    const u32 newn = npool + 1u < limit ? npool + 1u : limit;
                    // candidates [pos, newn - 1) move up by one, the highest 64 first (the last one drops when the array is full)
                    u64 *lw = arr + npop;
                    for (u32 end = newn - 1u; end > pos;) {
                        const u32 cnt = end - pos < 64u ? end - pos : 64u, start = end - cnt;
                        const u64 v = (u32)lane < cnt ? lw[start + (u32)lane] : 0ull;
                        wave_lds_sync();
                        if ((u32)lane < cnt) lw[start + (u32)lane + 1u] = v;
                        wave_lds_sync();
                        end = start;
                    }
                    if (lane == 0) lw[pos] = key;
                    wave_lds_sync();
                    npool = newn;
                }
            }
        }
        n_exp += npop;
        adj_bytes += (u64)npop * slots * 4;
        n_evals += lev_evals;
        if (EXACT) { // undo: zero exactly the words this level set (walk_kernel.inc)
            __atomic_signal_fence(__ATOMIC_SEQ_CST);
            __builtin_amdgcn_s_waitcnt(0);
            for (u32 i = lane; i < nlog; i += 64) {
                const u32 w = __hip_atomic_load(&vlog[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(&vis[w], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            __atomic_signal_fence(__ATOMIC_SEQ_CST);
            __builtin_amdgcn_s_waitcnt(0);
        }
        if (failed) { status = COS_ERR_CALCULATION; break; }

        // keep the best `keep` of the popped entries, sorted descending (vector_store.rs:1194-1201): repeated selection of the largest
        // key still in the list (keys are distinct: a node enters a level's candidates once)
        u32 cnt = npop < wa.keep ? npop : wa.keep;
        const u64 obase = ((u64)qi * (L + 1) + out_slot) * wa.keep;
        u32 best_node = 0;
        if (npop == 0) { // only if ef == 0: the entry node's own distance (vector_store.rs:329-380)
            const u32 erow = uniform_u32(lv.node_vec ? lv.node_vec[entry] : entry);
            float s0;
            if (!row_similarity(erow, s0)) { status = COS_ERR_CALCULATION; break; }
            if (lane == 0) {
                wa.out_ids[obase] = erow == N ? COS_ROOT_ID : erow * ix.id_stride;
                wa.out_sims[obase] = metric_key_inv(metric, metric_key(metric, s0));
                if (wa.out_nodes) wa.out_nodes[obase] = entry;
            }
            best_node = entry;
            cnt = 1;
        }
        for (u32 k = 0; k < cnt && npop > 0; k++) {
            u64 mx = 0ull;
            u32 mi = 0;
            for (u32 i = lane; i < npop; i += 64) {
                const u64 v = arr[i];
                if (v > mx) { mx = v; mi = i; }
            }
            u64 wm = mx;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const u64 o = shfl_xor_u64(wm, d);
                wm = o > wm ? o : wm;
            }
            if (mx == wm && mx != 0ull) arr[mi] = 0ull; // its owner takes it out of the list
            wave_lds_sync();
            const u32 nd = (u32)wm;
            if (k == 0) best_node = nd;
            if (lane == 0) {
                const u32 vrow = lv.node_vec ? lv.node_vec[nd] : nd;
                wa.out_ids[obase + k] = vrow == N ? COS_ROOT_ID : vrow * ix.id_stride;
                wa.out_sims[obase + k] = metric_key_inv(metric, (u32)(wm >> 32));
                if (wa.out_nodes) wa.out_nodes[obase + k] = nd;
            }
        }
        if (lane == 0) wa.out_counts[(u64)qi * (L + 1) + out_slot] = cnt;
        // descend through the best hit's child link (vector_store.rs:382-385)
        if (level > 0) entry = uniform_u32(lv.child[uniform_u32(best_node)]);
    }

    if (lane == 0) {
        wa.out_status[qi] = status;
        if (wa.out_stats) {
            wa.out_stats[(u64)qi * 4 + 0] = n_evals;
            wa.out_stats[(u64)qi * 4 + 1] = n_exp;
            wa.out_stats[(u64)qi * 4 + 2] = adj_bytes;
            wa.out_stats[(u64)qi * 4 + 3] = n_exp;
        }
        if (wa.out_stats2) {
            wa.out_stats2[(u64)qi * 4 + 0] = n_evals;
            wa.out_stats2[(u64)qi * 4 + 1] = n_exp;
            wa.out_stats2[(u64)qi * 4 + 2] = adj_bytes;
            wa.out_stats2[(u64)qi * 4 + 3] = 0ull;
        }
    }
}

template <int ENG>
hipError_t launch_general_eng(const IndexDev &ix, const WalkArgs &wa, size_t smem, hipStream_t st) {
    dim3 grid(wa.B), block(64);
    hipError_t e;
    if (ix.visited_mode != 0) {
        e = hipFuncSetAttribute((const void *)walk_general_kernel<ENG, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL((walk_general_kernel<ENG, true>), grid, block, smem, st, ix, wa);
    } else {
        e = hipFuncSetAttribute((const void *)walk_general_kernel<ENG, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL((walk_general_kernel<ENG, false>), grid, block, smem, st, ix, wa);
    }
    return hipGetLastError();
}

} // namespace

namespace cosdev {

// the launches walk_kernel and the latency kernels cannot hold (engine.hip: cos_index_create's domain checks follow this)
bool walk_general_needed(const IndexDev &ix, u32 ef) {
    if (ef > WALK_FAST_MAX_EF) return true;
    // code rows wider than walk_kernel's chunk passes: u8 above 4096 dims (four passes of 64 lanes x 16 B), SubByte above 64 chunks
    if (ix.nchunks != 0u && ix.G != 0u && (ix.nchunks + ix.G - 1u) / ix.G > (ix.storage == 0u /* COS_STORAGE_U8 */ ? 4u : 1u)) return true;
    for (u32 l = 0; l <= ix.num_layers; l++)
        if (std::min(ix.lv[l].M, ix.shortlist) > 64u) return true;
    return false;
}

hipError_t launch_walk_general(int eng, const IndexDev &ix, const WalkArgs &wa_in, hipStream_t st) {
    if (wa_in.phase != 0u || wa_in.ef > WALK_GENERAL_MAX_EF) return hipErrorInvalidValue;
    WalkArgs wa = wa_in;
    u32 mmax = 0;
    for (u32 l = 0; l <= ix.num_layers; l++) mmax = std::max(mmax, ix.lv[l].M);
    wa.smem_mmax = mmax;
    wa.tab = nullptr;
    const size_t qbytes = eng == ENG_F16 ? (size_t)ix.dim * 4 : (size_t)ix.row_stride;
    const size_t smem = (size_t)std::max(wa.ef, 1u) * 8 + (size_t)mmax * 8 + 16 + ((qbytes + 15) & ~(size_t)15) + 16;
    switch (eng) {
    case ENG_U8: return launch_general_eng<ENG_U8>(ix, wa, smem, st);
    case ENG_Q2: return launch_general_eng<ENG_Q2>(ix, wa, smem, st);
    case ENG_Q1: return launch_general_eng<ENG_Q1>(ix, wa, smem, st);
    case ENG_Q3: return launch_general_eng<ENG_Q3>(ix, wa, smem, st);
    case ENG_F32: return launch_general_eng<ENG_F32>(ix, wa, smem, st);
    case ENG_F16: return launch_general_eng<ENG_F16>(ix, wa, smem, st);
    default: return hipErrorInvalidValue;
    }
}

} // namespace cosdev
