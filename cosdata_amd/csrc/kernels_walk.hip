// kernels_walk.hip — gfx950 kernels of the dense search path:
//   quantize_rows  : ScalarQuantization::quantize           (quantization/scalar.rs:10-52)
//   walk           : ann_search / traverse_find_nearest     (vector_store.rs:256-402, 1112-1204)
//   finalize       : remove_duplicates_and_filter + finalize_ann_results
//                                                           (models/common.rs:381-412, vector_store.rs:404-445)
// One wavefront (64 lanes) owns one query: 64 lanes <-> the <=64 neighbour slots of an expansion
// (level_0_neighbors_count = shortlist_size = 64, config.toml:21,32).  HBM-bound gather work:
// adjacency rows and code rows are read with coalesced 16-byte-per-lane loads; the candidate pool
// lives in registers, the visited filter and the popped list in LDS.  No MFMA here by design.
//
// Compile with -ffp-contract=off: the reference's arithmetic order (fused only where AVX2 FMA is
// used, x86_64.rs:418-444) is reproduced with explicit __fmaf_rn / __fmul_rn / __fadd_rn.
#include <hip/hip_runtime.h>

#include <algorithm>
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

// ------------------------------------------------------------------------------------------------
// quantize_rows: one wave per row.  Writes the DEVICE code layout:
//   U8  : dim bytes, zero padded to a multiple of 16
//   Q2  : per 64-dim chunk 16 bytes = [plane0 (MSB as quantize_to_u8_bits stores it) 8 B | plane1 8 B]
//   F32 : dim floats, zero padded to a multiple of 16 bytes
// ------------------------------------------------------------------------------------------------
template <int ENG>
__global__ __launch_bounds__(256) void quantize_rows_kernel(const float *__restrict__ x, u64 x_stride, u32 n, u32 dim, float lo,
                                                            float hi, uint8_t *__restrict__ codes, u64 row_stride,
                                                            float *__restrict__ mags, float *__restrict__ raw_mags) {
    const int lane = threadIdx.x & 63;
    const u32 row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= n) return;
    const float *xr = x + (u64)row * x_stride;
    uint8_t *cr = codes + (u64)row * row_stride;
    float rn = 0.0f;
    const bool need_seq = (raw_mags != nullptr) || (ENG != ENG_U8);
    if (need_seq) rn = seq_norm_wave(xr, dim, lane); // wave-uniform branch; every lane gets the value
    if (raw_mags && lane == 0) raw_mags[row] = rn;
    if constexpr (ENG == ENG_U8) {
        u32 ss = 0;
        for (u32 i = lane; i < (u32)row_stride; i += 64) {
            uint8_t q = 0;
            if (i < dim) {
                float c = rust_min(rust_max(xr[i], lo), hi);
                float v = __fmul_rn(__fdiv_rn(__fsub_rn(c, lo), __fsub_rn(hi, lo)), 255.0f);
                q = rust_f32_as_u8(v);
            }
            cr[i] = q;
            ss += (u32)q * (u32)q;
        }
#pragma unroll
        for (int m = 32; m > 0; m >>= 1) ss += (u32)__shfl_xor((int)ss, m, 64);
        if (lane == 0) mags[row] = sqrtf((float)ss); // (sum::<u32>() as f32).sqrt()
    } else if constexpr (ENG == ENG_Q2) {
        const u32 nch = (u32)(row_stride / 16);
        for (u32 c = 0; c < nch; c++) {
            u32 i = c * 64 + lane;
            u32 lvl = 0;
            if (i < dim) lvl = rust_f32_as_usize_low2(floorf(__fdiv_rn(__fadd_rn(xr[i], 1.0f), 0.5f))); // step = 2/4
            u64 p0 = __ballot((lvl >> 1) & 1u); // plane 0 = MSB (common.rs:230-233)
            u64 p1 = __ballot(lvl & 1u);
            if (lane == 0) {
                *(u64 *)(cr + (u64)c * 16) = p0;
                *(u64 *)(cr + (u64)c * 16 + 8) = p1;
            }
        }
        if (lane == 0) mags[row] = rn; // norm of the ORIGINAL vector (scalar.rs:31-32)
    } else if constexpr (ENG == ENG_Q1) {
        // binary: one plane, level = floor((x+1)/1.0) & 1 (common.rs:226-275 with resolution 1); 128 dims per 16 B chunk
        const u32 nch = (u32)(row_stride / 16);
        for (u32 c = 0; c < 2 * nch; c++) {
            const u32 i = c * 64 + lane;
            u32 lvl = 0;
            if (i < dim) lvl = rust_f32_as_usize_low(floorf(__fdiv_rn(__fadd_rn(xr[i], 1.0f), 1.0f)), 1u);
            const u64 p0 = __ballot(lvl & 1u);
            if (lane == 0) *(u64 *)(cr + (u64)c * 8) = p0;
        }
        if (lane == 0) mags[row] = rn;
    } else if constexpr (ENG == ENG_Q3) {
        // octal: three planes, MSB first; 32 dims per 16 B chunk [p0 | p1 | p2 | 0]; one ballot covers two chunks
        const u32 nch = (u32)(row_stride / 16);
        for (u32 c = 0; c < nch; c += 2) {
            const u32 i = c * 32 + lane;
            u32 lvl = 0;
            if (i < dim) lvl = rust_f32_as_usize_low(floorf(__fdiv_rn(__fadd_rn(xr[i], 1.0f), 0.25f)), 7u); // step = 2/8
            const u64 p0 = __ballot((lvl >> 2) & 1u), p1 = __ballot((lvl >> 1) & 1u), p2 = __ballot(lvl & 1u);
            if (lane == 0) {
                *(uint4 *)(cr + (u64)c * 16) = make_uint4((u32)p0, (u32)p1, (u32)p2, 0u);
                if (c + 1 < nch) *(uint4 *)(cr + (u64)(c + 1) * 16) = make_uint4((u32)(p0 >> 32), (u32)(p1 >> 32), (u32)(p2 >> 32), 0u);
            }
        }
        if (lane == 0) mags[row] = rn;
    } else if constexpr (ENG == ENG_F16) {
        __half *ch = (__half *)cr;
        for (u32 i = lane; i < (u32)(row_stride / 2); i += 64) ch[i] = i < dim ? __float2half_rn(xr[i]) : __float2half_rn(0.0f); // f16::from_f32 (RNE)
        if (lane == 0) mags[row] = rn; // norm of the ORIGINAL vector (scalar.rs:41)
    } else {
        float *cf = (float *)cr;
        for (u32 i = lane; i < (u32)(row_stride / 4); i += 64) cf[i] = i < dim ? xr[i] : 0.0f;
        if (lane == 0) mags[row] = rn;
    }
}

#define COS_WALK_KERNEL_NAME walk_kernel
#include "walk_kernel.inc"

// ------------------------------------------------------------------------------------------------
// finalize: one wave per query.
//   remove_duplicates_and_filter: first-seen dedup, drop root, sort desc by quantized score, keep 5k
//   finalize_ann_results: cs = dot_f32(q, raw) / (|q| * |raw|) in the reference order, sort desc, top k
// ------------------------------------------------------------------------------------------------
struct FinalizeArgs {
    const float *queries; // raw f32 [B][q_stride]
    u64 q_stride;
    const float *q_raw_mags; // [B] sqrt(seq sum q*q)
    const u32 *walk_ids;     // [B][L+1][100]
    const float *walk_sims;
    const u32 *walk_counts;  // [B][L+1]
    const int32_t *walk_status;
    u32 B, top_k;
    u32 *out_ids;     // [B][top_k] (id_base + local id)
    float *out_scores;
    u32 *out_counts;
    int32_t *out_status;
    u64 *out_rerank_rows; // [B]
    const u32 *q_order;   // optional [B]: workgroup -> query, the locality order the walk of this launch ended with (engine_types.h):
                          // neighbouring queries rerank many of the same raw rows
    u32 *slow_list;       // finalize_fast_kernel: [B] queries it leaves to finalize_list_kernel (more than FAST_CAP survivors) ...
    u32 *slow_count;      // ... and how many (zeroed before the launch)
};

// Small launches (one client batch: 256 workgroups on 256 CUs) rerank with NWV waves per query.  One wave streams its 5k raw rows
// (3 KB each at 768 dims) with ~8 KB in flight: ~24 dependent trips to memory for 50 rows, 60 for the 150 rows of a hybrid query's
// top_k x 3 (0.05 / 0.16 ms per batch: profiles/r06_final_kernel_trace_c5_rocprofv3.txt).  With NWV > 1 wave 0 selects the candidates
// as before, every wave then dots 8 of them per pass with the eight-lane chains of f32_oct_dot (the pair version's bits, dot_engines.h)
// and wave 0 sorts the keys the waves left in LDS: same lists, same bits.
template <int NWV>
__device__ __forceinline__ void finalize_sync() {
    if constexpr (NWV > 1) __syncthreads();
    else { __builtin_amdgcn_s_waitcnt(0); __builtin_amdgcn_wave_barrier(); }
}
// the exact rerank of candidates [0, ncand) by every wave of a wide workgroup (vector_store.rs:404-445): keys[j] = (score, id) of cand[j]
template <int NWV>
__device__ __forceinline__ void rerank_wide(const IndexDev &ix, const float *qf, const u32 *cand, u32 ncand, float mag_query, u64 *keys, int lane, int wave) {
    for (u32 base = (u32)wave * 8u; base < ncand; base += 8u * (u32)NWV) { // (wave-uniform bounds)
        const u32 my = base + (u32)(lane >> 3);
        const u32 id = my < ncand ? cand[my] : 0u;
        const u32 rrow = id / ix.id_stride; // raw embedding of an id = its base id (collection.rs:368-384)
        const float dp = f32_oct_dot(ix.raw + (u64)rrow * ix.raw_stride, qf, ix.dim, lane & 7);
        const float cs = x86_div(dp, __fmul_rn(mag_query, ix.raw_mags[rrow])); // 0/0 (zero raw vector or query) -> x86's -NaN
        if ((lane & 7) == 0 && my < ncand) keys[my] = pack_key(simkey(cs), id); // total_cmp desc; larger id first on ties
    }
}

template <int FR, int NWV = 1>
__device__ __forceinline__ void finalize_one(const IndexDev &ix, const FinalizeArgs &fa, const u32 qi, unsigned char *smem_raw) {
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const u32 L = ix.num_layers;
    const u32 metric = ix.metric;
    float *qf = (float *)smem_raw;                                   // dim floats (padded to 16 B)
    u32 *cand = (u32 *)(smem_raw + (((size_t)ix.dim * 4 + 15) & ~(size_t)15)); // 64*FR candidate ids
    u64 *wkeys = (u64 *)(cand + 64 * FR);                            // NWV > 1: 64*FR reranked keys, then [0] of the word behind them = ncand
    u32 *wmisc = (u32 *)(wkeys + 64 * FR);

    const int32_t wst = fa.walk_status[qi];
    if (wst != COS_OK) {
        if (tid == 0) { fa.out_status[qi] = wst; fa.out_counts[qi] = 0; if (fa.out_rerank_rows) fa.out_rerank_rows[qi] = 0; }
        return;
    }
    const float *q = fa.queries + (u64)qi * fa.q_stride;
    for (u32 i = tid; i < ix.dim; i += 64 * NWV) qf[i] = q[i];
    const float mag_query = fa.q_raw_mags[qi];
    if constexpr (NWV > 1) {
        if (wave != 0) { // the other waves wait for the candidates, rerank their share and are done
            __syncthreads();
            rerank_wide<NWV>(ix, qf, cand, wmisc[0], mag_query, wkeys, lane, wave);
            __syncthreads();
            return;
        }
    }

    // gather the concatenated per-level lists (top level first) as (key, id) pairs
    u64 k[FR];
#pragma unroll
    for (int r = 0; r < FR; r++) k[r] = 0ull;
    {
        u32 off = 0;
        for (u32 s = 0; s <= L; s++) {
            // a level list is sorted: entries past its first 5k+1 cannot reach the global top 5k (each is preceded by 5k
            // distinct better non-root nodes — one of the first 5k+1 may be the root, which is filtered), so only
            // min(count, 5k+1) entries per level take part in the sort
            u32 c = fa.walk_counts[(u64)qi * (L + 1) + s];
            if (c > 5u * fa.top_k + 1u) c = 5u * fa.top_k + 1u;
            const u64 b = ((u64)qi * (L + 1) + s) * KEEP_SEARCH;
            // element index off + j  lives in lane (off+j)/FR, register (off+j)%FR
#pragma unroll
            for (int r = 0; r < FR; r++) {
                const u32 e = (u32)lane * FR + r;
                if (e >= off && e < off + c) {
                    const u32 j = e - off;
                    const u32 id = fa.walk_ids[b + j];
                    const float sv = fa.walk_sims[b + j];
                    // root filtered (common.rs:397); so are the pseudo nodes of a metadata collection (common.rs:400-402)
                    const bool drop = id == COS_ROOT_ID || (ix.mdim != 0u && id >= 0xFFFFFEFEu && id <= 0xFFFFFFFDu);
                    k[r] = drop ? 0ull : pack_key(metric_key(metric, sv), id);
                }
            }
            off += c;
        }
    }
    bitonic_sort_desc<FR>(k, lane);
    // duplicates of a node carry identical (score, id) keys and are adjacent after the sort: keep the first
    const u64 prev_last = shfl_up1_u64(k[FR - 1]);
    u32 keepbits = 0;
#pragma unroll
    for (int r = 0; r < FR; r++) {
        const u64 prev = (r == 0) ? (lane == 0 ? ~0ull : prev_last) : k[r > 0 ? r - 1 : 0];
        if (k[r] != 0ull && k[r] != prev) keepbits |= 1u << r;
    }
    // exclusive prefix of kept counts across lanes
    const u32 mine = (u32)__popc(keepbits);
    u32 incl = mine;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        u32 t = (u32)__shfl_up((int)incl, d, 64);
        if (lane >= d) incl += t;
    }
    u32 pos = incl - mine;
    const u32 total = readlane_u32(incl, 63);
    const u32 want = 5u * fa.top_k; // truncate(5*k) (common.rs:409)
    const u32 ncand = total < want ? total : want;
#pragma unroll
    for (int r = 0; r < FR; r++) {
        if (keepbits & (1u << r)) {
            if (pos < ncand) cand[pos] = (u32)k[r];
            pos++;
        }
    }
    const u32 nout = ncand < fa.top_k ? ncand : fa.top_k;
    if constexpr (NWV > 1) {
        if (lane == 0) wmisc[0] = ncand;
        __syncthreads();
        rerank_wide<NWV>(ix, qf, cand, ncand, mag_query, wkeys, lane, 0);
        __syncthreads();
        u64 rk[FR];
#pragma unroll
        for (int r = 0; r < FR; r++) {
            const u32 e = (u32)lane * FR + r;
            rk[r] = e < ncand ? wkeys[e] : 0ull;
        }
        bitonic_sort_desc<FR>(rk, lane);
#pragma unroll
        for (int r = 0; r < FR; r++) {
            const u32 e = (u32)lane * FR + r;
            if (e < nout) {
                fa.out_ids[(u64)qi * fa.top_k + e] = (u32)rk[r] + ix.id_base;
                fa.out_scores[(u64)qi * fa.top_k + e] = simkey_inv((u32)(rk[r] >> 32));
            }
        }
        if (lane == 0) {
            fa.out_counts[qi] = nout;
            fa.out_status[qi] = COS_OK;
            if (fa.out_rerank_rows) fa.out_rerank_rows[qi] = ncand;
        }
        return;
    }
    __builtin_amdgcn_s_waitcnt(0);
    __builtin_amdgcn_wave_barrier();

    // exact rerank on raw f32 (vector_store.rs:404-445); 32 candidates per pass (one lane pair each)
    const int pair = lane >> 1;
    if (ncand <= 64) {
        // common case (5k <= 64): survivor j ends in lane j and one 64-key sort finishes the job
        u64 res[1] = {0ull};
        for (u32 base = 0; base < ncand; base += 32) {
            const u32 my = base + (u32)pair;
            const u32 id = my < ncand ? cand[my] : 0u;
            const u32 rrow = id / ix.id_stride; // raw embedding of an id = its base id (collection.rs:368-384)
            const float dp = f32_pair_dot(ix.raw + (u64)rrow * ix.raw_stride, qf, ix.dim, lane & 1);
            const float cs = x86_div(dp, __fmul_rn(mag_query, ix.raw_mags[rrow])); // 0/0 (zero raw vector or query) -> x86's -NaN
            const u64 key = my < ncand ? pack_key(simkey(cs), id) : 0ull; // total_cmp desc; larger id first on ties
            const int from = (2 * (lane - (int)base)) & 63;
            const u32 klo = (u32)__shfl((int)(u32)key, from, 64), khi = (u32)__shfl((int)(u32)(key >> 32), from, 64);
            if ((u32)lane >= base && (u32)lane < base + 32) res[0] = ((u64)khi << 32) | klo;
        }
        bitonic_sort_desc<1>(res, lane);
        if ((u32)lane < nout) {
            fa.out_ids[(u64)qi * fa.top_k + lane] = (u32)res[0] + ix.id_base;
            fa.out_scores[(u64)qi * fa.top_k + lane] = simkey_inv((u32)(res[0] >> 32));
        }
    } else {
        u64 rk[FR];
#pragma unroll
        for (int r = 0; r < FR; r++) rk[r] = 0ull;
        for (u32 base = 0; base < ncand; base += 32) {
            const u32 my = base + (u32)pair;
            const u32 id = my < ncand ? cand[my] : 0u;
            const u32 rrow = id / ix.id_stride;
            const float dp = f32_pair_dot(ix.raw + (u64)rrow * ix.raw_stride, qf, ix.dim, lane & 1);
            const float cs = x86_div(dp, __fmul_rn(mag_query, ix.raw_mags[rrow])); // 0/0 (zero raw vector or query) -> x86's -NaN
            const u64 key = pack_key(simkey(cs), id);
            // scatter into the blocked register layout: element my -> lane my/FR, reg my%FR
            for (u32 j = 0; j < 32 && base + j < ncand; j++) {
                const u64 kv = readlane_u64(key, (int)(2 * j));
                const u32 e = base + j;
                if ((u32)lane == e / FR) {
#pragma unroll
                    for (int r = 0; r < FR; r++)
                        if ((e % FR) == (u32)r) rk[r] = kv;
                }
            }
        }
        bitonic_sort_desc<FR>(rk, lane);
#pragma unroll
        for (int r = 0; r < FR; r++) {
            const u32 e = (u32)lane * FR + r;
            if (e < nout) {
                fa.out_ids[(u64)qi * fa.top_k + e] = (u32)rk[r] + ix.id_base;
                fa.out_scores[(u64)qi * fa.top_k + e] = simkey_inv((u32)(rk[r] >> 32));
            }
        }
    }
    if (lane == 0) {
        fa.out_counts[qi] = nout;
        fa.out_status[qi] = COS_OK;
        if (fa.out_rerank_rows) fa.out_rerank_rows[qi] = ncand;
    }
}

// grid = B: one wave per query (q_order: the launch's locality order)
template <int FR>
__global__ __launch_bounds__(64) void finalize_kernel(const IndexDev ix, const FinalizeArgs fa) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    if (blockIdx.x >= fa.B) return;
    finalize_one<FR>(ix, fa, fa.q_order ? fa.q_order[blockIdx.x] : blockIdx.x, smem_raw);
}
template <int FR, int NWV>
__global__ __launch_bounds__(64 * NWV) void finalize_wide_kernel(const IndexDev ix, const FinalizeArgs fa) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    if (blockIdx.x >= fa.B) return;
    finalize_one<FR, NWV>(ix, fa, fa.q_order ? fa.q_order[blockIdx.x] : blockIdx.x, smem_raw);
}
// the queries finalize_fast_kernel left over (slow_list / slow_count): a small fixed grid walks the list — nothing to do costs a
// launch of a few workgroups instead of B that look at a flag and leave (60 us per 32768-query launch)
template <int FR>
__global__ __launch_bounds__(64) void finalize_list_kernel(const IndexDev ix, const FinalizeArgs fa) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const u32 n = *fa.slow_count;
    for (u32 i = blockIdx.x; i < n; i += gridDim.x) {
        finalize_one<FR>(ix, fa, fa.slow_list[i], smem_raw);
        __builtin_amdgcn_s_waitcnt(0);
        __builtin_amdgcn_wave_barrier(); // the next query rewrites the LDS buffers
    }
}

// ------------------------------------------------------------------------------------------------
// finalize_fast: the same result for the usual shape of a search (5 * top_k <= 64), at a fraction of the work.
// finalize_kernel sorts the first 5k + 1 entries of EVERY level list — 510 keys at ten levels, a 512-key bitonic network in 121
// VGPRs (4 waves per SIMD) — to keep the best 5k distinct ones.  But level 0's own list already holds 5k + 1 distinct nodes (at most
// one of them the root): an entry of any list below level 0's (5k + 1)-th key has 5k distinct non-root entries above it and can never
// be kept.  So: T = that key; the entries >= T of all lists — level 0's 5k + 1 and the few upper-level entries that are among the
// query's nearest (the same nodes at the same scores: duplicates) — are compacted into LDS, typically ~70 of them; up to 128 are sorted
// by a 128-key network (two keys per lane), deduplicated and reranked exactly as below.  More than 128 survivors (small graphs, tiny
// ef: level 0 has no (5k + 1)-th entry and nothing is screened) leave the query to finalize_list_kernel through slow_list.
// ------------------------------------------------------------------------------------------------
constexpr int FAST_CAP = 128;
template <int NWV>
__global__ __launch_bounds__(64 * NWV) __attribute__((amdgpu_waves_per_eu(NWV == 1 ? 7 : 2, 8))) void finalize_fast_kernel(const IndexDev ix, const FinalizeArgs fa) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    if (blockIdx.x >= fa.B) return;
    const u32 qi = fa.q_order ? fa.q_order[blockIdx.x] : blockIdx.x;
    const u32 L = ix.num_layers;
    const u32 metric = ix.metric;
    float *qf = (float *)smem_raw;                                                       // dim floats (padded to 16 B)
    u64 *surv = (u64 *)(smem_raw + (((size_t)ix.dim * 4 + 15) & ~(size_t)15));           // FAST_CAP survivor keys
    u32 *cand = (u32 *)(surv + FAST_CAP);                                                // 64 candidate ids, then one word: ncand (NWV > 1)

    const int32_t wst = fa.walk_status[qi];
    if (wst != COS_OK) {
        if (tid == 0) { fa.out_status[qi] = wst; fa.out_counts[qi] = 0; if (fa.out_rerank_rows) fa.out_rerank_rows[qi] = 0; }
        return;
    }
    const float *q = fa.queries + (u64)qi * fa.q_stride;
    for (u32 i = tid; i < ix.dim; i += 64 * NWV) qf[i] = q[i];
    const float mag_query = fa.q_raw_mags[qi];
    if constexpr (NWV > 1) { // (finalize_one: the other waves wait for the candidates and rerank their share; keys land where the survivors were)
        if (wave != 0) {
            __syncthreads();
            const u32 nc = cand[64];
            if (nc != 0xFFFFFFFFu) rerank_wide<NWV>(ix, qf, cand, nc, mag_query, surv, lane, wave);
            __syncthreads();
            return;
        }
    }

    const u32 want = 5u * fa.top_k, per = want + 1u; // truncate(5k) (common.rs:409); entries of a level that can matter
    // counts of every level in one load; the threshold from level 0's list (slot L)
    const u32 cnt_l = (u32)lane <= L ? fa.walk_counts[(u64)qi * (L + 1) + lane] : 0u;
    const u32 c0 = readlane_u32(cnt_l, (int)L);
    const u64 b0 = ((u64)qi * (L + 1) + L) * KEEP_SEARCH;
    u64 T = 0ull;
    if (c0 >= per) T = pack_key(metric_key(metric, uniform_f32(fa.walk_sims[b0 + want])), uniform_u32(fa.walk_ids[b0 + want]));
    // every level's first min(count, 5k + 1) entries, one per lane (5k + 1 <= 64): loads of all levels first, then the screen
    u32 n = 0;
    bool overflow = false;
    for (u32 s0 = 0; s0 <= L; s0 += 4) { // four levels' loads in flight
        u32 idv[4];
        float smv[4];
        u32 cc[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const u32 s = s0 + (u32)u;
            u32 c = s <= L ? readlane_u32(cnt_l, (int)(s <= L ? s : L)) : 0u;
            c = c > per ? per : c;
            cc[u] = c;
            const u64 b = ((u64)qi * (L + 1) + (s <= L ? s : L)) * KEEP_SEARCH;
            const u32 j = (u32)lane < c ? (u32)lane : 0u;
            idv[u] = fa.walk_ids[b + j];
            smv[u] = fa.walk_sims[b + j];
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const u32 id = idv[u];
            // root filtered (common.rs:397); so are the pseudo nodes of a metadata collection (common.rs:400-402)
            const bool drop = id == COS_ROOT_ID || (ix.mdim != 0u && id >= 0xFFFFFEFEu && id <= 0xFFFFFFFDu);
            const u64 key = pack_key(metric_key(metric, smv[u]), id);
            const u64 keep = ballot64((u32)lane < cc[u] && !drop && key >= T);
            const u32 nk = (u32)__popcll(keep);
            if (n + nk > (u32)FAST_CAP) { overflow = true; break; }
            if (__builtin_amdgcn_inverse_ballot_w64(keep)) surv[n + (u32)__popcll(keep & ((1ull << lane) - 1ull))] = key;
            n += nk;
        }
        if (overflow) break;
    }
    if (overflow) {
        if (lane == 0) fa.slow_list[atomicAdd(fa.slow_count, 1u)] = qi;
        if constexpr (NWV > 1) {
            if (lane == 0) cand[64] = 0xFFFFFFFFu;
            __syncthreads();
            __syncthreads();
        }
        return;
    }
    __builtin_amdgcn_s_waitcnt(0);
    __builtin_amdgcn_wave_barrier();
    u64 k[2];
#pragma unroll
    for (int r = 0; r < 2; r++) {
        const u32 e = (u32)lane * 2u + (u32)r;
        k[r] = e < n ? surv[e] : 0ull;
    }
    bitonic_sort_desc<2>(k, lane);
    // duplicates of a node carry identical (score, id) keys and are adjacent after the sort: keep the first
    const u64 prev_last = shfl_up1_u64(k[1]);
    const bool keep0 = k[0] != 0ull && (lane == 0 || k[0] != prev_last), keep1 = k[1] != 0ull && k[1] != k[0];
    const u32 mine = (keep0 ? 1u : 0u) + (keep1 ? 1u : 0u);
    u32 incl = mine;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const u32 t = (u32)__shfl_up((int)incl, d, 64);
        if (lane >= d) incl += t;
    }
    u32 pos = incl - mine;
    const u32 total = readlane_u32(incl, 63);
    const u32 ncand = total < want ? total : want;
    if (keep0) { if (pos < ncand) cand[pos] = (u32)k[0]; pos++; }
    if (keep1) { if (pos < ncand) cand[pos] = (u32)k[1]; }
    const u32 nout = ncand < fa.top_k ? ncand : fa.top_k;
    u64 res[1] = {0ull};
    if constexpr (NWV > 1) {
        if (lane == 0) cand[64] = ncand;
        __syncthreads(); // (the survivors were read into registers before the sort: their LDS takes the reranked keys)
        rerank_wide<NWV>(ix, qf, cand, ncand, mag_query, surv, lane, 0);
        __syncthreads();
        res[0] = (u32)lane < ncand ? surv[lane] : 0ull;
    } else {
    __builtin_amdgcn_s_waitcnt(0);
    __builtin_amdgcn_wave_barrier();

    // exact rerank on raw f32 (vector_store.rs:404-445); 32 candidates per pass (one lane pair each); survivor j ends in lane j and
    // one 64-key sort finishes the job (5k <= 64)
    const int pair = lane >> 1;
    for (u32 base = 0; base < ncand; base += 32) {
        const u32 my = base + (u32)pair;
        const u32 id = my < ncand ? cand[my] : 0u;
        const u32 rrow = id / ix.id_stride; // raw embedding of an id = its base id (collection.rs:368-384)
        const float dp = f32_pair_dot(ix.raw + (u64)rrow * ix.raw_stride, qf, ix.dim, lane & 1);
        const float cs = x86_div(dp, __fmul_rn(mag_query, ix.raw_mags[rrow])); // 0/0 (zero raw vector or query) -> x86's -NaN
        const u64 key = my < ncand ? pack_key(simkey(cs), id) : 0ull; // total_cmp desc; larger id first on ties
        const int from = (2 * (lane - (int)base)) & 63;
        const u32 klo = (u32)__shfl((int)(u32)key, from, 64), khi = (u32)__shfl((int)(u32)(key >> 32), from, 64);
        if ((u32)lane >= base && (u32)lane < base + 32) res[0] = ((u64)khi << 32) | klo;
    }
    }
    bitonic_sort_desc<1>(res, lane);
    if ((u32)lane < nout) {
        fa.out_ids[(u64)qi * fa.top_k + lane] = (u32)res[0] + ix.id_base;
        fa.out_scores[(u64)qi * fa.top_k + lane] = simkey_inv((u32)(res[0] >> 32));
    }
    if (lane == 0) {
        fa.out_counts[qi] = nout;
        fa.out_status[qi] = COS_OK;
        if (fa.out_rerank_rows) fa.out_rerank_rows[qi] = ncand;
    }
}

} // namespace

// ------------------------------------------------------------------------------------------------
// host-side launchers (called from engine.hip)
// ------------------------------------------------------------------------------------------------
namespace cosdev {

hipError_t launch_quantize_rows(int eng, const float *x, u64 x_stride, u32 n, u32 dim, float lo, float hi, uint8_t *codes,
                                u64 row_stride, float *mags, float *raw_mags, hipStream_t st) {
    if (n == 0) return hipSuccess;
    dim3 block(256), grid((n + 3) / 4);
    switch (eng) {
    case ENG_U8: hipLaunchKernelGGL(quantize_rows_kernel<ENG_U8>, grid, block, 0, st, x, x_stride, n, dim, lo, hi, codes, row_stride, mags, raw_mags); break;
    case ENG_Q2: hipLaunchKernelGGL(quantize_rows_kernel<ENG_Q2>, grid, block, 0, st, x, x_stride, n, dim, lo, hi, codes, row_stride, mags, raw_mags); break;
    case ENG_F32: hipLaunchKernelGGL(quantize_rows_kernel<ENG_F32>, grid, block, 0, st, x, x_stride, n, dim, lo, hi, codes, row_stride, mags, raw_mags); break;
    case ENG_F16: hipLaunchKernelGGL(quantize_rows_kernel<ENG_F16>, grid, block, 0, st, x, x_stride, n, dim, lo, hi, codes, row_stride, mags, raw_mags); break;
    case ENG_Q1: hipLaunchKernelGGL(quantize_rows_kernel<ENG_Q1>, grid, block, 0, st, x, x_stride, n, dim, lo, hi, codes, row_stride, mags, raw_mags); break;
    case ENG_Q3: hipLaunchKernelGGL(quantize_rows_kernel<ENG_Q3>, grid, block, 0, st, x, x_stride, n, dim, lo, hi, codes, row_stride, mags, raw_mags); break;
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

// LDS of one walk_kernel wave (walk_kernel.inc, WalkSmem): filter | popped list | winners | window | f32 / f16 query
size_t walk_smem_bytes(const IndexDev &ix, u32 ef, int eng, u32 mmax, u32 win_bytes) {
    size_t b = (size_t)mmax * 8 + (size_t)ef * 8 + 64 * 4 * 2 + (size_t)win_bytes;
    b = (b + 15) & ~(size_t)15;
    if (eng == ENG_F32) b += (size_t)ix.row_stride;
    if (eng == ENG_F16) b += ((size_t)ix.dim * 4 + 15) & ~(size_t)15;
    return b + 16;
}

// rows in flight per wave on the G = 64 u8 path (see PB64 above); tuning knob walk_pb = 4|8 overrides the default (experiments).
// The widest pool (ef > 256: 16 VGPRs of pool) with eight row buffers needs 110 VGPRs = 4 waves per SIMD; with four it fits 5 waves
// without spills and measured 3.3 % faster (c2 at ef 512: 50.9 vs 52.6 ms per 32768 queries, profiles/archive/r03_order_probe_c2_ef512_pb4.jsonl).
// The UPPER range of a split walk (tuning knob walk_pb_upper = 4|8 overrides): with the level table that range is almost all table
// levels, which fetch no rows (c2: 91 M table evaluations against 2 M row evaluations), and the four-buffer variant leaves more waves
// per SIMD.  Measured in round 5 (profiles/r05_candidates_walk_pb_upper.txt, c2): ef 64 2.798 against 2.799 ms (nothing: 8 stays),
// ef 256 10.47 against 11.33 ms per 32 768 queries (+4.4 % QPS on the whole step): four buffers above ef 64.
static int walk_pb_policy(u32 B, u32 ef, bool upper_range_of_a_split_walk) {
    const int forced = (int)tune_or(TUNE_WALK_PB, 0), forced_upper = (int)tune_or(TUNE_WALK_PB_UPPER, 0);
    if (upper_range_of_a_split_walk && (forced_upper == 4 || forced_upper == 8)) return forced_upper;
    if (forced == 4 || forced == 8) return forced;
    if (upper_range_of_a_split_walk && ef > 64u) return 4;
    return ef > 256u ? 4 : 8;
}

// Table levels (walk_kernel.inc, commit_merge): winners past the screen from which an expansion takes the ranked merge instead of
// the serial insert.  Tuning knob walk_merge_min overrides (0 = never merge).  Defaults measured in round 6
// (profiles/r06_merge_probe_*.jsonl): one key per lane (ef <= 64) the serial insert is a dozen instructions and only a big batch
// of winners pays for the permutation; from two keys per lane on the merge wins from three winners.
static u32 walk_merge_min_policy(u32 ef) {
    const long long forced = tune_or(TUNE_WALK_MERGE_MIN, -1);
    if (forced >= 0) return (u32)std::min<long long>(forced, 65);
    return ef <= 64u ? 4u : 3u;
}

template <int ENG, int CH, bool G64>
static hipError_t launch_walk_r(const IndexDev &ix, const WalkArgs &wa_in, hipStream_t st) {
    WalkArgs wa = wa_in;
    wa.merge_min = wa.tab != nullptr ? walk_merge_min_policy(wa.ef) : 0u;
    // the LDS of a wave follows the levels THIS launch walks: the filter is as wide as their widest row, and a launch of table levels only
    // (the upper range of a split walk whose cut level is in the table) needs no window staging — just the ranked merge's scratch.  On the
    // metric's shard (M0 256, ef 112) that is 2.9 KB per wave instead of 6.6: the upper range's 60-register waves were capped at 6 per
    // SIMD by LDS, and that range is bound by the gathers in flight (fewer resident waves = proportionally slower:
    // profiles/r06_pad_probe_c4shard.jsonl) — now 8.
    const u32 lf = wa.phase ? wa.level_first : ix.num_layers, ll = wa.phase ? wa.level_last : 0u;
    u32 mmax = 1;
    for (u32 l = ll; l <= lf; l++) mmax = std::max(mmax, ix.lv[l].M);
    constexpr bool TABLE_ENG = ENG == ENG_U8 || ENG == ENG_Q2;
    const bool all_tab = TABLE_ENG && wa.tab != nullptr && ll >= wa.tab_level_min && tune_or(TUNE_WALK_UPPER_LDS_PAD, 0) >= 0;
    wa.smem_mmax = mmax;
    size_t pad = 0;
    // experiment knob: extra (unused) dynamic LDS per wave on the upper range of a split walk = fewer resident waves per CU (negative: the
    // full layout even for table-only launches)
    if (wa.phase != 0u && wa.level_last >= 1u) pad = (size_t)std::max<long long>(0, std::min<long long>(tune_or(TUNE_WALK_UPPER_LDS_PAD, 0), 48 << 10));
    dim3 grid(wa.B), block(64);
    const bool exact = ix.visited_mode != 0;
    constexpr bool HAS_PB8 = ENG == ENG_U8 && G64 && CH == 1; // the headline path (u8, 513..1024 dims)
    const bool pb8 = HAS_PB8 && walk_pb_policy(wa.B, wa.ef, wa.phase != 0u && wa.level_last >= 1u) == 8;
#define WALK(R_)                                                                                                          \
    do {                                                                                                                  \
        wa.smem_win_bytes = (all_tab && (R_) <= 4) ? (u32)(((64 * (R_) + 1) * 8 + 15) & ~15) : (u32)(LA * 64 * 4 * 3);         \
        const size_t smem = walk_smem_bytes(ix, wa.ef, ENG, mmax, wa.smem_win_bytes) + pad;                                \
        if constexpr (HAS_PB8) {                                                                                          \
            if (pb8) {                                                                                                    \
                if (exact) hipLaunchKernelGGL((walk_kernel<ENG, CH, R_, G64, true, 8>), grid, block, smem, st, ix, wa);   \
                else hipLaunchKernelGGL((walk_kernel<ENG, CH, R_, G64, false, 8>), grid, block, smem, st, ix, wa);        \
                break;                                                                                                    \
            }                                                                                                             \
        }                                                                                                                 \
        if (exact) hipLaunchKernelGGL((walk_kernel<ENG, CH, R_, G64, true>), grid, block, smem, st, ix, wa);              \
        else hipLaunchKernelGGL((walk_kernel<ENG, CH, R_, G64, false>), grid, block, smem, st, ix, wa);                   \
    } while (0)
    if (wa.ef <= 64) WALK(1);
    else if (wa.ef <= 128 && tune_or(TUNE_WALK_R2, 1) != 0) WALK(2); // round 6: the metric's shard runs at ef 128 — half the pool registers, half the insert
    else if (wa.ef <= 256) WALK(4);
    else if (wa.ef <= 512) WALK(8);
    else if (wa.ef <= 1024) WALK(16); // round 5: a pool of 64 x 16 keys per wave (103 VGPRs, no scratch: 4 waves per SIMD)
    else return hipErrorInvalidValue;
#undef WALK
    return hipGetLastError();
}

// kernels_walk_lat.hip
bool walk_lat_applicable(int eng, const IndexDev &ix, const WalkArgs &wa, u32 max_B);
hipError_t launch_walk_lat(int eng, const IndexDev &ix, const WalkArgs &wa, hipStream_t st);

// kernels_walk_lat4.hip
bool walk_lat4_applicable(int eng, const IndexDev &ix, const WalkArgs &wa, u32 max_B);
hipError_t launch_walk_lat4(int eng, const IndexDev &ix, const WalkArgs &wa, hipStream_t st);
// kernels_walk_general.hip
bool walk_general_needed(const IndexDev &ix, u32 ef);
hipError_t launch_walk_general(int eng, const IndexDev &ix, const WalkArgs &wa, hipStream_t st);

// lat_max_B: launches of at most this many queries take the latency kernel where it applies (cos_index_set_latency_mode; 0 = never);
// lat4_max_B: the smallest of them give every query four waves (cos_index_set_latency_waves; 0 = never).
// tuning knobs walk_lat / walk_lat4 = <n> override the handle's values (experiments: 0 = off, 4294967295 = always)
// does the kernel launch_walk would pick for this launch read WalkArgs::tab?  (the throughput kernel and the four-wave latency
// kernel do; the one-wave latency kernel does not: engine.hip then skips the table GEMM)
static void walk_env_overrides(u32 &lat_max_B, u32 &lat4_max_B) {
    const long long lat_env = tune_or(TUNE_WALK_LAT, -1), lat4_env = tune_or(TUNE_WALK_LAT4, -1);
    if (lat_env >= 0) lat_max_B = lat_env > 0xFFFFFFFFll ? 0xFFFFFFFFu : (u32)lat_env;
    if (lat4_env >= 0) lat4_max_B = lat4_env > 0xFFFFFFFFll ? 0xFFFFFFFFu : (u32)lat4_env;
}
// table_available: the launch can have a level table (u8 codes).  With the table the throughput kernel beats the one-wave latency
// kernel at every launch size (1M x 768, ms per launch at ef 64 / 256: 256 queries 0.77 / 2.20 against 0.99 / 2.37; 1 024 queries
// 0.84 / 2.33 against 1.13 / 2.57; 2 048 queries 0.95 / 2.58 against 1.50 / 2.91) and the four-wave kernel up to ef 64 (0.77
// against 0.82 with the table, 0.88 without); above ef 64 the four-wave kernel with the table stays ahead on one client batch
// (1.73 against 2.20 at ef 256).  profiles/r04_single_batch_probe.jsonl, r04_mid_size_probe.jsonl.  Round 6 (r06_single_batch_probe.jsonl): one
// client batch with the automatic table — ef 64 0.642 / 0.643 ms (throughput / four waves), ef 128 1.017 / 1.148, ef 256 1.747 / 1.658.
int walk_kernel_kind(int eng, const IndexDev &ix, const WalkArgs &wa, u32 lat_max_B, u32 lat4_max_B, bool table_available) { // 0 throughput, 1 one-wave latency, 4 four-wave latency
    walk_env_overrides(lat_max_B, lat4_max_B);
    const bool tk_small = tune_or(TUNE_WALK_SMALL_TABLE_TK, 1) != 0;
    if (table_available && tk_small && eng == ENG_U8) {
        if (wa.phase == 0u && wa.ef > 128u && walk_lat4_applicable(eng, ix, wa, lat4_max_B)) return 4; // (round 6: the 128-key pool + ranked merge moved the crossover from ef 64 to 128)
        return 0;
    }
    if (wa.phase == 0u) {
        if (walk_lat4_applicable(eng, ix, wa, lat4_max_B)) return 4;
        if (walk_lat_applicable(eng, ix, wa, lat_max_B)) return 1;
    }
    return 0;
}

hipError_t launch_walk(int eng, const IndexDev &ix, const WalkArgs &wa, u32 lat_max_B, u32 lat4_max_B, hipStream_t st) {
    if (wa.B == 0) return hipSuccess;
    if (walk_general_needed(ix, wa.ef)) return launch_walk_general(eng, ix, wa, st); // ef > 1024 or more than 64 scanned slots per node
    // the split (locality-ordered) walk and the unseeded filter of delete_embedding's walks (WalkArgs::no_self_seed) exist in the throughput kernel only
    const int kind = wa.no_self_seed ? 0 : walk_kernel_kind(eng, ix, wa, lat_max_B, lat4_max_B, wa.tab != nullptr);
    if (kind == 4) return launch_walk_lat4(eng, ix, wa, st);
    if (kind == 1) return launch_walk_lat(eng, ix, wa, st);
    const u32 ch = (eng == ENG_F32 || eng == ENG_F16) ? 1 : (ix.nchunks + ix.G - 1) / ix.G;
    switch (eng) {
    case ENG_U8:
        if (ch == 1) return ix.G == 64 ? launch_walk_r<ENG_U8, 1, true>(ix, wa, st) : launch_walk_r<ENG_U8, 1, false>(ix, wa, st);
        if (ch == 2) return launch_walk_r<ENG_U8, 2, true>(ix, wa, st);
        if (ch == 3) return launch_walk_r<ENG_U8, 3, true>(ix, wa, st); // 2049..3072 dimensions (round 6: three chunk passes)
        if (ch == 4) return launch_walk_r<ENG_U8, 4, true>(ix, wa, st); // 3073..4096 dimensions (four; wider rows: walk_general_kernel)
        return hipErrorInvalidValue;
    case ENG_Q2:
        if (ch == 1) return ix.G == 64 ? launch_walk_r<ENG_Q2, 1, true>(ix, wa, st) : launch_walk_r<ENG_Q2, 1, false>(ix, wa, st);
        return hipErrorInvalidValue;
    case ENG_Q1:
        if (ch == 1) return ix.G == 64 ? launch_walk_r<ENG_Q1, 1, true>(ix, wa, st) : launch_walk_r<ENG_Q1, 1, false>(ix, wa, st);
        return hipErrorInvalidValue;
    case ENG_Q3:
        if (ch == 1) return ix.G == 64 ? launch_walk_r<ENG_Q3, 1, true>(ix, wa, st) : launch_walk_r<ENG_Q3, 1, false>(ix, wa, st);
        return hipErrorInvalidValue;
    case ENG_F32: return launch_walk_r<ENG_F32, 1, false>(ix, wa, st);
    case ENG_F16: return launch_walk_r<ENG_F16, 1, false>(ix, wa, st);
    default: return hipErrorInvalidValue;
    }
}

hipError_t launch_finalize(const IndexDev &ix, const float *queries, u64 q_stride, const float *q_raw_mags, const u32 *walk_ids,
                           const float *walk_sims, const u32 *walk_counts, const int32_t *walk_status, u32 B, u32 top_k,
                           u32 *out_ids, float *out_scores, u32 *out_counts, int32_t *out_status, u64 *out_rerank_rows,
                           hipStream_t st, const u32 *q_order, u32 *slow_list /* [B + 1] scratch: count, then the list */) {
    if (B == 0) return hipSuccess;
    FinalizeArgs fa{queries, q_stride, q_raw_mags, walk_ids, walk_sims, walk_counts, walk_status, B, top_k,
                    out_ids, out_scores, out_counts, out_status, out_rerank_rows, q_order, nullptr, nullptr};
    const u32 per_level = std::min<u32>(KEEP_SEARCH, 5u * top_k + 1u);
    const u32 total = (ix.num_layers + 1) * per_level;
    dim3 grid(B), block(64);
    // small launches: FIN_WAVES waves per query (finalize_one); tuning knob finalize_wide_max_b = the largest such launch (0 = never)
    constexpr int FIN_WAVES = 8;
    const bool wide = B <= (u32)std::max<long long>(0, tune_or(TUNE_FINALIZE_WIDE_MAX_B, 1024));
    // the screened kernel first (5k + 1 <= 64 entries per level list), then the general kernel over the list of queries it left;
    // tuning knob finalize_fast = 0 keeps the general kernel alone (experiments)
    const bool fast_on = tune_or(TUNE_FINALIZE_FAST, 1) != 0;
    bool listed = false;
    if (fast_on && slow_list && ix.mdim == 0u && 5u * top_k + 1u <= 64u && KEEP_SEARCH >= 5u * top_k + 1u) { // base collections: only the root is ever dropped
        fa.slow_list = slow_list + 1;
        fa.slow_count = slow_list;
        hipError_t e = hipMemsetAsync(slow_list, 0, 4, st);
        if (e != hipSuccess) return e;
        const size_t smem_f = (((size_t)ix.dim * 4 + 15) & ~(size_t)15) + (size_t)FAST_CAP * 8 + 64 * 4 + 16;
        if (wide) hipLaunchKernelGGL(finalize_fast_kernel<FIN_WAVES>, grid, dim3(64 * FIN_WAVES), smem_f, st, ix, fa);
        else hipLaunchKernelGGL(finalize_fast_kernel<1>, grid, block, smem_f, st, ix, fa);
        e = hipGetLastError();
        if (e != hipSuccess) return e;
        listed = true;
        grid = dim3(std::min<u32>(B, 2048u));
    }
#define FIN(FR_)                                                                                                   \
    do {                                                                                                           \
        const size_t smem = (((size_t)ix.dim * 4 + 15) & ~(size_t)15) + 64 * (FR_) * 4;                            \
        if (listed) hipLaunchKernelGGL(finalize_list_kernel<FR_>, grid, block, smem, st, ix, fa);                 \
        else if (wide) hipLaunchKernelGGL((finalize_wide_kernel<FR_, FIN_WAVES>), grid, dim3(64 * FIN_WAVES), smem + 64 * (FR_) * 8 + 16, st, ix, fa); \
        else hipLaunchKernelGGL(finalize_kernel<FR_>, grid, block, smem, st, ix, fa);                             \
    } while (0)
    if (total <= 64 * 4) FIN(4);
    else if (total <= 64 * 8) FIN(8);
    else if (total <= 64 * 16) FIN(16);
    else if (total <= 64 * 32) FIN(32);
    else return hipErrorInvalidValue;
#undef FIN
    return hipGetLastError();
}

} // namespace cosdev
