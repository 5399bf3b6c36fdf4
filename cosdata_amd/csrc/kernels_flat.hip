// kernels_flat.hip — exhaustive (flat) cosine scan of the resident raw vectors: the one genuine dense
// contraction of the path, S = Q[B x d] . X^T[d x N], on the f32 MFMA
// (v_mfma_f32_32x32x2_f32: exact f32 products, k-ordered fmaf chain, 157 TF peak on gfx950).
// Used for recall ground truth and as the exhaustive per-shard mode.  Pipeline per N-chunk:
//   flat_gemm_f32      128x128 tile per workgroup (4 waves, 2x2 MFMA tiles of 32x32 per wave), LDS-staged K panels,
//                      epilogue divides by |q|*|x| and writes cosine scores [B][chunk]
//   flat_select_*      waves per (query, segment) keep a top-64 (sorted register pool, threshold filter); merged per query
//   flat_rescore       exact reference-order re-score (dot_product_f32_simd order) of the 64 survivors -> top-k
// MFMA accumulation order differs from the reference's 8-lane tree in the last ulp, so the GEMM only
// GENERATES candidates (64 >= 2k with margin); the returned ids/scores come from the reference-order kernel
// and are bit-identical to the oracle's brute force.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <type_traits>
#include <vector>

#include "dot_engines.h"
#include "engine_internal.h"
#include "flat_scan.h"

using namespace cosdev;

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BM = 128, BN = 128, BK = 32, LDT = BK + 4; // 36-float rows: 16 B aligned, conflict-free ds_read/ds_write_b128
constexpr int SEL = 64;                                  // survivors per query
constexpr size_t GEMM_SMEM = 2ull * (BM + BN) * LDT * sizeof(float); // double-buffered Q and X panels: 73 728 B -> 2 workgroups per CU

// S = Q . X^T on v_mfma_f32_32x32x2_f32.  128x128 tile per 4-wave workgroup (each wave 64x64 = 2x2 MFMA blocks), K panels
// of 32 double-buffered in LDS: panel p+1 travels HBM -> registers while panel p is multiplied, then registers -> LDS, one
// barrier per panel.  A lane's MFMA operand for k-step s of an 8-wide k block is element s of ONE ds_read_b128
// (k = 4*(lane>>5) + s): the k permutation is the same for both operands, so the sum is unchanged.
template <bool VEC, bool FUSED>
__global__ __launch_bounds__(256) void flat_gemm_f32(const float *__restrict__ Q, u64 q_stride, const float *__restrict__ qmags, u32 B,
                                                     const float *__restrict__ X, u64 x_stride, const float *__restrict__ xmags, u32 n0,
                                                     u32 n_chunk, u32 dim, float *__restrict__ scores /*[B][n_chunk_padded]*/, u64 s_stride,
                                                     const FusedOut fo /*FUSED: thresholds in, survivors out (see flat_codes_gemm_i8)*/) {
    extern __shared__ __attribute__((aligned(16))) float gsm[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    // XCD-aware tile order: consecutive workgroup ids land on different XCDs (id % 8); remap so that each XCD walks a
    // contiguous strip of the tile sequence, and order the sequence query-tile-fastest: the tiles_m workgroups that share
    // one 128-row X tile run back to back on one XCD, so X streams from HBM once and is re-used from that XCD's L2
    // (the query panel, a few MB, stays in L2 / Infinity Cache).
    const u32 tiles_n = gridDim.x, tiles_m = gridDim.y;
    u32 wg = blockIdx.y * tiles_n + blockIdx.x;
    const u32 total = tiles_n * tiles_m;
    {
        const u32 q = total / 8, r = total % 8, xcd = wg % 8, idx = wg / 8;
        wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx; // bijective for any total
    }
    const u32 tm = wg % tiles_m, tn = wg / tiles_m;
    const u32 row0 = tm * BM, col0 = tn * BN;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[i][j][r] = 0.0f;

    // staging: 8 consecutive lanes fetch one row's 128 B of the panel (float4 each), 32 rows per pass, 4 passes per operand
    const int sr = tid >> 3, sc = (tid & 7) * 4;
    float4 ra[4], rb[4];
    auto fetch = [&](u32 k0) {
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const u32 qr = row0 + i * 32 + sr, xr = col0 + i * 32 + sr, k = k0 + sc;
            const float *pa = Q + (u64)qr * q_stride + k;
            const float *pb = X + (u64)(n0 + xr) * x_stride + k;
            if constexpr (VEC) { // dim % 4 == 0 and 16 B aligned rows
                ra[i] = (qr < B && k < dim) ? *(const float4 *)pa : make_float4(0.f, 0.f, 0.f, 0.f);
                rb[i] = (xr < n_chunk && k < dim) ? *(const float4 *)pb : make_float4(0.f, 0.f, 0.f, 0.f);
            } else {
                float a[4], b[4];
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    a[e] = (qr < B && k + e < dim) ? pa[e] : 0.f;
                    b[e] = (xr < n_chunk && k + e < dim) ? pb[e] : 0.f;
                }
                ra[i] = make_float4(a[0], a[1], a[2], a[3]);
                rb[i] = make_float4(b[0], b[1], b[2], b[3]);
            }
        }
    };
    auto stash = [&](int buf) {
        float *As = gsm + (size_t)buf * (BM + BN) * LDT, *Bs = As + BM * LDT;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            *(float4 *)(As + (i * 32 + sr) * LDT + sc) = ra[i];
            *(float4 *)(Bs + (i * 32 + sr) * LDT + sc) = rb[i];
        }
    };

    const u32 npanels = (dim + BK - 1) / BK;
    fetch(0);
    stash(0);
    __syncthreads();
    const int orow = lane & 31, okq = (lane >> 5) * 4;
    for (u32 p = 0; p < npanels; p++) {
        if (p + 1 < npanels) fetch((p + 1) * BK);
        const float *As = gsm + (size_t)(p & 1) * (BM + BN) * LDT, *Bs = As + BM * LDT;
#pragma unroll
        for (int kb = 0; kb < BK; kb += 8) {
            const float4 a0 = *(const float4 *)(As + (wr * 64 + orow) * LDT + kb + okq);
            const float4 a1 = *(const float4 *)(As + (wr * 64 + 32 + orow) * LDT + kb + okq);
            const float4 b0 = *(const float4 *)(Bs + (wc * 64 + orow) * LDT + kb + okq);
            const float4 b1 = *(const float4 *)(Bs + (wc * 64 + 32 + orow) * LDT + kb + okq);
            const float av0[4] = {a0.x, a0.y, a0.z, a0.w}, av1[4] = {a1.x, a1.y, a1.z, a1.w};
            const float bv0[4] = {b0.x, b0.y, b0.z, b0.w}, bv1[4] = {b1.x, b1.y, b1.z, b1.w};
#pragma unroll
            for (int s = 0; s < 4; s++) {
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av0[s], bv0[s], acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av0[s], bv1[s], acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av1[s], bv0[s], acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av1[s], bv1[s], acc[1][1], 0, 0, 0);
            }
        }
        if (p + 1 < npanels) stash((p + 1) & 1); // the other buffer: last read before the previous barrier
        __syncthreads();
    }
    // epilogue: C/D layout of 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
    if constexpr (FUSED) {
        // The score matrix never reaches HBM: only what beats the query's 64th best so far is kept.  Per-row filter state lives in LDS
        // (the operand panels are dead: the k loop ended with a barrier) and a reciprocal-based estimate screens the 64 elements a
        // lane holds — the exact quotient, the key and the compare are formed only for estimates within 4e-6 (relative) of the
        // row's threshold or above.  The first version of this epilogue loaded |q| and the threshold from global memory per
        // ELEMENT and waited for each pair: 64 dependent L2 round trips per lane per tile, as long as the tile's MFMAs.
        u64 *thr_key = (u64 *)gsm;                 // [BM]
        float *thr_lo = (float *)(thr_key + BM);   // [BM] estimate a candidate must reach to be worth the exact quotient
        float *rq = thr_lo + BM;                   // [BM] ~1 / |q|
        float *qm = rq + BM;                       // [BM] |q|
        for (int idx = tid; idx < BM; idx += 256) {
            const u32 row = row0 + idx;
            const u64 k = row < B ? fo.thr[row] : ~0ull;
            const float tsc = simkey_inv((u32)(k >> 32));
            const float qv = row < B ? qmags[row] : 1.0f;
            thr_key[idx] = k;
            thr_lo[idx] = k == 0ull ? -__builtin_inff() : tsc - 4e-6f * __builtin_fabsf(tsc); // 0 = pool not full: everything passes
            qm[idx] = qv;
            rq[idx] = __builtin_amdgcn_rcpf(qv);
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 2; i++)
#pragma unroll
            for (int j = 0; j < 2; j++) {
                const u32 col = col0 + wc * 64 + j * 32 + (lane & 31);
                const bool cv = col < n_chunk;
                const float xm = cv ? xmags[n0 + col] : 1.0f;
                const float rx = __builtin_amdgcn_rcpf(xm);
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    const int rl = wr * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                    const float est = acc[i][j][r] * rx * rq[rl];
                    // NaN-safe: an estimate that is not provably below the bar (incl. 0 * inf of a zero-norm operand) takes the exact path
                    if (row0 + (u32)rl < B && cv && !(est < thr_lo[rl])) {
                        const float sc = x86_div(acc[i][j][r], qm[rl] * xm);
                        const u64 key = pack_key(simkey(sc), n0 + col);
                        if (key > thr_key[rl]) {
                            const u32 row = row0 + (u32)rl;
                            const u32 pos = atomicAdd(&fo.app_cnt[row], 1u);
                            if (pos < fo.cap) fo.app[(u64)row * fo.cap + pos] = key;
                        }
                    }
                }
            }
        return;
    }
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++) {
            const u32 col = col0 + wc * 64 + j * 32 + (lane & 31);
            const float xm = col < n_chunk ? xmags[n0 + col] : 1.0f;
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const u32 row = row0 + wr * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (row < B && col < n_chunk) scores[(u64)row * s_stride + col] = x86_div(acc[i][j][r], qmags[row] * xm);
            }
        }
}

// Selection of the SEL best of a chunk in two steps so that a small query batch still fills the chip:
//   flat_select_segments  grid (B, S): one wave per (query, segment of the chunk) keeps the segment's top-SEL
//                         (sorted register pool, ballot threshold filter) -> part[B][S][SEL]
//   flat_select_merge     one wave per query folds the S sorted partial lists into the running pool[B][SEL]
// keys = (simkey(score), global id): ties go to the larger id, like everywhere else.
__global__ __launch_bounds__(64) void flat_select_segments(const float *__restrict__ scores, u64 s_stride, u32 B, u32 n0, u32 n_chunk, u32 seg_len,
                                                           u64 *__restrict__ part /*[B][S][64]*/) {
    const int lane = threadIdx.x;
    const u32 q = blockIdx.x, seg = blockIdx.y, S = gridDim.y;
    if (q >= B) return;
    Pool<1> pool;
    pool.clear();
    u64 thr = 0ull; // SEL-th best so far (0 while not full)
    const float *sr = scores + (u64)q * s_stride;
    const u32 c0 = seg * seg_len, c1 = (c0 + seg_len < n_chunk) ? c0 + seg_len : n_chunk;
    for (u32 c = c0; c < c1; c += 64) {
        const u32 col = c + lane;
        u64 key = 0ull;
        if (col < c1) key = pack_key(simkey(sr[col]), n0 + col);
        u64 m = __ballot(key > thr);
        while (m) {
            const int l = __ffsll((long long)m) - 1;
            m &= m - 1;
            const u64 kk = readlane_u64(key, l);
            if (kk > thr) {
                pool.insert_at(kk, pool.rank_of(kk), lane);
                thr = readlane_u64(pool.e[0], SEL - 1);
            }
        }
    }
    part[((u64)q * S + seg) * SEL + lane] = pool.e[0];
}

__global__ __launch_bounds__(64) void flat_select_merge(const u64 *__restrict__ part, u32 B, u32 S, u64 *__restrict__ pool_mem /*[B][64]*/,
                                                        u64 *__restrict__ thr_out /*optional [B]: the SEL-th best key so far (0 while the pool is not full)*/) {
    const int lane = threadIdx.x;
    const u32 q = blockIdx.x;
    if (q >= B) return;
    Pool<1> pool;
    pool.e[0] = pool_mem[(u64)q * SEL + lane];
    u64 thr = readlane_u64(pool.e[0], SEL - 1);
    for (u32 sgm = 0; sgm < S; sgm++) {
        const u64 key = part[((u64)q * S + sgm) * SEL + lane];
        u64 m = __ballot(key > thr);
        while (m) {
            const int l = __ffsll((long long)m) - 1;
            m &= m - 1;
            const u64 kk = readlane_u64(key, l);
            if (kk > thr) {
                pool.insert_at(kk, pool.rank_of(kk), lane);
                thr = readlane_u64(pool.e[0], SEL - 1);
            }
        }
    }
    pool_mem[(u64)q * SEL + lane] = pool.e[0];
    if (thr_out && lane == 0) thr_out[q] = thr;
}

// Fused scans (flat_codes_gemm_i8<ENG, true>): the GEMM epilogue appended only the candidates that beat the query's
// threshold to app[B][cap]; one wave per query folds them into the pool, publishes the new threshold and clears the counter.
// A counter above `cap` means entries were dropped: the flag makes the host repeat the search on the unfused path.
__global__ __launch_bounds__(64) void flat_select_append(const u64 *__restrict__ app, u32 *__restrict__ app_cnt, u32 cap, u32 B, u64 *__restrict__ pool_mem,
                                                         u64 *__restrict__ thr_out, u32 *__restrict__ overflow) {
    const int lane = threadIdx.x;
    const u32 q = blockIdx.x;
    if (q >= B) return;
    Pool<1> pool;
    pool.e[0] = pool_mem[(u64)q * SEL + lane];
    u64 thr = readlane_u64(pool.e[0], SEL - 1);
    u32 cnt = app_cnt[q];
    if (cnt > cap) { if (lane == 0) atomicOr(overflow, 1u); cnt = cap; }
    for (u32 c = 0; c < cnt; c += 64) {
        const u64 key = c + lane < cnt ? app[(u64)q * cap + c + lane] : 0ull;
        u64 m = __ballot(key > thr);
        while (m) {
            const int l = __ffsll((long long)m) - 1;
            m &= m - 1;
            const u64 kk = readlane_u64(key, l);
            if (kk > thr) {
                pool.insert_at(kk, pool.rank_of(kk), lane);
                thr = readlane_u64(pool.e[0], SEL - 1);
            }
        }
    }
    pool_mem[(u64)q * SEL + lane] = pool.e[0];
    if (lane == 0) { thr_out[q] = thr; app_cnt[q] = 0; }
}

// segments per query: enough waves to fill 256 CUs x 8, at least 4096 candidates per segment
static u32 select_segments(u32 B, u32 n_chunk) {
    u32 S = (4096 + B - 1) / B;
    const u32 max_s = (n_chunk + 4095) / 4096;
    if (S > max_s) S = max_s;
    if (S > 64) S = 64;
    return S ? S : 1;
}
static hipError_t launch_select(const float *d_scores, u64 s_stride, u32 B, u32 n0, u32 nc, u64 *d_part, u32 S, u64 *d_pool, hipStream_t st,
                                u64 *d_thr = nullptr) {
    const u32 seg_len = ((nc + S - 1) / S + 63) / 64 * 64;
    hipLaunchKernelGGL(flat_select_segments, dim3(B, S), dim3(64), 0, st, d_scores, s_stride, B, n0, nc, seg_len, d_part);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(flat_select_merge, dim3(B), dim3(64), 0, st, (const u64 *)d_part, B, S, d_pool, d_thr);
    return hipGetLastError();
}

// any zero among n floats -> *flag = 1 (cosine zero-norm screening without copying the norms to the host)
__global__ void any_zero_kernel(const float *__restrict__ v, u32 n, u32 *__restrict__ flag) {
    bool z = false;
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (u64)gridDim.x * blockDim.x) z |= (v[i] == 0.0f);
    if (__any(z) && (threadIdx.x & 63) == 0) atomicOr(flag, 1u);
}

// one launch instead of four memsets in front of every scan: the survivor pools, the thresholds, the append counters, the overflow flag
__global__ void flat_reset_kernel(u64 *__restrict__ pool, u64 n_pool, u64 *__restrict__ thr, u32 *__restrict__ appcnt, u32 B, u32 *__restrict__ overflow) {
    const u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x, step = (u64)gridDim.x * blockDim.x;
    for (u64 i = t; i < n_pool; i += step) pool[i] = 0ull;
    for (u64 i = t; i < B; i += step) { thr[i] = 0ull; appcnt[i] = 0u; }
    if (t == 0) *overflow = 0u;
}
// the call's results and its two flags side by side, for one copy back
__global__ void flat_pack_out_kernel(const u32 *__restrict__ oi, const float *__restrict__ os, const u32 *__restrict__ oc, const u32 *__restrict__ flags,
                                     u32 B, u32 k, u32 *__restrict__ out) {
    const u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x, nk = (u64)B * k;
    if (t < nk) { out[t] = oi[t]; out[nk + t] = __float_as_uint(os[t]); }
    if (t < B) out[2 * nk + t] = oc[t];
    if (t < 2) out[2 * nk + B + t] = flags[t];
}

// exact re-score of the survivors in the reference order, sort, top-k
__global__ __launch_bounds__(64) void flat_rescore(const float *__restrict__ Q, u64 q_stride, const float *__restrict__ qmags, u32 B,
                                                   const float *__restrict__ X, u64 x_stride, const float *__restrict__ xmags, u32 dim,
                                                   const u64 *__restrict__ pool_mem, u32 k, u32 id_base, u32 *__restrict__ out_ids,
                                                   float *__restrict__ out_scores) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float *qf = (float *)smem_raw;
    const int lane = threadIdx.x;
    const u32 q = blockIdx.x;
    if (q >= B) return;
    for (u32 i = lane; i < dim; i += 64) qf[i] = Q[(u64)q * q_stride + i];
    const u64 mine = pool_mem[(u64)q * SEL + lane];
    const float mq = qmags[q];
    u64 res[1] = {0ull};
    for (int base = 0; base < SEL; base += 32) { // 32 survivors per pass, one lane pair each
        const int src = base + (lane >> 1);
        const u32 sid = (u32)__shfl((int)(u32)mine, src, 64);
        const bool valid = __shfl((int)(u32)(mine >> 32), src, 64) != 0;
        const u32 row = valid ? sid : 0u;
        const float dp = f32_pair_dot(X + (u64)row * x_stride, qf, dim, lane & 1);
        const float cs = x86_div(dp, mq * xmags[row]); // dp / (mag_query * mag_raw), vector_store.rs:427
        const u64 key = valid ? pack_key(simkey(cs), sid) : 0ull;
        // survivor base + j was computed by lanes 2j and 2j+1 -> hand it to lane base + j
        const int from = (2 * (lane - base)) & 63;
        const u32 klo = (u32)__shfl((int)(u32)key, from, 64), khi = (u32)__shfl((int)(u32)(key >> 32), from, 64);
        if (lane >= base && lane < base + 32) res[0] = ((u64)khi << 32) | klo;
    }
    bitonic_sort_desc<1>(res, lane);
    if ((u32)lane < k) {
        const bool ok = res[0] != 0ull;
        out_ids[(u64)q * k + lane] = ok ? (u32)res[0] + id_base : 0xFFFFFFFFu;
        out_scores[(u64)q * k + lane] = ok ? simkey_inv((u32)(res[0] >> 32)) : 0.0f;
    }
}

// ------------------------------------------------------------------------------------------------
// Exhaustive scan over the QUANTIZED codes as an exact-integer i8 MFMA GEMM (config c3 / per-shard flat mode).
//   u8 codes   : a' = a - 128 (flip the top bit) makes them signed; sum(a*b) = S' + 128*sum(a) + 128*sum(b) - 16384*K
//                with S' = sum(a'*b') from v_mfma_i32_32x32x32_i8 — every term an exact integer
//   quaternary : the two bit planes are expanded to the digit (plane0 bit + 2 * plane1 bit), i.e. exactly the
//                value dot_product_quaternary multiplies (dot_product.rs:35-57), 4 digits per u32 via
//                (nibble * 0x00204081) & 0x01010101
// A/B fragments are read K-contiguous (16 B per lane, k-block = lane >> 5) for both operands, so the result is
// the plain sum over k whatever the hardware's internal k order is; C/D layout as in flat_gemm_f32.
// Workgroup = 8 waves, tile 256 queries x 128 candidates, K step 64; rows padded to 80 B in LDS (conflict-free
// ds_read_b128).  Integer dots are converted with RNE and divided by |q|*|v| like cosine.rs:223-235.
// ------------------------------------------------------------------------------------------------
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x16 __attribute__((ext_vector_type(16)));
constexpr int CM = 256, CN = 128, CK = 64, CLD = 80;

__device__ __forceinline__ u32 spread4(u32 nib) { return (nib * 0x00204081u) & 0x01010101u; }

// 16 digits (dims k0..k0+15 of a 64-dim chunk) -> 16 i8 values, in two halves so the global load of a later k panel can be in
// flight while the current one is multiplied: load16_raw (16 B as stored) and unpack16 (what the MFMA reads from LDS)
template <int ENG>
__device__ __forceinline__ uint4 load16_raw(const uint8_t *__restrict__ row, u32 k0 /*multiple of 16*/, bool valid) {
    if (!valid) return make_uint4(0, 0, 0, 0);
    if constexpr (ENG == ENG_U8) return *(const uint4 *)(row + k0);
    else return *(const uint4 *)(row + (u64)(k0 >> 6) * 16); // 16 B per 64 dims: [plane0 8 B | plane1 8 B]
}
template <int ENG>
__device__ __forceinline__ uint4 unpack16(uint4 raw, u32 k0, bool valid) {
    uint4 o = make_uint4(0, 0, 0, 0);
    if constexpr (ENG == ENG_U8) {
        o = raw;
        o.x ^= 0x80808080u; o.y ^= 0x80808080u; o.z ^= 0x80808080u; o.w ^= 0x80808080u; // padding (0) becomes -128 too: see epilogue
    } else {
        if (valid) {
            const u32 sh = k0 & 63;
            const u64 p0 = (u64)raw.x | ((u64)raw.y << 32), p1 = (u64)raw.z | ((u64)raw.w << 32);
            const u32 b0 = (u32)(p0 >> sh) & 0xFFFFu, b1 = (u32)(p1 >> sh) & 0xFFFFu;
            o.x = spread4(b0 & 15u) + 2u * spread4(b1 & 15u);
            o.y = spread4((b0 >> 4) & 15u) + 2u * spread4((b1 >> 4) & 15u);
            o.z = spread4((b0 >> 8) & 15u) + 2u * spread4((b1 >> 8) & 15u);
            o.w = spread4((b0 >> 12) & 15u) + 2u * spread4((b1 >> 12) & 15u);
        }
    }
    return o;
}
template <int ENG>
__device__ __forceinline__ uint4 stage16(const uint8_t *__restrict__ row, u32 k0 /*multiple of 16*/, bool valid) {
    return unpack16<ENG>(load16_raw<ENG>(row, k0, valid), k0, valid);
}

// queries' quaternary planes -> the i8 digits the GEMM multiplies, once per batch ([B][kdims] bytes): the scan kernel then
// stages the query operand with plain 16 B copies instead of re-expanding it for every one of the N / 128 candidate tiles
__global__ void expand_q2_digits_kernel(const uint8_t *__restrict__ qcodes, u64 row_stride, u32 B, u32 kdims, uint8_t *__restrict__ digits) {
    const u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    const u32 pieces = kdims / 16;
    if (t >= (u64)B * pieces) return;
    const u32 q = (u32)(t / pieces), k0 = (u32)(t % pieces) * 16;
    const u32 code_k = (u32)(row_stride / 16) * 64;
    *(uint4 *)(digits + (u64)q * kdims + k0) = stage16<ENG_Q2>(qcodes + (u64)q * row_stride, k0, k0 < code_k);
}

template <int ENG, bool FUSED, int PF>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(4, 4))) void flat_codes_gemm_i8(const uint8_t *__restrict__ qcodes, const float *__restrict__ qmags,
                                                          const u32 *__restrict__ qsums, u32 B, const uint8_t *__restrict__ codes,
                                                          const float *__restrict__ mags, const u32 *__restrict__ csums, u64 row_stride,
                                                          u32 n0, u32 n_chunk, u32 kdims /*padded to 64*/, u32 metric,
                                                          float *__restrict__ scores, u64 s_stride, const FusedOut fo) {
    __shared__ __attribute__((aligned(16))) unsigned char As[CM * CLD];
    __shared__ __attribute__((aligned(16))) unsigned char Bs[CN * CLD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1; // 4 x 2 waves, 64 x 64 outputs each
    const u32 row0 = blockIdx.y * CM, col0 = blockIdx.x * CN;
    i32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[i][j][r] = 0;
    const u32 code_k = ENG == ENG_U8 ? (u32)row_stride : (u32)(row_stride / 16) * 64; // dims covered by stored bytes
    // Register ring of PF k panels: the 16 B pieces of panels k+1 .. k+PF are in flight (global -> VGPR) while panel k is
    // unpacked into LDS and multiplied.  With only 2 workgroups per CU the un-prefetched loop exposed the full HBM latency on
    // every panel (0.21 of the MFMA peak); the ring hides it behind PF panels of MFMA work per workgroup.
    // A: 256 rows x 64 B = 1024 x 16 B pieces, 2 per thread;  B: 128 rows x 64 B = 512 pieces, 1 per thread
    const int ar[2] = {(tid * 2) >> 2, (tid * 2 + 1) >> 2}, akq[2] = {((tid * 2) & 3) * 16, ((tid * 2 + 1) & 3) * 16};
    const int br = tid >> 2, bkq = (tid & 3) * 16;
    const bool b_in = col0 + br < n_chunk;
    const uint8_t *b_row = codes + (u64)(n0 + (b_in ? col0 + br : 0)) * row_stride;
    uint4 ra[PF][2], rb[PF];
    auto fetch = [&](int s, u32 k0) {
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const u32 qr = row0 + ar[h];
            if constexpr (FUSED && ENG == ENG_Q2)
                ra[s][h] = qr < B ? *(const uint4 *)(fo.qdigits + (u64)qr * kdims + k0 + akq[h]) : make_uint4(0, 0, 0, 0);
            else
                ra[s][h] = load16_raw<ENG>(qcodes + (u64)(qr < B ? qr : 0) * row_stride, k0 + akq[h], qr < B && k0 + akq[h] < code_k);
        }
        rb[s] = load16_raw<ENG>(b_row, k0 + bkq, b_in && k0 + bkq < code_k);
    };
#pragma unroll
    for (int s = 0; s < PF; s++)
        if ((u32)s * CK < kdims) fetch(s, (u32)s * CK);
    for (u32 kb = 0; kb < kdims; kb += PF * CK) {
#pragma unroll
        for (int s = 0; s < PF; s++) {
            const u32 k0 = kb + (u32)s * CK;
            if (k0 >= kdims) break; // uniform
#pragma unroll
            for (int h = 0; h < 2; h++) {
                if constexpr (FUSED && ENG == ENG_Q2)
                    *(uint4 *)(As + ar[h] * CLD + akq[h]) = ra[s][h];
                else
                    *(uint4 *)(As + ar[h] * CLD + akq[h]) = unpack16<ENG>(ra[s][h], k0 + akq[h], row0 + ar[h] < B && k0 + akq[h] < code_k);
            }
            *(uint4 *)(Bs + br * CLD + bkq) = unpack16<ENG>(rb[s], k0 + bkq, b_in && k0 + bkq < code_k);
            if (k0 + PF * CK < kdims) fetch(s, k0 + PF * CK);
            __syncthreads();
#pragma unroll
            for (int ks = 0; ks < CK; ks += 32) {
                const int ko = ks + (lane >> 5) * 16;
                const i32x4 a0 = *(const i32x4 *)(As + (wr * 64 + (lane & 31)) * CLD + ko), a1 = *(const i32x4 *)(As + (wr * 64 + 32 + (lane & 31)) * CLD + ko);
                const i32x4 b0 = *(const i32x4 *)(Bs + (wc * 64 + (lane & 31)) * CLD + ko), b1 = *(const i32x4 *)(Bs + (wc * 64 + 32 + (lane & 31)) * CLD + ko);
                acc[0][0] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a0, b0, acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a0, b1, acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a1, b0, acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a1, b1, acc[1][1], 0, 0, 0);
            }
            __syncthreads();
        }
    }
    if constexpr (FUSED) {
        // per-row filter state in LDS (the operand panels are dead: the k loop ended with a barrier)
        u64 *thr_key = (u64 *)As;                         // [CM]
        float *thr_lo = (float *)(As + CM * 8);           // [CM] score a candidate must reach to be worth the exact quotient
        float *rq = (float *)(As + CM * 12);              // [CM] ~1/|q|
        for (int idx = tid; idx < CM; idx += 512) {
            const u32 row = row0 + idx;
            const u64 k = row < B ? fo.thr[row] : ~0ull;
            thr_key[idx] = k;
            thr_lo[idx] = k == 0ull ? -1.0f : simkey_inv((u32)(k >> 32)) * (1.0f - 4e-6f); // scores of u8 / quaternary codes are >= 0
            rq[idx] = row < B ? __builtin_amdgcn_rcpf(qmags[row]) : 0.0f;
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 2; i++)
#pragma unroll
            for (int j = 0; j < 2; j++) {
                const u32 col = col0 + wc * 64 + j * 32 + (lane & 31);
                const bool cv = col < n_chunk;
                const float xm = cv ? mags[n0 + col] : 1.0f;
                const float rx = __builtin_amdgcn_rcpf(xm);
                const int cs = (ENG == ENG_U8 && cv) ? (int)csums[n0 + col] : 0;
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    const int rl = wr * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                    const u32 row = row0 + rl;
                    int dot = acc[i][j][r];
                    if constexpr (ENG == ENG_U8) dot += 128 * (int)qsums[row < B ? row : 0] + 128 * cs - 16384 * (int)kdims;
                    const float dotf = (float)(u32)dot;
                    const float est = metric == 0u ? dotf * rx * rq[rl] : dotf;
                    if (row < B && cv && est >= thr_lo[rl]) {
                        const float sc = metric == 0u ? __fdiv_rn(dotf, __fmul_rn(qmags[row], xm)) : dotf;
                        const u64 key = pack_key(simkey(sc), n0 + col);
                        if (key > thr_key[rl]) {
                            const u32 pos = atomicAdd(&fo.app_cnt[row], 1u);
                            if (pos < fo.cap) fo.app[(u64)row * fo.cap + pos] = key;
                        }
                    }
                }
            }
        return;
    }
    // Unfused epilogue (score matrix out: the unfused flat scan and the walk's level table, where it is most of the kernel — K = 768
    // is short and every output needs an exact quotient).  The per-row terms (|q| and the query's share of the u8 recentring) are
    // staged in LDS once per workgroup (the operand panels are dead: the k loop ended with a barrier) instead of two global loads per
    // output; the output pointer of a 32 x 32 block advances by a constant stride; tiles inside the matrix skip the bounds tests.
    float *rqm = (float *)As;             // [CM] |q|
    int *rqs = (int *)(As + CM * 4);      // [CM] 128 * sum(q) - 16384 * kdims (u8), 0 otherwise
    for (int idx = tid; idx < CM; idx += 512) {
        const u32 row = row0 + idx;
        rqm[idx] = row < B ? qmags[row] : 1.0f;
        rqs[idx] = (ENG == ENG_U8 && row < B) ? 128 * (int)qsums[row] - 16384 * (int)kdims : 0;
    }
    __syncthreads();
    const bool inside = row0 + CM <= B && col0 + CN <= n_chunk; // workgroup-uniform
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++) {
            const u32 col = col0 + wc * 64 + j * 32 + (lane & 31);
            const bool cv = col < n_chunk;
            const float xm = cv ? mags[n0 + col] : 1.0f;
            const int cs = (ENG == ENG_U8 && cv) ? 128 * (int)csums[n0 + col] : 0;
            const int rl0 = wr * 64 + i * 32 + 4 * (lane >> 5);
            float *out = scores + (u64)(row0 + rl0) * s_stride + col;
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int dr = (r & 3) + 8 * (r >> 2);
                const int rl = rl0 + dr;
                if (inside || (row0 + (u32)rl < B && cv)) {
                    int dot = acc[i][j][r];
                    if constexpr (ENG == ENG_U8) dot += rqs[rl] + cs;
                    const float dotf = (float)(u32)dot; // exact integer -> f32 RNE, like `as f32` on the u64 dot
                    float sc = dotf;
                    if (metric == 0u) sc = __fdiv_rn(dotf, __fmul_rn(rqm[rl], xm)); // zero norms are screened on the host side
                    out[(u64)dr * s_stride] = sc;
                }
            }
        }
}

__global__ void code_sums_kernel(const uint8_t *__restrict__ codes, u64 row_stride, u32 n, u32 *__restrict__ sums) {
    const int lane = threadIdx.x & 63;
    const u32 row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= n) return;
    u32 s = 0;
    for (u32 i = lane * 4; i < (u32)row_stride; i += 256) {
        const u32 w = *(const u32 *)(codes + (u64)row * row_stride + i);
        s += (w & 255u) + ((w >> 8) & 255u) + ((w >> 16) & 255u) + (w >> 24);
    }
#pragma unroll
    for (int m = 32; m > 0; m >>= 1) s += (u32)__shfl_xor((int)s, m, 64);
    if (lane == 0) sums[row] = s;
}

// exact rerank of the best `ncand` survivors (pool is sorted desc by quantized score, larger id first) -> top k
__global__ __launch_bounds__(64) void flat_rerank_top5k(const float *__restrict__ Q, u64 q_stride, const float *__restrict__ qraw_mags, u32 B,
                                                        const float *__restrict__ X, u64 x_stride, const float *__restrict__ xmags, u32 dim,
                                                        const u64 *__restrict__ pool_mem, u32 ncand_max, u32 k, u32 id_base,
                                                        u32 *__restrict__ out_ids, float *__restrict__ out_scores, u32 *__restrict__ out_counts) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float *qf = (float *)smem_raw;
    const int lane = threadIdx.x;
    const u32 q = blockIdx.x;
    if (q >= B) return;
    for (u32 i = lane; i < dim; i += 64) qf[i] = Q[(u64)q * q_stride + i];
    const u64 mine = pool_mem[(u64)q * SEL + lane];
    const u32 have = (u32)__popcll(__ballot(mine != 0ull));
    const u32 ncand = have < ncand_max ? have : ncand_max;
    const float mq = qraw_mags[q];
    u64 res[1] = {0ull};
    for (int base = 0; base < SEL; base += 32) {
        if ((u32)base >= ncand) break;
        const int src = base + (lane >> 1);
        const u32 sid = (u32)__shfl((int)(u32)mine, src, 64);
        const bool valid = (u32)src < ncand;
        const u32 row = valid ? sid : 0u;
        const float dp = f32_pair_dot(X + (u64)row * x_stride, qf, dim, lane & 1);
        const float cs = x86_div(dp, mq * xmags[row]);
        const u64 key = valid ? pack_key(simkey(cs), sid) : 0ull;
        const int from = (2 * (lane - base)) & 63;
        const u32 klo = (u32)__shfl((int)(u32)key, from, 64), khi = (u32)__shfl((int)(u32)(key >> 32), from, 64);
        if (lane >= base && lane < base + 32) res[0] = ((u64)khi << 32) | klo;
    }
    bitonic_sort_desc<1>(res, lane);
    const u32 nout = ncand < k ? ncand : k;
    if ((u32)lane < nout) {
        out_ids[(u64)q * k + lane] = (u32)res[0] + id_base;
        out_scores[(u64)q * k + lane] = simkey_inv((u32)(res[0] >> 32));
    }
    if (lane == 0) out_counts[q] = nout;
}

} // namespace

extern "C" int32_t cos_bruteforce_topk(cos_index *ix, const float *queries, uint32_t B, uint32_t k, uint32_t *out_ids, float *out_scores) {
    if (!ix || !queries || !out_ids || !out_scores || B == 0) return cos_fail(COS_ERR_INVALID, "bad argument");
    if (!ix->have_vectors) return cos_fail(COS_ERR_NOT_READY, "upload vectors first");
    if (k == 0 || k > 32 || k > ix->n) return cos_fail(COS_ERR_INVALID, "k must be in [1, min(32, n)]");
    int32_t rc = cos_set_device(ix);
    if (rc) return rc;
    const u32 dim = ix->p.dim, n = ix->n;
    // candidates per pass: the [B][chunk] score buffer stays <= 1 GiB
    u32 chunk = (u32)std::min<u64>(1u << 18, ((1ull << 30) / B / 4) / BN * BN);
    chunk = std::min(n, std::max<u32>(chunk, BN));
    // Same schedule as cos_flat_search_batch: the first SEED candidates go through the score matrix + segmented selection and
    // seed every query's threshold; later chunks grow 8x and the GEMM epilogue appends only what beats the threshold.  An
    // append-buffer overflow repeats the call on the unfused path (tuning knob flat_unfused = 1 forces it).
    constexpr u32 SEED = 16384, APP_CAP = 4096;
    const bool allow_fused = tune_or(TUNE_FLAT_UNFUSED, 0) == 0 && n > SEED;
    const u64 s_stride = ((u64)chunk + 63) & ~63ull;
    float *d_q = nullptr, *d_qm = nullptr, *d_scores = nullptr, *d_os = nullptr, *d_dummy = nullptr;
    u64 *d_pool = nullptr, *d_part = nullptr, *d_thr = nullptr, *d_app = nullptr;
    u32 *d_oi = nullptr, *d_appcnt = nullptr, *d_over = nullptr;
    uint8_t *d_codes = nullptr;
    hipStream_t st = ix->own_stream;
    const u32 S = select_segments(B, chunk);
    hipError_t e = hipMalloc(&d_q, (size_t)B * dim * 4);
    if (e == hipSuccess) e = hipMalloc(&d_part, (size_t)B * S * SEL * 8);
    if (e == hipSuccess) e = hipMalloc(&d_qm, (size_t)B * 4);
    if (e == hipSuccess) e = hipMalloc(&d_dummy, (size_t)B * 4);
    if (e == hipSuccess) e = hipMalloc(&d_codes, (size_t)B * (((size_t)dim * 4 + 15) & ~(size_t)15));
    if (e == hipSuccess) e = hipMalloc(&d_scores, (size_t)B * s_stride * 4);
    if (e == hipSuccess) e = hipMalloc(&d_pool, (size_t)B * SEL * 8);
    if (e == hipSuccess) e = hipMalloc(&d_thr, (size_t)B * 8);
    if (e == hipSuccess && allow_fused) e = hipMalloc(&d_app, (size_t)B * APP_CAP * 8);
    if (e == hipSuccess) e = hipMalloc(&d_appcnt, (size_t)B * 4);
    if (e == hipSuccess) e = hipMalloc(&d_over, 4);
    if (e == hipSuccess) e = hipMalloc(&d_oi, (size_t)B * k * 4);
    if (e == hipSuccess) e = hipMalloc(&d_os, (size_t)B * k * 4);
    if (e == hipSuccess) e = hipMemcpyAsync(d_q, queries, (size_t)B * dim * 4, hipMemcpyHostToDevice, st);
    // |q| in the reference's sequential order (vector_store.rs:414): reuse the F32 quantize kernel's raw_mags output
    if (e == hipSuccess) e = launch_quantize_rows(ENG_F32, d_q, dim, B, dim, 0.f, 0.f, d_codes, ((u64)dim * 4 + 15) & ~15ull, d_dummy, d_qm, st);
    for (const void *f : {(const void *)flat_gemm_f32<true, false>, (const void *)flat_gemm_f32<false, false>, (const void *)flat_gemm_f32<true, true>,
                          (const void *)flat_gemm_f32<false, true>})
        if (e == hipSuccess) e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)GEMM_SMEM);
    // float4 staging needs 16 B aligned rows: dim % 4 == 0 (hipMalloc'd bases are 256 B aligned; a borrowed raw pointer is checked)
    const bool vec = dim % 4 == 0 && ((uintptr_t)ix->d_raw & 15) == 0;
    for (int attempt = 0; attempt < 2 && e == hipSuccess; attempt++) {
        const bool fused = allow_fused && attempt == 0;
        if (e == hipSuccess) e = hipMemsetAsync(d_pool, 0, (size_t)B * SEL * 8, st);
        if (e == hipSuccess) e = hipMemsetAsync(d_thr, 0, (size_t)B * 8, st);
        if (e == hipSuccess) e = hipMemsetAsync(d_appcnt, 0, (size_t)B * 4, st);
        if (e == hipSuccess) e = hipMemsetAsync(d_over, 0, 4, st);
        const FusedOut fo{d_thr, d_app, d_appcnt, APP_CAP, nullptr};
        u32 n0 = 0;
        while (n0 < n && e == hipSuccess) {
            const bool use_fused = fused && n0 > 0;
            u32 nc = use_fused ? (u32)std::min<u64>((u64)n0 * 8, 1ull << 22) : (fused ? std::min(SEED, chunk) : chunk);
            nc = std::min(nc, n - n0);
            dim3 grid((nc + BN - 1) / BN, (B + BM - 1) / BM);
#define GEMM_ARGS d_q, (u64)dim, d_qm, B, ix->d_raw, (u64)dim, ix->d_raw_mags, n0, nc, dim, d_scores, s_stride, fo
            if (use_fused) {
                if (vec) hipLaunchKernelGGL((flat_gemm_f32<true, true>), grid, dim3(256), GEMM_SMEM, st, GEMM_ARGS);
                else hipLaunchKernelGGL((flat_gemm_f32<false, true>), grid, dim3(256), GEMM_SMEM, st, GEMM_ARGS);
                e = hipGetLastError();
                if (e == hipSuccess) {
                    hipLaunchKernelGGL(flat_select_append, dim3(B), dim3(64), 0, st, (const u64 *)d_app, d_appcnt, APP_CAP, B, d_pool, d_thr, d_over);
                    e = hipGetLastError();
                }
            } else {
                if (vec) hipLaunchKernelGGL((flat_gemm_f32<true, false>), grid, dim3(256), GEMM_SMEM, st, GEMM_ARGS);
                else hipLaunchKernelGGL((flat_gemm_f32<false, false>), grid, dim3(256), GEMM_SMEM, st, GEMM_ARGS);
                e = hipGetLastError();
                if (e == hipSuccess) e = launch_select(d_scores, s_stride, B, n0, nc, d_part, select_segments(B, nc), d_pool, st, d_thr);
            }
#undef GEMM_ARGS
            n0 += nc;
        }
        u32 hover = 0;
        if (e == hipSuccess) e = hipMemcpyAsync(&hover, d_over, 4, hipMemcpyDeviceToHost, st);
        if (e == hipSuccess) e = hipStreamSynchronize(st);
        if (e != hipSuccess || !fused || hover == 0) break; // done; otherwise the append buffer overflowed: repeat unfused
    }
    if (e == hipSuccess) {
        hipLaunchKernelGGL(flat_rescore, dim3(B), dim3(64), (((size_t)dim * 4 + 15) & ~(size_t)15), st, d_q, (u64)dim, d_qm, B, ix->d_raw, (u64)dim,
                           ix->d_raw_mags, dim, d_pool, k, ix->p.id_base, d_oi, d_os);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(out_ids, d_oi, (size_t)B * k * 4, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipMemcpyAsync(out_scores, d_os, (size_t)B * k * 4, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    void *ptrs[] = {d_q, d_qm, d_dummy, d_codes, d_scores, d_pool, d_part, d_thr, d_app, d_appcnt, d_over, d_oi, d_os};
    for (void *p : ptrs) if (p) (void)hipFree(p);
    HIP_TRY(e);
    return COS_OK;
}

// Per-index workspace of cos_flat_search_batch: every buffer is grow-only and lives until the index is destroyed (or its
// vectors are replaced), so a scan makes no allocation and creates no event on the query path — 17 hipMalloc / hipFree pairs
// and a dozen event create / destroy calls used to cost about as much wall time as the scan kernel itself.
struct FlatWs {
    std::mutex mu; // the scan runs on the index's own stream: one call at a time
    struct Buf { void *p = nullptr; size_t cap = 0; };
    Buf q, zero, qm, qrm, qc, qd, qdp, qs, cs, pool, thr, app, appcnt, oi, os, oc, scores, part;
    bool cs_valid = false; // code sums of the stored vectors (u8 engine) computed for the current upload
    u32 cs_n = 0;
    bool zn_valid = false, zn_zero = false; // "a stored vector has a zero norm" for the current upload (cosine's CalculationError screen)
    u32 zn_n = 0;
    Buf out;                  // [ids B x k | scores B x k | counts B | zero-norm flag, overflow flag]: ONE copy back per call
    void *h_out = nullptr;    // its pinned landing area
    size_t h_out_cap = 0;
    hipError_t need_host(size_t bytes) {
        if (bytes <= h_out_cap && h_out) return hipSuccess;
        if (h_out) (void)hipHostFree(h_out);
        h_out = nullptr; h_out_cap = 0;
        hipError_t e = hipHostMalloc(&h_out, bytes);
        if (e == hipSuccess) h_out_cap = bytes;
        return e;
    }
    std::vector<hipEvent_t> evs;
    hipError_t need(Buf &b, size_t bytes) {
        if (bytes <= b.cap && b.p) return hipSuccess;
        if (b.p) (void)hipFree(b.p);
        b.p = nullptr; b.cap = 0;
        hipError_t e = hipMalloc(&b.p, bytes ? bytes : 1);
        if (e == hipSuccess) b.cap = bytes ? bytes : 1;
        return e;
    }
    hipError_t event(size_t i, hipEvent_t *out) {
        while (evs.size() <= i) {
            hipEvent_t ev = nullptr;
            hipError_t e = hipEventCreate(&ev);
            if (e != hipSuccess) return e;
            evs.push_back(ev);
        }
        *out = evs[i];
        return hipSuccess;
    }
    ~FlatWs() {
        for (Buf *b : {&q, &zero, &qm, &qrm, &qc, &qd, &qdp, &qs, &cs, &pool, &thr, &app, &appcnt, &oi, &os, &oc, &scores, &part, &out})
            if (b->p) (void)hipFree(b->p);
        if (h_out) (void)hipHostFree(h_out);
        for (hipEvent_t ev : evs) (void)hipEventDestroy(ev);
    }
};

void cos_flat_ws_release(cos_index *ix) { // cos_index_destroy, and every upload that replaces the stored vectors
    delete ix->flat_ws;
    ix->flat_ws = nullptr;
}

// ------------------------------------------------------------------------------------------------
// cos_flat_search_batch: exhaustive search over the index's quantized codes (i8 MFMA) + exact rerank of the best 5k
// ------------------------------------------------------------------------------------------------
extern "C" int32_t cos_flat_search_batch(cos_index *ix, const float *queries, uint32_t B, uint32_t top_k, uint32_t *out_ids, float *out_scores,
                                         uint32_t *out_counts, cos_flat_stats *stats) {
    if (!ix || !queries || !out_ids || !out_scores || !out_counts || B == 0) return cos_fail(COS_ERR_INVALID, "bad argument");
    if (!ix->have_vectors) return cos_fail(COS_ERR_NOT_READY, "upload vectors first");
    if (top_k == 0 || 5 * top_k > (u32)SEL) return cos_fail(COS_ERR_UNIMPLEMENTED, "flat search keeps 64 survivors: top_k must be in [1, 12]");
    if (ix->eng != ENG_U8 && ix->eng != ENG_Q2) return cos_fail(COS_ERR_UNIMPLEMENTED, "flat search over codes implements u8 and quaternary storage");
    int32_t rc = cos_set_device(ix);
    if (rc) return rc;
    const u32 dim = ix->p.dim, n = ix->n;
    const u32 kdims = ix->eng == ENG_U8 ? (u32)((ix->row_stride + 63) / 64 * 64) : (u32)(ix->row_stride / 16) * 64;
    // Scan schedule.  The first candidates (SEED of them) go through the unfused path — score matrix + segmented selection —
    // and seed every query's threshold; from then on chunks grow 8x (128 K, 1 M, 4 M ...) and the fused GEMM appends only what
    // beats the threshold: a chunk 8x the size of everything seen before lets ~64 * 8 entries per query through, an eighth of
    // the append capacity.  An adversarially ordered corpus can still overflow it; that is detected on the device and the
    // call is repeated on the unfused path, so the result never depends on the shortcut.  Tuning knob flat_unfused = 1 forces that path.
    const bool allow_fused = tune_or(TUNE_FLAT_UNFUSED, 0) == 0;
    const int pf = (int)tune_or(TUNE_FLAT_PF, 1); // k panels prefetched into registers (1..3); results do not depend on it
    // fused chunks of quaternary codes run on the query-resident kernel when K has an instantiation (tuning knob flat_tile_kernel = 1: never)
    const bool use_areg = ix->eng == ENG_Q2 && flat_scan_supported(kdims) && tune_or(TUNE_FLAT_TILE_KERNEL, 0) == 0;
    const bool use_fp4 = use_areg && tune_or(TUNE_FLAT_FP4, 1) != 0; // e2m1 digits on the scaled MFMA (kernels_scan.hip flat_scan_q2_fp4); 0 = i8 digits
    // fused chunks of u8 codes too (round 6: flat_scan_u8_areg), when a row is exactly a supported number of 64-byte chunks
    const bool use_u8q = ix->eng == ENG_U8 && ix->row_stride == (u64)kdims && flat_scan_supported(kdims) && tune_or(TUNE_FLAT_TILE_KERNEL, 0) == 0; // (1024 dims: 32 query rows per wave instead of 64)
    int n_cus = 0;
    if (hipDeviceGetAttribute(&n_cus, hipDeviceAttributeMultiprocessorCount, ix->p.device) != hipSuccess || n_cus <= 0) n_cus = 256;
    constexpr u32 SEED = 16384, APP_CAP = 4096;
    u32 chunk = (u32)std::min<u64>(1u << 20, ((1ull << 31) / B / 4) / CN * CN); // unfused: the [B][chunk] score buffer stays <= 2 GiB
    chunk = std::min(n, std::max<u32>(chunk, CN));
    hipStream_t st = ix->own_stream;
    {
        std::lock_guard<std::mutex> g(ix->mu);
        if (!ix->flat_ws) ix->flat_ws = new FlatWs();
    }
    FlatWs *W = ix->flat_ws;
    std::lock_guard<std::mutex> wg(W->mu);
    float *d_q = nullptr, *d_qm = nullptr, *d_qrm = nullptr, *d_scores = nullptr, *d_os = nullptr;
    uint8_t *d_qc = nullptr, *d_qd = nullptr, *d_qdp = nullptr;
    u32 *d_qs = nullptr, *d_cs = nullptr, *d_oi = nullptr, *d_oc = nullptr, *d_zero = nullptr, *d_appcnt = nullptr;
    u64 *d_pool = nullptr, *d_part = nullptr, *d_thr = nullptr, *d_app = nullptr;
    size_t n_ev = 0; // (start, stop) events of the GEMM launches, read after the last one
    float gemm_ms = 0.f;
    u32 launches = 0;
    double streamed = 0.0;
    bool zero = false;
    hipError_t e = hipSuccess;
    for (int attempt = 0; attempt < 2 && e == hipSuccess; attempt++) {
        const bool fused = allow_fused && attempt == 0 && n > SEED;
        const u32 first = fused ? SEED : chunk;                              // size of the (unfused) first chunk
        const u64 s_stride = ((u64)std::min(first, n) + 63) & ~63ull;
        const u32 S = select_segments(B, std::min(first, n));
        if (attempt == 0) {
#define WS_NEED(buf, ptr, bytes) if (e == hipSuccess) { e = W->need(W->buf, (bytes)); ptr = (decltype(ptr))W->buf.p; }
            WS_NEED(q, d_q, (size_t)B * dim * 4)
            WS_NEED(zero, d_zero, 8)
            if (e == hipSuccess) e = hipMemsetAsync(d_zero, 0, 8, st);
            WS_NEED(qm, d_qm, (size_t)B * 4)
            WS_NEED(qrm, d_qrm, (size_t)B * 4)
            WS_NEED(qc, d_qc, (size_t)B * ix->row_stride)
            WS_NEED(qd, d_qd, (size_t)B * kdims)
            if (use_areg) WS_NEED(qdp, d_qdp, (size_t)B * kdims)
            WS_NEED(qs, d_qs, (size_t)B * 4)
            WS_NEED(cs, d_cs, ((size_t)n + 1) * 4)
            WS_NEED(pool, d_pool, (size_t)B * SEL * 8)
            WS_NEED(thr, d_thr, (size_t)B * 8)
            WS_NEED(app, d_app, (size_t)B * APP_CAP * 8)
            WS_NEED(appcnt, d_appcnt, (size_t)B * 4)
            WS_NEED(oi, d_oi, (size_t)B * top_k * 4)
            WS_NEED(os, d_os, (size_t)B * top_k * 4)
            WS_NEED(oc, d_oc, (size_t)B * 4)
            if (e == hipSuccess) e = hipMemcpyAsync(d_q, queries, (size_t)B * dim * 4, hipMemcpyHostToDevice, st);
            if (e == hipSuccess) e = launch_quantize_rows(ix->eng, d_q, dim, B, dim, ix->p.range_lo, ix->p.range_hi, d_qc, ix->row_stride, d_qm, d_qrm, st);
            if (e == hipSuccess && ix->eng == ENG_U8) {
                hipLaunchKernelGGL(code_sums_kernel, dim3((B + 3) / 4), dim3(256), 0, st, d_qc, ix->row_stride, B, d_qs);
                if (!W->cs_valid || W->cs_n != n) { // the stored vectors' sums: once per upload
                    hipLaunchKernelGGL(code_sums_kernel, dim3((n + 3) / 4), dim3(256), 0, st, ix->d_codes, ix->row_stride, n, d_cs);
                    W->cs_valid = true;
                    W->cs_n = n;
                }
                e = hipGetLastError();
            }
            if (e == hipSuccess && ix->eng == ENG_Q2) {
                const u64 pieces = (u64)B * (kdims / 16);
                hipLaunchKernelGGL(expand_q2_digits_kernel, dim3((u32)((pieces + 255) / 256)), dim3(256), 0, st, d_qc, ix->row_stride, B, kdims, d_qd);
                e = hipGetLastError();
                if (e == hipSuccess && use_areg) e = launch_flat_scan_expand_queries(d_qc, ix->row_stride, B, kdims, d_qdp, st, use_fp4);
            }
            // zero-norm screening (cosine): the reference aborts a search on the first zero denominator it meets; an
            // exhaustive scan meets every vector, so any zero |q| or zero |v| is a CalculationError for the call.
            // The stored vectors' answer is computed once per upload (and read back then); the queries' flag stays on the device and comes
            // back with the results: a call with a zero-norm query runs its scan for nothing and then reports the error — until round 6
            // every call stopped here for a device round trip (~50 us of a 1.9 ms call) before its first GEMM.
            if (e == hipSuccess && ix->p.metric == COS_METRIC_COSINE) {
                if (!W->zn_valid || W->zn_n != n) {
                    u32 hz = 0;
                    hipLaunchKernelGGL(any_zero_kernel, dim3(1024), dim3(256), 0, st, (const float *)ix->d_mags, n, d_zero);
                    e = hipGetLastError();
                    if (e == hipSuccess) e = hipMemcpyAsync(&hz, d_zero, 4, hipMemcpyDeviceToHost, st);
                    if (e == hipSuccess) e = hipStreamSynchronize(st);
                    if (e == hipSuccess) { W->zn_valid = true; W->zn_n = n; W->zn_zero = hz != 0; }
                    if (e == hipSuccess) e = hipMemsetAsync(d_zero, 0, 4, st);
                }
                if (e == hipSuccess && W->zn_zero) { zero = true; break; }
                if (e == hipSuccess) {
                    hipLaunchKernelGGL(any_zero_kernel, dim3(64), dim3(256), 0, st, (const float *)d_qm, B, d_zero);
                    e = hipGetLastError();
                }
            }
        }
        WS_NEED(scores, d_scores, (size_t)B * s_stride * 4)
        WS_NEED(part, d_part, (size_t)B * S * SEL * 8)
#undef WS_NEED
        if (e == hipSuccess) {
            hipLaunchKernelGGL(flat_reset_kernel, dim3(64), dim3(256), 0, st, d_pool, (u64)B * SEL, d_thr, d_appcnt, B, d_zero + 1); // (d_zero + 1: overflow flag of the fused path)
            e = hipGetLastError();
        }
        FusedOut fo{d_thr, d_app, d_appcnt, APP_CAP, d_qd};
        u32 n0 = 0, seen = 0;
        while (n0 < n && e == hipSuccess) {
            const bool use_fused = fused && n0 > 0;
            // chunk cap: the tile kernel's grid (and its event granularity) like 4 M; the query-resident kernel is persistent and
            // pays ~35 us of prologue per launch, so it takes everything that is left once the 8x rule allows it
            u32 nc = use_fused ? (u32)std::min<u64>((u64)seen * 8, use_areg || use_u8q ? (1ull << 31) : (1ull << 22)) : first;
            nc = std::min(nc, n - n0);
            dim3 grid((nc + CN - 1) / CN, (B + CM - 1) / CM);
            hipEvent_t ev0 = nullptr, ev1 = nullptr;
            e = W->event(n_ev, &ev0);
            if (e == hipSuccess) e = W->event(n_ev + 1, &ev1);
            if (e == hipSuccess) { n_ev += 2; e = hipEventRecord(ev0, st); }
            if (e != hipSuccess) break;
#define FLAT_ARGS d_qc, d_qm, d_qs, B, ix->d_codes, ix->d_mags, d_cs, ix->row_stride, n0, nc, kdims, ix->p.metric, d_scores, s_stride, fo
#define FLAT_LAUNCH(E, F, P) hipLaunchKernelGGL((flat_codes_gemm_i8<E, F, P>), grid, dim3(512), 0, st, FLAT_ARGS)
#define FLAT_LAUNCH_PF(E, F) do { if (pf == 2) FLAT_LAUNCH(E, F, 2); else if (pf == 3) FLAT_LAUNCH(E, F, 3); else FLAT_LAUNCH(E, F, 1); } while (0)
            if (use_fused && use_areg) {
                e = launch_flat_scan(kdims, (u32)n_cus, st, d_qdp, d_qm, B, ix->d_codes, ix->d_mags, ix->row_stride, n0, nc, ix->p.metric, fo, use_fp4);
                if (e != hipSuccess) break;
            } else if (use_fused && use_u8q) {
                e = launch_flat_scan_u8(kdims, (u32)n_cus, st, d_qc, d_qs, d_qm, B, ix->d_codes, d_cs, ix->d_mags, ix->row_stride, n0, nc, ix->p.metric, fo);
                if (e != hipSuccess) break;
            } else if (ix->eng == ENG_U8) {
                if (use_fused) FLAT_LAUNCH_PF(ENG_U8, true); else FLAT_LAUNCH_PF(ENG_U8, false);
            } else {
                if (use_fused) FLAT_LAUNCH_PF(ENG_Q2, true); else FLAT_LAUNCH_PF(ENG_Q2, false);
            }
#undef FLAT_LAUNCH_PF
#undef FLAT_LAUNCH
#undef FLAT_ARGS
            e = hipGetLastError();
            if (e == hipSuccess) e = hipEventRecord(ev1, st);
            if (e == hipSuccess) {
                if (use_fused) {
                    hipLaunchKernelGGL(flat_select_append, dim3(B), dim3(64), 0, st, (const u64 *)d_app, d_appcnt, APP_CAP, B, d_pool, d_thr, d_zero + 1);
                    e = hipGetLastError();
                } else
                    e = launch_select(d_scores, s_stride, B, n0, nc, d_part, S, d_pool, st, d_thr);
            }
            launches++;
            streamed += (double)nc * (double)ix->row_stride * (double)((B + CM - 1) / CM);
            n0 += nc;
            seen += nc;
        }
        // rerank, then ONE copy back: results, the zero-norm flag of the queries and the overflow flag of the fused path.  (Until round 6: a
        // device round trip for the overflow flag, then the rerank, then three pageable copies.)  An overflow repeats the scan unfused.
        const size_t nk = (size_t)B * top_k, out_words = 2 * nk + B + 2;
        if (e == hipSuccess) e = W->need(W->out, out_words * 4);
        if (e == hipSuccess) e = W->need_host(out_words * 4);
        if (e != hipSuccess) break;
        hipLaunchKernelGGL(flat_rerank_top5k, dim3(B), dim3(64), (((size_t)dim * 4 + 15) & ~(size_t)15), st, d_q, (u64)dim, d_qrm, B, ix->d_raw, (u64)dim,
                           ix->d_raw_mags, dim, d_pool, 5 * top_k, top_k, ix->p.id_base, d_oi, d_os, d_oc);
        hipLaunchKernelGGL(flat_pack_out_kernel, dim3((u32)((std::max<size_t>(nk, B) + 255) / 256)), dim3(256), 0, st, d_oi, d_os, d_oc, d_zero, B, top_k, (u32 *)W->out.p);
        e = hipGetLastError();
        if (e == hipSuccess) e = hipMemcpyAsync(W->h_out, W->out.p, out_words * 4, hipMemcpyDeviceToHost, st);
        if (e == hipSuccess) e = hipStreamSynchronize(st);
        if (e != hipSuccess) break;
        const u32 *ho = (const u32 *)W->h_out;
        zero = ho[2 * nk + B] != 0;
        const u32 hover = ho[2 * nk + B + 1];
        if (zero) break;
        if (fused && hover != 0) continue; // the append buffer overflowed: repeat unfused
        memcpy(out_ids, ho, nk * 4);
        memcpy(out_scores, ho + nk, nk * 4);
        memcpy(out_counts, ho + 2 * nk, (size_t)B * 4);
        break;
    }
    if (e == hipSuccess && !zero) {
        for (size_t i = 0; i + 1 < n_ev && e == hipSuccess; i += 2) {
            float ms = 0.f;
            e = hipEventElapsedTime(&ms, W->evs[i], W->evs[i + 1]);
            gemm_ms += ms;
        }
    }
    if (stats) {
        stats->gemm_ms = gemm_ms;
        stats->gemm_launches = launches;
        stats->int8_ops = 2.0 * (double)B * (double)n * (double)kdims;
        stats->code_bytes = streamed;
    }
    HIP_TRY(e);
    if (zero) return cos_fail(COS_ERR_CALCULATION, "zero-norm query or stored vector: DistanceError::CalculationError (cosine.rs:228-232)");
    return COS_OK;
}

// ------------------------------------------------------------------------------------------------
// Level table of the walk (WalkArgs::tab, engine.hip ensure_level_table / run_search): similarity of every query of a big launch to
// every node of the graph's small upper levels, as ONE pass of the exact-integer i8 GEMM above with its unfused epilogue —
// `(u32 dot) as f32` and the IEEE quotient by |q| * |v|, the very operations the walk applies to a row it dots itself
// (cosine.rs:223-235), so the table holds the bits the walk would have computed.  The walk then reads 4 bytes per evaluation on
// those levels.  The nodes' code rows are gathered once per graph into a compact operand (level_table_gather).
// ------------------------------------------------------------------------------------------------
namespace {
__global__ __launch_bounds__(256) void level_table_gather_kernel(const uint8_t *__restrict__ codes, const float *__restrict__ mags, u64 row_stride,
                                                                 const u32 *__restrict__ node_vec, u32 n, u32 col0, uint8_t *__restrict__ tcodes,
                                                                 float *__restrict__ tmags) {
    const int lane = threadIdx.x & 63;
    const u32 node = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (node >= n) return;
    const u32 row = node_vec[node];
    const uint4 *src = (const uint4 *)(codes + (u64)row * row_stride);
    uint4 *dst = (uint4 *)(tcodes + (u64)(col0 + node) * row_stride);
    for (u32 c = lane; c < (u32)(row_stride / 16); c += 64) dst[c] = src[c];
    if (lane == 0) tmags[col0 + node] = mags[row];
}
} // namespace

namespace cosdev {

hipError_t launch_level_table_gather(const uint8_t *codes, const float *mags, u64 row_stride, const u32 *node_vec, u32 n, u32 col0,
                                     uint8_t *tcodes, float *tmags, hipStream_t st) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(level_table_gather_kernel, dim3((n + 3) / 4), dim3(256), 0, st, codes, mags, row_stride, node_vec, n, col0, tcodes, tmags);
    return hipGetLastError();
}

hipError_t launch_code_sums(const uint8_t *codes, u64 row_stride, u32 n, u32 *sums, hipStream_t st) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(code_sums_kernel, dim3((n + 3) / 4), dim3(256), 0, st, codes, row_stride, n, sums);
    return hipGetLastError();
}

// storages with a level table: u8 codes (any row length: the tile kernel covers what the query-resident one does not) and quaternary
// codes whose dims the query-resident kernel is instantiated for (the tile kernel's quaternary path assumes whole 64-dim chunks too)
bool level_table_eng_supported(int eng, u64 row_stride) {
    return eng == ENG_U8 || (eng == ENG_Q2 && level_table_areg_supported(ENG_Q2, row_stride));
}

// tab[q][c] = the integer dot of query q with table column c (dot_product_u8 / dot_product_quaternary) `as f32` for q < B, c < ncols;
// u8 and quaternary codes.  The walk divides by |q| * |v|
// for the entries it reads (walk_kernel.inc, table levels): the GEMM's epilogue is an add, a convert and a store.  The
// query-resident kernel (kernels_scan.hip level_table_areg) wherever the code rows are whole 64-byte chunks with an
// instantiation (768, 1024, ...), the 256 x 128 tile kernel otherwise (and with tuning knob walk_table_gemm = 0); same table.
hipError_t launch_level_table(int eng, const uint8_t *qcodes, const float *qmags, u32 *qsums /*[B] scratch (u8)*/, uint8_t *qdig /*[B][dims] scratch (quaternary)*/,
                              u32 B, const uint8_t *tcodes, const float *tmags, const u32 *tcsums, u64 row_stride, u32 ncols, float *tab, u64 tab_stride,
                              u32 n_cus, hipStream_t st) {
    if (B == 0 || ncols == 0) return hipSuccess;
    const bool q2 = eng == ENG_Q2;
    const u32 kdims = q2 ? (u32)(row_stride / 16) * 64 : (u32)((row_stride + 63) / 64 * 64);
    hipError_t e = hipSuccess;
    if (!q2) e = launch_code_sums(qcodes, row_stride, B, qsums, st);
    if (e != hipSuccess) return e;
    if (level_table_areg_supported(eng, row_stride) && tune_or(TUNE_WALK_TABLE_GEMM, 1) != 0) {
        if (q2) { // the queries' planes -> the permuted i8 digit rows the kernel keeps resident
            e = launch_flat_scan_expand_queries(qcodes, row_stride, B, kdims, qdig, st, false);
            if (e != hipSuccess) return e;
        }
        // (one workgroup per CU; on half the CUs — so that the previous launch's walk keeps the other half — the GEMM takes 1.31 instead
        // of 0.85 ms and the step 6.57 instead of 6.49: profiles/r05_table_gemm_half_the_cus_probe_not_kept.jsonl)
        return launch_level_table_areg(eng, n_cus ? n_cus : 256u, st, q2 ? qdig : qcodes, (const u32 *)qsums, B, tcodes, tcsums, row_stride, ncols, tab, tab_stride);
    }
    const u32 metric = 1u; // the tile kernel's unfused epilogue with the dot-product metric: the converted integer dot, no quotient
    dim3 grid((ncols + CN - 1) / CN, (B + CM - 1) / CM);
    FusedOut fo{nullptr, nullptr, nullptr, 0u, nullptr};
    // two k panels in flight (one and three were measured slower in round 4)
    if (q2)
        hipLaunchKernelGGL((flat_codes_gemm_i8<ENG_Q2, false, 2>), grid, dim3(512), 0, st, qcodes, qmags, (const u32 *)qsums, B, tcodes, tmags, tcsums, row_stride, 0u,
                           ncols, kdims, metric, tab, tab_stride, fo);
    else
        hipLaunchKernelGGL((flat_codes_gemm_i8<ENG_U8, false, 2>), grid, dim3(512), 0, st, qcodes, qmags, (const u32 *)qsums, B, tcodes, tmags, tcsums, row_stride, 0u,
                           ncols, kdims, metric, tab, tab_stride, fo);
    return hipGetLastError();
}

} // namespace cosdev
