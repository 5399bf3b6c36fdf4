// kernels_misc.hip — small gfx950 kernels around the walk: S-way top-k merge for the sharded index
//  and the src/distance operator on explicit pairs (cos_distance_batch).
#include <hip/hip_runtime.h>

#include <cstring>
#include <vector>

#include "engine_internal.h"

using namespace cosdev;

namespace {

// One wave per query: gather S per-shard lists (already sorted, but any order is accepted), sort by
// (total_cmp score desc, larger id first) and emit the best k.  Shard s's rows start at s*stride_rows
// elements in ids/scores and its counts at s*stride_counts in counts, so both the dense [S][B][k] layout
// and a packed per-shard record [ids B*k | scores B*k | counts B] (ONE all-gather) are accepted.
template <int R>
__global__ __launch_bounds__(64) void merge_topk_kernel(const u32 *__restrict__ ids, const float *__restrict__ scores,
                                                        const u32 *__restrict__ counts, u64 stride_rows, u64 stride_counts,
                                                        u32 S, u32 B, u32 k, u32 *__restrict__ out_ids,
                                                        float *__restrict__ out_scores, u32 *__restrict__ out_counts) {
    const int lane = threadIdx.x;
    const u32 q = blockIdx.x;
    if (q >= B) return;
    u64 key[R];
    u32 total = 0;
#pragma unroll
    for (int r = 0; r < R; r++) {
        const u32 e = (u32)lane * R + r;
        key[r] = 0ull;
        if (e < S * k) {
            const u32 s = e / k, j = e % k;
            if (j < counts[(u64)s * stride_counts + q]) {
                const u64 off = (u64)s * stride_rows + (u64)q * k + j;
                key[r] = pack_key(simkey(scores[off]), ids[off]);
            }
        }
    }
    for (u32 s = 0; s < S; s++) total += counts[(u64)s * stride_counts + q];
    bitonic_sort_desc<R>(key, lane);
    const u32 n = total < k ? total : k;
#pragma unroll
    for (int r = 0; r < R; r++) {
        const u32 e = (u32)lane * R + r;
        if (e < n) {
            out_ids[(u64)q * k + e] = (u32)key[r];
            out_scores[(u64)q * k + e] = simkey_inv((u32)(key[r] >> 32));
        }
    }
    if (lane == 0) out_counts[q] = n;
}

int32_t merge_topk_launch(const uint32_t *d_ids, const float *d_scores, const uint32_t *d_counts, u64 stride_rows, u64 stride_counts,
                          uint32_t S, uint32_t B, uint32_t k, uint32_t *d_out_ids, float *d_out_scores, uint32_t *d_out_counts,
                          int32_t device, void *stream) {
    if (!d_ids || !d_scores || !d_counts || !d_out_ids || !d_out_scores || !d_out_counts || S == 0 || B == 0 || k == 0)
        return cos_fail(COS_ERR_INVALID, "bad argument");
    HIP_TRY(hipSetDevice(device));
    const u32 total = S * k;
    hipStream_t st = (hipStream_t)stream;
    dim3 grid(B), block(64);
#define LAUNCH(R) hipLaunchKernelGGL(merge_topk_kernel<R>, grid, block, 0, st, d_ids, d_scores, d_counts, stride_rows, stride_counts, S, B, k, d_out_ids, d_out_scores, d_out_counts)
    if (total <= 64) LAUNCH(1);
    else if (total <= 128) LAUNCH(2);
    else if (total <= 256) LAUNCH(4);
    else if (total <= 512) LAUNCH(8);
    else if (total <= 1024) LAUNCH(16);
    else return cos_fail(COS_ERR_UNIMPLEMENTED, "S*k > 1024 not supported by the merge kernel");
#undef LAUNCH
    HIP_TRY(hipGetLastError());
    return COS_OK;
}

} // namespace

extern "C" int32_t cos_merge_topk_device(const uint32_t *d_ids, const float *d_scores, const uint32_t *d_counts, uint32_t S, uint32_t B,
                                         uint32_t k, uint32_t *d_out_ids, float *d_out_scores, uint32_t *d_out_counts, int32_t device,
                                         void *stream) {
    return merge_topk_launch(d_ids, d_scores, d_counts, (u64)B * k, B, S, B, k, d_out_ids, d_out_scores, d_out_counts, device, stream);
}

extern "C" int32_t cos_merge_topk_packed_device(const uint32_t *d_packed, uint32_t S, uint32_t B, uint32_t k, uint32_t *d_out_ids,
                                                float *d_out_scores, uint32_t *d_out_counts, int32_t device, void *stream) {
    if (!d_packed) return cos_fail(COS_ERR_INVALID, "bad argument");
    const u64 rec = (u64)B * (2ull * k + 1ull);
    return merge_topk_launch(d_packed, reinterpret_cast<const float *>(d_packed + (u64)B * k), d_packed + 2ull * B * k, rec, rec, S, B, k,
                             d_out_ids, d_out_scores, d_out_counts, device, stream);
}

// ================================================================================================
// cos_distance_batch — DistanceMetric::calculate (models/types.rs:469) on explicit pairs
// ================================================================================================
#include "dot_engines.h"

namespace {

struct DistArgs {
    const uint8_t *x_codes, *y_codes; // device layout
    const float *x_mags, *y_mags;
    const u32 *pair_x, *pair_y;
    u64 row_stride;
    u32 n_pairs, dim, metric, nchunks;
    float *out;
    int32_t *status;
};

// one wave per pair
template <int ENG>
__global__ __launch_bounds__(64) void distance_pairs_kernel(const DistArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int lane = threadIdx.x;
    const u32 p = blockIdx.x;
    if (p >= a.n_pairs) return;
    const uint8_t *xr = a.x_codes + (u64)a.pair_x[p] * a.row_stride;
    const uint8_t *yr = a.y_codes + (u64)a.pair_y[p] * a.row_stride;
    const float xm = a.x_mags[a.pair_x[p]], ym = a.y_mags[a.pair_y[p]];
    float val = 0.0f;
    int32_t st = COS_OK;
    if (a.metric == COS_METRIC_COSINE || a.metric == COS_METRIC_DOT) {
        float dotf;
        if constexpr (ENG == ENG_F32) {
            float *qf = (float *)smem_raw;
            for (u32 i = lane; i < (u32)(a.row_stride / 4); i += 64) qf[i] = ((const float *)xr)[i];
            dotf = f32_pair_dot((const float *)yr, qf, a.dim, lane & 1);
            if (a.metric == COS_METRIC_DOT) st = COS_ERR_STORAGE_MISMATCH; // dotproduct.rs: no FullPrecisionFP arm
        } else if constexpr (ENG == ENG_F16) { // dot_product_f16 (dot_product.rs:13-19): sequential sum of f32(a) * f32(b), one lane's chain
            float *qf = (float *)smem_raw;
            for (u32 i = lane; i < a.dim; i += 64) qf[i] = __half2float(((const __half *)xr)[i]);
            __builtin_amdgcn_s_waitcnt(0);
            __builtin_amdgcn_wave_barrier();
            dotf = __uint_as_float(readlane_u32(__float_as_uint(f16_lane_dot(yr, qf, a.dim)), 0));
        } else {
            u32 acc = 0;
            for (u32 c = lane; c < a.nchunks; c += 64) acc = chunk_dot<ENG>(*(const uint4 *)(xr + (u64)c * 16), *(const uint4 *)(yr + (u64)c * 16), acc);
#pragma unroll
            for (int m = 32; m > 0; m >>= 1) acc += (u32)__shfl_xor((int)acc, m, 64);
            dotf = (float)acc;
        }
        if (a.metric == COS_METRIC_COSINE) { // cosine.rs:223-235
            const float den = __fmul_rn(xm, ym);
            if (den == 0.0f) st = COS_ERR_CALCULATION;
            else val = __fdiv_rn(dotf, den);
        } else
            val = dotf;
    } else if (a.metric == COS_METRIC_HAMMING) { // hamming.rs:60-98 (integer counts < 2^24: exact in f32 in any order)
        if constexpr (ENG == ENG_F32 || ENG == ENG_F16) st = COS_ERR_STORAGE_MISMATCH;
        else {
            u32 acc = 0;
            for (u32 c = lane; c < a.nchunks; c += 64) {
                const uint4 x = *(const uint4 *)(xr + (u64)c * 16), y = *(const uint4 *)(yr + (u64)c * 16);
                acc += __popc(x.x ^ y.x) + __popc(x.y ^ y.y) + __popc(x.z ^ y.z) + __popc(x.w ^ y.w);
            }
#pragma unroll
            for (int m = 32; m > 0; m >>= 1) acc += (u32)__shfl_xor((int)acc, m, 64);
            val = (float)acc;
        }
    } else if (a.metric == COS_METRIC_EUCLIDEAN) { // euclidean.rs:42-53: sequential f32 sum of (diff*diff) as i16 (wrapping)
        if constexpr (ENG == ENG_U8) {
            float acc = -0.0f;
            if (lane == 0)
                for (u32 i = 0; i < a.dim; i++) {
                    const int16_t diff = (int16_t)((int16_t)xr[i] - (int16_t)yr[i]);
                    const int16_t sq = (int16_t)(diff * diff);
                    acc = __fadd_rn(acc, (float)sq);
                }
            val = sqrtf(__uint_as_float(readlane_u32(__float_as_uint(acc), 0)));
        } else if constexpr (ENG == ENG_Q2)
            st = COS_ERR_UNIMPLEMENTED; // euclidean.rs:34-37 unimplemented!()
        else
            st = COS_ERR_STORAGE_MISMATCH;
    }
    if (lane == 0) { a.out[p] = st == COS_OK ? val : 0.0f; a.status[p] = st; }
}

// reference layout -> device layout (host side)
void rows_to_device_layout(int eng, u32 dim, const uint8_t *ref, size_t cb, u32 n, u64 row_stride, std::vector<uint8_t> &dev) {
    dev.assign((size_t)n * row_stride, 0);
    for (u32 r = 0; r < n; r++) {
        const uint8_t *s = ref + (size_t)r * cb;
        uint8_t *d = dev.data() + (size_t)r * row_stride;
        if (eng == ENG_U8) memcpy(d, s, dim);
        else if (eng == ENG_F32) memcpy(d, s, (size_t)dim * 4);
        else {
            const u32 pb = (dim + 7) / 8;
            for (u32 p = 0; p < 2; p++)
                for (u32 b = 0; b < pb; b++) d[(size_t)(b / 8) * 16 + p * 8 + (b % 8)] = s[(size_t)p * pb + b];
        }
    }
}

} // namespace

namespace cosdev {
// DistanceMetric::calculate for explicit (row, row) pairs of ONE resident code table (cos_index_delete's re-scoring of an orphan's
// candidates, vector_store.rs:1326-1337): the operator's own kernel on device arrays.  Every storage an index can have (cosine / dot).
hipError_t launch_index_pair_distances(int eng, const uint8_t *codes, const float *mags, u64 row_stride, u32 nchunks, u32 dim, u32 metric, const u32 *d_pair_x,
                                       const u32 *d_pair_y, u32 n_pairs, float *d_out, int32_t *d_status, hipStream_t st) {
    if (n_pairs == 0) return hipSuccess;
    DistArgs a{codes, codes, mags, mags, d_pair_x, d_pair_y, row_stride, n_pairs, dim, metric, nchunks, d_out, d_status};
    dim3 grid(n_pairs), block(64);
    if (eng == ENG_U8) hipLaunchKernelGGL(distance_pairs_kernel<ENG_U8>, grid, block, 0, st, a);
    else if (eng == ENG_Q2) hipLaunchKernelGGL(distance_pairs_kernel<ENG_Q2>, grid, block, 0, st, a);
    else if (eng == ENG_F32) hipLaunchKernelGGL(distance_pairs_kernel<ENG_F32>, grid, block, (size_t)row_stride, st, a);
    else if (eng == ENG_Q1) hipLaunchKernelGGL(distance_pairs_kernel<ENG_Q1>, grid, block, 0, st, a);
    else if (eng == ENG_Q3) hipLaunchKernelGGL(distance_pairs_kernel<ENG_Q3>, grid, block, 0, st, a);
    else if (eng == ENG_F16) hipLaunchKernelGGL(distance_pairs_kernel<ENG_F16>, grid, block, ((size_t)dim * 4 + 15) & ~(size_t)15, st, a);
    else return hipErrorInvalidValue;
    return hipGetLastError();
}
} // namespace cosdev

extern "C" int32_t cos_distance_batch(uint32_t metric, uint32_t storage, uint32_t resolution, uint32_t dim, const void *x_codes,
                                      const float *x_mags, uint32_t nx, const void *y_codes, const float *y_mags, uint32_t ny,
                                      const uint32_t *pair_x, const uint32_t *pair_y, uint32_t n_pairs, float *out, int32_t *status) {
    if (!x_codes || !y_codes || !x_mags || !y_mags || !pair_x || !pair_y || !out || !status || dim == 0 || n_pairs == 0)
        return cos_fail(COS_ERR_INVALID, "bad argument");
    if (metric > COS_METRIC_DOT) return cos_fail(COS_ERR_INVALID, "unknown metric");
    int eng;
    u64 row_stride;
    u32 nchunks = 0;
    if (storage == COS_STORAGE_U8) { eng = ENG_U8; row_stride = ((u64)dim + 15) & ~15ull; nchunks = (u32)(row_stride / 16); }
    else if (storage == COS_STORAGE_SUBBYTE && resolution == 2) { eng = ENG_Q2; nchunks = (dim + 63) / 64; row_stride = (u64)nchunks * 16; }
    else if (storage == COS_STORAGE_F32) { eng = ENG_F32; row_stride = ((u64)dim * 4 + 15) & ~15ull; }
    else if (storage == COS_STORAGE_F16 || storage == COS_STORAGE_SUBBYTE) { eng = -1; row_stride = 0; }
    else return cos_fail(COS_ERR_INVALID, "unknown storage kind");
    for (u32 p = 0; p < n_pairs; p++)
        if (pair_x[p] >= nx || pair_y[p] >= ny) return cos_fail(COS_ERR_INVALID, "pair %u out of range", p);
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return cos_fail(COS_ERR_NO_DEVICE, "no HIP device visible; the GPU path has no CPU fallback");
    if (eng < 0) return distance_ref_layout(metric, storage, resolution, dim, x_codes, x_mags, nx, y_codes, y_mags, ny, pair_x, pair_y, n_pairs, out, status);
    const size_t cb = cos_code_bytes(storage, resolution, dim);
    std::vector<uint8_t> hx, hy;
    rows_to_device_layout(eng, dim, (const uint8_t *)x_codes, cb, nx, row_stride, hx);
    rows_to_device_layout(eng, dim, (const uint8_t *)y_codes, cb, ny, row_stride, hy);
    uint8_t *dx = nullptr, *dy = nullptr;
    float *dxm = nullptr, *dym = nullptr, *dout = nullptr;
    u32 *dpx = nullptr, *dpy = nullptr;
    int32_t *dst = nullptr;
    hipError_t e = hipMalloc(&dx, hx.size());
    if (e == hipSuccess) e = hipMalloc(&dy, hy.size());
    if (e == hipSuccess) e = hipMalloc(&dxm, (size_t)nx * 4);
    if (e == hipSuccess) e = hipMalloc(&dym, (size_t)ny * 4);
    if (e == hipSuccess) e = hipMalloc(&dpx, (size_t)n_pairs * 4);
    if (e == hipSuccess) e = hipMalloc(&dpy, (size_t)n_pairs * 4);
    if (e == hipSuccess) e = hipMalloc(&dout, (size_t)n_pairs * 4);
    if (e == hipSuccess) e = hipMalloc(&dst, (size_t)n_pairs * 4);
    if (e == hipSuccess) e = hipMemcpy(dx, hx.data(), hx.size(), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(dy, hy.data(), hy.size(), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(dxm, x_mags, (size_t)nx * 4, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(dym, y_mags, (size_t)ny * 4, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(dpx, pair_x, (size_t)n_pairs * 4, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(dpy, pair_y, (size_t)n_pairs * 4, hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        DistArgs a{dx, dy, dxm, dym, dpx, dpy, row_stride, n_pairs, dim, metric, nchunks, dout, dst};
        dim3 grid(n_pairs), block(64);
        if (eng == ENG_U8) hipLaunchKernelGGL(distance_pairs_kernel<ENG_U8>, grid, block, 0, 0, a);
        else if (eng == ENG_Q2) hipLaunchKernelGGL(distance_pairs_kernel<ENG_Q2>, grid, block, 0, 0, a);
        else hipLaunchKernelGGL(distance_pairs_kernel<ENG_F32>, grid, block, (size_t)row_stride, 0, a);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpy(out, dout, (size_t)n_pairs * 4, hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(status, dst, (size_t)n_pairs * 4, hipMemcpyDeviceToHost);
    void *ptrs[] = {dx, dy, dxm, dym, dpx, dpy, dout, dst};
    for (void *p : ptrs) if (p) (void)hipFree(p);
    HIP_TRY(e);
    return COS_OK;
}

// ================================================================================================
// cos_sample_values_range — "auto" quantization range sampling (indexes/hnsw/mod.rs:202-351)
// ================================================================================================
namespace {
__global__ void sample_counts_kernel(const float *__restrict__ x, u64 total, unsigned long long *__restrict__ counts /*[14]*/) {
    // thresholds of sample_embedding: value > t for t in {.025,.05,.1,.2,.3,.4,.5} and value < -t
    const float T[7] = {0.025f, 0.05f, 0.1f, 0.2f, 0.3f, 0.4f, 0.5f};
    u32 c[14];
#pragma unroll
    for (int i = 0; i < 14; i++) c[i] = 0;
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (u64)gridDim.x * blockDim.x) {
        const float v = x[i];
#pragma unroll
        for (int t = 0; t < 7; t++) { c[t] += v > T[t]; c[7 + t] += v < -T[t]; }
    }
#pragma unroll
    for (int i = 0; i < 14; i++) {
        u32 s = c[i];
#pragma unroll
        for (int m = 32; m > 0; m >>= 1) s += (u32)__shfl_xor((int)s, m, 64);
        if ((threadIdx.x & 63) == 0 && s) atomicAdd(&counts[i], (unsigned long long)s);
    }
}
} // namespace

extern "C" int32_t cos_sample_values_range(const float *x, uint32_t n, uint32_t dim, float clamp_margin_percent, float *range_lo, float *range_hi) {
    if (!x || !range_lo || !range_hi || n == 0 || dim == 0) return cos_fail(COS_ERR_INVALID, "bad argument");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return cos_fail(COS_ERR_NO_DEVICE, "no HIP device visible; the GPU path has no CPU fallback");
    const u64 total = (u64)n * dim;
    float *d_x = nullptr;
    unsigned long long *d_c = nullptr, h_c[14];
    hipError_t e = hipMalloc(&d_x, total * 4);
    if (e == hipSuccess) e = hipMalloc(&d_c, 14 * 8);
    if (e == hipSuccess) e = hipMemcpy(d_x, x, total * 4, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemset(d_c, 0, 14 * 8);
    if (e == hipSuccess) {
        const u32 blocks = (u32)std::min<u64>(2048, (total + 255) / 256);
        hipLaunchKernelGGL(sample_counts_kernel, dim3(blocks), dim3(256), 0, 0, d_x, total, d_c);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpy(h_c, d_c, 14 * 8, hipMemcpyDeviceToHost);
    if (d_x) (void)hipFree(d_x);
    if (d_c) (void)hipFree(d_c);
    HIP_TRY(e);
    // finalize_sampling: first threshold whose tail holds <= clamp_margin_percent of all values, else +-1.0
    const float values_count = (float)total; // (dimension * embeddings.len()) as f32
    const float T[7] = {0.025f, 0.05f, 0.1f, 0.2f, 0.3f, 0.4f, 0.5f};
    float hi = 1.0f, lo = -1.0f;
    for (int t = 0; t < 7; t++)
        if (((float)h_c[t] / values_count) * 100.0f <= clamp_margin_percent) { hi = T[t]; break; }
    for (int t = 0; t < 7; t++)
        if (((float)h_c[7 + t] / values_count) * 100.0f <= clamp_margin_percent) { lo = -T[t]; break; }
    *range_lo = lo;
    *range_hi = hi;
    return COS_OK;
}

// ================================================================================================
// Remaining Storage kinds of the operators (reference layouts, no device re-layout): HalfPrecisionFP and
// SubByte with any resolution 1..3 — quantize (scalar.rs:29-42, common.rs:226-275) and
// DistanceMetric::calculate for all four metrics with the reference's error arms.
// ================================================================================================
#include <hip/hip_fp16.h>

namespace {

// one wave per row; writes the REFERENCE layout directly
__global__ __launch_bounds__(256) void quantize_ref_kernel(const float *__restrict__ x, u32 n, u32 dim, u32 storage, u32 res,
                                                           uint8_t *__restrict__ codes, u64 cb, float *__restrict__ mags) {
    const int lane = threadIdx.x & 63;
    const u32 row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= n) return;
    const float *xr = x + (u64)row * dim;
    uint8_t *cr = codes + (u64)row * cb;
    {
        const float rn = seq_norm_wave(xr, dim, lane); // norm of the ORIGINAL vector for both kinds
        if (lane == 0) mags[row] = rn;
    }
    if (storage == COS_STORAGE_F16) {
        __half *h = (__half *)cr;
        for (u32 i = lane; i < dim; i += 64) h[i] = __float2half_rn(xr[i]); // half::f16::from_f32 (RNE)
        return;
    }
    const u32 pb = (dim + 7) / 8;
    const float step = 2.0f / (float)(1u << res);
    for (u32 c = 0; c * 64 < dim; c++) {
        const u32 i = c * 64 + lane;
        u32 lvl = 0;
        if (i < dim) {
            const float f = floorf(__fdiv_rn(__fadd_rn(xr[i], 1.0f), step));
            // `as usize` saturates (NaN -> 0); only the low `res` bits reach the planes
            if (f == f && f > 0.0f) lvl = f >= 18446744073709551616.0f ? 0xFFFFFFFFu : (u32)((u64)f & 0xFFull);
        }
        for (u32 p = 0; p < res; p++) {
            const u64 m = __ballot((lvl >> (res - 1 - p)) & 1u); // plane 0 = MSB
            if (lane < 8 && c * 8 + lane < pb) cr[(u64)p * pb + c * 8 + lane] = (uint8_t)(m >> (8 * lane));
        }
    }
}

struct DistRefArgs {
    const uint8_t *x_codes, *y_codes;
    const float *x_mags, *y_mags;
    const u32 *pair_x, *pair_y;
    u64 cb;
    u32 n_pairs, dim, metric, storage, res;
    float *out;
    int32_t *status;
};

// one wave per pair, reference layouts
__global__ __launch_bounds__(64) void distance_ref_kernel(const DistRefArgs a) {
    const int lane = threadIdx.x;
    const u32 p = blockIdx.x;
    if (p >= a.n_pairs) return;
    const uint8_t *xr = a.x_codes + (u64)a.pair_x[p] * a.cb, *yr = a.y_codes + (u64)a.pair_y[p] * a.cb;
    const float xm = a.x_mags[a.pair_x[p]], ym = a.y_mags[a.pair_y[p]];
    float val = 0.0f;
    int32_t st = COS_OK;
    const bool want_dot = a.metric == COS_METRIC_COSINE || a.metric == COS_METRIC_DOT;
    if (a.storage == COS_STORAGE_F16) {
        const __half *xh = (const __half *)xr, *yh = (const __half *)yr;
        if (a.metric == COS_METRIC_HAMMING) { // hamming.rs:100-115: popcount of the bit patterns, exact in any order
            u32 acc = 0;
            for (u32 i = lane; i < a.dim; i += 64) acc += __popc((u32)(__half_as_ushort(xh[i]) ^ __half_as_ushort(yh[i])));
#pragma unroll
            for (int m = 32; m > 0; m >>= 1) acc += (u32)__shfl_xor((int)acc, m, 64);
            val = (float)acc;
        } else { // sequential f32 chains (dot_product.rs:13-19, euclidean.rs:55-66): one lane
            float acc = -0.0f;
            if (lane == 0) {
                if (want_dot)
                    for (u32 i = 0; i < a.dim; i++) acc = __fadd_rn(acc, __fmul_rn(__half2float(xh[i]), __half2float(yh[i])));
                else
                    for (u32 i = 0; i < a.dim; i++) {
                        const float d = __fsub_rn(__half2float(xh[i]), __half2float(yh[i]));
                        acc = __fadd_rn(acc, __fmul_rn(d, d));
                    }
            }
            acc = __uint_as_float(readlane_u32(__float_as_uint(acc), 0));
            if (a.metric == COS_METRIC_EUCLIDEAN) val = sqrtf(acc);
            else if (a.metric == COS_METRIC_DOT) val = acc;
            else {
                const float den = __fmul_rn(xm, ym);
                if (den == 0.0f) st = COS_ERR_CALCULATION; else val = __fdiv_rn(acc, den);
            }
        }
    } else { // SubByte, plane-major; plane index as multiplied by the reference: x_vec[0] = least significant
        const u32 pb = (a.dim + 7) / 8, res = a.res;
        if (a.metric == COS_METRIC_EUCLIDEAN) st = COS_ERR_UNIMPLEMENTED; // euclidean.rs:34-37
        else if (a.metric == COS_METRIC_HAMMING) { // hamming.rs:73-98: only (8/res)*res low bits of every byte take part
            if (res == 0 || res > 8) val = INFINITY;
            else {
                const u32 mask = (1u << ((8 / res) * res)) - 1u;
                u32 acc = 0;
                for (u32 i = lane; i < res * pb; i += 64) acc += __popc((u32)(xr[i] ^ yr[i]) & mask);
#pragma unroll
                for (int m = 32; m > 0; m >>= 1) acc += (u32)__shfl_xor((int)acc, m, 64);
                val = (float)acc;
            }
        } else if (res < 1 || res > 3) st = COS_ERR_CALCULATION; // cosine.rs:151-153
        else {
            // sum over dims of x*y with x = sum_a 2^a x_a: sum_{a,b} 2^(a+b) popcount(plane_a(x) & plane_b(y))
            // (res 2: identical to msbs*4 + carry*4 + mid*2 + lsbs of dot_product_quaternary; res 3: the octal LUT)
            u32 acc = 0;
            for (u32 i = lane; i < pb; i += 64)
                for (u32 pa = 0; pa < res; pa++)
                    for (u32 pbb = 0; pbb < res; pbb++) acc += (u32)__popc((u32)(xr[(u64)pa * pb + i] & yr[(u64)pbb * pb + i])) << (pa + pbb);
#pragma unroll
            for (int m = 32; m > 0; m >>= 1) acc += (u32)__shfl_xor((int)acc, m, 64);
            const float dotf = (float)acc;
            if (a.metric == COS_METRIC_DOT) val = dotf;
            else {
                const float den = __fmul_rn(xm, ym);
                if (den == 0.0f) st = COS_ERR_CALCULATION; else val = __fdiv_rn(dotf, den);
            }
        }
    }
    if (lane == 0) { a.out[p] = st == COS_OK ? val : 0.0f; a.status[p] = st; }
}

} // namespace

namespace cosdev {
// reference-layout operators for the kinds that have no device index layout
int32_t quantize_ref_layout(uint32_t storage, uint32_t res, uint32_t dim, const float *x, uint32_t n, void *codes, float *mags) {
    const size_t cb = cos_code_bytes(storage, res, dim);
    float *d_x = nullptr, *d_m = nullptr;
    uint8_t *d_c = nullptr;
    hipError_t e = hipMalloc(&d_x, (size_t)n * dim * 4);
    if (e == hipSuccess) e = hipMalloc(&d_m, (size_t)n * 4);
    if (e == hipSuccess) e = hipMalloc(&d_c, (size_t)n * cb);
    if (e == hipSuccess) e = hipMemcpy(d_x, x, (size_t)n * dim * 4, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemset(d_c, 0, (size_t)n * cb);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(quantize_ref_kernel, dim3((n + 3) / 4), dim3(256), 0, 0, d_x, n, dim, storage, res, d_c, (u64)cb, d_m);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpy(codes, d_c, (size_t)n * cb, hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(mags, d_m, (size_t)n * 4, hipMemcpyDeviceToHost);
    (void)hipFree(d_x); (void)hipFree(d_m); (void)hipFree(d_c);
    HIP_TRY(e);
    return COS_OK;
}

int32_t distance_ref_layout(uint32_t metric, uint32_t storage, uint32_t res, uint32_t dim, const void *x_codes, const float *x_mags, uint32_t nx,
                            const void *y_codes, const float *y_mags, uint32_t ny, const uint32_t *pair_x, const uint32_t *pair_y, uint32_t n_pairs,
                            float *out, int32_t *status) {
    const size_t cb = cos_code_bytes(storage, res, dim);
    uint8_t *dx = nullptr, *dy = nullptr;
    float *dxm = nullptr, *dym = nullptr, *dout = nullptr;
    u32 *dpx = nullptr, *dpy = nullptr;
    int32_t *dst = nullptr;
    hipError_t e = hipMalloc(&dx, std::max<size_t>((size_t)nx * cb, 16));
    if (e == hipSuccess) e = hipMalloc(&dy, std::max<size_t>((size_t)ny * cb, 16));
    if (e == hipSuccess) e = hipMalloc(&dxm, (size_t)nx * 4);
    if (e == hipSuccess) e = hipMalloc(&dym, (size_t)ny * 4);
    if (e == hipSuccess) e = hipMalloc(&dpx, (size_t)n_pairs * 4);
    if (e == hipSuccess) e = hipMalloc(&dpy, (size_t)n_pairs * 4);
    if (e == hipSuccess) e = hipMalloc(&dout, (size_t)n_pairs * 4);
    if (e == hipSuccess) e = hipMalloc(&dst, (size_t)n_pairs * 4);
    if (e == hipSuccess) e = hipMemcpy(dx, x_codes, (size_t)nx * cb, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(dy, y_codes, (size_t)ny * cb, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(dxm, x_mags, (size_t)nx * 4, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(dym, y_mags, (size_t)ny * 4, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(dpx, pair_x, (size_t)n_pairs * 4, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(dpy, pair_y, (size_t)n_pairs * 4, hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        DistRefArgs a{dx, dy, dxm, dym, dpx, dpy, (u64)cb, n_pairs, dim, metric, storage, res, dout, dst};
        hipLaunchKernelGGL(distance_ref_kernel, dim3(n_pairs), dim3(64), 0, 0, a);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpy(out, dout, (size_t)n_pairs * 4, hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(status, dst, (size_t)n_pairs * 4, hipMemcpyDeviceToHost);
    void *ptrs[] = {dx, dy, dxm, dym, dpx, dpy, dout, dst};
    for (void *p : ptrs) if (p) (void)hipFree(p);
    HIP_TRY(e);
    return COS_OK;
}
} // namespace cosdev
