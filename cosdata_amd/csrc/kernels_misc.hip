// kernels_misc.hip — small gfx950 kernels around the walk: S-way top-k merge for the sharded index
//  and the src/distance operator on explicit pairs (cos_distance_batch).
#include <hip/hip_runtime.h>

#include <vector>

#include "engine_internal.h"

using namespace cosdev;

namespace {

// One wave per query: gather S per-shard lists (already sorted, but any order is accepted), sort by
// (total_cmp score desc, larger id first) and emit the best k.
template <int R>
__global__ __launch_bounds__(64) void merge_topk_kernel(const u32 *__restrict__ ids, const float *__restrict__ scores,
                                                        const u32 *__restrict__ counts, u32 S, u32 B, u32 k, u32 *__restrict__ out_ids,
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
            if (j < counts[(u64)s * B + q]) {
                const u64 off = ((u64)s * B + q) * k + j;
                key[r] = pack_key(simkey(scores[off]), ids[off]);
            }
        }
    }
    for (u32 s = 0; s < S; s++) total += counts[(u64)s * B + q];
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

} // namespace

extern "C" int32_t cos_merge_topk_device(const uint32_t *d_ids, const float *d_scores, const uint32_t *d_counts, uint32_t S, uint32_t B,
                                         uint32_t k, uint32_t *d_out_ids, float *d_out_scores, uint32_t *d_out_counts, int32_t device,
                                         void *stream) {
    if (!d_ids || !d_scores || !d_counts || !d_out_ids || !d_out_scores || !d_out_counts || S == 0 || B == 0 || k == 0)
        return cos_fail(COS_ERR_INVALID, "bad argument");
    HIP_TRY(hipSetDevice(device));
    const u32 total = S * k;
    hipStream_t st = (hipStream_t)stream;
    dim3 grid(B), block(64);
#define LAUNCH(R) hipLaunchKernelGGL(merge_topk_kernel<R>, grid, block, 0, st, d_ids, d_scores, d_counts, S, B, k, d_out_ids, d_out_scores, d_out_counts)
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
