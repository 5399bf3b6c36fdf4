// wave_reduce_check.hip — device check of cosdev::wave_reduce_rows<4|8> (cosdata_amd/csrc/device_common.h) against host sums.
// Test infrastructure: compiled and run by tests/test_gpu_wave_reduce.py on the GPU box (hipcc --offload-arch=gfx950).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include "device_common.h"

using cosdev::u32;

template <int NP>
__global__ __launch_bounds__(64) void reduce_rows_kernel(const u32 *in, u32 *out) { // in [blocks][NP][64], out [blocks][64]
    u32 a[NP];
#pragma unroll
    for (int p = 0; p < NP; p++) a[p] = in[((size_t)blockIdx.x * NP + p) * 64 + threadIdx.x];
    out[(size_t)blockIdx.x * 64 + threadIdx.x] = cosdev::wave_reduce_rows<NP>(a);
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)

template <int NP>
static int run(int blocks) {
    const size_t n_in = (size_t)blocks * NP * 64, n_out = (size_t)blocks * 64;
    u32 *h_in = (u32 *)malloc(n_in * 4), *h_out = (u32 *)malloc(n_out * 4);
    uint64_t s = 0x9E3779B97F4A7C15ull + NP;
    for (size_t i = 0; i < n_in; i++) { s = s * 6364136223846793005ull + 1442695040888963407ull; h_in[i] = (u32)(s >> 33); } // sums wrap: u32 arithmetic
    u32 *d_in, *d_out;
    CK(hipMalloc(&d_in, n_in * 4)); CK(hipMalloc(&d_out, n_out * 4));
    CK(hipMemcpy(d_in, h_in, n_in * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(reduce_rows_kernel<NP>, dim3(blocks), dim3(64), 0, 0, d_in, d_out);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(h_out, d_out, n_out * 4, hipMemcpyDeviceToHost));
    int bad = 0;
    for (int b = 0; b < blocks; b++)
        for (int p = 0; p < NP; p++) {
            u32 want = 0;
            for (int l = 0; l < 64; l++) want += h_in[((size_t)b * NP + p) * 64 + l];
            for (int l = 0; l < 64 / NP; l++) {
                const u32 got = h_out[(size_t)b * 64 + p * (64 / NP) + l];
                if (got != want && bad++ < 8) fprintf(stderr, "NP %d block %d row %d lane %d: got %u want %u\n", NP, b, p, p * (64 / NP) + l, got, want);
            }
        }
    (void)hipFree(d_in); (void)hipFree(d_out); free(h_in); free(h_out);
    return bad ? 1 : 0;
}

int main() {
    int rc = run<8>(257);
    rc |= run<4>(257);
    printf(rc ? "MISMATCH\n" : "OK wave_reduce_rows<8>, <4>: every lane of every group holds its row's sum\n");
    return rc;
}
