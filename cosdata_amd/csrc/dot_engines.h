// dot_engines.h — device-side arithmetic shared by the gfx950 kernels: Rust cast semantics, the
// sequential norm, the integer chunk dots and the reference-order f32 dot.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include "device_common.h"

namespace cosdev {

// ------------------------------------------------------------------------------------------------
// Rust `as` casts and f32::max/min
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint8_t rust_f32_as_u8(float v) {
    if (!(v == v)) return 0;
    if (v <= 0.0f) return 0;
    if (v >= 255.0f) return 255;
    return (uint8_t)(int)v;
}
__device__ __forceinline__ u32 rust_f32_as_usize_low2(float v) { // low 2 bits of `v as usize` (saturating, NaN -> 0)
    if (!(v == v)) return 0;
    if (v <= 0.0f) return 0;
    if (v >= 18446744073709551616.0f) return 3u;
    return (u32)((u64)v & 3ull);
}
__device__ __forceinline__ u32 rust_f32_as_usize_low(float v, u32 mask) { // low bits of `v as usize` (saturating, NaN -> 0)
    if (!(v == v)) return 0;
    if (v <= 0.0f) return 0;
    if (v >= 18446744073709551616.0f) return mask;
    return (u32)((u64)v & (u64)mask);
}
__device__ __forceinline__ float rust_max(float a, float b) { return (a != a) ? b : ((b != b) ? a : (a > b ? a : b)); }
__device__ __forceinline__ float rust_min(float a, float b) { return (a != a) ? b : ((b != b) ? a : (a < b ? a : b)); }

// IEEE division with x86's NaN convention.  The reference runs on x86-64 (its SIMD paths are AVX2), where an invalid
// operation (0/0, inf/inf) yields the NEGATIVE default NaN 0xFFC00000; gfx950 yields 0x7FC00000.  f32::total_cmp puts the
// two at opposite ends of the order (a zero raw vector would rank first instead of last in finalize_ann_results), so the
// quotient of two non-NaN operands is canonicalised to the x86 pattern.
__device__ __forceinline__ float x86_div(float a, float b) {
    float r = __fdiv_rn(a, b);
    if (r != r && a == a && b == b) r = __uint_as_float(0xFFC00000u);
    return r;
}

// sqrt(sequential, non-fused sum of x*x): scalar.rs:31,41,45 / vector_store.rs:414,426.  Executed by ONE lane.
__device__ __forceinline__ float seq_norm(const float *x, u32 n) {
    float acc = -0.0f;
    for (u32 i = 0; i < n; i++) acc = __fadd_rn(acc, __fmul_rn(x[i], x[i]));
    return sqrtf(acc);
}

// The same value computed by a whole wave (every lane returns it): the squares are formed 64 at a time from one coalesced load,
// the sequential chain of additions — the part the reference's order fixes — reads them back lane by lane (v_readlane) in index
// order.  One lane doing everything issued `n` dependent scalar loads per row: 150 us per 768-dim row, which made the
// quantization of a 256-query batch 10 % of its latency and the upload of 10M rows 0.2 s.
__device__ __forceinline__ float seq_norm_wave(const float *__restrict__ x, u32 n, int lane) {
    float acc = -0.0f;
    for (u32 base = 0; base < n; base += 64) {
        const u32 i = base + (u32)lane;
        const float v = i < n ? x[i] : 0.0f;
        const u32 sq = __float_as_uint(__fmul_rn(v, v));
        if (n - base >= 64) {
#pragma unroll
            for (int j = 0; j < 64; j++) acc = __fadd_rn(acc, __uint_as_float((u32)__builtin_amdgcn_readlane((int)sq, j)));
        } else {
            for (u32 j = 0; j < n - base; j++) acc = __fadd_rn(acc, __uint_as_float((u32)__builtin_amdgcn_readlane((int)sq, (int)j)));
        }
    }
    return sqrtf(acc);
}

// ------------------------------------------------------------------------------------------------
// integer dot engines: G lanes cooperate on one code row, 16 bytes per lane per chunk
// ------------------------------------------------------------------------------------------------
template <int ENG>
__device__ __forceinline__ u32 chunk_dot(uint4 q, uint4 y, u32 acc);

template <>
__device__ __forceinline__ u32 chunk_dot<ENG_U8>(uint4 q, uint4 y, u32 acc) { // dot_product_u8 (x86_64.rs:22-66): exact integer
    acc = __builtin_amdgcn_udot4(q.x, y.x, acc, false);
    acc = __builtin_amdgcn_udot4(q.y, y.y, acc, false);
    acc = __builtin_amdgcn_udot4(q.z, y.z, acc, false);
    acc = __builtin_amdgcn_udot4(q.w, y.w, acc, false);
    return acc;
}
template <>
__device__ __forceinline__ u32 chunk_dot<ENG_Q2>(uint4 q, uint4 y, u32 acc) { // dot_product_quaternary (dot_product.rs:35-57)
    // plane 0 is what the reference multiplies as "lsb", plane 1 as "msb"
    u64 xl = ((u64)q.y << 32) | q.x, xm = ((u64)q.w << 32) | q.z;
    u64 yl = ((u64)y.y << 32) | y.x, ym = ((u64)y.w << 32) | y.z;
    u64 mid1 = xl & ym, mid2 = yl & xm;
    u32 lsbs = (u32)__popcll(xl & yl), carry = (u32)__popcll(mid1 & mid2);
    u32 msbs = (u32)__popcll(xm & ym), mid = (u32)__popcll(mid1 ^ mid2);
    return acc + (msbs << 2) + (carry << 2) + (mid << 1) + lsbs;
}

template <>
__device__ __forceinline__ u32 chunk_dot<ENG_Q1>(uint4 q, uint4 y, u32 acc) { // dot_product_binary (dot_product.rs:21-33): 128 dims per chunk
    return acc + (u32)__popc(q.x & y.x) + (u32)__popc(q.y & y.y) + (u32)__popc(q.z & y.z) + (u32)__popc(q.w & y.w);
}
template <>
__device__ __forceinline__ u32 chunk_dot<ENG_Q3>(uint4 q, uint4 y, u32 acc) { // dot_product_octal (dot_product.rs:64-90): 32 dims per chunk
    // chunk = [plane0 | plane1 | plane2 | 0] of the same 32 dims; the reference's digit is (p2<<2)|(p1<<1)|p0 on the planes
    // AS STORED (plane 0 is the level's MSB, multiplied as the LSB — same quirk as quaternary):
    // sum_dims x*y = sum_{i,j} 2^(i+j) popcount(x_i & y_j)
    const u32 x0 = q.x, x1 = q.y, x2 = q.z, y0 = y.x, y1 = y.y, y2 = y.z;
    u32 s = (u32)__popc(x0 & y0);
    s += ((u32)__popc(x0 & y1) + (u32)__popc(x1 & y0)) << 1;
    s += ((u32)__popc(x0 & y2) + (u32)__popc(x1 & y1) + (u32)__popc(x2 & y0)) << 2;
    s += ((u32)__popc(x1 & y2) + (u32)__popc(x2 & y1)) << 3;
    s += (u32)__popc(x2 & y2) << 4;
    return acc + s;
}

// reference-order f32 dot (dot_product_f32_simd, x86_64.rs:418-444) by a PAIR of lanes:
// even lane owns accumulators 0..3, odd lane 4..7; returns the full dot in both lanes.
__device__ __forceinline__ float f32_pair_dot(const float *__restrict__ row, const float *__restrict__ q_lds, u32 dim, int half) {
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    const u32 chunks = dim >> 3;
    const float4 *rp = (const float4 *)row + half;
    const float4 *qp = (const float4 *)q_lds + half;
    u32 c = 0;
    for (; c + 8 <= chunks; c += 8) {
        float4 y[8];
#pragma unroll
        for (int u = 0; u < 8; u++) y[u] = rp[2 * (c + u)];
#pragma unroll
        for (int u = 0; u < 8; u++) {
            float4 qv = qp[2 * (c + u)];
            a0 = __fmaf_rn(qv.x, y[u].x, a0);
            a1 = __fmaf_rn(qv.y, y[u].y, a1);
            a2 = __fmaf_rn(qv.z, y[u].z, a2);
            a3 = __fmaf_rn(qv.w, y[u].w, a3);
        }
    }
    for (; c < chunks; c++) {
        float4 yv = rp[2 * c], qv = qp[2 * c];
        a0 = __fmaf_rn(qv.x, yv.x, a0);
        a1 = __fmaf_rn(qv.y, yv.y, a1);
        a2 = __fmaf_rn(qv.z, yv.z, a2);
        a3 = __fmaf_rn(qv.w, yv.w, a3);
    }
    float t = __fadd_rn(__fadd_rn(a0, a1), __fadd_rn(a2, a3)); // (s0+s1)+(s2+s3) | (s4+s5)+(s6+s7)
    float r = __fadd_rn(t, __shfl_xor(t, 1, 64));               // low half + high half
    for (u32 i = chunks * 8; i < dim; i++) r = __fadd_rn(r, __fmul_rn(q_lds[i], row[i])); // scalar tail, non-fused
    return r;
}



// The same reference-order f32 dot by EIGHT lanes, lane c owning accumulator (chain) c: element 8s + c of the row at step s.
// A walk expansion discovers ~7 new neighbours: with a lane PAIR per row 25 of the 32 pairs of a wave idle while each busy lane
// streams 1.5 KB with 8 loads in flight — the bytes in flight per CU, not the HBM, bounded the f32-storage walk at 0.44 of the
// roof.  With 8 lanes per row a pass evaluates 8 rows, every lane streams 1/8 of a row with 32 dword loads in flight, and the
// per-lane FMA count drops 4x.  Same chains, same order inside each chain, same pairwise tree — ((s0+s1)+(s2+s3)) +
// ((s4+s5)+(s6+s7)), IEEE addition being commutative — so the result is the pair version's bit for bit.  Returns the full dot
// in all 8 lanes of the group (lanes 8g .. 8g+7).
__device__ __forceinline__ float f32_oct_dot(const float *__restrict__ row, const float *__restrict__ q_lds, u32 dim, int c) {
    float a = 0.f;
    const u32 chunks = dim >> 3;
    const float *rp = row + c;
    const float *qp = q_lds + c;
    u32 s = 0;
    for (; s + 32 <= chunks; s += 32) {
        float y[32];
#pragma unroll
        for (int u = 0; u < 32; u++) y[u] = rp[8 * (s + u)];
#pragma unroll
        for (int u = 0; u < 32; u++) a = __fmaf_rn(qp[8 * (s + u)], y[u], a);
    }
    for (; s + 8 <= chunks; s += 8) {
        float y[8];
#pragma unroll
        for (int u = 0; u < 8; u++) y[u] = rp[8 * (s + u)];
#pragma unroll
        for (int u = 0; u < 8; u++) a = __fmaf_rn(qp[8 * (s + u)], y[u], a);
    }
    for (; s < chunks; s++) a = __fmaf_rn(qp[8 * s], rp[8 * s], a);
    float t = __fadd_rn(a, __shfl_xor(a, 1, 64)); // s0+s1 | s2+s3 | s4+s5 | s6+s7
    t = __fadd_rn(t, __shfl_xor(t, 2, 64));       // (s0+s1)+(s2+s3) | (s4+s5)+(s6+s7)
    float r = __fadd_rn(t, __shfl_xor(t, 4, 64)); // low half + high half
    for (u32 i = chunks * 8; i < dim; i++) r = __fadd_rn(r, __fmul_rn(q_lds[i], row[i])); // scalar tail, non-fused
    return r;
}

// dot_product_f16 (dot_product.rs:13-19): sequential sum of f32(a) * f32(b), no SIMD in the reference.  ONE lane
// walks one stored row; the query's f16 code is held as f32 in LDS (f32::from(b) is exact).
__device__ __forceinline__ float f16_lane_dot(const uint8_t *__restrict__ row, const float *__restrict__ q_lds, u32 dim) {
    float acc = -0.0f; // <f32 as Sum> folds from -0.0
    u32 i = 0;
    for (; i + 8 <= dim; i += 8) {
        const uint4 v = *(const uint4 *)(row + (u64)i * 2);
        const u32 w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int k = 0; k < 4; k++) {
            acc = __fadd_rn(acc, __fmul_rn(__half2float(__ushort_as_half((unsigned short)(w[k] & 0xFFFFu))), q_lds[i + 2 * k]));
            acc = __fadd_rn(acc, __fmul_rn(__half2float(__ushort_as_half((unsigned short)(w[k] >> 16))), q_lds[i + 2 * k + 1]));
        }
    }
    for (; i < dim; i++) acc = __fadd_rn(acc, __fmul_rn(__half2float(((const __half *)row)[i]), q_lds[i]));
    return acc;
}

} // namespace cosdev
