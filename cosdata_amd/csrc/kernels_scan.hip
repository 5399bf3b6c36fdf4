// Query-resident scan of quaternary codes: its own translation unit because it is compiled with
// -mllvm -amdgpu-mfma-vgpr-form=1 (MFMA accumulators in VGPRs, query fragments usable straight from AccVGPRs; see areg_mfma).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>
#include <type_traits>

#include "device_common.h"
#include "flat_scan.h"
#include "tuning.h"

using namespace cosdev;

namespace {

typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x16 __attribute__((ext_vector_type(16)));

// ------------------------------------------------------------------------------------------------
// Query-resident scan (quaternary codes, fused epilogue).  Counters on flat_codes_gemm_i8 at 10M x 768 (profiles/
// r02_c3_i8_scan_sq_lds_counters.txt): VALU 49 % busy, LDS 42 %, MFMA 26 % — every 256 x 128 tile restages the same 196 KB
// of query digits through LDS, and the per-element epilogue costs ten VALU ops.  Here the query operand never moves:
//   * a workgroup is 4 waves, one per SIMD; wave w keeps the MFMA A fragments of query rows 64w .. 64w+63 for the WHOLE k
//     range in AccVGPRs (K / 32 fragments x 2 row blocks x 4 = 192 registers at K = 768) for the life of the kernel.  A lone
//     wave per SIMD issues in order, so whatever should run in an MFMA's 32-cycle shadow has to sit right behind it in the
//     instruction stream: the loop body is written as (MFMA, <= 6 VALU) pairs and pinned that way.  (An 8-wave variant, two
//     waves of 32 rows per SIMD, was measured at 2.98 ms against 1.72 ms for this one on 10M x 768: the compiler splits a
//     256-register wave 128 VGPR / 128 AccVGPR, the VGPR half spills, and every candidate fragment is read from LDS twice.);
//   * candidates stream through LDS in tiles of 64 rows x K digit bytes, double-buffered: while the MFMAs of tile t run,
//     the same waves expand tile t+1 (planes -> digits, 4 VALU ops per dword thanks to a dim permutation inside each 64-dim
//     chunk that the query digits share) and the raw loads of tiles t+2 and t+3 are in flight.  One barrier per tile;
//   * LDS rows are K + 16 bytes, so the 16 lanes of a ds_read_b128 group (distinct rows mod 16) hit 16 distinct 16 B slots,
//     and the staging stores (lane = candidate) are conflict-free too;
//   * two accumulator sets: the epilogue of tile t runs in the MFMA shadows of tile t+1.  It is a per-lane "does any of my
//     16 rows pass" test (cvt + fma + max3 per element) against thresholds held in registers; the rare survivors are parked
//     in LDS and get their exact quotient, key compare and global append at the end of the kernel.  Same survivors as the
//     tile kernel: the estimate only decides who gets the exact test.
// One workgroup per CU, persistent over the tiles of the launch.
// ------------------------------------------------------------------------------------------------
// 16 digit bytes of piece pp (0..3) of a 64-dim chunk, permuted: byte b of dword jj is dim 32*(pp>>1) + 4*(pp&1) + jj + 8*b
__device__ __forceinline__ u32 q2_dword_perm(uint4 raw /*[plane0 8 B | plane1 8 B]*/, int pp, int jj) {
    const u32 w0 = pp < 2 ? raw.x : raw.y, w1 = pp < 2 ? raw.z : raw.w;
    const int sh = 4 * (pp & 1) + jj;
    const u32 x0 = w0 >> sh, x1 = sh ? (w1 >> (sh - 1)) : (w1 << 1);
    return (x0 & 0x01010101u) | (x1 & 0x02020202u);
}
__device__ __forceinline__ uint4 q2_piece_perm(uint4 raw, int pp) {
    return make_uint4(q2_dword_perm(raw, pp, 0), q2_dword_perm(raw, pp, 1), q2_dword_perm(raw, pp, 2), q2_dword_perm(raw, pp, 3));
}

__global__ void expand_q2_digits_perm_kernel(const uint8_t *__restrict__ qcodes, u64 row_stride, u32 B, u32 kdims, uint8_t *__restrict__ digits) {
    const u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    const u32 chunks = kdims / 64;
    if (t >= (u64)B * chunks) return;
    const u32 q = (u32)(t / chunks), j = (u32)(t % chunks);
    const uint4 raw = (u64)j * 16 < row_stride ? *(const uint4 *)(qcodes + (u64)q * row_stride + (u64)j * 16) : make_uint4(0, 0, 0, 0);
#pragma unroll
    for (int pp = 0; pp < 4; pp++) *(uint4 *)(digits + (u64)q * kdims + (u64)j * 64 + pp * 16) = q2_piece_perm(raw, pp);
}

// The MFMA of the query-resident kernel.  What the kernel needs from the register allocator: query fragments resident in
// AccVGPRs and used from there, accumulators in VGPRs (the epilogue reads them with VALU ops).  With the default "AGPR form" of
// the builtin the allocator parked the fragments in AccVGPRs but copied each back (4 x v_accvgpr_read) before every use and kept
// the accumulators in AccVGPRs (64 more copies per tile).  Writing the instruction as inline asm with explicit register classes
// gives the right operands but hides the instruction from the hazard recognizer — and this one needs it: results read or spilled
// right after issue, sources rewritten right before it, both observed as wrong dot products.  So: the builtin, in a translation
// unit compiled with -mllvm -amdgpu-mfma-vgpr-form=1, and the fragments pinned to AccVGPRs once when they are loaded.
template <bool FIRST>
__device__ __forceinline__ void areg_mfma(i32x16 &acc, const i32x4 &afrag, const i32x4 &bfrag) {
    const i32x16 z = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(afrag, bfrag, FIRST ? z : acc, 0, 0, 0);
}

constexpr int AREG_STAGE = 2048; // survivors a workgroup can park in LDS per launch (expected: a few hundred)

// exact score of one survivor of the estimate, compared with the query's threshold key, appended if it beats it
__device__ __forceinline__ void areg_append(const FusedOut &fo, const float *__restrict__ qmags, const float *__restrict__ mags, u32 metric, u32 n0,
                                            u32 col, u32 row, u32 dot) {
    const float dotf = (float)dot;
    const float sc = metric == 0u ? __fdiv_rn(dotf, __fmul_rn(qmags[row], mags[n0 + col])) : dotf;
    const u64 key = pack_key(simkey(sc), n0 + col);
    if (key > fo.thr[row]) {
        const u32 pos = atomicAdd(&fo.app_cnt[row], 1u);
        if (pos < fo.cap) fo.app[(u64)row * fo.cap + pos] = key;
    }
}

// compile-time loop: the body sees its index as an integral_constant, so register arrays stay statically indexed whatever
// the unroller's size thresholds say (a `#pragma unroll` over the 24 k steps of the query-resident kernel is refused)
template <int I, int N, class F>
__device__ __forceinline__ void static_for(F &&f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

template <int KC>
__global__ __launch_bounds__(256) void flat_scan_q2_areg(const uint8_t *__restrict__ qdig /*[B][64 KC] permuted digits*/,
                                                         const float *__restrict__ qmags, u32 B, const uint8_t *__restrict__ codes,
                                                         const float *__restrict__ mags, u64 row_stride, u32 n0, u32 n_chunk, u32 metric,
                                                         const FusedOut fo) {
    constexpr int K = KC * 64, KS = KC * 2, LDB = K + 16, ITS = (KC + 3) / 4, PIECES = ITS * 4;
    extern __shared__ __attribute__((aligned(16))) unsigned char areg_lds[]; // [2][64][LDB] | survivors [AREG_STAGE][3] u32 | count
    u32 *stage = (u32 *)(areg_lds + (size_t)2 * 64 * LDB), *stage_cnt = stage + 3 * AREG_STAGE;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, half = lane >> 5, l31 = lane & 31;
    const u32 row0 = blockIdx.y * 256 + w * 64;
    const u32 n_tiles = (n_chunk + 63) / 64, G = gridDim.x;
    u32 t = blockIdx.x;
    if (t >= n_tiles) return; // uniform
    if (tid == 0) *stage_cnt = 0; // published by the barrier after the first tile's expansion
    i32x4 a[2][KS];
    static_for<0, 2 * KS>([&](auto Ic) __attribute__((always_inline)) {
        constexpr int i = decltype(Ic)::value / KS, s = decltype(Ic)::value % KS;
        const u32 row = row0 + 32 * i + l31; // rows past B: clamped loads (no branches in the prologue), zeroed
        const i32x4 v = *(const i32x4 *)(qdig + (u64)(row < B ? row : B - 1) * K + 32 * s + 16 * half);
        a[i][s] = row < B ? v : i32x4{0, 0, 0, 0};
        asm volatile("" : "+a"(a[i][s])); // the value now IS an AccVGPR tuple: its MFMA uses need no copies
    });
    // thresholds of this lane's 32 accumulator rows, in the units of dot * (1 / |x|): T = thr_lo * |q| (cosine) or thr_lo (dot)
    float T[2][16];
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const u32 row = row0 + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * half, rc = row < B ? row : B - 1;
            const u64 k = fo.thr[rc];
            const float qm = qmags[rc];
            const float lo = k == 0ull ? -1.0f : simkey_inv((u32)(k >> 32)) * (1.0f - 4e-6f); // scores of quaternary codes are >= 0
            const float v = metric == 0u ? lo * qm : lo;
            T[i][r] = row < B ? v : __builtin_inff();
        }
    // Staging: lane = candidate of the tile, wave w expands chunks w, w+4, w+8, ...  Two raw sets: while tile t is multiplied,
    // set (PAR^1) (tile t+G) is expanded into the other LDS buffer and reloaded with tile t+3G; set PAR (tile t+2G) is in
    // flight.  Loads are unconditional (addresses clamped to the last row of the chunk: a duplicated row is never appended,
    // its columns fail the `col < n_chunk` test), so they stay in flight across the MFMAs.
    uint4 raw[2][ITS];
    auto load_raw = [&](u32 tile, uint4 *dst, int it) __attribute__((always_inline)) {
        const int j = w + 4 * it;
        const u32 c = tile * 64 + lane, cc = c < n_chunk ? c : n_chunk - 1;
        dst[it] = *(const uint4 *)(codes + (u64)(n0 + cc) * row_stride + (u64)(KC % 4 == 0 || j < KC ? j : 0) * 16);
    };
    auto store_piece = [&](int buf, const uint4 *src, int it, int pp) __attribute__((always_inline)) {
        const int j = w + 4 * it;
        if (KC % 4 == 0 || j < KC) *(uint4 *)(areg_lds + (size_t)buf * 64 * LDB + lane * LDB + j * 64 + pp * 16) = q2_piece_perm(src[it], pp);
    };
    float xm[2][2]; // candidate norms of the tile in accumulator set 0 / 1 (consumed by the deferred epilogue)
    auto load_norms = [&](u32 tile, float *dst) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < 2; j++) {
            const u32 col = tile * 64 + 32 * j + l31;
            dst[j] = mags[n0 + (col < n_chunk ? col : n_chunk - 1)];
        }
    };
#pragma unroll
    for (int it = 0; it < ITS; it++) load_raw(t, raw[0], it);
#pragma unroll
    for (int it = 0; it < ITS; it++)
#pragma unroll
        for (int pp = 0; pp < 4; pp++) store_piece(0, raw[0], it, pp);
#pragma unroll
    for (int it = 0; it < ITS; it++) {
        load_raw(t + G, raw[1], it);
        load_raw(t + 2 * G, raw[0], it);
    }
    xm[1][0] = xm[1][1] = 1.0f;
    __syncthreads();
    i32x16 acc[2][2][2]; // [set][row block][column block]
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[1][i][j][r] = 0;

    // deferred epilogue of accumulator set P (tile tp), in 8 slices of 8 rows so that each slice fits in the shadow of one k step
    // of the next tile: slice e -> column block j = e >> 2, row block i = (e >> 1) & 1, rows r = 8 (e & 1) .. +7
    float best = 0.0f;
    // rows h8 + r0, h8 + r0 + 1 of slice e: two (cvt, fma, max) triples — what fits beside one MFMA
    auto epi_rows = [&](auto Pc, int e, int r0) __attribute__((always_inline)) {
        constexpr int P = decltype(Pc)::value;
        const int j = e >> 2, i = (e >> 1) & 1, h8 = (e & 1) * 8;
        const float rx = metric == 0u ? __builtin_amdgcn_rcpf(xm[P][j]) : 1.0f;
        if (h8 == 0 && r0 == 0) best = -__builtin_inff();
#pragma unroll
        for (int r = r0; r < r0 + 2; r++) best = fmaxf(best, __builtin_fmaf((float)(u32)acc[P][i][j][h8 + r], rx, -T[i][h8 + r]));
    };
    // after the 16th row of a (row block, column block): the rare lanes with a row that may beat its query's threshold park
    // the survivors in LDS for the end of the kernel — the exact quotient, the key compare and the global append counter are
    // round trips to memory that a lone wave per SIMD cannot hide
    auto epi_finish = [&](auto Pc, int e, u32 tp, bool tp_valid) __attribute__((always_inline)) {
        constexpr int P = decltype(Pc)::value;
        const int j = e >> 2, i = (e >> 1) & 1;
        if ((e & 1) == 0) return;
        const u32 col = tp * 64 + 32 * j + l31;
        if (best >= 0.0f && tp_valid && col < n_chunk) {
            const float rx = metric == 0u ? __builtin_amdgcn_rcpf(xm[P][j]) : 1.0f;
            u32 rbase = row0 + 32 * i + 4 * half;
            asm volatile("" : "+v"(rbase)); // opaque: keeps the 32 rows' addresses from being hoisted out of the tile loop
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const u32 row = rbase + (r & 3) + 8 * (r >> 2);
                const float dotf = (float)(u32)acc[P][i][j][r];
                if (row < B && __builtin_fmaf(dotf, rx, -T[i][r]) >= 0.0f) {
                    const u32 sp = atomicAdd(stage_cnt, 1u);
                    if (sp < AREG_STAGE) {
                        stage[3 * sp] = col;
                        stage[3 * sp + 1] = row;
                        stage[3 * sp + 2] = (u32)acc[P][i][j][r];
                    } else
                        areg_append(fo, qmags, mags, metric, n0, col, row, (u32)acc[P][i][j][r]); // staging full: append from here
                }
            }
        }
    };
    auto epi_slice = [&](auto Pc, int e, u32 tp, bool tp_valid) __attribute__((always_inline)) {
        epi_rows(Pc, e, 0); epi_rows(Pc, e, 2); epi_rows(Pc, e, 4); epi_rows(Pc, e, 6);
        epi_finish(Pc, e, tp, tp_valid);
    };
    // One tile: MFMAs of tile tt into set P from LDS buffer P.  A lone wave per SIMD issues in order, so whatever is to run in
    // the shadow of an MFMA has to sit right behind it in the instruction stream: every k step is four (MFMA, <= 6 VALU) pairs —
    // one dword of the expansion of tile tt+G during the first PIECES steps, two rows of the epilogue of tile tt-G (set P^1)
    // during the next 8 — pinned by sched_barriers.
    constexpr int EPI0 = PIECES + 8 <= KS ? PIECES : (KS >= 8 ? KS - 8 : 0); // first step that carries an epilogue slice
    auto tile_body = [&](auto Pc, u32 tt, bool prev_valid) __attribute__((always_inline)) {
        constexpr int P = decltype(Pc)::value;
        const unsigned char *bt = areg_lds + (size_t)P * 64 * LDB + l31 * LDB + 16 * half;
        i32x4 bf[3][2]; // candidate fragments, read two k steps ahead of their MFMAs
#pragma unroll
        for (int s = 0; s < 2 && s < KS; s++) {
            bf[s][0] = *(const i32x4 *)(bt + 32 * s);
            bf[s][1] = *(const i32x4 *)(bt + 32 * LDB + 32 * s);
        }
        load_norms(tt, xm[P]);
        static_for<0, KS>([&](auto Sc) __attribute__((always_inline)) {
            constexpr int s = decltype(Sc)::value;
            constexpr bool expand = s < PIECES, epi = s >= EPI0 && s < EPI0 + 8;
            constexpr int it = (s < PIECES ? s : 0) >> 2, pp = s & 3, j = 0;
            if (s + 2 < KS) {
                bf[(s + 2) % 3][0] = *(const i32x4 *)(bt + 32 * (s + 2));
                bf[(s + 2) % 3][1] = *(const i32x4 *)(bt + 32 * LDB + 32 * (s + 2));
            }
            __builtin_amdgcn_sched_barrier(0);
            const i32x4 b0 = bf[s % 3][0], b1 = bf[s % 3][1];
            const bool lane_stores = KC % 4 == 0 || w + 4 * it < KC;
            u32 o[4] = {0, 0, 0, 0};
            areg_mfma<s == 0>(acc[P][0][0], a[0][s], b0);
            if (expand) { o[0] = q2_dword_perm(raw[P ^ 1][it], pp, 0); asm volatile("" : "+v"(o[0])); } // pinned behind its MFMA
            if (epi) epi_rows(std::integral_constant<int, (P ^ 1)>{}, s - EPI0, 0);
            __builtin_amdgcn_sched_barrier(0);
            areg_mfma<s == 0>(acc[P][0][1], a[0][s], b1);
            if (expand) { o[1] = q2_dword_perm(raw[P ^ 1][it], pp, 1); asm volatile("" : "+v"(o[1])); } // pinned behind its MFMA
            if (epi) epi_rows(std::integral_constant<int, (P ^ 1)>{}, s - EPI0, 2);
            __builtin_amdgcn_sched_barrier(0);
            areg_mfma<s == 0>(acc[P][1][0], a[1][s], b0);
            if (expand) { o[2] = q2_dword_perm(raw[P ^ 1][it], pp, 2); asm volatile("" : "+v"(o[2])); } // pinned behind its MFMA
            if (epi) epi_rows(std::integral_constant<int, (P ^ 1)>{}, s - EPI0, 4);
            __builtin_amdgcn_sched_barrier(0);
            areg_mfma<s == 0>(acc[P][1][1], a[1][s], b1);
            if (expand) {
                o[3] = q2_dword_perm(raw[P ^ 1][it], pp, 3);
                asm volatile("" : "+v"(o[3]));
                if (lane_stores)
                    *(uint4 *)(areg_lds + (size_t)(P ^ 1) * 64 * LDB + lane * LDB + (w + 4 * it) * 64 + pp * 16) = make_uint4(o[0], o[1], o[2], o[3]);
                if (pp == 3) load_raw(tt + 3 * G, raw[P ^ 1], it); // the chunk's registers are free: reload them for tile tt + 3G
            }
            if (epi) {
                epi_rows(std::integral_constant<int, (P ^ 1)>{}, s - EPI0, 6);
                epi_finish(std::integral_constant<int, (P ^ 1)>{}, s - EPI0, tt - G, prev_valid);
            }
            __builtin_amdgcn_sched_barrier(0);
        });
        if constexpr (EPI0 + 8 > KS) { // short K: the slices that did not fit in the k loop
            static_for<KS - EPI0, 8>([&](auto Ec) __attribute__((always_inline)) { epi_slice(std::integral_constant<int, (P ^ 1)>{}, decltype(Ec)::value, tt - G, prev_valid); });
        }
        __syncthreads();
    };
    bool prev_valid = false;
    int last = 1;
    while (true) {
        tile_body(std::integral_constant<int, 0>{}, t, prev_valid);
        prev_valid = true;
        last = 0;
        t += G;
        if (t >= n_tiles) break;
        tile_body(std::integral_constant<int, 1>{}, t, true);
        last = 1;
        t += G;
        if (t >= n_tiles) break;
    }
    // drain: epilogue of the last tile (t was advanced once past it)
    if (last == 0) {
        static_for<0, 8>([&](auto Ec) __attribute__((always_inline)) { epi_slice(std::integral_constant<int, 0>{}, decltype(Ec)::value, t - G, true); });
    } else {
        static_for<0, 8>([&](auto Ec) __attribute__((always_inline)) { epi_slice(std::integral_constant<int, 1>{}, decltype(Ec)::value, t - G, true); });
    }
    __syncthreads();
    const u32 staged = min(*stage_cnt, (u32)AREG_STAGE);
    for (u32 e = tid; e < staged; e += 256) areg_append(fo, qmags, mags, metric, n0, stage[3 * e], stage[3 * e + 1], stage[3 * e + 2]);
}


// ------------------------------------------------------------------------------------------------
// The same scan on the FP4 path of the scaled MFMA (v_mfma_scale_f32_32x32x64_f8f6f4, gfx950).  A quaternary digit d in {0,1,2,3}
// IS the e2m1 code of d / 2 (0000 = 0, 0001 = 0.5, 0010 = 1.0, 0011 = 1.5), so a code row becomes 4-bit operands with ONE and-or
// per dword (nibble = plane0 bit | plane1 bit << 1), half the LDS and AccVGPR bytes of the i8 digits, and one MFMA covers 64 dims
// where the i8 form covers 32 at the same issue cost.  Exactness: every product is a multiple of 1/4 below 2.25, a row's sum is
// below 2.25 x 4096 — all exactly representable in the f32 accumulator whatever the order of the additions — and unit block scales
// (E8M0 127 = 2^0) leave the values alone; acc x 4 is the integer dot_product_quaternary returns (dot_product.rs:125, x86_64.rs:
// 103-160), the epilogue folds the 4 into its reciprocal.  Only k ORDER inside a 64-dim chunk differs from the i8 kernel (both
// operands share it: the sum does not care).  Structure, staging, deferred epilogue, survivors: flat_scan_q2_areg above.
// ------------------------------------------------------------------------------------------------
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// 16 nibble bytes of piece pp (0, 1) of a 64-dim chunk: nibble n of dword jj is dim 32 * pp + jj + 4 * n
__device__ __forceinline__ u32 q2_nibble_dword(uint4 raw /*[plane0 8 B | plane1 8 B]*/, int pp, int jj) {
    const u32 w0 = pp == 0 ? raw.x : raw.y, w1 = pp == 0 ? raw.z : raw.w;
    const u32 x0 = w0 >> jj, x1 = jj ? (w1 >> (jj - 1)) : (w1 << 1);
    return (x0 & 0x11111111u) | (x1 & 0x22222222u);
}
__device__ __forceinline__ uint4 q2_nibble_piece(uint4 raw, int pp) {
    return make_uint4(q2_nibble_dword(raw, pp, 0), q2_nibble_dword(raw, pp, 1), q2_nibble_dword(raw, pp, 2), q2_nibble_dword(raw, pp, 3));
}

__global__ void expand_q2_nibbles_kernel(const uint8_t *__restrict__ qcodes, u64 row_stride, u32 B, u32 kdims, uint8_t *__restrict__ nib /*[B][kdims / 2]*/) {
    const u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    const u32 chunks = kdims / 64;
    if (t >= (u64)B * chunks) return;
    const u32 q = (u32)(t / chunks), j = (u32)(t % chunks);
    const uint4 raw = (u64)j * 16 < row_stride ? *(const uint4 *)(qcodes + (u64)q * row_stride + (u64)j * 16) : make_uint4(0, 0, 0, 0);
#pragma unroll
    for (int pp = 0; pp < 2; pp++) *(uint4 *)(nib + (u64)q * (kdims / 2) + (u64)j * 32 + pp * 16) = q2_nibble_piece(raw, pp);
}

template <bool FIRST>
__device__ __forceinline__ void fp4_mfma(f32x16 &acc, const i32x4 &afrag, const i32x4 &bfrag) {
    const f32x16 z = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    const i32x8 a8 = {afrag[0], afrag[1], afrag[2], afrag[3], 0, 0, 0, 0}, b8 = {bfrag[0], bfrag[1], bfrag[2], bfrag[3], 0, 0, 0, 0};
    // cbsz = blgp = 4: both operands FP4 (e2m1), 32 values per lane in the low four registers; scales: E8M0 127 in every byte
    acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, b8, FIRST ? z : acc, 4, 4, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
}

template <int KC>
__global__ __launch_bounds__(256) void flat_scan_q2_fp4(const uint8_t *__restrict__ qnib /*[B][32 KC] nibble bytes*/,
                                                        const float *__restrict__ qmags, u32 B, const uint8_t *__restrict__ codes,
                                                        const float *__restrict__ mags, u64 row_stride, u32 n0, u32 n_chunk, u32 metric,
                                                        const FusedOut fo) {
    constexpr int KB = KC * 32, KS = KC, LDB = KB + 16, ITS = (KC + 3) / 4, PIECES = ITS * 2; // KB = operand bytes per row, one k step per chunk
    extern __shared__ __attribute__((aligned(16))) unsigned char areg_lds[]; // [2][64][LDB] | survivors [AREG_STAGE][3] u32 | count
    u32 *stage = (u32 *)(areg_lds + (size_t)2 * 64 * LDB), *stage_cnt = stage + 3 * AREG_STAGE;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, half = lane >> 5, l31 = lane & 31;
    const u32 row0 = blockIdx.y * 256 + w * 64;
    const u32 n_tiles = (n_chunk + 63) / 64, G = gridDim.x;
    u32 t = blockIdx.x;
    if (t >= n_tiles) return; // uniform
    if (tid == 0) *stage_cnt = 0; // published by the barrier after the first tile's expansion
    i32x4 a[2][KS];
    static_for<0, 2 * KS>([&](auto Ic) __attribute__((always_inline)) {
        constexpr int i = decltype(Ic)::value / KS, s = decltype(Ic)::value % KS;
        const u32 row = row0 + 32 * i + l31; // rows past B: clamped loads (no branches in the prologue), zeroed
        const i32x4 v = *(const i32x4 *)(qnib + (u64)(row < B ? row : B - 1) * KB + 32 * s + 16 * half);
        a[i][s] = row < B ? v : i32x4{0, 0, 0, 0};
        asm volatile("" : "+a"(a[i][s])); // the value now IS an AccVGPR tuple
    });
    // thresholds of this lane's 32 accumulator rows, in the units of dot * (1 / |x|): T = thr_lo * |q| (cosine) or thr_lo (dot)
    float T[2][16];
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const u32 row = row0 + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * half, rc = row < B ? row : B - 1;
            const u64 k = fo.thr[rc];
            const float qm = qmags[rc];
            const float lo = k == 0ull ? -1.0f : simkey_inv((u32)(k >> 32)) * (1.0f - 4e-6f); // scores of quaternary codes are >= 0
            const float v = metric == 0u ? lo * qm : lo;
            T[i][r] = row < B ? v : __builtin_inff();
        }
    uint4 raw[2][ITS];
    auto load_raw = [&](u32 tile, uint4 *dst, int it) __attribute__((always_inline)) {
        const int j = w + 4 * it;
        const u32 c = tile * 64 + lane, cc = c < n_chunk ? c : n_chunk - 1;
        dst[it] = *(const uint4 *)(codes + (u64)(n0 + cc) * row_stride + (u64)(KC % 4 == 0 || j < KC ? j : 0) * 16);
    };
    auto store_piece = [&](int buf, const uint4 *src, int it, int pp) __attribute__((always_inline)) {
        const int j = w + 4 * it;
        if (KC % 4 == 0 || j < KC) *(uint4 *)(areg_lds + (size_t)buf * 64 * LDB + lane * LDB + j * 32 + pp * 16) = q2_nibble_piece(src[it], pp);
    };
    float xm[2][2]; // candidate norms of the tile in accumulator set 0 / 1 (consumed by the deferred epilogue)
    auto load_norms = [&](u32 tile, float *dst) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < 2; j++) {
            const u32 col = tile * 64 + 32 * j + l31;
            dst[j] = mags[n0 + (col < n_chunk ? col : n_chunk - 1)];
        }
    };
#pragma unroll
    for (int it = 0; it < ITS; it++) load_raw(t, raw[0], it);
#pragma unroll
    for (int it = 0; it < ITS; it++)
#pragma unroll
        for (int pp = 0; pp < 2; pp++) store_piece(0, raw[0], it, pp);
#pragma unroll
    for (int it = 0; it < ITS; it++) {
        load_raw(t + G, raw[1], it);
        load_raw(t + 2 * G, raw[0], it);
    }
    xm[1][0] = xm[1][1] = 1.0f;
    __syncthreads();
    f32x16 acc[2][2][2]; // [set][row block][column block]: dot / 4, an exact multiple of 1/4
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[1][i][j][r] = 0.0f;

    float best = 0.0f;
    auto epi_rows = [&](auto Pc, int e, int r0) __attribute__((always_inline)) {
        constexpr int P = decltype(Pc)::value;
        const int j = e >> 2, i = (e >> 1) & 1, h8 = (e & 1) * 8;
        const float rx = metric == 0u ? 4.0f * __builtin_amdgcn_rcpf(xm[P][j]) : 4.0f; // the accumulators hold dot / 4
        if (h8 == 0 && r0 == 0) best = -__builtin_inff();
#pragma unroll
        for (int r = r0; r < r0 + 2; r++) best = fmaxf(best, __builtin_fmaf(acc[P][i][j][h8 + r], rx, -T[i][h8 + r]));
    };
    auto epi_finish = [&](auto Pc, int e, u32 tp, bool tp_valid) __attribute__((always_inline)) {
        constexpr int P = decltype(Pc)::value;
        const int j = e >> 2, i = (e >> 1) & 1;
        if ((e & 1) == 0) return;
        const u32 col = tp * 64 + 32 * j + l31;
        if (best >= 0.0f && tp_valid && col < n_chunk) {
            const float rx = metric == 0u ? 4.0f * __builtin_amdgcn_rcpf(xm[P][j]) : 4.0f;
            u32 rbase = row0 + 32 * i + 4 * half;
            asm volatile("" : "+v"(rbase)); // opaque: keeps the 32 rows' addresses from being hoisted out of the tile loop
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const u32 row = rbase + (r & 3) + 8 * (r >> 2);
                const float q4 = acc[P][i][j][r];
                if (row < B && __builtin_fmaf(q4, rx, -T[i][r]) >= 0.0f) {
                    const u32 dot = (u32)(q4 * 4.0f); // exact: q4 is a multiple of 1/4 below 2^22
                    const u32 sp = atomicAdd(stage_cnt, 1u);
                    if (sp < AREG_STAGE) {
                        stage[3 * sp] = col;
                        stage[3 * sp + 1] = row;
                        stage[3 * sp + 2] = dot;
                    } else
                        areg_append(fo, qmags, mags, metric, n0, col, row, dot); // staging full: append from here
                }
            }
        }
    };
    auto epi_slice = [&](auto Pc, int e, u32 tp, bool tp_valid) __attribute__((always_inline)) {
        epi_rows(Pc, e, 0); epi_rows(Pc, e, 2); epi_rows(Pc, e, 4); epi_rows(Pc, e, 6);
        epi_finish(Pc, e, tp, tp_valid);
    };
    // One tile: MFMAs of tile tt into set P from LDS buffer P; every k step is four (MFMA, VALU) pairs — one dword of the
    // expansion of tile tt+G during the first PIECES steps, two rows of the epilogue of tile tt-G (set P^1) during the last 8
    constexpr int EPI0 = KS >= 8 ? KS - 8 : 0; // first step that carries an epilogue slice
    auto tile_body = [&](auto Pc, u32 tt, bool prev_valid) __attribute__((always_inline)) {
        constexpr int P = decltype(Pc)::value;
        const unsigned char *bt = areg_lds + (size_t)P * 64 * LDB + l31 * LDB + 16 * half;
        i32x4 bf[3][2]; // candidate fragments, read two k steps ahead of their MFMAs
#pragma unroll
        for (int s = 0; s < 2 && s < KS; s++) {
            bf[s][0] = *(const i32x4 *)(bt + 32 * s);
            bf[s][1] = *(const i32x4 *)(bt + 32 * LDB + 32 * s);
        }
        load_norms(tt, xm[P]);
        static_for<0, KS>([&](auto Sc) __attribute__((always_inline)) {
            constexpr int s = decltype(Sc)::value;
            constexpr bool expand = s < PIECES, epi = s >= EPI0 && s < EPI0 + 8;
            constexpr int it = (s < PIECES ? s : 0) >> 1, pp = s & 1;
            if (s + 2 < KS) {
                bf[(s + 2) % 3][0] = *(const i32x4 *)(bt + 32 * (s + 2));
                bf[(s + 2) % 3][1] = *(const i32x4 *)(bt + 32 * LDB + 32 * (s + 2));
            }
            __builtin_amdgcn_sched_barrier(0);
            const i32x4 b0 = bf[s % 3][0], b1 = bf[s % 3][1];
            const bool lane_stores = KC % 4 == 0 || w + 4 * it < KC;
            u32 o[4] = {0, 0, 0, 0};
            fp4_mfma<s == 0>(acc[P][0][0], a[0][s], b0);
            if (expand) { o[0] = q2_nibble_dword(raw[P ^ 1][it], pp, 0); asm volatile("" : "+v"(o[0])); } // pinned behind its MFMA
            if (epi) epi_rows(std::integral_constant<int, (P ^ 1)>{}, s - EPI0, 0);
            __builtin_amdgcn_sched_barrier(0);
            fp4_mfma<s == 0>(acc[P][0][1], a[0][s], b1);
            if (expand) { o[1] = q2_nibble_dword(raw[P ^ 1][it], pp, 1); asm volatile("" : "+v"(o[1])); }
            if (epi) epi_rows(std::integral_constant<int, (P ^ 1)>{}, s - EPI0, 2);
            __builtin_amdgcn_sched_barrier(0);
            fp4_mfma<s == 0>(acc[P][1][0], a[1][s], b0);
            if (expand) { o[2] = q2_nibble_dword(raw[P ^ 1][it], pp, 2); asm volatile("" : "+v"(o[2])); }
            if (epi) epi_rows(std::integral_constant<int, (P ^ 1)>{}, s - EPI0, 4);
            __builtin_amdgcn_sched_barrier(0);
            fp4_mfma<s == 0>(acc[P][1][1], a[1][s], b1);
            if (expand) {
                o[3] = q2_nibble_dword(raw[P ^ 1][it], pp, 3);
                asm volatile("" : "+v"(o[3]));
                if (lane_stores)
                    *(uint4 *)(areg_lds + (size_t)(P ^ 1) * 64 * LDB + lane * LDB + (w + 4 * it) * 32 + pp * 16) = make_uint4(o[0], o[1], o[2], o[3]);
                if (pp == 1) load_raw(tt + 3 * G, raw[P ^ 1], it); // the chunk's registers are free: reload them for tile tt + 3G
            }
            if (epi) {
                epi_rows(std::integral_constant<int, (P ^ 1)>{}, s - EPI0, 6);
                epi_finish(std::integral_constant<int, (P ^ 1)>{}, s - EPI0, tt - G, prev_valid);
            }
            __builtin_amdgcn_sched_barrier(0);
        });
        if constexpr (PIECES > KS) { // (never: 2 * ceil(KC / 4) <= KC for every instantiated KC >= 2; kept as a guard)
            static_assert(PIECES <= KS, "the expansion must fit the k loop");
        }
        if constexpr (EPI0 + 8 > KS) { // short K: the slices that did not fit in the k loop
            static_for<KS - EPI0, 8>([&](auto Ec) __attribute__((always_inline)) { epi_slice(std::integral_constant<int, (P ^ 1)>{}, decltype(Ec)::value, tt - G, prev_valid); });
        }
        __syncthreads();
    };
    bool prev_valid = false;
    int last = 1;
    while (true) {
        tile_body(std::integral_constant<int, 0>{}, t, prev_valid);
        prev_valid = true;
        last = 0;
        t += G;
        if (t >= n_tiles) break;
        tile_body(std::integral_constant<int, 1>{}, t, true);
        last = 1;
        t += G;
        if (t >= n_tiles) break;
    }
    if (last == 0) {
        static_for<0, 8>([&](auto Ec) __attribute__((always_inline)) { epi_slice(std::integral_constant<int, 0>{}, decltype(Ec)::value, t - G, true); });
    } else {
        static_for<0, 8>([&](auto Ec) __attribute__((always_inline)) { epi_slice(std::integral_constant<int, 1>{}, decltype(Ec)::value, t - G, true); });
    }
    __syncthreads();
    const u32 staged = min(*stage_cnt, (u32)AREG_STAGE);
    for (u32 e = tid; e < staged; e += 256) areg_append(fo, qmags, mags, metric, n0, stage[3 * e], stage[3 * e + 1], stage[3 * e + 2]);
}

// ------------------------------------------------------------------------------------------------
// The FP4 scan with TWO waves per SIMD (round 6; tuning knob flat_fp4_w8).  The kernel above runs one wave per SIMD: every MFMA latency, LDS
// refill and barrier skew is exposed, and whatever should hide in an MFMA's shadow has to be pinned behind it by hand.  Here a workgroup is
// EIGHT waves: wave w keeps the query rows of row group w & 3 (64 rows, the same fragments as above) and multiplies them with ONE of the
// tile's two column blocks (w >> 2), so a wave's share of a tile is half the MFMAs, half the expansion and half the epilogue, the LDS
// traffic of the workgroup is what it was (a fragment is still read once per row group), and the two waves of a SIMD cover each other.
// What makes two waves fit the 128 + 128 register split of a 256-register wave: ONE accumulator set (32 VGPRs instead of 128) — the
// epilogue of a tile runs in line, right after its last k step, while the SIMD's other wave multiplies.  Same estimate, same survivors,
// same staging and exact test as above: the results are the tile kernel's bit for bit.
// ------------------------------------------------------------------------------------------------
template <int KC, int ORDER>
__global__ __launch_bounds__(512) void flat_scan_q2_fp4_w8(const uint8_t *__restrict__ qnib /*[B][32 KC] nibble bytes*/, const float *__restrict__ qmags, u32 B,
                                                           const uint8_t *__restrict__ codes, const float *__restrict__ mags, u64 row_stride, u32 n0,
                                                           u32 n_chunk, u32 metric, const FusedOut fo) {
    // ORDER (tuning knob flat_fp4_w8 = ORDER + 1; each step measured on 10M x 768, profiles/r06_fp4_w8_*):
    //   0  as described above, instruction order left to the compiler;
    //   1  (KC 12) the expansion dealt evenly: 64 columns x 12 chunks x 2 sixteen-byte pieces = 1536 pieces = THREE per thread — wave w
    //      expands chunk w (both pieces) and piece w & 1 of chunk 8 + (w >> 1), where ORDER 0 gives chunks 8..11 whole to waves 0..3;
    //   2  + the two row blocks' MFMAs pinned to ALTERNATE on two accumulators.  Left alone the allocator folds both row blocks onto one
    //      16-register accumulator: twelve dependent MFMAs with an s_waitcnt between each pair (an issue slot between two MFMAs on the
    //      same accumulator costs ~43 cycles), the second row block's chain sunk below the first one's epilogue;
    //   3  + tiles of 128 columns (two sub-tiles per barrier: a wave multiplies column blocks cb and cb + 2): half the barriers, so
    //      half the tile tails in which a SIMD's first-finished wave is parked and its other wave runs alone.
    static_assert(ORDER == 0 || KC == 12, "the even deal is written for 12 chunks");
    constexpr int KB = KC * 32, KS = KC, LDB = KB + 16, ITS = (KC + 7) / 8, PIECES = ORDER >= 1 ? 3 : ITS * 2;
    constexpr int SUB = ORDER >= 3 ? 2 : 1, TW = 64 * SUB, NSET = SUB == 2 ? 1 : 2, AHEAD = NSET + 1; // raw register sets; tiles ahead of a reload
    static_assert(PIECES <= KS, "the expansion must fit the k loop");
    extern __shared__ __attribute__((aligned(16))) unsigned char areg_lds[]; // [2][TW][LDB] | survivors [AREG_STAGE][3] u32 | count
    u32 *stage = (u32 *)(areg_lds + (size_t)2 * TW * LDB), *stage_cnt = stage + 3 * AREG_STAGE;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, rg = w & 3, cb = w >> 2, half = lane >> 5, l31 = lane & 31;
    const u32 row0 = blockIdx.y * 256 + rg * 64;
    const u32 n_tiles = (n_chunk + TW - 1) / TW, G = gridDim.x;
    u32 t = blockIdx.x;
    if (t >= n_tiles) return; // uniform
    if (tid == 0) *stage_cnt = 0; // published by the barrier after the first tile's expansion
    i32x4 a[2][KS];
    static_for<0, 2 * KS>([&](auto Ic) __attribute__((always_inline)) {
        constexpr int i = decltype(Ic)::value / KS, s = decltype(Ic)::value % KS;
        const u32 row = row0 + 32 * i + l31; // rows past B: clamped loads, zeroed
        const i32x4 v = *(const i32x4 *)(qnib + (u64)(row < B ? row : B - 1) * KB + 32 * s + 16 * half);
        a[i][s] = row < B ? v : i32x4{0, 0, 0, 0};
        asm volatile("" : "+a"(a[i][s])); // the value now IS an AccVGPR tuple
    });
    float T[2][16]; // thresholds of this lane's 32 accumulator rows (flat_scan_q2_fp4)
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const u32 row = row0 + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * half, rc = row < B ? row : B - 1;
            const u64 k = fo.thr[rc];
            const float qm = qmags[rc];
            const float lo = k == 0ull ? -1.0f : simkey_inv((u32)(k >> 32)) * (1.0f - 4e-6f);
            const float v = metric == 0u ? lo * qm : lo;
            T[i][r] = row < B ? v : __builtin_inff();
        }
    // staging registers of one tile column (lane + 64 u): slot 0 = chunk w; slot 1 = ORDER 0: chunk w + 8 (waves 0..3 at KC 12), ORDER >= 1:
    // the two plane words (.x plane 0, .z plane 1) of piece w & 1 of chunk 8 + (w >> 1)
    const int xw = __builtin_amdgcn_readfirstlane(w);
    const int xj = 8 + (xw >> 1), xp = xw & 1;
    uint4 raw[NSET][SUB][ITS];
    auto load_raw = [&](u32 tile, uint4 *dst, int u, int it) __attribute__((always_inline)) {
        const u32 c = tile * TW + 64 * u + lane, cc = c < n_chunk ? c : n_chunk - 1;
        const uint8_t *rowp = codes + (u64)(n0 + cc) * row_stride;
        if (ORDER >= 1 && it == 1) {
            dst[1].x = *(const u32 *)(rowp + (u64)xj * 16 + 4 * xp);
            dst[1].z = *(const u32 *)(rowp + (u64)xj * 16 + 8 + 4 * xp);
            return;
        }
        const int j = w + 8 * it;
        dst[it] = *(const uint4 *)(rowp + (u64)(KC % 8 == 0 || j < KC ? j : 0) * 16);
    };
    auto store_piece = [&](int buf, const uint4 *src, int u, int it, int pp) __attribute__((always_inline)) {
        unsigned char *rowl = areg_lds + (size_t)buf * TW * LDB + (size_t)(64 * u + lane) * LDB;
        if (ORDER >= 1 && it == 1) { // the one extra piece (pp ignored: the registers hold piece xp's words as piece 0)
            *(uint4 *)(rowl + xj * 32 + xp * 16) = q2_nibble_piece(make_uint4(src[1].x, 0u, src[1].z, 0u), 0);
            return;
        }
        const int j = w + 8 * it;
        if (KC % 8 == 0 || j < KC) *(uint4 *)(rowl + j * 32 + pp * 16) = q2_nibble_piece(src[it], pp);
    };
#pragma unroll
    for (int u = 0; u < SUB; u++)
#pragma unroll
        for (int it = 0; it < ITS; it++) load_raw(t, raw[0][u], u, it);
#pragma unroll
    for (int u = 0; u < SUB; u++)
#pragma unroll
        for (int it = 0; it < ITS; it++)
#pragma unroll
            for (int pp = 0; pp < (ORDER >= 1 && it == 1 ? 1 : 2); pp++) store_piece(0, raw[0][u], u, it, pp);
#pragma unroll
    for (int u = 0; u < SUB; u++)
#pragma unroll
        for (int it = 0; it < ITS; it++) {
            if (NSET == 2) {
                load_raw(t + G, raw[NSET - 1][u], u, it);
                load_raw(t + 2 * G, raw[0][u], u, it);
            } else
                load_raw(t + G, raw[0][u], u, it);
        }
    __syncthreads();

    // one tile: this wave's column block(s) of tile tt from LDS buffer P, the expansion of tile tt + G into buffer P ^ 1 spread over the
    // first PIECES k steps of each sub-tile, each sub-tile's epilogue in line
    auto tile_body = [&](auto Pc, u32 tt) __attribute__((always_inline)) {
        constexpr int P = decltype(Pc)::value, RS = NSET == 2 ? (P ^ 1) : 0; // the register set that holds tile tt + G
        static_for<0, SUB>([&](auto Uc) __attribute__((always_inline)) {
            constexpr int u = decltype(Uc)::value;
            const unsigned char *bt = areg_lds + (size_t)P * TW * LDB + (size_t)(32 * (cb + 2 * u) + l31) * LDB + 16 * half;
            i32x4 bf[3]; // candidate fragments, read two k steps ahead of their MFMAs
#pragma unroll
            for (int s = 0; s < 2 && s < KS; s++) bf[s] = *(const i32x4 *)(bt + 32 * s);
            const u32 colx = tt * TW + 32 * (cb + 2 * u) + l31;
            const float xm = mags[n0 + (colx < n_chunk ? colx : n_chunk - 1)];
            f32x16 acc[2];
            static_for<0, KS>([&](auto Sc) __attribute__((always_inline)) {
                constexpr int s = decltype(Sc)::value;
                constexpr bool expand = s < PIECES;
                constexpr int it = ORDER >= 1 ? (s == 2 ? 1 : 0) : (s < PIECES ? s : 0) >> 1, pp = ORDER >= 1 && s == 2 ? 0 : s & 1;
                constexpr bool last_of_slot = ORDER >= 1 ? s >= 1 : pp == 1;
                if (s + 2 < KS) bf[(s + 2) % 3] = *(const i32x4 *)(bt + 32 * (s + 2));
                const i32x4 b = bf[s % 3];
                if constexpr (ORDER >= 2) { // pinned: MFMA row block 0 | half a piece | MFMA row block 1 | the other half, its store, the reload
                    const uint4 src = it == 1 ? make_uint4(raw[RS][u][1].x, 0u, raw[RS][u][1].z, 0u) : raw[RS][u][it];
                    u32 o[4] = {0, 0, 0, 0};
                    __builtin_amdgcn_sched_barrier(0);
                    fp4_mfma<s == 0>(acc[0], a[0][s], b);
                    __builtin_amdgcn_sched_barrier(0);
                    if (expand) {
                        o[0] = q2_nibble_dword(src, pp, 0);
                        o[1] = q2_nibble_dword(src, pp, 1);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    fp4_mfma<s == 0>(acc[1], a[1][s], b);
                    __builtin_amdgcn_sched_barrier(0);
                    if (expand) {
                        o[2] = q2_nibble_dword(src, pp, 2);
                        o[3] = q2_nibble_dword(src, pp, 3);
                        const int j = it == 1 ? xj : w, ppx = it == 1 ? xp : pp;
                        *(uint4 *)(areg_lds + (size_t)(P ^ 1) * TW * LDB + (size_t)(64 * u + lane) * LDB + j * 32 + ppx * 16) = make_uint4(o[0], o[1], o[2], o[3]);
                        if (last_of_slot) load_raw(tt + AHEAD * G, raw[RS][u], u, it);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                } else {
                    fp4_mfma<s == 0>(acc[0], a[0][s], b);
                    fp4_mfma<s == 0>(acc[1], a[1][s], b);
                    if (expand) {
                        store_piece(P ^ 1, raw[RS][u], u, it, pp);
                        if (last_of_slot) load_raw(tt + AHEAD * G, raw[RS][u], u, it); // the slot's registers are free: reload them
                    }
                }
            });
            // ORDER >= 2: both accumulators are complete HERE (without this the second row block's whole MFMA chain is sunk below the first
            // row block's epilogue, next to its only use, and the pinned order above is undone)
            if constexpr (ORDER >= 2) asm volatile("" : "+v"(acc[0]), "+v"(acc[1]));
            // the sub-tile's epilogue, in line (the SIMD's other wave multiplies meanwhile): "does any of my 16 rows pass" per row block,
            // then the exact staging of the rare survivors
            const float rx = metric == 0u ? 4.0f * __builtin_amdgcn_rcpf(xm) : 4.0f; // the accumulators hold dot / 4
#pragma unroll
            for (int i = 0; i < 2; i++) {
                float best = -__builtin_inff();
#pragma unroll
                for (int r = 0; r < 16; r++) best = fmaxf(best, __builtin_fmaf(acc[i][r], rx, -T[i][r]));
                if (best >= 0.0f && colx < n_chunk) {
                    u32 rbase = row0 + 32 * i + 4 * half;
                    asm volatile("" : "+v"(rbase));
#pragma unroll
                    for (int r = 0; r < 16; r++) {
                        const u32 row = rbase + (r & 3) + 8 * (r >> 2);
                        const float q4 = acc[i][r];
                        if (row < B && __builtin_fmaf(q4, rx, -T[i][r]) >= 0.0f) {
                            const u32 dot = (u32)(q4 * 4.0f); // exact: q4 is a multiple of 1/4 below 2^22
                            const u32 sp = atomicAdd(stage_cnt, 1u);
                            if (sp < AREG_STAGE) {
                                stage[3 * sp] = colx;
                                stage[3 * sp + 1] = row;
                                stage[3 * sp + 2] = dot;
                            } else
                                areg_append(fo, qmags, mags, metric, n0, colx, row, dot); // staging full: append from here
                        }
                    }
                }
            }
        });
        __syncthreads();
    };
    while (true) {
        tile_body(std::integral_constant<int, 0>{}, t);
        t += G;
        if (t >= n_tiles) break;
        tile_body(std::integral_constant<int, 1>{}, t);
        t += G;
        if (t >= n_tiles) break;
    }
    const u32 staged = min(*stage_cnt, (u32)AREG_STAGE);
    for (u32 e = tid; e < staged; e += 512) areg_append(fo, qmags, mags, metric, n0, stage[3 * e], stage[3 * e + 1], stage[3 * e + 2]);
}

// ------------------------------------------------------------------------------------------------
// Level table of the walk (WalkArgs::tab; engine.hip ensure_level_table / run_search) as a query-resident GEMM over u8 codes.
// Round 4 ran this product on flat_codes_gemm_i8 (the 256 x 128 tile kernel): 0.15-0.18 of the i8 peak, because every tile
// restages 256 query rows through LDS for 64 k steps' worth of MFMAs, and its epilogue formed the exact quotient dot / (|q| |v|)
// for each of B x cols outputs (685 M per launch at c2) of which the walk reads 13 %.  Round 5:
//   * the table holds the exact integer dot converted like the reference's `as f32` (x86_64.rs:22-66: u64 -> f32, RNE); the WALK
//     divides by |q| * |v| for the entries it reads — one vector quotient per expansion, the same two roundings in the same order
//     (cosine.rs:223-235), so the bits are what they were.  The epilogue of an output is one add, one convert, one store;
//   * the query operand never moves (the scan kernel's structure, above): a workgroup is 4 waves, wave w keeps the MFMA A
//     fragments of 64 query rows for the whole k range in AccVGPRs; column tiles of 64 table nodes x K code bytes stream
//     through double-buffered LDS (global -> registers while the MFMAs of the current tile run, registers -> LDS after its
//     epilogue, one barrier per tile); u8 codes are recentred by 128 on the way in (a' = a ^ 0x80 as i8) and the epilogue adds
//     128 (sum q + sum c) - 16384 K back (kernels_flat.hip: the same correction);
//   * what bounds it is the table itself: B x cols x 4 bytes written once (2.7 GB per 32 768 queries at c2) against 1.05 TOP —
//     13 TB/s of output at the i8 peak — so this is an HBM WRITE stream with the MFMA at ~40 % duty, and is reported as such.
// One workgroup per CU, persistent over its column tiles; the grid is (column groups, query groups of 256).
// ------------------------------------------------------------------------------------------------
// Q2 = quaternary codes (round 5: the c3 walk's table): the operands are the i8 digits of the exhaustive scan above — queries
// pre-expanded once per launch (expand_q2_digits_perm_kernel), table columns expanded planes -> digits while they are staged —
// and the dot needs no recentring.
template <int KC, bool Q2>
__global__ __launch_bounds__(256) void level_table_areg(const uint8_t *__restrict__ qcodes /* u8: [B][64 KC] codes; Q2: [B][64 KC] permuted digits */,
                                                        const u32 *__restrict__ qsums, u32 B, const uint8_t *__restrict__ tcodes,
                                                        const u32 *__restrict__ tcsums, u64 row_stride /* u8: 64 KC; Q2: 16 KC */, u32 ncols,
                                                        float *__restrict__ tab, u64 tab_stride) {
    constexpr int K = KC * 64, KS = KC * 2, LDB = K + 16;
    constexpr int PC = Q2 ? KC : K / 16;           // 16-byte RAW pieces per column (Q2: one per 64 dims, expanded to four in LDS)
    constexpr int NP = (64 * PC + 255) / 256;      // raw pieces a thread stages per tile
    extern __shared__ __attribute__((aligned(16))) unsigned char areg_lds[]; // [2][64][LDB]
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, half = lane >> 5, l31 = lane & 31;
    const u32 row0 = blockIdx.y * 256 + w * 64;
    const u32 n_tiles = (ncols + 63) / 64, G = gridDim.x;
    u32 t = blockIdx.x;
    if (t >= n_tiles) return; // uniform
    // resident query fragments (u8: recentred); rows past B are zero (their outputs are never stored)
    i32x4 a[2][KS];
    static_for<0, 2 * KS>([&](auto Ic) __attribute__((always_inline)) {
        constexpr int i = decltype(Ic)::value / KS, s = decltype(Ic)::value % KS;
        const u32 row = row0 + 32 * i + l31;
        i32x4 v = *(const i32x4 *)(qcodes + (u64)(row < B ? row : B - 1) * K + 32 * s + 16 * half);
        if constexpr (!Q2) v = v ^ (int)0x80808080;
        a[i][s] = row < B ? v : i32x4{0, 0, 0, 0};
        asm volatile("" : "+a"(a[i][s])); // the value now IS an AccVGPR tuple: its MFMA uses need no copies
    });
    // u8: per-row share of the recentring, 128 * sum(q) - 16384 * K, for this lane's 32 accumulator rows
    int rqs[2][16];
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const u32 row = row0 + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * half;
            rqs[i][r] = Q2 ? 0 : 128 * (int)qsums[row < B ? row : B - 1] - 16384 * K;
        }
    // staging: raw piece g = p * 256 + tid of the tile's 64 * PC pieces is bytes [16 (g % PC), +16) of column g / PC — consecutive
    // threads read consecutive 16 B of one code row (coalesced); columns past ncols re-read the last one (never stored)
    uint4 raw[NP];
    auto load_tile = [&](u32 tile) __attribute__((always_inline)) {
#pragma unroll
        for (int p = 0; p < NP; p++) {
            const u32 g = (u32)p * 256u + (u32)tid, gg = g < 64u * PC ? g : 0u, c = gg / (u32)PC, pc = gg % (u32)PC;
            const u32 col = tile * 64 + c, cc = col < ncols ? col : ncols - 1;
            raw[p] = *(const uint4 *)(tcodes + (u64)cc * row_stride + (u64)pc * 16);
        }
    };
    auto store_tile = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
        for (int p = 0; p < NP; p++) {
            const u32 g = (u32)p * 256u + (u32)tid, c = g / (u32)PC, pc = g % (u32)PC;
            if (64 * PC % 256 != 0 && g >= 64u * PC) continue;
            unsigned char *dst = areg_lds + (size_t)buf * 64 * LDB + (size_t)c * LDB;
            if constexpr (Q2) {
#pragma unroll
                for (int pp = 0; pp < 4; pp++) *(uint4 *)(dst + (size_t)pc * 64 + pp * 16) = q2_piece_perm(raw[p], pp);
            } else {
                uint4 v = raw[p];
                v.x ^= 0x80808080u; v.y ^= 0x80808080u; v.z ^= 0x80808080u; v.w ^= 0x80808080u;
                *(uint4 *)(dst + (size_t)pc * 16) = v;
            }
        }
    };
    load_tile(t);
    store_tile(0);
    __syncthreads();
    int P = 0;
    while (true) {
        const bool more = t + G < n_tiles; // uniform
        if (more) load_tile(t + G);       // in flight while this tile is multiplied
        const unsigned char *bt = areg_lds + (size_t)P * 64 * LDB + l31 * LDB + 16 * half;
        i32x16 acc[2][2];
        i32x4 bf[3][2]; // column fragments, read two k steps ahead of their MFMAs
#pragma unroll
        for (int s = 0; s < 2 && s < KS; s++) {
            bf[s][0] = *(const i32x4 *)(bt + 32 * s);
            bf[s][1] = *(const i32x4 *)(bt + 32 * LDB + 32 * s);
        }
        static_for<0, KS>([&](auto Sc) __attribute__((always_inline)) {
            constexpr int s = decltype(Sc)::value;
            if (s + 2 < KS) { // (pinned: left alone the scheduler sinks each read to just before its MFMAs and exposes the LDS latency)
                bf[(s + 2) % 3][0] = *(const i32x4 *)(bt + 32 * (s + 2));
                bf[(s + 2) % 3][1] = *(const i32x4 *)(bt + 32 * LDB + 32 * (s + 2));
            }
            __builtin_amdgcn_sched_barrier(0);
            const i32x4 b0 = bf[s % 3][0], b1 = bf[s % 3][1];
            areg_mfma<s == 0>(acc[0][0], a[0][s], b0);
            areg_mfma<s == 0>(acc[0][1], a[0][s], b1);
            areg_mfma<s == 0>(acc[1][0], a[1][s], b0);
            areg_mfma<s == 0>(acc[1][1], a[1][s], b1);
            __builtin_amdgcn_sched_barrier(0);
        });
        // epilogue: exact u32 dot -> f32 (RNE), 2 rows x 32 columns (two full 128-byte lines) per store instruction; a tile inside
        // the matrix (all but the last row group / column tile) stores without predication
        auto epilogue = [&](auto fullc) __attribute__((always_inline)) {
            constexpr bool FULL = decltype(fullc)::value;
#pragma unroll
            for (int j = 0; j < 2; j++) {
                const u32 col = t * 64 + 32 * j + l31;
                const bool cv = FULL || col < ncols;
                int cs = 0;
                if constexpr (!Q2) cs = 128 * (int)tcsums[cv ? col : ncols - 1];
#pragma unroll
                for (int i = 0; i < 2; i++) {
                    const u32 rbase = row0 + 32 * i + 4 * half;
                    float *out = tab + (u64)rbase * tab_stride + col;
#pragma unroll
                    for (int r = 0; r < 16; r++) {
                        const u32 dr = (r & 3) + 8 * (r >> 2);
                        const u32 dot = (u32)(acc[i][j][r] + rqs[i][r] + cs);
                        if (FULL || (cv && rbase + dr < B)) out[(u64)dr * tab_stride] = (float)dot;
                    }
                }
            }
        };
        if (row0 + 64 <= B && t * 64 + 64 <= ncols) epilogue(std::true_type{}); // wave-uniform
        else epilogue(std::false_type{});
        if (!more) break;
        store_tile(P ^ 1); // the other buffer: every wave finished reading it before the last barrier
        __syncthreads();
        P ^= 1;
        t += G;
    }
}

// ------------------------------------------------------------------------------------------------
// Query-resident scan of u8 codes (round 6; the exhaustive mode of u8 storage ran the 256 x 128 tile kernel flat_codes_gemm_i8 until
// now: every tile restages 256 query rows through LDS).  level_table_areg's loop — a workgroup is 4 waves, wave w keeps the recentred
// (^ 0x80) MFMA A fragments of 64 query rows for the whole k range in AccVGPRs, column tiles of 64 code rows stream through
// double-buffered LDS — with the fused scan's epilogue instead of a store: the exact integer dot (accumulator + 128 (sum q + sum c)
// - 16384 K, kernels_flat.hip: the same correction), converted like the reference's `as f32` (x86_64.rs:22-66), screened per lane
// against the 32 thresholds of its rows; the rare survivors are parked in LDS and get their exact quotient, key compare and global
// append at the end of the kernel (areg_append: the value that is ranked is dot / (|q| |v|) exactly as the tile kernel forms it, so the
// results are the tile kernel's bit for bit; tuning knob flat_tile_kernel = 1 keeps that one).
// ------------------------------------------------------------------------------------------------
template <int KC, int RB /* row blocks of 32 queries per wave: 2 (K <= 768), 1 (K = 1024: 128 AccVGPRs of fragments) */>
__global__ __launch_bounds__(256) void flat_scan_u8_areg(const uint8_t *__restrict__ qcodes /*[B][64 KC]*/, const u32 *__restrict__ qsums,
                                                         const float *__restrict__ qmags, u32 B, const uint8_t *__restrict__ codes,
                                                         const u32 *__restrict__ csums, const float *__restrict__ mags, u64 row_stride, u32 n0,
                                                         u32 n_chunk, u32 metric, const FusedOut fo) {
    constexpr int K = KC * 64, KS = KC * 2, LDB = K + 16, PC = K / 16, NP = (64 * PC + 255) / 256;
    extern __shared__ __attribute__((aligned(16))) unsigned char areg_lds[]; // [2][64][LDB] | survivors [AREG_STAGE][3] u32 | count
    u32 *stage = (u32 *)(areg_lds + (size_t)2 * 64 * LDB), *stage_cnt = stage + 3 * AREG_STAGE;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, half = lane >> 5, l31 = lane & 31;
    const u32 row0 = blockIdx.y * (128 * RB) + w * (32 * RB);
    const u32 n_tiles = (n_chunk + 63) / 64, G = gridDim.x;
    u32 t = blockIdx.x;
    if (t >= n_tiles) return; // uniform
    if (tid == 0) *stage_cnt = 0; // published by the first barrier
    i32x4 a[RB][KS];
    static_for<0, RB * KS>([&](auto Ic) __attribute__((always_inline)) {
        constexpr int i = decltype(Ic)::value / KS, s = decltype(Ic)::value % KS;
        const u32 row = row0 + 32 * i + l31;
        i32x4 v = *(const i32x4 *)(qcodes + (u64)(row < B ? row : B - 1) * K + 32 * s + 16 * half);
        v = v ^ (int)0x80808080;
        a[i][s] = row < B ? v : i32x4{0, 0, 0, 0};
        asm volatile("" : "+a"(a[i][s]));
    });
    int rqs[RB][16]; // 128 * sum(q) - 16384 K of this lane's 32 accumulator rows
    float T[RB][16]; // and their thresholds (flat_scan_q2_areg)
#pragma unroll
    for (int i = 0; i < RB; i++)
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const u32 row = row0 + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * half, rc = row < B ? row : B - 1;
            rqs[i][r] = 128 * (int)qsums[rc] - 16384 * K;
            const u64 k = fo.thr[rc];
            const float qm = qmags[rc];
            const float lo = k == 0ull ? -1.0f : simkey_inv((u32)(k >> 32)) * (1.0f - 4e-6f);
            const float v = metric == 0u ? lo * qm : lo;
            T[i][r] = row < B ? v : __builtin_inff();
        }
    uint4 raw[NP];
    auto load_tile = [&](u32 tile) __attribute__((always_inline)) {
#pragma unroll
        for (int p = 0; p < NP; p++) {
            const u32 g = (u32)p * 256u + (u32)tid, gg = g < 64u * PC ? g : 0u, c = gg / (u32)PC, pc = gg % (u32)PC;
            const u32 col = tile * 64 + c, cc = col < n_chunk ? col : n_chunk - 1;
            raw[p] = *(const uint4 *)(codes + (u64)(n0 + cc) * row_stride + (u64)pc * 16);
        }
    };
    auto store_tile = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
        for (int p = 0; p < NP; p++) {
            const u32 g = (u32)p * 256u + (u32)tid, c = g / (u32)PC, pc = g % (u32)PC;
            if (64 * PC % 256 != 0 && g >= 64u * PC) continue;
            uint4 v = raw[p];
            v.x ^= 0x80808080u; v.y ^= 0x80808080u; v.z ^= 0x80808080u; v.w ^= 0x80808080u;
            *(uint4 *)(areg_lds + (size_t)buf * 64 * LDB + (size_t)c * LDB + (size_t)pc * 16) = v;
        }
    };
    load_tile(t);
    store_tile(0);
    __syncthreads();
    int P = 0;
    while (true) {
        const bool more = t + G < n_tiles; // uniform
        if (more) load_tile(t + G);       // in flight while this tile is multiplied
        const unsigned char *bt = areg_lds + (size_t)P * 64 * LDB + l31 * LDB + 16 * half;
        // the two column blocks' code sums and norms: requested HERE, in front of the k loop, used after it (asked for in the epilogue they
        // cost a global-load latency per tile with nothing to cover it: a third of the kernel's wave time was s_waitcnt)
        int csj[2];
        float xmj[2];
#pragma unroll
        for (int j = 0; j < 2; j++) {
            const u32 col = t * 64 + 32 * j + l31, gc = n0 + (col < n_chunk ? col : n_chunk - 1);
            csj[j] = (int)csums[gc];
            xmj[j] = mags[gc];
        }
        i32x16 acc[RB][2];
        i32x4 bf[3][2];
#pragma unroll
        for (int s = 0; s < 2 && s < KS; s++) {
            bf[s][0] = *(const i32x4 *)(bt + 32 * s);
            bf[s][1] = *(const i32x4 *)(bt + 32 * LDB + 32 * s);
        }
        static_for<0, KS>([&](auto Sc) __attribute__((always_inline)) {
            constexpr int s = decltype(Sc)::value;
            if (s + 2 < KS) {
                bf[(s + 2) % 3][0] = *(const i32x4 *)(bt + 32 * (s + 2));
                bf[(s + 2) % 3][1] = *(const i32x4 *)(bt + 32 * LDB + 32 * (s + 2));
            }
            __builtin_amdgcn_sched_barrier(0);
            const i32x4 b0 = bf[s % 3][0], b1 = bf[s % 3][1];
            areg_mfma<s == 0>(acc[0][0], a[0][s], b0);
            areg_mfma<s == 0>(acc[0][1], a[0][s], b1);
            if constexpr (RB == 2) {
                areg_mfma<s == 0>(acc[RB - 1][0], a[RB - 1][s], b0);
                areg_mfma<s == 0>(acc[RB - 1][1], a[RB - 1][s], b1);
            }
            __builtin_amdgcn_sched_barrier(0);
        });
#pragma unroll
        for (int j = 0; j < 2; j++) {
            const u32 col = t * 64 + 32 * j + l31;
            const bool cv = col < n_chunk;
            const int cs = 128 * csj[j];
            const float rx = metric == 0u ? __builtin_amdgcn_rcpf(xmj[j]) : 1.0f;
#pragma unroll
            for (int i = 0; i < RB; i++) {
                float best = -__builtin_inff();
#pragma unroll
                for (int r = 0; r < 16; r++) best = fmaxf(best, __builtin_fmaf((float)(u32)(acc[i][j][r] + rqs[i][r] + cs), rx, -T[i][r]));
                if (best >= 0.0f && cv) {
                    u32 rbase = row0 + 32 * i + 4 * half;
                    asm volatile("" : "+v"(rbase));
#pragma unroll
                    for (int r = 0; r < 16; r++) {
                        const u32 row = rbase + (r & 3) + 8 * (r >> 2);
                        const u32 dot = (u32)(acc[i][j][r] + rqs[i][r] + cs);
                        if (row < B && __builtin_fmaf((float)dot, rx, -T[i][r]) >= 0.0f) {
                            const u32 sp = atomicAdd(stage_cnt, 1u);
                            if (sp < AREG_STAGE) {
                                stage[3 * sp] = col;
                                stage[3 * sp + 1] = row;
                                stage[3 * sp + 2] = dot;
                            } else
                                areg_append(fo, qmags, mags, metric, n0, col, row, dot); // staging full: append from here
                        }
                    }
                }
            }
        }
        if (!more) break;
        store_tile(P ^ 1); // the other buffer: every wave finished reading it before the last barrier
        __syncthreads();
        P ^= 1;
        t += G;
    }
    __syncthreads();
    const u32 staged = min(*stage_cnt, (u32)AREG_STAGE);
    for (u32 e = tid; e < staged; e += 256) areg_append(fo, qmags, mags, metric, n0, stage[3 * e], stage[3 * e + 1], stage[3 * e + 2]);
}

} // namespace

namespace cosdev {

// KC = k chunks of 64 dims; false when this K has no instantiation (the tile kernel runs instead)
bool flat_scan_supported(u32 kdims) {
    const u32 kc = kdims / 64;
    return kdims % 64 == 0 && (kc == 2 || kc == 4 || kc == 6 || kc == 8 || kc == 12 || kc == 16);
}
template <int KC>
static hipError_t launch_areg_kc(dim3 grid, hipStream_t st, const uint8_t *qdig, const float *qmags, u32 B, const uint8_t *codes, const float *mags,
                                 u64 row_stride, u32 n0, u32 nc, u32 metric, const FusedOut &fo) {
    const size_t lds = (size_t)2 * 64 * (KC * 64 + 16) + (size_t)AREG_STAGE * 12 + 16;
    hipError_t e = hipFuncSetAttribute((const void *)flat_scan_q2_areg<KC>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((flat_scan_q2_areg<KC>), grid, dim3(256), lds, st, qdig, qmags, B, codes, mags, row_stride, n0, nc, metric, fo);
    return hipGetLastError();
}
template <int KC>
static hipError_t launch_fp4_kc(dim3 grid, hipStream_t st, const uint8_t *qnib, const float *qmags, u32 B, const uint8_t *codes, const float *mags,
                                u64 row_stride, u32 n0, u32 nc, u32 metric, const FusedOut &fo) {
    const size_t lds = (size_t)2 * 64 * (KC * 32 + 16) + (size_t)AREG_STAGE * 12 + 16;
    if constexpr (KC == 12) { // the two-waves-per-SIMD kernel (768 dims: 96 AccVGPRs of query fragments; at 1024 dims they take all 128 and the rest spills): tuning knob flat_fp4_w8 = 0 (off) or the kernel's ORDER + 1
        const long long w8 = tune_or(TUNE_FLAT_FP4_W8, 4);
        auto go = [&](auto Oc) -> hipError_t {
            constexpr int ORDER = decltype(Oc)::value, TW = ORDER >= 3 ? 128 : 64;
            const size_t lds8 = (size_t)2 * TW * (KC * 32 + 16) + (size_t)AREG_STAGE * 12 + 16;
            hipError_t e8 = hipFuncSetAttribute((const void *)flat_scan_q2_fp4_w8<KC, ORDER>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds8);
            if (e8 != hipSuccess) return e8;
            const dim3 g8(std::min(grid.x, (nc + TW - 1) / TW), grid.y);
            hipLaunchKernelGGL((flat_scan_q2_fp4_w8<KC, ORDER>), g8, dim3(512), lds8, st, qnib, qmags, B, codes, mags, row_stride, n0, nc, metric, fo);
            return hipGetLastError();
        };
        if (w8 == 1) return go(std::integral_constant<int, 0>{});
        if (w8 == 2) return go(std::integral_constant<int, 1>{});
        if (w8 == 3) return go(std::integral_constant<int, 2>{});
        if (w8 >= 4) return go(std::integral_constant<int, 3>{});
    }
    hipError_t e = hipFuncSetAttribute((const void *)flat_scan_q2_fp4<KC>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((flat_scan_q2_fp4<KC>), grid, dim3(256), lds, st, qnib, qmags, B, codes, mags, row_stride, n0, nc, metric, fo);
    return hipGetLastError();
}
// fp4 = the e2m1 form of the digits on the scaled MFMA (flat_scan_q2_fp4; qdig then holds the nibble rows of
// launch_flat_scan_expand_queries(..., fp4 = true)), else the i8 digits (flat_scan_q2_areg)
hipError_t launch_flat_scan(u32 kdims, u32 n_cus, hipStream_t st, const uint8_t *qdig, const float *qmags, u32 B, const uint8_t *codes,
                              const float *mags, u64 row_stride, u32 n0, u32 nc, u32 metric, const FusedOut &fo, bool fp4) {
    const u32 n_tiles = (nc + 63) / 64;
    dim3 grid(std::min(n_tiles, n_cus), (B + 255) / 256);
    if (fp4) {
#define FP4_CASE(KC) case KC: return launch_fp4_kc<KC>(grid, st, qdig, qmags, B, codes, mags, row_stride, n0, nc, metric, fo)
        switch (kdims / 64) {
            FP4_CASE(2); FP4_CASE(4); FP4_CASE(6); FP4_CASE(8); FP4_CASE(12); FP4_CASE(16);
            default: return hipErrorInvalidValue;
        }
#undef FP4_CASE
    }
#define AREG_CASE(KC) case KC: return launch_areg_kc<KC>(grid, st, qdig, qmags, B, codes, mags, row_stride, n0, nc, metric, fo)
    switch (kdims / 64) {
        AREG_CASE(2); AREG_CASE(4); AREG_CASE(6); AREG_CASE(8); AREG_CASE(12); AREG_CASE(16);
        default: return hipErrorInvalidValue;
    }
#undef AREG_CASE
}


template <int KC>
static hipError_t launch_u8_kc(u32 n_cus, hipStream_t st, const uint8_t *qcodes, const u32 *qsums, const float *qmags, u32 B, const uint8_t *codes,
                               const u32 *csums, const float *mags, u64 row_stride, u32 n0, u32 nc, u32 metric, const FusedOut &fo) {
    constexpr int RB = KC > 12 ? 1 : 2;
    const size_t lds = (size_t)2 * 64 * (KC * 64 + 16) + (size_t)AREG_STAGE * 12 + 16;
    hipError_t e = hipFuncSetAttribute((const void *)flat_scan_u8_areg<KC, RB>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    const dim3 grid(std::min((nc + 63) / 64, n_cus), (B + 128 * RB - 1) / (128 * RB));
    hipLaunchKernelGGL((flat_scan_u8_areg<KC, RB>), grid, dim3(256), lds, st, qcodes, qsums, qmags, B, codes, csums, mags, row_stride, n0, nc, metric, fo);
    return hipGetLastError();
}
// fused chunk of u8 codes on the query-resident kernel (rows of exactly kdims = 64 KC bytes)
hipError_t launch_flat_scan_u8(u32 kdims, u32 n_cus, hipStream_t st, const uint8_t *qcodes, const u32 *qsums, const float *qmags, u32 B, const uint8_t *codes,
                               const u32 *csums, const float *mags, u64 row_stride, u32 n0, u32 nc, u32 metric, const FusedOut &fo) {
#define U8_CASE(KC) case KC: return launch_u8_kc<KC>(n_cus, st, qcodes, qsums, qmags, B, codes, csums, mags, row_stride, n0, nc, metric, fo)
    switch (kdims / 64) {
        U8_CASE(2); U8_CASE(4); U8_CASE(6); U8_CASE(8); U8_CASE(12); U8_CASE(16);
        default: return hipErrorInvalidValue;
    }
#undef U8_CASE
}

// level table as a query-resident GEMM: u8 codes whose rows are a whole number of 64-byte chunks with an instantiation, quaternary
// codes whose rows (16 B per 64 dims) have one
bool level_table_areg_supported(int eng, u64 row_stride) {
    if (eng == ENG_U8) return row_stride % 64 == 0 && flat_scan_supported((u32)row_stride);
    if (eng == ENG_Q2) return row_stride % 16 == 0 && flat_scan_supported((u32)(row_stride / 16) * 64);
    return false;
}
template <int KC, bool Q2>
static hipError_t launch_table_kc(dim3 grid, hipStream_t st, const uint8_t *qcodes, const u32 *qsums, u32 B, const uint8_t *tcodes, const u32 *tcsums,
                                  u64 row_stride, u32 ncols, float *tab, u64 tab_stride) {
    const size_t lds = (size_t)2 * 64 * (KC * 64 + 16);
    hipError_t e = hipFuncSetAttribute((const void *)level_table_areg<KC, Q2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((level_table_areg<KC, Q2>), grid, dim3(256), lds, st, qcodes, qsums, B, tcodes, tcsums, row_stride, ncols, tab, tab_stride);
    return hipGetLastError();
}
// tab[q][c] = (f32) integer dot of query q with table column c; grid = (column groups, query groups of 256), about one workgroup per CU.
// u8: qcodes = the queries' code rows, qsums / tcsums = code sums.  Q2: qcodes = the queries' permuted digit rows
// (launch_flat_scan_expand_queries, fp4 = false), qsums / tcsums unused.
hipError_t launch_level_table_areg(int eng, u32 n_cus, hipStream_t st, const uint8_t *qcodes, const u32 *qsums, u32 B, const uint8_t *tcodes, const u32 *tcsums,
                                   u64 row_stride, u32 ncols, float *tab, u64 tab_stride) {
    const u32 n_tiles = (ncols + 63) / 64, row_groups = (B + 255) / 256;
    const u32 G = std::max(1u, std::min(n_tiles, n_cus / std::max(1u, std::min(row_groups, n_cus))));
    dim3 grid(G, row_groups);
    const u32 kc = eng == ENG_Q2 ? (u32)(row_stride / 16) : (u32)(row_stride / 64);
#define TAB_CASE(KC)                                                                                                                          \
    case KC:                                                                                                                                  \
        return eng == ENG_Q2 ? launch_table_kc<KC, true>(grid, st, qcodes, qsums, B, tcodes, tcsums, row_stride, ncols, tab, tab_stride)     \
                             : launch_table_kc<KC, false>(grid, st, qcodes, qsums, B, tcodes, tcsums, row_stride, ncols, tab, tab_stride)
    switch (kc) {
        TAB_CASE(2); TAB_CASE(4); TAB_CASE(6); TAB_CASE(8); TAB_CASE(12); TAB_CASE(16);
        default: return hipErrorInvalidValue;
    }
#undef TAB_CASE
}

hipError_t launch_flat_scan_expand_queries(const uint8_t *qcodes, u64 row_stride, u32 B, u32 kdims, uint8_t *digits, hipStream_t st, bool fp4) {
    if (fp4) {
        hipLaunchKernelGGL(expand_q2_nibbles_kernel, dim3((u32)(((u64)B * (kdims / 64) + 255) / 256)), dim3(256), 0, st, qcodes, row_stride, B, kdims, digits);
        return hipGetLastError();
    }
    hipLaunchKernelGGL(expand_q2_digits_perm_kernel, dim3((u32)(((u64)B * (kdims / 64) + 255) / 256)), dim3(256), 0, st, qcodes, row_stride, B, kdims, digits);
    return hipGetLastError();
}

} // namespace cosdev
