// device_common.h — wave64 helpers shared by the gfx950 kernels (CDNA4 only; no CUDA path).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

namespace cosdev {

typedef uint32_t u32;
typedef unsigned long long u64;

constexpr int WAVE = 64;
constexpr u32 ROW_EMPTY = 0xFFFFFFFFu; // empty neighbour slot in the device adjacency (vec-row space)
constexpr int KEEP_SEARCH = 100;       // vector_store.rs:1194 (search)
constexpr int KEEP_INDEX = 64;         // vector_store.rs:1194 (indexing)
constexpr int MAX_LEVELS = 16;

enum : int { ENG_U8 = 0, ENG_Q2 = 1, ENG_F32 = 2, ENG_F16 = 3, ENG_Q1 = 4, ENG_Q3 = 5 }; // Q1/Q2/Q3 = SubByte resolution 1/2/3

// f32::total_cmp as an unsigned key: a < b (total order)  <=>  simkey(a) < simkey(b)
__device__ __forceinline__ u32 simkey(float v) {
    u32 b = __float_as_uint(v);
    return b ^ ((b >> 31) ? 0xFFFFFFFFu : 0x80000000u);
}
__device__ __forceinline__ float simkey_inv(u32 k) {
    u32 b = (k & 0x80000000u) ? (k ^ 0x80000000u) : ~k;
    return __uint_as_float(b);
}
// MetricResult::cmp (models/types.rs:401-411): cosine/dot larger-is-better, euclid/hamming reversed.
__device__ __forceinline__ u32 metric_key(u32 metric, float v) {
    u32 k = simkey(v);
    return (metric == 1u || metric == 2u) ? ~k : k;
}
__device__ __forceinline__ float metric_key_inv(u32 metric, u32 k) {
    return simkey_inv((metric == 1u || metric == 2u) ? ~k : k);
}
__device__ __forceinline__ u64 pack_key(u32 hi, u32 lo) { return ((u64)hi << 32) | (u64)lo; }

// The IEEE round-to-nearest quotient num / den for the walk's cosine (cosine.rs:223-235: integer dot `as f32` over |q| * |v|): the
// algorithm of the compiler's own f32 division (v_rcp, one Newton step, two residual corrections through FMAs) WITHOUT v_div_scale /
// v_div_fmas / v_div_fixup.  Those three only act when an operand is denormal, the denominator's exponent is near the top of the
// range or the two exponents differ by ~96 or more; for u8 codes 0 <= num < 2^27 and 1 <= den < 2^28 (norms of code rows: roots of
// integer sums; the SubByte / float storages, whose norms come from the raw vectors, keep the full division), so the scale
// factors are 1, div_fmas is a plain FMA and the fix-up is the identity: same bits, four instructions less per quotient.
// (den == 0 never reaches it: the callers test for CalculationError first.)
__device__ __forceinline__ float div_rn_unscaled(float num, float den) {
    float r = __builtin_amdgcn_rcpf(den);
    const float e = __builtin_fmaf(-den, r, 1.0f);
    r = __builtin_fmaf(e, r, r);
    float q = num * r;
    float t = __builtin_fmaf(-den, q, num);
    q = __builtin_fmaf(t, r, q);
    t = __builtin_fmaf(-den, q, num);
    return __builtin_fmaf(t, r, q);
}

__device__ __forceinline__ u64 shfl_xor_u64(u64 v, int m) {
    u32 lo = (u32)v, hi = (u32)(v >> 32);
    lo = (u32)__shfl_xor((int)lo, m, WAVE);
    hi = (u32)__shfl_xor((int)hi, m, WAVE);
    return ((u64)hi << 32) | lo;
}
__device__ __forceinline__ u64 shfl_up1_u64(u64 v) { // lane l <- lane l-1 (lane 0 keeps its own value)
    u32 lo = (u32)v, hi = (u32)(v >> 32);
    lo = (u32)__shfl_up((int)lo, 1, WAVE);
    hi = (u32)__shfl_up((int)hi, 1, WAVE);
    return ((u64)hi << 32) | lo;
}
__device__ __forceinline__ u64 shfl_down1_u64(u64 v) { // lane l <- lane l+1 (lane 63 keeps its own value)
    u32 lo = (u32)v, hi = (u32)(v >> 32);
    lo = (u32)__shfl_down((int)lo, 1, WAVE);
    hi = (u32)__shfl_down((int)hi, 1, WAVE);
    return ((u64)hi << 32) | lo;
}
// ---- DPP (data-parallel primitives): cross-lane moves inside the VALU, no LDS round trip -------------
// dpp_ctrl encodings (gfx9): quad_perm = 0x00-0xFF, row_shr:n = 0x110+n, wave_shl:1 = 0x130, wave_shr:1 = 0x138,
// row_mirror = 0x140, row_half_mirror = 0x141.
template <int CTRL>
__device__ __forceinline__ u32 dpp_mov(u32 fill, u32 v) {
    return (u32)__builtin_amdgcn_update_dpp((int)fill, (int)v, CTRL, 0xf, 0xf, false);
}
// the same moves with bound_ctrl: the lane without a source reads 0 and no `old` operand has to be prepared
template <int CTRL>
__device__ __forceinline__ u32 dpp_mov_z(u32 v) {
    return (u32)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xf, 0xf, true);
}
__device__ __forceinline__ u64 dpp_wave_shr1_u64(u64 v, u64 fill) { // lane l <- lane l-1, lane 0 <- fill
    u32 lo = dpp_mov<0x138>((u32)fill, (u32)v), hi = dpp_mov<0x138>((u32)(fill >> 32), (u32)(v >> 32));
    return ((u64)hi << 32) | lo;
}
__device__ __forceinline__ u64 dpp_wave_shl1_u64(u64 v, u64 fill) { // lane l <- lane l+1, lane 63 <- fill
    u32 lo = dpp_mov<0x130>((u32)fill, (u32)v), hi = dpp_mov<0x130>((u32)(fill >> 32), (u32)(v >> 32));
    return ((u64)hi << 32) | lo;
}
// sum over aligned groups of G lanes (G power of two, 2..64); every lane of a group (G <= 32) or of the
// wave (G = 64) ends with the group total.  Integer adds: order-independent, exact.
__device__ __forceinline__ u32 group_reduce_add_u32(u32 v, int G) {
    if (G >= 2) v += dpp_mov<0xB1>(0u, v);     // quad_perm [1,0,3,2]: xor 1
    if (G >= 4) v += dpp_mov<0x4E>(0u, v);     // quad_perm [2,3,0,1]: xor 2
    if (G >= 8) v += dpp_mov<0x141>(0u, v);    // row_half_mirror: the other quad of the 8-lane half
    if (G >= 16) v += dpp_mov<0x140>(0u, v);   // row_mirror: the other half of the 16-lane row
    if (G == 32) v += (u32)__shfl_xor((int)v, 16, WAVE);
    if (G == 64) {
        const u32 t = (u32)__builtin_amdgcn_readlane((int)v, 0) + (u32)__builtin_amdgcn_readlane((int)v, 16) +
                      (u32)__builtin_amdgcn_readlane((int)v, 32) + (u32)__builtin_amdgcn_readlane((int)v, 48);
        v = t;
    }
    return v;
}

// ---- sums of several vectors at once ---------------------------------------------------------------------------------
// wave_reduce_rows<NP>(a): NP vectors in, ONE vector out; every lane of the aligned group g (64/NP lanes, g = 0..NP-1) ends with
// the sum of a[g] over the whole wave.  A butterfly that halves the number of live vectors at each of the wide strides
// (gfx950's v_permlane32_swap / v_permlane16_swap move half of each of two vectors in one instruction, the add folds both),
// then finishes inside the groups with DPP.  18 VALU for 8 rows where group_reduce_add_u32(., 64) spends 8 per row (4 DPP adds,
// each waiting on the last, + 4 v_readlane) plus 3 scalar adds: the walk's evaluation block was its largest VALU consumer.
// Integer adds: order-independent, exact.
__device__ __forceinline__ u32 fold_halves(u32 a, u32 b) { // lanes 0-31: a[l] + a[l+32]; lanes 32-63: b[l-32] + b[l]
    const auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false); // r[0] = {a.lo, b.lo}, r[1] = {a.hi, b.hi}
    return r[0] + r[1];
}
__device__ __forceinline__ u32 fold_row_pairs(u32 a, u32 b) { // 16-lane rows: {a.r0+a.r1, b.r0+b.r1, a.r2+a.r3, b.r2+b.r3}
    const auto r = __builtin_amdgcn_permlane16_swap(a, b, false, false); // r[0] = {a.r0, b.r0, a.r2, b.r2}, r[1] = {a.r1, b.r1, a.r3, b.r3}
    return r[0] + r[1];
}
__device__ __forceinline__ u32 fold_row_halves(u32 c, u32 d) { // lanes 0-7 of a row: c[l] + c[l+8]; lanes 8-15: d[l-8] + d[l]
    // row_ror:8 (0x128) is "xor 8" inside a 16-lane row; bank_mask picks the half of the row a DPP move writes
    const u32 x = (u32)__builtin_amdgcn_update_dpp((int)d, (int)c, 0x128, 0xf, 0x3, false); // lanes 0-7: c[l^8]; lanes 8-15: d[l]
    const u32 y = (u32)__builtin_amdgcn_update_dpp((int)c, (int)d, 0x128, 0xf, 0xc, false); // lanes 0-7: c[l];   lanes 8-15: d[l^8]
    return x + y;
}
template <int NP>
__device__ __forceinline__ u32 wave_reduce_rows(const u32 (&a)[NP]) {
    static_assert(NP == 4 || NP == 8, "4 or 8 rows");
    if constexpr (NP == 8) {
        // group g must end with row g: {a0|a4} {a1|a5} {a2|a6} {a3|a7} -> rows {a0,a2,a4,a6} {a1,a3,a5,a7} -> halves in row order
        const u32 b0 = fold_halves(a[0], a[4]), b1 = fold_halves(a[1], a[5]), b2 = fold_halves(a[2], a[6]), b3 = fold_halves(a[3], a[7]);
        const u32 c = fold_row_pairs(b0, b2), d = fold_row_pairs(b1, b3);
        u32 e = fold_row_halves(c, d);
        e += dpp_mov<0xB1>(0u, e);  // xor 1
        e += dpp_mov<0x4E>(0u, e);  // xor 2
        e += dpp_mov<0x141>(0u, e); // row_half_mirror: the other quad of the 8-lane group
        return e;
    } else {
        const u32 b0 = fold_halves(a[0], a[2]), b1 = fold_halves(a[1], a[3]); // {a0|a2} {a1|a3}
        u32 e = fold_row_pairs(b0, b1);                                       // rows {a0, a1, a2, a3}
        e += dpp_mov<0xB1>(0u, e);
        e += dpp_mov<0x4E>(0u, e);
        e += dpp_mov<0x141>(0u, e);
        e += dpp_mov<0x140>(0u, e); // row_mirror: the other half of the 16-lane row
        return e;
    }
}

__device__ __forceinline__ u64 readlane_u64(u64 v, int lane) {
    u32 lo = (u32)__builtin_amdgcn_readlane((int)(u32)v, lane);
    u32 hi = (u32)__builtin_amdgcn_readlane((int)(u32)(v >> 32), lane);
    return ((u64)hi << 32) | lo;
}
__device__ __forceinline__ u32 readlane_u32(u32 v, int lane) { return (u32)__builtin_amdgcn_readlane((int)v, lane); }
// A value every lane of the wave holds alike, handed to the compiler as such (v_readfirstlane: the result lives in an SGPR).  A
// wave-uniform value that arrives through a vector load (a pointer the kernel also writes through cannot take the scalar cache) is
// "divergent" to the compiler, and so is every branch, loop counter and exit condition that depends on it: the loop is then
// run with exec-mask bookkeeping (s_and_saveexec / s_andn2 exec chains, counters in VGPRs) instead of scalar branches.
// Wave vote straight from the compare's SGPR mask.  hip's __ballot / __any take an int: the bool is first turned into 0 / 1
// (v_cndmask) and compared again (v_cmp_ne) — two VALU instructions per vote, and the walk votes half a dozen times per expansion.
__device__ __forceinline__ u64 ballot64(bool p) { return __builtin_amdgcn_ballot_w64(p); }
// lane `lane` of `old` <- `val`, both wave-uniform (handed over through v_readfirstlane, which folds away when the compiler already
// holds them in SGPRs).  gfx9 reads one SGPR per VALU instruction over the constant bus: the lane select goes through M0.
__device__ __forceinline__ u32 writelane_dyn(u32 old, u32 val, int lane) {
    const u32 sv = (u32)__builtin_amdgcn_readfirstlane((int)val);
    const int sl = __builtin_amdgcn_readfirstlane(lane);
    asm("s_mov_b32 m0, %2\n\tv_writelane_b32 %0, %1, m0" : "+v"(old) : "s"(sv), "s"(sl) : "m0");
    return old;
}
__device__ __forceinline__ u32 uniform_u32(u32 v) { return (u32)__builtin_amdgcn_readfirstlane((int)v); }
__device__ __forceinline__ float uniform_f32(float v) { return __uint_as_float((u32)__builtin_amdgcn_readfirstlane((int)__float_as_uint(v))); }
// base + row * stride for wave-uniform operands, on the scalar ALU (4 SALU, result in an SGPR pair), as a GLOBAL pointer: the row
// load then takes it as its saddr operand with a 32-bit per-lane offset.  Left to the compiler, `base + (u64)row * stride +
// lane_offset` becomes v_mov + v_mad_u64_u32 (quarter rate) + v_add per row address.
__device__ __forceinline__ const uint8_t *row_ptr_scalar(const uint8_t *base, u32 row, u32 stride) {
    u32 lo, hi;
    const u32 blo = (u32)(u64)base, bhi = (u32)((u64)base >> 32);
    asm("s_mul_i32 %0, %2, %3\n\ts_mul_hi_u32 %1, %2, %3\n\ts_add_u32 %0, %0, %4\n\ts_addc_u32 %1, %1, %5"
        : "=&s"(lo), "=&s"(hi) : "s"(row), "s"(stride), "s"(blo), "s"(bhi) : "scc");
    return (const uint8_t *)(const __attribute__((address_space(1))) uint8_t *)(((u64)hi << 32) | lo);
}
// m with bit l cleared, one s_bitset0_b64 (the compiler's m & (m - 1) is three scalar instructions)
__device__ __forceinline__ u64 clear_bit_u64(u64 m, int l) {
    asm("s_bitset0_b64 %0, %1" : "+s"(m) : "s"(l));
    return m;
}
// lane LANE of `old` <- the wave-uniform `val` (v_writelane_b32 with an immediate lane select; this clang has no builtin for it)
template <int LANE>
__device__ __forceinline__ u32 writelane_u32(u32 old, u32 val) {
    asm("v_writelane_b32 %0, %1, %2" : "+v"(old) : "s"(val), "n"(LANE));
    return old;
}

// compile-time loop: f(std::integral_constant<int, I>) for I in [I0, N)
template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F &&f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

// Bitonic sort (descending) of 64*R u64 keys held R per lane, blocked layout e = lane*R + r.
// Empty entries are 0 and sink to the end.  Fully unrolled: every register index is static.
template <int R>
__device__ __forceinline__ void bitonic_sort_desc(u64 (&k)[R], int lane) {
    constexpr int N = WAVE * R;
#pragma unroll
    for (int size = 2; size <= N; size <<= 1) {
#pragma unroll
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            if (stride >= R) {
                const int lmask = stride / R;
#pragma unroll
                for (int r = 0; r < R; r++) {
                    u64 other = shfl_xor_u64(k[r], lmask);
                    int e = lane * R + r;
                    bool desc = (e & size) == 0;
                    bool lower = (e & stride) == 0;
                    bool keepmax = (desc == lower);
                    u64 mx = k[r] > other ? k[r] : other;
                    u64 mn = k[r] > other ? other : k[r];
                    k[r] = keepmax ? mx : mn;
                }
            } else {
#pragma unroll
                for (int r = 0; r < R; r++) {
                    if ((r & stride) == 0) {
                        int e = lane * R + r;
                        bool desc = (e & size) == 0;
                        u64 a = k[r], b = k[r | stride];
                        u64 mx = a > b ? a : b, mn = a > b ? b : a;
                        k[r] = desc ? mx : mn;
                        k[r | stride] = desc ? mn : mx;
                    }
                }
            }
        }
    }
}

// Sorted (descending) candidate pool of 64*R u64 keys, blocked layout e = lane*R + r; 0 = empty.
// Replaces the reference's BinaryHeap (vector_store.rs:1125): only the best (ef - popped) entries
// can ever be popped, so a bounded sorted pool reproduces the pop sequence exactly.
template <int R>
struct Pool {
    u64 e[R];
    __device__ __forceinline__ void clear() {
#pragma unroll
        for (int r = 0; r < R; r++) e[r] = 0;
    }
    __device__ __forceinline__ u64 head() const { return readlane_u64(e[0], 0); }
    // entry at sorted position I (wave-uniform result); static register index
    template <int I>
    __device__ __forceinline__ u64 peek() const { return readlane_u64(e[I % R], I / R); }
    // entry at sorted position pos, pos wave-uniform but only known at run time.  Every register's candidate is read with
    // v_readlane and the choice is made among SCALARS: selecting the vector register first (x = e[pos % R]) made the compiler
    // index the pool as an array and move it to scratch memory (40 B per lane, the ef 256 walk 35 % slower).
    __device__ __forceinline__ u64 peek_dyn(u32 pos) const {
        const int l = (int)(pos / (u32)R);
        const u32 rr = pos % (u32)R;
        u64 v = readlane_u64(e[0], l);
#pragma unroll
        for (int r = 1; r < R; r++) {
            const u64 t = readlane_u64(e[r], l);
            v = rr == (u32)r ? t : v;
        }
        return v;
    }
    // node index (low half) of the entry at sorted position I: one v_readlane
    template <int I>
    __device__ __forceinline__ u32 peek_node() const { return readlane_u32((u32)e[I % R], I / R); }
    __device__ __forceinline__ void pop_head(int lane) {
        // lane l <- lane l+1's first entry; lane 63 <- empty (bound_ctrl zero fill: two DPP moves, nothing else)
        const u64 nxt = ((u64)dpp_mov_z<0x130>((u32)(e[0] >> 32)) << 32) | dpp_mov_z<0x130>((u32)e[0]);
#pragma unroll
        for (int r = 0; r + 1 < R; r++) e[r] = e[r + 1];
        e[R - 1] = nxt;
    }
    // number of entries strictly greater than k (= insertion position); k is wave-uniform
    __device__ __forceinline__ int rank_of(u64 k) const {
        int p = 0;
#pragma unroll
        for (int r = 0; r < R; r++) p += __popcll(__ballot(e[r] > k));
        return p;
    }
    // insert wave-uniform key k at position p (entries >= p shift up by one, the last one drops)
    __device__ __forceinline__ void insert_at(u64 k, int p, int lane) {
        if constexpr (R == 1) {
            // one entry per lane: lanes above p take their lower neighbour's entry (two DPP moves + one compare + two selects),
            // lane p takes k by v_writelane (no compare against p, no broadcast of k into a VGPR pair)
            u32 lo = (u32)e[0], hi = (u32)(e[0] >> 32);
            const u32 slo = dpp_mov_z<0x138>(lo), shi = dpp_mov_z<0x138>(hi); // lane 0 has no lower neighbour and never shifts
            const bool up = lane > p;
            lo = up ? slo : lo;
            hi = up ? shi : hi;
            lo = writelane_dyn(lo, (u32)k, p);
            hi = writelane_dyn(hi, (u32)(k >> 32), p);
            e[0] = ((u64)hi << 32) | lo;
            return;
        }
        const int lp = p / R, rp = p % R;
        const u64 prev_last = dpp_wave_shr1_u64(e[R - 1], 0ull); // lane l <- lane l-1's last entry
#pragma unroll
        for (int r = R - 1; r >= 0; r--) {
            u64 src = (r == 0) ? prev_last : e[r > 0 ? r - 1 : 0];
            bool shift = (lane > lp) || (lane == lp && r > rp);
            bool ins = (lane == lp && r == rp);
            e[r] = ins ? k : (shift ? src : e[r]);
        }
    }
};

} // namespace cosdev
