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

__device__ __forceinline__ u64 readlane_u64(u64 v, int lane) {
    u32 lo = (u32)__builtin_amdgcn_readlane((int)(u32)v, lane);
    u32 hi = (u32)__builtin_amdgcn_readlane((int)(u32)(v >> 32), lane);
    return ((u64)hi << 32) | lo;
}
__device__ __forceinline__ u32 readlane_u32(u32 v, int lane) { return (u32)__builtin_amdgcn_readlane((int)v, lane); }

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
    __device__ __forceinline__ void pop_head(int lane) {
        const u64 nxt = dpp_wave_shl1_u64(e[0], 0ull); // lane l <- lane l+1's first entry; lane 63 <- empty
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
