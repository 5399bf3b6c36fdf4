// kernels_sparse.hip — learned-sparse (SPLADE-style) inverted index search.
//   SparseAnnQueryBasic::sequential_search          models/sparse_ann_query.rs:68-147
//   InvertedIndexNode::quantize                     models/inverted_index.rs:168-172
//   InvertedIndex::search_internal / finalize_sparse_ann_results (raw-value rerank)   indexes/inverted/mod.rs:278-381
// The reference keeps, per dimension, one list of vector ids per QUANTIZED value (key); a query dimension with quantized value qq
// adds qq * key to every vector of every key list it visits (all keys when qq is above the early-termination threshold, the
// upper keys otherwise).  The caller hands the lists over as CSR — dims[T] ascending, key_off[T][2^bits + 1], vec_ids; the device
// keeps one id-sorted (id, key) list per dimension and accumulates in LDS tiles of the vector-id space (below): HBM-bound streaming
// of the postings, no MFMA (integer adds).  Sums are exact u32 -> atomic adds in any order give the reference's value.  The reference returns the survivors of
// select_nth_unstable in hash-map order; here (as in the oracle) they are ordered by similarity descending, larger id first.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <vector>

#include "engine_internal.h"

using namespace cosdev;

namespace {

constexpr u32 SEL = 64; // candidates kept per query: top_k * reranking_factor <= 64

struct SparseDev {
    const u32 *dims;      // [T]
    const u64 *key_off;   // [T][Q + 1]
    const u32 *vec_ids;   // postings
    const u64 *row_off;   // raw vectors (rerank): [n + 1]
    const u32 *raw_dims;
    const float *raw_vals;
    u32 T, Q, n, bits;
    float upper;
};

// Rust `as u8` / `as u32` on f32 (saturating, NaN -> 0) and f32::clamp
__device__ __forceinline__ u32 f32_as_u8(float v) { return !(v == v) || v <= 0.0f ? 0u : (v >= 255.0f ? 255u : (u32)(int)v); }
__device__ __forceinline__ u32 f32_as_u32(float v) { return !(v == v) || v <= 0.0f ? 0u : (v >= 4294967296.0f ? 0xFFFFFFFFu : (u32)v); }
__device__ __forceinline__ u32 sparse_quantize(float value, float upper, u32 bits) { // inverted_index.rs:168-172
    const u32 quantization = (1u << bits) - 1u;
    const float max_val = (float)quantization;
    float t = __fmul_rn(__fdiv_rn(value, upper), max_val);
    t = t < 0.0f ? 0.0f : (t > max_val ? max_val : t); // clamp keeps NaN
    const u32 q = f32_as_u8(t);
    return q < quantization ? q : quantization;
}

// ---- device layout ----------------------------------------------------------------------------------------------------------
// The caller's CSR keeps, per dimension, one list per key (the reference's map key -> Vec<vec_id>).  On the device every dimension
// is ONE list sorted by vector id, the key of a posting next to it (m_ids u32 + m_keys u8 = 5 B per posting): the vector-id space
// can then be cut into tiles whose accumulators live in LDS, exactly like the BM25 kernel (kernels_hybrid.hip) —
//   * no [B][n] accumulator array in HBM at all (round 2: 268 M scattered global atomics into 410 MB + 512 MB cleared and scanned
//     per 256-query batch = 10.3 ms, 0.013 of the HBM roof on the postings);
//   * a term's early termination (keys below k0 are not visited) is a compare on the key byte.
// Sums are exact u32 adds (qq * key), so the order in which a tile's terms and postings arrive is irrelevant: LDS atomic adds,
// no barrier between terms.  A vector is a result as soon as any visited list holds it, also with similarity 0 (key 0, or a query
// value that quantizes to 0): those rare postings set a bit in a per-tile flag word instead.
constexpr u32 STILE = 8192;      // vector ids per LDS accumulator tile (32 KB of u32)
constexpr u32 SDIR_MIN = 256;    // dimensions with more postings get a tile directory; shorter lists are scanned whole per tile
constexpr u32 SNO_DIR = 0xFFFFFFFFu;
constexpr int SPU = 8;           // postings per thread per chunk
constexpr u32 SLICES = 256;      // (tile, term) slices a block resolves up front (LDS table); beyond that they are looked up on the way

struct STerm { // one resolved query term (host: find_node, quantize, early-termination rule)
    u64 begin, end; // the dimension's posting list in m_ids / m_keys
    u32 dir;        // row in the tile directory, SNO_DIR for short lists
    u32 qq_k0;      // quantized query value | first visited key << 8
};

struct SCursor { // block-uniform
    u32 tile, t;
    u64 base, e; // postings [base, min(base + SPU * 256, e)) of term t's slice of the tile
    bool valid;
};

// grid = B * splits blocks, heaviest query first: block (q, s) owns the tiles s, s + splits, ...; every wave keeps a private pool
// of the best SEL keys ((similarity + 1) << 32 | id) it has flushed, the block's four pools are merged into part[q][s][64].
__global__ __launch_bounds__(256) void sparse_tile_kernel(const u32 *__restrict__ m_ids, const uint8_t *__restrict__ m_keys, const STerm *__restrict__ terms,
                                                          const u32 *__restrict__ qt_off, u32 n, const u32 *__restrict__ tile_dir,
                                                          const u32 *__restrict__ order, u32 splits, u64 *__restrict__ part /*[B][splits][64]*/) {
    __shared__ u32 acc[STILE + 64]; // [STILE + lane] = the lane's dummy slot for postings that do not count (one per lane: same-address LDS atomics serialise)
    __shared__ u32 zflag[STILE / 32];
    __shared__ u64 wpool[4][SEL];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const u32 q = order[blockIdx.x / splits];
    const u32 split = blockIdx.x % splits;
    const u32 t0 = qt_off[q], nt = qt_off[q + 1] - t0;
    const u32 n_tiles = (n + STILE - 1) / STILE;
    u64 *out = part + ((u64)q * splits + split) * SEL;
    if (nt == 0 || split >= n_tiles) { // nothing to visit: an empty pool (the finish kernel reads every split)
        if (threadIdx.x < SEL) out[threadIdx.x] = 0ull;
        return;
    }
    for (u32 i = threadIdx.x; i < STILE; i += blockDim.x) acc[i] = 0u;
    for (u32 i = threadIdx.x; i < STILE / 32; i += blockDim.x) zflag[i] = 0u;
    const STerm *qt = terms + t0;
    Pool<1> pool;
    pool.clear();
    u64 thr = 0ull;

    auto slice_global = [&](u32 tile, u32 t, u64 &b, u64 &e) {
        const u32 dr = qt[t].dir;
        b = qt[t].begin;
        e = qt[t].end;
        if (dr != SNO_DIR) {
            const u32 *row = tile_dir + (u64)dr * (n_tiles + 1);
            e = b + row[tile + 1];
            b = b + row[tile];
        }
    };
    // The (tile, term) slices of this block, resolved once, all lookups in flight together: a chunk used to start with two
    // DEPENDENT global loads (the term's directory row, then its two entries) that nothing overlapped — with ~900 postings per slice
    // the kernel spent more time finding its slices than streaming them (0.95 ms per 256-query batch, 0.14 of the HBM roof).
    __shared__ u64 sl_b[SLICES];
    __shared__ u32 sl_n[SLICES], sl_w[SLICES]; // slice = [sl_b, sl_b + sl_n); a list has < 2^32 postings (cos_sparse_create)
    const u32 my_tiles = (n_tiles - split + splits - 1) / splits;
    const bool tabled = (u64)my_tiles * nt <= SLICES;
    if (tabled)
        for (u32 p = threadIdx.x; p < my_tiles * nt; p += blockDim.x) {
            u64 b, e;
            slice_global(split + (p / nt) * splits, p % nt, b, e);
            sl_b[p] = b;
            sl_n[p] = (u32)(e - b);
            if (p < nt) sl_w[p] = qt[p].qq_k0;
        }
    auto slice = [&](u32 tile, u32 t, u64 &b, u64 &e) {
        if (tabled) {
            const u32 p = ((tile - split) / splits) * nt + t;
            b = sl_b[p];
            e = b + sl_n[p];
        } else
            slice_global(tile, t, b, e);
    };
    auto advance = [&](const SCursor &c) -> SCursor {
        SCursor nx = c;
        if (c.base + (u64)SPU * 256 < c.e) { nx.base = c.base + (u64)SPU * 256; return nx; }
        if (c.t + 1 < nt) nx.t = c.t + 1;
        else { nx.t = 0; nx.tile = c.tile + splits; }
        nx.valid = nx.tile < n_tiles;
        if (nx.valid) slice(nx.tile, nx.t, nx.base, nx.e);
        return nx;
    };
    // every load is issued unconditionally (masked lanes read posting 0 and drop it): a fixed number of loads per chunk lets the
    // compiler wait for exactly the older chunk while the newer one stays in flight (see bm25_score_kernel)
    auto fetch = [&](const SCursor &c, u32 (&iv)[SPU], u32 (&kv)[SPU]) -> u32 {
        u32 mask = 0;
#pragma unroll
        for (int u = 0; u < SPU; u++) {
            const u64 i = c.base + threadIdx.x + (u64)u * 256;
            const bool in = c.valid && i < c.e;
            const u64 ii = in ? i : 0ull;
            iv[u] = m_ids[ii];
            kv[u] = m_keys[ii];
            mask |= (in ? 1u : 0u) << u;
        }
        return mask;
    };
    auto apply = [&](const SCursor &c, const SCursor &nx, const u32 (&iv)[SPU], const u32 (&kv)[SPU], const u32 mask) {
        const u32 d0 = c.tile * STILE;
        const u32 w_t = tabled ? sl_w[c.t] : qt[c.t].qq_k0;
        const u32 qq = w_t & 255u, k0 = w_t >> 8;
#pragma unroll
        for (int u = 0; u < SPU; u++) {
            const u32 slot = iv[u] - d0; // a short list's postings of other tiles wrap to >= STILE
            if (((mask >> u) & 1u) && slot < STILE && kv[u] >= k0) {
                const u32 w = qq * kv[u];
                if (w) atomicAdd(&acc[slot], w);
                else atomicOr(&zflag[slot >> 5], 1u << (slot & 31u));
            }
        }
        const bool tile_done = !nx.valid || nx.tile != c.tile;
        if (tile_done) {
            __syncthreads();
            // flush: wave w scans slots [w * 2048, (w + 1) * 2048) of the tile, 64 at a time, into its pool
            for (u32 s0 = (u32)wave * (STILE / 4); s0 < (u32)(wave + 1) * (STILE / 4); s0 += 64) {
                const u32 slot = s0 + (u32)lane;
                const u32 a = acc[slot];
                const u32 fw = zflag[slot >> 5];
                acc[slot] = 0u;
                const bool reached = a != 0u || ((fw >> (slot & 31u)) & 1u);
                const u64 key = reached ? (((u64)a + 1ull) << 32 | (u64)(d0 + slot)) : 0ull;
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
            __syncthreads(); // everybody has read the flag words of its slots
            for (u32 i = threadIdx.x; i < STILE / 32; i += blockDim.x) zflag[i] = 0u;
            __syncthreads();
        }
    };

    __syncthreads(); // the slice table
    // ---- stepped path (the usual case: the table holds every slice of the block and a query has <= 64 terms) --------------------
    // A tile's slices average ~900 postings, so the block-wide chunks below (2048 posting slots per (tile, term)) ran 43 % full and
    // the kernel was instruction-bound (135 lane-instructions per posting, profiles/archive/r03_sparse_tile_kernel_sq_counters_*.txt).  Here every WAVE
    // pulls STEPS — 512 consecutive postings of one slice — from a per-tile counter in LDS: a slice of L postings is ceil(L / 512)
    // steps, waves never wait for each other inside a tile (LDS atomic adds commute), a long slice spreads over the four waves, and
    // the next step's postings are in flight while the current one is applied.
    if (tabled && nt <= 64u) {
        __shared__ u32 st_pre[65]; // exclusive prefix of the tile's per-term step counts; [nt] = total
        __shared__ u32 st_ctr;
        constexpr u32 STEP = 64u * SPU;
        for (u32 ti = 0; ti < my_tiles; ti++) {
            const u32 tile = split + ti * splits, d0 = tile * STILE;
            if (wave == 0) { // step counts of this tile's slices -> exclusive prefix (one lane per term)
                u32 ns = 0;
                if ((u32)lane < nt) ns = (sl_n[ti * nt + lane] + STEP - 1u) / STEP;
                u32 incl = ns;
#pragma unroll
                for (int dd = 1; dd < 64; dd <<= 1) {
                    const u32 o = (u32)__shfl_up((int)incl, dd, 64);
                    if (lane >= dd) incl += o;
                }
                if ((u32)lane < nt) st_pre[lane] = incl - ns;
                if ((u32)lane == nt - 1u) st_pre[nt] = incl;
                if (lane == 0) st_ctr = 0u;
            }
            __syncthreads();
            const u32 total = st_pre[nt];
            // which slice does step k belong to: one lane per term tests its range, a ballot finds it (no dependent LDS chain)
            const u32 my_lo = (u32)lane < nt ? st_pre[lane] : 0xFFFFFFFFu, my_hi = (u32)lane < nt ? st_pre[lane + 1] : 0u;
            auto pull = [&]() -> u32 {
                u32 k = 0;
                if (lane == 0) k = atomicAdd(&st_ctr, 1u);
                return readlane_u32(k, 0);
            };
            struct Step { u64 base; u32 len, w; bool valid; };
            auto locate = [&](u32 k) -> Step {
                Step sp;
                sp.valid = k < total;
                sp.base = 0; sp.len = 0; sp.w = 0;
                if (sp.valid) {
                    const u64 m = __ballot(k >= my_lo && k < my_hi);
                    const u32 t = (u32)(__ffsll((long long)m) - 1);
                    const u32 p = ti * nt + t;
                    const u32 done = (k - readlane_u32(my_lo, (int)t)) * STEP, left = sl_n[p] - done;
                    sp.base = sl_b[p] + done;
                    sp.len = left < STEP ? left : STEP;
                    sp.w = sl_w[t];
                }
                return sp;
            };
            // The posting arrays carry STEP entries of padding (cos_sparse_create), so a step's 8 loads per lane are one base address
            // plus compile-time offsets, never clamped; lanes past the step's end read something and drop it below.
            auto fetch_s = [&](const Step &sp, u32 (&iv)[SPU], u32 (&kv)[SPU]) {
                const u32 *ip = m_ids + sp.base + lane;
                const uint8_t *kp = m_keys + sp.base + lane;
#pragma unroll
                for (int u = 0; u < SPU; u++) {
                    iv[u] = ip[u * 64];
                    kv[u] = kp[u * 64];
                }
            };
            // Branch-free: every lane issues its ds_add — a posting that does not count (past the step's end, another tile of a short
            // list, a key below the term's first visited key) adds 0 to the lane's dummy slot behind the tile.  The block-chunk version spent
            // ~70 instructions per posting slot on exec-mask bookkeeping around two predicated atomics (89 lane-instructions per posting,
            // profiles/archive/r03_sparse_tile_kernel_sq_counters_wave_steps.txt).  Weight-0 postings (key 0, or a query value that
            // quantizes to 0) are rare: one wave-level test per step.
            auto apply_s = [&](const Step &sp, const u32 (&iv)[SPU], const u32 (&kv)[SPU]) {
                const u32 qq = sp.w & 255u, k0 = sp.w >> 8;
                bool zero_any = false;
#pragma unroll
                for (int u = 0; u < SPU; u++) {
                    const u32 slot = iv[u] - d0; // a short list's postings of other tiles wrap to >= STILE
                    const bool ok = (u32)lane + (u32)u * 64u < sp.len && slot < STILE && kv[u] >= k0;
                    const u32 w = __umul24(qq, kv[u]); // both < 256: the full-rate 24-bit multiply
                    atomicAdd(&acc[ok ? slot : STILE + (u32)lane], ok ? w : 0u);
                    zero_any |= ok && w == 0u;
                }
                if (__any(zero_any)) {
#pragma unroll
                    for (int u = 0; u < SPU; u++) {
                        const u32 slot = iv[u] - d0;
                        if ((u32)lane + (u32)u * 64u < sp.len && slot < STILE && kv[u] >= k0 && qq * kv[u] == 0u) atomicOr(&zflag[slot >> 5], 1u << (slot & 31u));
                    }
                }
            };
            u32 ia[SPU], ib[SPU], ka[SPU], kb[SPU];
            Step sa = locate(pull()), sb;
            if (sa.valid) fetch_s(sa, ia, ka);
            while (sa.valid) { // ping-pong between the two register sets
                sb = locate(pull());
                if (sb.valid) fetch_s(sb, ib, kb);
                apply_s(sa, ia, ka);
                if (!sb.valid) break;
                sa = locate(pull());
                if (sa.valid) fetch_s(sa, ia, ka);
                apply_s(sb, ib, kb);
            }
            __syncthreads();
            // flush: wave w scans slots [w * 2048, (w + 1) * 2048) of the tile, 64 at a time, into its pool
            for (u32 s0 = (u32)wave * (STILE / 4); s0 < (u32)(wave + 1) * (STILE / 4); s0 += 64) {
                const u32 slot = s0 + (u32)lane;
                const u32 a = acc[slot];
                const u32 fw = zflag[slot >> 5];
                acc[slot] = 0u;
                const bool reached = a != 0u || ((fw >> (slot & 31u)) & 1u);
                const u64 key = reached ? (((u64)a + 1ull) << 32 | (u64)(d0 + slot)) : 0ull;
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
            __syncthreads(); // everybody has read the flag words of its slots
            for (u32 i = threadIdx.x; i < STILE / 32; i += blockDim.x) zflag[i] = 0u;
            // (the next tile's prefix barrier orders these stores before its first atomic)
        }
        wpool[wave][lane] = pool.e[0];
        __syncthreads();
        if (wave == 0) {
            for (int w = 1; w < 4; w++) {
                const u64 key = wpool[w][lane];
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
            out[lane] = pool.e[0];
        }
        return;
    }
    // ---- block-wide chunks: queries of more than 64 terms, or more (tile, term) slices than the table holds ----------------------
    SCursor cur;
    cur.tile = split; cur.t = 0; cur.valid = true;
    slice(cur.tile, 0, cur.base, cur.e);
    u32 ia[SPU], ib[SPU], ka[SPU], kb[SPU];
    u32 ma = fetch(cur, ia, ka), mb;
    __syncthreads();
    for (;;) { // ping-pong between the two register sets
        const SCursor n1 = advance(cur);
        mb = fetch(n1, ib, kb);
        apply(cur, n1, ia, ka, ma);
        if (!n1.valid) break;
        const SCursor n2 = advance(n1);
        ma = fetch(n2, ia, ka);
        apply(n1, n2, ib, kb, mb);
        if (!n2.valid) break;
        cur = n2;
    }
    // merge the four wave pools
    wpool[wave][lane] = pool.e[0];
    __syncthreads();
    if (wave == 0) {
        for (int w = 1; w < 4; w++) {
            const u64 key = wpool[w][lane];
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
        out[lane] = pool.e[0];
    }
}

// ---- packed layout (cos_sparse::packed: the default wherever vector ids fit 24 bits; tuning knob sparse_layout = 0 keeps the other) ----
// One u32 per posting, key << 24 | (vector id + 1), instead of a u32 id + a u8 key: 4 B instead of 5 B of HBM traffic per posting,
// one load instead of two, and the apply step of a posting shrinks from ~19 to ~9 instructions:
//   * a step's postings are fetched through a buffer descriptor whose num_records is the step's own length — a lane past the step's
//     end reads 0 = "vector id -1", which is outside every tile — so there is no `lane + 64 u < len` mask;
//   * slot = min((p - (d0 + 1)) & 0xFFFFFF, STILE + lane): postings of other tiles (a short list is scanned whole per tile) and the
//     out-of-range zeros wrap to a huge value and land on a dummy slot behind the tile; no compare, no select;
//   * COUNTED blocks (the usual case): a posting adds qq * key + 2^22, so a slot's high 10 bits count the postings that reached the
//     vector and the low 22 bits hold the similarity — "visited with similarity 0" needs no flag word and no second pass.  The
//     host checks per query that neither field can overflow (sum over its terms of multiplicity x qq x (Q - 1) < 2^22, at most
//     1023 touches; `multiplicity` = how often one vector id occurs in the dimension's list, 1 unless the caller's CSR repeats ids);
//     other queries run the same steps with the flag words of the unpacked kernel.
// Vector ids need 24 bits: n_vectors <= SPK_MAX_N, larger collections keep the unpacked layout.  Queries of more than 64 terms are
// walked in groups of 64 terms (the step directory is one lane per term); slices the LDS table does not hold are looked up on the way.
constexpr u32 SPK_CNT = 1u << 22;
constexpr u32 SPK_SUM = SPK_CNT - 1u;
constexpr u32 SPK_MAX_N = (1u << 24) - 2u * STILE;
constexpr u32 SPK_COUNTED = 0x80000000u; // bit of order[]: this query's blocks may count touches in the accumulator

template <bool COUNTED, int PU /* postings per lane per step */>
__device__ __forceinline__ void sparse_packed_body(const u32 *__restrict__ m_pk, const STerm *__restrict__ qt, const u32 nt, const u32 n_tiles,
                                                   const u32 *__restrict__ tile_dir, const u32 split, const u32 splits, u64 *__restrict__ out, u32 *acc,
                                                   u32 *zflag, u64 (*wpool)[SEL], u64 *sl_b, u32 *sl_n, u32 *sl_w, u32 *st_pre, u32 *st_ctr) {
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    constexpr u32 STEP = 64u * PU;
    for (u32 i = threadIdx.x; i < STILE / 4; i += blockDim.x) reinterpret_cast<uint4 *>(acc)[i] = make_uint4(0u, 0u, 0u, 0u);
    if (!COUNTED)
        for (u32 i = threadIdx.x; i < STILE / 32; i += blockDim.x) zflag[i] = 0u;
    Pool<1> pool;
    pool.clear();
    u64 thr = 0ull;
    auto insert_keys = [&](u64 key) { // the keys of a wave's lanes that beat the pool's last entry
        u64 m = ballot64(key > thr);
        while (m) {
            const int l = __ffsll((long long)m) - 1;
            m &= m - 1;
            const u64 kk = readlane_u64(key, l);
            if (kk > thr) {
                pool.insert_at(kk, pool.rank_of(kk), lane);
                thr = readlane_u64(pool.e[0], SEL - 1);
            }
        }
    };
    // The (tile, term) slices are resolved into the LDS table a WINDOW at a time, all lookups of a window in flight together: TB tiles
    // x all nt terms when nt <= SLICES (one window for the whole block in the usual case), one tile x SLICES terms otherwise.
    const u32 my_tiles = (n_tiles - split + splits - 1) / splits;
    const u32 TW = nt < SLICES ? nt : SLICES;
    const u32 TB = nt < SLICES ? (SLICES / nt < my_tiles ? SLICES / nt : my_tiles) : 1u;
    const u32 dummy4 = (STILE + (u32)lane) * 4u; // byte address of the lane's dummy slot
    char *accb = reinterpret_cast<char *>(acc);
    for (u32 ti0 = 0; ti0 < my_tiles; ti0 += TB) {
        const u32 tbn = my_tiles - ti0 < TB ? my_tiles - ti0 : TB;
        for (u32 tw0 = 0; tw0 < nt; tw0 += TW) {
            const u32 twn = nt - tw0 < TW ? nt - tw0 : TW;
            __syncthreads(); // the previous window is done with the table
            for (u32 p = threadIdx.x; p < tbn * twn; p += blockDim.x) {
                const u32 tile = split + (ti0 + p / twn) * splits, t = tw0 + p % twn;
                const u32 dr = qt[t].dir;
                u64 b = qt[t].begin, e = qt[t].end;
                if (dr != SNO_DIR) {
                    const u32 *row = tile_dir + (u64)dr * (n_tiles + 1);
                    e = b + row[tile + 1];
                    b = b + row[tile];
                }
                sl_b[p] = b;
                sl_n[p] = (u32)(e - b);
                if (p < twn) sl_w[p] = qt[t].qq_k0;
            }
            const u32 groups = (twn + 63u) >> 6;
            for (u32 tj = 0; tj < tbn; tj++) {
                const u32 tile = split + (ti0 + tj) * splits, d0 = tile * STILE;
                const u32 d1 = d0 + 1u; // a posting holds vector id + 1 in its low 24 bits
                for (u32 g = 0; g < groups; g++) {
                    const u32 tb = g << 6, ng = twn - tb < 64u ? twn - tb : 64u;
                    const u32 row0 = tj * twn + tb; // table row of the group's first term
                    __syncthreads(); // the table; the previous group's directory is no longer read; the previous tile's flush
                    if (wave == 0) { // step counts of this group's slices of the tile -> exclusive prefix (one lane per term)
                        u32 ns = 0;
                        if ((u32)lane < ng) ns = (sl_n[row0 + lane] + STEP - 1u) / STEP;
                        u32 incl = ns;
#pragma unroll
                        for (int dd = 1; dd < 64; dd <<= 1) {
                            const u32 o = (u32)__shfl_up((int)incl, dd, 64);
                            if (lane >= dd) incl += o;
                        }
                        if ((u32)lane < ng) st_pre[lane] = incl - ns;
                        if ((u32)lane == ng - 1u) st_pre[ng] = incl;
                        if (lane == 0) *st_ctr = 0u;
                    }
                    __syncthreads();
                    const u32 total = uniform_u32(st_pre[ng]);
                    const u32 my_lo = (u32)lane < ng ? st_pre[lane] : 0xFFFFFFFFu, my_hi = (u32)lane < ng ? st_pre[lane + 1] : 0u;
                    auto pull = [&]() -> u32 {
                        u32 k = 0;
                        if (lane == 0) k = atomicAdd(st_ctr, 1u);
                        return readlane_u32(k, 0);
                    };
                    struct Step { u32 base_lo, base_hi, len, w; bool valid; };
                    auto locate = [&](u32 k) -> Step {
                        Step sp;
                        sp.valid = k < total;
                        sp.base_lo = 0; sp.base_hi = 0; sp.len = 0; sp.w = 0;
                        if (sp.valid) {
                            const u64 m = ballot64(k >= my_lo && k < my_hi);
                            const u32 tl = (u32)(__ffsll((long long)m) - 1);
                            const u32 done = (k - readlane_u32(my_lo, (int)tl)) * STEP, left = sl_n[row0 + tl] - done;
                            const u64 b = sl_b[row0 + tl] + done;
                            // wave-uniform by construction; say so (values that arrive through a vector load count as divergent)
                            sp.base_lo = uniform_u32((u32)b);
                            sp.base_hi = uniform_u32((u32)(b >> 32));
                            sp.len = uniform_u32(left < STEP ? left : STEP);
                            sp.w = uniform_u32(sl_w[tb + tl]);
                        }
                        return sp;
                    };
                    auto fetch_p = [&](const Step &sp, u32 (&pv)[PU]) {
                        const u32 *bp = m_pk + ((u64)sp.base_hi << 32 | sp.base_lo);
                        const auto rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)bp, (short)0, (int)(sp.len * 4u), 0x00020000);
#pragma unroll
                        for (int u = 0; u < PU; u++) pv[u] = (u32)__builtin_amdgcn_raw_buffer_load_b32(rsrc, ((u32)lane + (u32)u * 64u) * 4u, 0, 0);
                    };
                    // a posting: slot = its id relative to the tile (other tiles' postings and the zeros past the step's end wrap to >=
                    // STILE and fall on the lane's dummy slot behind the tile through the min), weight = qq * key
                    auto apply_p = [&](const Step &sp, const u32 (&pv)[PU]) {
                        const u32 qq = sp.w & 255u, k0 = sp.w >> 8;
                        if (COUNTED) {
                            auto one = [&](u32 p, bool keyed) {
                                const u32 rel = p - d1;
                                // (rel & 0xFFFFFF) * 4, the slot's byte address, and qq * (rel >> 24) as ONE full-rate instruction each
                                // (left alone the compiler shifts and masks for the first and picks the quarter-rate v_mul_lo_u32 for the second)
                                u32 at, w;
                                asm("v_mul_u32_u24 %0, %1, 4" : "=v"(at) : "v"(rel));
                                asm("v_mul_u32_u24_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3 src1_sel:DWORD" : "=v"(w) : "v"(rel), "s"(qq));
                                if (keyed) at = (rel >> 24) >= k0 ? at : dummy4;
                                at = min(at, dummy4);
                                atomicAdd(reinterpret_cast<u32 *>(accb + at), w | SPK_CNT);
                            };
                            if (sp.len == STEP) { // a full step: no tests at all
                                if (k0 == 0u) {
#pragma unroll
                                    for (int u = 0; u < PU; u++) one(pv[u], false);
                                } else {
#pragma unroll
                                    for (int u = 0; u < PU; u++) one(pv[u], true);
                                }
                            } else {
#pragma unroll
                                for (int u = 0; u < PU; u++)
                                    if ((u32)u * 64u < sp.len) one(pv[u], true);
                                // every path consumes the whole register set: a load still in flight into a register the next
                                // iteration reuses would make the loop head wait for ALL loads, the prefetched step's included
                                asm volatile("" ::"v"(pv[PU - 1]));
                            }
                        } else {
                            bool zero_any = false;
#pragma unroll
                            for (int u = 0; u < PU; u++) {
                                const u32 rel = pv[u] - d1, key = rel >> 24, slot = rel & 0xFFFFFFu;
                                const bool ok = slot < STILE && key >= k0;
                                const u32 w = __umul24(qq, key);
                                atomicAdd(reinterpret_cast<u32 *>(accb + (ok ? slot * 4u : dummy4)), ok ? w : 0u);
                                zero_any |= ok && w == 0u;
                            }
                            if (__any(zero_any)) {
#pragma unroll
                                for (int u = 0; u < PU; u++) {
                                    const u32 rel = pv[u] - d1, key = rel >> 24, slot = rel & 0xFFFFFFu;
                                    if (slot < STILE && key >= k0 && qq * key == 0u) atomicOr(&zflag[slot >> 5], 1u << (slot & 31u));
                                }
                            }
                        }
                    };
                    // Two steps per iteration, no exit in the middle: a step behind the last one has length 0 — eight out-of-range loads,
                    // no memory traffic, nothing applied.  Every fetch is issued and every register set consumed on every path, so the
                    // wait for the older step's postings always leaves exactly the newer step's eight loads in flight (a load that
                    // some path leaves pending makes the loop head wait for ALL loads, the prefetched step's included).
                    u32 pa[PU], pb[PU];
                    Step sa = locate(pull()), sb;
                    fetch_p(sa, pa);
                    while (sa.valid) { // ping-pong between the two register sets
                        sb = locate(pull());
                        fetch_p(sb, pb);
                        apply_p(sa, pa);
                        sa = locate(pull());
                        fetch_p(sa, pa);
                        apply_p(sb, pb);
                    }
                }
                if (tw0 + TW < nt) continue; // more term windows of this tile to come (TB == 1 then)
                __syncthreads();             // every wave's adds to the tile are done
                // flush: wave w scans slots [w * 2048, (w + 1) * 2048) of the tile into its pool and clears them
                if (COUNTED) {
                    for (u32 s0 = (u32)wave * (STILE / 4); s0 < (u32)(wave + 1) * (STILE / 4); s0 += 256) {
                        const u32 slot = s0 + (u32)lane * 4u;
                        const uint4 a4 = *reinterpret_cast<const uint4 *>(&acc[slot]);
                        *reinterpret_cast<uint4 *>(&acc[slot]) = make_uint4(0u, 0u, 0u, 0u);
                        const u32 thr_hi = (u32)(thr >> 32);
                        const u32 a[4] = {a4.x, a4.y, a4.z, a4.w};
                        bool c[4];
#pragma unroll
                        for (int j = 0; j < 4; j++) c[j] = a[j] != 0u && (a[j] & SPK_SUM) + 1u >= thr_hi;
                        if (__any(c[0] | c[1] | c[2] | c[3])) {
#pragma unroll
                            for (int j = 0; j < 4; j++) insert_keys(c[j] ? ((u64)((a[j] & SPK_SUM) + 1u) << 32 | (u64)(d0 + slot + (u32)j)) : 0ull);
                        }
                    }
                } else {
                    for (u32 s0 = (u32)wave * (STILE / 4); s0 < (u32)(wave + 1) * (STILE / 4); s0 += 64) {
                        const u32 slot = s0 + (u32)lane;
                        const u32 a = acc[slot];
                        const u32 fw = zflag[slot >> 5];
                        acc[slot] = 0u;
                        const bool reached = a != 0u || ((fw >> (slot & 31u)) & 1u);
                        insert_keys(reached ? (((u64)a + 1ull) << 32 | (u64)(d0 + slot)) : 0ull);
                    }
                    zflag[(u32)wave * (STILE / 128) + (u32)lane] = 0u; // the flag words of this wave's own slots (nobody else reads them)
                }
                // (the next group's barrier orders the cleared slots before the next tile's first add)
            }
        }
    }
    wpool[wave][lane] = pool.e[0];
    __syncthreads();
    if (wave == 0) {
        for (int w = 1; w < 4; w++) insert_keys(wpool[w][lane]);
        out[lane] = pool.e[0];
    }
}

// grid / blocks / pools as sparse_tile_kernel; order[i] = query | SPK_COUNTED
template <int PU>
__global__ __launch_bounds__(256) void sparse_packed_kernel(const u32 *__restrict__ m_pk, const STerm *__restrict__ terms, const u32 *__restrict__ qt_off, u32 n,
                                                            const u32 *__restrict__ tile_dir, const u32 *__restrict__ order, u32 splits,
                                                            u64 *__restrict__ part /*[B][splits][64]*/) {
    __shared__ __attribute__((aligned(16))) u32 acc[STILE + 64];
    __shared__ u32 zflag[STILE / 32];
    __shared__ u64 wpool[4][SEL];
    __shared__ u64 sl_b[SLICES];
    __shared__ u32 sl_n[SLICES], sl_w[SLICES];
    __shared__ u32 st_pre[65];
    __shared__ u32 st_ctr;
    const u32 oq = order[blockIdx.x / splits];
    const u32 q = oq & ~SPK_COUNTED;
    const u32 split = blockIdx.x % splits;
    const u32 t0 = qt_off[q], nt = qt_off[q + 1] - t0;
    const u32 n_tiles = (n + STILE - 1) / STILE;
    u64 *out = part + ((u64)q * splits + split) * SEL;
    if (nt == 0 || split >= n_tiles) { // nothing to visit: an empty pool (the finish kernel reads every split)
        if (threadIdx.x < SEL) out[threadIdx.x] = 0ull;
        return;
    }
    if (oq & SPK_COUNTED) sparse_packed_body<true, PU>(m_pk, terms + t0, nt, n_tiles, tile_dir, split, splits, out, acc, zflag, wpool, sl_b, sl_n, sl_w, st_pre, &st_ctr);
    else sparse_packed_body<false, PU>(m_pk, terms + t0, nt, n_tiles, tile_dir, split, splits, out, acc, zflag, wpool, sl_b, sl_n, sl_w, st_pre, &st_ctr);
}

// one wave per query: merge the segment pools, optional raw-value rerank, write the top k
__global__ __launch_bounds__(64) void sparse_finish_kernel(const SparseDev ix, const u64 *__restrict__ part, u32 S, const u32 *__restrict__ q_dims,
                                                           const float *__restrict__ q_vals, const u32 *__restrict__ q_off, u32 top_k, u32 k_with_reranking,
                                                           int rerank, u32 *__restrict__ out_ids, float *__restrict__ out_scores, u32 *__restrict__ out_counts) {
    const int lane = threadIdx.x;
    const u32 q = blockIdx.x;
    Pool<1> pool;
    pool.clear();
    u64 thr = 0ull;
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
    const u32 have = (u32)__popcll(__ballot(pool.e[0] != 0ull));
    const u32 ncand = have < k_with_reranking ? have : k_with_reranking; // select_nth + truncate(k * reranking_factor)
    const u64 mine = pool.e[0];
    if (!rerank) {
        const u32 nout = ncand < top_k ? ncand : top_k;
        if ((u32)lane < nout) {
            out_ids[(u64)q * top_k + lane] = (u32)mine;
            out_scores[(u64)q * top_k + lane] = (float)((u32)(mine >> 32) - 1u); // `similarity as f32`
        }
        if (lane == 0) out_counts[q] = nout;
        return;
    }
    // finalize_sparse_ann_results: dp over the QUERY pairs in order, f32 multiply then add; sort by total_cmp descending
    u64 res[1] = {0ull};
    if ((u32)lane < ncand) {
        const u32 v = (u32)mine;
        const u64 b = ix.row_off[v], e = ix.row_off[v + 1];
        float dp = 0.0f;
        for (u32 i = q_off[q]; i < q_off[q + 1]; i++) {
            const u32 d = q_dims[i];
            u64 lo = b, hi = e;
            while (lo < hi) { const u64 mid = lo + (hi - lo) / 2; if (ix.raw_dims[mid] < d) lo = mid + 1; else hi = mid; }
            if (lo < e && ix.raw_dims[lo] == d) dp = __fadd_rn(dp, __fmul_rn(ix.raw_vals[lo], q_vals[i]));
        }
        res[0] = pack_key(simkey(dp), v);
    }
    bitonic_sort_desc<1>(res, lane);
    const u32 nout = ncand < top_k ? ncand : top_k;
    if ((u32)lane < nout) {
        out_ids[(u64)q * top_k + lane] = (u32)res[0];
        out_scores[(u64)q * top_k + lane] = simkey_inv((u32)(res[0] >> 32));
    }
    if (lane == 0) out_counts[q] = nout;
}

} // namespace

struct cos_sparse {
    int32_t device = 0;
    u32 bits = 0, T = 0, n = 0, n_tiles = 0;
    float upper = 1.0f;
    bool have_raw = false;
    // host side of a query's preparation: find_node, the early-termination rule, posting counts
    std::vector<u32> h_dims;    // [T] ascending
    std::vector<u64> h_key_off; // [T][Q + 1] (the caller's CSR offsets: list begin/end and the postings a term visits from key k0 on)
    std::vector<u32> h_dir;     // [T] row in the tile directory or SNO_DIR
    std::vector<u32> h_mult;    // [T] how often one vector id occurs in the dimension's list at most (1 unless the caller's CSR repeats ids)
    bool packed = false;        // device layout: one u32 per posting (d_pk) instead of d_ids + d_keys
    // device: one id-sorted list per dimension (same offsets as the caller's CSR: list t = [key_off[t][0], key_off[t][Q]))
    u32 *d_ids = nullptr;
    uint8_t *d_keys = nullptr;
    u32 *d_pk = nullptr; // packed layout: key << 24 | (vector id + 1)
    u32 *d_tile_dir = nullptr; // [rows][n_tiles + 1], offsets relative to the list's begin
    u32 *d_raw_dims = nullptr;
    u64 *d_row_off = nullptr;
    float *d_raw_vals = nullptr;
    // grow-only workspace of cos_sparse_search_batch (no allocation on the query path once warm); `mu` serialises callers
    std::mutex mu;
    struct Buf { void *p = nullptr; size_t cap = 0; } w_qd, w_qv, w_qo, w_terms, w_qt_off, w_order, w_part, w_oi, w_os, w_oc;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    cos_sparse_stats last{};
};

struct SparseView { void *p; template <typename T> T *as() const { return (T *)p; } };

static hipError_t sparse_grow(cos_sparse::Buf &b, size_t need) {
    if (need <= b.cap) return hipSuccess;
    if (b.p) (void)hipFree(b.p);
    b.p = nullptr;
    b.cap = 0;
    hipError_t e = hipMalloc(&b.p, need);
    if (e == hipSuccess) b.cap = need;
    return e;
}

extern "C" int32_t cos_sparse_destroy(cos_sparse *s) {
    if (!s) return COS_OK;
    (void)hipSetDevice(s->device);
    void *ptrs[] = {s->d_ids, s->d_keys, s->d_pk, s->d_tile_dir, s->d_raw_dims, s->d_row_off, s->d_raw_vals, s->w_qd.p, s->w_qv.p, s->w_qo.p,
                    s->w_terms.p, s->w_qt_off.p, s->w_order.p, s->w_part.p, s->w_oi.p, s->w_os.p, s->w_oc.p};
    for (void *p : ptrs) if (p) (void)hipFree(p);
    if (s->ev0) (void)hipEventDestroy(s->ev0);
    if (s->ev1) (void)hipEventDestroy(s->ev1);
    delete s;
    return COS_OK;
}

extern "C" int32_t cos_sparse_create(int32_t device, uint32_t quantization_bits, float values_upper_bound, const uint32_t *dims, uint32_t n_dims,
                                     const uint64_t *key_offsets, const uint32_t *vec_ids, uint32_t n_vectors, const uint64_t *row_offsets,
                                     const uint32_t *raw_dims, const float *raw_vals, cos_sparse **out) {
    if (!dims || !key_offsets || !vec_ids || !out || n_dims == 0 || n_vectors == 0) return cos_fail(COS_ERR_INVALID, "bad argument");
    *out = nullptr;
    if (quantization_bits < 1 || quantization_bits > 8) return cos_fail(COS_ERR_INVALID, "quantization_bits must be in [1, 8] (keys are u8)");
    const u32 Q = 1u << quantization_bits;
    for (u32 t = 0; t < n_dims; t++) {
        if (t && dims[t] <= dims[t - 1]) return cos_fail(COS_ERR_INVALID, "dimension indices must be strictly ascending");
        for (u32 k = 0; k < Q; k++)
            if (key_offsets[(size_t)t * (Q + 1) + k] > key_offsets[(size_t)t * (Q + 1) + k + 1]) return cos_fail(COS_ERR_INVALID, "key offsets of dimension %u decrease", dims[t]);
        if (t && key_offsets[(size_t)t * (Q + 1)] != key_offsets[(size_t)(t - 1) * (Q + 1) + Q]) return cos_fail(COS_ERR_INVALID, "posting ranges of consecutive dimensions must be contiguous");
    }
    if (key_offsets[0] != 0) return cos_fail(COS_ERR_INVALID, "the first posting list must start at offset 0");
    const u64 nnz = key_offsets[(size_t)(n_dims - 1) * (Q + 1) + Q];
    for (u64 p = 0; p < nnz; p++)
        if (vec_ids[p] >= n_vectors) return cos_fail(COS_ERR_INVALID, "posting %llu names vector %u of %u", (unsigned long long)p, vec_ids[p], n_vectors);
    if ((row_offsets != nullptr) != (raw_dims != nullptr) || (raw_dims != nullptr) != (raw_vals != nullptr)) return cos_fail(COS_ERR_INVALID, "raw CSR: all three arrays or none");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return cos_fail(COS_ERR_NO_DEVICE, "no HIP device visible; the GPU path has no CPU fallback");
    HIP_TRY(hipSetDevice(device));
    cos_sparse *s = new cos_sparse();
    s->device = device; s->bits = quantization_bits; s->T = n_dims; s->n = n_vectors; s->upper = values_upper_bound;
    s->n_tiles = (n_vectors + STILE - 1) / STILE;
    s->h_dims.assign(dims, dims + n_dims);
    s->h_key_off.assign(key_offsets, key_offsets + (size_t)n_dims * (Q + 1));
    // device layout: per dimension the Q key lists merged into one list sorted by vector id ((id, key) pairs; a stable order among
    // equal ids is irrelevant, the sums commute), and a tile directory for the long lists
    constexpr size_t PAD = 64 * SPU; // one step of padding behind the last posting: the kernel's loads are never clamped
    std::vector<u32> m_ids((size_t)nnz + PAD, 0u);
    std::vector<uint8_t> m_keys((size_t)nnz + PAD, 0);
    std::vector<u32> tile_dir;
    s->h_dir.assign(n_dims, SNO_DIR);
    s->h_mult.assign(n_dims, 1u);
    // the one-word-per-posting layout and its kernel (sparse_packed_kernel) wherever vector ids fit 24 bits — measured in round 5:
    // 0.453 ms against 0.540 ms per 256-query batch, 256 / 256 queries identical to the oracle (profiles/r05_candidates_sparse.txt);
    // tuning knob sparse_layout = 0 keeps the (u32 id, u8 key) layout (the parity tests run both)
    s->packed = tune_or(TUNE_SPARSE_LAYOUT, 1) != 0 && n_vectors <= SPK_MAX_N;
    std::vector<u64> tmp;
    u32 rows = 0;
    const u32 nt1 = s->n_tiles + 1;
    for (u32 t = 0; t < n_dims; t++) {
        const uint64_t *ko = key_offsets + (size_t)t * (Q + 1);
        const u64 b = ko[0], e = ko[Q];
        tmp.clear();
        for (u32 k = 0; k < Q; k++)
            for (u64 p = ko[k]; p < ko[k + 1]; p++) tmp.push_back((u64)vec_ids[p] << 8 | k);
        std::sort(tmp.begin(), tmp.end());
        for (u64 p = b; p < e; p++) { m_ids[p] = (u32)(tmp[p - b] >> 8); m_keys[p] = (uint8_t)(tmp[p - b] & 255u); }
        u32 run = 0, mult = 1;
        for (u64 p = b; p < e; p++) {
            run = p > b && m_ids[p] == m_ids[p - 1] ? run + 1 : 1;
            mult = std::max(mult, run);
        }
        s->h_mult[t] = mult;
        if (e - b > SDIR_MIN) {
            if (e - b > 0xFFFFFFFFull) { cos_sparse_destroy(s); return cos_fail(COS_ERR_UNIMPLEMENTED, "dimension %u holds more than 2^32 postings", dims[t]); }
            s->h_dir[t] = rows++;
            tile_dir.resize((size_t)rows * nt1);
            u32 *row = tile_dir.data() + (size_t)(rows - 1) * nt1;
            u64 p = b;
            for (u32 tile = 0; tile <= s->n_tiles; tile++) {
                const u64 first = (u64)tile * STILE;
                while (p < e && m_ids[p] < first) p++;
                row[tile] = (u32)(p - b);
            }
        }
    }
    auto up = [&](void **dst, const void *src, size_t bytes) -> hipError_t {
        hipError_t e = hipMalloc(dst, bytes ? bytes : 1);
        return e == hipSuccess && bytes ? hipMemcpy(*dst, src, bytes, hipMemcpyHostToDevice) : e;
    };
    hipError_t e = hipSuccess;
    if (s->packed) {
        std::vector<u32> m_pk((size_t)nnz + 1, 0u);
        for (u64 p = 0; p < nnz; p++) m_pk[p] = (u32)m_keys[p] << 24 | (m_ids[p] + 1u);
        e = up((void **)&s->d_pk, m_pk.data(), m_pk.size() * 4);
    } else {
        e = up((void **)&s->d_ids, m_ids.data(), m_ids.size() * 4);
        if (e == hipSuccess) e = up((void **)&s->d_keys, m_keys.data(), m_keys.size());
    }
    if (e == hipSuccess) e = up((void **)&s->d_tile_dir, tile_dir.data(), tile_dir.size() * 4);
    if (e == hipSuccess && row_offsets) {
        const u64 rnnz = row_offsets[n_vectors];
        e = up((void **)&s->d_row_off, row_offsets, ((size_t)n_vectors + 1) * 8);
        if (e == hipSuccess) e = up((void **)&s->d_raw_dims, raw_dims, (size_t)rnnz * 4);
        if (e == hipSuccess) e = up((void **)&s->d_raw_vals, raw_vals, (size_t)rnnz * 4);
        s->have_raw = true;
    }
    if (e == hipSuccess) e = hipEventCreate(&s->ev0);
    if (e == hipSuccess) e = hipEventCreate(&s->ev1);
    if (e != hipSuccess) { cos_sparse_destroy(s); HIP_TRY(e); }
    *out = s;
    return COS_OK;
}

// InvertedIndexNode::quantize on the host (the same f32 operations as sparse_quantize above; this file is built with -ffp-contract=off)
static inline uint32_t host_sparse_quantize(float value, float upper, uint32_t bits) {
    const uint32_t quantization = (1u << bits) - 1u;
    const float max_val = (float)quantization;
    float t = (value / upper) * max_val;
    t = t < 0.0f ? 0.0f : (t > max_val ? max_val : t); // f32::clamp keeps NaN
    const uint32_t q = !(t == t) || t <= 0.0f ? 0u : (t >= 255.0f ? 255u : (uint32_t)(int)t); // `as u8`
    return q < quantization ? q : quantization;
}

// InvertedIndex::insert for a whole collection (indexes/inverted/mod.rs + models/inverted_index.rs:176-200): the vectors are taken in
// id order and every (dimension, value) pair pushes the id to the END of the list of (dimension, quantize(value)) — so the CSR this
// produces is what the host's tree holds after inserting ids 0 .. n-1.  Host code, no device.
extern "C" int32_t cos_sparse_build_csr(uint32_t quantization_bits, float values_upper_bound, uint32_t n_vectors, const uint64_t *row_offsets,
                                        const uint32_t *raw_dims, const float *raw_vals, uint32_t *out_dims, uint64_t *out_key_offsets,
                                        uint32_t *out_vec_ids, uint32_t *n_dims) {
    if (!row_offsets || !raw_dims || !raw_vals || !n_dims || n_vectors == 0) return cos_fail(COS_ERR_INVALID, "bad argument");
    if (quantization_bits < 1 || quantization_bits > 8) return cos_fail(COS_ERR_INVALID, "quantization_bits must be in [1, 8] (keys are u8)");
    const u32 Q = 1u << quantization_bits;
    const u64 nnz = row_offsets[n_vectors];
    for (u32 v = 0; v < n_vectors; v++)
        if (row_offsets[v + 1] < row_offsets[v]) return cos_fail(COS_ERR_INVALID, "row offsets decrease at vector %u", v);
    std::vector<u32> dims(raw_dims, raw_dims + nnz);
    std::sort(dims.begin(), dims.end());
    dims.erase(std::unique(dims.begin(), dims.end()), dims.end());
    const u32 T = (u32)dims.size();
    const bool sizes_only = !out_dims && !out_key_offsets && !out_vec_ids;
    if (sizes_only) { *n_dims = T; return COS_OK; }
    if (!out_dims || !out_key_offsets || !out_vec_ids) return cos_fail(COS_ERR_INVALID, "all three output arrays or none");
    if (*n_dims < T) { *n_dims = T; return cos_fail(COS_ERR_INVALID, "%u distinct dimensions, room for fewer", T); }
    *n_dims = T;
    std::vector<u64> count((size_t)T * Q, 0);
    std::vector<u32> slot(nnz);
    for (u64 p = 0; p < nnz; p++) {
        const u32 t = (u32)(std::lower_bound(dims.begin(), dims.end(), raw_dims[p]) - dims.begin());
        slot[p] = t * Q + host_sparse_quantize(raw_vals[p], values_upper_bound, quantization_bits);
        count[slot[p]]++;
    }
    std::vector<u64> cursor((size_t)T * Q);
    u64 run = 0;
    for (u32 t = 0; t < T; t++) {
        for (u32 k = 0; k < Q; k++) {
            out_key_offsets[(size_t)t * (Q + 1) + k] = run;
            cursor[(size_t)t * Q + k] = run;
            run += count[(size_t)t * Q + k];
        }
        out_key_offsets[(size_t)t * (Q + 1) + Q] = run;
        out_dims[t] = dims[t];
    }
    for (u32 v = 0; v < n_vectors; v++) // id order = push order
        for (u64 p = row_offsets[v]; p < row_offsets[v + 1]; p++) out_vec_ids[cursor[slot[p]]++] = v;
    return COS_OK;
}

// cos_sparse_build_csr + cos_sparse_create in one call; keep_raw != 0 also uploads the raw vectors for the raw-value rerank
// (their dims must then ascend within a row, finalize_sparse_ann_results' lookup is a binary search)
extern "C" int32_t cos_sparse_create_from_vectors(int32_t device, uint32_t quantization_bits, float values_upper_bound, uint32_t n_vectors,
                                                  const uint64_t *row_offsets, const uint32_t *raw_dims, const float *raw_vals, int32_t keep_raw,
                                                  cos_sparse **out) {
    if (!out) return cos_fail(COS_ERR_INVALID, "null argument");
    *out = nullptr;
    u32 T = 0;
    int32_t rc = cos_sparse_build_csr(quantization_bits, values_upper_bound, n_vectors, row_offsets, raw_dims, raw_vals, nullptr, nullptr, nullptr, &T);
    if (rc) return rc;
    const u32 Q = 1u << quantization_bits;
    std::vector<u32> dims(std::max(T, 1u)), ids((size_t)std::max<u64>(row_offsets[n_vectors], 1));
    std::vector<uint64_t> ko((size_t)std::max(T, 1u) * (Q + 1));
    rc = cos_sparse_build_csr(quantization_bits, values_upper_bound, n_vectors, row_offsets, raw_dims, raw_vals, dims.data(), ko.data(), ids.data(), &T);
    if (rc) return rc;
    if (T == 0) return cos_fail(COS_ERR_INVALID, "no postings");
    return cos_sparse_create(device, quantization_bits, values_upper_bound, dims.data(), T, ko.data(), ids.data(), n_vectors,
                             keep_raw ? row_offsets : nullptr, keep_raw ? raw_dims : nullptr, keep_raw ? raw_vals : nullptr, out);
}

// Rust `as u8` / `as u32` on f32 (saturating, NaN -> 0), host side
static inline uint32_t host_f32_as_u8(float v) { return !(v == v) || v <= 0.0f ? 0u : (v >= 255.0f ? 255u : (uint32_t)(int)v); }
static inline uint32_t host_f32_as_u32(float v) { return !(v == v) || v <= 0.0f ? 0u : (v >= 4294967296.0f ? 0xFFFFFFFFu : (uint32_t)v); }

extern "C" int32_t cos_sparse_search_batch(cos_sparse *s, const uint32_t *q_dims, const float *q_vals, const uint32_t *q_offsets, uint32_t B, uint32_t top_k,
                                           float early_terminate_threshold, uint32_t reranking_factor, uint32_t *out_ids, float *out_scores,
                                           uint32_t *out_counts) {
    if (!s || !q_dims || !q_vals || !q_offsets || !out_ids || !out_scores || !out_counts || B == 0 || top_k == 0) return cos_fail(COS_ERR_INVALID, "bad argument");
    if (B >= SPK_COUNTED) return cos_fail(COS_ERR_INVALID, "batch of %u queries", B);
    const bool rerank = reranking_factor != 0;
    if (rerank && !s->have_raw) return cos_fail(COS_ERR_NOT_READY, "raw-value rerank needs the raw sparse vectors (cos_sparse_create row_offsets / raw_dims / raw_vals)");
    const u32 kwr = top_k * (rerank ? reranking_factor : 1u);
    if (kwr > SEL) return cos_fail(COS_ERR_UNIMPLEMENTED, "top_k x reranking_factor must be <= %u", SEL);
    HIP_TRY(hipSetDevice(s->device));
    for (u32 b = 0; b < B; b++)
        if (q_offsets[b + 1] < q_offsets[b]) return cos_fail(COS_ERR_INVALID, "query offsets decrease");
    const u32 nq = q_offsets[B], Q = 1u << s->bits;
    // ---- resolve the query terms on the host (sparse_ann_query.rs:80-125): find_node, quantize, which keys the term visits ----
    const float qf = (float)Q;
    float etv = qf * early_terminate_threshold;
    etv = etv > 255.0f ? 255.0f : etv;
    const u32 early_terminate_value = host_f32_as_u8(etv), low_threshold = host_f32_as_u32(early_terminate_threshold * qf);
    std::vector<STerm> terms;
    terms.reserve(nq);
    std::vector<u32> qt_off(B + 1, 0), order(B);
    std::vector<u64> weight(B, 0);
    std::vector<uint8_t> counted(B, 0);
    u64 visited = 0;
    for (u32 b = 0; b < B; b++) {
        u64 sum_bound = 0, touch_bound = 0; // packed layout: may this query's blocks count touches next to the sum (sparse_packed_body)?
        for (u32 i = q_offsets[b]; i < q_offsets[b + 1]; i++) {
            auto it = std::lower_bound(s->h_dims.begin(), s->h_dims.end(), q_dims[i]);
            if (it == s->h_dims.end() || *it != q_dims[i]) continue;
            const u32 t = (u32)(it - s->h_dims.begin());
            const u32 qq = host_sparse_quantize(q_vals[i], s->upper, s->bits);
            const u32 k0 = qq > low_threshold ? 0u : early_terminate_value;
            if (k0 >= Q) continue;
            const u64 *ko = s->h_key_off.data() + (size_t)t * (Q + 1);
            if (ko[Q] == ko[0]) continue;
            terms.push_back(STerm{ko[0], ko[Q], s->h_dir[t], qq | k0 << 8});
            weight[b] += ko[Q] - ko[0];
            visited += ko[Q] - ko[k0];
            sum_bound += (u64)s->h_mult[t] * qq * (Q - 1u);
            touch_bound += s->h_mult[t];
        }
        counted[b] = sum_bound < SPK_CNT && touch_bound <= 1023u;
        qt_off[b + 1] = (u32)terms.size();
    }
    for (u32 b = 0; b < B; b++) order[b] = b;
    std::stable_sort(order.begin(), order.end(), [&](u32 a, u32 c) { return weight[a] > weight[c]; }); // heaviest query first
    if (s->packed)
        for (u32 b = 0; b < B; b++)
            if (counted[order[b]]) order[b] |= SPK_COUNTED;
    // blocks: enough to fill the chip several times over, at most one per tile
    const u32 splits = std::max<u32>(1u, std::min<u32>(s->n_tiles, (4096u + B - 1) / B));
    std::lock_guard<std::mutex> guard(s->mu);
    HIP_TRY(sparse_grow(s->w_qd, (size_t)std::max(nq, 1u) * 4));
    HIP_TRY(sparse_grow(s->w_qv, (size_t)std::max(nq, 1u) * 4));
    HIP_TRY(sparse_grow(s->w_qo, ((size_t)B + 1) * 4));
    HIP_TRY(sparse_grow(s->w_terms, std::max<size_t>(terms.size(), 1) * sizeof(STerm)));
    HIP_TRY(sparse_grow(s->w_qt_off, ((size_t)B + 1) * 4));
    HIP_TRY(sparse_grow(s->w_order, (size_t)B * 4));
    HIP_TRY(sparse_grow(s->w_part, (size_t)B * splits * SEL * 8));
    HIP_TRY(sparse_grow(s->w_oi, (size_t)B * top_k * 4));
    HIP_TRY(sparse_grow(s->w_os, (size_t)B * top_k * 4));
    HIP_TRY(sparse_grow(s->w_oc, (size_t)B * 4));
    const SparseView d_qd{s->w_qd.p}, d_qv{s->w_qv.p}, d_qo{s->w_qo.p}, d_terms{s->w_terms.p}, d_qt_off{s->w_qt_off.p}, d_order{s->w_order.p},
        d_part{s->w_part.p}, d_oi{s->w_oi.p}, d_os{s->w_os.p}, d_oc{s->w_oc.p};
    HIP_TRY(hipMemcpy(d_qd.p, q_dims, (size_t)nq * 4, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(d_qv.p, q_vals, (size_t)nq * 4, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(d_qo.p, q_offsets, ((size_t)B + 1) * 4, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(d_terms.p, terms.data(), terms.size() * sizeof(STerm), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(d_qt_off.p, qt_off.data(), ((size_t)B + 1) * 4, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(d_order.p, order.data(), (size_t)B * 4, hipMemcpyHostToDevice));
    SparseDev dev{nullptr, nullptr, nullptr, s->d_row_off, s->d_raw_dims, s->d_raw_vals, s->T, Q, s->n, s->bits, s->upper};
    HIP_TRY(hipEventRecord(s->ev0, 0));
    // eight postings per lane and step; sixteen measured the same (0.447 against 0.453 ms) and was dropped
    if (s->packed)
        hipLaunchKernelGGL(sparse_packed_kernel<8>, dim3(B * splits), dim3(256), 0, 0, s->d_pk, d_terms.as<STerm>(), d_qt_off.as<u32>(), s->n, s->d_tile_dir,
                           d_order.as<u32>(), splits, d_part.as<u64>());
    else
        hipLaunchKernelGGL(sparse_tile_kernel, dim3(B * splits), dim3(256), 0, 0, s->d_ids, s->d_keys, d_terms.as<STerm>(), d_qt_off.as<u32>(), s->n, s->d_tile_dir,
                           d_order.as<u32>(), splits, d_part.as<u64>());
    HIP_TRY(hipGetLastError());
    hipLaunchKernelGGL(sparse_finish_kernel, dim3(B), dim3(64), 0, 0, dev, d_part.as<u64>(), splits, d_qd.as<u32>(), d_qv.as<float>(), d_qo.as<u32>(), top_k, kwr,
                       rerank ? 1 : 0, d_oi.as<u32>(), d_os.as<float>(), d_oc.as<u32>());
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(s->ev1, 0));
    HIP_TRY(hipMemcpy(out_ids, d_oi.p, (size_t)B * top_k * 4, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(out_scores, d_os.p, (size_t)B * top_k * 4, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(out_counts, d_oc.p, (size_t)B * 4, hipMemcpyDeviceToHost));
    float ms = 0.0f;
    HIP_TRY(hipEventElapsedTime(&ms, s->ev0, s->ev1));
    s->last.kernel_ms = ms;
    s->last.postings_visited = visited;
    s->last.posting_bytes = visited * 4;
    s->last.blocks = B * splits;
    return COS_OK;
}

extern "C" int32_t cos_sparse_layout(cos_sparse *s, uint32_t *packed) {
    if (!s || !packed) return cos_fail(COS_ERR_INVALID, "null argument");
    *packed = s->packed ? 1u : 0u;
    return COS_OK;
}

extern "C" int32_t cos_sparse_last_stats(cos_sparse *s, cos_sparse_stats *out) {
    if (!s || !out) return cos_fail(COS_ERR_INVALID, "null argument");
    std::lock_guard<std::mutex> guard(s->mu);
    *out = s->last;
    return COS_OK;
}
