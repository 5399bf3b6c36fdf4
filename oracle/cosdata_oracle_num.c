/*
 * cosdata_oracle_num.c — numeric kernels of the oracle (TEST INFRASTRUCTURE, see cosdata_oracle.h).
 *
 * Build flags matter: -ffp-contract=off (Rust never contracts a*b+c), -mavx2 -mfma -mf16c -mpopcnt
 * (the reference's runtime dispatch takes the AVX2/FMA path on any x86-64 server CPU,
 * src/models/dot_product.rs:92-157).
 */
#include "cosdata_oracle.h"

#include <immintrin.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------------
 * Rust `as` casts (saturating, NaN -> 0)
 * ---------------------------------------------------------------------------------------- */
static inline uint8_t rust_f32_as_u8(float v) {
    if (!(v == v)) return 0;
    if (v <= 0.0f) return 0;
    if (v >= 255.0f) return 255;
    return (uint8_t)v; /* truncation toward zero */
}
static inline uint64_t rust_f32_as_usize(float v) {
    if (!(v == v)) return 0;
    if (v <= 0.0f) return 0;
    if (v >= 18446744073709551616.0f) return UINT64_MAX;
    return (uint64_t)v;
}
/* Rust f32::max / f32::min: if one operand is NaN the other is returned. */
static inline float rust_max(float a, float b) { return (a != a) ? b : ((b != b) ? a : (a > b ? a : b)); }
static inline float rust_min(float a, float b) { return (a != a) ? b : ((b != b) ? a : (a < b ? a : b)); }

/* sqrt(Σ x*x) with a sequential, non-fused multiply-then-add — `iter().map(|x| x*x).sum::<f32>().sqrt()`
 * (scalar.rs:31,41,45; vector_store.rs:414,426).  <f32 as Sum> folds from -0.0 on rustc 1.85
 * (rust-toolchain.toml), which is the additive identity, so the first term enters unchanged. */
float coso_seq_norm_f32(const float *x, int n) {
    volatile float acc = -0.0f; /* volatile: forbid re-association / vectorised reduction */
    for (int i = 0; i < n; i++) {
        float p = x[i] * x[i];
        acc = acc + p;
    }
    return sqrtf(acc);
}

uint16_t coso_f32_to_f16(float x) { return (uint16_t)_cvtss_sh(x, _MM_FROUND_TO_NEAREST_INT); } /* half::f16::from_f32, RNE */
float coso_f16_to_f32(uint16_t h) { return _cvtsh_ss(h); }

size_t coso_code_bytes(int storage, int resolution, int dim) {
    switch (storage) {
    case COSO_STORAGE_U8: return (size_t)dim;
    case COSO_STORAGE_SUBBYTE: return (size_t)resolution * (size_t)((dim + 7) / 8);
    case COSO_STORAGE_F16: return (size_t)dim * 2;
    case COSO_STORAGE_F32: return (size_t)dim * 4;
    default: return 0;
    }
}

/* src/models/common.rs:226-275 — quantize_to_u8_bits / to_float_flag.
 * level n = floor((x+1)/step) as usize; plane p receives bit (n >> (res-1-p)) & 1 (MSB FIRST);
 * dim i -> bit i%8 of byte i/8.  values_range is ignored (hard-wired [-1,1)). */
static void quantize_subbyte(const float *x, int dim, int res, uint8_t *planes) {
    const int pb = (dim + 7) / 8;
    const uint64_t parts = 1ull << res;
    const float step = 2.0f / (float)parts;
    memset(planes, 0, (size_t)res * pb);
    for (int i = 0; i < dim; i++) {
        uint64_t n = rust_f32_as_usize(floorf((x[i] + 1.0f) / step));
        for (int p = res - 1; p >= 0; p--) { /* fill from least significant to most significant */
            if (n & 1) planes[(size_t)p * pb + (i >> 3)] |= (uint8_t)(1u << (i & 7));
            n >>= 1;
        }
    }
}

/* src/quantization/scalar.rs:10-52 */
int coso_quantize(const float *x, int dim, int storage, int resolution, float lo, float hi, void *code,
                  float *mag) {
    switch (storage) {
    case COSO_STORAGE_U8: {
        uint8_t *q = (uint8_t *)code;
        uint32_t s = 0;
        for (int i = 0; i < dim; i++) {
            float c = rust_min(rust_max(x[i], lo), hi);
            float v = ((c - lo) / (hi - lo)) * 255.0f;
            q[i] = rust_f32_as_u8(v);
            s += (uint32_t)q[i] * (uint32_t)q[i]; /* sum::<u32>() (wrapping in release) */
        }
        *mag = sqrtf((float)s);
        return COSO_OK;
    }
    case COSO_STORAGE_SUBBYTE:
        if (resolution < 1 || resolution > 8) return COSO_ERR_INVALID;
        quantize_subbyte(x, dim, resolution, (uint8_t *)code);
        *mag = coso_seq_norm_f32(x, dim); /* norm of the ORIGINAL vector, scalar.rs:31-32 */
        return COSO_OK;
    case COSO_STORAGE_F16: {
        uint16_t *q = (uint16_t *)code;
        for (int i = 0; i < dim; i++) q[i] = coso_f32_to_f16(x[i]);
        *mag = coso_seq_norm_f32(x, dim);
        return COSO_OK;
    }
    case COSO_STORAGE_F32:
        memcpy(code, x, (size_t)dim * 4);
        *mag = coso_seq_norm_f32(x, dim);
        return COSO_OK;
    default: return COSO_ERR_INVALID;
    }
}

/* ------------------------------------------------------------------------------------------
 * dot products
 * ---------------------------------------------------------------------------------------- */
uint64_t coso_dot_u8_scalar(const uint8_t *a, const uint8_t *b, int n) { /* dot_product.rs:9-11 */
    uint64_t s = 0;
    for (int i = 0; i < n; i++) s += (uint64_t)a[i] * (uint64_t)b[i];
    return s;
}

/* x86_64.rs:22-66 — 32 bytes/iter, unpack to u16, madd_epi16 into two i32 accumulators,
 * horizontal u32 sums, scalar tail.  Integer-exact, so it equals the scalar sum whenever the
 * i32 lanes do not overflow (n <= ~500k), which the reference also silently assumes. */
uint64_t coso_dot_u8(const uint8_t *a, const uint8_t *b, int n) {
    __m256i sumlo = _mm256_setzero_si256(), sumhi = _mm256_setzero_si256();
    const __m256i z = _mm256_setzero_si256();
    int i = 0;
    for (; i + 32 <= n; i += 32) {
        __m256i va = _mm256_loadu_si256((const __m256i *)(a + i));
        __m256i vb = _mm256_loadu_si256((const __m256i *)(b + i));
        __m256i pl = _mm256_madd_epi16(_mm256_unpacklo_epi8(va, z), _mm256_unpacklo_epi8(vb, z));
        __m256i ph = _mm256_madd_epi16(_mm256_unpackhi_epi8(va, z), _mm256_unpackhi_epi8(vb, z));
        sumlo = _mm256_add_epi32(sumlo, pl);
        sumhi = _mm256_add_epi32(sumhi, ph);
    }
    uint32_t lanes[8];
    uint64_t dot = 0;
    uint32_t acc = 0;
    _mm256_storeu_si256((__m256i *)lanes, sumlo);
    for (int k = 0; k < 8; k++) acc += lanes[k]; /* accumulate_u32: wrapping u32 */
    dot += acc;
    acc = 0;
    _mm256_storeu_si256((__m256i *)lanes, sumhi);
    for (int k = 0; k < 8; k++) acc += lanes[k];
    dot += acc;
    for (; i < n; i++) dot += (uint64_t)a[i] * (uint64_t)b[i];
    return dot;
}

/* x86_64.rs:190-211 — nibble-LUT popcount of 32 bytes; here over n bytes with the same LUT. */
uint64_t coso_count_ones(const uint8_t *p, int n) {
    static const uint8_t lut[16] = {0, 1, 1, 2, 1, 2, 2, 3, 1, 2, 2, 3, 2, 3, 3, 4};
    uint64_t s = 0;
    for (int i = 0; i < n; i++) s += lut[p[i] & 0x0F] + lut[p[i] >> 4];
    return s;
}

/* dot_product.rs:35-57 — x_vec[0] is treated as the LSB plane, x_vec[1] as the MSB plane. */
float coso_dot_quaternary_scalar(const uint8_t *x, const uint8_t *y, int pb) {
    const uint8_t *xl = x, *xm = x + pb, *yl = y, *ym = y + pb;
    uint32_t dot = 0;
    for (int i = 0; i < pb; i++) {
        uint32_t lsbs = (uint32_t)__builtin_popcount(xl[i] & yl[i]);
        uint8_t mid1 = xl[i] & ym[i], mid2 = yl[i] & xm[i];
        uint32_t carry = (uint32_t)__builtin_popcount(mid1 & mid2);
        uint32_t msbs = (uint32_t)__builtin_popcount(xm[i] & ym[i]);
        uint32_t mid = (uint32_t)__builtin_popcount(mid1 ^ mid2);
        dot += (msbs << 2) + (carry << 2) + (mid << 1) + lsbs;
    }
    return (float)dot;
}

/* x86_64.rs:103-160 — 32-byte blocks while i+32 < len (strict), u64 accumulation, scalar tail.
 * Integer-exact; 64-bit popcnt replaces the vpshufb LUT (identical counts). */
static float dot_quaternary_fast(const uint8_t *x, const uint8_t *y, int pb) {
    const uint8_t *xl = x, *xm = x + pb, *yl = y, *ym = y + pb;
    uint64_t dot = 0;
    int i = 0;
    for (; i + 8 <= pb; i += 8) {
        uint64_t a0, a1, b0, b1;
        memcpy(&a0, xl + i, 8); memcpy(&a1, xm + i, 8); memcpy(&b0, yl + i, 8); memcpy(&b1, ym + i, 8);
        uint64_t mid1 = a0 & b1, mid2 = b0 & a1;
        dot += ((uint64_t)__builtin_popcountll(a1 & b1) << 2) + ((uint64_t)__builtin_popcountll(mid1 & mid2) << 2) +
               ((uint64_t)__builtin_popcountll(mid1 ^ mid2) << 1) + (uint64_t)__builtin_popcountll(a0 & b0);
    }
    for (; i < pb; i++) {
        uint8_t mid1 = xl[i] & ym[i], mid2 = yl[i] & xm[i];
        dot += ((uint64_t)__builtin_popcount(xm[i] & ym[i]) << 2) + ((uint64_t)__builtin_popcount(mid1 & mid2) << 2) +
               ((uint64_t)__builtin_popcount(mid1 ^ mid2) << 1) + (uint64_t)__builtin_popcount(xl[i] & yl[i]);
    }
    return (float)dot; /* u64 as f32: RNE */
}

static float dot_binary(const uint8_t *x, const uint8_t *y, int pb) { /* dot_product.rs:21-33 / x86_64.rs:163-187 */
    uint64_t dot = 0;
    for (int i = 0; i < pb; i++) dot += (uint64_t)__builtin_popcount(x[i] & y[i]);
    return (float)dot;
}

static float dot_octal(const uint8_t *x, const uint8_t *y, int pb) { /* dot_product.rs:64-90 == LUT path x86_64.rs:284-407 */
    const uint8_t *x0 = x, *x1 = x + pb, *x2 = x + 2 * pb, *y0 = y, *y1 = y + pb, *y2 = y + 2 * pb;
    uint64_t dot = 0;
    for (int i = 0; i < pb; i++)
        for (int bit = 0; bit < 8; bit++) {
            uint32_t xv = (((x2[i] >> bit) & 1u) << 2) | (((x1[i] >> bit) & 1u) << 1) | ((x0[i] >> bit) & 1u);
            uint32_t yv = (((y2[i] >> bit) & 1u) << 2) | (((y1[i] >> bit) & 1u) << 1) | ((y0[i] >> bit) & 1u);
            dot += xv * yv;
        }
    return (float)dot;
}

float coso_dot_subbyte(const uint8_t *x, const uint8_t *y, int res, int pb, int *status) {
    *status = COSO_OK;
    switch (res) { /* cosine.rs:147-154 */
    case 1: return dot_binary(x, y, pb);
    case 2: return dot_quaternary_fast(x, y, pb);
    case 3: return dot_octal(x, y, pb);
    default: *status = COSO_ERR_CALCULATION; return 0.0f;
    }
}

/* x86_64.rs:418-444 — 8 independent FMA lanes over stride-8 elements, then
 * ((s0+s1)+(s2+s3)) + ((s4+s5)+(s6+s7)), then a non-fused scalar tail. */
float coso_dot_f32(const float *a, const float *b, int n) {
    __m256 sum = _mm256_setzero_ps();
    int chunks = n / 8;
    for (int i = 0; i < chunks; i++)
        sum = _mm256_fmadd_ps(_mm256_loadu_ps(a + 8 * i), _mm256_loadu_ps(b + 8 * i), sum);
    __m256 t = _mm256_hadd_ps(sum, sum);
    t = _mm256_hadd_ps(t, t);
    __m128 f = _mm_add_ps(_mm256_castps256_ps128(t), _mm256_extractf128_ps(t, 1));
    volatile float result = _mm_cvtss_f32(f);
    for (int i = chunks * 8; i < n; i++) {
        float p = a[i] * b[i];
        result = result + p;
    }
    return result;
}

float coso_dot_f32_scalar_order(const float *a, const float *b, int n) {
    float s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int chunks = n / 8;
    for (int i = 0; i < chunks; i++)
        for (int j = 0; j < 8; j++) s[j] = fmaf(a[8 * i + j], b[8 * i + j], s[j]);
    volatile float s01 = s[0] + s[1], s23 = s[2] + s[3], s45 = s[4] + s[5], s67 = s[6] + s[7];
    volatile float lo = s01 + s23, hi = s45 + s67;
    volatile float result = lo + hi;
    for (int i = chunks * 8; i < n; i++) {
        float p = a[i] * b[i];
        result = result + p;
    }
    return result;
}

float coso_dot_f16(const uint16_t *a, const uint16_t *b, int n) { /* dot_product.rs:13-19, sequential */
    volatile float acc = -0.0f;
    for (int i = 0; i < n; i++) {
        float p = coso_f16_to_f32(a[i]) * coso_f16_to_f32(b[i]);
        acc = acc + p;
    }
    return acc;
}

/* ------------------------------------------------------------------------------------------
 * metrics
 * ---------------------------------------------------------------------------------------- */
static inline int32_t total_key(float v) { /* f32::total_cmp key */
    int32_t b;
    memcpy(&b, &v, 4);
    b ^= (int32_t)(((uint32_t)(b >> 31)) >> 1);
    return b;
}
int coso_metric_cmp(int metric, float a, float b) { /* types.rs:401-411 */
    int32_t ka = total_key(a), kb = total_key(b);
    int c = (ka > kb) - (ka < kb);
    return (metric == COSO_METRIC_EUCLIDEAN || metric == COSO_METRIC_HAMMING) ? -c : c;
}

static int storage_dot(int storage, int res, int dim, const void *x, const void *y, float *dot) {
    int st = COSO_OK;
    switch (storage) {
    case COSO_STORAGE_U8: *dot = (float)coso_dot_u8((const uint8_t *)x, (const uint8_t *)y, dim); return COSO_OK; /* u64 as f32 */
    case COSO_STORAGE_SUBBYTE: *dot = coso_dot_subbyte((const uint8_t *)x, (const uint8_t *)y, res, (dim + 7) / 8, &st); return st;
    case COSO_STORAGE_F16: *dot = coso_dot_f16((const uint16_t *)x, (const uint16_t *)y, dim); return COSO_OK;
    case COSO_STORAGE_F32: *dot = coso_dot_f32((const float *)x, (const float *)y, dim); return COSO_OK;
    default: return COSO_ERR_INVALID;
    }
}

int coso_distance(int metric, int storage, int res, int dim, const void *x, float x_mag, const void *y,
                  float y_mag, float *out) {
    switch (metric) {
    case COSO_METRIC_COSINE: { /* cosine.rs:104-235 */
        float dot;
        int st = storage_dot(storage, res, dim, x, y, &dot);
        if (st != COSO_OK) return st;
        float den = x_mag * y_mag;
        if (den == 0.0f) return COSO_ERR_CALCULATION;
        *out = dot / den;
        return COSO_OK;
    }
    case COSO_METRIC_DOT: { /* dotproduct.rs:14-64 — no FullPrecisionFP arm */
        if (storage == COSO_STORAGE_F32) return COSO_ERR_STORAGE_MISMATCH;
        return storage_dot(storage, res, dim, x, y, out);
    }
    case COSO_METRIC_EUCLIDEAN: { /* euclidean.rs:9-66 */
        if (storage == COSO_STORAGE_U8) {
            const uint8_t *a = (const uint8_t *)x, *b = (const uint8_t *)y;
            volatile float acc = -0.0f;
            for (int i = 0; i < dim; i++) {
                int16_t diff = (int16_t)((int16_t)a[i] - (int16_t)b[i]);
                int16_t sq = (int16_t)(diff * diff); /* i16 multiply wraps in release builds (|diff| >= 182) */
                acc = acc + (float)sq;
            }
            *out = sqrtf(acc);
            return COSO_OK;
        }
        if (storage == COSO_STORAGE_F16) {
            const uint16_t *a = (const uint16_t *)x, *b = (const uint16_t *)y;
            volatile float acc = -0.0f;
            for (int i = 0; i < dim; i++) {
                float d = coso_f16_to_f32(a[i]) - coso_f16_to_f32(b[i]);
                float p = d * d;
                acc = acc + p;
            }
            *out = sqrtf(acc);
            return COSO_OK;
        }
        if (storage == COSO_STORAGE_SUBBYTE) return COSO_ERR_UNIMPLEMENTED; /* euclidean.rs:34-37 */
        return COSO_ERR_STORAGE_MISMATCH;
    }
    case COSO_METRIC_HAMMING: { /* hamming.rs:10-115 */
        if (storage == COSO_STORAGE_U8) {
            const uint8_t *a = (const uint8_t *)x, *b = (const uint8_t *)y;
            volatile float acc = -0.0f;
            for (int i = 0; i < dim; i++) acc = acc + (float)__builtin_popcount(a[i] ^ b[i]);
            *out = acc;
            return COSO_OK;
        }
        if (storage == COSO_STORAGE_SUBBYTE) {
            if (res == 0 || res > 8) { *out = INFINITY; return COSO_OK; }
            const int pb = (dim + 7) / 8;
            const uint8_t mask = (uint8_t)((1u << res) - 1);
            const int per = 8 / res;
            const uint8_t *a = (const uint8_t *)x, *b = (const uint8_t *)y;
            volatile float acc = 0.0f;
            for (int p = 0; p < res; p++)
                for (int i = 0; i < pb; i++)
                    for (int k = 0; k < per; k++) {
                        int sh = k * res;
                        uint8_t vx = (uint8_t)((a[p * pb + i] >> sh) & mask), vy = (uint8_t)((b[p * pb + i] >> sh) & mask);
                        acc = acc + (float)__builtin_popcount(vx ^ vy);
                    }
            *out = acc;
            return COSO_OK;
        }
        if (storage == COSO_STORAGE_F16) {
            const uint16_t *a = (const uint16_t *)x, *b = (const uint16_t *)y;
            volatile float acc = -0.0f;
            for (int i = 0; i < dim; i++) acc = acc + (float)__builtin_popcount((uint32_t)(a[i] ^ b[i]));
            *out = acc;
            return COSO_OK;
        }
        return COSO_ERR_STORAGE_MISMATCH;
    }
    default: return COSO_ERR_INVALID;
    }
}

/* common.rs:421-429 / 373-379 */
static double rust_powi(double a, int b) { /* f64::powi -> llvm.powi -> compiler-rt __powidf2 */
    const int recip = b < 0;
    double r = 1.0;
    for (;;) {
        if (b & 1) r *= a;
        b /= 2;
        if (b == 0) break;
        a *= a;
    }
    return recip ? 1.0 / r : r;
}
void coso_level_probs(double x, int num_levels, double *values, uint8_t *levels) {
    int k = 0;
    for (int n = num_levels; n >= 0; n--, k++) {
        values[k] = 1.0 - rust_powi(x, -n);
        levels[k] = (uint8_t)n;
    }
}
int coso_max_insert_level(double x, const double *values, const uint8_t *levels, int n) {
    for (int i = 0; i < n; i++)
        if (x >= values[i]) return levels[i];
    return -1; /* reference panics */
}

/* fixedset.rs:1-29 */
void coso_fixedset_insert(coso_fixedset *s, uint32_t v) {
    uint32_t mask = s->len - 1;
    s->buckets[(v >> 6) & mask] |= 1ull << (v & 0x3f);
}
int coso_fixedset_is_member(const coso_fixedset *s, uint32_t v) {
    uint32_t mask = s->len - 1;
    return (s->buckets[(v >> 6) & mask] >> (v & 0x3f)) & 1ull;
}

/* indexes/hnsw/mod.rs:202-351 — sample_embedding counters + finalize_sampling thresholds */
void coso_sample_values_range(const float *x, uint64_t total, float clamp_margin_percent, float *lo, float *hi) {
    static const float T[7] = {0.025f, 0.05f, 0.1f, 0.2f, 0.3f, 0.4f, 0.5f};
    uint64_t above[7] = {0}, below[7] = {0};
    for (uint64_t i = 0; i < total; i++)
        for (int t = 0; t < 7; t++) {
            if (x[i] > T[t]) above[t]++;
            if (x[i] < -T[t]) below[t]++;
        }
    const float values_count = (float)total;
    *hi = 1.0f;
    *lo = -1.0f;
    for (int t = 0; t < 7; t++)
        if (((float)above[t] / values_count) * 100.0f <= clamp_margin_percent) { *hi = T[t]; break; }
    for (int t = 0; t < 7; t++)
        if (((float)below[t] / values_count) * 100.0f <= clamp_margin_percent) { *lo = -T[t]; break; }
}
