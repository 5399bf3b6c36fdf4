// text_terms.hip — text -> BM25 terms, host code: what TFIDFIndex does before the postings exist.
//   tokenize                         indexes/tf_idf/mod.rs:288-308   (runs of char::is_alphanumeric() or '_')
//   STOPWORDS                        indexes/tf_idf/mod.rs:282-286
//   process_text                     indexes/tf_idf/mod.rs:310-360   (skip tokens longer than max_token_len BYTES, lowercase,
//                                    stopwords, stem, xxhash32 seed 0 of the stemmed bytes, count per hash)
//   compute_bm25_term_frequency      indexes/tf_idf/mod.rs:362-371
//   count_tokens                     indexes/tf_idf/mod.rs:373-389
// The stemmer is the reference's un-vendored git dependency (snowball-stemmer 0.1.0, Cargo.lock:2571-2573): its source is not
// available here.  Stemming is a CALLBACK: cos_stem_english (stem_english.hip — the published English Snowball algorithm that
// crate ports, pinned by the algorithm's sample vocabulary) or a shim over the host's own Stemmer; with NULL the lowercased token
// is hashed unstemmed.  Parity of everything else is pinned: xxhash32 against the `xxhash` package, the
// tokenizer / stopwords / tf arithmetic against a Python restatement (tests/test_text_terms.py).  Non-ASCII classification and
// lowercasing use the C library's Unicode tables (C.UTF-8), which agree with Rust's for letters and decimal digits; Rust also
// treats the Nl / No number classes (e.g. superscripts, fractions) as alphanumeric and has a few multi-character lowercase
// mappings — documented differences, outside the ASCII + Latin text the tests pin.
// The reference returns the terms in FxHashMap order; here they come out by ascending hash.
#include <algorithm>
#include <clocale>
#include <cstring>
#include <cwctype>
#include <locale.h>
#include <map>
#include <string>
#include <vector>

#include "engine_internal.h"

namespace {

const char *const STOPWORDS[35] = {"a", "and", "are", "as", "at", "be", "but", "by", "for", "if", "in", "into", "is", "it", "no", "not", "of", "on",
                                   "or", "s", "such", "t", "that", "the", "their", "then", "there", "these", "they", "this", "to", "was", "will",
                                   "with", "www"};

locale_t utf8_locale() {
    static locale_t loc = [] {
        locale_t l = newlocale(LC_CTYPE_MASK, "C.UTF-8", (locale_t)0);
        if (!l) l = newlocale(LC_CTYPE_MASK, "en_US.UTF-8", (locale_t)0);
        return l;
    }();
    return loc;
}

// one UTF-8 scalar at s[i..]; invalid bytes are passed through as U+FFFD of length 1 (Rust &str is always valid UTF-8)
inline uint32_t decode(const unsigned char *s, size_t n, size_t i, size_t &len) {
    const unsigned char c = s[i];
    if (c < 0x80) { len = 1; return c; }
    const int extra = (c >= 0xF0) ? 3 : (c >= 0xE0) ? 2 : (c >= 0xC0) ? 1 : -1;
    if (extra < 0 || i + (size_t)extra >= n) { len = 1; return 0xFFFD; }
    uint32_t cp = c & (0x3F >> extra);
    for (int k = 1; k <= extra; k++) {
        if ((s[i + k] & 0xC0) != 0x80) { len = 1; return 0xFFFD; }
        cp = (cp << 6) | (s[i + k] & 0x3F);
    }
    len = (size_t)extra + 1;
    return cp;
}
inline void encode(uint32_t cp, std::string &out) {
    if (cp < 0x80) out.push_back((char)cp);
    else if (cp < 0x800) { out.push_back((char)(0xC0 | (cp >> 6))); out.push_back((char)(0x80 | (cp & 0x3F))); }
    else if (cp < 0x10000) { out.push_back((char)(0xE0 | (cp >> 12))); out.push_back((char)(0x80 | ((cp >> 6) & 0x3F))); out.push_back((char)(0x80 | (cp & 0x3F))); }
    else { out.push_back((char)(0xF0 | (cp >> 18))); out.push_back((char)(0x80 | ((cp >> 12) & 0x3F))); out.push_back((char)(0x80 | ((cp >> 6) & 0x3F))); out.push_back((char)(0x80 | (cp & 0x3F))); }
}
inline bool is_token_char(uint32_t cp) { // char::is_alphanumeric() || '_'
    if (cp < 0x80) return (cp >= '0' && cp <= '9') || (cp >= 'a' && cp <= 'z') || (cp >= 'A' && cp <= 'Z') || cp == '_';
    locale_t l = utf8_locale();
    return l ? iswalnum_l((wint_t)cp, l) != 0 : false;
}
inline uint32_t lower(uint32_t cp) {
    if (cp < 0x80) return (cp >= 'A' && cp <= 'Z') ? cp + 32 : cp;
    locale_t l = utf8_locale();
    return l ? (uint32_t)towlower_l((wint_t)cp, l) : cp;
}

// XXH32 (xxHash, 32-bit, seed) — the algorithm twox-hash's XxHash32 implements (Cargo.lock:2967)
inline uint32_t rotl(uint32_t x, int r) { return (x << r) | (x >> (32 - r)); }
inline uint32_t rd32(const unsigned char *p) { uint32_t v; memcpy(&v, p, 4); return v; } // little endian host
uint32_t xxh32(const unsigned char *p, size_t len, uint32_t seed) {
    const uint32_t P1 = 2654435761u, P2 = 2246822519u, P3 = 3266489917u, P4 = 668265263u, P5 = 374761393u;
    const unsigned char *end = p + len;
    uint32_t h;
    if (len >= 16) {
        uint32_t v1 = seed + P1 + P2, v2 = seed + P2, v3 = seed, v4 = seed - P1;
        const unsigned char *limit = end - 16;
        do {
            v1 = rotl(v1 + rd32(p) * P2, 13) * P1; p += 4;
            v2 = rotl(v2 + rd32(p) * P2, 13) * P1; p += 4;
            v3 = rotl(v3 + rd32(p) * P2, 13) * P1; p += 4;
            v4 = rotl(v4 + rd32(p) * P2, 13) * P1; p += 4;
        } while (p <= limit);
        h = rotl(v1, 1) + rotl(v2, 7) + rotl(v3, 12) + rotl(v4, 18);
    } else
        h = seed + P5;
    h += (uint32_t)len;
    while (p + 4 <= end) { h = rotl(h + rd32(p) * P3, 17) * P4; p += 4; }
    while (p < end) { h = rotl(h + (*p) * P5, 11) * P1; p++; }
    h ^= h >> 15; h *= P2; h ^= h >> 13; h *= P3; h ^= h >> 16;
    return h;
}

// tokenize + per-token filters shared by process_text and count_tokens: calls f(lowercased token) for every kept token
template <typename F>
void for_each_kept_token(const unsigned char *s, size_t n, size_t max_token_len, F &&f) {
    size_t i = 0, start = (size_t)-1;
    std::string lowered;
    auto flush = [&](size_t endpos) {
        if (start == (size_t)-1) return;
        const size_t blen = endpos - start;
        if (blen <= max_token_len) { // token.len() > max_token_len -> skipped (byte length)
            lowered.clear();
            for (size_t k = start; k < endpos;) { size_t l; const uint32_t cp = decode(s, n, k, l); encode(lower(cp), lowered); k += l; }
            bool stop = false;
            for (const char *w : STOPWORDS) if (lowered == w) { stop = true; break; }
            if (!stop) f(lowered);
        }
        start = (size_t)-1;
    };
    while (i < n) {
        size_t l;
        const uint32_t cp = decode(s, n, i, l);
        if (is_token_char(cp)) { if (start == (size_t)-1) start = i; }
        else flush(i);
        i += l;
    }
    flush(n);
}

} // namespace

extern "C" uint32_t cos_xxhash32(const void *data, size_t len, uint32_t seed) { return xxh32((const unsigned char *)data, len, seed); }

extern "C" uint32_t cos_text_count_tokens(const char *utf8, size_t len, uint32_t max_token_len) {
    uint32_t c = 0;
    if (utf8) for_each_kept_token((const unsigned char *)utf8, len, max_token_len, [&](const std::string &) { c++; });
    return c;
}

extern "C" float cos_bm25_term_frequency(uint32_t count, uint32_t document_length, float average_document_length, float k1, float b) {
    // count as f32 * (k1 + 1.0) / (count as f32 + k1 * (1.0 - b + b * (document_length as f32 / average_document_length)))
    const float c = (float)count;
    const float inner = (1.0f - b) + b * ((float)document_length / average_document_length);
    return c * (k1 + 1.0f) / (c + k1 * inner);
}

extern "C" int32_t cos_text_process(const char *utf8, size_t len, uint32_t max_token_len, float average_document_length, float k1, float b,
                                    cos_stem_fn stem, void *stem_ctx, uint32_t *out_hashes, float *out_tfs, uint32_t cap, uint32_t *out_n) {
    if (!utf8 || !out_n || (cap && (!out_hashes || !out_tfs))) return cos_fail(COS_ERR_INVALID, "null argument");
    std::map<uint32_t, uint32_t> freq; // ascending hash: the order the postings upload wants anyway
    uint32_t document_length = 0;
    std::vector<char> buf;
    for_each_kept_token((const unsigned char *)utf8, len, max_token_len, [&](const std::string &tok) {
        document_length++;
        uint32_t h;
        if (stem) {
            buf.resize(tok.size() * 2 + 16);
            size_t m = stem(stem_ctx, tok.data(), tok.size(), buf.data(), buf.size());
            if (m > buf.size()) m = buf.size();
            h = xxh32((const unsigned char *)buf.data(), m, 0);
        } else
            h = xxh32((const unsigned char *)tok.data(), tok.size(), 0);
        freq[h]++;
    });
    *out_n = (uint32_t)freq.size();
    if (freq.size() > cap) return cos_fail(COS_ERR_INVALID, "%zu distinct terms, room for %u", freq.size(), cap);
    uint32_t i = 0;
    for (const auto &kv : freq) {
        out_hashes[i] = kv.first;
        out_tfs[i] = cos_bm25_term_frequency(kv.second, document_length, average_document_length, k1, b);
        i++;
    }
    return COS_OK;
}
