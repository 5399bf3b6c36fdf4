// Links libcosdata_hip.so through include/cosdata_hip.h exactly as a C/C++/Rust host would and checks the
// error contract that needs no GPU: bad params are rejected with a message, and without a device every
// entry point fails loudly (COS_ERR_NO_DEVICE) instead of falling back to a CPU path.
#include <cstdio>
#include <cstring>

#include "cosdata_hip.h"

int main() {
    cos_params p;
    memset(&p, 0, sizeof(p));
    cos_index *ix = nullptr;
    int rc = cos_index_create(&p, &ix); // struct_size/abi_version missing
    if (rc != COS_ERR_INVALID || ix != nullptr || strlen(cos_last_error_string()) == 0) { printf("FAIL invalid-params rc=%d\n", rc); return 1; }
    p.struct_size = sizeof(p); p.abi_version = COS_ABI_VERSION; p.dim = 96; p.metric = COS_METRIC_COSINE; p.storage = COS_STORAGE_U8;
    p.range_lo = -1.f; p.range_hi = 1.f; p.num_layers = 9; p.neighbors_count = 32; p.level0_neighbors_count = 64;
    p.ef_construction = 128; p.ef_search = 256; p.shortlist_size = 64;
    p.neighbors_count = 24; // not a power of two (PerformantFixedSet needs one)
    rc = cos_index_create(&p, &ix);
    if (rc != COS_ERR_INVALID) { printf("FAIL pow2 rc=%d\n", rc); return 1; }
    p.neighbors_count = 32;
    // shard set: argument contract (no GPU needed)
    cos_shardset *ss = nullptr;
    if (cos_shardset_create(nullptr, 0, 0, 0, nullptr, &ss) != COS_ERR_INVALID || ss != nullptr) { printf("FAIL shardset null list\n"); return 1; }
    if (cos_shardset_search_batch(nullptr, nullptr, 0, 0, nullptr, nullptr, nullptr) != COS_ERR_INVALID) { printf("FAIL shardset search args\n"); return 1; }
    if (cos_shardset_destroy(nullptr) != COS_OK) { printf("FAIL shardset destroy(null)\n"); return 1; }
    // host-side entry points (no GPU needed): text -> BM25 terms with the built-in English stemmer, sparse index construction
    {
        char st[32];
        const size_t n = cos_stem_english(nullptr, "consolidating", 13, st, sizeof(st));
        if (n != 8 || memcmp(st, "consolid", 8) != 0) { printf("FAIL stem\n"); return 1; }
        const char *text = "The knights were knitting; a knight's consolation.";
        uint32_t hashes[16], nterms = 0;
        float tfs[16];
        if (cos_text_process(text, strlen(text), 40, 6.0f, 1.5f, 0.75f, cos_stem_english, nullptr, hashes, tfs, 16, &nterms) != COS_OK) { printf("FAIL text_process\n"); return 1; }
        // kept tokens: knights were knitting knight consolation ("the", "a", "s" are stop words) -> stems knight x2, were, knit, consol
        if (cos_text_count_tokens(text, strlen(text), 40) != 5 || nterms != 4) { printf("FAIL text terms %u\n", nterms); return 1; }
        bool found = false;
        for (uint32_t i = 0; i < nterms; i++) found |= hashes[i] == cos_xxhash32("knight", 6, 0) && tfs[i] == cos_bm25_term_frequency(2, 5, 6.0f, 1.5f, 0.75f);
        if (!found) { printf("FAIL stemmed term\n"); return 1; }
        // two vectors, dims {3, 7} / {7}: dimension 7 holds both ids, filed under their quantized values
        const uint64_t row_off[3] = {0, 2, 3};
        const uint32_t rdims[3] = {3, 7, 7};
        const float rvals[3] = {1.5f, 0.1f, 2.9f};
        uint32_t nd = 0;
        if (cos_sparse_build_csr(2, 3.0f, 2, row_off, rdims, rvals, nullptr, nullptr, nullptr, &nd) != COS_OK || nd != 2) { printf("FAIL csr sizes\n"); return 1; }
        uint32_t dims[2], ids[3];
        uint64_t ko[2 * 5];
        if (cos_sparse_build_csr(2, 3.0f, 2, row_off, rdims, rvals, dims, ko, ids, &nd) != COS_OK) { printf("FAIL csr\n"); return 1; }
        // quantize(v) = min(3, trunc(v / 3 * 3)): 1.5 -> 1, 0.1 -> 0, 2.9 -> 2
        if (dims[0] != 3 || dims[1] != 7 || ko[0] != 0 || ko[1] != 0 || ko[2] != 1 || ko[4] != 1 || ko[5] != 1 || ko[6] != 2 || ko[7] != 2 || ko[8] != 3 ||
            ko[9] != 3 || ids[0] != 0 || ids[1] != 0 || ids[2] != 1) { printf("FAIL csr content\n"); return 1; }
    }
    int ndev = -1;
    int rcd = cos_device_count(&ndev);
    rc = cos_index_create(&p, &ix);
    if (rcd != COS_OK || ndev <= 0) {
        if (rc != COS_ERR_NO_DEVICE || ix != nullptr) { printf("FAIL expected NO_DEVICE, rc=%d\n", rc); return 1; }
        float x[96] = {0}; unsigned char codes[96]; float mag;
        if (cos_quantize_batch(COS_STORAGE_U8, 0, 96, -1.f, 1.f, x, 1, codes, &mag) != COS_ERR_NO_DEVICE) { printf("FAIL quantize fallback\n"); return 1; }
        printf("OK no-device: %s\n", cos_last_error_string());
        return 0;
    }
    if (rc != COS_OK || !ix) { printf("FAIL create on device rc=%d %s\n", rc, cos_last_error_string()); return 1; }
    float q[96] = {0}; unsigned ids[4]; float sc[4]; unsigned cnt;
    rc = cos_search_batch(ix, q, 1, 4, ids, sc, &cnt, nullptr); // nothing uploaded yet
    if (rc != COS_ERR_NOT_READY) { printf("FAIL not-ready rc=%d\n", rc); return 1; }
    // two shards of one collection on this device behind ONE call: build both on the device, search through the shard set
    {
        const unsigned n = 600, d = 96, B = 8, k = 5;
        static float X[2][600 * 96], Q[8 * 96];
        unsigned long long st = 88172645463325252ull;
        auto rnd = [&]() { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return (float)((st >> 40) & 0xFFFFFF) / 16777216.0f * 2.0f - 1.0f; };
        for (auto &sh : X) for (float &v : sh) v = rnd();
        for (unsigned b = 0; b < B; b++) for (unsigned j = 0; j < d; j++) Q[b * d + j] = X[b & 1][(b * 37) * d + j] + 0.01f * rnd();
        cos_index *sh[2] = {nullptr, nullptr};
        p.num_layers = 3; p.ef_construction = 32; p.ef_search = 32;
        for (int s = 0; s < 2; s++) {
            p.id_base = s * n; p.seed = 7 + s;
            if (cos_index_create(&p, &sh[s]) != COS_OK || cos_index_upload_vectors(sh[s], X[s], n, COS_UPLOAD_DEFAULT) != COS_OK ||
                cos_index_build(sh[s], 128) != COS_OK) { printf("FAIL shard %d: %s\n", s, cos_last_error_string()); return 1; }
        }
        if (cos_shardset_create(sh, 2, 0, 2, nullptr, &ss) != COS_OK) { printf("FAIL shardset create: %s\n", cos_last_error_string()); return 1; }
        unsigned mi[8 * 5], mc[8], si[8 * 5], sc_[8]; float ms[8 * 5], ssc[8 * 5];
        if (cos_shardset_search_batch(ss, Q, B, k, mi, ms, mc) != COS_OK) { printf("FAIL shardset search: %s\n", cos_last_error_string()); return 1; }
        bool seen[2] = {false, false};
        for (unsigned b = 0; b < B; b++) {
            if (mc[b] != k) { printf("FAIL merged count %u\n", mc[b]); return 1; }
            for (unsigned j = 0; j < k; j++) {
                if (mi[b * k + j] >= 2 * n) { printf("FAIL merged id out of range\n"); return 1; }
                if (j && ms[b * k + j] > ms[b * k + j - 1]) { printf("FAIL merged scores not sorted\n"); return 1; }
                seen[mi[b * k + j] / n] = true;
            }
            // the merged head must be the better of the two shards' own heads
            float best = -2.f;
            for (int s = 0; s < 2; s++) {
                if (cos_search_batch(sh[s], Q + b * d, 1, k, si, ssc, sc_, nullptr) != COS_OK) { printf("FAIL shard search\n"); return 1; }
                if (ssc[0] > best) best = ssc[0];
            }
            if (ms[b * k] != best) { printf("FAIL merged head %g != best shard head %g\n", ms[b * k], best); return 1; }
        }
        if (!seen[0] || !seen[1]) { printf("FAIL a shard never contributed\n"); return 1; }
        cos_shardset_destroy(ss);
        cos_index_destroy(sh[0]); cos_index_destroy(sh[1]);
    }
    cos_index_destroy(ix);
    printf("OK device\n");
    return 0;
}
