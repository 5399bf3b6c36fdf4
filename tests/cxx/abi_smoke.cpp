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
    cos_index_destroy(ix);
    printf("OK device\n");
    return 0;
}
