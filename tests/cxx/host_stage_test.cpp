// StagePool (cosdata_amd/csrc/host_stage.h): the parallel staging copy of the host-buffer API, checked on the CPU — sizes below and
// above the wake-up threshold, sizes that are not multiples of the slice granule, no workers at all, and two callers sharing a pool.
#include <cstdio>
#include <cstdlib>
#include <random>
#include <thread>
#include <vector>

#include "host_stage.h"

static int check(cosdev::StagePool &pool, size_t bytes, unsigned seed) {
    std::vector<unsigned char> src(bytes + 64), dst(bytes + 64, 0xAB);
    std::mt19937 rng(seed);
    for (auto &b : src) b = (unsigned char)rng();
    pool.copy(dst.data() + 7, src.data() + 3, bytes); // unaligned on both sides
    if (bytes && memcmp(dst.data() + 7, src.data() + 3, bytes) != 0) return 1;
    for (size_t i = 0; i < 7; i++) if (dst[i] != 0xAB) return 2;                    // nothing before the range
    for (size_t i = 7 + bytes; i < dst.size(); i++) if (dst[i] != 0xAB) return 3;   // nothing behind it
    return 0;
}

int main() {
    const size_t sizes[] = {0, 1, 4095, 4096, 65536, 300000, 1u << 20, (1u << 20) + 1, 5000003, 25165824, 25165824 + 4097};
    for (unsigned workers : {0u, 1u, 3u, 7u}) {
        cosdev::StagePool pool(workers);
        unsigned seed = 1;
        for (size_t s : sizes) {
            const int rc = check(pool, s, seed++);
            if (rc) { printf("FAIL workers=%u bytes=%zu rc=%d\n", workers, s, rc); return 1; }
        }
        // two callers, one pool: copies take turns
        int bad = 0;
        std::thread a([&] { for (int i = 0; i < 20; i++) bad |= check(pool, 3000000 + 17 * i, 100 + i); });
        std::thread b([&] { for (int i = 0; i < 20; i++) bad |= check(pool, 2000000 + 4099 * i, 200 + i); });
        a.join();
        b.join();
        if (bad) { printf("FAIL concurrent workers=%u\n", workers); return 1; }
    }
    printf("host_stage ok\n");
    return 0;
}
