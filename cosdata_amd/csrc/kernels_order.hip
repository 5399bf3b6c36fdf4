// kernels_order.hip — the locality order of a big search launch (engine_types.h, WalkArgs::phase).
//
// Between the two launches of a split walk: the launch's queries are sorted by ORDER KEY (written by the upper-level phase: the
// depth-first position of the best node found on the key level) and dealt to the XCDs in contiguous runs.  gfx950 hands workgroup b of a grid to XCD b % 8, and each
// XCD has its own 4 MB L2: with the queries in arrival order every XCD walks the whole graph and no row is found twice in an L2;
// sorted and dealt, the ~600 waves resident on one XCD walk neighbouring regions.  The per-query walk is untouched — the order
// only decides when and where a query's workgroup runs (scripts/locality_probe.py: 11.7 -> 9.6 ms per 32768-query launch with
// an oracle key, every result identical).
//
// Not a reference interface: the reference answers one query per rayon task (indexes/mod.rs:260-272) and has no launch to order.
#include <hipcub/hipcub.hpp>

#include "engine_internal.h"

namespace cosdev {

// num_xcd: the device's XCD count (hipDeviceAttributeNumberOfXccs, read at cos_index_create; MI355X: 8 XCDs x 32 CUs)
__global__ void deal_to_xcds_kernel(const u32 *__restrict__ sorted, u32 B, u32 num_xcd, u32 *__restrict__ q_order) {
    const u32 b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    // XCD x receives workgroups x, x + num_xcd, ...: B / num_xcd of them, one more for x < B % num_xcd.  Its run of the sorted order
    // starts after the runs of the XCDs before it.
    const u32 q = B / num_xcd, r = B % num_xcd, x = b % num_xcd, j = b / num_xcd;
    q_order[b] = sorted[x * q + (x < r ? x : r) + j];
}

hipError_t walk_order_reserve(WalkOrder &o, u32 B) {
    if (B <= o.cap) return hipSuccess;
    walk_order_free(o);
    hipError_t e;
    if ((e = hipMalloc((void **)&o.entry0, (size_t)B * 4)) != hipSuccess) return e;
    if ((e = hipMalloc((void **)&o.order_key, (size_t)B * 4)) != hipSuccess) return e;
    if ((e = hipMalloc((void **)&o.keys_sorted, (size_t)B * 4)) != hipSuccess) return e;
    if ((e = hipMalloc((void **)&o.iota, (size_t)B * 4)) != hipSuccess) return e;
    if ((e = hipMalloc((void **)&o.vals_sorted, (size_t)B * 4)) != hipSuccess) return e;
    if ((e = hipMalloc((void **)&o.q_order, (size_t)B * 4)) != hipSuccess) return e;
    size_t bytes = 0;
    if ((e = hipcub::DeviceRadixSort::SortPairs(nullptr, bytes, o.order_key, o.keys_sorted, o.iota, o.vals_sorted, (int)B, 0, 32, (hipStream_t)0)) != hipSuccess)
        return e;
    if ((e = hipMalloc(&o.tmp, bytes ? bytes : 16)) != hipSuccess) return e;
    o.tmp_bytes = bytes;
    o.cap = B;
    return hipSuccess;
}

void walk_order_free(WalkOrder &o) {
    void *ptrs[] = {o.entry0, o.order_key, o.keys_sorted, o.iota, o.vals_sorted, o.q_order, o.tmp};
    for (void *p : ptrs) if (p) (void)hipFree(p);
    o = WalkOrder();
}

// order_key[0..B) / iota[0..B) (both written by the upper-level phase; keys <= key_max) -> q_order[0..B), all on `st`
hipError_t launch_walk_order(WalkOrder &o, u32 B, u32 key_max, u32 num_xcd, hipStream_t st) {
    if (B > o.cap) return hipErrorInvalidValue;
    int end_bit = 1;
    while (end_bit < 32 && (key_max >> end_bit)) end_bit++; // radix passes over the bits the keys have
    size_t bytes = 0;
    hipError_t e = hipcub::DeviceRadixSort::SortPairs(nullptr, bytes, o.order_key, o.keys_sorted, o.iota, o.vals_sorted, (int)B, 0, end_bit, st);
    if (e != hipSuccess) return e;
    if (bytes > o.tmp_bytes) return hipErrorOutOfMemory; // sized for cap >= B items: cannot happen unless the library's sizing is not monotonic
    bytes = o.tmp_bytes;
    e = hipcub::DeviceRadixSort::SortPairs(o.tmp, bytes, o.order_key, o.keys_sorted, o.iota, o.vals_sorted, (int)B, 0, end_bit, st);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(deal_to_xcds_kernel, dim3((B + 255) / 256), dim3(256), 0, st, o.vals_sorted, B, num_xcd ? num_xcd : 1u, o.q_order);
    return hipGetLastError();
}

} // namespace cosdev
