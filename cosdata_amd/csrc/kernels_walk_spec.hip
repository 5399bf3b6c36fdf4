// kernels_walk_spec.hip — the walk kernel with table values gathered ahead (walk_kernel.inc, COS_WALK_SPEC): a CANDIDATE, launched
// only when COS_WALK_SPEC_TABLE=1 is set (kernels_walk.hip launch_walk_r), for the launches the headline path takes: u8 codes of
// 513..1024 dimensions, reference visited filter, a level table present.  Its own translation unit and its own
// kernel name so that the shipped kernels of kernels_walk.hip stay byte for byte what was measured.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>

#include "engine_types.h"
#include "dot_engines.h"

using namespace cosdev;

#define COS_OK 0
#define COS_ERR_CALCULATION 2
#define COS_QUERY_ID 0xFFFFFFFEu
#define COS_ROOT_ID 0xFFFFFFFFu

// two instances of the kernel source: table values of the first 2 / of all 4 window entries gathered ahead (COS_WALK_SPEC_TABLE=2|4;
// 1 = 2).  Each in its own namespace: the include file defines the kernel's constants and its LDS layout too.
namespace spec2 {
#define COS_WALK_SPEC 2
#define COS_WALK_KERNEL_NAME walk_spec_kernel
#include "walk_kernel.inc"
#undef COS_WALK_SPEC
} // namespace spec2
namespace spec4 {
#define COS_WALK_SPEC 4
#include "walk_kernel.inc"
#undef COS_WALK_SPEC
} // namespace spec4
namespace cosdev {

// kernels_walk_spec_wide.hip: windows of six / eight entries, all gathered ahead
hipError_t launch_walk_spec_wide(const IndexDev &ix, const WalkArgs &wa, int entries, int row_buffers, size_t smem, hipStream_t st);

static int walk_spec_entries() { // COS_WALK_SPEC_TABLE: 0 / unset = off (kernels_walk.hip does not come here), 4 / 6 / 8 = that many entries, anything else = two
    static const int n = [] { const char *e = getenv("COS_WALK_SPEC_TABLE"); const int v = e ? atoi(e) : 0; return v == 4 || v == 6 || v == 8 ? v : 2; }();
    return n;
}
// bytes on top of walk_smem_bytes (which sizes the four-entry window of the shipped kernels): the gathered values, and the second half
// of a six- or eight-entry window
size_t walk_spec_extra_smem() {
    const int n = walk_spec_entries();
    return (size_t)n * 64 * 4 + (n > 4 ? (size_t)(n - 4) * 64 * 4 * 2 : 0);
}

// smem = walk_smem_bytes of the launch + walk_spec_extra_smem(); row_buffers = 8 | 4 (walk_pb_policy)
hipError_t launch_walk_spec(const IndexDev &ix, const WalkArgs &wa_in, int row_buffers, size_t smem, hipStream_t st) {
    dim3 grid(wa_in.B), block(64);
    static const bool warm = [] { const char *e = getenv("COS_WALK_SPEC_WARM"); return e && atoi(e) != 0; }();
    WalkArgs wa = wa_in;
    if (warm) wa.tab_level_min |= 0x80000000u; // walk_kernel.inc: fetch the top of the query's table row before the first level
    const int n = walk_spec_entries();
    if (n > 4) return launch_walk_spec_wide(ix, wa, n, row_buffers, smem, st);
#define SPEC_LAUNCH(NS, R_)                                                                                                                \
    do {                                                                                                                                   \
        if (row_buffers == 8) hipLaunchKernelGGL((NS::walk_spec_kernel<ENG_U8, 1, R_, true, false, 8>), grid, block, smem, st, ix, wa);     \
        else hipLaunchKernelGGL((NS::walk_spec_kernel<ENG_U8, 1, R_, true, false, 4>), grid, block, smem, st, ix, wa);                     \
    } while (0)
#define SPEC_WALK(R_)                                                                                                                      \
    do {                                                                                                                                   \
        if (n == 4) SPEC_LAUNCH(spec4, R_);                                                                                                \
        else SPEC_LAUNCH(spec2, R_);                                                                                                       \
    } while (0)
    if (wa.ef <= 64) SPEC_WALK(1);
    else if (wa.ef <= 256) SPEC_WALK(4);
    else if (wa.ef <= 512) SPEC_WALK(8);
    else return hipErrorInvalidValue;
#undef SPEC_WALK
#undef SPEC_LAUNCH
    return hipGetLastError();
}

} // namespace cosdev
