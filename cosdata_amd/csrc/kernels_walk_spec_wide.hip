// kernels_walk_spec_wide.hip — the gather-ahead walk candidate (walk_kernel.inc, COS_WALK_SPEC; see kernels_walk_spec.hip) with a WIDER
// lookahead window: six or eight adjacency rows per round, every entry's table values gathered ahead (COS_WALK_SPEC_TABLE=6|8).
// The walk consumes ~3 entries of its four-entry window per round on average (17.5 M expansions in 5.9 M rounds at c2) and, in the
// CPU model of its rounds (scripts/window_stats_model.py), ALL four in 80 % of them: the window is the limit.  On a table level a
// consumed entry costs nothing but LDS reads once its values are there, so a wider window turns more expansions into the same two
// round trips.  The price is LDS: 6.1 KB per wave at six entries (6 waves per SIMD), 7.7 KB at eight (5), against 3.6 KB (7).
// Its own translation unit so that the two build in parallel.
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
#define COS_WALK_KERNEL_NAME walk_spec_kernel

namespace spec6 {
#define COS_WALK_SPEC 6
#define COS_WALK_LA 6
#include "walk_kernel.inc"
#undef COS_WALK_SPEC
#undef COS_WALK_LA
} // namespace spec6
namespace spec8 {
#define COS_WALK_SPEC 8
#define COS_WALK_LA 8
#include "walk_kernel.inc"
#undef COS_WALK_SPEC
#undef COS_WALK_LA
} // namespace spec8

namespace cosdev {

hipError_t launch_walk_spec_wide(const IndexDev &ix, const WalkArgs &wa, int entries, int row_buffers, size_t smem, hipStream_t st) {
    dim3 grid(wa.B), block(64);
#define SPEC_LAUNCH(NS, R_)                                                                                                                \
    do {                                                                                                                                   \
        if (row_buffers == 8) hipLaunchKernelGGL((NS::walk_spec_kernel<ENG_U8, 1, R_, true, false, 8>), grid, block, smem, st, ix, wa);     \
        else hipLaunchKernelGGL((NS::walk_spec_kernel<ENG_U8, 1, R_, true, false, 4>), grid, block, smem, st, ix, wa);                     \
    } while (0)
#define SPEC_WALK(R_)                                                                                                                      \
    do {                                                                                                                                   \
        if (entries == 8) SPEC_LAUNCH(spec8, R_);                                                                                          \
        else SPEC_LAUNCH(spec6, R_);                                                                                                       \
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
