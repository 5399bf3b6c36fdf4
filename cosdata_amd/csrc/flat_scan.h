// Shared between kernels_flat.hip (host flow, tile kernel) and kernels_scan.hip (query-resident quaternary scan kernel).
#pragma once
#include <hip/hip_runtime.h>

#include "engine_types.h"

namespace cosdev {

// Threshold-filtered epilogue (FUSED): instead of writing the [B][chunk] score matrix to HBM for a second kernel to select from
// (10 GB written + 10 GB re-read per 256 x 10M scan against 1.9 GB of codes), every score is compared with its query's current
// SEL-th best key and only the rare survivors are appended to app[B][cap] through a per-query counter.  A reciprocal-based
// estimate (4 VALU ops) screens the elements; the exact IEEE quotient — the value that is ranked — is formed only for those
// within 4e-6 of the threshold or above it, so results are identical to the unfused path.
struct FusedOut {
    const u64 *thr;   // [B] SEL-th best (key) so far; 0 = pool not full yet, everything passes
    u64 *app;         // [B][cap]
    u32 *app_cnt;     // [B]
    u32 cap;
    const uint8_t *qdigits; // ENG_Q2: [B][kdims] pre-expanded query digits (expand_q2_digits_kernel)
};


// query-resident quaternary scan (kernels_scan.hip)
bool flat_scan_supported(u32 kdims);
// queries' planes -> permuted digit bytes [B][kdims] (the layout the scan kernel multiplies)
// (fp4: the operands as e2m1 nibbles for the scaled MFMA — flat_scan_q2_fp4 — instead of i8 digits; the buffer is half as long)
hipError_t launch_flat_scan_expand_queries(const uint8_t *qcodes, u64 row_stride, u32 B, u32 kdims, uint8_t *digits, hipStream_t st, bool fp4);
hipError_t launch_flat_scan(u32 kdims, u32 n_cus, hipStream_t st, const uint8_t *qdig, const float *qmags, u32 B, const uint8_t *codes,
                            const float *mags, u64 row_stride, u32 n0, u32 nc, u32 metric, const FusedOut &fo, bool fp4);

// the same for u8 codes (flat_scan_u8_areg): qcodes = the queries' code rows [B][kdims], qsums / csums = code sums; rows must be exactly kdims bytes
hipError_t launch_flat_scan_u8(u32 kdims, u32 n_cus, hipStream_t st, const uint8_t *qcodes, const u32 *qsums, const float *qmags, u32 B, const uint8_t *codes,
                               const u32 *csums, const float *mags, u64 row_stride, u32 n0, u32 nc, u32 metric, const FusedOut &fo);

// the walk's level table as a query-resident GEMM (kernels_scan.hip): tab[q][c] = (f32) exact integer dot; u8 or quaternary codes
bool level_table_areg_supported(int eng, u64 row_stride);
hipError_t launch_level_table_areg(int eng, u32 n_cus, hipStream_t st, const uint8_t *qcodes, const u32 *qsums, u32 B, const uint8_t *tcodes,
                                   const u32 *tcsums, u64 row_stride, u32 ncols, float *tab, u64 tab_stride);

} // namespace cosdev
