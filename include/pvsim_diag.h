/*
 * pvsim_diag.h -- DIAGNOSTIC entry points of libpvsim_hip.so.  Not part of the product ABI (include/pvsim.h): nothing in the
 * package's encode / similarity / retrieval path calls them; they exist for tests/tools/{assign,fused}_profile.py, which read
 * in-kernel cycle stamps of specially built kernel variants.  The product kernels carry no stamps.
 */
#ifndef PVSIM_DIAG_H
#define PVSIM_DIAG_H

#include "pvsim.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Diagnostic build of the fused VLAD encode (in-kernel cycle stamps of one wave per workgroup; the product kernel carries
 * none).  enable != 0: following fused launches run the stamped kernel and add into 16 counters; out16 (optional) receives and
 * resets them: [0..6] shader cycles in P0 / A / reduce / exact re-evaluation / K2 / epilogue / image switch summed over the
 * workgroups, [8] stages, [9] stages with a re-evaluation, [10] re-evaluation entries, [11] rows the margin did not settle. */
int pvs_fused_profile(pvs_ctx* ctx, int enable, int64_t* out16);

#ifdef __cplusplus
}
#endif
#endif /* PVSIM_DIAG_H */
