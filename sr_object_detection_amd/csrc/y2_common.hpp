// Shared helpers for the HIP translation units (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>
#include "y2_hip.h"

extern "C" void y2h_set_error_(const char *what, const char *detail);

#define Y2H_CHECK(expr)                                                      \
    do {                                                                     \
        hipError_t e_ = (expr);                                              \
        if (e_ != hipSuccess) {                                              \
            y2h_set_error_(#expr, hipGetErrorString(e_));                    \
            return Y2H_EHIP;                                                 \
        }                                                                    \
    } while (0)

#define Y2H_LAUNCH_CHECK()                                                   \
    do {                                                                     \
        hipError_t e_ = hipGetLastError();                                   \
        if (e_ != hipSuccess) {                                              \
            y2h_set_error_("kernel launch", hipGetErrorString(e_));          \
            return Y2H_EHIP;                                                 \
        }                                                                    \
    } while (0)

static inline hipStream_t S(y2h_stream s) { return (hipStream_t)s; }

// memory-bound elementwise launches: cap the grid and grid-stride the rest
static inline unsigned y2h_grid(long n, int block, int max_blocks = 256 * 16)
{
    long g = (n + block - 1) / block;
    if (g > max_blocks) g = max_blocks;
    if (g < 1) g = 1;
    return (unsigned)g;
}
