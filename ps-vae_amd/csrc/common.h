// Shared host-side helpers of libpsvae_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/psvae_hip.h"

void psvae_set_error(const char *fmt, ...);

#define PSVAE_REQUIRE(cond, code, ...)        \
    do {                                      \
        if (!(cond)) {                        \
            psvae_set_error(__VA_ARGS__);     \
            return (code);                    \
        }                                     \
    } while (0)

// After a kernel launch: report launch-configuration errors as the function's result.
#define PSVAE_CHECK_LAUNCH(name)                                               \
    do {                                                                       \
        hipError_t e_ = hipGetLastError();                                     \
        if (e_ != hipSuccess) {                                                \
            psvae_set_error("%s: %s", name, hipGetErrorString(e_));            \
            return (int)e_;                                                    \
        }                                                                      \
    } while (0)

static inline bool psvae_aligned16(const void *p) { return (((uintptr_t)p) & 15u) == 0; }
static inline int psvae_cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }

// A 16-byte store of rows that are written once and read back by a later kernel, far more of them than the caches
// hold: marked non-temporal (one global_store_dwordx4 ... nt).  Measured on the aggregation outputs of a
// 4096-molecule step (84.6 against 89.4 us forward, 97 against 103 us backward, the products that read the rows
// unchanged) and on the pair rows of generation (1.30 against 1.54 ms).
__device__ __forceinline__ void store_streamed(float4 *dst, const float4 v) {
    __builtin_nontemporal_store(v.x, &dst->x);
    __builtin_nontemporal_store(v.y, &dst->y);
    __builtin_nontemporal_store(v.z, &dst->z);
    __builtin_nontemporal_store(v.w, &dst->w);
}
