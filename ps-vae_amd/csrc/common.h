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
