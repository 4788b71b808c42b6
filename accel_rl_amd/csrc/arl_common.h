// Shared helpers for the gfx950 kernels (internal; the public ABI is include/accel_rl_hip.h).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "accel_rl_hip.h"
#include "accel_rl_hip_dev.h"

#define ARL_WAVE 64

namespace arl {

void set_error(const char* fmt, ...);

inline int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("%s: %s", what, hipGetErrorString(e));
        return (int)e;
    }
    return 0;
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }
inline bool aligned4(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 3u) == 0; }

// grid for a streaming kernel: enough blocks to fill 256 CUs x 8, never more than needed
inline unsigned stream_grid(int64_t work_items, int per_block) {
    int64_t b = (work_items + per_block - 1) / per_block;
    if (b < 1) b = 1;
    if (b > 2048) b = 2048;
    return (unsigned)b;
}

}  // namespace arl

#define ARL_REQUIRE(cond, code, msg)            \
    do {                                        \
        if (!(cond)) {                          \
            arl::set_error("%s: %s", __func__, msg); \
            return (code);                      \
        }                                       \
    } while (0)
