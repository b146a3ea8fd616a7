// Shared helpers for the gfx950 kernels behind include/snn_hip.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "../../include/snn_hip.h"

void snn_set_error(const char* fmt, ...);

#define SNN_REQUIRE(cond, ...)            \
    do {                                  \
        if (!(cond)) {                    \
            snn_set_error(__VA_ARGS__);   \
            return 1;                     \
        }                                 \
    } while (0)

#define SNN_CHECK_LAUNCH(name)                                                   \
    do {                                                                         \
        hipError_t e_ = hipGetLastError();                                       \
        if (e_ != hipSuccess) {                                                  \
            snn_set_error("%s: launch failed: %s", name, hipGetErrorString(e_)); \
            return 2;                                                            \
        }                                                                        \
    } while (0)

static inline int64_t snn_ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }
__device__ __forceinline__ int64_t snn_ceil_div_dev(int64_t a, int64_t b) { return (a + b - 1) / b; }

// MI355X: 256 CUs; memory-bound kernels cap the grid at 8 blocks/CU and grid-stride the rest.
static constexpr int SNN_NUM_CU = 256;
static constexpr int SNN_MAX_BLOCKS = SNN_NUM_CU * 8;

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
