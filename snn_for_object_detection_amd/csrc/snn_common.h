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

// Compute units of the current device (MI355X: 256), asked once from the runtime.  Host-only planning helpers
// (snn_conv2d_wgrad_splitk, *_size) may be called in a process without a device: they then plan for 256 CUs.
// Memory-bound kernels cap the grid at 8 blocks / CU and grid-stride the rest.
static inline int snn_num_cu() {
    static int cached = 0;
    if (cached > 0) return cached;
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) == hipSuccess &&
        hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0) {
        cached = n;
        return n;
    }
    (void)hipGetLastError();  // no device in this process: not an error of the caller
    return 256;
}
static inline int snn_max_blocks() { return snn_num_cu() * 8; }

// Tuning / bisecting knobs exist only in builds made with -DSNN_TUNING (python -m snn_for_object_detection_amd._build
// --tuning); the product library reads no environment variable.
#ifdef SNN_TUNING
#include <stdlib.h>
static inline const char* snn_tuning_env(const char* name) { return getenv(name); }
#else
static inline const char* snn_tuning_env(const char*) { return nullptr; }
#endif

// ---- halo-resident 3x3 weight gradient (wgrad_halo.hip), used by snn_conv2d_wgrad / snn_conv2d_wgrad_splitk
struct SnnWgradHaloPlan {
    int ok;                      // 0: shape not covered (the implicit-GEMM weight gradient takes it)
    int R, CW, npr, npc, nks;    // patch rows / columns, patches per image column / row, K16 steps per patch
    int HR, HC, HWD, HWD2;       // halo rows / columns, LDS row pitch in pixels, parity offset (stride 2)
    int wco, wk;                 // waves over output channels / over the K-steps of a patch
    int tiles_co, tiles_ci, splits, pps, patches;
    int slabs;                   // workspace slabs the launch writes (= splits: the K-sharing waves add up in LDS)
};
SnnWgradHaloPlan snn_wgrad_halo_plan(int64_t N, int H, int W, int Cin, int Ho, int Wo, int Cout, int KH, int KW,
                                     int stride, int pad);
// 0 ok, 2 launch error, -1 buffers not addressable by this kernel (caller falls back; the plan must not have been
// used to size the workspace in that case - the caller checks the same conditions before planning)
int snn_wgrad_halo_launch(const SnnWgradHaloPlan& p, const float* x, int64_t ldx, const float* dy, int64_t lddy,
                          float* workspace, int64_t N, int H, int W, int Cin, int Ho, int Wo, int Cout, int stride,
                          int nprod /* 3: bf16 x 3, 1: bf16 x 1 */, hipStream_t st);

typedef float f32x4 __attribute__((ext_vector_type(4)));

// BatchNorm statistics partials (fp64 pairs), laid out [t][c][chunk]: the lanes of the finalize kernel that share one
// (t, c) read consecutive chunks - 512 contiguous bytes per load instruction.  With the chunk index outside the
// channel ([t][chunk][c], as the producers would like to write it) every lane's 16 bytes sat in a different cache line:
// 16.2 -> 9.3 us per finalize launch, for 1-4 % more time in the producing epilogues (their stores become 16-byte
// pieces) - about 0.1 ms per step net.
__host__ __device__ inline int64_t snn_bn_partial_index(int64_t t, int64_t chunk, int64_t c, int64_t chunks, int64_t C) {
    return ((t * C + c) * chunks + chunk) * 2;
}
typedef float f32x16 __attribute__((ext_vector_type(16)));
