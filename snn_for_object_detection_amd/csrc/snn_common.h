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
                          int nprod /* 3: bf16 x 3, 1: bf16 x 1 */, bool bf16_storage /* x, dy bf16 (nprod 1) */,
                          hipStream_t st, float x_th = 0.0f);

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

// ---- activation storage codec (the opt-in bf16-STORAGE throughput mode, SNN_PREC_BF16S / SNN_SCAN_BF16_STORAGE): an
// activation tensor in HBM is fp32, or bf16 = the upper half of the fp32 pattern, rounded to nearest even when stored.
// Kernels compute in fp32 registers either way; SnnStore<BF> moves 1 or 4 consecutive elements at ELEMENT index i.
#ifdef __HIPCC__
typedef unsigned snn_u32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 snn_bf16x2 __attribute__((ext_vector_type(2)));
typedef float snn_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x4 snn_unpack_bf16x4(snn_u32x2 r) {
    f32x4 v;
    v[0] = __builtin_bit_cast(float, r[0] << 16);
    v[1] = __builtin_bit_cast(float, r[0] & 0xffff0000u);
    v[2] = __builtin_bit_cast(float, r[1] << 16);
    v[3] = __builtin_bit_cast(float, r[1] & 0xffff0000u);
    return v;
}
__device__ __forceinline__ snn_u32x2 snn_pack_bf16x4(f32x4 v) {
    const snn_bf16x2 a = __builtin_convertvector(snn_f32x2{v[0], v[1]}, snn_bf16x2);   // v_cvt_pk_bf16_f32: RNE, NaN stays NaN
    const snn_bf16x2 b = __builtin_convertvector(snn_f32x2{v[2], v[3]}, snn_bf16x2);
    return snn_u32x2{__builtin_bit_cast(unsigned, a), __builtin_bit_cast(unsigned, b)};
}
template <bool BF> struct SnnStore;
template <> struct SnnStore<false> {
    static constexpr int ES = 4;   // bytes per element
    static __device__ __forceinline__ f32x4 ld4(const void* base, int64_t i) {
        return *reinterpret_cast<const f32x4*>(static_cast<const float*>(base) + i);
    }
    // the same for a tensor nobody reads again (non-temporal: a fused gradient addend is consumed by exactly this load)
    static __device__ __forceinline__ f32x4 ld4_last(const void* base, int64_t i) {
        return __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(static_cast<const float*>(base) + i));
    }
    static __device__ __forceinline__ float ld1(const void* base, int64_t i) { return static_cast<const float*>(base)[i]; }
    static __device__ __forceinline__ void st4(void* base, int64_t i, f32x4 v) {
        *reinterpret_cast<f32x4*>(static_cast<float*>(base) + i) = v;
    }
    static __device__ __forceinline__ void st1(void* base, int64_t i, float v) { static_cast<float*>(base)[i] = v; }
};
template <> struct SnnStore<true> {
    static constexpr int ES = 2;
    static __device__ __forceinline__ f32x4 ld4(const void* base, int64_t i) {
        return snn_unpack_bf16x4(*reinterpret_cast<const snn_u32x2*>(static_cast<const unsigned short*>(base) + i));
    }
    static __device__ __forceinline__ f32x4 ld4_last(const void* base, int64_t i) {
        return snn_unpack_bf16x4(__builtin_nontemporal_load(reinterpret_cast<const snn_u32x2*>(static_cast<const unsigned short*>(base) + i)));
    }
    static __device__ __forceinline__ float ld1(const void* base, int64_t i) {
        return __builtin_bit_cast(float, (unsigned)static_cast<const unsigned short*>(base)[i] << 16);
    }
    static __device__ __forceinline__ void st4(void* base, int64_t i, f32x4 v) {
        *reinterpret_cast<snn_u32x2*>(static_cast<unsigned short*>(base) + i) = snn_pack_bf16x4(v);
    }
    static __device__ __forceinline__ void st1(void* base, int64_t i, float v) {
        const snn_bf16x2 a = __builtin_convertvector(snn_f32x2{v, 0.f}, snn_bf16x2);
        static_cast<unsigned short*>(base)[i] = (unsigned short)(__builtin_bit_cast(unsigned, a) & 0xffffu);
    }
};
#endif
