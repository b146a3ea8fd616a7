// Fused BatchNorm-apply + spiking-neuron temporal scan, forward and BPTT backward (gfx950).
//
// Memory-bound kernels: one thread owns VEC(=4) consecutive channels of one pixel and walks the
// T timesteps with the membrane state (v, i) in registers; every HBM access is a 16-byte
// lane-contiguous vector.  Compiled with -ffp-contract=off so each statement rounds like the
// reference's unfused torch ops (oracle/neurons.py).
//
// Reference semantics: layer_gen.py:211-214 (BatchNorm2d, per-timestep batch statistics),
// layer_gen.py:232-235 / 252-254 (norse LIFCell / LICell), tiny_yolo.py:39-44 (LI -> Tanh).
#include <stdlib.h>
#include <type_traits>
#include "snn_common.h"

#ifdef SNN_TUNING
// tuning builds only: timing experiments on the reverse scan's BatchNorm sums (WRONG results): bit 0 no LDS accumulation,
// bit 1 no shuffles either, bit 2 no slab zeroing / write-out
__device__ int g_bwd_abl = 0;
extern "C" int snn_debug_set_bwd_abl(int v) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_bwd_abl), &v, sizeof(int)); }
#define BWD_ABL(bit) ((g_bwd_abl >> (bit)) & 1)
#else
#define BWD_ABL(bit) 0
#endif

namespace {

constexpr int kThreads = 256;
#ifndef SNN_SCAN_NT_AUX
#define SNN_SCAN_NT_AUX 2   // cache-policy operand of the reverse scan's last-use loads (gfx950: bit 1 = nt)
#endif

template <int VEC> struct Vec;
template <> struct Vec<4> {
    typedef f32x4 type;
    static __device__ __forceinline__ f32x4 load(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
    static __device__ __forceinline__ void store(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }
};
// 8 channels per thread: the bf16-storage scans (16 bytes of bf16 per access; fp32 side tensors as two 16-byte halves)
typedef float f32x8 __attribute__((ext_vector_type(8)));
template <> struct Vec<8> {
    typedef f32x8 type;
    static __device__ __forceinline__ f32x8 load(const float* p) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(p), b = *reinterpret_cast<const f32x4*>(p + 4);
        return __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
    }
    static __device__ __forceinline__ void store(float* p, f32x8 v) {
        *reinterpret_cast<f32x4*>(p) = __builtin_shufflevector(v, v, 0, 1, 2, 3);
        *reinterpret_cast<f32x4*>(p + 4) = __builtin_shufflevector(v, v, 4, 5, 6, 7);
    }
};
template <> struct Vec<1> {
    typedef float type;
    static __device__ __forceinline__ float load(const float* p) { return *p; }
    static __device__ __forceinline__ void store(float* p, float v) { *p = v; }
};
// activation tensors in the storage type (fp32, or bf16 in the bf16-storage mode: snn_common.h SnnStore): element index
template <int VEC, bool SB> struct VecS;
template <bool SB> struct VecS<4, SB> {
    static __device__ __forceinline__ f32x4 load(const float* base, int64_t i) { return SnnStore<SB>::ld4(base, i); }
    // the same for a tensor nobody reads again soon (non-temporal: what stays in L2 / the memory-side cache should be the
    // tensors that go from a producer straight to its consumer - conv -> scan -> conv, scan -> apply -> data gradient)
    static __device__ __forceinline__ f32x4 load_last(const float* base, int64_t i) {
        if constexpr (SNN_SCAN_NT_AUX == 0) return SnnStore<SB>::ld4(base, i);
        else if constexpr (SB) return snn_unpack_bf16x4(__builtin_nontemporal_load(
                                   reinterpret_cast<const snn_u32x2*>(reinterpret_cast<const unsigned short*>(base) + i)));
        else return __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(base + i));
    }
    static __device__ __forceinline__ void store(float* base, int64_t i, f32x4 v) { SnnStore<SB>::st4(base, i, v); }
};
template <> struct VecS<8, true> {   // 8 bf16 values = 16 bytes
    static __device__ __forceinline__ f32x8 load(const float* base, int64_t i) {
        typedef unsigned u32x4_ __attribute__((ext_vector_type(4)));
        const u32x4_ r = *reinterpret_cast<const u32x4_*>(reinterpret_cast<const unsigned short*>(base) + i);
        const f32x4 a = snn_unpack_bf16x4(snn_u32x2{r[0], r[1]}), b = snn_unpack_bf16x4(snn_u32x2{r[2], r[3]});
        return __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
    }
    static __device__ __forceinline__ f32x8 load_last(const float* base, int64_t i) {
        typedef unsigned u32x4_ __attribute__((ext_vector_type(4)));
        const u32x4_* src = reinterpret_cast<const u32x4_*>(reinterpret_cast<const unsigned short*>(base) + i);
        const u32x4_ r = SNN_SCAN_NT_AUX == 0 ? *src : __builtin_nontemporal_load(src);
        const f32x4 a = snn_unpack_bf16x4(snn_u32x2{r[0], r[1]}), b = snn_unpack_bf16x4(snn_u32x2{r[2], r[3]});
        return __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
    }
    static __device__ __forceinline__ void store(float* base, int64_t i, f32x8 v) {
        typedef unsigned u32x4_ __attribute__((ext_vector_type(4)));
        const snn_u32x2 a = snn_pack_bf16x4(__builtin_shufflevector(v, v, 0, 1, 2, 3));
        const snn_u32x2 b = snn_pack_bf16x4(__builtin_shufflevector(v, v, 4, 5, 6, 7));
        *reinterpret_cast<u32x4_*>(reinterpret_cast<unsigned short*>(base) + i) = u32x4_{a[0], a[1], b[0], b[1]};
    }
};
template <bool SB> struct VecS<1, SB> {
    static __device__ __forceinline__ float load(const float* base, int64_t i) { return SnnStore<SB>::ld1(base, i); }
    static __device__ __forceinline__ float load_last(const float* base, int64_t i) { return SnnStore<SB>::ld1(base, i); }
    static __device__ __forceinline__ void store(float* base, int64_t i, float v) { SnnStore<SB>::st1(base, i, v); }
};
template <int VEC> __device__ __forceinline__ float& lane(typename Vec<VEC>::type& v, int j);
template <> __device__ __forceinline__ float& lane<4>(f32x4& v, int j) { return reinterpret_cast<float*>(&v)[j]; }
template <> __device__ __forceinline__ float& lane<8>(f32x8& v, int j) { return reinterpret_cast<float*>(&v)[j]; }
template <> __device__ __forceinline__ float& lane<1>(float& v, int) { return v; }

// ------------------------------------------------------------------------------------------
// BatchNorm statistics: per (t, c) sum and sum of squares over the M pixels of timestep t.
// grid = (chunks, T, channel blocks); partial[t][c][chunk][2] in fp64.
// ------------------------------------------------------------------------------------------
struct StatsPlan {
    int vec, cvb, zblocks, chunks;
};

static StatsPlan stats_plan(int T, int64_t M, int C) {
    StatsPlan pl;
    pl.vec = (C % 4 == 0) ? 4 : 1;
    int cv = C / pl.vec;
    pl.cvb = cv < kThreads ? cv : kThreads;
    pl.zblocks = (int)snn_ceil_div(cv, pl.cvb);
    int P = kThreads / pl.cvb;
    int64_t want = snn_ceil_div(snn_max_blocks(), (int64_t)T * pl.zblocks);
    int64_t maxc = snn_ceil_div(M, (int64_t)P * 8);  // at least ~8 pixels per thread
    if (want > maxc) want = maxc;
    if (want < 1) want = 1;
    pl.chunks = (int)want;
    return pl;
}

template <int VEC, bool SB = false>
__global__ __launch_bounds__(kThreads) void k_bn_stats(const float* __restrict__ y, int64_t ldy, int64_t M, int C,
                                                       int cvb, double* __restrict__ partial) {
    __shared__ double red[kThreads * 2 * VEC];
    const int chunks = gridDim.x, chunk = blockIdx.x, t = blockIdx.y;
    const int cv = C / VEC;
    const int P = kThreads / cvb;
    const int tid = threadIdx.x;
    const int cgl = tid % cvb, ps = tid / cvb;
    const int cg = blockIdx.z * cvb + cgl;
    const bool active = (ps < P) && (cg < cv);
    const int64_t per = snn_ceil_div_dev(M, chunks);
    double s[VEC], q[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) s[j] = q[j] = 0.0;
    if (active) {
        const int64_t m0 = (int64_t)chunk * per;
        int64_t m1 = m0 + per;
        if (m1 > M) m1 = M;
        const int64_t base = ((int64_t)t * M) * ldy + (int64_t)cg * VEC;   // element index (y: fp32, or bf16 with SB)
        for (int64_t m = m0 + ps; m < m1; m += P) {
            typename Vec<VEC>::type v = VecS<VEC, SB>::load(y, base + m * ldy);
#pragma unroll
            for (int j = 0; j < VEC; ++j) {
                double d = (double)lane<VEC>(v, j);
                s[j] += d;
                q[j] += d * d;
            }
        }
    }
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
        red[(tid * VEC + j) * 2 + 0] = s[j];
        red[(tid * VEC + j) * 2 + 1] = q[j];
    }
    __syncthreads();
    if (ps == 0 && cg < cv) {
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            double ss = 0.0, qq = 0.0;
            for (int k = 0; k < P; ++k) {
                ss += red[((k * cvb + cgl) * VEC + j) * 2 + 0];
                qq += red[((k * cvb + cgl) * VEC + j) * 2 + 1];
            }
            double* dst = partial + snn_bn_partial_index(t, chunk, (int64_t)cg * VEC + j, chunks, C);
            dst[0] = ss;
            dst[1] = qq;
        }
    }
}

__global__ void k_bn_stats_finalize(const double* __restrict__ partial, int chunks, int T, int64_t M, int C,
                                    const float* __restrict__ gamma, const float* __restrict__ bias, float eps,
                                    const float* __restrict__ running_mean, const float* __restrict__ running_var,
                                    int use_running, float* __restrict__ mean, float* __restrict__ invstd,
                                    float* __restrict__ alpha, float* __restrict__ beta,
                                    double* __restrict__ var_unbiased) {
    int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= T * C) return;
    int t = idx / C, c = idx % C;
    float mu, is;
    if (use_running) {
        mu = running_mean[c];
        is = 1.0f / sqrtf(running_var[c] + eps);  // ATen eval path: invstd in fp32
    } else {
        double s = 0.0, q = 0.0;
        for (int k = 0; k < chunks; ++k) {
            const double* src = partial + snn_bn_partial_index(t, k, c, chunks, C);
            s += src[0];
            q += src[1];
        }
        double n = (double)M;
        double m = s / n;
        double var = q / n - m * m;
        if (var < 0.0) var = 0.0;
        mu = (float)m;
        is = (float)(1.0 / sqrt(var + (double)eps));
        if (var_unbiased) var_unbiased[idx] = (M > 1) ? var * n / (n - 1.0) : var;
    }
    mean[idx] = mu;
    invstd[idx] = is;
    float g = gamma ? gamma[c] : 1.0f;
    float b = bias ? bias[c] : 0.0f;
    float a = is * g;
    alpha[idx] = a;
    beta[idx] = b - mu * a;
}

// Partials written by a convolution epilogue (conv.hip) come in row tiles of `rows_per_chunk` output pixels that do
// not line up with the timesteps: chunk k of step t is the part of tile (first tile of t) + k that lies in t, so
// the number of written slots differs by one between steps.  rows_per_chunk == 0: every one of `chunks` is written.
__device__ __forceinline__ int chunks_of_step(int chunks, int rows_per_chunk, int t, int64_t M) {
    if (rows_per_chunk <= 0) return chunks;
    return (int)((((int64_t)t + 1) * M - 1) / rows_per_chunk - ((int64_t)t * M) / rows_per_chunk) + 1;
}

// One launch for the whole statistics second phase of a layer (was: finalize + running update, 27 us of two
// latency-bound kernels 22 times per step).  One block per channel; SUB lanes share the chunk partials of one
// (t, c) (each sums every SUB-th chunk in order, then a fixed xor tree), 32 timesteps per pass; thread 0 applies the T
// sequential running-stat updates of one reference forward from LDS.  Fixed summation order: deterministic.
// SUB = 8 for the few chunks snn_bn_stats writes, 32 for the hundreds of row tiles a convolution epilogue leaves.
template <int SUB>
__global__ __launch_bounds__(32 * SUB) void k_bn_stats_finalize_fused(
    const double* __restrict__ partial, int chunks, int rows_per_chunk, int T, int64_t M, int C,
    const float* __restrict__ gamma,
    const float* __restrict__ bias, float eps, float momentum, float* __restrict__ running_mean,
    float* __restrict__ running_var, int use_running, float* __restrict__ mean, float* __restrict__ invstd,
    float* __restrict__ alpha, float* __restrict__ beta) {
    __shared__ float sm_mean[32];
    __shared__ double sm_var[32];
    const int c = blockIdx.x;
    const int sub = threadIdx.x % SUB, tl = threadIdx.x / SUB;
    const bool update = !use_running && running_mean && running_var;
    float rm = 0.f, rv = 0.f;
    if (update && threadIdx.x == 0) {
        rm = running_mean[c];
        rv = running_var[c];
    }
    const float g = gamma ? gamma[c] : 1.0f;
    const float b = bias ? bias[c] : 0.0f;
    const double mom = (double)momentum;
    for (int tb = 0; tb < T; tb += 32) {
        const int t = tb + tl;
        double s = 0.0, q = 0.0;
        if (!use_running && t < T) {
            const int nk = chunks_of_step(chunks, rows_per_chunk, t, M);
            const double* base = partial;
            int k = sub;
            constexpr int U = SUB >= 32 ? 8 : 4;   // loads in flight (the loop is latency-bound), added in chunk order
            for (; k + (U - 1) * SUB < nk; k += U * SUB) {
                double2 p[U];
#pragma unroll
                for (int u = 0; u < U; ++u)
                    p[u] = *reinterpret_cast<const double2*>(base + snn_bn_partial_index(t, k + u * SUB, c, chunks, C));
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    s += p[u].x;
                    q += p[u].y;
                }
            }
            for (; k < nk; k += SUB) {
                const double2 p0 = *reinterpret_cast<const double2*>(base + snn_bn_partial_index(t, k, c, chunks, C));
                s += p0.x; q += p0.y;
            }
        }
        for (int stride = SUB / 2; stride >= 1; stride >>= 1) {
            s += __shfl_xor(s, stride, 64);
            q += __shfl_xor(q, stride, 64);
        }
        if (sub == 0 && t < T) {
            const int idx = t * C + c;
            float mu, is;
            if (use_running) {
                mu = running_mean[c];
                is = 1.0f / sqrtf(running_var[c] + eps);  // ATen eval path: invstd in fp32
            } else {
                const double n = (double)M;
                const double m = s / n;
                double var = q / n - m * m;
                if (var < 0.0) var = 0.0;
                mu = (float)m;
                is = (float)(1.0 / sqrt(var + (double)eps));
                sm_mean[tl] = mu;
                sm_var[tl] = (M > 1) ? var * n / (n - 1.0) : var;
            }
            mean[idx] = mu;
            invstd[idx] = is;
            const float a = is * g;
            alpha[idx] = a;
            beta[idx] = b - mu * a;
        }
        if (update) {
            __syncthreads();
            if (threadIdx.x == 0) {
                const int nt = T - tb < 32 ? T - tb : 32;
                for (int k = 0; k < nt; ++k) {
                    rm = (float)(mom * (double)sm_mean[k] + (1.0 - mom) * (double)rm);
                    rv = (float)(mom * sm_var[k] + (1.0 - mom) * (double)rv);
                }
            }
            __syncthreads();
        }
    }
    if (update && threadIdx.x == 0) {
        running_mean[c] = rm;
        running_var[c] = rv;
    }
}

// chunk partials -> sums[t][c][2] (the quantity a SyncBatchNorm exchange all-reduces, config.yaml:76)
__global__ void k_bn_stats_reduce(const double* __restrict__ partial, int chunks, int rows_per_chunk, int T, int64_t M,
                                  int C, double* __restrict__ sums) {
    int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= T * C) return;
    int t = idx / C, c = idx % C;
    double s = 0.0, q = 0.0;
    const int nk = chunks_of_step(chunks, rows_per_chunk, t, M);
    for (int k = 0; k < nk; ++k) {
        const double* src = partial + snn_bn_partial_index(t, k, c, chunks, C);
        s += src[0];
        q += src[1];
    }
    sums[(int64_t)idx * 2 + 0] = s;
    sums[(int64_t)idx * 2 + 1] = q;
}

// T sequential running-stat updates of one reference forward (one BatchNorm call per timestep).
__global__ void k_bn_running_update(const float* __restrict__ mean, const double* __restrict__ var_unbiased, int T,
                                    int C, float momentum, float* __restrict__ running_mean,
                                    float* __restrict__ running_var) {
    int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    float rm = running_mean[c], rv = running_var[c];
    const double mom = (double)momentum;
    for (int t = 0; t < T; ++t) {
        rm = (float)(mom * (double)mean[t * C + c] + (1.0 - mom) * (double)rm);
        rv = (float)(mom * var_unbiased[t * C + c] + (1.0 - mom) * (double)rv);
    }
    running_mean[c] = rm;
    running_var[c] = rv;
}

// ------------------------------------------------------------------------------------------
// Forward scan
// ------------------------------------------------------------------------------------------
// SAVE: 0 nothing for the backward pass; 1 the per-step state (vdec); 2 (LIF) the state (v, i) BEFORE every kCkpt-th step
// into vdec = ckpt[chunk][2][M][C]: the backward scan recomputes the steps of a chunk from it (k_lif_bwd_ckpt) instead
// of reading one saved value per step.  Half the saved-state memory of mode 1 at the same speed (forward faster,
// backward slower by about as much); opt-in from functional.LIF_CHECKPOINT_BYTES.
constexpr int kCkpt = 4;
#ifndef SNN_SCAN_PREFETCH
#define SNN_SCAN_PREFETCH 2   // steps of operands in flight ahead of the recurrence (forward scan)
#endif
// SB (SNN_SCAN_BF16_STORAGE): y, out, addend and vdec are bf16 tensors (pointers passed as float*, strides in elements)
template <int NEURON, int VEC, int SAVE, bool SB = false>
__global__ __launch_bounds__(kThreads) void k_affine_neuron_fwd(
    const float* __restrict__ y, int64_t ldy, const float* __restrict__ alpha, const float* __restrict__ beta,
    const float* __restrict__ v0, const float* __restrict__ i0, float* __restrict__ out, int64_t ldo,
    const float* __restrict__ addend, int64_t ld_add, float* __restrict__ vT, float* __restrict__ iT,
    float* __restrict__ vdec, int T, int64_t M, int C, snn_neuron_params p, int last_only) {
    // last_only (SNN_SCAN_LAST_STEP_ONLY): `out` is [M][ldo], only the last timestep's output is kept (the detection
    // head: soda.py:141-144 returns the predictions of the last step) - T-1 of T output stores never happen
    typedef typename Vec<VEC>::type V;
    static_assert(!SB || SAVE != 2, "the checkpointed scan keeps fp32 checkpoints: not combined with bf16 storage");
    const int cv = C / VEC;
    const int64_t total = M * cv;
    for (int64_t col = (int64_t)blockIdx.x * kThreads + threadIdx.x; col < total;
         col += (int64_t)gridDim.x * kThreads) {
        const int64_t m = col / cv;
        const int c = (int)(col % cv) * VEC;
        V v, i;
        if (NEURON != SNN_NEURON_NONE) {
            if (v0) v = Vec<VEC>::load(v0 + m * C + c);
            else {
#pragma unroll
                for (int j = 0; j < VEC; ++j) lane<VEC>(v, j) = p.v_leak;
            }
            if (i0) i = Vec<VEC>::load(i0 + m * C + c);
            else {
#pragma unroll
                for (int j = 0; j < VEC; ++j) lane<VEC>(i, j) = 0.0f;
            }
        }
        // The operands of a step - y, the BatchNorm affine (alpha, beta)[t][c], the shortcut - are requested kPrefetch steps
        // ahead of the recurrence and rotate through registers: with the loads inside the step every iteration waited for
        // its own alpha / beta loads (s_waitcnt vmcnt(0): a full memory latency per timestep, whatever was prefetched
        // before them).  Steps past the end re-read the last one.  The pipelined loop must be free of branches around its
        // memory operations (at a control-flow join the compiler's wait-count pass falls back to vmcnt(0)), so it exists
        // in the two forms the layer-major step uses - BatchNorm affine, all T outputs, with / without a shortcut - and
        // everything else (no affine, last step only) takes the plain loop with its run-time checks.
        auto time_loop = [&](auto piped_c, auto add_c, auto noout_c, auto last_c) {
        constexpr bool PIPED = decltype(piped_c)::value;       // affine present, ADD known, outputs: all steps or (LAST) one
        // LAST (PIPED only; SNN_SCAN_LAST_STEP_ONLY, the detection heads' LI + Tanh): nothing is stored inside the loop, the
        // last step's output once behind it - the plain loop below waited for every step's own loads, 32 dependent memory
        // round trips (66 us for the 30x38 head at 2.9 TB/s)
        constexpr bool LAST = decltype(last_c)::value;
        constexpr bool ADD = decltype(add_c)::value;
        // NOOUT (SNN_SCAN_SPIKES_FROM_VDEC; LIF without a shortcut, v_dec saved): no output tensor at all - the consumer
        // forms the spikes itself, z = (v_dec > v_th), while it reads the saved potentials (snn_conv1x1_spikes_*)
        constexpr bool NOOUT = decltype(noout_c)::value;
        constexpr int kPrefetch = PIPED ? SNN_SCAN_PREFETCH : 0;
        struct StepOps { V x, a, b, ad; };
        auto fetch_step = [&](int t) {
            const int tc = t < T ? t : T - 1;
            const int64_t row = (int64_t)tc * M + m;
            StepOps o;
            o.x = VecS<VEC, SB>::load_last(y, row * ldy + c);   // (the convolution's output: next read in the backward pass)
            if (PIPED || alpha) {
                o.a = Vec<VEC>::load(alpha + (int64_t)tc * C + c);
                o.b = Vec<VEC>::load(beta + (int64_t)tc * C + c);
            }
            if (PIPED ? ADD : addend != nullptr) o.ad = VecS<VEC, SB>::load_last(addend, row * ld_add + c);   // (as y)
            return o;
        };
        StepOps sq[kPrefetch > 0 ? kPrefetch : 1];
        if constexpr (kPrefetch > 0) {
#pragma unroll
            for (int k = 0; k < kPrefetch; ++k) sq[k] = fetch_step(k);
        }
        [[maybe_unused]] V o_keep;
        for (int t = 0; t < T; ++t) {
            const int64_t row = (int64_t)t * M + m;
            if (SAVE == 2 && NEURON == SNN_NEURON_LIF && (t % kCkpt) == 0) {
                float* ck = vdec + ((int64_t)(t / kCkpt) * 2 * M + m) * C + c;
                Vec<VEC>::store(ck, v);
                Vec<VEC>::store(ck + M * C, i);
            }
            StepOps cur;
            if constexpr (kPrefetch > 0) {
                cur = sq[0];
#pragma unroll
                for (int k = 0; k + 1 < kPrefetch; ++k) sq[k] = sq[k + 1];
                sq[kPrefetch - 1] = fetch_step(t + kPrefetch);
            } else {
                cur = fetch_step(t);
            }
            V x = cur.x;
            if (PIPED || alpha) {
#pragma unroll
                for (int j = 0; j < VEC; ++j) lane<VEC>(x, j) = lane<VEC>(x, j) * lane<VEC>(cur.a, j) + lane<VEC>(cur.b, j);
            }
            V o, vd;
#pragma unroll
            for (int j = 0; j < VEC; ++j) {
                float xj = lane<VEC>(x, j);
                if (NEURON == SNN_NEURON_NONE) {
                    lane<VEC>(o, j) = xj;
                } else {
                    float vj = lane<VEC>(v, j), ij = lane<VEC>(i, j);
                    if (NEURON == SNN_NEURON_SYNAPSE) {
                        const float tau = (xj > 0.0f) ? p.tau_sec : p.tau_dis;
                        const float p_new = vj + ((xj - vj) * tau) * p.dt;
                        float gsyn = p_new;
                        if (p.sigma != 0.0f) gsyn = (4.0f * p.sigma) * (p_new - p.sigma * (p_new * p_new));
                        lane<VEC>(v, j) = p_new;
                        lane<VEC>(o, j) = gsyn < 0.0f ? 0.0f : gsyn;
                        lane<VEC>(vd, j) = p_new;
                        continue;
                    }
                    float xin = xj;
                    if (NEURON == SNN_NEURON_SLI) {
                        xin = xj * (1.0f / (1.0f + expf(-(p.v_st - fabsf(vj)))));
                        lane<VEC>(vd, j) = vj;
                    }
                    float i_new = ij + xin;
                    float dv = p.c_mem * ((p.v_leak - vj) + i_new);
                    float v_dec = vj + dv;
                    float di = p.c_syn * i_new;
                    lane<VEC>(i, j) = i_new + di;
                    if (NEURON == SNN_NEURON_LIF) {
                        float u = v_dec - p.v_th;
                        float z = (u > 0.0f) ? 1.0f : 0.0f;
                        lane<VEC>(v, j) = (1.0f - z) * v_dec + z * p.v_reset;
                        lane<VEC>(o, j) = z;
                        lane<VEC>(vd, j) = v_dec;
                    } else {
                        lane<VEC>(v, j) = v_dec;
                        lane<VEC>(o, j) = (NEURON == SNN_NEURON_LI_TANH) ? tanhf(v_dec) : v_dec;
                    }
                }
            }
            if (PIPED ? ADD : addend != nullptr) {  // residual shortcut folded into the store
#pragma unroll
                for (int j = 0; j < VEC; ++j) lane<VEC>(o, j) += lane<VEC>(cur.ad, j);
            }
            if constexpr (LAST) {
                o_keep = o;   // stored once behind the loop (no branch around a memory operation inside it)
            } else if constexpr (!NOOUT) {
                if (PIPED || !last_only) VecS<VEC, SB>::store(out, row * ldo + c, o);
                else if (t == T - 1) VecS<VEC, SB>::store(out, m * ldo + c, o);
            }
            if (SAVE == 1 && (NEURON == SNN_NEURON_LIF || NEURON == SNN_NEURON_SLI || NEURON == SNN_NEURON_SYNAPSE)) {
                // (non-temporal where the access is one plain 16-byte store: nobody reads v_dec before the backward pass,
                // while `out` is the next convolution's operand and should be what stays in the caches)
                if constexpr (VEC == 4 && !SB && SNN_SCAN_NT_AUX != 0) __builtin_nontemporal_store(vd, reinterpret_cast<f32x4*>(vdec + row * C + c));
                else VecS<VEC, SB>::store(vdec, row * C + c, vd);
            }
        }
        if constexpr (LAST) VecS<VEC, SB>::store(out, m * ldo + c, o_keep);
        };
        if constexpr (VEC > 1 && SAVE != 2) {
            if (alpha && !last_only) {
                if (addend) time_loop(std::true_type{}, std::true_type{}, std::false_type{}, std::false_type{});
                else if (SAVE == 1 && NEURON == SNN_NEURON_LIF && !SB && out == nullptr)
                    time_loop(std::true_type{}, std::false_type{}, std::true_type{}, std::false_type{});
                else time_loop(std::true_type{}, std::false_type{}, std::false_type{}, std::false_type{});
            } else if (alpha && last_only && !addend) {
                time_loop(std::true_type{}, std::false_type{}, std::false_type{}, std::true_type{});
            } else {
                time_loop(std::false_type{}, std::false_type{}, std::false_type{}, std::false_type{});
            }
        } else {
            time_loop(std::false_type{}, std::false_type{}, std::false_type{}, std::false_type{});
        }
        if (NEURON != SNN_NEURON_NONE) {
            if (vT) Vec<VEC>::store(vT + m * C + c, v);
            if (iT) Vec<VEC>::store(iT + m * C + c, i);
        }
    }
}

// value of the partner lane l ^ 32 (valid in the UPPER 32 lanes) / l ^ 16 (valid in the odd 16-lane rows)
__device__ __forceinline__ float swap32_partner(float v) {
    const unsigned u = __builtin_bit_cast(unsigned, v);
    return __builtin_bit_cast(float, __builtin_amdgcn_permlane32_swap(u, u, false, false)[0]);
}
__device__ __forceinline__ float swap16_partner(float v) {
    const unsigned u = __builtin_bit_cast(unsigned, v);
    return __builtin_bit_cast(float, __builtin_amdgcn_permlane16_swap(u, u, false, false)[0]);
}

// Wave-level combine of the per-thread BatchNorm partial sums: lanes l and l ^ stride (stride a multiple of cvb)
// hold the same channels.  The two wide strides go through the permute-swap VALU instructions of gfx950
// (v_permlane32_swap / v_permlane16_swap) instead of ds_bpermute shuffles - one dependent LDS round trip per value and
// stride less in a latency-bound scan (36 -> 30.5 us on the 19x15 and 10x8 maps).  The sum therefore ends up in the
// TOP lanes of the wave (row 3, or the upper half when cvb = 32): wave_sum_owner() names them.  Both reverse scans use
// this pair, so their sums associate identically (the checkpointed scan is tested bit-equal to the plain one).
template <int VEC>
__device__ __forceinline__ void wave_sum_channels(float (&s1)[VEC], float (&s2)[VEC], int cvb) {
    if (cvb >= 64) return;
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
        s1[j] += swap32_partner(s1[j]);
        s2[j] += swap32_partner(s2[j]);
        if (cvb < 32) {
            s1[j] += swap16_partner(s1[j]);
            s2[j] += swap16_partner(s2[j]);
        }
        for (int stride = cvb; stride < 16; stride <<= 1) {   // (inside a 16-lane row)
            s1[j] += __shfl_xor(s1[j], stride, 64);
            s2[j] += __shfl_xor(s2[j], stride, 64);
        }
    }
}
__device__ __forceinline__ bool wave_sum_owner(int lane, int cvb) {
    const int own_lo = cvb >= 32 ? 64 - cvb : 48;   // (cvb >= 64: every lane owns its channels)
    return cvb >= 64 || (lane >= own_lo && lane < own_lo + cvb);
}

// ------------------------------------------------------------------------------------------
// Backward (reverse-time) scan with per-(t,c) partial sums for the BatchNorm backward.
// grid = (GX pixel groups, GY channel blocks).  A thread owns NP pixels x VEC channels; per timestep it
// sums its NP contributions in registers, lanes of a wave that share channels combine with xor-shuffles,
// and one lane per (wave, channel) adds into that wave's private LDS slab red[wave][T][cb][2] (plain
// read-modify-write, fixed order).  Waves and blocks are combined in fixed order afterwards, so the
// BatchNorm gradients are bitwise reproducible.  (Channel counts whose per-block group count is not a
// power of two fall back to LDS float atomics on one shared slab.)
// ------------------------------------------------------------------------------------------
#ifndef SNN_BWD_NP
#define SNN_BWD_NP 4
#endif
constexpr int kBwdNP = SNN_BWD_NP;
#ifndef SNN_BWD_SB_DEPTH
#define SNN_BWD_SB_DEPTH 3   // operand sets in flight in the bf16-storage reverse scan (2 or 3)
#endif
constexpr int kWaves = kThreads / 64;

struct BwdPlan {
    int vec, cvb, gy, gx, mode;  // mode 0: no sums, 1: ordered (shuffle + per-wave slabs), 2: LDS atomics
    size_t lds_bytes;
    int64_t rpb;                 // pixel rows per block
};

static bool is_pow2(int v) { return v > 0 && (v & (v - 1)) == 0; }

// (Measured and rejected: a smaller slab budget - 32 / 16 KiB, more blocks of fewer channels - for the mid-size maps whose
// 64 KiB plan leaves CUs idle.  The 30x38 x 128-channel scan alone went 81 -> 59 us in isolation, but inside the step the
// family average rose from 98 to 125 us (bf16 storage) and 147 to 151 us (fp32): narrower channel runs per pixel and a
// second round of blocks on the small maps cost more than the shorter serial chains gain.)
static BwdPlan bwd_plan(int T, int64_t M, int C, bool with_sums) {
    constexpr int lds_kib = 64;
    BwdPlan pl;
    pl.vec = (C % 4 == 0) ? 4 : 1;
    int cv = C / pl.vec;
    int cvb = cv < kThreads ? cv : kThreads;
    pl.mode = 0;
    pl.lds_bytes = 0;
    if (with_sums) {
        const bool ordered = is_pow2(cvb) || cvb >= 64;
        pl.mode = ordered ? 1 : 2;
        const int slabs = ordered ? kWaves : 1;
        // LDS budget: slabs * T * cb * 2 floats
        int64_t max_cvb = ((int64_t)lds_kib * 1024) / ((int64_t)slabs * T * 8 * pl.vec);
        if (max_cvb < 1) max_cvb = 1;
        if (cvb > max_cvb) {
            cvb = (int)max_cvb;
            if (ordered) {  // keep a power of two
                int p2 = 1;
                while (p2 * 2 <= cvb) p2 *= 2;
                cvb = p2;
            }
        }
        pl.lds_bytes = (size_t)slabs * T * cvb * pl.vec * 2 * sizeof(float);
    }
    pl.cvb = cvb;
    pl.gy = (int)snn_ceil_div(cv, cvb);
    int P = kThreads / cvb;
    // Every block owns a contiguous run of pixel rows (P pixels each) of EQUAL length, processed kBwdNP rows at a
    // time with the tail masked: all blocks are resident at once and finish together.  (A grid-stride loop over
    // kBwdNP-row groups left e.g. 713 groups on 512 blocks: 2 rounds for 1.4 rounds of work.)
    const int64_t rows = snn_ceil_div(M, (int64_t)P);
    // blocks resident per CU by LDS (160 KiB per CU; 64 KiB slabs: 2, 32 KiB: 4, 16 KiB: 8 = the wave limit)
    int64_t cap = with_sums ? (int64_t)(128 / lds_kib) * snn_num_cu() : snn_max_blocks();
    if (const char* force = snn_tuning_env("SNN_BWD_CAP")) cap = atoi(force) > 0 ? atoi(force) : cap;  // tuning aid
    cap = cap / pl.gy;
    if (cap < 1) cap = 1;
    const int64_t rpb = snn_ceil_div(rows, rows < cap ? rows : cap);
    pl.gx = (int)snn_ceil_div(rows, rpb);
    pl.rpb = rpb;
    return pl;
}

// BUF (VEC = 4 only): the per-timestep operands are addressed through raw buffer resources - one per tensor and
// timestep, lane offset -1 for lanes outside the tensor, so loads return zeros and stores are dropped by the hardware
// range check.  The time loop is then straight-line code.  With per-pixel `if (ok)` branches the compiler's waitcnt
// pass could not tell the prefetched loads of step t-1 from the ones step t needs and waited for ALL of them before
// every pixel (vmcnt(0)): the prefetch bought nothing and the kernel ran at 3.9 TB/s with the texture addresser 16 %
// busy.  Host-checked: one timestep of every tensor is < 2 GiB.
// SB (SNN_SCAN_BF16_STORAGE): g_out, state, y and gx are bf16 tensors (pointers passed as float*, strides in elements).
// YF (SNN_SCAN_SUMS_FROM_STATE; LIF, MODE 1, BUF, fp32 tensors, initial state (v_leak, 0)): y is NOT read.  The only use the
// scan has for y is the BatchNorm statistic sum(gx * y), and the neuron's input x[t] = alpha*y[t] + beta - which carries the
// same information, sum(gx * x) = alpha * sum(gx * y) + beta * sum(gx) - can be rebuilt from the saved potentials the scan
// reads anyway: with v[t-1] = (vd[t-1] > v_th ? v_reset : vd[t-1]) the forward step vd[t] = v[t-1] + c_mem*((v_leak - v[t-1])
// + i'[t]) gives i'[t], and x[t] = i'[t] - (i'[t-1] + c_syn*i'[t-1]).  Walking backwards, step t holds vd[t] and has kept
// vd[t+1]: it forms i'[t+1], with the i'[t+2] of the step before x[t+2], and adds gx[t+2] * x[t+2] (kept two steps) to the
// slab row of step t+2; x[1] and x[0] follow after the loop from the initial state.  4 of the 16 bytes per neuron-timestep
// are never read; the second value of a sums pair is then sum(gx * x) (snn_bn_bwd_finalize_from_state converts).
template <int NEURON, int VEC, int MODE, bool BUF, int NP, bool SB = false, bool YF = false>
__global__ __launch_bounds__(kThreads, YF ? 2 : 1) void k_affine_neuron_bwd(
    const float* __restrict__ g_out, int64_t ldg, const float* __restrict__ state, const float* __restrict__ y,
    int64_t ldy, const float* __restrict__ g_vT, const float* __restrict__ g_iT, const float* __restrict__ alpha,
    const float* __restrict__ beta, int apply_scale, float* __restrict__ gx, float* __restrict__ g_v0,
    float* __restrict__ g_i0, double* __restrict__ sums, int T, int64_t M, int C, int cvb, snn_neuron_params p,
    int last_only_or_lookback) {
    // last_only (SNN_SCAN_LAST_STEP_ONLY; LIF / LI / LI+Tanh): g_out (and LI+Tanh's saved output) are [M][..] tensors of
    // the LAST timestep; the output gradient of every earlier step is zero and nothing is read for it.
    // YF instances (never last_only) take SNN_SCAN_STATE_LOOKBACK in the same argument slot.
    const int last_only = YF ? 0 : last_only_or_lookback;
    [[maybe_unused]] const int lookback = YF ? last_only_or_lookback : 0;
    typedef typename Vec<VEC>::type V;
    constexpr int ES = SnnStore<SB>::ES;   // bytes per element of the activation tensors
    constexpr bool kNeedsX = (NEURON == SNN_NEURON_SLI || NEURON == SNN_NEURON_SYNAPSE);
    constexpr bool kNeedsState = (NEURON == SNN_NEURON_LIF || NEURON == SNN_NEURON_LI_TANH || kNeedsX);
    extern __shared__ __attribute__((aligned(16))) float red[];  // MODE 1: [wave][T][cb][2]; MODE 2: [T][cb][2]
    const int cv = C / VEC;
    const int P = kThreads / cvb;
    const int tid = threadIdx.x;
    const int cgl = tid % cvb, ps = tid / cvb;
    const int cg = blockIdx.y * cvb + cgl;
    const int cb = cvb * VEC;
    const bool lane_ok = (ps < P) && (cg < cv);
    const int c = lane_ok ? cg * VEC : 0;
    const int wave = tid >> 6;
    if (MODE != 0 && !BWD_ABL(2)) {
        const int n = (MODE == 1 ? kWaves : 1) * T * cb * 2;
        for (int k = tid; k < n; k += kThreads) red[k] = 0.0f;
        __syncthreads();
    }
    static_assert(!YF || (NEURON == SNN_NEURON_LIF && MODE == 1 && BUF && VEC == 4 && !SB), "sums from the saved state: LIF, ordered sums");
    const float one_m_cmem = 1.0f - p.c_mem;
    const float one_p_csyn = 1.0f + p.c_syn;
    [[maybe_unused]] const float inv_cmem = 1.0f / p.c_mem;
    const int64_t rows = (M + P - 1) / P;
    const int64_t rpb = (rows + gridDim.x - 1) / gridDim.x;  // pixel rows per block (see bwd_plan)
    const int64_t row_lo = (int64_t)blockIdx.x * rpb;
    const int64_t row_hi = row_lo + rpb < rows ? row_lo + rpb : rows;
    for (int64_t rb = row_lo; rb < row_hi; rb += NP) {
        int64_t mq[NP];
        bool ok[NP];
        V gv[NP], gi[NP];
        // YF: what a step keeps for the statistic of two steps later - vd[t+1], i'[t+2], gx[t+1], gx[t+2]
        [[maybe_unused]] V yf_vd1[YF ? NP : 1], yf_in2[YF ? NP : 1], yf_g1[YF ? NP : 1], yf_g2[YF ? NP : 1];
#pragma unroll
        for (int q = 0; q < NP; ++q) {
            mq[q] = (rb + q) * P + ps;
            ok[q] = lane_ok && rb + q < row_hi && mq[q] < M;
#pragma unroll
            for (int j = 0; j < VEC; ++j) lane<VEC>(gv[q], j) = lane<VEC>(gi[q], j) = 0.0f;
            if constexpr (YF) {
#pragma unroll
                for (int j = 0; j < VEC; ++j)
                    lane<VEC>(yf_vd1[q], j) = lane<VEC>(yf_in2[q], j) = lane<VEC>(yf_g1[q], j) = lane<VEC>(yf_g2[q], j) = 0.0f;
            }
            if (NEURON != SNN_NEURON_NONE && ok[q]) {
                if (g_vT) gv[q] = Vec<VEC>::load(g_vT + mq[q] * C + c);
                if (g_iT) gi[q] = Vec<VEC>::load(g_iT + mq[q] * C + c);
            }
        }
        // Two register sets: the loads of timestep t-1 are in flight while timestep t is processed.
        int og[NP], os[NP], oy[NP];  // BUF: byte offsets inside one timestep of g_out / (state, gx) / y; -1 = no pixel
        if constexpr (BUF) {
#pragma unroll
            for (int q = 0; q < NP; ++q) {
                og[q] = ok[q] ? (int)((mq[q] * ldg + c) * ES) : -1;
                os[q] = ok[q] ? (int)((mq[q] * C + c) * ES) : -1;
                oy[q] = ok[q] ? (int)((mq[q] * ldy + c) * ES) : -1;
            }
        }
        auto slab = [&](const float* base, int t, int64_t ld) {  // buffer resource of timestep t of a [T][M][ld] tensor
            char* b = reinterpret_cast<char*>(const_cast<float*>(base)) + (int64_t)t * M * ld * ES;
            return __builtin_amdgcn_make_buffer_rsrc(b, 0, (int)(M * ld * ES), 0x00020000);
        };
        auto slab_out = [&](const float* base, int t, int64_t ld) {  // g_out / saved output: all steps, or the last one only
            if (!last_only) return slab(base, t, ld);
            return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, t == T - 1 ? (int)(M * ld * ES) : 0,
                                                     0x00020000);   // zero records: every load returns 0
        };
        // Operand sets in flight hold what the loads return: fp32 quads, or (bf16 storage) the 8 raw bytes of 4 bf16 values -
        // half the registers, which pays for a THIRD set: with half the bytes per step, one step of prefetch no longer
        // covers the memory latency (3.0 - 4.5 TB/s measured with two sets against 5 TB/s on fp32 tensors).
        using R = typename std::conditional<(SB && BUF), snn_u32x2, V>::type;
        constexpr int DEPTH = (SB && BUF) ? SNN_BWD_SB_DEPTH : 2;
        auto widen = [](const R& r) -> V {
            if constexpr (SB && BUF) return snn_unpack_bf16x4(r);
            else return r;
        };
        auto bload = [&](const auto& rs, int off) -> R {   // 4 elements of a storage-type tensor (generic: BUF instances only)
            if constexpr (SB && BUF) return __builtin_bit_cast(snn_u32x2, __builtin_amdgcn_raw_buffer_load_b64(rs, off, 0, 0));
            else return __builtin_bit_cast(V, __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0));
        };
        auto bload_last = [&](const auto& rs, int off) -> R {   // the same for a tensor nobody reads again (aux 2 = nt)
            if constexpr (SB && BUF) return __builtin_bit_cast(snn_u32x2, __builtin_amdgcn_raw_buffer_load_b64(rs, off, 0, SNN_SCAN_NT_AUX));
            else return __builtin_bit_cast(V, __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, SNN_SCAN_NT_AUX));
        };
        // eval-mode BatchNorm scale alpha[t][c] (apply_scale): fetched with the operand set - a load inside a branch of the
        // time loop makes the compiler's wait-count pass fall back to small vmcnt values for the whole step (it cannot
        // know whether the load was issued), which stalls on the prefetched set.  Zero records when unused.
        [[maybe_unused]] __amdgpu_buffer_rsrc_t rsc;
        if constexpr (BUF)
            rsc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(apply_scale ? alpha : g_out), 0,
                                                    apply_scale ? (int)((int64_t)T * C * 4) : 0, 0x00020000);
        auto fetch = [&](int t, R (&go)[NP], R (&st)[NP], R (&yv)[NP], V& sc) {
            if constexpr (BUF) {
                sc = __builtin_bit_cast(V, __builtin_amdgcn_raw_buffer_load_b128(rsc, lane_ok ? (t * C + c) * 4 : -1, 0, 0));
                const __amdgpu_buffer_rsrc_t rg = slab_out(g_out, t, ldg);
#pragma unroll
                for (int q = 0; q < NP; ++q)
                    go[q] = bload_last(rg, og[q]);
                if (kNeedsState) {
                    const __amdgpu_buffer_rsrc_t rs = (NEURON == SNN_NEURON_LI_TANH) ? slab_out(state, t, C) : slab(state, t, C);
#pragma unroll
                    for (int q = 0; q < NP; ++q)
                        st[q] = bload_last(rs, os[q]);
                }
                if ((MODE != 0 && !YF) || kNeedsX) {
                    const __amdgpu_buffer_rsrc_t ry = slab(y, t, ldy);
#pragma unroll
                    for (int q = 0; q < NP; ++q)
                        yv[q] = bload(ry, oy[q]);
                }
            } else {
#pragma unroll
                for (int q = 0; q < NP; ++q) {
                    if (ok[q]) {
                        const int64_t row = (int64_t)t * M + mq[q];
                        const int64_t row_o = last_only ? mq[q] : row;
                        const bool live = !last_only || t == T - 1;
#pragma unroll
                        for (int j = 0; j < VEC; ++j) lane<VEC>(go[q], j) = lane<VEC>(st[q], j) = 0.0f;
                        if (live) go[q] = VecS<VEC, SB>::load(g_out, row_o * ldg + c);
                        if (kNeedsState) {
                            if (NEURON != SNN_NEURON_LI_TANH) st[q] = VecS<VEC, SB>::load(state, row * C + c);
                            else if (live) st[q] = VecS<VEC, SB>::load(state, row_o * C + c);
                        }
                        if (MODE != 0 || kNeedsX) yv[q] = VecS<VEC, SB>::load(y, row * ldy + c);
                    }
                }
            }
        };
        auto process = [&](int t, R (&go_r)[NP], R (&st_r)[NP], R (&yv_r)[NP], const V& sc_set) {
            float s1[VEC], s2[VEC];
#pragma unroll
            for (int j = 0; j < VEC; ++j) s1[j] = s2[j] = 0.0f;
            V xa[NP];
            if (kNeedsX) {  // x[t] = y[t]*alpha + beta, as the forward computed it
                V a1, b1;
                if (alpha) {
                    a1 = Vec<VEC>::load(alpha + (int64_t)t * C + c);
                    b1 = Vec<VEC>::load(beta + (int64_t)t * C + c);
                }
#pragma unroll
                for (int q = 0; q < NP; ++q) {
                    V yq = widen(yv_r[q]);
#pragma unroll
                    for (int j = 0; j < VEC; ++j)
                        lane<VEC>(xa[q], j) = alpha ? lane<VEC>(yq, j) * lane<VEC>(a1, j) + lane<VEC>(b1, j) : lane<VEC>(yq, j);
                }
            }
            [[maybe_unused]] __amdgpu_buffer_rsrc_t rgx;
            if constexpr (BUF) rgx = slab(gx, t, C);
#pragma unroll
            for (int q = 0; q < NP; ++q) {
                if (!BUF && !ok[q]) continue;  // BUF: lanes without a pixel compute on zeros, their store is dropped
                const int64_t row = (int64_t)t * M + mq[q];
                V g;
                V go_q = widen(go_r[q]);
                [[maybe_unused]] V st_q, yv_q;
                if (kNeedsState) st_q = widen(st_r[q]);
                if (MODE != 0 && !YF) yv_q = widen(yv_r[q]);
#pragma unroll
                for (int j = 0; j < VEC; ++j) {
                    float goj = lane<VEC>(go_q, j);
                    if (NEURON == SNN_NEURON_NONE) {
                        lane<VEC>(g, j) = goj;
                    } else if (NEURON == SNN_NEURON_LIF) {
                        float vd = lane<VEC>(st_q, j);
                        float u = vd - p.v_th;
                        float z = (u > 0.0f) ? 1.0f : 0.0f;
                        float den = p.alpha * fabsf(u) + 1.0f;
                        // bf16 storage: the 1-ulp hardware reciprocal (the result is rounded to 8 bits on its way out; the
                        // IEEE division is ten VALU instructions and made this instance VALU-bound at 3.3 TB/s)
                        float sg = SB ? __builtin_amdgcn_rcpf(den * den) : 1.0f / (den * den);
                        float gvj = lane<VEC>(gv[q], j);
                        float gz = goj + gvj * (p.v_reset - vd);
                        float g_vd = gvj * (1.0f - z) + gz * sg;
                        float g_in = p.c_mem * g_vd + lane<VEC>(gi[q], j) * one_p_csyn;
                        lane<VEC>(gv[q], j) = g_vd * one_m_cmem;
                        lane<VEC>(gi[q], j) = g_in;
                        lane<VEC>(g, j) = g_in;
                        if constexpr (YF) {   // x[t+2] from vd[t+2], vd[t+1], vd[t] (zero gradient slots until t+2 exists)
                            const float vprev = (u > 0.0f) ? p.v_reset : vd;                       // v[t]
                            const float in1 = (lane<VEC>(yf_vd1[q], j) - vprev) * inv_cmem - (p.v_leak - vprev);   // i'[t+1]
                            const float x2 = lane<VEC>(yf_in2[q], j) - (in1 + p.c_syn * in1);
                            s2[j] += lane<VEC>(yf_g2[q], j) * x2;
                            lane<VEC>(yf_in2[q], j) = in1;
                            lane<VEC>(yf_vd1[q], j) = vd;
                            lane<VEC>(yf_g2[q], j) = lane<VEC>(yf_g1[q], j);
                            lane<VEC>(yf_g1[q], j) = g_in;
                        }
                    } else if (NEURON == SNN_NEURON_SLI) {
                        const float v_old = lane<VEC>(st_q, j);
                        const float xj = lane<VEC>(xa[q], j);
                        const float s = 1.0f / (1.0f + expf(-(p.v_st - fabsf(v_old))));
                        const float sgn = (v_old > 0.0f) ? 1.0f : ((v_old < 0.0f) ? -1.0f : 0.0f);
                        float g_vn = goj + lane<VEC>(gv[q], j);
                        float g_ij = p.c_mem * g_vn + lane<VEC>(gi[q], j) * one_p_csyn;
                        lane<VEC>(gv[q], j) = g_vn * one_m_cmem + g_ij * xj * (s * (1.0f - s)) * (-sgn);
                        lane<VEC>(gi[q], j) = g_ij;
                        lane<VEC>(g, j) = g_ij * s;
                    } else if (NEURON == SNN_NEURON_SYNAPSE) {
                        const float p_new = lane<VEC>(st_q, j);
                        const float xj = lane<VEC>(xa[q], j);
                        const float td = ((xj > 0.0f) ? p.tau_sec : p.tau_dis) * p.dt;
                        float gpre = p_new, dg = 1.0f;
                        if (p.sigma != 0.0f) {
                            gpre = (4.0f * p.sigma) * (p_new - p.sigma * (p_new * p_new));
                            dg = (4.0f * p.sigma) * (1.0f - 2.0f * p.sigma * p_new);
                        }
                        const float g_pn = goj * ((gpre >= 0.0f) ? dg : 0.0f) + lane<VEC>(gv[q], j);
                        lane<VEC>(gv[q], j) = g_pn * (1.0f - td);
                        lane<VEC>(g, j) = g_pn * td;
                    } else {
                        float d = 1.0f;
                        if (NEURON == SNN_NEURON_LI_TANH) {
                            float o = lane<VEC>(st_q, j);
                            d = 1.0f - o * o;
                        }
                        float g_vn = goj * d + lane<VEC>(gv[q], j);
                        float g_in = p.c_mem * g_vn + lane<VEC>(gi[q], j) * one_p_csyn;
                        lane<VEC>(gv[q], j) = g_vn * one_m_cmem;
                        lane<VEC>(gi[q], j) = g_in;
                        lane<VEC>(g, j) = g_in;
                    }
                    if (MODE != 0) {
                        s1[j] += lane<VEC>(g, j);
                        if constexpr (!YF) s2[j] += lane<VEC>(g, j) * lane<VEC>(yv_q, j);
                    }
                }
                if (apply_scale) {
                    V sc = sc_set;
                    if constexpr (!BUF) sc = Vec<VEC>::load(alpha + (int64_t)t * C + c);
#pragma unroll
                    for (int j = 0; j < VEC; ++j) lane<VEC>(g, j) = lane<VEC>(g, j) * lane<VEC>(sc, j);
                }
                if constexpr (BUF && SB)
                    __builtin_amdgcn_raw_buffer_store_b64(
                        __builtin_bit_cast(decltype(__builtin_amdgcn_raw_buffer_load_b64(rgx, 0, 0, 0)), snn_pack_bf16x4(g)), rgx,
                        os[q], 0, 0);
                else if constexpr (BUF)
                    __builtin_amdgcn_raw_buffer_store_b128(
                        __builtin_bit_cast(decltype(__builtin_amdgcn_raw_buffer_load_b128(rgx, 0, 0, 0)), g), rgx, os[q],
                        0, 0);
                else
                    VecS<VEC, SB>::store(gx, row * C + c, g);
            }
            if (MODE == 1) {
                if (!BWD_ABL(1)) wave_sum_channels<VEC>(s1, s2, cvb);
                if (wave_sum_owner(tid & 63, cvb) && lane_ok && !BWD_ABL(0)) {
                    float* r = red + (((int64_t)wave * T + t) * cb + cgl * VEC) * 2;
                    if constexpr (YF) {   // s2 belongs to step t + 2
#pragma unroll
                        for (int j = 0; j < VEC; ++j) r[j * 2 + 0] += s1[j];
                        if (t + 2 < T) {
#pragma unroll
                            for (int j = 0; j < VEC; ++j) r[(2 * cb + j) * 2 + 1] += s2[j];
                        }
                    } else {
#pragma unroll
                        for (int j = 0; j < VEC; ++j) {
                            r[j * 2 + 0] += s1[j];
                            r[j * 2 + 1] += s2[j];
                        }
                    }
                }
            } else if (MODE == 2) {
                if (lane_ok) {
                    float* r = red + ((int64_t)t * cb + cgl * VEC) * 2;
#pragma unroll
                    for (int j = 0; j < VEC; ++j) {
                        atomicAdd(r + j * 2 + 0, s1[j]);
                        atomicAdd(r + j * 2 + 1, s2[j]);
                    }
                }
            }
                };
        R goA[NP], stA[NP], yvA[NP], goB[NP], stB[NP], yvB[NP];
        V scA, scB;
        // steps below 0 re-read step 0: no branch around the loads of the BUF form (see the note on `rsc`)
        auto fetch0 = [&](int t, R (&go)[NP], R (&st)[NP], R (&yv)[NP], V& sc) { fetch(t > 0 ? t : 0, go, st, yv, sc); };
        if constexpr (DEPTH == 3) {
            // three sets, two steps ahead
            R goC[NP], stC[NP], yvC[NP];
            V scC;
            fetch(T - 1, goA, stA, yvA, scA);
            fetch0(T - 2, goB, stB, yvB, scB);
            for (int t = T - 1; t >= 0; t -= 3) {
                fetch0(t - 2, goC, stC, yvC, scC);
                process(t, goA, stA, yvA, scA);
                if (t >= 1) {
                    fetch0(t - 3, goA, stA, yvA, scA);
                    process(t - 1, goB, stB, yvB, scB);
                }
                if (t >= 2) {
                    fetch0(t - 4, goB, stB, yvB, scB);
                    process(t - 2, goC, stC, yvC, scC);
                }
            }
        } else if constexpr (BUF) {
            fetch(T - 1, goA, stA, yvA, scA);
            for (int t = T - 1; t >= 0; t -= 2) {
                fetch0(t - 1, goB, stB, yvB, scB);
                process(t, goA, stA, yvA, scA);
                if (t >= 1) {
                    fetch0(t - 2, goA, stA, yvA, scA);
                    process(t - 1, goB, stB, yvB, scB);
                }
            }
        } else {
            fetch(T - 1, goA, stA, yvA, scA);
            for (int t = T - 1; t >= 0; t -= 2) {
                if (t >= 1) fetch(t - 1, goB, stB, yvB, scB);
                process(t, goA, stA, yvA, scA);
                if (t >= 1) {
                    if (t >= 2) fetch(t - 2, goA, stA, yvA, scA);
                    process(t - 1, goB, stB, yvB, scB);
                }
            }
        }
        if constexpr (YF) {
            // the two statistics the loop still owes: x[1] = i'[1] - i[0] and x[0] = i'[0] - i[-1] (after the loop: vd1 =
            // vd[0], in2 = i'[1], g1 = gx[0], g2 = gx[1]).  The state before step 0 is the initial one, (v_leak, 0) - or,
            // for a segment of a longer scan (SNN_SCAN_STATE_LOOKBACK), what the two saved potentials in front of the
            // segment say: v[-1] from vd[-1], i[-1] from vd[-1] and vd[-2].
            V vm1[NP], vm2[NP];
            if (lookback) {
                const __amdgpu_buffer_rsrc_t rs1 = slab(state, -1, C), rs2 = slab(state, -2, C);
#pragma unroll
                for (int q = 0; q < NP; ++q) {
                    vm1[q] = bload(rs1, os[q]);
                    vm2[q] = bload(rs2, os[q]);
                }
            }
            float sa[VEC], sb0[VEC];
#pragma unroll
            for (int j = 0; j < VEC; ++j) sa[j] = sb0[j] = 0.0f;
#pragma unroll
            for (int q = 0; q < NP; ++q) {
#pragma unroll
                for (int j = 0; j < VEC; ++j) {
                    float vp1 = p.v_leak, ip = 0.0f;
                    if (lookback) {
                        const float a = lane<VEC>(vm1[q], j), b = lane<VEC>(vm2[q], j);
                        const float vp2 = (b - p.v_th > 0.0f) ? p.v_reset : b;
                        const float im1 = (a - vp2) * inv_cmem - (p.v_leak - vp2);
                        ip = im1 + p.c_syn * im1;
                        vp1 = (a - p.v_th > 0.0f) ? p.v_reset : a;
                    }
                    const float in0 = (lane<VEC>(yf_vd1[q], j) - vp1) * inv_cmem - (p.v_leak - vp1);
                    const float x1 = lane<VEC>(yf_in2[q], j) - (in0 + p.c_syn * in0);
                    sa[j] += lane<VEC>(yf_g2[q], j) * x1;
                    sb0[j] += lane<VEC>(yf_g1[q], j) * (in0 - ip);
                }
            }
            wave_sum_channels<VEC>(sa, sb0, cvb);
            if (wave_sum_owner(tid & 63, cvb) && lane_ok) {
                float* r = red + (((int64_t)wave * T) * cb + cgl * VEC) * 2;
#pragma unroll
                for (int j = 0; j < VEC; ++j) r[j * 2 + 1] += sb0[j];
                if (T >= 2) {
#pragma unroll
                    for (int j = 0; j < VEC; ++j) r[(cb + j) * 2 + 1] += sa[j];
                }
            }
        }
        if (NEURON != SNN_NEURON_NONE) {
#pragma unroll
            for (int q = 0; q < NP; ++q) {
                if (!ok[q]) continue;
                if (g_v0) Vec<VEC>::store(g_v0 + mq[q] * C + c, gv[q]);
                if (g_i0) Vec<VEC>::store(g_i0 + mq[q] * C + c, gi[q]);
            }
        }
    }
    if (MODE != 0 && !BWD_ABL(2)) {
        __syncthreads();
        // sums[bx][t][c][2], stored as fp32 (the block sums ARE fp32; doubles would only double the bytes the
        // finalize kernel reads back: up to 512 x T x C x 2 values per layer)
        float* dst = reinterpret_cast<float*>(sums) + (int64_t)blockIdx.x * T * C * 2;
        const int c_lo = blockIdx.y * cb;
        for (int k = tid; k < T * cb; k += kThreads) {
            int t = k / cb, cl = k % cb;
            if (c_lo + cl < C) {
                float a = 0.0f, b = 0.0f;
                if (MODE == 1) {
#pragma unroll
                    for (int w = 0; w < kWaves; ++w) {
                        a += red[((int64_t)w * T * cb + k) * 2 + 0];
                        b += red[((int64_t)w * T * cb + k) * 2 + 1];
                    }
                } else {
                    a = red[k * 2 + 0];
                    b = red[k * 2 + 1];
                }
                dst[((int64_t)t * C + c_lo + cl) * 2 + 0] = a;
                dst[((int64_t)t * C + c_lo + cl) * 2 + 1] = b;
            }
        }
    }
}

// LIF backward scan from checkpoints (forward SAVE mode 2).  Same grid, block roles, reduction order and outputs as
// k_affine_neuron_bwd<LIF>; per chunk of kCkpt steps (last chunk first) a thread re-runs the forward recurrence from
// the chunk's saved (v, i) - the very expressions of k_affine_neuron_fwd, so the recomputed membrane values are the
// forward's bit for bit - and then walks the chunk backwards.  The next chunk's loads are issued before the current
// chunk is processed.
constexpr int kCkptNP = 2;
template <int VEC, int MODE>
__global__ __launch_bounds__(kThreads) void k_lif_bwd_ckpt(
    const float* __restrict__ g_out, int64_t ldg, const float* __restrict__ ckpt, const float* __restrict__ y,
    int64_t ldy, const float* __restrict__ g_vT, const float* __restrict__ g_iT, const float* __restrict__ alpha,
    const float* __restrict__ beta, int apply_scale, float* __restrict__ gx, float* __restrict__ g_v0,
    float* __restrict__ g_i0, double* __restrict__ sums, int T, int64_t M, int C, int cvb, snn_neuron_params p) {
    typedef typename Vec<VEC>::type V;
    constexpr int NP = kCkptNP;
    constexpr int K = kCkpt;
    extern __shared__ __attribute__((aligned(16))) float red[];
    const int cv = C / VEC;
    const int P = kThreads / cvb;
    const int tid = threadIdx.x;
    const int cgl = tid % cvb, ps = tid / cvb;
    const int cg = blockIdx.y * cvb + cgl;
    const int cb = cvb * VEC;
    const bool lane_ok = (ps < P) && (cg < cv);
    const int c = lane_ok ? cg * VEC : 0;
    const int wave = tid >> 6;
    if (MODE != 0) {
        const int n = (MODE == 1 ? kWaves : 1) * T * cb * 2;
        for (int k = tid; k < n; k += kThreads) red[k] = 0.0f;
        __syncthreads();
    }
    const float one_m_cmem = 1.0f - p.c_mem;
    const float one_p_csyn = 1.0f + p.c_syn;
    const int64_t rows = (M + P - 1) / P;
    const int64_t rpb = (rows + gridDim.x - 1) / gridDim.x;
    const int64_t row_lo = (int64_t)blockIdx.x * rpb;
    const int64_t row_hi = row_lo + rpb < rows ? row_lo + rpb : rows;
    const int nchunks = (T + K - 1) / K;
    struct Set {
        V v[NP], i[NP], yv[K][NP], go[K][NP];
    };
    for (int64_t rb = row_lo; rb < row_hi; rb += NP) {
        int64_t mq[NP];
        bool ok[NP];
        V gv[NP], gi[NP];
#pragma unroll
        for (int q = 0; q < NP; ++q) {
            mq[q] = (rb + q) * P + ps;
            ok[q] = lane_ok && rb + q < row_hi && mq[q] < M;
#pragma unroll
            for (int j = 0; j < VEC; ++j) lane<VEC>(gv[q], j) = lane<VEC>(gi[q], j) = 0.0f;
            if (ok[q]) {
                if (g_vT) gv[q] = Vec<VEC>::load(g_vT + mq[q] * C + c);
                if (g_iT) gi[q] = Vec<VEC>::load(g_iT + mq[q] * C + c);
            }
        }
        auto fetch = [&](int ch, Set& s) {
            const int t0 = ch * K;
#pragma unroll
            for (int q = 0; q < NP; ++q) {
                if (!ok[q]) continue;
                const float* ck = ckpt + ((int64_t)ch * 2 * M + mq[q]) * C + c;
                s.v[q] = Vec<VEC>::load(ck);
                s.i[q] = Vec<VEC>::load(ck + M * C);
#pragma unroll
                for (int k = 0; k < K; ++k) {
                    if (t0 + k < T) {
                        const int64_t row = (int64_t)(t0 + k) * M + mq[q];
                        s.yv[k][q] = Vec<VEC>::load(y + row * ldy + c);
                        s.go[k][q] = Vec<VEC>::load(g_out + row * ldg + c);
                    }
                }
            }
        };
        auto process = [&](int ch, Set& s) {
            const int t0 = ch * K;
            V vd[K][NP];
            // forward recurrence of the chunk (k_affine_neuron_fwd, LIF branch)
#pragma unroll
            for (int k = 0; k < K; ++k) {
                if (t0 + k >= T) continue;
                V a1, b1;
                if (alpha) {
                    a1 = Vec<VEC>::load(alpha + (int64_t)(t0 + k) * C + c);
                    b1 = Vec<VEC>::load(beta + (int64_t)(t0 + k) * C + c);
                }
#pragma unroll
                for (int q = 0; q < NP; ++q)
#pragma unroll
                    for (int j = 0; j < VEC; ++j) {
                        float xj = lane<VEC>(s.yv[k][q], j);
                        if (alpha) xj = xj * lane<VEC>(a1, j) + lane<VEC>(b1, j);
                        const float vj = lane<VEC>(s.v[q], j), ij = lane<VEC>(s.i[q], j);
                        const float i_new = ij + xj;
                        const float dv = p.c_mem * ((p.v_leak - vj) + i_new);
                        const float v_dec = vj + dv;
                        const float di = p.c_syn * i_new;
                        lane<VEC>(s.i[q], j) = i_new + di;
                        const float u = v_dec - p.v_th;
                        const float z = (u > 0.0f) ? 1.0f : 0.0f;
                        lane<VEC>(s.v[q], j) = (1.0f - z) * v_dec + z * p.v_reset;
                        lane<VEC>(vd[k][q], j) = v_dec;
                    }
            }
            // reverse-time walk of the chunk (k_affine_neuron_bwd, LIF branch)
#pragma unroll
            for (int k = K - 1; k >= 0; --k) {
                const int t = t0 + k;
                if (t >= T) continue;
                float s1[VEC], s2[VEC];
#pragma unroll
                for (int j = 0; j < VEC; ++j) s1[j] = s2[j] = 0.0f;
#pragma unroll
                for (int q = 0; q < NP; ++q) {
                    if (!ok[q]) continue;
                    const int64_t row = (int64_t)t * M + mq[q];
                    V g;
#pragma unroll
                    for (int j = 0; j < VEC; ++j) {
                        const float goj = lane<VEC>(s.go[k][q], j);
                        const float vdj = lane<VEC>(vd[k][q], j);
                        const float u = vdj - p.v_th;
                        const float z = (u > 0.0f) ? 1.0f : 0.0f;
                        const float den = p.alpha * fabsf(u) + 1.0f;
                        const float sg = 1.0f / (den * den);
                        const float gvj = lane<VEC>(gv[q], j);
                        const float gz = goj + gvj * (p.v_reset - vdj);
                        const float g_vd = gvj * (1.0f - z) + gz * sg;
                        const float g_in = p.c_mem * g_vd + lane<VEC>(gi[q], j) * one_p_csyn;
                        lane<VEC>(gv[q], j) = g_vd * one_m_cmem;
                        lane<VEC>(gi[q], j) = g_in;
                        lane<VEC>(g, j) = g_in;
                        if (MODE != 0) {
                            s1[j] += g_in;
                            s2[j] += g_in * lane<VEC>(s.yv[k][q], j);
                        }
                    }
                    if (apply_scale) {
                        V sc = Vec<VEC>::load(alpha + (int64_t)t * C + c);
#pragma unroll
                        for (int j = 0; j < VEC; ++j) lane<VEC>(g, j) = lane<VEC>(g, j) * lane<VEC>(sc, j);
                    }
                    Vec<VEC>::store(gx + row * C + c, g);
                }
                if (MODE == 1) {
                    wave_sum_channels<VEC>(s1, s2, cvb);
                    if (wave_sum_owner(tid & 63, cvb) && lane_ok) {
                        float* r = red + (((int64_t)wave * T + t) * cb + cgl * VEC) * 2;
#pragma unroll
                        for (int j = 0; j < VEC; ++j) {
                            r[j * 2 + 0] += s1[j];
                            r[j * 2 + 1] += s2[j];
                        }
                    }
                } else if (MODE == 2) {
                    if (lane_ok) {
                        float* r = red + ((int64_t)t * cb + cgl * VEC) * 2;
#pragma unroll
                        for (int j = 0; j < VEC; ++j) {
                            atomicAdd(r + j * 2 + 0, s1[j]);
                            atomicAdd(r + j * 2 + 1, s2[j]);
                        }
                    }
                }
            }
        };
        Set A, B;
        fetch(nchunks - 1, A);
        for (int ch = nchunks - 1; ch >= 0; ch -= 2) {
            if (ch >= 1) fetch(ch - 1, B);
            process(ch, A);
            if (ch >= 1) {
                if (ch >= 2) fetch(ch - 2, A);
                process(ch - 1, B);
            }
        }
#pragma unroll
        for (int q = 0; q < NP; ++q) {
            if (!ok[q]) continue;
            if (g_v0) Vec<VEC>::store(g_v0 + mq[q] * C + c, gv[q]);
            if (g_i0) Vec<VEC>::store(g_i0 + mq[q] * C + c, gi[q]);
        }
    }
    if (MODE != 0) {
        __syncthreads();
        float* dst = reinterpret_cast<float*>(sums) + (int64_t)blockIdx.x * T * C * 2;
        const int c_lo = blockIdx.y * cb;
        for (int k = tid; k < T * cb; k += kThreads) {
            int t = k / cb, cl = k % cb;
            if (c_lo + cl < C) {
                float a = 0.0f, b = 0.0f;
                if (MODE == 1) {
#pragma unroll
                    for (int w = 0; w < kWaves; ++w) {
                        a += red[((int64_t)w * T * cb + k) * 2 + 0];
                        b += red[((int64_t)w * T * cb + k) * 2 + 1];
                    }
                } else {
                    a = red[k * 2 + 0];
                    b = red[k * 2 + 1];
                }
                dst[((int64_t)t * C + c_lo + cl) * 2 + 0] = a;
                dst[((int64_t)t * C + c_lo + cl) * 2 + 1] = b;
            }
        }
    }
}

// reduce block partials -> raw[t][c] = (sum gx, sum gx*y).  32 lanes per (t,c): lane k sums blocks k, k+32, ...
// then a fixed xor tree combines the lanes.  `raw` must not alias the partial buffer (fp32 partials, fp64 result).
// from_state (the scan ran with SNN_SCAN_SUMS_FROM_STATE): the second partial is sum(gx * x); raw still receives sum(gx * y)
// = mean * sum(gx) + (sum(gx * x) - bias * sum(gx)) / (gamma * invstd) - what the all-reduce and k_bn_bwd_coef expect - and
// the sum itself, from gx and y, for a channel whose gamma is exactly 0 (see k_bn_bwd_finalize_fused).
__device__ __forceinline__ double sum_gx_y_from_state(double s1, double p, double mu, double is, double gam, double bnb) {
    return mu * s1 + (p - bnb * s1) / (gam * is);
}

__global__ __launch_bounds__(256) void k_bn_bwd_reduce(const double* __restrict__ sums_, int gx_blocks, int T, int C,
                                                       double* __restrict__ raw, int from_state, int64_t M,
                                                       const float* __restrict__ gamma, const float* __restrict__ bn_bias,
                                                       const float* __restrict__ mean, const float* __restrict__ invstd,
                                                       const float* __restrict__ gx, const float* __restrict__ y,
                                                       int64_t ldy) {
    const float* __restrict__ sums = reinterpret_cast<const float*>(sums_);
    const int sub = threadIdx.x & 31;
    const int idx = blockIdx.x * (blockDim.x / 32) + (threadIdx.x >> 5);
    const bool live = idx < T * C;
    const int c = live ? idx % C : 0, t = live ? idx / C : 0;
    const double gam = (double)((from_state && gamma) ? gamma[c] : 1.0f);
    const bool direct = from_state && gam == 0.0;   // uniform over the 32 lanes of an (t, c)
    double s1 = 0.0, sy = 0.0;
    if (live) {
        for (int b = sub; b < gx_blocks; b += 32) {
            const float2 v = *reinterpret_cast<const float2*>(sums + ((int64_t)b * T * C + idx) * 2);
            s1 += (double)v.x;
            sy += (double)v.y;
        }
        if (direct) {
            sy = 0.0;
            for (int64_t m = sub; m < M; m += 32)
                sy += (double)gx[((int64_t)t * M + m) * C + c] * (double)y[((int64_t)t * M + m) * ldy + c];
        }
    }
    for (int stride = 16; stride >= 1; stride >>= 1) {
        s1 += __shfl_xor(s1, stride, 64);
        sy += __shfl_xor(sy, stride, 64);
    }
    if (!live || sub != 0) return;
    if (from_state && !direct)
        sy = sum_gx_y_from_state(s1, sy, (double)mean[idx], (double)invstd[idx], gam, (double)(bn_bias ? bn_bias[c] : 0.0f));
    raw[(int64_t)idx * 2 + 0] = s1;
    raw[(int64_t)idx * 2 + 1] = sy;
}

// raw sums over M pixels (all-reduced over the ranks under SyncBatchNorm) -> backward coefficients;
// raw_local (this rank's sums) -> (sum gx, sum gx*xhat) for the parameter gradients, written to `param_sums`
__global__ void k_bn_bwd_coef(const double* __restrict__ raw, const double* __restrict__ raw_local, int T, int64_t M,
                              int C, const float* __restrict__ gamma, const float* __restrict__ mean,
                              const float* __restrict__ invstd, float* __restrict__ coefA, float* __restrict__ coefB,
                              float* __restrict__ coefC, double* __restrict__ param_sums) {
    int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= T * C) return;
    const int c = idx % C;
    const double mu = (double)mean[idx], is = (double)invstd[idx];
    const double s1 = raw[(int64_t)idx * 2 + 0], sy = raw[(int64_t)idx * 2 + 1];
    const double l1 = raw_local[(int64_t)idx * 2 + 0], ly = raw_local[(int64_t)idx * 2 + 1];
    const double s2 = is * (sy - mu * s1);  // sum gx * xhat
    const double n = (double)M;
    const double a = (double)(gamma ? gamma[c] : 1.0f) * is;
    const double m1 = s1 / n, m2 = s2 / n;
    coefA[idx] = (float)a;
    coefB[idx] = (float)(-a * is * m2);
    coefC[idx] = (float)(-a * m1 + a * is * mu * m2);
    param_sums[(int64_t)idx * 2 + 0] = l1;
    param_sums[(int64_t)idx * 2 + 1] = is * (ly - mu * l1);
}

__global__ void k_bn_bwd_params(const double* __restrict__ sums, int T, int C, float* __restrict__ dgamma,
                                float* __restrict__ dbias, int accumulate) {
    int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    double dg = 0.0, db = 0.0;
    for (int t = 0; t < T; ++t) {
        db += sums[((int64_t)t * C + c) * 2 + 0];
        dg += sums[((int64_t)t * C + c) * 2 + 1];
    }
    if (dgamma) dgamma[c] = accumulate ? dgamma[c] + (float)dg : (float)dg;
    if (dbias) dbias[c] = accumulate ? dbias[c] + (float)db : (float)db;
}

// One launch for the BatchNorm-backward second phase of a layer (was: reduce + coefficients + parameter gradients).
// One block per channel; 32 lanes share the block partials of one (t, c) (lane k sums blocks k, k+32, ... in order,
// then a fixed xor tree), 32 timesteps per pass (1024 threads); thread 0 adds the per-timestep parameter sums in t order.
__global__ __launch_bounds__(1024) void k_bn_bwd_finalize_fused(
    const double* __restrict__ sums_, int gx_blocks, int T, int64_t M, int C, const float* __restrict__ gamma,
    const float* __restrict__ mean, const float* __restrict__ invstd, float* __restrict__ coefA,
    float* __restrict__ coefB, float* __restrict__ coefC, float* __restrict__ dgamma, float* __restrict__ dbias,
    int accumulate, int from_state, const float* __restrict__ bn_bias, const float* __restrict__ gx,
    const float* __restrict__ y, int64_t ldy) {
    // from_state (the scan ran with SNN_SCAN_SUMS_FROM_STATE): the second partial is sum(gx * x), x = gamma*xhat + bias the
    // neuron's input, so sum(gx * xhat) = (sum(gx * x) - bias * sum(gx)) / gamma.  A channel whose gamma is exactly 0 carries
    // no xhat in x: for it (and only for it) the sum is formed here from gx and y - slow, one block per such channel.
    const float* __restrict__ sums = reinterpret_cast<const float*>(sums_);  // fp32 block partials
    __shared__ double sm_b[32], sm_g[32];
    const int c = blockIdx.x;
    const int sub = threadIdx.x & 31, tl = threadIdx.x >> 5;
    double dg = 0.0, db = 0.0;
    const double gam = (double)(gamma ? gamma[c] : 1.0f);
    const double bnb = (double)((from_state && bn_bias) ? bn_bias[c] : 0.0f);
    const bool direct = from_state && gam == 0.0;
    for (int tb = 0; tb < T; tb += 32) {
        const int t = tb + tl;
        const int idx = t * C + c;
        double s1 = 0.0, sy = 0.0;
        if (t < T) {
#pragma unroll 4
            for (int bk = sub; bk < gx_blocks; bk += 32) {
                const float2 v = *reinterpret_cast<const float2*>(sums + ((int64_t)bk * T * C + idx) * 2);
                s1 += (double)v.x;
                sy += (double)v.y;
            }
        }
        if (direct) {   // uniform over the block
            sy = 0.0;
            if (t < T) {
                for (int64_t m = sub; m < M; m += 32)
                    sy += (double)gx[((int64_t)t * M + m) * C + c] * (double)y[((int64_t)t * M + m) * ldy + c];
            }
        }
        for (int stride = 16; stride >= 1; stride >>= 1) {
            s1 += __shfl_xor(s1, stride, 64);
            sy += __shfl_xor(sy, stride, 64);
        }
        if (sub == 0 && t < T) {
            const double mu = (double)mean[idx], is = (double)invstd[idx];
            if (from_state && !direct) sy = sum_gx_y_from_state(s1, sy, mu, is, gam, bnb);   // (as k_bn_bwd_reduce: same bits)
            const double s2 = is * (sy - mu * s1);  // sum gx * xhat
            const double n = (double)M;
            const double a = gam * is;
            const double m1 = s1 / n, m2 = s2 / n;
            coefA[idx] = (float)a;
            coefB[idx] = (float)(-a * is * m2);
            coefC[idx] = (float)(-a * m1 + a * is * mu * m2);
            sm_b[tl] = s1;
            sm_g[tl] = s2;
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            const int nt = T - tb < 32 ? T - tb : 32;
            for (int k = 0; k < nt; ++k) {
                db += sm_b[k];
                dg += sm_g[k];
            }
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        if (dgamma) dgamma[c] = accumulate ? dgamma[c] + (float)dg : (float)dg;
        if (dbias) dbias[c] = accumulate ? dbias[c] + (float)db : (float)db;
    }
}

template <int VEC, bool SB = false>
__global__ __launch_bounds__(kThreads) void k_bn_bwd_apply(const float* __restrict__ gx, const float* __restrict__ y,
                                                           int64_t ldy, const float* __restrict__ coefA,
                                                           const float* __restrict__ coefB,
                                                           const float* __restrict__ coefC, float* __restrict__ dy,
                                                           int64_t lddy, int T, int64_t M, int C, int accumulate) {
    typedef typename Vec<VEC>::type V;
    const int cv = C / VEC;
    const int64_t total = (int64_t)T * M * cv;
    for (int64_t e = (int64_t)blockIdx.x * kThreads + threadIdx.x; e < total; e += (int64_t)gridDim.x * kThreads) {
        const int64_t row = e / cv;
        const int c = (int)(e % cv) * VEC;
        const int64_t t = row / M;
        V g = VecS<VEC, SB>::load_last(gx, row * C + c);      // last reads of both: dy is what the next kernels want cached
        V yv = VecS<VEC, SB>::load_last(y, row * ldy + c);
        V a = Vec<VEC>::load(coefA + t * C + c);
        V b = Vec<VEC>::load(coefB + t * C + c);
        V k = Vec<VEC>::load(coefC + t * C + c);
        V r;
#pragma unroll
        for (int j = 0; j < VEC; ++j)
            lane<VEC>(r, j) = lane<VEC>(a, j) * lane<VEC>(g, j) + lane<VEC>(b, j) * lane<VEC>(yv, j) + lane<VEC>(k, j);
        if (accumulate) {
            V old = VecS<VEC, SB>::load(dy, row * lddy + c);
#pragma unroll
            for (int j = 0; j < VEC; ++j) lane<VEC>(r, j) += lane<VEC>(old, j);
        }
        VecS<VEC, SB>::store(dy, row * lddy + c, r);
    }
}

static bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }
static bool aligned8(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 7u) == 0; }

}  // namespace

// -------------------------------------------------------------------------------------------- C ABI
extern "C" size_t snn_bn_stats_partial_size(int T, int64_t M, int C) {
    if (T <= 0 || M <= 0 || C <= 0) return 0;
    StatsPlan pl = stats_plan(T, M, C);
    // partial sums followed by T*C unbiased variances (scratch of the finalize step)
    return (size_t)T * pl.chunks * C * 2 + (size_t)T * C;
}

extern "C" int snn_bn_stats(const float* y, int64_t ldy, int T, int64_t M, int C, double* partial, void* stream) {
    SNN_REQUIRE(y && partial, "snn_bn_stats: null pointer");
    SNN_REQUIRE(T > 0 && M > 0 && C > 0 && ldy >= C, "snn_bn_stats: bad shape T=%d M=%lld C=%d ldy=%lld", T,
                (long long)M, C, (long long)ldy);
    StatsPlan pl = stats_plan(T, M, C);
    if (pl.vec == 4) SNN_REQUIRE(aligned16(y) && ldy % 4 == 0, "snn_bn_stats: y must be 16-byte aligned with ldy%%4==0");
    dim3 grid(pl.chunks, T, pl.zblocks);
    if (pl.vec == 4)
        hipLaunchKernelGGL(k_bn_stats<4>, grid, dim3(kThreads), 0, (hipStream_t)stream, y, ldy, M, C, pl.cvb, partial);
    else
        hipLaunchKernelGGL(k_bn_stats<1>, grid, dim3(kThreads), 0, (hipStream_t)stream, y, ldy, M, C, pl.cvb, partial);
    SNN_CHECK_LAUNCH("snn_bn_stats");
    return 0;
}

extern "C" int snn_bn_stats_bf16(const float* y, int64_t ldy, int T, int64_t M, int C, double* partial, void* stream) {
    SNN_REQUIRE(y && partial, "snn_bn_stats_bf16: null pointer");
    SNN_REQUIRE(T > 0 && M > 0 && C > 0 && ldy >= C && C % 4 == 0 && ldy % 4 == 0 && aligned8(y),
                "snn_bn_stats_bf16: bad shape (T=%d M=%lld C=%d ldy=%lld; C and ldy multiples of 4, y 8-byte aligned)", T,
                (long long)M, C, (long long)ldy);
    StatsPlan pl = stats_plan(T, M, C);
    dim3 grid(pl.chunks, T, pl.zblocks);
    hipLaunchKernelGGL((k_bn_stats<4, true>), grid, dim3(kThreads), 0, (hipStream_t)stream, y, ldy, M, C, pl.cvb, partial);
    SNN_CHECK_LAUNCH("snn_bn_stats_bf16");
    return 0;
}

extern "C" int snn_bn_stats_finalize(const double* partial, int chunks, int rows_per_chunk, int T, int64_t M, int C,
                                     const float* gamma,
                                     const float* bias, float eps, float momentum, float* running_mean,
                                     float* running_var, int use_running, float* mean, float* invstd, float* alpha,
                                     float* beta, void* stream) {
    SNN_REQUIRE(mean && invstd && alpha && beta, "snn_bn_stats_finalize: null output");
    SNN_REQUIRE(T > 0 && M > 0 && C > 0, "snn_bn_stats_finalize: bad shape");
    SNN_REQUIRE(use_running ? (running_mean && running_var) : (partial != nullptr),
                "snn_bn_stats_finalize: missing statistics source");
    SNN_REQUIRE(chunks >= 0 && rows_per_chunk >= 0 && (chunks > 0 || rows_per_chunk == 0),
                "snn_bn_stats_finalize: bad partial layout (chunks %d, rows per chunk %d)", chunks, rows_per_chunk);
    if (chunks == 0) chunks = stats_plan(T, M, C).chunks;   // the layout snn_bn_stats writes
    if (chunks > 64 && !use_running)
        hipLaunchKernelGGL(k_bn_stats_finalize_fused<32>, dim3(C), dim3(1024), 0, (hipStream_t)stream, partial, chunks,
                           rows_per_chunk, T, M, C, gamma, bias, eps, momentum, running_mean, running_var, use_running,
                           mean, invstd, alpha, beta);
    else
        hipLaunchKernelGGL(k_bn_stats_finalize_fused<8>, dim3(C), dim3(256), 0, (hipStream_t)stream, partial, chunks,
                           rows_per_chunk, T, M, C, gamma, bias, eps, momentum, running_mean, running_var, use_running,
                           mean, invstd, alpha, beta);
    SNN_CHECK_LAUNCH("snn_bn_stats_finalize");
    return 0;
}

extern "C" int snn_bn_stats_reduce(const double* partial, int chunks, int rows_per_chunk, int T, int64_t M, int C,
                                   double* sums, void* stream) {
    SNN_REQUIRE(partial && sums && T > 0 && M > 0 && C > 0, "snn_bn_stats_reduce: bad arguments");
    SNN_REQUIRE(chunks >= 0 && rows_per_chunk >= 0 && (chunks > 0 || rows_per_chunk == 0),
                "snn_bn_stats_reduce: bad partial layout (chunks %d, rows per chunk %d)", chunks, rows_per_chunk);
    if (chunks == 0) chunks = stats_plan(T, M, C).chunks;
    int n = T * C;
    hipLaunchKernelGGL(k_bn_stats_reduce, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, partial, chunks,
                       rows_per_chunk, T, M, C, sums);
    SNN_CHECK_LAUNCH("snn_bn_stats_reduce");
    return 0;
}

extern "C" int snn_bn_stats_from_sums(const double* sums, int T, int64_t M_total, int C, const float* gamma,
                                      const float* bias, float eps, float momentum, float* running_mean,
                                      float* running_var, float* mean, float* invstd, float* alpha, float* beta,
                                      double* var_scratch, void* stream) {
    SNN_REQUIRE(sums && mean && invstd && alpha && beta && var_scratch, "snn_bn_stats_from_sums: null pointer");
    SNN_REQUIRE(T > 0 && M_total > 0 && C > 0, "snn_bn_stats_from_sums: bad shape");
    int n = T * C;
    hipLaunchKernelGGL(k_bn_stats_finalize, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, sums, 1, T,
                       M_total, C, gamma, bias, eps, running_mean, running_var, 0, mean, invstd, alpha, beta,
                       var_scratch);
    SNN_CHECK_LAUNCH("snn_bn_stats_from_sums");
    if (running_mean && running_var) {
        hipLaunchKernelGGL(k_bn_running_update, dim3((C + 255) / 256), dim3(256), 0, (hipStream_t)stream, mean,
                           var_scratch, T, C, momentum, running_mean, running_var);
        SNN_CHECK_LAUNCH("snn_bn_running_update");
    }
    return 0;
}

#define SNN_DISPATCH_FWD(NEURON, SAVE)                                                                            \
    do {                                                                                                          \
        if (vec == 4)                                                                                             \
            hipLaunchKernelGGL((k_affine_neuron_fwd<NEURON, 4, SAVE>), grid, dim3(kThreads), 0, (hipStream_t)stream, \
                               y, ldy, alpha, beta, v0, i0, out, ldo, addend, ld_addend, vT, iT, vdec, T, M, C, *p, last_only); \
        else                                                                                                      \
            hipLaunchKernelGGL((k_affine_neuron_fwd<NEURON, 1, SAVE>), grid, dim3(kThreads), 0, (hipStream_t)stream, \
                               y, ldy, alpha, beta, v0, i0, out, ldo, addend, ld_addend, vT, iT, vdec, T, M, C, *p, last_only); \
    } while (0)

static int neuron_fwd(int neuron, const float* y, int64_t ldy, const float* alpha, const float* beta, const float* v0,
                      const float* i0, float* out, int64_t ldo, const float* addend, int64_t ld_addend, float* vT,
                      float* iT, float* vdec, int ckpt_mode, int T, int64_t M, int C, const snn_neuron_params* p,
                      int flags, void* stream) {
    SNN_REQUIRE((flags & ~(SNN_SCAN_LAST_STEP_ONLY | SNN_SCAN_BF16_STORAGE | SNN_SCAN_SPIKES_FROM_VDEC)) == 0,
                "snn_affine_neuron_fwd: unknown flags 0x%x", flags);
    const int last_only = (flags & SNN_SCAN_LAST_STEP_ONLY) != 0;
    const bool sb = (flags & SNN_SCAN_BF16_STORAGE) != 0;   // y, out, addend, vdec are bf16 tensors
    const bool no_out = (flags & SNN_SCAN_SPIKES_FROM_VDEC) != 0;   // no output tensor: the consumer thresholds vdec
    SNN_REQUIRE(y && p && (out || no_out), "snn_affine_neuron_fwd: null pointer");
    if (no_out) {
        SNN_REQUIRE(neuron == SNN_NEURON_LIF && vdec && !ckpt_mode && !addend && !last_only && !sb && alpha && !out &&
                        C % 4 == 0 && ldy % 4 == 0 && aligned16(y) && aligned16(vdec),
                    "snn_affine_neuron_fwd: SNN_SCAN_SPIKES_FROM_VDEC is for Norm -> LIF with saved potentials, no shortcut, "
                    "all T steps, fp32 tensors, 4-channel groups; out must be NULL");
        ldo = C;   // (unused; keeps the checks below meaningful)
    }
    SNN_REQUIRE(!last_only || ((neuron == SNN_NEURON_LIF || neuron == SNN_NEURON_LI || neuron == SNN_NEURON_LI_TANH) &&
                               !addend),
                "snn_affine_neuron_fwd: SNN_SCAN_LAST_STEP_ONLY is for LIF / LI / LI+Tanh without a shortcut");
    SNN_REQUIRE(!addend || (ld_addend >= C && neuron != SNN_NEURON_LI_TANH),
                "snn_affine_neuron_fwd: addend needs ld_addend >= C and is not allowed with LI_TANH");
    SNN_REQUIRE(T > 0 && M > 0 && C > 0 && ldy >= C && ldo >= C, "snn_affine_neuron_fwd: bad shape");
    SNN_REQUIRE((alpha == nullptr) == (beta == nullptr), "snn_affine_neuron_fwd: alpha/beta must come together");
    SNN_REQUIRE(neuron >= SNN_NEURON_NONE && neuron <= SNN_NEURON_SYNAPSE, "snn_affine_neuron_fwd: bad neuron %d",
                neuron);
    int vec = (C % 4 == 0 && ldy % 4 == 0 && ldo % 4 == 0 && aligned16(y) && aligned16(out) && aligned16(alpha) &&
               aligned16(beta) && aligned16(v0) && aligned16(i0) && aligned16(vT) && aligned16(iT) && aligned16(vdec) &&
               (!addend || (ld_addend % 4 == 0 && aligned16(addend))))
                  ? 4
                  : 1;
    if (sb) {   // bf16 tensors: 4 elements = 8 bytes per access
        const bool ok8 = C % 4 == 0 && ldy % 4 == 0 && ldo % 4 == 0 && aligned8(y) && aligned8(out) && aligned8(vdec) &&
                         aligned16(alpha) && aligned16(beta) && aligned16(v0) && aligned16(i0) && aligned16(vT) &&
                         aligned16(iT) && (!addend || (ld_addend % 4 == 0 && aligned8(addend)));
        SNN_REQUIRE(ok8 && !ckpt_mode && (neuron == SNN_NEURON_NONE || neuron == SNN_NEURON_LIF || neuron == SNN_NEURON_LI ||
                                          neuron == SNN_NEURON_LI_TANH),
                    "snn_affine_neuron_fwd: bf16 storage covers NONE / LIF / LI / LI+Tanh on channel counts and strides that "
                    "are multiples of 4 (8-byte aligned tensors), without checkpointing");
        vec = 4;
        // 8 channels (16 bytes of bf16) per access when the layout allows: the scan is bound by the number of memory
        // instructions, not by their bytes (8-byte accesses: 3.3 TB/s of bf16 against 5.1 TB/s with fp32 tensors)
        static const bool no_v8 = snn_tuning_env("SNN_SCAN_NO_VEC8") != nullptr;   // tuning / bisecting aid
        if (!no_v8 && C % 8 == 0 && ldy % 8 == 0 && ldo % 8 == 0 && aligned16(y) && aligned16(out) && aligned16(vdec) &&
            (!addend || (ld_addend % 8 == 0 && aligned16(addend))))
            vec = 8;
    }
    int64_t total = M * (C / vec);
    // every thread scans the same number of (pixel, channel group) items over all T (grid-stride, tail masked) and
    // all blocks are resident at once: a capped grid with 1.4 items per thread would run 2 rounds for 1.4 of work
    const int64_t per_thread = snn_ceil_div(total, (int64_t)snn_max_blocks() * kThreads);
    int64_t blocks = snn_ceil_div(total, kThreads * per_thread);
    dim3 grid((unsigned)blocks);
#define SNN_DISPATCH_FWD_S(NEURON, SAVE)                                                                          \
    do {                                                                                                          \
        if (vec == 8)                                                                                             \
            hipLaunchKernelGGL((k_affine_neuron_fwd<NEURON, 8, SAVE, true>), grid, dim3(kThreads), 0, (hipStream_t)stream, \
                               y, ldy, alpha, beta, v0, i0, out, ldo, addend, ld_addend, vT, iT, vdec, T, M, C, *p, last_only); \
        else                                                                                                      \
            hipLaunchKernelGGL((k_affine_neuron_fwd<NEURON, 4, SAVE, true>), grid, dim3(kThreads), 0, (hipStream_t)stream, \
                               y, ldy, alpha, beta, v0, i0, out, ldo, addend, ld_addend, vT, iT, vdec, T, M, C, *p, last_only); \
    } while (0)
    if (sb) {
        switch (neuron) {
            case SNN_NEURON_NONE: SNN_DISPATCH_FWD_S(SNN_NEURON_NONE, 0); break;
            case SNN_NEURON_LIF:
                if (vdec) SNN_DISPATCH_FWD_S(SNN_NEURON_LIF, 1);
                else SNN_DISPATCH_FWD_S(SNN_NEURON_LIF, 0);
                break;
            case SNN_NEURON_LI: SNN_DISPATCH_FWD_S(SNN_NEURON_LI, 0); break;
            default: SNN_DISPATCH_FWD_S(SNN_NEURON_LI_TANH, 0); break;
        }
        SNN_CHECK_LAUNCH("snn_affine_neuron_fwd");
        return 0;
    }
#undef SNN_DISPATCH_FWD_S
    switch (neuron) {
        case SNN_NEURON_NONE: SNN_DISPATCH_FWD(SNN_NEURON_NONE, 0); break;
        case SNN_NEURON_LIF:
            if (vdec && ckpt_mode) SNN_DISPATCH_FWD(SNN_NEURON_LIF, 2);
            else if (vdec) SNN_DISPATCH_FWD(SNN_NEURON_LIF, 1);
            else SNN_DISPATCH_FWD(SNN_NEURON_LIF, 0);
            break;
        case SNN_NEURON_LI: SNN_DISPATCH_FWD(SNN_NEURON_LI, 0); break;
        case SNN_NEURON_LI_TANH: SNN_DISPATCH_FWD(SNN_NEURON_LI_TANH, 0); break;
        case SNN_NEURON_SLI:
            if (vdec) SNN_DISPATCH_FWD(SNN_NEURON_SLI, 1);
            else SNN_DISPATCH_FWD(SNN_NEURON_SLI, 0);
            break;
        default:
            if (vdec) SNN_DISPATCH_FWD(SNN_NEURON_SYNAPSE, 1);
            else SNN_DISPATCH_FWD(SNN_NEURON_SYNAPSE, 0);
            break;
    }
    SNN_CHECK_LAUNCH("snn_affine_neuron_fwd");
    return 0;
}

extern "C" int snn_affine_neuron_fwd(int neuron, const float* y, int64_t ldy, const float* alpha, const float* beta,
                                     const float* v0, const float* i0, float* out, int64_t ldo, const float* addend,
                                     int64_t ld_addend, float* vT, float* iT, float* vdec, int T, int64_t M, int C,
                                     const snn_neuron_params* p, int flags, void* stream) {
    return neuron_fwd(neuron, y, ldy, alpha, beta, v0, i0, out, ldo, addend, ld_addend, vT, iT, vdec, 0, T, M, C, p,
                      flags, stream);
}

extern "C" int snn_lif_ckpt_interval(void) { return kCkpt; }

extern "C" int snn_lif_fwd_ckpt(const float* y, int64_t ldy, const float* alpha, const float* beta, const float* v0,
                                const float* i0, float* out, int64_t ldo, const float* addend, int64_t ld_addend,
                                float* vT, float* iT, float* ckpt, int T, int64_t M, int C, const snn_neuron_params* p,
                                void* stream) {
    SNN_REQUIRE(ckpt, "snn_lif_fwd_ckpt: null checkpoint buffer");
    return neuron_fwd(SNN_NEURON_LIF, y, ldy, alpha, beta, v0, i0, out, ldo, addend, ld_addend, vT, iT, ckpt, 1, T, M,
                      C, p, 0, stream);
}

extern "C" size_t snn_affine_neuron_bwd_sums_size(int T, int64_t M, int C) {
    if (T <= 0 || M <= 0 || C <= 0) return 0;
    BwdPlan pl = bwd_plan(T, M, C, true);
    return (size_t)pl.gx * T * C * 2;
}

#define SNN_LAUNCH_BWD_(NEURON, VEC_, MODE_, BUF_, NP_)                                                          \
    hipLaunchKernelGGL((k_affine_neuron_bwd<NEURON, VEC_, MODE_, BUF_, NP_>), grid, dim3(kThreads), pl.lds_bytes, \
                       (hipStream_t)stream, g_out, ldg, state, y, ldy, g_vT, g_iT, alpha, beta, apply_scale, gx, \
                       g_v0, g_i0, sums, T, M, C, pl.cvb, *p, last_only)
#define SNN_LAUNCH_BWD(NEURON, VEC_, MODE_)                                             \
    do {                                                                                \
        if (VEC_ == 4 && buf_ok && pl.rpb == 1) SNN_LAUNCH_BWD_(NEURON, 4, MODE_, true, 1); \
        else if (VEC_ == 4 && buf_ok) SNN_LAUNCH_BWD_(NEURON, 4, MODE_, true, kBwdNP);  \
        else SNN_LAUNCH_BWD_(NEURON, VEC_, MODE_, false, kBwdNP);                       \
    } while (0)
#define SNN_DISPATCH_BWD(NEURON)                                  \
    do {                                                          \
        if (pl.vec == 4) {                                        \
            if (pl.mode == 0) SNN_LAUNCH_BWD(NEURON, 4, 0);       \
            else if (pl.mode == 1) SNN_LAUNCH_BWD(NEURON, 4, 1);  \
            else SNN_LAUNCH_BWD(NEURON, 4, 2);                    \
        } else {                                                  \
            if (pl.mode == 0) SNN_LAUNCH_BWD(NEURON, 1, 0);       \
            else if (pl.mode == 1) SNN_LAUNCH_BWD(NEURON, 1, 1);  \
            else SNN_LAUNCH_BWD(NEURON, 1, 2);                    \
        }                                                         \
    } while (0)

static bool sums_from_state_ok(int neuron, int T, int64_t M, int C, int64_t ldg, const snn_neuron_params* p, int flags) {
    if (neuron != SNN_NEURON_LIF || !p || T <= 0 || M <= 0 || C <= 0 || ldg < C) return false;
    if (flags & (SNN_SCAN_WIDE_ADDRESSING | SNN_SCAN_LAST_STEP_ONLY | SNN_SCAN_BF16_STORAGE)) return false;
    // the rebuilt input divides by c_mem: keep the amplification of the potentials' rounding error bounded
    if (!(p->c_mem >= 1.0f / 64.0f && p->c_mem <= 1.0f)) return false;
    const BwdPlan pl = bwd_plan(T, M, C, true);
    const int64_t ld_max = ldg > C ? ldg : C;
    return pl.vec == 4 && pl.mode == 1 && ldg % 4 == 0 && M * ld_max * 4 < 0x7fffffffLL;
}

extern "C" int snn_affine_neuron_bwd_sums_from_state(int neuron, int T, int64_t M, int C, int64_t ldg,
                                                     const snn_neuron_params* p, int flags) {
    return sums_from_state_ok(neuron, T, M, C, ldg, p, flags) ? 1 : 0;
}

extern "C" int snn_affine_neuron_bwd(int neuron, const float* g_out, int64_t ldg, const float* state, const float* y,
                                     int64_t ldy, const float* g_vT, const float* g_iT, const float* alpha,
                                     const float* beta, int apply_scale, float* gx, float* g_v0, float* g_i0,
                                     double* sums, int T, int64_t M, int C, const snn_neuron_params* p,
                                     int flags, void* stream) {
    SNN_REQUIRE(g_out && gx && p, "snn_affine_neuron_bwd: null pointer");
    SNN_REQUIRE((flags & ~(SNN_SCAN_WIDE_ADDRESSING | SNN_SCAN_LAST_STEP_ONLY | SNN_SCAN_BF16_STORAGE |
                           SNN_SCAN_SUMS_FROM_STATE | SNN_SCAN_STATE_LOOKBACK)) == 0,
                "snn_affine_neuron_bwd: unknown flags 0x%x", flags);
    const bool yfree = (flags & SNN_SCAN_SUMS_FROM_STATE) != 0;
    const int lookback = (flags & SNN_SCAN_STATE_LOOKBACK) != 0;
    SNN_REQUIRE(yfree || !lookback, "snn_affine_neuron_bwd: SNN_SCAN_STATE_LOOKBACK belongs to SNN_SCAN_SUMS_FROM_STATE");
    flags &= ~(SNN_SCAN_SUMS_FROM_STATE | SNN_SCAN_STATE_LOOKBACK);
    if (yfree) {
        SNN_REQUIRE(sums && state && !apply_scale && sums_from_state_ok(neuron, T, M, C, ldg, p, flags),
                    "snn_affine_neuron_bwd: SNN_SCAN_SUMS_FROM_STATE not covered (ask snn_affine_neuron_bwd_sums_from_state; "
                    "LIF from the initial state, sums wanted, train-mode BatchNorm)");
        SNN_REQUIRE(aligned16(g_out) && aligned16(state) && aligned16(g_vT) && aligned16(g_iT) && aligned16(gx) &&
                    aligned16(g_v0) && aligned16(g_i0), "snn_affine_neuron_bwd: buffers must be 16-byte aligned");
        const BwdPlan pl = bwd_plan(T, M, C, true);
        dim3 grid(pl.gx, pl.gy);
        // three pixels per thread: the four values a pixel keeps for the statistic of two steps later do not fit the
        // 256 registers of two waves per SIMD beside four pixels' operand sets (287, or 21 spilled)
#define SNN_LAUNCH_YF(NP_)                                                                                              \
    hipLaunchKernelGGL((k_affine_neuron_bwd<SNN_NEURON_LIF, 4, 1, true, NP_, false, true>), grid, dim3(kThreads),         \
                       pl.lds_bytes, (hipStream_t)stream, g_out, ldg, state, g_out, ldg, g_vT, g_iT, alpha, beta, 0, gx, \
                       g_v0, g_i0, sums, T, M, C, pl.cvb, *p, lookback)
        if (pl.rpb == 1) SNN_LAUNCH_YF(1);
        else if (pl.rpb == 2) SNN_LAUNCH_YF(2);
        else SNN_LAUNCH_YF(3);
#undef SNN_LAUNCH_YF
        SNN_CHECK_LAUNCH("snn_affine_neuron_bwd");
        return 0;
    }
    const int last_only = (flags & SNN_SCAN_LAST_STEP_ONLY) != 0;
    const bool sb = (flags & SNN_SCAN_BF16_STORAGE) != 0;   // g_out, state, y, gx are bf16 tensors
    SNN_REQUIRE(!last_only || neuron == SNN_NEURON_LIF || neuron == SNN_NEURON_LI || neuron == SNN_NEURON_LI_TANH,
                "snn_affine_neuron_bwd: SNN_SCAN_LAST_STEP_ONLY is for LIF / LI / LI+Tanh");
    SNN_REQUIRE(T > 0 && M > 0 && C > 0 && ldg >= C, "snn_affine_neuron_bwd: bad shape");
    SNN_REQUIRE(neuron >= SNN_NEURON_NONE && neuron <= SNN_NEURON_SYNAPSE, "snn_affine_neuron_bwd: bad neuron %d",
                neuron);
    const bool needs_x = neuron == SNN_NEURON_SLI || neuron == SNN_NEURON_SYNAPSE;
    SNN_REQUIRE(!(neuron == SNN_NEURON_LIF || neuron == SNN_NEURON_LI_TANH || needs_x) || state,
                "snn_affine_neuron_bwd: saved state required");
    SNN_REQUIRE(!(sums || needs_x) || (y && ldy >= C), "snn_affine_neuron_bwd: y required");
    SNN_REQUIRE((alpha == nullptr) == (beta == nullptr), "snn_affine_neuron_bwd: alpha/beta must come together");
    SNN_REQUIRE(!apply_scale || alpha, "snn_affine_neuron_bwd: apply_scale needs alpha");
    BwdPlan pl = bwd_plan(T, M, C, sums != nullptr);
    if (sb) {
        const bool ok = pl.vec == 4 && ldg % 4 == 0 && aligned8(g_out) && aligned8(state) && aligned8(gx) && aligned16(g_vT) &&
                        aligned16(g_iT) && aligned16(alpha) && aligned16(beta) && aligned16(g_v0) && aligned16(g_i0) &&
                        (!sums || (ldy % 4 == 0 && aligned8(y))) &&
                        (neuron == SNN_NEURON_NONE || neuron == SNN_NEURON_LIF || neuron == SNN_NEURON_LI ||
                         neuron == SNN_NEURON_LI_TANH);
        SNN_REQUIRE(ok, "snn_affine_neuron_bwd: bf16 storage covers NONE / LIF / LI / LI+Tanh on channel counts and strides "
                        "that are multiples of 4 (8-byte aligned tensors)");
    } else if (pl.vec == 4) {
        bool ok = ldg % 4 == 0 && aligned16(g_out) && aligned16(state) && aligned16(g_vT) && aligned16(g_iT) &&
                  aligned16(alpha) && aligned16(beta) && aligned16(gx) && aligned16(g_v0) && aligned16(g_i0) &&
                  (!(sums || needs_x) || (ldy % 4 == 0 && aligned16(y)));
        SNN_REQUIRE(ok, "snn_affine_neuron_bwd: buffers must be 16-byte aligned when C%%4==0");
    }
    dim3 grid(pl.gx, pl.gy);
    // buffer addressing (see k_affine_neuron_bwd): one timestep of every tensor must fit a 31-bit byte offset
    const bool no_buf = (flags & SNN_SCAN_WIDE_ADDRESSING) != 0;
    const int64_t ld_max = ldg > ldy ? (ldg > C ? ldg : C) : (ldy > C ? ldy : C);
    // (blocks with a single pixel row take the one-pixel-per-thread instance: the three empty pixel slots of the
    // four-pixel one are computed and issued in straight-line code - measured 58 us against 45 for the branchy kernel)
    const bool buf_ok = !no_buf && M * ld_max * 4 < 0x7fffffffLL;
    if (sb) {
#define SNN_LAUNCH_BWD_S(NEURON, MODE_)                                                                              \
    do {                                                                                                             \
        if (buf_ok && pl.rpb == 1)                                                                                   \
            hipLaunchKernelGGL((k_affine_neuron_bwd<NEURON, 4, MODE_, true, 1, true>), grid, dim3(kThreads), pl.lds_bytes, \
                               (hipStream_t)stream, g_out, ldg, state, y, ldy, g_vT, g_iT, alpha, beta, apply_scale, gx, \
                               g_v0, g_i0, sums, T, M, C, pl.cvb, *p, last_only);                                    \
        else if (buf_ok)                                                                                             \
            hipLaunchKernelGGL((k_affine_neuron_bwd<NEURON, 4, MODE_, true, kBwdNP, true>), grid, dim3(kThreads),      \
                               pl.lds_bytes, (hipStream_t)stream, g_out, ldg, state, y, ldy, g_vT, g_iT, alpha, beta,  \
                               apply_scale, gx, g_v0, g_i0, sums, T, M, C, pl.cvb, *p, last_only);                   \
        else                                                                                                         \
            hipLaunchKernelGGL((k_affine_neuron_bwd<NEURON, 4, MODE_, false, kBwdNP, true>), grid, dim3(kThreads),     \
                               pl.lds_bytes, (hipStream_t)stream, g_out, ldg, state, y, ldy, g_vT, g_iT, alpha, beta,  \
                               apply_scale, gx, g_v0, g_i0, sums, T, M, C, pl.cvb, *p, last_only);                   \
    } while (0)
#define SNN_DISPATCH_BWD_S(NEURON)                        \
    do {                                                  \
        if (pl.mode == 0) SNN_LAUNCH_BWD_S(NEURON, 0);    \
        else if (pl.mode == 1) SNN_LAUNCH_BWD_S(NEURON, 1); \
        else SNN_LAUNCH_BWD_S(NEURON, 2);                 \
    } while (0)
        switch (neuron) {
            case SNN_NEURON_NONE: SNN_DISPATCH_BWD_S(SNN_NEURON_NONE); break;
            case SNN_NEURON_LIF: SNN_DISPATCH_BWD_S(SNN_NEURON_LIF); break;
            case SNN_NEURON_LI: SNN_DISPATCH_BWD_S(SNN_NEURON_LI); break;
            default: SNN_DISPATCH_BWD_S(SNN_NEURON_LI_TANH); break;
        }
#undef SNN_DISPATCH_BWD_S
#undef SNN_LAUNCH_BWD_S
        SNN_CHECK_LAUNCH("snn_affine_neuron_bwd");
        return 0;
    }
    switch (neuron) {
        case SNN_NEURON_NONE: SNN_DISPATCH_BWD(SNN_NEURON_NONE); break;
        case SNN_NEURON_LIF: SNN_DISPATCH_BWD(SNN_NEURON_LIF); break;
        case SNN_NEURON_LI: SNN_DISPATCH_BWD(SNN_NEURON_LI); break;
        case SNN_NEURON_LI_TANH: SNN_DISPATCH_BWD(SNN_NEURON_LI_TANH); break;
        case SNN_NEURON_SLI: SNN_DISPATCH_BWD(SNN_NEURON_SLI); break;
        default: SNN_DISPATCH_BWD(SNN_NEURON_SYNAPSE); break;
    }
    SNN_CHECK_LAUNCH("snn_affine_neuron_bwd");
    return 0;
}

#define SNN_LAUNCH_CKPT(VEC_, MODE_)                                                                              \
    hipLaunchKernelGGL((k_lif_bwd_ckpt<VEC_, MODE_>), grid, dim3(kThreads), pl.lds_bytes, (hipStream_t)stream, g_out, \
                       ldg, ckpt, y, ldy, g_vT, g_iT, alpha, beta, apply_scale, gx, g_v0, g_i0, sums, T, M, C, pl.cvb, \
                       *p)

// LIF backward from the checkpoints of snn_lif_fwd_ckpt; sums / outputs exactly as snn_affine_neuron_bwd(LIF)
extern "C" int snn_lif_bwd_ckpt(const float* g_out, int64_t ldg, const float* ckpt, const float* y, int64_t ldy,
                                const float* g_vT, const float* g_iT, const float* alpha, const float* beta,
                                int apply_scale, float* gx, float* g_v0, float* g_i0, double* sums, int T, int64_t M,
                                int C, const snn_neuron_params* p, void* stream) {
    SNN_REQUIRE(g_out && gx && p && ckpt && y, "snn_lif_bwd_ckpt: null pointer");
    SNN_REQUIRE(T > 0 && M > 0 && C > 0 && ldg >= C && ldy >= C, "snn_lif_bwd_ckpt: bad shape");
    SNN_REQUIRE((alpha == nullptr) == (beta == nullptr), "snn_lif_bwd_ckpt: alpha/beta must come together");
    SNN_REQUIRE(!apply_scale || alpha, "snn_lif_bwd_ckpt: apply_scale needs alpha");
    BwdPlan pl = bwd_plan(T, M, C, sums != nullptr);
    if (pl.vec == 4) {
        bool ok = ldg % 4 == 0 && ldy % 4 == 0 && aligned16(g_out) && aligned16(ckpt) && aligned16(y) &&
                  aligned16(g_vT) && aligned16(g_iT) && aligned16(alpha) && aligned16(beta) && aligned16(gx) &&
                  aligned16(g_v0) && aligned16(g_i0);
        SNN_REQUIRE(ok, "snn_lif_bwd_ckpt: buffers must be 16-byte aligned when C%%4==0");
    }
    dim3 grid(pl.gx, pl.gy);
    if (pl.vec == 4) {
        if (pl.mode == 0) SNN_LAUNCH_CKPT(4, 0);
        else if (pl.mode == 1) SNN_LAUNCH_CKPT(4, 1);
        else SNN_LAUNCH_CKPT(4, 2);
    } else {
        if (pl.mode == 0) SNN_LAUNCH_CKPT(1, 0);
        else if (pl.mode == 1) SNN_LAUNCH_CKPT(1, 1);
        else SNN_LAUNCH_CKPT(1, 2);
    }
    SNN_CHECK_LAUNCH("snn_lif_bwd_ckpt");
    return 0;
}

extern "C" int snn_bn_bwd_reduce(const double* sums, int T, int64_t M, int C, double* raw, void* stream) {
    SNN_REQUIRE(sums && raw && T > 0 && M > 0 && C > 0, "snn_bn_bwd_reduce: bad arguments");
    SNN_REQUIRE(sums != raw, "snn_bn_bwd_reduce: raw must not alias the partial sums");
    BwdPlan pl = bwd_plan(T, M, C, true);
    int n = T * C;
    hipLaunchKernelGGL(k_bn_bwd_reduce, dim3((n + 7) / 8), dim3(256), 0, (hipStream_t)stream, sums, pl.gx, T, C, raw, 0, M,
                       nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0);
    SNN_CHECK_LAUNCH("snn_bn_bwd_reduce");
    return 0;
}

// the same for sums of a scan that ran with SNN_SCAN_SUMS_FROM_STATE: raw receives (sum gx, sum gx*y) all the same
extern "C" int snn_bn_bwd_reduce_from_state(const double* sums, int T, int64_t M, int C, const float* gamma, const float* bias,
                                            const float* mean, const float* invstd, const float* gx, const float* y,
                                            int64_t ldy, double* raw, void* stream) {
    SNN_REQUIRE(sums && raw && mean && invstd && gx && y && T > 0 && M > 0 && C > 0 && ldy >= C,
                "snn_bn_bwd_reduce_from_state: bad arguments");
    SNN_REQUIRE(sums != raw, "snn_bn_bwd_reduce_from_state: raw must not alias the partial sums");
    BwdPlan pl = bwd_plan(T, M, C, true);
    int n = T * C;
    hipLaunchKernelGGL(k_bn_bwd_reduce, dim3((n + 7) / 8), dim3(256), 0, (hipStream_t)stream, sums, pl.gx, T, C, raw, 1, M,
                       gamma, bias, mean, invstd, gx, y, ldy);
    SNN_CHECK_LAUNCH("snn_bn_bwd_reduce_from_state");
    return 0;
}

extern "C" int snn_bn_bwd_coef(const double* raw, const double* raw_local, double* param_sums, int T, int64_t M_total,
                               int C, const float* gamma, const float* mean, const float* invstd, float* coefA,
                               float* coefB, float* coefC, float* dgamma, float* dbias, int accumulate,
                               void* stream) {
    SNN_REQUIRE(raw && raw_local && param_sums && mean && invstd && coefA && coefB && coefC,
                "snn_bn_bwd_coef: null pointer");
    SNN_REQUIRE(T > 0 && M_total > 0 && C > 0, "snn_bn_bwd_coef: bad shape");
    int n = T * C;
    hipLaunchKernelGGL(k_bn_bwd_coef, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, raw, raw_local, T,
                       M_total, C, gamma, mean, invstd, coefA, coefB, coefC, param_sums);
    SNN_CHECK_LAUNCH("snn_bn_bwd_coef");
    if (dgamma || dbias) {
        hipLaunchKernelGGL(k_bn_bwd_params, dim3((C + 255) / 256), dim3(256), 0, (hipStream_t)stream, param_sums, T, C,
                           dgamma, dbias, accumulate);
        SNN_CHECK_LAUNCH("snn_bn_bwd_params");
    }
    return 0;
}

// single-process form: reduce the block partials in place, then coefficients and parameter gradients
extern "C" int snn_bn_bwd_finalize(double* sums, int T, int64_t M, int C, const float* gamma, const float* mean,
                                   const float* invstd, float* coefA, float* coefB, float* coefC, float* dgamma,
                                   float* dbias, int accumulate, void* stream) {
    SNN_REQUIRE(sums && mean && invstd && coefA && coefB && coefC, "snn_bn_bwd_finalize: null pointer");
    SNN_REQUIRE(T > 0 && M > 0 && C > 0, "snn_bn_bwd_finalize: bad shape");
    BwdPlan pl = bwd_plan(T, M, C, true);
    hipLaunchKernelGGL(k_bn_bwd_finalize_fused, dim3(C), dim3(1024), 0, (hipStream_t)stream, sums, pl.gx, T, M, C, gamma,
                       mean, invstd, coefA, coefB, coefC, dgamma, dbias, accumulate, 0, nullptr, nullptr, nullptr, 0);
    SNN_CHECK_LAUNCH("snn_bn_bwd_finalize");
    return 0;
}

// the same for sums written by a scan that ran with SNN_SCAN_SUMS_FROM_STATE (second partial: sum(gx * x), see the kernel)
extern "C" int snn_bn_bwd_finalize_from_state(double* sums, int T, int64_t M, int C, const float* gamma, const float* bias,
                                              const float* mean, const float* invstd, const float* gx, const float* y,
                                              int64_t ldy, float* coefA, float* coefB, float* coefC, float* dgamma,
                                              float* dbias, int accumulate, void* stream) {
    SNN_REQUIRE(sums && mean && invstd && coefA && coefB && coefC && gx && y, "snn_bn_bwd_finalize_from_state: null pointer");
    SNN_REQUIRE(T > 0 && M > 0 && C > 0 && ldy >= C, "snn_bn_bwd_finalize_from_state: bad shape");
    BwdPlan pl = bwd_plan(T, M, C, true);
    hipLaunchKernelGGL(k_bn_bwd_finalize_fused, dim3(C), dim3(1024), 0, (hipStream_t)stream, sums, pl.gx, T, M, C, gamma,
                       mean, invstd, coefA, coefB, coefC, dgamma, dbias, accumulate, 1, bias, gx, y, ldy);
    SNN_CHECK_LAUNCH("snn_bn_bwd_finalize_from_state");
    return 0;
}

extern "C" int snn_bn_bwd_apply(const float* gx, const float* y, int64_t ldy, const float* coefA, const float* coefB,
                                const float* coefC, float* dy, int64_t lddy, int T, int64_t M, int C, int accumulate,
                                void* stream) {
    SNN_REQUIRE(gx && y && coefA && coefB && coefC && dy, "snn_bn_bwd_apply: null pointer");
    SNN_REQUIRE(T > 0 && M > 0 && C > 0 && ldy >= C && lddy >= C, "snn_bn_bwd_apply: bad shape");
    int vec = (C % 4 == 0 && ldy % 4 == 0 && lddy % 4 == 0 && aligned16(gx) && aligned16(y) && aligned16(dy) &&
               aligned16(coefA) && aligned16(coefB) && aligned16(coefC))
                  ? 4
                  : 1;
    int64_t total = (int64_t)T * M * (C / vec);
    int64_t blocks = snn_ceil_div(total, kThreads);
    if (blocks > snn_max_blocks()) blocks = snn_max_blocks();
    if (const char* force = snn_tuning_env("SNN_APPLY_CAP")) {   // tuning aid: blocks per launch of the apply pass
        if (atoi(force) > 0 && blocks > atoi(force)) blocks = atoi(force);
    }
    if (vec == 4)
        hipLaunchKernelGGL(k_bn_bwd_apply<4>, dim3((unsigned)blocks), dim3(kThreads), 0, (hipStream_t)stream, gx, y,
                           ldy, coefA, coefB, coefC, dy, lddy, T, M, C, accumulate);
    else
        hipLaunchKernelGGL(k_bn_bwd_apply<1>, dim3((unsigned)blocks), dim3(kThreads), 0, (hipStream_t)stream, gx, y,
                           ldy, coefA, coefB, coefC, dy, lddy, T, M, C, accumulate);
    SNN_CHECK_LAUNCH("snn_bn_bwd_apply");
    return 0;
}

extern "C" int snn_bn_bwd_apply_bf16(const float* gx, const float* y, int64_t ldy, const float* coefA, const float* coefB,
                                     const float* coefC, float* dy, int64_t lddy, int T, int64_t M, int C, int accumulate,
                                     void* stream) {
    SNN_REQUIRE(gx && y && coefA && coefB && coefC && dy, "snn_bn_bwd_apply_bf16: null pointer");
    SNN_REQUIRE(T > 0 && M > 0 && C > 0 && ldy >= C && lddy >= C && C % 4 == 0 && ldy % 4 == 0 && lddy % 4 == 0 &&
                    aligned8(gx) && aligned8(y) && aligned8(dy) && aligned16(coefA) && aligned16(coefB) && aligned16(coefC),
                "snn_bn_bwd_apply_bf16: bad shape (C and strides multiples of 4, bf16 tensors 8-byte aligned)");
    int64_t total = (int64_t)T * M * (C / 4);
    int64_t blocks = snn_ceil_div(total, kThreads);
    if (blocks > snn_max_blocks()) blocks = snn_max_blocks();
    hipLaunchKernelGGL((k_bn_bwd_apply<4, true>), dim3((unsigned)blocks), dim3(kThreads), 0, (hipStream_t)stream, gx, y, ldy,
                       coefA, coefB, coefC, dy, lddy, T, M, C, accumulate);
    SNN_CHECK_LAUNCH("snn_bn_bwd_apply_bf16");
    return 0;
}
