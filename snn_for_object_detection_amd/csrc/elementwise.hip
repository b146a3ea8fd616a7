// Layout adapters, merges (Residual sum / Dense concat slices), pointwise activations, pooling,
// nearest upsampling, fused Adamax and event voxelisation for gfx950.  All HBM-bound: 16-byte
// lane-contiguous accesses where the shape allows, grid-stride loops capped at 8 blocks / CU.
#include <stdarg.h>
#include "snn_common.h"

// ------------------------------------------------------------------------------------------ errors
static thread_local char g_err[512] = "";

void snn_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* snn_last_error(void) { return g_err; }
extern "C" int snn_abi_version(void) { return SNN_ABI_VERSION; }

namespace {

constexpr int kThreads = 256;

static unsigned grid_for(int64_t n) {
    int64_t b = snn_ceil_div(n, kThreads);
    if (b > snn_max_blocks()) b = snn_max_blocks();
    if (b < 1) b = 1;
    return (unsigned)b;
}
static bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// ------------------------------------------------------------------------------------------ layout
// [N][C][HW] -> [N][HW][C] through a 32x33 LDS tile: both sides coalesced.
__global__ void k_transpose_cp(const float* __restrict__ src, float* __restrict__ dst, int rows, int cols) {
    // per image: src is [rows][cols], dst is [cols][rows]
    __shared__ float tile[32][33];
    const int64_t img = blockIdx.z;
    const float* s = src + img * (int64_t)rows * cols;
    float* d = dst + img * (int64_t)rows * cols;
    const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
    for (int j = threadIdx.y; j < 32; j += blockDim.y) {
        int r = r0 + j, c = c0 + threadIdx.x;
        if (r < rows && c < cols) tile[j][threadIdx.x] = s[(int64_t)r * cols + c];
    }
    __syncthreads();
    for (int j = threadIdx.y; j < 32; j += blockDim.y) {
        int c = c0 + j, r = r0 + threadIdx.x;
        if (r < rows && c < cols) d[(int64_t)c * rows + r] = tile[threadIdx.x][j];
    }
}

// small-channel NCHW -> NHWC (C <= 4, e.g. the 2-polarity event frames): one thread per pixel
template <int C>
__global__ void k_nchw_to_nhwc_small(const float* __restrict__ src, float* __restrict__ dst, int64_t N, int64_t HW) {
    const int64_t total = N * HW;
    for (int64_t e = (int64_t)blockIdx.x * kThreads + threadIdx.x; e < total; e += (int64_t)gridDim.x * kThreads) {
        const int64_t n = e / HW, px = e % HW;
        float v[C];
#pragma unroll
        for (int c = 0; c < C; ++c) v[c] = src[(n * C + c) * HW + px];
#pragma unroll
        for (int c = 0; c < C; ++c) dst[e * C + c] = v[c];
    }
}

__global__ void k_weight_transpose(const float* __restrict__ w, float* __restrict__ wt, int Cout, int taps, int Cin) {
    const int64_t total = (int64_t)Cout * taps * Cin;
    for (int64_t e = (int64_t)blockIdx.x * kThreads + threadIdx.x; e < total; e += (int64_t)gridDim.x * kThreads) {
        // e indexes wt[ci][tap][co]
        int co = (int)(e % Cout);
        int64_t r = e / Cout;
        int tap = (int)(r % taps);
        int ci = (int)(r / taps);
        wt[e] = w[((int64_t)co * taps + tap) * Cin + ci];
    }
}

// All conv weights of a flat parameter buffer in one launch: table[l] = {offset, Cout, taps, Cin}; block (x, l)
// walks layer l with stride gridDim.x.  (47 per-layer launches of 6.5 us each sat in the backward critical path.)
__global__ void k_weight_transpose_batched(const float* __restrict__ flat_w, float* __restrict__ flat_wt,
                                           const int64_t* __restrict__ table) {
    // Per tap a Cout x Cin matrix is transposed in 32 x 32 tiles through LDS: rows of 32 consecutive ci are read (128
    // contiguous bytes), rows of 32 consecutive co are written.  (The first form gathered one element per lane with a
    // stride of taps * Cin floats - every lane its own cache line: 43 us between the optimiser and the first forward kernel
    // of every step.)  32-bit index arithmetic: a layer's weights are far below 2^31 elements (the caller that builds the
    // device-side table checks it).
    __shared__ float tile[32][33];
    const int64_t* d = table + (int64_t)blockIdx.y * 4;
    const int64_t off = d[0];
    const unsigned Cout = (unsigned)d[1], taps = (unsigned)d[2], Cin = (unsigned)d[3];
    const float* w = flat_w + off;
    float* wt = flat_wt + off;
    const unsigned tco = (Cout + 31) / 32, tci = (Cin + 31) / 32;
    const unsigned ntiles = taps * tco * tci;
    const unsigned tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8 threads
    for (unsigned t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const unsigned tap = t / (tco * tci), rem = t - tap * (tco * tci);
        const unsigned co0 = (rem / tci) * 32, ci0 = (rem % tci) * 32;
#pragma unroll
        for (unsigned k = 0; k < 32; k += 8) {
            const unsigned co = co0 + ty + k, ci = ci0 + tx;
            if (co < Cout && ci < Cin) tile[ty + k][tx] = w[(co * taps + tap) * Cin + ci];
        }
        __syncthreads();
#pragma unroll
        for (unsigned k = 0; k < 32; k += 8) {
            const unsigned ci = ci0 + ty + k, co = co0 + tx;
            if (ci < Cin && co < Cout) wt[(ci * taps + tap) * Cout + co] = tile[tx][ty + k];
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------ merges
template <int VEC, bool ADD>
__global__ void k_channels(const float* __restrict__ src, int64_t lds, float* __restrict__ dst, int64_t ldd, int64_t M,
                           int C) {
    const int cv = C / VEC;
    const int64_t total = M * cv;
    for (int64_t e = (int64_t)blockIdx.x * kThreads + threadIdx.x; e < total; e += (int64_t)gridDim.x * kThreads) {
        const int64_t m = e / cv;
        const int c = (int)(e % cv) * VEC;
        if (VEC == 4) {
            f32x4 v = *reinterpret_cast<const f32x4*>(src + m * lds + c);
            f32x4* d = reinterpret_cast<f32x4*>(dst + m * ldd + c);
            if (ADD) v += *d;
            *d = v;
        } else {
            float v = src[m * lds + c];
            if (ADD) v += dst[m * ldd + c];
            dst[m * ldd + c] = v;
        }
    }
}

template <int VEC>
__global__ void k_add(const float* __restrict__ a, int64_t lda, const float* __restrict__ b, int64_t ldb,
                      float* __restrict__ dst, int64_t ldd, int64_t M, int C) {
    const int cv = C / VEC;
    const int64_t total = M * cv;
    for (int64_t e = (int64_t)blockIdx.x * kThreads + threadIdx.x; e < total; e += (int64_t)gridDim.x * kThreads) {
        const int64_t m = e / cv;
        const int c = (int)(e % cv) * VEC;
        if (VEC == 4) {
            f32x4 x = *reinterpret_cast<const f32x4*>(a + m * lda + c);
            f32x4 y = *reinterpret_cast<const f32x4*>(b + m * ldb + c);
            *reinterpret_cast<f32x4*>(dst + m * ldd + c) = x + y;
        } else {
            dst[m * ldd + c] = a[m * lda + c] + b[m * ldb + c];
        }
    }
}

// the merges of the bf16-storage mode (SnnStore<true>, snn_common.h): bf16 tensors, the sum formed in fp32 and rounded once.
// MODE 0: dst = a (copy), 1: dst = a + b.  VEC 4 (8-byte accesses) or 1.
template <int VEC, int MODE>
__global__ void k_merge_bf16(const float* __restrict__ a, int64_t lda, const float* __restrict__ b, int64_t ldb,
                             float* __restrict__ dst, int64_t ldd, int64_t M, int C) {
    typedef SnnStore<true> St;
    const int cv = C / VEC;
    const int64_t total = M * cv;
    for (int64_t e = (int64_t)blockIdx.x * kThreads + threadIdx.x; e < total; e += (int64_t)gridDim.x * kThreads) {
        const int64_t m = e / cv;
        const int c = (int)(e % cv) * VEC;
        if (VEC == 4) {
            f32x4 v = St::ld4(a, m * lda + c);
            if (MODE == 1) v += St::ld4(b, m * ldb + c);
            St::st4(dst, m * ldd + c, v);
        } else {
            float v = St::ld1(a, m * lda + c);
            if (MODE == 1) v += St::ld1(b, m * ldb + c);
            St::st1(dst, m * ldd + c, v);
        }
    }
}

// fp32 <-> bf16 (round to nearest even), dense: the boundary of the bf16-storage domain (the head's last-step read-out)
template <bool TO_BF16>
__global__ void k_convert_bf16(const void* __restrict__ src, void* __restrict__ dst, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < n; i += (int64_t)gridDim.x * kThreads) {
        if (TO_BF16) SnnStore<true>::st1(dst, i, static_cast<const float*>(src)[i]);
        else static_cast<float*>(dst)[i] = SnnStore<true>::ld1(src, i);
    }
}

// ------------------------------------------------------------------------------------------ activations
__device__ __forceinline__ float act_f(int act, float x) {
    switch (act) {
        case SNN_ACT_RELU: return x > 0.0f ? x : 0.0f;
        case SNN_ACT_SILU: return x / (1.0f + expf(-x));
        default: return tanhf(x);
    }
}
__device__ __forceinline__ float act_g(int act, float x, float y) {
    switch (act) {
        case SNN_ACT_RELU: return x > 0.0f ? 1.0f : 0.0f;
        case SNN_ACT_SILU: {
            float s = 1.0f / (1.0f + expf(-x));
            return s * (1.0f + x * (1.0f - s));
        }
        default: return 1.0f - y * y;
    }
}
__global__ void k_act_fwd(int act, const float* __restrict__ x, float* __restrict__ y, int64_t n) {
    for (int64_t e = (int64_t)blockIdx.x * kThreads + threadIdx.x; e < n; e += (int64_t)gridDim.x * kThreads)
        y[e] = act_f(act, x[e]);
}
__global__ void k_act_bwd(int act, const float* __restrict__ x, const float* __restrict__ y,
                          const float* __restrict__ gy, float* __restrict__ gx, int64_t n) {
    for (int64_t e = (int64_t)blockIdx.x * kThreads + threadIdx.x; e < n; e += (int64_t)gridDim.x * kThreads)
        gx[e] = gy[e] * act_g(act, x[e], y[e]);
}

// ------------------------------------------------------------------------------------------ ConvLSTM cell
// gates[m][4C] = (input, forget, output, candidate) pre-activations, conv_lstm.py:66-76
__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

__global__ void k_lstm_fwd(const float* __restrict__ gates, const float* __restrict__ c_prev, float* __restrict__ h,
                           float* __restrict__ c, int64_t M, int C) {
    const int64_t total = M * C;
    for (int64_t e = (int64_t)blockIdx.x * kThreads + threadIdx.x; e < total; e += (int64_t)gridDim.x * kThreads) {
        const int64_t m = e / C;
        const int ch = (int)(e % C);
        const float* g = gates + m * 4 * C;
        const float I = sigmoidf_(g[ch]), F = sigmoidf_(g[C + ch]), O = sigmoidf_(g[2 * C + ch]);
        const float G = tanhf(g[3 * C + ch]);
        const float cp = c_prev ? c_prev[e] : 0.0f;
        const float cn = F * cp + I * G;
        c[e] = cn;
        h[e] = O * tanhf(cn);
    }
}

__global__ void k_lstm_bwd(const float* __restrict__ gates, const float* __restrict__ c_prev,
                           const float* __restrict__ c, const float* __restrict__ gh, const float* __restrict__ gc,
                           float* __restrict__ g_gates, float* __restrict__ g_c_prev, int64_t M, int C) {
    const int64_t total = M * C;
    for (int64_t e = (int64_t)blockIdx.x * kThreads + threadIdx.x; e < total; e += (int64_t)gridDim.x * kThreads) {
        const int64_t m = e / C;
        const int ch = (int)(e % C);
        const float* g = gates + m * 4 * C;
        const float I = sigmoidf_(g[ch]), F = sigmoidf_(g[C + ch]), O = sigmoidf_(g[2 * C + ch]);
        const float G = tanhf(g[3 * C + ch]);
        const float tc = tanhf(c[e]);
        const float dh = gh ? gh[e] : 0.0f;
        const float dc = (gc ? gc[e] : 0.0f) + dh * O * (1.0f - tc * tc);
        const float cp = c_prev ? c_prev[e] : 0.0f;
        float* d = g_gates + m * 4 * C;
        d[ch] = (dc * G) * (I * (1.0f - I));
        d[C + ch] = (dc * cp) * (F * (1.0f - F));
        d[2 * C + ch] = (dh * tc) * (O * (1.0f - O));
        d[3 * C + ch] = (dc * I) * (1.0f - G * G);
        if (g_c_prev) g_c_prev[e] = dc * F;
    }
}

// ------------------------------------------------------------------------------------------ pooling
__global__ void k_pool_fwd(int kind, const float* __restrict__ x, float* __restrict__ y, int64_t N, int H, int W, int C,
                           int Ho, int Wo, int k, int stride) {
    const int64_t total = N * Ho * Wo * C;
    for (int64_t e = (int64_t)blockIdx.x * kThreads + threadIdx.x; e < total; e += (int64_t)gridDim.x * kThreads) {
        int c = (int)(e % C);
        int64_t r = e / C;
        int wo = (int)(r % Wo);
        r /= Wo;
        int ho = (int)(r % Ho);
        int64_t n = r / Ho;
        float acc = (kind == SNN_POOL_MAX) ? -INFINITY : 0.0f;
        for (int i = 0; i < k; ++i)
            for (int j = 0; j < k; ++j) {
                float v = x[((n * H + ho * stride + i) * W + wo * stride + j) * C + c];
                if (kind == SNN_POOL_MAX) acc = (v > acc || v != v) ? v : acc;
                else acc += v;
            }
        if (kind == SNN_POOL_AVG) acc = acc / (float)(k * k);
        // SumPool2d = avg_pool2d * k * k (common.py:44-48): divide then multiply twice, as the reference does
        if (kind == SNN_POOL_SUM) acc = ((acc / (float)(k * k)) * (float)k) * (float)k;
        y[e] = acc;
    }
}

// one thread per INPUT element: gathers from every window that covers it (windows overlap when stride < k)
__global__ void k_pool_bwd(int kind, const float* __restrict__ x, const float* __restrict__ gy, float* __restrict__ gx,
                           int64_t N, int H, int W, int C, int Ho, int Wo, int k, int stride) {
    const int64_t total = N * H * W * C;
    for (int64_t e = (int64_t)blockIdx.x * kThreads + threadIdx.x; e < total; e += (int64_t)gridDim.x * kThreads) {
        int c = (int)(e % C);
        int64_t r = e / C;
        int w = (int)(r % W);
        r /= W;
        int h = (int)(r % H);
        int64_t n = r / H;
        float acc = 0.0f;
        int ho_lo = (h - k + stride) / stride;
        if (h - k + 1 < 0) ho_lo = 0;
        int wo_lo = (w - k + stride) / stride;
        if (w - k + 1 < 0) wo_lo = 0;
        for (int ho = ho_lo; ho < Ho && ho * stride <= h; ++ho)
            for (int wo = wo_lo; wo < Wo && wo * stride <= w; ++wo) {
                if (h - ho * stride >= k || w - wo * stride >= k) continue;
                float g = gy[((n * Ho + ho) * Wo + wo) * C + c];
                if (kind == SNN_POOL_MAX) {
                    // gradient goes to the FIRST maximal element of the window (ATen max_pool2d)
                    float best = -INFINITY;
                    int bi = 0, bj = 0;
                    for (int i = 0; i < k; ++i)
                        for (int j = 0; j < k; ++j) {
                            float v = x[((n * H + ho * stride + i) * W + wo * stride + j) * C + c];
                            if (v > best || v != v) { best = v; bi = i; bj = j; }
                        }
                    if (ho * stride + bi == h && wo * stride + bj == w) acc += g;
                } else if (kind == SNN_POOL_AVG) {
                    acc += g / (float)(k * k);
                } else {
                    acc += ((g * (float)k) * (float)k) / (float)(k * k);
                }
            }
        gx[e] = acc;
    }
}

__global__ void k_upsample_fwd(const float* __restrict__ x, float* __restrict__ y, int64_t N, int H, int W, int C,
                               int s) {
    const int Ho = H * s, Wo = W * s;
    const int64_t total = N * Ho * Wo * C;
    for (int64_t e = (int64_t)blockIdx.x * kThreads + threadIdx.x; e < total; e += (int64_t)gridDim.x * kThreads) {
        int c = (int)(e % C);
        int64_t r = e / C;
        int wo = (int)(r % Wo);
        r /= Wo;
        int ho = (int)(r % Ho);
        int64_t n = r / Ho;
        y[e] = x[((n * H + ho / s) * W + wo / s) * C + c];
    }
}
__global__ void k_upsample_bwd(const float* __restrict__ gy, float* __restrict__ gx, int64_t N, int H, int W, int C,
                               int s) {
    const int Ho = H * s, Wo = W * s;
    const int64_t total = N * H * W * C;
    for (int64_t e = (int64_t)blockIdx.x * kThreads + threadIdx.x; e < total; e += (int64_t)gridDim.x * kThreads) {
        int c = (int)(e % C);
        int64_t r = e / C;
        int w = (int)(r % W);
        r /= W;
        int h = (int)(r % H);
        int64_t n = r / H;
        float acc = 0.0f;
        for (int i = 0; i < s; ++i)
            for (int j = 0; j < s; ++j) acc += gy[((n * Ho + h * s + i) * Wo + w * s + j) * C + c];
        gx[e] = acc;
    }
}

// ------------------------------------------------------------------------------------------ optimizer
__global__ void k_adamax(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                         float* __restrict__ u, int64_t n, float w1, float b2, float eps, float clr, float gscale) {
    for (int64_t e = (int64_t)blockIdx.x * kThreads + threadIdx.x; e < n; e += (int64_t)gridDim.x * kThreads) {
        float ge = g[e] * gscale;  // gscale = 1/world_size after a SUM all-reduce (exact for powers of two)
        // torch.optim.Adamax single-tensor: exp_avg.lerp_(grad, 1-beta1); exp_inf = max(exp_inf*beta2, |g|+eps)
        float me = m[e] + w1 * (ge - m[e]);
        float ue = fmaxf(u[e] * b2, fabsf(ge) + eps);
        m[e] = me;
        u[e] = ue;
        p[e] = p[e] - clr * (me / ue);
    }
}

// ------------------------------------------------------------------------------------------ events
__global__ void k_events(const int32_t* __restrict__ tb, const int32_t* __restrict__ xs, const int32_t* __restrict__ ys,
                         const int32_t* __restrict__ ps, int64_t n, float* __restrict__ frames, int T, int H, int W) {
    for (int64_t e = (int64_t)blockIdx.x * kThreads + threadIdx.x; e < n; e += (int64_t)gridDim.x * kThreads) {
        int t = tb[e], x = xs[e], y = ys[e], p = ps[e];
        if (t < 0 || t >= T || y < 0 || y >= H || p < 0 || p > 1) continue;
        x = x < 0 ? 0 : (x > W - 1 ? W - 1 : x);
        frames[(((int64_t)t * H + y) * W + x) * 2 + p] = 1.0f;  // idempotent store: races are benign
    }
}

// ------------------------------------------------------------------------------------------ small GEMM
// C (+)= op(A) x op(B) for weight-sized matrices (<= a few hundred per side): the three products of a composed pair of
// 1x1 convolutions (functional._ComposedConv1x1: w2 w1, G w1^T, w2^T G).  32 x 32 tile per block, 2 x 2 per thread;
// K goes through LDS in chunks of 128 whose loads are all in flight at once (these launches are latency-, not
// throughput-bound: one memory round trip per chunk instead of one per 16 k); every element is an fp32 fmaf chain in
// k order (same arithmetic as the fp32 MFMA).
constexpr int GEMM_KC = 128;
template <bool TA, bool TB>
__device__ __forceinline__ void small_gemm_tile(const float* __restrict__ A, int64_t lda, const float* __restrict__ B,
                                                int64_t ldb, float* __restrict__ C, int64_t ldc, int M, int N, int K,
                                                int accumulate, float* __restrict__ Ct, int64_t ldct, int m0, int n0) {
    __shared__ float As[GEMM_KC][33], Bs[GEMM_KC][33];
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    float acc[2][2] = {{0.f, 0.f}, {0.f, 0.f}};
    for (int k0 = 0; k0 < K; k0 += GEMM_KC) {
        float ra[GEMM_KC * 32 / kThreads], rb[GEMM_KC * 32 / kThreads];
#pragma unroll
        for (int r = 0; r < GEMM_KC * 32 / kThreads; ++r) {   // lanes run along the contiguous dimension of each operand
            const int idx = threadIdx.x + r * kThreads;
            const int ak = TA ? idx >> 5 : idx & (GEMM_KC - 1), am = TA ? idx & 31 : idx / GEMM_KC;
            const int bk = TB ? idx & (GEMM_KC - 1) : idx >> 5, bn = TB ? idx / GEMM_KC : idx & 31;
            const int m = m0 + am, n = n0 + bn;
            ra[r] = (m < M && k0 + ak < K) ? (TA ? A[(int64_t)(k0 + ak) * lda + m] : A[(int64_t)m * lda + k0 + ak]) : 0.f;
            rb[r] = (n < N && k0 + bk < K) ? (TB ? B[(int64_t)n * ldb + k0 + bk] : B[(int64_t)(k0 + bk) * ldb + n]) : 0.f;
        }
        __syncthreads();   // the previous chunk has been consumed
#pragma unroll
        for (int r = 0; r < GEMM_KC * 32 / kThreads; ++r) {
            const int idx = threadIdx.x + r * kThreads;
            const int ak = TA ? idx >> 5 : idx & (GEMM_KC - 1), am = TA ? idx & 31 : idx / GEMM_KC;
            const int bk = TB ? idx & (GEMM_KC - 1) : idx >> 5, bn = TB ? idx / GEMM_KC : idx & 31;
            As[ak][am] = ra[r];
            Bs[bk][bn] = rb[r];
        }
        __syncthreads();
        const int kend = K - k0 < GEMM_KC ? K - k0 : GEMM_KC;
        for (int kk = 0; kk < kend; ++kk)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = fmaf(As[kk][ty + 16 * i], Bs[kk][tx + 16 * j], acc[i][j]);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int m = m0 + ty + 16 * i, n = n0 + tx + 16 * j;
            if (m < M && n < N) {
                float* c = C + (int64_t)m * ldc + n;
                *c = accumulate ? *c + acc[i][j] : acc[i][j];
                if (Ct) Ct[(int64_t)n * ldct + m] = acc[i][j];   // the transposed copy (weight-sized: strided stores are fine)
            }
        }
}

template <bool TA, bool TB>
__global__ __launch_bounds__(kThreads) void k_small_gemm(const float* __restrict__ A, int64_t lda,
                                                         const float* __restrict__ B, int64_t ldb,
                                                         float* __restrict__ C, int64_t ldc, int M, int N, int K,
                                                         int accumulate, float* __restrict__ Ct, int64_t ldct) {
    small_gemm_tile<TA, TB>(A, lda, B, ldb, C, ldc, M, N, K, accumulate, Ct, ldct, blockIdx.y * 32, blockIdx.x * 32);
}

// n independent products C_i = A_i B_i (+ the transposed copy) in one launch: table row {A offset, B offset, C offset, Ct
// offset, M, N, K, ldct} in floats relative to base_a / base_b / base_c / base_ct; blockIdx.y = the product, blockIdx.x = its
// tile (blocks past a product's last tile leave at once).  Dense row-major operands; ldct (0 = M) lets several products
// write column blocks of ONE transposed matrix (row-stacked sibling weights).
__global__ __launch_bounds__(kThreads) void k_small_gemm_batched(const float* __restrict__ base_a,
                                                                 const float* __restrict__ base_b,
                                                                 float* __restrict__ base_c, float* __restrict__ base_ct,
                                                                 const int64_t* __restrict__ table) {
    const int64_t* row = table + (int64_t)blockIdx.y * 8;
    const int M = (int)row[4], N = (int)row[5], K = (int)row[6];
    const int64_t ldct = row[7] > 0 ? row[7] : M;
    const int tn = (N + 31) / 32, tm = (M + 31) / 32;
    if ((int)blockIdx.x >= tn * tm) return;   // whole block, before any barrier
    small_gemm_tile<false, false>(base_a + row[0], K, base_b + row[1], N, base_c + row[2], N, M, N, K, 0,
                                  base_ct ? base_ct + row[3] : nullptr, ldct, ((int)blockIdx.x / tn) * 32,
                                  ((int)blockIdx.x % tn) * 32);
}

}  // namespace

// -------------------------------------------------------------------------------------------- C ABI
extern "C" int snn_nchw_to_nhwc(const float* src, float* dst, int64_t N, int C, int H, int W, void* stream) {
    SNN_REQUIRE(src && dst && N > 0 && C > 0 && H > 0 && W > 0, "snn_nchw_to_nhwc: bad arguments");
    const int64_t HW = (int64_t)H * W;
    hipStream_t st = (hipStream_t)stream;
    if (C == 1) {
        hipError_t ce = hipMemcpyAsync(dst, src, sizeof(float) * N * HW, hipMemcpyDeviceToDevice, st);
        SNN_REQUIRE(ce == hipSuccess, "snn_nchw_to_nhwc: copy failed: %s", hipGetErrorString(ce));
    } else if (C == 2) {
        hipLaunchKernelGGL(k_nchw_to_nhwc_small<2>, dim3(grid_for(N * HW)), dim3(kThreads), 0, st, src, dst, N, HW);
    } else if (C == 3) {
        hipLaunchKernelGGL(k_nchw_to_nhwc_small<3>, dim3(grid_for(N * HW)), dim3(kThreads), 0, st, src, dst, N, HW);
    } else if (C == 4) {
        hipLaunchKernelGGL(k_nchw_to_nhwc_small<4>, dim3(grid_for(N * HW)), dim3(kThreads), 0, st, src, dst, N, HW);
    } else {
        SNN_REQUIRE(N <= 65535, "snn_nchw_to_nhwc: more than 65535 frames");
        dim3 grid((unsigned)snn_ceil_div(HW, 32), (unsigned)snn_ceil_div(C, 32), (unsigned)N);
        hipLaunchKernelGGL(k_transpose_cp, grid, dim3(32, 8), 0, st, src, dst, C, (int)HW);
    }
    SNN_CHECK_LAUNCH("snn_nchw_to_nhwc");
    return 0;
}

extern "C" int snn_nhwc_to_nchw(const float* src, float* dst, int64_t N, int C, int H, int W, void* stream) {
    SNN_REQUIRE(src && dst && N > 0 && C > 0 && H > 0 && W > 0, "snn_nhwc_to_nchw: bad arguments");
    SNN_REQUIRE(N <= 65535, "snn_nhwc_to_nchw: more than 65535 frames");
    const int64_t HW = (int64_t)H * W;
    dim3 grid((unsigned)snn_ceil_div(C, 32), (unsigned)snn_ceil_div(HW, 32), (unsigned)N);
    hipLaunchKernelGGL(k_transpose_cp, grid, dim3(32, 8), 0, (hipStream_t)stream, src, dst, (int)HW, C);
    SNN_CHECK_LAUNCH("snn_nhwc_to_nchw");
    return 0;
}

extern "C" int snn_weight_transpose(const float* w, float* wt, int Cout, int KH, int KW, int Cin, void* stream) {
    SNN_REQUIRE(w && wt && Cout > 0 && KH > 0 && KW > 0 && Cin > 0, "snn_weight_transpose: bad arguments");
    int64_t n = (int64_t)Cout * KH * KW * Cin;
    hipLaunchKernelGGL(k_weight_transpose, dim3(grid_for(n)), dim3(kThreads), 0, (hipStream_t)stream, w, wt, Cout,
                       KH * KW, Cin);
    SNN_CHECK_LAUNCH("snn_weight_transpose");
    return 0;
}

extern "C" int snn_weight_transpose_batched(const float* flat_w, float* flat_wt, const int64_t* table, int n_layers,
                                            void* stream) {
    SNN_REQUIRE(flat_w && flat_wt && table && n_layers > 0 && n_layers <= 65535,
                "snn_weight_transpose_batched: bad arguments");
    hipLaunchKernelGGL(k_weight_transpose_batched, dim3(64, (unsigned)n_layers), dim3(kThreads), 0, (hipStream_t)stream,
                       flat_w, flat_wt, table);
    SNN_CHECK_LAUNCH("snn_weight_transpose_batched");
    return 0;
}

static int channels_op(bool add, const float* src, int64_t lds, float* dst, int64_t ldd, int64_t M, int C,
                       void* stream) {
    SNN_REQUIRE(src && dst && M > 0 && C > 0 && lds >= C && ldd >= C, "snn_%s_channels: bad arguments",
                add ? "add" : "copy");
    bool v4 = C % 4 == 0 && lds % 4 == 0 && ldd % 4 == 0 && aligned16(src) && aligned16(dst);
    int64_t n = M * (v4 ? C / 4 : C);
    hipStream_t st = (hipStream_t)stream;
    if (v4 && add) hipLaunchKernelGGL((k_channels<4, true>), dim3(grid_for(n)), dim3(kThreads), 0, st, src, lds, dst, ldd, M, C);
    else if (v4) hipLaunchKernelGGL((k_channels<4, false>), dim3(grid_for(n)), dim3(kThreads), 0, st, src, lds, dst, ldd, M, C);
    else if (add) hipLaunchKernelGGL((k_channels<1, true>), dim3(grid_for(n)), dim3(kThreads), 0, st, src, lds, dst, ldd, M, C);
    else hipLaunchKernelGGL((k_channels<1, false>), dim3(grid_for(n)), dim3(kThreads), 0, st, src, lds, dst, ldd, M, C);
    SNN_CHECK_LAUNCH("snn_channels");
    return 0;
}
extern "C" int snn_copy_channels(const float* src, int64_t lds, float* dst, int64_t ldd, int64_t M, int C, void* stream) {
    return channels_op(false, src, lds, dst, ldd, M, C, stream);
}
extern "C" int snn_add_channels(const float* src, int64_t lds, float* dst, int64_t ldd, int64_t M, int C, void* stream) {
    return channels_op(true, src, lds, dst, ldd, M, C, stream);
}

extern "C" int snn_add(const float* a, int64_t lda, const float* b, int64_t ldb, float* dst, int64_t ldd, int64_t M,
                       int C, void* stream) {
    SNN_REQUIRE(a && b && dst && M > 0 && C > 0 && lda >= C && ldb >= C && ldd >= C, "snn_add: bad arguments");
    bool v4 = C % 4 == 0 && lda % 4 == 0 && ldb % 4 == 0 && ldd % 4 == 0 && aligned16(a) && aligned16(b) &&
              aligned16(dst);
    if (v4) hipLaunchKernelGGL(k_add<4>, dim3(grid_for(M * (C / 4))), dim3(kThreads), 0, (hipStream_t)stream, a, lda,
                               b, ldb, dst, ldd, M, C);
    else hipLaunchKernelGGL(k_add<1>, dim3(grid_for(M * C)), dim3(kThreads), 0, (hipStream_t)stream, a, lda, b, ldb,
                            dst, ldd, M, C);
    SNN_CHECK_LAUNCH("snn_add");
    return 0;
}

static bool aligned8(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 7u) == 0; }

extern "C" int snn_copy_channels_bf16(const float* src, int64_t lds, float* dst, int64_t ldd, int64_t M, int C, void* stream) {
    SNN_REQUIRE(src && dst && M > 0 && C > 0 && lds >= C && ldd >= C, "snn_copy_channels_bf16: bad arguments");
    const bool v4 = C % 4 == 0 && lds % 4 == 0 && ldd % 4 == 0 && aligned8(src) && aligned8(dst);
    if (v4) hipLaunchKernelGGL((k_merge_bf16<4, 0>), dim3(grid_for(M * (C / 4))), dim3(kThreads), 0, (hipStream_t)stream, src,
                               lds, nullptr, 0, dst, ldd, M, C);
    else hipLaunchKernelGGL((k_merge_bf16<1, 0>), dim3(grid_for(M * C)), dim3(kThreads), 0, (hipStream_t)stream, src, lds,
                            nullptr, 0, dst, ldd, M, C);
    SNN_CHECK_LAUNCH("snn_copy_channels_bf16");
    return 0;
}

extern "C" int snn_add_bf16(const float* a, int64_t lda, const float* b, int64_t ldb, float* dst, int64_t ldd, int64_t M,
                            int C, void* stream) {
    SNN_REQUIRE(a && b && dst && M > 0 && C > 0 && lda >= C && ldb >= C && ldd >= C, "snn_add_bf16: bad arguments");
    const bool v4 = C % 4 == 0 && lda % 4 == 0 && ldb % 4 == 0 && ldd % 4 == 0 && aligned8(a) && aligned8(b) && aligned8(dst);
    if (v4) hipLaunchKernelGGL((k_merge_bf16<4, 1>), dim3(grid_for(M * (C / 4))), dim3(kThreads), 0, (hipStream_t)stream, a,
                               lda, b, ldb, dst, ldd, M, C);
    else hipLaunchKernelGGL((k_merge_bf16<1, 1>), dim3(grid_for(M * C)), dim3(kThreads), 0, (hipStream_t)stream, a, lda, b,
                            ldb, dst, ldd, M, C);
    SNN_CHECK_LAUNCH("snn_add_bf16");
    return 0;
}

extern "C" int snn_convert_bf16(const void* src, void* dst, int64_t n, int to_bf16, void* stream) {
    SNN_REQUIRE(src && dst && n > 0, "snn_convert_bf16: bad arguments");
    if (to_bf16) hipLaunchKernelGGL(k_convert_bf16<true>, dim3(grid_for(n)), dim3(kThreads), 0, (hipStream_t)stream, src, dst, n);
    else hipLaunchKernelGGL(k_convert_bf16<false>, dim3(grid_for(n)), dim3(kThreads), 0, (hipStream_t)stream, src, dst, n);
    SNN_CHECK_LAUNCH("snn_convert_bf16");
    return 0;
}

extern "C" int snn_act_fwd(int act, const float* x, float* y, int64_t n, void* stream) {
    SNN_REQUIRE(x && y && n > 0 && act >= SNN_ACT_RELU && act <= SNN_ACT_TANH, "snn_act_fwd: bad arguments");
    hipLaunchKernelGGL(k_act_fwd, dim3(grid_for(n)), dim3(kThreads), 0, (hipStream_t)stream, act, x, y, n);
    SNN_CHECK_LAUNCH("snn_act_fwd");
    return 0;
}
extern "C" int snn_act_bwd(int act, const float* x, const float* y, const float* gy, float* gx, int64_t n,
                           void* stream) {
    SNN_REQUIRE(x && y && gy && gx && n > 0 && act >= SNN_ACT_RELU && act <= SNN_ACT_TANH, "snn_act_bwd: bad arguments");
    hipLaunchKernelGGL(k_act_bwd, dim3(grid_for(n)), dim3(kThreads), 0, (hipStream_t)stream, act, x, y, gy, gx, n);
    SNN_CHECK_LAUNCH("snn_act_bwd");
    return 0;
}

extern "C" int snn_small_gemm(const float* A, int64_t lda, int transA, const float* B, int64_t ldb, int transB, float* C,
                              int64_t ldc, int M, int N, int K, int accumulate, float* Ct, int64_t ldct, void* stream) {
    SNN_REQUIRE(A && B && C && M > 0 && N > 0 && K > 0, "snn_small_gemm: bad arguments");
    SNN_REQUIRE(lda >= (transA ? M : K) && ldb >= (transB ? K : N) && ldc >= N, "snn_small_gemm: leading dimension "
                "smaller than the row length");
    SNN_REQUIRE(!Ct || (ldct >= M && !accumulate), "snn_small_gemm: the transposed copy needs ldct >= M and accumulate == 0");
    dim3 grid((unsigned)snn_ceil_div(N, 32), (unsigned)snn_ceil_div(M, 32));
    hipStream_t st = (hipStream_t)stream;
    if (transA && transB) hipLaunchKernelGGL((k_small_gemm<true, true>), grid, dim3(kThreads), 0, st, A, lda, B, ldb, C, ldc, M, N, K, accumulate, Ct, ldct);
    else if (transA) hipLaunchKernelGGL((k_small_gemm<true, false>), grid, dim3(kThreads), 0, st, A, lda, B, ldb, C, ldc, M, N, K, accumulate, Ct, ldct);
    else if (transB) hipLaunchKernelGGL((k_small_gemm<false, true>), grid, dim3(kThreads), 0, st, A, lda, B, ldb, C, ldc, M, N, K, accumulate, Ct, ldct);
    else hipLaunchKernelGGL((k_small_gemm<false, false>), grid, dim3(kThreads), 0, st, A, lda, B, ldb, C, ldc, M, N, K, accumulate, Ct, ldct);
    SNN_CHECK_LAUNCH("snn_small_gemm");
    return 0;
}

extern "C" int snn_small_gemm_batched(const float* base_a, const float* base_b, float* base_c, float* base_ct,
                                      const int64_t* table, int n, int max_tiles, void* stream) {
    SNN_REQUIRE(base_a && base_b && base_c && table && n > 0 && n <= 65535 && max_tiles > 0,
                "snn_small_gemm_batched: bad arguments");
    hipLaunchKernelGGL(k_small_gemm_batched, dim3((unsigned)max_tiles, (unsigned)n), dim3(kThreads), 0, (hipStream_t)stream,
                       base_a, base_b, base_c, base_ct, table);
    SNN_CHECK_LAUNCH("snn_small_gemm_batched");
    return 0;
}

extern "C" int snn_lstm_cell_fwd(const float* gates, const float* c_prev, float* h, float* c, int64_t M, int C,
                                 void* stream) {
    SNN_REQUIRE(gates && h && c && M > 0 && C > 0, "snn_lstm_cell_fwd: bad arguments");
    hipLaunchKernelGGL(k_lstm_fwd, dim3(grid_for(M * C)), dim3(kThreads), 0, (hipStream_t)stream, gates, c_prev, h, c,
                       M, C);
    SNN_CHECK_LAUNCH("snn_lstm_cell_fwd");
    return 0;
}
extern "C" int snn_lstm_cell_bwd(const float* gates, const float* c_prev, const float* c, const float* gh,
                                 const float* gc, float* g_gates, float* g_c_prev, int64_t M, int C, void* stream) {
    SNN_REQUIRE(gates && c && g_gates && M > 0 && C > 0, "snn_lstm_cell_bwd: bad arguments");
    hipLaunchKernelGGL(k_lstm_bwd, dim3(grid_for(M * C)), dim3(kThreads), 0, (hipStream_t)stream, gates, c_prev, c,
                       gh, gc, g_gates, g_c_prev, M, C);
    SNN_CHECK_LAUNCH("snn_lstm_cell_bwd");
    return 0;
}

extern "C" int snn_pool_fwd(int kind, const float* x, float* y, int64_t N, int H, int W, int C, int Ho, int Wo, int k,
                            int stride, void* stream) {
    SNN_REQUIRE(x && y && N > 0 && H > 0 && W > 0 && C > 0 && k > 0 && stride > 0, "snn_pool_fwd: bad arguments");
    SNN_REQUIRE(kind >= SNN_POOL_AVG && kind <= SNN_POOL_SUM, "snn_pool_fwd: bad pool kind %d", kind);
    SNN_REQUIRE(Ho == (H - k) / stride + 1 && Wo == (W - k) / stride + 1 && Ho > 0 && Wo > 0,
                "snn_pool_fwd: output size mismatch");
    hipLaunchKernelGGL(k_pool_fwd, dim3(grid_for(N * Ho * Wo * C)), dim3(kThreads), 0, (hipStream_t)stream, kind, x, y,
                       N, H, W, C, Ho, Wo, k, stride);
    SNN_CHECK_LAUNCH("snn_pool_fwd");
    return 0;
}
extern "C" int snn_pool_bwd(int kind, const float* x, const float* gy, float* gx, int64_t N, int H, int W, int C,
                            int Ho, int Wo, int k, int stride, void* stream) {
    SNN_REQUIRE(gy && gx && N > 0 && H > 0 && W > 0 && C > 0 && k > 0 && stride > 0, "snn_pool_bwd: bad arguments");
    SNN_REQUIRE(kind >= SNN_POOL_AVG && kind <= SNN_POOL_SUM, "snn_pool_bwd: bad pool kind %d", kind);
    SNN_REQUIRE(kind != SNN_POOL_MAX || x, "snn_pool_bwd: max pooling needs the forward input");
    hipLaunchKernelGGL(k_pool_bwd, dim3(grid_for(N * H * W * C)), dim3(kThreads), 0, (hipStream_t)stream, kind, x, gy,
                       gx, N, H, W, C, Ho, Wo, k, stride);
    SNN_CHECK_LAUNCH("snn_pool_bwd");
    return 0;
}

extern "C" int snn_upsample_fwd(const float* x, float* y, int64_t N, int H, int W, int C, int scale, void* stream) {
    SNN_REQUIRE(x && y && N > 0 && H > 0 && W > 0 && C > 0 && scale > 0, "snn_upsample_fwd: bad arguments");
    hipLaunchKernelGGL(k_upsample_fwd, dim3(grid_for(N * H * scale * W * scale * C)), dim3(kThreads), 0,
                       (hipStream_t)stream, x, y, N, H, W, C, scale);
    SNN_CHECK_LAUNCH("snn_upsample_fwd");
    return 0;
}
extern "C" int snn_upsample_bwd(const float* gy, float* gx, int64_t N, int H, int W, int C, int scale, void* stream) {
    SNN_REQUIRE(gy && gx && N > 0 && H > 0 && W > 0 && C > 0 && scale > 0, "snn_upsample_bwd: bad arguments");
    hipLaunchKernelGGL(k_upsample_bwd, dim3(grid_for(N * H * W * C)), dim3(kThreads), 0, (hipStream_t)stream, gy, gx, N,
                       H, W, C, scale);
    SNN_CHECK_LAUNCH("snn_upsample_bwd");
    return 0;
}

extern "C" int snn_adamax_step(float* param, const float* grad, float* exp_avg, float* exp_inf, int64_t n, float lr,
                               float beta1, float beta2, float eps, int step, float grad_scale, void* stream) {
    SNN_REQUIRE(param && grad && exp_avg && exp_inf && n > 0 && step >= 1, "snn_adamax_step: bad arguments");
    // clr = lr / (1 - beta1^step)
    double bias_corr = 1.0 - pow((double)beta1, (double)step);
    float clr = (float)((double)lr / bias_corr);
    float w1 = (float)(1.0 - (double)beta1);
    hipLaunchKernelGGL(k_adamax, dim3(grid_for(n)), dim3(kThreads), 0, (hipStream_t)stream, param, grad, exp_avg,
                       exp_inf, n, w1, beta2, eps, clr, grad_scale);
    SNN_CHECK_LAUNCH("snn_adamax_step");
    return 0;
}

extern "C" int snn_events_to_frames(const int32_t* t_bin, const int32_t* x, const int32_t* y, const int32_t* p,
                                    int64_t n_events, float* frames, int T, int H, int W, void* stream) {
    SNN_REQUIRE(frames && T > 0 && H > 0 && W > 0 && n_events >= 0, "snn_events_to_frames: bad arguments");
    hipError_t me = hipMemsetAsync(frames, 0, sizeof(float) * (size_t)T * H * W * 2, (hipStream_t)stream);
    SNN_REQUIRE(me == hipSuccess, "snn_events_to_frames: memset failed: %s", hipGetErrorString(me));
    if (n_events == 0) return 0;
    SNN_REQUIRE(t_bin && x && y && p, "snn_events_to_frames: null event arrays");
    hipLaunchKernelGGL(k_events, dim3(grid_for(n_events)), dim3(kThreads), 0, (hipStream_t)stream, t_bin, x, y, p,
                       n_events, frames, T, H, W);
    SNN_CHECK_LAUNCH("snn_events_to_frames");
    return 0;
}
