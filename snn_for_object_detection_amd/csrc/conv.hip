// Implicit-GEMM 2-D convolution on the gfx950 fp32 matrix cores (v_mfma_f32_32x32x2_f32),
// LDS-tiled, channels-last, all T*B frames of a layer in one launch.
//
// Replaces nn.Conv2d(bias=False, padding=int(k/2)) forward and ATen's conv backward
// (reference layer_gen.py:129-136) for the layer-major schedule.
//
//   forward / data-gradient ("gather conv", one kernel, two pixel mappings):
//       out[m][n] = sum_k A[m][k] * Wk[n][k]
//       m = output pixel (img, oy, ox); k = (tap, c); n = output channel
//       FWD  : A = x [img][oy*s-pad+kh][ox*s-pad+kw][c],             Wk = w  [Cout][taps][Cin]
//       DGRAD: A = dy[img][(oy+pad-kh)/s][(ox+pad-kw)/s][c] (exact), Wk = wt [Cin][taps][Cout]
//     block tile 128 pixels x BN channels x 32 k, 4 waves; LDS images [row][32+4] fp32 read
//     with ds_read_b128 (the +4 pad makes the four 16-lane groups conflict-free); global ->
//     register prefetch of tile k+1 overlaps the MFMAs of tile k.
//
//   weight-gradient:
//       dw[co][kc] = sum_pix dy[pix][co] * xg[pix][kc],  kc = (tap, ci)
//     block tile 64 x 64 over (co, kc), K = pixels, split over gridDim.z pixel ranges into
//     workspace slabs, reduced in fixed order by k_wgrad_reduce (bitwise reproducible).
//
// fp32 MFMA is an exact k-ordered fmaf chain (no reduced precision), which is what the 1e-4
// parity target against the CPU reference needs.
#include "snn_common.h"

namespace {

constexpr int kThreads = 256;
constexpr int BM = 128;  // output pixels per block
constexpr int BK = 32;   // k elements per LDS stage
constexpr int LDK = BK + 4;

struct ConvGeom {
    int64_t Mtot;      // img * OH * OW
    int IH, IW, IC;    // gathered tensor
    int OH, OW, OC;    // produced tensor
    int KH, KW, stride, pad;
    int64_t ldi, ldo;
    int Ktot;          // KH*KW*IC
};

template <bool DGRAD>
__device__ __forceinline__ bool src_pixel(const ConvGeom& g, int y0, int x0, int kh, int kw, int& iy, int& ix) {
    if (!DGRAD) {
        iy = y0 + kh;
        ix = x0 + kw;
        return (unsigned)iy < (unsigned)g.IH && (unsigned)ix < (unsigned)g.IW;
    } else {
        int ty = y0 - kh, tx = x0 - kw;
        if (ty < 0 || tx < 0) return false;
        if (g.stride == 1) {
            iy = ty;
            ix = tx;
        } else {
            if ((ty % g.stride) != 0 || (tx % g.stride) != 0) return false;
            iy = ty / g.stride;
            ix = tx / g.stride;
        }
        return iy < g.IH && ix < g.IW;
    }
}

template <int BN, int WM, int WN, bool DGRAD, bool VEC>
__global__ __launch_bounds__(kThreads) void k_conv_gather(const float* __restrict__ in, const float* __restrict__ wk,
                                                          float* __restrict__ out, ConvGeom g, int accumulate) {
    constexpr int TM = BM / WM / 32;
    constexpr int TN = BN / WN / 32;
    constexpr int BROWS = BN / 32;  // B rows loaded per thread
    static_assert(WM * WN == 4, "4 waves");
    __shared__ __attribute__((aligned(16))) float As[BM * LDK];
    __shared__ __attribute__((aligned(16))) float Bs[BN * LDK];

    const int tid = threadIdx.x;
    const int lane_id = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int r = lane_id & 31, h = lane_id >> 5;

    const int64_t m0 = (int64_t)blockIdx.x * BM;
    const int n0 = blockIdx.y * BN;

    // ---- per-thread loader geometry: rows lr + 32*j, k offset kq
    const int lr = tid >> 3, kq = (tid & 7) * 4;
    int a_y0[4], a_x0[4];
    int64_t a_base[4];
    bool a_ok[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        int64_t m = m0 + lr + 32 * j;
        a_ok[j] = m < g.Mtot;
        int64_t mm = a_ok[j] ? m : 0;
        int ox = (int)(mm % g.OW);
        int64_t t = mm / g.OW;
        int oy = (int)(t % g.OH);
        int64_t img = t / g.OH;
        a_base[j] = img * g.IH * (int64_t)g.IW;
        if (!DGRAD) {
            a_y0[j] = oy * g.stride - g.pad;
            a_x0[j] = ox * g.stride - g.pad;
        } else {
            a_y0[j] = oy + g.pad;
            a_x0[j] = ox + g.pad;
        }
    }

    f32x4 ra[4], rb[BROWS];

    auto load_tiles = [&](int k0) {
        const int kk = k0 + kq;
        if (VEC) {
            const bool kin = kk < g.Ktot;
            int tap = kk / g.IC, c = kk - tap * g.IC;
            int kh = tap / g.KW, kw = tap - kh * g.KW;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                int iy, ix;
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (kin && a_ok[j] && src_pixel<DGRAD>(g, a_y0[j], a_x0[j], kh, kw, iy, ix))
                    v = *reinterpret_cast<const f32x4*>(in + (a_base[j] + (int64_t)iy * g.IW + ix) * g.ldi + c);
                ra[j] = v;
            }
#pragma unroll
            for (int j = 0; j < BROWS; ++j) {
                int n = n0 + lr + 32 * j;
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (kin && n < g.OC) v = *reinterpret_cast<const f32x4*>(wk + (int64_t)n * g.Ktot + kk);
                rb[j] = v;
            }
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int ke = kk + e;
                const bool kin = ke < g.Ktot;
                int tap = ke / g.IC, c = ke - tap * g.IC;
                int kh = tap / g.KW, kw = tap - kh * g.KW;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    int iy, ix;
                    float v = 0.f;
                    if (kin && a_ok[j] && src_pixel<DGRAD>(g, a_y0[j], a_x0[j], kh, kw, iy, ix))
                        v = in[(a_base[j] + (int64_t)iy * g.IW + ix) * g.ldi + c];
                    ra[j][e] = v;
                }
#pragma unroll
                for (int j = 0; j < BROWS; ++j) {
                    int n = n0 + lr + 32 * j;
                    rb[j][e] = (kin && n < g.OC) ? wk[(int64_t)n * g.Ktot + ke] : 0.f;
                }
            }
        }
    };
    auto store_tiles = [&]() {
#pragma unroll
        for (int j = 0; j < 4; ++j) *reinterpret_cast<f32x4*>(&As[(lr + 32 * j) * LDK + kq]) = ra[j];
#pragma unroll
        for (int j = 0; j < BROWS; ++j) *reinterpret_cast<f32x4*>(&Bs[(lr + 32 * j) * LDK + kq]) = rb[j];
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    load_tiles(0);
    store_tiles();
    __syncthreads();

    for (int k0 = 0; k0 < g.Ktot; k0 += BK) {
        const bool more = (k0 + BK) < g.Ktot;
        if (more) load_tiles(k0 + BK);
#pragma unroll
        for (int ks = 0; ks < BK / 8; ++ks) {
            f32x4 a[TM], b[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i)
                a[i] = *reinterpret_cast<const f32x4*>(&As[((wm * TM + i) * 32 + r) * LDK + ks * 8 + 4 * h]);
#pragma unroll
            for (int j = 0; j < TN; ++j)
                b[j] = *reinterpret_cast<const f32x4*>(&Bs[((wn * TN + j) * 32 + r) * LDK + ks * 8 + 4 * h]);
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][e], b[j][e], acc[i][j], 0, 0, 0);
        }
        __syncthreads();
        if (more) {
            store_tiles();
            __syncthreads();
        }
    }

    // ---- epilogue: C/D layout of the 32x32 MFMA: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = n0 + (wn * TN + j) * 32 + r;
            if (n >= g.OC) continue;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int64_t m = m0 + (wm * TM + i) * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                if (m < g.Mtot) {
                    float* p = out + m * g.ldo + n;
                    float v = acc[i][j][e];
                    if (accumulate) v += *p;
                    *p = v;
                }
            }
        }
}

// ------------------------------------------------------------------------------------------ wgrad
constexpr int WB_M = 64;   // out channels per block
constexpr int WB_N = 64;   // (tap, ci) columns per block
constexpr int WB_K = 32;   // pixels per LDS stage

struct WgradGeom {
    int64_t Mtot;  // N * Ho * Wo
    int H, W, Cin, Ho, Wo, Cout, KH, KW, stride, pad;
    int64_t ldx, lddy;
    int Ktot;
    int64_t pix_per_split;
};

template <bool VEC>
__global__ __launch_bounds__(kThreads) void k_conv_wgrad(const float* __restrict__ x, const float* __restrict__ dy,
                                                         float* __restrict__ ws, WgradGeom g) {
    __shared__ __attribute__((aligned(16))) float Ds[WB_K * WB_M];
    __shared__ __attribute__((aligned(16))) float Xs[WB_K * WB_N];

    const int tid = threadIdx.x;
    const int lane_id = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int r = lane_id & 31, h = lane_id >> 5;

    const int co0 = blockIdx.x * WB_M;
    const int kc0 = blockIdx.y * WB_N;
    const int64_t p_lo = (int64_t)blockIdx.z * g.pix_per_split;
    int64_t p_hi = p_lo + g.pix_per_split;
    if (p_hi > g.Mtot) p_hi = g.Mtot;

    // loader: pixel rows pr + 16*j (j < 2), 4-wide column group cq
    const int pr = tid >> 4, cq = (tid & 15) * 4;
    // the gathered column (tap, ci) of this thread never changes
    int x_kh[4], x_kw[4], x_ci[4];
    bool x_ok[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        int kc = kc0 + cq + e;
        x_ok[e] = kc < g.Ktot;
        int kcc = x_ok[e] ? kc : 0;
        int tap = kcc / g.Cin;
        x_ci[e] = kcc - tap * g.Cin;
        x_kh[e] = tap / g.KW;
        x_kw[e] = tap - x_kh[e] * g.KW;
    }
    const bool d_ok = VEC ? (co0 + cq < g.Cout) : true;

    f32x4 rd[2], rx[2];
    auto load_tiles = [&](int64_t p0) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int64_t p = p0 + pr + 16 * j;
            f32x4 dv = {0.f, 0.f, 0.f, 0.f}, xv = {0.f, 0.f, 0.f, 0.f};
            if (p < p_hi) {
                int ox = (int)(p % g.Wo);
                int64_t t = p / g.Wo;
                int oy = (int)(t % g.Ho);
                int64_t img = t / g.Ho;
                const int y0 = oy * g.stride - g.pad, x0 = ox * g.stride - g.pad;
                const int64_t ibase = img * g.H * (int64_t)g.W;
                if (VEC) {
                    if (d_ok) dv = *reinterpret_cast<const f32x4*>(dy + p * g.lddy + co0 + cq);
                    if (x_ok[0]) {
                        int iy = y0 + x_kh[0], ix = x0 + x_kw[0];
                        if ((unsigned)iy < (unsigned)g.H && (unsigned)ix < (unsigned)g.W)
                            xv = *reinterpret_cast<const f32x4*>(x + (ibase + (int64_t)iy * g.W + ix) * g.ldx + x_ci[0]);
                    }
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        if (co0 + cq + e < g.Cout) dv[e] = dy[p * g.lddy + co0 + cq + e];
                        if (x_ok[e]) {
                            int iy = y0 + x_kh[e], ix = x0 + x_kw[e];
                            if ((unsigned)iy < (unsigned)g.H && (unsigned)ix < (unsigned)g.W)
                                xv[e] = x[(ibase + (int64_t)iy * g.W + ix) * g.ldx + x_ci[e]];
                        }
                    }
                }
            }
            rd[j] = dv;
            rx[j] = xv;
        }
    };
    auto store_tiles = [&]() {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            *reinterpret_cast<f32x4*>(&Ds[(pr + 16 * j) * WB_M + cq]) = rd[j];
            *reinterpret_cast<f32x4*>(&Xs[(pr + 16 * j) * WB_N + cq]) = rx[j];
        }
    };

    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;

    if (p_lo < p_hi) {
        load_tiles(p_lo);
        store_tiles();
        __syncthreads();
        for (int64_t p0 = p_lo; p0 < p_hi; p0 += WB_K) {
            const bool more = (p0 + WB_K) < p_hi;
            if (more) load_tiles(p0 + WB_K);
#pragma unroll
            for (int ks = 0; ks < WB_K / 2; ++ks) {
                float a = Ds[(ks * 2 + h) * WB_M + wm * 32 + r];
                float b = Xs[(ks * 2 + h) * WB_N + wn * 32 + r];
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
            }
            __syncthreads();
            if (more) {
                store_tiles();
                __syncthreads();
            }
        }
    }

    float* slab = ws + (int64_t)blockIdx.z * g.Cout * (int64_t)g.Ktot;
    const int kc = kc0 + wn * 32 + r;
    if (kc < g.Ktot) {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int co = co0 + wm * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
            if (co < g.Cout) slab[(int64_t)co * g.Ktot + kc] = acc[e];
        }
    }
}

__global__ void k_wgrad_reduce(const float* __restrict__ ws, float* __restrict__ dw, int64_t n, int splitk,
                               int accumulate) {
    for (int64_t e = (int64_t)blockIdx.x * kThreads + threadIdx.x; e < n; e += (int64_t)gridDim.x * kThreads) {
        float s = 0.f;
        for (int k = 0; k < splitk; ++k) s += ws[(int64_t)k * n + e];
        dw[e] = accumulate ? dw[e] + s : s;
    }
}

static bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

template <bool DGRAD>
static int launch_gather(const float* in, const float* wk, float* out, const ConvGeom& g, int accumulate,
                         hipStream_t st, const char* name) {
    const bool vec = (g.IC % 4 == 0) && (g.ldi % 4 == 0) && aligned16(in) && aligned16(wk);
    const int64_t gm = snn_ceil_div(g.Mtot, BM);
    SNN_REQUIRE(gm <= 0x7fffffff, "%s: too many pixels", name);
#define SNN_CONV_LAUNCH(BN_, WM_, WN_)                                                                      \
    do {                                                                                                    \
        dim3 grid((unsigned)gm, (unsigned)snn_ceil_div(g.OC, BN_));                                         \
        if (vec)                                                                                            \
            hipLaunchKernelGGL((k_conv_gather<BN_, WM_, WN_, DGRAD, true>), grid, dim3(kThreads), 0, st, in, \
                               wk, out, g, accumulate);                                                     \
        else                                                                                                \
            hipLaunchKernelGGL((k_conv_gather<BN_, WM_, WN_, DGRAD, false>), grid, dim3(kThreads), 0, st,   \
                               in, wk, out, g, accumulate);                                                 \
    } while (0)
    if (g.OC <= 32) SNN_CONV_LAUNCH(32, 4, 1);
    else if (g.OC <= 64) SNN_CONV_LAUNCH(64, 2, 2);
    else SNN_CONV_LAUNCH(128, 2, 2);
#undef SNN_CONV_LAUNCH
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        snn_set_error("%s: launch failed: %s", name, hipGetErrorString(e));
        return 2;
    }
    return 0;
}

static int check_conv_shape(const char* name, int64_t N, int H, int W, int Cin, int Ho, int Wo, int Cout, int KH,
                            int KW, int stride, int pad) {
    SNN_REQUIRE(N > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0 && KH > 0 && KW > 0 && stride > 0 && pad >= 0,
                "%s: bad shape", name);
    SNN_REQUIRE(Ho == (H + 2 * pad - KH) / stride + 1 && Wo == (W + 2 * pad - KW) / stride + 1 && Ho > 0 && Wo > 0,
                "%s: output size %dx%d does not match input %dx%d k=%dx%d s=%d p=%d", name, Ho, Wo, H, W, KH, KW,
                stride, pad);
    return 0;
}

}  // namespace

extern "C" int snn_conv2d_fwd(const float* x, int64_t ldx, const float* w, float* y, int64_t ldy, int64_t N, int H,
                              int W, int Cin, int Ho, int Wo, int Cout, int KH, int KW, int stride, int pad,
                              int accumulate, void* stream) {
    SNN_REQUIRE(x && w && y, "snn_conv2d_fwd: null pointer");
    if (check_conv_shape("snn_conv2d_fwd", N, H, W, Cin, Ho, Wo, Cout, KH, KW, stride, pad)) return 1;
    SNN_REQUIRE(ldx >= Cin && ldy >= Cout, "snn_conv2d_fwd: pixel stride smaller than channel count");
    ConvGeom g;
    g.Mtot = N * Ho * (int64_t)Wo;
    g.IH = H; g.IW = W; g.IC = Cin;
    g.OH = Ho; g.OW = Wo; g.OC = Cout;
    g.KH = KH; g.KW = KW; g.stride = stride; g.pad = pad;
    g.ldi = ldx; g.ldo = ldy;
    g.Ktot = KH * KW * Cin;
    return launch_gather<false>(x, w, y, g, accumulate, (hipStream_t)stream, "snn_conv2d_fwd");
}

extern "C" int snn_conv2d_dgrad(const float* dy, int64_t lddy, const float* wt, float* dx, int64_t lddx, int64_t N,
                                int H, int W, int Cin, int Ho, int Wo, int Cout, int KH, int KW, int stride, int pad,
                                int accumulate, void* stream) {
    SNN_REQUIRE(dy && wt && dx, "snn_conv2d_dgrad: null pointer");
    if (check_conv_shape("snn_conv2d_dgrad", N, H, W, Cin, Ho, Wo, Cout, KH, KW, stride, pad)) return 1;
    SNN_REQUIRE(lddy >= Cout && lddx >= Cin, "snn_conv2d_dgrad: pixel stride smaller than channel count");
    ConvGeom g;
    g.Mtot = N * H * (int64_t)W;      // one GEMM row per INPUT pixel
    g.IH = Ho; g.IW = Wo; g.IC = Cout;  // gathered tensor is dy
    g.OH = H; g.OW = W; g.OC = Cin;
    g.KH = KH; g.KW = KW; g.stride = stride; g.pad = pad;
    g.ldi = lddy; g.ldo = lddx;
    g.Ktot = KH * KW * Cout;
    return launch_gather<true>(dy, wt, dx, g, accumulate, (hipStream_t)stream, "snn_conv2d_dgrad");
}

extern "C" int snn_conv2d_wgrad_splitk(int64_t N, int Ho, int Wo, int Cin, int Cout, int KH, int KW) {
    if (N <= 0 || Ho <= 0 || Wo <= 0 || Cin <= 0 || Cout <= 0 || KH <= 0 || KW <= 0) return 1;
    const int64_t M = N * Ho * (int64_t)Wo;
    const int64_t Ktot = (int64_t)KH * KW * Cin;
    const int64_t tiles = snn_ceil_div(Cout, WB_M) * snn_ceil_div(Ktot, WB_N);
    int64_t s = snn_ceil_div(4 * SNN_NUM_CU, tiles);                 // ~4 blocks per CU
    const int64_t max_by_work = snn_ceil_div(M, 8 * WB_K);            // >= 8 LDS stages per block
    const int64_t max_by_mem = (int64_t)(64 << 20) / (Cout * Ktot);   // workspace <= 256 MiB
    if (s > max_by_work) s = max_by_work;
    if (s > max_by_mem) s = max_by_mem;
    if (s > 65535) s = 65535;
    if (s < 1) s = 1;
    return (int)s;
}

extern "C" int snn_conv2d_wgrad(const float* x, int64_t ldx, const float* dy, int64_t lddy, float* dw, int64_t N,
                                int H, int W, int Cin, int Ho, int Wo, int Cout, int KH, int KW, int stride, int pad,
                                int accumulate, float* workspace, int splitk, void* stream) {
    SNN_REQUIRE(x && dy && dw && workspace, "snn_conv2d_wgrad: null pointer");
    if (check_conv_shape("snn_conv2d_wgrad", N, H, W, Cin, Ho, Wo, Cout, KH, KW, stride, pad)) return 1;
    SNN_REQUIRE(ldx >= Cin && lddy >= Cout, "snn_conv2d_wgrad: pixel stride smaller than channel count");
    SNN_REQUIRE(splitk >= 1 && splitk <= 65535, "snn_conv2d_wgrad: bad splitk %d", splitk);
    WgradGeom g;
    g.Mtot = N * Ho * (int64_t)Wo;
    g.H = H; g.W = W; g.Cin = Cin; g.Ho = Ho; g.Wo = Wo; g.Cout = Cout;
    g.KH = KH; g.KW = KW; g.stride = stride; g.pad = pad;
    g.ldx = ldx; g.lddy = lddy;
    g.Ktot = KH * KW * Cin;
    g.pix_per_split = snn_ceil_div(snn_ceil_div(g.Mtot, splitk), WB_K) * WB_K;
    const bool vec = (Cin % 4 == 0) && (Cout % 4 == 0) && (ldx % 4 == 0) && (lddy % 4 == 0) && aligned16(x) &&
                     aligned16(dy);
    dim3 grid((unsigned)snn_ceil_div(Cout, WB_M), (unsigned)snn_ceil_div(g.Ktot, WB_N), (unsigned)splitk);
    hipStream_t st = (hipStream_t)stream;
    if (vec) hipLaunchKernelGGL(k_conv_wgrad<true>, grid, dim3(kThreads), 0, st, x, dy, workspace, g);
    else hipLaunchKernelGGL(k_conv_wgrad<false>, grid, dim3(kThreads), 0, st, x, dy, workspace, g);
    SNN_CHECK_LAUNCH("snn_conv2d_wgrad");
    const int64_t n = (int64_t)Cout * g.Ktot;
    int64_t blocks = snn_ceil_div(n, kThreads);
    if (blocks > SNN_MAX_BLOCKS) blocks = SNN_MAX_BLOCKS;
    hipLaunchKernelGGL(k_wgrad_reduce, dim3((unsigned)blocks), dim3(kThreads), 0, st, workspace, dw, n, splitk,
                       accumulate);
    SNN_CHECK_LAUNCH("snn_conv2d_wgrad_reduce");
    return 0;
}
