// Implicit-GEMM 2-D convolution on the gfx950 matrix cores, LDS-tiled, channels-last, all T*B frames of a layer
// in one launch.  Tensors are fp32 in HBM and accumulation is fp32; the PRODUCTS run on the 16-bit matrix pipe from
// pieces of the fp32 operands (split on the way into LDS), selected per call by the `precision` argument
// (include/snn_hip.h, SNN_PREC_*): fp16 x 3 (default forward: v_mfma_f32_32x32x16_f16, fp32-grade), bf16 x 6
// (fp32-grade for any range), bf16 x 3 (default backward: v_mfma_f32_32x32x16_bf16, rel 1e-5), or the exact fp32
// MFMA (v_mfma_f32_32x32x2_f32, a k-ordered fmaf chain).
//
// Replaces nn.Conv2d(bias=False, padding=int(k/2)) forward and ATen's conv backward
// (reference layer_gen.py:129-136) for the layer-major schedule.
//
//   forward / data-gradient ("gather conv", one kernel, two pixel mappings):
//       out[m][n] = sum_k A[m][k] * Wk[n][k]
//       m = output pixel (img, oy, ox); k = (tap, c); n = output channel
//       FWD  : A = x [img][oy*s-pad+kh][ox*s-pad+kw][c],             Wk = w  [Cout][taps][Cin]
//       DGRAD: A = dy[img][(oy+pad-kh)/s][(ox+pad-kw)/s][c] (exact), Wk = wt [Cin][taps][Cout]
//     block tile 128 pixels x BN channels x 32 k, 4 waves; LDS images of 16-bit pieces, [row][32+8] with an 80-byte
//     pitch read with ds_read_b128 (fp32 mode: [row][32+4] floats); software-pipelined main loop (convert tile k+1
//     in the MFMA shadow of tile k, loads of tile k+2 in flight); the data gradient of a strided conv is split
//     into stride x stride phase classes that only visit reachable taps; the epilogue can add up to two
//     same-shaped tensors (fused gradient accumulation).
//
//   weight-gradient:
//       dw[co][kc] = sum_pix dy[pix][co] * xg[pix][kc],  kc = (tap, ci)
//     block tile up to 128 x 128 over (co, kc) (six variants, least padding wins), K = pixels, split over
//     pixel ranges sized to ONE resident wave of blocks, splits pinned to XCDs (L2 reuse of dy / x),
//     workspace slabs reduced in fixed order by k_wgrad_reduce (bitwise reproducible).
//
// The split modes keep the 1e-4 parity target against the CPU reference: see DESIGN.md section 3 for the measured
// errors of each mode.
#include <stdlib.h>
#include <type_traits>
#include "snn_common.h"

#if defined(SNN_STAMP) || defined(SNN_CLOCK)
// tuning aid (scratch builds only).  -DSNN_CLOCK: shader-clock and 100 MHz wall-clock stamps at the begin and end of
// every block of k_conv_gather (the in-kernel clock under load, tools/clock_conv.py); -DSNN_STAMP additionally the
// per-phase cycle totals of wave 0 of the first 2048 blocks of the pipelined loop (tools/stamp_conv.py; costs ~10 %)
__device__ unsigned long long g_stamps[2048 * 8];
__device__ unsigned long long g_stamps2[2048 * 4];
extern "C" int snn_debug_stamps(unsigned long long* out, int n) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stamps), sizeof(unsigned long long) * n);
}
extern "C" int snn_debug_stamps2(unsigned long long* out, int n) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stamps2), sizeof(unsigned long long) * n);
}
#endif
#ifdef SNN_STAMP
#define STAMP(i) do { unsigned long long t_ = __builtin_readcyclecounter(); st_acc[i] += t_ - st_last; st_last = t_; } while (0)
#else
#define STAMP(i) do {} while (0)
#endif

namespace {

constexpr int kThreads = 256;
constexpr int BM = 128;  // output pixels per block
constexpr int BK = 32;   // k elements per LDS stage
constexpr int LDK = BK + 4;
// 3 waves / SIMD (<= 168 VGPRs): measured +5..+25 % over 2 waves / SIMD with a second LDS stage
#define SNN_CONV_MIN_WAVES 3
#ifndef SNN_GATHER_SB_WAVES
#define SNN_GATHER_SB_WAVES 3   // waves per SIMD the bf16-storage FORWARD instances of the pipelined kernel are compiled for
                                // (same-call A/B: 2 -> 3 waves 88 -> 79 us; 4 spills 13 registers, 80 us); the data-gradient
                                // instances (no statistics) fit 4 waves: 88 -> 81 us
#endif

struct ConvGeom {
    int64_t Mtot;      // GEMM rows: img * OH * OW (FWD) or img * OHc * OWc (DGRAD, one stride-phase class)
    int IH, IW, IC;    // gathered tensor
    int OH, OW, OC;    // produced tensor
    int KH, KW, stride, pad;
    int64_t ldi, ldo;
    int Ktot;          // k extent of this launch: KH*KW*IC (FWD) or nkh*nkw*IC (DGRAD class)
    int KtotFull;      // row length of the weight matrix: KH*KW*IC
    // DGRAD only.  Output pixels (hi, wi) with hi % stride == ph, wi % stride == pw form one class; only the
    // taps kh = kh0 + stride*jh (jh < nkh), kw = kw0 + stride*jw (jw < nkw) reach them, with source pixel
    // iy = (hi + pad - kh0)/stride - jh (exact).  For stride 1 there is a single class with every tap.
    int ph, pw, kh0, kw0, nkh, nkw, OHc, OWc;
    // ceil(2^32 / d) for d = IC and d = (DGRAD ? nkw : KW): q = umulhi(n, magic) == n / d for n * d < 2^32
    unsigned magic_ic, magic_kw;
    int out_vec;  // output (and addend) rows may be stored 16 bytes per lane
    int nimg;     // images in the gathered tensor (FAST loader: extent of its buffer resource)
    int mtiles, mtiles_per_xcd, ntiles;  // XCD-aware tile order (see k_conv_gather)
    // FWD only: statistics partials of the BatchNorm that follows (null: none), see stat_flush below
    double* bn_partial;
    int64_t bn_rows;   // output pixels per timestep (>= BM: a row tile meets at most two timesteps)
    int bn_chunks;     // chunk slots per timestep
    float x_th;        // XSP kernels: the gathered tensor holds saved LIF potentials, the operand is z = (v_dec > x_th)
};

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
// fp16 x 3 ("SPLIT = 4"): operands are pre-scaled by powers of two (exact) so that the LOW pieces stay in fp16's normal
// range: weights x 2^8 (|w| < 255), activations x 2^4 (|x| < 4094; full 22-bit precision down to |x| = 0.008, graceful
// below: absolute error 4e-9).  The accumulators are scaled back by 2^-12 in the epilogue.
constexpr float kF16WeightScale = 256.0f;
constexpr float kF16ActScale = 16.0f;
constexpr float kF16Unscale = 1.0f / (kF16WeightScale * kF16ActScale);
constexpr int LDB = BK + 8;  // bf16 row stride of the split-precision LDS images: 80 B keeps ds_read_b128 conflict-free

// ---- BatchNorm statistics out of a forward epilogue.  The separate pass (snn_bn_stats) re-reads the whole layer
// output from HBM; the epilogue has every value in registers on its way to the store.  Layout of the partials is the
// one snn_bn_stats_finalize reads: partial[t][c][chunk][2] = (sum y, sum y^2) in fp64, a chunk being whatever set of
// pixels of timestep t one block (tile) owns.  The MFMA accumulator layout already is "one channel per lane": lane
// (r, h) of wave (wm, wn) holds channel (wn*TN + j)*32 + r of the 16 rows (wm*TM + i)*32 + (e&3) + 8*(e>>2) + 4*h,
// so a lane sums its own registers, the two half-waves are added by one shuffle and the WM waves through LDS, in
// that fixed order: deterministic, run to run.  (A convolution with statistics takes no addend: the sums are of the
// accumulators, which then are the stored values.)
//
// red: 4 * TN * 32 * 2 doubles of LDS, free to use; dst: the [C][2] slot of this block's chunk.  Block-uniform call.
template <int WM, int WN, int TN>
__device__ __forceinline__ void stat_flush(double (&s)[TN], double (&q)[TN], double* red, double* __restrict__ partial,
                                           int64_t step, int64_t chunk, int64_t chunks, int n0, int OC, int tid) {
    const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        s[j] += __shfl_xor(s[j], 32, 64);
        q[j] += __shfl_xor(q[j], 32, 64);
        if (lane < 32) {
            red[((wave * TN + j) * 32 + lane) * 2 + 0] = s[j];
            red[((wave * TN + j) * 32 + lane) * 2 + 1] = q[j];
        }
    }
    __syncthreads();
    if (tid < WN * TN * 32) {
        const int wn = tid / (TN * 32), jr = tid % (TN * 32);
        double ss = 0.0, qq = 0.0;
#pragma unroll
        for (int wm = 0; wm < WM; ++wm) {
            const double* src = red + (((wm * WN + wn) * TN) * 32 + jr) * 2;
            ss += src[0];
            qq += src[1];
        }
        if (n0 + tid < OC) {
            double* dst = partial + snn_bn_partial_index(step, chunk, n0 + tid, chunks, OC);
            dst[0] = ss;
            dst[1] = qq;
        }
    }
    __syncthreads();
}

static unsigned magic_u32(int d) { return d <= 1 ? 0u : (unsigned)((0x100000000ULL + (unsigned)d - 1) / (unsigned)d); }
__device__ __forceinline__ int div_magic(int n, int d, unsigned magic) { return d == 1 ? n : (int)__umulhi((unsigned)n, magic); }

// SPLIT = 0: exact fp32 MFMA (v_mfma_f32_32x32x2_f32): an fmaf chain, the reference arithmetic.
// SPLIT = 2: "bf16 x 3": every fp32 operand is split on the way into LDS into hi = bf16(x) and
//   lo = bf16(x - hi); the product is hi*hi + hi*lo + lo*hi on v_mfma_f32_32x32x16_bf16 with fp32 accumulation
//   (relative error ~2^-16 per product instead of 2^-24, at 16/3 of the fp32 matrix rate).  Default for the
//   data gradient, where a 1e-5 relative error is far inside the gradient tolerance.
// SPLIT = 4: "fp16 x 3": x = h + l with fp16 pieces (11 + 11 significant bits), products hh + hl + lh on
//   v_mfma_f32_32x32x16_f16: relative error 2^-22, fp32-grade like bf16 x 6 at HALF its matrix work and two LDS
//   images instead of three.  fp16 has the range for forward values (|x| < 65504; activations of a normalised
//   spiking net are O(1), weights are pre-scaled by 2^8), not for gradients - the backward kernels stay on bf16.
// SPLIT = 5: "bf16 x 1": the opt-in THROUGHPUT mode - every operand is rounded once to bf16 (8 significant bits) and
//   multiplied as it is: one product, fp32 accumulation and storage.  Not a parity mode (relative error 2^-9 per
//   product); tolerance stated in tests/test_gpu_bf16_mode.py.
// SPLIT = 3: "bf16 x 6": three-way split x = h + m + l (24 significant bits, i.e. the fp32 value itself) and the
//   six products hh + hm + mh + mm + hl + lh; the dropped terms are 2^-25 relative - fp32-grade accuracy at
//   16/6 of the fp32 matrix rate.
// FAST (host-checked: VEC, IC % 32 == 0, <= 31 taps, 4 images of the gathered tensor < 2 GiB): a k-step of 32 lies
// inside ONE filter tap, so the tap decode is scalar (SALU) and a row's address is "row offset + scalar tap
// offset".  Loads are raw buffer loads relative to the block's first image; padding / out-of-range rows get the
// offset 0xFFFFFFFF and the hardware range check returns zeros - no clamps, no value selects, no 64-bit address
// arithmetic in the loop (the generic loader spends more VALU cycles on addresses than the MFMAs take).
// PRESPLIT (FAST, SPLIT 2 or 4 only): `wk` is not the fp32 weight matrix but its pre-split image (snn_weight_presplit:
// per 4 consecutive k, 4 hi pieces then 4 lo pieces - the same 16 bytes at the same offsets), written once per optimiser
// step; the loader is unchanged and the per-block conversion of the weight tile (half of the conversion VALU of a
// k-step, repeated by every one of the ~1 400 blocks of a launch) disappears.  Same bits as converting on the fly.
// SB (FAST, SPLIT 5 only; SNN_PREC_BF16S, the bf16-STORAGE throughput mode): `in`, `out` and the addends are bf16 tensors
// (strides in elements).  The gathered rows arrive as 8-byte loads and go to LDS as they are - no conversion; the
// epilogue rounds the fp32 accumulators to bf16 on their way out.  Weights stay fp32 and are rounded in the loader.
// XSP (FAST, forward, SPLIT 4 only; snn_conv1x1_spikes_fwd): `in` holds the pre-reset potentials v_dec a LIF layer saved for
// its backward pass, NOT its output - that layer wrote no spike tensor at all (SNN_SCAN_SPIKES_FROM_VDEC) and the operand
// is formed here, z = (v_dec > x_th), on the way into LDS.  A spike is exact in ONE fp16 piece (16.0 or 0 after the 2^4
// pre-scale): no low image is written or read and the product low(x) * high(w) - identically zero - is not issued: two
// MFMA products per multiply-add, same bits as the three-product kernel fed the stored spikes.
template <int BN, int WM, int WN, bool DGRAD, bool VEC, int SPLIT, bool FAST, bool PRESPLIT = false, bool SB = false,
          bool XSP = false>
__global__ __launch_bounds__(kThreads, (FAST && SPLIT) ? (PRESPLIT ? 3 : (SB ? (DGRAD ? 4 : SNN_GATHER_SB_WAVES) : 2)) : SNN_CONV_MIN_WAVES) void k_conv_gather(const float* __restrict__ in, const float* __restrict__ wk,
                                                          float* __restrict__ out, ConvGeom g,
                                                          const float* __restrict__ addend, int64_t ld_add,
                                                          const float* __restrict__ addend2, int64_t ld_add2) {
    constexpr int TM = BM / WM / 32;
    constexpr int TN = BN / WN / 32;
    constexpr int BROWS = BN / 32;  // B rows loaded per thread
    static_assert(WM * WN == 4, "4 waves");
    static_assert(!SB || (FAST && SPLIT == 5 && !PRESPLIT), "bf16 storage: the pipelined one-product kernel");
    static_assert(!XSP || (FAST && SPLIT == 4 && !DGRAD && !PRESPLIT && !SB), "spikes from potentials: fp16 x 3 forward");
    constexpr int ES = SB ? 2 : 4;   // bytes per activation element in HBM
    constexpr int NPIECE = SPLIT == 3 ? 3 : (SPLIT == 5 ? 1 : 2);  // 16-bit images per operand
    constexpr int A_BYTES = SPLIT ? NPIECE * BM * LDB * 2 : BM * LDK * 4;
    constexpr int B_BYTES = SPLIT ? NPIECE * BN * LDB * 2 : BN * LDK * 4;
    constexpr int STAGE_BYTES = 4 * 32 * (TN * 32 + 4) * 4;   // epilogue: 32 staged rows per wave (see below)
    constexpr int SMEM_BYTES = A_BYTES + B_BYTES > STAGE_BYTES ? A_BYTES + B_BYTES : STAGE_BYTES;
    __shared__ __attribute__((aligned(16))) unsigned char smem[SMEM_BYTES];
    float* As = reinterpret_cast<float*>(smem);
    float* Bs = reinterpret_cast<float*>(smem + A_BYTES);
    __bf16* Ah = reinterpret_cast<__bf16*>(smem);                 // [BM][LDB] high parts
    __bf16* Al = Ah + BM * LDB;                                   // [BM][LDB] low parts
    __bf16* Am = Al + BM * LDB;                                   // [BM][LDB] middle parts (SPLIT == 3)
    __bf16* Bh = reinterpret_cast<__bf16*>(smem + A_BYTES);
    __bf16* Bl = Bh + BN * LDB;
    __bf16* Bm = Bl + BN * LDB;

    const int tid = threadIdx.x;
#if defined(SNN_STAMP) || defined(SNN_CLOCK)
    const unsigned long long st_kernel_begin = __builtin_amdgcn_s_memtime();
    const unsigned long long st_real_begin = __builtin_amdgcn_s_memrealtime();
#endif
    const int lane_id = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int r = lane_id & 31, h = lane_id >> 5;

    // XCD-aware tile order.  Workgroups are dealt round-robin to the 8 XCDs (each with its own 4 MiB L2): ids
    // L, L+8, L+16, ... run on one XCD.  XCD x gets the x-th CONTIGUOUS eighth of the pixel tiles (and all channel
    // tiles of a pixel tile back to back), so the blocks resident together on an XCD cover neighbouring image rows
    // and the 3x3 taps that reach into the rows above / below hit that XCD's L2.  With the plain order every XCD
    // held scattered 128-pixel segments and re-fetched the neighbouring rows from HBM (PMC: the 32-channel 3x3
    // data gradient read its input 5x).
    const int bid_xcd = blockIdx.x & 7, bid_q = blockIdx.x >> 3;
    const int bid_n = bid_q % g.ntiles;
    const int64_t bid_m = (int64_t)bid_xcd * g.mtiles_per_xcd + bid_q / g.ntiles;
    if (bid_m >= g.mtiles) return;  // padding block of the last XCD share (whole block, before any barrier)
    const int64_t m0 = bid_m * BM;
    const int n0 = bid_n * BN;

    // ---- per-thread loader geometry: rows lr + 32*j, k offset kq
    // Rows are permuted so that the two rows written by one 16-lane LDS store group lie 4 rows (320 B) apart: with
    // the 80-byte row pitch adjacent rows would overlap by 4 banks (measured: a third of all LDS cycles were
    // bank conflicts); 16 dwords apart modulo 32 banks they tile the banks exactly.
    const int lrr = tid >> 3;
    const int lr = ((lrr >> 1) & 3) + 4 * (lrr & 1) + 8 * (lrr >> 3), kq = (tid & 7) * 4;
    int a_y0[4], a_x0[4];
    int a_base[4];  // first pixel of the image (the host checks img * IH * IW < 2^31)
    bool a_ok[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        // 32-bit arithmetic (the host checks Mtot < 2^31): a 64-bit division costs ~10x a 32-bit one, and the
        // 13 of them per thread made the prologue 14 % of a block's lifetime (measured with s_memtime stamps)
        const unsigned m = (unsigned)m0 + lr + 32 * j;
        a_ok[j] = m < (unsigned)g.Mtot;
        const unsigned mm = a_ok[j] ? m : 0u;
        const unsigned ow_ = DGRAD ? g.OWc : g.OW, oh_ = DGRAD ? g.OHc : g.OH;
        const unsigned t = mm / ow_;
        const int ox = (int)(mm - t * ow_);
        const unsigned img = t / oh_;
        const int oy = (int)(t - img * oh_);
        a_base[j] = (int)(img * (unsigned)(g.IH * g.IW));
        if (!DGRAD) {
            a_y0[j] = oy * g.stride - g.pad;
            a_x0[j] = ox * g.stride - g.pad;
        } else if (g.stride == 1) {
            a_y0[j] = oy + g.pad;  // stride 1: ph = pw = kh0 = kw0 = 0
            a_x0[j] = ox + g.pad;
        } else {
            a_y0[j] = (int)((unsigned)(oy * g.stride + g.ph + g.pad - g.kh0) / (unsigned)g.stride);
            a_x0[j] = (int)((unsigned)(ox * g.stride + g.pw + g.pad - g.kw0) / (unsigned)g.stride);
        }
    }

    // k index -> (source pixel offset, weight column)
    auto decode_k = [&](int kk, int& dy, int& dx, int& c, int& wcol) {
        int tap = div_magic(kk, g.IC, g.magic_ic);
        c = kk - tap * g.IC;
        if (!DGRAD) {
            int kh = div_magic(tap, g.KW, g.magic_kw), kw = tap - kh * g.KW;
            dy = kh;
            dx = kw;
            wcol = kk;
        } else {
            int jh = div_magic(tap, g.nkw, g.magic_kw), jw = tap - jh * g.nkw;
            dy = -jh;
            dx = -jw;
            wcol = ((g.kh0 + g.stride * jh) * g.KW + (g.kw0 + g.stride * jw)) * g.IC + c;
        }
    };

    // A rows on their way to LDS: 4 fp32 values, or (SB) 4 bf16 values as two dwords.  (Integer-typed on purpose: carried
    // in float lanes and bit-cast back element by element, hipcc 7.2 narrows the 8-byte buffer load to 4 bytes.)
    using AReg = typename std::conditional<SB, u32x2, f32x4>::type;
    AReg ra[4];
    f32x4 rb[BROWS];

    // ---- FAST loader state
    __amdgpu_buffer_rsrc_t rs_a, rs_b;
    int a_rel[4];            // byte offset of (row pixel origin, channel kq) from the block's first image
    unsigned a_mask[4];      // bit t: tap t of this row reads inside the image
    unsigned b_rel[BROWS];   // byte offset of (weight row, column kq); >= 2^31 for rows past OC
    if (FAST) {
        const int ow_ = DGRAD ? g.OWc : g.OW, oh_ = DGRAD ? g.OHc : g.OH;
        const int64_t img0 = (unsigned)m0 / (unsigned)(oh_ * ow_);
        const int64_t ipix = (int64_t)g.IH * g.IW;
        const int64_t bytes = ((((int64_t)g.nimg - img0) * ipix - 1) * g.ldi + g.IC) * ES;
        rs_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(reinterpret_cast<const char*>(in) + img0 * ipix * g.ldi * ES), 0,
                                                 bytes > 0xffffffffLL ? (int)0xffffffffu : (int)(unsigned)bytes,
                                                 0x00020000);
        rs_b = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(wk), 0, g.OC * g.KtotFull * 4, 0x00020000);
        const int tw_n = DGRAD ? g.nkw : g.KW;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int relpix = (a_base[j] - (int)(img0 * ipix)) + a_y0[j] * g.IW + a_x0[j];
            a_rel[j] = (relpix * (int)g.ldi + kq) * ES;
            // bit (th * ntw + tw) = tap inside the image: row validity x column validity, branch-free (the taps of a
            // FAST launch are at most 5 x 5: ntaps <= 31)
            const int nth = DGRAD ? g.nkh : g.KH;
            unsigned xm = 0, mask = 0;
#pragma unroll
            for (int tw = 0; tw < 6; ++tw) {
                const int ix = DGRAD ? a_x0[j] - tw : a_x0[j] + tw;
                xm |= (tw < tw_n && (unsigned)ix < (unsigned)g.IW) ? 1u << tw : 0u;
            }
#pragma unroll
            for (int th = 0; th < 6; ++th) {
                const int iy = DGRAD ? a_y0[j] - th : a_y0[j] + th;
                mask |= (th < nth && (unsigned)iy < (unsigned)g.IH) ? xm << (th * tw_n) : 0u;
            }
            a_mask[j] = a_ok[j] ? mask : 0u;
        }
#pragma unroll
        for (int j = 0; j < BROWS; ++j) {
            const int n = n0 + lr + 32 * j;
            b_rel[j] = n < g.OC ? (unsigned)(n * g.KtotFull + kq) * 4u : 0x80000000u;
        }
    }
    const unsigned fast_tw_n = DGRAD ? g.nkw : g.KW;
    const unsigned fast_tw_one = fast_tw_n == 1 ? 1u : 0u;  // magic_u32(1) is 0: q = umulhi(n, 0) + n
    // which = 1: the A rows, 2: the B rows, 3: both
    auto load_tiles_fast = [&](int k0n, AReg (&ra)[4], f32x4 (&rb)[BROWS], int which = 3) {  // k0n is block-uniform: everything up to the per-row adds is scalar
        const bool kin = k0n < g.Ktot;
        const int tap = (int)__umulhi((unsigned)k0n, g.magic_ic);  // IC >= 32 here
        const int c0 = k0n - tap * g.IC;
        const int th = (int)(__umulhi((unsigned)tap, g.magic_kw) + (unsigned)tap * fast_tw_one);
        const int tw = tap - th * (int)fast_tw_n;
        int toff, wcol0;
        if (!DGRAD) {
            toff = ((th * g.IW + tw) * (int)g.ldi + c0) * ES;
            wcol0 = k0n;
        } else {
            toff = (c0 - (th * g.IW + tw) * (int)g.ldi) * ES;
            wcol0 = ((g.kh0 + g.stride * th) * g.KW + (g.kw0 + g.stride * tw)) * g.IC + c0;
        }
        const int tbit = kin ? tap : 31;  // bit 31 is never set: a prefetch past the last k-step loads zeros
        if (which & 1) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int voff = ((a_mask[j] >> tbit) & 1u) ? a_rel[j] + toff : -1;
                if constexpr (SB) ra[j] = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(rs_a, voff, 0, 0));
                else ra[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_a, voff, 0, 0));
            }
        }
        if (which & 2) {
            const unsigned wb = (unsigned)wcol0 * 4u;
#pragma unroll
            for (int j = 0; j < BROWS; ++j)
                rb[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_b, (int)(b_rel[j] + wb), 0, 0));
        }
    };

    auto load_tiles = [&](int k0) {
        if (FAST) {
            load_tiles_fast(k0, ra, rb);
            return;
        }
        if constexpr (!SB) {   // (the generic loaders hold fp32 rows; SB kernels are FAST by construction)
        const int kk = k0 + kq;
        if (VEC) {
            // Branch-free: every lane always loads from a clamped (valid) address and masks the value afterwards,
            // so the whole k-step stays one basic block and the scheduler can interleave these loads with MFMAs.
            const bool kin = kk < g.Ktot;
            int dy, dx, c, wcol;
            decode_k(kin ? kk : g.Ktot - 4, dy, dx, c, wcol);
            const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int iy = a_y0[j] + dy, ix = a_x0[j] + dx;
                const bool ok = kin & a_ok[j] & ((unsigned)iy < (unsigned)g.IH) & ((unsigned)ix < (unsigned)g.IW);
                const int iyc = min(max(iy, 0), g.IH - 1), ixc = min(max(ix, 0), g.IW - 1);
                f32x4 v = *reinterpret_cast<const f32x4*>(in + (int64_t)(a_base[j] + iyc * g.IW + ixc) * g.ldi + c);
                ra[j] = ok ? v : zero;
            }
#pragma unroll
            for (int j = 0; j < BROWS; ++j) {
                const int n = n0 + lr + 32 * j;
                const int nc = min(n, g.OC - 1);
                f32x4 v = *reinterpret_cast<const f32x4*>(wk + (int64_t)nc * g.KtotFull + wcol);
                rb[j] = (kin & (n < g.OC)) ? v : zero;
            }
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int ke = kk + e;
                const bool kin = ke < g.Ktot;
                int dy, dx, c, wcol;
                decode_k(kin ? ke : 0, dy, dx, c, wcol);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int iy = a_y0[j] + dy, ix = a_x0[j] + dx;
                    float v = 0.f;
                    if (kin && a_ok[j] && (unsigned)iy < (unsigned)g.IH && (unsigned)ix < (unsigned)g.IW)
                        v = in[(int64_t)(a_base[j] + iy * g.IW + ix) * g.ldi + c];
                    ra[j][e] = v;
                }
#pragma unroll
                for (int j = 0; j < BROWS; ++j) {
                    int n = n0 + lr + 32 * j;
                    rb[j][e] = (kin && n < g.OC) ? wk[(int64_t)n * g.KtotFull + wcol] : 0.f;
                }
            }
        }
        }
    };
    auto split_store = [&](const f32x4& v, __bf16* hi_img, __bf16* mid_img, __bf16* lo_img, int row) {
        bf16x4 hi, mid, lo;
#pragma unroll
        for (int e = 0; e < 4; e += 2) {  // two elements per v_cvt_pk_bf16_f32; widening back is a shift / mask
            f32x2 rest = {v[e], v[e + 1]};
            bf16x2 p = __builtin_convertvector(rest, bf16x2);
            unsigned bits = __builtin_bit_cast(unsigned, p);
            hi[e] = p[0]; hi[e + 1] = p[1];
            rest[0] -= __builtin_bit_cast(float, bits << 16);
            rest[1] -= __builtin_bit_cast(float, bits & 0xffff0000u);
            if (SPLIT == 3) {
                p = __builtin_convertvector(rest, bf16x2);
                bits = __builtin_bit_cast(unsigned, p);
                mid[e] = p[0]; mid[e + 1] = p[1];
                rest[0] -= __builtin_bit_cast(float, bits << 16);
                rest[1] -= __builtin_bit_cast(float, bits & 0xffff0000u);
            }
            p = __builtin_convertvector(rest, bf16x2);
            lo[e] = p[0]; lo[e + 1] = p[1];
        }
        *reinterpret_cast<bf16x4*>(&hi_img[row * LDB + kq]) = hi;
        if (SPLIT == 3) *reinterpret_cast<bf16x4*>(&mid_img[row * LDB + kq]) = mid;
        *reinterpret_cast<bf16x4*>(&lo_img[row * LDB + kq]) = lo;
    };
    auto store_tiles = [&]() {
        if constexpr (SB) return;
        else if (SPLIT) {
#pragma unroll
            for (int j = 0; j < 4; ++j) split_store(ra[j], Ah, Am, Al, lr + 32 * j);
#pragma unroll
            for (int j = 0; j < BROWS; ++j) split_store(rb[j], Bh, Bm, Bl, lr + 32 * j);
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) *reinterpret_cast<f32x4*>(&As[(lr + 32 * j) * LDK + kq]) = ra[j];
#pragma unroll
            for (int j = 0; j < BROWS; ++j) *reinterpret_cast<f32x4*>(&Bs[(lr + 32 * j) * LDK + kq]) = rb[j];
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    if constexpr (FAST && SPLIT != 0) {
        // ---- software-pipelined main loop (2 waves / SIMD).  While the MFMAs of tile k run from LDS, the SAME wave
        // converts tile k+1 (raw fp32 in registers since the previous k-step) into its bf16 pieces in the MFMA
        // shadow - about 4 VALU per MFMA gap, which the matrix pipe hides - and then issues the loads of tile
        // k+2.  Between the two barriers only the LDS writes remain.  Measured without this (convert + write
        // between the barriers): MFMA pipe busy 36 % even with the global loads removed.
        constexpr int NP = NPIECE;                      // 16-bit images per operand
        constexpr int NPROD = SPLIT == 3 ? 6 : (SPLIT == 5 ? 1 : 3);   // MFMA products per accumulator and k16
        bf16x4 pa[4][NP], pb[BROWS][NP];                // [.][0] hi, [.][1] lo, [.][2] mid
        auto convert = [&](const f32x4& v, bf16x4* out, float scale) {
            if constexpr (SPLIT == 5) {  // one bf16 piece: round to nearest even
#pragma unroll
                for (int e = 0; e < 4; e += 2) {
                    const bf16x2 p = __builtin_convertvector(f32x2{v[e], v[e + 1]}, bf16x2);
                    out[0][e] = p[0]; out[0][e + 1] = p[1];
                }
                return;
            }
            if constexpr (SPLIT == 4) {  // fp16 pieces (v_cvt_pk_f16_f32); the residual x - hi is exact in fp32
                u32x2 hi, lo;
#pragma unroll
                for (int e = 0; e < 4; e += 2) {
                    const float a = v[e] * scale, b = v[e + 1] * scale;
                    const f16x2 ph = __builtin_convertvector(f32x2{a, b}, f16x2);  // RNE: out of range -> inf (loud)
                    const f16x2 pl = __builtin_convertvector(f32x2{a - (float)ph[0], b - (float)ph[1]}, f16x2);
                    hi[e >> 1] = __builtin_bit_cast(unsigned, ph);
                    lo[e >> 1] = __builtin_bit_cast(unsigned, pl);
                }
                out[0] = __builtin_bit_cast(bf16x4, hi);
                out[1] = __builtin_bit_cast(bf16x4, lo);
                return;
            }
#pragma unroll
            for (int e = 0; e < 4; e += 2) {
                f32x2 rest = {v[e], v[e + 1]};
                bf16x2 p = __builtin_convertvector(rest, bf16x2);
                unsigned bits = __builtin_bit_cast(unsigned, p);
                out[0][e] = p[0]; out[0][e + 1] = p[1];
                rest[0] -= __builtin_bit_cast(float, bits << 16);
                rest[1] -= __builtin_bit_cast(float, bits & 0xffff0000u);
                if (SPLIT == 3) {
                    p = __builtin_convertvector(rest, bf16x2);
                    bits = __builtin_bit_cast(unsigned, p);
                    out[2][e] = p[0]; out[2][e + 1] = p[1];
                    rest[0] -= __builtin_bit_cast(float, bits << 16);
                    rest[1] -= __builtin_bit_cast(float, bits & 0xffff0000u);
                }
                p = __builtin_convertvector(rest, bf16x2);
                out[1][e] = p[0]; out[1][e + 1] = p[1];
            }
        };
        auto convert_a = [&](const AReg& v, bf16x4* out) {
            if constexpr (SB) out[0] = __builtin_bit_cast(bf16x4, v);   // already the bf16 values
            else if constexpr (XSP) {   // z = (v_dec > th) as ONE fp16 piece of z * 2^4: 0x4C00 (16.0) or 0
                u32x2 hi;
#pragma unroll
                for (int e = 0; e < 4; e += 2)
                    hi[e >> 1] = (v[e] > g.x_th ? 0x4C00u : 0u) | (v[e + 1] > g.x_th ? 0x4C000000u : 0u);
                out[0] = __builtin_bit_cast(bf16x4, hi);
            } else convert(v, out, kF16ActScale);
        };
        auto convert_b = [&](const f32x4& v, bf16x4* out) {
            if constexpr (PRESPLIT) {   // the 16 bytes already are (4 hi, 4 lo)
                static_assert(!PRESPLIT || SPLIT == 2 || SPLIT == 4, "pre-split weights: two-piece modes only");
                out[0] = __builtin_bit_cast(bf16x4, f32x2{v[0], v[1]});
                out[1] = __builtin_bit_cast(bf16x4, f32x2{v[2], v[3]});
            } else {
                convert(v, out, kF16WeightScale);
            }
        };
        auto write_tiles = [&]() {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int o = (lr + 32 * j) * LDB + kq;
                *reinterpret_cast<bf16x4*>(&Ah[o]) = pa[j][0];
                if constexpr (NP >= 2 && !XSP) *reinterpret_cast<bf16x4*>(&Al[o]) = pa[j][1];
                if constexpr (NP >= 3) *reinterpret_cast<bf16x4*>(&Am[o]) = pa[j][2];
            }
#pragma unroll
            for (int j = 0; j < BROWS; ++j) {
                const int o = (lr + 32 * j) * LDB + kq;
                *reinterpret_cast<bf16x4*>(&Bh[o]) = pb[j][0];
                if constexpr (NP >= 2) *reinterpret_cast<bf16x4*>(&Bl[o]) = pb[j][1];
                if constexpr (NP >= 3) *reinterpret_cast<bf16x4*>(&Bm[o]) = pb[j][2];
            }
        };
        auto mfma_group = [&](int ks) {  // lane (r, h) holds k = 16*ks + 8*h .. +7 of its row
            bf16x8 ah[TM], am[TM], al[TM], bh[TN], bm[TN], bl[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int off = ((wm * TM + i) * 32 + r) * LDB + ks * 16 + 8 * h;
                ah[i] = *reinterpret_cast<const bf16x8*>(&Ah[off]);
                if constexpr (NP >= 2 && !XSP) al[i] = *reinterpret_cast<const bf16x8*>(&Al[off]);
                if constexpr (NP >= 3) am[i] = *reinterpret_cast<const bf16x8*>(&Am[off]);
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int off = ((wn * TN + j) * 32 + r) * LDB + ks * 16 + 8 * h;
                bh[j] = *reinterpret_cast<const bf16x8*>(&Bh[off]);
                if constexpr (NP >= 2) bl[j] = *reinterpret_cast<const bf16x8*>(&Bl[off]);
                if constexpr (NP >= 3) bm[j] = *reinterpret_cast<const bf16x8*>(&Bm[off]);
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {  // small terms first
                    if constexpr (SPLIT == 5) {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);
                        continue;
                    }
                    if constexpr (SPLIT == 4 && XSP) {   // the low image of a spike is zero: two products
                        const f16x8 xah = __builtin_bit_cast(f16x8, ah[i]);
                        const f16x8 xbh = __builtin_bit_cast(f16x8, bh[j]), xbl = __builtin_bit_cast(f16x8, bl[j]);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(xah, xbl, acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(xah, xbh, acc[i][j], 0, 0, 0);
                        continue;
                    }
                    if constexpr (SPLIT == 4) {
                        const f16x8 xah = __builtin_bit_cast(f16x8, ah[i]), xal = __builtin_bit_cast(f16x8, al[i]);
                        const f16x8 xbh = __builtin_bit_cast(f16x8, bh[j]), xbl = __builtin_bit_cast(f16x8, bl[j]);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(xal, xbh, acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(xah, xbl, acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(xah, xbh, acc[i][j], 0, 0, 0);
                        continue;
                    }
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh[j], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl[j], acc[i][j], 0, 0, 0);
                    if (SPLIT == 3) {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am[i], bm[j], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am[i], bh[j], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bm[j], acc[i][j], 0, 0, 0);
                    }
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);
                }
        };
        constexpr int NM = TM * TN * (XSP ? 2 : NPROD);  // MFMAs per k16 group
        constexpr int NREAD = XSP ? TM + TN * NP : (TM + TN) * NP;   // ds_read_b128 per k16 group
        constexpr int CONV_OPS = SPLIT == 3 ? 24 : (SPLIT == 5 ? 2 : 14);   // VALU per converted f32x4 (approx.)
        constexpr int VPG_A = ((XSP ? 4 * 8 : 4 * CONV_OPS) + NM - 1) / NM;
        constexpr int VPG_B = PRESPLIT ? 1 : (BROWS * CONV_OPS + NM - 1) / NM;   // pre-split: only register moves
        if (g.Ktot > 0) {
            // both first tiles are requested back to back (the accumulators are not live yet, registers are free):
            // one exposed memory latency per block instead of two
            AReg ra0[4];
            f32x4 rb0[BROWS];
            load_tiles_fast(0, ra0, rb0);
            load_tiles_fast(BK, ra, rb);
#pragma unroll
            for (int j = 0; j < 4; ++j) convert_a(ra0[j], pa[j]);
#pragma unroll
            for (int j = 0; j < BROWS; ++j) convert_b(rb0[j], pb[j]);
            write_tiles();
        }
        __syncthreads();
#ifdef SNN_STAMP
        unsigned long long st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        unsigned long long st_last = __builtin_readcyclecounter();
        const unsigned long long st_begin = st_last;
#endif
#pragma unroll 1
        for (int k0 = 0; k0 < g.Ktot; k0 += BK) {
            mfma_group(0);
#pragma unroll
            for (int j = 0; j < 4; ++j) convert_a(ra[j], pa[j]);
            // shape the schedule: operand reads, then every MFMA followed by its share of the conversion VALU
            __builtin_amdgcn_sched_group_barrier(0x100, NREAD, 0);
#pragma unroll
            for (int m = 0; m < NM; ++m) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, VPG_A, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            STAMP(0);
            mfma_group(1);
#pragma unroll
            for (int j = 0; j < BROWS; ++j) convert_b(rb[j], pb[j]);
            __builtin_amdgcn_sched_group_barrier(0x100, NREAD, 0);
#pragma unroll
            for (int m = 0; m < NM; ++m) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, VPG_B, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            STAMP(1);
            load_tiles_fast(k0 + 2 * BK, ra, rb);   // (requesting the A rows one MFMA group earlier: measured neutral)
            STAMP(2);
            __syncthreads();
            STAMP(3);
            write_tiles();
            STAMP(4);
            __syncthreads();
            STAMP(5);
        }
#ifdef SNN_STAMP
        if (tid == 0 && bid_n == 0 && blockIdx.x < 2048) {
            st_acc[6] = __builtin_readcyclecounter() - st_begin;
            st_acc[7] = st_begin;
            for (int i = 0; i < 8; ++i) g_stamps[blockIdx.x * 8 + i] = st_acc[i];
        }
#endif
    } else {
        if (g.Ktot > 0) {  // a dgrad stride-phase class may have no tap at all: its pixels are plain zeros
            load_tiles(0);
            store_tiles();
        }
        __syncthreads();

        // Branch-free steady state (a tile past Ktot loads zeros and is never read): keeping the MFMA chain in
        // one basic block lets the accumulators stay in their registers across iterations.
    #pragma unroll 1
        for (int k0 = 0; k0 < g.Ktot; k0 += BK) {
            load_tiles(k0 + BK);
            // keep the prefetch ahead of the MFMA chain: its latency must be covered by the whole k-step
            __builtin_amdgcn_sched_barrier(0);
            if (SPLIT) {
    #pragma unroll
                for (int ks = 0; ks < BK / 16; ++ks) {  // lane (r, h) holds k = 16*ks + 8*h .. +7 of its row
                    bf16x8 ah[TM], am[TM], al[TM], bh[TN], bm[TN], bl[TN];
    #pragma unroll
                    for (int i = 0; i < TM; ++i) {
                        const int off = ((wm * TM + i) * 32 + r) * LDB + ks * 16 + 8 * h;
                        ah[i] = *reinterpret_cast<const bf16x8*>(&Ah[off]);
                        al[i] = *reinterpret_cast<const bf16x8*>(&Al[off]);
                        if (SPLIT == 3) am[i] = *reinterpret_cast<const bf16x8*>(&Am[off]);
                    }
    #pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        const int off = ((wn * TN + j) * 32 + r) * LDB + ks * 16 + 8 * h;
                        bh[j] = *reinterpret_cast<const bf16x8*>(&Bh[off]);
                        bl[j] = *reinterpret_cast<const bf16x8*>(&Bl[off]);
                        if (SPLIT == 3) bm[j] = *reinterpret_cast<const bf16x8*>(&Bm[off]);
                    }
    #pragma unroll
                    for (int i = 0; i < TM; ++i)
    #pragma unroll
                        for (int j = 0; j < TN; ++j) {  // small terms first
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh[j], acc[i][j], 0, 0, 0);
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl[j], acc[i][j], 0, 0, 0);
                            if (SPLIT == 3) {
                                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am[i], bm[j], acc[i][j], 0, 0, 0);
                                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am[i], bh[j], acc[i][j], 0, 0, 0);
                                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bm[j], acc[i][j], 0, 0, 0);
                            }
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);
                        }
                }
            }
            const float* Ac = As;
            const float* Bc = Bs;
    #pragma unroll
            for (int ks = 0; ks < (SPLIT ? 0 : BK / 8); ++ks) {
                f32x4 a[TM], b[TN];
    #pragma unroll
                for (int i = 0; i < TM; ++i)
                    a[i] = *reinterpret_cast<const f32x4*>(&Ac[((wm * TM + i) * 32 + r) * LDK + ks * 8 + 4 * h]);
    #pragma unroll
                for (int j = 0; j < TN; ++j)
                    b[j] = *reinterpret_cast<const f32x4*>(&Bc[((wn * TN + j) * 32 + r) * LDK + ks * 8 + 4 * h]);
    #pragma unroll
                for (int e = 0; e < 4; ++e)
    #pragma unroll
                    for (int i = 0; i < TM; ++i)
    #pragma unroll
                        for (int j = 0; j < TN; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][e], b[j][e], acc[i][j], 0, 0, 0);
            }
            __syncthreads();
            store_tiles();
            __syncthreads();
        }
    }

    // ---- epilogue.  C/D layout of the 32x32 MFMA: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5), i.e. a
    // lane holds ONE channel of 16 pixels.  Each wave transposes its accumulators through LDS (the operand tiles
    // are dead by now) so that a lane stores 16 contiguous bytes and a pixel row goes out as TN*128-byte runs:
    // 4x fewer, wider store instructions (the k-short 1x1 convolutions are store-issue bound otherwise).
    constexpr int EW = TN * 32 + 4;   // staged row length in floats
    constexpr int LPR = TN * 8;       // lanes per staged row (4 floats each)
    constexpr int RPP = 64 / LPR;     // rows per pass
    static_assert(4 * 32 * EW * 4 <= SMEM_BYTES, "epilogue staging does not fit the operand tiles");
    float* stage = reinterpret_cast<float*>(smem) + wave * 32 * EW;
    const bool ovec = g.out_vec != 0;
    if (!DGRAD && g.bn_partial != nullptr) {
        // BatchNorm partials: rows below `split` belong to timestep bn_t, the rest (up to Mtot) to bn_t + 1
        const int64_t bn_t = m0 / g.bn_rows;
        const int64_t split = (bn_t + 1) * g.bn_rows;
        const bool whole = split >= m0 + BM && m0 + BM <= g.Mtot;   // one timestep, no rows past the end
        double s_lo[TN], q_lo[TN], s_hi[TN], q_hi[TN];
#pragma unroll
        for (int j = 0; j < TN; ++j) s_lo[j] = q_lo[j] = s_hi[j] = q_hi[j] = 0.0;
        if (whole) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const double d = (double)(SPLIT == 4 ? acc[i][j][e] * kF16Unscale : acc[i][j][e]);
                        s_lo[j] += d;
                        q_lo[j] = fma(d, d, q_lo[j]);
                    }
        } else {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const int64_t m = m0 + (wm * TM + i) * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                        const double d = (double)(SPLIT == 4 ? acc[i][j][e] * kF16Unscale : acc[i][j][e]);
                        const double lo = m < split ? d : 0.0, hi = (m >= split && m < g.Mtot) ? d : 0.0;
                        s_lo[j] += lo;
                        q_lo[j] = fma(lo, lo, q_lo[j]);
                        s_hi[j] += hi;
                        q_hi[j] = fma(hi, hi, q_hi[j]);
                    }
        }
        // the operand tiles are dead (the k loop ended with a barrier); the staging below starts after stat_flush's
        double* red = reinterpret_cast<double*>(smem);
        static_assert(4 * TN * 32 * 2 * 8 <= SMEM_BYTES, "statistics scratch does not fit");
        const int chunk = (int)(m0 / BM - (bn_t * g.bn_rows) / BM);
        stat_flush<WM, WN, TN>(s_lo, q_lo, red, g.bn_partial, bn_t, chunk, g.bn_chunks, n0, g.OC, tid);
        if (split < m0 + BM && split < g.Mtot)   // this tile is also the first one of the next timestep
            stat_flush<WM, WN, TN>(s_hi, q_hi, red, g.bn_partial, bn_t + 1, 0, g.bn_chunks, n0, g.OC, tid);
    }
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e)
                stage[((e & 3) + 8 * (e >> 2) + 4 * h) * EW + j * 32 + r] =
                    SPLIT == 4 ? acc[i][j][e] * kF16Unscale : acc[i][j][e];  // undo the operand pre-scales
        __syncthreads();
        // Address arithmetic: for the common case (output pixel = GEMM row) everything but a per-lane 32-bit offset
        // is wave-uniform: base pointers of the 32-row group live in SGPRs, the lane adds (its row) * ld + channel.
        // (The general form spent ~50 64-bit multiplies per wave here - a third of the epilogue.)
        const bool linear = !(DGRAD && g.stride > 1);
        const int64_t mrow0 = m0 + (wm * TM + i) * 32;   // wave-uniform
        const int lrow = lane_id / LPR;
        const int c4 = (lane_id % LPR) * 4;
        const int n = n0 + wn * TN * 32 + c4;
        typedef SnnStore<SB> St;   // fp32 tensors, or bf16 (rounded here) in the bf16-storage mode
        char* const out_b = reinterpret_cast<char*>(out);
        const char* const ad1_b = reinterpret_cast<const char*>(addend);
        const char* const ad2_b = reinterpret_cast<const char*>(addend2);
        char* out_g = out_b + mrow0 * g.ldo * ES;
        const char* ad1_g = addend ? ad1_b + mrow0 * ld_add * ES : nullptr;
        const char* ad2_g = addend2 ? ad2_b + mrow0 * ld_add2 * ES : nullptr;
        const int o_l = lrow * (int)g.ldo + n, a1_l = lrow * (int)ld_add + n, a2_l = lrow * (int)ld_add2 + n;
#pragma unroll
        for (int pass = 0; pass < 32 / RPP; ++pass) {
            const int row = pass * RPP + lrow;
            const int64_t m = mrow0 + row;
            if (m >= g.Mtot || n >= g.OC) continue;
            f32x4 v = *reinterpret_cast<const f32x4*>(&stage[row * EW + c4]);
            char* dst;
            const char *a1p, *a2p;
            if (linear) {
                dst = out_g + (o_l + pass * RPP * (int)g.ldo) * ES;
                a1p = ad1_g + (a1_l + pass * RPP * (int)ld_add) * ES;
                a2p = ad2_g + (a2_l + pass * RPP * (int)ld_add2) * ES;
            } else {
                const unsigned t = (unsigned)m / (unsigned)g.OWc;
                const int b = (int)((unsigned)m - t * (unsigned)g.OWc);
                const unsigned img = t / (unsigned)g.OHc;
                const int a = (int)(t - img * (unsigned)g.OHc);
                const int64_t pix = ((int64_t)img * g.OH + (a * g.stride + g.ph)) * g.OW + (b * g.stride + g.pw);
                dst = out_b + (pix * g.ldo + n) * ES;
                a1p = ad1_b + (pix * ld_add + n) * ES;
                a2p = ad2_b + (pix * ld_add2 + n) * ES;
            }
            if (ovec && n + 3 < g.OC) {
                if (addend) v += St::ld4_last(a1p, 0);  // fused accumulation (the addend's only reader)
                if (addend2) v += St::ld4_last(a2p, 0);
                St::st4(dst, 0, v);
            } else {
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    if (n + q < g.OC) {
                        float o = v[q];
                        if (addend) o += St::ld1(a1p, q);
                        if (addend2) o += St::ld1(a2p, q);
                        St::st1(dst, q, o);
                    }
            }
        }
        __syncthreads();
    }
#if defined(SNN_STAMP) || defined(SNN_CLOCK)
    if (tid == 0 && bid_n == 0 && blockIdx.x < 2048) {
        g_stamps2[blockIdx.x * 4 + 0] = st_kernel_begin;
        g_stamps2[blockIdx.x * 4 + 1] = __builtin_amdgcn_s_memtime();
        g_stamps2[blockIdx.x * 4 + 2] = st_real_begin;
        g_stamps2[blockIdx.x * 4 + 3] = __builtin_amdgcn_s_memrealtime();
    }
#endif
}

// ------------------------------------------------------------------------------------------ wgrad
constexpr int WB_K = 32;   // pixels per LDS stage

struct WgradGeom {
    int64_t Mtot;  // N * Ho * Wo
    int H, W, Cin, Ho, Wo, Cout, KH, KW, stride, pad;
    int64_t ldx, lddy;
    int Ktot;
    int64_t pix_per_split;
    int tiles_m, tiles_n, splitk;
    int nimg;
    float x_th;    // XSP kernels: x holds saved LIF potentials, the operand is z = (v_dec > x_th)
};

// Block tile (32*TM*WM) out-channels x (32*TN*WN) (tap,ci) columns; each wave owns TM x TN accumulators of
// 32x32.  K = pixels, 32 per LDS stage; the decode pixel -> (image base, y0, x0) of a stage is done by 32
// lanes and published through LDS one stage ahead.  Blocks of one pixel split are mapped to one XCD
// (block ids congruent mod 8) so the dy / x tiles they share are served from that XCD's L2.
template <int TM, int TN, int WM, int WN, bool VEC>
__global__ __launch_bounds__(kThreads, SNN_CONV_MIN_WAVES) void k_conv_wgrad(const float* __restrict__ x, const float* __restrict__ dy,
                                                         float* __restrict__ ws, WgradGeom g) {
    static_assert(WM * WN == 4, "4 waves");
    constexpr int BMc = 32 * TM * WM, BNk = 32 * TN * WN;
    constexpr int DG = BMc / 4, XG = BNk / 4;       // float4 groups per pixel row
    constexpr int DP = kThreads / DG, XP = kThreads / XG;  // pixel rows per pass
    constexpr int DJ = WB_K / DP, XJ = WB_K / XP;   // passes per stage
    static_assert(DJ >= 1 && XJ >= 1, "tile too narrow");
    __shared__ __attribute__((aligned(16))) float Ds[WB_K * BMc];
    __shared__ __attribute__((aligned(16))) float Xs[WB_K * BNk];
    __shared__ int Pinfo[2][WB_K][4];  // {image base pixel, y0, x0, valid}

    const int tid = threadIdx.x;
    const int lane_id = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int r = lane_id & 31, h = lane_id >> 5;

    // ---- block -> (tile, split): blocks L, L+8, L+16, ... (one XCD) walk the tiles of one split
    const int tiles = g.tiles_m * g.tiles_n;
    int L = blockIdx.x, z, tile;
    if (g.splitk % 8 == 0) {
        z = (L % 8) + 8 * (L / (8 * tiles));
        tile = (L / 8) % tiles;
    } else {
        z = L / tiles;
        tile = L % tiles;
    }
    const int co0 = (tile % g.tiles_m) * BMc;
    const int kc0 = (tile / g.tiles_m) * BNk;
    const int64_t p_lo = (int64_t)z * g.pix_per_split;
    int64_t p_hi = p_lo + g.pix_per_split;
    if (p_hi > g.Mtot) p_hi = g.Mtot;

    // ---- loader geometry
    const int d_cq = (tid % DG) * 4, d_pr = tid / DG;
    const int x_cq = (tid % XG) * 4, x_pr = tid / XG;
    int x_kh[4], x_kw[4], x_ci[4];
    bool x_ok[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        int kc = kc0 + x_cq + e;
        x_ok[e] = kc < g.Ktot;
        int kcc = x_ok[e] ? kc : 0;
        int tap = kcc / g.Cin;
        x_ci[e] = kcc - tap * g.Cin;
        x_kh[e] = tap / g.KW;
        x_kw[e] = tap - x_kh[e] * g.KW;
    }
    const bool d_ok = (co0 + d_cq) < g.Cout;

    // Pixel decode (image, oy, ox) of the 32 pixels of a stage: lane t < 32 owns pixel p0 + t, decodes it ONCE
    // with a division and then walks forward 32 pixels per stage with carries (no division in the loop).
    int d_img = 0, d_oy = 0, d_ox = 0;
    int64_t d_p = p_lo + tid;
    if (tid < WB_K) {
        int64_t pp = d_p < g.Mtot ? d_p : 0;
        d_ox = (int)(pp % g.Wo);
        int64_t t = pp / g.Wo;
        d_oy = (int)(t % g.Ho);
        d_img = (int)(t / g.Ho);
    }
    auto decode = [&](int slot) {  // publish the current stage's pixels, then advance to the next stage
        if (tid < WB_K) {
            Pinfo[slot][tid][0] = d_p < p_hi ? d_img * g.H * g.W : 0;  // invalid pixels read (and discard) image 0
            Pinfo[slot][tid][1] = d_oy * g.stride - g.pad;
            Pinfo[slot][tid][2] = d_ox * g.stride - g.pad;
            Pinfo[slot][tid][3] = d_p < p_hi ? 1 : 0;
            d_p += WB_K;
            d_ox += WB_K;
            while (d_ox >= g.Wo) {
                d_ox -= g.Wo;
                if (++d_oy == g.Ho) {
                    d_oy = 0;
                    ++d_img;
                }
            }
        }
    };

    f32x4 rd[DJ], rx[XJ];
    auto load_tiles = [&](int64_t p0, int slot) {
        const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < DJ; ++j) {
            const int row = d_pr + DP * j;
            const int64_t p = p0 + row;
            const bool ok = p < p_hi;
            const int64_t pc = ok ? p : 0;  // clamped address: always load, mask afterwards (no branch)
            f32x4 v = zero;
            if (VEC) {
                v = *reinterpret_cast<const f32x4*>(dy + pc * g.lddy + (d_ok ? co0 + d_cq : 0));
                v = (ok & d_ok) ? v : zero;
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (ok && co0 + d_cq + e < g.Cout) v[e] = dy[p * g.lddy + co0 + d_cq + e];
            }
            rd[j] = v;
        }
#pragma unroll
        for (int j = 0; j < XJ; ++j) {
            const int row = x_pr + XP * j;
            const int ibase = Pinfo[slot][row][0], y0 = Pinfo[slot][row][1], x0 = Pinfo[slot][row][2];
            const bool pok = Pinfo[slot][row][3] != 0;
            f32x4 v = zero;
            if (VEC) {
                const int iy = y0 + x_kh[0], ix = x0 + x_kw[0];
                const bool ok = pok & x_ok[0] & ((unsigned)iy < (unsigned)g.H) & ((unsigned)ix < (unsigned)g.W);
                const int iyc = min(max(iy, 0), g.H - 1), ixc = min(max(ix, 0), g.W - 1);
                v = *reinterpret_cast<const f32x4*>(x + (int64_t)(ibase + iyc * g.W + ixc) * g.ldx + x_ci[0]);
                v = ok ? v : zero;
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int iy = y0 + x_kh[e], ix = x0 + x_kw[e];
                    if (pok && x_ok[e] && (unsigned)iy < (unsigned)g.H && (unsigned)ix < (unsigned)g.W)
                        v[e] = x[(int64_t)(ibase + iy * g.W + ix) * g.ldx + x_ci[e]];
                }
            }
            rx[j] = v;
        }
    };
    auto store_tiles = [&]() {
#pragma unroll
        for (int j = 0; j < DJ; ++j) *reinterpret_cast<f32x4*>(&Ds[(d_pr + DP * j) * BMc + d_cq]) = rd[j];
#pragma unroll
        for (int j = 0; j < XJ; ++j) *reinterpret_cast<f32x4*>(&Xs[(x_pr + XP * j) * BNk + x_cq]) = rx[j];
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    decode(0);
    __syncthreads();
    load_tiles(p_lo, 0);
    decode(1);
    store_tiles();
    __syncthreads();

    int slot = 1;
#pragma unroll 1
    for (int64_t p0 = p_lo; p0 < p_hi; p0 += WB_K) {
        load_tiles(p0 + WB_K, slot);  // rows past p_hi load zeros
        decode(slot ^ 1);             // pixels of stage p0 + 2*WB_K; slot^1 was last read before a barrier
#pragma unroll
        for (int ks = 0; ks < WB_K / 2; ++ks) {
            float a[TM], b[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) a[i] = Ds[(ks * 2 + h) * BMc + (wm * TM + i) * 32 + r];
#pragma unroll
            for (int j = 0; j < TN; ++j) b[j] = Xs[(ks * 2 + h) * BNk + (wn * TN + j) * 32 + r];
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        __syncthreads();
        store_tiles();
        __syncthreads();
        slot ^= 1;
    }

    float* slab = ws + (int64_t)z * g.Cout * (int64_t)g.Ktot;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int kc = kc0 + (wn * TN + j) * 32 + r;
            if (kc >= g.Ktot) continue;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int co = co0 + (wm * TM + i) * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                if (co < g.Cout) slab[(int64_t)co * g.Ktot + kc] = acc[i][j][e];
            }
        }
}

// bf16 x 3 weight gradient (see k_conv_gather<..., SPLIT>): both operands are split into bf16 hi / lo on the way into
// LDS.  K = pixels must be contiguous per lane for the bf16 MFMA, so every loader thread takes FOUR consecutive
// pixels of its 4-channel group, transposes the 4x4 block in registers and writes [column][pixel] images.
// Host-checked: one pixel split of x spans < 2 GiB, so 32-bit byte offsets relative to the split's first image
// address every gathered pixel (larger problems take the exact-fp32 kernel above).  Pipelined like k_conv_gather:
//   * raw buffer loads with hardware range checking (offset 0xFFFFFFFF -> zeros): no clamps, selects or 64-bit
//     address arithmetic; dy rows past the split's last pixel fall off the end of the buffer resource;
//   * tile k+1 is converted to its bf16 pieces in the shadow of tile k's MFMAs and the loads of tile k+2 are
//     issued before the barrier; between the two barriers only the LDS writes remain.
// SB (ONE only; bf16-storage mode): x and dy are bf16 tensors - 8-byte loads, the 4 x 4 transposition to "4 pixels of a
// channel" is bit shuffling, nothing is converted.
// XSP (bf16 x 3 only; snn_conv1x1_spikes_wgrad): x holds the saved potentials v_dec of a LIF layer that wrote no spike tensor
// (see k_conv_gather XSP); the operand z = (v_dec > x_th) is formed in the conversion - one exact bf16 piece (0x3F80 or 0),
// no low image, and the product high(dy) * low(x) is not issued: two MFMA products per multiply-add.
template <int TM, int TN, int WM, int WN, int WBK, bool ONE, bool SB = false, bool XSP = false>   // ONE: bf16 x 1 (hi pieces only, one product)
__global__ __launch_bounds__(kThreads, 2) void k_conv_wgrad_pipe(const float* __restrict__ x,
                                                                 const float* __restrict__ dy,
                                                                 float* __restrict__ ws, WgradGeom g) {
    static_assert(WM * WN == 4, "4 waves");
    constexpr int BMc = 32 * TM * WM, BNk = 32 * TN * WN;
    constexpr int DG = BMc / 4, XG = BNk / 4;
    // WBK pixels per LDS stage: 32 for the large tiles; 64 for the small ones, whose 6-MFMA stages were shorter than
    // the memory latency they have to cover (PMC: 58 % of the wave cycles parked in s_waitcnt / barriers)
    static_assert(WBK == 32 || WBK == 64, "stage length");
    static_assert(!SB || ONE, "bf16 storage: one product");
    static_assert(!XSP || (!ONE && !SB), "spikes from potentials: the bf16 x 3 kernel");
    constexpr int ES = SB ? 2 : 4;   // bytes per activation element in HBM
    constexpr int LDW = WBK + 8;      // bf16 row pitch: 80 / 144 bytes, conflict-free ds_read_b128 fragments
    constexpr int NQ = WBK / 4;       // pixel quads per stage
    constexpr int GPP = kThreads / NQ;
    constexpr int DQ = (DG + GPP - 1) / GPP, XQ = (XG + GPP - 1) / GPP;
    __shared__ __attribute__((aligned(16))) __bf16 Dh[BMc * LDW];
    __shared__ __attribute__((aligned(16))) __bf16 Dl[ONE ? 8 : BMc * LDW];
    __shared__ __attribute__((aligned(16))) __bf16 Xh[BNk * LDW];
    __shared__ __attribute__((aligned(16))) __bf16 Xl[(ONE || XSP) ? 8 : BNk * LDW];
    __shared__ __attribute__((aligned(16))) int Pinfo[2][WBK][4];  // {byte offset of the pixel origin, y0, x0, valid}

    const int tid = threadIdx.x;
    const int lane_id = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int r = lane_id & 31, h = lane_id >> 5;

    const int tiles = g.tiles_m * g.tiles_n;
    int L = blockIdx.x, z, tile;
    if (g.splitk % 8 == 0) {
        z = (L % 8) + 8 * (L / (8 * tiles));
        tile = (L / 8) % tiles;
    } else {
        z = L / tiles;
        tile = L % tiles;
    }
    const int co0 = (tile % g.tiles_m) * BMc;
    const int kc0 = (tile / g.tiles_m) * BNk;
    const unsigned p_lo = (unsigned)((int64_t)z * g.pix_per_split);
    unsigned p_hi = p_lo + (unsigned)g.pix_per_split;
    if (p_hi > (unsigned)g.Mtot) p_hi = (unsigned)g.Mtot;
    if (p_lo >= p_hi) p_hi = p_lo;  // an empty split still writes its (zero) slab

    // ---- buffer resources: dy rows of this split, x from the split's first image on
    const unsigned opix = (unsigned)(g.Ho * g.Wo), ipix = (unsigned)(g.H * g.W);
    const unsigned img_lo = p_lo / opix;
    __amdgpu_buffer_rsrc_t rs_d, rs_x;
    {
        const int64_t dbytes = p_hi > p_lo ? (((int64_t)(p_hi - p_lo) - 1) * g.lddy + g.Cout) * ES : 0;
        rs_d = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<char*>(reinterpret_cast<const char*>(dy) + (int64_t)p_lo * g.lddy * ES), 0, (int)dbytes, 0x00020000);
        const int64_t xbytes = ((((int64_t)g.nimg - img_lo) * ipix - 1) * g.ldx + g.Cin) * ES;
        rs_x = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<char*>(reinterpret_cast<const char*>(x) + (int64_t)img_lo * ipix * g.ldx * ES), 0,
                                                 xbytes > 0x7fffffffLL ? 0x7fffffff : (xbytes < 0 ? 0 : (int)xbytes), 0x00020000);
    }

    // ---- loader geometry: 4 consecutive lanes take 4 consecutive channel groups (64 contiguous bytes) of one pixel
    // quad, the next 4 lanes the next quad: thread -> (group = tid % 4 + 4 * (tid / (4 NQ)) + GPP * pass, quad =
    // (tid / 4) % NQ).  The vector memory path then sees 64-byte accesses (with one lane per pixel it handled 64
    // separate 16-byte accesses per load instruction: TA busy 75 % of the kernel on the 32-channel layers), and the
    // LDS stores of a half-wave still fall on 32 distinct bank pairs: (16 g + 2 quad + const) mod 64, g < 4, quad < 8.
    // (ALL lanes on consecutive channel groups collide - 75 % of the LDS cycles were bank conflicts.)
    const int quad = (tid >> 2) % NQ, grp0 = (tid & 3) + 4 * (tid / (4 * NQ));
    int d_off[DQ];       // byte offset of (pixel quad*4, channel group) inside a stage; -1: channels past Cout
    int x_tapoff[XQ];    // byte offset of (tap, ci) relative to a pixel origin
    int x_kh[XQ], x_kw[XQ], x_cq[XQ], d_cq[DQ];
    bool x_ok[XQ];
#pragma unroll
    for (int q = 0; q < DQ; ++q) {
        d_cq[q] = (grp0 + GPP * q) * 4;
        const bool ok = (grp0 + GPP * q) < DG && (co0 + d_cq[q]) < g.Cout;
        d_off[q] = ok ? ((quad * 4) * (int)g.lddy + co0 + d_cq[q]) * ES : -1;
    }
#pragma unroll
    for (int q = 0; q < XQ; ++q) {
        x_cq[q] = (grp0 + GPP * q) * 4;
        const int kc = kc0 + x_cq[q];
        x_ok[q] = (grp0 + GPP * q) < XG && kc < g.Ktot;
        const int kcc = x_ok[q] ? kc : 0;
        const int tap = kcc / g.Cin, ci = kcc - tap * g.Cin;
        x_kh[q] = tap / g.KW;
        x_kw[q] = tap - x_kh[q] * g.KW;
        x_tapoff[q] = ((x_kh[q] * g.W + x_kw[q]) * (int)g.ldx + ci) * ES;
    }

    // ---- pixel decode, 32 lanes, one stage ahead (carries instead of divisions inside the loop)
    int d_img = 0, d_oy = 0, d_ox = 0;
    unsigned d_p = p_lo + tid;
    if (tid < WBK) {
        const unsigned pp = d_p < (unsigned)g.Mtot ? d_p : 0u;
        const unsigned t = pp / (unsigned)g.Wo;
        d_ox = (int)(pp - t * (unsigned)g.Wo);
        const unsigned im = t / (unsigned)g.Ho;
        d_oy = (int)(t - im * (unsigned)g.Ho);
        d_img = (int)(im - img_lo);
    }
    auto decode = [&](int slot) {
        if (tid < WBK) {
            const int y0 = d_oy * g.stride - g.pad, x0 = d_ox * g.stride - g.pad;
            int4 info;
            info.x = ((d_img * (int)ipix + y0 * g.W + x0) * (int)g.ldx) * ES;
            info.y = y0;
            info.z = x0;
            info.w = d_p < p_hi ? 1 : 0;
            *reinterpret_cast<int4*>(&Pinfo[slot][tid][0]) = info;
            d_p += WBK;
            d_ox += WBK;
            while (d_ox >= g.Wo) {
                d_ox -= g.Wo;
                if (++d_oy == g.Ho) {
                    d_oy = 0;
                    ++d_img;
                }
            }
        }
    };

    // operand quads on their way to LDS: 4 fp32 values, or (SB) 4 bf16 values as two dwords (integer-typed: see k_conv_gather)
    using OReg = typename std::conditional<SB, u32x2, f32x4>::type;
    OReg rd[DQ][4], rx[XQ][4];
    auto load_tiles = [&](unsigned p0, int slot) {
        const int dstage = (int)(p0 - p_lo) * (int)g.lddy * ES;  // scalar
        auto fetch = [&](__amdgpu_buffer_rsrc_t rs, int voff) -> OReg {
            if constexpr (SB) return __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(rs, voff, 0, 0));
            else return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, voff, 0, 0));
        };
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int4 info = *reinterpret_cast<const int4*>(&Pinfo[slot][quad * 4 + e][0]);
#pragma unroll
            for (int q = 0; q < DQ; ++q) {
                const int voff = d_off[q] < 0 ? -1 : d_off[q] + e * (int)g.lddy * ES + dstage;
                rd[q][e] = fetch(rs_d, voff);
            }
#pragma unroll
            for (int q = 0; q < XQ; ++q) {
                const int iy = info.y + x_kh[q], ix = info.z + x_kw[q];
                const bool ok = (info.w != 0) & x_ok[q] & ((unsigned)iy < (unsigned)g.H) & ((unsigned)ix < (unsigned)g.W);
                const int voff = ok ? info.x + x_tapoff[q] : -1;
                rx[q][e] = fetch(rs_x, voff);
            }
        }
    };
    // 4 pixels x 4 channels -> per channel the 4 pixels as bf16 hi / lo (8 bytes each), kept in registers
    bf16x4 pd[DQ][4][2], px[XQ][4][2];
    auto convert_quad = [&](const OReg (&v)[4], bf16x4 (&out)[4][2]) {
        if constexpr (SB) {   // v[pixel] holds channels (0, 1) in element 0 and (2, 3) in element 1, as bf16 pairs
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const unsigned a0 = v[0][c >> 1], a1 = v[1][c >> 1], a2 = v[2][c >> 1], a3 = v[3][c >> 1];
                u32x2 o;
                if (c & 1) o = u32x2{(a0 >> 16) | (a1 & 0xffff0000u), (a2 >> 16) | (a3 & 0xffff0000u)};
                else o = u32x2{(a0 & 0xffffu) | (a1 << 16), (a2 & 0xffffu) | (a3 << 16)};
                out[c][0] = __builtin_bit_cast(bf16x4, o);
            }
        } else {
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int e = 0; e < 4; e += 2) {
                f32x2 rest = {v[e][c], v[e + 1][c]};
                bf16x2 pp = __builtin_convertvector(rest, bf16x2);
                const unsigned bits = __builtin_bit_cast(unsigned, pp);
                out[c][0][e] = pp[0]; out[c][0][e + 1] = pp[1];
                if constexpr (!ONE) {
                    rest[0] -= __builtin_bit_cast(float, bits << 16);
                    rest[1] -= __builtin_bit_cast(float, bits & 0xffff0000u);
                    pp = __builtin_convertvector(rest, bf16x2);
                    out[c][1][e] = pp[0]; out[c][1][e + 1] = pp[1];
                }
            }
        }
    };
    auto convert_spikes = [&](const OReg (&v)[4], bf16x4 (&out)[4][2]) {   // XSP: 4 pixels of a channel as bf16 {0, 1}
        if constexpr (XSP) {
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                u32x2 o;
#pragma unroll
                for (int e = 0; e < 4; e += 2)
                    o[e >> 1] = (v[e][c] > g.x_th ? 0x3F80u : 0u) | (v[e + 1][c] > g.x_th ? 0x3F800000u : 0u);
                out[c][0] = __builtin_bit_cast(bf16x4, o);
            }
        }
    };
    auto write_tiles = [&]() {
#pragma unroll
        for (int q = 0; q < DQ; ++q)
            if (grp0 + GPP * q < DG) {
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    *reinterpret_cast<bf16x4*>(&Dh[(d_cq[q] + c) * LDW + quad * 4]) = pd[q][c][0];
                    if constexpr (!ONE) *reinterpret_cast<bf16x4*>(&Dl[(d_cq[q] + c) * LDW + quad * 4]) = pd[q][c][1];
                }
            }
#pragma unroll
        for (int q = 0; q < XQ; ++q)
            if (grp0 + GPP * q < XG) {
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    *reinterpret_cast<bf16x4*>(&Xh[(x_cq[q] + c) * LDW + quad * 4]) = px[q][c][0];
                    if constexpr (!ONE && !XSP) *reinterpret_cast<bf16x4*>(&Xl[(x_cq[q] + c) * LDW + quad * 4]) = px[q][c][1];
                }
            }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    auto mfma_group = [&](int ks) {
        bf16x8 ah[TM], al[TM], bh[TN], bl[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int off = ((wm * TM + i) * 32 + r) * LDW + ks * 16 + 8 * h;
            ah[i] = *reinterpret_cast<const bf16x8*>(&Dh[off]);
            if constexpr (!ONE) al[i] = *reinterpret_cast<const bf16x8*>(&Dl[off]);
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int off = ((wn * TN + j) * 32 + r) * LDW + ks * 16 + 8 * h;
            bh[j] = *reinterpret_cast<const bf16x8*>(&Xh[off]);
            if constexpr (!ONE && !XSP) bl[j] = *reinterpret_cast<const bf16x8*>(&Xl[off]);
        }
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                if constexpr (!ONE) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh[j], acc[i][j], 0, 0, 0);
                    if constexpr (!XSP) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl[j], acc[i][j], 0, 0, 0);
                }
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);
            }
    };
    constexpr int NM = TM * TN * (ONE ? 1 : (XSP ? 2 : 3));
    constexpr int NREAD = XSP ? 2 * TM + TN : (TM + TN) * (ONE ? 1 : 2);
    constexpr int CQ_OPS = SB ? 12 : 56;   // VALU per converted quad (approx.)
    constexpr int VPG_D = (DQ * CQ_OPS + NM - 1) / NM, VPG_X = (XQ * (XSP ? 24 : CQ_OPS) + NM - 1) / NM;

    decode(0);
    __syncthreads();
    load_tiles(p_lo, 0);
    decode(1);
    __syncthreads();
#pragma unroll
    for (int q = 0; q < DQ; ++q) convert_quad(rd[q], pd[q]);
#pragma unroll
    for (int q = 0; q < XQ; ++q) {
        if constexpr (XSP) convert_spikes(rx[q], px[q]);
        else convert_quad(rx[q], px[q]);
    }
    write_tiles();
    load_tiles(p_lo + WBK, 1);
    decode(0);
    __syncthreads();

    int slot = 0;  // Pinfo slot of tile k+2
#pragma unroll 1
    for (unsigned p0 = p_lo; p0 < p_hi; p0 += WBK) {
        mfma_group(0);
#pragma unroll
        for (int q = 0; q < DQ; ++q) convert_quad(rd[q], pd[q]);
        __builtin_amdgcn_sched_group_barrier(0x100, NREAD, 0);
#pragma unroll
        for (int m = 0; m < NM; ++m) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, VPG_D, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        mfma_group(1);
#pragma unroll
        for (int q = 0; q < XQ; ++q) {
            if constexpr (XSP) convert_spikes(rx[q], px[q]);
            else convert_quad(rx[q], px[q]);
        }
        __builtin_amdgcn_sched_group_barrier(0x100, NREAD, 0);
#pragma unroll
        for (int m = 0; m < NM; ++m) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, VPG_X, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ks = 2; ks < WBK / 16; ++ks) mfma_group(ks);
        load_tiles(p0 + 2 * WBK, slot);
        __syncthreads();
        write_tiles();
        decode(slot ^ 1);
        __syncthreads();
        slot ^= 1;
    }

    float* slab = ws + (int64_t)z * g.Cout * (int64_t)g.Ktot;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int kcol = kc0 + (wn * TN + j) * 32 + r;
            if (kcol >= g.Ktot) continue;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int co = co0 + (wm * TM + i) * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                if (co < g.Cout) slab[(int64_t)co * g.Ktot + kcol] = acc[i][j][e];
            }
        }
}

// Ordered reduction of the split-K slabs ws[splitk][n] -> dw[n].  KG thread groups share the slabs of one element
// (each sums a contiguous run in slab order), then group 0 adds the KG partial sums in group order: fixed order,
// bitwise reproducible, and the early layers (n of a few hundred, splitk of several hundred) are no longer one
// latency-bound serial chain per thread.
template <int KG>
__global__ void k_wgrad_reduce(const float* __restrict__ ws, float* __restrict__ dw, int64_t n, int splitk,
                               int accumulate) {
    constexpr int EL = kThreads / KG;
    __shared__ float part[KG][EL];
    const int el = threadIdx.x % EL, kg = threadIdx.x / EL;
    const int64_t e = (int64_t)blockIdx.x * EL + el;
    const int per = (splitk + KG - 1) / KG;
    const int k0 = kg * per;
    const int k1 = k0 + per < splitk ? k0 + per : splitk;
    float s = 0.f;
    if (e < n) {
        int k = k0;
        for (; k + 4 <= k1; k += 4) {
            const float a = ws[(int64_t)k * n + e], b = ws[(int64_t)(k + 1) * n + e];
            const float c = ws[(int64_t)(k + 2) * n + e], d = ws[(int64_t)(k + 3) * n + e];
            s = (((s + a) + b) + c) + d;
        }
        for (; k < k1; ++k) s += ws[(int64_t)k * n + e];
    }
    if (KG > 1) {
        part[kg][el] = s;
        __syncthreads();
        if (kg == 0) {
            s = part[0][el];
#pragma unroll
            for (int g = 1; g < KG; ++g) s += part[g][el];
        }
    }
    if (kg == 0 && e < n) dw[e] = accumulate ? dw[e] + s : s;
}

// The same ordered reduction, 16 bytes per lane and whole 4 KiB runs per block (the kernel above reads 16 ... 64 bytes
// per slab row and block: 0.8 TB/s on the 2 000-slab workspaces of the narrow layers).  dst[g][e] = sum of rows
// [g * per, (g + 1) * per) of src in row order (+ dst when accumulate); a first pass reduces groups of rows IN PLACE
// (into the first row of each group - every thread only overwrites positions it has read itself), a second pass adds
// the group heads.
__global__ __launch_bounds__(kThreads) void k_wgrad_reduce4(const float* __restrict__ src, int64_t n, int rows, int per,
                                                            int64_t row_stride, float* __restrict__ dst,
                                                            int64_t dst_group_stride, int accumulate) {
    const int64_t e = ((int64_t)blockIdx.x * kThreads + threadIdx.x) * 4;
    if (e >= n) return;
    const int g = blockIdx.y;
    const int r0 = g * per;
    const int r1 = r0 + per < rows ? r0 + per : rows;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    int r = r0;
    for (; r + 4 <= r1; r += 4) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(src + (int64_t)r * row_stride + e);
        const f32x4 b = *reinterpret_cast<const f32x4*>(src + (int64_t)(r + 1) * row_stride + e);
        const f32x4 c = *reinterpret_cast<const f32x4*>(src + (int64_t)(r + 2) * row_stride + e);
        const f32x4 d = *reinterpret_cast<const f32x4*>(src + (int64_t)(r + 3) * row_stride + e);
        s = (((s + a) + b) + c) + d;
    }
    for (; r < r1; ++r) s = s + *reinterpret_cast<const f32x4*>(src + (int64_t)r * row_stride + e);
    float* out = dst + (int64_t)g * dst_group_stride + e;
    if (accumulate) s = *reinterpret_cast<const f32x4*>(out) + s;
    *reinterpret_cast<f32x4*>(out) = s;
}

// The ordered reduction in ONE launch (round 4; the two-pass form above cost two latency-bound launches behind every one of
// the ~38 weight gradients of a step): a block owns 256 consecutive elements (64 lanes x 16 bytes = 1 KiB runs per slab
// row), its KG waves each sum a contiguous run of slab rows in row order, then wave 0 adds the KG partial sums in wave
// order: fixed order, bitwise reproducible.
template <int KG>
__global__ __launch_bounds__(64 * KG) void k_wgrad_reduce_once(const float* __restrict__ src, int64_t n, int rows,
                                                              float* __restrict__ dst, int accumulate) {
    __shared__ f32x4 part[KG][64];
    const int lane = threadIdx.x & 63, kg = threadIdx.x >> 6;
    const int64_t e = ((int64_t)blockIdx.x * 64 + lane) * 4;
    const int per = (rows + KG - 1) / KG;
    const int r0 = kg * per;
    const int r1 = r0 + per < rows ? r0 + per : rows;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    if (e < n) {
        int r = r0;
        for (; r + 4 <= r1; r += 4) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(src + (int64_t)r * n + e);
            const f32x4 b = *reinterpret_cast<const f32x4*>(src + (int64_t)(r + 1) * n + e);
            const f32x4 c = *reinterpret_cast<const f32x4*>(src + (int64_t)(r + 2) * n + e);
            const f32x4 d = *reinterpret_cast<const f32x4*>(src + (int64_t)(r + 3) * n + e);
            s = (((s + a) + b) + c) + d;
        }
        for (; r < r1; ++r) s = s + *reinterpret_cast<const f32x4*>(src + (int64_t)r * n + e);
    }
    if (KG > 1) {
        part[kg][lane] = s;
        __syncthreads();
        if (kg == 0) {
            s = part[0][lane];
#pragma unroll
            for (int g = 1; g < KG; ++g) s = s + part[g][lane];
        }
    }
    if (kg == 0 && e < n) {
        float* out = dst + e;
        if (accumulate) s = *reinterpret_cast<const f32x4*>(out) + s;
        *reinterpret_cast<f32x4*>(out) = s;
    }
}

static bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

static bool aligned8(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 7u) == 0; }

template <bool DGRAD, int SPLIT, bool SB = false, bool XSP = false>
static int launch_gather(const float* in, const float* wk, const void* wk_split, float* out, const ConvGeom& g,
                         const float* addend, int64_t ld_add, const float* addend2, int64_t ld_add2, hipStream_t st,
                         const char* name) {
    const bool vec = (g.IC % 4 == 0) && (g.ldi % 4 == 0) && (SB ? aligned8(in) : aligned16(in)) && aligned16(wk);
    const int64_t gm = snn_ceil_div(g.Mtot, BM);
    SNN_REQUIRE(g.Mtot < 0x7fffffffLL && (int64_t)g.IH * g.IW < 0x7fffffffLL, "%s: too many pixels", name);
    const int ntaps = DGRAD ? g.nkh * g.nkw : g.KH * g.KW;
    static const bool no_fast = snn_tuning_env("SNN_CONV_NO_FAST") != nullptr;  // tuning / bisecting aid
    const bool fast = vec && !no_fast && g.IC % BK == 0 && ntaps >= 1 && ntaps <= 31 &&
                      (DGRAD ? g.nkh : g.KH) <= 6 && (DGRAD ? g.nkw : g.KW) <= 6 &&
                      (int64_t)g.IH * g.IW * g.ldi * 16 < 0x7fffffffLL && (int64_t)g.OC * g.KtotFull * 4 < 0x7fffffffLL;
    ConvGeom gg = g;
    const auto avec = [](const void* p) { return SB ? aligned8(p) : aligned16(p); };   // 4 elements per access
    gg.out_vec = (g.ldo % 4 == 0) && avec(out) && (!addend || (ld_add % 4 == 0 && avec(addend))) &&
                 (!addend2 || (ld_add2 % 4 == 0 && avec(addend2)));
    if constexpr (SB)
        SNN_REQUIRE(fast, "%s: bf16 storage covers the pipelined implicit GEMM only (channels a multiple of 32, pixel "
                    "stride a multiple of 4, 8-byte aligned tensors): %d channels, stride %lld", name, g.IC, (long long)g.ldi);
    if constexpr (XSP)
        SNN_REQUIRE(fast, "%s: covers the pipelined implicit GEMM only (input channels a multiple of 32, pixel stride a "
                    "multiple of 4, 16-byte aligned tensors): %d channels, stride %lld", name, g.IC, (long long)g.ldi);
    // the pre-split weight image serves the pipelined kernel in its two-piece modes; every other path converts wk itself
    const bool presplit = wk_split != nullptr && (SPLIT == 2 || SPLIT == 4) && aligned16(wk_split);
#define SNN_CONV_LAUNCH(BN_, WM_, WN_)                                                                      \
    do {                                                                                                    \
        gg.mtiles = (int)gm;                                                                                \
        gg.mtiles_per_xcd = (int)snn_ceil_div(gm, 8);                                                       \
        gg.ntiles = (int)snn_ceil_div(g.OC, BN_);                                                           \
        SNN_REQUIRE((int64_t)gg.mtiles_per_xcd * 8 * gg.ntiles <= 0x7fffffffLL, "%s: grid too large", name); \
        dim3 grid((unsigned)(gg.mtiles_per_xcd * 8 * gg.ntiles));                                           \
        if constexpr (SB) {                                                                                 \
            hipLaunchKernelGGL((k_conv_gather<BN_, WM_, WN_, DGRAD, true, 5, true, false, true>), grid,     \
                               dim3(kThreads), 0, st, in, wk, out, gg, addend, ld_add, addend2, ld_add2);   \
        } else if constexpr (XSP) {                                                                         \
            hipLaunchKernelGGL((k_conv_gather<BN_, WM_, WN_, false, true, 4, true, false, false, true>), grid, \
                               dim3(kThreads), 0, st, in, wk, out, gg, addend, ld_add, addend2, ld_add2);   \
        } else if (fast && presplit) {                                                                      \
            if constexpr (SPLIT == 2 || SPLIT == 4)                                                         \
                hipLaunchKernelGGL((k_conv_gather<BN_, WM_, WN_, DGRAD, true, SPLIT, true, true>), grid,    \
                                   dim3(kThreads), 0, st, in, static_cast<const float*>(wk_split), out, gg, addend, \
                                   ld_add, addend2, ld_add2);                                               \
        } else if (fast)                                                                                    \
            hipLaunchKernelGGL((k_conv_gather<BN_, WM_, WN_, DGRAD, true, SPLIT, true>), grid, dim3(kThreads), 0, \
                               st, in, wk, out, gg, addend, ld_add, addend2, ld_add2);                                        \
        else if (vec)                                                                                       \
            hipLaunchKernelGGL((k_conv_gather<BN_, WM_, WN_, DGRAD, true, (SPLIT == 4 ? 3 : (SPLIT == 5 ? 2 : SPLIT)), false>), grid, dim3(kThreads), 0, \
                               st, in, wk, out, gg, addend, ld_add, addend2, ld_add2);                                        \
        else                                                                                                \
            hipLaunchKernelGGL((k_conv_gather<BN_, WM_, WN_, DGRAD, false, 0, false>), grid, dim3(kThreads), 0, st, \
                               in, wk, out, gg, addend, ld_add, addend2, ld_add2);                        \
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

// ------------------------------------------------------------------------------------------ direct 3x3
// Direct 3x3 / stride 1 / pad 1 convolution for layers with <= 64 output channels (the 32- and 64-channel
// bottleneck convolutions of the two high-resolution stages).  In the implicit-GEMM kernel above every input element
// is fetched and split into its bf16 pieces once PER TAP (9x), and with so few output channels that conversion work,
// not the MFMAs, sets the speed.  Here a block owns an 8 x 16 patch of output pixels: per 32-channel chunk the
// 10 x 18 input halo is fetched and converted ONCE into LDS and all nine taps read it at shifted row offsets; only
// the weight tiles stream through the register pipeline (convert in the MFMA shadow, as in k_conv_gather).
// FLIP selects the data-gradient form: out = conv(dy, w^T with the taps mirrored).
struct DirectGeom {
    int IH, IW, IC, OC;     // gathered tensor / produced channels (output is IH x IW as well)
    int64_t ldi, ldo;
    int nimg, pht, pwt;     // patches per image column / row
    int tiles, tiles_per_xcd;
    int KtotFull;           // 9 * IC
    int out_vec;
    // forward only: statistics partials of the BatchNorm that follows (null: none); a patch lies in ONE frame, so
    // chunk = the patch's index among the patches of its timestep (bn_chunks = frames per step * patches per image)
    double* bn_partial;
    int bn_chunks;
};
constexpr int DPH = 8, DPW = 16, DHW = DPW + 2, DHALO = (DPH + 2) * DHW;  // 180 halo pixels
constexpr int DHROWS = (DHALO + 7) / 8 * 8;                               // 184 LDS rows (whole groups of 8)

template <int BN, int WM, int WN, int SPLIT, int TPS, bool FLIP>
__global__ __launch_bounds__(kThreads, 2) void k_conv_direct3(const float* __restrict__ in, const float* __restrict__ wk,
                                                              float* __restrict__ out, DirectGeom g,
                                                              const float* __restrict__ addend, int64_t ld_add,
                                                              const float* __restrict__ addend2, int64_t ld_add2) {
    static_assert(WM * WN == 4 && (SPLIT == 2 || SPLIT == 3 || SPLIT == 4) && (TPS == 1 || TPS == 3), "configuration");
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
    constexpr int NP = SPLIT == 3 ? 3 : 2, NPROD = SPLIT == 3 ? 6 : 3;
    constexpr int SPC = 9 / TPS;                 // stages per 32-channel chunk
    constexpr int BJ = TPS * BN / 32;            // weight f32x4 per thread and stage
    constexpr int AJ = (DHROWS * 8 + kThreads - 1) / kThreads;  // halo f32x4 per thread and chunk (6)
    constexpr int A_BYTES = NP * DHROWS * LDB * 2, B_BYTES = NP * TPS * BN * LDB * 2;
    __shared__ __attribute__((aligned(16))) unsigned char smem[A_BYTES + B_BYTES];
    __bf16* Ai[3];
    __bf16* Bi[3];
#pragma unroll
    for (int pz = 0; pz < NP; ++pz) {  // piece images: 0 hi, 1 lo, 2 mid
        Ai[pz] = reinterpret_cast<__bf16*>(smem) + pz * DHROWS * LDB;
        Bi[pz] = reinterpret_cast<__bf16*>(smem + A_BYTES) + pz * TPS * BN * LDB;
    }

    const int tid = threadIdx.x;
    const int lane_id = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int r = lane_id & 31, h = lane_id >> 5;

    // ---- persistent blocks, XCD-aware: XCD x owns the x-th contiguous eighth of the patches; its blocks (ids x, x+8,
    // ...) walk that range with stride "blocks per XCD".  A 32-channel layer has only 3 stages per patch, so the
    // per-patch prologue (first halo + weights with their full load latency) and the epilogue are overlapped with
    // the neighbouring patches instead of being paid per block.
    const int nbx = gridDim.x >> 3;
    const int t_lo = (blockIdx.x & 7) * g.tiles_per_xcd;
    const int t_hi = t_lo + g.tiles_per_xcd < g.tiles ? t_lo + g.tiles_per_xcd : g.tiles;
    int tile = t_lo + (blockIdx.x >> 3);
    if (tile >= t_hi) return;
    const int ppi = g.pht * g.pwt;
    const int64_t ipix = (int64_t)g.IH * g.IW;
    int img = 0, oy0 = 0, ox0 = 0;
    __amdgpu_buffer_rsrc_t rs_a, rs_b;
    rs_b = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(wk), 0, g.OC * g.KtotFull * 4, 0x00020000);

    // ---- loader geometry.  Row permutation inside groups of 8 (rows R, R+1 -> R', R'+4): conflict-free LDS stores.
    const int kq = (tid & 7) * 4;
    int a_off[AJ], a_row[AJ];  // byte offset of (halo pixel, channel kq) in the image (-1: outside), LDS row (-1: none)
    int a_hy[AJ], a_hx[AJ];    // halo coordinates of the thread's rows (tile independent)
#pragma unroll
    for (int j = 0; j < AJ; ++j) {
        const int rp = (tid >> 3) + 32 * j;
        const int hrow = (rp & ~7) | ((rp >> 1) & 3) | ((rp & 1) << 2);
        a_hy[j] = hrow / DHW;
        a_hx[j] = hrow - a_hy[j] * DHW;
        a_row[j] = rp < DHROWS ? hrow : -1;
        if (hrow >= DHALO) a_hy[j] = -0x10000;  // never inside an image
    }
    auto setup_tile = [&](int t) {  // t is block-uniform
        img = t / ppi;
        const int prem = t - img * ppi;
        const int ty = prem / g.pwt, tx = prem - ty * g.pwt;
        oy0 = ty * DPH;
        ox0 = tx * DPW;
        rs_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(in + (int64_t)img * ipix * g.ldi), 0,
                                                 (int)(((ipix - 1) * g.ldi + g.IC) * 4), 0x00020000);
#pragma unroll
        for (int j = 0; j < AJ; ++j) {
            const int iy = oy0 - 1 + a_hy[j], ix = ox0 - 1 + a_hx[j];
            const bool inside = a_row[j] >= 0 && (unsigned)iy < (unsigned)g.IH && (unsigned)ix < (unsigned)g.IW;
            a_off[j] = inside ? ((iy * g.IW + ix) * (int)g.ldi + kq) * 4 : -1;
        }
    };
    const int lrr = tid >> 3;
    const int lr = ((lrr >> 1) & 3) + 4 * (lrr & 1) + 8 * (lrr >> 3);
    unsigned b_off[BJ];  // byte offset of (weight row, column kq); >= 2^31 for rows past OC
    int b_tp[BJ];
#pragma unroll
    for (int j = 0; j < BJ; ++j) {
        const int idx = j * 32 + lr;
        b_tp[j] = idx / BN;
        const int n = idx - b_tp[j] * BN;
        b_off[j] = n < g.OC ? (unsigned)(n * g.KtotFull + kq) * 4u : 0x80000000u;
    }

    f32x4 ra[AJ], rb[BJ];
    bf16x4 pb[BJ][NP];
    auto load_a = [&](int chunk) {  // chunk is block-uniform
        const int c4 = chunk * BK * 4;
        const bool live = chunk * BK < g.IC;
#pragma unroll
        for (int j = 0; j < AJ; ++j) {
            const int voff = (live && a_off[j] >= 0) ? a_off[j] + c4 : -1;
            ra[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_a, voff, 0, 0));
        }
    };
    auto load_b = [&](int stage) {  // stage = chunk * SPC + tap group (block-uniform)
        const int chunk = stage / SPC, tg = stage - chunk * SPC;
        const bool live = chunk * BK < g.IC;
#pragma unroll
        for (int j = 0; j < BJ; ++j) {
            const int tap = tg * TPS + b_tp[j];
            const int wtap = FLIP ? 8 - tap : tap;
            const unsigned col = (unsigned)(wtap * g.IC + chunk * BK) * 4u;
            const int voff = live ? (int)(b_off[j] + col) : -1;
            rb[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_b, voff, 0, 0));
        }
    };
    auto convert = [&](const f32x4& v, bf16x4* o, float scale) {  // o[0] hi, o[1] lo, o[2] mid
        if constexpr (SPLIT == 4) {  // fp16 pieces (see k_conv_gather)
            u32x2 hi, lo;
#pragma unroll
            for (int e = 0; e < 4; e += 2) {
                const float a = v[e] * scale, b = v[e + 1] * scale;
                const f16x2 ph = __builtin_convertvector(f32x2{a, b}, f16x2);  // RNE: out of range -> inf (loud)
                const f16x2 pl = __builtin_convertvector(f32x2{a - (float)ph[0], b - (float)ph[1]}, f16x2);
                hi[e >> 1] = __builtin_bit_cast(unsigned, ph);
                lo[e >> 1] = __builtin_bit_cast(unsigned, pl);
            }
            o[0] = __builtin_bit_cast(bf16x4, hi);
            o[1] = __builtin_bit_cast(bf16x4, lo);
            return;
        }
#pragma unroll
        for (int e = 0; e < 4; e += 2) {
            f32x2 rest = {v[e], v[e + 1]};
            bf16x2 p = __builtin_convertvector(rest, bf16x2);
            unsigned bits = __builtin_bit_cast(unsigned, p);
            o[0][e] = p[0]; o[0][e + 1] = p[1];
            rest[0] -= __builtin_bit_cast(float, bits << 16);
            rest[1] -= __builtin_bit_cast(float, bits & 0xffff0000u);
            if (SPLIT == 3) {
                p = __builtin_convertvector(rest, bf16x2);
                bits = __builtin_bit_cast(unsigned, p);
                o[2][e] = p[0]; o[2][e + 1] = p[1];
                rest[0] -= __builtin_bit_cast(float, bits << 16);
                rest[1] -= __builtin_bit_cast(float, bits & 0xffff0000u);
            }
            p = __builtin_convertvector(rest, bf16x2);
            o[1][e] = p[0]; o[1][e + 1] = p[1];
        }
    };
    auto write_a = [&]() {  // convert + store the halo of the next chunk (once per 9 taps: not worth hiding)
#pragma unroll
        for (int j = 0; j < AJ; ++j) {
            if (a_row[j] < 0) continue;
            bf16x4 o[3];
            convert(ra[j], o, kF16ActScale);
#pragma unroll
            for (int pz = 0; pz < NP; ++pz) *reinterpret_cast<bf16x4*>(&Ai[pz][a_row[j] * LDB + kq]) = o[pz];
        }
    };
    auto write_b = [&]() {
#pragma unroll
        for (int j = 0; j < BJ; ++j)
#pragma unroll
            for (int pz = 0; pz < NP; ++pz)
                *reinterpret_cast<bf16x4*>(&Bi[pz][(j * 32 + lr) * LDB + kq]) = pb[j][pz];
    };

    // A-fragment row of lane r in M-tile i: patch pixel p = (wm*TM + i)*32 + r -> halo pixel (p/16, p%16) + tap shift
    int a_frag[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int p = (wm * TM + i) * 32 + r;
        a_frag[i] = ((p >> 4) * DHW + (p & 15)) * LDB + 8 * h;
    }

    f32x16 acc[TM][TN];
    struct Frag {
        bf16x8 a[TM][NP], b[TN][NP];
    };
    auto read_frag = [&](Frag& f, int tp, int kh, int kw, int ks) {
        const int toff = (kh * DHW + kw) * LDB + ks * 16;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int pz = 0; pz < NP; ++pz) f.a[i][pz] = *reinterpret_cast<const bf16x8*>(&Ai[pz][a_frag[i] + toff]);
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int pz = 0; pz < NP; ++pz)
                f.b[j][pz] = *reinterpret_cast<const bf16x8*>(
                    &Bi[pz][(tp * BN + (wn * TN + j) * 32 + r) * LDB + ks * 16 + 8 * h]);
    };
    auto mfma_frag = [&](const Frag& f) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {  // small terms first; pieces: 0 hi, 1 lo, 2 mid
                if constexpr (SPLIT == 4) {
                    const f16x8 ah = __builtin_bit_cast(f16x8, f.a[i][0]), al = __builtin_bit_cast(f16x8, f.a[i][1]);
                    const f16x8 bh = __builtin_bit_cast(f16x8, f.b[j][0]), bl = __builtin_bit_cast(f16x8, f.b[j][1]);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc[i][j], 0, 0, 0);
                    continue;
                }
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a[i][1], f.b[j][0], acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a[i][0], f.b[j][1], acc[i][j], 0, 0, 0);
                if (SPLIT == 3) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a[i][2], f.b[j][2], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a[i][2], f.b[j][0], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a[i][0], f.b[j][2], acc[i][j], 0, 0, 0);
                }
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a[i][0], f.b[j][0], acc[i][j], 0, 0, 0);
            }
    };

    const int nchunks = g.IC / BK;
    const int nstages = nchunks * SPC;
    constexpr int NM = TM * TN * NPROD;                      // MFMAs per (tap, k16) group
    constexpr int NGRP = TPS * 2;                            // groups per stage
    constexpr int CONV_OPS = SPLIT == 3 ? 24 : 14;
    constexpr int VPG = (BJ * CONV_OPS + NGRP * NM - 1) / (NGRP * NM);
    setup_tile(tile);
    load_a(0);
    load_b(0);
#pragma unroll 1
    for (;;) {
        // ---- patch prologue: halo of chunk 0 and weights of stage 0 into LDS; the next ones into registers
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
        write_a();
#pragma unroll
        for (int j = 0; j < BJ; ++j) convert(rb[j], pb[j], kF16WeightScale);
        write_b();
        load_a(1);
        load_b(1);
        __syncthreads();
        int chunk = 0, tg = 0;
    #pragma unroll 1
        for (int s = 0; s < nstages; ++s) {
            // fragment reads run one (tap, k16) group ahead of the MFMAs that consume them
            Frag fr[2];
            {
                const int tap0 = tg * TPS;
                read_frag(fr[0], 0, TPS == 3 ? tg : tap0 / 3, TPS == 3 ? 0 : tap0 - (tap0 / 3) * 3, 0);
            }
            __builtin_amdgcn_sched_group_barrier(0x100, (TM + TN) * NP, 0);
#pragma unroll
            for (int gq = 0; gq < NGRP; ++gq) {
                if (gq + 1 < NGRP) {
                    const int tp = (gq + 1) >> 1, ks = (gq + 1) & 1;
                    const int tap = tg * TPS + tp;
                    read_frag(fr[(gq + 1) & 1], tp, TPS == 3 ? tg : tap / 3, TPS == 3 ? tp : tap - (tap / 3) * 3, ks);
                }
                mfma_frag(fr[gq & 1]);
                if (gq == 0) {
#pragma unroll
                    for (int j = 0; j < BJ; ++j) convert(rb[j], pb[j], kF16WeightScale);  // stage s+1, in the MFMA shadow
                }
            }
            // schedule shape: [reads of group g+1] then the MFMAs of group g, each followed by its share of the VALU
#pragma unroll
            for (int gq = 0; gq < NGRP; ++gq) {
                __builtin_amdgcn_sched_group_barrier(0x100, (TM + TN) * NP, 0);
#pragma unroll
                for (int m = 0; m < NM; ++m) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002, VPG, 0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            load_b(s + 2);
            __syncthreads();
            write_b();
            if (tg == SPC - 1) {  // the next stage starts a new chunk: replace the halo, fetch the one after
                write_a();
                load_a(chunk + 2);
                ++chunk;
                tg = 0;
            } else {
                ++tg;
            }
            __syncthreads();
        }
        // ---- the next patch's first halo / weights are fetched while this patch's results are stored
        const int e_img = img, e_oy0 = oy0, e_ox0 = ox0, e_tile = tile;
        tile += nbx;
        const bool more = tile < t_hi;
        if (more) {
            setup_tile(tile);
            load_a(0);
            load_b(0);
        }
        // ---- epilogue: accumulators -> LDS (per wave) -> 16-byte stores with up to two fused addends
        constexpr int EW = TN * 32 + 4, LPR = TN * 8, RPP = 64 / LPR;
        static_assert(4 * 32 * EW * 4 <= A_BYTES + B_BYTES, "epilogue staging does not fit");
        float* stage = reinterpret_cast<float*>(smem) + wave * 32 * EW;
        const bool ovec = g.out_vec != 0;
        if (!FLIP && g.bn_partial != nullptr) {
            // BatchNorm partials of this patch (one frame, hence one timestep); pixels past the image edge are dropped
            const bool whole = e_oy0 + DPH <= g.IH && e_ox0 + DPW <= g.IW;
            double ss[TN], qq[TN];
    #pragma unroll
            for (int j = 0; j < TN; ++j) ss[j] = qq[j] = 0.0;
    #pragma unroll
            for (int i = 0; i < TM; ++i)
    #pragma unroll
                for (int j = 0; j < TN; ++j)
    #pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const int p = (wm * TM + i) * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                        const bool ok = whole || (e_oy0 + (p >> 4) < g.IH && e_ox0 + (p & 15) < g.IW);
                        const float v = SPLIT == 4 ? acc[i][j][e] * kF16Unscale : acc[i][j][e];
                        const double d = ok ? (double)v : 0.0;
                        ss[j] += d;
                        qq[j] = fma(d, d, qq[j]);
                    }
            static_assert(4 * TN * 32 * 2 * 8 <= A_BYTES + B_BYTES, "statistics scratch does not fit");
            // patches are numbered frame by frame: timestep = patch number / (patches per timestep)
            stat_flush<WM, WN, TN>(ss, qq, reinterpret_cast<double*>(smem), g.bn_partial, e_tile / g.bn_chunks,
                                   e_tile % g.bn_chunks, g.bn_chunks, 0, g.OC, tid);
        }
    #pragma unroll
        for (int i = 0; i < TM; ++i) {
    #pragma unroll
            for (int j = 0; j < TN; ++j)
    #pragma unroll
                for (int e = 0; e < 16; ++e)
                    stage[((e & 3) + 8 * (e >> 2) + 4 * h) * EW + j * 32 + r] =
                        SPLIT == 4 ? acc[i][j][e] * kF16Unscale : acc[i][j][e];
            __syncthreads();
    #pragma unroll
            for (int pass = 0; pass < 32 / RPP; ++pass) {
                const int row = pass * RPP + lane_id / LPR;
                const int c4 = (lane_id % LPR) * 4;
                const int p = (wm * TM + i) * 32 + row;
                const int oy = e_oy0 + (p >> 4), ox = e_ox0 + (p & 15);
                const int n = wn * TN * 32 + c4;
                if (oy >= g.IH || ox >= g.IW || n >= g.OC) continue;
                const int64_t pix = ((int64_t)e_img * g.IH + oy) * g.IW + ox;
                f32x4 v = *reinterpret_cast<const f32x4*>(&stage[row * EW + c4]);
                float* dst = out + pix * g.ldo + n;
                if (ovec && n + 3 < g.OC) {
                    if (addend) v += *reinterpret_cast<const f32x4*>(addend + pix * ld_add + n);
                    if (addend2) v += *reinterpret_cast<const f32x4*>(addend2 + pix * ld_add2 + n);
                    *reinterpret_cast<f32x4*>(dst) = v;
                } else {
    #pragma unroll
                    for (int q = 0; q < 4; ++q)
                        if (n + q < g.OC) {
                            float o = v[q];
                            if (addend) o += addend[pix * ld_add + n + q];
                            if (addend2) o += addend2[pix * ld_add2 + n + q];
                            dst[q] = o;
                        }
                }
            }
            __syncthreads();
        }
        if (!more) break;
    }
}

// returns -1 when the shape is not covered (caller falls back to the implicit-GEMM kernel)
template <bool FLIP>
static int launch_direct3(const float* in, int64_t ldi, const float* wk, float* out, int64_t ldo, int64_t N, int H,
                          int W, int IC, int OC, int split, const float* addend, int64_t ld_add,
                          const float* addend2, int64_t ld_add2, double* bn_partial, int bn_chunks, hipStream_t st,
                          const char* name) {
    static const bool off = snn_tuning_env("SNN_CONV_NO_DIRECT") != nullptr;  // tuning / bisecting aid
    // measured: a win (12-15 %) for <= 32 output channels; at 64 the implicit-GEMM kernel is as fast or faster
    // (both are bound by LDS operand traffic there), so it stays the default; SNN_CONV_DIRECT_MAX_OC=64 to compare
    static const int max_oc = snn_tuning_env("SNN_CONV_DIRECT_MAX_OC") ? atoi(snn_tuning_env("SNN_CONV_DIRECT_MAX_OC")) : 32;
    if (off || (split != 2 && split != 3 && split != 4) || IC % BK != 0 || OC > max_oc || OC > 64 || OC % 4 != 0) return -1;
    if (ldi % 4 != 0 || !aligned16(in) || !aligned16(wk)) return -1;
    if ((int64_t)H * W * ldi * 4 >= 0x7fffffffLL || (int64_t)OC * 9 * IC * 4 >= 0x7fffffffLL) return -1;
    DirectGeom g;
    g.IH = H; g.IW = W; g.IC = IC; g.OC = OC;
    g.ldi = ldi; g.ldo = ldo;
    g.nimg = (int)N;
    g.pht = (H + DPH - 1) / DPH;
    g.pwt = (W + DPW - 1) / DPW;
    const int64_t tiles = N * g.pht * g.pwt;
    if (tiles > 0x0fffffffLL) return -1;
    g.tiles = (int)tiles;
    g.tiles_per_xcd = (int)snn_ceil_div(tiles, 8);
    g.KtotFull = 9 * IC;
    g.out_vec = (ldo % 4 == 0) && aligned16(out) && (!addend || (ld_add % 4 == 0 && aligned16(addend))) &&
                (!addend2 || (ld_add2 % 4 == 0 && aligned16(addend2)));
    g.bn_partial = bn_partial;
    g.bn_chunks = bn_chunks;
    // persistent: at most (CUs per XCD) x (resident blocks per CU) blocks per XCD
    const int resident = (OC <= 32 && split != 3) ? 3 : 2;
    int nbx = (snn_num_cu() / 8) * resident;
    if (nbx > g.tiles_per_xcd) nbx = g.tiles_per_xcd;
    dim3 grid((unsigned)(nbx * 8));
#define SNN_DIRECT_LAUNCH(BN_, WM_, WN_, SPLIT_, TPS_)                                                            \
    hipLaunchKernelGGL((k_conv_direct3<BN_, WM_, WN_, SPLIT_, TPS_, FLIP>), grid, dim3(kThreads), 0, st, in, wk, out, \
                       g, addend, ld_add, addend2, ld_add2)
    if (OC <= 32) {
        if (split == 3) SNN_DIRECT_LAUNCH(32, 4, 1, 3, 3);
        else if (split == 4) SNN_DIRECT_LAUNCH(32, 4, 1, 4, 3);
        else SNN_DIRECT_LAUNCH(32, 4, 1, 2, 3);
    } else {
        if (split == 3) SNN_DIRECT_LAUNCH(64, 2, 2, 3, 1);
        else if (split == 4) SNN_DIRECT_LAUNCH(64, 2, 2, 4, 3);
        else SNN_DIRECT_LAUNCH(64, 2, 2, 2, 3);
    }
#undef SNN_DIRECT_LAUNCH
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

// ------------------------------------------------------------------------------------------ first layer
// The convolution over the 2-channel event frames (Cin = 2, 3x3: K = 18) does not belong on the matrix pipe
// (SURVEY 8d): its cost is writing y (forward) / reading dy (weight gradient).  Direct kernels: a thread owns 4
// output channels with their 4 x 18 weights (forward) or 4 x 18 gradient accumulators (backward) in registers and
// walks output pixels; the 16 threads of a pixel read the same 9 input positions (one broadcast access each).
// (Skipping the taps whose input is 0 - event frames are sparse - was measured and does not pay: the 18 divergent
// branches per pixel cost more issue slots than the 72 fmaf they save.)
// Arithmetic: fp32 fmaf chain over (kh, kw, ci) in order, for every precision mode.
namespace {
struct FirstGeom {
    int64_t ldx, ldy;
    int rows;  // N * Ho output rows
    int H, W, Ho, Wo, Cout, stride, pad;
    // Rows are dealt to the blocks in groups: group q = blockIdx / group_blocks owns rows [q, q+1) * group_rows and
    // its group_blocks blocks walk them with that stride.  One group (all rows) unless the forward pass also emits
    // BatchNorm partials: then a group is a TIMESTEP and block j of it writes chunk j of partial[t][c][chunk][2].
    int group_rows, group_blocks;
    double* bn_partial;
    // weight gradient with the BatchNorm-backward affine applied on the fly (BNAPPLY): the `dy` operand is gx and
    // dy = A[t][c] * gx + B[t][c] * y + C[t][c], t = image / frames_per_step; coef = [3][T][C]
    const float* bn_y;
    int64_t bn_ldy;
    const float* bn_coef;
    int bn_tc;   // T * C: distance between the three coefficient planes
    int bn_fps;  // frames per timestep
    int rs;      // output rows a block stages (input rows -> LDS) and computes between two barriers: first_layer_rs()
};

// One block walks output ROWS (block-uniform row index: the image / row split and the vertical bounds are scalar
// work), its PP pixel lanes walk the row; per pixel all nine input positions are loaded before any is tested.
// SB (bf16-storage mode): the wide tensors - y (forward), dy / gx and the saved y (weight gradient) - are bf16; the event
// frames x stay fp32.
// (the weight gradient's grid is four blocks per CU - snn_conv2d_wgrad_splitk - so its instances are held to four waves per
// SIMD: the BatchNorm-apply form needed 138 registers, three waves, and ran a quarter of its blocks in a second round)
template <int CIN, int KS, bool WGRAD, bool BNAPPLY = false, bool SB = false>
__global__ __launch_bounds__(kThreads, WGRAD ? 4 : 1) void k_conv_first(const float* __restrict__ x, const float* __restrict__ w,
                                                         const float* __restrict__ dy, float* __restrict__ out,
                                                         FirstGeom g) {
    static_assert(CIN == 2, "float2 input pixels");
    typedef SnnStore<SB> St;
    constexpr int KT = KS * KS * CIN;
    __shared__ float red[WGRAD ? kThreads : 1][KT + 1];
    __shared__ double sred[WGRAD ? 1 : kThreads][9];               // statistics of the forward pass (8 used: odd pitch)
    extern __shared__ __attribute__((aligned(16))) float2 srow[];   // [KS][W + 2 pad] input rows of the current output row
    const int cgs = g.Cout / 4;                       // channel groups: a power of two <= 64
    const int cg = threadIdx.x % cgs, pl = threadIdx.x / cgs, PP = kThreads / cgs;
    float wr[4][KT];                                  // forward: weights; weight gradient: accumulators
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int k = 0; k < KT; ++k) wr[c][k] = WGRAD ? 0.f : w[(cg * 4 + c) * KT + k];
    const int ldx = (int)g.ldx, ldy = (int)g.ldy;
    const int grp = blockIdx.x / g.group_blocks, grp_j = blockIdx.x - grp * g.group_blocks;
    const int r_end = (grp + 1) * g.group_rows < g.rows ? (grp + 1) * g.group_rows : g.rows;
    // BatchNorm partials of the forward pass: a thread sums the <= ceil(Wo / PP) pixels it owns of ONE row in fp32
    // (per-pixel fp64 work cost this kernel 30 %), the rows and everything above in fp64
    // (the fp64 sums live in the thread's own LDS slot: in registers they cost the kernel a wave of occupancy)
    const bool stats = !WGRAD && g.bn_partial != nullptr;
    if (!WGRAD && stats) {
#pragma unroll
        for (int c = 0; c < 8; ++c) sred[threadIdx.x][c] = 0.0;
    }
    // The KS input rows of an output row (with their zero padding) go through LDS: the 16 lanes of a pixel read the same nine
    // positions, and same-address lanes of a global load are separate accesses for the texture addresser.  A block stages
    // the input rows of g.rs of its output rows at once - ONE pair of barriers and ONE exposed memory latency per g.rs rows,
    // and the staging loads of a thread are all requested before the first is written to LDS (four at a time, addresses
    // clamped instead of branched around: the loop used to wait for every single load, six dependent round trips per row).
    const int LW = g.W + 2 * g.pad;
    const int stage_elems = KS * LW;                   // float2 elements of one output row's input rows
    for (int r0 = grp * g.group_rows + grp_j; r0 < r_end; r0 += g.group_blocks * g.rs) {
        int nrows = (r_end - r0 + g.group_blocks - 1) / g.group_blocks;   // block-uniform
        nrows = nrows < g.rs ? nrows : g.rs;
        const int total = nrows * stage_elems;
        __syncthreads();
        for (int e0 = threadIdx.x; e0 < total; e0 += 4 * kThreads) {
            float2 t[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int e = e0 + u * kThreads;
                const int ec = e < total ? e : total - 1;
                const int jk = ec / LW, ixp = ec - jk * LW;       // (row of the stage) * KS + kh, padded column
                const int j = jk / KS, kh = jk - j * KS;
                const int r = r0 + j * g.group_blocks;
                const int img = r / g.Ho, oy = r - img * g.Ho;
                const int iy = oy * g.stride - g.pad + kh, ix = ixp - g.pad;
                const bool ok = (unsigned)iy < (unsigned)g.H && (unsigned)ix < (unsigned)g.W;
                const float2 v = *reinterpret_cast<const float2*>(
                    x + ((int64_t)img * g.H + (ok ? iy : 0)) * g.W * g.ldx + (ok ? ix * ldx : 0));
                t[u] = ok ? v : make_float2(0.f, 0.f);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int e = e0 + u * kThreads;
                if (e < total) srow[e] = t[u];
            }
        }
        __syncthreads();
      for (int jrow = 0; jrow < nrows; ++jrow) {
        const int r = r0 + jrow * g.group_blocks;
        const int img = r / g.Ho;
        const float2* srow_r = srow + jrow * stage_elems;
        const int64_t dyrow = (int64_t)r * g.Wo * g.ldy + cg * 4;   // element index of the row's first pixel in dy / out
        int64_t byrow = 0;
        f32x4 ca = {0.f, 0.f, 0.f, 0.f}, cb = ca, cc = ca;
        if constexpr (WGRAD && BNAPPLY) {   // the row's timestep is block-uniform: three coefficient quads per row
            byrow = (int64_t)r * g.Wo * g.bn_ldy + cg * 4;
            const float* cf = g.bn_coef + (int64_t)(img / g.bn_fps) * g.Cout + cg * 4;
            ca = *reinterpret_cast<const f32x4*>(cf);
            cb = *reinterpret_cast<const f32x4*>(cf + g.bn_tc);
            cc = *reinterpret_cast<const f32x4*>(cf + 2 * (int64_t)g.bn_tc);
        }
        float row_s[4] = {0.f, 0.f, 0.f, 0.f}, row_q[4] = {0.f, 0.f, 0.f, 0.f};
        // weight gradient: the dy (gx, y) quads of a pixel are requested one pixel AHEAD of their use.  With the loads in
        // front of the 72 fmaf that consume them a wave had two 16-byte accesses in flight and then none: 2.2 TB/s for a
        // kernel that runs alone at the end of the backward pass (the step's tail).  The index of the pixel after the
        // row's last is clamped (its quads are loaded and dropped): no branch around the loads.
        [[maybe_unused]] f32x4 gv_next = {0.f, 0.f, 0.f, 0.f}, yv_next = gv_next;
        if (WGRAD && pl < g.Wo) {
            gv_next = St::ld4_last(dy, dyrow + pl * ldy);
            if constexpr (BNAPPLY) yv_next = St::ld4_last(g.bn_y, byrow + pl * (int)g.bn_ldy);
        }
        for (int ox = pl; ox < g.Wo; ox += PP) {
            float2 taps[KS][KS];
#pragma unroll
            for (int kh = 0; kh < KS; ++kh)
#pragma unroll
                for (int kw = 0; kw < KS; ++kw) taps[kh][kw] = srow_r[kh * LW + ox * g.stride + kw];
            f32x4 gv = {0.f, 0.f, 0.f, 0.f};
            if (WGRAD) {
                gv = gv_next;
                const int oxn = ox + PP < g.Wo ? ox + PP : ox;
                gv_next = St::ld4_last(dy, dyrow + oxn * ldy);
                if constexpr (BNAPPLY) {   // the statement of k_bn_bwd_apply (neuron.hip): same roundings
                    const f32x4 yv = yv_next;
                    yv_next = St::ld4_last(g.bn_y, byrow + oxn * (int)g.bn_ldy);
#pragma unroll
                    for (int c = 0; c < 4; ++c) gv[c] = ca[c] * gv[c] + cb[c] * yv[c] + cc[c];
                }
            }
            float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kh = 0; kh < KS; ++kh)
#pragma unroll
                for (int kw = 0; kw < KS; ++kw) {
                    const float2 v = taps[kh][kw];
                    const int k0 = (kh * KS + kw) * CIN;
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        if (WGRAD) {
                            wr[c][k0] = fmaf(gv[c], v.x, wr[c][k0]);
                            wr[c][k0 + 1] = fmaf(gv[c], v.y, wr[c][k0 + 1]);
                        } else {
                            acc[c] = fmaf(v.x, wr[c][k0], acc[c]);
                            acc[c] = fmaf(v.y, wr[c][k0 + 1], acc[c]);
                        }
                    }
                }
            if (!WGRAD) {
                f32x4 o = {acc[0], acc[1], acc[2], acc[3]};
                St::st4(out, dyrow + ox * ldy, o);
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    row_s[c] += acc[c];
                    row_q[c] = fmaf(acc[c], acc[c], row_q[c]);
                }
            }
        }
        if (!WGRAD && stats) {
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                sred[threadIdx.x][c] += (double)row_s[c];
                sred[threadIdx.x][4 + c] += (double)row_q[c];
            }
        }
      }
    }
    if (!WGRAD && stats) {
        // block sum over the PP pixel lanes of every channel, in lane order
        __syncthreads();
        if ((int)threadIdx.x < g.Cout) {
            const int gq = threadIdx.x >> 2, c = threadIdx.x & 3;
            double ss = 0.0, qq = 0.0;
            for (int q = 0; q < PP; ++q) {
                ss += sred[q * cgs + gq][c];
                qq += sred[q * cgs + gq][4 + c];
            }
            double* dst = g.bn_partial + snn_bn_partial_index(grp, grp_j, threadIdx.x, g.group_blocks, g.Cout);
            dst[0] = ss;
            dst[1] = qq;
        }
    }
    if (WGRAD) {
        // block sum over the PP pixel lanes of every channel group, in lane order; out = workspace, one slab
        // [Cout][KT] per block, summed in fixed order by k_wgrad_reduce
        float* slab = out + (int64_t)blockIdx.x * g.Cout * KT;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
#pragma unroll
            for (int k = 0; k < KT; ++k) red[threadIdx.x][k] = wr[c][k];
            __syncthreads();
            for (int e = threadIdx.x; e < cgs * KT; e += kThreads) {
                const int gq = e / KT, k = e - gq * KT;
                float sum = 0.f;
                for (int q = 0; q < PP; ++q) sum += red[q * cgs + gq][k];
                slab[(gq * 4 + c) * KT + k] = sum;
            }
            __syncthreads();
        }
    }
}

// the shapes the two kernels above take: the caller checks alignment of its buffers on top
static bool first_layer_shape(int Cin, int Cout, int KH, int KW) {
    static const bool off = snn_tuning_env("SNN_CONV_NO_FIRST") != nullptr;  // tuning / bisecting aid
    if (off || Cin != 2 || KH != 3 || KW != 3 || Cout % 4 != 0 || Cout > 256) return false;
    const int cgs = Cout / 4;
    return (cgs & (cgs - 1)) == 0;
}

// output rows a block stages at once: as many as fit 20 KiB of LDS - next to the 19 KiB of reduction scratch both forms
// carry, four blocks per CU stay resident (with 29 KiB the weight gradient fell to three and lost a fifth) - at most 4
static int first_layer_rs(int W, int pad) {
    const int per_row = 3 * (W + 2 * pad) * (int)sizeof(float2);
    int rs = (20 << 10) / per_row;
    return rs < 1 ? 1 : (rs > 4 ? 4 : rs);
}
static size_t first_layer_lds(int W, int pad) { return (size_t)first_layer_rs(W, pad) * 3 * (W + 2 * pad) * sizeof(float2); }

static int first_layer_blocks(int64_t rows) {  // grid of the row-walking kernels = slabs of the weight gradient
    int64_t b = rows < 4 * snn_num_cu() ? rows : 4 * snn_num_cu();
    return b < 1 ? 1 : (int)b;
}
}  // namespace

// ---- pre-split weight images (see PRESPLIT of k_conv_gather).  Elementwise over groups of 4 consecutive floats: the
// group's 16 bytes become (4 hi pieces, 4 lo pieces) with exactly the arithmetic of the in-kernel conversion - fp16
// pieces of w * 2^8 (forward, SNN_PREC_FP16X3) or bf16 pieces of w (data gradient, SNN_PREC_BF16X3; apply it to the
// transposed weights).  A weight row (KH*KW*Cin floats) must start on a group boundary.
namespace {
template <bool F16>
__global__ void k_weight_presplit(const f32x4* __restrict__ w, u32x4* __restrict__ out, int64_t groups) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < groups; i += (int64_t)gridDim.x * blockDim.x) {
        const f32x4 v = w[i];
        u32x4 o;
#pragma unroll
        for (int e = 0; e < 4; e += 2) {
            if (F16) {
                const float a = v[e] * kF16WeightScale, b = v[e + 1] * kF16WeightScale;
                const f16x2 ph = __builtin_convertvector(f32x2{a, b}, f16x2);
                const f16x2 pl = __builtin_convertvector(f32x2{a - (float)ph[0], b - (float)ph[1]}, f16x2);
                o[e >> 1] = __builtin_bit_cast(unsigned, ph);
                o[2 + (e >> 1)] = __builtin_bit_cast(unsigned, pl);
            } else {
                f32x2 rest = {v[e], v[e + 1]};
                const bf16x2 ph = __builtin_convertvector(rest, bf16x2);
                const unsigned bits = __builtin_bit_cast(unsigned, ph);
                rest[0] -= __builtin_bit_cast(float, bits << 16);
                rest[1] -= __builtin_bit_cast(float, bits & 0xffff0000u);
                const bf16x2 pl = __builtin_convertvector(rest, bf16x2);
                o[e >> 1] = bits;
                o[2 + (e >> 1)] = __builtin_bit_cast(unsigned, pl);
            }
        }
        out[i] = o;
    }
}
}  // namespace

extern "C" int snn_weight_presplit(const float* w, void* out, int64_t n, int precision, void* stream) {
    SNN_REQUIRE(w && out && n > 0 && n % 4 == 0, "snn_weight_presplit: bad arguments (n = %lld must be a multiple of 4)",
                (long long)n);
    SNN_REQUIRE(aligned16(w) && aligned16(out), "snn_weight_presplit: buffers must be 16-byte aligned");
    SNN_REQUIRE(precision == SNN_PREC_FP16X3 || precision == SNN_PREC_BF16X3,
                "snn_weight_presplit: precision must be SNN_PREC_FP16X3 (forward) or SNN_PREC_BF16X3 (data gradient)");
    const int64_t groups = n / 4;
    int64_t blocks = snn_ceil_div(groups, kThreads);
    if (blocks > 8 * snn_num_cu()) blocks = 8 * snn_num_cu();
    if (precision == SNN_PREC_FP16X3)
        hipLaunchKernelGGL(k_weight_presplit<true>, dim3((unsigned)blocks), dim3(kThreads), 0, (hipStream_t)stream,
                           reinterpret_cast<const f32x4*>(w), reinterpret_cast<u32x4*>(out), groups);
    else
        hipLaunchKernelGGL(k_weight_presplit<false>, dim3((unsigned)blocks), dim3(kThreads), 0, (hipStream_t)stream,
                           reinterpret_cast<const f32x4*>(w), reinterpret_cast<u32x4*>(out), groups);
    SNN_CHECK_LAUNCH("snn_weight_presplit");
    return 0;
}

// Chunk slots per timestep of the three forward kernels' statistics partials (see stat_flush); 0: not produced.
namespace {
struct FirstGroups { int rows, blocks; };
// first-layer kernel, one group of rows per timestep: an equal number of rows for every block of the group
static FirstGroups first_layer_groups(int rows_per_step, int steps) {
    int target = 8 * snn_num_cu() / (steps > 0 ? steps : 1);
    if (target < 1) target = 1;
    if (target > rows_per_step) target = rows_per_step;
    const int per_block = (rows_per_step + target - 1) / target;
    return {rows_per_step, (rows_per_step + per_block - 1) / per_block};
}
static int64_t gather_bn_chunks(int64_t rows_per_step) { return (rows_per_step + BM - 1) / BM + 1; }
static int64_t direct_bn_chunks(int frames_per_step, int Ho, int Wo) {
    return (int64_t)frames_per_step * ((Ho + DPH - 1) / DPH) * ((Wo + DPW - 1) / DPW);
}
}  // namespace

extern "C" size_t snn_conv2d_fwd_bn_partial_size(int64_t N, int frames_per_step, int Ho, int Wo, int Cout) {
    if (N <= 0 || frames_per_step <= 0 || N % frames_per_step != 0 || Ho <= 0 || Wo <= 0 || Cout <= 0) return 0;
    const int64_t T = N / frames_per_step, rows = (int64_t)frames_per_step * Ho * Wo;
    int64_t chunks = gather_bn_chunks(rows);
    const int64_t d = direct_bn_chunks(frames_per_step, Ho, Wo);
    if (d > chunks) chunks = d;
    if ((int64_t)frames_per_step * Ho > chunks) chunks = (int64_t)frames_per_step * Ho;   // first layer: <= one block per row
    const int64_t hc = snn_conv3x3_halo_bn_chunks(frames_per_step, Ho, Wo);                // halo-resident 3x3 (conv_halo.hip)
    if (hc > chunks) chunks = hc;
    return (size_t)(T * chunks * Cout * 2);
}

extern "C" int snn_conv2d_fwd(const float* x, int64_t ldx, const float* w, const void* w_split, float* y, int64_t ldy,
                              int64_t N, int H, int W, int Cin, int Ho, int Wo, int Cout, int KH, int KW, int stride,
                              int pad, const float* addend, int64_t ld_addend, double* bn_partial, int frames_per_step,
                              int* bn_layout, int precision, void* stream) {
    SNN_REQUIRE(x && w && y, "snn_conv2d_fwd: null pointer");
    SNN_REQUIRE(!w_split || precision == SNN_PREC_FP16X3,
                "snn_conv2d_fwd: a pre-split weight image exists for SNN_PREC_FP16X3 only (precision %d)", precision);
    SNN_REQUIRE(precision == SNN_PREC_FP32 || precision == SNN_PREC_BF16X6 || precision == SNN_PREC_FP16X3 ||
                    precision == SNN_PREC_BF16X1 || precision == SNN_PREC_BF16S,
                "snn_conv2d_fwd: precision must be SNN_PREC_FP32, _BF16X6, _FP16X3, _BF16X1 or _BF16S (got %d)", precision);
    const bool sbf = precision == SNN_PREC_BF16S;   // x (but for the fp32 event frames), y, addend are bf16
    const int fwd_split = precision;
    if (check_conv_shape("snn_conv2d_fwd", N, H, W, Cin, Ho, Wo, Cout, KH, KW, stride, pad)) return 1;
    SNN_REQUIRE(ldx >= Cin && ldy >= Cout, "snn_conv2d_fwd: pixel stride smaller than channel count");
    SNN_REQUIRE(!bn_partial || (bn_layout && frames_per_step > 0 && N % frames_per_step == 0),
                "snn_conv2d_fwd: statistics need bn_layout and a frames_per_step that divides N (%lld frames, %d per step)",
                (long long)N, frames_per_step);
    SNN_REQUIRE(!(bn_partial && addend), "snn_conv2d_fwd: statistics are of the convolution itself - no addend with bn_partial");
    if (bn_layout) bn_layout[0] = bn_layout[1] = 0;
    const int64_t step_rows = bn_partial ? (int64_t)frames_per_step * Ho * Wo : 0;
    ConvGeom g;
    g.Mtot = N * Ho * (int64_t)Wo;
    g.IH = H; g.IW = W; g.IC = Cin;
    g.OH = Ho; g.OW = Wo; g.OC = Cout;
    g.KH = KH; g.KW = KW; g.stride = stride; g.pad = pad;
    g.ldi = ldx; g.ldo = ldy;
    g.Ktot = g.KtotFull = KH * KW * Cin;
    g.nimg = (int)N;
    g.ph = g.pw = g.kh0 = g.kw0 = 0; g.nkh = KH; g.nkw = KW; g.OHc = Ho; g.OWc = Wo;
    g.magic_ic = magic_u32(Cin); g.magic_kw = magic_u32(KW);
    g.bn_partial = nullptr; g.bn_rows = 0; g.bn_chunks = 0; g.x_th = 0.0f;
    SNN_REQUIRE(N * (int64_t)H * W < 0x7fffffffLL && (int64_t)g.Ktot * Cin < 0xffffffffLL,
                "snn_conv2d_fwd: tensor too large for 32-bit pixel indexing");
    SNN_REQUIRE(!addend || ld_addend >= Cout, "snn_conv2d_fwd: addend pixel stride smaller than channel count");
    if (first_layer_shape(Cin, Cout, KH, KW) && !addend && ldx % 2 == 0 && aligned8(x) && ldy % 4 == 0 &&
        (sbf ? aligned8(y) : aligned16(y)) &&
        (int64_t)W * ldx < 0x7fffffffLL && (int64_t)Wo * ldy < 0x7fffffffLL && W + 2 * pad <= 1408 &&
        (Wo - 1) * stride + 3 <= W + 2 * pad) {
        FirstGeom fg = {ldx, ldy, (int)(N * Ho), H, W, Ho, Wo, Cout, stride, pad, (int)(N * Ho), 0, nullptr,
                        nullptr, 0, nullptr, 0, 1, first_layer_rs(W, pad)};
        int blocks = fg.rows < 8 * snn_num_cu() ? fg.rows : 8 * snn_num_cu();
        fg.group_blocks = blocks;
        if (bn_partial) {
            const int steps = (int)(N / frames_per_step);
            const FirstGroups fgr = first_layer_groups(frames_per_step * Ho, steps);
            fg.group_rows = fgr.rows;
            fg.group_blocks = fgr.blocks;
            fg.bn_partial = bn_partial;
            blocks = steps * fgr.blocks;
            bn_layout[0] = fgr.blocks;
        }
        if (sbf)
            hipLaunchKernelGGL((k_conv_first<2, 3, false, false, true>), dim3((unsigned)blocks), dim3(kThreads),
                               first_layer_lds(W, pad), (hipStream_t)stream, x, w, nullptr, y, fg);
        else
            hipLaunchKernelGGL((k_conv_first<2, 3, false>), dim3((unsigned)blocks), dim3(kThreads),
                               first_layer_lds(W, pad), (hipStream_t)stream, x, w, nullptr, y, fg);
        SNN_CHECK_LAUNCH("snn_conv2d_fwd");
        return 0;
    }
    SNN_REQUIRE(!sbf || Cin % 32 == 0, "snn_conv2d_fwd: bf16 storage covers the event-frame layer (fp32 frames, Cin = 2, "
                "3x3) and layers with a multiple of 32 input channels (got %d)", Cin);
    if (KH == 3 && KW == 3 && stride == 1 && pad == 1 && !sbf) {
        const int64_t chunks = bn_partial ? direct_bn_chunks(frames_per_step, Ho, Wo) : 0;
        const int rc = launch_direct3<false>(x, ldx, w, y, ldy, N, H, W, Cin, Cout, fwd_split, addend, ld_addend,
                                             nullptr, 0, chunks <= 0x7fffffff ? bn_partial : nullptr, (int)chunks,
                                             (hipStream_t)stream, "snn_conv2d_fwd");
        if (rc >= 0) {
            if (rc == 0 && bn_partial && chunks <= 0x7fffffff) bn_layout[0] = (int)chunks;
            return rc;
        }
    }
    if (bn_partial && step_rows >= BM && gather_bn_chunks(step_rows) <= 0x7fffffff) {
        g.bn_partial = bn_partial;
        g.bn_rows = step_rows;
        g.bn_chunks = (int)gather_bn_chunks(step_rows);
        bn_layout[0] = g.bn_chunks;
        bn_layout[1] = BM;
    }
    if (sbf)
        return launch_gather<false, 5, true>(x, w, nullptr, y, g, addend, ld_addend, nullptr, 0, (hipStream_t)stream, "snn_conv2d_fwd");
    if (fwd_split == 5)
        return launch_gather<false, 5>(x, w, nullptr, y, g, addend, ld_addend, nullptr, 0, (hipStream_t)stream, "snn_conv2d_fwd");
    if (fwd_split == 4)
        return launch_gather<false, 4>(x, w, w_split, y, g, addend, ld_addend, nullptr, 0, (hipStream_t)stream, "snn_conv2d_fwd");
    if (fwd_split == 3)
        return launch_gather<false, 3>(x, w, nullptr, y, g, addend, ld_addend, nullptr, 0, (hipStream_t)stream, "snn_conv2d_fwd");
    return launch_gather<false, 0>(x, w, nullptr, y, g, addend, ld_addend, nullptr, 0, (hipStream_t)stream, "snn_conv2d_fwd");
}

extern "C" int snn_conv2d_dgrad(const float* dy, int64_t lddy, const float* wt, const void* wt_split, float* dx,
                                int64_t lddx, int64_t N, int H, int W, int Cin, int Ho, int Wo, int Cout, int KH, int KW,
                                int stride, int pad, const float* addend, int64_t ld_addend, const float* addend2,
                                int64_t ld_addend2, int precision, void* stream) {
    SNN_REQUIRE(dy && wt && dx, "snn_conv2d_dgrad: null pointer");
    SNN_REQUIRE(!wt_split || precision == SNN_PREC_BF16X3,
                "snn_conv2d_dgrad: a pre-split weight image exists for SNN_PREC_BF16X3 only (precision %d)", precision);
    SNN_REQUIRE(precision == SNN_PREC_FP32 || precision == SNN_PREC_BF16X3 || precision == SNN_PREC_BF16X1 ||
                    precision == SNN_PREC_BF16S,
                "snn_conv2d_dgrad: precision must be SNN_PREC_FP32, _BF16X3, _BF16X1 or _BF16S (got %d)", precision);
    const bool sbf = precision == SNN_PREC_BF16S;   // dy, dx and the addends are bf16
    const int bwd_split = precision;
    SNN_REQUIRE(!addend2 || ld_addend2 >= Cin, "snn_conv2d_dgrad: addend2 pixel stride smaller than channel count");
    if (check_conv_shape("snn_conv2d_dgrad", N, H, W, Cin, Ho, Wo, Cout, KH, KW, stride, pad)) return 1;
    SNN_REQUIRE(lddy >= Cout && lddx >= Cin, "snn_conv2d_dgrad: pixel stride smaller than channel count");
    SNN_REQUIRE(!addend || ld_addend >= Cin, "snn_conv2d_dgrad: addend pixel stride smaller than channel count");
    ConvGeom g;
    g.IH = Ho; g.IW = Wo; g.IC = Cout;  // gathered tensor is dy
    g.OH = H; g.OW = W; g.OC = Cin;     // one GEMM row per INPUT pixel
    g.KH = KH; g.KW = KW; g.stride = stride; g.pad = pad;
    g.ldi = lddy; g.ldo = lddx;
    g.KtotFull = KH * KW * Cout;
    g.nimg = (int)N;
    g.bn_partial = nullptr; g.bn_rows = 0; g.bn_chunks = 0;
    SNN_REQUIRE(N * (int64_t)Ho * Wo < 0x7fffffffLL && (int64_t)g.KtotFull * Cout < 0xffffffffLL,
                "snn_conv2d_dgrad: tensor too large for 32-bit pixel indexing");
    const bool split = bwd_split != 0;
    if (KH == 3 && KW == 3 && stride == 1 && pad == 1 && bwd_split != SNN_PREC_BF16X1 && !sbf) {  // dx = conv(dy, mirrored taps of w^T)
        const int rc = launch_direct3<true>(dy, lddy, wt, dx, lddx, N, H, W, Cout, Cin, split ? 2 : 0, addend,
                                            ld_addend, addend2, ld_addend2, nullptr, 0, (hipStream_t)stream,
                                            "snn_conv2d_dgrad");
        if (rc >= 0) return rc;
    }
    // one launch per stride phase: each class multiplies only the taps that can reach it
    for (int ph = 0; ph < stride && ph < H; ++ph)
        for (int pw = 0; pw < stride && pw < W; ++pw) {
            g.ph = ph; g.pw = pw;
            g.kh0 = (ph + pad) % stride; g.kw0 = (pw + pad) % stride;
            g.nkh = g.kh0 < KH ? (KH - g.kh0 + stride - 1) / stride : 0;
            g.nkw = g.kw0 < KW ? (KW - g.kw0 + stride - 1) / stride : 0;
            g.OHc = (H - ph + stride - 1) / stride;
            g.OWc = (W - pw + stride - 1) / stride;
            g.Mtot = N * g.OHc * (int64_t)g.OWc;
            g.Ktot = g.nkh * g.nkw * Cout;
            g.magic_ic = magic_u32(Cout); g.magic_kw = magic_u32(g.nkw);
            int rc = sbf ? launch_gather<true, 5, true>(dy, wt, nullptr, dx, g, addend, ld_addend, addend2, ld_addend2,
                                                        (hipStream_t)stream, "snn_conv2d_dgrad")
                     : bwd_split == SNN_PREC_BF16X1
                         ? launch_gather<true, 5>(dy, wt, nullptr, dx, g, addend, ld_addend, addend2, ld_addend2,
                                                  (hipStream_t)stream, "snn_conv2d_dgrad")
                         : (split ? launch_gather<true, 2>(dy, wt, wt_split, dx, g, addend, ld_addend, addend2, ld_addend2,
                                                           (hipStream_t)stream, "snn_conv2d_dgrad")
                                  : launch_gather<true, 0>(dy, wt, nullptr, dx, g, addend, ld_addend, addend2, ld_addend2,
                                                           (hipStream_t)stream, "snn_conv2d_dgrad"));
            if (rc) return rc;
        }
    return 0;
}

namespace {
struct WgradTile { int bm, bn, id, blocks_per_cu; };
// candidate block tiles (out-channels x (tap,ci) columns); pick the one that wastes the least MFMA work on
// padding, larger tiles first on ties (fewer LDS / L2 bytes per FLOP)
static WgradTile wgrad_tile(int Cout, int Ktot, bool split, int64_t M) {
    // blocks_per_cu: residency of each variant (registers / LDS), used to size the pixel split to ONE full wave
    static const WgradTile cand[] = {{128, 128, 0, 3}, {64, 256, 1, 3}, {32, 256, 2, 4},
                                     {128, 64, 3, 3},  {64, 64, 4, 3},  {32, 128, 5, 3}};
    if (const char* force = snn_tuning_env("SNN_WGRAD_TILE")) {  // tuning aid
        int id = atoi(force);
        if (id >= 0 && id < 6) return cand[id];
    }
    WgradTile best = cand[0];
    double best_eff = -1.0;
    for (const WgradTile& c : cand) {
        double padded = (double)(snn_ceil_div(Cout, c.bm) * c.bm) * (double)(snn_ceil_div(Ktot, c.bn) * c.bn);
        double eff = (double)Cout * Ktot / padded;
        // with the bf16x3 MFMAs (5x cheaper) the per-stage overhead dominates: favour the 128 x 128 tile (measured;
        // 64 -> 64 3x3 is faster on nine 64 x 64 tiles than on three 64 x 256 ones since the loader is coalesced)
        // On very long pixel ranges (the 304x240 T=128 backbone: 18.7 M pixels) the 64 x 256 tile wins again - x is
        // then re-read from HBM once per column tile, 3 instead of 9 times (6.9 vs 8.5 ms).
        if (split && (c.id == 0 || (c.id == 1 && M > 4000000))) eff *= 1.4;
        if (eff > best_eff + 1e-9) {
            best_eff = eff;
            best = c;
        }
    }
    return best;
}
}  // namespace

extern "C" int snn_conv2d_wgrad_splitk(int64_t N, int H, int W, int Cin, int Ho, int Wo, int Cout, int KH, int KW,
                                       int stride, int pad, int precision) {
    if (N <= 0 || Ho <= 0 || Wo <= 0 || Cin <= 0 || Cout <= 0 || KH <= 0 || KW <= 0) return 1;
    const int bwd_split = precision == SNN_PREC_FP32 ? 0 : 1;   // SNN_PREC_BF16S plans like the other 16-bit modes
    if (bwd_split) {  // 3x3 layers with whole 32-channel tiles: the halo-resident kernel (wgrad_halo.hip)
        const SnnWgradHaloPlan hp = snn_wgrad_halo_plan(N, H, W, Cin, Ho, Wo, Cout, KH, KW, stride, pad);
        if (hp.ok) return hp.slabs;
    }
    const int64_t M = N * Ho * (int64_t)Wo;
    const int64_t Ktot = (int64_t)KH * KW * Cin;
    if (first_layer_shape(Cin, Cout, KH, KW)) return first_layer_blocks(N * Ho);  // one slab per block
    const bool split_mode = bwd_split && Cin % 4 == 0 && Cout % 4 == 0;
    const WgradTile t = wgrad_tile(Cout, (int)Ktot, split_mode, M);
    const int64_t tiles = snn_ceil_div(Cout, t.bm) * snn_ceil_div(Ktot, t.bn);
    // all blocks resident at once (a second, nearly empty wave of equal-length blocks would double the time);
    // residency of the bf16x3 (pipelined) variants by registers / LDS
    // (tools/wgrad_sweep.py over the layer shapes of TinyYolo GEN1, residency 2..4 per variant)
    static const int split_resident[6] = {3, 2, 2, 3, 3, 3};
    int resident = split_mode ? split_resident[t.id] : t.blocks_per_cu;
    if (split_mode && t.id == 4) resident = KH * KW > 1 ? 4 : 2;
    if (split_mode && t.id == 0) {
        // three blocks per CU only while a split keeps >= 24 stages of 32 pixels; shorter splits are all prologue
        const int64_t s3 = (3 * (int64_t)snn_num_cu()) / tiles;
        const int64_t s3r = s3 >= 32 ? s3 / 8 * 8 : (s3 < 1 ? 1 : s3);
        if (M / s3r < 24 * WB_K) resident = 2;
    }
    if (const char* force = snn_tuning_env("SNN_WGRAD_RESIDENT")) resident = atoi(force) > 0 ? atoi(force) : resident;  // tuning aid
    int64_t s = ((int64_t)resident * snn_num_cu()) / tiles;
    const int64_t max_by_work = snn_ceil_div(M, 8 * WB_K);            // >= 8 LDS stages per block
    const int64_t max_by_mem = (int64_t)(64 << 20) / (Cout * Ktot);   // workspace <= 256 MiB
    if (s > max_by_work) s = max_by_work;
    if (s > max_by_mem) s = max_by_mem;
    if (s >= 32) s = s / 8 * 8;  // whole groups of 8 splits: one split per XCD at a time (XCD-aware mapping)
    if (s > 32768) s = 32768;
    if (s < 1) s = 1;
    return (int)s;
}

// which kernel snn_conv2d_wgrad launches for a shape: 0 the implicit GEMM (k_conv_wgrad_pipe / k_conv_wgrad), 1 the
// halo-resident kernel (k_conv_wgrad_halo), 2 the event-frame row kernel (k_conv_first) - for measurement labels; host-only
extern "C" int snn_conv2d_wgrad_kernel(int64_t N, int H, int W, int Cin, int Ho, int Wo, int Cout, int KH, int KW, int stride,
                                       int pad, int precision) {
    if (N <= 0 || Ho <= 0 || Wo <= 0 || Cin <= 0 || Cout <= 0 || KH <= 0 || KW <= 0) return 0;
    if (first_layer_shape(Cin, Cout, KH, KW)) return 2;
    if (precision != SNN_PREC_FP32 && snn_wgrad_halo_plan(N, H, W, Cin, Ho, Wo, Cout, KH, KW, stride, pad).ok) return 1;
    return 0;
}

// dw (+)= sum over the splitk workspace slabs, fixed order
static int wgrad_reduce_slabs(float* workspace, float* dw, int64_t n, int splitk, int accumulate, hipStream_t st) {
    if (n % 4 == 0 && aligned16(workspace) && aligned16(dw) && splitk > 1) {
        // one launch when the rows a wave has to walk stay short: blocks of 256 elements, KG waves sharing the slab rows
        const int64_t nb1 = snn_ceil_div(n, 256);
        int kg = 1;
        while (kg < 16 && nb1 * kg < 8 * (int64_t)snn_num_cu() && splitk / (2 * kg) >= 4) kg *= 2;
        if (snn_ceil_div(splitk, kg) <= 48) {
#define SNN_REDUCE_ONCE(KG_)                                                                                   \
    hipLaunchKernelGGL((k_wgrad_reduce_once<KG_>), dim3((unsigned)nb1), dim3(64 * KG_), 0, st, workspace, n, splitk, dw, \
                       accumulate)
            switch (kg) {
                case 1: SNN_REDUCE_ONCE(1); break;
                case 2: SNN_REDUCE_ONCE(2); break;
                case 4: SNN_REDUCE_ONCE(4); break;
                case 8: SNN_REDUCE_ONCE(8); break;
                default: SNN_REDUCE_ONCE(16); break;
            }
#undef SNN_REDUCE_ONCE
            SNN_CHECK_LAUNCH("snn_conv2d_wgrad_reduce");
            return 0;
        }
    }
    if (n % 4 == 0 && aligned16(workspace) && aligned16(dw) && splitk > 8) {
        const int64_t nb = snn_ceil_div(n / 4, kThreads);
        int64_t groups = snn_ceil_div(4 * snn_num_cu(), nb);   // ~4 blocks per CU in the first pass
        if (groups > splitk / 4) groups = splitk / 4;          // at least 4 rows per group
        if (groups < 1) groups = 1;
        const int per = (int)snn_ceil_div(splitk, groups);
        groups = snn_ceil_div(splitk, per);
        if (groups > 1) {
            hipLaunchKernelGGL(k_wgrad_reduce4, dim3((unsigned)nb, (unsigned)groups), dim3(kThreads), 0, st, workspace,
                               n, splitk, per, n, workspace, (int64_t)per * n, 0);
            hipLaunchKernelGGL(k_wgrad_reduce4, dim3((unsigned)nb, 1), dim3(kThreads), 0, st, workspace, n,
                               (int)groups, (int)groups, (int64_t)per * n, dw, 0, accumulate);
        } else {
            hipLaunchKernelGGL(k_wgrad_reduce4, dim3((unsigned)nb, 1), dim3(kThreads), 0, st, workspace, n, splitk,
                               splitk, n, dw, 0, accumulate);
        }
        SNN_CHECK_LAUNCH("snn_conv2d_wgrad_reduce");
        return 0;
    }
#define SNN_REDUCE_LAUNCH(KG_)                                                                                  \
    hipLaunchKernelGGL((k_wgrad_reduce<KG_>), dim3((unsigned)snn_ceil_div(n, kThreads / KG_)), dim3(kThreads), 0, \
                       st, workspace, dw, n, splitk, accumulate)
    if (splitk <= 8) SNN_REDUCE_LAUNCH(1);
    else if (splitk <= 64) SNN_REDUCE_LAUNCH(4);
    else if (splitk <= 256) SNN_REDUCE_LAUNCH(16);
    else SNN_REDUCE_LAUNCH(64);
#undef SNN_REDUCE_LAUNCH
    SNN_CHECK_LAUNCH("snn_conv2d_wgrad_reduce");
    return 0;
}

namespace {
static bool first_layer_wgrad_ok(const float* x, int64_t ldx, const float* dy, int64_t lddy, int64_t N, int H, int W, int Cin,
                                 int Ho, int Wo, int Cout, int KH, int KW, int stride, int pad, bool dy_bf16 = false) {
    return first_layer_shape(Cin, Cout, KH, KW) && ldx % 2 == 0 && aligned8(x) && lddy % 4 == 0 &&
           (dy_bf16 ? aligned8(dy) : aligned16(dy)) &&
           (int64_t)W * ldx < 0x7fffffffLL && N * Ho < 0x7fffffffLL && W + 2 * pad <= 1408 &&
           (Wo - 1) * stride + 3 <= W + 2 * pad;
}
}  // namespace

extern "C" int snn_conv2d_wgrad_bn_supported(int64_t N, int H, int W, int Cin, int Ho, int Wo, int Cout, int KH, int KW,
                                             int stride, int pad) {
    // the event-frame layer's row kernel (pointer alignment is checked by the call itself)
    return (N > 0 && first_layer_shape(Cin, Cout, KH, KW) && N * (int64_t)Ho < 0x7fffffffLL && W + 2 * pad <= 1408 &&
            (Wo - 1) * stride + 3 <= W + 2 * pad) ? 1 : 0;
}

extern "C" int snn_conv2d_wgrad_bn(const float* x, int64_t ldx, const float* gx, int64_t ldgx, const float* y, int64_t ldy,
                                   const float* coef, int T, int frames_per_step, float* dw, int64_t N, int H, int W,
                                   int Cin, int Ho, int Wo, int Cout, int KH, int KW, int stride, int pad, int accumulate,
                                   float* workspace, int splitk, void* stream) {
    SNN_REQUIRE(x && gx && y && coef && dw && workspace, "snn_conv2d_wgrad_bn: null pointer");
    if (check_conv_shape("snn_conv2d_wgrad_bn", N, H, W, Cin, Ho, Wo, Cout, KH, KW, stride, pad)) return 1;
    SNN_REQUIRE(frames_per_step > 0 && T > 0 && (int64_t)T * frames_per_step == N,
                "snn_conv2d_wgrad_bn: %lld frames are not %d timesteps of %d", (long long)N, T, frames_per_step);
    SNN_REQUIRE(ldx >= Cin && ldgx >= Cout && ldy >= Cout, "snn_conv2d_wgrad_bn: pixel stride smaller than channel count");
    SNN_REQUIRE(splitk >= 1 && splitk <= 32768, "snn_conv2d_wgrad_bn: bad splitk %d", splitk);
    SNN_REQUIRE(first_layer_wgrad_ok(x, ldx, gx, ldgx, N, H, W, Cin, Ho, Wo, Cout, KH, KW, stride, pad) && ldy % 4 == 0 &&
                    aligned16(y) && aligned16(coef) && (int64_t)Wo * ldy < 0x7fffffffLL && (int64_t)Wo * ldgx < 0x7fffffffLL,
                "snn_conv2d_wgrad_bn: shape / alignment not covered (ask snn_conv2d_wgrad_bn_supported)");
    FirstGeom fg = {ldx, ldgx, (int)(N * Ho), H, W, Ho, Wo, Cout, stride, pad, (int)(N * Ho), splitk, nullptr,
                    y, ldy, coef, T * Cout, frames_per_step, first_layer_rs(W, pad)};
    hipLaunchKernelGGL((k_conv_first<2, 3, true, true>), dim3((unsigned)splitk), dim3(kThreads),
                       first_layer_lds(W, pad), (hipStream_t)stream, x, nullptr, gx, workspace, fg);
    SNN_CHECK_LAUNCH("snn_conv2d_wgrad_bn");
    return wgrad_reduce_slabs(workspace, dw, (int64_t)Cout * KH * KW * Cin, splitk, accumulate, (hipStream_t)stream);
}

// xsp: x holds saved LIF potentials, the operand is z = (x > x_th) (snn_conv1x1_spikes_wgrad; pipelined bf16 x 3 kernel only)
static int wgrad_common(const float* x, int64_t ldx, const float* dy, int64_t lddy, float* dw, int64_t N,
                        int H, int W, int Cin, int Ho, int Wo, int Cout, int KH, int KW, int stride, int pad,
                        int accumulate, float* workspace, int splitk, int precision, void* stream, bool xsp, float x_th) {
    SNN_REQUIRE(x && dy && dw && workspace, "snn_conv2d_wgrad: null pointer");
    SNN_REQUIRE(precision == SNN_PREC_FP32 || precision == SNN_PREC_BF16X3 || precision == SNN_PREC_BF16X1 ||
                    precision == SNN_PREC_BF16S,
                "snn_conv2d_wgrad: precision must be SNN_PREC_FP32, _BF16X3, _BF16X1 or _BF16S (got %d)", precision);
    const int bwd_split = precision;
    const bool sbf = precision == SNN_PREC_BF16S;   // x (but for the fp32 event frames) and dy are bf16
    if (check_conv_shape("snn_conv2d_wgrad", N, H, W, Cin, Ho, Wo, Cout, KH, KW, stride, pad)) return 1;
    SNN_REQUIRE(ldx >= Cin && lddy >= Cout, "snn_conv2d_wgrad: pixel stride smaller than channel count");
    SNN_REQUIRE(splitk >= 1 && splitk <= 32768, "snn_conv2d_wgrad: bad splitk %d", splitk);
    SNN_REQUIRE(N * (int64_t)H * W < 0x7fffffffLL, "snn_conv2d_wgrad: more than 2^31 input pixels");
    WgradGeom g;
    g.Mtot = N * Ho * (int64_t)Wo;
    g.H = H; g.W = W; g.Cin = Cin; g.Ho = Ho; g.Wo = Wo; g.Cout = Cout;
    g.KH = KH; g.KW = KW; g.stride = stride; g.pad = pad;
    g.ldx = ldx; g.lddy = lddy;
    g.Ktot = KH * KW * Cin;
    g.x_th = x_th;
    if (!xsp && first_layer_wgrad_ok(x, ldx, dy, lddy, N, H, W, Cin, Ho, Wo, Cout, KH, KW, stride, pad, sbf)) {
        FirstGeom fg = {ldx, lddy, (int)(N * Ho), H, W, Ho, Wo, Cout, stride, pad, (int)(N * Ho), splitk, nullptr,
                        nullptr, 0, nullptr, 0, 1, first_layer_rs(W, pad)};
        if (sbf)
            hipLaunchKernelGGL((k_conv_first<2, 3, true, false, true>), dim3((unsigned)splitk), dim3(kThreads),
                               first_layer_lds(W, pad), (hipStream_t)stream, x, nullptr, dy, workspace, fg);
        else
            hipLaunchKernelGGL((k_conv_first<2, 3, true>), dim3((unsigned)splitk), dim3(kThreads),
                               first_layer_lds(W, pad), (hipStream_t)stream, x, nullptr, dy, workspace, fg);
        SNN_CHECK_LAUNCH("snn_conv2d_wgrad");
        return wgrad_reduce_slabs(workspace, dw, (int64_t)Cout * g.Ktot, splitk, accumulate, (hipStream_t)stream);
    }
    if (bwd_split) {
        const SnnWgradHaloPlan hp = snn_wgrad_halo_plan(N, H, W, Cin, Ho, Wo, Cout, KH, KW, stride, pad);
        if (hp.ok) {
            SNN_REQUIRE(splitk == hp.slabs, "snn_conv2d_wgrad: splitk %d, expected %d (snn_conv2d_wgrad_splitk)", splitk,
                        hp.slabs);
            const int rc = snn_wgrad_halo_launch(hp, x, ldx, dy, lddy, workspace, N, H, W, Cin, Ho, Wo, Cout, stride,
                                                 xsp ? 2 : ((precision == SNN_PREC_BF16X1 || sbf) ? 1 : 3), sbf,
                                                 (hipStream_t)stream, x_th);
            if (rc == 0)
                return wgrad_reduce_slabs(workspace, dw, (int64_t)Cout * g.Ktot, hp.slabs, accumulate,
                                          (hipStream_t)stream);
            if (rc > 0) return rc;
            // rc < 0: buffers this kernel cannot address (unaligned / > 2 GiB per image): the implicit-GEMM kernel
        }
    }
    const bool vec = (Cin % 4 == 0) && (Cout % 4 == 0) && (ldx % 4 == 0) && (lddy % 4 == 0) &&
                     (sbf ? aligned8(x) && aligned8(dy) : aligned16(x) && aligned16(dy));
    const WgradTile t = wgrad_tile(Cout, g.Ktot, bwd_split && Cin % 4 == 0 && Cout % 4 == 0, g.Mtot);
    // small tiles (64 x 64, 32 x 128) run 64-pixel stages in the pipelined kernel (latency cover), the others 32
    static const int wbk_small = snn_tuning_env("SNN_WGRAD_WBK") ? atoi(snn_tuning_env("SNN_WGRAD_WBK")) : 64;  // tuning aid
    const int wbk = (t.id >= 4 && wbk_small == 64) ? 64 : 32;
    g.pix_per_split = snn_ceil_div(snn_ceil_div(g.Mtot, splitk), wbk) * wbk;
    g.nimg = (int)N;
    // pipelined kernel: 32-bit byte offsets relative to the first image of a pixel split
    const int64_t span_pix = g.pix_per_split * (int64_t)stride * stride + 3 * (int64_t)H * W;
    static const bool no_pipe = snn_tuning_env("SNN_WGRAD_NO_PIPE") != nullptr;  // tuning / bisecting aid
    const bool pipe = vec && bwd_split && !no_pipe && g.Mtot < 0x7fffffffLL && span_pix * ldx * 4 < 0x7fffffffLL &&
                      g.pix_per_split * lddy * 4 < 0x7fffffffLL && (int64_t)H * W * ldx * 4 < 0x7fffffffLL;
    const bool one = precision == SNN_PREC_BF16X1 || sbf;
    SNN_REQUIRE(!sbf || pipe, "snn_conv2d_wgrad: bf16 storage covers the event-frame layer and the pipelined kernels only "
                "(channels and strides multiples of 4, 8-byte aligned tensors, < 2 GiB per pixel split)");
    SNN_REQUIRE(!xsp || (pipe && !one), "snn_conv1x1_spikes_wgrad: covers the pipelined bf16 x 3 kernel only (channels and "
                "strides multiples of 4, 16-byte aligned tensors, < 2 GiB per pixel split)");
    g.tiles_m = (int)snn_ceil_div(Cout, t.bm);
    g.tiles_n = (int)snn_ceil_div(g.Ktot, t.bn);
    g.splitk = splitk;
    const int64_t nblocks = (int64_t)g.tiles_m * g.tiles_n * splitk;
    SNN_REQUIRE(nblocks <= 0x7fffffff, "snn_conv2d_wgrad: grid too large");
    dim3 grid((unsigned)nblocks);
    hipStream_t st = (hipStream_t)stream;
#define SNN_WGRAD_LAUNCH(TM_, TN_, WM_, WN_)                                                                   \
    do {                                                                                                       \
        if (xsp && wbk == 64)                                                                                  \
            hipLaunchKernelGGL((k_conv_wgrad_pipe<TM_, TN_, WM_, WN_, 64, false, false, true>), grid, dim3(kThreads), 0, st, x, \
                               dy, workspace, g);                                                              \
        else if (xsp)                                                                                          \
            hipLaunchKernelGGL((k_conv_wgrad_pipe<TM_, TN_, WM_, WN_, 32, false, false, true>), grid, dim3(kThreads), 0, st, x, \
                               dy, workspace, g);                                                              \
        else if (sbf && wbk == 64)                                                                             \
            hipLaunchKernelGGL((k_conv_wgrad_pipe<TM_, TN_, WM_, WN_, 64, true, true>), grid, dim3(kThreads), 0, st, x, dy, \
                               workspace, g);                                                                  \
        else if (sbf)                                                                                          \
            hipLaunchKernelGGL((k_conv_wgrad_pipe<TM_, TN_, WM_, WN_, 32, true, true>), grid, dim3(kThreads), 0, st, x, dy, \
                               workspace, g);                                                                  \
        else if (pipe && one && wbk == 64)                                                                     \
            hipLaunchKernelGGL((k_conv_wgrad_pipe<TM_, TN_, WM_, WN_, 64, true>), grid, dim3(kThreads), 0, st, x, dy, \
                               workspace, g);                                                                  \
        else if (pipe && one)                                                                                  \
            hipLaunchKernelGGL((k_conv_wgrad_pipe<TM_, TN_, WM_, WN_, 32, true>), grid, dim3(kThreads), 0, st, x, dy, \
                               workspace, g);                                                                  \
        else if (pipe && wbk == 64)                                                                            \
            hipLaunchKernelGGL((k_conv_wgrad_pipe<TM_, TN_, WM_, WN_, 64, false>), grid, dim3(kThreads), 0, st, x, dy, \
                               workspace, g);                                                                  \
        else if (pipe)                                                                                         \
            hipLaunchKernelGGL((k_conv_wgrad_pipe<TM_, TN_, WM_, WN_, 32, false>), grid, dim3(kThreads), 0, st, x, dy, \
                               workspace, g);                                                                  \
        else if (vec) hipLaunchKernelGGL((k_conv_wgrad<TM_, TN_, WM_, WN_, true>), grid, dim3(kThreads), 0, st, x,  \
                                    dy, workspace, g);                                                         \
        else hipLaunchKernelGGL((k_conv_wgrad<TM_, TN_, WM_, WN_, false>), grid, dim3(kThreads), 0, st, x, dy, \
                                workspace, g);                                                                 \
    } while (0)
    switch (t.id) {
        case 0: SNN_WGRAD_LAUNCH(2, 2, 2, 2); break;   // 128 x 128
        case 1: SNN_WGRAD_LAUNCH(2, 2, 1, 4); break;   //  64 x 256
        case 2: SNN_WGRAD_LAUNCH(1, 2, 1, 4); break;   //  32 x 256
        case 3: SNN_WGRAD_LAUNCH(2, 1, 2, 2); break;   // 128 x  64
        case 4: SNN_WGRAD_LAUNCH(1, 1, 2, 2); break;   //  64 x  64
        default: SNN_WGRAD_LAUNCH(1, 1, 1, 4); break;  //  32 x 128
    }
#undef SNN_WGRAD_LAUNCH
    SNN_CHECK_LAUNCH("snn_conv2d_wgrad");
    return wgrad_reduce_slabs(workspace, dw, (int64_t)Cout * g.Ktot, splitk, accumulate, st);
}

extern "C" int snn_conv2d_wgrad(const float* x, int64_t ldx, const float* dy, int64_t lddy, float* dw, int64_t N,
                                int H, int W, int Cin, int Ho, int Wo, int Cout, int KH, int KW, int stride, int pad,
                                int accumulate, float* workspace, int splitk, int precision, void* stream) {
    return wgrad_common(x, ldx, dy, lddy, dw, N, H, W, Cin, Ho, Wo, Cout, KH, KW, stride, pad, accumulate, workspace, splitk,
                        precision, stream, false, 0.0f);
}

// ---- convolutions over spikes that were never stored (see k_conv_gather XSP, include/snn_hip.h)
static bool spikes_shape_ok(int64_t N, int H, int W, int Cin, int Ho, int Wo, int Cout, int KH, int KW, int stride, int pad,
                            int64_t ld) {
    // the pipelined implicit GEMM (forward FAST path, pipelined weight gradient); 3x3 / stride 1 layers the halo-resident
    // kernels cover take those instead (snn_conv3x3_halo_spikes, k_conv_wgrad_halo NPROD 2)
    return N > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0 && KH >= 1 && KW >= 1 && KH <= 5 && KW <= 5 && stride >= 1 &&
           Ho == (H + 2 * pad - KH) / stride + 1 && Wo == (W + 2 * pad - KW) / stride + 1 && Ho > 0 && Wo > 0 &&
           Cin % 32 == 0 && Cout % 4 == 0 && ld % 4 == 0 && ld >= Cin && N * (int64_t)H * W < 0x7fffffffLL &&
           N * (int64_t)Ho * Wo < 0x7fffffffLL && (int64_t)H * W * ld * 16 < 0x7fffffffLL;
}

extern "C" int snn_conv2d_spikes_supported(int64_t N, int H, int W, int Cin, int Ho, int Wo, int Cout, int KH, int KW,
                                           int stride, int pad, int64_t ld, int fwd_precision, int bwd_precision) {
    return (fwd_precision == SNN_PREC_FP16X3 && bwd_precision == SNN_PREC_BF16X3 &&
            spikes_shape_ok(N, H, W, Cin, Ho, Wo, Cout, KH, KW, stride, pad, ld)) ? 1 : 0;
}

extern "C" int snn_conv1x1_spikes_supported(int64_t N, int H, int W, int Cin, int Cout, int64_t ld, int fwd_precision,
                                            int bwd_precision) {
    return snn_conv2d_spikes_supported(N, H, W, Cin, H, W, Cout, 1, 1, 1, 0, ld, fwd_precision, bwd_precision);
}

extern "C" int snn_conv2d_spikes_fwd(const float* vdec, int64_t ld, float v_th, const float* w, float* y, int64_t ldy,
                                     int64_t N, int H, int W, int Cin, int Ho, int Wo, int Cout, int KH, int KW, int stride,
                                     int pad, double* bn_partial, int frames_per_step, int* bn_layout, void* stream) {
    SNN_REQUIRE(vdec && w && y, "snn_conv2d_spikes_fwd: null pointer");
    SNN_REQUIRE(v_th >= 0.0f, "snn_conv2d_spikes_fwd: a negative threshold would turn padding into spikes");
    if (check_conv_shape("snn_conv2d_spikes_fwd", N, H, W, Cin, Ho, Wo, Cout, KH, KW, stride, pad)) return 1;
    SNN_REQUIRE(spikes_shape_ok(N, H, W, Cin, Ho, Wo, Cout, KH, KW, stride, pad, ld) && ldy >= Cout,
                "snn_conv2d_spikes_fwd: shape not covered (ask snn_conv2d_spikes_supported)");
    SNN_REQUIRE(!bn_partial || (bn_layout && frames_per_step > 0 && N % frames_per_step == 0),
                "snn_conv2d_spikes_fwd: statistics need bn_layout and a frames_per_step that divides N");
    if (bn_layout) bn_layout[0] = bn_layout[1] = 0;
    ConvGeom g;
    g.Mtot = N * Ho * (int64_t)Wo;
    g.IH = H; g.IW = W; g.IC = Cin;
    g.OH = Ho; g.OW = Wo; g.OC = Cout;
    g.KH = KH; g.KW = KW; g.stride = stride; g.pad = pad;
    g.ldi = ld; g.ldo = ldy;
    g.Ktot = g.KtotFull = KH * KW * Cin;
    g.nimg = (int)N;
    g.ph = g.pw = g.kh0 = g.kw0 = 0; g.nkh = KH; g.nkw = KW; g.OHc = Ho; g.OWc = Wo;
    g.magic_ic = magic_u32(Cin); g.magic_kw = magic_u32(KW);
    g.bn_partial = nullptr; g.bn_rows = 0; g.bn_chunks = 0;
    g.x_th = v_th;
    const int64_t step_rows = bn_partial ? (int64_t)frames_per_step * Ho * Wo : 0;
    if (bn_partial && step_rows >= BM && gather_bn_chunks(step_rows) <= 0x7fffffff) {
        g.bn_partial = bn_partial;
        g.bn_rows = step_rows;
        g.bn_chunks = (int)gather_bn_chunks(step_rows);
        bn_layout[0] = g.bn_chunks;
        bn_layout[1] = BM;
    }
    return launch_gather<false, 4, false, true>(vdec, w, nullptr, y, g, nullptr, 0, nullptr, 0, (hipStream_t)stream,
                                                "snn_conv2d_spikes_fwd");
}

extern "C" int snn_conv1x1_spikes_fwd(const float* vdec, int64_t ld, float v_th, const float* w, float* y, int64_t ldy,
                                      int64_t N, int H, int W, int Cin, int Cout, void* stream) {
    return snn_conv2d_spikes_fwd(vdec, ld, v_th, w, y, ldy, N, H, W, Cin, H, W, Cout, 1, 1, 1, 0, nullptr, 0, nullptr, stream);
}

extern "C" int snn_conv2d_spikes_wgrad(const float* vdec, int64_t ld, float v_th, const float* dy, int64_t lddy, float* dw,
                                       int64_t N, int H, int W, int Cin, int Ho, int Wo, int Cout, int KH, int KW, int stride,
                                       int pad, int accumulate, float* workspace, int splitk, void* stream) {
    SNN_REQUIRE(v_th >= 0.0f, "snn_conv2d_spikes_wgrad: a negative threshold would turn padding into spikes");
    SNN_REQUIRE(spikes_shape_ok(N, H, W, Cin, Ho, Wo, Cout, KH, KW, stride, pad, ld),
                "snn_conv2d_spikes_wgrad: shape not covered (ask snn_conv2d_spikes_supported)");
    return wgrad_common(vdec, ld, dy, lddy, dw, N, H, W, Cin, Ho, Wo, Cout, KH, KW, stride, pad, accumulate, workspace, splitk,
                        SNN_PREC_BF16X3, stream, true, v_th);
}

extern "C" int snn_conv1x1_spikes_wgrad(const float* vdec, int64_t ld, float v_th, const float* dy, int64_t lddy, float* dw,
                                        int64_t N, int H, int W, int Cin, int Cout, int accumulate, float* workspace,
                                        int splitk, void* stream) {
    return snn_conv2d_spikes_wgrad(vdec, ld, v_th, dy, lddy, dw, N, H, W, Cin, H, W, Cout, 1, 1, 1, 0, accumulate, workspace,
                                   splitk, stream);
}
