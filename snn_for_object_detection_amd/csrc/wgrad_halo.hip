// Halo-resident weight gradient of the 3x3 convolutions (stride 1 and 2, pad 1), bf16 x 3 split products.
//
//     dw[co][kh][kw][ci] = sum_{n,oy,ox} dy[n,oy,ox,co] * x[n, oy*s-1+kh, ox*s-1+kw, ci]
//
// Replaces ATen's conv backward-weight behind nn.Conv2d (reference models/modules/layer_gen.py:129-136) for the
// layer-major schedule; called from snn_conv2d_wgrad (conv.hip), results go through the same ordered slab reduction.
//
// Why a second kernel.  The implicit-GEMM weight gradient (k_conv_wgrad_pipe) gathers one shifted copy of x PER TAP:
// every x element is fetched through L1 and split into its bf16 pieces nine times, and the dy tile is re-converted
// for every (tap, ci) column tile - PMC (profiles/r02_*): matrix pipe 22 % busy, VALU and the texture addresser are
// what the kernel waits for.  Here the operands are staged ONCE per pixel patch:
//   * a block owns 32*WCO output channels x 9 taps x 32 input channels (9 accumulators of 32x32 per wave) and walks
//     patches of R x CW output pixels;
//   * the x halo of a patch ((R-1)s+3 rows x (CW-1)s+3 columns, zeros outside the image) is fetched and split into
//     bf16 hi / lo ONCE into LDS as [piece][pixel][32 channels] (64-byte rows);
//   * K = 16 consecutive patch pixels per MFMA step.  The B operand (x, 16 pixels x 32 channels) of tap (kh, kw) is
//     the SAME LDS image read at a shifted pixel row with ds_read_b64_tr_b16 (transposing read: the image stays
//     pixel-major as it arrives from HBM, the MFMA wants k = pixel contiguous per lane); 4 consecutive pixels x 64 B
//     tile the 64 LDS banks exactly - conflict-free without padding.  For stride 2 the halo columns are stored
//     de-interleaved by parity so that "every second pixel" is again a run of consecutive rows;
//   * the A operand (dy, 32 channels x 16 pixels) needs no transposition at all: lane (co, half) loads its 8 pixels
//     of channel co straight from HBM (two 128-byte segments per load instruction), one K-step ahead, and splits
//     them in registers;
//   * per K-step a wave issues 27 MFMAs (9 taps x 3 products) against 36 transposing LDS reads, 8 global loads and
//     ~90 VALU - the matrix pipe is the limiter by construction; two blocks per CU cover each other's barriers.
// Narrow layers split K over the waves instead of output channels (WK waves take every WK-th K-step of a patch and
// write their own slab), so a 32-channel layer keeps all four waves busy.
// Determinism: every block accumulates in a fixed order and the slabs are summed in slab order (k_wgrad_reduce).
#include <math.h>
#include <stdlib.h>
#include <type_traits>
#include "snn_common.h"

#ifdef SNN_TUNING
// tuning builds only: cycle totals per phase of wave 0 of the first 1024 blocks {prologue, K loops, staging + barriers,
// epilogue, total}
__device__ unsigned long long g_halo_stamps[1024 * 8];
extern "C" int snn_debug_halo_stamps(unsigned long long* out, int n) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_halo_stamps), sizeof(unsigned long long) * n);
}
#define HSTAMP(i) do { unsigned long long t_ = __builtin_readcyclecounter(); hst[i] += t_ - hst_last; hst_last = t_; } while (0)
#else
#define HSTAMP(i) do {} while (0)
#endif

namespace {

constexpr int kThreads = 256;
constexpr int HALO_CAP = 256;          // halo pixels a block can stage
constexpr int PLANE = HALO_CAP * 64;   // bytes of one piece image: [pixel][32 channels] bf16
constexpr int NJ = HALO_CAP / 32;      // staging f32x4 per thread

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

struct HaloGeom {
    int H, W, Cin, OH, OW, Cout, stride;
    int64_t ldx, lddy;
    int R, CW, npr, npc, ppi;      // patch rows / columns, patches per image column / row / image
    int HR, HC, HWD, HWD2, halo;   // halo rows / columns, storage pitch (pixels), parity offset (stride 2), HR*HC
    int nks, npix;                 // K16 steps per patch, R*CW
    int patches, pps;              // N * ppi, patches per split
    int tiles_co, tiles_ci, splits;
    int Ktot;                      // 9 * Cin
    unsigned m_cw, m_hc, m_ppi, m_npc;  // ceil(2^32 / d): q = umulhi(n, m) for n * d < 2^32
    int ablate;                         // tuning builds only: bit 0 drops the dy loads, bit 1 the x loads (timing aid)
    float x_th;                         // NPROD 2: x holds saved LIF potentials, the operand is z = (v_dec > x_th)
};

static unsigned magic_u32(int d) { return d <= 1 ? 0u : (unsigned)((0x100000000ULL + (unsigned)d - 1) / (unsigned)d); }
__device__ __forceinline__ int div_magic(int n, int d, unsigned m) { return d == 1 ? n : (int)__umulhi((unsigned)n, m); }
__device__ __forceinline__ int div_magic2(int n, unsigned m) { return (int)__umulhi((unsigned)n, m); }  // divisor >= 2

__device__ __forceinline__ void split_bf16(float a, float b, unsigned& hi, unsigned& lo) {
    const bf16x2 ph = __builtin_convertvector(f32x2{a, b}, bf16x2);
    hi = __builtin_bit_cast(unsigned, ph);
    const float ra = a - __builtin_bit_cast(float, hi << 16);
    const float rb = b - __builtin_bit_cast(float, hi & 0xffff0000u);
    lo = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{ra, rb}, bf16x2));
}

// SB (bf16-storage mode, NPROD 1): x and dy are bf16 tensors - the halo goes to LDS as it arrives, the dy fragment is
// eight 2-byte loads; nothing is converted.
// NPROD 2 (snn_conv2d_spikes_wgrad): x holds the pre-reset potentials a LIF layer saved for its backward pass, NOT its spikes -
// that layer wrote no spike tensor (SNN_SCAN_SPIKES_FROM_VDEC) - and the operand z = (v_dec > x_th) is formed while the halo
// is written to LDS: one exact bf16 piece (1.0 or 0), no low image, the product high(dy) * low(x) is not issued.  Same bits
// as NPROD 3 fed the stored spikes.
template <int WCO, int WK, int S, int NPROD, bool SB = false>   // NPROD: 3 = bf16 x 3 (hi + lo pieces), 1 = bf16 x 1 (hi pieces only)
__global__ __launch_bounds__(kThreads, 2) void k_conv_wgrad_halo(const float* __restrict__ x,
                                                                 const float* __restrict__ dy,
                                                                 float* __restrict__ ws, HaloGeom g) {
    static_assert(WCO * WK == 4 && (S == 1 || S == 2), "4 waves; stride 1 or 2");
    static_assert(!SB || NPROD == 1, "bf16 storage: one product");
    constexpr int ES = SB ? 2 : 4;   // bytes per activation element in HBM
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * PLANE];  // hi image, lo image
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wco = wave % WCO, wk = wave / WCO;

    // ---- block -> (channel tile, patch split); the tiles of one split share an XCD (ids congruent mod 8) and read
    // the same dy / neighbouring x, so the second reader hits that XCD's L2
    const int tiles = g.tiles_co * g.tiles_ci;
    const int L = blockIdx.x;
    int z, tile;
    if (g.splits % 8 == 0) {
        z = (L % 8) + 8 * (L / (8 * tiles));
        tile = (L / 8) % tiles;
    } else {
        z = L / tiles;
        tile = L % tiles;
    }
    const int co0 = (tile % g.tiles_co) * (32 * WCO) + 32 * wco;
    const int ci0 = (tile / g.tiles_co) * 32;
    const int p_lo = z * g.pps;
    const int p_hi = p_lo + g.pps < g.patches ? p_lo + g.pps : g.patches;

    // ---- per-lane constants of the two operand maps
    const int ar = lane & 31, ah = lane >> 5;                                   // A: channel co0 + ar, pixels 8*ah + j
    const int bh = lane >> 5, bcb = (lane >> 4) & 1, bq = (lane >> 2) & 3, bp = lane & 3;  // B: see ds_read_b64_tr_b16
    const int b_lane_off = bcb * 32 + bp * 8;
    const int sl8 = tid & 7, sps = tid >> 3;                                    // staging: channel quad, pixel slot
    // tap (kh, kw) of output pixel (rr, cc) reads halo row S*rr + kh and, stride 1, column cc + kw; stride 2: the
    // even columns are stored first, then the odd ones (HWD2 pixels further): kw = 0 -> even[cc], 1 -> odd[cc],
    // 2 -> even[cc + 1].  So per kh there are two row bases and the three taps are IMMEDIATE offsets from them.
    const int row_bytes = g.HWD * 64;
    const int odd_bytes = S == 1 ? 64 : g.HWD2 * 64;   // kw = 1 relative to kw = 0
    constexpr int KW2 = S == 1 ? 128 : 64;              // kw = 2 relative to kw = 0

    struct Patch {
        int oy0, ox0, rv, cwv;   // first output pixel, valid rows / columns
        __amdgpu_buffer_rsrc_t rs_x, rs_d;
    };
    auto setup = [&](int P) {    // P is block-uniform: scalar arithmetic
        Patch q;
        const int img = div_magic(P, g.ppi, g.m_ppi);
        const int rem = P - img * g.ppi;
        const int pr = div_magic(rem, g.npc, g.m_npc), pc = rem - pr * g.npc;
        q.oy0 = pr * g.R;
        q.ox0 = pc * g.CW;
        q.rv = g.OH - q.oy0 < g.R ? g.OH - q.oy0 : g.R;
        q.cwv = g.OW - q.ox0 < g.CW ? g.OW - q.ox0 : g.CW;
        const int64_t ipix = (int64_t)g.H * g.W, opix = (int64_t)g.OH * g.OW;
        q.rs_x = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<char*>(reinterpret_cast<const char*>(x) + (int64_t)img * ipix * g.ldx * ES), 0,
            (int)(((ipix - 1) * g.ldx + g.Cin) * ES), 0x00020000);
        q.rs_d = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<char*>(reinterpret_cast<const char*>(dy) + (int64_t)img * opix * g.lddy * ES), 0,
            (int)(((opix - 1) * g.lddy + g.Cout) * ES), 0x00020000);
        return q;
    };

    // halo of the NEXT patch on its way to LDS: 4 fp32 values, or (SB) 4 bf16 values as two dwords (integer-typed: carried
    // in float lanes and bit-cast back per element, hipcc 7.2 narrows the 8-byte buffer load to 4 bytes)
    using SReg = typename std::conditional<SB, u32x2, f32x4>::type;
    SReg st[NJ];
    // `opq` is an opaque zero, re-made per patch: without it the compiler hoists the 16 (row, column) pairs of the
    // staging slots out of the patch loop into VGPRs that the B-fragment double buffer needs
    auto load_halo = [&](const Patch& q, int opq) {
        const int iy0 = q.oy0 * S - 1, ix0 = q.ox0 * S - 1;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int hl = sps + 32 * j + opq;
            const int hr = div_magic2(hl, g.m_hc), hc = hl - hr * g.HC;
            const int iy = iy0 + hr, ix = ix0 + hc;
            const bool ok = (hl < g.halo) & ((unsigned)iy < (unsigned)g.H) & ((unsigned)ix < (unsigned)g.W);
            const int off = ((iy * g.W + ix) * (int)g.ldx + ci0 + 4 * sl8) * ES;
            const int voff = off | -(int)!ok | -(g.ablate >> 1 & 1);  // all ones: out of range -> zeros (no branch)
            if constexpr (SB) st[j] = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(q.rs_x, voff, 0, 0));
            else st[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(q.rs_x, voff, 0, 0));
        }
    };
    auto write_halo = [&](int opq) {
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int hl = sps + 32 * j + opq;
            const int hr = div_magic2(hl, g.m_hc), hc = hl - hr * g.HC;
            const int sp = hr * g.HWD + (S == 1 ? hc : (hc & 1) * g.HWD2 + (hc >> 1));
            unsigned h0, l0 = 0, h1, l1 = 0;
            if constexpr (SB) {
                h0 = st[j][0];
                h1 = st[j][1];
            } else if constexpr (NPROD == 2) {   // spikes from potentials: bf16 1.0 = 0x3F80
                h0 = (st[j][0] > g.x_th ? 0x3F80u : 0u) | (st[j][1] > g.x_th ? 0x3F800000u : 0u);
                h1 = (st[j][2] > g.x_th ? 0x3F80u : 0u) | (st[j][3] > g.x_th ? 0x3F800000u : 0u);
            } else {
                split_bf16(st[j][0], st[j][1], h0, l0);
                split_bf16(st[j][2], st[j][3], h1, l1);
            }
            if (hl < g.halo) {
                *reinterpret_cast<u32x2*>(smem + sp * 64 + sl8 * 8) = u32x2{h0, h1};
                if constexpr (NPROD == 3) *reinterpret_cast<u32x2*>(smem + PLANE + sp * 64 + sl8 * 8) = u32x2{l0, l1};
            }
        }
    };

    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;

    // dy fragment of K-step ks: 8 consecutive pixels of ONE patch row (CW is a multiple of 8) of this lane's channel;
    // zeros (offset -1) for pixels outside the patch / the image and for ks past the last step - the prefetch needs
    // no branch.  The same (row, first column) also places the B fragment: both operands use k = 16 ks + 8 (lane / 32).
    using ARaw = typename std::conditional<SB, unsigned, float>::type;   // SB: the bf16 value in the low half
    ARaw araw[8];
    const int lddy4 = (int)g.lddy * ES;   // bytes per dy pixel
    auto load_a = [&](const Patch& q, int ks) {
        const int t0 = 16 * ks + 8 * ah;
        const int rr = div_magic2(t0, g.m_cw), cc = t0 - rr * g.CW;
        const int off = ((q.oy0 + rr) * g.OW + q.ox0 + cc) * lddy4 + (co0 + ar) * ES;
        const int lim = (ks < g.nks && rr < q.rv) ? q.cwv - cc : 0;   // valid pixels of this group of 8
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int voff = (j < lim ? off + j * lddy4 : -1) | -(g.ablate & 1);
            if constexpr (SB)   // the bf16 value in the low half
                araw[j] = (unsigned)(unsigned short)__builtin_amdgcn_raw_buffer_load_b16(q.rs_d, voff, 0, 0);
            else
                araw[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(q.rs_d, voff, 0, 0));
        }
    };

#ifdef SNN_TUNING
    unsigned long long hst[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long hst_last = __builtin_readcyclecounter();
    const unsigned long long hst_begin = hst_last;
#endif
    Patch cur;
    if (p_lo < p_hi) {
        int opq = 0;
        // (hoisting allowed)
        cur = setup(p_lo);
        load_halo(cur, opq);
        write_halo(opq);
    }
    __syncthreads();
    HSTAMP(0);
#pragma unroll 1
    for (int P = p_lo; P < p_hi; ++P) {
        int opq = 0;
        // (hoisting allowed)
        Patch nxt = cur;
        const bool more = P + 1 < p_hi;
        if (more) {
            nxt = setup(P + 1);
            load_halo(nxt, opq);   // in flight during the whole K loop of this patch
        }
        load_a(cur, wk);
        HSTAMP(2);
#pragma unroll 1
        for (int ks = wk; ks < g.nks; ks += WK) {
            // ---- A: split the fragment fetched one step ago, fetch the next one (masked past the last step)
            unsigned ahw[4], alw[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if constexpr (SB) {
                    ahw[e] = araw[2 * e] | (araw[2 * e + 1] << 16);
                    alw[e] = 0;
                } else {
                    split_bf16(araw[2 * e], araw[2 * e + 1], ahw[e], alw[e]);
                }
            }
            const bf16x8 Ah = __builtin_bit_cast(bf16x8, u32x4{ahw[0], ahw[1], ahw[2], ahw[3]});
            const bf16x8 Al = __builtin_bit_cast(bf16x8, u32x4{alw[0], alw[1], alw[2], alw[3]});
            load_a(cur, ks + WK);
            // ---- B: LDS rows of this lane's pixels (two groups of 4 per K-step, consecutive in one patch row); pixels
            // past the patch carry a zero A fragment, they only have to read INITIALISED memory (NaN * 0 = NaN):
            // clamp the row
            int brow[3][2];  // [kh][even / odd column plane] byte address of (pixel group 0, this lane)
            {
                const int t0 = 16 * ks + 8 * bh;
                int rr = div_magic2(t0, g.m_cw);
                const int cc = t0 - rr * g.CW;
                rr = rr < g.R ? rr : 0;
                const int b0 = (rr * S * g.HWD + cc + bq) * 64 + b_lane_off;
#pragma unroll
                for (int kh = 0; kh < 3; ++kh) {
                    brow[kh][0] = b0 + kh * row_bytes;
                    brow[kh][1] = brow[kh][0] + odd_bytes;
                }
            }
            // fragments of tap t+1 are requested before the products of tap t are issued (two register sets)
            s16x4 fh[2][2], fl[2][2];
            auto read_b = [&](int t, int buf) {
                const int kh = t / 3, kw = t - 3 * (t / 3);
                const unsigned char* base = smem + brow[kh][kw == 1 ? 1 : 0] + (kw == 2 ? KW2 : 0);
#pragma unroll
                for (int s = 0; s < 2; ++s) {   // the second group of 4 pixels: 4 rows = 256 bytes further
                    const auto ph = (__attribute__((address_space(3))) s16x4*)(base + 256 * s);
                    const auto pl = (__attribute__((address_space(3))) s16x4*)(base + 256 * s + PLANE);
                    fh[buf][s] = __builtin_amdgcn_ds_read_tr16_b64_v4i16(ph);
                    if constexpr (NPROD == 3) fl[buf][s] = __builtin_amdgcn_ds_read_tr16_b64_v4i16(pl);
                }
            };
            read_b(0, 0);
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                if (t + 1 < 9) read_b(t + 1, (t + 1) & 1);
                const int u = t & 1;
                const bf16x8 Bh = __builtin_bit_cast(bf16x8, __builtin_shufflevector(fh[u][0], fh[u][1], 0, 1, 2, 3, 4, 5, 6, 7));
                if constexpr (NPROD == 3) {
                    const bf16x8 Bl = __builtin_bit_cast(bf16x8, __builtin_shufflevector(fl[u][0], fl[u][1], 0, 1, 2, 3, 4, 5, 6, 7));
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Al, Bh, acc[t], 0, 0, 0);  // small terms first
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ah, Bl, acc[t], 0, 0, 0);
                } else if constexpr (NPROD == 2) {   // the low image of a spike is zero
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Al, Bh, acc[t], 0, 0, 0);
                }
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ah, Bh, acc[t], 0, 0, 0);
            }
            // schedule shape: [reads of tap t+1] then the products of tap t
            constexpr int NRD = NPROD == 3 ? 4 : 2;
            __builtin_amdgcn_sched_group_barrier(0x100, NRD, 0);
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                if (t + 1 < 9) __builtin_amdgcn_sched_group_barrier(0x100, NRD, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, NPROD, 0);
            }
        }
        HSTAMP(1);
        __syncthreads();          // every wave is done with this patch's halo
        HSTAMP(3);
        if (more) write_halo(opq);
        __syncthreads();
        cur = nxt;
        HSTAMP(2);
    }

    // ---- the WK waves that shared the K-steps of the patches add their accumulators through LDS, in wave order
    // (fixed order: reproducible); one tap (a 32x32 tile per wave) per round
    if constexpr (WK > 1) {
        float* red = reinterpret_cast<float*>(smem);
#pragma unroll
        for (int t = 0; t < 9; ++t) {   // unrolled: a runtime index into the accumulators would send them to scratch
            __syncthreads();
            if (wk > 0) {
#pragma unroll
                for (int e = 0; e < 16; ++e) red[((wk - 1) * WCO + wco) * 1024 + e * 64 + lane] = acc[t][e];
            }
            __syncthreads();
            if (wk == 0) {
                for (int w = 1; w < WK; ++w)
#pragma unroll
                    for (int e = 0; e < 16; ++e) acc[t][e] = acc[t][e] + red[((w - 1) * WCO + wco) * 1024 + e * 64 + lane];
            }
        }
    }
    // ---- slab of this split: [Cout][9][Cin]; C/D map: column (ci) = lane & 31, row (co) = (e&3) + 8*(e>>2) + 4*(lane>>5)
    if (wk == 0) {
        float* slab = ws + (int64_t)z * g.Cout * (int64_t)g.Ktot;
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int co = co0 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
                slab[(int64_t)co * g.Ktot + t * g.Cin + ci0 + (lane & 31)] = acc[t][e];
            }
    }
#ifdef SNN_TUNING
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    HSTAMP(4);
    if (tid == 0 && blockIdx.x < 1024) {
        hst[5] = __builtin_readcyclecounter() - hst_begin;
        for (int i = 0; i < 8; ++i) g_halo_stamps[blockIdx.x * 8 + i] = hst[i];
    }
#endif
}

static bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }
static bool aligned8(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 7u) == 0; }

}  // namespace

SnnWgradHaloPlan snn_wgrad_halo_plan(int64_t N, int H, int W, int Cin, int Ho, int Wo, int Cout, int KH, int KW,
                                     int stride, int pad) {
    SnnWgradHaloPlan p = {};
    if (snn_tuning_env("SNN_WGRAD_NO_HALO")) return p;
    if (KH != 3 || KW != 3 || pad != 1 || (stride != 1 && stride != 2)) return p;
    if (Cin % 32 != 0 || Cout % 32 != 0 || Wo < 8 || Ho < 1) return p;
    if (Ho != (H + 2 - 3) / stride + 1 || Wo != (W + 2 - 3) / stride + 1) return p;
    // measured (tools/wgrad_bench.py, GEN1 B=5 T=32): ahead of the implicit-GEMM kernel from the 30x38 maps upwards
    // (196 vs 214 us at 128->128, 323 vs 551 us at 32->32 120x152); on the small deep maps (15x19, 8x10: a few K-steps
    // per patch, a handful of patches per block) the per-patch staging and the slab traffic outweigh the reuse
    if (N * Ho * (int64_t)Wo < 150000 && !snn_tuning_env("SNN_WGRAD_HALO_ALWAYS")) return p;
    // ---- waves: 32 output channels each; narrow layers split the K-steps of a patch over the waves instead
    p.wco = Cout >= 128 ? 4 : (Cout >= 64 ? 2 : 1);
    p.wk = 4 / p.wco;
    // ---- patch shape: least time per image in a model of what a patch costs - its K-steps per wave (27 MFMAs of 32
    // cycles each) plus a FIXED part, the staging of its halo between two barriers (load_halo / write_halo run their eight
    // slots whatever the halo size) - then the halo bytes per output pixel.  Round 2 ranked by the useful share of the
    // executed K-steps alone and gave the 32 -> 32 layer at 120x152 8x8 patches: no masked pixel, but ONE K-step per wave and
    // patch - in-kernel stamps: 6 800 cycles per patch in staging (unchanged with every global load removed), as long as in
    // the K loops.  With 12x16 patches (three K-steps per wave and patch) 327 -> 232-244 us.  The fixed part is what the
    // co-resident block does NOT hide: 600 cycles reproduce both that gain and the measured tie between 10x8 and 4x40
    // patches on the 128-channel layers at 30x38 (3 000, the stamp-based guess, predicted 15 % for 4x40: measured -3 %).
    const char* cs_env = snn_tuning_env("SNN_HALO_PATCH_CYCLES");   // tuning builds: 0 = the round-2 ranking
    const double stage_cycles = cs_env ? atof(cs_env) : 600.0;
    double best = -1.0;
    for (int cw = 8; cw <= 64; cw += 8) {   // multiples of 8: a K-step's group of 8 pixels never straddles two rows
        if (cw >= Wo + 8) break;
        const int hc = (cw - 1) * stride + 3;
        const int hwd2 = (hc + 1) / 2, hwd = stride == 1 ? hc : 2 * hwd2;
        const int hr_max = HALO_CAP / hwd;
        if (hr_max < 3) continue;
        const int r_max = (hr_max - 3) / stride + 1;
        for (int r = 1; r <= r_max && r <= Ho; ++r) {
            const int npr = (Ho + r - 1) / r, npc = (Wo + cw - 1) / cw, nks = r * cw / 16 + (r * cw % 16 != 0);
            const int nks_w = (nks + p.wk - 1) / p.wk * p.wk;
            const double eff = (double)Ho * Wo / ((double)npr * npc * nks_w * 16);
            const double halo_per_px = (double)((r - 1) * stride + 3) * hc / ((double)r * cw * stride * stride);
            double score = eff - 0.03 * halo_per_px + 1e-4 * r * cw / 256.0;
            if (stage_cycles > 0.0)   // pixels per cycle of the model, the halo overhead as the tie-break
                score = (double)Ho * Wo / ((double)npr * npc * ((nks_w / p.wk) * 864.0 + stage_cycles)) - 1e-4 * halo_per_px;
            if (score > best) {
                best = score;
                p.R = r; p.CW = cw; p.npr = npr; p.npc = npc; p.nks = nks;
                p.HR = (r - 1) * stride + 3; p.HC = hc; p.HWD = hwd; p.HWD2 = hwd2;
            }
        }
    }
    if (best < 0) return p;
    const int64_t patches = N * p.npr * p.npc;
    if (patches <= 0 || patches > 0x3fffffffLL) return p;
    p.tiles_co = Cout / (32 * p.wco);
    p.tiles_ci = Cin / 32;
    const int64_t tiles = (int64_t)p.tiles_co * p.tiles_ci;
    // ---- patch splits.  More splits = more blocks working in parallel, but every split writes a slab that the
    // ordered reduce reads back: time(s) ~ a / s + b * s with a = K-steps per wave x 0.41 us (27 MFMAs of 32 cycles)
    // and b = slab megabytes x 0.5 us (written and read at ~4 TB/s) - the small deep layers want few splits.  At most
    // one resident wave of blocks (two per CU), whole groups of 8 (XCD mapping), not more than the patches,
    // workspace <= 256 MiB.
    const double a_us = (double)patches * ((p.nks + p.wk - 1) / p.wk) * 0.41;
    const double b_us = (double)Cout * 9 * Cin * 4 / 1e6 * 0.5;
    int64_t s = (int64_t)(sqrt(a_us / b_us) + 0.5);
    const int64_t by_residency = (2 * (int64_t)snn_num_cu()) / tiles;
    if (s > by_residency) s = by_residency;
    const int64_t by_mem = (int64_t)(64 << 20) / ((int64_t)Cout * 9 * Cin);
    if (s > by_mem) s = by_mem;
    if (s > patches) s = patches;
    if (s >= 16) s = s / 8 * 8;
    if (s < 1) s = 1;
    p.splits = (int)s;
    p.pps = (int)((patches + s - 1) / s);
    p.slabs = p.splits;
    p.patches = (int)patches;
    p.ok = 1;
    return p;
}

int snn_wgrad_halo_launch(const SnnWgradHaloPlan& p, const float* x, int64_t ldx, const float* dy, int64_t lddy,
                          float* workspace, int64_t N, int H, int W, int Cin, int Ho, int Wo, int Cout, int stride,
                          int nprod, bool bf16_storage, hipStream_t st, float x_th) {
    // 32-bit byte offsets inside one image (buffer addressing)
    if ((int64_t)H * W * ldx * 4 >= 0x7fffffffLL || (int64_t)Ho * Wo * lddy * 4 >= 0x7fffffffLL) return -1;
    if (ldx % 4 != 0 || !(bf16_storage ? aligned8(x) : aligned16(x))) return -1;
    if (bf16_storage && ((reinterpret_cast<uintptr_t>(dy) & 1u) || nprod != 1)) return -1;
    HaloGeom g;
    g.H = H; g.W = W; g.Cin = Cin; g.OH = Ho; g.OW = Wo; g.Cout = Cout; g.stride = stride;
    g.ldx = ldx; g.lddy = lddy;
    g.R = p.R; g.CW = p.CW; g.npr = p.npr; g.npc = p.npc; g.ppi = p.npr * p.npc;
    g.HR = p.HR; g.HC = p.HC; g.HWD = p.HWD; g.HWD2 = p.HWD2; g.halo = p.HR * p.HC;
    g.nks = p.nks; g.npix = p.R * p.CW;
    g.patches = p.patches; g.pps = p.pps;
    g.tiles_co = p.tiles_co; g.tiles_ci = p.tiles_ci; g.splits = p.splits;
    g.Ktot = 9 * Cin;
    g.ablate = snn_tuning_env("SNN_HALO_ABLATE") ? atoi(snn_tuning_env("SNN_HALO_ABLATE")) : 0;
    g.x_th = x_th;
    g.m_cw = magic_u32(g.CW); g.m_hc = magic_u32(g.HC); g.m_ppi = magic_u32(g.ppi); g.m_npc = magic_u32(g.npc);
    const int64_t nblocks = (int64_t)p.tiles_co * p.tiles_ci * p.splits;
    if (nblocks > 0x7fffffffLL) return -1;
    dim3 grid((unsigned)nblocks);
#define SNN_HALO_LAUNCH(WCO_, WK_)                                                                                  \
    do {                                                                                                            \
        if (bf16_storage && stride == 1)                                                                            \
            hipLaunchKernelGGL((k_conv_wgrad_halo<WCO_, WK_, 1, 1, true>), grid, dim3(kThreads), 0, st, x, dy, workspace, g); \
        else if (bf16_storage)                                                                                      \
            hipLaunchKernelGGL((k_conv_wgrad_halo<WCO_, WK_, 2, 1, true>), grid, dim3(kThreads), 0, st, x, dy, workspace, g); \
        else if (stride == 1 && nprod == 2)                                                                         \
            hipLaunchKernelGGL((k_conv_wgrad_halo<WCO_, WK_, 1, 2>), grid, dim3(kThreads), 0, st, x, dy, workspace, g); \
        else if (nprod == 2)                                                                                        \
            hipLaunchKernelGGL((k_conv_wgrad_halo<WCO_, WK_, 2, 2>), grid, dim3(kThreads), 0, st, x, dy, workspace, g); \
        else if (stride == 1 && nprod == 3)                                                                         \
            hipLaunchKernelGGL((k_conv_wgrad_halo<WCO_, WK_, 1, 3>), grid, dim3(kThreads), 0, st, x, dy, workspace, g); \
        else if (nprod == 3)                                                                                        \
            hipLaunchKernelGGL((k_conv_wgrad_halo<WCO_, WK_, 2, 3>), grid, dim3(kThreads), 0, st, x, dy, workspace, g); \
        else if (stride == 1)                                                                                       \
            hipLaunchKernelGGL((k_conv_wgrad_halo<WCO_, WK_, 1, 1>), grid, dim3(kThreads), 0, st, x, dy, workspace, g); \
        else                                                                                                        \
            hipLaunchKernelGGL((k_conv_wgrad_halo<WCO_, WK_, 2, 1>), grid, dim3(kThreads), 0, st, x, dy, workspace, g); \
    } while (0)
    if (p.wco == 4) SNN_HALO_LAUNCH(4, 1);
    else if (p.wco == 2) SNN_HALO_LAUNCH(2, 2);
    else SNN_HALO_LAUNCH(1, 4);
#undef SNN_HALO_LAUNCH
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        snn_set_error("snn_conv2d_wgrad: halo kernel launch failed: %s", hipGetErrorString(e));
        return 2;
    }
    return 0;
}
