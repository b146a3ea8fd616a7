// Halo-resident 3x3 / stride 1 / pad 1 convolution for the 64- and 128-channel layers, forward AND data gradient,
// on the gfx950 matrix cores with 16-bit split products (fp16 x 3 forward, bf16 x 3 backward; fp32 storage and
// accumulation, same arithmetic per product as conv.hip's k_conv_gather).
//
// Replaces nn.Conv2d(C, C', 3, padding=1, bias=False) forward and its data gradient (reference
// models/modules/layer_gen.py:129-136) for the layer-major schedule.  The data gradient is the SAME kernel: dx is a
// 3x3 convolution of dy with the mirrored taps of w^T, and the mirroring / transposition live in the weight image
// (snn_weight_frag_image, flip = 1, built from the [Cin][KH][KW][Cout] transpose).
//
// Why a second kernel.  The implicit GEMM (k_conv_gather) fetches one shifted copy of the activation tile PER TAP and
// splits it into its 16-bit pieces each time - nine fetches through L1 and nine conversions of every element - and
// pays two barriers per 32-deep k-step; its matrix pipe is 23-33 % busy (profiles/r02_pmc_mfma_gen1.txt).  Here:
//   * PADDED-STRIP pixel order.  All images are laid end to end as rows of PW = W + 1 cells with ONE zero cell after
//     every image row and ONE zero row after every image: (row y, column -1) and (row y-1, column W) are the same
//     zero cell, so tap (kh, kw) of cell s is cell s + (kh-1)*PW + (kw-1) - no border masks anywhere.  A block owns
//     128 consecutive strip cells (their outputs; pad cells are computed and dropped: W*H / ((W+1)*(H+1)) of the
//     work is useful, 94 % at 30x38, 97 % at 60x76) and needs the 128 + 2*PW + 2 cells around them.
//   * A operand: per 32-channel chunk that halo is fetched ONCE (raw buffer loads spread over the nine taps of the
//     previous chunk, out-of-image cells by the hardware range check), split ONCE into its two 16-bit pieces and
//     written to LDS as [piece][cell][32 ch] (64-byte cells, 16-byte slots XOR-swizzled with (cell >> 2) & 3: a
//     ds_read_b128 lane group always covers 16 cells that are distinct modulo 16, hence all 64 banks once).  The nine
//     taps read that image at nine cell offsets.
//   * B operand: the weights are pre-arranged ONCE per optimiser step in exactly the order the MFMA fragments are
//     read - [tap][32-ci chunk][32-co tile][k16][piece][lane] x 16 bytes (snn_weight_frag_image) - so a k-step's
//     tile is one contiguous 4 KiB x (CO/32) block that goes global -> LDS by LDS-DMA (global_load_lds, no registers,
//     no VALU) one k-step ahead into a double buffer, and every fragment read is lane-linear (conflict-free).
//   * one barrier per k-step of 24 MFMAs per wave; the only VALU in the loop is the fragment address arithmetic.
// Epilogue as in k_conv_gather: accumulators transposed through LDS, 16-byte stores, up to two fused addends
// (gradient accumulation), BatchNorm statistics partials of the stored values (forward).
#include <stdlib.h>
#include <type_traits>
#include "snn_common.h"

namespace {

constexpr int kThreads = 256;
constexpr int HBM_ = 128;               // strip cells (GEMM rows) per block
constexpr int HCELLS = 288;             // halo cells staged per chunk (9 passes of 32): PW <= 79
constexpr int HPIECE = HCELLS * 64;     // bytes of one piece image [cell][32 ch] x 16 bit
constexpr int NPASS = HCELLS / 32;      // == taps: one staging pass is requested during every tap of the previous chunk
static_assert(NPASS == 9, "one halo staging pass per tap");

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr float kF16WeightScale = 256.0f;   // the fp16 x 3 pre-scales of conv.hip (exact powers of two)
constexpr float kF16ActScale = 16.0f;
constexpr float kF16Unscale = 1.0f / (kF16WeightScale * kF16ActScale);

struct HaloGeom {
    int64_t ldx, ldy, ld_add, ld_add2;
    int N, H, W, Cin, Cout;
    int PW, PH;                 // strip row pitch (W + 1) and rows per image (H + 1)
    int G;                      // images per group: tiles do not cross groups (= frames per timestep with statistics)
    int group_cells;            // G * PH * PW
    int tiles_per_group, tiles, tiles_per_xcd, ntiles_n;
    unsigned magic_pw, magic_ph;   // ceil(2^32 / d): q = umulhi(n, magic) for n * d < 2^32 (small n only)
    int out_vec;
    double* bn_partial;         // forward statistics partials [t][c][chunk][2] (null: none); chunk = tile of the group
    int OH, OW;                 // k_conv_s2dgrad3 only: size of the produced tensor dx (the strip grid H x W is dy's)
    int bn_T, bn_tc, bn_fps;    // BNAP: timesteps, T * Cin (distance of the coefficient planes), frames per timestep
    int tiles_x, tiles_img;     // RECT: 4 x 32 pixel tiles per image row, tiles per image
    float x_th;                 // XSP: x holds saved LIF potentials, the operand is z = (x > x_th)
};

__device__ __forceinline__ unsigned udiv_small(unsigned n, unsigned magic) { return __umulhi(n, magic); }

// two 16-bit pieces of four fp32 values: fp16 (x * 2^4 = h + l) or bf16 (x = h + l); same arithmetic as conv.hip
template <bool F16>
__device__ __forceinline__ void split4(const f32x4& v, u32x2& hi, u32x2& lo) {
#pragma unroll
    for (int e = 0; e < 4; e += 2) {
        if constexpr (F16) {
            const float a = v[e] * kF16ActScale, b = v[e + 1] * kF16ActScale;
            const f16x2 ph = __builtin_convertvector(f32x2{a, b}, f16x2);   // RNE; out of range -> inf (loud)
            const f16x2 pl = __builtin_convertvector(f32x2{a - (float)ph[0], b - (float)ph[1]}, f16x2);
            hi[e >> 1] = __builtin_bit_cast(unsigned, ph);
            lo[e >> 1] = __builtin_bit_cast(unsigned, pl);
        } else {
            f32x2 rest = {v[e], v[e + 1]};
            const bf16x2 ph = __builtin_convertvector(rest, bf16x2);
            const unsigned bits = __builtin_bit_cast(unsigned, ph);
            rest[0] -= __builtin_bit_cast(float, bits << 16);
            rest[1] -= __builtin_bit_cast(float, bits & 0xffff0000u);
            const bf16x2 pl = __builtin_convertvector(rest, bf16x2);
            hi[e >> 1] = bits;
            lo[e >> 1] = __builtin_bit_cast(unsigned, pl);
        }
    }
}

// Wave combine of the fp64 statistics partials without LDS: value of lane l ^ 8 (DPP rotation inside a 16-lane row: every
// lane), l ^ 16 (v_permlane16_swap: valid in the ODD 16-lane rows) and l ^ 32 (v_permlane32_swap: valid in the UPPER 32
// lanes) - taken in this order the complete sum stands in lanes 48..63, through the same addition tree as an xor-shuffle
// ladder (same bits), with VALU moves instead of two ds_bpermute round trips per value and stride.
__device__ __forceinline__ double partner8(double v) {
    const u32x2 u = __builtin_bit_cast(u32x2, v);
    const u32x2 o = {(unsigned)__builtin_amdgcn_update_dpp(0, (int)u[0], 0x128, 0xf, 0xf, false),   // row_ror:8
                     (unsigned)__builtin_amdgcn_update_dpp(0, (int)u[1], 0x128, 0xf, 0xf, false)};
    return __builtin_bit_cast(double, o);
}
__device__ __forceinline__ double partner16(double v) {
    const u32x2 u = __builtin_bit_cast(u32x2, v);
    const u32x2 o = {__builtin_amdgcn_permlane16_swap(u[0], u[0], false, false)[0],
                     __builtin_amdgcn_permlane16_swap(u[1], u[1], false, false)[0]};
    return __builtin_bit_cast(double, o);
}
__device__ __forceinline__ double partner32(double v) {
    const u32x2 u = __builtin_bit_cast(u32x2, v);
    const u32x2 o = {__builtin_amdgcn_permlane32_swap(u[0], u[0], false, false)[0],
                     __builtin_amdgcn_permlane32_swap(u[1], u[1], false, false)[0]};
    return __builtin_bit_cast(double, o);
}

// byte offset of the 16-byte slot `slot` (0..3 = channels 8*slot .. 8*slot+7) of halo cell `cell` inside a piece image
__device__ __forceinline__ int cell_slot_off(int cell, int slot) { return cell * 64 + ((slot ^ ((cell >> 2) & 3)) << 4); }

// CO: output channels per block (64 | 128); F16: fp16 pieces (forward) or bf16 pieces (data gradient).
// 4 waves as 2 x 2: a wave owns 64 cells x CO/2 channels (TM = 2 row tiles, TN = CO/64 column tiles of 32 x 32).
// ABL (tuning builds only, timing experiments with WRONG results): 1 no weight DMA, 2 no per-k-step wait + barrier,
// 4 no halo prefetch loads, 8 no output stores, 16 no fragment reads of the halo image (one read per k-step instead)
// BNAP (data gradient behind a train-mode BatchNorm): the operand x is gx, the gradient BEFORE the BatchNorm-backward
// affine; dy = A[t][c] * gx + B[t][c] * y + C[t][c] (the statement of snn_bn_bwd_apply, t = image / frames per step) is
// formed when a staging pass has landed - one pass per tap, behind that tap's MFMAs - stored to dy_out for the cells
// the tile owns (the weight gradient reads it) and split into the LDS image.  The separate apply pass over the layer
// (12 bytes per element) disappears; coefficients of the <= 3 timesteps a halo can touch are staged in LDS once per block.
constexpr int BN_NT = 3;      // timesteps of coefficients a block stages
constexpr int BN_CMAX = 128;  // input channels (K) the BNAP variant supports

// RECT (rows longer than 78 pixels: the halo of a 128-cell strip tile would not fit): a block owns a 4 x 32 pixel
// rectangle of ONE image instead; the LDS image is its 6 x 34 halo (204 cells, row pitch 34), a row tile of the MFMA is
// one rectangle row = 32 consecutive cells again (conflict-free reads as in strip order), taps are offsets
// (kh-1)*34 + (kw-1).  Everything else - chunk staging, weight DMA, epilogue - is the strip kernel's.
constexpr int RTH = 4, RTW = 32, RPITCH = RTW + 2, RCELLS = (RTH + 2) * RPITCH;

// SBF (SNN_PREC_BF16S, the bf16-STORAGE throughput mode): x, y and the addends are bf16 tensors.  The staged halo is then
// the bf16 image itself (8-byte loads, no split: ONE piece), the weight image's high pieces are the bf16-rounded weights
// (the low pieces are neither copied nor multiplied): one MFMA product per multiply-add; results are rounded to bf16 in
// the epilogue (the BatchNorm partials are taken from the fp32 values before that rounding).
// XSP (snn_conv3x3_halo_spikes; forward only): x holds the pre-reset potentials a LIF layer saved for its backward pass, NOT
// its spikes - that layer wrote no spike tensor (SNN_SCAN_SPIKES_FROM_VDEC) - and the operand z = (v_dec > x_th) is formed while
// the halo goes to LDS: ONE exact fp16 piece (1.0 x 2^4 or 0), no low image, the product low(x) * high(w) is not issued (two
// MFMA products instead of three).  Same bits as the plain kernel on the stored spikes, whose low pieces are zeros.
template <int CO, bool F16, int ABL = 0, bool BNAP = false, bool RECT = false, bool SBF = false, bool XSP = false>
__global__ __launch_bounds__(kThreads, 2) void k_conv_halo3(const float* __restrict__ x,
                                                           const unsigned char* __restrict__ wimg,
                                                           float* __restrict__ y, HaloGeom g,
                                                           const float* __restrict__ addend,
                                                           const float* __restrict__ addend2,
                                                           const float* __restrict__ bn_y = nullptr,
                                                           const float* __restrict__ bn_coef = nullptr,
                                                           float* __restrict__ dy_out = nullptr) {
    // CO = 32 (the 32-channel layers of the full-resolution stage): the four waves side by side along the cells, one
    // 32 x 32 tile each
    constexpr int WM = CO == 32 ? 4 : 2, WN = 4 / WM, TM = CO == 32 ? 1 : 2, TN = CO == 32 ? 1 : CO / 64;
    static_assert(CO == 32 || CO == 64 || CO == 128, "channel tile");
    static_assert(!BNAP || CO >= 64, "the BatchNorm-apply variant covers the 64 / 128-channel tiles");
    constexpr int BTILE = (CO / 32) * 4096;     // bytes of one k-step's weight tile
    constexpr int NDMA = (CO / 32) * 4 / 4;     // 1-KiB LDS-DMA pieces per wave and k-step
    constexpr int ES = SBF ? 2 : 4;             // bytes per activation element in HBM
    static_assert(!SBF || (!F16 && !BNAP && ABL == 0), "bf16 storage: bf16 MFMA, plain variant");
    static_assert(!XSP || (F16 && !BNAP && !SBF && ABL == 0), "spikes from potentials: the forward arithmetic, plain variant");
    constexpr int CF_BYTES = BNAP ? 3 * BN_NT * BN_CMAX * 4 : 0;
    // cells the LDS halo image holds: a rectangle tile's halo is 6 x 34 = 204 cells (7 staging passes), not the 288 of the
    // widest strip tile - 37 instead of 45 KiB of LDS for the 32-channel instance, i.e. FOUR instead of three blocks per CU
    // on the 120x152 layers, whose blocks (54 MFMAs per wave) live on latency, not on the matrix pipe
    constexpr int HC = RECT ? (RCELLS + 31) / 32 * 32 : HCELLS;
    constexpr int HP = HC * 64;                        // bytes of one piece image
    constexpr int RED_BYTES = (F16 || SBF) ? 4 * (CO == 128 ? 64 : 32) * 16 : 0;   // statistics: [wave][channels of a wave][2] fp64
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * HP + 2 * BTILE + CF_BYTES + RED_BYTES];   // ONE array
    unsigned char* Aimg = smem;                        // [2 pieces][HC][64 B]
    unsigned char* Bimg = smem + 2 * HP;               // [2 buffers][BTILE]
    [[maybe_unused]] float* Cf = reinterpret_cast<float*>(smem + 2 * HP + 2 * BTILE);   // [3 planes][BN_NT][Cin]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int r = lane & 31, h = lane >> 5;

    // XCD-aware order: ids b, b + 8, ... share an XCD (its own L2); XCD q walks the q-th contiguous run of tiles, so
    // neighbouring tiles - whose halos overlap by 2*PW + 2 cells - meet in one L2
    const int bq = blockIdx.x >> 3;
    const int tile = (blockIdx.x & 7) * g.tiles_per_xcd + bq / g.ntiles_n;
    if (tile >= g.tiles) return;   // padding block of the last XCD share (whole block, before any barrier)
    const int n0c = (bq % g.ntiles_n) * CO;                     // first output channel of this block
    int grp, kt, c0 = 0, x0 = 0, y0 = 0, n0, nb;
    [[maybe_unused]] int ty = 0, tx = 0;
    if constexpr (RECT) {
        n0 = tile / g.tiles_img;                                // the tile's image
        const int rem = tile - n0 * g.tiles_img;
        ty = rem / g.tiles_x;
        tx = rem - ty * g.tiles_x;
        grp = n0 / g.G;
        kt = (n0 - grp * g.G) * g.tiles_img + rem;              // chunk slot of the statistics partials
        nb = n0;
    } else {
        grp = tile / g.tiles_per_group;
        kt = tile - grp * g.tiles_per_group;
        c0 = kt * HBM_;                                         // first cell of the tile inside its group
        const int R0g = c0 / g.PW;                              // its strip row inside the group
        x0 = c0 - R0g * g.PW;                                   // ... and its column
        const int ng = R0g / g.PH;
        y0 = R0g - ng * g.PH;
        n0 = grp * g.G + ng;                                    // image of the tile's first cell
        nb = n0 > 0 ? n0 - 1 : 0;                               // base image of the buffer resource
    }
    const int pitch = RECT ? RPITCH : g.PW;                     // cells per row of the LDS halo image

    // ---- halo loader: thread (cell = pass * 32 + tid / 8, channel quad = tid % 8)
    const int quad = tid & 7;
    const int64_t ipix = (int64_t)g.H * g.W;
    const int64_t xbytes = ((((int64_t)g.N - nb) * ipix - 1) * g.ldx + g.Cin) * ES;
    __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(
        reinterpret_cast<char*>(const_cast<float*>(x)) + (int64_t)nb * ipix * g.ldx * ES, 0,
        xbytes > 0x7fffffffLL ? 0x7fffffff : (int)xbytes, 0x00020000);
    [[maybe_unused]] __amdgpu_buffer_rsrc_t rs_y = rs_x, rs_dy = rs_x;
    [[maybe_unused]] const int t_lo = BNAP ? nb / g.bn_fps : 0;
    [[maybe_unused]] int cofs[NPASS];      // BNAP: float offset of (timestep of the cell, channel quad) inside a coefficient plane
    [[maybe_unused]] unsigned imask = 0;   // BNAP: bit p = the cell of pass p belongs to the tile (its dy is stored)
    [[maybe_unused]] unsigned vmask = 0;   // BNAP: bit p = the cell of pass p is an image pixel (pad cells stay ZERO: the
                                           // affine's constant term must not leak into the convolution's zero padding)
    if constexpr (BNAP) {   // gx, y and dy_out are dense tensors of one layout (host-checked): one set of offsets
        rs_y = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(bn_y + (int64_t)nb * ipix * g.ldx), 0,
                                                 xbytes > 0x7fffffffLL ? 0x7fffffff : (int)xbytes, 0x00020000);
        rs_dy = __builtin_amdgcn_make_buffer_rsrc(dy_out + (int64_t)nb * ipix * g.ldx, 0,
                                                  xbytes > 0x7fffffffLL ? 0x7fffffff : (int)xbytes, 0x00020000);
        for (int idx = tid; idx < 3 * BN_NT * g.Cin; idx += kThreads) {
            const int pl = idx / (BN_NT * g.Cin), rem = idx - pl * BN_NT * g.Cin;
            const int k = rem / g.Cin, c = rem - k * g.Cin;
            const int t = t_lo + k < g.bn_T ? t_lo + k : g.bn_T - 1;
            Cf[idx] = bn_coef[(int64_t)pl * g.bn_tc + (int64_t)t * g.Cin + c];
        }
    }
    unsigned voff[NPASS];
    int awr[NPASS];      // LDS byte offset (inside a piece) this thread writes for pass p
#pragma unroll
    for (int p = 0; p < NPASS; ++p) {
        const int cell = p * 32 + (tid >> 3);
        int xx, yy, n;
        bool ok;
        if constexpr (RECT) {
            const int rr = cell / RPITCH, cc = cell - rr * RPITCH;
            yy = ty * RTH - 1 + rr;
            xx = tx * RTW - 1 + cc;
            n = n0;
            ok = cell < RCELLS && yy >= 0 && yy < g.H && xx >= 0 && xx < g.W;
        } else {
            // cell c of the halo is strip cell (tile start) + c - PW - 1; counted from column 0 of strip row R0 - 2:
            const unsigned u = (unsigned)(x0 + cell + g.PW - 1);
            const unsigned dR = udiv_small(u, g.magic_pw);
            xx = (int)(u - dR * g.PW);
            // ... and from row 0 of image n0 - 1:
            const unsigned v = (unsigned)(y0 + (int)dR + g.PH - 2);
            const unsigned dn = udiv_small(v, g.magic_ph);
            yy = (int)(v - dn * g.PH);
            n = n0 - 1 + (int)dn;
            ok = n >= 0 && n < g.N && xx < g.W && yy < g.H;
        }
        const int64_t pix = (int64_t)(n - nb) * ipix + (int64_t)yy * g.W + xx;
        voff[p] = ok ? (unsigned)((pix * g.ldx + quad * 4) * ES) : 0x80000000u;   // >= 2 GiB: range check -> zeros
        awr[p] = cell_slot_off(cell, quad >> 1) + (quad & 1) * 8;
        if constexpr (BNAP) {
            int ts = ok ? n / g.bn_fps - t_lo : 0;
            ts = ts < BN_NT ? ts : BN_NT - 1;   // (never taken for host-accepted shapes)
            cofs[p] = ts * g.Cin + quad * 4;
            bool mine;
            if constexpr (RECT) {
                const int rr = cell / RPITCH, cc = cell - rr * RPITCH;
                mine = rr >= 1 && rr <= RTH && cc >= 1 && cc <= RTW;
            } else {
                const int m = cell - (g.PW + 1);
                mine = m >= 0 && m < HBM_ && m < g.group_cells - c0;
            }
            imask |= (ok && mine) ? 1u << p : 0u;
            vmask |= ok ? 1u << p : 0u;
        }
    }

    // The wait that closes a k-step DRAINS the vector-memory queue (vmcnt(0)).  Rounds 3's first form counted instead
    // ("vmcnt(1): everything but the youngest operation - the staging pass of the next chunk - has landed, so the weight tile's
    // LDS-DMA is complete") to leave that pass in flight for a second k-step.  The count is only valid while the youngest
    // operation really completes behind the older DMA, and a buffer load whose lanes are ALL out of range (a pass inside a pad
    // row, past the last image, outside a rectangle's halo; every pass when there is no next chunk) is answered at once: the
    // count was met with the DMA still in flight and the next tap multiplied the PREVIOUS weight tile - on a few tiles per
    // launch, differently from run to run (found on the rectangle form of the stride-2 kernel at 1 Mpx).  Whether a pass
    // that hits in cache can overtake the DMA in the same way is not something to bet results on: draining costs 3 % on the
    // two-chunk 64-channel layers (193 -> 198 us) and nothing elsewhere.
    // ---- fragment addressing
    int cellbase[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i)
        cellbase[i] = RECT ? (wm * TM + i + 1) * RPITCH + 1 + r : g.PW + 1 + (wm * TM + i) * 32 + r;
    const int nchunks = g.Cin >> 5;
    const int co_tiles = g.Cout >> 5;
    // weight image: [tap][chunk][co tile][k16][piece][lane] x 16 B; this wave copies pieces wave, wave + 4, ...
    const unsigned char* wsrc = wimg + (int64_t)(n0c >> 5) * 4096 + lane * 16;
    // LDS-DMA in inline asm: hipcc's wait-count bookkeeping turns conservative around the builtin (it drains vmcnt(0)
    // before every reuse of a load register, i.e. right behind the DMA it should leave in flight); hidden from it, the
    // compiler counts only its own loads - which errs on the early side - and the DMA's completion is counted by hand
    // (the vmcnt of the k-step's closing statement).  M0 (the LDS destination) is written in the statement that uses it.
    const unsigned lds_b = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)Bimg;
    auto dma_b = [&](int kk, int buf) {   // kk = chunk * 9 + tap
        if constexpr (SBF)
            if (wave & 1) return;   // pieces wave, wave + 4, ...: the odd waves hold the LOW pieces, unused with one product
        const int chunk = kk / 9, tap = kk - chunk * 9;
        const unsigned char* src = wsrc + ((int64_t)(tap * nchunks + chunk) * co_tiles) * 4096 + wave * 1024;
        const unsigned dst = lds_b + buf * BTILE + wave * 1024;
#pragma unroll
        for (int q = 0; q < NDMA; ++q) {
            unsigned keep;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep)
                         : "v"(src + q * 4096), "s"(__builtin_amdgcn_readfirstlane(dst + q * 4096))
                         : "memory");
        }
    };

    // a staging pass on its way to LDS: 4 fp32 values, or (SBF) 4 bf16 values as two dwords.  (Integer-typed on purpose:
    // carried in float lanes and bit-cast back element by element, hipcc 7.2 narrows the 8-byte buffer load to 4 bytes.)
    using PReg = typename std::conditional<SBF, u32x2, f32x4>::type;
    PReg pf[NPASS];
    auto store_halo = [&]() {
#pragma unroll
        for (int p = 0; p < NPASS; ++p) {
            if constexpr (RECT)
                if (p * 32 >= HC) continue;   // passes past the rectangle's halo (all their cells are out of range)
            if constexpr (SBF) {
                *reinterpret_cast<u32x2*>(Aimg + awr[p]) = pf[p];
            } else if constexpr (XSP) {   // fp16 16.0 = 0x4C00: the spike times the activation pre-scale
                const u32x2 hi = {(pf[p][0] > g.x_th ? 0x4C00u : 0u) | (pf[p][1] > g.x_th ? 0x4C000000u : 0u),
                                  (pf[p][2] > g.x_th ? 0x4C00u : 0u) | (pf[p][3] > g.x_th ? 0x4C000000u : 0u)};
                *reinterpret_cast<u32x2*>(Aimg + awr[p]) = hi;
            } else {
                u32x2 hi, lo;
                split4<F16>(pf[p], hi, lo);
                *reinterpret_cast<u32x2*>(Aimg + awr[p]) = hi;
                *reinterpret_cast<u32x2*>(Aimg + HP + awr[p]) = lo;
            }
        }
    };
    auto load_pass = [&](int off) -> PReg {   // one staging pass: 4 channels of one halo cell
        if constexpr (SBF) return __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(rs_x, off, 0, 0));
        else return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_x, off, 0, 0));
    };

    // BNAP: gx -> dy for one landed pass (the statement of k_bn_bwd_apply, same roundings) and the store of the tile's own cells
    [[maybe_unused]] f32x4 pfy[2];
    [[maybe_unused]] auto bn_combine = [&](f32x4& gxv, const f32x4& yv, int p, int chan_floats, bool live) {
        const int planes = BN_NT * g.Cin;
        const float* cf = Cf + cofs[p] + chan_floats;
        const f32x4 ca = *reinterpret_cast<const f32x4*>(cf), cb = *reinterpret_cast<const f32x4*>(cf + planes),
                    cc = *reinterpret_cast<const f32x4*>(cf + 2 * planes);
#pragma unroll
        for (int e = 0; e < 4; ++e) gxv[e] = ca[e] * gxv[e] + cb[e] * yv[e] + cc[e];
        if (!((vmask >> p) & 1u)) gxv = f32x4{0.f, 0.f, 0.f, 0.f};
        const unsigned so = (live && ((imask >> p) & 1u)) ? voff[p] + (unsigned)chan_floats * 4u : 0x80000000u;
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, gxv), rs_dy, (int)so, 0, 0);
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    auto kstep = [&](int tapoff, const unsigned char* Bb) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {   // lane (r, h) holds k = 16*ks + 8*h .. +7 of its row
            bf16x8 ah[TM], al[TM], bh[TN], bl[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int off = cell_slot_off(cellbase[i] + tapoff, 2 * ks + h);
                if constexpr (ABL & 16) {
                    if (i == 0 && ks == 0) ah[0] = *reinterpret_cast<const bf16x8*>(Aimg + off);
                    ah[i] = ah[0];
                    al[i] = ah[0];
                    continue;
                }
                ah[i] = *reinterpret_cast<const bf16x8*>(Aimg + off);
                if constexpr (!SBF && !XSP) al[i] = *reinterpret_cast<const bf16x8*>(Aimg + HP + off);
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int off = (wn * TN + j) * 4096 + ks * 2048 + lane * 16;
                bh[j] = *reinterpret_cast<const bf16x8*>(Bb + off);
                if constexpr (!SBF) bl[j] = *reinterpret_cast<const bf16x8*>(Bb + off + 1024);
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {   // small terms first
                    if constexpr (SBF) {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);
                    } else if constexpr (XSP) {
                        const f16x8 xah = __builtin_bit_cast(f16x8, ah[i]);
                        const f16x8 xbh = __builtin_bit_cast(f16x8, bh[j]), xbl = __builtin_bit_cast(f16x8, bl[j]);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(xah, xbl, acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(xah, xbh, acc[i][j], 0, 0, 0);
                    } else if constexpr (F16) {
                        const f16x8 xah = __builtin_bit_cast(f16x8, ah[i]), xal = __builtin_bit_cast(f16x8, al[i]);
                        const f16x8 xbh = __builtin_bit_cast(f16x8, bh[j]), xbl = __builtin_bit_cast(f16x8, bl[j]);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(xal, xbh, acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(xah, xbl, acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(xah, xbh, acc[i][j], 0, 0, 0);
                    } else {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh[j], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl[j], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);
                    }
                }
        }
    };

    // ---- prologue: halo of chunk 0 and the first weight tile
    dma_b(0, 0);
#pragma unroll
    for (int p = 0; p < NPASS; ++p) pf[p] = load_pass((int)voff[p]);
    if constexpr (BNAP) {
        f32x4 py[NPASS];
#pragma unroll
        for (int p = 0; p < NPASS; ++p)
            py[p] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_y, (int)voff[p], 0, 0));
        __syncthreads();   // the coefficient planes are staged
#pragma unroll
        for (int p = 0; p < NPASS; ++p) bn_combine(pf[p], py[p], p, 0, true);
    }
    store_halo();
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");

    const int nk = nchunks * 9;
#pragma unroll 1
    for (int chunk = 0; chunk < nchunks; ++chunk) {
        const bool more = chunk + 1 < nchunks;
        const int cbytes = (chunk + 1) * 32 * ES;   // channel offset of the NEXT chunk
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int kk = chunk * 9 + tap;
            const int cur = kk & 1;
            // the other buffer was read during the previous k-step (every wave has passed that step's barrier)
            if constexpr (!(ABL & 1))
                if (kk + 1 < nk) dma_b(kk + 1, cur ^ 1);
            __builtin_amdgcn_sched_barrier(0);
            // one staging pass of the next chunk's halo per tap (none when there is no next chunk)
            if constexpr (!(ABL & 4)) {
                if (BNAP || more) pf[tap] = load_pass(more ? (int)(voff[tap] + (unsigned)cbytes) : (int)0x80000000u);
            }
            if constexpr (BNAP)
                pfy[tap & 1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                                                              rs_y, more ? (int)(voff[tap] + (unsigned)cbytes) : (int)0x80000000u, 0, 0));
            __builtin_amdgcn_sched_barrier(0);
            const int kh = tap / 3, kw = tap - 3 * kh;
            kstep((kh - 1) * pitch + (kw - 1), Bimg + cur * BTILE);
            // this wave's share of the next weight tile has landed (the queue is drained, see above); the barrier makes every
            // wave's share visible and retires this step's reads of the current buffer
            // (lgkmcnt(0): this wave's fragment reads have really left the LDS before another wave may overwrite them)
            if constexpr (BNAP) {
                // the pass requested during the PREVIOUS tap has landed (it is older than this tap's weight DMA, which this
                // k-step's MFMAs have covered): gx -> dy in place, dy stored for the tile's own cells.  Behind it the queue
                // holds this tap's two loads and that store: the weight DMA is the fourth-youngest operation.
                const int pp = tap >= 1 ? tap - 1 : 0;   // (static after unrolling)
                __builtin_amdgcn_sched_barrier(0);       // not above the MFMAs: its wait would expose this tap's DMA
                if (tap >= 1) bn_combine(pf[pp], pfy[pp & 1], pp, more ? (chunk + 1) * 32 : 0, more);
                __builtin_amdgcn_sched_barrier(0);
                // (this variant drains the queue: besides the two loads it also queues the dy STORE of the previous pass, which
                // is out of range - answered at once, see above - for every pass outside the tile's own cells, so a
                // counted wait cannot tell whether the weight DMA has landed.  The variant is off by default.)
                asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
            } else if constexpr (ABL & 2) asm volatile("" ::: "memory");
            else if constexpr (ABL & 4) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        }
        if (more) {   // every wave is past its last read of this chunk's halo image: swap in the next one
            if constexpr (BNAP) bn_combine(pf[8], pfy[0], 8, (chunk + 1) * 32, true);
            store_halo();
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        }
    }

    // ---- epilogue: transpose the accumulators through LDS (operand images are dead), 16 bytes per lane and store
    constexpr int EW = TN * 32 + 4;    // staged row length in floats
    constexpr int LPR = TN * 8;        // lanes per staged row (4 floats each)
    constexpr int RPP = 64 / LPR;      // rows per pass
    float* stage = reinterpret_cast<float*>(smem) + wave * 32 * EW;
    const bool ovec = g.out_vec != 0;
    const int lrow = lane / LPR, c4 = (lane % LPR) * 4;
    const int nch = n0c + wn * TN * 32 + c4;
    const int cells_left = g.group_cells - c0;     // cells of the group from the tile start on
    double ssum[4] = {0.0, 0.0, 0.0, 0.0}, qsum[4] = {0.0, 0.0, 0.0, 0.0};
    const bool stats = (F16 || SBF) && g.bn_partial != nullptr;
    // cell of the tile -> pixel (image, row, column) and "is stored"
    auto cell_pixel = [&](int m, int64_t& pix) -> bool {
        int xx, yy, n;
        bool ok;
        if constexpr (RECT) {
            yy = ty * RTH + (m >> 5);
            xx = tx * RTW + (m & 31);
            n = n0;
            ok = yy < g.H && xx < g.W && nch < g.Cout;
        } else {
            const unsigned u = (unsigned)(x0 + m);
            const unsigned dR = udiv_small(u, g.magic_pw);
            xx = (int)(u - dR * g.PW);
            const unsigned v = (unsigned)(y0 + (int)dR);
            const unsigned dn = udiv_small(v, g.magic_ph);
            yy = (int)(v - dn * g.PH);
            n = n0 + (int)dn;
            ok = m < cells_left && xx < g.W && yy < g.H && n < g.N && nch < g.Cout;
        }
        pix = ((int64_t)n * g.H + yy) * g.W + xx;
        return ok;
    };
    typedef SnnStore<SBF> St;   // fp32 tensors, or bf16 (rounded here) in the bf16-storage mode
    constexpr int NPS = 32 / RPP;      // store passes per 32-cell row tile
    // Data gradient with fused addends (gradient accumulation): the addend rows of a WHOLE row tile are requested before the
    // accumulators go through LDS - branch-free, lanes without a pixel read element 0 - so their memory latency runs beside
    // the staging instead of once per store pass (the data gradient took the forward kernel's time PLUS the addends'
    // HBM time: 204 vs 160 us on the 128-channel layers, 367 vs 276 us on the 32-channel ones).
    const bool prefetch_add = ovec && (addend != nullptr || addend2 != nullptr) && !(ABL & 8);
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        f32x4 pa1[NPS], pa2[NPS];
        int64_t ppix[NPS];
        unsigned okmask = 0;
        if (prefetch_add) {
#pragma unroll
            for (int pass = 0; pass < NPS; ++pass) {
                const bool ok = cell_pixel((wm * TM + i) * 32 + pass * RPP + lrow, ppix[pass]);
                okmask |= ok ? 1u << pass : 0u;
                const int64_t pc = ok ? ppix[pass] : 0;
                const int nc = ok ? nch : 0;
                const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
                pa1[pass] = addend ? St::ld4_last(addend, pc * g.ld_add + nc) : zero;
                pa2[pass] = addend2 ? St::ld4_last(addend2, pc * g.ld_add2 + nc) : zero;
            }
        }
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e)
                stage[((e & 3) + 8 * (e >> 2) + 4 * h) * EW + j * 32 + r] = F16 ? acc[i][j][e] * kF16Unscale : acc[i][j][e];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // wave-private staging: no barrier needed, only the LDS order
        if (prefetch_add) {
#pragma unroll
            for (int pass = 0; pass < NPS; ++pass) {
                f32x4 val = *reinterpret_cast<const f32x4*>(&stage[(pass * RPP + lrow) * EW + c4]);
                if (addend) val += pa1[pass];        // (same order of the two additions as the plain path)
                if (addend2) val += pa2[pass];
                if ((okmask >> pass) & 1u) St::st4(y, ppix[pass] * g.ldy + nch, val);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            continue;
        }
#pragma unroll
        for (int pass = 0; pass < NPS; ++pass) {
            const int row = pass * RPP + lrow;
            int64_t pix;
            const bool ok = cell_pixel((wm * TM + i) * 32 + row, pix);
            f32x4 val = *reinterpret_cast<const f32x4*>(&stage[row * EW + c4]);
            if (!ok) continue;
            if constexpr (ABL & 8) {
                asm volatile("" :: "v"(val[0]), "v"(val[1]), "v"(val[2]), "v"(val[3]));
                continue;
            }
            if (ovec) {
                if (addend) val += St::ld4_last(addend, pix * g.ld_add + nch);   // fused accumulation
                if (addend2) val += St::ld4_last(addend2, pix * g.ld_add2 + nch);
                St::st4(y, pix * g.ldy + nch, val);
            } else {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float o = val[q];
                    if (addend) o += St::ld1(addend, pix * g.ld_add + nch + q);
                    if (addend2) o += St::ld1(addend2, pix * g.ld_add2 + nch + q);
                    St::st1(y, pix * g.ldy + nch + q, o);
                }
            }
            if (stats) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const double d = (double)val[q];
                    ssum[q] += d;
                    qsum[q] = fma(d, d, qsum[q]);
                }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    if (stats) {
        // BatchNorm partials of the STORED values: lanes that share a channel quad (same lane % LPR) are added in the wave
        // (partner8 / 16 / 32: the sum stands in lanes 48 .. 48 + LPR - 1), the row waves through LDS - a fixed order, run to run
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if constexpr (LPR <= 8) {
                ssum[q] += partner8(ssum[q]);
                qsum[q] += partner8(qsum[q]);
            }
            ssum[q] += partner16(ssum[q]);
            qsum[q] += partner16(qsum[q]);
            ssum[q] += partner32(ssum[q]);
            qsum[q] += partner32(qsum[q]);
        }
        // (a region of its own, not the staging rows: no barrier until the partials are written)
        double* red = reinterpret_cast<double*>(smem + 2 * HP + 2 * BTILE + CF_BYTES);   // [wave][LPR * 4 channels][2]
        if (lane >= 48 && lane < 48 + LPR) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                red[((wave * LPR + lane - 48) * 4 + q) * 2 + 0] = ssum[q];
                red[((wave * LPR + lane - 48) * 4 + q) * 2 + 1] = qsum[q];
            }
        }
        __syncthreads();
        if (tid < CO) {   // channel n0c + tid: column wave wn = tid / (TN*32), both row waves in order
            const int wn_ = tid / (TN * 32), cc = tid % (TN * 32);
            double s = 0.0, q2 = 0.0;
#pragma unroll
            for (int wm_ = 0; wm_ < WM; ++wm_) {
                const double* src = red + (((wm_ * WN + wn_) * LPR * 4) + cc) * 2;
                s += src[0];
                q2 += src[1];
            }
            if (n0c + tid < g.Cout) {
                double* dstp = g.bn_partial + snn_bn_partial_index(grp, kt, n0c + tid, g.tiles_per_group, g.Cout);
                dstp[0] = s;
                dstp[1] = q2;
            }
        }
    }
}

// ---- data gradient of the 3x3 / stride 2 / pad 1 convolutions in ONE pass over dy.
//     dx[n, hi, wi, ci] = sum_{kh,kw} dy[n, (hi+1-kh)/2, (wi+1-kw)/2, co] * w[co, kh, kw, ci]     (exact divisions only)
// The implicit GEMM runs one launch per stride-phase class (hi % 2, wi % 2): four launches that each gather dy again
// (PMC, profiles/r02: 1.4x the algorithmic bytes) with K loops of only 1, 2, 2 and 4 taps - prologue and epilogue
// dominate (64 -> 128 at 120x152: 643 us against 383 us for the forward of the same layer).  Here a block owns 128
// consecutive cells of dy's padded strip (rows of PW = Wo + 1 cells, see k_conv_halo3) and produces ALL FOUR classes
// of the 2x2 dx pixels under them: per 32-channel chunk the dy halo (the tile plus PW + 1 cells BEHIND it - the taps
// reach down / right only) is staged once, and the nine taps are nine k-steps that each add into the accumulator set of
// their class: kh = 1 -> even rows from dy row a; kh = 0 -> odd rows from dy row a + 1; kh = 2 -> odd rows from dy row a
// (columns alike).  Four accumulator sets of 32 cells x 64 channels per wave (128 registers), 4 waves over the cells,
// 64 dx channels per block.  The weight image is the stride-1 data-gradient image (mirrored taps): tap t is read at 8 - t.
// RECT (dy rows longer than 157 cells: the strip halo would not fit): a block owns a 4 x 32 rectangle of dy cells of ONE image,
// one row per wave; the staged halo is that rectangle plus one row below and one column to the right (5 x 33 cells).
constexpr int SPITCH = RTW + 1, SCELLS = (RTH + 1) * SPITCH;
template <bool SBF = false, bool RECT = false>   // SBF: dy, dx and the addends are bf16 tensors, one product (see k_conv_halo3)
__global__ __launch_bounds__(kThreads, 2) void k_conv_s2dgrad3(const float* __restrict__ x,   // dy
                                                              const unsigned char* __restrict__ wimg,
                                                              float* __restrict__ y,         // dx
                                                              HaloGeom g, const float* __restrict__ addend,
                                                              const float* __restrict__ addend2) {
    constexpr int CO = 64, TN = 2;
    constexpr int ES = SBF ? 2 : 4;
    constexpr int BTILE = (CO / 32) * 4096;
    constexpr int NDMA = (CO / 32) * 4 / 4;
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * HPIECE + 2 * BTILE];
    unsigned char* Aimg = smem;
    unsigned char* Bimg = smem + 2 * HPIECE;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // = row tile of the wave (4 x 32 cells)
    const int r = lane & 31, h = lane >> 5;

    const int bq = blockIdx.x >> 3;
    const int tile = (blockIdx.x & 7) * g.tiles_per_xcd + bq / g.ntiles_n;
    if (tile >= g.tiles) return;
    const int n0c = (bq % g.ntiles_n) * CO;
    int c0 = 0, x0 = 0, y0 = 0, n0;
    [[maybe_unused]] int ty = 0, tx = 0;
    if constexpr (RECT) {
        n0 = tile / g.tiles_img;
        const int rem = tile - n0 * g.tiles_img;
        ty = rem / g.tiles_x;
        tx = rem - ty * g.tiles_x;
    } else {
        c0 = tile * HBM_;                                        // one group: all N images
        const int R0g = c0 / g.PW;
        x0 = c0 - R0g * g.PW;
        n0 = R0g / g.PH;
        y0 = R0g - n0 * g.PH;
    }
    const int nb = n0;                                           // the halo reaches forward only
    const int pitch = RECT ? SPITCH : g.PW;

    const int quad = tid & 7;
    const int64_t ipix = (int64_t)g.H * g.W;
    const int64_t xbytes = ((((int64_t)g.N - nb) * ipix - 1) * g.ldx + g.Cin) * ES;
    __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(
        reinterpret_cast<char*>(const_cast<float*>(x)) + (int64_t)nb * ipix * g.ldx * ES, 0,
        xbytes > 0x7fffffffLL ? 0x7fffffff : (int)xbytes, 0x00020000);
    unsigned voff[NPASS];
    int awr[NPASS];
#pragma unroll
    for (int p = 0; p < NPASS; ++p) {
        const int cell = p * 32 + (tid >> 3);                    // halo cell = strip cell (tile start) + cell
        int xx, yy, n;
        bool ok;
        if constexpr (RECT) {
            const int rr = cell / SPITCH, cc = cell - rr * SPITCH;
            yy = ty * RTH + rr;
            xx = tx * RTW + cc;
            n = n0;
            ok = cell < SCELLS && xx < g.W && yy < g.H;
        } else {
            const unsigned u = (unsigned)(x0 + cell);
            const unsigned dR = udiv_small(u, g.magic_pw);
            xx = (int)(u - dR * g.PW);
            const unsigned v = (unsigned)(y0 + (int)dR);
            const unsigned dn = udiv_small(v, g.magic_ph);
            yy = (int)(v - dn * g.PH);
            n = n0 + (int)dn;
            ok = n < g.N && xx < g.W && yy < g.H;
        }
        const int64_t pix = (int64_t)(n - nb) * ipix + (int64_t)yy * g.W + xx;
        voff[p] = ok ? (unsigned)((pix * g.ldx + quad * 4) * ES) : 0x80000000u;
        awr[p] = cell_slot_off(cell, quad >> 1) + (quad & 1) * 8;
    }
    const int cellbase = RECT ? wave * SPITCH + r : wave * 32 + r;
    const int nchunks = g.Cin >> 5;
    const int co_tiles = g.Cout >> 5;
    const unsigned char* wsrc = wimg + (int64_t)(n0c >> 5) * 4096 + lane * 16;
    const unsigned lds_b = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)Bimg;
    auto dma_b = [&](int kk, int buf) {   // kk = chunk * 9 + tap; the image holds mirrored taps
        if constexpr (SBF)
            if (wave & 1) return;   // the odd waves' pieces are the LOW pieces: unused with one product
        const int chunk = kk / 9, tap = kk - chunk * 9;
        const unsigned char* src = wsrc + ((int64_t)((8 - tap) * nchunks + chunk) * co_tiles) * 4096 + wave * 1024;
        const unsigned dst = lds_b + buf * BTILE + wave * 1024;
#pragma unroll
        for (int q = 0; q < NDMA; ++q) {
            unsigned keep;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep)
                         : "v"(src + q * 4096), "s"(__builtin_amdgcn_readfirstlane(dst + q * 4096))
                         : "memory");
        }
    };
    using PReg = typename std::conditional<SBF, u32x2, f32x4>::type;   // (integer-typed with SBF: see k_conv_halo3)
    PReg pf[NPASS];
    auto store_halo = [&]() {
#pragma unroll
        for (int p = 0; p < NPASS; ++p) {
            if constexpr (SBF) {
                *reinterpret_cast<u32x2*>(Aimg + awr[p]) = pf[p];
            } else {
                u32x2 hi, lo;
                split4<false>(pf[p], hi, lo);
                *reinterpret_cast<u32x2*>(Aimg + awr[p]) = hi;
                *reinterpret_cast<u32x2*>(Aimg + HPIECE + awr[p]) = lo;
            }
        }
    };
    auto load_pass = [&](int off) -> PReg {
        if constexpr (SBF) return __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(rs_x, off, 0, 0));
        else return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_x, off, 0, 0));
    };
    f32x16 acc[4][TN];   // [class = 2 * (hi % 2) + (wi % 2)]
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[c][j][e] = 0.f;

    auto kstep = [&](int tapoff, const unsigned char* Bb, f32x16 (&a)[TN]) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int off = cell_slot_off(cellbase + tapoff, 2 * ks + h);
            const bf16x8 ah = *reinterpret_cast<const bf16x8*>(Aimg + off);
            bf16x8 al = ah;
            if constexpr (!SBF) al = *reinterpret_cast<const bf16x8*>(Aimg + HPIECE + off);
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int boff = j * 4096 + ks * 2048 + lane * 16;
                const bf16x8 bh = *reinterpret_cast<const bf16x8*>(Bb + boff);
                if constexpr (!SBF) {
                    const bf16x8 bl = *reinterpret_cast<const bf16x8*>(Bb + boff + 1024);
                    a[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, a[j], 0, 0, 0);
                    a[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, a[j], 0, 0, 0);
                }
                a[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, a[j], 0, 0, 0);
            }
        }
    };

    dma_b(0, 0);
#pragma unroll
    for (int p = 0; p < NPASS; ++p) pf[p] = load_pass((int)voff[p]);
    store_halo();
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");

    const int nk = nchunks * 9;
#pragma unroll 1
    for (int chunk = 0; chunk < nchunks; ++chunk) {
        const bool more = chunk + 1 < nchunks;
        const int cbytes = (chunk + 1) * 32 * ES;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int kk = chunk * 9 + tap;
            const int cur = kk & 1;
            if (kk + 1 < nk) dma_b(kk + 1, cur ^ 1);
            __builtin_amdgcn_sched_barrier(0);
            // one staging pass of the next chunk's halo per tap; the wait below drains the queue (see k_conv_halo3: a counted
            // wait is not valid behind a pass whose lanes are all out of range - this kernel's rectangle form showed it)
            if (more) pf[tap] = load_pass((int)(voff[tap] + (unsigned)cbytes));
            __builtin_amdgcn_sched_barrier(0);
            const int kh = tap / 3, kw = tap - 3 * kh;
            // kh = 1: even dx rows, dy row a; kh = 0: odd rows, dy row a + 1; kh = 2: odd rows, dy row a (columns alike)
            const int ph = kh == 1 ? 0 : 1, pw = kw == 1 ? 0 : 1;
            const int dh = kh == 0 ? 1 : 0, dw = kw == 0 ? 1 : 0;
            kstep(dh * pitch + dw, Bimg + cur * BTILE, acc[2 * ph + pw]);
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        }
        if (more) {
            store_halo();
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        }
    }

    // ---- epilogue: per class, the wave's 32 cells x 64 channels through LDS to 16-byte stores at dx pixel (2a+ph, 2b+pw)
    constexpr int EW = TN * 32 + 4, LPR = TN * 8, RPP = 64 / LPR;
    float* stage = reinterpret_cast<float*>(smem) + wave * 32 * EW;
    const bool ovec = g.out_vec != 0;
    const int lrow = lane / LPR, c4 = (lane % LPR) * 4;
    const int nch = n0c + c4;
    const int cells_left = g.group_cells - c0;
#pragma unroll
    for (int cls = 0; cls < 4; ++cls) {
        const int ph = cls >> 1, pw = cls & 1;
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e)
                stage[((e & 3) + 8 * (e >> 2) + 4 * h) * EW + j * 32 + r] = acc[cls][j][e];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int pass = 0; pass < 32 / RPP; ++pass) {
            const int row = pass * RPP + lrow;
            const int m = wave * 32 + row;
            int aa, bb, n;
            bool inside;
            if constexpr (RECT) {
                aa = ty * RTH + wave;
                bb = tx * RTW + row;
                n = n0;
                inside = true;
            } else {
                const unsigned u = (unsigned)(x0 + m);
                const unsigned dR = udiv_small(u, g.magic_pw);
                bb = (int)(u - dR * g.PW);
                const unsigned v = (unsigned)(y0 + (int)dR);
                const unsigned dn = udiv_small(v, g.magic_ph);
                aa = (int)(v - dn * g.PH);
                n = n0 + (int)dn;
                inside = m < cells_left && n < g.N;
            }
            const int hi = 2 * aa + ph, wi = 2 * bb + pw;
            const bool ok = inside && bb < g.W && aa < g.H && hi < g.OH && wi < g.OW && nch < g.Cout;
            f32x4 val = *reinterpret_cast<const f32x4*>(&stage[row * EW + c4]);
            if (!ok) continue;
            const int64_t pix = ((int64_t)n * g.OH + hi) * g.OW + wi;
            typedef SnnStore<SBF> St;
            if (ovec) {
                if (addend) val += St::ld4_last(addend, pix * g.ld_add + nch);
                if (addend2) val += St::ld4_last(addend2, pix * g.ld_add2 + nch);
                St::st4(y, pix * g.ldy + nch, val);
            } else {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float o = val[q];
                    if (addend) o += St::ld1(addend, pix * g.ld_add + nch + q);
                    if (addend2) o += St::ld1(addend2, pix * g.ld_add2 + nch + q);
                    St::st1(y, pix * g.ldy + nch + q, o);
                }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
}

// ---- weight image in MFMA-fragment order.  src: [O][3][3][I] fp32 (the forward's OHWI weights, or the transposed
// [Cin][KH][KW][Cout] matrix for the data gradient with flip = 1: tap t of the image is tap 8 - t of src).
// image: [tap][I/32][O/32][k16 (2)][piece (2)][lane (64)] x 16 bytes; lane (r, h) holds src[o = 32*ot + r][tap][i = 32*ic +
// 16*ks + 8*h .. + 7] as 8 hi pieces / 8 lo pieces.  One thread per (tap, ic, ot, ks, lane): two 16-byte loads, two stores.
template <bool F16>
__global__ void k_weight_frag_image(const float* __restrict__ flat_src, unsigned char* __restrict__ flat_dst,
                                    const int64_t* __restrict__ table, int flip) {
    // table row: {float offset of src in flat_src, byte offset of the image in flat_dst, O, I}
    const int64_t* row = table + (int64_t)blockIdx.y * 4;
    const float* src = flat_src + row[0];
    unsigned char* dst = flat_dst + row[1];
    const int O = (int)row[2], I = (int)row[3];
    const int ot_n = O >> 5, ic_n = I >> 5;
    const int64_t total = (int64_t)9 * ic_n * ot_n * 2 * 64;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int lane = (int)(idx & 63);
        const int ks = (int)((idx >> 6) & 1);
        int64_t rest = idx >> 7;
        const int ot = (int)(rest % ot_n); rest /= ot_n;
        const int ic = (int)(rest % ic_n);
        const int tap = (int)(rest / ic_n);
        const int r = lane & 31, h = lane >> 5;
        const int st = flip ? 8 - tap : tap;
        const float* p = src + ((int64_t)(ot * 32 + r) * 9 + st) * I + ic * 32 + ks * 16 + h * 8;
        const f32x4 v0 = *reinterpret_cast<const f32x4*>(p), v1 = *reinterpret_cast<const f32x4*>(p + 4);
        u32x4 hi, lo;
#pragma unroll
        for (int e = 0; e < 8; e += 2) {
            const float a0 = e < 4 ? v0[e] : v1[e - 4], a1 = e < 4 ? v0[e + 1] : v1[e - 3];
            if (F16) {
                const float a = a0 * kF16WeightScale, b = a1 * kF16WeightScale;
                const f16x2 ph = __builtin_convertvector(f32x2{a, b}, f16x2);
                const f16x2 pl = __builtin_convertvector(f32x2{a - (float)ph[0], b - (float)ph[1]}, f16x2);
                hi[e >> 1] = __builtin_bit_cast(unsigned, ph);
                lo[e >> 1] = __builtin_bit_cast(unsigned, pl);
            } else {
                f32x2 rr = {a0, a1};
                const bf16x2 ph = __builtin_convertvector(rr, bf16x2);
                const unsigned bits = __builtin_bit_cast(unsigned, ph);
                rr[0] -= __builtin_bit_cast(float, bits << 16);
                rr[1] -= __builtin_bit_cast(float, bits & 0xffff0000u);
                const bf16x2 pl = __builtin_convertvector(rr, bf16x2);
                hi[e >> 1] = bits;
                lo[e >> 1] = __builtin_bit_cast(unsigned, pl);
            }
        }
        unsigned char* o = dst + ((((int64_t)(tap * ic_n + ic) * ot_n + ot) * 2 + ks) * 2) * 1024 + lane * 16;
        *reinterpret_cast<u32x4*>(o) = hi;
        *reinterpret_cast<u32x4*>(o + 1024) = lo;
    }
}

static unsigned magic_u32(int d) { return (unsigned)((0x100000000ULL + (unsigned)d - 1) / (unsigned)d); }
static bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }
static bool aligned8(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 7u) == 0; }
// 4 consecutive elements of an activation tensor per access: 16 bytes, or 8 with bf16 storage
static bool out_aligned(const void* p, bool bf16) { return bf16 ? aligned8(p) : aligned16(p); }

// 0: not covered, 1: padded-strip tiles (rows of <= 78 pixels), 2: 4 x 32 rectangles (any width)
static int halo_mode(int64_t N, int H, int W, int Cin, int Cout) {
    if (N <= 0 || H <= 0 || W <= 0) return 0;
    if (Cin % 32 != 0 || Cin < 32 || (Cout % 64 != 0 && Cout != 32)) return 0;
    if (Cout == 32 && snn_tuning_env("SNN_HALO_NO_CO32")) return 0;   // tuning / bisecting aid
    if (W + 1 <= 79) {                                              // halo of a 128-cell tile: 128 + 2*PW + 2 <= 288 cells
        if (N * (int64_t)(H + 1) * (W + 1) >= 0x7fffffffLL) return 0;   // strip cells of a group in 32 bits
        return 1;
    }
    const int64_t tiles = N * snn_ceil_div(H, RTH) * snn_ceil_div(W, RTW);
    return tiles < 0x7fffffffLL ? 2 : 0;
}
static bool halo_shape_ok(int64_t N, int H, int W, int Cin, int Cout) { return halo_mode(N, H, W, Cin, Cout) != 0; }

// strip tiles need the tile and the PW + 1 cells behind it inside the staged halo; wider rows take 4 x 32 rectangles
static bool s2dgrad_rect(int Wo) { return 128 + (Wo + 1) + 2 > HCELLS; }
static bool s2dgrad_shape_ok(int64_t N, int H, int W, int Cin, int Ho, int Wo, int Cout) {
    // Cin: the layer's input channels (dx), Cout: its output channels (dy, the K dimension)
    if (N <= 0 || H <= 0 || W <= 0) return false;
    if (Ho != (H - 1) / 2 + 1 || Wo != (W - 1) / 2 + 1) return false;     // 3x3, stride 2, pad 1
    if (Cout % 32 != 0 || Cout < 32 || Cin % 64 != 0) return false;
    if (s2dgrad_rect(Wo)) return N * snn_ceil_div(Ho, RTH) * snn_ceil_div(Wo, RTW) < 0x7fffffffLL;   // rectangles: any width
    if (N * (int64_t)(Ho + 1) * (Wo + 1) >= 0x7fffffffLL) return false;
    return true;
}
}  // namespace

extern "C" int snn_conv3x3_s2_dgrad_supported(int64_t N, int H, int W, int Cin, int Ho, int Wo, int Cout) {
    return s2dgrad_shape_ok(N, H, W, Cin, Ho, Wo, Cout) ? 1 : 0;
}

extern "C" int snn_conv3x3_s2_dgrad(const float* dy, int64_t lddy, const void* wt_image, float* dx, int64_t lddx, int64_t N,
                                    int H, int W, int Cin, int Ho, int Wo, int Cout, const float* addend, int64_t ld_addend,
                                    const float* addend2, int64_t ld_addend2, int precision, void* stream) {
    SNN_REQUIRE(dy && wt_image && dx, "snn_conv3x3_s2_dgrad: null pointer");
    SNN_REQUIRE(precision == SNN_PREC_BF16X3 || precision == SNN_PREC_BF16S,
                "snn_conv3x3_s2_dgrad: precision must be SNN_PREC_BF16X3 or SNN_PREC_BF16S (got %d)", precision);
    const bool sbf = precision == SNN_PREC_BF16S;
    SNN_REQUIRE(s2dgrad_shape_ok(N, H, W, Cin, Ho, Wo, Cout),
                "snn_conv3x3_s2_dgrad: shape not covered (N %lld, %dx%d -> %dx%d, %d -> %d channels; ask "
                "snn_conv3x3_s2_dgrad_supported)", (long long)N, H, W, Ho, Wo, Cin, Cout);
    SNN_REQUIRE(lddy >= Cout && lddx >= Cin && lddy % 4 == 0, "snn_conv3x3_s2_dgrad: bad pixel strides (%lld, %lld)",
                (long long)lddy, (long long)lddx);
    SNN_REQUIRE((sbf ? aligned8(dy) : aligned16(dy)) && aligned16(wt_image),
                "snn_conv3x3_s2_dgrad: dy (16 bytes; 8 for bf16) and the weight image (16) must be aligned");
    SNN_REQUIRE(!addend || ld_addend >= Cin, "snn_conv3x3_s2_dgrad: addend pixel stride smaller than channel count");
    SNN_REQUIRE(!addend2 || ld_addend2 >= Cin, "snn_conv3x3_s2_dgrad: addend2 pixel stride smaller than channel count");
    HaloGeom g;
    g.ldx = lddy; g.ldy = lddx; g.ld_add = ld_addend; g.ld_add2 = ld_addend2;
    g.N = (int)N; g.H = Ho; g.W = Wo;          // the strip grid is dy's
    g.Cin = Cout;                              // K: dy channels
    g.Cout = Cin;                              // produced channels: dx
    g.OH = H; g.OW = W;
    g.PW = Wo + 1; g.PH = Ho + 1;
    g.G = (int)N;
    const bool rect = s2dgrad_rect(Wo);
    g.tiles_x = (int)snn_ceil_div(Wo, RTW);
    g.tiles_img = g.tiles_x * (int)snn_ceil_div(Ho, RTH);
    if (rect) {
        g.group_cells = 0;
        g.tiles_per_group = g.tiles_img;
        g.tiles = (int)(N * g.tiles_img);
        SNN_REQUIRE((int64_t)Ho * Wo * lddy * 4 < 0x7fffffffLL, "snn_conv3x3_s2_dgrad: a dy image must span less than 2 GiB");
    } else {
        SNN_REQUIRE((int64_t)4 * Ho * Wo * lddy * 4 < 0x7fffffffLL, "snn_conv3x3_s2_dgrad: four dy images must span less than 2 GiB");
        const int64_t cells = N * (int64_t)g.PH * g.PW;
        g.group_cells = (int)cells;
        g.tiles_per_group = (int)snn_ceil_div(cells, HBM_);
        g.tiles = g.tiles_per_group;
    }
    g.ntiles_n = Cin / 64;
    SNN_REQUIRE((int64_t)g.tiles * g.ntiles_n + 8 < 0x7fffffffLL, "snn_conv3x3_s2_dgrad: grid too large");
    g.tiles_per_xcd = (int)snn_ceil_div(g.tiles, 8);
    g.magic_pw = magic_u32(g.PW); g.magic_ph = magic_u32(g.PH);
    g.out_vec = (lddx % 4 == 0) && out_aligned(dx, sbf) && (!addend || (ld_addend % 4 == 0 && out_aligned(addend, sbf))) &&
                (!addend2 || (ld_addend2 % 4 == 0 && out_aligned(addend2, sbf)));
    g.bn_partial = nullptr;
    g.x_th = 0.0f;
    g.bn_T = g.bn_tc = 0; g.bn_fps = 1;
    dim3 grid((unsigned)((int64_t)g.tiles_per_xcd * 8 * g.ntiles_n));
    const unsigned char* wi = static_cast<const unsigned char*>(wt_image);
    if (sbf && rect)
        hipLaunchKernelGGL((k_conv_s2dgrad3<true, true>), grid, dim3(kThreads), 0, (hipStream_t)stream, dy, wi, dx, g, addend, addend2);
    else if (sbf)
        hipLaunchKernelGGL((k_conv_s2dgrad3<true, false>), grid, dim3(kThreads), 0, (hipStream_t)stream, dy, wi, dx, g, addend, addend2);
    else if (rect)
        hipLaunchKernelGGL((k_conv_s2dgrad3<false, true>), grid, dim3(kThreads), 0, (hipStream_t)stream, dy, wi, dx, g, addend, addend2);
    else
        hipLaunchKernelGGL((k_conv_s2dgrad3<false, false>), grid, dim3(kThreads), 0, (hipStream_t)stream, dy, wi, dx, g, addend, addend2);
    SNN_CHECK_LAUNCH("snn_conv3x3_s2_dgrad");
    return 0;
}

extern "C" int snn_conv3x3_halo_supported(int64_t N, int H, int W, int Cin, int Cout) {
    return halo_shape_ok(N, H, W, Cin, Cout) ? 1 : 0;
}

// chunk slots per timestep of the statistics partials snn_conv3x3_halo writes (= its tiles per group)
extern "C" int64_t snn_conv3x3_halo_bn_chunks(int frames_per_step, int H, int W) {
    if (W + 1 <= 79) return snn_ceil_div((int64_t)frames_per_step * (H + 1) * (W + 1), HBM_);
    return (int64_t)frames_per_step * snn_ceil_div(H, RTH) * snn_ceil_div(W, RTW);
}

extern "C" size_t snn_weight_frag_image_bytes(int O, int I) { return (size_t)9 * O * I * 4; }

extern "C" int snn_weight_frag_image_batched(const float* flat_src, void* flat_dst, const int64_t* table, int n,
                                             int max_groups, int flip, int precision, void* stream) {
    SNN_REQUIRE(flat_src && flat_dst && table && n > 0 && max_groups > 0, "snn_weight_frag_image_batched: bad arguments");
    SNN_REQUIRE(precision == SNN_PREC_FP16X3 || precision == SNN_PREC_BF16X3,
                "snn_weight_frag_image: precision must be SNN_PREC_FP16X3 (forward) or SNN_PREC_BF16X3 (data gradient)");
    SNN_REQUIRE(aligned16(flat_src) && aligned16(flat_dst), "snn_weight_frag_image: buffers must be 16-byte aligned");
    int64_t bx = snn_ceil_div(max_groups, kThreads);
    if (bx > 64) bx = 64;
    dim3 grid((unsigned)bx, (unsigned)n);
    if (precision == SNN_PREC_FP16X3)
        hipLaunchKernelGGL(k_weight_frag_image<true>, grid, dim3(kThreads), 0, (hipStream_t)stream, flat_src,
                           static_cast<unsigned char*>(flat_dst), table, flip);
    else
        hipLaunchKernelGGL(k_weight_frag_image<false>, grid, dim3(kThreads), 0, (hipStream_t)stream, flat_src,
                           static_cast<unsigned char*>(flat_dst), table, flip);
    SNN_CHECK_LAUNCH("snn_weight_frag_image_batched");
    return 0;
}

extern "C" int snn_conv3x3_halo_bn_supported(int64_t N, int H, int W, int Cin, int Cout, int frames_per_step) {
    // Cin: channels of gx / y / dy (the K dimension), Cout: channels of dx
    if (halo_mode(N, H, W, Cin, Cout) != 1 || Cin > BN_CMAX || frames_per_step <= 0 || N % frames_per_step != 0) return 0;
    if (Cout != 64 && Cout != 128) return 0;   // ONE channel tile: the block that computes dx also stores dy
    // images a 288-cell halo can touch, and the timesteps they belong to: at most BN_NT
    const int64_t images = HCELLS / ((int64_t)(H + 1) * (W + 1)) + 2;
    return ((images + frames_per_step - 1) / frames_per_step + 1 <= BN_NT) ? 1 : 0;
}

extern "C" int snn_conv3x3_halo_bn(const float* gx, const float* y, const float* coef, int frames_per_step, float* dy_out,
                                   const void* wt_image, float* dx, int64_t lddx, int64_t N, int H, int W, int Cin,
                                   int Cout, const float* addend, int64_t ld_addend, const float* addend2,
                                   int64_t ld_addend2, void* stream) {
    SNN_REQUIRE(gx && y && coef && dy_out && wt_image && dx, "snn_conv3x3_halo_bn: null pointer");
    SNN_REQUIRE(snn_conv3x3_halo_bn_supported(N, H, W, Cin, Cout, frames_per_step),
                "snn_conv3x3_halo_bn: shape not covered (N %lld, %dx%d, %d -> %d channels, %d frames per step; ask "
                "snn_conv3x3_halo_bn_supported)", (long long)N, H, W, Cin, Cout, frames_per_step);
    SNN_REQUIRE(lddx >= Cout, "snn_conv3x3_halo_bn: dx pixel stride smaller than channel count");
    SNN_REQUIRE(aligned16(gx) && aligned16(y) && aligned16(dy_out) && aligned16(wt_image) && aligned16(coef),
                "snn_conv3x3_halo_bn: operands must be 16-byte aligned");
    SNN_REQUIRE(!addend || ld_addend >= Cout, "snn_conv3x3_halo_bn: addend pixel stride smaller than channel count");
    SNN_REQUIRE(!addend2 || ld_addend2 >= Cout, "snn_conv3x3_halo_bn: addend2 pixel stride smaller than channel count");
    SNN_REQUIRE((int64_t)4 * H * W * Cin * 4 < 0x7fffffffLL, "snn_conv3x3_halo_bn: four images must span less than 2 GiB");
    HaloGeom g;
    g.ldx = Cin; g.ldy = lddx; g.ld_add = ld_addend; g.ld_add2 = ld_addend2;     // gx, y, dy_out: dense [N][H][W][Cin]
    g.N = (int)N; g.H = H; g.W = W; g.Cin = Cin; g.Cout = Cout;
    g.PW = W + 1; g.PH = H + 1;
    g.G = (int)N;
    const int64_t group_cells = (int64_t)g.G * g.PH * g.PW;
    g.group_cells = (int)group_cells;
    g.tiles_per_group = (int)snn_ceil_div(group_cells, HBM_);
    g.tiles = g.tiles_per_group;
    const int co_tile = Cout % 128 == 0 ? 128 : 64;
    g.ntiles_n = Cout / co_tile;
    // every channel tile would store the same dy: only ONE may, so the variant is for layers that fit one tile
    SNN_REQUIRE(g.ntiles_n == 1, "snn_conv3x3_halo_bn: %d output channels need more than one channel tile", Cout);
    g.tiles_per_xcd = (int)snn_ceil_div(g.tiles, 8);
    g.magic_pw = magic_u32(g.PW); g.magic_ph = magic_u32(g.PH);
    g.out_vec = (lddx % 4 == 0) && aligned16(dx) && (!addend || (ld_addend % 4 == 0 && aligned16(addend))) &&
                (!addend2 || (ld_addend2 % 4 == 0 && aligned16(addend2)));
    g.bn_partial = nullptr;
    g.x_th = 0.0f;
    g.OH = H; g.OW = W;
    g.bn_fps = frames_per_step; g.bn_T = (int)(N / frames_per_step); g.bn_tc = g.bn_T * Cin;
    g.tiles_x = g.tiles_img = 0;
    dim3 grid((unsigned)((int64_t)g.tiles_per_xcd * 8));
    const unsigned char* wi = static_cast<const unsigned char*>(wt_image);
    if (co_tile == 128)
        hipLaunchKernelGGL((k_conv_halo3<128, false, 0, true>), grid, dim3(kThreads), 0, (hipStream_t)stream, gx, wi, dx, g,
                           addend, addend2, y, coef, dy_out);
    else
        hipLaunchKernelGGL((k_conv_halo3<64, false, 0, true>), grid, dim3(kThreads), 0, (hipStream_t)stream, gx, wi, dx, g,
                           addend, addend2, y, coef, dy_out);
    SNN_CHECK_LAUNCH("snn_conv3x3_halo_bn");
    return 0;
}

static int conv3x3_halo_impl(const float* x, int64_t ldx, const void* w_image, float* y, int64_t ldy, int64_t N, int H,
                             int W, int Cin, int Cout, const float* addend, int64_t ld_addend, const float* addend2,
                             int64_t ld_addend2, double* bn_partial, int frames_per_step, int* bn_layout,
                             int precision, void* stream, bool xsp, float x_th) {
    SNN_REQUIRE(x && w_image && y, "snn_conv3x3_halo: null pointer");
    SNN_REQUIRE(precision == SNN_PREC_FP16X3 || precision == SNN_PREC_BF16X3 || precision == SNN_PREC_BF16S,
                "snn_conv3x3_halo: precision must be SNN_PREC_FP16X3, SNN_PREC_BF16X3 or SNN_PREC_BF16S (got %d)", precision);
    const bool sbf = precision == SNN_PREC_BF16S;   // x, y, addends bf16; the image holds bf16 pieces (SNN_PREC_BF16X3 image)
    SNN_REQUIRE(halo_shape_ok(N, H, W, Cin, Cout),
                "snn_conv3x3_halo: shape not covered (N %lld, %dx%d, %d -> %d channels; ask snn_conv3x3_halo_supported)",
                (long long)N, H, W, Cin, Cout);
    SNN_REQUIRE(ldx >= Cin && ldy >= Cout && ldx % 4 == 0, "snn_conv3x3_halo: bad pixel strides (%lld, %lld)",
                (long long)ldx, (long long)ldy);
    SNN_REQUIRE((sbf ? aligned8(x) : aligned16(x)) && aligned16(w_image),
                "snn_conv3x3_halo: x (16 bytes; 8 for bf16) and the weight image (16) must be aligned");
    SNN_REQUIRE(!addend || ld_addend >= Cout, "snn_conv3x3_halo: addend pixel stride smaller than channel count");
    SNN_REQUIRE(!addend2 || ld_addend2 >= Cout, "snn_conv3x3_halo: addend2 pixel stride smaller than channel count");
    SNN_REQUIRE(!bn_partial || ((precision == SNN_PREC_FP16X3 || sbf) && bn_layout && frames_per_step > 0 &&
                                N % frames_per_step == 0 && !addend && !addend2),
                "snn_conv3x3_halo: statistics need the forward arithmetic, bn_layout, no addend and a frames_per_step "
                "that divides N (%lld frames, %d per step)", (long long)N, frames_per_step);
    SNN_REQUIRE((int64_t)4 * H * W * ldx * 4 < 0x7fffffffLL, "snn_conv3x3_halo: four images must span less than 2 GiB");
    if (bn_layout) bn_layout[0] = bn_layout[1] = 0;
    HaloGeom g;
    g.ldx = ldx; g.ldy = ldy; g.ld_add = ld_addend; g.ld_add2 = ld_addend2;
    g.N = (int)N; g.H = H; g.W = W; g.Cin = Cin; g.Cout = Cout;
    const bool rect = halo_mode(N, H, W, Cin, Cout) == 2;
    g.PW = W + 1; g.PH = H + 1;
    g.G = bn_partial ? frames_per_step : (int)N;
    g.tiles_x = (int)snn_ceil_div(W, RTW);
    g.tiles_img = g.tiles_x * (int)snn_ceil_div(H, RTH);
    int64_t tiles;
    if (rect) {
        g.group_cells = 0;
        g.tiles_per_group = g.G * g.tiles_img;
        tiles = N * (int64_t)g.tiles_img;
        SNN_REQUIRE((int64_t)H * W * ldx * 4 < 0x7fffffffLL, "snn_conv3x3_halo: an image must span less than 2 GiB");
    } else {
        const int64_t group_cells = (int64_t)g.G * g.PH * g.PW;
        g.group_cells = (int)group_cells;
        g.tiles_per_group = (int)snn_ceil_div(group_cells, HBM_);
        tiles = (int64_t)(N / g.G) * g.tiles_per_group;
    }
    const int co_tile = Cout % 128 == 0 ? 128 : (Cout % 64 == 0 ? 64 : 32);
    g.ntiles_n = Cout / co_tile;
    SNN_REQUIRE(tiles * g.ntiles_n + 8 < 0x7fffffffLL, "snn_conv3x3_halo: grid too large");
    g.tiles = (int)tiles;
    g.tiles_per_xcd = (int)snn_ceil_div(tiles, 8);
    g.magic_pw = magic_u32(g.PW); g.magic_ph = magic_u32(g.PH);
    g.out_vec = (ldy % 4 == 0) && out_aligned(y, sbf) && (!addend || (ld_addend % 4 == 0 && out_aligned(addend, sbf))) &&
                (!addend2 || (ld_addend2 % 4 == 0 && out_aligned(addend2, sbf)));
    g.bn_partial = bn_partial;
    g.OH = H; g.OW = W;
    g.bn_T = g.bn_tc = 0; g.bn_fps = 1;
    g.x_th = x_th;
    if (bn_partial) bn_layout[0] = g.tiles_per_group;   // every slot of every step is written: rows_per_chunk stays 0
    dim3 grid((unsigned)((int64_t)g.tiles_per_xcd * 8 * g.ntiles_n));
    const unsigned char* wi = static_cast<const unsigned char*>(w_image);
    const bool f16 = precision == SNN_PREC_FP16X3;
#define SNN_HALO_LAUNCH(CO_, F16_)                                                                                   \
    do {                                                                                                              \
        if (rect)                                                                                                     \
            hipLaunchKernelGGL((k_conv_halo3<CO_, F16_, 0, false, true>), grid, dim3(kThreads), 0, (hipStream_t)stream, x, wi, \
                               y, g, addend, addend2, nullptr, nullptr, nullptr);                                     \
        else                                                                                                          \
            hipLaunchKernelGGL((k_conv_halo3<CO_, F16_>), grid, dim3(kThreads), 0, (hipStream_t)stream, x, wi, y, g, addend, \
                               addend2, nullptr, nullptr, nullptr);                                                   \
    } while (0)
#ifdef SNN_TUNING
    if (const char* e = snn_tuning_env("SNN_HALO_ABL")) {   // timing experiments (tools/halo_abl.py): WRONG results
        const int abl = atoi(e);
#define SNN_HALO_ABL_LAUNCH(A_) \
        if (abl == A_) { \
            if (co_tile == 128) hipLaunchKernelGGL((k_conv_halo3<128, true, A_>), grid, dim3(kThreads), 0, (hipStream_t)stream, x, wi, y, g, addend, addend2); \
            else hipLaunchKernelGGL((k_conv_halo3<64, true, A_>), grid, dim3(kThreads), 0, (hipStream_t)stream, x, wi, y, g, addend, addend2); \
            SNN_CHECK_LAUNCH("snn_conv3x3_halo"); return 0; }
        SNN_HALO_ABL_LAUNCH(1) SNN_HALO_ABL_LAUNCH(2) SNN_HALO_ABL_LAUNCH(4) SNN_HALO_ABL_LAUNCH(8) SNN_HALO_ABL_LAUNCH(16)
        SNN_HALO_ABL_LAUNCH(3) SNN_HALO_ABL_LAUNCH(7) SNN_HALO_ABL_LAUNCH(15) SNN_HALO_ABL_LAUNCH(31)
#undef SNN_HALO_ABL_LAUNCH
    }
#endif
    if (xsp) {   // (precision checked by the caller: the forward arithmetic)
#define SNN_HALO_LAUNCH_X(CO_)                                                                                       \
    do {                                                                                                              \
        if (rect)                                                                                                     \
            hipLaunchKernelGGL((k_conv_halo3<CO_, true, 0, false, true, false, true>), grid, dim3(kThreads), 0,          \
                               (hipStream_t)stream, x, wi, y, g, addend, addend2, nullptr, nullptr, nullptr);          \
        else                                                                                                          \
            hipLaunchKernelGGL((k_conv_halo3<CO_, true, 0, false, false, false, true>), grid, dim3(kThreads), 0,         \
                               (hipStream_t)stream, x, wi, y, g, addend, addend2, nullptr, nullptr, nullptr);          \
    } while (0)
        if (co_tile == 128) SNN_HALO_LAUNCH_X(128);
        else if (co_tile == 64) SNN_HALO_LAUNCH_X(64);
        else SNN_HALO_LAUNCH_X(32);
#undef SNN_HALO_LAUNCH_X
        SNN_CHECK_LAUNCH("snn_conv3x3_halo_spikes");
        return 0;
    }
    if (sbf) {
#define SNN_HALO_LAUNCH_S(CO_)                                                                                       \
    do {                                                                                                              \
        if (rect)                                                                                                     \
            hipLaunchKernelGGL((k_conv_halo3<CO_, false, 0, false, true, true>), grid, dim3(kThreads), 0, (hipStream_t)stream, x, \
                               wi, y, g, addend, addend2, nullptr, nullptr, nullptr);                                 \
        else                                                                                                          \
            hipLaunchKernelGGL((k_conv_halo3<CO_, false, 0, false, false, true>), grid, dim3(kThreads), 0, (hipStream_t)stream, \
                               x, wi, y, g, addend, addend2, nullptr, nullptr, nullptr);                              \
    } while (0)
        if (co_tile == 128) SNN_HALO_LAUNCH_S(128);
        else if (co_tile == 64) SNN_HALO_LAUNCH_S(64);
        else SNN_HALO_LAUNCH_S(32);
#undef SNN_HALO_LAUNCH_S
        SNN_CHECK_LAUNCH("snn_conv3x3_halo");
        return 0;
    }
    if (co_tile == 128) {
        if (f16) SNN_HALO_LAUNCH(128, true); else SNN_HALO_LAUNCH(128, false);
    } else if (co_tile == 64) {
        if (f16) SNN_HALO_LAUNCH(64, true); else SNN_HALO_LAUNCH(64, false);
    } else {
        if (f16) SNN_HALO_LAUNCH(32, true); else SNN_HALO_LAUNCH(32, false);
    }
#undef SNN_HALO_LAUNCH
    SNN_CHECK_LAUNCH("snn_conv3x3_halo");
    return 0;
}

extern "C" int snn_conv3x3_halo(const float* x, int64_t ldx, const void* w_image, float* y, int64_t ldy, int64_t N, int H,
                                int W, int Cin, int Cout, const float* addend, int64_t ld_addend, const float* addend2,
                                int64_t ld_addend2, double* bn_partial, int frames_per_step, int* bn_layout,
                                int precision, void* stream) {
    return conv3x3_halo_impl(x, ldx, w_image, y, ldy, N, H, W, Cin, Cout, addend, ld_addend, addend2, ld_addend2, bn_partial,
                             frames_per_step, bn_layout, precision, stream, false, 0.0f);
}

// forward over spikes that were never stored (k_conv_halo3 XSP): `vdec` holds the potentials, w_image the fp16 x 3 image
extern "C" int snn_conv3x3_halo_spikes(const float* vdec, int64_t ld, float v_th, const void* w_image, float* y, int64_t ldy,
                                       int64_t N, int H, int W, int Cin, int Cout, double* bn_partial, int frames_per_step,
                                       int* bn_layout, void* stream) {
    SNN_REQUIRE(v_th >= 0.0f, "snn_conv3x3_halo_spikes: a negative threshold would turn padding into spikes");
    return conv3x3_halo_impl(vdec, ld, w_image, y, ldy, N, H, W, Cin, Cout, nullptr, 0, nullptr, 0, bn_partial,
                             frames_per_step, bn_layout, SNN_PREC_FP16X3, stream, true, v_th);
}
