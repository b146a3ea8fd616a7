/*
 * snn_hip.h - C ABI of the MI355X (gfx950) spiking-CNN forward/backward step.
 *
 * This is the drop-in boundary below the reference's operator API
 * (models.generator {Conv, Norm, LIF, LI, Pool, ...}).  The reference has no
 * native code of its own: every entry point below replaces the ATen / norse
 * call sequence that one reference module issues per timestep, re-cut for a
 * layer-major / time-inner schedule (one call covers all T timesteps).
 * Citations are reference file:line (relative to the upstream tree).
 *
 * Conventions
 *  - every pointer is a DEVICE pointer (hipMalloc'ed by the caller, e.g. a
 *    torch storage); the library never allocates, frees or synchronises;
 *  - activations are channels-last: frame-major [N][H][W][C] with N = T*B
 *    frames ("T-major": frame index = t*B + b); `ld*` is the element stride
 *    between two pixels of a buffer (>= C; lets an op read / write a channel
 *    slice of a wider concat buffer, reference Dense merge generator.py:157-161);
 *  - conv weights are [Cout][KH][KW][Cin] (OHWI = torch channels_last memory
 *    of the reference's [Cout,Cin,KH,KW] parameter);
 *  - `stream` is a hipStream_t passed as void*;
 *  - return 0 on success, non-zero on error; snn_last_error() gives the text
 *    (thread-local).  Python wrappers turn non-zero into RuntimeError.
 *  - fp32 is the parity dtype (SNN_F32); per-(t,c) statistics are accumulated
 *    in fp64 like ATen's CPU batch-norm.
 */
#ifndef SNN_HIP_H
#define SNN_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SNN_ABI_VERSION 11

/* neuron kinds for the fused affine+neuron temporal scan */
enum {
    SNN_NEURON_NONE = 0,   /* plain per-(t,c) affine (BatchNorm apply)            */
    SNN_NEURON_LIF = 1,    /* norse LIFCell step, layer_gen.py:232-235            */
    SNN_NEURON_LI = 2,     /* norse LICell step,  layer_gen.py:252-254            */
    SNN_NEURON_LI_TANH = 3,/* LICell followed by nn.Tanh, tiny_yolo.py:39-44      */
    SNN_NEURON_SLI = 4,    /* saturable leaky integrator, sli.py:110-126          */
    SNN_NEURON_SYNAPSE = 5 /* mediator-concentration synapse, synapse.py:73-103   */
};

/* Arithmetic of a convolution call (the `precision` argument of snn_conv2d_*).  Tensors in HBM and the accumulators
 * are fp32 in every mode; the modes differ in how the PRODUCTS are formed on the matrix cores:
 *   SNN_PREC_FP32   exact fp32 MFMA (v_mfma_f32_32x32x2_f32): a k-ordered fmaf chain.               fwd, dgrad, wgrad
 *   SNN_PREC_BF16X3 operands split into bf16 hi + lo, product = hi*hi + hi*lo + lo*hi on the bf16 MFMA: relative
 *                   error ~2^-16 per product (measured 1e-5).  Default of the backward convolutions.   dgrad, wgrad
 *   SNN_PREC_BF16X6 three-way bf16 split of both operands (h + m + l = all 24 significant bits) and the six leading
 *                   products: fp32-grade (rel 5e-7 vs fp64) for any fp32 range.                                 fwd
 *   SNN_PREC_FP16X3 both operands split into two fp16 pieces (11 + 11 significant bits) after exact power-of-two
 *                   pre-scaling (weights x 2^8, activations x 2^4), products hh + hl + lh: relative error 2^-22
 *                   (measured 5e-7 vs fp64, equal to the fp32 MFMA's) at half the matrix work of BF16X6.  Range
 *                   contract: |x| < 4094 and |w| < 255 (beyond: inf/NaN in the output - visible, not silent);
 *                   values below |x| = 0.008 / |w| = 5e-4 keep an ABSOLUTE accuracy of 4e-9 / 2e-10 instead of 22
 *                   bits.  Default of the forward convolution (inputs are spikes / normalised activations).     fwd
 *   SNN_PREC_BF16X1 the opt-in THROUGHPUT mode ("bf16" of BASELINE configs[1]): every operand is rounded once to bf16
 *                   and multiplied as it is - one product, fp32 accumulation, fp32 tensors in HBM.  Relative error
 *                   2^-9 per product: NOT a parity mode (spike trains diverge from the fp32 reference within a few
 *                   layers; loss and gradients to ~1e-2).  Never a default.                     fwd, dgrad, wgrad
 *   SNN_PREC_BF16S  bf16 STORAGE (the opt-in throughput mode whose tensors are bf16 in HBM too): the activation
 *                   operands AND results of the call (x / y, dy / dx; every `float*` activation pointer of the
 *                   signature then addresses bf16 elements, pixel strides stay in ELEMENTS) are bf16, products are
 *                   single bf16 MFMA products (weights, fp32, are rounded to bf16 on the way in), accumulation and
 *                   BatchNorm statistics fp32 / fp64.  Weights, weight gradients, statistics, neuron state stay fp32.
 *                   Same tolerances class as SNN_PREC_BF16X1; never a default.  Covered shapes: the vectorised
 *                   (Cin % 32 == 0) implicit GEMM, the halo-resident 3x3 kernels, the event-frame kernels (input
 *                   frames fp32).                                                              fwd, dgrad, wgrad
 * The mode is an argument of every call - the library keeps no process-wide arithmetic state. */
enum { SNN_PREC_FP32 = 0, SNN_PREC_BF16X3 = 1, SNN_PREC_BF16X6 = 3, SNN_PREC_FP16X3 = 4, SNN_PREC_BF16X1 = 5,
       SNN_PREC_BF16S = 6 };

/* flags of snn_affine_neuron_fwd / _bwd */
enum { SNN_SCAN_WIDE_ADDRESSING = 1, /* bwd: use 64-bit pointer addressing even when one timestep of every tensor fits
                                        the 31-bit buffer offsets (the library switches by itself when it does not) */
       SNN_SCAN_LAST_STEP_ONLY = 2,  /* LIF / LI / LI+Tanh whose consumer keeps the last timestep only (the detection head,
                                        soda.py:141-144): fwd writes out[M][ldo] of step T-1 instead of [T][M][ldo];
                                        bwd takes g_out[M][ldg] (and, LI+Tanh, the saved output [M][C]) of that step,
                                        the output gradient of every earlier step being zero */
       SNN_SCAN_BF16_STORAGE = 4,    /* bf16-storage mode (see SNN_PREC_BF16S): the activation tensors of the call - fwd: y,
                                        out, addend, vdec; bwd: g_out, state, y, gx - are bf16 (pointers passed as float*,
                                        strides in elements); state (v, i), alpha / beta and the sums stay fp32.
                                        NONE / LIF / LI / LI+Tanh, C and strides multiples of 4 */
       SNN_SCAN_SPIKES_FROM_VDEC = 8,/* fwd, Norm -> LIF without a shortcut whose potentials are saved (vdec != NULL): write
                                        NO output tensor (out must be NULL) - the saved pre-reset potentials already hold
                                        the spikes, z = (v_dec > v_th), and the consumer forms them while it reads them
                                        (snn_conv1x1_spikes_fwd / _wgrad): 4 of the 12 bytes the scan moves per
                                        neuron-timestep never exist.  fp32 tensors, C and ldy multiples of 4 */
       SNN_SCAN_SUMS_FROM_STATE = 16,/* bwd, LIF from the initial state with `sums` wanted: y is NOT read (NULL allowed).
                                        The BatchNorm statistic the scan needs y for, sum(gx * y), is replaced by
                                        sum(gx * x) with the neuron's input x[t] = alpha*y[t] + beta rebuilt from the saved
                                        potentials (vd[t], vd[t-1], vd[t-2] and the reset rule determine x[t] to fp32
                                        rounding): 4 of the 16 bytes per neuron-timestep are never read.  The sums must
                                        then go through snn_bn_bwd_finalize_from_state.  Covered cases:
                                        snn_affine_neuron_bwd_sums_from_state().  The sequence must start from the
                                        initial state (v_leak, 0) - or carry the next flag */
       SNN_SCAN_STATE_LOOKBACK = 32  /* with SNN_SCAN_SUMS_FROM_STATE, for a SEGMENT [t0, t0 + T) of a longer saved
                                        sequence, t0 >= 2: `state` points at step t0 and state[-1], state[-2] (the two
                                        steps in front of it, same [M][C] layout) are read to rebuild the neuron state
                                        the segment starts from */ };

/* pooling kinds, layer_gen.py:139-173 / common.py:18-49 */
enum { SNN_POOL_AVG = 0, SNN_POOL_MAX = 1, SNN_POOL_SUM = 2 };

/* pointwise activations, layer_gen.py:257-284 */
enum { SNN_ACT_RELU = 0, SNN_ACT_SILU = 1, SNN_ACT_TANH = 2 };

/* neuron constants exactly as norse rounds them in fp32 (oracle/neurons.py:neuron_constants) */
typedef struct snn_neuron_params {
    float c_mem;   /* dt * tau_mem_inv            (0.1)  */
    float c_syn;   /* -dt * tau_syn_inv           (-0.2) */
    float v_leak;  /* 0 */
    float v_th;    /* 1 */
    float v_reset; /* 0 */
    float alpha;   /* SuperSpike slope, 100 */
    float v_st;    /* SLI saturation potential, 1 (sli.py:38-39)                              */
    float tau_sec; /* synapse: mediator secretion rate 1/1e-3 (synapse.py:26-27)              */
    float tau_dis; /* synapse: mediator dissociation rate 1/5e-3 (synapse.py:29-30)           */
    float dt;      /* synapse integration step 1e-3 (synapse.py:77)                           */
    float sigma;   /* synapse inhibition, 0 = off (synapse.py:32-36)                          */
} snn_neuron_params;

int snn_abi_version(void);
const char* snn_last_error(void);

/* ---------------------------------------------------------------- layout
 * [N][C][H][W] <-> [N][H][W][C].  Entry/exit adapters for callers that hold
 * the reference's NCHW event frames X[T,B,2,H,W] (soda.py:138-144). */
int snn_nchw_to_nhwc(const float* src, float* dst, int64_t N, int C, int H, int W, void* stream);
int snn_nhwc_to_nchw(const float* src, float* dst, int64_t N, int C, int H, int W, void* stream);
/* weights [Cout][KH][KW][Cin] -> [Cin][KH][KW][Cout] (operand of the data-gradient conv) */
int snn_weight_transpose(const float* w, float* wt, int Cout, int KH, int KW, int Cin, void* stream);
/* the same for every conv weight of a flat parameter buffer in ONE launch: table[l] = {element offset, Cout, KH*KW, Cin}
 * (device memory, int64); flat_wt mirrors the offsets of flat_w.  A layer is indexed in 32 bits: Cout*KH*KW*Cin < 2^31
 * (the table is device memory: the caller that builds it checks). */
int snn_weight_transpose_batched(const float* flat_w, float* flat_wt, const int64_t* table, int n_layers, void* stream);
/* Pre-split weight image for the convolution kernels: every group of 4 consecutive floats of `w` (n floats, n % 4 == 0,
 * both buffers 16-byte aligned, `out` as large as `w`) becomes 4 high + 4 low 16-bit pieces - fp16 pieces of w * 2^8 for
 * SNN_PREC_FP16X3 (forward, apply to the OHWI weights), bf16 pieces for SNN_PREC_BF16X3 (data gradient, apply to the
 * transposed weights) - exactly the pieces the kernels would derive themselves.  Elementwise, so ONE call over a flat
 * parameter buffer serves every layer whose weight rows start on a multiple of 4 floats; redo after an optimiser step. */
int snn_weight_presplit(const float* w, void* out, int64_t n, int precision, void* stream);

/* ---------------------------------------------------------------- convolution
 * Replaces nn.Conv2d(bias=False, padding=int(k/2), stride=s) of layer_gen.py:129-136
 * (forward) and its autograd (ATen conv backward) for all T*B frames at once.
 * Implicit GEMM on the matrix cores, LDS-tiled; `precision` = SNN_PREC_* (above).
 *
 * fwd  : y[n,ho,wo,co]   = sum_{kh,kw,ci} x[n, ho*s-pad+kh, wo*s-pad+kw, ci] * w[co,kh,kw,ci]
 * dgrad: dx[n,hi,wi,ci]  = sum_{kh,kw,co} dy[n,(hi+pad-kh)/s,(wi+pad-kw)/s,co] * wt[ci,kh,kw,co]
 *        (terms with a non-integral or out-of-range source pixel are 0; wt from snn_weight_transpose)
 * wgrad: dw[co,kh,kw,ci] = sum_{n,ho,wo} dy[n,ho,wo,co] * x[n, ho*s-pad+kh, wo*s-pad+kw, ci]
 *        partial sums over pixel ranges go to `workspace` ([splitk][Cout*KH*KW*Cin] floats, splitk from
 *        snn_conv2d_wgrad_splitk), then reduced in fixed order (bitwise reproducible).  3x3 / pad 1 / stride 1|2
 *        layers with Cin, Cout multiples of 32 take the halo-resident kernel (csrc/wgrad_halo.hip), the rest the
 *        implicit-GEMM kernel.
 * fwd / dgrad `addend` (may be NULL): a tensor of the result's shape with pixel stride ld_addend that is added
 *   in the epilogue (result = conv + addend); passing the destination itself accumulates in place.  This
 *   fuses the gradient sum of a tensor consumed by several branches (generator.py:181-187) into the dgrad.
 * wgrad accumulate != 0 : the result is added to dw instead of overwriting it.
 * fwd `bn_partial` (may be NULL): the convolution is followed by a train-mode BatchNorm (layer_gen.py:211-214 after
 *   :129-136) whose per-timestep statistics are then taken from the values on their way to the store instead of a
 *   second pass over y.  The N frames are T = N / frames_per_step timesteps of frames_per_step frames; bn_partial
 *   holds snn_conv2d_fwd_bn_partial_size() doubles; on return (host side, no synchronisation) bn_layout[0] =
 *   chunks per timestep and bn_layout[1] = rows per chunk, the two layout arguments of snn_bn_stats_finalize /
 *   snn_bn_stats_reduce.  bn_layout[0] == 0: this shape's kernel did not produce them - run snn_bn_stats.
 * fwd `w_split` / dgrad `wt_split` (may be NULL): the pre-split image of the same weights (snn_weight_presplit below),
 *   valid for SNN_PREC_FP16X3 (fwd) / SNN_PREC_BF16X3 (dgrad) only.  With it the kernels stop converting the weight
 *   tile in every block; the results are bit-identical to the conversion on the fly.  Shapes whose kernel cannot use
 *   it read `w` / `wt` as before, so both pointers are always passed. */
/* Weight gradient with the BatchNorm-backward affine applied while dy is READ (no materialised dy): the gradient that
 * reaches the convolution through a train-mode BatchNorm (layer_gen.py:211-214) is dy = A[t][c]*gx + B[t][c]*y + C[t][c]
 * (snn_bn_bwd_apply); when nothing else consumes dy - the event-frame layer, whose input needs no gradient - the
 * weight-gradient kernel forms it on the fly from gx (the scan's output, pixel stride ldgx), the saved convolution
 * output y (ldy) and coef = [3][T][Cout] (A, B, C planes), t = frame / frames_per_step.  Same statement, same
 * roundings as snn_bn_bwd_apply + snn_conv2d_wgrad.  snn_conv2d_wgrad_bn_supported: 1 for the shapes covered (the
 * event-frame row kernel: Cin = 2, 3x3); workspace / splitk as for snn_conv2d_wgrad. */
int snn_conv2d_wgrad_bn_supported(int64_t N, int H, int W, int Cin, int Ho, int Wo, int Cout, int KH, int KW, int stride,
                                  int pad);
int snn_conv2d_wgrad_bn(const float* x, int64_t ldx, const float* gx, int64_t ldgx, const float* y, int64_t ldy,
                        const float* coef, int T, int frames_per_step, float* dw, int64_t N, int H, int W, int Cin, int Ho,
                        int Wo, int Cout, int KH, int KW, int stride, int pad, int accumulate, float* workspace, int splitk,
                        void* stream);
/* ---- Halo-resident 3x3 / stride 1 / pad 1 convolution (csrc/conv_halo.hip): forward AND data gradient of the 32-, 64- and
 * 128-channel layers (reference models/modules/layer_gen.py:129-136, nn.Conv2d(C, C', 3, padding=1, bias=False)).
 * The activation halo of a tile is fetched and split into its 16-bit pieces once per 32-channel chunk (the implicit
 * GEMM does both once per tap); the weights arrive by LDS-DMA from an image in MFMA-fragment order.
 *   snn_conv3x3_halo_supported : 1 when the kernel covers the shape (Cin % 32 == 0; Cout = 32 or a multiple of 64; rows of
 *                                up to 78 pixels as padded strips, longer rows as 4 x 32 rectangles); otherwise use
 *                                snn_conv2d_fwd / snn_conv2d_dgrad.
 *   snn_weight_frag_image_batched : builds the weight images of n layers in ONE launch.  table (device, int64) rows
 *       {float offset of the layer's [O][3][3][I] matrix in flat_src, BYTE offset of its image in flat_dst, O, I}; an
 *       image takes snn_weight_frag_image_bytes(O, I) = 9*O*I*4 bytes (as many as the weights); max_threads = the
 *       largest 9 * (I/32) * (O/32) * 128 of the table.  Forward: src = the OHWI weights, flip = 0, SNN_PREC_FP16X3.
 *       Data gradient: src = the transposed weights [Cin][KH][KW][Cout] (snn_weight_transpose), flip = 1 (mirrored
 *       taps), SNN_PREC_BF16X3.  Redo after every optimiser step.
 *   snn_conv3x3_halo : y[N][H][W][ldy] = conv3x3(x[N][H][W][ldx], image) (+ addend + addend2), Cin input and Cout output
 *       channels (for a data gradient: x = dy, "Cin" = the layer's Cout and vice versa).  precision = the image's.
 *       bn_partial / frames_per_step / bn_layout as for snn_conv2d_fwd (forward arithmetic only, no addend); the partials
 *       hold snn_conv2d_fwd_bn_partial_size() doubles, bn_layout[0] = snn_conv3x3_halo_bn_chunks(), bn_layout[1] = 0. */
/* Data gradient of a 3x3 / stride 2 / pad 1 convolution in ONE pass over dy (csrc/conv_halo.hip, k_conv_s2dgrad3): the
 * four stride-phase classes of dx are produced from one staged dy halo instead of four launches that each gather dy
 * again.  wt_image = the layer's data-gradient image (snn_weight_frag_image_batched of the transposed weights, flip = 1,
 * SNN_PREC_BF16X3 - the image the stride-1 data gradient uses).  dx[N][H][W][lddx] (Cin channels) from dy[N][Ho][Wo][lddy]
 * (Cout channels), Ho = (H-1)/2 + 1; addend / addend2 as for snn_conv2d_dgrad.  bf16 x 3 arithmetic, or bf16 storage.
 * Covers Cout % 32 == 0, Cin % 64 == 0, any width (dy rows of up to 157 cells as padded strips, longer ones as rectangles). */
int snn_conv3x3_s2_dgrad_supported(int64_t N, int H, int W, int Cin, int Ho, int Wo, int Cout);
int snn_conv3x3_s2_dgrad(const float* dy, int64_t lddy, const void* wt_image, float* dx, int64_t lddx, int64_t N, int H, int W,
                         int Cin, int Ho, int Wo, int Cout, const float* addend, int64_t ld_addend, const float* addend2,
                         int64_t ld_addend2, int precision /* SNN_PREC_BF16X3 | SNN_PREC_BF16S */, void* stream);
/* Data gradient of a 3x3 / stride 1 / pad 1 convolution behind a train-mode BatchNorm with the BatchNorm-backward affine
 * applied while gx is staged (k_conv_halo3<BNAP>): dy = A[t][c]*gx + B[t][c]*y + C[t][c] (see snn_conv2d_wgrad_bn; coef =
 * [3][T][Cin], T = N / frames_per_step) is formed once per staged cell, written to dy_out (dense [N][H][W][Cin], for the
 * weight gradient) by the tile that owns the cell, and convolved with the data-gradient image into dx[N][H][W][lddx]
 * (Cout channels) (+ addends).  gx and y are dense [N][H][W][Cin].  Here Cin = channels of gx / y / dy (the layer's
 * OUTPUT channels), Cout = channels of dx (the layer's input channels).  bf16 x 3 arithmetic, same statement and
 * roundings for dy as snn_bn_bwd_apply. */
int snn_conv3x3_halo_bn_supported(int64_t N, int H, int W, int Cin, int Cout, int frames_per_step);
int snn_conv3x3_halo_bn(const float* gx, const float* y, const float* coef, int frames_per_step, float* dy_out,
                        const void* wt_image, float* dx, int64_t lddx, int64_t N, int H, int W, int Cin, int Cout,
                        const float* addend, int64_t ld_addend, const float* addend2, int64_t ld_addend2, void* stream);
int snn_conv3x3_halo_supported(int64_t N, int H, int W, int Cin, int Cout);
int64_t snn_conv3x3_halo_bn_chunks(int frames_per_step, int H, int W);
size_t snn_weight_frag_image_bytes(int O, int I);
int snn_weight_frag_image_batched(const float* flat_src, void* flat_dst, const int64_t* table, int n, int max_threads,
                                  int flip, int precision, void* stream);
int snn_conv3x3_halo(const float* x, int64_t ldx, const void* w_image, float* y, int64_t ldy, int64_t N, int H, int W,
                     int Cin, int Cout, const float* addend, int64_t ld_addend, const float* addend2, int64_t ld_addend2,
                     double* bn_partial, int frames_per_step, int* bn_layout, int precision, void* stream);
size_t snn_conv2d_fwd_bn_partial_size(int64_t N, int frames_per_step, int Ho, int Wo, int Cout);
int snn_conv2d_fwd(const float* x, int64_t ldx, const float* w, const void* w_split, float* y, int64_t ldy,
                   int64_t N, int H, int W, int Cin, int Ho, int Wo, int Cout,
                   int KH, int KW, int stride, int pad, const float* addend, int64_t ld_addend,
                   double* bn_partial, int frames_per_step, int* bn_layout, int precision, void* stream);
/* dgrad takes TWO optional addends (dx = conv^T(dy) + addend + addend2): a tensor consumed by a convolution, a
 * residual shortcut and a Dense pass-through (the YOLO bottleneck inside a C2f block) gets its whole gradient in one
 * epilogue instead of two extra add passes. */
int snn_conv2d_dgrad(const float* dy, int64_t lddy, const float* wt, const void* wt_split, float* dx, int64_t lddx,
                     int64_t N, int H, int W, int Cin, int Ho, int Wo, int Cout,
                     int KH, int KW, int stride, int pad, const float* addend, int64_t ld_addend,
                     const float* addend2, int64_t ld_addend2, int precision, void* stream);
int snn_conv2d_wgrad(const float* x, int64_t ldx, const float* dy, int64_t lddy, float* dw,
                     int64_t N, int H, int W, int Cin, int Ho, int Wo, int Cout,
                     int KH, int KW, int stride, int pad, int accumulate,
                     float* workspace, int splitk, int precision, void* stream);
/* number of workspace slabs snn_conv2d_wgrad wants for this shape and precision (workspace = splitk*Cout*KH*KW*Cin
 * floats); same geometry arguments as snn_conv2d_wgrad; host-only, callable without a device */
int snn_conv2d_wgrad_splitk(int64_t N, int H, int W, int Cin, int Ho, int Wo, int Cout, int KH, int KW,
                            int stride, int pad, int precision);
/* which kernel snn_conv2d_wgrad takes for this shape: 0 implicit GEMM (k_conv_wgrad_pipe), 1 halo-resident
 * (k_conv_wgrad_halo), 2 event-frame row kernel (k_conv_first); host-only - labels of the measurement tools */
int snn_conv2d_wgrad_kernel(int64_t N, int H, int W, int Cin, int Ho, int Wo, int Cout, int KH, int KW, int stride, int pad,
                            int precision);
/* ---- 1x1 convolutions over spikes that were never stored.  The stage-entry layers of the generated nets are
 * Conv -> Norm -> LIF feeding only 1x1 convolutions (reference models/tiny_yolo.py:76-85 behind :16-21); their LIF
 * (models/modules/layer_gen.py:232-235) saves v_dec for its backward pass anyway, so the spike tensor z = (v_dec > v_th)
 * is not written (SNN_SCAN_SPIKES_FROM_VDEC) and these entry points take the POTENTIALS as their input tensor:
 *   snn_conv1x1_spikes_fwd   : y[N][H][W][ldy] (Cout channels) = conv1x1(z, w[Cout][Cin]), fp16 x 3 arithmetic - a spike
 *                              is exact in one fp16 piece, so two of the three products are issued; same bits as
 *                              snn_conv2d_fwd(z, ...)
 *   snn_conv1x1_spikes_wgrad : dw[Cout][Cin] (+)= sum_pixels dy z^T, bf16 x 3 arithmetic (two products); workspace / splitk
 *                              as for snn_conv2d_wgrad with KH = KW = 1 (snn_conv2d_wgrad_splitk); same bits as
 *                              snn_conv2d_wgrad(z, ...)
 *   snn_conv1x1_spikes_supported : 1 for the shapes covered (Cin % 32 == 0, Cout % 4 == 0, ld % 4 == 0, the two default
 *                              arithmetics); otherwise the layer must write its spikes.
 * v_th >= 0 (padding rows read as potential 0 and must not spike).  The data gradient needs no input tensor. */
int snn_conv1x1_spikes_supported(int64_t N, int H, int W, int Cin, int Cout, int64_t ld, int fwd_precision,
                                 int bwd_precision);
int snn_conv1x1_spikes_fwd(const float* vdec, int64_t ld, float v_th, const float* w, float* y, int64_t ldy, int64_t N, int H,
                           int W, int Cin, int Cout, void* stream);
int snn_conv1x1_spikes_wgrad(const float* vdec, int64_t ld, float v_th, const float* dy, int64_t lddy, float* dw, int64_t N,
                             int H, int W, int Cin, int Cout, int accumulate, float* workspace, int splitk, void* stream);

/* The same for ANY convolution the pipelined implicit GEMM covers (KH, KW <= 5, Cin % 32 == 0, Cout % 4 == 0) - a plain
 * `Conv -> Norm -> LIF -> Conv` stack (models/modules/layer_gen.py:106-136 / 211-235: the deep backbones of BASELINE
 * configs[4]) writes no spike tensor between its layers:
 *   snn_conv2d_spikes_fwd   : as snn_conv2d_fwd on z = (vdec > v_th), fp16 x 3 arithmetic with two products; bn_partial /
 *                             frames_per_step / bn_layout as there
 *   snn_conv2d_spikes_wgrad : as snn_conv2d_wgrad (workspace / splitk from snn_conv2d_wgrad_splitk, SNN_PREC_BF16X3); 3x3
 *                             layers the halo-resident weight gradient covers take it (two products)
 *   snn_conv3x3_halo_spikes : the halo-resident 3x3 / stride 1 / pad 1 forward (snn_conv3x3_halo, fp16 x 3 image) on the
 *                             potentials; shapes: snn_conv3x3_halo_supported */
int snn_conv2d_spikes_supported(int64_t N, int H, int W, int Cin, int Ho, int Wo, int Cout, int KH, int KW, int stride,
                                int pad, int64_t ld, int fwd_precision, int bwd_precision);
int snn_conv2d_spikes_fwd(const float* vdec, int64_t ld, float v_th, const float* w, float* y, int64_t ldy, int64_t N, int H,
                          int W, int Cin, int Ho, int Wo, int Cout, int KH, int KW, int stride, int pad, double* bn_partial,
                          int frames_per_step, int* bn_layout, void* stream);
int snn_conv2d_spikes_wgrad(const float* vdec, int64_t ld, float v_th, const float* dy, int64_t lddy, float* dw, int64_t N,
                            int H, int W, int Cin, int Ho, int Wo, int Cout, int KH, int KW, int stride, int pad,
                            int accumulate, float* workspace, int splitk, void* stream);
int snn_conv3x3_halo_spikes(const float* vdec, int64_t ld, float v_th, const void* w_image, float* y, int64_t ldy, int64_t N,
                            int H, int W, int Cin, int Cout, double* bn_partial, int frames_per_step, int* bn_layout,
                            void* stream);

/* ---------------------------------------------------------------- batch-norm statistics
 * Train-mode nn.BatchNorm2d (layer_gen.py:211-214) applied per TIMESTEP: for every (t,c)
 * the biased mean/var over the M = B*h*w pixels of timestep t (generator.py:190-195 calls the
 * module once per timestep).  `partial` is scratch of snn_bn_stats_partial_size() doubles.
 * snn_bn_stats_finalize also performs the T sequential running-stat updates of one reference
 * forward (momentum 0.1, unbiased variance) when running_mean/var are non-NULL, and emits the
 * per-(t,c) affine  alpha = gamma*invstd, beta = bias - mean*alpha  that the scan consumes.
 * use_running != 0 (eval mode): alpha/beta come from the running statistics for every t.
 * chunks / rows_per_chunk describe the layout of `partial`: 0, 0 for the one snn_bn_stats writes, the two values
 * snn_conv2d_fwd returned in bn_layout for partials that came out of a convolution epilogue. */
size_t snn_bn_stats_partial_size(int T, int64_t M, int C);
int snn_bn_stats(const float* y, int64_t ldy, int T, int64_t M, int C, double* partial, void* stream);
int snn_bn_stats_finalize(const double* partial, int chunks, int rows_per_chunk, int T, int64_t M, int C,
                          const float* gamma, const float* bias, float eps, float momentum,
                          float* running_mean, float* running_var, int use_running,
                          float* mean, float* invstd, float* alpha, float* beta, void* stream);

/* SyncBatchNorm form (config.yaml:76) of the two steps above: reduce the chunk partials to sums[T][C][2]
 * (sum y, sum y^2), let the caller all-reduce that small tensor over the ranks, then finish from the global
 * sums over M_total = world * M pixels.  var_scratch: T*C doubles. */
int snn_bn_stats_reduce(const double* partial, int chunks, int rows_per_chunk, int T, int64_t M, int C, double* sums,
                        void* stream);
int snn_bn_stats_from_sums(const double* sums, int T, int64_t M_total, int C,
                           const float* gamma, const float* bias, float eps, float momentum,
                           float* running_mean, float* running_var,
                           float* mean, float* invstd, float* alpha, float* beta,
                           double* var_scratch, void* stream);

/* ---------------------------------------------------------------- fused affine + neuron scan
 * One kernel = BatchNorm apply + T-step neuron recurrence with the membrane state held in
 * registers.  Replaces, per layer, T x {BatchNorm2d apply, ~12 elementwise norse ops}
 * (layer_gen.py:211-235, norse lif_feed_forward_step / li_feed_forward_step).
 *
 *   x[t]   = y[t]*alpha[t,c] + beta[t,c]                 (alpha/beta NULL -> identity)
 *   LIF    : i' = i + x; vd = v + c_mem*((v_leak - v) + i'); i = i' + c_syn*i';
 *            z = (vd - v_th > 0); v = (1-z)*vd + z*v_reset; out[t] = z
 *   LI     : i' = i + x; v = v + c_mem*((v_leak - v) + i'); i = i' + c_syn*i'; out[t] = v (or tanh(v))
 *   SLI    : as LI with the input gated: i' = i + x*sigmoid(v_st - |v|)
 *   SYNAPSE: p = p + ((x - p)*(x > 0 ? tau_sec : tau_dis))*dt; g = sigma ? 4*sigma*(p - sigma*p^2) : p;
 *            out[t] = max(g, 0); the state is p alone (held in the v slot, i unused)
 *   v0/i0 NULL -> initial state (v = v_leak, i = 0); vT/iT NULL -> final state not written.
 *   vdec (training) receives what the backward scan needs per step: LIF the pre-reset potential vd[t],
 *   SLI the potential BEFORE the step, SYNAPSE the new concentration p[t].
 *   addend (may be NULL; [T][M][C-slice], pixel stride ld_addend): out[t] = neuron output + addend[t] - the residual
 *   shortcut (generator.py:145-146, stack + sum) folded into the store; not allowed with LI_TANH, whose backward
 *   scan reads its own output.
 * Layout: y/out/vdec are [T][M][C-slice] with pixel strides ldy/ldo (vdec dense, ld = C). */
int snn_affine_neuron_fwd(int neuron, const float* y, int64_t ldy,
                          const float* alpha, const float* beta,
                          const float* v0, const float* i0,
                          float* out, int64_t ldo, const float* addend, int64_t ld_addend,
                          float* vT, float* iT, float* vdec,
                          int T, int64_t M, int C, const snn_neuron_params* p, int flags, void* stream);

/* Reverse-time scan (BPTT through the neuron, SuperSpike surrogate dz/du = 1/(alpha|u|+1)^2,
 * reset path NOT detached).  g_out is dL/d out[t]; g_vT/g_iT (may be NULL = 0) are the
 * gradients of the final state; writes gx = dL/dx[t] (dense [T][M][C]), g_v0/g_i0 (may be NULL)
 * and, when `sums` != NULL, per-block partial sums of gx and gx*y per (t,c) for the BatchNorm
 * backward (`sums` scratch of snn_affine_neuron_bwd_sums_size() doubles, y must be given).
 * `state` is the forward's vdec buffer for LIF / SLI / SYNAPSE, out (tanh output) for LI_TANH, unused otherwise.
 * alpha/beta (may be NULL = identity): the forward's affine, needed to rebuild x[t] for SLI / SYNAPSE.
 * apply_scale != 0: gx is multiplied by alpha[t,c] before it is written (eval-mode BN: dy = alpha*gx).
 * flags: 0, SNN_SCAN_WIDE_ADDRESSING, SNN_SCAN_LAST_STEP_ONLY (or both). */
size_t snn_affine_neuron_bwd_sums_size(int T, int64_t M, int C);
/* 1 when snn_affine_neuron_bwd(flags | SNN_SCAN_SUMS_FROM_STATE) covers the call (LIF, the ordered-sums plan of the shape,
 * 32-bit buffer addressing, fp32 tensors, all T gradients, c_mem in [1/64, 1]) */
int snn_affine_neuron_bwd_sums_from_state(int neuron, int T, int64_t M, int C, int64_t ldg, const snn_neuron_params* p,
                                          int flags);
int snn_affine_neuron_bwd(int neuron, const float* g_out, int64_t ldg, const float* state,
                          const float* y, int64_t ldy, const float* g_vT, const float* g_iT,
                          const float* alpha, const float* beta, int apply_scale,
                          float* gx, float* g_v0, float* g_i0, double* sums,
                          int T, int64_t M, int C, const snn_neuron_params* p, int flags, void* stream);

/* Memory-saving LIF pair (same results, bit for bit, as the two calls above with neuron = SNN_NEURON_LIF).  Instead
 * of vd[t] for every step, the forward stores the state (v, i) BEFORE every K-th step, K = snn_lif_ckpt_interval():
 *   ckpt[ceil(T/K)][2][M][C]   (2*ceil(T/K)/T of the per-step buffer: half at K = 4)
 * and the backward scan re-runs the K forward steps of a chunk from its checkpoint (it needs y and alpha/beta for
 * that) before walking the chunk in reverse.  An opt-in memory lever for 1280x720-class inputs
 * (functional.LIF_CHECKPOINT_BYTES); speed-neutral. */
int snn_lif_ckpt_interval(void);
int snn_lif_fwd_ckpt(const float* y, int64_t ldy, const float* alpha, const float* beta,
                     const float* v0, const float* i0, float* out, int64_t ldo,
                     const float* addend, int64_t ld_addend, float* vT, float* iT, float* ckpt,
                     int T, int64_t M, int C, const snn_neuron_params* p, void* stream);
int snn_lif_bwd_ckpt(const float* g_out, int64_t ldg, const float* ckpt, const float* y, int64_t ldy,
                     const float* g_vT, const float* g_iT, const float* alpha, const float* beta, int apply_scale,
                     float* gx, float* g_v0, float* g_i0, double* sums,
                     int T, int64_t M, int C, const snn_neuron_params* p, void* stream);

/* BatchNorm backward, second phase.  From the partial sums: per-(t,c)
 *   dy = A*gx + Bc*y + Cc,  A = alpha, Bc = -alpha*invstd*s2, Cc = -alpha*s1 + alpha*invstd*mean*s2,
 *   s1 = mean_pix(gx), s2 = mean_pix(gx*xhat);  dgamma[c] = sum_t sum_pix gx*xhat, dbias[c] = sum_t sum_pix gx.
 * dgamma/dbias are accumulated (+=) when accumulate != 0.  `sums` is reduced in place. */
int snn_bn_bwd_finalize(double* sums, int T, int64_t M, int C,
                        const float* gamma, const float* mean, const float* invstd,
                        float* coefA, float* coefB, float* coefC,
                        float* dgamma, float* dbias, int accumulate, void* stream);
/* The same for sums of a scan that ran with SNN_SCAN_SUMS_FROM_STATE: sum(gx*y) = mean*sum(gx) + (sum(gx*x) - bias*sum(gx))
 * / (gamma*invstd) (bias NULL = 0, gamma NULL = 1), then as above.  A channel whose gamma is exactly 0 has no xhat left in x: its sum(gx*y) is formed
 * from gx[T][M][C] and y[T][M][ldy] inside this call (one block per such channel: slow, exact, rare). */
int snn_bn_bwd_finalize_from_state(double* sums, int T, int64_t M, int C,
                                   const float* gamma, const float* bias, const float* mean, const float* invstd,
                                   const float* gx, const float* y, int64_t ldy,
                                   float* coefA, float* coefB, float* coefC,
                                   float* dgamma, float* dbias, int accumulate, void* stream);
/* The same in two steps for SyncBatchNorm: raw[T][C][2] = (sum gx, sum gx*y) of this rank; the caller
 * all-reduces a copy; coefficients come from the global sums over M_total pixels while dgamma / dbias use
 * the rank-local sums (DDP averages them afterwards).  param_sums: T*C*2 doubles of scratch. */
int snn_bn_bwd_reduce(const double* sums, int T, int64_t M, int C, double* raw, void* stream);
/* ... for sums of a scan that ran with SNN_SCAN_SUMS_FROM_STATE: raw receives (sum gx, sum gx*y) all the same (converted
 * with this rank's view of gamma / bias / mean / invstd, which SyncBatchNorm keeps equal on every rank) */
int snn_bn_bwd_reduce_from_state(const double* sums, int T, int64_t M, int C, const float* gamma, const float* bias,
                                 const float* mean, const float* invstd, const float* gx, const float* y, int64_t ldy,
                                 double* raw, void* stream);
int snn_bn_bwd_coef(const double* raw, const double* raw_local, double* param_sums, int T, int64_t M_total, int C,
                    const float* gamma, const float* mean, const float* invstd,
                    float* coefA, float* coefB, float* coefC,
                    float* dgamma, float* dbias, int accumulate, void* stream);
int snn_bn_bwd_apply(const float* gx, const float* y, int64_t ldy,
                     const float* coefA, const float* coefB, const float* coefC,
                     float* dy, int64_t lddy, int T, int64_t M, int C, int accumulate, void* stream);
/* the same two passes on bf16 tensors (bf16-storage mode, see SNN_PREC_BF16S: gx / y / dy are bf16, coefficients and
 * partials as above; C and the strides multiples of 4) */
int snn_bn_bwd_apply_bf16(const float* gx, const float* y, int64_t ldy, const float* coefA, const float* coefB,
                          const float* coefC, float* dy, int64_t lddy, int T, int64_t M, int C, int accumulate, void* stream);
int snn_bn_stats_bf16(const float* y, int64_t ldy, int T, int64_t M, int C, double* partial, void* stream);

/* ---------------------------------------------------------------- merges and pointwise
 * Residual merge = torch.stack(out).sum(0) (generator.py:145-146); Dense merge = torch.cat(out,1)
 * (generator.py:157-158).  A channel-slice copy / add over M pixels covers both and their backward. */
int snn_copy_channels(const float* src, int64_t lds, float* dst, int64_t ldd, int64_t M, int C, void* stream);
int snn_add_channels(const float* src, int64_t lds, float* dst, int64_t ldd, int64_t M, int C, void* stream);
/* dst = a + b over M pixels x C channels, each operand with its own pixel stride */
int snn_add(const float* a, int64_t lda, const float* b, int64_t ldb, float* dst, int64_t ldd,
            int64_t M, int C, void* stream);
/* the same merges on bf16 tensors (bf16-storage mode, see SNN_PREC_BF16S; pointers typed float*, strides in elements; the sum
 * is formed in fp32 and rounded once), and the conversion at the boundary of the bf16 domain (dense, n elements;
 * to_bf16 != 0: fp32 -> bf16 round to nearest even, else bf16 -> fp32) */
int snn_copy_channels_bf16(const float* src, int64_t lds, float* dst, int64_t ldd, int64_t M, int C, void* stream);
int snn_add_bf16(const float* a, int64_t lda, const float* b, int64_t ldb, float* dst, int64_t ldd, int64_t M, int C,
                 void* stream);
int snn_convert_bf16(const void* src, void* dst, int64_t n, int to_bf16, void* stream);
int snn_act_fwd(int act, const float* x, float* y, int64_t n, void* stream);
int snn_act_bwd(int act, const float* x, const float* y, const float* gy, float* gx, int64_t n, void* stream);

/* C (+)= op(A) x op(B), row-major fp32, op = transpose when the flag is set: A is [M][K] ([K][M] transposed), B is [K][N]
 * ([N][K] transposed), C is [M][N]; every element is an fmaf chain in k order.  For weight-sized matrices: two 1x1
 * convolutions with nothing between them (the C2f entry, models/tiny_yolo.py:76-82) are composed into one, and
 * their weight gradients are products of the composed gradient with the other factor.
 * Ct (may be NULL; accumulate must be 0): the transposed result [N][M] from the same launch - the composed weight and
 * the operand of its data gradient. */
int snn_small_gemm(const float* A, int64_t lda, int transA, const float* B, int64_t ldb, int transB, float* C,
                  int64_t ldc, int M, int N, int K, int accumulate, float* Ct, int64_t ldct, void* stream);
/* n products C_i = A_i B_i of dense row-major matrices (A_i [M,K], B_i [K,N], C_i [M,N]; Ct_i [N,M] = the transposed copy,
 * base_ct may be NULL) in ONE launch: the composed 1x1 weights w2 w1 of every C2f entry (reference models/tiny_yolo.py:76-82)
 * once per optimiser step.  `table` (device): n rows {A offset, B offset, C offset, Ct offset, M, N, K, ldct} as int64 (ABI
 * v9: 8 columns), offsets in floats relative to the four base pointers; ldct = row length of the transposed copy (0 = M):
 * several products may write column blocks of ONE [N][sum M] matrix - the row-stacked weight of sibling 1x1 convolutions
 * (models/tiny_yolo.py:84-85) and its transpose; max_tiles >= ceil(M/32) * ceil(N/32) of every row.  Same fmaf chains
 * (k order) as snn_small_gemm. */
int snn_small_gemm_batched(const float* base_a, const float* base_b, float* base_c, float* base_ct, const int64_t* table,
                           int n, int max_tiles, void* stream);

/* ConvLSTM cell (conv_lstm.py:51-78), pointwise part after the 1x1 gate convolution.  gates is dense
 * [M][4C] = (input, forget, output, candidate); c_prev NULL = zero state.
 *   c = sigmoid(f)*c_prev + sigmoid(i)*tanh(g);  h = sigmoid(o)*tanh(c) */
int snn_lstm_cell_fwd(const float* gates, const float* c_prev, float* h, float* c, int64_t M, int C, void* stream);
int snn_lstm_cell_bwd(const float* gates, const float* c_prev, const float* c, const float* gh, const float* gc,
                      float* g_gates, float* g_c_prev, int64_t M, int C, void* stream);

/* Pool("A"/"M"/"S", k, stride) (layer_gen.py:146-173, common.py:18-49), no padding, floor mode. */
int snn_pool_fwd(int kind, const float* x, float* y, int64_t N, int H, int W, int C,
                 int Ho, int Wo, int k, int stride, void* stream);
int snn_pool_bwd(int kind, const float* x, const float* gy, float* gx, int64_t N, int H, int W, int C,
                 int Ho, int Wo, int k, int stride, void* stream);
/* nn.Upsample(scale_factor=s, mode="nearest") (layer_gen.py:176-194) */
int snn_upsample_fwd(const float* x, float* y, int64_t N, int H, int W, int C, int scale, void* stream);
int snn_upsample_bwd(const float* gy, float* gx, int64_t N, int H, int W, int C, int scale, void* stream);

/* ---------------------------------------------------------------- optimizer
 * torch.optim.Adamax(lr, betas=(0.9,0.999), eps=1e-8, weight_decay=0) single-tensor step
 * (soda.py:135-136) over a flat parameter / gradient buffer; grad is multiplied by grad_scale first
 * (1/world_size after the data-parallel SUM all-reduce, DDP's gradient averaging). */
int snn_adamax_step(float* param, const float* grad, float* exp_avg, float* exp_inf,
                    int64_t n, float lr, float beta1, float beta2, float eps, int step,
                    float grad_scale, void* stream);

/* ---------------------------------------------------------------- event voxelisation
 * utils/datasets.py:378-435: scatter events (t_bin, p, y, x) into binary frames [T][H][W][2]
 * (channels-last image of the reference's [T,2,H,W]); x is clipped to W-1; value 1 (not a count). */
int snn_events_to_frames(const int32_t* t_bin, const int32_t* x, const int32_t* y, const int32_t* p,
                         int64_t n_events, float* frames, int T, int H, int W, void* stream);

/* ---------------------------------------------------------------- detection decode
 * Tail of SODa.predict (models/soda.py:202-233): per anchor conf = max_k prob[k], class = argmax - 1 (background -1),
 * box = offset_inverse(anchor, offsets) (utils/box.py:72-79, 102-119).  prob [A][K], offsets / anchors / boxes [A][4]. */
int snn_detect_decode(const float* cls_prob, const float* offsets, const float* anchors, int A, int K,
                      float* conf, int* cls, float* boxes, void* stream);
/* Per-class greedy NMS (utils/box.py:82-99).  order[A]: anchor ids sorted by (class ascending, confidence descending);
 * seg[num_classes + 1]: members of class c are order[seg[c] .. seg[c+1]).  kept[seg[c] ..) receives the kept ids of
 * class c in keep order, nkept[c] their number; kept_flag[id] = 1 and kept_rank[id] = rank inside its class for kept
 * anchors (both arrays must be zeroed by the caller).  One block per class; a candidate survives
 * only while IoU <= iou_threshold against every kept box (so a NaN IoU suppresses, as utils/box.py:95-97 does). */
int snn_nms_sorted(const float* boxes, const int* order, const int* seg, int num_classes, float iou_threshold,
                   int* kept, int* nkept, unsigned char* kept_flag, int* kept_rank, void* stream);

/* ---------------------------------------------------------------- training targets and loss
 * snn_roi_assign: RoI.__call__ (utils/roi.py:18-109) for a whole batch, one block per sample.  anchors [A][4] corner
 * boxes, labels [B][N][5] rows (class, x1, y1, x2, y2) with -1 padding rows (which, as upstream, still claim an anchor
 * in the greedy phase).  An anchor takes the ground truth of highest IoU when that IoU >= iou_threshold, then every
 * label row claims the globally best remaining anchor (first maximum on ties).  Outputs: bbox_offset [B][A][4] =
 * offset_boxes(anchor, assigned box) * mask (utils/box.py:62-69), bbox_mask [B][A][4] (0 / 1), class_labels [B][A]
 * int64 (0 = background, label + 1 otherwise).  workspace: snn_roi_workspace_size(B, A, N) bytes. */
size_t snn_roi_workspace_size(int B, int A, int N);
int snn_roi_assign(const float* anchors, const float* labels, int B, int A, int N, float iou_threshold,
                   void* workspace, float* bbox_offset, float* bbox_mask, int64_t* class_labels, void* stream);
/* SODa._loss (models/soda.py:259-281) over rows = B*A anchors with K = classes + 1 logits each:
 *   loss = loss_ratio * mean(CE[label > 0]) + (1 - loss_ratio) * mean(CE[label == 0]) + mean |bbox*mask - offset*mask|
 * fwd writes the scalar loss and stats[5] = {sum CE pos, #pos, sum CE neg, #neg, sum L1} (fp64, kept for bwd);
 * bwd writes d loss / d logits [rows][K] and d loss / d bbox [rows][4], scaled by the device scalar *g_loss.
 * workspace: snn_det_loss_workspace_size(rows) bytes. */
size_t snn_det_loss_workspace_size(int64_t rows);
int snn_det_loss_fwd(const float* cls_logits, const float* bbox_preds, const float* bbox_offset, const float* bbox_mask,
                     const int64_t* class_labels, int64_t rows, int K, float loss_ratio, void* workspace, double* stats,
                     float* loss, void* stream);
int snn_det_loss_bwd(const float* cls_logits, const float* bbox_preds, const float* bbox_offset, const float* bbox_mask,
                     const int64_t* class_labels, int64_t rows, int K, float loss_ratio, const double* stats,
                     const float* g_loss, float* g_logits, float* g_bbox, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SNN_HIP_H */
