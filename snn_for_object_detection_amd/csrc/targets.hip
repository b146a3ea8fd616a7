// Training targets and the detection loss on the device: the step after the network.
//
//   snn_roi_assign : anchor <-> ground-truth assignment + offset / mask / class targets of one batch
//                    (utils/roi.py:18-109, utils/box.py:31-69) - one block per sample instead of ~150 tiny tensor
//                    launches per sample and a Python loop over the ground-truth rows;
//   snn_det_loss_fwd / _bwd : loss_ratio * mean(CE[positive]) + (1 - loss_ratio) * mean(CE[negative]) +
//                    mean(L1(bbox * mask, offset * mask)) and its gradient (models/soda.py:259-281).
//
// Arithmetic follows the reference expression by expression in fp32 (-ffp-contract=off), so the assignment (an
// integer result) is the reference's: first maximum wins in the per-anchor max and in the global argmax, padding rows
// (-1) of the label tensor take part in the greedy phase exactly as they do upstream (SURVEY a-11).  The loss sums are
// accumulated in fp64 (torch: fp32 pairwise sums) - a 1e-7 relative difference.
#include "snn_common.h"

namespace {

constexpr int kRoiThreads = 1024;

struct ArgMax {
    float v;
    int idx;
};
__device__ __forceinline__ ArgMax better(ArgMax a, ArgMax b) {  // larger value; on ties the smaller flat index
    return (b.v > a.v || (b.v == a.v && b.idx < a.idx)) ? b : a;
}

// box_iou (utils/box.py:31-59) of one anchor and one ground-truth box
__device__ __forceinline__ float iou_pair(const float4 p, const float* __restrict__ q) {
    const float area1 = (p.z - p.x) * (p.w - p.y);
    const float area2 = (q[2] - q[0]) * (q[3] - q[1]);
    const float lx = fmaxf(p.x, q[0]), ly = fmaxf(p.y, q[1]);
    const float rx = fminf(p.z, q[2]), ry = fminf(p.w, q[3]);
    const float ow = fmaxf(rx - lx, 0.0f), oh = fmaxf(ry - ly, 0.0f);
    const float overlap = ow * oh;
    return overlap / (area1 + area2 - overlap);
}

__global__ __launch_bounds__(kRoiThreads) void k_roi_assign(const float* __restrict__ anchors,
                                                            const float* __restrict__ labels, int A, int N, float thr,
                                                            float* __restrict__ iou_ws, int* __restrict__ amap_ws,
                                                            float* __restrict__ offset, float* __restrict__ mask,
                                                            int64_t* __restrict__ cls) {
    __shared__ ArgMax red[kRoiThreads / 64];
    __shared__ ArgMax winner;
    const int b = blockIdx.x, tid = threadIdx.x;
    const float* lab = labels + (int64_t)b * N * 5;
    float* iou = iou_ws + (int64_t)b * A * N;
    int* amap = amap_ws + (int64_t)b * A;
    // ---- IoU matrix; an anchor takes the ground truth of highest IoU when that reaches the threshold (roi.py:78-93)
    for (int a = tid; a < A; a += kRoiThreads) {
        const float4 anc = *reinterpret_cast<const float4*>(anchors + (int64_t)a * 4);
        float best = 0.f;
        int arg = 0;
        for (int j = 0; j < N; ++j) {
            const float v = iou_pair(anc, lab + j * 5 + 1);
            iou[(int64_t)a * N + j] = v;
            if (j == 0 || v > best) {  // first maximum wins, as torch.max(dim=1)
                best = v;
                arg = j;
            }
        }
        amap[a] = best >= thr ? arg : -1;
    }
    __syncthreads();
    // ---- every ground-truth ROW claims the globally best remaining anchor (roi.py:95-108), padding rows included
    const int total = A * N;
    for (int round = 0; round < N; ++round) {
        ArgMax m = {-INFINITY, 0x7fffffff};
        for (int i = tid; i < total; i += kRoiThreads) {
            const float v = iou[i];
            if (v > m.v) m = {v, i};   // strictly greater: the first (lowest) index of this thread's maxima stays
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            ArgMax other = {__shfl_xor(m.v, o), __shfl_xor(m.idx, o)};
            m = better(m, other);
        }
        if ((tid & 63) == 0) red[tid >> 6] = m;
        __syncthreads();
        if (tid == 0) {
            ArgMax w = red[0];
            for (int k = 1; k < kRoiThreads / 64; ++k) w = better(w, red[k]);
            if (w.idx == 0x7fffffff) w.idx = 0;   // every entry NaN / -inf: torch.argmax returns a valid index too
            winner = w;
        }
        __syncthreads();
        const int box_idx = winner.idx % N;
        const int anc_idx = (int)((float)winner.idx / (float)N);   // float division + truncation, as the reference
        if (tid == 0) amap[anc_idx] = box_idx;
        for (int a = tid; a < A; a += kRoiThreads) iou[(int64_t)a * N + box_idx] = -1.0f;
        for (int j = tid; j < N; j += kRoiThreads) iou[(int64_t)anc_idx * N + j] = -1.0f;
        __syncthreads();
    }
    // ---- targets (roi.py:41-58, box.py:62-69): class = label + 1, offsets of the assigned box, all times the mask
    for (int a = tid; a < A; a += kRoiThreads) {
        const int g = amap[a];
        const float m = g >= 0 ? 1.0f : 0.0f;
        const float* gt = lab + (g >= 0 ? g : 0) * 5;
        const float bx1 = g >= 0 ? gt[1] : 0.f, by1 = g >= 0 ? gt[2] : 0.f;
        const float bx2 = g >= 0 ? gt[3] : 0.f, by2 = g >= 0 ? gt[4] : 0.f;
        const float4 anc = *reinterpret_cast<const float4*>(anchors + (int64_t)a * 4);
        const float acx = (anc.x + anc.z) / 2, acy = (anc.y + anc.w) / 2, aw = anc.z - anc.x, ah = anc.w - anc.y;
        const float tcx = (bx1 + bx2) / 2, tcy = (by1 + by2) / 2, tw = bx2 - bx1, th = by2 - by1;
        float4 o;
        o.x = (10 * (tcx - acx) / aw) * m;
        o.y = (10 * (tcy - acy) / ah) * m;
        o.z = (5 * logf(1e-6f + tw / aw)) * m;
        o.w = (5 * logf(1e-6f + th / ah)) * m;
        const int64_t row = (int64_t)b * A + a;
        *reinterpret_cast<float4*>(offset + row * 4) = o;
        *reinterpret_cast<float4*>(mask + row * 4) = make_float4(m, m, m, m);
        cls[row] = g >= 0 ? (int64_t)gt[0] + 1 : 0;
    }
}

// ------------------------------------------------------------------------------------------ loss
constexpr int kLossThreads = 256;
constexpr int kMaxClasses = 64;

__device__ __forceinline__ double block_sum(double v, double* scratch) {  // all threads get the sum
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) scratch[threadIdx.x >> 6] = v;
    __syncthreads();
    double s = 0.0;
    for (int k = 0; k < kLossThreads / 64; ++k) s += scratch[k];
    return s;
}

// partial[block][5] = {sum CE over positives, positives, sum CE over negatives, negatives, sum |bbox*m - off*m|}
__global__ __launch_bounds__(kLossThreads) void k_det_loss_partial(const float* __restrict__ logits,
                                                                   const float* __restrict__ bbox,
                                                                   const float* __restrict__ offset,
                                                                   const float* __restrict__ mask,
                                                                   const int64_t* __restrict__ cls, int64_t R, int K,
                                                                   double* __restrict__ partial) {
    __shared__ double scratch[kLossThreads / 64];
    double s_pos = 0, n_pos = 0, s_neg = 0, n_neg = 0, s_l1 = 0;
    for (int64_t r = (int64_t)blockIdx.x * kLossThreads + threadIdx.x; r < R; r += (int64_t)gridDim.x * kLossThreads) {
        const float* x = logits + r * K;
        float mx = x[0];
        for (int k = 1; k < K; ++k) mx = fmaxf(mx, x[k]);
        float se = 0.f;
        for (int k = 0; k < K; ++k) se += expf(x[k] - mx);
        const int64_t y = cls[r];
        const float ce = (mx + logf(se)) - x[y];   // -log_softmax(x)[y]
        if (y > 0) {
            s_pos += ce;
            n_pos += 1;
        } else {
            s_neg += ce;
            n_neg += 1;
        }
        const float4 bb = *reinterpret_cast<const float4*>(bbox + r * 4);
        const float4 of = *reinterpret_cast<const float4*>(offset + r * 4);
        const float4 mk = *reinterpret_cast<const float4*>(mask + r * 4);
        s_l1 += (double)fabsf(bb.x * mk.x - of.x * mk.x) + (double)fabsf(bb.y * mk.y - of.y * mk.y) +
                (double)fabsf(bb.z * mk.z - of.z * mk.z) + (double)fabsf(bb.w * mk.w - of.w * mk.w);
    }
    const double a = block_sum(s_pos, scratch), bq = block_sum(n_pos, scratch), c = block_sum(s_neg, scratch);
    const double d = block_sum(n_neg, scratch), e = block_sum(s_l1, scratch);
    if (threadIdx.x == 0) {
        double* p = partial + (int64_t)blockIdx.x * 5;
        p[0] = a; p[1] = bq; p[2] = c; p[3] = d; p[4] = e;
    }
}

// stats[5] = totals (block order: reproducible); loss = ratio * pos / n_pos + (1 - ratio) * neg / n_neg + l1 / (4 R)
__global__ void k_det_loss_final(const double* __restrict__ partial, int nblocks, int64_t R, float ratio,
                                 double* __restrict__ stats, float* __restrict__ loss) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    double t[5] = {0, 0, 0, 0, 0};
    for (int b = 0; b < nblocks; ++b)
        for (int k = 0; k < 5; ++k) t[k] += partial[(int64_t)b * 5 + k];
    for (int k = 0; k < 5; ++k) stats[k] = t[k];
    const float gt = (float)(t[0] / t[1]), bg = (float)(t[2] / t[3]), l1 = (float)(t[4] / (4.0 * (double)R));
    *loss = (gt * ratio + bg * (1 - ratio)) + l1;   // soda.py:277-281, same order of the three terms
}

__global__ __launch_bounds__(kLossThreads) void k_det_loss_bwd(const float* __restrict__ logits,
                                                               const float* __restrict__ bbox,
                                                               const float* __restrict__ offset,
                                                               const float* __restrict__ mask,
                                                               const int64_t* __restrict__ cls, int64_t R, int K,
                                                               const double* __restrict__ stats, float ratio,
                                                               const float* __restrict__ g_loss,
                                                               float* __restrict__ g_logits, float* __restrict__ g_bbox) {
    const float g = *g_loss;
    const float w_pos = g * ratio / (float)stats[1], w_neg = g * (1 - ratio) / (float)stats[3];
    const float w_l1 = g / (4.0f * (float)R);
    for (int64_t r = (int64_t)blockIdx.x * kLossThreads + threadIdx.x; r < R; r += (int64_t)gridDim.x * kLossThreads) {
        const float* x = logits + r * K;
        float mx = x[0];
        for (int k = 1; k < K; ++k) mx = fmaxf(mx, x[k]);
        float se = 0.f;
        for (int k = 0; k < K; ++k) se += expf(x[k] - mx);
        const int64_t y = cls[r];
        const float w = y > 0 ? w_pos : w_neg;
        for (int k = 0; k < K; ++k) {
            const float p = expf(x[k] - mx) / se;
            g_logits[r * K + k] = w * (p - (k == y ? 1.0f : 0.0f));
        }
        const float4 bb = *reinterpret_cast<const float4*>(bbox + r * 4);
        const float4 of = *reinterpret_cast<const float4*>(offset + r * 4);
        const float4 mk = *reinterpret_cast<const float4*>(mask + r * 4);
        auto sgn = [](float v) { return v > 0.f ? 1.0f : (v < 0.f ? -1.0f : 0.0f); };
        float4 o;
        o.x = w_l1 * sgn(bb.x * mk.x - of.x * mk.x) * mk.x;
        o.y = w_l1 * sgn(bb.y * mk.y - of.y * mk.y) * mk.y;
        o.z = w_l1 * sgn(bb.z * mk.z - of.z * mk.z) * mk.z;
        o.w = w_l1 * sgn(bb.w * mk.w - of.w * mk.w) * mk.w;
        *reinterpret_cast<float4*>(g_bbox + r * 4) = o;
    }
}

static bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }
static int loss_blocks(int64_t R) {
    int64_t b = snn_ceil_div(R, (int64_t)kLossThreads * 4);
    if (b > 1024) b = 1024;
    return b < 1 ? 1 : (int)b;
}

}  // namespace

extern "C" size_t snn_roi_workspace_size(int B, int A, int N) {  // bytes: IoU matrix + assignment map per sample
    return (size_t)B * A * ((size_t)N * sizeof(float) + sizeof(int));
}

extern "C" int snn_roi_assign(const float* anchors, const float* labels, int B, int A, int N, float iou_threshold,
                              void* workspace, float* bbox_offset, float* bbox_mask, int64_t* class_labels,
                              void* stream) {
    SNN_REQUIRE(anchors && labels && workspace && bbox_offset && bbox_mask && class_labels, "snn_roi_assign: null pointer");
    SNN_REQUIRE(B > 0 && A > 0 && N > 0 && (int64_t)A * N < 0x7fffffffLL, "snn_roi_assign: bad shape");
    SNN_REQUIRE(aligned16(anchors) && aligned16(bbox_offset) && aligned16(bbox_mask) && aligned16(workspace),
                "snn_roi_assign: buffers must be 16-byte aligned");
    float* iou = static_cast<float*>(workspace);
    int* amap = reinterpret_cast<int*>(iou + (size_t)B * A * N);
    hipLaunchKernelGGL(k_roi_assign, dim3((unsigned)B), dim3(kRoiThreads), 0, (hipStream_t)stream, anchors, labels, A, N,
                       iou_threshold, iou, amap, bbox_offset, bbox_mask, class_labels);
    SNN_CHECK_LAUNCH("snn_roi_assign");
    return 0;
}

extern "C" size_t snn_det_loss_workspace_size(int64_t rows) { return (size_t)loss_blocks(rows) * 5 * sizeof(double); }

extern "C" int snn_det_loss_fwd(const float* cls_logits, const float* bbox_preds, const float* bbox_offset,
                                const float* bbox_mask, const int64_t* class_labels, int64_t rows, int K,
                                float loss_ratio, void* workspace, double* stats, float* loss, void* stream) {
    SNN_REQUIRE(cls_logits && bbox_preds && bbox_offset && bbox_mask && class_labels && workspace && stats && loss,
                "snn_det_loss_fwd: null pointer");
    SNN_REQUIRE(rows > 0 && K > 1 && K <= kMaxClasses, "snn_det_loss_fwd: bad shape");
    SNN_REQUIRE(aligned16(bbox_preds) && aligned16(bbox_offset) && aligned16(bbox_mask),
                "snn_det_loss_fwd: box tensors must be 16-byte aligned");
    const int nb = loss_blocks(rows);
    hipLaunchKernelGGL(k_det_loss_partial, dim3((unsigned)nb), dim3(kLossThreads), 0, (hipStream_t)stream, cls_logits,
                       bbox_preds, bbox_offset, bbox_mask, class_labels, rows, K, static_cast<double*>(workspace));
    hipLaunchKernelGGL(k_det_loss_final, dim3(1), dim3(64), 0, (hipStream_t)stream,
                       static_cast<const double*>(workspace), nb, rows, loss_ratio, stats, loss);
    SNN_CHECK_LAUNCH("snn_det_loss_fwd");
    return 0;
}

extern "C" int snn_det_loss_bwd(const float* cls_logits, const float* bbox_preds, const float* bbox_offset,
                                const float* bbox_mask, const int64_t* class_labels, int64_t rows, int K,
                                float loss_ratio, const double* stats, const float* g_loss, float* g_logits,
                                float* g_bbox, void* stream) {
    SNN_REQUIRE(cls_logits && bbox_preds && bbox_offset && bbox_mask && class_labels && stats && g_loss && g_logits &&
                    g_bbox, "snn_det_loss_bwd: null pointer");
    SNN_REQUIRE(rows > 0 && K > 1 && K <= kMaxClasses, "snn_det_loss_bwd: bad shape");
    SNN_REQUIRE(aligned16(bbox_preds) && aligned16(bbox_offset) && aligned16(bbox_mask) && aligned16(g_bbox),
                "snn_det_loss_bwd: box tensors must be 16-byte aligned");
    hipLaunchKernelGGL(k_det_loss_bwd, dim3((unsigned)loss_blocks(rows)), dim3(kLossThreads), 0, (hipStream_t)stream,
                       cls_logits, bbox_preds, bbox_offset, bbox_mask, class_labels, rows, K, stats, loss_ratio, g_loss,
                       g_logits, g_bbox);
    SNN_CHECK_LAUNCH("snn_det_loss_bwd");
    return 0;
}
