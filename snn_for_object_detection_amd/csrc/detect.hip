// Detection decode on the device: the tail of SODa.predict (models/soda.py:202-233,
// utils/box.py:72-153) without host round trips.
//
//   snn_detect_decode : per anchor conf = max_k p[k], class = argmax - 1 (background = -1), box = offset_inverse
//   snn_nms_sorted    : per-class greedy non-maximum suppression over candidates sorted by descending confidence
//
// Arithmetic follows the reference expression by expression (fp32, -ffp-contract=off), so the kept sets are those
// of utils/box.py:82-99 (pinned by tests/golden/detect_nms*.npz).
#include "snn_common.h"

namespace {

constexpr int kNmsThreads = 256;

// offset_inverse (utils/box.py:72-79) of one anchor
__device__ __forceinline__ void decode_box(const float* __restrict__ anc, const float* __restrict__ off, float* out) {
    const float ax1 = anc[0], ay1 = anc[1], ax2 = anc[2], ay2 = anc[3];
    const float acx = (ax1 + ax2) / 2, acy = (ay1 + ay2) / 2, aw = ax2 - ax1, ah = ay2 - ay1;
    const float cx = (off[0] * aw / 10) + acx;
    const float cy = (off[1] * ah / 10) + acy;
    const float w = expf(off[2] / 5) * aw;
    const float h = expf(off[3] / 5) * ah;
    out[0] = cx - 0.5f * w;
    out[1] = cy - 0.5f * h;
    out[2] = cx + 0.5f * w;
    out[3] = cy + 0.5f * h;
}

__global__ void k_detect_decode(const float* __restrict__ prob, const float* __restrict__ offsets,
                                const float* __restrict__ anchors, int A, int K, float* __restrict__ conf,
                                int* __restrict__ cls, float* __restrict__ boxes) {
    const int a = blockIdx.x * blockDim.x + threadIdx.x;
    if (a >= A) return;
    const float* p = prob + (int64_t)a * K;
    float best = p[0];
    int arg = 0;
    for (int k = 1; k < K; ++k)
        if (p[k] > best) {  // first maximum wins, as torch.max
            best = p[k];
            arg = k;
        }
    conf[a] = best;
    cls[a] = arg - 1;
    decode_box(anchors + (int64_t)a * 4, offsets + (int64_t)a * 4, boxes + (int64_t)a * 4);
}

// box_iou (utils/box.py:31-59) of two corner boxes
__device__ __forceinline__ float iou_of(const float4 p, const float4 q) {
    const float area1 = (p.z - p.x) * (p.w - p.y);
    const float area2 = (q.z - q.x) * (q.w - q.y);
    const float lx = fmaxf(p.x, q.x), ly = fmaxf(p.y, q.y);
    const float rx = fminf(p.z, q.z), ry = fminf(p.w, q.w);
    const float ow = fmaxf(rx - lx, 0.0f), oh = fmaxf(ry - ly, 0.0f);
    const float overlap = ow * oh;
    return overlap / (area1 + area2 - overlap);
}

// One block per class.  `order` holds anchor ids sorted by (class ascending, confidence descending); the members of
// class c are order[seg[c] .. seg[c+1]).  Greedy NMS in chunks of 256 candidates:
//   1. every candidate of the chunk is tested against the boxes kept so far (parallel over candidates);
//   2. the chunk's own 256 x 256 suppression relation is formed as bit masks (parallel), and one thread walks the
//      chunk in order applying them - the only sequential part, 256 trivial steps.
// Kept ids are appended to kept[seg[c] ..) in keep order; nkept[c] receives their number.
__global__ __launch_bounds__(kNmsThreads) void k_nms_sorted(const float* __restrict__ boxes,
                                                            const int* __restrict__ order,
                                                            const int* __restrict__ seg, float thr,
                                                            int* __restrict__ kept, int* __restrict__ nkept,
                                                            unsigned char* __restrict__ kept_flag,
                                                            int* __restrict__ kept_rank) {
    __shared__ float4 cand[kNmsThreads];
    __shared__ float4 ktile[kNmsThreads];
    __shared__ unsigned long long mask[kNmsThreads][kNmsThreads / 64];
    __shared__ unsigned char alive[kNmsThreads];
    __shared__ int s_nkept, s_new;
    __shared__ int newly[kNmsThreads];
    const int c = blockIdx.x;
    const int lo = seg[c], hi = seg[c + 1];
    const int tid = threadIdx.x;
    if (tid == 0) s_nkept = 0;
    __syncthreads();
    const float4* bx = reinterpret_cast<const float4*>(boxes);
    for (int base = lo; base < hi; base += kNmsThreads) {
        const int n = min(kNmsThreads, hi - base);
        const int my = tid < n ? order[base + tid] : -1;
        const float4 mine = tid < n ? bx[my] : make_float4(0.f, 0.f, 0.f, 0.f);
        cand[tid] = mine;
        bool live = tid < n;
        // 1. against everything kept so far (tiles of 256 kept boxes staged through LDS)
        const int nk = s_nkept;
        for (int kb = 0; kb < nk; kb += kNmsThreads) {
            const int m = min(kNmsThreads, nk - kb);
            __syncthreads();
            if (tid < m) ktile[tid] = bx[kept[lo + kb + tid]];
            __syncthreads();
            if (live)
                for (int j = 0; j < m; ++j)
                    if (!(iou_of(ktile[j], mine) <= thr)) {  // the reference KEEPS iou <= thr (box.py:95-97): a NaN IoU suppresses
                        live = false;
                        break;
                    }
        }
        alive[tid] = live ? 1 : 0;
        __syncthreads();
        // 2. suppression masks inside the chunk: bit j of row i = candidate i suppresses the later candidate j
        unsigned long long m4[kNmsThreads / 64] = {0ull, 0ull, 0ull, 0ull};
        if (tid < n)
            for (int j = tid + 1; j < n; ++j)
                if (!(iou_of(mine, cand[j]) <= thr)) m4[j >> 6] |= 1ull << (j & 63);
#pragma unroll
        for (int w = 0; w < kNmsThreads / 64; ++w) mask[tid][w] = m4[w];
        __syncthreads();
        if (tid == 0) {
            unsigned long long gone[kNmsThreads / 64] = {0ull, 0ull, 0ull, 0ull};
            int cnt = 0;
            for (int i = 0; i < n; ++i) {
                if (!alive[i] || ((gone[i >> 6] >> (i & 63)) & 1ull)) continue;
                newly[cnt++] = i;
#pragma unroll
                for (int w = 0; w < kNmsThreads / 64; ++w) gone[w] |= mask[i][w];
            }
            s_new = cnt;
        }
        __syncthreads();
        const int cnt = s_new, nk0 = s_nkept;
        __syncthreads();  // everyone has read the counters before thread 0 advances them
        if (tid < cnt) {
            const int id = order[base + newly[tid]];
            kept[lo + nk0 + tid] = id;
            kept_flag[id] = 1;
            kept_rank[id] = nk0 + tid;
        }
        // kept[] is re-read through global memory by the next chunk's step 1: agent-scope fence (write back + L1
        // invalidate) so that no stale line of this CU's L1 is served
        if (tid == 0) s_nkept = nk0 + cnt;
        __threadfence();
        __syncthreads();
    }
    if (tid == 0) nkept[c] = s_nkept;
}

}  // namespace

extern "C" int snn_detect_decode(const float* cls_prob, const float* offsets, const float* anchors, int A, int K,
                                 float* conf, int* cls, float* boxes, void* stream) {
    SNN_REQUIRE(cls_prob && offsets && anchors && conf && cls && boxes, "snn_detect_decode: null pointer");
    SNN_REQUIRE(A > 0 && K > 1, "snn_detect_decode: bad shape");
    hipLaunchKernelGGL(k_detect_decode, dim3((A + 255) / 256), dim3(256), 0, (hipStream_t)stream, cls_prob, offsets,
                       anchors, A, K, conf, cls, boxes);
    SNN_CHECK_LAUNCH("snn_detect_decode");
    return 0;
}

extern "C" int snn_nms_sorted(const float* boxes, const int* order, const int* seg, int num_classes,
                              float iou_threshold, int* kept, int* nkept, unsigned char* kept_flag, int* kept_rank,
                              void* stream) {
    SNN_REQUIRE(boxes && order && seg && kept && nkept && kept_flag && kept_rank, "snn_nms_sorted: null pointer");
    SNN_REQUIRE(num_classes > 0 && num_classes <= 65535, "snn_nms_sorted: bad class count");
    SNN_REQUIRE((reinterpret_cast<uintptr_t>(boxes) & 15u) == 0, "snn_nms_sorted: boxes must be 16-byte aligned");
    hipLaunchKernelGGL(k_nms_sorted, dim3((unsigned)num_classes), dim3(kNmsThreads), 0, (hipStream_t)stream, boxes,
                       order, seg, iou_threshold, kept, nkept, kept_flag, kept_rank);
    SNN_CHECK_LAUNCH("snn_nms_sorted");
    return 0;
}
