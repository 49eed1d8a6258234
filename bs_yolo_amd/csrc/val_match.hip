// Validator matching step on the device: for every image, which detections count as true positives at
// each IoU threshold.  Replaces, per batch instead of per image on the host,
//   utils/metrics.py:52-70            box_iou(gt_bboxes, detections[:, :4])            (fp32, eps 1e-7)
//   engine/validator.py:222-258       BaseValidator.match_predictions (default branch)
// as called by DetectionValidator._process_batch (models/yolo/detect/val.py:209-228).
//
// What the reference's numpy code computes, restated: zero the IoU of class-mismatched pairs; for a threshold t the
// candidate pairs are those with IoU >= t; sorted by IoU descending, `unique` over detections keeps each detection's
// best label l*(d); the following `unique` over labels runs on the detection-sorted list and therefore keeps, per label,
// the LOWEST-INDEX detection among those that chose it (not the highest-IoU one: the reference left that re-sort
// commented out).  l*(d) does not depend on t (it is the arg-max over all class-matching labels, admitted when its IoU
// reaches t), so one pass finds (l*, IoU*) per detection and each threshold is an atomicMin per label.
// Exact IoU ties between two labels of one detection are an implementation accident in the reference (unstable argsort);
// here the lower label index wins.  Compiled with -ffp-contract=off: the IoU must round exactly like the fp32 torch ops.
#include "common.h"

struct ValK {
    const float* det;
    const int32_t* counts;
    const float* gt;
    const float* gtc;
    const int32_t* gt_counts;
    unsigned char* out;
    int row, max_det, Lmax, n_iou;
    float iouv[16];
};

__global__ __launch_bounds__(256) void val_match_kernel(const ValK p) {
    extern __shared__ int winner[];  // [Lmax]
    const int b = blockIdx.x, tid = threadIdx.x;
    const int nd = min(p.counts[b], p.max_det), nl = min(p.gt_counts[b], p.Lmax);
    const float* det = p.det + (size_t)b * p.max_det * p.row;
    const float* gt = p.gt + (size_t)b * p.Lmax * 4;
    const float* gtc = p.gtc + (size_t)b * p.Lmax;
    unsigned char* out = p.out + (size_t)b * p.max_det * p.n_iou;
    constexpr int DPT = 4;  // detections per thread: max_det <= 1024
    float best[DPT];
    int bl[DPT];
#pragma unroll
    for (int j = 0; j < DPT; ++j) {
        const int d = tid + 256 * j;
        best[j] = 0.f;
        bl[j] = -1;
        if (d < nd) {
            const float x1 = det[(size_t)d * p.row], y1 = det[(size_t)d * p.row + 1], x2 = det[(size_t)d * p.row + 2],
                        y2 = det[(size_t)d * p.row + 3], cls = det[(size_t)d * p.row + 5];
            const float area2 = (x2 - x1) * (y2 - y1);
            for (int l = 0; l < nl; ++l) {
                if (gtc[l] != cls) continue;
                const float g0 = gt[4 * l], g1 = gt[4 * l + 1], g2 = gt[4 * l + 2], g3 = gt[4 * l + 3];
                const float w = fmaxf(fminf(g2, x2) - fmaxf(g0, x1), 0.f), h = fmaxf(fminf(g3, y2) - fmaxf(g1, y1), 0.f);
                const float inter = w * h;
                const float iou = inter / ((g2 - g0) * (g3 - g1) + area2 - inter + 1e-7f);
                if (iou > best[j]) { best[j] = iou; bl[j] = l; }
            }
        }
    }
    for (int i = 0; i < p.n_iou; ++i) {
        for (int l = tid; l < nl; l += 256) winner[l] = 0x7fffffff;
        __syncthreads();
#pragma unroll
        for (int j = 0; j < DPT; ++j)
            if (bl[j] >= 0 && best[j] >= p.iouv[i]) atomicMin(&winner[bl[j]], tid + 256 * j);
        __syncthreads();
#pragma unroll
        for (int j = 0; j < DPT; ++j) {
            const int d = tid + 256 * j;
            if (d < p.max_det) out[(size_t)d * p.n_iou + i] = (bl[j] >= 0 && best[j] >= p.iouv[i] && winner[bl[j]] == d) ? 1 : 0;
        }
        __syncthreads();
    }
}

extern "C" int bsy_val_match(const float* det, int row, const int32_t* counts, int B, int max_det, const float* gt_boxes,
                             const float* gt_cls, const int32_t* gt_counts, int Lmax, const float* iouv_host, int n_iou,
                             unsigned char* out, bsy_stream stream) {
    if (!det || !counts || !gt_boxes || !gt_cls || !gt_counts || !iouv_host || !out) BSY_FAIL(BSY_ERR_ARG, "val_match: null pointer");
    if (B <= 0 || max_det <= 0 || max_det > 1024 || row < 6 || Lmax <= 0 || Lmax > 8192 || n_iou <= 0 || n_iou > 16)
        BSY_FAIL(BSY_ERR_ARG, "val_match: bad sizes (max_det <= 1024, Lmax <= 8192, n_iou <= 16)");
    ValK k;
    k.det = det; k.counts = counts; k.gt = gt_boxes; k.gtc = gt_cls; k.gt_counts = gt_counts; k.out = out;
    k.row = row; k.max_det = max_det; k.Lmax = Lmax; k.n_iou = n_iou;
    for (int i = 0; i < 16; ++i) k.iouv[i] = i < n_iou ? iouv_host[i] : 2.f;
    hipLaunchKernelGGL(val_match_kernel, dim3(B), dim3(256), (size_t)Lmax * sizeof(int), (hipStream_t)stream, k);
    HIP_TRY(hipGetLastError());
    return BSY_OK;
}
