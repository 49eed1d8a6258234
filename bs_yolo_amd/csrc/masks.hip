// Instance-mask assembly for gfx950: process_mask (utils/ops.py:663-694) + crop_mask (:644-660).
//
//   masks = (masks_in @ protos.view(nm, -1)).view(n, mh, mw)            -> lowres_kernel (fp32 dot over nm prototypes,
//   masks = crop_mask(masks, boxes * (mw/iw, mh/ih))                       box crop applied in the same pass)
//   masks = F.interpolate(masks[None], (ih, iw), "bilinear", align_corners=False)   -> upsample_kernel (PyTorch's
//   return masks.gt_(0.0)                                                    half-pixel source coordinate, clamp at 0)
// process_mask_native (ops.py:696-709) and scale_masks (:712-737) -- the retina_masks path of segment/predict.py:48-50 -- resize
// the un-padded window of the prototype-resolution masks straight to the ORIGINAL image size and crop there.
// One image per call (the reference calls it per image from segment/predict.py:53).  Both kernels are HBM-bound:
// lowres reads nm*mh*mw prototypes once per 8 masks, upsample writes n*ih*iw outputs.
#include "common.h"

template <typename T, bool CROP>
__global__ __launch_bounds__(256) void mask_lowres_kernel(const T* __restrict__ protos, int nm, int mh, int mw,
                                                          const float* __restrict__ coef, int ldc,
                                                          const float* __restrict__ boxes, int ldb, int n, float wr,
                                                          float hr, float* __restrict__ low) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;  // pixel of the prototype map
    const int n0 = blockIdx.y * 8;                        // 8 masks per thread: each prototype value is read once for 8
    if (p >= mh * mw) return;
    float acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = 0.f;
    for (int k = 0; k < nm; ++k) {
        const float v = (float)protos[(size_t)k * mh * mw + p];
#pragma unroll
        for (int j = 0; j < 8; ++j)
            if (n0 + j < n) acc[j] = fmaf(coef[(size_t)(n0 + j) * ldc + k], v, acc[j]);
    }
    const float x = (float)(p % mw), y = (float)(p / mw);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        if (n0 + j >= n) break;
        // process_mask crops at prototype resolution (CROP); process_mask_native crops after the resize (boxes unused here)
        float v = acc[j];
        if (CROP) {
            const float* b = boxes + (size_t)(n0 + j) * ldb;
            const float x1 = b[0] * wr, y1 = b[1] * hr, x2 = b[2] * wr, y2 = b[3] * hr;  // ops.py:684-688
            const bool inside = x >= x1 && x < x2 && y >= y1 && y < y2;                  // ops.py:660
            v = inside ? v : 0.f;
        }
        low[((size_t)(n0 + j) * mh * mw) + p] = v;
    }
}

template <typename O>
__global__ __launch_bounds__(256) void mask_upsample_kernel(const float* __restrict__ low, int mh, int mw, int ih, int iw,
                                                            O* __restrict__ out) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y, m = blockIdx.z;
    if (x >= iw) return;
    // aten upsample_bilinear2d, align_corners=False: src = max(scale * (dst + 0.5) - 0.5, 0)
    const float sy = fmaxf(((float)mh / (float)ih) * ((float)y + 0.5f) - 0.5f, 0.f);
    const float sx = fmaxf(((float)mw / (float)iw) * ((float)x + 0.5f) - 0.5f, 0.f);
    const int y0 = (int)sy, x0 = (int)sx;
    const int y1 = y0 + (y0 < mh - 1 ? 1 : 0), x1 = x0 + (x0 < mw - 1 ? 1 : 0);
    const float ly1 = sy - (float)y0, lx1 = sx - (float)x0, ly0 = 1.f - ly1, lx0 = 1.f - lx1;
    const float* lp = low + (size_t)m * mh * mw;
    const float v = ly0 * (lx0 * lp[y0 * mw + x0] + lx1 * lp[y0 * mw + x1]) +
                    ly1 * (lx0 * lp[y1 * mw + x0] + lx1 * lp[y1 * mw + x1]);
    out[((size_t)m * ih + y) * iw + x] = (O)(v > 0.f ? 1 : 0);
}

template <typename O>
__global__ __launch_bounds__(256) void mask_threshold_kernel(const float* __restrict__ low, size_t total, O* __restrict__ out) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < total) out[i] = (O)(low[i] > 0.f ? 1 : 0);
}

// scale_masks (utils/ops.py:712-737): the window [top, bottom) x [left, right) of each (mh, mw) map -- the letterbox padding cut
// away -- resized to (oh, ow) with F.interpolate(bilinear, align_corners=False) arithmetic.  THRESH: process_mask_native's tail
// (ops.py:707-709): crop_mask with the boxes given in OUTPUT pixels, then `> 0`.
template <typename I, typename O, bool THRESH>
__global__ __launch_bounds__(256) void mask_window_resize_kernel(const I* __restrict__ src, int mh, int mw, int top, int left, int hs,
                                                                 int ws, int oh, int ow, const float* __restrict__ boxes, int ldb,
                                                                 O* __restrict__ out) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y, m = blockIdx.z;
    if (x >= ow) return;
    const float sy = fmaxf(((float)hs / (float)oh) * ((float)y + 0.5f) - 0.5f, 0.f);
    const float sx = fmaxf(((float)ws / (float)ow) * ((float)x + 0.5f) - 0.5f, 0.f);
    const int y0 = (int)sy, x0 = (int)sx;
    const int y1 = y0 + (y0 < hs - 1 ? 1 : 0), x1 = x0 + (x0 < ws - 1 ? 1 : 0);
    const float ly1 = sy - (float)y0, lx1 = sx - (float)x0, ly0 = 1.f - ly1, lx0 = 1.f - lx1;
    const I* lp = src + (size_t)m * mh * mw + (size_t)top * mw + left;
    const float v = ly0 * (lx0 * (float)lp[y0 * mw + x0] + lx1 * (float)lp[y0 * mw + x1]) +
                    ly1 * (lx0 * (float)lp[y1 * mw + x0] + lx1 * (float)lp[y1 * mw + x1]);
    const size_t oi = ((size_t)m * oh + y) * ow + x;
    if (THRESH) {
        const float* b = boxes + (size_t)m * ldb;
        const float fx = (float)x, fy = (float)y;
        const bool in = fx >= b[0] && fx < b[2] && fy >= b[1] && fy < b[3];  // crop_mask, ops.py:644-660
        out[oi] = (O)((in && v > 0.f) ? 1 : 0);
    } else {
        out[oi] = (O)v;
    }
}

extern "C" int bsy_scale_masks(const void* masks, int dtype, int n, int mh, int mw, int top, int left, int bottom, int right, int oh,
                               int ow, void* out, bsy_stream stream) {
    hipStream_t s = (hipStream_t)stream;
    if (n == 0) return BSY_OK;
    if (!masks || !out || n < 0 || mh <= 0 || mw <= 0 || oh <= 0 || ow <= 0) BSY_FAIL(BSY_ERR_ARG, "scale_masks: bad argument");
    if (top < 0 || left < 0 || bottom > mh || right > mw || bottom <= top || right <= left)
        BSY_FAIL(BSY_ERR_ARG, "scale_masks: empty or out-of-range window [%d:%d, %d:%d] of a %d x %d map", top, bottom, left, right, mh, mw);
    dim3 g((ow + 255) / 256, oh, n);
    if (dtype == BSY_F32)
        hipLaunchKernelGGL((mask_window_resize_kernel<float, float, false>), g, dim3(256), 0, s, (const float*)masks, mh, mw, top, left,
                           bottom - top, right - left, oh, ow, nullptr, 0, (float*)out);
    else if (dtype == BSY_F16)
        hipLaunchKernelGGL((mask_window_resize_kernel<half_t, half_t, false>), g, dim3(256), 0, s, (const half_t*)masks, mh, mw, top, left,
                           bottom - top, right - left, oh, ow, nullptr, 0, (half_t*)out);
    else
        BSY_FAIL(BSY_ERR_ARG, "scale_masks: dtype %d unsupported", dtype);
    HIP_TRY(hipGetLastError());
    return BSY_OK;
}

extern "C" int bsy_process_mask_native(const void* protos, int proto_dtype, int nm, int mh, int mw, const float* coef, int ldc,
                                       const float* boxes, int ldb, int n, int top, int left, int bottom, int right, int oh, int ow,
                                       float* lowres, void* out, int out_dtype, bsy_stream stream) {
    hipStream_t s = (hipStream_t)stream;
    if (n == 0) return BSY_OK;
    if (!protos || !coef || !boxes || !lowres || !out || nm <= 0 || mh <= 0 || mw <= 0 || n < 0 || oh <= 0 || ow <= 0)
        BSY_FAIL(BSY_ERR_ARG, "process_mask_native: bad argument");
    if (top < 0 || left < 0 || bottom > mh || right > mw || bottom <= top || right <= left)
        BSY_FAIL(BSY_ERR_ARG, "process_mask_native: empty or out-of-range window");
    if (out_dtype != BSY_U8 && out_dtype != BSY_F32) BSY_FAIL(BSY_ERR_ARG, "process_mask_native: output dtype must be u8 or f32");
    dim3 g1((mh * mw + 255) / 256, (n + 7) / 8);
    if (proto_dtype == BSY_F16)
        hipLaunchKernelGGL((mask_lowres_kernel<half_t, false>), g1, dim3(256), 0, s, (const half_t*)protos, nm, mh, mw, coef, ldc, nullptr, 0, n, 1.f, 1.f, lowres);
    else if (proto_dtype == BSY_F32)
        hipLaunchKernelGGL((mask_lowres_kernel<float, false>), g1, dim3(256), 0, s, (const float*)protos, nm, mh, mw, coef, ldc, nullptr, 0, n, 1.f, 1.f, lowres);
    else
        BSY_FAIL(BSY_ERR_ARG, "process_mask_native: proto dtype %d unsupported", proto_dtype);
    dim3 g2((ow + 255) / 256, oh, n);
    if (out_dtype == BSY_U8)
        hipLaunchKernelGGL((mask_window_resize_kernel<float, uint8_t, true>), g2, dim3(256), 0, s, lowres, mh, mw, top, left, bottom - top,
                           right - left, oh, ow, boxes, ldb, (uint8_t*)out);
    else
        hipLaunchKernelGGL((mask_window_resize_kernel<float, float, true>), g2, dim3(256), 0, s, lowres, mh, mw, top, left, bottom - top,
                           right - left, oh, ow, boxes, ldb, (float*)out);
    HIP_TRY(hipGetLastError());
    return BSY_OK;
}

extern "C" int bsy_process_mask(const void* protos, int proto_dtype, int nm, int mh, int mw, const float* coef, int ldc,
                                const float* boxes, int ldb, int n, int ih, int iw, int upsample, float* lowres,
                                void* out, int out_dtype, bsy_stream stream) {
    hipStream_t s = (hipStream_t)stream;
    if (n == 0) return BSY_OK;
    if (!protos || !coef || !boxes || !lowres || !out || nm <= 0 || mh <= 0 || mw <= 0 || n < 0 || ih <= 0 || iw <= 0)
        BSY_FAIL(BSY_ERR_ARG, "process_mask: bad argument");
    if (out_dtype != BSY_U8 && out_dtype != BSY_F32) BSY_FAIL(BSY_ERR_ARG, "process_mask: output dtype must be u8 or f32");
    const float wr = (float)mw / (float)iw, hr = (float)mh / (float)ih;
    dim3 g1((mh * mw + 255) / 256, (n + 7) / 8);
    if (proto_dtype == BSY_F16)
        hipLaunchKernelGGL((mask_lowres_kernel<half_t, true>), g1, dim3(256), 0, s, (const half_t*)protos, nm, mh, mw, coef, ldc,
                           boxes, ldb, n, wr, hr, lowres);
    else if (proto_dtype == BSY_F32)
        hipLaunchKernelGGL((mask_lowres_kernel<float, true>), g1, dim3(256), 0, s, (const float*)protos, nm, mh, mw, coef, ldc,
                           boxes, ldb, n, wr, hr, lowres);
    else
        BSY_FAIL(BSY_ERR_ARG, "process_mask: proto dtype %d unsupported", proto_dtype);
    if (upsample) {
        dim3 g2((iw + 255) / 256, ih, n);
        if (out_dtype == BSY_U8) hipLaunchKernelGGL(mask_upsample_kernel<uint8_t>, g2, dim3(256), 0, s, lowres, mh, mw, ih, iw, (uint8_t*)out);
        else hipLaunchKernelGGL(mask_upsample_kernel<float>, g2, dim3(256), 0, s, lowres, mh, mw, ih, iw, (float*)out);
    } else {
        const size_t total = (size_t)n * mh * mw;
        dim3 g2((unsigned)((total + 255) / 256));
        if (out_dtype == BSY_U8) hipLaunchKernelGGL(mask_threshold_kernel<uint8_t>, g2, dim3(256), 0, s, lowres, total, (uint8_t*)out);
        else hipLaunchKernelGGL(mask_threshold_kernel<float>, g2, dim3(256), 0, s, lowres, total, (float*)out);
    }
    HIP_TRY(hipGetLastError());
    return BSY_OK;
}
