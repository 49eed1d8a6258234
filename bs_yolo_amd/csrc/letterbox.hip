// LetterBox + predictor preprocess for gfx950: B device images (HWC, BGR, u8, ragged sizes) -> one (B,3,H2,W2) RGB
// tensor scaled by 1/255 (fp16 or fp32).  Compile with -ffp-contract=off (coordinate arithmetic must round as on the CPU).
//
// Replaces LetterBox.__call__ (data/augment.py:1535-1601: cv2.resize INTER_LINEAR + cv2.copyMakeBorder(114)) and
// BasePredictor.preprocess (engine/predictor.py:116-134: stack, BGR->RGB, HWC->CHW, /255).  The geometry
// (new_unpad, left, top -- Python round() semantics) is computed on the host (bs_yolo_amd/letterbox.py) and passed in
// `geom`; this kernel does the pixels: OpenCV's 8-bit bilinear (float source coordinate, 11-bit coefficients rounded
// half-to-even, int32 horizontal pass, ((b*(v>>4))>>16 ... +2)>>2 vertical pass) incl. the exact-2x box-mean shortcut,
// the constant border, the channel swap and the scaling, one thread per output pixel (3 channels), output written
// coalesced along x in each channel plane.  HBM-bound: reads h*w*3 bytes, writes 3*H2*W2 elements.
#include "common.h"

__device__ __forceinline__ void lin_coef(int d, int dst, int src, bool reset_on_clamp, int* s0, int* s1, int* c0, int* c1) {
    const double scale = (double)src / (double)dst;
    float f = (float)(((double)d + 0.5) * scale - 0.5);
    int s = (int)floorf(f);
    f -= (float)s;
    if (reset_on_clamp) {  // horizontal axis (resize.cpp: fx = 0 when the tap is clamped)
        if (s < 0) { f = 0.f; s = 0; }
        if (s >= src - 1) { f = 0.f; s = src - 1; }
        *s0 = s;
        *s1 = min(s + 1, src - 1);
    } else {  // vertical axis: rows are clamped at fetch time, coefficients kept
        *s0 = min(max(s, 0), src - 1);
        *s1 = min(max(s + 1, 0), src - 1);
    }
    *c0 = (int)rintf((1.0f - f) * 2048.0f);
    *c1 = (int)rintf(f * 2048.0f);
}

template <typename T>
__global__ __launch_bounds__(256) void letterbox_kernel(const uint8_t* const* __restrict__ imgs,
                                                        const int32_t* __restrict__ hw,
                                                        const int32_t* __restrict__ geom, int H2, int W2,
                                                        T* __restrict__ out) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y;
    const int b = blockIdx.z;
    if (x >= W2) return;
    const int h = hw[2 * b], w = hw[2 * b + 1];
    const int nw = geom[4 * b], nh = geom[4 * b + 1], left = geom[4 * b + 2], top = geom[4 * b + 3];
    int v[3] = {114, 114, 114};  // BGR
    const int dx = x - left, dy = y - top;
    if (dx >= 0 && dx < nw && dy >= 0 && dy < nh) {
        const uint8_t* src = imgs[b];
        if (nw == w && nh == h) {
            const uint8_t* sp = src + ((size_t)dy * w + dx) * 3;
            v[0] = sp[0]; v[1] = sp[1]; v[2] = sp[2];
        } else if (w == 2 * nw && h == 2 * nh) {
            const uint8_t* p0 = src + ((size_t)(2 * dy) * w + 2 * dx) * 3;
            const uint8_t* p1 = p0 + (size_t)w * 3;
#pragma unroll
            for (int c = 0; c < 3; ++c) v[c] = ((int)p0[c] + (int)p0[3 + c] + (int)p1[c] + (int)p1[3 + c] + 2) >> 2;
        } else {
            int sx0, sx1, a0, a1, r0, r1, b0, b1;
            lin_coef(dx, nw, w, true, &sx0, &sx1, &a0, &a1);
            lin_coef(dy, nh, h, false, &r0, &r1, &b0, &b1);
            const uint8_t* t0 = src + ((size_t)r0 * w + sx0) * 3;
            const uint8_t* t1 = src + ((size_t)r0 * w + sx1) * 3;
            const uint8_t* u0 = src + ((size_t)r1 * w + sx0) * 3;
            const uint8_t* u1 = src + ((size_t)r1 * w + sx1) * 3;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const int topv = (int)t0[c] * a0 + (int)t1[c] * a1;
                const int botv = (int)u0[c] * a0 + (int)u1[c] * a1;
                int r = (((b0 * (topv >> 4)) >> 16) + ((b1 * (botv >> 4)) >> 16) + 2) >> 2;
                v[c] = min(max(r, 0), 255);
            }
        }
    }
    const size_t plane = (size_t)H2 * W2;
    T* op = out + (size_t)b * 3 * plane + (size_t)y * W2 + x;
    // BGR -> RGB; `im /= 255` in the tensor's dtype (predictor.py:131-133)
    op[0] = (T)((float)(T)(float)v[2] / (float)(T)255.0f);
    op[plane] = (T)((float)(T)(float)v[1] / (float)(T)255.0f);
    op[2 * plane] = (T)((float)(T)(float)v[0] / (float)(T)255.0f);
}

extern "C" int bsy_letterbox(const uint8_t* const* imgs, const int32_t* hw, const int32_t* geom, int B, int H2, int W2,
                             void* out, int out_dtype, bsy_stream stream) {
    if (!imgs || !hw || !geom || !out || B <= 0 || H2 <= 0 || W2 <= 0) BSY_FAIL(BSY_ERR_ARG, "letterbox: bad argument");
    dim3 grid((W2 + 255) / 256, H2, B);
    if (out_dtype == BSY_F16)
        hipLaunchKernelGGL(letterbox_kernel<half_t>, grid, dim3(256), 0, (hipStream_t)stream, imgs, hw, geom, H2, W2,
                           (half_t*)out);
    else if (out_dtype == BSY_F32)
        hipLaunchKernelGGL(letterbox_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, imgs, hw, geom, H2, W2,
                           (float*)out);
    else
        BSY_FAIL(BSY_ERR_ARG, "letterbox: dtype %d unsupported", out_dtype);
    HIP_TRY(hipGetLastError());
    return BSY_OK;
}
