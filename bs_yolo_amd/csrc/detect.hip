// Detect head tail for gfx950: DFL expectation + dist2bbox + sigmoid -> (B, 4+nc+nm, A) channel-major prediction.
//
// Replaces Detect._inference (nn/modules/head.py:100-131), DFL.forward (nn/modules/block.py:58-77),
// make_anchors / dist2bbox (utils/tal.py:371-395) and, for Segment, the mask-coefficient concat (head.py:190-197).
// The reference runs ~10 elementwise/softmax launches over (B,144,A); here one thread owns one anchor: it reads its
// 64 box logits and nc class logits (fp32, written by the last 1x1 convs), does the four 16-bin softmax expectations
// in registers and writes each output channel coalesced along the anchor axis.  All arithmetic fp32.
//
// raw_nchw_kernel rebuilds the `x` list Detect.forward returns next to y (head.py:69-74: cat(cv2(x), cv3(x)) per
// level, BCHW); only launched when the caller asks for it.
#include "common.h"

struct DecodeK {
    const float* box[3];
    const float* cls[3];
    const float* msk[3];
    int ldb[3], ldc[3], ldm[3];
    int h[3], w[3], a0[3];  // a0 = first anchor index of the level
    float stride[3];
    int nl, B, nc, nm, A;
};

template <typename T>
__global__ __launch_bounds__(256) void decode_kernel(const DecodeK p, T* __restrict__ y) {
    const int a = blockIdx.x * blockDim.x + threadIdx.x;
    const int b = blockIdx.y;
    if (a >= p.A) return;
    int l = 0;
    if (p.nl > 1 && a >= p.a0[1]) l = 1;
    if (p.nl > 2 && a >= p.a0[2]) l = 2;
    const int la = a - p.a0[l];
    const int hw = p.h[l] * p.w[l];
    const size_t pix = (size_t)b * hw + la;
    const float ax = (float)(la % p.w[l]) + 0.5f;
    const float ay = (float)(la / p.w[l]) + 0.5f;
    const float* bp = p.box[l] + pix * p.ldb[l];
    float d[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        float v[16];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 t = *reinterpret_cast<const f32x4*>(bp + 16 * s + 4 * q);
            v[4 * q] = t[0]; v[4 * q + 1] = t[1]; v[4 * q + 2] = t[2]; v[4 * q + 3] = t[3];
        }
        float m = v[0];
#pragma unroll
        for (int i = 1; i < 16; ++i) m = fmaxf(m, v[i]);
        float den = 0.f, num = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const float e = __expf(v[i] - m);
            den += e;
            num += e * (float)i;
        }
        d[s] = num / den;
    }
    const float x1 = ax - d[0], y1 = ay - d[1], x2 = ax + d[2], y2 = ay + d[3];
    const float st = p.stride[l];
    T* yp = y + (size_t)b * (4 + p.nc + p.nm) * p.A + a;
    yp[0] = (T)(((x1 + x2) * 0.5f) * st);
    yp[(size_t)p.A] = (T)(((y1 + y2) * 0.5f) * st);
    yp[(size_t)2 * p.A] = (T)((x2 - x1) * st);
    yp[(size_t)3 * p.A] = (T)((y2 - y1) * st);
    const float* cp = p.cls[l] + pix * p.ldc[l];
    for (int c = 0; c < p.nc; ++c) yp[(size_t)(4 + c) * p.A] = (T)(1.0f / (1.0f + __expf(-cp[c])));
    if (p.nm) {
        const float* mp = p.msk[l] + pix * p.ldm[l];
        for (int c = 0; c < p.nm; ++c) yp[(size_t)(4 + p.nc + c) * p.A] = (T)mp[c];
    }
}

int launch_decode(const DecodeArgs& a, hipStream_t s) {
    if (a.nl < 1 || a.nl > 3 || a.nc < 1 || a.B < 1) BSY_FAIL(BSY_ERR_ARG, "decode: bad sizes");
    DecodeK k;
    int A = 0;
    for (int l = 0; l < 3; ++l) {
        const bool on = l < a.nl;
        k.box[l] = on ? a.box[l] : nullptr; k.cls[l] = on ? a.cls[l] : nullptr; k.msk[l] = on ? a.msk[l] : nullptr;
        k.ldb[l] = on ? a.ldb[l] : 0; k.ldc[l] = on ? a.ldc[l] : 0; k.ldm[l] = on ? a.ldm[l] : 0;
        k.h[l] = on ? a.h[l] : 0; k.w[l] = on ? a.w[l] : 0; k.stride[l] = on ? a.stride[l] : 0.f;
        k.a0[l] = A;
        if (on) {
            if (!a.box[l] || !a.cls[l] || (a.ldb[l] & 3) || ((uintptr_t)a.box[l] & 15) || (a.nm && !a.msk[l]))
                BSY_FAIL(BSY_ERR_ARG, "decode: level %d bad layout", l);
            A += a.h[l] * a.w[l];
        }
    }
    if (a.A && a.A != A) BSY_FAIL(BSY_ERR_ARG, "decode: anchor count mismatch (%d vs %d)", a.A, A);
    k.nl = a.nl; k.B = a.B; k.nc = a.nc; k.nm = a.nm; k.A = A;
    dim3 grid((A + 255) / 256, a.B);
    if (a.y_dtype == BSY_F16)
        hipLaunchKernelGGL(decode_kernel<half_t>, grid, dim3(256), 0, s, k, (half_t*)a.y);
    else if (a.y_dtype == BSY_F32)
        hipLaunchKernelGGL(decode_kernel<float>, grid, dim3(256), 0, s, k, (float*)a.y);
    else
        BSY_FAIL(BSY_ERR_ARG, "decode: y dtype %d unsupported", a.y_dtype);
    HIP_TRY(hipGetLastError());
    return BSY_OK;
}

template <typename T>
__global__ __launch_bounds__(256) void raw_nchw_kernel(const float* __restrict__ box, int ldb,
                                                       const float* __restrict__ cls, int ldc, int hw, int nc,
                                                       T* __restrict__ out) {
    // out (B, 64+nc, h*w); thread = (pixel, channel) with pixels fastest so writes coalesce
    const int pix = blockIdx.x * blockDim.x + threadIdx.x;
    const int c = blockIdx.y;
    const int b = blockIdx.z;
    if (pix >= hw) return;
    const size_t ip = (size_t)b * hw + pix;
    const float v = c < 64 ? box[ip * ldb + c] : cls[ip * ldc + (c - 64)];
    out[((size_t)b * (64 + nc) + c) * hw + pix] = (T)v;
}

int launch_raw_nchw(const float* box, int ldb, const float* cls, int ldc, int B, int h, int w, int nc, void* out,
                    int out_dtype, hipStream_t s) {
    if (!box || !cls || !out) BSY_FAIL(BSY_ERR_ARG, "raw_nchw: null pointer");
    dim3 grid((h * w + 255) / 256, 64 + nc, B);
    if (out_dtype == BSY_F16)
        hipLaunchKernelGGL(raw_nchw_kernel<half_t>, grid, dim3(256), 0, s, box, ldb, cls, ldc, h * w, nc, (half_t*)out);
    else if (out_dtype == BSY_F32)
        hipLaunchKernelGGL(raw_nchw_kernel<float>, grid, dim3(256), 0, s, box, ldb, cls, ldc, h * w, nc, (float*)out);
    else
        BSY_FAIL(BSY_ERR_ARG, "raw_nchw: dtype %d unsupported", out_dtype);
    HIP_TRY(hipGetLastError());
    return BSY_OK;
}
