// First convolution of the graph: image BCHW (fp16 or fp32, 3 channels) -> conv 3x3 stride 2 + bias + SiLU -> NHWC fp16.
//
// Replaces model.0 Conv.forward_fuse (nn/modules/conv.py:149-151; layer 0 of cfg/models/11/yolo11-seg.yaml:17) and
// absorbs the NCHW->NHWC layout change.  K = 27 is too thin for MFMA and the layer is HBM-bound (reads 3*H*W, writes
// Cout*H*W/4 halves), so this is a direct fp32 VALU kernel: one thread per output pixel, weights read through the
// scalar path (wave-uniform addresses -> s_load + SGPR operand FMAs), 16 output channels per register pass, and each
// thread stores its pixel's channels as contiguous 16-byte vectors so a wave writes one contiguous span.
#include "common.h"

template <typename T>
__global__ __launch_bounds__(256) void conv_first_kernel(const T* __restrict__ img, const float* __restrict__ w,
                                                         const float* __restrict__ bias, half_t* __restrict__ dst,
                                                         int B, int H, int W, int OH, int OW, int stride, int pad,
                                                         int ldd, int Cout, int act) {
    const long long total = (long long)B * OH * OW;
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int ow = (int)(idx % OW);
    const long long t = idx / OW;
    const int oh = (int)(t % OH);
    const int n = (int)(t / OH);
    float x[27];
    const size_t plane = (size_t)H * W;
    const T* ip = img + (size_t)n * 3 * plane;
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
        const int iy = oh * stride - pad + kh;
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
            const int ix = ow * stride - pad + kw;
            const bool ok = (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
#pragma unroll
            for (int c = 0; c < 3; ++c)
                x[(kh * 3 + kw) * 3 + c] = ok ? (float)ip[c * plane + (size_t)iy * W + ix] : 0.f;
        }
    }
    half_t* dp = dst + (size_t)idx * ldd;
    for (int c0 = 0; c0 < Cout; c0 += 16) {
        float acc[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[j] = bias[c0 + j];
#pragma unroll
        for (int k = 0; k < 27; ++k) {
            const float xv = x[k];
#pragma unroll
            for (int j = 0; j < 16; ++j) acc[j] = fmaf(xv, w[k * Cout + c0 + j], acc[j]);
        }
        half8 o0, o1;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float a = act ? silu_f(acc[j]) : acc[j];
            const float b = act ? silu_f(acc[8 + j]) : acc[8 + j];
            o0[j] = (half_t)a;
            o1[j] = (half_t)b;
        }
        *reinterpret_cast<half8*>(dp + c0) = o0;
        *reinterpret_cast<half8*>(dp + c0 + 8) = o1;
    }
}

int launch_conv_first(const ConvFirstArgs& a, hipStream_t s) {
    if (a.ksize != 3) BSY_FAIL(BSY_ERR_ARG, "conv_first: ksize %d unsupported", a.ksize);
    if (a.Cout % 16 || a.ldd % 8 || ((uintptr_t)a.dst & 15)) BSY_FAIL(BSY_ERR_ARG, "conv_first: Cout %% 16 / alignment");
    if (a.OH != (a.H + 2 * a.pad - 3) / a.stride + 1 || a.OW != (a.W + 2 * a.pad - 3) / a.stride + 1)
        BSY_FAIL(BSY_ERR_ARG, "conv_first: output extent mismatch");
    const long long total = (long long)a.B * a.OH * a.OW;
    const unsigned grid = (unsigned)((total + 255) / 256);
    if (a.img_dtype == BSY_F16)
        hipLaunchKernelGGL(conv_first_kernel<half_t>, dim3(grid), dim3(256), 0, s, (const half_t*)a.img, a.w, a.b, a.dst,
                           a.B, a.H, a.W, a.OH, a.OW, a.stride, a.pad, a.ldd, a.Cout, a.act);
    else if (a.img_dtype == BSY_F32)
        hipLaunchKernelGGL(conv_first_kernel<float>, dim3(grid), dim3(256), 0, s, (const float*)a.img, a.w, a.b, a.dst,
                           a.B, a.H, a.W, a.OH, a.OW, a.stride, a.pad, a.ldd, a.Cout, a.act);
    else
        BSY_FAIL(BSY_ERR_ARG, "conv_first: image dtype %d unsupported", a.img_dtype);
    HIP_TRY(hipGetLastError());
    return BSY_OK;
}
