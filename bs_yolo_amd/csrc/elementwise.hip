// HBM-bound helpers of the forward path, NHWC fp16, one 16-byte (8-channel) vector per thread.
//   dwconv3x3_kernel : DWConv (nn/modules/conv.py:224-229) 3x3 s1 p1 + bias (+SiLU) (+residual)
//                      used by Detect.cv3 (head.py:46-53) and Attention.pe (block.py:4264)
//   sppf_pool_kernel : the three chained MaxPool2d(5,1,2) of SPPF.forward (block.py:3145-3149); chained 5x5 max
//                      pools with -inf padding equal 5x5 / 9x9 / 13x13 window maxima, computed in one pass and
//                      written into the channel slices of the (virtual) concat buffer
#include "common.h"

__device__ __forceinline__ half8 hmax8(half8 a, half8 b) {
    half8 r;
#pragma unroll
    for (int j = 0; j < 8; ++j) r[j] = a[j] > b[j] ? a[j] : b[j];
    return r;
}

#define DW_PX 2   // output pixels per thread along x
#define DW_RY 8   // output rows per thread: a rolling 3-row register window -> 2 vector loads per output instead of 9,
                  // and the 9 x 8 weights are fetched once per 16 outputs
// Loads go through a buffer descriptor: a lane whose pixel lies outside the image gets voffset = out-of-range and
// receives zeros -- the conv's zero padding without a branch per load (the flat-load version spent more instructions
// on exec-mask juggling and 64-bit addresses than on the 9 FMAs per output).
#if defined(__HIP_DEVICE_COMPILE__)
typedef __amdgpu_buffer_rsrc_t dw_rsrc_t;
__device__ __forceinline__ dw_rsrc_t dw_make_rsrc(const void* base, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), (short)0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ half8 dw_load16(dw_rsrc_t r, unsigned voff) {
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    union { u32x4 u; half8 h; } v;
    v.u = __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, 0, 0);
    return v.h;
}
#else
typedef int dw_rsrc_t;
__device__ __forceinline__ dw_rsrc_t dw_make_rsrc(const void*, unsigned) { return 0; }
__device__ __forceinline__ half8 dw_load16(dw_rsrc_t, unsigned) { return half8{0, 0, 0, 0, 0, 0, 0, 0}; }
#endif
#define DW_OOB 0xFFFFFFF0u

// a[j] = fma(f32(v[j]), w[j], a[j]) for the 8 channels of a piece, one v_fma_mix_f32 each (same order per element as before:
// top, mid, bot of column kw = 0, 1, 2)
__device__ __forceinline__ void dw_fma8(float (&a)[8], const half8& v, const float (&w)[8]) {
    a[0] = fma_mix_e<0>(v, w[0], a[0]); a[1] = fma_mix_e<1>(v, w[1], a[1]); a[2] = fma_mix_e<2>(v, w[2], a[2]); a[3] = fma_mix_e<3>(v, w[3], a[3]);
    a[4] = fma_mix_e<4>(v, w[4], a[4]); a[5] = fma_mix_e<5>(v, w[5], a[5]); a[6] = fma_mix_e<6>(v, w[6], a[6]); a[7] = fma_mix_e<7>(v, w[7], a[7]);
}

template <bool ACT, bool RES>
__global__ __launch_bounds__(256) void dwconv3x3_kernel(const half_t* __restrict__ src, int lds_, int B, int H, int W,
                                                        int C, const float* __restrict__ w,
                                                        const float* __restrict__ bias, half_t* __restrict__ dst,
                                                        int ldd, const half_t* __restrict__ res, int ldr, unsigned span) {
    const int C8 = C >> 3;
    const int WG = (W + DW_PX - 1) / DW_PX, HG = (H + DW_RY - 1) / DW_RY;
    const long long total = (long long)B * HG * WG * C8;
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int c = (int)(idx % C8) * 8;  // channel chunk fastest: a pixel's channels are contiguous -> coalesced
    long long t = idx / C8;
    const int x0 = (int)(t % WG) * DW_PX;
    t /= WG;
    const int y0 = (int)(t % HG) * DW_RY;
    const int n = (int)(t / HG);
    float wk[9][8], bs[8];
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        const f32x4 w0 = *reinterpret_cast<const f32x4*>(w + k * C + c), w1 = *reinterpret_cast<const f32x4*>(w + k * C + c + 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) { wk[k][j] = w0[j]; wk[k][4 + j] = w1[j]; }
    }
    {
        const f32x4 b0 = *reinterpret_cast<const f32x4*>(bias + c), b1 = *reinterpret_cast<const f32x4*>(bias + c + 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) { bs[j] = b0[j]; bs[4 + j] = b1[j]; }
    }
    const dw_rsrc_t rs = dw_make_rsrc(src, span);
    // byte offsets of the 4 window columns inside a row (or out-of-range), row part added per row
    unsigned coloff[DW_PX + 2];
#pragma unroll
    for (int q = 0; q < DW_PX + 2; ++q) {
        const int ix = x0 + q - 1;
        coloff[q] = (unsigned)ix < (unsigned)W ? 2u * ((unsigned)ix * (unsigned)lds_ + (unsigned)c) : DW_OOB;
    }
    const unsigned rowbytes = 2u * (unsigned)W * (unsigned)lds_;
    auto load_row = [&](int iy, half8 (&r)[DW_PX + 2]) {
        const bool rowok = (unsigned)iy < (unsigned)H;
        const unsigned rb = (unsigned)(n * H + iy) * rowbytes;
#pragma unroll
        for (int q = 0; q < DW_PX + 2; ++q)
            r[q] = dw_load16(rs, (rowok && coloff[q] != DW_OOB) ? rb + coloff[q] : DW_OOB);
    };
    half8 top[DW_PX + 2], mid[DW_PX + 2], bot[DW_PX + 2];
    load_row(y0 - 1, top);
    load_row(y0, mid);
#pragma unroll 1
    for (int dy = 0; dy < DW_RY; ++dy) {
        const int y = y0 + dy;
        if (y >= H) break;
        load_row(y + 1, bot);
        float acc[DW_PX][8];
#pragma unroll
        for (int p = 0; p < DW_PX; ++p) {
            // v_fma_mix_f32 per element (common.h: the convert + packed-FMA form is not safe beside MFMA kernels)
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[p][j] = bs[j];
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                dw_fma8(acc[p], top[p + kw], wk[kw]);
                dw_fma8(acc[p], mid[p + kw], wk[3 + kw]);
                dw_fma8(acc[p], bot[p + kw], wk[6 + kw]);
            }
        }
#pragma unroll
        for (int p = 0; p < DW_PX; ++p) {
            const int x = x0 + p;
            const size_t pix = (size_t)(n * H + y) * W + (x < W ? x : W - 1);
            half8 o;
            if (RES) {
                const half8 r = *reinterpret_cast<const half8*>(res + pix * ldr + c);
#pragma unroll
                for (int j = 0; j < 8; ++j) o[j] = (half_t)((ACT ? silu_f(acc[p][j]) : acc[p][j]) + (float)r[j]);
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) o[j] = (half_t)(ACT ? silu_f(acc[p][j]) : acc[p][j]);
            }
            if (x < W) *reinterpret_cast<half8*>(dst + pix * ldd + c) = o;
        }
#pragma unroll
        for (int q = 0; q < DW_PX + 2; ++q) { top[q] = mid[q]; mid[q] = bot[q]; }
    }
}

int launch_dwconv(const DwArgs& a, hipStream_t s) {
    if ((a.C & 7) || (a.lds & 7) || (a.ldd & 7) || (a.res && (a.ldr & 7)) || ((uintptr_t)a.src & 15) ||
        ((uintptr_t)a.dst & 15) || ((uintptr_t)a.res & 15) || ((uintptr_t)a.w & 15) || ((uintptr_t)a.b & 15))
        BSY_FAIL(BSY_ERR_ARG, "dwconv: channels/strides must be multiples of 8 and pointers 16-byte aligned");
    const long long total = (long long)a.B * ((a.H + DW_RY - 1) / DW_RY) * ((a.W + DW_PX - 1) / DW_PX) * (a.C / 8);
    if (total <= 0) BSY_FAIL(BSY_ERR_ARG, "dwconv: empty");
    const long long elems = (long long)a.B * a.H * a.W * a.lds;
    if (elems >= (1LL << 31)) BSY_FAIL(BSY_ERR_ARG, "dwconv: source view exceeds 2^31 elements (split the batch)");
    const unsigned span = (unsigned)((((long long)a.B * a.H * a.W - 1) * a.lds + a.C) * 2);  // bytes from the view's first element
    const dim3 grid((unsigned)((total + 255) / 256));
#define DW_LAUNCH(ACT_, RES_)                                                                                              \
    hipLaunchKernelGGL((dwconv3x3_kernel<ACT_, RES_>), grid, dim3(256), 0, s, a.src, a.lds, a.B, a.H, a.W, a.C, a.w, a.b, \
                       a.dst, a.ldd, a.res, a.ldr, span)
    if (a.act && a.res) DW_LAUNCH(true, true);
    else if (a.act) DW_LAUNCH(true, false);
    else if (a.res) DW_LAUNCH(false, true);
    else DW_LAUNCH(false, false);
#undef DW_LAUNCH
    HIP_TRY(hipGetLastError());
    return BSY_OK;
}


__global__ __launch_bounds__(256) void sppf_pool_kernel(half_t* __restrict__ buf, int ld, int B, int H, int W, int C) {
    const int C8 = C >> 3;
    const long long total = (long long)B * H * W * C8;
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int c = (int)(idx % C8) * 8;
    const long long pix = idx / C8;
    const int x = (int)(pix % W);
    const long long t = pix / W;
    const int y = (int)(t % H);
    const int n = (int)(t / H);
    const half_t NEG = (half_t)(-65504.0f);
    half8 m5, m9, m13;
#pragma unroll
    for (int j = 0; j < 8; ++j) m5[j] = m9[j] = m13[j] = NEG;
    for (int dy = -6; dy <= 6; ++dy) {
        const int iy = y + dy;
        if ((unsigned)iy >= (unsigned)H) continue;
        const int ady = dy < 0 ? -dy : dy;
        for (int dx = -6; dx <= 6; ++dx) {
            const int ix = x + dx;
            if ((unsigned)ix >= (unsigned)W) continue;
            const int adx = dx < 0 ? -dx : dx;
            const int r = ady > adx ? ady : adx;
            const half8 v = *reinterpret_cast<const half8*>(buf + ((size_t)(n * H + iy) * W + ix) * ld + c);
            m13 = hmax8(m13, v);
            if (r <= 4) m9 = hmax8(m9, v);
            if (r <= 2) m5 = hmax8(m5, v);
        }
    }
    half_t* op = buf + (size_t)pix * ld + c;
    *reinterpret_cast<half8*>(op + C) = m5;
    *reinterpret_cast<half8*>(op + 2 * C) = m9;
    *reinterpret_cast<half8*>(op + 3 * C) = m13;
}

// LDS version: one workgroup owns the whole H x W map of one image and CG adjacent 8-channel chunks (CG = as many as LDS takes, up
// to 8: a pixel's piece is then 16 CG contiguous bytes -- with one chunk per workgroup every load and store was an isolated 16-byte
// piece 2 ld bytes away from the next, and the 20 x 20 x 256 pool of YOLO11s took 45 us for 105 MB); each MaxPool2d(5,1,2) is a row
// pass + a column pass (10 LDS reads instead of 25 global ones), chained three times without leaving the CU.  Item = (pixel, chunk),
// chunk fastest.  Maxima are exact: the result does not depend on CG.
__global__ __launch_bounds__(256) void sppf_pool_lds_kernel(half_t* __restrict__ buf, int ld, int H, int W, int C, int CG) {
    extern __shared__ __attribute__((aligned(16))) half8 sp[];
    const int HW = H * W, NI = HW * CG;
    half8* A = sp;        // current map  [pixel][chunk]
    half8* R = sp + NI;   // row-pooled map
    const int c = blockIdx.x * CG * 8;
    const int n = blockIdx.y;
    half_t* base = buf + (size_t)n * HW * ld + c;
    for (int it = threadIdx.x; it < NI; it += 256) {
        const int i = it / CG, ck = it - i * CG;
        A[it] = *reinterpret_cast<const half8*>(base + (size_t)i * ld + ck * 8);
    }
    __syncthreads();
    for (int pass = 1; pass <= 3; ++pass) {
        for (int it = threadIdx.x; it < NI; it += 256) {
            const int i = it / CG, y = i / W, x = i - y * W;
            half8 m = A[it];
#pragma unroll
            for (int d = 1; d <= 2; ++d) {
                if (x - d >= 0) m = hmax8(m, A[it - d * CG]);
                if (x + d < W) m = hmax8(m, A[it + d * CG]);
            }
            R[it] = m;
        }
        __syncthreads();
        for (int it = threadIdx.x; it < NI; it += 256) {
            const int i = it / CG, ck = it - i * CG, y = i / W;
            half8 m = R[it];
#pragma unroll
            for (int d = 1; d <= 2; ++d) {
                if (y - d >= 0) m = hmax8(m, R[it - d * W * CG]);
                if (y + d < H) m = hmax8(m, R[it + d * W * CG]);
            }
            A[it] = m;
            *reinterpret_cast<half8*>(base + (size_t)i * ld + pass * C + ck * 8) = m;
        }
        lds_barrier();  // orders the LDS maps only: this pass's global stores stay in flight under the next pass (__syncthreads() drained them)
    }
}

int launch_sppf_pool(half_t* buf, int ld, int B, int H, int W, int C, hipStream_t s) {
    if ((C & 7) || (ld & 7) || ld < 4 * C || ((uintptr_t)buf & 15)) BSY_FAIL(BSY_ERR_ARG, "sppf_pool: bad layout");
    const long long total = (long long)B * H * W * (C / 8);
    if (total <= 0) BSY_FAIL(BSY_ERR_ARG, "sppf_pool: empty");
    if ((size_t)H * W * 32 <= 64 * 1024) {  // both LDS maps fit: the usual 20x20 .. 40x40 P5 maps
        // chunks per workgroup: the largest divisor of C / 8 (at most 8) whose two maps fit 64 KiB and that still leaves two
        // workgroups per CU (a workgroup walks its map serially: at 8 images, 4 chunks each took 28 us against 11 us for 1)
        int cg = 1;
        for (int k = 8; k > 1; --k)
            if ((C / 8) % k == 0 && (size_t)H * W * 32 * k <= 64 * 1024 && (long long)(C / 8 / k) * B >= 512) { cg = k; break; }
        hipLaunchKernelGGL(sppf_pool_lds_kernel, dim3(C / 8 / cg, B), dim3(256), (size_t)H * W * 32 * cg, s, buf, ld, H, W, C, cg);
        HIP_TRY(hipGetLastError());
        return BSY_OK;
    }
    hipLaunchKernelGGL(sppf_pool_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, buf, ld, B, H, W, C);
    HIP_TRY(hipGetLastError());
    return BSY_OK;
}


// ---------------------------------------------------------------------------------------------------------------------
// Space-to-depth of the input image for YOLOv5u's 6x6 stride-2 pad-2 stem (cfg/models/v5/yolov5.yaml:16): BCHW f16 / f32
// -> NHWC f16 (B, H/2, W/2, ld >= 16), channel (dy * 2 + dx) * 3 + c = image[c][2Y + dy][2X + dx], channels 12 .. 15 zero.
// The stem then is an ordinary 3x3 stride-1 pad-1 conv over 16 channels (weights re-laid-out by weights.py "first_s2d").
// HBM-bound: reads 3 H W elements, writes 8 H W bytes.  One thread per output pixel.
// ---------------------------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void s2d_kernel(const T* __restrict__ img, int B, int H, int W, half_t* __restrict__ dst, int ldd) {
    const int OW = W >> 1, OH = H >> 1;
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long long)B * OH * OW) return;
    const int X = (int)(idx % OW);
    const long long t = idx / OW;
    const int Y = (int)(t % OH), n = (int)(t / OH);
    half8 lo, hi;
#pragma unroll
    for (int j = 0; j < 8; ++j) { lo[j] = (half_t)0.f; hi[j] = (half_t)0.f; }
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const half_t v = (half_t)(float)img[((size_t)(n * 3 + c) * H + 2 * Y + (q >> 1)) * W + 2 * X + (q & 1)];
            const int k = q * 3 + c;
            if (k < 8) lo[k] = v; else hi[k - 8] = v;
        }
    half_t* o = dst + (size_t)idx * ldd;
    *reinterpret_cast<half8*>(o) = lo;
    *reinterpret_cast<half8*>(o + 8) = hi;
}

int launch_s2d(const void* img, int img_dtype, int B, int H, int W, half_t* dst, int ldd, hipStream_t s) {
    if (!img || !dst || B <= 0 || H <= 0 || W <= 0 || (H & 1) || (W & 1) || ldd < 16 || (ldd & 7) || ((uintptr_t)dst & 15))
        BSY_FAIL(BSY_ERR_ARG, "s2d: bad argument (even H, W; dst row stride >= 16, 16-byte aligned)");
    const long long total = (long long)B * (H / 2) * (W / 2);
    const unsigned nb = (unsigned)((total + 255) / 256);
    if (img_dtype == BSY_F16) hipLaunchKernelGGL(s2d_kernel<half_t>, dim3(nb), dim3(256), 0, s, (const half_t*)img, B, H, W, dst, ldd);
    else if (img_dtype == BSY_F32) hipLaunchKernelGGL(s2d_kernel<float>, dim3(nb), dim3(256), 0, s, (const float*)img, B, H, W, dst, ldd);
    else BSY_FAIL(BSY_ERR_ARG, "s2d: image dtype %d unsupported", img_dtype);
    HIP_TRY(hipGetLastError());
    return BSY_OK;
}
