// Kernels for the modules only the BS-YOLO graph uses (cfg/models/11/yolo11.yaml of the fork):
//   dwconv_generic_kernel : depthwise kh x kw conv, stride 1 / 2, "same" padding, + bias (+SiLU on the first act_c
//                           channels: PMSFA's 7x7 runs on half of its input and passes the other half through)  -- PMSFA's 5x5 / 7x7
//                           (block.py:3035-3054), SCDown.cv2 (block.py:4503-4535), MSCAAttention's 5x5 and strip convs
//                           1x5 .. 21x1 (nn/Addmodules/MSCA.py:26-39)
//   copy_view_kernel      : materialise one operand of a Concat (optionally through nearest x2) into a channel slice
//   gap_kernel            : per (image, channel) mean over H x W                  (MSCA.py:69-72, ELA.py:50)
//   msca_mix_kernel       : softmax_i(sigmoid(l_i)) weighted sum of the four branch maps   (MSCA.py:74-82)
//   mul_kernel            : elementwise product of two maps                        (MSCA.py:86 `attn * u`)
//   ela_stats / ela_gate / ela_apply : ELA (nn/Addmodules/ELA.py:77-101)
// All NHWC f16 channel-slice views (8-channel = 16-byte pieces); reductions and gates in f32.  These layers are a few
// percent of the BS-YOLO forward; the kernels are written for clarity and coalescing, not tuned.
#include "common.h"

#if defined(__HIP_DEVICE_COMPILE__)
typedef __amdgpu_buffer_rsrc_t bo_rsrc_t;
__device__ __forceinline__ bo_rsrc_t bo_make_rsrc(const void* base, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), (short)0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ half8 bo_load16(bo_rsrc_t r, unsigned voff) {
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    union { u32x4 u; half8 h; } v;
    v.u = __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, 0, 0);
    return v.h;
}
#else
typedef int bo_rsrc_t;
__device__ __forceinline__ bo_rsrc_t bo_make_rsrc(const void*, unsigned) { return 0; }
__device__ __forceinline__ half8 bo_load16(bo_rsrc_t, unsigned) { return half8{0, 0, 0, 0, 0, 0, 0, 0}; }
#endif
#define BO_OOB 0xFFFFFFF0u

// ---------------------------------------------------------------------------------------------------------------------
// depthwise kh x kw, stride s, pad (kh/2, kw/2).  thread = (image, output pixel, 8-channel chunk); weights f32
// [kh*kw][wld] (wld = channels of the whole weight tensor: a launch may cover a channel sub-range of it).
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void dwconv_generic_kernel(const half_t* __restrict__ src, int lds_, int B, int H, int W,
                                                             int C, int OH, int OW, int kh, int kw, int stride,
                                                             const float* __restrict__ w, int wld,
                                                             const float* __restrict__ bias, half_t* __restrict__ dst,
                                                             int ldd, int act_c, unsigned span) {
    const int C8 = C >> 3;
    const long long total = (long long)B * OH * OW * C8;
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int c = (int)(idx % C8) * 8;
    long long t = idx / C8;
    const int ox = (int)(t % OW);
    t /= OW;
    const int oy = (int)(t % OH);
    const int n = (int)(t / OH);
    const bo_rsrc_t rs = bo_make_rsrc(src, span);
    float acc[8];
    {
        const f32x4 b0 = *reinterpret_cast<const f32x4*>(bias + c), b1 = *reinterpret_cast<const f32x4*>(bias + c + 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) { acc[j] = b0[j]; acc[4 + j] = b1[j]; }
    }
    const int iy0 = oy * stride - kh / 2, ix0 = ox * stride - kw / 2;
    for (int dy = 0; dy < kh; ++dy) {
        const int iy = iy0 + dy;
        const bool rowok = (unsigned)iy < (unsigned)H;
        for (int dx = 0; dx < kw; ++dx) {
            const int ix = ix0 + dx;
            const unsigned off = (rowok && (unsigned)ix < (unsigned)W)
                                     ? 2u * ((unsigned)((n * H + iy) * W + ix) * (unsigned)lds_ + (unsigned)c) : BO_OOB;
            const half8 v = bo_load16(rs, off);
            const float* wp = w + (size_t)(dy * kw + dx) * wld + c;
            const f32x4 w0 = *reinterpret_cast<const f32x4*>(wp), w1 = *reinterpret_cast<const f32x4*>(wp + 4);
            fma_mix8(acc, v, w0, w1);  // v_fma_mix_f32 per element (common.h)
        }
    }
    half8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = (half_t)(c + j < act_c ? silu_f(acc[j]) : acc[j]);  // SiLU on the first act_c channels
    *reinterpret_cast<half8*>(dst + ((size_t)(n * OH + oy) * OW + ox) * ldd + c) = o;
}

// Same conv, 4 consecutive outputs per thread along x with the input row segment they share held in registers: a row
// costs 3*S + KW loads and KW weight fetches instead of 4*KW of each.  KW / stride are compile-time so the window is a
// register array; instantiated for the kernel widths the BS-YOLO graph uses (1, 3 s2, 5, 7, 11, 21).
#define DWG_PX 4
template <int KW, int S>
__global__ __launch_bounds__(256) void dwconv_win_kernel(const half_t* __restrict__ src, int lds_, int B, int H, int W, int C,
                                                         int OH, int OW, int kh, const float* __restrict__ w, int wld,
                                                         const float* __restrict__ bias, half_t* __restrict__ dst, int ldd,
                                                         int act_c, unsigned span) {
    constexpr int NWIN = (DWG_PX - 1) * S + KW;
    const int C8 = C >> 3, OWG = (OW + DWG_PX - 1) / DWG_PX;
    const long long total = (long long)B * OH * OWG * C8;
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int c = (int)(idx % C8) * 8;
    long long t = idx / C8;
    const int ox0 = (int)(t % OWG) * DWG_PX;
    t /= OWG;
    const int oy = (int)(t % OH);
    const int n = (int)(t / OH);
    const bo_rsrc_t rs = bo_make_rsrc(src, span);
    float acc[DWG_PX][8];
    {
        const f32x4 b0 = *reinterpret_cast<const f32x4*>(bias + c), b1 = *reinterpret_cast<const f32x4*>(bias + c + 4);
#pragma unroll
        for (int p = 0; p < DWG_PX; ++p)
#pragma unroll
            for (int j = 0; j < 4; ++j) { acc[p][j] = b0[j]; acc[p][4 + j] = b1[j]; }
    }
    const int iy0 = oy * S - kh / 2, ix0 = ox0 * S - KW / 2;
    unsigned coloff[NWIN];
#pragma unroll
    for (int q = 0; q < NWIN; ++q) {
        const int ix = ix0 + q;
        coloff[q] = (unsigned)ix < (unsigned)W ? 2u * ((unsigned)ix * (unsigned)lds_ + (unsigned)c) : BO_OOB;
    }
    const unsigned rowbytes = 2u * (unsigned)W * (unsigned)lds_;
    for (int dy = 0; dy < kh; ++dy) {
        const int iy = iy0 + dy;
        const bool rowok = (unsigned)iy < (unsigned)H;
        const unsigned rb = (unsigned)(n * H + iy) * rowbytes;
        half8 win[NWIN];
#pragma unroll
        for (int q = 0; q < NWIN; ++q) win[q] = bo_load16(rs, (rowok && coloff[q] != BO_OOB) ? rb + coloff[q] : BO_OOB);
#pragma unroll
        for (int dx = 0; dx < KW; ++dx) {
            const float* wp = w + (size_t)(dy * KW + dx) * wld + c;
            const f32x4 w0 = *reinterpret_cast<const f32x4*>(wp), w1 = *reinterpret_cast<const f32x4*>(wp + 4);
#pragma unroll
            for (int p = 0; p < DWG_PX; ++p) fma_mix8(acc[p], win[p * S + dx], w0, w1);  // v_fma_mix_f32 per element (common.h)
        }
    }
#pragma unroll
    for (int p = 0; p < DWG_PX; ++p) {
        if (ox0 + p >= OW) break;
        half8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = (half_t)(c + j < act_c ? silu_f(acc[p][j]) : acc[p][j]);
        *reinterpret_cast<half8*>(dst + ((size_t)(n * OH + oy) * OW + ox0 + p) * ldd + c) = o;
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// LDS-tiled form for the square stride-1 kernels of PMSFA (5 x 5, 7 x 7; block.py:3035-3054) and MSCA's conv0: PMSFA runs them on
// 16 .. 64 channels, where a pixel's piece of the map is 32 .. 128 bytes and the window form above turns every load instruction
// into 32 cache lines (model.2's 5 x 5 on 160 x 160 x 16: 0.18 ms for 105 MB of traffic).  Here a workgroup owns an 8 x TW pixel
// tile of NCH 8-channel chunks (TW = 128 / NCH: 32 pixels x 32 channels .. 128 pixels x 8 channels), stages the (8 + K - 1) x
// (TW + K - 1) input patch with row-contiguous 16-byte loads (zero padding = the descriptor's out-of-range result) and the K * K
// weight rows of its channels into LDS, and every thread computes 4 consecutive pixels of one chunk from a register window of
// 4 + K - 1 LDS reads per kernel row.  Arithmetic = dwconv_win_kernel's, step for step (bias first, taps in (dy, dx) order as
// f32 FMAs on the f16 inputs): the same bits.
// ---------------------------------------------------------------------------------------------------------------------
// TH = tile rows: 8 (TW = 128 / NCH pixels wide), or 16 for a single chunk (16 x 64 instead of 8 x 128: a 160-wide map is 3 tiles of 64, not 2 of 128)
template <int K, int NCH, int DWT_TH>
__global__ __launch_bounds__(256) void dwconv_tile_kernel(const half_t* __restrict__ src, int lds_, int H, int W, int C,
                                                          const float* __restrict__ w, int wld, const float* __restrict__ bias,
                                                          half_t* __restrict__ dst, int ldd, int act_c, unsigned span, int tiles_x,
                                                          int tiles_y, int ncg, int ident_c0) {
    constexpr int TW = 1024 / (DWT_TH * NCH), PH = DWT_TH + K - 1, PW = TW + K - 1;
    constexpr int PXS = NCH * 8 + 8;            // halves per patch pixel: one 16-byte piece of padding spreads the window reads over the banks
    constexpr int NPIECE = PH * PW * NCH;
    constexpr int NWIN = 4 + K - 1;
    __shared__ __attribute__((aligned(16))) half_t sp[PH * PW * PXS];
    __shared__ __attribute__((aligned(16))) float sw[(K * K + 1) * NCH * 8];  // [tap][channel of this group], then the bias row
    const int tid = threadIdx.x;
    int t = blockIdx.x;
    const int cg = t % ncg; t /= ncg;
    const int tx = t % tiles_x; t /= tiles_x;
    const int ty = t % tiles_y;
    const int n = t / tiles_y;
    const int c0 = cg * NCH * 8;                     // first channel of this workgroup
    const int nch = min(NCH, (C - c0) >> 3);         // chunks that exist (the last channel group may be narrower)
    const int oy0 = ty * DWT_TH, ox0 = tx * TW;
    const bo_rsrc_t rs = bo_make_rsrc(src, span);
    for (int i = tid; i < (K * K + 1) * NCH * 8; i += 256) {
        const int row = i / (NCH * 8), c = i - row * (NCH * 8);
        sw[i] = c < nch * 8 ? (row < K * K ? w[(size_t)row * wld + c0 + c] : bias[c0 + c]) : 0.f;
    }
    for (int i = tid; i < NPIECE; i += 256) {
        const int ch = i % NCH, px = (i / NCH) % PW, py = i / (NCH * PW);
        const int y = oy0 - K / 2 + py, x = ox0 - K / 2 + px;
        const bool ok = ch < nch && (unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W;
        const half8 v = bo_load16(rs, ok ? 2u * ((unsigned)((n * H + y) * W + x) * (unsigned)lds_ + (unsigned)(c0 + ch * 8)) : BO_OOB);
        *reinterpret_cast<half8*>(sp + (py * PW + px) * PXS + ch * 8) = v;
    }
    if (ident_c0 > 0) {
        // pass-through half (identity kernel: bias 0, centre tap 1): the tap loop returns fma(x, 1, +0) = x + 0, every other tap adds a
        // signed zero to it -- written directly for this tile's pixels of the identity chunks ident_c0 / 8 + (this workgroup's chunks),
        // pixel-contiguous across the threads (the launch's tiles cover the conv half's channels only)
        for (int i = tid; i < DWT_TH * TW * NCH; i += 256) {
            const int ch = i % NCH, px = (i / NCH) % TW, py = i / (NCH * TW);
            const int y = oy0 + py, x = ox0 + px;
            if (ch >= nch || y >= H || x >= W) continue;
            const size_t pix = (size_t)(n * H + y) * W + x;
            const int ci = ident_c0 + c0 + ch * 8;
            const half8 v = *reinterpret_cast<const half8*>(src + pix * lds_ + ci);
            half8 o;
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = (half_t)((float)v[j] + 0.0f);
            *reinterpret_cast<half8*>(dst + pix * ldd + ci) = o;
        }
    }
    __syncthreads();
    // item = (tile row, group of 4 pixels, chunk): TH x (TW / 4) x NCH = 256 items, one per thread
    const int ch = tid % NCH, gx = (tid / NCH) % (TW / 4), gy = tid / (NCH * (TW / 4));
    if (ch >= nch) return;
    float acc[4][8];
    {
        const f32x4 b0 = *reinterpret_cast<const f32x4*>(sw + K * K * NCH * 8 + ch * 8), b1 = *reinterpret_cast<const f32x4*>(sw + K * K * NCH * 8 + ch * 8 + 4);
#pragma unroll
        for (int p = 0; p < 4; ++p)
#pragma unroll
            for (int j = 0; j < 4; ++j) { acc[p][j] = b0[j]; acc[p][4 + j] = b1[j]; }
    }
#pragma unroll 1
    for (int dy = 0; dy < K; ++dy) {
        const half_t* row = sp + ((gy + dy) * PW + 4 * gx) * PXS + ch * 8;
        half8 win[NWIN];
#pragma unroll
        for (int q = 0; q < NWIN; ++q) win[q] = *reinterpret_cast<const half8*>(row + q * PXS);
#pragma unroll
        for (int dx = 0; dx < K; ++dx) {
            const float* wp = sw + (dy * K + dx) * NCH * 8 + ch * 8;
            const f32x4 w0 = *reinterpret_cast<const f32x4*>(wp), w1 = *reinterpret_cast<const f32x4*>(wp + 4);
#pragma unroll
            for (int p = 0; p < 4; ++p) fma_mix8(acc[p], win[p + dx], w0, w1);
        }
    }
    const int oy = oy0 + gy, c = c0 + ch * 8;
    if (oy >= H) return;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int ox = ox0 + 4 * gx + p;
        if (ox >= W) break;
        half8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = (half_t)(c + j < act_c ? silu_f(acc[p][j]) : acc[p][j]);
        *reinterpret_cast<half8*>(dst + ((size_t)(n * H + oy) * W + ox) * ldd + c) = o;
    }
}

template <int K>
static void launch_dw_tile(const DwGenArgs& a, unsigned span, hipStream_t s) {
    // identity-kernel half (ident_c0 = C / 2): tiles cover the conv half's chunks; each thread also copies its pixels of the other half
    const bool ident = a.ident_c0 > 0 && 2 * a.ident_c0 == a.C && !(a.ident_c0 & 7) && a.act_c <= a.ident_c0;
    const int nchunks = (ident ? a.ident_c0 : a.C) / 8;
    const int NCH = nchunks >= 4 ? 4 : (nchunks >= 2 ? 2 : 1);
    const int TH = NCH == 1 ? 16 : 8, TW = 1024 / (TH * NCH);
    const int tiles_x = (a.W + TW - 1) / TW, tiles_y = (a.H + TH - 1) / TH, ncg = (nchunks + NCH - 1) / NCH;
    const unsigned grid = (unsigned)((long long)a.B * tiles_y * tiles_x * ncg);
#define DWT_GO(NCH_, TH_)                                                                                                        \
    hipLaunchKernelGGL((dwconv_tile_kernel<K, NCH_, TH_>), dim3(grid), dim3(256), 0, s, a.src, a.lds, a.H, a.W, ident ? a.ident_c0 : a.C, a.w, a.wld, a.b, \
                       a.dst, a.ldd, a.act_c, span, tiles_x, tiles_y, ncg, ident ? a.ident_c0 : 0)
    if (NCH == 4) DWT_GO(4, 8);
    else if (NCH == 2) DWT_GO(2, 8);
    else DWT_GO(1, 16);
#undef DWT_GO
}

int launch_dwconv_generic(const DwGenArgs& a, hipStream_t s) {
    if (!a.src || !a.dst || !a.w || !a.b) BSY_FAIL(BSY_ERR_ARG, "dwconv: null pointer");
    if ((a.C & 7) || (a.lds & 7) || (a.ldd & 7) || (a.wld & 3) || ((uintptr_t)a.src & 15) || ((uintptr_t)a.dst & 15) ||
        ((uintptr_t)a.w & 15) || ((uintptr_t)a.b & 15))
        BSY_FAIL(BSY_ERR_ARG, "dwconv: channels/strides must be multiples of 8 and pointers 16-byte aligned");
    if (a.kh < 1 || a.kw < 1 || a.kh > 31 || a.kw > 31 || !(a.kh & 1) || !(a.kw & 1) || (a.stride != 1 && a.stride != 2))
        BSY_FAIL(BSY_ERR_ARG, "dwconv: kernel %d x %d stride %d unsupported (odd sizes up to 31, stride 1 or 2)", a.kh, a.kw, a.stride);
    const int OH = (a.H + 2 * (a.kh / 2) - a.kh) / a.stride + 1, OW = (a.W + 2 * (a.kw / 2) - a.kw) / a.stride + 1;
    if (a.OH != OH || a.OW != OW) BSY_FAIL(BSY_ERR_ARG, "dwconv: output extent mismatch");
    const long long elems = (long long)a.B * a.H * a.W * a.lds;
    if (elems >= (1LL << 31)) BSY_FAIL(BSY_ERR_ARG, "dwconv: source view exceeds 2^31 elements (split the batch)");
    const unsigned span = (unsigned)((((long long)a.B * a.H * a.W - 1) * a.lds + a.C) * 2);
    const long long total = (long long)a.B * OH * OW * (a.C / 8);
    if (total <= 0) BSY_FAIL(BSY_ERR_ARG, "dwconv: empty");
    const long long totw = (long long)a.B * OH * ((OW + DWG_PX - 1) / DWG_PX) * (a.C / 8);
#define DWG_WIN(KW_, S_)                                                                                                     \
    hipLaunchKernelGGL((dwconv_win_kernel<KW_, S_>), dim3((unsigned)((totw + 255) / 256)), dim3(256), 0, s, a.src, a.lds, a.B, \
                       a.H, a.W, a.C, OH, OW, a.kh, a.w, a.wld, a.b, a.dst, a.ldd, a.act_c, span)
    // square 5 x 5 / 7 x 7 stride-1 kernels: the LDS-tiled form (tile grid small enough for a 32-bit block index)
    const bool tiled = a.kh == a.kw && (a.kw == 5 || a.kw == 7) && a.stride == 1 && !getenv("BSY_NO_DWTILE") &&
                       (long long)a.B * ((a.H + 7) / 8) * ((a.W + 31) / 32) * (a.C / 8) < 0x7fffffffLL;
    if (tiled && a.kw == 5) launch_dw_tile<5>(a, span, s);
    else if (tiled) launch_dw_tile<7>(a, span, s);
    else if (a.kw == 1 && a.stride == 1) DWG_WIN(1, 1);
    else if (a.kw == 3 && a.stride == 2) DWG_WIN(3, 2);
    else if (a.kw == 5 && a.stride == 1) DWG_WIN(5, 1);
    else if (a.kw == 7 && a.stride == 1) DWG_WIN(7, 1);
    else if (a.kw == 11 && a.stride == 1) DWG_WIN(11, 1);
    else if (a.kw == 21 && a.stride == 1) DWG_WIN(21, 1);
    else
        hipLaunchKernelGGL(dwconv_generic_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, a.src, a.lds, a.B, a.H,
                           a.W, a.C, OH, OW, a.kh, a.kw, a.stride, a.w, a.wld, a.b, a.dst, a.ldd, a.act_c, span);
#undef DWG_WIN
    HIP_TRY(hipGetLastError());
    return BSY_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// dst[n, y, x, 0:C] = src[n, y >> up, x >> up, 0:C]
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void copy_view_kernel(const half_t* __restrict__ src, int lds_, int up, int B, int H,
                                                        int W, int C, half_t* __restrict__ dst, int ldd) {
    const int C8 = C >> 3;
    const long long total = (long long)B * H * W * C8;
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int c = (int)(idx % C8) * 8;
    long long t = idx / C8;
    const int x = (int)(t % W);
    t /= W;
    const int y = (int)(t % H);
    const int n = (int)(t / H);
    const int SH = H >> up, SW = W >> up;
    const half8 v = *reinterpret_cast<const half8*>(src + ((size_t)(n * SH + (y >> up)) * SW + (x >> up)) * lds_ + c);
    *reinterpret_cast<half8*>(dst + ((size_t)(n * H + y) * W + x) * ldd + c) = v;
}

int launch_copy_view(const half_t* src, int lds_, int up, int B, int H, int W, int C, half_t* dst, int ldd, hipStream_t s) {
    if (!src || !dst || (C & 7) || (lds_ & 7) || (ldd & 7) || ((uintptr_t)src & 15) || ((uintptr_t)dst & 15) ||
        (up && ((H | W) & 1)) || B <= 0 || H <= 0 || W <= 0)
        BSY_FAIL(BSY_ERR_ARG, "copy_view: bad layout");
    const long long total = (long long)B * H * W * (C / 8);
    hipLaunchKernelGGL(copy_view_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, src, lds_, up, B, H, W, C, dst, ldd);
    HIP_TRY(hipGetLastError());
    return BSY_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// Global average pool: out[n, c] = mean over H*W (f16 vector (B, ldo), f32 accumulation).  One workgroup per
// (image, 8-channel chunk): 256 threads stride the pixels, LDS tree.
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gap_kernel(const half_t* __restrict__ src, int lds_, int HW, half_t* __restrict__ out, int ldo) {
    __shared__ float red[256][8];
    const int c = blockIdx.x * 8, n = blockIdx.y, tid = threadIdx.x;
    float a[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int p = tid; p < HW; p += 256) {
        const half8 v = *reinterpret_cast<const half8*>(src + ((size_t)n * HW + p) * lds_ + c);
#pragma unroll
        for (int j = 0; j < 8; ++j) a[j] += (float)v[j];
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) red[tid][j] = a[j];
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if (tid < st)
#pragma unroll
            for (int j = 0; j < 8; ++j) red[tid][j] += red[tid + st][j];
        __syncthreads();
    }
    if (tid < 8) out[(size_t)n * ldo + c + tid] = (half_t)(red[0][tid] / (float)HW);
}

int launch_gap(const half_t* src, int lds_, int B, int H, int W, int C, half_t* out, int ldo, hipStream_t s) {
    if (!src || !out || (C & 7) || (lds_ & 7) || ((uintptr_t)src & 15) || B <= 0 || H <= 0 || W <= 0 || ldo < C)
        BSY_FAIL(BSY_ERR_ARG, "gap: bad layout");
    hipLaunchKernelGGL(gap_kernel, dim3(C / 8, B), dim3(256), 0, s, src, lds_, H * W, out, ldo);
    HIP_TRY(hipGetLastError());
    return BSY_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// MSCAAttention's spatial part in one launch (nn/Addmodules/MSCA.py:53-75): attn = conv0(x) (5 x 5), the four strip-conv
// branches attn_i = conv{i}_2(conv{i}_1(attn)) (1 x k then k x 1, k = 5 / 7 / 11 / 21; `dilconv` folded into conv{i}_2 at
// pack time) and gap(attn_i).  Everything is depthwise, so a workgroup owns (image, 8 channels) and keeps that H x W
// slab in LDS: x, attn and the row-conv intermediate never reach HBM and the nine conv launches + four pooling launches
// of the unfused plan (0.39 ms of BS-YOLO11s' 4.8 ms forward at 20 x 20 x 512, B = 64) become one that reads the map
// once and writes the four branch maps.  Arithmetic is that of dwconv_win_kernel / gap_kernel step for step (bias first,
// taps in (dy, dx) order as f32 FMAs, f16 rounding of every intermediate map, the same strided partial sums and LDS
// tree for the mean), so the results are bit-identical to the unfused plan.  H * W <= MSCA_SP_MAXPIX (covers the 40 x 40
// level of a 1280-pixel input; larger maps keep the separate launches).
// ---------------------------------------------------------------------------------------------------------------------
struct MscaSpK {
    const half_t* src;
    int lds, HW, H, W, C;
    const float* w[9];  // conv0, conv0_1, conv0_2, conv1_1, conv1_2, conv2_1, conv2_2, conv3_1, conv3_2: [taps][C]
    const float* b[9];
    half_t* br[4];
    int ldb[4];
    half_t* gap[4];
    int ldg[4];
};

// attn and the row-conv intermediate live in LDS as f32 copies of their f16-rounded values, so the 88 strip taps per
// element cost one 8-byte LDS read + one packed FMA per channel PAIR and no conversions.  A thread owns one channel pair
// (tid & 3) of the pixels slot, slot + 64, .. (slot = tid >> 2): the taps' weights of its pair stay in registers across
// the pixel loop (2 K floats; with all 8 channels per thread the 21-tap strips spill), and 1600 items over 256 threads
// leave 11 % of the lanes idle instead of 22 %.
typedef float f32x2 __attribute__((ext_vector_type(2)));
// two scalar v_fma_f32 (bit-identical to the packed v_pk_fma_f32 the vector builtin emits): since round 2 NO kernel of the
// library carries packed f32 arithmetic (tests/test_host_logic.py::test_isa_has_no_packed_f32 scans the shipped objects) --
// the engine cannot promise that this kernel never shares CUs with another engine's or a user stream's MFMA kernels
__device__ __forceinline__ f32x2 msca_fma2(f32x2 a, f32x2 b, f32x2 c) { return f32x2{fmaf(a[0], b[0], c[0]), fmaf(a[1], b[1], c[1])}; }
typedef _Float16 half2_t __attribute__((ext_vector_type(2)));

__device__ __forceinline__ f32x2 msca_round2(f32x2 a, half2_t& o) {  // f16 rounding of a map element, kept as f32
    // The f32 sums pass through an empty asm before the conversion: with scalar FMAs the compiler otherwise folds the last
    // FMA of a chain and the conversion into v_fma_mixlo/hi_f16, which rounds the exact sum ONCE -- the separate launches this
    // kernel must equal bit for bit round twice (f32, then f16), and 3e-4 of the elements differed by one f16 ulp.
    float a0 = a[0], a1 = a[1];
    asm volatile("" : "+v"(a0), "+v"(a1));
    const half_t o0 = (half_t)a0, o1 = (half_t)a1;  // converted back from two separate registers (no SDWA half select)
    o[0] = o0; o[1] = o1;
    return f32x2{(float)o0, (float)o1};
}

// Tap t of conv number ci (0 = conv0, 1 + 2 i / 2 + 2 i = row / column conv of branch i) sits at row MSCA_TAP0[ci] + t of
// the LDS weight table sw[122][8] (f32; rows 113 .. 121 = the nine biases): staged once per workgroup, then every phase
// pulls the 2 K floats of its channel pair into registers with immediate-offset LDS reads.  (Read from global memory
// the compiler keeps one 64-bit address per tap alive for all nine convs: 380 VGPRs.)
#define MSCA_NTAP 113
__device__ __constant__ const int MSCA_TAP0[9] = {0, 25, 30, 35, 42, 49, 60, 71, 92};

// (Round-2 alternatives of the one-output-per-thread form, same results, all slower than its 0.23 ms at 20 x 20 x 512, B = 64:
// zero-bordered slabs without bounds tests -- 72 KiB of LDS, two workgroups per CU instead of three: 0.25 ms; 512 threads with two
// pixels in flight per thread: 0.29 ms; eight channels per thread on f16 slabs: 0.26 ms.)
#define MSCA_SP_MAXPIX 1890  // (16 + 32 + 32) B x H*W + 8 KiB of reduction scratch + 4 KiB of weights <= 160 KiB of LDS

// One strip-conv branch: st = rows(sa) (1 x K), branch = cols(st) (K x 1) -> HBM and, as f16, into the dead x slab (the mean
// reads it back).  Round 3: a thread computes MSCA_RUN consecutive outputs along the conv's axis from a register window of
// MSCA_RUN + K - 1 inputs -- 6 LDS reads and 6 bounds tests per output of the 21-tap strips instead of 21 of each (the one-output
// form took 0.24 ms at 20 x 20 x 512, B = 64, 6 x its FMA issue time).  Each output still sums bias first, then its taps in
// ascending order, one FMA per tap (taps outside the map multiply a zero, as before): the same bits.
#define MSCA_RUN 7
template <int K, int T0>
__device__ __forceinline__ void msca_branch(const MscaSpK& p, int bi, const f32x2* sa, f32x2* st, const f32x2* sw, half2_t* sx, int c2,
                                            int n, int tid) {
    constexpr int R = K / 2, NWIN = MSCA_RUN + K - 1;
    const int H = p.H, W = p.W, HW = p.HW, pr = tid & 3;
    {   // rows: task = (y, run of MSCA_RUN columns); consecutive threads = consecutive runs of a row
        f32x2 wr[K];
#pragma unroll
        for (int t = 0; t < K; ++t) wr[t] = sw[(T0 + t) * 4 + pr];
        const f32x2 br = sw[(MSCA_NTAP + 1 + 2 * bi) * 4 + pr];
        const int runs = (W + MSCA_RUN - 1) / MSCA_RUN, ntask = H * runs;
        for (int t = tid >> 2; t < ntask; t += 64) {
            const int y = t / runs, x0 = (t - y * runs) * MSCA_RUN;
            const f32x2* row = sa + (size_t)(y * W) * 4 + pr;
            f32x2 win[NWIN];
#pragma unroll
            for (int jx = 0; jx < NWIN; ++jx) {
                const int ix = x0 + jx - R;
                win[jx] = (unsigned)ix < (unsigned)W ? row[ix * 4] : f32x2{0.f, 0.f};
            }
#pragma unroll
            for (int o = 0; o < MSCA_RUN; ++o) {
                f32x2 acc = br;
#pragma unroll
                for (int dx = 0; dx < K; ++dx) acc = msca_fma2(win[o + dx], wr[dx], acc);
                half2_t h;
                const f32x2 r = msca_round2(acc, h);
                if (x0 + o < W) st[(size_t)(y * W + x0 + o) * 4 + pr] = r;
            }
        }
    }
    __syncthreads();
    {   // columns: task = (run of MSCA_RUN rows, x); consecutive threads = consecutive columns
        f32x2 wc[K];
#pragma unroll
        for (int t = 0; t < K; ++t) wc[t] = sw[(T0 + K + t) * 4 + pr];
        const f32x2 bc = sw[(MSCA_NTAP + 2 + 2 * bi) * 4 + pr];
        const int runs = (H + MSCA_RUN - 1) / MSCA_RUN, ntask = runs * W;
        for (int t = tid >> 2; t < ntask; t += 64) {
            const int yr = t / W, x = t - yr * W, y0 = yr * MSCA_RUN;
            const f32x2* col = st + (size_t)x * 4 + pr;
            f32x2 win[NWIN];
#pragma unroll
            for (int jy = 0; jy < NWIN; ++jy) {
                const int iy = y0 + jy - R;
                win[jy] = (unsigned)iy < (unsigned)H ? col[(size_t)iy * W * 4] : f32x2{0.f, 0.f};
            }
#pragma unroll
            for (int o = 0; o < MSCA_RUN; ++o) {
                f32x2 acc = bc;
#pragma unroll
                for (int dy = 0; dy < K; ++dy) acc = msca_fma2(win[o + dy], wc[dy], acc);
                half2_t h;
                msca_round2(acc, h);
                if (y0 + o < H) {
                    const int q = (y0 + o) * W + x;
                    sx[(size_t)q * 4 + pr] = h;
                    *reinterpret_cast<half2_t*>(p.br[bi] + ((size_t)n * HW + q) * p.ldb[bi] + c2) = h;
                }
            }
        }
    }
}

__global__ __launch_bounds__(256) void msca_spatial_kernel(const MscaSpK p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char msca_smem[];
    f32x2* sa = reinterpret_cast<f32x2*>(msca_smem);                  // attn  [HW][4 pairs]
    f32x2* st = sa + (size_t)p.HW * 4;                                 // row-conv output
    half2_t* sx = reinterpret_cast<half2_t*>(st + (size_t)p.HW * 4);  // x [HW][4 pairs]; after conv0: the current branch map (f16)
    float(*red)[8] = reinterpret_cast<float(*)[8]>(sx + (size_t)p.HW * 4);
    f32x2* sw = reinterpret_cast<f32x2*>(red + 256);                   // weight table [122][4 pairs]
    const int c = blockIdx.x * 8, n = blockIdx.y, tid = threadIdx.x;
    const int pr = tid & 3, slot = tid >> 2, c2 = c + 2 * pr;
    const int H = p.H, W = p.W, HW = p.HW, C = p.C;

    for (int q = tid; q < HW; q += 256)
        *reinterpret_cast<half8*>(sx + (size_t)q * 4) = *reinterpret_cast<const half8*>(p.src + ((size_t)n * HW + q) * p.lds + c);
    for (int e = tid; e < (MSCA_NTAP + 9) * 2; e += 256) {  // 16-byte pieces of the weight table
        const int row = e >> 1, h4 = (e & 1) * 4;
        const float* g;
        if (row >= MSCA_NTAP) {
            g = p.b[row - MSCA_NTAP] + c + h4;
        } else {
            int ci = 0;
#pragma unroll
            for (int k = 1; k < 9; ++k) ci += row >= MSCA_TAP0[k];
            g = p.w[ci] + (size_t)(row - MSCA_TAP0[ci]) * C + c + h4;
        }
        *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(sw) + row * 8 + h4) = *reinterpret_cast<const f32x4*>(g);
    }
    __syncthreads();
    {   // attn = conv0(x): 5 x 5; task = (y, run of MSCA_RUN columns), a 5 x (MSCA_RUN + 4) register window of f16 pairs
        f32x2 w0[25];
#pragma unroll
        for (int t = 0; t < 25; ++t) w0[t] = sw[t * 4 + pr];
        const f32x2 b0 = sw[MSCA_NTAP * 4 + pr];
        const int runs = (W + MSCA_RUN - 1) / MSCA_RUN, ntask = H * runs;
        for (int t = slot; t < ntask; t += 64) {
            const int y = t / runs, x0 = (t - y * runs) * MSCA_RUN;
            f32x2 acc[MSCA_RUN];
#pragma unroll
            for (int o = 0; o < MSCA_RUN; ++o) acc[o] = b0;
#pragma unroll
            for (int dy = 0; dy < 5; ++dy) {
                const int iy = y + dy - 2;
                const bool rowin = (unsigned)iy < (unsigned)H;
                unsigned win[MSCA_RUN + 4];
#pragma unroll
                for (int jx = 0; jx < MSCA_RUN + 4; ++jx) {
                    const int ix = x0 + jx - 2;
                    union { half2_t h; unsigned u; } hv;
                    hv.u = 0u;
                    if (rowin && (unsigned)ix < (unsigned)W) hv.h = sx[(size_t)(iy * W + ix) * 4 + pr];
                    win[jx] = hv.u;
                }
#pragma unroll
                for (int o = 0; o < MSCA_RUN; ++o)
#pragma unroll
                    for (int dx = 0; dx < 5; ++dx) {
                        acc[o][0] = fma_mix_lo(win[o + dx], w0[dy * 5 + dx][0], acc[o][0]);  // no convert + packed FMA here (common.h)
                        acc[o][1] = fma_mix_hi(win[o + dx], w0[dy * 5 + dx][1], acc[o][1]);
                    }
            }
#pragma unroll
            for (int o = 0; o < MSCA_RUN; ++o) {
                half2_t h;
                const f32x2 r = msca_round2(acc[o], h);
                if (x0 + o < W) sa[(size_t)(y * W + x0 + o) * 4 + pr] = r;
            }
        }
    }
    __syncthreads();  // attn complete; x is dead: its slab takes the branch maps
    for (int i = 0; i < 4; ++i) {
        if (i == 0) msca_branch<5, 25>(p, 0, sa, st, sw, sx, c2, n, tid);
        else if (i == 1) msca_branch<7, 35>(p, 1, sa, st, sw, sx, c2, n, tid);
        else if (i == 2) msca_branch<11, 49>(p, 2, sa, st, sw, sx, c2, n, tid);
        else msca_branch<21, 71>(p, 3, sa, st, sw, sx, c2, n, tid);
        __syncthreads();  // branch map complete in sx; every thread is done reading st
        // gap_kernel's 256 strided partial sums (pixels q with q % 256 == slot + 64 m, ascending), then its pairwise tree
        // red[t] += red[t + s], s = 128 .. 1, with the same operands in the same order but without its eight barriers: partial sums
        // slot + 64 m of one thread meet in registers (s = 128: m with m + 2; s = 64: 0 with 1), s = 32 and 16 pair slots of different
        // waves (one LDS exchange, read by wave 0), s = 8 .. 1 pair lanes of wave 0 (lane = 4 slot + pair).
        f32x2 gs[4] = {f32x2{0.f, 0.f}, f32x2{0.f, 0.f}, f32x2{0.f, 0.f}, f32x2{0.f, 0.f}};
        int m = 0;
        for (int q = slot; q < HW; q += 64, m = (m + 1) & 3) {
            const half2_t h = sx[(size_t)q * 4 + pr];
#pragma unroll
            for (int mm = 0; mm < 4; ++mm)  // (m is not a compile-time index)
                if (mm == m) { gs[mm][0] += (float)h[0]; gs[mm][1] += (float)h[1]; }
        }
        float v0 = (gs[0][0] + gs[2][0]) + (gs[1][0] + gs[3][0]), v1 = (gs[0][1] + gs[2][1]) + (gs[1][1] + gs[3][1]);
        red[slot][2 * pr] = v0;
        red[slot][2 * pr + 1] = v1;
        __syncthreads();
        if (tid < 64) {  // slots 0 .. 15
            v0 = (red[slot][2 * pr] + red[slot + 32][2 * pr]) + (red[slot + 16][2 * pr] + red[slot + 48][2 * pr]);
            v1 = (red[slot][2 * pr + 1] + red[slot + 32][2 * pr + 1]) + (red[slot + 16][2 * pr + 1] + red[slot + 48][2 * pr + 1]);
#pragma unroll
            for (int d = 32; d >= 4; d >>= 1) {
                v0 += __shfl_down(v0, d, 64);
                v1 += __shfl_down(v1, d, 64);
            }
            if (tid < 4) {
                p.gap[i][(size_t)n * p.ldg[i] + c2] = (half_t)(v0 / (float)HW);
                p.gap[i][(size_t)n * p.ldg[i] + c2 + 1] = (half_t)(v1 / (float)HW);
            }
        }
        __syncthreads();  // red and sx are free for the next branch
    }
}

bool msca_spatial_supported(int H, int W) { return H > 0 && W > 0 && H * W <= MSCA_SP_MAXPIX; }

int launch_msca_spatial(const MscaSpArgs& a, hipStream_t s) {
    if (!msca_spatial_supported(a.H, a.W)) BSY_FAIL(BSY_ERR_ARG, "msca_spatial: %d x %d map does not fit LDS (H*W <= %d)", a.H, a.W, MSCA_SP_MAXPIX);
    if (!a.src || (a.C & 7) || (a.lds & 7) || ((uintptr_t)a.src & 15) || a.B <= 0) BSY_FAIL(BSY_ERR_ARG, "msca_spatial: bad source layout");
    MscaSpK k;
    k.src = a.src; k.lds = a.lds; k.H = a.H; k.W = a.W; k.HW = a.H * a.W; k.C = a.C;
    for (int i = 0; i < 9; ++i) {
        if (!a.w[i] || !a.b[i] || ((uintptr_t)a.w[i] & 15) || ((uintptr_t)a.b[i] & 15)) BSY_FAIL(BSY_ERR_ARG, "msca_spatial: weight %d missing / misaligned", i);
        k.w[i] = a.w[i]; k.b[i] = a.b[i];
    }
    for (int i = 0; i < 4; ++i) {
        if (!a.br[i] || !a.gap[i] || (a.ldb[i] & 7) || ((uintptr_t)a.br[i] & 15) || a.ldg[i] < a.C) BSY_FAIL(BSY_ERR_ARG, "msca_spatial: bad output %d", i);
        k.br[i] = a.br[i]; k.ldb[i] = a.ldb[i]; k.gap[i] = a.gap[i]; k.ldg[i] = a.ldg[i];
    }
    const size_t smem = (size_t)k.HW * 80 + 256 * 8 * 4 + (MSCA_NTAP + 9) * 32;
    static bool attr_set = false;
    if (!attr_set) {
        HIP_TRY(hipFuncSetAttribute((const void*)msca_spatial_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, MSCA_SP_MAXPIX * 80 + 8192 + (MSCA_NTAP + 9) * 32));
        attr_set = true;
    }
    hipLaunchKernelGGL(msca_spatial_kernel, dim3(a.C / 8, a.B), dim3(256), smem, s, k);
    HIP_TRY(hipGetLastError());
    return BSY_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// MSCA branch mix: weight_i[n, c] = softmax over i of sigmoid(logit_i[n, c]); out = sum_i weight_i * branch_i
// ---------------------------------------------------------------------------------------------------------------------
struct MixK {
    const half_t* br[4];
    int ldb[4];
    const float* lg[4];
    int ldl[4];
};
// grid (pixel blocks, B): a thread keeps ONE 8-channel chunk of one image and walks pixels, so the branch weights (4 x 8 sigmoids,
// exponentials and quotients) are computed once per thread instead of once per pixel (round 3: 52 -> see docs/experiments.md; the
// per-pixel form spent half its time on them).  Same expressions, same order: the same bits.
#define MIX_PIX 8  // pixels per thread
__global__ __launch_bounds__(256) void msca_mix_kernel(const MixK k, int B, int HW, int C, half_t* __restrict__ dst, int ldd) {
    const int C8 = C >> 3, n = blockIdx.y;
    const int per = 256 / C8 > 0 ? 256 / C8 : 1;            // pixels in flight per pass of the workgroup (C8 <= 256)
    const int c8 = threadIdx.x % C8, slot = threadIdx.x / C8;
    if (slot >= per) return;
    const int c = c8 * 8;
    float w[4][8], den[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) den[j] = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float sg = 1.0f / (1.0f + __expf(-k.lg[i][(size_t)n * k.ldl[i] + c + j]));
            w[i][j] = __expf(sg);  // sigmoid output is in (0, 1): no max subtraction needed
            den[j] += w[i][j];
        }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) w[i][j] = w[i][j] / den[j];
    const int q0 = blockIdx.x * per * MIX_PIX + slot;
#pragma unroll 2
    for (int t = 0; t < MIX_PIX; ++t) {
        const int q = q0 + t * per;
        if (q >= HW) break;
        const size_t pix = (size_t)n * HW + q;
        float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const half8 v = *reinterpret_cast<const half8*>(k.br[i] + pix * k.ldb[i] + c);
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] = fmaf(w[i][j], (float)v[j], acc[j]);
        }
        half8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = (half_t)acc[j];
        *reinterpret_cast<half8*>(dst + pix * ldd + c) = o;
    }
}

int launch_msca_mix(const MixArgs& a, hipStream_t s) {
    MixK k;
    for (int i = 0; i < 4; ++i) {
        if (!a.br[i] || !a.lg[i] || (a.ldb[i] & 7) || ((uintptr_t)a.br[i] & 15)) BSY_FAIL(BSY_ERR_ARG, "msca_mix: branch %d bad layout", i);
        k.br[i] = a.br[i]; k.ldb[i] = a.ldb[i]; k.lg[i] = a.lg[i]; k.ldl[i] = a.ldl[i];
    }
    if (!a.dst || (a.C & 7) || (a.ldd & 7) || ((uintptr_t)a.dst & 15) || a.B <= 0 || a.HW <= 0 || a.C > 2048) BSY_FAIL(BSY_ERR_ARG, "msca_mix: bad layout");
    const int C8 = a.C / 8, per = 256 / C8 > 0 ? 256 / C8 : 1;
    const int nblk = (a.HW + per * MIX_PIX - 1) / (per * MIX_PIX);
    hipLaunchKernelGGL(msca_mix_kernel, dim3((unsigned)nblk, (unsigned)a.B), dim3(256), 0, s, k, a.B, a.HW, a.C, a.dst, a.ldd);
    HIP_TRY(hipGetLastError());
    return BSY_OK;
}

__global__ __launch_bounds__(256) void mul_kernel(const half_t* __restrict__ a, int lda, const half_t* __restrict__ b, int ldb,
                                                  long long npix, int C, half_t* __restrict__ dst, int ldd) {
    const int C8 = C >> 3;
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= npix * C8) return;
    const int c = (int)(idx % C8) * 8;
    const long long pix = idx / C8;
    const half8 x = *reinterpret_cast<const half8*>(a + (size_t)pix * lda + c), y = *reinterpret_cast<const half8*>(b + (size_t)pix * ldb + c);
    half8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = (half_t)((float)x[j] * (float)y[j]);
    *reinterpret_cast<half8*>(dst + (size_t)pix * ldd + c) = o;
}

int launch_mul(const half_t* a, int lda, const half_t* b, int ldb, long long npix, int C, half_t* dst, int ldd, hipStream_t s) {
    if (!a || !b || !dst || (C & 7) || (lda & 7) || (ldb & 7) || (ldd & 7) || (((uintptr_t)a | (uintptr_t)b | (uintptr_t)dst) & 15) || npix <= 0)
        BSY_FAIL(BSY_ERR_ARG, "mul: bad layout");
    const long long total = npix * (C / 8);
    hipLaunchKernelGGL(mul_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, a, lda, b, ldb, npix, C, dst, ldd);
    HIP_TRY(hipGetLastError());
    return BSY_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// ELA.  scratch (f32) per image: [rowmean H*C][colmean W*C][gmean C][hgate H*C][wgate W*C][cgate C]
//   ela_stats : row means (over W) and column means (over H).  One workgroup per (row | column, image).
//   ela_gate  : v = dilated (2) depthwise conv1d of the means (k taps, zero pad k-1) -> GroupNorm over (16 channels x L)
//               -> sigmoid; plus the channel gate sigmoid(w_centre * global mean).  One workgroup per (image, group, direction).
//   ela_apply : out = x * (a * cgate[c] + b * hgate[y, c] * wgate[x, c]) + r * x
// ---------------------------------------------------------------------------------------------------------------------
// grid (max(H, W), B, 2): z = 0 -> workgroup = one row (mean over W), z = 1 -> one column (mean over H).  Threads cover
// (position along the line, 8-channel chunk): `nslot` positions in flight per chunk, then a serial fold over the slots.
__global__ __launch_bounds__(256) void ela_stats_kernel(const half_t* __restrict__ src, int lds_, int H, int W, int C,
                                                        float* __restrict__ scratch, size_t per_img) {
    const int line = blockIdx.x, n = blockIdx.y, dir = blockIdx.z, tid = threadIdx.x;
    const int L = dir ? W : H, R = dir ? H : W;  // lines of this direction, positions per line
    if (line >= L) return;
    const int C8 = C >> 3;
    const int nslot = 256 / C8 > 0 ? 256 / C8 : 1;
    float* out = scratch + (size_t)n * per_img + (dir ? (size_t)H * C : 0) + (size_t)line * C;
    const half_t* ip = src + (size_t)n * H * W * lds_;
    const size_t lstep = dir ? (size_t)W * lds_ : (size_t)lds_;             // step between positions of a line
    const size_t lbase = dir ? (size_t)line * lds_ : (size_t)line * W * lds_;
    __shared__ float part[256][8];
    {
        const int c8 = tid % C8, slot = tid / C8;  // threads past nslot * C8 (256 not a multiple of C8) contribute zeros
        float a[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        if (slot < nslot)
            for (int r = slot; r < R; r += nslot) {
                const half8 v = *reinterpret_cast<const half8*>(ip + lbase + (size_t)r * lstep + c8 * 8);
#pragma unroll
                for (int j = 0; j < 8; ++j) a[j] += (float)v[j];
            }
#pragma unroll
        for (int j = 0; j < 8; ++j) part[tid][j] = a[j];
    }
    __syncthreads();
    if (tid < C8) {
        float a[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int sl = 0; sl < nslot; ++sl)
#pragma unroll
            for (int j = 0; j < 8; ++j) a[j] += part[sl * C8 + tid][j];
#pragma unroll
        for (int j = 0; j < 8; ++j) out[tid * 8 + j] = a[j] / (float)R;
    }
}

// grid (groups, B, 2): z = 0 rows (L = H), 1 columns (L = W).  LDS: conv output of the group [L][16].
__global__ __launch_bounds__(256) void ela_gate_kernel(float* __restrict__ scratch, size_t per_img, int H, int W, int C, int k,
                                                       int gsz, const float* __restrict__ wsp, const float* __restrict__ wch,
                                                       const float* __restrict__ gnw, const float* __restrict__ gnb) {
    extern __shared__ float sv[];  // [L][gsz] + reduction scratch [256][2]
    const int g = blockIdx.x, n = blockIdx.y, dir = blockIdx.z, tid = threadIdx.x;
    const int L = dir ? W : H;
    float* base = scratch + (size_t)n * per_img;
    const float* mean = dir ? base + (size_t)H * C : base;
    float* gate = base + (size_t)(H + W + 1) * C + (dir ? (size_t)H * C : 0);
    const int c0 = g * gsz;
    float* red = sv + (size_t)L * gsz;
    float s1 = 0.f, s2 = 0.f;
    for (int i = tid; i < L * gsz; i += 256) {
        const int l = i / gsz, cc = i - l * gsz, c = c0 + cc;
        float v = 0.f;
        for (int t = 0; t < k; ++t) {  // padding (k-1)*2/2 = k-1, dilation 2: tap t reads position l - (k-1) + 2t
            const int p = l - (k - 1) + 2 * t;
            if ((unsigned)p < (unsigned)L) v = fmaf(wsp[(size_t)c * k + t], mean[(size_t)p * C + c], v);
        }
        sv[i] = v;
        s1 += v;
        s2 += v * v;
    }
    red[2 * tid] = s1;
    red[2 * tid + 1] = s2;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if (tid < st) { red[2 * tid] += red[2 * (tid + st)]; red[2 * tid + 1] += red[2 * (tid + st) + 1]; }
        __syncthreads();
    }
    const float cnt = (float)(L * gsz);
    const float mu = red[0] / cnt;
    const float var = fmaxf(red[1] / cnt - mu * mu, 0.f);  // biased variance (F.group_norm)
    const float rstd = rsqrtf(var + 1e-5f);
    for (int i = tid; i < L * gsz; i += 256) {
        const int l = i / gsz, c = c0 + (i - l * gsz);
        const float z = (sv[i] - mu) * rstd * gnw[c] + gnb[c];
        gate[(size_t)l * C + c] = 1.0f / (1.0f + __expf(-z));
    }
    if (dir == 0) {  // channel gate: the Conv1d sees a length-1 sequence -> only its centre tap contributes
        // global mean = mean of the row means (rows have equal length).  Round 3: the sum over the rows is dealt to 256 / gsz slots per
        // channel and folded in slot order (fixed: the same bits on every run) -- one thread per channel walking H dependent loads
        // made this the longest path of the launch (14 us per ELA at 80 x 80)
        __syncthreads();  // sv / red are free again
        const int nsl = 256 / gsz > 0 ? 256 / gsz : 1, cc = tid % gsz, sl = tid / gsz;
        float part = 0.f;
        if (sl < nsl)
            for (int l = sl; l < H; l += nsl) part += base[(size_t)l * C + c0 + cc];
        if (sl < nsl) sv[sl * gsz + cc] = part;
        __syncthreads();
        if (tid < gsz) {
            const int c = c0 + tid;
            float gmv = 0.f;
            for (int q = 0; q < nsl; ++q) gmv += sv[q * gsz + tid];
            gmv /= (float)H;
            base[(size_t)(2 * (H + W) + 1) * C + c] = 1.0f / (1.0f + __expf(-(wch[(size_t)c * k + (k - 1) / 2] * gmv)));
        }
    }
}

__global__ __launch_bounds__(256) void ela_apply_kernel(const half_t* __restrict__ src, int lds_, int B, int H, int W, int C,
                                                        const float* __restrict__ scratch, size_t per_img, float ca, float sb,
                                                        float rr, half_t* __restrict__ dst, int ldd) {
    const int C8 = C >> 3;
    const long long total = (long long)B * H * W * C8;
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int c = (int)(idx % C8) * 8;
    long long t = idx / C8;
    const int x = (int)(t % W);
    t /= W;
    const int y = (int)(t % H);
    const int n = (int)(t / H);
    const float* base = scratch + (size_t)n * per_img + (size_t)(H + W + 1) * C;
    const float* hg = base + (size_t)y * C + c;
    const float* wg = base + (size_t)H * C + (size_t)x * C + c;
    const float* cg = base + (size_t)(H + W) * C + c;
    const size_t pix = (size_t)(n * H + y) * W + x;
    const half8 v = *reinterpret_cast<const half8*>(src + pix * lds_ + c);
    half8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float xv = (float)v[j];
        o[j] = (half_t)(xv * (ca * cg[j] + sb * (hg[j] * wg[j])) + rr * xv);
    }
    *reinterpret_cast<half8*>(dst + pix * ldd + c) = o;
}

size_t ela_scratch_floats(int H, int W, int C) { return (size_t)(2 * (H + W) + 2) * C; }

// means in the scratch -> gates in the scratch (f32 only: shared by the fp16 path and the fp32 correctness mode, ref32.hip)
// GroupNorm(max(1, C // 16), C) (ELA.py:70): channels per group -- 16 when C is a multiple of 16, otherwise C / (C // 16)
// (C = 40: 2 groups of 20; C = 24: one group of 24); 0 when the group count does not divide C (the reference's constructor raises)
static int ela_group_size(int C) {
    const int groups = C >= 16 ? C / 16 : 1;
    return C % groups ? 0 : C / groups;
}

int launch_ela_gate(const ElaArgs& a, hipStream_t s) {
    const int gsz = ela_group_size(a.C);
    if (gsz <= 0 || gsz > 256) BSY_FAIL(BSY_ERR_ARG, "ela: %d channels do not split into %d GroupNorm groups", a.C, a.C / 16);
    const int Lmax = a.H > a.W ? a.H : a.W;
    const size_t lds = ((size_t)Lmax * gsz + 512) * sizeof(float);
    if (lds > 64 * 1024) BSY_FAIL(BSY_ERR_ARG, "ela: map side %d too long for the gate kernel", Lmax);
    hipLaunchKernelGGL(ela_gate_kernel, dim3(a.C / gsz, a.B, 2), dim3(256), lds, s, a.scratch, ela_scratch_floats(a.H, a.W, a.C), a.H, a.W,
                       a.C, a.k, gsz, a.wsp, a.wch, a.gnw, a.gnb);
    HIP_TRY(hipGetLastError());
    return BSY_OK;
}

int launch_ela(const ElaArgs& a, hipStream_t s) {
    if (!a.src || !a.dst || !a.scratch || !a.wsp || !a.wch || !a.gnw || !a.gnb) BSY_FAIL(BSY_ERR_ARG, "ela: null pointer");
    if ((a.C & 7) || (a.lds & 7) || (a.ldd & 7) || (((uintptr_t)a.src | (uintptr_t)a.dst) & 15) || a.k < 1 || !(a.k & 1) || a.k > 15 ||
        a.B <= 0 || a.H <= 0 || a.W <= 0)
        BSY_FAIL(BSY_ERR_ARG, "ela: bad layout (C must be a multiple of 8, odd kernel <= 15)");
    const int gsz = ela_group_size(a.C);
    if (gsz <= 0) BSY_FAIL(BSY_ERR_ARG, "ela: %d channels do not split into %d GroupNorm groups", a.C, a.C / 16);
    const size_t per_img = ela_scratch_floats(a.H, a.W, a.C);
    const int Lmax = a.H > a.W ? a.H : a.W;
    const size_t lds = ((size_t)Lmax * gsz + 512) * sizeof(float);
    if (lds > 64 * 1024) BSY_FAIL(BSY_ERR_ARG, "ela: map side %d too long for the gate kernel", Lmax);
    if (a.C / 8 > 256) BSY_FAIL(BSY_ERR_ARG, "ela: more than 2048 channels");
    hipLaunchKernelGGL(ela_stats_kernel, dim3(Lmax, a.B, 2), dim3(256), 0, s, a.src, a.lds, a.H, a.W, a.C, a.scratch, per_img);
    {
        const int rc = launch_ela_gate(a, s);
        if (rc != BSY_OK) return rc;
    }
    const long long total = (long long)a.B * a.H * a.W * (a.C / 8);
    hipLaunchKernelGGL(ela_apply_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, a.src, a.lds, a.B, a.H, a.W, a.C,
                       a.scratch, per_img, a.ch_coef, a.sp_coef, a.res_coef, a.dst, a.ldd);
    HIP_TRY(hipGetLastError());
    return BSY_OK;
}
