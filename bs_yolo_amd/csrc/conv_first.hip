// First convolution of the graph: image BCHW (fp16 or fp32, 3 channels) -> conv 3x3 stride 2 pad 1 + bias + SiLU -> NHWC fp16.
//
// Replaces model.0 Conv.forward_fuse (nn/modules/conv.py:149-151; layer 0 of cfg/models/11/yolo11-seg.yaml:17) and
// absorbs the NCHW->NHWC layout change.  HBM-bound (reads 3*H*W, writes Cout*H*W/4 elements per image), so the job is
// to touch every input byte ~once, coalesced, and keep the 27-deep contraction off the VALU:
//   * a workgroup owns a 4 x 64 output tile; its 9 x 132 x 3 input patch is read in aligned groups of 4 pixels per
//     plane and parked in LDS pixel-interleaved with a zero 4th channel (image_conv.h);
//   * the contraction runs on v_mfma_f32_32x32x16_f16 with K = 9 taps x 4 = 36 padded to 48 -- weights packed like every
//     other conv ([CoutPad][Kpad], k = (kh, kw, c)) but with the zero 4th input channel, held in registers as the A
//     operand; a lane's B fragment is two taps = two 8-byte LDS reads at compile-time offsets;
//   * epilogue as conv_mfma.hip: the tile goes through LDS so every store is a coalesced 16-byte piece.
#include "image_conv.h"

#define CF_TH 4
#define CF_TW 64
#define CF_PR (2 * CF_TH + 1)   // 9 patch rows
#define CF_NG 33                // 4-pixel groups per patch row: image columns 128*tx - 4 .. 128*tx + 127
#define CF_ROWPX (4 * CF_NG)    // 132 patch pixels per row (8 bytes each)

template <typename T, int NT>
__global__ __launch_bounds__(256) void conv_first_mfma_kernel(const T* __restrict__ img, const half_t* __restrict__ wgt,
                                                              const float* __restrict__ bias, half_t* __restrict__ dst,
                                                              int H, int W, int OH, int OW, int ldd, int Cout, int act,
                                                              int tiles_x, int tiles_y) {
    constexpr int OT = CF_TH * CF_TW * (32 * NT + 8);  // fp16 output tile of the coalesced epilogue
    constexpr int PT = CF_PR * CF_ROWPX * 4;
    constexpr int SM = PT > OT ? PT : OT;
    __shared__ __attribute__((aligned(16))) half_t stile[SM];
    half_t* patch = stile;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lrow = lane & 31, lh = lane >> 5;
    int bid = blockIdx.x;
    const int tx = bid % tiles_x;
    bid /= tiles_x;
    const int ty = bid % tiles_y;
    const int n = bid / tiles_y;
    const int iy0 = ty * CF_TH * 2 - 1, ix0 = tx * CF_TW * 2 - 4;
    const T* ip = img + (size_t)n * 3 * H * W;
    const bool vec = !(W & 3) && !((uintptr_t)img & (4 * sizeof(T) - 1));
    for (int idx = tid; idx < CF_PR * CF_NG; idx += 256) {
        const int r = idx / CF_NG, j = idx - r * CF_NG;
        const ImgItem<T> it = vec ? img_item_load<T>(ip, H, W, iy0 + r, ix0 + 4 * j, true)
                                  : img_item_load_slow<T>(ip, H, W, iy0 + r, ix0 + 4 * j, true);
        img_item_park<T>(it, patch + (r * CF_ROWPX + 4 * j) * 4);
    }
    // weights: A operand, rows = output channels, [CoutPad][64] with k = (kh, kw, c4)
    half8 afr[NT][IMGC_KSUB];
#pragma unroll
    for (int a = 0; a < NT; ++a)
#pragma unroll
        for (int s = 0; s < IMGC_KSUB; ++s)
            afr[a][s] = *reinterpret_cast<const half8*>(wgt + (size_t)(a * 32 + lrow) * IMGC_KROW + 16 * s + 8 * lh);
    __syncthreads();

    f32x16 acc[NT][2];  // start at the bias of their couts (common.h acc_bias; the packed bias is padded to 128 values)
#pragma unroll
    for (int a = 0; a < NT; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc_bias(acc[a][b], bias + a * 32, lh);
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        // output pixel (row `wave`, column x = 32 b + lrow): its window starts at patch pixel (2 wave, 2 x + 3)
        const int win = ((2 * wave) * CF_ROWPX + 2 * (b * 32 + lrow) + 3) * 4;
#pragma unroll
        for (int s = 0; s < IMGC_KSUB; ++s) {
            const half8 bf = img_frag<CF_ROWPX>(patch, win, s, lh);
#pragma unroll
            for (int a = 0; a < NT; ++a) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(afr[a][s], bf, acc[a][b], 0, 0, 0);
        }
    }
    // epilogue: bias + SiLU -> fp16 tile [4 rows][64 px][32*NT ch] in LDS (the patch is dead), then coalesced 16-byte
    // stores: consecutive lanes write consecutive channels of one pixel, consecutive pixels are adjacent in NHWC
    __syncthreads();
    constexpr int CT = 32 * NT;       // channels in the tile
    constexpr int LDT = CT + 8;       // padded row (halves)
    half_t* tile = stile;
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        const int prow = wave * CF_TW + b * 32 + lrow;
#pragma unroll
        for (int a = 0; a < NT; ++a)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int c = a * 32 + 8 * g + 4 * lh;
                half4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float t = acc[a][b][4 * g + e];
                    o[e] = (half_t)(act ? silu_f(t) : t);
                }
                *reinterpret_cast<half4*>(tile + prow * LDT + c) = o;
            }
    }
    __syncthreads();
    constexpr int CPRW = CT / 8;
    for (int id = tid; id < CF_TH * CF_TW * CPRW; id += 256) {
        const int prow = id / CPRW, cc = (id % CPRW) * 8;
        const int oy = ty * CF_TH + prow / CF_TW, ox = tx * CF_TW + prow % CF_TW;
        if (oy >= OH || ox >= OW || cc >= Cout) continue;
        *reinterpret_cast<half8*>(dst + ((size_t)(n * OH + oy) * OW + ox) * ldd + cc) =
            *reinterpret_cast<const half8*>(tile + prow * LDT + cc);
    }
}

template <typename T>
static int launch_t(const ConvFirstArgs& a, hipStream_t s) {
    const int tiles_x = ceil_div(a.OW, CF_TW), tiles_y = ceil_div(a.OH, CF_TH);
    const long long nblk = (long long)a.B * tiles_x * tiles_y;
    if (nblk > 0x7fffffffLL) BSY_FAIL(BSY_ERR_ARG, "conv_first: grid too large");
    dim3 grid((unsigned)nblk);
    const T* img = (const T*)a.img;
    const half_t* w = (const half_t*)a.w;
    if (a.Cout <= 32)
        hipLaunchKernelGGL((conv_first_mfma_kernel<T, 1>), grid, dim3(256), 0, s, img, w, a.b, a.dst, a.H, a.W, a.OH, a.OW,
                           a.ldd, a.Cout, a.act, tiles_x, tiles_y);
    else if (a.Cout <= 64)
        hipLaunchKernelGGL((conv_first_mfma_kernel<T, 2>), grid, dim3(256), 0, s, img, w, a.b, a.dst, a.H, a.W, a.OH, a.OW,
                           a.ldd, a.Cout, a.act, tiles_x, tiles_y);
    else if (a.Cout <= 96)
        hipLaunchKernelGGL((conv_first_mfma_kernel<T, 3>), grid, dim3(256), 0, s, img, w, a.b, a.dst, a.H, a.W, a.OH, a.OW,
                           a.ldd, a.Cout, a.act, tiles_x, tiles_y);
    else
        hipLaunchKernelGGL((conv_first_mfma_kernel<T, 4>), grid, dim3(256), 0, s, img, w, a.b, a.dst, a.H, a.W, a.OH, a.OW,
                           a.ldd, a.Cout, a.act, tiles_x, tiles_y);
    HIP_TRY(hipGetLastError());
    return BSY_OK;
}

int launch_conv_first(const ConvFirstArgs& a, hipStream_t s) {
    if (a.ksize != 3 || a.stride != 2 || a.pad != 1) BSY_FAIL(BSY_ERR_ARG, "conv_first: only 3x3 stride 2 pad 1 (got k=%d s=%d p=%d)", a.ksize, a.stride, a.pad);
    if (a.Cout % 8 || a.Cout > 128 || a.ldd % 8 || ((uintptr_t)a.dst & 15) || ((uintptr_t)a.w & 15) || ((uintptr_t)a.b & 15))
        BSY_FAIL(BSY_ERR_ARG, "conv_first: Cout must be a multiple of 8 and <= 128; 16-byte aligned pointers");
    if (a.OH != (a.H + 2 - 3) / 2 + 1 || a.OW != (a.W + 2 - 3) / 2 + 1) BSY_FAIL(BSY_ERR_ARG, "conv_first: output extent mismatch");
    if (a.img_dtype == BSY_F16) return launch_t<half_t>(a, s);
    if (a.img_dtype == BSY_F32) return launch_t<float>(a, s);
    BSY_FAIL(BSY_ERR_ARG, "conv_first: image dtype %d unsupported", a.img_dtype);
}
