// First convolution of the graph: image BCHW (fp16 or fp32, 3 channels) -> conv 3x3 stride 2 pad 1 + bias + SiLU -> NHWC fp16.
//
// Replaces model.0 Conv.forward_fuse (nn/modules/conv.py:149-151; layer 0 of cfg/models/11/yolo11-seg.yaml:17) and
// absorbs the NCHW->NHWC layout change.  HBM-bound (reads 3*H*W, writes Cout*H*W/4 elements per image), so the job is
// to touch every input byte ~once, coalesced, and keep the 27-deep contraction off the VALU:
//   * a workgroup owns a 4 x 64 output tile; its 9 x 129 x 3 input patch is read plane by plane with 16-byte vector
//     loads (the 128 image-aligned columns of every row) and parked in LDS as fp16;
//   * the contraction runs on v_mfma_f32_32x32x16_f16 with K = 27 padded to 32 -- the weights use the SAME packed layout
//     as every other conv ([CoutPad][Kpad], k = (kh, kw, c)), held in registers as the A operand; the B operand
//     (pixel on the lane) is gathered from the LDS patch: lane stride 2 pixels = 1 dword -> conflict-free 16-bit reads;
//   * epilogue as conv_mfma.hip: the tile goes through LDS so every store is a coalesced 16-byte piece.
#include "common.h"

#define CF_TH 4
#define CF_TW 64
#define CF_PR (2 * CF_TH + 1)   // 9 patch rows
#define CF_PC (2 * CF_TW + 1)   // 129 patch cols
#define CF_ROW 136              // row = [7 unused][1 halo col][128 cols]: the 128 image-aligned columns start 16-B aligned
#define CF_C0 7                 // LDS column of patch column 0
#define CF_PLANE (CF_PR * CF_ROW)
#define CF_ZERO (3 * CF_PLANE)  // one zero element for the K padding

template <typename T, int NT>
__global__ __launch_bounds__(256) void conv_first_mfma_kernel(const T* __restrict__ img, const half_t* __restrict__ wgt,
                                                              const float* __restrict__ bias, half_t* __restrict__ dst,
                                                              int H, int W, int OH, int OW, int ldd, int Cout, int act,
                                                              int tiles_x, int tiles_y) {
    constexpr int OT = CF_TH * CF_TW * (32 * NT + 8);  // fp16 output tile of the coalesced epilogue
    constexpr int SM = (3 * CF_PLANE + 8) > OT ? (3 * CF_PLANE + 8) : OT;
    __shared__ __attribute__((aligned(16))) half_t stile[SM];
    half_t* patch = stile;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lrow = lane & 31, lh = lane >> 5;
    int bid = blockIdx.x;
    const int tx = bid % tiles_x;
    bid /= tiles_x;
    const int ty = bid % tiles_y;
    const int n = bid / tiles_y;
    const int iy0 = ty * CF_TH * 2 - 1, ix0 = tx * CF_TW * 2 - 1;
    const T* ip = img + (size_t)n * 3 * H * W;
    // patch column 0 is the halo (image column ix0 = 128*tx - 1); columns 1..128 are image columns 128*tx .. 128*tx+127,
    // i.e. 16 aligned groups of 8 pixels per (channel, row): one vector load + one 16-byte LDS store each when the
    // image row allows it (W % 8 == 0), element-wise otherwise
    const bool vec = !(W & 7) && !((uintptr_t)img & 15);
    for (int idx = tid; idx < 3 * CF_PR * 17; idx += 256) {
        const int cr = idx / 17, j = idx - cr * 17;       // (channel,row) pair, group j: 0 = halo, 1..16 = 8-pixel groups
        const int c = cr / CF_PR, r = cr - c * CF_PR;
        const int iy = iy0 + r;
        const bool rowok = (unsigned)iy < (unsigned)H;
        const T* rp = ip + ((size_t)c * H + (rowok ? iy : 0)) * W;
        half_t* lp = patch + c * CF_PLANE + r * CF_ROW + CF_C0;
        if (j == 0) {
            const int ix = ix0;
            lp[0] = (rowok && ix >= 0) ? (half_t)(float)rp[ix] : (half_t)0.f;
        } else {
            const int x0 = ix0 + 1 + 8 * (j - 1);
            half8 v = {0, 0, 0, 0, 0, 0, 0, 0};
            if (rowok && x0 < W) {
                if (vec) {  // x0 % 8 == 0 and W % 8 == 0: the whole group is inside the row
                    if (sizeof(T) == 2) {
                        v = *reinterpret_cast<const half8*>(rp + x0);
                    } else {
                        const f32x4 a = *reinterpret_cast<const f32x4*>(rp + x0), b = *reinterpret_cast<const f32x4*>(rp + x0 + 4);
#pragma unroll
                        for (int e = 0; e < 4; ++e) { v[e] = (half_t)a[e]; v[4 + e] = (half_t)b[e]; }
                    }
                } else {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = (x0 + e < W) ? (half_t)(float)rp[x0 + e] : (half_t)0.f;
                }
            }
            *reinterpret_cast<half8*>(lp + 1 + 8 * (j - 1)) = v;
        }
    }
    if (tid < 8) patch[CF_ZERO + tid] = (half_t)0.f;

    // weights: A operand, rows = output channels
    half8 afr[NT][2];
#pragma unroll
    for (int a = 0; a < NT; ++a)
#pragma unroll
        for (int s = 0; s < 2; ++s)
            afr[a][s] = *reinterpret_cast<const half8*>(wgt + (size_t)(a * 32 + lrow) * 32 + 16 * s + 8 * lh);
    // LDS offsets of this lane's 16 k-values relative to its pixel's patch origin
    int koff[2][8];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int k = 16 * s + 8 * lh + j;
            const int tap = k / 3, c = k - tap * 3;
            const int kh = tap / 3, kw = tap - kh * 3;
            koff[s][j] = k < 27 ? c * CF_PLANE + kh * CF_ROW + kw : -1;
        }
    __syncthreads();

    f32x16 acc[NT][2];
#pragma unroll
    for (int a = 0; a < NT; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        const int pixbase = (2 * wave) * CF_ROW + CF_C0 + 2 * (b * 32 + lrow);
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            half8 bf;
#pragma unroll
            for (int j = 0; j < 8; ++j) bf[j] = patch[koff[s][j] >= 0 ? pixbase + koff[s][j] : CF_ZERO];
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
                const f32x4 bv = *reinterpret_cast<const f32x4*>(bias + c);
                half4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float t = acc[a][b][4 * g + e] + bv[e];
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
