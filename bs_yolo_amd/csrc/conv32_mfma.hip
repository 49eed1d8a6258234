// Dense convolution of the fp32 engine mode on the fp32 matrix pipe: y = act(conv2d(x, W) + b) [+ residual], NHWC f32.
//
// Replaces Conv.forward_fuse (nn/modules/conv.py:149-151) for callers that hand the engine fp32 images -- the reference's
// predict() default is half=False (cfg/default.yaml:54, cast at engine/predictor.py:131) -- with the fp32 model's own
// arithmetic: v_mfma_f32_32x32x2_f32 multiplies f32 by f32 and accumulates in f32 with one rounding per product, bit for bit a
// k-ordered fmaf chain (MI355X_MICROARCH.md, Matrix cores).  The K walk below feeds it k = (kh, kw, cin) in ascending order with
// the accumulator starting at the bias, i.e. EXACTLY the chain conv32_kernel (ref32.hip, one thread per output) computes: the two
// kernels return the same bits (tests/test_gpu_parity.py::test_conv32_mfma_equals_scalar), which is also what makes results
// independent of which of them a shape is routed to.  Peak 157 TFLOP/s (64 FLOP/clk/SIMD), 1/16 of the fp16 MFMA rate.
//
// GEMM view as in conv_mfma.hip: D[cout][pixel] = sum_k W[k][cout] * P[pixel][k]; A operand = weights (rows = cout), B operand =
// pixels (cols = pixel) => a lane owns ONE pixel and groups of 4 consecutive couts.  Workgroup = 4 waves (2 x 2), tile 128 pixels
// x (64 NT) couts, K-step = 32 k values; wave tile 64 pixels x 32 NT couts = 2 x NT accumulator tiles of 32 x 32.
// Staging: global_load_dwordx4 -> registers -> LDS, double-buffered (one barrier per K-step); an f32 K-step is 16 x 4 x 64
// cycles of MFMA per wave, so the plain register pipeline hides the loads and no DMA ring is needed.
//   pixels : [128][36] floats (row = 32 k values + 4 of padding: ds_read_b128 of 16 consecutive rows hits 64 distinct banks);
//            inside every group of 8 k the even k are stored first, then the odd ones, so that lane half h reads four floats
//            k = 8g + 2r + h (r = 0..3) with ONE ds_read_b128 and MFMA r takes (k = 8g + 2r | 8g + 2r + 1) = ascending k;
//   weights: [32][64 NT] floats straight from the packed [K][Cout] matrix; an A fragment is a ds_read_b32 of 32 consecutive
//            floats of one k row per lane half (conflict-free).
// Restrictions (launch_conv32 falls back to the scalar kernel otherwise): C0, C1 multiples of 8, row strides multiples of 4,
// Cout a multiple of 4, 16-byte aligned views.
#include "common.h"

namespace {
__device__ __forceinline__ float silu32m(float x) { return x / (1.0f + expf(-x)); }  // = ref32.hip silu32

// Staging loads go through buffer descriptors (round 3): an offset past the view's range returns zeros, so padding taps, pixels
// past M, couts past Cout and the K tail need neither a branch around the load nor a zero-initialised destination -- the K-step's
// address arithmetic (32-bit offsets, no per-step division) shrank from ~150 to ~60 VALU instructions.
#if defined(__HIP_DEVICE_COMPILE__)
typedef __amdgpu_buffer_rsrc_t c32_rsrc_t;
__device__ __forceinline__ c32_rsrc_t c32_rsrc(const void* base, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), (short)0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ f32x4 c32_load(c32_rsrc_t r, unsigned voff) {
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    union { u32x4 u; f32x4 f; } v;
    v.u = __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, 0, 0);
    return v.f;
}
#else
typedef int c32_rsrc_t;
__device__ __forceinline__ c32_rsrc_t c32_rsrc(const void*, unsigned) { return 0; }
__device__ __forceinline__ f32x4 c32_load(c32_rsrc_t, unsigned) { return f32x4{0.f, 0.f, 0.f, 0.f}; }
#endif
#define C32_OOB 0xFFFFFFE0u  // (+16 for the second half of a piece stays out of range)

// FIRST: the image conv (BCHW image, f16 or f32, 3 channels).  Every tap is widened to 4 k values (3 real channels + 1 zero, in
// the pixel operand AND in the weight rows fetched for them; an 8-k piece = two taps), so K = 8 ceil(k^2 / 2) and the real products
// still arrive in ascending (kh, kw, c) order -- the zero products leave the chain's value unchanged.
// THIN (NT == 1): 128 pixels x 32 couts, wave grid 4 x 1 (a wave = 32 pixels x 32 couts, one accumulator tile) -- layers of 16 / 32
// output channels (the image conv, the Bottlenecks inside C3k2 at 1/4 and 1/8 resolution) spent half or three quarters of their
// 64-cycle MFMAs on cout padding in the 64-cout tile.
template <int NT, bool FIRST, bool THIN = false>
__global__ __launch_bounds__(256, NT == 2 ? 3 : 2) void conv32_mfma_kernel(const Conv32Args a, const int M, const int ntn) {
    static_assert(!THIN || NT == 1, "thin tile: one 32-cout accumulator tile per wave");
    constexpr int TM = 128, TN = THIN ? 32 : 64 * NT, BK = 32, LDP = 36, PB = THIN ? 1 : 2;
    constexpr int WPT = BK * TN / 4 / 256;  // 16-byte weight pieces per thread per K-step (2 NT)
    __shared__ __attribute__((aligned(16))) float sP[2][TM * LDP];
    // 128-cout tile: ONE weight stage (53 KiB of LDS in all = three workgroups per CU instead of two; the stage is rewritten behind a
    // second barrier per K-step, 1-2 % of a step's 64 x 64-cycle MFMAs)
    constexpr int NWS = NT == 2 ? 1 : 2;
    __shared__ __attribute__((aligned(16))) float sW[NWS][BK * TN];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = THIN ? wave : wave >> 1, wn = THIN ? 0 : wave & 1;
    const int prow0 = THIN ? wm * 32 : wm * 64;  // first pixel row of this wave's accumulator tiles
    const int lj = lane & 31, lh = lane >> 5;
    const int tn_idx = blockIdx.x % ntn, tm_idx = blockIdx.x / ntn;
    const int m0 = tm_idx * TM, n0 = tn_idx * TN;
    const int Cin = FIRST ? 8 : a.C0 + a.C1, Cin8 = Cin >> 3;
    const int K = FIRST ? 8 * ((a.ks * a.ks + 1) / 2) : a.ks * a.ks * Cin, nk = (K + BK - 1) / BK;  // FIRST: `tap` below counts tap PAIRS
    const int ohw = a.OH * a.OW;

    // ---- pixel items of this thread: pixels (tid >> 2) and (tid >> 2) + 64, k group (tid & 3) of every K-step ----
    const int gq = tid & 3;
    int pn[2], piy[2], pix_[2];
    bool pok[2];
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int m = m0 + (tid >> 2) + 64 * it;
        pok[it] = m < M;
        const int mm = pok[it] ? m : 0;
        pn[it] = mm / ohw;
        const int rem = mm - pn[it] * ohw;
        const int oh = rem / a.OW;
        piy[it] = oh * a.stride - a.pad;
        pix_[it] = (rem - oh * a.OW) * a.stride - a.pad;
    }
    int tap = 0, c8 = gq, tkh = 0, tkw = 0;  // this thread's 8-channel piece of the current K-step: k = (tap = (tkh, tkw), 8 c8 ..)
    while (c8 >= Cin8) { c8 -= Cin8; ++tap; if (++tkw == a.ks) { tkw = 0; ++tkh; } }
    const int H0 = a.H >> a.up0, W0 = a.W >> a.up0, H1 = a.H >> a.up1, W1 = a.W >> a.up1;
    const c32_rsrc_t rs0 = c32_rsrc(a.src0, FIRST ? 0u : (unsigned)((((long long)a.B * H0 * W0 - 1) * a.ld0 + a.C0) * 4));
    const c32_rsrc_t rsw = c32_rsrc(a.w, (unsigned)((long long)(FIRST ? a.ks * a.ks * 3 : K) * a.Cout * 4));

    f32x4 pv[2][2], wv[WPT];
    auto load_step = [&](int kt) {
        const int kh = tkh, kw = tkw;
        const int c = c8 * 8;
        const bool s1 = c >= a.C0, kvalid = tap < a.ks * a.ks;
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int iy = piy[it] + kh, ix = pix_[it] + kw;
            const bool ok = kvalid && pok[it] && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
            if (FIRST) {
                pv[it][0] = f32x4{0.f, 0.f, 0.f, 0.f};
                pv[it][1] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int hf = 0; hf < 2; ++hf) {  // taps 2 tap, 2 tap + 1 -> k 0..3 / 4..7 of this piece
                    const int t = 2 * tap + hf, th = t / a.ks, tw = t - th * a.ks;
                    const int jy = piy[it] + th, jx = pix_[it] + tw;
                    if (t < a.ks * a.ks && pok[it] && (unsigned)jy < (unsigned)a.H && (unsigned)jx < (unsigned)a.W) {
                        const size_t hw = (size_t)a.H * a.W, ii = (size_t)pn[it] * 3 * hw + (size_t)jy * a.W + jx;
#pragma unroll
                        for (int ch = 0; ch < 3; ++ch)
                            pv[it][hf][ch] = a.src_dtype == BSY_F16 ? (float)reinterpret_cast<const half_t*>(a.src0)[ii + ch * hw]
                                                                    : reinterpret_cast<const float*>(a.src0)[ii + ch * hw];
                    }
                }
            } else {
                if (a.C1) {  // (uniform) two concat operands: the piece's source differs from lane to lane -> plain pointers
                    pv[it][0] = f32x4{0.f, 0.f, 0.f, 0.f};
                    pv[it][1] = f32x4{0.f, 0.f, 0.f, 0.f};
                    if (ok) {
                        const float* p = s1 ? reinterpret_cast<const float*>(a.src1) + ((size_t)(pn[it] * H1 + (iy >> a.up1)) * W1 + (ix >> a.up1)) * a.ld1 + (c - a.C0)
                                            : reinterpret_cast<const float*>(a.src0) + ((size_t)(pn[it] * H0 + (iy >> a.up0)) * W0 + (ix >> a.up0)) * a.ld0 + c;
                        pv[it][0] = *reinterpret_cast<const f32x4*>(p);
                        pv[it][1] = *reinterpret_cast<const f32x4*>(p + 4);
                    }
                } else {
                    const unsigned o0 = ok ? 4u * ((unsigned)((pn[it] * H0 + (iy >> a.up0)) * W0 + (ix >> a.up0)) * (unsigned)a.ld0 + (unsigned)c) : C32_OOB;
                    pv[it][0] = c32_load(rs0, o0);
                    pv[it][1] = c32_load(rs0, o0 + 16u);
                }
            }
        }
#pragma unroll
        for (int j = 0; j < WPT; ++j) {
            const int id = tid + 256 * j;
            const int kr = id / (TN / 4), col = (id % (TN / 4)) * 4;
            const int k = kt * BK + kr;
            if (FIRST) {
                wv[j] = f32x4{0.f, 0.f, 0.f, 0.f};  // k = (tap = k / 4, channel = k % 4): rows of the packed [k^2 * 3][Cout] matrix for channel < 3, zeros for the padding
                if (k < K && (k & 3) < 3 && (k >> 2) < a.ks * a.ks && n0 + col < a.Cout)
                    wv[j] = *reinterpret_cast<const f32x4*>(a.w + (size_t)((k >> 2) * 3 + (k & 3)) * a.Cout + n0 + col);
            } else {
                wv[j] = c32_load(rsw, (k < K && n0 + col < a.Cout) ? 4u * ((unsigned)k * (unsigned)a.Cout + (unsigned)(n0 + col)) : C32_OOB);
            }
        }
        c8 += 4;
        while (c8 >= Cin8) { c8 -= Cin8; ++tap; if (++tkw == a.ks) { tkw = 0; ++tkh; } }
    };
    auto store_step = [&](int st) {
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            float* d = &sP[st][((tid >> 2) + 64 * it) * LDP + 8 * gq];
            *reinterpret_cast<f32x4*>(d) = f32x4{pv[it][0][0], pv[it][0][2], pv[it][1][0], pv[it][1][2]};      // even k
            *reinterpret_cast<f32x4*>(d + 4) = f32x4{pv[it][0][1], pv[it][0][3], pv[it][1][1], pv[it][1][3]};  // odd k
        }
#pragma unroll
        for (int j = 0; j < WPT; ++j) {
            const int id = tid + 256 * j;
            *reinterpret_cast<f32x4*>(&sW[st & (NWS - 1)][(id / (TN / 4)) * TN + (id % (TN / 4)) * 4]) = wv[j];
        }
    };

    // accumulators start at the bias (conv32_kernel: acc = bias, then the fmaf chain)
    f32x16 acc[NT][PB];
#pragma unroll
    for (int an = 0; an < NT; ++an)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int c = n0 + (wn * NT + an) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            const float bv = c < a.Cout ? a.bias[c] : 0.f;
#pragma unroll
            for (int b = 0; b < PB; ++b) acc[an][b][r] = bv;
        }

    load_step(0);
    store_step(0);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int st = kt & 1;
        if (kt + 1 < nk) load_step(kt + 1);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            f32x4 pb[PB];
#pragma unroll
            for (int b = 0; b < PB; ++b) pb[b] = *reinterpret_cast<const f32x4*>(&sP[st][(prow0 + b * 32 + lj) * LDP + 8 * g + 4 * lh]);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float af[NT];
#pragma unroll
                for (int an = 0; an < NT; ++an) af[an] = sW[st & (NWS - 1)][(8 * g + 2 * r + lh) * TN + (wn * NT + an) * 32 + lj];
#pragma unroll
                for (int an = 0; an < NT; ++an)
#pragma unroll
                    for (int b = 0; b < PB; ++b) acc[an][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[an], pb[b][r], acc[an][b], 0, 0, 0);
            }
        }
        if (NWS == 1) __syncthreads();         // every wave is done reading the single weight stage
        if (kt + 1 < nk) store_step(st ^ 1);  // stage st ^ 1 was last read in step kt - 1, behind that step's barrier
        __syncthreads();
    }

    // ---- epilogue: lane = pixel, registers 4 q .. 4 q + 3 = couts 8 q + 4 lh + {0..3} of the 32-cout tile ----
#pragma unroll
    for (int b = 0; b < PB; ++b) {
        const int m = m0 + prow0 + b * 32 + lj;
        if (m >= M) continue;
        const int n = m / ohw, rem = m - n * ohw, oh = rem / a.OW, ow = rem - oh * a.OW;
        const size_t pix = (size_t)(n * a.OH + oh) * a.OW + ow;
        size_t dp = pix;
        if (a.dst_scale != 1)
            dp = ((size_t)n * (a.OH * a.dst_scale) + (oh * a.dst_scale + a.dst_dy)) * (size_t)(a.OW * a.dst_scale) + (ow * a.dst_scale + a.dst_dx);
#pragma unroll
        for (int an = 0; an < NT; ++an)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int c = n0 + (wn * NT + an) * 32 + 8 * q + 4 * lh;
                if (c >= a.Cout) continue;  // Cout % 4 == 0: a group of four is inside or outside as a whole
                f32x4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float t = acc[an][b][4 * q + e];
                    v[e] = a.act ? silu32m(t) : t;
                }
                if (a.res) {
                    const f32x4 rv = *reinterpret_cast<const f32x4*>(a.res + pix * a.ldr + c);
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] += rv[e];
                }
                *reinterpret_cast<f32x4*>(a.dst + dp * a.ldd + c) = v;
            }
    }
}
}  // namespace

bool conv32_mfma_supported(const Conv32Args& a) {
    if (a.Cout <= 0 || (a.Cout & 3) || (a.ldd & 3) || (a.res && (a.ldr & 3))) return false;
    if (((uintptr_t)a.w | (uintptr_t)a.dst | (uintptr_t)a.res) & 15) return false;
    if (a.first) return a.C0 == 3 && !a.C1 && !a.up0 && (a.src_dtype == BSY_F16 || a.src_dtype == BSY_F32);
    if ((a.C0 & 7) || (a.C1 & 7) || a.C0 <= 0 || (a.ld0 & 3) || (a.C1 && (a.ld1 & 3))) return false;
    if (((uintptr_t)a.src0 | (uintptr_t)a.src1) & 15) return false;
    if ((a.up0 && ((a.H | a.W) & 1)) || (a.up1 && ((a.H | a.W) & 1))) return false;
    // buffer descriptors: every view and the weight matrix below 4 GiB - 64 bytes (32-bit byte offsets)
    const long long lim = 0xFFFFFFC0LL;
    if (((long long)a.B * (a.H >> a.up0) * (a.W >> a.up0) * a.ld0) * 4 >= lim) return false;
    if (a.C1 && ((long long)a.B * (a.H >> a.up1) * (a.W >> a.up1) * a.ld1) * 4 >= lim) return false;
    if ((long long)a.ks * a.ks * (a.C0 + a.C1) * a.Cout * 4 >= lim) return false;
    return true;
}

int launch_conv32_mfma(const Conv32Args& a, hipStream_t s) {
    if (!conv32_mfma_supported(a)) BSY_FAIL(BSY_ERR_ARG, "conv32_mfma: unsupported shape / alignment");
    const long long M = (long long)a.B * a.OH * a.OW;
    if (M <= 0 || M > 0x7fffffffLL) BSY_FAIL(BSY_ERR_ARG, "conv32_mfma: M out of range");
    const bool wide = a.Cout > 64, thin = a.Cout <= 32 && !getenv("BSY_CONV32_NO_THIN");
    const int ntn = ceil_div(a.Cout, wide ? 128 : (thin ? 32 : 64));
    const long long nblk = (long long)ceil_div((int)M, 128) * ntn;
    if (nblk > 0x7fffffffLL) BSY_FAIL(BSY_ERR_ARG, "conv32_mfma: grid out of range");
    if (a.first && thin) hipLaunchKernelGGL((conv32_mfma_kernel<1, true, true>), dim3((unsigned)nblk), dim3(256), 0, s, a, (int)M, ntn);
    else if (thin) hipLaunchKernelGGL((conv32_mfma_kernel<1, false, true>), dim3((unsigned)nblk), dim3(256), 0, s, a, (int)M, ntn);
    else if (a.first && wide) hipLaunchKernelGGL((conv32_mfma_kernel<2, true>), dim3((unsigned)nblk), dim3(256), 0, s, a, (int)M, ntn);
    else if (a.first) hipLaunchKernelGGL((conv32_mfma_kernel<1, true>), dim3((unsigned)nblk), dim3(256), 0, s, a, (int)M, ntn);
    else if (wide) hipLaunchKernelGGL((conv32_mfma_kernel<2, false>), dim3((unsigned)nblk), dim3(256), 0, s, a, (int)M, ntn);
    else hipLaunchKernelGGL((conv32_mfma_kernel<1, false>), dim3((unsigned)nblk), dim3(256), 0, s, a, (int)M, ntn);
    HIP_TRY(hipGetLastError());
    return BSY_OK;
}
