// Dense convolution of the "fp32x" engine mode: fp32 storage, fp32-class accuracy, fp16 matrix pipe.
//
// Replaces Conv.forward_fuse (nn/modules/conv.py:149-151) for fp32 callers at ~22 significant bits instead of the exact fp32
// chain of conv32_mfma.hip: every f32 operand is split into an f16 pair  v = hi + lo  (hi = f16(v), lo = f16(v - hi): v - hi is
// exact in f32, lo carries its leading 11 bits, so hi + lo reproduces v to ~2^-21 relative) and the product is taken as
//     w x  ~=  hi_w hi_x + hi_w lo_x + lo_w hi_x            (the dropped lo_w lo_x term is 2^-22 of the product)
// = THREE v_mfma_f32_32x32x16_f16 into ONE f32 accumulator tile per 16-deep K sub-step.  Products of two f16 values are exact in
// f32 and the pipe accumulates in f32, so what is lost against conv32_mfma is the 2^-21 representation error of the operands and the
// summation order -- far inside the north-star's 1e-3 (measured: tests/test_gpu_parity.py, fp32x cases).  Matrix-pipe roof:
// 2.5 PFLOP/s / 3 = 833 TFLOP/s of useful work against 157 TFLOP/s for v_mfma_f32_32x32x2_f32.
//
// Storage stays the fp32 mode's (NHWC f32 views, so every non-conv kernel of ref32.hip serves this mode unchanged); the split is
// done where it is cheapest:
//   weights : once, on the host (weights.py pack_record: two f16 planes [Cout][Kpad], K = (kh, kw, cin) padded to 32, behind the
//             f32 matrix the exact kernels read);
//   pixels  : in the staging path -- global_load_dwordx4 (f32) -> registers -> v_cvt_pkrtz_f16_f32 / subtract / convert ->
//             ds_write_b128 into an f16 hi tile and an f16 lo tile.  (Round-toward-zero for hi is as good as round-to-nearest here:
//             v - hi stays exact, lo picks up the difference; and it never produces an infinity from a finite input.)
// GEMM view as in conv32_mfma.hip: D[cout][pixel], A = weights (rows = cout), B = pixels => lane = pixel, registers = couts.
// Workgroup = 4 waves, tile 128 pixels x (64 NT | 32 THIN) couts, K-step 32; LDS rows of 32 halfs + 8 of padding (80 bytes: the
// 16 lanes of a ds_read_b128 group land on 16 distinct 16-byte bank slots).  Double-buffered stages, one barrier per K-step.
// The image conv (3 channels) and shapes this kernel does not take stay on the exact kernels (launch_conv32).
#include "common.h"

namespace {
#if defined(__HIP_DEVICE_COMPILE__)
typedef __amdgpu_buffer_rsrc_t cx_rsrc_t;
__device__ __forceinline__ cx_rsrc_t cx_rsrc(const void* base, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), (short)0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ f32x4 cx_load(cx_rsrc_t r, unsigned voff) {
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    union { u32x4 u; f32x4 f; } v;
    v.u = __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, 0, 0);
    return v.f;
}
#else
typedef int cx_rsrc_t;
__device__ __forceinline__ cx_rsrc_t cx_rsrc(const void*, unsigned) { return 0; }
__device__ __forceinline__ f32x4 cx_load(cx_rsrc_t, unsigned) { return f32x4{0.f, 0.f, 0.f, 0.f}; }
#endif
#define CX_OOB 0xFFFFFFE0u

typedef __fp16 pk2_t __attribute__((ext_vector_type(2)));
union H8 {
    half8 h;
    pk2_t p[4];
    f32x4 f;
};

// 8 consecutive f32 (two 16-byte pieces) -> hi / lo f16 pieces
__device__ __forceinline__ void split8(const f32x4& a, const f32x4& b, H8& hi, H8& lo) {
    const float v[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        hi.p[j] = __builtin_amdgcn_cvt_pkrtz(v[2 * j], v[2 * j + 1]);
        const float r0 = v[2 * j] - (float)hi.p[j][0], r1 = v[2 * j + 1] - (float)hi.p[j][1];
        lo.p[j] = __builtin_amdgcn_cvt_pkrtz(r0, r1);
    }
}

__device__ __forceinline__ float silu_x(float x) { return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x * -1.4426950408889634f)); }

template <int NT, bool THIN = false>
__global__ __launch_bounds__(256, 2) void conv32x_mfma_kernel(const Conv32Args a, const int M, const int ntn) {
    static_assert(!THIN || NT == 1, "thin tile: one 32-cout accumulator tile per wave");
    constexpr int TM = 128, TN = THIN ? 32 : 64 * NT, BK = 32, LDH = 40, PB = THIN ? 1 : 2;
    constexpr int WPIECES = TN * 4;                       // 16-byte pieces of one weight plane per K-step
    constexpr int WPT = (WPIECES + 255) / 256;            // per thread (the thin tile: threads 0..127 only)
    __shared__ __attribute__((aligned(16))) half_t sP[2][2][TM * LDH];  // [stage][hi | lo]
    __shared__ __attribute__((aligned(16))) half_t sW[2][2][TN * LDH];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = THIN ? wave : wave >> 1, wn = THIN ? 0 : wave & 1;
    const int prow0 = THIN ? wm * 32 : wm * 64;
    const int lj = lane & 31, lh = lane >> 5;
    const int tn_idx = blockIdx.x % ntn, tm_idx = blockIdx.x / ntn;
    const int m0 = tm_idx * TM, n0 = tn_idx * TN;
    const int Cin = a.C0 + a.C1, Cin8 = Cin >> 3;
    const int K = a.ks * a.ks * Cin, nk = (K + BK - 1) / BK;
    const int ohw = a.OH * a.OW;

    // ---- pixel items of this thread: pixels (tid >> 2) and (tid >> 2) + 64, 8-channel group (tid & 3) of every K-step ----
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
    int tap = 0, c8 = gq, tkh = 0, tkw = 0;
    while (c8 >= Cin8) { c8 -= Cin8; ++tap; if (++tkw == a.ks) { tkw = 0; ++tkh; } }
    const int H0 = a.H >> a.up0, W0 = a.W >> a.up0, H1 = a.H >> a.up1, W1 = a.W >> a.up1;
    const cx_rsrc_t rs0 = cx_rsrc(a.src0, (unsigned)((((long long)a.B * H0 * W0 - 1) * a.ld0 + a.C0) * 4));
    const unsigned plane_bytes = (unsigned)a.Cout * (unsigned)a.wx_kpad * 2u;
    const cx_rsrc_t rwh = cx_rsrc(a.wx_hi, plane_bytes), rwl = cx_rsrc(a.wx_lo, plane_bytes);

    f32x4 pv[2][2], wvh[WPT], wvl[WPT];
    auto load_step = [&](int kt) {
        const int kh = tkh, kw = tkw;
        const int c = c8 * 8;
        const bool s1 = c >= a.C0, kvalid = tap < a.ks * a.ks;
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int iy = piy[it] + kh, ix = pix_[it] + kw;
            const bool ok = kvalid && pok[it] && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
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
                const unsigned o0 = ok ? 4u * ((unsigned)((pn[it] * H0 + (iy >> a.up0)) * W0 + (ix >> a.up0)) * (unsigned)a.ld0 + (unsigned)c) : CX_OOB;
                pv[it][0] = cx_load(rs0, o0);
                pv[it][1] = cx_load(rs0, o0 + 16u);
            }
        }
#pragma unroll
        for (int j = 0; j < WPT; ++j) {
            const int id = tid + 256 * j;
            const int row = id >> 2, q = id & 3;  // cout row of the tile, 8-k piece of the K-step
            const bool ok = (WPIECES >= 256 || id < WPIECES) && n0 + row < a.Cout;  // k < Kpad always: the planes are padded to 32
            const unsigned off = ok ? 2u * ((unsigned)(n0 + row) * (unsigned)a.wx_kpad + (unsigned)(kt * BK + 8 * q)) : CX_OOB;
            wvh[j] = cx_load(rwh, off);
            wvl[j] = cx_load(rwl, off);
        }
        c8 += 4;
        while (c8 >= Cin8) { c8 -= Cin8; ++tap; if (++tkw == a.ks) { tkw = 0; ++tkh; } }
    };
    auto store_step = [&](int st) {
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            H8 hi, lo;
            split8(pv[it][0], pv[it][1], hi, lo);
            const int o = ((tid >> 2) + 64 * it) * LDH + 8 * gq;
            *reinterpret_cast<half8*>(&sP[st][0][o]) = hi.h;
            *reinterpret_cast<half8*>(&sP[st][1][o]) = lo.h;
        }
#pragma unroll
        for (int j = 0; j < WPT; ++j) {
            const int id = tid + 256 * j;
            if (WPIECES >= 256 || id < WPIECES) {
                const int o = (id >> 2) * LDH + 8 * (id & 3);
                H8 t;
                t.f = wvh[j];
                *reinterpret_cast<half8*>(&sW[st][0][o]) = t.h;
                t.f = wvl[j];
                *reinterpret_cast<half8*>(&sW[st][1][o]) = t.h;
            }
        }
    };

    // accumulators start at the bias, as in every conv kernel of the library
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
        for (int s = 0; s < 2; ++s) {
            half8 bh[PB], bl[PB], ah[NT], al[NT];
#pragma unroll
            for (int b = 0; b < PB; ++b) {
                const int o = (prow0 + b * 32 + lj) * LDH + 16 * s + 8 * lh;
                bh[b] = *reinterpret_cast<const half8*>(&sP[st][0][o]);
                bl[b] = *reinterpret_cast<const half8*>(&sP[st][1][o]);
            }
#pragma unroll
            for (int an = 0; an < NT; ++an) {
                const int o = ((wn * NT + an) * 32 + lj) * LDH + 16 * s + 8 * lh;
                ah[an] = *reinterpret_cast<const half8*>(&sW[st][0][o]);
                al[an] = *reinterpret_cast<const half8*>(&sW[st][1][o]);
            }
#pragma unroll
            for (int an = 0; an < NT; ++an)
#pragma unroll
                for (int b = 0; b < PB; ++b) {
                    acc[an][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[an], bh[b], acc[an][b], 0, 0, 0);
                    acc[an][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[an], bl[b], acc[an][b], 0, 0, 0);
                    acc[an][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[an], bh[b], acc[an][b], 0, 0, 0);
                }
        }
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
                    v[e] = a.act ? silu_x(t) : t;
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

bool conv32x_mfma_supported(const Conv32Args& a) {
    if (a.first || !a.wx_hi || !a.wx_lo || a.wx_kpad <= 0 || (a.wx_kpad & 31)) return false;
    if (!conv32_mfma_supported(a)) return false;  // the same view / alignment rules as the exact MFMA kernel
    if (((uintptr_t)a.wx_hi | (uintptr_t)a.wx_lo) & 15) return false;
    if (a.wx_kpad < a.ks * a.ks * (a.C0 + a.C1)) return false;
    if ((long long)a.Cout * a.wx_kpad * 2 >= 0xFFFFFFC0LL) return false;
    return true;
}

int launch_conv32x_mfma(const Conv32Args& a, hipStream_t s) {
    if (!conv32x_mfma_supported(a)) BSY_FAIL(BSY_ERR_ARG, "conv32x_mfma: unsupported shape / alignment");
    const long long M = (long long)a.B * a.OH * a.OW;
    if (M <= 0 || M > 0x7fffffffLL) BSY_FAIL(BSY_ERR_ARG, "conv32x_mfma: M out of range");
    const bool wide = a.Cout > 64, thin = a.Cout <= 32;
    const int ntn = ceil_div(a.Cout, wide ? 128 : (thin ? 32 : 64));
    const long long nblk = (long long)ceil_div((int)M, 128) * ntn;
    if (nblk > 0x7fffffffLL) BSY_FAIL(BSY_ERR_ARG, "conv32x_mfma: grid out of range");
    if (thin) hipLaunchKernelGGL((conv32x_mfma_kernel<1, true>), dim3((unsigned)nblk), dim3(256), 0, s, a, (int)M, ntn);
    else if (wide) hipLaunchKernelGGL((conv32x_mfma_kernel<2>), dim3((unsigned)nblk), dim3(256), 0, s, a, (int)M, ntn);
    else hipLaunchKernelGGL((conv32x_mfma_kernel<1>), dim3((unsigned)nblk), dim3(256), 0, s, a, (int)M, ntn);
    HIP_TRY(hipGetLastError());
    return BSY_OK;
}
