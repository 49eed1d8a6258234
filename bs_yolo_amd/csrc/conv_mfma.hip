// Implicit-GEMM convolution for gfx950 (MI355X): y = act(conv2d(x, W) + b) [+ residual], NHWC fp16 activations.
//
// Replaces Conv.forward_fuse (nn/modules/conv.py:149-151) for k in {1,3}, stride in {1,2}, groups=1, together with
// the tensor plumbing around it that the reference materialises in HBM:
//   * torch.cat (Concat conv.py:445-455; C2f/C3/SPPF/C2PSA cat) -> the K loop walks up to two source views,
//     producers write straight into channel slices of the consumer's buffer (dst ld/offset);
//   * nn.Upsample(nearest, x2) -> a source flagged `up` is read at (iy>>1, ix>>1);
//   * chunk/split -> a source view is a channel slice (pointer offset + row stride);
//   * Bottleneck / PSABlock shortcut add -> residual added after the activation in the epilogue.
//
// GEMM view: D[cout][pixel] = sum_k Wt[cout][k] * P[pixel][k], k = (kh, kw, cin) with cin fastest, so both MFMA
// operands are K-contiguous 16-byte fragments: NHWC gives P, the host packs Wt as [CoutPad][Kpad].
// MFMA: v_mfma_f32_32x32x16_f16, A operand = weights (rows = cout), B operand = pixels (cols = pixel).  The
// accumulator then holds, per lane, ONE pixel and groups of 4 consecutive output channels -> 8-byte NHWC stores.
//
// Tile: 256 threads = 4 waves (WAVES_M x WAVES_N), wave tile (MT*32 pixels) x (NT*32 couts), BK = 32.
// LDS rows are 64 B (32 halves) with the 16-B chunk index XOR-swizzled by (row>>2)&3 so that ds_read_b128 of 16
// consecutive rows hits 16 distinct slots of the 256-B bank row (guide T2).  Global->LDS staging is register
// prefetch (loads for step t+1 issued before the MFMAs of step t, written after them; one barrier per step).
#include "common.h"

struct ConvK {
    const half_t* src0;
    const half_t* src1;
    int ld0, ld1, C0, C1, up0, up1;
    int H, W, OH, OW, stride, pad;
    int M;            // B*OH*OW
    int Cin8;         // (C0+C1)/8
    int ntaps;        // ks*ks
    int nk;           // Kpad/32
    int Kpad;
    const half_t* wgt;
    const float* bias;
    void* dst;
    int ldd, Cout, out_f32;
    const half_t* res;
    int ldr;
    int act;
    int dst_scale, dst_dy, dst_dx;
    int ntn;  // number of cout tiles
};

template <int KS, int WAVES_M, int WAVES_N, int MT, int NT>
__global__ __launch_bounds__(256) void conv_mfma_kernel(const ConvK p) {
    constexpr int TM = WAVES_M * MT * 32;
    constexpr int TN = WAVES_N * NT * 32;
    constexpr int PI = TM / 64;             // pixel rows staged per thread
    constexpr int WI = (TN + 63) / 64;      // weight rows staged per thread
    constexpr int STAGE = (TM + TN) * 32;   // halves per LDS stage
    __shared__ __attribute__((aligned(16))) half_t smem[2 * STAGE];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;

    // XCD-aware, bijective block remap: blocks that share an XCD (bid % 8) get consecutive logical tiles, so the
    // cout tiles of one pixel tile re-read the pixel operand from that XCD's L2.
    int wg;
    {
        const int nb = gridDim.x, bid = blockIdx.x;
        const int q = nb >> 3, r = nb & 7, x = bid & 7;
        wg = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (bid >> 3);
    }
    const int tn_idx = wg % p.ntn;
    const int tm_idx = wg / p.ntn;
    const int m0 = tm_idx * TM;
    const int n0 = tn_idx * TN;

    // ---- per-thread staging coordinates ----
    const int kc = tid & 3;   // which 16-B chunk of the 64-B K row
    const int r0 = tid >> 2;  // 0..63
    int img[PI], iy0[PI], ix0[PI];
    bool rvalid[PI];
    const int ohw = p.OH * p.OW;
#pragma unroll
    for (int i = 0; i < PI; ++i) {
        const int m = m0 + r0 + 64 * i;
        rvalid[i] = m < p.M;
        const int mm = rvalid[i] ? m : 0;
        const int n = mm / ohw;
        const int rem = mm - n * ohw;
        const int oh = rem / p.OW;
        const int ow = rem - oh * p.OW;
        img[i] = n;
        iy0[i] = oh * p.stride - p.pad;
        ix0[i] = ow * p.stride - p.pad;
    }
    // K position of this thread's chunk: tap index and channel-chunk index inside the (concatenated) Cin
    int tap = kc / p.Cin8;
    int c8 = kc - tap * p.Cin8;

    half8 pre_p[PI];
    half8 pre_w[WI];
    const half_t* wrow[WI];
#pragma unroll
    for (int j = 0; j < WI; ++j) wrow[j] = p.wgt + (size_t)(n0 + r0 + 64 * j) * p.Kpad + kc * 8;

    auto load_global = [&](int kt) {
        const bool kvalid = tap < p.ntaps;
        const int kh = (KS == 1) ? 0 : tap / KS;
        const int kw = (KS == 1) ? 0 : tap - kh * KS;
        const int cc = c8 * 8;
        const bool s1 = cc >= p.C0;
        const half_t* base = s1 ? p.src1 : p.src0;
        const int ld = s1 ? p.ld1 : p.ld0;
        const int up = s1 ? p.up1 : p.up0;
        const int c = s1 ? cc - p.C0 : cc;
        const int Hs = p.H >> up, Ws = p.W >> up;
#pragma unroll
        for (int i = 0; i < PI; ++i) {
            const int iy = iy0[i] + kh, ix = ix0[i] + kw;
            const bool ok = rvalid[i] && kvalid && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
            half8 v = {0, 0, 0, 0, 0, 0, 0, 0};
            if (ok) {
                const size_t pix = (size_t)(img[i] * Hs + (iy >> up)) * Ws + (ix >> up);
                v = *reinterpret_cast<const half8*>(base + pix * ld + c);
            }
            pre_p[i] = v;
        }
#pragma unroll
        for (int j = 0; j < WI; ++j) {
            if (TN >= 64 || r0 < TN) pre_w[j] = *reinterpret_cast<const half8*>(wrow[j] + (size_t)kt * 32);
        }
        // advance to the next K step (4 chunks further)
        c8 += 4;
        while (c8 >= p.Cin8) {
            c8 -= p.Cin8;
            ++tap;
        }
    };
    auto store_lds = [&](int buf) {
        half_t* sP = smem + buf * STAGE;
        half_t* sW = sP + TM * 32;
#pragma unroll
        for (int i = 0; i < PI; ++i) {
            const int row = r0 + 64 * i;
            *reinterpret_cast<half8*>(sP + row * 32 + ((kc ^ ((row >> 2) & 3)) << 3)) = pre_p[i];
        }
#pragma unroll
        for (int j = 0; j < WI; ++j) {
            const int row = r0 + 64 * j;
            if (TN >= 64 || r0 < TN) *reinterpret_cast<half8*>(sW + row * 32 + ((kc ^ ((row >> 2) & 3)) << 3)) = pre_w[j];
        }
    };

    f32x16 acc[NT][MT];
#pragma unroll
    for (int a = 0; a < NT; ++a)
#pragma unroll
        for (int b = 0; b < MT; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    const int lrow = lane & 31;
    const int lh = lane >> 5;

    load_global(0);
    store_lds(0);
    __syncthreads();

    for (int kt = 0; kt < p.nk; ++kt) {
        const int cur = kt & 1;
        const bool more = kt + 1 < p.nk;
        if (more) load_global(kt + 1);
        const half_t* sP = smem + cur * STAGE;
        const half_t* sW = sP + TM * 32;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int chunk = 2 * ks + lh;
            half8 bfr[MT], afr[NT];
#pragma unroll
            for (int b = 0; b < MT; ++b) {
                const int row = (wm * MT + b) * 32 + lrow;
                bfr[b] = *reinterpret_cast<const half8*>(sP + row * 32 + ((chunk ^ ((row >> 2) & 3)) << 3));
            }
#pragma unroll
            for (int a = 0; a < NT; ++a) {
                const int row = (wn * NT + a) * 32 + lrow;
                afr[a] = *reinterpret_cast<const half8*>(sW + row * 32 + ((chunk ^ ((row >> 2) & 3)) << 3));
            }
#pragma unroll
            for (int a = 0; a < NT; ++a)
#pragma unroll
                for (int b = 0; b < MT; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(afr[a], bfr[b], acc[a][b], 0, 0, 0);
        }
        if (more) store_lds(cur ^ 1);
        __syncthreads();
    }

    // ---- epilogue: bias + SiLU (+ residual) -> NHWC store, 4 consecutive couts per 8-byte (f16) / 16-byte (f32) store
#pragma unroll
    for (int b = 0; b < MT; ++b) {
        const int m = m0 + (wm * MT + b) * 32 + lrow;
        if (m >= p.M) continue;
        size_t dpix = (size_t)m;
        if (p.dst_scale != 1) {
            const int n = m / ohw;
            const int rem = m - n * ohw;
            const int oh = rem / p.OW;
            const int ow = rem - oh * p.OW;
            dpix = ((size_t)n * (p.OH * p.dst_scale) + (oh * p.dst_scale + p.dst_dy)) * (size_t)(p.OW * p.dst_scale) +
                   (ow * p.dst_scale + p.dst_dx);
        }
#pragma unroll
        for (int a = 0; a < NT; ++a) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int c = n0 + (wn * NT + a) * 32 + 8 * g + 4 * lh;
                if (c >= p.Cout) continue;
                const f32x4 bv = *reinterpret_cast<const f32x4*>(p.bias + c);  // bias is padded to CoutPad
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float t = acc[a][b][4 * g + e] + bv[e];
                    if (p.act) t = silu_f(t);
                    v[e] = t;
                }
                const bool full = c + 3 < p.Cout;
                if (p.res) {
                    const half_t* rp = p.res + dpix * p.ldr + c;
                    if (full) {
                        const half4 rv = *reinterpret_cast<const half4*>(rp);
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] += (float)rv[e];
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            if (c + e < p.Cout) v[e] += (float)rp[e];
                    }
                }
                if (p.out_f32) {
                    float* dp = reinterpret_cast<float*>(p.dst) + dpix * p.ldd + c;
                    if (full) {
                        f32x4 o = {v[0], v[1], v[2], v[3]};
                        *reinterpret_cast<f32x4*>(dp) = o;
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            if (c + e < p.Cout) dp[e] = v[e];
                    }
                } else {
                    half_t* dp = reinterpret_cast<half_t*>(p.dst) + dpix * p.ldd + c;
                    if (full) {
                        half4 o = {(half_t)v[0], (half_t)v[1], (half_t)v[2], (half_t)v[3]};
                        *reinterpret_cast<half4*>(dp) = o;
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            if (c + e < p.Cout) dp[e] = (half_t)v[e];
                    }
                }
            }
        }
    }
}

template <int KS, int WM, int WN, int MT, int NT>
static int launch_cfg(const ConvK& k, hipStream_t s) {
    constexpr int TM = WM * MT * 32, TN = WN * NT * 32;
    ConvK p = k;
    p.ntn = ceil_div(k.Cout, TN);
    const long long nblk = (long long)ceil_div(k.M, TM) * p.ntn;
    if (nblk <= 0 || nblk > 0x7fffffffLL) BSY_FAIL(BSY_ERR_ARG, "conv: grid %lld out of range", nblk);
    hipLaunchKernelGGL((conv_mfma_kernel<KS, WM, WN, MT, NT>), dim3((unsigned)nblk), dim3(256), 0, s, p);
    HIP_TRY(hipGetLastError());
    return BSY_OK;
}

int bsy_conv_packed_dims(int C2, int C1, int ksize, int* cout_pad, int* k_pad) {
    if (C2 <= 0 || C1 <= 0 || (ksize != 1 && ksize != 3)) BSY_FAIL(BSY_ERR_ARG, "conv_packed_dims: bad shape");
    if (cout_pad) *cout_pad = round_up(C2, 128);
    if (k_pad) *k_pad = round_up(ksize * ksize * C1, 32);
    return BSY_OK;
}

int launch_conv(const ConvArgs& a, hipStream_t s) {
    const int Cin = a.C0 + a.C1;
    if (a.ksize != 1 && a.ksize != 3) BSY_FAIL(BSY_ERR_ARG, "conv: ksize %d unsupported", a.ksize);
    if (a.stride != 1 && a.stride != 2) BSY_FAIL(BSY_ERR_ARG, "conv: stride %d unsupported", a.stride);
    if ((a.C0 & 7) || (a.C1 & 7) || (a.ld0 & 7) || (a.C1 && (a.ld1 & 7)))
        BSY_FAIL(BSY_ERR_ARG, "conv: channel counts/strides must be multiples of 8 (C0=%d C1=%d ld0=%d ld1=%d)", a.C0,
                 a.C1, a.ld0, a.ld1);
    if ((a.ldd & 3) || (a.res && (a.ldr & 3))) BSY_FAIL(BSY_ERR_ARG, "conv: dst/res row stride must be a multiple of 4");
    if (((uintptr_t)a.src0 & 15) || ((uintptr_t)a.src1 & 15) || ((uintptr_t)a.wgt & 15) || ((uintptr_t)a.bias & 15) ||
        ((uintptr_t)a.dst & 7) || ((uintptr_t)a.res & 7))
        BSY_FAIL(BSY_ERR_ARG, "conv: misaligned pointer");
    if (a.C1 && !a.src1) BSY_FAIL(BSY_ERR_ARG, "conv: src1 missing");
    if ((a.up0 && ((a.H | a.W) & 1)) || (a.up1 && ((a.H | a.W) & 1))) BSY_FAIL(BSY_ERR_ARG, "conv: odd size with upsample");
    if (a.OH != (a.H + 2 * a.pad - a.ksize) / a.stride + 1 || a.OW != (a.W + 2 * a.pad - a.ksize) / a.stride + 1)
        BSY_FAIL(BSY_ERR_ARG, "conv: output extent mismatch");
    const long long M = (long long)a.B * a.OH * a.OW;
    if (M <= 0 || M > 0x7fffffffLL) BSY_FAIL(BSY_ERR_ARG, "conv: M out of range");
    ConvK k;
    k.src0 = a.src0; k.src1 = a.src1; k.ld0 = a.ld0; k.ld1 = a.ld1; k.C0 = a.C0; k.C1 = a.C1;
    k.up0 = a.up0; k.up1 = a.up1; k.H = a.H; k.W = a.W; k.OH = a.OH; k.OW = a.OW; k.stride = a.stride; k.pad = a.pad;
    k.M = (int)M; k.Cin8 = Cin / 8; k.ntaps = a.ksize * a.ksize;
    k.Kpad = round_up(a.ksize * a.ksize * Cin, 32); k.nk = k.Kpad / 32;
    k.wgt = a.wgt; k.bias = a.bias; k.dst = a.dst; k.ldd = a.ldd; k.Cout = a.Cout; k.out_f32 = a.out_f32;
    k.res = a.res; k.ldr = a.ldr; k.act = a.act;
    k.dst_scale = a.dst_scale > 0 ? a.dst_scale : 1; k.dst_dy = a.dst_dy; k.dst_dx = a.dst_dx; k.ntn = 1;
    // tile choice: weights are padded to 128 output rows, so any TN <= 128 may over-read safely
    if (a.ksize == 1) {
        if (a.Cout > 64) return launch_cfg<1, 2, 2, 2, 2>(k, s);
        if (a.Cout > 32) return launch_cfg<1, 4, 1, 2, 2>(k, s);
        return launch_cfg<1, 4, 1, 2, 1>(k, s);
    } else {
        if (a.Cout > 64) return launch_cfg<3, 2, 2, 2, 2>(k, s);
        if (a.Cout > 32) return launch_cfg<3, 4, 1, 2, 2>(k, s);
        return launch_cfg<3, 4, 1, 2, 1>(k, s);
    }
}
