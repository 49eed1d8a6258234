// Fused Bottleneck: y = x + Conv3x3(Conv3x3(x)), both convs with folded BN + SiLU (block.py:3405-3419 with k = (3, 3),
// shortcut, g = 1), for the thin bottlenecks of C3k2 (block.py:3796-3804: c = 32 channels, hidden c/2 = 16 at 160 x 160
// in YOLO11s).  NHWC f16 channel-slice views in and out (x and y are slices of the C3k2 concat buffer).
//
// Unfused these two layers were 0.25-0.28 ms of a 3.9 ms forward: K = 288 / 144 and 16 / 32 output channels are too
// thin for the implicit-GEMM kernel (it stages every input pixel nine times through LDS), and the hidden map makes a
// round trip through HBM.  Here a workgroup (4 waves, persistent) owns an 8 x 16 output tile:
//   * the 12 x 20 input patch (halo 2, zeros outside the image) is fetched into registers one tile ahead and parked in
//     LDS with 16 bytes of padding per pixel, so that every B fragment of both convs is ONE ds_read_b128 at a
//     compile-time offset from a per-lane base (no swizzle arithmetic) and consecutive lanes fall on different banks;
//   * conv 1 produces the 10 x 18 hidden patch (6 MFMA pixel tiles) straight into LDS (zeros outside the map = conv 2's
//     padding); conv 2 produces the 8 x 16 tile, adds the residual from the input patch already in LDS and leaves
//     through an LDS tile as coalesced 16-byte stores;
//   * conv 1's weights live in registers as MFMA A fragments for the whole launch (72 VGPRs), conv 2's in LDS.
// K order (tap-major, channels ascending, 16 per MFMA) and the epilogue arithmetic (f16(SiLU) then + residual in f32)
// equal conv_mfma.hip's, so the result is bit-identical to the two-launch path.
#include "common.h"

#define BN_TH 8
#define BN_TW 16
#define BN_XR (BN_TH + 4)
#define BN_XC (BN_TW + 4)
#define BN_NX (BN_XR * BN_XC)  // 240 input patch pixels
#define BN_MR (BN_TH + 2)
#define BN_MC (BN_TW + 2)
#define BN_NM (BN_MR * BN_MC)  // 180 hidden patch pixels
#define BN_NMT ((BN_NM + 31) / 32)

struct BneckK {
    const half_t* src;  // x view (channel offset applied)
    half_t* dst;        // y view
    const half_t *w1, *w2;
    const float *b1, *b2;
    int B, H, W, lds, ldd, Kpad1, Kpad2, act, tiles_x, tiles_y, ntiles;
    unsigned magic_x, magic_y;
};

template <int C, int CH>
__global__ __launch_bounds__(256, 3) void bneck_fused_kernel(const BneckK p) {
    constexpr int XS = C + 8, MS = CH + 8, LDO = C + 8;  // padded entries (halves)
    constexpr int KSA = 9 * C / 16, KSB = 9 * CH / 16;   // MFMA steps of conv 1 / conv 2
    constexpr int XCH = C / 8;                           // 16-byte chunks per input pixel
    constexpr int NITEM = BN_NX * XCH, NLOAD = (NITEM + 255) / 256;
    constexpr int SX = BN_NX * XS, SM = BN_NM * MS, SO = BN_TH * BN_TW * LDO;
    static_assert(C == 32 && CH == 16, "instantiated for the YOLO11s / YOLOv8s C3k2 bottleneck at 1/4 resolution");
    constexpr int W2S = 9 * CH + 8;  // conv-2 weight row in LDS (halves): 84-dword pitch -> conflict-free A reads
    __shared__ __attribute__((aligned(16))) half_t lds[SX + SM + SO + C * W2S + 2 * (C + CH)];
    half_t* sx = lds;
    half_t* smid = lds + SX;
    half_t* sout = smid + SM;
    half_t* sw2 = sout + SO;
    float* sb1 = reinterpret_cast<float*>(sw2 + C * W2S);
    float* sb2 = sb1 + CH;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lrow = lane & 31, lh = lane >> 5;

    if (tid < CH) sb1[tid] = p.b1[tid];
    if (tid < C) sb2[tid] = p.b2[tid];
    // conv-1 weights: MFMA A fragments in registers for the whole launch (72 VGPRs; rows >= CH of the packed matrix are
    // zero).  conv-2 weights: LDS (read once per tile per wave) -- with both in registers the kernel spills, and every
    // scratch reload waits for vmcnt(0), i.e. for the NEXT tile's prefetch.
    half8 a1[KSA];
#pragma unroll
    for (int ks = 0; ks < KSA; ++ks) a1[ks] = *reinterpret_cast<const half8*>(p.w1 + (size_t)lrow * p.Kpad1 + 16 * ks + 8 * lh);
    for (int i = tid; i < C * (9 * CH / 8); i += 256) {
        const int row = i / (9 * CH / 8), ch = i - row * (9 * CH / 8);
        *reinterpret_cast<half8*>(sw2 + row * W2S + ch * 8) = *reinterpret_cast<const half8*>(p.w2 + (size_t)row * p.Kpad2 + ch * 8);
    }
    const half_t* a2base = sw2 + lrow * W2S + 8 * lh;

    // tile-independent lane tables
    int it_off[NLOAD], it_rc[NLOAD];  // LDS offset; (row | col << 8 | chunk << 16) of the patch entry
#pragma unroll
    for (int i = 0; i < NLOAD; ++i) {
        const int idx = tid + 256 * i;
        const int e = idx / XCH, ch = idx - e * XCH;
        const int r = e / BN_XC, c = e - r * BN_XC;
        it_rc[i] = r | (c << 8) | (ch << 16);
        it_off[i] = e * XS + ch * 8;
    }
    const int ty2 = 2 * wave + (lrow >> 4), tx2 = lrow & 15;             // conv-2 lane pixel inside the tile
    const half_t* b2base = smid + (ty2 * BN_MC + tx2) * MS + 8 * lh;     // hidden entry of tap (0, 0), this lane half's chunk
    const half_t* rbase = sx + ((ty2 + 2) * BN_XC + tx2 + 2) * XS + 4 * lh;  // residual: the pixel's own input entry

    auto tile_origin = [&](int tile, int& n, int& oy0, int& ox0) {
        const int r = (int)__umulhi((unsigned)tile, p.magic_x);
        const int tx = tile - r * p.tiles_x;
        n = (int)__umulhi((unsigned)r, p.magic_y);
        const int ty = r - n * p.tiles_y;
        oy0 = ty * BN_TH;
        ox0 = tx * BN_TW;
    };
    half8 pre[NLOAD];
    int nn = 0, noy0 = 0, nox0 = 0;
    auto fetch = [&](int tile) {
        tile_origin(tile, nn, noy0, nox0);
#pragma unroll
        for (int i = 0; i < NLOAD; ++i) {
            const int y = noy0 - 2 + (it_rc[i] & 255), x = nox0 - 2 + ((it_rc[i] >> 8) & 255);
            pre[i] = half8{0, 0, 0, 0, 0, 0, 0, 0};
            if (tid + 256 * i < NITEM && (unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)p.W)
                pre[i] = *reinterpret_cast<const half8*>(p.src + ((size_t)(nn * p.H + y) * p.W + x) * p.lds + (it_rc[i] >> 16) * 8);
        }
    };

    // Retire the weight / bias loads HERE: left pending, the compiler waits for them at their first use inside the tile
    // loop with vmcnt(0) -- on every iteration, which then also waits for the prefetch issued just before.
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
    const TileWalk tw = xcd_tile_walk(blockIdx.x, gridDim.x, p.ntiles);  // XCD-aware tile order (common.h)
    int tile = tw.tile;
    if (tile < tw.end) fetch(tile);
    for (; tile < tw.end; tile += tw.step) {
        const int n = nn, oy0 = noy0, ox0 = nox0;
#pragma unroll
        for (int i = 0; i < NLOAD; ++i)
            if (tid + 256 * i < NITEM) *reinterpret_cast<half8*>(sx + it_off[i]) = pre[i];
        lds_barrier();  // input patch visible; every wave is done with the previous tile's LDS
        if (tile + tw.step < tw.end) fetch(tile + tw.step);

        // ---- conv 1: 180 hidden pixels = 6 MFMA pixel tiles over 4 waves ----------------------------------------------
        for (int mt = wave; mt < BN_NMT; mt += 4) {
            const int mm = mt * 32 + lrow;
            const int mc = mm < BN_NM ? mm : BN_NM - 1;
            const int r = mc / BN_MC, c = mc - r * BN_MC;
            const half_t* xb = sx + (r * BN_XC + c) * XS + 8 * lh;  // input entry of tap (0, 0), this lane half's chunk
            f32x16 acc;
            acc_bias(acc, sb1, lh);  // accumulators start at the bias (common.h acc_bias)
#pragma unroll
            for (int ks = 0; ks < KSA; ++ks) {
                const int k0 = 16 * ks, tap = k0 / C, ch0 = (k0 % C) / 8;  // compile-time; lane half 1 = next chunk
                const half8 bf = *reinterpret_cast<const half8*>(xb + ((tap / 3) * BN_XC + tap % 3) * XS + ch0 * 8);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1[ks], bf, acc, 0, 0, 0);
            }
            const unsigned keep = ((unsigned)(oy0 - 1 + r) < (unsigned)p.H && (unsigned)(ox0 - 1 + c) < (unsigned)p.W) ? 0xffffffffu : 0u;
            if (mm < BN_NM) {
#pragma unroll
                for (int g = 0; g < CH / 8; ++g) {
                    union { half4 h; unsigned u[2]; } o;
                    f32x4 t = f32x4{acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]};
                    if (p.act) t = silu4_f(t);
#pragma unroll
                    for (int q = 0; q < 4; ++q) o.h[q] = (half_t)t[q];
                    o.u[0] &= keep;  // outside the map: conv 2's zero padding
                    o.u[1] &= keep;
                    *reinterpret_cast<half4*>(smid + mm * MS + 8 * g + 4 * lh) = o.h;
                }
            }
        }
        lds_barrier();  // hidden patch complete

        // ---- conv 2 + residual: one MFMA pixel tile (2 rows x 16) per wave ---------------------------------------------
        {
            f32x16 acc;
            acc_bias(acc, sb2, lh);
#pragma unroll
            for (int ks = 0; ks < KSB; ++ks) {
                const int k0 = 16 * ks, tap = k0 / CH, ch0 = (k0 % CH) / 8;
                const half8 bf = *reinterpret_cast<const half8*>(b2base + ((tap / 3) * BN_MC + tap % 3) * MS + ch0 * 8);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(*reinterpret_cast<const half8*>(a2base + 16 * ks), bf, acc, 0, 0, 0);
            }
            const int prow = wave * 32 + lrow;
#pragma unroll
            for (int g = 0; g < C / 8; ++g) {
                const half4 rv = *reinterpret_cast<const half4*>(rbase + 8 * g);
                half4 o;
                f32x4 t = f32x4{acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]};
                if (p.act) t = silu4_f(t);
#pragma unroll
                for (int q = 0; q < 4; ++q) o[q] = (half_t)((float)(half_t)t[q] + (float)rv[q]);
                *reinterpret_cast<half4*>(sout + prow * LDO + 8 * g + 4 * lh) = o;
            }
        }
        lds_barrier();  // output tile complete

        constexpr int CPRW = C / 8;
#pragma unroll
        for (int id = tid; id < BN_TH * BN_TW * CPRW; id += 256) {
            const int prow = id / CPRW, cc = (id % CPRW) * 8;
            const int oy = oy0 + prow / BN_TW, ox = ox0 + prow % BN_TW;
            if (oy < p.H && ox < p.W)
                *reinterpret_cast<half8*>(p.dst + ((size_t)(n * p.H + oy) * p.W + ox) * p.ldd + cc) =
                    *reinterpret_cast<const half8*>(sout + prow * LDO + cc);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// The 64 -> 32 -> 64 Bottleneck (YOLO11s model.4 / model.16 at 80 x 80; round 2).  Two launches of the patch kernel ran these
// thin layers at 350-430 TFLOP/s (a 64-wide cout tile is half empty for the 32-channel hidden map, and the hidden map
// round-trips through HBM): 0.078 ms per Bottleneck.  Same scheme as above with both weight matrices in LDS (conv 1's A
// fragments no longer fit registers: 36 K-steps), 8 waves, one workgroup per CU (143 KiB of LDS):
//   conv 1: 6 MFMA pixel tiles of the hidden patch on waves 0-5, 36 K-steps each;
//   conv 2: 4 pixel tiles x 2 cout tiles on the 8 waves, 18 K-steps each, + shortcut from the input patch.
// Lane-constant address arithmetic is hoisted out of the tile loop (c3k2_fused.hip).  K walks (conv_mfma.hip conv_korder: conv 1 with its
// 64 input channels chunk-major over two 32-channel chunks, conv 2 one chunk) and epilogue arithmetic equal the conv kernels': bit-identical
// to the two-launch path through any of their configurations.
// ---------------------------------------------------------------------------------------------------------------------
template <int C, int CH>
__global__ __launch_bounds__(512) void bneck_fused_wide_kernel(const BneckK p) {
    constexpr int XS = C + 8, MS = CH + 8, LDO = C + 8;
    constexpr int KSA = 9 * C / 16, KSB = 9 * CH / 16;
    constexpr int XCH = C / 8;
    constexpr int NITEM = BN_NX * XCH, NLOAD = (NITEM + 511) / 512;
    constexpr int W1S = 9 * C + 8, W2S = 9 * CH + 8;  // padded weight rows (halves): conflict-free A reads
    constexpr int SX = BN_NX * XS, SM = BN_NM * MS, SO = BN_TH * BN_TW * LDO;
    static_assert(C == 64 && CH == 32, "instantiated for the YOLO11s C3k2 bottleneck at 1/8 resolution");
    __shared__ __attribute__((aligned(16))) half_t lds[SX + SM + SO + CH * W1S + C * W2S + 2 * (C + CH)];
    half_t* sx = lds;
    half_t* smid = lds + SX;
    half_t* sout = smid + SM;
    half_t* sw1 = sout + SO;
    half_t* sw2 = sw1 + CH * W1S;
    float* sb1 = reinterpret_cast<float*>(sw2 + C * W2S);
    float* sb2 = sb1 + CH;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lrow = lane & 31, lh = lane >> 5;

    if (tid < CH) sb1[tid] = p.b1[tid];
    if (tid < C) sb2[tid] = p.b2[tid];
    for (int i = tid; i < CH * (9 * C / 8); i += 512) {
        const int row = i / (9 * C / 8), ch = i - row * (9 * C / 8);
        *reinterpret_cast<half8*>(sw1 + row * W1S + ch * 8) = *reinterpret_cast<const half8*>(p.w1 + (size_t)row * p.Kpad1 + ch * 8);
    }
    for (int i = tid; i < C * (9 * CH / 8); i += 512) {
        const int row = i / (9 * CH / 8), ch = i - row * (9 * CH / 8);
        *reinterpret_cast<half8*>(sw2 + row * W2S + ch * 8) = *reinterpret_cast<const half8*>(p.w2 + (size_t)row * p.Kpad2 + ch * 8);
    }

    // tile-independent lane tables
    int it_off[NLOAD], it_r[NLOAD], it_c[NLOAD];
    unsigned it_rel[NLOAD];
#pragma unroll
    for (int i = 0; i < NLOAD; ++i) {
        const int idx = tid + 512 * i;
        const int e = idx / XCH, ch = idx - e * XCH;
        const int r = e / BN_XC, c = e - r * BN_XC;
        it_r[i] = idx < NITEM ? r : 0x40000000;  // items past the patch: never in bounds
        it_c[i] = c;
        it_off[i] = e * XS + ch * 8;
        it_rel[i] = (unsigned)((r * p.W + c) * p.lds + ch * 8);
    }
    constexpr int CPRW = C / 8, NST = BN_TH * BN_TW * CPRW / 512;
    int st_y[NST], st_x[NST], st_off[NST];
    unsigned st_rel[NST];
#pragma unroll
    for (int j = 0; j < NST; ++j) {
        const int id = tid + 512 * j;
        const int pr = id / CPRW, cc = (id % CPRW) * 8;
        st_y[j] = pr / BN_TW;
        st_x[j] = pr % BN_TW;
        st_off[j] = pr * LDO + cc;
        st_rel[j] = (unsigned)((st_y[j] * p.W + st_x[j]) * p.ldd + cc);
    }
    // conv 1: hidden pixel of this lane (waves 0-5); conv 2: output pixel tile wave & 3, cout tile wave >> 2
    const int mm1 = wave * 32 + lrow, mc1 = mm1 < BN_NM ? mm1 : BN_NM - 1, r1 = mc1 / BN_MC, c1 = mc1 - r1 * BN_MC;
    const half_t* xb = sx + (r1 * BN_XC + c1) * XS + 8 * lh;
    const half_t* a1base = sw1 + lrow * W1S + 8 * lh;
    const int pt = wave & 3, ct = wave >> 2;
    const int ty2 = 2 * pt + (lrow >> 4), tx2 = lrow & 15;
    const half_t* b2base = smid + (ty2 * BN_MC + tx2) * MS + 8 * lh;
    const half_t* a2base = sw2 + (32 * ct + lrow) * W2S + 8 * lh;
    const half_t* rbase = sx + ((ty2 + 2) * BN_XC + tx2 + 2) * XS + 32 * ct + 4 * lh;  // shortcut: the pixel's own input entry
    half_t* obase = sout + (pt * 32 + lrow) * LDO + 32 * ct + 4 * lh;

    auto tile_origin = [&](int tile, int& n, int& oy0, int& ox0) {
        const int r = (int)__umulhi((unsigned)tile, p.magic_x);
        const int tx = tile - r * p.tiles_x;
        n = (int)__umulhi((unsigned)r, p.magic_y);
        const int ty = r - n * p.tiles_y;
        oy0 = ty * BN_TH;
        ox0 = tx * BN_TW;
    };
    half8 pre[NLOAD];
    int nn = 0, noy0 = 0, nox0 = 0;
    auto fetch = [&](int tile) {
        tile_origin(tile, nn, noy0, nox0);
        const half_t* tb = p.src + ((long long)(nn * p.H + noy0 - 2) * p.W + (nox0 - 2)) * p.lds;  // patch origin (a base only)
#pragma unroll
        for (int i = 0; i < NLOAD; ++i) {
            pre[i] = half8{0, 0, 0, 0, 0, 0, 0, 0};
            if ((unsigned)(noy0 - 2 + it_r[i]) < (unsigned)p.H && (unsigned)(nox0 - 2 + it_c[i]) < (unsigned)p.W)
                pre[i] = *reinterpret_cast<const half8*>(tb + it_rel[i]);
        }
    };

    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): retire the weight loads here, not inside the tile loop
    const TileWalk tw = xcd_tile_walk(blockIdx.x, gridDim.x, p.ntiles);  // XCD-aware tile order (common.h)
    int tile = tw.tile;
    if (tile < tw.end) fetch(tile);
    for (; tile < tw.end; tile += tw.step) {
        const int n = nn, oy0 = noy0, ox0 = nox0;
#pragma unroll
        for (int i = 0; i < NLOAD; ++i)
            if (tid + 512 * i < NITEM) *reinterpret_cast<half8*>(sx + it_off[i]) = pre[i];
        half_t* ob = p.dst + ((long long)(n * p.H + oy0) * p.W + ox0) * p.ldd;
        lds_barrier();  // input patch visible (first iteration: the weights too); the previous tile's output has been read
        if (tile + tw.step < tw.end) fetch(tile + tw.step);

        // ---- conv 1: 180 hidden pixels = 6 MFMA pixel tiles on waves 0-5 -----------------------------------------------------
        if (wave < BN_NMT) {
            f32x16 acc;
            acc_bias(acc, sb1, lh);
#pragma unroll
            for (int ks = 0; ks < KSA; ++ks) {
                // the layer's K walk (conv_mfma.hip conv_korder: 3x3, Cin % 32 == 0 -> chunk-major, 32-channel chunks): chunk, tap, 16-wide half
                const int chunk = ks / 18, tap = (ks % 18) >> 1, k0 = tap * C + 32 * chunk + 16 * (ks & 1), ch0 = (k0 % C) / 8;  // compile-time
                const half8 bf = *reinterpret_cast<const half8*>(xb + ((tap / 3) * BN_XC + tap % 3) * XS + ch0 * 8);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(*reinterpret_cast<const half8*>(a1base + k0), bf, acc, 0, 0, 0);
            }
            const unsigned keep = ((unsigned)(oy0 - 1 + r1) < (unsigned)p.H && (unsigned)(ox0 - 1 + c1) < (unsigned)p.W) ? 0xffffffffu : 0u;
            if (mm1 < BN_NM) {
#pragma unroll
                for (int g = 0; g < CH / 8; ++g) {
                    union { half4 h; unsigned u[2]; } o;
                    f32x4 t = f32x4{acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]};
                    if (p.act) t = silu4_f(t);
#pragma unroll
                    for (int q = 0; q < 4; ++q) o.h[q] = (half_t)t[q];
                    o.u[0] &= keep;  // outside the map: conv 2's zero padding
                    o.u[1] &= keep;
                    *reinterpret_cast<half4*>(smid + mm1 * MS + 8 * g + 4 * lh) = o.h;
                }
            }
        }
        lds_barrier();  // hidden patch complete

        // ---- conv 2 + shortcut: MFMA pixel tile pt (2 rows x 16), cout tile ct (32 channels) per wave ----------------------------
        {
            f32x16 acc;
            acc_bias(acc, sb2 + 32 * ct, lh);
#pragma unroll
            for (int ks = 0; ks < KSB; ++ks) {
                const int k0 = 16 * ks, tap = k0 / CH, ch0 = (k0 % CH) / 8;
                const half8 bf = *reinterpret_cast<const half8*>(b2base + ((tap / 3) * BN_MC + tap % 3) * MS + ch0 * 8);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(*reinterpret_cast<const half8*>(a2base + 16 * ks), bf, acc, 0, 0, 0);
            }
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const half4 rv = *reinterpret_cast<const half4*>(rbase + 8 * g);
                half4 o;
                f32x4 t = f32x4{acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]};
                if (p.act) t = silu4_f(t);
#pragma unroll
                for (int q = 0; q < 4; ++q) o[q] = (half_t)((float)(half_t)t[q] + (float)rv[q]);
                *reinterpret_cast<half4*>(obase + 8 * g) = o;
            }
        }
        lds_barrier();  // output tile complete
#pragma unroll
        for (int j = 0; j < NST; ++j)
            if (oy0 + st_y[j] < p.H && ox0 + st_x[j] < p.W)
                *reinterpret_cast<half8*>(ob + st_rel[j]) = *reinterpret_cast<const half8*>(sout + st_off[j]);
    }
}

bool bneck_fused_supported(int C, int CH) { return (C == 32 && CH == 16) || (C == 64 && CH == 32); }

int launch_bneck_fused(const BneckArgs& a, hipStream_t s) {
    if (!bneck_fused_supported(a.C, a.CH)) BSY_FAIL(BSY_ERR_ARG, "bottleneck: unsupported widths (C %d, hidden %d): need (32, 16) or (64, 32)", a.C, a.CH);
    if (a.B <= 0 || a.H <= 0 || a.W <= 0) BSY_FAIL(BSY_ERR_ARG, "bottleneck: bad extent");
    if (((uintptr_t)a.src & 15) || ((uintptr_t)a.dst & 15) || (a.lds & 7) || (a.ldd & 7) || a.lds < a.C || a.ldd < a.C ||
        ((uintptr_t)a.w1 & 15) || ((uintptr_t)a.w2 & 15) || ((uintptr_t)a.b1 & 15) || ((uintptr_t)a.b2 & 15) ||
        a.Kpad1 < 9 * a.C || a.Kpad2 < 9 * a.CH || (a.Kpad1 & 7) || (a.Kpad2 & 7))
        BSY_FAIL(BSY_ERR_ARG, "bottleneck: misaligned pointer / leading dimension");
    BneckK k;
    k.src = a.src; k.dst = a.dst; k.w1 = (const half_t*)a.w1; k.w2 = (const half_t*)a.w2; k.b1 = a.b1; k.b2 = a.b2;
    k.B = a.B; k.H = a.H; k.W = a.W; k.lds = a.lds; k.ldd = a.ldd; k.Kpad1 = a.Kpad1; k.Kpad2 = a.Kpad2; k.act = a.act;
    k.tiles_x = ceil_div(a.W, BN_TW); k.tiles_y = ceil_div(a.H, BN_TH);
    if (k.tiles_x < 2) k.tiles_x = 2;  // keep the multiply-high divisions exact (magic for 1 would be 2^32); the extra
    if (k.tiles_y < 2) k.tiles_y = 2;  // tiles lie outside the map and store nothing
    const long long nt = (long long)a.B * k.tiles_x * k.tiles_y;
    if (nt * (k.tiles_x > k.tiles_y ? k.tiles_x : k.tiles_y) >= (1LL << 32)) BSY_FAIL(BSY_ERR_ARG, "bottleneck: tile count out of range");
    k.ntiles = (int)nt;
    k.magic_x = (unsigned)(((1ULL << 32) + k.tiles_x - 1) / k.tiles_x);
    k.magic_y = (unsigned)(((1ULL << 32) + k.tiles_y - 1) / k.tiles_y);
    if (a.C == 64) {
        const int grid = k.ntiles < 256 ? k.ntiles : 256;  // one 512-thread workgroup (143 KiB of LDS) per CU
        hipLaunchKernelGGL((bneck_fused_wide_kernel<64, 32>), dim3(grid), dim3(512), 0, s, k);
        HIP_TRY(hipGetLastError());
        return BSY_OK;
    }
    const int grid = k.ntiles < 768 ? k.ntiles : 768;  // 3 workgroups per CU
    hipLaunchKernelGGL((bneck_fused_kernel<32, 16>), dim3(grid), dim3(256), 0, s, k);
    HIP_TRY(hipGetLastError());
    return BSY_OK;
}
