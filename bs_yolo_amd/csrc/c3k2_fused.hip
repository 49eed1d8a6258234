// Fused C3k2 block (block.py:3796-3804 with c3k = False, n = 1; C2f.forward :3308-3312; Bottleneck :3405-3419):
//     y0, y1 = cv1(x).chunk(2);  y2 = y1 + m.cv2(m.cv1(y1));  out = cv2(cat(y0, y1, y2))
// -- four Conv+BN+SiLU layers and a shortcut, 1x1 / 3x3 / 3x3 / 1x1 -- as ONE launch for the thin block at 1/4 resolution
// (YOLO11s model.2: x 64 ch -> c = 32, hidden 16 -> out 128 ch at 160 x 160; YOLO11n model.4 has the same widths).
//
// Unfused (round 1: cv1 conv + fused Bottleneck + cv2 conv) this block moved 1.36 GB through HBM per batch of 64 for
// 0.21 GB in and 0.42 GB out and took 0.33 ms of a 3.3 ms forward: the [y0 | y1 | y2] concat buffer is written by three
// launches and read back by two.  Here a workgroup (4 waves, persistent, two per CU) owns a 4 x 16 output tile:
//   * the 8 x 20 input patch (halo 2 for the two 3x3 convs; zeros outside the image) is fetched one tile ahead straight INTO
//     THE MFMA B FRAGMENTS of cv1 (round 3: lane = patch pixel, 8 consecutive channels per 16-byte load -- exactly the fragment
//     layout, so x never passes through LDS; the 23 KiB it used to park there were what held the kernel at two workgroups per CU);
//   * S1  cv1 on all 160 patch pixels -> y1 patch (zeros outside the MAP: the bottleneck's zero padding applies to y1, not
//         to x) and the interior y0 tile, both in LDS;
//   * S2  m.cv1 3x3 on the 6 x 18 hidden patch (zeros outside the map), S3  m.cv2 3x3 + shortcut -> y2 tile: the
//         arithmetic of bneck_fused.hip;
//   * S4  cv2 over K = [y0 | y1 | y2] straight from the three LDS tiles -> output tile in LDS -> coalesced 16-byte stores.
// All weights stay on chip for the whole launch (cv1's and m.cv2's in LDS; m.cv1's and each wave's 32 output channels of
// cv2 as MFMA A fragments in registers).  K orders (channels
// ascending for the 1x1 convs, tap-major for the 3x3 convs) and the epilogue arithmetic (f16(SiLU), shortcut added in f32)
// equal conv_mfma.hip's / bneck_fused.hip's, so the block returns bit for bit what the three-launch path returns.
#include "common.h"

#define CK_TH 4
#define CK_TW 16
#define CK_XR (CK_TH + 4)
#define CK_XC (CK_TW + 4)
#define CK_NX (CK_XR * CK_XC)  // 160 patch pixels (x and y1) = 5 MFMA pixel tiles
#define CK_MR (CK_TH + 2)
#define CK_MC (CK_TW + 2)
#define CK_NM (CK_MR * CK_MC)  // 108 hidden pixels (4 MFMA pixel tiles)
#define CK_NPX (CK_TH * CK_TW)  // 64 output pixels (2 MFMA pixel tiles)

struct C3k2K {
    const half_t* src;
    half_t* dst;
    const half_t *w1, *wa, *wb, *w4;  // packed [CoutPad][Kpad] (cv1, m.cv1, m.cv2, cv2)
    const float *b1, *ba, *bb, *b4;
    int B, H, W, lds, ldd, K1, Ka, Kb, K4, tiles_x, tiles_y, ntiles;
    unsigned magic_x, magic_y;
};

// 4 waves, THREE workgroups per CU (48 KiB of LDS each, round 3; two of 71 KiB before).  The block's SiLU evaluations (38 k per 128
// output pixels) make it look VALU-bound on paper, but the SQ counters of the two-workgroup form say its waves WAIT: 50 % of the wave
// cycles in s_waitcnt / s_barrier, 26 % issuing VALU, i.e. a SIMD's VALU busy half of the time (profiles/r03_pmc_fused_kernels.txt) --
// the five phases of a tile are separated by workgroup barriers and only other workgroups fill the gaps.  A third resident
// workgroup per CU is what the LDS diet buys: x patch no longer parked (23 KiB), the output tile aliases the dead [y1 | y0] tiles.
// (A single 8-wave workgroup -- first version, 8 x 16 tile, 131 KiB -- kept both waves of a SIMD in the same phase and ran at the
// unfused speed, 0.31 ms vs 0.33 ms.)
template <int CIN, int C, int C2>
__global__ __launch_bounds__(256, 3) void c3k2_fused_kernel(const C3k2K p) {
    constexpr int CH = C / 2;
    constexpr int YS = C + 8, MS = CH + 8, OS = C2 + 8;                       // padded LDS entries (halves)
    constexpr int WAS = 9 * C + 8, WBS = 9 * CH + 8;                           // padded weight rows (halves)
    constexpr int KS1 = CIN / 16, KSA = 9 * C / 16, KSB = 9 * CH / 16, KSEG = C / 16;  // MFMA K-steps
    static_assert(C == 32 && CH == 16 && CIN % 16 == 0 && C2 == 128 && CK_NX == 160 && CK_NPX == 64, "instantiated for the c = 32 C3k2 block");
    constexpr int SY1 = CK_NX * YS, SY0 = CK_NPX * YS, SM = CK_NM * MS, SY2 = CK_NPX * YS;
    constexpr int SOUT = CK_NPX * OS;  // the output tile aliases [y1 patch | y0 tile], dead once cv2's MFMAs have read them (barrier D')
    static_assert(SOUT <= SY1 + SY0, "output tile must fit the y1 + y0 tiles it aliases");
    constexpr int SWA = CH * WAS, SWB = C * WBS;
    constexpr int W0S = CIN + 8, SW0 = C * W0S;                                // cv1's y0 rows (waves 1 and 2 only): LDS, not registers
    __shared__ __attribute__((aligned(16))) half_t lds[SY1 + SY0 + SM + SY2 + SWA + SWB + SW0 + 2 * (2 * C + CH + C + C2)];
    half_t* sy1 = lds;
    half_t* sy0 = sy1 + SY1;
    half_t* sout = lds;
    half_t* smid = sy0 + SY0;
    half_t* sy2 = smid + SM;
    half_t* swa = sy2 + SY2;
    half_t* swb = swa + SWA;
    half_t* sw0 = swb + SWB;
    float* sb1 = reinterpret_cast<float*>(sw0 + SW0);
    float* sba = sb1 + 2 * C;
    float* sbb = sba + CH;
    float* sb4 = sbb + C;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lrow = lane & 31, lh = lane >> 5;

    for (int i = tid; i < 2 * C; i += 256) sb1[i] = p.b1[i];
    if (tid < CH) sba[tid] = p.ba[tid];
    if (tid < C) sbb[tid] = p.bb[tid];
    for (int i = tid; i < C2; i += 256) sb4[i] = p.b4[i];
    // register-resident MFMA A fragments for the whole launch: cv1's y1 rows (every wave multiplies with them) and THIS wave's 32
    // output channels of cv2 (S4 splits cv2 by cout tile: wave = cout tile).  cv1's y0 rows (second job of waves 1 and 2), m.cv1's
    // and m.cv2's weights live in LDS (m.cv1: its CH = 16 real rows; lanes of the zero rows 16..31 read row lrow - 16 instead --
    // what they produce are accumulator rows no epilogue reads).  With the y0 rows in registers too the kernel spilled 12 VGPRs at
    // three workgroups per CU (168 registers).
    half8 w1y1[KS1], a4[3 * KSEG];
#pragma unroll
    for (int ks = 0; ks < KS1; ++ks) w1y1[ks] = *reinterpret_cast<const half8*>(p.w1 + (size_t)(C + lrow) * p.K1 + 16 * ks + 8 * lh);
    for (int i = tid; i < C * (CIN / 8); i += 256) {
        const int row = i / (CIN / 8), ch = i - row * (CIN / 8);
        *reinterpret_cast<half8*>(sw0 + row * W0S + ch * 8) = *reinterpret_cast<const half8*>(p.w1 + (size_t)row * p.K1 + ch * 8);
    }
#pragma unroll
    for (int ks = 0; ks < 3 * KSEG; ++ks) a4[ks] = *reinterpret_cast<const half8*>(p.w4 + (size_t)(32 * wave + lrow) * p.K4 + 16 * ks + 8 * lh);
    for (int i = tid; i < CH * (9 * C / 8); i += 256) {
        const int row = i / (9 * C / 8), ch = i - row * (9 * C / 8);
        *reinterpret_cast<half8*>(swa + row * WAS + ch * 8) = *reinterpret_cast<const half8*>(p.wa + (size_t)row * p.Ka + ch * 8);
    }
    for (int i = tid; i < C * (9 * CH / 8); i += 256) {
        const int row = i / (9 * CH / 8), ch = i - row * (9 * CH / 8);
        *reinterpret_cast<half8*>(swb + row * WBS + ch * 8) = *reinterpret_cast<const half8*>(p.wb + (size_t)row * p.Kb + ch * 8);
    }

    // Tile-independent lane constants: nothing that depends only on the lane is recomputed per tile.
    auto tile_origin = [&](int tile, int& n, int& oy0, int& ox0) {
        const int r = (int)__umulhi((unsigned)tile, p.magic_x);
        const int tx = tile - r * p.tiles_x;
        n = (int)__umulhi((unsigned)r, p.magic_y);
        const int ty = r - n * p.tiles_y;
        oy0 = ty * CK_TH;
        ox0 = tx * CK_TW;
    };
    // S1 jobs of this wave (compile-time job list per wave, lane-constant entries): job A = y1 on patch pixel tile `wave`;
    // job B = y1 on pixel tile 4 (wave 0) or y0 on output pixel tile wave - 1 (waves 1, 2); wave 3 has no job B
    const int lty = lrow >> 4, ltx = lrow & 15;
    const int eA = 32 * wave + lrow, rA = eA / CK_XC, cA = eA - rA * CK_XC;
    const bool b_y0 = wave == 1 || wave == 2;
    const int eB = b_y0 ? (2 * (wave - 1) + lty + 2) * CK_XC + ltx + 2 : 128 + lrow, rB = eB / CK_XC, cB = eB - rB * CK_XC;
    const unsigned relA = (unsigned)((rA * p.W + cA) * p.lds + 8 * lh), relB = (unsigned)((rB * p.W + cB) * p.lds + 8 * lh);
    // x of the NEXT tile, already in cv1's B-fragment layout (lane = patch pixel, k-step ks = channels 16 ks + 8 lh ..): zeros outside the image
    half8 xa[KS1], xb[KS1];
    int nn = 0, noy0 = 0, nox0 = 0;
    auto fetch = [&](int tile) {
        tile_origin(tile, nn, noy0, nox0);
        // patch origin pixel (oy0 - 2, ox0 - 2): may lie outside the image -- only a base for pointer arithmetic
        const half_t* tb = p.src + ((long long)(nn * p.H + noy0 - 2) * p.W + (nox0 - 2)) * p.lds;
        const bool inA = (unsigned)(noy0 - 2 + rA) < (unsigned)p.H && (unsigned)(nox0 - 2 + cA) < (unsigned)p.W;
        const bool inB = wave < 3 && (unsigned)(noy0 - 2 + rB) < (unsigned)p.H && (unsigned)(nox0 - 2 + cB) < (unsigned)p.W;
#pragma unroll
        for (int ks = 0; ks < KS1; ++ks) {
            xa[ks] = half8{0, 0, 0, 0, 0, 0, 0, 0};
            xb[ks] = half8{0, 0, 0, 0, 0, 0, 0, 0};
            if (inA) xa[ks] = *reinterpret_cast<const half8*>(tb + relA + 16 * ks);
            if (inB) xb[ks] = *reinterpret_cast<const half8*>(tb + relB + 16 * ks);
        }
    };
    constexpr int CPRW = C2 / 8, NST = CK_NPX * CPRW / 256;  // output items of a thread: 16-byte piece cc of output pixel pr
    // S2: hidden pixel of this lane
    const int mm2 = wave * 32 + lrow, mc2 = mm2 < CK_NM ? mm2 : CK_NM - 1, r2 = mc2 / CK_MC, c2 = mc2 - r2 * CK_MC;

    const half_t* a2base = swb + lrow * WBS + 8 * lh;
    const int a1off = (lrow & (CH - 1)) * WAS + 8 * lh;

    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): retire the weight loads here, not inside the tile loop (bneck_fused.hip)
    const TileWalk tw = xcd_tile_walk(blockIdx.x, gridDim.x, p.ntiles);  // XCD-aware tile order (common.h)
    int tile = tw.tile;
    if (tile < tw.end) fetch(tile);
    __syncthreads();  // weights and biases visible
    for (; tile < tw.end; tile += tw.step) {
        const int n = nn, oy0 = noy0, ox0 = nox0;
        half_t* ob = p.dst + ((long long)(n * p.H + oy0) * p.W + ox0) * p.ldd;  // output tile origin (uniform)
        // ---- S1: cv1 (1x1, CIN -> 2C).  Seven jobs of one MFMA tile (32 pixels x 32 channels) over four waves: y1 = channels
        //      C .. 2C-1 on the five pixel tiles of the patch (zero outside the MAP: the Bottleneck's convs pad y1, not x), y0 =
        //      channels 0 .. C-1 on the two pixel tiles of the tile's own pixels (only cv2 reads y0) --------------------------------
        auto s1_job = [&](const bool is_y0, const half8 (&wf)[KS1], const half8 (&xf)[KS1], const int e, const int r, const int c, const int opix) {
            // e = patch entry of this lane's pixel, (r, c) its patch coordinates; y0 jobs write output pixel opix
            f32x16 acc;
            acc_bias(acc, sb1 + (is_y0 ? 0 : C), lh);  // accumulators start at the bias (common.h acc_bias)
#pragma unroll
            for (int ks = 0; ks < KS1; ++ks)
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[ks], xf[ks], acc, 0, 0, 0);
            const unsigned keep = (is_y0 || ((unsigned)(oy0 - 2 + r) < (unsigned)p.H && (unsigned)(ox0 - 2 + c) < (unsigned)p.W)) ? 0xffffffffu : 0u;
            half_t* d = is_y0 ? sy0 + opix * YS : sy1 + e * YS;
#pragma unroll
            for (int g = 0; g < C / 8; ++g) {
                union { half4 h; unsigned u[2]; } o;
                const f32x4 tv = silu4_f(f32x4{acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]});
#pragma unroll
                for (int q = 0; q < 4; ++q) o.h[q] = (half_t)tv[q];
                o.u[0] &= keep;
                o.u[1] &= keep;
                *reinterpret_cast<half4*>(d + 8 * g + 4 * lh) = o.h;
            }
        };
        s1_job(false, w1y1, xa, eA, rA, cA, 0);
        if (wave == 0) s1_job(false, w1y1, xb, eB, rB, cB, 0);
        else if (wave < 3) {
            int wo = lrow * W0S + 8 * lh;
            asm volatile("" : "+v"(wo));  // per-tile opaque, as for m.cv1's weights below: the fragments stay in LDS
            half8 w1y0[KS1];
#pragma unroll
            for (int ks = 0; ks < KS1; ++ks) w1y0[ks] = *reinterpret_cast<const half8*>(sw0 + wo + 16 * ks);
            s1_job(true, w1y0, xb, eB, rB, cB, 32 * (wave - 1) + lrow);
        }
        if (tile + tw.step < tw.end) fetch(tile + tw.step);  // x of the next tile: its registers are free now, its latency has S2..S4 to pass
        lds_barrier();  // (B) y1 patch + y0 tile complete

        // ---- S2: m.cv1 3x3 (C -> CH) on the 6 x 18 hidden patch: MFMA pixel tile `wave` ------------------------------------------
        {
            const int mm = mm2, r = r2, c = c2;
            const half_t* yb = sy1 + (r * CK_XC + c) * YS + 8 * lh;
            int ao = a1off;
            asm volatile("" : "+v"(ao));  // per-tile opaque: keeps the 18 weight fragments in LDS (hoisted out of the tile loop they cost 72 VGPRs)
            const half_t* a1base = swa + ao;
            f32x16 acc;
            acc_bias(acc, sba, lh);  // (rows >= CH read into the next bias array: accumulator rows no epilogue uses)
#pragma unroll
            for (int ks = 0; ks < KSA; ++ks) {
                const int k0 = 16 * ks, tap = k0 / C, ch0 = (k0 % C) / 8;  // compile-time
                const half8 bf = *reinterpret_cast<const half8*>(yb + ((tap / 3) * CK_XC + tap % 3) * YS + ch0 * 8);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(*reinterpret_cast<const half8*>(a1base + 16 * ks), bf, acc, 0, 0, 0);
            }
            const unsigned keep = ((unsigned)(oy0 - 1 + r) < (unsigned)p.H && (unsigned)(ox0 - 1 + c) < (unsigned)p.W) ? 0xffffffffu : 0u;
            if (mm < CK_NM) {
#pragma unroll
                for (int g = 0; g < CH / 8; ++g) {
                    union { half4 h; unsigned u[2]; } o;
                    const f32x4 tv = silu4_f(f32x4{acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]});
#pragma unroll
                    for (int q = 0; q < 4; ++q) o.h[q] = (half_t)tv[q];
                    o.u[0] &= keep;  // outside the map: m.cv2's zero padding
                    o.u[1] &= keep;
                    *reinterpret_cast<half4*>(smid + mm * MS + 8 * g + 4 * lh) = o.h;
                }
            }
        }
        lds_barrier();  // (C) hidden patch complete

        // ---- S3: m.cv2 3x3 (CH -> C) + shortcut: the two MFMA pixel tiles of the output tile, waves 0 and 1 ------------------------
        if (wave < 2) {
            const int ty2 = 2 * wave + lty;
            const half_t* b2base = smid + (ty2 * CK_MC + ltx) * MS + 8 * lh;       // hidden entry of tap (0, 0)
            const half_t* y1own = sy1 + ((ty2 + 2) * CK_XC + ltx + 2) * YS;        // the pixel's own y1 entry (the shortcut)
            f32x16 acc;
            acc_bias(acc, sbb, lh);
#pragma unroll
            for (int ks = 0; ks < KSB; ++ks) {
                const int k0 = 16 * ks, tap = k0 / CH, ch0 = (k0 % CH) / 8;
                const half8 bf = *reinterpret_cast<const half8*>(b2base + ((tap / 3) * CK_MC + tap % 3) * MS + ch0 * 8);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(*reinterpret_cast<const half8*>(a2base + 16 * ks), bf, acc, 0, 0, 0);
            }
#pragma unroll
            for (int g = 0; g < C / 8; ++g) {
                const half4 rv = *reinterpret_cast<const half4*>(y1own + 8 * g + 4 * lh);
                half4 o;
                const f32x4 tv = silu4_f(f32x4{acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]});
#pragma unroll
                for (int q = 0; q < 4; ++q) o[q] = (half_t)((float)(half_t)tv[q] + (float)rv[q]);
                *reinterpret_cast<half4*>(sy2 + (32 * wave + lrow) * YS + 8 * g + 4 * lh) = o;
            }
        }
        lds_barrier();  // (D) y2 tile complete

        // ---- S4: cv2 (1x1, 3C -> C2) over [y0 | y1 | y2]: cout tile `wave`, both pixel tiles ----------------------------------------
        {
            f32x16 acc[2];
#pragma unroll
            for (int q2 = 0; q2 < 2; ++q2) acc_bias(acc[q2], sb4 + 32 * wave, lh);
#pragma unroll
            for (int ks = 0; ks < 3 * KSEG; ++ks) {
                const int seg = ks / KSEG, kk = ks - seg * KSEG;  // compile-time: 0 = y0, 1 = y1, 2 = y2
#pragma unroll
                for (int q2 = 0; q2 < 2; ++q2) {
                    const int prow = 32 * q2 + lrow;
                    const half_t* sp = seg == 0 ? sy0 + prow * YS : (seg == 1 ? sy1 + ((2 * q2 + lty + 2) * CK_XC + ltx + 2) * YS : sy2 + prow * YS);
                    acc[q2] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a4[ks], *reinterpret_cast<const half8*>(sp + 16 * kk + 8 * lh), acc[q2], 0, 0, 0);
                }
            }
            lds_barrier();  // (D') every wave's cv2 MFMAs have read y0 / y1: their tiles become the output tile
#pragma unroll
            for (int q2 = 0; q2 < 2; ++q2)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int cc = 32 * wave + 8 * g + 4 * lh;
                    half4 o;
                    const f32x4 tv = silu4_f(f32x4{acc[q2][4 * g], acc[q2][4 * g + 1], acc[q2][4 * g + 2], acc[q2][4 * g + 3]});
#pragma unroll
                    for (int q = 0; q < 4; ++q) o[q] = (half_t)tv[q];
                    *reinterpret_cast<half4*>(sout + (32 * q2 + lrow) * OS + cc) = o;
                }
        }
        lds_barrier();  // (E) output tile complete

#pragma unroll
        for (int j = 0; j < NST; ++j) {
            const int id = tid + 256 * j, pr = id / CPRW, cc = (id % CPRW) * 8, sy = pr / CK_TW, sx_ = pr % CK_TW;
            if (oy0 + sy < p.H && ox0 + sx_ < p.W)
                *reinterpret_cast<half8*>(ob + (size_t)(sy * p.W + sx_) * p.ldd + cc) = *reinterpret_cast<const half8*>(sout + pr * OS + cc);
        }
        lds_barrier();  // (F) output tile read: the next iteration's S1 overwrites it with the next y1 / y0 tiles
    }
}

bool c3k2_fused_supported(int Cin, int C, int C2) { return Cin == 64 && C == 32 && C2 == 128; }

int launch_c3k2_fused(const C3k2Args& a, hipStream_t s) {
    if (!c3k2_fused_supported(a.Cin, a.C, a.C2)) BSY_FAIL(BSY_ERR_ARG, "c3k2: unsupported widths (Cin %d, c %d, C2 %d): need (64, 32, 128)", a.Cin, a.C, a.C2);
    if (a.B <= 0 || a.H <= 0 || a.W <= 0) BSY_FAIL(BSY_ERR_ARG, "c3k2: bad extent");
    const void* ptrs[10] = {a.src, a.dst, a.w1, a.wa, a.wb, a.w4, a.b1, a.ba, a.bb, a.b4};
    for (const void* q : ptrs)
        if (!q || ((uintptr_t)q & 15)) BSY_FAIL(BSY_ERR_ARG, "c3k2: null or misaligned pointer");
    if ((a.lds & 7) || (a.ldd & 7) || a.lds < a.Cin || a.ldd < a.C2) BSY_FAIL(BSY_ERR_ARG, "c3k2: bad leading dimension");
    C3k2K k;
    k.src = a.src; k.dst = a.dst; k.w1 = (const half_t*)a.w1; k.wa = (const half_t*)a.wa; k.wb = (const half_t*)a.wb; k.w4 = (const half_t*)a.w4;
    k.b1 = a.b1; k.ba = a.ba; k.bb = a.bb; k.b4 = a.b4;
    k.B = a.B; k.H = a.H; k.W = a.W; k.lds = a.lds; k.ldd = a.ldd;
    int cp = 0;
    bsy_conv_packed_dims(2 * a.C, a.Cin, 1, &cp, &k.K1);
    bsy_conv_packed_dims(a.C / 2, a.C, 3, &cp, &k.Ka);
    bsy_conv_packed_dims(a.C, a.C / 2, 3, &cp, &k.Kb);
    bsy_conv_packed_dims(a.C2, 3 * a.C, 1, &cp, &k.K4);
    k.tiles_x = ceil_div(a.W, CK_TW); k.tiles_y = ceil_div(a.H, CK_TH);
    if (k.tiles_x < 2) k.tiles_x = 2;  // keep the multiply-high divisions exact; the extra tiles lie outside the map
    if (k.tiles_y < 2) k.tiles_y = 2;
    const long long nt = (long long)a.B * k.tiles_x * k.tiles_y;
    if (nt * (k.tiles_x > k.tiles_y ? k.tiles_x : k.tiles_y) >= (1LL << 32)) BSY_FAIL(BSY_ERR_ARG, "c3k2: tile count out of range");
    k.ntiles = (int)nt;
    k.magic_x = (unsigned)(((1ULL << 32) + k.tiles_x - 1) / k.tiles_x);
    k.magic_y = (unsigned)(((1ULL << 32) + k.tiles_y - 1) / k.tiles_y);
    const int grid = k.ntiles < 768 ? k.ntiles : 768;  // three 256-thread workgroups (48 KiB of LDS each) per CU
    hipLaunchKernelGGL((c3k2_fused_kernel<64, 32, 128>), dim3(grid), dim3(256), 0, s, k);
    HIP_TRY(hipGetLastError());
    return BSY_OK;
}
