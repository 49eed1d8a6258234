// PMSFA's tail -- depthwise 5x5, depthwise 7x7, concat, 1x1 conv, shortcut -- as ONE launch (round 4).
// Reference: /root/reference/ultralytics/nn/modules/block.py:3035-3054
//     conv1_out = conv1(x);  p1, p2 = conv1_out.chunk(2);  conv2_out = conv2(p1) [5x5 depthwise];  q1, q2 = conv2_out.chunk(2)
//     conv3_out = conv3(q1) [7x7 depthwise];  out = conv4(cat([conv3_out, q2, p2])) + x
// conv1 (a dense 3x3, the patch kernel of conv_mfma.hip) stays its own launch and writes P = [p1 | p2].  The unfused plan then runs
// three launches that move the map five times (5x5: read p1, write Q; 7x7: read Q, write S; 1x1: read [S | p2] and x, write): at
// 160 x 160 x 32 channels and 64 images 52 + 63 + 82 us for ~0.9 GB of traffic.  Here a workgroup owns a TH x TW pixel tile:
//   phase 1  p1 of the tile + a halo of 5 pixels -> LDS (row-contiguous 16-byte buffer loads, zero padding = out-of-range result)
//   phase 2  5x5 on p1: q1 over the tile + a halo of 3 -> LDS (ZERO outside the image: conv3 pads its INPUT, it does not see
//            the 5x5 of padding), q2 over the tile -> LDS
//   phase 3  7x7 on q1 -> LDS; p2 of the tile: global -> LDS (over the dead p1 patch)
//   phase 4  conv4 on the matrix pipe: B operand = the tile's pixels, read per 8-channel piece from the three LDS arrays that make
//            up the reference's concat (no concat is ever materialised); bias-started accumulators, SiLU, round to f16, + x, store
// so the map is read once (p1 with its halo, p2, x) and written once.  Workgroups are persistent (two per CU, XCD-aware tile walk): the three
// convs' weights are staged once, and a tile's p2 / x pieces and the NEXT tile's patch are requested before the depthwise arithmetic starts.
// What bounds it (rocprofv3 SQ counters, profiles/r04_h_pmc_pmsfa_tail.txt): vector-ALU issue -- 8 250 wave instructions per 16 x 16 tile,
// v_fma_mix_f32 at 4.3-4.8 cycles (tools/valu_rate.hip) -- at two waves per SIMD: VALU active 33 % of the wave cycles, LDS 5 %, MFMA 0.5 %.
// Arithmetic = the unfused kernels', step for step: depthwise taps in (dy, dx) order as f32 FMAs on f16 inputs starting at the bias
// (bsyolo_ops.hip dwconv_tile_kernel), SiLU in f32 and one rounding to f16 per intermediate map, conv4's K walk in ascending
// 16-channel MFMA steps from accumulators that start at the bias, activation rounded to f16 BEFORE the shortcut is added
// (conv_mfma.hip conv_epilogue_lds) -- the fused and the unfused plan return the same bits (tests/test_gpu_parity.py).
#include "common.h"

namespace {

#if defined(__HIP_DEVICE_COMPILE__)
typedef __amdgpu_buffer_rsrc_t pf_rsrc_t;
__device__ __forceinline__ pf_rsrc_t pf_make_rsrc(const void* base, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), (short)0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ half8 pf_load16(pf_rsrc_t r, unsigned voff) {
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    union { u32x4 u; half8 h; } v;
    v.u = __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, 0, 0);
    return v.h;
}
#else
typedef int pf_rsrc_t;
__device__ __forceinline__ pf_rsrc_t pf_make_rsrc(const void*, unsigned) { return 0; }
__device__ __forceinline__ half8 pf_load16(pf_rsrc_t, unsigned) { return half8{0, 0, 0, 0, 0, 0, 0, 0}; }
#endif
#define PF_OOB 0xFFFFFFF0u

constexpr int pf_max(int a, int b) { return a > b ? a : b; }

// __launch_bounds__(256, 2): left to itself the compiler took 228 VGPRs + 32 AGPRs = 260 > 256, ONE workgroup per CU (a persistent grid of
// 512 then runs as two batches: SQ_WAVE_CYCLES showed every wave alive for half the kernel) -- 169-207 us on model.2 instead of 147.
template <int C, int TH, int TW>
__global__ __launch_bounds__(256, 2) void pmsfa_tail_kernel(const PmsfaArgs a, int tiles_x, int tiles_y, int ntiles, unsigned spanP) {
    constexpr int CH = C / 2, CQ = C / 4;     // channels of p1 (= p2) and of q1 (= q2 = conv3_out)
    constexpr int NP1 = CH / 8, NQ1 = CQ / 8;  // their 8-channel pieces
    constexpr int P1H = TH + 10, P1W = TW + 10, Q1H = TH + 6, Q1W = TW + 6, NPX = TH * TW;
    // LDS geometry of the two depthwise patches.  The depthwise phases put consecutive LANES on consecutive ROWS, so a row pitch of
    // 16 x odd bytes makes the 16-byte window reads conflict-free: pixels of CH + 8 halves (48 / 80 bytes) resp. 24 halves (48 bytes),
    // rows one pixel longer than the patch (27 x 48 = 16 x 81, 27 x 80 = 16 x 135, 23 x 48 = 16 x 69).  Tried instead (docs/experiments.md
    // section 0.4): dense patches, 43 KiB and THREE workgroups per CU at C = 32 -- 144 vs 147 us, and 91 vs 80 us at C = 64 (two either way).
    constexpr int PXS1 = CH + 8, PXSQ = 24, P1WP = P1W + 1, Q1WP = Q1W + 1;
    static_assert(((P1WP * PXS1 * 2) % 32) == 16 && ((Q1WP * PXSQ * 2) % 32) == 16 && CQ <= 16, "row pitch must be 16 x odd bytes");
    constexpr int LDP2 = CH + 8, LDS1 = CQ + 8, LDQ2 = CQ + 8, LDO = C + 8, LDW = C + 8;  // MFMA operand rows: 16 x odd bytes
    constexpr int SLACK = 8;  // pixels: the last 4-pixel group of a halo row reads its window past the row end (results discarded)
    constexpr int A_HALVES = pf_max(pf_max((P1H * P1WP + SLACK) * PXS1, NPX * LDO), NPX * (LDP2 + LDS1));
    static_assert(C % 32 == 0 && TW % 4 == 0 && NPX % 128 == 0, "tile / width not supported");
    __shared__ __attribute__((aligned(16))) half_t sA[A_HALVES];                      // p1 patch; then p2 + conv3_out; then the output tile
    __shared__ __attribute__((aligned(16))) half_t sQ1[(Q1H * Q1WP + SLACK) * PXSQ];   // q1 with its halo of 3
    __shared__ __attribute__((aligned(16))) half_t sQ2[NPX * LDQ2];
    __shared__ __attribute__((aligned(16))) half_t sW4[C * LDW];
    __shared__ __attribute__((aligned(16))) float sW2[26 * CH];  // [tap][channel], then the bias row
    __shared__ __attribute__((aligned(16))) float sW3[50 * CQ];

    const int tid = threadIdx.x;
    const int H = a.H, W = a.W;
    const int lane = tid & 63, wave = tid >> 6, lrow = lane & 31, lh = lane >> 5;

    // ---- once per workgroup: the three convs' weights -> LDS ----------------------------------------------------------
    for (int i = tid; i < 26 * CH; i += 256) {
        const int row = i / CH, c = i - row * CH;
        sW2[i] = row < 25 ? a.w2[(size_t)row * a.wld2 + c] : a.b2[c];
    }
    for (int i = tid; i < 50 * CQ; i += 256) {
        const int row = i / CQ, c = i - row * CQ;
        sW3[i] = row < 49 ? a.w3[(size_t)row * a.wld3 + c] : a.b3[c];
    }
    for (int i = tid; i < C * (C / 8); i += 256) {
        const int row = i / (C / 8), pc = i - row * (C / 8);
        *reinterpret_cast<half8*>(sW4 + row * LDW + pc * 8) = *reinterpret_cast<const half8*>(a.w4 + (size_t)row * a.kpad4 + pc * 8);
    }

    // ---- per thread, tile-independent: which pieces of the p1 patch / of p2 / of the output it moves -----------------------
    constexpr int NPATCH = (P1H * P1W * NP1 + 255) / 256;
    constexpr int NLD = NPX * NP1 / 256;
    constexpr int RW = NPX / 4, PCS = C / 8;  // phase 4: rows per wave, 16-byte pieces per row
    constexpr int NOUT = RW * PCS / 64;       // output pieces per lane
    const pf_rsrc_t rs = pf_make_rsrc(a.P, spanP);
    auto load_patch = [&](half8 (&pv)[NPATCH], int tile) {
        int t = tile;
        const int tx = t % tiles_x; t /= tiles_x;
        const int ty = t % tiles_y;
        const int n = t / tiles_y;
        const int oy0 = ty * TH, ox0 = tx * TW;
#pragma unroll
        for (int i = 0; i < NPATCH; ++i) {
            const int id = tid + 256 * i;
            const int ch = id % NP1, px = (id / NP1) % P1W, py = id / (NP1 * P1W);
            const int y = oy0 - 5 + py, x = ox0 - 5 + px;
            const bool ok = id < P1H * P1W * NP1 && (unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W;
            pv[i] = pf_load16(rs, ok ? 2u * ((unsigned)((n * H + y) * W + x) * (unsigned)a.ldp + (unsigned)(ch * 8)) : PF_OOB);
        }
    };

    const TileWalk tw = xcd_tile_walk(blockIdx.x, gridDim.x, ntiles);  // neighbouring tiles share an XCD's L2 (halos)
    constexpr bool PREF = true;  // the next tile's patch is fetched under the current tile's arithmetic
    half8 patch[NPATCH];
    if (PREF && tw.tile < tw.end) load_patch(patch, tw.tile);
    for (int tile = tw.tile; tile < tw.end; tile += tw.step) {
        int t = tile;
        const int tx = t % tiles_x; t /= tiles_x;
        const int ty = t % tiles_y;
        const int n = t / tiles_y;
        const int oy0 = ty * TH, ox0 = tx * TW;

        // ---- phase 1: the p1 patch (loaded while the previous tile was computed) -> LDS; this tile's p2 and x and the NEXT tile's patch
        //      are requested here and stay in flight through the two depthwise phases
        if (!PREF) load_patch(patch, tile);
        lds_barrier();  // the previous tile's output rows have left sA (first tile: nothing to wait for)
#pragma unroll
        for (int i = 0; i < NPATCH; ++i) {
            const int id = tid + 256 * i;
            const int ch = id % NP1, pp = id / NP1;
            if (id < P1H * P1W * NP1) *reinterpret_cast<half8*>(sA + ((pp / P1W) * P1WP + pp % P1W) * PXS1 + ch * 8) = patch[i];
        }
        half8 p2v[NLD], xv[NOUT];
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int id = tid + 256 * i;
            const int pc = id % NP1, px = id / NP1;
            const int y = oy0 + px / TW, x = ox0 + px % TW;
            p2v[i] = (y < H && x < W) ? *reinterpret_cast<const half8*>(a.P + ((size_t)(n * H + y) * W + x) * a.ldp + CH + pc * 8) : half8{0, 0, 0, 0, 0, 0, 0, 0};
        }
#pragma unroll
        for (int i = 0; i < NOUT; ++i) {
            const int id = lane + 64 * i;
            const int row = wave * RW + id / PCS, pc = id % PCS;
            const int y = oy0 + row / TW, x = ox0 + row % TW;
            xv[i] = (y < H && x < W) ? *reinterpret_cast<const half8*>(a.x + ((size_t)(n * H + y) * W + x) * a.ldx + pc * 8) : half8{0, 0, 0, 0, 0, 0, 0, 0};
        }
        if (PREF && tile + tw.step < tw.end) load_patch(patch, tile + tw.step);
        lds_barrier();

        // ---- phase 2: depthwise 5x5 + SiLU.  item = (row, group of 4 pixels, 8-channel piece): first q1 over the tile + halo 3, then q2 over the tile
        {
            constexpr int G1 = (Q1W + 3) / 4, G2 = TW / 4;
            constexpr int N5A = Q1H * G1 * NQ1, N5B = TH * G2 * NQ1;
#pragma unroll 1
            for (int it = tid; it < N5A + N5B; it += 256) {
                const bool halo = it < N5A;
                int j = halo ? it : it - N5A;
                const int R = halo ? Q1H : TH, G = halo ? G1 : G2;
                const int r = j % R; j /= R;  // lanes along y
                const int g = j % G, ch = j / G;
                const int prow0 = halo ? r : r + 3, pcol0 = halo ? 4 * g : 4 * g + 3;  // window origin in the p1 patch
                const int cch = halo ? ch : NQ1 + ch;                                   // piece of p1
                float acc[4][8];
                {
                    const f32x4 b0 = *reinterpret_cast<const f32x4*>(sW2 + 25 * CH + cch * 8), b1 = *reinterpret_cast<const f32x4*>(sW2 + 25 * CH + cch * 8 + 4);
#pragma unroll
                    for (int p = 0; p < 4; ++p)
#pragma unroll
                        for (int e = 0; e < 4; ++e) { acc[p][e] = b0[e]; acc[p][4 + e] = b1[e]; }
                }
#pragma unroll
                for (int dy = 0; dy < 5; ++dy) {
                    const half_t* row = sA + ((prow0 + dy) * P1WP + pcol0) * PXS1 + cch * 8;
                    half8 win[8];
#pragma unroll
                    for (int q = 0; q < 8; ++q) win[q] = *reinterpret_cast<const half8*>(row + q * PXS1);
#pragma unroll
                    for (int dx = 0; dx < 5; ++dx) {
                        const float* wp = sW2 + (dy * 5 + dx) * CH + cch * 8;
                        const f32x4 w0 = *reinterpret_cast<const f32x4*>(wp), w1 = *reinterpret_cast<const f32x4*>(wp + 4);
#pragma unroll
                        for (int p = 0; p < 4; ++p) fma_mix8(acc[p], win[p + dx], w0, w1);
                    }
                }
#pragma unroll
                for (int p = 0; p < 4; ++p) {
                    half8 o;
#pragma unroll
                    for (int e = 0; e < 8; ++e) o[e] = (half_t)silu_f(acc[p][e]);
                    const int col = 4 * g + p;
                    if (halo) {
                        if (col >= Q1W) break;
                        const int y = oy0 - 3 + r, x = ox0 - 3 + col;
                        if (!((unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W)) o = half8{0, 0, 0, 0, 0, 0, 0, 0};  // conv3's zero padding
                        *reinterpret_cast<half8*>(sQ1 + (r * Q1WP + col) * PXSQ + ch * 8) = o;
                    } else {
                        *reinterpret_cast<half8*>(sQ2 + (r * TW + col) * LDQ2 + ch * 8) = o;
                    }
                }
            }
        }
        lds_barrier();  // the p1 patch is dead from here: sA becomes [p2 | conv3_out]

        half_t* sP2 = sA;
        half_t* sS1 = sA + NPX * LDP2;
        // ---- phase 3: depthwise 7x7 + SiLU on q1; p2 of the tile registers -> LDS ---------------------------------------
        {
            constexpr int PX7 = (NPX * (CQ / 4)) / 256;  // pixels per thread along x: every thread owns PX7 pixels of one 4-channel group
            static_assert((PX7 == 2 || PX7 == 4) && TW % PX7 == 0, "7x7 item shape");
            constexpr int GX = TW / PX7, NC4 = CQ / 4;
            const int cq = tid % NC4, py = (tid / NC4) % TH, gx = tid / (NC4 * TH);  // lanes along (channel group, y)
            static_assert(NC4 * TH * GX == 256, "7x7 item count");
            float acc[PX7][4];
            {
                const f32x4 b = *reinterpret_cast<const f32x4*>(sW3 + 49 * CQ + cq * 4);
#pragma unroll
                for (int p = 0; p < PX7; ++p)
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[p][e] = b[e];
            }
#pragma unroll
            for (int dy = 0; dy < 7; ++dy) {
                const half_t* row = sQ1 + ((py + dy) * Q1WP + PX7 * gx) * PXSQ + cq * 4;
                unsigned win[PX7 + 6][2];
#pragma unroll
                for (int q = 0; q < PX7 + 6; ++q) {
                    const uint2 v = *reinterpret_cast<const uint2*>(row + q * PXSQ);
                    win[q][0] = v.x; win[q][1] = v.y;
                }
#pragma unroll
                for (int dx = 0; dx < 7; ++dx) {
                    const f32x4 w = *reinterpret_cast<const f32x4*>(sW3 + (dy * 7 + dx) * CQ + cq * 4);
#pragma unroll
                    for (int p = 0; p < PX7; ++p) {
                        acc[p][0] = fma_mix_lo(win[p + dx][0], w[0], acc[p][0]);
                        acc[p][1] = fma_mix_hi(win[p + dx][0], w[1], acc[p][1]);
                        acc[p][2] = fma_mix_lo(win[p + dx][1], w[2], acc[p][2]);
                        acc[p][3] = fma_mix_hi(win[p + dx][1], w[3], acc[p][3]);
                    }
                }
            }
#pragma unroll
            for (int p = 0; p < PX7; ++p) {
                half4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = (half_t)silu_f(acc[p][e]);
                *reinterpret_cast<half4*>(sS1 + (py * TW + PX7 * gx + p) * LDS1 + cq * 4) = o;
            }
#pragma unroll
            for (int i = 0; i < NLD; ++i) {
                const int id = tid + 256 * i;
                const int pc = id % NP1, px = id / NP1;
                *reinterpret_cast<half8*>(sP2 + px * LDP2 + pc * 8) = p2v[i];
            }
        }
        lds_barrier();

        // ---- phase 4: conv4 (1x1, C -> C) on the matrix pipe, SiLU, + x --------------------------------------------------
        constexpr int NCT = C / 32, MTW = NPX / 128;  // cout tiles; 32-pixel tiles per wave
        f32x16 acc[NCT][MTW];
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
            for (int mt = 0; mt < MTW; ++mt) acc_bias(acc[ct][mt], a.b4 + ct * 32, lh);
#pragma unroll
        for (int ks = 0; ks < C / 16; ++ks) {
            const int k = 16 * ks + 8 * lh;  // this lane's 8 channels of the concat [conv3_out | q2 | p2]
            const half_t* bb = k < CQ ? sS1 + k : (k < CH ? sQ2 + (k - CQ) : sP2 + (k - CH));
            const int bld = k < CQ ? LDS1 : (k < CH ? LDQ2 : LDP2);
            half8 afr[NCT], bfr[MTW];
#pragma unroll
            for (int mt = 0; mt < MTW; ++mt) bfr[mt] = *reinterpret_cast<const half8*>(bb + ((wave * MTW + mt) * 32 + lrow) * bld);
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) afr[ct] = *reinterpret_cast<const half8*>(sW4 + (ct * 32 + lrow) * LDW + k);
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
                for (int mt = 0; mt < MTW; ++mt) acc[ct][mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(afr[ct], bfr[mt], acc[ct][mt], 0, 0, 0);
        }
        lds_barrier();  // every wave has read its operands: sA becomes the output tile [pixel][C + 8]
        half_t* sO = sA;
#pragma unroll
        for (int mt = 0; mt < MTW; ++mt) {
            const int prow = (wave * MTW + mt) * 32 + lrow;
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4 tv = silu4_f(f32x4{acc[ct][mt][4 * g], acc[ct][mt][4 * g + 1], acc[ct][mt][4 * g + 2], acc[ct][mt][4 * g + 3]});
                    half4 o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = (half_t)tv[e];
                    *reinterpret_cast<half4*>(sO + prow * LDO + ct * 32 + 8 * g + 4 * lh) = o;
                }
        }
        // rows [wave * RW, (wave + 1) * RW) were written by this wave only; a wave's LDS operations execute in order
#pragma unroll
        for (int i = 0; i < NOUT; ++i) {
            const int id = lane + 64 * i;
            const int row = wave * RW + id / PCS, pc = id % PCS;
            const int y = oy0 + row / TW, x = ox0 + row % TW;
            half8 v = *reinterpret_cast<const half8*>(sO + row * LDO + pc * 8);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = (half_t)((float)v[e] + (float)xv[i][e]);
            if (y < H && x < W) *reinterpret_cast<half8*>(a.dst + ((size_t)(n * H + y) * W + x) * a.ldd + pc * 8) = v;
        }
    }
}

template <int C, int TH, int TW>
int launch_tail(const PmsfaArgs& a, hipStream_t s) {
    const int tiles_x = (a.W + TW - 1) / TW, tiles_y = (a.H + TH - 1) / TH;
    const long long ntiles = (long long)a.B * tiles_x * tiles_y;
    if (ntiles <= 0 || ntiles >= 0x7fffffffLL) BSY_FAIL(BSY_ERR_ARG, "pmsfa_tail: bad grid");
    // persistent workgroups (the weights are staged once, the next tile's patch is fetched under the current tile's arithmetic): as many
    // as are resident at once -- two per CU at 71 / 76 KiB of LDS and <= 256 registers
    const int resident = 256 * 2;  // MI355X: 256 CUs
    const unsigned grid = (unsigned)(ntiles < resident ? ntiles : resident);
    const unsigned spanP = (unsigned)((((long long)a.B * a.H * a.W - 1) * a.ldp + C / 2) * 2);  // p1 = the first half of every pixel of P
    hipLaunchKernelGGL((pmsfa_tail_kernel<C, TH, TW>), dim3(grid), dim3(256), 0, s, a, tiles_x, tiles_y, (int)ntiles, spanP);
    HIP_TRY(hipGetLastError());
    return BSY_OK;
}

}  // namespace

bool pmsfa_tail_supported(int C) { return C == 32 || C == 64; }

int launch_pmsfa_tail(const PmsfaArgs& a, hipStream_t s) {
    if (!a.P || !a.x || !a.dst || !a.w2 || !a.b2 || !a.w3 || !a.b3 || !a.w4 || !a.b4) BSY_FAIL(BSY_ERR_ARG, "pmsfa_tail: null pointer");
    if (!pmsfa_tail_supported(a.C)) BSY_FAIL(BSY_ERR_ARG, "pmsfa_tail: width %d not supported (32, 64)", a.C);
    if (a.B <= 0 || a.H <= 0 || a.W <= 0) BSY_FAIL(BSY_ERR_ARG, "pmsfa_tail: empty");
    if ((a.ldp & 7) || (a.ldx & 7) || (a.ldd & 7) || a.ldp < a.C || a.ldx < a.C || a.ldd < a.C || (a.wld2 & 3) || (a.wld3 & 3) || a.wld2 < a.C / 2 ||
        a.wld3 < a.C / 4 || (a.kpad4 & 7) || a.kpad4 < a.C)
        BSY_FAIL(BSY_ERR_ARG, "pmsfa_tail: strides must be multiples of 8 channels and cover the views");
    if (((uintptr_t)a.P & 15) || ((uintptr_t)a.x & 15) || ((uintptr_t)a.dst & 15) || ((uintptr_t)a.w2 & 15) || ((uintptr_t)a.b2 & 15) ||
        ((uintptr_t)a.w3 & 15) || ((uintptr_t)a.b3 & 15) || ((uintptr_t)a.w4 & 15) || ((uintptr_t)a.b4 & 15))
        BSY_FAIL(BSY_ERR_ARG, "pmsfa_tail: pointers must be 16-byte aligned");
    if ((long long)a.B * a.H * a.W * a.ldp >= (1LL << 31)) BSY_FAIL(BSY_ERR_ARG, "pmsfa_tail: source view exceeds 2^31 elements (split the batch)");
    if (a.dst == a.x || a.dst == a.P) BSY_FAIL(BSY_ERR_ARG, "pmsfa_tail: the output must not alias an input (tiles read their neighbours' halo)");
    if (a.C == 32) return launch_tail<32, 16, 16>(a, s);
    return launch_tail<64, 8, 16>(a, s);
}
