// Fused stem: image BCHW (f16 / f32) -> Conv 3x3 s2 (3 -> C0) + SiLU -> Conv 3x3 s2 (C0 -> C1) + SiLU -> NHWC f16.
//
// Replaces model.0 + model.1 (both Conv.forward_fuse, nn/modules/conv.py:149-151; layers 0 and 1 of
// cfg/models/11/yolo11.yaml and cfg/models/v8/yolov8.yaml).  Unfused, layer 0's output (B x H/2 x W/2 x C0 f16, the
// largest activation of the graph: 419 MB at B = 64, 640 x 640, C0 = 32) is written to HBM and read straight back:
// 840 MB of the ~10 GB a forward moves, and the two launches took 0.36 ms of 3.9.  Here it never leaves the CU:
//   * a workgroup (4 waves) owns a 4 x 16 tile of layer-1 pixels.  It needs a 9 x 33 patch of layer-0 pixels, which
//     needs a 19 x 67 x 3 image patch (68 columns are loaded so that every piece is an aligned group of 4 pixels);
//   * layer 0 runs on v_mfma_f32_32x32x16_f16 exactly like conv_first.hip (image_conv.h: pixel-interleaved patch with a
//     zero 4th channel, K = 36 -> 48, a lane's B fragment = two 8-byte LDS reads), bias + SiLU, rounds to f16 -- the same values the unfused path stores -- and parks the
//     patch in LDS ([row][column parity][column / 2][C0], 16-byte chunks XOR-swizzled).  Patch positions outside the
//     layer-0 map are layer 1's zero padding and are stored as zeros;
//   * layer 1 reads its B fragments straight from that patch (stride-2 taps = consecutive entries of one parity
//     plane) and keeps its weights -- the A operand, 9*C0/16 fragments per wave -- in registers for the whole launch;
//   * workgroups are persistent (grid = 3 per CU) and fetch the next tile's image patch into registers before they
//     start computing the current one, so HBM latency hides under the MFMA / LDS work; stores go through an LDS tile
//     as coalesced 16-byte pieces.
// Accumulation order (K ascending in 16-wide MFMA steps, k = (kh, kw, c)) equals conv_first.hip's and conv_mfma.hip's,
// so the result is bit-identical to the unfused pair (tests/test_gpu_parity.py::test_stem_fused_matches_unfused).
#include "image_conv.h"

#define ST_TH 4
#define ST_TW 16
#define ST_R0 (2 * ST_TH + 1)   // 9 layer-0 rows
#define ST_C0W (2 * ST_TW + 1)  // 33 layer-0 columns
#define ST_NE (ST_R0 * ST_C0W)  // 297 layer-0 patch entries
#define ST_NMT ((ST_NE + 31) / 32)  // 10 MFMA pixel tiles of layer 0
#define ST_EV (ST_TW + 1)       // 17 even-parity columns per patch row (then 16 odd ones)
#define ST_IR (2 * ST_R0 + 1)   // 19 image rows
#define ST_NG 17                // 4-pixel groups per image row (68 columns, first = image column 4*ox0 - 4)
#define ST_ROWPX (4 * ST_NG)    // 68 patch pixels per row (8 bytes each: 3 channels + zero)
#define ST_NITEM (ST_IR * ST_NG)  // 323 load items (4 px x 3 ch) per tile
#define ST_NLOAD ((ST_NITEM + 255) / 256)

struct StemK {
    const void* img;
    const half_t* w0;
    const float* b0;
    const half_t* w1;
    const float* b1;
    half_t* dst;
    int B, H, W, OH0, OW0, OH1, OW1, ldd, Kpad1, act, tiles_x, tiles_y, ntiles;
    unsigned magic_x, magic_y;  // ceil(2^32 / tiles_x), ceil(2^32 / tiles_y): exact quotients for tile < 2^32 / divisor
};

// f32 images: the prefetched patch items are twice as wide; at three workgroups per CU (168 VGPRs) the kernel spilled 13 registers -> two
template <typename T, int C0, int C1>
__global__ __launch_bounds__(256, sizeof(T) == 4 ? 2 : 3) void stem_fused_kernel(const StemK p) {
    constexpr int NCH = C0 / 8;               // 16-byte chunks per layer-0 patch entry
    constexpr int KS1 = (9 * C0 + 15) / 16;   // layer-1 K sub-steps (16 wide)
    constexpr int NH = C1 / 32;               // layer-1 cout tiles; waves 0 .. 2*NH-1 compute layer 1
    constexpr int LDO = C1 + 8;               // padded output-tile row (halves)
    constexpr int IMG = ST_IR * ST_ROWPX * 4;
    constexpr int MID = ST_NE * C0;
    constexpr int OUT = ST_TH * ST_TW * LDO;
    static_assert(C0 == 16 || C0 == 32, "layer-0 width");
    static_assert(NH == 1 || NH == 2, "layer-1 width");
    __shared__ __attribute__((aligned(16))) half_t lds[IMG + MID + OUT + 2 * (C0 + C1)];
    half_t* simg = lds;
    half_t* smid = lds + IMG;
    half_t* sout = smid + MID;
    float* sb0 = reinterpret_cast<float*>(sout + OUT);  // biases: LDS reads instead of a global load + wait per tile
    float* sb1 = sb0 + C0;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lrow = lane & 31, lh = lane >> 5;
    const T* img = reinterpret_cast<const T*>(p.img);

    // ---- launch-invariant operands and per-lane tables -------------------------------------------------------------
    if (tid < C0) sb0[tid] = p.b0[tid];
    if (tid < C1) sb1[tid] = p.b1[tid];
    half8 a0[IMGC_KSUB];  // layer-0 weights: rows = couts (C0 <= 32 -> one tile), [CoutPad][64] packed, k = (kh, kw, c4)
#pragma unroll
    for (int s = 0; s < IMGC_KSUB; ++s)
        a0[s] = *reinterpret_cast<const half8*>(p.w0 + (size_t)lrow * IMGC_KROW + 16 * s + 8 * lh);
    const int mt1 = wave & 1, nh1 = wave >> 1;  // layer-1 role: pixel tile (2 rows x 16) and cout tile
    const bool l1 = nh1 < NH;
    half8 a1[KS1];  // layer-1 weights, [CoutPad][Kpad1] packed, k = (kh, kw, c)
#pragma unroll
    for (int ks = 0; ks < KS1; ++ks)
        a1[ks] = l1 ? *reinterpret_cast<const half8*>(p.w1 + (size_t)(nh1 * 32 + lrow) * p.Kpad1 + 16 * ks + 8 * lh)
                    : half8{0, 0, 0, 0, 0, 0, 0, 0};
    // load items of this thread: (image row r, group j) -> patch position; tile-independent
    int it_r[ST_NLOAD], it_j[ST_NLOAD];
#pragma unroll
    for (int i = 0; i < ST_NLOAD; ++i) {
        const int idx = tid + 256 * i;
        it_r[i] = idx / ST_NG;
        it_j[i] = idx - it_r[i] * ST_NG;
    }
    const int e1base = (2 * (2 * mt1 + (lrow >> 4))) * ST_C0W + (lrow & 15);  // layer-1 lane pixel -> patch entry of tap (0, 0)

    auto tile_origin = [&](int tile, int& n, int& oy0, int& ox0) {  // divisions by multiply-high (host-made magics)
        const int r = (int)__umulhi((unsigned)tile, p.magic_x);
        const int tx = tile - r * p.tiles_x;
        n = (int)__umulhi((unsigned)r, p.magic_y);
        const int ty = r - n * p.tiles_y;
        oy0 = ty * ST_TH;
        ox0 = tx * ST_TW;
    };
    ImgItem<T> pre[ST_NLOAD];
    int nn = 0, noy0 = 0, nox0 = 0;  // coordinates of the tile whose patch is in `pre`
    auto fetch = [&](int tile) {
        tile_origin(tile, nn, noy0, nox0);
        const int n = nn, oy0 = noy0, ox0 = nox0;
        const T* ip = img + (size_t)n * 3 * p.H * p.W;
#pragma unroll
        for (int i = 0; i < ST_NLOAD; ++i)
            pre[i] = img_item_load<T>(ip, p.H, p.W, 4 * oy0 - 3 + it_r[i], 4 * ox0 - 4 + 4 * it_j[i], tid + 256 * i < ST_NITEM);
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
        for (int i = 0; i < ST_NLOAD; ++i)
            if (tid + 256 * i < ST_NITEM) img_item_park<T>(pre[i], simg + (it_r[i] * ST_ROWPX + 4 * it_j[i]) * 4);
        lds_barrier();  // image patch visible; every wave is done with the previous tile's LDS
        if (tile + tw.step < tw.end) fetch(tile + tw.step);

        // ---- layer 0: 297 patch pixels = 10 MFMA pixel tiles, dealt round-robin to the 4 waves ----------------------
        const int r00 = 2 * oy0 - 1, c00 = 2 * ox0 - 1;  // layer-0 coordinates of patch entry (0, 0)
        for (int mt = wave; mt < ST_NMT; mt += 4) {
            const int pp = mt * 32 + lrow;
            const int pc = pp < ST_NE ? pp : ST_NE - 1;
            const int r = pc / ST_C0W, c = pc - r * ST_C0W;
            const int win = (2 * r * ST_ROWPX + 2 * c + 1) * 4;  // window origin: image-patch pixel (2r, 2c + 1)
            f32x16 acc;
            acc_bias(acc, sb0, lh);  // accumulators start at the bias (common.h acc_bias)
#pragma unroll
            for (int s = 0; s < IMGC_KSUB; ++s)
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0[s], img_frag<ST_ROWPX>(simg, win, s, lh), acc, 0, 0, 0);
            // bias + SiLU -> f16; positions outside the layer-0 map are layer 1's zero padding
            // (an AND with an all-ones / all-zeros word: a select on `inside` makes the compiler branch around each SiLU)
            const unsigned keep = ((unsigned)(r00 + r) < (unsigned)p.OH0 && (unsigned)(c00 + c) < (unsigned)p.OW0) ? 0xffffffffu : 0u;
            const int e = r * ST_C0W + (c & 1) * ST_EV + (c >> 1);
            if (pp < ST_NE) {
#pragma unroll
                for (int g = 0; g < NCH; ++g) {  // couts 8g + 4lh .. +3  (C0 = 16: g < 2 only)
                    union { half4 h; unsigned u[2]; } o;
                    f32x4 t = f32x4{acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]};
                    if (p.act) t = silu4_f(t);
#pragma unroll
                    for (int q = 0; q < 4; ++q) o.h[q] = (half_t)t[q];
                    o.u[0] &= keep;
                    o.u[1] &= keep;
                    *reinterpret_cast<half4*>(smid + (e * NCH + (g ^ ((e >> 2) & (NCH - 1)))) * 8 + 4 * lh) = o.h;
                }
            }
        }
        lds_barrier();  // layer-0 patch complete

        // ---- layer 1: wave = (pixel tile of 2 rows x 16, cout tile of 32) --------------------------------------------
        if (l1) {
            f32x16 acc;
            acc_bias(acc, sb1 + nh1 * 32, lh);
#pragma unroll
            for (int ks = 0; ks < KS1; ++ks) {
                // k = 16 ks + 8 lh + (0..7) = tap * C0 + channel: one tap and one 8-channel chunk per lane half
                const int k0 = 16 * ks, k1 = k0 + 8;
                const int t0 = k0 / C0, t1 = k1 / C0;  // compile-time
                const int ch0 = (k0 % C0) / 8, ch1 = (k1 % C0) / 8;
                const int e0 = t0 < 9 ? (t0 / 3) * ST_C0W + ((t0 % 3) & 1) * ST_EV + ((t0 % 3) >> 1) : 0;
                const int e1 = t1 < 9 ? (t1 / 3) * ST_C0W + ((t1 % 3) & 1) * ST_EV + ((t1 % 3) >> 1) : 0;
                const int e = e1base + (e0 == e1 ? e0 : (lh ? e1 : e0));
                const int ch = lh ? ch1 : ch0;
                half8 bf = *reinterpret_cast<const half8*>(smid + (e * NCH + (ch ^ ((e >> 2) & (NCH - 1)))) * 8);
                if (t1 >= 9 && (t0 >= 9 || lh)) bf = half8{0, 0, 0, 0, 0, 0, 0, 0};  // K padding (C0 = 16: k >= 144)
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1[ks], bf, acc, 0, 0, 0);
            }
            const int prow = mt1 * 32 + lrow;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int c = nh1 * 32 + 8 * g + 4 * lh;
                half4 o;
                f32x4 t = f32x4{acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]};
                if (p.act) t = silu4_f(t);
#pragma unroll
                for (int q = 0; q < 4; ++q) o[q] = (half_t)t[q];
                *reinterpret_cast<half4*>(sout + prow * LDO + c) = o;
            }
        }
        lds_barrier();  // output tile complete

        constexpr int CPRW = C1 / 8;
#pragma unroll
        for (int id = tid; id < ST_TH * ST_TW * CPRW; id += 256) {
            const int prow = id / CPRW, cc = (id % CPRW) * 8;
            const int oy = oy0 + prow / ST_TW, ox = ox0 + prow % ST_TW;
            if (oy < p.OH1 && ox < p.OW1)
                *reinterpret_cast<half8*>(p.dst + ((size_t)(n * p.OH1 + oy) * p.OW1 + ox) * p.ldd + cc) =
                    *reinterpret_cast<const half8*>(sout + prow * LDO + cc);
        }
    }
}

bool stem_fused_supported(int C0, int C1, int H, int W) {
    return ((C0 == 32 && C1 == 64) || (C0 == 16 && C1 == 32)) && !(W & 3) && H >= 4 && W >= 4;
}

template <typename T>
static int launch_t(const StemArgs& a, const StemK& k, hipStream_t s) {
    const int slots = sizeof(T) == 4 ? 512 : 768;      // 3 workgroups per CU (register-limited: a fourth needs <= 128 VGPRs and spills 61; f32 images: 2), tiles dealt round-robin
    const int grid = k.ntiles < slots ? k.ntiles : slots;
    if (a.C0 == 32) hipLaunchKernelGGL((stem_fused_kernel<T, 32, 64>), dim3(grid), dim3(256), 0, s, k);
    else hipLaunchKernelGGL((stem_fused_kernel<T, 16, 32>), dim3(grid), dim3(256), 0, s, k);
    HIP_TRY(hipGetLastError());
    return BSY_OK;
}

int launch_stem_fused(const StemArgs& a, hipStream_t s) {
    if (!stem_fused_supported(a.C0, a.C1, a.H, a.W))
        BSY_FAIL(BSY_ERR_ARG, "stem: unsupported shape (C0 %d, C1 %d, H %d, W %d): need (32,64) or (16,32) channels, W %% 4 == 0", a.C0, a.C1, a.H, a.W);
    const int OH0 = (a.H - 1) / 2 + 1, OW0 = (a.W - 1) / 2 + 1;
    const int OH1 = (OH0 - 1) / 2 + 1, OW1 = (OW0 - 1) / 2 + 1;
    if (a.OH != OH1 || a.OW != OW1) BSY_FAIL(BSY_ERR_ARG, "stem: output extent mismatch (%d x %d, expected %d x %d)", a.OH, a.OW, OH1, OW1);
    const int esz = a.img_dtype == BSY_F32 ? 4 : 2;
    if (((uintptr_t)a.img & (4 * esz - 1)) || ((uintptr_t)a.dst & 15) || (a.ldd & 7) || ((uintptr_t)a.w0 & 15) ||
        ((uintptr_t)a.w1 & 15) || ((uintptr_t)a.b0 & 15) || ((uintptr_t)a.b1 & 15) || (a.Kpad1 & 7) || a.Kpad1 < 9 * a.C0 || a.Kpad1 < 16 * ((9 * a.C0 + 15) / 16))
        BSY_FAIL(BSY_ERR_ARG, "stem: misaligned pointer / leading dimension");
    StemK k;
    k.img = a.img; k.w0 = (const half_t*)a.w0; k.b0 = a.b0; k.w1 = (const half_t*)a.w1; k.b1 = a.b1; k.dst = a.dst;
    k.B = a.B; k.H = a.H; k.W = a.W; k.OH0 = OH0; k.OW0 = OW0; k.OH1 = OH1; k.OW1 = OW1; k.ldd = a.ldd; k.Kpad1 = a.Kpad1;
    k.act = a.act;
    k.tiles_x = ceil_div(OW1, ST_TW); k.tiles_y = ceil_div(OH1, ST_TH);
    const long long nt = (long long)a.B * k.tiles_x * k.tiles_y;
    if (nt <= 0 || nt > 0x7fffffffLL) BSY_FAIL(BSY_ERR_ARG, "stem: tile count out of range");
    k.ntiles = (int)nt;
    if (k.tiles_x < 2 || k.tiles_y < 2) {
        // magic for divisor 1 would be 2^32: keep the kernel's multiply-high path by padding the tile grid instead
        if (k.tiles_x < 2) k.tiles_x = 2;
        if (k.tiles_y < 2) k.tiles_y = 2;
        k.ntiles = a.B * k.tiles_x * k.tiles_y;  // the extra tiles lie outside the map: nothing is stored for them
    }
    if ((long long)k.ntiles * (k.tiles_x > k.tiles_y ? k.tiles_x : k.tiles_y) >= (1LL << 32))
        BSY_FAIL(BSY_ERR_ARG, "stem: tile count out of range");
    k.magic_x = (unsigned)(((1ULL << 32) + k.tiles_x - 1) / k.tiles_x);
    k.magic_y = (unsigned)(((1ULL << 32) + k.tiles_y - 1) / k.tiles_y);
    if (a.img_dtype == BSY_F16) return launch_t<half_t>(a, k, s);
    if (a.img_dtype == BSY_F32) return launch_t<float>(a, k, s);
    BSY_FAIL(BSY_ERR_ARG, "stem: image dtype %d unsupported", a.img_dtype);
}
