// Shared declarations for the gfx950 kernels of libbsyolo_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string>

#include "../../include/bsyolo.h"

typedef _Float16 half_t;
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// ---- error plumbing (thread-local message, no C++ exceptions across the ABI) ----
void bsy_set_error(const char* fmt, ...);
#define BSY_FAIL(code, ...)        \
    do {                           \
        bsy_set_error(__VA_ARGS__); \
        return (code);             \
    } while (0)
#define HIP_TRY(expr)                                                                                     \
    do {                                                                                                  \
        hipError_t _e = (expr);                                                                           \
        if (_e != hipSuccess) BSY_FAIL(BSY_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), \
                                       __FILE__, __LINE__);                                               \
    } while (0)

static inline int ceil_div(int a, int b) { return (a + b - 1) / b; }
static inline int round_up(int a, int b) { return ceil_div(a, b) * b; }

// ---- resolved (pointer-level) launch arguments; filled by the engine or by the stand-alone C entry points ----
struct ConvArgs {
    const half_t* src0;  // incl. channel offset
    const half_t* src1;  // second concat operand or nullptr
    int ld0, ld1;        // channels per pixel row of each source buffer
    int C0, C1;          // channels taken from each source (Cin = C0 + C1), multiples of 8
    int up0, up1;        // nearest-x2 read
    int B, H, W, OH, OW; // logical input / output extents
    int ksize, stride, pad;
    const half_t* wgt;   // [CoutPad][Kpad]
    const float* bias;   // [CoutPad]
    void* dst;           // incl. channel offset; f16 or f32
    int ldd, Cout, out_f32;
    const half_t* res;   // incl. channel offset or nullptr
    int ldr;
    int act;
    int dst_scale, dst_dy, dst_dx;
    int cfg;  // kernel configuration id (tile << 4 | variant) chosen by the autotuner; < 0 = heuristic
    // Fused Detect decoder (epi != 0: the conv is the last layer of a head branch and writes the decoded prediction tensor
    // y (B, nrows, A) directly instead of a logit map; `dst` is then unused):
    //   epi 2: class scores  y[n, 4 + c, a0 + pix] = sigmoid(logit)                 (head.py:147)
    //   epi 3: boxes         y[n, 0..3, a0 + pix]  = dist2bbox(DFL(logits)) * stride (head.py:141-146, block.py:58-77)
    // raw (optional, may be null): the level's raw map (B, rawC, OH, OW) that Detect.forward also returns (head.py:74),
    // box logits at channels 0..63, class logits at 64...
    int epi;
    void* y;
    int y_f32, A, a0, nrows;
    float lvl_stride;
    void* raw;
    int raw_f32, rawC;
    // Box-branch tail (3x3 conv + the branch's last 1x1 conv + DFL decode in one launch; conv_mfma.hip conv3x3_patch_kernel<.., TAIL>):
    // this op is the 3x3 conv (Cout = 64, its map never reaches HBM), tail_wgt / tail_bias the packed 64 -> 64 1x1 conv, epi = 3 and
    // the decoder fields above describe what the tail writes.  nullptr: no tail.
    const half_t* tail_wgt;
    const float* tail_bias;
    // Split-K (latency mode): nsl_c channel slices x nsl_t tap slices of the layer's K walk, computed by separate workgroups of one launch
    // into f32 slabs of split_ws (nsl_c * nsl_t * B * OH * OW * round_up(Cout, 32) floats) and summed in slice order by a second
    // launch.  0 / 1 x 0 / 1: off.  Implicit-GEMM kernel only (conv_cfg_valid).
    int nsl_c = 0, nsl_t = 0;
    float* split_ws = nullptr;
};
int launch_conv(const ConvArgs& a, hipStream_t s);
#define BSY_CONV_MAX_CFG 64
int conv_candidates(const ConvArgs& a, int* out, int max_out);  // valid configuration ids, heuristic best first
bool conv_cfg_valid(const ConvArgs& a, int cfg);
int conv_korder(const ConvArgs& a);  // the layer's K walk (0 packed order, 1 / 2 chunk-major with 32- / 64-channel chunks): a function of its shape

struct ConvFirstArgs {
    const void* img;
    int img_dtype;
    int B, H, W, OH, OW, ksize, stride, pad;
    const void* w;   // packed f16 [CoutPad][32], k = (kh, kw, c)  (same layout as every other conv)
    const float* b;  // [CoutPad]
    half_t* dst;
    int ldd, Cout, act;
};
int launch_conv_first(const ConvFirstArgs& a, hipStream_t s);

// Fused stem (stem_fused.hip): image -> Conv 3x3 s2 (3 -> C0) -> Conv 3x3 s2 (C0 -> C1), layer 0 kept on chip.
struct StemArgs {
    const void* img;
    int img_dtype;
    int B, H, W, OH, OW;    // image extent; OH/OW = layer-1 output extent
    const void* w0;         // layer 0: packed f16 [CoutPad][32]
    const float* b0;
    const void* w1;         // layer 1: packed f16 [CoutPad][Kpad1], k = (kh, kw, c)
    const float* b1;
    int C0, C1, Kpad1;
    half_t* dst;
    int ldd, act;
};
bool stem_fused_supported(int C0, int C1, int H, int W);
int launch_stem_fused(const StemArgs& a, hipStream_t s);

// Fused Bottleneck (bneck_fused.hip): y = x + conv3x3(conv3x3(x)), hidden map kept on chip.
struct BneckArgs {
    const half_t* src;  // (B,H,W,lds) view of C channels
    int lds;
    int B, H, W, C, CH;
    const void *w1, *w2;   // packed f16 [CoutPad][Kpad1] (C -> CH) and [CoutPad][Kpad2] (CH -> C)
    const float *b1, *b2;
    int Kpad1, Kpad2;
    half_t* dst;
    int ldd, act;
};
bool bneck_fused_supported(int C, int CH);
int launch_bneck_fused(const BneckArgs& a, hipStream_t s);

// Fused C3k2 block (c3k2_fused.hip): out = cv2(cat(y0, y1, y1 + m.cv2(m.cv1(y1)))), [y0 | y1] = cv1(x); nothing but x and out in HBM.
struct C3k2Args {
    const half_t* src;  // (B,H,W,lds) view of Cin channels
    int lds;
    int B, H, W, Cin, C, C2;           // C = hidden width c of the block (cv1 -> 2c), bottleneck hidden c/2, output C2
    const void *w1, *wa, *wb, *w4;     // packed f16 [CoutPad][Kpad]: cv1 (Cin -> 2c), m.cv1 (c -> c/2, 3x3), m.cv2 (c/2 -> c, 3x3), cv2 (3c -> C2)
    const float *b1, *ba, *bb, *b4;
    half_t* dst;
    int ldd;
};
bool c3k2_fused_supported(int Cin, int C, int C2);
int launch_c3k2_fused(const C3k2Args& a, hipStream_t s);

// Two chained 1x1 convs in one launch (chain1x1.hip): d2 = act2(W2 [h2 | keep(s1)] + b2) (+ r2), s1 = act1(W1 [a0 | a1] + b1) (+ r1);
// s1 goes to d1 when set, its couts [keep0, keep0 + LC) stay in LDS as the LAST LC input channels of the second conv.
struct ChainArgs {
    const half_t *a0, *a1;     // stage-1 sources incl. channel offset (a1: second concat operand or nullptr)
    int lda0, lda1, CA0, CA1;
    const half_t* w1;          // packed [N1 pad 128][K1 pad 32]
    const float* b1;
    int N1, act1;
    half_t* d1;                // stage-1 output view or nullptr
    int ldd1;
    const half_t* r1;          // shortcut operand of stage 1 or nullptr
    int ldr1;
    int keep0, LC;
    const half_t* h2;          // stage-2 K part read from HBM (first CH2 input channels) or nullptr
    int ldh2, CH2;
    const half_t* w2;
    const float* b2;
    int N2, act2;
    half_t* d2;
    int ldd2;
    const half_t* r2;
    int ldr2;
    long long M;               // pixels
};
bool chain_supported(int CA0, int CA1, int N1, int keep0, int LC, int CH2, int N2);
int launch_chain(const ChainArgs& a, hipStream_t s);

// ---- BS-YOLO-only modules (bsyolo_ops.hip) ----------------------------------------------------------------------------
struct DwGenArgs {       // depthwise kh x kw, stride 1 / 2, "same" padding, + bias (+SiLU)
    const half_t* src;
    int lds;
    int B, H, W, C, OH, OW, kh, kw, stride;
    const float* w;      // [kh*kw][wld] (already offset to this launch's first channel)
    int wld;
    const float* b;
    half_t* dst;
    int ldd, act_c;      // SiLU on channels [0, act_c) of this launch, identity on the rest
    int ident_c0 = 0;    // > 0: channels [ident_c0, C) carry an identity kernel (centre tap 1, bias 0: PMSFA.conv3's pass-through half,
                         // weights.py "dwg_ext") -- the tiled kernels write x + 0 for them, which is what the full tap loop returns
};
int launch_dwconv_generic(const DwGenArgs& a, hipStream_t s);
struct PmsfaArgs {       // PMSFA's tail in one launch (pmsfa_fused.hip): dst = conv4([conv3(conv2(p1)[:C/4]) | conv2(p1)[C/4:] | p2]) + x, P = [p1 | p2] = conv1(x)
    const half_t* P;     // conv1's output, C channels
    int ldp;
    const half_t* x;     // the module's input (shortcut operand), C channels
    int ldx;
    half_t* dst;
    int ldd;
    int B, H, W, C;
    const float* w2;     // conv2: depthwise 5x5 on C/2 channels, f32 [25][wld2]
    const float* b2;
    int wld2;
    const float* w3;     // conv3: depthwise 7x7 on C/4 channels, f32 [49][wld3]
    const float* b3;
    int wld3;
    const half_t* w4;    // conv4: packed 1x1 weights [cout][kpad4]
    const float* b4;
    int kpad4;
};
bool pmsfa_tail_supported(int C);
int launch_pmsfa_tail(const PmsfaArgs& a, hipStream_t s);
int launch_copy_view(const half_t* src, int lds_, int up, int B, int H, int W, int C, half_t* dst, int ldd, hipStream_t s);
int launch_gap(const half_t* src, int lds_, int B, int H, int W, int C, half_t* out, int ldo, hipStream_t s);
struct MscaSpArgs {
    const half_t* src;
    int lds, B, H, W, C;
    const float* w[9];  // conv0, conv0_1, conv0_2, conv1_1, conv1_2, conv2_1, conv2_2, conv3_1, conv3_2
    const float* b[9];
    half_t* br[4];
    int ldb[4];
    half_t* gap[4];
    int ldg[4];
};
bool msca_spatial_supported(int H, int W);
int launch_msca_spatial(const MscaSpArgs& a, hipStream_t s);
struct MixArgs {
    const half_t* br[4];
    int ldb[4];
    const float* lg[4];  // branch logits (B, ldl) f32
    int ldl[4];
    int B, HW, C;
    half_t* dst;
    int ldd;
};
int launch_msca_mix(const MixArgs& a, hipStream_t s);
int launch_mul(const half_t* a, int lda, const half_t* b, int ldb, long long npix, int C, half_t* dst, int ldd, hipStream_t s);
struct ElaArgs {
    const half_t* src;
    int lds;
    int B, H, W, C, k;
    const float *wsp, *wch, *gnw, *gnb;  // spatial_conv [C][k], ch_att conv [C][k], GroupNorm weight / bias [C]
    float ch_coef, sp_coef, res_coef;    // sigmoid(ch_weight), sigmoid(sp_weight), sigmoid(res_weight)
    float* scratch;                      // ela_scratch_floats(H, W, C) * B floats
    half_t* dst;
    int ldd;
};
size_t ela_scratch_floats(int H, int W, int C);
int launch_ela(const ElaArgs& a, hipStream_t s);
int launch_ela_gate(const ElaArgs& a, hipStream_t s);

// ---- fp32 correctness mode (ref32.hip): plans with bsy_op.prec == 1; NHWC f32 views --------------------------------------------
struct Conv32Args {
    const void* src0;    // f32 view, or (first) the BCHW image in src_dtype
    const void* src1;
    int src_dtype, first;
    int ld0, ld1, C0, C1, up0, up1;
    int B, H, W, OH, OW, ks, stride, pad;
    const float* w;      // f32 [k*k*Cin][Cout], k = (kh, kw, cin)
    const float* bias;   // f32 [Cout]
    float* dst;
    int ldd, Cout;
    const float* res;
    int ldr, act, dst_scale, dst_dy, dst_dx;
    // fp32x mode (conv32x_mfma.hip): the same weights split into two f16 planes [Cout][wx_kpad], w = hi + lo, K padded to 32
    // with zeros; nullptr = the exact fp32 kernels only
    const half_t* wx_hi;
    const half_t* wx_lo;
    int wx_kpad;
};
int launch_conv32(const Conv32Args& a, hipStream_t s);          // routes to the MFMA kernel where it applies, else the scalar one
bool conv32x_mfma_supported(const Conv32Args& a);               // conv32x_mfma.hip: split-f16 operands, three f16 MFMAs per product
int launch_conv32x_mfma(const Conv32Args& a, hipStream_t s);
int launch_conv32_scalar(const Conv32Args& a, hipStream_t s);   // ref32.hip: one thread per output, sequential fmaf chain
bool conv32_mfma_supported(const Conv32Args& a);                // conv32_mfma.hip: v_mfma_f32_32x32x2_f32, the same chain bit for bit
int launch_conv32_mfma(const Conv32Args& a, hipStream_t s);
struct Dw32Args {
    const float* src;
    int lds, B, H, W, C, OH, OW, kh, kw, stride;
    const float* w;      // f32 [kh*kw][wld]
    int wld;
    const float* b;
    float* dst;
    int ldd, act_c;
    const float* res;
    int ldr;
};
int launch_dw32(const Dw32Args& a, hipStream_t s);
struct Mix32Args {
    const float* br[4];
    int ldb[4];
    const float* lg[4];
    int ldl[4];
    int B, HW, C;
    float* dst;
    int ldd;
};
int launch_mix32(const Mix32Args& a, hipStream_t s);
int launch_sppf32(float* buf, int ld, int B, int H, int W, int C, hipStream_t s);
int launch_attn32(const float* qkv, int ld, int B, int N, int heads, int kd, int hd, float scale, float* out, int ldo, hipStream_t s,
                  int impl = 0);  // impl 0: tiled kernel where it applies (key_dim 32, head_dim 64), 1: generic kernel, 2: tiled or error
bool attn32x_supported(int ld, int ldo, int kd, int hd, const void* qkv, const void* out);  // attention32x.hip (fp32x mode)
int launch_attn32x(const float* qkv, int ld, int B, int N, int heads, int kd, int hd, float scale, float* out, int ldo, hipStream_t s);
int launch_nhwc2nchw32(const float* src, int ld, int B, int C, int hw, void* out, int out_dtype, hipStream_t s);
int launch_copy32(const float* src, int lds_, int up, int B, int H, int W, int C, float* dst, int ldd, hipStream_t s);
int launch_gap32(const float* src, int lds_, int B, int H, int W, int C, float* out, int ldo, hipStream_t s);
int launch_mul32(const float* x, int ldx, const float* y, int ldy, long long npix, int C, float* dst, int ldd, hipStream_t s);
int launch_ela32(const ElaArgs& a, const float* src, float* dst, hipStream_t s);  // a.src / a.dst unused

// Fused DWConv 3x3 (+SiLU) -> Conv 1x1 (+act) (conv_mfma.hip: dwpw_fused_kernel)
struct DwPwArgs {
    const half_t* src;
    int lds;
    int B, H, W, C;
    const float *dww, *dwb;  // depthwise f32 [9][C], [C]
    const void* wgt;         // 1x1 packed f16 [CoutPad][Kpad]
    const float* bias;
    half_t* dst;
    int ldd, Cout, act;
};
bool dwpw_fused_supported(int C, int Cout);
int launch_dwpw_fused(const DwPwArgs& a, hipStream_t s);

struct DwArgs {
    const half_t* src;
    int lds;
    int B, H, W, C;
    const float* w;  // [9][C]
    const float* b;
    half_t* dst;
    int ldd;
    int act;
    const half_t* res;
    int ldr;
};
int launch_dwconv(const DwArgs& a, hipStream_t s);
int launch_sppf_pool(half_t* buf, int ld, int B, int H, int W, int C, hipStream_t s);
int launch_s2d(const void* img, int img_dtype, int B, int H, int W, half_t* dst, int ldd, hipStream_t s);

struct AttnArgs {
    const half_t* qkv;
    int ld;
    int B, N, heads, key_dim, head_dim;
    float scale;
    half_t* out;
    int ldo;
};
int launch_attention(const AttnArgs& a, hipStream_t s);

struct DecodeArgs {
    const float* box[3];
    const float* cls[3];
    const float* msk[3];
    int ldb[3], ldc[3], ldm[3];
    int h[3], w[3];
    float stride[3];
    int nl, B, nc, nm, A;
    void* y;
    int y_dtype;
};
int launch_decode(const DecodeArgs& a, hipStream_t s);
int launch_nhwc2nchw(const half_t* src, int ld, int B, int C, int hw, void* out, int out_dtype, hipStream_t s);
int launch_raw_nchw(const float* box, int ldb, const float* cls, int ldc, int B, int h, int w, int nc, void* out,
                    int out_dtype, hipStream_t s);

// WHY these exist (round 1): the natural form `fmaf((float)h[j], w[j], a[j])` compiles to
// v_cvt_f32_f16 (SDWA WORD_1 for the odd elements) feeding SLP-packed v_pk_fma_f32 -- and that sequence returned wrong
// odd elements, sporadically, whenever MFMA kernels of another stream shared the CUs (SCDown's depthwise conv: 26 of 48
// forwards; 0 of 48 with packing off).  Depthwise kernels therefore use v_fma_mix_f32 (no conversion, no packing) and
// the whole library is built with -fno-slp-vectorize (build.py).
// acc + f32(h) * w with ONE instruction per element: v_fma_mix_f32 reads the f16 operand (low / high half of a packed
// register) directly, so a depthwise tap costs 8 VALU issues per 8 channels instead of 8 v_cvt_f32_f16 + 4 v_pk_fma_f32
// (and packed f32 math is the slower choice beside MFMAs).  Same value as fmaf((float)h, w, acc): the conversion is exact.
__device__ __forceinline__ float fma_mix_lo(unsigned hpair, float w, float acc) {
    float r;
    asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(hpair), "v"(w), "v"(acc));
    return r;
}
__device__ __forceinline__ float fma_mix_hi(unsigned hpair, float w, float acc) {
    float r;
    asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(hpair), "v"(w), "v"(acc));
    return r;
}
// element J (compile-time) of a half8 register quadruple
template <int J>
__device__ __forceinline__ float fma_mix_e(const half8& v, float w, float acc) {
    union { half8 h; unsigned u[4]; } x;
    x.h = v;
    return (J & 1) ? fma_mix_hi(x.u[J >> 1], w, acc) : fma_mix_lo(x.u[J >> 1], w, acc);
}
// a[0..7] += f32(v[0..7]) * (w0, w1)
__device__ __forceinline__ void fma_mix8(float (&a)[8], const half8& v, const f32x4& w0, const f32x4& w1) {
    union { half8 h; unsigned u[4]; } x;
    x.h = v;
    a[0] = fma_mix_lo(x.u[0], w0[0], a[0]); a[1] = fma_mix_hi(x.u[0], w0[1], a[1]);
    a[2] = fma_mix_lo(x.u[1], w0[2], a[2]); a[3] = fma_mix_hi(x.u[1], w0[3], a[3]);
    a[4] = fma_mix_lo(x.u[2], w1[0], a[4]); a[5] = fma_mix_hi(x.u[2], w1[1], a[5]);
    a[6] = fma_mix_lo(x.u[3], w1[2], a[6]); a[7] = fma_mix_hi(x.u[3], w1[3], a[7]);
}

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also waits for vmcnt(0) -- every global load and STORE the
// wave has in flight -- so in a persistent tile loop each barrier after a prefetch or behind the tile's output stores exposed a
// full HBM round trip (round 3, SQ counters of the fused kernels: 44-60 % of the wave cycles spent waiting).  Registers filled
// by a prefetch are still safe to use: the compiler places its own counted vmcnt wait in front of their first use.
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

// Tile walk of a persistent workgroup, XCD-aware.  Workgroup b of a grid of G lands on XCD b % 8 (round-robin dispatch): with the
// plain walk (tile = b, b + G, ...) the four / eight spatial neighbours of a tile run on OTHER XCDs, so the halo every fused kernel
// re-reads (the C3k2 block reads its input 2.5 times) misses the XCD's L2 each time: PMC traffic of the block 927 MB for 629 MB.
// Here XCD x owns one contiguous eighth of the tiles and its workgroups walk it side by side: at any moment an XCD works on ~G / 8
// consecutive tiles -- a few tile rows of one image -- and the halos hit its L2.
struct TileWalk {
    int tile, step, end;
};
__device__ __forceinline__ TileWalk xcd_tile_walk(int b, int G, int ntiles) {
    const int X = G < 8 ? G : 8;
    const int x = b % X, i = b / X;
    const int gx = (G - x + X - 1) / X;          // workgroups on this XCD
    const int q = ntiles / X, r = ntiles % X;
    const int start = x * q + (x < r ? x : r);
    return TileWalk{start + i, gx, start + q + (x < r ? 1 : 0)};
}

// An MFMA accumulator tile (32 couts x 32 pixels, lane = pixel) that STARTS at the bias of its rows (round 3: every conv kernel of the
// fp16 path accumulates bias + sum instead of adding the bias in its epilogue -- one VALU instruction less per activation in
// kernels whose SIMDs are ~80 % busy issuing SiLU; all kernels do it alike, so fused and un-fused launches still agree bit for bit).
// Register r of lane half lh is cout (r & 3) + 8 (r >> 2) + 4 lh of the tile; `bias` points at the tile's first cout (f32, 16-byte
// aligned, 32 readable values: the packed bias vectors are padded to a multiple of 128).
__device__ __forceinline__ void acc_bias(f32x16& acc, const float* bias, int lh) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const f32x4 bv = *reinterpret_cast<const f32x4*>(bias + 8 * g + 4 * lh);
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[4 * g + e] = bv[e];
    }
}

// SiLU in fp32: x * sigmoid(x) = x / (1 + 2^(-x*log2 e)); v_exp_f32 + v_rcp_f32 (1 ulp each) -- the result is rounded
// to fp16 right after, so the IEEE-division expansion (~10 VALU) would buy nothing.  exp2 overflow -> inf -> rcp -> 0.
// The product passes through an empty asm: where a conversion to f16 follows directly, the compiler otherwise folds the final
// multiply and the conversion into v_fma_mixlo/hi_f16 -- ONE rounding of the exact product -- in some kernels and not in others
// (it did in c3k2_fused.hip's Bottleneck stage, not in bneck_fused.hip: 2 pixels in 1000 differed by one f16 ulp).  Every
// kernel rounds twice (f32, then f16), so fused and unfused launches agree bit for bit.
__device__ __forceinline__ float silu_f(float x) {
    float r = x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x * -1.4426950408889634f));
    asm("" : "+v"(r));
    return r;
}
// a + b element by element: a vector-typed `a + b` is a pair of v_pk_add_f32, this is four v_add_f32 (the kernels that run
// beside other streams' MFMA kernels carry no packed f32 arithmetic at all; see the note on fma_mix_lo above)
__device__ __forceinline__ f32x4 add4_f(f32x4 a, f32x4 b) {
    f32x4 r;
#pragma unroll
    for (int i = 0; i < 4; ++i) r[i] = a[i] + b[i];
    return r;
}
// Four at a time, element by element (with -fno-slp-vectorize these stay scalar instructions; packed, the compiler's
// choice under plain -O3, they were not one microsecond faster over a forward).
__device__ __forceinline__ f32x4 silu4_f(f32x4 x) {
    f32x4 r;
#pragma unroll
    for (int i = 0; i < 4; ++i) r[i] = silu_f(x[i]);
    return r;
}
