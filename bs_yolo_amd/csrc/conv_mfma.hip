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
// accumulator then holds, per lane, ONE pixel and groups of 4 consecutive output channels.
//
// Tile: 256 threads = 4 waves (WAVES_M x WAVES_N), wave tile (MT*32 pixels) x (NT*32 couts), BK = 32.
// LDS rows are 64 B (32 halves) with the 16-B chunk index XOR-swizzled by (row>>2)&3 so that ds_read_b128 of 16
// consecutive rows hits 16 distinct slots of the 256-B bank row (guide T2).
//
// Staging (v2): LDS-DMA (`global_load_lds_dwordx4`, 1 KiB per wave-instruction = 16 rows x 64 B, lane-linear LDS image;
// the swizzle is applied to the per-lane SOURCE chunk, guide rule 21) into a ring of STAGES K-steps.  A conv K-step is
// only 8 MFMAs per wave (~0.1-0.3 us) while a fetch takes 1-3 us under load, so the ring keeps STAGES-1 K-steps in
// flight behind a COUNTED `s_waitcnt vmcnt(N)` and one raw `s_barrier` per K-step (never `__syncthreads()`, which would
// drain the DMA queue).  Out-of-image taps / K padding read a zero page instead of branching.
// The epilogue pairs lanes l and l+32 with v_permlane32_swap so each lane stores 16 B (8 consecutive channels).
#include <stdlib.h>

#include "common.h"

struct ConvK {
    const half_t* src0;
    const half_t* src1;
    int ld0, ld1, C0, C1, up0, up1;
    int H, W, OH, OW, stride, pad;
    int M;            // B*OH*OW
    int Cin8;         // (C0+C1)/8
    int ntaps;        // ks*ks
    int nk;           // Kpad/32 (informational; the kernel uses Kpad / BK)
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
    int ntm;  // number of pixel tiles (persistent kernel)
    int tiles_x, tiles_y, B;  // patch kernel: 8 x 16 output tiles per image
    int epi, y_f32, A, a0, nrows, raw_f32, rawC;  // fused Detect decoder (ConvArgs::epi)
    float lvl_stride;
    void* y;
    void* raw;
    const half_t* tail_wgt;  // box-branch tail (ConvArgs::tail_wgt): packed [128][64] 1x1 weights, f32 bias
    const float* tail_bias;
    unsigned span0, span1, wspan;  // bytes addressable from src0 / src1 / wgt (buffer-descriptor num_records)
    // Split-K (latency mode, round 4; implicit-GEMM kernel, ALIGNED variants): the K walk is cut into nsl_c x nsl_t slices -- slice
    // (sc, st) sums channels [sc * cb_per, (sc + 1) * cb_per) of taps [st * taps_per, (st + 1) * taps_per) in the layer's own order --
    // each computed by its own workgroups of ONE launch into an f32 slab of split_ws ([slice][pixel][ldw] raw sums; the bias enters in
    // slice 0); splitk_reduce_kernel adds the slabs in slice order and applies activation / shortcut.  nsl_c * nsl_t <= 1: off.
    int nsl_c, nsl_t, cb_per, taps_per, ldw;
    float* split_ws;
    long long slab;   // floats per slice
    int dbg;  // ablation switches for profiling (BSY_CONV_DBG; results are WRONG under them): 1 = no DMA, 4 = no epilogue
    int korder;  // the layer's K walk (conv_korder below: a function of the layer's SHAPE, never of the configuration): 0 = the packed
                 // order (taps outer, channels inner), 1 = chunk-major with 32-channel chunks (32 channels of all ntaps taps, then
                 // the next 32), 2 = chunk-major with 64-channel chunks (a BK-32 kernel visits the two halves of a chunk tap by tap).
                 // Chunk-major: a stride-2 3x3 tile re-reads its input lines within ntaps K-steps -- from the XCD's L2 -- instead
                 // of once per tap, Cin / BK K-steps apart; and it is the order the patch kernel sums in.
};

__device__ __attribute__((aligned(16))) unsigned int bsy_zero_page[16];  // zero-initialised; source of padded taps

// 16-byte LDS-DMA: the wave's 64 lanes land at lds_wave_base + 16*lane (lds_wave_base must be wave-uniform).
// Guarded so that the host pass (which only needs the launch stub) never type-checks the LDS address-space builtin.
__device__ __forceinline__ void dma16(const void* g, half_t* lds_wave_base) {
#if defined(__HIP_DEVICE_COMPILE__)
    __builtin_amdgcn_global_load_lds(g, lds_wave_base, 16, 0, 0);
#else
    (void)g;
    (void)lds_wave_base;
#endif
}

// Same through a buffer descriptor: address = base(rsrc) + voff + soff, lanes whose voff is out of range (>= num_records)
// write ZEROS to LDS -- that is how padded taps / rows past M / K padding are filled, with no zero page and no 64-bit
// per-lane pointer arithmetic (the scalar K-step displacement rides in soff).
#if defined(__HIP_DEVICE_COMPILE__)
typedef __amdgpu_buffer_rsrc_t bsy_rsrc_t;
__device__ __forceinline__ bsy_rsrc_t make_rsrc(const void* base, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), (short)0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ void dma16_buf(bsy_rsrc_t r, unsigned voff, unsigned soff, half_t* lds_wave_base) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, lds_wave_base, 16, (int)voff, (int)soff, 0, 0);
}
#else
typedef int bsy_rsrc_t;
__device__ __forceinline__ bsy_rsrc_t make_rsrc(const void*, unsigned) { return 0; }
__device__ __forceinline__ void dma16_buf(bsy_rsrc_t, unsigned, unsigned, half_t*) {}
#endif
#define BSY_OOB 0xFFFFFFF0u  // voffset that is out of range for every descriptor (num_records < 2^32 - 16)

__device__ __forceinline__ int xcd_remap(int bid, int nb) {
    // XCD-aware, bijective block remap: blocks that share an XCD (bid % 8) get consecutive logical tiles, so the
    // cout tiles of one pixel tile re-read the pixel operand from that XCD's L2.
    const int q = nb >> 3, r = nb & 7, x = bid & 7;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (bid >> 3);
}

template <int MT, int NT>
__device__ __forceinline__ void conv_epilogue(const ConvK& p, f32x16 (&acc)[NT][MT], int m0, int n0, int wm, int wn,
                                              int lrow, int lh) {
    const int ohw = p.OH * p.OW;
    const bool wide = !p.out_f32 && !(p.Cout & 15) && !(p.ldd & 7) && !((uintptr_t)p.dst & 15);
#pragma unroll
    for (int b = 0; b < MT; ++b) {
        const int m = m0 + (wm * MT + b) * 32 + lrow;
        const bool mvalid = m < p.M;
        size_t dpix = (size_t)(mvalid ? m : 0);
        if (p.dst_scale != 1) {
            const int mm = mvalid ? m : 0;
            const int n = mm / ohw;
            const int rem = mm - n * ohw;
            const int oh = rem / p.OW;
            const int ow = rem - oh * p.OW;
            dpix = ((size_t)n * (p.OH * p.dst_scale) + (oh * p.dst_scale + p.dst_dy)) * (size_t)(p.OW * p.dst_scale) +
                   (ow * p.dst_scale + p.dst_dx);
        }
#pragma unroll
        for (int a = 0; a < NT; ++a) {
            const int cbase = n0 + (wn * NT + a) * 32;
            if (cbase >= p.Cout) continue;  // wave-uniform
            if (wide) {
                // all 32 couts of this MFMA tile exist (Cout % 16 == 0 and cbase + 16 <= Cout; second half checked below)
                unsigned pk[4][2];
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int c = cbase + 8 * g + 4 * lh;
                    float v[4] = {0.f, 0.f, 0.f, 0.f};
                    if (c < p.Cout) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            float t = acc[a][b][4 * g + e];  // (the accumulators started at the bias)
                            v[e] = p.act ? silu_f(t) : t;
                        }
                        if (p.res && mvalid) {
                            const half4 rv = *reinterpret_cast<const half4*>(p.res + dpix * p.ldr + c);
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[e] += (float)rv[e];
                        }
                    }
                    half2v h0 = {(half_t)v[0], (half_t)v[1]}, h1 = {(half_t)v[2], (half_t)v[3]};
                    pk[g][0] = __builtin_bit_cast(unsigned, h0);
                    pk[g][1] = __builtin_bit_cast(unsigned, h1);
                }
#pragma unroll
                for (int g = 0; g < 4; g += 2) {
                    // lanes l / l+32 hold couts 8g+{0..3} / 8g+{4..7}: swap so each lane owns 8 consecutive couts
                    const auto r0 = __builtin_amdgcn_permlane32_swap(pk[g][0], pk[g + 1][0], false, false);
                    const auto r1 = __builtin_amdgcn_permlane32_swap(pk[g][1], pk[g + 1][1], false, false);
                    const int c = cbase + 8 * (g + lh);
                    if (mvalid && c < p.Cout) {
                        uint4 o = {r0[0], r1[0], r0[1], r1[1]};
                        *reinterpret_cast<uint4*>(reinterpret_cast<half_t*>(p.dst) + dpix * p.ldd + c) = o;
                    }
                }
                continue;
            }
            if (!mvalid) continue;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int c = cbase + 8 * g + 4 * lh;
                if (c >= p.Cout) continue;
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float t = acc[a][b][4 * g + e];
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

// Fused Detect decoder (ConvArgs::epi): the last conv of a head branch turns its accumulators straight into rows of the
// prediction tensor y (B, 4 + nc, A) -- what Detect._inference (head.py:113-148) computes from the concatenated logit
// maps -- instead of storing f32 logits for a separate decode kernel to read back (0.3 GB per batch of 64, and the
// decode launch itself).  In the accumulator layout a lane owns ONE pixel (= anchor), so for a fixed register the 32
// lanes of a half-wave write 32 consecutive anchors of one y row: coalesced without staging.
//   epi 2 (class logits): sigmoid, rows 4 + c.
//   epi 3 (box logits, 4 sides x 16 DFL bins = 64 couts in one wave: WAVES_N == 1, NT == 2): a side's 16 bins sit in
//          8 registers of this lane and 8 of lane ^ 32 -> softmax expectation with three cross-lane exchanges, then
//          dist2bbox (xywh) * stride, rows 0..3.
// The raw logits go to the level's raw map (B, 64 + nc, h, w) when the caller bound one.
template <typename T>
__device__ __forceinline__ void head_store(void* base, size_t idx, float v) { reinterpret_cast<T*>(base)[idx] = (T)v; }

// DFL + dist2bbox of one pixel per lane pair (head.py:141-146, block.py:58-77): lo / hi = the 32 x 32 accumulators of couts 0..31 /
// 32..63 of this lane's pixel; side sd's 16 bins sit in 8 registers of this lane and 8 of lane ^ 32.  Writes rows 0..3 of y (and the
// raw logits when a raw map is bound).
__device__ __forceinline__ void dfl_decode_store(const ConvK& p, const f32x16& lo, const f32x16& hi, bool mv,
                                                 size_t ybase, size_t rbase, int pix, int lh) {
    const int hw = p.OH * p.OW;
    float dist[4];
#pragma unroll
    for (int sd = 0; sd < 4; ++sd) {
        const int rb = 8 * (sd & 1);
        float v[8];
#pragma unroll
        for (int g = 0; g < 2; ++g)
#pragma unroll
            for (int e = 0; e < 4; ++e) v[4 * g + e] = ((sd >> 1) ? hi : lo)[rb + 4 * g + e];  // logits incl. bias (accumulator start)
        if (p.raw && mv) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const size_t ri = rbase + (size_t)(16 * sd + (i & 3) + 8 * (i >> 2) + 4 * lh) * hw;
                if (p.raw_f32) head_store<float>(p.raw, ri, v[i]); else head_store<half_t>(p.raw, ri, v[i]);
            }
        }
        float mx = v[0];
#pragma unroll
        for (int i = 1; i < 8; ++i) mx = fmaxf(mx, v[i]);
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        float den = 0.f, num = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float ex = __expf(v[i] - mx);
            den += ex;
            num += ex * (float)((i & 3) + 8 * (i >> 2) + 4 * lh);  // bin index of this register
        }
        den += __shfl_xor(den, 32, 64);
        num += __shfl_xor(num, 32, 64);
        dist[sd] = num / den;
    }
    if (mv && lh == 0) {
        const int ay_i = pix / p.OW, ax_i = pix - ay_i * p.OW;
        const float ax = (float)ax_i + 0.5f, ay = (float)ay_i + 0.5f;
        const float x1 = ax - dist[0], y1 = ay - dist[1], x2 = ax + dist[2], y2 = ay + dist[3];
        const float o[4] = {((x1 + x2) * 0.5f) * p.lvl_stride, ((y1 + y2) * 0.5f) * p.lvl_stride,
                            (x2 - x1) * p.lvl_stride, (y2 - y1) * p.lvl_stride};
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const size_t yi = ybase + (size_t)r * p.A;
            if (p.y_f32) head_store<float>(p.y, yi, o[r]); else head_store<half_t>(p.y, yi, o[r]);
        }
    }
}

template <int MT, int NT>
__device__ __forceinline__ void conv_epilogue_head(const ConvK& p, f32x16 (&acc)[NT][MT], int m0, int n0, int wm, int wn,
                                                   int lrow, int lh) {
    const int hw = p.OH * p.OW;
#pragma unroll
    for (int b = 0; b < MT; ++b) {
        const int m = m0 + (wm * MT + b) * 32 + lrow;
        const bool mv = m < p.M;
        const int mm = mv ? m : 0;
        const int n = mm / hw, pix = mm - n * hw;
        const size_t ybase = (size_t)n * p.nrows * p.A + p.a0 + pix;
        const size_t rbase = (size_t)n * p.rawC * hw + pix;
        if (p.epi == 2) {
#pragma unroll
            for (int a = 0; a < NT; ++a) {
                const int cl = n0 + (wn * NT + a) * 32;
                if (cl >= p.Cout) continue;  // wave-uniform
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int c0 = cl + 8 * g + 4 * lh;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int c = c0 + e;
                        if (!mv || c >= p.Cout) continue;
                        const float v = acc[a][b][4 * g + e];
                        const float sg = 1.0f / (1.0f + __expf(-v));
                        const size_t yi = ybase + (size_t)(4 + c) * p.A;
                        if (p.y_f32) head_store<float>(p.y, yi, sg); else head_store<half_t>(p.y, yi, sg);
                        if (p.raw) {
                            const size_t ri = rbase + (size_t)(64 + c) * hw;
                            if (p.raw_f32) head_store<float>(p.raw, ri, v); else head_store<half_t>(p.raw, ri, v);
                        }
                    }
                }
            }
        } else if (NT == 2) {  // epi 3; host guarantees WAVES_N == 1, n0 == 0, Cout == 64
            dfl_decode_store(p, acc[0][b], acc[NT - 1][b], mv, ybase, rbase, pix, lh);
        }
    }
}

// Class-score epilogue through LDS (f16 y, no raw map requested): the direct form above writes 64-byte pieces (32 lanes x
// one f16 anchor) -- the class conv of the 80 x 80 level ran at 2.6 TB/s.  Here sigmoid(logit) is parked transposed,
// [cout][TM + 8] f16, and leaves as 16-byte pieces of 8 consecutive anchors of one y row (falling back to single
// elements for a piece that straddles two images or starts on an odd boundary).
template <int TM, int TN, int MT, int NT, int NTHR>
__device__ __forceinline__ void conv_epilogue_cls_lds(const ConvK& p, f32x16 (&acc)[NT][MT], half_t* stile, int m0, int n0,
                                                      int wm, int wn, int lrow, int lh, int tid) {
    constexpr int LDP = TM + 8;  // padded row of the transposed tile (halves)
#pragma unroll
    for (int b = 0; b < MT; ++b) {
        const int prow = (wm * MT + b) * 32 + lrow;
#pragma unroll
        for (int a = 0; a < NT; ++a) {
            const int cl = (wn * NT + a) * 32;
            if (n0 + cl >= p.Cout) continue;  // wave-uniform; bias is padded to CoutPad only
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int c = cl + 8 * g + 4 * lh;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float v = acc[a][b][4 * g + e];
                    stile[(c + e) * LDP + prow] = (half_t)(1.0f / (1.0f + __expf(-v)));
                }
            }
        }
    }
    __syncthreads();
    const int hw = p.OH * p.OW;
    half_t* y = reinterpret_cast<half_t*>(p.y);
    constexpr int GPR = TM / 8;  // 8-anchor pieces per cout row
    for (int id = tid; id < TN * GPR; id += NTHR) {
        const int cr = id / GPR, g8 = (id - cr * GPR) * 8;
        const int c = n0 + cr, m = m0 + g8;
        if (c >= p.Cout || m >= p.M) continue;
        const int n = m / hw, pix = m - n * hw;
        const size_t yi = ((size_t)n * p.nrows + 4 + c) * p.A + p.a0 + pix;
        if (pix + 8 <= hw && m + 8 <= p.M && !(yi & 7)) {
            *reinterpret_cast<half8*>(y + yi) = *reinterpret_cast<const half8*>(stile + cr * LDP + g8);
        } else {
            for (int e = 0; e < 8 && m + e < p.M; ++e) {
                const int me = m + e, ne = me / hw, pe = me - ne * hw;
                y[((size_t)ne * p.nrows + 4 + c) * p.A + p.a0 + pe] = stile[cr * LDP + g8 + e];
            }
        }
    }
}

// Coalesced epilogue (fp16 outputs, Cout % 8 == 0): the accumulator layout (lane = pixel, 4 channels per register
// group) makes every direct store touch 32 different pixel rows; for the HBM-bound 1x1 layers that store pattern was
// the whole kernel (3x).  Instead: bias + SiLU in registers -> fp16 tile [TM][TN] in LDS (the DMA ring is free after
// the K loop) -> every thread moves 16-byte pieces with consecutive lanes on consecutive channels of one pixel row,
// adding the residual (also read coalesced) on the way.
template <int TM, int TN, int MT, int NT, int NTHR>
__device__ __forceinline__ void conv_epilogue_lds(const ConvK& p, f32x16 (&acc)[NT][MT], half_t* stile, int m0, int n0,
                                                  int wm, int wn, int lrow, int lh, int tid) {
    constexpr int LDT = TN + 8;  // padded tile row (halves)
#pragma unroll
    for (int b = 0; b < MT; ++b) {
        const int prow = (wm * MT + b) * 32 + lrow;
#pragma unroll
        for (int a = 0; a < NT; ++a) {
            const int cl = (wn * NT + a) * 32;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int c = cl + 8 * g + 4 * lh;
                half4 o;
                f32x4 t = f32x4{acc[a][b][4 * g], acc[a][b][4 * g + 1], acc[a][b][4 * g + 2], acc[a][b][4 * g + 3]};
                if (p.act) t = silu4_f(t);
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = (half_t)t[e];
                *reinterpret_cast<half4*>(stile + prow * LDT + c) = o;
            }
        }
    }
    __syncthreads();
    constexpr int CPRW = TN / 8;                 // 16-byte pieces per tile row
    constexpr int ITER = TM * CPRW / NTHR;
    half_t* dst = reinterpret_cast<half_t*>(p.dst);
#pragma unroll
    for (int i = 0; i < ITER; ++i) {
        const int id = tid + NTHR * i;
        const int row = id / CPRW, cc = (id % CPRW) * 8;
        const int m = m0 + row, c = n0 + cc;
        if (m >= p.M || c >= p.Cout) continue;
        half8 v = *reinterpret_cast<const half8*>(stile + row * LDT + cc);
        if (p.res) {
            const half8 r = *reinterpret_cast<const half8*>(p.res + (size_t)m * p.ldr + c);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = (half_t)((float)v[e] + (float)r[e]);
        }
        *reinterpret_cast<half8*>(dst + (size_t)m * p.ldd + c) = v;
    }
}

// BK = channels per K-step (32 or 64).  LDS rows hold BK halves (64 / 128 B); a 1-KiB DMA wave-instruction fills
// RPI = 512/BK rows.  BK = 64 needs Cin % 64 == 0: every staged row is then a full 128-B line of its source, which halves
// the L2 request count per byte (the LDS-DMA gather rate from L2 is request-bound: ~35 GB/s/CU with 64-B pieces).
// ALIGNED: Cin % BK == 0 and C0 % BK == 0, so every K-step lies inside ONE tap of ONE source: all of the K bookkeeping
// (tap, source, channel base, tap displacement) is wave-uniform scalar work and a DMA address is row_offset + scalar,
// issued through a buffer descriptor (out-of-range lanes -> zeros).  The generic variant (thin layers: Cin = 8 / 16 /
// 48 ..., BK = 32 only) keeps it per lane and uses flat LDS-DMA + a zero page.
template <int KS, int WAVES_M, int WAVES_N, int MT, int NT, int STAGES, bool ALIGNED, int BK>
__global__ __launch_bounds__(64 * WAVES_M * WAVES_N) void conv_mfma_kernel(const ConvK p) {
    constexpr int NW = WAVES_M * WAVES_N;   // waves per workgroup (4 or 8)
    constexpr int NTHR = 64 * NW;
    static_assert(BK == 32 || (BK == 64 && ALIGNED), "BK = 64 only for aligned layers");
    constexpr int TM = WAVES_M * MT * 32;
    constexpr int TN = WAVES_N * NT * 32;
    constexpr int TNS = TN < 64 ? 64 : TN;  // weight rows staged (>= 64 so that every wave issues the same DMA count;
                                            // the packed weights have >= 128 rows, the extra rows are simply unused)
    constexpr int CPR = BK / 8;             // 16-byte chunks per LDS row
    constexpr int RPI = 64 / CPR;           // rows filled by one DMA wave-instruction
    constexpr int PIW = TM / (RPI * NW);    // pixel-tile DMA instructions per wave
    constexpr int WIW = TNS / (RPI * NW);   // weight-tile DMA instructions per wave
    static_assert(PIW >= 1 && WIW >= 1 && (BK == 32 || (!(PIW & 1) && !(WIW & 1))), "tile too small for this wave count");
    constexpr int STAGE = (TM + TNS) * BK;  // halves per LDS stage
    constexpr int NDMA = PIW + WIW;         // DMA instructions per thread per K-step (identical for all waves)
    constexpr int SWS = BK == 32 ? 2 : 1;   // read-side swizzle: chunk ^ ((row >> SWS) & (CPR - 1))
    constexpr int OTILE = (TM > TN ? TM : TN) * ((TM > TN ? TN : TM) + 8);  // fp16 output tile staged by the coalesced epilogues:
                                                                             // [TM][TN + 8], or transposed [TN][TM + 8] (class scores)
    constexpr int SMEM = STAGES * STAGE > OTILE ? STAGES * STAGE : OTILE;
    __shared__ __attribute__((aligned(16))) half_t smem[SMEM];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    int wg = xcd_remap(blockIdx.x, gridDim.x);
    // split-K: workgroups [slice * tiles, (slice + 1) * tiles) compute slice `slice` of every tile
    const bool split = ALIGNED && p.nsl_c * p.nsl_t > 1;
    int slice = 0;
    if (split) {
        const int per = p.ntn * ((p.M + TM - 1) / TM);
        slice = wg / per;
        wg -= slice * per;
    }
    const int tap0 = split ? (slice % p.nsl_t) * p.taps_per : 0, tap1 = split ? tap0 + p.taps_per : p.ntaps;
    const int cb0 = split ? (slice / p.nsl_t) * p.cb_per : 0, cb1 = split ? cb0 + p.cb_per : p.Cin8 * 8;
    const int tn_idx = wg % p.ntn;
    const int tm_idx = wg / p.ntn;
    const int m0 = tm_idx * TM;
    const int n0 = tn_idx * TN;

    // ---- per-thread DMA coordinates: lane -> (row = lane / CPR, slot = lane % CPR) of the RPI x CPR block one
    //      wave-instruction fills; the slot holds source chunk slot ^ swizzle(row)  (== the read-side swizzle).
    //      BK = 32: swizzle(row) = (row>>2)&3 = (rsub>>2)&3.  BK = 64: (row>>1)&7 = ((q&1)<<2 | rsub>>1), q = instruction
    //      index inside the tile; PIW / WIW are even there, so the parity is that of the unrolled index.
    const int rsub = lane / CPR;
    const int slot = lane % CPR;
    const int kc0 = BK == 32 ? (slot ^ ((rsub >> 2) & 3)) : (slot ^ (rsub >> 1));  // even instructions
    const int kc1 = BK == 32 ? kc0 : (slot ^ (4 | (rsub >> 1)));                     // odd instructions (BK = 64)
    // per staged pixel row: element offset of the window origin pixel (tap 0,0 -> may be "negative" = wraps, only used
    // when the tap is valid) in each source, and a bitmask of the taps that fall inside the image
    unsigned off0[PIW], off1[PIW], vmask[PIW];
    const int ohw = p.OH * p.OW;
#pragma unroll
    for (int i = 0; i < PIW; ++i) {
        const int m = m0 + (wave * PIW + i) * RPI + rsub;
        const bool rv = m < p.M;
        const int mm = rv ? m : 0;
        int n, oh, ow;
        if (KS == 1 && p.stride == 1) {  // pixel index == m unless a source is upsampled
            n = 0; oh = 0; ow = 0;
            if (p.up0 | p.up1) { n = mm / ohw; const int rem = mm - n * ohw; oh = rem / p.OW; ow = rem - oh * p.OW; }
        } else {
            n = mm / ohw; const int rem = mm - n * ohw; oh = rem / p.OW; ow = rem - oh * p.OW;
        }
        const int iy0 = oh * p.stride - p.pad, ix0 = ow * p.stride - p.pad;
        if (KS == 1) {
            vmask[i] = rv ? 1u : 0u;
            if (p.stride == 1 && !(p.up0 | p.up1)) {
                off0[i] = (unsigned)mm * (unsigned)p.ld0;
                off1[i] = (unsigned)mm * (unsigned)p.ld1;
            } else {
                const int H0 = p.H >> p.up0, W0 = p.W >> p.up0, H1 = p.H >> p.up1, W1 = p.W >> p.up1;
                off0[i] = (unsigned)((n * H0 + (iy0 >> p.up0)) * W0 + (ix0 >> p.up0)) * (unsigned)p.ld0;
                off1[i] = (unsigned)((n * H1 + (iy0 >> p.up1)) * W1 + (ix0 >> p.up1)) * (unsigned)p.ld1;
            }
        } else {
            unsigned rb = 0, cb = 0;
#pragma unroll
            for (int t = 0; t < KS; ++t) {
                rb |= ((unsigned)(iy0 + t) < (unsigned)p.H ? 1u : 0u) << t;
                cb |= ((unsigned)(ix0 + t) < (unsigned)p.W ? 1u : 0u) << t;
            }
            unsigned vm = 0;
#pragma unroll
            for (int t = 0; t < KS; ++t) vm |= ((rb >> t) & 1u) ? (cb << (t * KS)) : 0u;
            vmask[i] = rv ? vm : 0u;
            const unsigned pix = (unsigned)((n * p.H + iy0) * p.W + ix0);  // wraps for border rows; masked there
            off0[i] = pix * (unsigned)p.ld0;
            off1[i] = pix * (unsigned)p.ld1;
        }
    }
    unsigned woff[WIW];
#pragma unroll
    for (int j = 0; j < WIW; ++j)
    {
        // rows past the packed matrix (a 32-wide cout tile stages 64 rows; the last cout tile of a 128-row matrix would read
        // rows 128..159): the descriptor path returns zeros for them, the flat-DMA (generic) path must not touch them --
        // clamp to the last packed row, the values are never used
        const int wrow = min(n0 + (wave * WIW + j) * RPI + rsub, ((p.Cout + 127) & ~127) - 1);
        woff[j] = (unsigned)wrow * (unsigned)p.Kpad + ((j & 1) ? kc1 : kc0) * 8;
    }
    const half_t* zero = reinterpret_cast<const half_t*>(bsy_zero_page);
    const bsy_rsrc_t rs0 = make_rsrc(p.src0, p.span0), rs1 = make_rsrc(p.src1 ? p.src1 : p.src0, p.src1 ? p.span1 : 0u),
                     rsw = make_rsrc(p.wgt, p.wspan);

    // K-step state.  ALIGNED: uniform (scalar) tap / channel base.  Generic: per-lane tap / 8-channel chunk index.
    int s_tap = tap0, s_cb = cb0;
    int tap = ALIGNED ? 0 : kc0 / p.Cin8;
    int c8 = ALIGNED ? 0 : kc0 - tap * p.Cin8;
    const int Cin = p.Cin8 * 8;

    // K-step state travels BY VALUE (cur_*) and is advanced by advance() below on plain locals: captured by reference and
    // mutated inside this lambda the four ints ended up in a 12-byte private (scratch) frame -- reloaded per lane and then
    // fed to the DMA's scalar offset operands through waterfall loops, in every ALIGNED instantiation (round-1 VERDICT)
    auto issue = [&](int kt, int stage, const int s_tap, const int s_cb, const int tap, const int c8) {
        half_t* sP = smem + stage * STAGE;
        half_t* sW = sP + TM * BK;
        if (p.dbg & 1) return;
        if (ALIGNED) {
            const bool s1 = s_cb >= p.C0;
            const int ld = s1 ? p.ld1 : p.ld0;
            const int kh = (KS == 1) ? 0 : s_tap / KS;
            const int kw = (KS == 1) ? 0 : s_tap - kh * KS;
            // scalar part of the byte offset: tap displacement + channel base inside the source
            const unsigned sc = 2u * (unsigned)((kh * p.W + kw) * ld + (s1 ? s_cb - p.C0 : s_cb));
            const unsigned tbit = 1u << s_tap;
#pragma unroll
            for (int i = 0; i < PIW; ++i) {
                const unsigned kcb = 16u * (unsigned)((i & 1) ? kc1 : kc0);
                const unsigned o = (vmask[i] & tbit) ? 2u * (s1 ? off1[i] : off0[i]) + sc + kcb : BSY_OOB;
                if (s1) dma16_buf(rs1, o, 0u, sP + (wave * PIW + i) * RPI * BK);
                else dma16_buf(rs0, o, 0u, sP + (wave * PIW + i) * RPI * BK);
            }
#pragma unroll
            for (int j = 0; j < WIW; ++j)
                dma16_buf(rsw, 2u * woff[j], 2u * (unsigned)(s_tap * Cin + s_cb), sW + (wave * WIW + j) * RPI * BK);
        } else {
            const bool kvalid = tap < p.ntaps;
            const int kh = (KS == 1) ? 0 : tap / KS;
            const int kw = (KS == 1) ? 0 : tap - kh * KS;
            const int cc = c8 * 8;
            const bool s1 = cc >= p.C0;
            const half_t* base = s1 ? p.src1 : p.src0;
            const int ld = s1 ? p.ld1 : p.ld0;
            const unsigned sc = (unsigned)((kh * p.W + kw) * ld + (s1 ? cc - p.C0 : cc));
            const unsigned tbit = kvalid ? (1u << tap) : 0u;
#pragma unroll
            for (int i = 0; i < PIW; ++i) {
                const unsigned o = (s1 ? off1[i] : off0[i]) + sc;
                const half_t* g = (vmask[i] & tbit) ? base + (size_t)o : zero;
                dma16(g, sP + (wave * PIW + i) * RPI * BK);
            }
#pragma unroll
            for (int j = 0; j < WIW; ++j)
                dma16(p.wgt + (size_t)(woff[j] + (unsigned)kt * 32u), sW + (wave * WIW + j) * RPI * BK);
        }
    };
#define BSY_ADVANCE_K()                                                                                  \
    do {                                                                                                 \
        if (ALIGNED) {                                                                                   \
            if (p.korder == 1 || (p.korder == 2 && BK == 64)) { /* chunk-major, chunk = this kernel's BK */ \
                if (++s_tap >= tap1) { s_tap = tap0; s_cb += BK; }                                       \
            } else if (p.korder == 2) { /* 64-channel chunks walked by a BK-32 kernel: (tap, half) */       \
                if (s_cb & 32) { s_cb -= 32; if (++s_tap >= tap1) { s_tap = tap0; s_cb += 64; } }        \
                else s_cb += 32;                                                                         \
            } else {                                                                                     \
                s_cb += BK;                                                                              \
                if (s_cb >= cb1) { s_cb = cb0; ++s_tap; }                                                \
            }                                                                                            \
        } else {                                                                                         \
            c8 += 4;                                                                                     \
            while (c8 >= p.Cin8) { c8 -= p.Cin8; ++tap; }                                                \
        }                                                                                                \
    } while (0)

    const int lrow = lane & 31;
    const int lh = lane >> 5;
    const int nk = split ? (tap1 - tap0) * (cb1 - cb0) / BK : p.Kpad / BK;
    f32x16 acc[NT][MT];  // start at the bias of their couts (common.h acc_bias); split-K: in slice 0 only
#pragma unroll
    for (int a = 0; a < NT; ++a)
#pragma unroll
        for (int b = 0; b < MT; ++b) {
            acc_bias(acc[a][b], p.bias + n0 + (wn * NT + a) * 32, lh);
            if (slice) {
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
            }
        }

    // prologue: STAGES-1 K-steps in flight
#pragma unroll
    for (int s = 0; s < STAGES - 1; ++s)
        if (s < nk) { issue(s, s, s_tap, s_cb, tap, c8); BSY_ADVANCE_K(); }

    for (int kt = 0; kt < nk; ++kt) {
        // wait for K-step kt: everything issued after it may stay in flight
        const int later = min(STAGES - 2, nk - 1 - kt);  // K-steps issued after kt that are still outstanding
        if (later >= STAGES - 2 && STAGES > 2) {
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"((STAGES - 2) * NDMA) : "memory");
        } else if (STAGES > 3 && later == STAGES - 3) {
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"((STAGES > 3 ? STAGES - 3 : 0) * NDMA) : "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();  // K-step kt visible to every wave; every wave is done reading stage (kt-1)%STAGES
        if (kt + STAGES - 1 < nk) { issue(kt + STAGES - 1, (kt + STAGES - 1) % STAGES, s_tap, s_cb, tap, c8); BSY_ADVANCE_K(); }
        const half_t* sP = smem + (kt % STAGES) * STAGE;
        const half_t* sW = sP + TM * BK;
        // Fragment reads run ONE 16-wide sub-step ahead of the MFMAs that use them (two register sets; the sched_barriers pin the
        // order -- left alone the scheduler reuses one set and serialises read -> wait -> MFMAs).  The workgroup's waves leave the
        // barrier together, so with all reads of the K-step in front of its MFMA burst (the round-1 form) every wave read while the
        // matrix pipes idled and then every wave multiplied while the LDS idled: 20 us of a 10-us MFMA floor on model.8.cv2 with
        // DMA and epilogue switched off.  Same MFMA order (ks, cout tile, pixel tile): same bits.
        constexpr int KSUB = BK / 16;
        half8 bfr[2][MT], afr[2][NT];
        auto rd = [&](const int ks, const int buf) {
            const int chunk = 2 * ks + lh;
#pragma unroll
            for (int b = 0; b < MT; ++b) {
                const int row = (wm * MT + b) * 32 + lrow;
                bfr[buf][b] = *reinterpret_cast<const half8*>(sP + row * BK + ((chunk ^ ((row >> SWS) & (CPR - 1))) << 3));
            }
#pragma unroll
            for (int a = 0; a < NT; ++a) {
                const int row = (wn * NT + a) * 32 + lrow;
                afr[buf][a] = *reinterpret_cast<const half8*>(sW + row * BK + ((chunk ^ ((row >> SWS) & (CPR - 1))) << 3));
            }
        };
        rd(0, 0);
#pragma unroll
        for (int ks = 0; ks < KSUB; ++ks) {
            if (ks + 1 < KSUB) rd(ks + 1, (ks + 1) & 1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int a = 0; a < NT; ++a)
#pragma unroll
                for (int b = 0; b < MT; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(afr[ks & 1][a], bfr[ks & 1][b], acc[a][b], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    if (p.dbg & 4) {
        if (acc[0][0][0] == 12345.678f) reinterpret_cast<float*>(p.dst)[0] = 1.f;  // keep the accumulators live
        return;
    }
    if (split) {  // raw f32 sums of this slice -> its slab [pixel][ldw]; lane = pixel, registers 4 g .. 4 g + 3 = couts 8 g + 4 lh + {0..3}
        float* ws = p.split_ws + (size_t)slice * (size_t)p.slab;
#pragma unroll
        for (int b = 0; b < MT; ++b) {
            const int m = m0 + (wm * MT + b) * 32 + lrow;
            if (m >= p.M) continue;
#pragma unroll
            for (int a = 0; a < NT; ++a) {
                const int cbase = n0 + (wn * NT + a) * 32;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int c = cbase + 8 * g + 4 * lh;
                    if (c < p.ldw) *reinterpret_cast<f32x4*>(ws + (size_t)m * p.ldw + c) = f32x4{acc[a][b][4 * g], acc[a][b][4 * g + 1], acc[a][b][4 * g + 2], acc[a][b][4 * g + 3]};
                }
            }
        }
        return;
    }
    if (p.epi) {  // fused Detect decoder
        if (p.epi == 2 && !p.y_f32 && !p.raw) {
            __syncthreads();  // every wave has finished reading the last K-step: LDS becomes the transposed score tile
            conv_epilogue_cls_lds<TM, TN, MT, NT, NTHR>(p, acc, smem, m0, n0, wm, wn, lrow, lh, tid);
        } else {
            conv_epilogue_head<MT, NT>(p, acc, m0, n0, wm, wn, lrow, lh);
        }
        return;
    }
    const bool lds_epi = !p.out_f32 && !(p.Cout & 7) && !(p.ldd & 7) && !((uintptr_t)p.dst & 15) && p.dst_scale == 1 &&
                         (!p.res || (!(p.ldr & 7) && !((uintptr_t)p.res & 15)));
    if (lds_epi) {
        __syncthreads();  // every wave has finished reading the last K-step (all DMA already drained by vmcnt(0))
        conv_epilogue_lds<TM, TN, MT, NT, NTHR>(p, acc, smem, m0, n0, wm, wn, lrow, lh, tid);
    } else {
        conv_epilogue<MT, NT>(p, acc, m0, n0, wm, wn, lrow, lh);
    }
}

#undef BSY_ADVANCE_K

// ---------------------------------------------------------------------------------------------------------------------
// Persistent 1x1 kernel with dedicated store waves ("thin-K" layers: 2-24 K-steps per tile).
//
// With one tile per workgroup the load -> MFMA -> store phases of those layers only overlap across the 2-5 resident
// workgroups: removing the epilogue alone made them 2.4x faster, removing the DMA alone 2x.  Here a workgroup of 8 waves
// walks tiles (XCD-aware order, common.h xcd_tile_walk) with ONE continuous DMA ring:
//   waves 0-3 (compute): issue the LDS-DMA (running STAGES-1 K-steps ahead, ACROSS tile boundaries), MFMA, then park
//                        bias + SiLU results as an fp16 tile in LDS and carry straight on with the next tile;
//   waves 4-7 (store)  : walk the same barrier sequence, and after each tile's hand-over barrier move the parked tile to
//                        HBM with coalesced 16-byte stores.  Their stores sit in THEIR OWN vmcnt queues, so the compute
//                        waves' counted `s_waitcnt vmcnt(N)` keeps seeing nothing but DMA.
// Barrier protocol (every wave executes every s_barrier): per tile nk "K-step" barriers B(t,k) + one hand-over barrier
// E(t) after the compute waves' tile writes.  Store waves read tile t between E(t) and B(t+1,0); compute waves overwrite
// the parked tile only after B(t+1,nk-1) >= B(t+1,0): no second tile buffer needed.
// Restrictions (checked by conv_cfg_valid): ksize 1, ALIGNED channels, fp16 output, no residual, Cout % 8 == 0.
// ---------------------------------------------------------------------------------------------------------------------
#define PERSIST_MAX_COUT 1024
template <int WAVES_M, int WAVES_N, int MT, int NT, int STAGES, int BK>
__global__ __launch_bounds__(512) void conv1x1_persist_kernel(const ConvK p) {
    constexpr int TM = WAVES_M * MT * 32, TN = WAVES_N * NT * 32;
    constexpr int TNS = TN < 64 ? 64 : TN;
    constexpr int CPR = BK / 8, RPI = 64 / CPR;
    constexpr int PIW = TM / (RPI * 4), WIW = TNS / (RPI * 4);
    constexpr int STAGE = (TM + TNS) * BK, NDMA = PIW + WIW;
    constexpr int SWS = BK == 32 ? 2 : 1;
    constexpr int LDT = TN + 8, OT = TM * LDT;
    static_assert(WAVES_M * WAVES_N == 4 && PIW >= 1 && WIW >= 1 && (BK == 32 || (!(PIW & 1) && !(WIW & 1))), "bad tile");
    __shared__ __attribute__((aligned(16))) half_t smem[STAGES * STAGE + OT + 2 * PERSIST_MAX_COUT];
    half_t* otile = smem + STAGES * STAGE;
    float* sbias = reinterpret_cast<float*>(otile + OT);  // the whole (padded) bias vector, loaded once: keeps ordinary VMEM
                                                          // loads -- and the vmcnt(0) they imply -- out of the tile loop

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nk = p.Kpad / BK;
    const int ntiles = p.ntm * p.ntn;
    // XCD-aware tile walk (common.h xcd_tile_walk): the cout tiles of one pixel tile are neighbours in the tile order, so they run on
    // ONE XCD at the same time and the second reads the pixels from its L2 (model.4.cv2: 540 MB of traffic for 367 MB otherwise)
    const TileWalk tw = xcd_tile_walk(blockIdx.x, gridDim.x, ntiles);
    const int T0 = tw.tile, G = tw.step;
    const int my_tiles = T0 < tw.end ? (tw.end - T0 + G - 1) / G : 0;
    for (int i = tid; i < p.ntn * TN; i += 512) sbias[i] = p.bias[i];  // bias is padded to CoutPad (multiple of 128)
    __syncthreads();

    if (wave >= 4) {
        // ------------------------------------------------ store waves ------------------------------------------------
        const int st = tid - 256;
        constexpr int CPRW = TN / 8, ITER = TM * CPRW / 256;
        half_t* dst = reinterpret_cast<half_t*>(p.dst);
        for (int t = 0; t < my_tiles; ++t) {
            const int tile = T0 + t * G;
            const int m0 = (tile / p.ntn) * TM, n0 = (tile % p.ntn) * TN;
            for (int k = 0; k < nk; ++k) __builtin_amdgcn_s_barrier();  // B(t, k)
            __builtin_amdgcn_s_barrier();                              // E(t): tile t is parked
            half8 v[ITER];
#pragma unroll
            for (int i = 0; i < ITER; ++i) {
                const int id = st + 256 * i;
                v[i] = *reinterpret_cast<const half8*>(otile + (id / CPRW) * LDT + (id % CPRW) * 8);
            }
#pragma unroll
            for (int i = 0; i < ITER; ++i) {
                const int id = st + 256 * i;
                const int m = m0 + id / CPRW, c = n0 + (id % CPRW) * 8;
                if (m < p.M && c < p.Cout) *reinterpret_cast<half8*>(dst + (size_t)m * p.ldd + c) = v[i];
            }
            // all LDS reads have returned (their values were consumed by the stores' operands): the next barrier this
            // wave joins is B(t+1, 0), before which the compute waves cannot touch the parked tile again
        }
        return;
    }

    // ------------------------------------------------ compute + DMA waves ------------------------------------------------
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int rsub = lane / CPR, slot = lane % CPR;
    const int kc0 = BK == 32 ? (slot ^ ((rsub >> 2) & 3)) : (slot ^ (rsub >> 1));
    const int kc1 = BK == 32 ? kc0 : (slot ^ (4 | (rsub >> 1)));
    const bsy_rsrc_t rs0 = make_rsrc(p.src0, p.span0), rs1 = make_rsrc(p.src1 ? p.src1 : p.src0, p.src1 ? p.span1 : 0u),
                     rsw = make_rsrc(p.wgt, p.wspan);
    const int ohw = p.OH * p.OW;
    const int Cin = p.Cin8 * 8;

    // issue-side state (runs ahead of the compute side, across tile boundaries)
    unsigned off0[PIW], off1[PIW], woff[WIW];
    bool rv[PIW];
    int it = 0, ikt = 0, s_cb = 0;
    auto setup_tile = [&](int tcount) {
        const int tile = T0 + tcount * G;
        const int m0 = (tile / p.ntn) * TM, n0 = (tile % p.ntn) * TN;
#pragma unroll
        for (int i = 0; i < PIW; ++i) {
            const int m = m0 + (wave * PIW + i) * RPI + rsub;
            rv[i] = m < p.M;
            const int mm = rv[i] ? m : 0;
            if (!(p.up0 | p.up1) && p.stride == 1) {
                off0[i] = (unsigned)mm * (unsigned)p.ld0;
                off1[i] = (unsigned)mm * (unsigned)p.ld1;
            } else {
                const int n = mm / ohw, rem = mm - n * ohw, oh = rem / p.OW, ow = rem - oh * p.OW;
                const int iy = oh * p.stride, ix = ow * p.stride;
                const int H0 = p.H >> p.up0, W0 = p.W >> p.up0, H1 = p.H >> p.up1, W1 = p.W >> p.up1;
                off0[i] = (unsigned)((n * H0 + (iy >> p.up0)) * W0 + (ix >> p.up0)) * (unsigned)p.ld0;
                off1[i] = (unsigned)((n * H1 + (iy >> p.up1)) * W1 + (ix >> p.up1)) * (unsigned)p.ld1;
            }
        }
#pragma unroll
        for (int j = 0; j < WIW; ++j)
            woff[j] = (unsigned)(n0 + (wave * WIW + j) * RPI + rsub) * (unsigned)p.Kpad + ((j & 1) ? kc1 : kc0) * 8;
        s_cb = 0;
    };
    auto issue_step = [&](int stage) {
        if (ikt == 0) setup_tile(it);
        half_t* sP = smem + stage * STAGE;
        half_t* sW = sP + TM * BK;
        const bool s1 = s_cb >= p.C0;
        const unsigned sc = 2u * (unsigned)(s1 ? s_cb - p.C0 : s_cb);
#pragma unroll
        for (int i = 0; i < PIW; ++i) {
            const unsigned kcb = 16u * (unsigned)((i & 1) ? kc1 : kc0);
            const unsigned o = rv[i] ? 2u * (s1 ? off1[i] : off0[i]) + sc + kcb : BSY_OOB;
            if (s1) dma16_buf(rs1, o, 0u, sP + (wave * PIW + i) * RPI * BK);
            else dma16_buf(rs0, o, 0u, sP + (wave * PIW + i) * RPI * BK);
        }
#pragma unroll
        for (int j = 0; j < WIW; ++j)
            dma16_buf(rsw, 2u * woff[j], (unsigned)ikt * (2u * BK), sW + (wave * WIW + j) * RPI * BK);
        s_cb += BK;
        if (++ikt == nk) { ikt = 0; ++it; }
    };

    const int lrow = lane & 31, lh = lane >> 5;
    f32x16 acc[NT][MT];  // start at the bias of the tile's couts (common.h acc_bias), from the LDS copy of the bias vector
#pragma unroll
    for (int a = 0; a < NT; ++a)
#pragma unroll
        for (int b = 0; b < MT; ++b) acc_bias(acc[a][b], sbias + (T0 % p.ntn) * TN + (wn * NT + a) * 32, lh);
    const int total = my_tiles * nk;
    int ip = 0;
    for (; ip < STAGES - 1 && ip < total; ++ip) issue_step(ip % STAGES);
    int ckt = 0, ct = 0;
    for (int cp = 0; cp < total; ++cp) {
        const int later = min(STAGES - 2, total - 1 - cp);
        if (later >= STAGES - 2 && STAGES > 2) {
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"((STAGES - 2) * NDMA) : "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();  // B(ct, ckt)
        if (ip < total) { issue_step(ip % STAGES); ++ip; }
        const half_t* sP = smem + (cp % STAGES) * STAGE;
        const half_t* sW = sP + TM * BK;
        constexpr int KSUB = BK / 16;
        half8 bfr[KSUB][MT], afr[KSUB][NT];
#pragma unroll
        for (int ks = 0; ks < KSUB; ++ks) {
            const int chunk = 2 * ks + lh;
#pragma unroll
            for (int b = 0; b < MT; ++b) {
                const int row = (wm * MT + b) * 32 + lrow;
                bfr[ks][b] = *reinterpret_cast<const half8*>(sP + row * BK + ((chunk ^ ((row >> SWS) & (CPR - 1))) << 3));
            }
#pragma unroll
            for (int a = 0; a < NT; ++a) {
                const int row = (wn * NT + a) * 32 + lrow;
                afr[ks][a] = *reinterpret_cast<const half8*>(sW + row * BK + ((chunk ^ ((row >> SWS) & (CPR - 1))) << 3));
            }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ks = 0; ks < KSUB; ++ks)
#pragma unroll
            for (int a = 0; a < NT; ++a)
#pragma unroll
                for (int b = 0; b < MT; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(afr[ks][a], bfr[ks][b], acc[a][b], 0, 0, 0);
        if (++ckt == nk) {
            ckt = 0;
            const int n1 = ((T0 + (ct + 1) * G) % p.ntn) * TN;  // cout tile of the NEXT tile: its biases restart the accumulators
            // park SiLU(acc) as fp16 (the store waves finished reading the previous tile before B(ct, 0))
#pragma unroll
            for (int b = 0; b < MT; ++b) {
                const int prow = (wm * MT + b) * 32 + lrow;
#pragma unroll
                for (int a = 0; a < NT; ++a) {
                    const int cl = (wn * NT + a) * 32;
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int c = cl + 8 * g + 4 * lh;
                        const f32x4 bn = *reinterpret_cast<const f32x4*>(sbias + n1 + c);
                        half4 o;
                        f32x4 tv = f32x4{acc[a][b][4 * g], acc[a][b][4 * g + 1], acc[a][b][4 * g + 2], acc[a][b][4 * g + 3]};
                        if (p.act) tv = silu4_f(tv);
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            o[e] = (half_t)tv[e];
                            acc[a][b][4 * g + e] = bn[e];
                        }
                        *reinterpret_cast<half4*>(otile + prow * LDT + c) = o;
                    }
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();  // E(ct)
            ++ct;
        }
    }
}

template <int WM, int WN, int MT, int NT, int STAGES, int BK>
static int launch_persist(const ConvK& k, hipStream_t s) {
    constexpr int TM = WM * MT * 32, TN = WN * NT * 32;
    ConvK p = k;
    p.ntn = ceil_div(k.Cout, TN);
    p.ntm = ceil_div(k.M, TM);
    const long long ntiles = (long long)p.ntm * p.ntn;
    if (ntiles <= 0 || ntiles > 0x7fffffffLL) BSY_FAIL(BSY_ERR_ARG, "conv: tile count %lld out of range", ntiles);
    constexpr int lds_bytes = ((TM + (TN < 64 ? 64 : TN)) * BK * STAGES + TM * (TN + 8) + 2 * PERSIST_MAX_COUT) * 2;
    const int per_cu = 160 * 1024 / lds_bytes > 2 ? 2 : (160 * 1024 / lds_bytes);  // 512-thread workgroups: <= 2 per CU
    const long long grid = ntiles < 256LL * per_cu ? ntiles : 256LL * per_cu;
    hipLaunchKernelGGL((conv1x1_persist_kernel<WM, WN, MT, NT, STAGES, BK>), dim3((unsigned)grid), dim3(512), 0, s, p);
    HIP_TRY(hipGetLastError());
    return BSY_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// Weights-resident streaming 1x1 kernel (configuration tiles 14 / 15; round 3) for the HBM-bound 1x1 layers at 160^2 / 80^2.
//
// What held those layers at 3.3-4.6 TB/s (a device copy runs at 5.4): every 128-pixel tile re-staged the layer's whole weight
// matrix from L2 into LDS (model.4.cv2: 49 KB of pixels + 98 KB of weights per tile pair = 14 B/clk/CU, the LDS-DMA rate of a CU),
// and load / MFMA / store phases of a tile only overlap across workgroups.  Here a persistent workgroup (4 waves, two per CU)
// works on ONE cout tile for the whole launch and keeps its weights -- every wave the A fragments of its 32 NT couts over all of K
// -- in REGISTERS; only pixels stream through LDS (one continuous DMA ring across tile boundaries, STAGES K-steps deep), so the
// L2 -> LDS traffic is the input itself.  Workgroups b and b + 8 (one XCD under round-robin dispatch) take the same pixel tiles and
// different cout tiles: the second reads the pixels from that XCD's L2.
// A tile's results leave through an LDS tile as 16-byte stores issued by the same waves; they sit in the vmcnt queue BEHIND the DMA
// of the next steps and are accounted for in the counted waits (a wait for K-step g sees the stores of the previous tile as younger
// operations during the first STAGES - 1 steps of a tile).
// Same K order and epilogue arithmetic as the other 1x1 kernels: the same bits.
// Restrictions (conv_cfg_valid): as the persistent kernel, and K / 16 x NT <= 32 fragments at NT = 1, 24 at NT = 2 (more spills).
// ---------------------------------------------------------------------------------------------------------------------
// NFR = A fragments held per wave (K / 16 x NT <= NFR): 32 = 128 VGPRs
template <int NT, int STAGES, int NFR>
__global__ __launch_bounds__(256, 2) void conv1x1_wres_kernel(const ConvK p) {
    constexpr int BK = 32, MT = 2, TM = 128, TN = 64 * NT;
    constexpr int CPR = 4, RPI = 16;
    constexpr int PIW = TM / (RPI * 4);  // 2 pixel DMA wave-instructions per wave per K-step
    constexpr int STAGE = TM * BK;
    constexpr int LDT = TN + 8, OT = TM * LDT;
    constexpr int CPRW = TN / 8, NST = TM * CPRW / 256;  // 16-byte output pieces per thread per tile
    __shared__ __attribute__((aligned(16))) half_t smem[STAGES * STAGE + OT + 2 * TN];
    half_t* otile = smem + STAGES * STAGE;
    float* sbias = reinterpret_cast<float*>(otile + OT);  // this cout tile's biases: no ordinary VMEM load inside the tile loop

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int lrow = lane & 31, lh = lane >> 5;
    const int nk = p.Kpad / BK;
    // workgroup -> (cout tile, walk over the pixel tiles): blocks b, b + 8, .. b + 8 (ntn - 1) share their pixel tiles
    // (host: gridDim.x is a multiple of 8 ntn)
    constexpr int X = 8;
    const int x = blockIdx.x % X, q = blockIdx.x / X;
    const int ct = q % p.ntn, slot = q / p.ntn;                   // cout tile, walker index on this XCD
    const int walkers = (gridDim.x / X) / p.ntn;
    const int per = p.ntm / X, rem = p.ntm % X;
    const int pt0 = x * per + (x < rem ? x : rem), ptn = per + (x < rem ? 1 : 0);  // this XCD's range of pixel tiles
    const int my_tiles = slot < ptn ? (ptn - slot + walkers - 1) / walkers : 0;
    const int n0 = ct * TN;

    // A fragments of this wave's couts for the whole of K (rows past Cout: the packed matrix is padded to 128 rows with zeros)
    half8 afr[NFR];
    const int nfr = nk * 2 * NT;  // (k sub-step of 16, cout tile a)
#pragma unroll
    for (int i = 0; i < NFR; ++i) {
        afr[i] = half8{0, 0, 0, 0, 0, 0, 0, 0};
        if (i < nfr) {
            const int ks = i / NT, a = i - ks * NT;
            afr[i] = *reinterpret_cast<const half8*>(p.wgt + (size_t)(n0 + (wn * NT + a) * 32 + lrow) * p.Kpad + 16 * ks + 8 * lh);
        }
    }
    if (tid < TN) sbias[tid] = p.bias[n0 + tid];  // bias is padded to CoutPad (a multiple of 128)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // weights and bias are in: from here on the vector-memory queue holds DMA and stores only
    __syncthreads();

    const int rsub = lane / CPR, slot4 = lane % CPR;
    const int kc = slot4 ^ ((rsub >> 2) & 3);
    const bsy_rsrc_t rs0 = make_rsrc(p.src0, p.span0), rs1 = make_rsrc(p.src1 ? p.src1 : p.src0, p.src1 ? p.span1 : 0u);
    const int ohw = p.OH * p.OW;
    unsigned off0[PIW], off1[PIW];
    bool rv[PIW];
    int it = 0, ikt = 0, s_cb = 0;
    auto setup_tile = [&](int tcount) {
        const int m0 = (pt0 + slot + tcount * walkers) * TM;
#pragma unroll
        for (int i = 0; i < PIW; ++i) {
            const int m = m0 + (wave * PIW + i) * RPI + rsub;
            rv[i] = m < p.M;
            const int mm = rv[i] ? m : 0;
            if (!(p.up0 | p.up1)) {
                off0[i] = (unsigned)mm * (unsigned)p.ld0;
                off1[i] = (unsigned)mm * (unsigned)p.ld1;
            } else {
                const int n = mm / ohw, r2 = mm - n * ohw, oh = r2 / p.OW, ow = r2 - oh * p.OW;
                const int H0 = p.H >> p.up0, W0 = p.W >> p.up0, H1 = p.H >> p.up1, W1 = p.W >> p.up1;
                off0[i] = (unsigned)((n * H0 + (oh >> p.up0)) * W0 + (ow >> p.up0)) * (unsigned)p.ld0;
                off1[i] = (unsigned)((n * H1 + (oh >> p.up1)) * W1 + (ow >> p.up1)) * (unsigned)p.ld1;
            }
        }
        s_cb = 0;
    };
    auto issue_step = [&](int stage) {
        if (ikt == 0) setup_tile(it);
        half_t* sP = smem + stage * STAGE;
        const bool s1 = s_cb >= p.C0;
        const unsigned sc = 2u * (unsigned)(s1 ? s_cb - p.C0 : s_cb) + 16u * (unsigned)kc;
#pragma unroll
        for (int i = 0; i < PIW; ++i) {
            const unsigned o = rv[i] ? 2u * (s1 ? off1[i] : off0[i]) + sc : BSY_OOB;
            if (s1) dma16_buf(rs1, o, 0u, sP + (wave * PIW + i) * RPI * BK);
            else dma16_buf(rs0, o, 0u, sP + (wave * PIW + i) * RPI * BK);
        }
        s_cb += BK;
        if (++ikt == nk) { ikt = 0; ++it; }
    };

    f32x16 acc[NT][MT];  // start at the bias of this cout tile (common.h acc_bias)
#pragma unroll
    for (int a = 0; a < NT; ++a)
#pragma unroll
        for (int b = 0; b < MT; ++b) acc_bias(acc[a][b], sbias + (wn * NT + a) * 32, lh);
    const int total = my_tiles * nk;
    int ip = 0;
    for (; ip < STAGES - 1 && ip < total; ++ip) issue_step(ip % STAGES);
    int ckt = 0, ct_done = 0;
    half_t* dst = reinterpret_cast<half_t*>(p.dst);
    for (int cp = 0; cp < total; ++cp) {
        // wait for K-step cp.  Younger in this thread's vector-memory queue: the DMA of the next min(STAGES - 2, remaining) steps (PIW
        // each) and, while cp is one of the first STAGES - 1 steps of a tile after the first, the NST stores of the previous tile
        // (they were issued after the DMA of step cp, which went out STAGES - 1 steps ago)
        const int later = min(STAGES - 2, total - 1 - cp);
        const bool st_young = ct_done > 0 && ckt < STAGES - 1;
        if (later >= STAGES - 2) {
            if (st_young) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((STAGES - 2) * PIW + NST) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" ::"n"((STAGES - 2) * PIW) : "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
        if (ip < total) { issue_step(ip % STAGES); ++ip; }
        const half_t* sP = smem + (cp % STAGES) * STAGE;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int chunk = 2 * ks + lh;
            half8 bfr[MT];
#pragma unroll
            for (int b = 0; b < MT; ++b) {
                const int row = (wm * MT + b) * 32 + lrow;
                bfr[b] = *reinterpret_cast<const half8*>(sP + row * BK + ((chunk ^ ((row >> 2) & 3)) << 3));
            }
            // the A fragment index depends on the K-step: a uniform switch over compile-time register indices
#pragma unroll
            for (int kk = 0; kk < NFR / (2 * NT); ++kk)
                if (kk == ckt) {
#pragma unroll
                    for (int a = 0; a < NT; ++a)
#pragma unroll
                        for (int b = 0; b < MT; ++b)
                            acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(afr[(2 * kk + ks) * NT + a], bfr[b], acc[a][b], 0, 0, 0);
                }
        }
        if (++ckt == nk) {
            ckt = 0;
            const int m0 = (pt0 + slot + ct_done * walkers) * TM;
            // bias + SiLU -> fp16 tile in LDS (the previous tile's stores have read it: their LDS reads completed before they were issued,
            // and every wave passed at least one barrier since)
#pragma unroll
            for (int b = 0; b < MT; ++b) {
                const int prow = (wm * MT + b) * 32 + lrow;
#pragma unroll
                for (int a = 0; a < NT; ++a)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int c = (wn * NT + a) * 32 + 8 * g + 4 * lh;
                        half4 o;
                        const f32x4 bn = *reinterpret_cast<const f32x4*>(sbias + c);  // the next tile's accumulators restart at the bias
                        f32x4 tv = f32x4{acc[a][b][4 * g], acc[a][b][4 * g + 1], acc[a][b][4 * g + 2], acc[a][b][4 * g + 3]};
                        if (p.act) tv = silu4_f(tv);
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            o[e] = (half_t)tv[e];
                            acc[a][b][4 * g + e] = bn[e];
                        }
                        *reinterpret_cast<half4*>(otile + prow * LDT + c) = o;
                    }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();  // (never __syncthreads(): it would drain the DMA ring)
            half8 v[NST];
#pragma unroll
            for (int i = 0; i < NST; ++i) {
                const int id = tid + 256 * i;
                v[i] = *reinterpret_cast<const half8*>(otile + (id / CPRW) * LDT + (id % CPRW) * 8);
            }
#pragma unroll
            for (int i = 0; i < NST; ++i) {
                const int id = tid + 256 * i;
                const int m = m0 + id / CPRW, c = n0 + (id % CPRW) * 8;
                // The counted waits assume NST store instructions per wave and tile.  A store whose predicate fails in SOME lanes is
                // still one instruction; one that fails in all lanes of a wave is skipped -- only possible in a tile that reaches past M,
                // which is the last tile of its walker (no wait follows it), since a cout tile always starts below Cout.
                if (m < p.M && c < p.Cout) *reinterpret_cast<half8*>(dst + (size_t)m * p.ldd + c) = v[i];
            }
            ++ct_done;
        }
    }
}

template <int NT, int STAGES, int NFR>
static int launch_wres(const ConvK& k, hipStream_t s) {
    constexpr int TM = 128, TN = 64 * NT;
    ConvK p = k;
    p.ntn = ceil_div(k.Cout, TN);
    p.ntm = ceil_div(k.M, TM);
    // two workgroups per CU; a multiple of 8 ntn so that every XCD runs the same number of walkers for every cout tile
    int grid = 512 / (8 * p.ntn) * (8 * p.ntn);
    if (grid < 8 * p.ntn) grid = 8 * p.ntn;
    const int walkers_per_xcd = grid / 8 / p.ntn;
    if (walkers_per_xcd > ceil_div(p.ntm, 8)) grid = 8 * p.ntn * (ceil_div(p.ntm, 8) > 0 ? ceil_div(p.ntm, 8) : 1);
    hipLaunchKernelGGL((conv1x1_wres_kernel<NT, STAGES, NFR>), dim3((unsigned)grid), dim3(256), 0, s, p);
    HIP_TRY(hipGetLastError());
    return BSY_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// Patch-based 3x3 stride-1 kernel (configuration tiles 10 / 11).
//
// The implicit-GEMM kernel above stages every input pixel NINE times (once per tap) through the vector-memory path and
// LDS; at 128 x 128 tiles that is 64 B/clk/CU of LDS-DMA -- the texture-addresser's peak -- plus the matching LDS
// writes, and measured MFMA utilisation stalls near 25-30 %.  A 3x3 stride-1 conv only needs each input pixel once
// per workgroup: here a workgroup owns an 8 x 16 output tile, keeps the 10 x 18 input patch of the current 32-channel
// chunk in LDS and reads all nine taps' B fragments from it (a tap is a constant entry offset); only the weights
// (3 taps x TN x 32 per (chunk, kernel row) step) stream through the DMA ring.  Vector-memory bytes per MFMA drop ~2.5x.
//   * K order: chunk-major, taps inside (the packed weights stay k = (kh, kw, c): tap (kh, kw) of a chunk reads K offset
//     (3 kh + kw) * Cin + 32 * chunk).  That differs from the implicit-GEMM kernel's tap-major order, so results agree with it
//     to fp32 accumulation rounding, not bit for bit.
//   * patch: 192 entries (180 used) x 64 B, filled by LDS-DMA 16 entries per wave-instruction; slot s of entry q holds
//     channel chunk s ^ ((q >> 2) & 3) (the DMA lane picks its source chunk), so 16 consecutive entries read
//     conflict-free.  Double-buffered: chunk c + 1 is fetched at tap 0 of chunk c.
//   * counted vmcnt: a wave has WIW weight DMAs per step in flight for STAGES - 1 steps, plus PIW patch DMAs issued at
//     kernel row 0 of every chunk.
// Restrictions (conv_cfg_valid): 3x3, stride 1, one source, no upsample, Cin % 32 == 0, fp16 output.
// ---------------------------------------------------------------------------------------------------------------------
#define CP_TW 16
#define CP_PW (CP_TW + 2)
// A K-step is one (32-channel chunk, kernel row): 3 taps x 32 channels = 6 MFMA sub-steps per wave tile.  (One tap per
// step left 4-8 MFMAs between barriers and a two-step prefetch distance far below the L2 latency.)
// WM = wave rows: 2 -> 8 x 16 output tile, 4 waves (the shipped form).  4 -> 16 x 16 tile, 8 waves: measured 8-25 % SLOWER
// on every YOLO11s layer (r01 notes in DESIGN.md) and therefore not instantiated.
// TW_ = tile width: 16 (8 x 16 tile: 128 pixels = the 128 pixel slots of the four MFMA pixel tiles) or 20 (6 x 20 tile: 120
// pixels in the 128 slots, slot i = pixel (i / 20, i % 20)) -- a 20 x 20 map is 4 tiles of 6 x 20 instead of 6 tiles of
// 8 x 16 (48 % of whose slots lie outside the map), a 40 x 40 map 14 instead of 15.
// TAIL (NT == 1, four waves): the Detect box branch's last two layers in one launch -- this 3x3 conv (64 couts), then the
// branch's 1x1 conv (64 -> 64 box logits) on the activated fp16 tile straight from LDS and the DFL decoder on its accumulators
// (ConvArgs::tail_wgt).  Wave w takes pixel slots 32 w .. 32 w + 31 and all 64 couts (the decoder's layout: a side's 16 bins in
// one lane pair).  Same operands, same K order as the separate 1x1 launch -> the same bits; the 64-channel map stays on chip.
template <int NT, int STAGES, int WM, int TW_ = CP_TW, bool TAIL = false>
__global__ __launch_bounds__(128 * WM) void conv3x3_patch_kernel(const ConvK p) {
    constexpr int CP_TH = TW_ == 16 ? 4 * WM : 6, NW = 2 * WM, NTHR = 64 * NW;
    constexpr int PW_ = TW_ + 2;                            // patch row length (entries)
    constexpr int NVALID = CP_TH * TW_;                     // pixels of the tile (<= TM slots)
    static_assert(TW_ == 16 || (TW_ == 20 && WM == 2), "tile shapes: 8 x 16, 16 x 16 (8 waves), 6 x 20");
    constexpr int CP_NPX = (CP_TH + 2) * PW_;               // 180 / 324 / 176 patch entries
    constexpr int PIW = (CP_NPX + 16 * NW - 1) / (16 * NW);  // patch DMA wave-instructions per wave per chunk (16 entries each)
    constexpr int CP_NPI = PIW * NW;
    constexpr int TN = 64 * NT, TM = 64 * WM;
    constexpr int PBUF = CP_NPI * 16 * 32;  // halves per patch buffer
    constexpr int WTAP = TN * 32;           // halves per tap of a weight stage
    constexpr int WST = 3 * WTAP;           // halves per weight stage (3 taps)
    constexpr int WIW = 3 * TN / 64;        // weight DMA wave-instructions per wave per step (4-wave form; counted waits)
    constexpr int OTILE = TM * (TN + 8);
    constexpr int RING = 2 * PBUF + STAGES * WST;
    constexpr int SMEM = RING > OTILE ? RING : OTILE;
    static_assert(STAGES == 2 || ((STAGES == 3 || STAGES == 4) && WM == 2), "ring depth (the 8-wave form deals the weight DMA unevenly: vmcnt(0) only)");
    // STAGES <= 4: the patch of chunk c is issued at kernel row 0 of chunk c - 1, i.e. before (or in the same step as, and then ahead
    // of) the weights of chunk c's first step as long as those are issued at most three steps ahead; a deeper ring would let that
    // step's wait pass with the patch still in flight.
    __shared__ __attribute__((aligned(16))) half_t smem[SMEM];
    half_t* sW = smem + 2 * PBUF;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;  // wave grid WM x 2: 64 pixels (4 tile rows) x NT*32 couts each
    const int lrow = lane & 31, lh = lane >> 5;
    const int wg = xcd_remap(blockIdx.x, gridDim.x);
    const int tn_idx = wg % p.ntn;
    int t = wg / p.ntn;
    const int tx = t % p.tiles_x;
    t /= p.tiles_x;
    const int ty = t % p.tiles_y;
    const int n = t / p.tiles_y;
    const int oy0 = ty * CP_TH, ox0 = tx * TW_, n0 = tn_idx * TN;

    const bsy_rsrc_t rs0 = make_rsrc(p.src0, p.span0), rsw = make_rsrc(p.wgt, p.wspan);
    // patch DMA coordinates: instruction i of this wave fills entries 16 (wave + 4 i) .. +15; lane -> (entry, slot)
    unsigned poff[PIW];
#pragma unroll
    for (int i = 0; i < PIW; ++i) {
        const int q = 16 * (wave + NW * i) + (lane >> 2);
        const int pr = q / PW_, pc = q - pr * PW_;
        const int y = oy0 - 1 + pr, x = ox0 - 1 + pc;
        const int chunk = (lane & 3) ^ ((q >> 2) & 3);
        poff[i] = (q < CP_NPX && (unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)p.W)
                      ? 2u * ((unsigned)((n * p.H + y) * p.W + x) * (unsigned)p.ld0 + 8u * chunk) : BSY_OOB;
    }
    // weight DMA coordinates: the stage is [tap kw][TN rows][64 B] = 3 * TN/16 wave-instructions of 16 rows.  Four waves:
    // wave w takes row groups w, w+4, .. of every tap.  Eight waves: row group g of tap kw goes to wave (g & 3) + 4 * (kw & 1)
    // (waves 0-3 serve taps 0 and 2, waves 4-7 tap 1).
    constexpr int WPT = TN / 64;  // row groups per wave per tap
    unsigned woff[WPT];
#pragma unroll
    for (int j = 0; j < WPT; ++j) {
        const int r = 16 * (4 * j + (wave & 3)) + (lane >> 2);
        woff[j] = 2u * ((unsigned)(n0 + r) * (unsigned)p.Kpad + 8u * ((lane & 3) ^ ((r >> 2) & 3)));
    }
    const int Cin = p.Cin8 * 8;
    const int nchunks = Cin >> 5, nsteps = 3 * nchunks;

    auto issue_patch = [&](int chunk) {
        half_t* dst = smem + (chunk & 1) * PBUF;
#pragma unroll
        for (int i = 0; i < PIW; ++i) dma16_buf(rs0, poff[i], 64u * (unsigned)chunk, dst + (wave + NW * i) * 512);
    };
    int w_kh = 0, w_chunk = 0;  // (chunk, kernel row) of the next weight step to issue
    auto issue_weights = [&](int step) {
        half_t* dst = sW + (step % STAGES) * WST;
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
            if (WM == 4 && (kw & 1) != (wave >> 2)) continue;  // wave-uniform
            const unsigned koff = 2u * (unsigned)((w_kh * 3 + kw) * Cin + 32 * w_chunk);
#pragma unroll
            for (int j = 0; j < WPT; ++j) dma16_buf(rsw, woff[j], koff, dst + kw * WTAP + (4 * j + (wave & 3)) * 512);
        }
        if (++w_kh == 3) { w_kh = 0; ++w_chunk; }
    };

    f32x16 acc[NT][2];  // start at the bias of their couts (common.h acc_bias)
#pragma unroll
    for (int a = 0; a < NT; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc_bias(acc[a][b], p.bias + n0 + (wn * NT + a) * 32, lh);
    // lane pixel of MFMA tile b: tile row wm*4 + 2b + (lrow >> 4), column lrow & 15 -> patch entry of tap (0, 0)
    int lq[2];
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        const int slot = (wm * 2 + b) * 32 + lrow;  // pixel slot -> pixel (slot / TW, slot % TW); slots past the tile read entry 0
        lq[b] = slot < NVALID ? (slot / TW_) * PW_ + slot % TW_ : 0;
    }
    int arow[NT], asw[NT];
#pragma unroll
    for (int a = 0; a < NT; ++a) {
        arow[a] = ((wn * NT + a) * 32 + lrow) * 32;
        asw[a] = (lh ^ ((lrow >> 2) & 3)) << 3;
    }

    issue_patch(0);
#pragma unroll
    for (int s = 0; s < STAGES - 1; ++s)
        if (s < nsteps) issue_weights(s);
    int kh = 0, chunk = 0;
    for (int k = 0; k < nsteps; ++k) {
        // wait for W(k) (and everything older: this chunk's patch); younger: W(k+1 .. k+STAGES-2) and, when this is the
        // kernel row after a patch issue, the next chunk's patch
        const int younger = min(STAGES - 2, nsteps - 1 - k);
        // the next chunk's patch is issued at kernel row 0, behind W(k + STAGES - 2) of that step: it is younger than W(k) during the
        // following STAGES - 2 steps
        const bool p_young = STAGES > 2 ? (kh >= 1 && kh <= STAGES - 2 && chunk + 1 < nchunks) : false;
        if (younger <= 0) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else if (STAGES > 3 && younger >= 2) {
            if (p_young) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * WIW + PIW) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * WIW) : "memory");
        } else if (p_young) {
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WIW + PIW) : "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WIW) : "memory");
        }
        __builtin_amdgcn_s_barrier();  // step k visible to every wave; every wave is done reading step k-1
        if (kh == 0 && chunk + 1 < nchunks) issue_patch(chunk + 1);
        if (k + STAGES - 1 < nsteps) issue_weights(k + STAGES - 1);
        const half_t* sP = smem + (chunk & 1) * PBUF;
        const half_t* sWk = sW + (k % STAGES) * WST;
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
            const int toff = kh * PW_ + kw;
            half8 bfr[2][2], afr[2][NT];
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                const int q = lq[b] + toff;
                const int s0 = lh ^ ((q >> 2) & 3);
                const half_t* e = sP + q * 32;
                bfr[0][b] = *reinterpret_cast<const half8*>(e + (s0 << 3));
                bfr[1][b] = *reinterpret_cast<const half8*>(e + ((s0 ^ 2) << 3));
            }
#pragma unroll
            for (int a = 0; a < NT; ++a) {
                const half_t* e = sWk + kw * WTAP + arow[a];
                afr[0][a] = *reinterpret_cast<const half8*>(e + asw[a]);
                afr[1][a] = *reinterpret_cast<const half8*>(e + (asw[a] ^ 16));
            }
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int a = 0; a < NT; ++a)
#pragma unroll
                    for (int b = 0; b < 2; ++b)
                        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(afr[ks][a], bfr[ks][b], acc[a][b], 0, 0, 0);
        }
        if (++kh == 3) { kh = 0; ++chunk; }
    }
    // TAIL: the 1x1 weights (A operand, [128][64] packed, k = channel; rows = box logits a * 32 + lrow) are fetched now, so that
    // their latency passes under the SiLU / LDS stage below (every DMA has drained: the loop ends on vmcnt(0))
    half8 ta[TAIL ? 4 : 1][2];
    if (TAIL) {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int a = 0; a < 2; ++a)
                ta[TAIL ? ks : 0][a] = *reinterpret_cast<const half8*>(p.tail_wgt + (size_t)(a * 32 + lrow) * 64 + 16 * ks + 8 * lh);
    }
    __syncthreads();  // every wave has finished reading the last step: LDS becomes the output tile

    constexpr int LDT = TN + 8;
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        const int prow = (wm * 2 + b) * 32 + lrow;  // = tile row * 16 + column
#pragma unroll
        for (int a = 0; a < NT; ++a) {
            const int cl = (wn * NT + a) * 32;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int c = cl + 8 * g + 4 * lh;
                half4 o;
                f32x4 tv = f32x4{acc[a][b][4 * g], acc[a][b][4 * g + 1], acc[a][b][4 * g + 2], acc[a][b][4 * g + 3]};
                if (p.act) tv = silu4_f(tv);
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = (half_t)tv[e];
                *reinterpret_cast<half4*>(smem + prow * LDT + c) = o;
            }
        }
    }
    __syncthreads();
    if (TAIL) {
        static_assert(!TAIL || (NT == 1 && WM == 2), "tail: 64-cout tiles, four waves");
        const int slot = wave * 32 + lrow;
        f32x16 t0, t1;  // box logits 0..31 / 32..63 of this lane's pixel, starting at the 1x1 conv's bias
        acc_bias(t0, p.tail_bias, lh);
        acc_bias(t1, p.tail_bias + 32, lh);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const half8 tb = *reinterpret_cast<const half8*>(smem + slot * LDT + 16 * ks + 8 * lh);
            t0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ta[TAIL ? ks : 0][0], tb, t0, 0, 0, 0);
            t1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ta[TAIL ? ks : 0][1], tb, t1, 0, 0, 0);
        }
        const int oy = oy0 + slot / TW_, ox = ox0 + slot % TW_;
        const bool mv = slot < NVALID && oy < p.H && ox < p.W;
        const int pix = mv ? oy * p.W + ox : 0;
        const int hw = p.H * p.W;
        dfl_decode_store(p, t0, t1, mv, (size_t)n * p.nrows * p.A + p.a0 + pix, (size_t)n * p.rawC * hw + pix, pix, lh);
        return;
    }
    constexpr int CPRW = TN / 8, ITER = TM * CPRW / NTHR;
    half_t* dst = reinterpret_cast<half_t*>(p.dst);
#pragma unroll
    for (int i = 0; i < ITER; ++i) {
        const int id = tid + NTHR * i;
        const int prow = id / CPRW, cc = (id % CPRW) * 8;
        const int oy = oy0 + prow / TW_, ox = ox0 + prow % TW_, c = n0 + cc;
        if (prow >= NVALID || oy >= p.H || ox >= p.W || c >= p.Cout) continue;
        const size_t pix = (size_t)(n * p.H + oy) * p.W + ox;
        half8 v = *reinterpret_cast<const half8*>(smem + prow * LDT + cc);
        if (p.res) {
            const half8 r = *reinterpret_cast<const half8*>(p.res + pix * p.ldr + c);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = (half_t)((float)v[e] + (float)r[e]);
        }
        *reinterpret_cast<half8*>(dst + pix * p.ldd + c) = v;
    }
}

template <int NT, int STAGES, int WM, int TW_ = CP_TW, bool TAIL = false>
static int launch_patch(const ConvK& k, hipStream_t s) {
    ConvK p = k;
    p.ntn = ceil_div(k.Cout, 64 * NT);
    p.tiles_x = ceil_div(k.W, TW_);
    p.tiles_y = ceil_div(k.H, TW_ == 16 ? 4 * WM : 6);
    const long long nblk = (long long)k.B * p.tiles_x * p.tiles_y * p.ntn;
    if (nblk <= 0 || nblk > 0x7fffffffLL) BSY_FAIL(BSY_ERR_ARG, "conv: tile count %lld out of range", nblk);
    hipLaunchKernelGGL((conv3x3_patch_kernel<NT, STAGES, WM, TW_, TAIL>), dim3((unsigned)nblk), dim3(128 * WM), 0, s, p);
    HIP_TRY(hipGetLastError());
    return BSY_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// Fused DWConv 3x3 (+BN+SiLU) -> Conv 1x1 (+BN+SiLU): the two-layer units of YOLO11's class branch (head.py:49-57:
// nn.Sequential(DWConv(x, x, 3), Conv(x, c3, 1))).  Unfused, the depthwise map makes a round trip through HBM and the
// pair costs two launches (0.12 ms per pair at 80 x 80 x 128, B = 64).  Here, per 8 x 16 output tile and 32-channel chunk:
//   * the 10 x 18 input patch arrives by LDS-DMA exactly as in the patch kernel above (one chunk ahead);
//   * the VALU computes the depthwise conv + SiLU of the chunk from the patch (f32, same tap order as dwconv3x3_kernel ->
//     bit-identical values) and parks it as the [128 px][32 ch] B-operand tile in the layout the MFMA stage reads;
//   * the MFMAs multiply it with the 1x1 weights of the chunk (DMA ring of 2).
// The kernel is bound by the depthwise VALU work (9 FMA + SiLU per element) -- the 1x1 conv rides along for free.
// Depthwise weights [9][C] + bias [C] (f32) sit in LDS for the whole launch (C <= 256).
// ---------------------------------------------------------------------------------------------------------------------
struct DwPwK {
    const half_t* src;
    int ld0, B, H, W, C;
    unsigned span0, wspan;
    const float* dww;   // [9][C]
    const float* dwb;   // [C]
    const half_t* wgt;  // packed [CoutPad][Kpad]
    const float* bias;
    half_t* dst;
    int ldd, Cout, Kpad, act, tiles_x, tiles_y, ntn;
};
#define DWPW_MAXC 256
// LDS: 2 patch buffers (chunk c + 1 lands while chunk c is processed), ONE weight stage (chunk c's 1x1 weights are issued
// after the barrier that retires chunk c - 1's MFMAs and are only needed after the depthwise stage, which is longer than
// their latency), the B-operand tile, and the depthwise weights / bias (dynamic: 10 * C floats) -> 46 KiB + 40 C bytes:
// three workgroups per CU for C <= 128.
template <int NT>
__global__ __launch_bounds__(256, NT == 2 ? 3 : 2) void dwpw_fused_kernel(const DwPwK p) {
    constexpr int TN = 64 * NT, TM = 128;
    constexpr int PIW = 3, NPI = 12, NPX = 10 * CP_PW;  // patch: 180 entries in 12 DMA wave-instructions
    constexpr int PBUF = NPI * 16 * 32, WST = TN * 32, PT = TM * 32;
    constexpr int WIW = TN / 64;
    constexpr int OTILE = TM * (TN + 8);
    constexpr int RING = 2 * PBUF + WST + PT;
    constexpr int SMEM = RING > OTILE ? RING : OTILE;
    __shared__ __attribute__((aligned(16))) half_t smem[SMEM];
    extern __shared__ __attribute__((aligned(16))) float sdw[];  // [9][C] then [C]
    half_t* sW = smem + 2 * PBUF;
    half_t* sPt = sW + WST;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int lrow = lane & 31, lh = lane >> 5;
    const int wg = xcd_remap(blockIdx.x, gridDim.x);
    const int tn_idx = wg % p.ntn;
    int t = wg / p.ntn;
    const int tx = t % p.tiles_x;
    t /= p.tiles_x;
    const int ty = t % p.tiles_y;
    const int n = t / p.tiles_y;
    const int oy0 = ty * 8, ox0 = tx * CP_TW, n0 = tn_idx * TN;
    const int C = p.C, nchunks = C >> 5;

    for (int i = tid; i < 10 * C; i += 256) sdw[i] = i < 9 * C ? p.dww[i] : p.dwb[i - 9 * C];
    const bsy_rsrc_t rs0 = make_rsrc(p.src, p.span0), rsw = make_rsrc(p.wgt, p.wspan);
    unsigned poff[PIW];
#pragma unroll
    for (int i = 0; i < PIW; ++i) {
        const int q = 16 * (wave + 4 * i) + (lane >> 2);
        const int pr = q / CP_PW, pc = q - pr * CP_PW;
        const int y = oy0 - 1 + pr, x = ox0 - 1 + pc;
        const int chunk = (lane & 3) ^ ((q >> 2) & 3);
        poff[i] = (q < NPX && (unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)p.W)
                      ? 2u * ((unsigned)((n * p.H + y) * p.W + x) * (unsigned)p.ld0 + 8u * chunk) : BSY_OOB;
    }
    unsigned woff[WIW];
#pragma unroll
    for (int j = 0; j < WIW; ++j) {
        const int r = 16 * (4 * j + wave) + (lane >> 2);
        woff[j] = 2u * ((unsigned)(n0 + r) * (unsigned)p.Kpad + 8u * ((lane & 3) ^ ((r >> 2) & 3)));
    }
    auto issue_patch = [&](int c) {
        half_t* dp = smem + (c & 1) * PBUF;
#pragma unroll
        for (int i = 0; i < PIW; ++i) dma16_buf(rs0, poff[i], 64u * (unsigned)c, dp + (wave + 4 * i) * 512);
    };
    auto issue_weights = [&](int c) {
#pragma unroll
        for (int j = 0; j < WIW; ++j) dma16_buf(rsw, woff[j], 64u * (unsigned)c, sW + (4 * j + wave) * 512);
    };

    f32x16 acc[NT][2];  // the 1x1 conv's accumulators start at its bias (common.h acc_bias)
#pragma unroll
    for (int a = 0; a < NT; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc_bias(acc[a][b], p.bias + n0 + (wn * NT + a) * 32, lh);
    // depthwise stage: this thread's two items = pixels (tid >> 2) and (tid >> 2) + 64 (4 tile rows apart), 8-channel
    // piece (tid & 3).  LDS offsets (halves) of the item's nine taps, swizzle included, are chunk-independent.
    const int ch8 = tid & 3;
    int tapoff[9];
    {
        const int px = tid >> 2;
        const int q0 = (px >> 4) * CP_PW + (px & 15);
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            const int q = q0 + (k / 3) * CP_PW + k % 3;
            tapoff[k] = q * 32 + ((ch8 ^ ((q >> 2) & 3)) << 3);
        }
    }
    // the second item sits 4 rows = 72 entries further: 72 % 16 == 8 -> (q >> 2) & 3 flips bit 1 -> slot ^ 2 -> +-16 halves
    int ptoff[2];
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int row = (tid >> 2) + 64 * it;
        ptoff[it] = row * 32 + ((ch8 ^ ((row >> 2) & 3)) << 3);
    }

    issue_patch(0);
    for (int c = 0; c < nchunks; ++c) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();  // patch c landed; every wave is done with chunk c - 1 (its weights, tile and patch)
        if (c + 1 < nchunks) issue_patch(c + 1);
        issue_weights(c);
        const half_t* sP = smem + (c & 1) * PBUF;
        {   // depthwise 3x3 + bias + SiLU of the chunk -> B-operand tile
            const float* wk = sdw + 32 * c + 8 * ch8;
            const f32x4 b0 = *reinterpret_cast<const f32x4*>(sdw + 9 * C + 32 * c + 8 * ch8), b1 = *reinterpret_cast<const f32x4*>(sdw + 9 * C + 32 * c + 8 * ch8 + 4);
            float a[2][8];
#pragma unroll
            for (int it = 0; it < 2; ++it)
#pragma unroll
                for (int j = 0; j < 4; ++j) { a[it][j] = b0[j]; a[it][4 + j] = b1[j]; }
#pragma unroll
            for (int kw = 0; kw < 3; ++kw)
#pragma unroll
                for (int kh = 0; kh < 3; ++kh) {  // same order as dwconv3x3_kernel: (top, mid, bot) of column kw
                    const int k = kh * 3 + kw;
                    const f32x4 w0 = *reinterpret_cast<const f32x4*>(wk + k * C), w1 = *reinterpret_cast<const f32x4*>(wk + k * C + 4);
#pragma unroll
                    for (int it = 0; it < 2; ++it) {
                        // item 1 = item 0 + 72 entries: (q >> 2) & 3 changes by 2 (72 / 4 = 18) -> slot ^ 2
                        const half8 v = *reinterpret_cast<const half8*>(sP + (it ? ((tapoff[k] + 72 * 32) ^ 16) : tapoff[k]));
                        fma_mix8(a[it], v, w0, w1);
                    }
                }
#pragma unroll
            for (int it = 0; it < 2; ++it) {
                half8 o;
                const f32x4 s0 = silu4_f(f32x4{a[it][0], a[it][1], a[it][2], a[it][3]}), s1 = silu4_f(f32x4{a[it][4], a[it][5], a[it][6], a[it][7]});
#pragma unroll
                for (int j = 0; j < 4; ++j) { o[j] = (half_t)s0[j]; o[4 + j] = (half_t)s1[j]; }
                *reinterpret_cast<half8*>(sPt + ptoff[it]) = o;
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this chunk's 1x1 weights (and the next patch) have landed
        __syncthreads();  // B-operand tile complete, weights visible
        half8 bfr[2][2], afr[2][NT];
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int row = (wm * 2 + b) * 32 + lrow;
            const int s0 = lh ^ ((row >> 2) & 3);
            bfr[0][b] = *reinterpret_cast<const half8*>(sPt + row * 32 + (s0 << 3));
            bfr[1][b] = *reinterpret_cast<const half8*>(sPt + row * 32 + ((s0 ^ 2) << 3));
        }
#pragma unroll
        for (int a_ = 0; a_ < NT; ++a_) {
            const int row = (wn * NT + a_) * 32 + lrow;
            const int s0 = lh ^ ((row >> 2) & 3);
            afr[0][a_] = *reinterpret_cast<const half8*>(sW + row * 32 + (s0 << 3));
            afr[1][a_] = *reinterpret_cast<const half8*>(sW + row * 32 + ((s0 ^ 2) << 3));
        }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int a_ = 0; a_ < NT; ++a_)
#pragma unroll
                for (int b = 0; b < 2; ++b)
                    acc[a_][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(afr[ks][a_], bfr[ks][b], acc[a_][b], 0, 0, 0);
    }
    __syncthreads();

    constexpr int LDT = TN + 8;
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        const int prow = (wm * 2 + b) * 32 + lrow;
#pragma unroll
        for (int a_ = 0; a_ < NT; ++a_) {
            const int cl = (wn * NT + a_) * 32;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int cc = cl + 8 * g + 4 * lh;
                half4 o;
                f32x4 tv = f32x4{acc[a_][b][4 * g], acc[a_][b][4 * g + 1], acc[a_][b][4 * g + 2], acc[a_][b][4 * g + 3]};
                if (p.act) tv = silu4_f(tv);
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = (half_t)tv[e];
                *reinterpret_cast<half4*>(smem + prow * LDT + cc) = o;
            }
        }
    }
    __syncthreads();
    constexpr int CPRW = TN / 8, ITER = TM * CPRW / 256;
#pragma unroll
    for (int i = 0; i < ITER; ++i) {
        const int id = tid + 256 * i;
        const int prow = id / CPRW, cc = (id % CPRW) * 8;
        const int oy = oy0 + (prow >> 4), ox = ox0 + (prow & 15), cgl = n0 + cc;
        if (oy >= p.H || ox >= p.W || cgl >= p.Cout) continue;
        *reinterpret_cast<half8*>(p.dst + ((size_t)(n * p.H + oy) * p.W + ox) * p.ldd + cgl) =
            *reinterpret_cast<const half8*>(smem + prow * LDT + cc);
    }
}

bool dwpw_fused_supported(int C, int Cout) { return C > 0 && !(C & 31) && C <= DWPW_MAXC && Cout > 0 && !(Cout & 7); }

int launch_dwpw_fused(const DwPwArgs& a, hipStream_t s) {
    if (!dwpw_fused_supported(a.C, a.Cout)) BSY_FAIL(BSY_ERR_ARG, "dwpw: unsupported widths (C %d, Cout %d): C %% 32 == 0, C <= %d, Cout %% 8 == 0", a.C, a.Cout, DWPW_MAXC);
    if (!a.src || !a.dst || !a.dww || !a.dwb || !a.wgt || !a.bias) BSY_FAIL(BSY_ERR_ARG, "dwpw: null pointer");
    if ((a.lds & 7) || (a.ldd & 7) || a.lds < a.C || a.ldd < a.Cout ||
        (((uintptr_t)a.src | (uintptr_t)a.dst | (uintptr_t)a.dww | (uintptr_t)a.dwb | (uintptr_t)a.wgt | (uintptr_t)a.bias) & 15))
        BSY_FAIL(BSY_ERR_ARG, "dwpw: misaligned pointer / leading dimension");
    if ((long long)a.B * a.H * a.W * a.lds >= (1LL << 31)) BSY_FAIL(BSY_ERR_ARG, "dwpw: source view exceeds 2^31 elements (split the batch)");
    DwPwK k;
    k.src = a.src; k.ld0 = a.lds; k.B = a.B; k.H = a.H; k.W = a.W; k.C = a.C;
    k.span0 = (unsigned)((((long long)a.B * a.H * a.W - 1) * a.lds + a.C) * 2);
    k.dww = a.dww; k.dwb = a.dwb; k.wgt = (const half_t*)a.wgt; k.bias = a.bias; k.dst = a.dst; k.ldd = a.ldd; k.Cout = a.Cout;
    k.Kpad = round_up(a.C, 32); k.act = a.act;
    k.wspan = (unsigned)((size_t)round_up(a.Cout, 128) * k.Kpad * 2);
    k.tiles_x = ceil_div(a.W, CP_TW); k.tiles_y = ceil_div(a.H, 8);
    const bool wide = a.Cout > 64;
    k.ntn = ceil_div(a.Cout, wide ? 128 : 64);
    const long long nblk = (long long)a.B * k.tiles_x * k.tiles_y * k.ntn;
    if (nblk <= 0 || nblk > 0x7fffffffLL) BSY_FAIL(BSY_ERR_ARG, "dwpw: tile count out of range");
    const size_t dyn = (size_t)10 * a.C * sizeof(float);  // depthwise weights + bias
    if (wide) hipLaunchKernelGGL((dwpw_fused_kernel<2>), dim3((unsigned)nblk), dim3(256), dyn, s, k);
    else hipLaunchKernelGGL((dwpw_fused_kernel<1>), dim3((unsigned)nblk), dim3(256), dyn, s, k);
    HIP_TRY(hipGetLastError());
    return BSY_OK;
}

// Split-K second step: out[pixel][c] = act(sum over slices, IN SLICE ORDER, of the raw f32 sums) (+ shortcut) -> f16 NHWC.  One thread
// per (pixel, 8 couts); the order of the adds is fixed by the slice index, so the result does not depend on which workgroup finished
// first, on the tile configuration or on the batch.
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ ws, const int nsl, const long long slab, const int ldw,
                                                           const int M, const int Cout, const int act, const half_t* __restrict__ res, const int ldr,
                                                           half_t* __restrict__ dst, const int ldd) {
    const int c8n = Cout >> 3;
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long long)M * c8n) return;
    const int c = (int)(idx % c8n) * 8;
    const size_t m = (size_t)(idx / c8n);
    const float* wp = ws + m * ldw + c;
    f32x4 a0 = *reinterpret_cast<const f32x4*>(wp), a1 = *reinterpret_cast<const f32x4*>(wp + 4);
    for (int sl = 1; sl < nsl; ++sl) {
        const float* q = wp + (size_t)sl * (size_t)slab;
        a0 = add4_f(a0, *reinterpret_cast<const f32x4*>(q));
        a1 = add4_f(a1, *reinterpret_cast<const f32x4*>(q + 4));
    }
    float v[8] = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
    if (act) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = silu_f(v[e]);
    }
    if (res) {
        const half8 rv = *reinterpret_cast<const half8*>(res + m * ldr + c);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] += (float)rv[e];
    }
    half8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (half_t)v[e];
    *reinterpret_cast<half8*>(dst + m * ldd + c) = o;
}

template <int KS, int WM, int WN, int MT, int NT, int STAGES, bool ALIGNED, int BK>
static int launch_cfg(const ConvK& k, hipStream_t s) {
    constexpr int TM = WM * MT * 32, TN = WN * NT * 32;
    ConvK p = k;
    p.ntn = ceil_div(k.Cout, TN);
    const int nsl = k.nsl_c * k.nsl_t;
    if (nsl > 1 && (!ALIGNED || (k.cb_per % BK))) BSY_FAIL(BSY_ERR_ARG, "conv: split-K needs an aligned configuration whose K-step divides the channel slices");
    const long long nblk = (long long)ceil_div(k.M, TM) * p.ntn * (nsl > 1 ? nsl : 1);
    if (nblk <= 0 || nblk > 0x7fffffffLL) BSY_FAIL(BSY_ERR_ARG, "conv: grid %lld out of range", nblk);
    hipLaunchKernelGGL((conv_mfma_kernel<KS, WM, WN, MT, NT, STAGES, ALIGNED, BK>), dim3((unsigned)nblk), dim3(64 * WM * WN), 0, s, p);
    HIP_TRY(hipGetLastError());
    if (nsl > 1) {
        const long long items = (long long)k.M * (k.Cout >> 3);
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)((items + 255) / 256)), dim3(256), 0, s, k.split_ws, nsl, k.slab, k.ldw, k.M, k.Cout, k.act,
                           k.res, k.ldr, reinterpret_cast<half_t*>(k.dst), k.ldd);
        HIP_TRY(hipGetLastError());
    }
    return BSY_OK;
}

extern "C" int bsy_conv_packed_dims(int C2, int C1, int ksize, int* cout_pad, int* k_pad) {
    if (C2 <= 0 || C1 <= 0 || (ksize != 1 && ksize != 3)) BSY_FAIL(BSY_ERR_ARG, "conv_packed_dims: bad shape");
    if (cout_pad) *cout_pad = round_up(C2, 128);
    if (C1 == 3) C1 = 4;  // image convs are packed with a zero 4th input channel (image_conv.h)
    if (k_pad) *k_pad = round_up(ksize * ksize * C1, 32);
    return BSY_OK;
}

// The K walk of a layer -- the order in which its products enter the fp32 accumulators, hence its result's bits -- is a function of
// the layer's SHAPE alone; every kernel configuration that is valid for the layer sums in that order (round 3: a tuned plan
// returns bit for bit what the heuristic plan returns, on any box).
//   0: the packed order (taps outer, channels inner): 1x1 convs (one tap) and unaligned channel counts (generic variant only);
//   1: chunk-major, 32-channel chunks: aligned 3x3 layers -- the patch kernel's order; implicit-GEMM tiles walk it with BK 32;
//   2: chunk-major, 64-channel chunks: aligned 3x3 STRIDE-2 layers with Cin % 64 == 0 (no patch kernel there; the BK-64 tiles
//      these layers run fastest on sum in this order, BK-32 tiles visit the two halves of a chunk tap by tap).
int conv_korder(const ConvArgs& a) {
    const int Cin = a.C0 + a.C1;
    if (a.ksize != 3 || (Cin & 31) || (a.C0 & 31)) return 0;
    return (a.stride == 2 && !(Cin & 63) && !(a.C0 & 63)) ? 2 : 1;
}

// Configuration ids: tile << 4 | variant.
//   tile   : 0 = 256 px x 32 couts, 1 = 256 x 64, 2 = 128 x 128, 3 = 128 x 64 (4 waves); 4 = 256 x 128 (8 waves);
//            5 = 64 x 128, 6 = 64 x 64 (4 waves; small tiles = many resident workgroups for the latency-bound thin-K layers);
//            7 = 256 x 256 (8 waves, wave tile 64 x 128: half the L2->LDS bytes per FLOP of 128 x 128);
//            8 = 128 x 128, 9 = 128 x 64 persistent 1x1 kernel (4 compute + 4 store waves, tiles walked per workgroup);
//            10 = 8x16 px x 128 couts (2-stage weight ring), 11 = 8x16 px x 64 couts (variant 1: 3 stages, 2: 2 stages, 3: 4 stages):
//            patch-based 3x3 stride-1 kernel; 12 / 13 = the same with 6x20-pixel tiles (maps whose width is a multiple of 20);
//            14 / 15 = weights-resident streaming 1x1 kernel, 128 px x 128 / 64 couts (variant 1: 4-stage pixel ring, 2: 3 stages)
//   variant: 0 = generic (per-lane K bookkeeping, flat DMA, BK 32, 3 stages), 1 = aligned BK 32 / 3 stages,
//            2 = aligned BK 32 / 2 stages, 3 = aligned BK 64 / 2 stages; + 4 on the implicit-GEMM tiles of a layer whose K walk is
//            chunk-major (conv_korder != 0): part of the id so that ids recorded before round 3 for the packed order on such a
//            layer are recognised as stale (invalid -> heuristic)
bool conv_cfg_valid(const ConvArgs& a, int cfg) {
    const int Cin = a.C0 + a.C1, tile = cfg >> 4, var = cfg & 3, kbit = (cfg >> 2) & 3;
    const int ko = conv_korder(a);
    if (cfg < 0 || tile > 15 || kbit > 1) return false;
    // the id's chunk-major bit must say what the layer's shape says (implicit-GEMM tiles; the other kernels have one order each)
    if (tile < 8 && kbit != (ko ? 1 : 0)) return false;
    if (tile >= 8 && kbit) return false;
    if (ko && tile < 8 && var < 1) return false;
    if (ko == 1 && tile < 8 && var == 3) return false;  // 32-channel chunks: a 64-deep K-step would pair the halves of two chunks
    if (a.tail_wgt) {  // box-branch tail: the 64-cout patch tiles only (validity of the 3x3 conv itself: the same op without tail / decoder)
        if (tile != 11 && tile != 13) return false;
        ConvArgs b = a;
        b.tail_wgt = nullptr; b.tail_bias = nullptr; b.epi = 0;
        return a.epi == 3 && a.Cout == 64 && !a.res && conv_cfg_valid(b, cfg);
    }
    if (a.epi && tile >= 8) return false;           // fused decoder: implicit-GEMM kernel only
    if (a.nsl_c * a.nsl_t > 1) {                    // split-K: implicit-GEMM tiles, aligned variants, K-step dividing the channel slices
        const int cbp = Cin / (a.nsl_c > 0 ? a.nsl_c : 1);
        if (tile >= 8 || var < 1 || (var == 3 && (cbp & 63)) || (cbp & 31)) return false;
    }
    if (tile >= 14) {  // weights-resident streaming 1x1 kernel (14: 128-cout tiles, 15: 64-cout tiles; variant 1: 4-stage pixel ring, 2: 3 stages)
        const int K = Cin, nt = tile == 14 ? 2 : 1, stages = var == 1 ? 4 : 3;
        return a.ksize == 1 && a.stride == 1 && (var == 1 || var == 2) && !(Cin & 31) && !(a.C0 & 31) && !a.out_f32 && !a.res && !(a.Cout & 7) &&
               !(a.ldd & 7) && !((uintptr_t)a.dst & 15) && a.dst_scale <= 1 && nt * (K / 16) <= (nt == 2 ? 24 : 32) && K / 32 >= stages - 1 &&  // (128-cout tiles with 32 fragments spill: not built)
              
               (tile == 14 ? a.Cout > 64 : true);
    }
    if (a.epi == 3 && tile != 1) return false;      // DFL needs all 64 box couts in one wave: the 256 x 64 tile (4 x 1 waves, NT 2)
    if (tile >= 12) {  // 6x20-pixel patch tiles
        ConvArgs b = a;
        return conv_cfg_valid(b, ((tile - 2) << 4) | var);
    }
    if (tile >= 10) {  // patch-based 3x3 stride-1 kernel (TN 128 / 64): sums chunk-major over 32-channel chunks = K walk 1
        return ko == 1 && (var == 1 || ((var == 2 || var == 3) && tile == 11)) && a.ksize == 3 && a.stride == 1 && a.pad == 1 && !a.C1 && !a.up0 && !(a.C0 & 31) && !a.out_f32 &&
               !(a.Cout & 7) && !(a.ldd & 7) && !((uintptr_t)a.dst & 15) && a.dst_scale <= 1 &&
               (!a.res || (!(a.ldr & 7) && !((uintptr_t)a.res & 15))) && (tile == 10 ? a.Cout > 64 : true);
    }
    if (tile >= 8) {  // persistent 1x1 kernel with store waves
        const bool aligned_ = !(Cin & 31) && !(a.C0 & 31), aligned64_ = !(Cin & 63) && !(a.C0 & 63);
        return a.ksize == 1 && aligned_ && var >= 1 && (var != 3 || aligned64_) && !a.out_f32 && !a.res && !(a.Cout & 7) && a.Cout <= 1024 &&
               !(a.ldd & 7) && !((uintptr_t)a.dst & 15) && a.dst_scale <= 1 && a.stride == 1;
    }
    if (tile == 7 && ((a.Cout & 255) || var < 1)) return false;  // 256x256: whole 256-cout tiles only (variant 3: BK 64, 128 KiB of LDS)
    const bool aligned = !(Cin & 31) && !(a.C0 & 31), aligned64 = !(Cin & 63) && !(a.C0 & 63);
    if (var >= 1 && !aligned) return false;
    if (var == 3 && !aligned64) return false;
    if (tile == 4 && var < 1) return false;  // 8-wave 256 x 128: aligned variants only (3 = BK 64, 96 KiB of LDS)
    return true;
}

int conv_candidates(const ConvArgs& a, int* out, int max_out) {
    const int Cin = a.C0 + a.C1;
    const bool aligned = !(Cin & 31) && !(a.C0 & 31), aligned64 = !(Cin & 63) && !(a.C0 & 63);
    const long long M = (long long)a.B * a.OH * a.OW;
    const int tile = a.epi == 3 ? 1 : (a.Cout > 64 ? 2 : (a.Cout > 32 ? 1 : 0));  // heuristic default: widest cout tile that is not wasted
    const int ko = conv_korder(a), km = ko ? 4 : 0;
    int n = 0;
    auto add = [&](int t, int v) {
        const int c = (t << 4) | v;
        for (int i = 0; i < n; ++i) if (out[i] == c) return;
        if (n < max_out && conv_cfg_valid(a, c)) out[n++] = c;
    };
    if (a.tail_wgt) {  // 3x3 conv + 1x1 + DFL: the four 64-cout patch configurations
        if (ceil_div(a.W, 20) * ceil_div(a.H, 6) < ceil_div(a.W, 16) * ceil_div(a.H, 8)) { add(13, 2); add(13, 1); add(13, 3); }
        add(11, 2); add(11, 1); add(11, 3); add(13, 2); add(13, 1); add(13, 3);
        return n;
    }
    if (!aligned) {
        add(tile, 0);
        if (a.Cout > 32) { add(3, 0); add(6, 0); }
        if (a.Cout > 64) add(5, 0);
        return n;
    }
    // heuristic first (what an un-tuned plan runs: the r01 measurements' usual winners), then the exhaustive
    // (tile x variant) sweep the autotuner times.  Every candidate of a layer sums in the layer's K walk (conv_cfg_valid).
    // 6x20 tiles where they need fewer tiles than 8x16 ones (20 x 20 maps: 4 instead of 6 per image)
    const bool t20 = ceil_div(a.W, 20) * ceil_div(a.H, 6) < ceil_div(a.W, 16) * ceil_div(a.H, 8);
    const bool patch_ok = !getenv("BSY_NO_PATCH");
    if (!a.epi && a.ksize == 3 && a.stride == 1 && a.H >= 16 && a.W >= 16 && patch_ok) add(t20 ? 13 : 11, 2);  // patch kernel
    const int vbest = (aligned64 && ko != 1) ? 3 : 1;  // 64-deep K-steps where the K walk allows them
    if (!a.epi && !(a.Cout & 255) && aligned64 && ko != 1 && M >= 16384) add(7, 3 | km);  // 256 x 256, 64-deep K-steps
    add(tile, vbest | km);
    static const int tn[10] = {32, 64, 128, 64, 128, 128, 64, 256, 128, 64};
    for (int t = 0; t < 10; ++t) {
        if (tn[t] >= 2 * round_up(a.Cout, 32)) continue;      // more than half of the cout tile would be padding
        if (tn[t] == 32 && a.Cout > 32) continue;               // 32-wide tiles re-read the pixels once per 32 couts
        if (t == 4 && (a.Cout < 128 || M < 32768)) continue;    // 8-wave 256x128 only for wide, large layers
        if (t == 7 && M < 16384) continue;
        if (t >= 8 && (M < 65536 || getenv("BSY_NO_PERSIST"))) continue;                      // persistent tiles need several tiles per workgroup
        for (int v = 1; v <= 3; ++v) add(t, v | (t < 8 ? km : 0));
    }
    if (a.ksize == 1 && M >= 65536 && !getenv("BSY_NO_WRES")) { add(14, 1); add(14, 2); add(15, 1); add(15, 2); }  // HBM-bound 1x1 layers
    if (a.H >= 16 && a.W >= 16 && patch_ok) {  // 3x3 s1 patch kernel
        add(10, 1); add(11, 1); add(11, 2); add(11, 3);
        if (t20) { add(12, 1); add(13, 1); add(13, 2); add(13, 3); }
    }
    return n;
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
    if (a.epi) {
        if ((a.epi != 2 && a.epi != 3) || !a.y || a.A <= 0 || a.a0 < 0 || a.a0 + a.OH * a.OW > a.A || a.dst_scale > 1 || a.res)
            BSY_FAIL(BSY_ERR_ARG, "conv: bad fused-decoder arguments (epi %d)", a.epi);
        if (a.epi == 3 && (a.Cout != 64 || a.nrows < 4)) BSY_FAIL(BSY_ERR_ARG, "conv: DFL epilogue needs 64 box channels");
        if (a.tail_wgt && (a.epi != 3 || !a.tail_bias || a.ksize != 3 || ((uintptr_t)a.tail_wgt & 15) || ((uintptr_t)a.tail_bias & 15)))
            BSY_FAIL(BSY_ERR_ARG, "conv: bad box-branch tail arguments");
        if (a.epi == 2 && a.nrows < 4 + a.Cout) BSY_FAIL(BSY_ERR_ARG, "conv: y has %d rows, class epilogue needs %d", a.nrows, 4 + a.Cout);
        if (a.raw && a.rawC < (a.epi == 3 ? 64 : 64 + a.Cout)) BSY_FAIL(BSY_ERR_ARG, "conv: raw map has too few channels");
    }
    ConvK k;
    k.src0 = a.src0; k.src1 = a.src1; k.ld0 = a.ld0; k.ld1 = a.ld1; k.C0 = a.C0; k.C1 = a.C1;
    k.up0 = a.up0; k.up1 = a.up1; k.H = a.H; k.W = a.W; k.OH = a.OH; k.OW = a.OW; k.stride = a.stride; k.pad = a.pad;
    k.M = (int)M; k.Cin8 = Cin / 8; k.ntaps = a.ksize * a.ksize;
    k.Kpad = round_up(a.ksize * a.ksize * Cin, 32); k.nk = k.Kpad / 32;
    k.wgt = a.wgt; k.bias = a.bias; k.dst = a.dst; k.ldd = a.ldd; k.Cout = a.Cout; k.out_f32 = a.out_f32;
    k.res = a.res; k.ldr = a.ldr; k.act = a.act; k.B = a.B; k.tiles_x = k.tiles_y = 0;
    k.epi = a.epi; k.y = a.y; k.y_f32 = a.y_f32; k.A = a.A; k.a0 = a.a0; k.nrows = a.nrows; k.lvl_stride = a.lvl_stride;
    k.raw = a.raw; k.raw_f32 = a.raw_f32; k.rawC = a.rawC;
    k.tail_wgt = a.tail_wgt; k.tail_bias = a.tail_bias;
    if (a.tail_wgt && !a.epi) BSY_FAIL(BSY_ERR_ARG, "conv: a box-branch tail needs the decoder arguments (epi 3)");
    k.dst_scale = a.dst_scale > 0 ? a.dst_scale : 1; k.dst_dy = a.dst_dy; k.dst_dx = a.dst_dx; k.ntn = 1;
    k.nsl_c = k.nsl_t = 1; k.cb_per = Cin; k.taps_per = a.ksize * a.ksize; k.split_ws = nullptr; k.slab = 0; k.ldw = 0;
    if (a.nsl_c * a.nsl_t > 1) {
        const int nt = a.ksize * a.ksize;
        if (a.nsl_c < 1 || a.nsl_t < 1 || !a.split_ws || a.epi || a.out_f32 || a.tail_wgt || (a.dst_scale > 1) || (Cin % a.nsl_c) || ((Cin / a.nsl_c) & 31) || (a.C0 & 31) ||
            (nt % a.nsl_t) || (a.Cout & 7) || (a.ldd & 7) || ((uintptr_t)a.dst & 15) || (a.res && ((a.ldr & 7) || ((uintptr_t)a.res & 15))) || ((uintptr_t)a.split_ws & 15))
            BSY_FAIL(BSY_ERR_ARG, "conv: bad split-K arguments (%d x %d slices)", a.nsl_c, a.nsl_t);
        if (conv_korder(a) == 2 && ((Cin / a.nsl_c) & 63)) BSY_FAIL(BSY_ERR_ARG, "conv: split-K slices of a 64-channel-chunk layer must be multiples of 64 channels");
        k.nsl_c = a.nsl_c; k.nsl_t = a.nsl_t; k.cb_per = Cin / a.nsl_c; k.taps_per = nt / a.nsl_t; k.split_ws = a.split_ws;
        k.ldw = round_up(a.Cout, 32);               // whole 32-cout MFMA tiles: the slab rows take every accumulator group unconditionally
        k.slab = (long long)M * k.ldw;
    }
    // ---- configuration: explicit (autotuned, ConvArgs::cfg) or heuristic -----------------------------------------
    if (a.ksize == 3 && (a.up0 || a.up1)) BSY_FAIL(BSY_ERR_ARG, "conv: upsampled source only with ksize 1");
    // element offsets are kept in 32 bits inside the kernel
    if ((long long)a.B * a.H * a.W * (long long)(a.ld0 > a.ld1 ? a.ld0 : a.ld1) >= (1LL << 31))
        BSY_FAIL(BSY_ERR_ARG, "conv: source view exceeds 2^31 elements (split the batch)");
    static const int dbg = [] { const char* e = getenv("BSY_CONV_DBG"); return e ? atoi(e) : 0; }();
    k.dbg = dbg;
    {   // bytes addressable from each view's first element up to its last element (descriptor range check)
        const long long px0 = (long long)a.B * (a.H >> a.up0) * (a.W >> a.up0), px1 = (long long)a.B * (a.H >> a.up1) * (a.W >> a.up1);
        k.span0 = (unsigned)(((px0 - 1) * a.ld0 + a.C0) * 2);
        k.span1 = a.C1 ? (unsigned)(((px1 - 1) * a.ld1 + a.C1) * 2) : 0u;
        k.wspan = (unsigned)((long long)round_up(a.Cout, 128) * k.Kpad * 2);
    }
    int cfg = a.cfg;
    if (cfg < 0 || !conv_cfg_valid(a, cfg)) {
        int list[BSY_CONV_MAX_CFG];
        if (conv_candidates(a, list, BSY_CONV_MAX_CFG) <= 0) BSY_FAIL(BSY_ERR_ARG, "conv: no kernel configuration for this shape");
        cfg = list[0];
    }
    const int tile = cfg >> 4, var = cfg & 3;
    k.korder = conv_korder(a);
    // tile: 0 = 256x32, 1 = 256x64, 2 = 128x128, 3 = 128x64, 4 = 256x128 (8 waves), 5 = 64x128, 6 = 64x64
    // var : 0 = generic BK32 S3, 1 = aligned BK32 S3, 2 = aligned BK32 S2, 3 = aligned BK64 S2
#define BSY_VAR(KS_, WM_, WN_, MT_, NT_)                                                  \
    do {                                                                                  \
        if (var == 0) return launch_cfg<KS_, WM_, WN_, MT_, NT_, 3, false, 32>(k, s);     \
        if (var == 1) return launch_cfg<KS_, WM_, WN_, MT_, NT_, 3, true, 32>(k, s);      \
        if (var == 2) return launch_cfg<KS_, WM_, WN_, MT_, NT_, 2, true, 32>(k, s);      \
        return launch_cfg<KS_, WM_, WN_, MT_, NT_, 2, true, 64>(k, s);                    \
    } while (0)
#define BSY_TILE(KS_)                                                                     \
    do {                                                                                  \
        if (k.tail_wgt && tile == 13 && var == 1) return launch_patch<1, 3, 2, 20, true>(k, s); \
        if (k.tail_wgt && tile == 13 && var == 3) return launch_patch<1, 4, 2, 20, true>(k, s); \
        if (k.tail_wgt && tile == 13) return launch_patch<1, 2, 2, 20, true>(k, s);       \
        if (k.tail_wgt && tile == 11 && var == 1) return launch_patch<1, 3, 2, 16, true>(k, s); \
        if (k.tail_wgt && tile == 11 && var == 3) return launch_patch<1, 4, 2, 16, true>(k, s); \
        if (k.tail_wgt) return launch_patch<1, 2, 2, 16, true>(k, s);                     \
        if (tile >= 14) {                                                                 \
            const int fr = (tile == 14 ? 2 : 1) * (Cin / 16);                             \
            if (tile == 14 && var == 1) { if (fr <= 16) return launch_wres<2, 4, 16>(k, s); return launch_wres<2, 4, 24>(k, s); } \
            if (tile == 14) { if (fr <= 16) return launch_wres<2, 3, 16>(k, s); return launch_wres<2, 3, 24>(k, s); } \
            if (var == 1) { if (fr <= 16) return launch_wres<1, 4, 16>(k, s); return launch_wres<1, 4, 32>(k, s); } \
            if (fr <= 16) return launch_wres<1, 3, 16>(k, s);                             \
            return launch_wres<1, 3, 32>(k, s);                                           \
        }                                                                                 \
        if (tile == 12) return launch_patch<2, 2, 2, 20>(k, s);                           \
        if (tile == 13 && var == 1) return launch_patch<1, 3, 2, 20>(k, s);               \
        if (tile == 13 && var == 3) return launch_patch<1, 4, 2, 20>(k, s);               \
        if (tile == 13) return launch_patch<1, 2, 2, 20>(k, s);                           \
        if (tile == 10) return launch_patch<2, 2, 2>(k, s);                               \
        if (tile == 11 && var == 1) return launch_patch<1, 3, 2>(k, s);                   \
        if (tile == 11 && var == 3) return launch_patch<1, 4, 2>(k, s);                   \
        if (tile == 11) return launch_patch<1, 2, 2>(k, s);                               \
        if (tile == 8 && var == 1) return launch_persist<2, 2, 2, 2, 3, 32>(k, s);        \
        if (tile == 8 && var == 2) return launch_persist<2, 2, 2, 2, 2, 32>(k, s);        \
        if (tile == 8) return launch_persist<2, 2, 2, 2, 2, 64>(k, s);                    \
        if (tile == 9 && var == 1) return launch_persist<2, 2, 2, 1, 3, 32>(k, s);        \
        if (tile == 9 && var == 2) return launch_persist<2, 2, 2, 1, 2, 32>(k, s);        \
        if (tile == 9) return launch_persist<2, 2, 2, 1, 2, 64>(k, s);                    \
        if (tile == 0) BSY_VAR(KS_, 4, 1, 2, 1);                                          \
        if (tile == 1) BSY_VAR(KS_, 4, 1, 2, 2);                                          \
        if (tile == 2) BSY_VAR(KS_, 2, 2, 2, 2);                                          \
        if (tile == 3) BSY_VAR(KS_, 2, 2, 2, 1);                                          \
        if (tile == 5) BSY_VAR(KS_, 1, 4, 2, 1);                                          \
        if (tile == 6) BSY_VAR(KS_, 2, 2, 1, 1);                                          \
        if (tile == 7 && var == 3) return launch_cfg<KS_, 4, 2, 2, 4, 2, true, 64>(k, s); \
        if (tile == 7 && var == 1) return launch_cfg<KS_, 4, 2, 2, 4, 3, true, 32>(k, s); \
        if (tile == 7) return launch_cfg<KS_, 4, 2, 2, 4, 2, true, 32>(k, s);             \
        if (var == 1) return launch_cfg<KS_, 4, 2, 2, 2, 3, true, 32>(k, s);              \
        if (var == 3) return launch_cfg<KS_, 4, 2, 2, 2, 2, true, 64>(k, s);              \
        return launch_cfg<KS_, 4, 2, 2, 2, 2, true, 32>(k, s);                            \
    } while (0)
    if (a.ksize == 1) BSY_TILE(1);
    BSY_TILE(3);
#undef BSY_TILE
#undef BSY_VAR
}
