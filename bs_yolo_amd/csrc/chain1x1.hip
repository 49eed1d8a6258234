// Two chained 1x1 convolutions as ONE launch (round 4): out = act2(W2 * [h | keep(act1(W1 * x + b1) (+ r1))] + b2) (+ r2), per pixel.
//
// Replaces, bit for bit, the pairs of Conv.forward_fuse launches (nn/modules/conv.py:149-151) that the reference's block structure
// chains per pixel at 40 x 40 / 20 x 20:
//   * C3k2.cv1 -> C3k.cv1 | C3k.cv2   (block.py:3796-3804, :3320-3334): the chunk y1 of cv1's output feeds the inner block's two 1x1
//     convs; cv1's output still goes to HBM (the block's cv2 reads it), the y1 half ALSO stays in LDS and is the second conv's input;
//   * C2PSA.cv1 -> Attention.qkv      (block.py:4429-4468, :4274): the same shape of chain (b = second half of cv1's output);
//   * C3k.cv3 -> C3k2.cv2             (cat(y0, y1, C3k(y1)) -> cv2): cv3's output exists only in LDS (nobody else reads it); the second
//     conv's K walk takes [y0 | y1] from HBM and then the resident tile -- the concat order, i.e. the un-fused launch's K order;
//   * PSABlock.ffn[1] (+ shortcut) -> C2PSA.cv2 (block.py:4382, :4467): likewise, with the shortcut operand added in the first epilogue.
//
// One workgroup = 128 pixels, 8 waves (4 along cout x 2 along pixels, wave tile 64 couts x 64 pixels), one workgroup per CU:
//   stage 1: for every 256- (or 128-) cout pass: K loop over the HBM sources (pixels + weights staged by LDS-DMA, 64-deep K-steps:
//            conv_mfma.hip's ALIGNED BK-64 form) -> SiLU -> f16 -> optional HBM store + the kept channel range parked in the RESIDENT
//            tile, laid out as K-step pixel blocks [block][128 px][64 ch] with the ring's XOR swizzle, so that stage 2 reads its B
//            fragments from it exactly as from a ring stage;
//   stage 2: per cout pass: K loop = HBM part (ring) then resident blocks (weights only through the ring) -> epilogue -> HBM.
// ONE continuous stream of K-steps runs through the two ring stages across pass boundaries (global step g -> stage g & 1): the next
// pass's first K-step is fetched under this pass's last K-step and its epilogue.  A pass's output leaves through a staging tile that
// aliases only the ring stage the pass consumed last, 64 pixel rows at a time, as 16-byte descriptor stores that are ALWAYS issued (rows
// past M: out of range = dropped), so that every thread has a compile-time number of stores in flight and the next pass's first wait can
// be a counted vmcnt that lets them fly on.  Which 8-row group of a K-step's weights each DMA instruction fetches is rotated per
// workgroup: this kernel's workgroups march in lockstep over the same weight lines, and asking for them in different orders is worth
// 4-9 % (docs/experiments.md section 0.5).
// Every accumulator starts at the bias and takes its products in ascending k, 16 at a time, through v_mfma_f32_32x32x16_f16 -- the K walk
// (order 0) of the un-fused 1x1 launches; activations are rounded to f16 where the un-fused pair stores them, the shortcut is added to
// the ROUNDED value in f32 and rounded again (conv_mfma.hip conv_epilogue_lds): the same bits (tests/test_gpu_parity.py
// test_engine_chain_fusion_is_bit_identical).
// LDS: ring 2 x (128 + 256) x 64 halves = 96 KiB + resident tile <= 256 ch x 128 px = 64 KiB = all 160 KiB.
// MEASURED (YOLO11s, 64 images): 10-25 % SLOWER than the pairs it replaces -- the kernel is its operand stream (36-40 GB/s per CU; the
// MFMAs hide under it completely) and a 128-pixel tile stages 11.7 KB per MFLOP against 7.8 for the pairs' 256 x 256 tiles.  Opt-in
// (bs_yolo_amd/plan.py fuse_chain / BSY_FUSE_CHAIN=1); BSY_CHAIN_DBG holds the ablation switches behind those numbers.
#include <stdlib.h>

#include "common.h"

struct ChainK {
    const half_t *a0, *a1;      // stage-1 sources (a1 may be null)
    int lda0, lda1, CA0, CA1;
    unsigned spa0, spa1;
    const half_t* w1;
    const float* b1;
    int K1pad, N1, act1;
    unsigned wsp1;
    half_t* d1;                 // stage-1 output in HBM (null: resident tile only)
    int ldd1;
    unsigned spd1, spd2;        // bytes addressable from d1 / d2 (stores go through descriptors)
    const half_t* r1;           // stage-1 shortcut operand (null: none)
    int ldr1;
    int keep0, LC;              // resident tile = stage-1 couts [keep0, keep0 + LC)
    const half_t* h2;           // stage-2 HBM K part (null: none); its channels come FIRST in stage 2's K order
    int ldh2, CH2;
    unsigned sph2;
    const half_t* w2;
    const float* b2;
    int K2pad, N2, act2;
    unsigned wsp2;
    half_t* d2;
    int ldd2;
    const half_t* r2;
    int ldr2;
    int M;
    int dbg;  // ablation switches for profiling (BSY_CHAIN_DBG; results are WRONG under them): 1 = no DMA, 2 = no fragment reads / MFMAs, 4 = no HBM stores, 8 = no pixel DMA, 16 = no weight DMA, 32 = weight addresses of a K-step-major layout [kt][cout][64] (timing only), 64 = weights global -> registers -> LDS instead of LDS-DMA, 128 = NO per-workgroup rotation of the weight row-group order (results stay right), 256 = the next K-step's DMA issued between this step's MFMA groups (results stay right)
};

namespace {
#if defined(__HIP_DEVICE_COMPILE__)
typedef __amdgpu_buffer_rsrc_t ch_rsrc_t;
typedef unsigned int ch_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ ch_rsrc_t ch_make_rsrc(const void* base, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), (short)0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ void ch_dma16(ch_rsrc_t r, unsigned voff, unsigned soff, half_t* lds_wave_base) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, lds_wave_base, 16, (int)voff, (int)soff, 0, 0);
}
// 16-byte store through a descriptor: ALWAYS issued (a lane whose voff is out of range stores nothing), so that every thread has the
// same, compile-time number of stores in flight behind an epilogue -- the next pass's counted vmcnt wait relies on it
__device__ __forceinline__ void ch_store16(ch_rsrc_t r, unsigned voff, const half8& v) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(ch_u32x4, v), r, (int)voff, 0, 0);
}
__device__ __forceinline__ half8 ch_load16(ch_rsrc_t r, unsigned voff, unsigned soff) {
    return __builtin_bit_cast(half8, __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, (int)soff, 0));
}
#else
typedef int ch_rsrc_t;
__device__ __forceinline__ half8 ch_load16(ch_rsrc_t, unsigned, unsigned) { return half8{0, 0, 0, 0, 0, 0, 0, 0}; }
__device__ __forceinline__ ch_rsrc_t ch_make_rsrc(const void*, unsigned) { return 0; }
__device__ __forceinline__ void ch_dma16(ch_rsrc_t, unsigned, unsigned, half_t*) {}
__device__ __forceinline__ void ch_store16(ch_rsrc_t, unsigned, const half8&) {}
#endif
constexpr unsigned CH_OOB = 0xFFFFFFF0u;  // out of range for every descriptor: a load lands as zeros, a store is dropped
constexpr int CH_P = 128, CH_BK = 64;
constexpr int CH_PBLK = CH_P * CH_BK;            // one K-step of pixels: [128][64] halves, 16 KiB
constexpr int CH_STAGE = (CH_P + 256) * CH_BK;   // ring stage: pixels + up to 256 weight rows
constexpr int CH_RING = 2 * CH_STAGE;
constexpr int CH_TILE = 4 * CH_PBLK;             // resident tile: up to 256 channels

// What a thread needs to stage any K-step of any pass: ONE continuous stream of K-steps runs through the two ring stages, across pass
// boundaries (global step g lands in stage g & 1), so that the first K-step of a pass is fetched under the last K-step and the
// epilogue of the pass before it.
struct ChCtx {
    ch_rsrc_t rsa0, rsa1, rsh2, rsw1, rsw2;
    unsigned offa0[2], offa1[2], offh2[2];  // byte offset of this lane's two staged pixel rows (+ its swizzled 16-byte chunk) or CH_OOB
    unsigned woff1[4], woff2[4];            // byte offset of this lane's staged weight rows (+ chunk) inside a pass's rows
    int wq1[4], wq2[4];                     // which 8-row group of the pass each of this wave's weight DMA instructions fetches
    int na0, na01, nh2, nl;                 // K-steps: stage 1 = a0 then a1; stage 2 = h2 then nl resident blocks
    int np1, np2, w1, w2;                   // passes per stage and their width in 128-cout units (1 or 2)
    unsigned k1pad, k2pad;
};

// LDS-DMA of K-step kt of pass pi into ring stage st
// part < 0: the whole K-step; part = 0..3: the slice issued behind MFMA sub-step `part` (interleaved form: one pixel piece and one weight piece)
__device__ __forceinline__ void chain_issue(half_t* ring, const ChCtx& c, const int pi, const int kt, const int st, const int wave, const int dbg, const int part = -1) {
    half_t* sP = ring + st * CH_STAGE;
    half_t* sW = sP + CH_PBLK;
    if (pi < c.np1) {
        const unsigned cout0 = 128u * (unsigned)(pi * c.w1);
        if (dbg & 8) {
        } else if (kt < c.na0) {
#pragma unroll
            for (int i = 0; i < 2; ++i) if (part < 0 || part == i) ch_dma16(c.rsa0, c.offa0[i], 128u * (unsigned)kt, sP + (wave * 2 + i) * 512);
        } else {
#pragma unroll
            for (int i = 0; i < 2; ++i) if (part < 0 || part == i) ch_dma16(c.rsa1, c.offa1[i], 128u * (unsigned)(kt - c.na0), sP + (wave * 2 + i) * 512);
        }
        unsigned so = 2u * (cout0 * c.k1pad + 64u * (unsigned)kt);
        if (dbg & 32) so = 128u * ((unsigned)kt * 128u * (unsigned)(c.np1 * c.w1) + cout0);  // K-step-major block [kt][cout][64] (timing experiment: wrong data)
        if (dbg & 16) {
        } else if (c.w1 == 2) {
#pragma unroll
            for (int j = 0; j < 4; ++j) if (part < 0 || part == j) ch_dma16(c.rsw1, c.woff1[j], so, sW + c.wq1[j] * 512);
        } else {
#pragma unroll
            for (int j = 0; j < 2; ++j) if (part < 0 || part == j) ch_dma16(c.rsw1, c.woff1[j], so, sW + c.wq1[j] * 512);
        }
    } else {
        const unsigned cout0 = 128u * (unsigned)((pi - c.np1) * c.w2);
        if (kt < c.nh2 && !(dbg & 8)) {
#pragma unroll
            for (int i = 0; i < 2; ++i) if (part < 0 || part == i) ch_dma16(c.rsh2, c.offh2[i], 128u * (unsigned)kt, sP + (wave * 2 + i) * 512);
        }
        unsigned so = 2u * (cout0 * c.k2pad + 64u * (unsigned)kt);
        if (dbg & 32) so = 128u * ((unsigned)kt * 128u * (unsigned)(c.np2 * c.w2) + cout0);
        if (dbg & 16) {
        } else if (c.w2 == 2) {
#pragma unroll
            for (int j = 0; j < 4; ++j) if (part < 0 || part == j) ch_dma16(c.rsw2, c.woff2[j], so, sW + c.wq2[j] * 512);
        } else {
#pragma unroll
            for (int j = 0; j < 2; ++j) if (part < 0 || part == j) ch_dma16(c.rsw2, c.woff2[j], so, sW + c.wq2[j] * 512);
        }
    }
}

// One pass: K loop (acc += W[cout0 + (wn NT + a) 32 ..][k] * B[k][wm 64 + b 32 ..]) + epilogue.
//   g    : global K-step counter (ring stage = g & 1); the DMA of this pass's first K-step is already in flight when the pass starts
//   pend : 16-byte stores the previous epilogue left in flight per thread (issued AFTER that DMA: the first wait lets them fly on)
// Epilogue: activation -> f16 (-> + shortcut operand, rounded again); STAGE1: the kept couts go to the resident tile; when `dst` is set
// the pass's tile goes to HBM through a staging tile that aliases ONLY the ring stage this pass consumed last (the other stage is
// receiving the next pass's first K-step), 64 pixel rows at a time, as coalesced 16-byte pieces.
template <int NT, bool STAGE1>
__device__ __forceinline__ void chain_pass(half_t* smem, const ChCtx& c, const ChainK& p, const int pi, const int cout0, int& g, int& pend,
                                           const int m0, const int wave, const int lane, const int tid) {
    half_t* ring = smem;
    half_t* tile = smem + CH_RING;
    const int wn = wave >> 1, wm = wave & 1, lrow = lane & 31, lh = lane >> 5;
    const int n01 = STAGE1 ? c.na01 : c.nh2, nk = n01 + (STAGE1 ? 0 : c.nl);
    const int npass = c.np1 + c.np2;
    f32x16 acc[NT][2];
    const float* bias = (STAGE1 ? p.b1 : p.b2) + cout0;
#pragma unroll
    for (int a = 0; a < NT; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc_bias(acc[a][b], bias + (wn * NT + a) * 32, lh);
    for (int kt = 0; kt < nk; ++kt, ++g) {
        if (kt == 0) {  // this K-step's DMA is older than the previous epilogue's stores; its LDS writes (resident tile) must have landed too
            if (pend == 8) asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
            else if (pend == 4) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();  // K-step g landed for every wave; every wave is done with the other stage
        half8 wreg[4];
        const bool wr = (p.dbg & 64) && kt + 1 < nk;  // timing experiment: the next K-step's weights global -> registers -> LDS instead of LDS-DMA
        const bool inter = (p.dbg & 256) != 0;  // the next K-step's DMA instructions issued BETWEEN this step's MFMA groups instead of in front of them
        const int npi = kt + 1 < nk ? pi : pi + 1, nkt = kt + 1 < nk ? kt + 1 : 0;
        const bool has_next = kt + 1 < nk || pi + 1 < npass;
        if (!(p.dbg & 1) && !inter) {
            if (kt + 1 < nk) chain_issue(ring, c, pi, kt + 1, (g + 1) & 1, wave, wr ? (p.dbg | 16) : p.dbg);
            else if (pi + 1 < npass) chain_issue(ring, c, pi + 1, 0, (g + 1) & 1, wave, p.dbg);
            if (wr) {
                const unsigned so = 2u * ((unsigned)cout0 * (STAGE1 ? c.k1pad : c.k2pad) + 64u * (unsigned)(kt + 1));
#pragma unroll
                for (int j = 0; j < 2 * NT; ++j) wreg[j] = ch_load16(STAGE1 ? c.rsw1 : c.rsw2, STAGE1 ? c.woff1[j] : c.woff2[j], so);
            }
        }
        if (p.dbg & 2) {
            if (wr) {
#pragma unroll
                for (int j = 0; j < 2 * NT; ++j) *reinterpret_cast<half8*>(ring + ((g + 1) & 1) * CH_STAGE + CH_PBLK + (STAGE1 ? c.wq1[j] : c.wq2[j]) * 512 + lane * 8) = wreg[j];
            }
            continue;
        }
        const half_t* sP = kt < n01 ? ring + (g & 1) * CH_STAGE : tile + (kt - n01) * CH_PBLK;
        const half_t* sW = ring + (g & 1) * CH_STAGE + CH_PBLK;
        half8 bfr[2][2], afr[2][NT];
        auto rd = [&](const int ks, const int buf) {
            const int chunk = 2 * ks + lh;
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                const int row = wm * 64 + b * 32 + lrow;
                bfr[buf][b] = *reinterpret_cast<const half8*>(sP + row * CH_BK + ((chunk ^ ((row >> 1) & 7)) << 3));
            }
#pragma unroll
            for (int a = 0; a < NT; ++a) {
                const int row = (wn * NT + a) * 32 + lrow;
                afr[buf][a] = *reinterpret_cast<const half8*>(sW + row * CH_BK + ((chunk ^ ((row >> 1) & 7)) << 3));
            }
        };
        rd(0, 0);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            if (ks + 1 < 4) rd(ks + 1, (ks + 1) & 1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int a = 0; a < NT; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(afr[ks & 1][a], bfr[ks & 1][b], acc[a][b], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (inter && has_next && !(p.dbg & 1)) {
                chain_issue(ring, c, npi, nkt, (g + 1) & 1, wave, p.dbg, ks);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (wr) {
#pragma unroll
            for (int j = 0; j < 2 * NT; ++j) *reinterpret_cast<half8*>(ring + ((g + 1) & 1) * CH_STAGE + CH_PBLK + (STAGE1 ? c.wq1[j] : c.wq2[j]) * 512 + lane * 8) = wreg[j];
        }
    }
    // ---- epilogue ----
    constexpr int NP = 128 * NT, LDT = NP + 8;
    const int act = STAGE1 ? p.act1 : p.act2;
    half_t* dst = STAGE1 ? p.d1 : p.d2;
    const int ldd = STAGE1 ? p.ldd1 : p.ldd2;
    half4 o[2][NT][4];
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        const int prow = wm * 64 + b * 32 + lrow;
#pragma unroll
        for (int a = 0; a < NT; ++a) {
            const int cl = (wn * NT + a) * 32;
            const int ck0 = cout0 + cl - p.keep0;
            const bool keep = STAGE1 && ck0 >= 0 && ck0 < p.LC;  // wave-uniform (keep0, LC multiples of 64)
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                f32x4 t = f32x4{acc[a][b][4 * gq], acc[a][b][4 * gq + 1], acc[a][b][4 * gq + 2], acc[a][b][4 * gq + 3]};
                if (act) t = silu4_f(t);
                half4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = (half_t)t[e];
                if (STAGE1 && p.r1 && m0 + prow < p.M) {  // stage-1 shortcut: in registers (the resident tile must hold the sum)
                    const half4 r = *reinterpret_cast<const half4*>(p.r1 + (size_t)(m0 + prow) * p.ldr1 + cout0 + cl + 8 * gq + 4 * lh);
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = (half_t)((float)v[e] + (float)r[e]);
                }
                o[b][a][gq] = v;
                if (keep) {  // nobody reads the resident tile during stage 1
                    const int ck = ck0 + 8 * gq + 4 * lh, cc = ck & 63;
                    *reinterpret_cast<half4*>(tile + (ck >> 6) * CH_PBLK + prow * CH_BK + (((cc >> 3) ^ ((prow >> 1) & 7)) << 3) + (cc & 7)) = v;
                }
            }
        }
    }
    pend = 0;
    if (!dst || (p.dbg & 4)) return;
    half_t* stage = ring + ((g - 1) & 1) * CH_STAGE;  // the stage of this pass's last K-step; 64 x LDT halves <= CH_STAGE
    const ch_rsrc_t rsd = ch_make_rsrc(dst, STAGE1 ? p.spd1 : p.spd2);
    const half_t* res = STAGE1 ? nullptr : p.r2;
    constexpr int CPRW = NP / 8, ITERH = 64 * CPRW / 512;  // 16-byte pieces per row; pieces per thread and half (4 or 2)
    __builtin_amdgcn_s_barrier();  // every wave has left the K loop (its fragment reads were consumed by its MFMAs)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        if (wm == h) {
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int a = 0; a < NT; ++a)
#pragma unroll
                    for (int gq = 0; gq < 4; ++gq)
                        *reinterpret_cast<half4*>(stage + (b * 32 + lrow) * LDT + (wn * NT + a) * 32 + 8 * gq + 4 * lh) = o[b][a][gq];
        }
        lds_barrier();
#pragma unroll
        for (int i = 0; i < ITERH; ++i) {
            const int id = tid + 512 * i;
            const int row = id / CPRW, cc = (id % CPRW) * 8;
            const int m = m0 + h * 64 + row;
            half8 v = *reinterpret_cast<const half8*>(stage + row * LDT + cc);
            if (res && m < p.M) {
                const half8 r = *reinterpret_cast<const half8*>(res + (size_t)m * p.ldr2 + cout0 + cc);
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = (half_t)((float)v[e] + (float)r[e]);
            }
            ch_store16(rsd, m < p.M ? 2u * ((unsigned)m * (unsigned)ldd + (unsigned)(cout0 + cc)) : CH_OOB, v);
        }
        lds_barrier();  // the staging rows have been read (lgkmcnt(0)): the other half, or the next DMA into this stage, may overwrite them
    }
    pend = 2 * ITERH;
}
}  // namespace

__global__ __launch_bounds__(512) void chain1x1_kernel(const ChainK p) {
    __shared__ __attribute__((aligned(16))) half_t smem[CH_RING + CH_TILE];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m0 = blockIdx.x * CH_P;
    const int rsub = lane >> 3, slot = lane & 7;
    const int kc[2] = {slot ^ (rsub >> 1), slot ^ (4 | (rsub >> 1))};
    ChCtx c;
    c.rsa0 = ch_make_rsrc(p.a0, p.spa0);
    c.rsa1 = ch_make_rsrc(p.a1 ? p.a1 : p.a0, p.a1 ? p.spa1 : 0u);
    c.rsh2 = ch_make_rsrc(p.h2 ? p.h2 : p.a0, p.h2 ? p.sph2 : 0u);
    c.rsw1 = ch_make_rsrc(p.w1, p.wsp1);
    c.rsw2 = ch_make_rsrc(p.w2, p.wsp2);
    c.na0 = p.CA0 >> 6; c.na01 = c.na0 + (p.a1 ? p.CA1 >> 6 : 0); c.nh2 = p.h2 ? p.CH2 >> 6 : 0; c.nl = p.LC >> 6;
    c.w1 = (p.N1 & 255) ? 1 : 2; c.w2 = (p.N2 & 255) ? 1 : 2;
    c.np1 = p.N1 / (128 * c.w1); c.np2 = p.N2 / (128 * c.w2);
    c.k1pad = (unsigned)p.K1pad; c.k2pad = (unsigned)p.K2pad;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int m = m0 + (wave * 2 + i) * 8 + rsub;
        const bool v = m < p.M;
        c.offa0[i] = v ? 2u * (unsigned)m * (unsigned)p.lda0 + 16u * (unsigned)kc[i] : CH_OOB;
        c.offa1[i] = v ? 2u * (unsigned)m * (unsigned)p.lda1 + 16u * (unsigned)kc[i] : CH_OOB;
        c.offh2[i] = v ? 2u * (unsigned)m * (unsigned)p.ldh2 + 16u * (unsigned)kc[i] : CH_OOB;
    }
    // 8-row group q of a pass (16 x width groups) is fetched by instruction j of wave (q - rot) / WIW: rot rotates the assignment per
    // workgroup (dbg & 128 switches it off), so that the CUs of an XCD -- which all want the same weight lines in the same K-step -- ask for them in
    // different orders.  q & 1 = parity of the group (the swizzle's odd / even instruction form).
    const int rot = (p.dbg & 128) ? 0 : (int)((blockIdx.x >> 3) * 5u);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int q1 = (wave * 2 * c.w1 + j + rot) % (16 * c.w1), q2 = (wave * 2 * c.w2 + j + rot) % (16 * c.w2);
        c.wq1[j] = q1; c.wq2[j] = q2;
        c.woff1[j] = (unsigned)(q1 * 8 + rsub) * ((p.dbg & 32) ? 64u : c.k1pad) * 2u + 16u * (unsigned)kc[q1 & 1];
        c.woff2[j] = (unsigned)(q2 * 8 + rsub) * ((p.dbg & 32) ? 64u : c.k2pad) * 2u + 16u * (unsigned)kc[q2 & 1];
    }
    int g = 0, pend = 0;
    if (!(p.dbg & 1)) chain_issue(smem, c, 0, 0, 0, wave, p.dbg);
    for (int pi = 0; pi < c.np1; ++pi) {
        if (c.w1 == 2) chain_pass<2, true>(smem, c, p, pi, 256 * pi, g, pend, m0, wave, lane, tid);
        else chain_pass<1, true>(smem, c, p, pi, 128 * pi, g, pend, m0, wave, lane, tid);
    }
    for (int pi = 0; pi < c.np2; ++pi) {
        if (c.w2 == 2) chain_pass<2, false>(smem, c, p, c.np1 + pi, 256 * pi, g, pend, m0, wave, lane, tid);
        else chain_pass<1, false>(smem, c, p, c.np1 + pi, 128 * pi, g, pend, m0, wave, lane, tid);
    }
}

// Shapes the kernel takes (mirror: bs_yolo_amd/plan.py chain_supported).  CA0 / CA1 / CH2 / LC: whole 64-deep K-steps; couts in whole
// 128-cout passes; the resident tile holds at most 256 channels and must lie inside stage 1's couts.
bool chain_supported(int CA0, int CA1, int N1, int keep0, int LC, int CH2, int N2) {
    return CA0 > 0 && !(CA0 & 63) && CA1 >= 0 && !(CA1 & 63) && N1 > 0 && !(N1 & 127) && N2 > 0 && !(N2 & 127) && LC > 0 && LC <= 256 && !(LC & 63) &&
           keep0 >= 0 && !(keep0 & 63) && keep0 + LC <= N1 && CH2 >= 0 && !(CH2 & 63);
}

int launch_chain(const ChainArgs& a, hipStream_t s) {
    if (!a.a0 || !a.w1 || !a.b1 || !a.w2 || !a.b2 || !a.d2) BSY_FAIL(BSY_ERR_ARG, "chain: null pointer");
    if (!chain_supported(a.CA0, a.a1 ? a.CA1 : 0, a.N1, a.keep0, a.LC, a.h2 ? a.CH2 : 0, a.N2))
        BSY_FAIL(BSY_ERR_ARG, "chain: shape (%d + %d -> %d, keep %d + %d; %d + keep -> %d) not supported", a.CA0, a.CA1, a.N1, a.keep0, a.LC, a.CH2, a.N2);
    if (a.M <= 0 || a.M > 0x7fffffffLL) BSY_FAIL(BSY_ERR_ARG, "chain: M out of range");
    if ((a.lda0 & 7) || (a.a1 && (a.lda1 & 7)) || (a.h2 && (a.ldh2 & 7)) || (a.d1 && (a.ldd1 & 7)) || (a.ldd2 & 7) || (a.r1 && (a.ldr1 & 3)) || (a.r2 && (a.ldr2 & 7)))
        BSY_FAIL(BSY_ERR_ARG, "chain: row strides must be multiples of 8 channels");
    if (((uintptr_t)a.a0 & 15) || ((uintptr_t)a.a1 & 15) || ((uintptr_t)a.h2 & 15) || ((uintptr_t)a.w1 & 15) || ((uintptr_t)a.w2 & 15) || ((uintptr_t)a.b1 & 15) ||
        ((uintptr_t)a.b2 & 15) || ((uintptr_t)a.d1 & 15) || ((uintptr_t)a.d2 & 15) || ((uintptr_t)a.r1 & 7) || ((uintptr_t)a.r2 & 15))
        BSY_FAIL(BSY_ERR_ARG, "chain: misaligned pointer");
    const int ldmax = a.lda0 > a.lda1 ? (a.lda0 > a.ldh2 ? a.lda0 : a.ldh2) : (a.lda1 > a.ldh2 ? a.lda1 : a.ldh2);
    if (a.M * (long long)ldmax >= (1LL << 31)) BSY_FAIL(BSY_ERR_ARG, "chain: source view exceeds 2^31 elements (split the batch)");
    ChainK k;
    k.a0 = a.a0; k.a1 = a.a1; k.lda0 = a.lda0; k.lda1 = a.a1 ? a.lda1 : 0; k.CA0 = a.CA0; k.CA1 = a.a1 ? a.CA1 : 0;
    k.spa0 = (unsigned)(((a.M - 1) * a.lda0 + a.CA0) * 2);
    k.spa1 = a.a1 ? (unsigned)(((a.M - 1) * a.lda1 + a.CA1) * 2) : 0u;
    k.w1 = a.w1; k.b1 = a.b1; k.K1pad = round_up(k.CA0 + k.CA1, 32); k.N1 = a.N1; k.act1 = a.act1;
    k.wsp1 = (unsigned)((long long)round_up(a.N1, 128) * k.K1pad * 2);
    k.d1 = a.d1; k.ldd1 = a.ldd1; k.r1 = a.r1; k.ldr1 = a.ldr1; k.keep0 = a.keep0; k.LC = a.LC;
    k.h2 = a.h2; k.ldh2 = a.h2 ? a.ldh2 : 0; k.CH2 = a.h2 ? a.CH2 : 0;
    k.sph2 = a.h2 ? (unsigned)(((a.M - 1) * a.ldh2 + a.CH2) * 2) : 0u;
    k.w2 = a.w2; k.b2 = a.b2; k.K2pad = round_up(k.CH2 + a.LC, 32); k.N2 = a.N2; k.act2 = a.act2;
    k.wsp2 = (unsigned)((long long)round_up(a.N2, 128) * k.K2pad * 2);
    k.d2 = a.d2; k.ldd2 = a.ldd2; k.r2 = a.r2; k.ldr2 = a.ldr2; k.M = (int)a.M;
    k.spd1 = a.d1 ? (unsigned)(((a.M - 1) * a.ldd1 + a.N1) * 2) : 0u;
    k.spd2 = (unsigned)(((a.M - 1) * a.ldd2 + a.N2) * 2);
    if (a.M * (long long)(a.ldd1 > a.ldd2 ? a.ldd1 : a.ldd2) >= (1LL << 31)) BSY_FAIL(BSY_ERR_ARG, "chain: destination view exceeds 2^31 elements (split the batch)");
    static const int dbg = [] { const char* e = getenv("BSY_CHAIN_DBG"); return e ? atoi(e) : 0; }();
    k.dbg = dbg;
    const long long nblk = (a.M + CH_P - 1) / CH_P;
    hipLaunchKernelGGL(chain1x1_kernel, dim3((unsigned)nblk), dim3(512), 0, s, k);
    HIP_TRY(hipGetLastError());
    return BSY_OK;
}
