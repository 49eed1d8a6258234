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
//   stage 1: for every 256- (or 128-) cout pass: K loop over the HBM sources (pixels + weights staged by LDS-DMA, 64-deep K-steps,
//            two ring stages: conv_mfma.hip's ALIGNED BK-64 form) -> SiLU -> f16 -> optional HBM store through an LDS staging tile
//            (coalesced 16-byte pieces) + the kept channel range parked in the RESIDENT tile, laid out as K-step pixel blocks
//            [block][128 px][64 ch] with the ring's XOR swizzle, so that stage 2 reads its B fragments from it exactly as from a ring stage;
//   stage 2: per cout pass: K loop = HBM part (ring) then resident blocks (weights only through the ring) -> epilogue -> HBM.
// Every accumulator starts at the bias and takes its products in ascending k, 16 at a time, through v_mfma_f32_32x32x16_f16 -- the K walk
// (order 0) of the un-fused 1x1 launches; activations are rounded to f16 where the un-fused pair stores them, the shortcut is added to
// the ROUNDED value in f32 and rounded again (conv_mfma.hip conv_epilogue_lds): the same bits (tests/test_gpu_parity.py
// test_engine_chain_fusion_is_bit_identical).
// LDS: ring 2 x (128 + 256) x 64 halves = 96 KiB (staging tile aliases it between passes) + resident tile <= 256 ch x 128 px = 64 KiB.
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
};

namespace {
#if defined(__HIP_DEVICE_COMPILE__)
typedef __amdgpu_buffer_rsrc_t ch_rsrc_t;
__device__ __forceinline__ ch_rsrc_t ch_make_rsrc(const void* base, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), (short)0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ void ch_dma16(ch_rsrc_t r, unsigned voff, unsigned soff, half_t* lds_wave_base) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, lds_wave_base, 16, (int)voff, (int)soff, 0, 0);
}
#else
typedef int ch_rsrc_t;
__device__ __forceinline__ ch_rsrc_t ch_make_rsrc(const void*, unsigned) { return 0; }
__device__ __forceinline__ void ch_dma16(ch_rsrc_t, unsigned, unsigned, half_t*) {}
#endif
constexpr unsigned CH_OOB = 0xFFFFFFF0u;  // out of range for every descriptor: the lane's 16 bytes land as zeros
constexpr int CH_P = 128, CH_BK = 64;
constexpr int CH_PBLK = CH_P * CH_BK;            // one K-step of pixels: [128][64] halves, 16 KiB
constexpr int CH_STAGE = (CH_P + 256) * CH_BK;   // ring stage: pixels + up to 256 weight rows
constexpr int CH_RING = 2 * CH_STAGE;
constexpr int CH_TILE = 4 * CH_PBLK;             // resident tile: up to 256 channels

struct ChSeg {  // one HBM K segment of a pass
    ch_rsrc_t rs;
    unsigned off[2];  // per staged pixel row of this lane: byte offset of the row's first channel + this lane's swizzled chunk, or CH_OOB
    int nsteps;
};

// K loop of one pass: acc[a][b] += W[cout0 + (wn NT + a) 32 ..][k] * B[k][wm 64 + b 32 ..] over seg0, seg1 (HBM, through the ring) and
// then `nl` resident blocks starting at block lb0.
template <int NT>
__device__ __forceinline__ void chain_pass(half_t* smem, const ChSeg& s0, const ChSeg& s1, const int nl, const int lb0, const ch_rsrc_t rsw,
                                           const unsigned kpad, const int cout0, f32x16 (&acc)[NT][2], const int wave, const int lane) {
    constexpr int WIW = 2 * NT;  // weight DMA instructions per wave and K-step (128 NT rows, 8 rows each, 8 waves)
    half_t* ring = smem;
    const half_t* tile = smem + CH_RING;
    const int rsub = lane >> 3, slot = lane & 7;
    const int kc0 = slot ^ (rsub >> 1), kc1 = slot ^ (4 | (rsub >> 1));
    const int wn = wave >> 1, wm = wave & 1, lrow = lane & 31, lh = lane >> 5;
    unsigned woff[WIW];
#pragma unroll
    for (int j = 0; j < WIW; ++j) woff[j] = (unsigned)((wave * WIW + j) * 8 + rsub) * kpad * 2u + 16u * (unsigned)((j & 1) ? kc1 : kc0);
    const int n0 = s0.nsteps, n01 = n0 + s1.nsteps, nk = n01 + nl;
    auto issue = [&](const int kt, const int st) {
        half_t* sP = ring + st * CH_STAGE;
        half_t* sW = sP + CH_PBLK;
        if (kt < n0) {
#pragma unroll
            for (int i = 0; i < 2; ++i) ch_dma16(s0.rs, s0.off[i], 128u * (unsigned)kt, sP + (wave * 2 + i) * 512);
        } else if (kt < n01) {
#pragma unroll
            for (int i = 0; i < 2; ++i) ch_dma16(s1.rs, s1.off[i], 128u * (unsigned)(kt - n0), sP + (wave * 2 + i) * 512);
        }
#pragma unroll
        for (int j = 0; j < WIW; ++j) ch_dma16(rsw, woff[j], 2u * ((unsigned)cout0 * kpad + 64u * (unsigned)kt), sW + (wave * WIW + j) * 512);
    };
    issue(0, 0);
    for (int kt = 0; kt < nk; ++kt) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();  // K-step kt landed for every wave; every wave is done reading the other stage
        if (kt + 1 < nk) issue(kt + 1, (kt + 1) & 1);
        const half_t* sP = kt < n01 ? ring + (kt & 1) * CH_STAGE : tile + (lb0 + kt - n01) * CH_PBLK;
        const half_t* sW = ring + (kt & 1) * CH_STAGE + CH_PBLK;
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
        }
    }
}

// Epilogue of one pass.  STAGE1: the kept couts go to the resident tile; either way the pass's tile goes to HBM through the staging tile
// when `dst` is set.  Activation -> f16, then the shortcut operand added to the rounded value in f32 and rounded again.
template <int NT, bool STAGE1>
__device__ __forceinline__ void chain_epilogue(half_t* smem, f32x16 (&acc)[NT][2], const int act, half_t* dst, const int ldd, const half_t* res,
                                               const int ldr, const int keep0, const int LC, const int cout0, const int m0, const int M,
                                               const int wave, const int lane, const int tid) {
    constexpr int NP = 128 * NT, LDT = NP + 8;
    half_t* stage = smem;
    half_t* tile = smem + CH_RING;
    const int wn = wave >> 1, wm = wave & 1, lrow = lane & 31, lh = lane >> 5;
    __syncthreads();  // every wave has left the K loop: the ring may become the staging tile
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        const int prow = wm * 64 + b * 32 + lrow;
#pragma unroll
        for (int a = 0; a < NT; ++a) {
            const int cl = (wn * NT + a) * 32;
            const int ck0 = cout0 + cl - keep0;
            const bool keep = STAGE1 && ck0 >= 0 && ck0 < LC;  // wave-uniform (keep0, LC multiples of 64)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int c = cl + 8 * g + 4 * lh;
                f32x4 t = f32x4{acc[a][b][4 * g], acc[a][b][4 * g + 1], acc[a][b][4 * g + 2], acc[a][b][4 * g + 3]};
                if (act) t = silu4_f(t);
                half4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = (half_t)t[e];
                if (STAGE1 && res && m0 + prow < M) {  // stage-1 shortcut: in registers (the resident tile must hold the sum)
                    const half4 r = *reinterpret_cast<const half4*>(res + (size_t)(m0 + prow) * ldr + cout0 + c);
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = (half_t)((float)o[e] + (float)r[e]);
                }
                if (dst) *reinterpret_cast<half4*>(stage + prow * LDT + c) = o;
                if (keep) {
                    const int ck = ck0 + 8 * g + 4 * lh, cc = ck & 63;
                    *reinterpret_cast<half4*>(tile + (ck >> 6) * CH_PBLK + prow * CH_BK + (((cc >> 3) ^ ((prow >> 1) & 7)) << 3) + (cc & 7)) = o;
                }
            }
        }
    }
    __syncthreads();
    if (!dst) return;
    constexpr int CPRW = NP / 8, ITER = CH_P * CPRW / 512;
#pragma unroll
    for (int i = 0; i < ITER; ++i) {
        const int id = tid + 512 * i;
        const int row = id / CPRW, cc = (id % CPRW) * 8;
        const int m = m0 + row;
        if (m >= M) continue;
        half8 v = *reinterpret_cast<const half8*>(stage + row * LDT + cc);
        if (!STAGE1 && res) {
            const half8 r = *reinterpret_cast<const half8*>(res + (size_t)m * ldr + cout0 + cc);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = (half_t)((float)v[e] + (float)r[e]);
        }
        *reinterpret_cast<half8*>(dst + (size_t)m * ldd + cout0 + cc) = v;
    }
    __syncthreads();  // the staging tile has been read: the next pass's first DMA may overwrite the ring
}

template <int NT>
__device__ __forceinline__ void chain_bias(f32x16 (&acc)[NT][2], const float* bias, const int cout0, const int wave, const int lane) {
#pragma unroll
    for (int a = 0; a < NT; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc_bias(acc[a][b], bias + cout0 + ((wave >> 1) * NT + a) * 32, lane >> 5);
}
}  // namespace

__global__ __launch_bounds__(512) void chain1x1_kernel(const ChainK p) {
    __shared__ __attribute__((aligned(16))) half_t smem[CH_RING + CH_TILE];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m0 = blockIdx.x * CH_P;
    const int rsub = lane >> 3, slot = lane & 7;
    const int kc[2] = {slot ^ (rsub >> 1), slot ^ (4 | (rsub >> 1))};
    // byte offsets of this lane's two staged pixel rows in each HBM source
    ChSeg sa0, sa1, sh2, none;
    none.rs = ch_make_rsrc(p.a0, 0u); none.off[0] = none.off[1] = CH_OOB; none.nsteps = 0;
    sa0.rs = ch_make_rsrc(p.a0, p.spa0); sa0.nsteps = p.CA0 >> 6;
    sa1.rs = ch_make_rsrc(p.a1 ? p.a1 : p.a0, p.a1 ? p.spa1 : 0u); sa1.nsteps = p.a1 ? p.CA1 >> 6 : 0;
    sh2.rs = ch_make_rsrc(p.h2 ? p.h2 : p.a0, p.h2 ? p.sph2 : 0u); sh2.nsteps = p.h2 ? p.CH2 >> 6 : 0;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int m = m0 + (wave * 2 + i) * 8 + rsub;
        const bool v = m < p.M;
        sa0.off[i] = v ? 2u * (unsigned)m * (unsigned)p.lda0 + 16u * (unsigned)kc[i] : CH_OOB;
        sa1.off[i] = v ? 2u * (unsigned)m * (unsigned)p.lda1 + 16u * (unsigned)kc[i] : CH_OOB;
        sh2.off[i] = v ? 2u * (unsigned)m * (unsigned)p.ldh2 + 16u * (unsigned)kc[i] : CH_OOB;
    }
    const ch_rsrc_t rw1 = ch_make_rsrc(p.w1, p.wsp1), rw2 = ch_make_rsrc(p.w2, p.wsp2);
    // ---- stage 1 ----
    if (!(p.N1 & 255)) {
        for (int c0 = 0; c0 < p.N1; c0 += 256) {
            f32x16 acc[2][2];
            chain_bias<2>(acc, p.b1, c0, wave, lane);
            chain_pass<2>(smem, sa0, sa1, 0, 0, rw1, (unsigned)p.K1pad, c0, acc, wave, lane);
            chain_epilogue<2, true>(smem, acc, p.act1, p.d1, p.ldd1, p.r1, p.ldr1, p.keep0, p.LC, c0, m0, p.M, wave, lane, tid);
        }
    } else {
        for (int c0 = 0; c0 < p.N1; c0 += 128) {
            f32x16 acc[1][2];
            chain_bias<1>(acc, p.b1, c0, wave, lane);
            chain_pass<1>(smem, sa0, sa1, 0, 0, rw1, (unsigned)p.K1pad, c0, acc, wave, lane);
            chain_epilogue<1, true>(smem, acc, p.act1, p.d1, p.ldd1, p.r1, p.ldr1, p.keep0, p.LC, c0, m0, p.M, wave, lane, tid);
        }
    }
    // ---- stage 2 ----
    const int nl = p.LC >> 6;
    if (!(p.N2 & 255)) {
        for (int c0 = 0; c0 < p.N2; c0 += 256) {
            f32x16 acc[2][2];
            chain_bias<2>(acc, p.b2, c0, wave, lane);
            chain_pass<2>(smem, sh2, none, nl, 0, rw2, (unsigned)p.K2pad, c0, acc, wave, lane);
            chain_epilogue<2, false>(smem, acc, p.act2, p.d2, p.ldd2, p.r2, p.ldr2, 0, 0, c0, m0, p.M, wave, lane, tid);
        }
    } else {
        for (int c0 = 0; c0 < p.N2; c0 += 128) {
            f32x16 acc[1][2];
            chain_bias<1>(acc, p.b2, c0, wave, lane);
            chain_pass<1>(smem, sh2, none, nl, 0, rw2, (unsigned)p.K2pad, c0, acc, wave, lane);
            chain_epilogue<1, false>(smem, acc, p.act2, p.d2, p.ldd2, p.r2, p.ldr2, 0, 0, c0, m0, p.M, wave, lane, tid);
        }
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
    const long long nblk = (a.M + CH_P - 1) / CH_P;
    hipLaunchKernelGGL(chain1x1_kernel, dim3((unsigned)nblk), dim3(512), 0, s, k);
    HIP_TRY(hipGetLastError());
    return BSY_OK;
}
