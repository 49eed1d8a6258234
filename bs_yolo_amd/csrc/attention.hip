// Attention core of C2PSA for gfx950: out = softmax(q^T k * scale) applied to v, per (image, head).
//
// Replaces the matmul/softmax/matmul lines of Attention.forward (nn/modules/block.py:4279-4286):
//     attn = (q.transpose(-2,-1) @ k) * scale ; attn = attn.softmax(-1) ; x = v @ attn.transpose(-2,-1)
// The reference materialises the (B, heads, N, N) matrix in HBM; here it never leaves registers (flash-style online
// softmax).  qkv is NHWC fp16 with the channel order [q(all heads) | k(all heads) | v(all heads)] (the host permutes
// the rows of the qkv 1x1-conv weight accordingly, bs_yolo_amd/weights.py), so v is a plain channel slice for `pe`.
//
// One workgroup = 4 waves = 128 queries of one (image, head); each wave owns 32 queries and walks all keys in tiles
// of 32.  Both products run on v_mfma_f32_32x32x16_f16 with the QUERY on the lane:
//     S^T[key][q]  = sum_c K[key][c] Q[q][c]     A = K rows (LDS), B = Q (registers, loaded once)
//     O^T[d][q]   += sum_key V^T[d][key] P^T[key][q]   A = V^T (LDS, transposed at staging), B = P^T = the S^T
// accumulator itself converted to fp16 (guide: "an accumulator tile as the next MFMA's operand": element j of lane
// half h of k-step s is key 16s + 8(j>>2) + 4h + (j&3); the V^T fragment is read in the same permuted order).
// Row statistics are per lane (+ one exchange with lane^32).
#include "common.h"

#define VT_LD 36  // halves per V^T row in LDS (72 B: conflict-free 8-byte reads across 32 rows)

// KD = key_dim (channels of q and k per head), HD = head_dim (channels of v per head).  The stock scales give 32 / 64 (block.py:4253-4258:
// num_heads = dim // 64, key_dim = head_dim / 2); other width multiples give other pairs (0.1875: one head of 48 / 96) -- instantiated for
// KD in {16, 32, 48, 64} and HD in {32, 64, 96, 128}.  K rows in LDS are KD + 8 halves long (16 x odd bytes: conflict-free 16-byte reads).
template <int KD, int HD>
__global__ __launch_bounds__(256) void attention_kernel(const half_t* __restrict__ qkv, int ld, int N, int heads,
                                                        float scale, half_t* __restrict__ out, int ldo) {
    constexpr int KS = KD / 16, DT = HD / 32, NCK = KD / 8, NCV = HD / 8, KLD = KD + 8;
    constexpr int NV = (32 * NCV + 255) / 256;  // V pieces per thread per key tile
    static_assert(KD % 16 == 0 && KD <= 64 && HD % 32 == 0 && HD <= 128, "head shape");
    __shared__ __attribute__((aligned(16))) half_t sK[32 * KLD];
    __shared__ __attribute__((aligned(16))) half_t sVT[HD * VT_LD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lrow = lane & 31, lh = lane >> 5;
    const int head = blockIdx.y, b = blockIdx.z;
    const int qoff = head * KD, koff = heads * KD + head * KD, voff = heads * 2 * KD + head * HD;
    const half_t* base = qkv + (size_t)b * N * ld;

    const int q = blockIdx.x * 128 + wave * 32 + lrow;
    const bool qvalid = q < N;
    half8 qf[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        half8 z = {0, 0, 0, 0, 0, 0, 0, 0};
        qf[ks] = qvalid ? *reinterpret_cast<const half8*>(base + (size_t)q * ld + qoff + 16 * ks + 8 * lh) : z;
    }

    f32x16 o[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[dt][r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;

    const int kkey = tid / NCK, kchunk = tid % NCK;  // K staging (threads 0 .. 32 * NCK - 1)
    const int ntiles = (N + 31) / 32;
    // K / V rows of key tile t are fetched into registers while tile t - 1 is being multiplied (round 3: fetched at the top of their
    // own iteration, every one of the 13 tiles of a 20 x 20 map exposed a full global-load latency: 38 us for 8 GFLOP)
    const half8 z = {0, 0, 0, 0, 0, 0, 0, 0};
    half8 kreg = z, vreg[NV];
    auto fetch = [&](const int k0) {
        kreg = z;
        if (tid < 32 * NCK && k0 + kkey < N)
            kreg = *reinterpret_cast<const half8*>(base + (size_t)(k0 + kkey) * ld + koff + kchunk * 8);
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int id = tid + 256 * i;
            const int vkey = id / NCV, vchunk = id % NCV;
            vreg[i] = z;
            if (id < 32 * NCV && k0 + vkey < N) vreg[i] = *reinterpret_cast<const half8*>(base + (size_t)(k0 + vkey) * ld + voff + vchunk * 8);
        }
    };
    fetch(0);
    for (int t = 0; t < ntiles; ++t) {
        const int k0 = t * 32;
        __syncthreads();  // previous tile fully consumed
        if (tid < 32 * NCK) *reinterpret_cast<half8*>(sK + kkey * KLD + kchunk * 8) = kreg;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int id = tid + 256 * i;
            const int vkey = id / NCV, vchunk = id % NCV;
            if (id < 32 * NCV) {
#pragma unroll
                for (int j = 0; j < 8; ++j) sVT[(vchunk * 8 + j) * VT_LD + vkey] = vreg[i][j];
            }
        }
        __syncthreads();
        if (t + 1 < ntiles) fetch(k0 + 32);

        // S^T tile
        f32x16 s;
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const half8 a = *reinterpret_cast<const half8*>(sK + lrow * KLD + 16 * ks + 8 * lh);
            s = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, qf[ks], s, 0, 0, 0);
        }
        float mt = -INFINITY;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int key = k0 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            const float v = key < N ? s[r] * scale : -INFINITY;
            s[r] = v;
            mt = fmaxf(mt, v);
        }
        mt = fmaxf(mt, __shfl_xor(mt, 32));
        const float m_new = fmaxf(m_run, mt);
        const float alpha = __expf(m_run - m_new);
        float rs = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float pv = __expf(s[r] - m_new);
            s[r] = pv;
            rs += pv;
        }
        rs += __shfl_xor(rs, 32);
        l_run = l_run * alpha + rs;
        m_run = m_new;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[dt][r] *= alpha;
        // O^T += V^T P^T
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            half8 pb;
#pragma unroll
            for (int j = 0; j < 8; ++j) pb[j] = (half_t)s[8 * ks + j];
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
                const half_t* vp = sVT + (dt * 32 + lrow) * VT_LD + 16 * ks + 4 * lh;
                const half4 v0 = *reinterpret_cast<const half4*>(vp);
                const half4 v1 = *reinterpret_cast<const half4*>(vp + 8);
                const half8 a = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
                o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, pb, o[dt], 0, 0, 0);
            }
        }
    }
    if (!qvalid) return;
    const float inv = 1.0f / l_run;
    half_t* op = out + ((size_t)b * N + q) * ldo + head * HD;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            half4 v = {(half_t)(o[dt][4 * g] * inv), (half_t)(o[dt][4 * g + 1] * inv), (half_t)(o[dt][4 * g + 2] * inv),
                       (half_t)(o[dt][4 * g + 3] * inv)};
            *reinterpret_cast<half4*>(op + dt * 32 + 8 * g + 4 * lh) = v;
        }
}

bool attention_supported(int key_dim, int head_dim) {
    return key_dim > 0 && key_dim <= 64 && !(key_dim & 15) && head_dim > 0 && head_dim <= 128 && !(head_dim & 31);
}

int launch_attention(const AttnArgs& a, hipStream_t s) {
    if (!attention_supported(a.key_dim, a.head_dim))
        BSY_FAIL(BSY_ERR_ARG, "attention: key_dim must be 16 / 32 / 48 / 64 and head_dim 32 / 64 / 96 / 128 (got %d / %d)", a.key_dim, a.head_dim);
    if ((a.ld & 7) || (a.ldo & 3) || ((uintptr_t)a.qkv & 15) || ((uintptr_t)a.out & 7) || a.N <= 0 || a.heads <= 0)
        BSY_FAIL(BSY_ERR_ARG, "attention: bad layout");
    if (a.ld < a.heads * (2 * a.key_dim + a.head_dim) || a.ldo < a.heads * a.head_dim) BSY_FAIL(BSY_ERR_ARG, "attention: row stride too small");
    dim3 grid((a.N + 127) / 128, a.heads, a.B);
#define ATT_GO(KD_, HD_) \
    if (a.key_dim == KD_ && a.head_dim == HD_) hipLaunchKernelGGL((attention_kernel<KD_, HD_>), grid, dim3(256), 0, s, a.qkv, a.ld, a.N, a.heads, a.scale, a.out, a.ldo)
#define ATT_ROW(KD_) ATT_GO(KD_, 32); ATT_GO(KD_, 64); ATT_GO(KD_, 96); ATT_GO(KD_, 128)
    ATT_ROW(16); ATT_ROW(32); ATT_ROW(48); ATT_ROW(64);
#undef ATT_ROW
#undef ATT_GO
    HIP_TRY(hipGetLastError());
    return BSY_OK;
}
