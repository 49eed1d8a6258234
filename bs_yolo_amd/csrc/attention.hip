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

__global__ __launch_bounds__(256) void attention_kernel(const half_t* __restrict__ qkv, int ld, int N, int heads,
                                                        float scale, half_t* __restrict__ out, int ldo) {
    __shared__ __attribute__((aligned(16))) half_t sK[32 * 32];
    __shared__ __attribute__((aligned(16))) half_t sVT[64 * VT_LD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lrow = lane & 31, lh = lane >> 5;
    const int head = blockIdx.y, b = blockIdx.z;
    const int qoff = head * 32, koff = heads * 32 + head * 32, voff = heads * 64 + head * 64;
    const half_t* base = qkv + (size_t)b * N * ld;

    const int q = blockIdx.x * 128 + wave * 32 + lrow;
    const bool qvalid = q < N;
    half8 qf[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        half8 z = {0, 0, 0, 0, 0, 0, 0, 0};
        qf[ks] = qvalid ? *reinterpret_cast<const half8*>(base + (size_t)q * ld + qoff + 16 * ks + 8 * lh) : z;
    }

    f32x16 o[2];
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[dt][r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;

    const int kkey = tid >> 2, kchunk = tid & 3;  // K staging (threads 0..127)
    const int vkey = tid >> 3, vchunk = tid & 7;  // V staging (all threads)
    const int ntiles = (N + 31) / 32;
    // K / V rows of key tile t are fetched into registers while tile t - 1 is being multiplied (round 3: fetched at the top of their
    // own iteration, every one of the 13 tiles of a 20 x 20 map exposed a full global-load latency: 38 us for 8 GFLOP)
    const half8 z = {0, 0, 0, 0, 0, 0, 0, 0};
    half8 kreg = z, vreg = z;
    auto fetch = [&](const int k0) {
        kreg = z;
        vreg = z;
        if (tid < 128 && k0 + kkey < N)
            kreg = *reinterpret_cast<const half8*>(base + (size_t)(k0 + kkey) * ld + koff + kchunk * 8);
        if (k0 + vkey < N) vreg = *reinterpret_cast<const half8*>(base + (size_t)(k0 + vkey) * ld + voff + vchunk * 8);
    };
    fetch(0);
    for (int t = 0; t < ntiles; ++t) {
        const int k0 = t * 32;
        __syncthreads();  // previous tile fully consumed
        if (tid < 128) *reinterpret_cast<half8*>(sK + kkey * 32 + ((kchunk ^ ((kkey >> 2) & 3)) << 3)) = kreg;
#pragma unroll
        for (int i = 0; i < 8; ++i) sVT[(vchunk * 8 + i) * VT_LD + vkey] = vreg[i];
        __syncthreads();
        if (t + 1 < ntiles) fetch(k0 + 32);

        // S^T tile
        f32x16 s;
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int chunk = 2 * ks + lh;
            const half8 a = *reinterpret_cast<const half8*>(sK + lrow * 32 + ((chunk ^ ((lrow >> 2) & 3)) << 3));
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
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[dt][r] *= alpha;
        // O^T += V^T P^T
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            half8 pb;
#pragma unroll
            for (int j = 0; j < 8; ++j) pb[j] = (half_t)s[8 * ks + j];
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) {
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
    half_t* op = out + ((size_t)b * N + q) * ldo + head * 64;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            half4 v = {(half_t)(o[dt][4 * g] * inv), (half_t)(o[dt][4 * g + 1] * inv), (half_t)(o[dt][4 * g + 2] * inv),
                       (half_t)(o[dt][4 * g + 3] * inv)};
            *reinterpret_cast<half4*>(op + dt * 32 + 8 * g + 4 * lh) = v;
        }
}

int launch_attention(const AttnArgs& a, hipStream_t s) {
    if (a.key_dim != 32 || a.head_dim != 64) BSY_FAIL(BSY_ERR_ARG, "attention: key_dim/head_dim must be 32/64 (got %d/%d)", a.key_dim, a.head_dim);
    if ((a.ld & 7) || (a.ldo & 3) || ((uintptr_t)a.qkv & 15) || ((uintptr_t)a.out & 7) || a.N <= 0 || a.heads <= 0)
        BSY_FAIL(BSY_ERR_ARG, "attention: bad layout");
    if (a.ld < a.heads * 128 || a.ldo < a.heads * 64) BSY_FAIL(BSY_ERR_ARG, "attention: row stride too small");
    dim3 grid((a.N + 127) / 128, a.heads, a.B);
    hipLaunchKernelGGL(attention_kernel, grid, dim3(256), 0, s, a.qkv, a.ld, a.N, a.heads, a.scale, a.out, a.ldo);
    HIP_TRY(hipGetLastError());
    return BSY_OK;
}
