// Attention core of C2PSA in the fp32x engine mode: fp32 q / k / v in, fp32 out, both products on the fp16 matrix pipe with
// split-f16 operands (see conv32x_mfma.hip: v = hi + lo, three MFMAs per product, f32 accumulation).
//
// Replaces Attention.forward's matmul / softmax / matmul (nn/modules/block.py:4279-4286) for fp32 callers at ~2^-21 operand
// precision; the exact mode's kernel (ref32.hip attn32_tiled_kernel: one thread per query, sequential f32 chains on the VALU)
// took 0.42 ms of an 8.4-ms YOLO11s forward for 8 GFLOP.  Structure = attention.hip (flash-style, the query on the lane, S^T never
// leaves the accumulators and becomes the B operand of the second product), with every operand carried as an f16 pair:
//     S^T = Kh Qh + Kh Ql + Kl Qh            P = exp(S^T * scale - m)  (f32, in the accumulators)  -> (Ph, Pl)
//     O^T += V^T_h Ph + V^T_h Pl + V^T_l Ph
// K / V tiles are fetched as f32 (two 16-byte loads per thread and matrix), split in registers and parked in LDS as hi / lo planes.
#include "common.h"

namespace {
#define VT_LD 36
typedef __fp16 pk2x_t __attribute__((ext_vector_type(2)));
union HX8 {
    half8 h;
    pk2x_t p[4];
};
__device__ __forceinline__ void splitx8(const f32x4& a, const f32x4& b, HX8& hi, HX8& lo) {
    const float v[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        hi.p[j] = __builtin_amdgcn_cvt_pkrtz(v[2 * j], v[2 * j + 1]);
        lo.p[j] = __builtin_amdgcn_cvt_pkrtz(v[2 * j] - (float)hi.p[j][0], v[2 * j + 1] - (float)hi.p[j][1]);
    }
}

__global__ __launch_bounds__(256) void attention32x_kernel(const float* __restrict__ qkv, int ld, int N, int heads, float scale,
                                                          float* __restrict__ out, int ldo) {
    __shared__ __attribute__((aligned(16))) half_t sK[2][32 * 32];
    __shared__ __attribute__((aligned(16))) half_t sVT[2][64 * VT_LD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lrow = lane & 31, lh = lane >> 5;
    const int head = blockIdx.y, b = blockIdx.z;
    const int qoff = head * 32, koff = heads * 32 + head * 32, voff = heads * 64 + head * 64;
    const float* base = qkv + (size_t)b * N * ld;
    const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};

    const int q = blockIdx.x * 128 + wave * 32 + lrow;
    const bool qvalid = q < N;
    HX8 qh[2], ql[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        const float* qp = base + (size_t)(qvalid ? q : 0) * ld + qoff + 16 * ks + 8 * lh;
        const f32x4 a = qvalid ? *reinterpret_cast<const f32x4*>(qp) : z4, c = qvalid ? *reinterpret_cast<const f32x4*>(qp + 4) : z4;
        splitx8(a, c, qh[ks], ql[ks]);
    }
    f32x16 o[2];
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[dt][r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;

    const int kkey = tid >> 2, kchunk = tid & 3;  // K staging (threads 0..127): 8 dims of one key
    const int vkey = tid >> 3, vchunk = tid & 7;  // V staging (all threads): 8 dims of one key
    const int ntiles = (N + 31) / 32;
    f32x4 kr[2] = {z4, z4}, vr[2] = {z4, z4};
    auto fetch = [&](const int k0) {
        kr[0] = kr[1] = vr[0] = vr[1] = z4;
        if (tid < 128 && k0 + kkey < N) {
            const float* p = base + (size_t)(k0 + kkey) * ld + koff + kchunk * 8;
            kr[0] = *reinterpret_cast<const f32x4*>(p);
            kr[1] = *reinterpret_cast<const f32x4*>(p + 4);
        }
        if (k0 + vkey < N) {
            const float* p = base + (size_t)(k0 + vkey) * ld + voff + vchunk * 8;
            vr[0] = *reinterpret_cast<const f32x4*>(p);
            vr[1] = *reinterpret_cast<const f32x4*>(p + 4);
        }
    };
    fetch(0);
    for (int t = 0; t < ntiles; ++t) {
        const int k0 = t * 32;
        __syncthreads();  // previous tile fully consumed
        if (tid < 128) {
            HX8 h, l;
            splitx8(kr[0], kr[1], h, l);
            const int o_ = kkey * 32 + ((kchunk ^ ((kkey >> 2) & 3)) << 3);
            *reinterpret_cast<half8*>(sK[0] + o_) = h.h;
            *reinterpret_cast<half8*>(sK[1] + o_) = l.h;
        }
        {
            HX8 h, l;
            splitx8(vr[0], vr[1], h, l);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                sVT[0][(vchunk * 8 + i) * VT_LD + vkey] = h.h[i];
                sVT[1][(vchunk * 8 + i) * VT_LD + vkey] = l.h[i];
            }
        }
        __syncthreads();
        if (t + 1 < ntiles) fetch(k0 + 32);

        f32x16 s;
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int chunk = 2 * ks + lh;
            const int o_ = lrow * 32 + ((chunk ^ ((lrow >> 2) & 3)) << 3);
            const half8 ah = *reinterpret_cast<const half8*>(sK[0] + o_);
            const half8 al = *reinterpret_cast<const half8*>(sK[1] + o_);
            s = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, qh[ks].h, s, 0, 0, 0);
            s = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, ql[ks].h, s, 0, 0, 0);
            s = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, qh[ks].h, s, 0, 0, 0);
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
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            HX8 ph, pl;
            splitx8(f32x4{s[8 * ks], s[8 * ks + 1], s[8 * ks + 2], s[8 * ks + 3]}, f32x4{s[8 * ks + 4], s[8 * ks + 5], s[8 * ks + 6], s[8 * ks + 7]}, ph, pl);
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) {
                const int o_ = (dt * 32 + lrow) * VT_LD + 16 * ks + 4 * lh;
                const half4 h0 = *reinterpret_cast<const half4*>(sVT[0] + o_), h1 = *reinterpret_cast<const half4*>(sVT[0] + o_ + 8);
                const half4 l0 = *reinterpret_cast<const half4*>(sVT[1] + o_), l1 = *reinterpret_cast<const half4*>(sVT[1] + o_ + 8);
                const half8 ah = {h0[0], h0[1], h0[2], h0[3], h1[0], h1[1], h1[2], h1[3]};
                const half8 al = {l0[0], l0[1], l0[2], l0[3], l1[0], l1[1], l1[2], l1[3]};
                o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, ph.h, o[dt], 0, 0, 0);
                o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, pl.h, o[dt], 0, 0, 0);
                o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, ph.h, o[dt], 0, 0, 0);
            }
        }
    }
    if (!qvalid) return;
    const float inv = 1.0f / l_run;
    float* op = out + ((size_t)b * N + q) * ldo + head * 64;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int g = 0; g < 4; ++g)
            *reinterpret_cast<f32x4*>(op + dt * 32 + 8 * g + 4 * lh) = f32x4{o[dt][4 * g] * inv, o[dt][4 * g + 1] * inv, o[dt][4 * g + 2] * inv, o[dt][4 * g + 3] * inv};
}
}  // namespace

bool attn32x_supported(int ld, int ldo, int kd, int hd, const void* qkv, const void* out) {
    return kd == 32 && hd == 64 && !(ld & 3) && !(ldo & 3) && !(((uintptr_t)qkv | (uintptr_t)out) & 15);
}

int launch_attn32x(const float* qkv, int ld, int B, int N, int heads, int kd, int hd, float scale, float* out, int ldo, hipStream_t s) {
    if (!qkv || !out || N <= 0 || heads <= 0 || B <= 0 || !attn32x_supported(ld, ldo, kd, hd, qkv, out))
        BSY_FAIL(BSY_ERR_ARG, "attention32x: key_dim / head_dim must be 32 / 64, row strides multiples of 4, 16-byte aligned views");
    if (ld < heads * 128 || ldo < heads * 64) BSY_FAIL(BSY_ERR_ARG, "attention32x: row stride too small");
    hipLaunchKernelGGL(attention32x_kernel, dim3((N + 127) / 128, heads, B), dim3(256), 0, s, qkv, ld, N, heads, scale, out, ldo);
    HIP_TRY(hipGetLastError());
    return BSY_OK;
}
