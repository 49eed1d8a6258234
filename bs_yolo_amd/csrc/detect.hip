// Detect head tail for gfx950: DFL expectation + dist2bbox + sigmoid -> (B, 4+nc+nm, A) channel-major prediction.
//
// Replaces Detect._inference (nn/modules/head.py:100-131), DFL.forward (nn/modules/block.py:58-77),
// make_anchors / dist2bbox (utils/tal.py:371-395) and, for Segment, the mask-coefficient concat (head.py:190-197).
// The reference runs ~10 elementwise/softmax launches over (B,144,A); here one kernel reads the fp32 logits written by
// the last 1x1 convs once (coalesced, through LDS), does the four 16-bin softmax expectations per anchor and writes
// every output channel coalesced along the anchor axis.  HBM-bound: (64+nc+nm)*4 B in, (4+nc+nm)*sizeof(T) out per
// anchor.  All arithmetic fp32.
//
// raw_nchw_kernel rebuilds the `x` list Detect.forward returns next to y (head.py:69-74: cat(cv2(x), cv3(x)) per
// level, BCHW); only launched when the caller asks for it.
#include "common.h"

struct DecodeK {
    const float* box[3];
    const float* cls[3];
    const float* msk[3];
    int ldb[3], ldc[3], ldm[3];
    int h[3], w[3], a0[3];  // a0 = first anchor index of the level
    float stride[3];
    int nl, B, nc, nm, A;
};

// One workgroup = 64 consecutive anchors of one (image, level).  Their logits are contiguous in HBM
// ([64][64] box floats, [64][ldc] class floats), so they are staged through LDS with coalesced 16-byte loads and then
// read column-wise (row stride padded by one dword -> conflict-free); outputs are written one channel at a time,
// 64 consecutive anchors per wave-store.
#define DEC_TA 64
template <typename T>
__global__ __launch_bounds__(256) void decode_kernel(const DecodeK p, T* __restrict__ y) {
    extern __shared__ __attribute__((aligned(16))) float dsm[];
    const int tid = threadIdx.x;
    const int b = blockIdx.y;
    // which level / tile
    int l = 0, t = blockIdx.x;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int nt = i < p.nl ? (p.h[i] * p.w[i] + DEC_TA - 1) / DEC_TA : 0;
        if (i == l && t >= nt && i + 1 < p.nl) { t -= nt; l = i + 1; }
    }
    const int hw = p.h[l] * p.w[l];
    const int la0 = t * DEC_TA;
    const int na = min(DEC_TA, hw - la0);
    if (na <= 0) return;
    const int nch = p.nc + p.nm;
    const int cst = nch + 1;               // padded row stride of the class/mask tile
    float* sbox = dsm;                     // [64][65]
    float* scls = dsm + DEC_TA * 65;       // [64][nc+nm+1]
    float* sd = scls + DEC_TA * cst;       // [64][4] ltrb
    const size_t pix0 = (size_t)b * hw + la0;
    {   // box tile: na*64 floats contiguous (ldb == 64) or strided rows
        const float* bp = p.box[l] + pix0 * p.ldb[l];
        for (int i = tid; i < na * 16; i += 256) {
            const int r = i >> 4, q = i & 15;
            const f32x4 v = *reinterpret_cast<const f32x4*>(bp + (size_t)r * p.ldb[l] + 4 * q);
            float* d = sbox + r * 65 + 4 * q;
            d[0] = v[0]; d[1] = v[1]; d[2] = v[2]; d[3] = v[3];
        }
        const float* cp = p.cls[l] + pix0 * p.ldc[l];
        const int nq = (p.nc + 3) >> 2;    // ldc is a multiple of 4 >= nc
        for (int i = tid; i < na * nq; i += 256) {
            const int r = i / nq, q = i - r * nq;
            const f32x4 v = *reinterpret_cast<const f32x4*>(cp + (size_t)r * p.ldc[l] + 4 * q);
            float* d = scls + r * cst + 4 * q;
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (4 * q + e < p.nc) d[e] = v[e];
        }
        if (p.nm) {
            const float* mp = p.msk[l] + pix0 * p.ldm[l];
            const int nqm = (p.nm + 3) >> 2;
            for (int i = tid; i < na * nqm; i += 256) {
                const int r = i / nqm, q = i - r * nqm;
                const f32x4 v = *reinterpret_cast<const f32x4*>(mp + (size_t)r * p.ldm[l] + 4 * q);
                float* d = scls + r * cst + p.nc + 4 * q;
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (4 * q + e < p.nm) d[e] = v[e];
            }
        }
    }
    __syncthreads();
    const int a = tid & 63, part = tid >> 6;  // 4 waves: wave `part` does DFL side `part` of every anchor
    if (a < na) {
        const float* v = sbox + a * 65 + 16 * part;
        float m = v[0];
#pragma unroll
        for (int i = 1; i < 16; ++i) m = fmaxf(m, v[i]);
        float den = 0.f, num = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const float e = __expf(v[i] - m);
            den += e;
            num += e * (float)i;
        }
        sd[a * 4 + part] = num / den;
    }
    __syncthreads();
    if (a >= na) return;
    const int la = la0 + a;
    T* yp = y + (size_t)b * (4 + nch) * p.A + p.a0[l] + la;
    if (part == 0) {
        const float ax = (float)(la % p.w[l]) + 0.5f, ay = (float)(la / p.w[l]) + 0.5f;
        const float x1 = ax - sd[a * 4 + 0], y1 = ay - sd[a * 4 + 1], x2 = ax + sd[a * 4 + 2], y2 = ay + sd[a * 4 + 3];
        const float st = p.stride[l];
        yp[0] = (T)(((x1 + x2) * 0.5f) * st);
        yp[(size_t)p.A] = (T)(((y1 + y2) * 0.5f) * st);
        yp[(size_t)2 * p.A] = (T)((x2 - x1) * st);
        yp[(size_t)3 * p.A] = (T)((y2 - y1) * st);
    }
    const float* cr = scls + a * cst;
    for (int c = part; c < p.nc; c += 4) yp[(size_t)(4 + c) * p.A] = (T)(1.0f / (1.0f + __expf(-cr[c])));
    for (int c = part; c < p.nm; c += 4) yp[(size_t)(4 + p.nc + c) * p.A] = (T)cr[p.nc + c];
}

int launch_decode(const DecodeArgs& a, hipStream_t s) {
    if (a.nl < 1 || a.nl > 3 || a.nc < 1 || a.B < 1) BSY_FAIL(BSY_ERR_ARG, "decode: bad sizes");
    DecodeK k;
    int A = 0;
    for (int l = 0; l < 3; ++l) {
        const bool on = l < a.nl;
        k.box[l] = on ? a.box[l] : nullptr; k.cls[l] = on ? a.cls[l] : nullptr; k.msk[l] = on ? a.msk[l] : nullptr;
        k.ldb[l] = on ? a.ldb[l] : 0; k.ldc[l] = on ? a.ldc[l] : 0; k.ldm[l] = on ? a.ldm[l] : 0;
        k.h[l] = on ? a.h[l] : 0; k.w[l] = on ? a.w[l] : 0; k.stride[l] = on ? a.stride[l] : 0.f;
        k.a0[l] = A;
        if (on) {
            if (!a.box[l] || !a.cls[l] || (a.ldb[l] & 3) || ((uintptr_t)a.box[l] & 15) || (a.nm && !a.msk[l]))
                BSY_FAIL(BSY_ERR_ARG, "decode: level %d bad layout", l);
            A += a.h[l] * a.w[l];
        }
    }
    if (a.A && a.A != A) BSY_FAIL(BSY_ERR_ARG, "decode: anchor count mismatch (%d vs %d)", a.A, A);
    k.nl = a.nl; k.B = a.B; k.nc = a.nc; k.nm = a.nm; k.A = A;
    int tiles = 0;
    for (int l = 0; l < a.nl; ++l) {
        tiles += (a.h[l] * a.w[l] + DEC_TA - 1) / DEC_TA;
        if ((a.ldc[l] & 3) || a.ldc[l] < ((a.nc + 3) & ~3) || ((uintptr_t)a.cls[l] & 15) ||
            (a.nm && ((a.ldm[l] & 3) || a.ldm[l] < ((a.nm + 3) & ~3) || ((uintptr_t)a.msk[l] & 15))))
            BSY_FAIL(BSY_ERR_ARG, "decode: level %d class/mask rows must be 16-byte aligned multiples of 4 floats", l);
    }
    dim3 grid(tiles, a.B);
    const size_t lds = (size_t)DEC_TA * (65 + (a.nc + a.nm + 1) + 4) * sizeof(float);
    if (lds > 64 * 1024) BSY_FAIL(BSY_ERR_ARG, "decode: nc + nm = %d too large", a.nc + a.nm);
    if (a.y_dtype == BSY_F16)
        hipLaunchKernelGGL(decode_kernel<half_t>, grid, dim3(256), lds, s, k, (half_t*)a.y);
    else if (a.y_dtype == BSY_F32)
        hipLaunchKernelGGL(decode_kernel<float>, grid, dim3(256), lds, s, k, (float*)a.y);
    else
        BSY_FAIL(BSY_ERR_ARG, "decode: y dtype %d unsupported", a.y_dtype);
    HIP_TRY(hipGetLastError());
    return BSY_OK;
}

template <typename T>
__global__ __launch_bounds__(256) void raw_nchw_kernel(const float* __restrict__ box, int ldb,
                                                       const float* __restrict__ cls, int ldc, int hw, int nc,
                                                       T* __restrict__ out) {
    // out (B, 64+nc, h*w); thread = (pixel, channel) with pixels fastest so writes coalesce
    const int pix = blockIdx.x * blockDim.x + threadIdx.x;
    const int c = blockIdx.y;
    const int b = blockIdx.z;
    if (pix >= hw) return;
    const size_t ip = (size_t)b * hw + pix;
    const float v = c < 64 ? box[ip * ldb + c] : cls[ip * ldc + (c - 64)];
    out[((size_t)b * (64 + nc) + c) * hw + pix] = (T)v;
}

int launch_raw_nchw(const float* box, int ldb, const float* cls, int ldc, int B, int h, int w, int nc, void* out,
                    int out_dtype, hipStream_t s) {
    if (!box || !cls || !out) BSY_FAIL(BSY_ERR_ARG, "raw_nchw: null pointer");
    dim3 grid((h * w + 255) / 256, 64 + nc, B);
    if (out_dtype == BSY_F16)
        hipLaunchKernelGGL(raw_nchw_kernel<half_t>, grid, dim3(256), 0, s, box, ldb, cls, ldc, h * w, nc, (half_t*)out);
    else if (out_dtype == BSY_F32)
        hipLaunchKernelGGL(raw_nchw_kernel<float>, grid, dim3(256), 0, s, box, ldb, cls, ldc, h * w, nc, (float*)out);
    else
        BSY_FAIL(BSY_ERR_ARG, "raw_nchw: dtype %d unsupported", out_dtype);
    HIP_TRY(hipGetLastError());
    return BSY_OK;
}

// NHWC fp16 view -> BCHW (Segment's prototype masks `p`, head.py:184 / block.py:95-97).  32 x 32 (pixel x channel)
// tiles through LDS so that both the NHWC reads (channels fastest) and the BCHW writes (pixels fastest) coalesce.
template <typename T>
__global__ __launch_bounds__(256) void nhwc2nchw_kernel(const half_t* __restrict__ src, int ld, int C, int hw,
                                                        T* __restrict__ out) {
    __shared__ half_t t[32][33];
    const int b = blockIdx.z, p0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    for (int r = ty; r < 32; r += 8) {
        const int p = p0 + r, c = c0 + tx;
        t[r][tx] = (p < hw && c < C) ? src[((size_t)b * hw + p) * ld + c] : (half_t)0.f;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        const int c = c0 + r, p = p0 + tx;
        if (c < C && p < hw) out[((size_t)b * C + c) * hw + p] = (T)(float)t[tx][r];
    }
}

int launch_nhwc2nchw(const half_t* src, int ld, int B, int C, int hw, void* out, int out_dtype, hipStream_t s) {
    if (!src || !out || B <= 0 || C <= 0 || hw <= 0) BSY_FAIL(BSY_ERR_ARG, "nhwc2nchw: bad argument");
    dim3 grid((hw + 31) / 32, (C + 31) / 32, B);
    if (out_dtype == BSY_F16)
        hipLaunchKernelGGL(nhwc2nchw_kernel<half_t>, grid, dim3(256), 0, s, src, ld, C, hw, (half_t*)out);
    else if (out_dtype == BSY_F32)
        hipLaunchKernelGGL(nhwc2nchw_kernel<float>, grid, dim3(256), 0, s, src, ld, C, hw, (float*)out);
    else
        BSY_FAIL(BSY_ERR_ARG, "nhwc2nchw: dtype %d unsupported", out_dtype);
    HIP_TRY(hipGetLastError());
    return BSY_OK;
}
