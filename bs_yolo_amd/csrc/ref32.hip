// fp32 correctness mode of the engine ("precision = fp32" plans): the same op list as the fp16 product path, executed with
// fp32 activation storage and fp32 arithmetic by deliberately simple kernels (one thread per output element or small group,
// sequential fmaf over K).  It exists for callers that hand the engine fp32 images -- the reference's predict() default is
// half=False (engine/predictor.py:131) -- and expect the fp32 model's numbers: the fp16-storage path differs from the fp32
// reference by up to 6e-3 in a score, this one by ~1e-5, inside the north-star's 1e-3.  Slow by
// design (a few TFLOP/s); never on the benchmarked path.  Layout: NHWC f32, views (pointer incl. channel offset, ld).
// Each kernel restates the reference arithmetic it replaces: Conv.forward_fuse conv.py:149-151, DWConv :224-229,
// SPPF block.py:3145-3149, Attention :4279-4286, MSCAAttention nn/Addmodules/MSCA.py:53-88, ELA ELA.py:77-101.
#include <stdlib.h>

#include "common.h"

namespace {
__device__ __forceinline__ float silu32(float x) { return x / (1.0f + expf(-x)); }
__device__ __forceinline__ float sigmoid32(float x) { return 1.0f / (1.0f + expf(-x)); }

// ---- dense conv, k x k, stride 1 / 2, up to two concatenated sources (each optionally read through nearest x2) ----------
template <bool FIRST>
__global__ __launch_bounds__(256) void conv32_kernel(const Conv32Args a) {
    constexpr int PX = 4;  // output pixels per thread: one weight load feeds four FMAs
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    const int o = (int)(idx % a.Cout);
    const long long m0 = idx / a.Cout * PX;
    const long long M = (long long)a.B * a.OH * a.OW;
    if (m0 >= M) return;
    const int Cin = a.C0 + a.C1;
    float acc[PX];
    int n[PX], oh[PX], ow[PX];
    bool ok[PX];
#pragma unroll
    for (int j = 0; j < PX; ++j) {
        const long long m = m0 + j;
        ok[j] = m < M;
        const long long mm = ok[j] ? m : m0;
        n[j] = (int)(mm / (a.OH * a.OW));
        const int rem = (int)(mm - (long long)n[j] * a.OH * a.OW);
        oh[j] = rem / a.OW;
        ow[j] = rem - oh[j] * a.OW;
        acc[j] = a.bias[o];
    }
    for (int kh = 0; kh < a.ks; ++kh)
        for (int kw = 0; kw < a.ks; ++kw) {
            const float* wrow = a.w + (size_t)((kh * a.ks + kw) * Cin) * a.Cout + o;
#pragma unroll
            for (int j = 0; j < PX; ++j) {
                const int iy = oh[j] * a.stride - a.pad + kh, ix = ow[j] * a.stride - a.pad + kw;
                if (!ok[j] || (unsigned)iy >= (unsigned)a.H || (unsigned)ix >= (unsigned)a.W) continue;
                if (FIRST) {  // BCHW image, f16 or f32
                    for (int c = 0; c < a.C0; ++c) {
                        const size_t ii = ((size_t)(n[j] * a.C0 + c) * a.H + iy) * a.W + ix;
                        const float xv = a.src_dtype == BSY_F16 ? (float)reinterpret_cast<const half_t*>(a.src0)[ii]
                                                                : reinterpret_cast<const float*>(a.src0)[ii];
                        acc[j] = fmaf(xv, wrow[(size_t)c * a.Cout], acc[j]);
                    }
                } else {
                    const float* x0 = reinterpret_cast<const float*>(a.src0) +
                                      ((size_t)(n[j] * (a.H >> a.up0) + (iy >> a.up0)) * (a.W >> a.up0) + (ix >> a.up0)) * a.ld0;
                    for (int c = 0; c < a.C0; ++c) acc[j] = fmaf(x0[c], wrow[(size_t)c * a.Cout], acc[j]);
                    if (a.C1) {
                        const float* x1 = reinterpret_cast<const float*>(a.src1) +
                                          ((size_t)(n[j] * (a.H >> a.up1) + (iy >> a.up1)) * (a.W >> a.up1) + (ix >> a.up1)) * a.ld1;
                        const float* w1 = wrow + (size_t)a.C0 * a.Cout;
                        for (int c = 0; c < a.C1; ++c) acc[j] = fmaf(x1[c], w1[(size_t)c * a.Cout], acc[j]);
                    }
                }
            }
        }
#pragma unroll
    for (int j = 0; j < PX; ++j) {
        if (!ok[j]) continue;
        float v = a.act ? silu32(acc[j]) : acc[j];
        const size_t pix = (size_t)(n[j] * a.OH + oh[j]) * a.OW + ow[j];
        if (a.res) v += a.res[pix * a.ldr + o];  // shortcut: after the activation (Bottleneck / PSABlock)
        size_t dp = pix;
        if (a.dst_scale != 1)
            dp = ((size_t)n[j] * (a.OH * a.dst_scale) + (oh[j] * a.dst_scale + a.dst_dy)) * (size_t)(a.OW * a.dst_scale) +
                 (ow[j] * a.dst_scale + a.dst_dx);
        a.dst[dp * a.ldd + o] = v;
    }
}

// ---- depthwise kh x kw, stride 1 / 2, "same" padding; SiLU on channels < act_c; optional residual ----------------------
__global__ __launch_bounds__(256) void dw32_kernel(const Dw32Args a) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long total = (long long)a.B * a.OH * a.OW * a.C;
    if (idx >= total) return;
    const int c = (int)(idx % a.C);
    long long t = idx / a.C;
    const int ow = (int)(t % a.OW);
    t /= a.OW;
    const int oh = (int)(t % a.OH);
    const int n = (int)(t / a.OH);
    float acc = a.b[c];
    const int ph = a.kh / 2, pw = a.kw / 2;
    for (int i = 0; i < a.kh; ++i)
        for (int j = 0; j < a.kw; ++j) {
            const int iy = oh * a.stride - ph + i, ix = ow * a.stride - pw + j;
            if ((unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W)
                acc = fmaf(a.src[((size_t)(n * a.H + iy) * a.W + ix) * a.lds + c], a.w[(size_t)(i * a.kw + j) * a.wld + c], acc);
        }
    float v = c < a.act_c ? silu32(acc) : acc;
    const size_t pix = (size_t)(n * a.OH + oh) * a.OW + ow;
    if (a.res) v += a.res[pix * a.ldr + c];
    a.dst[pix * a.ldd + c] = v;
}

// The same conv with four channels per thread (16-byte loads and stores; C, row strides and the weight row stride multiples of 4):
// per element the same taps in the same order as dw32_kernel, so the same bits.  The scalar form moved 4 bytes per lane and load
// instruction: 1.4 ms of a 20-ms YOLO11s forward for seven depthwise layers.
__global__ __launch_bounds__(256) void dw32x4_kernel(const Dw32Args a) {
    const int C4 = a.C >> 2;
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long long)a.B * a.OH * a.OW * C4) return;
    const int c = (int)(idx % C4) * 4;
    long long t = idx / C4;
    const int ow = (int)(t % a.OW);
    t /= a.OW;
    const int oh = (int)(t % a.OH);
    const int n = (int)(t / a.OH);
    f32x4 acc = *reinterpret_cast<const f32x4*>(a.b + c);
    const int ph = a.kh / 2, pw = a.kw / 2;
    for (int i = 0; i < a.kh; ++i)
        for (int j = 0; j < a.kw; ++j) {
            const int iy = oh * a.stride - ph + i, ix = ow * a.stride - pw + j;
            if ((unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W) {
                const f32x4 x = *reinterpret_cast<const f32x4*>(a.src + ((size_t)(n * a.H + iy) * a.W + ix) * a.lds + c);
                const f32x4 w = *reinterpret_cast<const f32x4*>(a.w + (size_t)(i * a.kw + j) * a.wld + c);
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[e] = fmaf(x[e], w[e], acc[e]);
            }
        }
    const size_t pix = (size_t)(n * a.OH + oh) * a.OW + ow;
    f32x4 v;
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = c + e < a.act_c ? silu32(acc[e]) : acc[e];
    if (a.res) {
        const f32x4 r = *reinterpret_cast<const f32x4*>(a.res + pix * a.ldr + c);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] += r[e];
    }
    *reinterpret_cast<f32x4*>(a.dst + pix * a.ldd + c) = v;
}

// ---- SPPF: three chained MaxPool2d(5, 1, 2) of channels [0, C) into [C, 2C), [2C, 3C), [3C, 4C) of the same rows ----------
__global__ __launch_bounds__(256) void sppf32_kernel(float* buf, int ld, int B, int H, int W, int C) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long long)B * H * W * C) return;
    const int c = (int)(idx % C);
    long long t = idx / C;
    const int x = (int)(t % W);
    t /= W;
    const int y = (int)(t % H);
    const int n = (int)(t / H);
    float m[3] = {-INFINITY, -INFINITY, -INFINITY};  // chained 5x5 pools with -inf padding = windows of 5, 9, 13 clipped to the map
    for (int dy = -6; dy <= 6; ++dy)
        for (int dx = -6; dx <= 6; ++dx) {
            const int iy = y + dy, ix = x + dx;
            if ((unsigned)iy >= (unsigned)H || (unsigned)ix >= (unsigned)W) continue;
            const float v = buf[((size_t)(n * H + iy) * W + ix) * ld + c];
            const int r = max(abs(dy), abs(dx));
            if (r <= 2) m[0] = fmaxf(m[0], v);
            if (r <= 4) m[1] = fmaxf(m[1], v);
            m[2] = fmaxf(m[2], v);
        }
    float* o = buf + ((size_t)(n * H + y) * W + x) * ld + c;
    o[C] = m[0];
    o[2 * C] = m[1];
    o[3 * C] = m[2];
}

// four channels per thread (16-byte loads / stores; C and ld multiples of 4); maxima are exact: the same values
__global__ __launch_bounds__(256) void sppf32x4_kernel(float* buf, int ld, int B, int H, int W, int C) {
    const int C4 = C >> 2;
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long long)B * H * W * C4) return;
    const int c = (int)(idx % C4) * 4;
    long long t = idx / C4;
    const int x = (int)(t % W);
    t /= W;
    const int y = (int)(t % H);
    const int n = (int)(t / H);
    f32x4 m0, m1, m2;
#pragma unroll
    for (int e = 0; e < 4; ++e) m0[e] = m1[e] = m2[e] = -INFINITY;
    for (int dy = -6; dy <= 6; ++dy)
        for (int dx = -6; dx <= 6; ++dx) {
            const int iy = y + dy, ix = x + dx;
            if ((unsigned)iy >= (unsigned)H || (unsigned)ix >= (unsigned)W) continue;
            const f32x4 v = *reinterpret_cast<const f32x4*>(buf + ((size_t)(n * H + iy) * W + ix) * ld + c);
            const int r = max(abs(dy), abs(dx));
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if (r <= 2) m0[e] = fmaxf(m0[e], v[e]);
                if (r <= 4) m1[e] = fmaxf(m1[e], v[e]);
                m2[e] = fmaxf(m2[e], v[e]);
            }
        }
    float* o = buf + ((size_t)(n * H + y) * W + x) * ld + c;
    *reinterpret_cast<f32x4*>(o + C) = m0;
    *reinterpret_cast<f32x4*>(o + 2 * C) = m1;
    *reinterpret_cast<f32x4*>(o + 3 * C) = m2;
}

// 3 x 3, stride 1: four outputs along x per thread from a register window of 3 x 6 pieces (18 16-byte loads for four outputs instead
// of 36; the depthwise layers of Detect's class branch at 80 x 80 x 128 were bound by load issue, 0.19 ms each for 0.42 GB of traffic).
// Every output still sums its taps in (kh, kw) order starting at the bias, and a tap outside the map contributes fmaf(0, w, acc) = acc:
// the same bits as dw32x4_kernel / dw32_kernel.
__global__ __launch_bounds__(256) void dw32_3x3x4_kernel(const Dw32Args a) {
    const int C4 = a.C >> 2, XB = (a.OW + 3) >> 2;
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long long)a.B * a.OH * XB * C4) return;
    const int c = (int)(idx % C4) * 4;
    long long t = idx / C4;
    const int ow0 = (int)(t % XB) * 4;
    t /= XB;
    const int oh = (int)(t % a.OH);
    const int n = (int)(t / a.OH);
    const f32x4 bias = *reinterpret_cast<const f32x4*>(a.b + c);
    f32x4 acc[4] = {bias, bias, bias, bias};
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int iy = oh - 1 + i;
        const bool rowok = (unsigned)iy < (unsigned)a.H;
        const float* row = a.src + ((size_t)(n * a.H + (rowok ? iy : 0)) * a.W) * a.lds + c;
        f32x4 x[6];
#pragma unroll
        for (int col = 0; col < 6; ++col) {
            const int ix = ow0 - 1 + col;
            x[col] = (rowok && (unsigned)ix < (unsigned)a.W) ? *reinterpret_cast<const f32x4*>(row + (size_t)ix * a.lds) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const f32x4 w = *reinterpret_cast<const f32x4*>(a.w + (size_t)(i * 3 + j) * a.wld + c);
#pragma unroll
            for (int o = 0; o < 4; ++o)
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[o][e] = fmaf(x[o + j][e], w[e], acc[o][e]);
        }
    }
#pragma unroll
    for (int o = 0; o < 4; ++o) {
        if (ow0 + o >= a.OW) break;
        const size_t pix = (size_t)(n * a.OH + oh) * a.OW + ow0 + o;
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = c + e < a.act_c ? silu32(acc[o][e]) : acc[o][e];
        if (a.res) {
            const f32x4 r = *reinterpret_cast<const f32x4*>(a.res + pix * a.ldr + c);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] += r[e];
        }
        *reinterpret_cast<f32x4*>(a.dst + pix * a.ldd + c) = v;
    }
}

// SPPF's three chained 5 x 5 max pools out of LDS: one workgroup per (image, 4 V channels) holds the map in two LDS buffers and
// runs every pool as a row pass and a column pass (separable; -inf padding = windows clipped to the map), writing the three results
// behind the input channels.  Maxima are exact, so this returns sppf32_kernel's bits; that kernel read 169 taps per output (0.22 ms
// per forward for 26 MB in and 79 MB out), this one reads 30.
template <int V>
__global__ __launch_bounds__(256) void sppf32_lds_kernel(float* buf, int ld, int H, int W, int C) {
    extern __shared__ __attribute__((aligned(16))) unsigned char sppf_smem[];
    const int HW = H * W, NV = HW * V;
    f32x4* A = reinterpret_cast<f32x4*>(sppf_smem);
    f32x4* Bf = A + NV;
    const int tid = threadIdx.x, n = blockIdx.y, c0 = blockIdx.x * 4 * V;
    float* base = buf + (size_t)n * HW * ld + c0;
    for (int id = tid; id < NV; id += 256) A[id] = *reinterpret_cast<const f32x4*>(base + (size_t)(id / V) * ld + 4 * (id % V));
    __syncthreads();
    for (int pool = 1; pool <= 3; ++pool) {
        for (int id = tid; id < NV; id += 256) {  // row pass
            const int p = id / V, x = p % W;
            f32x4 m = A[id];
#pragma unroll
            for (int d = -2; d <= 2; ++d)
                if (d && (unsigned)(x + d) < (unsigned)W) {
                    const f32x4 v = A[id + d * V];
#pragma unroll
                    for (int e = 0; e < 4; ++e) m[e] = fmaxf(m[e], v[e]);
                }
            Bf[id] = m;
        }
        __syncthreads();
        for (int id = tid; id < NV; id += 256) {  // column pass; the result is the next pool's input and this pool's output
            const int p = id / V, y = p / W;
            f32x4 m = Bf[id];
#pragma unroll
            for (int d = -2; d <= 2; ++d)
                if (d && (unsigned)(y + d) < (unsigned)H) {
                    const f32x4 v = Bf[id + d * W * V];
#pragma unroll
                    for (int e = 0; e < 4; ++e) m[e] = fmaxf(m[e], v[e]);
                }
            A[id] = m;
            *reinterpret_cast<f32x4*>(base + (size_t)p * ld + pool * C + 4 * (id % V)) = m;
        }
        __syncthreads();
    }
}

// ---- attention: out[i] = sum_j softmax_j(scale * q_i . k_j) v_j per (image, head); qkv = [q | k | v] by heads ----------------
#define ATT32_MAXD 128
__global__ __launch_bounds__(64) void attn32_kernel(const float* qkv, int ld, int B, int N, int heads, int kd, int hd, float scale,
                                                   float* out, int ldo) {
    const int i = blockIdx.x * 64 + threadIdx.x, head = blockIdx.y, b = blockIdx.z;
    if (i >= N) return;
    const float* base = qkv + (size_t)b * N * ld;
    const float* q = base + (size_t)i * ld + head * kd;
    const int koff = heads * kd + head * kd, voff = 2 * heads * kd + head * hd;
    float acc[ATT32_MAXD];
    for (int d = 0; d < hd; ++d) acc[d] = 0.f;
    float mx = -INFINITY, den = 0.f;
    for (int j = 0; j < N; ++j) {
        const float* kj = base + (size_t)j * ld + koff;
        float s = 0.f;
        for (int c = 0; c < kd; ++c) s = fmaf(q[c], kj[c], s);
        s *= scale;
        const float nm = fmaxf(mx, s);
        const float corr = expf(mx - nm), pj = expf(s - nm);
        den = den * corr + pj;
        const float* vj = base + (size_t)j * ld + voff;
        for (int d = 0; d < hd; ++d) acc[d] = acc[d] * corr + pj * vj[d];
        mx = nm;
    }
    float* o = out + ((size_t)b * N + i) * ldo + head * hd;
    for (int d = 0; d < hd; ++d) o[d] = acc[d] / den;
}

// The same arithmetic -- one thread per query, keys visited in order, online softmax, every sum a sequential chain -- with the
// head dimensions as compile-time constants (q and the output accumulator live in registers: attn32_kernel's runtime-indexed
// acc[] sits in scratch memory) and the keys / values staged through LDS 64 rows at a time (all lanes read the same row:
// broadcast reads).  YOLO11's C2PSA always has key_dim 32, head_dim 64 (block.py:4256-4262: head_dim = dim / num_heads with
// num_heads = dim / 64, attn_ratio 0.5).  Same operations in the same order as attn32_kernel: the same bits.
template <int KD, int HD>
__global__ __launch_bounds__(256) void attn32_tiled_kernel(const float* qkv, int ld, int N, int heads, float scale, float* out, int ldo) {
    constexpr int TK = 64;
    __shared__ __attribute__((aligned(16))) float sK[TK][KD];
    __shared__ __attribute__((aligned(16))) float sV[TK][HD];
    const int tid = threadIdx.x, i = blockIdx.x * 256 + tid, head = blockIdx.y, b = blockIdx.z;
    const float* base = qkv + (size_t)b * N * ld;
    const int koff = heads * KD + head * KD, voff = 2 * heads * KD + head * HD;
    const bool live = i < N;
    float q[KD], acc[HD];
    {
        const float* qp = base + (size_t)(live ? i : 0) * ld + head * KD;
#pragma unroll
        for (int c = 0; c < KD; c += 4) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(qp + c);
            q[c] = v[0]; q[c + 1] = v[1]; q[c + 2] = v[2]; q[c + 3] = v[3];
        }
    }
#pragma unroll
    for (int d = 0; d < HD; ++d) acc[d] = 0.f;
    float mx = -INFINITY, den = 0.f;
    for (int j0 = 0; j0 < N; j0 += TK) {
        const int nj = N - j0 < TK ? N - j0 : TK;
        __syncthreads();  // the previous tile has been consumed
        for (int id = tid; id < TK * (KD + HD) / 4; id += 256) {
            const int r = id / ((KD + HD) / 4), c4 = (id % ((KD + HD) / 4)) * 4;
            if (r < nj) {
                const float* rowp = base + (size_t)(j0 + r) * ld;
                if (c4 < KD) *reinterpret_cast<f32x4*>(&sK[r][c4]) = *reinterpret_cast<const f32x4*>(rowp + koff + c4);
                else *reinterpret_cast<f32x4*>(&sV[r][c4 - KD]) = *reinterpret_cast<const f32x4*>(rowp + voff + c4 - KD);
            }
        }
        __syncthreads();
        for (int jj = 0; jj < nj; ++jj) {
            float s = 0.f;
#pragma unroll
            for (int c = 0; c < KD; ++c) s = fmaf(q[c], sK[jj][c], s);
            s *= scale;
            const float nm = fmaxf(mx, s);
            const float corr = expf(mx - nm), pj = expf(s - nm);
            den = den * corr + pj;
#pragma unroll
            for (int d = 0; d < HD; ++d) acc[d] = acc[d] * corr + pj * sV[jj][d];
            mx = nm;
        }
    }
    if (!live) return;
    float* o = out + ((size_t)b * N + i) * ldo + head * HD;
#pragma unroll
    for (int d = 0; d < HD; d += 4) *reinterpret_cast<f32x4*>(o + d) = f32x4{acc[d] / den, acc[d + 1] / den, acc[d + 2] / den, acc[d + 3] / den};
}

__global__ __launch_bounds__(256) void nhwc2nchw32_kernel(const float* src, int ld, int B, int C, int HW, void* out, int out_dtype) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long long)B * C * HW) return;
    const int p = (int)(idx % HW);
    long long t = idx / HW;
    const int c = (int)(t % C);
    const int n = (int)(t / C);
    const float v = src[((size_t)n * HW + p) * ld + c];
    if (out_dtype == BSY_F32) reinterpret_cast<float*>(out)[idx] = v;
    else reinterpret_cast<half_t*>(out)[idx] = (half_t)v;
}

__global__ __launch_bounds__(256) void copy32_kernel(const float* src, int lds_, int up, int B, int H, int W, int C, float* dst, int ldd) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long long)B * H * W * C) return;
    const int c = (int)(idx % C);
    long long t = idx / C;
    const int x = (int)(t % W);
    t /= W;
    const int y = (int)(t % H);
    const int n = (int)(t / H);
    dst[((size_t)(n * H + y) * W + x) * ldd + c] = src[((size_t)(n * (H >> up) + (y >> up)) * (W >> up) + (x >> up)) * lds_ + c];
}

// one workgroup per (image, channel): mean over H*W (MSCAAttention's AdaptiveAvgPool2d(1))
__global__ __launch_bounds__(256) void gap32_kernel(const float* src, int lds_, int HW, float* out, int ldo) {
    __shared__ float red[256];
    const int c = blockIdx.x, n = blockIdx.y, tid = threadIdx.x;
    float a = 0.f;
    for (int p = tid; p < HW; p += 256) a += src[((size_t)n * HW + p) * lds_ + c];
    red[tid] = a;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if (tid < st) red[tid] += red[tid + st];
        __syncthreads();
    }
    if (tid == 0) out[(size_t)n * ldo + c] = red[0] / (float)HW;
}

__global__ __launch_bounds__(256) void mix32_kernel(const Mix32Args a) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long long)a.B * a.HW * a.C) return;
    const int c = (int)(idx % a.C);
    const long long pix = idx / a.C;
    const int n = (int)(pix / a.HW);
    float w[4], den = 0.f;
    for (int i = 0; i < 4; ++i) {  // softmax over the four branches of sigmoid(SE logit) (MSCA.py:69-82)
        w[i] = expf(sigmoid32(a.lg[i][(size_t)n * a.ldl[i] + c]));
        den += w[i];
    }
    float acc = 0.f;
    for (int i = 0; i < 4; ++i) acc = fmaf(w[i] / den, a.br[i][(size_t)pix * a.ldb[i] + c], acc);
    a.dst[(size_t)pix * a.ldd + c] = acc;
}

__global__ __launch_bounds__(256) void mul32_kernel(const float* x, int ldx, const float* y, int ldy, long long npix, int C, float* dst, int ldd) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= npix * C) return;
    const int c = (int)(idx % C);
    const long long pix = idx / C;
    dst[(size_t)pix * ldd + c] = x[(size_t)pix * ldx + c] * y[(size_t)pix * ldy + c];
}

// ELA row / column means into the scratch layout of bsyolo_ops.hip (the gate kernel there is f32 already and is reused)
__global__ __launch_bounds__(256) void ela_stats32_kernel(const float* src, int lds_, int H, int W, int C, float* scratch, size_t per_img) {
    const int n = blockIdx.y, dir = blockIdx.z;
    const int L = dir ? W : H, R = dir ? H : W;
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long long)L * C) return;
    const int c = (int)(idx % C), line = (int)(idx / C);
    float a = 0.f;
    for (int r = 0; r < R; ++r) {
        const int y = dir ? r : line, x = dir ? line : r;
        a += src[((size_t)(n * H + y) * W + x) * lds_ + c];
    }
    scratch[(size_t)n * per_img + (dir ? (size_t)H * C : 0) + (size_t)line * C + c] = a / (float)R;
}

__global__ __launch_bounds__(256) void ela_apply32_kernel(const float* src, int lds_, int B, int H, int W, int C, const float* scratch,
                                                          size_t per_img, float ca, float sb, float rr, float* dst, int ldd) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long long)B * H * W * C) return;
    const int c = (int)(idx % C);
    long long t = idx / C;
    const int x = (int)(t % W);
    t /= W;
    const int y = (int)(t % H);
    const int n = (int)(t / H);
    const float* base = scratch + (size_t)n * per_img + (size_t)(H + W + 1) * C;
    const float hg = base[(size_t)y * C + c], wg = base[(size_t)H * C + (size_t)x * C + c], cg = base[(size_t)(H + W) * C + c];
    const size_t pix = (size_t)(n * H + y) * W + x;
    const float xv = src[pix * lds_ + c];
    dst[pix * ldd + c] = xv * (ca * cg + sb * (hg * wg)) + rr * xv;
}

inline unsigned nblk(long long n) { return (unsigned)((n + 255) / 256); }
}  // namespace

int launch_conv32(const Conv32Args& a, hipStream_t s) {
    // BSY_CONV32_SCALAR=1 (test / A-B aid): every conv on the scalar kernel
    static const bool scalar_only = [] { const char* e = getenv("BSY_CONV32_SCALAR"); return e && atoi(e) != 0; }();
    if (!scalar_only && conv32_mfma_supported(a)) return launch_conv32_mfma(a, s);
    return launch_conv32_scalar(a, s);
}

int launch_conv32_scalar(const Conv32Args& a, hipStream_t s) {
    if (!a.src0 || !a.w || !a.bias || !a.dst || a.Cout <= 0 || a.B <= 0) BSY_FAIL(BSY_ERR_ARG, "conv32: bad argument");
    if (a.C1 && !a.src1) BSY_FAIL(BSY_ERR_ARG, "conv32: src1 missing");
    const long long M = (long long)a.B * a.OH * a.OW;
    const long long threads = (M + 3) / 4 * a.Cout;
    if (threads <= 0 || threads > 0x7fffffffLL * 256) BSY_FAIL(BSY_ERR_ARG, "conv32: extent out of range");
    if (a.first) hipLaunchKernelGGL(conv32_kernel<true>, dim3(nblk(threads)), dim3(256), 0, s, a);
    else hipLaunchKernelGGL(conv32_kernel<false>, dim3(nblk(threads)), dim3(256), 0, s, a);
    HIP_TRY(hipGetLastError());
    return BSY_OK;
}

int launch_dw32(const Dw32Args& a, hipStream_t s) {
    if (!a.src || !a.w || !a.b || !a.dst || a.kh < 1 || a.kw < 1 || (a.stride != 1 && a.stride != 2)) BSY_FAIL(BSY_ERR_ARG, "dw32: bad argument");
    const bool vec4 = !(a.C & 3) && !(a.lds & 3) && !(a.ldd & 3) && !(a.wld & 3) && (!a.res || !(a.ldr & 3)) &&
                      !(((uintptr_t)a.src | (uintptr_t)a.dst | (uintptr_t)a.w | (uintptr_t)a.b | (uintptr_t)a.res) & 15);
    const char* slow_e = getenv("BSY_REF32_SLOW");  // test / A-B aid: the round-3 kernels (read per call: tests flip it)
    const bool slow = slow_e && atoi(slow_e) != 0;
    if (vec4 && a.kh == 3 && a.kw == 3 && a.stride == 1 && a.OH == a.H && a.OW == a.W && !slow)
        hipLaunchKernelGGL(dw32_3x3x4_kernel, dim3(nblk((long long)a.B * a.OH * ((a.OW + 3) / 4) * (a.C / 4))), dim3(256), 0, s, a);
    else if (vec4) hipLaunchKernelGGL(dw32x4_kernel, dim3(nblk((long long)a.B * a.OH * a.OW * (a.C / 4))), dim3(256), 0, s, a);
    else hipLaunchKernelGGL(dw32_kernel, dim3(nblk((long long)a.B * a.OH * a.OW * a.C)), dim3(256), 0, s, a);
    HIP_TRY(hipGetLastError());
    return BSY_OK;
}

int launch_sppf32(float* buf, int ld, int B, int H, int W, int C, hipStream_t s) {
    if (!buf || ld < 4 * C) BSY_FAIL(BSY_ERR_ARG, "sppf32: bad argument");
    const char* slow_e = getenv("BSY_REF32_SLOW");
    const bool slow = slow_e && atoi(slow_e) != 0;
    const bool vec4 = !(C & 3) && !(ld & 3) && !((uintptr_t)buf & 15);
    const long long hw = (long long)H * W;
    // the LDS form: the map of 4 V channels of one image twice in LDS (<= 64 KiB), V as large as fits and divides C / 4
    int V = 0;
    for (int v = 4; v >= 1 && !V; v >>= 1)
        if (hw * v * 32 <= 65536 && (C / 4) % v == 0) V = v;
    if (vec4 && V && !slow && B <= 65535) {
        const size_t smem = (size_t)hw * V * 32;
        const dim3 grid(C / (4 * V), B);
        if (V == 4) hipLaunchKernelGGL((sppf32_lds_kernel<4>), grid, dim3(256), smem, s, buf, ld, H, W, C);
        else if (V == 2) hipLaunchKernelGGL((sppf32_lds_kernel<2>), grid, dim3(256), smem, s, buf, ld, H, W, C);
        else hipLaunchKernelGGL((sppf32_lds_kernel<1>), grid, dim3(256), smem, s, buf, ld, H, W, C);
    } else if (vec4)
        hipLaunchKernelGGL(sppf32x4_kernel, dim3(nblk((long long)B * H * W * (C / 4))), dim3(256), 0, s, buf, ld, B, H, W, C);
    else
        hipLaunchKernelGGL(sppf32_kernel, dim3(nblk((long long)B * H * W * C)), dim3(256), 0, s, buf, ld, B, H, W, C);
    HIP_TRY(hipGetLastError());
    return BSY_OK;
}

int launch_attn32(const float* qkv, int ld, int B, int N, int heads, int kd, int hd, float scale, float* out, int ldo, hipStream_t s, int impl) {
    if (!qkv || !out || hd > ATT32_MAXD || kd < 1 || hd < 1 || heads < 1) BSY_FAIL(BSY_ERR_ARG, "attn32: bad argument (head_dim <= %d)", ATT32_MAXD);
    static const bool generic_only = [] { const char* e = getenv("BSY_ATTN32_GENERIC"); return e && atoi(e) != 0; }();  // test / A-B aid
    if (impl == 2 && !(kd == 32 && hd == 64 && !(ld & 3) && !(ldo & 3) && !(((uintptr_t)qkv | (uintptr_t)out) & 15)))
        BSY_FAIL(BSY_ERR_ARG, "attn32: the tiled kernel takes key_dim 32, head_dim 64, 16-byte aligned views");
    if (impl != 1 && kd == 32 && hd == 64 && !generic_only && !(ld & 3) && !(ldo & 3) && !(((uintptr_t)qkv | (uintptr_t)out) & 15))
        hipLaunchKernelGGL((attn32_tiled_kernel<32, 64>), dim3((N + 255) / 256, heads, B), dim3(256), 0, s, qkv, ld, N, heads, scale, out, ldo);
    else
        hipLaunchKernelGGL(attn32_kernel, dim3((N + 63) / 64, heads, B), dim3(64), 0, s, qkv, ld, B, N, heads, kd, hd, scale, out, ldo);
    HIP_TRY(hipGetLastError());
    return BSY_OK;
}

int launch_nhwc2nchw32(const float* src, int ld, int B, int C, int hw, void* out, int out_dtype, hipStream_t s) {
    if (!src || !out) BSY_FAIL(BSY_ERR_ARG, "nhwc2nchw32: null pointer");
    hipLaunchKernelGGL(nhwc2nchw32_kernel, dim3(nblk((long long)B * C * hw)), dim3(256), 0, s, src, ld, B, C, hw, out, out_dtype);
    HIP_TRY(hipGetLastError());
    return BSY_OK;
}

int launch_copy32(const float* src, int lds_, int up, int B, int H, int W, int C, float* dst, int ldd, hipStream_t s) {
    if (!src || !dst) BSY_FAIL(BSY_ERR_ARG, "copy32: null pointer");
    hipLaunchKernelGGL(copy32_kernel, dim3(nblk((long long)B * H * W * C)), dim3(256), 0, s, src, lds_, up, B, H, W, C, dst, ldd);
    HIP_TRY(hipGetLastError());
    return BSY_OK;
}

int launch_gap32(const float* src, int lds_, int B, int H, int W, int C, float* out, int ldo, hipStream_t s) {
    if (!src || !out) BSY_FAIL(BSY_ERR_ARG, "gap32: null pointer");
    hipLaunchKernelGGL(gap32_kernel, dim3(C, B), dim3(256), 0, s, src, lds_, H * W, out, ldo);
    HIP_TRY(hipGetLastError());
    return BSY_OK;
}

int launch_mix32(const Mix32Args& a, hipStream_t s) {
    for (int i = 0; i < 4; ++i)
        if (!a.br[i] || !a.lg[i]) BSY_FAIL(BSY_ERR_ARG, "mix32: branch %d missing", i);
    hipLaunchKernelGGL(mix32_kernel, dim3(nblk((long long)a.B * a.HW * a.C)), dim3(256), 0, s, a);
    HIP_TRY(hipGetLastError());
    return BSY_OK;
}

int launch_mul32(const float* x, int ldx, const float* y, int ldy, long long npix, int C, float* dst, int ldd, hipStream_t s) {
    if (!x || !y || !dst) BSY_FAIL(BSY_ERR_ARG, "mul32: null pointer");
    hipLaunchKernelGGL(mul32_kernel, dim3(nblk(npix * C)), dim3(256), 0, s, x, ldx, y, ldy, npix, C, dst, ldd);
    HIP_TRY(hipGetLastError());
    return BSY_OK;
}

int launch_ela32(const ElaArgs& a, const float* src, float* dst, hipStream_t s) {
    if (!src || !dst || !a.scratch || !a.wsp || !a.wch || !a.gnw || !a.gnb || (a.C & 7) || a.k < 1 || !(a.k & 1) || a.k > 15)
        BSY_FAIL(BSY_ERR_ARG, "ela32: bad argument");
    const size_t per_img = ela_scratch_floats(a.H, a.W, a.C);
    const int Lmax = a.H > a.W ? a.H : a.W;
    hipLaunchKernelGGL(ela_stats32_kernel, dim3(nblk((long long)Lmax * a.C), a.B, 2), dim3(256), 0, s, src, a.lds, a.H, a.W, a.C, a.scratch, per_img);
    HIP_TRY(hipGetLastError());
    const int rc = launch_ela_gate(a, s);
    if (rc != BSY_OK) return rc;
    hipLaunchKernelGGL(ela_apply32_kernel, dim3(nblk((long long)a.B * a.H * a.W * a.C)), dim3(256), 0, s, src, a.lds, a.B, a.H, a.W, a.C,
                       a.scratch, per_img, a.ch_coef, a.sp_coef, a.res_coef, dst, a.ldd);
    HIP_TRY(hipGetLastError());
    return BSY_OK;
}
