// Batched non_max_suppression for gfx950, no host synchronisation.  Compile with -ffp-contract=off: the IoU test must
// round exactly like the fp32 CPU reference (inter / (area_i + area_j - inter) > thr, no FMA contraction).
//
// Replaces utils/ops.py:167-316 (non_max_suppression), :416-433 (xywh2xyxy) and the greedy suppression the reference
// delegates to torchvision.ops.nms (:296).  Three kernels per call, whole batch each:
//   1. nms_filter_kernel  (thread = anchor): xywh->xyxy in the prediction dtype (optionally written back, as the
//      reference mutates its input :243-244); confidence filter -- best class (:273-275) or multi-label expansion
//      (:270-272) --; optional class filter (:278-279); survivors are appended as 64-bit keys
//      (fp32 score bits << 32 | ~candidate_index): ONE descending sort gives "score descending, ties by ascending
//      candidate index", which is the order torchvision's sort + the reference's max_nms argsort produce.
//   2. nms_sort_kernel    (block = image): bitonic sort of the keys (LDS up to 8192 keys, else in global memory).
//   3. nms_greedy_kernel  (block = image): walks the sorted candidates 256 at a time.  Each candidate is first
//      tested against the boxes already kept (LDS), then the survivors of the chunk are resolved among themselves
//      by a parallel fixed point on the chunk's 256x256 overlap bitmask (keep = every overlapping earlier candidate
//      is suppressed; suppressed = some overlapping earlier candidate is kept) -- identical to the sequential greedy
//      order, without 256 serial steps.  Stops at max_det kept (:297 `i[:max_det]`).
// The class offset `boxes + cls * max_wh` (:289-295) is applied in fp32 exactly as on the CPU.  Input fp16 is
// upcast to fp32 (the fp16 reference path overflows 7680*cls; we follow the fp32 CPU reference instead).
#include "common.h"

typedef unsigned long long u64;

struct NmsK {
    void* pred;
    int B, nc, nm, A, C;  // C = 4+nc+nm
    float conf, iou;
    const int32_t* classes;
    int n_classes, agnostic, multi_label, max_det, max_nms;
    float max_wh;
    int in_place;
    int cap;  // keys per image
    u64* keys;
    int32_t* cand_count;
    float* out;
    int32_t* counts;
};

template <typename T>
__device__ __forceinline__ void load_xyxy(const T* pb, int A, int a, bool is_xyxy, float* o) {
    const float v0 = (float)pb[a], v1 = (float)pb[(size_t)A + a], v2 = (float)pb[(size_t)2 * A + a],
                v3 = (float)pb[(size_t)3 * A + a];
    if (is_xyxy) {
        o[0] = v0; o[1] = v1; o[2] = v2; o[3] = v3;
    } else {  // xywh2xyxy in the tensor's own precision (ops.py:428-432)
        const float hw = (float)(T)(v2 / 2.0f), hh = (float)(T)(v3 / 2.0f);
        o[0] = (float)(T)(v0 - hw); o[1] = (float)(T)(v1 - hh); o[2] = (float)(T)(v0 + hw); o[3] = (float)(T)(v1 + hh);
    }
}

template <typename T>
__global__ __launch_bounds__(256) void nms_filter_kernel(const NmsK p) {
    const int a = blockIdx.x * blockDim.x + threadIdx.x;
    const int b = blockIdx.y;
    if (a >= p.A) return;
    T* pb = reinterpret_cast<T*>(p.pred) + (size_t)b * p.C * p.A;
    if (p.in_place) {
        float bx[4];
        load_xyxy<T>(pb, p.A, a, false, bx);
        pb[a] = (T)bx[0]; pb[(size_t)p.A + a] = (T)bx[1]; pb[(size_t)2 * p.A + a] = (T)bx[2]; pb[(size_t)3 * p.A + a] = (T)bx[3];
    }
    const T* cp = pb + (size_t)4 * p.A + a;
    u64* keys = p.keys + (size_t)b * p.cap;
    if (p.multi_label) {
        for (int c = 0; c < p.nc; ++c) {
            const float s = (float)cp[(size_t)c * p.A];
            if (!(s > p.conf)) continue;
            bool ok = p.n_classes == 0;
            for (int k = 0; k < p.n_classes; ++k) ok |= (p.classes[k] == c);
            if (!ok) continue;
            const int pos = atomicAdd(p.cand_count + b, 1);
            const unsigned idx = (unsigned)a * (unsigned)p.nc + (unsigned)c;
            keys[pos] = ((u64)__float_as_uint(s) << 32) | (u64)(~idx);
        }
    } else {
        float best = (float)cp[0];
        int bc = 0;
        for (int c = 1; c < p.nc; ++c) {
            const float s = (float)cp[(size_t)c * p.A];
            if (s > best) { best = s; bc = c; }
        }
        if (!(best > p.conf)) return;
        bool ok = p.n_classes == 0;
        for (int k = 0; k < p.n_classes; ++k) ok |= (p.classes[k] == bc);
        if (!ok) return;
        const int pos = atomicAdd(p.cand_count + b, 1);
        keys[pos] = ((u64)__float_as_uint(best) << 32) | (u64)(~(unsigned)a);
    }
}

// Best-class filter, 4 anchors per thread (fp16 predictions, A % 4 == 0): every class row is
// read as 8-byte pieces instead of 2-byte ones -- the one-anchor form issues 80 two-byte loads per lane and ran at 1.4 TB/s.
// Same keys as nms_filter_kernel (their order in the list differs, the sort that follows removes that).
__global__ __launch_bounds__(256) void nms_filter4_kernel(const NmsK p) {
    const int a0 = (blockIdx.x * blockDim.x + threadIdx.x) * 4;
    const int b = blockIdx.y;
    if (a0 >= p.A) return;
    half_t* pb = reinterpret_cast<half_t*>(p.pred) + (size_t)b * p.C * p.A + a0;
    if (p.in_place) {  // xywh -> xyxy in the tensor's own precision, written back (ops.py:243-244, :428-432)
        const half4 x = *reinterpret_cast<const half4*>(pb), y = *reinterpret_cast<const half4*>(pb + (size_t)p.A),
                    w = *reinterpret_cast<const half4*>(pb + (size_t)2 * p.A), h = *reinterpret_cast<const half4*>(pb + (size_t)3 * p.A);
        half4 o0, o1, o2, o3;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float hw = (float)(half_t)((float)w[j] / 2.0f), hh = (float)(half_t)((float)h[j] / 2.0f);
            o0[j] = (half_t)((float)x[j] - hw); o1[j] = (half_t)((float)y[j] - hh);
            o2[j] = (half_t)((float)x[j] + hw); o3[j] = (half_t)((float)y[j] + hh);
        }
        *reinterpret_cast<half4*>(pb) = o0; *reinterpret_cast<half4*>(pb + (size_t)p.A) = o1;
        *reinterpret_cast<half4*>(pb + (size_t)2 * p.A) = o2; *reinterpret_cast<half4*>(pb + (size_t)3 * p.A) = o3;
    }
    const half_t* cp = pb + (size_t)4 * p.A;
    half4 best = *reinterpret_cast<const half4*>(cp);
    int bc[4] = {0, 0, 0, 0};
    for (int c = 1; c < p.nc; ++c) {
        const half4 v = *reinterpret_cast<const half4*>(cp + (size_t)c * p.A);
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if ((float)v[j] > (float)best[j]) { best[j] = v[j]; bc[j] = c; }
    }
    u64* keys = p.keys + (size_t)b * p.cap;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float s = (float)best[j];
        if (!(s > p.conf)) continue;
        bool ok = p.n_classes == 0;
        for (int k = 0; k < p.n_classes; ++k) ok |= (p.classes[k] == bc[j]);
        if (!ok) continue;
        const int pos = atomicAdd(p.cand_count + b, 1);
        keys[pos] = ((u64)__float_as_uint(s) << 32) | (u64)(~(unsigned)(a0 + j));
    }
}

#define SORT_LDS_KEYS 8192
__global__ __launch_bounds__(1024) void nms_sort_kernel(const NmsK p) {
    __shared__ u64 sk[SORT_LDS_KEYS];
    const int b = blockIdx.x, tid = threadIdx.x, nt = blockDim.x;
    const int n = p.cand_count[b];
    if (n <= 1) return;
    int np2 = 2;
    while (np2 < n) np2 <<= 1;
    u64* keys = p.keys + (size_t)b * p.cap;
    if (np2 <= SORT_LDS_KEYS) {
        for (int i = tid; i < np2; i += nt) sk[i] = i < n ? keys[i] : 0ull;
        __syncthreads();
        for (int k = 2; k <= np2; k <<= 1)
            for (int j = k >> 1; j > 0; j >>= 1) {
                for (int i = tid; i < np2; i += nt) {
                    const int l = i ^ j;
                    if (l > i) {
                        const u64 x = sk[i], y = sk[l];
                        const bool desc = (i & k) == 0;  // descending overall
                        if (desc ? (x < y) : (x > y)) { sk[i] = y; sk[l] = x; }
                    }
                }
                __syncthreads();
            }
        for (int i = tid; i < n; i += nt) keys[i] = sk[i];
    } else {
        for (int i = n + tid; i < np2; i += nt) keys[i] = 0ull;  // cap is a power of two >= np2
        __syncthreads();
        for (int k = 2; k <= np2; k <<= 1)
            for (int j = k >> 1; j > 0; j >>= 1) {
                for (int i = tid; i < np2; i += nt) {
                    const int l = i ^ j;
                    if (l > i) {
                        const u64 x = keys[i], y = keys[l];
                        const bool desc = (i & k) == 0;
                        if (desc ? (x < y) : (x > y)) { keys[i] = y; keys[l] = x; }
                    }
                }
                __syncthreads();
            }
    }
}

__device__ __forceinline__ bool iou_gt(const float* bi, float ai, const float* bj, float aj, float thr) {
    const float xx1 = fmaxf(bi[0], bj[0]), yy1 = fmaxf(bi[1], bj[1]);
    const float xx2 = fminf(bi[2], bj[2]), yy2 = fminf(bi[3], bj[3]);
    const float w = fmaxf(0.0f, xx2 - xx1), h = fmaxf(0.0f, yy2 - yy1);
    const float inter = w * h;
    const float ovr = inter / (ai + aj - inter);
    return ovr > thr;
}

template <typename T>
__global__ __launch_bounds__(256) void nms_greedy_kernel(const NmsK p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    // layout: 64-bit words first, then ints, then floats (the host sizes it in bsy_nms)
    u64* keepw = reinterpret_cast<u64*>(smem_raw);       // [4]
    u64* supw = keepw + 4;                               // [4]
    u64* alw = supw + 4;                                 // [4]
    int* sh_i = reinterpret_cast<int*>(alw + 4);         // [4]: [0] = undecided count, [1] = n_kept
    float* cbox = reinterpret_cast<float*>(sh_i + 4);    // [256][4]
    float* carea = cbox + 256 * 4;                       // [256]
    float* kbox = carea + 256;                           // [max_det][4]  kept boxes (offset by class)
    float* karea = kbox + (size_t)p.max_det * 4;         // [max_det]

    const int b = blockIdx.x, tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const T* pb = reinterpret_cast<const T*>(p.pred) + (size_t)b * p.C * p.A;
    const u64* keys = p.keys + (size_t)b * p.cap;
    int n = p.cand_count[b];
    if (n > p.max_nms) n = p.max_nms;
    const int row = 6 + p.nm;
    float* ob = p.out + (size_t)b * p.max_det * row;
    if (tid == 0) sh_i[1] = 0;
    __syncthreads();

    for (int c0 = 0; c0 < n; c0 += 256) {
        const int n_kept = sh_i[1];
        if (n_kept >= p.max_det) break;
        const int i = c0 + tid;
        const bool have = i < n;
        float bx[4] = {0.f, 0.f, 0.f, 0.f}, obx[4] = {0.f, 0.f, 0.f, 0.f};
        float score = 0.f, area = 0.f;
        int cls = 0, anchor = 0;
        if (have) {
            const u64 k = keys[i];
            score = __uint_as_float((unsigned)(k >> 32));
            const unsigned idx = ~(unsigned)(k & 0xffffffffull);
            if (p.multi_label) { anchor = (int)(idx / (unsigned)p.nc); cls = (int)(idx - (unsigned)anchor * p.nc); }
            else {
                anchor = (int)idx;
                // best class = first maximum (ops.py:274 cls.max(1))
                const T* cp = pb + (size_t)4 * p.A + anchor;
                float best = (float)cp[0];
                for (int c = 1; c < p.nc; ++c) {
                    const float s = (float)cp[(size_t)c * p.A];
                    if (s > best) { best = s; cls = c; }
                }
            }
            load_xyxy<T>(pb, p.A, anchor, p.in_place != 0, bx);
            const float off = (float)cls * (p.agnostic ? 0.0f : p.max_wh);
#pragma unroll
            for (int e = 0; e < 4; ++e) obx[e] = bx[e] + off;
            area = (obx[2] - obx[0]) * (obx[3] - obx[1]);
        }
        // phase A: against everything kept so far
        bool alive = have;
        for (int j = 0; j < n_kept && alive; ++j)
            if (iou_gt(kbox + 4 * j, karea[j], obx, area, p.iou)) alive = false;
        // stage the chunk
#pragma unroll
        for (int e = 0; e < 4; ++e) cbox[4 * tid + e] = obx[e];
        carea[tid] = area;
        if (tid < 4) { keepw[tid] = 0ull; supw[tid] = 0ull; }
        __syncthreads();
        // overlap bitmask of this candidate with EARLIER candidates of the chunk that survived phase A
        const u64 aliveb = __ballot(alive);
        if (lane == 0) alw[wave] = aliveb;
        __syncthreads();
        u64 alivew[4];
#pragma unroll
        for (int w = 0; w < 4; ++w) alivew[w] = alw[w];
        u64 mrow[4] = {0ull, 0ull, 0ull, 0ull};
        if (alive) {
            for (int j = 0; j < tid; ++j) {
                if (!((alivew[j >> 6] >> (j & 63)) & 1ull)) continue;
                if (iou_gt(cbox + 4 * j, carea[j], obx, area, p.iou)) mrow[j >> 6] |= 1ull << (j & 63);
            }
        }
        // fixed point: state per candidate 0 = undecided, 1 = keep, 2 = suppressed
        int state = alive ? 0 : 2;
        if (!alive) atomicOr(&supw[wave], 1ull << lane);
        for (int it = 0; it < 257; ++it) {
            if (tid == 0) sh_i[0] = 0;
            __syncthreads();
            u64 kw[4], sw[4];
#pragma unroll
            for (int w = 0; w < 4; ++w) { kw[w] = keepw[w]; sw[w] = supw[w]; }
            int ns = state;
            if (state == 0) {
                bool hit = false, pending = false;
#pragma unroll
                for (int w = 0; w < 4; ++w) {
                    hit |= (mrow[w] & kw[w]) != 0ull;
                    pending |= (mrow[w] & ~(kw[w] | sw[w])) != 0ull;
                }
                if (hit) ns = 2;
                else if (!pending) ns = 1;
            }
            __syncthreads();  // everyone has read keepw/supw
            if (ns != state) {
                if (ns == 1) atomicOr(&keepw[wave], 1ull << lane);
                else atomicOr(&supw[wave], 1ull << lane);
                state = ns;
            }
            if (state == 0) atomicAdd(&sh_i[0], 1);
            __syncthreads();
            if (sh_i[0] == 0) break;
            __syncthreads();
        }
        // append kept candidates in order
        u64 kw[4];
#pragma unroll
        for (int w = 0; w < 4; ++w) kw[w] = keepw[w];
        int before = 0;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            if (w < wave) before += __popcll(kw[w]);
        }
        before += __popcll(kw[wave] & ((1ull << lane) - 1ull));
        const int slot = n_kept + before;
        if (state == 1 && slot < p.max_det) {
#pragma unroll
            for (int e = 0; e < 4; ++e) kbox[4 * slot + e] = obx[e];
            karea[slot] = area;
            float* o = ob + (size_t)slot * row;
            o[0] = bx[0]; o[1] = bx[1]; o[2] = bx[2]; o[3] = bx[3];
            o[4] = score;
            o[5] = (float)cls;
            for (int k = 0; k < p.nm; ++k) o[6 + k] = (float)pb[(size_t)(4 + p.nc + k) * p.A + anchor];
        }
        __syncthreads();
        if (tid == 0) {
            const int tot = n_kept + __popcll(kw[0]) + __popcll(kw[1]) + __popcll(kw[2]) + __popcll(kw[3]);
            sh_i[1] = tot < p.max_det ? tot : p.max_det;
        }
        __syncthreads();
    }
    if (tid == 0) p.counts[b] = sh_i[1];
}

static int next_pow2(long long v) {
    long long p = 2;
    while (p < v) p <<= 1;
    return (int)p;
}

extern "C" size_t bsy_nms_workspace_bytes(int B, int A, int nc, int multi_label, int max_nms) {
    (void)max_nms;
    if (B <= 0 || A <= 0 || nc <= 0) return 0;
    const long long per = (multi_label && nc > 1) ? (long long)A * nc : (long long)A;
    const size_t cap = (size_t)next_pow2(per);
    return 256 + (size_t)B * 4 + (size_t)B * cap * 8 + 256;
}

extern "C" int bsy_nms(void* pred, int pred_dtype, int B, int nc, int nm, int A, float conf_thres, float iou_thres,
                       const int32_t* classes, int n_classes, int agnostic, int multi_label, int max_det, int max_nms,
                       float max_wh, int in_place, float* out, int32_t* counts, void* workspace, size_t workspace_bytes,
                       bsy_stream stream) {
    hipStream_t s = (hipStream_t)stream;
    if (!pred || !out || !counts || !workspace) BSY_FAIL(BSY_ERR_ARG, "nms: null pointer");
    if (B <= 0 || nc <= 0 || nm < 0 || A <= 0) BSY_FAIL(BSY_ERR_ARG, "nms: bad sizes B=%d nc=%d nm=%d A=%d", B, nc, nm, A);
    if (!(conf_thres >= 0.f && conf_thres <= 1.f))
        BSY_FAIL(BSY_ERR_ARG, "Invalid Confidence threshold %g, valid values are between 0.0 and 1.0", conf_thres);
    if (!(iou_thres >= 0.f && iou_thres <= 1.f))
        BSY_FAIL(BSY_ERR_ARG, "Invalid IoU %g, valid values are between 0.0 and 1.0", iou_thres);
    if (max_det <= 0 || max_det > 4096) BSY_FAIL(BSY_ERR_ARG, "nms: max_det %d out of range (1..4096)", max_det);
    if (max_nms <= 0) BSY_FAIL(BSY_ERR_ARG, "nms: max_nms must be positive");
    if (pred_dtype != BSY_F16 && pred_dtype != BSY_F32) BSY_FAIL(BSY_ERR_ARG, "nms: dtype %d unsupported", pred_dtype);
    if ((long long)A * nc > 0x7fffffffLL) BSY_FAIL(BSY_ERR_ARG, "nms: A*nc too large");
    multi_label = (multi_label && nc > 1) ? 1 : 0;  // ops.py:239
    const size_t need = bsy_nms_workspace_bytes(B, A, nc, multi_label, max_nms);
    if (workspace_bytes < need) BSY_FAIL(BSY_ERR_ARG, "nms: workspace %zu < %zu bytes", workspace_bytes, need);
    NmsK k;
    k.pred = pred; k.B = B; k.nc = nc; k.nm = nm; k.A = A; k.C = 4 + nc + nm;
    k.conf = conf_thres; k.iou = iou_thres; k.classes = classes; k.n_classes = classes ? n_classes : 0;
    k.agnostic = agnostic; k.multi_label = multi_label; k.max_det = max_det; k.max_nms = max_nms; k.max_wh = max_wh;
    k.in_place = in_place;
    k.cap = next_pow2(multi_label ? (long long)A * nc : (long long)A);
    uintptr_t w = ((uintptr_t)workspace + 255) & ~(uintptr_t)255;
    k.cand_count = (int32_t*)w;
    k.keys = (u64*)(((w + (size_t)B * 4) + 255) & ~(uintptr_t)255);
    k.out = out; k.counts = counts;
    HIP_TRY(hipMemsetAsync(k.cand_count, 0, (size_t)B * 4, s));
    HIP_TRY(hipMemsetAsync(out, 0, (size_t)B * max_det * (6 + nm) * sizeof(float), s));
    dim3 g1((A + 255) / 256, B);
    const size_t lds = 12 * 8 + 4 * 4 + ((size_t)256 * 5 + (size_t)max_det * 5) * 4;
    if (pred_dtype == BSY_F16) {
        if (!multi_label && !(A & 3) && !((uintptr_t)pred & 7))
            hipLaunchKernelGGL(nms_filter4_kernel, dim3((A / 4 + 255) / 256, B), dim3(256), 0, s, k);
        else
            hipLaunchKernelGGL(nms_filter_kernel<half_t>, g1, dim3(256), 0, s, k);
        hipLaunchKernelGGL(nms_sort_kernel, dim3(B), dim3(1024), 0, s, k);
        hipLaunchKernelGGL(nms_greedy_kernel<half_t>, dim3(B), dim3(256), lds, s, k);
    } else {
        hipLaunchKernelGGL(nms_filter_kernel<float>, g1, dim3(256), 0, s, k);
        hipLaunchKernelGGL(nms_sort_kernel, dim3(B), dim3(1024), 0, s, k);
        hipLaunchKernelGGL(nms_greedy_kernel<float>, dim3(B), dim3(256), lds, s, k);
    }
    HIP_TRY(hipGetLastError());
    return BSY_OK;
}

__global__ void scale_boxes_kernel(float* det, const int32_t* counts, int B, int max_det, int row, const float* gain,
                                   const float* pad_x, const float* pad_y, const float* h0, const float* w0) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int b = blockIdx.y;
    if (i >= max_det || i >= counts[b]) return;
    float* o = det + ((size_t)b * max_det + i) * row;
    const float g = gain[b], px = pad_x[b], py = pad_y[b], hh = h0[b], ww = w0[b];
    // ops.py:116-127: subtract pad, divide by gain, clip to the original image
    float x1 = (o[0] - px) / g, y1 = (o[1] - py) / g, x2 = (o[2] - px) / g, y2 = (o[3] - py) / g;
    o[0] = fminf(fmaxf(x1, 0.f), ww);
    o[1] = fminf(fmaxf(y1, 0.f), hh);
    o[2] = fminf(fmaxf(x2, 0.f), ww);
    o[3] = fminf(fmaxf(y2, 0.f), hh);
}

extern "C" int bsy_scale_boxes(float* det, const int32_t* counts, int B, int max_det, int row, const float* gain,
                               const float* pad_x, const float* pad_y, const float* h0, const float* w0,
                               bsy_stream stream) {
    if (!det || !counts || !gain || !pad_x || !pad_y || !h0 || !w0 || B <= 0 || max_det <= 0 || row < 4)
        BSY_FAIL(BSY_ERR_ARG, "scale_boxes: bad argument");
    dim3 grid((max_det + 127) / 128, B);
    hipLaunchKernelGGL(scale_boxes_kernel, grid, dim3(128), 0, (hipStream_t)stream, det, counts, B, max_det, row, gain,
                       pad_x, pad_y, h0, w0);
    HIP_TRY(hipGetLastError());
    return BSY_OK;
}
