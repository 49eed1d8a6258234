// Sliced inference ("SAHI") on the device for gfx950: tile extraction and the cross-tile merge of per-tile detections.
// Compile with -ffp-contract=off (the match tests must round like the CPU restatement, oracle/sahi_ref.py).
//
// The reference reaches this path through the un-vendored `sahi` package (detect-sahi.py:1-13 -> sahi.predict.predict;
// examples/YOLOv8-SAHI-Inference-Video/yolov8_sahi.py:70-75 -> get_sliced_prediction); what is implemented is the
// published algorithm of sahi 0.11.x as restated in oracle/sahi_ref.py (parity unpinned: nothing in the reference
// fixes it).
//
//   bsy_slice_tiles   : one (H, W, 3) u8 image -> (T, 3, th, tw) planes scaled by 1/255, channel order optionally
//                       swapped.  A thread makes 4 pixels of one row: 12 source bytes through four aligned dword
//                       loads + a funnel shift, 8-byte stores into each plane.  HBM-bound.
//   bsy_sahi_merge    : (T, max_det, row) NMS outputs + per-tile shifts -> merged full-image detections:
//     1. sahi_gather_kernel  clamp / validity test / shift per detection; survivors appended as 64-bit keys
//                            (~class : score bits : ~flat index) -> ONE descending sort gives class ascending, score
//                            descending, ties by ascending flat index.
//     2. sahi_sort_kernel    bitonic sort (one workgroup; LDS up to 8192 keys, global memory above).
//     3. sahi_greedy_kernel  sahi's greedy_nmm keep/merge assignment, 256 candidates at a time: against the boxes kept
//                            so far (first kept box of the same class with metric >= thr becomes the keeper), then
//                            inside the chunk by the same parallel fixed point as nms_greedy_kernel.
//     4. sahi_merge_kernel   one wavefront per kept box walks its members in score order and applies has_match
//                            (float64, metric > thr against the GROWING box) + box union / max score.
//     3'. sahi_nmm_kernel   (postprocess NMM) sahi's non-greedy `nmm` assignment: the predictions of a class take turns in score
//                            order (one workgroup, one barrier per turn); a turn tests the prediction against every other
//                            box of the class in parallel and hands its unassigned matches to its keeper (itself if it has
//                            none).  Members remember the turn that assigned them: the merge walks them in (turn, ascending
//                            score) order -- the order sahi's lists are appended in.
// No host synchronisation anywhere; `out_count` stays on the device.
#include "common.h"

typedef unsigned long long u64;

// --------------------------------------------------------------------------------------------------------------------
// tiles
// --------------------------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void slice_tiles_kernel(const uint8_t* __restrict__ img, unsigned img_bytes, int pitch,
                                                          const int32_t* __restrict__ boxes, int th, int tw, int swap_rb,
                                                          T* __restrict__ out) {
    const int x4 = (blockIdx.x * blockDim.x + threadIdx.x) * 4;
    const int y = blockIdx.y, t = blockIdx.z;
    if (x4 >= tw) return;
    const int x0 = boxes[4 * t], y0 = boxes[4 * t + 1];
    const unsigned a = (unsigned)(y0 + y) * (unsigned)pitch + (unsigned)(x0 + x4) * 3u;
    const unsigned a4 = a & ~3u, sh = a & 3u;
    unsigned w[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const unsigned off = a4 + 4 * i;
        if (off + 4 <= img_bytes) {
            w[i] = *reinterpret_cast<const unsigned*>(img + off);
        } else {  // the image's last, partial dword (H * pitch need not be a multiple of 4): never read past the end
            w[i] = 0;
            for (unsigned e = 0; e < 4 && off + e < img_bytes; ++e) w[i] |= (unsigned)img[off + e] << (8 * e);
        }
    }
    unsigned d[3];
#pragma unroll
    for (int i = 0; i < 3; ++i)
        d[i] = sh == 0 ? w[i] : (unsigned)((((u64)w[i + 1] << 32) | (u64)w[i]) >> (8 * sh));
    uint8_t px[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) px[i] = (uint8_t)(d[i >> 2] >> (8 * (i & 3)));
    const size_t plane = (size_t)th * tw;
    T* op = out + (size_t)t * 3 * plane + (size_t)y * tw + x4;
    const T k255 = (T)255.0f;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const int sc = swap_rb ? 2 - c : c;
        T v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = (T)((float)(T)(float)px[3 * e + sc] / (float)k255);  // `im /= 255` in T
        if (x4 + 4 <= tw) {
            if (sizeof(T) == 2) *reinterpret_cast<uint2*>(op + c * plane) = *reinterpret_cast<const uint2*>(v);
            else *reinterpret_cast<uint4*>(op + c * plane) = *reinterpret_cast<const uint4*>(v);
        } else {
            for (int e = 0; e < 4 && x4 + e < tw; ++e) op[c * plane + e] = v[e];
        }
    }
}

extern "C" int bsy_slice_tiles(const uint8_t* img, int H, int W, int pitch, const int32_t* boxes, int T, int th, int tw,
                               int swap_rb, void* out, int out_dtype, bsy_stream stream) {
    if (!img || !boxes || !out || H <= 0 || W <= 0 || T < 0 || th <= 0 || tw <= 0 || pitch < 3 * W)
        BSY_FAIL(BSY_ERR_ARG, "bsy_slice_tiles: bad argument");
    if (th > H || tw > W) BSY_FAIL(BSY_ERR_ARG, "bsy_slice_tiles: slice larger than the image");
    if (((uintptr_t)img & 3) || (tw & 3) || ((uintptr_t)out & 15))
        BSY_FAIL(BSY_ERR_ARG, "bsy_slice_tiles: needs a 4-byte aligned image, tw % 4 == 0");
    if ((unsigned long long)H * pitch >= 0xFFFFFFF0ull) BSY_FAIL(BSY_ERR_ARG, "bsy_slice_tiles: image >= 4 GiB");
    if (out_dtype != BSY_F16 && out_dtype != BSY_F32) BSY_FAIL(BSY_ERR_ARG, "bsy_slice_tiles: out dtype");
    if (T == 0) return BSY_OK;
    const dim3 grid((tw / 4 + 255) / 256, th, T), block(256);
    const unsigned bytes = (unsigned)((size_t)H * pitch);
    hipStream_t s = (hipStream_t)stream;
    if (out_dtype == BSY_F16)
        hipLaunchKernelGGL(slice_tiles_kernel<half_t>, grid, block, 0, s, img, bytes, pitch, boxes, th, tw, swap_rb, (half_t*)out);
    else
        hipLaunchKernelGGL(slice_tiles_kernel<float>, grid, block, 0, s, img, bytes, pitch, boxes, th, tw, swap_rb, (float*)out);
    HIP_TRY(hipGetLastError());
    return BSY_OK;
}

// --------------------------------------------------------------------------------------------------------------------
// merge
// --------------------------------------------------------------------------------------------------------------------
struct SahiK {
    const float* det;
    const int32_t* counts;
    const float* shift;  // (T, 2) = ox, oy
    int T, max_det, row;
    int metric;  // 0 IOU, 1 IOS
    float thr;
    int agnostic, do_merge;
    float full_w, full_h;  // <= 0: no clamp to the full image
    int cap;               // power of two >= T * max_det
    u64* keys;             // [cap]
    float* fbox;           // [T*max_det][4] shifted boxes by flat index
    float* sbox;           // [cap][4]  boxes in sorted order
    float* sscore;         // [cap]
    int32_t* scls;         // [cap]
    int32_t* keeper;       // [cap]  sorted position -> kept slot
    float* kbox;           // [cap][4]  kept boxes (original, not merged)
    float* karea;          // [cap]
    int32_t* kcls;         // [cap]
    int32_t* kpos;         // [cap]  kept slot -> sorted position
    int32_t* klast;        // [cap]  kept slot -> last sorted position that belongs to it
    int32_t* assigned;     // [cap]  NMM: sorted position -> sorted position of its keeper (-1: none yet); aliases kbox
    int32_t* turn;         // [cap]  NMM: the turn (sorted position of the prediction being processed) that assigned it
    int32_t* iskeep;       // [cap]  NMM: 1 = a key of keep_to_merge_list
    int32_t* mcls;         // [cap]  NMM: class used for matching (0 when class-agnostic)
    int32_t* n_cand;       // [0] candidates, [1] kept
    float* out;            // (max_out, 6)
    int32_t* out_count;
    int max_out;
};

__global__ __launch_bounds__(256) void sahi_gather_kernel(const SahiK p) {
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= p.T * p.max_det) return;
    const int t = f / p.max_det, j = f - t * p.max_det;
    if (j >= p.counts[t]) return;
    const float* d = p.det + (size_t)f * p.row;
    float b[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) b[e] = fmaxf(d[e], 0.0f);
    if (p.full_w > 0.0f) { b[0] = fminf(p.full_w, b[0]); b[2] = fminf(p.full_w, b[2]); }
    if (p.full_h > 0.0f) { b[1] = fminf(p.full_h, b[1]); b[3] = fminf(p.full_h, b[3]); }
    if (!(b[0] < b[2]) || !(b[1] < b[3])) return;
    const float ox = p.shift[2 * t], oy = p.shift[2 * t + 1];
    float* fb = p.fbox + (size_t)f * 4;
    fb[0] = b[0] + ox; fb[1] = b[1] + oy; fb[2] = b[2] + ox; fb[3] = b[3] + oy;
    const unsigned cls = p.agnostic ? 0u : (unsigned)(int)d[5];
    const int pos = atomicAdd(p.n_cand, 1);
    p.keys[pos] = ((u64)(0xFFFFu - (cls & 0xFFFFu)) << 48) | ((u64)__float_as_uint(d[4]) << 16) | (u64)(0xFFFFu - (unsigned)f);
}

#define SAHI_SORT_LDS 8192
__global__ __launch_bounds__(1024) void sahi_sort_kernel(const SahiK p) {
    __shared__ u64 sk[SAHI_SORT_LDS];
    const int tid = threadIdx.x, nt = blockDim.x;
    const int n = p.n_cand[0];
    if (n <= 1) return;
    int np2 = 2;
    while (np2 < n) np2 <<= 1;
    u64* keys = p.keys;
    const bool in_lds = np2 <= SAHI_SORT_LDS;
    u64* a = in_lds ? sk : keys;
    for (int i = tid; i < np2; i += nt) {
        if (in_lds) sk[i] = i < n ? keys[i] : 0ull;
        else if (i >= n) keys[i] = 0ull;  // cap >= np2
    }
    __syncthreads();
    for (int k = 2; k <= np2; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = tid; i < np2; i += nt) {
                const int l = i ^ j;
                if (l > i) {
                    const u64 x = a[i], y = a[l];
                    const bool desc = (i & k) == 0;
                    if (desc ? (x < y) : (x > y)) { a[i] = y; a[l] = x; }
                }
            }
            __syncthreads();
        }
    if (in_lds)
        for (int i = tid; i < n; i += nt) keys[i] = sk[i];
}

// greedy_nmm's float32 test of a kept box against a candidate: "matched" unless metric < thr (NaN -> matched).
__device__ __forceinline__ bool nmm_matched(const float* kb, float ka, const float* cb, float ca, int metric, float thr) {
    const float xx1 = fmaxf(cb[0], kb[0]), yy1 = fmaxf(cb[1], kb[1]);
    const float xx2 = fminf(cb[2], kb[2]), yy2 = fminf(cb[3], kb[3]);
    const float w = fmaxf(xx2 - xx1, 0.0f), h = fmaxf(yy2 - yy1, 0.0f);
    const float inter = w * h;
    const float v = metric == 0 ? inter / ((ca - inter) + ka) : inter / fminf(ca, ka);
    return !(v < thr);
}

__global__ __launch_bounds__(256) void sahi_greedy_kernel(const SahiK p) {
    __shared__ u64 keepw[4], supw[4], alw[4];
    __shared__ int sh_i[4];  // [0] undecided, [1] n_kept, [2] first kept slot that can share a class with this chunk
    __shared__ float cbox[256 * 4];
    __shared__ float carea[256];
    __shared__ int ccls[256];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = p.n_cand[0];
    if (tid == 0) { sh_i[1] = 0; sh_i[2] = 0; }
    __syncthreads();

    for (int c0 = 0; c0 < n; c0 += 256) {
        const int n_kept = sh_i[1];
        const int i = c0 + tid;
        const bool have = i < n;
        float bx[4] = {0.f, 0.f, 0.f, 0.f};
        float score = 0.f, area = 0.f;
        int cls = -1, mcls = -1;  // cls: the detection's class; mcls: the class used for matching (0 when agnostic)
        if (have) {
            const u64 k = p.keys[i];
            const int f = (int)(0xFFFFu - (unsigned)(k & 0xFFFFull));
            score = __uint_as_float((unsigned)(k >> 16));
            const float* fb = p.fbox + (size_t)f * 4;
#pragma unroll
            for (int e = 0; e < 4; ++e) bx[e] = fb[e];
            cls = (int)p.det[(size_t)f * p.row + 5];
            mcls = p.agnostic ? 0 : cls;
            area = (bx[2] - bx[0]) * (bx[3] - bx[1]);
#pragma unroll
            for (int e = 0; e < 4; ++e) p.sbox[(size_t)i * 4 + e] = bx[e];
            p.sscore[i] = score;
            p.scls[i] = cls;
        }
        // kept slots of lower classes can never match again: move the scan start past them (classes ascend)
        if (tid == 0) {
            int lo = sh_i[2];
            while (lo < n_kept && p.kcls[lo] < mcls) ++lo;
            sh_i[2] = lo;
        }
        __syncthreads();
        const int lo = sh_i[2];
        // phase A: first kept box (in keep order) of the same class that matches
        int keep_slot = -1;
        if (have)
            for (int j = lo; j < n_kept; ++j) {
                if (p.kcls[j] != mcls) continue;
                if (nmm_matched(p.kbox + (size_t)j * 4, p.karea[j], bx, area, p.metric, p.thr)) { keep_slot = j; break; }
            }
        bool alive = have && keep_slot < 0;
#pragma unroll
        for (int e = 0; e < 4; ++e) cbox[4 * tid + e] = bx[e];
        carea[tid] = area;
        ccls[tid] = mcls;
        if (tid < 4) { keepw[tid] = 0ull; supw[tid] = 0ull; }
        const u64 aliveb = __ballot(alive);
        if (lane == 0) alw[wave] = aliveb;
        __syncthreads();
        u64 alivew[4];
#pragma unroll
        for (int w = 0; w < 4; ++w) alivew[w] = alw[w];
        u64 mrow[4] = {0ull, 0ull, 0ull, 0ull};
        if (alive)
            for (int j = 0; j < tid; ++j) {
                if (!((alivew[j >> 6] >> (j & 63)) & 1ull) || ccls[j] != mcls) continue;
                if (nmm_matched(cbox + 4 * j, carea[j], bx, area, p.metric, p.thr)) mrow[j >> 6] |= 1ull << (j & 63);
            }
        int state = alive ? 0 : 2;  // 0 undecided, 1 keep, 2 suppressed
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
            __syncthreads();
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
        u64 kw[4];
#pragma unroll
        for (int w = 0; w < 4; ++w) kw[w] = keepw[w];
        // kept candidates take the next slots in order
        if (state == 1) {
            int before = 0;
#pragma unroll
            for (int w = 0; w < 4; ++w)
                if (w < wave) before += __popcll(kw[w]);
            before += __popcll(kw[wave] & ((1ull << lane) - 1ull));
            const int slot = n_kept + before;
#pragma unroll
            for (int e = 0; e < 4; ++e) p.kbox[(size_t)slot * 4 + e] = bx[e];
            p.karea[slot] = area;
            p.kcls[slot] = mcls;
            p.kpos[slot] = i;
            p.klast[slot] = i;
            p.keeper[i] = slot;
        }
        __syncthreads();
        // members: the keeper is the first kept box in order that matches -- an earlier chunk's (phase A) or the
        // lowest kept bit of this chunk's overlap row
        if (have && state == 2) {
            if (keep_slot < 0) {
                int jj = -1;
#pragma unroll
                for (int w = 3; w >= 0; --w) {
                    const u64 m = mrow[w] & kw[w];
                    if (m) jj = 64 * w + __ffsll((long long)m) - 1;
                }
                int before = 0;
#pragma unroll
                for (int w = 0; w < 4; ++w) {
                    if (64 * w + 64 <= jj) before += __popcll(kw[w]);
                    else if (64 * w <= jj) before += __popcll(kw[w] & ((1ull << (jj & 63)) - 1ull));
                }
                keep_slot = n_kept + before;
            }
            p.keeper[i] = keep_slot;
            atomicMax(p.klast + keep_slot, i);
        }
        __syncthreads();
        if (tid == 0) sh_i[1] = n_kept + __popcll(kw[0]) + __popcll(kw[1]) + __popcll(kw[2]) + __popcll(kw[3]);
        __syncthreads();
    }
    if (tid == 0) {
        p.n_cand[1] = sh_i[1];
        *p.out_count = sh_i[1] < p.max_out ? sh_i[1] : p.max_out;
    }
}

// sahi's non-greedy nmm (sahi/postprocess/combine.py `nmm`, restated in oracle/sahi_ref.py nmm()).  One workgroup: the
// assignment is a chain over the predictions in score order; every turn is parallel over the other boxes of the class.
__global__ __launch_bounds__(1024) void sahi_nmm_kernel(const SahiK p) {
    __shared__ int sh_end, sh_cnt[16], sh_base;
    const int tid = threadIdx.x, nt = blockDim.x, lane = tid & 63, wave = tid >> 6;
    const int n = p.n_cand[0];
    for (int i = tid; i < n; i += nt) {
        const u64 k = p.keys[i];
        const int f = (int)(0xFFFFu - (unsigned)(k & 0xFFFFull));
        const float* fb = p.fbox + (size_t)f * 4;
#pragma unroll
        for (int e = 0; e < 4; ++e) p.sbox[(size_t)i * 4 + e] = fb[e];
        p.sscore[i] = __uint_as_float((unsigned)(k >> 16));
        const int cls = (int)p.det[(size_t)f * p.row + 5];
        p.scls[i] = cls;
        p.mcls[i] = p.agnostic ? 0 : cls;
        p.assigned[i] = -1;
        p.turn[i] = 0;
        p.iskeep[i] = 0;
    }
    __syncthreads();
    int seg_s = 0, seg_e = 0;
    for (int i = 0; i < n; ++i) {
        if (i == seg_e) {  // uniform: next class segment [i, first position of another class)
            if (tid == 0) sh_end = n;
            __syncthreads();
            const int c = p.mcls[i];
            for (int j = i + 1 + tid; j < n; j += nt)
                if (p.mcls[j] != c) { atomicMin(&sh_end, j); break; }  // classes ascend: a thread's first mismatch is its lowest
            __syncthreads();
            seg_s = i;
            seg_e = sh_end;
            __syncthreads();
        }
        const int a_i = p.assigned[i];
        const bool first = a_i < 0;  // "pred_ind not in merge_to_keep": it becomes a keeper
        const int keep = first ? i : a_i;
        if (first && tid == 0) p.iskeep[i] = 1;
        float pb[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) pb[e] = p.sbox[(size_t)i * 4 + e];
        const float pa = (pb[2] - pb[0]) * (pb[3] - pb[1]);
        for (int j = seg_s + tid; j < seg_e; j += nt) {
            if (j == i || p.assigned[j] >= 0 || (!first && p.iskeep[j])) continue;
            float cb[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) cb[e] = p.sbox[(size_t)j * 4 + e];
            const float ca = (cb[2] - cb[0]) * (cb[3] - cb[1]);
            if (nmm_matched(pb, pa, cb, ca, p.metric, p.thr)) { p.assigned[j] = keep; p.turn[j] = i; }
        }
        __syncthreads();  // position i + 1 reads what this turn assigned
    }
    // kept slots in position order (class ascending, score descending): chunked prefix count of iskeep
    if (tid == 0) sh_base = 0;
    __syncthreads();
    for (int c0 = 0; c0 < n; c0 += nt) {
        const int i = c0 + tid;
        const bool kp = i < n && p.iskeep[i];
        const u64 b = __ballot(kp);
        if (lane == 0) sh_cnt[wave] = __popcll(b);
        __syncthreads();
        int before = sh_base;
        for (int w = 0; w < wave; ++w) before += sh_cnt[w];
        if (kp) {
            const int slot = before + __popcll(b & ((1ull << lane) - 1ull));
            p.keeper[i] = slot;
            p.kpos[slot] = i;
            p.klast[slot] = i;
        }
        __syncthreads();
        if (tid == 0) {
            int t = sh_base;
            for (int w = 0; w < nt / 64; ++w) t += sh_cnt[w];
            sh_base = t;
        }
        __syncthreads();
    }
    for (int i = tid; i < n; i += nt)
        if (!p.iskeep[i]) {
            const int slot = p.keeper[p.assigned[i]];  // every non-keeper was assigned (its own turn would have made it a keeper)
            p.keeper[i] = slot;
            atomicMax(p.klast + slot, i);
        }
    if (tid == 0) {
        p.n_cand[1] = sh_base;
        *p.out_count = sh_base < p.max_out ? sh_base : p.max_out;
    }
}

// has_match: float64 metric of the growing merged box against a member, "> thr".
__device__ __forceinline__ bool has_match64(const double* a, const double* b, int metric, double thr) {
    const double a1 = (a[2] - a[0]) * (a[3] - a[1]), a2 = (b[2] - b[0]) * (b[3] - b[1]);
    const double w = fmax(fmin(a[2], b[2]) - fmax(a[0], b[0]), 0.0), h = fmax(fmin(a[3], b[3]) - fmax(a[1], b[1]), 0.0);
    const double inter = w * h;
    const double v = metric == 0 ? inter / (a1 + a2 - inter) : inter / fmin(a1, a2);
    return v > thr;
}

__global__ __launch_bounds__(256) void sahi_merge_kernel(const SahiK p) {
    const int lane = threadIdx.x & 63;
    const int wv = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwv = (gridDim.x * blockDim.x) >> 6;
    int n_kept = p.n_cand[1];
    if (n_kept > p.max_out) n_kept = p.max_out;
    for (int k = wv; k < n_kept; k += nwv) {
        const int pos0 = p.kpos[k];
        double cur[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) cur[e] = (double)p.sbox[(size_t)pos0 * 4 + e];
        float score = p.sscore[pos0];
        int cls = p.scls[pos0];
        if (p.do_merge == 2) {  // NMM: members in (turn, ascending score = descending position) order
            const int last = p.klast[k];
            bool have_prev = false;
            u64 prev = 0ull;
            for (;;) {
                u64 best = ~0ull;
                for (int pos = pos0 + 1 + lane; pos <= last; pos += 64) {
                    if (p.keeper[pos] != k) continue;
                    const u64 key = ((u64)(unsigned)p.turn[pos] << 32) | (u64)(0xFFFFFFFFu - (unsigned)pos);
                    if ((!have_prev || key > prev) && key < best) best = key;
                }
#pragma unroll
                for (int d = 32; d > 0; d >>= 1) {
                    const u64 o = ((u64)(unsigned)__shfl_xor((int)(best >> 32), d) << 32) | (u64)(unsigned)__shfl_xor((int)(best & 0xFFFFFFFFull), d);
                    if (o < best) best = o;
                }
                if (best == ~0ull) break;  // wave-uniform
                prev = best;
                have_prev = true;
                const int q = (int)(0xFFFFFFFFu - (unsigned)(best & 0xFFFFFFFFull));
                double cb[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) cb[e] = (double)p.sbox[(size_t)q * 4 + e];
                if (has_match64(cur, cb, p.metric, (double)p.thr)) {
                    const float cs = p.sscore[q];
                    if (!(score > cs)) cls = p.scls[q];
                    score = fmaxf(score, cs);
                    cur[0] = fmin(cur[0], cb[0]); cur[1] = fmin(cur[1], cb[1]);
                    cur[2] = fmax(cur[2], cb[2]); cur[3] = fmax(cur[3], cb[3]);
                }
            }
        } else if (p.do_merge) {
            const int last = p.klast[k];
            for (int base = pos0 + 1; base <= last; base += 64) {
                const int pos = base + lane;
                u64 m = __ballot(pos <= last && p.keeper[pos] == k);
                while (m) {  // wave-uniform, members in score order
                    const int q = base + __ffsll((long long)m) - 1;
                    m &= m - 1;
                    double cb[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) cb[e] = (double)p.sbox[(size_t)q * 4 + e];
                    if (has_match64(cur, cb, p.metric, (double)p.thr)) {
                        const float cs = p.sscore[q];
                        if (!(score > cs)) cls = p.scls[q];  // get_merged_category: pred1 only when strictly higher
                        score = fmaxf(score, cs);
                        cur[0] = fmin(cur[0], cb[0]); cur[1] = fmin(cur[1], cb[1]);
                        cur[2] = fmax(cur[2], cb[2]); cur[3] = fmax(cur[3], cb[3]);
                    }
                }
            }
        }
        if (lane == 0) {
            float* o = p.out + (size_t)k * 6;
            o[0] = (float)cur[0]; o[1] = (float)cur[1]; o[2] = (float)cur[2]; o[3] = (float)cur[3];
            o[4] = score;
            o[5] = (float)cls;
        }
    }
}

static int sahi_cap(long long n) {
    long long c = 2;
    while (c < n) c <<= 1;
    return (int)c;
}

// per-candidate words: keys 2, fbox 4, sbox 4, sscore 1, scls 1, keeper 1, kbox 4, karea 1, kcls 1, kpos 1, klast 1
extern "C" size_t bsy_sahi_merge_workspace_bytes(int T, int max_det) {
    if (T <= 0 || max_det <= 0) return 0;
    return (size_t)sahi_cap((long long)T * max_det) * 21 * 4 + 256;
}

extern "C" int bsy_sahi_merge(const float* det, const int32_t* counts, const float* shift, int T, int max_det, int row,
                              int match_metric, float match_threshold, int class_agnostic, int do_merge, float full_w,
                              float full_h, float* out, int32_t* out_count, int max_out, void* workspace,
                              size_t workspace_bytes, bsy_stream stream) {
    if (!det || !counts || !shift || !out || !out_count || T <= 0 || max_det <= 0 || row < 6 || max_out <= 0 ||
        (match_metric != 0 && match_metric != 1) || do_merge < 0 || do_merge > 2)
        BSY_FAIL(BSY_ERR_ARG, "bsy_sahi_merge: bad argument");
    if ((long long)T * max_det > 65536) BSY_FAIL(BSY_ERR_ARG, "bsy_sahi_merge: more than 65536 detection slots");
    if (!workspace || workspace_bytes < bsy_sahi_merge_workspace_bytes(T, max_det))
        BSY_FAIL(BSY_ERR_ARG, "bsy_sahi_merge: workspace too small");
    hipStream_t s = (hipStream_t)stream;
    SahiK p;
    p.det = det; p.counts = counts; p.shift = shift;
    p.T = T; p.max_det = max_det; p.row = row;
    p.metric = match_metric; p.thr = match_threshold; p.agnostic = class_agnostic; p.do_merge = do_merge;
    p.full_w = full_w; p.full_h = full_h;
    p.cap = sahi_cap((long long)T * max_det);
    const size_t cap = (size_t)p.cap;
    unsigned char* w = (unsigned char*)workspace;
    p.n_cand = (int32_t*)w; w += 256;
    p.keys = (u64*)w; w += cap * 8;
    p.fbox = (float*)w; w += cap * 16;
    p.sbox = (float*)w; w += cap * 16;
    p.kbox = (float*)w;  // the NMM assignment keeps no box copies: its four per-position arrays live here
    p.assigned = (int32_t*)w; p.turn = p.assigned + cap; p.iskeep = p.turn + cap; p.mcls = p.iskeep + cap;
    w += cap * 16;
    p.sscore = (float*)w; w += cap * 4;
    p.karea = (float*)w; w += cap * 4;
    p.scls = (int32_t*)w; w += cap * 4;
    p.keeper = (int32_t*)w; w += cap * 4;
    p.kcls = (int32_t*)w; w += cap * 4;
    p.kpos = (int32_t*)w; w += cap * 4;
    p.klast = (int32_t*)w; w += cap * 4;
    p.out = out; p.out_count = out_count; p.max_out = max_out;
    HIP_TRY(hipMemsetAsync(p.n_cand, 0, 256, s));
    const int slots = T * max_det;
    hipLaunchKernelGGL(sahi_gather_kernel, dim3((slots + 255) / 256), dim3(256), 0, s, p);
    hipLaunchKernelGGL(sahi_sort_kernel, dim3(1), dim3(1024), 0, s, p);
    if (do_merge == 2) hipLaunchKernelGGL(sahi_nmm_kernel, dim3(1), dim3(1024), 0, s, p);
    else hipLaunchKernelGGL(sahi_greedy_kernel, dim3(1), dim3(256), 0, s, p);
    hipLaunchKernelGGL(sahi_merge_kernel, dim3(256), dim3(256), 0, s, p);
    HIP_TRY(hipGetLastError());
    return BSY_OK;
}
