// ap_per_class on the device (second half): utils/metrics.py:620-706 ap_per_class + :588-617 compute_ap.
// Compile with -ffp-contract=off; all arithmetic is float64 like the reference's numpy code.
//
//   1. ap_keys_kernel     one 64-bit key per detection: (~class : conf bits : ~index) -> ONE descending sort gives class
//                         ascending, confidence descending, ties by ascending index (np.argsort(-conf) then the per-class
//                         masks `pred_cls == c`, both in one order).
//   2. ap_bitonic_kernel  global-memory bitonic sort, one launch per (k, j) step (N <= 2^20: <= 210 launches).
//   3. ap_class_kernel    one workgroup per class that has labels:
//        * thread t < T walks the class's detections in rank order: tpc = cumsum(tp[:, t]); recall = tpc / (n_l + eps),
//          precision = tpc / (tpc + fpc) with tpc + fpc = rank (:667-676), stored per detection;
//        * r_curve / p_curve: np.interp(-x, -conf, recall[:, 0] / precision[:, 0], left = 0 / 1) at the 1000 x (:672,:676);
//        * thread t: precision envelope (running max from the right, :606) in place, then compute_ap's 101-point
//          interpolation integrated by the trapezoid rule (:609-612);  prec_values = np.interp(x, mrec, mpre) at IoU 0.5 (:682).
//      np.interp is restated as numpy computes it: last knot <= x, exact hits return the knot, else slope * (x - xp) + fp.
// The max-F1 operating point (:690-706: a box filter over a (nc, 1000) array) stays on the host (bs_yolo_amd/val.py).
#include "common.h"

typedef unsigned long long u64;
#define AP_MAXT 16

struct ApK {
    const uint8_t* tp;      // (N, T)
    const float* conf;      // (N)
    const float* pred_cls;  // (N)
    int N, T, Np2;
    const int32_t* classes;  // (nc) unique target classes, ascending
    const int32_t* nt;       // (nc) labels per class
    int nc;
    const double* x101;
    const double* x1000;
    double eps;
    u64* keys;     // [Np2]
    double* rec;   // [N][T]
    double* pre;   // [N][T]
    double* negc;  // [N]
    double* ap;           // (nc, T)
    double* p_curve;      // (nc, 1000)
    double* r_curve;      // (nc, 1000)
    double* prec_values;  // (nc, 1000)
    int32_t* n_pred;      // (nc)
};

__global__ __launch_bounds__(256) void ap_keys_kernel(const ApK p) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= p.Np2) return;
    u64 k = 0ull;
    if (i < p.N) {
        const int c = (int)p.pred_cls[i];
        if (c >= 0 && c < 4095) k = ((u64)(0xFFFu - (unsigned)c) << 52) | ((u64)__float_as_uint(p.conf[i]) << 20) | (u64)(0xFFFFFu - (unsigned)i);
    }
    p.keys[i] = k;
}

__global__ __launch_bounds__(256) void ap_bitonic_kernel(u64* keys, int n, int j, int k) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int l = i ^ j;
    if (l > i) {
        const u64 x = keys[i], y = keys[l];
        const bool desc = (i & k) == 0;
        if (desc ? (x < y) : (x > y)) { keys[i] = y; keys[l] = x; }
    }
}

// numpy.interp for one x over n knots given by accessors (xp non-decreasing)
template <typename XP, typename FP>
__device__ __forceinline__ double ap_interp(double v, int n, XP xp, FP fp, double left, double right) {
    if (v < xp(0)) return left;
    if (v > xp(n - 1)) return right;
    int lo = 0, hi = n;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (v >= xp(mid)) lo = mid + 1; else hi = mid;
    }
    const int j = lo - 1;
    const double xj = xp(j), fj = fp(j);
    if (j == n - 1 || xj == v) return fj;
    const double slope = (fp(j + 1) - fj) / (xp(j + 1) - xj);
    return slope * (v - xj) + fj;
}

__global__ __launch_bounds__(256) void ap_class_kernel(const ApK p) {
    __shared__ int seg[2];
    const int ci = blockIdx.x, tid = threadIdx.x, T = p.T;
    const int c = p.classes[ci];
    const double n_l = (double)p.nt[ci];
    if (tid == 0) {  // the class's segment of the sorted keys: field (0xFFF - c) in the top 12 bits, keys descending
        int b[2];
        for (int w = 0; w < 2; ++w) {  // w = 0: first key with field <= f (start); w = 1: first key with field < f (end)
            const u64 f = (u64)(0xFFFu - (unsigned)c);
            int lo = 0, hi = p.N;
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                const u64 fm = p.keys[mid] >> 52;
                if (w == 0 ? (fm > f) : (fm >= f)) lo = mid + 1; else hi = mid;
            }
            b[w] = lo;
        }
        seg[0] = b[0];
        seg[1] = (c >= 0 && c < 4095) ? b[1] : b[0];  // class ids 0 .. 4094 (field 0 = padding / out-of-range keys)
    }
    __syncthreads();
    const int o = seg[0], n_p = seg[1] - seg[0];
    if (tid == 0) p.n_pred[ci] = n_p;
    if (n_p <= 0 || p.nt[ci] <= 0) return;  // rows stay zero (:663-664)
    double* rec = p.rec + (size_t)o * T;
    double* pre = p.pre + (size_t)o * T;
    double* negc = p.negc + o;
    // ---- cumulative TP -> recall / precision per detection and threshold ------------------------------------------
    if (tid < T) {
        double tpc = 0.0;
        for (int i = 0; i < n_p; ++i) {
            const u64 k = p.keys[o + i];
            const unsigned idx = 0xFFFFFu - (unsigned)(k & 0xFFFFFull);
            tpc += p.tp[(size_t)idx * T + tid] ? 1.0 : 0.0;
            rec[(size_t)i * T + tid] = tpc / (n_l + p.eps);
            pre[(size_t)i * T + tid] = tpc / (double)(i + 1);  // tpc + fpc = number of detections so far
            if (tid == 0) negc[i] = -(double)__uint_as_float((unsigned)(k >> 20));
        }
    }
    __syncthreads();
    // ---- recall / precision against confidence (threshold 0) ------------------------------------------------------------
    {
        auto xp = [&](int i) { return negc[i]; };
        auto fr = [&](int i) { return rec[(size_t)i * T]; };
        auto fq = [&](int i) { return pre[(size_t)i * T]; };
        for (int k = tid; k < 1000; k += 256) {
            const double v = -p.x1000[k];
            p.r_curve[(size_t)ci * 1000 + k] = ap_interp(v, n_p, xp, fr, 0.0, fr(n_p - 1));
            p.p_curve[(size_t)ci * 1000 + k] = ap_interp(v, n_p, xp, fq, 1.0, fq(n_p - 1));
        }
    }
    __syncthreads();
    // ---- precision envelope in place, AP per threshold -----------------------------------------------------------------
    if (tid < T) {
        double run = 0.0;  // mpre's end sentinel
        for (int i = n_p - 1; i >= 0; --i) {
            const double v = pre[(size_t)i * T + tid];
            run = v > run ? v : run;
            pre[(size_t)i * T + tid] = run;
        }
    }
    __syncthreads();
    // mrec = [0, recall, 1], mpre = [max(1, envelope) = 1, envelope, 0]
    auto make_xp = [&](int t) { return [=](int i) { return i == 0 ? 0.0 : (i == n_p + 1 ? 1.0 : rec[(size_t)(i - 1) * T + t]); }; };
    auto make_fp = [&](int t) { return [=](int i) { return i == 0 ? 1.0 : (i == n_p + 1 ? 0.0 : pre[(size_t)(i - 1) * T + t]); }; };
    if (tid < T) {
        auto xp = make_xp(tid);
        auto fp = make_fp(tid);
        double acc = 0.0, yprev = 0.0;
        for (int k = 0; k < 101; ++k) {
            const double y = ap_interp(p.x101[k], n_p + 2, xp, fp, 1.0, 0.0);
            if (k) acc += (p.x101[k] - p.x101[k - 1]) * (y + yprev) / 2.0;
            yprev = y;
        }
        p.ap[(size_t)ci * T + tid] = acc;
    }
    {
        auto xp = make_xp(0);
        auto fp = make_fp(0);
        for (int k = tid; k < 1000; k += 256) p.prec_values[(size_t)ci * 1000 + k] = ap_interp(p.x1000[k], n_p + 2, xp, fp, 1.0, 0.0);
    }
}

static int ap_pow2(int n) {
    int p = 2;
    while (p < n) p <<= 1;
    return p;
}

extern "C" size_t bsy_ap_workspace_bytes(int N, int T) {
    if (N <= 0 || T <= 0) return 0;
    return (size_t)ap_pow2(N) * 8 + (size_t)N * T * 16 + (size_t)N * 8 + 256;
}

extern "C" int bsy_ap_per_class(const uint8_t* tp, const float* conf, const float* pred_cls, int N, int T, const int32_t* classes,
                                const int32_t* nt, int nc, const double* x101, const double* x1000, double eps, double* ap,
                                double* p_curve, double* r_curve, double* prec_values, int32_t* n_pred, void* workspace,
                                size_t workspace_bytes, bsy_stream stream) {
    if (!tp || !conf || !pred_cls || !classes || !nt || !x101 || !x1000 || !ap || !p_curve || !r_curve || !prec_values || !n_pred ||
        N <= 0 || nc <= 0 || T <= 0 || T > AP_MAXT)
        BSY_FAIL(BSY_ERR_ARG, "ap_per_class: bad argument");
    if (N > (1 << 20)) BSY_FAIL(BSY_ERR_ARG, "ap_per_class: more than 2^20 detections (keys carry a 20-bit index)");
    if (!workspace || workspace_bytes < bsy_ap_workspace_bytes(N, T)) BSY_FAIL(BSY_ERR_ARG, "ap_per_class: workspace too small");
    hipStream_t s = (hipStream_t)stream;
    ApK p;
    p.tp = tp; p.conf = conf; p.pred_cls = pred_cls; p.N = N; p.T = T; p.Np2 = ap_pow2(N);
    p.classes = classes; p.nt = nt; p.nc = nc; p.x101 = x101; p.x1000 = x1000; p.eps = eps;
    unsigned char* w = (unsigned char*)workspace;
    p.keys = (u64*)w; w += (size_t)p.Np2 * 8;
    p.rec = (double*)w; w += (size_t)N * T * 8;
    p.pre = (double*)w; w += (size_t)N * T * 8;
    p.negc = (double*)w;
    p.ap = ap; p.p_curve = p_curve; p.r_curve = r_curve; p.prec_values = prec_values; p.n_pred = n_pred;
    HIP_TRY(hipMemsetAsync(ap, 0, (size_t)nc * T * 8, s));
    HIP_TRY(hipMemsetAsync(p_curve, 0, (size_t)nc * 1000 * 8, s));
    HIP_TRY(hipMemsetAsync(r_curve, 0, (size_t)nc * 1000 * 8, s));
    HIP_TRY(hipMemsetAsync(prec_values, 0, (size_t)nc * 1000 * 8, s));
    const int nb = (p.Np2 + 255) / 256;
    hipLaunchKernelGGL(ap_keys_kernel, dim3(nb), dim3(256), 0, s, p);
    for (int k = 2; k <= p.Np2; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) hipLaunchKernelGGL(ap_bitonic_kernel, dim3(nb), dim3(256), 0, s, p.keys, p.Np2, j, k);
    hipLaunchKernelGGL(ap_class_kernel, dim3(nc), dim3(256), 0, s, p);
    HIP_TRY(hipGetLastError());
    return BSY_OK;
}
