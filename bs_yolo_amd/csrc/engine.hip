// Engine / plan objects behind the C ABI (include/bsyolo.h) and the stand-alone operator entry points.
//
// The host side (bs_yolo_amd/plan.py) flattens the model graph -- the work of BaseModel._predict_once
// (nn/tasks.py:138-165) and of every module forward on the path -- into a list of bsy_op records that reference
// workspace buffers by index.  A plan owns that workspace in HBM (one allocation, 256-byte aligned slices) and replays
// the op list on the caller's stream; nothing here synchronises or allocates per call.
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <vector>

#include "common.h"

struct bsy_engine {
    int device = 0;
    void* weights = nullptr;
    size_t weight_bytes = 0;
    // ONE activation arena shared by every plan created with bsy_plan_create_arena (sized for the largest of them; a plan
    // addresses it through host-assigned offsets).  Grows, never shrinks; plans resolve the base at run time.
    char* arena = nullptr;
    size_t arena_bytes = 0;
    unsigned arena_gen = 0;  // bumped whenever the arena moves: captured graphs hold its addresses
    unsigned weights_gen = 0;  // bumped by every bsy_engine_load_weights: captured graphs hold addresses inside the weight blob too
};

struct bsy_plan {
    bsy_engine* eng = nullptr;
    std::vector<bsy_op> ops;
    std::vector<size_t> buf_off;
    std::vector<size_t> buf_size;
    char* workspace = nullptr;     // plan-owned workspace (bsy_plan_create) ...
    size_t workspace_bytes = 0;
    bool use_arena = false;        // ... or the engine's shared arena (bsy_plan_create_arena): base = eng->arena
    char* base() const { return use_arena ? eng->arena : workspace; }
    std::vector<hipEvent_t> events;
    // side streams ("lanes") for independent op chains + the events that fork / join them
    std::vector<hipStream_t> lanes;      // index 0 unused (lane 0 = caller's stream)
    std::vector<hipEvent_t> lane_done;   // per lane: recorded at its tail before a join
    std::vector<hipEvent_t> lane_fork;   // per lane: recorded on the caller's stream where the lane forks (one event per lane:
                                         // an event that still has a waiter pending is never re-recorded)
    float last_event_overhead_ms = 0.f;  // bsy_plan_profile: median empty event interval of the last call
    size_t guard = 0;                    // bytes of guard band behind every buffer (BSY_PLAN_GUARD)
    // captured forwards (bsy_plan_graph_launch): one executable graph per set of external pointers, oldest first
    struct Captured {
        std::vector<void*> ext;
        unsigned arena_gen, weights_gen;
        hipGraphExec_t exec;
    };
    std::vector<Captured> graphs;
    bool graph_unavailable = false;      // a capture failed once: this plan runs eagerly from then on
    void drop_graphs() {
        for (auto& g : graphs) (void)hipGraphExecDestroy(g.exec);
        graphs.clear();
    }
};

extern "C" int bsy_sizeof_op(void) { return (int)sizeof(bsy_op); }  // the host mirrors the record with ctypes: checked at load

extern "C" int bsy_engine_create(int device, bsy_engine** out) {
    if (!out) BSY_FAIL(BSY_ERR_ARG, "engine_create: null out");
    int ndev = 0;
    HIP_TRY(hipGetDeviceCount(&ndev));
    if (device < 0 || device >= ndev) BSY_FAIL(BSY_ERR_ARG, "engine_create: device %d of %d", device, ndev);
    bsy_engine* e = new bsy_engine();
    e->device = device;
    *out = e;
    return BSY_OK;
}

extern "C" void bsy_engine_destroy(bsy_engine* e) {
    if (!e) return;
    if (e->weights) (void)hipFree(e->weights);
    if (e->arena) (void)hipFree(e->arena);
    delete e;
}

extern "C" int bsy_engine_load_weights(bsy_engine* e, const void* host_blob, size_t bytes) {
    if (!e || !host_blob || !bytes) BSY_FAIL(BSY_ERR_ARG, "load_weights: bad argument");
    HIP_TRY(hipSetDevice(e->device));
    if (e->weights) {
        // a reload on an engine that has run: nothing may still be reading the old blob (plans resolve `weights + w_off` when they
        // are enqueued), and graphs captured so far hold its addresses -- weights_gen below makes bsy_plan_graph_launch drop them
        HIP_TRY(hipDeviceSynchronize());
        HIP_TRY(hipFree(e->weights));
        e->weights = nullptr;
    }
    ++e->weights_gen;
    // BSY_WEIGHT_GUARD=<bytes> (test aid): that many bytes of 0x7C behind the blob -- as f16 every pair is a NaN (0x7C7C), as
    // f32 a huge finite number -- so a kernel that reads past the packed weights AND uses what it read changes the outputs
    // (tests compare a poisoned engine with a plain one; round 1 found such a read in the flat-DMA path by accident, 8ab4d40).
    // Reads past the blob whose values are discarded (padding rows of a cout tile) stay invisible by design.
    const char* g = getenv("BSY_WEIGHT_GUARD");
    const size_t guard = g ? ((size_t)atoll(g) + 255) & ~(size_t)255 : 0;
    if (hipMalloc(&e->weights, bytes + 256 + guard) != hipSuccess) BSY_FAIL(BSY_ERR_ALLOC, "load_weights: hipMalloc(%zu) failed", bytes);
    HIP_TRY(hipMemcpy(e->weights, host_blob, bytes, hipMemcpyHostToDevice));
    HIP_TRY(hipMemset((char*)e->weights + bytes, guard ? 0x7C : 0, 256 + guard));
    e->weight_bytes = bytes;
    return BSY_OK;
}

extern "C" void bsy_plan_destroy(bsy_plan* p);

static int plan_make_lanes(bsy_plan* p) {
    int max_lane = 0;
    for (const auto& o : p->ops) max_lane = o.lane > max_lane ? o.lane : max_lane;
    if (max_lane > 31) BSY_FAIL(BSY_ERR_ARG, "plan_create: too many lanes");
    p->lanes.assign(max_lane + 1, nullptr);
    p->lane_done.assign(max_lane + 1, nullptr);
    p->lane_fork.assign(max_lane + 1, nullptr);
    bool ok = true;
    for (int l = 1; l <= max_lane && ok; ++l)
        ok = hipStreamCreateWithFlags(&p->lanes[l], hipStreamNonBlocking) == hipSuccess &&
             hipEventCreateWithFlags(&p->lane_done[l], hipEventDisableTiming) == hipSuccess &&
             hipEventCreateWithFlags(&p->lane_fork[l], hipEventDisableTiming) == hipSuccess;
    if (!ok) BSY_FAIL(BSY_ERR_HIP, "plan_create: stream/event creation failed");
    return BSY_OK;
}

extern "C" int bsy_plan_create(bsy_engine* e, const bsy_op* ops, int n_ops, const int64_t* buf_bytes, int n_bufs,
                               bsy_plan** out) {
    if (!e || !ops || n_ops <= 0 || !out || n_bufs < 0 || (n_bufs && !buf_bytes)) BSY_FAIL(BSY_ERR_ARG, "plan_create: bad argument");
    if (!e->weights) BSY_FAIL(BSY_ERR_STATE, "plan_create: load weights first");
    HIP_TRY(hipSetDevice(e->device));
    bsy_plan* p = new bsy_plan();
    p->eng = e;
    p->ops.assign(ops, ops + n_ops);
    // BSY_PLAN_GUARD=<bytes> (test aid): a guard band of that many bytes, filled with 0xA5, behind every buffer;
    // bsy_plan_check_guards reports the first band a kernel wrote into (an out-of-bounds store).
    {
        const char* g = getenv("BSY_PLAN_GUARD");
        p->guard = g ? ((size_t)atoll(g) + 255) & ~(size_t)255 : 0;
    }
    size_t off = 0;
    for (int i = 0; i < n_bufs; ++i) {
        if (buf_bytes[i] < 0) { delete p; BSY_FAIL(BSY_ERR_ARG, "plan_create: negative buffer size"); }
        p->buf_off.push_back(off);
        p->buf_size.push_back((size_t)buf_bytes[i]);
        off += (((size_t)buf_bytes[i] + 255) & ~(size_t)255) + p->guard;
    }
    p->workspace_bytes = off + 256;
    if (hipMalloc((void**)&p->workspace, p->workspace_bytes) != hipSuccess) {
        delete p;
        BSY_FAIL(BSY_ERR_ALLOC, "plan_create: hipMalloc(%zu) failed", off + 256);
    }
    // zero once so that padding channels nobody writes are finite
    if (hipMemset(p->workspace, 0, p->workspace_bytes) != hipSuccess) {
        (void)hipFree(p->workspace);
        delete p;
        BSY_FAIL(BSY_ERR_HIP, "plan_create: hipMemset failed");
    }
    for (int i = 0; i < n_bufs && p->guard; ++i)
        if (hipMemset(p->workspace + p->buf_off[i] + ((p->buf_size[i] + 255) & ~(size_t)255), 0xA5, p->guard) != hipSuccess) {
            (void)hipFree(p->workspace);
            delete p;
            BSY_FAIL(BSY_ERR_HIP, "plan_create: guard memset failed");
        }
    const int rc = plan_make_lanes(p);
    if (rc != BSY_OK) { bsy_plan_destroy(p); return rc; }
    *out = p;
    return BSY_OK;
}

// The activation buffers of this plan live at HOST-assigned byte offsets (multiples of 256; they may overlap where
// bs_yolo_amd/plan.py found the buffers' lifetimes disjoint) inside the engine's shared arena of at least `arena_bytes`.
// The arena is (re)allocated here when it is too small -- after a device synchronisation, so no earlier plan can still be
// running in the old one.  Plans of one engine share the arena: run them on one stream at a time (the reference serialises
// predict() the same way, engine/predictor.py:113,229).
extern "C" int bsy_plan_create_arena(bsy_engine* e, const bsy_op* ops, int n_ops, const int64_t* buf_bytes, const int64_t* buf_off,
                                     int n_bufs, int64_t arena_bytes, bsy_plan** out) {
    if (!e || !ops || n_ops <= 0 || !out || n_bufs < 0 || (n_bufs && (!buf_bytes || !buf_off)) || arena_bytes < 0)
        BSY_FAIL(BSY_ERR_ARG, "plan_create_arena: bad argument");
    if (!e->weights) BSY_FAIL(BSY_ERR_STATE, "plan_create_arena: load weights first");
    for (int i = 0; i < n_bufs; ++i)
        if (buf_bytes[i] < 0 || buf_off[i] < 0 || (buf_off[i] & 255) || buf_off[i] + buf_bytes[i] > arena_bytes)
            BSY_FAIL(BSY_ERR_ARG, "plan_create_arena: buffer %d (offset %lld, %lld bytes) outside the arena of %lld bytes", i,
                     (long long)buf_off[i], (long long)buf_bytes[i], (long long)arena_bytes);
    HIP_TRY(hipSetDevice(e->device));
    const size_t need = (size_t)arena_bytes + 256;
    if (need > e->arena_bytes) {
        HIP_TRY(hipDeviceSynchronize());
        if (e->arena) { HIP_TRY(hipFree(e->arena)); e->arena = nullptr; e->arena_bytes = 0; }
        if (hipMalloc((void**)&e->arena, need) != hipSuccess) BSY_FAIL(BSY_ERR_ALLOC, "plan_create_arena: hipMalloc(%zu) failed", need);
        e->arena_bytes = need;
        ++e->arena_gen;
        // zero once so that padding channels nobody writes start finite (later tenants of a region leave ordinary f16 / f32
        // activations behind; no kernel consumes padding channels)
        HIP_TRY(hipMemset(e->arena, 0, need));
    }
    bsy_plan* p = new bsy_plan();
    p->eng = e;
    p->use_arena = true;
    p->ops.assign(ops, ops + n_ops);
    for (int i = 0; i < n_bufs; ++i) {
        p->buf_off.push_back((size_t)buf_off[i]);
        p->buf_size.push_back((size_t)buf_bytes[i]);
    }
    const int rc = plan_make_lanes(p);
    if (rc != BSY_OK) { bsy_plan_destroy(p); return rc; }
    *out = p;
    return BSY_OK;
}

extern "C" size_t bsy_engine_arena_bytes(const bsy_engine* e) { return e ? e->arena_bytes : 0; }

// Test aid: synchronises the device and checks the guard bands (plans created under BSY_PLAN_GUARD).  Returns BSY_OK and
// *bad_buf = -1 when every band is intact, else the index of the first buffer whose band was written and the byte offset
// of the first damaged byte inside the band.
extern "C" int bsy_plan_check_guards(bsy_plan* p, int32_t* bad_buf, int64_t* bad_off) {
    if (!p || !bad_buf) BSY_FAIL(BSY_ERR_ARG, "plan_check_guards: bad argument");
    *bad_buf = -1;
    if (bad_off) *bad_off = 0;
    if (!p->guard || p->use_arena) return BSY_OK;  // arena plans carry no guard bands (buffers alias by design)
    HIP_TRY(hipDeviceSynchronize());
    std::vector<unsigned char> host(p->guard);
    for (size_t i = 0; i < p->buf_off.size(); ++i) {
        HIP_TRY(hipMemcpy(host.data(), p->workspace + p->buf_off[i] + ((p->buf_size[i] + 255) & ~(size_t)255), p->guard, hipMemcpyDeviceToHost));
        for (size_t k = 0; k < p->guard; ++k)
            if (host[k] != 0xA5) {
                *bad_buf = (int32_t)i;
                if (bad_off) *bad_off = (int64_t)k;
                return BSY_OK;
            }
    }
    return BSY_OK;
}

extern "C" void bsy_plan_destroy(bsy_plan* p) {
    if (!p) return;
    p->drop_graphs();
    for (auto ev : p->events) (void)hipEventDestroy(ev);
    for (auto ev : p->lane_done) if (ev) (void)hipEventDestroy(ev);
    for (auto st : p->lanes) if (st) (void)hipStreamDestroy(st);
    for (auto ev : p->lane_fork) if (ev) (void)hipEventDestroy(ev);
    if (p->workspace) (void)hipFree(p->workspace);
    delete p;
}

namespace {
struct Resolver {
    const bsy_plan* p;
    void* const* ext;
    int n_ext;
    bool ok = true;
    // returns base pointer of the buffer (no channel offset) or nullptr when the view is absent
    char* base(const bsy_view& v) {
        if (v.buf < 0) return nullptr;
        if (v.buf >= BSY_EXT_BASE) {
            const int s = v.buf - BSY_EXT_BASE;
            if (s >= n_ext || !ext[s]) { ok = false; bsy_set_error("plan_run: external slot %d not bound", s); return nullptr; }
            return (char*)ext[s];
        }
        if ((size_t)v.buf >= p->buf_off.size()) { ok = false; bsy_set_error("plan_run: buffer %d out of range", v.buf); return nullptr; }
        return p->base() + p->buf_off[v.buf];
    }
    half_t* h(const bsy_view& v) { char* b = base(v); return b ? (half_t*)b + v.coff : nullptr; }
    float* f(const bsy_view& v) { char* b = base(v); return b ? (float*)b + v.coff : nullptr; }
};

// fp32 correctness mode (bsy_op.prec == 1): same op records, f32 views, kernels of ref32.hip.  DECODE and RAW_NCHW consume f32
// logit maps in both modes and fall through to the common path.
// Byte offsets of the two f16 weight planes an fp32x plan's conv record carries behind its f32 matrix (weights.py pack_record):
// [f32 K x Cout][pad to 256][hi: Cout x Kpad f16][pad to 256][lo: Cout x Kpad f16], Kpad = K rounded up to 32.
static inline size_t fp32x_align(size_t n) { return (n + 255) & ~(size_t)255; }

int run_op_f32(const bsy_plan* p, const bsy_op& op, Resolver& R, hipStream_t s, bool& handled) {
    const char* wb = (const char*)p->eng->weights;
    handled = true;
    switch (op.kind) {
        case BSY_OP_CONV_FIRST:
        case BSY_OP_CONV: {
            Conv32Args a;
            memset(&a, 0, sizeof(a));
            a.first = op.kind == BSY_OP_CONV_FIRST;
            a.src0 = a.first ? (const void*)R.base(op.src0) : (const void*)R.f(op.src0);
            a.src_dtype = op.in_dtype;
            a.src1 = R.f(op.src1);
            a.ld0 = op.src0.ld; a.ld1 = op.src1.buf >= 0 ? op.src1.ld : 0;
            a.C0 = op.src0.C; a.C1 = op.src1.buf >= 0 ? op.src1.C : 0;
            a.up0 = op.up0; a.up1 = op.up1;
            a.B = op.B; a.H = op.H; a.W = op.W; a.OH = op.OH; a.OW = op.OW; a.ks = op.ksize; a.stride = op.stride; a.pad = op.pad;
            a.w = (const float*)(wb + op.w_off); a.bias = (const float*)(wb + op.b_off);
            a.dst = R.f(op.dst); a.ldd = op.dst.ld; a.Cout = op.dst.C;
            a.res = R.f(op.res); a.ldr = op.res.buf >= 0 ? op.res.ld : 0;
            a.act = op.act; a.dst_scale = op.dst_scale > 0 ? op.dst_scale : 1; a.dst_dy = op.dst_dy; a.dst_dx = op.dst_dx;
            if (op.out_f32 >= 2) BSY_FAIL(BSY_ERR_ARG, "fp32 mode: fused Detect decoder ops are not part of fp32 plans");
            if (!R.ok) return BSY_ERR_ARG;
            if (op.prec == 2) {  // fp32x: split-f16 planes behind the f32 matrix; shapes the kernel does not take stay exact
                const size_t K = (size_t)a.ks * a.ks * (a.C0 + a.C1), kpad = (K + 31) & ~(size_t)31;
                const size_t hi_off = fp32x_align(K * a.Cout * 4), lo_off = hi_off + fp32x_align((size_t)a.Cout * kpad * 2);
                a.wx_hi = (const half_t*)(wb + op.w_off + hi_off);
                a.wx_lo = (const half_t*)(wb + op.w_off + lo_off);
                a.wx_kpad = (int)kpad;
                if (conv32x_mfma_supported(a)) return launch_conv32x_mfma(a, s);
            }
            return launch_conv32(a, s);
        }
        case BSY_OP_DWCONV:
        case BSY_OP_DWCONV_G: {
            Dw32Args a;
            memset(&a, 0, sizeof(a));
            const bool g = op.kind == BSY_OP_DWCONV_G;
            a.src = R.f(op.src0); a.lds = op.src0.ld; a.B = op.B; a.H = op.H; a.W = op.W; a.C = op.src0.C; a.OH = op.OH; a.OW = op.OW;
            a.kh = g ? op.ksize : 3; a.kw = g ? op.pad : 3; a.stride = g ? op.stride : 1;
            a.wld = g ? op.heads : op.src0.C;
            a.w = (const float*)(wb + op.w_off) + (g ? op.key_dim : 0); a.b = (const float*)(wb + op.b_off) + (g ? op.key_dim : 0);
            a.dst = R.f(op.dst); a.ldd = op.dst.ld; a.act_c = g ? op.act : (op.act ? op.src0.C : 0);
            a.res = g ? nullptr : R.f(op.res); a.ldr = (!g && op.res.buf >= 0) ? op.res.ld : 0;
            if (!R.ok) return BSY_ERR_ARG;
            return launch_dw32(a, s);
        }
        case BSY_OP_SPPF_POOL: {
            float* b = R.f(op.src0);
            if (!R.ok) return BSY_ERR_ARG;
            return launch_sppf32(b, op.src0.ld, op.B, op.H, op.W, op.src0.C, s);
        }
        case BSY_OP_ATTN: {
            const float* q = R.f(op.src0);
            float* o = R.f(op.dst);
            if (!R.ok) return BSY_ERR_ARG;
            if (op.prec == 2 && attn32x_supported(op.src0.ld, op.dst.ld, op.key_dim, op.head_dim, q, o))  // fp32x: split-f16 products on the matrix pipe
                return launch_attn32x(q, op.src0.ld, op.B, op.H * op.W, op.heads, op.key_dim, op.head_dim, op.scale, o, op.dst.ld, s);
            return launch_attn32(q, op.src0.ld, op.B, op.H * op.W, op.heads, op.key_dim, op.head_dim, op.scale, o, op.dst.ld, s);
        }
        case BSY_OP_NHWC2NCHW: {
            if (op.dst.buf >= BSY_EXT_BASE && (op.dst.buf - BSY_EXT_BASE >= R.n_ext || !R.ext[op.dst.buf - BSY_EXT_BASE])) return BSY_OK;
            const float* src = R.f(op.src0);
            void* out = R.base(op.dst);
            if (!R.ok) return BSY_ERR_ARG;
            return launch_nhwc2nchw32(src, op.src0.ld, op.B, op.src0.C, op.H * op.W, out, op.out_dtype, s);
        }
        case BSY_OP_COPY: {
            const float* src = R.f(op.src0);
            float* dst = R.f(op.dst);
            if (!R.ok) return BSY_ERR_ARG;
            return launch_copy32(src, op.src0.ld, op.up0, op.B, op.H, op.W, op.src0.C, dst, op.dst.ld, s);
        }
        case BSY_OP_GAP: {
            const float* src = R.f(op.src0);
            float* dst = R.f(op.dst);
            if (!R.ok) return BSY_ERR_ARG;
            return launch_gap32(src, op.src0.ld, op.B, op.H, op.W, op.src0.C, dst, op.dst.ld, s);
        }
        case BSY_OP_MSCA_MIX: {
            Mix32Args a;
            for (int i = 0; i < 4; ++i) {
                const bsy_view& bv = i < 3 ? op.box[i] : op.res;
                const bsy_view& lv = i < 3 ? op.cls[i] : op.msk[0];
                a.br[i] = R.f(bv); a.ldb[i] = bv.ld; a.lg[i] = R.f(lv); a.ldl[i] = lv.ld;
            }
            a.B = op.B; a.HW = op.H * op.W; a.C = op.dst.C; a.dst = R.f(op.dst); a.ldd = op.dst.ld;
            if (!R.ok) return BSY_ERR_ARG;
            return launch_mix32(a, s);
        }
        case BSY_OP_MUL: {
            const float* x = R.f(op.src0);
            const float* y = R.f(op.src1);
            float* dst = R.f(op.dst);
            if (!R.ok) return BSY_ERR_ARG;
            return launch_mul32(x, op.src0.ld, y, op.src1.ld, (long long)op.B * op.H * op.W, op.dst.C, dst, op.dst.ld, s);
        }
        case BSY_OP_ELA: {
            ElaArgs a;
            memset(&a, 0, sizeof(a));
            a.lds = op.src0.ld; a.B = op.B; a.H = op.H; a.W = op.W; a.C = op.src0.C; a.k = op.ksize;
            const float* blob = (const float*)(wb + op.w_off);
            a.wsp = blob; a.wch = blob + (size_t)a.C * a.k; a.gnw = a.wch + (size_t)a.C * a.k; a.gnb = a.gnw + a.C;
            a.ch_coef = op.scale; a.sp_coef = op.lvl_stride[0]; a.res_coef = op.lvl_stride[1];
            a.scratch = R.f(op.res); a.ldd = op.dst.ld;
            const float* src = R.f(op.src0);
            float* dst = R.f(op.dst);
            if (!R.ok) return BSY_ERR_ARG;
            return launch_ela32(a, src, dst, s);
        }
        case BSY_OP_DECODE:
        case BSY_OP_RAW_NCHW:
            handled = false;
            return BSY_OK;
        default:
            BSY_FAIL(BSY_ERR_ARG, "fp32 mode: op kind %d has no fp32 implementation (fused kinds are fp16-only)", op.kind);
    }
}

int run_op(const bsy_plan* p, const bsy_op& op, Resolver& R, hipStream_t s, ConvArgs* cargs = nullptr) {
    const char* wb = (const char*)p->eng->weights;
    if (op.prec == 1 || op.prec == 2) {
        bool handled = false;
        const int rc = run_op_f32(p, op, R, s, handled);
        if (rc != BSY_OK || handled) return rc;
    } else if (op.prec != 0) {
        BSY_FAIL(BSY_ERR_ARG, "plan_run: unknown precision mode %d", op.prec);
    }
    switch (op.kind) {
        case BSY_OP_CONV_FIRST: {
            ConvFirstArgs a;
            a.img = R.base(op.src0); a.img_dtype = op.in_dtype;
            a.B = op.B; a.H = op.H; a.W = op.W; a.OH = op.OH; a.OW = op.OW;
            a.ksize = op.ksize; a.stride = op.stride; a.pad = op.pad;
            a.w = (const void*)(wb + op.w_off); a.b = (const float*)(wb + op.b_off);
            a.dst = R.h(op.dst); a.ldd = op.dst.ld; a.Cout = op.dst.C; a.act = op.act;
            if (!R.ok) return BSY_ERR_ARG;
            return launch_conv_first(a, s);
        }
        case BSY_OP_STEM: {
            StemArgs a;
            a.img = R.base(op.src0); a.img_dtype = op.in_dtype;
            a.B = op.B; a.H = op.H; a.W = op.W; a.OH = op.OH; a.OW = op.OW;
            a.w0 = (const void*)(wb + op.w_off); a.b0 = (const float*)(wb + op.b_off);
            a.w1 = (const void*)(wb + op.w2_off); a.b1 = (const float*)(wb + op.b2_off);
            a.C0 = op.mid_c; a.C1 = op.dst.C;
            int cp = 0;
            bsy_conv_packed_dims(a.C1, a.C0, 3, &cp, &a.Kpad1);
            a.dst = R.h(op.dst); a.ldd = op.dst.ld; a.act = op.act;
            if (!R.ok) return BSY_ERR_ARG;
            return launch_stem_fused(a, s);
        }
        case BSY_OP_BNECK: {
            BneckArgs a;
            a.src = R.h(op.src0); a.lds = op.src0.ld; a.B = op.B; a.H = op.H; a.W = op.W;
            a.C = op.dst.C; a.CH = op.mid_c;
            a.w1 = (const void*)(wb + op.w_off); a.b1 = (const float*)(wb + op.b_off);
            a.w2 = (const void*)(wb + op.w2_off); a.b2 = (const float*)(wb + op.b2_off);
            int cp = 0;
            bsy_conv_packed_dims(a.CH, a.C, 3, &cp, &a.Kpad1);
            bsy_conv_packed_dims(a.C, a.CH, 3, &cp, &a.Kpad2);
            a.dst = R.h(op.dst); a.ldd = op.dst.ld; a.act = op.act;
            if (!R.ok) return BSY_ERR_ARG;
            return launch_bneck_fused(a, s);
        }
        case BSY_OP_S2D: {
            const void* img = R.base(op.src0);
            half_t* dst = R.h(op.dst);
            if (!R.ok) return BSY_ERR_ARG;
            return launch_s2d(img, op.in_dtype, op.B, op.H, op.W, dst, op.dst.ld, s);
        }
        case BSY_OP_C3K2: {
            C3k2Args a;
            a.src = R.h(op.src0); a.lds = op.src0.ld; a.B = op.B; a.H = op.H; a.W = op.W;
            a.Cin = op.src0.C; a.C = op.mid_c; a.C2 = op.dst.C;
            a.w1 = wb + op.aux_off[0]; a.b1 = (const float*)(wb + op.aux_off[1]);
            a.wa = wb + op.aux_off[2]; a.ba = (const float*)(wb + op.aux_off[3]);
            a.wb = wb + op.aux_off[4]; a.bb = (const float*)(wb + op.aux_off[5]);
            a.w4 = wb + op.aux_off[6]; a.b4 = (const float*)(wb + op.aux_off[7]);
            a.dst = R.h(op.dst); a.ldd = op.dst.ld;
            if (!R.ok) return BSY_ERR_ARG;
            return launch_c3k2_fused(a, s);
        }
        case BSY_OP_DWPW: {
            DwPwArgs a;
            a.src = R.h(op.src0); a.lds = op.src0.ld; a.B = op.B; a.H = op.H; a.W = op.W; a.C = op.src0.C;
            a.dww = (const float*)(wb + op.w_off); a.dwb = (const float*)(wb + op.b_off);
            a.wgt = (const void*)(wb + op.w2_off); a.bias = (const float*)(wb + op.b2_off);
            a.dst = R.h(op.dst); a.ldd = op.dst.ld; a.Cout = op.dst.C; a.act = op.act;
            if (!R.ok) return BSY_ERR_ARG;
            return launch_dwpw_fused(a, s);
        }
        case BSY_OP_DWCONV_G: {
            DwGenArgs a;
            a.src = R.h(op.src0); a.lds = op.src0.ld; a.B = op.B; a.H = op.H; a.W = op.W; a.C = op.src0.C;
            a.OH = op.OH; a.OW = op.OW; a.kh = op.ksize; a.kw = op.pad; a.stride = op.stride;
            a.wld = op.heads; a.w = (const float*)(wb + op.w_off) + op.key_dim; a.b = (const float*)(wb + op.b_off) + op.key_dim;
            a.dst = R.h(op.dst); a.ldd = op.dst.ld; a.act_c = op.act;  // act = number of leading channels with SiLU
            a.ident_c0 = op.mid_c;  // first channel of an identity-kernel half (plan.py dwconv_g kind "dwg_ext"), 0 = none
            if (!R.ok) return BSY_ERR_ARG;
            return launch_dwconv_generic(a, s);
        }
        case BSY_OP_PMSFA_TAIL: {
            PmsfaArgs a;
            a.P = R.h(op.src0); a.ldp = op.src0.ld; a.x = R.h(op.res); a.ldx = op.res.ld; a.dst = R.h(op.dst); a.ldd = op.dst.ld;
            a.B = op.B; a.H = op.H; a.W = op.W; a.C = op.dst.C;
            a.w2 = (const float*)(wb + op.aux_off[0]); a.b2 = (const float*)(wb + op.aux_off[1]); a.wld2 = op.heads;
            a.w3 = (const float*)(wb + op.aux_off[2]); a.b3 = (const float*)(wb + op.aux_off[3]); a.wld3 = op.key_dim;
            a.w4 = (const half_t*)(wb + op.aux_off[4]); a.b4 = (const float*)(wb + op.aux_off[5]);
            int cp = 0;
            bsy_conv_packed_dims(a.C, a.C, 1, &cp, &a.kpad4);
            if (!R.ok || op.src0.C != a.C || op.res.C != a.C) return BSY_ERR_ARG;
            return launch_pmsfa_tail(a, s);
        }
        case BSY_OP_CHAIN: {
            ChainArgs a;
            memset(&a, 0, sizeof(a));
            a.a0 = R.h(op.src0); a.a1 = R.h(op.src1);
            a.lda0 = op.src0.ld; a.lda1 = op.src1.buf >= 0 ? op.src1.ld : 0;
            a.CA0 = op.src0.C; a.CA1 = op.src1.buf >= 0 ? op.src1.C : 0;
            a.w1 = (const half_t*)(wb + op.w_off); a.b1 = (const float*)(wb + op.b_off); a.N1 = op.heads; a.act1 = op.nl;
            a.d1 = R.h(op.box[0]); a.ldd1 = op.box[0].buf >= 0 ? op.box[0].ld : 0;
            a.r1 = R.h(op.box[2]); a.ldr1 = op.box[2].buf >= 0 ? op.box[2].ld : 0;
            a.keep0 = op.key_dim; a.LC = op.mid_c;
            a.h2 = R.h(op.box[1]); a.ldh2 = op.box[1].buf >= 0 ? op.box[1].ld : 0; a.CH2 = op.box[1].buf >= 0 ? op.box[1].C : 0;
            a.w2 = (const half_t*)(wb + op.w2_off); a.b2 = (const float*)(wb + op.b2_off); a.N2 = op.dst.C; a.act2 = op.act;
            a.d2 = R.h(op.dst); a.ldd2 = op.dst.ld;
            a.r2 = R.h(op.res); a.ldr2 = op.res.buf >= 0 ? op.res.ld : 0;
            a.M = (long long)op.B * op.H * op.W;
            if (!R.ok) return BSY_ERR_ARG;
            if (op.up0 || op.up1 || (op.box[0].buf >= 0 && op.box[0].C != a.N1) || (op.box[2].buf >= 0 && op.box[2].C != a.N1) || (op.res.buf >= 0 && op.res.C != a.N2))
                BSY_FAIL(BSY_ERR_ARG, "chain: inconsistent views");
            return launch_chain(a, s);
        }
        case BSY_OP_COPY: {
            const half_t* src = R.h(op.src0);
            half_t* dst = R.h(op.dst);
            if (!R.ok) return BSY_ERR_ARG;
            return launch_copy_view(src, op.src0.ld, op.up0, op.B, op.H, op.W, op.src0.C, dst, op.dst.ld, s);
        }
        case BSY_OP_GAP: {
            const half_t* src = R.h(op.src0);
            half_t* dst = R.h(op.dst);
            if (!R.ok) return BSY_ERR_ARG;
            return launch_gap(src, op.src0.ld, op.B, op.H, op.W, op.src0.C, dst, op.dst.ld, s);
        }
        case BSY_OP_MSCA_SPATIAL: {
            MscaSpArgs a;
            a.src = R.h(op.src0); a.lds = op.src0.ld; a.B = op.B; a.H = op.H; a.W = op.W; a.C = op.src0.C;
            for (int i = 0; i < 9; ++i) {
                a.w[i] = (const float*)(wb + op.aux_off[2 * i]);
                a.b[i] = (const float*)(wb + op.aux_off[2 * i + 1]);
            }
            for (int i = 0; i < 4; ++i) {
                const bsy_view& bv = i < 3 ? op.box[i] : op.res;
                const bsy_view& gv = i < 3 ? op.cls[i] : op.msk[0];
                a.br[i] = R.h(bv); a.ldb[i] = bv.ld; a.gap[i] = R.h(gv); a.ldg[i] = gv.ld;
            }
            if (!R.ok) return BSY_ERR_ARG;
            return launch_msca_spatial(a, s);
        }
        case BSY_OP_MSCA_MIX: {
            MixArgs a;
            for (int i = 0; i < 4; ++i) {
                const bsy_view& bv = i < 3 ? op.box[i] : op.res;
                const bsy_view& lv = i < 3 ? op.cls[i] : op.msk[0];
                a.br[i] = R.h(bv); a.ldb[i] = bv.ld; a.lg[i] = R.f(lv); a.ldl[i] = lv.ld;
            }
            a.B = op.B; a.HW = op.H * op.W; a.C = op.dst.C; a.dst = R.h(op.dst); a.ldd = op.dst.ld;
            if (!R.ok) return BSY_ERR_ARG;
            return launch_msca_mix(a, s);
        }
        case BSY_OP_MUL: {
            const half_t* x = R.h(op.src0);
            const half_t* y = R.h(op.src1);
            half_t* dst = R.h(op.dst);
            if (!R.ok) return BSY_ERR_ARG;
            return launch_mul(x, op.src0.ld, y, op.src1.ld, (long long)op.B * op.H * op.W, op.dst.C, dst, op.dst.ld, s);
        }
        case BSY_OP_ELA: {
            ElaArgs a;
            a.src = R.h(op.src0); a.lds = op.src0.ld; a.B = op.B; a.H = op.H; a.W = op.W; a.C = op.src0.C; a.k = op.ksize;
            const float* blob = (const float*)(wb + op.w_off);
            a.wsp = blob; a.wch = blob + (size_t)a.C * a.k; a.gnw = a.wch + (size_t)a.C * a.k; a.gnb = a.gnw + a.C;
            a.ch_coef = op.scale; a.sp_coef = op.lvl_stride[0]; a.res_coef = op.lvl_stride[1];
            a.scratch = R.f(op.res); a.dst = R.h(op.dst); a.ldd = op.dst.ld;
            if (!R.ok) return BSY_ERR_ARG;
            return launch_ela(a, s);
        }
        case BSY_OP_CONV: {
            ConvArgs a;
            a.src0 = R.h(op.src0); a.src1 = R.h(op.src1);
            a.ld0 = op.src0.ld; a.ld1 = op.src1.buf >= 0 ? op.src1.ld : 0;
            a.C0 = op.src0.C; a.C1 = op.src1.buf >= 0 ? op.src1.C : 0;
            a.up0 = op.up0; a.up1 = op.up1;
            a.B = op.B; a.H = op.H; a.W = op.W; a.OH = op.OH; a.OW = op.OW;
            a.ksize = op.ksize; a.stride = op.stride; a.pad = op.pad;
            a.wgt = (const half_t*)(wb + op.w_off); a.bias = (const float*)(wb + op.b_off);
            a.dst = op.out_f32 ? (void*)R.f(op.dst) : (void*)R.h(op.dst);
            a.ldd = op.dst.ld; a.Cout = op.dst.C; a.out_f32 = op.out_f32;
            a.res = R.h(op.res); a.ldr = op.res.buf >= 0 ? op.res.ld : 0;
            a.act = op.act; a.dst_scale = op.dst_scale; a.dst_dy = op.dst_dy; a.dst_dx = op.dst_dx;
            a.epi = 0; a.y = nullptr; a.raw = nullptr; a.y_f32 = a.raw_f32 = a.A = a.a0 = a.nrows = a.rawC = 0; a.lvl_stride = 0.f;
            a.tail_wgt = nullptr; a.tail_bias = nullptr;
            if (op.out_f32 >= 2) {  // fused Detect decoder: dst = the prediction tensor y, box[0] = the level's raw map (optional)
                a.epi = op.out_f32; a.out_f32 = 0; a.dst = nullptr; a.ldd = 8; a.Cout = op.nl;
                a.y = R.base(op.dst); a.y_f32 = op.out_dtype == BSY_F32; a.A = op.A; a.a0 = op.lvl_h[1]; a.nrows = op.dst.C;
                a.lvl_stride = op.lvl_stride[0];
                const bsy_view& rv = op.box[0];
                const bool bound = rv.buf >= BSY_EXT_BASE && rv.buf - BSY_EXT_BASE < R.n_ext && R.ext[rv.buf - BSY_EXT_BASE];
                a.raw = bound ? R.base(rv) : nullptr; a.raw_f32 = a.y_f32; a.rawC = rv.C;
                if (op.out_f32 == 3 && op.mid_c > 0) {  // box-branch tail: this op is the 3x3 conv, (w2, b2) the branch's last 1x1 conv
                    a.Cout = op.mid_c;
                    a.tail_wgt = (const half_t*)(wb + op.w2_off); a.tail_bias = (const float*)(wb + op.b2_off);
                }
            }
            if (op.ksplit > 0 && op.out_f32 == 0) {  // split-K (latency-mode plans): ksplit = channel slices | tap slices << 8, box[0] = the f32 slab buffer
                a.nsl_c = op.ksplit & 255; a.nsl_t = (op.ksplit >> 8) & 255;
                a.split_ws = R.f(op.box[0]);
            }
            a.cfg = op.tuned_cfg - 1;  // 0 = not tuned -> heuristic
            if (!R.ok) return BSY_ERR_ARG;
            if (cargs) { *cargs = a; return BSY_OK; }
            return launch_conv(a, s);
        }
        case BSY_OP_DWCONV: {
            DwArgs a;
            a.src = R.h(op.src0); a.lds = op.src0.ld; a.B = op.B; a.H = op.H; a.W = op.W; a.C = op.src0.C;
            a.w = (const float*)(wb + op.w_off); a.b = (const float*)(wb + op.b_off);
            a.dst = R.h(op.dst); a.ldd = op.dst.ld; a.act = op.act;
            a.res = R.h(op.res); a.ldr = op.res.buf >= 0 ? op.res.ld : 0;
            if (!R.ok) return BSY_ERR_ARG;
            return launch_dwconv(a, s);
        }
        case BSY_OP_SPPF_POOL: {
            half_t* b = R.h(op.src0);
            if (!R.ok) return BSY_ERR_ARG;
            return launch_sppf_pool(b, op.src0.ld, op.B, op.H, op.W, op.src0.C, s);
        }
        case BSY_OP_ATTN: {
            AttnArgs a;
            a.qkv = R.h(op.src0); a.ld = op.src0.ld; a.B = op.B; a.N = op.H * op.W; a.heads = op.heads;
            a.key_dim = op.key_dim; a.head_dim = op.head_dim; a.scale = op.scale;
            a.out = R.h(op.dst); a.ldo = op.dst.ld;
            if (!R.ok) return BSY_ERR_ARG;
            return launch_attention(a, s);
        }
        case BSY_OP_DECODE: {
            DecodeArgs a;
            memset(&a, 0, sizeof(a));
            a.nl = op.nl; a.B = op.B; a.nc = op.nc; a.nm = op.nm; a.A = op.A;
            for (int l = 0; l < op.nl && l < 3; ++l) {
                a.box[l] = R.f(op.box[l]); a.cls[l] = R.f(op.cls[l]); a.msk[l] = op.nm ? R.f(op.msk[l]) : nullptr;
                a.ldb[l] = op.box[l].ld; a.ldc[l] = op.cls[l].ld; a.ldm[l] = op.nm ? op.msk[l].ld : 0;
                a.h[l] = op.lvl_h[l]; a.w[l] = op.lvl_w[l]; a.stride[l] = op.lvl_stride[l];
            }
            a.y = R.base(op.dst); a.y_dtype = op.out_dtype;
            if (!R.ok) return BSY_ERR_ARG;
            return launch_decode(a, s);
        }
        case BSY_OP_RAW_NCHW: {
            const int l = op.level;
            if (l < 0 || l > 2) BSY_FAIL(BSY_ERR_ARG, "raw_nchw: level %d", l);
            // optional output: silently skipped when the slot is not bound
            if (op.dst.buf >= BSY_EXT_BASE && (op.dst.buf - BSY_EXT_BASE >= R.n_ext || !R.ext[op.dst.buf - BSY_EXT_BASE]))
                return BSY_OK;
            const float* box = R.f(op.box[l]);
            const float* cls = R.f(op.cls[l]);
            void* out = R.base(op.dst);
            if (!R.ok) return BSY_ERR_ARG;
            return launch_raw_nchw(box, op.box[l].ld, cls, op.cls[l].ld, op.B, op.lvl_h[l], op.lvl_w[l], op.nc, out,
                                   op.out_dtype, s);
        }
        case BSY_OP_NHWC2NCHW: {
            if (op.dst.buf >= BSY_EXT_BASE && (op.dst.buf - BSY_EXT_BASE >= R.n_ext || !R.ext[op.dst.buf - BSY_EXT_BASE]))
                return BSY_OK;  // optional output
            const half_t* src = R.h(op.src0);
            void* out = R.base(op.dst);
            if (!R.ok) return BSY_ERR_ARG;
            return launch_nhwc2nchw(src, op.src0.ld, op.B, op.src0.C, op.H * op.W, out, op.out_dtype, s);
        }
        default:
            BSY_FAIL(BSY_ERR_ARG, "plan_run: unknown op kind %d", op.kind);
    }
}
}  // namespace

extern "C" void bsy_plan_destroy(bsy_plan* p);

// Ops are listed in a valid serial order.  An op on lane k > 0 runs on the plan's side stream k, which is forked from
// the caller's stream (event wait) the first time it is used after a join; an op flagged `join` first waits for every
// active side stream.  Chains on different lanes must not depend on each other between a fork and the next join
// (plan.py guarantees it: per-level Detect / Segment branches).
extern "C" int bsy_plan_run(bsy_plan* p, void* const* ext, int n_ext, bsy_stream stream) {
    if (!p || (n_ext && !ext)) BSY_FAIL(BSY_ERR_ARG, "plan_run: bad argument");
    hipStream_t main = (hipStream_t)stream;
    Resolver R{p, ext, n_ext};
    unsigned active = 0;  // bit k: side stream k has work since the last join
    int rc = BSY_OK;
    for (size_t i = 0; i < p->ops.size() && rc == BSY_OK; ++i) {
        const bsy_op& op = p->ops[i];
        // no early return inside this loop: every exit goes through the tail join below
        if (op.join && active) {
            for (size_t l = 1; l < p->lanes.size() && rc == BSY_OK; ++l)
                if (active & (1u << l)) {
                    if (hipEventRecord(p->lane_done[l], p->lanes[l]) != hipSuccess ||
                        hipStreamWaitEvent(main, p->lane_done[l], 0) != hipSuccess) {
                        bsy_set_error("plan_run: joining lane %zu failed", l);
                        rc = BSY_ERR_HIP;
                    } else {
                        active &= ~(1u << l);
                    }
                }
            if (rc != BSY_OK) break;
        }
        hipStream_t s = main;
        if (op.lane > 0) {
            s = p->lanes[op.lane];
            if (!(active & (1u << op.lane))) {
                if (hipEventRecord(p->lane_fork[op.lane], main) != hipSuccess ||
                    hipStreamWaitEvent(s, p->lane_fork[op.lane], 0) != hipSuccess) {
                    bsy_set_error("plan_run: forking lane %d failed", op.lane);
                    rc = BSY_ERR_HIP;
                    break;
                }
                active |= 1u << op.lane;
            }
        }
        rc = run_op(p, op, R, s);
    }
    // never leave side streams un-joined (error paths included): the caller only synchronises its own stream.  After an
    // error the lanes are drained on the host as well, so that nothing of this forward can still be writing the workspace
    // or the caller's outputs when the error is reported (the last-error text of the failing call is kept).
    for (size_t l = 1; l < p->lanes.size(); ++l)
        if (active & (1u << l)) {
            if (rc != BSY_OK) (void)hipStreamSynchronize(p->lanes[l]);
            (void)hipEventRecord(p->lane_done[l], p->lanes[l]);
            (void)hipStreamWaitEvent(main, p->lane_done[l], 0);
        }
    return rc;
}

// The forward as ONE graph launch.  At a rank's share of a strong-scaled batch (8 or 16 images: 70-odd launches of 4-15 us) the
// eager replay is bound by the host's launch rate and by the fork / join events of the head lanes; a captured graph hands the
// whole dependency structure -- lanes included, they become edges -- to the GPU's command processor in one submission.
// The graph bakes in every address: it is cached per set of external pointers (callers that cycle through a few input / output
// buffers hit the cache; at most BSY_GRAPH_MAX = 8 graphs per plan, least recently used dropped), and dropped when the arena
// moves, the weight blob is replaced (bsy_engine_load_weights) or the tuning changes.  Captures on `stream` in thread-local mode (bsy_plan_run issues nothing but kernel launches and event
// record / wait pairs); if a capture fails the plan runs eagerly, now and later.  *captured (may be null): 1 when this call
// had to capture, 0 when it replayed a cached graph, -1 when it ran eagerly.
#define BSY_GRAPH_MAX 8
extern "C" int bsy_plan_graph_launch(bsy_plan* p, void* const* ext, int n_ext, bsy_stream stream, int* captured) {
    if (!p || (n_ext && !ext)) BSY_FAIL(BSY_ERR_ARG, "plan_graph_launch: bad argument");
    hipStream_t s = (hipStream_t)stream;
    if (captured) *captured = -1;
    if (p->graph_unavailable || !s) return bsy_plan_run(p, ext, n_ext, stream);  // (the null stream cannot be captured)
    const unsigned gen = p->use_arena ? p->eng->arena_gen : 0u, wgen = p->eng->weights_gen;
    std::vector<void*> key(ext, ext + n_ext);
    for (size_t i = 0; i < p->graphs.size(); ++i)
        if (p->graphs[i].arena_gen == gen && p->graphs[i].weights_gen == wgen && p->graphs[i].ext == key) {
            if (captured) *captured = 0;
            if (i + 1 != p->graphs.size()) std::rotate(p->graphs.begin() + i, p->graphs.begin() + i + 1, p->graphs.end());  // most recently used last
            HIP_TRY(hipGraphLaunch(p->graphs.back().exec, s));
            return BSY_OK;
        }
    for (size_t i = 0; i < p->graphs.size();)  // graphs of an arena that has moved, or of a weight blob that has been replaced since
        if (p->graphs[i].arena_gen != gen || p->graphs[i].weights_gen != wgen) { (void)hipGraphExecDestroy(p->graphs[i].exec); p->graphs.erase(p->graphs.begin() + i); } else ++i;
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    if (hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal) != hipSuccess) {
        (void)hipGetLastError();
        p->graph_unavailable = true;
        return bsy_plan_run(p, ext, n_ext, stream);
    }
    const int rc = bsy_plan_run(p, ext, n_ext, stream);
    const hipError_t ec = hipStreamEndCapture(s, &graph);
    if (rc != BSY_OK || ec != hipSuccess || !graph || hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0) != hipSuccess) {
        if (graph) (void)hipGraphDestroy(graph);
        (void)hipGetLastError();
        p->graph_unavailable = true;
        if (rc != BSY_OK) return rc;  // the plan itself is at fault: report it
        return bsy_plan_run(p, ext, n_ext, stream);
    }
    (void)hipGraphDestroy(graph);
    if (p->graphs.size() >= BSY_GRAPH_MAX) { (void)hipGraphExecDestroy(p->graphs.front().exec); p->graphs.erase(p->graphs.begin()); }
    p->graphs.push_back(bsy_plan::Captured{key, gen, wgen, exec});
    if (captured) *captured = 1;
    HIP_TRY(hipGraphLaunch(exec, s));
    return BSY_OK;
}

// Per-op autotuning of the conv kernel configuration (tile shape / K-step / ring depth): runs the plan once, timing
// every valid configuration of every conv op with HIP events on `stream` (1 warm-up + 3 timed launches each) and
// records the fastest in the plan.  All configurations of a layer are bit-identical in their results: the K walk is a function of the
// layer's shape (conv_mfma.hip conv_korder), never of the configuration.
extern "C" int bsy_plan_autotune(bsy_plan* p, void* const* ext, int n_ext, bsy_stream stream) {
    if (!p || (n_ext && !ext)) BSY_FAIL(BSY_ERR_ARG, "plan_autotune: bad argument");
    p->drop_graphs();
    hipStream_t s = (hipStream_t)stream;
    hipEvent_t e0, e1;
    HIP_TRY(hipEventCreate(&e0));
    HIP_TRY(hipEventCreate(&e1));
    Resolver R{p, ext, n_ext};
    int rc = BSY_OK;
    for (size_t i = 0; i < p->ops.size() && rc == BSY_OK; ++i) {
        bsy_op& op = p->ops[i];
        if (op.kind != BSY_OP_CONV || op.tuned_cfg > 0 || op.prec != 0) { rc = run_op(p, op, R, s); continue; }  // preset (bsy_plan_set_tuning): kept
        ConvArgs a;
        rc = run_op(p, op, R, s, &a);
        if (rc != BSY_OK) break;
        int cand[BSY_CONV_MAX_CFG];
        const int nc = conv_candidates(a, cand, BSY_CONV_MAX_CFG);
        // Candidates are timed ROUND-ROBIN (three rounds; per round one warm-up + a burst of five launches each) and ranked by
        // their best burst: clock ramps and neighbours' activity drift over a sweep, and timing every candidate in every phase
        // of it keeps near-ties from going to whichever kernel happened to run in a quiet moment (r02: the same layer flipped
        // between a 0.045 ms and a 0.056 ms configuration from box to box).
        float best_of[BSY_CONV_MAX_CFG];
        for (int c = 0; c < nc; ++c) best_of[c] = 1e30f;
        for (int round = 0; round < 3 && rc == BSY_OK; ++round)
            for (int c = 0; c < nc && rc == BSY_OK; ++c) {
                a.cfg = cand[c];
                rc = launch_conv(a, s);  // warm-up
                if (rc != BSY_OK) break;
                if (hipEventRecord(e0, s) != hipSuccess) { rc = BSY_ERR_HIP; break; }
                for (int r = 0; r < 5 && rc == BSY_OK; ++r) rc = launch_conv(a, s);
                if (rc != BSY_OK) break;
                float ms = 0.f;
                if (hipEventRecord(e1, s) != hipSuccess || hipEventSynchronize(e1) != hipSuccess ||
                    hipEventElapsedTime(&ms, e0, e1) != hipSuccess) { rc = BSY_ERR_HIP; bsy_set_error("plan_autotune: event timing failed"); break; }
                if (ms < best_of[c]) best_of[c] = ms;
            }
        float best = 1e30f;
        int best_cfg = cand[0];
        for (int c = 0; c < nc; ++c)
            if (best_of[c] < best) { best = best_of[c]; best_cfg = cand[c]; }
        if (rc != BSY_OK) break;
        op.tuned_cfg = best_cfg + 1;
        // runner-up within 12 %: the host re-times the two IN PLACE (bsy_plan_profile passes of the whole forward) and keeps the one
        // that is faster where it runs -- back-to-back launches of one layer see its operands in the caches, the forward does not
        // (two runners-up: reserved0 and, for conv ops unused, head_dim hold them as id + 1)
        float alt = 1e30f, alt2 = 1e30f;
        op.reserved0 = 0;
        op.head_dim = 0;
        for (int c = 0; c < nc; ++c) {
            if (cand[c] == best_cfg || best_of[c] > 1.12f * best) continue;
            if (best_of[c] < alt) { alt2 = alt; op.head_dim = op.reserved0; alt = best_of[c]; op.reserved0 = cand[c] + 1; }
            else if (best_of[c] < alt2) { alt2 = best_of[c]; op.head_dim = cand[c] + 1; }
        }
        a.cfg = best_cfg;
        rc = launch_conv(a, s);
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    return rc;
}

// HOST array out[n_ops]: the configuration id chosen for each op (-1 = not a conv / not tuned).
extern "C" int bsy_plan_get_tuning(bsy_plan* p, int32_t* out, int n) {
    if (!p || !out || n != (int)p->ops.size()) BSY_FAIL(BSY_ERR_ARG, "plan_get_tuning: bad argument");
    for (int i = 0; i < n; ++i) out[i] = p->ops[i].kind == BSY_OP_CONV ? p->ops[i].tuned_cfg - 1 : -1;
    return BSY_OK;
}

// HOST array out[n_ops]: the rank-th runner-up (1 or 2) of the last bsy_plan_autotune for each op it timed (-1 = none within 12 % of
// the winner, not a conv, or not timed by that call).
extern "C" int bsy_plan_get_tuning_alt(bsy_plan* p, int rank, int32_t* out, int n) {
    if (!p || !out || n != (int)p->ops.size() || rank < 1 || rank > 2) BSY_FAIL(BSY_ERR_ARG, "plan_get_tuning_alt: bad argument");
    for (int i = 0; i < n; ++i) out[i] = p->ops[i].kind == BSY_OP_CONV ? (rank == 1 ? p->ops[i].reserved0 : p->ops[i].head_dim) - 1 : -1;
    return BSY_OK;
}

// HOST array cfg[n_ops]: >= 0 presets that configuration for op i (a later bsy_plan_autotune leaves the op alone), -1 keeps
// the op as it is, -2 clears it (heuristic / to be tuned).  A preset that is not valid for the op's shape is ignored at launch.
extern "C" int bsy_plan_set_tuning(bsy_plan* p, const int32_t* cfg, int n) {
    if (!p || !cfg || n != (int)p->ops.size()) BSY_FAIL(BSY_ERR_ARG, "plan_set_tuning: bad argument");
    p->drop_graphs();  // captured forwards hold the old configurations' launches
    for (int i = 0; i < n; ++i) {
        if (p->ops[i].kind != BSY_OP_CONV || cfg[i] == -1) continue;
        p->ops[i].tuned_cfg = cfg[i] >= 0 ? cfg[i] + 1 : 0;
    }
    return BSY_OK;
}

// HOST array valid[n_ops]: 1 = the configuration preset for op i (bsy_plan_set_tuning) can run the op's shape, 0 = it cannot (a launch
// would fall back to the heuristic configuration: a stale or foreign tune cache), -1 = not a conv op or not preset.  Launches nothing.
extern "C" int bsy_plan_check_tuning(bsy_plan* p, void* const* ext, int n_ext, int32_t* valid, int n) {
    if (!p || !valid || (n_ext && !ext) || n != (int)p->ops.size()) BSY_FAIL(BSY_ERR_ARG, "plan_check_tuning: bad argument");
    Resolver R{p, ext, n_ext};
    for (int i = 0; i < n; ++i) {
        const bsy_op& op = p->ops[i];
        valid[i] = -1;
        if (op.kind != BSY_OP_CONV || op.tuned_cfg <= 0 || op.prec != 0) continue;
        ConvArgs a;
        const int rc = run_op(p, op, R, nullptr, &a);  // arguments only: nothing is launched when `cargs` is given
        if (rc != BSY_OK) return rc;
        valid[i] = conv_cfg_valid(a, op.tuned_cfg - 1) ? 1 : 0;
    }
    return BSY_OK;
}

extern "C" int bsy_plan_copy_buffer(bsy_plan* p, int buf, void* host_dst, size_t bytes) {
    if (!p || !host_dst || buf < 0 || (size_t)buf >= p->buf_off.size()) BSY_FAIL(BSY_ERR_ARG, "copy_buffer: bad argument");
    if (bytes > p->buf_size[buf]) BSY_FAIL(BSY_ERR_ARG, "copy_buffer: %zu > buffer size %zu", bytes, p->buf_size[buf]);
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(host_dst, p->base() + p->buf_off[buf], bytes, hipMemcpyDeviceToHost));
    return BSY_OK;
}

extern "C" int bsy_plan_profile(bsy_plan* p, void* const* ext, int n_ext, bsy_stream stream, float* ms_per_op) {
    if (!p || !ms_per_op || (n_ext && !ext)) BSY_FAIL(BSY_ERR_ARG, "plan_profile: bad argument");
    hipStream_t s = (hipStream_t)stream;
    const size_t n = p->ops.size();
    constexpr size_t NCAL = 16;  // empty event intervals recorded after the ops: the cost of the event pair itself
    while (p->events.size() < n + 1 + NCAL) {
        hipEvent_t ev;
        HIP_TRY(hipEventCreate(&ev));
        p->events.push_back(ev);
    }
    Resolver R{p, ext, n_ext};
    HIP_TRY(hipEventRecord(p->events[0], s));
    for (size_t i = 0; i < n; ++i) {
        const int rc = run_op(p, p->ops[i], R, s);
        if (rc != BSY_OK) return rc;
        HIP_TRY(hipEventRecord(p->events[i + 1], s));
    }
    for (size_t i = 0; i < NCAL; ++i) HIP_TRY(hipEventRecord(p->events[n + 1 + i], s));
    HIP_TRY(hipEventSynchronize(p->events[n + NCAL]));
    // An interval between two recorded events holds the kernel AND the event packet between it and its neighbour (a few
    // microseconds that a normal run, which records no events between kernels, does not pay: rocprofv3's kernel
    // durations are that much shorter).  The median empty interval measures it; it is subtracted from every op.
    float cal[NCAL];
    for (size_t i = 0; i < NCAL; ++i) HIP_TRY(hipEventElapsedTime(&cal[i], p->events[n + i], p->events[n + 1 + i]));
    std::sort(cal, cal + NCAL);
    const float ev_ms = cal[NCAL / 2];
    for (size_t i = 0; i < n; ++i) {
        HIP_TRY(hipEventElapsedTime(&ms_per_op[i], p->events[i], p->events[i + 1]));
        ms_per_op[i] = ms_per_op[i] > ev_ms ? ms_per_op[i] - ev_ms : 0.f;
    }
    p->last_event_overhead_ms = ev_ms;
    return BSY_OK;
}

// In-place autotune: every candidate configuration of every untuned conv op is timed WHERE IT RUNS -- serial profile passes of the
// whole forward (bsy_plan_profile's event scheme), pass k running candidate k of every op at once, `rounds` passes per candidate index,
// best time kept -- instead of back-to-back launches of one layer, which see their operands in L2 / Infinity Cache and ranked near-ties
// differently from the forward (round 2: re-timing just the top three in place already moved the forward by 0.6-2.4 %).  Ops with a
// preset (bsy_plan_set_tuning) are left alone.  Costs max_i(#candidates) x rounds forwards (~0.4 s for YOLO11s at 64 x 640 x 640).
extern "C" int bsy_plan_autotune_in_place(bsy_plan* p, void* const* ext, int n_ext, bsy_stream stream, int rounds) {
    if (!p || (n_ext && !ext) || rounds < 1 || rounds > 16) BSY_FAIL(BSY_ERR_ARG, "plan_autotune_in_place: bad argument");
    p->drop_graphs();
    hipStream_t s = (hipStream_t)stream;
    const size_t n = p->ops.size();
    Resolver R{p, ext, n_ext};
    std::vector<std::vector<int>> cand(n);
    size_t K = 0;
    for (size_t i = 0; i < n; ++i) {  // one plain pass: the operands of every op exist, and the conv ops yield their arguments
        bsy_op& op = p->ops[i];
        if (op.kind == BSY_OP_CONV && op.tuned_cfg <= 0 && op.prec == 0) {
            ConvArgs a;
            int rc = run_op(p, op, R, s, &a);
            if (rc != BSY_OK) return rc;
            int list[BSY_CONV_MAX_CFG];
            const int nc = conv_candidates(a, list, BSY_CONV_MAX_CFG);
            cand[i].assign(list, list + (nc > 0 ? nc : 0));
            K = cand[i].size() > K ? cand[i].size() : K;
        }
        const int rc = run_op(p, op, R, s);
        if (rc != BSY_OK) return rc;
    }
    if (K == 0) return BSY_OK;
    std::vector<float> best(n, 1e30f), ms(n);
    std::vector<std::vector<float>> tk(n);  // best time of every candidate
    for (size_t i = 0; i < n; ++i) tk[i].assign(cand[i].size(), 1e30f);
    std::vector<int> best_c(n, -1);
    for (size_t i = 0; i < n; ++i)
        if (!cand[i].empty()) best_c[i] = cand[i][0];
    for (size_t k = 0; k < K; ++k) {
        for (size_t i = 0; i < n; ++i)
            if (!cand[i].empty()) p->ops[i].tuned_cfg = (k < cand[i].size() ? cand[i][k] : best_c[i]) + 1;
        for (int r = 0; r < rounds; ++r) {
            const int rc = bsy_plan_profile(p, ext, n_ext, stream, ms.data());
            if (rc != BSY_OK) {
                for (size_t i = 0; i < n; ++i)
                    if (!cand[i].empty()) p->ops[i].tuned_cfg = 0;
                return rc;
            }
            for (size_t i = 0; i < n; ++i)
                if (k < cand[i].size()) {
                    if (ms[i] < tk[i][k]) tk[i][k] = ms[i];
                    if (ms[i] < best[i]) { best[i] = ms[i]; best_c[i] = cand[i][k]; }
                }
        }
    }
    for (size_t i = 0; i < n; ++i)
        if (!cand[i].empty()) {
            p->ops[i].tuned_cfg = best_c[i] + 1;
            p->ops[i].reserved0 = 0;
            p->ops[i].head_dim = 0;
        }
    return BSY_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// Stand-alone operators
// ---------------------------------------------------------------------------------------------------------------
extern "C" int bsy_conv2d(const void* x, int ldx, int B, int H, int W, int C1, const void* w, const float* b, void* y,
                          int ldy, int C2, int ksize, int stride, int act, const void* res, int ldr, int y_f32,
                          bsy_stream stream) {
    if (!x || !w || !b || !y) BSY_FAIL(BSY_ERR_ARG, "conv2d: null pointer");
    ConvArgs a;
    memset(&a, 0, sizeof(a));
    a.src0 = (const half_t*)x; a.ld0 = ldx; a.C0 = C1; a.B = B; a.H = H; a.W = W;
    a.ksize = ksize; a.stride = stride; a.pad = ksize / 2;  // autopad (nn/modules/conv.py:29-35)
    a.OH = (H + 2 * a.pad - ksize) / stride + 1; a.OW = (W + 2 * a.pad - ksize) / stride + 1;
    a.wgt = (const half_t*)w; a.bias = b; a.dst = y; a.ldd = ldy; a.Cout = C2; a.out_f32 = y_f32;
    a.res = (const half_t*)res; a.ldr = ldr; a.act = act; a.dst_scale = 1;
    const char* forced = getenv("BSY_CONV_CFG");  // experiments: BSY_CONV_CFG=<tile<<4|variant>; an id that is not valid
    a.cfg = forced ? atoi(forced) : -1;           // for this shape is an error (so A/B scripts see it), unset = heuristic
    if (a.cfg >= 0 && !conv_cfg_valid(a, a.cfg)) BSY_FAIL(BSY_ERR_ARG, "conv: BSY_CONV_CFG=%d is not valid for this shape", a.cfg);
    return launch_conv(a, (hipStream_t)stream);
}

extern "C" int bsy_conv2d_f32(const float* x, int ldx, int B, int H, int W, int C1, const float* w, const float* b, float* y, int ldy,
                              int C2, int ksize, int stride, int act, const float* res, int ldr, int impl, bsy_stream stream) {
    if (!x || !w || !b || !y) BSY_FAIL(BSY_ERR_ARG, "conv2d_f32: null pointer");
    if (B <= 0 || H <= 0 || W <= 0 || C1 <= 0 || C2 <= 0 || ksize < 1 || !(ksize & 1) || (stride != 1 && stride != 2) || ldx < C1 || ldy < C2 ||
        (res && ldr < C2) || impl < 0 || impl > 2)
        BSY_FAIL(BSY_ERR_ARG, "conv2d_f32: bad shape");
    Conv32Args a;
    memset(&a, 0, sizeof(a));
    a.src0 = x; a.src_dtype = BSY_F32; a.ld0 = ldx; a.C0 = C1; a.B = B; a.H = H; a.W = W;
    a.ks = ksize; a.stride = stride; a.pad = ksize / 2;  // autopad (nn/modules/conv.py:29-35)
    a.OH = (H + 2 * a.pad - ksize) / stride + 1; a.OW = (W + 2 * a.pad - ksize) / stride + 1;
    a.w = w; a.bias = b; a.dst = y; a.ldd = ldy; a.Cout = C2; a.res = res; a.ldr = ldr; a.act = act; a.dst_scale = 1;
    if (impl == 1) return launch_conv32_scalar(a, (hipStream_t)stream);
    if (impl == 2) return launch_conv32_mfma(a, (hipStream_t)stream);
    return launch_conv32(a, (hipStream_t)stream);
}

extern "C" int bsy_conv2d_f32x(const float* x, int ldx, int B, int H, int W, int C1, const void* w_hi, const void* w_lo, int k_pad,
                               const float* b, float* y, int ldy, int C2, int ksize, int stride, int act, const float* res, int ldr,
                               bsy_stream stream) {
    if (!x || !w_hi || !w_lo || !b || !y) BSY_FAIL(BSY_ERR_ARG, "conv2d_f32x: null pointer");
    if (B <= 0 || H <= 0 || W <= 0 || C1 <= 0 || C2 <= 0 || ksize < 1 || !(ksize & 1) || (stride != 1 && stride != 2) || ldx < C1 || ldy < C2 ||
        (res && ldr < C2) || k_pad < ksize * ksize * C1 || (k_pad & 31))
        BSY_FAIL(BSY_ERR_ARG, "conv2d_f32x: bad shape");
    Conv32Args a;
    memset(&a, 0, sizeof(a));
    a.src0 = x; a.src_dtype = BSY_F32; a.ld0 = ldx; a.C0 = C1; a.B = B; a.H = H; a.W = W;
    a.ks = ksize; a.stride = stride; a.pad = ksize / 2;
    a.OH = (H + 2 * a.pad - ksize) / stride + 1; a.OW = (W + 2 * a.pad - ksize) / stride + 1;
    a.w = (const float*)w_hi;  // only its alignment is looked at (conv32_mfma_supported); this entry point never runs the exact kernels
    a.bias = b; a.dst = y; a.ldd = ldy; a.Cout = C2; a.res = res; a.ldr = ldr; a.act = act; a.dst_scale = 1;
    a.wx_hi = (const half_t*)w_hi; a.wx_lo = (const half_t*)w_lo; a.wx_kpad = k_pad;
    return launch_conv32x_mfma(a, (hipStream_t)stream);
}

extern "C" int bsy_conv_first_f32(const void* img, int img_dtype, int B, int H, int W, const float* w, const float* b, float* y, int ldy,
                                 int C2, int ksize, int stride, int act, int impl, bsy_stream stream) {
    if (!img || !w || !b || !y) BSY_FAIL(BSY_ERR_ARG, "conv_first_f32: null pointer");
    if (B <= 0 || H <= 0 || W <= 0 || C2 <= 0 || ksize < 1 || !(ksize & 1) || (stride != 1 && stride != 2) || ldy < C2 || impl < 0 || impl > 2 ||
        (img_dtype != BSY_F16 && img_dtype != BSY_F32))
        BSY_FAIL(BSY_ERR_ARG, "conv_first_f32: bad shape");
    Conv32Args a;
    memset(&a, 0, sizeof(a));
    a.first = 1; a.src0 = img; a.src_dtype = img_dtype; a.C0 = 3; a.B = B; a.H = H; a.W = W;
    a.ks = ksize; a.stride = stride; a.pad = ksize / 2;
    a.OH = (H + 2 * a.pad - ksize) / stride + 1; a.OW = (W + 2 * a.pad - ksize) / stride + 1;
    a.w = w; a.bias = b; a.dst = y; a.ldd = ldy; a.Cout = C2; a.act = act; a.dst_scale = 1;
    if (impl == 1) return launch_conv32_scalar(a, (hipStream_t)stream);
    if (impl == 2) return launch_conv32_mfma(a, (hipStream_t)stream);
    return launch_conv32(a, (hipStream_t)stream);
}

extern "C" int bsy_attention_f32(const float* qkv, int ld, int B, int N, int heads, int key_dim, int head_dim, float scale, float* out,
                                 int ldo, int impl, bsy_stream stream) {
    if (!qkv || !out || B <= 0 || N <= 0 || impl < 0 || impl > 3) BSY_FAIL(BSY_ERR_ARG, "attention_f32: bad argument");
    if (impl == 3) return launch_attn32x(qkv, ld, B, N, heads, key_dim, head_dim, scale, out, ldo, (hipStream_t)stream);
    return launch_attn32(qkv, ld, B, N, heads, key_dim, head_dim, scale, out, ldo, (hipStream_t)stream, impl);
}

extern "C" int bsy_conv_first(const void* img, int img_dtype, int B, int H, int W, const void* w, const float* b,
                              void* y, int ldy, int C2, int ksize, int stride, int act, bsy_stream stream) {
    if (!img || !w || !b || !y) BSY_FAIL(BSY_ERR_ARG, "conv_first: null pointer");
    ConvFirstArgs a;
    a.img = img; a.img_dtype = img_dtype; a.B = B; a.H = H; a.W = W; a.ksize = ksize; a.stride = stride; a.pad = ksize / 2;
    a.OH = (H + 2 * a.pad - ksize) / stride + 1; a.OW = (W + 2 * a.pad - ksize) / stride + 1;
    a.w = w; a.b = b; a.dst = (half_t*)y; a.ldd = ldy; a.Cout = C2; a.act = act;
    return launch_conv_first(a, (hipStream_t)stream);
}

extern "C" int bsy_stem_fused(const void* img, int img_dtype, int B, int H, int W, const void* w0, const float* b0, int C0,
                              const void* w1, const float* b1, int C1, void* y, int ldy, int act, bsy_stream stream) {
    if (!img || !w0 || !b0 || !w1 || !b1 || !y) BSY_FAIL(BSY_ERR_ARG, "stem: null pointer");
    StemArgs a;
    a.img = img; a.img_dtype = img_dtype; a.B = B; a.H = H; a.W = W;
    const int OH0 = (H - 1) / 2 + 1, OW0 = (W - 1) / 2 + 1;
    a.OH = (OH0 - 1) / 2 + 1; a.OW = (OW0 - 1) / 2 + 1;
    a.w0 = w0; a.b0 = b0; a.w1 = w1; a.b1 = b1; a.C0 = C0; a.C1 = C1;
    int cp = 0;
    bsy_conv_packed_dims(C1, C0, 3, &cp, &a.Kpad1);
    a.dst = (half_t*)y; a.ldd = ldy; a.act = act;
    return launch_stem_fused(a, (hipStream_t)stream);
}

extern "C" int bsy_stem_fused_supported(int C0, int C1, int H, int W) { return stem_fused_supported(C0, C1, H, W) ? 1 : 0; }

extern "C" int bsy_bottleneck_fused(const void* x, int ldx, int B, int H, int W, int C, int CH, const void* w1,
                                    const float* b1, const void* w2, const float* b2, void* y, int ldy, int act,
                                    bsy_stream stream) {
    if (!x || !w1 || !b1 || !w2 || !b2 || !y) BSY_FAIL(BSY_ERR_ARG, "bottleneck: null pointer");
    BneckArgs a;
    a.src = (const half_t*)x; a.lds = ldx; a.B = B; a.H = H; a.W = W; a.C = C; a.CH = CH;
    a.w1 = w1; a.b1 = b1; a.w2 = w2; a.b2 = b2;
    int cp = 0;
    if (C <= 0 || CH <= 0) BSY_FAIL(BSY_ERR_ARG, "bottleneck: bad widths");
    bsy_conv_packed_dims(CH, C, 3, &cp, &a.Kpad1);
    bsy_conv_packed_dims(C, CH, 3, &cp, &a.Kpad2);
    a.dst = (half_t*)y; a.ldd = ldy; a.act = act;
    return launch_bneck_fused(a, (hipStream_t)stream);
}

extern "C" int bsy_bottleneck_fused_supported(int C, int CH) { return bneck_fused_supported(C, CH) ? 1 : 0; }

extern "C" int bsy_c3k2_fused(const void* x, int ldx, int B, int H, int W, int Cin, int c, int C2, const void* w1, const float* b1,
                              const void* wa, const float* ba, const void* wb, const float* bb, const void* w4, const float* b4,
                              void* y, int ldy, bsy_stream stream) {
    C3k2Args a;
    a.src = (const half_t*)x; a.lds = ldx; a.B = B; a.H = H; a.W = W; a.Cin = Cin; a.C = c; a.C2 = C2;
    a.w1 = w1; a.b1 = b1; a.wa = wa; a.ba = ba; a.wb = wb; a.bb = bb; a.w4 = w4; a.b4 = b4;
    a.dst = (half_t*)y; a.ldd = ldy;
    return launch_c3k2_fused(a, (hipStream_t)stream);
}

extern "C" int bsy_c3k2_fused_supported(int Cin, int c, int C2) { return c3k2_fused_supported(Cin, c, C2) ? 1 : 0; }

extern "C" int bsy_dwconv(const void* x, int ldx, int B, int H, int W, int C, int kh, int kw, int stride, const float* w,
                          int wld, const float* b, void* y, int ldy, int act, bsy_stream stream) {
    DwGenArgs a;
    a.src = (const half_t*)x; a.lds = ldx; a.B = B; a.H = H; a.W = W; a.C = C; a.kh = kh; a.kw = kw; a.stride = stride;
    if (kh < 1 || kw < 1 || (stride != 1 && stride != 2)) BSY_FAIL(BSY_ERR_ARG, "dwconv: bad kernel / stride");
    a.OH = (H + 2 * (kh / 2) - kh) / stride + 1; a.OW = (W + 2 * (kw / 2) - kw) / stride + 1;
    a.w = w; a.wld = wld; a.b = b; a.dst = (half_t*)y; a.ldd = ldy; a.act_c = act ? C : 0;
    return launch_dwconv_generic(a, (hipStream_t)stream);
}

extern "C" size_t bsy_ela_scratch_bytes(int B, int H, int W, int C) {
    return (B > 0 && H > 0 && W > 0 && C > 0) ? ela_scratch_floats(H, W, C) * (size_t)B * sizeof(float) : 0;
}

extern "C" int bsy_ela(const void* x, int ldx, int B, int H, int W, int C, int k, const float* wsp, const float* wch,
                       const float* gnw, const float* gnb, const float* coef, void* scratch, void* y, int ldy,
                       bsy_stream stream) {
    if (!coef) BSY_FAIL(BSY_ERR_ARG, "ela: null coefficient pointer");
    ElaArgs a;
    a.src = (const half_t*)x; a.lds = ldx; a.B = B; a.H = H; a.W = W; a.C = C; a.k = k;
    a.wsp = wsp; a.wch = wch; a.gnw = gnw; a.gnb = gnb; a.ch_coef = coef[0]; a.sp_coef = coef[1]; a.res_coef = coef[2];
    a.scratch = (float*)scratch; a.dst = (half_t*)y; a.ldd = ldy;
    return launch_ela(a, (hipStream_t)stream);
}

extern "C" int bsy_dwpw_fused(const void* x, int ldx, int B, int H, int W, int C, const float* dww, const float* dwb,
                              const void* w, const float* b, void* y, int ldy, int C2, int act, bsy_stream stream) {
    DwPwArgs a;
    a.src = (const half_t*)x; a.lds = ldx; a.B = B; a.H = H; a.W = W; a.C = C; a.dww = dww; a.dwb = dwb; a.wgt = w; a.bias = b;
    a.dst = (half_t*)y; a.ldd = ldy; a.Cout = C2; a.act = act;
    return launch_dwpw_fused(a, (hipStream_t)stream);
}

extern "C" int bsy_dwpw_fused_supported(int C, int C2) { return dwpw_fused_supported(C, C2) ? 1 : 0; }

extern "C" int bsy_dwconv3x3(const void* x, int ldx, int B, int H, int W, int C, const float* w, const float* b, void* y,
                             int ldy, int act, const void* res, int ldr, bsy_stream stream) {
    if (!x || !w || !b || !y) BSY_FAIL(BSY_ERR_ARG, "dwconv3x3: null pointer");
    DwArgs a;
    a.src = (const half_t*)x; a.lds = ldx; a.B = B; a.H = H; a.W = W; a.C = C; a.w = w; a.b = b;
    a.dst = (half_t*)y; a.ldd = ldy; a.act = act; a.res = (const half_t*)res; a.ldr = ldr;
    return launch_dwconv(a, (hipStream_t)stream);
}

extern "C" int bsy_sppf_pool(void* buf, int ld, int B, int H, int W, int C, bsy_stream stream) {
    if (!buf) BSY_FAIL(BSY_ERR_ARG, "sppf_pool: null pointer");
    return launch_sppf_pool((half_t*)buf, ld, B, H, W, C, (hipStream_t)stream);
}

extern "C" int bsy_attention(const void* qkv, int ld, int B, int N, int heads, int key_dim, int head_dim, float scale,
                             void* out, int ldo, bsy_stream stream) {
    if (!qkv || !out) BSY_FAIL(BSY_ERR_ARG, "attention: null pointer");
    AttnArgs a;
    a.qkv = (const half_t*)qkv; a.ld = ld; a.B = B; a.N = N; a.heads = heads; a.key_dim = key_dim; a.head_dim = head_dim;
    a.scale = scale; a.out = (half_t*)out; a.ldo = ldo;
    return launch_attention(a, (hipStream_t)stream);
}

extern "C" int bsy_detect_decode(const float* const* box, const int* ldb, const float* const* cls, const int* ldc,
                                 const float* const* msk, const int* ldm, const int* lvl_h, const int* lvl_w,
                                 const float* lvl_stride, int nl, int B, int nc, int nm, void* y, int y_dtype,
                                 bsy_stream stream) {
    if (!box || !ldb || !cls || !ldc || !lvl_h || !lvl_w || !lvl_stride || !y || nl < 1 || nl > 3)
        BSY_FAIL(BSY_ERR_ARG, "detect_decode: bad argument");
    if (nm && (!msk || !ldm)) BSY_FAIL(BSY_ERR_ARG, "detect_decode: mask inputs missing");
    DecodeArgs a;
    memset(&a, 0, sizeof(a));
    a.nl = nl; a.B = B; a.nc = nc; a.nm = nm; a.A = 0;
    for (int l = 0; l < nl; ++l) {
        a.box[l] = box[l]; a.cls[l] = cls[l]; a.msk[l] = nm ? msk[l] : nullptr;
        a.ldb[l] = ldb[l]; a.ldc[l] = ldc[l]; a.ldm[l] = nm ? ldm[l] : 0;
        a.h[l] = lvl_h[l]; a.w[l] = lvl_w[l]; a.stride[l] = lvl_stride[l];
    }
    a.y = y; a.y_dtype = y_dtype;
    return launch_decode(a, (hipStream_t)stream);
}
