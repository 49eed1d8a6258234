/*
 * bsyolo.h -- C ABI of libbsyolo_hip.so, the MI355X (gfx950) YOLO detection forward path.
 *
 * Drop-in boundary for feiyeha/BS-YOLO (Ultralytics 8.3.56 fork, pure Python/PyTorch).  The reference has no
 * native code on this path, so there is no existing FFI to mirror; each entry point below names the reference
 * Python interface it replaces (paths relative to /root/reference/ultralytics).  All pointers are raw device
 * pointers unless marked HOST; no torch types cross this boundary.  Every function returns 0 on success or a
 * negative bsy_status; bsy_last_error() returns a thread-local message.  No function synchronises the stream
 * or allocates user-visible memory (engine/plan handles own their weights and workspace).
 *
 * Activation layout inside the library is NHWC fp16 ("pixel rows of channels"); the boundary tensors keep the
 * reference's layouts: input image BCHW (fp16/fp32), prediction (B, 4+nc+nm, A) channel-major, raw feature maps
 * BCHW, detections (B, max_det, 6+nm) row-major.
 */
#ifndef BSYOLO_H
#define BSYOLO_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* bsy_stream; /* hipStream_t */

enum bsy_status {
    BSY_OK = 0,
    BSY_ERR_ARG = -1,     /* bad argument / unsupported shape */
    BSY_ERR_HIP = -2,     /* a HIP runtime call failed        */
    BSY_ERR_ALLOC = -3,   /* device allocation failed         */
    BSY_ERR_STATE = -4    /* call order (e.g. weights not loaded) */
};

enum bsy_dtype { BSY_F16 = 0, BSY_F32 = 1, BSY_U8 = 2 };

/* ---------------------------------------------------------------------------------------------------------
 * Graph engine: replaces BaseModel._predict_once (nn/tasks.py:138-165) + the forward of every module on the
 * path (nn/modules/conv.py:133-151,224-229,445-455; block.py:58-97,3114-3149,3295-3334,3405-3419,3796-3815,
 * 4235-4288,4348-4383,4429-4468; head.py:21-197) for a model already parsed by the host side into a flat
 * list of device ops (bs_yolo_amd/plan.py).  Hook on the reference side: AutoBackend's in-memory nn.Module
 * branch, nn/autobackend.py:136-147 / :524.
 * --------------------------------------------------------------------------------------------------------- */
typedef struct bsy_engine bsy_engine;
typedef struct bsy_plan bsy_plan;

/* A view of `C` channels starting at channel `coff` inside an NHWC buffer whose pixel rows have `ld` channels.
 * buf <  BSY_EXT_BASE : index into the plan's workspace buffers
 * buf >= BSY_EXT_BASE : external pointer slot (buf - BSY_EXT_BASE) supplied to bsy_plan_run
 * buf <  0            : absent                                                                          */
#define BSY_EXT_BASE 0x100000
typedef struct bsy_view {
    int32_t buf, ld, coff, C;
} bsy_view;

enum bsy_op_kind {
    BSY_OP_CONV_FIRST = 0, /* image (BCHW f16/f32, C=3) -> conv kxk s2 + bias + act -> NHWC f16            */
    BSY_OP_CONV = 1,       /* implicit-GEMM conv k in {1,3}, s in {1,2}, MFMA f16, fused bias/SiLU/residual   */
    BSY_OP_DWCONV = 2,     /* depthwise 3x3 s1 + bias (+SiLU) (+residual)                                     */
    BSY_OP_SPPF_POOL = 3,  /* three chained MaxPool2d(5,1,2) -> channel slices of the SPPF concat buffer      */
    BSY_OP_ATTN = 4,       /* softmax(q^T k * scale) v, per (image, head); MFMA flash-style                   */
    BSY_OP_DECODE = 5,     /* Detect._inference: DFL + dist2bbox + sigmoid -> (B, 4+nc+nm, A)                 */
    BSY_OP_RAW_NCHW = 6,   /* raw per-level head maps NHWC f32 -> BCHW (the `x` list Detect.forward returns)  */
    BSY_OP_NHWC2NCHW = 7,  /* NHWC f16 view -> BCHW tensor (Segment protos, block.py:80-97 output)            */
    BSY_OP_STEM = 8,       /* layers 0 + 1 fused: image -> Conv 3x3 s2 (3 -> mid_c) -> Conv 3x3 s2 (mid_c -> dst.C);
                            * w_off/b_off = layer 0, w2_off/b2_off = layer 1; the layer-0 map never reaches HBM   */
    BSY_OP_BNECK = 9,      /* Bottleneck(c, c, shortcut, k=(3,3), e=0.5) fused: dst = src0 + cv2(cv1(src0)), hidden
                            * width mid_c; w_off/b_off = cv1, w2_off/b2_off = cv2                                  */
    /* ---- modules of the BS-YOLO graph (cfg/models/11/yolo11.yaml of the fork; csrc/bsyolo_ops.hip) ---- */
    BSY_OP_DWCONV_G = 10,  /* depthwise ksize (height) x pad (WIDTH; padding is always k/2) conv, stride 1/2, + bias,
                            * SiLU on the first `act` channels: PMSFA 5x5/7x7, SCDown.cv2, MSCAAttention strips.  heads = channels of the whole
                            * weight tensor (row length of the f32 [kh*kw][heads] weights), key_dim = first channel    */
    BSY_OP_COPY = 11,      /* dst slice <- src0 view (through nearest x2 when up0): materialises a Concat operand      */
    BSY_OP_GAP = 12,       /* dst (B,1,1,C) f16 <- mean over H x W of src0                                              */
    BSY_OP_MSCA_MIX = 13,  /* dst = sum_i softmax_i(sigmoid(logit_i)) * branch_i; branches box[0..2] + res, logits (f32
                            * (B,1,1,C) maps) cls[0..2] + msk[0]   (nn/Addmodules/MSCA.py:69-82)                        */
    BSY_OP_MUL = 14,       /* dst = src0 * src1 elementwise                                                             */
    BSY_OP_DWPW = 16,      /* DWConv 3x3 (+SiLU) -> Conv 1x1 (+act) as one launch (YOLO11 class branch, head.py:49-57):
                            * src0 (C channels) -> dst; w_off/b_off = depthwise f32 [9][C] / [C], w2_off/b2_off = 1x1    */
    BSY_OP_MSCA_SPATIAL = 17, /* MSCAAttention's depthwise part in one launch (nn/Addmodules/MSCA.py:53-75): src0 -> the four
                            * strip-conv branch maps box[0..2] + res and their global means cls[0..2] + msk[0] ((B,1,1,C) f16);
                            * aux_off = (w, b) byte offsets of conv0, conv0_1, conv0_2, conv1_1, .. conv3_2 (f32 [taps][C] /
                            * [C], `dilconv` folded into conv{0,1,2}_2).  H * W <= 1890                                  */
    BSY_OP_C3K2 = 18,      /* whole C3k2 block (c3k = False, n = 1; block.py:3796-3804) as one launch: src0 (Cin channels) -> dst
                            * (C2 channels); mid_c = the block's hidden width c; aux_off = (weights, bias) byte offsets of cv1,
                            * m.0.cv1, m.0.cv2, cv2.  Widths (Cin, c, C2) = (64, 32, 128): YOLO11s model.2, YOLO11n model.4        */
    BSY_OP_S2D = 19,       /* space-to-depth of the image for a 6x6 stride-2 pad-2 stem (YOLOv5u): src0 = image (BCHW, in_dtype) -> dst
                            * NHWC f16 (B, H/2, W/2, 16): channel (dy*2+dx)*3 + c, 12..15 zero; an ordinary 3x3 s1 conv follows   */
    BSY_OP_PMSFA_TAIL = 20, /* PMSFA after its conv1, as one launch (block.py:3046-3054; round 4, csrc/pmsfa_fused.hip): src0 = P = conv1's output
                            * [p1 | p2] (C channels), res = the module's input x, dst = conv4(cat(conv3(q1), q2, p2)) + x with
                            * [q1 | q2] = conv2(p1).  aux_off = (weights, bias) byte offsets of conv2 (f32 [25][heads]), conv3 (f32 [49][key_dim])
                            * and conv4 (packed 1x1); heads / key_dim = row lengths of the two depthwise weight tensors.  C in {32, 64, 128};
                            * the same bits as the three launches it replaces                                              */
    BSY_OP_CHAIN = 21,     /* two 1x1 Conv modules chained per pixel as one launch (round 4, csrc/chain1x1.hip; nn/modules/conv.py:149-151 twice):
                            * stage 1 = act(W1 [src0 | src1] + b1) (+ box[2] as shortcut operand), heads = its output channels, w_off / b_off;
                            * box[0] = its HBM output view (buf < 0: not written: nobody else reads it); its channels [key_dim, key_dim + mid_c)
                            * stay in LDS and are the LAST mid_c input channels of stage 2 = act2(W2 [box[1] | kept] + b2) (+ res) -> dst,
                            * w2_off / b2_off, nl = 1 when stage 1 has SiLU, act = stage 2's.  C3k2.cv1 -> C3k.cv1|cv2, C2PSA.cv1 -> qkv,
                            * C3k.cv3 -> C3k2.cv2, ffn[1] -> C2PSA.cv2 (block.py:3796-3815, :4429-4468); the same bits as the two launches    */
    BSY_OP_ELA = 15        /* ELA (nn/Addmodules/ELA.py:77-101): ksize = Conv1d taps; w_off -> f32 blob [spatial_conv C*k]
                            * [ch_att conv C*k][gn.weight C][gn.bias C]; scale, lvl_stride[0], lvl_stride[1] =
                            * sigmoid(ch_weight), sigmoid(sp_weight), sigmoid(res_weight);
                            * res = f32 scratch buffer of (2 (H + W) + 2) * C floats per image                          */
};

typedef struct bsy_op {
    int32_t kind;
    int32_t B, H, W;        /* logical input height/width (after any folded upsample) */
    int32_t OH, OW;
    bsy_view src0, src1;    /* src1: second concat operand (virtual Concat), buf<0 if none */
    int32_t up0, up1;       /* 1: that source is stored at (H/2, W/2) and read through nearest-x2 (virtual Upsample) */
    bsy_view dst;           /* dst.C = Cout */
    bsy_view res;           /* residual added AFTER the activation (Bottleneck / PSABlock shortcut); buf<0 if none */
    int32_t ksize, stride, pad;
    int32_t act;            /* 0 identity, 1 SiLU */
    int32_t out_f32;        /* output mode.  0: dst f16.  1: dst f32 (final head convs feeding BSY_OP_DECODE).
                             * 2 / 3: fused Detect decoder -- the conv is the last layer of a class (2) / box (3) branch
                             * of level `level` and writes rows of the prediction tensor directly: dst = y view
                             * (external slot, C = 4 + nc), nl = the conv's output channels (64 / nc), box[0] = that
                             * level's raw map (external slot, optional),
                             * A = anchors of all levels, lvl_h[1] = first anchor of this level, lvl_stride[0] = stride,
                             * out_dtype = dtype of y and of the raw map */
    int32_t dst_scale, dst_dy, dst_dx; /* ConvTranspose2d(2,2,s2) as 4 scattered 1x1 convs: out pixel (s*oh+dy, s*ow+dx); scale 1 = plain */
    int64_t w_off, b_off;   /* byte offsets into the engine's weight blob (packed f16 weights / f32 bias) */
    int32_t heads, key_dim, head_dim; /* ATTN */
    float scale;            /* ATTN softmax scale */
    /* DECODE / RAW_NCHW: per-level inputs */
    int32_t nl, nc, nm, A;  /* levels, classes, mask coeffs, total anchors */
    bsy_view box[3], cls[3], msk[3];
    int32_t lvl_h[3], lvl_w[3];
    float lvl_stride[3];
    int32_t in_dtype, out_dtype; /* CONV_FIRST input dtype; DECODE/RAW output dtype */
    int32_t level;          /* RAW_NCHW: which level; output = external slot in dst.buf */
    int32_t lane;           /* 0 = caller's stream; k > 0 = plan-owned side stream k (independent op chains, e.g. the
                             * per-level Detect branches, run concurrently; forked from lane 0 at first use) */
    int32_t tuned_cfg;      /* conv: 1 + configuration id recorded by bsy_plan_autotune (0 = heuristic) */
    int32_t join;           /* 1: every side stream is joined back into lane 0 before this op */
    int32_t mid_c;          /* STEM / BNECK: channels of the fused-away intermediate map */
    int64_t w2_off, b2_off; /* STEM / BNECK: second conv's weights / bias (byte offsets into the weight blob) */
    int64_t aux_off[18];    /* MSCA_SPATIAL: (weights, bias) byte offsets of its nine depthwise convs */
    int32_t prec;           /* 0: the fp16-storage product path.  1: fp32 correctness mode -- every workspace view is NHWC f32,
                             * dense conv weights are f32 [k*k*Cin][Cout], arithmetic is fp32 (csrc/ref32.hip); kinds CONV_FIRST,
                             * CONV, DWCONV, DWCONV_G, SPPF_POOL, ATTN, DECODE, RAW_NCHW, NHWC2NCHW, COPY, GAP, MSCA_MIX, MUL, ELA.
                             * 2: fp32x mode -- storage and every non-conv kernel as in mode 1; dense convs (CONV) multiply on the
                             * fp16 matrix pipe with both operands split into f16 pairs (csrc/conv32x_mfma.hip: ~2^-21 relative
                             * operand error instead of the exact fp32 chain); their weight record is the mode-1 f32 matrix followed
                             * by two f16 planes [Cout][K rounded up to 32] (hi, lo), each part padded to 256 bytes */
    int32_t reserved0;
    int32_t ksplit;         /* CONV, latency-mode plans (round 4): 0 = off; else (channel slices) | (tap slices) << 8 of the layer's K walk.  The
                             * slices are computed by separate workgroups of one launch into f32 slabs in the workspace buffer box[0]
                             * (slices x B x OH x OW x round_up(Cout, 32) floats) and summed IN SLICE ORDER by a second launch that applies
                             * activation and shortcut: deterministic, independent of the batch and of the tile configuration.  The host
                             * chooses the factors from the layer's SHAPE only (bs_yolo_amd/plan.py split_factors), never from the batch. */
    int32_t reserved1;
} bsy_op;

int bsy_sizeof_op(void); /* sizeof(bsy_op) of the library: a host binding that mirrors the record (bs_yolo_amd/lib.py) checks it at load */
int bsy_engine_create(int device, bsy_engine** out);
void bsy_engine_destroy(bsy_engine* e);
/* HOST blob: packed weights produced by bs_yolo_amd/weights.py (BN already folded: replaces
 * utils/torch_utils.py:242-269 fuse_conv_and_bn as called from nn/tasks.py:209-215).  Copied to the device.
 * May be called again on an engine that already has plans (same layout, new values: EMA / fine-tuned weights): the call synchronises
 * the device before it frees the old blob, plans resolve weight addresses when they are enqueued, and every graph captured by
 * bsy_plan_graph_launch so far is dropped on its plan's next launch (they hold addresses inside the old blob). */
int bsy_engine_load_weights(bsy_engine* e, const void* host_blob, size_t bytes);

/* ops: HOST array; buf_bytes: HOST array of workspace buffer sizes.  The plan owns its workspace. */
int bsy_plan_create(bsy_engine* e, const bsy_op* ops, int n_ops, const int64_t* buf_bytes, int n_bufs, bsy_plan** out);
/* The same with the buffers at HOST-assigned byte offsets buf_off[] (multiples of 256, may overlap) inside ONE activation
 * arena owned by the ENGINE and shared by all of its plans: bs_yolo_amd/plan.py assigns the offsets from buffer liveness, so a
 * forward needs the peak of its live activations instead of their sum, and 40 input shapes (val rect batches,
 * data/base.py:261-284; predict's `auto` letterbox) need the arena of the largest one instead of 40 workspaces.  The arena
 * grows (after a device synchronisation) when a plan needs more.  Plans of one engine must not run concurrently. */
int bsy_plan_create_arena(bsy_engine* e, const bsy_op* ops, int n_ops, const int64_t* buf_bytes, const int64_t* buf_off, int n_bufs,
                          int64_t arena_bytes, bsy_plan** out);
size_t bsy_engine_arena_bytes(const bsy_engine* e);
void bsy_plan_destroy(bsy_plan* p);
/* ext: HOST array of n_ext device pointers bound to the external slots (input image, y, raw maps ...). */
int bsy_plan_run(bsy_plan* p, void* const* ext, int n_ext, bsy_stream stream);
/* The same forward as ONE graph launch: the first call with a given set of external pointers captures bsy_plan_run on `stream`
 * (thread-local capture; side lanes become graph edges) and instantiates it, later calls with the same pointers replay it -- one
 * submission instead of 70-odd launches, which is what bounds a forward at the 8- or 16-image share of a strong-scaled batch.
 * Up to 8 graphs per plan (least recently used dropped); all are dropped when the arena moves, when bsy_engine_load_weights replaces
 * the weight blob, or when the plan's tuning changes.  `stream` must not
 * be the null stream (falls back to bsy_plan_run, as it does for good if a capture ever fails).  *captured (optional): 1 captured
 * now, 0 replayed, -1 ran eagerly.  Results are those of bsy_plan_run bit for bit (the same launches). */
int bsy_plan_graph_launch(bsy_plan* p, void* const* ext, int n_ext, bsy_stream stream, int* captured);
/* Times every valid kernel configuration of every conv op once (HIP events on `stream`, synchronises) and records the
 * fastest per op; later bsy_plan_run calls use it.  Results are bit-identical across configurations: the order in which a conv's
 * products are summed (its K walk) is a function of the layer's SHAPE (3x3 convs with Cin % 32 == 0 sum chunk-major over 32-channel
 * chunks -- 64-channel chunks for stride 2 with Cin % 64 == 0 --, everything else in the packed (kh, kw, cin) order), and every
 * configuration that is valid for a layer walks K that way; ids recorded for another walk (tune caches written before round 3)
 * are rejected as invalid, i.e. ignored.  A tuned plan therefore returns the bits of an untuned one, on every box. */
int bsy_plan_autotune(bsy_plan* p, void* const* ext, int n_ext, bsy_stream stream);
/* HOST array cfg[n_ops]: cfg[i] >= 0 presets conv op i to that configuration (bsy_plan_autotune then skips it: results of an
 * earlier plan with the same conv shape are reused), -1 leaves the op as it is, -2 clears it. */
int bsy_plan_set_tuning(bsy_plan* p, const int32_t* cfg, int n_ops);
/* HOST out[n_ops]: configuration id per op (tile << 4 | variant), -1 for non-conv / untuned ops. */
int bsy_plan_get_tuning(bsy_plan* p, int32_t* out, int n_ops);
/* HOST array valid[n_ops]: 1 = the configuration preset for op i can run the op's shape, 0 = it cannot (bsy_plan_run would fall back to
 * the heuristic configuration -- an id from a stale or foreign tune cache), -1 = not a conv op or not preset.  Launches nothing; `ext` as
 * for bsy_plan_run.  bs_yolo_amd/engine.py drops the entries reported 0 from its tune cache before timing. */
int bsy_plan_check_tuning(bsy_plan* p, void* const* ext, int n_ext, int32_t* valid, int n_ops);
/* HOST array out[n_ops]: the rank-th runner-up (1 or 2) of the last bsy_plan_autotune per op (-1: none within 12 % of the winner).  The
 * host side re-times winner and runners-up in place with bsy_plan_profile and keeps the fastest (bs_yolo_amd/engine.py). */
int bsy_plan_get_tuning_alt(bsy_plan* p, int rank, int32_t* out, int n_ops);
/* The same decision taken in place: candidate k of every untuned conv op runs in pass k of the whole forward (serial, events around every op,
 * `rounds` passes per k), the fastest configuration per op is recorded.  Costs max(#candidates) x rounds forwards; what bs_yolo_amd/engine.py
 * uses by default (BSY_TUNE_IN_PLACE=0: bsy_plan_autotune). */
int bsy_plan_autotune_in_place(bsy_plan* p, void* const* ext, int n_ext, bsy_stream stream, int rounds);
/* Test aid: plans created with BSY_PLAN_GUARD=<bytes> in the environment keep a guard band of that many bytes (0xA5) behind
 * every workspace buffer; this call synchronises the device and reports the first buffer whose band was written (an
 * out-of-bounds store), *bad_buf = -1 when all are intact. */
int bsy_plan_check_guards(bsy_plan* p, int32_t* bad_buf, int64_t* bad_off);
/* Debug/test aid: synchronous copy of one workspace buffer to HOST memory (bytes <= the buffer's size). */
int bsy_plan_copy_buffer(bsy_plan* p, int buf, void* host_dst, size_t bytes);
/* Per-op device time (ms): runs the plan once with a HIP event recorded after every op on `stream`, syncs.  The cost of
 * the event packet itself (median of 16 empty intervals recorded in the same call, a few microseconds) is subtracted
 * from every op, so that the figures agree with rocprofv3's kernel durations. */
int bsy_plan_profile(bsy_plan* p, void* const* ext, int n_ext, bsy_stream stream, float* ms_per_op /* n_ops */);

/* ---------------------------------------------------------------------------------------------------------
 * Stand-alone operators (same kernels the engine launches), for per-module hooks and parity tests.
 * --------------------------------------------------------------------------------------------------------- */

/* Conv.forward_fuse (nn/modules/conv.py:149-151): y = act(conv2d(x, w) + b) [+ res], NHWC f16 in/out.
 * x: (B,H,W,ldx) view of C1 channels; w: packed [CoutPad][Kpad] f16 (bsy_pack_conv_weight layout);
 * b: f32 [CoutPad]; y: (B,OH,OW,ldy); res same geometry as y or NULL. */
int bsy_conv2d(const void* x, int ldx, int B, int H, int W, int C1, const void* w, const float* b, void* y, int ldy,
               int C2, int ksize, int stride, int act, const void* res, int ldr, int y_f32, bsy_stream stream);
/* The same module in fp32 (the engine's fp32 mode: callers of predict() with the reference's default half=False,
 * cfg/default.yaml:54, engine/predictor.py:131): NHWC f32 in/out, w f32 [k*k*C1][C2] with K order (kh, kw, cin), b f32 [C2].
 * impl 0: the engine's routing (fp32 MFMA kernel, v_mfma_f32_32x32x2_f32, where C1 % 8 == 0, C2 % 4 == 0, strides % 4 == 0 and
 * 16-byte aligned views; else the scalar kernel); 1: scalar kernel (one thread per output, sequential fmaf chain); 2: MFMA kernel
 * (BSY_ERR_ARG where it does not apply).  Both kernels evaluate the same k-ordered fmaf chain: bit-identical results. */
int bsy_conv2d_f32(const float* x, int ldx, int B, int H, int W, int C1, const float* w, const float* b, float* y, int ldy,
                   int C2, int ksize, int stride, int act, const float* res, int ldr, int impl, bsy_stream stream);
/* The same module in the fp32x mode (fp32 storage, split-f16 arithmetic; what plugin.accelerate gives fp32 callers by default since
 * round 4): w_hi / w_lo f16 [C2][k_pad] with w = hi + lo, K order (kh, kw, cin), k_pad = k*k*C1 rounded up to 32 with zeros.
 * Needs C1 % 8 == 0, C2 % 4 == 0, row strides % 4 == 0, 16-byte aligned views (BSY_ERR_ARG otherwise: the engine then runs the
 * exact kernel).  |y - exact| <~ 2^-20 * sum |w x|: three v_mfma_f32_32x32x16_f16 per product, f32 accumulation. */
int bsy_conv2d_f32x(const float* x, int ldx, int B, int H, int W, int C1, const void* w_hi, const void* w_lo, int k_pad,
                    const float* b, float* y, int ldy, int C2, int ksize, int stride, int act, const float* res, int ldr,
                    bsy_stream stream);
/* The image conv of the fp32 mode: img BCHW (f16 / f32) -> NHWC f32; w f32 [k*k*3][C2], K order (kh, kw, c).  impl as above (the
 * MFMA kernel widens every tap to 4 k values, one of them zero, two taps per 8-k piece, K = 8 ceil(k^2 / 2): the same chain, the same bits). */
int bsy_conv_first_f32(const void* img, int img_dtype, int B, int H, int W, const float* w, const float* b, float* y, int ldy,
                       int C2, int ksize, int stride, int act, int impl, bsy_stream stream);
/* Attention (block.py:4279-4286) of the fp32 mode: qkv (B, N, ld) f32 rows [q (heads x key_dim) | k | v (heads x head_dim)] ->
 * out (B, N, ldo) f32, one thread per query, keys in order, online softmax.  impl 0: the LDS-tiled kernel for key_dim 32 /
 * head_dim 64 (every YOLO11 C2PSA), the generic one otherwise; 1: generic; 2: tiled (BSY_ERR_ARG where it does not apply).
 * Same operations in the same order: bit-identical.  impl 3: the fp32x mode's kernel (csrc/attention32x.hip: flash-style on the fp16
 * matrix pipe, q / k / v / p split into f16 pairs, three MFMAs per product; key_dim 32 / head_dim 64 only) -- fp32-class accuracy
 * (~1e-6 of the output range), not the bits of impl 0-2. */
int bsy_attention_f32(const float* qkv, int ld, int B, int N, int heads, int key_dim, int head_dim, float scale, float* out,
                      int ldo, int impl, bsy_stream stream);
/* Packed sizes for a (C2, C1, k, k) conv: rows padded to 128 output channels, K = k*k*C1 padded to 32.
 * C1 == 3 (the image conv) packs a zero 4th input channel: K = k*k*4. */
int bsy_conv_packed_dims(int C2, int C1, int ksize, int* cout_pad, int* k_pad);

/* First conv (3x3, stride 2) from a BCHW image (f16 or f32): w packed like bsy_conv2d's with a zero 4th input
 * channel ([CoutPad][64] f16, k = (kh, kw, c4)), b f32 [CoutPad] -> NHWC f16. */
int bsy_conv_first(const void* img, int img_dtype, int B, int H, int W, const void* w, const float* b, void* y,
                   int ldy, int C2, int ksize, int stride, int act, bsy_stream stream);

/* Layers 0 + 1 of the stock graphs as ONE launch (cfg/models/11/yolo11.yaml:17-18, two Conv.forward_fuse calls,
 * conv.py:149-151): img BCHW (f16 / f32) -> Conv(3, C0, 3, 2) -> Conv(C0, C1, 3, 2) -> NHWC f16 (B, OH1, OW1, ldy).
 * w0/b0 as bsy_conv_first, w1/b1 as bsy_conv2d.  (C0, C1) in {(32, 64), (16, 32)}, W % 4 == 0; anything else is
 * BSY_ERR_ARG (callers then run the two layers separately).  Bit-identical to bsy_conv_first + bsy_conv2d. */
int bsy_stem_fused(const void* img, int img_dtype, int B, int H, int W, const void* w0, const float* b0, int C0,
                   const void* w1, const float* b1, int C1, void* y, int ldy, int act, bsy_stream stream);
/* 1 when bsy_stem_fused accepts this shape. */
int bsy_stem_fused_supported(int C0, int C1, int H, int W);

/* Bottleneck.forward (block.py:3417-3419) with k = (3, 3), shortcut, g = 1: y = x + cv2(cv1(x)), cv1: C -> CH, cv2: CH -> C,
 * both folded Conv+BN+SiLU, as ONE launch with the hidden map kept in LDS.  x: (B,H,W,ldx) view of C channels,
 * y: (B,H,W,ldy); w1/b1, w2/b2 as bsy_conv2d.  (C, CH) = (32, 16) only; anything else is BSY_ERR_ARG (callers then
 * run two bsy_conv2d).  Bit-identical to the two-launch path. */
int bsy_bottleneck_fused(const void* x, int ldx, int B, int H, int W, int C, int CH, const void* w1, const float* b1,
                         const void* w2, const float* b2, void* y, int ldy, int act, bsy_stream stream);
int bsy_bottleneck_fused_supported(int C, int CH);
/* Whole C3k2 block (c3k = False, one Bottleneck; block.py:3796-3804 / :3308-3312 / :3405-3419) in one launch:
 * y = cv2(cat(y0, y1, y1 + m.cv2(m.cv1(y1)))) with [y0 | y1] = cv1(x); x (B,H,W,ldx) NHWC f16 view of Cin channels, y (B,H,W,ldy)
 * of C2 channels; weights packed as for bsy_conv2d (cv1: Cin -> 2c 1x1; m.cv1: c -> c/2 3x3; m.cv2: c/2 -> c 3x3; cv2: 3c -> C2 1x1).
 * bsy_c3k2_fused_supported(Cin, c, C2) tells which widths the kernel is instantiated for. */
int bsy_c3k2_fused(const void* x, int ldx, int B, int H, int W, int Cin, int c, int C2, const void* w1, const float* b1, const void* wa,
                   const float* ba, const void* wb, const float* bb, const void* w4, const float* b4, void* y, int ldy, bsy_stream stream);
int bsy_c3k2_fused_supported(int Cin, int c, int C2);

/* Modules of the BS-YOLO graph (csrc/bsyolo_ops.hip), NHWC f16 views.
 * bsy_dwconv: depthwise kh x kw conv (odd sizes <= 31), stride 1 / 2, padding k/2, + bias (+SiLU when act);
 *   w f32 [kh*kw][wld] offset to the first channel of this call, b f32 [C].
 * bsy_ela: ELA.forward (nn/Addmodules/ELA.py:77-101); wsp/wch (C,k) Conv1d weights, gnw/gnb GroupNorm affine,
 *   coef[3] HOST floats = sigmoid(ch_weight), sigmoid(sp_weight), sigmoid(res_weight); scratch: device f32,
 *   bsy_ela_scratch_bytes(B, H, W, C) bytes. */
int bsy_dwconv(const void* x, int ldx, int B, int H, int W, int C, int kh, int kw, int stride, const float* w, int wld,
               const float* b, void* y, int ldy, int act, bsy_stream stream);
size_t bsy_ela_scratch_bytes(int B, int H, int W, int C);
int bsy_ela(const void* x, int ldx, int B, int H, int W, int C, int k, const float* wsp, const float* wch,
            const float* gnw, const float* gnb, const float* coef, void* scratch, void* y, int ldy, bsy_stream stream);

/* nn.Sequential(DWConv(c, c, 3), Conv(c, c2, 1)) (head.py:49-57) as ONE launch: depthwise 3x3 + SiLU computed per tile
 * on the VALU, fed straight to the 1x1 conv's MFMAs.  dww f32 [9][C], dwb f32 [C] as bsy_dwconv3x3; w / b as bsy_conv2d.
 * C % 32 == 0, C <= 256, C2 % 8 == 0; anything else is BSY_ERR_ARG.  Bit-identical to bsy_dwconv3x3 + bsy_conv2d. */
int bsy_dwpw_fused(const void* x, int ldx, int B, int H, int W, int C, const float* dww, const float* dwb, const void* w,
                   const float* b, void* y, int ldy, int C2, int act, bsy_stream stream);
int bsy_dwpw_fused_supported(int C, int C2);

/* DWConv (conv.py:224-229) 3x3 s1: w f32 [9][C], b f32 [C]. */
int bsy_dwconv3x3(const void* x, int ldx, int B, int H, int W, int C, const float* w, const float* b, void* y, int ldy,
                  int act, const void* res, int ldr, bsy_stream stream);

/* SPPF pooling (block.py:3145-3149): x1 = buf[..., 0:C]; writes m(x1), m(m(x1)), m(m(m(x1))) to channel slices
 * [C:2C], [2C:3C], [3C:4C] of the same NHWC buffer (ld >= 4C). */
int bsy_sppf_pool(void* buf, int ld, int B, int H, int W, int C, bsy_stream stream);

/* Attention core (block.py:4267-4286): qkv NHWC f16 with channel order [q(all heads) | k(all heads) | v(all heads)],
 * out[pixel][head*head_dim + d] = sum_j softmax_j(q_i . k_j * scale) v_j[d].  key_dim in {16, 32, 48, 64}, head_dim in {32, 64, 96, 128}
 * (the stock scales: 32 / 64; other width multiples give e.g. one head of 48 / 96, block.py:4253-4258); BSY_ERR_ARG otherwise. */
int bsy_attention(const void* qkv, int ld, int B, int N, int heads, int key_dim, int head_dim, float scale, void* out,
                  int ldo, bsy_stream stream);

/* Detect._inference (head.py:100-131) + DFL (block.py:58-77) + make_anchors/dist2bbox (utils/tal.py:371-395).
 * box[l]: f32 (B*h*w, ldb) 64 DFL logits; cls[l]: f32 (B*h*w, ldc) nc logits; msk[l]: f32 nm coeffs or NULL.
 * y: (B, 4+nc+nm, A) of y_dtype. */
int bsy_detect_decode(const float* const* box, const int* ldb, const float* const* cls, const int* ldc,
                      const float* const* msk, const int* ldm, const int* lvl_h, const int* lvl_w,
                      const float* lvl_stride, int nl, int B, int nc, int nm, void* y, int y_dtype, bsy_stream stream);

/* ---------------------------------------------------------------------------------------------------------
 * non_max_suppression (utils/ops.py:167-316) incl. xywh2xyxy (:416-433) and the greedy IoU suppression the
 * reference delegates to torchvision.ops.nms (:296).  One call handles the whole batch with no host sync.
 *   pred      : (B, 4+nc+nm, A) f16/f32, xywh boxes; when in_place != 0 channels 0..3 are rewritten as xyxy
 *               exactly as the reference mutates its input (:243-244)
 *   classes   : optional DEVICE int32 list of class ids to keep (:278-279), n_classes = 0 -> all
 *   out       : (B, max_det, 6+nm) f32 rows [x1,y1,x2,y2,conf,cls,masks...], rows >= counts[b] are zero
 *   counts    : (B) int32 kept detections per image
 *   workspace : DEVICE scratch of at least bsy_nms_workspace_bytes(...) bytes
 * Arithmetic is fp32 whatever the input dtype (matches the fp32 CPU reference); ties in score are broken by
 * ascending candidate index.  The reference's wall-clock bail-out (:238,:312-314) is deliberately absent.
 * --------------------------------------------------------------------------------------------------------- */
size_t bsy_nms_workspace_bytes(int B, int A, int nc, int multi_label, int max_nms);
int bsy_nms(void* pred, int pred_dtype, int B, int nc, int nm, int A, float conf_thres, float iou_thres,
            const int32_t* classes, int n_classes, int agnostic, int multi_label, int max_det, int max_nms,
            float max_wh, int in_place, float* out, int32_t* counts, void* workspace, size_t workspace_bytes,
            bsy_stream stream);

/* scale_boxes + clip_boxes (utils/ops.py:92-127, :319-337) applied to all rows of `out` in place:
 * per image b: gain[b], pad_x[b], pad_y[b] (HOST-computed with Python round()), orig shape (h0[b], w0[b]). */
int bsy_scale_boxes(float* det, const int32_t* counts, int B, int max_det, int row, const float* gain,
                    const float* pad_x, const float* pad_y, const float* h0, const float* w0, bsy_stream stream);

/* ---------------------------------------------------------------------------------------------------------
 * LetterBox + BasePredictor.preprocess (data/augment.py:1535-1601, engine/predictor.py:116-134): device images
 * (HWC BGR u8, each with its own (h,w)) -> one (B,3,H2,W2) RGB tensor scaled by 1/255.
 *   imgs      : DEVICE array of B device pointers;  hw: DEVICE int32 (B,2)
 *   geom      : DEVICE int32 (B,4) = new_unpad_w, new_unpad_h, left, top  (HOST-computed with Python round())
 * Bilinear arithmetic is OpenCV's 8-bit fixed point (INTER_LINEAR) incl. the exact-2x box shortcut.
 * --------------------------------------------------------------------------------------------------------- */
int bsy_letterbox(const uint8_t* const* imgs, const int32_t* hw, const int32_t* geom, int B, int H2, int W2, void* out,
                  int out_dtype, bsy_stream stream);

/* ---------------------------------------------------------------------------------------------------------
 * process_mask + crop_mask (utils/ops.py:663-694, :644-660) for ONE image: protos (nm, mh, mw) f16/f32 BCHW slice,
 * coef (n, >=nm) f32 rows of stride ldc (the mask columns of the NMS output), boxes (n, >=4) f32 xyxy in input-image
 * pixels (stride ldb), lowres = n*mh*mw f32 scratch.  out = (n, ih, iw) when upsample else (n, mh, mw), 0/1 as
 * u8 or f32 (the reference's `masks.gt_(0.0)` yields a float tensor).
 * --------------------------------------------------------------------------------------------------------- */
int bsy_process_mask(const void* protos, int proto_dtype, int nm, int mh, int mw, const float* coef, int ldc,
                     const float* boxes, int ldb, int n, int ih, int iw, int upsample, float* lowres, void* out,
                     int out_dtype, bsy_stream stream);

/* process_mask_native (utils/ops.py:696-709; segment/predict.py:48-50 with retina_masks) for ONE image: the (n, mh, mw) masks
 * coef @ protos, their window [top:bottom, left:right] (scale_masks' letterbox-padding cut, computed by the HOST with the
 * reference's Python arithmetic) resized bilinearly (align_corners=False) to (oh, ow) = the ORIGINAL image, cropped to `boxes`
 * (xyxy in original-image pixels), thresholded at 0 -> out (n, oh, ow) u8 / f32.  lowres = n*mh*mw f32 scratch. */
int bsy_process_mask_native(const void* protos, int proto_dtype, int nm, int mh, int mw, const float* coef, int ldc,
                            const float* boxes, int ldb, int n, int top, int left, int bottom, int right, int oh, int ow,
                            float* lowres, void* out, int out_dtype, bsy_stream stream);
/* scale_masks (utils/ops.py:712-737): masks (n, mh, mw) f16 / f32 -> (n, oh, ow) of the same dtype, window as above. */
int bsy_scale_masks(const void* masks, int dtype, int n, int mh, int mw, int top, int left, int bottom, int right, int oh, int ow,
                    void* out, bsy_stream stream);

/* Validator matching (engine/validator.py:222-258 match_predictions on utils/metrics.py:52-70 box_iou, as called by
 * DetectionValidator._process_batch, models/yolo/detect/val.py:209-228), whole batch at once.
 * det (B, max_det, row >= 6) f32 rows [x1 y1 x2 y2 conf cls ...] + counts (B): the layout bsy_nms writes;
 * gt_boxes (B, Lmax, 4) f32 xyxy, gt_cls (B, Lmax) f32, gt_counts (B); iouv: HOST array of n_iou <= 16 thresholds;
 * out (B, max_det, n_iou) uint8: 1 = true positive at that threshold.  max_det <= 1024, Lmax <= 8192. */
int bsy_val_match(const float* det, int row, const int32_t* counts, int B, int max_det, const float* gt_boxes,
                  const float* gt_cls, const int32_t* gt_counts, int Lmax, const float* iouv, int n_iou,
                  unsigned char* out, bsy_stream stream);

/* ap_per_class + compute_ap (utils/metrics.py:620-706, :588-617) without the plots and the max-F1 pick (host side).
 * tp (N, T) uint8, conf (N) f32 in [0, 1], pred_cls (N) f32 -- the concatenated statistics DetectionValidator.get_stats
 * (models/yolo/detect/val.py:160-170) hands to DetMetrics.process; classes / nt: DEVICE (nc) = np.unique(target_cls,
 * return_counts=True); x101 / x1000: DEVICE np.linspace(0, 1, 101 / 1000).  Outputs (float64, DEVICE): ap (nc, T),
 * p_curve / r_curve / prec_values (nc, 1000), n_pred (nc) int32 = detections of each class (rows of classes with
 * n_pred == 0 stay zero; the reference appends a prec_values row only for classes with predictions).
 * N <= 2^20, T <= 16, class ids 0 .. 4094.  Ties in conf: lower index first (np.argsort's order there is unpinned). */
size_t bsy_ap_workspace_bytes(int N, int T);
int bsy_ap_per_class(const uint8_t* tp, const float* conf, const float* pred_cls, int N, int T, const int32_t* classes,
                     const int32_t* nt, int nc, const double* x101, const double* x1000, double eps, double* ap, double* p_curve,
                     double* r_curve, double* prec_values, int32_t* n_pred, void* workspace, size_t workspace_bytes,
                     bsy_stream stream);

/* ---------------------------------------------------------------------------------------------------------
 * Sliced inference.  The reference reaches it through the un-vendored `sahi` package (detect-sahi.py:1-13
 * sahi.predict.predict; examples/YOLOv8-SAHI-Inference-Video/yolov8_sahi.py:70-75 get_sliced_prediction); the two
 * calls below replace sahi.slicing.slice_image + the predictor's preprocess of each crop, and
 * sahi.postprocess.combine.{GreedyNMMPostprocess, NMSPostprocess} (algorithm of sahi 0.11.x, see oracle/sahi_ref.py).
 *
 * bsy_slice_tiles: img (H, W, 3) u8 DEVICE, rows `pitch` bytes apart, 4-byte aligned; boxes DEVICE int32 (T, 4) =
 *   x0, y0, x1, y1 with x1 - x0 == tw, y1 - y0 == th (get_slice_bboxes keeps border slices full-size by shifting them
 *   inwards; an image smaller than the slice goes through bsy_letterbox instead);  out (T, 3, th, tw) f16/f32 = pixel
 *   / 255, channels reversed when swap_rb != 0.  tw % 4 == 0.
 * bsy_sahi_merge: det (T, max_det, row >= 6) f32 rows [x1 y1 x2 y2 score cls ...] in TILE pixels + counts (T): the
 *   layout bsy_nms writes;  shift (T, 2) f32 DEVICE = tile origin (x0, y0);  boxes are clamped to >= 0 and to
 *   (full_w, full_h) when those are > 0, dropped unless x1 < x2 and y1 < y2, then shifted.
 *   match_metric 0 = IOU, 1 = IOS; a lower-scored box joins the first kept box (score order, same class unless
 *   class_agnostic) with metric >= match_threshold; do_merge == 1 (GREEDYNMM) folds members whose metric against the
 *   growing merged box is > match_threshold into it (box union, max score), do_merge == 0 (NMS / LSNMS) only drops them;
 *   do_merge == 2 (NMM) assigns members the non-greedy way instead -- every box, in score order, hands the unassigned
 *   boxes it matches to its own keeper -- and folds them in the order they were assigned.
 *   out (max_out, 6) f32 in class-ascending (unless agnostic), score-descending keep order; out_count DEVICE int32 =
 *   min(kept, max_out).  T * max_det <= 65536.  No host synchronisation.
 * --------------------------------------------------------------------------------------------------------- */
int bsy_slice_tiles(const uint8_t* img, int H, int W, int pitch, const int32_t* boxes, int T, int th, int tw,
                    int swap_rb, void* out, int out_dtype, bsy_stream stream);
size_t bsy_sahi_merge_workspace_bytes(int T, int max_det);
int bsy_sahi_merge(const float* det, const int32_t* counts, const float* shift, int T, int max_det, int row,
                   int match_metric, float match_threshold, int class_agnostic, int do_merge, float full_w,
                   float full_h, float* out, int32_t* out_count, int max_out, void* workspace, size_t workspace_bytes,
                   bsy_stream stream);

const char* bsy_last_error(void);
int bsy_version(void);

#ifdef __cplusplus
}
#endif
#endif /* BSYOLO_H */
