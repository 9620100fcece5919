/*
 * i8ie_hip.h -- C-ABI of libi8ie_hip.so: the MI355X (gfx950) implementation of
 * the INT8 hot path of t0037799/INT8InferenceEngine.
 *
 * This is the boundary the rebuilt pybind11 module `_CXX_i8ie` (and any other
 * FFI: ctypes, cgo, JNI) binds.  Plain pointers and sizes only; no C++ types,
 * no torch types, no exceptions cross it.  Every entry point cites the
 * reference interface it replaces (paths relative to the reference repo).
 *
 * Conventions
 *   - every function returns an int status: I8IE_OK (0) or a negative code;
 *     i8ie_last_error() returns a thread-local message for the last failure.
 *     (The reference throws message-less std::exception or segfaults on
 *     misuse: include/tensor.h:114-130, src/conv2d.cc:105.)
 *   - "dev" pointers are device (HBM) pointers, obtained from i8ie_malloc or
 *     from any other HIP allocation in the same process (e.g. a torch tensor's
 *     data_ptr()).  Ops enqueue on the ctx's HIP stream and return without
 *     waiting; i8ie_sync() or a D2H copy waits.
 *   - tensors use the reference's layouts: activations u8 NCHW / [m,k]
 *     row-major, weights s8 [out, in*kh*kw] row-major (K ordered c,kh,kw),
 *     per-tensor fp32 scale + u8 zero point (include/tensor.h:152-154).
 *   - arithmetic is the reference's, bit for bit: exact int32 accumulation,
 *     fp32 epilogue without contraction or reassociation.
 */
#ifndef I8IE_HIP_H
#define I8IE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define I8IE_OK 0
#define I8IE_ERR_ARG (-1)   /* bad argument (null pointer, non-positive size, stride 0 ...) */
#define I8IE_ERR_STATE (-2) /* call made in the wrong state (e.g. forward before qparams set) */
#define I8IE_ERR_HIP (-3)   /* a HIP runtime call failed; message has hipGetErrorString */
#define I8IE_ERR_OOM (-4)   /* device allocation failed */

typedef struct i8ie_ctx i8ie_ctx;     /* one per (thread-of-control, device): stream + workspace */
typedef struct i8ie_layer i8ie_layer; /* a converted (INT8) Linear or Conv2d with device-resident weights */
/* Threading: a ctx and the layer handles created on it belong to ONE host thread at a time.  A layer handle caches
 * state keyed by the call (offset vectors per (s_in, zp_in), weight panels re-packed per kernel shape) without locks:
 * do not forward the same handle from two host threads at once; one ctx + its own handles per thread is the model. */

/* ---- library / context ------------------------------------------------- */
const char* i8ie_last_error(void);
int i8ie_version(void);                 /* ABI version, currently 1 */
int i8ie_device_count(int* n);
int i8ie_ctx_create(int device, i8ie_ctx** out);                       /* own stream */
int i8ie_ctx_create_on_stream(int device, void* hip_stream, i8ie_ctx** out); /* borrow a hipStream_t */
int i8ie_ctx_destroy(i8ie_ctx* ctx);
void* i8ie_ctx_stream(i8ie_ctx* ctx);   /* the hipStream_t, as void* */
int i8ie_sync(i8ie_ctx* ctx);

/* Options.  I8IE_OPT_FORCE_FALLBACK = 1 makes every layer handle take the any-geometry
 * path (materialised im2col + v1 GEMM, separate ReLU) instead of the implicit-GEMM paths;
 * results are identical, it exists so that tests can cover both. */
#define I8IE_OPT_FORCE_FALLBACK 1
/* I8IE_OPT_KERNEL_VARIANT = 2 selects among compiled variants of the contraction kernel (all
 * produce identical bytes; a tuning / A-B timing aid, 0 = default). */
#define I8IE_OPT_KERNEL_VARIANT 2
/* The values the product library understands (anything else behaves like 0; the diagnostic build of tools/diag
 * adds timing experiments under further numbers, listed in tools/README.md).  A value changes the one thing it
 * names: 3 and 5 also keep the tiled conv at 128 x 128 tiles (0 picks 192 x 128 for large launches), the others
 * leave the tiled kernel's own defaults alone. */
#define I8IE_VARIANT_AUTO 0            /* automatic selection per launch */
#define I8IE_VARIANT_IGEMM_REGSTAGE 3  /* tiled contraction kernel, one LDS stage filled through registers (Linear's default) */
#define I8IE_VARIANT_IGEMM_DMA 5       /* tiled contraction kernel, one LDS stage filled by LDS-DMA (the tiled conv default) */
#define I8IE_VARIANT_TILED 11          /* tiled contraction kernel everywhere: no patch-stationary conv, split-K Linear, dot4 head */
#define I8IE_VARIANT_STEM_WHOLE 12     /* first-stage kernel (csrc/i8ie_stem.hip): whole images per block at any batch size (no parts) */
#define I8IE_VARIANT_STEM_SIMD_ROLES 13 /* the same with its two wave roles on separate SIMDs (A/B of the placement: 8 % slower, profiles/r04_stem_roles.txt) */
#define I8IE_VARIANT_PCONV 50          /* patch-stationary conv kernel (csrc/i8ie_pconv.hip) at any batch size */
#define I8IE_VARIANT_PCONV_2PASS 54    /* the same, N = 384 as two passes of 192 and no 128-wide pass split */
#define I8IE_VARIANT_TCONV 70          /* two-team patch-stationary conv kernel (csrc/i8ie_tconv.hip) wherever its shape rules allow */
#define I8IE_VARIANT_FLIN 80           /* few-row Linear kernel (csrc/i8ie_flin.hip) below its automatic feature threshold */
#define I8IE_VARIANT_MLIN 83           /* many-row Linear kernel (csrc/i8ie_mlin.hip) from 257 rows on and below its automatic feature threshold */
#define I8IE_VARIANT_MLIN_64 84        /* many-row Linear kernel with 64-row block tiles (automatic where 128-row tiles give at most half the CUs a block) */
#define I8IE_VARIANT_MLIN_128 85       /* many-row Linear kernel with 128-row block tiles at any row count above 256 */
#define I8IE_VARIANT_FLIN_128 81       /* few-row Linear kernel in its 128-row x 16-feature form at up to 128 rows (default above 64 rows: 64 x 32) */
/* I8IE_OPT_PROFILE_STRIDE = 3: while profiling, bracket only every value-th eligible launch
 * (default 1 = all).  Event packets cost a few microseconds each on the stream; a stride that is
 * coprime with the launches per batch samples every kernel over a few batches. */
#define I8IE_OPT_PROFILE_STRIDE 3
/* I8IE_OPT_CU_LIMIT = 4: the compute units the ctx's stream may use, for a ctx created on a stream with a CU mask
 * (hipExtStreamCreateWithCUMask + i8ie_ctx_create_on_stream): one-block-per-CU kernels size their grids and their
 * work splits by min(device CUs, value).  0 (default) = all of the device's. */
#define I8IE_OPT_CU_LIMIT 4
int i8ie_ctx_set_option(i8ie_ctx* ctx, int option, int value);

/* Activation layouts accepted by the *_fused / *_nhwc entry points.  NCHW is the
 * reference's layout (include/tensor.h); NHWC is the engine's internal layout between
 * layers (the reference's own GEMM output is HWC before its transpose, src/conv2d.cc:134-136). */
#define I8IE_LAYOUT_NCHW 0
#define I8IE_LAYOUT_NHWC 1
/* NHWC with every byte stored as value ^ 0x80 (= value - 128 as s8; border bytes: zero point ^ 0x80).  The matrix
 * cores have no unsigned 8-bit operand: a conv kernel re-biases its activations on the way in (the exact term
 * 128 * sum_k q_w[j,k] joins oc[j]).  Between two conv layers of this library the producer can store that form
 * directly and the consumer skip its re-bias pass.  Accepted as in_layout / out_layout of the conv layer forwards;
 * i8ie_rebias_u8 converts a whole buffer either way. */
#define I8IE_LAYOUT_NHWC_S8 2

/* ---- per-kernel timing (measurement aid; nothing like it in the reference) ---
 * Between start and stop every kernel launch of this ctx is bracketed by HIP
 * events on the ctx's stream; stop waits for the stream and returns one entry
 * per kernel name: launches, summed device time, summed algorithmic integer
 * ops (2 x MACs, unpadded dimensions) and algorithmic bytes. */
typedef struct i8ie_profile_entry {
  char name[64];
  uint64_t launches;
  double total_ms;
  double total_ops;
  double total_bytes;
} i8ie_profile_entry;
/* mfma_kernels_only != 0: bracket only the contraction kernels (those with algorithmic
 * ops), which keeps the instrumentation overhead of a timed region small */
int i8ie_profile_start(i8ie_ctx* ctx, int mfma_kernels_only);
int i8ie_profile_stop(i8ie_ctx* ctx, i8ie_profile_entry* entries, int max_entries, int* n_entries);

/* ---- device memory (replaces `new T[]` + py::capsule, include/tensor.h:26-61) */
/* Blocks come from a per-ctx caching allocator: i8ie_free() never blocks, and a freed
 * block may be handed to the next i8ie_malloc() at once.  That is safe because every
 * consumer of ctx memory runs on the ctx's stream; do not read a block from another
 * stream after freeing it.  i8ie_trim() returns cached blocks to the driver. */
int i8ie_malloc(i8ie_ctx* ctx, size_t bytes, void** dev);
int i8ie_free(i8ie_ctx* ctx, void* dev);
int i8ie_trim(i8ie_ctx* ctx);
int i8ie_memory_stats(i8ie_ctx* ctx, size_t* bytes_live, size_t* bytes_cached, size_t* n_device_allocs);
int i8ie_memcpy_h2d(i8ie_ctx* ctx, void* dst_dev, const void* src_host, size_t bytes);
int i8ie_memcpy_d2h(i8ie_ctx* ctx, void* dst_host, const void* src_dev, size_t bytes); /* waits */
int i8ie_memcpy_d2d(i8ie_ctx* ctx, void* dst_dev, const void* src_dev, size_t bytes);
int i8ie_memset(i8ie_ctx* ctx, void* dst_dev, int byte, size_t bytes);

/* ---- whole-forward replay as one HIP graph (small per-GPU batches are launch-bound) ----------
 * Nothing like it in the reference (its forward is a chain of host calls, i8ie/module.py); on the GPU a 125-image
 * AlexNet shard spends more time between its ~14 dependent launches than inside them.  Between begin and end every
 * launch and asynchronous copy this ctx issues is recorded instead of run; device blocks freed meanwhile stay owned
 * by the graph (their addresses are baked into it) until i8ie_graph_destroy.  Run the same calls once eagerly
 * first: workspace growth, weight re-packing caches and offset vectors must exist before the capture (anything
 * that would synchronise inside it fails the capture with I8IE_ERR_HIP).  Replays are ordered on the ctx's
 * stream like any other launch; inputs and outputs are the device buffers the captured calls used. */
typedef struct i8ie_graph i8ie_graph;
/* While graphs captured on a ctx are alive its workspace is pinned: a later call that needs a larger one gets a new
 * allocation and the old one stays until the last graph is destroyed.  Weights re-packed by a kernel are kept per
 * packing key in the layer handle and never overwritten, so a replay after an eager forward at another batch size
 * still finds its own.  A layer handle must outlive the graphs that captured its forward.
 * i8ie_ctx_is_capturing: *yes = 1 between begin and end (owners of cached device blocks use it to route frees
 * to i8ie_free, which hands the block to the graph, instead of to a cache of their own). */
int i8ie_ctx_is_capturing(i8ie_ctx* ctx, int* yes);
int i8ie_graph_begin(i8ie_ctx* ctx);
int i8ie_graph_end(i8ie_ctx* ctx, i8ie_graph** out);
int i8ie_graph_launch(i8ie_graph* g);
int i8ie_graph_nodes(i8ie_graph* g, int* kernel_nodes, int* all_nodes);
int i8ie_graph_destroy(i8ie_graph* g);

/* ---- asynchronous host<->device transfers -------
 * Nothing like it in the reference (its tensors are host arrays, include/tensor.h:26-61); this is
 * what lets a caller overlap the logits read-back and the next batch's upload with the kernels.
 * Host buffers given to the *_async copies must come from i8ie_host_malloc (pinned); a pageable
 * pointer is refused with I8IE_ERR_ARG.  `on_copy_stream` != 0 issues the copy on the ctx's second
 * (transfer) stream so that it runs beside the kernels; order it against the compute stream with
 * events: record on one stream, i8ie_stream_wait_event on the other (device-side wait, the host
 * does not block), or i8ie_event_synchronize to block the host until the recorded point. */
typedef struct i8ie_event i8ie_event;
int i8ie_host_malloc(i8ie_ctx* ctx, size_t bytes, void** host);
int i8ie_host_free(i8ie_ctx* ctx, void* host);
/* *yes = 1 when [p, p+bytes) lies inside one live i8ie_host_malloc block of this ctx */
int i8ie_host_is_pinned(i8ie_ctx* ctx, const void* p, size_t bytes, int* yes);
int i8ie_memcpy_h2d_async(i8ie_ctx* ctx, void* dst_dev, const void* src_pinned, size_t bytes, int on_copy_stream);
int i8ie_memcpy_d2h_async(i8ie_ctx* ctx, void* dst_pinned, const void* src_dev, size_t bytes, int on_copy_stream);
int i8ie_event_create(i8ie_ctx* ctx, i8ie_event** out);
int i8ie_event_record(i8ie_ctx* ctx, i8ie_event* ev, int on_copy_stream);
int i8ie_stream_wait_event(i8ie_ctx* ctx, i8ie_event* ev, int copy_stream_waits);
int i8ie_event_synchronize(i8ie_event* ev);
int i8ie_event_query(i8ie_event* ev, int* done);
int i8ie_event_destroy(i8ie_event* ev);

/* ---- elementwise ops ---------------------------------------------------- */
/* quantize(Tensor<float>&, scale, zp)  src/quantize_utils.cc:44-52, src/pybind11.cc:41-45
 * q = (u8)(x / scale + zp): fp32 divide, add, truncate, low 8 bits (no clamp). */
int i8ie_quantize_f32_u8(i8ie_ctx* ctx, const float* in_dev, uint8_t* out_dev, int64_t n,
                         float scale, uint8_t zero_point);
/* dequantize(Tensor<u8>&)  src/quantize_utils.cc:38-42,54-58, src/pybind11.cc:46-48 */
int i8ie_dequantize_u8_f32(i8ie_ctx* ctx, const uint8_t* in_dev, float* out_dev, int64_t n,
                           float scale, uint8_t zero_point);
/* out = in ^ 0x80 over n bytes (in place allowed): plain u8 <-> the re-biased bytes of I8IE_LAYOUT_NHWC_S8 */
int i8ie_rebias_u8(i8ie_ctx* ctx, const uint8_t* in_dev, uint8_t* out_dev, int64_t n);
/* relu<u8_t>  src/functional.cc:15-26:  out = in > zp ? in : zp */
int i8ie_relu_u8(i8ie_ctx* ctx, const uint8_t* in_dev, uint8_t* out_dev, int64_t n,
                 uint8_t zero_point);
/* max_pool2d<u8_t>  src/functional.cc:36-64: NCHW, k x k window, stride s, floor, no padding */
int i8ie_maxpool2d_u8(i8ie_ctx* ctx, const uint8_t* in_dev, uint8_t* out_dev, int n, int c,
                      int h, int w, int kernel_size, int stride);
/* the s8 instantiations the reference registers as well (src/functional.cc:78-82): the generic templates,
 * relu = x > 0 ? x : 0 (src/functional.cc:5-13), max-pool with a running maximum starting at -127
 * (src/functional.cc:28-31, 36-64).  NCHW, no quantisation parameters involved. */
int i8ie_relu_s8(i8ie_ctx* ctx, const int8_t* in_dev, int8_t* out_dev, int64_t n);
int i8ie_maxpool2d_s8(i8ie_ctx* ctx, const int8_t* in_dev, int8_t* out_dev, int n, int c, int h, int w,
                      int kernel_size, int stride);
/* Calibrator::sample  src/calibrator.cc:6-23 on the device, without copying the layer output to the host.
 * The reference keeps 1000 samples: the first 1000 values seen fill the slots, every later value draws
 * idx uniform in [0, 2000] from an UNSEEDED mt19937 (std::random_device) and overwrites slot idx when idx < 1000, so
 * each slot ends up holding the last value that drew it.  Here the draw of global element number g is a
 * counter-based hash of (seed, g): same distribution of slot contents, a different (and, unlike the reference's,
 * reproducible) random stream.  `samples_dev` holds 1000 floats, `scratch_dev` 1000 ints; `seen_before` = values
 * already sampled by earlier calls on this layer.  The host-side replay of the reference's own mt19937 stream
 * (golden-pinned, tests/golden/ref_calibrator*.npz) stays in _CXX_i8ie for seeded runs. */
int i8ie_calib_sample_f32(i8ie_ctx* ctx, const float* data_dev, int64_t n, int64_t seen_before, uint64_t seed,
                          float* samples_dev, int* scratch_dev);
/* down_scale  src/quantize_utils.cc:27-36 (standalone requantiser; the layers fuse it) */
int i8ie_down_scale(i8ie_ctx* ctx, const int32_t* acc_dev, uint8_t* out_dev, int64_t n, float sa,
                    float sb, float sc, uint8_t zp_c);

/* ---- FP32 ops: the path taken before convert() and while calibrating -------
 * Conv2d/Linear::forward_prop(Tensor<float>&&)  src/conv2d.cc:63-98, src/fully_connected.cc:5-21
 * (cblas_sgemm + bias); relu<float>, max_pool2d<float>  src/functional.cc:5-13,36-64.
 * Off the timed INT8 path: plain correct kernels, pinned to the reference's own
 * tolerance (atol 0.1 vs torch, unittest/test_layers.py:10-11). */
int i8ie_linear_f32(i8ie_ctx* ctx, const float* in_dev, int m, int k, const float* w_dev,
                    const float* b_dev, int n, float* out_dev);
int i8ie_conv2d_f32(i8ie_ctx* ctx, const float* in_dev, int n, int c, int h, int w,
                    const float* w_dev, const float* b_dev, int kc, int kh, int kw, int stride,
                    int pad, float* out_dev);
int i8ie_relu_f32(i8ie_ctx* ctx, const float* in_dev, float* out_dev, int64_t n);
int i8ie_maxpool2d_f32(i8ie_ctx* ctx, const float* in_dev, float* out_dev, int n, int c, int h,
                       int w, int kernel_size, int stride);

/* ---- weight preparation (host side, one-shot) --------------------------- */
/* quantize_weight  src/layer.cc:6-26: joint min/max over weight and bias,
 * s_w = (max - min) / 127, q = (s8)(x / s_w) truncating, unclamped.  Host pointers. */
int i8ie_quantize_weight(const float* w_host, int64_t nw, const float* b_host, int64_t nb,
                         int8_t* qw_host, int8_t* qb_host, float* scale_out);

/* ---- zero-point offset vectors (device) --------------------------------- */
/* src/conv2d.cc:117-124:  t_j = sum_k zp_in*q_w[j,k] accumulated sequentially in fp32;
 * oc[j] = (int)((float)q_b[j] / s_in - t_j).  qw_dev: s8 [kc, K]; oc_dev: int32 [kc]. */
int i8ie_conv_offsets(i8ie_ctx* ctx, const int8_t* qw_dev, const int8_t* qb_dev, int kc, int K,
                      float s_in, uint8_t zp_in, int32_t* oc_dev);
/* src/fully_connected.cc:30-38:  oc[i] = (int)(-t_i) */
int i8ie_linear_offsets(i8ie_ctx* ctx, const int8_t* qw_dev, int n, int k, uint8_t zp_in,
                        int32_t* oc_dev);

/* ---- stateless layer entry points (raw device pointers) ------------------ */
/* Linear::forward_prop(Tensor<u8>&&)  src/fully_connected.cc:22-52
 *   C = X*W^T + oc (exact int32);  C = (int)((float)C + (float)q_b[j]/s_in);
 *   out = down_scale(C, s_in, s_w, s_out, zp_out).
 * in [m,k] u8, qw [n,k] s8, qb [n] s8, oc [n] int32 (from i8ie_linear_offsets), out [m,n] u8.
 * acc_dbg_dev: NULL, or int32 [m,n] receiving C BEFORE the float bias step (the
 * MKL cblas_gemm_s8u8s32 result, src/fully_connected.cc:39-41).               */
int i8ie_linear_u8s8(i8ie_ctx* ctx, const uint8_t* in_dev, int m, int k, const int8_t* qw_dev,
                     const int8_t* qb_dev, int n, const int32_t* oc_dev, float s_in, float s_w,
                     float s_out, uint8_t zp_out, uint8_t* out_dev, int32_t* acc_dbg_dev);
/* Conv2d::forward_prop(Tensor<u8>&&)  src/conv2d.cc:100-142
 *   per image: im2col (pad value zp_in) -> C = A*W^T + oc -> down_scale -> HWC->CHW.
 * in NCHW [n,c,h,w] u8, qw [kc, c*kh*kw] s8, oc [kc] int32 (from i8ie_conv_offsets),
 * out NCHW [n,kc,oh,ow] u8, oh = (h - kh + 2*pad)/stride + 1.
 * acc_dbg_dev: NULL, or int32 [n, oh*ow, kc]: the pre-requant accumulators
 * (the MKL result, src/conv2d.cc:131-133).                                   */
int i8ie_conv2d_u8s8(i8ie_ctx* ctx, const uint8_t* in_dev, int n, int c, int h, int w,
                     const int8_t* qw_dev, int kc, int kh, int kw, int stride, int pad,
                     uint8_t zp_in, const int32_t* oc_dev, float s_in, float s_w, float s_out,
                     uint8_t zp_out, uint8_t* out_dev, int32_t* acc_dbg_dev);

/* ---- layer handles: converted layers with device-resident packed weights -
 * What BaseLayer::convert() leaves behind (src/layer.cc:36-54: q_weight_,
 * q_bias_, scale_, zero_point_), kept on the device in MFMA operand order,
 * plus the offset vector cached per (s_in, zp_in) instead of being recomputed
 * on every call (src/conv2d.cc:117-124).                                     */
int i8ie_linear_create(i8ie_ctx* ctx, const int8_t* qw_host, const int8_t* qb_host, int n, int k,
                       float s_w, i8ie_layer** out);
int i8ie_conv2d_create(i8ie_ctx* ctx, const int8_t* qw_host, const int8_t* qb_host, int kc, int c,
                       int kh, int kw, int stride, int pad, float s_w, i8ie_layer** out);
/* the layer's output (scale_, zero_point_): src/layer.cc:44, include/layer.h:46-47 (default 1, 0) */
int i8ie_layer_set_output_qparams(i8ie_layer* layer, float s_out, uint8_t zp_out);
int i8ie_layer_get_output_qparams(const i8ie_layer* layer, float* s_out, uint8_t* zp_out);
/* Linear: in [m,k] -> out [m,n].  Conv2d: in NCHW [m,c,h,w] (m = batch) -> out NCHW.
 * h, w are ignored for Linear.  acc_dbg_dev as in the stateless calls.       */
int i8ie_layer_forward(i8ie_layer* layer, const uint8_t* in_dev, int m, int h, int w, float s_in,
                       uint8_t zp_in, uint8_t* out_dev, int32_t* acc_dbg_dev);
/* dequantize(linear(x)): src/quantize_utils.cc:54-58 applied to the result of Linear::forward_prop(u8)
 * (src/fully_connected.cc:22-52), in one call.  Layers with at most 16 output features (classifier heads)
 * run a fused kernel -- one wavefront per input row, v_dot4_i32_i8 over K, wavefront reduction, the
 * reference's bias/requant epilogue and the dequantize -- and may pass out_u8 = NULL; any other Linear
 * layer runs its ordinary forward into out_u8 (required) followed by the dequantize kernel.
 * in_layout / h / w as for i8ie_layer_forward_fused. */
int i8ie_layer_forward_dequant(i8ie_layer* layer, const uint8_t* in_dev, int in_layout, int m, int h, int w,
                               float s_in, uint8_t zp_in, int relu, uint8_t* out_u8_dev, float* out_f32_dev);

/* Same computation with the layout conversions and the following relu<u8>
 * (src/functional.cc:15-26: out = max(out, zp_out)) folded in.  in/out may each be NCHW or
 * NHWC; an NHWC tensor may carry a physical border of `border` pixels on each side of H and W
 * ([n][h+2b][w+2b][c]) whose bytes hold the tensor's zero point: a conv whose input border
 * covers its padding gathers with no bounds checks (the pad-with-zero-point rule of
 * src/conv2d.cc:24-28 is materialised by the producer).  When out_border > 0 the call writes
 * only the interior of `out`: the border bytes are the caller's (i8ie_fill_border_u8 with
 * zp_out, once per buffer: they stay valid while the buffer is reused for the same tensor).
 * i8ie_layer_preferred_layout tells which output layout avoids a conversion.
 * Linear layers: layouts do not apply to [m][k] rows, with one exception -- in_layout = NHWC together with
 * h, w > 1 declares the rows to be a flattened NHWC activation [m][h][w][c] (c = in_features / (h*w)) rather
 * than the reference's flattened NCHW (`x.reshape(n, -1)` of a Tensor<u8_t>, include/tensor.h:106-133): the
 * layer then walks K in (h, w, c) order with a weight panel permuted once, sparing the transpose. */
int i8ie_layer_forward_fused(i8ie_layer* layer, const uint8_t* in_dev, int in_layout, int in_border, int m,
                             int h, int w, float s_in, uint8_t zp_in, int relu, uint8_t* out_dev,
                             int out_layout, int out_border, int32_t* acc_dbg_dev);
int i8ie_layer_preferred_layout(const i8ie_layer* layer, int* layout);
/* relu(conv(x)) followed by max_pool2d<u8_t>(., kernel_size, stride) (src/functional.cc:36-64; the reference's models
 * call it right behind conv1 / conv2 / conv5, sample/notebooks/AlexNet_cifar10_resize224.ipynb:60-68) as ONE call:
 * out holds the POOLED tensor, [m, kc, (oh - k)/s + 1, (ow - k)/s + 1] in out_layout.  Where a kernel of this library
 * folds the pool into the convolution's epilogue (i8ie_layer_fuses_pool says so: it then pools the INT32
 * accumulators before requantising them -- down_scale and relu are monotone in C, so the bytes are those of
 * pooling afterwards), no unpooled tensor is ever written; otherwise the call runs the convolution into a
 * temporary and the max-pool kernel behind it.  Identical bytes to i8ie_layer_forward_fused followed by
 * i8ie_maxpool2d_u8(_nhwc) either way.  acc_dbg_dev: the convolution's (unpooled) accumulators [m, oh*ow, kc]. */
int i8ie_layer_fuses_pool(const i8ie_layer* layer, int m, int h, int w, int kernel_size, int stride, int* yes);
int i8ie_layer_forward_pool(i8ie_layer* layer, const uint8_t* in_dev, int in_layout, int in_border, int m, int h, int w,
                            float s_in, uint8_t zp_in, int relu, int kernel_size, int stride, uint8_t* out_dev,
                            int out_layout, int out_border, int32_t* acc_dbg_dev);
/* I8IE_LAYOUT_NHWC_S8 between two conv layers: *reads = 1 when this layer's kernel (at batch m, input h x w, with this
 * pool folded in; kernel_size <= 1: none) consumes re-biased input as it is, *stores = 1 when it can store its result
 * re-biased.  The forward calls accept the layout either way (they convert around a kernel that does not); a caller
 * that owns both ends asks here first, so that no conversion launch is ever needed. */
int i8ie_layer_rebiased_io(const i8ie_layer* layer, int m, int h, int w, int kernel_size, int stride, int* reads,
                           int* stores);
/* First layer fused with the input quantisation (Module.__call__ quantises the FP32 input with
 * 0.025 / 127, i8ie/module.py:20, and hands it straight to the first Conv2d): reads FP32 NCHW,
 * computes q = (u8)(x / q_scale + q_zp) exactly as src/quantize_utils.cc:44-52 and the conv of
 * src/conv2d.cc:100-142 on it, writes the interior of an NHWC u8 tensor (+ optional relu).  Only for layers and
 * geometries where i8ie_layer_accepts_f32_input says yes (<= 3 channels, stride % 4 == 0,
 * out features % 32 == 0): AlexNet's 11x11 stride-4 conv1.  Identical bytes to
 * i8ie_quantize_f32_u8 followed by i8ie_layer_forward_fused.  acc_dbg_dev: NULL, or int32 [m, oh*ow, kc] receiving
 * the pre-requant accumulators (the MKL result, src/conv2d.cc:131-133), as in the other forward calls. */
int i8ie_layer_accepts_f32_input(const i8ie_layer* layer, int h, int w, int* yes);
int i8ie_layer_forward_f32_input(i8ie_layer* layer, const float* in_nchw_dev, int m, int h, int w, float q_scale,
                                 uint8_t q_zp, int relu, uint8_t* out_nhwc_dev, int out_border, int32_t* acc_dbg_dev);
/* the same with max_pool2d<u8_t>(kernel_size, stride) behind the (relu'd) convolution, as in i8ie_layer_forward_pool:
 * quantize -> conv1 -> relu -> max-pool of AlexNet in one contraction launch (csrc/i8ie_stem.hip) */
int i8ie_layer_forward_f32_input_pool(i8ie_layer* layer, const float* in_nchw_dev, int m, int h, int w, float q_scale,
                                      uint8_t q_zp, int relu, int kernel_size, int stride, uint8_t* out_nhwc_dev,
                                      int out_layout /* NHWC or NHWC_S8 */, int out_border, int32_t* acc_dbg_dev);
/* padding of a conv layer (0 for Linear): the input border that makes its gather predicate-free */
int i8ie_layer_padding(const i8ie_layer* layer, int* pad);
int i8ie_layer_destroy(i8ie_layer* layer);

/* ---- NHWC companions of the elementwise ops (internal layout between layers) ------------ */
/* NCHW [n,c,h,w] <-> NHWC [n,h+2b,w+2b,c] of a u8 tensor (to_nhwc != 0: NCHW -> NHWC, the border
 * of the destination is filled with border_value) */
int i8ie_layout_convert_u8(i8ie_ctx* ctx, const uint8_t* in_dev, uint8_t* out_dev, int n, int c, int h,
                           int w, int to_nhwc, int border, uint8_t border_value);
/* border bytes of a bordered NHWC u8 tensor [n,h+2b,w+2b,c] := value (interior untouched) */
int i8ie_fill_border_u8(i8ie_ctx* ctx, uint8_t* buf_dev, int n, int c, int h, int w, int border,
                        uint8_t value);
/* max_pool2d<u8_t> (src/functional.cc:36-64) on NHWC data, channels % 16 == 0; h, w are the
 * logical input dims; only the interior of a bordered output is written */
int i8ie_maxpool2d_u8_nhwc(i8ie_ctx* ctx, const uint8_t* in_dev, int in_border, uint8_t* out_dev,
                           int out_border, int n, int c, int h, int w, int kernel_size, int stride);

#ifdef __cplusplus
}
#endif
#endif /* I8IE_HIP_H */
