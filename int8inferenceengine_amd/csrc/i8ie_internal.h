// i8ie_internal.h -- shared by the translation units of libi8ie_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstddef>
#include <cstdint>

#include "i8ie_hip.h"

struct i8ie_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  void* ws = nullptr;  // scratch for one op at a time (stream-ordered reuse)
  size_t ws_bytes = 0;
  void* pool = nullptr;  // I8iePool* (i8ie_ctx.hip): stream-ordered caching allocator
  void* prof = nullptr;  // I8ieProf* (i8ie_ctx.hip): HIP-event timing of every launch, when enabled
  unsigned options = 0;  // bit 0: I8IE_OPT_FORCE_FALLBACK
  int cu_limit = 0;  // I8IE_OPT_CU_LIMIT: compute units of the stream's CU mask (0 = the whole device)
  int variant = 0;  // I8IE_OPT_KERNEL_VARIANT: selects among compiled kernel variants (A/B timing aid)
  int prof_mfma_only = 0;  // time only the contraction kernels (fewer event packets in a timed region)
  int prof_stride = 1;     // I8IE_OPT_PROFILE_STRIDE: bracket every prof_stride-th eligible launch
  unsigned prof_seen = 0;  // eligible launches since i8ie_profile_start
  hipStream_t copy_stream = nullptr;  // transfer stream for the *_async copies (created on first use)
  void* pinned = nullptr;  // std::unordered_map<void*, size_t>* of i8ie_host_malloc blocks
  void* capture = nullptr;  // std::vector<void*>* while a graph is being captured: blocks freed meanwhile (i8ie_graph_*)
  int live_graphs = 0;      // graphs captured on this ctx and not yet destroyed: their kernels have ctx->ws baked in
  void* retired_ws = nullptr;  // std::vector<void*>*: workspaces outgrown while graphs were alive (freed with the last graph)
};

// Per-launch HIP-event timing on the ctx's stream (off unless i8ie_profile_start was
// called).  `ops` = algorithmic integer ops (2 * MACs), `bytes` = algorithmic bytes.
void i8ie_prof_begin(i8ie_ctx* ctx, const char* name, double ops, double bytes);
void i8ie_prof_end(i8ie_ctx* ctx);
struct I8ieProfScope {
  i8ie_ctx* c;
  I8ieProfScope(i8ie_ctx* ctx, const char* name, double ops, double bytes) : c(ctx) {
    on = c->prof != nullptr && (!c->prof_mfma_only || ops > 0.0);
    if (on && c->prof_stride > 1) on = (c->prof_seen++ % (unsigned)c->prof_stride) == 0;
    if (on) i8ie_prof_begin(c, name, ops, bytes);
  }
  ~I8ieProfScope() {
    if (on) i8ie_prof_end(c);
  }
  bool on;
};

void i8ie_set_error(const char* fmt, ...);

// I8IE_OPT_KERNEL_VARIANT values that name a Linear kernel (80, 81, 83) or a first-stage form (12, 13) leave the dispatch of the
// OTHER convolutions automatic (round 3's advice: a variant changes the one thing it names; with 83 set process-wide the conv
// layers used to lose the patch-stationary kernel their pools were planned for)
// compute units a one-block-per-CU kernel may count on: the device's, or the ctx's CU-mask share
inline int i8ie_cus(const i8ie_ctx* ctx, int device_cus) { return ctx->cu_limit > 0 && ctx->cu_limit < device_cus ? ctx->cu_limit : device_cus; }
inline bool i8ie_conv_variant_auto(int v) { return v == 0 || v == 12 || v == 13 || v == 16 || v == 80 || v == 81 || v == 83 || v == 84 || v == 85; }

#define I8IE_HIP_TRY(expr)                                                                  \
  do {                                                                                      \
    hipError_t e__ = (expr);                                                                \
    if (e__ != hipSuccess) {                                                                \
      i8ie_set_error("%s: %s (%s:%d)", #expr, hipGetErrorString(e__), __FILE__, __LINE__); \
      return I8IE_ERR_HIP;                                                                  \
    }                                                                                       \
  } while (0)

#define I8IE_REQUIRE(cond, msg)                                   \
  do {                                                            \
    if (!(cond)) {                                                \
      i8ie_set_error("%s: %s (%s)", __func__, msg, #cond);        \
      return I8IE_ERR_ARG;                                        \
    }                                                             \
  } while (0)

#define I8IE_TRY(expr)            \
  do {                            \
    int rc__ = (expr);            \
    if (rc__ != I8IE_OK) return rc__; \
  } while (0)

#define I8IE_LAUNCH_CHECK()  I8IE_HIP_TRY(hipGetLastError())

// Make sure ctx->ws holds at least `bytes`; grows by sync + realloc (rare).
int i8ie_ws_reserve(i8ie_ctx* ctx, size_t bytes);

static inline size_t i8ie_align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// ---- GEMM core (i8ie_gemm.hip) ---------------------------------------------
enum { I8IE_OUT_ROWMAJOR = 0, I8IE_OUT_NCHW = 1 };

struct I8ieGemmArgs {
  const uint8_t* A;  // u8 [M][lda], rows 16-B aligned, K-contiguous
  long lda;
  int M;
  int Ka;            // readable bytes per A row (multiple of 16)
  const int8_t* B;   // s8 packed [Npad][Kpad], zero padded, Npad % 128 == 0, Kpad % 64 == 0
  int Kpad;
  int N;
  const int32_t* oc;    // [N] reference offset vector
  const int32_t* wsum;  // [N] sum_k q_w[j,k] (exact): the u8 -> s8 re-biasing term
  const int8_t* qb;     // [N] or nullptr: Linear's float bias step
  float s_in, s_w, s_out;
  int zp_out;
  uint8_t* out;
  int out_mode;  // I8IE_OUT_ROWMAJOR: out[M][N]; I8IE_OUT_NCHW: out[(img*N + j)*P + p], row = img*P + p
  int P;
  int32_t* acc;  // nullptr or [M][N]
  int Ktrue;     // unpadded reduction length (profiling: algorithmic ops = 2*M*N*Ktrue)
};
int i8ie_gemm_launch(i8ie_ctx* ctx, const I8ieGemmArgs& a);
