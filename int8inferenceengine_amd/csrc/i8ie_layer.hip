// i8ie_layer.hip -- Linear / Conv2d INT8 forward entry points of the C-ABI:
// the stateless calls on raw device pointers and the layer handles that keep
// the converted weights resident (packed for the MFMA kernels) and cache the
// zero-point offset vector per (s_in, zp_in).
//   reference: src/fully_connected.cc:22-52, src/conv2d.cc:100-142,
//              src/layer.cc:6-26,36-54
//
// Conv2d has three execution paths, all producing the reference's bytes:
//   A  channels % 16 == 0: implicit GEMM over NHWC activations (i8ie_igemm.hip)
//   B  channels <= 4 and stride % 4 == 0 (AlexNet conv1): the NCHW input is repacked
//      once into a physically padded, 4-pixel-grouped NHWC image; then path A's kernel
//   F  anything else: materialised im2col + v1 GEMM (i8ie_gemm.hip)
// Activations cross the ABI as NCHW (the reference's layout) or, on request, as
// NHWC so that consecutive layers skip the layout conversion.
#include <cstring>
#include <new>
#include <vector>

#include "i8ie_internal.h"
#include "i8ie_calls.h"
#include "i8ie_stem.h"

int i8ie_launch_pad_rows(i8ie_ctx* ctx, const void* src, int rows, int k, void* dst, int rows_pad, int k_pad,
                         int fill);
int i8ie_launch_offsets(i8ie_ctx* ctx, bool conv, const int8_t* qw, const int8_t* qb, int n, int K, float s_in,
                        int zp_in, int32_t* oc, int32_t* wsum);
int i8ie_launch_im2col(i8ie_ctx* ctx, const uint8_t* in, uint8_t* col, int n, int c, int h, int w, int kh, int kw,
                       int oh, int ow, int stride, int pad, int K, int Kpad, int zp);
int i8ie_launch_finish_offsets(i8ie_ctx* ctx, const int32_t* oc, const int32_t* wsum, const int8_t* qb, float s_in,
                               int n, int32_t* ocp, float* biasf);
int i8ie_launch_nchw_to_nhwc(i8ie_ctx* ctx, const uint8_t* in, uint8_t* out, int n, int c, int h, int w, int b);
int i8ie_launch_nhwc_to_nchw(i8ie_ctx* ctx, const uint8_t* in, uint8_t* out, int n, int c, int h, int w, int b);
int i8ie_launch_reborder(i8ie_ctx* ctx, const uint8_t* in, uint8_t* out, int n, int c, int h, int w, int ib, int ob,
                         int zp);
int i8ie_launch_permute_k(i8ie_ctx* ctx, const int8_t* B, int8_t* Bp, int rows, int Kpad, int K, int c, int hw);
struct I8ieSmallNCall {
  const uint8_t* A;
  size_t lda;
  int M, K;
  const int8_t* B;
  int Kpad, N;
  const int32_t* ocp;
  const float* biasf;
  float s_in, s_w, s_out;
  int zp_out, relu;
  uint8_t* out;
  int32_t* acc;
  float* out_f32;
};
int i8ie_smalln_max_features();
#if defined(I8IE_DIAG)  // tools/diag/csrc/i8ie_skinny.hip
int i8ie_launch_frag_pack(i8ie_ctx* ctx, const int8_t* B, void* Bf, int Npad, int Kpad);
int i8ie_skinny_plan(int m, int n, int Kpad, int* nstep, int* slices);
int i8ie_launch_linear_skinny(i8ie_ctx* ctx, const uint8_t* A, size_t lda, int m, const void* Bf, int Kpad, int Npad,
                              int n, const int32_t* ocp, int32_t* partial, int nstep, int slices);
#endif
int i8ie_launch_splitk_reduce(i8ie_ctx* ctx, const int32_t* partial, int slices, int M, int N, const float* biasf,
                              float s_in, float s_w, float s_out, int zp_out, int relu, uint8_t* out, int32_t* acc);
int i8ie_launch_linear_smalln(i8ie_ctx* ctx, const I8ieSmallNCall& c);
int i8ie_launch_repack_smallc(i8ie_ctx* ctx, const uint8_t* in, uint8_t* out, int n, int c, int h, int w, int Hp,
                              int Wg, int ph, int pw, int zp, int rebias);
int i8ie_launch_fill_border(i8ie_ctx* ctx, uint8_t* out, int n, int c, int h, int w, int b, int zp);
int i8ie_launch_maxpool_nhwc(i8ie_ctx* ctx, const uint8_t* in, int ib, uint8_t* out, int ob, int n, int c, int h,
                             int w, int k, int s, int relu_zp);



struct I8ieFirstCall {
  const float* x;
  const uint8_t* grouped;
  uint8_t* scratch;
  int n, c, h, w;
  float q_scale;
  int q_zp;
  int KH, KW, KWG, stride, pad, OH, OW;
  const int8_t* B;
  int Kpad, K2, N;
  const int32_t* ocp;
  float s_in, s_w, s_out;
  int zp_out, relu;
  uint8_t* out;
  int ob;
  int32_t* acc;
};
int i8ie_first_supported(int c, int stride, int n_out, int K2, int KH, int KWG, int OW);
int i8ie_first_launch(i8ie_ctx* ctx, const I8ieFirstCall& c);
size_t i8ie_first_scratch_bytes(int n, int KH, int KWG, int stride, int OH, int OW);

namespace {

constexpr size_t kColBudget = (size_t)192 << 20;  // im2col scratch per chunk (fallback path F)

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }
inline int round_up(int x, int a) { return (x + a - 1) / a * a; }

enum { PATH_F = 0, PATH_A = 1, PATH_B = 2 };

struct ConvGeom {
  int c, h, w, kc, kh, kw, stride, pad, oh, ow, K, Kpad;
};

int conv_geom(int c, int h, int w, int kc, int kh, int kw, int stride, int pad, ConvGeom* g) {
  I8IE_REQUIRE(c > 0 && h > 0 && w > 0 && kc > 0 && kh > 0 && kw > 0, "non-positive dimension");
  I8IE_REQUIRE(stride > 0, "stride must be positive");  // include/conv2d.h:12-14
  I8IE_REQUIRE(pad >= 0, "negative padding");
  I8IE_REQUIRE(h - kh + 2 * pad >= 0 && w - kw + 2 * pad >= 0, "kernel larger than padded input");
  g->c = c; g->h = h; g->w = w; g->kc = kc; g->kh = kh; g->kw = kw; g->stride = stride; g->pad = pad;
  g->oh = (h - kh + 2 * pad) / stride + 1;  // src/conv2d.cc:108-109
  g->ow = (w - kw + 2 * pad) / stride + 1;
  g->K = c * kh * kw;
  g->Kpad = round_up(g->K, 128);
  return I8IE_OK;
}

inline int chunk_images(const ConvGeom& g, int n) {
  const size_t per_img = (size_t)g.oh * g.ow * g.Kpad;
  size_t imgs = kColBudget / per_img;
  if (imgs < 1) imgs = 1;
  return imgs > (size_t)n ? n : (int)imgs;
}

// ---- v1 (fallback) runners ------------------------------------------------------------------
int conv_run_v1(i8ie_ctx* ctx, const uint8_t* in, int n, const ConvGeom& cg, const int8_t* Bpack, const int32_t* oc,
                const int32_t* wsum, uint8_t zp_in, float s_in, float s_w, float s_out, uint8_t zp_out, uint8_t* out,
                int32_t* acc, uint8_t* col, int imgs_per_chunk) {
  const int P = cg.oh * cg.ow;
  for (int i0 = 0; i0 < n; i0 += imgs_per_chunk) {
    const int nb = (n - i0) < imgs_per_chunk ? (n - i0) : imgs_per_chunk;
    I8IE_TRY(i8ie_launch_im2col(ctx, in + (size_t)i0 * cg.c * cg.h * cg.w, col, nb, cg.c, cg.h, cg.w, cg.kh, cg.kw,
                                cg.oh, cg.ow, cg.stride, cg.pad, cg.K, cg.Kpad, zp_in));
    I8ieGemmArgs g{};
    g.A = col;
    g.lda = cg.Kpad;
    g.Ka = cg.Kpad;
    g.M = nb * P;
    g.B = Bpack;
    g.Kpad = cg.Kpad;
    g.N = cg.kc;
    g.oc = oc;
    g.wsum = wsum;
    g.qb = nullptr;  // conv folds the bias into oc (src/conv2d.cc:123)
    g.s_in = s_in;
    g.s_w = s_w;
    g.s_out = s_out;
    g.zp_out = zp_out;
    g.out = out + (size_t)i0 * cg.kc * P;
    g.out_mode = I8IE_OUT_NCHW;
    g.P = P;
    g.acc = acc ? acc + (size_t)i0 * P * cg.kc : nullptr;
    g.Ktrue = cg.K;
    I8IE_TRY(i8ie_gemm_launch(ctx, g));
  }
  return I8IE_OK;
}

int linear_run_v1(i8ie_ctx* ctx, const uint8_t* in, int m, int k, const int8_t* Bpack, int Kpad, const int8_t* qb,
                  int n, const int32_t* oc, const int32_t* wsum, float s_in, float s_w, float s_out, uint8_t zp_out,
                  uint8_t* out, int32_t* acc, uint8_t* scratch) {
  I8ieGemmArgs g{};
  if (k % 16 != 0 || !aligned16(in)) {
    I8IE_TRY(i8ie_launch_pad_rows(ctx, in, m, k, scratch, m, Kpad, 0));
    g.A = scratch;
    g.lda = Kpad;
    g.Ka = Kpad;
  } else {
    g.A = in;
    g.lda = k;
    g.Ka = k;
  }
  g.M = m; g.B = Bpack; g.Kpad = Kpad; g.N = n; g.oc = oc; g.wsum = wsum; g.qb = qb;
  g.s_in = s_in; g.s_w = s_w; g.s_out = s_out; g.zp_out = zp_out;
  g.out = out; g.out_mode = I8IE_OUT_ROWMAJOR; g.P = 1; g.acc = acc; g.Ktrue = k;
  return i8ie_gemm_launch(ctx, g);
}

}  // namespace

struct i8ie_layer {
  i8ie_ctx* ctx = nullptr;
  bool conv = false;
  int n = 0, K = 0;  // out features, reduction length (reference K order)
  int c = 0, kh = 0, kw = 0, stride = 1, pad = 0;
  int Kpad = 0, Npad = 0;
  float s_w = 1.0f;
  float s_out = 1.0f;   // include/layer.h:46
  uint8_t zp_out = 0;   // include/layer.h:47
  int8_t* qw = nullptr;     // [n][K] as converted, K ordered (c, kh, kw)
  int8_t* qb = nullptr;     // [n]
  int8_t* Bpack = nullptr;  // [Npad][Kpad] zero padded, reference K order (Linear; conv path F)
  int8_t* Bperm = nullptr;  // Linear fed by an NHWC-flattened activation: Bpack with K reordered (h, w, c)
  int perm_c = 0, perm_hw = 0;
  void* Bfrag = nullptr;       // Bpack in MFMA fragment order (few-row kernel, i8ie_skinny.hip), built on first use
  void* Bfrag_perm = nullptr;  // the same of Bperm
  bool bfrag_perm_valid = false, bfrag_valid = false;
  int path = PATH_F;        // conv: PATH_A / PATH_B / PATH_F
  int8_t* Bpack2 = nullptr; // conv paths A/B: [Npad][Kpad2], K ordered (kh, kw, c) / grouped
  int K2 = 0, Kpad2 = 0;    // valid / padded K of Bpack2 (bytes)
  I8ieWCache wc;            // Bpack2 in the fragment orders of i8ie_pconv.hip / i8ie_tconv.hip: one buffer per packing
                            // key, made on first use, never overwritten (captured graphs replay their addresses)
  int kwg = 0;              // path B: taps per row in 4-pixel groups
  int8_t* Bstem = nullptr;  // path B, when the first-stage kernel (i8ie_stem.hip) takes the layer: [n][KpadStem], K in its
  int KpadStem = 0;         // space-to-depth order (i8ie_stem_kindex), zero padded
  int32_t* wsum = nullptr;  // [n]
  int32_t* oc = nullptr;    // [n], valid for (oc_s_in, oc_zp_in)
  int32_t* ocp = nullptr;   // [n] oc + 128 * wsum
  float* biasf = nullptr;   // [n] (float)qb / s_in (Linear)
  bool oc_valid = false;
  float oc_s_in = 0.0f;
  int oc_zp_in = -1;
};

namespace {

int ensure_offsets(i8ie_layer* L, float s_in, uint8_t zp_in) {
  uint32_t a, b;
  memcpy(&a, &s_in, 4);
  memcpy(&b, &L->oc_s_in, 4);
  if (L->oc_valid && L->oc_zp_in == (int)zp_in && a == b) return I8IE_OK;
  // the reference recomputes this on every call (src/conv2d.cc:117-124); it only depends on
  // (s_in, zp_in), which are fixed once the network is converted
  i8ie_ctx* ctx = L->ctx;
  I8IE_TRY(i8ie_launch_offsets(ctx, L->conv, L->qw, L->qb, L->n, L->K, s_in, zp_in, L->oc, nullptr));
  I8IE_TRY(i8ie_launch_finish_offsets(ctx, L->oc, L->wsum, L->qb, s_in, L->n, L->ocp, L->conv ? nullptr : L->biasf));
  L->oc_valid = true;
  L->oc_s_in = s_in;
  L->oc_zp_in = zp_in;
  return I8IE_OK;
}

bool force_fallback(const i8ie_ctx* ctx) { return (ctx->options & 1) != 0; }

}  // namespace

extern "C" {

int i8ie_ctx_set_option(i8ie_ctx* ctx, int option, int value) {
  I8IE_REQUIRE(ctx != nullptr, "null ctx");
  I8IE_REQUIRE(option == I8IE_OPT_FORCE_FALLBACK || option == I8IE_OPT_KERNEL_VARIANT ||
                   option == I8IE_OPT_PROFILE_STRIDE || option == I8IE_OPT_CU_LIMIT,
               "unknown option");
  if (option == I8IE_OPT_CU_LIMIT) {
    I8IE_REQUIRE(value >= 0, "CU limit must be >= 0");
    ctx->cu_limit = value;
    return I8IE_OK;
  }
  if (option == I8IE_OPT_PROFILE_STRIDE) {
    I8IE_REQUIRE(value >= 1, "profile stride must be >= 1");
    ctx->prof_stride = value;
    return I8IE_OK;
  }
  if (option == I8IE_OPT_KERNEL_VARIANT) {
    ctx->variant = value;
    return I8IE_OK;
  }
  if (value)
    ctx->options |= 1;
  else
    ctx->options &= ~1;
  return I8IE_OK;
}

int i8ie_conv_offsets(i8ie_ctx* ctx, const int8_t* qw, const int8_t* qb, int kc, int K, float s_in, uint8_t zp_in,
                      int32_t* oc) {
  I8IE_REQUIRE(ctx && qw && qb && oc, "null argument");
  I8IE_REQUIRE(kc > 0 && K > 0, "non-positive dimension");
  I8IE_HIP_TRY(hipSetDevice(ctx->device));
  return i8ie_launch_offsets(ctx, true, qw, qb, kc, K, s_in, zp_in, oc, nullptr);
}

int i8ie_linear_offsets(i8ie_ctx* ctx, const int8_t* qw, int n, int k, uint8_t zp_in, int32_t* oc) {
  I8IE_REQUIRE(ctx && qw && oc, "null argument");
  I8IE_REQUIRE(n > 0 && k > 0, "non-positive dimension");
  I8IE_HIP_TRY(hipSetDevice(ctx->device));
  return i8ie_launch_offsets(ctx, false, qw, nullptr, n, k, 0.0f, zp_in, oc, nullptr);
}

// ---- stateless entry points: any geometry, v1 kernels, caller supplies oc --------------------
int i8ie_linear_u8s8(i8ie_ctx* ctx, const uint8_t* in, int m, int k, const int8_t* qw, const int8_t* qb, int n,
                     const int32_t* oc, float s_in, float s_w, float s_out, uint8_t zp_out, uint8_t* out,
                     int32_t* acc) {
  I8IE_REQUIRE(ctx && in && qw && qb && oc && out, "null argument");
  I8IE_REQUIRE(m > 0 && k > 0 && n > 0, "non-positive dimension");
  I8IE_HIP_TRY(hipSetDevice(ctx->device));
  const int Kpad = round_up(k, 128), Npad = round_up(n, 128);
  const size_t b_bytes = i8ie_align_up((size_t)Npad * Kpad, 256);
  const size_t s_bytes = i8ie_align_up((size_t)n * 4, 256);
  const size_t a_bytes = (size_t)m * Kpad;
  I8IE_TRY(i8ie_ws_reserve(ctx, b_bytes + s_bytes + a_bytes));
  uint8_t* ws = (uint8_t*)ctx->ws;
  int8_t* Bpack = (int8_t*)ws;
  int32_t* wsum = (int32_t*)(ws + b_bytes);
  uint8_t* scratch = ws + b_bytes + s_bytes;
  I8IE_TRY(i8ie_launch_pad_rows(ctx, qw, n, k, Bpack, Npad, Kpad, 0));
  I8IE_TRY(i8ie_launch_offsets(ctx, false, qw, nullptr, n, k, 0.0f, 0, nullptr, wsum));
  return linear_run_v1(ctx, in, m, k, Bpack, Kpad, qb, n, oc, wsum, s_in, s_w, s_out, zp_out, out, acc, scratch);
}

int i8ie_conv2d_u8s8(i8ie_ctx* ctx, const uint8_t* in, int n, int c, int h, int w, const int8_t* qw, int kc, int kh,
                     int kw, int stride, int pad, uint8_t zp_in, const int32_t* oc, float s_in, float s_w,
                     float s_out, uint8_t zp_out, uint8_t* out, int32_t* acc) {
  I8IE_REQUIRE(ctx && in && qw && oc && out, "null argument");
  I8IE_REQUIRE(n > 0, "non-positive batch");
  ConvGeom cg;
  I8IE_TRY(conv_geom(c, h, w, kc, kh, kw, stride, pad, &cg));
  I8IE_HIP_TRY(hipSetDevice(ctx->device));
  const int Npad = round_up(kc, 128);
  const int ipc = chunk_images(cg, n);
  const size_t b_bytes = i8ie_align_up((size_t)Npad * cg.Kpad, 256);
  const size_t s_bytes = i8ie_align_up((size_t)kc * 4, 256);
  const size_t col_bytes = (size_t)ipc * cg.oh * cg.ow * cg.Kpad;
  I8IE_TRY(i8ie_ws_reserve(ctx, b_bytes + s_bytes + col_bytes));
  uint8_t* ws = (uint8_t*)ctx->ws;
  int8_t* Bpack = (int8_t*)ws;
  int32_t* wsum = (int32_t*)(ws + b_bytes);
  uint8_t* col = ws + b_bytes + s_bytes;
  I8IE_TRY(i8ie_launch_pad_rows(ctx, qw, kc, cg.K, Bpack, Npad, cg.Kpad, 0));
  I8IE_TRY(i8ie_launch_offsets(ctx, true, qw, nullptr, kc, cg.K, 1.0f, 0, nullptr, wsum));
  return conv_run_v1(ctx, in, n, cg, Bpack, oc, wsum, zp_in, s_in, s_w, s_out, zp_out, out, acc, col, ipc);
}

// ---- NHWC helpers exposed on the ABI -----------------------------------------------------------
int i8ie_layout_convert_u8(i8ie_ctx* ctx, const uint8_t* in, uint8_t* out, int n, int c, int h, int w, int to_nhwc,
                           int border, uint8_t border_value) {
  I8IE_REQUIRE(ctx && in && out, "null argument");
  I8IE_REQUIRE(n > 0 && c > 0 && h > 0 && w > 0 && border >= 0, "bad dimension");
  I8IE_HIP_TRY(hipSetDevice(ctx->device));
  if (to_nhwc) {
    if (border > 0)
      I8IE_HIP_TRY(hipMemsetAsync(out, border_value, (size_t)n * (h + 2 * border) * (w + 2 * border) * c, ctx->stream));
    return i8ie_launch_nchw_to_nhwc(ctx, in, out, n, c, h, w, border);
  }
  return i8ie_launch_nhwc_to_nchw(ctx, in, out, n, c, h, w, border);
}

int i8ie_fill_border_u8(i8ie_ctx* ctx, uint8_t* buf, int n, int c, int h, int w, int border, uint8_t value) {
  I8IE_REQUIRE(ctx && buf, "null argument");
  I8IE_REQUIRE(n > 0 && c > 0 && h > 0 && w > 0 && border >= 0, "bad dimension");
  I8IE_HIP_TRY(hipSetDevice(ctx->device));
  return i8ie_launch_fill_border(ctx, buf, n, c, h, w, border, value);
}

int i8ie_maxpool2d_u8_nhwc(i8ie_ctx* ctx, const uint8_t* in, int in_border, uint8_t* out, int out_border, int n,
                           int c, int h, int w, int k, int s) {
  I8IE_REQUIRE(ctx && in && out, "null argument");
  I8IE_REQUIRE(n > 0 && c > 0 && h > 0 && w > 0 && in_border >= 0 && out_border >= 0, "bad dimension");
  I8IE_REQUIRE(c % 16 == 0, "NHWC max-pool needs channels % 16 == 0");
  I8IE_REQUIRE(k > 0 && s > 0, "kernel_size and stride must be positive");
  I8IE_REQUIRE(k <= h && k <= w, "window larger than the input");
  I8IE_REQUIRE(aligned16(in) && aligned16(out), "buffers must be 16-byte aligned");
  I8IE_HIP_TRY(hipSetDevice(ctx->device));
  return i8ie_launch_maxpool_nhwc(ctx, in, in_border, out, out_border, n, c, h, w, k, s, 0);
}

// ---- layer handles ----------------------------------------------------------------------------
static int layer_create(i8ie_ctx* ctx, bool conv, const int8_t* qw_host, const int8_t* qb_host, int n, int K, int c,
                        int kh, int kw, int stride, int pad, float s_w, i8ie_layer** out) {
  I8IE_REQUIRE(ctx && qw_host && qb_host && out, "null argument");
  I8IE_REQUIRE(n > 0 && K > 0, "non-positive dimension");
  I8IE_HIP_TRY(hipSetDevice(ctx->device));
  i8ie_layer* L = new (std::nothrow) i8ie_layer();
  if (!L) return I8IE_ERR_OOM;
  L->ctx = ctx; L->conv = conv; L->n = n; L->K = K; L->c = c; L->kh = kh; L->kw = kw;
  L->stride = stride; L->pad = pad; L->s_w = s_w;
  L->Kpad = round_up(K, 128);
  L->Npad = round_up(n, 128);
  // MFMA-order weight panel for the implicit-GEMM conv paths, built on the host once
  std::vector<int8_t> pack2;
  if (conv && c % 16 == 0) {
    L->path = PATH_A;
    L->K2 = kh * kw * c;
    L->Kpad2 = round_up(L->K2, 128);
    pack2.assign((size_t)L->Npad * L->Kpad2, 0);
    for (int j = 0; j < n; ++j)
      for (int ch = 0; ch < c; ++ch)
        for (int y = 0; y < kh; ++y)
          for (int x = 0; x < kw; ++x)
            pack2[(size_t)j * L->Kpad2 + ((size_t)y * kw + x) * c + ch] = qw_host[(((size_t)j * c + ch) * kh + y) * kw + x];
  } else if (conv && c <= 4 && stride % 4 == 0) {
    L->path = PATH_B;
    L->kwg = (kw + 3) / 4;
    L->K2 = kh * L->kwg * 16;
    L->Kpad2 = round_up(L->K2, 128);
    pack2.assign((size_t)L->Npad * L->Kpad2, 0);
    for (int j = 0; j < n; ++j)
      for (int ch = 0; ch < c; ++ch)
        for (int y = 0; y < kh; ++y)
          for (int x = 0; x < kw; ++x)
            pack2[(size_t)j * L->Kpad2 + ((size_t)y * L->kwg + x / 4) * 16 + (x % 4) * 4 + ch] =
                qw_host[(((size_t)j * c + ch) * kh + y) * kw + x];
  }
  std::vector<int8_t> packs;  // the same weights in the K order of the first-stage kernel (i8ie_stem.hip)
  if (L->path == PATH_B && c <= 3 && n % 32 == 0 && n <= 96 && (kh + 3) / 4 * ((kw + 3) / 4) * 48 <= 14 * 32) {
    L->KpadStem = i8ie_stem_kpad(kh, kw);
    packs.assign((size_t)n * L->KpadStem, 0);
    for (int j = 0; j < n; ++j)
      for (int ch = 0; ch < c; ++ch)
        for (int y = 0; y < kh; ++y)
          for (int x = 0; x < kw; ++x)
            packs[(size_t)j * L->KpadStem + i8ie_stem_kindex(kw, ch, y, x)] = qw_host[(((size_t)j * c + ch) * kh + y) * kw + x];
  }
  int rc = I8IE_OK;
  do {
    if ((rc = i8ie_malloc(ctx, (size_t)n * K, (void**)&L->qw)) != I8IE_OK) break;
    if ((rc = i8ie_malloc(ctx, (size_t)n, (void**)&L->qb)) != I8IE_OK) break;
    if ((rc = i8ie_malloc(ctx, (size_t)L->Npad * L->Kpad, (void**)&L->Bpack)) != I8IE_OK) break;
    if ((rc = i8ie_malloc(ctx, (size_t)n * 4, (void**)&L->wsum)) != I8IE_OK) break;
    if ((rc = i8ie_malloc(ctx, (size_t)n * 4, (void**)&L->oc)) != I8IE_OK) break;
    if ((rc = i8ie_malloc(ctx, (size_t)L->Npad * 4, (void**)&L->ocp)) != I8IE_OK) break;  // padded: vector loads
    if ((rc = i8ie_malloc(ctx, (size_t)L->Npad * 4, (void**)&L->biasf)) != I8IE_OK) break;
    if ((rc = i8ie_memset(ctx, L->ocp, 0, (size_t)L->Npad * 4)) != I8IE_OK) break;
    if ((rc = i8ie_memset(ctx, L->biasf, 0, (size_t)L->Npad * 4)) != I8IE_OK) break;
    if ((rc = i8ie_memcpy_h2d(ctx, L->qw, qw_host, (size_t)n * K)) != I8IE_OK) break;
    if ((rc = i8ie_memcpy_h2d(ctx, L->qb, qb_host, (size_t)n)) != I8IE_OK) break;
    if ((rc = i8ie_launch_pad_rows(ctx, L->qw, n, K, L->Bpack, L->Npad, L->Kpad, 0)) != I8IE_OK) break;
    if ((rc = i8ie_launch_offsets(ctx, conv, L->qw, nullptr, n, K, 1.0f, 0, nullptr, L->wsum)) != I8IE_OK) break;
    if (!pack2.empty()) {
      if ((rc = i8ie_malloc(ctx, pack2.size(), (void**)&L->Bpack2)) != I8IE_OK) break;
      if ((rc = i8ie_memcpy_h2d(ctx, L->Bpack2, pack2.data(), pack2.size())) != I8IE_OK) break;
    }
    if (!packs.empty()) {
      if ((rc = i8ie_malloc(ctx, packs.size(), (void**)&L->Bstem)) != I8IE_OK) break;
      if ((rc = i8ie_memcpy_h2d(ctx, L->Bstem, packs.data(), packs.size())) != I8IE_OK) break;
    }
  } while (0);
  if (rc != I8IE_OK) {
    i8ie_layer_destroy(L);
    return rc;
  }
  *out = L;
  return I8IE_OK;
}

int i8ie_linear_create(i8ie_ctx* ctx, const int8_t* qw_host, const int8_t* qb_host, int n, int k, float s_w,
                       i8ie_layer** out) {
  return layer_create(ctx, false, qw_host, qb_host, n, k, 0, 0, 0, 1, 0, s_w, out);
}

int i8ie_conv2d_create(i8ie_ctx* ctx, const int8_t* qw_host, const int8_t* qb_host, int kc, int c, int kh, int kw,
                       int stride, int pad, float s_w, i8ie_layer** out) {
  I8IE_REQUIRE(c > 0 && kh > 0 && kw > 0, "non-positive dimension");
  I8IE_REQUIRE(stride > 0, "stride must be positive");
  I8IE_REQUIRE(pad >= 0, "negative padding");
  return layer_create(ctx, true, qw_host, qb_host, kc, c * kh * kw, c, kh, kw, stride, pad, s_w, out);
}

int i8ie_layer_set_output_qparams(i8ie_layer* L, float s_out, uint8_t zp_out) {
  I8IE_REQUIRE(L != nullptr, "null layer");
  L->s_out = s_out;
  L->zp_out = zp_out;
  return I8IE_OK;
}

int i8ie_layer_get_output_qparams(const i8ie_layer* L, float* s_out, uint8_t* zp_out) {
  I8IE_REQUIRE(L && s_out && zp_out, "null argument");
  *s_out = L->s_out;
  *zp_out = L->zp_out;
  return I8IE_OK;
}

int i8ie_layer_preferred_layout(const i8ie_layer* L, int* layout) {
  I8IE_REQUIRE(L && layout, "null argument");
  const bool fast = L->conv && L->path != PATH_F && !force_fallback(L->ctx) && (L->n % 16 == 0);
  *layout = fast ? I8IE_LAYOUT_NHWC : I8IE_LAYOUT_NCHW;
  return I8IE_OK;
}

int i8ie_layer_padding(const i8ie_layer* L, int* pad) {
  I8IE_REQUIRE(L && pad, "null argument");
  *pad = L->conv ? L->pad : 0;
  return I8IE_OK;
}

static int layer_forward_impl(i8ie_layer* L, const uint8_t* in, int in_layout, int in_border, int m, int h, int w,
                              float s_in, uint8_t zp_in, int relu, int pool_k, int pool_s, uint8_t* out, int out_layout,
                              int out_border, int32_t* acc, float* out_f32);

int i8ie_layer_forward_fused(i8ie_layer* L, const uint8_t* in, int in_layout, int in_border, int m, int h, int w,
                             float s_in, uint8_t zp_in, int relu, uint8_t* out, int out_layout, int out_border,
                             int32_t* acc) {
  I8IE_REQUIRE(out != nullptr, "null argument");
  return layer_forward_impl(L, in, in_layout, in_border, m, h, w, s_in, zp_in, relu, 0, 0, out, out_layout, out_border,
                            acc, nullptr);
}

int i8ie_layer_forward_pool(i8ie_layer* L, const uint8_t* in, int in_layout, int in_border, int m, int h, int w,
                            float s_in, uint8_t zp_in, int relu, int pool_k, int pool_s, uint8_t* out, int out_layout,
                            int out_border, int32_t* acc) {
  I8IE_REQUIRE(L != nullptr && out != nullptr, "null argument");
  I8IE_REQUIRE(L->conv, "i8ie_layer_forward_pool: Conv2d layers only");
  I8IE_REQUIRE(pool_k > 0 && pool_s > 0, "kernel_size and stride must be positive");
  return layer_forward_impl(L, in, in_layout, in_border, m, h, w, s_in, zp_in, relu, pool_k, pool_s, out, out_layout,
                            out_border, acc, nullptr);
}

// dequantize(layer(x)) for a Linear layer: src/quantize_utils.cc:54-58 applied to the result of
// src/fully_connected.cc:22-52.  out_u8 may be null when the layer has at most 16 output features (the fused
// small-N kernel writes the FP32 values directly); otherwise it receives the u8 result as usual.
int i8ie_layer_forward_dequant(i8ie_layer* L, const uint8_t* in, int in_layout, int m, int h, int w, float s_in,
                               uint8_t zp_in, int relu, uint8_t* out_u8, float* out_f32) {
  I8IE_REQUIRE(L && out_f32, "null argument");
  I8IE_REQUIRE(!L->conv, "i8ie_layer_forward_dequant: Linear layers only");
  return layer_forward_impl(L, in, in_layout, 0, m, h, w, s_in, zp_in, relu, 0, 0, out_u8, I8IE_LAYOUT_NCHW, 0, nullptr,
                            out_f32);
}

// the first-stage kernel (i8ie_stem.hip) takes this conv layer at this output size (+ this pool behind it)?
static bool stem_takes(const i8ie_layer* L, const ConvGeom& cg, int pool_k, int pool_s) {
  return L->conv && L->path == PATH_B && L->Bstem != nullptr && !force_fallback(L->ctx) && L->ctx->variant != 11 &&
         i8ie_stem_supported(cg.c, cg.stride, L->n, cg.kh, cg.kw, cg.oh, cg.ow, pool_k, pool_s) != 0;
}

// would the patch-stationary kernel (i8ie_pconv.hip) take this conv launch, with this pool folded in and these
// re-biased layouts?  (it answers from the geometry and the batch; nothing is launched)
// (`out_border`: the border of the output the real call will write; the size limits of the kernel depend on it.  The layout
// negotiation queries below do not know it yet and ask with 0: layer_forward_impl asks again with the real one and runs the
// pool / the re-bias as launches of their own when the kernel then declines)
static bool pconv_probe(i8ie_layer* L, const ConvGeom& cg, int m, int in_border, int pool_k, int pool_s, bool a_s8, bool out_s8, int out_border = 0) {
  if (!L->conv || L->path != PATH_A || force_fallback(L->ctx)) return false;
  const int b = in_border > cg.pad ? in_border : cg.pad;
  I8ieIgemmCall q{};
  q.amode = 1; q.M = m * cg.oh * cg.ow; q.B = L->Bpack2; q.Kpad = L->Kpad2; q.Npad = L->Npad;
  q.Kchunks = L->K2 / 16; q.N = L->n; q.wcache = &L->wc; q.OH = cg.oh; q.OW = cg.ow;
  q.Hp = cg.h + 2 * b; q.Wp = cg.w + 2 * b; q.C = cg.c; q.KH = cg.kh; q.KW = cg.kw; q.sh = q.sw = cg.stride;
  q.a_bytes = (size_t)m * q.Hp * q.Wp * cg.c;
  q.pool_k = pool_k; q.pool_s = pool_s; q.a_s8 = a_s8 ? 1 : 0; q.out_s8 = out_s8 ? 1 : 0; q.ob = out_border;
  return q.a_bytes < i8ie_igemm_chunk_limit() && i8ie_pconv_takes(L->ctx, q) == 1;
}

static int layer_forward_impl(i8ie_layer* L, const uint8_t* in, int in_layout, int in_border, int m, int h, int w,
                              float s_in, uint8_t zp_in, int relu, int pool_k, int pool_s, uint8_t* out, int out_layout,
                              int out_border, int32_t* acc, float* out_f32) {
  I8IE_REQUIRE(L && in, "null argument");
  I8IE_REQUIRE(out != nullptr || (out_f32 != nullptr && !L->conv), "null output");
  I8IE_REQUIRE(m > 0, "non-positive batch");
  I8IE_REQUIRE(in_layout >= I8IE_LAYOUT_NCHW && in_layout <= I8IE_LAYOUT_NHWC_S8 && out_layout >= I8IE_LAYOUT_NCHW &&
                   out_layout <= I8IE_LAYOUT_NHWC_S8,
               "bad layout tag");
  I8IE_REQUIRE(L->conv || (in_layout != I8IE_LAYOUT_NHWC_S8 && out_layout != I8IE_LAYOUT_NHWC_S8),
               "the re-biased layout applies to conv layers only");
  I8IE_REQUIRE(in_border >= 0 && out_border >= 0, "negative border");
  I8IE_REQUIRE(in_layout != I8IE_LAYOUT_NCHW || in_border == 0, "only NHWC tensors carry a border");
  I8IE_REQUIRE(out_layout != I8IE_LAYOUT_NCHW || out_border == 0, "only NHWC tensors carry a border");
  i8ie_ctx* ctx = L->ctx;
  I8IE_HIP_TRY(hipSetDevice(ctx->device));
  I8IE_TRY(ensure_offsets(L, s_in, zp_in));

  if (!L->conv) {  // ---- Linear: row-major in / out -------------------------------------------------
    I8IE_REQUIRE(in_border == 0 && out_border == 0, "Linear tensors carry no border");
    // Rows that are a flattened NHWC activation [m][h][w][c] (the engine's layout between layers) instead of
    // the reference's flattened NCHW: same contraction with K walked in (h, w, c) order, so the weight panel
    // is permuted once per (c, h*w) and the input is used as it lies -- no transpose back to NCHW.  oc[] and
    // wsum[] are sums over all of K and keep the reference's accumulation order (they come from qw).
    const int hw = (in_layout == I8IE_LAYOUT_NHWC && h > 0 && w > 0) ? h * w : 1;
    const int8_t* panel = L->Bpack;
    if (hw > 1) {
      I8IE_REQUIRE(L->K % hw == 0, "Linear: in_features is not c * h * w for the given h, w");
      if (force_fallback(ctx) || L->K % 16 != 0 || !aligned16(in)) {
        // the any-geometry route wants reference order: transpose the input instead of the weights
        I8IE_TRY(i8ie_ws_reserve(ctx, (size_t)m * L->K + (size_t)m * L->Kpad + (size_t)8 * m * L->n * 4 + 4096));
        uint8_t* t = (uint8_t*)ctx->ws + i8ie_align_up((size_t)m * L->Kpad + (size_t)8 * m * L->n * 4, 256) + 512;
        I8IE_TRY(i8ie_launch_nhwc_to_nchw(ctx, in, t, m, L->K / hw, h, w, 0));
        return layer_forward_impl(L, t, I8IE_LAYOUT_NCHW, 0, m, 0, 0, s_in, zp_in, relu, 0, 0, out, out_layout, out_border,
                                  acc, out_f32);
      }
      if (L->Bperm == nullptr || L->perm_c != L->K / hw || L->perm_hw != hw) {
        if (L->Bperm == nullptr) I8IE_TRY(i8ie_malloc(ctx, (size_t)L->Npad * L->Kpad, (void**)&L->Bperm));
        I8IE_TRY(i8ie_launch_permute_k(ctx, L->Bpack, L->Bperm, L->Npad, L->Kpad, L->K, L->K / hw, hw));
        L->perm_c = L->K / hw;
        L->perm_hw = hw;
        L->bfrag_perm_valid = false;
      }
      panel = L->Bperm;
    }
    const bool need_pad = (L->K % 16 != 0) || !aligned16(in);
    if (!force_fallback(ctx) && !need_pad && L->n <= i8ie_smalln_max_features()) {
      // classifier head: one wave per row, dot4 + wavefront reduction, epilogue (and dequantize) fused
      I8ieSmallNCall sc{};
      sc.A = in; sc.lda = (size_t)L->K; sc.M = m; sc.K = L->K; sc.B = panel; sc.Kpad = L->Kpad; sc.N = L->n;
      sc.ocp = L->ocp; sc.biasf = L->biasf; sc.s_in = s_in; sc.s_w = L->s_w; sc.s_out = L->s_out;
      sc.zp_out = L->zp_out; sc.relu = relu; sc.out = out; sc.acc = acc; sc.out_f32 = out_f32;
      return i8ie_launch_linear_smalln(ctx, sc);
    }
    I8IE_REQUIRE(out != nullptr, "i8ie_layer_forward_dequant: this layer needs the u8 output buffer as well");
    if (out_f32 != nullptr) {  // general shape: the ordinary forward, then the dequantize kernel
      I8IE_TRY(layer_forward_impl(L, in, in_layout, 0, m, h, w, s_in, zp_in, relu, 0, 0, out, I8IE_LAYOUT_NCHW, 0, acc, nullptr));
      return i8ie_dequantize_u8_f32(ctx, out, out_f32, (int64_t)m * L->n, L->s_out, L->zp_out);
    }
#if defined(I8IE_DIAG)
    int sk_steps = 0, sk_slices = 0;
    // opt-in ($I8IE_SKINNY=1): once the tiled split-K kernel wrote its partial tiles as row segments it became
    // the faster one even at m = 125 (fc6 + fc7: 33 us against 42 us), see DESIGN.md
    static const bool use_skinny = std::getenv("I8IE_SKINNY") != nullptr;
    if (!force_fallback(ctx) && !need_pad && use_skinny && ctx->variant == 0 &&
        i8ie_skinny_plan(m, L->n, L->Kpad, &sk_steps, &sk_slices)) {
      // few input rows: activations' K slice resident in LDS, weights streamed once in fragment order
      const bool perm = panel == L->Bperm && L->Bperm != nullptr;
      void** frag = perm ? &L->Bfrag_perm : &L->Bfrag;
      if (*frag == nullptr) I8IE_TRY(i8ie_malloc(ctx, (size_t)L->Npad * L->Kpad, frag));
      if (perm ? !L->bfrag_perm_valid : !L->bfrag_valid) {
        I8IE_TRY(i8ie_launch_frag_pack(ctx, panel, *frag, L->Npad, L->Kpad));
        (perm ? L->bfrag_perm_valid : L->bfrag_valid) = true;
      }
      const size_t part_bytes = (size_t)sk_slices * m * L->n * 4;
      I8IE_TRY(i8ie_ws_reserve(ctx, part_bytes));
      I8IE_TRY(i8ie_launch_linear_skinny(ctx, in, (size_t)L->K, m, *frag, L->Kpad, L->Npad, L->n, L->ocp,
                                         (int32_t*)ctx->ws, sk_steps, sk_slices));
      return i8ie_launch_splitk_reduce(ctx, (const int32_t*)ctx->ws, sk_slices, m, L->n, L->biasf, s_in, L->s_w,
                                       L->s_out, L->zp_out, relu, out, acc);
    }
#endif
    if (force_fallback(ctx)) {
      if (need_pad) I8IE_TRY(i8ie_ws_reserve(ctx, (size_t)m * L->Kpad));
      I8IE_TRY(linear_run_v1(ctx, in, m, L->K, L->Bpack, L->Kpad, L->qb, L->n, L->oc, L->wsum, s_in, L->s_w, L->s_out,
                             L->zp_out, out, acc, (uint8_t*)ctx->ws));
      if (relu) I8IE_TRY(i8ie_relu_u8(ctx, out, out, (int64_t)m * L->n, L->zp_out));
      return I8IE_OK;
    }
    I8ieIgemmCall c{};
    // few rows: the one-launch kernel of i8ie_flin.hip (variant 11 keeps the tiled split-K kernel, for comparison)
    const bool flin = !need_pad && ctx->variant != 11 && L->K % 16 == 0 && i8ie_flin_wants(m, L->n, L->Kpad, ctx->variant == 80 || ctx->variant == 81);
    // split K when the output has too few tiles to fill the chip (small batch, or few features)
    const long tiles_m = (m + 127) / 128, blocks_est = tiles_m * ((L->n + 63) / 64);
    const int nk = L->Kpad / 128;
    int ksplit = 1;
#if defined(I8IE_DIAG)
    const bool lgemm = !need_pad && ctx->variant == 82 && i8ie_lgemm_wants(m, L->n, L->K, L->Kpad) && aligned16(out) &&
                       (size_t)m * L->K < i8ie_igemm_chunk_limit();
#else
    const bool lgemm = false;
#endif
    // many rows: the one-launch kernel of i8ie_mlin.hip (variants 3 / 11 keep the tiled kernel, for comparison)
    const bool mlin = !need_pad && !flin && !lgemm && ctx->variant != 11 && ctx->variant != 3 && L->K % 16 == 0 && aligned16(out) &&
                      L->Npad % 128 == 0 && i8ie_mlin_wants(m, L->n, L->Kpad, ctx->variant == 83 || ctx->variant == 84 || ctx->variant == 85);
    if (blocks_est < 256 && nk >= 4 && !flin && !lgemm && !mlin) {
      ksplit = (int)((512 + blocks_est - 1) / blocks_est);
      if (ksplit > 8) ksplit = 8;
      if (ksplit > nk / 2) ksplit = nk / 2;
    }
    const size_t pad_bytes = need_pad ? i8ie_align_up((size_t)m * L->Kpad, 256) : 0;
    const size_t part_bytes = ksplit > 1 ? (size_t)ksplit * m * L->n * 4 : 0;
    if (pad_bytes + part_bytes) I8IE_TRY(i8ie_ws_reserve(ctx, pad_bytes + part_bytes));
    c.ksplit = ksplit;
    c.partial = ksplit > 1 ? (int32_t*)((uint8_t*)ctx->ws + pad_bytes) : nullptr;
    if (need_pad) {
      I8IE_TRY(i8ie_launch_pad_rows(ctx, in, m, L->K, ctx->ws, m, L->Kpad, 0));
      c.A = (const uint8_t*)ctx->ws;
      c.lda = L->Kpad;
      c.Kchunks = L->Kpad / 16;
    } else {
      c.A = in;
      c.lda = L->K;
      c.Kchunks = L->K / 16;
    }
    c.a_bytes = (size_t)m * c.lda;
    c.amode = 0; c.M = m;
    c.B = panel; c.Kpad = L->Kpad; c.Npad = L->Npad; c.N = L->n; c.ocp = L->ocp; c.biasf = L->biasf;
    c.s_in = s_in; c.s_w = L->s_w; c.s_out = L->s_out; c.zp_out = L->zp_out; c.relu = relu;
    c.out = out; c.ob = 0; c.acc = acc; c.Ktrue = L->K;
    if (flin) return i8ie_flin_launch(ctx, c);
    if (mlin) return i8ie_mlin_launch(ctx, c);
#if defined(I8IE_DIAG)
    if (lgemm) {  // (experiment: tools/diag/csrc/i8ie_lgemm.hip)
      c.wcache = &L->wc;
      return i8ie_lgemm_launch(ctx, c, panel == L->Bperm && L->Bperm != nullptr);
    }
#endif
    return i8ie_igemm_launch(ctx, c);
  }

  // ---- Conv2d --------------------------------------------------------------------------------
  ConvGeom cg;
  I8IE_TRY(conv_geom(L->c, h, w, L->n, L->kh, L->kw, L->stride, L->pad, &cg));
  const bool pool = i8ie_is_pool(pool_k, pool_s);
  if (pool) I8IE_REQUIRE(pool_k <= cg.oh && pool_k <= cg.ow, "max-pool window larger than the convolution's output");
  const bool in_s8 = in_layout == I8IE_LAYOUT_NHWC_S8, out_s8 = out_layout == I8IE_LAYOUT_NHWC_S8;
  // Which kernel folds what: the first-stage kernel (path B) pools and can store re-biased; the patch-stationary kernel
  // (path A) pools, reads and stores re-biased -- when it takes the launch at all (batch, geometry, LDS: asked below).
  // Everything else: the plain call, with the pool / the re-bias as launches of their own around it.
  const bool stem = stem_takes(L, cg, pool ? pool_k : 0, pool_s) && (acc == nullptr || aligned16(acc));
  bool pconv = false;
  if (!stem && (pool || in_s8 || out_s8) && out_layout != I8IE_LAYOUT_NCHW && aligned16(in) && aligned16(out) &&
      (acc == nullptr || aligned16(acc)))
    pconv = pconv_probe(L, cg, m, in_border, pool ? pool_k : 0, pool_s, in_s8, out_s8, out_border);
  if (in_s8 && !pconv) {  // nobody reads the re-biased bytes as they are: plain copy first
    const size_t bytes = (size_t)m * (cg.h + 2 * in_border) * (cg.w + 2 * in_border) * cg.c;
    uint8_t* tmp = nullptr;
    I8IE_TRY(i8ie_malloc(ctx, bytes, (void**)&tmp));
    int rc = i8ie_rebias_u8(ctx, in, tmp, (int64_t)bytes);
    if (rc == I8IE_OK)
      rc = layer_forward_impl(L, tmp, I8IE_LAYOUT_NHWC, in_border, m, h, w, s_in, zp_in, relu, pool_k, pool_s, out, out_layout,
                              out_border, acc, nullptr);
    i8ie_free(ctx, tmp);
    return rc;
  }
  if (out_s8 && !pconv && !stem) {  // plain result into a temporary, re-biased, then laid into `out` with its border
    const int oph = pool ? (cg.oh - pool_k) / pool_s + 1 : cg.oh, opw = pool ? (cg.ow - pool_k) / pool_s + 1 : cg.ow;
    const size_t bytes = (size_t)m * L->n * oph * opw;
    I8IE_REQUIRE(L->n % 16 == 0, "re-biased NHWC output needs out features % 16 == 0");
    uint8_t* tmp = nullptr;
    I8IE_TRY(i8ie_malloc(ctx, bytes, (void**)&tmp));
    int rc = layer_forward_impl(L, in, in_layout, in_border, m, h, w, s_in, zp_in, relu, pool_k, pool_s, tmp, I8IE_LAYOUT_NHWC, 0,
                                acc, nullptr);
    if (rc == I8IE_OK) rc = i8ie_rebias_u8(ctx, tmp, tmp, (int64_t)bytes);
    if (rc == I8IE_OK) rc = i8ie_launch_reborder(ctx, tmp, out, m, L->n, oph, opw, 0, out_border, L->zp_out ^ 0x80);
    i8ie_free(ctx, tmp);
    return rc;
  }
  if (pool && !stem && !pconv) {
    // no kernel fuses this pool: max_pool2d<u8> (src/functional.cc:36-64) as its own launch behind the convolution
    const int oph = (cg.oh - pool_k) / pool_s + 1, opw = (cg.ow - pool_k) / pool_s + 1;
    const bool nhwc = out_layout == I8IE_LAYOUT_NHWC && L->n % 16 == 0;
    uint8_t* tmp = nullptr;
    I8IE_TRY(i8ie_malloc(ctx, (size_t)m * L->n * cg.oh * cg.ow, (void**)&tmp));
    int rc = layer_forward_impl(L, in, in_layout, in_border, m, h, w, s_in, zp_in, relu, 0, 0, tmp,
                                nhwc ? I8IE_LAYOUT_NHWC : I8IE_LAYOUT_NCHW, 0, acc, nullptr);
    if (rc == I8IE_OK) {
      if (nhwc) {
        rc = i8ie_launch_maxpool_nhwc(ctx, tmp, 0, out, out_border, m, L->n, cg.oh, cg.ow, pool_k, pool_s, 0);
      } else if (out_layout == I8IE_LAYOUT_NCHW) {
        rc = i8ie_maxpool2d_u8(ctx, tmp, out, m, L->n, cg.oh, cg.ow, pool_k, pool_s);
      } else {  // NHWC result with channels % 16 != 0: pool in NCHW, then lay out
        uint8_t* tmp2 = nullptr;
        rc = i8ie_malloc(ctx, (size_t)m * L->n * oph * opw, (void**)&tmp2);
        if (rc == I8IE_OK) rc = i8ie_maxpool2d_u8(ctx, tmp, tmp2, m, L->n, cg.oh, cg.ow, pool_k, pool_s);
        if (rc == I8IE_OK) rc = i8ie_launch_nchw_to_nhwc(ctx, tmp2, out, m, L->n, oph, opw, out_border);
        i8ie_free(ctx, tmp2);
      }
    }
    i8ie_free(ctx, tmp);
    return rc;
  }
  const int ph = pool ? (cg.oh - pool_k) / pool_s + 1 : cg.oh, pw = pool ? (cg.ow - pool_k) / pool_s + 1 : cg.ow;
  const size_t in_bytes = (size_t)m * cg.c * cg.h * cg.w;
  const size_t out_bytes = (size_t)m * cg.kc * ph * pw;
  const int path = force_fallback(ctx) ? PATH_F : L->path;

  if (path == PATH_F) {
    const int ipc = chunk_images(cg, m);
    const size_t col_bytes = i8ie_align_up((size_t)ipc * cg.oh * cg.ow * cg.Kpad, 256);
    const size_t a_bytes = in_layout == I8IE_LAYOUT_NHWC ? i8ie_align_up(in_bytes, 256) : 0;
    const size_t o_bytes = out_layout == I8IE_LAYOUT_NHWC ? i8ie_align_up(out_bytes, 256) : 0;
    I8IE_TRY(i8ie_ws_reserve(ctx, col_bytes + a_bytes + o_bytes));
    uint8_t* ws = (uint8_t*)ctx->ws;
    uint8_t* col = ws;
    const uint8_t* src = in;
    if (a_bytes) {
      I8IE_TRY(i8ie_launch_nhwc_to_nchw(ctx, in, ws + col_bytes, m, cg.c, cg.h, cg.w, in_border));
      src = ws + col_bytes;
    }
    uint8_t* dst = o_bytes ? ws + col_bytes + a_bytes : out;
    I8IE_TRY(conv_run_v1(ctx, src, m, cg, L->Bpack, L->oc, L->wsum, zp_in, s_in, L->s_w, L->s_out, L->zp_out, dst, acc,
                         col, ipc));
    if (relu) I8IE_TRY(i8ie_relu_u8(ctx, dst, dst, (int64_t)out_bytes, L->zp_out));
    if (o_bytes) {
      I8IE_TRY(i8ie_launch_nchw_to_nhwc(ctx, dst, out, m, cg.kc, cg.oh, cg.ow, out_border));
    }
    return I8IE_OK;
  }

  I8ieIgemmCall c{};
  c.amode = 1; c.M = m * cg.oh * cg.ow;
  c.B = L->Bpack2; c.Kpad = L->Kpad2; c.Npad = L->Npad; c.Kchunks = L->K2 / 16; c.N = L->n; c.ocp = L->ocp;
  c.biasf = nullptr; c.wcache = &L->wc;
  c.s_in = s_in; c.s_w = L->s_w; c.s_out = L->s_out; c.zp_out = L->zp_out; c.relu = relu;
  c.acc = acc; c.Ktrue = cg.K; c.OH = cg.oh; c.OW = cg.ow;
  if (pconv) {  // (the patch-stationary kernel said it takes this launch as it is)
    c.pool_k = pool ? pool_k : 0; c.pool_s = pool_s; c.a_s8 = in_s8 ? 1 : 0; c.out_s8 = out_s8 ? 1 : 0;
  }
  const size_t o_bytes = out_layout == I8IE_LAYOUT_NCHW ? i8ie_align_up(out_bytes, 256) : 0;

  if (path == PATH_A) {
    const int Hp = cg.h + 2 * cg.pad, Wp = cg.w + 2 * cg.pad;
    const bool direct = in_layout != I8IE_LAYOUT_NCHW && in_border >= cg.pad && aligned16(in);
    const size_t a_bytes = direct ? 0 : i8ie_align_up((size_t)m * Hp * Wp * cg.c, 256);
    I8IE_TRY(i8ie_ws_reserve(ctx, a_bytes + o_bytes + 256));
    uint8_t* ws = (uint8_t*)ctx->ws;
    if (direct) {
      const int iHp = cg.h + 2 * in_border, iWp = cg.w + 2 * in_border, d = in_border - cg.pad;
      const size_t shift = ((size_t)d * iWp + d) * cg.c;
      c.A = in + shift;
      c.a_bytes = (size_t)m * iHp * iWp * cg.c - shift;
      c.Hp = iHp; c.Wp = iWp;
    } else {
      if (in_layout != I8IE_LAYOUT_NCHW) {  // (re-biased bytes are copied as they are; their border value is zp ^ 0x80)
        I8IE_TRY(i8ie_launch_reborder(ctx, in, ws, m, cg.c, cg.h, cg.w, in_border, cg.pad, in_s8 ? (zp_in ^ 0x80) : zp_in));
      } else {
        if (cg.pad > 0) I8IE_HIP_TRY(hipMemsetAsync(ws, zp_in, (size_t)m * Hp * Wp * cg.c, ctx->stream));
        I8IE_TRY(i8ie_launch_nchw_to_nhwc(ctx, in, ws, m, cg.c, cg.h, cg.w, cg.pad));
      }
      c.A = ws;
      c.a_bytes = (size_t)m * Hp * Wp * cg.c;
      c.Hp = Hp; c.Wp = Wp;
    }
    c.C = cg.c; c.KH = cg.kh; c.KW = cg.kw; c.sh = c.sw = cg.stride;
    c.out = o_bytes ? ws + a_bytes : out;
  } else {  // PATH_B: small-C, stride % 4 == 0
    const int Hp = (cg.oh - 1) * cg.stride + cg.kh;
    const int Wg = (cg.ow - 1) * (cg.stride / 4) + L->kwg;
    const size_t t_bytes = in_layout != I8IE_LAYOUT_NCHW ? i8ie_align_up(in_bytes, 256) : 0;
    const size_t r_bytes = i8ie_align_up((size_t)m * Hp * Wg * 16, 256);
    if (stem && (acc == nullptr || aligned16(acc))) {
      // first-stage kernel (i8ie_stem.hip): space-to-depth image, conv (+ relu) (+ max-pool) in one contraction launch
      const size_t s_bytes = i8ie_align_up(i8ie_stem_scratch_bytes(m, cg.kh, cg.kw, cg.stride, cg.oh, cg.ow), 256);
      I8IE_TRY(i8ie_ws_reserve(ctx, t_bytes + s_bytes + o_bytes + 256));
      uint8_t* ws = (uint8_t*)ctx->ws;
      const uint8_t* src = in;
      if (t_bytes) {
        I8IE_TRY(i8ie_launch_nhwc_to_nchw(ctx, in, ws, m, cg.c, cg.h, cg.w, in_border));
        src = ws;
      }
      uint8_t* dst = o_bytes ? ws + t_bytes + s_bytes : out;
      I8ieStemCall f{};
      f.x = nullptr; f.xu8 = src; f.scratch = ws + t_bytes;
      f.n = m; f.c = cg.c; f.h = cg.h; f.w = cg.w; f.q_scale = s_in; f.q_zp = zp_in;
      f.KH = cg.kh; f.KW = cg.kw; f.stride = cg.stride; f.pad = cg.pad; f.OH = cg.oh; f.OW = cg.ow;
      f.B = L->Bstem; f.Kpad = L->KpadStem; f.N = L->n; f.ocp = L->ocp;
      f.s_in = s_in; f.s_w = L->s_w; f.s_out = L->s_out; f.zp_out = L->zp_out; f.relu = relu;
      f.pool_k = pool ? pool_k : 0; f.pool_s = pool_s;
      f.out = dst; f.ob = o_bytes ? 0 : out_border; f.out_s8 = out_s8 ? 1 : 0; f.acc = acc;
      I8IE_TRY(i8ie_stem_launch(ctx, f));
      if (o_bytes) I8IE_TRY(i8ie_launch_nhwc_to_nchw(ctx, dst, out, m, cg.kc, ph, pw, 0));
      return I8IE_OK;
    }
    I8IE_TRY(i8ie_ws_reserve(ctx, t_bytes + r_bytes + o_bytes + 256));
    uint8_t* ws = (uint8_t*)ctx->ws;
    const uint8_t* src = in;
    if (t_bytes) {
      I8IE_TRY(i8ie_launch_nhwc_to_nchw(ctx, in, ws, m, cg.c, cg.h, cg.w, in_border));
      src = ws;
    }
    uint8_t* rep = ws + t_bytes;
    const bool wstat = (acc == nullptr || (L->n % 4 == 0 && aligned16(acc))) && i8ie_first_supported(cg.c, cg.stride, L->n, L->K2, cg.kh, L->kwg, cg.ow);
    I8IE_TRY(i8ie_launch_repack_smallc(ctx, src, rep, m, cg.c, cg.h, cg.w, Hp, Wg, cg.pad, cg.pad, zp_in, wstat));
    if (wstat) {
      // weights-stationary small-C kernel (i8ie_first.hip) on the grouped image
      uint8_t* dst = o_bytes ? ws + t_bytes + r_bytes : out;
      const int ob = o_bytes ? 0 : out_border;
      I8ieFirstCall f{};
      f.x = nullptr; f.grouped = rep; f.scratch = nullptr;
      f.n = m; f.c = cg.c; f.h = cg.h; f.w = cg.w; f.q_scale = s_in; f.q_zp = zp_in;
      f.KH = cg.kh; f.KW = cg.kw; f.KWG = L->kwg; f.stride = cg.stride; f.pad = cg.pad; f.OH = cg.oh; f.OW = cg.ow;
      f.B = L->Bpack2; f.Kpad = L->Kpad2; f.K2 = L->K2; f.N = L->n; f.ocp = L->ocp;
      f.s_in = s_in; f.s_w = L->s_w; f.s_out = L->s_out; f.zp_out = L->zp_out; f.relu = relu;
      f.out = dst; f.ob = ob; f.acc = acc;
      I8IE_TRY(i8ie_first_launch(ctx, f));
      if (o_bytes) I8IE_TRY(i8ie_launch_nhwc_to_nchw(ctx, dst, out, m, cg.kc, cg.oh, cg.ow, 0));
      return I8IE_OK;
    }
    c.A = rep;
    c.a_bytes = (size_t)m * Hp * Wg * 16;
    c.Hp = Hp; c.Wp = Wg; c.C = 16; c.KH = cg.kh; c.KW = L->kwg;
    c.sh = cg.stride; c.sw = cg.stride / 4;
    c.out = o_bytes ? ws + t_bytes + r_bytes : out;
  }
  c.ob = o_bytes ? 0 : out_border;
  I8IE_TRY(i8ie_igemm_launch(ctx, c));
  if (o_bytes) I8IE_TRY(i8ie_launch_nhwc_to_nchw(ctx, c.out, out, m, cg.kc, cg.oh, cg.ow, 0));
  return I8IE_OK;
}

int i8ie_layer_accepts_f32_input(const i8ie_layer* L, int h, int w, int* yes) {
  I8IE_REQUIRE(L && yes, "null argument");
  *yes = 0;
  if (!L->conv || L->path != PATH_B || force_fallback(L->ctx)) return I8IE_OK;
  ConvGeom cg;
  if (conv_geom(L->c, h, w, L->n, L->kh, L->kw, L->stride, L->pad, &cg) != I8IE_OK) return I8IE_OK;
  *yes = (stem_takes(L, cg, 0, 0) || i8ie_first_supported(L->c, L->stride, L->n, L->K2, L->kh, L->kwg, cg.ow)) ? 1 : 0;
  return I8IE_OK;
}

int i8ie_layer_fuses_pool(const i8ie_layer* L, int m, int h, int w, int pool_k, int pool_s, int* yes) {
  I8IE_REQUIRE(L && yes, "null argument");
  *yes = 0;
  if (!L->conv || pool_k < 1 || pool_s < 1 || m < 1) return I8IE_OK;
  ConvGeom cg;
  if (conv_geom(L->c, h, w, L->n, L->kh, L->kw, L->stride, L->pad, &cg) != I8IE_OK) return I8IE_OK;
  if (pool_k > cg.oh || pool_k > cg.ow) return I8IE_OK;
  *yes = (stem_takes(L, cg, pool_k, pool_s) || pconv_probe(const_cast<i8ie_layer*>(L), cg, m, cg.pad, pool_k, pool_s, false, false)) ? 1 : 0;
  return I8IE_OK;
}

int i8ie_layer_rebiased_io(const i8ie_layer* L, int m, int h, int w, int pool_k, int pool_s, int* reads, int* stores) {
  I8IE_REQUIRE(L && reads && stores, "null argument");
  *reads = *stores = 0;
  if (!L->conv || m < 1) return I8IE_OK;
  ConvGeom cg;
  if (conv_geom(L->c, h, w, L->n, L->kh, L->kw, L->stride, L->pad, &cg) != I8IE_OK) return I8IE_OK;
  const bool pool = i8ie_is_pool(pool_k, pool_s);
  if (pool && (pool_s < 1 || pool_k > cg.oh || pool_k > cg.ow)) return I8IE_OK;
  i8ie_layer* Lm = const_cast<i8ie_layer*>(L);
  *reads = pconv_probe(Lm, cg, m, cg.pad, pool ? pool_k : 0, pool_s, true, false) ? 1 : 0;
  *stores = (stem_takes(L, cg, pool ? pool_k : 0, pool_s) || pconv_probe(Lm, cg, m, cg.pad, pool ? pool_k : 0, pool_s, false, true)) ? 1 : 0;
  return I8IE_OK;
}

int i8ie_layer_forward_f32_input_pool(i8ie_layer* L, const float* in, int m, int h, int w, float q_scale, uint8_t q_zp,
                                      int relu, int pool_k, int pool_s, uint8_t* out, int out_layout, int out_border,
                                      int32_t* acc) {
  I8IE_REQUIRE(L && in && out, "null argument");
  I8IE_REQUIRE(out_layout == I8IE_LAYOUT_NHWC || out_layout == I8IE_LAYOUT_NHWC_S8, "the fused first layer writes NHWC (plain or re-biased)");
  I8IE_REQUIRE(m > 0 && out_border >= 0, "bad argument");
  int yes = 0;
  I8IE_TRY(i8ie_layer_accepts_f32_input(L, h, w, &yes));
  if (!yes) {
    i8ie_set_error("i8ie_layer_forward_f32_input: layer/geometry not supported by the fused first-layer kernels");
    return I8IE_ERR_STATE;
  }
  i8ie_ctx* ctx = L->ctx;
  I8IE_HIP_TRY(hipSetDevice(ctx->device));
  I8IE_REQUIRE((reinterpret_cast<uintptr_t>(out) & 15u) == 0, "output must be 16-byte aligned");
  I8IE_REQUIRE(acc == nullptr || aligned16(acc), "accumulator buffer must be 16-byte aligned");
  I8IE_TRY(ensure_offsets(L, q_scale, q_zp));
  ConvGeom cg;
  I8IE_TRY(conv_geom(L->c, h, w, L->n, L->kh, L->kw, L->stride, L->pad, &cg));
  const bool pool = i8ie_is_pool(pool_k, pool_s);
  if (pool) {
    I8IE_REQUIRE(pool_s > 0 && pool_k <= cg.oh && pool_k <= cg.ow, "max-pool window larger than the convolution's output");
  }
  if (stem_takes(L, cg, pool ? pool_k : 0, pool_s)) {
    I8IE_TRY(i8ie_ws_reserve(ctx, i8ie_stem_scratch_bytes(m, cg.kh, cg.kw, cg.stride, cg.oh, cg.ow) + 256));
    I8ieStemCall f{};
    f.x = in; f.xu8 = nullptr; f.scratch = (uint8_t*)ctx->ws;
    f.n = m; f.c = cg.c; f.h = h; f.w = w; f.q_scale = q_scale; f.q_zp = q_zp;
    f.KH = cg.kh; f.KW = cg.kw; f.stride = cg.stride; f.pad = cg.pad; f.OH = cg.oh; f.OW = cg.ow;
    f.B = L->Bstem; f.Kpad = L->KpadStem; f.N = L->n; f.ocp = L->ocp;
    f.s_in = q_scale; f.s_w = L->s_w; f.s_out = L->s_out; f.zp_out = L->zp_out; f.relu = relu;
    f.pool_k = pool ? pool_k : 0; f.pool_s = pool_s;
    f.out = out; f.ob = out_border; f.out_s8 = out_layout == I8IE_LAYOUT_NHWC_S8 ? 1 : 0; f.acc = acc;
    return i8ie_stem_launch(ctx, f);
  }
  if (pool || out_layout == I8IE_LAYOUT_NHWC_S8) {
    // the older first-layer kernel neither pools nor stores re-biased: conv (+ relu) into a temporary, then the rest
    I8IE_REQUIRE(L->n % 16 == 0, "i8ie_layer_forward_f32_input_pool: out features % 16 != 0");
    const int oph = pool ? (cg.oh - pool_k) / pool_s + 1 : cg.oh, opw = pool ? (cg.ow - pool_k) / pool_s + 1 : cg.ow;
    uint8_t *tmp = nullptr, *tmp2 = nullptr;
    I8IE_TRY(i8ie_malloc(ctx, (size_t)m * L->n * cg.oh * cg.ow, (void**)&tmp));
    int rc = i8ie_layer_forward_f32_input_pool(L, in, m, h, w, q_scale, q_zp, relu, 0, 0, tmp, I8IE_LAYOUT_NHWC, 0, acc);
    if (rc == I8IE_OK && out_layout == I8IE_LAYOUT_NHWC) {
      rc = i8ie_launch_maxpool_nhwc(ctx, tmp, 0, out, out_border, m, L->n, cg.oh, cg.ow, pool_k, pool_s, 0);
    } else if (rc == I8IE_OK) {
      uint8_t* plain = tmp;
      if (pool) {
        rc = i8ie_malloc(ctx, (size_t)m * L->n * oph * opw, (void**)&tmp2);
        if (rc == I8IE_OK) rc = i8ie_launch_maxpool_nhwc(ctx, tmp, 0, tmp2, 0, m, L->n, cg.oh, cg.ow, pool_k, pool_s, 0);
        plain = tmp2;
      }
      if (rc == I8IE_OK) rc = i8ie_rebias_u8(ctx, plain, plain, (int64_t)m * L->n * oph * opw);
      if (rc == I8IE_OK) rc = i8ie_launch_reborder(ctx, plain, out, m, L->n, oph, opw, 0, out_border, L->zp_out ^ 0x80);
    }
    i8ie_free(ctx, tmp);
    if (tmp2) i8ie_free(ctx, tmp2);
    return rc;
  }
  I8IE_TRY(i8ie_ws_reserve(ctx, i8ie_first_scratch_bytes(m, cg.kh, L->kwg, cg.stride, cg.oh, cg.ow) + 256));
  I8ieFirstCall c{};
  c.x = in; c.grouped = nullptr; c.scratch = (uint8_t*)ctx->ws;
  c.n = m; c.c = cg.c; c.h = h; c.w = w; c.q_scale = q_scale; c.q_zp = q_zp;
  c.KH = cg.kh; c.KW = cg.kw; c.KWG = L->kwg; c.stride = cg.stride; c.pad = cg.pad; c.OH = cg.oh; c.OW = cg.ow;
  c.B = L->Bpack2; c.Kpad = L->Kpad2; c.K2 = L->K2; c.N = L->n; c.ocp = L->ocp;
  c.s_in = q_scale; c.s_w = L->s_w; c.s_out = L->s_out; c.zp_out = L->zp_out; c.relu = relu;
  c.out = out; c.ob = out_border; c.acc = acc;
  return i8ie_first_launch(ctx, c);
}

int i8ie_layer_forward_f32_input(i8ie_layer* L, const float* in, int m, int h, int w, float q_scale, uint8_t q_zp,
                                 int relu, uint8_t* out, int out_border, int32_t* acc) {
  return i8ie_layer_forward_f32_input_pool(L, in, m, h, w, q_scale, q_zp, relu, 0, 0, out, I8IE_LAYOUT_NHWC, out_border, acc);
}

int i8ie_layer_forward(i8ie_layer* L, const uint8_t* in, int m, int h, int w, float s_in, uint8_t zp_in, uint8_t* out,
                       int32_t* acc) {
  return i8ie_layer_forward_fused(L, in, I8IE_LAYOUT_NCHW, 0, m, h, w, s_in, zp_in, 0, out, I8IE_LAYOUT_NCHW, 0, acc);
}

int i8ie_layer_destroy(i8ie_layer* L) {
  if (!L) return I8IE_OK;
  i8ie_ctx* ctx = L->ctx;
  i8ie_free(ctx, L->qw);
  i8ie_free(ctx, L->qb);
  i8ie_free(ctx, L->Bpack);
  i8ie_free(ctx, L->Bpack2);
  if (L->Bstem) i8ie_free(ctx, L->Bstem);
  for (const I8ieWCache::Ent& e : L->wc.ents) i8ie_free(ctx, e.buf);
  i8ie_free(ctx, L->wsum);
  i8ie_free(ctx, L->oc);
  i8ie_free(ctx, L->ocp);
  if (L->Bperm) i8ie_free(ctx, L->Bperm);
  if (L->Bfrag) i8ie_free(ctx, L->Bfrag);
  if (L->Bfrag_perm) i8ie_free(ctx, L->Bfrag_perm);
  i8ie_free(ctx, L->biasf);
  delete L;
  return I8IE_OK;
}

// quantize_weight, src/layer.cc:6-26 (host side, one-shot at convert())
int i8ie_quantize_weight(const float* w, int64_t nw, const float* b, int64_t nb, int8_t* qw, int8_t* qb,
                         float* scale_out) {
  I8IE_REQUIRE(w && b && qw && qb && scale_out, "null argument");
  I8IE_REQUIRE(nw > 0 && nb > 0, "empty tensor");
  float mx = -3.402823466e+38f, mn = 3.402823466e+38f;
  for (int64_t i = 0; i < nw; ++i) {
    mn = w[i] < mn ? w[i] : mn;
    mx = w[i] > mx ? w[i] : mx;
  }
  for (int64_t i = 0; i < nb; ++i) {
    mn = b[i] < mn ? b[i] : mn;
    mx = b[i] > mx ? b[i] : mx;
  }
  const float s = (mx - mn) / 127;
  for (int64_t i = 0; i < nw; ++i) qw[i] = (int8_t)(int32_t)(w[i] / s);
  for (int64_t i = 0; i < nb; ++i) qb[i] = (int8_t)(int32_t)(b[i] / s);
  *scale_out = s;
  return I8IE_OK;
}

}  // extern "C"
