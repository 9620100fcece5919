/*
 * i8ie_oracle.c -- CPU restatement of the reference's INT8 hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  This file is the parity oracle for the HIP
 * product path.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load it; the product path never does.
 *
 * Reference: t0037799/INT8InferenceEngine (a CPU MKL/OpenMP engine).  Every
 * function cites the reference file:line it restates (paths relative to
 * /root/reference).  Arithmetic rules: "float" is
 * IEEE fp32, round-to-nearest-even, no FMA contraction, no reassociation;
 * float->int casts truncate toward zero.  Build with -ffp-contract=off.
 *
 * PINNING STATUS
 *   pinned against the reference's own compiled code (oracle/_ref, built from
 *   src/quantize_utils.cc + src/functional.cc + src/calibrator.cc where they lie):
 *     orc_quantize_f32_u8, orc_dequantize_u8_f32, orc_down_scale,
 *     orc_relu_u8, orc_max_pool2d_u8, orc_calib_range   (tests/golden/ref_*.npz)
 *   pinned against the reference's GEMM provider (Intel MKL cblas_gemm_s8u8s32, the call at
 *   src/conv2d.cc:131-133 / src/fully_connected.cc:39-41; MKL 2021.4 shared objects are in this image and
 *   are called directly through ctypes by tests/golden/make_golden_mkl.py -- nothing of the reference is
 *   built for this): orc_gemm_u8s8s32                   (tests/golden/mkl_gemm_s8u8s32.npz)
 *   PARITY UNPINNED by a reference run (src/conv2d.cc, src/fully_connected.cc
 *   and src/layer.cc include mkl.h, which this image lacks, so those
 *   translation units are unbuildable here; the reference's own tests hold no
 *   golden vectors for them):
 *     orc_quantize_weight, orc_conv_offsets, orc_linear_offsets, orc_im2col_u8,
 *     and the composition of the pinned pieces in orc_conv2d_u8 / orc_linear_u8
 *   For these the integer contraction is exact by definition
 *   (C = sum A*B + oc) and is cross-checked in tests/ against an independent
 *   int64 numpy/torch formulation; the float epilogue they feed is the pinned
 *   orc_down_scale.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>

#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------ a1 --
 * src/quantize_utils.cc:44-52  quantize(Tensor<float>&, float scale, u8 zp)
 *   out[i] = in[i] / scale + zp;   (float -> u8 C cast, NO clamp)
 * Out-of-range float->u8 is UB in C++; the compiled reference (x86-64, gcc)
 * behaves as ((int32)trunc(t)) & 0xFF, which is what we restate.            */
void orc_quantize_f32_u8(const float* x, uint8_t* q, int64_t n, float scale,
                         uint8_t zp) {
  for (int64_t i = 0; i < n; ++i) {
    float t = x[i] / scale + (float)zp;
    q[i] = (uint8_t)((int32_t)t & 0xFF);
  }
}

/* ------------------------------------------------------------------ a6 --
 * src/quantize_utils.cc:38-42,54-58  dequantize(Tensor<u8>&)
 *   M[i] = (Q[i] - zp) * scale;    (int subtraction, then int->float, * scale) */
void orc_dequantize_u8_f32(const uint8_t* q, float* x, int64_t n, float scale,
                           uint8_t zp) {
  for (int64_t i = 0; i < n; ++i) {
    x[i] = (float)((int)q[i] - (int)zp) * scale;
  }
}

/* ------------------------------------------------------------------ a5 --
 * src/quantize_utils.cc:27-36  down_scale(u8* M, int* Q, size, sa, sb, sc, zp_c)
 *   float dequant = Q[i] * sa * sb;  float quant = dequant / sc + zp_c;
 *   M[i] = quant >= 255 ? 255 : quant < 0 ? 0 : quant;   (trunc)             */
static inline uint8_t requant1(int32_t c, float sa, float sb, float sc,
                               uint8_t zp_c) {
  float dequant = ((float)c * sa) * sb;
  float quant = dequant / sc + (float)zp_c;
  return (quant >= 255.0f) ? (uint8_t)255
                           : ((quant < 0.0f) ? (uint8_t)0 : (uint8_t)quant);
}
void orc_down_scale(uint8_t* out, const int32_t* acc, int64_t n, float sa,
                    float sb, float sc, uint8_t zp_c) {
  for (int64_t i = 0; i < n; ++i) out[i] = requant1(acc[i], sa, sb, sc, zp_c);
}

/* ------------------------------------------------------------------ a7 --
 * src/functional.cc:15-26  relu<u8_t>:  out = in > zp ? in : zp              */
void orc_relu_u8(const uint8_t* in, uint8_t* out, int64_t n, uint8_t zp) {
  for (int64_t i = 0; i < n; ++i) out[i] = in[i] > zp ? in[i] : zp;
}

/* ------------------------------------------------------------------ a8 --
 * src/functional.cc:36-64  max_pool2d<u8_t>: running max from 0 over a k x k
 * window, stride s, oh = (h - k) / s + 1, no padding.                        */
void orc_max_pool2d_u8(const uint8_t* in, uint8_t* out, int n, int c, int h,
                       int w, int k, int s) {
  int oh = (h - k) / s + 1, ow = (w - k) / s + 1;
#pragma omp parallel for collapse(2)
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < c; ++j) {
      const uint8_t* p = in + ((int64_t)i * c + j) * h * w;
      uint8_t* o = out + ((int64_t)i * c + j) * oh * ow;
      for (int y = 0; y < oh; ++y)
        for (int x = 0; x < ow; ++x) {
          uint8_t mx = 0;
          for (int m = 0; m < k; ++m)
            for (int l = 0; l < k; ++l) {
              uint8_t v = p[(y * s + m) * w + (x * s + l)];
              mx = mx >= v ? mx : v;
            }
          o[y * ow + x] = mx;
        }
    }
}

/* ----------------------------------------------------------------- a10 --
 * src/layer.cc:6-26  quantize_weight: joint min/max over weight AND bias,
 *   s_w = (max - min) / 127;  q = (s8)(w / s_w)  (trunc, no clamp: wraps like
 *   a C cast through int on x86).  Returns s_w.                              */
float orc_quantize_weight(const float* w, int64_t nw, const float* b,
                          int64_t nb, int8_t* qw, int8_t* qb) {
  float mx = -3.402823466e+38f, mn = 3.402823466e+38f;
  for (int64_t i = 0; i < nw; ++i) {
    mn = w[i] < mn ? w[i] : mn;
    mx = w[i] > mx ? w[i] : mx;
  }
  for (int64_t i = 0; i < nb; ++i) {
    mn = b[i] < mn ? b[i] : mn;
    mx = b[i] > mx ? b[i] : mx;
  }
  float s = (mx - mn) / 127;
  for (int64_t i = 0; i < nw; ++i) qw[i] = (int8_t)(int32_t)(w[i] / s);
  for (int64_t i = 0; i < nb; ++i) qb[i] = (int8_t)(int32_t)(b[i] / s);
  return s;
}

/* ------------------------------------------------------------------ a4 --
 * src/conv2d.cc:117-124   per output channel j:
 *   float t = 0; for k: t += zp_in * q_w[j,k];     (int product -> float add)
 *   oc[j] = q_b[j] / s_in - t;                     (float -> int trunc)      */
void orc_conv_offsets(const int8_t* qw, const int8_t* qb, int kc, int K,
                      float s_in, uint8_t zp_in, int32_t* oc) {
  for (int j = 0; j < kc; ++j) {
    float t = 0;
    for (int k = 0; k < K; ++k)
      t = t + (float)((int)zp_in * (int)qw[(int64_t)j * K + k]);
    oc[j] = (int32_t)((float)qb[j] / s_in - t);
  }
}
/* src/fully_connected.cc:30-38   oc[i] = -t  (bias is added later, in float) */
void orc_linear_offsets(const int8_t* qw, int n, int k, uint8_t zp_in,
                        int32_t* oc) {
  for (int i = 0; i < n; ++i) {
    float t = 0;
    for (int j = 0; j < k; ++j)
      t = t + (float)((int)zp_in * (int)qw[(int64_t)i * k + j]);
    oc[i] = (int32_t)(-t);
  }
}

/* ----------------------------------------------------------------- a2' --
 * src/conv2d.cc:5-49  im2col (u8): row r = ti*ow + tj, col = ch*kh*kw + l*kw + m
 *   value = I[ch][ti*s - p + l][tj*s - p + m], or zero_point when outside.   */
void orc_im2col_u8(uint8_t* M, const uint8_t* I, int c, int h, int w, int kh,
                   int kw, int stride, int pad, uint8_t zp) {
  int oh = (h - kh + 2 * pad) / stride + 1;
  int ow = (w - kw + 2 * pad) / stride + 1;
  int K = c * kh * kw;
  for (int ti = 0; ti < oh; ++ti)
    for (int tj = 0; tj < ow; ++tj) {
      uint8_t* row = M + (int64_t)(ti * ow + tj) * K;
      int i = ti * stride - pad, j = tj * stride - pad;
      for (int ch = 0; ch < c; ++ch)
        for (int l = 0; l < kh; ++l)
          for (int m = 0; m < kw; ++m) {
            int y = i + l, x = j + m;
            row[ch * kh * kw + l * kw + m] =
                (y < 0 || x < 0 || y >= h || x >= w)
                    ? zp
                    : I[(int64_t)ch * h * w + y * w + x];
          }
    }
}

/* -------------------------------------------------------------------------
 * The contraction the reference delegates to Intel MKL (mkl_rt, not vendored;
 * reference pins 2019.5.281 by path, CMakeLists.txt:25):
 *   cblas_gemm_s8u8s32(RowMajor, NoTrans, Trans, RowOffset, M, N, K, alpha=1,
 *                      A(u8) lda=K, ao=0, B(s8) ldb=K, bo=0, beta=0, C, ldc=N, oc)
 * call sites src/conv2d.cc:131-133, src/fully_connected.cc:39-41.
 * Published semantics: C[i][j] = sum_k (int)A[i][k] * (int)B[j][k] + oc[j],
 * exact in int32.  target_clones only changes the ISA the loop is vectorised
 * for; integer sums are exact either way.                                    */
#if defined(__x86_64__) && defined(__GNUC__) && !defined(ORC_NO_CLONES)
__attribute__((target_clones("arch=icelake-server", "arch=skylake-avx512", "avx2", "default")))
#endif
static void gemm_rows(int M, int N, int K, const uint8_t* A, const int8_t* B,
                      const int32_t* oc, int32_t* C) {
  for (int i = 0; i < M; ++i) {
    const uint8_t* a = A + (int64_t)i * K;
    for (int j = 0; j < N; ++j) {
      const int8_t* b = B + (int64_t)j * K;
      int32_t s = 0;
      for (int k = 0; k < K; ++k) s += (int32_t)a[k] * (int32_t)b[k];
      C[(int64_t)i * N + j] = s + oc[j];
    }
  }
}
void orc_gemm_u8s8s32(int M, int N, int K, const uint8_t* A, const int8_t* B,
                      const int32_t* oc, int32_t* C) {
  gemm_rows(M, N, K, A, B, oc, C);
}

/* Optional GEMM provider for the TIMED CPU BASELINE: the address of the reference's own provider entry point,
 * Intel MKL's cblas_gemm_s8u8s32, when its runtime is installed on the host (oracle/orc.py loads it with
 * ctypes; nothing of the reference is built).  With a provider set, orc_conv2d_u8 / orc_linear_u8 call it
 * exactly where and how the reference does (src/conv2d.cc:131-133 inside the OpenMP loop over images,
 * src/fully_connected.cc:39-41 as one call that MKL threads itself); results must equal gemm_rows (the
 * caller verifies that on the host before trusting a time).                                              */
typedef void (*orc_gemm_provider_fn)(int layout, int transa, int transb, int offsetc, int m, int n, int k,
                                     float alpha, const void* a, int lda, int8_t ao, const void* b, int ldb,
                                     int8_t bo, float beta, int32_t* c, int ldc, const int32_t* co);
static orc_gemm_provider_fn g_gemm_provider = 0;
void orc_set_gemm_provider(void* fn) { g_gemm_provider = (orc_gemm_provider_fn)fn; }
static void provider_gemm(int M, int N, int K, const uint8_t* A, const int8_t* B, const int32_t* oc, int32_t* C) {
  /* CblasRowMajor, CblasNoTrans, CblasTrans, CblasRowOffset */
  g_gemm_provider(101, 111, 112, 171, M, N, K, 1.0f, A, K, 0, B, K, 0, 0.0f, C, N, oc);
}
void orc_gemm_u8s8s32_provider(int M, int N, int K, const uint8_t* A, const int8_t* B,
                               const int32_t* oc, int32_t* C) {
  if (g_gemm_provider) provider_gemm(M, N, K, A, B, oc, C);
  else gemm_rows(M, N, K, A, B, oc, C);
}

/* ------------------------------------------------------------------ a2 --
 * src/conv2d.cc:100-142  Conv2d::forward_prop(Tensor<u8>&&)
 *   oc (a4) once; then per image (OpenMP over the batch, :125):
 *   im2col(pad = zp_in) -> C = A*W^T + oc -> down_scale -> [M,N] -> [N,M]
 * in: u8 NCHW [n,c,h,w]; qw: s8 [kc, c*kh*kw]; out: u8 NCHW [n,kc,oh,ow].
 * acc (optional): int32 [n, oh*ow, kc], the pre-requant accumulators.        */
void orc_conv2d_u8(const uint8_t* in, int n, int c, int h, int w,
                   const int8_t* qw, const int8_t* qb, int kc, int kh, int kw,
                   int stride, int pad, float s_in, uint8_t zp_in, float s_w,
                   float s_out, uint8_t zp_out, uint8_t* out, int32_t* acc) {
  int oh = (h - kh + 2 * pad) / stride + 1;
  int ow = (w - kw + 2 * pad) / stride + 1;
  int M = oh * ow, N = kc, K = c * kh * kw;
  int32_t* oc = (int32_t*)malloc(sizeof(int32_t) * (size_t)N);
  orc_conv_offsets(qw, qb, kc, K, s_in, zp_in, oc);
#pragma omp parallel for
  for (int i = 0; i < n; ++i) {
    uint8_t* mat = (uint8_t*)malloc((size_t)M * K);
    int32_t* C = (int32_t*)malloc(sizeof(int32_t) * (size_t)M * N);
    uint8_t* hwc = (uint8_t*)malloc((size_t)M * N);
    orc_im2col_u8(mat, in + (int64_t)i * c * h * w, c, h, w, kh, kw, stride,
                  pad, zp_in);
    if (g_gemm_provider) provider_gemm(M, N, K, mat, qw, oc, C);
    else gemm_rows(M, N, K, mat, qw, oc, C);
    if (acc) memcpy(acc + (int64_t)i * M * N, C, sizeof(int32_t) * (size_t)M * N);
    orc_down_scale(hwc, C, (int64_t)M * N, s_in, s_w, s_out, zp_out);
    uint8_t* o = out + (int64_t)i * N * M; /* transpose, src/conv2d.cc:51-61 */
    for (int r = 0; r < M; ++r)
      for (int j = 0; j < N; ++j) o[(int64_t)j * M + r] = hwc[(int64_t)r * N + j];
    free(hwc);
    free(C);
    free(mat);
  }
  free(oc);
}

/* ------------------------------------------------------------------ a3 --
 * src/fully_connected.cc:22-52  Linear::forward_prop(Tensor<u8>&&)
 *   oc[i] = -t_i;  C = X*W^T + oc;  C[i][j] += q_b[j] / s_in  (int += float:
 *   C = (int)((float)C + (float)q_b[j]/s_in));  down_scale.
 * acc_pre (optional): C before the bias step; acc_post (optional): after.    */
void orc_linear_u8(const uint8_t* in, int m, int k, const int8_t* qw,
                   const int8_t* qb, int n, float s_in, uint8_t zp_in,
                   float s_w, float s_out, uint8_t zp_out, uint8_t* out,
                   int32_t* acc_pre, int32_t* acc_post) {
  int32_t* C = (int32_t*)malloc(sizeof(int32_t) * (size_t)m * n);
  int32_t* oc = (int32_t*)malloc(sizeof(int32_t) * (size_t)n);
  orc_linear_offsets(qw, n, k, zp_in, oc);
  if (g_gemm_provider) provider_gemm(m, n, k, in, qw, oc, C);
  else
#pragma omp parallel
  {
#ifdef _OPENMP
    int nt = omp_get_num_threads(), id = omp_get_thread_num();
#else
    int nt = 1, id = 0;
#endif
    int r0 = (int)((int64_t)m * id / nt), r1 = (int)((int64_t)m * (id + 1) / nt);
    if (r1 > r0)
      gemm_rows(r1 - r0, n, k, in + (int64_t)r0 * k, qw, oc, C + (int64_t)r0 * n);
  }
  if (acc_pre) memcpy(acc_pre, C, sizeof(int32_t) * (size_t)m * n);
  for (int i = 0; i < m; ++i)
    for (int j = 0; j < n; ++j) {
      int64_t e = (int64_t)i * n + j;
      C[e] = (int32_t)((float)C[e] + (float)qb[j] / s_in);
    }
  if (acc_post) memcpy(acc_post, C, sizeof(int32_t) * (size_t)m * n);
  orc_down_scale(out, C, (int64_t)m * n, s_in, s_w, s_out, zp_out);
  free(oc);
  free(C);
}

/* ----------------------------------------------------------------- A.6 --
 * src/calibrator.cc:24-37  Calibrator::get_range(quantile) on the 1000-slot
 * sample buffer (all 1000 slots are sorted, filled or not; unfilled slots are
 * zero because make_unique<Calibrator>() value-initialises, src/layer.cc:33).
 * The sampling itself (:6-23) is nondeterministic (std::random_device) and is
 * not restated here.                                                         */
static int cmp_float(const void* a, const void* b) {
  float x = *(const float*)a, y = *(const float*)b;
  return (x > y) - (x < y);
}
void orc_calib_range(float* samples /* [1000], sorted in place */, int64_t cnt,
                     float quantile, float* scale, uint8_t* zero_point) {
  qsort(samples, 1000, sizeof(float), cmp_float);
  float out_min = samples[(int64_t)((1.0 - quantile) * cnt)];
  float out_max = samples[(int64_t)(quantile * (cnt - 1))];
  out_min = (float)fmin(out_min, 0.);
  out_max = (float)fmax(out_max, 0.);
  uint8_t zp = (uint8_t)(int32_t)(255 * (0 - out_min) / (out_max - out_min + 1e-09));
  float s = (zp == 0) ? (out_max - out_min) / 255 : (0 - out_min) / zp;
  if (s == 0) s = 1;
  *scale = s;
  *zero_point = zp;
}

/* the host process may have initialised libgomp before OMP_NUM_THREADS could be set */
void orc_set_num_threads(int n) {
#ifdef _OPENMP
  if (n > 0) omp_set_num_threads(n);
#else
  (void)n;
#endif
}

/* number of OpenMP threads the baseline actually uses (bench.py reports it) */
int orc_num_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}
