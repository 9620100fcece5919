// i8ie_fp32.hip -- FP32 forward ops used before convert() and during
// calibration (prepare() -> one FP32 batch -> convert()).  They are off the
// timed INT8 path.  Conv2d and Linear run on the f32
// matrix instruction (round 4); relu / max-pool are plain streaming kernels.
// The reference computes these with cblas_sgemm
// (src/conv2d.cc:63-98, src/fully_connected.cc:5-21) and pins them only to
// atol 0.1 against torch (unittest/test_layers.py:10-11), so summation order
// is free here.
#include "i8ie_internal.h"


namespace {

// ---- one implicit-GEMM kernel for Conv2d and Linear in FP32 on the f32 matrix instruction -------------------------------
//   out[r][j] = sum_k X[r][k] * W[j][k] + b[j],   X = im2col(in) (zero padding; src/conv2d.cc:63-98) or in itself (Linear,
//   src/fully_connected.cc:5-21)
// v_mfma_f32_32x32x2_f32 is an exact fp32 fma chain (64 FLOP per clock and SIMD = the vector rate, 157 TFLOP/s on the chip) and
// leaves the vector unit to the gathers.  Round 3's kernels were one-thread-per-output VALU loops: 43 ms per launch on AlexNet's
// convolutions, 1.7 k images/s through the FP32 network (what prepare() / calibration / the FP32 teacher run).
//   * block = 256 threads, tile = 128 rows (output pixels / input rows) x 128 features, K walked 16 at a time through ONE LDS
//     stage per operand, k-major ([16][132] floats: a fragment read is 64 consecutive floats, conflict-free); the next K block's
//     16 + 16 values per thread are in registers before this block's MFMAs start.
//   * the im2col gather: a row's (image, oy, ox) is worked out once per thread (its row never changes); a K position's
//     (input offset, kernel row, kernel column) comes from a table of K entries built per launch (k_table_kernel) -- no
//     division in the loop.
//   * each wave owns 64 x 64 outputs (2 x 2 MFMA tiles, 64 accumulator registers).  Which operand is srcA decides what a lane
//     holds: for Conv2d a lane = an output pixel (NCHW stores: 32 consecutive pixels of one feature per instruction), for Linear
//     a lane = a feature (row-major stores).
constexpr int FB = 128, FK = 16, FP = 132;  // tile edge, K block, LDS row pitch (floats)
typedef float v16f __attribute__((ext_vector_type(16)));

struct F32Args {
  const float* in;
  const float* w;
  const float* b;
  float* out;
  const int2* ktab;  // [K]: {input offset of K position k relative to the window origin, (kernel row << 16) | kernel column}
  int M, N, K;
  int c, h, wd, oh, ow, stride, pad, P;  // P = oh * ow output pixels per image (1 for Linear)
};

__global__ __launch_bounds__(256) void k_table_kernel(int2* tab, int K, int h, int w, int kh, int kw) {
  for (int k = blockIdx.x * 256 + threadIdx.x; k < K; k += gridDim.x * 256) {
    const int ch = k / (kh * kw), rem = k - ch * (kh * kw), l = rem / kw, m = rem - l * kw;
    tab[k] = make_int2((ch * h + l) * w + m, (l << 16) | m);
  }
}

template <bool CONV>
__global__ __launch_bounds__(256, 2) void gemm_f32_kernel(F32Args p) {
  __shared__ float Xs[FK][FP], Ws[FK][FP];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int bm0 = blockIdx.y * FB, bn0 = blockIdx.x * FB;

  // ---- this thread's gather row (activations): row bm0 + tid % 128, K positions (tid / 128) + 2 e
  const int xr = bm0 + (tid & 127), xk0 = tid >> 7;
  const bool xrow_ok = xr < p.M;
  const int ximg = xr / p.P, xpix = xr - ximg * p.P;
  const int xoy = xpix / p.ow, xox = xpix - xoy * p.ow;
  const int y0 = xoy * p.stride - p.pad, x0 = xox * p.stride - p.pad;
  const float* xbase = p.in + ((size_t)ximg * p.c * p.h + y0) * (size_t)p.wd + x0;  // (may point before the image: only used with valid taps)
  // ---- this thread's weight elements: K position tid % 16, features tid / 16 + 16 e
  const int wk = tid & 15, wj0 = tid >> 4;

  float xv[8], wv[8];
  auto fetch = [&](int k0) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int k = k0 + xk0 + 2 * e;
      float v = 0.0f;
      if (xrow_ok && k < p.K) {
        const int2 t = p.ktab[k];
        const int yy = y0 + (t.y >> 16), xx = x0 + (t.y & 0xffff);
        if (!CONV || ((unsigned)yy < (unsigned)p.h && (unsigned)xx < (unsigned)p.wd)) v = xbase[t.x];
      }
      xv[e] = v;
      const int j = bn0 + wj0 + 16 * e, kw_ = k0 + wk;
      wv[e] = (j < p.N && kw_ < p.K) ? p.w[(size_t)j * p.K + kw_] : 0.0f;
    }
  };
  auto stage = [&]() {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      Xs[xk0 + 2 * e][tid & 127] = xv[e];
      Ws[wk][wj0 + 16 * e] = wv[e];
    }
  };

  v16f acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;
  // srcA's rows become the accumulator's registers, srcB's columns its lanes: Conv2d multiplies (features x K) by (K x pixels),
  // Linear (rows x K) by (K x features)
  const int wa = wave >> 1, wb = wave & 1;  // this wave's 64-wide slices of the srcA / srcB tiles
  const float(*As)[FP] = CONV ? Ws : Xs;
  const float(*Bs)[FP] = CONV ? Xs : Ws;
  const int hh = lane >> 5, l31 = lane & 31;

  fetch(0);
  for (int k0 = 0; k0 < p.K; k0 += FK) {
    __syncthreads();  // everyone is done with the previous K block
    stage();
    __syncthreads();
    if (k0 + FK < p.K) fetch(k0 + FK);
#pragma unroll
    for (int kk = 0; kk < FK; kk += 2) {
      float a[2], b[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        a[i] = As[kk + hh][64 * wa + 32 * i + l31];
        b[i] = Bs[kk + hh][64 * wb + 32 * i + l31];
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
    }
  }

  // ---- + bias, store.  Register 4 g + r of tile (i, j) = srcA row 64 wa + 32 i + 8 g + 4 hh + r, srcB column 64 wb + 32 j + l31
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int col = 64 * wb + 32 * j + l31;
    if (CONV) {
      const int r = bm0 + col;  // output pixel (global row)
      if (r >= p.M) continue;
      const int img = r / p.P, pix = r - img * p.P;
      float* obase = p.out + (size_t)img * p.N * p.P + pix;
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int f = bn0 + 64 * wa + 32 * i + 8 * (e >> 2) + 4 * hh + (e & 3);
          if (f < p.N) obase[(size_t)f * p.P] = acc[i][j][e] + p.b[f];
        }
    } else {
      const int f = bn0 + col;
      if (f >= p.N) continue;
      const float bias = p.b[f];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int r = bm0 + 64 * wa + 32 * i + 8 * (e >> 2) + 4 * hh + (e & 3);
          if (r < p.M) p.out[(size_t)r * p.N + f] = acc[i][j][e] + bias;
        }
    }
  }
}

template <bool CONV>
int launch_gemm_f32(i8ie_ctx* ctx, F32Args a, int kh, int kw, const char* name) {
  I8IE_REQUIRE((size_t)a.M * a.K < ((size_t)1 << 62) && a.K < (1 << 30), "f32 gemm: dimensions");
  I8IE_TRY(i8ie_ws_reserve(ctx, (size_t)a.K * sizeof(int2) + 256));
  int2* tab = reinterpret_cast<int2*>(ctx->ws);
  k_table_kernel<<<(a.K + 255) / 256 < 64 ? (a.K + 255) / 256 : 64, 256, 0, ctx->stream>>>(tab, a.K, a.h, a.wd, kh, kw);
  I8IE_LAUNCH_CHECK();
  a.ktab = tab;
  const dim3 grid((unsigned)((a.N + FB - 1) / FB), (unsigned)((a.M + FB - 1) / FB));
  // (blockIdx.y carries the row tiles: 65 535 x 128 = 8.4 M rows per launch; AlexNet's largest, 1000 x 55 x 55, is 3.0 M)
  I8IE_REQUIRE(grid.y <= 65535u, "f32 gemm: more than 8.4 M output rows in one call");
  I8ieProfScope prof(ctx, name, 0.0, (double)a.M * a.K * 4 + (double)a.N * a.K * 4 + (double)a.M * a.N * 4);
  gemm_f32_kernel<CONV><<<grid, 256, 0, ctx->stream>>>(a);
  I8IE_LAUNCH_CHECK();
  return I8IE_OK;
}

__global__ __launch_bounds__(256) void relu_f32_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                       int64_t n) {
  const int64_t gstride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gstride)
    out[i] = in[i] > 0.0f ? in[i] : 0.0f;  // src/functional.cc:5-13
}

// src/functional.cc:36-64 with T = float: running max starts at -FLT_MAX
__global__ __launch_bounds__(256) void maxpool_f32_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                          int64_t total, int h, int w, int oh, int ow, int k,
                                                          int s) {
  const int64_t gstride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += gstride) {
    const int x = (int)(e % ow);
    int64_t t = e / ow;
    const int y = (int)(t % oh);
    const int64_t plane = t / oh;
    const float* p = in + plane * h * w + (int64_t)(y * s) * w + x * s;
    float mx = -3.402823466e+38f;
    for (int m = 0; m < k; ++m)
      for (int l = 0; l < k; ++l) {
        const float v = p[m * w + l];
        mx = mx >= v ? mx : v;
      }
    out[e] = mx;
  }
}

inline int cap_grid(int64_t items) {
  int64_t b = (items + 255) / 256;
  if (b < 1) b = 1;
  return (int)(b > 256 * 32 ? 256 * 32 : b);
}

}  // namespace

extern "C" {

int i8ie_linear_f32(i8ie_ctx* ctx, const float* in, int m, int k, const float* w, const float* b, int n,
                    float* out) {
  I8IE_REQUIRE(ctx && in && w && b && out, "null argument");
  I8IE_REQUIRE(m > 0 && k > 0 && n > 0, "non-positive dimension");
  I8IE_HIP_TRY(hipSetDevice(ctx->device));
  F32Args a{};
  a.in = in; a.w = w; a.b = b; a.out = out;
  a.M = m; a.N = n; a.K = k;
  a.c = k; a.h = 1; a.wd = 1; a.oh = 1; a.ow = 1; a.stride = 1; a.pad = 0; a.P = 1;
  return launch_gemm_f32<false>(ctx, a, 1, 1, "linear_f32_mfma");
}

int i8ie_conv2d_f32(i8ie_ctx* ctx, const float* in, int n, int c, int h, int w, const float* wt,
                    const float* b, int kc, int kh, int kw, int stride, int pad, float* out) {
  I8IE_REQUIRE(ctx && in && wt && b && out, "null argument");
  I8IE_REQUIRE(n > 0 && c > 0 && h > 0 && w > 0 && kc > 0 && kh > 0 && kw > 0, "non-positive dimension");
  I8IE_REQUIRE(stride > 0 && pad >= 0, "bad stride/padding");
  I8IE_REQUIRE(h - kh + 2 * pad >= 0 && w - kw + 2 * pad >= 0, "kernel larger than padded input");
  const int oh = (h - kh + 2 * pad) / stride + 1, ow = (w - kw + 2 * pad) / stride + 1;
  I8IE_REQUIRE((int64_t)n * oh * ow < ((int64_t)1 << 31) && kh < 65536 && kw < 65536, "conv2d_f32: too many output pixels");
  I8IE_HIP_TRY(hipSetDevice(ctx->device));
  F32Args a{};
  a.in = in; a.w = wt; a.b = b; a.out = out;
  a.M = n * oh * ow; a.N = kc; a.K = c * kh * kw;
  a.c = c; a.h = h; a.wd = w; a.oh = oh; a.ow = ow; a.stride = stride; a.pad = pad; a.P = oh * ow;
  return launch_gemm_f32<true>(ctx, a, kh, kw, "conv2d_f32_mfma");
}

int i8ie_relu_f32(i8ie_ctx* ctx, const float* in, float* out, int64_t n) {
  I8IE_REQUIRE(ctx && in && out, "null argument");
  if (n <= 0) return I8IE_OK;
  I8IE_HIP_TRY(hipSetDevice(ctx->device));
  relu_f32_kernel<<<cap_grid(n), 256, 0, ctx->stream>>>(in, out, n);
  I8IE_LAUNCH_CHECK();
  return I8IE_OK;
}

int i8ie_maxpool2d_f32(i8ie_ctx* ctx, const float* in, float* out, int n, int c, int h, int w, int k, int s) {
  I8IE_REQUIRE(ctx && in && out, "null argument");
  I8IE_REQUIRE(n > 0 && c > 0 && h > 0 && w > 0 && k > 0 && s > 0, "non-positive dimension");
  I8IE_REQUIRE(k <= h && k <= w, "window larger than the input");
  const int oh = (h - k) / s + 1, ow = (w - k) / s + 1;
  const int64_t total = (int64_t)n * c * oh * ow;
  I8IE_HIP_TRY(hipSetDevice(ctx->device));
  maxpool_f32_kernel<<<cap_grid(total), 256, 0, ctx->stream>>>(in, out, total, h, w, oh, ow, k, s);
  I8IE_LAUNCH_CHECK();
  return I8IE_OK;
}

}  // extern "C"
