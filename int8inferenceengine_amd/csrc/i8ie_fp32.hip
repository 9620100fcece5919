// i8ie_fp32.hip -- FP32 forward ops used before convert() and during
// calibration (prepare() -> one FP32 batch -> convert()).  They are off the
// timed INT8 path: plain, correct, LDS-tiled, not
// tuned.  The reference computes these with cblas_sgemm
// (src/conv2d.cc:63-98, src/fully_connected.cc:5-21) and pins them only to
// atol 0.1 against torch (unittest/test_layers.py:10-11), so summation order
// is free here.
#include "i8ie_internal.h"


namespace {

constexpr int TS = 16;

// out[i][j] = sum_k in[i][k] * w[j][k] + b[j]      (16x16 LDS tiles)
__global__ __launch_bounds__(TS* TS) void linear_f32_kernel(const float* __restrict__ in,
                                                            const float* __restrict__ w,
                                                            const float* __restrict__ b,
                                                            float* __restrict__ out, int m, int k, int n) {
  __shared__ float sa[TS][TS + 1], sb[TS][TS + 1];
  const int tx = threadIdx.x % TS, ty = threadIdx.x / TS;
  const int row = blockIdx.y * TS + ty, col = blockIdx.x * TS + tx;
  float acc = 0.0f;
  for (int k0 = 0; k0 < k; k0 += TS) {
    const int ar = blockIdx.y * TS + ty, ak = k0 + tx;
    sa[ty][tx] = (ar < m && ak < k) ? in[(size_t)ar * k + ak] : 0.0f;
    const int br = blockIdx.x * TS + ty;
    sb[ty][tx] = (br < n && ak < k) ? w[(size_t)br * k + ak] : 0.0f;
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < TS; ++kk) acc += sa[ty][kk] * sb[tx][kk];
    __syncthreads();
  }
  if (row < m && col < n) out[(size_t)row * n + col] = acc + b[col];
}

// direct convolution, one thread per output element, NCHW, zero padding
__global__ __launch_bounds__(256) void conv2d_f32_kernel(const float* __restrict__ in,
                                                         const float* __restrict__ wt,
                                                         const float* __restrict__ b, float* __restrict__ out,
                                                         int64_t total, int c, int h, int w, int kc, int kh,
                                                         int kw, int oh, int ow, int stride, int pad) {
  const int64_t gstride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += gstride) {
    const int x = (int)(e % ow);
    int64_t t = e / ow;
    const int y = (int)(t % oh);
    t /= oh;
    const int j = (int)(t % kc);
    const int64_t img = t / kc;
    const int y0 = y * stride - pad, x0 = x * stride - pad;
    const float* ip = in + img * c * h * w;
    const float* wp = wt + (size_t)j * c * kh * kw;
    float acc = 0.0f;
    for (int ch = 0; ch < c; ++ch)
      for (int l = 0; l < kh; ++l) {
        const int yy = y0 + l;
        if (yy < 0 || yy >= h) continue;
        for (int mm = 0; mm < kw; ++mm) {
          const int xx = x0 + mm;
          if (xx < 0 || xx >= w) continue;
          acc += ip[((size_t)ch * h + yy) * w + xx] * wp[(ch * kh + l) * kw + mm];
        }
      }
    out[e] = acc + b[j];
  }
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
  dim3 grid((n + TS - 1) / TS, (m + TS - 1) / TS);
  linear_f32_kernel<<<grid, TS * TS, 0, ctx->stream>>>(in, w, b, out, m, k, n);
  I8IE_LAUNCH_CHECK();
  return I8IE_OK;
}

int i8ie_conv2d_f32(i8ie_ctx* ctx, const float* in, int n, int c, int h, int w, const float* wt,
                    const float* b, int kc, int kh, int kw, int stride, int pad, float* out) {
  I8IE_REQUIRE(ctx && in && wt && b && out, "null argument");
  I8IE_REQUIRE(n > 0 && c > 0 && h > 0 && w > 0 && kc > 0 && kh > 0 && kw > 0, "non-positive dimension");
  I8IE_REQUIRE(stride > 0 && pad >= 0, "bad stride/padding");
  I8IE_REQUIRE(h - kh + 2 * pad >= 0 && w - kw + 2 * pad >= 0, "kernel larger than padded input");
  const int oh = (h - kh + 2 * pad) / stride + 1, ow = (w - kw + 2 * pad) / stride + 1;
  const int64_t total = (int64_t)n * kc * oh * ow;
  I8IE_HIP_TRY(hipSetDevice(ctx->device));
  conv2d_f32_kernel<<<cap_grid(total), 256, 0, ctx->stream>>>(in, wt, b, out, total, c, h, w, kc, kh, kw, oh,
                                                              ow, stride, pad);
  I8IE_LAUNCH_CHECK();
  return I8IE_OK;
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
