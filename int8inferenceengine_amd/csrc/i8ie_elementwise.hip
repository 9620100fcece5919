// i8ie_elementwise.hip -- HBM-bound byte/float kernels of the hot path:
// quantize (a1), dequantize (a6), down_scale (a5), relu u8 (a7), max-pool u8 (a8).
// All are pure streaming kernels: 16 B per lane per access, grid capped at
// 256 CUs x 8 blocks and grid-strided (guide: Guideline 11/13).
// fp32 arithmetic follows SURVEY.md Appendix A exactly (IEEE divide, no FMA
// contraction: the library is built with -ffp-contract=off).
#include "i8ie_internal.h"

namespace {

constexpr int kThreads = 256;
constexpr int kMaxBlocks = 256 * 8;

inline int grid_for(int64_t work_items) {
  int64_t b = (work_items + kThreads - 1) / kThreads;
  if (b < 1) b = 1;
  return (int)(b > kMaxBlocks ? kMaxBlocks : b);
}

// ---- a1: src/quantize_utils.cc:44-52 ---------------------------------------
__device__ __forceinline__ uint32_t quant1(float x, float scale, float zpf) {
  float t = x / scale + zpf;          // divide, then add (no FMA possible)
  return (uint32_t)((int)t) & 0xFFu;  // truncate, keep the low 8 bits (unclamped cast)
}

__global__ __launch_bounds__(kThreads) void quantize_kernel(const float* __restrict__ in,
                                                            uint8_t* __restrict__ out, int64_t n,
                                                            float scale, float zpf) {
  const int64_t nvec = n >> 4;  // 16 elements per lane: 4 x float4 in, 1 x uint4 out
  const int64_t stride = (int64_t)gridDim.x * kThreads;
  for (int64_t v = (int64_t)blockIdx.x * kThreads + threadIdx.x; v < nvec; v += stride) {
    const float4* src = reinterpret_cast<const float4*>(in) + v * 4;
    uint32_t w[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float4 f = src[j];
      w[j] = quant1(f.x, scale, zpf) | (quant1(f.y, scale, zpf) << 8) |
             (quant1(f.z, scale, zpf) << 16) | (quant1(f.w, scale, zpf) << 24);
    }
    reinterpret_cast<uint4*>(out)[v] = make_uint4(w[0], w[1], w[2], w[3]);
  }
  // tail (< 16 elements)
  const int64_t t0 = nvec << 4;
  if (blockIdx.x == 0 && threadIdx.x < (n - t0))
    out[t0 + threadIdx.x] = (uint8_t)quant1(in[t0 + threadIdx.x], scale, zpf);
}

// ---- a6: src/quantize_utils.cc:38-42 ---------------------------------------
__global__ __launch_bounds__(kThreads) void dequantize_kernel(const uint8_t* __restrict__ in,
                                                              float* __restrict__ out, int64_t n,
                                                              float scale, int zp) {
  const int64_t stride = (int64_t)gridDim.x * kThreads;
  for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < n; i += stride)
    out[i] = (float)((int)in[i] - zp) * scale;
}

// ---- a5: src/quantize_utils.cc:27-36 ---------------------------------------
__global__ __launch_bounds__(kThreads) void down_scale_kernel(const int32_t* __restrict__ acc,
                                                              uint8_t* __restrict__ out, int64_t n,
                                                              float sa, float sb, float sc,
                                                              float zpf) {
  const int64_t stride = (int64_t)gridDim.x * kThreads;
  for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < n; i += stride) {
    float deq = ((float)acc[i] * sa) * sb;
    float q = deq / sc + zpf;
    out[i] = (q >= 255.0f) ? (uint8_t)255 : ((q < 0.0f) ? (uint8_t)0 : (uint8_t)(int)q);
  }
}

// ---- a7: src/functional.cc:15-26 -------------------------------------------
__device__ __forceinline__ uint32_t max_u8x4(uint32_t a, uint32_t b) {
  uint32_t r = 0;
#pragma unroll
  for (int s = 0; s < 32; s += 8) {
    uint32_t x = (a >> s) & 0xFFu, y = (b >> s) & 0xFFu;
    r |= (x > y ? x : y) << s;
  }
  return r;
}

__global__ __launch_bounds__(kThreads) void relu_u8_kernel(const uint8_t* in,  // may alias out (in-place)
                                                           uint8_t* out, int64_t n,
                                                           uint32_t zp4) {
  const int64_t nvec = n >> 4;
  const int64_t stride = (int64_t)gridDim.x * kThreads;
  for (int64_t v = (int64_t)blockIdx.x * kThreads + threadIdx.x; v < nvec; v += stride) {
    uint4 x = reinterpret_cast<const uint4*>(in)[v];
    x.x = max_u8x4(x.x, zp4);
    x.y = max_u8x4(x.y, zp4);
    x.z = max_u8x4(x.z, zp4);
    x.w = max_u8x4(x.w, zp4);
    reinterpret_cast<uint4*>(out)[v] = x;
  }
  const int64_t t0 = nvec << 4;
  if (blockIdx.x == 0 && threadIdx.x < (n - t0)) {
    uint8_t v = in[t0 + threadIdx.x], z = (uint8_t)(zp4 & 0xFF);
    out[t0 + threadIdx.x] = v > z ? v : z;
  }
}

// x ^ 0x80 over a buffer: u8 <-> the re-biased form (value - 128 as s8) of I8IE_LAYOUT_NHWC_S8; may run in place
__global__ __launch_bounds__(kThreads) void rebias_u8_kernel(const uint8_t* in, uint8_t* out, int64_t n) {
  const int64_t nvec = n >> 4;
  const int64_t stride = (int64_t)gridDim.x * kThreads;
  for (int64_t v = (int64_t)blockIdx.x * kThreads + threadIdx.x; v < nvec; v += stride) {
    uint4 x = reinterpret_cast<const uint4*>(in)[v];
    x.x ^= 0x80808080u; x.y ^= 0x80808080u; x.z ^= 0x80808080u; x.w ^= 0x80808080u;
    reinterpret_cast<uint4*>(out)[v] = x;
  }
  const int64_t t0 = nvec << 4;
  if (blockIdx.x == 0 && threadIdx.x < (n - t0)) out[t0 + threadIdx.x] = in[t0 + threadIdx.x] ^ 0x80;
}

// ---- a8: src/functional.cc:36-64 (NCHW, floor, no padding, running max from 0)
__global__ __launch_bounds__(kThreads) void maxpool_u8_nchw_kernel(const uint8_t* __restrict__ in,
                                                                   uint8_t* __restrict__ out,
                                                                   int64_t total, int h, int w,
                                                                   int oh, int ow, int k, int s) {
  const int64_t stride = (int64_t)gridDim.x * kThreads;
  for (int64_t e = (int64_t)blockIdx.x * kThreads + threadIdx.x; e < total; e += stride) {
    int x = (int)(e % ow);
    int64_t t = e / ow;
    int y = (int)(t % oh);
    int64_t plane = t / oh;  // img * c + channel
    const uint8_t* p = in + plane * h * w + (int64_t)(y * s) * w + x * s;
    uint32_t mx = 0;
    for (int m = 0; m < k; ++m)
      for (int l = 0; l < k; ++l) {
        uint32_t v = p[m * w + l];
        mx = mx >= v ? mx : v;
      }
    out[e] = (uint8_t)mx;
  }
}

// ---- the s8 instantiations the reference also registers (src/functional.cc:78-82): the generic templates,
// relu = x > 0 ? x : 0 (src/functional.cc:5-13), max-pool with a running maximum that starts at
// -numeric_limits<char>::max() = -127 (src/functional.cc:28-31).  Nothing on the INT8 inference path produces
// an s8 tensor; they exist so that the drop-in module offers every overload the reference does.
__global__ __launch_bounds__(kThreads) void relu_s8_kernel(const int8_t* in, int8_t* out, int64_t n) {
  const int64_t stride = (int64_t)gridDim.x * kThreads;
  for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < n; i += stride) out[i] = in[i] > 0 ? in[i] : (int8_t)0;
}
__global__ __launch_bounds__(kThreads) void maxpool_s8_nchw_kernel(const int8_t* __restrict__ in, int8_t* __restrict__ out,
                                                                   int64_t total, int h, int w, int oh, int ow, int k,
                                                                   int s) {
  const int64_t stride = (int64_t)gridDim.x * kThreads;
  for (int64_t e = (int64_t)blockIdx.x * kThreads + threadIdx.x; e < total; e += stride) {
    const int x = (int)(e % ow);
    const int64_t t = e / ow;
    const int y = (int)(t % oh);
    const int64_t plane = t / oh;
    const int8_t* p = in + plane * h * w + (int64_t)(y * s) * w + x * s;
    int mx = -127;
    for (int m = 0; m < k; ++m)
      for (int l = 0; l < k; ++l) {
        const int v = p[m * w + l];
        mx = mx >= v ? mx : v;
      }
    out[e] = (int8_t)mx;
  }
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }


// ---- Calibrator::sample on the device (src/calibrator.cc:6-23; include/i8ie_hip.h) ---------------------------
constexpr int kCalibSlots = 1000;
__device__ __forceinline__ uint32_t calib_draw(uint64_t seed, uint64_t g) {  // uniform in [0, 2000], splitmix64 of (seed, g)
  uint64_t z = seed + 0x9E3779B97F4A7C15ull * (g + 1);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z ^= z >> 31;
  return (uint32_t)(((z >> 32) * 2001ull) >> 32);
}
// pass 1: the first 1000 values seen fill their slots; a later value that draws slot idx records its position
// (the LAST one wins, as in the reference's sequential loop).  Round 4: the last writer of a slot is almost surely among the
// final few hundred thousand values (a value draws a given slot with probability 1 / 2001), so the TAIL [n - kCalibTail, n) is
// marked first, and a block of the pass over the head leaves at once unless some slot is still empty (atomicMax keeps the larger
// index either way: same result as one pass over everything, which cost 8 of the
// 17 ms of a 100-image calibration forward -- 14 M atomics on 1000 words for conv1's output alone).
constexpr int64_t kCalibTail = 1 << 18;
__global__ __launch_bounds__(256) void calib_mark_kernel(const float* __restrict__ data, int64_t lo, int64_t hi, int64_t seen, uint64_t seed,
                                                         float* __restrict__ samples, int* __restrict__ last, int check_empty) {
  if (check_empty) {  // (head pass: nothing to do once every slot has its last writer)
    int e = 0;
    for (int sl = threadIdx.x; sl < kCalibSlots; sl += 256) e |= last[sl] < 0 ? 1 : 0;
    if (!__syncthreads_or(e)) return;
  }
  const int64_t gstride = (int64_t)gridDim.x * 256;
  for (int64_t i = lo + (int64_t)blockIdx.x * 256 + threadIdx.x; i < hi; i += gstride) {
    const int64_t g = seen + i;
    if (g < kCalibSlots) {
      samples[g] = data[i];
    } else {
      const uint32_t idx = calib_draw(seed, (uint64_t)g);
      if (idx < (uint32_t)kCalibSlots) atomicMax(&last[idx], (int)(i > 0x7FFFFFFF ? 0x7FFFFFFF : i));
    }
  }
}
__global__ __launch_bounds__(256) void calib_take_kernel(const float* __restrict__ data, float* __restrict__ samples,
                                                         int* __restrict__ last) {
  const int s = blockIdx.x * 256 + threadIdx.x;
  if (s < kCalibSlots) {
    const int i = last[s];
    if (i >= 0) samples[s] = data[i];
    last[s] = -1;
  }
}
}  // namespace

extern "C" {

int i8ie_quantize_f32_u8(i8ie_ctx* ctx, const float* in, uint8_t* out, int64_t n, float scale,
                         uint8_t zp) {
  I8IE_REQUIRE(ctx && in && out, "null argument");
  I8IE_REQUIRE(n >= 0, "negative size");
  I8IE_REQUIRE(aligned16(in) && aligned16(out), "buffers must be 16-byte aligned");
  if (n == 0) return I8IE_OK;
  I8IE_HIP_TRY(hipSetDevice(ctx->device));
  I8ieProfScope prof(ctx, "quantize_f32_u8", 0.0, 5.0 * n);
  quantize_kernel<<<grid_for((n >> 4) + 1), kThreads, 0, ctx->stream>>>(in, out, n, scale, (float)zp);
  I8IE_LAUNCH_CHECK();
  return I8IE_OK;
}

int i8ie_dequantize_u8_f32(i8ie_ctx* ctx, const uint8_t* in, float* out, int64_t n, float scale,
                           uint8_t zp) {
  I8IE_REQUIRE(ctx && in && out, "null argument");
  I8IE_REQUIRE(n >= 0, "negative size");
  if (n == 0) return I8IE_OK;
  I8IE_HIP_TRY(hipSetDevice(ctx->device));
  I8ieProfScope prof(ctx, "dequantize_u8_f32", 0.0, 5.0 * n);
  dequantize_kernel<<<grid_for(n), kThreads, 0, ctx->stream>>>(in, out, n, scale, (int)zp);
  I8IE_LAUNCH_CHECK();
  return I8IE_OK;
}

int i8ie_down_scale(i8ie_ctx* ctx, const int32_t* acc, uint8_t* out, int64_t n, float sa, float sb,
                    float sc, uint8_t zp_c) {
  I8IE_REQUIRE(ctx && acc && out, "null argument");
  I8IE_REQUIRE(n >= 0, "negative size");
  if (n == 0) return I8IE_OK;
  I8IE_HIP_TRY(hipSetDevice(ctx->device));
  I8ieProfScope prof(ctx, "down_scale", 0.0, 5.0 * n);
  down_scale_kernel<<<grid_for(n), kThreads, 0, ctx->stream>>>(acc, out, n, sa, sb, sc, (float)zp_c);
  I8IE_LAUNCH_CHECK();
  return I8IE_OK;
}

int i8ie_relu_u8(i8ie_ctx* ctx, const uint8_t* in, uint8_t* out, int64_t n, uint8_t zp) {
  I8IE_REQUIRE(ctx && in && out, "null argument");
  I8IE_REQUIRE(n >= 0, "negative size");
  I8IE_REQUIRE(aligned16(in) && aligned16(out), "buffers must be 16-byte aligned");
  if (n == 0) return I8IE_OK;
  I8IE_HIP_TRY(hipSetDevice(ctx->device));
  uint32_t z = zp;
  I8ieProfScope prof(ctx, "relu_u8", 0.0, 2.0 * n);
  relu_u8_kernel<<<grid_for((n >> 4) + 1), kThreads, 0, ctx->stream>>>(in, out, n,
                                                                        z | (z << 8) | (z << 16) | (z << 24));
  I8IE_LAUNCH_CHECK();
  return I8IE_OK;
}

int i8ie_rebias_u8(i8ie_ctx* ctx, const uint8_t* in, uint8_t* out, int64_t n) {
  I8IE_REQUIRE(ctx && in && out, "null argument");
  I8IE_REQUIRE(n >= 0, "negative size");
  I8IE_REQUIRE(aligned16(in) && aligned16(out), "buffers must be 16-byte aligned");
  if (n == 0) return I8IE_OK;
  I8IE_HIP_TRY(hipSetDevice(ctx->device));
  I8ieProfScope prof(ctx, "rebias_u8", 0.0, 2.0 * n);
  rebias_u8_kernel<<<grid_for((n >> 4) + 1), kThreads, 0, ctx->stream>>>(in, out, n);
  I8IE_LAUNCH_CHECK();
  return I8IE_OK;
}

int i8ie_maxpool2d_u8(i8ie_ctx* ctx, const uint8_t* in, uint8_t* out, int n, int c, int h, int w,
                      int k, int s) {
  I8IE_REQUIRE(ctx && in && out, "null argument");
  I8IE_REQUIRE(n > 0 && c > 0 && h > 0 && w > 0, "non-positive dimension");
  I8IE_REQUIRE(k > 0 && s > 0, "kernel_size and stride must be positive");
  I8IE_REQUIRE(k <= h && k <= w, "window larger than the input");
  int oh = (h - k) / s + 1, ow = (w - k) / s + 1;
  int64_t total = (int64_t)n * c * oh * ow;
  I8IE_HIP_TRY(hipSetDevice(ctx->device));
  I8ieProfScope prof(ctx, "maxpool_u8_nchw", 0.0, (double)n * c * h * w + (double)total);
  maxpool_u8_nchw_kernel<<<grid_for(total), kThreads, 0, ctx->stream>>>(in, out, total, h, w, oh, ow,
                                                                       k, s);
  I8IE_LAUNCH_CHECK();
  return I8IE_OK;
}

int i8ie_calib_sample_f32(i8ie_ctx* ctx, const float* data, int64_t n, int64_t seen_before, uint64_t seed, float* samples,
                          int* scratch) {
  I8IE_REQUIRE(ctx && data && samples && scratch, "null argument");
  I8IE_REQUIRE(n >= 0 && seen_before >= 0, "negative size");
  I8IE_REQUIRE(n <= 0x7FFFFFFF, "one call samples at most 2^31 - 1 values");
  if (n == 0) return I8IE_OK;
  I8IE_HIP_TRY(hipSetDevice(ctx->device));
  if (seen_before == 0) I8IE_HIP_TRY(hipMemsetAsync(scratch, 0xFF, kCalibSlots * sizeof(int), ctx->stream));  // all -1
  I8ieProfScope prof(ctx, "calib_sample", 0.0, 4.0 * n);
  const int64_t tail_lo = n > kCalibTail ? n - kCalibTail : 0;
  calib_mark_kernel<<<grid_for(n - tail_lo), kThreads, 0, ctx->stream>>>(data, tail_lo, n, seen_before, seed, samples, scratch, 0);
  I8IE_LAUNCH_CHECK();
  if (tail_lo > 0) {
    // values that fill their own slot (the first kCalibSlots seen) lie in the head: that part always runs
    const int64_t fill_hi = seen_before < kCalibSlots ? (kCalibSlots - seen_before < tail_lo ? kCalibSlots - seen_before : tail_lo) : 0;
    if (fill_hi > 0) {
      calib_mark_kernel<<<grid_for(fill_hi), kThreads, 0, ctx->stream>>>(data, 0, fill_hi, seen_before, seed, samples, scratch, 0);
      I8IE_LAUNCH_CHECK();
    }
    calib_mark_kernel<<<grid_for(tail_lo - fill_hi), kThreads, 0, ctx->stream>>>(data, fill_hi, tail_lo, seen_before, seed, samples, scratch, 1);
    I8IE_LAUNCH_CHECK();
  }
  calib_take_kernel<<<(kCalibSlots + 255) / 256, 256, 0, ctx->stream>>>(data, samples, scratch);
  I8IE_LAUNCH_CHECK();
  return I8IE_OK;
}

int i8ie_relu_s8(i8ie_ctx* ctx, const int8_t* in, int8_t* out, int64_t n) {
  I8IE_REQUIRE(ctx && in && out, "null argument");
  I8IE_REQUIRE(n >= 0, "negative size");
  if (n == 0) return I8IE_OK;
  I8IE_HIP_TRY(hipSetDevice(ctx->device));
  I8ieProfScope prof(ctx, "relu_s8", 0.0, 2.0 * n);
  relu_s8_kernel<<<grid_for(n), kThreads, 0, ctx->stream>>>(in, out, n);
  I8IE_LAUNCH_CHECK();
  return I8IE_OK;
}

int i8ie_maxpool2d_s8(i8ie_ctx* ctx, const int8_t* in, int8_t* out, int n, int c, int h, int w, int k, int s) {
  I8IE_REQUIRE(ctx && in && out, "null argument");
  I8IE_REQUIRE(n > 0 && c > 0 && h > 0 && w > 0, "non-positive dimension");
  I8IE_REQUIRE(k > 0 && s > 0, "kernel_size and stride must be positive");
  I8IE_REQUIRE(k <= h && k <= w, "window larger than the input");
  const int oh = (h - k) / s + 1, ow = (w - k) / s + 1;
  const int64_t total = (int64_t)n * c * oh * ow;
  I8IE_HIP_TRY(hipSetDevice(ctx->device));
  I8ieProfScope prof(ctx, "maxpool_s8_nchw", 0.0, (double)n * c * h * w + (double)total);
  maxpool_s8_nchw_kernel<<<grid_for(total), kThreads, 0, ctx->stream>>>(in, out, total, h, w, oh, ow, k, s);
  I8IE_LAUNCH_CHECK();
  return I8IE_OK;
}

}  // extern "C"
