// i8ie_requant.h -- the one requantiser of libi8ie_hip.so (every kernel's epilogue includes this file).
//
// Reference: down_scale, src/quantize_utils.cc:27-36
//     deq = ((float)C * s_in) * s_w;  q = deq / s_out + (float)zp_out;
//     out = q >= 255 ? 255 : (q < 0 ? 0 : (u8)q)                      (IEEE fp32, no contraction)
// optionally followed by relu<u8> (src/functional.cc:15-26): out = max(out, zp_out).
//
// Evaluation modes, all bit-identical to that sequence:
//   I8IE_RQ_EXACT    the sequence itself.
//   I8IE_RQ_GUARDED  e = fma((float)C, ms, zp - 0.5), ms = fl(s_in*s_w/s_out), packed with v_cvt_pk_u8_f32
//                    (round-to-nearest-even, saturate); any dword holding a value closer than 2^-13 to a
//                    rounding boundary replays the exact sequence (error analysis below).
//   I8IE_RQ_PROVEN   (diagnostic build only, -DI8IE_DIAG) an estimate of the same form, fma((float)C, ms', bias'),
//                    with NO guard: i8ie_requant_fit() picks ms', bias' on the host and proves, by finding every
//                    step of both step functions, that the estimate equals the exact sequence for every int32 C.
//                    Measured in round 2: the fit fails on every AlexNet conv layer, so the product keeps the guard.
#pragma once
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>

enum { I8IE_RQ_EXACT = 0, I8IE_RQ_GUARDED = 1, I8IE_RQ_PROVEN = 2 };

struct I8ieRequant {
  float sa, sb, sc, zpf, ms;
  int fast;  // I8IE_RQ_*
};

// src/quantize_utils.cc:30-33, the exact sequence (+ relu lower bound `lo`, 0 when relu is not fused)
__host__ __device__ __forceinline__ int i8ie_requant_exact(float cf, const I8ieRequant& q, int lo) {
  const float deq = (cf * q.sa) * q.sb;
  const float v = deq / q.sc + q.zpf;
  const int u = (v >= 255.0f) ? 255 : ((v < 0.0f) ? 0 : (int)v);
  return u > lo ? u : lo;
}

#if defined(__HIPCC__)
// e = fma(cf, ms, zp - 0.5): an estimate of (reference value v) - 0.5.  While -1 < v < 256,
// |v - (e + 0.5)| < 9.2e-5 (reference: 3 roundings on |C*s_in*s_w/s_out| < 256 and one on |v| < 257; e: one
// rounding of ms, one of the fma).  So if e is further than 2^-13 from every half-integer, v lies strictly
// inside the unit interval [k, k+1) with k = rne(e), and the reference's trunc + clamp equals sat_u8(rne(e)),
// which is what v_cvt_pk_u8_f32 computes.  Outside (-1, 256) both sides clamp, with the same margin.  relu
// (max with zp_out) commutes with the monotone rounding: rne(max(e, lo)) for the integer lo.
__device__ __forceinline__ uint32_t i8ie_requant_pack4(const int (&c)[4], const I8ieRequant& q, int lo, float lof) {
  uint32_t packed = 0;
  float worst = q.fast ? 1.0f : 0.0f;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const float e = __builtin_fmaf((float)c[r], q.ms, q.zpf - 0.5f);
    packed = __builtin_amdgcn_cvt_pk_u8_f32(__builtin_fmaxf(e, lof), r, packed);
    worst = __builtin_fminf(worst, __builtin_fabsf(__builtin_amdgcn_fractf(e) - 0.5f));
  }
#if defined(I8IE_DIAG)
  if (q.fast == I8IE_RQ_PROVEN) return packed;
#endif
  if (worst >= 1.220703125e-4f) return packed;  // 2^-13 > 9.2e-5, the proven bound
  packed = 0;
#pragma unroll
  for (int r = 0; r < 4; ++r) packed |= (uint32_t)i8ie_requant_exact((float)c[r], q, lo) << (8 * r);
  return packed;
}

// Without the ReLU (callers that apply max(., zp_out) later, e.g. behind a max-pool, which it commutes with): the
// estimate needs no lower clamp at all -- v_cvt_pk_u8_f32 saturates at 0 -- so a value costs one instruction less.
__device__ __forceinline__ uint32_t i8ie_requant_pack4_norelu(const int (&c)[4], const I8ieRequant& q) {
  uint32_t packed = 0;
  float worst = q.fast ? 1.0f : 0.0f;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const float e = __builtin_fmaf((float)c[r], q.ms, q.zpf - 0.5f);
    packed = __builtin_amdgcn_cvt_pk_u8_f32(e, r, packed);
    worst = __builtin_fminf(worst, __builtin_fabsf(__builtin_amdgcn_fractf(e) - 0.5f));
  }
  if (worst >= 1.220703125e-4f) return packed;
  packed = 0;
#pragma unroll
  for (int r = 0; r < 4; ++r) packed |= (uint32_t)i8ie_requant_exact((float)c[r], q, 0) << (8 * r);
  return packed;
}

// The same in two halves, for epilogues that keep several independent packs in flight without a branch between them:
// the estimate (+ how far the closest value is from a rounding boundary), and the exact replay for a pack whose
// `worst` came out below 2^-13 (or whose scales do not allow the estimate: worst = 0 then).
__device__ __forceinline__ uint32_t i8ie_requant_est4(const int (&c)[4], const I8ieRequant& q, float lof, float& worst) {
  uint32_t packed = 0;
  float w = q.fast ? 1.0f : 0.0f;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const float e = __builtin_fmaf((float)c[r], q.ms, q.zpf - 0.5f);
    packed = __builtin_amdgcn_cvt_pk_u8_f32(__builtin_fmaxf(e, lof), r, packed);
    w = __builtin_fminf(w, __builtin_fabsf(__builtin_amdgcn_fractf(e) - 0.5f));
  }
  worst = w;
  return packed;
}
__device__ __forceinline__ bool i8ie_requant_est_ok(float worst) { return worst >= 1.220703125e-4f; }
__device__ __forceinline__ uint32_t i8ie_requant_exact4(const int (&c)[4], const I8ieRequant& q, int lo) {
  uint32_t packed = 0;
#pragma unroll
  for (int r = 0; r < 4; ++r) packed |= (uint32_t)i8ie_requant_exact((float)c[r], q, lo) << (8 * r);
  return packed;
}

#if defined(I8IE_DIAG)
// (Measured and dropped: the same estimate two values at a time with v_pk_fma_f32 / v_pk_add_f32 / v_min3_f32,
// 5.5 instead of 7 instructions per value, made the conv2-5 kernels 8-12 % SLOWER on MI355X: the packed fp32
// instructions do not issue at the rate of the plain ones next to MFMA waves.  DESIGN.md section 4.)
// the unguarded estimate alone (kernels compiled for I8IE_RQ_PROVEN): 4 VALU operations per value
__device__ __forceinline__ uint32_t i8ie_requant_pack4_proven(int c0, int c1, int c2, int c3, float ms, float bias,
                                                              float lof) {
  uint32_t packed = 0;
  packed = __builtin_amdgcn_cvt_pk_u8_f32(__builtin_fmaxf(__builtin_fmaf((float)c0, ms, bias), lof), 0, packed);
  packed = __builtin_amdgcn_cvt_pk_u8_f32(__builtin_fmaxf(__builtin_fmaf((float)c1, ms, bias), lof), 1, packed);
  packed = __builtin_amdgcn_cvt_pk_u8_f32(__builtin_fmaxf(__builtin_fmaf((float)c2, ms, bias), lof), 2, packed);
  packed = __builtin_amdgcn_cvt_pk_u8_f32(__builtin_fmaxf(__builtin_fmaf((float)c3, ms, bias), lof), 3, packed);
  return packed;
}
#endif  // I8IE_DIAG
#endif

// ---- host side --------------------------------------------------------------------------------------------
inline I8ieRequant i8ie_make_requant(float s_in, float s_w, float s_out, int zp_out) {
  I8ieRequant r;
  r.sa = s_in; r.sb = s_w; r.sc = s_out; r.zpf = (float)zp_out;
  const double ms = (double)s_in * (double)s_w / (double)s_out;
  r.ms = (float)ms;
  // estimate only for ordinary positive finite scales; anything else takes the exact sequence
  r.fast = (s_in > 1e-30f && s_w > 1e-30f && s_out > 1e-30f && s_in < 1e30f && s_w < 1e30f && s_out < 1e30f &&
            ms > 1e-30 && ms < 1e30) ? I8IE_RQ_GUARDED : I8IE_RQ_EXACT;
  return r;
}

#if defined(I8IE_DIAG)
// ---- proof of the unguarded estimate --------------------------------------------------------------------
// Both the exact sequence and sat_u8(rne(max(fma(cf, ms, bias), lo))) are non-decreasing step functions of the
// integer C with values in [lo, 255] (positive scales), so they are equal on every int32 iff they agree at both
// ends of the range and every one of the (at most 255) steps sits at the same C.  i8ie_requant_fit() finds each
// step of the exact sequence by bisection over C, then looks for (ms, bias) whose estimate steps at exactly
// those C: the nominal ms = fl(s_in*s_w/s_out) first, then its float neighbours (scales that are "round", like
// 0.025 * 0.002 / 0.05, put accumulators exactly on rounding boundaries, where a neighbour of ms reproduces the
// reference's tie-breaking and the nominal value does not).  Every candidate is verified by the same bisection
// in float arithmetic before it is accepted; when none passes, the caller keeps the guarded mode.
template <typename F>
inline int64_t i8ie_first_c_reaching(F&& f, int level) {  // smallest C in int32 with f(C) >= level, or 2^31
  int64_t lo = -2147483648LL, hi = 2147483648LL;
  while (lo < hi) {
    const int64_t mid = lo + (hi - lo) / 2;
    if (f((float)(int32_t)mid) >= level) hi = mid; else lo = mid + 1;
  }
  return lo;
}
inline int i8ie_estimate_host(float cf, float ms, float bias, int lo) {
  float e = std::fmaf(cf, ms, bias);
  if (e < (float)lo) e = (float)lo;
  if (!(e > 0.0f)) return 0;  // also NaN
  if (e >= 255.0f) return 255;
  return (int)std::nearbyintf(e);  // default rounding mode: to nearest, ties to even (= v_cvt_pk_u8_f32)
}
// On success *ms_out, *bias_out make i8ie_requant_pack4_proven() exact for every int32 accumulator.
inline bool i8ie_requant_fit(const I8ieRequant& q, int lo, float* ms_out, float* bias_out) {
  if (q.fast == I8IE_RQ_EXACT) return false;
  auto ex = [&](float cf) { return i8ie_requant_exact(cf, q, lo); };
  int64_t thr[256];
  for (int level = lo + 1; level <= 255; ++level) thr[level] = i8ie_first_c_reaching(ex, level);
  const int lo_end = ex(-2147483648.0f), hi_end = ex(2147483648.0f);
  auto verify = [&](float ms, float bias) {
    auto es = [&](float cf) { return i8ie_estimate_host(cf, ms, bias, lo); };
    if (es(-2147483648.0f) != lo_end || es(2147483648.0f) != hi_end) return false;
    for (int level = lo + 1; level <= 255; ++level)
      if (i8ie_first_c_reaching(es, level) != thr[level]) return false;
    return true;
  };
  const float bias0 = q.zpf - 0.5f;
  if (verify(q.ms, bias0)) {
    *ms_out = q.ms;
    *bias_out = bias0;
    return true;
  }
  // a step at C = thr[level] needs fma(thr, ms, b) > level - 0.5 and fma(thr - 1, ms, b) < level - 0.5: for a
  // given ms an interval of b; scan the float neighbours of ms outwards
  float up = q.ms, dn = q.ms;
  for (int step = 0; step <= 4096; ++step) {
    for (int side = 0; side < 2; ++side) {
      const float ms = side ? dn : up;
      if (step == 0 && side == 1) continue;
      double lower = -1e300, upper = 1e300;
      for (int level = lo + 1; level <= 255; ++level) {
        if (thr[level] >= 2147483648LL || thr[level] <= -2147483648LL) continue;  // never / always reached: ends cover it
        const double a = (double)level - 0.5 - (double)(float)(int32_t)thr[level] * (double)ms;
        const double b = (double)level - 0.5 - (double)(float)(int32_t)(thr[level] - 1) * (double)ms;
        if (a > lower) lower = a;
        if (b < upper) upper = b;
      }
      if (upper - lower > 1e-4) {
        const float bias = (float)(0.5 * (lower + upper));
        if (verify(ms, bias)) {
          *ms_out = ms;
          *bias_out = bias;
          return true;
        }
      }
    }
    up = std::nextafterf(up, 3.0e38f);
    dn = std::nextafterf(dn, 0.0f);
  }
  return false;
}
#endif  // I8IE_DIAG
