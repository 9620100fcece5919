/* i8ie_diag.h -- entry points that exist in the diagnostic build only (tools/diag/build_diag.py). */
#ifndef I8IE_DIAG_H
#define I8IE_DIAG_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* ---- host side of the requantiser (no device, no ctx) ---------------------------
 * The ping-pong kernel (csrc/i8ie_pp.hip) requantises with a two-operation estimate
 * fma((float)C, ms, bias) -> round-to-nearest-even -> saturate, used only after
 * i8ie_requant_fit_host() has PROVEN it equal to down_scale (+ relu) of
 * src/quantize_utils.cc:27-36 for every int32 accumulator (csrc/i8ie_requant.h).
 * fit: returns 1 and the constants when such (ms, bias) exist, 0 otherwise (the
 * kernel then keeps the guarded / exact sequence).  eval: both functions on a host
 * array of accumulators, for tests. */
int i8ie_requant_fit_host(float sa, float sb, float sc, int zp_c, int relu, float* ms, float* bias);
int i8ie_requant_eval_host(float sa, float sb, float sc, int zp_c, int relu, float ms, float bias,
                           const int32_t* acc_host, int64_t n, uint8_t* exact_host, uint8_t* estimate_host);

#ifdef __cplusplus
}
#endif
#endif
