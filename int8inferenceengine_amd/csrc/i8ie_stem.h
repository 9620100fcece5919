// i8ie_stem.h -- the first-stage kernel of i8ie_stem.hip as seen by i8ie_layer.hip
#pragma once
#include <cstddef>
#include <cstdint>

struct i8ie_ctx;

struct I8ieStemCall {
  const float* x;      // FP32 NCHW input (quantised on the way in with q_scale / q_zp), or nullptr when xu8 is given
  const uint8_t* xu8;  // u8 NCHW input (already quantised: zero point q_zp), or nullptr
  uint8_t* scratch;    // room for the space-to-depth image: i8ie_stem_scratch_bytes()
  int n, c, h, w;
  float q_scale;
  int q_zp;
  int KH, KW, stride, pad, OH, OW;
  const int8_t* B;  // [N][Kpad], K index i8ie_stem_kindex(), Kpad = i8ie_stem_kpad(), zero padded
  int Kpad, N;
  const int32_t* ocp;
  float s_in, s_w, s_out;
  int zp_out, relu;
  int pool_k, pool_s;  // max_pool2d folded in behind the (relu'd) convolution: window / stride; pool_k <= 1: none
  uint8_t* out;        // NHWC u8 [n][PH + 2 ob][PW + 2 ob][N] (PH, PW: after the pool), interior only
  int ob;
  int out_s8;    // 1: bytes stored re-biased (^0x80), the I8IE_LAYOUT_NHWC_S8 form
  int32_t* acc;  // null, or [n * OH * OW][N]: the convolution's pre-requant accumulators
};
int i8ie_stem_supported(int c, int stride, int N, int KH, int KW, int OH, int OW, int pool_k, int pool_s);
int i8ie_stem_kpad(int KH, int KW);
int i8ie_stem_kindex(int KW, int ch, int kh, int kw);
size_t i8ie_stem_scratch_bytes(int n, int KH, int KW, int stride, int OH, int OW);
int i8ie_stem_launch(i8ie_ctx* ctx, const I8ieStemCall& c);
