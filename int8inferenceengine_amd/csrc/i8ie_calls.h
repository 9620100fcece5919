// i8ie_calls.h -- argument blocks passed between the translation units of libi8ie_hip.so
#pragma once
#include <cstddef>
#include <cstdint>

#include <vector>

struct i8ie_ctx;

// Weights re-packed by a contraction kernel for its own K walk, cached in the layer handle: one buffer per packing
// key (the fragment order depends on the pass width, which depends on the batch size), never overwritten once made --
// a captured HIP graph keeps replaying the address it was captured with -- and freed with the layer.
struct I8ieWCache {
  struct Ent {
    unsigned long long key;
    void* buf;
  };
  std::vector<Ent> ents;
  void* find(unsigned long long key) const {
    for (const Ent& e : ents)
      if (e.key == key) return e.buf;
    return nullptr;
  }
};

// one contraction launch: Linear (amode 0) or implicit-GEMM Conv2d over bordered NHWC (amode 1)
struct I8ieIgemmCall {
  const uint8_t* A;   // amode 0: [M][lda]; amode 1: window origin of pixel (0,0) in a bordered NHWC input
  size_t a_bytes;     // bytes readable from A
  int amode;
  long lda;
  int M, Kchunks;
  int Hp, Wp, C, KH, KW, sh, sw, OH, OW;  // amode 1: physical input dims, kernel, strides, output dims
  const int8_t* B;
  int Kpad, Npad, N;
  const int32_t* ocp;
  const float* biasf;
  float s_in, s_w, s_out;
  int zp_out, relu;
  uint8_t* out;
  int ob;  // physical border of the NHWC output (amode 1 only)
  int32_t* acc;
  double Ktrue;
  int ksplit;        // amode 0 only: > 1 = split K over that many slices (partial must hold ksplit*M*N int32)
  int32_t* partial;
  I8ieWCache* wcache;  // amode 1: the layer handle's cache of weights re-packed by i8ie_pconv.hip / i8ie_tconv.hip; may be null
  // amode 1, honoured by i8ie_pconv.hip only (ask i8ie_pconv_takes first; the tiled kernel ignores them):
  int pool_k, pool_s;  // max_pool2d behind the (relu'd) convolution: `out` is then the pooled tensor, ob its border
  int a_s8, out_s8;    // input bytes / output bytes stored re-biased (^0x80: I8IE_LAYOUT_NHWC_S8)
};
// ONE definition of "this call carries a max-pool" for every file: a 1 x 1 window with a stride > 1 subsamples, so it IS a
// pool (src/functional.cc:36-64 makes no exception for it); kernels that fold pools take k > 1 only and must decline it
inline bool i8ie_is_pool(int pool_k, int pool_s) { return pool_k > 1 || (pool_k == 1 && pool_s > 1); }
int i8ie_igemm_launch(i8ie_ctx* ctx, const I8ieIgemmCall& c);
size_t i8ie_igemm_chunk_limit();  // activations at or beyond this many bytes run as several launches

#if defined(I8IE_DIAG)
// tools/diag/csrc/i8ie_pp.hip (diagnostic build only): the persistent ping-pong form of the amode-1 contraction.
int i8ie_pp_try_launch(i8ie_ctx* ctx, const I8ieIgemmCall& c);
#endif

// i8ie_pconv.hip: the patch-stationary form of the amode-1 contraction (input patch resident in LDS, weights
// streamed in fragment order).  Returns 1 when it took the launch, 0 when the shape is not its (the caller then
// runs the tiled kernel), < 0 on error.
int i8ie_pconv_try_launch(i8ie_ctx* ctx, const I8ieIgemmCall& c);
int i8ie_pconv_takes(i8ie_ctx* ctx, const I8ieIgemmCall& c);  // dry run of the same decision: 1 / 0

// i8ie_tconv.hip: the patch-stationary contraction with two wave teams half a tile apart (epilogues, patch waits
// and re-bias passes of one team under the MFMAs of the other).  Same return convention.
int i8ie_tconv_try_launch(i8ie_ctx* ctx, const I8ieIgemmCall& c);

// i8ie_flin.hip: Linear for few rows (m <= 256) in one launch: 16 features x 128 rows x all of K per block, both
// operands through LDS stages.  Takes the amode-0 fields of the call (ksplit / partial unused).
bool i8ie_flin_wants(int m, int n, int Kpad, bool force);
int i8ie_flin_launch(i8ie_ctx* ctx, const I8ieIgemmCall& c);

// i8ie_mlin.hip: Linear for many rows (m > 256) in one launch: 128 rows x 128 features x all of K per block (one per CU), both
// operands through a four-stage LDS-DMA ring fed by dedicated loader waves.  Takes the amode-0 fields of the call.
bool i8ie_mlin_wants(int m, int n, int Kpad, bool force);
int i8ie_mlin_launch(i8ie_ctx* ctx, const I8ieIgemmCall& c);

#if defined(I8IE_DIAG)
// tools/diag/csrc/i8ie_lgemm.hip (diagnostic build, variant 82): Linear for many rows (m > 256): 64 x 128 block tiles, weights
// straight from L2 in fragment order, activations register-staged through LDS.  Round 3: 57 + 30 us for fc6 + fc7 at 1000
// rows on their own (tiled kernel: 65 + 34), 98.5 us inside the AlexNet step (tiled: 99.5): not worth 72 MB of packed weights.
bool i8ie_lgemm_wants(int m, int n, int K, int Kpad);
int i8ie_lgemm_launch(i8ie_ctx* ctx, const I8ieIgemmCall& c, bool perm_panel);
#endif
