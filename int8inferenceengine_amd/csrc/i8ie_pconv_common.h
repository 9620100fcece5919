// i8ie_pconv_common.h -- what the patch-stationary convolution kernels share (i8ie_pconv.hip: 8 waves, 16 x 16 x 64 MFMA tiles,
// epilogue behind the K loop; tools/diag/csrc/i8ie_dconv.hip: 4 waves with two accumulator sets, 32 x 32 x 32 tiles, the requantiser of one set
// inside the other set's K loop): the kernel argument block, the LDS plan it describes, and a few device helpers.
#pragma once
#include <cstdint>

#include "i8ie_calls.h"
#include "i8ie_internal.h"
#include "i8ie_requant.h"

typedef int v4i __attribute__((ext_vector_type(4)));
typedef unsigned v2u __attribute__((ext_vector_type(2)));
typedef unsigned v4u __attribute__((ext_vector_type(4)));

static constexpr unsigned kRowInvalid = 0xC0000000u;  // beyond any output buffer this kernel accepts (< 2^31 bytes)
static constexpr int kTabPix = 256;                   // pixels per tile at most (16 MFMA row tiles)

struct PCArgs {
  const uint8_t* A;
  unsigned a_bytes;
  unsigned img_pitch, row_pitch, C;
  int OH, OW, s, KH, KW, CC, Wp;
  float rcpOW, rcpCC1;
  int RT, bands, n_tiles;  // output rows per tile, tiles per image, tiles in all
  int P;                   // LDS pixel pitch (bytes) = C + 16
  int row_gran;            // granules of a patch row in LDS: Wp (CC + 1) + pad, so that consecutive output pixels keep
                           // walking the 16 slots of 256 B across the row wrap (row pitch / 16 = OW s P / 16 mod 16)
  float rcpRowGran;
  int patch_gran;          // 16-byte granules of a full patch
  const int8_t* Bf;        // [pass][kt][ks][ntile][lane][16]
  unsigned bf_bytes;
  int nkt;                 // K tiles of 8 chunks
  const int* perm;         // [nkt * 8]: source chunk (tap * CC + channel chunk) of K position ci, -1 = zero padding
  int N, npass;
  const int32_t* ocp;
  int Npad;
  I8ieRequant rq;
  int relu_lo;
  uint8_t* out;
  unsigned out_bytes;
  int ob, OHp, OWp;
  int lds_patch, lds_ocp, lds_tab, lds_ktab, lds_src, lds_prog;  // LDS offsets (lds_src < 0: no room for the source table)
  int split;                // 1: a (band, feature pass) pair is a unit of its own (small batches: more units than CUs)
  int flags;                // 1 = weights of the next pass fetched by the last K tile (not with split)
  int32_t* acc;             // ACC kernels: [M][N] pre-requant accumulators (the cblas_gemm_s8u8s32 result, src/conv2d.cc:131-133)
  int a_s8;                 // 1: the input bytes are stored re-biased already (I8IE_LAYOUT_NHWC_S8): no xor pass over the patch
  unsigned xor_out;         // 0x80808080: the output is stored re-biased (I8IE_LAYOUT_NHWC_S8), else 0
  // POOL kernels: max_pool2d (pk x pk, stride ps) behind the (relu'd) convolution.  `out` is then the POOLED tensor
  // [n][PH + 2 ob][PW + 2 ob][N]; the requantised rows of a band go to an LDS ring of RB conv rows (pixel pitch opitch),
  // and the bands of an image run back to back in one block (seq) so that the rows a window shares with the previous
  // band are still there
  int pk, ps, PH, PW, RB, opitch, lds_otile, seq;
  float rcpPW, rcpC16, rcpRB;
  unsigned long long* dbg;  // diagnostic build, variant 51: per block, cycles spent per phase (wave 0); null otherwise
};

namespace {

#define PC_BAR() asm volatile("s_barrier" ::: "memory")
template <int N>
__device__ __forceinline__ void pc_wait_vm() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
__device__ __forceinline__ void pc_divmod(int x, int d, float rd, int& qo, int& ro) {  // 0 <= x < 2^23
  int qq = (int)((float)x * rd);
  int rr = x - qq * d;
  if (rr < 0) {
    rr += d;
    --qq;
  } else if (rr >= d) {
    rr -= d;
    ++qq;
  }
  qo = qq;
  ro = rr;
}

extern __shared__ __attribute__((aligned(16))) uint8_t pc_smem[];

// MFMA row r of a 16-row tile <-> pixel offset inside the tile (see the header: even pixels for rows 0-3, 12-15)
__device__ __forceinline__ int pc_row_to_pix(int r) { return (r < 4) ? 2 * r : (r >= 12 ? 2 * (r - 8) : 2 * (r - 4) + 1); }


}  // namespace

#if defined(I8IE_DIAG)
// tools/diag/csrc/i8ie_dconv.hip (diagnostic build): launches the deferred-epilogue kernel for a call pconv_impl (i8ie_pconv.hip) planned; `a` as for pconv_kernel
// except Bf / perm (packed for this kernel's fragment order here, cached in the layer handle).  Returns I8IE_OK or an error.
bool i8ie_dconv_eligible(int split, int nkt, int npass, int patch_gran, int PT, int bn, bool pool, int N);
int i8ie_dconv_launch(i8ie_ctx* ctx, const I8ieIgemmCall& c, PCArgs a, const int* perm_host, int PT, int bn, int grid, int lds, const char* name);
#endif
