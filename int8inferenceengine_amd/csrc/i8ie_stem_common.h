// i8ie_stem_common.h -- what the first-stage kernels share: argument block, strip table, LDS helpers.  Included by
// csrc/i8ie_stem.hip (the product kernel) and tools/diag/csrc/i8ie_stem_fused.hip (the every-wave-in-both-roles experiment).
#pragma once
#include <cstdint>
#include <type_traits>

#include "i8ie_internal.h"
#include "i8ie_requant.h"

// (at global scope: the diagnostic build passes it between translation units)
struct StemArgs {
  const uint8_t* img;  // s2d image [n][HY][WX][48], bytes re-biased (^0x80)
  unsigned img_pitch;  // bytes per image
  int n_img;
  int WX, rowB;        // s2d pixels per row; bytes per s2d row
  int OH, OW, sq;      // conv output dims; stride / 4
  int KC4, RC, nch;    // s2d rows a window spans; 16-byte chunks per run (3 x s2d pixels per run); chunks in all
  int NR, first, T;    // conv rows per strip; rows of a part's first strip; strips per part (the longest part's count)
  int parts, lg_parts; // an image is cut into 2^lg_parts parts of whole pooled rows, each a unit of its own (small batches:
                       // more units than CUs); a part recomputes the pk - ps conv rows it shares with the part above it
  int pk, ps, PH, PW;  // pool window / stride (1, 1 = no pool), output dims after the pool
  const int8_t* B;     // [N][Kpad] K ordered (s2d row, s2d px, row-in-4, px-in-4, ch), zero padded
  int Kpad, N;
  const int32_t* ocp;
  I8ieRequant rq;
  int relu_lo;
  uint8_t* out;  // NHWC [n][PH + 2 ob][PW + 2 ob][N]
  int ob;
  unsigned xor_out;  // 0, or 0x80808080: output stored re-biased (I8IE_LAYOUT_NHWC_S8)
  int32_t* acc;      // ACC kernels: [n * OH * OW][N] pre-requant accumulators of the convolution
  int pitchP, ringRowB, RING;           // INT32 ring: bytes per pixel (4 N + 16), per conv row, rows
  int patchB;                            // bytes of a patch buffer (whole 1 KiB pieces)
  int lds_patch, lds_ring, lds_ocp, lds_tab, lds_adv, lds_dump, lds_bfrag;  // LDS offsets (lds_dump: 2 KiB that lanes past a strip's last pixel store into; lds_bfrag: the last feature group's weights)
  unsigned out_bytes;
  float rcpOW;
  unsigned long long* dbg;   // diagnostic build ($I8IE_STEM_STAMPS): per block and wave, cycles per phase; null otherwise
  int dbg_flags;             // diagnostic build ($I8IE_STEM_FLAGS): 1 = vector waves at raised priority
  int role_split;            // 0: waves 0-3 multiply, 4-7 do the vector work (a multiplying and a vector wave on every SIMD);
                             // 1: waves 0, 1, 4, 5 multiply, 2, 3, 6, 7 do the vector work (waves w and w + 4 share a SIMD: two
                             //    SIMDs multiply, two do vector work -- no vector wave sits beside an MFMA stream)
};

// Strip t of an image, tabulated once per block in LDS (8 ints): the scalar arithmetic of a strip is then two LDS reads
//   [0] lo, [1] hi: conv rows [lo, hi)          [2] byte offset of its patch inside the image, [3] bytes of the patch
//   [4] j0, [5] j1: pooled rows it completes    [6] lo % RING, [7] (j0 * ps) % RING
struct StemStrip {
  int lo, hi, poff, pbytes, j0, j1, lom, jm;
};

namespace {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

template <int B, int E, typename F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (B < E) {
    f(std::integral_constant<int, B>{});
    static_for<B + 1, E>(f);
  }
}

constexpr int kStemMaxKS = 14;   // k-steps of 32 bytes
constexpr int kStemPix = 128;    // pixels of a strip: four 32-pixel MFMA tiles, one per multiplying wave


extern __shared__ __attribute__((aligned(16))) uint8_t stem_smem[];

// workgroup barrier that waits for this wave's LDS operations only: vector-memory operations (the patch DMA two strips
// ahead, the output stores) stay in flight across it (__syncthreads() would drain them: s_waitcnt vmcnt(0))
#define STEM_BAR() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")


// the strip table and the per-part ring advance of a block, filled by its first T * parts threads (<= 256: checked on the host)
__device__ __forceinline__ void stem_fill_tables(const StemArgs& p, uint8_t* smem, int tid, int T, int RING) {
  if (tid < T * p.parts) {  // (T * parts <= 256: checked on the host)
    int part = 0, t = tid;
    while (t >= T) {
      t -= T;
      ++part;
    }
    // part = pooled rows [J0, J1) = conv rows [R0, R1); its strips: `first` rows, then NR at a time (empty past the end)
    const int J0 = part * p.PH / p.parts, J1 = (part + 1) * p.PH / p.parts;
    const int R0 = J0 * p.ps, R1 = part + 1 == p.parts ? p.OH : (J1 - 1) * p.ps + p.pk;
    auto hi_of = [&](int tt) { const int v = R0 + p.first + tt * p.NR; return v < R1 ? v : R1; };
    auto done = [&](int hi) {  // pooled rows of the image whose window ends at or below conv row hi, inside this part
      if (hi < p.pk) return J0;
      int e = (hi - p.pk) / p.ps + 1;
      e = e < J1 ? e : J1;
      return e > J0 ? e : J0;
    };
    StemStrip e;
    e.hi = hi_of(t);
    e.lo = t == 0 ? R0 : hi_of(t - 1);
    const int rows = e.hi > e.lo ? (e.hi - e.lo - 1) * p.sq + p.KC4 : 0;
    e.poff = e.lo * p.sq * p.rowB;
    e.pbytes = rows * p.rowB;
    if ((unsigned)(e.poff + e.pbytes) > p.img_pitch) e.pbytes = (int)p.img_pitch - e.poff;  // (never: the image holds every window row)
    e.j1 = done(e.hi);
    e.j0 = t == 0 ? J0 : done(e.lo);
    e.lom = (e.lo - R0) % RING;
    e.jm = (e.j0 * p.ps - R0) % RING;
    reinterpret_cast<StemStrip*>(smem + p.lds_tab)[tid] = e;
    if (t == 0) reinterpret_cast<int*>(smem + p.lds_adv)[part] = (R1 - R0) % RING;
  }
}
}  // namespace

#if defined(I8IE_DIAG)
// tools/diag/csrc/i8ie_stem_fused.hip (variant 16): every wave multiplies and pools
int i8ie_stem_fused_launch(i8ie_ctx* ctx, const StemArgs& a, int NG, int KS, int grid, int lds);
#endif
