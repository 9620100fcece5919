// i8ie_mlin.hip -- Linear::forward_prop(Tensor<u8_t>&&) (src/fully_connected.cc:22-52) for MANY input rows (> 256): fc6 / fc7 of a
// 257 - 1000-image batch, in one launch without partial sums.
//
//   C[r][j] = sum_k A_u8[r][k] * W_s8[j][k] + oc[j];  C += (int)bias;  out = relu?(down_scale(C))
//
// Round 3 left fc6 + fc7 at 0.10 ms for 500 as for 1000 rows: the tiled kernel's 128 x 64 blocks pull 11 KB of operands per MOP
// (906 MB through the L2 -> CU ports for fc6, two blocks per CU coupled by a barrier per 128 bytes of K with ~50 KB in flight).
// Here:
//   * one 512-thread block per CU owns 128 rows x 128 features and ALL of K: 256 blocks for 1000 x 4096, 5.9 KB per MOP (604 MB for
//     fc6), which at the ~64 B/clk a CU takes from L2 equals its MFMA time -- the kernel is built to keep BOTH busy.
//   * Role-specialised waves.  Waves 4-7 only move bytes: K is walked in chunks of 128 bytes; a chunk (128 rows + 128 feature rows,
//     32 KiB) lands in one of FOUR LDS stages by LDS-DMA (1 KiB per wave-instruction, 8 per loader wave and chunk), two to three
//     chunks (64-96 KiB) in flight per CU, never drained.  16-byte slot c of row r sits at c ^ ((r >> 1) & 7) (the DMA lane picks
//     its source slot), so every ds_read_b128 fragment is conflict-free.
//   * Waves 0-3 only multiply: 2 x 2 waves of 64 rows x 64 features (v_mfma_i32_16x16x64_i8, 32 per chunk), the 16 fragments of
//     chunk s + 1 read into a second register set between the MFMAs of chunk s.  ONE barrier per chunk says two things: chunk
//     s + 2 has landed, and chunk s's fragments are in registers (its stage is free for chunk s + 4).
//   * Blocks that share weights share an XCD (the 8 row tiles of a 128-feature tile walk K in step: the weights cross the fabric
//     once, 37.7 MB for fc6, the activations once per XCD).
//   * Epilogue of src/fully_connected.cc:42-48 from the 64 accumulator registers: + oc', float bias, down_scale, ReLU.
//   * A second instantiation with 64-row block tiles (24 KiB per chunk, multiplying waves of 32 rows x 64 features) runs where the
//     128-row tiles would leave half the CUs without a block: 257-512 rows of fc6 / fc7 (the 500-image shard of a 2-GPU run).
#include "i8ie_calls.h"
#include "i8ie_internal.h"
#include "i8ie_requant.h"

namespace {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef unsigned v4u __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

struct MlinArgs {
  const uint8_t* A;  // [M][lda] u8
  size_t a_bytes;
  unsigned lda;
  int M;
  const int8_t* B;  // [Npad][Kpad] s8, K contiguous, zero beyond K
  unsigned b_bytes;
  int Kpad, N;
  const int32_t* ocp;  // oc + 128 * wsum (the activations enter re-biased by -128)
  const float* biasf;
  I8ieRequant rq;
  int relu_lo;
  uint8_t* out;
  int32_t* acc;
  int n_tiles, m_tiles;
  int ahead;  // a line is touched (-> L2) this many chunks before the chunk being multiplied
};

constexpr int kMlChunk = 128;                 // K bytes per chunk
constexpr int kMlFeats = 128;                 // features of a block tile
constexpr int kMlStages = 4;
// RB: rows of a block tile, 128 or -- when 128-row tiles would leave half the CUs without a block (the 500-row shard: 4 x 32 tiles
// for fc6) -- 64: twice the blocks, 24 KiB of operands per chunk for half the MFMAs (the kernel is bound by what a CU's
// vector-memory path delivers, 27 B/clk: 0.45 us per chunk instead of 0.6)
template <int RB>
struct MlShape {
  static constexpr int kRows = RB;
  static constexpr int kStage = (RB + kMlFeats) * kMlChunk;  // 32 / 24 KiB
  static constexpr int kLds = kMlStages * kStage;
  static constexpr int kPA = RB / 32, kPB = kMlFeats / 32;   // DMA pieces (8 rows of 128 bytes) per loader wave and chunk: A's, B's
  static constexpr int kPPW = kPA + kPB;
  static constexpr int kIT = RB / 32;                         // 16-row MFMA tiles per multiplying wave (2 x 2 waves)
};
// (A/B builds, tools/dbg/build_ab.sh i8ie_mlin wt-DML_AHEAD=<n> / wt-DML_NO_TOUCH=1; fc6 + fc7 inside the AlexNet step: 10 chunks ahead
//  0.0828 ms, 4: 0.0800, 2: 0.0826, no touches at all 0.0818; 20: slower)
#if !defined(ML_AHEAD)
#define ML_AHEAD 4
#endif
constexpr int kMlAhead = ML_AHEAD;                // a line is touched (-> L2) this many chunks before the chunk being multiplied

template <int N>
__device__ __forceinline__ void ml_wait_vm() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
#define ML_BAR() asm volatile("s_barrier" ::: "memory")

template <int RB>
__global__ __launch_bounds__(512, 2) void mlin_kernel(MlinArgs p) {
  using S = MlShape<RB>;
  constexpr int kMlRows = S::kRows, kMlStage = S::kStage, kMlPPW = S::kPPW, IT = S::kIT;
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // blocks b and b + 8 share an XCD: the row tiles of one feature tile go to one XCD
  const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
  const int per = (p.n_tiles + 7) >> 3;          // feature tiles per XCD
  const int nt = xcd * per + j / p.m_tiles, mt = j % p.m_tiles;
  if (nt >= p.n_tiles || j >= per * p.m_tiles) return;  // (whole block: before any barrier)
  const int n0 = nt * kMlFeats, row0 = mt * kMlRows;
  const int nch = p.Kpad / kMlChunk;
  if (wave >= 4) {
    // =============================== loader waves ===========================================================
    const int lw = wave - 4;
    const size_t a_off = (size_t)row0 * p.lda;
    const size_t a_left = p.a_bytes - a_off;
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(p.A + a_off), 0,
                                                                         (unsigned)(a_left < 0xFFFFF000u ? a_left : 0xFFFFF000u), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<int8_t*>(p.B), 0, p.b_bytes, 0x00020000);
    // this wave's pieces of a chunk: A's lw + 4 jj (jj < kPA: rows 8 pi .. 8 pi + 7 of the block's RB activation rows), then B's
    // lw + 4 jj' (jj' < kPB: of its 128 weight rows, behind A's RB rows in the stage); lane l: row 8 pi + (l >> 3), LDS slot l & 7,
    // source slot (l & 7) ^ ((row >> 1) & 7)
    unsigned src[kMlPPW];
#pragma unroll
    for (int jj = 0; jj < kMlPPW; ++jj) {
      const bool isA = jj < S::kPA;
      const int pi = lw + 4 * (isA ? jj : jj - S::kPA);
      const int row = 8 * pi + (lane >> 3);
      const unsigned slot = (unsigned)(((lane & 7) ^ ((row >> 1) & 7)) * 16);
      src[jj] = isA ? (unsigned)row * p.lda + slot : (unsigned)(n0 + row) * (unsigned)p.Kpad + slot;
    }
    auto issue = [&](int ch) {  // chunk ch -> stage ch % 4
      const unsigned k = (unsigned)ch * kMlChunk;
      uint8_t* st = smem + (ch & (kMlStages - 1)) * kMlStage;
#pragma unroll
      for (int jj = 0; jj < kMlPPW; ++jj) {
        if (jj < S::kPA)
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (__attribute__((address_space(3))) void*)(st + (lw + 4 * jj) * 1024), 16, (int)(src[jj] + k), 0, 0, 0);
        else
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (__attribute__((address_space(3))) void*)(st + kMlRows * kMlChunk + (lw + 4 * (jj - S::kPA)) * 1024), 16, (int)(src[jj] + k), 0, 0, 0);
      }
    };
    // wait until chunk c has landed, given that chunks up to `last` have been issued (this wave's pieces retire in order)
    auto wait_chunk = [&](int c, int last) {
      const int younger = last - c;
      if (younger >= 2) ml_wait_vm<2 * kMlPPW>();
      else if (younger == 1) ml_wait_vm<kMlPPW>();
      else ml_wait_vm<0>();
    };
    int issued = -1;
    for (int c = 0; c < kMlStages && c < nch; ++c) {
      issue(c);
      issued = c;
    }

    wait_chunk(0, issued);
    ML_BAR();  // B_pre: chunk 0 is in LDS
    wait_chunk(nch > 1 ? 1 : 0, issued);
    ML_BAR();  // B_0: chunk 1 is in LDS; the multiplying waves hold chunk 0's fragments
    for (int s = 0; s + 1 < nch; ++s) {
      // after B_s: stage s % 4 is free
      if (s + kMlStages < nch) {
        issue(s + kMlStages);
        issued = s + kMlStages;
      }
      wait_chunk(s + 2 < nch ? s + 2 : nch - 1, issued);
      ML_BAR();  // B_(s+1): chunk s + 2 is in LDS; chunk s + 1's fragments are in registers
    }
    return;
  }

  // =============================== multiplying waves ===========================================================
  // v_mfma_i32_16x16x64_i8, not 32x32x32: a wave that issues 32 x 32 MFMAs back to back starves the OTHER wave of its SIMD of
  // every vector and vector-memory issue slot (tools/valu_probe) -- here that other wave is the loader; beside 16 x 16 MFMAs it
  // gets a slot per MFMA (first build, 32 x 32: 0.8 us per chunk, the loaders could not issue their eight DMA pieces).
  const int wr = wave >> 1, wf = wave & 1;
  const int lq = lane >> 4, lr = lane & 15;
  // fragment addresses inside a stage: A row 16 IT wr + 16 i + lr, B feature row 64 wf + 16 n + lr; k-step ks (64 bytes of K) =
  // logical slots 4 ks + lq, physical slot = logical ^ ((row >> 1) & 7)
  int aoff[IT][2], boff[4][2];  // stage-relative byte offsets of this lane's fragments
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int ar = 16 * IT * wr + 16 * i + lr, br = 64 * wf + 16 * i + lr;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      if (i < IT) aoff[i][ks] = ar * kMlChunk + (((4 * ks + lq) ^ ((ar >> 1) & 7)) * 16);
      boff[i][ks] = kMlRows * kMlChunk + br * kMlChunk + (((4 * ks + lq) ^ ((br >> 1) & 7)) * 16);
    }
  }
  // A chunk reaches LDS two barrier intervals after its DMA was issued (four stages), and an LDS-DMA whose lines come from beyond
  // the XCD's L2 (weights: HBM, read once chip-wide; activations: 9.2 MB for fc6, more than an L2 holds) takes ~1.4 us to land:
  // 0.75 us per chunk whatever the kernel does (first builds; the tiled kernels of round 3 sat on the same wall).  So the lines are
  // TOUCHED kMlAhead chunks ahead (-> this XCD's L2, where a DMA finds them in ~0.4 us): one dword per 128-byte line, 64 lines per
  // instruction.  The blocks of an XCD that share a feature tile split its 128 weight rows (wave 0 touches this block's share), the
  // blocks that share a row tile split its 128 activation rows (wave 1).  These waves never wait for vector memory.
  const size_t ta_off = (size_t)row0 * p.lda;
  const size_t ta_left = p.a_bytes - ta_off;
  const __amdgpu_buffer_rsrc_t rsAt = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(p.A + ta_off), 0,
                                                                        (unsigned)(ta_left < 0xFFFFF000u ? ta_left : 0xFFFFF000u), 0x00020000);
  const __amdgpu_buffer_rsrc_t rsBt = __builtin_amdgcn_make_buffer_rsrc(const_cast<int8_t*>(p.B), 0, p.b_bytes, 0x00020000);
  const int rpb = (kMlFeats + p.m_tiles - 1) / p.m_tiles < 64 ? (kMlFeats + p.m_tiles - 1) / p.m_tiles : 64;  // weight rows this block touches
  const int rpa = (kMlRows + per - 1) / per < 64 ? (kMlRows + per - 1) / per : 64;                              // activation rows
  const int rpt = wave == 0 ? rpb : rpa;
  const int cpi = 64 / rpt;  // chunks per touch instruction (this wave's)
  const int t_row = (wave == 0 ? mt * rpb : (j / p.m_tiles) * rpa) + lane % rpt, t_ch = lane / rpt;
  const unsigned t_base = (t_row < (wave == 0 ? kMlFeats : kMlRows) && t_ch < cpi)
                              ? (wave == 0 ? (unsigned)(n0 + t_row) * (unsigned)p.Kpad : (unsigned)t_row * p.lda) + (unsigned)t_ch * kMlChunk
                              : 0xFFFFFF00u;
  auto touch = [&](int ch) {  // chunks ch .. ch + cpi - 1 (past the end of K: the next row's bytes, harmless; beyond the buffer: dropped)
#if defined(ML_NO_TOUCH)
    return;
#endif
    if (wave > 1 || ch >= nch) return;
    const unsigned off = t_base == 0xFFFFFF00u ? t_base : t_base + (unsigned)ch * kMlChunk;
    // (destination: v255, named as a clobber and otherwise unused -- the kernel needs ~220 registers; a compiler-visible output
    //  operand may be copied to another register between statements, and a load still in flight would then land in a register
    //  that has since been given to something else: that build computed wrong sums)
    if (wave == 0) asm volatile("buffer_load_dword v255, %0, %1, 0 offen" ::"v"(off), "s"(rsBt) : "v255", "memory");
    else asm volatile("buffer_load_dword v255, %0, %1, 0 offen" ::"v"(off), "s"(rsAt) : "v255", "memory");
  };
  for (int c = kMlStages; c < p.ahead + cpi; c += cpi) touch(c);
  int touch_in = 1;
  v4i acc[IT][4];
#pragma unroll
  for (int i = 0; i < IT; ++i)
#pragma unroll
    for (int n = 0; n < 4; ++n) acc[i][n] = v4i{0, 0, 0, 0};
  v4i FA0[IT][2], FB0[4][2], FA1[IT][2], FB1[4][2];
  auto read_chunk = [&](v4i (&fa)[IT][2], v4i (&fb)[4][2], int ch) {
    const uint8_t* st = smem + (ch & (kMlStages - 1)) * kMlStage;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if (i < IT) fa[i][ks] = *reinterpret_cast<const v4i*>(st + aoff[i][ks]);
        fb[i][ks] = *reinterpret_cast<const v4i*>(st + boff[i][ks]);
      }
  };
  auto mfma_chunk = [&](const v4i (&fa)[IT][2], const v4i (&fb)[4][2]) {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int i = 0; i < IT; ++i) {
        const v4i a = fa[i][ks] ^ (int)0x80808080;  // u8 -> s8 (128 * wsum is in ocp).  (Re-biasing in LDS by the loader lanes that
                                                    // fetched the bytes was tried: 56 -> 68 us, the pass sits in the barrier's path)
#pragma unroll
        for (int n = 0; n < 4; ++n) acc[i][n] = __builtin_amdgcn_mfma_i32_16x16x64_i8(fb[n][ks], a, acc[i][n], 0, 0, 0);
      }
  };
  // one chunk's 32 MFMAs with the next chunk's 16 fragment reads spread between them.  (Tied asm MFMAs with the reads fenced
  // between them gave a tidier stream -- no accumulator moves -- and the same time, 53 us for fc6: the step is not what bounds it)
  auto step = [&](const v4i (&fa)[IT][2], const v4i (&fb)[4][2], v4i (&na)[IT][2], v4i (&nb)[4][2], int next_ch) {
    if (--touch_in == 0) {  // (every cpi chunks; a countdown: `% cpi` with a run-time cpi cost ~300 cycles of every chunk)
      touch_in = cpi;
      touch(next_ch - 1 + p.ahead);
    }
    read_chunk(na, nb, next_ch);
    mfma_chunk(fa, fb);
#pragma unroll
    for (int q = 0; q < 4 * IT; ++q) {
      __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);                // 2 MFMAs
      __builtin_amdgcn_sched_group_barrier(0x100, IT == 4 ? 1 : 2, 0);  // 1 DS read (of 16; 64-row tiles: of 12 over 8 groups)
      __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);                // 2 VALU (the re-bias xor)
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the next chunk's fragments are in registers: its stage may be refilled
    ML_BAR();
  };
  ML_BAR();  // B_pre
  read_chunk(FA0, FB0, 0);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  ML_BAR();  // B_0
  int s = 0;
  for (; s + 2 < nch; s += 2) {
    step(FA0, FB0, FA1, FB1, s + 1);
    step(FA1, FB1, FA0, FB0, s + 2);
  }
  if (s + 1 < nch) {  // two chunks left
    step(FA0, FB0, FA1, FB1, s + 1);
    mfma_chunk(FA1, FB1);
  } else {            // one chunk left
    mfma_chunk(FA0, FB0);
  }

  asm volatile("s_waitcnt vmcnt(0)" ::: "v255", "memory");  // (the touches have landed long ago; nothing may land after the wave is gone)
  // ---- epilogue of src/fully_connected.cc:42-48: lane (lq, lr) holds row 16 IT wr + 16 i + lr and features 64 wf + 16 n + 4 lq .. + 3
  //      oc' and the float bias of this lane's 16 features come as eight 16-byte loads issued together (the first build read them
  //      one dword at a time inside the tile loop: 128 dependent round trips, 20 us of a 60 us kernel)
  const I8ieRequant rq = p.rq;
  const int lo = p.relu_lo;
  const float lof = (float)lo;
  const bool vec = (p.N & 3) == 0;  // (16-byte aligned quads: ocp / biasf are device allocations)
  typedef float v4f __attribute__((ext_vector_type(4)));
  v4i ocq[4];
  v4f bq[4];
#pragma unroll
  for (int n = 0; n < 4; ++n) {
    const int f = n0 + 64 * wf + 16 * n + 4 * lq;
    ocq[n] = v4i{0, 0, 0, 0};
    bq[n] = v4f{0.f, 0.f, 0.f, 0.f};
    if (vec) {
      if (f < p.N) {
        ocq[n] = *reinterpret_cast<const v4i*>(p.ocp + f);
        if (p.biasf != nullptr) bq[n] = *reinterpret_cast<const v4f*>(p.biasf + f);
      }
    } else {
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (f + r < p.N) {
          ocq[n][r] = p.ocp[f + r];
          if (p.biasf != nullptr) bq[n][r] = p.biasf[f + r];
        }
    }
  }
#pragma unroll
  for (int i = 0; i < IT; ++i) {
    const int row = row0 + 16 * IT * wr + 16 * i + lr;
#pragma unroll
    for (int n = 0; n < 4; ++n) {
      const int f = n0 + 64 * wf + 16 * n + 4 * lq;
      if (row < p.M && f < p.N) {
        int cv[4] = {acc[i][n].x + ocq[n].x, acc[i][n].y + ocq[n].y, acc[i][n].z + ocq[n].z, acc[i][n].w + ocq[n].w};
        if (p.acc != nullptr) {
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (f + r < p.N) p.acc[(size_t)row * p.N + f + r] = cv[r];
        }
        if (p.biasf != nullptr) {
#pragma unroll
          for (int r = 0; r < 4; ++r) cv[r] = (int)((float)cv[r] + bq[n][r]);
        }
        const uint32_t packed = i8ie_requant_pack4(cv, rq, lo, lof);
        uint8_t* o = p.out + (size_t)row * p.N + f;
        if (vec) {
          *reinterpret_cast<uint32_t*>(o) = packed;
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (f + r < p.N) o[r] = (uint8_t)(packed >> (8 * r));
        }
      }
    }
  }
}

}  // namespace

// more rows than the few-row kernel takes (256), a whole number of 128-byte chunks of K, and enough of them for the four-stage ring.
// With the 64-row tile this kernel wins from 257 rows on (fc6 + fc7 alone, us: 257 rows 45 against 63 tiled + split-K, 384 rows 46
// against 68, 500 rows 52 against 74; 128-row tiles from 513 rows: 640 rows 69, 1000 rows 80 against 98); inside the AlexNet step,
// where the weights come from HBM every time, 0.053 against 0.096 ms at 500 rows.  `force` (kernel variants 83-85) lifts the feature
// threshold for tests.
bool i8ie_mlin_wants(int m, int n, int Kpad, bool force) {
  return m > 256 && (force || n >= 2048) && Kpad % kMlChunk == 0 && Kpad >= 4 * kMlChunk;
}

template <int RB>
static int mlin_launch_t(i8ie_ctx* ctx, MlinArgs& a, const I8ieIgemmCall& c, const char* name) {
  using S = MlShape<RB>;
  I8IE_REQUIRE((size_t)RB * c.lda + c.Kpad < ((size_t)1 << 31), "mlin: offsets exceed 32 bits");
  a.m_tiles = (c.M + RB - 1) / RB;
  static bool raised[64] = {};
  const int dev = ctx->device & 63;
  if (!raised[dev]) {
    I8IE_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&mlin_kernel<RB>), hipFuncAttributeMaxDynamicSharedMemorySize, S::kLds));
    raised[dev] = true;
  }
  const int per = (a.n_tiles + 7) / 8;
  const int grid = 8 * per * a.m_tiles;
  const double ops = 2.0 * c.M * c.N * c.Ktrue, bytes = (double)c.M * c.Ktrue + (double)c.N * c.Ktrue + (double)c.M * c.N;
  I8ieProfScope prof(ctx, name, ops, bytes);
  mlin_kernel<RB><<<grid, 512, S::kLds, ctx->stream>>>(a);
  I8IE_LAUNCH_CHECK();
  return I8IE_OK;
}

int i8ie_mlin_launch(i8ie_ctx* ctx, const I8ieIgemmCall& c) {
  I8IE_REQUIRE(c.amode == 0 && c.M > 0 && c.N > 0 && c.Kpad % kMlChunk == 0 && c.Kpad >= 4 * kMlChunk, "mlin: shape");
  I8IE_REQUIRE((reinterpret_cast<uintptr_t>(c.A) & 15u) == 0 && (reinterpret_cast<uintptr_t>(c.B) & 15u) == 0 && c.lda % 16 == 0 &&
                   (reinterpret_cast<uintptr_t>(c.out) & 15u) == 0,
               "mlin: operands must be 16-byte aligned");
  I8IE_REQUIRE((size_t)c.Npad * c.Kpad < ((size_t)1 << 32) - 4096, "mlin: offsets exceed 32 bits");
  I8IE_REQUIRE(c.Npad % kMlFeats == 0, "mlin: the weight panel must be padded to whole 128-feature tiles");
  MlinArgs a{};
  a.A = c.A; a.a_bytes = c.a_bytes; a.lda = (unsigned)c.lda; a.M = c.M;
  a.B = c.B; a.b_bytes = (unsigned)((size_t)c.Npad * c.Kpad); a.Kpad = c.Kpad; a.N = c.N;
  a.ocp = c.ocp; a.biasf = c.biasf;
  a.rq = i8ie_make_requant(c.s_in, c.s_w, c.s_out, c.zp_out);
  a.relu_lo = c.relu ? c.zp_out : 0;
  a.out = c.out; a.acc = c.acc;
  a.n_tiles = (c.N + kMlFeats - 1) / kMlFeats;
  a.ahead = kMlAhead;
  static int cus[64] = {};
  const int dev = ctx->device & 63;
  if (cus[dev] == 0) {
    hipDeviceProp_t prop;
    I8IE_HIP_TRY(hipGetDeviceProperties(&prop, ctx->device));
    cus[dev] = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  }
  // 64-row tiles when 128-row tiles give at most half the CUs a block (variant 84 / 85 force 64 / 128 rows)
  const long blocks128 = (long)((c.M + 127) / 128) * a.n_tiles;
  const bool rows64 = ctx->variant == 84 || (ctx->variant != 85 && blocks128 * 2 <= i8ie_cus(ctx, cus[dev]));
  if (rows64) return mlin_launch_t<64>(ctx, a, c, "mlin_64x128");
  return mlin_launch_t<128>(ctx, a, c, "mlin_128x128");
}
