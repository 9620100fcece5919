// i8ie_flin.hip -- Linear::forward_prop(Tensor<u8_t>&&) (src/fully_connected.cc:22-52) for FEW input rows, in ONE launch.
//
//   C[r][j] = sum_k A_u8[r][k] * W_s8[j][k] + oc[j];  C += (int)bias;  out = relu?(down_scale(C))
//
// At m <= 128 rows (the per-GPU shard of a 1000-image batch on 8 GPUs) fc6 / fc7 are a stream over the weights
// (37.7 / 16.8 MB) with 2 m operations per byte.  The tiled kernel needs split-K to fill the chip at such m, and its
// partial sums then cost a second launch: 33 + 21 us for fc6 + fc7 at 125 rows, a fifth of the whole AlexNet step.
// Here one block owns 16 output features for 128 rows and ALL of K, so no partial sum leaves the CU:
//   * K is walked in chunks of 256 bytes.  A chunk of the activations (128 rows x 256 B) and of the block's 16 weight
//     rows (16 x 256 B) lands in one of four LDS stages by LDS-DMA (1 KiB per wave-instruction), three chunks in
//     flight, one barrier per chunk; 16-byte slot c of row r sits at c ^ (r & 15), which the DMA lane realises by
//     picking its source slot, so the ds_read_b128 fragments of both operands are conflict-free.
//   * wave w owns rows 16 w .. 16 w + 15: one 16 x 16 output tile, 4 MFMAs (v_mfma_i32_16x16x64_i8) per chunk, no
//     reduction across waves, epilogue (oc', float bias, down_scale, ReLU) straight from its 4 accumulator registers.
//   * every block starts its K walk at a different chunk (integer sums are exact in any order): 256 CUs asking L2 for
//     the same activation lines at the same moment would queue on those lines' channels.
// The activations (m x K <= 2.4 MB) are read by every CU from L2 (that is the bound: 1.15 MB per CU for fc6); the
// weights come from HBM exactly once chip-wide.
// Round 3: the block shape is a template parameter.  <8, 1> is the form above (128 rows x 16 features, up to 64 rows);
// <4, 2> = 64 rows x 32 features (wave = row tile w % 4, feature tile w / 4) moves 24 KB per chunk for the same 524 k MACs
// instead of 36: fc6 at 125 rows is 2 x 128 blocks of 0.86 MB instead of 256 of 1.3 MB, and row counts up to 256 (the
// 4-GPU shard) stay one launch without partial sums (four row groups, three stages so that two blocks share a CU).  (A first form of this kernel read both operands as fragments
// straight from global memory, 16-byte accesses 32 bytes apart: 39 us for fc6)
#include "i8ie_calls.h"
#include "i8ie_internal.h"
#include "i8ie_requant.h"

namespace {

typedef int v4i __attribute__((ext_vector_type(4)));

struct FlinArgs {
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
};

constexpr int kChunk = 256;  // K bytes per chunk

// RT: 16-row tiles per block (8 or 4); FT: 16-feature tiles per block (RT * FT = 8 waves, one MFMA tile each);
// STAGES: LDS stages (chunks in flight = STAGES - 1)
template <int RT, int FT, int STAGES>
struct FlinShape {
  static constexpr int kRows = RT * 16, kFeats = FT * 16;
  static constexpr int kPA = RT * 4, kPB = FT * 4;          // 1-KiB DMA pieces of a chunk: 4 rows each
  static constexpr int kPPW = (kPA + kPB + 7) / 8;          // pieces per wave and chunk (every wave issues the same number)
  static constexpr int kStageA = kRows * kChunk, kStageB = kFeats * kChunk;
  static constexpr int kStage = kPPW * 8 * 1024;            // (pieces past kPA + kPB land in the stage's padding)
  static constexpr int kLds = STAGES * kStage;
};

template <int RT, int FT, int STAGES>
__global__ __launch_bounds__(512) void flin_kernel(FlinArgs p) {
  using S = FlinShape<RT, FT, STAGES>;
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lq = lane >> 4, lr = lane & 15;
  const int wr = wave % RT, wf = wave / RT;  // this wave's row tile / feature tile
  const int n0 = blockIdx.x * S::kFeats, row0 = blockIdx.y * S::kRows;

  const size_t a_off = (size_t)row0 * p.lda;
  const size_t a_left = p.a_bytes - a_off;
  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(p.A + a_off), 0,
                                                                       (unsigned)(a_left < 0xFFFFF000u ? a_left : 0xFFFFF000u), 0x00020000);
  const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<int8_t*>(p.B), 0, p.b_bytes, 0x00020000);

  // ---- DMA pieces of a chunk: kPA of A (4 rows each), kPB of B (4 feature rows each), the rest padding; wave w issues
  //      pieces w, w + 8, ...  Lane l of a piece: row 4 p + (l >> 4), LDS slot l & 15, source slot (l & 15) ^ (row & 15).
  const int nch = p.Kpad / kChunk;
  const int rot = (int)(blockIdx.x % (unsigned)nch);
  unsigned src[S::kPPW];
#pragma unroll
  for (int j = 0; j < S::kPPW; ++j) {
    const int pi = wave + 8 * j;
    if (pi < S::kPA) {
      const int row = 4 * pi + (lane >> 4);
      src[j] = (unsigned)row * p.lda + (unsigned)(((lane & 15) ^ (row & 15)) * 16);
    } else {
      const int frow = (4 * (pi - S::kPA) + (lane >> 4)) % S::kFeats;  // (padding pieces: a second copy of a B piece, unused)
      src[j] = (unsigned)(n0 + frow) * (unsigned)p.Kpad + (unsigned)(((lane & 15) ^ (frow & 15)) * 16);
    }
  }
  auto issue = [&](int step) {  // the chunk of walk position `step` -> stage step % STAGES
    int ch = step + rot;
    while (ch >= nch) ch -= nch;
    const unsigned k = (unsigned)ch * kChunk;
    uint8_t* st = smem + (step % STAGES) * S::kStage;
#pragma unroll
    for (int j = 0; j < S::kPPW; ++j) {
      const int pi = wave + 8 * j;  // (pi < kPA is the same for all waves of a j whenever kPA % 8 == 0: both shapes)
      if (pi < S::kPA)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (__attribute__((address_space(3))) void*)(st + pi * 1024), 16, (int)(src[j] + k), 0, 0, 0);
      else
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (__attribute__((address_space(3))) void*)(st + pi * 1024), 16, (int)(src[j] + k), 0, 0, 0);
    }
  };

  v4i acc = {0, 0, 0, 0};
  // fragment addresses inside a stage: row 16 wr + r (A) / feature 16 wf + r (B), slot (4 ks + q) ^ (r & 15)
  const int arow = (wr * 16 + lr) * kChunk, brow = S::kStageA + (wf * 16 + lr) * kChunk;
  const int nsteps = nch;
#pragma unroll
  for (int i = 0; i < STAGES - 1; ++i) issue(nsteps > i ? i : 0);
  for (int step = 0; step < nsteps; ++step) {
    // my pieces of this chunk (the STAGES - 2 chunks after it may be in flight)
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((STAGES - 2) * S::kPPW) : "memory");
    asm volatile("s_barrier" ::: "memory");  // everyone's pieces; everyone is done with the previous chunk
    issue(step + STAGES - 1);  // -> the stage of the previous chunk (past the end the walk wraps: keeps the vmcnt counts uniform)
    const uint8_t* st = smem + (step % STAGES) * S::kStage;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const int slot = ((4 * ks + lq) ^ lr) * 16;
      const v4i a = *reinterpret_cast<const v4i*>(st + arow + slot) ^ (int)0x80808080;  // u8 -> s8 (128 * wsum is in ocp)
      const v4i b = *reinterpret_cast<const v4i*>(st + brow + slot);
      acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(b, a, acc, 0, 0, 0);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (the wrapped pieces: nothing may land after the block is gone)

// ---- epilogue of src/fully_connected.cc:42-48: lane (q, r) holds features 4 q .. 4 q + 3 of row 16 w + r
  const int row = row0 + wr * 16 + lr, n = n0 + wf * 16 + 4 * lq;
  if (row < p.M && n < p.N) {
    int cv[4] = {acc.x, acc.y, acc.z, acc.w};
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      if (n + r < p.N) {
        cv[r] += p.ocp[n + r];
        if (p.acc != nullptr) p.acc[(size_t)row * p.N + n + r] = cv[r];
        if (p.biasf != nullptr) cv[r] = (int)((float)cv[r] + p.biasf[n + r]);
      }
    }
    const uint32_t packed = i8ie_requant_pack4(cv, p.rq, p.relu_lo, (float)p.relu_lo);
    uint8_t* o = p.out + (size_t)row * p.N + n;
    if (n + 3 < p.N && (p.N & 3) == 0) {
      *reinterpret_cast<uint32_t*>(o) = packed;
    } else {
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (n + r < p.N) o[r] = (uint8_t)(packed >> (8 * r));
    }
  }
}

}  // namespace

// Which shape: up to 64 rows the 128 x 16 form (256 blocks for 4096 features; rows past m cost no traffic: out-of-bounds
// lanes of a DMA piece fetch nothing); 65-256 rows the 64 x 32 form (fc6 at 125 rows 21 -> 14 us; at 250 rows one launch
// instead of split-K + reduction); beyond that every CU would stream the activations too often and the tiled kernel wins.
// Enough features to give half the CUs a block (a block's time does not depend on N: it streams its rows' activations).
// `force` (kernel variant 80) lifts the feature threshold, for tests.
bool i8ie_flin_wants(int m, int n, int Kpad, bool force) {
  return m <= 256 && n >= (force ? 16 : 2048) && Kpad >= 1024 && Kpad % kChunk == 0;
}

template <int RT, int FT, int STAGES>
static int flin_launch_t(i8ie_ctx* ctx, const FlinArgs& a, int M, int N) {
  using S = FlinShape<RT, FT, STAGES>;
  static bool raised[64] = {};
  const int dev = ctx->device & 63;
  if (!raised[dev]) {
    I8IE_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&flin_kernel<RT, FT, STAGES>), hipFuncAttributeMaxDynamicSharedMemorySize, S::kLds));
    raised[dev] = true;
  }
  const dim3 grid((unsigned)((N + S::kFeats - 1) / S::kFeats), (unsigned)((M + S::kRows - 1) / S::kRows));
  flin_kernel<RT, FT, STAGES><<<grid, 512, S::kLds, ctx->stream>>>(a);
  I8IE_LAUNCH_CHECK();
  return I8IE_OK;
}

int i8ie_flin_launch(i8ie_ctx* ctx, const I8ieIgemmCall& c) {
  I8IE_REQUIRE(c.amode == 0 && c.M > 0 && c.M <= 256 && c.N > 0 && c.Kpad % kChunk == 0, "flin: shape");
  I8IE_REQUIRE((reinterpret_cast<uintptr_t>(c.A) & 15u) == 0 && (reinterpret_cast<uintptr_t>(c.B) & 15u) == 0 && c.lda % 16 == 0,
               "flin: operands must be 16-byte aligned");
  I8IE_REQUIRE((size_t)c.Npad * c.Kpad < ((size_t)1 << 32) - 4096 && (size_t)256 * c.lda + c.Kpad < ((size_t)1 << 31), "flin: offsets exceed 32 bits");
  // (the weight panel is padded to whole 128-feature rows, i8ie_layer.hip: a block's 16 or 32 feature rows always exist)
  FlinArgs a{};
  a.A = c.A; a.a_bytes = c.a_bytes; a.lda = (unsigned)c.lda; a.M = c.M;
  a.B = c.B; a.b_bytes = (unsigned)((size_t)c.Npad * c.Kpad); a.Kpad = c.Kpad; a.N = c.N;
  a.ocp = c.ocp; a.biasf = c.biasf;
  a.rq = i8ie_make_requant(c.s_in, c.s_w, c.s_out, c.zp_out);
  a.relu_lo = c.relu ? c.zp_out : 0;
  a.out = c.out; a.acc = c.acc;
  const double ops = 2.0 * c.M * c.N * c.Ktrue, bytes = (double)c.M * c.Ktrue + (double)c.N * c.Ktrue + (double)c.M * c.N;
  if (c.M <= 64 || (ctx->variant == 81 && c.M <= 128)) {  // (81: the 128-row form at any row count up to 128, for A/B runs)
    I8ieProfScope prof(ctx, "flin_128x16", ops, bytes);
    return flin_launch_t<8, 1, 4>(ctx, a, c.M, c.N);
  }
  I8ieProfScope prof(ctx, "flin_64x32", ops, bytes);
  if (c.M <= 128) return flin_launch_t<4, 2, 4>(ctx, a, c.M, c.N);
  return flin_launch_t<4, 2, 3>(ctx, a, c.M, c.N);  // (72 KB of LDS: two blocks per CU)
}
