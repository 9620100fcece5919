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
// weights come from HBM exactly once chip-wide.  (A first form of this kernel read both operands as fragments
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

constexpr int kRows = 128;            // rows per block: 8 waves x one MFMA tile of 16
constexpr int kChunk = 256;           // K bytes per chunk
constexpr int kStageA = kRows * kChunk;  // 32 KB
constexpr int kStageB = 16 * kChunk;     // 4 KB
constexpr int kStage = kStageA + 2 * kStageB;  // (+ 4 KB the fifth piece of waves 4-7 lands in: every wave issues 5)
constexpr int kStages = 4;

__global__ __launch_bounds__(512) void flin_kernel(FlinArgs p) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lq = lane >> 4, lr = lane & 15;
  const int n0 = blockIdx.x * 16, row0 = blockIdx.y * kRows;

  const size_t a_off = (size_t)row0 * p.lda;
  const size_t a_left = p.a_bytes - a_off;
  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(p.A + a_off), 0,
                                                                       (unsigned)(a_left < 0xFFFFF000u ? a_left : 0xFFFFF000u), 0x00020000);
  const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<int8_t*>(p.B), 0, p.b_bytes, 0x00020000);

  // ---- DMA pieces of a chunk: 32 of A (4 rows each), 4 of B (4 feature rows each), 4 unused; wave w issues pieces
  //      w, w + 8, w + 16, w + 24 of A and piece 32 + w (B for w < 4).  Lane l of a piece: row 4 p + (l >> 4), LDS slot
  //      l & 15, source slot (l & 15) ^ (row & 15).
  const int nch = p.Kpad / kChunk;
  const int rot = (int)(blockIdx.x % (unsigned)nch);
  unsigned srcA[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int row = 4 * (wave + 8 * j) + (lane >> 4);
    srcA[j] = (unsigned)row * p.lda + (unsigned)(((lane & 15) ^ (row & 15)) * 16);
  }
  const int frow = 4 * (wave & 3) + (lane >> 4);  // feature row of this lane's B piece (waves 4-7: a second copy, unused)
  const unsigned srcB = (unsigned)(n0 + frow) * (unsigned)p.Kpad + (unsigned)(((lane & 15) ^ (frow & 15)) * 16);
  auto issue = [&](int step) {  // the chunk of walk position `step` -> stage step % kStages
    int ch = step + rot;
    while (ch >= nch) ch -= nch;
    const unsigned k = (unsigned)ch * kChunk;
    uint8_t* st = smem + (step & (kStages - 1)) * kStage;
#pragma unroll
    for (int j = 0; j < 4; ++j)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (__attribute__((address_space(3))) void*)(st + (wave + 8 * j) * 1024), 16,
                                               (int)(srcA[j] + k), 0, 0, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (__attribute__((address_space(3))) void*)(st + kStageA + wave * 1024), 16,
                                             (int)(srcB + k), 0, 0, 0);
  };

  v4i acc = {0, 0, 0, 0};
  // fragment addresses inside a stage: row 16 w + r (A) / feature r (B), slot (4 ks + q) ^ (r & 15)
  const int arow = (wave * 16 + lr) * kChunk, brow = kStageA + lr * kChunk;
  const int nsteps = nch;
  issue(0);
  issue(nsteps > 1 ? 1 : 0);
  issue(nsteps > 2 ? 2 : 0);
  for (int step = 0; step < nsteps; ++step) {
    asm volatile("s_waitcnt vmcnt(10)" ::: "memory");  // my pieces of this chunk (the two chunks after it may be in flight)
    asm volatile("s_barrier" ::: "memory");             // everyone's pieces; everyone is done with the previous chunk
    issue(step + 3);  // -> the stage of the previous chunk (past the end the walk wraps: keeps the vmcnt counts uniform)
    const uint8_t* st = smem + (step & (kStages - 1)) * kStage;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const int slot = ((4 * ks + lq) ^ lr) * 16;
      const v4i a = *reinterpret_cast<const v4i*>(st + arow + slot) ^ (int)0x80808080;  // u8 -> s8 (128 * wsum is in ocp)
      const v4i b = *reinterpret_cast<const v4i*>(st + brow + slot);
      acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(b, a, acc, 0, 0, 0);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (the padding pieces: nothing may land after the block is gone)

  // ---- epilogue of src/fully_connected.cc:42-48: lane (q, r) holds features 4 q .. 4 q + 3 of row 16 w + r
  const int row = row0 + wave * 16 + lr, n = n0 + 4 * lq;
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

// one row group only: at 129-256 rows every CU would read the activations twice over and the tiled split-K kernel wins
// (fc6 at 250 rows: 38 us here, 30 us tiled; at 125 rows: 21 vs 23 + its reduction launch); and enough features to
// give half the CUs a block (a block's time does not depend on N: it streams all of the activations).  `force`
// (kernel variant 80) lifts the feature threshold, for tests.
bool i8ie_flin_wants(int m, int n, int Kpad, bool force) {
  return m <= kRows && n >= (force ? 16 : 2048) && Kpad >= 1024 && Kpad % kChunk == 0;
}

int i8ie_flin_launch(i8ie_ctx* ctx, const I8ieIgemmCall& c) {
  I8IE_REQUIRE(c.amode == 0 && c.M > 0 && c.N > 0 && c.Kpad % kChunk == 0, "flin: shape");
  I8IE_REQUIRE((reinterpret_cast<uintptr_t>(c.A) & 15u) == 0 && (reinterpret_cast<uintptr_t>(c.B) & 15u) == 0 && c.lda % 16 == 0,
               "flin: operands must be 16-byte aligned");
  I8IE_REQUIRE((size_t)c.Npad * c.Kpad < ((size_t)1 << 32) - 4096 && (size_t)kRows * c.lda + c.Kpad < ((size_t)1 << 31), "flin: offsets exceed 32 bits");
  static bool raised[64] = {};
  const int dev = ctx->device & 63;
  constexpr int lds = kStages * kStage;  // 160 KB
  if (!raised[dev]) {
    I8IE_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&flin_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    raised[dev] = true;
  }
  FlinArgs a{};
  a.A = c.A; a.a_bytes = c.a_bytes; a.lda = (unsigned)c.lda; a.M = c.M;
  a.B = c.B; a.b_bytes = (unsigned)((size_t)c.Npad * c.Kpad); a.Kpad = c.Kpad; a.N = c.N;
  a.ocp = c.ocp; a.biasf = c.biasf;
  a.rq = i8ie_make_requant(c.s_in, c.s_w, c.s_out, c.zp_out);
  a.relu_lo = c.relu ? c.zp_out : 0;
  a.out = c.out; a.acc = c.acc;
  const dim3 grid((unsigned)((c.N + 15) / 16), (unsigned)((c.M + kRows - 1) / kRows));
  I8ieProfScope prof(ctx, "flin_128x16", 2.0 * c.M * c.N * c.Ktrue, (double)c.M * c.Ktrue + (double)c.N * c.Ktrue + (double)c.M * c.N);
  flin_kernel<<<grid, 512, lds, ctx->stream>>>(a);
  I8IE_LAUNCH_CHECK();
  return I8IE_OK;
}
