// i8ie_lgemm.hip -- Linear::forward_prop(Tensor<u8_t>&&) (src/fully_connected.cc:22-52) for MANY input rows
// (fc6 / fc7 of a 500-1000 image batch).
//
//   C[r][j] = sum_k A_u8[r][k] * W_s8[j][k] + oc[j];  C = (int)((float)C + (float)q_b[j] / s_in);  out = relu?(down_scale(C))
//
// Round 2 left these two layers at 0.22 of the int8 MFMA peak (0.099 ms of the 1.47 ms AlexNet step for 5 % of its MACs):
// 1000 x 4096 outputs are only 256 tiles of 128 x 128, one per CU, each with a 9216-byte K walk, and the tiled kernel
// stages BOTH operands through LDS K tile by K tile -- 32 KB per 512 MFMA cycles, register-staged, with two barriers
// per K tile and nothing else resident on the CU to hide them.  What round 2 learnt on the convolutions applies here:
//   * the WEIGHTS never touch LDS.  They are packed once per layer in MFMA fragment order [feature tile of 32][k-step of
//     32 bytes][lane][16 B], so a wave's fragment is 1 KiB contiguous, read by one coalesced buffer load straight from L2
//     into registers one K tile ahead.  Wave w of the block's four owns features 32 w .. 32 w + 31 for ALL of the block's
//     rows: every weight byte is loaded by exactly one wave of the block, 16 KB per K tile and CU through the vector
//     memory path (any split of the rows over waves would double that and hit the path's 64 B/clk).
//   * the ACTIVATIONS go through LDS: a K tile of 128 bytes x 128 (or 64) rows per stage, loaded into registers three
//     tiles ahead (1 KiB per wave-instruction: 8 rows x 128 bytes) and written, re-biased, one tile ahead with
//     ds_write_b128; 16-byte chunk c of row r sits at c ^ ((r >> 1) & 7) (the loading lane picks its source chunk), so the
//     ds_read_b128 fragments are conflict-free; every wave reads all of them (LDS has the bandwidth: 64 KB per 512
//     cycles and CU).  One barrier per K tile.  Weights and activations of a tile are requested in the order
//     [weights][activations], so that one counted s_waitcnt vmcnt at the top of a tile covers both.
//   * blocks that share a weight tile (the 8 row tiles of one feature tile) run on one XCD: its L2 serves the tile
//     once from HBM.
// INT32 accumulators are the reference's sums; acc_dbg dumps them before the float bias step, as everywhere else.
#include <cstdio>

#include "i8ie_calls.h"
#include "i8ie_internal.h"
#include "i8ie_requant.h"

namespace {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

struct LgArgs {
  const uint8_t* A;  // [M][lda] u8
  unsigned a_bytes, lda;
  int M;
  const int8_t* Bf;  // fragment order [N / 32][Kpad / 32][64][16]
  unsigned bf_bytes;
  int Kpad, N;
  const int32_t* ocp;  // oc + 128 * wsum (the activations enter re-biased by -128)
  const float* biasf;
  I8ieRequant rq;
  int relu_lo;
  uint8_t* out;  // [M][N]
  int32_t* acc;  // null, or [M][N]: C before the float bias step
  int tiles_m, tiles_n;
};

constexpr int kStages = 4;

extern __shared__ __attribute__((aligned(16))) uint8_t lg_smem[];

// TM: 32-row MFMA tiles per block (4: 128 rows, 2: 64 rows)
template <int TM>
__global__ __launch_bounds__(256) void lgemm_kernel(LgArgs p) {
  uint8_t* const smem = lg_smem;
  constexpr int ROWS = TM * 32;
  constexpr int STAGE = ROWS * 128;      // bytes of a stage: ROWS rows of 128 bytes
  constexpr int PIECES = STAGE / 1024;   // 1 KiB DMA pieces per stage: 8 rows each
  constexpr int PPW = PIECES / 4;        // pieces per wave
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hh = lane >> 5, l31 = lane & 31;

  // block -> (row tile, feature tile): the row tiles of one feature tile on one XCD (blocks b, b + 8, ... share one)
  const int nwg = (int)gridDim.x, bid = (int)blockIdx.x;
  const int q8 = nwg >> 3, r8 = nwg & 7, xcd = bid & 7;
  const int t = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
  const int tile_n = t / p.tiles_m, tile_m = t - tile_n * p.tiles_m;
  const int row0 = tile_m * ROWS, n0 = tile_n * 128;

  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(p.A), 0, p.a_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<int8_t*>(p.Bf), 0, p.bf_bytes, 0x00020000);

  // ---- DMA pieces of a K tile: piece i = rows 8 i .. 8 i + 7; wave w issues pieces w, w + 4, ...  Lane l of a piece:
  //      row 8 i + (l >> 3), LDS chunk l & 7, source chunk (l & 7) ^ ((row >> 1) & 7)
  unsigned srcA[PPW];
#pragma unroll
  for (int j = 0; j < PPW; ++j) {
    const int lrow = 8 * (wave + 4 * j) + (lane >> 3);
    int grow = row0 + lrow;
    grow = grow < p.M ? grow : p.M - 1;  // rows past M: computed, never stored
    srcA[j] = (unsigned)grow * p.lda + (unsigned)(((lane & 7) ^ ((lrow >> 1) & 7)) * 16);
  }
  const int nkt = p.Kpad / 128;
  // activations of K tile kt -> registers (PPW loads of 1 KiB: 8 rows x 128 bytes each), and from there, re-biased, into
  // stage kt % 4 one tile before they are read.  (LDS-DMA was measured first: 16 pieces per tile and CU cost the four
  // issuing waves more than the tile's 512 MFMA cycles -- 1.7 k cycles per tile; a plain load + ds_write_b128 pair is cheap.)
  auto load_A = [&](v4i (&dst)[PPW], int kt) {
    const unsigned k = (unsigned)(kt < nkt ? kt : 0) * 128u;  // (past the end: tile 0 again, so that the vmcnt counts stay uniform)
#pragma unroll
    for (int j = 0; j < PPW; ++j) dst[j] = __builtin_bit_cast(v4i, __builtin_amdgcn_raw_buffer_load_b128(rsA, (int)(srcA[j] + k), 0, 0));
  };
  auto store_A = [&](const v4i (&src)[PPW], int kt) {
    uint8_t* st = smem + (kt & (kStages - 1)) * STAGE;
#pragma unroll
    for (int j = 0; j < PPW; ++j) *reinterpret_cast<v4i*>(st + (wave + 4 * j) * 1024 + lane * 16) = src[j] ^ (int)0x80808080;  // u8 -> s8
  };
  // weights of K tile kt: four fragments (k-steps) of this wave's 32 features
  const unsigned bbase = (unsigned)((n0 >> 5) + wave) * (unsigned)(p.Kpad / 32) * 1024u + (unsigned)lane * 16u;
  auto load_B = [&](v4i (&dst)[4], int kt) {
    const unsigned o = bbase + (unsigned)(kt < nkt ? kt : 0) * 4096u;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) dst[ks] = __builtin_bit_cast(v4i, __builtin_amdgcn_raw_buffer_load_b128(rsB, (int)(o + ks * 1024), 0, 0));
  };

  v16i acc[TM];
#pragma unroll
  for (int mi = 0; mi < TM; ++mi)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[mi][r] = 0;
  // fragment of k-step ks for row r of a stage: chunk (2 ks + hh) ^ ((r >> 1) & 7).  (Rows are 128 bytes: even rows start at
  // bank 0, odd rows at bank 32; the 16 lanes one ds_read_b128 cycle serves are 8 even and 8 odd rows whose (r >> 1) & 7
  // are all different -- with r & 7 as the key 40 % of the LDS cycles were bank conflicts)
  int ardk[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) ardk[ks] = l31 * 128 + (((2 * ks + hh) ^ ((l31 >> 1) & 7)) * 16);

  // Weights and activations of a tile are requested THREE tiles ahead (an L2 round trip under load is longer than a tile's
  // 512 MFMA cycles).  Program order per tile t: [weights of t + 3][activations of t + 3]; vector-memory operations retire
  // in issue order, so "all but the youngest 4 + PPW" at the top of tile t means: everything requested up to tile t - 2 has
  // arrived -- the weights of t and the activations of t + 1, which go to LDS now and are published by this tile's barrier.
  v4i Bq[4][4], Aq[4][PPW];
  load_B(Bq[0], 0);
  load_A(Aq[0], 0);
  load_B(Bq[1], 1);
  load_A(Aq[1], 1);
  load_B(Bq[2], 2);
  load_A(Aq[2], 2);
  if (PPW == 4) asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
  store_A(Aq[0], 0);
  // (tile 0 is published by the first barrier below; the fragment reads run one k-step ahead of the MFMAs from then on,
  // across tile boundaries: tile k + 1 is visible from the barrier at the top of tile k.  One wave per SIMD: nothing else
  // would cover an LDS round trip per k-step -- 1.7-2 k cycles per tile without this)
  v4i af[2][TM];
  auto read_A = [&](v4i (&dst)[TM], int kt, int ks) {
    const uint8_t* st = smem + (kt & (kStages - 1)) * STAGE;
#pragma unroll
    for (int mi = 0; mi < TM; ++mi) dst[mi] = *reinterpret_cast<const v4i*>(st + mi * 4096 + ardk[ks]);
  };
#pragma clang loop unroll(disable)
  for (int kt = 0; kt < nkt; kt += 4) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int k = kt + q;
      if (k < nkt) {  // (wave-uniform)
        if (PPW == 4) asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        store_A(Aq[(q + 1) & 3], k + 1);  // (its stage was last read in tile k - 3)
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // publishes tile k + 1; tile k was published a barrier ago
        load_B(Bq[(q + 3) & 3], k + 3);
        load_A(Aq[(q + 3) & 3], k + 3);
        if (k == 0) read_A(af[0], 0, 0);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          if (ks < 3) read_A(af[(ks + 1) & 1], k, ks + 1);
          else read_A(af[0], k + 1, 0);  // (past the last tile: a stage that exists, values unused)
#pragma unroll
          for (int mi = 0; mi < TM; ++mi) acc[mi] = __builtin_amdgcn_mfma_i32_32x32x32_i8(Bq[q][ks], af[ks & 1][mi], acc[mi], 0, 0, 0);
        }
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (the padding requests: nothing may land after the block is gone)
  __syncthreads();

  // ---- epilogue of src/fully_connected.cc:42-48 through an LDS tile [ROWS][128 + 4] -> 16-byte row stores.
  //      lane & 31 = row of the MFMA tile, feature = 32 wave + (reg & 3) + 8 (reg >> 2) + 4 hh
  constexpr int SROW = 128 + 4;
  const I8ieRequant rq = p.rq;
  const int lo = p.relu_lo;
  const float lof = (float)lo;
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const int lcol0 = wave * 32 + 8 * g + 4 * hh, gcol0 = n0 + lcol0;
    const bool cin = gcol0 < p.N;  // (N % 4 == 0: checked on the host)
    const int4 ocv = cin ? *reinterpret_cast<const int4*>(p.ocp + gcol0) : make_int4(0, 0, 0, 0);
    const float4 bfv = (cin && p.biasf != nullptr) ? *reinterpret_cast<const float4*>(p.biasf + gcol0) : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
#pragma unroll
    for (int mi = 0; mi < TM; ++mi) {
      const int lrow = mi * 32 + l31, grow = row0 + lrow;
      int cv[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        int c = acc[mi][g * 4 + r] + (r == 0 ? ocv.x : r == 1 ? ocv.y : r == 2 ? ocv.z : ocv.w);
        if (p.acc != nullptr && grow < p.M && cin) p.acc[(size_t)grow * p.N + gcol0 + r] = c;
        if (p.biasf != nullptr) c = (int)((float)c + (r == 0 ? bfv.x : r == 1 ? bfv.y : r == 2 ? bfv.z : bfv.w));  // src/fully_connected.cc:44
        cv[r] = c;
      }
      *reinterpret_cast<uint32_t*>(smem + lrow * SROW + lcol0) = i8ie_requant_pack4(cv, rq, lo, lof);
    }
  }
  __syncthreads();
  for (int idx = tid; idx < ROWS * 8; idx += 256) {
    const int lrow = idx >> 3, ch = idx & 7;
    const int grow = row0 + lrow, gcol = n0 + ch * 16;
    if (grow < p.M && gcol < p.N) {  // (N % 16 == 0: checked on the host)
      const uint32_t* s = reinterpret_cast<const uint32_t*>(smem + lrow * SROW + ch * 16);
      *reinterpret_cast<uint4*>(p.out + (size_t)grow * p.N + gcol) = make_uint4(s[0], s[1], s[2], s[3]);
    }
  }
}

// weights [Npad][Kpad] (K contiguous) -> fragment order [N / 32][Kpad / 32][64][16]
__global__ __launch_bounds__(256) void lgemm_pack_kernel(const int8_t* __restrict__ B, int8_t* __restrict__ Bf, int64_t total16, int Kpad) {
  const int ksteps = Kpad / 32;
  const int64_t gstride = (int64_t)gridDim.x * 256;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total16; e += gstride) {
    const int lane = (int)(e & 63);
    const int64_t t = e >> 6;
    const int ks = (int)(t % ksteps);
    const int nt = (int)(t / ksteps);
    reinterpret_cast<uint4*>(Bf)[e] =
        *reinterpret_cast<const uint4*>(B + (size_t)(nt * 32 + (lane & 31)) * Kpad + (size_t)ks * 32 + (lane >> 5) * 16);
  }
}

template <int TM>
int launch_lg(i8ie_ctx* ctx, const LgArgs& a, int grid) {
  static bool raised[64] = {};
  const int dev = ctx->device & 63;
  constexpr int lds = kStages * TM * 32 * 128;
  if (!raised[dev]) {
    I8IE_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&lgemm_kernel<TM>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    raised[dev] = true;
  }
  lgemm_kernel<TM><<<grid, 256, lds, ctx->stream>>>(a);
  I8IE_LAUNCH_CHECK();
  return I8IE_OK;
}

}  // namespace

// rows above the few-row kernel's range, shapes the fragment order covers, 16-byte row stores
bool i8ie_lgemm_wants(int m, int n, int K, int Kpad) {
  // ... and at least half the CUs' worth of 64 x 128 tiles (fewer: the tiled kernel with split-K fills the chip better)
  return m > 256 && n % 128 == 0 && K % 16 == 0 && Kpad % 128 == 0 && Kpad >= 512 && (long)((m + 63) / 64) * (n / 128) >= 128;
}

int i8ie_lgemm_launch(i8ie_ctx* ctx, const I8ieIgemmCall& c, bool perm_panel) {
  I8IE_REQUIRE(c.wcache != nullptr, "lgemm: no weight cache");
  I8IE_REQUIRE((reinterpret_cast<uintptr_t>(c.A) & 15u) == 0 && (reinterpret_cast<uintptr_t>(c.out) & 15u) == 0, "lgemm: operands must be 16-byte aligned");
  I8IE_REQUIRE(c.a_bytes < ((size_t)1 << 32) - 4096, "lgemm: activations beyond the 32-bit offset range");
  // fragment-ordered weights: once per layer and panel (reference K order / the (h, w, c) order of an NHWC-flattened input)
  const size_t bf_bytes = (size_t)c.N * c.Kpad;
  I8IE_REQUIRE(bf_bytes < ((size_t)1 << 32) - 4096, "lgemm: weights beyond the 32-bit offset range");
  const unsigned long long wkey = (3ull << 32) | (perm_panel ? 1ull : 0ull);
  void* wbuf = c.wcache->find(wkey);
  if (wbuf == nullptr) {
    I8IE_REQUIRE(ctx->capture == nullptr, "weight re-packing inside a graph capture: run the same calls once eagerly first");
    I8IE_TRY(i8ie_malloc(ctx, bf_bytes, &wbuf));
    const int64_t total16 = (int64_t)(bf_bytes / 16);
    int64_t blocks = (total16 + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    lgemm_pack_kernel<<<(int)blocks, 256, 0, ctx->stream>>>(c.B, (int8_t*)wbuf, total16, c.Kpad);
    if (hipGetLastError() != hipSuccess) {
      i8ie_free(ctx, wbuf);
      return I8IE_ERR_HIP;
    }
    c.wcache->ents.push_back(I8ieWCache::Ent{wkey, wbuf});
  }
  LgArgs a{};
  a.A = c.A; a.a_bytes = (unsigned)c.a_bytes; a.lda = (unsigned)c.lda; a.M = c.M;
  a.Bf = (const int8_t*)wbuf; a.bf_bytes = (unsigned)bf_bytes; a.Kpad = c.Kpad; a.N = c.N;
  a.ocp = c.ocp; a.biasf = c.biasf;
  a.rq = i8ie_make_requant(c.s_in, c.s_w, c.s_out, c.zp_out);
  a.relu_lo = c.relu ? c.zp_out : 0;
  a.out = c.out; a.acc = c.acc;
  a.tiles_n = c.N / 128;
  // 64-row tiles, two blocks per CU: measured 57 + 30 us for fc6 + fc7 at 1000 rows against 67 + 35 us with 128-row tiles
  // (one block of four waves per CU: a wave waiting at the tile's barrier or for its loads leaves its SIMD idle) and
  // 65 + 34 us for the tiled kernel.  128-row tiles (half the weight traffic per CU) only beyond two blocks per CU of them.
  const bool big = (long)((c.M + 127) / 128) * a.tiles_n >= 1024;
  a.tiles_m = big ? (c.M + 127) / 128 : (c.M + 63) / 64;
  const double ops = 2.0 * c.M * c.N * c.Ktrue;
  const double bytes = (double)c.M * c.Ktrue + (double)c.N * c.Ktrue + (double)c.M * c.N;
  char tag[64];
  snprintf(tag, sizeof(tag), "lgemm_%dx128|M%d,N%d,K%d", big ? 128 : 64, c.M, c.N, c.Kchunks * 16);
  I8ieProfScope prof(ctx, ctx->prof ? tag : (big ? "lgemm_128x128" : "lgemm_64x128"), ops, bytes);
  return big ? launch_lg<4>(ctx, a, a.tiles_m * a.tiles_n) : launch_lg<2>(ctx, a, a.tiles_m * a.tiles_n);
}
