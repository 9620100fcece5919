// i8ie_igemm.hip -- implicit-GEMM Conv2d over NHWC u8 activations (no materialised
// im2col) and the same MFMA core for Linear.
//
//   C[r][j] = sum_k A_u8[r][k] * W_s8[j][k] + oc[j]        exact int32, = the
//   cblas_gemm_s8u8s32 result of src/conv2d.cc:131-133 / src/fully_connected.cc:39-41
//   (K is walked in (kh, kw, c) order instead of (c, kh, kw); integer sums commute)
//
// Structure (gfx950):
//   * A operand gathered straight from the NHWC activation tensor, which is PHYSICALLY
//     padded with the input zero point (src/conv2d.cc:24-28) by its producer, so the
//     gather has no bounds predicates: one 32-bit offset add + buffer_load_dwordx4 per
//     16-byte K chunk (16 channels of one tap).  Rows past M and the K tail read
//     whatever is there (or 0 past the end of the buffer): their B columns are zero /
//     their outputs are never stored.
//   * BK = 128 bytes per stage.  Convolutions (the default, VAR 5): ONE LDS stage filled by LDS-DMA
//     (buffer_load ... lds, 1 KiB per wave-instruction) into unpadded 128-byte rows whose 16-byte chunk c of row r
//     sits at c ^ (r & 7) (conflict-free ds_read_b128 fragments); fill -> vmcnt(0) + barrier -> compute ->
//     barrier, three to four blocks per CU hide each other's fills.  Linear: register staging into 144-byte rows.
//   * u8 -> s8 re-bias (x ^ 0x80) on the A fragments after the LDS read (DMA) or on the way into LDS (register
//     staging); 128 * sum_k W joins oc[j] (ocp), which is added to the zero-initialised accumulators in the
//     epilogue (integer adds commute exactly), so acc + ocp is the reference's C bit for bit.
//   * MFMA operands swapped (weights are the row operand): the lane holds one
//     activation row and 4 consecutive registers hold 4 consecutive output features,
//     so the row-major (= NHWC) epilogue packs 4 u8 per lane.
//   * Requantiser fast path, provably identical to ((float)C*s_in)*s_w/s_out + zp with
//     an exact-sequence fallback near rounding boundaries; optional fused ReLU
//     (src/functional.cc:15-26); optional physically padded output.
#include <cstdio>
#include <cstdlib>
#include <type_traits>

#include "i8ie_internal.h"
#include "i8ie_calls.h"
#include "i8ie_requant.h"

namespace {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

constexpr int BK2 = 128;   // bytes of K per stage: 8 chunks of 16 B = 4 MFMA k-steps
constexpr int LROW = 144;  // LDS row stride

using Requant = I8ieRequant;  // i8ie_requant.h: the exact sequence, the guarded estimate and its proof
__device__ __forceinline__ int requant_exact(float cf, const Requant& q, int lo) { return i8ie_requant_exact(cf, q, lo); }
__device__ __forceinline__ uint32_t requant_pack4(const int (&c)[4], const Requant& q, int lo, float lof) {
  return i8ie_requant_pack4(c, q, lo, lof);
}

struct IgemmArgs {
  const uint8_t* A;
  unsigned a_bytes;  // readable bytes from A (buffer bound)
  unsigned lda;      // AMODE 0: row pitch in bytes
  int M;
  // AMODE 1: physically padded NHWC input [n][Hp][Wp][C]; A already points at the window
  // origin of output pixel (0, 0)
  unsigned img_pitch;  // Hp * Wp * C
  unsigned row_pitch;  // Wp * C
  unsigned C;          // channels (bytes per pixel), % 16 == 0
  int sh, sw, OH, OW;
  int RC;              // chunks per kernel row: KW * C / 16
  unsigned row_jump;   // row_pitch - RC * 16
  // B
  const int8_t* B;  // [Npad][Kpad] zero padded, K order matches the A walk
  unsigned b_bytes;
  int Kpad;
  int N;
  const int32_t* ocp;  // [N] oc[j] + 128 * wsum[j]
  const float* biasf;  // [N] (float)q_b[j] / s_in (Linear, src/fully_connected.cc:44)
  Requant rq;
  int relu_lo;  // zp_out when relu is fused, else 0
  int vec_store;
  uint8_t* out;  // [M][N], or bordered NHWC when ob > 0
  int ob, OHp, OWp;
  int32_t* acc;  // [M][N]
  // split-K (Linear with few output tiles): blockIdx.y = K slice; slices write INT32 partial slabs
  // [slice][M][N] to `partial` and skip the epilogue; splitk_reduce_kernel finishes
  int ksplit, tiles_per_slice;
  int32_t* partial;
};

// VAR (tuning variants, identical results): 0 = two LDS stages, loads one K tile ahead;
// 1 = loads two K tiles ahead (second register set); 2 = VAR 0 + s_setprio around the MFMAs;
// 3 = one LDS stage, two barriers per K tile (half the LDS: more blocks per CU);
// 5 = VAR 3 staged by LDS-DMA (buffer_load ... lds, 1 KiB per wave-instruction, no VGPR pass, no ds_write):
//     unpadded 128-B LDS rows, 16-B chunk c of row r stored at chunk c ^ (r & 7) (conflict-free
//     ds_read_b128 fragments), the u8 -> s8 re-bias applied to the A fragments after the read;
// 7 = the DMA form with two 64 KiB stages in dynamic LDS (one block per CU): the fill of K tile k+1 is in
//     flight under the MFMAs of K tile k, one barrier per K tile.  Meant for 256 x 256 / 256 x 192 block
//     tiles (4 waves of 128 x 128 / 128 x 96): twice the MFMA work per staged byte, which is what the
//     fill latency x LDS capacity product allows to keep the matrix cores busy
extern __shared__ __attribute__((aligned(16))) uint8_t i8ie_dyn_smem[];

template <int AMODE, int WM, int WN, int TM, int TN, bool BIAS, bool ACC, int VAR>
__global__ __launch_bounds__(WM* WN * 64) void igemm_u8s8_kernel(IgemmArgs p, int tiles_m, int tiles_n,
                                                                 int m_fastest) {
  constexpr int BM = WM * TM * 32, BN = WN * TN * 32, NT = WM * WN * 64;
  constexpr bool DMA2 = VAR == 7;
  constexpr bool DMA = VAR == 5 || DMA2;
  constexpr int NST = (VAR == 3 || VAR == 5) ? 1 : 2;
  constexpr int A_PER = BM * 8 / NT, B_PER = BN * 8 / NT;
  static_assert(BM * 8 % NT == 0 && BN * 8 % NT == 0, "staging map");
  static_assert(!DMA || NT % 64 == 0, "DMA rows per pass are a multiple of 8");
  constexpr int LR = DMA ? BK2 : LROW;  // LDS row pitch
  constexpr int STAGE = DMA2 ? 65536 : (BM + BN) * LR;  // DMA2: power-of-two pitch, stages toggle by XOR
  static_assert((BM + BN) * LR <= STAGE, "stage holds the A and B tiles");
  constexpr int SROW = BN + 4;  // epilogue tile row stride: odd dword count -> conflict-free ds_write_b32
  static_assert(BM * SROW <= NST * STAGE, "epilogue tile fits");
  __shared__ __attribute__((aligned(16))) uint8_t smem_static[DMA2 ? 16 : NST * STAGE];
  uint8_t* const smem = DMA2 ? i8ie_dyn_smem : smem_static;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;

  // XCD-aware, bijective block -> tile map (blocks with equal blockIdx % 8 share an L2)
  const int nwg = gridDim.x, bid = blockIdx.x;
  const int q8 = nwg >> 3, r8 = nwg & 7, xcd = bid & 7;
  const int t = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
  int tile_m, tile_n;
  if (m_fastest) {
    tile_m = t % tiles_m;
    tile_n = t / tiles_m;
  } else {
    tile_n = t % tiles_n;
    tile_m = t / tiles_n;
  }
  const int m0 = tile_m * BM, n0 = tile_n * BN;

  // ---- accumulators: D = W_tile x A_tile^T; lane & 31 -> activation row,
  //      feature = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
  // start at zero; oc'[j] = oc[j] + 128 * wsum[j] joins in the epilogue (integer adds commute exactly), which
  // keeps global loads and their latency out of the prologue
  v16i acc[TM][TN];
#pragma unroll
  for (int mi = 0; mi < TM; ++mi)
#pragma unroll
    for (int ni = 0; ni < TN; ++ni)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0;

  // ---- staging state ------------------------------------------------------------------
  const __amdgpu_buffer_rsrc_t rsA =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(p.A), 0, p.a_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsB =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<int8_t*>(p.B), 0, p.b_bytes, 0x00020000);
  // K chunk (16 B) this thread stages for each of its rows; with DMA the lane's LDS slot is fixed
  // (slot = lane), so the swizzle picks the source chunk instead: c = slot ^ (row & 7)
  const int cA = DMA ? ((tid & 7) ^ ((tid >> 3) & 7)) : (tid & 7);
  unsigned a_off[A_PER], b_off[B_PER];
  // AMODE 1: physical pixel index of this thread's row i in a bordered output (the store loop of the epilogue
  // walks the same rows).  One division for the first row, then a walk of NT/8 pixels per row.
  unsigned o_pix[(AMODE == 1) ? A_PER : 1];
  if (AMODE == 0) {
#pragma unroll
    for (int i = 0; i < A_PER; ++i) {
      int gr = m0 + (tid >> 3) + (NT >> 3) * i;
      gr = gr < p.M ? gr : p.M - 1;  // rows past M: computed, never stored
      a_off[i] = (unsigned)gr * p.lda;
    }
  } else {
    const int P = p.OH * p.OW;
    const int gr0 = m0 + (tid >> 3);
    int img = gr0 / P;
    const int rem = gr0 - img * P;
    int oh = rem / p.OW, ow = rem - oh * p.OW;
#pragma unroll
    for (int i = 0; i < A_PER; ++i) {
      const bool past = gr0 + (NT >> 3) * i >= p.M;  // rows past M read the last pixel: computed, never stored
      const int im = past ? p.M / P - 1 : img, y = past ? p.OH - 1 : oh, x = past ? p.OW - 1 : ow;
      a_off[i] = (unsigned)im * p.img_pitch + (unsigned)(y * p.sh) * p.row_pitch + (unsigned)(x * p.sw) * p.C;
      o_pix[i] = ((unsigned)im * p.OHp + y + p.ob) * p.OWp + x + p.ob;
      ow += NT >> 3;
      while (ow >= p.OW) { ow -= p.OW; ++oh; }
      while (oh >= p.OH) { oh -= p.OH; ++img; }
    }
  }
#pragma unroll
  for (int i = 0; i < B_PER; ++i) {
    const int idx = tid + i * NT;
    b_off[i] = (unsigned)(n0 + (idx >> 3)) * (unsigned)p.Kpad + (DMA ? cA : (idx & 7)) * 16;
  }
  // this thread's K chunk q = cA, cA + 8, ...: byte offset inside the window
  unsigned koff;
  int f = 0;
  if (AMODE == 0) {
    koff = cA * 16;
  } else {
    const int kh = cA / p.RC;
    f = cA - kh * p.RC;
    koff = (unsigned)kh * p.row_pitch + (unsigned)f * 16;
  }

  v4i ra[A_PER], rb[B_PER];
  v4i ra2[VAR == 1 ? A_PER : 1], rb2[VAR == 1 ? B_PER : 1];  // second register set (VAR 1)
  int kbase_l = 0;  // set below once the K slice of this block is known
  auto load_into = [&](int k0, v4i* da, v4i* db) {
#pragma unroll
    for (int i = 0; i < A_PER; ++i)
      da[i] = __builtin_bit_cast(v4i, __builtin_amdgcn_raw_buffer_load_b128(rsA, a_off[i] + koff, 0, 0));
#pragma unroll
    for (int i = 0; i < B_PER; ++i)
      db[i] = __builtin_bit_cast(v4i, __builtin_amdgcn_raw_buffer_load_b128(rsB, b_off[i], k0 + kbase_l, 0));
    koff += BK2;
    if (AMODE == 1) {
      f += 8;
      while (f >= p.RC) {  // next kernel row(s)
        f -= p.RC;
        koff += p.row_jump;
      }
    }
  };
  auto load_tile = [&](int k0) { load_into(k0, ra, rb); };
  // LDS-DMA: wave-instruction i of wave w fills LDS granules [i * NT + 64 w, + 64) = 8 rows x 128 B
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  // quarter q of the fill of one K tile (the fill is spread over the four k-steps of the tile before)
  auto dma_part = [&](int k0, int stage_off, int q) {
#pragma unroll
    for (int i = 0; i < A_PER; ++i)
      if (i * 4 / A_PER == q || (A_PER < 4 && q == 0))
        __builtin_amdgcn_raw_ptr_buffer_load_lds(
            rsA, (__attribute__((address_space(3))) void*)(smem + stage_off + (i * NT + wave_u * 64) * 16), 16,
            (int)(a_off[i] + koff), 0, 0, 0);
#pragma unroll
    for (int i = 0; i < B_PER; ++i)
      if (i * 4 / B_PER == q || (B_PER < 4 && q == 0))
        __builtin_amdgcn_raw_ptr_buffer_load_lds(
            rsB, (__attribute__((address_space(3))) void*)(smem + stage_off + BM * LR + (i * NT + wave_u * 64) * 16),
            16, (int)b_off[i], k0 + kbase_l, 0, 0);
  };
  auto dma_advance = [&]() {
    koff += BK2;
    if (AMODE == 1) {
      f += 8;
      while (f >= p.RC) {
        f -= p.RC;
        koff += p.row_jump;
      }
    }
  };
  auto dma_tile = [&](int k0, int stage_off) {
#pragma unroll
    for (int q = 0; q < 4; ++q) dma_part(k0, stage_off, q);
    dma_advance();
  };
  // LDS addresses (bytes from smem); stage s adds s * STAGE as an immediate
  int a_wr[A_PER], b_wr[B_PER], a_rd[TM], b_rd[TN];
#pragma unroll
  for (int i = 0; i < A_PER; ++i) a_wr[i] = ((tid >> 3) + (NT >> 3) * i) * LROW + cA * 16;
#pragma unroll
  for (int i = 0; i < B_PER; ++i) {
    const int idx = tid + i * NT;
    b_wr[i] = BM * LROW + (idx >> 3) * LROW + (idx & 7) * 16;
  }
#pragma unroll
  for (int mi = 0; mi < TM; ++mi) a_rd[mi] = ((wm * TM + mi) * 32 + (lane & 31)) * LROW + (lane >> 5) * 16;
#pragma unroll
  for (int ni = 0; ni < TN; ++ni)
    b_rd[ni] = BM * LROW + ((wn * TN + ni) * 32 + (lane & 31)) * LROW + (lane >> 5) * 16;

  auto store_from = [&](auto sc, const v4i* sa, const v4i* sb) {
    constexpr int S = decltype(sc)::value;
#pragma unroll
    for (int i = 0; i < A_PER; ++i)
      *reinterpret_cast<v4i*>(smem + S * STAGE + a_wr[i]) = sa[i] ^ (int)0x80808080;  // u8 -> s8
#pragma unroll
    for (int i = 0; i < B_PER; ++i) *reinterpret_cast<v4i*>(smem + S * STAGE + b_wr[i]) = sb[i];
  };
  auto store_tile = [&](auto sc) { store_from(sc, ra, rb); };
  // DMA image: fragment of k-step ks for row r sits at chunk (2 ks + (lane >> 5)) ^ (r & 7); r & 7 == lane & 7
  int a_rdk[DMA ? 4 : 1], b_rdk[DMA ? 4 : 1];
  if (DMA) {
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const int sw = ((ks * 2 + (lane >> 5)) ^ (lane & 7)) * 16;
      a_rdk[ks] = (wm * TM * 32 + (lane & 31)) * LR + sw;
      b_rdk[ks] = BM * LR + (wn * TN * 32 + (lane & 31)) * LR + sw;
    }
  }
  auto compute_dma = [&](bool fill_next, int k0, int stage_off) {
    v4i af[2][TM], bf[2][TN];
#pragma unroll
    for (int mi = 0; mi < TM; ++mi) af[0][mi] = *reinterpret_cast<const v4i*>(smem + a_rdk[0] + mi * 32 * LR);
#pragma unroll
    for (int ni = 0; ni < TN; ++ni) bf[0][ni] = *reinterpret_cast<const v4i*>(smem + b_rdk[0] + ni * 32 * LR);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      if (ks < 3) {
#pragma unroll
        for (int mi = 0; mi < TM; ++mi)
          af[(ks + 1) & 1][mi] = *reinterpret_cast<const v4i*>(smem + a_rdk[ks + 1] + mi * 32 * LR);
#pragma unroll
        for (int ni = 0; ni < TN; ++ni)
          bf[(ks + 1) & 1][ni] = *reinterpret_cast<const v4i*>(smem + b_rdk[ks + 1] + ni * 32 * LR);
      }
#pragma unroll
      for (int mi = 0; mi < TM; ++mi) {
        const v4i as8 = af[ks & 1][mi] ^ (int)0x80808080;  // u8 -> s8
#pragma unroll
        for (int ni = 0; ni < TN; ++ni)
          acc[mi][ni] = __builtin_amdgcn_mfma_i32_32x32x32_i8(bf[ks & 1][ni], as8, acc[mi][ni], 0, 0, 0);
      }
      if (DMA2 && fill_next) dma_part(k0, stage_off, ks);  // a quarter of the next tile's fill per k-step
    }
    if (DMA2 && fill_next) dma_advance();
  };
  auto compute = [&](auto sc) {
    constexpr int S = decltype(sc)::value;
    v4i af[2][TM], bf[2][TN];  // fragments of k-step ks+1 are fetched under the MFMAs of k-step ks
#pragma unroll
    for (int mi = 0; mi < TM; ++mi) af[0][mi] = *reinterpret_cast<const v4i*>(smem + S * STAGE + a_rd[mi]);
#pragma unroll
    for (int ni = 0; ni < TN; ++ni) bf[0][ni] = *reinterpret_cast<const v4i*>(smem + S * STAGE + b_rd[ni]);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      if (ks < 3) {
#pragma unroll
        for (int mi = 0; mi < TM; ++mi)
          af[(ks + 1) & 1][mi] = *reinterpret_cast<const v4i*>(smem + S * STAGE + (ks + 1) * 32 + a_rd[mi]);
#pragma unroll
        for (int ni = 0; ni < TN; ++ni)
          bf[(ks + 1) & 1][ni] = *reinterpret_cast<const v4i*>(smem + S * STAGE + (ks + 1) * 32 + b_rd[ni]);
      }
      if (VAR == 2) __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int mi = 0; mi < TM; ++mi)
#pragma unroll
        for (int ni = 0; ni < TN; ++ni)
          acc[mi][ni] = __builtin_amdgcn_mfma_i32_32x32x32_i8(bf[ks & 1][ni], af[ks & 1][mi], acc[mi][ni], 0, 0, 0);
      if (VAR == 2) __builtin_amdgcn_s_setprio(0);
    }
  };
  using S0 = std::integral_constant<int, 0>;
  using S1 = std::integral_constant<int, 1>;

  int nk = p.Kpad / BK2;
  int kbase = 0;  // first K byte of this block's slice
  if (AMODE == 0 && p.ksplit > 1) {
    const int t0 = blockIdx.y * p.tiles_per_slice;
    const int t1 = t0 + p.tiles_per_slice < nk ? t0 + p.tiles_per_slice : nk;
    kbase = t0 * BK2;
    nk = t1 - t0;  // >= 1 by construction
    koff += kbase;
    kbase_l = kbase;
  }
  if constexpr (DMA2) {  // two stages: fill of tile k+1 in flight under the MFMAs of tile k, one barrier per tile
    dma_tile(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int fill = STAGE;  // byte offset of the stage the next fill goes to
    for (int kt = 0; kt < nk; ++kt) {
      compute_dma(kt + 1 < nk, (kt + 1) * BK2, fill);
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {  // fragments of the next tile come from the other stage
        a_rdk[ks] ^= STAGE;
        b_rdk[ks] ^= STAGE;
      }
      fill ^= STAGE;
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
    }
  } else if constexpr (DMA) {  // one stage: DMA fill | wait + barrier | compute | barrier
    dma_tile(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
      compute_dma(false, 0, 0);
      __syncthreads();
      if (kt + 1 < nk) {
        dma_tile((kt + 1) * BK2, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
      }
    }
  } else if constexpr (VAR == 3) {  // one stage: compute | barrier | refill | barrier
    load_tile(0);
    store_tile(S0{});
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
      if (kt + 1 < nk) load_tile((kt + 1) * BK2);
      compute(S0{});
      __syncthreads();
      if (kt + 1 < nk) {
        store_tile(S0{});
        __syncthreads();
      }
    }
  } else if constexpr (VAR == 1) {  // loads run two K tiles ahead of the MFMAs
    load_into(0, ra, rb);
    store_from(S0{}, ra, rb);
    if (nk > 1) load_into(BK2, ra2, rb2);
    __syncthreads();
    for (int kt = 0; kt < nk; kt += 2) {
      if (kt + 2 < nk) load_into((kt + 2) * BK2, ra, rb);
      compute(S0{});
      if (kt + 1 < nk) store_from(S1{}, ra2, rb2);
      __syncthreads();
      if (kt + 1 >= nk) break;
      if (kt + 3 < nk) load_into((kt + 3) * BK2, ra2, rb2);
      compute(S1{});
      if (kt + 2 < nk) store_from(S0{}, ra, rb);
      __syncthreads();
    }
  } else {
    load_tile(0);
    store_tile(S0{});
    __syncthreads();
    for (int kt = 0; kt < nk; kt += 2) {
      if (kt + 1 < nk) load_tile((kt + 1) * BK2);  // global loads in flight under the MFMAs
      compute(S0{});
      if (kt + 1 < nk) store_tile(S1{});
      __syncthreads();
      if (kt + 1 >= nk) break;
      if (kt + 2 < nk) load_tile((kt + 2) * BK2);
      compute(S1{});
      if (kt + 2 < nk) store_tile(S0{});
      __syncthreads();
    }
  }

  if (AMODE == 0 && p.ksplit > 1) {  // split-K: raw INT32 partial sums, finished by splitk_reduce_kernel
    // The tile leaves through LDS, a pass of PR rows at a time, so that the slab is written in row segments of
    // BN * 4 bytes (one dword per lane and row was ~64 scattered line requests per wave-instruction).
    int32_t* slab = p.partial + (size_t)blockIdx.y * p.M * p.N;
    constexpr int CTP = BN + 4;  // ints per staged row: 16-byte aligned, conflict-free 16-byte column writes
    constexpr int PR = (NST * STAGE) / (CTP * 4) / 32 * 32 < BM ? (NST * STAGE) / (CTP * 4) / 32 * 32 : BM;
    static_assert(PR >= 32, "at least one MFMA row tile per pass");
    int* ct = reinterpret_cast<int*>(smem);
    const int hh2 = lane >> 5;
    for (int r0 = 0; r0 < BM; r0 += PR) {
      if (r0 > 0) __syncthreads();  // the previous pass has been read out
#pragma unroll
      for (int mi = 0; mi < TM; ++mi) {
        const int lrow = (wm * TM + mi) * 32 + (lane & 31);
        if (lrow >= r0 && lrow < r0 + PR) {
#pragma unroll
          for (int ni = 0; ni < TN; ++ni)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
              const int lcol = (wn * TN + ni) * 32 + 8 * g + 4 * hh2;
              const int gcol = n0 + lcol;
              int4 o = make_int4(0, 0, 0, 0);
              if (blockIdx.y == 0 && gcol < p.N) o = *reinterpret_cast<const int4*>(p.ocp + gcol);  // padded to Npad
              *reinterpret_cast<int4*>(ct + (lrow - r0) * CTP + lcol) =
                  make_int4(acc[mi][ni][g * 4 + 0] + o.x, acc[mi][ni][g * 4 + 1] + o.y, acc[mi][ni][g * 4 + 2] + o.z,
                            acc[mi][ni][g * 4 + 3] + o.w);
            }
        }
      }
      __syncthreads();
      if ((p.N & 3) == 0) {
        constexpr int CPR4 = BN / 4;
        for (int idx = tid; idx < PR * CPR4; idx += NT) {
          const int row = idx / CPR4, ch = idx - row * CPR4;
          const int grow = m0 + r0 + row, gcol = n0 + ch * 4;
          if (r0 + row < BM && grow < p.M && gcol < p.N)
            *reinterpret_cast<int4*>(slab + (size_t)grow * p.N + gcol) =
                *reinterpret_cast<const int4*>(ct + row * CTP + ch * 4);
        }
      } else {
        for (int idx = tid; idx < PR * BN; idx += NT) {
          const int row = idx / BN, col = idx - row * BN;
          const int grow = m0 + r0 + row, gcol = n0 + col;
          if (r0 + row < BM && grow < p.M && gcol < p.N) slab[(size_t)grow * p.N + gcol] = ct[row * CTP + col];
        }
      }
    }
    return;
  }

  // ---- epilogue: (bias) -> requant -> (relu) -> LDS tile [BM][BN] -> 16-B row stores ----
  const Requant rq = p.rq;
  const int lo = p.relu_lo;
  const float lof = (float)lo;
#pragma unroll
  for (int ni = 0; ni < TN; ++ni) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int lcol0 = (wn * TN + ni) * 32 + 8 * g + 4 * (lane >> 5);
      const int gcol0 = n0 + lcol0;
      float4 bfv = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
      if (BIAS && gcol0 < p.N) bfv = *reinterpret_cast<const float4*>(p.biasf + gcol0);  // padded to Npad
      // (feature tiles wider than 128 may run past Npad: columns >= N are never stored)
      const int4 ocv = gcol0 < p.N ? *reinterpret_cast<const int4*>(p.ocp + gcol0) : make_int4(0, 0, 0, 0);
#pragma unroll
      for (int mi = 0; mi < TM; ++mi) {
        const int lrow = (wm * TM + mi) * 32 + (lane & 31);
        int cv[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          int c = acc[mi][ni][g * 4 + r] + (r == 0 ? ocv.x : r == 1 ? ocv.y : r == 2 ? ocv.z : ocv.w);
          if (ACC) {
            const int grow = m0 + lrow, gcol = gcol0 + r;
            if (grow < p.M && gcol < p.N) p.acc[(size_t)grow * p.N + gcol] = c;
          }
          if (BIAS) c = (int)((float)c + (r == 0 ? bfv.x : r == 1 ? bfv.y : r == 2 ? bfv.z : bfv.w));  // src/fully_connected.cc:44
          cv[r] = c;
        }
        const uint32_t packed = requant_pack4(cv, rq, lo, lof);
        *reinterpret_cast<uint32_t*>(smem + lrow * SROW + lcol0) = packed;
      }
    }
  }
  __syncthreads();
  if (p.vec_store) {
    constexpr int CPR = BN / 16;
    if (AMODE == 1 && CPR == 8) {
      // thread -> (row (tid >> 3) + NT/8 * i, chunk tid & 7): the rows of the staging map, so o_pix applies
#pragma unroll
      for (int i = 0; i < A_PER; ++i) {
        const int lrow = (tid >> 3) + (NT >> 3) * i, ch = tid & 7;
        const int grow = m0 + lrow, gcol = n0 + ch * 16;
        if (grow < p.M && gcol < p.N) {
          const size_t orow = p.ob > 0 ? (size_t)o_pix[i] : (size_t)grow;
          const uint32_t* s = reinterpret_cast<const uint32_t*>(smem + lrow * SROW + ch * 16);
          *reinterpret_cast<uint4*>(p.out + orow * p.N + gcol) = make_uint4(s[0], s[1], s[2], s[3]);
        }
      }
    } else {
      for (int idx = tid; idx < BM * CPR; idx += NT) {
        const int lrow = idx / CPR, ch = idx - lrow * CPR;
        const int grow = m0 + lrow, gcol = n0 + ch * 16;
        if (grow < p.M && gcol < p.N) {
          size_t orow = (size_t)grow;
          if (p.ob > 0) {  // physically padded NHWC output: interior pixel (oh + ob, ow + ob)
            const int P = p.OH * p.OW;
            const int img = grow / P, rem = grow - img * P;
            const int oh = rem / p.OW, ow = rem - oh * p.OW;
            orow = ((size_t)img * p.OHp + oh + p.ob) * p.OWp + ow + p.ob;
          }
          const uint32_t* s = reinterpret_cast<const uint32_t*>(smem + lrow * SROW + ch * 16);
          *reinterpret_cast<uint4*>(p.out + orow * p.N + gcol) = make_uint4(s[0], s[1], s[2], s[3]);
        }
      }
    }
  } else {
    for (int idx = tid; idx < BM * BN; idx += NT) {
      const int lrow = idx / BN, lcol = idx - lrow * BN;
      const int grow = m0 + lrow, gcol = n0 + lcol;
      if (grow < p.M && gcol < p.N) p.out[(size_t)grow * p.N + gcol] = smem[lrow * SROW + lcol];
    }
  }
}

// ---- max-pool over (optionally bordered) NHWC u8: src/functional.cc:36-64 on the internal layout
__device__ __forceinline__ uint32_t bmax4(uint32_t a, uint32_t b) {
  // per-byte unsigned max via 16-bit lanes (even / odd bytes)
  const uint32_t ae = a & 0x00FF00FFu, ao = (a >> 8) & 0x00FF00FFu;
  const uint32_t be = b & 0x00FF00FFu, bo = (b >> 8) & 0x00FF00FFu;
  const uint32_t dev = (ae | 0x01000100u) - be;  // bit 8 of each 16-bit lane set iff a >= b
  const uint32_t dod = (ao | 0x01000100u) - bo;
  const uint32_t me = ((dev >> 8) & 0x00010001u) * 0xFFu;
  const uint32_t mo = ((dod >> 8) & 0x00010001u) * 0xFFu;
  const uint32_t re = (ae & me) | (be & ~me & 0x00FF00FFu);
  const uint32_t ro = (ao & mo) | (bo & ~mo & 0x00FF00FFu);
  return re | (ro << 8);
}

// Byte-wise unsigned max without SWAR arithmetic: a dword's even and odd bytes are spread into two pairs of
// 16-bit lanes (v_perm_b32, zero fill), the running maxima live in that form (v_pk_max_u16: two bytes per
// instruction) and are packed back once per output (one more v_perm_b32).  The SWAR form (bmax4 above) cost
// ~10 VALU operations per input dword and made this kernel VALU-bound (rocprofv3: 43 M VALU wave-instructions
// per launch); this form needs 4.
typedef unsigned short us2 __attribute__((ext_vector_type(2)));
struct BytesEO {
  us2 e, o;
};
__device__ __forceinline__ BytesEO spread_eo(uint32_t x) {
  BytesEO r;
  r.e = __builtin_bit_cast(us2, __builtin_amdgcn_perm(0u, x, 0x0c020c00u));  // [x0, 0, x2, 0]
  r.o = __builtin_bit_cast(us2, __builtin_amdgcn_perm(0u, x, 0x0c030c01u));  // [x1, 0, x3, 0]
  return r;
}
__device__ __forceinline__ void max_eo(BytesEO& m, uint32_t x) {
  const BytesEO v = spread_eo(x);
  m.e = __builtin_elementwise_max(m.e, v.e);
  m.o = __builtin_elementwise_max(m.o, v.o);
}
__device__ __forceinline__ uint32_t pack_eo(const BytesEO& m) {
  return __builtin_amdgcn_perm(__builtin_bit_cast(uint32_t, m.o), __builtin_bit_cast(uint32_t, m.e), 0x06020400u);
}

// KT: compile-time window size (0 = run-time k): the KT * KT window loads are issued together
template <int KT>
__global__ __launch_bounds__(256) void maxpool_u8_nhwc_kernel(const uint8_t* __restrict__ in,
                                                              uint8_t* __restrict__ out, uint32_t total, int inHp,
                                                              int inWp, int ib, int c16, int oh, int ow, int k, int s,
                                                              int outHp, int outWp, int ob, uint32_t lo4) {
  // XCD-aware partition (gridDim.x % 8 == 0): blocks with equal blockIdx % 8 share an L2, so each of the
  // eight groups sweeps one contiguous eighth of the output -- window rows shared by neighbouring output
  // rows are then fetched into one L2 once (rocprofv3 FETCH_SIZE: 1.37x the input bytes without this,
  // 1.07x with).  `total` (16-byte output chunks) < 2^31: the launcher splits larger batches, which keeps the
  // index arithmetic in 32 bits (three 64-bit divisions per chunk were a third of the instruction count).
  const uint32_t slab = ((total + 7) / 8 + 255) / 256 * 256;
  const uint32_t lo = (blockIdx.x & 7) * slab;
  const uint32_t hi = lo + slab < total ? lo + slab : total;
  const uint32_t gstride = (gridDim.x >> 3) * 256u;
  const BytesEO m0 = spread_eo(lo4);  // 0 (reference's running max start), or zp when relu is folded in
  for (uint32_t e = lo + (blockIdx.x >> 3) * 256u + threadIdx.x; e < hi; e += gstride) {
    const uint32_t t0 = e / (uint32_t)c16, cc = e - t0 * c16;
    const uint32_t t1 = t0 / (uint32_t)ow, x = t0 - t1 * ow;
    const uint32_t img = t1 / (uint32_t)oh, y = t1 - img * oh;
    const uint4* p = reinterpret_cast<const uint4*>(in) +
                     (((size_t)img * inHp + (size_t)y * s + ib) * inWp + (size_t)x * s + ib) * c16 + cc;
    BytesEO mx = m0, my = m0, mz = m0, mw = m0;
    if constexpr (KT > 0) {
      uint4 v[KT * KT];
#pragma unroll
      for (int a = 0; a < KT; ++a)
#pragma unroll
        for (int b = 0; b < KT; ++b) v[a * KT + b] = p[((size_t)a * inWp + b) * c16];
#pragma unroll
      for (int i = 0; i < KT * KT; ++i) {
        max_eo(mx, v[i].x);
        max_eo(my, v[i].y);
        max_eo(mz, v[i].z);
        max_eo(mw, v[i].w);
      }
    } else {
      for (int a = 0; a < k; ++a)
        for (int b = 0; b < k; ++b) {
          const uint4 v = p[((size_t)a * inWp + b) * c16];
          max_eo(mx, v.x);
          max_eo(my, v.y);
          max_eo(mz, v.z);
          max_eo(mw, v.w);
        }
    }
    reinterpret_cast<uint4*>(out)[(((size_t)img * outHp + y + ob) * outWp + x + ob) * c16 + cc] =
        make_uint4(pack_eo(mx), pack_eo(my), pack_eo(mz), pack_eo(mw));
  }
}

// ---- layout conversion u8 through a 64 x 64 LDS tile per image ---------------------------------
// "planar" side: [R][S] contiguous per image (NCHW: R = c, S = h*w).  "pixel" side: NHWC with an
// optional physical border: pixel s lives at row ((s / W + b) * Wp + s % W + b), pitch R bytes.
template <bool TO_NHWC>
__global__ __launch_bounds__(256) void transpose_u8_kernel(const uint8_t* __restrict__ in, uint8_t* __restrict__ out,
                                                           int R, int S, int W, int Wp, int b, int64_t img_rows_p) {
  __shared__ uint8_t tile[64][65];
  const int64_t img = blockIdx.z;
  const int r0 = blockIdx.y * 64, s0 = blockIdx.x * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  auto prow = [&](int s) -> int64_t { return ((int64_t)(s / W + b) * Wp + (s % W) + b); };
  if (TO_NHWC) {
    const uint8_t* src = in + img * (int64_t)R * S;
    uint8_t* dst = out + img * img_rows_p * R;
    for (int j = ty; j < 64; j += 4) {
      const int r = r0 + j, s = s0 + tx;
      tile[j][tx] = (r < R && s < S) ? src[(int64_t)r * S + s] : 0;
    }
    __syncthreads();
    for (int j = ty; j < 64; j += 4) {
      const int s = s0 + j, r = r0 + tx;
      if (r < R && s < S) dst[prow(s) * R + r] = tile[tx][j];
    }
  } else {
    const uint8_t* src = in + img * img_rows_p * R;
    uint8_t* dst = out + img * (int64_t)R * S;
    for (int j = ty; j < 64; j += 4) {
      const int s = s0 + j, r = r0 + tx;
      tile[j][tx] = (r < R && s < S) ? src[prow(s) * R + r] : 0;
    }
    __syncthreads();
    for (int j = ty; j < 64; j += 4) {
      const int r = r0 + j, s = s0 + tx;
      if (r < R && s < S) dst[(int64_t)r * S + s] = tile[tx][j];
    }
  }
}

// NHWC (border ib) -> NHWC (border ob >= 0), border bytes = zp; one thread per 16-B chunk of dst
__global__ __launch_bounds__(256) void reborder_u8_kernel(const uint8_t* __restrict__ in, uint8_t* __restrict__ out,
                                                          int64_t total, int h, int w, int c16, int ib, int ob,
                                                          uint32_t zp4) {
  const int inHp = h + 2 * ib, inWp = w + 2 * ib, oHp = h + 2 * ob, oWp = w + 2 * ob;
  (void)inHp;
  const int64_t gstride = (int64_t)gridDim.x * 256;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += gstride) {
    const int cc = (int)(e % c16);
    int64_t t = e / c16;
    const int xp = (int)(t % oWp);
    t /= oWp;
    const int yp = (int)(t % oHp);
    const int64_t img = t / oHp;
    const int y = yp - ob, x = xp - ob;
    uint4 v = make_uint4(zp4, zp4, zp4, zp4);
    if (y >= 0 && y < h && x >= 0 && x < w)
      v = reinterpret_cast<const uint4*>(in)[((img * (h + 2 * ib) + y + ib) * inWp + x + ib) * c16 + cc];
    reinterpret_cast<uint4*>(out)[e] = v;
  }
}

// fill only the border pixels of a bordered NHWC tensor [n][h+2b][w+2b][c] with the zero point
__global__ __launch_bounds__(256) void fill_border_kernel(uint8_t* __restrict__ out, int64_t total, int h, int w,
                                                          int c16, int b, uint32_t zp4) {
  const int Hp = h + 2 * b, Wp = w + 2 * b;
  const int top = b * Wp, sides = h * 2 * b, nb = 2 * top + sides;
  const int64_t gstride = (int64_t)gridDim.x * 256;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += gstride) {
    const int cc = (int)(e % c16);
    int64_t t = e / c16;
    const int bp = (int)(t % nb);
    const int64_t img = t / nb;
    int yp, xp;
    if (bp < top) {
      yp = bp / Wp;
      xp = bp - yp * Wp;
    } else if (bp < top + sides) {
      const int r = (bp - top) / (2 * b), s = (bp - top) - r * (2 * b);
      yp = b + r;
      xp = s < b ? s : w + s;
    } else {
      const int q = bp - top - sides;
      yp = b + h + q / Wp;
      xp = q % Wp;
    }
    reinterpret_cast<uint4*>(out)[((img * Hp + yp) * Wp + xp) * c16 + cc] = make_uint4(zp4, zp4, zp4, zp4);
  }
}

// ---- small-C repack: NCHW u8 [n][c<=4][h][w] -> physically padded "grouped" NHWC -----------------
// out [n][Hp][Wg][16]: pixel (y, x) of the padded image (pad rows/cols hold the zero point;
// x = 4*g + px) stores its channels at bytes 4*px .. 4*px+3 of group g (channels >= c hold zp;
// their weights are zero).  Makes the stride-4 11x11x3 first layer an ordinary C=16, 11x3-tap
// implicit GEMM with 16-byte-aligned, predicate-free gathers.
__global__ __launch_bounds__(256) void repack_smallc_kernel(const uint8_t* __restrict__ in, uint8_t* __restrict__ out,
                                                            int64_t total, int c, int h, int w, int Hp, int Wg,
                                                            int ph, int pw, uint32_t zp, uint32_t rebias) {
  const int64_t gstride = (int64_t)gridDim.x * 256;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += gstride) {
    const int g = (int)(e % Wg);
    int64_t t = e / Wg;
    const int yp = (int)(t % Hp);
    const int64_t img = t / Hp;
    const int y = yp - ph;
    uint32_t wds[4];
#pragma unroll
    for (int px = 0; px < 4; ++px) {
      const int x = g * 4 + px - pw;
      uint32_t v = zp * 0x01010101u;
      if (y >= 0 && y < h && x >= 0 && x < w) {
        v = 0;
        for (int ch = 0; ch < 4; ++ch) {
          const uint32_t b = ch < c ? (uint32_t)in[((img * c + ch) * h + y) * w + x] : zp;
          v |= b << (8 * ch);
        }
      }
      wds[px] = v ^ rebias;  // 0x80808080 when the weights-stationary kernel (i8ie_first.hip) consumes it
    }
    reinterpret_cast<uint4*>(out)[e] = make_uint4(wds[0], wds[1], wds[2], wds[3]);
  }
}

// ---- Linear with a handful of output features (the classifier head: fc8, N = 10) ------------------------
// One wave per input row: every lane walks K in 16-byte chunks (chunk = lane, lane + 64, ...), v_dot4_i32_i8
// against the N weight rows (the panel is a few tens of KB: L1/L2 resident for every wave), a wavefront
// butterfly sums the 64 partial dot products, lane j finishes feature j with the reference's Linear epilogue
// (src/fully_connected.cc:42-48: + oc, + (float)q_b / s_in in float, truncate, down_scale) and, when asked,
// the dequantize that follows it in every network (src/quantize_utils.cc:54-58).  Replaces a 128 x 32 MFMA
// tile launch that is 92 % padding, its split-K reduction and the separate dequantize launch.
constexpr int kSmallN = 16;

struct SmallNArgs {
  const uint8_t* A;
  size_t lda;
  int M, K;          // K % 16 == 0
  const int8_t* B;   // [>= N][Kpad], K order matching A, zero padded
  int Kpad, N;
  const int32_t* ocp;  // oc + 128 * wsum
  const float* biasf;
  Requant rq;
  int relu_lo;
  uint8_t* out;      // [M][N] or nullptr
  int32_t* acc;      // [M][N] or nullptr (pre-bias INT32 accumulators)
  float* out_f32;    // [M][N] or nullptr: (q - zp_out) * s_out
  float dq_scale;
  int dq_zp;
};

// WPR = waves per row: 1 (four rows per block) for short K, 4 (one row per block, K split over 256 threads
// and the four partial sums joined through LDS) when K is long enough to make one wave's walk a latency chain
template <int WPR>
__global__ __launch_bounds__(256) void linear_smalln_kernel(SmallNArgs p) {
  __shared__ int part[4][kSmallN];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int row = WPR == 1 ? blockIdx.x * 4 + wave : blockIdx.x;
  const bool live = row < p.M;
  int sum[kSmallN];
#pragma unroll
  for (int j = 0; j < kSmallN; ++j) sum[j] = 0;
  if (live) {
    const uint8_t* arow = p.A + (size_t)row * p.lda;
    const int t = WPR == 1 ? lane : (int)threadIdx.x;
    for (int k0 = t * 16; k0 < p.K; k0 += 64 * WPR * 16) {
      const v4i a = *reinterpret_cast<const v4i*>(arow + k0) ^ (int)0x80808080;  // u8 -> s8 (128 * wsum is in ocp)
#pragma unroll
      for (int j = 0; j < kSmallN; ++j) {
        if (j < p.N) {
          const v4i wv = *reinterpret_cast<const v4i*>(p.B + (size_t)j * p.Kpad + k0);
          int acc = sum[j];
          acc = __builtin_amdgcn_sdot4(a.x, wv.x, acc, false);
          acc = __builtin_amdgcn_sdot4(a.y, wv.y, acc, false);
          acc = __builtin_amdgcn_sdot4(a.z, wv.z, acc, false);
          acc = __builtin_amdgcn_sdot4(a.w, wv.w, acc, false);
          sum[j] = acc;
        }
      }
    }
  }
  // wavefront reduction (integer adds: any order is exact)
#pragma unroll
  for (int j = 0; j < kSmallN; ++j) {
    if (j < p.N) {
      int v = sum[j];
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
      sum[j] = v;
    }
  }
  int c = 0;
#pragma unroll
  for (int j = 0; j < kSmallN; ++j)
    if (lane == j) c = sum[j];
  if (WPR > 1) {
    if (lane < kSmallN) part[wave][lane] = c;
    __syncthreads();
    if (wave != 0) return;
    c = 0;
    if (lane < kSmallN) c = part[0][lane] + part[1][lane] + part[2][lane] + part[3][lane];
  }
  if (live && lane < p.N) {
    c += p.ocp[lane];
    const size_t o = (size_t)row * p.N + lane;
    if (p.acc) p.acc[o] = c;
    c = (int)((float)c + p.biasf[lane]);  // src/fully_connected.cc:44
    const int q = requant_exact((float)c, p.rq, p.relu_lo);
    if (p.out) p.out[o] = (uint8_t)q;
    if (p.out_f32) p.out_f32[o] = (float)(q - p.dq_zp) * p.dq_scale;  // src/quantize_utils.cc:38-42
  }
}

// The same head for long K on the MFMA path: one block per 16 rows, the K steps of 64 bytes dealt round-robin to the
// block's 8 waves, every wave's loads (one activation and one weight fragment per step, 16 bytes per lane) issued
// before its first v_mfma_i32_16x16x64_i8, the 8 partial tiles joined through LDS.  The dot4 form above walks a chain
// of 60 cross-lane adds per row after its loads (11-13 us for fc8 whatever the batch); this one is one memory
// round trip, 8 MFMAs and one LDS exchange.
constexpr int kHeadSteps = 8;  // K steps in flight per wave and round
__global__ __launch_bounds__(512) void linear_head_mfma_kernel(SmallNArgs p) {
  __shared__ __attribute__((aligned(16))) int part[8 * 16 * 16];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lq = lane >> 4, lr = lane & 15;
  const int row0 = blockIdx.x * 16;
  const size_t a_off = (size_t)row0 * p.lda;
  const size_t a_left = (size_t)p.M * p.lda - a_off;
  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(p.A + a_off), 0,
                                                                       (unsigned)(a_left < 0xFFFFF000u ? a_left : 0xFFFFF000u), 0x00020000);
  const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<int8_t*>(p.B), 0, (unsigned)(16 * p.Kpad), 0x00020000);
  const unsigned av = (unsigned)lr * (unsigned)p.lda + (unsigned)lq * 16u;    // row r, bytes 16 q .. of a 64-byte step
  const unsigned bv = (unsigned)lr * (unsigned)p.Kpad + (unsigned)lq * 16u;   // feature r
  const int nks = p.Kpad / 64;
  v4i acc = {0, 0, 0, 0};
  for (int s0 = wave; s0 < nks; s0 += 8 * kHeadSteps) {
    v4i a[kHeadSteps], b[kHeadSteps];
#pragma unroll
    for (int i = 0; i < kHeadSteps; ++i) {
      const int ks = s0 + 8 * i;
      const unsigned k = (unsigned)(ks < nks ? ks : 0) * 64u;
      a[i] = __builtin_bit_cast(v4i, __builtin_amdgcn_raw_buffer_load_b128(rsA, (int)(av + k), 0, 0));
      b[i] = __builtin_bit_cast(v4i, __builtin_amdgcn_raw_buffer_load_b128(rsB, (int)(bv + k), 0, 0));
      if (ks >= nks) b[i] = v4i{0, 0, 0, 0};
    }
#pragma unroll
    for (int i = 0; i < kHeadSteps; ++i) acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(b[i], a[i] ^ (int)0x80808080, acc, 0, 0, 0);
  }
  // lane (q, r) holds features 4q .. 4q + 3 of row r
  *reinterpret_cast<v4i*>(&part[(wave * 16 + lr) * 16 + 4 * lq]) = acc;
  __syncthreads();
  if (tid < 256) {
    const int r = tid >> 4, j = tid & 15;
    int c = 0;
#pragma unroll
    for (int w = 0; w < 8; ++w) c += part[(w * 16 + r) * 16 + j];
    const int row = row0 + r;
    if (row < p.M && j < p.N) {
      c += p.ocp[j];
      const size_t o = (size_t)row * p.N + j;
      if (p.acc) p.acc[o] = c;
      c = (int)((float)c + p.biasf[j]);  // src/fully_connected.cc:44
      const int q = requant_exact((float)c, p.rq, p.relu_lo);
      if (p.out) p.out[o] = (uint8_t)q;
      if (p.out_f32) p.out_f32[o] = (float)(q - p.dq_zp) * p.dq_scale;  // src/quantize_utils.cc:38-42
    }
  }
}

// Linear weight panel [rows][Kpad] with K reordered from the reference's flattened NCHW (c, h*w) to the
// flattened NHWC (h*w, c) of the engine's activations; padding columns stay zero
__global__ __launch_bounds__(256) void permute_k_kernel(const int8_t* __restrict__ B, int8_t* __restrict__ Bp,
                                                        int64_t total, int Kpad, int K, int c, int hw) {
  const int64_t gstride = (int64_t)gridDim.x * 256;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += gstride) {
    const int kq = (int)(e % Kpad);
    const int64_t row = e / Kpad;
    int8_t v = 0;
    if (kq < K) {
      const int pix = kq / c, ch = kq - pix * c;
      v = B[row * Kpad + (int64_t)ch * hw + pix];
    }
    Bp[e] = v;
  }
}

// cached per (layer, s_in, zp_in): ocp = oc + 128*wsum, biasf = (float)qb / s_in
__global__ __launch_bounds__(64) void finish_offsets_kernel(const int32_t* __restrict__ oc,
                                                            const int32_t* __restrict__ wsum,
                                                            const int8_t* __restrict__ qb, float s_in, int n,
                                                            int32_t* __restrict__ ocp, float* __restrict__ biasf) {
  const int j = blockIdx.x * 64 + threadIdx.x;
  if (j >= n) return;
  ocp[j] = oc[j] + 128 * wsum[j];
  if (biasf != nullptr) biasf[j] = (float)qb[j] / s_in;
}

// split-K finish: C = sum of the slices' partials (slice 0 carries ocp), then the Linear epilogue of
// src/fully_connected.cc:42-48 on 4 consecutive features per thread
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const int32_t* __restrict__ partial, int ksplit, int M,
                                                            int N, const float* __restrict__ biasf, Requant rq,
                                                            int relu_lo, uint8_t* __restrict__ out,
                                                            int32_t* __restrict__ acc) {
  const int64_t total = (int64_t)M * N;
  const int64_t gstride = (int64_t)gridDim.x * 256;
  const float lof = (float)relu_lo;
  for (int64_t e = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4; e < total; e += gstride * 4) {
    int cv[4];
    const int nvalid = total - e < 4 ? (int)(total - e) : 4;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      int c = 0;
      if (r < nvalid) {
        for (int s = 0; s < ksplit; ++s) c += partial[(size_t)s * total + e + r];
        if (acc != nullptr) acc[e + r] = c;
        if (biasf != nullptr) c = (int)((float)c + biasf[(e + r) % N]);
      }
      cv[r] = c;
    }
    const uint32_t packed = requant_pack4(cv, rq, relu_lo, lof);
#pragma unroll
    for (int r = 0; r < 4; ++r)
      if (r < nvalid) out[e + r] = (uint8_t)(packed >> (8 * r));
  }
}

inline int cap_grid(int64_t items, int threads, int max_blocks = 256 * 16) {
  int64_t b = (items + threads - 1) / threads;
  if (b < 1) b = 1;
  return (int)(b > max_blocks ? max_blocks : b);
}

template <int AMODE, int WM, int WN, int TM, int TN, bool BIAS, bool ACC, int VAR = 0>
int launch_cfg(i8ie_ctx* ctx, const IgemmArgs& a, const char* name, int kbytes, double ops, double bytes) {
  constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
  const int tiles_m = (a.M + BM - 1) / BM, tiles_n = (a.N + BN - 1) / BN;
  const int m_fastest = ((size_t)a.N > (size_t)a.M) ? 1 : 0;
  char tag[64];
  snprintf(tag, sizeof(tag), "%s|M%d,N%d,K%d", name, a.M, a.N, kbytes);
  I8ieProfScope prof(ctx, ctx->prof ? tag : name, ops, bytes);
  constexpr unsigned dyn_lds = VAR == 7 ? 2u * 65536u : 0u;
  if (VAR == 7) {
    static bool raised_on[64] = {};  // per instantiation and device: allow 128 KiB of dynamic LDS
    bool& raised = raised_on[ctx->device & 63];
    if (!raised) {
      I8IE_HIP_TRY(hipFuncSetAttribute(
          reinterpret_cast<const void*>(&igemm_u8s8_kernel<AMODE, WM, WN, TM, TN, BIAS, ACC, VAR>),
          hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn_lds));
      raised = true;
    }
  }
  igemm_u8s8_kernel<AMODE, WM, WN, TM, TN, BIAS, ACC, VAR>
      <<<dim3(tiles_m * tiles_n, a.ksplit > 1 ? a.ksplit : 1), WM * WN * 64, dyn_lds, ctx->stream>>>(
          a, tiles_m, tiles_n, m_fastest);
  I8IE_LAUNCH_CHECK();
  return I8IE_OK;
}

template <int AMODE, bool BIAS, bool ACC, int VAR>
int launch_tile_var(i8ie_ctx* ctx, const IgemmArgs& a, int kbytes, double ops, double bytes) {
  // widest N tile that still gives the chip enough blocks: small-M Linear layers (one or a few M tiles)
  // otherwise run on N/128 blocks only (M = 125, N = 4096: 32 blocks on 256 CUs)
  const long tiles_m = (a.M + 127) / 128;
  int bn = a.N <= 32 ? 32 : a.N <= 64 ? 64 : a.N <= 96 ? 96 : 128;
  if (AMODE == 0 && bn == 128) {
    if (tiles_m * ((a.N + 127) / 128) < 512) bn = 64;
    // (with split-K the K slices already fill the chip: 64-wide tiles then halve the activation re-reads;
    // measured 0.047 vs 0.052 ms for fc6 + fc7 at 125 rows)
    if (tiles_m * ((a.N + 63) / 64) < 256 && a.ksplit <= 1) bn = 32;
  }
  if (bn == 32)
    return launch_cfg<AMODE, 4, 1, 1, 1, BIAS, ACC, VAR>(ctx, a, AMODE ? "igemm_conv_128x32" : "igemm_lin_128x32", kbytes,
                                                         ops, bytes);
  if (bn == 64)
    return launch_cfg<AMODE, 2, 2, 2, 1, BIAS, ACC, VAR>(ctx, a, AMODE ? "igemm_conv_128x64" : "igemm_lin_128x64", kbytes,
                                                         ops, bytes);
  if (bn == 96)
    return launch_cfg<AMODE, 4, 1, 1, 3, BIAS, ACC, VAR>(ctx, a, AMODE ? "igemm_conv_128x96" : "igemm_lin_128x96", kbytes,
                                                         ops, bytes);
  return launch_cfg<AMODE, 2, 2, 2, 2, BIAS, ACC, VAR>(ctx, a, AMODE ? "igemm_conv_128x128" : "igemm_lin_128x128", kbytes,
                                                       ops, bytes);
}

// Default: one LDS stage filled by LDS-DMA (VAR 5).  Measured on MI355X (tools/bench_layer.py, AlexNet
// conv2-5): one stage (half the LDS) lets a third wave per SIMD in, +10 % over the two-stage form; DMA
// staging +3-6 % over register staging (VAR 3); loads two tiles ahead and s_setprio around the MFMAs
// within 2 % of the baseline; the 256-row two-stage DMA form (VAR 7) 10-20 % slower on these K depths
// ("what bounds the contraction kernel").  ctx->variant selects the others.
template <int AMODE, bool BIAS, bool ACC>
int launch_tile(i8ie_ctx* ctx, const IgemmArgs& a, int kbytes, double ops, double bytes) {
  if (AMODE == 1 && !BIAS && !ACC) {
#if defined(I8IE_DIAG)  // tile-shape experiments, compiled into the diagnostic build only
    if (ctx->variant == 10) return launch_tile_var<1, false, false, 0>(ctx, a, kbytes, ops, bytes);
    if (ctx->variant == 6 && a.N <= 96 && a.N > 64)  // 256 x 96 tile, 4 waves of 64 x 96
      return launch_cfg<1, 4, 1, 2, 3, false, false, 3>(ctx, a, "igemm_conv_256x96", kbytes, ops, bytes);
    if (ctx->variant == 4 && a.N % 256 == 0)  // 256 x 256 tile, 8 waves, two stages
      return launch_cfg<1, 2, 4, 4, 2, false, false, 0>(ctx, a, "igemm_conv_256x256", kbytes, ops, bytes);
    if (ctx->variant == 8 && a.N > 64)  // 256 x 128, 4 waves of 128 x 64, one DMA stage
      return launch_cfg<1, 2, 2, 4, 2, false, false, 5>(ctx, a, "igemm_conv_256x128", kbytes, ops, bytes);
    if (ctx->variant == 9 && a.N > 128)  // 128 x 256, 4 waves of 64 x 128
      return launch_cfg<1, 2, 2, 2, 4, false, false, 5>(ctx, a, "igemm_conv_128x256", kbytes, ops, bytes);
#endif
    // 192 x 128, 4 waves of 96 x 64, one DMA stage: 6.4 KB staged per MOP (128 x 128: 7.6) at three blocks
    // per CU; measured 2.5-4 % faster than 128 x 128 on AlexNet conv2-5.  Only the variants that name a staging form or tile
    // shape of THIS kernel (3, 5 and the diagnostic 4-10) switch it to 128 x 128; variants that select other kernels (12, 50,
    // 54, 70, 80, 81 ...) leave the tiled kernel's default alone, so an A/B run with them changes one thing
    const bool tile_variant = ctx->variant == 3 || ctx->variant == 5 || (ctx->variant >= 4 && ctx->variant <= 10);
    if (!tile_variant && a.N > 64 && a.M >= 192 * 256)
      return launch_cfg<1, 2, 2, 3, 2, false, false, 5>(ctx, a, "igemm_conv_192x128", kbytes, ops, bytes);
#if defined(I8IE_DIAG)
    if (ctx->variant == 7 && a.N > 128 && (long)((a.M + 255) / 256) >= 256) {
      // 4 waves of 128 x 128 or 128 x 96: whichever pads N less
      const int pad4 = (a.N + 255) / 256 * 256, pad3 = (a.N + 191) / 192 * 192;
      if (pad4 <= pad3) return launch_cfg<1, 2, 2, 4, 4, false, false, 7>(ctx, a, "igemm_conv_256x256", kbytes, ops, bytes);
      return launch_cfg<1, 2, 2, 4, 3, false, false, 7>(ctx, a, "igemm_conv_256x192", kbytes, ops, bytes);
    }
#endif
  }
#if defined(I8IE_DIAG)
  if (AMODE == 0 && ctx->variant == 13 && a.ksplit <= 1)  // Linear, 128 x 128 tiles, two 64 KiB DMA stages, one block per CU
    return launch_cfg<AMODE, 2, 2, 2, 2, BIAS, ACC, 7>(ctx, a, "igemm_lin_128x128_dma2", kbytes, ops, bytes);
  if (AMODE == 0 && ctx->variant == 14 && a.ksplit <= 1)  // Linear, 128 x 128 tiles, one DMA stage
    return launch_cfg<AMODE, 2, 2, 2, 2, BIAS, ACC, 5>(ctx, a, "igemm_lin_128x128_dma1", kbytes, ops, bytes);
  if (AMODE == 0 && ctx->variant == 15 && a.ksplit <= 1)  // Linear, 128 x 128 tiles, register staging
    return launch_cfg<AMODE, 2, 2, 2, 2, BIAS, ACC, 3>(ctx, a, "igemm_lin_128x128_reg", kbytes, ops, bytes);
#endif
  // Linear (few, short split-K slices per block) measured 20 % slower with DMA staging: register staging there
  if (ctx->variant == 3 || (AMODE == 0 && ctx->variant != 5))
    return launch_tile_var<AMODE, BIAS, ACC, 3>(ctx, a, kbytes, ops, bytes);
  return launch_tile_var<AMODE, BIAS, ACC, 5>(ctx, a, kbytes, ops, bytes);
}

}  // namespace

// ---- entry points used by i8ie_layer.hip ---------------------------------------------------

// The gathers address A with 32-bit buffer offsets.  A batch whose activations exceed that range (288 GB of
// HBM hold far more than 4 GiB) runs as several launches over whole images / rows; $I8IE_IGEMM_CHUNK_BYTES
// lowers the limit so that tests can walk the chunk loop with small tensors.
static size_t igemm_chunk_limit() {
  static const size_t v = [] {
    const char* e = std::getenv("I8IE_IGEMM_CHUNK_BYTES");
    const size_t hard = ((size_t)1 << 32) - 4096;
    const size_t want = e ? (size_t)std::strtoull(e, nullptr, 10) : hard;
    return want > 0 && want < hard ? want : hard;
  }();
  return v;
}

size_t i8ie_igemm_chunk_limit() { return igemm_chunk_limit(); }

int i8ie_igemm_launch(i8ie_ctx* ctx, const I8ieIgemmCall& c) {
  I8IE_REQUIRE(c.M > 0 && c.N > 0 && c.Kpad > 0 && c.Kpad % BK2 == 0, "igemm dimensions");
  I8IE_REQUIRE((reinterpret_cast<uintptr_t>(c.A) & 15u) == 0 && (reinterpret_cast<uintptr_t>(c.B) & 15u) == 0,
               "operands must be 16-byte aligned");
  // (the kernel also keeps bordered-output pixel indices in 32 bits)
  const size_t out_pixels = c.amode == 1 ? (size_t)(c.M / (c.OH * c.OW)) * (c.OH + 2 * c.ob) * (c.OW + 2 * c.ob) : 0;
  if (c.a_bytes >= igemm_chunk_limit() || out_pixels >= ((size_t)1 << 32)) {
    I8IE_REQUIRE(!(i8ie_is_pool(c.pool_k, c.pool_s) || c.a_s8 || c.out_s8), "igemm: pool / re-biased layout on a launch that needs chunking");
    const size_t limit = igemm_chunk_limit();
    const int P = c.amode == 1 ? c.OH * c.OW : 1;  // rows per unit (image / row)
    const size_t unit_in = c.amode == 1 ? (size_t)c.Hp * c.Wp * c.C : (size_t)c.lda;
    const int ob = c.amode == 1 ? c.ob : 0;
    const size_t unit_out = c.amode == 1 ? (size_t)(c.OH + 2 * ob) * (c.OW + 2 * ob) * c.N : (size_t)c.N;
    I8IE_REQUIRE(c.M % P == 0 && unit_in > 0 && unit_in < limit, "igemm: one image / row exceeds the offset range");
    const int units = c.M / P;
    int per = (int)((limit - 1) / unit_in);
    if (c.amode == 1) {
      const size_t opi = (size_t)(c.OH + 2 * ob) * (c.OW + 2 * ob);  // output pixels per image
      const size_t cap = (((size_t)1 << 32) - 1) / opi;
      if ((size_t)per > cap) per = (int)cap;
    }
    if (per < 1) per = 1;
    if (c.amode == 1 && (size_t)per * unit_out % 16 != 0) per -= per % 16;  // keep 16-byte aligned sub-outputs
    if (per < 1) per = 1;
    for (int u0 = 0; u0 < units; u0 += per) {
      const int nb = units - u0 < per ? units - u0 : per;
      I8ieIgemmCall sub = c;
      sub.A = c.A + (size_t)u0 * unit_in;
      const size_t left = c.a_bytes - (size_t)u0 * unit_in;
      sub.a_bytes = left < (size_t)nb * unit_in ? left : (size_t)nb * unit_in;
      sub.M = nb * P;
      sub.out = c.out + (size_t)u0 * unit_out;
      if (c.acc) sub.acc = c.acc + (size_t)u0 * P * c.N;
      I8IE_REQUIRE(sub.a_bytes < limit, "igemm: chunking failed to fit the offset range");
      I8IE_TRY(i8ie_igemm_launch(ctx, sub));
    }
    return I8IE_OK;
  }
#if defined(I8IE_DIAG)
  // The persistent ping-pong kernel (i8ie_pp.hip) is opt-in (variant 20; 21-49 are its diagnostic builds): measured
  // on MI355X it trails the tiled kernel by 3-8 % on AlexNet conv2 / conv5 as long as activations need the u8 -> s8
  // xor in its MFMA slots ("what bounds the ping-pong kernel"), and it pads N = 384 to 512.
  if (c.amode == 1 && ctx->variant >= 20 && ctx->variant < 50) {
    const int took = i8ie_pp_try_launch(ctx, c);
    if (took != 0) return took < 0 ? took : I8IE_OK;
  }
#endif
  // The two-team form of the patch-stationary kernel (i8ie_tconv.hip) comes first: it declines unless forced
  // (variant 70) or the launch is of the kind it measured faster on (one feature pass, >= 8 bands per CU).
  if (c.amode == 1 && (i8ie_conv_variant_auto(ctx->variant) || (ctx->variant >= 70 && ctx->variant < 80))) {
    const int took = i8ie_tconv_try_launch(ctx, c);
    if (took != 0) return took < 0 ? took : I8IE_OK;
  }
  // The patch-stationary kernel (i8ie_pconv.hip) takes the large convolutions it is built for (it declines the rest):
  // 2000-2280 TOP/s on AlexNet conv2-5 at 1000 images against 1790-1900 of the tiled kernel below (variant 11 forces
  // the tiled kernel, 50 forces this one at any batch size).
  if (c.amode == 1 && (i8ie_conv_variant_auto(ctx->variant) || (ctx->variant >= 50 && ctx->variant < 60))) {
    const int took = i8ie_pconv_try_launch(ctx, c);
    if (took != 0) return took < 0 ? took : I8IE_OK;
  }
  // (the kernels below know neither the folded max-pool nor the re-biased layouts: callers ask i8ie_pconv_takes first)
  I8IE_REQUIRE(!(c.amode == 1 && (i8ie_is_pool(c.pool_k, c.pool_s) || c.a_s8 || c.out_s8)), "igemm: pool / re-biased layout without the patch-stationary kernel");
  IgemmArgs a{};
  a.A = c.A;
  a.a_bytes = (unsigned)c.a_bytes;
  a.lda = (unsigned)c.lda;
  a.M = c.M;
  if (c.amode == 1) {
    I8IE_REQUIRE(c.C % 16 == 0 && c.C > 0, "NHWC gather needs channels % 16 == 0");
    a.C = (unsigned)c.C;
    a.row_pitch = (unsigned)c.Wp * (unsigned)c.C;
    a.img_pitch = (unsigned)c.Hp * a.row_pitch;
    a.sh = c.sh; a.sw = c.sw; a.OH = c.OH; a.OW = c.OW;
    a.RC = c.KW * (c.C / 16);
    a.row_jump = a.row_pitch - (unsigned)a.RC * 16u;
  }
  a.B = c.B;
  a.b_bytes = (unsigned)((size_t)c.Npad * c.Kpad);
  a.Kpad = c.Kpad;
  a.N = c.N;
  a.ocp = c.ocp;
  a.biasf = c.biasf;
  a.rq = i8ie_make_requant(c.s_in, c.s_w, c.s_out, c.zp_out);
  a.relu_lo = c.relu ? c.zp_out : 0;
  a.out = c.out;
  a.ob = c.amode == 1 ? c.ob : 0;
  a.OHp = c.OH + 2 * a.ob;
  a.OWp = c.OW + 2 * a.ob;
  a.vec_store = ((c.N & 15) == 0 && (reinterpret_cast<uintptr_t>(c.out) & 15u) == 0) ? 1 : 0;
  I8IE_REQUIRE(a.ob == 0 || a.vec_store, "bordered output needs N % 16 == 0");
  a.acc = c.acc;
  const double ops = 2.0 * c.M * c.N * c.Ktrue;
  const double bytes = (double)c.M * c.Ktrue + (double)c.N * c.Ktrue + (double)c.M * c.N;
  const int kb = c.Kchunks * 16;
  const bool bias = c.biasf != nullptr, accd = c.acc != nullptr;
  if (c.amode == 0 && c.ksplit > 1 && c.partial != nullptr) {
    const int nk = c.Kpad / BK2;
    a.tiles_per_slice = (nk + c.ksplit - 1) / c.ksplit;
    a.ksplit = (nk + a.tiles_per_slice - 1) / a.tiles_per_slice;
    a.partial = c.partial;
    if (a.ksplit > 1) {
      I8IE_TRY((launch_tile<0, false, false>(ctx, a, kb, ops, bytes)));
      I8ieProfScope prof(ctx, "splitk_reduce", 0.0, 4.0 * a.ksplit * c.M * c.N + (double)c.M * c.N);
      const int64_t quads = ((int64_t)c.M * c.N + 3) / 4;
      splitk_reduce_kernel<<<cap_grid(quads, 256), 256, 0, ctx->stream>>>(c.partial, a.ksplit, c.M, c.N, c.biasf, a.rq,
                                                                          a.relu_lo, c.out, c.acc);
      I8IE_LAUNCH_CHECK();
      return I8IE_OK;
    }
    a.ksplit = 0;
  }
  if (c.amode == 0) {
    if (bias)
      return accd ? launch_tile<0, true, true>(ctx, a, kb, ops, bytes) : launch_tile<0, true, false>(ctx, a, kb, ops, bytes);
    return accd ? launch_tile<0, false, true>(ctx, a, kb, ops, bytes) : launch_tile<0, false, false>(ctx, a, kb, ops, bytes);
  }
  return accd ? launch_tile<1, false, true>(ctx, a, kb, ops, bytes) : launch_tile<1, false, false>(ctx, a, kb, ops, bytes);
}

// the Linear epilogue over [slices][M][N] INT32 partial slabs (slice 0 carries oc'), for i8ie_skinny.hip
int i8ie_launch_splitk_reduce(i8ie_ctx* ctx, const int32_t* partial, int slices, int M, int N, const float* biasf,
                              float s_in, float s_w, float s_out, int zp_out, int relu, uint8_t* out, int32_t* acc) {
  const Requant rq = i8ie_make_requant(s_in, s_w, s_out, zp_out);
  I8ieProfScope prof(ctx, "splitk_reduce", 0.0, 4.0 * slices * M * N + (double)M * N);
  const int64_t quads = ((int64_t)M * N + 3) / 4;
  splitk_reduce_kernel<<<cap_grid(quads, 256), 256, 0, ctx->stream>>>(partial, slices, M, N, biasf, rq,
                                                                      relu ? zp_out : 0, out, acc);
  I8IE_LAUNCH_CHECK();
  return I8IE_OK;
}

int i8ie_launch_finish_offsets(i8ie_ctx* ctx, const int32_t* oc, const int32_t* wsum, const int8_t* qb, float s_in,
                               int n, int32_t* ocp, float* biasf) {
  finish_offsets_kernel<<<(n + 63) / 64, 64, 0, ctx->stream>>>(oc, wsum, qb, s_in, n, ocp, biasf);
  I8IE_LAUNCH_CHECK();
  return I8IE_OK;
}

// NCHW [n][c][h][w]  <->  NHWC [n][h + 2b][w + 2b][c] (border bytes untouched when writing)
int i8ie_launch_nchw_to_nhwc(i8ie_ctx* ctx, const uint8_t* in, uint8_t* out, int n, int c, int h, int w, int b) {
  I8ieProfScope prof(ctx, "layout_nchw_to_nhwc", 0.0, 2.0 * n * (double)c * h * w);
  dim3 grid((h * w + 63) / 64, (c + 63) / 64, n);
  transpose_u8_kernel<true><<<grid, 256, 0, ctx->stream>>>(in, out, c, h * w, w, w + 2 * b, b,
                                                           (int64_t)(h + 2 * b) * (w + 2 * b));
  I8IE_LAUNCH_CHECK();
  return I8IE_OK;
}
int i8ie_launch_nhwc_to_nchw(i8ie_ctx* ctx, const uint8_t* in, uint8_t* out, int n, int c, int h, int w, int b) {
  I8ieProfScope prof(ctx, "layout_nhwc_to_nchw", 0.0, 2.0 * n * (double)c * h * w);
  dim3 grid((h * w + 63) / 64, (c + 63) / 64, n);
  transpose_u8_kernel<false><<<grid, 256, 0, ctx->stream>>>(in, out, c, h * w, w, w + 2 * b, b,
                                                            (int64_t)(h + 2 * b) * (w + 2 * b));
  I8IE_LAUNCH_CHECK();
  return I8IE_OK;
}

int i8ie_launch_reborder(i8ie_ctx* ctx, const uint8_t* in, uint8_t* out, int n, int c, int h, int w, int ib, int ob,
                         int zp) {
  const int64_t total = (int64_t)n * (h + 2 * ob) * (w + 2 * ob) * (c / 16);
  I8ieProfScope prof(ctx, "reborder_u8_nhwc", 0.0, (double)n * c * h * w + 16.0 * total);
  reborder_u8_kernel<<<cap_grid(total, 256), 256, 0, ctx->stream>>>(in, out, total, h, w, c / 16, ib, ob,
                                                                    (uint32_t)(zp & 0xFF) * 0x01010101u);
  I8IE_LAUNCH_CHECK();
  return I8IE_OK;
}

// border of a bordered NHWC u8 tensor := zp (c % 16 == 0; otherwise the caller memsets the whole buffer)
int i8ie_launch_fill_border(i8ie_ctx* ctx, uint8_t* out, int n, int c, int h, int w, int b, int zp) {
  if (b <= 0) return I8IE_OK;
  if (c % 16 != 0 || (reinterpret_cast<uintptr_t>(out) & 15u) != 0) {
    I8IE_HIP_TRY(hipMemsetAsync(out, zp & 0xFF, (size_t)n * (h + 2 * b) * (w + 2 * b) * c, ctx->stream));
    return I8IE_OK;
  }
  const int nb = 2 * b * (w + 2 * b) + h * 2 * b;
  const int64_t total = (int64_t)n * nb * (c / 16);
  I8ieProfScope prof(ctx, "fill_border_u8", 0.0, 16.0 * total);
  fill_border_kernel<<<cap_grid(total, 256), 256, 0, ctx->stream>>>(out, total, h, w, c / 16, b,
                                                                    (uint32_t)(zp & 0xFF) * 0x01010101u);
  I8IE_LAUNCH_CHECK();
  return I8IE_OK;
}

struct I8ieSmallNCall {
  const uint8_t* A;
  size_t lda;
  int M, K;
  const int8_t* B;
  int Kpad, N;
  const int32_t* ocp;
  const float* biasf;
  float s_in, s_w, s_out;
  int zp_out, relu;
  uint8_t* out;
  int32_t* acc;
  float* out_f32;
};
int i8ie_smalln_max_features() { return kSmallN; }
int i8ie_launch_linear_smalln(i8ie_ctx* ctx, const I8ieSmallNCall& c) {
  I8IE_REQUIRE(c.M > 0 && c.N > 0 && c.N <= kSmallN && c.K > 0 && c.K % 16 == 0 && c.lda % 16 == 0, "small-N linear shape");
  I8IE_REQUIRE((reinterpret_cast<uintptr_t>(c.A) & 15u) == 0 && (reinterpret_cast<uintptr_t>(c.B) & 15u) == 0,
               "operands must be 16-byte aligned");
  SmallNArgs a{};
  a.A = c.A; a.lda = c.lda; a.M = c.M; a.K = c.K; a.B = c.B; a.Kpad = c.Kpad; a.N = c.N;
  a.ocp = c.ocp; a.biasf = c.biasf;
  a.rq = i8ie_make_requant(c.s_in, c.s_w, c.s_out, c.zp_out);
  a.rq.fast = I8IE_RQ_EXACT;
  a.relu_lo = c.relu ? c.zp_out : 0;
  a.out = c.out; a.acc = c.acc; a.out_f32 = c.out_f32; a.dq_scale = c.s_out; a.dq_zp = c.zp_out;
  I8ieProfScope prof(ctx, "linear_smalln_dot4", 2.0 * c.M * c.N * c.K, (double)c.M * c.K + (double)c.N * c.K + 5.0 * c.M * c.N);
  // (activation rows are lda = K bytes apart and the weights are zero from K to Kpad: a step past K multiplies
  // the next row's bytes, or the zeros the descriptor returns past the buffer, by zero)
  if (c.K >= 1024 && c.Kpad % 64 == 0 && (size_t)16 * c.lda + c.Kpad < ((size_t)1 << 31) && ctx->variant != 11)
    linear_head_mfma_kernel<<<(c.M + 15) / 16, 512, 0, ctx->stream>>>(a);
  else if (c.K >= 2048)
    linear_smalln_kernel<4><<<c.M, 256, 0, ctx->stream>>>(a);
  else
    linear_smalln_kernel<1><<<(c.M + 3) / 4, 256, 0, ctx->stream>>>(a);
  I8IE_LAUNCH_CHECK();
  return I8IE_OK;
}

int i8ie_launch_permute_k(i8ie_ctx* ctx, const int8_t* B, int8_t* Bp, int rows, int Kpad, int K, int c, int hw) {
  const int64_t total = (int64_t)rows * Kpad;
  permute_k_kernel<<<cap_grid(total, 256), 256, 0, ctx->stream>>>(B, Bp, total, Kpad, K, c, hw);
  I8IE_LAUNCH_CHECK();
  return I8IE_OK;
}

int i8ie_launch_repack_smallc(i8ie_ctx* ctx, const uint8_t* in, uint8_t* out, int n, int c, int h, int w, int Hp,
                              int Wg, int ph, int pw, int zp, int rebias) {
  const int64_t total = (int64_t)n * Hp * Wg;
  I8ieProfScope prof(ctx, "repack_smallc_u8", 0.0, (double)n * c * h * w + 16.0 * total);
  repack_smallc_kernel<<<cap_grid(total, 256), 256, 0, ctx->stream>>>(in, out, total, c, h, w, Hp, Wg, ph, pw,
                                                                      (uint32_t)(zp & 0xFF),
                                                                      rebias ? 0x80808080u : 0u);
  I8IE_LAUNCH_CHECK();
  return I8IE_OK;
}

int i8ie_launch_maxpool_nhwc(i8ie_ctx* ctx, const uint8_t* in, int ib, uint8_t* out, int ob, int n, int c, int h,
                             int w, int k, int s, int relu_zp) {
  const int oh = (h - k) / s + 1, ow = (w - k) / s + 1;
  const int c16 = c / 16;
  const int64_t per_img = (int64_t)oh * ow * c16;
  I8IE_REQUIRE(per_img > 0 && per_img < ((int64_t)1 << 31), "max-pool image too large");
  // 32-bit chunk indices inside the kernel: batches beyond 2^31 - 1 chunks run as several launches
  const int imgs_per = (int)((((int64_t)1 << 31) - 1) / per_img);
  for (int i0 = 0; i0 < n; i0 += imgs_per) {
    const int nb = n - i0 < imgs_per ? n - i0 : imgs_per;
    const int64_t total = (int64_t)nb * per_img;
    const uint8_t* src = in + (size_t)i0 * (h + 2 * ib) * (w + 2 * ib) * c;
    uint8_t* dst = out + (size_t)i0 * (oh + 2 * ob) * (ow + 2 * ob) * c;
    I8ieProfScope prof(ctx, "maxpool_u8_nhwc", 0.0, (double)nb * c * h * w + 16.0 * total);
    const int grid = (cap_grid(total, 256, 256 * 32) + 7) / 8 * 8;
    const uint32_t lo4 = (uint32_t)(relu_zp & 0xFF) * 0x01010101u;
#define I8IE_POOL_LAUNCH(KT)                                                                                        \
  maxpool_u8_nhwc_kernel<KT><<<grid, 256, 0, ctx->stream>>>(src, dst, (uint32_t)total, h + 2 * ib, w + 2 * ib, ib, \
                                                            c16, oh, ow, k, s, oh + 2 * ob, ow + 2 * ob, ob, lo4)
    if (k == 3) I8IE_POOL_LAUNCH(3);
    else if (k == 2) I8IE_POOL_LAUNCH(2);
    else I8IE_POOL_LAUNCH(0);
#undef I8IE_POOL_LAUNCH
    I8IE_LAUNCH_CHECK();
  }
  return I8IE_OK;
}
