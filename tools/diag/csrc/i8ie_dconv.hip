// i8ie_dconv.hip -- patch-stationary Conv2d with the requantiser INSIDE the K loop (deferred epilogue, two accumulator sets).
//
//   C[r][j] = sum_k A_u8[r][k] * W_s8[j][k] + oc[j]     (src/conv2d.cc:131-133: cblas_gemm_s8u8s32 + oc)
//   out     = max_pool?(relu?(down_scale(C)))           (src/quantize_utils.cc:27-36, src/functional.cc:15-64)
//
// Why (round 3's measurements): in i8ie_pconv.hip a band pass of conv2 takes ~62 k cycles for 38.9 k
// cycles of MFMA work: the epilogue (10 k, ~7 vector instructions per output value) and the hand-over sit behind the K loop
// with the matrix pipe idle, and the epilogue cannot be given to another wave: on a gfx950 SIMD a wave's vector instructions
// do not issue beside ANOTHER wave's back-to-back MFMAs (tools/valu_probe), they do issue in the shadow of the SAME wave's
// MFMAs (3 per v_mfma_i32_32x32x32_i8 for free).  So here:
//   * ONE wave per SIMD (256-thread blocks, one per CU, 512 registers per lane), 2 x 2 waves over (pixels, features).  A wave
//     owns TM tiles of 32 pixels x 2 halves x NT tiles of 32 features: two accumulator sets (half 0 in VGPRs, half 1 in
//     AGPRs; 128 + 128 registers for conv2's 128 x 128 wave tile).  The block tile is the same 256 pixels x 256 (384)
//     features as i8ie_pconv.hip's, and so is the operand traffic per MFMA (each half is a 128 x 64 wave tile).
//   * A band runs as two K loops over the resident patch (half 0, then half 1).  While one set accumulates, the OTHER set --
//     the previous half's finished sums -- is requantised by vector instructions placed between this wave's own MFMAs
//     (R per MFMA, a static schedule: every instruction of the K loop's first KF K tiles is an `asm volatile` statement, so
//     the order written is the order issued): + oc', int -> float, fma, (relu), v_cvt_pk_u8_f32, the 2^-13 guard of
//     i8ie_requant.h (fract / sub / min3), then per 32 x 32 tile the guard branches (exact replay), two v_permlane32_swap
//     and ONE 16-byte store (global, or the LDS ring of conv rows when a max-pool follows).  The last half of a block's last
//     band is requantised after its K loop, exposed.
//   * v_mfma_i32_32x32x32_i8 (a single wave issues them back to back at the pipe's rate; 16 x 16 x 64 measured 18 cycles per
//     16).  Accumulators start from the inline constant 0 in a tile's first MFMA; oc' joins in the requantiser.
//   * Everything else is i8ie_pconv.hip's: the patch of a band copied into LDS once by LDS-DMA (pixel pitch C + 16), fragment
//     address = base(pixel) + offset(K chunk), weights pre-packed in fragment order and read straight from L2 one K tile
//     ahead, bands of an image back to back when a pool follows (`seq`), XCD-contiguous units, the pool pass over the LDS
//     ring.  With a pool the two halves' bytes reach the ring half a band apart: half 0's features of band i and half 1's
//     features of band i - 1 are pooled in the hand-over of band i, under the landing patch.
// Takes: feature passes of 256 or 384 (N = 384 with at most 192 pixels per band), at least KF K tiles, a unit per band (no
// (band, pass) split: small batches stay with i8ie_pconv.hip), a pool only with one feature pass.
#include <cstdio>
#include <cstdlib>
#include <type_traits>
#include <utility>
#include <vector>

#include "i8ie_pconv_common.h"

#ifndef DC_EXP
#define DC_EXP 0
#endif

namespace {

typedef int v16i __attribute__((ext_vector_type(16)));
#if (DC_EXP + 0) & 4
#define DC_NOP ""
#else
#define DC_NOP "s_nop 1\n\t"
#endif

template <int B, typename F, int... Is>
__device__ __forceinline__ void dc_static_for_impl(F&& f, std::integer_sequence<int, Is...>) {
  (f(std::integral_constant<int, B + Is>{}), ...);  // (a fold, not a recursion: the drain's program is > 1000 statements)
}
template <int B, int E, typename F>
__device__ __forceinline__ void dc_static_for(F&& f) {
  if constexpr (B < E) dc_static_for_impl<B>(f, std::make_integer_sequence<int, E - B>{});
}

// ---- the requantiser as a static program of single vector instructions ------------------------------------------------
// Per output value (i8ie_requant_pack4's fast path, instruction for instruction): [v_accvgpr_read], v_add_u32 (+ oc'),
// v_cvt_f32_i32, v_fma_f32, [v_max_f32], v_cvt_pk_u8_f32, v_fract_f32, v_add_f32 -0.5; per pair of values one v_min3_f32.
// Two values run side by side (their chains alternate, so no instruction waits for the one before it).
constexpr int dc_ops_per_value(bool relu, bool agpr) { return 6 + (relu ? 1 : 0) + (agpr ? 1 : 0); }
constexpr int dc_ops_per_unit(bool relu, bool agpr) { return 8 * (2 * dc_ops_per_value(relu, agpr) + 1); }  // a unit = one 32 x 32 tile
#ifndef DC_EXP
#define DC_EXP 0  // timing experiments (tools/dbg/build_dconv_exp.sh; results are WRONG with any bit set): 1 = no exact-replay code,
#endif            // 2 = no requantiser in the K loops at all, 4 = no s_nop in front of the MFMAs, 8 = 2 instead of 3 instructions per MFMA,
// 16 = phase stamps, 32 = weight fragments fetched for every second K tile only (confounded by its branch), 64 = no weight
// fetches in the K loops, 128 = no activation fragment reads in the K loops (what do the operand streams cost at the power cap?)
constexpr int kDcOpsPerSlot = (DC_EXP & 8) ? 2 : 3;  // vector instructions behind every MFMA (tools/valu_probe: 3 are free beside a 32x32x32)
// K tiles (of 4 quarters x TM x NT MFMAs) the requantiser of the other half is spread over
constexpr int dc_fill_ktiles(int TM, int NT, bool relu) {
  const int ops = TM * NT * dc_ops_per_unit(relu, true);  // (the larger of the two halves' programs: the AGPR one)
  const int slots = (ops + kDcOpsPerSlot - 1) / kDcOpsPerSlot;
  return (slots + 4 * TM * NT - 1) / (4 * TM * NT);
}

// TM: 32-pixel tiles per wave (2 waves along the pixels); NT: 32-feature tiles per wave and half (2 waves x 2 halves along
// the features: a pass is 128 NT features).  ACC: also dump the INT32 accumulators.  POOL: max-pool behind the requantiser.
template <int TM, int NT, bool ACC, bool POOL>
__global__ __launch_bounds__(256, 1) void dconv_kernel(PCArgs p) {
  uint8_t* const smem = pc_smem;
  constexpr int BN = NT * 128;
  constexpr int NT32 = BN / 32;           // fragments of 32 features per quarter of a K tile
  constexpr int KT_BYTES = BN * 128;      // weights of one K tile of a pass: 4 quarters x NT32 fragments of 1 KiB
  constexpr int U = TM * NT;              // 32 x 32 tiles (fill units) per half
  constexpr bool RELU = !POOL;            // (with a pool the ReLU is the pool pass's initial maximum)
  constexpr int KF = dc_fill_ktiles(TM, NT, RELU);
  constexpr int SK = 4 * U;               // MFMA slots per K tile
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int hh = lane >> 5, l31 = lane & 31;

  // ---- units of this block: XCD-contiguous ranges, consecutive units to the blocks of one XCD (seq: a unit is an image)
  const int per = (int)gridDim.x >> 3;
  const int xcd = blockIdx.x & 7, jb = blockIdx.x >> 3;
  const int n_units = p.seq ? p.n_tiles / p.bands : p.n_tiles;
  const int Tx = (n_units + 7) >> 3;
  const int t_lo = xcd * Tx;
  const int t_hi = t_lo + Tx < n_units ? t_lo + Tx : n_units;
  int unit = t_lo + jb;
  if (unit >= t_hi) return;

  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(p.A), 0, p.a_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<int8_t*>(p.Bf), 0, p.bf_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsO = __builtin_amdgcn_make_buffer_rsrc(p.out, 0, p.out_bytes, 0x00020000);

  // ---- the patch of a band -> LDS (as i8ie_pconv.hip; 256 threads: 4 pieces of 1 KiB per round)
  const int CC1 = p.CC + 1;
  auto src_of = [&](int g) {
    int row, rem, pix, ch;
    pc_divmod(g, p.row_gran, p.rcpRowGran, row, rem);
    pc_divmod(rem, CC1, p.rcpCC1, pix, ch);
    if (pix >= p.Wp) pix = 0;  // (row padding: any readable bytes)
    return (unsigned)row * p.row_pitch + (unsigned)pix * p.C + (unsigned)(ch < p.CC ? ch : 0) * 16u;
  };
  auto tile_src0 = [&](int t) {
    const int img = t / p.bands, band = t - img * p.bands;
    return (unsigned)img * p.img_pitch + (unsigned)(band * p.RT * p.s) * p.row_pitch;
  };
  auto first_fill = [&](int t, int dst) {  // the block's first patch; leaves the source-offset table behind
    const unsigned src0 = tile_src0(t);
    for (int g0 = 0; g0 < p.patch_gran; g0 += 256) {
      const unsigned rel = src_of(g0 + tid);
      if (p.lds_src >= 0) reinterpret_cast<unsigned*>(smem + p.lds_src)[g0 + tid] = rel;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (__attribute__((address_space(3))) void*)(smem + dst + (g0 + wave * 64) * 16), 16,
                                               (int)(src0 + rel), 0, 0, 0);
    }
  };
  auto patch_fill = [&](int t, int dst) {
    const unsigned src0 = tile_src0(t);
    for (int g0 = 0; g0 < p.patch_gran; g0 += 256) {
      const unsigned so = src0 + (p.lds_src >= 0 ? reinterpret_cast<const unsigned*>(smem + p.lds_src)[g0 + tid] : src_of(g0 + tid));
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (__attribute__((address_space(3))) void*)(smem + dst + (g0 + wave * 64) * 16), 16,
                                               (int)so, 0, 0, 0);
    }
  };
  first_fill(p.seq ? unit * p.bands : unit, p.lds_patch);

  // ---- tables in LDS: oc', per pixel of a band its window origin in the patch / its output offset, per (K tile, lane half)
  //      the patch offsets of the four quarters' 16-byte K chunks
  for (int i = tid; i < p.npass * BN; i += 256) reinterpret_cast<int*>(smem + p.lds_ocp)[i] = i < p.Npad ? p.ocp[i] : 0;
  const int PT = p.RT * p.OW;
  for (int i = tid; i < kTabPix; i += 256) {
    int oy, ox;
    pc_divmod(i < PT ? i : 0, p.OW, p.rcpOW, oy, ox);
    reinterpret_cast<unsigned*>(smem + p.lds_tab)[i] = (unsigned)(oy * p.s) * (unsigned)(p.row_gran * 16) + (unsigned)(ox * p.s) * (unsigned)p.P;
    reinterpret_cast<unsigned*>(smem + p.lds_tab)[kTabPix + i] = (unsigned)(oy * p.OWp + ox) * (unsigned)p.N;
  }
  for (int i = tid; i < p.nkt * 8; i += 256) {
    // entry (kt, lane half h2, quarter q4 = 2 ks + kh): chunk q = 2 kh + h2 of k-step ks = K position 8 kt + 2 q + ks
    const int kt = i >> 3, h2 = (i >> 2) & 1, q4 = i & 3, ks = q4 >> 1, kh = q4 & 1;
    const int ci = 8 * kt + 2 * (2 * kh + h2) + ks;
    const int sc = p.perm[ci];
    int tap, cc, kh_, kw_;
    pc_divmod(sc < 0 ? 0 : sc, p.CC, 1.0f / (float)p.CC, tap, cc);
    pc_divmod(tap, p.KW, 1.0f / (float)p.KW, kh_, kw_);
    reinterpret_cast<unsigned*>(smem + p.lds_ktab)[i] = sc >= 0 ? (unsigned)kh_ * (unsigned)(p.row_gran * 16) + (unsigned)kw_ * (unsigned)p.P + (unsigned)cc * 16u : 0u;
  }
  __syncthreads();
  // this lane's pixels: tile (wm TM + i), row l31 (consecutive pixels: the 16 lanes one ds_read_b128 cycle serves lie in one
  // lane half, read the same K chunk of 16 pixels that are distinct mod 16, and the pixel pitch is an odd number of 16-byte
  // slots: conflict-free)
  const int pix0 = wm * TM * 32 + l31;
  unsigned abase[TM];
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int pi = pix0 + i * 32;
    abase[i] = (unsigned)p.lds_patch + reinterpret_cast<const unsigned*>(smem + p.lds_tab)[pi < kTabPix && pi < PT ? pi : 0];
  }

  auto patch_xor = [&](int dst) {  // every lane re-biases exactly the granules it fetched itself
    if (p.a_s8) return;
    for (int g = tid; g < p.patch_gran; g += 256) {
      v4i* q = reinterpret_cast<v4i*>(smem + dst + g * 16);
      *q = *q ^ (int)0x80808080;
    }
  };

  v16i accV_[TM][NT], accA_[TM][NT];  // half 0: VGPRs, half 1: AGPRs (the asm constraints below decide)
  const I8ieRequant rq = p.rq;
  const int lo = p.relu_lo;
  const float lof = (float)lo;
  const float rq_bias = rq.zpf - 0.5f;
  const float rq_w0 = rq.fast ? 1.0f : 0.0f;  // (scales that do not allow the estimate: every dword takes the exact sequence)

  // ---- MFMA statements.  srcA = weights (rows of D = features), srcB = activations (columns of D = pixels): lane (hh, l31)
  //      holds pixel l31 and, in register 4 g + r, feature 8 g + 4 hh + r of the tile
  auto mfma = [&](auto HC, auto FIRSTC, auto ic, auto nc, const v4i& b, const v4i& a) {
    constexpr int H = decltype(HC)::value, i = decltype(ic)::value, n = decltype(nc)::value;
    constexpr bool FIRST = decltype(FIRSTC)::value;
    auto& accV = accV_;  // (named outside the `if constexpr`s: clang captures a variable of the enclosing function only where a
    auto& accA = accA_;  //  use outside a discarded statement names it)
    // (the compiler sees one opaque statement: the early-clobber keeps the zero-initialising form's destination off its
    //  operands -- a destination that overlaps srcA / srcB is undefined -- and the leading s_nop 1 covers a fragment register
    //  the compiler itself restored with a vector instruction just before: it pads no hazards for instructions inside an asm)
    if constexpr (H == 0) {
      if constexpr (FIRST) asm volatile(DC_NOP "v_mfma_i32_32x32x32_i8 %0, %1, %2, 0" : "=&v"(accV[i][n]) : "v"(b), "v"(a));
      else asm volatile(DC_NOP "v_mfma_i32_32x32x32_i8 %0, %1, %2, %0" : "+v"(accV[i][n]) : "v"(b), "v"(a));
    } else {
      if constexpr (FIRST) asm volatile(DC_NOP "v_mfma_i32_32x32x32_i8 %0, %1, %2, 0" : "=&a"(accA[i][n]) : "v"(b), "v"(a));
      else asm volatile(DC_NOP "v_mfma_i32_32x32x32_i8 %0, %1, %2, %0" : "+a"(accA[i][n]) : "v"(b), "v"(a));
    }
  };

  // ---- fill context: the tile / pass whose sums the OTHER accumulator set holds
  struct Ctx {
    int tile, pass;
  };
  // state of the requantiser program (one 32 x 32 tile at a time)
  int ti0, ti1;
  float tf0, tf1, e0, e1, f0, f1;
  unsigned pk[4];
  float worst[4];
  v4i oc4[4];  // oc' of this lane's 16 features of the tile's feature tile n
  int fctx_obase = 0, fctx_valid = 0, fctx_oy0 = 0, fctx_img = 0, fctx_n0 = 0;  // per-phase scalars of the fill context
  auto fill_begin = [&](auto SRCC, const Ctx& c) {  // SRC = the half (accumulator set) being requantised
    constexpr int SRC = decltype(SRCC)::value;
    const int img = c.tile / p.bands, band = c.tile - img * p.bands;
    fctx_img = img;
    fctx_oy0 = band * p.RT;
    const int rows = p.OH - fctx_oy0 < p.RT ? p.OH - fctx_oy0 : p.RT;
    fctx_valid = rows * p.OW;
    fctx_obase = (int)(((unsigned)(img * p.OHp + fctx_oy0 + p.ob) * (unsigned)p.OWp + (unsigned)p.ob) * (unsigned)p.N);
    fctx_n0 = c.pass * BN + (SRC * 2 + wn) * (NT * 32);
  };
  auto load_oc = [&](auto nc) {
    constexpr int n = decltype(nc)::value;
#pragma unroll
    for (int g = 0; g < 4; ++g) oc4[g] = *reinterpret_cast<const v4i*>(smem + p.lds_ocp + (fctx_n0 + n * 32 + 8 * g + 4 * hh) * 4);
  };
  // unit u = n TM + i (feature tile major: oc' changes NT times per half)
  // one instruction of the program: op index o inside unit u
  auto fill_op = [&](auto SRCC, auto uc, auto oc_) {
    constexpr int SRC = decltype(SRCC)::value, u = decltype(uc)::value, o = decltype(oc_)::value;
    constexpr int n = u / TM, i = u % TM;
    constexpr bool AG = SRC == 1;
    auto& accV = accV_;
    auto& accA = accA_;
    auto& worst_ = worst;
    auto& pk_ = pk;
    (void)worst_; (void)pk_; (void)accV; (void)accA;
    constexpr int OPV = dc_ops_per_value(RELU, AG), PP = 2 * OPV + 1;
    constexpr int pr = o / PP, w = o % PP;  // pair of values, position inside the pair's 2 OPV + 1 instructions
    constexpr int g = pr / 2;               // dword (4 values = 2 pairs)
    if constexpr (w == 2 * OPV) {
      if constexpr ((pr & 1) == 0) asm volatile("v_min3_f32 %0, %1, |%2|, |%3|" : "=v"(worst[g]) : "v"(rq_w0), "v"(f0), "v"(f1));
      else asm volatile("v_min3_f32 %0, %0, |%1|, |%2|" : "+v"(worst[g]) : "v"(f0), "v"(f1));
    } else {
      constexpr int which = w & 1, st0 = w >> 1;           // value 2 pr + which, step st0 of its chain
      constexpr int v = 2 * pr + which, r = v & 3;          // register 4 g + r of the tile
      constexpr int step = AG ? st0 : st0 + 1;              // 0 RD, 1 ADD, 2 CVT, 3 FMA, 4 MAX / PK, ...
      int& ti = which ? ti1 : ti0;
      float& tf = which ? tf1 : tf0;
      float& e = which ? e1 : e0;
      float& f = which ? f1 : f0;
      const int ocv = oc4[g][r];
      if constexpr (step == 0) {
        asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(ti) : "a"(accA[i][n][4 * g + r]));
      } else if constexpr (step == 1) {
        if constexpr (AG) asm volatile("v_add_u32 %0, %0, %1" : "+v"(ti) : "v"(ocv));
        else asm volatile("v_add_u32 %0, %1, %2" : "=v"(ti) : "v"(accV[i][n][4 * g + r]), "v"(ocv));
      } else if constexpr (step == 2) {
        asm volatile("v_cvt_f32_i32 %0, %1" : "=v"(tf) : "v"(ti));
      } else if constexpr (step == 3) {
        asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(e) : "v"(tf), "s"(rq.ms), "v"(rq_bias));
      } else if constexpr (RELU && step == 4) {
        asm volatile("v_max_f32 %0, %1, %2" : "=v"(tf) : "s"(lof), "v"(e));
      } else if constexpr (step == (RELU ? 5 : 4)) {
        if constexpr (RELU) {
          if constexpr (r == 0) asm volatile("v_cvt_pk_u8_f32 %0, %1, 0, 0" : "=v"(pk[g]) : "v"(tf));
          else asm volatile("v_cvt_pk_u8_f32 %0, %1, %2, %0" : "+v"(pk[g]) : "v"(tf), "n"(r));
        } else {
          if constexpr (r == 0) asm volatile("v_cvt_pk_u8_f32 %0, %1, 0, 0" : "=v"(pk[g]) : "v"(e));
          else asm volatile("v_cvt_pk_u8_f32 %0, %1, %2, %0" : "+v"(pk[g]) : "v"(e), "n"(r));
        }
      } else if constexpr (step == (RELU ? 6 : 5)) {
        asm volatile("v_fract_f32 %0, %1" : "=v"(f) : "v"(e));
      } else {
        asm volatile("v_add_f32 %0, -0.5, %0" : "+v"(f));
      }
    }
  };
  // a unit's tail: guard branches (exact replay of a dword whose closest value lies within 2^-13 of a rounding boundary),
  // lanes l and l + 32 exchange dwords so that each holds 16 consecutive features of its pixel, one 16-byte store
  auto fill_finish = [&](auto SRCC, auto uc) {
    constexpr int SRC = decltype(SRCC)::value, u = decltype(uc)::value;
    constexpr int n = u / TM, i = u % TM;
    auto& accV = accV_;
    auto& accA = accA_;
    (void)accV; (void)accA;
    const int pi = pix0 + i * 32;
    const bool valid = pi < fctx_valid;
    int cc[16];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      if ((!(DC_EXP & 1) && !i8ie_requant_est_ok(worst[g])) || ACC) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          if constexpr (SRC == 0) cc[4 * g + r] = accV[i][n][4 * g + r] + oc4[g][r];
          else cc[4 * g + r] = accA[i][n][4 * g + r] + oc4[g][r];
        }
      }
      if (!(DC_EXP & 1) && !i8ie_requant_est_ok(worst[g])) {
        const int c4[4] = {cc[4 * g], cc[4 * g + 1], cc[4 * g + 2], cc[4 * g + 3]};
        pk[g] = i8ie_requant_exact4(c4, rq, POOL ? 0 : lo);
      }
    }
    const int n0 = fctx_n0 + n * 32;
    if constexpr (ACC) {  // row = image-major pixel index (bands are whole rows); 4 consecutive features per register quad
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int col = n0 + 8 * g + 4 * hh;
        if (valid && col < p.N) {
          v4i v;
          v.x = cc[4 * g]; v.y = cc[4 * g + 1]; v.z = cc[4 * g + 2]; v.w = cc[4 * g + 3];
          *reinterpret_cast<v4i*>(p.acc + ((size_t)fctx_img * (size_t)(p.OH * p.OW) + (size_t)(fctx_oy0 * p.OW + pi)) * (size_t)p.N + col) = v;
        }
      }
    }
    asm volatile("s_nop 1");  // (v_cvt_pk_u8_f32 above -> v_permlane32_swap: the hazard recognizer does not see into asm)
    // lower lanes: own dword 0 (features 0-3), the upper lane's dword 0 (4-7), own dword 1 (8-11), the upper lane's dword 1
    // (12-15); upper lanes: the lower lane's dwords 2, 3 and their own (features 16-31)
    const auto s02 = __builtin_amdgcn_permlane32_swap(pk[0], pk[2], false, false);
    const auto s13 = __builtin_amdgcn_permlane32_swap(pk[1], pk[3], false, false);
    v4u val;
    val.x = s02[0]; val.y = s02[1]; val.z = s13[0]; val.w = s13[1];
    const int col = n0 + 16 * hh;
    if constexpr (POOL) {
      int L = (fctx_oy0 % p.RB) * p.OW + pi;
      if (L >= p.RB * p.OW) L -= p.RB * p.OW;
      if (valid && col < p.N) *reinterpret_cast<v4u*>(smem + p.lds_otile + L * p.opitch + col) = val;
    } else {
      val ^= p.xor_out;
      const unsigned rowoff = (unsigned)fctx_obase + reinterpret_cast<const unsigned*>(smem + p.lds_tab)[kTabPix + (pi < kTabPix ? pi : 0)];
      __builtin_amdgcn_raw_buffer_store_b128(val, rsO, (int)((valid && col < p.N) ? rowoff + (unsigned)col : kRowInvalid), 0, 0);
    }
  };
  // the program's instructions [o_lo, o_hi) (global index: unit * OPU + o), with the tails of the units they complete
  auto fill_range = [&](auto SRCC, auto loc, auto hic) {
    constexpr int SRC = decltype(SRCC)::value;
    constexpr int OPU = dc_ops_per_unit(RELU, SRC == 1);
    constexpr int o_lo = decltype(loc)::value, o_hi0 = decltype(hic)::value;
    constexpr int o_hi = o_hi0 < U * OPU ? o_hi0 : U * OPU;
    dc_static_for<o_lo, o_hi>([&](auto oc_) {
      constexpr int og = decltype(oc_)::value;
      constexpr int u = og / OPU, o = og % OPU;
      if constexpr (o == 0 && u % TM == 0) load_oc(std::integral_constant<int, u / TM>{});
      fill_op(SRCC, std::integral_constant<int, u>{}, std::integral_constant<int, o>{});
      if constexpr (o == OPU - 1) fill_finish(SRCC, std::integral_constant<int, u>{});
    });
  };

  // ---- operand fragments
  v4i A0[TM], A1[TM], Bq[4][NT];
  v4u kq, kqn;  // patch offsets of the four quarters' chunks (this lane half) for the current / the next K tile
  auto load_k = [&](int kt) { return *reinterpret_cast<const v4u*>(smem + p.lds_ktab + (kt * 2 + hh) * 16); };
  auto load_A = [&](v4i (&dst)[TM], unsigned koff) {
#pragma unroll
    for (int i = 0; i < TM; ++i) dst[i] = *reinterpret_cast<const v4i*>(smem + abase[i] + koff);
  };
  // weight fragments of quarter q4 of the K tile at byte offset `kb` (pass, K tile, half and wave column folded in)
  auto bbase = [&](int pass, int kt, int h) {
    return (unsigned)(pass * p.nkt + kt) * (unsigned)KT_BYTES + (unsigned)(((h * 2 + wn) * NT) * 1024 + lane * 16);
  };
  auto load_B = [&](v4i (&dst)[NT], unsigned kb, int q4) {
#pragma unroll
    for (int n = 0; n < NT; ++n)
      dst[n] = __builtin_bit_cast(v4i, __builtin_amdgcn_raw_buffer_load_b128(rsB, (int)(kb + (unsigned)((q4 * NT32 + n) * 1024)), 0, 0));
  };

  // One quarter of a K tile: TM x NT MFMAs on the fragments (a, b), feature tile major (a weight fragment is free for its
  // refill after TM MFMAs); FB >= 0: the requantiser instructions of slots FB ...; behind every MFMA one operand request
  // `job(slot)`, fenced so that it stays there: with ONE wave per SIMD nothing else fills the matrix pipe while this wave
  // issues address adds, LDS reads and buffer loads -- clustered at the quarter's edge (8 + 4 + 2 instructions) they cost
  // ~80 of every 256 + 80 cycles (phase stamps of the first build: K loop at 76 % of the MFMA rate)
  auto quarter = [&](auto HC, auto FIRSTC, auto FBC, const v4i (&a)[TM], const v4i (&b)[NT], auto&& job) {
    constexpr int H = decltype(HC)::value, FB = decltype(FBC)::value;
    dc_static_for<0, NT>([&](auto nc) {
      dc_static_for<0, TM>([&](auto ic) {
        constexpr int sl = decltype(nc)::value * TM + decltype(ic)::value;
        mfma(HC, FIRSTC, ic, nc, b[decltype(nc)::value], a[decltype(ic)::value]);
        if constexpr (FB >= 0) {
          constexpr int s = FB + sl;
          fill_range(std::integral_constant<int, 1 - H>{}, std::integral_constant<int, s * kDcOpsPerSlot>{},
                     std::integral_constant<int, (s + 1) * kDcOpsPerSlot>{});
        }
        __builtin_amdgcn_sched_barrier(0);
        job(std::integral_constant<int, sl>{});
        __builtin_amdgcn_sched_barrier(0);
      });
    });
  };
  auto load_B1 = [&](v4i& dst, unsigned kb, int q4, int n) {
    if ((DC_EXP & 32) && ((kb / (unsigned)KT_BYTES) & 1u)) return;  // (timing experiment: every second K tile reuses stale weights)
    if constexpr ((DC_EXP & 64) != 0) return;                        // (timing experiment: no weight fetches inside the K loops at all)
    dst = __builtin_bit_cast(v4i, __builtin_amdgcn_raw_buffer_load_b128(rsB, (int)(kb + (unsigned)((q4 * NT32 + n) * 1024)), 0, 0));
  };
  // One K tile.  On entry: A0 = fragments of quarter 0, kq = its offsets, Bq = its weights.  Quarter q requests, one per MFMA
  // slot: the TM fragments of the quarter after it, then the refill (K tile `kbn`) of each of its weight fragments as it falls free.
  auto ktile = [&](auto HC, auto FIRSTC, auto FBC, int ktn, unsigned kbn) {
    constexpr int FB = decltype(FBC)::value;
    constexpr auto NOF = std::false_type{};
    auto jobs = [&](auto qc, v4i (&an)[TM], unsigned koff) {
      return [&, koff](auto slc) {
        constexpr int q = decltype(qc)::value, sl = decltype(slc)::value;
        if constexpr (sl < TM && !(DC_EXP & 128)) an[sl] = *reinterpret_cast<const v4i*>(smem + abase[sl] + koff);  // (128: no activation fragment reads)
        if constexpr (sl >= TM && sl % TM == 0) load_B1(Bq[q][sl / TM - 1], kbn, q, sl / TM - 1);
        if constexpr (sl == U - 1) load_B1(Bq[q][NT - 1], kbn, q, NT - 1);
        if constexpr (q == 0 && sl == (U > TM ? TM : TM - 1)) kqn = load_k(ktn);
      };
    };
    quarter(HC, FIRSTC, std::integral_constant<int, FB>{}, A0, Bq[0], jobs(std::integral_constant<int, 0>{}, A1, kq.y));
    quarter(HC, NOF, std::integral_constant<int, FB < 0 ? -1 : FB + U>{}, A1, Bq[1], jobs(std::integral_constant<int, 1>{}, A0, kq.z));
    quarter(HC, NOF, std::integral_constant<int, FB < 0 ? -1 : FB + 2 * U>{}, A0, Bq[2], jobs(std::integral_constant<int, 2>{}, A1, kq.w));
    quarter(HC, NOF, std::integral_constant<int, FB < 0 ? -1 : FB + 3 * U>{}, A1, Bq[3], jobs(std::integral_constant<int, 3>{}, A0, kqn.x));
    kq = kqn;
  };
  // One half of a band pass: all K tiles into accumulator set H; FILL: the other set is requantised under the first KF of them.
  // `kb_next`: the first K tile of the phase that follows (its weights are requested by this phase's last K tile).
  auto phase = [&](auto HC, auto FILLC, int pass, const Ctx& other, unsigned kb_next) {
    constexpr int H = decltype(HC)::value;
    constexpr bool FILL = decltype(FILLC)::value;
    constexpr auto T = std::true_type{};
    constexpr auto F = std::false_type{};
    kq = load_k(0);
    load_A(A0, kq.x);
    const unsigned kb0 = bbase(pass, 0, H);
    const int nkt = p.nkt;
    auto kbn_of = [&](int kt) { return kt + 1 < nkt ? kb0 + (unsigned)(kt + 1) * (unsigned)KT_BYTES : kb_next; };
    int kt0;
    if constexpr (FILL) {
      fill_begin(std::integral_constant<int, 1 - H>{}, other);
      dc_static_for<0, KF>([&](auto kc) {
        constexpr int kt = decltype(kc)::value;
        ktile(HC, std::integral_constant<bool, kt == 0>{}, std::integral_constant<int, kt * SK>{}, kt + 1 < nkt ? kt + 1 : kt, kbn_of(kt));
      });
      kt0 = KF;
    } else {
      ktile(HC, T, std::integral_constant<int, -1>{}, 1 < nkt ? 1 : 0, kbn_of(0));
      kt0 = 1;
    }
#pragma clang loop unroll(disable)
    for (int kt = kt0; kt < nkt; ++kt) ktile(HC, F, std::integral_constant<int, -1>{}, kt + 1 < nkt ? kt + 1 : kt, kbn_of(kt));
  };

  // ---- POOL: pooled rows [j0, j1) of image img, feature chunks [c16_lo, c16_lo + C16) from the LDS ring -> the output tensor
  //      (i8ie_pconv.hip's pool pass with 256 threads).  Returns the number of store instructions this wave issued.
  auto pool_pass = [&](int img, int j0, int j1, int c16_lo, int C16) {
    const int tasks = (j1 - j0) * p.PW * C16;
    const int PHp = p.PH + 2 * p.ob, PWp = p.PW + 2 * p.ob;
    int nst = 0;
    for (int id0 = 0; id0 < tasks; id0 += 256) {
      const int id = id0 + tid;
      if (id0 + wave * 64 < tasks) ++nst;
      if (id < tasks) {
        int t1, c16, jr, px;
        pc_divmod(id, C16, p.rcpC16, t1, c16);
        c16 += c16_lo;
        pc_divmod(t1, p.PW, p.rcpPW, jr, px);
        const int j = j0 + jr;
        typedef unsigned short us2 __attribute__((ext_vector_type(2)));
        const unsigned short lo16 = (unsigned short)lo;  // maxima start at the ReLU's lower bound (0 without one)
        us2 me[4], mo[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          me[q] = us2{lo16, lo16};
          mo[q] = us2{lo16, lo16};
        }
        int rq0, r0;
        pc_divmod(j * p.ps, p.RB, p.rcpRB, rq0, r0);  // first conv row of the window -> ring row
        for (int dy = 0; dy < p.pk; ++dy) {
          int r = r0 + dy;
          if (r >= p.RB) r -= p.RB;
          const uint8_t* rowp = smem + p.lds_otile + (r * p.OW + px * p.ps) * p.opitch + c16 * 16;
          for (int dx = 0; dx < p.pk; ++dx) {
            const v4i v = *reinterpret_cast<const v4i*>(rowp + dx * p.opitch);
            const uint32_t w4[4] = {(uint32_t)v.x, (uint32_t)v.y, (uint32_t)v.z, (uint32_t)v.w};
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const us2 e = __builtin_bit_cast(us2, __builtin_amdgcn_perm(0u, w4[q], 0x0c020c00u));  // [b0, 0, b2, 0]
              const us2 o = __builtin_bit_cast(us2, __builtin_amdgcn_perm(0u, w4[q], 0x0c030c01u));  // [b1, 0, b3, 0]
              me[q] = __builtin_elementwise_max(me[q], e);
              mo[q] = __builtin_elementwise_max(mo[q], o);
            }
          }
        }
        uint32_t r4[4];
#pragma unroll
        for (int q = 0; q < 4; ++q)
          r4[q] = __builtin_amdgcn_perm(__builtin_bit_cast(uint32_t, mo[q]), __builtin_bit_cast(uint32_t, me[q]), 0x06020400u) ^ p.xor_out;
        const unsigned off = ((unsigned)((img * PHp + j + p.ob) * PWp + px + p.ob)) * (unsigned)p.N + (unsigned)c16 * 16u;
        v4u val;
        val.x = r4[0]; val.y = r4[1]; val.z = r4[2]; val.w = r4[3];
        __builtin_amdgcn_raw_buffer_store_b128(val, rsO, (int)off, 0, 0);
      }
    }
    return nst;
  };
  auto rows_done = [&](int hi) {  // pooled rows whose window ends below conv row hi
    if (hi < p.pk) return 0;
    const int e = (hi - p.pk) / p.ps + 1;
    return e < p.PH ? e : p.PH;
  };
  // the pooled rows band `tile` completes, for feature half h
  auto pool_band = [&](int tile, int h) {
    const int img = tile / p.bands, bnd = tile - img * p.bands;
    const int oy0 = bnd * p.RT;
    const int hi = oy0 + p.RT < p.OH ? oy0 + p.RT : p.OH;
    const int j0 = rows_done(oy0), j1 = rows_done(hi);
    return j1 > j0 ? pool_pass(img, j0, j1, h * (BN / 32), BN / 32) : 0;
  };

  // timing builds (DC_EXP & 16): cycles per phase of wave 0..3, summed over the block's bands: [0] H0 K loop, [1] H1 K loop,
  // [2] waiting at the hand-over barrier, [3] DMA issue + pool pass, [4] waiting for the patch, [5] re-bias + second barrier
#if (DC_EXP & 16)
  unsigned long long st_ph[6] = {0, 0, 0, 0, 0, 0}, st_t = __builtin_amdgcn_s_memtime();
  auto stamp = [&](int i) {
    const unsigned long long now = __builtin_amdgcn_s_memtime();
    st_ph[i] += now - st_t;
    st_t = now;
  };
#else
  auto stamp = [](int) {};
#endif
  // =============================== tile loop ===========================================================
  const int patch = p.lds_patch;
  int band = 0;
  auto tile_at = [&](int u, int b) { return p.seq ? u * p.bands + b : u; };
  auto has_next = [&](int u, int b) { return (p.seq && b + 1 < p.bands) || u + per < t_hi; };
  auto next_tile = [&](int u, int b) { return (p.seq && b + 1 < p.bands) ? tile_at(u, b + 1) : tile_at(u + per, 0); };
  constexpr auto H0 = std::integral_constant<int, 0>{};
  constexpr auto H1 = std::integral_constant<int, 1>{};
  pc_wait_vm<0>();  // (the first patch: requested at the top of the kernel)
  patch_xor(patch);
  {
    const unsigned kb = bbase(0, 0, 0);
#pragma unroll
    for (int q4 = 0; q4 < 4; ++q4) load_B(Bq[q4], kb, q4);
  }
  __syncthreads();
  Ctx ctxA{-1, 0}, ctxV{0, 0};  // what the AGPR / VGPR set holds (tile < 0: nothing yet)
  while (unit < t_hi) {
    const bool more = has_next(unit, band);
    const int tile = tile_at(unit, band);
    for (int pass = 0; pass < p.npass; ++pass) {
      constexpr auto FILL_ON = std::integral_constant<bool, !(DC_EXP & 2)>{};
      if (ctxA.tile >= 0) phase(H0, FILL_ON, pass, ctxA, bbase(pass, 0, 1));
      else phase(H0, std::false_type{}, pass, ctxA, bbase(pass, 0, 1));
      stamp(0);
      ctxV = Ctx{tile, pass};
      phase(H1, FILL_ON, pass, ctxV, bbase(pass + 1 < p.npass ? pass + 1 : 0, 0, 0));
      stamp(1);
      const Ctx prevA = ctxA;
      ctxA = Ctx{tile, pass};
      if constexpr (POOL) ctxV = prevA;  // (POOL has one pass: remember whose half-1 bytes reached the ring during this band)
    }
    // ---- hand-over: everyone is done with the patch -> request the next one -> it lands under the pool pass
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // (the fill stores and weight prefetches stay in flight)
    stamp(2);
    if (more) patch_fill(next_tile(unit, band), patch);
    if constexpr (POOL) {
      // in the ring now: half 0's features of this band, half 1's of the band before (this block's previous tile)
      int nst = pool_band(tile, 0);
      if (ctxV.tile >= 0) nst += pool_band(ctxV.tile, 1);
      stamp(3);
      if (more) {
        if (ACC) pc_wait_vm<0>();
        else switch (nst) {  // vector-memory operations retire in issue order: the patch DMA is older than the pool pass's stores
          case 0: pc_wait_vm<0>(); break;
          case 1: pc_wait_vm<1>(); break;
          case 2: pc_wait_vm<2>(); break;
          case 3: pc_wait_vm<3>(); break;
          case 4: pc_wait_vm<4>(); break;
          case 5: pc_wait_vm<5>(); break;
          default: pc_wait_vm<6>(); break;
        }
        stamp(4);
        patch_xor(patch);
      }
    } else if (more) {
      stamp(3);
      pc_wait_vm<0>();
      stamp(4);
      patch_xor(patch);
    }
    __syncthreads();
    stamp(5);
    if (p.seq && band + 1 < p.bands) {
      ++band;
    } else {
      band = 0;
      unit += per;
    }
  }
  // ---- drain: the last half-1 sums, requantised in the open
  asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");  // (the last MFMAs' results -> vector reads)
  if (ctxA.tile >= 0 && !(DC_EXP & 2)) {
    fill_begin(H1, ctxA);
    fill_range(H1, std::integral_constant<int, 0>{}, std::integral_constant<int, U * dc_ops_per_unit(RELU, true)>{});
    if constexpr (POOL) {
      __syncthreads();
      pool_band(ctxA.tile, 1);
    }
  }
  pc_wait_vm<0>();
#if (DC_EXP & 16)
  if (p.dbg && lane == 0)
    for (int i = 0; i < 6; ++i) p.dbg[((size_t)blockIdx.x * 4 + wave) * 8 + i] = st_ph[i];
#endif
}

// ---- weights in fragment order for this kernel's K walk: [pass][kt][quarter q4 = 2 ks + kh][tile of 32 features][lane][16]:
//      lane (h2, l31) holds feature l31 of the tile and K chunk q = 2 kh + h2 of k-step ks = K position 8 kt + 2 q + ks
__global__ __launch_bounds__(256) void dconv_pack_kernel(const int8_t* __restrict__ B, int8_t* __restrict__ Bf, int64_t total16,
                                                         int Kpad, int Npad, const int* __restrict__ perm, int nkt, int bn) {
  const int nt = bn / 32;
  const int64_t gstride = (int64_t)gridDim.x * 256;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total16; e += gstride) {
    const int lane = (int)(e & 63);
    int64_t t = e >> 6;
    const int ntile = (int)(t % nt);
    t /= nt;
    const int q4 = (int)(t & 3);
    t >>= 2;
    const int kt = (int)(t % nkt);
    const int pass = (int)(t / nkt);
    const int ks = q4 >> 1, kh = q4 & 1, h2 = lane >> 5;
    const int chunk = perm[8 * kt + 2 * (2 * kh + h2) + ks];
    const int n = pass * bn + ntile * 32 + (lane & 31);
    uint4 v = make_uint4(0, 0, 0, 0);
    if (chunk >= 0 && chunk * 16 < Kpad && n < Npad) v = *reinterpret_cast<const uint4*>(B + (size_t)n * Kpad + (size_t)chunk * 16);
    reinterpret_cast<uint4*>(Bf)[e] = v;
  }
}

template <int TM, int NT, bool ACC, bool POOL>
int launch_dc_t(i8ie_ctx* ctx, const PCArgs& a, int grid, int lds) {
  static bool raised[64] = {};
  const int dev = ctx->device & 63;
  if (!raised[dev]) {
    I8IE_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&dconv_kernel<TM, NT, ACC, POOL>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    raised[dev] = true;
  }
  dconv_kernel<TM, NT, ACC, POOL><<<grid, 256, lds, ctx->stream>>>(a);
  I8IE_LAUNCH_CHECK();
  return I8IE_OK;
}
template <int TM, int NT>
int launch_dc(i8ie_ctx* ctx, const PCArgs& a, int grid, int lds) {
  if (a.pk > 1)
    return a.acc != nullptr ? launch_dc_t<TM, NT, true, true>(ctx, a, grid, lds) : launch_dc_t<TM, NT, false, true>(ctx, a, grid, lds);
  return a.acc != nullptr ? launch_dc_t<TM, NT, true, false>(ctx, a, grid, lds) : launch_dc_t<TM, NT, false, false>(ctx, a, grid, lds);
}

}  // namespace

// Does the deferred-epilogue kernel take what pconv_impl planned?  (bn: features per pass; PT: pixels per band)
bool i8ie_dconv_eligible(int split, int nkt, int npass, int patch_gran, int PT, int bn, bool pool, int N) {
  if (split || PT > 256 || PT < 129) return false;
  const int TM = (PT + 31) / 32 <= 6 ? 3 : 4;
  int NT;
  if (bn == 256) NT = 2;
  else return false;  // (384-wide passes: the 3 x 3-tile form needs 144 + 144 accumulator registers, and the compiler then moves
                      //  accumulator tiles between the files right behind the MFMAs that write them -- wrong sums; not built)
  if (pool && (npass != 1 || N != bn)) return false;
  if (nkt < dc_fill_ktiles(TM, NT, !pool) + 1) return false;
  if (patch_gran % 256 != 0) return false;
  return true;
}

int i8ie_dconv_launch(i8ie_ctx* ctx, const I8ieIgemmCall& c, PCArgs a, const int* perm_host, int PT, int bn, int grid, int lds, const char* name) {
  // ---- fragment-packed weights for this kernel: once per layer and packing key, kept in the layer handle (I8ieWCache)
  const int kt_bytes = bn * 128;
  const size_t perm_bytes = i8ie_align_up((size_t)a.nkt * 8 * sizeof(int), 256);
  const size_t bf_bytes = (size_t)a.npass * a.nkt * kt_bytes;
  const int row_par = (c.OW * c.sh) & 1;
  const unsigned long long wkey = (2ull << 32) | (unsigned long long)(row_par | (bn << 1));
  void* wbuf = c.wcache->find(wkey);
  if (wbuf == nullptr) {
    I8IE_REQUIRE(ctx->capture == nullptr, "weight re-packing inside a graph capture: run the same calls once eagerly first");
    I8IE_TRY(i8ie_malloc(ctx, perm_bytes + bf_bytes, &wbuf));
    int rc = i8ie_memcpy_h2d(ctx, wbuf, perm_host, (size_t)a.nkt * 8 * sizeof(int));
    if (rc == I8IE_OK) {
      const int64_t total16 = (int64_t)(bf_bytes / 16);
      int64_t blocks = (total16 + 255) / 256;
      if (blocks > 4096) blocks = 4096;
      dconv_pack_kernel<<<(int)blocks, 256, 0, ctx->stream>>>(c.B, (int8_t*)wbuf + perm_bytes, total16, c.Kpad, c.Npad, (const int*)wbuf, a.nkt, bn);
      if (hipGetLastError() != hipSuccess) rc = I8IE_ERR_HIP;
    }
    if (rc != I8IE_OK) {
      i8ie_free(ctx, wbuf);
      return rc;
    }
    c.wcache->ents.push_back(I8ieWCache::Ent{wkey, wbuf});
  }
  a.Bf = (const int8_t*)wbuf + perm_bytes;
  a.perm = (const int*)wbuf;
  a.bf_bytes = (unsigned)bf_bytes;
  a.rcpC16 = 1.0f / (float)(bn / 32);  // (the pool pass runs per feature half: bn / 2 features = bn / 32 chunks of 16)
  const int TM = (PT + 31) / 32 <= 6 ? 3 : 4;
  const double ops = 2.0 * c.M * c.N * c.Ktrue;
  const double bytes = (double)c.M * c.Ktrue + (double)c.N * c.Ktrue + (double)c.M * c.N;
  I8ieProfScope prof(ctx, name, ops, bytes);
#if (DC_EXP & 16)
  static unsigned long long* dbg = nullptr;
  if (!dbg) I8IE_HIP_TRY(hipMalloc(&dbg, 4096 * 8 * sizeof(unsigned long long)));
  I8IE_HIP_TRY(hipMemsetAsync(dbg, 0, 4096 * 8 * sizeof(unsigned long long), ctx->stream));
  a.dbg = dbg;
  {
    const int rc = (TM == 4) ? launch_dc<4, 2>(ctx, a, grid, lds) : launch_dc<3, 2>(ctx, a, grid, lds);
    if (rc == I8IE_OK && std::getenv("I8IE_DCONV_STAMPS") != nullptr) {
      std::vector<unsigned long long> h((size_t)grid * 32);
      I8IE_HIP_TRY(hipStreamSynchronize(ctx->stream));
      I8IE_HIP_TRY(hipMemcpy(h.data(), dbg, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
      double sum[4][6] = {};
      for (int b = 0; b < grid; ++b)
        for (int w = 0; w < 4; ++w)
          for (int i = 0; i < 6; ++i) sum[w][i] += (double)h[((size_t)b * 4 + w) * 8 + i];
      const double tp = (double)a.n_tiles;
      for (int w = 0; w < 4; ++w)
        fprintf(stderr, "dconv_stamps %s wave %d: per band, cycles: H0 K loop %.0f | H1 K loop %.0f | hand-over barrier %.0f | DMA issue + pool %.0f | patch wait %.0f | re-bias + barrier %.0f  (%d K tiles, %d bands)\n",
                name, w, sum[w][0] / tp, sum[w][1] / tp, sum[w][2] / tp, sum[w][3] / tp, sum[w][4] / tp, sum[w][5] / tp, a.nkt, a.n_tiles);
    }
    return rc;
  }
#endif
  if (TM == 4 && bn == 256) return launch_dc<4, 2>(ctx, a, grid, lds);
  if (TM == 3 && bn == 256) return launch_dc<3, 2>(ctx, a, grid, lds);
  i8ie_set_error("i8ie_dconv_launch: shape not planned for this kernel");
  return I8IE_ERR_ARG;
}
