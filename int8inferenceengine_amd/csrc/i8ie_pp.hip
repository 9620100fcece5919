// i8ie_pp.hip -- persistent ping-pong implicit-GEMM Conv2d over bordered NHWC u8 activations.
//
//   C[r][j] = sum_k A_u8[r][k] * W_s8[j][k] + oc[j]     (src/conv2d.cc:131-133: cblas_gemm_s8u8s32 + oc)
//   out     = relu?(down_scale(C))                      (src/quantize_utils.cc:27-36, src/functional.cc:15-26)
//
// Same arithmetic as igemm_u8s8_kernel<1,...> (i8ie_igemm.hip); what differs is the schedule, built for the
// large layers (AlexNet conv2-5 at batch 1000) where that kernel sat at 0.38 of the int8 MFMA rate:
//
//   * One 512-thread workgroup per CU, alive for the whole launch, walks a list of 256 x 256 (or 224 / 192 row)
//     output tiles.  LDS holds two K stages of 64 KiB (A: 2 half-tiles of 128 rows x 128 B, B: the same),
//     every byte arrives by LDS-DMA (buffer_load ... lds, 1 KiB per wave-instruction, source chunk chosen so
//     that 16-B chunk c of row r lands at chunk c ^ (r & 7): conflict-free ds_read_b128 fragments).
//   * The 8 waves form two groups of four (one wave of each group per SIMD).  A K tile is four phases, one per
//     quadrant of the wave's output (row half a x column half b).  In each phase a group first LOADS (fragment
//     reads, its share of the next K tile's DMA, a counted vmcnt) and then COMPUTES (16 MFMAs 16x16x64);
//     the groups run one barrier apart, so on every SIMD one wave issues MFMAs while the other loads.
//   * Quadrant-major order means a half-tile is read in one phase only, so the other stage is refilled a whole
//     K tile ahead with >= 6 barrier intervals of flight time and never more than 4 pieces issued per phase.
//   * Tiles follow each other without draining: the K-tile stream (and its DMA) runs across the tile boundary;
//     quadrant (a, b) of a finished tile is requantised and stored in the phase after its last MFMA, while the
//     other group's MFMAs (and the next tile's DMA) proceed.  oc'[j] enters as the C operand of the first MFMA.
//   * Epilogue per value: cvt, fma, max, cvt_pk (the estimate of i8ie_requant.h, proven equal to the exact
//     sequence on every int32 for the launch's scales before this mode is used), v_permlane16_swap to 8-byte
//     row pieces, buffer_store (out-of-range rows / features dropped by the descriptor: every wave issues the
//     same number of memory instructions, which the counted vmcnt waits rely on).
#include <cstdio>
#include <cstdlib>
#include <mutex>
#include <type_traits>
#include <vector>

#include "i8ie_calls.h"
#include "i8ie_internal.h"
#include "i8ie_requant.h"

namespace {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef unsigned v2u __attribute__((ext_vector_type(2)));

constexpr int kHalf = 16384;   // one half-tile buffer: 128 rows x 128 B
constexpr int kStage = 65536;  // A0 A1 B0 B1
constexpr int kOcpOff = 2 * kStage;
constexpr int kOcpMax = 1024;  // features covered by the LDS copy of oc'
constexpr int kLdsBytes = 2 * kStage + kOcpMax * 4;
constexpr int kBN = 256;
constexpr unsigned kRowInvalid = 0xC0000000u;  // beyond any output buffer this kernel accepts (< 2^31 bytes)

struct PPArgs {
  const uint8_t* A;
  unsigned a_bytes;
  int M;
  unsigned img_pitch, row_pitch, C;
  int sh, sw, OH, OW, RC;
  unsigned row_jump;
  float rcpP, rcpOW;  // 1 / (OH * OW), 1 / OW
  const int8_t* B;
  unsigned b_bytes;
  int Kpad, N, Npad;
  const int32_t* ocp;
  I8ieRequant rq;
  int relu_lo;
  uint8_t* out;
  unsigned out_bytes;
  int ob, OHp, OWp;
  int tiles_m, tiles_n, nk;
};

#define PP_BAR() asm volatile("s_barrier" ::: "memory")
#define PP_WAIT_VM(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")

// x / d and x % d for 0 <= x < 2^23 with rd = 1.0f / d: the float estimate is off by at most one
__device__ __forceinline__ void divmod_f(int x, int d, float rd, int& qo, int& ro) {
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

extern __shared__ __attribute__((aligned(16))) uint8_t pp_smem[];

// R: 16-row MFMA tiles per wave and row half (both groups): block tile = 64 R rows x 256 features
template <int R, bool PROVEN>
__global__ __launch_bounds__(512, 2) void pp_conv_kernel(PPArgs p) {
  constexpr int HR = 2 * R * 16;  // rows per A half-tile actually used
  constexpr int BM = 2 * HR;
  uint8_t* const smem = pp_smem;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = wave >> 2, wc = wave & 3;
  const int lq = lane >> 4;

  // ---- this block's tiles: XCD x (blocks with equal blockIdx % 8 share an L2) owns a contiguous range of tiles
  // and its blocks take consecutive tiles of it in every step, so neighbouring windows meet in one L2
  const int total = p.tiles_m * p.tiles_n;
  const int per = (int)gridDim.x >> 3;
  const int xcd = blockIdx.x & 7, jb = blockIdx.x >> 3;
  const int Tx = (total + 7) >> 3;
  const int t_lo = xcd * Tx;
  const int t_hi = t_lo + Tx < total ? t_lo + Tx : total;
  int tile_c = t_lo + jb;
  if (tile_c >= t_hi) return;  // (the whole workgroup)

  // oc'[j] = oc[j] + 128 * wsum[j] into LDS (features past Npad: zero)
  for (int i = tid; i < kOcpMax; i += 512)
    reinterpret_cast<int*>(smem + kOcpOff)[i] = i < p.Npad ? p.ocp[i] : 0;

  const __amdgpu_buffer_rsrc_t rsA =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(p.A), 0, p.a_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsB =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<int8_t*>(p.B), 0, p.b_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsO = __builtin_amdgcn_make_buffer_rsrc(p.out, 0, p.out_bytes, 0x00020000);

  // ---- fragment read addresses: lane -> row lane & 15, 16-B chunk (4 ks + lane / 16) ^ (row & 7)
  int loA[2], loB[2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    const int lo = (lane & 15) * 128 + (((ks * 4 + lq) ^ (lane & 7)) << 4);
    loA[ks] = lo + g * (R * 16 * 128);
    loB[ks] = lo + 2 * kHalf + wc * (32 * 128);
  }

  // ---- fill cursor (one K tile ahead of the MFMAs, crossing tile boundaries)
  const int cA = (lane & 7) ^ ((lane >> 3) & 7);  // source chunk of this lane's LDS slot
  const int prow = wave * 8 + (lane >> 3);         // row inside a 64-row piece group
  unsigned koff0;
  int f0;
  {
    const int kh = cA / p.RC;
    f0 = cA - kh * p.RC;
    koff0 = (unsigned)kh * p.row_pitch + (unsigned)f0 * 16;
  }
  unsigned a_off[4], b_off[4];  // [half * 2 + piece]
  unsigned koff = koff0;
  int f = f0;
  int kt_f = 0, tile_f = tile_c, fs = 0;
  bool fill_ok = true;
  const int P = p.OH * p.OW;
  auto tile_offsets = [&](int t) {
    const int tm = t / p.tiles_n, tn = t - tm * p.tiles_n;
    const int m0 = tm * BM, n0 = tn * kBN;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      int r = m0 + (i >> 1) * HR + (i & 1) * 64 + prow;
      r = r < p.M ? r : p.M - 1;  // rows past M: computed, never stored
      int img, rem, oh, ow;
      divmod_f(r, P, p.rcpP, img, rem);
      divmod_f(rem, p.OW, p.rcpOW, oh, ow);
      a_off[i] = (unsigned)img * p.img_pitch + (unsigned)(oh * p.sh) * p.row_pitch + (unsigned)(ow * p.sw) * p.C;
      int br = n0 + (i >> 1) * 128 + (i & 1) * 64 + prow;
      br = br < p.Npad ? br : p.Npad - 1;  // feature rows past the panel: computed, never stored
      b_off[i] = (unsigned)br * (unsigned)p.Kpad + (unsigned)cA * 16u;
    }
  };
  auto dma = [&](const __amdgpu_buffer_rsrc_t& rs, int lds_off, unsigned voff, int soff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(smem + lds_off), 16,
                                             (int)voff, soff, 0, 0);
  };
  auto fill_A = [&](int a) {
#pragma unroll
    for (int j = 0; j < 2; ++j) dma(rsA, fs + a * kHalf + (j * 64 + wave * 8) * 128, a_off[a * 2 + j] + koff, 0);
  };
  auto fill_B = [&](int b) {
#pragma unroll
    for (int j = 0; j < 2; ++j) dma(rsB, fs + (2 + b) * kHalf + (j * 64 + wave * 8) * 128, b_off[b * 2 + j], kt_f * 128);
  };
  auto advance = [&]() {  // the cursor moves to the next K tile (of the next tile of this block, at the end)
    fs ^= kStage;
    ++kt_f;
    koff += 128;
    f += 8;
    while (f >= p.RC) {
      f -= p.RC;
      koff += p.row_jump;
    }
    if (kt_f == p.nk) {
      kt_f = 0;
      tile_f += per;
      koff = koff0;
      f = f0;
      fill_ok = tile_f < t_hi;
      if (fill_ok) tile_offsets(tile_f);
    }
  };

  // ---- output rows of the tile whose quadrants are being stored: byte offset of row (a, mi, lane & 15)
  unsigned obase[2 * R];
  int n0e = 0;  // first feature of that tile
  auto tile_rows = [&](int t) {
    const int tm = t / p.tiles_n, tn = t - tm * p.tiles_n;
    n0e = tn * kBN;
    const int m0 = tm * BM;
#pragma unroll
    for (int i = 0; i < 2 * R; ++i) {
      const int r = m0 + (i / R) * HR + g * (R * 16) + (i % R) * 16 + (lane & 15);
      int img, rem, oh, ow;
      divmod_f(r < p.M ? r : 0, P, p.rcpP, img, rem);
      divmod_f(rem, p.OW, p.rcpOW, oh, ow);
      const unsigned pix = ((unsigned)img * p.OHp + oh + p.ob) * p.OWp + ow + p.ob;
      obase[i] = r < p.M ? pix * (unsigned)p.N : kRowInvalid;
    }
  };
  // this lane's 8 bytes of a 32-feature row piece after the swap below: features 16 (q & 1) + 8 (q >> 1) ...
  const int colb = wc * 32 + 16 * (lq & 1) + 8 * (lq >> 1);

  v4i acc[2][2][R][2];  // [row half][column half][row tile][feature tile]; lane & 15 = row, 4 regs = 4 features
  const I8ieRequant rq = p.rq;
  const int lo = p.relu_lo;
  const float lof = (float)lo, bias = rq.zpf - 0.5f;

  auto epilogue = [&](auto ac, auto bc) {
    constexpr int a = decltype(ac)::value, b = decltype(bc)::value;
    const int col = n0e + b * 128 + colb;
    const bool colok = col < p.N;
#pragma unroll
    for (int mi = 0; mi < R; ++mi) {
      uint32_t d[2];
#pragma unroll
      for (int ni = 0; ni < 2; ++ni) {
        const v4i c = acc[a][b][mi][ni];
        if (PROVEN) {
          d[ni] = i8ie_requant_pack4_proven(c.x, c.y, c.z, c.w, rq.ms, bias, lof);
        } else {
          const int cv[4] = {c.x, c.y, c.z, c.w};
          d[ni] = i8ie_requant_pack4(cv, rq, lo, lof);
        }
      }
      // rows of 16 lanes: odd rows of d[0] <-> even rows of d[1]: every lane then holds 8 consecutive features
      const auto sw = __builtin_amdgcn_permlane16_swap(d[0], d[1], false, false);
      const unsigned voff = colok ? obase[a * R + mi] + (unsigned)col : kRowInvalid;
      v2u val;
      val.x = sw[0];
      val.y = sw[1];
      __builtin_amdgcn_raw_buffer_store_b64(val, rsO, (int)voff, 0, 0);
    }
  };

  v4i Af[R][2], Bf[2][2];
  auto read_A = [&](auto ac) {
    constexpr int a = decltype(ac)::value;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int mi = 0; mi < R; ++mi)
        Af[mi][ks] = *reinterpret_cast<const v4i*>(smem + loA[ks] + a * kHalf + mi * 2048) ^ (int)0x80808080;  // u8 -> s8
  };
  auto read_B = [&](auto bc) {
    constexpr int b = decltype(bc)::value;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int ni = 0; ni < 2; ++ni) Bf[ni][ks] = *reinterpret_cast<const v4i*>(smem + loB[ks] + b * kHalf + ni * 2048);
  };
  int n0c = 0;  // first feature of the tile being accumulated
  // first K tile of an output tile: the accumulators of quadrant (a, b) start as oc'[j] (C = sum + oc', exact)
  auto init_acc = [&](auto ac, auto bc) {
    constexpr int a = decltype(ac)::value, b = decltype(bc)::value;
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
      const v4i* src = reinterpret_cast<const v4i*>(smem + kOcpOff + (n0c + b * 128 + wc * 32 + ni * 16 + 4 * lq) * 4);
#pragma unroll
      for (int mi = 0; mi < R; ++mi) acc[a][b][mi][ni] = *src;
    }
  };
  // The MFMAs are written as asm with the accumulator tied (D = C): with the builtin, hipcc gave the three
  // unrolled K-tile variants different accumulator registers and paid for it in copies and 140 spilled VGPRs.
  // Operand hazards: A/B fragments come from ds_read (the compiler's lgkmcnt covers asm inputs) and v_xor
  // (s_nop 1 ahead of the cluster); results are read by VALU at least one barrier later.
  auto mfma_quad = [&](auto ac, auto bc) {
    constexpr int a = decltype(ac)::value, b = decltype(bc)::value;
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
    asm volatile("s_nop 1");
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int mi = 0; mi < R; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
          asm volatile("v_mfma_i32_16x16x64_i8 %0, %1, %2, %0" : "+v"(acc[a][b][mi][ni]) : "v"(Bf[ni][ks]), "v"(Af[mi][ks]));
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
  };

  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  bool have_prev = false;

  // One K tile = four phases.  Memory instructions per phase, in program order (the counted waits depend on it):
  //   phase 1: [R stores: quadrant (1,0) of the previous tile, first K tile only] [4 DMA: A0, A1 of the next K tile]
  //   phase 2: [R stores: quadrant (0,0), last K tile only]                      [4 DMA: B0, B1 of the next K tile]
  //   phase 3: [R stores: quadrant (0,1), last K tile only]
  //   phase 4: [R stores: quadrant (1,1), last K tile only]
  // Needed before the reads of phase 1 of the next K tile: its A0 and B0 -> wait in phase 4 for all but
  // B1 (2) and the stores issued since (2 R in the last K tile).  Needed before the reads of phase 2: B1 of this
  // K tile (issued in phase 2 of the previous one) -> wait in phase 1 for all but what was issued after it.
  // `first` / `last` are run-time (workgroup-uniform) flags on purpose: three unrolled variants of this body
  // got three different accumulator register assignments from hipcc, with copies and 140 spilled VGPRs between.
  auto ktile = [&](bool first, bool last) {
    // ---- phase 1: quadrant (0, 0)
    read_A(I0{});
    read_B(I0{});
    if (first) {
      if (have_prev) epilogue(I1{}, I0{});
      init_acc(I0{}, I0{});
    }
    __builtin_amdgcn_sched_barrier(0);
    if (fill_ok) {
      fill_A(0);
      fill_A(1);
    }
    __builtin_amdgcn_sched_barrier(0);
    if (first && have_prev) {
      if (R == 4) PP_WAIT_VM(16); else if (R == 3) PP_WAIT_VM(13); else PP_WAIT_VM(10);  // 3 R stores + 4 DMA younger than B1
    } else if (!fill_ok) {
      PP_WAIT_VM(0);
    } else {
      PP_WAIT_VM(4);
    }
    PP_BAR();
    mfma_quad(I0{}, I0{});
    PP_BAR();
    // ---- phase 2: quadrant (0, 1)
    read_B(I1{});
    if (first) {
      init_acc(I0{}, I1{});
      tile_rows(tile_c);  // (quadrant (1,0) of the previous tile has left)
    }
    if (last) epilogue(I0{}, I0{});
    __builtin_amdgcn_sched_barrier(0);
    if (fill_ok) {
      fill_B(0);
      fill_B(1);
      advance();
    }
    __builtin_amdgcn_sched_barrier(0);
    PP_BAR();
    mfma_quad(I0{}, I1{});
    PP_BAR();
    // ---- phase 3: quadrant (1, 1)
    read_A(I1{});
    if (first) init_acc(I1{}, I1{});
    if (last) epilogue(I0{}, I1{});
    __builtin_amdgcn_sched_barrier(0);
    PP_BAR();
    mfma_quad(I1{}, I1{});
    PP_BAR();
    // ---- phase 4: quadrant (1, 0)
    read_B(I0{});
    if (first) init_acc(I1{}, I0{});
    if (last) epilogue(I1{}, I1{});
    __builtin_amdgcn_sched_barrier(0);
    if (last) {
      if (R == 4) PP_WAIT_VM(10); else if (R == 3) PP_WAIT_VM(8); else PP_WAIT_VM(6);  // B1 + 2 R stores younger than B0
    } else {
      PP_WAIT_VM(2);
    }
    PP_BAR();
    mfma_quad(I1{}, I0{});
    PP_BAR();
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      loA[ks] ^= kStage;
      loB[ks] ^= kStage;
    }
  };

  // ---- prologue: K tile 0 of the first tile into stage 0
  tile_offsets(tile_f);
  fill_A(0);
  fill_A(1);
  fill_B(0);
  fill_B(1);
  advance();
  PP_WAIT_VM(0);
  __syncthreads();  // (also publishes the oc' table)
  if (g == 1) PP_BAR();  // the second group runs one barrier behind the first

#pragma clang loop unroll(disable)
  for (; tile_c < t_hi; tile_c += per) {
    n0c = (tile_c % p.tiles_n) * kBN;
#pragma clang loop unroll(disable)
    for (int kt = 0; kt < p.nk; ++kt) ktile(kt == 0, kt + 1 == p.nk);
    have_prev = true;
  }
  epilogue(I1{}, I0{});
  if (g == 0) PP_BAR();
}

// ---- host side ----------------------------------------------------------------------------------------------
struct ProofKey {
  float sa, sb, sc;
  int zp, lo;
  bool ok;
};
bool proven_cached(const I8ieRequant& rq, int lo) {
  static std::mutex mu;
  static std::vector<ProofKey> cache;
  std::lock_guard<std::mutex> lk(mu);
  for (const ProofKey& k : cache)
    if (k.sa == rq.sa && k.sb == rq.sb && k.sc == rq.sc && k.zp == (int)rq.zpf && k.lo == lo) return k.ok;
  const bool ok = i8ie_requant_prove(rq, lo);
  if (cache.size() > 256) cache.clear();
  cache.push_back(ProofKey{rq.sa, rq.sb, rq.sc, (int)rq.zpf, lo, ok});
  return ok;
}

template <int R, bool PROVEN>
int launch_pp(i8ie_ctx* ctx, const PPArgs& a, int grid) {
  static bool raised[64] = {};  // per device: allow the 130 KiB of dynamic LDS
  const int dev = ctx->device & 63;
  if (!raised[dev]) {
    I8IE_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&pp_conv_kernel<R, PROVEN>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes));
    raised[dev] = true;
  }
  pp_conv_kernel<R, PROVEN><<<grid, 512, kLdsBytes, ctx->stream>>>(a);
  I8IE_LAUNCH_CHECK();
  return I8IE_OK;
}

}  // namespace

int i8ie_pp_try_launch(i8ie_ctx* ctx, const I8ieIgemmCall& c) {
  if (c.amode != 1 || c.acc != nullptr || c.biasf != nullptr) return 0;
  if (c.N % 16 != 0 || c.N <= 128 || c.N > kOcpMax || c.Npad > kOcpMax) return 0;
  if ((reinterpret_cast<uintptr_t>(c.out) & 15u) != 0 || c.C % 16 != 0) return 0;
  const int nk = c.Kpad / 128;
  if (nk < 2 || c.M >= (1 << 23) || c.M < 256 * 64) return 0;  // the tiled kernel serves small launches
  const size_t P = (size_t)c.OH * c.OW;
  const size_t out_pixels = (size_t)(c.M / (int)P) * (c.OH + 2 * c.ob) * (c.OW + 2 * c.ob);
  const size_t out_bytes = out_pixels * (size_t)c.N;
  if (out_bytes >= ((size_t)1 << 31) || c.a_bytes >= ((size_t)1 << 32) - 4096) return 0;

  PPArgs a{};
  a.A = c.A;
  a.a_bytes = (unsigned)c.a_bytes;
  a.M = c.M;
  a.C = (unsigned)c.C;
  a.row_pitch = (unsigned)c.Wp * (unsigned)c.C;
  a.img_pitch = (unsigned)c.Hp * a.row_pitch;
  a.sh = c.sh; a.sw = c.sw; a.OH = c.OH; a.OW = c.OW;
  a.RC = c.KW * (c.C / 16);
  a.row_jump = a.row_pitch - (unsigned)a.RC * 16u;
  a.rcpP = 1.0f / (float)P;
  a.rcpOW = 1.0f / (float)c.OW;
  a.B = c.B;
  a.b_bytes = (unsigned)((size_t)c.Npad * c.Kpad);
  a.Kpad = c.Kpad; a.N = c.N; a.Npad = c.Npad;
  a.ocp = c.ocp;
  a.rq = i8ie_make_requant(c.s_in, c.s_w, c.s_out, c.zp_out);
  a.relu_lo = c.relu ? c.zp_out : 0;
  a.out = c.out;
  a.out_bytes = (unsigned)out_bytes;
  a.ob = c.ob; a.OHp = c.OH + 2 * c.ob; a.OWp = c.OW + 2 * c.ob;
  constexpr int R = 4;
  a.tiles_m = (c.M + 64 * R - 1) / (64 * R);
  a.tiles_n = (c.N + kBN - 1) / kBN;
  a.nk = nk;
  const bool proven = a.rq.fast != I8IE_RQ_EXACT && proven_cached(a.rq, a.relu_lo);
  if (proven) a.rq.fast = I8IE_RQ_PROVEN;

  hipDeviceProp_t* prop = nullptr;
  static hipDeviceProp_t props[64];
  static bool have[64] = {};
  const int dev = ctx->device & 63;
  if (!have[dev]) {
    I8IE_HIP_TRY(hipGetDeviceProperties(&props[dev], ctx->device));
    have[dev] = true;
  }
  prop = &props[dev];
  int grid = prop->multiProcessorCount / 8 * 8;
  if (grid < 8) grid = 8;
  const double ops = 2.0 * c.M * c.N * c.Ktrue;
  const double bytes = (double)c.M * c.Ktrue + (double)c.N * c.Ktrue + (double)c.M * c.N;
  char tag[64];
  snprintf(tag, sizeof(tag), "pp_conv_256x256|M%d,N%d,K%d", c.M, c.N, c.Kchunks * 16);
  I8ieProfScope prof(ctx, ctx->prof ? tag : "pp_conv_256x256", ops, bytes);
  const int rc = proven ? launch_pp<R, true>(ctx, a, grid) : launch_pp<R, false>(ctx, a, grid);
  return rc == I8IE_OK ? 1 : rc;
}
