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
#include <algorithm>
#include <mutex>
#include <type_traits>
#include <vector>

#include "i8ie_calls.h"
#include "i8ie_diag.h"
#include "i8ie_internal.h"
#include "i8ie_requant.h"

namespace {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef unsigned v2u __attribute__((ext_vector_type(2)));

constexpr int kHalf = 16384;   // one half-tile buffer: 128 rows x 128 B
constexpr int kStage = 65536;  // A0 A1 B0 B1
constexpr int kB0x = 2 * kStage;          // third B0 buffer (B0 is read in phases 1 and 4: see the schedule below)
constexpr int kOcpOff = 2 * kStage + kHalf;
constexpr int kOcpMax = 512;   // features covered by the LDS copy of oc'
constexpr int kTabOff = kOcpOff + kOcpMax * 4;
constexpr int kMaxP = 1536;    // output pixels per image covered by the row tables (2 x 4 B each)
constexpr int kLdsBytes = kTabOff + 2 * 4 * kMaxP;
static_assert(kLdsBytes <= 160 * 1024, "LDS budget");
constexpr int kBN = 256;
constexpr unsigned kRowInvalid = 0xC0000000u;  // beyond any output buffer this kernel accepts (< 2^31 bytes)

struct PPArgs {
  const uint8_t* A;
  unsigned a_bytes;
  int M;
  unsigned img_pitch, row_pitch, C;
  int sh, sw, OH, OW, RC;
  unsigned row_jump;
  float rcpP, rcpOW, rcpRC;  // 1 / (OH * OW), 1 / OW, 1 / RC
  const int8_t* B;
  unsigned b_bytes;
  int Kpad, N, Npad;
  const int32_t* ocp;
  I8ieRequant rq;
  float pms, pbias;  // the proven estimate fma(C, pms, pbias) (i8ie_requant_fit), when rq.fast == I8IE_RQ_PROVEN
  int relu_lo;
  uint8_t* out;
  unsigned out_bytes;
  int ob, OHp, OWp;
  int tiles_m, tiles_n, nk;
  unsigned long long* dbg;  // DBG bit 3: [block][4] = {shader cycles, 100 MHz ticks, K tiles, 0}
};

#define PP_BAR() asm volatile("s_barrier" ::: "memory")
#define PP_WAIT_VM(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")
template <int N, int DBG = 0>
__device__ __forceinline__ void pp_wait_vm() {
  if (DBG & 4096) return;  // (timing experiment: no counted waits at all -- results are garbage)
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

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
// DBG (diagnostic builds only): bit 0 = no u8->s8 xor, bit 1 = no DMA after the prologue, bit 2 = no fragment reads
// (results wrong with any of those); bit 3 = stamp s_memtime / s_memrealtime around the tile loop into p.dbg (the
// in-kernel clock of MI355X_MICROARCH.md, DVFS item 6; stamps go nowhere else, results unchanged)
template <int R, bool PROVEN, int DBG = 0>
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

  // row tables: output pixel `rem` of an image -> byte offset of its window in the input image, and of its row in
  // the (bordered) output image: a row index then costs one division (by OH * OW) instead of two
  for (int i = tid; i < p.OH * p.OW; i += 512) {
    int oh, ow;
    divmod_f(i, p.OW, p.rcpOW, oh, ow);
    reinterpret_cast<unsigned*>(smem + kTabOff)[i] = (unsigned)(oh * p.sh) * p.row_pitch + (unsigned)(ow * p.sw) * p.C;
    reinterpret_cast<unsigned*>(smem + kTabOff + 4 * kMaxP)[i] = (unsigned)((oh + p.ob) * p.OWp + ow + p.ob) * (unsigned)p.N;
  }

  const __amdgpu_buffer_rsrc_t rsA =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(p.A), 0, p.a_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsB =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<int8_t*>(p.B), 0, p.b_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsO = __builtin_amdgcn_make_buffer_rsrc(p.out, 0, p.out_bytes, 0x00020000);

  // ---- fragment read addresses: lane -> row lane & 15, 16-B chunk (4 ks + lane / 16) ^ (row & 7).
  // loA carries the stage (toggled per K tile); B1 sits a wave-uniform distance from it; B0 lives in a ring of three
  int loA[2], lo0[2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    const int lo = (lane & 15) * 128 + (((ks * 4 + lq) ^ (lane & 7)) << 4);
    loA[ks] = lo + g * (R * 16 * 128);
    lo0[ks] = lo + wc * (32 * 128);
  }
  const int dB1 = 3 * kHalf + wc * (32 * 128) - g * (R * 16 * 128);  // B1 fragment address - loA
  int r0 = 2 * kHalf, r1 = kStage + 2 * kHalf, r2 = kB0x;            // B0 buffers of K tiles t, t + 1, t + 2

  // ---- fill schedule.  LDS-DMA takes about 1.1 us (2300 cycles) from issue to landed under load, a K tile is
  // 2048 MFMA cycles, and LDS holds two K tiles: so every half-tile buffer is refilled two phases after its
  // (last) read and waited for one phase before its next read -- five phases (10 barrier intervals) of flight.
  // While K tile t is multiplied (reads: phase 1 A0 B0, phase 2 B1, phase 3 A1, phase 4 B0 again):
  //   phase 1: A1 of K tile t+1 -> other stage    (A1 was last read in phase 3 of t-1)
  //   phase 2: B0 of K tile t+2 -> ring buffer r2 (last read in phase 4 of t-1; B0 has three buffers because its
  //                                                second read leaves only two phases to the next first read)
  //   phase 3: A0 of K tile t+2 -> this stage     (read in phase 1)
  //   phase 4: B1 of K tile t+2 -> this stage     (read in phase 2)
  // vmcnt retires in issue order, so "X has landed" = "all but the operations issued after X are done": ten DMA
  // pieces (five half-tiles) in steady state, plus the epilogue stores that were issued in between.
  // The cursors never stop: past the block's last tile they re-read its last tile into buffers nobody reads
  // again, which keeps the number of memory instructions per phase constant.
  const int cA = (DBG & 16) ? (lane & 7) : ((lane & 7) ^ ((lane >> 3) & 7));  // source chunk of this lane's LDS slot
  const int prow = wave * 8 + (lane >> 3);         // row inside a 64-row piece group
  const int P = p.OH * p.OW;
  auto koff_of = [&](int kt) -> unsigned {  // byte offset of this lane's chunk of K tile kt inside the window
    int kh, f;
    divmod_f(kt * 8 + cA, p.RC, p.rcpRC, kh, f);
    return (unsigned)kh * p.row_pitch + (unsigned)f * 16;
  };
  auto a_row_off = [&](int t, int half, int j) -> unsigned {
    const int tm = t / p.tiles_n;
    int r = tm * BM + half * HR + j * 64 + prow;
    r = r < p.M ? r : p.M - 1;  // rows past M: computed, never stored
    int img, rem;
    divmod_f(r, P, p.rcpP, img, rem);
    return (unsigned)img * p.img_pitch + reinterpret_cast<const unsigned*>(smem + kTabOff)[rem];
  };
  auto b_row_off = [&](int t, int half, int j) -> unsigned {
    const int tn = t % p.tiles_n;
    int br = tn * kBN + half * 128 + j * 64 + prow;
    br = br < p.Npad ? br : p.Npad - 1;  // feature rows past the panel: computed, never stored
    return (unsigned)br * (unsigned)p.Kpad + (unsigned)cA * 16u;
  };
  auto dma = [&](const __amdgpu_buffer_rsrc_t& rs, int lds_off, unsigned voff, int soff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(smem + lds_off), 16,
                                             (int)voff, soff, 0, 0);
  };
  // weights: every CU streams the same panel once per tile and never re-reads a line, so they go around the L1
  // (sc1: served by L2 as usual) and leave it to the activation windows, whose K tiles overlap
  auto dma_b = [&](const __amdgpu_buffer_rsrc_t& rs, int lds_off, unsigned voff, int soff) {
    if (DBG & 65536) {  // timing experiment: the same bytes as an ordinary load to VGPRs (no LDS write)
      const v4i v = __builtin_bit_cast(v4i, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)voff, soff, 0));
      asm volatile("" ::"v"(v));
      return;
    }
    if (DBG & 8192)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(smem + lds_off), 16,
                                               (int)voff, soff, 0, 16);
    else if (DBG & 16384)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(smem + lds_off), 16,
                                               (int)voff, soff, 0, 2);
    else
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(smem + lds_off), 16,
                                               (int)voff, soff, 0, 0);
  };
  // half-tile `hb` (0 A0, 1 A1, 2 B0, 3 B1) of stage `st`: this wave's two pieces
  // piece j (0 / 1) of a half-tile: the first one goes out in the load slot, the second behind the phase's last
  // MFMA (an LDS-DMA instruction holds its wave for ~150 cycles while the SIMD's other wave issues MFMAs, ~70
  // when it does not: two per load slot made that slot 1.7 x the MFMA slot)
  auto fill_A = [&](int st, int a, const unsigned (&off)[2], unsigned koff, int j) {
    if (DBG & 32) return;
    dma(rsA, st + a * kHalf + (j * 64 + wave * 8) * 128,
        (DBG & 32768) ? ((off[j] + koff) & 0xFFFu) : (DBG & 128) ? ((off[j] + koff) & 0xFFFFu) : off[j] + koff, 0);
  };
  auto fill_B = [&](int buf, const unsigned (&off)[2], int kt, int j) {  // buf: LDS offset of the half-tile buffer
    if (DBG & 64) return;
    dma_b(rsB, buf + (j * 64 + wave * 8) * 128, (DBG & 128) ? (off[j] & 0xFFFFu) : off[j], (DBG & 128) ? 0 : kt * 128);
  };
  unsigned aA1[2], koff1;                   // cursor 1 = K tile t + 1 (A1)
  unsigned aA0[2], bB0[2], bB1[2], koff2;   // cursor 2 = K tile t + 2 (B0, A0, B1)
  int c1_tile = tile_c, c1_kt = 0, c2_tile = tile_c, c2_kt = 0;
  auto next_kt = [&](int& tile, int& kt) -> bool {  // true when the cursor entered another tile
    if (++kt == p.nk) {
      kt = 0;
      if (tile + per < t_hi) {
        tile += per;
        return true;
      }
    }
    return false;
  };
  auto c2_rows = [&]() {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      aA0[j] = a_row_off(c2_tile, 0, j);
      bB0[j] = b_row_off(c2_tile, 0, j);
      bB1[j] = b_row_off(c2_tile, 1, j);
    }
  };
  auto c1_rows = [&]() {
#pragma unroll
    for (int j = 0; j < 2; ++j) aA1[j] = a_row_off(c1_tile, 1, j);
  };
  int cs = 0;  // LDS offset of the stage being multiplied

  // ---- output rows of the tile whose quadrants are being stored: byte offset of row (a, mi, lane & 15)
  unsigned obase[2 * R];
  int n0e = 0;  // first feature of that tile
  auto tile_rows = [&](int t) {
    const int tm = t / p.tiles_n, tn = t - tm * p.tiles_n;
    n0e = tn * kBN;
    const int r0 = tm * BM + g * (R * 16) + (lane & 15);  // row of (a = 0, mi = 0); the others are 16 mi + HR a further
    int img, rem;
    divmod_f(r0 < p.M ? r0 : 0, P, p.rcpP, img, rem);
    const unsigned out_img = (unsigned)p.OHp * p.OWp * (unsigned)p.N;
#pragma unroll
    for (int i = 0; i < 2 * R; ++i) {
      const int r = r0 + (i / R) * HR + (i % R) * 16;
      obase[i] = r < p.M ? (unsigned)img * out_img + reinterpret_cast<const unsigned*>(smem + kTabOff + 4 * kMaxP)[rem] : kRowInvalid;
      rem += (i % R == R - 1) ? HR - (R - 1) * 16 : 16;  // on to the next row
      while (rem >= P) {
        rem -= P;
        ++img;
      }
    }
  };
  // this lane's 8 bytes of a 32-feature row piece after the swap below: features 16 (q & 1) + 8 (q >> 1) ...
  const int colb = wc * 32 + 16 * (lq & 1) + 8 * (lq >> 1);

  v4i acc[2][2][R][2];  // [row half][column half][row tile][feature tile]; lane & 15 = row, 4 regs = 4 features
  const I8ieRequant rq = p.rq;
  const int lo = p.relu_lo;
  const float lof = (float)lo;

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
          d[ni] = i8ie_requant_pack4_proven(c.x, c.y, c.z, c.w, p.pms, p.pbias, lof);
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
  if (DBG & 4) {
    for (int i = 0; i < R; ++i) for (int k = 0; k < 2; ++k) Af[i][k] = v4i{tid * 77 + i, tid * 3 + k, tid ^ 0x55aa55, tid * 0x01010101};
    for (int i = 0; i < 2; ++i) for (int k = 0; k < 2; ++k) Bf[i][k] = v4i{tid * 177 + i, tid * 5 + k, tid ^ 0x33cc33, tid * 0x01030107};
  }
  // A fragments of row half a, k-step ks (raw u8; xor_frag re-biases them to s8 on their way to the MFMA)
  auto read_A = [&](auto ac, auto ksc) {
    constexpr int a = decltype(ac)::value, ks = decltype(ksc)::value;
#pragma unroll
    for (int mi = 0; mi < R; ++mi)
      if (!(DBG & 4)) Af[mi][ks] = *reinterpret_cast<const v4i*>(smem + loA[ks] + a * kHalf + mi * 2048);
  };
  auto xor_half = [&](int j, int h) {  // half of fragment j = ks * R + mi (compile-time after unrolling)
    if (!(DBG & 1)) {                  // u8 -> s8 (128 * wsum is part of oc')
      if (h == 0) {
        Af[j % R][j / R].x ^= (int)0x80808080;
        Af[j % R][j / R].y ^= (int)0x80808080;
      } else {
        Af[j % R][j / R].z ^= (int)0x80808080;
        Af[j % R][j / R].w ^= (int)0x80808080;
      }
    }
  };
  auto xor_frag = [&](int j) {
    xor_half(j, 0);
    xor_half(j, 1);
  };
  auto read_B = [&](auto bc) {
    constexpr int b = decltype(bc)::value;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int ni = 0; ni < 2; ++ni)
        if (!(DBG & 4))
          Bf[ni][ks] = *reinterpret_cast<const v4i*>(smem + (b == 0 ? lo0[ks] + r0 : loA[ks] + dB1) + ni * 2048);
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
  // XORS: the row half was read in this phase: fragments 0 and 1 were re-biased before the barrier, fragment
  // j + 2 is re-biased between the two MFMAs of fragment j (two VALU operations fit in the gap of a 16x16x64).
  auto mfma_quad = [&](auto ac, auto bc, auto xorsc) {
    constexpr int a = decltype(ac)::value, b = decltype(bc)::value;
    constexpr bool XORS = decltype(xorsc)::value;
    __builtin_amdgcn_sched_barrier(0);
    if (!(DBG & 512)) __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int j = 0; j < 2 * R; ++j) {
      const int mi = j % R, ks = j / R;
      asm volatile("v_mfma_i32_16x16x64_i8 %0, %1, %2, %0" : "+v"(acc[a][b][mi][0]) : "v"(Bf[0][ks]), "v"(Af[mi][ks]));
      __builtin_amdgcn_sched_barrier(0);
      if (XORS && j + 2 < 2 * R) xor_half(j + 2, 0);
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("v_mfma_i32_16x16x64_i8 %0, %1, %2, %0" : "+v"(acc[a][b][mi][1]) : "v"(Bf[1][ks]), "v"(Af[mi][ks]));
      __builtin_amdgcn_sched_barrier(0);
      if (XORS && j + 2 < 2 * R) xor_half(j + 2, 1);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (!(DBG & 512)) __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
  };

  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;

  // One K tile = four phases; `first` / `last` are run-time (workgroup-uniform) flags on purpose: three unrolled
  // variants of this body got three different accumulator register assignments from hipcc, with copies and
  // 140 spilled VGPRs between them.  Memory instructions per phase, in program order (the counted waits depend
  // on it): [R stores of a finished quadrant, first / last K tile only] then [2 DMA pieces]:
  //   phase 1: stores (1,0) of the previous tile | A1(t+1)        phase 3: stores (0,1) | A0(t+2)
  //   phase 2: stores (0,0)                      | B0(t+2)        phase 4: stores (1,1) | B1(t+2)
  // Waits (each in the phase before the read, behind that phase's own DMA), = 10 + R x (store groups in between):
  //   (the phase's second piece follows its MFMAs, so 9 of the 10 are out when the wait executes)
  //   phase 1 for B1(t)   (issued phase 4 of t-2): + 4 R when t is a first K tile (3 of the previous tile's last
  //                        K tile, 1 of this phase), + R when t-1 was a first K tile with stores;
  //   phase 2 for A1(t)   (issued phase 1 of t-1): + 4 R when t is first, + R when t is last;
  //   phase 4 for A0(t+1), B0(t+1) (issued phases 3 / 2 of t-1): + 2 R when t is first, + 3 R when t is last.
  bool tile_has_prev = false;  // the current tile's first K tile stored quadrant (1,0) of a previous tile
  // DBG bit 11: s_memtime stamps around the slots of phase 2 of every middle K tile (diagnostic build: the stamps'
  // lgkmcnt(0) forbids overlaps the real kernel has; read shares, not lengths)
  unsigned long long sl[7] = {0, 0, 0, 0, 0, 0, 0}, sn = 0, ta = 0, tb = 0, tc = 0, td = 0, te = 0, tf = 0, tg = 0, th = 0;
  auto stamp = [&](unsigned long long& t) {
    if (DBG & 2048) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  };
  auto ktile = [&](bool first, bool second, bool last) {
    const int os = cs ^ kStage;
    const bool fp = first && tile_has_prev;
    // ---- phase 1: quadrant (0, 0)
    if ((DBG & 1024) && !(DBG & 2)) {
      fill_A(os, 1, aA1, koff1, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    read_A(I0{}, I0{});
    read_A(I0{}, I1{});
    read_B(I0{});
    if (first) {
      if (tile_has_prev) epilogue(I1{}, I0{});
      init_acc(I0{}, I0{});
    }
    __builtin_amdgcn_sched_barrier(0);
    if (!(DBG & 2)) {
      if (!(DBG & 1024)) fill_A(os, 1, aA1, koff1, 0);
      __builtin_amdgcn_sched_barrier(0);
      if (fp) pp_wait_vm<9 + 4 * R, DBG>();
      else if (second && tile_has_prev) pp_wait_vm<9 + R, DBG>();
      else pp_wait_vm<9, DBG>();
    }
    __builtin_amdgcn_sched_barrier(0);
    xor_frag(0);
    xor_frag(1);
    __builtin_amdgcn_sched_barrier(0);
    PP_BAR();
    mfma_quad(I0{}, I0{}, std::true_type{});
    if (!(DBG & 2)) fill_A(os, 1, aA1, koff1, 1);
    PP_BAR();
    // ---- phase 2: quadrant (0, 1)
    stamp(ta);
    if ((DBG & 1024) && !(DBG & 2)) {
      fill_B(r2, bB0, c2_kt, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    read_B(I1{});
    if (first) {
      init_acc(I0{}, I1{});
      tile_rows(tile_c);  // (quadrant (1,0) of the previous tile has left)
    }
    if (last) epilogue(I0{}, I0{});
    __builtin_amdgcn_sched_barrier(0);
    stamp(tg);
    if (!(DBG & 2)) {
      if (!(DBG & 1024)) fill_B(r2, bB0, c2_kt, 0);
      __builtin_amdgcn_sched_barrier(0);
      stamp(th);
      if (fp) pp_wait_vm<9 + 4 * R, DBG>();
      else if (last) pp_wait_vm<9 + R, DBG>();
      else pp_wait_vm<9, DBG>();
    }
    __builtin_amdgcn_sched_barrier(0);
    stamp(tb);
    PP_BAR();
    stamp(tc);
    mfma_quad(I0{}, I1{}, std::false_type{});
    stamp(td);
    if (!(DBG & 2)) fill_B(r2, bB0, c2_kt, 1);
    stamp(te);
    PP_BAR();
    stamp(tf);
    if ((DBG & 2048) && !first && !last) {
      sl[0] += tb - ta;
      sl[1] += tc - tb;
      sl[2] += td - tc;
      sl[3] += te - td;
      sl[4] += tf - te;
      sl[5] += th - tg;  // the load slot's DMA piece
      sl[6] += tb - th;  // the counted vmcnt wait
      ++sn;
    }
    // ---- phase 3: quadrant (1, 1)
    if ((DBG & 1024) && !(DBG & 2)) {
      fill_A(cs, 0, aA0, koff2, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    read_A(I1{}, I0{});
    read_A(I1{}, I1{});
    if (first) init_acc(I1{}, I1{});
    if (last) epilogue(I0{}, I1{});
    __builtin_amdgcn_sched_barrier(0);
    if (!(DBG & 2) && !(DBG & 1024)) fill_A(cs, 0, aA0, koff2, 0);
    __builtin_amdgcn_sched_barrier(0);
    xor_frag(0);
    xor_frag(1);
    __builtin_amdgcn_sched_barrier(0);
    PP_BAR();
    mfma_quad(I1{}, I1{}, std::true_type{});
    if (!(DBG & 2)) fill_A(cs, 0, aA0, koff2, 1);
    PP_BAR();
    // ---- phase 4: quadrant (1, 0)
    if ((DBG & 1024) && !(DBG & 2)) {
      fill_B(cs + 3 * kHalf, bB1, c2_kt, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    read_B(I0{});
    if (first) init_acc(I1{}, I0{});
    if (last) epilogue(I1{}, I1{});
    __builtin_amdgcn_sched_barrier(0);
    if (!(DBG & 2)) {
      if (!(DBG & 1024)) fill_B(cs + 3 * kHalf, bB1, c2_kt, 0);
      __builtin_amdgcn_sched_barrier(0);
      if (fp) pp_wait_vm<9 + 2 * R, DBG>();
      else if (last) pp_wait_vm<9 + 3 * R, DBG>();
      else pp_wait_vm<9, DBG>();
    }
    __builtin_amdgcn_sched_barrier(0);
    PP_BAR();
    mfma_quad(I1{}, I0{}, std::false_type{});
    if (!(DBG & 2)) fill_B(cs + 3 * kHalf, bB1, c2_kt, 1);
    PP_BAR();
    // ---- both cursors move on by one K tile; the stages swap, the B0 ring turns
    koff1 = koff2;
    if (next_kt(c1_tile, c1_kt)) c1_rows();
    if (next_kt(c2_tile, c2_kt)) c2_rows();
    koff2 = koff_of(c2_kt);
    cs = os;
    {
      const int t = r0;
      r0 = r1;
      r1 = r2;
      r2 = t;
    }
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) loA[ks] ^= kStage;
  };

  // ---- prologue: K tile 0 of the first tile into stage 0, B0 / A0 / B1 of K tile 1 into stage 1
  __syncthreads();  // the row tables (and oc') are in LDS
  {
    c1_rows();
    c2_rows();
    const unsigned k0 = koff_of(0);
    for (int j = 0; j < 2; ++j) {
      fill_A(0, 0, aA0, k0, j);
      fill_A(0, 1, aA1, k0, j);
      fill_B(r0, bB0, 0, j);
      fill_B(3 * kHalf, bB1, 0, j);
    }
    next_kt(c2_tile, c2_kt);  // K tile 1 (nk >= 2: same tile)
    koff1 = koff_of(1);
    for (int j = 0; j < 2; ++j) {
      fill_B(r1, bB0, 1, j);
      fill_A(kStage, 0, aA0, koff1, j);
      fill_B(kStage + 3 * kHalf, bB1, 1, j);
    }
    next_kt(c1_tile, c1_kt);  // cursor 1 = K tile 1
    if (next_kt(c2_tile, c2_kt)) c2_rows();  // cursor 2 = K tile 2 (of the next tile when nk == 2)
    koff2 = koff_of(c2_kt);
  }
  PP_WAIT_VM(0);
  __syncthreads();
  if (g == 1 && !(DBG & 256)) PP_BAR();  // the second group runs one barrier behind the first

  unsigned long long st_c = 0, st_r = 0, st_k = 0;
  if (DBG & 8) {
    st_c = __builtin_amdgcn_s_memtime();
    st_r = __builtin_amdgcn_s_memrealtime();
  }
#pragma clang loop unroll(disable)
  for (; tile_c < t_hi; tile_c += per) {
    if (DBG & 8) st_k += p.nk;
    n0c = (tile_c % p.tiles_n) * kBN;
#pragma clang loop unroll(disable)
    for (int kt = 0; kt < p.nk; ++kt) ktile(kt == 0, kt == 1, kt + 1 == p.nk);
    tile_has_prev = true;
  }
  epilogue(I1{}, I0{});
  PP_WAIT_VM(0);  // the cursors' last (unread) DMA pieces must not land in LDS that already belongs to another workgroup
  if (g == 0 && !(DBG & 256)) PP_BAR();
  if (DBG & 8) {
    const unsigned long long e_c = __builtin_amdgcn_s_memtime(), e_r = __builtin_amdgcn_s_memrealtime();
    if ((DBG & 2048) && lane == 0) {
      for (int i = 0; i < 7; ++i) p.dbg[4096 * 4 + (blockIdx.x * 8 + wave) * 8 + i] = sl[i];
      p.dbg[4096 * 4 + (blockIdx.x * 8 + wave) * 8 + 7] = sn;
    }
    if (tid == 0) {
      p.dbg[blockIdx.x * 4 + 0] = e_c - st_c;
      p.dbg[blockIdx.x * 4 + 1] = e_r - st_r;
      p.dbg[blockIdx.x * 4 + 2] = st_k;
      p.dbg[blockIdx.x * 4 + 3] = 1;
    }
  }
}

// ---- host side ----------------------------------------------------------------------------------------------
struct ProofKey {
  float sa, sb, sc;
  int zp, lo;
  bool ok;
  float ms, bias;
};
bool proven_cached(const I8ieRequant& rq, int lo, float* ms, float* bias) {
  static std::mutex mu;
  static std::vector<ProofKey> cache;
  std::lock_guard<std::mutex> lk(mu);
  for (const ProofKey& k : cache)
    if (k.sa == rq.sa && k.sb == rq.sb && k.sc == rq.sc && k.zp == (int)rq.zpf && k.lo == lo) {
      *ms = k.ms;
      *bias = k.bias;
      return k.ok;
    }
  ProofKey k{rq.sa, rq.sb, rq.sc, (int)rq.zpf, lo, false, 0.0f, 0.0f};
  k.ok = i8ie_requant_fit(rq, lo, &k.ms, &k.bias);
  if (cache.size() > 256) cache.clear();
  cache.push_back(k);
  *ms = k.ms;
  *bias = k.bias;
  return k.ok;
}

template <int R, bool PROVEN, int DBG = 0>
int launch_pp(i8ie_ctx* ctx, const PPArgs& a, int grid) {
  static bool raised[64] = {};  // per device: allow the 130 KiB of dynamic LDS
  const int dev = ctx->device & 63;
  if (!raised[dev]) {
    I8IE_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&pp_conv_kernel<R, PROVEN, DBG>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes));
    raised[dev] = true;
  }
  pp_conv_kernel<R, PROVEN, DBG><<<grid, 512, kLdsBytes, ctx->stream>>>(a);
  I8IE_LAUNCH_CHECK();
  return I8IE_OK;
}

}  // namespace

extern "C" int i8ie_requant_fit_host(float sa, float sb, float sc, int zp_c, int relu, float* ms, float* bias) {
  if (ms == nullptr || bias == nullptr) return I8IE_ERR_ARG;
  const I8ieRequant rq = i8ie_make_requant(sa, sb, sc, zp_c);
  return i8ie_requant_fit(rq, relu ? zp_c : 0, ms, bias) ? 1 : 0;
}

extern "C" int i8ie_requant_eval_host(float sa, float sb, float sc, int zp_c, int relu, float ms, float bias,
                                      const int32_t* acc_host, int64_t n, uint8_t* exact_host, uint8_t* estimate_host) {
  if (acc_host == nullptr || n < 0) return I8IE_ERR_ARG;
  const I8ieRequant rq = i8ie_make_requant(sa, sb, sc, zp_c);
  const int lo = relu ? zp_c : 0;
  for (int64_t i = 0; i < n; ++i) {
    if (exact_host) exact_host[i] = (uint8_t)i8ie_requant_exact((float)acc_host[i], rq, lo);
    if (estimate_host) estimate_host[i] = (uint8_t)i8ie_estimate_host((float)acc_host[i], ms, bias, lo);
  }
  return I8IE_OK;
}

int i8ie_pp_try_launch(i8ie_ctx* ctx, const I8ieIgemmCall& c) {
  if (c.amode != 1 || c.acc != nullptr || c.biasf != nullptr) return 0;
  if (c.N % 16 != 0 || c.N <= 128 || c.N > kOcpMax || c.Npad > kOcpMax || c.OH * c.OW > kMaxP) return 0;
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
  a.rcpRC = 1.0f / (float)a.RC;
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
  const bool proven = a.rq.fast != I8IE_RQ_EXACT && proven_cached(a.rq, a.relu_lo, &a.pms, &a.pbias);
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
  if (const char* e = std::getenv("I8IE_PP_GRID")) grid = std::atoi(e) / 8 * 8 > 0 ? std::atoi(e) / 8 * 8 : grid;  // (experiments)
  const double ops = 2.0 * c.M * c.N * c.Ktrue;
  const double bytes = (double)c.M * c.Ktrue + (double)c.N * c.Ktrue + (double)c.M * c.N;
  char tag[64];
  snprintf(tag, sizeof(tag), "pp_conv_256x256|M%d,N%d,K%d", c.M, c.N, c.Kchunks * 16);
  I8ieProfScope prof(ctx, ctx->prof ? tag : "pp_conv_256x256", ops, bytes);
  int rc;
  // variants 21-23 / 24-27: diagnostic builds (see DBG above); 24-27 print the in-kernel clock to stderr
  static unsigned long long* dbg_dev = nullptr;
  const int v = ctx->variant;
  if ((v >= 24 && v <= 27) || v == 38 || v == 39) {
    if (dbg_dev == nullptr) I8IE_HIP_TRY(hipMalloc((void**)&dbg_dev, (4096 * 4 + 4096 * 8 * 8) * sizeof(unsigned long long)));
    I8IE_HIP_TRY(hipMemsetAsync(dbg_dev, 0, (4096 * 4 + 4096 * 8 * 8) * sizeof(unsigned long long), ctx->stream));
    a.dbg = dbg_dev;
  }
  if (v == 21) rc = launch_pp<R, true, 1>(ctx, a, grid);
  else if (v == 22) rc = launch_pp<R, true, 3>(ctx, a, grid);
  else if (v == 23) rc = launch_pp<R, true, 7>(ctx, a, grid);
  else if (v == 30) rc = launch_pp<R, true, 1 | 16>(ctx, a, grid);
  else if (v == 31) rc = launch_pp<R, true, 1 | 32>(ctx, a, grid);
  else if (v == 32) rc = launch_pp<R, true, 1 | 64>(ctx, a, grid);
  else if (v == 33) rc = launch_pp<R, true, 1 | 128>(ctx, a, grid);
  else if (v == 34) rc = launch_pp<R, true, 1 | 4>(ctx, a, grid);
  else if (v == 37) rc = launch_pp<R, true, 1 | 1024>(ctx, a, grid);
  else if (v == 40) rc = launch_pp<R, true, 1 | 4096>(ctx, a, grid);
  else if (v == 41) rc = launch_pp<R, true, 1 | 8192>(ctx, a, grid);
  else if (v == 42) rc = launch_pp<R, true, 1 | 16384>(ctx, a, grid);
  else if (v == 43) rc = launch_pp<R, true, 1 | 32768>(ctx, a, grid);  // A sources inside 4 KiB: L1 hits (timing only)
  else if (v == 44) rc = launch_pp<R, true, 1 | 32 | 65536>(ctx, a, grid);  // no A fills, B as VGPR loads (timing only)
  else if (v == 35) rc = launch_pp<R, true, 1 | 256>(ctx, a, grid);
  else if (v == 36) rc = launch_pp<R, true, 1 | 512>(ctx, a, grid);
  else if (v == 38) rc = launch_pp<R, true, 8 | 2048 | 1>(ctx, a, grid);
  else if (v == 39) rc = launch_pp<R, true, 8 | 2048 | 1 | 2>(ctx, a, grid);
  else if (v == 24) rc = launch_pp<R, true, 8>(ctx, a, grid);
  else if (v == 25) rc = launch_pp<R, true, 9>(ctx, a, grid);
  else if (v == 26) rc = launch_pp<R, true, 11>(ctx, a, grid);
  else if (v == 27) rc = launch_pp<R, true, 15>(ctx, a, grid);
  else rc = proven ? launch_pp<R, true>(ctx, a, grid) : launch_pp<R, false>(ctx, a, grid);
  if (rc == I8IE_OK && (v == 38 || v == 39) && std::getenv("I8IE_PP_CLOCK") != nullptr) {
    std::vector<unsigned long long> h((size_t)grid * 8 * 8);
    I8IE_HIP_TRY(hipStreamSynchronize(ctx->stream));
    I8IE_HIP_TRY(hipMemcpy(h.data(), dbg_dev + 4096 * 4, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    double s[2][7] = {}, n[2] = {};
    for (int b = 0; b < grid; ++b)
      for (int w = 0; w < 8; ++w) {
        const unsigned long long* e = &h[((size_t)b * 8 + w) * 8];
        if (e[7] == 0) continue;
        for (int i = 0; i < 7; ++i) s[w >> 2][i] += (double)e[i] / (double)e[7];
        n[w >> 2] += 1;
      }
    for (int gi = 0; gi < 2; ++gi)
      if (n[gi] > 0)
        fprintf(stderr, "pp_stamps variant %d group %d phase 2 (cycles): load slot %.0f | barrier %.0f | 16 MFMA %.0f | tail DMA %.0f | barrier %.0f ; inside the load slot: DMA piece %.0f, vmcnt wait %.0f\n", v,
                gi, s[gi][0] / n[gi], s[gi][1] / n[gi], s[gi][2] / n[gi], s[gi][3] / n[gi], s[gi][4] / n[gi], s[gi][5] / n[gi], s[gi][6] / n[gi]);
  }
  if (rc == I8IE_OK && v >= 24 && v <= 27 && std::getenv("I8IE_PP_CLOCK") != nullptr) {
    std::vector<unsigned long long> h((size_t)grid * 4);
    I8IE_HIP_TRY(hipStreamSynchronize(ctx->stream));
    I8IE_HIP_TRY(hipMemcpy(h.data(), dbg_dev, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    std::vector<double> mhz, cpk;
    for (int b = 0; b < grid; ++b)
      if (h[b * 4 + 3] && h[b * 4 + 1] && h[b * 4 + 2]) {
        mhz.push_back((double)h[b * 4 + 0] / (double)h[b * 4 + 1] * 100.0);
        cpk.push_back((double)h[b * 4 + 0] / (double)h[b * 4 + 2]);
      }
    if (!mhz.empty()) {
      std::sort(mhz.begin(), mhz.end());
      std::sort(cpk.begin(), cpk.end());
      fprintf(stderr, "pp_clock variant %d M %d N %d K %d: median in-kernel clock %.0f MHz (min %.0f max %.0f), %.0f shader cycles per K tile (median block)\n",
              v, c.M, c.N, c.Kchunks * 16, mhz[mhz.size() / 2], mhz.front(), mhz.back(), cpk[cpk.size() / 2]);
    }
  }
  return rc == I8IE_OK ? 1 : rc;
}
