// i8ie_tconv.hip -- patch-stationary implicit-GEMM Conv2d, two wave teams half a tile apart.
//
//   C[r][j] = sum_k A_u8[r][k] * W_s8[j][k] + oc[j]     (src/conv2d.cc:131-133: cblas_gemm_s8u8s32 + oc)
//   out     = relu?(down_scale(C))                      (src/quantize_utils.cc:27-36, src/functional.cc:15-26)
//
// i8ie_pconv.hip showed (phase stamps) that with the input patch resident in LDS and the
// weights streamed from L2 straight into registers the K loop reaches ~89 % of the MFMA rate, and that what is left
// is per-tile serial work during which no wave issues MFMAs: the requantising epilogue (7-9 k cycles), the wait for
// the next patch (5 k) and its u8 -> s8 pass (1.4-2.3 k): 25 % of a 60 k-cycle tile.
// This kernel keeps the MFMA pipe busy through all of that:
//   * The 8 waves of the block (one block per CU) form two TEAMS of four, one wave of each team per SIMD.  A team
//     owns one half of the tile's output features (wave = 128 (or 96) pixels x 64 (48) features, as in pconv).
//   * A tile pass is cut into two SEGMENTS (K halves; the second one ends with the epilogue).  Team 1 runs exactly one
//     segment behind team 0, and one workgroup barrier per segment keeps it there.  So on every SIMD, while one
//     wave requantises, stores, waits for a DMA or re-biases a patch, the other is in the middle of its K loop
//     and has the MFMA pipe to itself.  The first segment gets the larger share of K, so that the wave heading
//     for its epilogue leaves the pipe early (and it runs at raised priority).
//   * Patches live in a ring of LDS buffers.  Small patches (C <= 256): whole patches, two buffers.  Large ones
//     (C = 384: 90 KB): two channel slices per patch, K ordered slice-major so that segment = slice, three buffers.
//     The slice a segment needs is requested one segment earlier by all 8 waves (LDS-DMA, each lane later re-biases
//     exactly the 16 bytes it fetched itself, so no hand-over inside that step), and published by the barrier.
//   * Weights: fragment order for this K walk ([pass][K tile][k-step][feature tile][lane][16 B]), read with plain
//     coalesced buffer loads (1 KiB per wave-instruction) two k-steps ahead; the two waves of a team that
//     need the same fragment ask for it at about the same time (L1).  No weight bytes in LDS, no barrier in the K loop.
// Results are those of the other contraction kernels bit for bit (same exact integer sums, same requantiser).
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "i8ie_calls.h"
#include "i8ie_internal.h"
#include "i8ie_requant.h"

namespace {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef unsigned v2u __attribute__((ext_vector_type(2)));

constexpr unsigned kRowInvalid = 0xC0000000u;  // beyond any output buffer this kernel accepts (< 2^31 bytes)
constexpr int kTabPix = 256;                   // pixels per tile at most (16 MFMA row tiles)

struct TCArgs {
  const uint8_t* A;
  unsigned a_bytes;
  unsigned img_pitch, row_pitch, C;
  int OH, OW, s, KW, Wp;
  float rcpOW;
  int RT, bands, n_tiles;  // output rows per tile, tiles per image, tiles in all
  int S, R;                // channel slices per patch (1 or 2), ring buffers (2 or 3)
  int CCs, Ps;             // 16-byte chunks per pixel of a slice; LDS pixel pitch of a slice (bytes) = 16 CCs + 16
  float rcpCCs1;
  int slice_gran;          // 16-byte granules of a ring buffer (multiple of 512)
  int row_gran;            // granules of a slice row in LDS: Wp (CCs + 1) + pad, so that consecutive output pixels keep
                           // walking the 16 slots of 256 B across the row wrap (as in i8ie_pconv.hip)
  float rcpRowGran;
  const int* perm;         // S == 1: [nkt * 8] source chunk of K position ci (-1 = zero padding), pairs of equal LDS
                           // slot parity (i8ie_pconv.hip); null: natural (slice, tap, chunk) order
  int Ks, Ksp;             // K chunks of a slice: valid, padded to whole K tiles
  const int8_t* Bf;        // [pass][kt][ks][feature tile][lane][16]
  unsigned bf_bytes;
  int nkt, kt_split;       // K tiles of a pass; segment 0 = [0, kt_split), segment 1 = [kt_split, nkt)
  int N, npass;
  const int32_t* ocp;
  int Npad;
  I8ieRequant rq;
  int relu_lo;
  uint8_t* out;
  unsigned out_bytes;
  int ob, OHp, OWp;
  int lds_ring, lds_ocp, lds_tab, lds_ktab;  // LDS offsets
  int flags;                                 // timing experiments (diagnostic build): 1 = no priority changes, 2 = teams in phase
  int32_t* acc;  // ACC kernels: [M][N] pre-requant accumulators (the cblas_gemm_s8u8s32 result, src/conv2d.cc:131-133)
  unsigned long long* dbg;
};

#define TC_BAR() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")  // (LDS writes of the re-bias pass are out before the hand-over)
__device__ __forceinline__ void tc_wait_vm0() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
__device__ __forceinline__ void tc_divmod(int x, int d, float rd, int& qo, int& ro) {  // 0 <= x < 2^23
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

extern __shared__ __attribute__((aligned(16))) uint8_t tc_smem[];

// MFMA row r of a 16-row tile <-> pixel offset inside the tile (even pixels for rows 0-3 and 12-15, odd ones for
// rows 4-11: with lane group q reading chunk 2 q (+1), the 16 lanes ds_read_b128 serves per cycle hit 16 slots)
__device__ __forceinline__ int tc_row_to_pix(int r) { return (r < 4) ? 2 * r : (r >= 12 ? 2 * (r - 8) : 2 * (r - 4) + 1); }

// TMW: 16-pixel row tiles per wave (2 waves of a team along the pixels); NTW: 16-feature tiles per wave
// (2 waves of a team x 2 teams along the features: a pass is 64 NTW features wide)
// ACC: also dump the INT32 accumulators (acc_dbg of the C-ABI) -- a separate instantiation, the default one is untouched
template <int TMW, int NTW, bool ACC>
__global__ __launch_bounds__(512, 2) void tconv_kernel(TCArgs p) {
  uint8_t* const smem = tc_smem;
  constexpr int BN = NTW * 64;
  constexpr int KT_BYTES = BN * 128;  // weights of one K tile of a pass: 2 k-steps x (BN / 16) fragments of 1 KiB
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int team = (p.flags & 2) ? 0 : wave >> 2;  // (in-phase experiment: both teams keep team 0's clock)
  const int tm = (wave >> 1) & 1;
  const int ntile0 = ((wave >> 2) * 2 + (wave & 1)) * NTW;  // first feature tile of this wave inside a pass
  const int lq = lane >> 4, lr = lane & 15;

  // ---- tiles of this block: XCD-contiguous ranges, consecutive tiles to the blocks of one XCD
  const int per = (int)gridDim.x >> 3;
  const int xcd = blockIdx.x & 7, jb = blockIdx.x >> 3;
  const int Tx = (p.n_tiles + 7) >> 3;
  const int t_lo = xcd * Tx;
  const int t_hi = t_lo + Tx < p.n_tiles ? t_lo + Tx : p.n_tiles;
  const int tile0 = t_lo + jb;
  if (tile0 >= t_hi) return;
  const int nb = (t_hi - 1 - tile0) / per + 1;  // tiles of this block: tile0, tile0 + per, ...

  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(p.A), 0, p.a_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<int8_t*>(p.Bf), 0, p.bf_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsO = __builtin_amdgcn_make_buffer_rsrc(p.out, 0, p.out_bytes, 0x00020000);

  // ---- per-kernel tables in LDS: oc'; for pixel i of a tile its window origin in a slice and its output offset;
  //      for (K tile, k-step, lane group) the offset of its 16 channels inside a slice
  for (int i = tid; i < p.npass * BN; i += 512) reinterpret_cast<int*>(smem + p.lds_ocp)[i] = i < p.Npad ? p.ocp[i] : 0;
  const int PT = p.RT * p.OW;
  for (int i = tid; i < kTabPix; i += 512) {
    int oy, ox;
    tc_divmod(i < PT ? i : 0, p.OW, p.rcpOW, oy, ox);
    reinterpret_cast<unsigned*>(smem + p.lds_tab)[i] = (unsigned)(oy * p.s) * (unsigned)(p.row_gran * 16) + (unsigned)(ox * p.s) * (unsigned)p.Ps;
    reinterpret_cast<unsigned*>(smem + p.lds_tab)[kTabPix + i] = (unsigned)(oy * p.OWp + ox) * (unsigned)p.N;
  }
  for (int i = tid; i < p.nkt * 8; i += 512) {
    const int ci = 8 * (i >> 3) + 2 * (i & 3) + ((i >> 2) & 1);  // chunk of (K tile i / 8, k-step (i / 4) & 1, lane group i & 3)
    int r = ci, ok;
    if (p.perm != nullptr) {
      r = p.perm[ci];
      ok = r >= 0;
      if (!ok) r = 0;
    } else {
      const int sl = ci >= p.Ksp ? 1 : 0;  // (natural order: slice-major, each slice padded to whole K tiles)
      r = ci - sl * p.Ksp;
      ok = r < p.Ks;
    }
    int tap, cc, kh, kw;
    tc_divmod(r, p.CCs, 1.0f / (float)p.CCs, tap, cc);
    tc_divmod(tap, p.KW, 1.0f / (float)p.KW, kh, kw);
    reinterpret_cast<unsigned*>(smem + p.lds_ktab)[i] = ok ? (unsigned)kh * (unsigned)(p.row_gran * 16) + (unsigned)kw * (unsigned)p.Ps + (unsigned)cc * 16u : 0u;
  }
  __syncthreads();
  const int pix0 = tm * TMW * 16 + tc_row_to_pix(lr);
  unsigned abase[TMW];
#pragma unroll
  for (int mi = 0; mi < TMW; ++mi) {
    const int pi = pix0 + mi * 16;
    abase[mi] = reinterpret_cast<const unsigned*>(smem + p.lds_tab)[pi < kTabPix ? pi : 0];
  }
  const bool ghost = (tm * TMW + TMW - 1) * 16 >= PT;  // the last row tile of the upper wave row may not exist

  // ---- slice n = (tile index j, channel slice sl) -> ring buffer n % R: each lane fetches granule g0 + tid
  const int CCs1 = p.CCs + 1;
  const int slice_bytes = p.slice_gran * 16;
  auto ring_of = [&](int n) { return p.lds_ring + (n % p.R) * slice_bytes; };
  auto slice_fill = [&](int j, int sl) {
    const int t = tile0 + j * per;
    const int img = t / p.bands, band = t - img * p.bands;
    const unsigned src0 = (unsigned)img * p.img_pitch + (unsigned)(band * p.RT * p.s) * p.row_pitch + (unsigned)(sl * p.CCs * 16);
    const int dst = ring_of(j * p.S + sl);
    for (int g0 = 0; g0 < p.slice_gran; g0 += 512) {
      int row, rem, pix, ch;
      tc_divmod(g0 + tid, p.row_gran, p.rcpRowGran, row, rem);
      tc_divmod(rem, CCs1, p.rcpCCs1, pix, ch);
      if (pix >= p.Wp) pix = 0;  // (row padding: any readable bytes)
      const unsigned so = src0 + (unsigned)row * p.row_pitch + (unsigned)pix * p.C + (unsigned)(ch < p.CCs ? ch : 0) * 16u;  // (bounds: the descriptor)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (__attribute__((address_space(3))) void*)(smem + dst + (g0 + wave * 64) * 16), 16, (int)so, 0, 0, 0);
    }
  };
  auto slice_rebias = [&](int n) {  // u8 -> s8 of the granules THIS lane fetched (they have landed: vmcnt(0) before)
    const int dst = ring_of(n);
    for (int g = tid; g < p.slice_gran; g += 512) {
      v4i* q = reinterpret_cast<v4i*>(smem + dst + g * 16);
      *q = *q ^ (int)0x80808080;
    }
  };

  v4i acc[TMW][NTW];
  const I8ieRequant rq = p.rq;
  const int lo = p.relu_lo;
  const float lof = (float)lo;

  constexpr int HT = (TMW + 1) / 2;  // row tiles per half
  v4i Alo[HT], Ahi[HT], B0[NTW], B1[NTW];
  auto k_at = [&](int kt, int ks) { return reinterpret_cast<const unsigned*>(smem + p.lds_ktab)[kt * 8 + ks * 4 + lq]; };
  auto load_A = [&](v4i (&dst)[HT], int half, int patch, unsigned koff) {
#pragma unroll
    for (int i = 0; i < HT; ++i) {
      const int mi = half * HT + i;
      if (mi < TMW) dst[i] = *reinterpret_cast<const v4i*>(smem + patch + abase[mi] + koff);
    }
  };
  auto load_B = [&](v4i (&dst)[NTW], int pass, int kt, int ks) {
    const unsigned base = ((unsigned)pass * (unsigned)p.nkt + (unsigned)kt) * (unsigned)KT_BYTES + (unsigned)((ks * (BN / 16) + ntile0) * 1024 + lane * 16);
#pragma unroll
    for (int ni = 0; ni < NTW; ++ni) dst[ni] = __builtin_bit_cast(v4i, __builtin_amdgcn_raw_buffer_load_b128(rsB, (int)(base + ni * 1024), 0, 0));
  };
  auto mfma_half = [&](const v4i (&a)[HT], const v4i (&b)[NTW], int half) {
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < HT; ++i) {
      const int mi = half * HT + i;
      if (mi < TMW && !(mi == TMW - 1 && ghost)) {
#pragma unroll
        for (int ni = 0; ni < NTW; ++ni)
          asm volatile("v_mfma_i32_16x16x64_i8 %0, %1, %2, %0" : "+v"(acc[mi][ni]) : "v"(b[ni]), "v"(a[i]));
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  };

  auto epilogue = [&](int t, int pass) {
    const int img = t / p.bands, band = t - img * p.bands;
    const int oy0 = band * p.RT;
    const int rows = p.OH - oy0 < p.RT ? p.OH - oy0 : p.RT;
    const int valid = rows * p.OW;
    const unsigned obase = ((unsigned)(img * p.OHp + oy0 + p.ob) * (unsigned)p.OWp + (unsigned)p.ob) * (unsigned)p.N;
    const int n0 = pass * BN + ntile0 * 16;
#pragma unroll
    for (int mi = 0; mi < TMW; ++mi) {
      const int pi = pix0 + mi * 16;
      const unsigned rowoff = pi < valid ? obase + reinterpret_cast<const unsigned*>(smem + p.lds_tab)[kTabPix + pi] : kRowInvalid;
      uint32_t d[NTW];
#pragma unroll
      for (int ni = 0; ni < NTW; ++ni) {
        const v4i c = acc[mi][ni];
        const int cv[4] = {c.x, c.y, c.z, c.w};
        d[ni] = i8ie_requant_pack4(cv, rq, lo, lof);
        if constexpr (ACC) {  // row = image-major pixel index (bands are whole rows), 4 consecutive features per lane
          const int col = n0 + ni * 16 + 4 * lq;
          if (pi < valid && col < p.N)
            *reinterpret_cast<v4i*>(p.acc + ((size_t)img * (size_t)(p.OH * p.OW) + (size_t)(oy0 * p.OW + pi)) * (size_t)p.N + col) = c;
        }
      }
#pragma unroll
      for (int ni = 0; ni + 1 < NTW; ni += 2) {
        // rows of 16 lanes: odd rows of d[ni] <-> even rows of d[ni + 1]: every lane then holds 8 consecutive features
        const auto sw = __builtin_amdgcn_permlane16_swap(d[ni], d[ni + 1], false, false);
        const int col = n0 + ni * 16 + 16 * (lq & 1) + 8 * (lq >> 1);
        v2u val;
        val.x = sw[0];
        val.y = sw[1];
        __builtin_amdgcn_raw_buffer_store_b64(val, rsO, (int)((col < p.N && rowoff != kRowInvalid) ? rowoff + (unsigned)col : kRowInvalid), 0, 0);
      }
      if (NTW & 1) {  // the odd last feature tile: 4 features per lane
        const int col = n0 + (NTW - 1) * 16 + 4 * lq;
        __builtin_amdgcn_raw_buffer_store_b32(d[NTW - 1], rsO, (int)((col < p.N && rowoff != kRowInvalid) ? rowoff + (unsigned)col : kRowInvalid), 0, 0);
      }
    }
  };

#if defined(I8IE_DIAG)
  unsigned long long ph[4] = {0, 0, 0, 0}, tq = 0;
  auto stamp = [&](int i) {
    if (p.dbg) {
      const unsigned long long now = __builtin_amdgcn_s_memtime();
      ph[i] += now - tq;
      tq = now;
    }
  };
#else
  auto stamp = [](int) {};  // (phase stamps exist in the diagnostic build only: tools/diag)
#endif

  // =============================== segments =============================================================
  // Segment sg of a team = (tile index sg / (2 npass), pass (sg / 2) % npass, K half sg & 1).  In the interval
  // between two barriers team 0 runs segment I, team 1 segment I - 1.  The slice first needed in interval I + 1
  // (by team 0) is requested at the start of interval I by all waves and re-biased by them at its end.
  const int seg_per_tile = 2 * p.npass;
  const int segs = nb * seg_per_tile;
  slice_fill(0, 0);
  tc_wait_vm0();
  slice_rebias(0);
  __syncthreads();
#if defined(I8IE_DIAG)
  if (p.dbg) tq = __builtin_amdgcn_s_memtime();
#endif
  for (int I = 0; I <= segs; ++I) {
    // ---- duty of this interval: the slice with first use in interval I + 1
    int duty_n = -1;
    {
      const int I1 = I + 1;
      const int j = I1 / seg_per_tile, r = I1 - j * seg_per_tile;
      if (j < nb && r < p.S) {
        duty_n = j * p.S + r;
        slice_fill(j, r);
      }
    }
    const int sg = I - team;
    if (sg >= 0 && sg < segs) {
      const int j = sg / seg_per_tile, rem = sg - j * seg_per_tile;
      const int pass = rem >> 1, part = rem & 1;
      const int patch = ring_of(p.S == 2 ? j * 2 + part : j);
      const int k_lo = part ? p.kt_split : 0, k_hi = part ? p.nkt : p.kt_split;
      if (!(p.flags & 1)) {
        if (part) __builtin_amdgcn_s_setprio(2); else __builtin_amdgcn_s_setprio(0);
      }
      if (part == 0) {  // accumulators start as oc'[j] (C = sum + oc', exact)
#pragma unroll
        for (int ni = 0; ni < NTW; ++ni) {
          const v4i o = *reinterpret_cast<const v4i*>(smem + p.lds_ocp + (pass * BN + (ntile0 + ni) * 16 + 4 * lq) * 4);
#pragma unroll
          for (int mi = 0; mi < TMW; ++mi) acc[mi][ni] = o;
        }
      }
      unsigned k0 = k_at(k_lo, 0), k1 = k_at(k_lo, 1);
      load_B(B0, pass, k_lo, 0);
      load_B(B1, pass, k_lo, 1);
      load_A(Alo, 0, patch, k0);
      load_A(Ahi, 1, patch, k0);
#pragma clang loop unroll(disable)
      for (int kt = k_lo; kt < k_hi; ++kt) {
        const int ktn = kt + 1 < k_hi ? kt + 1 : kt;  // (the last K tile of a segment re-reads itself: no branch in the body)
        mfma_half(Alo, B0, 0);
        load_A(Alo, 0, patch, k1);
        k0 = k_at(ktn, 0);
        mfma_half(Ahi, B0, 1);
        load_A(Ahi, 1, patch, k1);
        load_B(B0, pass, ktn, 0);
        k1 = k_at(ktn, 1);
        mfma_half(Alo, B1, 0);
        load_A(Alo, 0, patch, k0);
        mfma_half(Ahi, B1, 1);
        load_A(Ahi, 1, patch, k0);
        load_B(B1, pass, ktn, 1);
      }
      stamp(0);  // K loop
      if (duty_n >= 0) {
        tc_wait_vm0();
        slice_rebias(duty_n);
      }
      stamp(1);  // wait for the slice + re-bias
      if (part) {
        if (p.flags & 4) __builtin_amdgcn_s_setprio(0);  // experiment: the requantising wave yields to the other team's MFMAs
        epilogue(tile0 + j * per, pass);
        stamp(2);  // epilogue
      }
    } else if (duty_n >= 0) {
      tc_wait_vm0();
      slice_rebias(duty_n);
    }
    TC_BAR();
    stamp(3);  // barrier
  }
  tc_wait_vm0();
#if defined(I8IE_DIAG)
  if (p.dbg && lane == 0 && (wave == 0 || wave == 4)) {
    for (int i = 0; i < 4; ++i) p.dbg[blockIdx.x * 8 + (wave >> 2) * 4 + i] = ph[i];
  }
#endif
}

// ---- weights in fragment order for this kernel's K walk: [pass][kt][ks][ntile][lane][16]; K is ordered
//      (channel slice, tap, channel chunk of the slice), each slice padded with zeros to whole K tiles ----------
__global__ __launch_bounds__(256) void tconv_pack_kernel(const int8_t* __restrict__ B, int8_t* __restrict__ Bf, int64_t total16,
                                                         int Kpad, int Npad, int CC, int CCs, int Ks, int Ksp, int S, int nkt, int bn,
                                                         const int* __restrict__ perm) {
  const int nt = bn / 16;
  const int64_t gstride = (int64_t)gridDim.x * 256;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total16; e += gstride) {
    const int lane = (int)(e & 63);
    int64_t t = e >> 6;
    const int ntile = (int)(t % nt);
    t /= nt;
    const int ks = (int)(t & 1);
    t >>= 1;
    const int kt = (int)(t % nkt);
    const int pass = (int)(t / nkt);
    const int q = lane >> 4, r = lane & 15;
    const int ci = 8 * kt + 2 * q + ks;
    int sl = ci >= Ksp ? 1 : 0;
    int rr = ci - sl * Ksp;
    if (perm != nullptr) {
      sl = 0;
      rr = perm[ci] >= 0 ? perm[ci] : Ks;
    }
    const int n = pass * bn + ntile * 16 + r;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (rr < Ks && sl < S && n < Npad) {
      const int tap = rr / CCs, cc = rr - tap * CCs;
      const int chunk = tap * CC + sl * CCs + cc;  // source: K ordered (kh, kw, c)
      if (chunk * 16 < Kpad) v = *reinterpret_cast<const uint4*>(B + (size_t)n * Kpad + (size_t)chunk * 16);
    }
    reinterpret_cast<uint4*>(Bf)[e] = v;
  }
}

template <int TMW, int NTW, bool ACC>
int launch_tc_acc(i8ie_ctx* ctx, const TCArgs& a, int grid, int lds) {
  static bool raised[64] = {};
  const int dev = ctx->device & 63;
  if (!raised[dev]) {
    I8IE_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&tconv_kernel<TMW, NTW, ACC>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    raised[dev] = true;
  }
  tconv_kernel<TMW, NTW, ACC><<<grid, 512, lds, ctx->stream>>>(a);
  I8IE_LAUNCH_CHECK();
  return I8IE_OK;
}
template <int TMW, int NTW>
int launch_tc(i8ie_ctx* ctx, const TCArgs& a, int grid, int lds) {
  return a.acc != nullptr ? launch_tc_acc<TMW, NTW, true>(ctx, a, grid, lds) : launch_tc_acc<TMW, NTW, false>(ctx, a, grid, lds);
}

}  // namespace

int i8ie_tconv_try_launch(i8ie_ctx* ctx, const I8ieIgemmCall& c) {
  if (c.amode != 1 || c.biasf != nullptr || c.wcache == nullptr) return 0;
  if (i8ie_is_pool(c.pool_k, c.pool_s) || c.a_s8 || c.out_s8) return 0;  // (max-pool folding and re-biased layouts: i8ie_pconv.hip)
  if (c.acc != nullptr && ((reinterpret_cast<uintptr_t>(c.acc) & 15u) != 0 || c.N % 4 != 0)) return 0;  // (16-byte accumulator stores)
  if (c.N % 16 != 0 || c.N < 192 || c.Npad > 1024 || c.C < 32 || c.C % 32 != 0 || c.sh != c.sw) return 0;
  if ((reinterpret_cast<uintptr_t>(c.out) & 15u) != 0) return 0;
  const int P = c.OH * c.OW;
  const int n_img = c.M / P;
  if (n_img * P != c.M || n_img < 64) return 0;  // whole images, and enough of them to fill the chip
  // feature passes: 256 wide, or 192 wide when that divides N better (384 = 2 x 192)
  const int bn = (c.N % 256 == 0) ? 256 : (c.N % 192 == 0 ? 192 : 256);
  const int npass = (c.N + bn - 1) / bn;
  // output rows per tile: as many whole rows as fit 256 pixels (16 MFMA row tiles)
  int RT = 256 / c.OW;
  if (RT < 1) return 0;
  if (RT > c.OH) RT = c.OH;
  const int bands = (c.OH + RT - 1) / RT;
  const int PT = RT * c.OW;
  const int TM = (PT + 15) / 16;
  if (TM < 9) return 0;  // (small images: the tiled kernel packs several of them into a tile)
  const int TMW = TM <= 12 ? 6 : 8;
  const int CC = c.C / 16;
  const int PR = (RT - 1) * c.sh + c.KH;
  const int taps = c.KH * c.KW;
  // patch ring: whole patches in two buffers if they fit, else two channel slices per patch in three buffers
  const int fixed = npass * bn * 4 + 2 * kTabPix * 4;
  int S = 1, R = 2, CCs = CC, slice_gran = 0, Ks = 0, Ksp = 0, nkt = 0, kt_split = 0;
  const int row_par = (c.OW * c.sh) & 1;
  int row_gran = 0;
  std::vector<int> perm;  // S == 1 only
  auto plan = [&](int S_, int R_) {
    S = S_; R = R_;
    CCs = CC / S;
    const int m16 = CCs + 1;  // slice pixel pitch / 16 (odd: CCs is even)
    const int row_pad = (((c.OW * c.sh - c.Wp) * m16) % 16 + 16) % 16;
    row_gran = c.Wp * (CCs + 1) + row_pad;
    slice_gran = (PR * row_gran + 511) / 512 * 512;
    Ks = taps * CCs;
    perm.clear();
    if (S == 1) {  // pairs of equal LDS slot parity (i8ie_pconv.hip)
      std::vector<int> cls[2];
      for (int sc = 0; sc < Ks; ++sc) {
        const int tap = sc / CCs, cc = sc - tap * CCs, kh = tap / c.KW, kw = tap - kh * c.KW;
        cls[(kh * row_par + kw + cc) & 1].push_back(sc);
      }
      std::vector<int> pairs;
      for (int par = 0; par < 2; ++par)
        for (size_t i = 0; i < cls[par].size(); i += 2) {
          pairs.push_back(cls[par][i]);
          pairs.push_back(i + 1 < cls[par].size() ? cls[par][i + 1] : -1);
        }
      const int npairs = (int)pairs.size() / 2;
      nkt = (npairs + 3) / 4;
      perm.assign((size_t)nkt * 8, -1);
      for (int j = 0; j < npairs; ++j) {
        const int kt = j >> 2, w = j & 3, base = 8 * kt + (w >> 1) * 4 + (w & 1);
        perm[base] = pairs[2 * j];
        perm[base + 2] = pairs[2 * j + 1];
      }
      Ksp = nkt * 8;
    } else {
      Ksp = (Ks + 7) / 8 * 8;
      nkt = S * Ksp / 8;
    }
    return R * slice_gran * 16 + fixed + nkt * 32 <= 160 * 1024;
  };
  if (!plan(1, 2)) {
    if (CC % 4 != 0 || !plan(2, 3)) return 0;  // (slices keep a chunk count that is a multiple of 4: natural K order pairs well)
  }
  if (nkt < 4 || slice_gran >= (1 << 22)) return 0;
  static hipDeviceProp_t props[64];
  static bool have[64] = {};
  const int dev = ctx->device & 63;
  if (!have[dev]) {
    I8IE_HIP_TRY(hipGetDeviceProperties(&props[dev], ctx->device));
    have[dev] = true;
  }
  int grid = i8ie_cus(ctx, props[dev].multiProcessorCount) / 8 * 8;
  if (grid < 8) grid = 8;
  // Chosen automatically only where it measured faster than i8ie_pconv.hip: one feature pass, whole patches in the
  // ring, and enough bands per CU to amortise the idle first / last segment of the two teams (AlexNet conv2 at
  // 1000 images: 3000 bands, 2360 vs 2280 TOP/s); variant 70 forces it.
  if (i8ie_conv_variant_auto(ctx->variant) && !(npass == 1 && S == 1 && (long)n_img * bands >= 8L * grid)) return 0;
  if (S == 2) {
    kt_split = Ksp / 8;
  } else {
    // the first K half is the longer one: the wave that goes on to the epilogue leaves the MFMA pipe to the other
    // team's wave earlier (epilogue ~ 5 k cycles ~ 2-3 K tiles of the pair)
    kt_split = (nkt + 1) / 2 + 1;
#if defined(I8IE_DIAG)
    static const char* const split_env = std::getenv("I8IE_TCONV_SPLIT");  // tuning aid, read once
    if (split_env) kt_split = std::atoi(split_env);
#endif
    if (kt_split > nkt - 1) kt_split = nkt - 1;
    if (kt_split < 1) kt_split = 1;
  }
  const size_t out_pixels = (size_t)n_img * (c.OH + 2 * c.ob) * (c.OW + 2 * c.ob);
  const size_t out_bytes = out_pixels * (size_t)c.N;
  if (out_bytes >= ((size_t)1 << 31) || c.a_bytes >= ((size_t)1 << 32) - 4096) return 0;

  // ---- fragment-packed weights: once per layer and packing key, kept in the layer handle (I8ieWCache)
  // (a buffer holds [perm: nkt * 8 ints, padded to 256 B][weights])
  const int kt_bytes = bn * 128;
  const size_t perm_bytes = i8ie_align_up((size_t)nkt * 8 * sizeof(int), 256);
  const size_t bf_bytes = (size_t)npass * nkt * kt_bytes;
  const unsigned long long wkey = (2ull << 32) | (unsigned long long)(row_par | (S << 1) | (bn << 4));
  void* wbuf = c.wcache->find(wkey);
  if (wbuf == nullptr) {
    I8IE_REQUIRE(ctx->capture == nullptr, "weight re-packing inside a graph capture: run the same calls once eagerly first");
    I8IE_TRY(i8ie_malloc(ctx, perm_bytes + bf_bytes, &wbuf));
    int rc = perm.empty() ? I8IE_OK : i8ie_memcpy_h2d(ctx, wbuf, perm.data(), perm.size() * sizeof(int));
    if (rc == I8IE_OK) {
      const int64_t total16 = (int64_t)(bf_bytes / 16);
      int64_t blocks = (total16 + 255) / 256;
      if (blocks > 4096) blocks = 4096;
      tconv_pack_kernel<<<(int)blocks, 256, 0, ctx->stream>>>(c.B, (int8_t*)wbuf + perm_bytes, total16, c.Kpad, c.Npad, CC, CCs, Ks, Ksp, S, nkt, bn,
                                                               perm.empty() ? nullptr : (const int*)wbuf);
      if (hipGetLastError() != hipSuccess) rc = I8IE_ERR_HIP;
    }
    if (rc != I8IE_OK) {
      i8ie_free(ctx, wbuf);
      return rc;
    }
    c.wcache->ents.push_back(I8ieWCache::Ent{wkey, wbuf});
  }

  TCArgs a{};
  a.A = c.A;
  a.a_bytes = (unsigned)c.a_bytes;
  a.C = (unsigned)c.C;
  a.row_pitch = (unsigned)c.Wp * (unsigned)c.C;
  a.img_pitch = (unsigned)c.Hp * a.row_pitch;
  a.OH = c.OH; a.OW = c.OW; a.s = c.sh; a.KW = c.KW; a.Wp = c.Wp;
  a.rcpOW = 1.0f / (float)c.OW;
  a.RT = RT; a.bands = bands; a.n_tiles = n_img * bands;
  a.S = S; a.R = R; a.CCs = CCs; a.Ps = CCs * 16 + 16;
  a.rcpCCs1 = 1.0f / (float)(CCs + 1);
  a.slice_gran = slice_gran;
  a.row_gran = row_gran;
  a.rcpRowGran = 1.0f / (float)row_gran;
  a.perm = perm.empty() ? nullptr : (const int*)wbuf;
  a.Ks = Ks; a.Ksp = S == 2 ? Ksp : nkt * 8;
  a.Bf = (const int8_t*)wbuf + perm_bytes;
  a.bf_bytes = (unsigned)bf_bytes;
  a.nkt = nkt; a.kt_split = kt_split;
  a.N = c.N; a.npass = npass;
  a.ocp = c.ocp; a.Npad = c.Npad;
  a.rq = i8ie_make_requant(c.s_in, c.s_w, c.s_out, c.zp_out);
  a.relu_lo = c.relu ? c.zp_out : 0;
  a.out = c.out;
  a.out_bytes = (unsigned)out_bytes;
  a.ob = c.ob; a.OHp = c.OH + 2 * c.ob; a.OWp = c.OW + 2 * c.ob;
  a.acc = c.acc;
  a.lds_ring = 0;
  a.lds_ocp = R * slice_gran * 16;
  a.lds_tab = a.lds_ocp + npass * bn * 4;
  a.lds_ktab = a.lds_tab + 2 * kTabPix * 4;
  const int lds = a.lds_ktab + nkt * 32;
#if defined(I8IE_DIAG)
  if (ctx->variant == 72) a.flags = 1;
  if (ctx->variant == 73) a.flags = 2;
  if (ctx->variant == 74) a.flags = 3;
  if (ctx->variant == 75) a.flags = 4;
#endif

  const double ops = 2.0 * c.M * c.N * c.Ktrue;
  const double bytes = (double)c.M * c.Ktrue + (double)c.N * c.Ktrue + (double)c.M * c.N;
  char tag[64];
  snprintf(tag, sizeof(tag), "tconv_%dx%d|M%d,N%d,K%d", TMW * 32, bn, c.M, c.N, c.Kchunks * 16);
  char nm[32];
  snprintf(nm, sizeof(nm), "tconv_%dx%d", TMW * 32, bn);
  I8ieProfScope prof(ctx, ctx->prof ? tag : nm, ops, bytes);
#if defined(I8IE_DIAG)
  static unsigned long long* dbg_dev[64] = {};  // per device
  unsigned long long*& dbg = dbg_dev[ctx->device & 63];
  const bool stamps = ctx->variant >= 71 && ctx->variant <= 74;
  if (stamps) {
    if (!dbg) I8IE_HIP_TRY(hipMalloc(&dbg, 4096 * 8 * sizeof(unsigned long long)));
    I8IE_HIP_TRY(hipMemsetAsync(dbg, 0, 4096 * 8 * sizeof(unsigned long long), ctx->stream));
    a.dbg = dbg;
  }
#endif
  int rc;
  if (TMW == 8 && bn == 256) rc = launch_tc<8, 4>(ctx, a, grid, lds);
  else if (TMW == 6 && bn == 256) rc = launch_tc<6, 4>(ctx, a, grid, lds);
  else if (TMW == 6 && bn == 192) rc = launch_tc<6, 3>(ctx, a, grid, lds);
  else rc = launch_tc<8, 3>(ctx, a, grid, lds);
#if defined(I8IE_DIAG)
  if (rc == I8IE_OK && stamps && std::getenv("I8IE_TCONV_STAMPS") != nullptr) {
    std::vector<unsigned long long> h((size_t)grid * 8);
    I8IE_HIP_TRY(hipStreamSynchronize(ctx->stream));
    I8IE_HIP_TRY(hipMemcpy(h.data(), dbg, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    double sum[2][4] = {};
    for (int b = 0; b < grid; ++b)
      for (int t = 0; t < 2; ++t)
        for (int i = 0; i < 4; ++i) sum[t][i] += (double)h[(size_t)b * 8 + t * 4 + i];
    const double tp = (double)a.n_tiles * npass;  // tile passes in all
    for (int t = 0; t < 2; ++t)
      fprintf(stderr, "tconv_stamps v%d team %d M %d N %d K %d (%d tiles x %d passes, %d K tiles split at %d, TMW %d bn %d S %d R %d): per tile pass, cycles: K loops %.0f (%.0f per K tile) | slice wait + re-bias %.0f | epilogue %.0f | at barriers %.0f\n",
              ctx->variant, t, c.M, c.N, c.Kchunks * 16, a.n_tiles, npass, nkt, kt_split, TMW, bn, S, R, sum[t][0] / tp, sum[t][0] / tp / nkt,
              sum[t][1] / tp, sum[t][2] / tp, sum[t][3] / tp);
  }
#endif
  return rc == I8IE_OK ? 1 : rc;
}
