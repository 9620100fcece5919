// i8ie_pconv.hip -- patch-stationary implicit-GEMM Conv2d over bordered NHWC u8 activations.
//
//   C[r][j] = sum_k A_u8[r][k] * W_s8[j][k] + oc[j]     (src/conv2d.cc:131-133: cblas_gemm_s8u8s32 + oc)
//   out     = relu?(down_scale(C))                      (src/quantize_utils.cc:27-36, src/functional.cc:15-26)
//
// What round 2's measurements on MI355X said about the tiled contraction kernels:
//   * every LDS-DMA piece (1 KiB) costs a CU about 18 cycles that no schedule hid, wherever its bytes come from:
//     a 256 x 256 tile fed by im2col'd K tiles needs 64 pieces per 2048 MFMA cycles;
//   * the u8 -> s8 xor of the activation fragments is fully exposed (4-5 cycles each, 64 per K tile and wave);
//   * a barrier per K tile couples 8 waves: every stall of one is a stall of all (13 % on conv2).
// This kernel removes all three from the K loop:
//   * The A operand is not staged K tile by K tile.  The INPUT PATCH of an output band (whole output rows of one
//     image: 9 rows of AlexNet conv2, a whole 13 x 13 image of conv3-5) is copied into LDS ONCE (LDS-DMA), re-biased
//     there once (each lane xors the 16 bytes it fetched itself), and every tap of every K step reads its fragments
//     from it: pixel pitch C + 16 bytes, fragment address = base(pixel) + offset(tap, channel chunk), additive.
//     The im2col redundancy (9x / 25x) stays inside LDS.
//   * Bank conflicts without a swizzle: the MFMA row -> pixel map inside a 16-row tile is permuted (rows 0-3 and
//     12-15 take the even pixels, rows 4-11 the odd ones) and lane group q reads 16-byte chunk 2q (+1 on odd
//     k-steps): the 16 lanes ds_read_b128 serves per cycle then hit 16 different slots for any odd pitch / 16.
//   * Weights never touch LDS: they are pre-packed in MFMA fragment order for exactly that K walk, so a wave's
//     fragment is 1 KiB contiguous, read by one coalesced buffer load straight from L2 into registers, one k-step
//     ahead (weights of a layer are 0.6-1.3 MB: L2-resident; the two waves that need the same fragment ask for it
//     at about the same time).  No barrier inside the K loop: 8 waves (2 along the pixels x 4 along the features,
//     128 x 64 outputs each) run free between the tile's two hand-over points.
//   * N = 384 runs as one pass of 384 over the resident patch (wave = 96 x 96 outputs) when a band has at most 12
//     row tiles, else as two passes of 192; tiles are whole bands, so 1000 images of 13 x 13 give 1000 equal
//     tiles.  With fewer bands than CUs the passes of a band (two of 192, or two of 128 cut from a 256-wide one)
//     become units of their own; the band height is the one that needs the fewest rounds x row tiles.
//   * K chunks are paired by the parity of their LDS slot (host-built permutation, shared with the weight
//     packing), patch rows are padded to keep the 16-slot walk across row wraps: 6 % conflict cycles left.
//   * The next band's patch is requested when the last K step of a tile has been read (one barrier) and lands
//     under the epilogue; a counted vmcnt leaves the epilogue's stores in flight.  (Two alternating patch buffers,
//     where they fit, were measured and bought nothing: vmcnt retires in order, so any wait on a weight fragment
//     younger than the patch DMA waits for the DMA as well.)
// History of the design (LDS-staged weights with a barrier per K tile: 5-10 % slower; two wave teams half a tile
// apart, i8ie_tconv.hip) and the phase stamps behind these statements: DESIGN.md section 4.
#include <cstdio>
#include <cstdlib>
#include <type_traits>
#include <vector>

#include "i8ie_pconv_common.h"

namespace {

// TMW: 16-pixel MFMA tiles per wave (2 waves along the pixels); NTW: 16-feature tiles per wave (4 waves along the
// features, block = 64 NTW features per pass)
// ACC: also dump the INT32 accumulators (acc_dbg of the C-ABI) -- a separate instantiation, the default one is untouched
// POOL: max-pool folded in behind the requantiser (PCArgs::pk ...)
template <int TMW, int NTW, bool ACC, bool POOL>
__global__ __launch_bounds__(512, 2) void pconv_kernel(PCArgs p) {
  uint8_t* const smem = pc_smem;
  constexpr int BN = NTW * 64;
  constexpr int KT_BYTES = BN * 128;  // weights of one K tile of a pass: 2 k-steps x (BN / 16) fragments of 1 KiB
  constexpr int ST = TMW * ((NTW + 1) / 2);  // buffer stores of one epilogue, per wave (same for every wave)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const int lq = lane >> 4, lr = lane & 15;

  // ---- tiles of this block: XCD-contiguous ranges, consecutive tiles to the blocks of one XCD
  const int per = (int)gridDim.x >> 3;
  const int xcd = blockIdx.x & 7, jb = blockIdx.x >> 3;
  const int n_units = p.seq ? p.n_tiles / p.bands : (p.split ? p.n_tiles * p.npass : p.n_tiles);  // (seq: a unit is an image)
  const int Tx = (n_units + 7) >> 3;
  const int t_lo = xcd * Tx;
  const int t_hi = t_lo + Tx < n_units ? t_lo + Tx : n_units;
  int unit = t_lo + jb;
  if (unit >= t_hi) return;

  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(p.A), 0, p.a_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<int8_t*>(p.Bf), 0, p.bf_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsO = __builtin_amdgcn_make_buffer_rsrc(p.out, 0, p.out_bytes, 0x00020000);

  // ---- the patch of a tile -> LDS (pixel pitch P: C bytes + one pad chunk), then one xor pass (u8 -> s8)
  const int CC1 = p.CC + 1;
  // source offset of granule g relative to the patch origin: the same for every band, so it is tabulated once
  // (two float divisions per granule at every fill took 2.2-3.7 k cycles per band)
  auto src_of = [&](int g) {
    int row, rem, pix, ch;
    pc_divmod(g, p.row_gran, p.rcpRowGran, row, rem);
    pc_divmod(rem, CC1, p.rcpCC1, pix, ch);
    if (pix >= p.Wp) pix = 0;  // (row padding: any readable bytes)
    return (unsigned)row * p.row_pitch + (unsigned)pix * p.C + (unsigned)(ch < p.CC ? ch : 0) * 16u;
  };
  // The block's FIRST fill goes out before anything else (it lands while the tables below are built) and leaves the source
  // table behind as a by-product: every thread tabulates exactly the granules it will fetch again for the later bands.
  auto first_fill = [&](int t, int dst) {
    const int img = t / p.bands, band = t - img * p.bands;
    const unsigned src0 = (unsigned)img * p.img_pitch + (unsigned)(band * p.RT * p.s) * p.row_pitch;
    for (int g0 = 0; g0 < p.patch_gran; g0 += 512) {
      const unsigned rel = src_of(g0 + tid);
      if (p.lds_src >= 0) reinterpret_cast<unsigned*>(smem + p.lds_src)[g0 + tid] = rel;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (__attribute__((address_space(3))) void*)(smem + dst + (g0 + wave * 64) * 16), 16,
                                               (int)(src0 + rel), 0, 0, 0);
    }
  };
  auto patch_fill = [&](int t, int dst) {
    const int img = t / p.bands, band = t - img * p.bands;
    const unsigned src0 = (unsigned)img * p.img_pitch + (unsigned)(band * p.RT * p.s) * p.row_pitch;
    for (int g0 = 0; g0 < p.patch_gran; g0 += 512) {
      const unsigned so = src0 + (p.lds_src >= 0 ? reinterpret_cast<const unsigned*>(smem + p.lds_src)[g0 + tid] : src_of(g0 + tid));  // (bounds: the descriptor)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (__attribute__((address_space(3))) void*)(smem + dst + (g0 + wave * 64) * 16), 16,
                                               (int)so, 0, 0, 0);
    }
  };
  first_fill(p.seq ? unit * p.bands : (p.split ? unit / p.npass : unit), p.lds_patch);

  // ---- per-kernel tables in LDS: oc', and for pixel i of a tile its window origin in the patch / its output row
  for (int i = tid; i < p.npass * BN; i += 512) reinterpret_cast<int*>(smem + p.lds_ocp)[i] = i < p.Npad ? p.ocp[i] : 0;
  const int PT = p.RT * p.OW;
  for (int i = tid; i < kTabPix; i += 512) {
    int oy, ox;
    pc_divmod(i < PT ? i : 0, p.OW, p.rcpOW, oy, ox);
    reinterpret_cast<unsigned*>(smem + p.lds_tab)[i] = (unsigned)(oy * p.s) * (unsigned)(p.row_gran * 16) + (unsigned)(ox * p.s) * (unsigned)p.P;
    reinterpret_cast<unsigned*>(smem + p.lds_tab)[kTabPix + i] = (unsigned)(oy * p.OWp + ox) * (unsigned)p.N;
  }
  for (int i = tid; i < p.nkt * 8; i += 512) {
    const int ci = 8 * (i >> 3) + 2 * (i & 3) + ((i >> 2) & 1);  // K position of (K tile i / 8, k-step (i / 4) & 1, lane group i & 3)
    const int sc = p.perm[ci];
    int tap, cc, kh, kw;
    pc_divmod(sc < 0 ? 0 : sc, p.CC, 1.0f / (float)p.CC, tap, cc);
    pc_divmod(tap, p.KW, 1.0f / (float)p.KW, kh, kw);
    reinterpret_cast<unsigned*>(smem + p.lds_ktab)[i] = sc >= 0 ? (unsigned)kh * (unsigned)(p.row_gran * 16) + (unsigned)kw * (unsigned)p.P + (unsigned)cc * 16u : 0u;
  }
  __syncthreads();
  // this lane's rows: tile (wm * TMW + mi), MFMA row lr -> pixel index inside the band
  const int pix0 = wm * TMW * 16 + pc_row_to_pix(lr);
  unsigned abase[TMW];
#pragma unroll
  for (int mi = 0; mi < TMW; ++mi) {
    const int pi = pix0 + mi * 16;
    abase[mi] = reinterpret_cast<const unsigned*>(smem + p.lds_tab)[pi < kTabPix ? pi : 0];
  }
  const int colb = wn * (NTW * 16);  // first feature of this wave inside the pass

  auto patch_xor = [&](int dst) {  // every lane re-biases exactly the granules it fetched itself
    if (p.a_s8) return;            // (the producer stored the bytes re-biased: nothing to do)
    for (int g = tid; g < p.patch_gran; g += 512) {
      v4i* q = reinterpret_cast<v4i*>(smem + dst + g * 16);
      *q = *q ^ (int)0x80808080;
    }
  };
  v4i acc[TMW][NTW];
  const I8ieRequant rq = p.rq;
  const int lo = p.relu_lo;
  const float lof = (float)lo;

  // this lane's walk over K: chunk ci = 8 kt + 2 q + ks -> tap ci / CC = (kh, kw), channel chunk ci % CC; the
  // patch offsets of all (kt, ks, q) sit in an LDS table (chunks past K: offset 0, their weights are zero)
  auto k_at = [&](int kt, int ks) { return reinterpret_cast<const unsigned*>(smem + p.lds_ktab)[kt * 8 + ks * 4 + lq]; };
  // the last row tile of the upper wave row may not exist (169 pixels = 10.6 tiles): skipped, wave-uniformly
  const bool ghost = (wm * TMW + TMW - 1) * 16 >= PT;

  constexpr int HT = (TMW + 1) / 2;  // row tiles per half (fragments are fetched half a k-step ahead)
  v4i Alo[2][HT], Ahi[HT], Bq[2][NTW];
  auto load_A = [&](v4i (&dst)[HT], int half, int patch, unsigned koff) {
#pragma unroll
    for (int i = 0; i < HT; ++i) {
      const int mi = half * HT + i;
      if (mi < TMW) dst[i] = *reinterpret_cast<const v4i*>(smem + patch + abase[mi] + koff);
    }
  };
  auto load_B = [&](v4i (&dst)[NTW], int lin, int ks) {  // lin = pass * nkt + kt
    const unsigned base = (unsigned)lin * (unsigned)KT_BYTES + (unsigned)((ks * (BN / 16) + wn * NTW) * 1024 + lane * 16);
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
    const int n0 = pass * BN + colb;
#pragma unroll
    for (int mi = 0; mi < TMW; ++mi) {
      const int pi = pix0 + mi * 16;
      const unsigned rowoff = pi < valid ? obase + reinterpret_cast<const unsigned*>(smem + p.lds_tab)[kTabPix + pi] : kRowInvalid;
      uint32_t d[NTW];
#pragma unroll
      for (int ni = 0; ni < NTW; ++ni) {
        const int c4[4] = {acc[mi][ni].x, acc[mi][ni].y, acc[mi][ni].z, acc[mi][ni].w};
        if constexpr (POOL) d[ni] = i8ie_requant_pack4_norelu(c4, rq);  // (the ReLU follows the pool: pool_pass)
        else d[ni] = i8ie_requant_pack4(c4, rq, lo, lof);
        if constexpr (ACC) {  // row = image-major pixel index (bands are whole rows), 4 consecutive features per lane
          const int col = n0 + ni * 16 + 4 * lq;
          if (pi < valid && col < p.N)
            *reinterpret_cast<v4i*>(p.acc + ((size_t)img * (size_t)(p.OH * p.OW) + (size_t)(oy0 * p.OW + pi)) * (size_t)p.N + col) = acc[mi][ni];
        }
      }
      if constexpr (POOL) {
        // requantised bytes -> the LDS ring of conv rows: pixel L = slot0 * OW + pi of a ring of RB * OW pixels (rows of a
        // band are contiguous in it; only the ring's end wraps).  16 lanes = 16 different pixels, 272-byte pitch:
        // conflict-free ds_write_b64
        int L = (oy0 % p.RB) * p.OW + pi;
        if (L >= p.RB * p.OW) L -= p.RB * p.OW;
        uint8_t* const orow = smem + p.lds_otile + L * p.opitch;
#pragma unroll
        for (int ni = 0; ni + 1 < NTW; ni += 2) {
          const auto sw = __builtin_amdgcn_permlane16_swap(d[ni], d[ni + 1], false, false);
          const int col = n0 + ni * 16 + 16 * (lq & 1) + 8 * (lq >> 1);
          v2u val;
          val.x = sw[0];
          val.y = sw[1];
          if (pi < valid && col < p.N) *reinterpret_cast<v2u*>(orow + col) = val;
        }
        if (NTW & 1) {
          const int col = n0 + (NTW - 1) * 16 + 4 * lq;
          if (pi < valid && col < p.N) *reinterpret_cast<uint32_t*>(orow + col) = d[NTW - 1];
        }
      } else {
#pragma unroll
        for (int ni = 0; ni + 1 < NTW; ni += 2) {
          // rows of 16 lanes: odd rows of d[ni] <-> even rows of d[ni + 1]: every lane then holds 8 consecutive features
          const auto sw = __builtin_amdgcn_permlane16_swap(d[ni], d[ni + 1], false, false);
          const int col = n0 + ni * 16 + 16 * (lq & 1) + 8 * (lq >> 1);
          v2u val;
          val.x = sw[0] ^ p.xor_out;
          val.y = sw[1] ^ p.xor_out;
          __builtin_amdgcn_raw_buffer_store_b64(val, rsO, (int)((col < p.N && rowoff != kRowInvalid) ? rowoff + (unsigned)col : kRowInvalid), 0, 0);
        }
        if (NTW & 1) {  // the odd last feature tile: 4 features per lane
          const int col = n0 + (NTW - 1) * 16 + 4 * lq;
          __builtin_amdgcn_raw_buffer_store_b32(d[NTW - 1] ^ p.xor_out, rsO, (int)((col < p.N && rowoff != kRowInvalid) ? rowoff + (unsigned)col : kRowInvalid), 0, 0);
        }
      }
    }
  };

  // ---- POOL: pooled rows [j0, j1) of image img from the LDS ring -> the output tensor.  Thread = (pooled row, pooled
  //      pixel, 16 features): pk x pk ds_read_b128, byte maxima as two v_pk_max_u16 per dword (even / odd bytes spread
  //      by v_perm_b32), one 16-byte store.  Returns the number of store instructions this wave issued.
  auto pool_pass = [&](int img, int j0, int j1, int c16_lo, int C16) {  // (features [16 c16_lo, 16 (c16_lo + C16)): all of them,
                                                                         // or those of the unit's own pass with split)
    const int tasks = (j1 - j0) * p.PW * C16;
    const int PHp = p.PH + 2 * p.ob, PWp = p.PW + 2 * p.ob;
    int nst = 0;
    for (int id0 = 0; id0 < tasks; id0 += 512) {
      const int id = id0 + tid;
      if (id0 + wave * 64 < tasks) ++nst;  // (this wave executes the store below: at least its first lane has a task)
      if (id < tasks) {
        int t1, c16, jr, px;
        pc_divmod(id, C16, p.rcpC16, t1, c16);
        c16 += c16_lo;
        pc_divmod(t1, p.PW, p.rcpPW, jr, px);
        const int j = j0 + jr;
        typedef unsigned short us2 __attribute__((ext_vector_type(2)));
        // maxima start at the ReLU's lower bound (0 without one): max-pool and max(., zp_out) commute, so the epilogue's
        // requantiser runs without its clamp (src/functional.cc:15-26 behind src/functional.cc:36-64, either order)
        const unsigned short lo16 = (unsigned short)lo;
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

  // =============================== tile loop ===========================================================
#if defined(I8IE_DIAG)
  unsigned long long ph[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tq = 0;
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
  const int lin_total = p.npass * p.nkt;
  const int patch = p.lds_patch;
  const bool xpre = (p.flags & 1) != 0;  // the last K tile of a pass already fetches the first weights of the next pass / band (+1 %)
  // The block's tiles, in order.  seq (POOL with several bands per image): a unit is an image and its bands follow each
  // other here; otherwise a unit is a band (or a (band, pass) pair with split).
  int band = 0;
  auto tile_at = [&](int u, int b) { return p.seq ? u * p.bands + b : (p.split ? u / p.npass : u); };
  auto has_next = [&](int u, int b) { return (p.seq && b + 1 < p.bands) || u + per < t_hi; };
  auto next_tile = [&](int u, int b) { return (p.seq && b + 1 < p.bands) ? tile_at(u, b + 1) : tile_at(u + per, 0); };
  pc_wait_vm<0>();  // (the first patch: requested at the top of the kernel)
  patch_xor(patch);
  if (xpre) {
    load_B(Bq[0], 0, 0);
    load_B(Bq[1], 0, 1);
  }
  __syncthreads();
#if defined(I8IE_DIAG)
  if (p.dbg) tq = __builtin_amdgcn_s_memtime();
#endif
  while (unit < t_hi) {
    const bool more = has_next(unit, band);
    const int tile = tile_at(unit, band);
    const int pass_lo = p.split ? unit - tile * p.npass : 0, pass_hi = p.split ? pass_lo + 1 : p.npass;
    for (int pass = pass_lo; pass < pass_hi; ++pass) {
      int lin = pass * p.nkt;
      if (!xpre) {
        load_B(Bq[0], lin, 0);
        load_B(Bq[1], lin, 1);
      }
      // accumulators start as oc'[j] (C = sum + oc', exact)
#pragma unroll
      for (int ni = 0; ni < NTW; ++ni) {
        const v4i o = *reinterpret_cast<const v4i*>(smem + p.lds_ocp + (pass * BN + colb + ni * 16 + 4 * lq) * 4);
#pragma unroll
        for (int mi = 0; mi < TMW; ++mi) acc[mi][ni] = o;
      }
      // K loop: A fragments half a k-step ahead (LDS), weight fragments one k-step ahead (L2)
      unsigned k0 = k_at(0, 0), k1;
      load_A(Alo[0], 0, patch, k0);
#pragma clang loop unroll(disable)
      for (int kt = 0; kt < p.nkt; ++kt, ++lin) {
        // (Measured and dropped: equalising the two waves of a SIMD -- priority turns per K tile, or the wave behind,
        // by a K-tile count published in LDS, at the higher priority.  Without it the older wave finishes its K
        // loop 8 k cycles early (37 k vs 45 k on conv2) and waits at the hand-over barrier; with it both take 46 k:
        // the pair's combined rate is what is limited, 85-88 % of the MFMA rate, and conv5 lost 4 %.)
        const int ktn = kt + 1 < p.nkt ? kt + 1 : kt;
        const int linn = kt + 1 < p.nkt ? lin + 1 : (xpre ? (lin + 1 < lin_total ? lin + 1 : 0) : lin);  // (xpre: not with split)
        k1 = k_at(kt, 1);
        load_A(Ahi, 1, patch, k0);
        mfma_half(Alo[0], Bq[0], 0);
        load_A(Alo[1], 0, patch, k1);
        k0 = k_at(ktn, 0);
        mfma_half(Ahi, Bq[0], 1);
        load_B(Bq[0], linn, 0);
        load_A(Ahi, 1, patch, k1);
        mfma_half(Alo[1], Bq[1], 0);
        load_A(Alo[0], 0, patch, k0);
        mfma_half(Ahi, Bq[1], 1);
        load_B(Bq[1], linn, 1);
      }
      stamp(0);  // K loop
      if (pass + 1 < pass_hi) {
        epilogue(tile, pass);
        stamp(1);
      }
    }
    // ---- hand-over: everyone is done with the patch -> request the next one -> it lands under the epilogue
    __syncthreads();
    stamp(2);  // waiting for the slowest wave's K loop
    if (more) patch_fill(next_tile(unit, band), patch);
    stamp(3);  // issuing the patch DMA
    epilogue(tile, pass_hi - 1);
    stamp(1);
    if constexpr (POOL) {
      // the band's requantised rows are in the LDS ring: pool what they complete, under the landing patch
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // (not __syncthreads(): the patch DMA and the ACC stores stay in flight)
      const int img = tile / p.bands, bnd = tile - img * p.bands;
      const int oy0 = bnd * p.RT;
      const int hi = oy0 + p.RT < p.OH ? oy0 + p.RT : p.OH;
      const int j0 = rows_done(oy0), j1 = rows_done(hi);
      const int c16n = p.split ? BN / 16 : p.N >> 4;  // (rcpC16 is the reciprocal of this count)
      const int nst = j1 > j0 ? pool_pass(img, j0, j1, p.split ? pass_lo * (BN / 16) : 0, c16n) : 0;
      if (more) {
        // vector-memory operations retire in issue order: the patch DMA is older than the pool pass's stores (and, with
        // ACC, younger than nothing that matters: ACC kernels just drain)
        if (ACC) pc_wait_vm<0>();
        else switch (nst) {
          case 0: pc_wait_vm<0>(); break;
          case 1: pc_wait_vm<1>(); break;
          case 2: pc_wait_vm<2>(); break;
          case 3: pc_wait_vm<3>(); break;
          default: pc_wait_vm<4>(); break;
        }
        stamp(4);
        patch_xor(patch);
        stamp(5);
      }
    } else if (more) {
      pc_wait_vm<ST>();  // all but the epilogue's stores
      stamp(4);          // patch DMA not yet landed after the epilogue
      patch_xor(patch);
      stamp(5);          // re-bias pass
    }
    __syncthreads();
    stamp(6);  // second barrier
    if (p.seq && band + 1 < p.bands) {
      ++band;
    } else {
      band = 0;
      unit += per;
    }
  }
  pc_wait_vm<0>();
#if defined(I8IE_DIAG)
  if (p.dbg && lane == 0)
    for (int i = 0; i < 7; ++i) p.dbg[((size_t)blockIdx.x * 8 + wave) * 8 + i] = ph[i];
#endif
}

// ---- weights in fragment order for this kernel's K walk: [pass][kt][ks][ntile][lane][16] --------------------
__global__ __launch_bounds__(256) void pconv_pack_kernel(const int8_t* __restrict__ B, int8_t* __restrict__ Bf, int64_t total16,
                                                         int Kpad, int Npad, const int* __restrict__ perm, int nkt, int bn) {
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
    const int chunk = perm[8 * kt + 2 * q + ks];
    const int n = pass * bn + ntile * 16 + r;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (chunk >= 0 && chunk * 16 < Kpad && n < Npad) v = *reinterpret_cast<const uint4*>(B + (size_t)n * Kpad + (size_t)chunk * 16);
    reinterpret_cast<uint4*>(Bf)[e] = v;
  }
}

template <int TMW, int NTW, bool ACC, bool POOL>
int launch_pc_t(i8ie_ctx* ctx, const PCArgs& a, int grid, int lds) {
  static bool raised[64] = {};
  const int dev = ctx->device & 63;
  if (!raised[dev]) {
    I8IE_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&pconv_kernel<TMW, NTW, ACC, POOL>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    raised[dev] = true;
  }
  pconv_kernel<TMW, NTW, ACC, POOL><<<grid, 512, lds, ctx->stream>>>(a);
  I8IE_LAUNCH_CHECK();
  return I8IE_OK;
}
template <int TMW, int NTW>
int launch_pc(i8ie_ctx* ctx, const PCArgs& a, int grid, int lds) {
  if (a.pk > 1)
    return a.acc != nullptr ? launch_pc_t<TMW, NTW, true, true>(ctx, a, grid, lds) : launch_pc_t<TMW, NTW, false, true>(ctx, a, grid, lds);
  return a.acc != nullptr ? launch_pc_t<TMW, NTW, true, false>(ctx, a, grid, lds) : launch_pc_t<TMW, NTW, false, false>(ctx, a, grid, lds);
}

}  // namespace

static int pconv_impl(i8ie_ctx* ctx, const I8ieIgemmCall& c, bool dry);
int i8ie_pconv_try_launch(i8ie_ctx* ctx, const I8ieIgemmCall& c) { return pconv_impl(ctx, c, false); }
// would i8ie_pconv_try_launch take this call (incl. its max-pool)?  Nothing is launched or packed.
int i8ie_pconv_takes(i8ie_ctx* ctx, const I8ieIgemmCall& c) { return pconv_impl(ctx, c, true); }

static int pconv_impl(i8ie_ctx* ctx, const I8ieIgemmCall& c, bool dry) {
  if (c.amode != 1 || c.biasf != nullptr || c.wcache == nullptr) return 0;
  if (c.pool_k == 1 && c.pool_s > 1) return 0;  // (a subsampling 1 x 1 pool is a pool, i8ie_is_pool, and this kernel does not fold it)
  const bool pool = c.pool_k > 1;
  if (pool && (c.pool_k > 3 || c.pool_s < 1 || c.pool_s > c.pool_k || c.pool_k > c.OH || c.pool_k > c.OW || c.N % 16 != 0)) return 0;
  if (c.acc != nullptr && ((reinterpret_cast<uintptr_t>(c.acc) & 15u) != 0 || c.N % 4 != 0)) return 0;  // (16-byte accumulator stores)
  if (c.N % 16 != 0 || c.N < 192 || c.Npad > 1024 || c.C < 32 || c.C % 32 != 0 || c.sh != c.sw) return 0;
  if ((reinterpret_cast<uintptr_t>(c.out) & 15u) != 0) return 0;
  const int P = c.OH * c.OW;
  const int n_img = c.M / P;
  if (n_img * P != c.M || n_img < 64) return 0;  // whole images, and enough of them to fill the chip
  // feature passes: 256 wide; N = 384 k: 384 wide (wave = 96 x 96 outputs, 144 accumulator registers) when the band
  // has at most 12 row tiles, else 192 wide (two passes).  One pass of 384 instead of two of 192 halves the A
  // fragment reads per MFMA and the per-pass overhead (variant 54 keeps two passes, for comparison).
  // Output rows per tile: whole rows, at most 256 pixels (16 MFMA row tiles), at least 129 (9 row tiles; smaller
  // images go to the tiled kernel, which packs several of them into a tile).  Among those, the band height that
  // needs the fewest (rounds over the CUs) x (row tiles per wave): 125 images of 27 x 27 are 375 bands of 9 rows
  // (2 rounds of 8 row tiles) or 500 bands of 7 (2 rounds of 6)
  static hipDeviceProp_t props[64];
  static bool have[64] = {};
  const int dev = ctx->device & 63;
  if (!have[dev]) {
    I8IE_HIP_TRY(hipGetDeviceProperties(&props[dev], ctx->device));
    have[dev] = true;
  }
  int grid = i8ie_cus(ctx, props[dev].multiProcessorCount) / 8 * 8;
  if (grid < 8) grid = 8;
  auto pick_rows = [&](int npass_, bool narrow_only) {
    int best_rt = 0;
    int rt_max = 256 / c.OW;
    if (rt_max > c.OH) rt_max = c.OH;
    double best = 0;
    for (int rt = rt_max; rt >= 1; --rt) {
      const int tm = (rt * c.OW + 15) / 16;
      if (tm < 9) break;
      if (narrow_only && tm > 12) continue;
      const int bd = (c.OH + rt - 1) / rt;
      long units = (long)n_img * bd;
      if (units < grid && npass_ > 1) units *= npass_;
      const long rounds = (units + grid - 1) / grid;
      const double cost = (double)rounds * ((tm <= 12 ? 6 : 8) + 1.5) * ((units < grid && npass_ > 1) ? 1 : npass_);
      if (best_rt == 0 || cost < best * 0.97) {
        best_rt = rt;
        best = cost;
      }
    }
    return best_rt;
  };
  int bn = (c.N % 256 == 0) ? 256 : (c.N % 192 == 0 ? 192 : 256);
  int npass = (c.N + bn - 1) / bn;
  int RT = pick_rows(npass, false);
  if (RT == 0) return 0;
  if (c.N % 384 == 0 && ctx->variant != 54) {
    const int rt6 = pick_rows(c.N / 384, true);
    // (only when 384-wide bands fill at least 3/4 of a whole-chip round: one such round costs 1.0, two rounds of 192-wide
    // (band, pass) units 2 x 0.62 (250 images of 13 x 13: 0.105 -> 0.085 ms for conv3 + conv4); with fewer bands the two passes
    // are units of their own and fit one round)
    if (rt6 != 0 && (long)n_img * ((c.OH + rt6 - 1) / rt6) * 4 >= (long)grid * 3 && (rt6 * c.OW + 15) / 16 <= 12 &&
        ((RT * c.OW + 15) / 16 <= 12)) {
      bn = 384;
      npass = c.N / 384;
      RT = rt6;
    }
  }
  // Few bands and a single 256-wide pass (conv5 of a 125-image shard: 125 bands on 256 CUs): two passes of 128 instead,
  // each (band, pass) a unit of its own, so that every CU has work (variant 54 keeps the wide pass)
  if (npass == 1 && bn == 256 && c.N % 128 == 0 && ctx->variant != 54 && (RT * c.OW + 15) / 16 <= 12 &&
      (long)n_img * ((c.OH + RT - 1) / RT) * 4 < (long)grid * 3) {
    bn = 128;
    npass = c.N / 128;
  }
  const int bands = (c.OH + RT - 1) / RT;
  const int PT = RT * c.OW;
  const int TM = (PT + 15) / 16;
  const int TMW = TM <= 12 ? 6 : 8;
  const int CC = c.C / 16, Pp = c.C + 16;
  const int PR = (RT - 1) * c.sh + c.KH;
  const int m16 = Pp / 16;
  const int row_pad = (((c.OW * c.sh - c.Wp) * m16) % 16 + 16) % 16;  // see PCArgs::row_gran
  const int row_gran = c.Wp * (CC + 1) + row_pad;
  const int patch_gran = (PR * row_gran + 511) / 512 * 512;
  // K order: the lanes one ds_read_b128 cycle serves come from two lane groups (q, q + 1), which read the K chunks at
  // positions ci and ci + 2 of a K tile.  Their 256-byte slots stay disjoint when the two chunks' LDS offsets
  // have the same parity in 16-byte units (even pixels + even offset vs odd pixels + even offset).  Pair the
  // chunks accordingly: with 96 channels (6 chunks per tap) the natural order pairs chunks of different taps,
  // an odd distance apart, a third of the time (PMC: 24 % of LDS-active cycles were bank conflicts on conv2).
  const int kchunks = c.KH * c.KW * CC;
  const int row_par = (c.OW * c.sh) & 1;  // parity of the LDS row pitch / 16 (see PCArgs::row_gran; P / 16 is odd)
  std::vector<int> perm;
  {
    std::vector<int> cls[2];
    for (int sc = 0; sc < kchunks; ++sc) {
      const int tap = sc / CC, cc = sc - tap * CC, kh = tap / c.KW, kw = tap - kh * c.KW;
      cls[(kh * row_par + kw + cc) & 1].push_back(sc);
    }
    std::vector<int> pairs;  // two entries per pair
    for (int par = 0; par < 2; ++par)
      for (size_t i = 0; i < cls[par].size(); i += 2) {
        pairs.push_back(cls[par][i]);
        pairs.push_back(i + 1 < cls[par].size() ? cls[par][i + 1] : -1);
      }
    const int npairs = (int)pairs.size() / 2;
    const int nkt_ = (npairs + 3) / 4;
    perm.assign((size_t)nkt_ * 8, -1);
    for (int j = 0; j < npairs; ++j) {  // pair j of K tile kt: (q, k-step) = (0|1, 0), (0|1, 1), (2|3, 0), (2|3, 1)
      const int kt = j >> 2, w = j & 3, base = 8 * kt + (w >> 1) * 4 + (w & 1);
      perm[base] = pairs[2 * j];
      perm[base + 2] = pairs[2 * j + 1];
    }
  }
  const int nkt = (int)perm.size() / 8;
  if (nkt < 2 || patch_gran >= (1 << 22)) return 0;
  const int PHo = pool ? (c.OH - c.pool_k) / c.pool_s + 1 : c.OH, PWo = pool ? (c.OW - c.pool_k) / c.pool_s + 1 : c.OW;
  const size_t out_pixels = (size_t)n_img * (PHo + 2 * c.ob) * (PWo + 2 * c.ob);
  const size_t out_bytes = out_pixels * (size_t)c.N;
  if (out_bytes >= ((size_t)1 << 31) || c.a_bytes >= ((size_t)1 << 32) - 4096) return 0;
  // LDS plan: [patch] [oc'] [tables] ([source offsets]) ([POOL: ring of requantised conv rows])
  const int kt_bytes = bn * 128;
  const int RB = bands > 1 ? RT + c.pool_k - 1 : c.OH;  // (rows a window shares with the band before stay in the ring)
  const int opitch = c.N + 16;
  const int otile = pool ? (RB * c.OW * opitch + 15) / 16 * 16 : 0;
  const int fixed = npass * bn * 4 + 2 * kTabPix * 4 + nkt * 32 + 64 + otile;
  if (patch_gran * 16 + fixed > 160 * 1024) return 0;
  const bool src_tab = patch_gran * 16 + fixed + patch_gran * 4 <= 160 * 1024;

  // One block per CU walks whole bands: with fewer bands than CUs, the passes of a band become units of their own;
  // below 3/4 of the CUs even then, the tiled kernel (finer tiles, two blocks per CU) is the faster one (measured
  // at 125 images: conv3/conv4 0.028/0.038 ms tiled vs 0.035/0.048 ms here) unless the caller forces this kernel.
  const int n_tiles = n_img * bands;
  const int split = (n_tiles < grid && npass > 1) ? 1 : 0;
  if (i8ie_conv_variant_auto(ctx->variant) && n_tiles * (split ? npass : 1) < grid * 3 / 4) return 0;
  const int seq = (pool && bands > 1) ? 1 : 0;  // the bands of an image back to back in one block
  // pooling wants whole images per block: bands of an image back to back (seq), or -- whole-image bands whose feature passes
  // are units of their own (conv5 of a 125-image shard) -- every unit pooling its own features
  if (pool && split && (bands > 1 || c.N % bn != 0)) return 0;
  if (pool && !split && i8ie_conv_variant_auto(ctx->variant) && n_img < grid * 3 / 4) return 0;
  if (dry) return 1;

  // The deferred-epilogue form (tools/diag/csrc/i8ie_dconv.hip, diagnostic build, variant 55: one wave per SIMD, the requantiser
  // of one accumulator set inside the other set's K loop).  Round 4 measured it (profiles/r04_dconv_power_limit.txt): 20 % fewer cycles per band
  // than this file's kernel and 3.0 against 2.6 POPS on constant operands -- but on data whose bits toggle (random bytes, and
  // the AlexNet step's activations) the chip sits at its power cap either way, its clock falls as the instruction stream gets
  // denser (1.6 GHz there, 1.9 here) and the step got SLOWER (conv2 + pool 0.469 against 0.405 ms).
#if defined(I8IE_DIAG)
  const bool dconv = ctx->variant == 55 && i8ie_dconv_eligible(split, nkt, npass, patch_gran, PT, bn, pool, c.N);
#else
  const bool dconv = false;
#endif

  // ---- fragment-packed weights: once per layer and packing key, kept in the layer handle (I8ieWCache)
  // (a buffer holds [perm: nkt * 8 ints, padded to 256 B][weights]; the fragment order depends on the pass width)
  const size_t perm_bytes = i8ie_align_up((size_t)nkt * 8 * sizeof(int), 256);
  const size_t bf_bytes = (size_t)npass * nkt * kt_bytes;
  const unsigned long long wkey = (1ull << 32) | (unsigned long long)(row_par | (bn << 1));
  void* wbuf = dconv ? nullptr : c.wcache->find(wkey);
  if (wbuf == nullptr && !dconv) {
    I8IE_REQUIRE(ctx->capture == nullptr, "weight re-packing inside a graph capture: run the same calls once eagerly first");
    I8IE_TRY(i8ie_malloc(ctx, perm_bytes + bf_bytes, &wbuf));
    int rc = i8ie_memcpy_h2d(ctx, wbuf, perm.data(), perm.size() * sizeof(int));
    if (rc == I8IE_OK) {
      const int64_t total16 = (int64_t)(bf_bytes / 16);
      int64_t blocks = (total16 + 255) / 256;
      if (blocks > 4096) blocks = 4096;
      pconv_pack_kernel<<<(int)blocks, 256, 0, ctx->stream>>>(c.B, (int8_t*)wbuf + perm_bytes, total16, c.Kpad, c.Npad, (const int*)wbuf, nkt, bn);
      if (hipGetLastError() != hipSuccess) rc = I8IE_ERR_HIP;
    }
    if (rc != I8IE_OK) {
      i8ie_free(ctx, wbuf);
      return rc;
    }
    c.wcache->ents.push_back(I8ieWCache::Ent{wkey, wbuf});
  }

  PCArgs a{};
  a.A = c.A;
  a.a_bytes = (unsigned)c.a_bytes;
  a.C = (unsigned)c.C;
  a.row_pitch = (unsigned)c.Wp * (unsigned)c.C;
  a.img_pitch = (unsigned)c.Hp * a.row_pitch;
  a.OH = c.OH; a.OW = c.OW; a.s = c.sh; a.KH = c.KH; a.KW = c.KW; a.CC = CC; a.Wp = c.Wp;
  a.rcpOW = 1.0f / (float)c.OW;
  a.rcpCC1 = 1.0f / (float)(CC + 1);
  a.RT = RT; a.bands = bands; a.n_tiles = n_img * bands;
  a.P = Pp;
  a.patch_gran = patch_gran;
  a.row_gran = row_gran;
  a.rcpRowGran = 1.0f / (float)row_gran;
  if (!dconv) {
    a.Bf = (const int8_t*)wbuf + perm_bytes;
    a.perm = (const int*)wbuf;
    a.bf_bytes = (unsigned)bf_bytes;
  }
  a.nkt = nkt;
  a.N = c.N; a.npass = npass;
  a.ocp = c.ocp; a.Npad = c.Npad;
  a.rq = i8ie_make_requant(c.s_in, c.s_w, c.s_out, c.zp_out);
  a.relu_lo = c.relu ? c.zp_out : 0;
  a.out = c.out;
  a.out_bytes = (unsigned)out_bytes;
  a.ob = c.ob; a.OHp = c.OH + 2 * c.ob; a.OWp = c.OW + 2 * c.ob;
  a.split = split;
  a.acc = c.acc;
  a.a_s8 = c.a_s8;
  a.xor_out = c.out_s8 ? 0x80808080u : 0u;
  a.pk = pool ? c.pool_k : 1; a.ps = pool ? c.pool_s : 1; a.PH = PHo; a.PW = PWo; a.RB = RB; a.opitch = opitch; a.seq = seq;
  a.rcpPW = 1.0f / (float)PWo; a.rcpC16 = 1.0f / (float)(split ? bn / 16 : (c.N / 16 > 0 ? c.N / 16 : 1)); a.rcpRB = 1.0f / (float)RB;
  a.flags = !split ? 1 : 0;
#if defined(I8IE_DIAG)
  if (ctx->variant == 53) a.flags = 0;  // (weights fetched at the start of every pass)
#endif
  a.lds_patch = 0;
  a.lds_ocp = patch_gran * 16;
  a.lds_tab = a.lds_ocp + npass * bn * 4;
  a.lds_ktab = a.lds_tab + 2 * kTabPix * 4;
  a.lds_prog = a.lds_ktab + nkt * 32;
  a.lds_src = src_tab ? a.lds_prog + 64 : -1;
  a.lds_otile = a.lds_prog + 64 + (src_tab ? patch_gran * 4 : 0);
  const int lds = a.lds_otile + otile;

  const double ops = 2.0 * c.M * c.N * c.Ktrue;
  const double bytes = (double)c.M * c.Ktrue + (double)c.N * c.Ktrue + (double)c.M * c.N;
  char tag[64];
  snprintf(tag, sizeof(tag), "pconv%s_%dx%d|M%d,N%d,K%d", pool ? "_pool" : "", TMW * 32, bn, c.M, c.N, c.Kchunks * 16);
  char nm[32];
  snprintf(nm, sizeof(nm), "pconv%s_%dx%d", pool ? "_pool" : "", TMW * 32, bn);
#if defined(I8IE_DIAG)
  if (dconv) {
    tag[0] = 'd';  // "dconv..."
    nm[0] = 'd';
    const int rcd = i8ie_dconv_launch(ctx, c, a, perm.data(), PT, bn, grid, lds, ctx->prof ? tag : nm);
    return rcd == I8IE_OK ? 1 : rcd;
  }
#endif
  I8ieProfScope prof(ctx, ctx->prof ? tag : nm, ops, bytes);
#if defined(I8IE_DIAG)
  static unsigned long long* dbg_dev[64] = {};  // per device
  unsigned long long*& dbg = dbg_dev[ctx->device & 63];
  if (ctx->variant == 51) {
    if (!dbg) I8IE_HIP_TRY(hipMalloc(&dbg, 4096 * 64 * sizeof(unsigned long long)));
    I8IE_HIP_TRY(hipMemsetAsync(dbg, 0, 4096 * 64 * sizeof(unsigned long long), ctx->stream));
    a.dbg = dbg;
  }
#endif
  int rc;
  if (TMW == 8 && bn == 256) rc = launch_pc<8, 4>(ctx, a, grid, lds);
  else if (TMW == 6 && bn == 256) rc = launch_pc<6, 4>(ctx, a, grid, lds);
  else if (TMW == 6 && bn == 192) rc = launch_pc<6, 3>(ctx, a, grid, lds);
  else if (TMW == 6 && bn == 384) rc = launch_pc<6, 6>(ctx, a, grid, lds);
  else if (TMW == 6 && bn == 128) rc = launch_pc<6, 2>(ctx, a, grid, lds);
  else rc = launch_pc<8, 3>(ctx, a, grid, lds);
#if defined(I8IE_DIAG)
  if (rc == I8IE_OK && ctx->variant == 51 && std::getenv("I8IE_PCONV_STAMPS") != nullptr) {
    std::vector<unsigned long long> h((size_t)grid * 64);
    I8IE_HIP_TRY(hipStreamSynchronize(ctx->stream));
    I8IE_HIP_TRY(hipMemcpy(h.data(), dbg, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    double sum[8][7] = {};
    for (int b = 0; b < grid; ++b)
      for (int w = 0; w < 8; ++w)
        for (int i = 0; i < 7; ++i) sum[w][i] += (double)h[((size_t)b * 8 + w) * 8 + i];
    const double tp = (double)a.n_tiles * npass;  // band passes in all
    fprintf(stderr, "pconv_stamps v%d M %d N %d K %d (%d tiles x %d passes, %d K tiles, TMW %d bn %d): per tile pass, cycles (wave 0): K loop %.0f (%.0f per K tile) | epilogue %.0f | hand-over: barrier after the K loop %.0f, DMA issue %.0f, DMA wait after the epilogue %.0f, re-bias %.0f, barrier %.0f ; K loop / first barrier per wave:",
            ctx->variant, c.M, c.N, c.Kchunks * 16, a.n_tiles, npass, nkt, TMW, bn, sum[0][0] / tp, sum[0][0] / tp / nkt, sum[0][1] / tp, sum[0][2] / tp,
            sum[0][3] / tp, sum[0][4] / tp, sum[0][5] / tp, sum[0][6] / tp);
    for (int w = 0; w < 8; ++w) fprintf(stderr, " %.0f/%.0f", sum[w][0] / tp, sum[w][2] / tp);
    fprintf(stderr, "\n");
  }
#endif
  return rc == I8IE_OK ? 1 : rc;
}
