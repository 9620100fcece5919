// i8ie_stem.hip -- the first stage of a network as ONE contraction launch: small-C strided Conv2d (AlexNet conv1:
// 3 -> 96 channels, 11 x 11, stride 4) + relu + max_pool2d, fed by a quantize + space-to-depth pass.
//
//   q   = (u8)(x / scale + zp)                              src/quantize_utils.cc:44-52
//   C   = im2col(q) . W^T + oc                              src/conv2d.cc:100-133 (cblas_gemm_s8u8s32 + oc)
//   out = max_pool2d(relu(down_scale(C)))                   src/quantize_utils.cc:27-36, src/functional.cc:15-64
//
// What round 2 measured on this stage: quantize + repack 0.15 ms, conv1 0.21 ms at 0.19 of the int8 MFMA
// peak (K = 363 padded to 544, 33 % LDS bank-conflict cycles, requantiser VALU work as long as the MFMA work and not
// overlapped with it), max-pool 0.075 ms: 30 % of the AlexNet step for 9 % of its MACs.  This file replaces all three:
//
//   * quantize_s2d_kernel: FP32 NCHW (or u8 NCHW) -> zero-point-padded, re-biased (^0x80) u8 image in a 4 x 4
//     space-to-depth layout [n][Y][X][4 rows][4 px][3 ch] = 48 bytes per s2d pixel.  With stride % 4 == 0 an output
//     pixel's window is ceil(KH / 4) runs of ceil(KW / 4) * 48 contiguous bytes: K = 432 for 11 x 11 x 3 (363 real),
//     against 528 (+16) in the 4-pixel-group layout of i8ie_first.hip, and the image is 25 % smaller (156 vs 207 MB).
//     A lane's MFMA fragment is one ds_read_b128 at base(pixel) + offset(chunk); the 48-byte pixel pitch is 3 x 16 B,
//     and 3 is coprime with 16: the sixteen lanes one ds_read_b128 cycle serves hit sixteen different 16-byte slots.
//   * stem_conv_kernel: one 512-thread block per CU walks whole images, strip by strip (NR conv rows: 2 x 55 pixels =
//     four 32-pixel MFMA tiles).  Waves 0-3 only multiply: wave w owns pixel tile w for ALL output features, its weight
//     slice [N][K] lives in registers for the whole kernel, 14 k-steps x N/32 v_mfma_i32_32x32x32_i8, and the INT32
//     accumulators go to an LDS ring of conv rows.  Waves 4-7 only do vector work, one strip behind: they max-pool the
//     INT32 rows in LDS (down_scale and relu are monotone in C, so pool(relu(down_scale(C))) = relu(down_scale(max C)):
//     the requantiser runs on the pooled quarter of the values only), add oc', requantise, store NHWC bytes, and issue
//     the LDS-DMA of the patch two strips ahead.  One s_barrier per strip.  A SIMD thus holds one MFMA-only and one
//     VALU-only wave, the pairing that co-issues (MI355X guide, "Wave scheduling").
//   * A strip's patch is a contiguous run of the s2d image: whole 1 KiB LDS-DMA pieces, three patch buffers.
// INT32 accumulators are the reference's sums (K order permuted, padded taps carry zero weights); acc_dbg dumps them.
#include <cstdio>
#include <cstdlib>
#include <type_traits>
#include <vector>

#include "i8ie_stem.h"
#include "i8ie_stem_common.h"

// who requests a strip's patch (LDS-DMA): 1 = the vector waves, 0 = the multiplying waves.  Measured both ways on one box
// (tools/dbg/build_stem_ab.sh wt wt-DSTEM_DMA_VEC=1): 0.240 ms from the multiplying waves, 0.245 from the vector waves
#if !defined(STEM_DMA_VEC)
#define STEM_DMA_VEC 0
#endif

#if defined(STEM_PRIO_SWAP)  // (A/B: the multiplying waves raised instead of the vector waves)
#define STEM_VPRIO 0
#define STEM_MPRIO 3
#endif

namespace {

// exact quantize: t = x / scale + zp (fp32 divide, fp32 add), q = ((int)t) & 0xFF (src/quantize_utils.cc:49).
// Fast path: est = fma(x, fl(1/scale), zp).  For |est| < 1024, |t - est| < 2e-4 (one rounding each of 1/scale and of
// the fma against the two roundings of the reference, all on magnitudes < 1151), so when est is further than 2^-10
// from an integer t truncates to the same integer.
__device__ __forceinline__ uint32_t quant_exact(float x, float scale, float zpf, float rscale) {
  const float est = __builtin_fmaf(x, rscale, zpf);
  const float fr = __builtin_amdgcn_fractf(est);
  if (__builtin_fabsf(est) < 1024.0f && __builtin_fabsf(fr - 0.5f) <= 0.5f - 9.765625e-4f)
    return (uint32_t)((int)est) & 0xFFu;
  const float t = x / scale + zpf;
  return (uint32_t)((int)t) & 0xFFu;
}

// ---- quantize + space-to-depth: thread = (image, Y, X, d): 4 pixels x 3 channels of one padded input row ----------
// F32: FP32 input quantised on the way; else u8 input (an already quantised NCHW tensor) copied.
template <bool F32>
__global__ __launch_bounds__(256) void quantize_s2d_kernel(const void* __restrict__ xin, uint8_t* __restrict__ out, uint32_t total,
                                                           int c, int h, int w, int HY, int WX, int pad, float scale, float zpf,
                                                           float rscale, uint32_t zp) {
  const uint32_t gstride = gridDim.x * 256u;
  const size_t cs = (size_t)h * w;
  for (uint32_t e = blockIdx.x * 256u + threadIdx.x; e < total; e += gstride) {
    const uint32_t d = e & 3u, t0 = e >> 2;
    const uint32_t t1 = t0 / (uint32_t)WX, X = t0 - t1 * WX;
    const uint32_t img = t1 / (uint32_t)HY, Y = t1 - img * HY;
    const int y = (int)(4 * Y + d) - pad, x0 = (int)(4 * X) - pad;
    const bool yin = y >= 0 && y < h;
    const int yc = y < 0 ? 0 : (y >= h ? h - 1 : y);
    uint32_t q[12];  // [px][ch]
    if constexpr (F32) {
      const float* plane = static_cast<const float*>(xin) + ((size_t)img * c * h + yc) * w;
      const bool vec = (w & 1) == 0 && (pad & 1) == 0 && w >= 4 && (reinterpret_cast<uintptr_t>(xin) & 7u) == 0;
      const int xb = x0 < 0 ? 0 : (x0 > w - 4 ? w - 4 : x0);
      const int sft = x0 - xb;
#pragma unroll
      for (int ch = 0; ch < 3; ++ch) {
        const float* row = plane + (size_t)(ch < c ? ch : 0) * cs;
        float L[4];
        if (vec) {  // two 8-byte loads from an in-image, 8-byte-aligned base
          const float2 a = *reinterpret_cast<const float2*>(row + xb), b = *reinterpret_cast<const float2*>(row + xb + 2);
          L[0] = a.x; L[1] = a.y; L[2] = b.x; L[3] = b.y;
        } else {
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int xx = xb + i;
            L[i] = row[xx < 0 ? 0 : (xx >= w ? w - 1 : xx)];
          }
        }
#pragma unroll
        for (int px = 0; px < 4; ++px) {
          int i = px + sft;
          i = i < 0 ? 0 : (i > 3 ? 3 : i);
          const float v = i == 0 ? L[0] : (i == 1 ? L[1] : (i == 2 ? L[2] : L[3]));
          q[px * 3 + ch] = quant_exact(v, scale, zpf, rscale);
        }
      }
    } else {
      const uint8_t* plane = static_cast<const uint8_t*>(xin) + ((size_t)img * c * h + yc) * w;
#pragma unroll
      for (int ch = 0; ch < 3; ++ch) {
        const uint8_t* row = plane + (size_t)(ch < c ? ch : 0) * cs;
#pragma unroll
        for (int px = 0; px < 4; ++px) {
          const int xx = x0 + px;
          q[px * 3 + ch] = row[xx < 0 ? 0 : (xx >= w ? w - 1 : xx)];
        }
      }
    }
    uint32_t wd[3] = {0, 0, 0};
#pragma unroll
    for (int px = 0; px < 4; ++px) {
      const int xx = x0 + px;
      const bool in = yin && xx >= 0 && xx < w;
#pragma unroll
      for (int ch = 0; ch < 3; ++ch) {
        const uint32_t v = (in && ch < c) ? q[px * 3 + ch] : zp;  // padding: the input zero point (src/conv2d.cc:24-28)
        const int b = px * 3 + ch;
        wd[b >> 2] |= v << (8 * (b & 3));
      }
    }
    // stored re-biased (u8 ^ 0x80): the contraction kernel's signed MFMA operand as is
    uint32_t* o = reinterpret_cast<uint32_t*>(out) + (size_t)e * 3;
    o[0] = wd[0] ^ 0x80808080u;
    o[1] = wd[1] ^ 0x80808080u;
    o[2] = wd[2] ^ 0x80808080u;
  }
}

// NG: 32-feature groups (N = 32 NG); KS: k-steps; ACC: also dump the INT32 accumulators (acc_dbg of the C-ABI)
template <int NG, int KS, bool ACC>
__global__ __launch_bounds__(512, 2) void stem_conv_kernel(StemArgs p) {
  uint8_t* const smem = stem_smem;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave_hw = __builtin_amdgcn_readfirstlane(tid >> 6);
  // role and index inside the role (0 .. 3).  `wave` below is the role-relative numbering the rest of the kernel was written
  // in: 0 .. 3 = the multiplying waves, 4 .. 7 = the vector waves
  const int wave = p.role_split ? ((wave_hw & 2) << 1) | ((wave_hw >> 2) << 1) | (wave_hw & 1) : wave_hw;
  const int hh = lane >> 5, l31 = lane & 31;

  // units (image, part) of this block: b, b + gridDim.x, ...; their strips are walked as one sequence g = 0 .. G - 1.
  // Conv rows are numbered through that sequence for the ring: row r of a part that starts at row R0 sits in slot
  // (ib + r - R0) % RING, ib = the rows of all earlier units of the block, mod RING.
  const int n_units = p.n_img << p.lg_parts;
  const int n_mine = ((int)blockIdx.x < n_units) ? (n_units - 1 - (int)blockIdx.x) / (int)gridDim.x + 1 : 0;
  if (n_mine == 0) return;
  const int T = p.T, G = n_mine * T, RING = p.RING;
  const int pmask = p.parts - 1;

  for (int i = tid; i < p.N; i += 512) reinterpret_cast<int*>(smem + p.lds_ocp)[i] = p.ocp[i];
  // weights of the LAST feature group, in MFMA fragment order [k-step][lane][16 B]: the multiplying waves read them from
  // here (one fragment per k-step) instead of holding them in registers: 4 x KS registers that the first NG - 1 groups' weights,
  // the accumulators and the fragment rings need (256 per wave at two waves per SIMD)
  for (int i = tid; i < KS * 64; i += 512) {
    const int j = i >> 6, ln = i & 63;
    *reinterpret_cast<v4i*>(smem + p.lds_bfrag + i * 16) =
        *reinterpret_cast<const v4i*>(p.B + (size_t)((NG - 1) * 32 + (ln & 31)) * p.Kpad + j * 32 + (ln >> 5) * 16);
  }
  stem_fill_tables(p, smem, tid, T, RING);
  __syncthreads();  // (every wave, before the roles part: the tables are read from here on)
  // a table entry in two steps: the LDS reads (issued an interval early, in front of the barrier the wave waits at anyway) and
  // the move to SGPRs -- a lookup in one piece exposes an LDS round trip under load to the wave's critical path, once per strip
  auto strip_fetch = [&](int unit, int t, v4i& a, v4i& b) {
    const int idx = (unit & pmask) * T + t;
    a = *reinterpret_cast<const v4i*>(smem + p.lds_tab + idx * 32);
    b = *reinterpret_cast<const v4i*>(smem + p.lds_tab + idx * 32 + 16);
  };
  auto strip_decode = [&](const v4i& a, const v4i& b) {  // (wave-uniform: the entry lands in SGPRs)
    StemStrip e;
    e.lo = __builtin_amdgcn_readfirstlane(a.x); e.hi = __builtin_amdgcn_readfirstlane(a.y);
    e.poff = __builtin_amdgcn_readfirstlane(a.z); e.pbytes = __builtin_amdgcn_readfirstlane(a.w);
    e.j0 = __builtin_amdgcn_readfirstlane(b.x); e.j1 = __builtin_amdgcn_readfirstlane(b.y);
    e.lom = __builtin_amdgcn_readfirstlane(b.z); e.jm = __builtin_amdgcn_readfirstlane(b.w);
    return e;
  };
  auto strip = [&](int unit, int t) {
    v4i a, b;
    strip_fetch(unit, t, a, b);
    return strip_decode(a, b);
  };
  // patch of strip t of image im -> patch buffer pb: one contiguous run of the s2d image, whole 1 KiB LDS-DMA pieces, dealt
  // to the four waves of a role (w4 = wave & 3).  The MULTIPLYING waves issue them (first thing in an interval, for the
  // strip after the one they are about to multiply; awaited before the interval's barrier): they have the slack -- the
  // vector waves are the kernel's critical path, and three pieces cost a wave ~700 cycles of issue (phase stamps)
  auto adv_of = [&](int unit) { return __builtin_amdgcn_readfirstlane(reinterpret_cast<const int*>(smem + p.lds_adv)[unit & pmask]); };
  auto patch_dma = [&](int unit, const StemStrip& e, int pb) {
    if (e.pbytes <= 0) return;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<uint8_t*>(p.img + (size_t)(unit >> p.lg_parts) * p.img_pitch + (unsigned)e.poff), 0, e.pbytes, 0x00020000);
    const int pieces = (e.pbytes + 1023) >> 10;  // the last piece reads zeros past the run (bounds-checked descriptor)
    uint8_t* dst = smem + p.lds_patch + pb * p.patchB;
    for (int q = wave & 3; q < pieces; q += 4)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(dst + q * 1024), 16, (q * 64 + lane) * 16, 0, 0, 0);
  };
#if defined(I8IE_DIAG)
  unsigned long long ph[6] = {0, 0, 0, 0, 0, 0}, tq = 0;
  auto stamp = [&](int i) {
    if (p.dbg) {
      const unsigned long long now = __builtin_amdgcn_s_memtime();
      ph[i] += now - tq;
      tq = now;
    }
  };
  auto stamps_out = [&]() {
    if (p.dbg && lane == 0)
      for (int i = 0; i < 6; ++i) p.dbg[((size_t)blockIdx.x * 8 + wave) * 8 + i] = ph[i];  // (role-relative wave number)
  };
#else
  auto stamp = [](int) {};
  auto stamps_out = []() {};
#endif

  if (wave < 4) {
#if defined(STEM_MPRIO)
    __builtin_amdgcn_s_setprio(STEM_MPRIO);
#endif
    // =========================== multiplying waves ==========================================================
    constexpr int NA = NG - 1;  // groups of pass A (weights in registers); the last group's come from LDS
    v4i breg[NA > 0 ? NA : 1][KS];
#pragma unroll
    for (int g = 0; g < NA; ++g) {
      const int8_t* brow = p.B + (size_t)(g * 32 + l31) * p.Kpad + hh * 16;
#pragma unroll
      for (int j = 0; j < KS; ++j) breg[g][j] = *reinterpret_cast<const v4i*>(brow + j * 32);
    }
    int koff[KS];  // byte offset of this lane's K chunk 2 j + hh inside a pixel's window (chunks past K: 0, zero weights)
    {
      int q = hh, run = 0, rem = hh;  // chunk q = run * RC + rem
#pragma unroll
      for (int j = 0; j < KS; ++j) {
        koff[j] = q < p.nch ? run * p.rowB + rem * 16 : 0;
        q += 2;
        rem += 2;
        while (rem >= p.RC) {
          rem -= p.RC;
          ++run;
        }
      }
    }
    // this lane's pixel of a strip: pp = 32 wave + lane -> (row r inside the strip, column x); fixed for the kernel
    const int pp = wave * 32 + l31;
    int lr = (int)((float)pp * p.rcpOW), lx = pp - lr * p.OW;
    if (lx < 0) { lx += p.OW; --lr; } else if (lx >= p.OW) { lx -= p.OW; ++lr; }
    const int aoff = (lr * p.sq * p.WX + lx * p.sq) * 48;
    const int roff = lx * p.pitchP + 16 * hh;
    int t = 0, un = (int)blockIdx.x, ib = 0, pbuf = 0;  // strip inside the unit; the unit; its ring base; patch buffer g & 1
    int tn = 0, imn = (int)blockIdx.x;  // the strip whose patch is requested next (one ahead of t): strip, unit
    auto next_n = [&]() {
      if (++tn == T) {
        tn = 0;
        imn += (int)gridDim.x;
      }
    };
    StemStrip e = strip(imn, tn);  // the strip multiplied in the coming interval (one table lookup per strip: it is read when
                                   // its patch is requested, an interval ahead)
    patch_dma(imn, e, 0);
    next_n();
    v4i fa, fb;  // the entry of the strip after: fetched in front of every barrier, decoded behind it
    strip_fetch(imn, tn, fa, fb);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    STEM_BAR();  // (the first patch)
#if defined(I8IE_DIAG)
    if (p.dbg) tq = __builtin_amdgcn_s_memtime();
#endif
    for (int g = 0; g < G; ++g) {
      StemStrip en{};
      if (g + 1 < G) {  // the patch of strip g + 1 -> the other buffer (strip g - 1 was its last reader)
        en = strip_decode(fa, fb);
        if (!STEM_DMA_VEC) patch_dma(imn, en, pbuf ^ 1);
        next_n();
      }
      const int npx = (e.hi - e.lo) * p.OW;
      if (wave * 32 < npx) {  // (wave-uniform: a short strip leaves the upper tiles without work)
        const bool valid = pp < npx;
        const uint8_t* abase = smem + p.lds_patch + pbuf * p.patchB + (valid ? aoff : 0);
        int slot = ib + e.lom;
        if (slot >= RING) slot -= RING;
        int ls = slot + lr;
        if (ls >= RING) ls -= RING;
        uint8_t* const rbase = smem + p.lds_ring + ls * p.ringRowB + roff;
        // One pass over K: a k-step is NG MFMAs on one fragment read (the first NG - 1 groups' weights from registers, the last
        // group's fragments from LDS), reads four k-steps ahead, the 4 NG ring stores (lane = pixel, register group q = features
        // 8 q + 4 hh .. + 3) behind the last MFMA.  Until late in round 4 this was two passes (groups 0 .. NG-2, then the last
        // group alone with pass A's ds_write_b128 spread between its MFMAs: with the round-3 vector role, four waves storing 12
        // each at the same moment held every MFMA pipe for ~800 cycles per strip); with the lighter vector role and equal wave
        // priority the one-pass form is the faster one: 0.1723 -> 0.1686 ms inside the AlexNet step (tools/dbg/run_step_ab.sh).
        v16i acc[NG];
        const v16i z = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        uint8_t* const wbase = valid ? rbase : smem + p.lds_dump + lane * 16;  // (lanes past the strip store into a 2 KiB dump)
        auto ring_store = [&](auto ggc, auto qc) {
          constexpr int gg = decltype(ggc)::value, q = decltype(qc)::value;
          v4i v;
          v.x = acc[gg][q * 4 + 0]; v.y = acc[gg][q * 4 + 1]; v.z = acc[gg][q * 4 + 2]; v.w = acc[gg][q * 4 + 3];
          *reinterpret_cast<v4i*>(wbase + (gg * 32 + 8 * q) * 4) = v;
        };
        {  // one pass over K: a k-step is NG MFMAs (the first NG - 1 groups' weights from registers, the last group's from LDS)
          constexpr int DB = KS < 4 ? KS : 4;
          v4i ring[DB], bring[DB];
          const uint8_t* const bbase = smem + p.lds_bfrag + lane * 16;
          static_for<0, DB>([&](auto jc) {
            constexpr int j = decltype(jc)::value;
            ring[j] = *reinterpret_cast<const v4i*>(abase + koff[j]);
            bring[j] = *reinterpret_cast<const v4i*>(bbase + j * 1024);
          });
          __builtin_amdgcn_sched_barrier(0);
          static_for<0, KS>([&](auto jc) {
            constexpr int j = decltype(jc)::value;
            const v4i af = ring[j % DB], bf = bring[j % DB];
            if constexpr (j + DB < KS) {
              ring[j % DB] = *reinterpret_cast<const v4i*>(abase + koff[j + DB]);
              bring[j % DB] = *reinterpret_cast<const v4i*>(bbase + (j + DB) * 1024);
              __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int gg = 0; gg < NA; ++gg) acc[gg] = __builtin_amdgcn_mfma_i32_32x32x32_i8(breg[gg][j], af, j == 0 ? z : acc[gg], 0, 0, 0);
            acc[NG - 1] = __builtin_amdgcn_mfma_i32_32x32x32_i8(bf, af, j == 0 ? z : acc[NG - 1], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
          });
          if constexpr (NA > 0)
            static_for<0, 4 * NA>([&](auto ic) { ring_store(std::integral_constant<int, decltype(ic)::value / 4>{}, std::integral_constant<int, decltype(ic)::value % 4>{}); });
        }
        stamp(0);  // addresses, fragment reads, MFMAs (+ the first NG - 1 groups' ring stores)
        static_for<0, 4>([&](auto qc) { ring_store(std::integral_constant<int, NG - 1>{}, qc); });
#if defined(I8IE_DIAG)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        stamp(1);  // ring writes
#endif
      }
      strip_fetch(imn, tn, fa, fb);
      if (!STEM_DMA_VEC) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the patch requested at the top of the interval has landed
      STEM_BAR();
      stamp(2);  // at the barrier
      if (++t == T) {
        t = 0;
        ib += adv_of(un);
        if (ib >= RING) ib -= RING;
        un += (int)gridDim.x;
      }
      pbuf ^= 1;
      e = en;
    }
    STEM_BAR();  // the vector waves' last interval
    stamps_out();
    return;
  }

  // =============================== vector waves ==============================================================
  const int vt = (wave - 4) * 64 + lane;  // 0 .. 255
  const int vwave = wave - 4;
  // Wave priority: in round 3 the vector waves (330 instructions a round, the critical path) ran 7 % faster raised to priority 3.
  // With the round-4 pool pass (210 instructions) the multiplying waves are the critical path and a raised vector role costs:
  // first stage inside the AlexNet step 0.1833 ms at priority 3, 0.183 at 1, 0.1727 at 0 (tools/dbg/run_step_ab.sh) -- equal priority.
#if !defined(STEM_VPRIO)
#define STEM_VPRIO 0
#endif
  if (STEM_VPRIO) __builtin_amdgcn_s_setprio(STEM_VPRIO);
#if defined(I8IE_DIAG)
  if (p.dbg_flags & 1) __builtin_amdgcn_s_setprio(3);  // experiment: the vector waves raised
#endif
  const I8ieRequant rq = p.rq;
  const int lo_relu = p.relu_lo;
  const float lof = (float)lo_relu;
  const int N4 = p.N >> 2;
  const int PHp = p.PH + 2 * p.ob, PWp = p.PW + 2 * p.ob;
  const __amdgpu_buffer_rsrc_t rsO =
      __builtin_amdgcn_make_buffer_rsrc(p.out, 0, p.out_bytes, 0x00020000);  // (< 2^31: checked on the host)
  // Lane map of the pool pass.  One ds_read_b128 cycle serves 16 lanes: {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} and
  // the same + 32.  Those 16 lanes take 8 pooled pixels x 2 consecutive feature quads: a pooled pixel is ps pixels of
  // (4 N + 16) bytes further on, which is 2 (mod 16) slots of 16 bytes for every N this kernel takes at ps = 2, so the
  // sixteen addresses fall into sixteen different slots of 256 bytes.  A wave: 8 pixels x 8 quads; a pass over a pooled
  // row is ceil(PW / 8) x (N / 32) such tasks, dealt round-robin to the four waves.
  const int hg = ((l31 >= 4 && l31 < 12) || (l31 >= 16 && l31 < 20) || l31 >= 28) ? 1 : 0;  // lane group inside a half
  const int gi = hg == 0 ? (l31 < 4 ? l31 : (l31 < 16 ? l31 - 8 : l31 - 12)) : (l31 < 12 ? l31 - 4 : (l31 < 20 ? l31 - 8 : l31 - 16));
  const int pl = gi & 7;                               // pixel inside the task's 8
  const int ql = 2 * (2 * hh + hg) + (gi >> 3);        // feature quad inside the task's 8
  const int nqo = N4 >> 3, npo = (p.PW + 7) >> 3, ntask = nqo * npo;
  // bytes of a ring pixel, known here (the host's pitchP: 4 N + 16 with N = 32 NG): the window's dx steps become the
  // immediate offsets of the ds_read_b128 instead of a vector add each -- the vector waves pay ~9 cycles per instruction
  // of any kind beside an MFMA stream (phase stamps), and a third of a pool round's instructions were address arithmetic
  constexpr int PITCH = NG * 128 + 16;
  const int rd_lane = pl * p.ps * PITCH + ql * 16;  // LDS byte offset of this lane's window origin inside a task
  const int st_lane = pl * p.N + ql * 4;               // output byte offset of this lane inside a task

  // A wave's pool tasks: k = vwave + 4 u (+ 12 per further round); task k = (pixel octet po, quad octet qo).  What does
  // not depend on the pooled row is worked out once: LDS offset of the lane's window origin, output offset (lanes past
  // the row's last pixel: read pixel 0 of the task, store beyond the descriptor's range), LDS address of its oc' quad
  constexpr int UN = 3;  // (2: 0.216 ms, 1: 0.193 ms against 0.176 inside the AlexNet step)
  struct PoolTask {
    int rd, st, oc;
  };
  auto task_of = [&](int k) {
    int po = k < ntask ? k : 0, qo = 0;  // k = qo * npo + po
    while (po >= npo) {
      po -= npo;
      ++qo;
    }
    const bool ok = 8 * po + pl < p.PW;
    PoolTask t;
    t.rd = (ok ? rd_lane : ql * 16) + (8 * po) * p.ps * PITCH + qo * 128;
    t.st = ok && k < ntask ? (8 * po) * p.N + qo * 32 + st_lane : (int)0x80000000;  // (a task past the row's last: stored nowhere)
    t.oc = p.lds_ocp + (qo * 8 + ql) * 16;
    return t;
  };
  PoolTask tk0[UN];
#pragma unroll
  for (int u = 0; u < UN; ++u) tk0[u] = task_of(vwave + 4 * u);
  v4i oc0[UN];  // oc' of the first round's tasks: the same for every pooled row (the table is complete: barrier above)
#pragma unroll
  for (int u = 0; u < UN; ++u) oc0[u] = *reinterpret_cast<const v4i*>(smem + tk0[u].oc);
  // pooled rows [j0, j1) of image im (ring base ib): max over the INT32 window, + oc', requantise, store.  The (up to
  // three) tasks of a round run side by side without a branch between them: all window reads first, then three
  // independent chains of maxima and requantiser -- one wave per SIMD does this work, nothing else covers its latencies
  int nst = 0;  // output stores this wave issued since its last patch request (they retire in order behind it)
  auto pool_rows = [&](auto pkc, int im, int ib, const StemStrip& e) {
    constexpr int PK = decltype(pkc)::value;
    int rm = ib + e.jm;  // ring slot of the row's first conv row: (ib + (j * ps) % RING) % RING
    if (rm >= RING) rm -= RING;
    for (int j = e.j0; j < e.j1; ++j) {
      int rowoff[PK];
#pragma unroll
      for (int dy = 0; dy < PK; ++dy) {
        const int sl = rm + dy;
        rowoff[dy] = p.lds_ring + (sl >= RING ? sl - RING : sl) * p.ringRowB;
      }
      const int orow = ((im * PHp + j + p.ob) * PWp + p.ob) * p.N;
      // one round: UN tasks side by side (tasks past the row's last are computed and stored nowhere)
      auto round = [&](const PoolTask (&tk)[UN], const v4i (&ocv)[UN]) {
        v4i v[UN][PK * PK];
#pragma unroll
        for (int u = 0; u < UN; ++u)
#pragma unroll
          for (int dy = 0; dy < PK; ++dy) {
            const uint8_t* const a = smem + (rowoff[dy] + tk[u].rd);  // one vector add per window row; dx: immediate offsets
#pragma unroll
            for (int dx = 0; dx < PK; ++dx) v[u][dy * PK + dx] = *reinterpret_cast<const v4i*>(a + dx * PITCH);
          }
#if defined(I8IE_DIAG)
        stamp(1);  // (pool pass: scalar part, addresses, read issue)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        stamp(4);  // (pool pass: window reads landing)
#endif
        uint32_t d[UN];
        float worst[UN];
        int c4[UN][4];
#pragma unroll
        for (int u = 0; u < UN; ++u) {
          v4i m = v[u][0];
#pragma unroll
          for (int tt = 1; tt < PK * PK; ++tt) {
            m.x = m.x > v[u][tt].x ? m.x : v[u][tt].x;
            m.y = m.y > v[u][tt].y ? m.y : v[u][tt].y;
            m.z = m.z > v[u][tt].z ? m.z : v[u][tt].z;
            m.w = m.w > v[u][tt].w ? m.w : v[u][tt].w;
          }
          c4[u][0] = m.x + ocv[u].x; c4[u][1] = m.y + ocv[u].y;  // max(C) + oc' = max(C + oc'): exact integer adds
          c4[u][2] = m.z + ocv[u].z; c4[u][3] = m.w + ocv[u].w;
          d[u] = i8ie_requant_est4(c4[u], rq, lof, worst[u]);
        }
        float wmin = worst[0];
#pragma unroll
        for (int u = 1; u < UN; ++u) wmin = __builtin_fminf(wmin, worst[u]);
        if (!i8ie_requant_est_ok(wmin)) {  // a value within 2^-13 of a rounding boundary: the exact sequence for its pack
#pragma unroll
          for (int u = 0; u < UN; ++u)
            if (!i8ie_requant_est_ok(worst[u])) d[u] = i8ie_requant_exact4(c4[u], rq, lo_relu);
        }
#pragma unroll
        for (int u = 0; u < UN; ++u) __builtin_amdgcn_raw_buffer_store_b32(d[u] ^ p.xor_out, rsO, tk[u].st, orow, 0);
        nst += UN;
        stamp(5);  // (pool pass: maxima, requantiser, stores)
      };
      round(tk0, oc0);
      for (int k0 = vwave + 4 * UN; k0 < ntask; k0 += 4 * UN) {  // (more than 12 tasks to a pooled row: none of AlexNet's)
        PoolTask tk[UN];
        v4i ocv[UN];
#pragma unroll
        for (int u = 0; u < UN; ++u) {
          tk[u] = task_of(k0 + 4 * u);
          ocv[u] = *reinterpret_cast<const v4i*>(smem + tk[u].oc);
        }
        round(tk, ocv);
      }
      rm += p.ps;
      if (rm >= RING) rm -= RING;
    }
  };
  auto dump_acc = [&](int im, int ib, const StemStrip& e) {  // conv rows [lo, hi) of the image: ring -> acc_dbg
    const int tasks = (e.hi - e.lo) * p.OW * N4;
    const float rn4 = 1.0f / (float)N4;
    for (int id = vt; id < tasks; id += 256) {
      int t1 = (int)((float)id * rn4), qq = id - t1 * N4;
      if (qq < 0) { qq += N4; --t1; } else if (qq >= N4) { qq -= N4; ++t1; }
      int r = 0, x = t1;
      while (x >= p.OW) {
        x -= p.OW;
        ++r;
      }
      int sl = ib + e.lom + r;
      while (sl >= RING) sl -= RING;
      const v4i v = *reinterpret_cast<const v4i*>(smem + p.lds_ring + sl * p.ringRowB + x * p.pitchP + qq * 16);
      const v4i o = *reinterpret_cast<const v4i*>(smem + p.lds_ocp + qq * 16);
      const int4 c = make_int4(v.x + o.x, v.y + o.y, v.z + o.z, v.w + o.w);
      *reinterpret_cast<int4*>(p.acc + ((size_t)im * (size_t)(p.OH * p.OW) + (size_t)((e.lo + r) * p.OW + x)) * (size_t)p.N + qq * 4) = c;
    }
  };
  // wait until at most `keep` of this wave's vector-memory operations are outstanding (they retire in issue order)
  auto wait_vm_keep = [&](int keep) {
    switch (keep < 16 ? keep : 16) {
      case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
      case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
      case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
      case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
      case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
      case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
      case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
      case 7: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
      case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
      case 9: asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); break;
      case 10: asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); break;
      case 11: asm volatile("s_waitcnt vmcnt(11)" ::: "memory"); break;
      case 12: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
      case 13: asm volatile("s_waitcnt vmcnt(13)" ::: "memory"); break;
      case 14: asm volatile("s_waitcnt vmcnt(14)" ::: "memory"); break;
      case 15: asm volatile("s_waitcnt vmcnt(15)" ::: "memory"); break;
      default: asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); break;
    }
  };

  // running state: strip g - 1 (the rows to pool): strip, unit, ring base
  int t1 = 0, im1 = (int)blockIdx.x, ib1 = 0;
  int tn = 1, imn = (int)blockIdx.x;  // the strip whose patch is requested next (strip 0's came with the multiplying waves' prologue)
  if (tn == T) {
    tn = 0;
    imn += (int)gridDim.x;
  }
  v4i fa, fb;  // the entry of the strip pooled in the next interval: fetched in front of every barrier, decoded behind it
  strip_fetch(im1, t1, fa, fb);
  STEM_BAR();
#if defined(I8IE_DIAG)
  if (p.dbg) tq = __builtin_amdgcn_s_memtime();
#endif
  for (int g = 0; g <= G; ++g) {
    // interval g: the multiplying waves work on strip g (none at g == G); here: the pooled rows completed by strip g - 1.
    // The output stores stay in flight across the barriers (STEM_BAR waits for LDS operations only).
    if (STEM_DMA_VEC && g + 1 < G) {  // the patch of strip g + 1 -> the buffer strip g - 1 was read from
      patch_dma(imn, strip(imn, tn), (g + 1) & 1);
      if (++tn == T) {
        tn = 0;
        imn += (int)gridDim.x;
      }
      nst = 0;
    }
    stamp(0);  // patch request
    if (g >= 1) {
      const StemStrip e = strip_decode(fa, fb);
      const int img = im1 >> p.lg_parts;
      if (e.j1 > e.j0) {
        if (p.pk == 3) pool_rows(std::integral_constant<int, 3>{}, img, ib1, e);
        else if (p.pk == 2) pool_rows(std::integral_constant<int, 2>{}, img, ib1, e);
        else pool_rows(std::integral_constant<int, 1>{}, img, ib1, e);
      }
      if constexpr (ACC) {
        if (e.hi > e.lo) dump_acc(img, ib1, e);
      }
      if (++t1 == T) {
        t1 = 0;
        ib1 += adv_of(im1);
        if (ib1 >= RING) ib1 -= RING;
        im1 += (int)gridDim.x;
      }
    }
    strip_fetch(im1, t1, fa, fb);
    stamp(1);  // pool pass
    if constexpr (ACC) wait_vm_keep(0);
    else if (STEM_DMA_VEC) wait_vm_keep(nst);  // the patch has landed; the stores issued behind it may stay in flight
    stamp(2);
    STEM_BAR();
    stamp(3);  // at the barrier
  }
  stamps_out();
}

template <int NG, int KS, bool ACC>
int launch_stem_t(i8ie_ctx* ctx, const StemArgs& a, int grid, int lds) {
  static bool raised[64] = {};
  const int dev = ctx->device & 63;
  if (!raised[dev]) {
    I8IE_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&stem_conv_kernel<NG, KS, ACC>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    raised[dev] = true;
  }
  stem_conv_kernel<NG, KS, ACC><<<grid, 512, lds, ctx->stream>>>(a);
  I8IE_LAUNCH_CHECK();
  return I8IE_OK;
}
template <int NG, int KS>
int launch_stem(i8ie_ctx* ctx, const StemArgs& a, int grid, int lds) {
  return a.acc != nullptr ? launch_stem_t<NG, KS, true>(ctx, a, grid, lds) : launch_stem_t<NG, KS, false>(ctx, a, grid, lds);
}

struct StemPlan {
  int KC4, KA4, HY, WX, K2, KS, KST, NR, first, T, parts, pk, ps, PH, PW, carry, RING, pitchP, ringRowB, patchB, lds_tab, lds_adv, lds_dump, lds_bfrag, lds;
};

// geometry this kernel takes: <= 3 channels, stride a multiple of 4, N in {32, 64, 96}, K within 14 k-steps, a strip of
// whole conv rows within 128 pixels, the INT32 ring and three patches within the CU's LDS
bool stem_plan(int c, int stride, int N, int KH, int KW, int OH, int OW, int pool_k, int pool_s, int parts, StemPlan* out) {
  if (c > 3 || c < 1 || stride < 4 || stride % 4 != 0) return false;
  if (N % 32 != 0 || N < 32 || N > 96) return false;
  if (pool_k == 1 && pool_s > 1) return false;  // (a subsampling 1 x 1 pool: the caller runs the max-pool kernel behind the conv)
  StemPlan s{};
  const int sq = stride / 4;
  s.KC4 = (KH + 3) / 4;
  s.KA4 = (KW + 3) / 4;
  s.HY = (OH - 1) * sq + s.KC4;
  s.WX = (OW - 1) * sq + s.KA4;
  s.K2 = s.KC4 * s.KA4 * 48;
  s.KS = (s.K2 + 31) / 32;
  if (s.KS > kStemMaxKS || s.KS < 2) return false;
  s.pk = pool_k > 1 ? pool_k : 1;
  s.ps = pool_k > 1 ? pool_s : 1;
  if (s.ps < 1 || s.ps > s.pk || s.pk > 3) return false;
  if (s.pk > OH || s.pk > OW) return false;
  s.PH = (OH - s.pk) / s.ps + 1;
  s.PW = (OW - s.pk) / s.ps + 1;
  s.carry = s.pk - s.ps;
  int NR = kStemPix / OW;
  NR -= NR % s.ps;
  if (NR < 1 || NR < s.carry) return false;
  if (NR > 8) NR = 8 - 8 % s.ps;
  s.NR = NR;
  s.first = s.carry > 0 ? s.carry : NR;
  if (s.first > OH) s.first = OH;
  if (parts != 1 && parts != 2 && parts != 4) return false;
  if (s.PH < 2 * parts) return false;  // (at least two pooled rows per part)
  s.parts = parts;
  // part q = pooled rows [J0, J1) = conv rows [R0, R1) (the kernel's table, restated)
  auto J0_of = [&](int q) { return q * s.PH / parts; };
  auto R0_of = [&](int q) { return J0_of(q) * s.ps; };
  auto R1_of = [&](int q) { return q + 1 == parts ? OH : (J0_of(q + 1) - 1) * s.ps + s.pk; };
  s.T = 1;
  for (int q = 0; q < parts; ++q) {
    const int rows = R1_of(q) - R0_of(q);
    const int t = 1 + (rows > s.first ? (rows - s.first + NR - 1) / NR : 0);
    if (t > s.T) s.T = t;
  }
  s.RING = 2 * NR + s.carry;
  s.pitchP = N * 4 + 16;
  s.ringRowB = OW * s.pitchP;
  const int prmax = (NR - 1) * sq + s.KC4;
  s.patchB = ((prmax * s.WX * 48 + 1023) & ~1023) + 1024;  // (+ one piece: the pad chunk of the last pixel may read past the run)
  if (s.T * parts > 256) return false;  // (the strip table is filled by the first 256 threads)
  s.KST = s.KS <= 6 ? 6 : (s.KS <= 10 ? 10 : kStemMaxKS);  // k-steps of the kernel instantiation
  s.lds_tab = 2 * s.patchB + s.RING * s.ringRowB + N * 4 + 64;
  s.lds_adv = s.lds_tab + s.T * parts * 32;
  s.lds_dump = s.lds_adv + 16;
  s.lds_bfrag = s.lds_dump + 2048;
  s.lds = s.lds_bfrag + s.KST * 1024;
  if (s.lds > 160 * 1024) return false;
  // the ring holds every conv row in flight: walk the intervals of two consecutive units (any part after any part; rows
  // numbered through: unit 2's row r sits rows(unit 1) + r - R0 further on)
  for (int qa = 0; qa < parts; ++qa)
    for (int qb = 0; qb < parts; ++qb) {
      const int q_of[2] = {qa, qb};
      const int base[2] = {0, R1_of(qa) - R0_of(qa)};
      auto hi_of = [&](int u, int t) { const int v = R0_of(q_of[u]) + s.first + t * s.NR; return v < R1_of(q_of[u]) ? v : R1_of(q_of[u]); };
      auto lo_of = [&](int u, int t) { return t == 0 ? R0_of(q_of[u]) : hi_of(u, t - 1); };
      auto done = [&](int u, int hi) {
        const int J0 = J0_of(q_of[u]), J1 = q_of[u] + 1 == parts ? s.PH : J0_of(q_of[u] + 1);
        if (hi < s.pk) return J0;
        int e = (hi - s.pk) / s.ps + 1;
        e = e < J1 ? e : J1;
        return e > J0 ? e : J0;
      };
      for (int g = 1; g < 2 * s.T; ++g) {  // interval g: strip g written, pooled rows completed by strip g - 1 read
        const int u1 = (g - 1) / s.T, t1 = (g - 1) % s.T;
        const int j0 = t1 == 0 ? J0_of(q_of[u1]) : done(u1, lo_of(u1, t1)), j1 = done(u1, hi_of(u1, t1));
        if (j1 <= j0) continue;
        const int lo_read = base[u1] + j0 * s.ps - R0_of(q_of[u1]);
        const int u2 = g / s.T, t2 = g % s.T;
        const int wlo = base[u2] + lo_of(u2, t2) - R0_of(q_of[u2]), whi = base[u2] + hi_of(u2, t2) - 1 - R0_of(q_of[u2]);
        if (whi >= wlo && whi - lo_read + 1 > s.RING) return false;
      }
    }
  *out = s;
  return true;
}

}  // namespace

int i8ie_stem_supported(int c, int stride, int N, int KH, int KW, int OH, int OW, int pool_k, int pool_s) {
  StemPlan s;
  return stem_plan(c, stride, N, KH, KW, OH, OW, pool_k, pool_s, 1, &s) ? 1 : 0;
}

// the weight panel's K length (bytes per output feature, a multiple of 32) and the K index of (ch, kh, kw)
int i8ie_stem_kpad(int KH, int KW) {  // whole k-steps of the kernel instantiation that takes this K (6, 10 or 14)
  const int KC4 = (KH + 3) / 4, KA4 = (KW + 3) / 4;
  const int ks = (KC4 * KA4 * 48 + 31) / 32;
  return (ks <= 6 ? 6 : ks <= 10 ? 10 : kStemMaxKS) * 32;
}
int i8ie_stem_kindex(int KW, int ch, int kh, int kw) {
  const int KA4 = (KW + 3) / 4;
  return (kh / 4) * (KA4 * 48) + (kw / 4) * 48 + (kh % 4) * 12 + (kw % 4) * 3 + ch;
}

size_t i8ie_stem_scratch_bytes(int n, int KH, int KW, int stride, int OH, int OW) {
  const int sq = stride / 4;
  const size_t HY = (size_t)(OH - 1) * sq + (KH + 3) / 4, WX = (size_t)(OW - 1) * sq + (KW + 3) / 4;
  return (size_t)n * HY * WX * 48;
}

int i8ie_stem_launch(i8ie_ctx* ctx, const I8ieStemCall& c) {
  static int cus[64] = {};
  const int dev = ctx->device & 63;
  if (cus[dev] == 0) {
    hipDeviceProp_t prop;
    I8IE_HIP_TRY(hipGetDeviceProperties(&prop, ctx->device));
    cus[dev] = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  }
  // Fewer images than CUs (the shard of a multi-GPU batch): an image is cut into 2 or 4 parts of whole pooled rows, each a
  // unit of its own; a part recomputes the pk - ps conv rows it shares with its neighbour (1 of 28 on AlexNet's conv1)
  StemPlan s;
  int parts = 1;
  const int ncu = i8ie_cus(ctx, cus[dev]);
  while (parts < 4 && (long)c.n * parts * 4 < (long)ncu * 3) parts *= 2;
  if (ctx->variant == 12) parts = 1;  // (I8IE_VARIANT_STEM_WHOLE)
  while (parts > 1 && !stem_plan(c.c, c.stride, c.N, c.KH, c.KW, c.OH, c.OW, c.pool_k, c.pool_s, parts, &s)) parts /= 2;
  if (!stem_plan(c.c, c.stride, c.N, c.KH, c.KW, c.OH, c.OW, c.pool_k, c.pool_s, parts, &s)) {
    i8ie_set_error("i8ie_stem_launch: geometry not supported");
    return I8IE_ERR_STATE;
  }
  const size_t img_pitch = (size_t)s.HY * s.WX * 48;
  const size_t out_bytes = (size_t)c.n * (s.PH + 2 * c.ob) * (s.PW + 2 * c.ob) * c.N;
  I8IE_REQUIRE(img_pitch < ((size_t)1 << 31) && out_bytes < ((size_t)1 << 31), "i8ie_stem_launch: tensor beyond the 32-bit offset range");
  I8IE_REQUIRE((size_t)c.n * 4 * s.HY * s.WX < ((size_t)1 << 32) - 65536, "i8ie_stem_launch: batch too large for one launch");
  // ---- pass 1: FP32 (or u8) NCHW -> quantised, zero-point padded, re-biased s2d image
  {
    const uint32_t total = (uint32_t)((size_t)c.n * s.HY * s.WX * 4);
    I8ieProfScope prof(ctx, c.x != nullptr ? "quantize_s2d_f32" : "repack_s2d_u8", 0.0,
                       (c.x != nullptr ? 4.0 : 1.0) * c.n * c.c * c.h * c.w + (double)c.n * img_pitch);
    uint32_t blocks = (total + 255) / 256;
    // (grid-stride beyond that; inside the AlexNet step at 1000 images: 2048 blocks 0.1376 ms, 8192: 0.1365, 32768: 0.1338, one block per
    //  256 threads of work (147 k blocks): 0.1403)
    if (blocks > 256 * 128) blocks = 256 * 128;
    const float rs = 1.0f / c.q_scale;
    if (c.x != nullptr)
      quantize_s2d_kernel<true><<<blocks, 256, 0, ctx->stream>>>(c.x, c.scratch, total, c.c, c.h, c.w, s.HY, s.WX, c.pad, c.q_scale,
                                                                (float)c.q_zp, rs, (uint32_t)(c.q_zp & 0xFF));
    else
      quantize_s2d_kernel<false><<<blocks, 256, 0, ctx->stream>>>(c.xu8, c.scratch, total, c.c, c.h, c.w, s.HY, s.WX, c.pad, 1.0f, 0.0f,
                                                                 1.0f, (uint32_t)(c.q_zp & 0xFF));
    I8IE_LAUNCH_CHECK();
  }
  // ---- pass 2: the contraction + pool
  StemArgs a{};
  a.img = c.scratch;
  a.img_pitch = (unsigned)img_pitch;
  a.n_img = c.n;
  a.WX = s.WX; a.rowB = s.WX * 48;
  a.OH = c.OH; a.OW = c.OW; a.sq = c.stride / 4; a.KC4 = s.KC4;
  a.NR = s.NR; a.first = s.first; a.T = s.T;
  a.pk = s.pk; a.ps = s.ps; a.PH = s.PH; a.PW = s.PW;
  a.B = c.B; a.Kpad = c.Kpad; a.N = c.N; a.ocp = c.ocp;
  a.rq = i8ie_make_requant(c.s_in, c.s_w, c.s_out, c.zp_out);
  a.relu_lo = c.relu ? c.zp_out : 0;
  a.out = c.out; a.ob = c.ob; a.xor_out = c.out_s8 ? 0x80808080u : 0u;
  a.acc = c.acc;
  a.pitchP = s.pitchP; a.ringRowB = s.ringRowB; a.RING = s.RING;
  a.patchB = s.patchB;
  a.lds_patch = 0;
  a.lds_ring = 2 * s.patchB;
  a.lds_ocp = a.lds_ring + s.RING * s.ringRowB;
  a.lds_tab = s.lds_tab;
  a.lds_dump = s.lds_dump;
  a.lds_bfrag = s.lds_bfrag;
  a.out_bytes = (unsigned)out_bytes;
  a.rcpOW = 1.0f / (float)c.OW;
  a.RC = s.KA4 * 3;
  a.nch = s.KC4 * a.RC;
  a.parts = s.parts;
  a.lg_parts = s.parts == 4 ? 2 : (s.parts == 2 ? 1 : 0);
  a.lds_adv = s.lds_adv;
  int grid = ncu;
  if (grid > c.n * s.parts) grid = c.n * s.parts;
  const double ops = 2.0 * c.n * c.OH * c.OW * (double)c.N * c.c * c.KH * c.KW;
  const double bytes = (double)c.n * img_pitch + (double)out_bytes;
  a.role_split = ctx->variant == 13 ? 1 : 0;  // (I8IE_VARIANT_STEM_SIMD_ROLES: A/B of the role placement)
#if defined(I8IE_DIAG)
  const bool fused = s.pk == 3 && ctx->variant == 16;  // (tools/diag/csrc/i8ie_stem_fused.hip: every wave in both roles)
#endif
  I8ieProfScope prof(ctx, s.pk > 1 ? "stem_conv_pool" : "stem_conv", ops, bytes);
#if defined(I8IE_DIAG)
  static unsigned long long* dbg_dev[64] = {};
  const bool stamps = std::getenv("I8IE_STEM_STAMPS") != nullptr;
  if (stamps) {
    if (!dbg_dev[dev]) I8IE_HIP_TRY(hipMalloc(&dbg_dev[dev], 4096 * 64 * sizeof(unsigned long long)));
    I8IE_HIP_TRY(hipMemsetAsync(dbg_dev[dev], 0, 4096 * 64 * sizeof(unsigned long long), ctx->stream));
    a.dbg = dbg_dev[dev];
  }
  if (const char* e = std::getenv("I8IE_STEM_FLAGS")) a.dbg_flags = std::atoi(e);
  struct Report {
    i8ie_ctx* ctx; unsigned long long* d; int grid, strips; bool on, fused;
    ~Report() {
      if (!on) return;
      std::vector<unsigned long long> h((size_t)grid * 64);
      if (hipStreamSynchronize(ctx->stream) != hipSuccess) return;
      if (hipMemcpy(h.data(), d, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost) != hipSuccess) return;
      double sm[8][6] = {};
      for (int b = 0; b < grid; ++b)
        for (int w = 0; w < 8; ++w)
          for (int i = 0; i < 6; ++i) sm[w][i] += (double)h[((size_t)b * 8 + w) * 8 + i];
      const double n = (double)strips;  // strips in all (sum over blocks)
      if (fused) {
        fprintf(stderr, "stem_stamps (every wave in both roles): %d strips; per strip, cycles [region | ring stores + rest of the pool pass | patch wait + barrier]:", strips);
        for (int w = 0; w < 8; ++w) fprintf(stderr, " w%d %.0f|%.0f|%.0f", w, sm[w][0] / n, sm[w][1] / n, sm[w][2] / n);
        fprintf(stderr, "\n");
        return;
      }
      fprintf(stderr, "stem_stamps: %d strips; per strip, cycles: multiplying waves [mfma | ring write | barrier]:", strips);
      for (int w = 0; w < 4; ++w) fprintf(stderr, " w%d %.0f|%.0f|%.0f", w, sm[w][0] / n, sm[w][1] / n, sm[w][2] / n);
      fprintf(stderr, " ; vector waves [patch request | pool: issue, reads landing, compute | patch wait | barrier]:");
      for (int w = 4; w < 8; ++w) fprintf(stderr, " w%d %.0f|%.0f,%.0f,%.0f|%.0f|%.0f", w, sm[w][0] / n, sm[w][1] / n, sm[w][4] / n, sm[w][5] / n, sm[w][2] / n, sm[w][3] / n);
      fprintf(stderr, "\n");
    }
  } report{ctx, a.dbg, grid, c.n * s.T * s.parts, a.dbg != nullptr, fused};
#endif
  const int NG = c.N / 32;
#if defined(I8IE_DIAG)
  if (fused) return i8ie_stem_fused_launch(ctx, a, NG, s.KS, grid, s.lds);
#endif
#define I8IE_STEM_KS(NGV)                                                   \
  if (s.KS <= 6) return launch_stem<NGV, 6>(ctx, a, grid, s.lds);           \
  if (s.KS <= 10) return launch_stem<NGV, 10>(ctx, a, grid, s.lds);         \
  return launch_stem<NGV, kStemMaxKS>(ctx, a, grid, s.lds);
  if (NG == 1) { I8IE_STEM_KS(1) }
  if (NG == 2) { I8IE_STEM_KS(2) }
  I8IE_STEM_KS(3)
#undef I8IE_STEM_KS
}
