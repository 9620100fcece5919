// i8ie_stem_fused.hip -- round-4 experiment on the first-stage kernel (diagnostic build only, variant 16): every wave of the
// block in both roles.  Bit-exact, and SLOWER than the role-specialised product kernel (0.218 vs 0.179 ms on the same
// box, profiles/r04_stem_fused_roles.txt): in this kernel the time of a strip is the SIMD's MFMA cycles plus ~4.5 cycles
// for every other instruction either of its waves issues -- nothing overlaps, whichever wave the instruction sits in.
#include <cstdio>
#include <cstdlib>

#include "i8ie_stem_common.h"

namespace {

// ---- the same stage with every wave in both roles (round 4) --------------------------------------------------------
// What the role-specialised kernel above measures (phase stamps): its multiplying waves take 3.3 k cycles over 1.34 k
// cycles of MFMAs and its vector waves 2.6 k over ~210 instructions -- a wave issues one instruction of any kind per
// ~9 cycles beside the other wave's MFMA stream, and the only vector work an MFMA hides is the issuing wave's own (three
// instructions per 32 x 32 x 32 MFMA; tools/valu_probe.hip).  Here all eight waves multiply AND pool: wave (w, h) owns
// pixel tile w of the strip for the feature groups of its half (h = 0: the first (NG + 1) / 2 groups, h = 1: the rest;
// all weights in registers), and in the same instruction stream -- one scheduling region, sched_group_barrier placing
// vector instructions behind each MFMA -- it pools and requantises its share of the pooled row the strip before
// completed (h = 1 waves: two tasks of 8 pixels x 32 features, h = 0: one; 12 tasks to an AlexNet row).  3 x 3 pools.
template <int NG, int KS, bool ACC>
__global__ __launch_bounds__(512, 2) void stem_fused_kernel(StemArgs p) {
  uint8_t* const smem = stem_smem;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int w4 = wave & 3, h = wave >> 2;  // (waves w and w + 4 share a SIMD)
  const int hh = lane >> 5, l31 = lane & 31;
  constexpr int PK = 3;
  constexpr int PITCH = NG * 128 + 16;              // bytes of a ring pixel (the host's pitchP)
  constexpr int NA = (NG + 1) / 2, NB = NG - NA;    // feature groups of an h = 0 / h = 1 wave

  const int n_units = p.n_img << p.lg_parts;
  const int n_mine = ((int)blockIdx.x < n_units) ? (n_units - 1 - (int)blockIdx.x) / (int)gridDim.x + 1 : 0;
  if (n_mine == 0) return;
  const int T = p.T, G = n_mine * T, RING = p.RING;
  const int pmask = p.parts - 1;

  for (int i = tid; i < p.N; i += 512) reinterpret_cast<int*>(smem + p.lds_ocp)[i] = p.ocp[i];
  stem_fill_tables(p, smem, tid, T, RING);
  __syncthreads();
  auto strip = [&](int unit, int t) {  // (wave-uniform: the entry lands in SGPRs)
    const int idx = (unit & pmask) * T + t;
    const v4i a = *reinterpret_cast<const v4i*>(smem + p.lds_tab + idx * 32), b = *reinterpret_cast<const v4i*>(smem + p.lds_tab + idx * 32 + 16);
    StemStrip e;
    e.lo = __builtin_amdgcn_readfirstlane(a.x); e.hi = __builtin_amdgcn_readfirstlane(a.y);
    e.poff = __builtin_amdgcn_readfirstlane(a.z); e.pbytes = __builtin_amdgcn_readfirstlane(a.w);
    e.j0 = __builtin_amdgcn_readfirstlane(b.x); e.j1 = __builtin_amdgcn_readfirstlane(b.y);
    e.lom = __builtin_amdgcn_readfirstlane(b.z); e.jm = __builtin_amdgcn_readfirstlane(b.w);
    return e;
  };
  auto adv_of = [&](int unit) { return __builtin_amdgcn_readfirstlane(reinterpret_cast<const int*>(smem + p.lds_adv)[unit & pmask]); };
  // patch of strip t of a unit -> patch buffer pb, 1 KiB LDS-DMA pieces dealt to the four h = 0 waves (as in the kernel above)
  auto patch_dma = [&](int unit, int t, int pb) {
    const StemStrip e = strip(unit, t);
    if (e.pbytes <= 0) return;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<uint8_t*>(p.img + (size_t)(unit >> p.lg_parts) * p.img_pitch + (unsigned)e.poff), 0, e.pbytes, 0x00020000);
    const int pieces = (e.pbytes + 1023) >> 10;
    uint8_t* dst = smem + p.lds_patch + pb * p.patchB;
    for (int q = w4; q < pieces; q += 4)
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
#else
  auto stamp = [](int) {};
#endif

  // The rest is written once and instantiated per role (h = 0 / h = 1: different numbers of feature groups and pool tasks;
  // a role's weights must not occupy registers in the other role's code)
  auto run = [&](auto hc) {
    // ---- multiplying role: weights of this wave's groups, its pixel of a strip --------------------------------------
    constexpr int H = decltype(hc)::value;
    constexpr int NGW = H == 0 ? NA : NB;  // feature groups of this wave
    constexpr int UNW = H == 0 ? 1 : 2;    // pool tasks inside its MFMA stream
    constexpr int gbase = H == 0 ? 0 : NA; // its first feature group
    v4i breg[NGW > 0 ? NGW : 1][KS];
#pragma unroll
    for (int g = 0; g < NGW; ++g) {
      const int8_t* brow = p.B + (size_t)((gbase + g) * 32 + l31) * p.Kpad + hh * 16;
#pragma unroll
      for (int j = 0; j < KS; ++j) breg[g][j] = *reinterpret_cast<const v4i*>(brow + j * 32);
    }
    int koff[KS];  // byte offset of this lane's K chunk 2 j + hh inside a pixel's window (chunks past K: 0, zero weights)
    {
      int q = hh, run = 0, rem = hh;
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
    const int pp = w4 * 32 + l31;  // this lane's pixel of a strip -> (row lr inside the strip, column lx)
    int lr = (int)((float)pp * p.rcpOW), lx = pp - lr * p.OW;
    if (lx < 0) { lx += p.OW; --lr; } else if (lx >= p.OW) { lx -= p.OW; ++lr; }
    const int aoff = (lr * p.sq * p.WX + lx * p.sq) * 48;
    const int roff = lx * PITCH + 16 * hh + gbase * 128;

    // ---- pooling role: lane map and tasks as in the kernel above; eight waves, the h = 1 waves first ---------------------
    const I8ieRequant rq = p.rq;
    const int lo_relu = p.relu_lo;
    const float lof = (float)lo_relu;
    const int N4 = p.N >> 2;
    const int PHp = p.PH + 2 * p.ob, PWp = p.PW + 2 * p.ob;
    const __amdgpu_buffer_rsrc_t rsO = __builtin_amdgcn_make_buffer_rsrc(p.out, 0, p.out_bytes, 0x00020000);
    const int hg = ((l31 >= 4 && l31 < 12) || (l31 >= 16 && l31 < 20) || l31 >= 28) ? 1 : 0;
    const int gi = hg == 0 ? (l31 < 4 ? l31 : (l31 < 16 ? l31 - 8 : l31 - 12)) : (l31 < 12 ? l31 - 4 : (l31 < 20 ? l31 - 8 : l31 - 16));
    const int pl = gi & 7;
    const int ql = 2 * (2 * hh + hg) + (gi >> 3);
    const int nqo = N4 >> 3, npo = (p.PW + 7) >> 3, ntask = nqo * npo;
    const int rd_lane = pl * p.ps * PITCH + ql * 16;
    const int st_lane = pl * p.N + ql * 4;
    const int pw = H == 0 ? 4 + w4 : w4;  // this wave's tasks: pw, pw + 8, ...
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
      t.st = ok && k < ntask ? (8 * po) * p.N + qo * 32 + st_lane : (int)0x80000000;  // (past the row's last task: stored nowhere)
      t.oc = p.lds_ocp + (qo * 8 + ql) * 16;
      return t;
    };
    PoolTask tk0[2];
    v4i oc0[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      tk0[u] = task_of(pw + 8 * u);
      oc0[u] = *reinterpret_cast<const v4i*>(smem + tk0[u].oc);
    }
    int nst = 0;  // output stores this wave issued since its last patch request (they retire in order behind it)

    // maxima of a 3 x 3 window of INT32 quads, + oc', the guarded estimate: UNW tasks side by side
    auto pool_math = [&](auto unwc, const v4i (&v)[2][PK * PK], const v4i (&ocv)[2], uint32_t (&d)[2], float (&worst)[2], int (&c4)[2][4]) {
      constexpr int UNW = decltype(unwc)::value;
#pragma unroll
      for (int u = 0; u < UNW; ++u) {
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
    };
    auto pool_finish = [&](auto unwc, const PoolTask (&tk)[2], uint32_t (&d)[2], const float (&worst)[2], const int (&c4)[2][4], int orow) {
      constexpr int UNW = decltype(unwc)::value;
      float wmin = worst[0];
#pragma unroll
      for (int u = 1; u < UNW; ++u) wmin = __builtin_fminf(wmin, worst[u]);
      if (!i8ie_requant_est_ok(wmin)) {  // a value within 2^-13 of a rounding boundary: the exact sequence for its pack
#pragma unroll
        for (int u = 0; u < UNW; ++u)
          if (!i8ie_requant_est_ok(worst[u])) d[u] = i8ie_requant_exact4(c4[u], rq, lo_relu);
      }
#pragma unroll
      for (int u = 0; u < UNW; ++u) __builtin_amdgcn_raw_buffer_store_b32(d[u] ^ p.xor_out, rsO, tk[u].st, orow, 0);
      nst += UNW;
    };
    // one task on its own (what the region below does not take: further rows of a strip, further rounds of a row)
    auto pool_one = [&](const PoolTask& t, const v4i& oc, const int (&rowoff)[PK], int orow) {
      v4i v[2][PK * PK];
#pragma unroll
      for (int dy = 0; dy < PK; ++dy) {
        const uint8_t* const a = smem + (rowoff[dy] + t.rd);
#pragma unroll
        for (int dx = 0; dx < PK; ++dx) v[0][dy * PK + dx] = *reinterpret_cast<const v4i*>(a + dx * PITCH);
      }
      const PoolTask tk[2] = {t, t};
      const v4i ocv[2] = {oc, oc};
      uint32_t d[2];
      float worst[2];
      int c4[2][4];
      pool_math(std::integral_constant<int, 1>{}, v, ocv, d, worst, c4);
      pool_finish(std::integral_constant<int, 1>{}, tk, d, worst, c4, orow);
    };
    auto rowoffs = [&](int rm, int (&rowoff)[PK]) {
#pragma unroll
      for (int dy = 0; dy < PK; ++dy) {
        const int sl = rm + dy;
        rowoff[dy] = p.lds_ring + (sl >= RING ? sl - RING : sl) * p.ringRowB;
      }
    };
    auto orow_of = [&](int img, int j) { return ((img * PHp + j + p.ob) * PWp + p.ob) * p.N; };

    // One strip of this wave's tile: NGW groups x KS MFMAs, and -- POOL -- the first UNW tasks of pooled row jp (ring rows
    // rowoff[], output row orow) inside the same scheduling region
    auto region = [&](auto poolc, int pbuf, int ibm, const StemStrip& em, const int (&rowoff)[PK], int orow) {
      constexpr std::integral_constant<int, UNW> unwc{};
      constexpr bool POOL = decltype(poolc)::value;
      const int npx = (em.hi - em.lo) * p.OW;
      const bool valid = pp < npx;
      const uint8_t* const abase = smem + p.lds_patch + pbuf * p.patchB + (valid ? aoff : 0);
      int slot = ibm + em.lom;
      if (slot >= RING) slot -= RING;
      int ls = slot + lr;
      if (ls >= RING) ls -= RING;
      uint8_t* const wbase = valid ? smem + p.lds_ring + ls * p.ringRowB + roff : smem + p.lds_dump + lane * 16;  // (lanes past the strip: a 2 KiB dump)
      // fragment reads run DEPTH k-steps ahead of their MFMAs (a k-step is NGW MFMAs of 32 cycles; an LDS round trip is
      // 150-200 cycles under load)
      constexpr int DEPTH = NGW >= 2 ? (KS < 4 ? KS : 4) : (KS < 6 ? KS : 6);
      const v16i z = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
      v16i acc[NGW > 0 ? NGW : 1];
      v4i v[2][PK * PK];
      uint32_t d[2] = {0, 0};
      float worst[2] = {rq.fast ? 1.0f : 0.0f, rq.fast ? 1.0f : 0.0f};
      int c4[2][4];
      float ef[2][4], gq[2][4];
      // The pool pass of a task, cut into 16 micro-steps of three vector instructions (component c of the quad: window maximum,
      // + oc', the guarded estimate of i8ie_requant_est4, bit for bit): the unit placed behind one MFMA below
      auto pool_step = [&](auto sc) {
        constexpr int s = decltype(sc)::value;
        constexpr int u = s % UNW, c = (s / UNW) / 4, ph = (s / UNW) % 4;
        auto mx3 = [](int a, int b, int cc) { const int t = a > b ? a : b; return t > cc ? t : cc; };
        if constexpr (ph == 0) {
          int m = mx3(v[u][0][c], v[u][1][c], v[u][2][c]);
          m = mx3(m, v[u][3][c], v[u][4][c]);
          c4[u][c] = mx3(m, v[u][5][c], v[u][6][c]);
        } else if constexpr (ph == 1) {
          c4[u][c] = mx3(c4[u][c], v[u][7][c], v[u][8][c]) + oc0[u][c];  // max(C) + oc' = max(C + oc'): exact integer adds
          ef[u][c] = (float)c4[u][c];
        } else if constexpr (ph == 2) {
          ef[u][c] = __builtin_fmaf(ef[u][c], rq.ms, rq.zpf - 0.5f);
          gq[u][c] = __builtin_amdgcn_fractf(ef[u][c]) - 0.5f;
        } else {
          d[u] = __builtin_amdgcn_cvt_pk_u8_f32(__builtin_fmaxf(ef[u][c], lof), c, d[u]);
          worst[u] = __builtin_fminf(worst[u], __builtin_fabsf(gq[u][c]));
        }
      };
      constexpr int NM = NGW * KS;                  // MFMAs of the strip
      constexpr int NS = POOL ? UNW * 16 : 0;       // micro-steps of the pool pass
      constexpr int PER = NGW >= 2 ? 1 : 2;         // micro-steps behind one MFMA
      constexpr int FIRST = NM - (NS + PER - 1) / PER - 1 > 5 ? NM - (NS + PER - 1) / PER - 1 : (NM > 5 ? 5 : 0);  // the MFMA that takes the first
      __builtin_amdgcn_sched_barrier(0);
      v4i ring[DEPTH];
      static_for<0, DEPTH>([&](auto jc) {
        constexpr int j = decltype(jc)::value;
        ring[j] = *reinterpret_cast<const v4i*>(abase + koff[j]);
      });
      if constexpr (POOL) {
#pragma unroll
        for (int u = 0; u < UNW; ++u)
#pragma unroll
          for (int dy = 0; dy < PK; ++dy) {
            const uint8_t* const a = smem + (rowoff[dy] + tk0[u].rd);
#pragma unroll
            for (int dx = 0; dx < PK; ++dx) v[u][dy * PK + dx] = *reinterpret_cast<const v4i*>(a + dx * PITCH);
          }
      }
      __builtin_amdgcn_sched_barrier(0);
      static_for<0, KS>([&](auto jc) {
        constexpr int j = decltype(jc)::value;
        const v4i af = ring[j % DEPTH];
        if constexpr (j + DEPTH < KS) ring[j % DEPTH] = *reinterpret_cast<const v4i*>(abase + koff[j + DEPTH]);
        static_for<0, NGW>([&](auto ggc) {
          constexpr int gg = decltype(ggc)::value;
          constexpr int i = j * NGW + gg;  // this MFMA
          acc[gg] = __builtin_amdgcn_mfma_i32_32x32x32_i8(breg[gg][j], af, j == 0 ? z : acc[gg], 0, 0, 0);
          if constexpr (i >= FIRST)
            static_for<0, PER>([&](auto qc) {
              constexpr int sidx = (i - FIRST) * PER + decltype(qc)::value;
              if constexpr (sidx < NS) pool_step(std::integral_constant<int, sidx>{});
            });
          __builtin_amdgcn_sched_barrier(0);
        });
      });
      {  // micro-steps the MFMAs did not cover
        constexpr int covered = (NM - FIRST) * PER < NS ? (NM - FIRST) * PER : NS;
        static_for<covered, NS>([&](auto sc) { pool_step(sc); });
      }
      __builtin_amdgcn_sched_barrier(0);
      stamp(0);  // the region
      if constexpr (POOL) pool_finish(unwc, tk0, d, worst, c4, orow);
      // accumulators -> the INT32 ring (lane = pixel, register group q = features 8 q + 4 hh .. + 3 of group gg)
#if defined(STEM_EXP)
      if (STEM_EXP & 2) {  // timing build (wrong results): one store instead of 4 NGW
        v4i o;
        o.x = acc[0][0]; o.y = acc[0][5]; o.z = acc[NGW - 1][10]; o.w = acc[NGW - 1][15];
        *reinterpret_cast<v4i*>(wbase) = o;
        return;
      }
#endif
#pragma unroll
      for (int gg = 0; gg < NGW; ++gg)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          v4i o;
          o.x = acc[gg][q * 4 + 0]; o.y = acc[gg][q * 4 + 1]; o.z = acc[gg][q * 4 + 2]; o.w = acc[gg][q * 4 + 3];
          *reinterpret_cast<v4i*>(wbase + (gg * 32 + 8 * q) * 4) = o;
        }
    };

    auto wait_vm_keep = [&](int keep) {
      switch (keep < 12 ? keep : 12) {
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
        default: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
      }
    };

    // running state: strip g (multiplied), the strip whose patch is requested next, strip g - 1 (pooled)
    int t = 0, un = (int)blockIdx.x, ib = 0, pbuf = 0;
    int tn = 0, imn = (int)blockIdx.x;
    int t1 = 0, im1 = (int)blockIdx.x, ib1 = 0;
    auto next_n = [&]() {
      if (++tn == T) {
        tn = 0;
        imn += (int)gridDim.x;
      }
    };
    if constexpr (H == 0) {
      patch_dma(imn, tn, 0);
      next_n();
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    STEM_BAR();  // (the first patch)
#if defined(I8IE_DIAG)
    if (p.dbg) tq = __builtin_amdgcn_s_memtime();
#endif
    for (int g = 0; g <= G; ++g) {
      // interval g: strip g is multiplied (none at g == G) and the pooled rows strip g - 1 completed leave (none at g == 0)
      if (H == 0 && g + 1 < G) {  // the patch of strip g + 1 -> the other buffer (strip g - 1 was its last reader)
        patch_dma(imn, tn, pbuf ^ 1);
        next_n();
        nst = 0;
      }
      StemStrip em{}, ep{};
      if (g < G) em = strip(un, t);
      if (g >= 1) ep = strip(im1, t1);
      const int img1 = im1 >> p.lg_parts;
#if defined(STEM_EXP)
      const bool have_pool = g >= 1 && ep.j1 > ep.j0 && !(STEM_EXP & 4);  // timing build (wrong results): no pool pass
#else
      const bool have_pool = g >= 1 && ep.j1 > ep.j0;
#endif
      int rm = ib1 + ep.jm;  // ring slot of the first pooled row's first conv row
      if (rm >= RING) rm -= RING;
      int rowoff[PK];
      rowoffs(rm, rowoff);
      const int orow0 = orow_of(img1, ep.j0);
      bool fused = false;
      const bool mult_here = NGW > 0 && g < G && w4 * 32 < (em.hi - em.lo) * p.OW;  // (wave-uniform: a short strip leaves the upper tiles without work)
      if (mult_here) {
        fused = have_pool && pw < ntask;
#if defined(STEM_EXP)
        if (STEM_EXP & 1) fused = false;  // timing build: the pool pass behind the region, not inside it
#endif
        if constexpr (NGW > 0) {
          if (fused) region(std::true_type{}, pbuf, ib, em, rowoff, orow0);
          else region(std::false_type{}, pbuf, ib, em, rowoff, orow0);
        }
      }
      if (have_pool) {  // what the region did not take
        const int done0 = fused ? UNW : 0;  // tasks of the first row already out
        for (int j = ep.j0; j < ep.j1; ++j) {
          if (j > ep.j0) {
            rm += p.ps;
            if (rm >= RING) rm -= RING;
            rowoffs(rm, rowoff);
          }
          const int orow = orow_of(img1, j);
          for (int r = j == ep.j0 ? done0 : 0; pw + 8 * r < ntask; ++r) {
            if (r == 0) pool_one(tk0[0], oc0[0], rowoff, orow);
            else if (r == 1) pool_one(tk0[1], oc0[1], rowoff, orow);
            else {
              const PoolTask tk = task_of(pw + 8 * r);
              pool_one(tk, *reinterpret_cast<const v4i*>(smem + tk.oc), rowoff, orow);
            }
          }
        }
      }
      stamp(1);  // ring stores, the rest of the pool pass
      if constexpr (ACC) {
        if (g >= 1 && ep.hi > ep.lo) {  // conv rows [lo, hi) of strip g - 1: ring -> acc_dbg
          const int tasks = (ep.hi - ep.lo) * p.OW * N4;
          const float rn4 = 1.0f / (float)N4;
          for (int id = tid; id < tasks; id += 512) {
            int q1 = (int)((float)id * rn4), qq = id - q1 * N4;
            if (qq < 0) { qq += N4; --q1; } else if (qq >= N4) { qq -= N4; ++q1; }
            int r = 0, x = q1;
            while (x >= p.OW) {
              x -= p.OW;
              ++r;
            }
            int sl = ib1 + ep.lom + r;
            while (sl >= RING) sl -= RING;
            const v4i a = *reinterpret_cast<const v4i*>(smem + p.lds_ring + sl * p.ringRowB + x * PITCH + qq * 16);
            const v4i o = *reinterpret_cast<const v4i*>(smem + p.lds_ocp + qq * 16);
            const int4 c = make_int4(a.x + o.x, a.y + o.y, a.z + o.z, a.w + o.w);
            *reinterpret_cast<int4*>(p.acc + ((size_t)img1 * (size_t)(p.OH * p.OW) + (size_t)((ep.lo + r) * p.OW + x)) * (size_t)p.N + qq * 4) = c;
          }
        }
      }
      if (g >= 1 && ++t1 == T) {
        t1 = 0;
        ib1 += adv_of(im1);
        if (ib1 >= RING) ib1 -= RING;
        im1 += (int)gridDim.x;
      }
      if (g < G) {
        if (++t == T) {
          t = 0;
          ib += adv_of(un);
          if (ib >= RING) ib -= RING;
          un += (int)gridDim.x;
        }
        pbuf ^= 1;
      }
      // the patch requested at the top of the interval has landed (the stores issued behind it may stay in flight)
      if constexpr (H == 0) {
        if constexpr (ACC) wait_vm_keep(0);
        else wait_vm_keep(nst);
      }
      STEM_BAR();
      stamp(2);  // patch wait, barrier
    }
  };
  if (h == 0) run(std::integral_constant<int, 0>{});
  else run(std::integral_constant<int, 1>{});
#if defined(I8IE_DIAG)
  if (p.dbg && lane == 0)
    for (int i = 0; i < 6; ++i) p.dbg[((size_t)blockIdx.x * 8 + wave) * 8 + i] = ph[i];
#endif
}

template <int NG, int KS, bool ACC>
int launch_stem_fused_t(i8ie_ctx* ctx, const StemArgs& a, int grid, int lds) {
  static bool raised[64] = {};
  const int dev = ctx->device & 63;
  if (!raised[dev]) {
    I8IE_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&stem_fused_kernel<NG, KS, ACC>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    raised[dev] = true;
  }
  stem_fused_kernel<NG, KS, ACC><<<grid, 512, lds, ctx->stream>>>(a);
  I8IE_LAUNCH_CHECK();
  return I8IE_OK;
}}  // namespace

int i8ie_stem_fused_launch(i8ie_ctx* ctx, const StemArgs& a, int NG, int KS, int grid, int lds) {
#define I8IE_STEM_F(NGV, KSV) \
  return a.acc != nullptr ? launch_stem_fused_t<NGV, KSV, true>(ctx, a, grid, lds) : launch_stem_fused_t<NGV, KSV, false>(ctx, a, grid, lds);
#define I8IE_STEM_FK(NGV)           \
  if (KS <= 6) { I8IE_STEM_F(NGV, 6) }   \
  if (KS <= 10) { I8IE_STEM_F(NGV, 10) } \
  I8IE_STEM_F(NGV, kStemMaxKS)
  if (NG == 1) { I8IE_STEM_FK(1) }
  if (NG == 2) { I8IE_STEM_FK(2) }
  I8IE_STEM_FK(3)
#undef I8IE_STEM_FK
#undef I8IE_STEM_F
}
