// i8ie_first.hip -- small-C strided Conv2d (AlexNet conv1: 3 -> 96 channels, 11x11, stride 4)
// as a weights-stationary kernel, plus the fused quantize+repack that feeds it from the FP32 input.
//
// conv1 is 9 % of AlexNet's MACs but was 28 % of the device time as quantize + repack + implicit
// GEMM: with K = 363 the GEMM is bound by staging (every input byte gathered ~8 times, the 61 KB
// weight panel re-staged for every 128 pixels) and by its epilogue.  Here instead:
//   * quantize_repack_kernel: FP32 NCHW -> "grouped" u8 image [n][Hp][WG][16] in one pass
//     (16 bytes = 4 pixels x (3 channels + pad byte), zero-point padded borders): the reference's
//     quantize (x / scale + zp, truncate, low 8 bits; src/quantize_utils.cc:44-52) with a fast
//     path that is exact by the same argument as the requantiser's, IEEE sequence as fallback;
//   * conv_smallc_kernel: persistent blocks, one wave per 32 output features whose weight slice
//     [32][K] lives in VGPRs for the whole kernel (no B traffic, no LDS for B); a block walks bands
//     of RB output rows, each band's input patch is a CONTIGUOUS piece of the grouped image (stored
//     already re-biased ^0x80 by the repack kernels), copied once into LDS by LDS-DMA (1 KiB per
//     wave-instruction, no VGPR pass) while the previous band is being multiplied; an MFMA A
//     fragment is one ds_read_b128 at base(pixel) + offset(tap); epilogue = the igemm requantiser,
//     wave-private LDS transpose, 16-byte NHWC stores.
// INT32 accumulators are the reference's sums (K order permuted, padded taps have zero weights).
#include <cstdlib>
#include <type_traits>

#include "i8ie_internal.h"
#include "i8ie_requant.h"

namespace {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

using FRequant = I8ieRequant;  // i8ie_requant.h
__device__ __forceinline__ uint32_t frequant_pack4(int c0, int c1, int c2, int c3, const FRequant& q, int lo,
                                                   float lof) {
  const int c[4] = {c0, c1, c2, c3};
  return i8ie_requant_pack4(c, q, lo, lof);
}

constexpr int kMaxKS = 24;

// compile-time loop: the body sees its index as a constant (a `#pragma unroll` loop left hipcc holding 26 more
// VGPRs in the first-layer kernel: 190 against 164)
template <int B, int E, typename F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (B < E) {
    f(std::integral_constant<int, B>{});
    static_for<B + 1, E>(f);
  }
}

struct FirstArgs {
  // FUSEQ: the FP32 NCHW input itself; every block quantises the rows of its bands into the LDS patch
  // (src/quantize_utils.cc:44-52 followed by the grouping of repack_smallc), no intermediate image in HBM
  const float* x;
  int xc, xh, xw, xpad;
  float q_scale, q_zpf, q_rscale;
  uint32_t q_zp;
  const uint8_t* img;  // grouped image [n][Hp][WG][16], bytes re-biased (^0x80)
  size_t img_bytes;
  int Hp;
  int KH, KWG, sh, swg, OH, OW;
  int RB, bands_per_img, total_bands;
  int PR, WG;  // patch rows, 16-byte groups per patch row
  const int8_t* B;  // [Npad][Kpad] K ordered (kh, group, px, ch), zero padded
  int Kpad, N;
  const int32_t* ocp;
  FRequant rq;
  int relu_lo;
  uint8_t* out;  // NHWC [n][OH + 2ob][OW + 2ob][N]
  int ob;
  int toff[2 * kMaxKS];  // LDS byte offset of K chunk q inside a pixel's window
  int32_t* acc;          // ACC kernels: [n * OH * OW][N] pre-requant accumulators (the cblas_gemm_s8u8s32 result, src/conv2d.cc:131-133)
};

// ---- fused quantize + repack: FP32 NCHW -> grouped u8 [n][Hp][WG][16] -------------------------
// exact quantize: t = x / scale + zp (fp32 divide, fp32 add), q = ((int)t) & 0xFF.
// Fast path: est = fma(x, fl(1/scale), zp).  For |est| < 1024, |t - est| < 2e-4 (one rounding each
// of 1/scale and of the fma against the two roundings of the reference, all on magnitudes < 1151),
// so when est is further than 2^-10 from an integer t truncates to the same integer.
__device__ __forceinline__ uint32_t quant_exact(float x, float scale, float zpf, float rscale) {
  const float est = __builtin_fmaf(x, rscale, zpf);
  const float fr = __builtin_amdgcn_fractf(est);
  if (__builtin_fabsf(est) < 1024.0f && __builtin_fabsf(fr - 0.5f) <= 0.5f - 9.765625e-4f)
    return (uint32_t)((int)est) & 0xFFu;
  const float t = x / scale + zpf;  // src/quantize_utils.cc:49
  return (uint32_t)((int)t) & 0xFFu;
}

// IDX = uint32_t while the image has fewer than 2^31 groups (three 64-bit divisions per group were a large part
// of this kernel's 53 M VALU wave-instructions per launch), int64_t beyond
// VEC: the 4 pixels of a group are read per channel as two 8-byte loads from an in-image, 8-byte-aligned base
// (w and pad even), whole 512-byte runs per wave-instruction instead of four 4-byte loads with a 16-byte stride
template <typename IDX, bool VEC>
__global__ __launch_bounds__(256) void quantize_repack_kernel(const float* __restrict__ x, uint8_t* __restrict__ out,
                                                              int64_t total, int c, int h, int w, int Hp, int WG,
                                                              int pad, float scale, float zpf, float rscale,
                                                              uint32_t zp) {
  const IDX gstride = (IDX)gridDim.x * 256;
  const size_t cs = (size_t)h * w;
  for (IDX e = (IDX)blockIdx.x * 256 + threadIdx.x; e < (IDX)total; e += gstride) {
    const IDX t = e / (IDX)WG;
    const int g = (int)(e - t * (IDX)WG);
    const IDX img = t / (IDX)Hp;
    const int yp = (int)(t - img * (IDX)Hp);
    const int y = yp - pad, x0 = 4 * g - pad;
    const bool yin = y >= 0 && y < h;
    const int yc = y < 0 ? 0 : (y >= h ? h - 1 : y);
    const float* plane = x + ((size_t)img * c * h + yc) * w;
    float v[12];
    if constexpr (VEC) {
      const int xb = x0 < 0 ? 0 : (x0 > w - 4 ? w - 4 : x0);
      const int sft = x0 - xb;  // in-image pixel px of this group is element px + sft of the 4 loaded ones
#pragma unroll
      for (int ch = 0; ch < 3; ++ch) {
        const float* row = plane + (ch < c ? ch : 0) * cs + xb;
        const float2 a = *reinterpret_cast<const float2*>(row), b = *reinterpret_cast<const float2*>(row + 2);
        const float L[4] = {a.x, a.y, b.x, b.y};
#pragma unroll
        for (int px = 0; px < 4; ++px) {
          int i = px + sft;
          i = i < 0 ? 0 : (i > 3 ? 3 : i);
          v[ch * 4 + px] = i == 0 ? L[0] : (i == 1 ? L[1] : (i == 2 ? L[2] : L[3]));
        }
      }
    } else {
      int xc[4];
#pragma unroll
      for (int px = 0; px < 4; ++px) {
        const int xx = x0 + px;
        xc[px] = xx < 0 ? 0 : (xx >= w ? w - 1 : xx);
      }
#pragma unroll
      for (int ch = 0; ch < 3; ++ch) {  // unconditional (clamped) loads: all 12 in flight together
        const float* row = plane + (ch < c ? ch : 0) * cs;
#pragma unroll
        for (int px = 0; px < 4; ++px) v[ch * 4 + px] = row[xc[px]];
      }
    }
    uint32_t wds[4];
#pragma unroll
    for (int px = 0; px < 4; ++px) {
      const int xx = x0 + px;
      const bool in = yin && xx >= 0 && xx < w;
      uint32_t d = zp * 0x01010101u;  // padding pixel / pad byte: the input zero point (src/conv2d.cc:24-28)
      if (in) {
        d = zp << 24;
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) d |= (ch < c ? quant_exact(v[ch * 4 + px], scale, zpf, rscale) : zp) << (8 * ch);
      }
      wds[px] = d;
    }
    // stored re-biased (u8 ^ 0x80 = s8 + 128 - 128): the contraction kernel's signed MFMA operand as is
    reinterpret_cast<uint4*>(out)[e] =
        make_uint4(wds[0] ^ 0x80808080u, wds[1] ^ 0x80808080u, wds[2] ^ 0x80808080u, wds[3] ^ 0x80808080u);
  }
}

// One 16-byte group of the grouped image straight from the FP32 planes: pixels x0 .. x0 + 3 of image row y, three
// channels + the zero point as fourth byte, out-of-image pixels = the zero point (src/conv2d.cc:24-28), re-biased
// (^ 0x80).  Split in two so that the twelve loads of the next group fly while the current one is multiplied.
struct GroupLoad {
  float v[12];
  int ok;  // bit px: pixel inside the image
};
__device__ __forceinline__ void group_issue(const FirstArgs& p, const __amdgpu_buffer_rsrc_t& rs_img, int yp, int g,
                                            GroupLoad& L) {
  // rs_img covers the c planes of ONE image; per-lane offsets are 32-bit, the plane offset rides in soffset
  const int y = yp - p.xpad, x0 = 4 * g - p.xpad;
  const bool yin = y >= 0 && y < p.xh;
  const int yc = y < 0 ? 0 : (y >= p.xh ? p.xh - 1 : y);
  const int cs4 = p.xh * p.xw * 4;
  L.ok = 0;
#pragma unroll
  for (int px = 0; px < 4; ++px) {
    const int xx = x0 + px;
    const int xcl = xx < 0 ? 0 : (xx >= p.xw ? p.xw - 1 : xx);
    if (yin && xx >= 0 && xx < p.xw) L.ok |= 1 << px;
    const int voff = (yc * p.xw + xcl) * 4;
#pragma unroll
    for (int ch = 0; ch < 3; ++ch)  // unconditional, clamped
      L.v[ch * 4 + px] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_img, voff, (ch < p.xc ? ch : 0) * cs4, 0));
  }
}
__device__ __forceinline__ uint4 group_finish(const FirstArgs& p, const GroupLoad& L) {
  uint32_t wds[4];
#pragma unroll
  for (int px = 0; px < 4; ++px) {
    uint32_t d = p.q_zp * 0x01010101u;
    if (L.ok & (1 << px)) {
      d = p.q_zp << 24;
#pragma unroll
      for (int ch = 0; ch < 3; ++ch)
        d |= (ch < p.xc ? quant_exact(L.v[ch * 4 + px], p.q_scale, p.q_zpf, p.q_rscale) : p.q_zp) << (8 * ch);
    }
    wds[px] = d ^ 0x80808080u;
  }
  return make_uint4(wds[0], wds[1], wds[2], wds[3]);
}

// ACC: also dump the INT32 accumulators (acc_dbg of the C-ABI) -- a separate instantiation, the default one is untouched
template <int KS, bool FUSEQ, bool ACC>
__global__ __launch_bounds__(512) void conv_smallc_kernel(FirstArgs p) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  const int nthreads = blockDim.x;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int hh = lane >> 5;
  const int nwaves = nthreads >> 6;
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  // LDS pitch of a patch: whole 1 KiB DMA pieces (the quantising fill writes exactly its groups)
  const int patch_bytes = FUSEQ ? p.PR * p.WG * 16 : (p.PR * p.WG * 16 + 1023) & ~1023;
  uint8_t* scratch = smem + 2 * patch_bytes + wave * (32 * 36);  // wave-private 32 px x 36 B

  // ---- this wave's weight slice -> registers (stays for the whole kernel) ---------------------
  v4i breg[KS];
  {
    const int8_t* brow = p.B + (size_t)(wave * 32 + (lane & 31)) * p.Kpad + hh * 16;
#pragma unroll
    for (int j = 0; j < KS; ++j) {
      const v4i z = {0, 0, 0, 0};
      breg[j] = (j * 32 + 32 <= p.Kpad) ? *reinterpret_cast<const v4i*>(brow + j * 32) : z;
    }
  }
  int init[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) init[r] = p.ocp[wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh];
  int koff[KS];
#pragma unroll
  for (int j = 0; j < KS; ++j) koff[j] = hh ? p.toff[2 * j + 1] : p.toff[2 * j];

  const int groups = p.PR * p.WG;
  const int pieces = (groups + 63) >> 6;                    // 1 KiB DMA wave-instructions per patch
  // iterations in which every wave issues one DMA piece / every thread quantises one group (FUSEQ)
  const int fill_iters = FUSEQ ? (groups + nthreads - 1) / nthreads : (pieces + nwaves - 1) / nwaves;
  const FRequant rq = p.rq;
  const int lo = p.relu_lo;
  const float lof = (float)lo;
  const int OHp = p.OH + 2 * p.ob, OWp = p.OW + 2 * p.ob;

  auto band_origin = [&](int band, int& img, int& rb0, int& rows) {
    img = band / p.bands_per_img;
    rb0 = (band - img * p.bands_per_img) * p.RB;
    rows = p.OH - rb0 < p.RB ? p.OH - rb0 : p.RB;
  };
  // a band's patch = PR consecutive rows of the grouped image = one contiguous run of 16-byte groups;
  // the buffer descriptor covers exactly that run, so the last (partial) DMA piece reads zeros past it
  auto band_rsrc = [&](int band) -> __amdgpu_buffer_rsrc_t {
    int img, rb0, rows;
    band_origin(band, img, rb0, rows);
    const size_t off = ((size_t)img * p.Hp + (size_t)rb0 * p.sh) * p.WG * 16;
    const size_t left = p.img_bytes - off;  // the last band of the last image is shorter than a full patch
    const int bytes = left < (size_t)groups * 16 ? (int)left : groups * 16;
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(p.img + off), 0, bytes, 0x00020000);
  };
  // DMA piece q of a patch: 64 groups -> LDS bytes [q * 1024, +1024) of `dst`
  auto dma_piece = [&](const __amdgpu_buffer_rsrc_t& rs, uint8_t* dst, int q) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(dst + q * 1024), 16,
                                             (q * 64 + lane) * 16, 0, 0, 0);
  };

  int band = blockIdx.x;
  if (band >= p.total_bands) return;
  int cur = 0;
  auto image_rsrc = [&](int img) -> __amdgpu_buffer_rsrc_t {
    const size_t per = (size_t)p.xc * p.xh * p.xw;
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x + (size_t)img * per), 0, (int)(per * 4), 0x00020000);
  };
  // FUSEQ: group gi of a patch = patch row gi / WG, group gi % WG; thread t owns groups t, t + nthreads, ...
  const int g_step_row = nthreads / p.WG, g_step_col = nthreads - g_step_row * p.WG;
  if constexpr (FUSEQ) {
    int img, rb0, rows;
    band_origin(band, img, rb0, rows);
    const __amdgpu_buffer_rsrc_t rs_img = image_rsrc(img);
    for (int gi = tid; gi < groups; gi += nthreads) {
      GroupLoad L;
      group_issue(p, rs_img, rb0 * p.sh + gi / p.WG, gi % p.WG, L);
      *reinterpret_cast<uint4*>(smem + (size_t)gi * 16) = group_finish(p, L);
    }
  } else {
    const __amdgpu_buffer_rsrc_t rs0 = band_rsrc(band);
    for (int q = wave_u; q < pieces; q += nwaves) dma_piece(rs0, smem, q);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __syncthreads();

  for (; band < p.total_bands; band += gridDim.x) {
    int img, rb0, rows;
    band_origin(band, img, rb0, rows);
    const int npix = rows * p.OW;
    const int ntiles = (npix + 31) >> 5;
    const int nband = band + gridDim.x;
    const bool has_next = nband < p.total_bands;
    const __amdgpu_buffer_rsrc_t nrs = band_rsrc((has_next && !FUSEQ) ? nband : band);
    // FUSEQ: this thread's walk over the next band's groups (row, group) and the loads in flight for the next one
    int nimg = 0, nrb0 = 0, nrows = 0, frow = tid / p.WG, fcol = tid - (tid / p.WG) * p.WG;
    GroupLoad nxt;
    nxt.ok = 0;
    if (FUSEQ && has_next) band_origin(nband, nimg, nrb0, nrows);
    const __amdgpu_buffer_rsrc_t rs_nimg = image_rsrc(FUSEQ ? nimg : 0);
    if (FUSEQ && has_next && tid < groups) group_issue(p, rs_nimg, nrb0 * p.sh + frow, fcol, nxt);
    const uint8_t* patch = smem + cur * patch_bytes;
    uint8_t* npatch = smem + (cur ^ 1) * patch_bytes;
    const int iters = ntiles > fill_iters ? ntiles : fill_iters;
    // pixel index of this lane inside the band, as (row, col): the MFMA lane map (lane & 31) and the
    // store map (lane >> 1)
    int pr = 0, pc = lane & 31, sr = 0, sc = lane >> 1;
    while (pc >= p.OW) { pc -= p.OW; ++pr; }
    while (sc >= p.OW) { sc -= p.OW; ++sr; }

    for (int it = 0; it < iters; ++it) {
      // the next band's patch streams in beside the tiles of this band: one DMA piece per wave per iteration
      if constexpr (FUSEQ) {
        // group `it * nthreads + tid` of the next band: its loads were issued one iteration ago; issue the next
        // group's loads, then quantise this one into the other patch (the MFMAs of this iteration follow)
        const int gi = it * nthreads + tid;
        if (has_next && gi < groups) {
          const GroupLoad cur_l = nxt;
          frow += g_step_row;
          fcol += g_step_col;
          if (fcol >= p.WG) {
            fcol -= p.WG;
            ++frow;
          }
          if (gi + nthreads < groups) group_issue(p, rs_nimg, nrb0 * p.sh + frow, fcol, nxt);
          *reinterpret_cast<uint4*>(npatch + (size_t)gi * 16) = group_finish(p, cur_l);
        }
      } else {
        const int q = it * nwaves + wave_u;
        if (has_next && q < pieces) dma_piece(nrs, npatch, q);
      }
      // one 32-pixel x 32-feature tile of the CURRENT band
      if (it < ntiles) {
        // (row, col) of this lane's pixel: walked incrementally, no division
        const int r = pr < rows ? pr : rows - 1, col = pr < rows ? pc : p.OW - 1;
        const uint8_t* abase = patch + ((r * p.sh) * p.WG + col * p.swg) * 16;
        v16i acc;
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[q] = init[q];
        // A fragments run DEPTH k-steps ahead of the MFMA chain (one ds_read_b128 each): with a
        // single accumulator chain an LDS round trip per step would otherwise be exposed
        constexpr int DEPTH = KS < 6 ? KS : 6;
        v4i ring[DEPTH];
        static_for<0, DEPTH>([&](auto jc) {
          constexpr int j = decltype(jc)::value;
          ring[j] = *reinterpret_cast<const v4i*>(abase + koff[j]);
        });
        __builtin_amdgcn_sched_barrier(0);  // hipcc otherwise sinks every read next to its MFMA (1 in flight)
        static_for<0, KS>([&](auto jc) {
          constexpr int j = decltype(jc)::value;
          const v4i af = ring[j % DEPTH];
          if constexpr (j + DEPTH < KS) {
            ring[j % DEPTH] = *reinterpret_cast<const v4i*>(abase + koff[j + DEPTH]);
            __builtin_amdgcn_sched_barrier(0);
          }
          acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(breg[j], af, acc, 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
        });
        // epilogue: lane = pixel, regs = features (4 consecutive per group)
        if constexpr (ACC) {
          const int am = it * 32 + (lane & 31);
          if (am < npix) {
            int32_t* arow = p.acc + ((size_t)img * (size_t)(p.OH * p.OW) + (size_t)(rb0 * p.OW + am)) * (size_t)p.N + wave * 32 + 4 * hh;
#pragma unroll
            for (int g = 0; g < 4; ++g)
              *reinterpret_cast<int4*>(arow + 8 * g) = make_int4(acc[g * 4 + 0], acc[g * 4 + 1], acc[g * 4 + 2], acc[g * 4 + 3]);
          }
        }
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const uint32_t packed =
              frequant_pack4(acc[g * 4 + 0], acc[g * 4 + 1], acc[g * 4 + 2], acc[g * 4 + 3], rq, lo, lof);
          *reinterpret_cast<uint32_t*>(scratch + (lane & 31) * 36 + 8 * g + 4 * hh) = packed;
        }
        // wave-private transpose: lane l stores 16 bytes (features 16*(l&1) .. +15) of pixel l >> 1
        const int sp = lane >> 1, sh16 = lane & 1;
        const int sm = it * 32 + sp;
        if (sm < npix) {
          const uint32_t* s = reinterpret_cast<const uint32_t*>(scratch + sp * 36 + sh16 * 16);
          const uint4 val = make_uint4(s[0], s[1], s[2], s[3]);
          const size_t pix = ((size_t)img * OHp + rb0 + sr + p.ob) * OWp + sc + p.ob;
          *reinterpret_cast<uint4*>(p.out + pix * p.N + wave * 32 + sh16 * 16) = val;
        }
      }
      // advance both pixel walks by 32 pixels
      pc += 32;
      while (pc >= p.OW) { pc -= p.OW; ++pr; }
      sc += 32;
      while (sc >= p.OW) { sc -= p.OW; ++sr; }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's DMA pieces have landed
    __syncthreads();  // everyone done with patch[cur]; patch[cur^1] complete
    cur ^= 1;
  }
}

template <int KS, bool FUSEQ, bool ACC>
int launch_first_qa(i8ie_ctx* ctx, const FirstArgs& a, int blocks, int threads, size_t lds) {
  I8IE_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_smallc_kernel<KS, FUSEQ, ACC>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  conv_smallc_kernel<KS, FUSEQ, ACC><<<blocks, threads, lds, ctx->stream>>>(a);
  I8IE_LAUNCH_CHECK();
  return I8IE_OK;
}
template <int KS, bool FUSEQ>
int launch_first_q(i8ie_ctx* ctx, const FirstArgs& a, int blocks, int threads, size_t lds) {
  return a.acc != nullptr ? launch_first_qa<KS, FUSEQ, true>(ctx, a, blocks, threads, lds) : launch_first_qa<KS, FUSEQ, false>(ctx, a, blocks, threads, lds);
}
template <int KS>
int launch_first(i8ie_ctx* ctx, const FirstArgs& a, int blocks, int threads, size_t lds) {
#if defined(I8IE_DIAG)  // quantize inside the patch fill: measured slower, diagnostic build only
  if (a.x != nullptr) return launch_first_q<KS, true>(ctx, a, blocks, threads, lds);
#endif
  return launch_first_q<KS, false>(ctx, a, blocks, threads, lds);
}

}  // namespace

struct I8ieFirstCall {
  const float* x;        // FP32 NCHW input (fused quantize), or nullptr when `grouped` is given
  const uint8_t* grouped;  // grouped u8 image produced elsewhere (repack_smallc), or nullptr
  uint8_t* scratch;      // room for the grouped image when x is given: n * Hp * WG * 16 bytes
  int n, c, h, w;
  float q_scale;
  int q_zp;
  int KH, KW, KWG, stride, pad, OH, OW;
  const int8_t* B;
  int Kpad, K2, N;
  const int32_t* ocp;
  float s_in, s_w, s_out;
  int zp_out, relu;
  uint8_t* out;
  int ob;
  int32_t* acc;  // null, or [n * OH * OW][N]
};

int i8ie_first_supported(int c, int stride, int n_out, int K2, int KH, int KWG, int OW) {
#if defined(I8IE_DIAG)
  static const bool off = std::getenv("I8IE_NO_SMALLC") != nullptr;  // A/B aid: route path B through the generic kernel
  if (off) return 0;
#endif
  if (c > 3 || stride % 4 != 0) return 0;  // 3 data channels + 1 pad byte per pixel
  if (n_out % 32 != 0 || n_out / 32 > 8) return 0;
  if ((K2 + 31) / 32 > kMaxKS) return 0;
  const int WG = (OW - 1) * (stride / 4) + KWG;
  if ((size_t)KH * WG * 16 * 2 > 60 * 1024) return 0;  // even one-row bands would not fit
  return 1;
}

// grouped-image geometry shared with repack_smallc (i8ie_igemm.hip): Hp = (OH-1)*stride + KH rows,
// WG = (OW-1)*stride/4 + KWG groups per row
size_t i8ie_first_scratch_bytes(int n, int KH, int KWG, int stride, int OH, int OW) {
  const size_t Hp = (size_t)(OH - 1) * stride + KH, WG = (size_t)(OW - 1) * (stride / 4) + KWG;
  return (size_t)n * Hp * WG * 16;
}

int i8ie_first_launch(i8ie_ctx* ctx, const I8ieFirstCall& c) {
  FirstArgs a{};
  a.KH = c.KH; a.KWG = c.KWG; a.sh = c.stride; a.swg = c.stride / 4; a.OH = c.OH; a.OW = c.OW;
  a.WG = (c.OW - 1) * a.swg + c.KWG;
  a.Hp = (c.OH - 1) * c.stride + c.KH;
  const uint8_t* grouped = c.grouped;
  // FP32 input: a separate HBM-bound pass writes the grouped u8 image first (default), or, with $I8IE_FIRST_FUSED,
  // every block quantises its bands inside the patch fill.  Measured on MI355X at batch 1000 (round 2): the fused
  // form is bit-exact but 0.67 ms against 0.22 + 0.17 ms: ~100 VALU operations per 16-byte group land in waves
  // that are already VALU-bound (12 VALU per MFMA in the epilogue), the halo rows of every band are quantised
  // again (x 1.35), and 204 VGPRs leave two blocks per CU instead of three.  Kept as a tested opt-in.
#if defined(I8IE_DIAG)
  static const bool fused_opt = std::getenv("I8IE_FIRST_FUSED") != nullptr;
  const bool fuse = c.x != nullptr && (fused_opt || ctx->variant == 60);  // (variant 60: per-ctx switch for tests / A-B runs)
#else
  const bool fuse = false;
#endif
  if (fuse) {
    a.x = c.x; a.xc = c.c; a.xh = c.h; a.xw = c.w; a.xpad = c.pad;
    a.q_scale = c.q_scale; a.q_zpf = (float)c.q_zp; a.q_rscale = 1.0f / c.q_scale; a.q_zp = (uint32_t)(c.q_zp & 0xFF);
  }
  if (c.x != nullptr && !fuse) {
    const int64_t total = (int64_t)c.n * a.Hp * a.WG;
    I8ieProfScope prof(ctx, "quantize_repack_f32", 0.0, 4.0 * c.n * c.c * c.h * c.w + 16.0 * total);
    int64_t blocks = (total + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;
    bool vec = (c.w & 1) == 0 && (c.pad & 1) == 0 && c.w >= 4 && (reinterpret_cast<uintptr_t>(c.x) & 7u) == 0;
#if defined(I8IE_DIAG)
    if (ctx->variant == 61) vec = false;  // (the scalar-load form, for comparison)
#endif
    const float rs = 1.0f / c.q_scale;
    const uint32_t zp8 = (uint32_t)(c.q_zp & 0xFF);
    if (total < ((int64_t)1 << 31) - 256 * 4096) {
      if (vec)
        quantize_repack_kernel<uint32_t, true><<<(int)blocks, 256, 0, ctx->stream>>>(c.x, c.scratch, total, c.c, c.h, c.w, a.Hp, a.WG, c.pad,
                                                                                  c.q_scale, (float)c.q_zp, rs, zp8);
      else
        quantize_repack_kernel<uint32_t, false><<<(int)blocks, 256, 0, ctx->stream>>>(c.x, c.scratch, total, c.c, c.h, c.w, a.Hp, a.WG, c.pad,
                                                                                   c.q_scale, (float)c.q_zp, rs, zp8);
    } else {
      quantize_repack_kernel<int64_t, false><<<(int)blocks, 256, 0, ctx->stream>>>(c.x, c.scratch, total, c.c, c.h, c.w, a.Hp, a.WG, c.pad,
                                                                                c.q_scale, (float)c.q_zp, rs, zp8);
    }
    I8IE_LAUNCH_CHECK();
    grouped = c.scratch;
  }
  a.img = grouped;
  a.img_bytes = (size_t)c.n * a.Hp * a.WG * 16;
  // rows per band: as many as keep two patches within ~44 KB (3 blocks per CU)
  int RB = 1;
  size_t patch_budget = fuse ? 50 * 1024 : 44 * 1024;  // bytes for the two patches of a block (fused: taller bands
                                                        // re-quantise fewer halo rows; 50 KB still fits 3 blocks per CU)
#if defined(I8IE_DIAG)
  if (const char* e = std::getenv("I8IE_FIRST_PATCH_KB")) patch_budget = (size_t)std::atoi(e) * 1024;
#endif
  while (RB < c.OH && (size_t)(RB * a.sh + c.KH) * a.WG * 16 * 2 <= patch_budget && (RB + 1) * c.OW <= 512) ++RB;
  a.RB = RB;
  a.PR = (RB - 1) * a.sh + c.KH;
  a.bands_per_img = (c.OH + RB - 1) / RB;
  a.total_bands = a.bands_per_img * c.n;
  a.B = c.B; a.Kpad = c.Kpad; a.N = c.N; a.ocp = c.ocp;
  a.rq = i8ie_make_requant(c.s_in, c.s_w, c.s_out, c.zp_out);
  a.relu_lo = c.relu ? c.zp_out : 0;
  a.out = c.out; a.ob = c.ob; a.acc = c.acc;
  const int ks_needed = (c.K2 + 31) / 32;
  const int nchunks = c.KH * c.KWG;
  for (int q = 0; q < 2 * kMaxKS; ++q) {
    const int kh = q / c.KWG, g = q - kh * c.KWG;
    a.toff[q] = q < nchunks ? (kh * a.WG + g) * 16 : 0;  // chunks past K carry zero weights
  }
  const int waves = c.N / 32, threads = waves * 64;
  const size_t patch_lds = fuse ? (size_t)a.PR * a.WG * 16 : (((size_t)a.PR * a.WG * 16 + 1023) & ~(size_t)1023);
  const size_t lds = 2 * patch_lds + (size_t)waves * 32 * 36;
  int per_cu = 3;  // blocks launched per CU (measured on AlexNet conv1 with the DMA fill: 3 beats 2, 4 and 6)
#if defined(I8IE_DIAG)
  if (const char* e = std::getenv("I8IE_FIRST_BLOCKS_PER_CU")) per_cu = std::atoi(e) > 0 ? std::atoi(e) : per_cu;
#endif
  int blocks = 256 * per_cu;
  if (blocks > a.total_bands) blocks = a.total_bands;
  const double ops = 2.0 * c.n * c.OH * c.OW * (double)c.N * c.c * c.KH * c.KW;
  const double bytes = 16.0 * c.n * a.Hp * a.WG + (double)c.n * c.OH * c.OW * c.N;
  const double bytes_f = 4.0 * c.n * c.c * c.h * c.w + (double)c.n * c.OH * c.OW * c.N;
  I8ieProfScope prof(ctx, fuse ? "conv_smallc_f32in" : "conv_smallc_wstat", ops, fuse ? bytes_f : bytes);
  if (ks_needed <= 6) return launch_first<6>(ctx, a, blocks, threads, lds);
  if (ks_needed <= 10) return launch_first<10>(ctx, a, blocks, threads, lds);
  if (ks_needed <= 17) return launch_first<17>(ctx, a, blocks, threads, lds);
  return launch_first<kMaxKS>(ctx, a, blocks, threads, lds);
}
