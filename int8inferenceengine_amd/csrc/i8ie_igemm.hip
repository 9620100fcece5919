// i8ie_igemm.hip -- second-generation contraction kernel: implicit-GEMM Conv2d
// over NHWC u8 activations (no materialised im2col) and the same core for Linear.
//
//   C[r][j] = sum_k A_u8[r][k] * W_s8[j][k] + oc[j]        exact int32, = the
//   cblas_gemm_s8u8s32 result of src/conv2d.cc:131-133 / src/fully_connected.cc:39-41
//   (K is walked in (kh, kw, c) order instead of (c, kh, kw); integer sums commute)
//
// Differences from the v1 kernel (i8ie_gemm.hip, kept as the any-geometry fallback):
//   * A operand is gathered straight from the NHWC activation tensor: a row of the
//     virtual im2col matrix is an output pixel, a 16-byte K chunk is 16 consecutive
//     channels of one (kh, kw) tap; out-of-bounds taps are filled with the input
//     zero point (src/conv2d.cc:24-28).  Removes the im2col write+read entirely.
//   * BK = 128 bytes: 8 lanes cover one 128-byte line of a row, so global loads are
//     whole cache lines (MI355X guide: fragment-shaped 64-B pieces cost TA cycles).
//   * MFMA operands are swapped (weights as the row operand): the accumulator then
//     has the activation row on the lane and 4 consecutive output features in 4
//     consecutive registers, so the row-major (= NHWC) epilogue packs 4 u8 per lane.
//   * The requantiser has a fast path that is provably identical to the reference's
//     ((float)C*s_in)*s_w/s_out + zp sequence and falls back to that exact sequence
//     whenever the fused estimate is within 2^-11 of an integer boundary.
//   * Optional fused ReLU (clamp-low at zp_out; relu<u8> of src/functional.cc:15-26).
#include "i8ie_internal.h"

namespace {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

constexpr int BK2 = 128;  // bytes of K per LDS tile: 8 chunks of 16 B, 4 MFMA k-steps

// 16-B chunk `c` (0..7) of tile row `row`; XOR with (row>>1)&7 makes every 16-lane
// ds_read_b128 group (rows distinct mod 16, same c) hit 16 distinct 4-bank slots.
__device__ __forceinline__ int lds_off2(int row, int c) { return row * BK2 + ((c ^ ((row >> 1) & 7)) << 4); }

struct Requant {
  float sa, sb, sc, zpf, ms;
  int fast;
};

// src/quantize_utils.cc:30-33, exact sequence
__device__ __forceinline__ int requant_exact(float cf, const Requant& q) {
  float deq = (cf * q.sa) * q.sb;
  float v = deq / q.sc + q.zpf;
  return (v >= 255.0f) ? 255 : ((v < 0.0f) ? 0 : (int)v);
}

// Fast path.  est = fma(cf, ms, zp) with ms = fl(s_in*s_w/s_out) differs from the
// reference's value by < 1.6e-4 whenever -1 < est < 256 (4 roundings of <= 2^-24
// relative on magnitudes < 512), so if est is more than 2^-11
// away from both neighbouring integers the reference value lies in the same unit
// interval and trunc/clamp give clamp(floor(est), 0, 255).  est >= 256 / est <= -1
// are the two clamps with the same margin.  Otherwise: the exact sequence.
__device__ __forceinline__ int requant(int c, const Requant& q) {
  const float cf = (float)c;
  if (q.fast) {
    const float est = __builtin_fmaf(cf, q.ms, q.zpf);
    const float fl = __builtin_floorf(est);
    const float fr = est - fl;
    const bool inside = (est > -1.0f) && (est < 256.0f);
    const bool safe = !inside || ((fr > 4.8828125e-4f) && (fr < 1.0f - 4.8828125e-4f));
    if (safe) {
      int k = (int)fl;  // saturating conversion; est is finite here or far outside
      k = est >= 256.0f ? 255 : (est <= -1.0f ? 0 : k);
      return k < 0 ? 0 : (k > 255 ? 255 : k);
    }
  }
  return requant_exact(cf, q);
}

struct IgemmArgs {
  const uint8_t* A;
  long lda;  // AMODE 0
  int M;
  int Kchunks;  // valid 16-B chunks of K
  // AMODE 1 (NHWC gather)
  int H, W, C16;  // C16 = channels / 16
  int KH, KW, sh, sw, ph, pw, OH, OW;
  uint32_t zp_fill;  // input zero point replicated into 4 bytes
  // B
  const int8_t* B;  // [Npad][Kpad] zero padded, K order matches the A walk
  int Kpad;
  int N;
  const int32_t* ocp;   // [N] oc[j] + 128 * wsum[j]
  const float* biasf;   // [N] (float)q_b[j] / s_in, Linear only (src/fully_connected.cc:44); else nullptr
  Requant rq;
  int zp_out;
  int relu;
  int vec_store;  // N % 16 == 0 and out 16-byte aligned: 16-B row stores
  uint8_t* out;  // [M][N]
  int32_t* acc;  // nullptr or [M][N]: C before the Linear bias step / before requant
};

template <int AMODE, int WM, int WN, int TM, int TN>
__global__ __launch_bounds__(WM* WN * 64) void igemm_u8s8_kernel(IgemmArgs p, int tiles_m, int tiles_n,
                                                                 int m_fastest) {
  constexpr int BM = WM * TM * 32, BN = WN * TN * 32, NT = WM * WN * 64;
  constexpr int A_PER = BM * 8 / NT;
  constexpr int B_CH = BN * 8;
  constexpr int B_PER = (B_CH + NT - 1) / NT;
  static_assert(BM * 8 % NT == 0, "A staging map");
  constexpr int SROW = BN + 4;  // epilogue tile row stride: odd dword count -> conflict-free ds_write_b32
  constexpr int MAIN_BYTES = (BM + BN) * BK2;
  constexpr int EPI_BYTES = BM * SROW;
  constexpr int SMEM = MAIN_BYTES > EPI_BYTES ? MAIN_BYTES : EPI_BYTES;
  __shared__ __attribute__((aligned(16))) uint8_t smem[SMEM];
  uint8_t* smA = smem;
  uint8_t* smB = smem + BM * BK2;

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

  // ---- accumulators: D = W_tile x A_tile^T, row = feature, col = activation row ---
  //   lane & 31 -> activation row; feature = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
  v16i acc[TM][TN];
#pragma unroll
  for (int ni = 0; ni < TN; ++ni) {
    const int nb = n0 + (wn * TN + ni) * 32 + 4 * (lane >> 5);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int gcol = nb + (r & 3) + 8 * (r >> 2);
      const int init = gcol < p.N ? p.ocp[gcol] : 0;
#pragma unroll
      for (int mi = 0; mi < TM; ++mi) acc[mi][ni][r] = init;
    }
  }

  // ---- A staging state: thread owns chunk column cA of rows (tid>>3) + 32*i ---------
  const int cA = tid & 7;
  const uint8_t* a_ptr[A_PER];
  int a_ih0[A_PER], a_iw0[A_PER];
#pragma unroll
  for (int i = 0; i < A_PER; ++i) {
    int gr = m0 + (tid >> 3) + (NT >> 3) * i;
    gr = gr < p.M ? gr : p.M - 1;  // rows past M are computed and discarded
    if (AMODE == 0) {
      a_ptr[i] = p.A + (size_t)gr * p.lda;
      a_ih0[i] = a_iw0[i] = 0;
    } else {
      const int P = p.OH * p.OW;
      const int img = gr / P, rem = gr - img * P;
      const int oh = rem / p.OW, ow = rem - oh * p.OW;
      a_ih0[i] = oh * p.sh - p.ph;
      a_iw0[i] = ow * p.sw - p.pw;
      a_ptr[i] = p.A + (((long)img * p.H + a_ih0[i]) * p.W + a_iw0[i]) * ((long)p.C16 * 16);
    }
  }
  // walk of this thread's K chunk q = cA, cA + 8, ... as (kh, kw, c16)
  int q = cA, kh = 0, kw = 0, c16 = 0;
  if (AMODE == 1) {
    c16 = cA % p.C16;
    const int tap = cA / p.C16;
    kw = tap % p.KW;
    kh = tap / p.KW;
  }
  const int8_t* b_ptr[B_PER];
#pragma unroll
  for (int i = 0; i < B_PER; ++i) {
    const int idx = tid + i * NT;
    b_ptr[i] = p.B + (size_t)(n0 + (idx >> 3)) * p.Kpad + (idx & 7) * 16;
  }

  v4i ra[A_PER], rb[B_PER];
  const v4i zfill = {(int)p.zp_fill, (int)p.zp_fill, (int)p.zp_fill, (int)p.zp_fill};

  auto load_tile = [&](int k0) {
    const bool kvalid = q < p.Kchunks;
    if (AMODE == 0) {
#pragma unroll
      for (int i = 0; i < A_PER; ++i)
        ra[i] = kvalid ? *reinterpret_cast<const v4i*>(a_ptr[i] + (size_t)q * 16) : zfill;
    } else {
      const long koff = ((long)kh * p.W + kw) * ((long)p.C16 * 16) + c16 * 16;
#pragma unroll
      for (int i = 0; i < A_PER; ++i) {
        const bool ok = kvalid && (unsigned)(a_ih0[i] + kh) < (unsigned)p.H && (unsigned)(a_iw0[i] + kw) < (unsigned)p.W;
        ra[i] = ok ? *reinterpret_cast<const v4i*>(a_ptr[i] + koff) : zfill;
      }
    }
#pragma unroll
    for (int i = 0; i < B_PER; ++i)
      if (B_CH % NT == 0 || tid + i * NT < B_CH) rb[i] = *reinterpret_cast<const v4i*>(b_ptr[i] + k0);
    // advance the chunk walk by one K tile (8 chunks)
    q += 8;
    if (AMODE == 1) {
      c16 += 8;
      while (c16 >= p.C16) {
        c16 -= p.C16;
        if (++kw == p.KW) {
          kw = 0;
          ++kh;
        }
      }
    }
  };
  auto store_tile = [&]() {
#pragma unroll
    for (int i = 0; i < A_PER; ++i) {
      const v4i v = ra[i] ^ (int)0x80808080;  // u8 -> s8 re-bias (the +128*wsum term is in ocp)
      *reinterpret_cast<v4i*>(smA + lds_off2((tid >> 3) + (NT >> 3) * i, cA)) = v;
    }
#pragma unroll
    for (int i = 0; i < B_PER; ++i) {
      const int idx = tid + i * NT;
      if (B_CH % NT == 0 || idx < B_CH) *reinterpret_cast<v4i*>(smB + lds_off2(idx >> 3, idx & 7)) = rb[i];
    }
  };

  const int nk = p.Kpad / BK2;
  load_tile(0);
  store_tile();
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    if (kt + 1 < nk) load_tile((kt + 1) * BK2);  // global loads in flight under the MFMAs
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const int chunk = ks * 2 + (lane >> 5);
      v4i af[TM], bf[TN];
#pragma unroll
      for (int mi = 0; mi < TM; ++mi)
        af[mi] = *reinterpret_cast<const v4i*>(smA + lds_off2((wm * TM + mi) * 32 + (lane & 31), chunk));
#pragma unroll
      for (int ni = 0; ni < TN; ++ni)
        bf[ni] = *reinterpret_cast<const v4i*>(smB + lds_off2((wn * TN + ni) * 32 + (lane & 31), chunk));
#pragma unroll
      for (int mi = 0; mi < TM; ++mi)
#pragma unroll
        for (int ni = 0; ni < TN; ++ni)
          acc[mi][ni] = __builtin_amdgcn_mfma_i32_32x32x32_i8(bf[ni], af[mi], acc[mi][ni], 0, 0, 0);
    }
    __syncthreads();
    if (kt + 1 < nk) {
      store_tile();
      __syncthreads();
    }
  }

  // ---- epilogue: (bias) -> requant -> (relu) -> LDS tile [BM][BN] -> 16-B row stores ----
  const Requant rq = p.rq;
  const int relu_lo = p.relu ? p.zp_out : 0;
#pragma unroll
  for (int mi = 0; mi < TM; ++mi) {
    const int lrow = (wm * TM + mi) * 32 + (lane & 31);
    const int grow = m0 + lrow;
#pragma unroll
    for (int ni = 0; ni < TN; ++ni) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int lcol0 = (wn * TN + ni) * 32 + 8 * g + 4 * (lane >> 5);
        const int gcol0 = n0 + lcol0;
        uint32_t packed = 0;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          int c = acc[mi][ni][g * 4 + r];
          const int gcol = gcol0 + r;
          if (p.acc != nullptr && grow < p.M && gcol < p.N) p.acc[(size_t)grow * p.N + gcol] = c;
          if (p.biasf != nullptr) {
            const float bfv = gcol < p.N ? p.biasf[gcol] : 0.0f;
            c = (int)((float)c + bfv);  // src/fully_connected.cc:44: int += float
          }
          int u = requant(c, rq);
          u = u > relu_lo ? u : relu_lo;
          packed |= (uint32_t)u << (8 * r);
        }
        *reinterpret_cast<uint32_t*>(smem + lrow * SROW + lcol0) = packed;
      }
    }
  }
  __syncthreads();
  if (p.vec_store) {
    constexpr int CPR = BN / 16;
    for (int idx = tid; idx < BM * CPR; idx += NT) {
      const int lrow = idx / CPR, ch = idx - lrow * CPR;
      const int grow = m0 + lrow, gcol = n0 + ch * 16;
      if (grow < p.M && gcol < p.N) {
        const uint32_t* s = reinterpret_cast<const uint32_t*>(smem + lrow * SROW + ch * 16);
        *reinterpret_cast<uint4*>(p.out + (size_t)grow * p.N + gcol) = make_uint4(s[0], s[1], s[2], s[3]);
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

// ---- max-pool over NHWC u8 (src/functional.cc:36-64 on the internal layout) ---------
// one thread = 16 channels of one output pixel; running max from 0, floor, no padding
__device__ __forceinline__ uint32_t bmax4(uint32_t a, uint32_t b) {
  // per-byte unsigned max: split even/odd bytes into 16-bit lanes
  const uint32_t ae = a & 0x00FF00FFu, ao = (a >> 8) & 0x00FF00FFu;
  const uint32_t be = b & 0x00FF00FFu, bo = (b >> 8) & 0x00FF00FFu;
  const uint32_t d = (ae | 0x01000100u) - be;                   // borrow-free per lane: bit 8 set iff ae >= be
  const uint32_t me = ((d >> 8) & 0x00010001u) * 0xFFu;          // 0xFF where ae >= be
  const uint32_t d2 = (ao | 0x01000100u) - bo;
  const uint32_t mo = ((d2 >> 8) & 0x00010001u) * 0xFFu;
  const uint32_t re = (ae & me) | (be & ~me & 0x00FF00FFu);
  const uint32_t ro = (ao & mo) | (bo & ~mo & 0x00FF00FFu);
  return re | (ro << 8);
}

__global__ __launch_bounds__(256) void maxpool_u8_nhwc_kernel(const uint8_t* __restrict__ in,
                                                              uint8_t* __restrict__ out, int64_t total, int h,
                                                              int w, int c16, int oh, int ow, int k, int s) {
  const int64_t gstride = (int64_t)gridDim.x * 256;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += gstride) {
    const int cc = (int)(e % c16);
    int64_t t = e / c16;
    const int x = (int)(t % ow);
    t /= ow;
    const int y = (int)(t % oh);
    const int64_t img = t / oh;
    const uint4* p = reinterpret_cast<const uint4*>(in) + ((img * h + (int64_t)y * s) * w + (int64_t)x * s) * c16 + cc;
    uint4 m = make_uint4(0, 0, 0, 0);
    for (int a = 0; a < k; ++a)
      for (int b = 0; b < k; ++b) {
        const uint4 v = p[((int64_t)a * w + b) * c16];
        m.x = bmax4(m.x, v.x);
        m.y = bmax4(m.y, v.y);
        m.z = bmax4(m.z, v.z);
        m.w = bmax4(m.w, v.w);
      }
    reinterpret_cast<uint4*>(out)[e] = m;
  }
}

// ---- layout conversion u8: NCHW <-> NHWC through a 32 x 32 LDS tile per (image) --------
// src [R][S] -> dst [S][R] per image (NCHW->NHWC: R = c, S = h*w; NHWC->NCHW: R = h*w, S = c)
__global__ __launch_bounds__(256) void transpose_u8_kernel(const uint8_t* __restrict__ in, uint8_t* __restrict__ out,
                                                           int R, int S) {
  __shared__ uint8_t tile[64][65];
  const int64_t img = blockIdx.z;
  const int r0 = blockIdx.y * 64, s0 = blockIdx.x * 64;
  const uint8_t* src = in + img * (int64_t)R * S;
  uint8_t* dst = out + img * (int64_t)R * S;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  for (int j = ty; j < 64; j += 4) {
    const int r = r0 + j, s = s0 + tx;
    tile[j][tx] = (r < R && s < S) ? src[(int64_t)r * S + s] : 0;
  }
  __syncthreads();
  for (int j = ty; j < 64; j += 4) {
    const int s = s0 + j, r = r0 + tx;
    if (r < R && s < S) dst[(int64_t)s * R + r] = tile[tx][j];
  }
}

// ---- small-C repack: NCHW u8 [n][c<=4][h][w] -> physically padded "grouped" NHWC -------
// out [n][Hp][Wg][16]: pixel (y, x) of the padded image (pad rows/cols hold the zero
// point; x = 4*g + px) stores its channels at bytes 4*px .. 4*px+3 of group g (channels
// >= c hold zp; their weights are zero).  Makes the stride-4 11x11x3 first layer an
// ordinary C=16, 11x3-tap implicit GEMM with 16-byte-aligned, predicate-free gathers.
__global__ __launch_bounds__(256) void repack_smallc_kernel(const uint8_t* __restrict__ in, uint8_t* __restrict__ out,
                                                            int64_t total, int c, int h, int w, int Hp, int Wg,
                                                            int ph, int pw, uint32_t zp) {
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
      wds[px] = v;
    }
    reinterpret_cast<uint4*>(out)[e] = make_uint4(wds[0], wds[1], wds[2], wds[3]);
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

inline int cap_grid(int64_t items, int threads, int max_blocks = 256 * 16) {
  int64_t b = (items + threads - 1) / threads;
  if (b < 1) b = 1;
  return (int)(b > max_blocks ? max_blocks : b);
}

template <int AMODE, int WM, int WN, int TM, int TN>
int launch_cfg(i8ie_ctx* ctx, const IgemmArgs& a, const char* name, double ops, double bytes) {
  constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
  const int tiles_m = (a.M + BM - 1) / BM, tiles_n = (a.N + BN - 1) / BN;
  const int m_fastest = ((size_t)a.N > (size_t)a.M) ? 1 : 0;
  I8ieProfScope prof(ctx, name, ops, bytes);
  igemm_u8s8_kernel<AMODE, WM, WN, TM, TN><<<tiles_m * tiles_n, WM * WN * 64, 0, ctx->stream>>>(a, tiles_m, tiles_n,
                                                                                              m_fastest);
  I8IE_LAUNCH_CHECK();
  return I8IE_OK;
}

template <int AMODE>
int launch_any(i8ie_ctx* ctx, const IgemmArgs& a, double ops, double bytes) {
  if (a.N <= 32) return launch_cfg<AMODE, 4, 1, 1, 1>(ctx, a, AMODE ? "igemm_conv_128x32" : "igemm_lin_128x32", ops, bytes);
  if (a.N <= 64) return launch_cfg<AMODE, 2, 2, 2, 1>(ctx, a, AMODE ? "igemm_conv_128x64" : "igemm_lin_128x64", ops, bytes);
  if (a.N <= 96) return launch_cfg<AMODE, 4, 1, 1, 3>(ctx, a, AMODE ? "igemm_conv_128x96" : "igemm_lin_128x96", ops, bytes);
  return launch_cfg<AMODE, 2, 2, 2, 2>(ctx, a, AMODE ? "igemm_conv_128x128" : "igemm_lin_128x128", ops, bytes);
}

}  // namespace

// ---- entry points used by i8ie_layer.hip ---------------------------------------------------
struct I8ieIgemmCall {
  const uint8_t* A;
  int amode;
  long lda;
  int M, Kchunks;
  int H, W, C, KH, KW, sh, sw, ph, pw, OH, OW;
  int zp_in;
  const int8_t* B;
  int Kpad, N;
  const int32_t* ocp;
  const float* biasf;
  float s_in, s_w, s_out;
  int zp_out, relu;
  uint8_t* out;
  int32_t* acc;
  double Ktrue;
};

int i8ie_igemm_launch(i8ie_ctx* ctx, const I8ieIgemmCall& c) {
  I8IE_REQUIRE(c.M > 0 && c.N > 0 && c.Kpad > 0 && c.Kpad % BK2 == 0, "igemm dimensions");
  I8IE_REQUIRE((reinterpret_cast<uintptr_t>(c.A) & 15u) == 0 && (reinterpret_cast<uintptr_t>(c.B) & 15u) == 0,
               "operands must be 16-byte aligned");
  IgemmArgs a{};
  a.A = c.A;
  a.lda = c.lda;
  a.M = c.M;
  a.Kchunks = c.Kchunks;
  a.H = c.H; a.W = c.W; a.C16 = c.C / 16;
  a.KH = c.KH; a.KW = c.KW; a.sh = c.sh; a.sw = c.sw; a.ph = c.ph; a.pw = c.pw; a.OH = c.OH; a.OW = c.OW;
  a.zp_fill = (uint32_t)(c.zp_in & 0xFF) * 0x01010101u;
  a.B = c.B;
  a.Kpad = c.Kpad;
  a.N = c.N;
  a.ocp = c.ocp;
  a.biasf = c.biasf;
  a.rq.sa = c.s_in; a.rq.sb = c.s_w; a.rq.sc = c.s_out; a.rq.zpf = (float)c.zp_out;
  const double ms = (double)c.s_in * (double)c.s_w / (double)c.s_out;
  a.rq.ms = (float)ms;
  // fast path only for ordinary positive finite scales; anything else takes the exact sequence
  a.rq.fast = (c.s_in > 0 && c.s_w > 0 && c.s_out > 0 && ms > 1e-30 && ms < 1e30 && c.s_in < 1e30f && c.s_w < 1e30f &&
               c.s_out < 1e30f && c.s_in > 1e-30f && c.s_w > 1e-30f && c.s_out > 1e-30f)
                  ? 1 : 0;
  a.zp_out = c.zp_out;
  a.relu = c.relu;
  a.vec_store = ((c.N & 15) == 0 && (reinterpret_cast<uintptr_t>(c.out) & 15u) == 0) ? 1 : 0;
  a.out = c.out;
  a.acc = c.acc;
  const double ops = 2.0 * c.M * c.N * c.Ktrue;
  const double bytes = (double)c.M * c.Ktrue + (double)c.N * c.Ktrue + (double)c.M * c.N;
  if (c.amode == 0) return launch_any<0>(ctx, a, ops, bytes);
  I8IE_REQUIRE(c.C % 16 == 0 && c.C > 0, "NHWC gather needs channels % 16 == 0");
  return launch_any<1>(ctx, a, ops, bytes);
}

int i8ie_launch_finish_offsets(i8ie_ctx* ctx, const int32_t* oc, const int32_t* wsum, const int8_t* qb, float s_in,
                               int n, int32_t* ocp, float* biasf) {
  finish_offsets_kernel<<<(n + 63) / 64, 64, 0, ctx->stream>>>(oc, wsum, qb, s_in, n, ocp, biasf);
  I8IE_LAUNCH_CHECK();
  return I8IE_OK;
}

int i8ie_launch_transpose_u8(i8ie_ctx* ctx, const uint8_t* in, uint8_t* out, int n, int R, int S) {
  I8ieProfScope prof(ctx, "layout_transpose_u8", 0.0, 2.0 * n * (double)R * S);
  dim3 grid((S + 63) / 64, (R + 63) / 64, n);
  transpose_u8_kernel<<<grid, 256, 0, ctx->stream>>>(in, out, R, S);
  I8IE_LAUNCH_CHECK();
  return I8IE_OK;
}

int i8ie_launch_repack_smallc(i8ie_ctx* ctx, const uint8_t* in, uint8_t* out, int n, int c, int h, int w, int Hp,
                              int Wg, int ph, int pw, int zp) {
  const int64_t total = (int64_t)n * Hp * Wg;
  I8ieProfScope prof(ctx, "repack_smallc_u8", 0.0, (double)n * c * h * w + 16.0 * total);
  repack_smallc_kernel<<<cap_grid(total, 256), 256, 0, ctx->stream>>>(in, out, total, c, h, w, Hp, Wg, ph, pw,
                                                                      (uint32_t)(zp & 0xFF));
  I8IE_LAUNCH_CHECK();
  return I8IE_OK;
}

int i8ie_launch_maxpool_nhwc(i8ie_ctx* ctx, const uint8_t* in, uint8_t* out, int n, int c, int h, int w, int k, int s) {
  const int oh = (h - k) / s + 1, ow = (w - k) / s + 1;
  const int64_t total = (int64_t)n * oh * ow * (c / 16);
  I8ieProfScope prof(ctx, "maxpool_u8_nhwc", 0.0, (double)n * c * h * w + 16.0 * total);
  maxpool_u8_nhwc_kernel<<<cap_grid(total, 256, 256 * 32), 256, 0, ctx->stream>>>(in, out, total, h, w, c / 16, oh, ow,
                                                                                 k, s);
  I8IE_LAUNCH_CHECK();
  return I8IE_OK;
}
