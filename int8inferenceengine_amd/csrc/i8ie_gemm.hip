// i8ie_gemm.hip -- the dense u8 x s8 -> s32 contraction of the hot path on the
// gfx950 matrix cores, with the reference's epilogue fused:
//     C[r][j] = sum_k A_u8[r][k] * W_s8[j][k] + oc[j]          (exact int32)
//     (Linear only)  C = (int)((float)C + (float)q_b[j] / s_in)
//     out = down_scale(C)  ->  u8, stored row-major or NCHW
// replacing cblas_gemm_s8u8s32 + the bias loop + down_scale + transpose of
// src/fully_connected.cc:39-48 and src/conv2d.cc:131-136.
//
// MFMA has no unsigned-int8 operand, so activations are re-biased on the way
// into LDS: a' = a ^ 0x80 (= a - 128 as s8) and the exact correction
// 128 * sum_k W[j][k] joins oc[j] in the accumulator's initial value.  The
// accumulator therefore holds the reference's C bit for bit when the K loop ends.
//
// Tiling: block tile BM x BN (128x128 default), BK = 64 bytes of K per step,
// 4 waves, each owning TM x TN tiles of v_mfma_i32_32x32x32_i8.  Both operands
// are K-contiguous rows, staged through registers (next K-tile's global loads
// are issued before the current tile's MFMAs) into a 16-B-chunk XOR-swizzled
// LDS image that makes the ds_read_b128 fragment reads conflict-free.
#include "i8ie_internal.h"
#include "i8ie_requant.h"

namespace {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

constexpr int BK = 64;  // bytes of K per LDS tile = 4 chunks of 16 B = 2 MFMA k-steps

// byte offset of 16-B chunk `c` (0..3) of tile row `row` in the swizzled LDS image
__device__ __forceinline__ int lds_chunk_off(int row, int c) {
  return row * BK + ((c ^ ((row >> 2) & 3)) << 4);
}

__device__ __forceinline__ uint8_t requant_u8(int c, float sa, float sb, float sc, float zpf) {
  const I8ieRequant q{sa, sb, sc, zpf, 0.0f, I8IE_RQ_EXACT};  // i8ie_requant.h: the exact sequence only
  return (uint8_t)i8ie_requant_exact((float)c, q, 0);
}

template <int WM, int WN, int TM, int TN>
__global__ __launch_bounds__(WM* WN * 64) void gemm_u8s8_kernel(I8ieGemmArgs p, int tiles_m,
                                                                int tiles_n, int m_fastest) {
  constexpr int BM = WM * TM * 32, BN = WN * TN * 32, NT = WM * WN * 64;
  constexpr int A_CH = BM * 4, B_CH = BN * 4;
  constexpr int A_PER = (A_CH + NT - 1) / NT, B_PER = (B_CH + NT - 1) / NT;
  constexpr int SROW_T = BM + 4;   // epilogue tile, NCHW mode: T[col][row]
  constexpr int SROW_R = BN + 16;  // epilogue tile, row-major mode: T[row][col]
  constexpr int MAIN_BYTES = (BM + BN) * BK;
  constexpr int EPI_T = BN * SROW_T, EPI_R = BM * SROW_R;
  constexpr int EPI_BYTES = EPI_T > EPI_R ? EPI_T : EPI_R;
  constexpr int SMEM = MAIN_BYTES > EPI_BYTES ? MAIN_BYTES : EPI_BYTES;
  __shared__ __attribute__((aligned(16))) uint8_t smem[SMEM];
  uint8_t* smA = smem;
  uint8_t* smB = smem + BM * BK;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;

  // XCD-aware block -> tile map: blocks with equal blockIdx % 8 share an XCD (and
  // its L2); give each XCD a contiguous run of tiles so that the blocks that
  // re-read one operand panel hit the same L2.  Bijective for any grid size.
  const int nwg = gridDim.x;
  const int bid = blockIdx.x;
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

  // ---- accumulators start at oc[j] + 128 * wsum[j] ---------------------------
  v16i acc[TM][TN];
#pragma unroll
  for (int ni = 0; ni < TN; ++ni) {
    const int gcol = n0 + (wn * TN + ni) * 32 + (lane & 31);
    int init = 0;
    if (gcol < p.N) init = p.oc[gcol] + 128 * p.wsum[gcol];
#pragma unroll
    for (int mi = 0; mi < TM; ++mi)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mi][ni][r] = init;
  }

  // ---- staging registers -----------------------------------------------------
  v4i ra[A_PER], rb[B_PER];
  const uint8_t* a_ptr[A_PER];
  const int8_t* b_ptr[B_PER];
  int a_koff[A_PER];
#pragma unroll
  for (int i = 0; i < A_PER; ++i) {
    const int idx = tid + i * NT;
    const int row = idx >> 2, c = idx & 3;
    int gr = m0 + row;
    gr = gr < p.M ? gr : p.M - 1;  // clamp: rows past M are computed but never stored
    a_ptr[i] = p.A + (size_t)gr * p.lda + c * 16;
    a_koff[i] = c * 16;
  }
#pragma unroll
  for (int i = 0; i < B_PER; ++i) {
    const int idx = tid + i * NT;
    const int row = idx >> 2, c = idx & 3;
    b_ptr[i] = p.B + (size_t)(n0 + row) * p.Kpad + c * 16;  // Npad rows exist: no clamp
  }

  auto load_tile = [&](int k0) {
#pragma unroll
    for (int i = 0; i < A_PER; ++i) {
      if (A_CH % NT == 0 || tid + i * NT < A_CH) {
        v4i z = {0, 0, 0, 0};
        ra[i] = (k0 + a_koff[i] < p.Ka) ? *reinterpret_cast<const v4i*>(a_ptr[i] + k0) : z;
      }
    }
#pragma unroll
    for (int i = 0; i < B_PER; ++i) {
      if (B_CH % NT == 0 || tid + i * NT < B_CH)
        rb[i] = *reinterpret_cast<const v4i*>(b_ptr[i] + k0);
    }
  };
  auto store_tile = [&]() {
#pragma unroll
    for (int i = 0; i < A_PER; ++i) {
      const int idx = tid + i * NT;
      if (A_CH % NT == 0 || idx < A_CH) {
        v4i v = ra[i] ^ (int)0x80808080;  // u8 -> s8 re-bias
        *reinterpret_cast<v4i*>(smA + lds_chunk_off(idx >> 2, idx & 3)) = v;
      }
    }
#pragma unroll
    for (int i = 0; i < B_PER; ++i) {
      const int idx = tid + i * NT;
      if (B_CH % NT == 0 || idx < B_CH)
        *reinterpret_cast<v4i*>(smB + lds_chunk_off(idx >> 2, idx & 3)) = rb[i];
    }
  };

  const int nk = p.Kpad / BK;
  load_tile(0);
  store_tile();
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    if (kt + 1 < nk) load_tile((kt + 1) * BK);  // in flight during the MFMAs below
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int chunk = ks * 2 + (lane >> 5);
      v4i af[TM], bf[TN];
#pragma unroll
      for (int mi = 0; mi < TM; ++mi)
        af[mi] = *reinterpret_cast<const v4i*>(smA + lds_chunk_off((wm * TM + mi) * 32 + (lane & 31), chunk));
#pragma unroll
      for (int ni = 0; ni < TN; ++ni)
        bf[ni] = *reinterpret_cast<const v4i*>(smB + lds_chunk_off((wn * TN + ni) * 32 + (lane & 31), chunk));
#pragma unroll
      for (int mi = 0; mi < TM; ++mi)
#pragma unroll
        for (int ni = 0; ni < TN; ++ni)
          acc[mi][ni] = __builtin_amdgcn_mfma_i32_32x32x32_i8(af[mi], bf[ni], acc[mi][ni], 0, 0, 0);
    }
    __syncthreads();  // everyone is done reading this tile
    if (kt + 1 < nk) {
      store_tile();
      __syncthreads();
    }
  }

  // ---- epilogue ----------------------------------------------------------------
  // C/D map of the 32x32 MFMA: column = lane & 31 (output feature j),
  // row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5) (activation row).
  const float sa = p.s_in, sb = p.s_w, sc = p.s_out, zpf = (float)p.zp_out;
  const bool nchw = p.out_mode == I8IE_OUT_NCHW;
#pragma unroll
  for (int ni = 0; ni < TN; ++ni) {
    const int lcol = (wn * TN + ni) * 32 + (lane & 31);
    const int gcol = n0 + lcol;
    float bias_f = 0.0f;
    if (p.qb != nullptr && gcol < p.N) bias_f = (float)p.qb[gcol] / sa;  // src/fully_connected.cc:44
#pragma unroll
    for (int mi = 0; mi < TM; ++mi) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int lrow0 = (wm * TM + mi) * 32 + 8 * g + 4 * (lane >> 5);
        uint32_t packed = 0;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          int c = acc[mi][ni][g * 4 + r];
          const int grow = m0 + lrow0 + r;
          if (p.acc != nullptr && grow < p.M && gcol < p.N) p.acc[(size_t)grow * p.N + gcol] = c;
          if (p.qb != nullptr) c = (int)((float)c + bias_f);
          const uint32_t u = requant_u8(c, sa, sb, sc, zpf);
          packed |= u << (8 * r);
          if (!nchw) smem[(lrow0 + r) * SROW_R + lcol] = (uint8_t)u;
        }
        if (nchw) *reinterpret_cast<uint32_t*>(smem + lcol * SROW_T + lrow0) = packed;
      }
    }
  }
  __syncthreads();

  if (nchw) {
    // T[col][row]; a thread keeps one row (pixel) and walks the columns, so its
    // (image, pixel) split is computed once and lanes write consecutive pixels.
    static_assert(NT % BM == 0, "thread -> row map");
    const int lrow = tid % BM;
    const int grow = m0 + lrow;
    if (grow < p.M) {
      const int img = grow / p.P, pix = grow - img * p.P;
      uint8_t* obase = p.out + (size_t)img * p.N * p.P + pix;
      for (int lcol = tid / BM; lcol < BN; lcol += NT / BM) {
        const int gcol = n0 + lcol;
        if (gcol < p.N) obase[(size_t)gcol * p.P] = smem[lcol * SROW_T + lrow];
      }
    }
  } else {
    const bool vec_ok = (p.N % 16 == 0) && ((reinterpret_cast<uintptr_t>(p.out) & 15u) == 0);
    if (vec_ok) {
      constexpr int CPR = BN / 16;  // 16-B chunks per tile row
      for (int idx = tid; idx < BM * CPR; idx += NT) {
        const int lrow = idx / CPR, ch = idx % CPR;
        const int grow = m0 + lrow, gcol = n0 + ch * 16;
        if (grow < p.M && gcol < p.N)
          *reinterpret_cast<uint4*>(p.out + (size_t)grow * p.N + gcol) =
              *reinterpret_cast<const uint4*>(smem + lrow * SROW_R + ch * 16);
      }
    } else {
      for (int idx = tid; idx < BM * BN; idx += NT) {
        const int lrow = idx / BN, lcol = idx % BN;
        const int grow = m0 + lrow, gcol = n0 + lcol;
        if (grow < p.M && gcol < p.N) p.out[(size_t)grow * p.N + gcol] = smem[lrow * SROW_R + lcol];
      }
    }
  }
}

// ---- a2': im2col of src/conv2d.cc:5-49, u8, NCHW input, pad value = zp_in ------
// Row r = (img, ti, tj), column k = ch*kh*kw + l*kw + m; one thread builds one
// 16-byte chunk of one row.  Columns K..Kpad-1 are zero (B is zero there too).
__global__ __launch_bounds__(256) void im2col_u8_nchw_kernel(const uint8_t* __restrict__ in,
                                                             uint8_t* __restrict__ col, int64_t rows,
                                                             int c, int h, int w, int kh, int kw,
                                                             int oh, int ow, int stride, int pad,
                                                             int K, int Kpad, uint32_t zp) {
  const int cpr = Kpad >> 4;
  const int64_t total = rows * cpr;
  const int64_t gstride = (int64_t)gridDim.x * blockDim.x;
  const int khw = kh * kw;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += gstride) {
    const int ch16 = (int)(e % cpr);
    const int64_t r = e / cpr;
    const int tj = (int)(r % ow);
    const int64_t t2 = r / ow;
    const int ti = (int)(t2 % oh);
    const int64_t img = t2 / oh;
    const int y0 = ti * stride - pad, x0 = tj * stride - pad;
    const uint8_t* base = in + img * c * h * w;
    int k = ch16 << 4;
    int chn = k / khw;
    int rem = k - chn * khw;
    int l = rem / kw, m = rem - l * kw;
    uint32_t wds[4] = {0, 0, 0, 0};
#pragma unroll
    for (int b = 0; b < 16; ++b, ++k) {
      uint32_t v = 0;
      if (k < K) {
        const int y = y0 + l, x = x0 + m;
        v = (y < 0 || x < 0 || y >= h || x >= w) ? zp : (uint32_t)base[((size_t)chn * h + y) * w + x];
        if (++m == kw) {
          m = 0;
          if (++l == kh) {
            l = 0;
            ++chn;
          }
        }
      }
      wds[b >> 2] |= v << (8 * (b & 3));
    }
    *reinterpret_cast<uint4*>(col + r * Kpad + ((size_t)ch16 << 4)) = make_uint4(wds[0], wds[1], wds[2], wds[3]);
  }
}

// copy [rows][k] bytes into a zero/`fill`-padded [rows_pad][k_pad] image (16-B chunks)
__global__ __launch_bounds__(256) void pad_rows_kernel(const uint8_t* __restrict__ src, int rows, int k,
                                                       uint8_t* __restrict__ dst, int rows_pad,
                                                       int k_pad, uint32_t fill) {
  const int cpr = k_pad >> 4;
  const int64_t total = (int64_t)rows_pad * cpr;
  const int64_t gstride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += gstride) {
    const int ch16 = (int)(e % cpr);
    const int64_t r = e / cpr;
    uint32_t wds[4] = {0, 0, 0, 0};
#pragma unroll
    for (int b = 0; b < 16; ++b) {
      const int kk = (ch16 << 4) + b;
      uint32_t v = (r < rows && kk < k) ? (uint32_t)src[r * k + kk] : fill;
      wds[b >> 2] |= v << (8 * (b & 3));
    }
    *reinterpret_cast<uint4*>(dst + r * k_pad + ((size_t)ch16 << 4)) = make_uint4(wds[0], wds[1], wds[2], wds[3]);
  }
}

// ---- a4: zero-point offset vectors ------------------------------------------
// One thread per output feature replays the reference's sequential fp32
// accumulation (k ascending), so the result is bit-identical even where prefix
// sums leave the exactly-representable range.  Also emits the exact integer row
// sum used for the u8 -> s8 re-biasing.  Runs once per (layer, s_in, zp_in).
template <bool CONV>
__global__ __launch_bounds__(64) void offsets_kernel(const int8_t* __restrict__ qw,
                                                     const int8_t* __restrict__ qb, int n, int K,
                                                     float s_in, int zp_in, int32_t* __restrict__ oc,
                                                     int32_t* __restrict__ wsum) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  const int8_t* row = qw + (size_t)j * K;
  float t = 0.0f;
  int s = 0;
  for (int k = 0; k < K; ++k) {
    const int q = row[k];
    t = t + (float)(zp_in * q);  // src/conv2d.cc:121, src/fully_connected.cc:35
    s += q;
  }
  if (oc != nullptr) {
    if (CONV)
      oc[j] = (int)((float)qb[j] / s_in - t);  // src/conv2d.cc:123
    else
      oc[j] = (int)(-t);  // src/fully_connected.cc:37
  }
  if (wsum != nullptr) wsum[j] = s;
}

inline int cap_grid(int64_t items, int threads) {
  int64_t b = (items + threads - 1) / threads;
  if (b < 1) b = 1;
  return (int)(b > 256 * 16 ? 256 * 16 : b);
}

template <int WM, int WN, int TM, int TN>
int launch_cfg(i8ie_ctx* ctx, const I8ieGemmArgs& a) {
  constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
  const int tiles_m = (a.M + BM - 1) / BM, tiles_n = (a.N + BN - 1) / BN;
  // walk the tile grid along the dimension whose operand panel is the larger one
  // to re-read: weights (N x K) vs activations (M x K)
  const int m_fastest = ((size_t)a.N > (size_t)a.M) ? 1 : 0;
  static const char* kName = BN == 128 ? "gemm_u8s8_128x128" : BN == 96 ? "gemm_u8s8_128x96"
                             : BN == 64 ? "gemm_u8s8_128x64" : "gemm_u8s8_128x32";
  const double kt = a.Ktrue > 0 ? a.Ktrue : a.Kpad;
  I8ieProfScope prof(ctx, kName, 2.0 * a.M * a.N * kt, (double)a.M * kt + (double)a.N * kt + (double)a.M * a.N);
  gemm_u8s8_kernel<WM, WN, TM, TN>
      <<<tiles_m * tiles_n, WM * WN * 64, 0, ctx->stream>>>(a, tiles_m, tiles_n, m_fastest);
  I8IE_LAUNCH_CHECK();
  return I8IE_OK;
}

}  // namespace

int i8ie_gemm_launch(i8ie_ctx* ctx, const I8ieGemmArgs& a) {
  I8IE_REQUIRE(a.M > 0 && a.N > 0 && a.Kpad > 0, "empty GEMM");
  I8IE_REQUIRE(a.Kpad % BK == 0 && a.Ka % 16 == 0 && a.lda % 16 == 0, "K layout");
  I8IE_REQUIRE((reinterpret_cast<uintptr_t>(a.A) & 15u) == 0 && (reinterpret_cast<uintptr_t>(a.B) & 15u) == 0,
               "operands must be 16-byte aligned");
  I8IE_REQUIRE((long long)a.M * a.N < (1LL << 40), "output too large");
  if (a.N <= 32) return launch_cfg<4, 1, 1, 1>(ctx, a);
  if (a.N <= 64) return launch_cfg<2, 2, 2, 1>(ctx, a);
  if (a.N <= 96) return launch_cfg<4, 1, 1, 3>(ctx, a);
  return launch_cfg<2, 2, 2, 2>(ctx, a);
}

// ---- helpers shared with i8ie_layer.hip ----------------------------------------
int i8ie_launch_pad_rows(i8ie_ctx* ctx, const void* src, int rows, int k, void* dst, int rows_pad,
                         int k_pad, int fill) {
  const int64_t items = (int64_t)rows_pad * (k_pad >> 4);
  I8ieProfScope prof(ctx, "pad_rows", 0.0, (double)rows * k + (double)rows_pad * k_pad);
  pad_rows_kernel<<<cap_grid(items, 256), 256, 0, ctx->stream>>>((const uint8_t*)src, rows, k, (uint8_t*)dst,
                                                                rows_pad, k_pad, (uint32_t)(fill & 0xFF));
  I8IE_LAUNCH_CHECK();
  return I8IE_OK;
}

int i8ie_launch_offsets(i8ie_ctx* ctx, bool conv, const int8_t* qw, const int8_t* qb, int n, int K,
                        float s_in, int zp_in, int32_t* oc, int32_t* wsum) {
  const int grid = (n + 63) / 64;
  I8ieProfScope prof(ctx, "offsets", 0.0, (double)n * K);
  if (conv)
    offsets_kernel<true><<<grid, 64, 0, ctx->stream>>>(qw, qb, n, K, s_in, zp_in, oc, wsum);
  else
    offsets_kernel<false><<<grid, 64, 0, ctx->stream>>>(qw, qb, n, K, s_in, zp_in, oc, wsum);
  I8IE_LAUNCH_CHECK();
  return I8IE_OK;
}

int i8ie_launch_im2col(i8ie_ctx* ctx, const uint8_t* in, uint8_t* col, int n, int c, int h, int w, int kh,
                       int kw, int oh, int ow, int stride, int pad, int K, int Kpad, int zp) {
  const int64_t rows = (int64_t)n * oh * ow;
  I8ieProfScope prof(ctx, "im2col_u8_nchw", 0.0, (double)n * c * h * w + (double)rows * Kpad);
  im2col_u8_nchw_kernel<<<cap_grid(rows * (Kpad >> 4), 256), 256, 0, ctx->stream>>>(
      in, col, rows, c, h, w, kh, kw, oh, ow, stride, pad, K, Kpad, (uint32_t)zp);
  I8IE_LAUNCH_CHECK();
  return I8IE_OK;
}
