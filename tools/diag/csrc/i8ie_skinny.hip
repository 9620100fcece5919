// i8ie_skinny.hip -- Linear::forward_prop(Tensor<u8_t>&&) (src/fully_connected.cc:22-52) for FEW input rows
// (m <= 128: the per-GPU shard of a batch of 1000 split over 8 GPUs, small serving batches).
//
// At such m the contraction is a stream over the weights (fc6: 37.7 MB) with little arithmetic per byte; the
// tiled kernel of i8ie_igemm.hip re-stages the activations from L2 for every 32-64 features and walks a
// latency chain of dependent K tiles.  Here instead:
//   * grid = (feature groups of 128) x (K slices) x (row blocks of 128); a block keeps ITS K slice of the
//     activations in LDS for its whole life (one LDS-DMA fill, 16-byte chunk c of row r at chunk c ^ (r & 7):
//     conflict-free ds_read_b128 fragments),
//   * each wave owns 32 features and issues ALL weight-fragment loads of the slice up front, from a panel stored
//     in MFMA fragment order ([feature tile][k-step][lane][16 B]: 1 KiB coalesced per load, read once from HBM),
//     into registers (one block per CU, so the register file is there to be used),
//   * then an unrolled chain of v_mfma_i32_32x32x32_i8 with no barrier and no further global traffic,
//   * INT32 partial slabs [slice][m][n] (slice 0 carries oc' = oc + 128 * wsum), finished by the split-K
//     reduction of i8ie_igemm.hip (bias in float, down_scale, ReLU: src/fully_connected.cc:42-48).
// Integer sums in any order are exact, so results equal the tiled kernel's and the reference's bit for bit.
#include <type_traits>

#include "i8ie_internal.h"

namespace {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

extern __shared__ __attribute__((aligned(16))) uint8_t skinny_smem[];

template <int B, int E, typename F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (B < E) {
    f(std::integral_constant<int, B>{});
    static_for<B + 1, E>(f);
  }
}

struct SkinnyArgs {
  const uint8_t* A;   // [M][lda] u8
  unsigned a_bytes;   // M * lda
  unsigned lda;
  int M;
  const v4i* Bf;      // fragment-ordered panel: [(tile * ksteps + kstep) * 64 + lane]
  int ksteps;         // Kpad / 32
  int Npad, N;
  const int32_t* ocp;
  int32_t* partial;   // [slices][M][N]
};

// Bpack [Npad][Kpad] (K contiguous) -> fragment order.  Lane l of the fragment for (tile t, k-step q) holds the
// 16 bytes of feature 32 t + (l & 31), K bytes 32 q + 16 (l >> 5) ...: the weights-as-row-operand form.
__global__ __launch_bounds__(256) void frag_pack_kernel(const int8_t* __restrict__ B, v4i* __restrict__ Bf,
                                                        int64_t total, int Kpad, int ksteps) {
  const int64_t gstride = (int64_t)gridDim.x * 256;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += gstride) {
    const int l = (int)(e & 63);
    const int64_t tq = e >> 6;
    const int q = (int)(tq % ksteps);
    const int64_t t = tq / ksteps;
    Bf[e] = *reinterpret_cast<const v4i*>(B + (t * 32 + (l & 31)) * Kpad + q * 32 + (l >> 5) * 16);
  }
}

template <int NSTEP>
__global__ __launch_bounds__(256) void linear_skinny_kernel(SkinnyArgs p) {
  constexpr int SLB = NSTEP * 32;  // bytes of K per slice = LDS row pitch
  constexpr int CPR = SLB / 16;    // 16-byte chunks per row (a multiple of 8)
  static_assert(NSTEP % 4 == 0, "whole 128-byte K blocks");
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int hh = lane >> 5, l7 = lane & 7;
  const int tile = blockIdx.x * 4 + wave;  // this wave's 32 features
  const int ks0 = blockIdx.y * NSTEP;      // first k-step of this block's slice
  const int m0 = blockIdx.z * 128;

  // ---- 1. every weight fragment of the slice, in flight at once -------------------------------------
  const bool live = tile * 32 < p.Npad;
  v4i bq[NSTEP];
  static_for<0, NSTEP>([&](auto jc) {
    constexpr int j = decltype(jc)::value;
    const v4i z = {0, 0, 0, 0};
    bq[j] = (live && ks0 + j < p.ksteps) ? p.Bf[((size_t)tile * p.ksteps + ks0 + j) * 64 + lane] : z;
  });

  // ---- 2. the activations' K slice -> LDS (DMA; rows past M repeat the last row, never stored) --------
  {
    const __amdgpu_buffer_rsrc_t rsA =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(p.A), 0, p.a_bytes, 0x00020000);
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    for (int base = wave_u * 64; base < 128 * CPR; base += 256) {
      const int idx = base + lane;
      const int row = idx / CPR, cs = idx - row * CPR;
      const int c = cs ^ (row & 7);
      int gr = m0 + row;
      gr = gr < p.M ? gr : p.M - 1;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (__attribute__((address_space(3))) void*)(skinny_smem + base * 16),
                                               16, (int)((unsigned)gr * p.lda + (unsigned)ks0 * 32u + (unsigned)c * 16u),
                                               0, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __syncthreads();

  // ---- 3. the contraction: 4 row tiles x NSTEP k-steps, weights from registers, activations from LDS -----
  v16i acc[4];
#pragma unroll
  for (int m = 0; m < 4; ++m)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[m][r] = 0;
  if (live) {
  int sw[4];  // swizzled chunk offset of k-step jj inside a 128-byte K block, for this lane's row (r & 7 == l7)
#pragma unroll
  for (int jj = 0; jj < 4; ++jj) sw[jj] = ((jj * 2 + hh) ^ l7) * 16;
  const uint8_t* arow = skinny_smem + (lane & 31) * SLB;
  static_for<0, NSTEP>([&](auto jc) {
    constexpr int j = decltype(jc)::value;
    constexpr int kb = j / 4, jj = j % 4;
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      const v4i af = *reinterpret_cast<const v4i*>(arow + m * 32 * SLB + kb * 128 + sw[jj]) ^ (int)0x80808080;
      acc[m] = __builtin_amdgcn_mfma_i32_32x32x32_i8(bq[j], af, acc[m], 0, 0, 0);
    }
  });
  }

  // ---- 4. partial slab of this slice (slice 0 carries oc'): through an LDS tile [128][128 + 4] so that the
  //         global writes are 512-byte row segments (lane-per-row stores of single dwords were ~2 M scattered
  //         line requests per launch and cost more than the contraction)
  constexpr int CT = 132;  // tile row pitch in ints: 16-byte aligned rows, conflict-free 16-byte column writes
  __syncthreads();          // every wave is done with the activations in LDS
  int* ct = reinterpret_cast<int*>(skinny_smem);
  if (live) {
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      const int row = m * 32 + (lane & 31);
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int col = wave * 32 + 8 * g + 4 * hh;
        const int gcol = blockIdx.x * 128 + col;
        int4 o = make_int4(0, 0, 0, 0);
        if (blockIdx.y == 0 && gcol < p.N) o = *reinterpret_cast<const int4*>(p.ocp + gcol);  // ocp is padded to Npad
        *reinterpret_cast<int4*>(ct + row * CT + col) =
            make_int4(acc[m][g * 4 + 0] + o.x, acc[m][g * 4 + 1] + o.y, acc[m][g * 4 + 2] + o.z, acc[m][g * 4 + 3] + o.w);
      }
    }
  }
  __syncthreads();
  int32_t* slab = p.partial + (size_t)blockIdx.y * p.M * p.N;
  const int n0 = blockIdx.x * 128;
  if ((p.N & 3) == 0) {
    for (int idx = tid; idx < 128 * 32; idx += 256) {
      const int row = idx >> 5, ch = idx & 31;
      const int grow = m0 + row, gcol = n0 + ch * 4;
      if (grow < p.M && gcol < p.N)
        *reinterpret_cast<int4*>(slab + (size_t)grow * p.N + gcol) = *reinterpret_cast<const int4*>(ct + row * CT + ch * 4);
    }
  } else {
    for (int idx = tid; idx < 128 * 128; idx += 256) {
      const int row = idx >> 7, col = idx & 127;
      const int grow = m0 + row, gcol = n0 + col;
      if (grow < p.M && gcol < p.N) slab[(size_t)grow * p.N + gcol] = ct[row * CT + col];
    }
  }
}

template <int NSTEP>
int launch_skinny(i8ie_ctx* ctx, const SkinnyArgs& a, dim3 grid) {
  constexpr int lds = 128 * NSTEP * 32 > 128 * 132 * 4 ? 128 * NSTEP * 32 : 128 * 132 * 4;  // slice, or the output tile
  static bool raised = false;
  if (!raised) {
    I8IE_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&linear_skinny_kernel<NSTEP>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    raised = true;
  }
  linear_skinny_kernel<NSTEP><<<grid, 256, lds, ctx->stream>>>(a);
  I8IE_LAUNCH_CHECK();
  return I8IE_OK;
}

}  // namespace

// ---- entry points used by i8ie_layer.hip -----------------------------------------------------------
int i8ie_launch_frag_pack(i8ie_ctx* ctx, const int8_t* B, void* Bf, int Npad, int Kpad) {
  const int64_t total = (int64_t)(Npad / 32) * (Kpad / 32) * 64;
  int64_t blocks = (total + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  frag_pack_kernel<<<(int)blocks, 256, 0, ctx->stream>>>(B, (v4i*)Bf, total, Kpad, Kpad / 32);
  I8IE_LAUNCH_CHECK();
  return I8IE_OK;
}

// Plan: k-steps per slice (multiple of 4, <= 36: 144 KiB of LDS) and the slice count, so that the grid has about
// one block per CU.  Returns 0 when the shape does not suit this kernel.
int i8ie_skinny_plan(int m, int n, int Kpad, int* nstep, int* slices) {
  // measured on AlexNet fc6 / fc7 (bench.py --batch 125 / 250 / 500): one row block (m <= 128) 37 us, against 47 us
  // for the tiled split-K kernel while that still stored its partial tiles a dword per lane and row, and against
  // 33 us once it stored row segments too.  Kept as an opt-in variant ($I8IE_SKINNY=1, single row block only).
  if (m < 1 || m > 128 || n < 64 || Kpad < 512) return 0;
  const int ksteps = Kpad / 32;
  const int groups = ((n + 31) / 32 + 3) / 4, mblocks = (m + 127) / 128;
  int ks = (256 + groups * mblocks - 1) / (groups * mblocks);
  if (ks < 1) ks = 1;
  if (ks > 8) ks = 8;
  int st = ((ksteps + ks - 1) / ks + 3) / 4 * 4;
  if (st > 36) st = 36;
  if (st < 8) st = 8;
  ks = (ksteps + st - 1) / st;
  if (ks > 16) return 0;  // too many partial slabs: the tiled kernel's trade-off is better
  *nstep = st;
  *slices = ks;
  return 1;
}

int i8ie_launch_linear_skinny(i8ie_ctx* ctx, const uint8_t* A, size_t lda, int m, const void* Bf, int Kpad, int Npad,
                              int n, const int32_t* ocp, int32_t* partial, int nstep, int slices) {
  I8IE_REQUIRE(m > 0 && (size_t)m * lda < ((size_t)1 << 32) - 4096 && lda % 16 == 0, "skinny linear shape");
  I8IE_REQUIRE((reinterpret_cast<uintptr_t>(A) & 15u) == 0, "activations must be 16-byte aligned");
  SkinnyArgs a{};
  a.A = A; a.a_bytes = (unsigned)((size_t)m * lda); a.lda = (unsigned)lda; a.M = m;
  a.Bf = (const v4i*)Bf; a.ksteps = Kpad / 32; a.Npad = Npad; a.N = n; a.ocp = ocp; a.partial = partial;
  const dim3 grid(((Npad / 32) + 3) / 4, slices, (m + 127) / 128);
  I8ieProfScope prof(ctx, "linear_skinny_mfma", 2.0 * m * n * (double)Kpad, (double)m * Kpad + (double)n * Kpad + 4.0 * slices * m * n);
  switch (nstep) {
    case 8: return launch_skinny<8>(ctx, a, grid);
    case 12: return launch_skinny<12>(ctx, a, grid);
    case 16: return launch_skinny<16>(ctx, a, grid);
    case 20: return launch_skinny<20>(ctx, a, grid);
    case 24: return launch_skinny<24>(ctx, a, grid);
    case 28: return launch_skinny<28>(ctx, a, grid);
    case 32: return launch_skinny<32>(ctx, a, grid);
    case 36: return launch_skinny<36>(ctx, a, grid);
    default: break;
  }
  i8ie_set_error("linear_skinny: unsupported slice length %d", nstep);
  return I8IE_ERR_ARG;
}
