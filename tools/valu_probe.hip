// What a vector instruction costs on a gfx950 SIMD, alone and next to a wave that issues MFMAs back to back
// (tuning aid, not part of the product).  The epilogues of the contraction kernels are VALU-issue bound
// (DESIGN.md section 4): this probe prices the instructions they are made of.
//
//   solo:  one wave per SIMD runs 8 independent chains of one instruction          -> cycles per instruction
//   pair:  waves 0-3 issue v_mfma_i32_32x32x32_i8 (or 16x16x64) on 4 (8) independent accumulators,
//          waves 4-7 (same SIMDs) run the instruction until the MFMA waves are done  -> both rates
//   two:   two waves per SIMD both run the instruction                             -> cycles per instruction and wave
// Cycles are s_memtime ticks (100 MHz reference on this part would show as ~20x fewer: the MFMA solo line calibrates it:
// 32x32x32 i8 is 8 passes = 32 cycles of the shader clock).
// build: hipcc --offload-arch=gfx950 -O3 tools/valu_probe.hip -o tools/valu_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
typedef float v2f __attribute__((ext_vector_type(2)));

#define OPS(X)                                                                              \
  X(0, "v_cvt_f32_i32 %0, %1", "=v", "v")                                                   \
  X(1, "v_fma_f32 %0, %1, %1, %1", "=v", "v")                                               \
  X(2, "v_max_f32 %0, %1, %1", "=v", "v")                                                   \
  X(3, "v_cvt_pk_u8_f32 %0, %1, 1, %1", "=v", "v")                                          \
  X(4, "v_fract_f32 %0, %1", "=v", "v")                                                     \
  X(5, "v_add_f32 %0, %1, %1", "=v", "v")                                                   \
  X(6, "v_min3_f32 %0, %1, |%1|, |%1|", "=v", "v")                                          \
  X(7, "v_perm_b32 %0, %1, %1, %1", "=v", "v")                                              \
  X(8, "v_pk_max_u16 %0, %1, %1", "=v", "v")                                                \
  X(9, "v_add_u32 %0, %1, %1", "=v", "v")                                                   \
  X(10, "v_max_i32 %0, %1, %1", "=v", "v")                                                  \
  X(11, "v_mov_b32 %0, %1", "=v", "v")                                                      \
  X(12, "v_max3_i32 %0, %1, %1, %1", "=v", "v")                                             \
  X(13, "v_lshl_add_u32 %0, %1, 2, %1", "=v", "v")                                          \
  X(14, "v_and_b32 %0, %1, %1", "=v", "v")                                                  \
  X(15, "v_cvt_i32_f32 %0, %1", "=v", "v")                                                  \
  X(16, "v_rcp_f32 %0, %1", "=v", "v")                                                      \
  X(17, "v_mul_lo_u32 %0, %1, %1", "=v", "v")                                               \
  X(18, "v_med3_f32 %0, %1, %1, %1", "=v", "v")                                             \
  X(19, "v_cmp_lt_f32 vcc, %1, %1\n\tv_mov_b32 %0, %1", "=v", "v")                          \
  X(20, "v_cvt_pk_i16_i32 %0, %1, %1", "=v", "v")                                           \
  X(21, "v_max_i32_sdwa %0, %1, %1 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD", "+v", "v") \
  X(22, "v_mul_f32 %0, %1, %1", "=v", "v")                                                  \
  X(23, "v_xor_b32 %0, %1, %1", "=v", "v")                                                  \
  X(24, "v_rndne_f32 %0, %1", "=v", "v")                                                    \
  X(25, "v_floor_f32 %0, %1", "=v", "v")                                                    \
  X(26, "v_cvt_f32_ubyte0 %0, %1", "=v", "v")                                               \
  X(27, "v_mad_i32_i24 %0, %1, %1, %1", "=v", "v")                                          \
  X(28, "v_cvt_u32_f32 %0, %1", "=v", "v")                                                  \
  X(29, "v_sat_pk_u8_i16 %0, %1", "=v", "v")
constexpr int kNumOps = 30;
static const char* kNames[] = {
    "v_cvt_f32_i32", "v_fma_f32", "v_max_f32", "v_cvt_pk_u8_f32", "v_fract_f32", "v_add_f32", "v_min3_f32 |..|", "v_perm_b32",
    "v_pk_max_u16", "v_add_u32", "v_max_i32", "v_mov_b32", "v_max3_i32", "v_lshl_add_u32", "v_and_b32", "v_cvt_i32_f32",
    "v_rcp_f32", "v_mul_lo_u32", "v_med3_f32", "v_cmp_lt_f32+v_mov", "v_cvt_pk_i16_i32", "v_max_i32_sdwa byte", "v_mul_f32", "v_xor_b32",
    "v_rndne_f32", "v_floor_f32", "v_cvt_f32_ubyte0", "v_mad_i32_i24", "v_cvt_u32_f32", "v_sat_pk_u8_i16"};
// packed fp32 (64-bit operands) and LDS reads get their own bodies below: ids 100.., see run_op
constexpr int kPkFma = 100, kPkAdd = 101, kPkMul = 102, kDsRead = 103, kDsWrite64 = 104;

template <int OP>
__device__ __forceinline__ void one(int (&r)[8], int x, v2f (&p)[8], v2f px, unsigned char* lds, v4i (&q)[8]) {
#define X(id, text, outc, inc)                                           \
  if constexpr (OP == id) {                                              \
    _Pragma("unroll") for (int i = 0; i < 8; ++i) asm volatile(text : outc(r[i]) : inc(x)); \
  }
  OPS(X)
#undef X
  if constexpr (OP == kPkFma) {
#pragma unroll
    for (int i = 0; i < 8; ++i) asm volatile("v_pk_fma_f32 %0, %1, %1, %1" : "=v"(p[i]) : "v"(px));
  }
  if constexpr (OP == kPkAdd) {
#pragma unroll
    for (int i = 0; i < 8; ++i) asm volatile("v_pk_add_f32 %0, %1, %1" : "=v"(p[i]) : "v"(px));
  }
  if constexpr (OP == kPkMul) {
#pragma unroll
    for (int i = 0; i < 8; ++i) asm volatile("v_pk_mul_f32 %0, %1, %1" : "=v"(p[i]) : "v"(px));
  }
  if constexpr (OP == kDsRead) {
#pragma unroll
    for (int i = 0; i < 8; ++i) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(q[i]) : "v"(x), "n"(0));
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  if constexpr (OP == kDsWrite64) {
#pragma unroll
    for (int i = 0; i < 8; ++i) asm volatile("ds_write_b64 %0, %1" ::"v"(x), "v"(p[i]) : "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
}

// MF: 0 = no MFMA waves (all waves run the instruction), 1 = 32x32x32 MFMA waves 0-3, 2 = 16x16x64 MFMA waves 0-3
template <int OP, int MFP>
__global__ __launch_bounds__(512) void probe(int iters, unsigned long long* out, int* sink) {
  constexpr int MF = MFP & 3;  // MFP bit 2: the instruction waves run at s_setprio 3; bit 3: the MFMA waves do
  __shared__ volatile int done;
  __shared__ __attribute__((aligned(16))) unsigned char lds[65536];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  if (tid == 0) done = 0;
  for (int i = tid; i < 16384; i += blockDim.x) reinterpret_cast<int*>(lds)[i] = i;
  __syncthreads();
  const bool mf_wave = MF != 0 && ((MFP & 16) ? wave >= 4 : wave < 4);  // (MFP bit 4: the MFMA waves are the YOUNGER ones)
  unsigned long long t0 = 0, t1 = 0, n = 0;
  if ((MFP & 4) && !mf_wave) __builtin_amdgcn_s_setprio(3);
  if ((MFP & 8) && mf_wave) __builtin_amdgcn_s_setprio(3);
  if (mf_wave) {
    v4i a = {tid, 1, 2, 3}, b = {9, tid, 2, 3};
    if constexpr (MF == 1) {
      v16i acc[4];
      for (int j = 0; j < 4; ++j)
        for (int e = 0; e < 16; ++e) acc[j][e] = e + tid;
      t0 = __builtin_amdgcn_s_memtime();
      for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int rep = 0; rep < 2; ++rep)
#pragma unroll
          for (int j = 0; j < 4; ++j) asm volatile("v_mfma_i32_32x32x32_i8 %0, %1, %2, %0" : "+v"(acc[j]) : "v"(a), "v"(b));
      }
      t1 = __builtin_amdgcn_s_memtime();
      n = (unsigned long long)iters * 8;
      int s = 0;
      for (int j = 0; j < 4; ++j) s += acc[j][0] + acc[j][7];
      if (s == 0x7fffffff) sink[tid] = s;
    } else {
      v4i acc[8];
      for (int j = 0; j < 8; ++j) acc[j] = v4i{tid, j, 2, 3};
      t0 = __builtin_amdgcn_s_memtime();
      for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 8; ++j) asm volatile("v_mfma_i32_16x16x64_i8 %0, %1, %2, %0" : "+v"(acc[j]) : "v"(a), "v"(b));
      }
      t1 = __builtin_amdgcn_s_memtime();
      n = (unsigned long long)iters * 8;
      int s = 0;
      for (int j = 0; j < 8; ++j) s += acc[j].x;
      if (s == 0x7fffffff) sink[tid] = s;
    }
    __builtin_amdgcn_s_waitcnt(0);
    if (lane == 0) atomicAdd(const_cast<int*>(&done), 1);
  } else {
    int r[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    v2f p[8];
    v4i q[8];
    for (int i = 0; i < 8; ++i) p[i] = v2f{1.0f, 2.0f}, q[i] = v4i{0, 0, 0, 0};
    int x = (lane * 272 + (tid >> 6) * 16) & 0xfff0;
    v2f px = {(float)tid, 1.5f};
    t0 = __builtin_amdgcn_s_memtime();
    if (MF == 0) {
      for (int it = 0; it < iters; ++it) one<OP>(r, x, p, px, lds, q);
      n = (unsigned long long)iters * 8;
    } else {
      while (done < 4) {
#pragma unroll
        for (int u = 0; u < 8; ++u) one<OP>(r, x, p, px, lds, q);
        n += 64;
      }
    }
    t1 = __builtin_amdgcn_s_memtime();
    int s = 0;
    for (int i = 0; i < 8; ++i) s += r[i] + (int)p[i].x + q[i].x;
    if (s == 0x7ffffff1) sink[tid] = s;
  }
  if (lane == 0) {
    out[((size_t)blockIdx.x * 8 + wave) * 2 + 0] = t1 - t0;
    out[((size_t)blockIdx.x * 8 + wave) * 2 + 1] = n;
  }
}

template <int OP>
static void run_op(const char* name, int iters, unsigned long long* d_out, int* sink, int blocks) {
  std::vector<unsigned long long> h((size_t)blocks * 16);
  auto rate = [&](int w0, int w1) {  // mean cycles per instruction over waves [w0, w1) of all blocks
    double c = 0, k = 0;
    for (int b = 0; b < blocks; ++b)
      for (int w = w0; w < w1; ++w) {
        c += (double)h[((size_t)b * 8 + w) * 2];
        k += (double)h[((size_t)b * 8 + w) * 2 + 1];
      }
    return k > 0 ? c / k : 0.0;
  };
  double solo, two, p32m, p32v, p16m, p16v, q32m, q32v, q16m, q16v, r16m, r16v;
  hipMemset(d_out, 0, h.size() * 8);
  probe<OP, 0><<<blocks, 256>>>(iters, d_out, sink);
  hipMemcpy(h.data(), d_out, h.size() * 8, hipMemcpyDeviceToHost);
  solo = rate(0, 4);
  probe<OP, 0><<<blocks, 512>>>(iters, d_out, sink);
  hipMemcpy(h.data(), d_out, h.size() * 8, hipMemcpyDeviceToHost);
  two = rate(0, 8);
  probe<OP, 1><<<blocks, 512>>>(iters, d_out, sink);
  hipMemcpy(h.data(), d_out, h.size() * 8, hipMemcpyDeviceToHost);
  p32m = rate(0, 4);
  p32v = rate(4, 8);
  probe<OP, 2><<<blocks, 512>>>(iters, d_out, sink);
  hipMemcpy(h.data(), d_out, h.size() * 8, hipMemcpyDeviceToHost);
  p16m = rate(0, 4);
  p16v = rate(4, 8);
  probe<OP, 5><<<blocks, 512>>>(iters, d_out, sink);
  hipMemcpy(h.data(), d_out, h.size() * 8, hipMemcpyDeviceToHost);
  q32m = rate(0, 4);
  q32v = rate(4, 8);
  probe<OP, 6><<<blocks, 512>>>(iters, d_out, sink);
  hipMemcpy(h.data(), d_out, h.size() * 8, hipMemcpyDeviceToHost);
  q16m = rate(0, 4);
  q16v = rate(4, 8);
  probe<OP, 10><<<blocks, 512>>>(iters, d_out, sink);
  hipMemcpy(h.data(), d_out, h.size() * 8, hipMemcpyDeviceToHost);
  r16m = rate(0, 4);
  r16v = rate(4, 8);
  probe<OP, 17><<<blocks, 512>>>(iters, d_out, sink);
  hipMemcpy(h.data(), d_out, h.size() * 8, hipMemcpyDeviceToHost);
  const double s32m = rate(4, 8), s32v = rate(0, 4);
  printf("%-22s solo %6.2f | two waves %6.2f each | beside 32x32x32: mfma %6.2f valu %6.2f | beside 16x16x64: mfma %6.2f valu %6.2f | instruction waves at setprio 3: 32x32x32 mfma %6.2f valu %6.2f, 16x16x64 mfma %6.2f valu %6.2f | mfma waves at setprio 3: 16x16x64 mfma %6.2f valu %6.2f | MFMA (32x32x32) on the YOUNGER waves 4-7: mfma %6.2f valu %6.2f\n", name, solo,
         two, p32m, p32v, p16m, p16v, q32m, q32v, q16m, q16v, r16m, r16v, s32m, s32v);
  fflush(stdout);
}


// mix: every wave issues [one MFMA, K independent VALU instructions] repeatedly -- does a wave's own vector work fit in the
// shadow of its own MFMAs?  (waves per SIMD: blockDim / 256)
template <int K, int MF>
__global__ __launch_bounds__(512) void mix(int iters, unsigned long long* out, int* sink) {
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  v4i a = {tid, 1, 2, 3}, b = {9, tid, 2, 3};
  float r[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  float x = (float)tid;
  unsigned long long t0, t1;
  if constexpr (MF == 1) {
    v16i acc[4];
    for (int j = 0; j < 4; ++j)
      for (int e = 0; e < 16; ++e) acc[j][e] = e + tid;
    t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        asm volatile("v_mfma_i32_32x32x32_i8 %0, %1, %2, %0" : "+v"(acc[j & 3]) : "v"(a), "v"(b));
#pragma unroll
        for (int k = 0; k < K; ++k) asm volatile("v_fma_f32 %0, %1, %1, %1" : "=v"(r[k & 7]) : "v"(x));
      }
    }
    t1 = __builtin_amdgcn_s_memtime();
    int s = 0;
    for (int j = 0; j < 4; ++j) s += acc[j][0] + acc[j][7];
    if (s == 0x7fffffff) sink[tid] = s;
  } else {
    v4i acc[8];
    for (int j = 0; j < 8; ++j) acc[j] = v4i{tid, j, 2, 3};
    t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        asm volatile("v_mfma_i32_16x16x64_i8 %0, %1, %2, %0" : "+v"(acc[j]) : "v"(a), "v"(b));
#pragma unroll
        for (int k = 0; k < K; ++k) asm volatile("v_fma_f32 %0, %1, %1, %1" : "=v"(r[k & 7]) : "v"(x));
      }
    }
    t1 = __builtin_amdgcn_s_memtime();
    int s = 0;
    for (int j = 0; j < 8; ++j) s += acc[j].x;
    if (s == 0x7fffffff) sink[tid] = s;
  }
  float s2 = 0;
  for (int i = 0; i < 8; ++i) s2 += r[i];
  if (s2 == 12345.0f) sink[tid] = 1;
  if (lane == 0) {
    out[((size_t)blockIdx.x * 8 + wave) * 2 + 0] = t1 - t0;
    out[((size_t)blockIdx.x * 8 + wave) * 2 + 1] = (unsigned long long)iters * 8;
  }
}
template <int K>
static void run_mix(int iters, unsigned long long* d_out, int blocks, int* sink) {
  std::vector<unsigned long long> h((size_t)blocks * 16);
  auto rate = [&](int w1) {
    double c = 0, k = 0;
    for (int b = 0; b < blocks; ++b)
      for (int w = 0; w < w1; ++w) {
        c += (double)h[((size_t)b * 8 + w) * 2];
        k += (double)h[((size_t)b * 8 + w) * 2 + 1];
      }
    return c / k;
  };
  double v[4];
  mix<K, 1><<<blocks, 256>>>(iters, d_out, sink);
  hipMemcpy(h.data(), d_out, h.size() * 8, hipMemcpyDeviceToHost);
  v[0] = rate(4);
  mix<K, 1><<<blocks, 512>>>(iters, d_out, sink);
  hipMemcpy(h.data(), d_out, h.size() * 8, hipMemcpyDeviceToHost);
  v[1] = rate(8);
  mix<K, 2><<<blocks, 256>>>(iters, d_out, sink);
  hipMemcpy(h.data(), d_out, h.size() * 8, hipMemcpyDeviceToHost);
  v[2] = rate(4);
  mix<K, 2><<<blocks, 512>>>(iters, d_out, sink);
  hipMemcpy(h.data(), d_out, h.size() * 8, hipMemcpyDeviceToHost);
  v[3] = rate(8);
  printf("mix: MFMA + %d v_fma_f32 per wave: ticks per group and wave | 32x32x32: one wave/SIMD %6.2f, two %6.2f | 16x16x64: one %6.2f, two %6.2f\n", K, v[0], v[1], v[2], v[3]);
  fflush(stdout);
}

// kloop: the K loop of a patch-stationary contraction as ONE wave would run it with the previous tile's requantisation
// folded in: per k-step of 64 (two MFMA k-steps of 32) 16 x v_mfma_i32_32x32x32_i8 on 8 accumulator tiles (wave tile
// 128 x 64), 8 ds_read_b128 (A fragments, conflict-free), 4 buffer loads of 1 KiB (weight fragments, L2-resident panel),
// 8 address adds, and E requantiser-like VALU instructions (cvt / fma / max / cvt_pk / fract / add / min3 mix) on a second
// set of 128 registers.  Ideal: 512 cycles per step per wave on its own SIMD (1 wave / SIMD), 1024 with two.
template <int E, int MODE>
__global__ __launch_bounds__(512) void kloop(int steps, unsigned long long* out, int* sink, const unsigned char* gsrc) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[65536];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  for (int i = tid; i < 16384; i += blockDim.x) reinterpret_cast<int*>(lds)[i] = i * 2654435761u;
  __syncthreads();
  v16i acc[8];
  for (int j = 0; j < 8; ++j)
    for (int e = 0; e < 16; ++e) acc[j][e] = e + tid;
  int old[32];  // (a slice of the previous tile's accumulators: the requantiser's operands)
  for (int j = 0; j < 32; ++j) old[j] = tid * j;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(gsrc), 0, 1u << 24, 0x00020000);
  unsigned abase[4];
  for (int m = 0; m < 4; ++m) abase[m] = (unsigned)(((m * 32 + (lane & 31)) * 272 + (lane >> 5) * 16) & 0xfff0);
  v4i A[2][4], B[2][2][2];
  for (int i = 0; i < 2; ++i)
    for (int m = 0; m < 4; ++m) A[i][m] = v4i{tid, m, i, 3};
  for (int i = 0; i < 2; ++i)
    for (int h = 0; h < 2; ++h)
      for (int n = 0; n < 2; ++n) B[i][h][n] = v4i{tid, h, n, i};
  unsigned kofs = 0;
  auto ldA = [&](v4i (&d)[4], unsigned ko) {
#pragma unroll
    for (int m = 0; m < 4; ++m)
      if (MODE != 4) d[m] = *reinterpret_cast<const v4i*>(lds + ((abase[m] + ko) & 0xfff0));
  };
  auto ldB = [&](v4i (&d)[2][2], int step) {
    if (MODE == 3) return;
    // MODE 0: every wave of the block reads the same 4 KiB per step; 1: the waves w and w + 4 (one SIMD) share theirs, as the
    // two pixel halves of i8ie_pconv.hip do; 2: every wave its own
    const unsigned wsel = MODE == 0 ? 0u : (MODE == 1 ? (unsigned)(wave & 3) : (unsigned)wave);
    const unsigned base = (unsigned)((((blockIdx.x & 3) * 64 + (step & 63)) * 8 + wsel) * 4096 + lane * 16);
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int n = 0; n < 2; ++n) d[h][n] = __builtin_bit_cast(v4i, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(base + (h * 2 + n) * 1024), 0, 0));
  };
  float worst = 1.0f;
  unsigned packed = 0;
  const float ms = 0.0123f, bz = 99.5f, lof = 100.0f;
  ldA(A[0], 0);
  ldB(B[0], 0);
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma clang loop unroll(disable)
  for (int st = 0; st < steps; st += 2) {
#pragma unroll
    for (int par = 0; par < 2; ++par) {
      ldB(B[par ^ 1], st + par + 1);
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        kofs += 16 * 17;
        ldA(A[h ^ 1], kofs);
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
          for (int n = 0; n < 2; ++n) {
            asm volatile("v_mfma_i32_32x32x32_i8 %0, %1, %2, %0" : "+v"(acc[m * 2 + n]) : "v"(B[par][h][n]), "v"(A[h][m]));
            // E / 16 requantiser instructions behind every MFMA
#pragma unroll
            for (int e = 0; e < E / 16; ++e) {
              const int idx = ((par * 2 + h) * 8 + m * 2 + n) * (E / 16) + e;
              const int c = old[idx & 31];
              const float f = __builtin_fmaf((float)c, ms, bz);
              if ((idx & 3) == 0) {
                packed = __builtin_amdgcn_cvt_pk_u8_f32(__builtin_fmaxf(f, lof), idx & 3, packed);
              } else if ((idx & 3) == 1) {
                worst = __builtin_fminf(worst, __builtin_fabsf(__builtin_amdgcn_fractf(f) - 0.5f));
              } else {
                old[idx & 31] = c + (int)packed;
              }
            }
          }
      }
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  int s = (int)packed + (int)worst;
  for (int j = 0; j < 8; ++j) s += acc[j][0] + acc[j][9];
  for (int j = 0; j < 32; ++j) s += old[j];
  if (s == 0x7fffffff) sink[tid] = s;
  if (lane == 0) {
    out[((size_t)blockIdx.x * 8 + wave) * 2 + 0] = t1 - t0;
    out[((size_t)blockIdx.x * 8 + wave) * 2 + 1] = (unsigned long long)steps;
  }
}
template <int E, int MODE>
static void run_kloop(int steps, unsigned long long* d_out, int blocks, int* sink, const unsigned char* gsrc) {
  std::vector<unsigned long long> h((size_t)blocks * 16);
  auto rate = [&](int w1) {
    double c = 0, k = 0;
    for (int b = 0; b < blocks; ++b)
      for (int w = 0; w < w1; ++w) {
        c += (double)h[((size_t)b * 8 + w) * 2];
        k += (double)h[((size_t)b * 8 + w) * 2 + 1];
      }
    return c / k;
  };
  hipMemset(d_out, 0, h.size() * 8);
  kloop<E, MODE><<<blocks, 256>>>(steps, d_out, sink, gsrc);
  hipMemcpy(h.data(), d_out, h.size() * 8, hipMemcpyDeviceToHost);
  const double one = rate(4);
  kloop<E, MODE><<<blocks, 512>>>(steps, d_out, sink, gsrc);
  hipMemcpy(h.data(), d_out, h.size() * 8, hipMemcpyDeviceToHost);
  const double two = rate(8);
  static const char* modes[] = {"weights shared by the block", "weights shared by the SIMD's two waves", "weights per wave", "no weight loads", "no A reads"};
  printf("kloop [%s]: 16 MFMA + 8 ds_read_b128 + 4 x 1 KiB weight loads + %2d requantiser instructions per k-step of 64: one wave/SIMD %7.1f ticks per step (512 = MFMA rate) | two waves/SIMD: mean %7.1f each, the SIMD's pair of steps %7.1f (1024)\n", modes[MODE], E, one, two, 2 * two - one);
  fflush(stdout);
}

template <int OP>
static void run_all(int iters, unsigned long long* d_out, int* sink, int blocks) {
  if constexpr (OP < kNumOps) {
    run_op<OP>(kNames[OP], iters, d_out, sink, blocks);
    run_all<OP + 1>(iters, d_out, sink, blocks);
  }
}

int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 4000;
  const int blocks = argc > 2 ? atoi(argv[2]) : 256;
  unsigned long long* d_out;
  int* sink;
  hipMalloc(&d_out, (size_t)blocks * 16 * 8);
  hipMalloc(&sink, 4096);
  printf("cycles (s_memtime ticks) per instruction and wave; %d blocks, %d iterations of 8 independent instructions\n", blocks, iters);
  printf("(the mfma columns: ticks per MFMA of a wave issuing them back to back; alone that is 32 for 32x32x32 i8 and 16 for 16x16x64 i8)\n");
  unsigned char* gsrc;
  hipMalloc(&gsrc, 1u << 24);
  hipMemset(gsrc, 3, 1u << 24);
  run_kloop<0, 0>(2048, d_out, blocks, sink, gsrc);
  run_kloop<0, 1>(2048, d_out, blocks, sink, gsrc);
  run_kloop<0, 2>(2048, d_out, blocks, sink, gsrc);
  run_kloop<0, 3>(2048, d_out, blocks, sink, gsrc);
  run_kloop<0, 4>(2048, d_out, blocks, sink, gsrc);
  run_kloop<16, 1>(2048, d_out, blocks, sink, gsrc);
  run_kloop<32, 1>(2048, d_out, blocks, sink, gsrc);
  run_kloop<48, 1>(2048, d_out, blocks, sink, gsrc);
  run_kloop<64, 1>(2048, d_out, blocks, sink, gsrc);
  if (argc > 3 && argv[3][0] == 'k') return 0;
  run_mix<0>(iters, d_out, blocks, sink);
  run_mix<1>(iters, d_out, blocks, sink);
  run_mix<2>(iters, d_out, blocks, sink);
  run_mix<3>(iters, d_out, blocks, sink);
  run_mix<4>(iters, d_out, blocks, sink);
  run_mix<6>(iters, d_out, blocks, sink);
  run_mix<8>(iters, d_out, blocks, sink);
  run_mix<12>(iters, d_out, blocks, sink);
  if (argc > 3) return 0;
  run_all<0>(iters, d_out, sink, blocks);
  run_op<kPkFma>("v_pk_fma_f32", iters, d_out, sink, blocks);
  run_op<kPkAdd>("v_pk_add_f32", iters, d_out, sink, blocks);
  run_op<kPkMul>("v_pk_mul_f32", iters, d_out, sink, blocks);
  run_op<kDsRead>("ds_read_b128 (+wait)", iters, d_out, sink, blocks);
  run_op<kDsWrite64>("ds_write_b64 (+wait)", iters, d_out, sink, blocks);
  hipDeviceSynchronize();
  printf("status %s\n", hipGetErrorString(hipGetLastError()));
  return 0;
}
