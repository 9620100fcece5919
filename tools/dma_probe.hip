// dma_probe.hip -- what does one LDS-DMA piece (buffer_load_dwordx4 ... lds, 1 KiB per wave-instruction) cost the
// wave that issues it, and what does a CU sustain?  Tuning aid for csrc/i8ie_pp.hip (not part of the product).
//   mode 0: buffer_load ... lds       mode 1: global_load_lds       mode 2: buffer_load to VGPRs (no LDS)
// Every wave issues `pieces` instructions per round back to back, then waits vmcnt(0); `rounds` rounds.
// The source is a per-CU window of `window` bytes (64 KiB: L2 / L1 resident; 64 MiB: streams from HBM / MALL).
// Reports shader cycles per piece per wave (s_memtime) and bytes per clock per CU.
// build: hipcc --offload-arch=gfx950 -O3 tools/dma_probe.hip -o tools/dma_probe
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

typedef int v4i __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(1024) void probe(const unsigned char* src, size_t window, int rounds, int pieces,
                                              unsigned long long* out, int* sink) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nw = blockDim.x >> 6;
  const unsigned char* base = src + (size_t)blockIdx.x * window;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(base), 0, (unsigned)window, 0x00020000);
  __syncthreads();
  v4i keep = {0, 0, 0, 0};
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  unsigned off = (unsigned)(wave * 1024 + lane * 16);
  for (int r = 0; r < rounds; ++r) {
    for (int i = 0; i < pieces; ++i) {
      const unsigned o = off & (unsigned)(window - 1);
      if (MODE == 0) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(smem + wave * 8192 + (i & 7) * 1024), 16, (int)o, 0, 0, 0);
      } else if (MODE == 1) {
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base + o),
                                         (__attribute__((address_space(3))) void*)(smem + wave * 8192 + (i & 7) * 1024), 16, 0, 0);
      } else {
        const v4i v = __builtin_bit_cast(v4i, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)o, 0, 0));
        keep ^= v;
      }
      off += (unsigned)(nw * 1024);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (lane == 0) out[blockIdx.x * 16 + wave] = t1 - t0;
  if (keep.x == 0x12345678) sink[0] = keep.y + keep.z + keep.w;
}

// MFMA waves beside DMA waves: waves [0, nd) issue LDS-DMA bursts, waves [nd, nw) run back-to-back 16x16x64 int8
// MFMAs from registers.  Reports the DMA rate and the MFMA rate when both run together.
__global__ __launch_bounds__(1024) void mixed(const unsigned char* src, size_t window, int rounds, int pieces, int nd,
                                              int mfma_per_round, unsigned long long* out, int* sink, int pitch = 128) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const unsigned char* base = src + (size_t)blockIdx.x * window;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(base), 0, (unsigned)window, 0x00020000);
  v4i a = {tid * 7, tid * 13, tid ^ 0x55, tid * 0x01010101}, b = {tid * 3, tid * 17, tid ^ 0x33, tid * 0x01030107};
  v4i c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  if (wave < nd) {
    // a piece = 8 rows x 128 B at `pitch` (128: one contiguous KiB)
    unsigned off = (unsigned)(wave * 8 * pitch + (lane >> 3) * pitch + (lane & 7) * 16);
    for (int r = 0; r < rounds; ++r) {
      for (int i = 0; i < pieces; ++i) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(smem + wave * 8192 + (i & 7) * 1024), 16,
                                                 (int)(off & (unsigned)(window - 1)), 0, 0, 0);
        off += (unsigned)(nd * 8 * pitch);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
  } else {
    for (int r = 0; r < rounds; ++r)
      for (int i = 0; i < mfma_per_round; i += 4) {
        c0 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_i32_16x16x64_i8(b, a, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, a, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_i32_16x16x64_i8(b, b, c3, 0, 0, 0);
      }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (lane == 0) out[blockIdx.x * 16 + wave] = t1 - t0;
  if ((c0.x ^ c1.y ^ c2.z ^ c3.w) == 0x12345678) sink[0] = 1;
}

int main() {
  const int CUS = 256;
  unsigned char* src;
  const size_t big = (size_t)CUS * (64u << 20) / 64;  // 256 x 1 MiB windows... sized below per test
  (void)big;
  const size_t total = (size_t)CUS * (4u << 20);
  hipMalloc(&src, total);
  hipMemset(src, 1, total);
  unsigned long long* out;
  int* sink;
  hipMalloc(&out, CUS * 16 * sizeof(unsigned long long));
  hipMalloc(&sink, 64);
  std::vector<unsigned long long> h(CUS * 16);
  const char* names[3] = {"buffer_load lds", "global_load_lds", "buffer_load vgpr"};
  hipFuncSetAttribute(reinterpret_cast<const void*>(&probe<0>), hipFuncAttributeMaxDynamicSharedMemorySize, 16 * 8192);
  hipFuncSetAttribute(reinterpret_cast<const void*>(&probe<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 16 * 8192);
  hipFuncSetAttribute(reinterpret_cast<const void*>(&probe<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 16 * 8192);
  for (int mode = 0; mode < 0; ++mode)
    for (size_t window : {(size_t)65536, (size_t)4 << 20})
      for (int waves : {1, 4, 8, 12})
        for (int pieces : {2, 8, 32}) {
          const int rounds = 2048 / pieces;
          hipMemset(out, 0, CUS * 16 * sizeof(unsigned long long));
          for (int rep = 0; rep < 2; ++rep) {
            if (mode == 0) probe<0><<<CUS, waves * 64, 16 * 8192, 0>>>(src, window, rounds, pieces, out, sink);
            if (mode == 1) probe<1><<<CUS, waves * 64, 16 * 8192, 0>>>(src, window, rounds, pieces, out, sink);
            if (mode == 2) probe<2><<<CUS, waves * 64, 16 * 8192, 0>>>(src, window, rounds, pieces, out, sink);
          }
          hipDeviceSynchronize();
          hipMemcpy(h.data(), out, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
          double sum = 0;
          int n = 0;
          for (int b = 0; b < CUS; ++b)
            for (int w = 0; w < waves; ++w) {
              sum += (double)h[b * 16 + w];
              ++n;
            }
          const double cyc = sum / n;  // cycles per wave for rounds * pieces pieces
          const double per_piece = cyc / (rounds * pieces);
          printf("%-16s window %7zu KiB  waves/CU %2d  burst %2d: %7.1f cycles per piece per wave, %6.1f B/clk/CU\n", names[mode],
                 window >> 10, waves, pieces, per_piece, 1024.0 * waves / per_piece);
        }
  hipFuncSetAttribute(reinterpret_cast<const void*>(&mixed), hipFuncAttributeMaxDynamicSharedMemorySize, 16 * 8192);
  for (int pitch : {128, 384, 3456})
   for (int nd : {4, 8})
    for (int nm : {0, 4})
      for (int pieces : {2, 8}) {
        const int rounds = 1024 / pieces, mpr = pieces * 16;  // MFMA waves: 16 MFMAs (256 cycles) per DMA piece of the others
        const int waves = nd + nm;
        for (int rep = 0; rep < 2; ++rep) mixed<<<CUS, waves * 64, 16 * 8192, 0>>>(src, (size_t)65536, rounds, pieces, nd, mpr, out, sink, pitch);
        hipDeviceSynchronize();
        hipMemcpy(h.data(), out, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
        double sd = 0, sm = 0;
        for (int b = 0; b < CUS; ++b) {
          for (int w = 0; w < nd; ++w) sd += (double)h[b * 16 + w];
          for (int w = nd; w < waves; ++w) sm += (double)h[b * 16 + w];
        }
        const double dcyc = sd / (CUS * nd) / (rounds * pieces);
        printf("mixed pitch %4d: %d DMA waves + %d MFMA waves per CU, burst %d: %6.1f cycles per piece per DMA wave (%5.1f B/clk/CU)", pitch, nd, nm, pieces,
               dcyc, 1024.0 * nd / dcyc);
        if (nm) printf(", MFMA waves %5.1f cycles per MFMA", sm / (CUS * nm) / ((double)rounds * mpr));
        printf("\n");
      }
  return 0;
}
