// What a CU can pull from L2 / HBM per clock on MI355X, by access path and sharing pattern (tuning aid, not part of the
// product).  Background: fc6 / fc7 at 1000 rows stream 2.4 MB of operands per CU and run at 20-30 B/clk/CU whatever the
// staging; this probe measures the ceilings those numbers should be read against.
//
//   path 0: buffer_load_b128 into registers (1 KiB per wave instruction, 8 in flight per wave)
//   path 1: LDS-DMA (buffer_load ... lds, 1 KiB pieces, 8 in flight per wave), no ds_read afterwards
//   sharing 0: every block streams ITS OWN region          (256 x bytes in all: HBM once the total exceeds the caches)
//   sharing 1: the 32 blocks of an XCD (blockIdx & 7) stream the SAME region  (L2 hits after the first touch)
//   sharing 2: all blocks stream the same region
// Each block walks its region `passes` times; B/clk/CU = bytes per block / s_memtime ticks of the block (median over blocks).
// build: hipcc --offload-arch=gfx950 -O3 tools/l2_probe.hip -o tools/l2_probe
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef int v4i __attribute__((ext_vector_type(4)));

template <int PATH>
__global__ __launch_bounds__(512) void stream(const unsigned char* src, size_t region, int sharing, int passes, unsigned long long* out, int* sink) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nw = blockDim.x >> 6;
  const size_t base = sharing == 0 ? (size_t)blockIdx.x * region : (sharing == 1 ? (size_t)(blockIdx.x & 7) * region : 0);
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(src + base), 0, (unsigned)region, 0x00020000);
  const unsigned pieces = (unsigned)(region >> 10);
  // every block starts at a different piece (as the Linear kernels do), every wave takes pieces w, w + nw, ...
  const unsigned rot = (blockIdx.x * 37u) % pieces;
  v4i acc = {0, 0, 0, 0};
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int ps = 0; ps < passes; ++ps) {
    for (unsigned p0 = wave; p0 < pieces; p0 += 8 * nw) {
      v4i v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        unsigned pi = p0 + j * nw + rot;
        pi = pi % pieces;
        const int off = (int)(pi * 1024u + lane * 16u);
        if constexpr (PATH == 0) {
          v[j] = __builtin_bit_cast(v4i, __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0));
        } else {
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(lds + (wave * 8 + j) * 1024), 16, off, 0, 0, 0);
        }
      }
      if constexpr (PATH == 0) {
#pragma unroll
        for (int j = 0; j < 8; ++j) acc ^= v[j];
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (acc.x == 0x7fffffff) sink[tid] = acc.y;
  if (tid == 0) out[blockIdx.x] = t1 - t0;
}

int main(int argc, char** argv) {
  const int blocks = 256;
  unsigned char* src;
  const size_t max_region = 4u << 20;
  hipMalloc(&src, (size_t)blocks * max_region);
  hipMemset(src, 3, (size_t)blocks * max_region);
  unsigned long long* d_out;
  int* sink;
  hipMalloc(&d_out, blocks * 8);
  hipMalloc(&sink, 4096);
  hipFuncSetAttribute(reinterpret_cast<const void*>(&stream<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
  std::vector<unsigned long long> h(blocks);
  printf("bytes per clock and CU (s_memtime ticks; median block), 256 blocks, 8 KiB in flight per wave\n");
  for (int threads : {256, 512}) {
    for (int path = 0; path < 2; ++path) {
      for (int sharing = 0; sharing < 3; ++sharing) {
        for (size_t region : {(size_t)256 << 10, (size_t)1 << 20, (size_t)4 << 20}) {
          const int passes = (int)((16u << 20) / region);  // 16 MiB streamed per block
          for (int rep = 0; rep < 2; ++rep) {
            if (path == 0) stream<0><<<blocks, threads, 0>>>(src, region, sharing, passes, d_out, sink);
            else stream<1><<<blocks, threads, 64 * 1024>>>(src, region, sharing, passes, d_out, sink);
            hipDeviceSynchronize();
          }
          hipMemcpy(h.data(), d_out, blocks * 8, hipMemcpyDeviceToHost);
          std::sort(h.begin(), h.end());
          const double bytes = (double)region * passes;
          printf("%d waves/CU  %-22s %-28s region %4zu KiB x %3d passes: %6.1f B/clk/CU (slowest block %6.1f)\n", threads / 64,
                 path == 0 ? "loads to registers" : "LDS-DMA", sharing == 0 ? "own region per block" : (sharing == 1 ? "one region per XCD" : "one region for all"),
                 region >> 10, passes, bytes / (double)h[blocks / 2], bytes / (double)h[blocks - 1]);
          fflush(stdout);
        }
      }
    }
  }
  printf("status %s\n", hipGetErrorString(hipGetLastError()));
  return 0;
}
