// Measured ceilings for the int8 MFMA path on this device (tuning aid, not part of the product):
//   mode 0: register-only v_mfma_i32_32x32x32_i8, 4 independent accumulators per wave
//   mode 1: same MFMAs, each k-step's 4 fragments (2 A + 2 B, 16 B per lane each) re-read from LDS
//           (the 64x64-per-wave tiling of the igemm kernel: 1 KiB of ds_read_b128 per MFMA)
//   mode 2: mode 1 + per "tile" 8 ds_write_b128 and one barrier (the staging traffic)
//   mode 3: as 2, but the 8 writes are spread over the 4 k-steps (2 after each MFMA group)
//   mode 4: as 3, writes placed before each MFMA group
//   mode 5: barrier only, no writes (what LDS-DMA staging could approach)
//   mode 6: as 3 with fragment double-buffering (next k-step's reads issued before the MFMAs)
//   mode 7: staging by LDS-DMA: 8 x global_load_lds_dwordx4 per thread per tile (L2-resident source),
//           issued before the MFMAs, vmcnt(0) + barrier after them; no ds_write at all
//   mode 8: as 7 with the DMA issue spread over the k-steps (2 per k-step)
//   mode 10: as 9 with the A tile staged by LDS-DMA (global_load_lds b128) instead of VGPRs + ds_write
//   mode 9: B fragments straight from a fragment-ordered global panel (1 KiB coalesced per wave-load,
//           L2/L1 resident, loaded one tile ahead), A through LDS: 2 A-fragment reads per k-step,
//           4 ds_write_b128 per thread per tile, one barrier
// build: hipcc --offload-arch=gfx950 -O3 tools/mfma_peak.hip -o tools/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

// source buffers hold pseudo-random bytes by default: constant operands let the clock run ~30 % higher
// (see k4), which flatters every structure that streams them.  MFMA_PEAK_CONST=1 restores memset(1).
__global__ void fill_random(unsigned* p, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    unsigned h = (unsigned)i * 2654435761u + 977u;
    h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
    p[i] = h;
  }
}
static void fill_source(unsigned char* g, size_t bytes) {
  if (getenv("MFMA_PEAK_CONST")) {
    hipMemset(g, 1, bytes);
  } else {
    fill_random<<<1024, 256>>>(reinterpret_cast<unsigned*>(g), bytes / 4);
    hipDeviceSynchronize();
  }
}


template <int MODE>
__global__ __launch_bounds__(256) void k(int iters, int* out, const unsigned char* gsrc) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * 256 * 144];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  v16i acc[2][2];
  for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = r + tid;
  v4i a[2] = {{tid, 1, 2, 3}, {tid + 1, 5, 6, 7}}, b[2] = {{9, tid, 2, 3}, {tid + 4, 5, 1, 7}};
  for (int i = tid; i < 2 * 256 * 144 / 4; i += 256) reinterpret_cast<int*>(smem)[i] = i * 2654435761u;
  __syncthreads();
  const int wm = wave >> 1, wn = wave & 1;
  int ard[2], brd[2];
  for (int m = 0; m < 2; ++m) ard[m] = ((wm * 2 + m) * 32 + (lane & 31)) * 144 + (lane >> 5) * 16;
  for (int n = 0; n < 2; ++n) brd[n] = 128 * 144 + ((wn * 2 + n) * 32 + (lane & 31)) * 144 + (lane >> 5) * 16;
  const int wr = (tid >> 3) * 144 + (tid & 7) * 16;
  v4i st[8];
  for (int i = 0; i < 8; ++i) st[i] = v4i{tid + i, i, 3, 4};
  v4i bq[2][2][4];  // mode 9: [parity][n-tile][k-step]
  if (MODE == 9 || MODE == 10) {
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
      for (int ks = 0; ks < 4; ++ks)
        bq[0][n][ks] = *reinterpret_cast<const v4i*>(gsrc + (size_t)(blockIdx.x & 255) * 65536 + ((wn * 2 + n) * 4 + ks) * 1024 + lane * 16);
  }
  v4i a2[2], b2[2];
  // DMA: lane's global source (1 KiB per wave-instruction, 16 B per lane), L2-resident 64 KiB window per block
  const unsigned char* gp = gsrc + (size_t)(blockIdx.x & 255) * 65536 + tid * 16;
  for (int it = 0; it < iters; ++it) {
    if (MODE == 9 || MODE == 10) {
      // static register parity: two tiles per loop trip
#pragma unroll
      for (int par = 0; par < 2; ++par) {
        const unsigned char* nb = gsrc + (size_t)(blockIdx.x & 255) * 65536 + (((it + par + 1) & 3) * 16384);
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
          for (int ks = 0; ks < 4; ++ks)
            bq[par ^ 1][n][ks] = *reinterpret_cast<const v4i*>(nb + ((wn * 2 + n) * 4 + ks) * 1024 + lane * 16);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
#pragma unroll
          for (int m = 0; m < 2; ++m) a[m] = *reinterpret_cast<const v4i*>(smem + ks * 32 + ard[m]);
#pragma unroll
          for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int n = 0; n < 2; ++n)
              acc[m][n] = __builtin_amdgcn_mfma_i32_32x32x32_i8(bq[par][n][ks], a[m], acc[m][n], 0, 0, 0);
          if (MODE == 9) *reinterpret_cast<v4i*>(smem + 256 * 144 + wr + ks * 32 * 144) = st[ks] ^ (int)0x80808080;
          if (MODE == 10)  // A tile of the next k-block by LDS-DMA: 4 x 1 KiB per wave per tile, one per k-step
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gp + ((it * 4 + par * 4 + ks) & 15) * 4096),
                                             (__attribute__((address_space(3))) void*)(smem + 256 * 144 + wave * 1024 + ks * 4096), 16, 0, 0);
        }
        __syncthreads();
      }
      st[it & 3].x += acc[0][0][0];
      ++it;
      continue;
    }
    if (MODE == 7) {
#pragma unroll
      for (int i = 0; i < 8; ++i)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gp + ((it * 8 + i) & 15) * 4096),
                                         (__attribute__((address_space(3))) void*)(smem + 256 * 144 + wave * 1024 + i * 4096), 16, 0, 0);
    }
    if (MODE == 6) {
#pragma unroll
      for (int m = 0; m < 2; ++m) a[m] = *reinterpret_cast<const v4i*>(smem + ard[m]);
#pragma unroll
      for (int n = 0; n < 2; ++n) b[n] = *reinterpret_cast<const v4i*>(smem + brd[n]);
    }
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      if (MODE >= 1 && MODE != 6) {
#pragma unroll
        for (int m = 0; m < 2; ++m) a[m] = *reinterpret_cast<const v4i*>(smem + ks * 32 + ard[m]);
#pragma unroll
        for (int n = 0; n < 2; ++n) b[n] = *reinterpret_cast<const v4i*>(smem + ks * 32 + brd[n]);
      }
      if (MODE == 6 && ks < 3) {
#pragma unroll
        for (int m = 0; m < 2; ++m) a2[m] = *reinterpret_cast<const v4i*>(smem + (ks + 1) * 32 + ard[m]);
#pragma unroll
        for (int n = 0; n < 2; ++n) b2[n] = *reinterpret_cast<const v4i*>(smem + (ks + 1) * 32 + brd[n]);
      }
      if (MODE == 8) {
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gp + ((it * 8 + 2 * ks) & 15) * 4096),
                                         (__attribute__((address_space(3))) void*)(smem + 256 * 144 + wave * 1024 + (2 * ks) * 4096), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gp + ((it * 8 + 2 * ks + 1) & 15) * 4096),
                                         (__attribute__((address_space(3))) void*)(smem + 256 * 144 + wave * 1024 + (2 * ks + 1) * 4096), 16, 0, 0);
      }
      if (MODE == 4) {
        *reinterpret_cast<v4i*>(smem + 256 * 144 + wr + (2 * ks) * 32 * 144) = st[2 * ks] ^ (int)0x80808080;
        *reinterpret_cast<v4i*>(smem + 256 * 144 + wr + (2 * ks + 1) * 32 * 144) = st[2 * ks + 1];
      }
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n) acc[m][n] = __builtin_amdgcn_mfma_i32_32x32x32_i8(b[n], a[m], acc[m][n], 0, 0, 0);
      if (MODE == 3 || MODE == 6) {
        *reinterpret_cast<v4i*>(smem + 256 * 144 + wr + (2 * ks) * 32 * 144) = st[2 * ks] ^ (int)0x80808080;
        *reinterpret_cast<v4i*>(smem + 256 * 144 + wr + (2 * ks + 1) * 32 * 144) = st[2 * ks + 1];
      }
      if (MODE == 6 && ks < 3) {
        a[0] = a2[0]; a[1] = a2[1]; b[0] = b2[0]; b[1] = b2[1];
      }
    }
    if (MODE == 2) {
#pragma unroll
      for (int i = 0; i < 8; ++i)
        *reinterpret_cast<v4i*>(smem + 256 * 144 + wr + i * 32 * 144) = st[i] ^ (int)0x80808080;
    }
    if (MODE >= 2) __syncthreads();
    if (MODE >= 2) st[it & 7].x += acc[0][0][0];
  }
  int s = 0;
  for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) s += acc[i][j][r];
  out[blockIdx.x * 256 + tid] = s;
}


// ---- 256 x 256 block tile, 4 waves of 128 x 128 (16 accumulator tiles = 256 AGPRs per lane) -------------
//   mode 0: LDS structure only: per k-step 4+4 fragment reads, 16 MFMAs, 4 ds_write_b128 from registers
//           into the other stage; one barrier per 128-deep k-block
//   mode 1: as 0 with the staged data loaded from an L2-resident global window (16 x 16 B per thread per k-block)
//   mode 2: as 1 with the ds_writes replaced by LDS-DMA
template <int MODE>
__global__ __launch_bounds__(256) void k2(int iters, int* out, const unsigned char* gsrc) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];  // 2 stages x 512 rows x 144 B
  constexpr int STAGE = 512 * 144;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  v16i acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = r + tid;
  for (int i = tid; i < 2 * STAGE / 4; i += 256) reinterpret_cast<int*>(smem)[i] = i * 2654435761u;
  __syncthreads();
  const int wm = wave >> 1, wn = wave & 1;
  const int ard = (wm * 128 + (lane & 31)) * 144 + (lane >> 5) * 16;
  const int brd = 256 * 144 + (wn * 128 + (lane & 31)) * 144 + (lane >> 5) * 16;
  const int wr = (tid >> 3) * 144 + (tid & 7) * 16;  // 32 rows x 128 B per pass of the block
  v4i st[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) st[i] = v4i{tid + i, i, 3, 4};
  const unsigned char* gp = gsrc + (size_t)(blockIdx.x & 255) * 65536 + tid * 16;
  for (int it = 0; it < iters; ++it) {
    const unsigned char* cur = smem + (it & 1) * STAGE;
    unsigned char* nxt = smem + ((it & 1) ^ 1) * STAGE;
    if (MODE == 1) {
#pragma unroll
      for (int i = 0; i < 16; ++i) st[i] = *reinterpret_cast<const v4i*>(gp + ((it * 16 + i) & 15) * 4096);
    }
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      v4i a[4], b[4];
#pragma unroll
      for (int m = 0; m < 4; ++m) a[m] = *reinterpret_cast<const v4i*>(cur + ard + m * 32 * 144 + ks * 32);
#pragma unroll
      for (int n = 0; n < 4; ++n) b[n] = *reinterpret_cast<const v4i*>(cur + brd + n * 32 * 144 + ks * 32);
#pragma unroll
      for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < 4; ++n) acc[m][n] = __builtin_amdgcn_mfma_i32_32x32x32_i8(b[n], a[m], acc[m][n], 0, 0, 0);
      if (MODE == 2) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gp + ((it * 16 + ks * 4 + i) & 15) * 4096),
                                           (__attribute__((address_space(3))) void*)(nxt + wave * 1024 + (ks * 4 + i) * 4096), 16, 0, 0);
      } else {
#pragma unroll
        for (int i = 0; i < 4; ++i)
          *reinterpret_cast<v4i*>(nxt + wr + (ks * 4 + i) * 32 * 144) = st[ks * 4 + i] ^ (int)0x80808080;
      }
    }
    __syncthreads();
    if (MODE == 0) st[it & 15].x += acc[0][0][0];
  }
  int s = 0;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) s += acc[i][j][r];
  out[blockIdx.x * 256 + tid] = s;
}

template <int MODE>
void run2(const char* name) {
  int* out;
  unsigned char* gsrc;
  const int blocks = 256, iters = 2000;
  hipMalloc(&out, blocks * 256 * 4);
  hipMalloc(&gsrc, 256 * 65536 + 65536);
  fill_source(gsrc, 256 * 65536 + 65536);
  const int lds = 2 * 512 * 144;
  hipFuncSetAttribute(reinterpret_cast<const void*>(k2<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  k2<MODE><<<blocks, 256, lds>>>(200, out, gsrc);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  k2<MODE><<<blocks, 256, lds>>>(iters, out, gsrc);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  hipError_t err = hipGetLastError();
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  const double ops = (double)blocks * 4 * iters * 64 * 65536.0;
  printf("%-36s 256x256/4 waves, 1 block/CU: %.3f ms  %.0f TOPS  (%s)\n", name, ms, ops / (ms * 1e-3) / 1e12, hipGetErrorString(err));
  hipFree(out);
  hipFree(gsrc);
}

// ---- structure sweep: 4 waves (2 x 2), wave tile TM x TN MFMA tiles, LDS-DMA staging, NST stages ----------
template <int TM, int TN, int NST, bool BUF = false, bool XORA = false>
__global__ __launch_bounds__(256) void k3(int iters, int* out, const unsigned char* gsrc) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int ROWS = 64 * (TM + TN);   // A rows + B rows of the block tile
  constexpr int STAGE = ROWS * 128;      // unpadded (the DMA writes 1 KiB runs); reads below use a swizzle
  constexpr int NDMA = ROWS / 8 / 4;     // 1 KiB wave-instructions per wave per k-block
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  v16i acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = r + tid;
  for (int i = tid; i < NST * STAGE / 4; i += 256) reinterpret_cast<int*>(smem)[i] = i * 2654435761u;
  __syncthreads();
  const int wm = wave >> 1, wn = wave & 1;
  int ard[4], brd[4];  // per k-step swizzled fragment addresses: chunk (2 ks + hi) ^ (row & 7)
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    const int sw = ((ks * 2 + (lane >> 5)) ^ (lane & 7)) * 16;
    ard[ks] = (wm * TM * 32 + (lane & 31)) * 128 + sw;
    brd[ks] = (2 * TM * 32 + wn * TN * 32 + (lane & 31)) * 128 + sw;
  }
  const unsigned char* gp = gsrc + (size_t)(blockIdx.x & 255) * 65536 + tid * 16;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(gsrc), 0, 256 * 65536 + 65536, 0x00020000);
  const int voff = (blockIdx.x & 255) * 65536 + tid * 16;
  for (int it = 0; it < iters; ++it) {
    const unsigned char* cur = smem + (NST == 2 ? (it & 1) * STAGE : 0);
    unsigned char* nxt = smem + (NST == 2 ? ((it & 1) ^ 1) * STAGE : 0);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      v4i a[TM], b[TN];
#pragma unroll
      for (int m = 0; m < TM; ++m) {
        a[m] = *reinterpret_cast<const v4i*>(cur + ard[ks] + m * 4096);
        if (XORA) a[m] = a[m] ^ (int)0x80808080;
      }
#pragma unroll
      for (int n = 0; n < TN; ++n) b[n] = *reinterpret_cast<const v4i*>(cur + brd[ks] + n * 4096);
#pragma unroll
      for (int m = 0; m < TM; ++m)
#pragma unroll
        for (int n = 0; n < TN; ++n) acc[m][n] = __builtin_amdgcn_mfma_i32_32x32x32_i8(b[n], a[m], acc[m][n], 0, 0, 0);
      if (NST == 2) {
#pragma unroll
        for (int i = ks * NDMA / 4; i < (ks + 1) * NDMA / 4; ++i) {
          if (BUF)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(nxt + wave * 1024 + i * 4096), 16,
                                                     voff + ((it * NDMA + i) & 15) * 4096, 0, 0, 0);
          else
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gp + ((it * NDMA + i) & 15) * 4096),
                                             (__attribute__((address_space(3))) void*)(nxt + wave * 1024 + i * 4096), 16, 0, 0);
        }
      }
    }
    if (NST == 1) {
      __syncthreads();  // every wave is done reading the stage
#pragma unroll
      for (int i = 0; i < NDMA; ++i)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gp + ((it * NDMA + i) & 15) * 4096),
                                         (__attribute__((address_space(3))) void*)(nxt + wave * 1024 + i * 4096), 16, 0, 0);
    }
    __syncthreads();
  }
  int s = 0;
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) s += acc[i][j][r];
  out[blockIdx.x * 256 + tid] = s;
}

template <int TM, int TN, int NST, bool BUF = false, bool XORA = false>
void run3(int blocks_per_cu) {
  int* out;
  unsigned char* gsrc;
  const int blocks = 256 * blocks_per_cu, iters = 2000;
  hipMalloc(&out, blocks * 256 * 4);
  hipMalloc(&gsrc, 256 * 65536 + 65536);
  fill_source(gsrc, 256 * 65536 + 65536);
  const int lds = NST * 64 * (TM + TN) * 128;
  hipFuncSetAttribute(reinterpret_cast<const void*>(k3<TM, TN, NST, BUF, XORA>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  int occ = 0;
  hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k3<TM, TN, NST, BUF, XORA>, 256, lds);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  k3<TM, TN, NST, BUF, XORA><<<blocks, 256, lds>>>(200, out, gsrc);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  k3<TM, TN, NST, BUF, XORA><<<blocks, 256, lds>>>(iters, out, gsrc);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  hipError_t err = hipGetLastError();
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  const double ops = (double)blocks * 4 * iters * 4 * TM * TN * 65536.0;
  printf("DMA sweep%s%s: wave tile %dx%d (block %dx%d) stages %d launched %d/CU (occupancy %d): %.3f ms  %.0f TOPS  (%s)\n", BUF ? " [buffer_load lds]" : "", XORA ? " [xor A]" : "", TM, TN,
         TM * 64, TN * 64, NST, blocks_per_cu, occ, ms, ops / (ms * 1e-3) / 1e12, hipGetErrorString(err));
  hipFree(out);
  hipFree(gsrc);
}

// ---- register-only MFMA rate as a function of operand toggling (sustained clock under power limits) ---------
//   RANDOM = false: every MFMA multiplies the same two constant fragments
//   RANDOM = true : eight pseudo-random A and B fragments, rotated every instruction
template <bool RANDOM>
__global__ __launch_bounds__(256) void k4(int iters, int* out) {
  const int tid = threadIdx.x + blockIdx.x * 256;
  v16i acc[2][2];
  for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0;
  v4i ap[8], bp[8];
  unsigned h = tid * 2654435761u + 12345u;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      h = h * 1664525u + 1013904223u;
      ap[i][q] = RANDOM ? (int)h : 0x01010101;
      h = h * 1664525u + 1013904223u;
      bp[i][q] = RANDOM ? (int)h : 0x01010101;
    }
  }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      acc[0][0] = __builtin_amdgcn_mfma_i32_32x32x32_i8(bp[u], ap[u], acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_i32_32x32x32_i8(bp[(u + 1) & 7], ap[u], acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_i32_32x32x32_i8(bp[u], ap[(u + 3) & 7], acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_i32_32x32x32_i8(bp[(u + 5) & 7], ap[(u + 2) & 7], acc[1][1], 0, 0, 0);
    }
  }
  int s = 0;
  for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) s += acc[i][j][r];
  out[tid] = s;
}

template <bool RANDOM>
void run4(int blocks_per_cu, int iters) {
  int* out;
  const int blocks = 256 * blocks_per_cu;
  hipMalloc(&out, blocks * 256 * 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  k4<RANDOM><<<blocks, 256>>>(iters / 10, out);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  k4<RANDOM><<<blocks, 256>>>(iters, out);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  const double ops = (double)blocks * 4 * iters * 32 * 65536.0;
  printf("register-only MFMA, %s operands, %d blocks/CU, %.1f ms run: %.0f TOPS\n", RANDOM ? "random" : "constant", blocks_per_cu, ms,
         ops / (ms * 1e-3) / 1e12);
  hipFree(out);
}

// same question for the other int8 MFMA shape: 16x16x64 (4 accumulator registers, 16 passes of work per 4)
template <bool RANDOM>
__global__ __launch_bounds__(256) void k5(int iters, int* out) {
  const int tid = threadIdx.x + blockIdx.x * 256;
  v4i acc[8];
  for (int i = 0; i < 8; ++i) acc[i] = v4i{0, 0, 0, 0};
  v4i ap[8], bp[8];
  unsigned h = tid * 2654435761u + 12345u;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      h = h * 1664525u + 1013904223u;
      ap[i][q] = RANDOM ? (int)h : 0x01010101;
      h = h * 1664525u + 1013904223u;
      bp[i][q] = RANDOM ? (int)h : 0x01010101;
    }
  }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
#pragma unroll
      for (int j = 0; j < 8; ++j)
        acc[j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(bp[(u + j) & 7], ap[(u + 3 * j) & 7], acc[j], 0, 0, 0);
    }
  }
  int s = 0;
  for (int i = 0; i < 8; ++i) for (int r = 0; r < 4; ++r) s += acc[i][r];
  out[tid] = s;
}

template <bool RANDOM>
void run5(int blocks_per_cu, int iters) {
  int* out;
  const int blocks = 256 * blocks_per_cu;
  hipMalloc(&out, blocks * 256 * 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  k5<RANDOM><<<blocks, 256>>>(iters / 10, out);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  k5<RANDOM><<<blocks, 256>>>(iters, out);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  const double ops = (double)blocks * 4 * iters * 64 * (2.0 * 16 * 16 * 64);
  printf("register-only MFMA 16x16x64, %s operands, %d blocks/CU, %.1f ms run: %.0f TOPS\n", RANDOM ? "random" : "constant", blocks_per_cu, ms,
         ops / (ms * 1e-3) / 1e12);
  hipFree(out);
}

template <int MODE>
void run(const char* name, int blocks_per_cu) {
  int* out;
  unsigned char* gsrc;
  const int blocks = 256 * blocks_per_cu, iters = 2000;
  hipMalloc(&out, blocks * 256 * 4);
  hipMalloc(&gsrc, 256 * 65536 + 65536);
  fill_source(gsrc, 256 * 65536 + 65536);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  k<MODE><<<blocks, 256>>>(200, out, gsrc);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  k<MODE><<<blocks, 256>>>(iters, out, gsrc);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double ops = (double)blocks * 4 * iters * 16 * 65536.0;
  printf("%-28s blocks/CU %d: %.3f ms  %.0f TOPS\n", name, blocks_per_cu, ms, ops / (ms * 1e-3) / 1e12);
  hipFree(out);
  hipFree(gsrc);
}

int main() {
  run4<false>(2, 4000);
  run4<true>(2, 4000);
  run4<false>(2, 40000);
  run4<true>(2, 40000);
  run4<true>(1, 40000);
  run5<false>(2, 20000);
  run5<true>(2, 20000);

  for (int b = 1; b <= 2; ++b) {
    run<0>("mfma only", b);
    run<1>("mfma + lds frag reads", b);
    run<2>("mfma + reads + writes + bar", b);
    run<3>("writes spread after mfma", b);
    run<4>("writes spread before mfma", b);
    run<5>("reads + barrier, no writes", b);
    run<6>("spread writes + frag dbuf", b);
    run<7>("LDS-DMA staging, up front", b);
    run<8>("LDS-DMA staging, spread", b);
    run<9>("B direct from global, A via LDS", b);
    run<10>("B direct, A by LDS-DMA", b);
  }
  run2<0>("LDS structure only");
  run2<1>("global loads + ds_write");
  run2<2>("LDS-DMA");
  run3<4, 4, 2>(1);
  run3<4, 4, 2, true, false>(1);
  run3<4, 4, 2, false, true>(1);
  run3<4, 4, 2, true, true>(1);
  run3<4, 3, 2, true, true>(1);
  run3<4, 4, 1>(1);
  run3<4, 3, 2>(1);
  run3<4, 2, 2>(1);
  run3<4, 2, 1>(2);
  run3<4, 2, 1>(3);
  run3<2, 4, 2>(1);
  run3<2, 4, 1>(2);
  run3<2, 2, 2>(2);
  run3<2, 2, 2>(3);
  run3<2, 2, 1>(3);
  run3<2, 2, 1>(4);
  return 0;
}
