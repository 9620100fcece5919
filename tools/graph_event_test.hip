// Probe: do hipEventRecord calls captured into a hipGraph give usable per-kernel timings on replay?
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ void spin(float* p, int n) {
  float v = p[threadIdx.x];
  for (int i = 0; i < n; ++i) v = v * 1.0001f + 0.5f;
  p[threadIdx.x] = v;
}
int main() {
  float* d; CK(hipMalloc(&d, 4096));
  hipStream_t s; CK(hipStreamCreate(&s));
  hipEvent_t e0, e1, e2; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&e2));
  spin<<<1, 64, 0, s>>>(d, 1000); CK(hipStreamSynchronize(s));
  hipGraph_t g; hipGraphExec_t ge;
  CK(hipStreamBeginCapture(s, hipStreamCaptureModeRelaxed));
  CK(hipEventRecord(e0, s));
  spin<<<1, 64, 0, s>>>(d, 200000);
  CK(hipEventRecord(e1, s));
  spin<<<1, 64, 0, s>>>(d, 400000);
  CK(hipEventRecord(e2, s));
  CK(hipStreamEndCapture(s, &g));
  CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  for (int r = 0; r < 3; ++r) {
    CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s));
    float a = -1, b = -1;
    hipError_t ea = hipEventElapsedTime(&a, e0, e1), eb = hipEventElapsedTime(&b, e1, e2);
    printf("replay %d: k1 %.3f ms (%s), k2 %.3f ms (%s)\n", r, a, hipGetErrorString(ea), b, hipGetErrorString(eb));
  }
  // eager reference
  CK(hipEventRecord(e0, s)); spin<<<1, 64, 0, s>>>(d, 200000); CK(hipEventRecord(e1, s));
  spin<<<1, 64, 0, s>>>(d, 400000); CK(hipEventRecord(e2, s)); CK(hipStreamSynchronize(s));
  float a, b; CK(hipEventElapsedTime(&a, e0, e1)); CK(hipEventElapsedTime(&b, e1, e2));
  printf("eager: k1 %.3f ms, k2 %.3f ms\n", a, b);
  return 0;
}
