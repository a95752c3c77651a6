// Micro-benchmark: does a VALU instruction between two fp32 MFMAs cost MFMA-pipe time on gfx950?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int KV>
__global__ void k(float* out, int iters) {
  f32x4 acc[8];
  for (int i = 0; i < 8; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
  float v[8];
  for (int i = 0; i < 8; ++i) v[i] = threadIdx.x + i;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < 8; ++m) {
      acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[m], 0, 0, 0);
#pragma unroll
      for (int q = 0; q < KV; ++q) v[(m + q) & 7] = v[(m + q) & 7] * 1.0001f + 0.5f;
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  float s = 0.f;
  for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3] + v[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int KV>
void run(int threads, float* out) {
  const int iters = 2000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<KV>, dim3(256), dim3(threads), 0, 0, out, 10);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<KV>, dim3(256), dim3(threads), 0, 0, out, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double n_mfma = (double)iters * 8;
  printf("threads/WG %4d  VALU per MFMA %d : %.1f ns per MFMA per wave (%.0f us total)\n", threads, KV, ms * 1e6 / n_mfma, ms * 1e3);
}
int main() {
  float* out; hipMalloc(&out, 256 * 1024 * 4);
  for (int threads : {256, 512}) {
    run<0>(threads, out); run<1>(threads, out); run<2>(threads, out); run<4>(threads, out); run<8>(threads, out);
  }
  return 0;
}
