// Micro-benchmark: 32 independent fp32 MFMA accumulators (128 VGPRs), operands from 8 A / 8 B registers
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int NACC, int KV, int SB>
__global__ __launch_bounds__(256, 2) void k(float* out, int iters) {
  f32x4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float a[8], b[8], v[8];
  for (int i = 0; i < 8; ++i) { a[i] = threadIdx.x * 1e-3f + i; b[i] = 1.0f + threadIdx.x * 1e-4f * i; v[i] = i; }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < NACC; ++m) {
      acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[(m >> 2) & 7], b[m & 7], acc[m], 0, 0, 0);
#pragma unroll
      for (int q = 0; q < KV; ++q) v[(m + q) & 7] = v[(m + q) & 7] * 1.0001f + 0.5f;
      if (SB && (m & 3) == 3) __builtin_amdgcn_sched_barrier(0);
    }
  }
  float s = 0.f;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  for (int i = 0; i < 8; ++i) s += v[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NACC, int KV, int SB>
void run(int blocks, float* out) {
  const int iters = 2000 * 8 / NACC;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<NACC, KV, SB>), dim3(blocks), dim3(256), 0, 0, out, 10);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<NACC, KV, SB>), dim3(blocks), dim3(256), 0, 0, out, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("blocks %4d  acc %2d  VALU/MFMA %d  sched_barrier %d : %.1f ns per MFMA per wave\n", blocks, NACC, KV, SB, ms * 1e6 / ((double)iters * NACC));
}
int main() {
  float* out; hipMalloc(&out, 1024 * 256 * 4);
  for (int blocks : {256, 512}) {
    run<8, 0, 0>(blocks, out); run<32, 0, 0>(blocks, out); run<32, 1, 0>(blocks, out); run<32, 1, 1>(blocks, out); run<32, 2, 1>(blocks, out);
  }
  return 0;
}
