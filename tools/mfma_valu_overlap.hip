// Micro-benchmark: how fp32 MFMA, VALU and LDS instructions of one wave / of several waves of a SIMD overlap on gfx950.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define SB __builtin_amdgcn_sched_barrier(0)
#define VADD(x, y) asm volatile("v_add_f32 %0, %1, %0" : "+v"(x) : "v"(y))
#define VADDI(x, y) asm volatile("v_add_u32 %0, %1, %0" : "+v"(x) : "v"(y))
#define VPK(x, y) asm volatile("v_pk_add_f32 %0, %1, %0" : "+v"(x) : "v"(y))
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(1024) void k(float* out, int iters, long long* clk) {
  __shared__ float lds[4096];
  f32x4 acc[16];
  for (int i = 0; i < 16; ++i) acc[i] = (f32x4){0, 0, 0, 0};
  float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
  float v[8]; for (int i = 0; i < 8; ++i) v[i] = i + threadIdx.x;
  int vi[8]; for (int i = 0; i < 8; ++i) vi[i] = i + threadIdx.x;
  f32x2 p[8]; for (int i = 0; i < 8; ++i) p[i] = (f32x2){(float)i, (float)threadIdx.x};
  f32x2 pb = (f32x2){1.f, 2.f};
  lds[threadIdx.x] = a; lds[threadIdx.x + 1024] = b;
  __syncthreads();
  const long long t0 = clock64();
  const long long w0 = wall_clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 32; ++i) {
      if (MODE != 3) { acc[i & 15] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i & 15], 0, 0, 0); SB; }
      if (MODE == 1 || MODE == 3) { VADD(v[i & 7], b); SB; }
      if (MODE == 5) { VADD(v[i & 7], b); SB; VADD(v[(i + 4) & 7], b); SB; }
      if (MODE == 4 && (i & 3) == 3) { VADD(v[0], b); VADD(v[1], b); VADD(v[2], b); VADD(v[3], b); SB; }
      if (MODE == 6 && (i & 1)) { VPK(p[(i >> 1) & 7], pb); SB; }
      if (MODE == 7 && (i & 3) == 3) { lds[threadIdx.x + 64 * (i >> 2)] = v[i >> 2]; SB; }
      if (MODE == 8 && (i & 3) == 3) { v[i >> 2] += lds[threadIdx.x + 64 * (i >> 2)]; SB; }
      if (MODE == 11) { VADDI(vi[i & 7], vi[(i + 1) & 7]); SB; }
      if (MODE == 12 && (i & 7) == 7) { VADD(v[0], b); VADD(v[1], b); VADD(v[2], b); VADD(v[3], b); VADD(v[4], b); VADD(v[5], b); VADD(v[6], b); VADD(v[7], b); SB; }
    }
    if (MODE == 2) {
#pragma unroll
      for (int i = 0; i < 32; ++i) { VADD(v[i & 7], b); SB; }
    }
    if (MODE == 9) __syncthreads();
  }
  const long long t1 = clock64();
  const long long w1 = wall_clock64();
  float s = 0;
  for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][3];
  for (int i = 0; i < 8; ++i) s += v[i] + vi[i] + p[i][0] + p[i][1];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = t1 - t0; clk[1] = w1 - w0; }
}

template <int MODE> void run(int threads, int iters, float* out, long long* clk) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(threads), 0, 0, out, 10, clk);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(threads), 0, 0, out, iters, clk);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  long long h[2]; hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
  const double wall_s = h[1] / 100e6;   // wall_clock64: 100 MHz
  printf("mode %2d waves/SIMD %d: %.1f us, %.1f shader-clk/iter/wave-slot, shader MHz %.0f, per-SIMD cycles per iteration of ALL its waves %.0f\n", MODE, threads / 256,
         ms * 1e3, (double)h[0] / iters, h[0] / wall_s / 1e6, (double)h[0] / iters);
}

int main() {
  float* out; long long* clk;
  hipMalloc(&out, 256 * 1024 * 4); hipMalloc(&clk, 16);
  const int iters = 2000;
  for (int threads : {256, 512, 768, 1024}) {
    run<0>(threads, iters, out, clk); run<1>(threads, iters, out, clk); run<2>(threads, iters, out, clk); run<3>(threads, iters, out, clk);
    run<4>(threads, iters, out, clk); run<5>(threads, iters, out, clk); run<6>(threads, iters, out, clk); run<7>(threads, iters, out, clk);
    run<8>(threads, iters, out, clk); run<9>(threads, iters, out, clk); run<11>(threads, iters, out, clk); run<12>(threads, iters, out, clk);
  }
  return 0;
}
