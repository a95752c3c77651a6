// Micro-benchmark: one wave per SIMD issues v_mfma_f32_16x16x32_bf16 back to back (12 per iteration); a SECOND wave of the same SIMD
// issues NV VALU / NS SALU / NL ds_read_b32 / ND ds_write_b32 instructions per iteration.  How much of the second wave's work hides
// under the first wave's matrix-pipe time?     hipcc --offload-arch=gfx950 -O3 tools/mfma_coissue_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));

template <int NV, int NS, int NL, int ND, int MF>
__global__ __launch_bounds__(512) void probe(float* out, int iters) {
  __shared__ __attribute__((aligned(16))) float sm[4096];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 4096; i += blockDim.x) sm[i] = 1.0f + i;
  __syncthreads();
  float s = 0.f;
  if (w < 4) {
    if (MF) {
      bf16x8 a[6], b[3];
      for (int i = 0; i < 6; ++i) a[i] = *reinterpret_cast<const bf16x8*>(reinterpret_cast<char*>(sm) + ((lane + 7 * i) & 255) * 16);
      for (int i = 0; i < 3; ++i) b[i] = *reinterpret_cast<const bf16x8*>(reinterpret_cast<char*>(sm) + ((lane + 11 * i) & 255) * 16 + 4096);
      f32x4 acc[4];
      for (int i = 0; i < 4; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
      __builtin_amdgcn_s_setprio(2);
      for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int q = 0; q < 12; ++q) acc[q & 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[q % 6], b[q % 3], acc[q & 3], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
      for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    }
  } else {
    float v[8]; for (int i = 0; i < 8; ++i) v[i] = i + threadIdx.x;
    int sacc = 0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < NV; ++i) asm volatile("v_fma_f32 %0, %1, %0, %1" : "+v"(v[i & 7]) : "v"(v[(i + 3) & 7]));
#pragma unroll
      for (int i = 0; i < NS; ++i) asm volatile("s_add_u32 %0, %0, 1" : "+s"(sacc));
#pragma unroll
      for (int i = 0; i < NL; ++i) v[i & 7] += sm[(lane + 67 * i + it) & 4095];
#pragma unroll
      for (int i = 0; i < ND; ++i) sm[(lane + 64 * i) & 4095] = v[i & 7];
      __builtin_amdgcn_sched_barrier(0);
    }
    for (int i = 0; i < 8; ++i) s += v[i];
    s += sacc;
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NV, int NS, int NL, int ND, int MF>
void run() {
  float* out;
  hipMalloc(&out, 256 * 512 * 4);
  const int iters = 2000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  probe<NV, NS, NL, ND, MF><<<256, 512>>>(out, iters);
  hipEventRecord(e0);
  probe<NV, NS, NL, ND, MF><<<256, 512>>>(out, iters);
  hipEventRecord(e1); hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("mfma=%d  per 12 MFMAs: VALU %3d SALU %3d ds_read %2d ds_write %2d : %.1f us = %.1f ns per iteration (12 MFMAs alone: ~110 ns)\n", MF, NV, NS, NL, ND, ms * 1e3,
         ms * 1e6 / iters);
  hipFree(out);
}

int main() {
  run<0, 0, 0, 0, 1>();
  run<12, 0, 0, 0, 1>(); run<24, 0, 0, 0, 1>(); run<36, 0, 0, 0, 1>(); run<48, 0, 0, 0, 1>(); run<72, 0, 0, 0, 1>();
  run<24, 0, 0, 0, 0>(); run<48, 0, 0, 0, 0>(); run<72, 0, 0, 0, 0>();
  run<0, 24, 0, 0, 1>(); run<0, 48, 0, 0, 1>();
  run<0, 0, 12, 0, 1>(); run<0, 0, 24, 0, 1>(); run<0, 0, 0, 12, 1>(); run<0, 0, 0, 24, 1>();
  run<24, 24, 6, 6, 1>(); run<24, 24, 6, 6, 0>();
  return 0;
}
