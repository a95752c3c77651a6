// Micro-benchmark: v_mfma_f32_16x16x32_bf16 issue rate of ONE wave per SIMD as a function of how many independent accumulators the
// instruction stream rotates through (distance between two MFMAs on the same accumulator), with and without ds_read_b128 traffic
// between groups of 12 MFMAs (the shape of the SepConv MFMA loops).   hipcc --offload-arch=gfx950 -O3 tools/mfma_chain_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));

template <int NACC, int LDS, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void probe(float* out, long long* cyc, int iters) {
  __shared__ __attribute__((aligned(16))) char sm[16384];
  const int lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < 4096; i += blockDim.x) reinterpret_cast<float*>(sm)[i] = 1.0f + i;
  __syncthreads();
  bf16x8 a[6], b[3];
  for (int i = 0; i < 6; ++i) a[i] = *reinterpret_cast<const bf16x8*>(sm + ((lane + 7 * i) & 255) * 16);
  for (int i = 0; i < 3; ++i) b[i] = *reinterpret_cast<const bf16x8*>(sm + ((lane + 11 * i) & 255) * 16 + 4096);
  f32x4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    if (LDS) {
#pragma unroll
      for (int i = 0; i < 6; ++i) a[i] = *reinterpret_cast<const bf16x8*>(sm + ((lane + 7 * i + it) & 255) * 16 + 1024 * i);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int q = 0; q < 12; ++q) {
      acc[q % NACC] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[q % 6], b[q % 3], acc[q % NACC], 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  const long long t1 = __builtin_readcyclecounter();
  float s = 0.f;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int NACC, int LDS, int WAVES>
void run(const char* name) {
  float* out; long long* cyc;
  hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 8);
  const int iters = 2000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  probe<NACC, LDS, WAVES><<<256, 64 * WAVES>>>(out, cyc, iters);
  hipEventRecord(e0);
  probe<NACC, LDS, WAVES><<<256, 64 * WAVES>>>(out, cyc, iters);
  hipEventRecord(e1); hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
  // s_memtime / readcyclecounter ticks at 100 MHz on this part? report both
  printf("%-28s NACC=%2d LDS=%d waves/CU=%d: %.1f us, %.2f ns per MFMA per wave, counter ticks per MFMA %.2f\n", name, NACC, LDS, WAVES, ms * 1e3,
         ms * 1e6 / (iters * 12.0), (double)c / (iters * 12.0));
  hipFree(out); hipFree(cyc);
}

int main() {
  run<1, 0, 4>("chain 1");
  run<2, 0, 4>("chain 2");
  run<3, 0, 4>("chain 3");
  run<4, 0, 4>("chain 4");
  run<6, 0, 4>("chain 6");
  run<12, 0, 4>("chain 12");
  run<2, 1, 4>("chain 2 + 6 ds_read_b128");
  run<4, 1, 4>("chain 4 + 6 ds_read_b128");
  run<12, 1, 4>("chain 12 + 6 ds_read_b128");
  run<2, 0, 8>("chain 2, 2 waves/SIMD");
  run<4, 0, 8>("chain 4, 2 waves/SIMD");
  run<2, 1, 8>("chain 2 + lds, 2 waves/SIMD");
  return 0;
}
