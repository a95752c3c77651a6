// Fused mean-reduced L1 / MSE loss (and gradient) for gfx950, plus savfi_version().
//
// Replaces nn.L1Loss / nn.MSELoss as used by the reference's Loss wrapper (loss.py:287-290,
// :325-350; `hr.clone()` + sub + abs/pow + mean = 3-4 launches and two temporaries).
// HBM streaming of both operands once; float4 per lane; per-wave butterfly + per-block LDS
// reduction and ONE atomic per workgroup into the (pre-zeroed) scalar.
#include "common.h"

namespace {

constexpr int NT = 256;
constexpr int PER_BLOCK = 4096;

template <int KIND>
__device__ __forceinline__ float term(float a, float b) {
  const float d = a - b;
  return KIND == 0 ? fabsf(d) : d * d;
}

template <int KIND>
__global__ __launch_bounds__(NT) void loss_fwd(const float* __restrict__ a, const float* __restrict__ b,
                                               float* __restrict__ result, long long n, float inv_n,
                                               int vec_ok) {
  __shared__ float red[NT / SAVFI_WAVE];
  const long long base = (long long)blockIdx.x * PER_BLOCK;
  const long long end = min(base + (long long)PER_BLOCK, n);
  float acc = 0.f;
  if (vec_ok) {
    const long long vend = base + ((end - base) & ~3LL);
    for (long long e = base + 4 * threadIdx.x; e < vend; e += 4 * NT) {
      const float4 x = *reinterpret_cast<const float4*>(a + e);
      const float4 y = *reinterpret_cast<const float4*>(b + e);
      acc += (term<KIND>(x.x, y.x) + term<KIND>(x.y, y.y)) + (term<KIND>(x.z, y.z) + term<KIND>(x.w, y.w));
    }
    for (long long e = vend + threadIdx.x; e < end; e += NT) acc += term<KIND>(a[e], b[e]);
  } else {
    for (long long e = base + threadIdx.x; e < end; e += NT) acc += term<KIND>(a[e], b[e]);
  }
  const float tot = block_sum<NT / SAVFI_WAVE>(acc, red);
  if (threadIdx.x == 0) atomicAdd(result, tot * inv_n);
}

template <int KIND>
__global__ __launch_bounds__(NT) void loss_bwd(const float* __restrict__ a, const float* __restrict__ b,
                                               const float* __restrict__ g_loss, float* __restrict__ g_a,
                                               long long n, float inv_n) {
  const float gs = g_loss[0] * inv_n;
  for (long long e = (long long)blockIdx.x * NT + threadIdx.x; e < n; e += (long long)gridDim.x * NT) {
    const float d = a[e] - b[e];
    // torch: d|x|/dx = sign(x) with sign(0) = 0
    g_a[e] = KIND == 0 ? gs * (float)((d > 0.f) - (d < 0.f)) : 2.f * gs * d;
  }
}

}  // namespace

extern "C" int savfi_version(void) { return SAVFI_ABI_VERSION; }

extern "C" int savfi_l1_mse_f32(int kind, const float* a, const float* b, float* result, int64_t n,
                                void* stream) {
  if (!a || !b || !result) return SAVFI_E_NULL;
  if (n <= 0) return SAVFI_E_SHAPE;
  if (kind != 0 && kind != 1) return SAVFI_E_UNSUPPORTED;
  const int blocks = savfi_cdiv(n, PER_BLOCK);
  const int vec_ok = (((uintptr_t)a | (uintptr_t)b) & 15u) == 0;
  const float inv_n = (float)(1.0 / (double)n);
  hipStream_t st = (hipStream_t)stream;
  if (kind == 0)
    hipLaunchKernelGGL(loss_fwd<0>, dim3(blocks), dim3(NT), 0, st, a, b, result, (long long)n, inv_n, vec_ok);
  else
    hipLaunchKernelGGL(loss_fwd<1>, dim3(blocks), dim3(NT), 0, st, a, b, result, (long long)n, inv_n, vec_ok);
  return savfi_launch_status();
}

extern "C" int savfi_l1_mse_bwd_f32(int kind, const float* a, const float* b, const float* g_loss,
                                    float* g_a, int64_t n, void* stream) {
  if (!a || !b || !g_loss || !g_a) return SAVFI_E_NULL;
  if (n <= 0) return SAVFI_E_SHAPE;
  if (kind != 0 && kind != 1) return SAVFI_E_UNSUPPORTED;
  const int blocks = (int)((n + NT - 1) / NT < 4096 ? (n + NT - 1) / NT : 4096);
  const float inv_n = (float)(1.0 / (double)n);
  hipStream_t st = (hipStream_t)stream;
  if (kind == 0)
    hipLaunchKernelGGL(loss_bwd<0>, dim3(blocks), dim3(NT), 0, st, a, b, g_loss, g_a, (long long)n, inv_n);
  else
    hipLaunchKernelGGL(loss_bwd<1>, dim3(blocks), dim3(NT), 0, st, a, b, g_loss, g_a, (long long)n, inv_n);
  return savfi_launch_status();
}
