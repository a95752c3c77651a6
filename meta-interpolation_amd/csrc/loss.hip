// Fused mean-reduced L1 / MSE loss (and gradient) for gfx950, plus savfi_version().
//
// Replaces nn.L1Loss / nn.MSELoss as used by the reference's Loss wrapper (loss.py:287-290,
// :325-350; `hr.clone()` + sub + abs/pow + mean = 3-4 launches and two temporaries).
// `rows` independent reductions per launch (one per sample: the tasks of a meta-batch in lockstep, the two support
// triplets of a step).  HBM streaming of both operands once; float4 per lane; per-wave butterfly + per-block LDS
// reduction into per-block partial sums, added in a fixed order by a second tiny kernel (deterministic).
#include "common.h"

namespace {

constexpr int NT = 256;
constexpr int PER_BLOCK = 4096;

template <int KIND>
__device__ __forceinline__ float term(float a, float b) {
  const float d = a - b;
  return KIND == 0 ? fabsf(d) : d * d;
}

// partial[row * blocks + blk] = sum over the block's chunk of row `row`; no atomics: loss_finish adds the partial sums of a
// row in a fixed order, so the loss value is bit-reproducible run to run (round-1 advisor finding).
template <int KIND>
__global__ __launch_bounds__(NT) void loss_fwd(const float* __restrict__ a, const float* __restrict__ b,
                                               float* __restrict__ partial, long long n, int vec_ok) {
  __shared__ float red[NT / SAVFI_WAVE];
  const long long row0 = (long long)blockIdx.y * n;
  const long long base = (long long)blockIdx.x * PER_BLOCK;
  const long long end = min(base + (long long)PER_BLOCK, n);
  const float* ar = a + row0;
  const float* br = b + row0;
  float acc = 0.f;
  if (vec_ok) {
    const long long vend = base + ((end - base) & ~3LL);
    for (long long e = base + 4 * threadIdx.x; e < vend; e += 4 * NT) {
      const float4 x = *reinterpret_cast<const float4*>(ar + e);
      const float4 y = *reinterpret_cast<const float4*>(br + e);
      acc += (term<KIND>(x.x, y.x) + term<KIND>(x.y, y.y)) + (term<KIND>(x.z, y.z) + term<KIND>(x.w, y.w));
    }
    for (long long e = vend + threadIdx.x; e < end; e += NT) acc += term<KIND>(ar[e], br[e]);
  } else {
    for (long long e = base + threadIdx.x; e < end; e += NT) acc += term<KIND>(ar[e], br[e]);
  }
  const float tot = block_sum<NT / SAVFI_WAVE>(acc, red);
  if (threadIdx.x == 0) partial[(size_t)blockIdx.y * gridDim.x + blockIdx.x] = tot;
}

// result[row] = (sum of the row's partial sums, lane-strided then a butterfly: always the same order) / n
__global__ __launch_bounds__(64) void loss_finish(const float* __restrict__ partial, float* __restrict__ result, int blocks,
                                                  float inv_n) {
  const float* p = partial + (size_t)blockIdx.x * blocks;
  float acc = 0.f;
  for (int i = threadIdx.x; i < blocks; i += 64) acc += p[i];
  acc = wave_sum(acc);
  if (threadIdx.x == 0) result[blockIdx.x] = acc * inv_n;
}

template <int KIND>
__global__ __launch_bounds__(NT) void loss_bwd(const float* __restrict__ a, const float* __restrict__ b,
                                               const float* __restrict__ g_loss, float* __restrict__ g_a,
                                               long long n, float inv_n) {
  const float gs = g_loss[blockIdx.y] * inv_n;
  const long long row0 = (long long)blockIdx.y * n;
  for (long long e = (long long)blockIdx.x * NT + threadIdx.x; e < n; e += (long long)gridDim.x * NT) {
    const float d = a[row0 + e] - b[row0 + e];
    // torch: d|x|/dx = sign(x) with sign(0) = 0
    g_a[row0 + e] = KIND == 0 ? gs * (float)((d > 0.f) - (d < 0.f)) : 2.f * gs * d;
  }
}

}  // namespace

extern "C" int savfi_version(void) { return SAVFI_ABI_VERSION; }

extern "C" int64_t savfi_l1_mse_scratch_floats(int rows, int64_t n) {
  if (rows <= 0 || n <= 0) return SAVFI_E_SHAPE;
  return (int64_t)rows * savfi_cdiv(n, PER_BLOCK);
}

extern "C" int savfi_l1_mse_f32(int kind, const float* a, const float* b, float* result, float* scratch, int rows, int64_t n,
                                void* stream) {
  if (!a || !b || !result || !scratch) return SAVFI_E_NULL;
  if (n <= 0 || rows <= 0 || rows > 65535) return SAVFI_E_SHAPE;
  if (kind != 0 && kind != 1) return SAVFI_E_UNSUPPORTED;
  const int blocks = savfi_cdiv(n, PER_BLOCK);
  const int vec_ok = ((((uintptr_t)a | (uintptr_t)b) & 15u) == 0) && (rows == 1 || n % 4 == 0);
  const float inv_n = (float)(1.0 / (double)n);
  hipStream_t st = (hipStream_t)stream;
  if (kind == 0)
    hipLaunchKernelGGL(loss_fwd<0>, dim3(blocks, rows), dim3(NT), 0, st, a, b, scratch, (long long)n, vec_ok);
  else
    hipLaunchKernelGGL(loss_fwd<1>, dim3(blocks, rows), dim3(NT), 0, st, a, b, scratch, (long long)n, vec_ok);
  if (int e = savfi_launch_status()) return e;
  hipLaunchKernelGGL(loss_finish, dim3(rows), dim3(64), 0, st, scratch, result, blocks, inv_n);
  return savfi_launch_status();
}

extern "C" int savfi_l1_mse_bwd_f32(int kind, const float* a, const float* b, const float* g_loss,
                                    float* g_a, int rows, int64_t n, void* stream) {
  if (!a || !b || !g_loss || !g_a) return SAVFI_E_NULL;
  if (n <= 0 || rows <= 0 || rows > 65535) return SAVFI_E_SHAPE;
  if (kind != 0 && kind != 1) return SAVFI_E_UNSUPPORTED;
  const int blocks = (int)((n + NT - 1) / NT < 4096 ? (n + NT - 1) / NT : 4096);
  const float inv_n = (float)(1.0 / (double)n);
  hipStream_t st = (hipStream_t)stream;
  if (kind == 0)
    hipLaunchKernelGGL(loss_bwd<0>, dim3(blocks, rows), dim3(NT), 0, st, a, b, g_loss, g_a, (long long)n, inv_n);
  else
    hipLaunchKernelGGL(loss_bwd<1>, dim3(blocks, rows), dim3(NT), 0, st, a, b, g_loss, g_a, (long long)n, inv_n);
  return savfi_launch_status();
}
