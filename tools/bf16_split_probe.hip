// Probe: fp32 GEMM arithmetic from error-free bf16 splits on the bf16 matrix pipe of gfx950.
//
//   a = a1 + a2 + a3 exactly (three bf16 pieces, round-to-nearest or truncation), same for b;
//   a*b ~= a1b1 + (a1b2 + a2b1) + (a1b3 + a3b1 + a2b2)          ["x6": dropped terms <= 2^-26 |ab| with RNE pieces]
//
// Part 1: numerics of C = A.B (fp32 inputs) against an fp64 host reference, for
//         f32 MFMA (the exact fmaf chain the product kernels use today), bf16 x1 / x3 / x6 / x6 with a separate
//         accumulator for the small terms / x9, RNE and truncating splits, K = 288 .. 4608,
//         uniform operands and a conv-like distribution (ReLU activations x small weights); a delta-operand check
//         (B = one-hot columns must reproduce A's elements bit for bit).
// Part 2: throughput of the six-MFMA group from registers and LDS-fed (16x16x32 and 32x32x16), 1 and 2 waves per SIMD,
//         and the VALU cost of a split.
//
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/bf16_split_probe tools/bf16_split_probe.hip && /tmp/bf16_split_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ unsigned short bf16_rne(float x) {
  unsigned u = __float_as_uint(x);
  return (unsigned short)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}
__device__ __forceinline__ unsigned short bf16_trunc(float x) { return (unsigned short)(__float_as_uint(x) >> 16); }
__device__ __forceinline__ float bf16_up(unsigned short h) { return __uint_as_float(((unsigned)h) << 16); }

template <bool RNE> __device__ __forceinline__ void split3(float a, short& p1, short& p2, short& p3) {
  unsigned short h1 = RNE ? bf16_rne(a) : bf16_trunc(a);
  float r1 = a - bf16_up(h1);
  unsigned short h2 = RNE ? bf16_rne(r1) : bf16_trunc(r1);
  float r2 = r1 - bf16_up(h2);
  unsigned short h3 = RNE ? bf16_rne(r2) : bf16_trunc(r2);
  p1 = (short)h1; p2 = (short)h2; p3 = (short)h3;
}

// ------------------------------------------------------------------------------------------------------------------
// Part 1: one wave per 16x16 tile of C[M][N] = A[M][K] . Bt[N][K]^T
// MODE 0 f32 MFMA | 1 | 3 | 6 (small terms first) | 7 (x6, two accumulators) | 9 | 16 (x6, big terms first)
// ------------------------------------------------------------------------------------------------------------------
template <int MODE, bool RNE>
__global__ __launch_bounds__(64) void gemm_probe(const float* __restrict__ A, const float* __restrict__ Bt, float* __restrict__ C,
                                                 int M, int N, int K) {
  const int l = threadIdx.x, tn = blockIdx.x % (N / 16), tm = blockIdx.x / (N / 16);
  f32x4 acc = {0, 0, 0, 0}, lo = {0, 0, 0, 0};
  if (MODE == 0) {
    const float* ap = A + (size_t)(tm * 16 + (l & 15)) * K + (l >> 4);
    const float* bp = Bt + (size_t)(tn * 16 + (l & 15)) * K + (l >> 4);
    for (int k0 = 0; k0 < K; k0 += 4) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(ap[k0], bp[k0], acc, 0, 0, 0);
  } else {
    const float* ap = A + (size_t)(tm * 16 + (l & 15)) * K + 8 * (l >> 4);
    const float* bp = Bt + (size_t)(tn * 16 + (l & 15)) * K + 8 * (l >> 4);
    for (int k0 = 0; k0 < K; k0 += 32) {
      bf16x8 a1, a2, a3, b1, b2, b3;
      for (int j = 0; j < 8; ++j) {
        short p1, p2, p3;
        split3<RNE>(ap[k0 + j], p1, p2, p3); a1[j] = p1; a2[j] = p2; a3[j] = p3;
        split3<RNE>(bp[k0 + j], p1, p2, p3); b1[j] = p1; b2[j] = p2; b3[j] = p3;
      }
#define MF(x, y, c) c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x, y, c, 0, 0, 0)
      if (MODE == 1) { MF(a1, b1, acc); }
      if (MODE == 3) { MF(a2, b1, acc); MF(a1, b2, acc); MF(a1, b1, acc); }
      if (MODE == 6) { MF(a3, b1, acc); MF(a1, b3, acc); MF(a2, b2, acc); MF(a2, b1, acc); MF(a1, b2, acc); MF(a1, b1, acc); }
      if (MODE == 16) { MF(a1, b1, acc); MF(a1, b2, acc); MF(a2, b1, acc); MF(a2, b2, acc); MF(a1, b3, acc); MF(a3, b1, acc); }
      if (MODE == 7) { MF(a3, b1, lo); MF(a1, b3, lo); MF(a2, b2, lo); MF(a2, b1, lo); MF(a1, b2, lo); MF(a1, b1, acc); }
      if (MODE == 9) { MF(a3, b3, acc); MF(a3, b2, acc); MF(a2, b3, acc); MF(a3, b1, acc); MF(a1, b3, acc); MF(a2, b2, acc);
                       MF(a2, b1, acc); MF(a1, b2, acc); MF(a1, b1, acc); }
    }
    if (MODE == 7) acc += lo;
  }
  for (int r = 0; r < 4; ++r) C[(size_t)(tm * 16 + 4 * (l >> 4) + r) * N + tn * 16 + (l & 15)] = acc[r];
}

struct Err { double max_rel_scale, rms_rel, max_over_sumabs; };

static Err compare(const std::vector<float>& C, const std::vector<double>& R, const std::vector<double>& S) {
  double maxabs = 0, se = 0, sr = 0, maxerr = 0, maxs = 0;
  for (size_t i = 0; i < C.size(); ++i) {
    double e = std::fabs((double)C[i] - R[i]);
    maxerr = std::max(maxerr, e); maxabs = std::max(maxabs, std::fabs(R[i]));
    se += e * e; sr += R[i] * R[i];
    maxs = std::max(maxs, e / S[i]);
  }
  return {maxerr / maxabs, std::sqrt(se / sr), maxs};
}

template <int MODE, bool RNE>
static void run_mode(const char* name, const float* dA, const float* dB, float* dC, int M, int N, int K,
                     const std::vector<double>& R, const std::vector<double>& S) {
  hipLaunchKernelGGL((gemm_probe<MODE, RNE>), dim3((M / 16) * (N / 16)), dim3(64), 0, 0, dA, dB, dC, M, N, K);
  CK(hipDeviceSynchronize());
  std::vector<float> C((size_t)M * N);
  CK(hipMemcpy(C.data(), dC, C.size() * 4, hipMemcpyDeviceToHost));
  Err e = compare(C, R, S);
  printf("    %-34s max|err|/max|C| %.3e   rms err/rms C %.3e   max|err|/sum|a||b| %.3e\n", name, e.max_rel_scale, e.rms_rel, e.max_over_sumabs);
}

static void numerics(int K, int dist) {
  const int M = 64, N = 64;
  std::mt19937 g(1234 + K + dist);
  std::uniform_real_distribution<float> U(-1.f, 1.f);
  std::normal_distribution<float> G(0.f, 1.f);
  std::vector<float> A((size_t)M * K), B((size_t)N * K);
  for (auto& v : A) v = dist == 0 ? U(g) : std::max(0.f, G(g));           // dist 1: ReLU activations
  for (auto& v : B) v = dist == 0 ? U(g) : 0.05f * G(g);                  //         x small weights
  std::vector<double> R((size_t)M * N), S((size_t)M * N);
  for (int i = 0; i < M; ++i)
    for (int j = 0; j < N; ++j) {
      double s = 0, sa = 0;
      for (int k = 0; k < K; ++k) { double p = (double)A[(size_t)i * K + k] * (double)B[(size_t)j * K + k]; s += p; sa += std::fabs(p); }
      R[(size_t)i * N + j] = s; S[(size_t)i * N + j] = sa;
    }
  float *dA, *dB, *dC;
  CK(hipMalloc(&dA, A.size() * 4)); CK(hipMalloc(&dB, B.size() * 4)); CK(hipMalloc(&dC, (size_t)M * N * 4));
  CK(hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice));
  printf("  K = %d, %s\n", K, dist == 0 ? "uniform(-1,1) x uniform(-1,1)" : "relu(N(0,1)) x N(0,0.05)  [conv-like]");
  run_mode<0, true>("f32 MFMA 16x16x4 (fmaf chain)", dA, dB, dC, M, N, K, R, S);
  run_mode<1, true>("bf16 x1 (rne)", dA, dB, dC, M, N, K, R, S);
  run_mode<3, true>("bf16 x3 (rne)", dA, dB, dC, M, N, K, R, S);
  run_mode<6, true>("bf16 x6 (rne, small first)", dA, dB, dC, M, N, K, R, S);
  run_mode<16, true>("bf16 x6 (rne, big first)", dA, dB, dC, M, N, K, R, S);
  run_mode<7, true>("bf16 x6 (rne, 2 accumulators)", dA, dB, dC, M, N, K, R, S);
  run_mode<9, true>("bf16 x9 (rne)", dA, dB, dC, M, N, K, R, S);
  run_mode<6, false>("bf16 x6 (trunc, small first)", dA, dB, dC, M, N, K, R, S);
  run_mode<7, false>("bf16 x6 (trunc, 2 accumulators)", dA, dB, dC, M, N, K, R, S);
  run_mode<9, false>("bf16 x9 (trunc)", dA, dB, dC, M, N, K, R, S);
  CK(hipFree(dA)); CK(hipFree(dB)); CK(hipFree(dC));
}

static void delta_check() {
  // B = one-hot columns (column j picks k = 7*j + 3): C[i][j] must be A[i][7*j+3] bit for bit, tiny / huge / subnormal values included
  const int M = 64, N = 64, K = 512;
  std::mt19937 g(7);
  std::uniform_real_distribution<float> U(-1.f, 1.f);
  std::vector<float> A((size_t)M * K), B((size_t)N * K, 0.f);
  for (size_t i = 0; i < A.size(); ++i) {
    float v = U(g);
    switch (i % 7) { case 1: v *= 1e-30f; break; case 2: v *= 1e30f; break; case 3: v *= 1e-38f; break; case 4: v *= 3e-41f; break; default: break; }
    A[i] = v;
  }
  for (int j = 0; j < N; ++j) B[(size_t)j * K + 7 * j + 3] = 1.f;
  float *dA, *dB, *dC;
  CK(hipMalloc(&dA, A.size() * 4)); CK(hipMalloc(&dB, B.size() * 4)); CK(hipMalloc(&dC, (size_t)M * N * 4));
  CK(hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice));
  std::vector<float> C((size_t)M * N);
  auto count = [&](const char* name) {
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(C.data(), dC, C.size() * 4, hipMemcpyDeviceToHost));
    int bad = 0, bad_normal = 0;
    for (int i = 0; i < M; ++i)
      for (int j = 0; j < N; ++j) {
        float want = A[(size_t)i * K + 7 * j + 3], got = C[(size_t)i * N + j];
        if (memcmp(&want, &got, 4) != 0) { ++bad; if (std::fabs(want) > 1e-30f) ++bad_normal; }
      }
    printf("    %-34s %d of %d elements differ (%d of them with |a| > 1e-30)\n", name, bad, M * N, bad_normal);
  };
  printf("  delta operand (one-hot B), values down to subnormals:\n");
  hipLaunchKernelGGL((gemm_probe<0, true>), dim3(16), dim3(64), 0, 0, dA, dB, dC, M, N, K); count("f32 MFMA");
  hipLaunchKernelGGL((gemm_probe<6, true>), dim3(16), dim3(64), 0, 0, dA, dB, dC, M, N, K); count("bf16 x6 (rne)");
  hipLaunchKernelGGL((gemm_probe<6, false>), dim3(16), dim3(64), 0, 0, dA, dB, dC, M, N, K); count("bf16 x6 (trunc)");
  hipLaunchKernelGGL((gemm_probe<7, true>), dim3(16), dim3(64), 0, 0, dA, dB, dC, M, N, K); count("bf16 x6 (rne, 2 accumulators)");
  CK(hipFree(dA)); CK(hipFree(dB)); CK(hipFree(dC));
}

// ------------------------------------------------------------------------------------------------------------------
// Part 2: throughput
// ------------------------------------------------------------------------------------------------------------------
// SHAPE 0: f32 16x16x4, 16 accumulators; 1: bf16 16x16x32 six-product groups on a 4x4 register tile; 2: bf16 32x32x16 on a 2x2 tile;
// 3: as 1 with the 24 operand fragments of a k-step re-read from LDS (ds_read_b128) every k-step
template <int SHAPE>
__global__ __launch_bounds__(512) void rate_probe(float* out, int iters) {
  __shared__ bf16x8 lds[24 * 64 * 2];
  const int l = threadIdx.x;
  float s = 0;
  if (SHAPE == 0) {
    f32x4 acc[16];
    for (int i = 0; i < 16; ++i) acc[i] = (f32x4){0, 0, 0, 0};
    float a = l * 1e-3f, b = 1.f + l * 1e-4f;
    for (int it = 0; it < iters; ++it)
#pragma unroll
      for (int i = 0; i < 96; ++i) acc[i & 15] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i & 15], 0, 0, 0);
    for (int i = 0; i < 16; ++i) s += acc[i][0];
  } else if (SHAPE == 1 || SHAPE == 3) {
    f32x4 acc[16];
    for (int i = 0; i < 16; ++i) acc[i] = (f32x4){0, 0, 0, 0};
    bf16x8 a[4][3], b[4][3];
    for (int i = 0; i < 4; ++i) for (int p = 0; p < 3; ++p) for (int j = 0; j < 8; ++j) { a[i][p][j] = (short)(0x3c00 + l + i + p + j); b[i][p][j] = (short)(0x3b00 + l + 2 * i + p + j); }
    if (SHAPE == 3) {
      for (int i = 0; i < 24; ++i) lds[(i * 64 + (l & 63)) + 24 * 64 * ((l >> 6) & 1)] = a[i & 3][i % 3];
      __syncthreads();
    }
    const bf16x8* base = lds + 24 * 64 * ((l >> 6) & 1) + (l & 63);
    for (int it = 0; it < iters; ++it) {
      if (SHAPE == 3) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int p = 0; p < 3; ++p) { a[i][p] = base[(i * 3 + p) * 64]; b[i][p] = base[(12 + i * 3 + p) * 64]; }
      }
#pragma unroll
      for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < 4; ++n) {
          f32x4 c = acc[m * 4 + n];
          c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[m][2], b[n][0], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[m][0], b[n][2], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[m][1], b[n][1], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[m][1], b[n][0], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[m][0], b[n][1], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[m][0], b[n][0], c, 0, 0, 0);
          acc[m * 4 + n] = c;
        }
    }
    for (int i = 0; i < 16; ++i) s += acc[i][0];
  } else {
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0;
    bf16x8 a[2][3], b[2][3];
    for (int i = 0; i < 2; ++i) for (int p = 0; p < 3; ++p) for (int j = 0; j < 8; ++j) { a[i][p][j] = (short)(0x3c00 + l + i + p + j); b[i][p][j] = (short)(0x3b00 + l + 2 * i + p + j); }
    for (int it = 0; it < iters; ++it)
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n) {
          f32x16 c = acc[m * 2 + n];
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[m][2], b[n][0], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[m][0], b[n][2], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[m][1], b[n][1], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[m][1], b[n][0], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[m][0], b[n][1], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[m][0], b[n][0], c, 0, 0, 0);
          acc[m * 2 + n] = c;
        }
    for (int i = 0; i < 4; ++i) s += acc[i][0];
  }
  out[blockIdx.x * blockDim.x + l] = s;
}

// VALU cost of splitting: 8 floats -> 3 x (8 bf16) per iteration and lane, v_cvt_pk_bf16_f32 for the rounding
__device__ __forceinline__ unsigned cvt_pk(float lo, float hi) { unsigned r; asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi)); return r; }
__global__ __launch_bounds__(256) void split_probe(float* out, int iters, long long* clk) {
  float x[8];
  for (int j = 0; j < 8; ++j) x[j] = 1.f + threadIdx.x * 1e-3f + j;
  unsigned acc = 0;
  const long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < 8; j += 2) {
      unsigned p1 = cvt_pk(x[j], x[j + 1]);
      float r0 = x[j] - __uint_as_float(p1 << 16), r1 = x[j + 1] - __uint_as_float(p1 & 0xffff0000u);
      unsigned p2 = cvt_pk(r0, r1);
      float q0 = r0 - __uint_as_float(p2 << 16), q1 = r1 - __uint_as_float(p2 & 0xffff0000u);
      unsigned p3 = cvt_pk(q0, q1);
      acc ^= p1 + p2 * 3 + p3 * 5;
      x[j] += 1.f; x[j + 1] += 1.f;
    }
  }
  const long long t1 = clock64();
  out[blockIdx.x * blockDim.x + threadIdx.x] = __uint_as_float(acc & 0x3fffffff);
  if (threadIdx.x == 0 && blockIdx.x == 0) clk[0] = t1 - t0;
}

template <int SHAPE> static void rate(const char* name, int threads, double flop_per_iter_per_wave, float* out) {
  const int iters = 2000, blocks = 256 * 2;
  hipLaunchKernelGGL(rate_probe<SHAPE>, dim3(blocks), dim3(threads), 0, 0, out, 10);
  CK(hipDeviceSynchronize());
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  hipLaunchKernelGGL(rate_probe<SHAPE>, dim3(blocks), dim3(threads), 0, 0, out, iters);
  hipEventRecord(e1); CK(hipEventSynchronize(e1));
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double flops = flop_per_iter_per_wave * iters * blocks * (threads / 64);
  printf("    %-44s %d threads/WG x %d WGs: %8.1f us   %7.1f TFLOP/s raw   %6.1f fp32-equivalent (raw / 6 for the split forms)\n", name, threads, blocks,
         ms * 1e3, flops / (ms * 1e-3) / 1e12, flops / (ms * 1e-3) / 1e12 / (SHAPE == 0 ? 1 : 6));
}

int main() {
  printf("== Part 1: numerics vs fp64 ==\n");
  for (int dist = 0; dist < 2; ++dist)
    for (int K : {288, 576, 2304, 4608}) numerics(K, dist);
  delta_check();
  printf("== Part 2: throughput ==\n");
  float* out; long long* clk;
  CK(hipMalloc(&out, 4 << 20)); CK(hipMalloc(&clk, 16));
  for (int threads : {256, 512}) {
    rate<0>("f32 16x16x4, 16 accumulators", threads, 96.0 * 2 * 16 * 16 * 4, out);
    rate<1>("bf16 16x16x32 x6, 4x4 tile, registers", threads, 96.0 * 2 * 16 * 16 * 32, out);
    rate<3>("bf16 16x16x32 x6, 4x4 tile, 24 ds_read_b128/k-step", threads, 96.0 * 2 * 16 * 16 * 32, out);
    rate<2>("bf16 32x32x16 x6, 2x2 tile, registers", threads, 24.0 * 2 * 32 * 32 * 16, out);
  }
  const int iters = 4000;
  hipLaunchKernelGGL(split_probe, dim3(256), dim3(256), 0, 0, out, iters, clk);
  CK(hipDeviceSynchronize());
  long long h; CK(hipMemcpy(&h, clk, 8, hipMemcpyDeviceToHost));
  printf("    split 8 floats -> 3 x 8 bf16 (v_cvt_pk_bf16_f32 + shift/and + sub): %.1f shader cycles per iteration of one wave (1 wave/SIMD) = %.2f per element\n",
         (double)h / iters, (double)h / iters / 8);
  return 0;
}
