// Shared helpers for the savfi HIP kernels (gfx950 only; wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "savfi_hip.h"

#define SAVFI_WAVE 64

// A/B knobs of the kernels are COMPILE-TIME macros (tools/build_variant.sh NAME unit.hip -DSAVFI_...=v builds a variant library for
// SAVFI_HIP_LIB): the shipped library reads no environment variable (tests/test_host_cpu.py greps for strays).

static inline int savfi_launch_status() {
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? SAVFI_OK : (int)e;
}

static inline int savfi_cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

// Opt-in to > 64 KB of dynamic LDS is a per-device function attribute: set once per (kernel, device); `done_mask` is the
// caller's static bit set of devices already configured.
static inline int savfi_ensure_dynamic_lds(const void* kernel, size_t bytes, uint32_t& done_mask) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return (int)hipGetLastError();
  const uint32_t bit = 1u << (dev & 31);
  if (__atomic_load_n(&done_mask, __ATOMIC_ACQUIRE) & bit) return SAVFI_OK;
  const hipError_t e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
  if (e != hipSuccess) return (int)e;
  __atomic_fetch_or(&done_mask, bit, __ATOMIC_RELEASE);
  return SAVFI_OK;
}

// 64-lane butterfly sum; every lane ends with the total.
__device__ __forceinline__ float wave_sum(float x) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) x += __shfl_xor(x, off, SAVFI_WAVE);
  return x;
}

// Block-wide sum for blocks of NW waves; result valid in thread 0 (and wave 0 lanes).
template <int NW>
__device__ __forceinline__ float block_sum(float x, float* lds /* >= NW floats */) {
  x = wave_sum(x);
  const int lane = threadIdx.x & (SAVFI_WAVE - 1), wid = threadIdx.x / SAVFI_WAVE;
  if (lane == 0) lds[wid] = x;
  __syncthreads();
  float t = (threadIdx.x < NW) ? lds[threadIdx.x] : 0.f;
  if (wid == 0) t = wave_sum(t);
  __syncthreads();
  return t;
}

// csrc/sepconv_x6.hip: gV and gH of the K = 51, C = 3 separable convolution on split-bf16 MFMAs (internal: reached through
// savfi_sepconv_bwd_f32)
int savfi_sepconv_bwd_x6_launch(const float* in, const float* v, const float* h, const float* gO, float* gV, float* gH, int B, int Ho,
                                int Wo, int cus, hipStream_t st);
int savfi_sepconv_fwd_x6_launch(const float* in, const float* v, const float* h, float* out, int B, int Ho, int Wo, int cus,
                                hipStream_t st);
// csrc/sepconv_ws.hip: the same gradients with MFMA waves and staging waves in pairs (widths that are a multiple of 4)
// TB = tap planes between two samples of v / h / gV / gH (51: contiguous tensors)
// in2 / cls2 != nullptr: the pair launch -- B virtual samples 2 b + f, frame f of sample b (`in` / `in2`), see the kernel
int savfi_sepconv_bwd_ws_launch(const float* in, const float* v, const float* h, const float* gO, float* gV, float* gH, int B, int Ho,
                                int Wo, int cus, int TB, const unsigned* cls, int taps_unit16, hipStream_t st, const float* in2 = nullptr,
                                const unsigned* cls2 = nullptr);
int savfi_sepconv_fwd_ws_launch(const float* in, const float* v, const float* h, float* out, int B, int Ho, int Wo, int cus, int TB,
                                const unsigned* cls, int taps_unit16, hipStream_t st, const float* in2 = nullptr,
                                const unsigned* cls2 = nullptr);
