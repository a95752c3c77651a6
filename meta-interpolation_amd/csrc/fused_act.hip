// Convolution epilogues for gfx950: bias + (leaky-)ReLU forward in place, and activation-backward fused
// with the bias gradient.
//
// The reference's backbones apply `conv -> + bias -> ReLU` as three kernels forward (MIOpen conv,
// elementwise add, clamp) and `threshold_backward` + a per-channel `sum` reduction backward
// (sepconv/model.py:172-194 Basic/Subnet blocks; model_utils.py:957-990 RCAB with LeakyReLU(0.2)).
// At 384x512 those elementwise passes are ~13 % of the inner step's GPU time.  Here:
//   fwd:  y = act(z + b[c])            in place on the conv output, one read + one write
//   bwd:  gz = gy * act'(y),  gb[c] = sum gz       one pass: two reads, one write, one partial sum / workgroup,
//         then a fixed-order sum of the partials (deterministic: no atomics)
// act(x) = x > 0 ? x : slope * x   (slope 0 = ReLU, 0.2 = CAIN's LeakyReLU, 1 = bias only).
// Pure HBM streaming: one workgroup per 4096-element chunk of an (n, c) plane, float4 per lane.
#include <cstdlib>

#include "common.h"

namespace {

constexpr int NT = 256, CHUNK = 4096;

__device__ __forceinline__ float act(float x, float slope) { return x > 0.f ? x : slope * x; }

// Planes whose size is not a multiple of 4 floats (SepConv's windowed tail: 137 x 233) start at any 4-byte phase: every chunk
// peels 0-3 leading elements to reach a 16-byte boundary, streams float4s, and finishes with the 0-3 that are left.  `phase_ok`: all
// operand pointers share their low four address bits (the host checks), so one peel aligns them all; otherwise scalar.
__device__ __forceinline__ int peel_to_16(const void* p, int count) {
  const int head = (int)((16u - ((unsigned)(uintptr_t)p & 15u)) & 15u) >> 2;
  return head < count ? head : count;
}

__global__ __launch_bounds__(NT) void bias_act_fwd(float* __restrict__ z, const float* __restrict__ bias, int C, int HW,
                                                   int chunks, float slope, int phase_ok) {
  const int plane = blockIdx.x / chunks, ch = blockIdx.x - plane * chunks;
  const float b = bias[plane % C];
  float* p = z + (size_t)plane * HW;
  const int base = ch * CHUNK, end = min(base + CHUNK, HW);
  if (phase_ok) {
    const int vbeg = base + peel_to_16(p + base, end - base);
    const int vend = vbeg + ((end - vbeg) & ~3);
    if (base + (int)threadIdx.x < vbeg) p[base + threadIdx.x] = act(p[base + threadIdx.x] + b, slope);
    for (int e = vbeg + 4 * threadIdx.x; e < vend; e += 4 * NT) {
      float4 v = *reinterpret_cast<float4*>(p + e);
      v.x = act(v.x + b, slope); v.y = act(v.y + b, slope); v.z = act(v.z + b, slope); v.w = act(v.w + b, slope);
      *reinterpret_cast<float4*>(p + e) = v;
    }
    if (vend + (int)threadIdx.x < end) p[vend + threadIdx.x] = act(p[vend + threadIdx.x] + b, slope);
  } else {
    for (int e = base + threadIdx.x; e < end; e += NT) p[e] = act(p[e] + b, slope);
  }
}

__global__ __launch_bounds__(NT) void bias_act_bwd(const float* __restrict__ gy, const float* __restrict__ y,
                                                   float* __restrict__ gz, float* __restrict__ partial, int C, int HW,
                                                   int chunks, float slope, int phase_ok) {
  __shared__ float red[NT / SAVFI_WAVE];
  const int plane = blockIdx.x / chunks, ch = blockIdx.x - plane * chunks;
  const size_t off = (size_t)plane * HW;
  const int base = ch * CHUNK, end = min(base + CHUNK, HW);
  float acc = 0.f;
  auto d = [slope](float g, float out) { return out > 0.f ? g : slope * g; };
  if (phase_ok) {
    const int vbeg = base + peel_to_16(gy + off + base, end - base);
    const int vend = vbeg + ((end - vbeg) & ~3);
    if (base + (int)threadIdx.x < vbeg) {
      const int e = base + threadIdx.x;
      const float r = d(gy[off + e], y[off + e]);
      if (gz) gz[off + e] = r;
      acc += r;
    }
    for (int e = vbeg + 4 * threadIdx.x; e < vend; e += 4 * NT) {
      const float4 g = *reinterpret_cast<const float4*>(gy + off + e);
      const float4 o = *reinterpret_cast<const float4*>(y + off + e);
      const float4 r = make_float4(d(g.x, o.x), d(g.y, o.y), d(g.z, o.z), d(g.w, o.w));
      if (gz) *reinterpret_cast<float4*>(gz + off + e) = r;
      acc += (r.x + r.y) + (r.z + r.w);
    }
    if (vend + (int)threadIdx.x < end) {
      const int e = vend + threadIdx.x;
      const float r = d(gy[off + e], y[off + e]);
      if (gz) gz[off + e] = r;
      acc += r;
    }
  } else {
    for (int e = base + threadIdx.x; e < end; e += NT) { const float r = d(gy[off + e], y[off + e]); if (gz) gz[off + e] = r; acc += r; }
  }
  if (partial) {     // one partial sum per (plane, chunk); reduced in a fixed order by bias_grad_finish (deterministic)
    const float tot = block_sum<NT / SAVFI_WAVE>(acc, red);
    if (threadIdx.x == 0) partial[blockIdx.x] = tot;
  }
}

// Small maps (N * HW <= CHUNK, one chunk per plane: CAIN's 16 x 16 maps): ONE launch.  Workgroup c walks the N planes of channel c
// with the element-to-thread assignment of bias_act_bwd, takes each plane's block sum and adds them in n order in thread 0 --
// exactly the additions of bias_act_bwd + bias_grad_finish (one partial per plane, summed by lane 0): bit-identical, one launch
// less per layer (982 of them in 200 ms of config C1).
__global__ __launch_bounds__(NT) void bias_act_bwd_small(const float* __restrict__ gy, const float* __restrict__ y,
                                                         float* __restrict__ gz, float* __restrict__ gbias, int N, int C, int HW,
                                                         float slope, int phase_ok) {
  __shared__ float red[NT / SAVFI_WAVE];
  const int c = blockIdx.x;
  auto d = [slope](float g, float out) { return out > 0.f ? g : slope * g; };
  float total = 0.f;
  for (int n = 0; n < N; ++n) {
    const size_t off = ((size_t)n * C + c) * HW;
    const int base = 0, end = HW;
    float acc = 0.f;
    if (phase_ok) {
      const int vbeg = base + peel_to_16(gy + off + base, end - base);
      const int vend = vbeg + ((end - vbeg) & ~3);
      if (base + (int)threadIdx.x < vbeg) {
        const int e = base + threadIdx.x;
        const float r = d(gy[off + e], y[off + e]);
        if (gz) gz[off + e] = r;
        acc += r;
      }
      for (int e = vbeg + 4 * threadIdx.x; e < vend; e += 4 * NT) {
        const float4 g = *reinterpret_cast<const float4*>(gy + off + e);
        const float4 o = *reinterpret_cast<const float4*>(y + off + e);
        const float4 r = make_float4(d(g.x, o.x), d(g.y, o.y), d(g.z, o.z), d(g.w, o.w));
        if (gz) *reinterpret_cast<float4*>(gz + off + e) = r;
        acc += (r.x + r.y) + (r.z + r.w);
      }
      if (vend + (int)threadIdx.x < end) {
        const int e = vend + threadIdx.x;
        const float r = d(gy[off + e], y[off + e]);
        if (gz) gz[off + e] = r;
        acc += r;
      }
    } else {
      for (int e = base + threadIdx.x; e < end; e += NT) { const float r = d(gy[off + e], y[off + e]); if (gz) gz[off + e] = r; acc += r; }
    }
    const float tot = block_sum<NT / SAVFI_WAVE>(acc, red);
    if (threadIdx.x == 0) total += tot;
  }
  if (threadIdx.x == 0) gbias[c] = total;
}

// gbias[c] = sum over n and chunks of partial[(n * C + c) * chunks + ch], always in the same order
__global__ __launch_bounds__(64) void bias_grad_finish(const float* __restrict__ partial, float* __restrict__ gbias, int N, int C,
                                                       int chunks) {
  const int c = blockIdx.x;
  float acc = 0.f;
  for (int n = 0; n < N; ++n) {
    const float* p = partial + ((size_t)n * C + c) * chunks;
    for (int i = threadIdx.x; i < chunks; i += 64) acc += p[i];
  }
  acc = wave_sum(acc);
  if (threadIdx.x == 0) gbias[c] = acc;
}

}  // namespace

extern "C" int savfi_bias_act_fwd_f32(float* z, const float* bias, int N, int C, int HW, float slope, void* stream) {
  if (!z || !bias) return SAVFI_E_NULL;
  if (N <= 0 || C <= 0 || HW <= 0) return SAVFI_E_SHAPE;
  const int chunks = savfi_cdiv(HW, CHUNK);
  const int64_t blocks = (int64_t)N * C * chunks;
  if (blocks > 0x7fffffffLL) return SAVFI_E_TOOBIG;
  const int phase_ok = 1;      // a single operand: the peel aligns it
  hipLaunchKernelGGL(bias_act_fwd, dim3((unsigned)blocks), dim3(NT), 0, (hipStream_t)stream, z, bias, C, HW, chunks, slope, phase_ok);
  return savfi_launch_status();
}

extern "C" int64_t savfi_bias_act_scratch_floats(int N, int C, int HW) {
  if (N <= 0 || C <= 0 || HW <= 0) return SAVFI_E_SHAPE;
  return (int64_t)N * C * savfi_cdiv(HW, CHUNK);
}

extern "C" int savfi_bias_act_bwd_f32(const float* gy, const float* y, float* gz, float* gbias, float* scratch, int N, int C,
                                      int HW, float slope, void* stream) {
  if (!gy || !y || (!gz && !gbias) || (gbias && !scratch)) return SAVFI_E_NULL;   // gz may be NULL: bias gradient only
  if (N <= 0 || C <= 0 || HW <= 0) return SAVFI_E_SHAPE;
  const int chunks = savfi_cdiv(HW, CHUNK);
  const int64_t blocks = (int64_t)N * C * chunks;
  if (blocks > 0x7fffffffLL) return SAVFI_E_TOOBIG;
  // one peel aligns all operands when they share their low address bits (a NULL gz takes gy's)
  const unsigned lo = (unsigned)(uintptr_t)gy & 15u;
  const int phase_ok = ((unsigned)(uintptr_t)y & 15u) == lo && (!gz || ((unsigned)(uintptr_t)gz & 15u) == lo) && (lo & 3u) == 0;
#ifdef SAVFI_BIAS_ACT_TWO_STAGE      // variant builds: the general path on small maps too
  constexpr bool two_stage = true;
#else
  constexpr bool two_stage = false;
#endif
  if (gbias && chunks == 1 && (int64_t)N * HW <= CHUNK && !two_stage) {
    hipLaunchKernelGGL(bias_act_bwd_small, dim3((unsigned)C), dim3(NT), 0, (hipStream_t)stream, gy, y, gz, gbias, N, C, HW, slope, phase_ok);
    return savfi_launch_status();
  }
  hipLaunchKernelGGL(bias_act_bwd, dim3((unsigned)blocks), dim3(NT), 0, (hipStream_t)stream, gy, y, gz, gbias ? scratch : nullptr,
                     C, HW, chunks, slope, phase_ok);
  if (int e = savfi_launch_status()) return e;
  if (gbias) {
    hipLaunchKernelGGL(bias_grad_finish, dim3(C), dim3(64), 0, (hipStream_t)stream, scratch, gbias, N, C, chunks);
    return savfi_launch_status();
  }
  return SAVFI_OK;
}
