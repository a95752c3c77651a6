// Per-plane spatial mean and its removal for gfx950: out[p][i] = x[p][i] - mean_p, mean_p = sum_i x[p][i] / hw.
//
// CAIN removes the per-channel mean of both input frames before the pixel shuffle and adds it back to the prediction (reference
// cain/model.py:70-94, sub_mean in model_utils.py:11-15: mean over H, then over W).  ATen runs that reduction with several
// workgroups per output once a frame is large (3 x 720 x 1280: 3840 outputs of 720 elements) and lets the LAST workgroup of an
// output, found through a semaphore array cleared by hipMemsetAsync, write it.  Inside a captured hipGraph that memset is a memset
// node, and on ROCm 7.2 a memset node writes zeros only in the first launch of the instantiated graph (later launches leave a
// small pattern of non-zero words: tools/graph_memset_probe.py, profiles/r03_graph_memset_nodes.txt): no workgroup is "last", the
// mean keeps the value of an earlier replay and every later layer sees frames with the wrong offset.  This file gives the
// operation in two launches that need no cleared memory and add in a fixed order: partial sums per (plane, chunk), then every
// workgroup of the second launch adds its plane's partials in the same serial order, subtracts, and workgroup 0 stores the mean.
#include "common.h"

namespace {

constexpr int SM_T = 256;
constexpr int SM_CHUNK = 16384;      // floats per workgroup: 64 KB, 57 chunks for a 720p plane

__global__ __launch_bounds__(SM_T) void plane_partial_sums(const float* __restrict__ x, float* __restrict__ partial, int hw, int chunks) {
  __shared__ float red[SM_T / SAVFI_WAVE];
  const int chunk = blockIdx.x, plane = blockIdx.y;
  const size_t base = (size_t)plane * hw;
  const int lo = chunk * SM_CHUNK, hi = min(hw, lo + SM_CHUNK);
  float acc = 0.f;
  if (((base + lo) & 3) == 0) {
    const float4* x4 = reinterpret_cast<const float4*>(x + base + lo);
    const int n4 = (hi - lo) >> 2;
    for (int i = threadIdx.x; i < n4; i += SM_T) {
      const float4 v = x4[i];
      acc += (v.x + v.y) + (v.z + v.w);
    }
    for (int i = lo + 4 * n4 + threadIdx.x; i < hi; i += SM_T) acc += x[base + i];
  } else {
    for (int i = lo + threadIdx.x; i < hi; i += SM_T) acc += x[base + i];
  }
  const float tot = block_sum<SM_T / SAVFI_WAVE>(acc, red);
  if (threadIdx.x == 0) partial[(size_t)plane * chunks + chunk] = tot;
}

__global__ __launch_bounds__(SM_T) void plane_sub_mean(const float* __restrict__ x, const float* __restrict__ partial, float* __restrict__ out,
                                                       float* __restrict__ mean, int hw, int chunks, float inv_hw) {
  const int chunk = blockIdx.x, plane = blockIdx.y;
  float tot = 0.f;
  for (int i = 0; i < chunks; ++i) tot += partial[(size_t)plane * chunks + i];      // same order in every thread of every workgroup
  const float m = tot * inv_hw;
  if (chunk == 0 && threadIdx.x == 0) mean[plane] = m;
  const size_t base = (size_t)plane * hw;
  const int lo = chunk * SM_CHUNK, hi = min(hw, lo + SM_CHUNK);
  if (((base + lo) & 3) == 0) {
    const float4* x4 = reinterpret_cast<const float4*>(x + base + lo);
    float4* o4 = reinterpret_cast<float4*>(out + base + lo);
    const int n4 = (hi - lo) >> 2;
    for (int i = threadIdx.x; i < n4; i += SM_T) {
      const float4 v = x4[i];
      o4[i] = make_float4(v.x - m, v.y - m, v.z - m, v.w - m);
    }
    for (int i = lo + 4 * n4 + threadIdx.x; i < hi; i += SM_T) out[base + i] = x[base + i] - m;
  } else {
    for (int i = lo + threadIdx.x; i < hi; i += SM_T) out[base + i] = x[base + i] - m;
  }
}

}  // namespace

extern "C" int64_t savfi_sub_mean_workspace_floats(int64_t planes, int hw) {
  if (planes <= 0 || hw <= 0) return SAVFI_E_SHAPE;
  return planes * savfi_cdiv(hw, SM_CHUNK);
}

// x, out [planes][hw] (out may alias x), mean [planes], workspace savfi_sub_mean_workspace_floats(planes, hw) floats
extern "C" int savfi_sub_mean_f32(const float* x, float* out, float* mean, float* workspace, int64_t planes, int hw, void* stream) {
  if (!x || !out || !mean || !workspace) return SAVFI_E_NULL;
  if (planes <= 0 || hw <= 0) return SAVFI_E_SHAPE;
  if (planes > 65535) return SAVFI_E_TOOBIG;
  const int chunks = savfi_cdiv(hw, SM_CHUNK);
  const dim3 grid(chunks, (unsigned)planes, 1);
  hipLaunchKernelGGL(plane_partial_sums, grid, dim3(SM_T), 0, (hipStream_t)stream, x, workspace, hw, chunks);
  if (int e = savfi_launch_status()) return e;
  hipLaunchKernelGGL(plane_sub_mean, grid, dim3(SM_T), 0, (hipStream_t)stream, x, workspace, out, mean, hw, chunks, 1.0f / (float)hw);
  return savfi_launch_status();
}
