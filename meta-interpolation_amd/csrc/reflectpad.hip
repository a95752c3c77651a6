// Adjoint of nn.ReflectionPad2d(p) for gfx950: gx[n,c,y,x] = sum of gp over the padded positions that mirror onto (y, x).
//
// CAIN's MetaConvNorm is ReflectionPad2d(1) + a 3x3 convolution without padding (reference model_utils.py:821-848).  On the direct
// kernels the forward reads the mirrored border while it stages its tile (savfi_convk_tasks_pre_reflect_f32) and the weight
// gradient does the same (savfi_convk_wgrad_tasks_reflect_f32): the padded copy of the activation no longer exists.  The data
// gradient is computed for the padded extent (the "full" data gradient of the unpadded convolution) and folded back here --
// a gather (each output reads its 1-9 sources; no zero fill, no atomics, deterministic) where ATen's
// reflection_pad2d_backward zero-fills and scatters with atomics (C5: 9.3 + 2.3 ms per meta-iteration, with the forward pad 17.5
// of 178 ms).
#include "common.h"

namespace {

// add != nullptr: gx = fold(gp) + add -- the map that was padded also went round the layer (CAIN's RCAB: x feeds the first mirrored
// convolution AND the block's skip connection); the two cotangents meet here instead of in an element-wise add of their own.
__global__ __launch_bounds__(256) void reflect_fold(const float* __restrict__ gp, const float* __restrict__ add, float* __restrict__ gx, int H, int W,
                                                    int p) {
  const int Wp = W + 2 * p, Hp = H + 2 * p;
  const int per_row = (W + 3) >> 2;
  const int item = blockIdx.x * 256 + threadIdx.x;
  if (item >= per_row * H) return;
  const int y = item / per_row, x0 = (item - y * per_row) * 4;
  const float* src = gp + (size_t)blockIdx.y * Hp * Wp;
  int ys[3], ny = 0;
  ys[ny++] = y + p;
  if (y >= 1 && y <= p) ys[ny++] = p - y;
  if (y >= H - 1 - p && y <= H - 2) ys[ny++] = 2 * H - 2 + p - y;
  float v[4] = {0.f, 0.f, 0.f, 0.f};
  for (int a = 0; a < ny; ++a) {
    const float* row = src + (size_t)ys[a] * Wp;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int x = x0 + k;
      if (x >= W) continue;
      float t = row[x + p];
      if (x >= 1 && x <= p) t += row[p - x];
      if (x >= W - 1 - p && x <= W - 2) t += row[2 * W - 2 + p - x];
      v[k] += t;
    }
  }
  const size_t at = ((size_t)blockIdx.y * H + y) * W + x0;
  float* o = gx + at;
  if (add) {
    const float* q = add + at;
    if (x0 + 3 < W && ((((uintptr_t)q) & 15u) == 0)) {
      const float4 t = *reinterpret_cast<const float4*>(q);
      v[0] += t.x; v[1] += t.y; v[2] += t.z; v[3] += t.w;
    } else
      for (int k = 0; k < 4 && x0 + k < W; ++k) v[k] += q[k];
  }
  if (x0 + 3 < W && ((((uintptr_t)o) & 15u) == 0)) *reinterpret_cast<float4*>(o) = make_float4(v[0], v[1], v[2], v[3]);
  else
    for (int k = 0; k < 4 && x0 + k < W; ++k) o[k] = v[k];
}

// nn.ReflectionPad2d(p) itself, for the layers that keep a padded copy (the Winograd kernels read zero-padded or unpadded maps only):
// four output columns per thread, each a mirrored gather; rows leave as 16- or 8-byte pieces where their address allows (a padded
// row of W + 2p floats starts 16-byte aligned every other row for even W).  ATen's reflection_pad2d_out_kernel takes 15.9 us for CAIN's
// [2,192,96,160] maps (384 launches of it in a C5 meta-iteration), this one the time of its 48 MB.
__global__ __launch_bounds__(256) void reflect_spread(const float* __restrict__ x, float* __restrict__ xp, int H, int W, int p) {
  const int Wp = W + 2 * p, Hp = H + 2 * p;
  const int per_row = (Wp + 3) >> 2;
  const int item = blockIdx.x * 256 + threadIdx.x;
  if (item >= per_row * Hp) return;
  const int yp = item / per_row, x0 = (item - yp * per_row) * 4;
  int ys = yp - p;
  ys = ys < 0 ? -ys : (ys >= H ? 2 * H - 2 - ys : ys);
  const float* row = x + ((size_t)blockIdx.y * H + ys) * W;
  float v[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    int xs = x0 + k - p;
    xs = xs < 0 ? -xs : (xs >= W ? 2 * W - 2 - xs : xs);
    v[k] = (x0 + k < Wp) ? row[xs] : 0.f;
  }
  float* o = xp + ((size_t)blockIdx.y * Hp + yp) * Wp + x0;
  const unsigned lo = (unsigned)(uintptr_t)o & 15u;
  if (x0 + 3 < Wp && lo == 0) *reinterpret_cast<float4*>(o) = make_float4(v[0], v[1], v[2], v[3]);
  else if (x0 + 3 < Wp && lo == 8) {
    *reinterpret_cast<float2*>(o) = make_float2(v[0], v[1]);
    *reinterpret_cast<float2*>(o + 2) = make_float2(v[2], v[3]);
  } else
    for (int k = 0; k < 4 && x0 + k < Wp; ++k) o[k] = v[k];
}

}  // namespace

extern "C" int savfi_reflect_pad_fwd_f32(const float* x, float* xp, int planes, int H, int W, int pad, void* stream) {
  if (!x || !xp) return SAVFI_E_NULL;
  if (planes <= 0 || H <= 0 || W <= 0 || pad < 0 || pad >= H || pad >= W) return SAVFI_E_SHAPE;
  if (planes > 65535) return SAVFI_E_TOOBIG;
  dim3 grid(savfi_cdiv((int64_t)(H + 2 * pad) * savfi_cdiv(W + 2 * pad, 4), 256), planes, 1);
  hipLaunchKernelGGL(reflect_spread, grid, dim3(256), 0, (hipStream_t)stream, x, xp, H, W, pad);
  return savfi_launch_status();
}

extern "C" int savfi_reflect_pad_bwd_add_f32(const float* gp, const float* add, float* gx, int planes, int H, int W, int pad, void* stream) {
  if (!gp || !gx) return SAVFI_E_NULL;
  if (planes <= 0 || H <= 0 || W <= 0 || pad < 0 || pad >= H || pad >= W) return SAVFI_E_SHAPE;
  if (planes > 65535) return SAVFI_E_TOOBIG;
  dim3 grid(savfi_cdiv((int64_t)H * savfi_cdiv(W, 4), 256), planes, 1);
  hipLaunchKernelGGL(reflect_fold, grid, dim3(256), 0, (hipStream_t)stream, gp, add, gx, H, W, pad);
  return savfi_launch_status();
}

extern "C" int savfi_reflect_pad_bwd_f32(const float* gp, float* gx, int planes, int H, int W, int pad, void* stream) {
  return savfi_reflect_pad_bwd_add_f32(gp, nullptr, gx, planes, H, W, pad, stream);
}
