// Backward warping by a pixel-unit optical flow for gfx950: out[n,c,y,x] = bilinear(img[n,c], x + u, y + v).
//
// Replaces SuperSloMo's backWarp (superslomo/model.py:231-307) and RRIN's warp (rrin/model.py:8-20): numpy meshgrid
// + H2D copy per call, coordinate arithmetic in five elementwise launches, stack, F.grid_sample(bilinear, zeros,
// align_corners=False) and the matching chain in backward.  Both normalise with 2 * (x / W - 0.5), which
// grid_sample's align_corners=False rule ((g + 1) * W - 1) / 2 turns into the source position x + u - 0.5: the
// reference samples half a pixel up-left of the flow target.  The same float expressions are evaluated here in the same
// order.  Corners outside the image contribute zero (padding_mode='zeros').
//
// The images that are warped on this path are always network INPUTS (I0, I1, x0, x1), so only the flow gradient is
// produced: a pure gather, no atomics, bit-reproducible.
//
// HBM-bound: per pixel 2 floats of flow in, C out, 4 corners x C channels from L2 (neighbouring lanes sample
// neighbouring texels).  One thread per pixel in 64 x 4 pixel workgroups: flow / out / gout accesses are coalesced
// 256-byte segments, and the two texel rows a pixel row samples are shared with the rows above / below through the CU's L1.
#include "common.h"

namespace {

constexpr int TX = 64, TY = 4;      // pixels per workgroup

struct Corner {
  int x0, y0;          // north-west corner (may lie outside)
  float fx, fy;        // fractional position inside the cell
};

__device__ __forceinline__ Corner source_cell(int x, int y, float u, float v, int H, int W) {
  // every step is a separately rounded fp32 operation in the reference (one torch op each): no FMA contraction here
#pragma clang fp contract(off)
  // reference: normx = 2 * ((gridX + u) / W - 0.5);  ATen: ((coord + 1) * size - 1) / 2
  const float nx = 2.f * (((float)x + u) / (float)W - 0.5f);
  const float ny = 2.f * (((float)y + v) / (float)H - 0.5f);
  const float ix = ((nx + 1.f) * (float)W - 1.f) / 2.f;
  const float iy = ((ny + 1.f) * (float)H - 1.f) / 2.f;
  const float fx0 = floorf(ix), fy0 = floorf(iy);
  Corner c;
  // positions far outside (also NaN / inf flows) sample nothing: park the cell outside the image
  c.x0 = (fx0 >= -2.f && fx0 <= (float)W) ? (int)fx0 : -2;
  c.y0 = (fy0 >= -2.f && fy0 <= (float)H) ? (int)fy0 : -2;
  c.fx = ix - fx0;
  c.fy = iy - fy0;
  return c;
}

__device__ __forceinline__ bool inside(int y, int x, int H, int W) { return x >= 0 && x < W && y >= 0 && y < H; }

// corner value times weight; a corner outside the image is skipped like ATen does (so that a NaN / inf weight of a
// parked cell does not turn the zero into NaN)
__device__ __forceinline__ float tap(const float* __restrict__ plane, int y, int x, int H, int W, float w) {
  return inside(y, x, H, W) ? plane[(size_t)y * W + x] * w : 0.f;
}

__device__ __forceinline__ float texel(const float* __restrict__ plane, int y, int x, int H, int W) {
  return inside(y, x, H, W) ? plane[(size_t)y * W + x] : 0.f;
}

__global__ __launch_bounds__(256) void flowwarp_fwd(const float* __restrict__ img, const float* __restrict__ flow,
                                                    float* __restrict__ out, int C, int H, int W) {
  const int x = blockIdx.x * TX + threadIdx.x;
  const int y = blockIdx.y * TY + threadIdx.y, n = blockIdx.z;
  if (x >= W || y >= H) return;
  const size_t plane = (size_t)H * W, p = (size_t)y * W + x;
  const float* f = flow + (size_t)n * 2 * plane + p;
  const Corner c = source_cell(x, y, f[0], f[plane], H, W);
  const float wnw = (1.f - c.fx) * (1.f - c.fy), wne = c.fx * (1.f - c.fy), wsw = (1.f - c.fx) * c.fy, wse = c.fx * c.fy;
  for (int ch = 0; ch < C; ++ch) {
    const float* a = img + ((size_t)n * C + ch) * plane;
    out[((size_t)n * C + ch) * plane + p] = tap(a, c.y0, c.x0, H, W, wnw) + tap(a, c.y0, c.x0 + 1, H, W, wne) +
                                            tap(a, c.y0 + 1, c.x0, H, W, wsw) + tap(a, c.y0 + 1, c.x0 + 1, H, W, wse);
  }
}

// gflow[n,0] = d L / d u = sum_c gout * d out / d ix  (d ix / d u = (W / 2) * (2 / W) = 1), gflow[n,1] likewise in y
__global__ __launch_bounds__(256) void flowwarp_bwd(const float* __restrict__ img, const float* __restrict__ flow,
                                                    const float* __restrict__ gout, float* __restrict__ gflow, int C, int H,
                                                    int W) {
  const int x = blockIdx.x * TX + threadIdx.x;
  const int y = blockIdx.y * TY + threadIdx.y, n = blockIdx.z;
  if (x >= W || y >= H) return;
  const size_t plane = (size_t)H * W, p = (size_t)y * W + x;
  const float* f = flow + (size_t)n * 2 * plane + p;
  const Corner c = source_cell(x, y, f[0], f[plane], H, W);
  float gx = 0.f, gy = 0.f;
  const bool any = inside(c.y0, c.x0, H, W) || inside(c.y0, c.x0 + 1, H, W) || inside(c.y0 + 1, c.x0, H, W) ||
                   inside(c.y0 + 1, c.x0 + 1, H, W);
  for (int ch = 0; any && ch < C; ++ch) {
    const float* a = img + ((size_t)n * C + ch) * plane;
    const float g = gout[((size_t)n * C + ch) * plane + p];
    const float nw = texel(a, c.y0, c.x0, H, W), ne = texel(a, c.y0, c.x0 + 1, H, W);
    const float sw = texel(a, c.y0 + 1, c.x0, H, W), se = texel(a, c.y0 + 1, c.x0 + 1, H, W);
    gx += g * ((ne - nw) * (1.f - c.fy) + (se - sw) * c.fy);
    gy += g * ((sw - nw) * (1.f - c.fx) + (se - ne) * c.fx);
  }
  // ATen multiplies by W / 2 (H / 2) and autograd of 2 * (x / W - 0.5) by 2 / W (2 / H)
  float* o = gflow + (size_t)n * 2 * plane + p;
  o[0] = gx * ((float)W / 2.f) * (2.f / (float)W);
  o[plane] = gy * ((float)H / 2.f) * (2.f / (float)H);
}

int check(const void* a, const void* b, const void* c, int N, int C, int H, int W) {
  if (!a || !b || !c) return SAVFI_E_NULL;
  if (N <= 0 || C <= 0 || H <= 0 || W <= 0) return SAVFI_E_SHAPE;
  if (savfi_cdiv(H, TY) > 65535 || N > 65535 || (int64_t)N * C * H * W >= ((int64_t)1 << 40)) return SAVFI_E_TOOBIG;
  return SAVFI_OK;
}

}  // namespace

extern "C" int savfi_flowwarp_fwd_f32(const float* img, const float* flow, float* out, int N, int C, int H, int W, void* stream) {
  if (int e = check(img, flow, out, N, C, H, W)) return e;
  hipLaunchKernelGGL(flowwarp_fwd, dim3(savfi_cdiv(W, TX), savfi_cdiv(H, TY), N), dim3(TX, TY), 0, (hipStream_t)stream, img, flow, out, C, H, W);
  return savfi_launch_status();
}

extern "C" int savfi_flowwarp_bwd_f32(const float* img, const float* flow, const float* gout, float* gflow, int N, int C,
                                      int H, int W, void* stream) {
  if (!gout) return SAVFI_E_NULL;
  if (int e = check(img, flow, gflow, N, C, H, W)) return e;
  hipLaunchKernelGGL(flowwarp_bwd, dim3(savfi_cdiv(W, TX), savfi_cdiv(H, TY), N), dim3(TX, TY), 0, (hipStream_t)stream, img, flow, gout, gflow, C,
                     H, W);
  return savfi_launch_status();
}
