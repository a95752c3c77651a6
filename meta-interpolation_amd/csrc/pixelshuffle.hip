// Pixel unshuffle / shuffle (space-to-depth / depth-to-space) for gfx950.
//
// Replaces the view -> permute -> contiguous copies of the reference
// (model_utils.py:202-217).  Channel order is the reference's:
//   unshuffle: out[b, c*r*r + i*r + j, y, x] = in[b, c, y*r + i, x*r + j]
//   shuffle  : exact inverse.
//
// Pure HBM-bound permutation (2 x 4 bytes per element).  One workgroup owns one full-resolution
// image row (b, c, y*r+i): on the full-resolution side that row is one contiguous run, on the
// packed side it is r contiguous runs of W/r floats (one per j).  The row goes through LDS so that
// BOTH global sides are accessed in coalesced 256-byte wave segments; the LDS image is skewed by
// one bank per 32 floats so the stride-r side of the transposition is conflict-free for r = 8.
#include "common.h"

namespace {

constexpr int NT = 256;
constexpr int CHUNK = 2048;  // full-resolution columns staged per pass

__device__ __forceinline__ int skew(int q) { return q + (q >> 5); }

// in [B,C,H,W] -> out [B,C*r*r,H/r,W/r]
__global__ __launch_bounds__(NT) void pixel_unshuffle_rows(const float* __restrict__ in,
                                                           float* __restrict__ out, int C, int H, int W,
                                                           int r) {
  __shared__ float row[CHUNK + CHUNK / 32 + 1];
  const int Y = blockIdx.x;             // full-res row
  const int bc = blockIdx.y;            // b*C + c
  const int b = bc / C, c = bc - b * C;
  const int y = Y / r, i = Y - y * r;
  const int Ho = H / r, Wo = W / r;
  const float* src = in + ((size_t)bc * H + Y) * W;
  const int cw = (CHUNK / r) * r;       // columns per pass, multiple of r
  for (int q0 = 0; q0 < W; q0 += cw) {
    const int n = min(cw, W - q0), no = n / r;
    __syncthreads();
    for (int q = threadIdx.x; q < n; q += NT) row[skew(q)] = src[q0 + q];
    __syncthreads();
    for (int j = 0; j < r; ++j) {
      float* dst = out + (((size_t)b * C * r * r + (size_t)c * r * r + i * r + j) * Ho + y) * Wo + q0 / r;
      for (int x = threadIdx.x; x < no; x += NT) dst[x] = row[skew(x * r + j)];
    }
  }
}

// in [B,C*r*r,H,W] -> out [B,C,H*r,W*r]   (C = output channels)
__global__ __launch_bounds__(NT) void pixel_shuffle_rows(const float* __restrict__ in,
                                                         float* __restrict__ out, int C, int H, int W,
                                                         int r) {
  __shared__ float row[CHUNK + CHUNK / 32 + 1];
  const int Y = blockIdx.x;             // full-res output row
  const int bc = blockIdx.y;
  const int b = bc / C, c = bc - b * C;
  const int y = Y / r, i = Y - y * r;
  const int Wf = W * r;
  float* dst = out + ((size_t)bc * H * r + Y) * Wf;
  const int cw = (CHUNK / r) * r;
  for (int q0 = 0; q0 < Wf; q0 += cw) {
    const int n = min(cw, Wf - q0), no = n / r;
    __syncthreads();
    for (int j = 0; j < r; ++j) {
      const float* src = in + (((size_t)b * C * r * r + (size_t)c * r * r + i * r + j) * H + y) * W + q0 / r;
      for (int x = threadIdx.x; x < no; x += NT) row[skew(x * r + j)] = src[x];
    }
    __syncthreads();
    for (int q = threadIdx.x; q < n; q += NT) dst[q0 + q] = row[skew(q)];
  }
}

}  // namespace

extern "C" int savfi_pixel_unshuffle_f32(const float* in, float* out, int B, int C, int H, int W, int r,
                                         void* stream) {
  if (!in || !out) return SAVFI_E_NULL;
  if (B <= 0 || C <= 0 || H <= 0 || W <= 0 || r <= 0) return SAVFI_E_SHAPE;
  if (H % r || W % r) return SAVFI_E_SHAPE;
  if (r > 64 || (int64_t)B * C > 65535) return SAVFI_E_UNSUPPORTED;
  dim3 grid(H, B * C);
  hipLaunchKernelGGL(pixel_unshuffle_rows, grid, dim3(NT), 0, (hipStream_t)stream, in, out, C, H, W, r);
  return savfi_launch_status();
}

extern "C" int savfi_pixel_shuffle_f32(const float* in, float* out, int B, int C, int H, int W, int r,
                                       void* stream) {
  if (!in || !out) return SAVFI_E_NULL;
  if (B <= 0 || C <= 0 || H <= 0 || W <= 0 || r <= 0) return SAVFI_E_SHAPE;
  if (C % (r * r)) return SAVFI_E_SHAPE;
  const int Co = C / (r * r);
  if (r > 64 || (int64_t)B * Co > 65535) return SAVFI_E_UNSUPPORTED;
  dim3 grid(H * r, B * Co);
  hipLaunchKernelGGL(pixel_shuffle_rows, grid, dim3(NT), 0, (hipStream_t)stream, in, out, Co, H, W, r);
  return savfi_launch_status();
}
