// Frame staging for gfx950: decoded frames travel over PCIe as uint8 HWC (a quarter of the fp32 bytes) and become the
// fp32 NCHW tensors of the meta-batch on the GPU, on a side stream, while the previous meta-iteration computes.
//
// Replaces the host-side tail of the dataset readers: `[2,1,0]` channel swap, `np.transpose(im, (2,0,1))`,
// `.float() / 255` and torchvision's Normalize (data/vimeo_septuplet.py:68-80, data/video.py:44-51).
//   dst[n][c][y][x] = (src[n][y][x][swap ? 2-c : c] / div - mean[c]) / std    (fp32, the reference's operation order)
// One thread per pixel: 3 adjacent bytes in, one float to each of the 3 planes (coalesced along x).
#include "common.h"

namespace {

__global__ __launch_bounds__(256) void frames_u8_to_f32(const unsigned char* __restrict__ src, float* __restrict__ dst,
                                                        size_t pixels_per_image, size_t total_pixels, int swap_rb, float div,
                                                        float mean0, float mean1, float mean2, float std) {
  const size_t p = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (p >= total_pixels) return;
  const size_t n = p / pixels_per_image, q = p - n * pixels_per_image;
  const unsigned char* s = src + p * 3;
  float* d = dst + n * 3 * pixels_per_image + q;
  const float mean[3] = {mean0, mean1, mean2};
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const float v = (float)s[swap_rb ? 2 - c : c];
    d[(size_t)c * pixels_per_image] = (v / div - mean[c]) / std;
  }
}

}  // namespace

extern "C" int savfi_frames_u8_to_f32(const unsigned char* src, float* dst, int64_t N, int H, int W, int swap_rb, float div,
                                      float mean_c0, float mean_c1, float mean_c2, float std, void* stream) {
  if (!src || !dst) return SAVFI_E_NULL;
  if (N <= 0 || H <= 0 || W <= 0 || div == 0.f || std == 0.f) return SAVFI_E_SHAPE;
  const size_t ppi = (size_t)H * W, total = (size_t)N * ppi;
  if ((total + 255) / 256 > 0x7fffffffULL) return SAVFI_E_TOOBIG;
  hipLaunchKernelGGL(frames_u8_to_f32, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, src, dst, ppi,
                     total, swap_rb ? 1 : 0, div, mean_c0, mean_c1, mean_c2, std);
  return savfi_launch_status();
}
