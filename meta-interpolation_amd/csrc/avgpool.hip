// 2x2 / stride 2 average pooling for gfx950 (the only pooling on the SepConv / RRIN / Super SloMo path).
//
// Replaces torch.nn.AvgPool2d(kernel_size=2, stride=2, count_include_pad=False) (sepconv/model.py:176-187) and
// F.avg_pool2d(x, 2) (rrin/unet.py:146, superslomo/model.py:66): ATen's generic avg_pool2d kernels, whose backward
// (`avg_pool2d_backward_out_cuda_frame`) runs a window search per input pixel - 27 us on average in the C2 loop
// for a pure copy-and-scale.
//   fwd: out[y][x] = ((in[2y][2x] + in[2y][2x+1]) + in[2y+1][2x] + in[2y+1][2x+1]) / 4     (ATen's summation order)
//   bwd: gin[y][x] = gout[y/2][x/2] / 4 inside the pooled area, 0 in the odd last row / column
// HBM-bound: 5 floats per output pixel either way.  One thread per OUTPUT (pooled) pixel, x fastest; float2 accesses on
// the full-resolution side when W is even.
#include "common.h"

namespace {

__global__ __launch_bounds__(256) void avgpool2x2_fwd(const float* __restrict__ in, float* __restrict__ out, int H, int W,
                                                      int Ho, int Wo, int vec_ok) {
  const int item = blockIdx.x * 256 + threadIdx.x;         // pooled pixels of one plane, row-major (narrow maps fill workgroups too)
  if (item >= Ho * Wo) return;
  const int y = item / Wo, x = item - y * Wo;
  const size_t pl = blockIdx.y;
  const float* r0 = in + (pl * H + 2 * y) * W + 2 * x;
  const float* r1 = r0 + W;
  float a, b, c, d;
  if (vec_ok) {
    const float2 t = *reinterpret_cast<const float2*>(r0), u = *reinterpret_cast<const float2*>(r1);
    a = t.x; b = t.y; c = u.x; d = u.y;
  } else {
    a = r0[0]; b = r0[1]; c = r1[0]; d = r1[1];
  }
  out[(pl * Ho + y) * Wo + x] = (((a + b) + c) + d) / 4.f;
}

// one thread per pooled pixel writes its 2x2 block; the threads of the last pooled row / column also clear the odd rest
__global__ __launch_bounds__(256) void avgpool2x2_bwd(const float* __restrict__ gout, float* __restrict__ gin, int H, int W,
                                                      int Ho, int Wo, int vec_ok) {
  const int item = blockIdx.x * 256 + threadIdx.x;
  if (item >= Ho * Wo) return;
  const int y = item / Wo, x = item - y * Wo;
  const size_t pl = blockIdx.y;
  const float g = gout[(pl * Ho + y) * Wo + x] / 4.f;
  float* r0 = gin + (pl * H + 2 * y) * W + 2 * x;
  float* r1 = r0 + W;
  if (vec_ok) {
    *reinterpret_cast<float2*>(r0) = make_float2(g, g);
    *reinterpret_cast<float2*>(r1) = make_float2(g, g);
  } else {
    r0[0] = g; r0[1] = g; r1[0] = g; r1[1] = g;
  }
  if (x == Wo - 1 && (W & 1)) { r0[2] = 0.f; r1[2] = 0.f; }
  if (y == Ho - 1 && (H & 1)) {
    float* r2 = r1 + W;
    r2[0] = 0.f; r2[1] = 0.f;
    if (x == Wo - 1 && (W & 1)) r2[2] = 0.f;
  }
}

// The adjoint of "pool AND keep" (round 5): an encoder block's activated output y feeds the pooling and, as a skip connection, the decoder.
// Autograd ran the pooling's adjoint, added the skip connection's cotangent and then the block's last ReLU derivative as three element-wise
// passes; here they are one:   gin = (gout[y/2][x/2] / 4 + gskip) * (y > 0 ? 1 : slope)      (gskip, y: optional)
__global__ __launch_bounds__(256) void avgpool2x2_bwd_fused(const float* __restrict__ gout, const float* __restrict__ gskip,
                                                            const float* __restrict__ act, float slope, float* __restrict__ gin, int H, int W,
                                                            int Ho, int Wo, int vec_ok) {
  const int item = blockIdx.x * 256 + threadIdx.x;
  if (item >= Ho * Wo) return;
  const int y = item / Wo, x = item - y * Wo;
  const size_t pl = blockIdx.y;
  const float g = gout ? gout[(pl * Ho + y) * Wo + x] / 4.f : 0.f;
  const size_t o0 = (pl * H + 2 * y) * W + 2 * x, o1 = o0 + W;
  auto one = [&](size_t o, float pooled) {          // one full-resolution element
    const float v = pooled + (gskip ? gskip[o] : 0.f);
    gin[o] = act ? (act[o] > 0.f ? v : slope * v) : v;
  };
  if (vec_ok) {
    auto two = [&](size_t o) {
      const float2 s = gskip ? *reinterpret_cast<const float2*>(gskip + o) : make_float2(0.f, 0.f);
      float2 v = make_float2(g + s.x, g + s.y);
      if (act) {
        const float2 a = *reinterpret_cast<const float2*>(act + o);
        v.x = a.x > 0.f ? v.x : slope * v.x;
        v.y = a.y > 0.f ? v.y : slope * v.y;
      }
      *reinterpret_cast<float2*>(gin + o) = v;
    };
    two(o0); two(o1);
  } else {
    one(o0, g); one(o0 + 1, g); one(o1, g); one(o1 + 1, g);
  }
  // the odd last column / row is not pooled: only the skip connection reaches it
  if (x == Wo - 1 && (W & 1)) { one(o0 + 2, 0.f); one(o1 + 2, 0.f); }
  if (y == Ho - 1 && (H & 1)) {
    const size_t o2 = o1 + W;
    one(o2, 0.f); one(o2 + 1, 0.f);
    if (x == Wo - 1 && (W & 1)) one(o2 + 2, 0.f);
  }
}

int check(const void* a, const void* b, int64_t planes, int H, int W) {
  if (!a || !b) return SAVFI_E_NULL;
  if (planes <= 0 || H < 2 || W < 2) return SAVFI_E_SHAPE;
  if (planes > 65535 || (int64_t)H * W >= ((int64_t)1 << 31) || planes * H * W >= ((int64_t)1 << 40)) return SAVFI_E_TOOBIG;
  return SAVFI_OK;
}

}  // namespace

extern "C" int savfi_avgpool2x2_fwd_f32(const float* in, float* out, int64_t planes, int H, int W, void* stream) {
  if (int e = check(in, out, planes, H, W)) return e;
  const int Ho = H / 2, Wo = W / 2;
  const int vec_ok = (W % 2 == 0) && (((uintptr_t)in & 7u) == 0);
  hipLaunchKernelGGL(avgpool2x2_fwd, dim3(savfi_cdiv((int64_t)Ho * Wo, 256), (unsigned)planes, 1), dim3(256), 0, (hipStream_t)stream, in, out, H,
                     W, Ho, Wo, vec_ok);
  return savfi_launch_status();
}

extern "C" int savfi_avgpool2x2_bwd_f32(const float* gout, float* gin, int64_t planes, int H, int W, void* stream) {
  if (int e = check(gout, gin, planes, H, W)) return e;
  const int Ho = H / 2, Wo = W / 2;
  const int vec_ok = (W % 2 == 0) && (((uintptr_t)gin & 7u) == 0);
  hipLaunchKernelGGL(avgpool2x2_bwd, dim3(savfi_cdiv((int64_t)Ho * Wo, 256), (unsigned)planes, 1), dim3(256), 0, (hipStream_t)stream, gout, gin,
                     H, W, Ho, Wo, vec_ok);
  return savfi_launch_status();
}

extern "C" int savfi_avgpool2x2_bwd_fused_f32(const float* gout, const float* gskip, const float* y, float slope, float* gin, int64_t planes,
                                              int H, int W, void* stream) {
  if (!gout && !gskip) return SAVFI_E_NULL;
  if (int e = check(gin, gin, planes, H, W)) return e;
  const int Ho = H / 2, Wo = W / 2;
  const int vec_ok = (W % 2 == 0) && (((uintptr_t)gin & 7u) == 0) && (!gskip || ((uintptr_t)gskip & 7u) == 0) && (!y || ((uintptr_t)y & 7u) == 0);
  hipLaunchKernelGGL(avgpool2x2_bwd_fused, dim3(savfi_cdiv((int64_t)Ho * Wo, 256), (unsigned)planes, 1), dim3(256), 0, (hipStream_t)stream, gout,
                     gskip, y, slope, gin, H, W, Ho, Wo, vec_ok);
  return savfi_launch_status();
}
