// Separable local convolution (SepConv) for gfx950 -- forward, filter gradients, input gradient.
//
// Replaces the four cupy/NVRTC kernels of the reference
// (sepconv/sepconv_op/sepconv.py:5-30, :32-63, :138-163, :165-190).
//
//   out[b,c,y,x] = sum_fy sum_fx in[b,c,y+fy,x+fx] * v[b,fy,y,x] * h[b,fx,y,x]
//
// Layout: fp32 NCHW, contiguous.  in [B,C,Ho+K-1,Wo+K-1], v/h [B,K,Ho,Wo], out/gO [B,C,Ho,Wo].
//
// Design (K = 51 fast path): one workgroup = 256 threads = an 8x32 output tile.  The
// (8+50)x(32+50) input halo of a channel is staged once in LDS with coalesced row reads
// and is then shared by the 256 pixels; each thread keeps its pixel's 51 horizontal taps
// in VGPRs and streams the vertical taps (coalesced along x: v/h planes are Ho*Wo apart,
// so a wave reads 51 x 256-byte segments per operand).  The product is factored
//   out = sum_fy v[fy] * (sum_fx in[y+fy][x+fx] * h[fx])            (K*K + K FMA / channel)
// and the backward shares one pass over the tile for both filter gradients:
//   P[fy][fx] = sum_c gO[c] * in[c][y+fy][x+fx]
//   gV[fy] = sum_fx P[fy][fx]*h[fx]     gH[fx] = sum_fy P[fy][fx]*v[fy]   (5*K*K FMA for C=3)
// Any other K (or C != 3 in the backward) takes the generic direct kernels below.
#include "common.h"

namespace {

constexpr int KFAST = 51;
constexpr int TY = 8, TX = 32, NT = TY * TX;

// ------------------------------------------------------------------------------------------
// forward, K = 51
// ------------------------------------------------------------------------------------------
template <int K>
__global__ __launch_bounds__(NT) void sepconv_fwd_tiled(const float* __restrict__ in,
                                                        const float* __restrict__ v,
                                                        const float* __restrict__ h,
                                                        float* __restrict__ out, int C, int Ho, int Wo) {
  constexpr int LH = TY + K - 1, LW = TX + K - 1;
  __shared__ float tile[LH * LW];
  const int tid = threadIdx.x, tx = tid % TX, ty = tid / TX;
  const int x0 = blockIdx.x * TX, y0 = blockIdx.y * TY, b = blockIdx.z;
  const int x = x0 + tx, y = y0 + ty;
  const bool valid = (x < Wo) && (y < Ho);
  const int Hi = Ho + K - 1, Wi = Wo + K - 1;
  const size_t plane = (size_t)Ho * Wo;
  const size_t pix = (size_t)b * K * plane + (size_t)(valid ? y : 0) * Wo + (valid ? x : 0);

  float hr[K];
#pragma unroll
  for (int f = 0; f < K; ++f) hr[f] = valid ? h[pix + f * plane] : 0.f;

  for (int c = 0; c < C; ++c) {
    const float* src = in + ((size_t)b * C + c) * Hi * Wi;
    __syncthreads();
    for (int i = tid; i < LH * LW; i += NT) {
      const int r = i / LW, q = i - r * LW;
      const int gy = y0 + r, gx = x0 + q;
      tile[i] = (gy < Hi && gx < Wi) ? src[(size_t)gy * Wi + gx] : 0.f;
    }
    __syncthreads();
    float acc = 0.f;
    for (int fy = 0; fy < K; ++fy) {
      const float vv = valid ? v[pix + fy * plane] : 0.f;
      const float* row = &tile[(ty + fy) * LW + tx];
      float t = 0.f;
#pragma unroll
      for (int fx = 0; fx < K; ++fx) t = fmaf(row[fx], hr[fx], t);
      acc = fmaf(vv, t, acc);
    }
    if (valid) out[((size_t)b * C + c) * plane + (size_t)y * Wo + x] = acc;
  }
}

// ------------------------------------------------------------------------------------------
// backward (gV, gH), K = 51, C = 3
// ------------------------------------------------------------------------------------------
template <int K, bool WANT_V, bool WANT_H>
__global__ __launch_bounds__(NT) void sepconv_bwd_filters_tiled(const float* __restrict__ in,
                                                                const float* __restrict__ v,
                                                                const float* __restrict__ h,
                                                                const float* __restrict__ gO,
                                                                float* __restrict__ gV,
                                                                float* __restrict__ gH, int Ho, int Wo) {
  constexpr int C = 3;
  constexpr int LH = TY + K - 1, LW = TX + K - 1, LP = LH * LW;
  extern __shared__ __attribute__((aligned(16))) float tile[];  // C * LP floats
  const int tid = threadIdx.x, tx = tid % TX, ty = tid / TX;
  const int x0 = blockIdx.x * TX, y0 = blockIdx.y * TY, b = blockIdx.z;
  const int x = x0 + tx, y = y0 + ty;
  const bool valid = (x < Wo) && (y < Ho);
  const int Hi = Ho + K - 1, Wi = Wo + K - 1;
  const size_t plane = (size_t)Ho * Wo;
  const size_t opix = (size_t)(valid ? y : 0) * Wo + (valid ? x : 0);
  const size_t pix = (size_t)b * K * plane + opix;

  for (int c = 0; c < C; ++c) {
    const float* src = in + ((size_t)b * C + c) * Hi * Wi;
    for (int i = tid; i < LP; i += NT) {
      const int r = i / LW, q = i - r * LW;
      const int gy = y0 + r, gx = x0 + q;
      tile[c * LP + i] = (gy < Hi && gx < Wi) ? src[(size_t)gy * Wi + gx] : 0.f;
    }
  }
  float hr[K], gh[K];
#pragma unroll
  for (int f = 0; f < K; ++f) {
    hr[f] = (WANT_V && valid) ? h[pix + f * plane] : 0.f;
    gh[f] = 0.f;
  }
  float go[C];
#pragma unroll
  for (int c = 0; c < C; ++c) go[c] = valid ? gO[((size_t)b * C + c) * plane + opix] : 0.f;
  __syncthreads();

  for (int fy = 0; fy < K; ++fy) {
    const float vv = (WANT_H && valid) ? v[pix + fy * plane] : 0.f;
    const float* r0 = &tile[(ty + fy) * LW + tx];
    float gv = 0.f;
#pragma unroll
    for (int fx = 0; fx < K; ++fx) {
      float p = go[0] * r0[fx];
      p = fmaf(go[1], r0[LP + fx], p);
      p = fmaf(go[2], r0[2 * LP + fx], p);
      if (WANT_V) gv = fmaf(p, hr[fx], gv);
      if (WANT_H) gh[fx] = fmaf(p, vv, gh[fx]);
    }
    if (WANT_V && valid) gV[pix + fy * plane] = gv;
  }
  if (WANT_H && valid) {
#pragma unroll
    for (int f = 0; f < K; ++f) gH[pix + f * plane] = gh[f];
  }
}

// ------------------------------------------------------------------------------------------
// generic direct kernels (any K, any C): one thread per output element, x fastest.
// ------------------------------------------------------------------------------------------
__global__ void sepconv_fwd_direct(const float* __restrict__ in, const float* __restrict__ v,
                                   const float* __restrict__ h, float* __restrict__ out, int C, int Ho,
                                   int Wo, int K) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x;
  const int y = blockIdx.y;
  const int bc = blockIdx.z, b = bc / C;
  if (x >= Wo) return;
  const int Hi = Ho + K - 1, Wi = Wo + K - 1;
  const size_t plane = (size_t)Ho * Wo;
  const size_t pix = (size_t)b * K * plane + (size_t)y * Wo + x;
  const float* src = in + (size_t)bc * Hi * Wi + (size_t)y * Wi + x;
  float acc = 0.f;
  for (int fy = 0; fy < K; ++fy) {
    float t = 0.f;
    for (int fx = 0; fx < K; ++fx) t = fmaf(src[(size_t)fy * Wi + fx], h[pix + fx * plane], t);
    acc = fmaf(v[pix + fy * plane], t, acc);
  }
  out[(size_t)bc * plane + (size_t)y * Wo + x] = acc;
}

// which = 0: gV[b,f,y,x] = sum_c gO * sum_fx in[y+f][x+fx]*h[fx]
// which = 1: gH[b,f,y,x] = sum_c gO * sum_fy in[y+fy][x+f]*v[fy]
__global__ void sepconv_bwd_filter_direct(const float* __restrict__ in, const float* __restrict__ other,
                                          const float* __restrict__ gO, float* __restrict__ gF, int C,
                                          int Ho, int Wo, int K, int which) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x;
  const int y = blockIdx.y;
  const int bf = blockIdx.z, b = bf / K, f = bf - b * K;
  if (x >= Wo) return;
  const int Hi = Ho + K - 1, Wi = Wo + K - 1;
  const size_t plane = (size_t)Ho * Wo;
  const size_t opix = (size_t)y * Wo + x;
  const size_t pix = (size_t)b * K * plane + opix;
  float acc = 0.f;
  for (int c = 0; c < C; ++c) {
    const float* src = in + ((size_t)b * C + c) * Hi * Wi;
    float t = 0.f;
    if (which == 0) {
      for (int fx = 0; fx < K; ++fx) t = fmaf(src[(size_t)(y + f) * Wi + x + fx], other[pix + fx * plane], t);
    } else {
      for (int fy = 0; fy < K; ++fy) t = fmaf(src[(size_t)(y + fy) * Wi + x + f], other[pix + fy * plane], t);
    }
    acc = fmaf(gO[((size_t)b * C + c) * plane + opix], t, acc);
  }
  gF[pix + f * plane] = acc;
}

// gI[b,c,Y,X] = sum over (fy,fx) with 0 <= Y-fy < Ho, 0 <= X-fx < Wo of
//               gO[b,c,Y-fy,X-fx] * v[b,fy,Y-fy,X-fx] * h[b,fx,Y-fy,X-fx]
// (exact adjoint of the forward; the reference kernel tests `> max` instead of `>= max`,
//  sepconv.py:51,54, and reads one row/column past the end).
__global__ void sepconv_bwd_input_direct(const float* __restrict__ v, const float* __restrict__ h,
                                         const float* __restrict__ gO, float* __restrict__ gI, int C,
                                         int Ho, int Wo, int K) {
  const int Hi = Ho + K - 1, Wi = Wo + K - 1;
  const int X = blockIdx.x * blockDim.x + threadIdx.x;
  const int Y = blockIdx.y;
  const int bc = blockIdx.z, b = bc / C;
  if (X >= Wi) return;
  const size_t plane = (size_t)Ho * Wo;
  const float* g = gO + (size_t)bc * plane;
  const float* vb = v + (size_t)b * K * plane;
  const float* hb = h + (size_t)b * K * plane;
  const int fy_lo = max(0, Y - (Ho - 1)), fy_hi = min(K - 1, Y);
  const int fx_lo = max(0, X - (Wo - 1)), fx_hi = min(K - 1, X);
  float acc = 0.f;
  for (int fy = fy_lo; fy <= fy_hi; ++fy) {
    const int y = Y - fy;
    for (int fx = fx_lo; fx <= fx_hi; ++fx) {
      const int x = X - fx;
      const size_t o = (size_t)y * Wo + x;
      acc = fmaf(g[o] * vb[fy * plane + o], hb[fx * plane + o], acc);
    }
  }
  gI[(size_t)bc * Hi * Wi + (size_t)Y * Wi + X] = acc;
}

int check_dims(int B, int C, int Ho, int Wo, int K) {
  if (B <= 0 || C <= 0 || Ho <= 0 || Wo <= 0 || K <= 0) return SAVFI_E_SHAPE;
  const int64_t Hi = (int64_t)Ho + K - 1, Wi = (int64_t)Wo + K - 1;
  if ((int64_t)B * K * Ho * Wo >= (int64_t)1 << 40 || (int64_t)B * C * Hi * Wi >= (int64_t)1 << 40)
    return SAVFI_E_TOOBIG;
  if ((int64_t)B * K > 65535 || (int64_t)B * C > 65535 || Ho + K - 1 > 65535) return SAVFI_E_TOOBIG;
  return SAVFI_OK;
}

}  // namespace

extern "C" int savfi_sepconv_fwd_f32(const float* in, const float* v, const float* h, float* out, int B,
                                     int C, int Ho, int Wo, int K, void* stream) {
  if (!in || !v || !h || !out) return SAVFI_E_NULL;
  if (int e = check_dims(B, C, Ho, Wo, K)) return e;
  hipStream_t st = (hipStream_t)stream;
  if (K == KFAST) {
    dim3 grid(savfi_cdiv(Wo, TX), savfi_cdiv(Ho, TY), B);
    hipLaunchKernelGGL(sepconv_fwd_tiled<KFAST>, grid, dim3(NT), 0, st, in, v, h, out, C, Ho, Wo);
  } else {
    dim3 grid(savfi_cdiv(Wo, 64), Ho, B * C);
    hipLaunchKernelGGL(sepconv_fwd_direct, grid, dim3(64), 0, st, in, v, h, out, C, Ho, Wo, K);
  }
  return savfi_launch_status();
}

extern "C" int savfi_sepconv_bwd_f32(const float* in, const float* v, const float* h, const float* gO,
                                     float* gI, float* gV, float* gH, int B, int C, int Ho, int Wo, int K,
                                     void* stream) {
  if (!in || !v || !h || !gO) return SAVFI_E_NULL;
  if (int e = check_dims(B, C, Ho, Wo, K)) return e;
  hipStream_t st = (hipStream_t)stream;
  if (gV || gH) {
    if (K == KFAST && C == 3) {
      constexpr int LP = (TY + KFAST - 1) * (TX + KFAST - 1);
      const size_t lds = (size_t)3 * LP * sizeof(float);
      dim3 grid(savfi_cdiv(Wo, TX), savfi_cdiv(Ho, TY), B);
      if (gV && gH)
        hipLaunchKernelGGL((sepconv_bwd_filters_tiled<KFAST, true, true>), grid, dim3(NT), lds, st, in, v, h,
                           gO, gV, gH, Ho, Wo);
      else if (gV)
        hipLaunchKernelGGL((sepconv_bwd_filters_tiled<KFAST, true, false>), grid, dim3(NT), lds, st, in, v, h,
                           gO, gV, gH, Ho, Wo);
      else
        hipLaunchKernelGGL((sepconv_bwd_filters_tiled<KFAST, false, true>), grid, dim3(NT), lds, st, in, v, h,
                           gO, gV, gH, Ho, Wo);
      if (int e = savfi_launch_status()) return e;
    } else {
      dim3 grid(savfi_cdiv(Wo, 64), Ho, B * K);
      if (gV) hipLaunchKernelGGL(sepconv_bwd_filter_direct, grid, dim3(64), 0, st, in, h, gO, gV, C, Ho, Wo, K, 0);
      if (gH) hipLaunchKernelGGL(sepconv_bwd_filter_direct, grid, dim3(64), 0, st, in, v, gO, gH, C, Ho, Wo, K, 1);
      if (int e = savfi_launch_status()) return e;
    }
  }
  if (gI) {
    dim3 grid(savfi_cdiv(Wo + K - 1, 64), Ho + K - 1, B * C);
    hipLaunchKernelGGL(sepconv_bwd_input_direct, grid, dim3(64), 0, st, v, h, gO, gI, C, Ho, Wo, K);
    if (int e = savfi_launch_status()) return e;
  }
  return SAVFI_OK;
}
