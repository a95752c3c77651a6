// Channel attention + residual of CAIN's RCAB (reference model_utils.py:931-953 MetaCALayer, :957-990 MetaRCAB) for gfx950:
//
//   s[n][c]  = mean over H x W of t[n][c]                                  (global average pool)
//   y[n][c]  = sigmoid( W2 . relu( W1 . s[n] + b1 ) + b2 )[c]              (1x1 convs C -> C/r -> C on a 1x1 map)
//   out      = t * y[n][c] + x                                             (scale, RCAB skip connection)
//
// The reference composes this from AdaptiveAvgPool2d, two F.conv2d on [N,C,1,1], ReLU, Sigmoid, a broadcast multiply and an add:
// 8 launches forward and ~14 backward per block, 60 blocks per pass, each of the two map-sized ones a full HBM round trip of an
// 11.8 MB map at 720p (config C5), the small ones pure launch latency.  Here: three launches forward (pool, MLP, scale + add) and
// three backward (dot, MLP backward incl. the parameter gradients, scale backward); every map is read once per launch.
// Per-task weights (tasks adapted in lockstep): T sets, sample n uses set n % T.  Deterministic (fixed-order reductions).
#include "common.h"

namespace {

constexpr int CA_T = 256;

// one workgroup per (n, c) plane: s = scale * sum_hw a[.] (* b[.] if b).  NT = 256 for the small maps; planes of >= 4096 elements take
// 1024 threads with four 16-byte loads of each operand in flight per thread (CAIN at 720p: 384 planes of 61 KB -- one or two workgroups
// per CU, so a plane's time is its loads' latency times the trips of the loop: 15 trips of one load at 256 threads, 9.3 us; one trip here)
template <int NT>
__global__ __launch_bounds__(NT) void ca_pool_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ s,
                                                     int hw, float scale) {
  __shared__ float red[NT / SAVFI_WAVE];
  const size_t base = (size_t)blockIdx.x * hw;
  float acc = 0.f;
  const int hw4 = hw & ~3;
  if (((base * 4) & 15) == 0) {
    const float4* a4 = reinterpret_cast<const float4*>(a + base);
    const float4* b4 = b ? reinterpret_cast<const float4*>(b + base) : nullptr;
    const int n4 = hw4 / 4;
    if (NT > 256) {
      float part[4] = {0.f, 0.f, 0.f, 0.f};
      for (int i0 = threadIdx.x; i0 < n4; i0 += 4 * NT) {
        float4 v[4], u[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int i = i0 + k * NT;
          v[k] = i < n4 ? a4[i] : make_float4(0.f, 0.f, 0.f, 0.f);
          if (b4) u[k] = i < n4 ? b4[i] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k)
          part[k] += b4 ? v[k].x * u[k].x + v[k].y * u[k].y + v[k].z * u[k].z + v[k].w * u[k].w : (v[k].x + v[k].y) + (v[k].z + v[k].w);
      }
      acc = (part[0] + part[1]) + (part[2] + part[3]);
    } else {
      for (int i = threadIdx.x; i < n4; i += NT) {
        const float4 v = a4[i];
        if (b4) { const float4 u = b4[i]; acc += v.x * u.x + v.y * u.y + v.z * u.z + v.w * u.w; }
        else acc += (v.x + v.y) + (v.z + v.w);
      }
    }
    for (int i = hw4 + threadIdx.x; i < hw; i += NT) acc += b ? a[base + i] * b[base + i] : a[base + i];
  } else {
    for (int i = threadIdx.x; i < hw; i += NT) acc += b ? a[base + i] * b[base + i] : a[base + i];
  }
  const float tot = block_sum<NT / SAVFI_WAVE>(acc, red);
  if (threadIdx.x == 0) s[blockIdx.x] = tot * scale;
}

// one workgroup per sample: y = sigmoid(W2 relu(W1 s + b1) + b2); a1 = relu(...) kept for the backward
__global__ __launch_bounds__(CA_T) void ca_mlp_fwd_kernel(const float* __restrict__ s, const float* __restrict__ w1, const float* __restrict__ b1,
                                                          const float* __restrict__ w2, const float* __restrict__ b2, float* __restrict__ y,
                                                          float* __restrict__ a1, int T, int C, int Cr) {
  __shared__ float red[CA_T / SAVFI_WAVE];
  __shared__ float hid[64];
  const int n = blockIdx.x, t = n % T;
  const float* sn = s + (size_t)n * C;
  const float* W1 = w1 + (size_t)t * Cr * C;
  const float* W2 = w2 + (size_t)t * C * Cr;
  for (int j = 0; j < Cr; ++j) {
    float p = 0.f;
    for (int c = threadIdx.x; c < C; c += CA_T) p += W1[(size_t)j * C + c] * sn[c];
    const float z = block_sum<CA_T / SAVFI_WAVE>(p, red);
    if (threadIdx.x == 0) {
      const float h = fmaxf(z + b1[t * Cr + j], 0.f);
      hid[j] = h;
      a1[(size_t)n * Cr + j] = h;
    }
  }
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += CA_T) {
    float z = b2[t * C + c];
    for (int j = 0; j < Cr; ++j) z += W2[(size_t)c * Cr + j] * hid[j];
    y[(size_t)n * C + c] = 1.f / (1.f + __expf(-z));
  }
}

// one workgroup per TASK: loops over its samples n = t, t + T, ...; r[n][c] = sum_hw g * t (from ca_pool_kernel)
//   dz2 = r y (1 - y); da1 = W2^T dz2; dz1 = da1 [a1 > 0]; ds = W1^T dz1 * inv_hw          -> ds[n][c]
//   gW2 += dz2 (x) a1, gb2 += dz2, gW1 += dz1 (x) s, gb1 += dz1                             -> per task
__global__ __launch_bounds__(CA_T) void ca_mlp_bwd_kernel(const float* __restrict__ r, const float* __restrict__ s, const float* __restrict__ y,
                                                          const float* __restrict__ a1, const float* __restrict__ w1, const float* __restrict__ w2,
                                                          float* __restrict__ ds, float* __restrict__ gw1, float* __restrict__ gb1,
                                                          float* __restrict__ gw2, float* __restrict__ gb2, int N, int T, int C, int Cr, float inv_hw) {
  __shared__ float red[CA_T / SAVFI_WAVE];
  __shared__ float dz1s[64], a1s[64];
  const int t = blockIdx.x;
  const float* W1 = w1 + (size_t)t * Cr * C;
  const float* W2 = w2 + (size_t)t * C * Cr;
  float* GW1 = gw1 + (size_t)t * Cr * C;
  float* GW2 = gw2 + (size_t)t * C * Cr;
  // thread c owns channels c, c + 256, ... (C <= 1024: up to 4) and their parameter-gradient rows
  for (int c = threadIdx.x; c < C; c += CA_T) {
    gb2[t * C + c] = 0.f;
    for (int j = 0; j < Cr; ++j) { GW2[(size_t)c * Cr + j] = 0.f; GW1[(size_t)j * C + c] = 0.f; }
  }
  if (threadIdx.x < Cr) gb1[t * Cr + threadIdx.x] = 0.f;
  __syncthreads();
  for (int n = t; n < N; n += T) {
    const float* rn = r + (size_t)n * C;
    const float* yn = y + (size_t)n * C;
    const float* sn = s + (size_t)n * C;
    if (threadIdx.x < Cr) a1s[threadIdx.x] = a1[(size_t)n * Cr + threadIdx.x];
    __syncthreads();
    float dz2[4];
    int k = 0;
    for (int c = threadIdx.x; c < C; c += CA_T, ++k) {
      const float yv = yn[c];
      dz2[k] = rn[c] * yv * (1.f - yv);
      gb2[t * C + c] += dz2[k];
      for (int j = 0; j < Cr; ++j) GW2[(size_t)c * Cr + j] += dz2[k] * a1s[j];
    }
    for (int j = 0; j < Cr; ++j) {
      float p = 0.f;
      k = 0;
      for (int c = threadIdx.x; c < C; c += CA_T, ++k) p += W2[(size_t)c * Cr + j] * dz2[k];
      const float da = block_sum<CA_T / SAVFI_WAVE>(p, red);
      if (threadIdx.x == 0) {
        const float d = a1s[j] > 0.f ? da : 0.f;
        dz1s[j] = d;
        gb1[t * Cr + j] += d;
      }
    }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += CA_T) {
      float v = 0.f;
      for (int j = 0; j < Cr; ++j) {
        v += W1[(size_t)j * C + c] * dz1s[j];
        GW1[(size_t)j * C + c] += dz1s[j] * sn[c];
      }
      ds[(size_t)n * C + c] = v * inv_hw;
    }
    __syncthreads();
  }
}

// ---- hidden width <= 16 (CAIN: C / 16 = 12) and C <= 256: one reduction round instead of one per hidden unit ---------------------
// The generic kernels above take a block reduction (two barriers) per hidden unit, 12 in a row forward and backward: 13 / 25 us per
// launch, 360 launches per C5 meta-iteration.  Here a thread carries all CRM partial dot products of its channel, the wave sums are
// lane exchanges and the four waves meet in LDS once.
constexpr int CRM = 16;

__device__ __forceinline__ void ca_block_sums(float (&p)[CRM], int Cr, float (*red)[CA_T / SAVFI_WAVE], float* out) {
  const int lane = threadIdx.x & (SAVFI_WAVE - 1), wid = threadIdx.x / SAVFI_WAVE;
#pragma unroll
  for (int j = 0; j < CRM; ++j)
    if (j < Cr) {
      const float v = wave_sum(p[j]);
      if (lane == 0) red[j][wid] = v;
    }
  __syncthreads();
  if ((int)threadIdx.x < Cr) {
    // the association of block_sum (a butterfly over the four wave totals): bit-identical to the generic kernels
    static_assert(CA_T / SAVFI_WAVE == 4, "four wave totals");
    out[threadIdx.x] = (red[threadIdx.x][0] + red[threadIdx.x][2]) + (red[threadIdx.x][1] + red[threadIdx.x][3]);
  }
  __syncthreads();
}

__global__ __launch_bounds__(CA_T) void ca_mlp_fwd_small(const float* __restrict__ s, const float* __restrict__ w1, const float* __restrict__ b1,
                                                         const float* __restrict__ w2, const float* __restrict__ b2, float* __restrict__ y,
                                                         float* __restrict__ a1, int T, int C, int Cr) {
  __shared__ float red[CRM][CA_T / SAVFI_WAVE];
  __shared__ float hid[CRM];
  const int n = blockIdx.x, t = n % T, c = threadIdx.x;
  const float* W1 = w1 + (size_t)t * Cr * C;
  const float* W2 = w2 + (size_t)t * C * Cr;
  const float sv = c < C ? s[(size_t)n * C + c] : 0.f;
  float p[CRM];
#pragma unroll
  for (int j = 0; j < CRM; ++j) p[j] = (j < Cr && c < C) ? W1[(size_t)j * C + c] * sv : 0.f;
  ca_block_sums(p, Cr, red, hid);
  if (c < Cr) {
    const float h = fmaxf(hid[c] + b1[t * Cr + c], 0.f);
    a1[(size_t)n * Cr + c] = h;
    hid[c] = h;
  }
  __syncthreads();
  if (c < C) {
    float z = b2[t * C + c];
    for (int j = 0; j < Cr; ++j) z += W2[(size_t)c * Cr + j] * hid[j];
    y[(size_t)n * C + c] = 1.f / (1.f + __expf(-z));
  }
}

__global__ __launch_bounds__(CA_T) void ca_mlp_bwd_small(const float* __restrict__ r, const float* __restrict__ s, const float* __restrict__ y,
                                                         const float* __restrict__ a1, const float* __restrict__ w1, const float* __restrict__ w2,
                                                         float* __restrict__ ds, float* __restrict__ gw1, float* __restrict__ gb1,
                                                         float* __restrict__ gw2, float* __restrict__ gb2, int N, int T, int C, int Cr, float inv_hw) {
  __shared__ float red[CRM][CA_T / SAVFI_WAVE];
  __shared__ float da[CRM], a1s[CRM], dz1s[CRM];
  const int t = blockIdx.x, c = threadIdx.x;
  const float* W1 = w1 + (size_t)t * Cr * C;
  const float* W2 = w2 + (size_t)t * C * Cr;
  // thread c owns channel c: its rows of the parameter gradients live in registers over the task's samples
  float g_w2[CRM], g_w1[CRM], g_b2 = 0.f, g_b1 = 0.f, w2r[CRM], w1r[CRM];
#pragma unroll
  for (int j = 0; j < CRM; ++j) {
    g_w2[j] = 0.f; g_w1[j] = 0.f;
    w2r[j] = (j < Cr && c < C) ? W2[(size_t)c * Cr + j] : 0.f;
    w1r[j] = (j < Cr && c < C) ? W1[(size_t)j * C + c] : 0.f;
  }
  for (int n = t; n < N; n += T) {
    if (c < Cr) a1s[c] = a1[(size_t)n * Cr + c];
    const float yv = c < C ? y[(size_t)n * C + c] : 0.f;
    const float dz2 = c < C ? r[(size_t)n * C + c] * yv * (1.f - yv) : 0.f;
    const float sv = c < C ? s[(size_t)n * C + c] : 0.f;
    float p[CRM];
#pragma unroll
    for (int j = 0; j < CRM; ++j) p[j] = w2r[j] * dz2;
    ca_block_sums(p, Cr, red, da);            // also publishes a1s (barrier inside)
    if (c < Cr) {
      const float d = a1s[c] > 0.f ? da[c] : 0.f;
      dz1s[c] = d;
      g_b1 += d;
    }
    __syncthreads();
    g_b2 += dz2;
    float v = 0.f;
#pragma unroll
    for (int j = 0; j < CRM; ++j)
      if (j < Cr) {
        g_w2[j] += dz2 * a1s[j];
        v += w1r[j] * dz1s[j];
        g_w1[j] += dz1s[j] * sv;
      }
    if (c < C) ds[(size_t)n * C + c] = v * inv_hw;
    __syncthreads();
  }
  if (c < C) {
    gb2[t * C + c] = g_b2;
#pragma unroll
    for (int j = 0; j < CRM; ++j)
      if (j < Cr) {
        gw2[(size_t)t * C * Cr + (size_t)c * Cr + j] = g_w2[j];
        gw1[(size_t)t * Cr * C + (size_t)j * C + c] = g_w1[j];
      }
  }
  if (c < Cr) gb1[t * Cr + c] = g_b1;
}

// out = a * y[plane] + (x ? x : ds[plane])        (forward: x = the skip connection; backward: the pooled-branch gradient)
__global__ __launch_bounds__(CA_T) void ca_apply_kernel(const float* __restrict__ a, const float* __restrict__ y, const float* __restrict__ x,
                                                        const float* __restrict__ ds, float* __restrict__ out, int hw, int chunks) {
  const int plane = blockIdx.x / chunks, chunk = blockIdx.x - plane * chunks;
  const float yv = y[plane], dv = ds ? ds[plane] : 0.f;
  const size_t base = (size_t)plane * hw;
  const int per = (hw + chunks - 1) / chunks;
  const int lo = chunk * per, hi = min(lo + per, hw);
  if (((base + lo) & 3) == 0) {
    const int n4 = (hi - lo) / 4;
    const float4* a4 = reinterpret_cast<const float4*>(a + base + lo);
    const float4* x4 = x ? reinterpret_cast<const float4*>(x + base + lo) : nullptr;
    float4* o4 = reinterpret_cast<float4*>(out + base + lo);
    for (int i = threadIdx.x; i < n4; i += CA_T) {
      const float4 v = a4[i];
      float4 u = x4 ? x4[i] : make_float4(dv, dv, dv, dv);
      o4[i] = make_float4(v.x * yv + u.x, v.y * yv + u.y, v.z * yv + u.z, v.w * yv + u.w);
    }
    for (int i = lo + 4 * n4 + threadIdx.x; i < hi; i += CA_T) out[base + i] = a[base + i] * yv + (x ? x[base + i] : dv);
  } else {
    for (int i = lo + threadIdx.x; i < hi; i += CA_T) out[base + i] = a[base + i] * yv + (x ? x[base + i] : dv);
  }
}

// Forward, hidden width <= 16 and C <= 256: the attention MLP inside the apply launch.  Every workgroup repeats ca_mlp_fwd_small's
// arithmetic for its sample (the same code on the same 256 threads: the same y bit for bit in every workgroup, 2 300 multiply-adds and
// 18 KB of weights from L2) and then scales its chunk of its plane; the workgroups of channel 0 keep a1, chunk 0 of every plane keeps y
// for the backward.  The MLP launch of its own was 10 us of latency between the pool and the apply, 180 times per C5 meta-iteration.
__global__ __launch_bounds__(CA_T) void ca_apply_mlp_kernel(const float* __restrict__ a, const float* __restrict__ s, const float* __restrict__ w1,
                                                            const float* __restrict__ b1, const float* __restrict__ w2, const float* __restrict__ b2,
                                                            const float* __restrict__ x, float* __restrict__ out, float* __restrict__ y,
                                                            float* __restrict__ a1, int hw, int chunks, int T, int C, int Cr) {
  __shared__ float red[CRM][CA_T / SAVFI_WAVE];
  __shared__ float hid[CRM];
  __shared__ float ysh;
  const int plane = blockIdx.x / chunks, chunk = blockIdx.x - plane * chunks;
  const int n = plane / C, cch = plane - n * C, t = n % T, c = threadIdx.x;
  const float* W1 = w1 + (size_t)t * Cr * C;
  const float* W2 = w2 + (size_t)t * C * Cr;
  const float sv = c < C ? s[(size_t)n * C + c] : 0.f;
  float p[CRM];
#pragma unroll
  for (int j = 0; j < CRM; ++j) p[j] = (j < Cr && c < C) ? W1[(size_t)j * C + c] * sv : 0.f;
  ca_block_sums(p, Cr, red, hid);
  if (c < Cr) {
    const float h = fmaxf(hid[c] + b1[t * Cr + c], 0.f);
    if (cch == 0 && chunk == 0) a1[(size_t)n * Cr + c] = h;
    hid[c] = h;
  }
  __syncthreads();
  if (c == cch) {
    float z = b2[t * C + c];
    for (int j = 0; j < Cr; ++j) z += W2[(size_t)c * Cr + j] * hid[j];
    const float yv = 1.f / (1.f + __expf(-z));
    if (chunk == 0) y[plane] = yv;
    ysh = yv;
  }
  __syncthreads();
  const float yv = ysh;
  const size_t base = (size_t)plane * hw;
  const int per = (hw + chunks - 1) / chunks;
  const int lo = chunk * per, hi = min(lo + per, hw);
  if (((base + lo) & 3) == 0) {
    const int n4 = (hi - lo) / 4;
    const float4* a4 = reinterpret_cast<const float4*>(a + base + lo);
    const float4* x4 = reinterpret_cast<const float4*>(x + base + lo);
    float4* o4 = reinterpret_cast<float4*>(out + base + lo);
    for (int i = threadIdx.x; i < n4; i += CA_T) {
      const float4 v = a4[i], u = x4[i];
      o4[i] = make_float4(v.x * yv + u.x, v.y * yv + u.y, v.z * yv + u.z, v.w * yv + u.w);
    }
    for (int i = lo + 4 * n4 + threadIdx.x; i < hi; i += CA_T) out[base + i] = a[base + i] * yv + x[base + i];
  } else {
    for (int i = lo + threadIdx.x; i < hi; i += CA_T) out[base + i] = a[base + i] * yv + x[base + i];
  }
}

// Backward of the same: gt = g * y + ds with ds recomputed per workgroup (ca_mlp_bwd_small's arithmetic for the workgroup's sample: the same
// ds bit for bit), and the four parameter gradients from T workgroups of their own AT THE FRONT of the grid (task t: the whole body of
// ca_mlp_bwd_small), which run beside the streaming workgroups instead of in a 10 us launch between the pool and the apply.
__global__ __launch_bounds__(CA_T) void ca_apply_bwd_mlp_kernel(const float* __restrict__ g, const float* __restrict__ r, const float* __restrict__ s,
                                                                const float* __restrict__ y, const float* __restrict__ a1, const float* __restrict__ w1,
                                                                const float* __restrict__ w2, float* __restrict__ gt, float* __restrict__ gw1,
                                                                float* __restrict__ gb1, float* __restrict__ gw2, float* __restrict__ gb2, int hw,
                                                                int chunks, int N, int T, int C, int Cr, float inv_hw) {
  __shared__ float red[CRM][CA_T / SAVFI_WAVE];
  __shared__ float da[CRM], a1s[CRM], dz1s[CRM];
  __shared__ float dsh;
  const int c = threadIdx.x;
  if ((int)blockIdx.x < T) {          // parameter gradients of task t (ca_mlp_bwd_small without the ds store)
    const int t = blockIdx.x;
    const float* W2 = w2 + (size_t)t * C * Cr;
    float g_w2[CRM], g_w1[CRM], g_b2 = 0.f, g_b1 = 0.f, w2r[CRM];
#pragma unroll
    for (int j = 0; j < CRM; ++j) {
      g_w2[j] = 0.f; g_w1[j] = 0.f;
      w2r[j] = (j < Cr && c < C) ? W2[(size_t)c * Cr + j] : 0.f;
    }
    for (int n = t; n < N; n += T) {
      if (c < Cr) a1s[c] = a1[(size_t)n * Cr + c];
      const float yv = c < C ? y[(size_t)n * C + c] : 0.f;
      const float dz2 = c < C ? r[(size_t)n * C + c] * yv * (1.f - yv) : 0.f;
      const float sv = c < C ? s[(size_t)n * C + c] : 0.f;
      float p[CRM];
#pragma unroll
      for (int j = 0; j < CRM; ++j) p[j] = w2r[j] * dz2;
      ca_block_sums(p, Cr, red, da);
      if (c < Cr) {
        const float d = a1s[c] > 0.f ? da[c] : 0.f;
        dz1s[c] = d;
        g_b1 += d;
      }
      __syncthreads();
      g_b2 += dz2;
#pragma unroll
      for (int j = 0; j < CRM; ++j)
        if (j < Cr) {
          g_w2[j] += dz2 * a1s[j];
          g_w1[j] += dz1s[j] * sv;
        }
      __syncthreads();
    }
    if (c < C) {
      gb2[t * C + c] = g_b2;
#pragma unroll
      for (int j = 0; j < CRM; ++j)
        if (j < Cr) {
          gw2[(size_t)t * C * Cr + (size_t)c * Cr + j] = g_w2[j];
          gw1[(size_t)t * Cr * C + (size_t)j * C + c] = g_w1[j];
        }
    }
    if (c < Cr) gb1[t * Cr + c] = g_b1;
    return;
  }
  const int blk = blockIdx.x - T;
  const int plane = blk / chunks, chunk = blk - plane * chunks;
  const int n = plane / C, cch = plane - n * C, t = n % T;
  const float* W1 = w1 + (size_t)t * Cr * C;
  const float* W2 = w2 + (size_t)t * C * Cr;
  if (c < Cr) a1s[c] = a1[(size_t)n * Cr + c];
  const float yc = c < C ? y[(size_t)n * C + c] : 0.f;
  const float dz2 = c < C ? r[(size_t)n * C + c] * yc * (1.f - yc) : 0.f;
  float p[CRM];
#pragma unroll
  for (int j = 0; j < CRM; ++j) p[j] = (j < Cr && c < C) ? W2[(size_t)c * Cr + j] * dz2 : 0.f;
  ca_block_sums(p, Cr, red, da);            // also publishes a1s (barrier inside)
  if (c == cch) {
    float v = 0.f;
    for (int j = 0; j < Cr; ++j) v += W1[(size_t)j * C + c] * (a1s[j] > 0.f ? da[j] : 0.f);
    dsh = v * inv_hw;
  }
  __syncthreads();
  const float yv = y[plane], dv = dsh;
  const size_t base = (size_t)plane * hw;
  const int per = (hw + chunks - 1) / chunks;
  const int lo = chunk * per, hi = min(lo + per, hw);
  if (((base + lo) & 3) == 0) {
    const int n4 = (hi - lo) / 4;
    const float4* a4 = reinterpret_cast<const float4*>(g + base + lo);
    float4* o4 = reinterpret_cast<float4*>(gt + base + lo);
    for (int i = threadIdx.x; i < n4; i += CA_T) {
      const float4 v = a4[i];
      o4[i] = make_float4(v.x * yv + dv, v.y * yv + dv, v.z * yv + dv, v.w * yv + dv);
    }
    for (int i = lo + 4 * n4 + threadIdx.x; i < hi; i += CA_T) gt[base + i] = g[base + i] * yv + dv;
  } else {
    for (int i = lo + threadIdx.x; i < hi; i += CA_T) gt[base + i] = g[base + i] * yv + dv;
  }
}

inline int ca_chunks(int64_t planes, int hw) {
  // enough workgroups to fill the GPU, at least 4096 elements each
  int64_t want = (2048 + planes - 1) / planes;
  int64_t maxc = (hw + 4095) / 4096;
  int c = (int)(want < maxc ? want : maxc);
  return c < 1 ? 1 : c;
}

}  // namespace

extern "C" int savfi_ca_pool_f32(const float* a, const float* b, float* s, int64_t planes, int hw, float scale, void* stream) {
  if (!a || !s) return SAVFI_E_NULL;
  if (planes <= 0 || hw <= 0) return SAVFI_E_SHAPE;
  if (planes >= (1ll << 31)) return SAVFI_E_TOOBIG;
  if (hw >= 4096) hipLaunchKernelGGL(ca_pool_kernel<1024>, dim3((unsigned)planes), dim3(1024), 0, (hipStream_t)stream, a, b, s, hw, scale);
  else hipLaunchKernelGGL(ca_pool_kernel<CA_T>, dim3((unsigned)planes), dim3(CA_T), 0, (hipStream_t)stream, a, b, s, hw, scale);
  return savfi_launch_status();
}

extern "C" int savfi_ca_mlp_fwd_f32(const float* s, const float* w1, const float* b1, const float* w2, const float* b2, float* y, float* a1,
                                    int N, int T, int C, int Cr, void* stream) {
  if (!s || !w1 || !b1 || !w2 || !b2 || !y || !a1) return SAVFI_E_NULL;
  if (N <= 0 || T <= 0 || N % T != 0 || C <= 0 || Cr <= 0) return SAVFI_E_SHAPE;
  if (C > 1024 || Cr > 64) return SAVFI_E_UNSUPPORTED;
  if (Cr <= CRM && C <= CA_T) hipLaunchKernelGGL(ca_mlp_fwd_small, dim3(N), dim3(CA_T), 0, (hipStream_t)stream, s, w1, b1, w2, b2, y, a1, T, C, Cr);
  else hipLaunchKernelGGL(ca_mlp_fwd_kernel, dim3(N), dim3(CA_T), 0, (hipStream_t)stream, s, w1, b1, w2, b2, y, a1, T, C, Cr);
  return savfi_launch_status();
}

extern "C" int savfi_ca_mlp_bwd_f32(const float* r, const float* s, const float* y, const float* a1, const float* w1, const float* w2,
                                    float* ds, float* gw1, float* gb1, float* gw2, float* gb2, int N, int T, int C, int Cr, float inv_hw,
                                    void* stream) {
  if (!r || !s || !y || !a1 || !w1 || !w2 || !ds || !gw1 || !gb1 || !gw2 || !gb2) return SAVFI_E_NULL;
  if (N <= 0 || T <= 0 || N % T != 0 || C <= 0 || Cr <= 0) return SAVFI_E_SHAPE;
  if (C > 1024 || Cr > 64) return SAVFI_E_UNSUPPORTED;
  if (Cr <= CRM && C <= CA_T)
    hipLaunchKernelGGL(ca_mlp_bwd_small, dim3(T), dim3(CA_T), 0, (hipStream_t)stream, r, s, y, a1, w1, w2, ds, gw1, gb1, gw2, gb2, N, T, C, Cr, inv_hw);
  else
    hipLaunchKernelGGL(ca_mlp_bwd_kernel, dim3(T), dim3(CA_T), 0, (hipStream_t)stream, r, s, y, a1, w1, w2, ds, gw1, gb1, gw2, gb2, N, T, C, Cr, inv_hw);
  return savfi_launch_status();
}

extern "C" int savfi_ca_apply_mlp_f32(const float* a, const float* s, const float* w1, const float* b1, const float* w2, const float* b2,
                                      const float* x, float* out, float* y, float* a1, int N, int T, int C, int Cr, int hw, void* stream) {
  if (!a || !s || !w1 || !b1 || !w2 || !b2 || !x || !out || !y || !a1) return SAVFI_E_NULL;
  if (N <= 0 || T <= 0 || N % T != 0 || C <= 0 || Cr <= 0 || hw <= 0) return SAVFI_E_SHAPE;
  if (Cr > CRM || C > CA_T) return SAVFI_E_UNSUPPORTED;       // the separate launches (savfi_ca_mlp_fwd_f32 + savfi_ca_apply_f32)
  const int64_t planes = (int64_t)N * C;
  const int chunks = ca_chunks(planes, hw);
  if (planes * chunks >= (1ll << 31)) return SAVFI_E_TOOBIG;
  hipLaunchKernelGGL(ca_apply_mlp_kernel, dim3((unsigned)(planes * chunks)), dim3(CA_T), 0, (hipStream_t)stream, a, s, w1, b1, w2, b2, x, out, y, a1,
                     hw, chunks, T, C, Cr);
  return savfi_launch_status();
}

extern "C" int savfi_ca_apply_bwd_mlp_f32(const float* g, const float* r, const float* s, const float* y, const float* a1, const float* w1,
                                          const float* w2, float* gt, float* gw1, float* gb1, float* gw2, float* gb2, int N, int T, int C, int Cr,
                                          int hw, void* stream) {
  if (!g || !r || !s || !y || !a1 || !w1 || !w2 || !gt || !gw1 || !gb1 || !gw2 || !gb2) return SAVFI_E_NULL;
  if (N <= 0 || T <= 0 || N % T != 0 || C <= 0 || Cr <= 0 || hw <= 0) return SAVFI_E_SHAPE;
  if (Cr > CRM || C > CA_T) return SAVFI_E_UNSUPPORTED;       // the separate launches (savfi_ca_mlp_bwd_f32 + savfi_ca_apply_f32)
  const int64_t planes = (int64_t)N * C;
  const int chunks = ca_chunks(planes, hw);
  if (planes * chunks + T >= (1ll << 31)) return SAVFI_E_TOOBIG;
  hipLaunchKernelGGL(ca_apply_bwd_mlp_kernel, dim3((unsigned)(planes * chunks + T)), dim3(CA_T), 0, (hipStream_t)stream, g, r, s, y, a1, w1, w2, gt,
                     gw1, gb1, gw2, gb2, hw, chunks, N, T, C, Cr, 1.f / (float)hw);
  return savfi_launch_status();
}

extern "C" int savfi_ca_apply_f32(const float* a, const float* y, const float* x, const float* ds, float* out, int64_t planes, int hw,
                                  void* stream) {
  if (!a || !y || !out || (!x && !ds)) return SAVFI_E_NULL;
  if (planes <= 0 || hw <= 0) return SAVFI_E_SHAPE;
  const int chunks = ca_chunks(planes, hw);
  if (planes * chunks >= (1ll << 31)) return SAVFI_E_TOOBIG;
  hipLaunchKernelGGL(ca_apply_kernel, dim3((unsigned)(planes * chunks)), dim3(CA_T), 0, (hipStream_t)stream, a, y, x, ds, out, hw, chunks);
  return savfi_launch_status();
}
