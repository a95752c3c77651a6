// Fused multi-tensor kernels for the MAML inner loop on gfx950:
//   * LSLR / Meta-SGD per-parameter update (SGD, Adam, Adamax-as-implemented) + its lr-gradient
//   * L2F per-tensor gradient mean and per-tensor attenuation scale (+ its backward)
//
// Replaces the per-tensor Python loops of the reference, which issue 2..12 tiny launches per
// tensor (inner_loop_optimizers.py:136-244, :324-425; meta_learning_system.py:249-253, :267-268)
// -- up to ~6k launches per inner step for CAIN's 494 tensors.
//
// Pure HBM streaming.  A launch carries a table of up to SAVFI_MT_MAX_TENSORS tensors BY VALUE in
// its kernel arguments (no device-side descriptor buffer, no H2D copy, nothing to keep alive), the
// tensors are cut into 4096-element chunks and the grid is one workgroup per chunk, so the load is
// balanced whatever the mix of 64-element biases and 2.4M-element conv weights.  A workgroup finds
// its tensor with a wave-uniform binary search over the chunk prefix table (scalar loads from the
// kernarg segment), then streams float4 per lane: every access is a 1 KiB coalesced wave segment.
#include "common.h"

namespace {

constexpr int MAXT = SAVFI_MT_MAX_TENSORS;
constexpr int NPTR = 7;
constexpr int NT = 256;
constexpr int CHUNK = 4096;

struct MtTable {
  void* p[NPTR][MAXT];
  int numel[MAXT];
  int chunk_start[MAXT + 1];
  float f0[MAXT], f1[MAXT];
  unsigned long long vec_ok;  // bit t: every pointer of tensor t is 16-byte aligned
  int n;
};
static_assert(sizeof(MtTable) <= 4096 - 64, "kernel argument block must stay under 4 KiB");

__device__ __forceinline__ int find_tensor(const MtTable& tb, int blk) {
  int lo = 0, hi = tb.n - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (tb.chunk_start[mid] <= blk) lo = mid; else hi = mid - 1;
  }
  return lo;
}

// Generic chunk walker: F(e) for scalar tail / unaligned, F4(e) for 4 consecutive elements.
template <typename F, typename F4>
__device__ __forceinline__ void walk_chunk(const MtTable& tb, int t, int blk, F f, F4 f4) {
  const int n = tb.numel[t];
  const int base = (blk - tb.chunk_start[t]) * CHUNK;
  const int end = min(base + CHUNK, n);
  if ((tb.vec_ok >> t) & 1ull) {
    const int vend = base + ((end - base) & ~3);
    for (int e = base + 4 * (int)threadIdx.x; e < vend; e += 4 * NT) f4(e);
    for (int e = vend + (int)threadIdx.x; e < end; e += NT) f(e);
  } else {
    for (int e = base + (int)threadIdx.x; e < end; e += NT) f(e);
  }
}

#define LD4(ptr, e) (*reinterpret_cast<const float4*>((ptr) + (e)))
#define ST4(ptr, e, val) (*reinterpret_cast<float4*>((ptr) + (e)) = (val))

struct Hyper { float b1, b2, omb1, omb2, eps; };  // omb = 1 - beta, rounded from double like the reference's Python floats

// One element of the rule.  Returns w'; `dir` receives d w' / d lr.
template <int RULE>
__device__ __forceinline__ float rule_elem(float w, float g, float lr, float* m, float* s, float bc1,
                                           float bc2s /* sqrt(1-b2^step) */, Hyper hp, float& dir) {
  if (RULE == SAVFI_RULE_SGD) {
    dir = -g;
    return w - lr * g;
  } else if (RULE == SAVFI_RULE_ADAM) {
    const float mn = hp.b1 * (*m) + hp.omb1 * g;
    const float sn = hp.b2 * (*s) + hp.omb2 * g * g;
    *m = mn; *s = sn;
    const float denom = sqrtf(sn) / bc2s + hp.eps;
    dir = -(mn / bc1) / denom;
    return w - ((lr / bc1) * mn) / denom;
  } else if (RULE == SAVFI_RULE_ADAMAX_LSLR) {
    const float mn = hp.b1 * (*m) + hp.omb1 * g;
    *m = mn;
    const float u = fabsf(g) + hp.eps;
    dir = -(mn / bc1) / u;
    return w - ((lr / bc1) * mn) / u;
  } else {  // SAVFI_RULE_ADAMAX_MSGD
    const float mn = hp.omb1 * g;
    const float u = fabsf(g) + hp.eps;
    dir = -(mn / bc1) / u;
    return w - ((lr / bc1) * mn) / u;
  }
}

template <int RULE, int LRMODE, bool COEF>
__global__ __launch_bounds__(NT) void mt_update_kernel(MtTable tb, Hyper hp) {
  const int blk = blockIdx.x;
  const int t = find_tensor(tb, blk);
  const float* w = (const float*)tb.p[0][t];
  const float* g = (const float*)tb.p[1][t];
  const float* lr = (const float*)tb.p[2][t];
  float* m = (float*)tb.p[3][t];
  float* s = (float*)tb.p[4][t];
  float* out = (float*)tb.p[5][t];
  float* coef = (float*)tb.p[6][t];
  const float bc1 = tb.f0[t], bc2s = tb.f1[t];
  const float lr_s = (LRMODE == SAVFI_LR_SCALAR) ? lr[0] : 0.f;
  constexpr bool USE_M = (RULE == SAVFI_RULE_ADAM || RULE == SAVFI_RULE_ADAMAX_LSLR);
  constexpr bool USE_S = (RULE == SAVFI_RULE_ADAM);

  auto one = [&](int e) {
    float mm = USE_M ? m[e] : 0.f, ss = USE_S ? s[e] : 0.f, d;
    const float l = (LRMODE == SAVFI_LR_SCALAR) ? lr_s : lr[e];
    out[e] = rule_elem<RULE>(w[e], g[e], l, &mm, &ss, bc1, bc2s, hp, d);
    if (USE_M) m[e] = mm;
    if (USE_S) s[e] = ss;
    if (COEF) coef[e] = d;
  };
  auto four = [&](int e) {
    const float4 w4 = LD4(w, e), g4 = LD4(g, e);
    float4 l4 = make_float4(lr_s, lr_s, lr_s, lr_s);
    if (LRMODE == SAVFI_LR_ELEMENT) l4 = LD4(lr, e);
    float4 m4 = make_float4(0, 0, 0, 0), s4 = m4, o4, d4;
    if (USE_M) m4 = LD4(m, e);
    if (USE_S) s4 = LD4(s, e);
    o4.x = rule_elem<RULE>(w4.x, g4.x, l4.x, &m4.x, &s4.x, bc1, bc2s, hp, d4.x);
    o4.y = rule_elem<RULE>(w4.y, g4.y, l4.y, &m4.y, &s4.y, bc1, bc2s, hp, d4.y);
    o4.z = rule_elem<RULE>(w4.z, g4.z, l4.z, &m4.z, &s4.z, bc1, bc2s, hp, d4.z);
    o4.w = rule_elem<RULE>(w4.w, g4.w, l4.w, &m4.w, &s4.w, bc1, bc2s, hp, d4.w);
    ST4(out, e, o4);
    if (USE_M) ST4(m, e, m4);
    if (USE_S) ST4(s, e, s4);
    if (COEF) ST4(coef, e, d4);
  };
  walk_chunk(tb, t, blk, one, four);
}

// g_lr = scale * g_out * dir   (ELEMENT)   or   g_lr[0] += scale * sum(g_out * dir)   (SCALAR)
template <int LRMODE>
__global__ __launch_bounds__(NT) void mt_update_bwd_kernel(MtTable tb, float scale) {
  __shared__ float red[NT / SAVFI_WAVE];
  const int blk = blockIdx.x;
  const int t = find_tensor(tb, blk);
  const float* go = (const float*)tb.p[0][t];
  const float* dir = (const float*)tb.p[1][t];
  float* glr = (float*)tb.p[2][t];
  float acc = 0.f;
  auto one = [&](int e) {
    const float v = scale * go[e] * dir[e];
    if (LRMODE == SAVFI_LR_ELEMENT) glr[e] = v; else acc += v;
  };
  auto four = [&](int e) {
    const float4 a = LD4(go, e), b = LD4(dir, e);
    const float4 v = make_float4(scale * a.x * b.x, scale * a.y * b.y, scale * a.z * b.z, scale * a.w * b.w);
    if (LRMODE == SAVFI_LR_ELEMENT) ST4(glr, e, v); else acc += (v.x + v.y) + (v.z + v.w);
  };
  walk_chunk(tb, t, blk, one, four);
  if (LRMODE == SAVFI_LR_SCALAR) {
    const float tot = block_sum<NT / SAVFI_WAVE>(acc, red);
    if (threadIdx.x == 0) atomicAdd(glr, tot);
  }
}

// out_vec[t] += sum(chunk) / numel
__global__ __launch_bounds__(NT) void mt_mean_kernel(MtTable tb, float* __restrict__ out_vec) {
  __shared__ float red[NT / SAVFI_WAVE];
  const int blk = blockIdx.x;
  const int t = find_tensor(tb, blk);
  const float* x = (const float*)tb.p[0][t];
  float acc = 0.f;
  auto one = [&](int e) { acc += x[e]; };
  auto four = [&](int e) { const float4 a = LD4(x, e); acc += (a.x + a.y) + (a.z + a.w); };
  walk_chunk(tb, t, blk, one, four);
  const float tot = block_sum<NT / SAVFI_WAVE>(acc, red);
  if (threadIdx.x == 0) atomicAdd(&out_vec[t], tot / (float)tb.numel[t]);
}

// out = gamma[t] * w
__global__ __launch_bounds__(NT) void mt_scale_kernel(MtTable tb, const float* __restrict__ gamma) {
  const int blk = blockIdx.x;
  const int t = find_tensor(tb, blk);
  const float* w = (const float*)tb.p[0][t];
  float* out = (float*)tb.p[1][t];
  const float gm = gamma[t];
  auto one = [&](int e) { out[e] = gm * w[e]; };
  auto four = [&](int e) {
    const float4 a = LD4(w, e);
    ST4(out, e, make_float4(gm * a.x, gm * a.y, gm * a.z, gm * a.w));
  };
  walk_chunk(tb, t, blk, one, four);
}

// g_w = gamma[t] * g_out ; g_gamma[t] += sum(g_out * w)
__global__ __launch_bounds__(NT) void mt_scale_bwd_kernel(MtTable tb, const float* __restrict__ gamma,
                                                          float* __restrict__ g_gamma) {
  __shared__ float red[NT / SAVFI_WAVE];
  const int blk = blockIdx.x;
  const int t = find_tensor(tb, blk);
  const float* go = (const float*)tb.p[0][t];
  const float* w = (const float*)tb.p[1][t];
  float* gw = (float*)tb.p[2][t];
  const float gm = gamma[t];
  float acc = 0.f;
  auto one = [&](int e) {
    const float a = go[e];
    if (gw) gw[e] = gm * a;
    acc += a * w[e];
  };
  auto four = [&](int e) {
    const float4 a = LD4(go, e), b = LD4(w, e);
    if (gw) ST4(gw, e, make_float4(gm * a.x, gm * a.y, gm * a.z, gm * a.w));
    acc += (a.x * b.x + a.y * b.y) + (a.z * b.z + a.w * b.w);
  };
  walk_chunk(tb, t, blk, one, four);
  const float tot = block_sum<NT / SAVFI_WAVE>(acc, red);
  if (threadIdx.x == 0 && g_gamma) atomicAdd(&g_gamma[t], tot);
}

// ---------------------------------------------------------------------------------------------
// host side: cut [0, n) into groups of <= MAXT tensors and launch one grid per group.
// ---------------------------------------------------------------------------------------------
struct Group {
  MtTable tb;
  int blocks;
  int first;  // index of the first tensor of the group in the caller's arrays
};

// ptrs[k] may be NULL (whole array absent) and ptrs[k][i] may be NULL (tensor lacks that operand).
template <typename Launch>
int for_each_group(int n, const void* const* const* ptrs, int nptr, const int64_t* numel, const float* f0,
                   const float* f1, Launch launch) {
  if (n < 0) return SAVFI_E_SHAPE;
  if (n > 0 && !numel) return SAVFI_E_NULL;
  int i = 0;
  while (i < n) {
    Group gr;
    gr.first = i;
    MtTable& tb = gr.tb;
    tb.vec_ok = 0ull;
    int cnt = 0, blocks = 0;
    for (; i < n && cnt < MAXT; ++i) {
      if (numel[i] < 0 || numel[i] > 0x7fffffffLL - CHUNK) return SAVFI_E_TOOBIG;
      if (numel[i] == 0) continue;  // empty tensors contribute nothing
      bool aligned = true;
      for (int k = 0; k < NPTR; ++k) {
        const void* q = (k < nptr && ptrs[k]) ? ptrs[k][i] : nullptr;
        tb.p[k][cnt] = const_cast<void*>(q);
        if (q && ((uintptr_t)q & 15u)) aligned = false;
      }
      tb.numel[cnt] = (int)numel[i];
      tb.chunk_start[cnt] = blocks;
      tb.f0[cnt] = f0 ? f0[i] : 1.f;
      tb.f1[cnt] = f1 ? f1[i] : 1.f;
      if (aligned) tb.vec_ok |= (1ull << cnt);
      blocks += savfi_cdiv(numel[i], CHUNK);
      // a tensor must not straddle groups: it occupies one slot; stop if the next would not fit
      ++cnt;
    }
    tb.chunk_start[cnt] = blocks;
    tb.n = cnt;
    gr.blocks = blocks;
    if (cnt == 0) continue;
    if (int e = launch(gr)) return e;
  }
  return SAVFI_OK;
}

}  // namespace

extern "C" int savfi_mt_update_f32(int rule, int lr_mode, int n, const float* const* w,
                                   const float* const* g, const float* const* lr, float* const* m,
                                   float* const* s, float* const* out, float* const* coef,
                                   const int64_t* numel, const float* bc1, const float* sqrt_bc2, double beta1,
                                   double beta2, double eps, void* stream) {
  if (n == 0) return SAVFI_OK;
  if (!w || !g || !lr || !out) return SAVFI_E_NULL;
  if (rule < 0 || rule > 3 || (lr_mode != SAVFI_LR_SCALAR && lr_mode != SAVFI_LR_ELEMENT))
    return SAVFI_E_UNSUPPORTED;
  if ((rule == SAVFI_RULE_ADAM && (!m || !s)) || (rule == SAVFI_RULE_ADAMAX_LSLR && !m)) return SAVFI_E_NULL;
  if (rule != SAVFI_RULE_SGD && !bc1) return SAVFI_E_NULL;
  if (rule == SAVFI_RULE_ADAM && !sqrt_bc2) return SAVFI_E_NULL;
  for (int i = 0; i < n; ++i) {
    if (numel && numel[i] > 0 && (!w[i] || !g[i] || !lr[i] || !out[i])) return SAVFI_E_NULL;
  }
  const void* const* ptrs[NPTR] = {(const void* const*)w, (const void* const*)g, (const void* const*)lr,
                                   (const void* const*)m, (const void* const*)s, (const void* const*)out,
                                   (const void* const*)coef};
  const Hyper hp{(float)beta1, (float)beta2, (float)(1.0 - beta1), (float)(1.0 - beta2), (float)eps};
  hipStream_t st = (hipStream_t)stream;
  const bool want_coef = coef != nullptr;
  return for_each_group(n, ptrs, NPTR, numel, bc1, sqrt_bc2, [&](Group& gr) -> int {
#define SAVFI_MT_CASE(R, L)                                                                            \
  if (rule == R && lr_mode == L) {                                                                     \
    if (want_coef)                                                                                     \
      hipLaunchKernelGGL((mt_update_kernel<R, L, true>), dim3(gr.blocks), dim3(NT), 0, st, gr.tb, hp); \
    else                                                                                               \
      hipLaunchKernelGGL((mt_update_kernel<R, L, false>), dim3(gr.blocks), dim3(NT), 0, st, gr.tb, hp); \
  }
    SAVFI_MT_CASE(SAVFI_RULE_SGD, SAVFI_LR_SCALAR)
    SAVFI_MT_CASE(SAVFI_RULE_SGD, SAVFI_LR_ELEMENT)
    SAVFI_MT_CASE(SAVFI_RULE_ADAM, SAVFI_LR_SCALAR)
    SAVFI_MT_CASE(SAVFI_RULE_ADAM, SAVFI_LR_ELEMENT)
    SAVFI_MT_CASE(SAVFI_RULE_ADAMAX_LSLR, SAVFI_LR_SCALAR)
    SAVFI_MT_CASE(SAVFI_RULE_ADAMAX_LSLR, SAVFI_LR_ELEMENT)
    SAVFI_MT_CASE(SAVFI_RULE_ADAMAX_MSGD, SAVFI_LR_SCALAR)
    SAVFI_MT_CASE(SAVFI_RULE_ADAMAX_MSGD, SAVFI_LR_ELEMENT)
#undef SAVFI_MT_CASE
    return savfi_launch_status();
  });
}

extern "C" int savfi_mt_update_bwd_f32(int lr_mode, int n, const float* const* g_out,
                                       const float* const* dir, float* const* g_lr, const int64_t* numel,
                                       float scale, void* stream) {
  if (n == 0) return SAVFI_OK;
  if (!g_out || !dir || !g_lr) return SAVFI_E_NULL;
  if (lr_mode != SAVFI_LR_SCALAR && lr_mode != SAVFI_LR_ELEMENT) return SAVFI_E_UNSUPPORTED;
  for (int i = 0; i < n; ++i)
    if (numel && numel[i] > 0 && (!g_out[i] || !dir[i] || !g_lr[i])) return SAVFI_E_NULL;
  const void* const* ptrs[3] = {(const void* const*)g_out, (const void* const*)dir, (const void* const*)g_lr};
  hipStream_t st = (hipStream_t)stream;
  if (lr_mode == SAVFI_LR_SCALAR) {
    // the scalar destination is a single float: keep it out of the alignment test
    const void* const* p2[2] = {ptrs[0], ptrs[1]};
    return for_each_group(n, p2, 2, numel, nullptr, nullptr, [&](Group& gr) -> int {
      // attach the destinations, skipping empty tensors exactly like for_each_group did
      int slot = 0;
      for (int i = gr.first; slot < gr.tb.n; ++i) {
        if (numel[i] == 0) continue;
        gr.tb.p[2][slot++] = (void*)g_lr[i];
      }
      hipLaunchKernelGGL(mt_update_bwd_kernel<SAVFI_LR_SCALAR>, dim3(gr.blocks), dim3(NT), 0, st, gr.tb, scale);
      return savfi_launch_status();
    });
  }
  return for_each_group(n, ptrs, 3, numel, nullptr, nullptr, [&](Group& gr) -> int {
    hipLaunchKernelGGL(mt_update_bwd_kernel<SAVFI_LR_ELEMENT>, dim3(gr.blocks), dim3(NT), 0, st, gr.tb, scale);
    return savfi_launch_status();
  });
}

extern "C" int savfi_mt_mean_f32(int n, const float* const* x, const int64_t* numel, float* out_vec,
                                 void* stream) {
  if (n == 0) return SAVFI_OK;
  if (!x || !out_vec) return SAVFI_E_NULL;
  for (int i = 0; i < n; ++i)
    if (numel && (numel[i] <= 0 || !x[i])) return numel[i] <= 0 ? SAVFI_E_SHAPE : SAVFI_E_NULL;
  const void* const* ptrs[1] = {(const void* const*)x};
  hipStream_t st = (hipStream_t)stream;
  return for_each_group(n, ptrs, 1, numel, nullptr, nullptr, [&](Group& gr) -> int {
    hipLaunchKernelGGL(mt_mean_kernel, dim3(gr.blocks), dim3(NT), 0, st, gr.tb, out_vec + gr.first);
    return savfi_launch_status();
  });
}

extern "C" int savfi_mt_scale_f32(int n, const float* const* w, const float* gamma, float* const* out,
                                  const int64_t* numel, void* stream) {
  if (n == 0) return SAVFI_OK;
  if (!w || !gamma || !out) return SAVFI_E_NULL;
  for (int i = 0; i < n; ++i)
    if (numel && (numel[i] <= 0 || !w[i] || !out[i])) return numel[i] <= 0 ? SAVFI_E_SHAPE : SAVFI_E_NULL;
  const void* const* ptrs[2] = {(const void* const*)w, (const void* const*)out};
  hipStream_t st = (hipStream_t)stream;
  return for_each_group(n, ptrs, 2, numel, nullptr, nullptr, [&](Group& gr) -> int {
    hipLaunchKernelGGL(mt_scale_kernel, dim3(gr.blocks), dim3(NT), 0, st, gr.tb, gamma + gr.first);
    return savfi_launch_status();
  });
}

extern "C" int savfi_mt_scale_bwd_f32(int n, const float* const* g_out, const float* const* w,
                                      const float* gamma, float* const* g_w, float* g_gamma,
                                      const int64_t* numel, void* stream) {
  if (n == 0) return SAVFI_OK;
  if (!g_out || !w || !gamma) return SAVFI_E_NULL;
  for (int i = 0; i < n; ++i)
    if (numel && (numel[i] <= 0 || !g_out[i] || !w[i])) return numel[i] <= 0 ? SAVFI_E_SHAPE : SAVFI_E_NULL;
  const void* const* ptrs[3] = {(const void* const*)g_out, (const void* const*)w, (const void* const*)g_w};
  hipStream_t st = (hipStream_t)stream;
  return for_each_group(n, ptrs, 3, numel, nullptr, nullptr, [&](Group& gr) -> int {
    hipLaunchKernelGGL(mt_scale_bwd_kernel, dim3(gr.blocks), dim3(NT), 0, st, gr.tb, gamma + gr.first,
                       g_gamma ? g_gamma + gr.first : nullptr);
    return savfi_launch_status();
  });
}
