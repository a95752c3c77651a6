// 3x3 / stride 1 / pad 1 convolution (forward and data gradient) for gfx950: Winograd F(2x2, 3x3) with the
// 16 batched GEMMs on the exact-fp32 matrix cores (v_mfma_f32_16x16x4_f32), fused in ONE kernel.
//
// Why: after the sepconv op, the SepConv / CAIN / VoxelFlow backbones are 3x3 convolutions (SURVEY.md 8(a) rows
// 13, 15, 17: sepconv/model.py:172-245, cain/model.py + model_utils.py:957-1053, voxel_flow.py:357-470); in the
// round-1 profile MIOpen's fp32 Winograd kernel (VALU) is ~50 % of the inner step at 80-100 direct-equivalent
// TFLOP/s.  Winograd cuts the multiplies 2.25x; putting them on the MFMA pipe (157 TFLOP/s fp32, the VALU's own
// rate, but fed from LDS at 1 dword per 1024 FMA) lifts the ceiling to ~350 direct-equivalent TFLOP/s.
//
//   Y = A^T [ (G g G^T) (.) (B^T d B) ] A          per 2x2 output tile, d = its 4x4 input patch
//   U[xi][ci][co] = (G g G^T)[xi]                   filter transform, a tiny pre-pass (wino_filter_transform)
//   V[xi][ci][t]  = (B^T d B)[xi]                   input transform, in registers, one (tile, ci) per thread
//   M[xi][co][t]  = sum_ci U[xi][ci][co] V[xi][ci][t]      16 GEMMs -> MFMA 16x16x4 f32
//
// Workgroup = 512 threads = 8 waves; it owns 64 tiles (4 x 16 tiles = 8 x 32 output pixels) x 64 output channels
// and walks the input channels in chunks of 8.  Wave w owns xi in {2w, 2w+1} for all 64 co x 64 tiles:
// 2 x 16 accumulator tiles = 128 registers.  Per chunk: every thread loads one 4x4 patch from global (issued one
// chunk ahead), transforms it in registers, writes 16 values to the V buffer in LDS (double buffered: one
// barrier per chunk); the A fragments (U) come straight from global / L2, also one chunk ahead.  Per k-step a
// wave reads 4 B fragments from LDS and holds 4 A fragments for 16 MFMAs.  After the channel loop the
// accumulators go through LDS 16 output channels at a time, and each thread finishes two (co, tile) pairs:
// A^T M A, + bias, (leaky) ReLU, 2x2 store.
//
// The data gradient is the same kernel on the flipped / transposed filter (wino_filter_transform mode 1).
#include "common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int WNT = 512;            // threads
constexpr int TBH = 4, TBW = 16;    // tiles per workgroup (rows x cols) -> 8 x 32 output pixels
static_assert(TBH * TBW == 64, "one tile per lane");
constexpr int COB = 64;             // output channels per workgroup
constexpr int CIB = 8;              // input channels per chunk (2 MFMA k-steps)
constexpr int VS = 80;              // pitch of a V row [ci] in floats: 64 tiles + 16 -> the 4 k-groups hit disjoint banks
constexpr int XS = 68;              // pitch of an exchange row [co] (output stage)
constexpr int VBUF = 16 * CIB * VS; // floats per V buffer
constexpr int XBUF = 16 * 16 * XS;  // floats of the exchange buffer (16 xi x 16 co)
constexpr int LDS_FLOATS = (2 * VBUF > XBUF) ? 2 * VBUF : XBUF;

// ---- filter transform ------------------------------------------------------------------------------------
// U = G g G^T for reduction channel k < KP and produced channel i < IP (zero padded), stored in MFMA A-fragment
// order so that a lane fetches the fragments of its 4 channel blocks with ONE aligned 16-byte load:
//   Uf[xi][k / 8][(k % 8) / 4][i / 64][lane = (k % 4) * 16 + i % 16][(i % 64) / 16]
// mode 0 (forward):        g[a][b] = w[i][k][a][b]           (w is [Co][Ci][3][3]; i = co, k = ci)
// mode 1 (data gradient):  g[a][b] = w[k][i][2-a][2-b]       (i = ci, k = co)
__global__ __launch_bounds__(256) void wino_filter_transform(const float* __restrict__ w, float* __restrict__ U,
                                                             int Co, int Ci, int K, int I, int KP, int IP, int mode) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= KP * IP) return;
  const int k = idx / IP, i = idx - k * IP;
  float g[3][3];
#pragma unroll
  for (int a = 0; a < 3; ++a)
#pragma unroll
    for (int b = 0; b < 3; ++b) {
      float val = 0.f;
      if (k < K && i < I)
        val = mode == 0 ? w[((size_t)i * Ci + k) * 9 + a * 3 + b] : w[((size_t)k * Ci + i) * 9 + (2 - a) * 3 + (2 - b)];
      g[a][b] = val;
    }
  // G g : 4 x 3
  float t[4][3];
#pragma unroll
  for (int b = 0; b < 3; ++b) {
    t[0][b] = g[0][b];
    t[1][b] = 0.5f * (g[0][b] + g[1][b] + g[2][b]);
    t[2][b] = 0.5f * (g[0][b] - g[1][b] + g[2][b]);
    t[3][b] = g[2][b];
  }
  const size_t plane = (size_t)KP * IP;
  const size_t pos = ((((size_t)(k >> 3) * 2 + ((k & 7) >> 2)) * (IP / COB) + (i >> 6)) * 64 + ((k & 3) * 16 + (i & 15))) * 4
                     + ((i & 63) >> 4);
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    const float u0 = t[a][0], u1 = 0.5f * (t[a][0] + t[a][1] + t[a][2]), u2 = 0.5f * (t[a][0] - t[a][1] + t[a][2]),
                u3 = t[a][2];
    U[(size_t)(4 * a + 0) * plane + pos] = u0;
    U[(size_t)(4 * a + 1) * plane + pos] = u1;
    U[(size_t)(4 * a + 2) * plane + pos] = u2;
    U[(size_t)(4 * a + 3) * plane + pos] = u3;
  }
}

// ---- fused convolution -------------------------------------------------------------------------------------
// x [N][K][H][W] -> out [N][I][H][W]; U [16][KP][IP]; bias [I] or null; act: y = v > 0 ? v : slope * v (slope 1 = none)
struct WinoArgs {
  const float* x;
  const float* U;
  const float* bias;
  float* out;
  int K, I, KP, IP, H, W, Ho, Wo, off, tiles_y, tiles_x;   // input H x W, output Ho x Wo, patch origin 2t - off;
  float slope;                                             // tiles_* = number of TBH x TBW tile blocks
  int dbg;
};

// The 4x4 patch of a thread sits at the same (y, x) for every channel: its row offsets, the zero-padding mask and the
// edge shift are computed once.  A row is ONE unaligned 16-byte load (the memory pipe, not the matrix pipe, was the
// limit with 16 dword loads per patch) at a column clamped into [0, W-4]; at the left / right image border the
// loaded window is shifted by `shift` columns against the patch and fixed up in registers (rare, divergent branch).
// Loads are unconditional and masked with integer ANDs afterwards: with `cond ? load : 0` the compiler sinks every
// load under its own branch + s_waitcnt, i.e. dependent HBM round trips.
typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));

struct Patch {
  unsigned off[4];    // row r: (clamp(y0 + r) * W + clamp(x0, 0, W - 4)) * 4 bytes from the (wave-uniform) plane base
  int shift;          // loaded column - patch column
  unsigned mask;      // bit 4r+c set = inside the image
};

__device__ __forceinline__ Patch make_patch(int y0, int x0, int H, int W) {
  Patch p;
  const int xl = min(max(x0, 0), W - 4);
  p.shift = xl - x0;
  p.mask = 0u;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int y = y0 + r;
    p.off[r] = (unsigned)(min(max(y, 0), H - 1) * W + xl) * 4u;
#pragma unroll
    for (int c = 0; c < 4; ++c)
      if (y >= 0 && y < H && x0 + c >= 0 && x0 + c < W) p.mask |= 1u << (4 * r + c);
  }
  return p;
}

__device__ __forceinline__ void load_patch(float (&d)[16], const float* __restrict__ plane, const Patch& p) {
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const f4u v = *reinterpret_cast<const f4u*>(reinterpret_cast<const char*>(plane) + p.off[r]);
    d[4 * r + 0] = v.x; d[4 * r + 1] = v.y; d[4 * r + 2] = v.z; d[4 * r + 3] = v.w;
  }
}

__device__ __forceinline__ void mask_patch(float (&d)[16], const Patch& p, bool channel_ok) {
  if (p.shift != 0) {        // image border: d[c] = loaded[c - shift] (columns that fall outside are masked below)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float l0 = d[4 * r], l1 = d[4 * r + 1], l2 = d[4 * r + 2], l3 = d[4 * r + 3];
      switch (p.shift) {
        case 1: d[4 * r + 1] = l0; d[4 * r + 2] = l1; d[4 * r + 3] = l2; break;
        case 2: d[4 * r + 2] = l0; d[4 * r + 3] = l1; break;
        case -1: d[4 * r + 0] = l1; d[4 * r + 1] = l2; d[4 * r + 2] = l3; break;
        case -2: d[4 * r + 0] = l2; d[4 * r + 1] = l3; break;
        case -3: d[4 * r + 0] = l3; break;
        default: break;
      }
    }
  }
  const unsigned mask = channel_ok ? p.mask : 0u;
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    const unsigned keep = 0u - ((mask >> e) & 1u);
    d[e] = __uint_as_float(__float_as_uint(d[e]) & keep);
  }
}

__device__ __forceinline__ void input_transform(float (&v)[16], const float (&d)[16]) {
  // B^T d B,  B^T = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 1 0 -1]
  float t[16];
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    t[0 * 4 + c] = d[0 * 4 + c] - d[2 * 4 + c];
    t[1 * 4 + c] = d[1 * 4 + c] + d[2 * 4 + c];
    t[2 * 4 + c] = d[2 * 4 + c] - d[1 * 4 + c];
    t[3 * 4 + c] = d[1 * 4 + c] - d[3 * 4 + c];
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    v[r * 4 + 0] = t[r * 4 + 0] - t[r * 4 + 2];
    v[r * 4 + 1] = t[r * 4 + 1] + t[r * 4 + 2];
    v[r * 4 + 2] = t[r * 4 + 2] - t[r * 4 + 1];
    v[r * 4 + 3] = t[r * 4 + 1] - t[r * 4 + 3];
  }
}

// One chunk of a wave: 64 MFMAs on V(ch) in 16 groups of 4, with everything else spread over the groups so that it
// runs in the MFMAs' shadow (a VALU / LDS / VMEM instruction issued right after an MFMA executes during its 32 cycles):
//   groups 0-3   B^T d of the patch of chunk ch+1 (loaded during the previous chunk), one column each
//   group  3     the patch registers are dead: the 4 row loads of chunk ch+2 are issued into them
//   groups 4-7   (B^T d) B, one row each, written straight into the other V buffer
//   group  g     after the last use of an A fragment (g = 3, 7, 11, 15) its successor is loaded in place
// sched_barrier(0) pins the interleave the source spells out.  No second copy of patch / A registers is needed.
__device__ __forceinline__ void chunk_body(f32x4 (&acc)[2][4][4], f32x4 (&afr)[2][2], const float* __restrict__ vcur,
                                           float (&d)[16], const Patch& patch, bool fixup, bool channel_ok, float* __restrict__ vnext,
                                           const float* __restrict__ plane2, const char* __restrict__ unext, unsigned ulane,
                                           size_t uq, size_t ustep) {
  float bfr[2][4], t[16];
#pragma unroll
  for (int tt = 0; tt < 4; ++tt) bfr[0][tt] = vcur[16 * tt];
  if (fixup) mask_patch(d, patch, channel_ok);     // wave-uniform: only waves that touch the image border / padded channels
#pragma unroll
  for (int g = 0; g < 16; ++g) {
    const int q = g >> 3, ks = (g >> 2) & 1, cb = g & 3, cur = (g >> 2) & 1;
    if (cb == 0 && g + 4 < 16) {   // B fragments of the next (q, ks) group
      const int qn = (g + 4) >> 3, ksn = ((g + 4) >> 2) & 1;
#pragma unroll
      for (int tt = 0; tt < 4; ++tt) bfr[cur ^ 1][tt] = vcur[qn * CIB * VS + 4 * ksn * VS + 16 * tt];
    }
#pragma unroll
    for (int tt = 0; tt < 4; ++tt)
      acc[q][cb][tt] = __builtin_amdgcn_mfma_f32_16x16x4f32(afr[q][ks][cb], bfr[cur][tt], acc[q][cb][tt], 0, 0, 0);
#if defined(WINO_ABL) && WINO_ABL == 2
    if (g == 0) {
#pragma unroll
      for (int e = 0; e < 16; ++e) asm volatile("" ::"v"(d[e]));
    }
    if (g == 3) load_patch(d, plane2, patch);
    if (false) {
#elif defined(WINO_ABL) && WINO_ABL == 3
    if (g < 4) {
      const int c = g;
      t[0 * 4 + c] = d[0 * 4 + c] - d[2 * 4 + c];
      t[1 * 4 + c] = d[1 * 4 + c] + d[2 * 4 + c];
      t[2 * 4 + c] = d[2 * 4 + c] - d[1 * 4 + c];
      t[3 * 4 + c] = d[1 * 4 + c] - d[3 * 4 + c];
#else
    if (g < 4) {               // B^T d : column g
      const int c = g;
      t[0 * 4 + c] = d[0 * 4 + c] - d[2 * 4 + c];
      t[1 * 4 + c] = d[1 * 4 + c] + d[2 * 4 + c];
      t[2 * 4 + c] = d[2 * 4 + c] - d[1 * 4 + c];
      t[3 * 4 + c] = d[1 * 4 + c] - d[3 * 4 + c];
      if (g == 3) load_patch(d, plane2, patch);
#endif
#if defined(WINO_ABL) && WINO_ABL == 5
    } else if (g == 4) {
      float sacc = 0.f;
#pragma unroll
      for (int e = 0; e < 16; ++e) sacc += t[e];
      if (sacc == 12345.f) vnext[0] = sacc;
    } else if (false) {
      const int r = 0;
#else
    } else if (g < 8) {        // (B^T d) B : row g - 4, written straight to the next V buffer
      const int r = g - 4;
#endif
      vnext[(r * 4 + 0) * CIB * VS] = t[r * 4 + 0] - t[r * 4 + 2];
      vnext[(r * 4 + 1) * CIB * VS] = t[r * 4 + 1] + t[r * 4 + 2];
      vnext[(r * 4 + 2) * CIB * VS] = t[r * 4 + 2] - t[r * 4 + 1];
      vnext[(r * 4 + 3) * CIB * VS] = t[r * 4 + 1] - t[r * 4 + 3];
    }
#if !defined(WINO_ABL) || WINO_ABL != 3
    if (cb == 3) afr[q][ks] = *reinterpret_cast<const f32x4*>(unext + (q * uq + ks * ustep) * 16 + ulane);
#endif
    __builtin_amdgcn_sched_barrier(0);
  }
}

__global__ __launch_bounds__(WNT) void wino_conv3x3(WinoArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 15, kg = lane >> 4;

  const int tb = blockIdx.x;                       // tile block within the image
  const int tby = tb / a.tiles_x, tbx = tb - tby * a.tiles_x;
  const int i0 = blockIdx.y * COB;                 // first produced channel
  const int n = blockIdx.z;
  const float* xp = a.x + (size_t)n * a.K * a.H * a.W;

  // this thread's tile for the input transform: lane -> tile (4 x 16), wave -> channel within the chunk
  const int ty = tby * TBH + (lane >> 4), tx = tbx * TBW + (lane & 15);
  const Patch patch = make_patch(2 * ty - a.off, 2 * tx - a.off, a.H, a.W);
  const size_t cplane = (size_t)a.H * a.W;

  f32x4 acc[2][4][4];
#pragma unroll
  for (int q = 0; q < 2; ++q)
#pragma unroll
    for (int cb = 0; cb < 4; ++cb)
#pragma unroll
      for (int t = 0; t < 4; ++t) acc[q][cb][t] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const size_t uplane = (size_t)a.KP * a.IP;
  // A fragments of this wave (xi = 2w + q): one f32x4 per (q, chunk, k-step) = its 4 channel blocks
  const int nblk = a.IP / COB;
  const char* ubase = reinterpret_cast<const char*>(a.U + (size_t)(2 * w) * uplane) + (size_t)blockIdx.y * 64 * 16;   // uniform
  const unsigned ulane = (unsigned)lane * 16u;
  const size_t ustep = (size_t)nblk * 64;       // f32x4 elements per k-step
  const size_t uq = uplane / 4;                 // ... per xi

  float d[16];
  f32x4 afr[2][2];
  const int nchunk = a.KP / CIB;
  // prologue: chunk 0 -> V buffer 0; chunk 1's patch in flight
  load_patch(d, xp + (size_t)min(w, a.K - 1) * cplane, patch);
#pragma unroll
  for (int q = 0; q < 2; ++q)
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) afr[q][ks] = *reinterpret_cast<const f32x4*>(ubase + (q * uq + ks * ustep) * 16 + ulane);
  {
    float v[16];
    mask_patch(d, patch, w < a.K);
    input_transform(v, d);
    float* vb = lds + w * VS + lane;
#pragma unroll
    for (int xi = 0; xi < 16; ++xi) vb[xi * CIB * VS] = v[xi];
  }
  load_patch(d, xp + (size_t)min(min(1, nchunk - 1) * CIB + w, a.K - 1) * cplane, patch);
  // Everything the prologue loaded must have landed before the loop: otherwise the compiler's wait-count
  // bookkeeping carries "A fragments may still be in flight" into the loop header and, vmcnt being in-order,
  // makes every iteration wait for its own freshly issued prefetches before the first MFMA.
  __builtin_amdgcn_s_waitcnt(0);
  __syncthreads();

  // waves whose 64 patches are all interior (and whose channels are all real) skip the padding fix-ups
  const bool plain = __builtin_amdgcn_ballot_w64(patch.mask != 0xffffu || patch.shift != 0) == 0 && a.K % CIB == 0;
  const int vwoff = w * VS + lane;
  const int vroff = (2 * w) * CIB * VS + kg * VS + j;

  for (int ch = 0; ch < nchunk; ++ch) {
    const float* vcur = lds + (ch & 1) * VBUF + vroff;
    float* vnext = lds + ((ch + 1) & 1) * VBUF + vwoff;
    // the last iterations re-load the last chunk so that the loop body stays branch-free
    const int c1 = min(ch + 1, nchunk - 1), c2 = min(ch + 2, nchunk - 1);
    const float* plane2 = xp + (size_t)min(c2 * CIB + w, a.K - 1) * cplane;
    const char* unext = ubase + (size_t)c1 * 2 * ustep * 16;
    chunk_body(acc, afr, vcur, d, patch, !plain, c1 * CIB + w < a.K, vnext, plane2, unext, ulane, uq, ustep);
#if !defined(WINO_ABL) || WINO_ABL != 1
    __syncthreads();
#endif
  }

  if (a.dbg & 1) {
    float sum = 0.f;
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
      for (int cb = 0; cb < 4; ++cb)
#pragma unroll
        for (int t = 0; t < 4; ++t) sum += acc[q][cb][t][0] + acc[q][cb][t][1] + acc[q][cb][t][2] + acc[q][cb][t][3];
    if (sum == 123.456f) a.out[0] = sum;
    return;
  }
  // ---- output stage: 16 produced channels at a time through LDS -------------------------------------------
  // accumulator tile layout: row (channel) = 4 * kg + reg, column (tile) = j
  const bool vec_ok = (a.Wo % 2 == 0);
#pragma unroll
  for (int cb = 0; cb < 4; ++cb) {
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) lds[((2 * w + q) * 16 + 4 * kg + r) * XS + 16 * t + j] = acc[q][cb][t][r];
    __syncthreads();
#pragma unroll
    for (int rep = 0; rep < 2; ++rep) {
      const int p = tid + rep * WNT;
      const int il = p >> 6, t = p & 63;            // channel within the block of 16, tile
      const int i = i0 + 16 * cb + il;
      float m[16];
#pragma unroll
      for (int xi = 0; xi < 16; ++xi) m[xi] = lds[(xi * 16 + il) * XS + t];
      // A^T M A,  A^T = [1 1 1 0; 0 1 -1 -1]
      float s[2][4];
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        s[0][c] = m[0 * 4 + c] + m[1 * 4 + c] + m[2 * 4 + c];
        s[1][c] = m[1 * 4 + c] - m[2 * 4 + c] - m[3 * 4 + c];
      }
      float y[2][2];
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        y[r][0] = s[r][0] + s[r][1] + s[r][2];
        y[r][1] = s[r][1] - s[r][2] - s[r][3];
      }
      if (i < a.I) {
        const float b = a.bias ? a.bias[i] : 0.f;
        const int oty = tby * TBH + (t >> 4), otx = tbx * TBW + (t & 15);
        const int oy = 2 * oty, ox = 2 * otx;
        float* op = a.out + (((size_t)n * a.I + i) * a.Ho + oy) * a.Wo + ox;
#pragma unroll
        for (int r = 0; r < 2; ++r) {
          if (oy + r >= a.Ho) break;
          float v0 = y[r][0] + b, v1 = y[r][1] + b;
          v0 = v0 > 0.f ? v0 : a.slope * v0;
          v1 = v1 > 0.f ? v1 : a.slope * v1;
          float* o = op + (size_t)r * a.Wo;
          if (vec_ok && ox + 1 < a.Wo) *reinterpret_cast<float2*>(o) = make_float2(v0, v1);
          else {
            if (ox < a.Wo) o[0] = v0;
            if (ox + 1 < a.Wo) o[1] = v1;
          }
        }
      }
    }
    __syncthreads();
  }
}

int round_up(int v, int m) { return (v + m - 1) / m * m; }

}  // namespace

extern "C" int64_t savfi_conv3x3_workspace_floats(int K, int I) {
  if (K <= 0 || I <= 0) return SAVFI_E_SHAPE;
  return (int64_t)16 * round_up(K, CIB) * round_up(I, COB);
}

// mode 0: out[n][co] = act(conv2d(x[n], w, zero padding `pad`)[co] + bias[co])   x [N][Ci][H][W] -> [N][Co][H+2pad-2][W+2pad-2]
// mode 1: its data gradient: x = gy [N][Co][H][W] -> out = gx [N][Ci][H+2-2pad][W+2-2pad]
extern "C" int savfi_conv3x3_f32(const float* x, const float* w, const float* bias, float* out, float* workspace,
                                 int N, int Ci, int Co, int H, int W, int pad, int mode, float slope, void* stream) {
  if (!x || !w || !out || !workspace) return SAVFI_E_NULL;
  if (N <= 0 || Ci <= 0 || Co <= 0 || H <= 0 || W <= 0) return SAVFI_E_SHAPE;
  if ((mode != 0 && mode != 1) || (pad != 0 && pad != 1) || W < 4) return SAVFI_E_UNSUPPORTED;   // rows are 16-byte loads
  const int K = mode == 0 ? Ci : Co, I = mode == 0 ? Co : Ci;       // reduction / produced channels
  const int KP = round_up(K, CIB), IP = round_up(I, COB);
  // forward: patch origin 2t - pad; gradient of a pad-p convolution = pad-(2-p) correlation with the flipped filter
  const int off = mode == 0 ? pad : 2 - pad;
  const int Ho = H + 2 * off - 2, Wo = W + 2 * off - 2;
  if (Ho <= 0 || Wo <= 0) return SAVFI_E_SHAPE;
  const int th = savfi_cdiv(savfi_cdiv(Ho, 2), TBH), tw = savfi_cdiv(savfi_cdiv(Wo, 2), TBW);
  if ((int64_t)th * tw > 0x7fffffffLL || IP / COB > 65535 || N > 65535) return SAVFI_E_TOOBIG;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(wino_filter_transform, dim3(savfi_cdiv(KP * IP, 256)), dim3(256), 0, st, w, workspace, Co, Ci, K, I,
                     KP, IP, mode);
  if (int e = savfi_launch_status()) return e;
  constexpr size_t lds = (size_t)LDS_FLOATS * sizeof(float);
  static const hipError_t attr = hipFuncSetAttribute((const void*)wino_conv3x3, hipFuncAttributeMaxDynamicSharedMemorySize,
                                                     (int)lds);
  if (attr != hipSuccess) return (int)attr;
  WinoArgs a{x, workspace, mode == 0 ? bias : nullptr, out, K, I, KP, IP, H, W, Ho, Wo, off, th, tw, slope,
             getenv("SAVFI_WINO_DBG") ? atoi(getenv("SAVFI_WINO_DBG")) : 0};
  hipLaunchKernelGGL(wino_conv3x3, dim3(th * tw, IP / COB, N), dim3(WNT), lds, st, a);
  return savfi_launch_status();
}
