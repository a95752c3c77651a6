// Bilinear x2 up-sampling (forward and exact adjoint) for gfx950.
//
// Replaces torch.nn.Upsample(scale_factor=2, mode='bilinear', align_corners=True) in the SepConv decoder
// and sub-networks (sepconv/model.py:191, :213, :220, :227, :234) and F.interpolate(..., align_corners=False)
// in VoxelFlow's decoder (voxel_flow.py:400, :407, :414).  ATen's generic kernels take ~8 % of the SepConv
// inner step at 384x512 (a 51-channel 192x256 -> 384x512 map runs at ~0.25 TB/s); this op is a pure
// HBM stream: read in once (neighbours hit L1/L2), write out once.
//
// Source coordinates follow ATen exactly (UpSample.h area_pixel_compute_source_index):
//   align_corners: src = dst * (in-1)/(out-1)            else: src = max((dst+0.5)*0.5 - 0.5, 0)
//   i0 = (int)src, i1 = i0 + (i0 < in-1), l1 = src - i0, l0 = 1 - l1
// Backward is a gather (no atomics, deterministic): every input pixel visits the <= 6 output rows/columns
// whose i0 / i1 can equal it and re-evaluates the forward's weights.
#include <cstdlib>

#include "common.h"

namespace {

struct Src { int i0, i1; float l0, l1; };

__device__ __forceinline__ Src source(int dst, int in, float scale, int align) {
  float s = align ? scale * (float)dst : fmaxf(((float)dst + 0.5f) * scale - 0.5f, 0.f);
  Src r;
  r.i0 = min((int)s, in - 1);
  r.i1 = r.i0 + ((r.i0 < in - 1) ? 1 : 0);
  r.l1 = s - (float)r.i0;
  r.l0 = 1.f - r.l1;
  return r;
}

__device__ __forceinline__ float scale_of(int in, int out, int align) {
  // ATen: align_corners ? (in-1)/(out-1) : 1/scale_factor (= 0.5 for x2; recompute_scale_factor unset)
  return align ? (out > 1 ? (float)(in - 1) / (float)(out - 1) : 0.f) : 0.5f;
}

// Window geometry.  The op is defined on a virtual [H, W] -> [2H, 2W] map; the source buffer holds the crop
// rows [sy0, sy0+Hs) x cols [sx0, sx0+Ws) of it and the output buffer the window rows [oy0, oy0+Hw) x cols
// [ox0, ox0+Ww) of the result.  The full op is the window (0, 0, H, W) -> (0, 0, 2H, 2W).  Used by SepConv's
// sub-networks, whose 51-tap maps are only consumed on the un-padded frame area (sepconv/model.py).
struct Win { int H, W, sy0, sx0, Hs, Ws, oy0, ox0, Hw, Ww, align; };

// Forward, tiled (round 3; re-tiled in round 4): a workgroup owns 32 output rows x 128 output columns, a thread 4 columns of rows
// ly, ly + 8, ly + 16, ly + 24.  The <= 19 x 68 source pixels the tile reads are staged in LDS with coalesced loads and every output
// takes its four taps from there (the first version issued 16 global loads per thread; L1 hits, but the address path was the limit).
// Round 4: the 8-row tiles of round 3 spent most of their instructions on per-thread bookkeeping -- a run-time integer division
// per staged element, four source() evaluations in x and one in y per FOUR outputs -- at 2.3 TB/s on the 51-channel sub-network maps;
// here the staging index splits by a constant, the x sources are evaluated once per thread and serve four rows.
// Same source indices, weights and order of operations per output as every earlier version: bit-identical results.
// Round 6: a thread owns 4 output columns x UFR consecutive output rows and walks DOWN its source rows.  The horizontal blend of a source row
// (xl0 * s[xi0] + xl1 * s[xi1], ATen's inner parentheses) is evaluated ONCE per (source row, output column) and serves the two to three
// output rows that read that row, where the earlier form re-read four taps and re-blended per output: ~5 instead of ~11 instructions per
// output element (the kernel was issue-bound at 3.3 TB/s: 4 LDS reads + 7 VALU per element).  Same products, same order: bit-identical.
constexpr int UFR = 8;                                                         // output rows per thread
constexpr int UFH = 8 * UFR, UFW = 128, UFSR = UFH / 2 + 3, UFSC = UFW / 2 + 4;      // tile (64 x 128), staged source rows / columns
__global__ __launch_bounds__(256) void upsample2x_fwd(const float* __restrict__ in, float* __restrict__ out, Win g) {
  __shared__ float src[UFSR][UFSC];
  const int Ho = 2 * g.H, Wo = 2 * g.W;
  const int tiles_x = (g.Ww + UFW - 1) / UFW;
  const int ty = blockIdx.x / tiles_x, tx = blockIdx.x - ty * tiles_x;
  const int wy0 = ty * UFH, wx0 = tx * UFW;
  const float sh = scale_of(g.H, Ho, g.align), sw = scale_of(g.W, Wo, g.align);
  const float* p = in + (size_t)blockIdx.y * g.Hs * g.Ws;
  // source rows / columns the tile touches (virtual coordinates), from its first and last output
  const int last_y = min(wy0 + UFH, g.Hw) - 1, last_x = min(wx0 + UFW, g.Ww) - 1;
  const int ry0 = source(g.oy0 + wy0, g.H, sh, g.align).i0, ry1 = source(g.oy0 + last_y, g.H, sh, g.align).i1;
  const int rx0 = source(g.ox0 + wx0, g.W, sw, g.align).i0, rx1 = source(g.ox0 + last_x, g.W, sw, g.align).i1;
  const int nr = ry1 - ry0 + 1, nc = rx1 - rx0 + 1;             // <= UFSR, <= UFSC (scale <= 0.5)
  const float* pw = p + (size_t)(ry0 - g.sy0) * g.Ws + (rx0 - g.sx0);
#pragma unroll
  for (int k = 0; k < (UFSR * UFSC + 255) / 256; ++k) {
    const int i = threadIdx.x + 256 * k, r = i / UFSC, c = i - r * UFSC;       // constant divisor
    if (r < nr && c < nc) src[r][c] = pw[(size_t)r * g.Ws + c];
  }
  __syncthreads();
  const int lx = threadIdx.x & 31, ly = threadIdx.x >> 5;        // 32 groups of 4 output columns, 8 groups of UFR rows
  const int wx = wx0 + 4 * lx, wyb = wy0 + UFR * ly;
  if (wx >= g.Ww || wyb >= g.Hw) return;
  int xi0[4], xi1[4];
  float xl0[4], xl1[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const Src sx = source(min(g.ox0 + wx + k, g.ox0 + g.Ww - 1), g.W, sw, g.align);
    xi0[k] = sx.i0 - rx0; xi1[k] = sx.i1 - rx0; xl0[k] = sx.l0; xl1[k] = sx.l1;
  }
  auto blend_row = [&](int r, float (&hrow)[4]) {       // ATen's horizontal blend of staged source row r for this thread's 4 columns
#pragma unroll
    for (int k = 0; k < 4; ++k) hrow[k] = xl0[k] * src[r][xi0[k]] + xl1[k] * src[r][xi1[k]];
  };
  // two rolling rows: `lo` = source row cur, `hi` = source row cur + 1 (consecutive output rows read i0 in {cur, cur + 1}: scale <= 0.5)
  int cur = source(g.oy0 + wyb, g.H, sh, g.align).i0 - ry0;
  float lo[4], hi[4];
  blend_row(cur, lo);
  blend_row(min(cur + 1, nr - 1), hi);
  float* o = out + ((size_t)blockIdx.y * g.Hw + wyb) * g.Ww + wx;
  const bool full = wx + 3 < g.Ww;
#pragma unroll
  for (int rr = 0; rr < UFR; ++rr) {
    const int wy = wyb + rr;
    if (wy >= g.Hw) break;
    const Src sy = source(g.oy0 + wy, g.H, sh, g.align);
    const int i0 = sy.i0 - ry0;
    if (i0 != cur) {                 // one source row further (never two: the source index advances by <= 0.5 per output row)
      cur = i0;
#pragma unroll
      for (int k = 0; k < 4; ++k) lo[k] = hi[k];
      blend_row(min(cur + 1, nr - 1), hi);
    }
    // i1 == i0 only on the map's last source row, where l1 == 0 for align_corners and the row is clamped: `hi` then holds the same row
    float v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = sy.l0 * lo[k] + sy.l1 * (sy.i1 - ry0 == cur ? lo[k] : hi[k]);
    if (full && ((((uintptr_t)o) & 15u) == 0)) *reinterpret_cast<float4*>(o) = make_float4(v[0], v[1], v[2], v[3]);
    else if (full && ((((uintptr_t)o) & 7u) == 0)) {      // (a window row of 450 floats: every other row is only 8-byte aligned)
      *reinterpret_cast<float2*>(o) = make_float2(v[0], v[1]);
      *reinterpret_cast<float2*>(o + 2) = make_float2(v[2], v[3]);
    } else
      for (int k = 0; k < 4 && wx + k < g.Ww; ++k) o[k] = v[k];
    o += g.Ww;
  }
}

// (leaky) ReLU derivative of the producer of the up-sampled map, folded into the adjoint's store: gin *= (y > 0 ? 1 : slope), y = that
// producer's activated output = this op's input (same shape as gin); y == nullptr: the plain adjoint.  (round 5: the element-wise pass
// savfi_bias_act_bwd_f32 that the producing convolution ran over its cotangent, reference: ReLU's backward under autograd)
struct UpMask { const float* y; float slope; };
__device__ __forceinline__ float up_masked(float v, const UpMask& m, size_t idx) { return m.y ? (m.y[idx] > 0.f ? v : m.slope * v) : v; }

__global__ __launch_bounds__(256) void upsample2x_bwd(const float* __restrict__ gout, float* __restrict__ gin, Win g, UpMask um) {
  const int Ho = 2 * g.H, Wo = 2 * g.W;
  const int item = blockIdx.x * 256 + threadIdx.x;           // pixels of one source plane, row-major
  if (item >= g.Hs * g.Ws) return;
  const int cy = item / g.Ws, cx = item - cy * g.Ws;
  const int ix = g.sx0 + cx, iy = g.sy0 + cy;               // coordinates in the virtual source
  const float* gp = gout + (size_t)blockIdx.y * g.Hw * g.Ww;
  const float sh = scale_of(g.H, Ho, g.align), sw = scale_of(g.W, Wo, g.align);
  // candidate outputs: src(o) in (i-1, i+1).  For both index rules that is o in {2i-1 .. 2i+2}; one more on each
  // side is visited so that a rounding of src at an integer cannot drop a contribution (weights are re-evaluated
  // with the forward's arithmetic, so extra candidates simply weigh zero); clipped to the output window
  constexpr int NC = 6;
  const int oy_lo = max(g.oy0, 2 * iy - 2), oy_hi = min(g.oy0 + g.Hw - 1, 2 * iy + 3);
  const int ox_lo = max(g.ox0, 2 * ix - 2), ox_hi = min(g.ox0 + g.Ww - 1, 2 * ix + 3);
  float wxs[NC];
#pragma unroll
  for (int k = 0; k < NC; ++k) {
    const int ox = ox_lo + k;
    float w = 0.f;
    if (ox <= ox_hi) {
      const Src s = source(ox, g.W, sw, g.align);
      w = (s.i0 == ix ? s.l0 : 0.f) + (s.i1 == ix ? s.l1 : 0.f);
    }
    wxs[k] = w;
  }
  float acc = 0.f;
  for (int oy = oy_lo; oy <= oy_hi; ++oy) {
    const Src s = source(oy, g.H, sh, g.align);
    const float wy = (s.i0 == iy ? s.l0 : 0.f) + (s.i1 == iy ? s.l1 : 0.f);
    if (wy == 0.f) continue;
    const float* row = gp + (size_t)(oy - g.oy0) * g.Ww - g.ox0;
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < NC; ++k)
      if (ox_lo + k <= ox_hi) t = fmaf(wxs[k], row[ox_lo + k], t);
    acc = fmaf(wy, t, acc);
  }
  const size_t oi = ((size_t)blockIdx.y * g.Hs + cy) * g.Ws + cx;
  gin[oi] = up_masked(acc, um, oi);
}

// The same adjoint, separable and tiled (round 2): a workgroup owns 8 source rows x 64 source columns.  Pass 1 folds the x
// direction -- tmp[oy][ix] = sum over the <= 6 candidate columns of wx * gout[oy][ox] -- for the <= 20 output rows the tile's
// rows can touch, into LDS; pass 2 folds y from LDS.  Same candidates, same weights, same order of the two nested sums as
// upsample2x_bwd (bit-identical results), but 15 global loads per source pixel instead of 36 and the x weights once per thread.
#ifndef SAVFI_UBH
#define SAVFI_UBH 8
#endif
constexpr int UBH = SAVFI_UBH, UBW = 64, UBR = 2 * UBH + 4;     // tile rows / cols, output rows a tile can touch (16- and 32-row tiles measured slower: LDS per workgroup, profiles/r04_upsample_bench.txt)
constexpr int UBC = 2 * UBW + 4;                        // output columns a tile can touch (2 ix - 2 .. 2 ix + 3)
__global__ __launch_bounds__(256) void upsample2x_bwd_tiled(const float* __restrict__ gout, float* __restrict__ gin, Win g, UpMask um) {
  // round 3: the output-gradient tile itself is staged first (coalesced, each element once: 10.6 loads per thread where the x
  // pass read 30 scattered ones per thread from global / L1); both passes then run out of LDS with the same candidates, weights
  // and order of sums as before (bit-identical).
  __shared__ float gt[UBR][UBC + 1];
  __shared__ float tmp[UBR][UBW];
  const int Ho = 2 * g.H, Wo = 2 * g.W;
  const int tiles_x = (g.Ws + UBW - 1) / UBW;
  const int ty = blockIdx.x / tiles_x, tx = blockIdx.x - ty * tiles_x;
  const int lx = threadIdx.x & (UBW - 1), rg = threadIdx.x >> 6;          // column in the tile, row group (0..3)
  const int cx = tx * UBW + lx, cy0 = ty * UBH;
  const int ix = g.sx0 + min(cx, g.Ws - 1), iy0 = g.sy0 + cy0;            // virtual source coordinates
  const int ix_first = g.sx0 + tx * UBW, ix_last = g.sx0 + min(tx * UBW + UBW, g.Ws) - 1;
  const float* gp = gout + (size_t)blockIdx.y * g.Hw * g.Ww;
  const float sh = scale_of(g.H, Ho, g.align), sw = scale_of(g.W, Wo, g.align);
  constexpr int NC = 6;
  // output rows / columns this tile's source pixels can touch, clipped to the window
  const int row_lo = max(g.oy0, 2 * iy0 - 2), row_hi = min(g.oy0 + g.Hw - 1, 2 * (iy0 + UBH - 1) + 3);
  const int col_lo = max(g.ox0, 2 * ix_first - 2), col_hi = min(g.ox0 + g.Ww - 1, 2 * ix_last + 3);
  const int ncol = col_hi - col_lo + 1, nrow = row_hi - row_lo + 1;
  {
    const float* gw = gp + (size_t)(row_lo - g.oy0) * g.Ww + (col_lo - g.ox0);
#pragma unroll
    for (int k = 0; k < (UBR * UBC + 255) / 256; ++k) {
      const int i = threadIdx.x + 256 * k, r = i / UBC, c = i - r * UBC;         // constant divisor (round 4: was a run-time division)
      if (r < nrow && c < ncol) gt[r][c] = gw[(size_t)r * g.Ww + c];
    }
  }
  // round 4: the weights of the tile's 64 columns and 8 rows are evaluated ONCE per workgroup into LDS (6 candidates each: 384 + 48
  // source() evaluations spread over the 256 threads) -- every thread used to evaluate its column's six x weights (four threads per
  // column) and six y weights per output (64 threads per row): ~300 of its ~500 instructions.  Same candidates, same source()
  // arithmetic, same order of the sums: bit-identical.
  __shared__ float wxl[UBW][NC], wyl[UBH][NC];
  for (int i = threadIdx.x; i < UBW * NC; i += 256) {
    const int c = i / NC, k = i - c * NC;
    const int ixc = g.sx0 + min(tx * UBW + c, g.Ws - 1);
    const int ox = max(g.ox0, 2 * ixc - 2) + k;
    float w = 0.f;
    if (ox <= min(g.ox0 + g.Ww - 1, 2 * ixc + 3)) {
      const Src s = source(ox, g.W, sw, g.align);
      w = (s.i0 == ixc ? s.l0 : 0.f) + (s.i1 == ixc ? s.l1 : 0.f);
    }
    wxl[c][k] = w;
  }
  if (threadIdx.x < UBH * NC) {
    const int r = threadIdx.x / NC, k = threadIdx.x - r * NC;
    const int iy = iy0 + r;
    const int oy = max(g.oy0, 2 * iy - 2) + k;
    float w = 0.f;
    if (oy <= min(g.oy0 + g.Hw - 1, 2 * iy + 3)) {
      const Src s = source(oy, g.H, sh, g.align);
      w = (s.i0 == iy ? s.l0 : 0.f) + (s.i1 == iy ? s.l1 : 0.f);
    }
    wyl[r][k] = w;
  }
  const int ox_lo = max(g.ox0, 2 * ix - 2), ox_hi = min(g.ox0 + g.Ww - 1, 2 * ix + 3);
  __syncthreads();
  float wxs[NC];
#pragma unroll
  for (int k = 0; k < NC; ++k) wxs[k] = wxl[lx][k];
  for (int r = row_lo + rg; r <= row_hi; r += 4) {
    const float* row = &gt[r - row_lo][0] - col_lo;
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < NC; ++k)
      if (ox_lo + k <= ox_hi) t = fmaf(wxs[k], row[ox_lo + k], t);
    tmp[r - row_lo][lx] = t;
  }
  __syncthreads();
#pragma unroll
  for (int rr = 0; rr < UBH / 4; ++rr) {
    const int cy = cy0 + rg + 4 * rr, iy = g.sy0 + cy;
    if (cy >= g.Hs || cx >= g.Ws) continue;
    const int oy_lo = max(g.oy0, 2 * iy - 2), oy_hi = min(g.oy0 + g.Hw - 1, 2 * iy + 3);
    float acc = 0.f;
#pragma unroll
    for (int k = 0; k < NC; ++k) {
      const float wy = wyl[rg + 4 * rr][k];
      if (oy_lo + k > oy_hi || wy == 0.f) continue;
      acc = fmaf(wy, tmp[oy_lo + k - row_lo][lx], acc);
    }
    const size_t oi = ((size_t)blockIdx.y * g.Hs + cy) * g.Ws + cx;
    gin[oi] = up_masked(acc, um, oi);
  }
}

// The same adjoint as a stream (round 5): a WAVE owns a strip of 64 source columns x USR source rows and walks down the <= 2 USR + 5 output
// rows it can touch, four at a time: the rows' <= 132 columns come in as three coalesced dword loads per lane, pass through a wave-private
// LDS row (no workgroup barrier: a wave's DS operations execute in order), every lane folds its six x candidates, and the row's sum goes
// to the one or two source rows it belongs to (src(oy)'s i0 / i1: exactly the candidates of the tiled form that weigh non-zero) -- three
// rolling accumulators, a finished source row leaves as one coalesced 256-byte store.  Same candidates, weights and order of the sums as
// upsample2x_bwd / _tiled (bit-identical); the tiled form moved 10.5 KB in and 2 KB out per workgroup behind two barriers and reached
// 2.1 - 2.4 TB/s on the large maps (profiles/r05_upsample_bench.txt).
constexpr int USW = 64, USB = 4, USP = 2 * USW + 8;        // strip columns, output rows per batch, LDS row pitch (floats)
// USR: source rows per strip (the host picks 32, 16 or 8: enough waves to hide the walk's latency -- a wave's batches are serial)
__global__ __launch_bounds__(256) void upsample2x_bwd_strip(const float* __restrict__ gout, float* __restrict__ gin, Win g, int USR, UpMask um) {
  __shared__ float rows[4][USB][USP];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int Ho = 2 * g.H, Wo = 2 * g.W;
  const int strips_x = (g.Ws + USW - 1) / USW, strips_y = (g.Hs + USR - 1) / USR;
  const int strip = blockIdx.x * 4 + w;
  if (strip >= strips_x * strips_y) return;                 // (whole waves leave: no barrier below)
  const int sy = strip / strips_x, sx = strip - sy * strips_x;
  const int cx = sx * USW + lane, cy_a = sy * USR, cy_b = min(cy_a + USR, g.Hs);
  const bool col_ok = cx < g.Ws;
  const int ix = g.sx0 + min(cx, g.Ws - 1);
  const int ix_first = g.sx0 + sx * USW, ix_last = g.sx0 + min(sx * USW + USW, g.Ws) - 1;
  const int iy_a = g.sy0 + cy_a, iy_b = g.sy0 + cy_b;       // virtual source rows [iy_a, iy_b)
  const float* gp = gout + (size_t)blockIdx.y * g.Hw * g.Ww;
  float* gq = gin + (size_t)blockIdx.y * g.Hs * g.Ws;
  const float sh = scale_of(g.H, Ho, g.align), sw = scale_of(g.W, Wo, g.align);
  constexpr int NC = 6;
  const int oy_end = g.oy0 + g.Hw - 1, ox_end = g.ox0 + g.Ww - 1;
  const int row_lo = max(g.oy0, 2 * iy_a - 2), row_hi = min(oy_end, 2 * (iy_b - 1) + 3);
  const int col_lo = max(g.ox0, 2 * ix_first - 2), col_hi = min(ox_end, 2 * ix_last + 3);
  const int ox_lo = max(g.ox0, 2 * ix - 2), ox_hi = min(ox_end, 2 * ix + 3);
  float wxs[NC];
#pragma unroll
  for (int k = 0; k < NC; ++k) {
    const int ox = ox_lo + k;
    float wgt = 0.f;
    if (ox <= ox_hi) {
      const Src s = source(ox, g.W, sw, g.align);
      wgt = (s.i0 == ix ? s.l0 : 0.f) + (s.i1 == ix ? s.l1 : 0.f);
    }
    wxs[k] = wgt;
  }
  const int rel = ox_lo - col_lo;                           // the lane's first candidate inside the staged row
  const int ncol = col_hi - col_lo + 1;
  float a0 = 0.f, a1 = 0.f, a2 = 0.f;                       // source rows cur, cur + 1, cur + 2
  int cur = iy_a;
  auto emit = [&]() {                                        // source row `cur` is complete
    if (col_ok) {
      const size_t oi = (size_t)(cur - g.sy0) * g.Ws + cx;
      gq[oi] = up_masked(a0, um, (size_t)blockIdx.y * g.Hs * g.Ws + oi);
    }
    a0 = a1; a1 = a2; a2 = 0.f; ++cur;
  };
  float gnext[USB][3];
  auto load_batch = [&](int r0) {
#pragma unroll
    for (int b = 0; b < USB; ++b) {
      const int oy = min(r0 + b, row_hi);
      const float* src = gp + (size_t)(oy - g.oy0) * g.Ww + (col_lo - g.ox0);
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const int col = lane + 64 * c;
        gnext[b][c] = col < ncol ? src[col] : 0.f;
      }
    }
  };
  if (row_lo <= row_hi) load_batch(row_lo);
  for (int r0 = row_lo; r0 <= row_hi; r0 += USB) {
    float gl[USB][3];
#pragma unroll
    for (int b = 0; b < USB; ++b)
#pragma unroll
      for (int c = 0; c < 3; ++c) gl[b][c] = gnext[b][c];
    if (r0 + USB <= row_hi) load_batch(r0 + USB);           // the next batch's rows are in flight under this batch's folds
#pragma unroll
    for (int b = 0; b < USB; ++b) {
      rows[w][b][lane] = gl[b][0];
      rows[w][b][lane + 64] = gl[b][1];
      if (lane < USP - 128) rows[w][b][lane + 128] = gl[b][2];
    }
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int b = 0; b < USB; ++b) {
      const int oy = r0 + b;
      if (oy > row_hi) break;
      while (cur < iy_b && oy > min(oy_end, 2 * cur + 3)) emit();
      const float* row = &rows[w][b][rel];
      float t = 0.f;
#pragma unroll
      for (int k = 0; k < NC; ++k)
        if (ox_lo + k <= ox_hi) t = fmaf(wxs[k], row[k], t);
      const Src s = source(oy, g.H, sh, g.align);
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const int iy = e == 0 ? s.i0 : s.i1;
        if (e == 1 && s.i1 == s.i0) break;
        const float wy = (s.i0 == iy ? s.l0 : 0.f) + (s.i1 == iy ? s.l1 : 0.f);
        // a candidate of source row iy in the tiled form: oy inside [2 iy - 2, 2 iy + 3] (it always is, by construction of that window)
        if (wy == 0.f || iy < cur || iy >= iy_b || oy < 2 * iy - 2 || oy > 2 * iy + 3) continue;
        const int d = iy - cur;
        if (d == 0) a0 = fmaf(wy, t, a0);
        else if (d == 1) a1 = fmaf(wy, t, a1);
        else a2 = fmaf(wy, t, a2);
      }
    }
    __builtin_amdgcn_wave_barrier();
  }
  while (cur < iy_b) emit();
}

// host mirror of source(): first / last source index an output range touches
void touched(int o_first, int o_last, int in, int out, int align, int* lo, int* hi) {
  const float scale = align ? (out > 1 ? (float)(in - 1) / (float)(out - 1) : 0.f) : 0.5f;
  auto idx0 = [&](int dst) {
    float s = align ? scale * (float)dst : fmaxf(((float)dst + 0.5f) * scale - 0.5f, 0.f);
    int i0 = (int)s;
    return i0 < in - 1 ? i0 : in - 1;
  };
  *lo = idx0(o_first);
  const int i0 = idx0(o_last);
  *hi = i0 + (i0 < in - 1 ? 1 : 0);
}

int check_window(const Win& g, int planes) {
  if (planes <= 0 || g.H <= 0 || g.W <= 0 || g.Hs <= 0 || g.Ws <= 0 || g.Hw <= 0 || g.Ww <= 0) return SAVFI_E_SHAPE;
  if (g.sy0 < 0 || g.sx0 < 0 || g.sy0 + g.Hs > g.H || g.sx0 + g.Ws > g.W) return SAVFI_E_SHAPE;
  if (g.oy0 < 0 || g.ox0 < 0 || g.oy0 + g.Hw > 2 * g.H || g.ox0 + g.Ww > 2 * g.W) return SAVFI_E_SHAPE;
  int lo, hi;   // every source pixel the window reads must be inside the crop
  touched(g.oy0, g.oy0 + g.Hw - 1, g.H, 2 * g.H, g.align, &lo, &hi);
  if (lo < g.sy0 || hi >= g.sy0 + g.Hs) return SAVFI_E_SHAPE;
  touched(g.ox0, g.ox0 + g.Ww - 1, g.W, 2 * g.W, g.align, &lo, &hi);
  if (lo < g.sx0 || hi >= g.sx0 + g.Ws) return SAVFI_E_SHAPE;
  if (planes > 65535 || (int64_t)g.Hw * g.Ww >= ((int64_t)1 << 31) || (int64_t)g.Hs * g.Ws >= ((int64_t)1 << 31)) return SAVFI_E_TOOBIG;
  return 0;
}

}  // namespace

extern "C" int savfi_upsample2x_window_fwd_f32(const float* in, float* out, int planes, int H, int W, int sy0, int sx0,
                                               int Hs, int Ws, int oy0, int ox0, int Hw, int Ww, int align_corners,
                                               void* stream) {
  if (!in || !out) return SAVFI_E_NULL;
  const Win g{H, W, sy0, sx0, Hs, Ws, oy0, ox0, Hw, Ww, align_corners ? 1 : 0};
  if (int rc = check_window(g, planes)) return rc;
  dim3 grid(savfi_cdiv(Hw, UFH) * savfi_cdiv(Ww, UFW), planes, 1);
  hipLaunchKernelGGL(upsample2x_fwd, grid, dim3(256), 0, (hipStream_t)stream, in, out, g);
  return savfi_launch_status();
}

extern "C" int savfi_upsample2x_window_bwd_f32(const float* gout, float* gin, int planes, int H, int W, int sy0, int sx0,
                                               int Hs, int Ws, int oy0, int ox0, int Hw, int Ww, int align_corners,
                                               void* stream) {
  return savfi_upsample2x_window_bwd_masked_f32(gout, nullptr, 1.f, gin, planes, H, W, sy0, sx0, Hs, Ws, oy0, ox0, Hw, Ww, align_corners, stream);
}

extern "C" int savfi_upsample2x_window_bwd_masked_f32(const float* gout, const float* y, float slope, float* gin, int planes, int H, int W,
                                                      int sy0, int sx0, int Hs, int Ws, int oy0, int ox0, int Hw, int Ww,
                                                      int align_corners, void* stream) {
  if (!gout || !gin) return SAVFI_E_NULL;
  const UpMask um{y, slope};
  const Win g{H, W, sy0, sx0, Hs, Ws, oy0, ox0, Hw, Ww, align_corners ? 1 : 0};
  if (int rc = check_window(g, planes)) return rc;
#ifndef SAVFI_UPSAMPLE_BWD_FORM
#define SAVFI_UPSAMPLE_BWD_FORM 0      // variant builds: 1 = the tiled form of rounds 2-4
#endif
  constexpr int form = SAVFI_UPSAMPLE_BWD_FORM;
  // the streaming form where it has the waves to hide its serial walk (>= 8 per SIMD at some strip height), else the tiled form
  int usr = 0;
  for (int cand = 32; cand >= 8 && !usr; cand >>= 1)
    if ((int64_t)savfi_cdiv(Ws, USW) * savfi_cdiv(Hs, cand) * planes >= 8192) usr = cand;
  if (form == 2) usr = 32;
  if (Ws >= 32 && form != 1 && usr) {
    dim3 grid(savfi_cdiv(savfi_cdiv(Ws, USW) * savfi_cdiv(Hs, usr), 4), planes, 1);
    hipLaunchKernelGGL(upsample2x_bwd_strip, grid, dim3(256), 0, (hipStream_t)stream, gout, gin, g, usr, um);
  } else if (Ws >= 32) {      // the separable, LDS-tiled form
    dim3 grid(savfi_cdiv(Hs, UBH) * savfi_cdiv(Ws, UBW), planes, 1);
    hipLaunchKernelGGL(upsample2x_bwd_tiled, grid, dim3(256), 0, (hipStream_t)stream, gout, gin, g, um);
  } else {
    dim3 grid(savfi_cdiv((int64_t)Hs * Ws, 256), planes, 1);
    hipLaunchKernelGGL(upsample2x_bwd, grid, dim3(256), 0, (hipStream_t)stream, gout, gin, g, um);
  }
  return savfi_launch_status();
}

extern "C" int savfi_upsample2x_fwd_f32(const float* in, float* out, int planes, int H, int W, int align_corners,
                                        void* stream) {
  return savfi_upsample2x_window_fwd_f32(in, out, planes, H, W, 0, 0, H, W, 0, 0, 2 * H, 2 * W, align_corners, stream);
}

extern "C" int savfi_upsample2x_bwd_f32(const float* gout, float* gin, int planes, int H, int W, int align_corners,
                                        void* stream) {
  return savfi_upsample2x_window_bwd_f32(gout, gin, planes, H, W, 0, 0, H, W, 0, 0, 2 * H, 2 * W, align_corners, stream);
}
