// Weight gradient of the K x K / stride 1 convolutions (K = 3, 5, 7; any zero padding) for gfx950, NCHW in and out,
// deterministic, on the bf16 matrix cores with error-free 3-way operand splits (six products, fp32 accumulate: the
// arithmetic of csrc/convk.hip -- fp32-equivalent, see the header there):
//
//   gw[t][co][ci][ky][kx] = sum over n = t (mod T), y, x of  gz[n][co][y][x] * x[n][ci][y + ky - pad][x + kx - pad]
//
// Replaces MIOpen's implicit-GEMM weight-gradient kernels (+ their two NCHW<->NHWC transposes, + atomics) under
// aten::convolution_backward for the 5x5 / 7x7 layers of VoxelFlow and Super SloMo (reference voxelflow/core/models/
// voxel_flow.py:357-470, superslomo/model.py:547-646; model_utils.py:308-366 MetaConv2dLayer -> F.conv2d): 25 % of a config-C3
// iteration + 5 % in the transposes (profiles/r02_c3_voxelflow_one_iteration.txt).
//
// GEMM view: M = co (A operand from gz), N = ci (B operand from x, shifted by the tap), K = pixels.  The k index of
// v_mfma_f32_16x16x32_bf16 is 8 consecutive values per lane, so a pixel-major operand would need 8 consecutive PIXELS of
// one channel per lane -- and a tap shift of one pixel would misalign every 16-byte read.  Instead both tiles are staged
// CHANNELS-LAST (three bf16 planes of [octet][row][col][8 channels], the image csrc/convk.hip stages) and the fragments
// come out of LDS through the gfx950 transpose read ds_read_b64_tr_b16: a 16-lane group supplies sixteen 8-byte addresses
// (4 pixels x 4 channel quads) and lane i receives channel i of those 4 pixels.  A tap shift is then an ADDRESS offset.
//
// Workgroup = 4 waves, all on the same 16*MT output channels x 16*NT input channels and the same pixels; wave w owns the
// taps w, w+4, w+8, ... (own accumulators, no reduction between waves).  Unit of staging = 4 output rows x 32 columns of one
// sample (gz tile + x tile with its K-1 halo); a workgroup walks a contiguous range of units, keeps its accumulators, and
// writes ONE partial block; convk_wgrad_reduce adds the partial blocks in a fixed order (no atomics).
#include "common.h"
#include <stdlib.h>
#include <algorithm>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef short bf16x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
__device__ float ckw_raw_buffer_load_f32(i32x4 rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.load.f32");

namespace {

constexpr int WG_THREADS = 256;
constexpr int UR = 4, UW = 32;          // unit: rows x columns of the cotangent

__device__ __forceinline__ i32x4 ckw_rsrc(const void* base, unsigned bytes) {
  const unsigned long long p = reinterpret_cast<unsigned long long>(base);
  i32x4 r;
  r.x = (int)(unsigned)p;
  r.y = (int)(unsigned)(p >> 32);
  r.z = (int)bytes;
  r.w = 0x00020000;
  return r;
}

__device__ __forceinline__ unsigned ckw_cvt_pk_bf16(float lo, float hi) {
  unsigned r;
  asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
  return r;
}

typedef float f32x2k __attribute__((ext_vector_type(2)));
// (the two subtractions of a pair as packed fp32 operations: v_pk_add_f32, 9 VALU per pair of elements instead of 11)
__device__ __forceinline__ void ckw_split8(const float (&v)[8], u32x4& p1, u32x4& p2, u32x4& p3) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const f32x2k ab = {v[2 * i], v[2 * i + 1]};
    const unsigned h1 = ckw_cvt_pk_bf16(ab.x, ab.y);
    const f32x2k r = ab - (f32x2k){__uint_as_float(h1 << 16), __uint_as_float(h1 & 0xffff0000u)};
    const unsigned h2 = ckw_cvt_pk_bf16(r.x, r.y);
    const f32x2k q = r - (f32x2k){__uint_as_float(h2 << 16), __uint_as_float(h2 & 0xffff0000u)};
    p1[i] = h1; p2[i] = h2; p3[i] = ckw_cvt_pk_bf16(q.x, q.y);
  }
}

// bytes per channel octet of an LDS image of `cells` 16-byte cells: = 64 (mod 256), so that the 32 lanes of a transpose read
// (4 pixels x 2 pixel-octets x 2 channel halves x 2 channel octets) touch 32 distinct 8-byte slots
__host__ __device__ constexpr int ckw_oct(int cells) { return (cells * 16 - 64 + 255) / 256 * 256 + 64; }

struct WgArgs {
  const float* x;       // [N][Ci][H][W]
  const float* gz;      // [N][Co][Ho][Wo]
  float* partial;       // [splits][T][Co][Ci][K*K]
  int N, T, Ci, Co, H, W, Ho, Wo, pad;
  int reflect;          // 1: x's border of width pad mirrors the image (the forward ran on nn.ReflectionPad2d(x))
  int cobs, cibs, splits, units, units_per_split, upr, ups;   // upr = row pairs per sample, ups = column segments per row
};

template <int KS, int MT, int NT>
struct WgGeom {
  static constexpr int TAPS = KS * KS;
  // taps w, w + 4, ... of wave w for all rows of a unit (FULL each), and the TAPS % 4 left-over taps for row w only: every wave
  // issues the same number of MFMAs (9 taps over 4 waves as 3 / 2 / 2 / 2 whole taps left three SIMDs idle a third of the time:
  // PMC 41 % MFMA busy with the first SIMD of every CU at 61 %)
  static constexpr int FULL = TAPS / 4, SHARED = TAPS % 4, TPW = FULL + SHARED;
  static_assert(UR == 4, "a shared tap's rows go one to each wave");
  static constexpr int GOCTS = 2 * MT, XOCTS = 2 * NT;
  static constexpr int GCELLS = UR * UW, XROWS = UR + KS - 1, XCOLS = UW + KS - 1, XCELLS = XROWS * XCOLS;
  static constexpr int GOCT = ckw_oct(GCELLS), XOCT = ckw_oct(XCELLS);
  static constexpr int GPLANE = GOCTS * GOCT, XPLANE = XOCTS * XOCT;
  static constexpr int XBASE = 3 * GPLANE;                    // x image behind the three cotangent planes
  static constexpr int LDS = 3 * GPLANE + 3 * XPLANE;
  static constexpr int GITEMS = GOCTS * GCELLS, XITEMS = XOCTS * XCELLS, ITEMS = GITEMS + XITEMS;
};

// P2: the cross terms of the six-product sum in their own accumulators (see csrc/convk.hip)
// NG: groups of four waves per workgroup.  Group ng works on input-channel tiles ng NT .. ng NT + NT - 1 of the workgroup's 16 NT NG input
// channels exactly as the four waves of an NG = 1 workgroup work on theirs (taps w & 3, w & 3 + 4, ...), against ONE staged cotangent tile:
// the cotangent (16 MT output channels x 128 pixels: most of a unit's bytes and of its split arithmetic) is loaded and split once for twice
// the MFMAs, and a CU holds one 8-wave workgroup instead of two 4-wave ones (the same two waves per SIMD).
template <int KS, int MT, int NT, bool P2, int NG = 1>
__global__ __launch_bounds__(WG_THREADS * NG, NG == 1 ? 2 : 1) void convk_wgrad_kernel(const WgArgs a) {
  using G = WgGeom<KS, MT, NT * NG>;
  constexpr int NTHR = WG_THREADS * NG;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wq = wv & 3, ng = wv >> 2;              // tap owner inside the group; group
  const int g = lane >> 4, sl = lane & 15, kq = sl >> 2, c4 = sl & 3;     // transpose-read source lane: pixel kq, channel quad c4

  int b = blockIdx.x;
  const int split = b % a.splits; b /= a.splits;
  const int cib = b % a.cibs; b /= a.cibs;
  const int cob = b % a.cobs;
  const int t = b / a.cobs;
  const int co0 = cob * 16 * MT, ci0 = cib * 16 * NT * NG;
  const int u_begin = split * a.units_per_split, u_end = min(u_begin + a.units_per_split, a.units);

  const size_t gplane = (size_t)a.Ho * a.Wo, xplane = (size_t)a.H * a.W;
  const int gplane_b = (int)(gplane * 4), xplane_b = (int)(xplane * 4);

  // ---- staging: items = (octet, cell); thread tid takes cotangent items tid + 256 k and input items tid + 256 k ----
  // An item's offset inside its sample (row dy / column dx of the unit, first channel of its octet) is fixed for the whole
  // launch: per unit one v_add moves it to the unit's origin, a compare pair masks what hangs over an image edge (the raw
  // buffer returns 0 beyond num_records), and the channel of each of the 8 loads rides in the scalar offset.  The first
  // version rebuilt item -> (octet, row, column) -> address with a predicate per element: ~40 VALU per item per unit, as much
  // as the split itself -- on a datapath the MFMAs share (PMC round 3: 3 VALU per MFMA).
  constexpr int GIPT = (G::GITEMS + NTHR - 1) / NTHR, XIPT = (G::XITEMS + NTHR - 1) / NTHR;
  constexpr int OOB = 0x7fffffff;
  int g_rel[GIPT], g_yx[GIPT], x_rel[XIPT], x_yx[XIPT];     // byte offset from the unit origin (OOB: no such item); dy << 16 | dx
  // octets whose channels all exist take the scalar-offset path; the channel tail of a ragged layer (Co = 3, 51, ...) is masked
  // per element (wave-uniform choice per launch)
  const bool g_full = (a.Co & 7) == 0, x_full = (a.Ci & 7) == 0;
#pragma unroll
  for (int k = 0; k < GIPT; ++k) {
    const int item = tid + NTHR * k;
    const int o = item / G::GCELLS, cell = item - o * G::GCELLS;
    const int dy = cell / UW, dx = cell - dy * UW;
    g_yx[k] = dy << 16 | dx;
    g_rel[k] = (item < G::GITEMS && co0 + 8 * o < a.Co) ? (dy * a.Wo + dx) * 4 + (co0 + 8 * o) * gplane_b : OOB;
  }
#pragma unroll
  for (int k = 0; k < XIPT; ++k) {
    const int item = tid + NTHR * k;
    const int o = item / G::XCELLS, cell = item - o * G::XCELLS;
    const int dy = cell / G::XCOLS, dx = cell - dy * G::XCOLS;
    x_yx[k] = dy << 16 | dx;
    x_rel[k] = (item < G::XITEMS && ci0 + 8 * o < a.Ci) ? (dy * a.W + dx) * 4 + (ci0 + 8 * o) * xplane_b : OOB;
  }
  float stage_g[GIPT][8], stage_x[XIPT][8];
  auto stage_load = [&](int u) {
    const int seg = u % a.ups, rp = (u / a.ups) % a.upr, n = (u / (a.ups * a.upr)) * a.T + t;
    const int y0 = rp * UR, x0 = seg * UW;
    const i32x4 grs = ckw_rsrc(a.gz + (size_t)n * a.Co * gplane, (unsigned)((size_t)a.Co * gplane * 4));
    const i32x4 xrs = ckw_rsrc(a.x + (size_t)n * a.Ci * xplane, (unsigned)((size_t)a.Ci * xplane * 4));
    const int g_org = (y0 * a.Wo + x0) * 4, x_org = ((y0 - a.pad) * a.W + (x0 - a.pad)) * 4;
    const bool g_in = y0 + UR <= a.Ho && x0 + UW <= a.Wo;                                   // the unit lies inside the map
    const bool x_in = y0 >= a.pad && x0 >= a.pad && y0 - a.pad + G::XROWS <= a.H && x0 - a.pad + G::XCOLS <= a.W;
#pragma unroll
    for (int k = 0; k < GIPT; ++k) {
      int voff = g_rel[k] == OOB ? OOB : g_rel[k] + g_org;
      if (!g_in) {
        const int dy = g_yx[k] >> 16, dx = g_yx[k] & 0xffff;
        voff = (y0 + dy < a.Ho && x0 + dx < a.Wo) ? voff : OOB;
      }
      if (g_full) {
#pragma unroll
        for (int e = 0; e < 8; ++e) stage_g[k][e] = ckw_raw_buffer_load_f32(grs, voff, e * gplane_b, 0);
      } else {
        const int ch0 = co0 + 8 * ((tid + NTHR * k) / G::GCELLS);
#pragma unroll
        for (int e = 0; e < 8; ++e) stage_g[k][e] = ckw_raw_buffer_load_f32(grs, ch0 + e < a.Co ? voff + e * gplane_b : OOB, 0, 0);
      }
    }
#pragma unroll
    for (int k = 0; k < XIPT; ++k) {
      int voff = x_rel[k] == OOB ? OOB : x_rel[k] + x_org;
      if (!x_in) {
        const int dy = x_yx[k] >> 16, dx = x_yx[k] & 0xffff;
        int yy = y0 - a.pad + dy, xx2 = x0 - a.pad + dx;
        if (a.reflect && x_rel[k] != OOB) {      // mirrored border: the item's own channel offset + the mirrored position
          yy = yy < 0 ? -yy : (yy >= a.H ? 2 * a.H - 2 - yy : yy);
          xx2 = xx2 < 0 ? -xx2 : (xx2 >= a.W ? 2 * a.W - 2 - xx2 : xx2);
          voff = x_rel[k] - (dy * a.W + dx) * 4 + (yy * a.W + xx2) * 4;
        }
        voff = ((unsigned)yy < (unsigned)a.H && (unsigned)xx2 < (unsigned)a.W) ? voff : OOB;
      }
      if (x_full) {
#pragma unroll
        for (int e = 0; e < 8; ++e) stage_x[k][e] = ckw_raw_buffer_load_f32(xrs, voff, e * xplane_b, 0);
      } else {
        const int ch0 = ci0 + 8 * ((tid + NTHR * k) / G::XCELLS);
#pragma unroll
        for (int e = 0; e < 8; ++e) stage_x[k][e] = ckw_raw_buffer_load_f32(xrs, ch0 + e < a.Ci ? voff + e * xplane_b : OOB, 0, 0);
      }
    }
  };
  auto stage_write = [&]() {
#pragma unroll
    for (int k = 0; k < GIPT; ++k) {
      const int item = tid + NTHR * k;
      if (item >= G::GITEMS) continue;
      const int o = item / G::GCELLS, cell = item - o * G::GCELLS;
      char* dst = smem + o * G::GOCT + cell * 16;
      u32x4 p1, p2, p3;
      ckw_split8(stage_g[k], p1, p2, p3);
      *reinterpret_cast<u32x4*>(dst) = p1;
      *reinterpret_cast<u32x4*>(dst + G::GPLANE) = p2;
      *reinterpret_cast<u32x4*>(dst + 2 * G::GPLANE) = p3;
    }
#pragma unroll
    for (int k = 0; k < XIPT; ++k) {
      const int item = tid + NTHR * k;
      if (item >= G::XITEMS) continue;
      const int o = item / G::XCELLS, cell = item - o * G::XCELLS;
      char* dst = smem + G::XBASE + o * G::XOCT + cell * 16;
      u32x4 p1, p2, p3;
      ckw_split8(stage_x[k], p1, p2, p3);
      *reinterpret_cast<u32x4*>(dst) = p1;
      *reinterpret_cast<u32x4*>(dst + G::XPLANE) = p2;
      *reinterpret_cast<u32x4*>(dst + 2 * G::XPLANE) = p3;
    }
  };

  // ---- fragment addresses (transpose reads): source lane (kq, c4) of 16-lane group g points at pixel 8 g + kq (+ 4) ----
  const int frag_lane = (c4 >> 1) * 0 + (c4 & 1) * 8;       // channel half inside a cell; the octet is added per tile
  int a_addr[MT], b_addr[NT];
#pragma unroll
  for (int m = 0; m < MT; ++m) a_addr[m] = (2 * m + (c4 >> 1)) * G::GOCT + (8 * g + kq) * 16 + frag_lane;
#pragma unroll
  for (int nn = 0; nn < NT; ++nn) b_addr[nn] = G::XBASE + (2 * (ng * NT + nn) + (c4 >> 1)) * G::XOCT + (8 * g + kq) * 16 + frag_lane;

  f32x4 acc[G::TPW][MT][NT], lo[P2 ? G::TPW : 1][P2 ? MT : 1][P2 ? NT : 1];
#pragma unroll
  for (int tp = 0; tp < G::TPW; ++tp)
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int nn = 0; nn < NT; ++nn) {
        acc[tp][m][nn] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (P2) lo[P2 ? tp : 0][P2 ? m : 0][P2 ? nn : 0] = (f32x4){0.f, 0.f, 0.f, 0.f};
      }

  auto tr_read = [&](int addr) -> bf16x8 {
    typedef __attribute__((address_space(3))) bf16x4 lds_b4;
    const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_b4*)(smem + addr));
    const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_b4*)(smem + addr + 4 * 16));
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  };

  constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};
  if (u_begin < u_end) stage_load(u_begin);
  for (int u = u_begin; u < u_end; ++u) {
    stage_write();
    __syncthreads();
    if (u + 1 < u_end) stage_load(u + 1);
    // software pipeline over the UR x TPW x NT groups of this unit (one group = one tap x one input-channel tile x MT
    // output-channel tiles = 6 MT MFMAs): the B fragments of the next group are read from LDS while the current group's MFMAs issue
    constexpr int NGRP = UR * G::FULL * NT;
    bf16x8 aq[MT][3], bq[2][3];
    auto load_a = [&](int r) {
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int p = 0; p < 3; ++p) aq[m][p] = tr_read(p * G::GPLANE + a_addr[m] + r * UW * 16);
    };
    auto load_b_at = [&](bf16x8 (&dst)[3], int r, int tap, int nn) {
      const int ky = tap / KS, kx = tap - ky * KS;
      const int boff = ((r + ky) * G::XCOLS + kx) * 16;
#pragma unroll
      for (int p = 0; p < 3; ++p) dst[p] = tr_read(p * G::XPLANE + b_addr[nn] + boff);
    };
    auto load_b = [&](int gi) {
      const int r = gi / (G::FULL * NT), tp = (gi / NT) % G::FULL, nn = gi % NT;
      load_b_at(bq[gi & 1], r, wq + 4 * tp, nn);
    };
    auto mfmas = [&](int tp, int nn, const bf16x8 (&b)[3]) {
#pragma unroll
      for (int q = 0; q < 6; ++q)
#pragma unroll
        for (int m = 0; m < MT; ++m) {
          if (P2 && q < 5)
            lo[P2 ? tp : 0][P2 ? m : 0][P2 ? nn : 0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                aq[m][PA[q]], b[PB[q]], lo[P2 ? tp : 0][P2 ? m : 0][P2 ? nn : 0], 0, 0, 0);
          else
            acc[tp][m][nn] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(aq[m][PA[q]], b[PB[q]], acc[tp][m][nn], 0, 0, 0);
        }
    };
    load_b(0);
#pragma unroll
    for (int gi = 0; gi < NGRP; ++gi) {
      const int r = gi / (G::FULL * NT), tp = (gi / NT) % G::FULL, nn = gi % NT;
      if (gi % (G::FULL * NT) == 0) load_a(r);          // A fragments of a row: once per row (their registers are busy until then)
      if (gi + 1 < NGRP) load_b(gi + 1);
      mfmas(tp, nn, bq[gi & 1]);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (G::SHARED > 0) {          // the left-over taps: this wave's row of each
      load_a(wq);
      load_b_at(bq[0], wq, 4 * G::FULL, 0);
#pragma unroll
      for (int gi = 0; gi < G::SHARED * NT; ++gi) {
        const int ts = gi / NT, nn = gi % NT;
        if (gi + 1 < G::SHARED * NT) load_b_at(bq[(gi + 1) & 1], wq, 4 * G::FULL + (gi + 1) / NT, (gi + 1) % NT);
        mfmas(G::FULL + ts, nn, bq[gi & 1]);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    __syncthreads();
  }

  // ---- this wave's taps of the partial block: D row 4 g + j = co, D column sl = ci ----
  float* __restrict__ pout = a.partial + ((size_t)split * a.T + t) * a.Co * a.Ci * G::TAPS;
  auto put = [&](int tap, int m, int nng, const f32x4& v) {      // nng: input-channel tile of the workgroup (group * NT + tile)
    const int ci = ci0 + 16 * nng + sl;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int co = co0 + 16 * m + 4 * g + j;
      if (co < a.Co && ci < a.Ci) pout[((size_t)co * a.Ci + ci) * G::TAPS + tap] = v[j];
    }
  };
#pragma unroll
  for (int tp = 0; tp < G::FULL; ++tp)
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int nn = 0; nn < NT; ++nn) {
        f32x4 v = acc[tp][m][nn];
        if (P2) v += lo[P2 ? tp : 0][P2 ? m : 0][P2 ? nn : 0];
        put(wq + 4 * tp, m, ng * NT + nn, v);
      }
  if (G::SHARED > 0) {
    // a shared tap's four row sums meet in LDS (the images are dead: every wave left the last unit's barrier), added in wave order
    f32x4* red = reinterpret_cast<f32x4*>(smem);
#pragma unroll
    for (int ts = 0; ts < G::SHARED; ++ts)
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int nn = 0; nn < NT; ++nn) {
          f32x4 v = acc[G::FULL + ts][m][nn];
          if (P2) v += lo[P2 ? G::FULL + ts : 0][P2 ? m : 0][P2 ? nn : 0];
          red[(((((ng * G::SHARED + ts) * MT + m) * NT + nn)) * 4 + wq) * 64 + lane] = v;
        }
    __syncthreads();
    // wave w sums the (group, tap, m, nn) items w, w + 4 NG, ...
    constexpr int RITEMS = G::SHARED * MT * NT;        // per group
    static_assert((size_t)NG * RITEMS * 4 * 64 * sizeof(f32x4) <= (size_t)G::LDS, "the shared taps' row sums fit the dead images");
#pragma unroll
    for (int it = 0; it < (NG * RITEMS + 4 * NG - 1) / (4 * NG); ++it) {
      const int item = wv + 4 * NG * it;
      if (item < NG * RITEMS) {
        const int ngi = item / RITEMS, rr = item - ngi * RITEMS;
        const int ts = rr / (MT * NT), m = (rr / NT) % MT, nn = rr % NT;
        f32x4 v = red[(item * 4 + 0) * 64 + lane];
#pragma unroll
        for (int k = 1; k < 4; ++k) v += red[(item * 4 + k) * 64 + lane];
        put(4 * G::FULL + ts, m, ngi * NT + nn, v);
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------------------------
// 3 x 3, ALL NINE TAPS PER WAVE on a RING of input rows (round 5).  The kernel above is bound by its LDS reads and its staging, not by the
// matrix pipe: with the taps dealt out to the waves every wave re-reads the whole cotangent tile (0.5 KB of fragments per MFMA -- what the
// LDS delivers per SIMD in the 16 cycles of an MFMA), a unit of 4 x 32 pixels needs a 6 x 34 input tile (1.6 x the split work) behind two
// workgroup barriers, and LDS has no room for a second buffer.
// Here a wave owns a 32 x 16 block of (output, input) channels with all nine taps in its own accumulators (9 x 2 tiles = 72 registers): the
// cotangent fragments of a 32-pixel row (6 transpose reads) are read ONCE for nine taps, an input fragment (3 reads) feeds 12 MFMAs -- 0.31 KB
// per MFMA.  Workgroup = 8 waves on 64 x 64 channels and the same pixels.  A unit is TWO rows x 32 pixels and a workgroup walks DOWN the rows
// of a 32-column segment: the input rows live in a ring of three row pairs (a unit reads two pairs, the third is being written), the
// cotangent tile is double buffered, so every input element is split ONCE (34 / 32 of the pixels), the staging of the next unit sits
// between the two rows of MFMAs of this one, and a step has ONE barrier.  An "item" of the pipeline = one input row pair (+ the cotangent
// rows of the unit it completes); the first item of a run of units in a segment brings only the pair above them.  512 threads: one
// cotangent cell x 8 channels and one input cell x 8 channels each (the 32 cells of the two halo columns go to the first half wave).  A
// lane ends up with the nine taps of four (co, ci) pairs: 36 contiguous bytes each.  BIAS: the channel sums of the cotangent (the bias
// gradient) ride on the staging registers -- wave o stages octet o of the 64 output channels.
// Measured (profiles/r05_wgrad3_forms.txt): 128 -> 128 @96x128, T = 4 x 2 samples: 161 us against 200 (tap-split kernel) and 183 (Winograd
// F(3x3, 2x2) on the fp32 matrix cores); timing-only ablations: no staging 123, MFMAs + fragment reads + epilogue alone 110 (86 = the MFMAs).
// ------------------------------------------------------------------------------------------------------------------
namespace ring {
constexpr int NTHR = 512, MT = 2, UR2 = 2;
constexpr int GCELLS = UR2 * UW;                       // 64 cells per cotangent buffer
constexpr int XCOLS = UW + 2, XPAIR = UR2 * XCOLS;     // 34 columns; 68 cells per row pair
constexpr int GOCT = ckw_oct(2 * GCELLS), XOCT = ckw_oct(3 * XPAIR);
constexpr int GPLANE = 8 * GOCT, XPLANE = 8 * XOCT, XBASE = 3 * GPLANE, LDS = 3 * GPLANE + 3 * XPLANE;
static_assert(8 * XPAIR == NTHR + 32, "one input cell per thread + 32 for the first half wave");
struct Item {            // wave-uniform
  int n, seg, xpair;     // sample, column segment, input row pair (rows 2 xpair - pad, + 1)
  int has_g;             // this item also brings the cotangent rows 2 (xpair - 1), + 1 and completes that unit
  int xslot, gbuf;       // ring slot of the pair (0..2), cotangent buffer (0 / 1)
  int valid;
};
}  // namespace ring

template <bool BIAS>
__global__ __launch_bounds__(ring::NTHR, 1) void convk_wgrad3_ring(const WgArgs a, float* __restrict__ bias_partial) {
  using namespace ring;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wv & 1, wn = wv >> 1;
  const int g = lane >> 4, sl = lane & 15, kq = sl >> 2, c4 = sl & 3;     // transpose-read source lane: pixel kq, channel quad c4

  int b = blockIdx.x;
  const int split = b % a.splits; b /= a.splits;
  const int cib = b % a.cibs; b /= a.cibs;
  const int cob = b % a.cobs;
  const int t = b / a.cobs;
  const int co0 = cob * 64, ci0 = cib * 64;
  const int u_begin = split * a.units_per_split, u_end = min(u_begin + a.units_per_split, a.units);
  const size_t gplane = (size_t)a.Ho * a.Wo, xplane = (size_t)a.H * a.W;
  const int gplane_b_ = (int)(gplane * 4), xplane_b_ = (int)(xplane * 4);
  constexpr int OOB = 0x7fffffff;

  // ---- this thread's staging cells (fixed for the launch) ----
  const int go = tid >> 6, gcell = tid & 63, gdy = gcell >> 5, gdx = gcell & 31;
  const int g_rel = co0 + 8 * go < a.Co ? (gdy * a.Wo + gdx) * 4 + (co0 + 8 * go) * gplane_b_ : OOB;
  const int g_dst = go * GOCT + gcell * 16;
  int xo[2], xdy[2], xdx[2], x_rel[2], x_dst[2];
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int item = k == 0 ? tid : NTHR + (tid & 31);
    xo[k] = item / XPAIR;
    const int cell = item - xo[k] * XPAIR;
    xdy[k] = cell / XCOLS; xdx[k] = cell - xdy[k] * XCOLS;
    x_rel[k] = ci0 + 8 * xo[k] < a.Ci ? (xdy[k] * a.W + xdx[k]) * 4 + (ci0 + 8 * xo[k]) * xplane_b_ : OOB;
    x_dst[k] = XBASE + xo[k] * XOCT + cell * 16;
  }
  const bool extra = tid < 32;                               // wave 0, first half: the 32 cells beyond 512
  const bool g_full = (a.Co & 7) == 0, x_full = (a.Ci & 7) == 0;

  float sg[8], sx[2][8];
  auto stage_load = [&](const Item& it) {
    // the channel strides enter the loads' scalar offsets as e * stride; kept opaque here so that the fourteen products are recomputed
    // (SALU) per item instead of living in SGPRs across the MFMA rows (they were spilled to VGPR lanes: 49 v_readlane + hazard nops per step)
    int gplane_b = gplane_b_, xplane_b = xplane_b_;
    asm volatile("" : "+s"(gplane_b), "+s"(xplane_b));
    const int x0 = it.seg * UW;
    if (it.has_g) {
      const int y0 = (it.xpair - 1) * UR2;
      const i32x4 grs = ckw_rsrc(a.gz + (size_t)it.n * a.Co * gplane, (unsigned)((size_t)a.Co * gplane * 4));
      int voff = g_rel == OOB ? OOB : g_rel + (y0 * a.Wo + x0) * 4;
      if (!(y0 + UR2 <= a.Ho && x0 + UW <= a.Wo)) voff = (y0 + gdy < a.Ho && x0 + gdx < a.Wo) ? voff : OOB;
      if (g_full) {
#pragma unroll
        for (int e = 0; e < 8; ++e) sg[e] = ckw_raw_buffer_load_f32(grs, voff, e * gplane_b, 0);
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) sg[e] = ckw_raw_buffer_load_f32(grs, co0 + 8 * go + e < a.Co ? voff + e * gplane_b : OOB, 0, 0);
      }
    }
    const int yx = it.xpair * UR2 - a.pad, xx = x0 - a.pad;
    const i32x4 xrs = ckw_rsrc(a.x + (size_t)it.n * a.Ci * xplane, (unsigned)((size_t)a.Ci * xplane * 4));
    const bool x_in = yx >= 0 && xx >= 0 && yx + UR2 <= a.H && xx + XCOLS <= a.W;
    const int x_org = (yx * a.W + xx) * 4;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      if (k == 1 && !extra) continue;
      int voff = x_rel[k] == OOB ? OOB : x_rel[k] + x_org;
      if (!x_in) {
        int yy = yx + xdy[k], xx2 = xx + xdx[k];
        if (a.reflect && x_rel[k] != OOB) {
          yy = yy < 0 ? -yy : (yy >= a.H ? 2 * a.H - 2 - yy : yy);
          xx2 = xx2 < 0 ? -xx2 : (xx2 >= a.W ? 2 * a.W - 2 - xx2 : xx2);
          voff = x_rel[k] - (xdy[k] * a.W + xdx[k]) * 4 + (yy * a.W + xx2) * 4;
        }
        voff = ((unsigned)yy < (unsigned)a.H && (unsigned)xx2 < (unsigned)a.W) ? voff : OOB;
      }
      if (x_full) {
#pragma unroll
        for (int e = 0; e < 8; ++e) sx[k][e] = ckw_raw_buffer_load_f32(xrs, voff, e * xplane_b, 0);
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) sx[k][e] = ckw_raw_buffer_load_f32(xrs, ci0 + 8 * xo[k] + e < a.Ci ? voff + e * xplane_b : OOB, 0, 0);
      }
    }
  };
  float bsum[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) bsum[e] = 0.f;
  const bool do_bias = BIAS && cib == 0;
  auto stage_write = [&](const Item& it) {
    u32x4 p1, p2, p3;
    if (it.has_g) {
      if (do_bias) {
#pragma unroll
        for (int e = 0; e < 8; ++e) bsum[e] += sg[e];
      }
      char* dst = smem + g_dst + it.gbuf * (GCELLS * 16);
      ckw_split8(sg, p1, p2, p3);
      *reinterpret_cast<u32x4*>(dst) = p1;
      *reinterpret_cast<u32x4*>(dst + GPLANE) = p2;
      *reinterpret_cast<u32x4*>(dst + 2 * GPLANE) = p3;
    }
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      if (k == 1 && !extra) continue;
      char* dst = smem + x_dst[k] + it.xslot * (XPAIR * 16);
      ckw_split8(sx[k], p1, p2, p3);
      *reinterpret_cast<u32x4*>(dst) = p1;
      *reinterpret_cast<u32x4*>(dst + XPLANE) = p2;
      *reinterpret_cast<u32x4*>(dst + 2 * XPLANE) = p3;
    }
  };

  // ---- the item sequence of this workgroup: units u_begin .. u_end - 1, row pair fastest inside (sample, segment) ----
  int u = u_begin, xcount = 0, gcount = 0;
  bool fresh = true;
  auto next_item = [&]() -> Item {
    Item it;
    it.valid = u < u_end;
    if (!it.valid) { it.has_g = 0; it.n = it.seg = it.xpair = it.xslot = it.gbuf = 0; return it; }
    const int rp = u % a.upr, rest = u / a.upr;
    it.seg = rest % a.ups;
    it.n = (rest / a.ups) * a.T + t;
    it.xslot = xcount % 3; ++xcount;
    if (fresh) {
      it.xpair = rp; it.has_g = 0; it.gbuf = 0; fresh = false;
    } else {
      it.xpair = rp + 1; it.has_g = 1; it.gbuf = gcount & 1; ++gcount;
      ++u;
      fresh = (u % a.upr) == 0;            // the next unit starts a new segment (or sample)
    }
    return it;
  };

  // ---- fragment addresses (transpose reads) ----
  const int frag_lane = (c4 & 1) * 8;
  int a_addr[MT];
#pragma unroll
  for (int m = 0; m < MT; ++m) a_addr[m] = (2 * (MT * wm + m) + (c4 >> 1)) * GOCT + (8 * g + kq) * 16 + frag_lane;
  const int b_addr = XBASE + (2 * wn + (c4 >> 1)) * XOCT + (8 * g + kq) * 16 + frag_lane;

  f32x4 acc[9][MT];
#pragma unroll
  for (int tp = 0; tp < 9; ++tp)
#pragma unroll
    for (int m = 0; m < MT; ++m) acc[tp][m] = (f32x4){0.f, 0.f, 0.f, 0.f};

  auto tr_read = [&](int addr, int imm) -> bf16x8 {
    typedef __attribute__((address_space(3))) bf16x4 lds_b4;
    const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_b4*)(smem + addr + imm));
    const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_b4*)(smem + addr + imm + 4 * 16));
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  };
  constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};

  Item cur = next_item(), prev;
  prev.valid = 0; prev.has_g = 0; prev.xslot = 0; prev.gbuf = 0;
  if (cur.valid) stage_load(cur);
  while (cur.valid || prev.has_g) {
    const Item nxt = next_item();
    // the unit completed by `prev`: cotangent buffer prev.gbuf, input rows 0, 1 in the slot before prev.xslot, rows 2, 3 in prev.xslot
    bf16x8 aq[2][MT][3], bq[2][3];
    int brow[4];
    {
      const int s1 = prev.xslot, s0 = s1 == 0 ? 2 : s1 - 1;
      brow[0] = b_addr + (s0 * UR2) * XCOLS * 16; brow[1] = brow[0] + XCOLS * 16;
      brow[2] = b_addr + (s1 * UR2) * XCOLS * 16; brow[3] = brow[2] + XCOLS * 16;
    }
    const int abase = prev.gbuf * (GCELLS * 16);
    auto load_a = [&](int r) {
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int p = 0; p < 3; ++p) aq[r][m][p] = tr_read(a_addr[m] + abase, p * GPLANE + r * UW * 16);
    };
    auto load_b = [&](int gi) {
      const int r = gi / 9, tap = gi % 9, ky = tap / 3, kx = tap % 3;
#pragma unroll
      for (int p = 0; p < 3; ++p) bq[gi & 1][p] = tr_read(brow[r + ky], p * XPLANE + kx * 16);
    };
    // row 0 reads its own fragments first (the one exposed LDS latency of a step); row 1's cotangent fragments and first input fragment are
    // requested inside row 0, so that the staging between the rows does not sit in front of a wait
    auto row_of_mfmas = [&](int r) {
      if (r == 0) { load_a(0); load_b(0); }
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        const int gi = 9 * r + tap;
        if (gi + 1 < 18) load_b(gi + 1);
        if (r == 0 && tap == 4) load_a(1);
#pragma unroll
        for (int q = 0; q < 6; ++q)
#pragma unroll
          for (int m = 0; m < MT; ++m)
            acc[tap][m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(aq[r][m][PA[q]], bq[gi & 1][PB[q]], acc[tap][m], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    };
    if (prev.has_g) row_of_mfmas(0);
    if (cur.valid) stage_write(cur);                 // (waits for its loads: issued a step ago)
    if (nxt.valid) stage_load(nxt);
    __builtin_amdgcn_sched_barrier(0);
    if (prev.has_g) row_of_mfmas(1);
    __syncthreads();
    prev = cur; cur = nxt;
  }

  if (do_bias) {          // wave `go` holds the sums of its octet's eight channels over this workgroup's units: lanes meet, lane 0 stores
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float s = wave_sum(bsum[e]);
      const int co = co0 + 8 * go + e;
      if (lane == 0 && co < a.Co) bias_partial[((size_t)split * a.T + t) * a.Co + co] = s;
    }
  }

  // ---- D row 4 g + j = co, D column sl = ci; this lane's nine taps of a (co, ci) pair are contiguous in the partial block ----
  float* __restrict__ pout = a.partial + ((size_t)split * a.T + t) * a.Co * a.Ci * 9;
  const int ci = ci0 + 16 * wn + sl;
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int co = co0 + 16 * (MT * wm + m) + 4 * g + j;
      if (co < a.Co && ci < a.Ci) {
        float* o = pout + ((size_t)co * a.Ci + ci) * 9;
#pragma unroll
        for (int tp = 0; tp < 9; ++tp) o[tp] = acc[tp][m][j];
      }
    }
}

// Sum of the partial blocks in a FIXED order (deterministic): a workgroup = 32 outputs x 8 split groups; thread (output, group)
// adds the splits k = group (mod 8) in increasing order, the eight group sums meet in LDS and are added in group order.
__device__ __forceinline__ void wgrad_reduce_block(const float* __restrict__ partial, float* __restrict__ out, long long n, int splits, unsigned block) {
  __shared__ float part[8][32];
  const int o = threadIdx.x & 31, grp = threadIdx.x >> 5;
  const long long i = (long long)block * 32 + o;
  float s = 0.f;
  if (i < n)
    for (int k = grp; k < splits; k += 8) s += partial[(size_t)k * n + i];
  part[grp][o] = s;
  __syncthreads();
  if (grp == 0 && i < n) {
    float tot = part[0][o];
#pragma unroll
    for (int q = 1; q < 8; ++q) tot += part[q][o];
    out[i] = tot;
  }
}

// The same sums in the same order, four neighbouring outputs per thread as one 16-byte load per split (n a multiple of 4: every
// partial block starts 16-byte aligned): a quarter of the workgroups, 512-byte runs per half wave instead of 128.
__device__ __forceinline__ void wgrad_reduce_block4(const float* __restrict__ partial, float* __restrict__ out, long long n, int splits, unsigned block) {
  __shared__ float4 part[8][32];
  const int o = threadIdx.x & 31, grp = threadIdx.x >> 5;
  const long long i = ((long long)block * 32 + o) * 4;
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  if (i < n)
    for (int k = grp; k < splits; k += 8) {
      const float4 v = *reinterpret_cast<const float4*>(partial + (size_t)k * n + i);
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
  part[grp][o] = s;
  __syncthreads();
  if (grp == 0 && i < n) {
    float4 tot = part[0][o];
#pragma unroll
    for (int q = 1; q < 8; ++q) { tot.x += part[q][o].x; tot.y += part[q][o].y; tot.z += part[q][o].z; tot.w += part[q][o].w; }
    *reinterpret_cast<float4*>(out + i) = tot;
  }
}

// ONE launch for the weight gradient and, behind its workgroups, the bias sums (a launch of their own was 4 us, 381 times in a CAIN
// meta-iteration at 720p)
template <bool V4>
__global__ __launch_bounds__(256) void convk_wgrad_reduce(const float* __restrict__ partial, float* __restrict__ gw, long long n, int splits,
                                                          unsigned main_blocks, const float* __restrict__ bias_partial, float* __restrict__ gb,
                                                          long long nb) {
  if (blockIdx.x >= main_blocks) { wgrad_reduce_block(bias_partial, gb, nb, splits, blockIdx.x - main_blocks); return; }
  if (V4) wgrad_reduce_block4(partial, gw, n, splits, blockIdx.x);
  else wgrad_reduce_block(partial, gw, n, splits, blockIdx.x);
}

struct WgPlan { int mt, nt, ng, cobs, cibs, units, upr, ups, splits, ups_per_split, form; };   // form (3 x 3): 0 taps dealt out to the waves, 2 all taps per wave on a ring of input rows

inline int savfi_cu_count() {
  static int cus[32] = {0};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return 256;
  int& n = cus[dev & 31];
  if (n == 0) {
    int v = 0;
    n = (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) ? v : 256;
  }
  return n;
}

inline int wg_plan(WgPlan& p, int N, int T, int Ci, int Co, int H, int W, int K, int pad, bool precise) {
  if (N <= 0 || T <= 0 || Ci <= 0 || Co <= 0 || H <= 0 || W <= 0 || N % T != 0 || pad < 0 || pad > K - 1) return SAVFI_E_SHAPE;
  if (K != 3 && K != 5 && K != 7) return SAVFI_E_UNSUPPORTED;
  const int Ho = H + 2 * pad - K + 1, Wo = W + 2 * pad - K + 1;
  if (Ho <= 0 || Wo <= 0) return SAVFI_E_SHAPE;
  if ((int64_t)Ci * H * W >= (1ll << 29) || (int64_t)Co * Ho * Wo >= (1ll << 29)) return SAVFI_E_TOOBIG;
  p.mt = (K == 3 && Co >= 192) ? 4 : 2;
#ifdef SAVFI_WGRAD_MT4          // variant builds: the 4-tile form from this many output channels (5x5 / 7x7: 7 / 13 taps per wave x 4 tiles would not fit)
  if (K == 3 && Co >= SAVFI_WGRAD_MT4) p.mt = 4;
#endif
  // two input-channel tiles per workgroup split the cotangent tile once for 32 input channels (0.7x the VALU per MFMA) but cost the
  // third workgroup per CU: measured 10-15 % SLOWER on every layer (profiles/r03_wgrad_variants.txt); kept behind SAVFI_WGRAD_NT=2
  p.nt = 1;
#if defined(SAVFI_WGRAD_NT) && SAVFI_WGRAD_NT == 2
  if (!precise && K != 7 && p.mt == 2 && Ci > 16) p.nt = 2;
#endif
  // 8-wave workgroups on 64 x 32 channels (NG = 2, see the kernel) for the 4-tile variant, opt-in (SAVFI_WGRAD_NG=2): measured equal to
  // the 4-wave form (CAIN 192 -> 192 @96x160 N = 2: 149 vs 153 us per call; C5 slice 7.23 vs 7.34 steps/s) -- splitting the cotangent tile
  // once for twice the MFMAs buys nothing, the kernel is not bound by its VALU work
  p.ng = 1;
#if defined(SAVFI_WGRAD_NG) && SAVFI_WGRAD_NG == 2
  if (K == 3 && p.mt == 4 && !precise && Ci >= 32) p.ng = 2;
#endif
  // 3 x 3 with at least 48 channels on both sides: all nine taps per wave on 64 x 64 channel blocks, one 8-wave workgroup per CU
  // (SAVFI_WGRAD3_FORM=0: the tap-split kernel, for A/Bs)
#ifndef SAVFI_WGRAD3_FORM
#define SAVFI_WGRAD3_FORM 2
#endif
  p.form = (K == 3 && !precise && Co >= 48 && Ci >= 48 && SAVFI_WGRAD3_FORM == 2) ? 2 : 0;
  if (p.form) { p.mt = 4; p.nt = 4; p.ng = 1; }
  p.cobs = (Co + 16 * p.mt - 1) / (16 * p.mt);
  p.cibs = (Ci + 16 * p.nt * p.ng - 1) / (16 * p.nt * p.ng);
  const int ur = p.form == 2 ? ring::UR2 : UR;
  p.upr = (Ho + ur - 1) / ur;
  p.ups = (Wo + UW - 1) / UW;
  p.units = (N / T) * p.upr * p.ups;
  const int blocks = T * p.cobs * p.cibs;
  // How many pieces the pixel range is cut into.  A launch runs in ROUNDS of `slots` resident workgroups (2 per CU for the 4-tile and the
  // 5x5 / 7x7 / two-accumulator variants: 176-256 registers; 3 for <3,2,1>: 126 registers, 46 KB of LDS), and a round lasts as long as a
  // workgroup's units: time ~ rounds x (units per split + 1) + a little per split for the partial blocks and their reduction.  The old rule
  // (~768 workgroups per launch) left CAIN's 192 -> 192 @96x160 layer with 792 workgroups for 512 slots: a second round 55 % full, 167 us;
  // 504 workgroups of 18 units instead of 792 of 11: 145 us (SAVFI_WGRAD_TARGET=n restores the old rule with n workgroups).
  int splits;
#ifndef SAVFI_WGRAD_TARGET
#define SAVFI_WGRAD_TARGET 0
#endif
  constexpr int target = SAVFI_WGRAD_TARGET;
  if (target > 0) {
    splits = (target + blocks - 1) / blocks;
  } else {
    const int per_cu = (p.ng == 2 || p.form) ? 1 : (K == 3 && p.mt == 2 && p.nt == 1 && !precise) ? 3 : 2;
    const int64_t slots = (int64_t)per_cu * savfi_cu_count();
    double best_cost = 0.0;
    splits = 1;
    const int smax = (int)std::min<int64_t>(p.units, 4 * slots / blocks + 1);
    for (int sp = 1; sp <= smax; ++sp) {
      const int ups = (p.units + sp - 1) / sp;
      if ((p.units + ups - 1) / ups != sp) continue;                   // not a distinct cut
      const int64_t rounds = ((int64_t)blocks * sp + slots - 1) / slots;
      const double cost = p.form == 2 ? (double)rounds * (ups + 3.0) + 0.12 * sp : (double)rounds * (ups + 1.0) + 0.06 * sp;
      if (sp == 1 || cost < best_cost) { best_cost = cost; splits = sp; }
    }
  }
  if (splits > p.units) splits = p.units;
  if (splits < 1) splits = 1;
  p.ups_per_split = (p.units + splits - 1) / splits;
  p.splits = (p.units + p.ups_per_split - 1) / p.ups_per_split;
  return SAVFI_OK;
}

int launch_wgrad3_ring(const WgArgs& a, int blocks, float* bias_partial, hipStream_t stream) {
  static uint32_t configured = 0, configured_b = 0;
  const int rc = bias_partial ? savfi_ensure_dynamic_lds(reinterpret_cast<const void*>(convk_wgrad3_ring<true>), ring::LDS, configured_b)
                              : savfi_ensure_dynamic_lds(reinterpret_cast<const void*>(convk_wgrad3_ring<false>), ring::LDS, configured);
  if (rc != SAVFI_OK) return rc;
  if (bias_partial) hipLaunchKernelGGL(convk_wgrad3_ring<true>, dim3(blocks), dim3(ring::NTHR), ring::LDS, stream, a, bias_partial);
  else hipLaunchKernelGGL(convk_wgrad3_ring<false>, dim3(blocks), dim3(ring::NTHR), ring::LDS, stream, a, bias_partial);
  return savfi_launch_status();
}

template <int KS, int MT, int NT, bool P2 = false, int NG = 1>
int launch_wgrad(const WgArgs& a, int blocks, hipStream_t stream) {
  using G = WgGeom<KS, MT, NT * NG>;
  static uint32_t configured = 0;
  auto kern = convk_wgrad_kernel<KS, MT, NT, P2, NG>;
  if (G::LDS > 64 * 1024) {
    const int rc = savfi_ensure_dynamic_lds(reinterpret_cast<const void*>(kern), G::LDS, configured);
    if (rc != SAVFI_OK) return rc;
  }
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(WG_THREADS * NG), G::LDS, stream, a);
  return savfi_launch_status();
}

}  // namespace

extern "C" int64_t savfi_convk_wgrad_workspace_floats(int N, int T, int Ci, int Co, int H, int W, int K, int pad) {
  WgPlan p, q;
  int rc = wg_plan(p, N, T, Ci, Co, H, W, K, pad, false);
  if (rc != SAVFI_OK) return rc;
  rc = wg_plan(q, N, T, Ci, Co, H, W, K, pad, true);
  if (rc != SAVFI_OK) return rc;
  return (int64_t)(p.splits > q.splits ? p.splits : q.splits) * T * Co * (Ci * K * K + 1);     // (+ 1: the bias sums' partial blocks)
}

extern "C" int savfi_convk_wgrad_sums_bias(int N, int T, int Ci, int Co, int H, int W, int K, int pad) {
  WgPlan p;
  return wg_plan(p, N, T, Ci, Co, H, W, K, pad, false) == SAVFI_OK && p.form == 2 ? 1 : 0;
}

extern "C" int savfi_convk_wgrad_tasks_f32(const float* x, const float* gz, float* gw, float* workspace, int N, int T, int Ci, int Co,
                                           int H, int W, int K, int pad, int precise, void* stream) {
  return savfi_convk_wgrad_tasks_reflect_f32(x, gz, gw, workspace, N, T, Ci, Co, H, W, K, pad, precise, 0, stream);
}

namespace {
int convk_wgrad_run(const float* x, const float* gz, float* gw, float* gb, float* workspace, int N, int T, int Ci, int Co, int H, int W, int K,
                    int pad, int precise, int reflect, void* stream);
}

extern "C" int savfi_convk_wgrad_tasks_reflect_f32(const float* x, const float* gz, float* gw, float* workspace, int N, int T, int Ci,
                                                   int Co, int H, int W, int K, int pad, int precise, int reflect, void* stream) {
  return convk_wgrad_run(x, gz, gw, nullptr, workspace, N, T, Ci, Co, H, W, K, pad, precise, reflect, stream);
}

extern "C" int savfi_convk_wgrad_tasks_bias_f32(const float* x, const float* gz, float* gw, float* gb, float* workspace, int N, int T, int Ci,
                                                int Co, int H, int W, int K, int pad, int reflect, void* stream) {
  if (!gb) return SAVFI_E_NULL;
  if (!savfi_convk_wgrad_sums_bias(N, T, Ci, Co, H, W, K, pad)) return SAVFI_E_UNSUPPORTED;
  return convk_wgrad_run(x, gz, gw, gb, workspace, N, T, Ci, Co, H, W, K, pad, 0, reflect, stream);
}

namespace {
int convk_wgrad_run(const float* x, const float* gz, float* gw, float* gb, float* workspace, int N, int T, int Ci, int Co, int H, int W, int K,
                    int pad, int precise, int reflect, void* stream) {
  if (reflect && (pad >= H || pad >= W)) return SAVFI_E_UNSUPPORTED;
  if (!x || !gz || !gw || !workspace) return SAVFI_E_NULL;
  WgPlan p;
  int rc = wg_plan(p, N, T, Ci, Co, H, W, K, pad, precise != 0);
  if (rc != SAVFI_OK) return rc;
  WgArgs a;
  a.x = x; a.gz = gz; a.partial = workspace;
  a.N = N; a.T = T; a.Ci = Ci; a.Co = Co; a.H = H; a.W = W; a.pad = pad; a.reflect = reflect ? 1 : 0;
  a.Ho = H + 2 * pad - K + 1; a.Wo = W + 2 * pad - K + 1;
  a.cobs = p.cobs; a.cibs = p.cibs; a.splits = p.splits; a.units = p.units; a.units_per_split = p.ups_per_split;
  a.upr = p.upr; a.ups = p.ups;
  const int blocks = T * p.cobs * p.cibs * p.splits;
  hipStream_t st = (hipStream_t)stream;
  float* bias_partial = gb ? workspace + (size_t)p.splits * T * Co * Ci * K * K : nullptr;
  if (p.form == 2) rc = launch_wgrad3_ring(a, blocks, bias_partial, st);
  else if (K == 3 && precise) rc = p.mt == 4 ? launch_wgrad<3, 4, 1, true>(a, blocks, st) : launch_wgrad<3, 2, 1, true>(a, blocks, st);
  else if (K == 3 && p.ng == 2) rc = launch_wgrad<3, 4, 1, false, 2>(a, blocks, st);
  else if (K == 3) rc = p.mt == 4 ? launch_wgrad<3, 4, 1>(a, blocks, st) : p.nt == 2 ? launch_wgrad<3, 2, 2>(a, blocks, st) : launch_wgrad<3, 2, 1>(a, blocks, st);
  else if (K == 5) rc = precise ? launch_wgrad<5, 2, 1, true>(a, blocks, st) : p.nt == 2 ? launch_wgrad<5, 2, 2>(a, blocks, st) : launch_wgrad<5, 2, 1>(a, blocks, st);
  else rc = launch_wgrad<7, 2, 1>(a, blocks, st);       // 13 taps per wave: no room for a second accumulator set
  if (rc != SAVFI_OK) return rc;
  const long long n = (long long)T * Co * Ci * K * K;
  const bool v4 = n % 4 == 0 && (((uintptr_t)workspace | (uintptr_t)gw) & 15u) == 0;
  const long long nb = gb ? (long long)T * Co : 0;
  const unsigned main_blocks = (unsigned)(v4 ? (n / 4 + 31) / 32 : (n + 31) / 32), bias_blocks = (unsigned)((nb + 31) / 32);
  if (v4) hipLaunchKernelGGL(convk_wgrad_reduce<true>, dim3(main_blocks + bias_blocks), dim3(256), 0, st, workspace, gw, n, p.splits, main_blocks, bias_partial, gb, nb);
  else hipLaunchKernelGGL(convk_wgrad_reduce<false>, dim3(main_blocks + bias_blocks), dim3(256), 0, st, workspace, gw, n, p.splits, main_blocks, bias_partial, gb, nb);
  return savfi_launch_status();
}
}  // namespace
