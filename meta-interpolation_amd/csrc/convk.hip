// K x K / stride 1 convolution (K = 3, 5, 7; forward and data gradient, any zero padding) for gfx950 as a DIRECT implicit
// GEMM on the bf16 matrix cores, with fp32 operands split error-free into three bf16 pieces each:
//
//     a = a1 + a2 + a3   (exactly: 3 x 8 significant bits, round-to-nearest pieces),  same for b
//     a*b = a1b1 + (a1b2 + a2b1) + (a1b3 + a3b1 + a2b2)  + O(2^-26 |ab|)          6 products, fp32 accumulate
//
// Why.  fp32 MFMA runs at the fp32 VECTOR rate (157 TFLOP/s); v_mfma_f32_16x16x32_bf16 is 16x faster, so six of them do the
// work of one fp32 MFMA K-slab in 6/16 of the time: 350-390 "fp32-equivalent" TFLOP/s from registers or LDS
// (tools/bf16_split_probe.hip, profiles/r03_bf16_split_probe.txt).  The result is NOT narrower than fp32 arithmetic: against
// an fp64 reference the 6-product sum is as close as the fmaf chain of the fp32 MFMA or closer (K = 288..4608: 1.0-1.7e-6
// vs 0.5-1.9e-6 of max|C|; the dropped terms are below half an fp32 ulp of every product), a one-hot operand reproduces
// the other one bit for bit, and unlike Winograd F(2x2,3x3) nothing is amplified (2e-7 there).  This is what lets the
// 5x5 / 7x7 layers of VoxelFlow and Super SloMo (reference voxelflow/core/models/voxel_flow.py:357-470, superslomo/
// model.py:547-646) and VoxelFlow's rounding-sensitive 3x3 layers leave MIOpen.
//
//   out[n][co][y][x] = act( bias[t][co] + sum_{ci,ky,kx} w[t][co][ci][ky][kx] * in[n][ci][y+ky-pad][x+kx-pad] ),  t = n % T
//
// GEMM view: M = 16 consecutive output pixels of a row (A operand: input patches), N = 16 output channels (B: weights),
// K = (tap, 8 input channels): one MFMA k-step = 4 (tap, channel-octet) pairs.  Workgroup = 4 waves = a TH x TW pixel tile
// (8 x 32 or 16 x 16) x 16*NT output channels; a wave owns 4 M-tiles x NT N-tiles (64 px x 64 channels at NT = 4).
//   * The input tile (+ halo) of a chunk of 8*QC channels is staged ONCE per workgroup: fp32 NCHW loads (coalesced along x),
//     split in registers (v_cvt_pk_bf16_f32), written as three bf16 planes in [octet][row][col][8 channels] order, so that an
//     A fragment (16 pixels x 8 channels of one tap) is ONE conflict-free ds_read_b128 per plane with an immediate offset
//     per tap.  The split costs ~5.5 VALU per element and is amortised over K*K*16*NT MACs.
//   * Weights are packed once per weight version (convk_pack) as bf16 triples in B-fragment order: a lane fetches its 8 values
//     of a plane with one coalesced 16-byte load straight from L2 -- no LDS, no transform in the loop.
//   * Main loop per k-step and wave: 12 ds_read_b128 + 3*NT global 16-byte loads feed 96 (NT = 4) MFMAs; the six products
//     of a tile are issued across the 16 accumulators, small terms first.
//   * Epilogue: bias + (leaky) ReLU, 16-byte stores (4 consecutive pixels per lane).
// The data gradient is the same kernel on a filter packed flipped / transposed (mode 1) with padding K-1-pad.
#include "common.h"
#include <stdio.h>
#include <stdlib.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

typedef int i32x4 __attribute__((ext_vector_type(4)));
__device__ float ck_raw_buffer_load_f32(i32x4 rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.load.f32");
__device__ u32x4 ck_raw_buffer_load_x4(i32x4 rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.load.v4i32");

namespace {

constexpr int CK_THREADS = 256;

__device__ __forceinline__ i32x4 ck_rsrc(const void* base, unsigned bytes) {
  const unsigned long long p = reinterpret_cast<unsigned long long>(base);
  i32x4 r;
  r.x = (int)(unsigned)p;
  r.y = (int)(unsigned)(p >> 32);       // stride 0: raw buffer
  r.z = (int)bytes;
  r.w = 0x00020000;
  return r;
}

__host__ __device__ constexpr int ck_round_up(int a, int b) { return (a + b - 1) / b * b; }

// chunk width in channel octets for a reduction of `cin` channels: a k-step holds 4 (tap, octet) pairs, so 4 octets make
// every step one tap; 5x5 / 7x7 keep 2 (LDS: the tile + halo of 4 octets would not leave room for two workgroups per CU)
__host__ __device__ inline int ck_qc(int cin, int ks) {
  const int q = (cin + 7) / 8;
  if (q >= 4 && ks == 3) return 4;
  return q >= 2 ? 2 : 1;
}
__host__ __device__ inline int ck_steps(int ks, int qc) { return (ks * ks * qc + 3) / 4; }
__host__ __device__ inline int ck_chunks(int cin, int qc) { return ((cin + 7) / 8 + qc - 1) / qc; }

__device__ __forceinline__ unsigned cvt_pk_bf16(float lo, float hi) {
  unsigned r;
  asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
  return r;
}

// 8 floats -> three planes of 8 bf16 (element e in the low / high half of dword e/2)
__device__ __forceinline__ void split8(const float (&v)[8], u32x4& p1, u32x4& p2, u32x4& p3) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float a = v[2 * i], b = v[2 * i + 1];
    const unsigned h1 = cvt_pk_bf16(a, b);
    const float ra = a - __uint_as_float(h1 << 16), rb = b - __uint_as_float(h1 & 0xffff0000u);
    const unsigned h2 = cvt_pk_bf16(ra, rb);
    const float qa = ra - __uint_as_float(h2 << 16), qb = rb - __uint_as_float(h2 & 0xffff0000u);
    p1[i] = h1; p2[i] = h2; p3[i] = cvt_pk_bf16(qa, qb);
  }
}

// ---- filter packing ----------------------------------------------------------------------------------------------
// packed[t][co16][chunk c][step s][plane][lane][8 bf16]: the B fragment of MFMA k-step (c, s) for output channels
// 16*co16 .. +15; lane = 16 g + (co % 16) holds pair j = 4 s + g -> tap = j / QC, octet o = j % QC, channels 8 (c QC + o) + e.
// mode 0 (forward):        value = w[t][co][ci][tap]                 (w is [T][Co][Ci][K][K]; the conv reduces over Ci)
// mode 1 (data gradient):  value = w[t][ci'][co'][K*K-1-tap]         (the conv reduces over Co and produces Ci: co' = produced, ci' = reduced)
__device__ __forceinline__ void convk_pack_one(const float* __restrict__ w, u32x4* __restrict__ out, int T, int cin, int cout,
                                               int ks, int qc, int C, int S, int co16s, int mode, long long idx, long long total) {
  if (idx >= total) return;

  const int lane = (int)(idx & 63);
  long long frag = idx >> 6;
  const int s = (int)(frag % S); frag /= S;
  const int c = (int)(frag % C); frag /= C;
  const int co16 = (int)(frag % co16s);
  const int t = (int)(frag / co16s);
  const int taps = ks * ks;
  const int co = co16 * 16 + (lane & 15), j = 4 * s + (lane >> 4);
  const int tap = j / qc, o = j % qc;
  float v[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int ci = (c * qc + o) * 8 + e;
    float val = 0.f;
    if (tap < taps && co < cout && ci < cin)
      val = mode == 0 ? w[(((size_t)t * cout + co) * cin + ci) * taps + tap]
                      : w[(((size_t)t * cin + ci) * cout + co) * taps + (taps - 1 - tap)];
    v[e] = val;
  }
  u32x4 p1, p2, p3;
  split8(v, p1, p2, p3);
  u32x4* dst = out + ((idx >> 6) * 3) * 64 + lane;
  dst[0] = p1; dst[64] = p2; dst[128] = p3;
}


__global__ __launch_bounds__(256) void convk_pack(const float* __restrict__ w, u32x4* __restrict__ out, int T, int cin, int cout,
                                                  int ks, int qc, int C, int S, int co16s, int mode, long long total) {
  convk_pack_one(w, out, T, cin, cout, ks, qc, C, S, co16s, mode, (long long)blockIdx.x * 256 + threadIdx.x, total);
}

// The filters of MANY layers in one launch (a training step packs every layer's fast weights after each inner update: 270 launches
// of ~9 us per SepConv meta-iteration, 760 per CAIN one).  Jobs ride in the kernel argument block; a workgroup finds its job by its
// first block (binary search, as csrc/mt_update.hip).
constexpr int CK_PACK_JOBS = 56;
struct PackJob {
  const float* w;
  u32x4* out;
  long long total;
  int T, cin, cout, ks, qc, C, S, co16s, mode, first_block;
};
struct PackTable {
  PackJob job[CK_PACK_JOBS];
  int n;
};
static_assert(sizeof(PackTable) <= 4096 - 64, "kernel argument block must stay under 4 KiB");

__global__ __launch_bounds__(256) void convk_pack_multi(const PackTable tb) {
  int lo = 0, hi = tb.n - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (tb.job[mid].first_block <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
  }
  const PackJob& j = tb.job[lo];
  convk_pack_one(j.w, j.out, j.T, j.cin, j.cout, j.ks, j.qc, j.C, j.S, j.co16s, j.mode,
                 (long long)((int)blockIdx.x - j.first_block) * 256 + threadIdx.x, j.total);
}

// ---- convolution ---------------------------------------------------------------------------------------------------
struct ConvkArgs {
  const float* x;      // [N][cin][H][W]
  const u32x4* wp;     // packed filter
  const float* bias;   // [T][cout] or null
  float* out;          // [N][cout][Ho][Wo]
  int N, T, cin, cout, H, W, Ho, Wo, pad;
  int tiles_x, tiles_y, CB, CBnt, C, co16s;     // CB: workgroup-level channel blocks, CBnt: blocks of 16*NT channels
  int total, per_xcd, order, cg;
  int reflect;         // 1: the border of width `pad` mirrors the image (nn.ReflectionPad2d) instead of reading zeros
  float slope;
  const float* mask;   // data gradient only, or null: out *= (mask > 0 ? 1 : mask_slope), mask laid out like out -- the (leaky) ReLU
  float mask_slope;    // derivative of the layer that PRODUCED this convolution's input, folded into this layer's data gradient
};

template <int KS, int QC, int NT, int TW>
struct ConvkGeom {
  static constexpr int TH = 256 / TW;
  static constexpr int ROWS = TH + KS - 1, COLS = TW + KS - 1, POS = ROWS * COLS;
  static constexpr int OCT = ck_round_up(POS * 16, 256);      // bytes per channel octet (multiple of 256: the four k-groups of a read hit disjoint banks)
  static constexpr int PLANE = QC * OCT;
  static constexpr int LDS = 3 * PLANE;
  static constexpr int TAPS = KS * KS;
  static constexpr int S = (TAPS * QC + 3) / 4;
  static constexpr int ITEMS = POS * QC;                       // (position, octet) items of a chunk
  static constexpr int ipt(int threads) { return (ITEMS + threads - 1) / threads; }
  static constexpr int MPR = TW / 16;                          // M-tiles per tile row
};

// P2 ("precise"): the five cross terms a1b2 .. a2b2 of a tile accumulate in their OWN registers and meet the a1b1 sum in the
// epilogue.  One accumulator takes six roundings per k-step at the magnitude of the running sum (error of an fp32 fmaf chain);
// split, the large sum takes one (of exact 16-bit-mantissa products) and the small one rounds 2^-8 lower: 3x closer to
// float64 (tools/bf16_split_probe.hip "2 accumulators"), better than a blocked fp32 CPU convolution.  For networks that
// amplify conv rounding (VoxelFlow's flow-to-pixel map: ~1000x into its gradients); costs the registers of half the tile.
//
// CG = 2 ("paired" workgroups, 8 waves): waves 0-3 and 4-7 compute the SAME pixel tile for two neighbouring blocks of 16*NT output
// channels.  The input tile is staged once for both (the split costs ~0.35 VALU per MFMA at CG = 1 -- and every VALU instruction
// is paid in matrix-pipe time, PMC: 65 % MFMA + 26 % VALU busy -- half of that at CG = 2), and with one workgroup per CU there is
// room for TWO input buffers: the next chunk is split and written to the other buffer in the middle of this chunk's MFMAs
// (one barrier per chunk, no staging bubble).
// MASK: the data gradient's epilogue multiplies by the producer's activation derivative (a.mask).  A template parameter, not a run-time
// branch: with the mask loads behind `if (a.mask)` between the epilogue's stores every variant of the kernel -- the forward included --
// lost 6-16 % (64 -> 64 @ 137 x 233: 116 -> 135 us; vmcnt counts loads and stores in one in-order counter, csrc/winograd.hip).
template <int KS, int QC, int NT, int TW, bool P2, int CG, bool MASK>
__global__ __launch_bounds__(CK_THREADS * CG, 2) void convk_kernel(const ConvkArgs a) {
  using G = ConvkGeom<KS, QC, NT, TW>;
  constexpr int THREADS = CK_THREADS * CG, IPT = G::ipt(THREADS);
  constexpr bool DB = CG == 2;              // double-buffered input tile
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, px = lane & 15, g = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), wv = wave & 3, cg = wave >> 2;      // pixel quadrant, channel group

  // XCD-contiguous block order: hardware block b runs on XCD b % 8; give each XCD a contiguous range of logical blocks so
  // that the blocks sharing an input tile (or a filter block) meet in ONE L2
  const int logical = (int)(blockIdx.x & 7) * a.per_xcd + (int)(blockIdx.x >> 3);
  if (logical >= a.total) return;
  int cbw, tile;                             // workgroup-level channel block (CG blocks of 16*NT channels), pixel tile
  if (a.order == 0) { cbw = logical % a.CB; tile = logical / a.CB; }
  else { const int ntiles = a.total / a.CB; tile = logical % ntiles; cbw = logical / ntiles; }
  const int cb = cbw * CG + cg;
  const bool active = cb < a.CBnt;           // an odd block count leaves the last workgroup's second wave group without work
  const int tx = tile % a.tiles_x;
  const int ty = (tile / a.tiles_x) % a.tiles_y;
  const int n = tile / (a.tiles_x * a.tiles_y);
  const int t = n % a.T;
  const int y0 = ty * G::TH, x0 = tx * TW;

  const size_t plane_in = (size_t)a.H * a.W;
  const float* __restrict__ xin = a.x + (size_t)n * a.cin * plane_in;
  // raw buffer over this sample: an item outside the image gets an offset beyond num_records and reads as 0 (no branches)
  const i32x4 xrs = ck_rsrc(xin, (unsigned)((size_t)a.cin * plane_in * 4));
  const int plane_bytes = (int)(plane_in * 4);

  // staging geometry of this thread's items (fixed over the chunks): LDS byte offset, buffer offset (octet included)
  int s_lds[IPT], s_voff[IPT], s_oct[IPT];
#pragma unroll
  for (int k = 0; k < IPT; ++k) {
    const int item = tid + THREADS * k;
    const int o = item / G::POS, pos = item - o * G::POS;
    const int r = pos / G::COLS, c = pos - r * G::COLS;
    int iy = y0 - a.pad + r, ix = x0 - a.pad + c;
    if (a.reflect) {      // mirrored border (pad <= H - 1, W - 1, host-checked); positions past the mirrored band feed no stored output
      iy = iy < 0 ? -iy : (iy >= a.H ? 2 * a.H - 2 - iy : iy);
      ix = ix < 0 ? -ix : (ix >= a.W ? 2 * a.W - 2 - ix : ix);
    }
    const bool ok = item < G::ITEMS && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
    s_voff[k] = ok ? (o * 8 * (int)plane_in + iy * a.W + ix) * 4 : 0x7fffffff;
    s_oct[k] = o;
    s_lds[k] = item < G::ITEMS ? o * G::OCT + pos * 16 : -1;
  }

  float stage[IPT][8];
  auto stage_load = [&](int chunk) {
    const int soff = chunk * QC * 8 * plane_bytes;
    if ((chunk + 1) * QC * 8 <= a.cin) {       // every channel of the chunk exists (wave-uniform)
#pragma unroll
      for (int k = 0; k < IPT; ++k)
#pragma unroll
        for (int e = 0; e < 8; ++e) stage[k][e] = ck_raw_buffer_load_f32(xrs, s_voff[k], soff + e * plane_bytes, 0);
    } else {                                   // the channel tail: the scalar offset is not range checked, test per element
#pragma unroll
      for (int k = 0; k < IPT; ++k)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const bool have = (chunk * QC + s_oct[k]) * 8 + e < a.cin;
          stage[k][e] = ck_raw_buffer_load_f32(xrs, have ? s_voff[k] : 0x7fffffff, have ? soff + e * plane_bytes : 0, 0);
        }
    }
  };
  auto stage_write = [&](int buf) {
    char* dst = smem + (DB ? buf * G::LDS : 0);
#pragma unroll
    for (int k = 0; k < IPT; ++k) {
      if (s_lds[k] < 0) continue;
      u32x4 p1, p2, p3;
      split8(stage[k], p1, p2, p3);
      *reinterpret_cast<u32x4*>(dst + s_lds[k]) = p1;
      *reinterpret_cast<u32x4*>(dst + G::PLANE + s_lds[k]) = p2;
      *reinterpret_cast<u32x4*>(dst + 2 * G::PLANE + s_lds[k]) = p3;
    }
  };

  // A-fragment base addresses of the wave's four M-tiles (pixel px of tile i), without tap / octet
  int abase[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = wv * (4 / G::MPR) + i / G::MPR, col = 16 * (i % G::MPR) + px;
    abase[i] = (row * G::COLS + col) * 16;
  }
  // B fragments: wp + ((((t co16s + cb NT + nn) C + c) S + s) 3 + plane) 64 + lane
  // raw buffer over this task's packed filter: per-lane offset lane * 16 once, everything else in the scalar offset (no
  // 64-bit address arithmetic in VGPRs between the MFMAs)
  const size_t wtask = (size_t)a.co16s * a.C * G::S * 3 * 1024;                 // bytes of one task's packed filter (< 2^31, host-checked)
  const i32x4 wrs = ck_rsrc(reinterpret_cast<const char*>(a.wp) + (size_t)t * wtask, (unsigned)wtask);
  const int wtile = a.C * G::S * 3 * 1024;                                      // bytes between 16-channel blocks
  const int wblock = cb * NT * wtile;                                           // this workgroup's first 16-channel block
  const int wlane = lane * 16;

  f32x4 acc[4][NT], lo[P2 ? 4 : 1][P2 ? NT : 1];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int nn = 0; nn < NT; ++nn) {
      acc[i][nn] = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (P2) lo[P2 ? i : 0][P2 ? nn : 0] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }

  // Software pipeline of a chunk.  A "group" = one k-step x one N-tile = 24 MFMAs; weights ride a ring of three fragment
  // sets fetched two groups ahead (global / L2 latency), the A fragments of the next k-step are re-read from LDS into the
  // registers of an M-tile pair right behind that pair's last MFMAs of the step.
  constexpr int GT = G::S * NT;
  constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};   // six products, small terms first
  bf16x8 bq[3][3], aq[4][3];
  auto load_b = [&](bf16x8 (&dst)[3], int wc, int grp) {
    const int s = grp / NT, nn = grp % NT;
#pragma unroll
    for (int p = 0; p < 3; ++p) dst[p] = __builtin_bit_cast(bf16x8, ck_raw_buffer_load_x4(wrs, wlane, wc + nn * wtile + (s * 3 + p) * 1024, 0));
  };
  auto load_a = [&](int i, int s, int buf) {
    const int j = 4 * s + g;
    int tap = j / QC;
    const int o = j % QC;
    tap = tap < G::TAPS ? tap : G::TAPS - 1;     // padding slots of the last step: any valid address (their weights are zero)
    const int ky = tap / KS, kx = tap - ky * KS;
    const int aoff = o * G::OCT + (ky * G::COLS + kx) * 16;
#pragma unroll
    for (int p = 0; p < 3; ++p) aq[i][p] = *reinterpret_cast<const bf16x8*>(smem + (DB ? buf * G::LDS : 0) + p * G::PLANE + abase[i] + aoff);
  };

  stage_load(0);
  if (DB) {
    stage_write(0);
    __syncthreads();
  }
  for (int c = 0; c < a.C; ++c) {
    const int wc = wblock + c * (G::S * 3 * 1024);
    const int buf = c & 1;
    if (active) {
      load_b(bq[0], wc, 0);                       // in flight across the staging bubble
      if (GT > 1) load_b(bq[1], wc, 1);
    }
    if (!DB) {
      stage_write(0);
      __syncthreads();
    }
    if (c + 1 < a.C) stage_load(c + 1);          // in flight while this chunk's MFMAs issue
    if (active) {
#pragma unroll
      for (int i = 0; i < 4; ++i) load_a(i, 0, buf);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int s = 0; s < G::S; ++s) {
#pragma unroll
      for (int nn = 0; nn < NT; ++nn) {
        const int grp = s * NT + nn;
        if (DB && s == G::S / 2 && nn == 0 && c + 1 < a.C) {
          // paired workgroups: the next chunk goes to the OTHER buffer now (its loads were issued a half chunk ago; every
          // wave left that buffer at the barrier that ended the previous chunk)
          stage_write(buf ^ 1);
          __builtin_amdgcn_sched_barrier(0);
        }
        if (active) {
          if (grp + 2 < GT) load_b(bq[(grp + 2) % 3], wc, grp + 2);
          const bool next_a = nn == NT - 1 && s + 1 < G::S;
#pragma unroll
          for (int h = 0; h < 2; ++h) {
#pragma unroll
            for (int q = 0; q < 6; ++q)
#pragma unroll
              for (int i = 2 * h; i < 2 * h + 2; ++i) {
                if (P2 && q < 5)
                  lo[P2 ? i : 0][P2 ? nn : 0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(aq[i][PA[q]], bq[grp % 3][PB[q]], lo[P2 ? i : 0][P2 ? nn : 0], 0, 0, 0);
                else
                  acc[i][nn] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(aq[i][PA[q]], bq[grp % 3][PB[q]], acc[i][nn], 0, 0, 0);
              }
            if (next_a) { load_a(2 * h, s + 1, buf); load_a(2 * h + 1, s + 1, buf); }
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    __syncthreads();
  }
  if (!active) return;

  // epilogue: lane holds pixels 4 g .. 4 g + 3 of M-tile i (rows of D) for channel 16 nn + px (column of D)
  const size_t plane_out = (size_t)a.Ho * a.Wo;
  float* __restrict__ outn = a.out + (size_t)n * a.cout * plane_out;
  const bool vec_ok = (a.Wo & 3) == 0;
#pragma unroll
  for (int nn = 0; nn < NT; ++nn) {
    const int co = (cb * NT + nn) * 16 + px;
    if (co >= a.cout) continue;
    const float b = a.bias ? a.bias[(size_t)t * a.cout + co] : 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int oy = y0 + wv * (4 / G::MPR) + i / G::MPR;
      const int ox = x0 + 16 * (i % G::MPR) + 4 * g;
      if (oy >= a.Ho || ox >= a.Wo) continue;
      f32x4 v = acc[i][nn];
      if (P2) v += lo[P2 ? i : 0][P2 ? nn : 0];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float u = v[r] + b;
        v[r] = u > 0.f ? u : u * a.slope;
      }
      const size_t oidx = (size_t)co * plane_out + (size_t)oy * a.Wo + ox;
      float* dst = outn + oidx;
      if (vec_ok && ox + 3 < a.Wo) {
        if (MASK) {
          const f32x4 m = *reinterpret_cast<const f32x4*>(a.mask + (size_t)n * a.cout * plane_out + oidx);
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = m[r] > 0.f ? v[r] : v[r] * a.mask_slope;
        }
        *reinterpret_cast<f32x4*>(dst) = v;
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (ox + r < a.Wo) {
            float u = v[r];
            if (MASK) u = a.mask[(size_t)n * a.cout * plane_out + oidx + r] > 0.f ? u : u * a.mask_slope;
            dst[r] = u;
          }
      }
    }
  }
}

template <int KS, int QC, int NT, int TW, bool P2, int CG, bool MASK = false>
int launch_convk_cg(const ConvkArgs& a, hipStream_t stream) {
  using G = ConvkGeom<KS, QC, NT, TW>;
  if constexpr (!MASK) {
    if (a.mask) {           // masked epilogue: the 3 x 3 kernels of the conv -> act -> conv chains only (hip_ops masks the others itself)
      if constexpr (KS == 3 && !P2) return launch_convk_cg<KS, QC, NT, TW, P2, CG, true>(a, stream);
      else return SAVFI_E_UNSUPPORTED;
    }
  }
  static uint32_t configured = 0;
  auto kern = convk_kernel<KS, QC, NT, TW, P2, CG, MASK>;
  constexpr int lds = G::LDS * CG;          // CG = 2: two input buffers
  static_assert(lds <= 160 * 1024, "input tile buffers exceed the LDS of a CU");
  if (lds > 64 * 1024) {
    const int rc = savfi_ensure_dynamic_lds(reinterpret_cast<const void*>(kern), lds, configured);
    if (rc != SAVFI_OK) return rc;
  }
  hipLaunchKernelGGL(kern, dim3(a.per_xcd * 8), dim3(CK_THREADS * CG), lds, stream, a);
  return savfi_launch_status();
}

template <int KS, int QC, int NT, int TW, bool P2 = false>
int launch_convk(const ConvkArgs& a, hipStream_t stream) {
  if constexpr (NT >= 2) {
    if (a.cg == 2) return launch_convk_cg<KS, QC, NT, TW, P2, 2>(a, stream);
  }
  return launch_convk_cg<KS, QC, NT, TW, P2, 1>(a, stream);
}

template <int KS, int QC>
int dispatch_nt_tw(const ConvkArgs& a, int nt, int tw, bool precise, hipStream_t stream) {
  if (precise) {          // two accumulator sets: at most 32 output channels per wave
    if (tw == 32) return nt >= 2 ? launch_convk<KS, QC, 2, 32, true>(a, stream) : launch_convk<KS, QC, 1, 32, true>(a, stream);
    return nt >= 2 ? launch_convk<KS, QC, 2, 16, true>(a, stream) : launch_convk<KS, QC, 1, 16, true>(a, stream);
  }
  if (tw == 32) {
    if (nt == 4) return launch_convk<KS, QC, 4, 32>(a, stream);
    if (nt == 2) return launch_convk<KS, QC, 2, 32>(a, stream);
    return launch_convk<KS, QC, 1, 32>(a, stream);
  }
  if (nt == 4) return launch_convk<KS, QC, 4, 16>(a, stream);
  if (nt == 2) return launch_convk<KS, QC, 2, 16>(a, stream);
  return launch_convk<KS, QC, 1, 16>(a, stream);
}

int dispatch_convk(const ConvkArgs& a, int ks, int qc, int nt, int tw, bool precise, hipStream_t stream) {
  if (ks == 3) {
    if (qc == 4) return dispatch_nt_tw<3, 4>(a, nt, tw, precise, stream);
    if (qc == 2) return dispatch_nt_tw<3, 2>(a, nt, tw, precise, stream);
    return dispatch_nt_tw<3, 1>(a, nt, tw, precise, stream);
  }
  if (ks == 5) {
    if (qc == 2) return dispatch_nt_tw<5, 2>(a, nt, tw, precise, stream);
    return dispatch_nt_tw<5, 1>(a, nt, tw, precise, stream);
  }
  if (qc == 2) return dispatch_nt_tw<7, 2>(a, nt, tw, precise, stream);
  return dispatch_nt_tw<7, 1>(a, nt, tw, precise, stream);
}

inline bool ck_supported_k(int k) { return k == 3 || k == 5 || k == 7; }

}  // namespace

extern "C" int64_t savfi_convk_filter_floats(int T, int Ci, int Co, int K, int mode) {
  if (T <= 0 || Ci <= 0 || Co <= 0) return SAVFI_E_SHAPE;
  if (!ck_supported_k(K) || (mode != 0 && mode != 1)) return SAVFI_E_UNSUPPORTED;
  const int cin = mode == 0 ? Ci : Co, cout = mode == 0 ? Co : Ci;
  const int qc = ck_qc(cin, K);
  return (int64_t)T * ck_round_up((cout + 15) / 16, 4) * ck_chunks(cin, qc) * ck_steps(K, qc) * 3 * 64 * 4;   // 4-byte units
}

extern "C" int savfi_convk_filters_f32(const float* w, float* p_fwd, float* p_bwd, int T, int Ci, int Co, int K, void* stream) {
  if (!w || (!p_fwd && !p_bwd)) return SAVFI_E_NULL;
  if (T <= 0 || Ci <= 0 || Co <= 0) return SAVFI_E_SHAPE;
  if (!ck_supported_k(K)) return SAVFI_E_UNSUPPORTED;
  for (int mode = 0; mode < 2; ++mode) {
    float* dst = mode == 0 ? p_fwd : p_bwd;
    if (!dst) continue;
    const int cin = mode == 0 ? Ci : Co, cout = mode == 0 ? Co : Ci;
    const int qc = ck_qc(cin, K), C = ck_chunks(cin, qc), S = ck_steps(K, qc), co16s = ck_round_up((cout + 15) / 16, 4);
    const long long total = (long long)T * co16s * C * S * 64;
    hipLaunchKernelGGL(convk_pack, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w,
                       reinterpret_cast<u32x4*>(dst), T, cin, cout, K, qc, C, S, co16s, mode, total);
    const int rc = savfi_launch_status();
    if (rc != SAVFI_OK) return rc;
  }
  return SAVFI_OK;
}

extern "C" int savfi_convk_filters_multi_f32(const float* const* w, float* const* p_fwd, float* const* p_bwd, const int* T, const int* Ci,
                                             const int* Co, const int* K, int n, void* stream) {
  if (!w || !p_fwd || !p_bwd || !T || !Ci || !Co || !K) return SAVFI_E_NULL;
  if (n <= 0) return SAVFI_E_SHAPE;
  PackTable tb;
  tb.n = 0;
  int blocks = 0;
  auto flush = [&]() {
    if (tb.n == 0) return (int)SAVFI_OK;
    hipLaunchKernelGGL(convk_pack_multi, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, tb);
    tb.n = 0;
    blocks = 0;
    return savfi_launch_status();
  };
  for (int i = 0; i < n; ++i) {
    if (!w[i] || (!p_fwd[i] && !p_bwd[i])) return SAVFI_E_NULL;
    if (T[i] <= 0 || Ci[i] <= 0 || Co[i] <= 0) return SAVFI_E_SHAPE;
    if (!ck_supported_k(K[i])) return SAVFI_E_UNSUPPORTED;
    for (int mode = 0; mode < 2; ++mode) {
      float* dst = mode == 0 ? p_fwd[i] : p_bwd[i];
      if (!dst) continue;
      const int cin = mode == 0 ? Ci[i] : Co[i], cout = mode == 0 ? Co[i] : Ci[i];
      PackJob& j = tb.job[tb.n];
      j.w = w[i]; j.out = reinterpret_cast<u32x4*>(dst);
      j.T = T[i]; j.cin = cin; j.cout = cout; j.ks = K[i]; j.mode = mode;
      j.qc = ck_qc(cin, K[i]); j.C = ck_chunks(cin, j.qc); j.S = ck_steps(K[i], j.qc); j.co16s = ck_round_up((cout + 15) / 16, 4);
      j.total = (long long)j.T * j.co16s * j.C * j.S * 64;
      j.first_block = blocks;
      const long long nb = (j.total + 255) / 256;
      if (nb > (1 << 24)) return SAVFI_E_TOOBIG;
      blocks += (int)nb;
      if (++tb.n == CK_PACK_JOBS) {
        const int rc = flush();
        if (rc != SAVFI_OK) return rc;
      }
    }
  }
  return flush();
}

extern "C" int savfi_convk_tasks_pre_f32(const float* x, const float* packed, const float* bias, float* out, int N, int T, int Ci,
                                         int Co, int H, int W, int K, int pad, int mode, float slope, int precise, void* stream) {
  return savfi_convk_tasks_pre_reflect_f32(x, packed, bias, out, N, T, Ci, Co, H, W, K, pad, mode, slope, precise, 0, stream);
}

static int convk_tasks_pre_impl(const float* x, const float* packed, const float* bias, float* out, int N, int T, int Ci, int Co, int H, int W,
                                int K, int pad, int mode, float slope, int precise, int reflect, const float* mask, float mask_slope,
                                void* stream);

extern "C" int savfi_convk_tasks_pre_reflect_f32(const float* x, const float* packed, const float* bias, float* out, int N, int T, int Ci,
                                                 int Co, int H, int W, int K, int pad, int mode, float slope, int precise, int reflect,
                                                 void* stream) {
  return convk_tasks_pre_impl(x, packed, bias, out, N, T, Ci, Co, H, W, K, pad, mode, slope, precise, reflect, nullptr, 1.f, stream);
}

// data gradient (mode 1) with the activation derivative of the layer that produced this convolution's input folded into the epilogue:
// gx = dgrad(gy) * (mask > 0 ? 1 : mask_slope), mask [N,Ci,H+K-1-2p,W+K-1-2p] = this convolution's forward input (include/savfi_hip.h)
extern "C" int savfi_convk_dgrad_masked_f32(const float* gy, const float* packed, const float* mask, float mask_slope, float* gx, int N, int T,
                                            int Ci, int Co, int H, int W, int K, int pad, int precise, void* stream) {
  if (!mask) return SAVFI_E_NULL;
  return convk_tasks_pre_impl(gy, packed, nullptr, gx, N, T, Ci, Co, H, W, K, pad, 1, 1.f, precise, 0, mask, mask_slope, stream);
}

static int convk_tasks_pre_impl(const float* x, const float* packed, const float* bias, float* out, int N, int T, int Ci, int Co, int H, int W,
                                int K, int pad, int mode, float slope, int precise, int reflect, const float* mask, float mask_slope,
                                void* stream) {
  if (reflect && (mode != 0 || pad >= H || pad >= W)) return SAVFI_E_UNSUPPORTED;
  if (!x || !packed || !out) return SAVFI_E_NULL;
  if (N <= 0 || T <= 0 || Ci <= 0 || Co <= 0 || H <= 0 || W <= 0 || N % T != 0 || pad < 0 || pad > K - 1) return SAVFI_E_SHAPE;
  if (!ck_supported_k(K) || (mode != 0 && mode != 1)) return SAVFI_E_UNSUPPORTED;
  ConvkArgs a;
  a.x = x; a.wp = reinterpret_cast<const u32x4*>(packed); a.bias = bias; a.out = out;
  a.N = N; a.T = T; a.H = H; a.W = W;
  a.cin = mode == 0 ? Ci : Co; a.cout = mode == 0 ? Co : Ci;
  a.pad = mode == 0 ? pad : K - 1 - pad;
  a.Ho = H + 2 * a.pad - K + 1; a.Wo = W + 2 * a.pad - K + 1;
  if (a.Ho <= 0 || a.Wo <= 0) return SAVFI_E_SHAPE;
  if ((int64_t)a.cin * H * W >= (1ll << 29) || (int64_t)a.cout * a.Ho * a.Wo >= (1ll << 31)) return SAVFI_E_TOOBIG;   // byte offsets of a sample fit 31 bits
  a.slope = slope;
  a.mask = mask; a.mask_slope = mask_slope;
  a.reflect = reflect ? 1 : 0;
  const int qc = ck_qc(a.cin, K);
  a.C = ck_chunks(a.cin, qc);
  if ((int64_t)ck_round_up((a.cout + 15) / 16, 4) * a.C * ck_steps(K, qc) * 3 * 1024 >= (1ll << 31)) return SAVFI_E_TOOBIG;
  a.co16s = ck_round_up((a.cout + 15) / 16, 4);       // packed filter blocks (zero padded to a multiple of 4)
  // tile shape: the better-filled of 8 x 32 and 16 x 16
  auto fill = [&](int th, int tw) { return (double)a.Ho * a.Wo / ((double)((a.Ho + th - 1) / th * th) * ((a.Wo + tw - 1) / tw * tw)); };
  // 8 x 32 tiles have the smaller halo (340 vs 324 staged cells but 128-byte output rows and half the row count): 16 x 16 only
  // where it fills clearly better
  int tw = fill(16, 16) > 1.08 * fill(8, 32) ? 16 : 32;
#ifdef SAVFI_CONVK_TW          // variant builds: 16 or 32
  tw = SAVFI_CONVK_TW;
#endif
  const int th = 256 / tw;
  a.tiles_x = (a.Wo + tw - 1) / tw; a.tiles_y = (a.Ho + th - 1) / th;
  const int64_t tiles = (int64_t)N * a.tiles_x * a.tiles_y;
  // Output channels per wave (16 nt) and wave groups per workgroup (cg), from tools/convk_bench.py sweeps with SAVFI_CONVK_TILE
  // (profiles/r03_convk_tile_sweep.txt).  W1 = single workgroups of 64 channels (two share a CU), W2 = paired workgroups (8 waves,
  // one per CU: the input tile is split once for both channel blocks and double buffered).  Pairs of 64-channel blocks win
  // when they fill the GPU in exactly one round; otherwise single 64-channel workgroups while they oversubscribe the CUs; deep
  // layers on small maps take pairs of 32-channel blocks (twice the workgroups); the rest single 32-channel workgroups.
  const int co16 = (a.cout + 15) / 16, co64 = (co16 + 3) / 4, co32 = (co16 + 1) / 2;
  const int64_t W1 = tiles * co64, W2_4 = tiles * ((co64 + 1) / 2), W2_2 = tiles * ((co32 + 1) / 2);
  const bool ragged64 = a.cout % 64 != 0 && a.cout % 64 <= 32 && a.cout < 128;      // a half-empty 64-channel block
  int nt, cg;
  if (a.cout <= 16) { nt = 1; cg = 1; }
  else if (!precise && co16 > 2 && co64 >= 2 && W2_4 >= 176 && W2_4 <= 256) { nt = 4; cg = 2; }
  else if (!precise && co16 > 2 && !ragged64 && W1 >= 448) { nt = 4; cg = 1; }
  else if (co32 >= 2 && W2_2 >= 160) { nt = 2; cg = 2; }
  else { nt = 2; cg = 1; }
#if defined(SAVFI_CONVK_TILE_NT) && defined(SAVFI_CONVK_TILE_CG)      // variant builds (tools/convk_tile_sweep.py): nt in {1, 2, 4}, cg in {1, 2}
  if (!(precise && SAVFI_CONVK_TILE_NT == 4) && !(SAVFI_CONVK_TILE_NT == 1 && SAVFI_CONVK_TILE_CG == 2)) { nt = SAVFI_CONVK_TILE_NT; cg = SAVFI_CONVK_TILE_CG; }
#endif
  a.CBnt = (co16 + nt - 1) / nt;
  a.cg = cg;
  a.CB = (a.CBnt + cg - 1) / cg;
  if (tiles * a.CB >= (1ll << 30)) return SAVFI_E_TOOBIG;
  a.total = (int)(tiles * a.CB);
  a.per_xcd = (a.total + 7) / 8;
  // blocks that share an input tile next to each other, unless the filter is the larger object
  const int64_t w_bytes = (int64_t)a.co16s * 16 * a.cin * K * K * 6, in_bytes = (int64_t)a.cin * H * W * 4 * (N / T);
  a.order = w_bytes > in_bytes ? 1 : 0;
  return dispatch_convk(a, K, qc, nt, tw, precise != 0, (hipStream_t)stream);
}
