// 3x3 / stride 1 convolution (forward and data gradient, zero padding 0 or 1) for gfx950: Winograd F(2x2, 3x3) with
// its 16 batched GEMMs on the exact-fp32 matrix cores (v_mfma_f32_16x16x4_f32), fused in ONE kernel.
//
// Why: after the sepconv op, the SepConv / CAIN / VoxelFlow backbones are 3x3 convolutions (SURVEY.md 8(a) rows
// 13, 15, 17: sepconv/model.py:172-245, cain/model.py + model_utils.py:957-1053, voxel_flow.py:357-470); in the
// round-1 profile MIOpen's fp32 Winograd kernel is ~50 % of the inner step at 80-100 direct-equivalent TFLOP/s.
// Winograd cuts the multiplies 2.25x; on the MFMA pipe one LDS dword feeds 1024 FMAs, so the multiplies can run
// close to the 157 TFLOP/s fp32 peak (~350 direct-equivalent) instead of being operand-bound.
//
//   Y = A^T [ (G g G^T) (.) (B^T d B) ] A          per 2x2 output tile, d = its 4x4 input patch
//   U[xi][k][i] = (G g G^T)[xi]                     filter transform, a tiny pre-pass (wino_filter_transform)
//   V[xi][k][t] = (B^T d B)[xi]                     input transform, in registers, one (tile, channel) per thread
//   M[xi][i][t] = sum_k U[xi][k][i] V[xi][k][t]     16 GEMMs -> MFMA 16x16x4 f32        (k: reduction channel, i: produced)
//
// Workgroup = 256 threads = 4 waves, TWO workgroups per CU (each ~210 VGPRs, 40 KB LDS) so that one's prologue /
// output stage overlaps the other's channel loop.  A workgroup owns 64 tiles (4 x 16 tiles = 8 x 32 output pixels)
// x 32 produced channels and walks the reduction channels 4 at a time (one MFMA k-step).  Wave w owns the Winograd
// row xi = 4w .. 4w+3 for all 32 channels x 64 tiles: 4 x 2 x 4 accumulator tiles = 128 registers.
// Per chunk every thread transforms ONE 4x4 patch (tile = lane, channel = wave) in registers and writes its 16 values
// to the V buffer in LDS (double buffered, one barrier per chunk); a wave reads 16 B fragments from LDS and holds its
// A fragments (straight from global / L2, pre-swizzled so that a lane gets them with two 16-byte loads) for 32 MFMAs.
// The matrix pipe of a SIMD executes nothing else while a VALU / LDS / memory instruction of ANY of its waves issues
// (micro-benchmark, round 2: 32 MFMAs + 32 VALU take 1028 + ~5 x 32 cycles, also when they come from two waves), so the
// loop is kept at ~40 VALU per 32 MFMAs; patches and A fragments are prefetched TWO chunks ahead into the registers
// their predecessors just vacated.  Output stage: the wave reduces its row in registers (column half of A^T M A), the
// four rows meet in LDS 16 channels at a time, + bias, (leaky) ReLU, raw buffer stores.
//
// Memory instructions: vmcnt counts loads AND stores in one in-order counter.  Every load whose value is needed after a
// store has been issued is therefore placed BEFORE the stores of that stretch (bias: prologue), and the stores are
// branch-free (out-of-range offsets instead of per-lane branches) so that the compiler can count them: a conservative
// vmcnt(0) in front of a load's first use waits for every outstanding store's write acknowledge (~0.8 us each time).
// tools/wino_trace.py (per-workgroup timestamps, -DWINO_TRACE build) is how these stalls were found.
//
// The data gradient is the same kernel on the flipped / transposed filter (wino_filter_transform mode 1).
#include "common.h"
#include <type_traits>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
// 16-byte raw buffer load (the clang builtin of this release narrows the b128 form to one dword)
__device__ f32x4 savfi_raw_buffer_load_x4(i32x4 rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.load.v4f32");
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ void savfi_raw_buffer_store_x2(f32x2 data, i32x4 rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.store.v2f32");
__device__ f32x2 savfi_raw_buffer_load_x2(i32x4 rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.load.v2f32");
__device__ float savfi_raw_buffer_load_x1(i32x4 rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.load.f32");
__device__ void savfi_raw_buffer_store_x1(float data, i32x4 rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.store.f32");
__device__ void savfi_raw_buffer_store_x4(f32x4 data, i32x4 rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.store.v4f32");

namespace {

#include "winograd4.h"      // Winograd F(4x4, 3x3) for the layers of at most 512 -> 512 channels (namespace w4)

constexpr int WNT = 256;            // threads
// 64 tiles per workgroup, one per lane, as a 2^(6-s) x 2^s block of tiles (s = WinoArgs::tile_shift, picked per launch):
// 4 x 16 tiles (8 x 32 output pixels: wide rows, the default), 8 x 8 for small maps (a 12 x 16 map is 6 x 8 tiles: one 8 x 8
// block holds it at 75 % where 4 x 16 blocks reach 37.5 %), 2 x 32 / 16 x 4 for very flat / narrow ones.
constexpr int TILES_WG = 64;
constexpr int COB = 32;             // produced channels per workgroup
constexpr int CIB = 4;              // reduction channels per chunk (one MFMA k-step), one per wave
#ifndef WINO_VS
#define WINO_VS 80
#endif
constexpr int VS = WINO_VS;         // pitch of a V row [k] in floats: 64 tiles + 16 -> the 4 k-groups hit disjoint banks
constexpr int XS = 68;              // pitch of an exchange row [channel] (output stage)
constexpr int VBUF = 16 * CIB * VS; // floats per V buffer (20 KB)
constexpr int XBUF = 4 * 2 * 16 * XS;  // exchange: [row xi_r][column c][16 channels][XS]
constexpr int LDS_FLOATS = (2 * VBUF > XBUF) ? 2 * VBUF : XBUF;

// ---- filter transform ------------------------------------------------------------------------------------
// U = G g G^T for reduction channel k < KP and produced channel i < IP (zero padded), stored in MFMA A-fragment
// order so that the wave owning Winograd row r fetches a chunk's fragments with two aligned 16-byte loads per lane:
//   Uf[k / 4][i / 32][r][lane = (k % 4) * 16 + i % 16][c][(i % 32) / 16]        (xi = 4 r + c)
// mode 0 (forward):        g[a][b] = w[i][k][a][b]           (w is [Co][Ci][3][3]; i = co, k = ci)
// mode 1 (data gradient):  g[a][b] = w[k][i][2-a][2-b]       (i = ci, k = co)
// Filter set `task` (tasks adapted in lockstep own their weights): w + task * Co*Ci*9 -> U + task * 16*KP*IP.
__device__ __forceinline__ void filter_transform_block(const float* __restrict__ w, float* __restrict__ U, int Co, int Ci, int K,
                                                       int I, int KP, int IP, int mode, int bx, int by) {
  // One workgroup = 4 reduction channels (one MFMA k-step) x 64 produced channels: it fills two whole
  // [4 rows][64 lanes][8 floats] fragment blocks (16 KB) of U by itself -- full cache lines leave the CU, where a
  // k-major thread numbering made every 2 KB block the target of four workgroups' partial writes -- and its reads of a
  // forward filter (mode 0: w[i][k][3][3]) are runs of 4 channels = 144 contiguous bytes per lane quad.
  const int kk = threadIdx.x & 3, ii = threadIdx.x >> 2;
  const int k = 4 * bx + kk, i = 64 * (by % ((IP + 63) / 64)) + ii;
  const int task = by / ((IP + 63) / 64);
  if (i >= IP) return;
  w += (size_t)task * Co * Ci * 9;
  U += (size_t)task * 16 * KP * IP;
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
  const int lane = (k & 3) * 16 + (i & 15);
  float* base = U + ((((size_t)(k >> 2) * (IP / COB) + (i >> 5)) * 4) * 64 + lane) * 8 + ((i & 31) >> 4);
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    float* o = base + (size_t)r * 64 * 8;
    o[0] = t[r][0];
    o[2] = 0.5f * (t[r][0] + t[r][1] + t[r][2]);
    o[4] = 0.5f * (t[r][0] - t[r][1] + t[r][2]);
    o[6] = t[r][2];
  }
}

__global__ __launch_bounds__(256) void wino_filter_transform(const float* __restrict__ w, float* __restrict__ U,
                                                             int Co, int Ci, int K, int I, int KP, int IP, int mode, int f4) {
  if (f4) w4::filter_transform_block4(w, U, Co, Ci, K, I, KP, IP, mode, blockIdx.x, blockIdx.y);
  else filter_transform_block(w, U, Co, Ci, K, I, KP, IP, mode, blockIdx.x, blockIdx.y);
}

// Both transforms of a layer in ONE launch (blockIdx.z = mode): the forward pass of a training step knows that the data
// gradient will need the flipped / transposed filter, and a transform launch costs more than it computes (538 of them were
// 5 % of a SepConv meta-iteration).  The grid covers the larger of the two block ranges; surplus blocks leave at once.
__global__ __launch_bounds__(256) void wino_filter_transform_dual(const float* __restrict__ w, float* __restrict__ Uf,
                                                                  float* __restrict__ Ub, int Co, int Ci, int KPf, int IPf,
                                                                  int KPb, int IPb, int T, int f4) {
  const int mode = blockIdx.z;
  const int KP = mode == 0 ? KPf : KPb, IP = mode == 0 ? IPf : IPb;
  if ((int)blockIdx.x >= KP / 4 || (int)blockIdx.y >= ((IP + 63) / 64) * T) return;
  if (f4) w4::filter_transform_block4(w, mode == 0 ? Uf : Ub, Co, Ci, mode == 0 ? Ci : Co, mode == 0 ? Co : Ci, KP, IP, mode, blockIdx.x, blockIdx.y);
  else filter_transform_block(w, mode == 0 ? Uf : Ub, Co, Ci, mode == 0 ? Ci : Co, mode == 0 ? Co : Ci, KP, IP, mode, blockIdx.x, blockIdx.y);
}


// The transforms of MANY layers in one launch (see convk_pack_multi, csrc/convk.hip): job = (layer, mode); a workgroup's position
// inside its job gives the (bx, by) of wino_filter_transform.
constexpr int WINO_FT_JOBS = 56;
struct FtJob {
  const float* w;
  float* U;
  int Co, Ci, K, I, KP, IP, mode, nbx, first_block, f4;
};
struct FtTable {
  FtJob job[WINO_FT_JOBS];
  int n;
};
static_assert(sizeof(FtTable) <= 4096 - 64, "kernel argument block must stay under 4 KiB");

__global__ __launch_bounds__(256) void wino_filter_transform_multi(const FtTable tb) {
  int lo = 0, hi = tb.n - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (tb.job[mid].first_block <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
  }
  const FtJob& j = tb.job[lo];
  const int rel = (int)blockIdx.x - j.first_block;
  if (j.f4) w4::filter_transform_block4(j.w, j.U, j.Co, j.Ci, j.K, j.I, j.KP, j.IP, j.mode, rel % j.nbx, rel / j.nbx);
  else filter_transform_block(j.w, j.U, j.Co, j.Ci, j.K, j.I, j.KP, j.IP, j.mode, rel % j.nbx, rel / j.nbx);
}

// ---- fused convolution -------------------------------------------------------------------------------------
// x [N][K][H][W] -> out [N][I][Ho][Wo]; bias [I] or null; act: y = v > 0 ? v : slope * v (slope 1 = none)
struct WinoArgs {
  const float* x;
  const float* U;
  const float* bias;
  float* out;
  int K, I, KP, IP, H, W, Ho, Wo, off, tiles_y, tiles_x;   // input H x W, output Ho x Wo, patch origin 2t - off;
  float slope;                                             // tiles_* = number of tile blocks per image
  int tile_shift;                                          // log2 of the tile block's width in tiles (see TILES_WG)
  int nsplit, chunks_per_split;                            // reduction channels split over workgroups (deep layers)
  float* partial;                                          // nsplit > 1: raw partial outputs [split][N][I][Ho][Wo]
  int T;                                                   // filter sets: sample n uses set n % T (U [T][...], bias [T][I])
  int N;                                                   // samples
  const float* mask;                                       // or null: out *= (mask > 0 ? 1 : mask_slope), mask laid out like out (the
  float mask_slope;                                        // activation derivative of the layer that produced this data gradient's output)
  int out_unit16;                                          // 1: out is written [n][y][x / 16][channel][16] (see savfi_conv3x3_tasks_pre_unit16_f32)
#ifdef WINO_TRACE
  unsigned long long* trace;                               // [workgroup][8]: timestamps (100 MHz) + hardware ids
#endif
};

// The 4x4 patch of a thread sits at the same (y, x) for every channel: its four row offsets are computed once.  A row is
// ONE dword-aligned 16-byte RAW BUFFER load at the patch's own column, through a descriptor that spans exactly the
// channel plane (base and size in SGPRs, rebuilt per chunk with two scalar adds), and the hardware's per-dword range
// check does most of the zero padding:
//   * a row above / below the image gets an offset beyond the plane: the load returns zeros;
//   * a row that runs past the END of the plane (bottom-right tiles) returns zeros for the dwords beyond it;
//   * a row that crosses the left / right image border picks up the neighbouring row's pixels: those COLUMNS are zeroed
//     with one v_cndmask per element, the lane masks of the four columns sitting in SGPR pairs (one ballot each per
//     workgroup) and only for the columns that are partial somewhere in the wave: 4 VALU per chunk in a wave on the left
//     edge, none in an interior wave;
//   * a row that would start BEFORE the plane (image row 0 of the tile at x = 0: offset -4 or -8) is the one case the range
//     check gets wrong -- a negative offset zeroes the whole load (measured on gfx950) -- so that row is loaded from
//     offset 0 and moved up by `off` elements in lane 0 of the workgroup that owns tile (0, 0): 2-3 v_cndmask.
// The first version used global loads with clamped addresses, rebuilt each lane's predicate from a bit mask in a VGPR
// (~60 VALU per chunk in every border wave) and fixed the corner rows with per-lane divergent code: the two workgroups per
// image that hold a corner ran several times longer than the rest, and because the bottom-right one is the LAST block
// of a launch it was the tail of every launch -- 25 % of the kernel's time on 96 x 128 maps (190 -> 148 us, 128 -> 128).
// Reduction channels beyond K need no masking: their filter transform is zero and the patch is a finite duplicate of
// channel K - 1 at the same tile.

struct Patch {
  unsigned voff[4];   // row r: byte offset of the 16-byte window from the plane base (>= 2^31: a row outside the image)
};

struct LaneMasks {
  unsigned long long col[4];  // bit l of col[c]: column c of lane l's patch is inside the image
  int partial_cols;           // bit c: some lane of the wave has column c outside
};

__device__ __forceinline__ Patch make_patch(int y0, int x0, int H, int W) {
  Patch p;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int y = y0 + r;
    const int lin = y * W + x0;                    // < 0 only for y == 0, x0 < 0: loaded from 0, see fix_corner
    p.voff[r] = (y >= 0 && y < H) ? (unsigned)max(lin, 0) * 4u : 0x80000000u;
  }
  return p;
}

// Unit-major INPUT (savfi_conv3x3_tasks_pre_in_unit16_f32: a sample is [y][x / 16][channel][16]).  A 4-pixel patch row may straddle two
// units, so a row is TWO 8-byte loads (x0 is even and W a multiple of 16: a pair is whole inside one unit and whole inside or outside the
// image); rows and pairs outside the image get an offset beyond the buffer (zeros): no column masks, no corner fix-up.  The offsets are
// those of channel 0; a channel is 64 bytes further (it rides in the descriptor's base).
struct Patch16 {
  unsigned va[4], vb[4];
};
__device__ __forceinline__ Patch16 make_patch16(int y0, int x0, int H, int W, int K) {
  Patch16 p;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int y = y0 + r;
    const bool row = y >= 0 && y < H;
    const int xa = x0, xb = x0 + 2;
    p.va[r] = (row && xa >= 0 && xa < W) ? (unsigned)((y * (W >> 4) + (xa >> 4)) * K) * 64u + (unsigned)(xa & 15) * 4u : 0x80000000u;
    p.vb[r] = (row && xb >= 0 && xb < W) ? (unsigned)((y * (W >> 4) + (xb >> 4)) * K) * 64u + (unsigned)(xb & 15) * 4u : 0x80000000u;
  }
  return p;
}

__device__ __forceinline__ LaneMasks make_lane_masks(int x0, int W) {
  LaneMasks lm;
  lm.partial_cols = 0;
  const unsigned long long all = __builtin_amdgcn_ballot_w64(true);
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    lm.col[c] = __builtin_amdgcn_ballot_w64(x0 + c >= 0 && x0 + c < W);
    if (lm.col[c] != all) lm.partial_cols |= 1 << c;
  }
  return lm;
}

__device__ __forceinline__ i32x4 plane_rsrc(const float* plane, unsigned bytes) {
  const unsigned long long p = reinterpret_cast<unsigned long long>(plane);
  i32x4 r;
  r.x = (int)(unsigned)p;
  r.y = (int)(unsigned)(p >> 32);       // stride 0: raw buffer
  r.z = (int)bytes;
  r.w = 0x00020000;
  return r;
}

__device__ __forceinline__ void load_patch(float (&d)[16], const float* __restrict__ plane, unsigned plane_bytes, const Patch& p) {
  const i32x4 rs = plane_rsrc(plane, plane_bytes);
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const f32x4 v = savfi_raw_buffer_load_x4(rs, (int)p.voff[r], 0, 0);
    d[4 * r + 0] = v.x; d[4 * r + 1] = v.y; d[4 * r + 2] = v.z; d[4 * r + 3] = v.w;
  }
}

__device__ __forceinline__ void load_patch(float (&d)[16], const float* __restrict__ chan, unsigned bytes, const Patch16& p) {
  const i32x4 rs = plane_rsrc(chan, bytes);
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const f32x2 a = savfi_raw_buffer_load_x2(rs, (int)p.va[r], 0, 0), b = savfi_raw_buffer_load_x2(rs, (int)p.vb[r], 0, 0);
    d[4 * r + 0] = a.x; d[4 * r + 1] = a.y; d[4 * r + 2] = b.x; d[4 * r + 3] = b.y;
  }
}

// Workgroup of tile (0, 0): the patch row that is image row 0 was loaded from column 0 instead of column -off in the
// tiles of tile column 0 that contain it -- tile (0, 0) = lane 0, patch row `off`, and for off = 2 also tile (1, 0) = lane
// `tbw` (the block's width in tiles), patch row 0: d[c] = loaded[c - off] for c >= off (the columns below `off` are
// cleared by the column masks).
__device__ __forceinline__ void fix_corner(float (&d)[16], int off, int tbw) {
  const unsigned long long lane0 = 1ull;
  if (off == 1) {
    asm volatile("v_cndmask_b32 %0, %0, %1, %2" : "+v"(d[7]) : "v"(d[6]), "s"(lane0));
    asm volatile("v_cndmask_b32 %0, %0, %1, %2" : "+v"(d[6]) : "v"(d[5]), "s"(lane0));
    asm volatile("v_cndmask_b32 %0, %0, %1, %2" : "+v"(d[5]) : "v"(d[4]), "s"(lane0));
  } else {
    const unsigned long long lane1 = 1ull << tbw;
    asm volatile("v_cndmask_b32 %0, %0, %1, %2" : "+v"(d[11]) : "v"(d[9]), "s"(lane0));
    asm volatile("v_cndmask_b32 %0, %0, %1, %2" : "+v"(d[10]) : "v"(d[8]), "s"(lane0));
    asm volatile("v_cndmask_b32 %0, %0, %1, %2" : "+v"(d[3]) : "v"(d[1]), "s"(lane1));
    asm volatile("v_cndmask_b32 %0, %0, %1, %2" : "+v"(d[2]) : "v"(d[0]), "s"(lane1));
  }
}

__device__ __forceinline__ void mask_patch(float (&d)[16], const LaneMasks& lm) {
#pragma unroll
  for (int c = 0; c < 4; ++c)
    if (lm.partial_cols & (1 << c)) {       // wave-uniform
#pragma unroll
      for (int r = 0; r < 4; ++r)
        asm volatile("v_cndmask_b32 %0, 0, %0, %1" : "+v"(d[4 * r + c]) : "s"(lm.col[c]));
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

// One chunk of a wave: 32 MFMAs on V(c) in 8 groups of 4, interleaved (sched_barrier pins the order) with
//   groups 0-3   B^T d of the patch of chunk c+1 (loaded two chunks ago), one column each
//   group  3     those patch registers are dead: the 4 row loads of chunk c+3 are issued into them
//   groups 4-7   (B^T d) B, one row each, written straight into the other V buffer
//   group  7     last use of this chunk's A fragments: those of chunk c+2 are loaded in place
template <typename PatchT>
__device__ __forceinline__ void chunk_body(f32x4 (&acc)[4][2][4], f32x4 (&afr)[2], const float* __restrict__ vcur,
                                           float (&d)[16], const PatchT& patch, int fixup, int tbw, const LaneMasks& lm,
                                           float* __restrict__ vnext,
                                           const float* __restrict__ plane3, unsigned plane_bytes, i32x4 urs, unsigned unext, unsigned ulane) {
  f32x4 bfr[2];
  float t[16];
  bfr[0] = *reinterpret_cast<const f32x4*>(vcur);
  if (fixup & 6) fix_corner(d, fixup >> 1, tbw);     // wave-uniform branches
  if (fixup & 1) mask_patch(d, lm);
#pragma unroll
  for (int g = 0; g < 8; ++g) {
    const int c = g >> 1, cb = g & 1, cur = c & 1;       // xi column, channel block
    if (cb == 0 && c < 3) {   // B fragments of the next xi
      bfr[cur ^ 1] = *reinterpret_cast<const f32x4*>(vcur + (c + 1) * CIB * VS);
    }
    const float a = afr[c >> 1][2 * (c & 1) + cb];
#pragma unroll
    for (int tt = 0; tt < 4; ++tt)
      acc[c][cb][tt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bfr[cur][tt], acc[c][cb][tt], 0, 0, 0);
    if (g < 4) {               // B^T d : column g
      t[0 * 4 + g] = d[0 * 4 + g] - d[2 * 4 + g];
      t[1 * 4 + g] = d[1 * 4 + g] + d[2 * 4 + g];
      t[2 * 4 + g] = d[2 * 4 + g] - d[1 * 4 + g];
      t[3 * 4 + g] = d[1 * 4 + g] - d[3 * 4 + g];
      if (g == 3) load_patch(d, plane3, plane_bytes, patch);
    } else {                   // (B^T d) B : row g - 4, written straight to the next V buffer
      const int r = g - 4;
      vnext[(r * 4 + 0) * CIB * VS] = t[r * 4 + 0] - t[r * 4 + 2];
      vnext[(r * 4 + 1) * CIB * VS] = t[r * 4 + 1] + t[r * 4 + 2];
      vnext[(r * 4 + 2) * CIB * VS] = t[r * 4 + 2] - t[r * 4 + 1];
      vnext[(r * 4 + 3) * CIB * VS] = t[r * 4 + 1] - t[r * 4 + 3];
    }
    if (g == 7) {
      afr[0] = savfi_raw_buffer_load_x4(urs, (int)ulane, (int)unext, 0);
      afr[1] = savfi_raw_buffer_load_x4(urs, (int)ulane + 16, (int)unext, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
  }
}

// VEC: the output width is even (pairs never straddle a row end and are 8-byte aligned)
// IN16: the input is unit-major (Patch16)
template <bool VEC, bool IN16 = false>
__global__ __launch_bounds__(WNT, 2) void wino_conv3x3(WinoArgs a) {
#ifdef WINO_TRACE
  const unsigned long long tr0 = __builtin_amdgcn_s_memrealtime();
#endif
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 15, kg = lane >> 4;

  // Work item of this workgroup: logical order [sample][channel block x split][tile block], tile block fastest.
  // Blocks are dispatched round-robin over the 8 XCDs (block b runs on XCD b % 8, each with its own 4 MB L2).  For the
  // layers with at most two channel blocks -- the large maps, where the input is most of the traffic -- the block index is
  // remapped (bijectively, any grid size) so that one XCD walks a CONTIGUOUS eighth of that order: neighbouring tile blocks
  // (shared halo rows, shared 128-byte lines) then meet in the same L2.  PMC fetch traffic: 3.0 x the input -> 1.0 x for
  // 32 -> 32 @ 384 x 512, 7.3 x -> 2.1 x for 51 -> 51 @ 258 x 450 (two channel blocks).  The time moves by 0-5 % only (the re-reads were served by
  // the Infinity Cache); with more channel blocks the remap measured SLOWER (256 -> 256 @ 48 x 64: 153 -> 175 us), so those keep
  // the dispatch order.
  const int nblk = a.IP / COB;
  int tb, cob, sp, n;
  {
    const unsigned flat = blockIdx.x, nwg = gridDim.x;
    const unsigned ncs = (unsigned)(nblk * a.nsplit), ntb = (unsigned)(a.tiles_y * a.tiles_x);
    const unsigned xcd = flat & 7u, q = nwg >> 3, rem = nwg & 7u;
    unsigned item = ncs <= 2u ? xcd * q + min(xcd, rem) + (flat >> 3) : flat;
    tb = (int)(item % ntb);
    item /= ntb;
    const unsigned cs = item % ncs;
    n = (int)(item / ncs);
    cob = (int)(cs % (unsigned)nblk);
    sp = (int)(cs / (unsigned)nblk);
  }
  const int tby = tb / a.tiles_x, tbx = tb - tby * a.tiles_x;
  const int i0 = cob * COB;                        // first produced channel
  const int task = a.T > 1 ? n % a.T : 0;
  const float* xp = a.x + (size_t)n * a.K * a.H * a.W;
  const size_t cplane = (size_t)a.H * a.W;

  // this thread's tile for the input transform: lane -> tile (4 x 16), wave -> channel within the chunk
  const int tsh = a.tile_shift, tbw = 1 << tsh, tbh = TILES_WG >> tsh;
  const int px0 = 2 * (tbx * tbw + (lane & (tbw - 1))) - a.off;
  using PatchT = typename std::conditional<IN16, Patch16, Patch>::type;
  PatchT patch;
  if constexpr (IN16) patch = make_patch16(2 * (tby * tbh + (lane >> tsh)) - a.off, px0, a.H, a.W, a.K);
  else patch = make_patch(2 * (tby * tbh + (lane >> tsh)) - a.off, px0, a.H, a.W);
  // IN16: the descriptor starts at the channel's first 64 bytes inside the sample and ends where the LAST channel's would: every real
  // access lies inside, 0x80000000 outside
  const unsigned plane_bytes = IN16 ? (unsigned)a.K * (unsigned)cplane * 4u - (unsigned)(a.K - 1) * 64u : (unsigned)cplane * 4u;

  f32x4 acc[4][2][4];
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
      for (int t = 0; t < 4; ++t) acc[c][cb][t] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // A fragments of this wave for chunk c: 2 x f32x4 at ((c * nblk + cob) * 4 + w) * 64 lanes * 32 bytes
  // this workgroup reduces over chunks [cbeg, nchunk) of the KP / CIB chunks (an even count)
  const int cbeg = sp * a.chunks_per_split, nchunk = min(cbeg + a.chunks_per_split, a.KP / CIB);
  const size_t ustride = (size_t)nblk * 4 * 64 * 32;                                     // bytes per chunk
  // through a raw buffer descriptor over this task's U: a per-lane offset computed once and a wave-uniform byte offset per chunk
  // (SALU) instead of a 64-bit per-lane address per load
  const unsigned ubytes = 16u * (unsigned)a.KP * (unsigned)a.IP * 4u;
  const i32x4 urs = plane_rsrc(a.U + (size_t)task * 16 * a.KP * a.IP, ubytes);
  const unsigned ubase = ((unsigned)cob * 4u + (unsigned)w) * 64u * 32u;
  const unsigned ulane = (unsigned)lane * 32u;
  auto plane_of = [&](int chunk) { return xp + (size_t)min(min(chunk, nchunk - 1) * CIB + w, a.K - 1) * (IN16 ? (size_t)16 : cplane); };
  auto u_of = [&](int chunk) { return ubase + (unsigned)min(chunk, nchunk - 1) * (unsigned)ustride; };

  // fix-ups (wave-uniform): bit 0 = some column of some lane is outside the image; bits 1-2 = `off` in the workgroup that
  // owns tile (0, 0) of a padded map (its lane 0 holds the row that would start before the plane)
  const LaneMasks lm = make_lane_masks(px0, a.W);
  const int fixup = IN16 ? 0 : (lm.partial_cols != 0 ? 1 : 0) | ((tb == 0 && a.off > 0) ? 2 * a.off : 0);

  // The 8 bias values of this wave are fetched HERE, with the first patches: vmcnt counts loads and stores in one in-order
  // counter, so a bias load issued between the output stage's stores (as the first version did, once per channel) can only be
  // waited for together with every older store's write acknowledge -- 8 round trips of ~0.8 us per workgroup, which made the
  // output stage as long as the channel loop of a 32-channel layer (7.3 of 18.7 us per workgroup, per-workgroup timestamps).
  float bvals[2][4];
#pragma unroll
  for (int cb = 0; cb < 2; ++cb)
#pragma unroll
    for (int rep = 0; rep < 4; ++rep) {
      const int i = i0 + 16 * cb + w + 4 * rep;
      bvals[cb][rep] = (a.bias && a.nsplit == 1 && i < a.I) ? a.bias[task * a.I + i] : 0.f;
    }

  // prologue: P(0) -> V(0); P(1), P(2), A(0), A(1) in flight / resident
  float dA[16], dB[16];        // dA: patches of even chunks, dB: odd chunks
  f32x4 afrA[2], afrB[2];
  load_patch(dA, plane_of(cbeg), plane_bytes, patch);
  load_patch(dB, plane_of(cbeg + 1), plane_bytes, patch);
  afrA[0] = savfi_raw_buffer_load_x4(urs, (int)ulane, (int)u_of(cbeg), 0);
  afrA[1] = savfi_raw_buffer_load_x4(urs, (int)ulane + 16, (int)u_of(cbeg), 0);
  afrB[0] = savfi_raw_buffer_load_x4(urs, (int)ulane, (int)u_of(cbeg + 1), 0);
  afrB[1] = savfi_raw_buffer_load_x4(urs, (int)ulane + 16, (int)u_of(cbeg + 1), 0);
  {
    float v[16];
    if (fixup & 6) fix_corner(dA, fixup >> 1, tbw);
    if (fixup & 1) mask_patch(dA, lm);
    input_transform(v, dA);
    float* vb = lds + w * VS + 4 * (lane & 15) + (lane >> 4);       // = vwoff below: [k = w][j][t]
#pragma unroll
    for (int xi = 0; xi < 16; ++xi) vb[xi * CIB * VS] = v[xi];
  }
  load_patch(dA, plane_of(cbeg + 2), plane_bytes, patch);
  // Everything the prologue loaded must have landed before the loop: otherwise the compiler's wait-count bookkeeping
  // carries "may still be in flight" into the loop header and, vmcnt being in-order, makes every iteration wait for
  // its own freshly issued prefetches before the first MFMA.
  __builtin_amdgcn_s_waitcnt(0);
  __syncthreads();
#ifdef WINO_TRACE
  const unsigned long long tr1 = __builtin_amdgcn_s_memrealtime();
#endif

  // V[xi][k][j][t] for tile 16 t + j: a lane's four B fragments (t = 0..3) of one (xi, k) are ONE ds_read_b128; the 64 lanes of
  // a writing wave still cover the 64 dwords of a row exactly once (4 j + t is a permutation): no bank conflicts either way
  const int vwoff = w * VS + 4 * (lane & 15) + (lane >> 4);   // V write: [xi][k = w][j = lane % 16][t = lane / 16]
  const int vroff = (4 * w) * CIB * VS + kg * VS + 4 * j;      // V read:  [xi = 4w + c][k = kg][j][t = 0..3]
  for (int ch = cbeg; ch < nchunk; ch += 2) {
    // even chunk: MFMAs on V(0) with A(ch); transforms P(ch+1) = dB -> V(1); reloads dB <- P(ch+3), afrA <- A(ch+2)
    chunk_body(acc, afrA, lds + vroff, dB, patch, fixup, tbw, lm, lds + VBUF + vwoff,
               plane_of(ch + 3), plane_bytes, urs, u_of(ch + 2), ulane);
    __syncthreads();
    // nchunk is even (KP is a multiple of 2 * CIB): an `if (ch + 1 < nchunk)` here would make the compiler assume the
    // A fragments loaded at the end of the even chunk may be the youngest load in flight -> vmcnt(0) every iteration
    chunk_body(acc, afrB, lds + VBUF + vroff, dA, patch, fixup, tbw, lm, lds + vwoff,
               plane_of(ch + 4), plane_bytes, urs, u_of(ch + 3), ulane);
    __syncthreads();
  }

#ifdef WINO_TRACE
  const unsigned long long tr2 = __builtin_amdgcn_s_memrealtime();
#endif
  // ---- output stage ------------------------------------------------------------------------------------------
  // column half of A^T M A in registers (this wave holds the whole row r = w):  s0 = m0 + m1 + m2,  s1 = m1 - m2 - m3
  // accumulator tile layout: row (channel) = 4 * kg + reg, column (tile) = j
  // Stores are RAW BUFFER stores through a descriptor that spans one output plane (zero bytes for a padded channel):
  // a lane whose pixel lies outside the map gets an offset beyond the plane and the hardware drops its store.  The
  // output stage is therefore straight-line code -- no per-lane branches around the stores, a fixed number of memory
  // instructions per wave -- which is also what lets the compiler count vmcnt exactly instead of waiting for zero.
  unsigned ooff[2][2];                          // [row][element]: byte offset inside the plane, or out of range
  {
    const int oty = tby * tbh + (lane >> tsh), otx = tbx * tbw + (lane & (tbw - 1));
    const int oy = 2 * oty, ox = 2 * otx;
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
      for (int c = 0; c < 2; ++c)
        ooff[r][c] = (oy + r < a.Ho && ox + c < a.Wo) ? (unsigned)((oy + r) * a.Wo + ox + c) * 4u : 0x80000000u;
    if (a.out_unit16) {      // unit-major: a sample is [y][x / 16][channel][16]; ox is even, so the pair of a row store stays inside its unit
#pragma unroll
      for (int r = 0; r < 2; ++r)
        ooff[r][0] = (oy + r < a.Ho && ox < a.Wo) ? (unsigned)(((oy + r) * (a.Wo >> 4) + (ox >> 4)) * a.I) * 64u + (unsigned)(ox & 15) * 4u : 0x80000000u;
    }
  }
  float* const obase = a.nsplit == 1 ? a.out : a.partial + (size_t)sp * a.N * a.I * a.Ho * a.Wo;
  const float slope = a.nsplit == 1 ? a.slope : 1.f;       // bias / activation happen in wino_split_reduce
  const unsigned oplane_bytes = (unsigned)(a.Ho * a.Wo) * 4u;
  // the mask values of this lane's 32 outputs, ALL before the first store (a load issued between stores can only be waited for
  // together with every older store's write acknowledge: see the bias values above)
  const bool masked = a.mask != nullptr && a.nsplit == 1;
  f32x2 mk[2][4][2];
  if (masked) {
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
      for (int rep = 0; rep < 4; ++rep) {
        const int i = i0 + 16 * cb + w + 4 * rep;
        const i32x4 mrs = plane_rsrc(a.mask + ((size_t)n * a.I + min(i, a.I - 1)) * a.Ho * a.Wo, i < a.I ? oplane_bytes : 0u);
#pragma unroll
        for (int r = 0; r < 2; ++r) {
          if (VEC) mk[cb][rep][r] = savfi_raw_buffer_load_x2(mrs, (int)ooff[r][0], 0, 0);
          else mk[cb][rep][r] = (f32x2){savfi_raw_buffer_load_x1(mrs, (int)ooff[r][0], 0, 0), savfi_raw_buffer_load_x1(mrs, (int)ooff[r][1], 0, 0)};
        }
      }
  }
#pragma unroll
  for (int cb = 0; cb < 2; ++cb) {
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float m0 = acc[0][cb][t][r], m1 = acc[1][cb][t][r], m2 = acc[2][cb][t][r], m3 = acc[3][cb][t][r];
        float* xw = lds + ((2 * w) * 16 + 4 * kg + r) * XS + 16 * t + j;
        xw[0] = m0 + m1 + m2;
        xw[16 * XS] = m1 - m2 - m3;
      }
    __syncthreads();
#pragma unroll
    for (int rep = 0; rep < 4; ++rep) {
      const int il = w + 4 * rep, t = lane;         // channel within the block of 16 (wave-uniform), tile
      const int i = i0 + 16 * cb + il;
      float s[4][2];
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = 0; c < 2; ++c) s[r][c] = lds[((2 * r + c) * 16 + il) * XS + t];
      float y[2][2];
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        y[0][c] = s[0][c] + s[1][c] + s[2][c];
        y[1][c] = s[1][c] - s[2][c] - s[3][c];
      }
      const float b = bvals[cb][rep];
      const i32x4 ors = a.out_unit16 ? plane_rsrc(obase + (size_t)n * a.I * a.Ho * a.Wo, i < a.I ? (unsigned)a.I * oplane_bytes : 0u)
                                     : plane_rsrc(obase + ((size_t)n * a.I + min(i, a.I - 1)) * a.Ho * a.Wo, i < a.I ? oplane_bytes : 0u);
      const unsigned ch_off = a.out_unit16 ? (unsigned)min(i, a.I - 1) * 64u : 0u;
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        float v0 = y[r][0] + b, v1 = y[r][1] + b;
        v0 = fmaxf(v0, 0.f) + slope * fminf(v0, 0.f);      // v > 0 ? v : slope * v
        v1 = fmaxf(v1, 0.f) + slope * fminf(v1, 0.f);
        if (masked) {
          v0 = mk[cb][rep][r].x > 0.f ? v0 : v0 * a.mask_slope;
          v1 = mk[cb][rep][r].y > 0.f ? v1 : v1 * a.mask_slope;
        }
        if (VEC) {
          savfi_raw_buffer_store_x2((f32x2){v0, v1}, ors, (int)(ooff[r][0] + ch_off), 0, 0);
        } else {
          savfi_raw_buffer_store_x1(v0, ors, (int)ooff[r][0], 0, 0);
          savfi_raw_buffer_store_x1(v1, ors, (int)ooff[r][1], 0, 0);
        }
      }
    }
    __syncthreads();
  }
#ifdef WINO_TRACE
  if (threadIdx.x == 0) {
    const unsigned long long tr3 = __builtin_amdgcn_s_memrealtime();
    __builtin_amdgcn_s_waitcnt(0);
    const unsigned long long tr4 = __builtin_amdgcn_s_memrealtime();
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    unsigned long long* t = a.trace + (size_t)blockIdx.x * 8;
    t[0] = tr0; t[1] = tr1; t[2] = tr2; t[3] = tr3; t[4] = tr4; t[5] = hw; t[6] = xcc; t[7] = blockIdx.x;
  }
#endif
}

// out = act(sum over splits of partial (fixed order: deterministic) + bias[c])
__global__ __launch_bounds__(256) void wino_split_reduce(const float* __restrict__ partial, const float* __restrict__ bias,
                                                         float* __restrict__ out, int nsplit, size_t total, int I, int HW,
                                                         float slope, int T, const float* __restrict__ mask, float mask_slope) {
  const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= total) return;
  float acc = 0.f;
  for (int s = 0; s < nsplit; ++s) acc += partial[(size_t)s * total + e];
  if (bias) {
    const size_t plane = e / HW;                    // n * I + c
    acc += bias[((plane / I) % T) * I + plane % I];
  }
  acc = acc > 0.f ? acc : slope * acc;
  if (mask) acc = mask[e] > 0.f ? acc : acc * mask_slope;
  out[e] = acc;
}

int round_up(int v, int m) { return (v + m - 1) / m * m; }

// Launch plan shared by the workspace query and the launch.  Deep layers have few tiles and many reduction channels
// (256->256 at 48x64: 192 workgroups x 64 chunks on 512 slots): the chunks are split over up to 8 workgroups whose
// raw partial outputs are added by wino_split_reduce.
#ifndef SAVFI_WINO_SPLIT_SLOTS
#define SAVFI_WINO_SPLIT_SLOTS 512      // variant builds: 0 = never split
#endif

struct WinoPlan {
  int K, I, KP, IP, off, Ho, Wo, th, tw, tile_shift, nsplit, chunks_per_split;
  int64_t u_floats, partial_floats;
  bool f4;            // Winograd F(4x4, 3x3) (winograd4.h): 4x4-pixel tiles, 32 per workgroup, no reduction split
};

// padded reduction / produced channel counts of a layer's transformed filter (both forms)
// `form` 2: the F(2x2) kernel whatever the channel counts (a caller's choice per layer AND map: savfi_conv3x3_*_form_f32 / bit 1 of `mode`);
// 0: by the channel counts (F(4x4) up to 512 -> 512)
inline bool f4_for(int K, int I, int form) { return form != 2 && w4::use_f4(K, I); }
inline int kp_for(int K, int I, int form) { return f4_for(K, I, form) ? w4::kp_of(K) : (K + 2 * CIB - 1) / (2 * CIB) * (2 * CIB); }
inline int ip_for(int K, int I, int form) { return f4_for(K, I, form) ? w4::ip_of(I) : (I + COB - 1) / COB * COB; }
inline int64_t u_floats_for(int K, int I, int form) { return (int64_t)(f4_for(K, I, form) ? w4::PTS : 16) * kp_for(K, I, form) * ip_for(K, I, form); }

bool make_plan(WinoPlan& p, int N, int Ci, int Co, int H, int W, int pad, int mode) {
  const int form = (mode & 2) ? 2 : 0;       // bit 1 of `mode`: the F(2x2) form
  mode &= 1;
  p.K = mode == 0 ? Ci : Co;
  p.I = mode == 0 ? Co : Ci;       // reduction / produced channels
  p.f4 = f4_for(p.K, p.I, form);
  p.KP = kp_for(p.K, p.I, form);   // F(2x2): an even number of chunks (see the channel loop)
  p.IP = ip_for(p.K, p.I, form);
  // forward: patch origin 2t - pad; gradient of a pad-p convolution = pad-(2-p) correlation with the flipped filter
  p.off = mode == 0 ? pad : 2 - pad;
  p.Ho = H + 2 * p.off - 2;
  p.Wo = W + 2 * p.off - 2;
  if (p.Ho <= 0 || p.Wo <= 0) return false;
  if (p.f4) {
    // 32 tiles of 4 x 4 pixels as a 2^(5-s) x 2^s block: fewest blocks, ties to the WIDEST block (1 x 32 tiles = 4 x 128 pixels: 512-byte
    // row segments in and out).  Measured against the 4 x 8 preference of the first version, tools/r6/wino4_time.py: 32 -> 32 @384x512
    // 161 -> 152 us, 51 -> 51 @258x450 813 -> 760, 64 -> 64 @192x256 113 -> 110, the others within +-1 % (SAVFI_W4_TS_ORDER=0: the old order)
    const int ty4 = savfi_cdiv(p.Ho, 4), tx4 = savfi_cdiv(p.Wo, 4);
    int64_t best4 = -1;
#ifndef SAVFI_W4_TS_ORDER
#define SAVFI_W4_TS_ORDER 2
#endif
#if SAVFI_W4_TS_ORDER == 0
    for (int s : {3, 4, 2, 5, 1, 0}) {
#elif SAVFI_W4_TS_ORDER == 1
    for (int s : {4, 3, 5, 2, 1, 0}) {
#else
    for (int s : {5, 4, 3, 2, 1, 0}) {
#endif
      const int64_t blocks = (int64_t)savfi_cdiv(ty4, w4::TT >> s) * savfi_cdiv(tx4, 1 << s);
      if (best4 < 0 || blocks < best4) { best4 = blocks; p.tile_shift = s; }
    }
    p.th = savfi_cdiv(ty4, w4::TT >> p.tile_shift);
    p.tw = savfi_cdiv(tx4, 1 << p.tile_shift);
    // deep layers on small maps: the reduction chunks split over workgroups where the launch would leave most of the 512 workgroup slots
    // empty (at least 4 chunks = 32 channels per workgroup); raw partial outputs, summed by wino_split_reduce
    const int nchunk4 = p.KP / w4::KC;
    const int64_t wgs4 = (int64_t)p.th * p.tw * (p.IP / w4::COB) * N;
    // (measured, tools/r6/wino4_time.py small: 256 -> 256 @24x32 T = 4 x 2: 128 workgroups -> 4 x 128: 63 -> 46 us; 512 -> 512 @24x32: 256 -> 2 x 256:
    // 138 us against the direct kernel's 164; 128 -> 128 @48x64, 192 workgroups of 16 chunks: 37 us unsplit, 47 split in two -- a split needs a
    // long reduction to pay for its second pass: from 256 channels, launches of at most 256 workgroups)
    int want4 = (int)(SAVFI_WINO_SPLIT_SLOTS / (wgs4 > 0 ? wgs4 : 1));
    want4 = (want4 < 2 || nchunk4 < 32) ? 1 : (want4 > 8 ? 8 : want4);
    int cps4 = savfi_cdiv(nchunk4, want4);
    if (cps4 < 8) cps4 = nchunk4 < 8 ? nchunk4 : 8;
    p.chunks_per_split = cps4;
    p.nsplit = savfi_cdiv(nchunk4, cps4);
    p.u_floats = u_floats_for(p.K, p.I, form);
    p.partial_floats = p.nsplit > 1 ? (int64_t)p.nsplit * N * p.I * p.Ho * p.Wo : 0;
    return true;
  }
  // tile block shape: the one that covers the tile map with the fewest blocks (ties: the widest rows, 4 x 16 first)
  const int ty = savfi_cdiv(p.Ho, 2), tx = savfi_cdiv(p.Wo, 2);
  int64_t best = -1;
  for (int s : {4, 3, 5, 2}) {
    const int64_t blocks = (int64_t)savfi_cdiv(ty, TILES_WG >> s) * savfi_cdiv(tx, 1 << s);
    if (best < 0 || blocks < best) { best = blocks; p.tile_shift = s; }
  }
  p.th = savfi_cdiv(ty, TILES_WG >> p.tile_shift);
  p.tw = savfi_cdiv(tx, 1 << p.tile_shift);
  const int nchunk = p.KP / CIB;
  const int64_t wgs = (int64_t)p.th * p.tw * (p.IP / COB) * N;
  // split only launches that would leave workgroup slots empty (2 per CU x 256 CUs): the partial outputs cost an
  // extra pass, which loses on launches that already fill the chip (64->64 at 192x256: 64 -> 87 us when split)
  constexpr int slots = SAVFI_WINO_SPLIT_SLOTS;
  int want = (int)(slots / wgs);
  want = want < 1 ? 1 : (want > 8 ? 8 : want);
  int cps = round_up(savfi_cdiv(nchunk, want), 2);
  if (cps < 8) cps = nchunk < 8 ? nchunk : 8;          // at least 8 chunks per workgroup (prologue / output stage cost)
  p.chunks_per_split = cps;
  p.nsplit = savfi_cdiv(nchunk, cps);
  if (SAVFI_WINO_SPLIT_SLOTS == 0) { p.nsplit = 1; p.chunks_per_split = nchunk; }
  p.u_floats = (int64_t)16 * p.KP * p.IP;
  p.partial_floats = p.nsplit > 1 ? (int64_t)p.nsplit * N * p.I * p.Ho * p.Wo : 0;
  return true;
}



}  // namespace

#ifdef WINO_TRACE
static unsigned long long* g_trace = nullptr;
static size_t g_trace_wgs = 0;
static unsigned long long* savfi_wino_trace_buffer(size_t wgs) {
  if (!g_trace) (void)hipMalloc(&g_trace, (size_t)(1 << 20) * 64);
  g_trace_wgs = wgs;
  return g_trace;
}
extern "C" int savfi_debug_wino_trace(unsigned long long* host, long long cap) {   // copies the last launch's records
  (void)hipDeviceSynchronize();
  const size_t n = g_trace_wgs < (size_t)cap ? g_trace_wgs : (size_t)cap;
  (void)hipMemcpy(host, g_trace, n * 64, hipMemcpyDeviceToHost);
  return (int)n;
}
#endif

extern "C" int64_t savfi_conv3x3_tasks_workspace_floats(int N, int T, int Ci, int Co, int H, int W, int pad, int mode) {
  if (N <= 0 || T <= 0 || N % T != 0 || Ci <= 0 || Co <= 0 || H <= 0 || W <= 0) return SAVFI_E_SHAPE;
  if ((mode & ~3) || (pad != 0 && pad != 1)) return SAVFI_E_UNSUPPORTED;
  WinoPlan p;
  if (!make_plan(p, N, Ci, Co, H, W, pad, mode)) return SAVFI_E_SHAPE;
  return (int64_t)T * p.u_floats + p.partial_floats;
}

// Workgroups the F(4x4, 3x3) kernel (winograd4.h) launches for this call -- 0 when the layer's channel counts keep it on F(2x2).  The
// host routes by it: F(4x4) has no reduction split, so a launch that cannot fill the chip's 512 workgroup slots belongs elsewhere.
extern "C" int64_t savfi_conv3x3_f4_workgroups(int N, int Ci, int Co, int H, int W, int pad, int mode) {
  if (N <= 0 || Ci <= 0 || Co <= 0 || H <= 0 || W <= 0) return SAVFI_E_SHAPE;
  if ((mode & ~3) || (pad != 0 && pad != 1)) return SAVFI_E_UNSUPPORTED;
  WinoPlan p;
  if (!make_plan(p, N, Ci, Co, H, W, pad, mode)) return SAVFI_E_SHAPE;
  return p.f4 ? (int64_t)p.th * p.tw * (p.IP / w4::COB) * p.nsplit * N : 0;
}

extern "C" int64_t savfi_conv3x3_workspace_floats(int N, int Ci, int Co, int H, int W, int pad, int mode) {
  return savfi_conv3x3_tasks_workspace_floats(N, 1, Ci, Co, H, W, pad, mode);
}

namespace {

// launches wino_conv3x3 (+ the split reduction) on an already transformed filter U [T][16 * KP * IP]
int launch_conv(const WinoPlan& p, const float* x, const float* U, const float* bias, float* out, float* partial, int N, int T,
                int H, int W, int mode, float slope, hipStream_t st, const float* mask = nullptr, float mask_slope = 1.f,
                int out_unit16 = 0, int in_unit16 = 0) {
  if (p.f4) {
    const int64_t wgs4 = (int64_t)p.th * p.tw * (p.IP / w4::COB) * p.nsplit * N;
    if (wgs4 > 0x7fffffffLL || T > 65535) return SAVFI_E_TOOBIG;
    if (mask && (out_unit16 || in_unit16)) return SAVFI_E_UNSUPPORTED;
    if (p.nsplit > 1 && (out_unit16 || in_unit16 || !partial)) return SAVFI_E_UNSUPPORTED;
    if (in_unit16 && ((p.off != 1 && p.off != 2) || p.Wo % 2 != 0)) return SAVFI_E_UNSUPPORTED;
    constexpr size_t lds4 = (size_t)w4::LDS_FLOATS * sizeof(float);      // 72 KB: two workgroups per CU
    w4::W4Args a4{x, U, (mode & 1) == 0 ? bias : nullptr, out, p.K, p.I, p.KP, p.IP, H, W, p.Ho, p.Wo, p.off, p.th, p.tw, slope,
                  p.tile_shift, T, N, p.nsplit > 1 ? nullptr : mask, mask_slope, out_unit16, p.nsplit, p.chunks_per_split, partial};
    const int vecw = p.Wo % 4 == 0 ? 4 : (p.Wo % 2 == 0 ? 2 : 1);
    auto go = [&](auto kern) -> int {
      static uint32_t configured = 0;
      if (int rc = savfi_ensure_dynamic_lds(reinterpret_cast<const void*>(kern), lds4, configured)) return rc;
      hipLaunchKernelGGL(kern, dim3((unsigned)wgs4), dim3(256), lds4, st, a4);
      return savfi_launch_status();
    };
    if (in_unit16 && p.off == 1) return vecw == 4 ? go(w4::wino4_conv3x3<4, 2, false>) : go(w4::wino4_conv3x3<2, 2, false>);
    if (in_unit16) return vecw == 4 ? go(w4::wino4_conv3x3<4, 3, false>) : go(w4::wino4_conv3x3<2, 3, false>);
    if (p.nsplit > 1) {
      const int rc = vecw == 4 ? go(w4::wino4_conv3x3<4, 0, false>) : vecw == 2 ? go(w4::wino4_conv3x3<2, 0, false>) : go(w4::wino4_conv3x3<1, 0, false>);
      if (rc != SAVFI_OK) return rc;
      const size_t total = (size_t)N * p.I * p.Ho * p.Wo;
      hipLaunchKernelGGL(wino_split_reduce, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, partial, (mode & 1) == 0 ? bias : nullptr, out,
                         p.nsplit, total, p.I, p.Ho * p.Wo, slope, T, mask, mask_slope);
      return savfi_launch_status();
    }
    if (mask) return vecw == 4 ? go(w4::wino4_conv3x3<4, 0, true>) : vecw == 2 ? go(w4::wino4_conv3x3<2, 0, true>) : go(w4::wino4_conv3x3<1, 0, true>);
    return vecw == 4 ? go(w4::wino4_conv3x3<4, 0, false>) : vecw == 2 ? go(w4::wino4_conv3x3<2, 0, false>) : go(w4::wino4_conv3x3<1, 0, false>);
  }
  const int64_t wgs = (int64_t)p.th * p.tw * (p.IP / COB) * p.nsplit * N;
  if (wgs > 0x7fffffffLL || T > 65535) return SAVFI_E_TOOBIG;
  constexpr size_t lds = (size_t)LDS_FLOATS * sizeof(float);      // 40 KB: two workgroups per CU
  const float* b = (mode & 1) == 0 ? bias : nullptr;
  WinoArgs a{x, U, b, out, p.K, p.I, p.KP, p.IP, H, W, p.Ho, p.Wo, p.off, p.th, p.tw, slope, p.tile_shift, p.nsplit,
             p.chunks_per_split, partial, T, N, mask, mask_slope, out_unit16
#ifdef WINO_TRACE
             , savfi_wino_trace_buffer((size_t)wgs)
#endif
  };
  if (in_unit16) hipLaunchKernelGGL((wino_conv3x3<true, true>), dim3((unsigned)wgs), dim3(WNT), lds, st, a);
  else if (p.Wo % 2 == 0) hipLaunchKernelGGL(wino_conv3x3<true>, dim3((unsigned)wgs), dim3(WNT), lds, st, a);
  else hipLaunchKernelGGL(wino_conv3x3<false>, dim3((unsigned)wgs), dim3(WNT), lds, st, a);
  if (int e = savfi_launch_status()) return e;
  if (p.nsplit > 1) {
    const size_t total = (size_t)N * p.I * p.Ho * p.Wo;
    hipLaunchKernelGGL(wino_split_reduce, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, partial, b, out, p.nsplit, total,
                       p.I, p.Ho * p.Wo, slope, T, mask, mask_slope);
    return savfi_launch_status();
  }
  return SAVFI_OK;
}

int check_conv_args(WinoPlan& p, int N, int T, int Ci, int Co, int H, int W, int pad, int mode) {
  if (N <= 0 || T <= 0 || N % T != 0 || Ci <= 0 || Co <= 0 || H <= 0 || W <= 0) return SAVFI_E_SHAPE;
  if ((mode & ~3) || (pad != 0 && pad != 1) || (int64_t)H * W < 4) return SAVFI_E_UNSUPPORTED;   // rows are 16-byte loads
  if (!make_plan(p, N, Ci, Co, H, W, pad, mode)) return SAVFI_E_SHAPE;
  // 32-bit byte offsets inside a channel plane, and 0x80000000 must lie beyond the input and the output plane
  if ((int64_t)H * W >= ((int64_t)1 << 29) || (int64_t)p.Ho * p.Wo >= ((int64_t)1 << 29)) return SAVFI_E_TOOBIG;
  if (p.u_floats >= ((int64_t)1 << 29)) return SAVFI_E_TOOBIG;           // one task's transformed filter: 32-bit byte offsets
  // F(4x4): a sample's channels ride in ONE descriptor (two channels per wave); the output's too
  if (p.f4 && ((int64_t)p.K * H * W >= ((int64_t)1 << 29) || (int64_t)p.I * p.Ho * p.Wo >= ((int64_t)1 << 29))) return SAVFI_E_TOOBIG;
  return SAVFI_OK;
}

}  // namespace

// mode 0: out[n][co] = act(conv2d(x[n], w[n % T], zero padding `pad`)[co] + bias[n % T][co])   x [N][Ci][H][W] -> [N][Co][H+2pad-2][W+2pad-2]
// mode 1: its data gradient: x = gy [N][Co][H][W] -> out = gx [N][Ci][H+2-2pad][W+2-2pad]
extern "C" int savfi_conv3x3_tasks_f32(const float* x, const float* w, const float* bias, float* out, float* workspace,
                                       int N, int T, int Ci, int Co, int H, int W, int pad, int mode, float slope, void* stream) {
  if (!x || !w || !out || !workspace) return SAVFI_E_NULL;
  WinoPlan p;
  if (int e = check_conv_args(p, N, T, Ci, Co, H, W, pad, mode)) return e;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(wino_filter_transform, dim3(p.KP / 4, savfi_cdiv(p.IP, 64) * T), dim3(256), 0, st, w, workspace, Co, Ci, p.K,
                     p.I, p.KP, p.IP, mode & 1, p.f4 ? 1 : 0);
  if (int e = savfi_launch_status()) return e;
  return launch_conv(p, x, workspace, bias, out, workspace + (int64_t)T * p.u_floats, N, T, H, W, mode, slope, st);
}

// The two halves of savfi_conv3x3_tasks_f32 for a caller that runs forward AND data gradient on the same filters (a training
// step): savfi_conv3x3_filters_f32 writes the forward filter transform to u_fwd and / or the data-gradient one to u_bwd (either
// may be NULL; savfi_conv3x3_filter_floats(T, Ci, Co, mode) floats each) in ONE launch, savfi_conv3x3_tasks_pre_f32 convolves
// with an already transformed filter (workspace: savfi_conv3x3_tasks_workspace_floats minus the filter, i.e.
// savfi_conv3x3_tasks_pre_workspace_floats, possibly 0 -> may be NULL).
extern "C" int64_t savfi_conv3x3_filter_floats(int T, int Ci, int Co, int mode) {
  if (T <= 0 || Ci <= 0 || Co <= 0 || (mode & ~3)) return SAVFI_E_SHAPE;
  const int K = (mode & 1) == 0 ? Ci : Co, I = (mode & 1) == 0 ? Co : Ci;
  return (int64_t)T * u_floats_for(K, I, (mode & 2) ? 2 : 0);
}

// `form` 0: the kernel form follows the channel counts; 2: the F(2x2) form (for a layer the caller will run with bit 1 of `mode` set)
extern "C" int savfi_conv3x3_filters_form_f32(const float* w, float* u_fwd, float* u_bwd, int T, int Ci, int Co, int form, void* stream) {
  if (!w || (!u_fwd && !u_bwd)) return SAVFI_E_NULL;
  if (T <= 0 || T > 65535 || Ci <= 0 || Co <= 0) return SAVFI_E_SHAPE;
  if (form != 0 && form != 2) return SAVFI_E_UNSUPPORTED;
  const int KPf = kp_for(Ci, Co, form), IPf = ip_for(Ci, Co, form), KPb = kp_for(Co, Ci, form), IPb = ip_for(Co, Ci, form);
  const int f4 = f4_for(Ci, Co, form) ? 1 : 0;
  hipStream_t st = (hipStream_t)stream;
  if (u_fwd && u_bwd) {
    const int gx = (KPf > KPb ? KPf : KPb) / 4, gy = savfi_cdiv(IPf > IPb ? IPf : IPb, 64) * T;
    hipLaunchKernelGGL(wino_filter_transform_dual, dim3(gx, gy, 2), dim3(256), 0, st, w, u_fwd, u_bwd, Co, Ci, KPf, IPf, KPb, IPb, T, f4);
  } else if (u_fwd) {
    hipLaunchKernelGGL(wino_filter_transform, dim3(KPf / 4, savfi_cdiv(IPf, 64) * T), dim3(256), 0, st, w, u_fwd, Co, Ci, Ci, Co, KPf, IPf, 0, f4);
  } else {
    hipLaunchKernelGGL(wino_filter_transform, dim3(KPb / 4, savfi_cdiv(IPb, 64) * T), dim3(256), 0, st, w, u_bwd, Co, Ci, Co, Ci, KPb, IPb, 1, f4);
  }
  return savfi_launch_status();
}
extern "C" int savfi_conv3x3_filters_f32(const float* w, float* u_fwd, float* u_bwd, int T, int Ci, int Co, void* stream) {
  return savfi_conv3x3_filters_form_f32(w, u_fwd, u_bwd, T, Ci, Co, 0, stream);
}

// form: per layer, as savfi_conv3x3_filters_form_f32 (NULL: 0 for every layer)
extern "C" int savfi_conv3x3_filters_multi_form_f32(const float* const* w, float* const* u_fwd, float* const* u_bwd, const int* T,
                                                    const int* Ci, const int* Co, const int* form, int n, void* stream) {
  if (!w || !u_fwd || !u_bwd || !T || !Ci || !Co) return SAVFI_E_NULL;
  if (n <= 0) return SAVFI_E_SHAPE;
  FtTable tb;
  tb.n = 0;
  int blocks = 0;
  auto flush = [&]() {
    if (tb.n == 0) return (int)SAVFI_OK;
    hipLaunchKernelGGL(wino_filter_transform_multi, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, tb);
    tb.n = 0;
    blocks = 0;
    return savfi_launch_status();
  };
  for (int i = 0; i < n; ++i) {
    if (!w[i] || (!u_fwd[i] && !u_bwd[i])) return SAVFI_E_NULL;
    if (T[i] <= 0 || T[i] > 65535 || Ci[i] <= 0 || Co[i] <= 0) return SAVFI_E_SHAPE;
    const int fm = form ? form[i] : 0;
    if (fm != 0 && fm != 2) return SAVFI_E_UNSUPPORTED;
    for (int mode = 0; mode < 2; ++mode) {
      float* dst = mode == 0 ? u_fwd[i] : u_bwd[i];
      if (!dst) continue;
      FtJob& j = tb.job[tb.n];
      j.w = w[i]; j.U = dst; j.Co = Co[i]; j.Ci = Ci[i]; j.mode = mode;
      j.K = mode == 0 ? Ci[i] : Co[i];
      j.I = mode == 0 ? Co[i] : Ci[i];
      j.KP = kp_for(j.K, j.I, fm);
      j.IP = ip_for(j.K, j.I, fm);
      j.nbx = j.KP / 4;
      j.first_block = blocks;
      j.f4 = f4_for(j.K, j.I, fm) ? 1 : 0;
      blocks += j.nbx * savfi_cdiv(j.IP, 64) * T[i];
      if (++tb.n == WINO_FT_JOBS) {
        const int rc = flush();
        if (rc != SAVFI_OK) return rc;
      }
    }
  }
  return flush();
}
extern "C" int savfi_conv3x3_filters_multi_f32(const float* const* w, float* const* u_fwd, float* const* u_bwd, const int* T,
                                               const int* Ci, const int* Co, int n, void* stream) {
  return savfi_conv3x3_filters_multi_form_f32(w, u_fwd, u_bwd, T, Ci, Co, nullptr, n, stream);
}

extern "C" int64_t savfi_conv3x3_tasks_pre_workspace_floats(int N, int T, int Ci, int Co, int H, int W, int pad, int mode) {
  WinoPlan p;
  if (int e = check_conv_args(p, N, T, Ci, Co, H, W, pad, mode)) return e;
  return p.partial_floats;
}

extern "C" int savfi_conv3x3_tasks_pre_f32(const float* x, const float* u, const float* bias, float* out, float* workspace,
                                           int N, int T, int Ci, int Co, int H, int W, int pad, int mode, float slope, void* stream) {
  if (!x || !u || !out) return SAVFI_E_NULL;
  WinoPlan p;
  if (int e = check_conv_args(p, N, T, Ci, Co, H, W, pad, mode)) return e;
  if (p.partial_floats > 0 && !workspace) return SAVFI_E_NULL;
  return launch_conv(p, x, u, bias, out, workspace, N, T, H, W, mode, slope, (hipStream_t)stream);
}

// savfi_conv3x3_tasks_pre_f32, forward only, with the result written UNIT-MAJOR: out[n] is [Ho][Wo / 16][Co][16] instead of [Co][Ho][Wo] --
// the 16-pixel units of csrc/sepconv_ws.hip with a unit's Co x 64 bytes contiguous.  The SepConv plugin's last Subnet convolution writes
// its 51 taps per pixel in this layout and the 51-tap op reads them as contiguous runs instead of 64-byte pieces of 51 planes a multiple
// of 64 KB apart.  Wo % 16 == 0, a sample below 2^31 bytes, no reduction split; SAVFI_E_UNSUPPORTED otherwise.
static int unit16_ok(const WinoPlan& p) {
  // (F(4x4) marks a dropped channel with bit 30 of the offset: a sample below 2^30 bytes)
  return p.nsplit == 1 && p.Wo % 16 == 0 && (int64_t)p.I * p.Ho * p.Wo * 4 < ((int64_t)1 << (p.f4 ? 30 : 31));
}
extern "C" int savfi_conv3x3_unit16_supported(int N, int T, int Ci, int Co, int H, int W, int pad) {
  WinoPlan p;
  if (check_conv_args(p, N, T, Ci, Co, H, W, pad, 0)) return 0;
  return unit16_ok(p) ? 1 : 0;
}
extern "C" int savfi_conv3x3_tasks_pre_unit16_f32(const float* x, const float* u, const float* bias, float* out, int N, int T, int Ci, int Co,
                                                  int H, int W, int pad, float slope, void* stream) {
  if (!x || !u || !out) return SAVFI_E_NULL;
  WinoPlan p;
  if (int e = check_conv_args(p, N, T, Ci, Co, H, W, pad, 0)) return e;
  if (!unit16_ok(p)) return SAVFI_E_UNSUPPORTED;
  return launch_conv(p, x, u, bias, out, nullptr, N, T, H, W, 0, slope, (hipStream_t)stream, nullptr, 1.f, 1);
}

// The data gradient (mode 1) of savfi_conv3x3_tasks_pre_unit16_f32's layer on a cotangent that is UNIT-MAJOR too: gy[n] is
// [H][W / 16][Co][16] (what savfi_sepconv_bwd_frames8_f32 writes with bit 1 of taps_unit16), gx is [Ci][H+2-2pad][W+2-2pad] as always.
// W % 16 == 0, an even output width, a sample of gy below 2^31 bytes, no reduction split; SAVFI_E_UNSUPPORTED otherwise.
static int in_unit16_ok(const WinoPlan& p, int H, int W) {
  // (F(2x2): a patch row is two 8-byte pairs of an EVEN first column -- pad 0 only; found by tools/r6/wino4_check.py, the plugin's tail is pad 0)
  if (!p.f4 && p.off % 2 != 0) return 0;
  return p.nsplit == 1 && W % 16 == 0 && p.Wo % 2 == 0 && (int64_t)p.K * H * W * 4 < ((int64_t)1 << 31);
}
extern "C" int savfi_conv3x3_in_unit16_supported(int N, int T, int Ci, int Co, int H, int W, int pad) {
  WinoPlan p;
  if (check_conv_args(p, N, T, Ci, Co, H, W, pad, 1)) return 0;
  return in_unit16_ok(p, H, W) ? 1 : 0;
}
extern "C" int savfi_conv3x3_dgrad_in_unit16_f32(const float* gy, const float* u, float* gx, int N, int T, int Ci, int Co, int H, int W,
                                                 int pad, void* stream) {
  if (!gy || !u || !gx) return SAVFI_E_NULL;
  WinoPlan p;
  if (int e = check_conv_args(p, N, T, Ci, Co, H, W, pad, 1)) return e;
  if (!in_unit16_ok(p, H, W)) return SAVFI_E_UNSUPPORTED;
  return launch_conv(p, gy, u, nullptr, gx, nullptr, N, T, H, W, 1, 1.f, (hipStream_t)stream, nullptr, 1.f, 0, 1);
}

// data gradient (mode 1) on a transformed filter with the activation derivative of the layer that produced this convolution's input
// folded into the output stage: gx = dgrad(gy) * (mask > 0 ? 1 : mask_slope), mask [N,Ci,H+2-2pad,W+2-2pad] = the forward input
extern "C" int savfi_conv3x3_dgrad_masked_form_f32(const float* gy, const float* u, const float* mask, float mask_slope, float* gx,
                                                   float* workspace, int N, int T, int Ci, int Co, int H, int W, int pad, int form,
                                                   void* stream) {
  if (!gy || !u || !gx || !mask) return SAVFI_E_NULL;
  if (form != 0 && form != 2) return SAVFI_E_UNSUPPORTED;
  WinoPlan p;
  if (int e = check_conv_args(p, N, T, Ci, Co, H, W, pad, 1 | form)) return e;
  if (p.partial_floats > 0 && !workspace) return SAVFI_E_NULL;
  return launch_conv(p, gy, u, nullptr, gx, workspace, N, T, H, W, 1, 1.f, (hipStream_t)stream, mask, mask_slope);
}
extern "C" int savfi_conv3x3_dgrad_masked_f32(const float* gy, const float* u, const float* mask, float mask_slope, float* gx,
                                              float* workspace, int N, int T, int Ci, int Co, int H, int W, int pad, void* stream) {
  return savfi_conv3x3_dgrad_masked_form_f32(gy, u, mask, mask_slope, gx, workspace, N, T, Ci, Co, H, W, pad, 0, stream);
}

extern "C" int savfi_conv3x3_f32(const float* x, const float* w, const float* bias, float* out, float* workspace,
                                 int N, int Ci, int Co, int H, int W, int pad, int mode, float slope, void* stream) {
  return savfi_conv3x3_tasks_f32(x, w, bias, out, workspace, N, 1, Ci, Co, H, W, pad, mode, slope, stream);
}
