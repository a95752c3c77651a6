// Weight gradient of the 3x3 / stride 1 convolutions (zero padding 0 or 1) for gfx950, NCHW in and out, on the exact-fp32
// matrix cores (v_mfma_f32_16x16x4_f32), deterministic.
//
//   gw[co][ci][a][b] = sum over n, y, x of  gz[n][co][y][x] * x[n][ci][y + a - pad][x + b - pad]
//
// Replaces MIOpen's implicit-GEMM weight-gradient kernels under `aten::convolution_backward` for the backbones' 3x3
// layers (model_utils.py:308-366 MetaConv2dLayer -> F.conv2d; sepconv/model.py:172-245): those want NHWC, so every
// call pays two layout transposes (input and cotangent) plus a zero fill and accumulates with atomics (round-1
// profile: 14 % of the inner step in the kernels + 6 % in the transposes; not run-to-run reproducible).
//
// GEMM view: M = co, N = (ci, tap), K = pixels.  One MFMA takes 4 consecutive pixels of a row as its k-step:
//   A[i][k] = gz[co0 + i][y][x + k],   B[k][j] = x[ci0 + j][y + a - pad][x + k + b - pad]   for one tap (a, b)
// Workgroup = 256 threads = 4 waves (two workgroups per CU); it owns 32 output channels x 32 input channels x 9 taps
// (2 x 2 x 9 accumulator tiles = 144 registers per lane) over a strip of R rows x 64 columns of one image.  Per row the
// cotangent row [32][64] and the three input rows [3][32][66] are staged in LDS (double buffered, one barrier per row,
// pitch 68 so that both fragment reads are bank-conflict free); wave w multiplies pixels [16w, 16w+16): 4 k-steps x 36
// MFMAs against 14 LDS reads (b128 / b64) per row.  The four waves' accumulators are added through LDS in a fixed order, every
// workgroup writes one partial block, and wgrad_reduce adds the partial blocks in a fixed order (no atomics).
#include "common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int GNT = 256;      // threads
constexpr int GCO = 32, GCI = 32, GSEG = 64;
constexpr int GP = 68;        // LDS pitch of a channel row: 64 pixels (+2 halo) padded so that lane (i, k) -> bank 4i + k
constexpr int G_FLOATS = GCO * GP;            // cotangent row
constexpr int X_FLOATS = 3 * GCI * GP;        // three input rows
constexpr int BUF_FLOATS = G_FLOATS + X_FLOATS;
constexpr int TILE_FLOATS = 2 * 2 * 9 * 256;  // one wave's accumulators (36 tiles x 64 lanes x 4)
constexpr int LDS_FLOATS_G = (2 * BUF_FLOATS > TILE_FLOATS) ? 2 * BUF_FLOATS : TILE_FLOATS;

struct WgradArgs {
  const float* x;     // [N][Ci][H][W]
  const float* gz;    // [N][Co][Ho][Wo]
  float* partial;     // [splits][cobs * cibs][TILE_FLOATS]
  int Ci, Co, H, W, Ho, Wo, pad, rows, nseg, nrowchunk, cibs;
  int T;              // tasks (blockIdx.z): task t reduces over samples n' * T + t and writes its own partial blocks
};

// Staging of one row step: cotangent row y ([32][64]) and input rows y - pad .. y - pad + 2 ([3][32][66]).  Wave w
// stages channels w, w+4, ..., w+28 with lane = pixel, so every address is a wave-uniform base (channel plane + row, SALU)
// plus one per-lane column offset, and every LDS write is a conflict-free row.  The loads go to registers first (issued
// before the row's MFMAs; unconditional - a `cond ? load : 0` makes the compiler put every load under its own branch +
// wait) and are written to the other LDS buffer afterwards.
struct Staged {
  float g[8], x[3][8], halo;
};

// Everything outside the tensors is zeroed BY THE LOADS: a row or channel that does not exist gets the wave-uniform offset
// ROW_OOR, a column that does not exist the per-lane offset COL_OOR; either puts the address beyond num_records (< 2^31, checked
// by the host) and the hardware returns 0; together they still fit 32 bits.  (The first version loaded clamped addresses and
// zeroed with an integer mask per element while writing to LDS: ~100 VALU per row and wave, paid in matrix-pipe time.)
constexpr unsigned ROW_OOR = 0x80000000u, COL_OOR = 0x7ffffffcu;


struct Cols {
  unsigned gcol, xcol, hcol;     // byte offsets of this lane's cotangent / input / halo column inside a row (COL_OOR: none)
  bool hok;
  int hr, hc;                    // halo element of this thread: input row hr (0..2), channel hc, column 64 + (tid & 1)
};

__device__ __forceinline__ Cols make_cols(const WgradArgs& a, int x0, int tid) {
  Cols c;
  const int lane = tid & 63;
  const int gx = x0 + lane, xx = x0 - a.pad + lane;
  c.gcol = gx < a.Wo ? (unsigned)gx * 4u : COL_OOR;
  c.xcol = (xx >= 0 && xx < a.W) ? (unsigned)xx * 4u : COL_OOR;
  // the 2 halo columns (64, 65) x 32 channels x 3 rows = 192 elements: one per thread of the first 192
  c.hr = tid >> 6;
  c.hc = (tid & 63) >> 1;
  const int hx = x0 - a.pad + 64 + (tid & 1);
  c.hcol = (unsigned)min(max(hx, 0), a.W - 1) * 4u;
  c.hok = tid < 192 && hx >= 0 && hx < a.W;
  return c;
}

// raw buffer loads: one resource per image, a per-lane column offset (VGPR) and a wave-uniform row offset (SGPR) - no
// 64-bit per-load addresses in VGPRs (the accumulators leave ~100 registers for everything else)
__device__ __forceinline__ float bload(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)voff, (int)soff, 0));
}

__device__ __forceinline__ void load_rows(Staged& s, const WgradArgs& a, const Cols& c, __amdgpu_buffer_rsrc_t gr,
                                          __amdgpu_buffer_rsrc_t xr, int co0, int ci0, int y, int w) {
  const bool yok = y < a.Ho;
#pragma unroll
  for (int jc = 0; jc < 8; ++jc) {
    const int co = co0 + w + 4 * jc;
    s.g[jc] = bload(gr, c.gcol, (yok && co < a.Co) ? (unsigned)((co * a.Ho + y) * a.Wo) * 4u : ROW_OOR);
  }
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    const int yy = y - a.pad + r;
    const bool rok = yok && yy >= 0 && yy < a.H;
#pragma unroll
    for (int jc = 0; jc < 8; ++jc) {
      const int ci = ci0 + w + 4 * jc;
      s.x[r][jc] = bload(xr, c.xcol, (rok && ci < a.Ci) ? (unsigned)((ci * a.H + yy) * a.W) * 4u : ROW_OOR);
    }
  }
  const int yyh = y - a.pad + min(c.hr, 2), cih = ci0 + c.hc;
  const bool hok = c.hok && yok && yyh >= 0 && yyh < a.H && cih < a.Ci;
  s.halo = bload(xr, hok ? c.hcol + (unsigned)((cih * a.H + yyh) * a.W) * 4u : COL_OOR, 0u);
}

__device__ __forceinline__ void store_rows(float* __restrict__ buf, const Staged& s, const Cols& c, int tid, int w) {
  const int lane = tid & 63;
#pragma unroll
  for (int jc = 0; jc < 8; ++jc) buf[(w + 4 * jc) * GP + lane] = s.g[jc];
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int jc = 0; jc < 8; ++jc) buf[G_FLOATS + (r * GCI + w + 4 * jc) * GP + lane] = s.x[r][jc];
  if (tid < 192) buf[G_FLOATS + (c.hr * GCI + c.hc) * GP + 64 + (tid & 1)] = s.halo;
}

__global__ __launch_bounds__(GNT, 2) void wgrad3x3(WgradArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int i = lane & 15, k = lane >> 4;     // A fragment: (channel i, pixel k);  B fragment: (pixel k, channel i)

  // spatial strip: blockIdx.x = (n * nrowchunk + rowchunk) * nseg + seg;  channel tile: blockIdx.y = cob * cibs + cib
  const int seg = blockIdx.x % a.nseg, rc = (blockIdx.x / a.nseg) % a.nrowchunk;
  const int n = (blockIdx.x / (a.nseg * a.nrowchunk)) * a.T + blockIdx.z;
  const int cob = blockIdx.y / a.cibs, cib = blockIdx.y - cob * a.cibs;
  const int co0 = cob * GCO, ci0 = cib * GCI, x0 = seg * GSEG;
  const int y0 = rc * a.rows, y1 = min(y0 + a.rows, a.Ho);
  // per-image buffer resources (host checks that an image's channels fit 2^31 bytes)
  const __amdgpu_buffer_rsrc_t gr = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(a.gz + (size_t)n * a.Co * a.Ho * a.Wo), 0, a.Co * a.Ho * a.Wo * 4, 0x00020000);
  const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(a.x + (size_t)n * a.Ci * a.H * a.W), 0, a.Ci * a.H * a.W * 4, 0x00020000);

  f32x4 acc[2][2][9];
#pragma unroll
  for (int rb = 0; rb < 2; ++rb)
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
      for (int t = 0; t < 9; ++t) acc[rb][cb][t] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const Cols cols = make_cols(a, x0, tid);
  Staged st;
  load_rows(st, a, cols, gr, xr, co0, ci0, y0, w);
  store_rows(lds, st, cols, tid, w);
  __syncthreads();
  for (int y = y0; y < y1; ++y) {
    const float* cur = lds + ((y - y0) & 1) * BUF_FLOATS;
    float* nxt = lds + ((y - y0 + 1) & 1) * BUF_FLOATS;
    load_rows(st, a, cols, gr, xr, co0, ci0, y + 1, w);       // next row: in flight during this row's MFMAs
    __builtin_amdgcn_sched_barrier(0);                       // ... which the scheduler otherwise sinks behind ~120 of the 144
    // k-slot (step ks, lane group k) takes pixel 16 w + 4 k + ks -- any assignment works as long as A and B agree -- so that a
    // lane's four steps are CONSECUTIVE pixels: the A values of a row block are one ds_read_b128, the B values of an input row
    // block (pixels 4k .. 4k + 5: three column taps) one b128 + one b64, 14 LDS reads per row where the ks-major assignment
    // needed 80 dwords (44 read2).  Pitch 68: lane (i, k) starts at bank 4 i + const, a quarter wave covers all 64 banks.
    const float* G = cur + i * GP + 16 * w + 4 * k;                  // A: G[rb * 16 + i][16 w + 4 k + ks]
    const float* X = cur + G_FLOATS + i * GP + 16 * w + 4 * k;       // B: X[r][cb * 16 + i][16 w + 4 k + ks + b]
    const f32x4 ga0 = *reinterpret_cast<const f32x4*>(G), ga1 = *reinterpret_cast<const f32x4*>(G + 16 * GP);
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      float xs[2][6];
#pragma unroll
      for (int cb = 0; cb < 2; ++cb) {
        const float* xp = X + (r * GCI + 16 * cb) * GP;
        const f32x4 lo = *reinterpret_cast<const f32x4*>(xp);
        const float2 hi = *reinterpret_cast<const float2*>(xp + 4);
        xs[cb][0] = lo[0]; xs[cb][1] = lo[1]; xs[cb][2] = lo[2]; xs[cb][3] = lo[3]; xs[cb][4] = hi.x; xs[cb][5] = hi.y;
      }
#pragma unroll
      for (int ks = 0; ks < 4; ++ks)
#pragma unroll
        for (int b = 0; b < 3; ++b) {
          const int t = r * 3 + b;
          acc[0][0][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(ga0[ks], xs[0][ks + b], acc[0][0][t], 0, 0, 0);
          acc[0][1][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(ga0[ks], xs[1][ks + b], acc[0][1][t], 0, 0, 0);
          acc[1][0][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(ga1[ks], xs[0][ks + b], acc[1][0][t], 0, 0, 0);
          acc[1][1][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(ga1[ks], xs[1][ks + b], acc[1][1][t], 0, 0, 0);
        }
    }
    store_rows(nxt, st, cols, tid, w);
    __syncthreads();
  }

  // add the four waves' accumulators in wave order through LDS; wave 3 writes the workgroup's partial block
  float* red = lds + lane * 4;
  for (int turn = 0; turn < 4; ++turn) {
    if (w == turn) {
#pragma unroll
      for (int rb = 0; rb < 2; ++rb)
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
          for (int t = 0; t < 9; ++t) {
            float* p = red + ((rb * 2 + cb) * 9 + t) * 256;
            f32x4 v = acc[rb][cb][t];
            if (turn > 0) v += *reinterpret_cast<const f32x4*>(p);
            if (turn < 3) *reinterpret_cast<f32x4*>(p) = v;
            else acc[rb][cb][t] = v;
          }
    }
    __syncthreads();
  }
  if (w == 3) {
    float* out = a.partial + (((size_t)blockIdx.z * gridDim.x + blockIdx.x) * gridDim.y + blockIdx.y) * TILE_FLOATS + lane * 4;
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
      for (int cb = 0; cb < 2; ++cb)
#pragma unroll
        for (int t = 0; t < 9; ++t) *reinterpret_cast<f32x4*>(out + ((rb * 2 + cb) * 9 + t) * 256) = acc[rb][cb][t];
  }
}

// Two-level, fixed-order reduction of the partial blocks (a single pass with one thread per weight would run 9216
// threads x up to 1024 dependent reads: 300 us).
// level 1: stage2[g][tile][e] = sum over the splits s in group g (RG consecutive splits) of partial[s][tile][e]   (float4 lanes)
constexpr int RG = 16;
__global__ __launch_bounds__(256) void wgrad_reduce1(const float* __restrict__ partial, float* __restrict__ stage2,
                                                     size_t block_floats, int nsplit) {
  const size_t e4 = (size_t)blockIdx.x * 256 + threadIdx.x;         // float4 index inside one split's block
  if (e4 * 4 >= block_floats) return;
  partial += (size_t)blockIdx.z * nsplit * block_floats;              // blockIdx.z = task
  stage2 += (size_t)blockIdx.z * gridDim.y * block_floats;
  const int s0 = blockIdx.y * RG, s1 = min(s0 + RG, nsplit);
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (int sp = s0; sp < s1; ++sp) acc += *reinterpret_cast<const f32x4*>(partial + (size_t)sp * block_floats + e4 * 4);
  *reinterpret_cast<f32x4*>(stage2 + (size_t)blockIdx.y * block_floats + e4 * 4) = acc;
}

// level 2: gw[co][ci][tap] = sum over the groups (fixed order) of stage2[g][cob * cibs + cib][(rb, cb, tap)][lane][reg]
// accumulator tile layout: row (co within the block of 16) = 4 * (lane >> 4) + reg, column (ci within 16) = lane & 15
__global__ __launch_bounds__(256) void wgrad_reduce2(const float* __restrict__ stage2, float* __restrict__ gw, int Co, int Ci,
                                                     int cibs, int ntiles, int ngroups) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= Co * Ci * 9) return;
  stage2 += (size_t)blockIdx.y * ngroups * ntiles * TILE_FLOATS;      // blockIdx.y = task
  gw += (size_t)blockIdx.y * Co * Ci * 9;
  const int tap = e % 9, ci = (e / 9) % Ci, co = e / (9 * Ci);
  const int cob = co >> 5, cib = ci >> 5, rb = (co >> 4) & 1, cb = (ci >> 4) & 1;
  const int row = co & 15, col = ci & 15, lane = (row >> 2) * 16 + col, reg = row & 3;
  const size_t off = (size_t)(cob * cibs + cib) * TILE_FLOATS + ((rb * 2 + cb) * 9 + tap) * 256 + lane * 4 + reg;
  float s = 0.f;
  for (int g = 0; g < ngroups; ++g) s += stage2[(size_t)g * ntiles * TILE_FLOATS + off];
  gw[e] = s;
}

struct WgradPlan {
  int Ho, Wo, cobs, cibs, nseg, rows, nrowchunk, ngroups;
  int64_t splits, partial_floats, stage2_floats;
};

bool wgrad_plan(WgradPlan& p, int N, int T, int Ci, int Co, int H, int W, int pad) {
  p.Ho = H + 2 * pad - 2;
  p.Wo = W + 2 * pad - 2;
  if (p.Ho <= 0 || p.Wo <= 0) return false;
  p.cobs = savfi_cdiv(Co, GCO);
  p.cibs = savfi_cdiv(Ci, GCI);
  p.nseg = savfi_cdiv(p.Wo, GSEG);
  // Rows per workgroup: 512 workgroup slots (2 per CU), so a launch takes ceil(workgroups / 512) rounds of about
  // (rows + 3) row times (3 ~ prologue, cross-wave reduction and the 37 KB partial block that is written and read again);
  // pick the strip height that minimises it (e.g. 192->192 at 96x160, N=2: 2 chunks = 432 workgroups in one round instead
  // of 3 chunks = 648 in two).
  const int64_t per_row = (int64_t)N * p.nseg * p.cobs * p.cibs;
  int best_rows = p.Ho;
  double best_cost = -1.0;
  for (int chunks = 1; chunks <= 128; ++chunks) {
    const int rows = savfi_cdiv(p.Ho, chunks);
    if (rows < 4 && chunks > 1) break;
    const int64_t wgs = per_row * savfi_cdiv(p.Ho, rows);
    const double cost = (double)((wgs + 511) / 512) * (rows + 3) + 0.004 * (double)wgs;
    if (best_cost < 0 || cost < best_cost) { best_cost = cost; best_rows = rows; }
  }
  p.rows = best_rows;
  p.nrowchunk = savfi_cdiv(p.Ho, best_rows);
  // per task: N / T samples (the cost model above counts the workgroups of all tasks: they share the launch)
  p.splits = (int64_t)(N / T) * p.nrowchunk * p.nseg;
  p.partial_floats = (int64_t)T * p.splits * p.cobs * p.cibs * TILE_FLOATS;
  p.ngroups = (int)((p.splits + RG - 1) / RG);
  p.stage2_floats = (int64_t)T * p.ngroups * p.cobs * p.cibs * TILE_FLOATS;
  return true;
}

}  // namespace

extern "C" int64_t savfi_conv3x3_wgrad_tasks_workspace_floats(int N, int T, int Ci, int Co, int H, int W, int pad) {
  if (N <= 0 || T <= 0 || N % T != 0 || Ci <= 0 || Co <= 0 || H <= 0 || W <= 0) return SAVFI_E_SHAPE;
  if (pad != 0 && pad != 1) return SAVFI_E_UNSUPPORTED;
  WgradPlan p;
  if (!wgrad_plan(p, N, T, Ci, Co, H, W, pad)) return SAVFI_E_SHAPE;
  return p.partial_floats + p.stage2_floats;
}

extern "C" int64_t savfi_conv3x3_wgrad_workspace_floats(int N, int Ci, int Co, int H, int W, int pad) {
  return savfi_conv3x3_wgrad_tasks_workspace_floats(N, 1, Ci, Co, H, W, pad);
}

// gw[t] = weight gradient over the samples n with n % T == t      gw [T][Co][Ci][3][3]
extern "C" int savfi_conv3x3_wgrad_tasks_f32(const float* x, const float* gz, float* gw, float* workspace, int N, int T, int Ci,
                                             int Co, int H, int W, int pad, void* stream) {
  if (!x || !gz || !gw || !workspace) return SAVFI_E_NULL;
  if (N <= 0 || T <= 0 || N % T != 0 || Ci <= 0 || Co <= 0 || H <= 0 || W <= 0) return SAVFI_E_SHAPE;
  if (pad != 0 && pad != 1) return SAVFI_E_UNSUPPORTED;
  WgradPlan p;
  if (!wgrad_plan(p, N, T, Ci, Co, H, W, pad)) return SAVFI_E_SHAPE;
  if ((int64_t)Ci * H * W >= ((int64_t)1 << 29) || (int64_t)Co * p.Ho * p.Wo >= ((int64_t)1 << 29)) return SAVFI_E_TOOBIG;   // 32-bit byte offsets
  if (p.ngroups > 65535 || p.splits > 0x7fffffffLL || (int64_t)p.cobs * p.cibs > 65535 || (int64_t)Co * Ci * 9 > 0x7fffffffLL || T > 65535) return SAVFI_E_TOOBIG;
  hipStream_t st = (hipStream_t)stream;
  constexpr size_t lds = (size_t)LDS_FLOATS_G * sizeof(float);
  static uint32_t attr_done = 0;
  if (int e = savfi_ensure_dynamic_lds((const void*)wgrad3x3, lds, attr_done)) return e;
  WgradArgs a{x, gz, workspace, Ci, Co, H, W, p.Ho, p.Wo, pad, p.rows, p.nseg, p.nrowchunk, p.cibs, T};
  hipLaunchKernelGGL(wgrad3x3, dim3((unsigned)p.splits, p.cobs * p.cibs, T), dim3(GNT), lds, st, a);
  if (int e = savfi_launch_status()) return e;
  const int ntiles = p.cobs * p.cibs;
  const size_t block_floats = (size_t)ntiles * TILE_FLOATS;
  float* stage2 = workspace + p.partial_floats;
  if (p.ngroups == 1) {
    // at most RG splits: the blocks of the splits are laid out like the blocks of the groups, and level 2 adds them in the order level 1
    // would have (0 + p0 + p1 ...): one launch instead of two, the same bits (round 5: 276 reduction launches per C2 iteration)
    hipLaunchKernelGGL(wgrad_reduce2, dim3(savfi_cdiv(Co * Ci * 9, 256), T), dim3(256), 0, st, workspace, gw, Co, Ci, p.cibs, ntiles,
                       (int)p.splits);
    return savfi_launch_status();
  }
  hipLaunchKernelGGL(wgrad_reduce1, dim3((unsigned)((block_floats / 4 + 255) / 256), p.ngroups, T), dim3(256), 0, st, workspace,
                     stage2, block_floats, (int)p.splits);
  if (int e = savfi_launch_status()) return e;
  hipLaunchKernelGGL(wgrad_reduce2, dim3(savfi_cdiv(Co * Ci * 9, 256), T), dim3(256), 0, st, stage2, gw, Co, Ci, p.cibs, ntiles,
                     p.ngroups);
  return savfi_launch_status();
}

extern "C" int savfi_conv3x3_wgrad_f32(const float* x, const float* gz, float* gw, float* workspace, int N, int Ci, int Co,
                                       int H, int W, int pad, void* stream) {
  return savfi_conv3x3_wgrad_tasks_f32(x, gz, gw, workspace, N, 1, Ci, Co, H, W, pad, stream);
}

// ------------------------------------------------------------------------------------------------------------------
// Winograd form of the same weight gradient: F(3x3, 2x2).  Per 2x2 tile of the cotangent and its 4x4 input patch
//   dW[co][ci] += A'^T [ (G' g G'^T) (.) (B^T d B) ] A'        B^T as in winograd.hip (same interpolation points),
//   G' = [1 0; .5 .5; .5 -.5; 0 1],   A'^T = [1 1 1 0; 0 1 -1 0; 0 1 1 -1]
// so the sum over tiles is 16 GEMMs  M[xi][co][ci] = sum_tile Y[xi][co][tile] V[xi][ci][tile]  with 2.25x fewer multiplies than
// the direct form above; the output transform runs once per workgroup.  The factors .5 of G' are applied to M in the output
// stage (M is linear in Y), so the cotangent transform is 12 additions.
//
// Workgroup = 4 waves = 32 co x 32 ci of one task over a range of tile chunks (8 consecutive tiles of a tile row); wave w owns
// the Winograd row r = w (xi = 4 w .. 4 w + 3): 4 x 2 x 2 accumulator tiles = 64 registers.  Per chunk every thread transforms
// one input patch (ci = tid / 8, tile = tid % 8) and one cotangent tile (co = tid / 8) and writes each as four ds_write_b128
// ([r][channel][tile][c], pitch 36 floats per channel: a quarter wave reads 16 channels x 16 bytes over all 64 banks); the MFMA
// k-step is 4 tiles.  LDS is double buffered (one barrier per chunk), the next chunk's loads are in flight during the MFMAs.
// ------------------------------------------------------------------------------------------------------------------
namespace {

typedef int i32x4w __attribute__((ext_vector_type(4)));
typedef float f32x2w __attribute__((ext_vector_type(2)));

constexpr int WW_TK = 8;                  // tiles per chunk (two MFMA k-steps)
constexpr int WW_PY = WW_TK * 4 + 4;      // floats per channel row of a buffer: [tile][c] + pad
constexpr int WW_CB = 32;                 // channels per block (co and ci)
constexpr int WW_HALF = 4 * WW_CB * WW_PY;        // floats of Y (or V) in one buffer: [r][channel][tile][c]
constexpr int WW_BUF = 2 * WW_HALF;               // Y + V
constexpr int WW_BLOCK_FLOATS = WW_CB * WW_CB * 9;  // one workgroup's partial result [co][ci][3][3]
#ifndef WW_DOUBLE
#define WW_DOUBLE 0
#endif
// single buffered (two barriers per chunk): 36.9 KB and ~120 VGPRs -> FOUR workgroups per CU, which hides the barriers and the
// LDS / memory latencies of a chunk better than a second buffer does with two
constexpr int WW_LDS_FLOATS = (WW_DOUBLE ? 2 : 1) * WW_BUF;

struct WWArgs {
  const float* x;      // [N][Ci][H][W]
  const float* gz;     // [N][Co][Ho][Wo]
  float* partial;      // [T][nsplit][cobs * cibs][32][32][9]
  int Ci, Co, H, W, Ho, Wo, pad, T, cibs;
  int ty_cnt, tx_cnt, cpr;        // tile rows, tile columns, chunks per tile row
  int chunks_total;               // per task: (N / T) * ty_cnt * cpr
  int chunks_per_split;
  float* bias_partial;            // or null: [T][nsplit][cobs * 32] sums of the cotangent per produced channel (round 5: the bias gradient
                                  // rides on the weight gradient's read of gz instead of a pass of its own over the map)
};

}  // namespace

__device__ f32x4 savfi_wg_buffer_load_x4(i32x4w rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.load.v4f32");
__device__ f32x2w savfi_wg_buffer_load_x2(i32x4w rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.load.v2f32");

namespace {

__device__ __forceinline__ i32x4w ww_rsrc(const float* base, unsigned bytes) {
  const unsigned long long p = reinterpret_cast<unsigned long long>(base);
  i32x4w r;
  r.x = (int)(unsigned)p; r.y = (int)(unsigned)(p >> 32); r.z = (int)bytes; r.w = 0x00020000;
  return r;
}

struct WWChunk {          // what a thread holds of one chunk between its loads and its transforms
  float d[16];            // input patch (row-major 4x4)
  float g[4];             // cotangent tile (row-major 2x2)
  int cx;                 // wave-uniform: chunk column (first chunk of a padded row: lane tile 0 was loaded from column 0 instead
                          // of -1 -> shift right; last chunk of a row: columns beyond the image are cleared)
};

// Position in the chunk sequence: sample, tile row and chunk column are wave-uniform, and so are the byte offsets of the four
// patch rows and the two cotangent rows inside a channel plane (SGPRs: they go into the loads' scalar offset); the channel plane
// of a lane is a constant per-lane offset.  A row outside the map / a channel beyond Ci or Co is marked WW_OOR in its part of
// the offset: either part alone puts the address beyond num_records (the host keeps a sample below 2^30 bytes for this
// kernel), both together still fit 32 bits.
constexpr unsigned WW_OOR = 0x40000000u;
struct WWPos {
  int nq, ty, cx;
  unsigned xrow[4], grow[2];
};

__device__ __forceinline__ void ww_set_row(WWPos& p, const WWArgs& a) {
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int y = 2 * p.ty - a.pad + r;
    p.xrow[r] = (y >= 0 && y < a.H) ? (unsigned)(y * a.W) * 4u : WW_OOR;
  }
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int y = 2 * p.ty + u;
    p.grow[u] = y < a.Ho ? (unsigned)(y * a.Wo) * 4u : WW_OOR;
  }
}

__device__ __forceinline__ void ww_load(WWChunk& c, const WWArgs& a, const WWPos& p, int task, int tl, unsigned xplane, unsigned gplane) {
  const int n = p.nq * a.T + task;
  const int tx = p.cx * WW_TK + tl;
  // input patch: rows 2 ty - pad + (0..3), columns x0 .. x0 + 3; rows outside the image and channels beyond Ci come back as zeros
  const i32x4w xr = ww_rsrc(a.x + (size_t)n * a.Ci * a.H * a.W, (unsigned)(a.Ci * a.H * a.W) * 4u);
  const int x0 = 2 * tx - a.pad;
  c.cx = p.cx;
  const unsigned xoff = xplane + (unsigned)max(x0, 0) * 4u;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const f32x4 v = savfi_wg_buffer_load_x4(xr, (int)xoff, (int)p.xrow[r], 0);
    c.d[4 * r] = v.x; c.d[4 * r + 1] = v.y; c.d[4 * r + 2] = v.z; c.d[4 * r + 3] = v.w;
  }
  // cotangent tile: rows 2 ty, 2 ty + 1, columns 2 tx, 2 tx + 1 (a tile beyond the last column: out of range -> zeros)
  const i32x4w gr = ww_rsrc(a.gz + (size_t)n * a.Co * a.Ho * a.Wo, (unsigned)(a.Co * a.Ho * a.Wo) * 4u);
  const unsigned goff = 2 * tx < a.Wo ? gplane + (unsigned)(2 * tx) * 4u : WW_OOR;
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const f32x2w v = savfi_wg_buffer_load_x2(gr, (int)goff, (int)p.grow[u], 0);
    c.g[2 * u] = v.x; c.g[2 * u + 1] = v.y;
  }
}

// registers -> transforms -> this thread's rows of the V and Y buffers
// bias_acc: or null -- this thread's slot of the workgroup's channel sums in LDS (the 128-register budget of four workgroups per CU has no
// room for one more live value across the MFMA loop: a register accumulator spilled an accumulator tile to scratch inside the loop)
__device__ __forceinline__ void ww_transform_store(WWChunk& c, const WWArgs& a, float* __restrict__ ybuf, float* __restrict__ vbuf, int ch, int tl,
                                                   float* __restrict__ bias_acc) {
  float (&d)[16] = c.d;
  if (c.cx == 0 && a.pad > 0) {     // wave-uniform: only the first chunk of a tile row
    if (tl == 0) {
#pragma unroll
      for (int r = 0; r < 4; ++r) { d[4 * r + 3] = d[4 * r + 2]; d[4 * r + 2] = d[4 * r + 1]; d[4 * r + 1] = d[4 * r]; d[4 * r] = 0.f; }
    }
  }
  if (c.cx == a.cpr - 1) {          // wave-uniform: only the last chunk of a tile row
    const int tx = c.cx * WW_TK + tl;
    const int wcols = a.W - (2 * tx - a.pad), gcols = a.Wo - 2 * tx;      // columns of the patch / cotangent tile inside the map
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      d[4 * r] = wcols > 0 ? d[4 * r] : 0.f;
      d[4 * r + 1] = wcols > 1 ? d[4 * r + 1] : 0.f;
      d[4 * r + 2] = wcols > 2 ? d[4 * r + 2] : 0.f;
      d[4 * r + 3] = wcols > 3 ? d[4 * r + 3] : 0.f;
    }
    c.g[1] = gcols > 1 ? c.g[1] : 0.f;
    c.g[3] = gcols > 1 ? c.g[3] : 0.f;
  }
  // V = B^T d B,  B^T = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 1 0 -1]
  float t[16];
#pragma unroll
  for (int cc = 0; cc < 4; ++cc) {
    t[cc] = d[cc] - d[8 + cc];
    t[4 + cc] = d[4 + cc] + d[8 + cc];
    t[8 + cc] = d[8 + cc] - d[4 + cc];
    t[12 + cc] = d[4 + cc] - d[12 + cc];
  }
#pragma unroll
  for (int r = 0; r < 4; ++r)
    *reinterpret_cast<f32x4*>(vbuf + (r * WW_CB + ch) * WW_PY + 4 * tl) =
        (f32x4){t[4 * r] - t[4 * r + 2], t[4 * r + 1] + t[4 * r + 2], t[4 * r + 2] - t[4 * r + 1], t[4 * r + 1] - t[4 * r + 3]};
  // Y' = G'' g G''^T with G'' = [1 0; 1 1; 1 -1; 0 1] (the factors .5 of rows / columns 1, 2 are applied in the output stage)
  const float g00 = c.g[0], g01 = c.g[1], g10 = c.g[2], g11 = c.g[3];
  if (bias_acc) *bias_acc += (g00 + g01) + (g10 + g11);       // (cotangent outside the map / beyond Co is zero here)
  const float s0[2] = {g00, g01}, s1[2] = {g00 + g10, g01 + g11}, s2[2] = {g00 - g10, g01 - g11}, s3[2] = {g10, g11};
  *reinterpret_cast<f32x4*>(ybuf + (0 * WW_CB + ch) * WW_PY + 4 * tl) = (f32x4){s0[0], s0[0] + s0[1], s0[0] - s0[1], s0[1]};
  *reinterpret_cast<f32x4*>(ybuf + (1 * WW_CB + ch) * WW_PY + 4 * tl) = (f32x4){s1[0], s1[0] + s1[1], s1[0] - s1[1], s1[1]};
  *reinterpret_cast<f32x4*>(ybuf + (2 * WW_CB + ch) * WW_PY + 4 * tl) = (f32x4){s2[0], s2[0] + s2[1], s2[0] - s2[1], s2[1]};
  *reinterpret_cast<f32x4*>(ybuf + (3 * WW_CB + ch) * WW_PY + 4 * tl) = (f32x4){s3[0], s3[0] + s3[1], s3[0] - s3[1], s3[1]};
}

template <bool BIAS>
__global__ __launch_bounds__(256, WW_DOUBLE ? 2 : 4) void wino_wgrad3x3(WWArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tl = tid & 7, ch = tid >> 3;
  const int task = blockIdx.z;
  const int cob = blockIdx.y / a.cibs, cib = blockIdx.y - cob * a.cibs;
  const int q0 = blockIdx.x * a.chunks_per_split, q1 = min(q0 + a.chunks_per_split, a.chunks_total);
  const int ci = cib * WW_CB + ch, co = cob * WW_CB + ch;

  f32x4 acc[4][2][2];
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[c][i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // fragment reads: row = channel lane % 16 of the co / ci tile, k = tile 4 ks + lane / 16, the four xi columns in one b128
  const int frag = (w * WW_CB + (lane & 15)) * WW_PY + 4 * (lane >> 4);
  WWChunk cur;
  WWPos pos;
  {
    const int per_img = a.ty_cnt * a.cpr;
    pos.nq = q0 / per_img;
    const int rem = q0 - pos.nq * per_img;
    pos.ty = rem / a.cpr;
    pos.cx = rem - pos.ty * a.cpr;
  }
  // this lane's channel planes inside a sample (constant for the whole kernel)
  const unsigned xplane = ci < a.Ci ? (unsigned)(ci * a.H * a.W) * 4u : WW_OOR;
  const unsigned gplane = co < a.Co ? (unsigned)(co * a.Ho * a.Wo) * 4u : WW_OOR;
  ww_set_row(pos, a);
  if (q0 < q1) ww_load(cur, a, pos, task, tl, xplane, gplane);
  // BIAS: this thread's tiles of produced channel co, summed in LDS behind the chunk buffers (first ci block of every co block only)
  // (the slot's address is recomputed from the thread index where it is used: a pointer kept across the loop was spilled)
  const bool do_bias = BIAS && cib == 0;
  auto bias_slot = [&]() { return lds + WW_LDS_FLOATS + threadIdx.x; };
  if (do_bias) *bias_slot() = 0.f;
  for (int q = q0; q < q1; ++q) {
    float* buf = lds + (WW_DOUBLE ? ((q - q0) & 1) * WW_BUF : 0);
    if (!WW_DOUBLE && q != q0) __syncthreads();          // every wave is done with the previous chunk's fragments
    ww_transform_store(cur, a, buf, buf + WW_HALF, ch, tl, do_bias ? bias_slot() : nullptr);
    if (q + 1 < q1) {                                  // next chunk: in flight during this chunk's MFMAs
      if (++pos.cx == a.cpr) {                         // wave-uniform
        pos.cx = 0;
        if (++pos.ty == a.ty_cnt) { pos.ty = 0; ++pos.nq; }
        ww_set_row(pos, a);
      }
      ww_load(cur, a, pos, task, tl, xplane, gplane);
    }
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();
    const float* yb = buf + frag;
    const float* vb = buf + WW_HALF + frag;
#pragma unroll
    for (int ks = 0; ks < WW_TK / 4; ++ks) {
      f32x4 ya[2], va[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        ya[i] = *reinterpret_cast<const f32x4*>(yb + 16 * i * WW_PY + 16 * ks);
        va[i] = *reinterpret_cast<const f32x4*>(vb + 16 * i * WW_PY + 16 * ks);
      }
#pragma unroll
      for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[c][i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(ya[i][c], va[j][c], acc[c][i][j], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);       // one k-step's fragments at a time: 16 registers, not 32 (no spills at 128 VGPRs)
    }
  }
  __syncthreads();
  if (do_bias) {                                       // workgroup-uniform: the first ci block of every co block hands out the channel sums
    auto dpp = [](float v, auto ctrl) {
      return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), decltype(ctrl)::value, 0xf, 0xf, true));
    };
    float bsum = *bias_slot();                          // the eight tile lanes of a channel: lanes 8 ch' .. 8 ch' + 7
    bsum += dpp(bsum, std::integral_constant<int, 0xB1>{});      // quad_perm [1,0,3,2]
    bsum += dpp(bsum, std::integral_constant<int, 0x4E>{});      // quad_perm [2,3,0,1]
    bsum += dpp(bsum, std::integral_constant<int, 0x141>{});     // row_half_mirror
    if (tl == 0) a.bias_partial[(((size_t)task * gridDim.x + blockIdx.x) * (gridDim.y / a.cibs) + cob) * WW_CB + ch] = bsum;
  }

  // output stage.  Column pass in registers (this wave holds row r = w, scale s_r s_c with s = 1, .5, .5, 1):
  //   u[b] = sum_c A'^T[b][c] s_c M[r][c]:  u0 = m0 + (m1 + m2)/2,  u1 = (m1 - m2)/2,  u2 = (m1 + m2)/2 - m3;   times s_r
  // rows meet in LDS:  dW[0][b] = u_0 + u_1 + u_2,  dW[1][b] = u_1 - u_2,  dW[2][b] = u_1 + u_2 - u_3
  const float sr = (w == 1 || w == 2) ? 0.5f : 1.f;
  float* ex = lds;      // one co tile at a time: [r][b][ci tile j][lane][reg] = 4 x 3 x 2 x 256 floats = 24 KB
  // accumulator tile layout: row (co within 16) = 4 * (lane >> 4) + reg, column (ci within 16) = lane & 15
  float* out = a.partial + (((size_t)task * gridDim.x + blockIdx.x) * gridDim.y + blockIdx.y) * WW_BLOCK_FLOATS;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      f32x4 u0, u1, u2;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float m0 = acc[0][i][j][e], m1 = acc[1][i][j][e], m2 = acc[2][i][j][e], m3 = acc[3][i][j][e];
        const float h = 0.5f * (m1 + m2);
        u0[e] = sr * (m0 + h);
        u1[e] = sr * (0.5f * (m1 - m2));
        u2[e] = sr * (h - m3);
      }
      float* p = ex + ((w * 3) * 2 + j) * 256 + lane * 4;
      *reinterpret_cast<f32x4*>(p) = u0;
      *reinterpret_cast<f32x4*>(p + 2 * 256) = u1;
      *reinterpret_cast<f32x4*>(p + 4 * 256) = u2;
    }
    __syncthreads();
    for (int item = tid; item < 3 * 2 * 256; item += 256) {         // (b, ci tile, lane, reg)
      const int reg = item & 3, ln = (item >> 2) & 63, j = (item >> 8) & 1, b = item >> 9;
      const float* p = ex + (b * 2 + j) * 256 + ln * 4 + reg;
      const float u_0 = p[0], u_1 = p[3 * 2 * 256], u_2 = p[2 * 3 * 2 * 256], u_3 = p[3 * 3 * 2 * 256];
      const int cor = 16 * i + 4 * (ln >> 4) + reg, cic = 16 * j + (ln & 15);
      float* o = out + (cor * WW_CB + cic) * 9 + b;
      o[0] = u_0 + u_1 + u_2;
      o[3] = u_1 - u_2;
      o[6] = u_1 + u_2 - u_3;
    }
    __syncthreads();
  }
}

// gw[t][co][ci][k] = sum over the groups (fixed order) of stage2[t][g][cob * cibs + cib][co % 32][ci % 32][k]
__global__ __launch_bounds__(256) void wino_wgrad_reduce2(const float* __restrict__ stage2, float* __restrict__ gw, int Co, int Ci,
                                                          int cibs, int ntiles, int ngroups) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= Co * Ci * 9) return;
  stage2 += (size_t)blockIdx.y * ngroups * ntiles * WW_BLOCK_FLOATS;      // blockIdx.y = task
  gw += (size_t)blockIdx.y * Co * Ci * 9;
  const int k = e % 9, ci = (e / 9) % Ci, co = e / (9 * Ci);
  const size_t off = (size_t)((co / WW_CB) * cibs + ci / WW_CB) * WW_BLOCK_FLOATS + ((co % WW_CB) * WW_CB + ci % WW_CB) * 9 + k;
  float s = 0.f;
  for (int g = 0; g < ngroups; ++g) s += stage2[(size_t)g * ntiles * WW_BLOCK_FLOATS + off];
  gw[e] = s;
}

// gb[t][co] = sum over the splits (fixed order) of bias_partial[t][split][co]
__global__ __launch_bounds__(64) void wino_wgrad_bias_reduce(const float* __restrict__ bias_partial, float* __restrict__ gb, int Co, int cobs,
                                                             int nsplit) {
  const int co = blockIdx.x * 64 + threadIdx.x, task = blockIdx.y;
  if (co >= Co) return;
  const float* p = bias_partial + (size_t)task * nsplit * cobs * WW_CB + co;
  float s = 0.f;
  for (int sp = 0; sp < nsplit; ++sp) s += p[(size_t)sp * cobs * WW_CB];
  gb[(size_t)task * Co + co] = s;
}

struct WWPlan {
  int Ho, Wo, cobs, cibs, ty_cnt, tx_cnt, cpr, chunks_total, chunks_per_split, nsplit, ngroups;
  int64_t partial_floats, stage2_floats;
};

bool ww_plan(WWPlan& p, int N, int T, int Ci, int Co, int H, int W, int pad) {
  p.Ho = H + 2 * pad - 2;
  p.Wo = W + 2 * pad - 2;
  if (p.Ho <= 0 || p.Wo <= 0) return false;
  p.cobs = savfi_cdiv(Co, WW_CB);
  p.cibs = savfi_cdiv(Ci, WW_CB);
  p.ty_cnt = savfi_cdiv(p.Ho, 2);
  p.tx_cnt = savfi_cdiv(p.Wo, 2);
  p.cpr = savfi_cdiv(p.tx_cnt, WW_TK);
  const int64_t chunks = (int64_t)(N / T) * p.ty_cnt * p.cpr;
  if (chunks > 0x7fffffffLL) return false;
  p.chunks_total = (int)chunks;
  // splits: enough workgroups for a few rounds of the 512 slots, at least 16 chunks each (output stage + 36 KB partial block; 32 left the
  // small task-batched maps -- 256 -> 256 @24x32, 64 -> 64 @96x128, 128 -> 128 @48x64 at T = 4 -- with a quarter to a third of the slots:
  // 90 / 77 / 76 us -> 74 / 70 / 68 with 16, no further gain below; tools/wino_wgrad_time.py)
  const int64_t blocks = (int64_t)T * p.cobs * p.cibs;
  int64_t want = (512 * 2 + blocks - 1) / blocks;
#ifndef SAVFI_WWGRAD_MIN_CHUNKS
#define SAVFI_WWGRAD_MIN_CHUNKS 16
#endif
  constexpr int min_chunks = SAVFI_WWGRAD_MIN_CHUNKS;
  const int64_t most = chunks / min_chunks > 0 ? chunks / min_chunks : 1;
  if (want > most) want = most;
  if (want < 1) want = 1;
  p.chunks_per_split = savfi_cdiv(chunks, want);
  p.nsplit = savfi_cdiv(chunks, p.chunks_per_split);
  p.ngroups = savfi_cdiv(p.nsplit, RG);
  p.partial_floats = (int64_t)T * p.nsplit * p.cobs * p.cibs * WW_BLOCK_FLOATS;
  p.stage2_floats = (int64_t)T * p.ngroups * p.cobs * p.cibs * WW_BLOCK_FLOATS;
  return true;
}

}  // namespace

extern "C" int64_t savfi_conv3x3_wgrad_wino_tasks_workspace_floats(int N, int T, int Ci, int Co, int H, int W, int pad) {
  if (N <= 0 || T <= 0 || N % T != 0 || Ci <= 0 || Co <= 0 || H <= 0 || W <= 0) return SAVFI_E_SHAPE;
  if (pad != 0 && pad != 1) return SAVFI_E_UNSUPPORTED;
  WWPlan p;
  if (!ww_plan(p, N, T, Ci, Co, H, W, pad)) return SAVFI_E_SHAPE;
  return p.partial_floats + p.stage2_floats;
}

extern "C" int64_t savfi_conv3x3_wgrad_wino_tasks_bias_workspace_floats(int N, int T, int Ci, int Co, int H, int W, int pad) {
  const int64_t base = savfi_conv3x3_wgrad_wino_tasks_workspace_floats(N, T, Ci, Co, H, W, pad);
  if (base < 0) return base;
  WWPlan p;
  ww_plan(p, N, T, Ci, Co, H, W, pad);
  return base + (int64_t)T * p.nsplit * p.cobs * WW_CB;
}

static int wino_wgrad_tasks(const float* x, const float* gz, float* gw, float* gb, float* workspace, int N, int T, int Ci, int Co, int H,
                            int W, int pad, void* stream);

// Same contract as savfi_conv3x3_wgrad_tasks_f32 (gw [T][Co][Ci][3][3], gw[t] over the samples n % T == t), Winograd form.
extern "C" int savfi_conv3x3_wgrad_wino_tasks_f32(const float* x, const float* gz, float* gw, float* workspace, int N, int T,
                                                  int Ci, int Co, int H, int W, int pad, void* stream) {
  return wino_wgrad_tasks(x, gz, gw, nullptr, workspace, N, T, Ci, Co, H, W, pad, stream);
}

// ... and the bias gradient with it: gb [T][Co], gb[t][co] = the sum of gz over the samples n % T == t and the map -- what a bias added to
// the convolution's output gets.  The weight gradient reads gz anyway; its first ci block of every co block sums what it reads (the
// separate pass, savfi_bias_act_bwd_f32 in its sums-only use, read the whole map once more: 34 us for a [8,32,384,512] map).
// workspace: savfi_conv3x3_wgrad_wino_tasks_bias_workspace_floats.
extern "C" int savfi_conv3x3_wgrad_wino_tasks_bias_f32(const float* x, const float* gz, float* gw, float* gb, float* workspace, int N, int T,
                                                       int Ci, int Co, int H, int W, int pad, void* stream) {
  if (!gb) return SAVFI_E_NULL;
  return wino_wgrad_tasks(x, gz, gw, gb, workspace, N, T, Ci, Co, H, W, pad, stream);
}

static int wino_wgrad_tasks(const float* x, const float* gz, float* gw, float* gb, float* workspace, int N, int T, int Ci, int Co, int H,
                            int W, int pad, void* stream) {
  if (!x || !gz || !gw || !workspace) return SAVFI_E_NULL;
  if (N <= 0 || T <= 0 || N % T != 0 || Ci <= 0 || Co <= 0 || H <= 0 || W <= 0) return SAVFI_E_SHAPE;
  if (pad != 0 && pad != 1) return SAVFI_E_UNSUPPORTED;
  WWPlan p;
  if (!ww_plan(p, N, T, Ci, Co, H, W, pad)) return SAVFI_E_SHAPE;
  // a sample below 2^30 bytes: byte offsets plus the out-of-range markers (WW_OOR) stay inside 32 bits
  if ((int64_t)Ci * H * W >= ((int64_t)1 << 28) || (int64_t)Co * p.Ho * p.Wo >= ((int64_t)1 << 28)) return SAVFI_E_TOOBIG;
  if (p.ngroups > 65535 || (int64_t)p.cobs * p.cibs > 65535 || (int64_t)Co * Ci * 9 > 0x7fffffffLL || T > 65535) return SAVFI_E_TOOBIG;
  hipStream_t st = (hipStream_t)stream;
  constexpr size_t lds = (size_t)WW_LDS_FLOATS * sizeof(float), lds_bias = lds + 256 * sizeof(float);
  static uint32_t attr_done = 0, attr_done_b = 0;
  if (int e = gb ? savfi_ensure_dynamic_lds((const void*)wino_wgrad3x3<true>, lds_bias, attr_done_b)
                 : savfi_ensure_dynamic_lds((const void*)wino_wgrad3x3<false>, lds, attr_done)) return e;
  float* bias_partial = gb ? workspace + p.partial_floats + p.stage2_floats : nullptr;
  WWArgs a{x, gz, workspace, Ci, Co, H, W, p.Ho, p.Wo, pad, T, p.cibs, p.ty_cnt, p.tx_cnt, p.cpr, p.chunks_total, p.chunks_per_split,
           bias_partial};
  if (gb) hipLaunchKernelGGL(wino_wgrad3x3<true>, dim3(p.nsplit, p.cobs * p.cibs, T), dim3(256), lds_bias, st, a);
  else hipLaunchKernelGGL(wino_wgrad3x3<false>, dim3(p.nsplit, p.cobs * p.cibs, T), dim3(256), lds, st, a);
  if (int e = savfi_launch_status()) return e;
  if (gb) {
    hipLaunchKernelGGL(wino_wgrad_bias_reduce, dim3(savfi_cdiv(Co, 64), T), dim3(64), 0, st, bias_partial, gb, Co, p.cobs, p.nsplit);
    if (int e = savfi_launch_status()) return e;
  }
  const int ntiles = p.cobs * p.cibs;
  const size_t block_floats = (size_t)ntiles * WW_BLOCK_FLOATS;
  float* stage2 = workspace + p.partial_floats;
  if (p.ngroups == 1) {      // at most RG splits: level 2 alone, on the splits' blocks (see savfi_conv3x3_wgrad_tasks_f32)
    hipLaunchKernelGGL(wino_wgrad_reduce2, dim3(savfi_cdiv(Co * Ci * 9, 256), T), dim3(256), 0, st, workspace, gw, Co, Ci, p.cibs, ntiles,
                       p.nsplit);
    return savfi_launch_status();
  }
  hipLaunchKernelGGL(wgrad_reduce1, dim3((unsigned)((block_floats / 4 + 255) / 256), p.ngroups, T), dim3(256), 0, st, workspace,
                     stage2, block_floats, p.nsplit);
  if (int e = savfi_launch_status()) return e;
  hipLaunchKernelGGL(wino_wgrad_reduce2, dim3(savfi_cdiv(Co * Ci * 9, 256), T), dim3(256), 0, st, stage2, gw, Co, Ci, p.cibs, ntiles,
                     p.ngroups);
  return savfi_launch_status();
}
