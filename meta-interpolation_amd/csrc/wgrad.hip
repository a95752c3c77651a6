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
