// Winograd F(4x4, 3x3) for the 3x3 / stride 1 layers of at most 512 -> 512 channels (included by winograd.hip only: the
// multi-layer filter transform launches both forms from one kernel).
//
// Why.  The F(2x2, 3x3) kernel above multiplies 16 Winograd points per 4 outputs (2.25x fewer multiplies than the direct sum);
// F(4x4, 3x3) multiplies 36 points per 16 outputs (4x fewer), and its transformed volume per output pixel is SMALLER (36 / 16
// against 16 / 4 values): fewer MFMAs, fewer LDS dwords and fewer transform results per pixel.  On the shallow, wide layers of
// the SepConv tail (51 -> 51 @258x450, 64 -> 51, 32 -> 32 @384x512: 22 ms of a C2 meta-iteration at F(2x2)) the multiplies are
// the cost.  Rounding: interpolation points 0, +-1, +-2, inf (Lavin & Gray); against float64 the fp32 result is 3.6e-7 rms /
// 4.6e-6 max of the output's scale (direct fp32 sum: 7e-8 / 6e-7, F(2x2): 6e-8 / 4e-7; numpy transcription of the same
// arithmetic, tools/r6/wino4_rounding.py) -- two orders below north_star's 1e-4 pixel L1.  Deeper layers keep F(2x2) / the split-bf16
// direct kernel: their reduction splits, and the error of F(4x4) grows with the reduction length.
//
//   Y = A^T [ (G g G^T) (.) (B^T d B) ] A        per 4x4 output tile, d = its 6x6 input patch
//   M[xi][i][t] = sum_k U[xi][k][i] V[xi][k][t]   36 GEMMs on v_mfma_f32_16x16x4_f32   (k: reduction channel, i: produced)
//
// Workgroup = 4 waves, two per CU (73.7 KB LDS, <= 256 registers): 32 tiles (a 2^(5-s) x 2^s block = 512 output pixels) x 32
// produced channels; wave w owns Winograd points 9w .. 9w+8 for all of them (9 x 2 x 2 accumulator tiles = 144 registers) and
// walks the reduction channels 8 at a time (two MFMA k-steps, 72 MFMAs).  Per chunk every thread transforms ONE 6x6 patch
// (tile = lane % 32, channel = 2 wave + lane / 32): first pass over the patch rows for TWO columns at once on packed fp32
// instructions (v_pk_fma_f32 / v_pk_add_f32: the loads deliver neighbouring columns in neighbouring registers; -5 % on the 51-channel
// layers), second pass along one transformed row, which leaves as 6 conflict-free ds_write_b32 into the other V buffer -- and that
// patch row's registers are reloaded for the chunk after next (10-15 steps of 4 MFMAs ahead of their first use).  A fragments: 18
// floats per lane and k-step from a pre-swizzled U (four 16-byte + one 8-byte load), reloaded in place half a chunk ahead; B
// fragments: one ds_read_b64 per (point, k-step), two steps ahead.  Output stage: the 36 points of 16 channels meet in LDS
// (bank-swizzled by the writer's row group), each thread finishes two (tile, channel) pairs per round: A^T M A (first pass packed),
// bias / (leaky) ReLU (specialised per launch: 0-4 VALU per element), mask, row stores.
//
// Measured (tools/r6/wino4_time.py, profiles/r06_wino4_*.txt; T = 4 x 2 samples unless N is given): 64 -> 64 @192x256 95 us = 307
// direct-equivalent TFLOP/s (F(2x2): 170 us, split-bf16 direct: 142), 128 -> 128 @96x128 94 us = 309, 51 -> 51 @258x450 N = 32
// 814 us = 211 (F(2x2): 1230), 32 -> 32 @384x512 164 us = 176 (HBM: 400 MB).  Timing-only ablations: the channel loop alone runs at
// 0.75-0.8 of the fp32 matrix peak; what remains is every non-MFMA instruction (fp32 VALU shares the matrix pipe's datapath: the
// transform's VALU does not hide under MFMAs), the output stage (14-26 %) and, for 51 channels, padding (56 x 64 of 51 x 51: 38 %).
#pragma once

namespace w4 {

constexpr int TT = 32;              // tiles per workgroup
constexpr int COB = 32;             // produced channels per workgroup
constexpr int KC = 8;               // reduction channels per chunk
constexpr int PTS = 36;
constexpr int VBUF = PTS * KC * TT;           // floats per V buffer (36 KB)
constexpr int XBUF = PTS * 16 * TT;           // exchange of one 16-channel round (72 KB)
constexpr int LDS_FLOATS = XBUF > 2 * VBUF ? XBUF : 2 * VBUF;
#ifndef SAVFI_W4_MAXC
#define SAVFI_W4_MAXC 512
#endif

// which layers run on this form: by channel counts only (the filter transform does not know the map size)
__host__ __device__ inline bool use_f4(int K, int I) { return K <= SAVFI_W4_MAXC && I <= SAVFI_W4_MAXC; }
__host__ __device__ inline int kp_of(int K) { return (K + KC - 1) / KC * KC; }
__host__ __device__ inline int ip_of(int I) { return (I + COB - 1) / COB * COB; }

// ---- filter transform --------------------------------------------------------------------------------------------
// U = G g G^T (6x6) per (k, i), stored in the order a wave fetches it:
//   Uf[chunk = k / 8][cob = i / 32][wave = xi / 9][s = (k % 8) / 4][18 floats x 64 lanes],  lane = (k % 4) * 16 + i % 16,
//   value v = 2 (xi % 9) + (i % 32) / 16:  v < 16 -> float4 block v / 4 of the lane, component v % 4; v >= 16 -> the float2 tail
// mode 0 / 1 and the task stride as filter_transform_block above.
__device__ __forceinline__ void g_rows(const float (&g)[3], float (&t)[6]) {
  const float s = g[0] + g[2];
  const float s4 = fmaf(4.f, g[2], g[0]);
  t[0] = 0.25f * g[0];
  t[1] = (s + g[1]) * (-1.f / 6.f);
  t[2] = (s - g[1]) * (-1.f / 6.f);
  t[3] = fmaf(2.f, g[1], s4) * (1.f / 24.f);
  t[4] = fmaf(-2.f, g[1], s4) * (1.f / 24.f);
  t[5] = g[2];
}

// One workgroup = 4 reduction channels (one k-step: chunk and s fixed) x 64 produced channels = the eight whole [18 x 64] fragment blocks
// (2 channel blocks x 4 waves, 36 KB) of that k-step.  A thread's 36 values belong to 36 different 4-byte slots of those blocks: they meet
// in LDS and leave as 16-byte stores of whole blocks (round 6: the scattered dword stores ran at 3.3 TB/s, and with 36 points per weight
// the transforms of the fast weights are 2.3 GB per inner step of config C2 -- 4 ms per meta-iteration).
__device__ __forceinline__ void filter_transform_block4(const float* __restrict__ w, float* __restrict__ U, int Co, int Ci, int K, int I,
                                                        int KP, int IP, int mode, int bx, int by) {
  __shared__ __attribute__((aligned(16))) float stage[8 * 18 * 64];
  const int kk = threadIdx.x & 3, ii = threadIdx.x >> 2;
  const int nI = (IP + 63) / 64;
  const int k = 4 * bx + kk, i = 64 * (by % nI) + ii;
  const int task = by / nI;
  w += (size_t)task * Co * Ci * 9;
  U += (size_t)task * PTS * KP * IP;
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
  float t[6][3];      // G g
#pragma unroll
  for (int b = 0; b < 3; ++b) {
    const float col[3] = {g[0][b], g[1][b], g[2][b]};
    float o[6];
    g_rows(col, o);
#pragma unroll
    for (int r = 0; r < 6; ++r) t[r][b] = o[r];
  }
  const int cbl = ii >> 5, cb = (ii >> 4) & 1, il = ii & 15;      // channel block of the pair, 16-channel half, lane column
  const int lane = kk * 16 + il;
#pragma unroll
  for (int r = 0; r < 6; ++r) {
    float o[6];
    g_rows(t[r], o);
#pragma unroll
    for (int c = 0; c < 6; ++c) {
      const int xi = 6 * r + c, wave = xi / 9, v = 2 * (xi % 9) + cb;
      float* blk = stage + (cbl * 4 + wave) * (18 * 64);
      if (v < 16) blk[((v >> 2) * 64 + lane) * 4 + (v & 3)] = o[c];
      else blk[16 * 64 + lane * 2 + (v - 16)] = o[c];
    }
  }
  __syncthreads();
  // block (cbl, wave) of this k-step -> U block (((chunk ncob + cob) 4 + wave) 2 + s)
  const int chunk = bx >> 1, s2 = bx & 1, ncob = IP / COB, cob0 = 2 * (by % nI);
  if (4 * bx >= KP) return;
#pragma unroll
  for (int it = 0; it < 8 * 18 * 64 / 4 / 256; ++it) {           // 9 float4 per thread
    const int e = (it * 256 + threadIdx.x) * 4;                  // float index inside the 8 staged blocks
    const int blkid = e / (18 * 64), off = e - blkid * (18 * 64);
    const int cobq = cob0 + (blkid >> 2), wv = blkid & 3;
    if (cobq >= ncob) continue;
    float* dst = U + ((((size_t)chunk * ncob + cobq) * 4 + wv) * 2 + s2) * (18 * 64) + off;
    *reinterpret_cast<f32x4*>(dst) = *reinterpret_cast<const f32x4*>(stage + e);
  }
}

// ---- fused convolution ---------------------------------------------------------------------------------------------
struct W4Args {
  const float* x;
  const float* U;
  const float* bias;
  float* out;
  int K, I, KP, IP, H, W, Ho, Wo, off, tiles_y, tiles_x;
  float slope;
  int tile_shift;
  int T, N;
  const float* mask;
  float mask_slope;
  int out_unit16;
  int nsplit, chunks_per_split;       // reduction chunks split over workgroups (deep layers on small maps): raw partial outputs
  float* partial;                     // [split][N][I][Ho][Wo], summed (+ bias, activation, mask) by wino_split_reduce
};

__device__ __forceinline__ i32x4 w4_rsrc(const void* base, unsigned bytes) {
  const unsigned long long p = reinterpret_cast<unsigned long long>(base);
  i32x4 r;
  r.x = (int)(unsigned)p;
  r.y = (int)(unsigned)(p >> 32);
  r.z = (int)bytes;
  r.w = 0x00020000;
  return r;
}

// a descriptor the compiler computed with vector instructions (selects): back to SGPRs, or every store becomes a waterfall loop
__device__ __forceinline__ i32x4 w4_uniform(i32x4 r) {
  r.x = __builtin_amdgcn_readfirstlane(r.x); r.y = __builtin_amdgcn_readfirstlane(r.y);
  r.z = __builtin_amdgcn_readfirstlane(r.z); r.w = __builtin_amdgcn_readfirstlane(r.w);
  return r;
}

// one 6-vector of B^T d:  B^T = [4 0 -5 0 1 0; 0 -4 -4 1 1 0; 0 4 -4 -1 1 0; 0 -2 -1 2 1 0; 0 2 -1 -2 1 0; 0 4 0 -5 0 1]
__device__ __forceinline__ void bt6(float d0, float d1, float d2, float d3, float d4, float d5, float (&o)[6]) {
  const float a = fmaf(-4.f, d2, d4), b = fmaf(-4.f, d1, d3);
  const float c = d4 - d2, e = d3 - d1;
  o[0] = fmaf(4.f, d0, fmaf(-5.f, d2, d4));
  o[1] = a + b;
  o[2] = a - b;
  o[3] = fmaf(2.f, e, c);
  o[4] = fmaf(-2.f, e, c);
  o[5] = fmaf(4.f, d1, fmaf(-5.f, d3, d5));
}

// the same on two columns at once (v_pk_fma_f32 / v_pk_add_f32)
__device__ __forceinline__ f32x2 w4_fma2(float s, f32x2 x, f32x2 y) { return __builtin_elementwise_fma((f32x2){s, s}, x, y); }
__device__ __forceinline__ void bt6p(f32x2 d0, f32x2 d1, f32x2 d2, f32x2 d3, f32x2 d4, f32x2 d5, f32x2 (&o)[6]) {
  const f32x2 a = w4_fma2(-4.f, d2, d4), b = w4_fma2(-4.f, d1, d3);
  const f32x2 c = d4 - d2, e = d3 - d1;
  o[0] = w4_fma2(4.f, d0, w4_fma2(-5.f, d2, d4));
  o[1] = a + b;
  o[2] = a - b;
  o[3] = w4_fma2(2.f, e, c);
  o[4] = w4_fma2(-2.f, e, c);
  o[5] = w4_fma2(4.f, d1, w4_fma2(-5.f, d3, d5));
}

// one 6-vector of A^T m:  A^T = [1 1 1 1 1 0; 0 1 -1 2 -2 0; 0 1 1 4 4 0; 0 1 -1 8 -8 1]
__device__ __forceinline__ void at6(float m0, float m1, float m2, float m3, float m4, float m5, float (&o)[4]) {
  const float s1 = m1 + m2, d1 = m1 - m2, s2 = m3 + m4, d2 = m3 - m4;
  o[0] = m0 + s1 + s2;
  o[1] = fmaf(2.f, d2, d1);
  o[2] = fmaf(4.f, s2, s1);
  o[3] = fmaf(8.f, d2, d1) + m5;
}

__device__ __forceinline__ void at6p(f32x2 m0, f32x2 m1, f32x2 m2, f32x2 m3, f32x2 m4, f32x2 m5, f32x2 (&o)[4]) {
  const f32x2 s1 = m1 + m2, d1 = m1 - m2, s2 = m3 + m4, d2 = m3 - m4;
  o[0] = m0 + s1 + s2;
  o[1] = w4_fma2(2.f, d2, d1);
  o[2] = w4_fma2(4.f, s2, s1);
  o[3] = w4_fma2(8.f, d2, d1) + m5;
}

// VECW: elements per row store (4: Wo % 4 == 0, 2: Wo even, 1); IN16 = 1 + off (1 or 2): the input is unit-major ([y][x / 16][channel][16],
// W % 16 == 0; 0: planar) -- column `off` of a patch row is then 16-byte aligned inside a unit: one 16-byte load + the two columns beside it;
// MASK: out *= (mask > 0 ? 1 : mask_slope), mask laid out like the planar output (64 more registers: its own instance)
template <int VECW, int IN16, bool MASK>
__global__ __launch_bounds__(256, 2) void wino4_conv3x3(W4Args a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 15, kg = lane >> 4;

  // work item: [sample][tile block][channel block], channel block fastest; one XCD walks a contiguous eighth (winograd.hip): the workgroups
  // that read the same input tiles run side by side on one XCD's L2 (tile block fastest, the first version: 51 -> 51 @258x450 777 / 928 us
  // forward / data gradient against 752 / 886, 64 -> 64 @192x256 93 against 90.5; the transformed filters are L2-resident either way)
  const int nblk = a.IP / COB;
  int tb, cob, n, sp;
  {
    const unsigned flat = blockIdx.x, nwg = gridDim.x;
    const unsigned ntb = (unsigned)(a.tiles_y * a.tiles_x);
    const unsigned xcd = flat & 7u, q8 = nwg >> 3, rem = nwg & 7u;
    unsigned item = xcd * q8 + min(xcd, rem) + (flat >> 3);
    cob = (int)(item % (unsigned)nblk);
    item /= (unsigned)nblk;
    tb = (int)(item % ntb);
    item /= ntb;
    sp = (int)(item % (unsigned)a.nsplit);
    n = (int)(item / (unsigned)a.nsplit);
  }
  const int tby = tb / a.tiles_x, tbx = tb - tby * a.tiles_x;
  const int i0 = cob * COB;
  const int task = a.T > 1 ? n % a.T : 0;
  const size_t cplane = (size_t)a.H * a.W;
  const float* xp = a.x + (size_t)n * a.K * cplane;
  const int tsh = a.tile_shift, tbw = 1 << tsh, tbh = TT >> tsh;

  // ---- transform role: tile tl, channel kc of the chunk ----
  const int tl = lane & 31, kc = 2 * w + (lane >> 5);
  const int y0 = 4 * (tby * tbh + (tl >> tsh)) - a.off, x0 = 4 * (tbx * tbw + (tl & (tbw - 1))) - a.off;
  constexpr int WC = IN16 ? IN16 - 1 : 0;       // first column of a row's 16-byte load
  unsigned pv[6], pn[IN16 ? 6 : 1], pn2[IN16 == 2 ? 6 : 1];
  unsigned long long shl = 0, colm[6] = {~0ull, ~0ull, ~0ull, ~0ull, ~0ull, ~0ull};
  int partial = 0;
  unsigned chunk_bytes;              // channel step of the descriptor per chunk
  unsigned plane_b;                  // bytes between two channels as this thread addresses them
  if constexpr (IN16) {
    // sample [y][x / 16][K][16].  off 2: columns 2..5 = 16 bytes, columns 0, 1 = 8 bytes (whole inside one unit: inside or outside the
    // image together); off 1: columns 1..4 = 16 bytes, columns 0 and 5 a dword each.  Outside the image: an offset beyond the buffer.
    plane_b = 64u;
    chunk_bytes = KC * 64u;
    auto at = [&](bool row, int y, int x) {
      return (row && x >= 0 && x < a.W) ? (unsigned)((y * (a.W >> 4) + (x >> 4)) * a.K) * 64u + (unsigned)(x & 15) * 4u + (unsigned)kc * 64u : 0x80000000u;
    };
#pragma unroll
    for (int r = 0; r < 6; ++r) {
      const int y = y0 + r;
      const bool row = y >= 0 && y < a.H;
      pv[r] = at(row, y, x0 + WC);
      pn[IN16 ? r : 0] = at(row, y, x0);
      if constexpr (IN16 == 2) pn2[r] = at(row, y, x0 + 5);
    }
  } else {
    plane_b = (unsigned)cplane * 4u;
    chunk_bytes = KC * plane_b;
    const int xa = x0 < 0 ? 0 : x0;
#pragma unroll
    for (int r = 0; r < 6; ++r) {
      const int y = y0 + r;
      pv[r] = (y >= 0 && y < a.H) ? (unsigned)(y * a.W + xa) * 4u + (unsigned)kc * plane_b : 0x80000000u;
    }
    shl = __builtin_amdgcn_ballot_w64(x0 < 0);
    const unsigned long long all = __builtin_amdgcn_ballot_w64(true);
#pragma unroll
    for (int c = 0; c < 6; ++c) {
      colm[c] = __builtin_amdgcn_ballot_w64(x0 + c >= 0 && x0 + c < a.W);
      if (colm[c] != all) partial |= 1 << c;      // (bit c: some lane of this wave must clear column c)
    }
  }
  // this workgroup reduces over chunks [cbeg, cbeg + nch) of the KP / KC chunks; everything below counts chunks from cbeg
  const int cbeg = sp * a.chunks_per_split;
  const int nch = min(cbeg + a.chunks_per_split, a.KP / KC) - cbeg;
  // descriptor of chunk c: starts at its first channel, ends with the tensor's last real channel (a lane whose channel does not exist reads
  // zeros or a neighbour's finite values: its filter transform is zero)
  const unsigned sample_bytes = (unsigned)a.K * (unsigned)cplane * 4u;
  auto xrs = [&](int c) {
    c = (c < nch - 1 ? c : nch - 1) + cbeg;
    const unsigned first = (unsigned)c * chunk_bytes;
    return w4_rsrc(reinterpret_cast<const char*>(xp) + first, sample_bytes - first);
  };

  float d[6][6];
  auto load_wide = [&](const i32x4& rs) {
#pragma unroll
    for (int r = 0; r < 6; ++r) {
      const f32x4 v = savfi_raw_buffer_load_x4(rs, (int)pv[r], 0, 0);
      d[r][WC] = v.x; d[r][WC + 1] = v.y; d[r][WC + 2] = v.z; d[r][WC + 3] = v.w;
    }
  };
  auto load_narrow = [&](const i32x4& rs) {
#pragma unroll
    for (int r = 0; r < 6; ++r) {
      if constexpr (IN16 == 3) {
        const f32x2 v = savfi_raw_buffer_load_x2(rs, (int)pn[r], 0, 0);
        d[r][0] = v.x; d[r][1] = v.y;
      } else if constexpr (IN16 == 2) {
        d[r][0] = savfi_raw_buffer_load_x1(rs, (int)pn[r], 0, 0);
        d[r][5] = savfi_raw_buffer_load_x1(rs, (int)pn2[r], 0, 0);
      } else {
        const f32x2 v = savfi_raw_buffer_load_x2(rs, (int)pv[r] + 16, 0, 0);
        d[r][4] = v.x; d[r][5] = v.y;
      }
    }
  };
  // Waves with a tile on the left / right image border (wave-uniform `edge`: ONE branch per chunk -- per-row and per-column branches
  // were 48 per chunk, in every wave): left-edge tiles were loaded from column 0 -- move the rows right by `off` --, then clear the
  // columns outside the image.  Selects on lane masks behind wave-uniform tests: only what the wave needs (with tile blocks 32 tiles wide
  // every workgroup of a map up to 256 pixels wide is a border workgroup: a right-edge wave clears one column, 6 selects, where the
  // unconditional form spent 30 + 36 on masks that change nothing).
  const bool edge = !IN16 && (shl != 0 || partial != 0);
  auto fix_patch = [&]() {
    if (shl != 0) {
      if (a.off == 1) {
#pragma unroll
        for (int r = 0; r < 6; ++r)
#pragma unroll
          for (int c = 5; c >= 1; --c) asm volatile("v_cndmask_b32 %0, %0, %1, %2" : "+v"(d[r][c]) : "v"(d[r][c - 1]), "s"(shl));
      } else if (a.off == 2) {
#pragma unroll
        for (int r = 0; r < 6; ++r)
#pragma unroll
          for (int c = 5; c >= 2; --c) asm volatile("v_cndmask_b32 %0, %0, %1, %2" : "+v"(d[r][c]) : "v"(d[r][c - 2]), "s"(shl));
      }
    }
#pragma unroll
    for (int c = 0; c < 6; ++c)
      if (partial & (1 << c)) {
#pragma unroll
        for (int r = 0; r < 6; ++r) asm volatile("v_cndmask_b32 %0, 0, %0, %1" : "+v"(d[r][c]) : "s"(colm[c]));
      }
  };
  // V[xi][s][kg][j][tb] (floats): a reader's two tile blocks of one (xi, k) are one ds_read_b64; the 64 lanes of a writer cover 64 banks
  const int vw = (((kc >> 2) * 4 + (kc & 3)) * 16 + (tl & 15)) * 2 + (tl >> 4);
  // packed form: first pass over the patch ROWS for two columns at once (the loads deliver neighbouring columns in neighbouring registers),
  // second pass along the columns of one transformed row, which then leaves as 6 stores -- and its patch row's registers are reloaded
  f32x2 tp[6][3];
  auto pk_cols = [&](int cp) {
    f32x2 o[6];
    bt6p((f32x2){d[0][2 * cp], d[0][2 * cp + 1]}, (f32x2){d[1][2 * cp], d[1][2 * cp + 1]}, (f32x2){d[2][2 * cp], d[2][2 * cp + 1]},
         (f32x2){d[3][2 * cp], d[3][2 * cp + 1]}, (f32x2){d[4][2 * cp], d[4][2 * cp + 1]}, (f32x2){d[5][2 * cp], d[5][2 * cp + 1]}, o);
#pragma unroll
    for (int r = 0; r < 6; ++r) tp[r][cp] = o[r];
  };
  auto row_out = [&](int rp, float* vdst) {
    float o[6];
    bt6(tp[rp][0].x, tp[rp][0].y, tp[rp][1].x, tp[rp][1].y, tp[rp][2].x, tp[rp][2].y, o);
#pragma unroll
    for (int c = 0; c < 6; ++c) vdst[vw + (6 * rp + c) * 256] = o[c];
  };
  auto load_row = [&](const i32x4& rs, int r) {
    const f32x4 v = savfi_raw_buffer_load_x4(rs, (int)pv[r], 0, 0);
    d[r][WC] = v.x; d[r][WC + 1] = v.y; d[r][WC + 2] = v.z; d[r][WC + 3] = v.w;
    if constexpr (IN16 == 3) {
      const f32x2 u = savfi_raw_buffer_load_x2(rs, (int)pn[r], 0, 0);
      d[r][0] = u.x; d[r][1] = u.y;
    } else if constexpr (IN16 == 2) {
      d[r][0] = savfi_raw_buffer_load_x1(rs, (int)pn[r], 0, 0);
      d[r][5] = savfi_raw_buffer_load_x1(rs, (int)pn2[r], 0, 0);
    } else {
      const f32x2 u = savfi_raw_buffer_load_x2(rs, (int)pv[r] + 16, 0, 0);
      d[r][4] = u.x; d[r][5] = u.y;
    }
  };

  // ---- MFMA role ----
  f32x4 acc[9][2][2];
#pragma unroll
  for (int q = 0; q < 9; ++q)
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
      for (int t = 0; t < 2; ++t) acc[q][cb][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const unsigned ubytes = (unsigned)PTS * (unsigned)a.KP * (unsigned)a.IP * 4u;
  const i32x4 urs = w4_rsrc(a.U + (size_t)task * PTS * a.KP * a.IP, ubytes);
  const unsigned ulane4 = (unsigned)lane * 16u, ulane2 = 16u * 64u * 4u + (unsigned)lane * 8u;
  // byte offset of the fragment block of (chunk c, half s)
  auto u_of = [&](int c, int s) {
    c = (c < nch - 1 ? c : nch - 1) + cbeg;
    return ((((unsigned)c * (unsigned)nblk + (unsigned)cob) * 4u + (unsigned)w) * 2u + (unsigned)s) * (18u * 64u * 4u);
  };
  f32x4 a4[4];
  f32x2 a2;
  auto load_a4 = [&](int b, unsigned uo) { a4[b] = savfi_raw_buffer_load_x4(urs, (int)(ulane4 + (unsigned)b * 1024u), (int)uo, 0); };
  auto load_a2 = [&](unsigned uo) { a2 = savfi_raw_buffer_load_x2(urs, (int)ulane2, (int)uo, 0); };
  const int vr = (9 * w * 8 + kg) * 32 + 2 * j;        // + (q * 2 + s) * 128 floats

  // ---- prologue: P(0) -> V(0); P(1) and A(0, half 0) in flight ----
  {
    const i32x4 rs = xrs(0);
    load_wide(rs);
    load_narrow(rs);
  }
  load_a4(0, u_of(0, 0)); load_a4(1, u_of(0, 0)); load_a4(2, u_of(0, 0)); load_a4(3, u_of(0, 0)); load_a2(u_of(0, 0));
  if (edge) fix_patch();
#pragma unroll
  for (int cp = 0; cp < 3; ++cp) pk_cols(cp);
#pragma unroll
  for (int r = 0; r < 6; ++r) row_out(r, lds);
  {
    const i32x4 rs = xrs(1);
    load_wide(rs);
    load_narrow(rs);
  }
  __builtin_amdgcn_s_waitcnt(0);
  __syncthreads();

  // One chunk: 18 steps of 4 MFMAs (half s = step / 9, point q = step % 9).  Steps 9-11: row pass of the next chunk's patch; steps
  // 12-17: its column pass into the other V buffer; the patch after that is requested behind the 4th / 6th column.
  const bool half_last = cbeg + nch == a.KP / KC && a.K - (a.KP - KC) <= 4;
  auto chunk = [&](int ch, const float* vcur, float* vnext, auto last_tag) {
    constexpr bool last = decltype(last_tag)::value;
    f32x2 bq[3];
    bq[0] = *reinterpret_cast<const f32x2*>(vcur + vr);
    bq[1] = *reinterpret_cast<const f32x2*>(vcur + vr + 256);
    const i32x4 rs2 = xrs(ch + 2);
#pragma unroll
    for (int g = 0; g < 18; ++g) {
      const int s = g / 9, q = g % 9;
      if (last && g == 9 && half_last) break;      // the tensor ends in this chunk's first four channels (K = 51: 7 % of the layer's MFMAs)
      if (g + 2 < 18) {
        const int s2 = (g + 2) / 9, q2 = (g + 2) % 9;
        bq[(g + 2) % 3] = *reinterpret_cast<const f32x2*>(vcur + vr + q2 * 256 + s2 * 128);
      }
#pragma unroll
      for (int cb = 0; cb < 2; ++cb) {
        const int v = 2 * q + cb;
        const float av = v < 16 ? a4[v >> 2][v & 3] : a2[v - 16];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          acc[q][cb][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bq[g % 3][t], acc[q][cb][t], 0, 0, 0);
        }
      }
      // A fragments of the next half, in place behind their last use
      const unsigned un = s == 0 ? u_of(ch, 1) : u_of(ch + 1, 0);
      if (q == 1 || q == 3 || q == 5 || q == 7) load_a4(q >> 1, un);
      if (q == 8) load_a2(un);
      if (!last) {
        if (g == 9 && edge) fix_patch();
        if (g >= 9 && g < 12) pk_cols(g - 9);
        if (g >= 12) { row_out(g - 12, vnext); load_row(rs2, g - 12); }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  for (int ch = 0; ch < nch - 1; ++ch) {
    const int cur = ch & 1;
    chunk(ch, lds + cur * VBUF, lds + (cur ^ 1) * VBUF, std::false_type{});
    __syncthreads();
  }
  chunk(nch - 1, lds + ((nch - 1) & 1) * VBUF, lds, std::true_type{});
  __builtin_amdgcn_s_waitcnt(0);
  __syncthreads();

  // ---- output stage ------------------------------------------------------------------------------------------------
  // bias of this thread's output channels: round cb, pair qq -> channel i0 + 16 cb + (tid >> 5) + 8 qq (fetched here, behind the channel
  // loop, where no store is outstanding yet: a load issued between stores waits for every older store's acknowledge)
  const int split = a.nsplit > 1 ? 1 : 0;          // bias / activation / mask then happen in wino_split_reduce
  float* const obase = split ? a.partial + (size_t)sp * a.N * a.I * a.Ho * a.Wo : a.out;
  float bvals[2][2];
#pragma unroll
  for (int cb = 0; cb < 2; ++cb)
#pragma unroll
    for (int qq = 0; qq < 2; ++qq) {
      const int i = i0 + 16 * cb + (tid >> 5) + 8 * qq;
      bvals[cb][qq] = (a.bias && split == 0 && i < a.I) ? a.bias[task * a.I + i] : 0.f;
    }

  // this thread finishes pairs p = tid + 256 qq of a round: tile p % 32 (= tid % 32), channel p / 32 of the round's 16
  const int otl = tid & 31, och = tid >> 5;
  const int oy = 4 * (tby * tbh + (otl >> tsh)), ox = 4 * (tbx * tbw + (otl & (tbw - 1)));
  const unsigned oplane = (unsigned)(a.Ho * a.Wo) * 4u;
  constexpr int NS = 4 / VECW;                     // stores per row
  unsigned ooff[4][NS];
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int e = 0; e < NS; ++e) {
      const int x = ox + e * VECW;
      const bool ok = oy + r < a.Ho && x < a.Wo;
      if (a.out_unit16) ooff[r][e] = ok ? (unsigned)(((oy + r) * (a.Wo >> 4) + (x >> 4)) * a.I) * 64u + (unsigned)(x & 15) * 4u : 0x80000000u;
      else ooff[r][e] = ok ? (unsigned)((oy + r) * a.Wo + x) * 4u + (unsigned)(lane >> 5) * oplane : 0x80000000u;
    }
  // planar: a descriptor over the wave's two channel planes of a pair (och = 2 w + lane / 32); a channel beyond I shrinks it
  auto pair_rsrc = [&](const float* base, int cb, int qq) {
    const int ie = i0 + 16 * cb + 2 * w + 8 * qq;        // the wave's even channel
    const int have = a.I - ie < 0 ? 0 : (a.I - ie > 2 ? 2 : a.I - ie);
    return w4_rsrc(base + ((size_t)n * a.I + (ie < a.I ? ie : 0)) * a.Ho * a.Wo, (unsigned)have * oplane);
  };
  float mk[MASK ? 2 : 1][MASK ? 2 : 1][4][4];
  if constexpr (MASK) {          // every mask value before the first store (one in-order counter for loads and stores)
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
      for (int qq = 0; qq < 2; ++qq) {
        const i32x4 mrs = w4_uniform(pair_rsrc(a.mask, cb, qq));
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int e = 0; e < NS; ++e) {
            if constexpr (VECW == 4) {
              const f32x4 v = savfi_raw_buffer_load_x4(mrs, (int)ooff[r][e], 0, 0);
              mk[cb][qq][r][0] = v.x; mk[cb][qq][r][1] = v.y; mk[cb][qq][r][2] = v.z; mk[cb][qq][r][3] = v.w;
            } else if constexpr (VECW == 2) {
              const f32x2 v = savfi_raw_buffer_load_x2(mrs, (int)ooff[r][e], 0, 0);
              mk[cb][qq][r][2 * e] = v.x; mk[cb][qq][r][2 * e + 1] = v.y;
            } else {
              mk[cb][qq][r][e] = savfi_raw_buffer_load_x1(mrs, (int)ooff[r][e], 0, 0);
            }
          }
      }
  }
  const float slope = a.slope;
#pragma unroll
  for (int cb = 0; cb < 2; ++cb) {
    // X[xi][channel 0..15][tile ^ swizzle]: the four row groups of an accumulator tile (channels 4 kg + r) would meet in the same banks
#pragma unroll
    for (int q = 0; q < 9; ++q)
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          lds[((9 * w + q) * 16 + 4 * kg + r) * 32 + ((16 * t + j) ^ (16 * (kg & 1)))] = acc[q][cb][t][r];
    __syncthreads();
    // ACT (wave-uniform, one branch per round): 0 = nothing to add (a data gradient), 1 = bias, 2 = bias + ReLU, 3 = bias + leaky ReLU --
    // 1 / 2 / 4 instead of 4 VALU per output element
    auto finish = [&](auto act_tag) {
      constexpr int ACT = decltype(act_tag)::value;
#pragma unroll
      for (int qq = 0; qq < 2; ++qq) {
        const int chl = och + 8 * qq;
        const int col = otl ^ (16 * ((chl >> 2) & 1));
        f32x2 t4p[4][3];           // A^T m, two columns at a time (a ds_read2st64_b32 delivers the pair: 512 floats apart)
#pragma unroll
        for (int cp = 0; cp < 3; ++cp) {
          f32x2 m[6], o[4];
#pragma unroll
          for (int r = 0; r < 6; ++r)
            m[r] = (f32x2){lds[((6 * r + 2 * cp) * 16 + chl) * 32 + col], lds[((6 * r + 2 * cp + 1) * 16 + chl) * 32 + col]};
          at6p(m[0], m[1], m[2], m[3], m[4], m[5], o);
#pragma unroll
          for (int r = 0; r < 4; ++r) t4p[r][cp] = o[r];
        }
        const float b = bvals[cb][qq];
        const int i = i0 + 16 * cb + chl;
        const i32x4 ors = w4_uniform(a.out_unit16 ? w4_rsrc(obase + (size_t)n * a.I * a.Ho * a.Wo, (unsigned)a.I * oplane) : pair_rsrc(obase, cb, qq));
        const unsigned choff = a.out_unit16 ? (i < a.I ? (unsigned)i * 64u : 0x40000000u) : 0u;      // (+ 0x80000000 of a dropped pixel: still out of range)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float y[4];
          at6(t4p[r][0].x, t4p[r][0].y, t4p[r][1].x, t4p[r][1].y, t4p[r][2].x, t4p[r][2].y, y);
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            float v = y[c];
            if constexpr (ACT >= 1) v += b;
            if constexpr (ACT == 2) v = fmaxf(v, 0.f);
            if constexpr (ACT == 3) v = fmaxf(v, 0.f) + slope * fminf(v, 0.f);
            if constexpr (MASK) v = mk[MASK ? cb : 0][MASK ? qq : 0][r][c] > 0.f ? v : v * a.mask_slope;
            y[c] = v;
          }
          if constexpr (VECW == 4) {
            savfi_raw_buffer_store_x4((f32x4){y[0], y[1], y[2], y[3]}, ors, (int)(ooff[r][0] + choff), 0, 0);
          } else if constexpr (VECW == 2) {
            savfi_raw_buffer_store_x2((f32x2){y[0], y[1]}, ors, (int)(ooff[r][0] + choff), 0, 0);
            savfi_raw_buffer_store_x2((f32x2){y[2], y[3]}, ors, (int)(ooff[r][1] + choff), 0, 0);
          } else {
#pragma unroll
            for (int c = 0; c < 4; ++c) savfi_raw_buffer_store_x1(y[c], ors, (int)(ooff[r][c] + choff), 0, 0);
          }
        }
      }
    };
    if (slope == 1.f || split) {
      if (a.bias == nullptr || split) finish(std::integral_constant<int, 0>{});
      else finish(std::integral_constant<int, 1>{});
    } else if (slope == 0.f) {
      finish(std::integral_constant<int, 2>{});
    } else {
      finish(std::integral_constant<int, 3>{});
    }
    __syncthreads();
  }
}

}  // namespace w4
