// SepConv filter gradients (gV, gH; K = 51, C = 3) on the bf16 matrix cores with error-free 3-way operand splits
// ("bf16x6": every fp32 product as six bf16 products of exactly split operands, fp32 accumulate -- the arithmetic of
// csrc/convk.hip, as close to float64 as an fp32 fmaf chain; tools/bf16_split_probe.hip).
//
//   gV[b,fy,y,x] = sum_c gO[b,c,y,x] * sum_fx in[b,c,y+fy,x+fx] * h[b,fx,y,x]
//   gH[b,fx,y,x] = sum_c gO[b,c,y,x] * sum_fy in[b,c,y+fy,x+fx] * v[b,fy,y,x]
//
// Replaces the reference's two cupy/NVRTC filter-gradient kernels (sepconv/sepconv_op/sepconv.py:32-63, :138-190 backward) --
// same op, same layout -- and supersedes sepconv_bwd_mfma_p (csrc/sepconv.hip) for the shape the model uses.  That kernel
// runs the two banded GEMMs on v_mfma_f32_16x16x4_f32 and is bound by that pipe (360 MFMAs of 32 cycles per 16 pixels:
// 88 % busy at 350 us for B = 8, 256 x 448).  v_mfma_f32_16x16x32_bf16 does eight times the k per instruction at half the
// cycles: six products cost 96 cycles per 16 x 16 x 32 where the fp32 instruction needs 256.
//
// Formulation per wave = 16 pixels (y, x0 + 16 wc + j) of one output row, per channel c (own accumulators: gO[c] multiplies the
// channel's sum afterwards, so the B operands are channel independent):
//   gV:  D[fy][j] = sum_i In_c[y + fy][16 wc + i] * Hb[i][j],   Hb[i][j] = h[j][i - j] (0 <= i - j < 51): M = fy (4 tiles),
//        K = i = 64 window columns (2 steps) -- the two columns i = 64, 65 that pixels 14, 15 still reach are a VALU tail;
//   gH:  D[q][j]  = sum_fy In_c[y + fy][16 wc + q] * v[j][fy],  gH[fx][j] = D[j + fx][j]: M = q (4 tiles + the same two
//        columns as a tail), K = fy (2 steps, taps 51..63 zero).  Its A operand runs DOWN the window's columns: the gfx950
//        transpose read ds_read_b64_tr_b16 delivers it from the same row-major window.
// 288 MFMAs of 16 cycles per 16 pixels (was 360 of 32).
//
// LDS (159 KB, one workgroup of 8 waves per CU):
//   window   three bf16 pieces x 3 channels x 64 circular rows x 80 columns, in blocks of 8 columns:
//            [piece][c][column / 8][row & 63][8]; a block is 64 x 16 B + 128 B.  gV's 16-byte A fragments (lane = row, 16
//            consecutive cells = 256 B; the k-groups a hardware lane group mixes sit two blocks = 2304 B apart) and gH's
//            transpose reads (8 consecutive rows x 4 column quads, the quads of the second block 128 B further) are both
//            free of bank conflicts;
//   taps     per wave ONE table [piece][k / 8][j][8] (6 KB) that holds the skewed h band for gV, then v for gH (the B
//            fragments of a pass live in registers), then the 64 x 16 transpose tile gH leaves through;
//   side     fp32 copies of window columns 64, 65, 80, 81 (the tails), tail sums of gV.
// Persistent launch as sepconv_bwd_mfma_p: one workgroup per CU walks an equal share of "phases" (4 output rows of a
// 32-column strip), the window slides by 8 rows every other phase.
#include "sepconv_x6_shared.h"

namespace {


template <bool VEC>
__global__ __launch_bounds__(XNT) void sepconv_bwd_x6(const float* __restrict__ in, const float* __restrict__ v,
                                                      const float* __restrict__ h, const float* __restrict__ gO,
                                                      float* __restrict__ gV, float* __restrict__ gH,
                                                      int B, int Ho, int Wo, int nph, int ncol, int per_wg) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wc = w & 1, wr = w >> 1;
  const int j = lane & 15, kg = lane >> 4;
  char* const tab = smem + XWINB + w * XTAB;
  float* const side = reinterpret_cast<float*>(smem + XSIDE_OFF);
  float* const tailb = reinterpret_cast<float*>(smem + XTAIL_OFF + w * XTAILB);

  const int total = B * ncol * nph;
  const int g0 = blockIdx.x * per_wg, g1 = min(g0 + per_wg, total);
  if (g0 >= g1) return;
  const int Hi = Ho + XK - 1, Wi = Wo + XK - 1;
  const unsigned plane_b = (unsigned)Ho * (unsigned)Wo * 4u;
  const __amdgpu_buffer_rsrc_t hsrc = x6_rsrc(h, (unsigned)(B * XK) * plane_b);
  const __amdgpu_buffer_rsrc_t vsrc = x6_rsrc(v, (unsigned)(B * XK) * plane_b);
  const __amdgpu_buffer_rsrc_t gsrc = x6_rsrc(gO, (unsigned)(B * XC) * plane_b);
  const __amdgpu_buffer_rsrc_t isrc = x6_rsrc(in, (unsigned)(B * XC) * (unsigned)(Hi * Wi) * 4u);
  const __amdgpu_buffer_rsrc_t gvdst = x6_rsrc(gV, (unsigned)(B * XK) * plane_b);
  const __amdgpu_buffer_rsrc_t ghdst = x6_rsrc(gH, (unsigned)(B * XK) * plane_b);

  auto pos_of = [&](int g, int& b, int& x0, int& ph) {
    const int s = g / nph;
    ph = g - s * nph;
    b = s / ncol;
    x0 = (s - b * ncol) * XMC;
  };
  auto pix_off = [&](int b, int x0, int y, int ch) {
    return (unsigned)b * (unsigned)ch * plane_b + (unsigned)(min(y, Ho - 1) * Wo + min(x0 + 16 * wc + j, Wo - 1)) * 4u;
  };
  // Tap registers: lane (j, kg) holds 7 PAIRS of neighbouring taps t0 + 8 a + {0, 1}, a = 0..6, so that a split pair
  // (x6_split2: low half = first tap, high half = second) is exactly the dword a table position takes -- 21 ds_write_b32 per
  // table where single taps needed 39-48 ds_write_b16 plus shifts and per-tap address arithmetic (PMC: those stores were
  // half of the kernel's LDS cycles).  v: t0 = 2 kg.  h: t0 = 2 kg - (j & 1): the band position i = tap + j of a pair then
  // starts even for every pixel.  Taps outside 0..50 are loaded from a clamped plane and zeroed by the table writers.
  auto load_taps = [&](float (&regs)[XNP][2], __amdgpu_buffer_rsrc_t src, int b, int x0, int y, int t0) {
    const unsigned pix = pix_off(b, x0, y, XK);
    const unsigned voff = pix + (unsigned)(t0 + 1) * plane_b;        // plane t0 + 1 >= 0: the range check sees this offset only
    regs[0][0] = x6_bload(src, pix + (unsigned)max(t0, 0) * plane_b, 0u);
    regs[0][1] = x6_bload(src, voff, 0u);
#pragma unroll
    for (int a = 1; a < XNP - 1; ++a)
#pragma unroll
      for (int e = 0; e < 2; ++e) regs[a][e] = x6_bload(src, voff, (unsigned)(8 * a + e - 1) * plane_b);
#pragma unroll
    for (int e = 0; e < 2; ++e) regs[XNP - 1][e] = x6_bload(src, pix + (unsigned)min(8 * (XNP - 1) + t0 + e, XK - 1) * plane_b, 0u);
  };
  auto tap_or_zero = [&](const float (&regs)[XNP][2], int a, int e, int t0) {
    if (a == 0 && e == 0) return t0 < 0 ? 0.f : regs[0][0];
    if (a == XNP - 1) return (8 * (XNP - 1) + t0 + e < XK) ? regs[a][e] : 0.f;
    return regs[a][e];
  };
  // h band of the wave's 16 pixels -> table [piece][i / 8][j][i % 8], i = tap + j < 64 (zero elsewhere)
  const int h_t0 = 2 * kg - (j & 1), v_t0 = 2 * kg;
  auto write_h_table = [&](const float (&regs)[XNP][2]) {
#pragma unroll
    for (int k = 0; k < XTAB / 1024; ++k) *reinterpret_cast<u32x4*>(tab + (k * 64 + lane) * 16) = (u32x4){0u, 0u, 0u, 0u};
    const int base2 = 2 * kg + (j & ~1);                             // position of the lane's first pair: even, <= 20
    char* const lb = tab + (base2 >> 3) * 256 + j * 16 + (base2 & 7) * 2;
#pragma unroll
    for (int a = 0; a < XNP; ++a) {
      unsigned h1, h2, h3;
      x6_split2(tap_or_zero(regs, a, 0, h_t0), tap_or_zero(regs, a, 1, h_t0), h1, h2, h3);
      char* d = lb + a * 256;
      if (a < XNP - 1 || base2 < 16) {                               // positions 64, 65 belong to the tail
        *reinterpret_cast<unsigned*>(d) = h1;
        *reinterpret_cast<unsigned*>(d + XTABP) = h2;
        *reinterpret_cast<unsigned*>(d + 2 * XTABP) = h3;
      }
    }
  };
  // v taps -> table position of tap fy: k step fy / 32, k group (fy % 16) / 4, element fy % 4 + 4 * ((fy / 16) % 2) (the
  // order in which gH's two transpose reads deliver the window rows); taps 51..63 are written as zeros
  auto write_v_table = [&](const float (&regs)[XNP][2]) {
    char* const lb = tab + (kg >> 1) * 256 + j * 16 + (kg & 1) * 4;
#pragma unroll
    for (int a = 0; a < 8; ++a) {
      unsigned h1 = 0u, h2 = 0u, h3 = 0u;
      if (a < XNP) x6_split2(tap_or_zero(regs, a, 0, v_t0), tap_or_zero(regs, a, 1, v_t0), h1, h2, h3);
      char* d = lb + (4 * (a >> 2) + 2 * (a & 1)) * 256 + 8 * ((a >> 1) & 1);
      *reinterpret_cast<unsigned*>(d) = h1;
      *reinterpret_cast<unsigned*>(d + XTABP) = h2;
      *reinterpret_cast<unsigned*>(d + 2 * XTABP) = h3;
    }
  };
  auto tr_read = [&](int addr) -> bf16x4 {
    typedef __attribute__((address_space(3))) bf16x4 lds_b4;
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_b4*)(smem + addr));
  };

  constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};   // six products, small terms first
  const int permk = ((kg & 1) << 1) | (kg >> 1);      // gV: k group kg <-> columns 8 permk .. (lane-group mates two blocks apart)
  const int L = lane & 15;
  const int pq = lane & 3, fq = lane >> 2;
  constexpr bool vec_store = VEC;      // Wo % 4 == 0 (the launcher picks)

  // A workgroup's phases form RUNS inside a strip.  Each run has a prologue (its first taps, the whole window, the first h band) and
  // a steady loop whose body issues a FIXED number of memory instructions -- the slide's loads included, wanted or not --, so that
  // the compiler can count vmcnt for the prefetched registers.  With the strip change and the slide loads behind branches inside
  // the loop it fell back to vmcnt(0) at the top of every phase: a wait for the previous phase's 32 stores to be acknowledged and,
  // every other phase, for six loads issued a few instructions earlier (experiment: the same kernel with its loads hitting cache
  // ran 50 us / 16 % faster).
  float hreg[XNP][2], vreg[XNP][2], gnext[XC];
  int g = g0;
#pragma unroll 1
  while (g < g1) {
  int b, x0, ph;
  pos_of(g, b, x0, ph);
  const int run_end = min(g1, g + (nph - ph)), ph_last = ph + (run_end - g) - 1;
  load_taps(hreg, hsrc, b, x0, XPR * ph + wr, h_t0);
  load_taps(vreg, vsrc, b, x0, XPR * ph + wr, v_t0);
  {
    const unsigned go = pix_off(b, x0, XPR * ph + wr, XC);
#pragma unroll
    for (int c = 0; c < XC; ++c) gnext[c] = x6_bload(gsrc, go, (unsigned)c * plane_b);
  }
  __syncthreads();                                  // every wave has left the previous run's window
#pragma unroll 1
  for (int r = 0; r < XWIN; r += 16) {
    X6Rows<16> sr;
    x6_rows_load<16>(sr, isrc, b, x0, XPR * ph + r, Hi, Wi, tid);
    x6_rows_write<16>(sr, smem, XPR * ph + r, tid);
  }
  int loaded_hi = XPR * ph + XWIN;
  write_h_table(hreg);
  __builtin_amdgcn_s_waitcnt(0);
  __syncthreads();
  if (w >= 4 && g == g0) __builtin_amdgcn_s_sleep(40);       // de-phase the two waves of a SIMD

#pragma unroll 1
  for (; g < run_end; ++g) {
    const int nb = b, nx0 = x0, nph_ = min(ph + 1, ph_last);      // the run's last phase re-reads its own taps
    const bool slide = (g + 1 < run_end) && (XPR * nph_ + XPR + XK - 1 > loaded_hi);
    X6Rows<XAHEAD> slid;
    x6_rows_load<XAHEAD>(slid, isrc, b, x0, loaded_hi, Hi, Wi, tid);      // unconditional (rows clamped): a fixed instruction count

    const int y = XPR * ph + wr;
    const int x = x0 + 16 * wc + j;
    const bool pvalid = (x < Wo) && (y < Ho);
    const unsigned opix_b = (unsigned)b * (unsigned)XK * plane_b + (unsigned)(min(y, Ho - 1) * Wo + min(x, Wo - 1)) * 4u;
    // vector stores (Wo % 4 == 0): lane = (pixel quad pq, tap row fq of a group of 16); a quad is wholly inside the map or outside
    const int xq = x0 + 16 * wc + 4 * pq;
    const unsigned qoff = (y < Ho && xq < Wo) ? (unsigned)b * (unsigned)XK * plane_b + (unsigned)(y * Wo + xq) * 4u + (unsigned)fq * plane_b : X_OOR;
    float g_[XC];
#pragma unroll
    for (int c = 0; c < XC; ++c) g_[c] = gnext[c];
    // what the two tail columns need of this phase's taps and cotangent, before the registers take the next phase's
    // pixel 14 (t0 = 2 kg): tap 50 = 48 + 2 -> lane (14, kg = 1), pair 6, first; pixel 15 (t0 = 2 kg - 1): taps 49, 50 = 48 + 1 + {0, 1} -> lane (15, 1)
    const float h50_14 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, hreg[6][0]), 14 + 16));
    const float h49_15 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, hreg[6][0]), 15 + 16));
    const float h50_15 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, hreg[6][1]), 15 + 16));
    float g14[XC], g15[XC];
#pragma unroll
    for (int c = 0; c < XC; ++c) {
      g14[c] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, g_[c]), 14));
      g15[c] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, g_[c]), 15));
    }

    // ---- gV -------------------------------------------------------------------------------------------------------
    bf16x8 bq[2][3];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int p = 0; p < 3; ++p) bq[s][p] = *reinterpret_cast<const bf16x8*>(tab + p * XTABP + (4 * s + permk) * 256 + j * 16);
    // next phase's h taps and cotangent: HBM -> registers (the table holds this phase's)
    load_taps(hreg, hsrc, nb, nx0, XPR * nph_ + wr, h_t0);
    {
      const unsigned go = pix_off(nb, nx0, XPR * nph_ + wr, XC);
#pragma unroll
      for (int c = 0; c < XC; ++c) gnext[c] = x6_bload(gsrc, go, (unsigned)c * plane_b);
    }

    const int fyl = min(lane, XK - 1);
    const int tslot = (y + fyl) & (XWIN - 1);
    float a64[XC], a65[XC];
#pragma unroll
    for (int c = 0; c < XC; ++c) {
      const f32x2 sv = *reinterpret_cast<const f32x2*>(side + (c * XWIN + tslot) * 4 + 2 * wc);
      a64[c] = sv.x; a65[c] = sv.y;
    }
    {   // columns i = 64 (pixel 14: tap 50; pixel 15: tap 49) and 65 (pixel 15: tap 50): lane = tap row fy
      float t14 = 0.f, t15 = 0.f;
#pragma unroll
      for (int c = 0; c < XC; ++c) {
        t14 = fmaf(g14[c] * h50_14, a64[c], t14);
        t15 = fmaf(g15[c] * h49_15, a64[c], t15);
        t15 = fmaf(g15[c] * h50_15, a65[c], t15);
      }
      tailb[lane] = lane < XK ? t14 : 0.f;          // [column][tap row]: a lane of the epilogue reads its four rows as one 16-byte piece
      tailb[64 + lane] = lane < XK ? t15 : 0.f;
    }

    {
      f32x4 acc[XC][4];
#pragma unroll
      for (int c = 0; c < XC; ++c)
#pragma unroll
        for (int m = 0; m < 4; ++m) acc[c][m] = (f32x4){0.f, 0.f, 0.f, 0.f};
      int rowoff[4];
#pragma unroll
      for (int m = 0; m < 4; ++m) rowoff[m] = ((y + min(16 * m + j, XK - 1)) & (XWIN - 1)) * 16 + (2 * wc + permk) * XBLK;
      bf16x8 aq[2][2][3];
      auto load_a = [&](int slot, int u) {
        const int c = u >> 2, s = (u >> 1) & 1, mp = u & 1;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int p = 0; p < 3; ++p)
            aq[slot][t][p] = *reinterpret_cast<const bf16x8*>(smem + (p * 3 + c) * XPLANE + 4 * s * XBLK + rowoff[2 * mp + t]);
      };
      __builtin_amdgcn_sched_barrier(0);
      X6_PRIO(X6_PRIO_LOOP);
      load_a(0, 0);
#pragma unroll
      for (int u = 0; u < 12; ++u) {
        if (u + 1 < 12) load_a((u + 1) & 1, u + 1);
        __builtin_amdgcn_sched_barrier(0);          // the next unit's fragments are requested BEFORE this unit's MFMAs issue
      __builtin_amdgcn_sched_barrier(0);          // the next unit's fragments are requested BEFORE this unit's MFMAs issue
        const int c = u >> 2, s = (u >> 1) & 1, mp = u & 1;
#pragma unroll
        for (int q = 0; q < 6; ++q)
#pragma unroll
          for (int t = 0; t < 2; ++t)
            acc[c][2 * mp + t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(aq[u & 1][t][PA[q]], bq[s][PB[q]], acc[c][2 * mp + t], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
      X6_PRIO(0);
      // D row 4 kg + r of tile m = tap row fy = 16 m + 4 kg + r, column = pixel j; pixels 14, 15 add their tail columns
      const float tsel = j >= 14 ? 1.f : 0.f;
      const int tcol = j == 15 ? 1 : 0;
      float val[4][4];
#pragma unroll
      for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float t = g_[0] * acc[0][m][r];
          t = fmaf(g_[1], acc[1][m][r], t);
          t = fmaf(g_[2], acc[2][m][r], t);
          val[m][r] = t;
        }
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        const f32x4 tv = *reinterpret_cast<const f32x4*>(tailb + 64 * tcol + 16 * m + 4 * kg);
#pragma unroll
        for (int r = 0; r < 4; ++r) val[m][r] = fmaf(tsel, tv[r], val[m][r]);
      }
      if constexpr (vec_store) {
        // rows of 4 pixels per lane through the wave's table (free: the h fragments are in registers, v is written below): a store
        // instruction then carries sixteen 64-byte runs as 16-byte pieces instead of four as dwords -- 4 instructions instead of 15
        X6_ORDER();
        float* tile = reinterpret_cast<float*>(tab);
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
          for (int r = 0; r < 4; ++r) tile[(16 * m + 4 * kg + r) * XTP + j] = val[m][r];
        X6_ORDER();
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const f32x4 v4 = *reinterpret_cast<const f32x4*>(tile + (fq + 16 * q) * XTP + 4 * pq);
          x6_bstore4(v4, gvdst, (q < 3 || fq < 3) ? qoff : X_OOR, (unsigned)(16 * q) * plane_b);
        }
        X6_ORDER();
      } else {
        const unsigned vo = opix_b + (unsigned)(4 * kg) * plane_b;
        const unsigned voA = pvalid ? vo : X_OOR, voB = (pvalid && kg == 0) ? vo : X_OOR;
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
          for (int r = 0; r < (m < 3 ? 4 : 3); ++r) x6_bstore(val[m][r], gvdst, m < 3 ? voA : voB, (unsigned)(16 * m + r) * plane_b);
      }
    }
    // the table takes v now, and the v registers the next phase's taps
    X6_ORDER();
    write_v_table(vreg);
    X6_ORDER();
    load_taps(vreg, vsrc, nb, nx0, XPR * nph_ + wr, v_t0);

    // ---- gH -------------------------------------------------------------------------------------------------------
    {
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int p = 0; p < 3; ++p) bq[s][p] = *reinterpret_cast<const bf16x8*>(tab + p * XTABP + (4 * s + kg) * 256 + j * 16);
      // tail columns q = 64, 65 (taps 50 / 49, 50 of pixels 14, 15): lane = tap row fy, v from the table's pieces
      float s6414 = 0.f, s6415 = 0.f, s6515 = 0.f;
      {
        const int f5 = fyl & 31;
        const char* tp = tab + (4 * (fyl >> 5) + ((f5 & 15) >> 2)) * 256 + ((f5 & 3) + 4 * ((f5 >> 4) & 1)) * 2;
        float v14 = 0.f, v15 = 0.f;
#pragma unroll
        for (int p = 0; p < 3; ++p) {
          v14 += __uint_as_float((unsigned)*reinterpret_cast<const unsigned short*>(tp + p * XTABP + 14 * 16) << 16);
          v15 += __uint_as_float((unsigned)*reinterpret_cast<const unsigned short*>(tp + p * XTABP + 15 * 16) << 16);
        }
        const float live = lane < XK ? 1.f : 0.f;
        v14 *= live; v15 *= live;
#pragma unroll
        for (int c = 0; c < XC; ++c) {
          s6414 = fmaf(g14[c] * v14, a64[c], s6414);
          s6415 = fmaf(g15[c] * v15, a64[c], s6415);
          s6515 = fmaf(g15[c] * v15, a65[c], s6515);
        }
        s6414 = wave_sum(s6414);
        s6415 = wave_sum(s6415);
        s6515 = wave_sum(s6515);
      }
      f32x4 acc[XC][4];
#pragma unroll
      for (int c = 0; c < XC; ++c)
#pragma unroll
        for (int m = 0; m < 4; ++m) acc[c][m] = (f32x4){0.f, 0.f, 0.f, 0.f};
      // transpose read: source lane L of group kg points at window row y + 32 s + 16 half + 4 kg + L / 4, column quad L % 4 of the tile
      int rowh[2][2];
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int hf = 0; hf < 2; ++hf)
          rowh[s][hf] = ((y + 32 * s + 16 * hf + 4 * kg + (L >> 2)) & (XWIN - 1)) * 16 + (2 * wc + ((L & 3) >> 1)) * XBLK + (L & 1) * 8;
      bf16x8 aq[2][2][3];
      auto load_a = [&](int slot, int u) {
        const int c = u >> 2, s = (u >> 1) & 1, mp = u & 1;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int p = 0; p < 3; ++p) {
            const int base = (p * 3 + c) * XPLANE + 2 * (2 * mp + t) * XBLK;
            const bf16x4 lo = tr_read(base + rowh[s][0]), hi = tr_read(base + rowh[s][1]);
            aq[slot][t][p] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
          }
      };
      __builtin_amdgcn_sched_barrier(0);
      X6_PRIO(X6_PRIO_LOOP);
      load_a(0, 0);
#pragma unroll
      for (int u = 0; u < 12; ++u) {
        if (u + 1 < 12) load_a((u + 1) & 1, u + 1);
        __builtin_amdgcn_sched_barrier(0);          // the next unit's fragments are requested BEFORE this unit's MFMAs issue
      __builtin_amdgcn_sched_barrier(0);          // the next unit's fragments are requested BEFORE this unit's MFMAs issue
        const int c = u >> 2, s = (u >> 1) & 1, mp = u & 1;
#pragma unroll
        for (int q = 0; q < 6; ++q)
#pragma unroll
          for (int t = 0; t < 2; ++t)
            acc[c][2 * mp + t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(aq[u & 1][t][PA[q]], bq[s][PB[q]], acc[c][2 * mp + t], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
      X6_PRIO(0);
      // gH[fx][j] = D[j + fx][j]: through a 64 x 16 fp32 tile in the wave's (now dead) tap table, out as 13 x four 64-byte runs
      {
        X6_ORDER();
        float* tile = reinterpret_cast<float*>(tab);
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float val = g_[0] * acc[0][m][r];
            val = fmaf(g_[1], acc[1][m][r], val);
            val = fmaf(g_[2], acc[2][m][r], val);
            tile[(16 * m + 4 * kg + r) * 16 + j] = val;
          }
        X6_ORDER();
        if constexpr (vec_store) {
          // lane (pq, fq): pixels 4 pq .. + 3 of tap fx = fq + 16 q; the three tail sums go into their places (pixels 14 / 15, taps 49 / 50)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int fx = fq + 16 * q;
            f32x4 v4;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const int jj = 4 * pq + e;
              float t = tile[min(jj + fx, 63) * 16 + jj];
              if (jj + fx >= 64) t = (jj == 14) ? s6414 : (fx == 49 ? s6415 : s6515);
              v4[e] = t;
            }
            x6_bstore4(v4, ghdst, (q < 3 || fq < 3) ? qoff : X_OOR, (unsigned)(16 * q) * plane_b);
          }
        } else {
          const unsigned ho = opix_b + (unsigned)kg * plane_b;
#pragma unroll
          for (int t = 0; t < XNREG; ++t) {
            const int fx = 4 * t + kg;
            const float val = tile[min(j + fx, 63) * 16 + j];
            x6_bstore(val, ghdst, (pvalid && fx < XK && j + fx < 64) ? ho : X_OOR, (unsigned)(4 * t) * plane_b);
          }
        }
        X6_ORDER();
      }
      if constexpr (!vec_store) {
        x6_bstore(s6414, ghdst, (pvalid && lane == 14) ? opix_b : X_OOR, 50u * plane_b);
        x6_bstore(s6415, ghdst, (pvalid && lane == 15) ? opix_b : X_OOR, 49u * plane_b);
        x6_bstore(s6515, ghdst, (pvalid && lane == 15) ? opix_b : X_OOR, 50u * plane_b);
      }
    }

    // the next phase's h band takes the table
    write_h_table(hreg);
    X6_ORDER();
    if (slide) {                                   // the eight oldest rows are behind every wave's next phase
      __syncthreads();
      x6_rows_write<XAHEAD>(slid, smem, loaded_hi, tid);
      loaded_hi += XAHEAD;
      __syncthreads();
    }
    ph = nph_;
  }
  }
}


// ------------------------------------------------------------------------------------------------------------------------
// Forward on the same machinery:  out[b,c,y,x] = sum_fy v[b,fy,y,x] * T_c[fy],   T_c[fy][j] = sum_i In_c[y + fy][16 wc + i] * Hb[i][j]
// T is the gV product above without the cotangent (144 MFMAs per 16 pixels; the fp32-MFMA forward issues 170 of twice the
// cycles), its tail columns i = 64, 65 per channel, and the vertical pass runs on the accumulators: lane (j, kg) holds rows
// fy = 16 m + 4 kg + r of T and loads v for exactly those rows; the four row groups of a pixel meet through two lane exchanges.
// Replaces the reference's forward kernel (sepconv/sepconv_op/sepconv.py:5-30) -- same op, same layout.
// ------------------------------------------------------------------------------------------------------------------------
template <bool VEC>
__global__ __launch_bounds__(XNT) void sepconv_fwd_x6(const float* __restrict__ in, const float* __restrict__ v,
                                                      const float* __restrict__ h, float* __restrict__ out,
                                                      int B, int Ho, int Wo, int nph, int ncol, int per_wg) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wc = w & 1, wr = w >> 1;
  const int j = lane & 15, kg = lane >> 4;
  char* const tab = smem + XWINB + w * XTAB;
  float* const side = reinterpret_cast<float*>(smem + XSIDE_OFF);

  const int total = B * ncol * nph;
  const int g0 = blockIdx.x * per_wg, g1 = min(g0 + per_wg, total);
  if (g0 >= g1) return;
  const int Hi = Ho + XK - 1, Wi = Wo + XK - 1;
  const unsigned plane_b = (unsigned)Ho * (unsigned)Wo * 4u;
  const __amdgpu_buffer_rsrc_t hsrc = x6_rsrc(h, (unsigned)(B * XK) * plane_b);
  const __amdgpu_buffer_rsrc_t vsrc = x6_rsrc(v, (unsigned)(B * XK) * plane_b);
  const __amdgpu_buffer_rsrc_t isrc = x6_rsrc(in, (unsigned)(B * XC) * (unsigned)(Hi * Wi) * 4u);
  const __amdgpu_buffer_rsrc_t odst = x6_rsrc(out, (unsigned)(B * XC) * plane_b);

  auto pos_of = [&](int g, int& b, int& x0, int& ph) {
    const int s = g / nph;
    ph = g - s * nph;
    b = s / ncol;
    x0 = (s - b * ncol) * XMC;
  };
  auto pix_off = [&](int b, int x0, int y, int ch) {
    return (unsigned)b * (unsigned)ch * plane_b + (unsigned)(min(y, Ho - 1) * Wo + min(x0 + 16 * wc + j, Wo - 1)) * 4u;
  };
  const int h_t0 = 2 * kg - (j & 1);
  auto load_h = [&](float (&regs)[XNP][2], int b, int x0, int y) {          // the pair layout of the backward's h taps
    const unsigned pix = pix_off(b, x0, y, XK);
    const unsigned voff = pix + (unsigned)(h_t0 + 1) * plane_b;
    regs[0][0] = x6_bload(hsrc, pix + (unsigned)max(h_t0, 0) * plane_b, 0u);
    regs[0][1] = x6_bload(hsrc, voff, 0u);
#pragma unroll
    for (int a = 1; a < XNP - 1; ++a)
#pragma unroll
      for (int e = 0; e < 2; ++e) regs[a][e] = x6_bload(hsrc, voff, (unsigned)(8 * a + e - 1) * plane_b);
#pragma unroll
    for (int e = 0; e < 2; ++e) regs[XNP - 1][e] = x6_bload(hsrc, pix + (unsigned)min(8 * (XNP - 1) + h_t0 + e, XK - 1) * plane_b, 0u);
  };
  auto h_or_zero = [&](const float (&regs)[XNP][2], int a, int e) {
    if (a == 0 && e == 0) return h_t0 < 0 ? 0.f : regs[0][0];
    if (a == XNP - 1) return (8 * (XNP - 1) + h_t0 + e < XK) ? regs[a][e] : 0.f;
    return regs[a][e];
  };
  auto write_h_table = [&](const float (&regs)[XNP][2]) {
#pragma unroll
    for (int k = 0; k < XTAB / 1024; ++k) *reinterpret_cast<u32x4*>(tab + (k * 64 + lane) * 16) = (u32x4){0u, 0u, 0u, 0u};
    const int base2 = 2 * kg + (j & ~1);
    char* const lb = tab + (base2 >> 3) * 256 + j * 16 + (base2 & 7) * 2;
#pragma unroll
    for (int a = 0; a < XNP; ++a) {
      unsigned h1, h2, h3;
      x6_split2(h_or_zero(regs, a, 0), h_or_zero(regs, a, 1), h1, h2, h3);
      char* d = lb + a * 256;
      if (a < XNP - 1 || base2 < 16) {
        *reinterpret_cast<unsigned*>(d) = h1;
        *reinterpret_cast<unsigned*>(d + XTABP) = h2;
        *reinterpret_cast<unsigned*>(d + 2 * XTABP) = h3;
      }
    }
  };
  // v is wanted in the accumulator layout, vcur[m][r] = v[j][16 m + 4 kg + r].  Widths that are a multiple of 4: lane (pq, fq) loads
  // pixels 4 pq .. + 3 of tap fq + 16 q as ONE 16-byte piece (4 load instructions of sixteen 64-byte runs instead of 16 of four) and
  // the layout change goes through a 52 x 16 tile (pitch 20) behind the tail sums in the wave's table; other widths load the
  // accumulator layout directly.  Rows >= 51: clamped plane, zeroed at use.
  constexpr bool vec_v = VEC;          // Wo % 4 == 0 (the launcher picks)
  const int pq = lane & 3, fq = lane >> 2;
  auto load_v = [&](float (&regs)[4][4], int b, int x0, int y) {
    if constexpr (vec_v) {
      const unsigned base = (unsigned)b * (unsigned)XK * plane_b + (unsigned)(min(y, Ho - 1) * Wo + min(x0 + 16 * wc + 4 * pq, Wo - 4)) * 4u;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const u32x4 t = __builtin_amdgcn_raw_buffer_load_b128(vsrc, (int)(base + (unsigned)min(fq + 16 * q, XK - 1) * plane_b), 0, 0);
#pragma unroll
        for (int e = 0; e < 4; ++e) regs[q][e] = __uint_as_float(t[e]);
      }
      return;
    }
    const unsigned pix = pix_off(b, x0, y, XK);
    const unsigned voff = pix + (unsigned)(4 * kg) * plane_b;
#pragma unroll
    for (int m = 0; m < 3; ++m)
#pragma unroll
      for (int r = 0; r < 4; ++r) regs[m][r] = x6_bload(vsrc, voff, (unsigned)(16 * m + r) * plane_b);
#pragma unroll
    for (int r = 0; r < 4; ++r) regs[3][r] = x6_bload(vsrc, pix + (unsigned)min(48 + 4 * kg + r, XK - 1) * plane_b, 0u);
  };

  constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};
  const int permk = ((kg & 1) << 1) | (kg >> 1);
  float* const tailb = reinterpret_cast<float*>(tab);                  // [c][column][64 tap rows] in the wave's table once its fragments are in registers

  float hreg[XNP][2], vD[4][4];
  int g = g0;
#pragma unroll 1
  while (g < g1) {          // runs of phases inside a strip: see sepconv_bwd_x6
  int b, x0, ph;
  pos_of(g, b, x0, ph);
  const int run_end = min(g1, g + (nph - ph)), ph_last = ph + (run_end - g) - 1;
  load_h(hreg, b, x0, XPR * ph + wr);
  load_v(vD, b, x0, XPR * ph + wr);
  __syncthreads();
#pragma unroll 1
  for (int r = 0; r < XWIN; r += 16) {
    X6Rows<16> sr;
    x6_rows_load<16>(sr, isrc, b, x0, XPR * ph + r, Hi, Wi, tid);
    x6_rows_write<16>(sr, smem, XPR * ph + r, tid);
  }
  int loaded_hi = XPR * ph + XWIN;
  write_h_table(hreg);
  __builtin_amdgcn_s_waitcnt(0);
  __syncthreads();
  if (w >= 4 && g == g0) __builtin_amdgcn_s_sleep(40);

#pragma unroll 1
  for (; g < run_end; ++g) {
    const int nb = b, nx0 = x0, nph_ = min(ph + 1, ph_last);
    const bool slide = (g + 1 < run_end) && (XPR * nph_ + XPR + XK - 1 > loaded_hi);
    X6Rows<XAHEAD> slid;
    x6_rows_load<XAHEAD>(slid, isrc, b, x0, loaded_hi, Hi, Wi, tid);

    const int y = XPR * ph + wr;
    const int x = x0 + 16 * wc + j;
    const bool pvalid = (x < Wo) && (y < Ho);
    const float h50_14 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, hreg[6][0]), 14 + 16));
    const float h49_15 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, hreg[6][0]), 15 + 16));
    const float h50_15 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, hreg[6][1]), 15 + 16));
    bf16x8 bq[2][3];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int p = 0; p < 3; ++p) bq[s][p] = *reinterpret_cast<const bf16x8*>(tab + p * XTABP + (4 * s + permk) * 256 + j * 16);
    float vcur[4][4];
    if constexpr (vec_v) {      // pixel quads per tap -> taps per pixel, through the table (its h fragments are in registers now)
      float* const vt = reinterpret_cast<float*>(tab + 6 * 64 * 4);
      X6_ORDER();
#pragma unroll
      for (int q = 0; q < 4; ++q)
        if (q < 3 || fq < 4) *reinterpret_cast<f32x4*>(vt + (fq + 16 * q) * XTP + 4 * pq) = (f32x4){vD[q][0], vD[q][1], vD[q][2], vD[q][3]};
      X6_ORDER();
#pragma unroll
      for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float t = vt[(16 * m + 4 * kg + r) * XTP + j];        // rows >= 52 of the last tile lie past the tile: read, never used
          vcur[m][r] = (m < 3 || (kg == 0 && r < 3)) ? t : 0.f;
        }
      X6_ORDER();
    } else {
#pragma unroll
      for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) vcur[m][r] = (m < 3 || (kg == 0 && r < 3)) ? vD[m][r] : 0.f;
    }
    load_h(hreg, nb, nx0, XPR * nph_ + wr);
    load_v(vD, nb, nx0, XPR * nph_ + wr);
    X6_ORDER();
    {   // tail columns per channel: lane = tap row fy
      const int fyl = min(lane, XK - 1);
      const int tslot = (y + fyl) & (XWIN - 1);
#pragma unroll
      for (int c = 0; c < XC; ++c) {
        const f32x2 sv = *reinterpret_cast<const f32x2*>(side + (c * XWIN + tslot) * 4 + 2 * wc);
        tailb[(2 * c) * 64 + lane] = lane < XK ? sv.x * h50_14 : 0.f;
        tailb[(2 * c + 1) * 64 + lane] = lane < XK ? fmaf(sv.y, h50_15, sv.x * h49_15) : 0.f;
      }
    }
    X6_ORDER();

    f32x4 acc[XC][4];
#pragma unroll
    for (int c = 0; c < XC; ++c)
#pragma unroll
      for (int m = 0; m < 4; ++m) acc[c][m] = (f32x4){0.f, 0.f, 0.f, 0.f};
    int rowoff[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) rowoff[m] = ((y + min(16 * m + j, XK - 1)) & (XWIN - 1)) * 16 + (2 * wc + permk) * XBLK;
    bf16x8 aq[2][2][3];
    auto load_a = [&](int slot, int u) {
      const int c = u >> 2, s = (u >> 1) & 1, mp = u & 1;
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int p = 0; p < 3; ++p)
          aq[slot][t][p] = *reinterpret_cast<const bf16x8*>(smem + (p * 3 + c) * XPLANE + 4 * s * XBLK + rowoff[2 * mp + t]);
    };
    __builtin_amdgcn_sched_barrier(0);
    X6_PRIO(X6_PRIO_LOOP);
    load_a(0, 0);
#pragma unroll
    for (int u = 0; u < 12; ++u) {
      if (u + 1 < 12) load_a((u + 1) & 1, u + 1);
      __builtin_amdgcn_sched_barrier(0);
      const int c = u >> 2, s = (u >> 1) & 1, mp = u & 1;
#pragma unroll
      for (int q = 0; q < 6; ++q)
#pragma unroll
        for (int t = 0; t < 2; ++t)
          acc[c][2 * mp + t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(aq[u & 1][t][PA[q]], bq[s][PB[q]], acc[c][2 * mp + t], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    X6_PRIO(0);
    // vertical pass: this lane's 16 rows of T (+ the tail columns for pixels 14, 15), then the four row groups of the pixel
    const float tsel = j >= 14 ? 1.f : 0.f;
    const int tcol = j == 15 ? 1 : 0;
    float o[XC];
#pragma unroll
    for (int c = 0; c < XC; ++c) {
      float sum = 0.f;
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        const f32x4 tv = *reinterpret_cast<const f32x4*>(tailb + (2 * c + tcol) * 64 + 16 * m + 4 * kg);
#pragma unroll
        for (int r = 0; r < 4; ++r) sum = fmaf(vcur[m][r], fmaf(tsel, tv[r], acc[c][m][r]), sum);
      }
      sum += __shfl_xor(sum, 16, SAVFI_WAVE);
      sum += __shfl_xor(sum, 32, SAVFI_WAVE);
      o[c] = sum;
    }
    {   // lane group kg stores channel kg
      const float val = kg == 0 ? o[0] : kg == 1 ? o[1] : o[2];
      const unsigned oo = (unsigned)b * (unsigned)XC * plane_b + (unsigned)(min(y, Ho - 1) * Wo + min(x, Wo - 1)) * 4u + (unsigned)kg * plane_b;
      x6_bstore(val, odst, (pvalid && kg < XC) ? oo : X_OOR, 0u);
    }
    X6_ORDER();
    write_h_table(hreg);
    X6_ORDER();
    if (slide) {
      __syncthreads();
      x6_rows_write<XAHEAD>(slid, smem, loaded_hi, tid);
      loaded_hi += XAHEAD;
      __syncthreads();
    }
    ph = nph_;
  }
  }
}

}  // namespace

// gV and gH of the K = 51, C = 3 op; every tensor below 2^31 bytes (the caller checks).  Declared in csrc/common.h.
int savfi_sepconv_bwd_x6_launch(const float* in, const float* v, const float* h, const float* gO, float* gV, float* gH, int B, int Ho,
                                int Wo, int cus, hipStream_t st) {
  const int nph = savfi_cdiv(Ho, XPR), ncol = savfi_cdiv(Wo, XMC);
  const int64_t total = (int64_t)B * ncol * nph;
  const int per_wg = savfi_cdiv(total, cus);
  const int grid = savfi_cdiv(total, per_wg);
  if ((Wo & 3) == 0 && !X6_SCALAR_STORES) {
    static uint32_t done = 0;
    if (int e = savfi_ensure_dynamic_lds((const void*)sepconv_bwd_x6<true>, XLDS, done)) return e;
    hipLaunchKernelGGL(sepconv_bwd_x6<true>, dim3(grid), dim3(XNT), XLDS, st, in, v, h, gO, gV, gH, B, Ho, Wo, nph, ncol, per_wg);
  } else {
    static uint32_t done = 0;
    if (int e = savfi_ensure_dynamic_lds((const void*)sepconv_bwd_x6<false>, XLDS, done)) return e;
    hipLaunchKernelGGL(sepconv_bwd_x6<false>, dim3(grid), dim3(XNT), XLDS, st, in, v, h, gO, gV, gH, B, Ho, Wo, nph, ncol, per_wg);
  }
  return savfi_launch_status();
}

// forward of the same op (declared in csrc/common.h)
int savfi_sepconv_fwd_x6_launch(const float* in, const float* v, const float* h, float* out, int B, int Ho, int Wo, int cus,
                                hipStream_t st) {
  const int nph = savfi_cdiv(Ho, XPR), ncol = savfi_cdiv(Wo, XMC);
  const int64_t total = (int64_t)B * ncol * nph;
  const int per_wg = savfi_cdiv(total, cus);
  const int grid = savfi_cdiv(total, per_wg);
  if ((Wo & 3) == 0 && !X6_SCALAR_STORES) {
    static uint32_t done = 0;
    if (int e = savfi_ensure_dynamic_lds((const void*)sepconv_fwd_x6<true>, XLDS, done)) return e;
    hipLaunchKernelGGL(sepconv_fwd_x6<true>, dim3(grid), dim3(XNT), XLDS, st, in, v, h, out, B, Ho, Wo, nph, ncol, per_wg);
  } else {
    static uint32_t done = 0;
    if (int e = savfi_ensure_dynamic_lds((const void*)sepconv_fwd_x6<false>, XLDS, done)) return e;
    hipLaunchKernelGGL(sepconv_fwd_x6<false>, dim3(grid), dim3(XNT), XLDS, st, in, v, h, out, B, Ho, Wo, nph, ncol, per_wg);
  }
  return savfi_launch_status();
}
