// Separable local convolution (SepConv) for gfx950 -- forward, filter gradients, input gradient.
//
// Replaces the four cupy/NVRTC kernels of the reference
// (sepconv/sepconv_op/sepconv.py:5-30, :32-63, :138-163, :165-190).
//
//   out[b,c,y,x] = sum_fy sum_fx in[b,c,y+fy,x+fx] * v[b,fy,y,x] * h[b,fx,y,x]
//
// Layout: fp32 NCHW, contiguous.  in [B,C,Ho+K-1,Wo+K-1], v/h [B,K,Ho,Wo], out/gO [B,C,Ho,Wo].
//
// K = 51, C = 3 (the only shape the model uses): sepconv_fwd_mfma / sepconv_bwd_mfma below -- the horizontal pass as a
// banded GEMM on the exact-fp32 matrix cores, the input window of R rows x 64 columns (+50 halo) staged once in LDS,
// wave-private tap rows, no workgroup barrier in the row loop.  Every other (K, C) takes the generic direct kernels at the
// end of the file (one thread per output element; cold path, kept for the op's full surface).
// The earlier VALU generations (one / two pixels per thread, LDS-bandwidth bound: 163 / 114 us forward at 384x512 against
// 52 us here) were removed in round 2; DESIGN.md section 4 keeps their numbers.
#include "common.h"
#include <stdlib.h>

// Compile-time experiment switches (both on in the shipped library; profiles/r02_sepconv_variants.txt has the A/B numbers):
//   SC_PIN     pin the LDS-read / MFMA interleave with sched_group_barrier (software-pipelined A fragments)
//   SC_GH_LDS  gH leaves through an LDS transpose as 64-byte runs instead of 64 scattered dwords per store instruction
#ifndef SC_PIN
#define SC_PIN 1
#endif
#ifndef SC_GH_LDS
#define SC_GH_LDS 1
#endif

namespace {

constexpr int KFAST = 51;


// Stage an LH x SPAN window of a [Hi, Wi] plane (top-left at (y0, x0)) into LDS rows of pitch LW.
// Thread -> (column q = tid % 128, row group tid / 128): a thread keeps its column and walks rows in steps of NTHREADS / 128,
// so a load is one raw buffer load (per-lane byte offset = clamped row * Wi + clamped column: one v_min + one v_mad) and an
// LDS write is a constant offset from a per-thread base -- ~3 VALU per element where the flattened index (i / SPAN, i % SPAN,
// 64-bit address) cost ~30: the PMC profile of round 2 showed the staging prologue issuing more VALU than the whole row loop,
// on a datapath the fp32 MFMAs share.  All loads of a thread are issued back to back, the ds_writes follow.  Coordinates
// past the plane are clamped -- such entries only ever feed output pixels that lie outside the image and are never stored.
template <int LH, int SPAN, int LW, int NTHREADS>
__device__ __forceinline__ void stage_window(float* __restrict__ tile, const float* __restrict__ src, int y0,
                                             int x0, int Hi, int Wi, int tid) {
  static_assert(SPAN <= 128 && NTHREADS % 128 == 0, "one column per thread");
  constexpr int RG = NTHREADS / 128, NIT = (LH + RG - 1) / RG;
  const int q = tid & 127, rg = tid >> 7;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src), 0, Hi * Wi * 4, 0x00020000);
  const int colb = min(x0 + q, Wi - 1) * 4, rowb = Wi * 4;
  float buf[NIT];
#pragma unroll
  for (int it = 0; it < NIT; ++it)
    buf[it] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, min(y0 + rg + it * RG, Hi - 1) * rowb + colb, 0, 0));
  float* dst = tile + rg * LW + q;
  if (q < SPAN) {
#pragma unroll
    for (int it = 0; it < NIT; ++it)
      if (it * RG + RG <= LH || rg + it * RG < LH) dst[it * RG * LW] = buf[it];
  }
}


// ------------------------------------------------------------------------------------------
// K = 51 fast path, third generation: the horizontal pass on the matrix cores (exact fp32 MFMA).
//
// For 16 pixels of one output row (y, x0..x0+15) the horizontal pass is a small GEMM
//     T[(c,fy), p] = sum_q  In[c][y+fy][x0+q] * Hb[q][p],      Hb[q][p] = h[q-p][y][x0+p]  (0 <= q-p < 51)
// with M = 3*51 = 153 rows (c,fy), K = 16+50 = 66 input columns, N = 16 pixels: a banded ("Toeplitz
// with per-pixel taps") B operand.  v_mfma_f32_16x16x4_f32 is bit-for-bit an fp32 fmaf chain at the
// fp32 VALU rate, but it needs ONE ds_read_b32 per 1024 FMA instead of one LDS float per 1-4 FMA, which
// is what bounds the VALU kernels above (LDS bandwidth, not FMA issue).  The band wastes 68/51 of the
// MFMA work (170 MFMA = 32 cycles each per 16 pixels -> 340 cycles/pixel/SIMD vs 249 for ideal VALU).
// The vertical pass  out[c,p] = sum_fy v[fy,p] * T[(c,fy),p]  runs on the accumulator registers.
//
// Workgroup = 4 waves = a 4-row x 64-column output tile; wave w owns columns 16w..16w+15 and walks the
// 4 rows.  LDS: the 3 x 54 x 116 input window (pitch 132: 16-byte aligned A fragments) + double-buffered [51][64] h and v rows, which
// are fetched from HBM one row ahead with fully coalesced 256-byte segments.
// ------------------------------------------------------------------------------------------
typedef float f32x4 __attribute__((ext_vector_type(4)));

// geometry of the MFMA kernels: 8 waves = 2 row lanes x 4 column groups of 16 pixels; MROWS output rows
// per workgroup are processed in phases of 2 rows.
// MROWS is a template parameter (8 / 12 / 16) picked per launch so that the grid fills the 256 CUs in as few
// rounds as possible (1 workgroup per CU: the window + tap rows take ~150 KB of LDS).
constexpr int MC = 64, MLW = 132, MSPAN = 116, MNT = 512;
constexpr int MKP = 52;   // rows per channel in the M dimension (51 taps + 1 zero row): lanes never straddle channels

// Tap rows are PRIVATE to a wave: wave (wr, wc) needs h / v of its own 16 pixels only ([K][16] floats = 64-byte
// runs per tap plane).  Keeping them private removes every workgroup barrier from the row loop, so the two waves
// that share a SIMD drift out of phase and one's VALU / LDS / store phases hide under the other's MFMAs.
// lane = (tap group tg = lane >> 4, column j = lane & 15); NREG = ceil(K / 4) dwords per lane.
// Raw buffer loads: one resource per sample's [K][Ho][Wo] tap tensor, a per-lane byte offset (tap group + pixel) computed once
// per row and a wave-uniform byte offset per tap quadruple -- no 64-bit per-load addresses on the VALU (26 of them per row
// before), and tap 51 of the last quadruple lies past num_records = K planes, so the hardware returns 0 for it.
__device__ __forceinline__ float sc_bload(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)voff, (int)soff, 0));
}
// store with the same addressing; a lane whose offset is >= num_records (SC_OOR) writes nothing: predication without a branch,
// so a row's stores are a FIXED number of memory instructions and the compiler can count vmcnt for the loads around them
constexpr unsigned SC_OOR = 0x80000000u;
__device__ __forceinline__ void sc_bstore(float val, __amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
  __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, val), r, (int)voff, (int)soff, 0);
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t sc_rsrc(const float* base, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, (int)bytes, 0x00020000);
}
template <int K, int NREG>
__device__ __forceinline__ void mfma_load_taps(float (&regs)[NREG], __amdgpu_buffer_rsrc_t src, unsigned plane_bytes,
                                               int Ho, int Wo, int y, int xw, int lane) {
  const int yy = min(y, Ho - 1), xx = min(xw + (lane & 15), Wo - 1);
  const unsigned voff = (unsigned)(lane >> 4) * plane_bytes + (unsigned)(yy * Wo + xx) * 4u;
#pragma unroll
  for (int it = 0; it < NREG; ++it) regs[it] = sc_bload(src, voff, (unsigned)(4 * it) * plane_bytes);
}
template <int K, int NREG>
__device__ __forceinline__ void mfma_store_taps(float* __restrict__ dst /* [>=K][16] */, const float (&regs)[NREG],
                                                int lane) {
  // Branch-free: row K (one past the last tap) is written with zeros.  For the h rows that is row 0 of the v rows right behind
  // them, which the v copy that always follows overwrites; for the v rows it is the zero tap row itself.  (A lane-predicated
  // write made the compiler wait for vmcnt(0) -- every outstanding gradient store -- in front of the last quadruple.)
#pragma unroll
  for (int it = 0; it < NREG; ++it) {
    const int tap = 4 * it + (lane >> 4);
    if (4 * it + 3 < K) dst[tap * 16 + (lane & 15)] = regs[it];
    else dst[min(tap, K) * 16 + (lane & 15)] = tap < K ? regs[it] : 0.f;
  }
}

template <int K, int MROWS>
__global__ __launch_bounds__(MNT) void sepconv_fwd_mfma(const float* __restrict__ in, const float* __restrict__ v,
                                                          const float* __restrict__ h, float* __restrict__ out,
                                                          int Ho, int Wo) {
  constexpr int C = 3, LH = MROWS + K - 1, LP = LH * MLW;
  constexpr int M = C * MKP, MT = (M + 15) / 16, KT = (16 + K - 1 + 3) / 4;   // 156 rows -> 10 tiles; 17 k-steps
  constexpr int NREG = (K + 3) / 4;                                            // 13 dwords per lane per tap array
  static_assert(K < MKP && MKP % 4 == 0 && MT % 2 == 0, "M layout");
  static_assert(KT == 17 && 16 * 3 + 4 * KT <= MSPAN && MSPAN <= MLW && MLW % 4 == 0 && MROWS % 2 == 0, "A fragment layout");
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* inT = lds;                       // [C][LH][MLW]
  float* hB = lds + C * LP + (threadIdx.x >> 6) * (K + MKP) * 16;   // this wave's [K][16] h taps ...
  float* vB = hB + K * 16;                                  // ... and [MKP][16] v taps (row K is zero)

  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int wc = w & 3, wr = w >> 2;      // column group, row lane
  const int j = lane & 15, ks = lane >> 4;
  const int x0 = blockIdx.x * MC, y0 = blockIdx.y * MROWS, b = blockIdx.z;
  const int Hi = Ho + K - 1, Wi = Wo + K - 1;
  const size_t plane = (size_t)Ho * Wo;
  const unsigned plane_b = (unsigned)plane * 4u;
  const __amdgpu_buffer_rsrc_t hsrc = sc_rsrc(h + (size_t)b * K * plane, (unsigned)K * plane_b);
  const __amdgpu_buffer_rsrc_t vsrc = sc_rsrc(v + (size_t)b * K * plane, (unsigned)K * plane_b);
  const __amdgpu_buffer_rsrc_t osrc = sc_rsrc(out + (size_t)b * C * plane, (unsigned)C * plane_b);

  float hreg[NREG], vreg[NREG];
  mfma_load_taps<K, NREG>(hreg, hsrc, plane_b, Ho, Wo, y0 + wr, x0 + 16 * wc, lane);
  mfma_load_taps<K, NREG>(vreg, vsrc, plane_b, Ho, Wo, y0 + wr, x0 + 16 * wc, lane);
#pragma unroll
  for (int c = 0; c < C; ++c)
    stage_window<LH, MSPAN, MLW, MNT>(inT + c * LP, in + ((size_t)b * C + c) * Hi * Wi, y0, x0, Hi, Wi, tid);
  if (lane < 16) vB[K * 16 + lane] = 0.f;                                       // the zero tap row
  mfma_store_taps<K, NREG>(hB, hreg, lane);
  mfma_store_taps<K, NREG>(vB, vreg, lane);

  // per-lane A-row bases: M index mi = 16*m + j -> (c, fy) = (mi / 52, mi % 52); rows fy = 51 and mi >= 156
  // are padding (their T rows meet the zero tap row / are never read) and clamp to valid LDS rows
  int abase[MT];
#pragma unroll
  for (int m = 0; m < MT; ++m) {
    const int mi = min(16 * m + j, M - 1);
    const int c = mi / MKP, fy = min(mi - c * MKP, K - 1);
    abase[m] = (c * LH + fy + wr) * MLW + 16 * wc + 4 * ks;
  }
  __syncthreads();
  // The two waves of a SIMD (row lanes wr = 0 / 1) run identical instruction streams and would stay in
  // lockstep: both in their VALU / LDS / store phases, then both competing for the matrix pipe.  A one-off
  // ~2.5k-cycle head start for one of them lets each wave's non-MFMA work hide under the other's MFMAs
  // (measured: fwd 47.9 -> 46.1 us, bwd 105.5 -> 96.4 us at 384x512).
  if (wr == 1) __builtin_amdgcn_s_sleep(40);

#pragma unroll
  for (int ph = 0; ph < MROWS / 2; ++ph) {
    const int y = y0 + 2 * ph + wr;
    if (ph + 1 < MROWS / 2) {   // next row's taps: HBM -> registers while this row computes
      mfma_load_taps<K, NREG>(hreg, hsrc, plane_b, Ho, Wo, y + 2, x0 + 16 * wc, lane);
      mfma_load_taps<K, NREG>(vreg, vsrc, plane_b, Ho, Wo, y + 2, x0 + 16 * wc, lane);
    }
    const float* hb = hB + j;
    const float* vb = vB + j;

    // The order in which window columns q are assigned to the MFMA k-slots is free as long as A and B agree.
    // Slot (step s, lane group ks) takes column q = 16*(s/4) + 4*ks + s%4 for s < 16 and q = 64 + ks for s = 16, so
    // that a lane's 16 A values of an M-tile are four aligned 16-byte LDS reads (+ one dword) instead of 17 dwords:
    // the LDS read instructions, not their latency, were what held the MFMA pipe at 50 %.
    float bf[KT];
#pragma unroll
    for (int t = 0; t < KT; ++t) {
      const int q = (t < 16) ? (16 * (t >> 2) + 4 * ks + (t & 3)) : (64 + ks);
      const int tap = q - j;
      const float val = hb[min(max(tap, 0), K - 1) * 16];
      bf[t] = (tap >= 0 && tap < K) ? val : 0.f;
    }

    float o0 = 0.f, o1 = 0.f, o2 = 0.f;
    const int rowoff = 2 * ph * MLW;
    // A fragments of the NEXT pair of M-tiles are in flight while this pair's 34 MFMAs issue (the compiler, left alone, sinks
    // every LDS read to just before its first use: each pair then starts with an exposed LDS round trip)
    auto load_pair = [&](f32x4 (&d0)[4], f32x4 (&d1)[4], float& t0, float& t1, int mp) {
      const float* a0p = inT + abase[2 * mp] + rowoff;
      const float* a1p = inT + abase[2 * mp + 1] + rowoff;
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        d0[u] = *reinterpret_cast<const f32x4*>(a0p + 16 * u);
        d1[u] = *reinterpret_cast<const f32x4*>(a1p + 16 * u);
      }
      t0 = a0p[64 - 3 * ks];                                             // column 64 + ks (the base holds + 4*ks)
      t1 = a1p[64 - 3 * ks];
    };
    f32x4 a0[4], a1[4], n0[4], n1[4];
    float a0t, a1t, n0t, n1t;
#if SC_PIN
    __builtin_amdgcn_sched_barrier(0);
#endif
    load_pair(a0, a1, a0t, a1t, 0);
#if SC_PIN
    __builtin_amdgcn_sched_group_barrier(0x100, 10, 0);
#endif
#pragma unroll
    for (int mp = 0; mp < MT / 2; ++mp) {
      f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
      if (mp + 1 < MT / 2) load_pair(n0, n1, n0t, n1t, mp + 1);
#pragma unroll
      for (int t = 0; t < 16; ++t) {
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[t >> 2][t & 3], bf[t], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[t >> 2][t & 3], bf[t], acc1, 0, 0, 0);
      }
      acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0t, bf[16], acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1t, bf[16], acc1, 0, 0, 0);
#if SC_PIN
      if (mp + 1 < MT / 2) __builtin_amdgcn_sched_group_barrier(0x100, 10, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 34, 0);
#endif
      // vertical pass on the accumulators: lane holds T rows 16*m + 4*ks + e (e = 0..3) of pixel j, all of
      // one channel (52 = 4*13), at taps fy0..fy0+3 (tap 51 is the zero row)
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        const f32x4 acc = half ? acc1 : acc0;
        const int mi0 = 16 * (2 * mp + half) + 4 * ks;
        const int c = (mi0 >= MKP) + (mi0 >= 2 * MKP);
        const int fy0 = min(mi0 - c * MKP, MKP - 4);
        const float* vp = vb + fy0 * 16;
        float sdot = vp[0] * acc[0];
        sdot = fmaf(vp[16], acc[1], sdot);
        sdot = fmaf(vp[32], acc[2], sdot);
        sdot = fmaf(vp[48], acc[3], sdot);
        sdot = (mi0 < M) ? sdot : 0.f;
        o0 += (c == 0) ? sdot : 0.f;
        o1 += (c == 1) ? sdot : 0.f;
        o2 += (c == 2) ? sdot : 0.f;
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) { a0[u] = n0[u]; a1[u] = n1[u]; }
      a0t = n0t;
      a1t = n1t;
    }
    // the 4 k-lanes of a pixel hold disjoint row subsets: fold them
    o0 += __shfl_xor(o0, 16, 64); o0 += __shfl_xor(o0, 32, 64);
    o1 += __shfl_xor(o1, 16, 64); o1 += __shfl_xor(o1, 32, 64);
    o2 += __shfl_xor(o2, 16, 64); o2 += __shfl_xor(o2, 32, 64);
    {   // branch-free stores (see sc_bstore): a fixed three per row, so the tap prefetch above is waited for with an exact count
      const int x = x0 + 16 * wc + j;
      const unsigned oo = (ks == 0 && x < Wo && y < Ho) ? (unsigned)(y * Wo + x) * 4u : SC_OOR;
      sc_bstore(o0, osrc, oo, 0u);
      sc_bstore(o1, osrc, oo, plane_b);
      sc_bstore(o2, osrc, oo, 2u * plane_b);
    }

    if (ph + 1 < MROWS / 2) {
      // wave-private buffers: the LDS queue of a wave is in order, so no workgroup barrier is needed
      __builtin_amdgcn_wave_barrier();
      mfma_store_taps<K, NREG>(hB, hreg, lane);
      mfma_store_taps<K, NREG>(vB, vreg, lane);
      __builtin_amdgcn_wave_barrier();
    }
  }
}


// ------------------------------------------------------------------------------------------
// Filter gradients on the matrix cores.  Same workgroup geometry, LDS window and tap-row pipeline as
// sepconv_fwd_mfma; per 16-pixel tile two banded GEMMs share the staged input window:
//
//   gV[fy,p] = sum_{c,q} In[c][y+fy][x0+q] * (gO[c,p] * Hb[q][p])            M = fy (51 -> 64), K = 3*68
//   D'[q,p]  = sum_{c,fy} In[c][y+fy][x0+q] * (gO[c,p] * v[fy,p])            M = q (66 -> 80),  K = 3*52
//   gH[fx,p] = D'[p+fx, p]                                                   (the band of D')
//
// i.e. the upstream gradient is folded into the B operand, so the accumulators ARE the results (no
// cross-lane reduction): 204 + 195 MFMA per tile.  gV rows leave as 64-byte runs per tap plane; the gH
// band is stored element-wise (each lane owns a different tap plane: 16 stores of 4 B per instruction --
// ~10 % of the MFMA time, and L2 merges the sectors before they reach HBM).
// ------------------------------------------------------------------------------------------
template <int K, int MROWS, bool WANT_V, bool WANT_H>
__global__ __launch_bounds__(MNT) void sepconv_bwd_mfma(const float* __restrict__ in, const float* __restrict__ v,
                                                       const float* __restrict__ h, const float* __restrict__ gO,
                                                       float* __restrict__ gV, float* __restrict__ gH,
                                                       int Ho, int Wo) {
  constexpr int C = 3, LH = MROWS + K - 1, LP = LH * MLW;
  constexpr int KT = (16 + K - 1 + 3) / 4;                 // 17 column steps of the banded H operand
  constexpr int MTV = (K + 15) / 16;                       // 4 M-tiles of taps fy
  constexpr int KTV = (K + 3) / 4;                         // 13 row steps (taps fy = 4t+ks; tap 51 is the zero row)
  constexpr int MTH = (16 + K - 1 + 15) / 16;              // 5 M-tiles of window columns q
  constexpr int NREG = (K + 3) / 4;
  static_assert(KT == 17 && MTV == 4 && MLW % 4 == 0 && 4 * KTV == MKP && 16 * 3 + 16 * MTH <= MLW, "operand geometry");
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* inT = lds;
  float* hB = lds + C * LP + (threadIdx.x >> 6) * (K + MKP) * 16;   // wave-private tap rows (see the forward kernel)
  float* vB = hB + K * 16;

  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int wc = w & 3, wr = w >> 2;
  const int j = lane & 15, ks = lane >> 4;
  const int x0 = blockIdx.x * MC, y0 = blockIdx.y * MROWS, b = blockIdx.z;
  const int Hi = Ho + K - 1, Wi = Wo + K - 1;
  const size_t plane = (size_t)Ho * Wo;
  const unsigned plane_b = (unsigned)plane * 4u;
  const __amdgpu_buffer_rsrc_t hsrc = sc_rsrc(h + (size_t)b * K * plane, (unsigned)K * plane_b);
  const __amdgpu_buffer_rsrc_t vsrc = sc_rsrc(v + (size_t)b * K * plane, (unsigned)K * plane_b);
  const __amdgpu_buffer_rsrc_t gsrc = sc_rsrc(gO + (size_t)b * C * plane, (unsigned)C * plane_b);
  const __amdgpu_buffer_rsrc_t gvdst = sc_rsrc(gV + (size_t)b * K * plane, WANT_V ? (unsigned)K * plane_b : 0u);
  const __amdgpu_buffer_rsrc_t ghdst = sc_rsrc(gH + (size_t)b * K * plane, WANT_H ? (unsigned)K * plane_b : 0u);
  // byte offset of this lane's pixel in row y of a plane (clamped: out-of-image pixels are computed and never stored)
  auto pix_off = [&](int y) { return (unsigned)(min(y, Ho - 1) * Wo + min(x0 + 16 * wc + j, Wo - 1)) * 4u; };

  float hreg[NREG], vreg[NREG], gnext[C];
  mfma_load_taps<K, NREG>(hreg, hsrc, plane_b, Ho, Wo, y0 + wr, x0 + 16 * wc, lane);
  mfma_load_taps<K, NREG>(vreg, vsrc, plane_b, Ho, Wo, y0 + wr, x0 + 16 * wc, lane);
#pragma unroll
  for (int c = 0; c < C; ++c) gnext[c] = sc_bload(gsrc, pix_off(y0 + wr), (unsigned)c * plane_b);
#pragma unroll
  for (int c = 0; c < C; ++c)
    stage_window<LH, MSPAN, MLW, MNT>(inT + c * LP, in + ((size_t)b * C + c) * Hi * Wi, y0, x0, Hi, Wi, tid);
  // columns MSPAN..MLW-1 of the window are only read by gH rows q >= 66, which are discarded, but keep them finite
  static_assert(MLW - MSPAN == 16, "zero-fill indexing");
  for (int i = tid; i < C * LH * 16; i += MNT) inT[(i >> 4) * MLW + MSPAN + (i & 15)] = 0.f;
  if (lane < 16) vB[K * 16 + lane] = 0.f;
  mfma_store_taps<K, NREG>(hB, hreg, lane);
  mfma_store_taps<K, NREG>(vB, vreg, lane);

  // A-row bases.  gV: M index = tap fy = 16m + j (clamped), column = k (b128 k-slot order, see the forward kernel).
  // gH: M index = window column q, k index = tap row 4t + ks.  The assignment of window columns to (M-tile m, row i) is free
  // as long as the store below agrees: q = 4 i + m puts the four M-tiles' A values of a lane (i = j) into ONE aligned
  // 16-byte LDS read -- 39 ds_read_b128 per row instead of 156 dword reads, each fetched two steps ahead of its MFMAs.
  int abV[MTV];
#pragma unroll
  for (int m = 0; m < MTV; ++m) abV[m] = (min(16 * m + j, K - 1) + wr) * MLW + 16 * wc + 4 * ks;
  const int abH = (wr + ks) * MLW + 16 * wc + 4 * j;
  __syncthreads();
  // The two waves of a SIMD (row lanes wr = 0 / 1) run identical instruction streams and would stay in lockstep: a one-off
  // head start for one of them lets each wave's non-MFMA phases hide under the other's MFMAs.
  if (wr == 1) __builtin_amdgcn_s_sleep(40);

#pragma unroll 1
  for (int ph = 0; ph < MROWS / 2; ++ph) {
    const int y = y0 + 2 * ph + wr;
    const int x = x0 + 16 * wc + j;
    const bool pvalid = (x < Wo) && (y < Ho);
    const unsigned opix_b = (unsigned)(min(y, Ho - 1) * Wo + min(x, Wo - 1)) * 4u;   // this lane's pixel inside a tap plane
    float g[C];
#pragma unroll
    for (int c = 0; c < C; ++c) g[c] = gnext[c];              // fetched during the previous row
    if (ph + 1 < MROWS / 2) {                                   // next row's taps and upstream gradient: HBM -> registers
      mfma_load_taps<K, NREG>(hreg, hsrc, plane_b, Ho, Wo, y + 2, x0 + 16 * wc, lane);
      mfma_load_taps<K, NREG>(vreg, vsrc, plane_b, Ho, Wo, y + 2, x0 + 16 * wc, lane);
#pragma unroll
      for (int c = 0; c < C; ++c) gnext[c] = sc_bload(gsrc, pix_off(y + 2), (unsigned)c * plane_b);
    }
    const float* hb = hB + j;
    const float* vb = vB + j;
    const int rowoff = 2 * ph * MLW;

    if (WANT_V) {
      float bf[KT];      // same k-slot order as the forward kernel: q = 16*(t/4) + 4*ks + t%4, last step q = 64 + ks
#pragma unroll
      for (int t = 0; t < KT; ++t) {
        const int q = (t < 16) ? (16 * (t >> 2) + 4 * ks + (t & 3)) : (64 + ks);
        const int tap = q - j;
        const float val = hb[min(max(tap, 0), K - 1) * 16];
        bf[t] = (tap >= 0 && tap < K) ? val : 0.f;
      }
      f32x4 acc[MTV];
#pragma unroll
      for (int m = 0; m < MTV; ++m) acc[m] = (f32x4){0.f, 0.f, 0.f, 0.f};
      // groups of 4 k-steps (one 16-byte A read per M-tile) + the tail step per channel; the next group's A fragments are
      // in flight while this group's 16 MFMAs issue
      constexpr int NG = C * 5;            // per channel: u = 0..3 (4 steps each), u = 4: the single step t = 16
      auto load_group = [&](f32x4 (&dst)[MTV], int gi) {
        const int c = gi / 5, u = gi - 5 * c;
#pragma unroll
        for (int m = 0; m < MTV; ++m) {
          const float* ap = inT + abV[m] + rowoff + c * LP;
          if (u < 4) dst[m] = *reinterpret_cast<const f32x4*>(ap + 16 * u);
          else dst[m][0] = ap[64 - 3 * ks];
        }
      };
      f32x4 acur[MTV], anxt[MTV];
      // The schedule is pinned (sched_group_barrier): left alone, the compiler sinks every LDS read to just before its first
      // use and each group of MFMAs then starts with an exposed LDS round trip.
#if SC_PIN
      __builtin_amdgcn_sched_barrier(0);
#endif
      load_group(acur, 0);
#if SC_PIN
      __builtin_amdgcn_sched_group_barrier(0x100, MTV, 0);
#endif
#pragma unroll
      for (int gi = 0; gi < NG; ++gi) {
        if (gi + 1 < NG) load_group(anxt, gi + 1);
        const int c = gi / 5, u = gi - 5 * c;
#pragma unroll
        for (int e = 0; e < (u < 4 ? 4 : 1); ++e) {
          const float bb = g[c] * bf[u < 4 ? 4 * u + e : 16];
#pragma unroll
          for (int m = 0; m < MTV; ++m)
            acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(acur[m][e], bb, acc[m], 0, 0, 0);
        }
#pragma unroll
        for (int m = 0; m < MTV; ++m) acur[m] = anxt[m];
#if SC_PIN
        // next group's A fragments first, then this group's B products and MFMAs
        if (gi + 1 < NG) __builtin_amdgcn_sched_group_barrier(0x100, MTV, 0);
        if (u < 4) {
          __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
          __builtin_amdgcn_sched_group_barrier(0x008, 4 * MTV, 0);
        } else {
          __builtin_amdgcn_sched_group_barrier(0x002, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x008, MTV, 0);
        }
#endif
      }
#if SC_PIN
      __builtin_amdgcn_sched_barrier(0);
#endif
      {   // branch-free stores (see sc_bstore); fy = 16 m + 4 ks + e: all real for m < 3, for m = 3 only ks = 0, e < 3
        static_assert(K == 51 && MTV == 4, "gV store predication");
        const unsigned vo = opix_b + (unsigned)(4 * ks) * plane_b;
        const unsigned voA = pvalid ? vo : SC_OOR, voB = (pvalid && ks == 0) ? vo : SC_OOR;
#pragma unroll
        for (int m = 0; m < MTV; ++m)
#pragma unroll
          for (int e = 0; e < (m < 3 ? 4 : 3); ++e) sc_bstore(acc[m][e], gvdst, m < 3 ? voA : voB, (unsigned)(16 * m + e) * plane_b);
      }
    }

    if (WANT_H) {
      // Window columns q = 0..63 go through the matrix cores (4 M-tiles).  Of the fifth tile (q = 64..79) only three
      // entries lie inside the band 0 <= q - j <= 50: (q, j) = (64, 14), (64, 15), (65, 15); they are three 153-term dot
      // products on the VALU below instead of 39 more MFMAs per row (10 % of this kernel's matrix work).
      constexpr int MTHM = MTH - 1;
      static_assert(K == 51 && 16 * MTHM == 64, "band geometry of the VALU tail");
      float bv[KTV];
#pragma unroll
      for (int t = 0; t < KTV; ++t) bv[t] = vb[(4 * t + ks) * 16];           // tap 51 reads the zero row
      f32x4 acc[MTHM];
#pragma unroll
      for (int m = 0; m < MTHM; ++m) acc[m] = (f32x4){0.f, 0.f, 0.f, 0.f};
      // step (c, t): window row of tap 4t + ks; the zero tap (51, lane group ks = 3 of the last step) would be one row past
      // the window: step back one row (its B operand is zero)
      const float* abase = inT + abH + rowoff;
      const int last_row = ((ks == 3) ? (4 * (KTV - 1) - 1) : 4 * (KTV - 1)) * MLW;
      auto a_of = [&](int idx) {
        const int c = idx / KTV, t = idx - KTV * c;
        return *reinterpret_cast<const f32x4*>(abase + c * LP + (t == KTV - 1 ? last_row : 4 * t * MLW));
      };
      constexpr int NS = C * KTV;
#if SC_PIN
      __builtin_amdgcn_sched_barrier(0);
#endif
      f32x4 a0 = a_of(0), a1 = a_of(1);
#if SC_PIN
      __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
#endif
#pragma unroll
      for (int idx = 0; idx < NS; ++idx) {
        f32x4 a2 = a0;
        if (idx + 2 < NS) a2 = a_of(idx + 2);
        const int c = idx / KTV, t = idx - KTV * c;
        const float bb = g[c] * bv[t];
#pragma unroll
        for (int m = 0; m < MTHM; ++m) acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[m], bb, acc[m], 0, 0, 0);
        a0 = a1;
        a1 = a2;
#if SC_PIN
        if (idx + 2 < NS) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);     // the A read of step idx + 2 ...
        __builtin_amdgcn_sched_group_barrier(0x002, 1, 0);                        // ... this step's B product ...
        __builtin_amdgcn_sched_group_barrier(0x008, MTHM, 0);                     // ... and its 4 MFMAs
#endif
      }
#if SC_PIN
      __builtin_amdgcn_sched_barrier(0);
#endif
      // VALU tail: lane = tap row fy (lanes 51..63 idle), then a 64-lane sum of the three partial products
      float s6414 = 0.f, s6415 = 0.f, s6515 = 0.f;
      {
        const int fy = min(lane, K - 1);
        const float live = lane < K ? 1.f : 0.f;
        const float v14 = vB[fy * 16 + 14] * live, v15 = vB[fy * 16 + 15] * live;
        const float* col = inT + rowoff + (wr + fy) * MLW + 16 * wc + 64;
#pragma unroll
        for (int c = 0; c < C; ++c) {
          const float a64 = col[c * LP], a65 = col[c * LP + 1];
          const float g14 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, g[c]), 14));
          const float g15 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, g[c]), 15));
          s6414 = fmaf(g14 * v14, a64, s6414);
          s6415 = fmaf(g15 * v15, a64, s6415);
          s6515 = fmaf(g15 * v15, a65, s6515);
        }
        s6414 = wave_sum(s6414);
        s6415 = wave_sum(s6415);
        s6515 = wave_sum(s6515);
      }
      // accumulator (m, e) of lane (j, ks) is D'[q][j] with q = 4 * (4 ks + e) + m  (the M permutation chosen above)
#if SC_GH_LDS
      // gH[fx][p] = D'[p + fx][p]: straight from the accumulators every lane of a store instruction would hit a different tap
      // plane (64 separate 4-byte writes per instruction, 64 instructions per row).  The wave's tap rows are dead from here
      // to the end of the row, so the 64 x 16 tile goes through them and leaves as 13 instructions of four 64-byte runs.
      {
        __builtin_amdgcn_wave_barrier();
        float* tile = hB;                                   // [64 q][16 p] floats = 4 KB of the wave's 6.4 KB tap buffer
#pragma unroll
        for (int m = 0; m < MTHM; ++m)
#pragma unroll
          for (int e = 0; e < 4; ++e) tile[(16 * ks + 4 * e + m) * 16 + j] = acc[m][e];
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int t = 0; t < KTV; ++t) {
          const int fx = 4 * t + ks;
          const float val = tile[min(j + fx, 63) * 16 + j];
          sc_bstore(val, ghdst, (pvalid && fx < K && j + fx < 64) ? opix_b + (unsigned)ks * plane_b : SC_OOR, (unsigned)(4 * t) * plane_b);
        }
        __builtin_amdgcn_wave_barrier();
      }
#else
#pragma unroll
      for (int m = 0; m < MTHM; ++m) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int fx = 16 * ks + 4 * e + m - j;
          sc_bstore(acc[m][e], ghdst, (pvalid && fx >= 0 && fx < K) ? opix_b + (unsigned)max(fx, 0) * plane_b : SC_OOR, 0u);
        }
      }
#endif
      sc_bstore(s6414, ghdst, (pvalid && lane == 14) ? opix_b : SC_OOR, 50u * plane_b);
      sc_bstore(s6415, ghdst, (pvalid && lane == 15) ? opix_b : SC_OOR, 49u * plane_b);
      sc_bstore(s6515, ghdst, (pvalid && lane == 15) ? opix_b : SC_OOR, 50u * plane_b);
    }

    if (ph + 1 < MROWS / 2) {
      __builtin_amdgcn_wave_barrier();
      mfma_store_taps<K, NREG>(hB, hreg, lane);
      mfma_store_taps<K, NREG>(vB, vreg, lane);
      __builtin_amdgcn_wave_barrier();
    }
  }
}

// ------------------------------------------------------------------------------------------
// Persistent form of the filter-gradient kernel (round 2).
//
// The tiled kernel above stages (R + 50) input rows for R output rows and its launch is a whole number of workgroup rounds:
// PMC showed 12 % of the kernel in the partly filled last round and ~8 % in staging prologues (one workgroup per CU).
// Here the launch has ONE workgroup per CU and the output is cut into "phases" (two output rows of one 64-column strip of
// one sample, the unit the 2 x 4 waves process), numbered strip-major; workgroup w takes the phases [w Q, (w + 1) Q): equal
// shares, no tail.  Consecutive phases of a strip slide down by two rows, so the input window lives in a CIRCULAR buffer of 64
// rows per channel (slot = input row & 63): a phase reads rows 2 ph .. 2 ph + 51, twelve further rows are staged every sixth
// phase (two barriers), the full 64 only when a workgroup enters a strip.  Compute per phase is that of sepconv_bwd_mfma.
// ------------------------------------------------------------------------------------------
constexpr int PWIN = 64;                       // window rows per channel (power of two >= K + 1)
constexpr int PAHEAD = PWIN - (KFAST + 1);     // rows staged per refill (12)

// rows [r_lo, r_lo + nrows) of the strip (b, x0) -> their circular slots, all three channels; thread -> (column, row group).
// Two halves so that the slide (twelve rows every sixth phase) can be in flight for a whole phase: loads -> registers, then,
// behind a barrier, registers -> LDS.
template <int NROWS>
struct StagedRows {
  static constexpr int C = 3, RG = MNT / 128, NIT = (NROWS + RG - 1) / RG;
  float buf[C][NIT];
};
template <int NROWS>
__device__ __forceinline__ void stage_rows_load(StagedRows<NROWS>& sr, __amdgpu_buffer_rsrc_t in_rs, int b, int x0, int r_lo,
                                                int Hi, int Wi, int tid) {
  constexpr int C = 3, RG = MNT / 128, NIT = StagedRows<NROWS>::NIT;
  const int q = tid & 127, rg = tid >> 7;
  const int colb = min(x0 + q, Wi - 1) * 4;
#pragma unroll
  for (int c = 0; c < C; ++c)
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int r = min(r_lo + rg + it * RG, Hi - 1);
      sr.buf[c][it] = sc_bload(in_rs, (unsigned)(((b * C + c) * Hi + r) * Wi * 4 + colb), 0u);
    }
}
template <int NROWS>
__device__ __forceinline__ void stage_rows_write(const StagedRows<NROWS>& sr, float* __restrict__ inT, int r_lo, int tid) {
  constexpr int C = 3, RG = MNT / 128, NIT = StagedRows<NROWS>::NIT, LP = PWIN * MLW;
  const int q = tid & 127, rg = tid >> 7;
  // branch-free: all 128 columns are written.  Columns MSPAN .. 127 of a window row are only read by gH rows that are
  // discarded (they must merely stay finite, and image data is), and no lane skips the reads of its staging registers -- a
  // skipped lane leaves its loads "possibly in flight" for the compiler, which then waits for vmcnt(0) wherever those
  // registers are reused.
  static_assert(128 <= MLW, "staged columns fit a window row");
#pragma unroll
  for (int c = 0; c < C; ++c)
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int rr = rg + it * RG;
      if (rr < NROWS) inT[c * LP + ((r_lo + rr) & (PWIN - 1)) * MLW + q] = sr.buf[c][it];
    }
}
template <int NROWS>
__device__ __forceinline__ void stage_rows_circular(float* __restrict__ inT, __amdgpu_buffer_rsrc_t in_rs, int b, int x0, int r_lo,
                                                    int Hi, int Wi, int tid) {
  StagedRows<NROWS> sr;
  stage_rows_load<NROWS>(sr, in_rs, b, x0, r_lo, Hi, Wi, tid);
  stage_rows_write<NROWS>(sr, inT, r_lo, tid);
}

template <int K, bool WANT_V, bool WANT_H>
__global__ __launch_bounds__(MNT) void sepconv_bwd_mfma_p(const float* __restrict__ in, const float* __restrict__ v,
                                                         const float* __restrict__ h, const float* __restrict__ gO,
                                                         float* __restrict__ gV, float* __restrict__ gH,
                                                         int B, int Ho, int Wo, int nph, int ncol, int per_wg) {
  constexpr int C = 3, LP = PWIN * MLW;
  constexpr int KT = (16 + K - 1 + 3) / 4, MTV = (K + 15) / 16, KTV = (K + 3) / 4, MTH = (16 + K - 1 + 15) / 16, NREG = (K + 3) / 4;
  static_assert(KT == 17 && MTV == 4 && MLW % 4 == 0 && 4 * KTV == MKP && 16 * 3 + 16 * MTH <= MLW && K + 1 <= PWIN, "operand geometry");
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* inT = lds;
  float* hB = lds + C * LP + (threadIdx.x >> 6) * (K + MKP) * 16;
  float* vB = hB + K * 16;

  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int wc = w & 3, wr = w >> 2;
  const int j = lane & 15, ks = lane >> 4;
  const int total = B * ncol * nph;
  const int g0 = blockIdx.x * per_wg, g1 = min(g0 + per_wg, total);
  if (g0 >= g1) return;
  const int Hi = Ho + K - 1, Wi = Wo + K - 1;
  const size_t plane = (size_t)Ho * Wo;
  const unsigned plane_b = (unsigned)plane * 4u;
  // whole-tensor resources (the launcher checks that they fit 2^31 bytes); the sample goes into the per-lane offset
  const __amdgpu_buffer_rsrc_t hsrc = sc_rsrc(h, (unsigned)(B * K) * plane_b);
  const __amdgpu_buffer_rsrc_t vsrc = sc_rsrc(v, (unsigned)(B * K) * plane_b);
  const __amdgpu_buffer_rsrc_t gsrc = sc_rsrc(gO, (unsigned)(B * C) * plane_b);
  const __amdgpu_buffer_rsrc_t isrc = sc_rsrc(in, (unsigned)(B * C) * (unsigned)(Hi * Wi) * 4u);
  const __amdgpu_buffer_rsrc_t gvdst = sc_rsrc(gV, WANT_V ? (unsigned)(B * K) * plane_b : 0u);
  const __amdgpu_buffer_rsrc_t ghdst = sc_rsrc(gH, WANT_H ? (unsigned)(B * K) * plane_b : 0u);

  auto pos_of = [&](int g, int& b, int& x0, int& ph) {
    const int s = g / nph;
    ph = g - s * nph;
    b = s / ncol;
    x0 = (s - b * ncol) * MC;
  };
  // byte offset of this lane's pixel (row y of sample b, clamped) inside a [B][*][Ho][Wo] tensor with `ch` planes per sample
  auto pix_off = [&](int b, int x0, int y, int ch) {
    return (unsigned)b * (unsigned)ch * plane_b + (unsigned)(min(y, Ho - 1) * Wo + min(x0 + 16 * wc + j, Wo - 1)) * 4u;
  };
  auto load_taps = [&](float (&regs)[NREG], __amdgpu_buffer_rsrc_t src, int b, int x0, int y) {
    const unsigned voff = pix_off(b, x0, y, K) + (unsigned)(lane >> 4) * plane_b;
#pragma unroll
    for (int it = 0; it < NREG; ++it) regs[it] = sc_bload(src, voff, (unsigned)(4 * it) * plane_b);
  };

  int b, x0, ph;
  pos_of(g0, b, x0, ph);
  float hreg[NREG], vreg[NREG], gnext[C];
  load_taps(hreg, hsrc, b, x0, 2 * ph + wr);
  load_taps(vreg, vsrc, b, x0, 2 * ph + wr);
  {
    const unsigned go = pix_off(b, x0, 2 * ph + wr, C);
#pragma unroll
    for (int c = 0; c < C; ++c) gnext[c] = sc_bload(gsrc, go, (unsigned)c * plane_b);
  }
  stage_rows_circular<PWIN>(inT, isrc, b, x0, 2 * ph, Hi, Wi, tid);
  int loaded_hi = 2 * ph + PWIN;
  // columns MSPAN..MLW-1 are only read by gH rows q >= 66, which are discarded, but keep them finite (never restaged)
  static_assert(MLW - MSPAN == 16, "zero-fill indexing");
  for (int i = tid; i < C * PWIN * 16; i += MNT) inT[(i >> 4) * MLW + MSPAN + (i & 15)] = 0.f;
  if (lane < 16) vB[K * 16 + lane] = 0.f;
  mfma_store_taps<K, NREG>(hB, hreg, lane);
  mfma_store_taps<K, NREG>(vB, vreg, lane);
  __syncthreads();
  if (wr == 1) __builtin_amdgcn_s_sleep(40);      // de-phase the two waves of a SIMD (see the tiled kernel)

#pragma unroll 1
  for (int g = g0; g < g1; ++g) {
    // ---- window upkeep (workgroup-uniform decisions) ------------------------------------------------------------
    if (g != g0 && ph == 0) {                      // entered the next strip: the whole window is new
      __syncthreads();
#pragma unroll 1
      for (int r = 0; r < PWIN; r += 16)           // in four pieces: 12 staging registers instead of 48 (rare path)
        stage_rows_circular<16>(inT, isrc, b, x0, r, Hi, Wi, tid);
      loaded_hi = PWIN;
      __syncthreads();
    }
    // next phase (position and whether it needs the window slid by twelve rows first)
    int nb, nx0, nph_;
    pos_of(min(g + 1, g1 - 1), nb, nx0, nph_);
    const bool slide = (g + 1 < g1) && nph_ != 0 && (2 * nph_ + K + 1 > loaded_hi);
    // The slide's rows travel HBM -> registers during this phase and registers -> LDS behind the barrier at its end: issued
    // first, they are older than the tap prefetch, so the exact vmcnt wait of the tap copy covers them and nobody waits for
    // memory with the matrix pipe idle (before: two barriers around a load + wait, every sixth phase, with all eight waves idle).
    StagedRows<PAHEAD> slid;
    if (slide) stage_rows_load<PAHEAD>(slid, isrc, b, x0, loaded_hi, Hi, Wi, tid);
    const int y = 2 * ph + wr;
    const int x = x0 + 16 * wc + j;
    const bool pvalid = (x < Wo) && (y < Ho);
    // byte offset of this lane's pixel in tap plane 0 of sample b (gV / gH are [B][K][Ho][Wo]); out of range = no store
    const unsigned opix_b = (unsigned)b * (unsigned)K * plane_b + (unsigned)(min(y, Ho - 1) * Wo + min(x, Wo - 1)) * 4u;
    float g_[C];
#pragma unroll
    for (int c = 0; c < C; ++c) g_[c] = gnext[c];
    // next phase's taps and upstream gradient: HBM -> registers.  Unconditional (the last phase re-reads itself), and the
    // stores below are a fixed number of instructions, so the wait in front of the LDS copy at the end of the phase is an
    // exact vmcnt(#stores) -- not a wait for every store's write acknowledge.
    load_taps(hreg, hsrc, nb, nx0, 2 * nph_ + wr);
    load_taps(vreg, vsrc, nb, nx0, 2 * nph_ + wr);
    {
      const unsigned go = pix_off(nb, nx0, 2 * nph_ + wr, C);
#pragma unroll
      for (int c = 0; c < C; ++c) gnext[c] = sc_bload(gsrc, go, (unsigned)c * plane_b);
    }
    const float* hb = hB + j;
    const float* vb = vB + j;

    if (WANT_V) {
      float bf[KT];
#pragma unroll
      for (int t = 0; t < KT; ++t) {
        const int q = (t < 16) ? (16 * (t >> 2) + 4 * ks + (t & 3)) : (64 + ks);
        const int tap = q - j;
        const float val = hb[min(max(tap, 0), K - 1) * 16];
        bf[t] = (tap >= 0 && tap < K) ? val : 0.f;
      }
      // A rows: tap fy = 16 m + j (clamped) of this wave's output row -> input row y + fy -> its circular slot
      int abV[MTV];
#pragma unroll
      for (int m = 0; m < MTV; ++m) abV[m] = ((y + min(16 * m + j, K - 1)) & (PWIN - 1)) * MLW + 16 * wc + 4 * ks;
      f32x4 acc[MTV];
#pragma unroll
      for (int m = 0; m < MTV; ++m) acc[m] = (f32x4){0.f, 0.f, 0.f, 0.f};
      constexpr int NG = C * 5;
      auto load_group = [&](f32x4 (&dst)[MTV], int gi) {
        const int c = gi / 5, u = gi - 5 * c;
#pragma unroll
        for (int m = 0; m < MTV; ++m) {
          const float* ap = inT + abV[m] + c * LP;
          if (u < 4) dst[m] = *reinterpret_cast<const f32x4*>(ap + 16 * u);
          else dst[m][0] = ap[64 - 3 * ks];
        }
      };
      f32x4 acur[MTV], anxt[MTV];
#if SC_PIN
      __builtin_amdgcn_sched_barrier(0);
#endif
      load_group(acur, 0);
#if SC_PIN
      __builtin_amdgcn_sched_group_barrier(0x100, MTV, 0);
#endif
#pragma unroll
      for (int gi = 0; gi < NG; ++gi) {
        if (gi + 1 < NG) load_group(anxt, gi + 1);
        const int c = gi / 5, u = gi - 5 * c;
#pragma unroll
        for (int e = 0; e < (u < 4 ? 4 : 1); ++e) {
          const float bb = g_[c] * bf[u < 4 ? 4 * u + e : 16];
#pragma unroll
          for (int m = 0; m < MTV; ++m)
            acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(acur[m][e], bb, acc[m], 0, 0, 0);
        }
#pragma unroll
        for (int m = 0; m < MTV; ++m) acur[m] = anxt[m];
#if SC_PIN
        if (gi + 1 < NG) __builtin_amdgcn_sched_group_barrier(0x100, MTV, 0);
        if (u < 4) {
          __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
          __builtin_amdgcn_sched_group_barrier(0x008, 4 * MTV, 0);
        } else {
          __builtin_amdgcn_sched_group_barrier(0x002, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x008, MTV, 0);
        }
#endif
      }
#if SC_PIN
      __builtin_amdgcn_sched_barrier(0);
#endif
      {   // fy = 16 m + 4 ks + e: all real for m < 3; for m = 3 only ks = 0, e < 3 (fy = 48, 49, 50)
        static_assert(K == 51 && MTV == 4, "gV store predication");
        const unsigned vo = opix_b + (unsigned)(4 * ks) * plane_b;
        const unsigned voA = pvalid ? vo : SC_OOR, voB = (pvalid && ks == 0) ? vo : SC_OOR;
#pragma unroll
        for (int m = 0; m < MTV; ++m)
#pragma unroll
          for (int e = 0; e < (m < 3 ? 4 : 3); ++e) sc_bstore(acc[m][e], gvdst, m < 3 ? voA : voB, (unsigned)(16 * m + e) * plane_b);
      }
    }

    if (WANT_H) {
      constexpr int MTHM = MTH - 1;
      static_assert(K == 51 && 16 * MTHM == 64, "band geometry of the VALU tail");
      float bv[KTV];
#pragma unroll
      for (int t = 0; t < KTV; ++t) bv[t] = vb[(4 * t + ks) * 16];
      f32x4 acc[MTHM];
#pragma unroll
      for (int m = 0; m < MTHM; ++m) acc[m] = (f32x4){0.f, 0.f, 0.f, 0.f};
      // step (c, t) reads input row y + 4 t + ks (tap 4t + ks; the zero tap 51 steps back one row): slot (s0 + 4 t) & 63.
      // The 13 rows span 48 < 64 slots, so they wrap at most once: two bases, picked per step by a compare.
      const int s0 = (y + ks) & (PWIN - 1);
      const float* pA = inT + s0 * MLW + 16 * wc + 4 * j;
      const float* pB = pA - PWIN * MLW;
      const int s0l = (y + 4 * (KTV - 1) + ks - (ks == 3 ? 1 : 0)) & (PWIN - 1);      // last step's row
      const float* pL = inT + s0l * MLW + 16 * wc + 4 * j;
      auto a_of = [&](int idx) {
        const int c = idx / KTV, t = idx - KTV * c;
        const float* p = (t == KTV - 1) ? pL : ((s0 + 4 * t >= PWIN) ? pB : pA) + 4 * t * MLW;
        return *reinterpret_cast<const f32x4*>(p + c * LP);
      };
      constexpr int NS = C * KTV;
#if SC_PIN
      __builtin_amdgcn_sched_barrier(0);
#endif
      f32x4 a0 = a_of(0), a1 = a_of(1);
#if SC_PIN
      __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
#endif
#pragma unroll
      for (int idx = 0; idx < NS; ++idx) {
        f32x4 a2 = a0;
        if (idx + 2 < NS) a2 = a_of(idx + 2);
        const int c = idx / KTV, t = idx - KTV * c;
        const float bb = g_[c] * bv[t];
#pragma unroll
        for (int m = 0; m < MTHM; ++m) acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[m], bb, acc[m], 0, 0, 0);
        a0 = a1;
        a1 = a2;
#if SC_PIN
        if (idx + 2 < NS) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, MTHM, 0);
#endif
      }
#if SC_PIN
      __builtin_amdgcn_sched_barrier(0);
#endif
      // VALU tail (q = 64, 65): lane = tap row fy, then a 64-lane sum of the three partial products
      float s6414 = 0.f, s6415 = 0.f, s6515 = 0.f;
      {
        const int fy = min(lane, K - 1);
        const float live = lane < K ? 1.f : 0.f;
        const float v14 = vB[fy * 16 + 14] * live, v15 = vB[fy * 16 + 15] * live;
        const float* col = inT + ((y + fy) & (PWIN - 1)) * MLW + 16 * wc + 64;
#pragma unroll
        for (int c = 0; c < C; ++c) {
          const float a64 = col[c * LP], a65 = col[c * LP + 1];
          const float g14 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, g_[c]), 14));
          const float g15 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, g_[c]), 15));
          s6414 = fmaf(g14 * v14, a64, s6414);
          s6415 = fmaf(g15 * v15, a64, s6415);
          s6515 = fmaf(g15 * v15, a65, s6515);
        }
        s6414 = wave_sum(s6414);
        s6415 = wave_sum(s6415);
        s6515 = wave_sum(s6515);
      }
      {   // gH through an LDS transpose of the wave's (dead) tap rows: 13 x four 64-byte runs
        __builtin_amdgcn_wave_barrier();
        float* tile = hB;
#pragma unroll
        for (int m = 0; m < MTHM; ++m)
#pragma unroll
          for (int e = 0; e < 4; ++e) tile[(16 * ks + 4 * e + m) * 16 + j] = acc[m][e];
        __builtin_amdgcn_wave_barrier();
        const unsigned ho = opix_b + (unsigned)ks * plane_b;
#pragma unroll
        for (int t = 0; t < KTV; ++t) {
          const int fx = 4 * t + ks;
          const float val = tile[min(j + fx, 63) * 16 + j];
          sc_bstore(val, ghdst, (pvalid && fx < K && j + fx < 64) ? ho : SC_OOR, (unsigned)(4 * t) * plane_b);
        }
        __builtin_amdgcn_wave_barrier();
      }
      sc_bstore(s6414, ghdst, (pvalid && lane == 14) ? opix_b : SC_OOR, 50u * plane_b);
      sc_bstore(s6415, ghdst, (pvalid && lane == 15) ? opix_b : SC_OOR, 49u * plane_b);
      sc_bstore(s6515, ghdst, (pvalid && lane == 15) ? opix_b : SC_OOR, 50u * plane_b);
    }

    __builtin_amdgcn_wave_barrier();
    mfma_store_taps<K, NREG>(hB, hreg, lane);
    mfma_store_taps<K, NREG>(vB, vreg, lane);
    __builtin_amdgcn_wave_barrier();
    if (slide) {                                   // the twelve oldest rows are behind every wave's next phase
      __syncthreads();
      stage_rows_write<PAHEAD>(slid, inT, loaded_hi, tid);
      loaded_hi += PAHEAD;
      __syncthreads();
    }
    b = nb; x0 = nx0; ph = nph_;
  }
}

// ------------------------------------------------------------------------------------------
// generic direct kernels (any K, any C): one thread per output element, x fastest.
// ------------------------------------------------------------------------------------------
__global__ void sepconv_fwd_direct(const float* __restrict__ in, const float* __restrict__ v,
                                   const float* __restrict__ h, float* __restrict__ out, int C, int Ho,
                                   int Wo, int K) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x;
  const int y = blockIdx.y;
  const int bc = blockIdx.z, b = bc / C;
  if (x >= Wo) return;
  const int Hi = Ho + K - 1, Wi = Wo + K - 1;
  const size_t plane = (size_t)Ho * Wo;
  const size_t pix = (size_t)b * K * plane + (size_t)y * Wo + x;
  const float* src = in + (size_t)bc * Hi * Wi + (size_t)y * Wi + x;
  float acc = 0.f;
  for (int fy = 0; fy < K; ++fy) {
    float t = 0.f;
    for (int fx = 0; fx < K; ++fx) t = fmaf(src[(size_t)fy * Wi + fx], h[pix + fx * plane], t);
    acc = fmaf(v[pix + fy * plane], t, acc);
  }
  out[(size_t)bc * plane + (size_t)y * Wo + x] = acc;
}

// which = 0: gV[b,f,y,x] = sum_c gO * sum_fx in[y+f][x+fx]*h[fx]
// which = 1: gH[b,f,y,x] = sum_c gO * sum_fy in[y+fy][x+f]*v[fy]
__global__ void sepconv_bwd_filter_direct(const float* __restrict__ in, const float* __restrict__ other,
                                          const float* __restrict__ gO, float* __restrict__ gF, int C,
                                          int Ho, int Wo, int K, int which) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x;
  const int y = blockIdx.y;
  const int bf = blockIdx.z, b = bf / K, f = bf - b * K;
  if (x >= Wo) return;
  const int Hi = Ho + K - 1, Wi = Wo + K - 1;
  const size_t plane = (size_t)Ho * Wo;
  const size_t opix = (size_t)y * Wo + x;
  const size_t pix = (size_t)b * K * plane + opix;
  float acc = 0.f;
  for (int c = 0; c < C; ++c) {
    const float* src = in + ((size_t)b * C + c) * Hi * Wi;
    float t = 0.f;
    if (which == 0) {
      for (int fx = 0; fx < K; ++fx) t = fmaf(src[(size_t)(y + f) * Wi + x + fx], other[pix + fx * plane], t);
    } else {
      for (int fy = 0; fy < K; ++fy) t = fmaf(src[(size_t)(y + fy) * Wi + x + f], other[pix + fy * plane], t);
    }
    acc = fmaf(gO[((size_t)b * C + c) * plane + opix], t, acc);
  }
  gF[pix + f * plane] = acc;
}

// gI[b,c,Y,X] = sum over (fy,fx) with 0 <= Y-fy < Ho, 0 <= X-fx < Wo of
//               gO[b,c,Y-fy,X-fx] * v[b,fy,Y-fy,X-fx] * h[b,fx,Y-fy,X-fx]
// (exact adjoint of the forward; the reference kernel tests `> max` instead of `>= max`,
//  sepconv.py:51,54, and reads one row/column past the end).
__global__ void sepconv_bwd_input_direct(const float* __restrict__ v, const float* __restrict__ h,
                                         const float* __restrict__ gO, float* __restrict__ gI, int C,
                                         int Ho, int Wo, int K) {
  const int Hi = Ho + K - 1, Wi = Wo + K - 1;
  const int X = blockIdx.x * blockDim.x + threadIdx.x;
  const int Y = blockIdx.y;
  const int bc = blockIdx.z, b = bc / C;
  if (X >= Wi) return;
  const size_t plane = (size_t)Ho * Wo;
  const float* g = gO + (size_t)bc * plane;
  const float* vb = v + (size_t)b * K * plane;
  const float* hb = h + (size_t)b * K * plane;
  const int fy_lo = max(0, Y - (Ho - 1)), fy_hi = min(K - 1, Y);
  const int fx_lo = max(0, X - (Wo - 1)), fx_hi = min(K - 1, X);
  float acc = 0.f;
  for (int fy = fy_lo; fy <= fy_hi; ++fy) {
    const int y = Y - fy;
    for (int fx = fx_lo; fx <= fx_hi; ++fx) {
      const int x = X - fx;
      const size_t o = (size_t)y * Wo + x;
      acc = fmaf(g[o] * vb[fy * plane + o], hb[fx * plane + o], acc);
    }
  }
  gI[(size_t)bc * Hi * Wi + (size_t)Y * Wi + X] = acc;
}

// Kernel-generation switches of VARIANT BUILDS (compile-time: tools/build_variant.sh NAME sepconv.hip -DSAVFI_SEPCONV_...; the shipped
// library has none set): SAVFI_SEPCONV_F32_MFMA keeps the filter gradients on the fp32 matrix-core kernel (sepconv_bwd_mfma_p) instead of the
// split-bf16 one (csrc/sepconv_x6.hip), SAVFI_SEPCONV_NO_MFMA forces the direct kernels, SAVFI_SEPCONV_NO_WS the one-program-per-wave
// split-bf16 kernel instead of the wave-specialised one (csrc/sepconv_ws.hip; _NO_WS_FWD: forward only), SAVFI_SEPCONV_TILED the tiled
// (non-persistent) MFMA kernels, SAVFI_SEPCONV_MFMA_ROWS = 8 | 12 | 16 pins their rows per workgroup.
struct SepconvEnv {
  bool no_mfma, tiled, f32_mfma, no_ws, no_ws_fwd;
  int rows;
};
#ifdef SAVFI_SEPCONV_NO_MFMA
#define SV_NO_MFMA true
#else
#define SV_NO_MFMA false
#endif
#ifdef SAVFI_SEPCONV_TILED
#define SV_TILED true
#else
#define SV_TILED false
#endif
#ifdef SAVFI_SEPCONV_F32_MFMA
#define SV_F32_MFMA true
#else
#define SV_F32_MFMA false
#endif
#ifdef SAVFI_SEPCONV_NO_WS
#define SV_NO_WS true
#else
#define SV_NO_WS false
#endif
#ifdef SAVFI_SEPCONV_NO_WS_FWD
#define SV_NO_WS_FWD true
#else
#define SV_NO_WS_FWD false
#endif
#ifndef SAVFI_SEPCONV_MFMA_ROWS
#define SAVFI_SEPCONV_MFMA_ROWS 0
#endif
const SepconvEnv& sepconv_env() {
  static const SepconvEnv env{SV_NO_MFMA, SV_TILED, SV_F32_MFMA, SV_NO_WS, SV_NO_WS_FWD, SAVFI_SEPCONV_MFMA_ROWS};
  return env;
}

constexpr size_t mfma_lds_bytes(int rows) {
  return ((size_t)3 * (rows + KFAST - 1) * MLW + (size_t)(MNT / 64) * (KFAST + MKP) * 16) * sizeof(float);
}

// Rows per workgroup: one workgroup per CU, so the launch takes ceil(workgroups / 256) rounds of ~(rows + 2) row
// times each (the 2 stands for staging the 50-row halo and the prologue).  E.g. 256x448, B=2: 16 rows -> 224
// workgroups in one round instead of 308 in two with 12; 384x512: 12 rows -> exactly 256 per image.
int mfma_rows(int B, int Ho, int Wo) {
  const int cand[3] = {12, 8, 16};
  int best = 12;
  long best_cost = -1;
  for (int r : cand) {
    const long wgs = (long)B * savfi_cdiv(Wo, MC) * savfi_cdiv(Ho, r);
    const long cost = ((wgs + 255) / 256) * (r + 2);
    if (best_cost < 0 || cost < best_cost) { best_cost = cost; best = r; }
  }
  return sepconv_env().rows ? sepconv_env().rows : best;
}

template <int R>
int launch_fwd_mfma(const float* in, const float* v, const float* h, float* out, int B, int Ho, int Wo, hipStream_t st) {
  constexpr size_t lds = mfma_lds_bytes(R);
  static_assert(lds <= 160 * 1024, "LDS per CU");
  static uint32_t done = 0;
  if (int e = savfi_ensure_dynamic_lds((const void*)sepconv_fwd_mfma<KFAST, R>, lds, done)) return e;
  dim3 grid(savfi_cdiv(Wo, MC), savfi_cdiv(Ho, R), B);
  hipLaunchKernelGGL((sepconv_fwd_mfma<KFAST, R>), grid, dim3(MNT), lds, st, in, v, h, out, Ho, Wo);
  return savfi_launch_status();
}

template <int R, bool WV, bool WH>
int launch_bwd_mfma_one(const float* in, const float* v, const float* h, const float* gO, float* gV, float* gH, int B,
                        int Ho, int Wo, hipStream_t st) {
  constexpr size_t lds = mfma_lds_bytes(R);
  static uint32_t done = 0;
  if (int e = savfi_ensure_dynamic_lds((const void*)sepconv_bwd_mfma<KFAST, R, WV, WH>, lds, done)) return e;
  dim3 grid(savfi_cdiv(Wo, MC), savfi_cdiv(Ho, R), B);
  hipLaunchKernelGGL((sepconv_bwd_mfma<KFAST, R, WV, WH>), grid, dim3(MNT), lds, st, in, v, h, gO, gV, gH, Ho, Wo);
  return savfi_launch_status();
}

template <int R>
int launch_bwd_mfma(const float* in, const float* v, const float* h, const float* gO, float* gV, float* gH, int B,
                    int Ho, int Wo, hipStream_t st) {
  if (gV && gH) return launch_bwd_mfma_one<R, true, true>(in, v, h, gO, gV, gH, B, Ho, Wo, st);
  if (gV) return launch_bwd_mfma_one<R, true, false>(in, v, h, gO, gV, gH, B, Ho, Wo, st);
  return launch_bwd_mfma_one<R, false, true>(in, v, h, gO, gV, gH, B, Ho, Wo, st);
}

constexpr size_t persistent_lds_bytes() {
  return ((size_t)3 * PWIN * MLW + (size_t)(MNT / 64) * (KFAST + MKP) * 16) * sizeof(float);
}

int device_cu_count() {
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

// whole-tensor buffer resources: every tensor must stay below 2^31 bytes
// the MFMA kernels address one sample's tap planes / output planes through raw buffer descriptors: offsets, and the
// out-of-range marker SC_OOR, need the sample to stay below 2^31 bytes (K = 51: frames up to 10.5 Mpixel); larger ones take
// the generic kernels
bool mfma_fits(int Ho, int Wo) {
  return (int64_t)KFAST * Ho * Wo * 4 < ((int64_t)1 << 31);
}

bool persistent_ok(int B, int Ho, int Wo) {
  const int64_t taps = (int64_t)B * KFAST * Ho * Wo * 4, win = (int64_t)B * 3 * (Ho + KFAST - 1) * (Wo + KFAST - 1) * 4;
  return taps < ((int64_t)1 << 31) && win < ((int64_t)1 << 31);
}

template <bool WV, bool WH>
int launch_bwd_persistent_one(const float* in, const float* v, const float* h, const float* gO, float* gV, float* gH, int B,
                              int Ho, int Wo, hipStream_t st) {
  constexpr size_t lds = persistent_lds_bytes();
  static_assert(lds <= 160 * 1024, "LDS per CU");
  static uint32_t done = 0;
  if (int e = savfi_ensure_dynamic_lds((const void*)sepconv_bwd_mfma_p<KFAST, WV, WH>, lds, done)) return e;
  const int nph = savfi_cdiv(Ho, 2), ncol = savfi_cdiv(Wo, MC);
  const int64_t total = (int64_t)B * ncol * nph;
  const int per_wg = savfi_cdiv(total, device_cu_count());
  const int grid = savfi_cdiv(total, per_wg);
  hipLaunchKernelGGL((sepconv_bwd_mfma_p<KFAST, WV, WH>), dim3(grid), dim3(MNT), lds, st, in, v, h, gO, gV, gH, B, Ho, Wo, nph, ncol,
                     per_wg);
  return savfi_launch_status();
}

int launch_bwd_persistent(const float* in, const float* v, const float* h, const float* gO, float* gV, float* gH, int B, int Ho,
                          int Wo, hipStream_t st) {
  if (gV && gH) return launch_bwd_persistent_one<true, true>(in, v, h, gO, gV, gH, B, Ho, Wo, st);
  if (gV) return launch_bwd_persistent_one<true, false>(in, v, h, gO, gV, gH, B, Ho, Wo, st);
  return launch_bwd_persistent_one<false, true>(in, v, h, gO, gV, gH, B, Ho, Wo, st);
}

int check_dims(int B, int C, int Ho, int Wo, int K) {
  if (B <= 0 || C <= 0 || Ho <= 0 || Wo <= 0 || K <= 0) return SAVFI_E_SHAPE;
  const int64_t Hi = (int64_t)Ho + K - 1, Wi = (int64_t)Wo + K - 1;
  if ((int64_t)B * K * Ho * Wo >= (int64_t)1 << 40 || (int64_t)B * C * Hi * Wi >= (int64_t)1 << 40)
    return SAVFI_E_TOOBIG;
  if ((int64_t)B * K > 65535 || (int64_t)B * C > 65535 || Ho + K - 1 > 65535) return SAVFI_E_TOOBIG;
  return SAVFI_OK;
}

}  // namespace

extern "C" int savfi_sepconv_fwd_f32(const float* in, const float* v, const float* h, float* out, int B,
                                     int C, int Ho, int Wo, int K, void* stream) {
  if (!in || !v || !h || !out) return SAVFI_E_NULL;
  if (int e = check_dims(B, C, Ho, Wo, K)) return e;
  hipStream_t st = (hipStream_t)stream;
  if (K == KFAST && C == 3 && !sepconv_env().no_mfma && !sepconv_env().tiled && !sepconv_env().f32_mfma && persistent_ok(B, Ho, Wo)) {
    // widths that are a multiple of 4: the wave-specialised kernel (csrc/sepconv_ws.hip); others: one program per wave (csrc/sepconv_x6.hip)
    if ((Wo & 3) == 0 && !sepconv_env().no_ws && !sepconv_env().no_ws_fwd)
      return savfi_sepconv_fwd_ws_launch(in, v, h, out, B, Ho, Wo, device_cu_count(), KFAST, nullptr, 0, st);
    return savfi_sepconv_fwd_x6_launch(in, v, h, out, B, Ho, Wo, device_cu_count(), st);
  }
  if (K == KFAST && C == 3 && !sepconv_env().no_mfma && mfma_fits(Ho, Wo)) {
    switch (mfma_rows(B, Ho, Wo)) {
      case 8: return launch_fwd_mfma<8>(in, v, h, out, B, Ho, Wo, st);
      case 16: return launch_fwd_mfma<16>(in, v, h, out, B, Ho, Wo, st);
      default: return launch_fwd_mfma<12>(in, v, h, out, B, Ho, Wo, st);
    }
  } else {
    dim3 grid(savfi_cdiv(Wo, 64), Ho, B * C);
    hipLaunchKernelGGL(sepconv_fwd_direct, grid, dim3(64), 0, st, in, v, h, out, C, Ho, Wo, K);
  }
  return savfi_launch_status();
}

// The same op on tap tensors that are SLICES of one interleaved buffer: sample b of v / h (gV / gH) starts tap_bstride planes (of Ho * Wo
// floats) after sample b - 1 (51 = contiguous).  sepconv/model.py runs its four sub-networks as ONE task-batched launch per layer; their
// outputs form a [B * 4, 51, Ho, Wo] tensor whose sample 4 b + s belongs to sub-network s, and the two local convolutions read (and their
// gradients write) that buffer in place with tap_bstride = 4 * 51.  K = 51, C = 3, Wo % 4 == 0 (the wave-specialised kernels);
// anything else: SAVFI_E_UNSUPPORTED (the caller copies the slices and uses the contiguous entry points).
static int taps_strided_ok(int B, int C, int Ho, int Wo, int K, int tap_bstride) {
  if (int e = check_dims(B, C, Ho, Wo, K)) return e;
  if (tap_bstride < K) return SAVFI_E_SHAPE;
  if (K != KFAST || C != 3 || (Wo & 3) != 0 || sepconv_env().no_mfma || sepconv_env().tiled || sepconv_env().f32_mfma || sepconv_env().no_ws)
    return SAVFI_E_UNSUPPORTED;
  const int64_t taps = ((int64_t)(B - 1) * tap_bstride + K) * Ho * Wo * 4;
  if (taps >= ((int64_t)1 << 31) || !persistent_ok(B, Ho, Wo)) return SAVFI_E_UNSUPPORTED;
  return SAVFI_OK;
}

// 1: the strided / frames8 / pair entry points take this problem NOW (shapes, sizes and the A/B switches of this process:
// SAVFI_SEPCONV_NO_WS, _NO_WS_FWD, _NO_MFMA, _TILED, _F32_MFMA make them refuse), 0: the caller uses the contiguous entry points
extern "C" int savfi_sepconv_taps_strided_supported(int B, int C, int Ho, int Wo, int K, int tap_bstride) {
  return taps_strided_ok(B, C, Ho, Wo, K, tap_bstride) == SAVFI_OK && !sepconv_env().no_ws_fwd ? 1 : 0;
}

extern "C" int savfi_sepconv_fwd_taps_strided_f32(const float* in, const float* v, const float* h, float* out, int B, int C, int Ho, int Wo,
                                                  int K, int tap_bstride, void* stream) {
  if (!in || !v || !h || !out) return SAVFI_E_NULL;
  if (int e = taps_strided_ok(B, C, Ho, Wo, K, tap_bstride)) return e;
  return savfi_sepconv_fwd_ws_launch(in, v, h, out, B, Ho, Wo, device_cu_count(), tap_bstride, nullptr, 0, (hipStream_t)stream);
}

extern "C" int savfi_sepconv_bwd_taps_strided_f32(const float* in, const float* v, const float* h, const float* gO, float* gV, float* gH,
                                                  int B, int C, int Ho, int Wo, int K, int tap_bstride, void* stream) {
  if (!in || !v || !h || !gO || !gV || !gH) return SAVFI_E_NULL;
  if (int e = taps_strided_ok(B, C, Ho, Wo, K, tap_bstride)) return e;
  return savfi_sepconv_bwd_ws_launch(in, v, h, gO, gV, gH, B, Ho, Wo, device_cu_count(), tap_bstride, nullptr, 0, (hipStream_t)stream);
}

// the strided entry points with the words of savfi_frames8_classify_f32(in) (csrc/sepconv_ws.hip): the device picks the three-product
// kernel for frames of 8-bit images and the six-product kernel for anything else
extern "C" int savfi_sepconv_fwd_frames8_f32(const float* in, const float* v, const float* h, float* out, const unsigned* cls, int B, int C,
                                             int Ho, int Wo, int K, int tap_bstride, int taps_unit16, void* stream) {
  if (!in || !v || !h || !out || !cls) return SAVFI_E_NULL;
  if (int e = taps_strided_ok(B, C, Ho, Wo, K, tap_bstride)) return e;
  if (sepconv_env().no_ws_fwd) return SAVFI_E_UNSUPPORTED;
  return savfi_sepconv_fwd_ws_launch(in, v, h, out, B, Ho, Wo, device_cu_count(), tap_bstride, cls, taps_unit16 & 1, (hipStream_t)stream);
}

extern "C" int savfi_sepconv_bwd_frames8_f32(const float* in, const float* v, const float* h, const float* gO, float* gV, float* gH,
                                             const unsigned* cls, int B, int C, int Ho, int Wo, int K, int tap_bstride, int taps_unit16,
                                             void* stream) {
  if (!in || !v || !h || !gO || !gV || !gH || !cls) return SAVFI_E_NULL;
  if (int e = taps_strided_ok(B, C, Ho, Wo, K, tap_bstride)) return e;
  return savfi_sepconv_bwd_ws_launch(in, v, h, gO, gV, gH, B, Ho, Wo, device_cu_count(), tap_bstride, cls, taps_unit16 & 3,
                                     (hipStream_t)stream);
}

// The forward of both local convolutions in ONE launch: out [B][2][C][Ho][Wo], out[b][f] = the local convolution of frame f of sample b with
// sub-networks 2 f / 2 f + 1 of taps [4 B][K][Ho][Wo]; the caller adds out[:, 0] and out[:, 1] (sepconv/model.py:346-347).  Bit for bit the
// results of the two calls of savfi_sepconv_fwd_frames8_f32 it replaces.
extern "C" int savfi_sepconv_fwd_pair_frames8_f32(const float* in0, const float* in1, const float* taps, float* out, const unsigned* cls0,
                                                  const unsigned* cls1, int B, int C, int Ho, int Wo, int K, int taps_unit16, void* stream) {
  if (!in0 || !in1 || !taps || !out || !cls0 || !cls1) return SAVFI_E_NULL;
  if (int e = taps_strided_ok(B, C, Ho, Wo, K, 4 * K)) return e;
  if (sepconv_env().no_ws_fwd || B > 0x3fffffff / 2 || !persistent_ok(2 * B, Ho, Wo)) return SAVFI_E_UNSUPPORTED;
  const size_t plane = (size_t)K * Ho * Wo;
  return savfi_sepconv_fwd_ws_launch(in0, taps, taps + plane, out, 2 * B, Ho, Wo, device_cu_count(), 2 * K, cls0, taps_unit16 & 1,
                                     (hipStream_t)stream, in1, cls1);
}

// Both local convolutions of an interleaved tap tensor in ONE launch: taps / gtaps [4 B][K][Ho][Wo] with sample 4 b + s = sub-network s
// (0: v of frame 0, 1: h of frame 0, 2: v of frame 1, 3: h of frame 1), in0 / in1 the two frames, gO the cotangent of their sum.  The kernel
// sees 2 B virtual samples at a tap stride of 2 K planes (csrc/sepconv_ws.hip `pair`).  Same results, bit for bit, as the two calls of
// savfi_sepconv_bwd_frames8_f32 it replaces.
extern "C" int savfi_sepconv_bwd_pair_frames8_f32(const float* in0, const float* in1, const float* taps, const float* gO, float* gtaps,
                                                  const unsigned* cls0, const unsigned* cls1, int B, int C, int Ho, int Wo, int K,
                                                  int taps_unit16, void* stream) {
  if (!in0 || !in1 || !taps || !gO || !gtaps || !cls0 || !cls1) return SAVFI_E_NULL;
  if (int e = taps_strided_ok(B, C, Ho, Wo, K, 4 * K)) return e;
  if (B > 0x3fffffff / 2 || !persistent_ok(2 * B, Ho, Wo)) return SAVFI_E_UNSUPPORTED;
  const size_t plane = (size_t)K * Ho * Wo;
  return savfi_sepconv_bwd_ws_launch(in0, taps, taps + plane, gO, gtaps, gtaps + plane, 2 * B, Ho, Wo, device_cu_count(), 2 * K, cls0,
                                     taps_unit16 & 3, (hipStream_t)stream, in1, cls1);
}

extern "C" int savfi_sepconv_bwd_f32(const float* in, const float* v, const float* h, const float* gO,
                                     float* gI, float* gV, float* gH, int B, int C, int Ho, int Wo, int K,
                                     void* stream) {
  if (!in || !v || !h || !gO) return SAVFI_E_NULL;
  if (int e = check_dims(B, C, Ho, Wo, K)) return e;
  hipStream_t st = (hipStream_t)stream;
  if (gV || gH) {
    if (K == KFAST && C == 3 && gV && gH && !sepconv_env().no_mfma && !sepconv_env().tiled && !sepconv_env().f32_mfma &&
        persistent_ok(B, Ho, Wo)) {
      // widths that are a multiple of 4: the wave-specialised kernel (csrc/sepconv_ws.hip); others: one program per wave (csrc/sepconv_x6.hip)
      if ((Wo & 3) == 0 && !sepconv_env().no_ws) {
        if (int e = savfi_sepconv_bwd_ws_launch(in, v, h, gO, gV, gH, B, Ho, Wo, device_cu_count(), KFAST, nullptr, 0, st)) return e;
      } else if (int e = savfi_sepconv_bwd_x6_launch(in, v, h, gO, gV, gH, B, Ho, Wo, device_cu_count(), st)) return e;
    } else if (K == KFAST && C == 3 && !sepconv_env().no_mfma && !sepconv_env().tiled && persistent_ok(B, Ho, Wo)) {
      if (int e = launch_bwd_persistent(in, v, h, gO, gV, gH, B, Ho, Wo, st)) return e;
    } else if (K == KFAST && C == 3 && !sepconv_env().no_mfma && mfma_fits(Ho, Wo)) {
      int e;
      switch (mfma_rows(B, Ho, Wo)) {
        case 8: e = launch_bwd_mfma<8>(in, v, h, gO, gV, gH, B, Ho, Wo, st); break;
        case 16: e = launch_bwd_mfma<16>(in, v, h, gO, gV, gH, B, Ho, Wo, st); break;
        default: e = launch_bwd_mfma<12>(in, v, h, gO, gV, gH, B, Ho, Wo, st); break;
      }
      if (e) return e;
    } else {
      dim3 grid(savfi_cdiv(Wo, 64), Ho, B * K);
      if (gV) hipLaunchKernelGGL(sepconv_bwd_filter_direct, grid, dim3(64), 0, st, in, h, gO, gV, C, Ho, Wo, K, 0);
      if (gH) hipLaunchKernelGGL(sepconv_bwd_filter_direct, grid, dim3(64), 0, st, in, v, gO, gH, C, Ho, Wo, K, 1);
      if (int e = savfi_launch_status()) return e;
    }
  }
  if (gI) {
    dim3 grid(savfi_cdiv(Wo + K - 1, 64), Ho + K - 1, B * C);
    hipLaunchKernelGGL(sepconv_bwd_input_direct, grid, dim3(64), 0, st, v, h, gO, gI, C, Ho, Wo, K);
    if (int e = savfi_launch_status()) return e;
  }
  return SAVFI_OK;
}
