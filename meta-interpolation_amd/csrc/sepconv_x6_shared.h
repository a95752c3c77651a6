// Shared pieces of the split-bf16 SepConv kernels (csrc/sepconv_x6.hip: one program per wave; csrc/sepconv_ws.hip: MFMA waves and
// staging waves in pairs): geometry of the LDS window, buffer-descriptor accessors, the exact 3-way bf16 split, window row staging.
#pragma once
#include "common.h"
#include <stdlib.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef short bf16x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// cache-policy bits of the kernels' buffer loads / 16-byte stores (aux operand: 1 = sc0, 2 = nt, 16 = sc1); experiment switches
#ifndef X6_LOAD_AUX
#define X6_LOAD_AUX 0
#endif
// stores: sc1 (16).  Round 5, pair launches in the bench loop, two repeats on one box (profiles/r05_dma_policy_ab_box2.txt): default 291.9 /
// 293.7 us, sc1 289.0 / 288.0, nt 292.0 / 294.7 with the half-sized launch 168 -> 191.  (Round 4 on the dword-load kernels: all neutral.)
#ifndef X6_STORE_AUX
#define X6_STORE_AUX 16
#endif

namespace {

constexpr int XK = 51, XC = 3;
constexpr int XNT = 512;                         // 8 waves: 4 output rows x 2 groups of 16 pixels
constexpr int XMC = 32, XPR = 4;                 // strip width, rows per phase
constexpr int XWIN = 64, XAHEAD = 8;             // circular window rows, rows per slide
constexpr int XBLK = 64 * 16 + 128, XNBLK = 10;  // bytes per 8-column block, blocks (80 columns)
constexpr int XPLANE = XNBLK * XBLK, XWINB = 9 * XPLANE;
constexpr int XTABP = 8 * 256, XTAB = 3 * XTABP; // tap table of a wave: [piece][k / 8][16 j][8]
constexpr int XSIDE_OFF = XWINB + 8 * XTAB, XSIDE = XC * XWIN * 4 * 4;
constexpr int XTAIL_OFF = XSIDE_OFF + XSIDE, XTAILB = 512;      // gV tail sums of a wave: [column][64 tap rows] floats
constexpr int XLDS = XTAIL_OFF + 8 * XTAILB;
static_assert(XLDS <= 160 * 1024, "LDS per CU");
constexpr int XNREG = (XK + 3) / 4;              // gH leaves in 13 instructions of four taps
constexpr int XNP = 7;                           // tap pairs per lane
constexpr unsigned X_OOR = 0x80000000u;

__device__ __forceinline__ float x6_bload(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
#ifdef X6_EXP_LOADHIT      // experiment: every load hits the same few cache lines
  voff &= 0xfffu; soff = 0u;
#endif
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)voff, (int)soff, X6_LOAD_AUX));
}
__device__ __forceinline__ void x6_bstore(float val, __amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
#ifdef X6_EXP_NOSTORE      // experiment: every store is dropped by the range check
  voff = X_OOR;
#endif
  __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, val), r, (int)voff, (int)soff, 0);
}
__device__ __forceinline__ void x6_bstore4(f32x4 val, __amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
#ifdef X6_EXP_NOSTORE
  voff = 0x80000000u;
#endif
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, val), r, (int)voff, (int)soff, X6_STORE_AUX);
  // A 128-bit buffer store reads its data registers over several cycles; a VALU write of one of them in the very next issue slot
  // reaches the register first (gfx950, measured: element 0 of the rows 32 + fq of gV wrong in the last lanes of every 16, run-to-run
  // different -- the store was followed at once by the v_add that reuses its first data register).  hipcc (ROCm 7.2) inserts the wait
  // state only for the form WITHOUT an SGPR soffset (GCNHazardRecognizer: "the hazard only exists if soffset is not a register"); ours
  // carries one.  The data stays live through two wait states after the store.
  asm volatile("s_nop 1" ::"v"(val));
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t x6_rsrc(const float* base, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ unsigned x6_cvt_pk(float lo, float hi) {
  unsigned r;
  asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
  return r;
}
// a = a1 + a2 + a3, b = b1 + b2 + b3 exactly (bf16 pieces, round to nearest even); h_p = piece p of a (low half), of b (high half)
__device__ __forceinline__ void x6_split2(float a, float b, unsigned& h1, unsigned& h2, unsigned& h3) {
  h1 = x6_cvt_pk(a, b);
  const float ra = a - __uint_as_float(h1 << 16), rb = b - __uint_as_float(h1 & 0xffff0000u);
  h2 = x6_cvt_pk(ra, rb);
  const float qa = ra - __uint_as_float(h2 << 16), qb = rb - __uint_as_float(h2 & 0xffff0000u);
  h3 = x6_cvt_pk(qa, qb);
}
__device__ __forceinline__ void x6_st16(char* p, unsigned v) { *reinterpret_cast<unsigned short*>(p) = (unsigned short)v; }
// The tap table of a wave changes type as a phase goes on (bf16 pieces -> fp32 transpose tile -> bf16 pieces): LDS operations
// of one wave execute in program order, this keeps the compiler from reordering them by type-based alias analysis.
#define X6_ORDER()                      \
  do {                                  \
    __builtin_amdgcn_wave_barrier();    \
    asm volatile("" ::: "memory");      \
  } while (0)

// pitch of the fp32 transpose tile gV leaves through (floats): 4 * 20 = 16 (mod 64) banks between the four row groups of a store
constexpr int XTP = 20;
#ifndef X6_SCALAR_STORES
#define X6_SCALAR_STORES 0
#endif
// experiment switch: wave priority inside the MFMA loops (-DX6_PRIO_LOOP=n)
#ifndef X6_PRIO_LOOP
#define X6_PRIO_LOOP 0
#endif
#define X6_PRIO(n) do { if (X6_PRIO_LOOP) __builtin_amdgcn_s_setprio(n); } while (0)

// rows [r_lo, r_lo + NROWS) of the strip (b, x0): HBM -> registers -> (split) -> their circular slots; thread = (column, row group)
template <int NROWS>
struct X6Rows {
  static constexpr int NIT = NROWS / 4;
  static_assert(NROWS % 4 == 0 && (XC * NIT) % 2 == 0, "whole row groups, element pairs");
  float buf[XC][NIT];
};
template <int NROWS>
__device__ __forceinline__ void x6_rows_load(X6Rows<NROWS>& sr, __amdgpu_buffer_rsrc_t in_rs, int b, int x0, int r_lo, int Hi, int Wi, int tid) {
  const int q = tid & 127, rg = tid >> 7;
  const int colb = min(x0 + q, Wi - 1) * 4;
#pragma unroll
  for (int c = 0; c < XC; ++c)
#pragma unroll
    for (int it = 0; it < X6Rows<NROWS>::NIT; ++it) {
      const int r = min(r_lo + rg + 4 * it, Hi - 1);
      sr.buf[c][it] = x6_bload(in_rs, (unsigned)(((b * XC + c) * Hi + r) * Wi * 4 + colb), 0u);
    }
}
// U8 (csrc/sepconv_ws.hip, frames of 8-bit images): the window holds k = 255 w, an integer 0..255 and so ONE exact bf16 piece (plane c);
// the side columns keep the fp32 values.
template <int NROWS, bool U8 = false>
__device__ __forceinline__ void x6_rows_write(const X6Rows<NROWS>& sr, char* __restrict__ smem, int r_lo, int tid, int side_off = XSIDE_OFF) {
  constexpr int NIT = X6Rows<NROWS>::NIT, NE = XC * NIT;
  const int q = tid & 127, rg = tid >> 7;
  const int cell = (q >> 3) * XBLK + (q & 7) * 2;
  const int sidx = q == 64 ? 0 : q == 65 ? 1 : q == 80 ? 2 : q == 81 ? 3 : -1;
  float* side = reinterpret_cast<float*>(smem + side_off);
#pragma unroll
  for (int e = 0; e < NE; e += 2) {
    const int c0 = e / NIT, i0 = e % NIT, c1 = (e + 1) / NIT, i1 = (e + 1) % NIT;
    const float a = sr.buf[c0][i0], bb = sr.buf[c1][i1];
    const int s0 = (r_lo + rg + 4 * i0) & (XWIN - 1), s1 = (r_lo + rg + 4 * i1) & (XWIN - 1);
    if (q < 8 * XNBLK) {
      char* d0 = smem + c0 * XPLANE + cell + s0 * 16;
      char* d1 = smem + c1 * XPLANE + cell + s1 * 16;
      if constexpr (U8) {
        const unsigned k1 = x6_cvt_pk(rintf(a * 255.f), rintf(bb * 255.f));
        x6_st16(d0, k1); x6_st16(d1, k1 >> 16);
      } else {
        unsigned h1, h2, h3;
        x6_split2(a, bb, h1, h2, h3);
        x6_st16(d0, h1); x6_st16(d0 + 3 * XPLANE, h2); x6_st16(d0 + 6 * XPLANE, h3);
        x6_st16(d1, h1 >> 16); x6_st16(d1 + 3 * XPLANE, h2 >> 16); x6_st16(d1 + 6 * XPLANE, h3 >> 16);
      }
    }
    if (sidx >= 0) {
      side[(c0 * XWIN + s0) * 4 + sidx] = a;
      side[(c1 * XWIN + s1) * 4 + sidx] = bb;
    }
  }
}

}  // namespace
