// SepConv filter gradients (gV, gH; K = 51, C = 3) on split-bf16 MFMAs, WAVE-SPECIALISED: the arithmetic, the LDS window and the
// table / fragment layouts of csrc/sepconv_x6.hip, but the eight waves of a workgroup no longer run the same program.
//
//   gV[b,fy,y,x] = sum_c gO[b,c,y,x] * sum_fx in[b,c,y+fy,x+fx] * h[b,fx,y,x]
//   gH[b,fx,y,x] = sum_c gO[b,c,y,x] * sum_fy in[b,c,y+fy,x+fx] * v[b,fy,y,x]
//
// Replaces the reference's two cupy/NVRTC filter-gradient kernels (sepconv/sepconv_op/sepconv.py:32-63, :138-190 backward).
//
// Why.  sepconv_bwd_x6 keeps every wave on one serial program per 16 pixels: taps HBM -> registers -> split -> table -> B fragments
// -> 144 MFMAs -> scale -> tile -> stores, twice, plus the window slide between two workgroup barriers.  Its section trace
// (profiles/r03_sepconv_x6_section_trace.txt) shows three quarters of a wave's cycles OUTSIDE the MFMA loops and the matrix pipe 40 %
// busy: with two waves per SIMD (LDS allows one workgroup per CU) nothing hides a wave's memory latency, table building and
// barrier skew but the one other wave, which is in the same state half of the time.
//
// Here waves w = 0..3 are MFMA waves and waves 4..7 staging waves; pair p = w & 3 (both on one SIMD: a workgroup's waves go to SIMDs
// cyclically) owns column group p & 1 and output rows (p >> 1) and (p >> 1) + 2 of every phase (4 rows x 32 columns), i.e. two
// "units" of 16 pixels per phase.  Per unit n of a pair:
//   staging wave   h taps (prefetched one unit ahead) -> split -> band table -> [flag]; VALU tails of gV; v taps -> split -> table ->
//                  [flag]; VALU tails of gH; drains the pair's output tile (gV, then gH) into 16-byte stores; loads and splits two new
//                  window rows per unit (the window slides by 2 rows per unit instead of 8 rows behind two barriers every other phase);
//   MFMA wave      [flag] 6 B fragments -> 144 MFMAs on the window -> cotangent scaling -> output tile -> [flag], for gV then gH.
// The only barriers are at the start of a run (a workgroup's stretch of phases inside one 32-column strip: window prologue).  All
// other ordering is by sequence numbers in LDS: DS operations of a wave execute in program order and the LDS serves them in arrival
// order, so "data, then flag" by the writer and "flag, then data" by the reader is enough -- no s_waitcnt on VMEM, no barrier.
//   tab_full / tab_free [pair]    the pair's one tap table (h band of unit n, then v of unit n, then h of unit n + 1 ...): the staging
//                                 wave refills it as soon as the MFMA wave holds the previous content's fragments in registers
//   out_full / out_free [pair]    the pair's output tile
//   prog [wave]                   phases whose window reads the wave has finished;  slide_cnt: window row pairs written (x 4 waves)
// Every wait is bounded (a wedged protocol ends with wrong numbers and savfi_sepconv_ws_errors() > 0, never with a hung GPU).
//
// LDS: window 103 680 B (as sepconv_x6) + 4 x (table 6 144 + output tile 6 400 + tail sums 512) + side columns 3 072 + flags = 159 KB.
#include "sepconv_x6_shared.h"
#include <type_traits>

typedef int ws_i32x4 __attribute__((ext_vector_type(4)));
// 16-byte raw buffer load (the clang builtin of this release narrows the b128 form to one dword: csrc/winograd.hip)
__device__ f32x4 ws_raw_load_x4(ws_i32x4 rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.load.v4f32");

namespace {

constexpr int WNT = 768;                            // 12 waves: per SIMD one MFMA wave and the two staging waves of its pair
constexpr int WTILEB = 6400;                       // output tile of a pair: gV 64 tap rows x pitch 20 floats; gH 79 rows (fx + 15) x pitch 20
constexpr int WTAILB = 512;                        // gV tail sums of a pair: [column 14 | 15][64 tap rows] floats
constexpr int WPAIRB = XTAB + WTILEB + WTAILB;
constexpr int WPAIR_OFF = XWINB;
constexpr int WSIDE_OFF = WPAIR_OFF + 4 * WPAIRB;
constexpr int WSIDEB = XC * XWIN * 4 * 4;
constexpr int WFLAG_OFF = WSIDE_OFF + WSIDEB;
constexpr int WLDS = WFLAG_OFF + 256;
static_assert(WLDS <= 160 * 1024, "LDS per CU");
constexpr int WRINGB = 4096;                       // DMA: a staging wave's slot for the raw taps of one unit (51 x 64 bytes, four 1 KB pieces)
constexpr int WRING_GP = 3328;                     // the unit's cotangent [channel][16 pixels] behind the taps
constexpr int WS_LGKMCNT0 = 0xC07F;                // s_waitcnt lgkmcnt(0), the other counters unconstrained
constexpr int WRING_ROW = 3584;                    // the wave's 64 columns x two channels of the window row of the unit's iteration
constexpr int WS_DMA_UNIT = 7;                     // LDS-DMA instructions per unit: four pieces of the taps, the cotangent, two of the window row
constexpr int WS_DMA_ITER = 4 + WS_DMA_UNIT;       // vector memory operations of a staging wave's iteration: four drain stores, one unit's fetch
typedef __attribute__((address_space(3))) void ws_lds_void;
#ifndef WS_TILE_PITCH
#define WS_TILE_PITCH 20
#endif
constexpr int WPV = WS_TILE_PITCH;                            // tile pitch (floats): rows leave as 16-byte pieces of 4 pixels.  gH: the MFMA wave writes
                                                   // D[R][j] (R = window column) to row R - j + 15 = fx + 15, i.e. resolves gH[fx][j] = D[j + fx][j]
                                                   // by its store ADDRESS (conflict free: 16 kg - 19 j mod 32 is a bijection of a half wave)
enum { F_TAB_FULL = 0, F_TAB_FREE = 4, F_OUT_FULL = 8, F_OUT_FREE = 12, F_PROG = 16, F_SLIDE = 28, F_ERR = 29,
       F_LIMIT = 30,
       F_ERRW = 62 };                              // two words: the address of the mapped host word (savfi_sepconv_ws_watch) or 0                             // spins of a bounded wait (x s_sleep 2) before it gives up: a kernel argument, kept in LDS
// experiment switches (timing only, results wrong): -DWS_EXP_NOMFMA the MFMA waves skip both MFMA loops (what the staging waves alone
// sustain), -DWS_EXP_NOSTAGE the staging waves only run the protocol (what the MFMA waves alone sustain)
#ifndef WS_EXP_NOMFMA
#define WS_EXP_NOMFMA 0
#endif
#ifndef WS_EXP_NOSTAGE
#define WS_EXP_NOSTAGE 0
#endif
#ifndef WS_PRIO                 // wave priority of the MFMA waves (the staging waves stay at 0)
#define WS_PRIO 2
#endif
#ifndef WS_INTERLEAVE           // 1: the next block's A-fragment reads are spread between this block's MFMAs (one LDS read per matrix-pipe gap)
#define WS_INTERLEAVE 1
#endif
#ifndef WS_PRIO_TABLE           // wave priority of a staging wave while it builds a table (what the MFMA wave waits for next)
#define WS_PRIO_TABLE 3
#endif
#ifndef WS_GH_READS_PER_GAP       // gH: transpose reads of the next block per MFMA gap (12 per block: 2 = in the first six gaps, 1 = one per gap)
#define WS_GH_READS_PER_GAP 2
#endif
#ifndef WS_EXP_NOAREAD          // the MFMA loops reuse the first unit's A fragments (no LDS reads inside the loops)
#define WS_EXP_NOAREAD 0
#endif
// (nt: the taps are read once.  With both local convolutions in one launch -- 1.5 GB per launch -- the non-temporal fetch is worth 4 %:
// 306 -> 294.5 us in the bench loop, twice on one box, profiles/r05_dma_policy_ab.txt; on the half-sized launches it was 1 %)
#ifndef WS_DMA_TAP_POLICY_ID    // cache policy of the tap fetches (DMA kernels): 0 default, 1 nt (non-temporal), 2 sc1, 3 sc1 nt, 4 sc0 sc1
#define WS_DMA_TAP_POLICY_ID 1
#endif
#if WS_DMA_TAP_POLICY_ID == 1
#define WS_DMA_TAP_POLICY " nt"
#elif WS_DMA_TAP_POLICY_ID == 2
#define WS_DMA_TAP_POLICY " sc1"
#elif WS_DMA_TAP_POLICY_ID == 3
#define WS_DMA_TAP_POLICY " sc1 nt"
#elif WS_DMA_TAP_POLICY_ID == 4
#define WS_DMA_TAP_POLICY " sc0 sc1"
#else
#define WS_DMA_TAP_POLICY ""
#endif
#ifndef WS_EXP_UNITMAJOR        // see load_taps of sepconv_bwd_ws
#define WS_EXP_UNITMAJOR 0
#endif
#ifndef WS_EXP_NOMFMAONLY       // the A fragments are read, the MFMAs are not issued
#define WS_EXP_NOMFMAONLY 0
#endif

__device__ unsigned ws_error_count = 0;
typedef __attribute__((address_space(1))) unsigned ws_global_u32;
// -DWS_TRACE: wave cycles (s_memtime) per section, summed over the units of workgroup 0, in ws_trace[wave][section]
#ifndef WS_TRACE
#define WS_TRACE 0
#endif
#if WS_TRACE
__device__ unsigned long long ws_trace[16][16];
__device__ unsigned long long ws_trace_wg[1024][2];           // per workgroup: cycles of wave 0 from entry to exit, runs
#define WS_TRACE_ADD(wave_, slot_, val_) (ws_trace[wave_][slot_] += (val_))
#define WS_TRACE_WG(slot_, val_) do { if (blockIdx.x < 1024 && threadIdx.x == 0) ws_trace_wg[blockIdx.x][slot_] += (val_); } while (0)
#else
#define WS_TRACE_ADD(wave_, slot_, val_) ((void)0)
#define WS_TRACE_WG(slot_, val_) ((void)0)
#endif
#define WS_T(k) do { if (WS_TRACE) { const unsigned long long t_ = __builtin_readcyclecounter(); tr_[k] += t_ - tlast_; tlast_ = t_; } } while (0)

typedef __attribute__((address_space(3))) unsigned lds_u32;

__device__ __forceinline__ ws_i32x4 ws_rsrc4(const float* base, unsigned bytes) {
  const unsigned long long q = reinterpret_cast<unsigned long long>(base);
  return (ws_i32x4){(int)(unsigned)q, (int)(unsigned)(q >> 32), (int)bytes, 0x00020000};
}
__device__ __forceinline__ unsigned ws_peek(const unsigned* f) {
  const unsigned v = __hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  return (unsigned)__builtin_amdgcn_readfirstlane((int)v);
}
// a bounded wait gave up: the workgroup's error flag (every later wait of the workgroup leaves at its next check), the device counter
// and -- when savfi_sepconv_ws_watch() armed it -- the mapped host word the product reads without a device synchronisation
__device__ __forceinline__ void ws_give_up(unsigned* fl) {
  if ((threadIdx.x & 63) == 0) {
    __hip_atomic_store(fl + F_ERR, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    atomicAdd(&ws_error_count, 1u);
    // the mapped host word: its address is a kernel argument kept in LDS (a load from a device symbol here, or a flat atomic, makes
    // hipcc's vmcnt bookkeeping pessimistic for every wait the staging loop has: its row loads then waited for all drain stores)
    const unsigned long long q = (unsigned long long)fl[F_ERRW] | ((unsigned long long)fl[F_ERRW + 1] << 32);
    if (q) __hip_atomic_fetch_add(reinterpret_cast<ws_global_u32*>(q), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}
// one step of a spin loop; true = give up (limit < 0: at once -- the test hook savfi_sepconv_ws_debug_spin_limit)
__device__ __forceinline__ bool ws_spin_over(unsigned* fl, int& spins, int limit) {
  return ((++spins & 255) == 0 || limit < 0) && (ws_peek(fl + F_ERR) != 0u || spins > limit);
}
// flag >= target (sequence numbers of one run: no wrap)
__device__ __forceinline__ void ws_wait(unsigned* fl, int idx, int target) {
  asm volatile("" ::: "memory");
  if ((int)ws_peek(fl + idx) < target) {
    int spins = 0;
    const int limit = (int)ws_peek(fl + F_LIMIT);
    while (true) {
      __builtin_amdgcn_s_sleep(2);
      if ((int)ws_peek(fl + idx) >= target) break;
      if (ws_spin_over(fl, spins, limit)) { ws_give_up(fl); break; }
    }
  }
  asm volatile("" ::: "memory");
}
__device__ __forceinline__ void ws_set(unsigned* fl, int idx, int value) {
  asm volatile("" ::: "memory");
  __hip_atomic_store(fl + idx, (unsigned)value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  asm volatile("" ::: "memory");
}
// every wave's prog >= target (nw progress slots from fl[first])
template <int FIRST = F_PROG, int NW = 12>
__device__ __forceinline__ void ws_wait_all_prog(unsigned* fl, int target) {
  asm volatile("" ::: "memory");
  const int slot = min((int)(threadIdx.x & 15), NW - 1);
  int spins = 0;
  const int limit = (int)ws_peek(fl + F_LIMIT);
  while (true) {
    const unsigned v = __hip_atomic_load(fl + FIRST + slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (__builtin_amdgcn_ballot_w64((int)v < target) == 0ull) break;
    __builtin_amdgcn_s_sleep(2);
    if (ws_spin_over(fl, spins, limit)) { ws_give_up(fl); break; }
  }
  asm volatile("" ::: "memory");
}

// sum over the 64 lanes, every lane gets the total: four DPP steps inside each row of 16 lanes, then the four row totals by
// readlane (the ds_bpermute butterfly of __shfl_xor costs an LDS round trip per step: 18 of them per unit were a fifth of a
// staging wave's cycles)
__device__ __forceinline__ float ws_wave_sum(float x) {
  auto dpp = [](float v, auto ctrl) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), decltype(ctrl)::value, 0xf, 0xf, true));
  };
  x += dpp(x, std::integral_constant<int, 0xB1>{});      // quad_perm [1,0,3,2]
  x += dpp(x, std::integral_constant<int, 0x4E>{});      // quad_perm [2,3,0,1]
  x += dpp(x, std::integral_constant<int, 0x141>{});     // row_half_mirror
  x += dpp(x, std::integral_constant<int, 0x140>{});     // row_mirror
  auto rl = [](float v, int l) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l)); };
  return (rl(x, 0) + rl(x, 16)) + (rl(x, 32) + rl(x, 48));
}

// Six sums over the 64 lanes at once (round 5).  gfx950's v_permlane32_swap / v_permlane16_swap exchange halves / odd and even rows
// of 16 lanes between two registers: swap + add folds TWO values' halves (rows) and packs the partial sums into one register, so six
// values need 5 swaps + 5 adds and then ONE row reduction (4 DPP steps) on each of two registers -- 18 instructions where six
// ws_wave_sum take ~90 (24 of them v_readlane) and 2 300 of the forward's 5 100 cycles per unit on its v-side wave.
//   q : every lane of row 0 / 1 / 2 / 3 holds the total of x[0] / x[2] / x[1] / x[3];   q2: rows 0 / 2 hold the totals of x[4] / x[5]
__device__ __forceinline__ void ws_wave_sum6(const float (&x)[6], float& q, float& q2) {
  auto fold32 = [](float a, float b) {               // lanes 0..31: a[l] + a[l + 32], lanes 32..63: b[l - 32] + b[l]
    const auto r = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, a), __builtin_bit_cast(unsigned, b), false, false);
    const unsigned r0 = r[0], r1 = r[1];             // (a bit_cast of a vector ELEMENT reads element 0 whatever the index: DESIGN.md 4f)
    return __builtin_bit_cast(float, r0) + __builtin_bit_cast(float, r1);
  };
  auto fold16 = [](float a, float b) {               // rows: a.r0 + a.r1, b.r0 + b.r1, a.r2 + a.r3, b.r2 + b.r3
    const auto r = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned, a), __builtin_bit_cast(unsigned, b), false, false);
    const unsigned r0 = r[0], r1 = r[1];
    return __builtin_bit_cast(float, r0) + __builtin_bit_cast(float, r1);
  };
  auto dpp = [](float v, auto ctrl) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), decltype(ctrl)::value, 0xf, 0xf, true));
  };
  auto row_sum = [&](float v) {
    v += dpp(v, std::integral_constant<int, 0xB1>{});      // quad_perm [1,0,3,2]
    v += dpp(v, std::integral_constant<int, 0x4E>{});      // quad_perm [2,3,0,1]
    v += dpp(v, std::integral_constant<int, 0x141>{});     // row_half_mirror
    v += dpp(v, std::integral_constant<int, 0x140>{});     // row_mirror
    return v;
  };
  const float p01 = fold32(x[0], x[1]), p23 = fold32(x[2], x[3]), p45 = fold32(x[4], x[5]);
  q = row_sum(fold16(p01, p23));
  q2 = row_sum(fold16(p45, 0.f));
}

// the lane id through an opaque copy: values derived from it are temporaries of the place that uses them, not kernel-long live
// ranges (at the 168-register budget of three waves per SIMD the allocator spills hoisted per-lane constants, and one scratch
// reload in the MFMA wave's path costs a memory round trip: 1 700 cycles in front of a pass)
__device__ __forceinline__ int ws_lane() {
  int t = threadIdx.x;
  asm volatile("" : "+v"(t));
  return t & 63;
}

// Frames of 8-bit images.  The reference feeds the op decoded PNG frames (data/vimeo_septuplet.py: ToTensor = k / 255, k = 0..255;
// sepconv/model.py:346-347 passes them through a replication pad).  255 w = k is then an integer that ONE bf16 holds exactly: the window needs
// one piece instead of three, an fp32 product three bf16 products (k x the taps' three pieces, each exact) instead of six, and the result is
// scaled by 1 / 255 once (the cotangent in the filter gradients, the channel sums in the forward).  Whether a frame tensor has the property is
// decided ON THE DEVICE, per call: savfi_frames8_classify_f32 leaves one word per classifier workgroup (non-zero = it met an element that is
// not the fp32 quotient k / 255 to within 2 ulp), the <U8 = true> and <U8 = false> instances of a kernel are both launched and the one the
// words do not select returns at once -- no host round trip (graph-capture safe), any other input takes the six-product path unchanged.
// Which phases a workgroup takes.  The phases of all strips in strip-major order (position g: strip g / nph, phase g % nph) are cut into
// gridDim.x consecutive pieces; a piece that crosses into the next strip starts a second RUN there (a new window: 64 rows through two
// round trips, the first units' fetches, the pipeline's ramp -- 20 000 cycles = 1.5 phases measured, profiles/r05_ws_workgroup_times.txt:
// workgroups with two runs 422 000 cycles, with one 402 000, and the launch waits for the slowest).  So the cut is made on a COST axis of
// half phases on which every strip is WS_RUN_COST slots longer than its phases: a piece with a strip start inside gets that many half
// phases fewer.  (An "aligned" order -- the workgroups of a chunk walking the same rows of neighbouring strips together -- was measured
// slower three times, planar and unit-major taps alike: spread over the DRAM channels beats locality here; tools/r5/membench.hip.)
#ifndef WS_RUN_COST
#define WS_RUN_COST 2
#endif
__device__ __forceinline__ int ws_cost_to_phase(long long t, int nph) {
  const int C = 2 * nph + WS_RUN_COST;
  const int strip = (int)(t / C), r = (int)(t - (long long)strip * C);
  return strip * nph + min(max((r - WS_RUN_COST + 1) >> 1, 0), nph);
}
__device__ __forceinline__ void ws_work_range(int S, int nph, int& g0, int& g1) {
  const long long T = (long long)S * (2 * nph + WS_RUN_COST), G = gridDim.x, bx = blockIdx.x;
  g0 = ws_cost_to_phase(bx * T / G, nph);
  g1 = ws_cost_to_phase((bx + 1) * T / G, nph);
}
constexpr int CLS_WG = 256, CLS_NT = 1024;         // classifier grid: one 16-byte load of the words per lane of the consumer
template <bool U8>
__device__ __forceinline__ bool ws_frames8_mine(const unsigned* __restrict__ cls, const unsigned* __restrict__ cls2 = nullptr) {
  if (cls == nullptr) return !U8;
  u32x4 pv = reinterpret_cast<const u32x4*>(cls)[threadIdx.x & 63];
  if (cls2 != nullptr) {                             // two frame tensors (the pair launch): both have to qualify
    const u32x4 qv = reinterpret_cast<const u32x4*>(cls2)[threadIdx.x & 63];
    pv.x |= qv.x; pv.y |= qv.y; pv.z |= qv.z; pv.w |= qv.w;
  }
  const bool bad = __builtin_amdgcn_ballot_w64((pv.x | pv.y | pv.z | pv.w) != 0u) != 0ull;
  return bad != U8;
}
__device__ __forceinline__ bool ws_not_frames8(float w) {
  const float k = rintf(w * 255.f);
  const float d = fmaf(w, 255.f, -k);              // 255 w - k, rounded once: k delta for w = (k / 255)(1 + delta)
  return !(k >= 0.f && k <= 255.f && fabsf(d) <= k * 2.4e-7f);   // NaN, negative zero's k = -0 passes (w = -0 is k = 0)
}
__global__ __launch_bounds__(CLS_NT) void frames8_classify(const float* __restrict__ x, long long n, unsigned* __restrict__ cls) {
  const long long n4 = (reinterpret_cast<unsigned long long>(x) & 15ull) == 0ull ? n / 4 : 0;
  const long long t = (long long)blockIdx.x * CLS_NT + threadIdx.x, stride = (long long)CLS_WG * CLS_NT;
  bool bad = false;
  for (long long i = t; i < n4; i += 4 * stride) {
    f32x4 q[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) q[u] = reinterpret_cast<const f32x4*>(x)[min(i + u * stride, n4 - 1)];
#pragma unroll
    for (int u = 0; u < 4; ++u) bad |= (int)ws_not_frames8(q[u].x) | (int)ws_not_frames8(q[u].y) | (int)ws_not_frames8(q[u].z) | (int)ws_not_frames8(q[u].w);
  }
  for (long long i = 4 * n4 + t; i < n; i += stride) bad |= ws_not_frames8(x[i]);
  const int any = __syncthreads_or(bad ? 1 : 0);
  if (threadIdx.x == 0) cls[blockIdx.x] = any ? 1u : 0u;
}

template <bool U8, bool DMA = false>
__global__ __launch_bounds__(WNT) void sepconv_bwd_ws(const float* __restrict__ in, const float* __restrict__ v,
                                                      const float* __restrict__ h, const float* __restrict__ gO,
                                                      float* __restrict__ gV, float* __restrict__ gH,
                                                      int B, int Ho, int Wo, int nph, int ncol, int TB,
                                                      const unsigned* __restrict__ cls, int unit16, int spin_limit, unsigned* errw,
                                                      const float* __restrict__ in2, const unsigned* __restrict__ cls2, int pair) {
  // pair != 0 (savfi_sepconv_bwd_pair_frames8_f32): the TWO local convolutions of the plugin's tail in one launch.  B counts virtual samples
  // b' = 2 b + f: frame f of sample b -- `in` for f = 0, `in2` for f = 1 --, the cotangent of sample b, and the taps of sub-networks 2 f
  // (v) and 2 f + 1 (h) of the interleaved tensor, which are simply virtual sample b' at a stride of TB = 2 K planes.  Nothing else
  // changes: a launch has a fixed cost of about 24 us (the window prologue and the pipeline's ramp of every workgroup, the last
  // workgroup's tail, the other instance's early exit; B = 4: 97 us, B = 8: 170 us in the bench loop), so one launch of 2 B samples is
  // faster than two of B.
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int p = w & 3, wc = p & 1, wr0 = p >> 1;
  const int role = w >> 2;                          // 0: MFMA wave, 1: h-side staging wave, 2: v-side staging wave
  const bool staging = role != 0;
  const int j = lane & 15, kg = lane >> 4;
  char* const tab = smem + WPAIR_OFF + p * WPAIRB;
  float* const tile = reinterpret_cast<float*>(tab + XTAB);
  char* const tabv = tab;                          // one table per pair: the h band of unit n, then v of unit n, then h of unit n + 1 ...
  float* const tileh = tile;                       // one output tile per pair: gV, then gH
  float* const tailb = reinterpret_cast<float*>(tab + XTAB + WTILEB);
  float* const side = reinterpret_cast<float*>(smem + WSIDE_OFF);
  unsigned* const fl = reinterpret_cast<unsigned*>(smem + WFLAG_OFF);

  [[maybe_unused]] const unsigned long long t_kernel0 = WS_TRACE ? __builtin_readcyclecounter() : 0ull;
  // a workgroup's phases: positions [g0, g1) of the strip-major order (ws_work_range)
  int g0, g1;
  constexpr int base = 0;
  const int span = nph;
  ws_work_range(B * ncol, nph, g0, g1);
  if (g0 >= g1) return;
  if (!ws_frames8_mine<U8>(cls, pair ? cls2 : nullptr)) return;
  const int Hi = Ho + XK - 1, Wi = Wo + XK - 1;
  const unsigned plane_b = (unsigned)Ho * (unsigned)Wo * 4u;
  const unsigned tap_bytes = (unsigned)((B - 1) * TB + XK) * plane_b;
  const __amdgpu_buffer_rsrc_t hsrc = x6_rsrc(h, tap_bytes);
  const __amdgpu_buffer_rsrc_t vsrc = x6_rsrc(v, tap_bytes);
  const int Bimg = pair ? B >> 1 : B;               // samples of the frames and of the cotangent
  const __amdgpu_buffer_rsrc_t gsrc = x6_rsrc(gO, (unsigned)(Bimg * XC) * plane_b);
  const unsigned in_bytes = (unsigned)(Bimg * XC) * (unsigned)(Hi * Wi) * 4u;
  const __amdgpu_buffer_rsrc_t gvdst = x6_rsrc(gV, (unsigned)((B - 1) * TB + XK) * plane_b);
  const __amdgpu_buffer_rsrc_t ghdst = x6_rsrc(gH, (unsigned)((B - 1) * TB + XK) * plane_b);

  auto pix_off = [&](int b, int x0, int y, int ch) {
    return (unsigned)b * (unsigned)ch * plane_b + (unsigned)(min(y, Ho - 1) * Wo + min(x0 + 16 * wc + j, Wo - 1)) * 4u;
  };
  // Tap registers (as sepconv_bwd_x6): lane (j, kg) holds 7 PAIRS of neighbouring taps t0 + 8 a + {0, 1}: a split pair is exactly the
  // dword a table position takes.  v: t0 = 2 kg.  h: t0 = 2 kg - (j & 1): the band position i = tap + j of a pair starts even.
  // unit16: v and h are laid out [y][x / 16][tap][16] per sample (savfi_conv3x3_tasks_pre_unit16_f32) -- a unit's 51 x 64 bytes are
  // contiguous -- instead of [tap][y][x]; gV / gH stay [tap][y][x].  -DWS_EXP_UNITMAJOR=1 (timing only, results wrong): loads AND stores
  // addressed that way: what the op would cost with its gradients in that layout too
  const bool umaj = WS_EXP_UNITMAJOR || (unit16 & 1);
  const bool umaj_st = WS_EXP_UNITMAJOR == 1 || (unit16 & 2);        // gV / gH unit-major as well (bit 1 of taps_unit16)
  const unsigned tap_stride = umaj ? 64u : plane_b;
  auto unit_off = [&](int b, int x0, int y) {
    return (unsigned)b * (unsigned)TB * plane_b + (unsigned)((min(y, Ho - 1) * (Wo >> 4) + min((x0 >> 4) + wc, (Wo >> 4) - 1)) * XK) * 64u;
  };
  auto load_taps = [&](float (&regs)[XNP][2], __amdgpu_buffer_rsrc_t src, int b, int x0, int y, int t0) {
    const unsigned pix = umaj ? unit_off(b, x0, y) + (unsigned)j * 4u : pix_off(b, x0, y, TB);
    const unsigned voff = pix + (unsigned)(t0 + 1) * tap_stride;
    regs[0][0] = x6_bload(src, pix + (unsigned)max(t0, 0) * tap_stride, 0u);
    regs[0][1] = x6_bload(src, voff, 0u);
#pragma unroll
    for (int a = 1; a < XNP - 1; ++a)
#pragma unroll
      for (int e = 0; e < 2; ++e) regs[a][e] = x6_bload(src, voff, (unsigned)(8 * a + e - 1) * tap_stride);
#pragma unroll
    for (int e = 0; e < 2; ++e) regs[XNP - 1][e] = x6_bload(src, pix + (unsigned)min(8 * (XNP - 1) + t0 + e, XK - 1) * tap_stride, 0u);
  };
  auto tap_or_zero = [&](const float (&regs)[XNP][2], int a, int e, int t0) {
    if (a == 0 && e == 0) return t0 < 0 ? 0.f : regs[0][0];
    if (a == XNP - 1) return (8 * (XNP - 1) + t0 + e < XK) ? regs[a][e] : 0.f;
    return regs[a][e];
  };
  const int h_t0 = 2 * kg - (j & 1), v_t0 = 2 * kg;
  // DMA (U8, unit-major taps): a unit's 51 x 64 bytes are ONE contiguous run, and the U8 window leaves planes 3..8 of the window area
  // unused.  Every staging wave owns two slots there and fetches a unit's bytes TWO units ahead with LDS-DMA loads (buffer_load ... lds:
  // no destination registers, so depth costs nothing): four 16-byte-per-lane pieces of the run (1 KB per instruction; lanes past byte 3264
  // switched off by the range check) and one dword piece with the unit's cotangent [channel][16 pixels] behind them.  It picks its seven
  // tap pairs out of the slot with ds_read_b32.  The slot's bytes are ordered for the wave's own reads by a COUNTED s_waitcnt: every
  // iteration issues exactly WS_DMA_ITER vector memory operations after the fetch of the unit that the next iteration reads.
  static_assert(!DMA || U8, "the ring lives in the window planes only the U8 window leaves free");
  static_assert(3 * XPLANE + 8 * 2 * WRINGB <= XWINB, "eight waves x two slots in planes 3..8");
  char* const ring = smem + 3 * XPLANE + (staging ? w - 4 : 0) * 2 * WRINGB;
  // (inline assembly, not the raw_ptr_buffer_load_lds builtin: hipcc orders every later LDS read that may alias behind a fetch it knows
  // of -- the next flag read would wait for the unit just requested.  M0 carries the slot's LDS address and is restored; the instruction
  // offset moves the memory and the LDS address alike.)
  // n: the unit's number in the run; the wave's share (64 columns x two channels) of the window row that iteration n writes rides along
  auto dma_unit = [&](const float* src, unsigned src_bytes, const float* frame, int b, int bi, int x0, int R0, int wr0_, int n, int slot) {
    const int y = R0 + XPR * (n >> 1) + wr0_ + 2 * (n & 1);
    const unsigned dst = (unsigned)(unsigned long long)(ws_lds_void*)(ring + slot * WRINGB);
    const unsigned run = unit_off(b, x0, y) + (unsigned)lane * 16u;
    static_assert((XK * 64 - 3072) / 16 == 12, "lanes of the fourth piece");
    const unsigned go = lane < 16 * XC ? pix_off(bi, x0, y, XC) + (unsigned)kg * plane_b : X_OOR;
    const int ws_ = w - 4, gcol_ = (ws_ & 1) * 64 + lane, gc0_ = (ws_ & 2) ? 2 : 0, gc1_ = (ws_ & 2) ? 2 : 1;
    const int rr = min(R0 + 60 + 2 * n + (ws_ >> 2), Hi - 1);
    const unsigned colb = (unsigned)min(x0 + gcol_, Wi - 1) * 4u;
    const unsigned r0 = (unsigned)(((bi * XC + gc0_) * Hi + rr) * Wi) * 4u + colb, r1 = (unsigned)(((bi * XC + gc1_) * Hi + rr) * Wi) * 4u + colb;
    const ws_i32x4 rs = ws_rsrc4(src, src_bytes), rg = ws_rsrc4(gO, (unsigned)(Bimg * XC) * plane_b);
    const ws_i32x4 ri = ws_rsrc4(frame, in_bytes);
    unsigned keep;
    unsigned long long keepx;
    // the fourth piece under an EXEC of twelve lanes: a lane switched off by the range check would still write zeros over the slot's tail
    asm volatile("s_mov_b32 %0, m0\n\t"
                 "s_mov_b32 m0, %2\n\t"
                 "s_mov_b64 %1, exec\n\t"
                 "buffer_load_dwordx4 %6, %10, 0 offen" WS_DMA_TAP_POLICY " lds\n\t"
                 "buffer_load_dwordx4 %6, %10, 0 offen offset:1024" WS_DMA_TAP_POLICY " lds\n\t"
                 "buffer_load_dwordx4 %6, %10, 0 offen offset:2048" WS_DMA_TAP_POLICY " lds\n\t"
                 "s_mov_b64 exec, 0xfff\n\t"
                 "buffer_load_dwordx4 %6, %10, 0 offen offset:3072" WS_DMA_TAP_POLICY " lds\n\t"
                 "s_mov_b64 exec, %1\n\t"
                 "s_mov_b32 m0, %3\n\t"
                 "s_nop 0\n\t"
                 "buffer_load_dword %7, %11, 0 offen lds\n\t"
                 "s_mov_b32 m0, %4\n\t"
                 "s_nop 0\n\t"
                 "buffer_load_dword %8, %12, 0 offen lds\n\t"
                 "s_mov_b32 m0, %5\n\t"
                 "s_nop 0\n\t"
                 "buffer_load_dword %9, %12, 0 offen lds\n\t"
                 "s_mov_b32 m0, %0"
                 : "=&s"(keep), "=&s"(keepx)
                 : "s"(dst), "s"(dst + WRING_GP), "s"(dst + WRING_ROW), "s"(dst + WRING_ROW + 256), "v"(run), "v"(go), "v"(r0), "v"(r1),
                   "s"(rs), "s"(rg), "s"(ri)
                 : "memory");
  };
  auto slot_taps = [&](float (&regs)[XNP][2], int t0, int slot) {   // the registers load_taps fills, out of the slot [tap][16 pixels]
    const float* const sl = reinterpret_cast<const float*>(ring + slot * WRINGB) + j;
    regs[0][0] = sl[max(t0, 0) * 16];
    regs[0][1] = sl[(t0 + 1) * 16];
#pragma unroll
    for (int a = 1; a < XNP - 1; ++a)
#pragma unroll
      for (int e = 0; e < 2; ++e) regs[a][e] = sl[(t0 + 8 * a + e) * 16];
#pragma unroll
    for (int e = 0; e < 2; ++e) regs[XNP - 1][e] = sl[min(8 * (XNP - 1) + t0 + e, XK - 1) * 16];
  };
  // h band of the unit's 16 pixels -> table [piece][i / 8][j][i % 8], i = tap + j < 64 (zero elsewhere)
  auto write_h_table = [&](const float (&regs)[XNP][2]) {
#pragma unroll
    for (int k = 0; k < XTAB / 1024; ++k) *reinterpret_cast<u32x4*>(tab + (k * 64 + lane) * 16) = (u32x4){0u, 0u, 0u, 0u};
    const int base2 = 2 * kg + (j & ~1);
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
  // v taps -> table position of tap fy: k step fy / 32, k group (fy % 16) / 4, element fy % 4 + 4 * ((fy / 16) % 2)
  auto write_v_table = [&](const float (&regs)[XNP][2]) {
    char* const lb = tabv + (kg >> 1) * 256 + j * 16 + (kg & 1) * 4;
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
  auto rdlane = [&](float x, int l) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x), l)); };

  constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};   // six products, small terms first
  constexpr int PB8[3] = {2, 1, 0};                  // U8: the window is one piece (PA = 0)
  constexpr int NPC = U8 ? 1 : 3, NQ = U8 ? 3 : 6;   // window pieces, products per fp32 product
  constexpr int DEPTH = U8 ? 2 : 1, NSL = DEPTH + 1; // A fragments are requested DEPTH blocks ahead (a U8 block is 6 MFMAs: half the cover)
  const int pq = lane & 3, fq = lane >> 2;

  // The run loop exists twice, once per role (a wave never changes role): in one loop with a run-time branch every per-lane constant
  // of the staging program stays live through the MFMA program and vice versa (168 registers: 44 spilled, 180 bytes of scratch per lane).
  auto run_all = [&](auto stg_) __attribute__((always_inline)) {
  constexpr bool STG = decltype(stg_)::value;
  int g = g0;
#pragma unroll 1
  while (g < g1) {
    // ---- a run: phases ph0 .. ph0 + nrun - 1 of strip (b, x0) ---------------------------------------------------------------
    const int s = g / span, gin = g - s * span, ph0 = base + gin, b = s / ncol, x0 = (s - b * ncol) * XMC;
    const int bi = pair ? b >> 1 : b;
    const float* const frame = (pair && (b & 1)) ? in2 : in;
    const __amdgpu_buffer_rsrc_t isrc = x6_rsrc(frame, in_bytes);
    const int run_end = min(g1, g + (span - gin)), nrun = run_end - g, N = 2 * nrun;
    const int R0 = XPR * ph0;
    auto unit_y = [&](int n) { return R0 + XPR * (n >> 1) + wr0 + 2 * (n & 1); };

    float gp[XC];                                   // cotangent of the unit at hand (staging: of the next one once the tails have theirs)
    __syncthreads();                                // every wave has left the previous run's window, tables and flags
    if (tid < 64) {
      const unsigned long long q = reinterpret_cast<unsigned long long>(errw);
      fl[tid] = tid == F_LIMIT ? (unsigned)spin_limit : tid == F_ERRW ? (unsigned)q : tid == F_ERRW + 1 ? (unsigned)(q >> 32) : 0u;
    }
    {
      const unsigned go = pix_off(bi, x0, unit_y(0), XC);
#pragma unroll
      for (int c = 0; c < XC; ++c) gp[c] = x6_bload(gsrc, go, (unsigned)c * plane_b);
    }
    if constexpr (DMA && STG) {                      // the first two units' fetches: their latency runs under the window prologue
      dma_unit(role == 1 ? h : v, tap_bytes, frame, b, bi, x0, R0, wr0, 0, 0);
      dma_unit(role == 1 ? h : v, tap_bytes, frame, b, bi, x0, R0, wr0, min(1, N - 1), 1);
    }
    if (tid < XNT) {                                 // the window prologue keeps sepconv_x6's mapping of 512 threads
      constexpr int PR = U8 ? 32 : 16;               // rows per round trip (the U8 kernel has the registers for two rounds instead of four)
#pragma unroll 1
      for (int r = 0; r < XWIN; r += PR) {
        X6Rows<PR> sr;
        x6_rows_load<PR>(sr, isrc, bi, x0, R0 + r, Hi, Wi, tid);
        x6_rows_write<PR, U8>(sr, smem, R0 + r, tid, WSIDE_OFF);
      }
    }
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();

    if constexpr (!STG) {
      // =========================================== MFMA wave ===================================================================
      // One stream of passes: gV(0), gH(0), gV(1), ...  Behind the last MFMA block of a pass the NEXT pass's flag checks, B fragments
      // and first A fragments are issued BEFORE this pass's epilogue (cotangent scaling, tile write), so that their LDS latencies
      // run under the epilogue's VALU work instead of in front of the next pass's first MFMA.
      __builtin_amdgcn_s_setprio(WS_PRIO);
      unsigned long long tr_[16] = {0}, tlast_ = __builtin_readcyclecounter();
      bf16x8 bq[2][3], aq[NSL][2][NPC];
      if (WS_EXP_NOAREAD) {                          // (timing / counter experiment: the A fragments are never read)
#pragma unroll
        for (int a = 0; a < NSL; ++a)
#pragma unroll
          for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int pc = 0; pc < NPC; ++pc) aq[a][t][pc] = (bf16x8){1, 2, 3, 4, 5, 6, 7, 8};
      }
      int rowoff[4], rowh[2][2];
      // the same offsets + 6 planes: a DS instruction's immediate offset is 16 bits and the window is 103 KB -- reads of the third bf16 piece
      // (planes 6..8) through the low bases took a v_add each, inside the MFMA loops (4 per gH block, 2 per gV block)
      int rowoff_hi[4], rowh_hi[2][2];
      auto peek_raw = [&](int idx) { return __hip_atomic_load(fl + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); };
      auto set_rows = [&](int y) {
        // (recomputed from the lane id each time: hoisted out of the unit loop these per-lane constants are spilled to scratch at the
        // 168-register budget of three waves per SIMD, and every reload is a memory round trip in front of the next pass)
        const int lo_ = ws_lane(), jo = lo_ & 15, ko = lo_ >> 4, Lo = jo;
        const int pk = ((ko & 1) << 1) | (ko >> 1);
#pragma unroll
        for (int m = 0; m < 3; ++m) rowoff[m] = ((y + 16 * m + jo) & (XWIN - 1)) * 16 + (2 * wc + pk) * XBLK;
        // the packed tile: lane row 4 c + r = tap row 48 + r of channel c (r < 3, c < 3; the other four rows are computed and ignored)
        rowoff[3] = ((y + 48 + min(jo & 3, 2)) & (XWIN - 1)) * 16 + (2 * wc + pk) * XBLK + min(jo >> 2, 2) * XPLANE;
        // transpose read: source lane L of group kg points at window row y + 32 s + 16 half + 4 kg + L / 4, column quad L % 4 of the tile
#pragma unroll
        for (int st = 0; st < 2; ++st)
#pragma unroll
          for (int hf = 0; hf < 2; ++hf)
            rowh[st][hf] = ((y + 32 * st + 16 * hf + 4 * ko + (Lo >> 2)) & (XWIN - 1)) * 16 + (2 * wc + ((Lo & 3) >> 1)) * XBLK + (Lo & 1) * 8;
        if constexpr (!U8) {
#pragma unroll
          for (int m = 0; m < 4; ++m) { rowoff_hi[m] = rowoff[m] + 6 * XPLANE; asm volatile("" : "+v"(rowoff_hi[m])); }
#pragma unroll
          for (int st = 0; st < 2; ++st)
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) { rowh_hi[st][hf] = rowh[st][hf] + 6 * XPLANE; asm volatile("" : "+v"(rowh_hi[st][hf])); }
        }
      };
      // gV blocks (12 MFMAs each: two tiles x six products).  Tap rows 0..47 of a channel are three tiles of 16; rows 48..50 of the
      // THREE channels share one more tile (a lane's A row is any window row of any channel), so 51 rows x 3 channels cost 10 tiles
      // instead of 12: blocks 0..5 = (channel, k step) x tiles {0, 1}; 6, 7 = k step x {tile 2 of channel 0, of channel 1};
      // 8, 9 = k step x {tile 2 of channel 2, packed tile}.  Fragment (uu, t) -> accumulator index, plane of the channel, row base.
      auto gv_acc = [](int uu, int t) { return uu < 6 ? 3 * (uu >> 1) + t : uu < 8 ? 3 * t + 2 : (t == 0 ? 8 : 9); };
      auto gv_st = [](int uu) { return uu & 1; };
      auto load_av = [&](int slot, int uu) {           // lane = window row of the tile, 16 B = 8 columns of k group permk
        if (WS_EXP_NOAREAD) return;
        const int st = gv_st(uu);
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          const int ai = gv_acc(uu, t), c = ai == 9 ? 0 : ai / 3, m = ai == 9 ? 3 : ai % 3;
#pragma unroll
          for (int pc = 0; pc < NPC; ++pc)
          {
            const int plane = pc * 3 + c;
            aq[slot][t][pc] = *reinterpret_cast<const bf16x8*>(smem + (plane >= 6 ? plane - 6 : plane) * XPLANE + 4 * st * XBLK + (plane >= 6 ? rowoff_hi[m] : rowoff[m]));
          }
        }
      };
      auto load_ah = [&](int slot, int uu) {           // gH: two transpose reads per fragment
        if (WS_EXP_NOAREAD) return;
        const int c = uu >> 2, st = (uu >> 1) & 1, mp = uu & 1;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int pc = 0; pc < NPC; ++pc) {
            const int plane = pc * 3 + c;
            const int base = (plane >= 6 ? plane - 6 : plane) * XPLANE + 2 * (2 * mp + t) * XBLK;
            const bf16x4 lo = tr_read(base + (plane >= 6 ? rowh_hi[st][0] : rowh[st][0])), hi = tr_read(base + (plane >= 6 ? rowh_hi[st][1] : rowh[st][1]));
            aq[slot][t][pc] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
          }
      };
      auto read_bh = [&]() {
        const int lo_ = ws_lane(), jo = lo_ & 15, ko = lo_ >> 4, pk = ((ko & 1) << 1) | (ko >> 1);
#pragma unroll
        for (int st = 0; st < 2; ++st)
#pragma unroll
          for (int pc = 0; pc < 3; ++pc) bq[st][pc] = *reinterpret_cast<const bf16x8*>(tab + pc * XTABP + (4 * st + pk) * 256 + jo * 16);
      };
      auto read_bv = [&]() {
        const int lo_ = ws_lane(), jo = lo_ & 15, ko = lo_ >> 4;
#pragma unroll
        for (int st = 0; st < 2; ++st)
#pragma unroll
          for (int pc = 0; pc < 3; ++pc) bq[st][pc] = *reinterpret_cast<const bf16x8*>(tabv + pc * XTABP + (4 * st + ko) * 256 + jo * 16);
      };
      // the twelve MFMA blocks of a pass (block = channel c, k step st, tile pair mp: 12 MFMAs); the next block's fragments are
      // requested between this block's MFMAs (one LDS read per matrix-pipe gap).  The caller has issued load_a(0, 0).
#define WS_MFMA_BLOCK(uu)                                                                                                             \
      {                                                                                                                               \
        const int c = (uu) >> 2, st = ((uu) >> 1) & 1, mp = (uu) & 1;                                                                 \
        _Pragma("unroll") for (int qq = 0; qq < NQ; ++qq)                                                                             \
          _Pragma("unroll") for (int t = 0; t < 2; ++t)                                                                               \
            acc[c][2 * mp + t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(aq[(uu) % NSL][t][U8 ? 0 : PA[qq]], bq[st][U8 ? PB8[qq] : PB[qq]], acc[c][2 * mp + t], 0, 0, 0); \
      }

      // before the first pass: h fragments of unit 0, first A fragments
      set_rows(unit_y(0));
      ws_wait(fl, F_TAB_FULL + p, 1);
      read_bh();
      ws_set(fl, F_TAB_FREE + p, 1);
#pragma unroll
      for (int d = 0; d < DEPTH; ++d) load_av(d, d);
#pragma unroll 1
      for (int n = 0; n < N; ++n) {
        const int q = n >> 1, u = n & 1;
        WS_T(0);
        float g_[XC];                                  // U8: the window holds 255 w -- the cotangent carries the 1 / 255
#pragma unroll
        for (int c = 0; c < XC; ++c) g_[c] = U8 ? gp[c] * (1.0f / 255.0f) : gp[c];
        {
          const unsigned go = pix_off(bi, x0, unit_y(min(n + 1, N - 1)), XC);
#pragma unroll
          for (int c = 0; c < XC; ++c) gp[c] = x6_bload(gsrc, go, (unsigned)c * plane_b);
        }
        // ---- gV: bq = h band, aq[0] = block 0 ----
        {
          f32x4 acc[10];                                 // [3 c + m] (m < 3), [9] = packed tile
#pragma unroll
          for (int i = 0; i < 10; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
          unsigned pre_tab = 0u, pre_out = 0u;
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int uu = 0; uu < 10; ++uu) {
            if (uu + DEPTH < 10) load_av((uu + DEPTH) % NSL, uu + DEPTH);
            if (uu == 9) { pre_tab = peek_raw(F_TAB_FULL + p); pre_out = peek_raw(F_OUT_FREE + p); }
            {
              const int st = gv_st(uu);
#pragma unroll
              for (int qq = 0; qq < NQ; ++qq)
#pragma unroll
                for (int t = 0; t < 2; ++t)
                  acc[gv_acc(uu, t)] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(aq[uu % NSL][t][U8 ? 0 : PA[qq]], bq[st][U8 ? PB8[qq] : PB[qq]], acc[gv_acc(uu, t)], 0, 0, 0);
            }
            if (WS_INTERLEAVE && uu + DEPTH < 10) {       // MFMA, DS read, MFMA, DS read ... (2 NPC reads), then the remaining MFMAs
#pragma unroll
              for (int i = 0; i < 2 * NPC; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
              }
              __builtin_amdgcn_sched_group_barrier(0x008, 2 * NQ - 2 * NPC, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
          }
          WS_T(3);
          // the next pass (gH of this unit): v fragments, first A fragments -- before this pass's epilogue
          if ((int)__builtin_amdgcn_readfirstlane((int)pre_tab) < (2 * n + 2)) ws_wait(fl, F_TAB_FULL + p, 2 * n + 2);
          read_bv();
          ws_set(fl, F_TAB_FREE + p, 2 * n + 2);
#pragma unroll
          for (int d = 0; d < DEPTH; ++d) load_ah(d, d);
          WS_T(4);
          // D row 4 kg + r of tile m = tap row fy = 16 m + 4 kg + r (m < 3), column = pixel j; the packed tile's lane group kg holds
          // rows 48 + r of channel kg: scaled by that channel's cotangent it goes to tile row 64 + 4 kg + r, the h-side wave adds the
          // three channels when it drains rows 48..50
          float val[4][4];
#pragma unroll
          for (int m = 0; m < 3; ++m)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              float t = g_[0] * acc[m][r];
              t = fmaf(g_[1], acc[3 + m][r], t);
              t = fmaf(g_[2], acc[6 + m][r], t);
              val[m][r] = t;
            }
          const int lo_ = ws_lane();
          {
            const int ko = lo_ >> 4;
            const float gs = ko == 0 ? g_[0] : ko == 1 ? g_[1] : ko == 2 ? g_[2] : 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) val[3][r] = gs * acc[9][r];
          }
          if ((int)__builtin_amdgcn_readfirstlane((int)pre_out) < (2 * n)) ws_wait(fl, F_OUT_FREE + p, 2 * n);
          WS_T(5);
          {
            float* const tw = tile + (4 * (lo_ >> 4)) * WPV + (lo_ & 15);
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
              for (int r = 0; r < 4; ++r) tw[((m < 3 ? 16 * m : 64) + r) * WPV] = val[m][r];
          }
          ws_set(fl, F_OUT_FULL + p, 2 * n + 1);
          WS_T(6);
        }
        // ---- gH: bq = v, aq[0] = block 0 ----
        {
          f32x4 acc[XC][4];
#pragma unroll
          for (int c = 0; c < XC; ++c)
#pragma unroll
            for (int m = 0; m < 4; ++m) acc[c][m] = (f32x4){0.f, 0.f, 0.f, 0.f};
          unsigned pre_tab = 0u, pre_out = 0u, pre_slide = 0u;
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int uu = 0; uu < 12; ++uu) {
            if (uu + DEPTH < 12) load_ah((uu + DEPTH) % NSL, uu + DEPTH);
            if (uu == 11) { pre_tab = peek_raw(F_TAB_FULL + p); pre_out = peek_raw(F_OUT_FREE + p); pre_slide = peek_raw(F_SLIDE); }
            WS_MFMA_BLOCK(uu)
            if (WS_INTERLEAVE && uu + DEPTH < 12) {
              if constexpr (U8) {                          // MFMA, transpose read, ... (4 reads over the first 4 of 6 MFMAs)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                  __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                  __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                }
                __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
              } else {                                     // MFMA, 2 transpose reads, ... (24 reads over the 12 MFMAs)
#pragma unroll
                for (int i = 0; i < 12; ++i) {
                  __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                  __builtin_amdgcn_sched_group_barrier(0x100, WS_GH_READS_PER_GAP, 0);
                }
              }
            }
            __builtin_amdgcn_sched_barrier(0);
          }
          WS_T(9);
          if (u == 1) ws_set(fl, F_PROG + w, q + 1);           // this phase's window reads are over
          // the next pass (gV of the next unit): h fragments, first A fragments -- before this pass's epilogue
          if (n + 1 < N) {
            const int q1 = (n + 1) >> 1;
            if ((int)__builtin_amdgcn_readfirstlane((int)pre_tab) < (2 * n + 3)) ws_wait(fl, F_TAB_FULL + p, 2 * n + 3);
            WS_T(13);
            read_bh();
            ws_set(fl, F_TAB_FREE + p, 2 * n + 3);
            if (u == 1 && q1 >= 2 && (int)__builtin_amdgcn_readfirstlane((int)pre_slide) < 8 * (2 * q1 - 3)) ws_wait(fl, F_SLIDE, 8 * (2 * q1 - 3));
            WS_T(14);
            set_rows(unit_y(n + 1));
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) load_av(d, d);
          }
          WS_T(10);
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
          if ((int)__builtin_amdgcn_readfirstlane((int)pre_out) < (2 * n + 1)) ws_wait(fl, F_OUT_FREE + p, 2 * n + 1);
          WS_T(11);
          {
            const int lo_ = ws_lane();
            float* const tw = tileh + (4 * (lo_ >> 4) - (lo_ & 15) + 15) * WPV + (lo_ & 15);
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
              for (int r = 0; r < 4; ++r) tw[(16 * m + r) * WPV] = val[m][r];
          }
          ws_set(fl, F_OUT_FULL + p, 2 * n + 2);
          WS_T(12);
        }
      }
#undef WS_MFMA_BLOCK
      __builtin_amdgcn_s_setprio(0);
      if (WS_TRACE && blockIdx.x == 0 && lane == 0)
        for (int k = 0; k < 16; ++k) WS_TRACE_ADD(w, k, tr_[k]);
    } else {
      // =========================================== staging waves ===============================================================
      // role 1 (waves 4..7): the h side of the pair's units -- h taps, band table, gV tail columns, gV tile -> HBM, window row 60 + 2 n
      // role 2 (waves 8..11): the v side -- v taps, table, gH tail columns, gH tile -> HBM, window row 61 + 2 n
      const bool hside = role == 1;
      const int ptid = tid - (hside ? 256 : 512);
      const int gcol = ptid & 127, ggrp = ptid >> 7;        // granule: one window row per role, thread = (column, channels {0, 1} | {2})
      const int gcell = (gcol >> 3) * XBLK + (gcol & 7) * 2;
      const int gsidx = gcol == 64 ? 0 : gcol == 65 ? 1 : gcol == 80 ? 2 : gcol == 81 ? 3 : -1;
      const unsigned gcolb = (unsigned)min(x0 + gcol, Wi - 1) * 4u;
      const int gc0 = ggrp == 0 ? 0 : 2, gc1 = ggrp == 0 ? 1 : 2;
      unsigned long long tr_[16] = {0}, tlast_ = __builtin_readcyclecounter();
      // taps of the unit at hand, then of the next one (one of the two per role; DMA: ONE array -- the slot reads assign it at the top of
      // every iteration, and a second, never assigned array would be carried around the loop as sixteen register copies)
      float treg_a[XNP][2], treg_b[XNP][2];
      float (&hreg)[XNP][2] = treg_a;
      float (&vreg)[XNP][2] = DMA ? treg_a : treg_b;
      if constexpr (DMA) {
        // (units 0 and 1 were requested before the window prologue)
      } else if (hside) load_taps(hreg, hsrc, b, x0, unit_y(0), h_t0);
      else load_taps(vreg, vsrc, b, x0, unit_y(0), v_t0);
      float s6414 = 0.f, s6415 = 0.f, s6515 = 0.f;   // v side: gH tail sums of the unit whose tile is drained next
      unsigned qoff_prev = X_OOR;
      // Iteration n: (1) the table of unit n -- FIRST: it is what the MFMA wave waits for next --, (2) the tile of unit n - 1 -> HBM,
      // (3) tail columns of unit n, taps of unit n + 1, (4) one new window row.  Iteration N only drains the last tile.
#pragma unroll 1
      for (int n = 0; n <= N; ++n) {
        const bool live = n < N;
        const int nn = min(n, N - 1);
        const int q = nn >> 1, u = nn & 1;
        const int y = unit_y(nn), y1 = unit_y(min(nn + 1, N - 1));
        const int xq = x0 + 16 * wc + 4 * pq;
        const unsigned qoff = !(live && y < Ho && xq < Wo) ? X_OOR
                              : umaj_st ? unit_off(b, x0, y) + (unsigned)pq * 16u + (unsigned)fq * 64u
                                                 : (unsigned)b * (unsigned)TB * plane_b + (unsigned)(y * Wo + xq) * 4u + (unsigned)fq * plane_b;
        if constexpr (DMA) {
          // unit n's bytes have landed in slot n & 1: everything but the last iteration's WS_DMA_ITER operations (before the first
          // iteration: but the second unit's fetch) has completed
          asm volatile("" ::: "memory");
          // (assembly: with the builtin hipcc's own bookkeeping of THIS loop turned pessimistic -- its wait for the row loads became vmcnt(1))
          static_assert(WS_DMA_UNIT == 7 && WS_DMA_ITER == 11, "the immediates below");
          if (n == 0) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
          else asm volatile("s_waitcnt vmcnt(11)" ::: "memory");
          WS_T(11);
          asm volatile("" ::: "memory");
          slot_taps(treg_a, hside ? h_t0 : v_t0, n & 1);
          const float* const gs = reinterpret_cast<const float*>(ring + (n & 1) * WRINGB + WRING_GP) + j;
#pragma unroll
          for (int c = 0; c < XC; ++c) gp[c] = gs[16 * c];
          asm volatile("" ::: "memory");
        }
        // one new window row per role and unit (rows R0 + 60 + 2 n + {0 | 1})
        float gr0, gr1;
        const int grow = R0 + 60 + 2 * nn + (hside ? 0 : 1);
        {
          const int rr = min(grow, Hi - 1);
          if constexpr (DMA) {                       // fetched with the unit: read back where the row is written
            gr0 = gr1 = 0.f;
          } else {
            gr0 = WS_EXP_NOSTAGE ? 0.f : x6_bload(isrc, (unsigned)(((bi * XC + gc0) * Hi + rr) * Wi) * 4u + gcolb, 0u);
            gr1 = WS_EXP_NOSTAGE ? 0.f : x6_bload(isrc, (unsigned)(((bi * XC + gc1) * Hi + rr) * Wi) * 4u + gcolb, 0u);
          }
        }
        float g14[XC], g15[XC];
#pragma unroll
        for (int c = 0; c < XC; ++c) { g14[c] = rdlane(gp[c], 14); g15[c] = rdlane(gp[c], 15); }
        const int fyl = min(lane, XK - 1);
        const int tslot = (y + fyl) & (XWIN - 1);
        WS_T(0);
        if (hside) {
          // what the tail columns need of this unit's taps, before the registers take the next unit's
          const float h50_14 = rdlane(hreg[6][0], 14 + 16), h49_15 = rdlane(hreg[6][0], 15 + 16), h50_15 = rdlane(hreg[6][1], 15 + 16);
          // (1) the h band of unit n takes the table (the MFMA wave holds v of unit n - 1 in registers)
          if (live) {
            ws_wait(fl, F_TAB_FREE + p, 2 * n);
            WS_T(1);
            __builtin_amdgcn_s_setprio(WS_PRIO_TABLE);
            if (!WS_EXP_NOSTAGE) write_h_table(hreg);
            ws_set(fl, F_TAB_FULL + p, 2 * n + 1);
            __builtin_amdgcn_s_setprio(0);
            WS_T(2);
          }
          // (2) the gV tile of unit n - 1 -> HBM (tail sums of that unit still in tailb)
          if (n > 0) ws_wait(fl, F_OUT_FULL + p, 2 * n - 1);
          WS_T(6);
#pragma unroll
          for (int qq = 0; qq < (WS_EXP_NOSTAGE ? 0 : 4); ++qq) {
            const int row = fq + 16 * qq;
            f32x4 v4 = *reinterpret_cast<const f32x4*>(tile + (qq < 3 ? row : 64 + (fq & 3)) * WPV + 4 * pq);
            if (qq == 3) {      // tap rows 48..50 (fq < 3): the three channels' scaled rows of the packed tile
              const f32x4 c1 = *reinterpret_cast<const f32x4*>(tile + (68 + (fq & 3)) * WPV + 4 * pq);
              const f32x4 c2 = *reinterpret_cast<const f32x4*>(tile + (72 + (fq & 3)) * WPV + 4 * pq);
#pragma unroll
              for (int e = 0; e < 4; ++e) v4[e] = (v4[e] + c1[e]) + c2[e];
            }
            const float t14 = tailb[row], t15 = tailb[64 + row];
            if (pq == 3) { v4[2] += t14; v4[3] += t15; }
            x6_bstore4(v4, gvdst, (qq < 3 || fq < 3) ? qoff_prev : X_OOR, (unsigned)(16 * qq) * (umaj_st ? 64u : plane_b));
          }
          if (n > 0) ws_set(fl, F_OUT_FREE + p, 2 * n - 1);
          WS_T(7);
          // (3) tail columns of gV (i = 64: pixel 14 tap 50, pixel 15 tap 49; i = 65: pixel 15 tap 50): lane = tap row fy
          if (live && u == 0 && q >= 2) ws_wait(fl, F_SLIDE, 8 * (2 * q - 3));
          float a64[XC], a65[XC];
#pragma unroll
          for (int c = 0; c < XC; ++c) {
            const f32x2 sv = *reinterpret_cast<const f32x2*>(side + (c * XWIN + tslot) * 4 + 2 * wc);
            a64[c] = sv.x; a65[c] = sv.y;
          }
          asm volatile("" ::: "memory");
          if (live && u == 1) ws_set(fl, F_PROG + w, q + 1);
          if constexpr (!DMA) if (!WS_EXP_NOSTAGE) load_taps(hreg, hsrc, b, x0, y1, h_t0);
          {
            float t14 = 0.f, t15 = 0.f;
#pragma unroll
            for (int c = 0; c < XC; ++c) {
              t14 = fmaf(g14[c] * h50_14, a64[c], t14);
              t15 = fmaf(g15[c] * h49_15, a64[c], t15);
              t15 = fmaf(g15[c] * h50_15, a65[c], t15);
            }
            tailb[lane] = lane < XK ? t14 : 0.f;
            tailb[64 + lane] = lane < XK ? t15 : 0.f;
          }
          WS_T(3);
        } else {
          // (1) v of unit n takes the table (the MFMA wave holds the h band of unit n in registers)
          float v14 = 0.f, v15 = 0.f;
          if (live) {
            ws_wait(fl, F_TAB_FREE + p, 2 * n + 1);
            WS_T(1);
            __builtin_amdgcn_s_setprio(WS_PRIO_TABLE);
            if (!WS_EXP_NOSTAGE) write_v_table(vreg);
            // v of pixels 14, 15 by tap row (lane = fy) for the tail columns: read back from the table's pieces BEFORE the table is
            // published -- once the MFMA wave has taken its fragments the h-side wave refills the table
            if (!WS_EXP_NOSTAGE) {
              const int f5 = fyl & 31;
              const char* tp = tabv + (4 * (fyl >> 5) + ((f5 & 15) >> 2)) * 256 + ((f5 & 3) + 4 * ((f5 >> 4) & 1)) * 2;
              unsigned short r14[3], r15[3];
#pragma unroll
              for (int pc = 0; pc < 3; ++pc) {
                r14[pc] = *reinterpret_cast<const unsigned short*>(tp + pc * XTABP + 14 * 16);
                r15[pc] = *reinterpret_cast<const unsigned short*>(tp + pc * XTABP + 15 * 16);
              }
              asm volatile("" ::: "memory");
              ws_set(fl, F_TAB_FULL + p, 2 * n + 2);
#pragma unroll
              for (int pc = 0; pc < 3; ++pc) {
                v14 += __uint_as_float((unsigned)r14[pc] << 16);
                v15 += __uint_as_float((unsigned)r15[pc] << 16);
              }
            } else {
              ws_set(fl, F_TAB_FULL + p, 2 * n + 2);
            }
            __builtin_amdgcn_s_setprio(0);
            WS_T(2);
          }
          // (2) the gH tile of unit n - 1 -> HBM
          if (n > 0) ws_wait(fl, F_OUT_FULL + p, 2 * n);
          WS_T(6);
#pragma unroll
          for (int qq = 0; qq < (WS_EXP_NOSTAGE ? 0 : 4); ++qq) {
            const int fx = fq + 16 * qq;
            f32x4 v4 = *reinterpret_cast<const f32x4*>(tileh + (fx + 15) * WPV + 4 * pq);
            // window columns 64, 65 (pixel 14 tap 50; pixel 15 taps 49, 50) are the VALU tail sums
            if (pq == 3 && fx == 50) { v4[2] = s6414; v4[3] = s6515; }
            if (pq == 3 && fx == 49) v4[3] = s6415;
            x6_bstore4(v4, ghdst, (qq < 3 || fq < 3) ? qoff_prev : X_OOR, (unsigned)(16 * qq) * (umaj_st ? 64u : plane_b));
          }
          if (n > 0) ws_set(fl, F_OUT_FREE + p, 2 * n);
          WS_T(7);
          // (3) tail columns of gH (q = 64, 65: taps 50 / 49, 50 of pixels 14, 15): lane = tap row fy
          if (live && u == 0 && q >= 2) ws_wait(fl, F_SLIDE, 8 * (2 * q - 3));
          float a64[XC], a65[XC];
#pragma unroll
          for (int c = 0; c < XC; ++c) {
            const f32x2 sv = *reinterpret_cast<const f32x2*>(side + (c * XWIN + tslot) * 4 + 2 * wc);
            a64[c] = sv.x; a65[c] = sv.y;
          }
          asm volatile("" ::: "memory");
          if (live && u == 1) ws_set(fl, F_PROG + w, q + 1);
          WS_T(8);
          if (!WS_EXP_NOSTAGE) {
            const float lv = lane < XK ? 1.f : 0.f;
            v14 *= lv; v15 *= lv;
            float sa = 0.f, sb = 0.f, sc = 0.f;
#pragma unroll
            for (int c = 0; c < XC; ++c) {
              sa = fmaf(g14[c] * v14, a64[c], sa);
              sb = fmaf(g15[c] * v15, a64[c], sb);
              sc = fmaf(g15[c] * v15, a65[c], sc);
            }
            {                                              // three sums as one packed reduction (ws_wave_sum6): rows sa, sc, sb, 0
              const float xs[6] = {sa, sb, sc, 0.f, 0.f, 0.f};
              float qa, qb;
              ws_wave_sum6(xs, qa, qb);
              s6414 = rdlane(qa, 0); s6415 = rdlane(qa, 32); s6515 = rdlane(qa, 16);
            }
          }
          asm volatile("" ::: "memory");
          WS_T(9);
          if constexpr (!DMA) if (!WS_EXP_NOSTAGE) load_taps(vreg, vsrc, b, x0, y1, v_t0);
          WS_T(3);
        }
        qoff_prev = qoff;
        if constexpr (!DMA) if (!WS_EXP_NOSTAGE) {
          const unsigned go = pix_off(bi, x0, y1, XC);
#pragma unroll
          for (int c = 0; c < XC; ++c) gp[c] = x6_bload(gsrc, go, (unsigned)c * plane_b);
        }
        // (4) the new window row (slot of row 4 q - 4 + 2 u + {0 | 1}: behind every wave once phase q - 1 is done)
        if (live) {
          if (q >= 1) ws_wait_all_prog(fl, q);
          WS_T(4);
          const int slot = grow & (XWIN - 1);
          if constexpr (DMA) {
            const float* const rs_ = reinterpret_cast<const float*>(ring + (n & 1) * WRINGB + WRING_ROW) + lane;
            gr0 = rs_[0]; gr1 = rs_[64];
          }
          if constexpr (U8) {
            const unsigned k1 = x6_cvt_pk(rintf(gr0 * 255.f), rintf(gr1 * 255.f));
            if (gcol < 8 * XNBLK && !WS_EXP_NOSTAGE) {
              char* d0 = smem + gcell + slot * 16 + gc0 * XPLANE;
              x6_st16(d0, k1);
              if (ggrp == 0) x6_st16(d0 + XPLANE, k1 >> 16);
            }
          } else {
            unsigned h1, h2, h3;
            x6_split2(gr0, gr1, h1, h2, h3);
            if (gcol < 8 * XNBLK && !WS_EXP_NOSTAGE) {
              char* d0 = smem + gcell + slot * 16 + gc0 * XPLANE;
              x6_st16(d0, h1); x6_st16(d0 + 3 * XPLANE, h2); x6_st16(d0 + 6 * XPLANE, h3);
              if (ggrp == 0) { x6_st16(d0 + XPLANE, h1 >> 16); x6_st16(d0 + 4 * XPLANE, h2 >> 16); x6_st16(d0 + 7 * XPLANE, h3 >> 16); }
            }
          }
          if (gsidx >= 0) {
            side[(gc0 * XWIN + slot) * 4 + gsidx] = gr0;
            side[(gc1 * XWIN + slot) * 4 + gsidx] = gr1;
          }
          asm volatile("" ::: "memory");
          if (lane == 0) __hip_atomic_fetch_add(fl + F_SLIDE, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          asm volatile("" ::: "memory");
          WS_T(5);
        }
        if constexpr (DMA) {
          // LAST in the iteration (the count of the wait above): slot n & 1 -- its reads returned long ago -- takes unit n + 2
          asm volatile("" ::: "memory");
          __builtin_amdgcn_s_waitcnt(WS_LGKMCNT0);
          dma_unit(hside ? h : v, tap_bytes, frame, b, bi, x0, R0, wr0, min(n + 2, N - 1), n & 1);
          asm volatile("" ::: "memory");
          WS_T(10);
        }
      }
      if (WS_TRACE && blockIdx.x == 0 && lane == 0)
        for (int k = 0; k < 16; ++k) WS_TRACE_ADD(w, k, tr_[k]);
    }
    g = run_end;
    if (WS_TRACE) WS_TRACE_WG(1, 1);
  }
  };
  if (!staging) run_all(std::false_type{});
  else run_all(std::true_type{});
  if (WS_TRACE && blockIdx.x == 0 && (threadIdx.x & 63) == 0) WS_TRACE_ADD(threadIdx.x >> 6, 15, __builtin_readcyclecounter() - t_kernel0);
  if (WS_TRACE) WS_TRACE_WG(0, __builtin_readcyclecounter() - t_kernel0);
}

// ------------------------------------------------------------------------------------------------------------------------------------
// Forward on the same machinery:  out[b,c,y,x] = sum_fy v[b,fy,y,x] * T_c[fy],   T_c[fy][j] = sum_i In_c[y + fy][16 wc + i] * Hb[i][j].
// Replaces the reference's forward kernel (sepconv/sepconv_op/sepconv.py:5-30).
//   MFMA wave       T on the window with the h band (the gV product without the cotangent: 120 MFMAs per 16 pixels, tap rows 48..50 of the
//                   three channels packed into one tile), then the vertical pass on its accumulators: lane (j, kg) holds rows 16 m + 4 kg + r
//                   and takes v for exactly those rows from the pair's v tile (4 x 16 bytes); its three channel sums go to the pair's
//                   partial-sum tile, one row per lane group kg
//   h-side wave     h taps -> band table; the tail columns i = 64, 65 as six dot products over the tap rows (lane = tap row: window
//                   columns 64, 65 from the side copies, v of pixels 14, 15 by one 8-byte load per lane); adds the four lane groups'
//                   partial sums and the tails and stores the 3 x 16 results; window row 60 + 2 n
//   v-side wave     v taps (pairs of neighbouring taps per lane) -> v tile [pixel][tap] as 8-byte stores; window row 61 + 2 n
// LDS: window 103 680 B + 4 x (table 6 144 + v tile 4 608 + partial sums 768) + side columns + flags = 153 KB.
// ------------------------------------------------------------------------------------------------------------------------------------
constexpr int FVP = 72;                             // pitch of a v tile row (floats): [pixel j][tap]; 16-byte fragments of the 16 lanes of a
                                                    // read group fall on 8 j + 4 kg: conflict free
constexpr int FVTB = 16 * FVP * 4, FOPB = 4 * XC * 16 * 4 + 32;   // partial sums [kg][c][16] + the six tail-column sums of the unit
constexpr int FPAIRB = XTAB + FVTB + FOPB;
constexpr int FSIDE_OFF = XWINB + 4 * FPAIRB;
constexpr int FFLAG_OFF = FSIDE_OFF + WSIDEB;
constexpr int FLDS = FFLAG_OFF + 256;
static_assert(FLDS <= 160 * 1024, "LDS per CU");
enum { F_VT_FULL = 8, F_VT_FREE = 12, F_OP_FULL = 32, F_OP_FREE = 36, F_TL_FULL = 40, F_TL_FREE = 44 };     // + F_TAB_*, F_PROG, F_SLIDE, F_ERR

template <bool U8>
__global__ __launch_bounds__(WNT) void sepconv_fwd_ws(const float* __restrict__ in, const float* __restrict__ v,
                                                      const float* __restrict__ h, float* __restrict__ out,
                                                      int B, int Ho, int Wo, int nph, int ncol, int TB,
                                                      const unsigned* __restrict__ cls, int unit16, int spin_limit, unsigned* errw,
                                                      const float* __restrict__ in2, const unsigned* __restrict__ cls2, int pair) {
  // pair != 0 (savfi_sepconv_fwd_pair_frames8_f32; see sepconv_bwd_ws): B virtual samples b' = 2 b + f -- frame f of sample b from `in`
  // (f = 0) or `in2` (f = 1), taps at a stride of TB = 2 K planes, result b' of `out` ([B / 2][2][C][Ho][Wo]: the caller adds the two)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int p = w & 3, wc = p & 1, wr0 = p >> 1;
  const int role = w >> 2;                          // 0: MFMA wave, 1: h-side staging wave, 2: v-side staging wave
  const int j = lane & 15, kg = lane >> 4;
  char* const tab = smem + XWINB + p * FPAIRB;
  float* const vt = reinterpret_cast<float*>(tab + XTAB);
  float* const op = reinterpret_cast<float*>(tab + XTAB + FVTB);
  float* const tl = op + 4 * XC * 16;              // [pixel 14 | 15][channel]: the unit's tail-column sums
  float* const side = reinterpret_cast<float*>(smem + FSIDE_OFF);
  unsigned* const fl = reinterpret_cast<unsigned*>(smem + FFLAG_OFF);

  [[maybe_unused]] const unsigned long long t_kernel0 = WS_TRACE ? __builtin_readcyclecounter() : 0ull;
  // a workgroup's phases: positions [g0, g1) of the strip-major order (ws_work_range)
  int g0, g1;
  constexpr int base = 0;
  const int span = nph;
  ws_work_range(B * ncol, nph, g0, g1);
  if (g0 >= g1) return;
  if (!ws_frames8_mine<U8>(cls, pair ? cls2 : nullptr)) return;
  const int Hi = Ho + XK - 1, Wi = Wo + XK - 1;
  const unsigned plane_b = (unsigned)Ho * (unsigned)Wo * 4u;
  const __amdgpu_buffer_rsrc_t hsrc = x6_rsrc(h, (unsigned)((B - 1) * TB + XK) * plane_b);
  const __amdgpu_buffer_rsrc_t vsrc = x6_rsrc(v, (unsigned)((B - 1) * TB + XK) * plane_b);
  const unsigned in_bytes = (unsigned)((pair ? B >> 1 : B) * XC) * (unsigned)(Hi * Wi) * 4u;
  const __amdgpu_buffer_rsrc_t odst = x6_rsrc(out, (unsigned)(B * XC) * plane_b);

  auto pix_off = [&](int b, int x0, int y, int ch) {
    return (unsigned)b * (unsigned)ch * plane_b + (unsigned)(min(y, Ho - 1) * Wo + min(x0 + 16 * wc + j, Wo - 1)) * 4u;
  };
  // unit16: v and h are laid out [y][x / 16][tap][16] per sample (see sepconv_bwd_ws)
  const bool umaj = (unit16 & 1) != 0;
  const unsigned tap_stride = umaj ? 64u : plane_b;
  auto unit_off = [&](int b, int x0, int y) {
    return (unsigned)b * (unsigned)TB * plane_b + (unsigned)((min(y, Ho - 1) * (Wo >> 4) + min((x0 >> 4) + wc, (Wo >> 4) - 1)) * XK) * 64u;
  };
  auto load_taps = [&](float (&regs)[XNP][2], __amdgpu_buffer_rsrc_t src, int b, int x0, int y, int t0) {
    const unsigned pix = umaj ? unit_off(b, x0, y) + (unsigned)j * 4u : pix_off(b, x0, y, TB);
    const unsigned voff = pix + (unsigned)(t0 + 1) * tap_stride;
    regs[0][0] = x6_bload(src, pix + (unsigned)max(t0, 0) * tap_stride, 0u);
    regs[0][1] = x6_bload(src, voff, 0u);
#pragma unroll
    for (int a = 1; a < XNP - 1; ++a)
#pragma unroll
      for (int e = 0; e < 2; ++e) regs[a][e] = x6_bload(src, voff, (unsigned)(8 * a + e - 1) * tap_stride);
#pragma unroll
    for (int e = 0; e < 2; ++e) regs[XNP - 1][e] = x6_bload(src, pix + (unsigned)min(8 * (XNP - 1) + t0 + e, XK - 1) * tap_stride, 0u);
  };
  auto tap_or_zero = [&](const float (&regs)[XNP][2], int a, int e, int t0) {
    if (a == 0 && e == 0) return t0 < 0 ? 0.f : regs[0][0];
    if (a == XNP - 1) return (8 * (XNP - 1) + t0 + e < XK) ? regs[a][e] : 0.f;
    return regs[a][e];
  };
  const int h_t0 = 2 * kg - (j & 1), v_t0 = 2 * kg;
  auto write_h_table = [&](const float (&regs)[XNP][2]) {
#pragma unroll
    for (int k = 0; k < XTAB / 1024; ++k) *reinterpret_cast<u32x4*>(tab + (k * 64 + lane) * 16) = (u32x4){0u, 0u, 0u, 0u};
    const int base2 = 2 * kg + (j & ~1);
    char* const lb = tab + (base2 >> 3) * 256 + j * 16 + (base2 & 7) * 2;
#pragma unroll
    for (int a = 0; a < XNP; ++a) {
      unsigned h1, h2, h3;
      x6_split2(tap_or_zero(regs, a, 0, h_t0), tap_or_zero(regs, a, 1, h_t0), h1, h2, h3);
      char* d = lb + a * 256;
      if (a < XNP - 1 || base2 < 16) {
        *reinterpret_cast<unsigned*>(d) = h1;
        *reinterpret_cast<unsigned*>(d + XTABP) = h2;
        *reinterpret_cast<unsigned*>(d + 2 * XTABP) = h3;
      }
    }
  };
  auto rdlane = [&](float x, int l) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x), l)); };
  constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};
  constexpr int PB8[3] = {2, 1, 0};                  // U8: see sepconv_bwd_ws
  constexpr int NPC = U8 ? 1 : 3, NQ = U8 ? 3 : 6;
  constexpr int DEPTH = U8 ? 2 : 1, NSL = DEPTH + 1;

  auto run_all = [&](auto stg_) __attribute__((always_inline)) {       // one run loop per role (see sepconv_bwd_ws)
  constexpr bool STG = decltype(stg_)::value;
  int g = g0;
#pragma unroll 1
  while (g < g1) {
    const int s = g / span, gin = g - s * span, ph0 = base + gin, b = s / ncol, x0 = (s - b * ncol) * XMC;
    const int bi = pair ? b >> 1 : b;
    const __amdgpu_buffer_rsrc_t isrc = x6_rsrc((pair && (b & 1)) ? in2 : in, in_bytes);
    const int run_end = min(g1, g + (span - gin)), nrun = run_end - g, N = 2 * nrun;
    const int R0 = XPR * ph0;
    auto unit_y = [&](int n) { return R0 + XPR * (n >> 1) + wr0 + 2 * (n & 1); };
    __syncthreads();
    if (tid < 64) {
      const unsigned long long q = reinterpret_cast<unsigned long long>(errw);
      fl[tid] = tid == F_LIMIT ? (unsigned)spin_limit : tid == F_ERRW ? (unsigned)q : tid == F_ERRW + 1 ? (unsigned)(q >> 32) : 0u;
    }
    if (tid < XNT) {
#pragma unroll 1
      for (int r = 0; r < XWIN; r += 16) {
        X6Rows<16> sr;
        x6_rows_load<16>(sr, isrc, bi, x0, R0 + r, Hi, Wi, tid);
        x6_rows_write<16, U8>(sr, smem, R0 + r, tid, FSIDE_OFF);
      }
    }
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();

    if constexpr (!STG) {
      // =========================================== MFMA wave ===================================================================
      __builtin_amdgcn_s_setprio(WS_PRIO);
      unsigned long long tr_[16] = {0}, tlast_ = __builtin_readcyclecounter();
      bf16x8 bq[2][3], aq[NSL][2][NPC];
      int rowoff[4], rowoff_hi[4];                   // + 6 planes: the third bf16 piece within a DS immediate offset (see sepconv_bwd_ws)
      auto peek_raw = [&](int idx) { return __hip_atomic_load(fl + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); };
      auto set_rows = [&](int y) {
        const int lo_ = ws_lane(), jo = lo_ & 15, ko = lo_ >> 4;
        const int pk = ((ko & 1) << 1) | (ko >> 1);
#pragma unroll
        for (int m = 0; m < 3; ++m) rowoff[m] = ((y + 16 * m + jo) & (XWIN - 1)) * 16 + (2 * wc + pk) * XBLK;
        rowoff[3] = ((y + 48 + min(jo & 3, 2)) & (XWIN - 1)) * 16 + (2 * wc + pk) * XBLK + min(jo >> 2, 2) * XPLANE;
        if constexpr (!U8) {
#pragma unroll
          for (int m = 0; m < 4; ++m) { rowoff_hi[m] = rowoff[m] + 6 * XPLANE; asm volatile("" : "+v"(rowoff_hi[m])); }
        }
      };
      auto gv_acc = [](int uu, int t) { return uu < 6 ? 3 * (uu >> 1) + t : uu < 8 ? 3 * t + 2 : (t == 0 ? 8 : 9); };
      auto load_av = [&](int slot, int uu) {
        const int st = uu & 1;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          const int ai = gv_acc(uu, t), c = ai == 9 ? 0 : ai / 3, m = ai == 9 ? 3 : ai % 3;
#pragma unroll
          for (int pc = 0; pc < NPC; ++pc)
          {
            const int plane = pc * 3 + c;
            aq[slot][t][pc] = *reinterpret_cast<const bf16x8*>(smem + (plane >= 6 ? plane - 6 : plane) * XPLANE + 4 * st * XBLK + (plane >= 6 ? rowoff_hi[m] : rowoff[m]));
          }
        }
      };
      auto read_bh = [&]() {
        const int lo_ = ws_lane(), jo = lo_ & 15, ko = lo_ >> 4, pk = ((ko & 1) << 1) | (ko >> 1);
#pragma unroll
        for (int st = 0; st < 2; ++st)
#pragma unroll
          for (int pc = 0; pc < 3; ++pc) bq[st][pc] = *reinterpret_cast<const bf16x8*>(tab + pc * XTABP + (4 * st + pk) * 256 + jo * 16);
      };
      set_rows(unit_y(0));
      ws_wait(fl, F_TAB_FULL + p, 1);
      read_bh();
      ws_set(fl, F_TAB_FREE + p, 1);
#pragma unroll
      for (int d = 0; d < DEPTH; ++d) load_av(d, d);
#pragma unroll 1
      for (int n = 0; n < N; ++n) {
        const int q = n >> 1, u = n & 1;
        WS_T(0);
        f32x4 acc[10];                                 // [3 c + m] (m < 3), [9] = packed tile (rows 48 + r of channel kg)
#pragma unroll
        for (int i = 0; i < 10; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
        unsigned pre_tab = 0u, pre_vt = 0u, pre_op = 0u, pre_slide = 0u;
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int uu = 0; uu < 10; ++uu) {
          if (uu + DEPTH < 10) load_av((uu + DEPTH) % NSL, uu + DEPTH);
          if (uu == 9) { pre_tab = peek_raw(F_TAB_FULL + p); pre_vt = peek_raw(F_VT_FULL + p); pre_op = peek_raw(F_OP_FREE + p); pre_slide = peek_raw(F_SLIDE); }
          {
            const int st = uu & 1;
#pragma unroll
            for (int qq = 0; qq < NQ; ++qq)
#pragma unroll
              for (int t = 0; t < 2; ++t)
                acc[gv_acc(uu, t)] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(aq[uu % NSL][t][U8 ? 0 : PA[qq]], bq[st][U8 ? PB8[qq] : PB[qq]], acc[gv_acc(uu, t)], 0, 0, 0);
          }
          if (WS_INTERLEAVE && uu + DEPTH < 10) {
#pragma unroll
            for (int i = 0; i < 2 * NPC; ++i) {
              __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
              __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
            __builtin_amdgcn_sched_group_barrier(0x008, 2 * NQ - 2 * NPC, 0);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
        WS_T(1);
        if (u == 1) ws_set(fl, F_PROG + w, q + 1);             // this phase's window reads are over
        // the next unit's h fragments and first A fragments -- before this unit's vertical pass
        if (n + 1 < N) {
          const int q1 = (n + 1) >> 1;
          if ((int)__builtin_amdgcn_readfirstlane((int)pre_tab) < n + 2) ws_wait(fl, F_TAB_FULL + p, n + 2);
          read_bh();
          ws_set(fl, F_TAB_FREE + p, n + 2);
          if (u == 1 && q1 >= 2 && (int)__builtin_amdgcn_readfirstlane((int)pre_slide) < 8 * (2 * q1 - 3)) ws_wait(fl, F_SLIDE, 8 * (2 * q1 - 3));
          set_rows(unit_y(n + 1));
#pragma unroll
          for (int d = 0; d < DEPTH; ++d) load_av(d, d);
        }
        WS_T(2);
        // vertical pass: v of this lane's rows from the pair's v tile
        if ((int)__builtin_amdgcn_readfirstlane((int)pre_vt) < n + 1) ws_wait(fl, F_VT_FULL + p, n + 1);
        WS_T(3);
        const int lo_ = ws_lane(), jo = lo_ & 15, ko = lo_ >> 4;
        f32x4 vc[4];
#pragma unroll
        for (int m = 0; m < 3; ++m) vc[m] = *reinterpret_cast<const f32x4*>(vt + jo * FVP + 16 * m + 4 * ko);
        vc[3] = *reinterpret_cast<const f32x4*>(vt + jo * FVP + 48);
        ws_set(fl, F_VT_FREE + p, n + 1);
        float part[XC];
#pragma unroll
        for (int c = 0; c < XC; ++c) {
          float sum = 0.f;
#pragma unroll
          for (int m = 0; m < 3; ++m)
#pragma unroll
            for (int r = 0; r < 4; ++r) sum = fmaf(vc[m][r], acc[3 * c + m][r], sum);
          part[c] = sum;
        }
        {
          float t = vc[3][0] * acc[9][0];
          t = fmaf(vc[3][1], acc[9][1], t);
          t = fmaf(vc[3][2], acc[9][2], t);
#pragma unroll
          for (int c = 0; c < XC; ++c) part[c] += (ko == c) ? t : 0.f;
        }
        if constexpr (U8) {                            // the window holds 255 w
#pragma unroll
          for (int c = 0; c < XC; ++c) part[c] *= 1.0f / 255.0f;
        }
        WS_T(4);
        if ((int)__builtin_amdgcn_readfirstlane((int)pre_op) < n) ws_wait(fl, F_OP_FREE + p, n);
        WS_T(5);
#pragma unroll
        for (int c = 0; c < XC; ++c) op[(ko * XC + c) * 16 + jo] = part[c];
        ws_set(fl, F_OP_FULL + p, n + 1);
        WS_T(6);
      }
      __builtin_amdgcn_s_setprio(0);
      if (WS_TRACE && blockIdx.x == 0 && lane == 0)
        for (int k = 0; k < 16; ++k) WS_TRACE_ADD(w, k, tr_[k]);
    } else {
      // =========================================== staging waves ===============================================================
      const bool hside = role == 1;
      const int ptid = tid - (hside ? 256 : 512);
      const int gcol = ptid & 127, ggrp = ptid >> 7;
      const int gcell = (gcol >> 3) * XBLK + (gcol & 7) * 2;
      const int gsidx = gcol == 64 ? 0 : gcol == 65 ? 1 : gcol == 80 ? 2 : gcol == 81 ? 3 : -1;
      const unsigned gcolb = (unsigned)min(x0 + gcol, Wi - 1) * 4u;
      const int gc0 = ggrp == 0 ? 0 : 2, gc1 = ggrp == 0 ? 1 : 2;
      unsigned long long tr_[16] = {0}, tlast_ = __builtin_readcyclecounter();
      float hreg[XNP][2], vreg[XNP][2];
      if (hside) load_taps(hreg, hsrc, b, x0, unit_y(0), h_t0);
      else load_taps(vreg, vsrc, b, x0, unit_y(0), v_t0);
      unsigned ooff_prev = X_OOR, ooff_prev2 = X_OOR;
      // The h side hands out the result of unit n - 2 in iteration n, AFTER it has written the table of unit n (round 5).  One unit
      // behind, its iteration n waited for the MFMA wave's vertical pass of unit n - 1 -- which that wave starts only once it holds the
      // table of unit n -- and wrote the table of unit n + 1 after that wait, the output, the window row: the MFMA wave then waited
      // 2 900 of its 4 900 cycles per unit for the next table (profiles/r05_ws_trace_fwd_before.txt).  Two units behind nothing in the
      // h side's loop waits for the unit the MFMA wave is working on.
      const int NIT = N + (hside ? 1 : 0);
#pragma unroll 1
      for (int n = 0; n <= NIT; ++n) {
        const bool live = n < N;
        const int nn = min(n, N - 1);
        const int q = nn >> 1, u = nn & 1;
        const int y = unit_y(nn), y1 = unit_y(min(nn + 1, N - 1));
        float gr0, gr1;
        const int grow = R0 + 60 + 2 * nn + (hside ? 0 : 1);
        {
          const int rr = min(grow, Hi - 1);
          gr0 = x6_bload(isrc, (unsigned)(((bi * XC + gc0) * Hi + rr) * Wi) * 4u + gcolb, 0u);
          gr1 = x6_bload(isrc, (unsigned)(((bi * XC + gc1) * Hi + rr) * Wi) * 4u + gcolb, 0u);
        }
        const int fyl = min(lane, XK - 1);
        const int tslot = (y + fyl) & (XWIN - 1);
        WS_T(0);
        if (hside) {
          // (1) the h band of unit n takes the table
          if (live) {
            ws_wait(fl, F_TAB_FREE + p, n);
            WS_T(1);
            __builtin_amdgcn_s_setprio(WS_PRIO_TABLE);
            write_h_table(hreg);
            ws_set(fl, F_TAB_FULL + p, n + 1);
            __builtin_amdgcn_s_setprio(0);
            WS_T(2);
          }
          if (live && u == 1) ws_set(fl, F_PROG + w, q + 1);      // this wave reads neither the window nor the side columns
          load_taps(hreg, hsrc, b, x0, y1, h_t0);
          WS_T(3);
          // (2) unit n - 2: the four lane groups' partial sums + the tail-column sums of the v-side wave -> HBM (lane = (channel, pixel))
          {
            if (n > 1) { ws_wait(fl, F_OP_FULL + p, n - 1); ws_wait(fl, F_TL_FULL + p, n - 1); }
            WS_T(6);
            const int c = min(lane >> 4, XC - 1);
            const float* o = op + c * 16 + j;
            float val = (o[0] + o[XC * 16]) + (o[2 * XC * 16] + o[3 * XC * 16]);
            const float t14 = tl[c], t15 = tl[XC + c];
            val += j == 14 ? t14 : j == 15 ? t15 : 0.f;
            x6_bstore(val, odst, lane < 16 * XC ? ooff_prev2 : X_OOR, 0u);
            if (n > 1) { ws_set(fl, F_OP_FREE + p, n - 1); ws_set(fl, F_TL_FREE + p, n - 1); }
            WS_T(7);
          }
          {
            const int c = min(lane >> 4, XC - 1), x = x0 + 16 * wc + j;
            ooff_prev2 = ooff_prev;
            ooff_prev = (live && y < Ho && x < Wo) ? (unsigned)(b * XC + c) * plane_b + (unsigned)(y * Wo + x) * 4u : X_OOR;
          }
        } else {
          // taps 50 of pixel 14 and 49, 50 of pixel 15 of the h band (the tail columns' weights): lanes 0..2, in flight under the tile write
          const int hl = min(lane, 2);
          const unsigned hoff = umaj ? unit_off(b, x0, y) + (unsigned)(hl == 1 ? 49 : 50) * 64u + (unsigned)(hl == 0 ? 14 : 15) * 4u
                                       : (unsigned)b * (unsigned)TB * plane_b + (unsigned)(hl == 1 ? 49 : 50) * plane_b
                                             + (unsigned)(min(y, Ho - 1) * Wo + min(x0 + 16 * wc + (hl == 0 ? 14 : 15), Wo - 1)) * 4u;
          const float hraw = x6_bload(hsrc, hoff, 0u);
          // (1) v of unit n -> the pair's v tile [pixel][tap]: the lane's pairs of neighbouring taps as 8-byte stores
          float v14 = 0.f, v15 = 0.f;
          if (live) {
            ws_wait(fl, F_VT_FREE + p, n);
            WS_T(1);
            float* const vw = vt + j * FVP + 2 * kg;
#pragma unroll
            for (int a = 0; a < XNP; ++a)
              *reinterpret_cast<f32x2*>(vw + 8 * a) = (f32x2){tap_or_zero(vreg, a, 0, v_t0), tap_or_zero(vreg, a, 1, v_t0)};
            asm volatile("" ::: "memory");
            v14 = vt[14 * FVP + fyl];                 // v of pixels 14, 15 by tap row (lane = fy): back from the tile, before it is published
            v15 = vt[15 * FVP + fyl];
            asm volatile("" ::: "memory");
            ws_set(fl, F_VT_FULL + p, n + 1);
            WS_T(2);
          }
          load_taps(vreg, vsrc, b, x0, y1, v_t0);
          WS_T(3);
          // (3) tail columns of unit n: T_c[fy][14] += In_c[y + fy][64] h50_14, T_c[fy][15] += In_c[..][64] h49_15 + In_c[..][65] h50_15,
          //     times v of the pixel, summed over the tap rows -> six sums for the h-side wave's final add
          if (live && u == 0 && q >= 2) ws_wait(fl, F_SLIDE, 8 * (2 * q - 3));
          float a64[XC], a65[XC];
#pragma unroll
          for (int c = 0; c < XC; ++c) {
            const f32x2 sv = *reinterpret_cast<const f32x2*>(side + (c * XWIN + tslot) * 4 + 2 * wc);
            a64[c] = sv.x; a65[c] = sv.y;
          }
          asm volatile("" ::: "memory");
          if (live && u == 1) ws_set(fl, F_PROG + w, q + 1);
          WS_T(8);
          if (live) {
            const float h50_14 = rdlane(hraw, 0), h49_15 = rdlane(hraw, 1), h50_15 = rdlane(hraw, 2);
            const float lv = lane < XK ? 1.f : 0.f;
            v14 *= lv; v15 *= lv;
            float xs[6];                                   // tl[c] = sum over the tap rows of xs[c] (pixel 14), tl[3 + c] = of xs[3 + c] (pixel 15)
#pragma unroll
            for (int c = 0; c < XC; ++c) {
              xs[c] = v14 * (a64[c] * h50_14);
              xs[3 + c] = v15 * fmaf(a65[c], h50_15, a64[c] * h49_15);
            }
            float qa, qb;
            ws_wave_sum6(xs, qa, qb);                      // rows of qa: xs[0], xs[2], xs[1], xs[3]; rows 0 / 2 of qb: xs[4], xs[5]
            WS_T(9);
            ws_wait(fl, F_TL_FREE + p, n);
            WS_T(10);
            {
              const int row = lane >> 4;
              if ((lane & 15) == 0) tl[row == 0 ? 0 : row == 1 ? 2 : row == 2 ? 1 : 3] = qa;
              if (lane == 0 || lane == 32) tl[4 + (lane >> 5)] = qb;
            }
            ws_set(fl, F_TL_FULL + p, n + 1);
          }
        }
        // (4) the new window row
        WS_T(11);
        if (live) {
          if (q >= 1) ws_wait_all_prog(fl, q);
          WS_T(4);
          const int slot = grow & (XWIN - 1);
          if constexpr (U8) {
            const unsigned k1 = x6_cvt_pk(rintf(gr0 * 255.f), rintf(gr1 * 255.f));
            if (gcol < 8 * XNBLK) {
              char* d0 = smem + gcell + slot * 16 + gc0 * XPLANE;
              x6_st16(d0, k1);
              if (ggrp == 0) x6_st16(d0 + XPLANE, k1 >> 16);
            }
          } else {
            unsigned h1, h2, h3;
            x6_split2(gr0, gr1, h1, h2, h3);
            if (gcol < 8 * XNBLK) {
              char* d0 = smem + gcell + slot * 16 + gc0 * XPLANE;
              x6_st16(d0, h1); x6_st16(d0 + 3 * XPLANE, h2); x6_st16(d0 + 6 * XPLANE, h3);
              if (ggrp == 0) { x6_st16(d0 + XPLANE, h1 >> 16); x6_st16(d0 + 4 * XPLANE, h2 >> 16); x6_st16(d0 + 7 * XPLANE, h3 >> 16); }
            }
          }
          if (gsidx >= 0) {
            side[(gc0 * XWIN + slot) * 4 + gsidx] = gr0;
            side[(gc1 * XWIN + slot) * 4 + gsidx] = gr1;
          }
          asm volatile("" ::: "memory");
          if (lane == 0) __hip_atomic_fetch_add(fl + F_SLIDE, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          asm volatile("" ::: "memory");
          WS_T(5);
        }
      }
      if (WS_TRACE && blockIdx.x == 0 && lane == 0)
        for (int k = 0; k < 16; ++k) WS_TRACE_ADD(w, k, tr_[k]);
    }
    g = run_end;
    if (WS_TRACE) WS_TRACE_WG(1, 1);
  }
  };
  if (role == 0) run_all(std::false_type{});
  else run_all(std::true_type{});
  if (WS_TRACE && blockIdx.x == 0 && (threadIdx.x & 63) == 0) WS_TRACE_ADD(threadIdx.x >> 6, 15, __builtin_readcyclecounter() - t_kernel0);
  if (WS_TRACE) WS_TRACE_WG(0, __builtin_readcyclecounter() - t_kernel0);
}

}  // namespace

// gV and gH of the K = 51, C = 3 op, widths that are a multiple of 4; every tensor below 2^31 bytes (the caller checks).
// TB: tap planes between two samples of v / h / gV / gH (51 for contiguous [B,51,Ho,Wo] tensors; larger when the tensors are slices of one
// interleaved [B * S, 51, Ho, Wo] buffer: sepconv/model.py runs its four sub-networks as one task-batched launch per layer)
static int ws_spin_limit_host = 1 << 19;      // savfi_sepconv_ws_debug_spin_limit; handed to every launch
namespace {
constexpr int WS_MAX_DEV = 64;
unsigned* ws_watch_word[WS_MAX_DEV] = {nullptr};     // host addresses of the mapped words (savfi_sepconv_ws_watch), per device
unsigned* ws_watch_dev[WS_MAX_DEV] = {nullptr};      // their device addresses
// the word of the device this launch goes to (a process may drive several GPUs: a per-launch hipGetDevice, ~50 ns, not a global)
unsigned* ws_watch_device_word() {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= WS_MAX_DEV) return nullptr;
  return ws_watch_dev[dev];
}
}
// one workgroup per CU (LDS), fewer when there are fewer phases than CUs
static int ws_grid(int64_t total, int cus) { return (int)savfi_cdiv(total, savfi_cdiv(total, cus)); }

// cls: the words of savfi_frames8_classify_f32 on `in` (device memory; both instances of the kernel are launched and the device picks one),
// or nullptr (the six-product kernel only)
int savfi_sepconv_bwd_ws_launch(const float* in, const float* v, const float* h, const float* gO, float* gV, float* gH, int B, int Ho,
                                int Wo, int cus, int TB, const unsigned* cls, int taps_unit16, hipStream_t st, const float* in2,
                                const unsigned* cls2) {
  const int pair = in2 != nullptr ? 1 : 0;          // B virtual samples = two frames per sample (see the kernel)
  const int nph = savfi_cdiv(Ho, XPR), ncol = savfi_cdiv(Wo, XMC);
  const int64_t total = (int64_t)B * ncol * nph;
  const int grid = ws_grid(total, cus);
  if (taps_unit16 && (Wo & 15) != 0) return SAVFI_E_UNSUPPORTED;
  static uint32_t done = 0, done8 = 0;
  if (cls) {
    if (taps_unit16 & 1) {
      static uint32_t done8d = 0;
      if (int e = savfi_ensure_dynamic_lds((const void*)sepconv_bwd_ws<true, true>, WLDS, done8d)) return e;
      hipLaunchKernelGGL((sepconv_bwd_ws<true, true>), dim3(grid), dim3(WNT), WLDS, st, in, v, h, gO, gV, gH, B, Ho, Wo, nph, ncol, TB, cls, taps_unit16, ws_spin_limit_host, ws_watch_device_word(), in2, cls2, pair);
    } else {
      if (int e = savfi_ensure_dynamic_lds((const void*)sepconv_bwd_ws<true, false>, WLDS, done8)) return e;
      hipLaunchKernelGGL((sepconv_bwd_ws<true, false>), dim3(grid), dim3(WNT), WLDS, st, in, v, h, gO, gV, gH, B, Ho, Wo, nph, ncol, TB, cls, taps_unit16, ws_spin_limit_host, ws_watch_device_word(), in2, cls2, pair);
    }
  }
  if (int e = savfi_ensure_dynamic_lds((const void*)sepconv_bwd_ws<false, false>, WLDS, done)) return e;
  hipLaunchKernelGGL((sepconv_bwd_ws<false, false>), dim3(grid), dim3(WNT), WLDS, st, in, v, h, gO, gV, gH, B, Ho, Wo, nph, ncol, TB, cls, taps_unit16, ws_spin_limit_host, ws_watch_device_word(), in2, cls2, pair);
  return savfi_launch_status();
}

// forward of the same op, widths that are a multiple of 4 (declared in csrc/common.h)
int savfi_sepconv_fwd_ws_launch(const float* in, const float* v, const float* h, float* out, int B, int Ho, int Wo, int cus, int TB,
                                const unsigned* cls, int taps_unit16, hipStream_t st, const float* in2, const unsigned* cls2) {
  const int pair = in2 != nullptr ? 1 : 0;
  const int nph = savfi_cdiv(Ho, XPR), ncol = savfi_cdiv(Wo, XMC);
  const int64_t total = (int64_t)B * ncol * nph;
  const int grid = ws_grid(total, cus);
  if (taps_unit16 && (Wo & 15) != 0) return SAVFI_E_UNSUPPORTED;
  static uint32_t done = 0, done8 = 0;
  if (cls) {
    if (int e = savfi_ensure_dynamic_lds((const void*)sepconv_fwd_ws<true>, FLDS, done8)) return e;
    hipLaunchKernelGGL(sepconv_fwd_ws<true>, dim3(grid), dim3(WNT), FLDS, st, in, v, h, out, B, Ho, Wo, nph, ncol, TB, cls, taps_unit16, ws_spin_limit_host, ws_watch_device_word(), in2, cls2, pair);
  }
  if (int e = savfi_ensure_dynamic_lds((const void*)sepconv_fwd_ws<false>, FLDS, done)) return e;
  hipLaunchKernelGGL(sepconv_fwd_ws<false>, dim3(grid), dim3(WNT), FLDS, st, in, v, h, out, B, Ho, Wo, nph, ncol, TB, cls, taps_unit16, ws_spin_limit_host, ws_watch_device_word(), in2, cls2, pair);
  return savfi_launch_status();
}

// one word per classifier workgroup into cls[0 .. SAVFI_FRAMES8_WORDS): non-zero = an element of x[0 .. n) is not k / 255 (see ws_frames8_mine)
extern "C" int savfi_frames8_classify_f32(const float* x, int64_t n, unsigned* cls, void* stream) {
  if (!x || !cls) return SAVFI_E_NULL;
  if (n <= 0) return SAVFI_E_SHAPE;
  static_assert(CLS_WG == SAVFI_FRAMES8_WORDS, "include/savfi_hip.h");
  hipLaunchKernelGGL(frames8_classify, dim3(CLS_WG), dim3(CLS_NT), 0, (hipStream_t)stream, x, (long long)n, cls);
  return savfi_launch_status();
}

#if WS_TRACE
extern "C" int savfi_sepconv_ws_trace_wg(unsigned long long* out /* [1024][2] host */, int reset) {
  if (hipDeviceSynchronize() != hipSuccess) return -1;
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(ws_trace_wg), sizeof(unsigned long long) * 2048) != hipSuccess) return -1;
  if (reset) { static unsigned long long z[2048] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(ws_trace_wg), z, sizeof(z)) != hipSuccess) return -1; }
  return 0;
}
extern "C" int savfi_sepconv_ws_trace(unsigned long long* out /* [16][16] host */, int reset) {
  if (hipDeviceSynchronize() != hipSuccess) return -1;
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(ws_trace), sizeof(unsigned long long) * 256) != hipSuccess) return -1;
  if (reset) { unsigned long long z[256] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(ws_trace), z, sizeof(z)) != hipSuccess) return -1; }
  return 0;
}
#endif

// protocol time-outs since the library was loaded (0 unless a wait of the kernels above gave up: a bug, or a GPU shared with another
// process for longer than the spin limit)
extern "C" int savfi_sepconv_ws_errors(void) {
  unsigned n = 0;
  if (hipDeviceSynchronize() != hipSuccess) return -1;
  if (hipMemcpyFromSymbol(&n, HIP_SYMBOL(ws_error_count), sizeof(n)) != hipSuccess) return -1;
  return (int)(n & 0x7fffffffu);
}

// The same count without a device synchronisation: savfi_sepconv_ws_watch() (once per device, outside a stream capture) maps one host
// word into the device; a wait that gives up adds to it with a system-scope atomic, which the host sees at the latest when the launch
// has completed.  savfi_sepconv_ws_errors_peek() only reads that word: the product calls it wherever it has synchronised anyway (after
// reading the loss) and raises.
extern "C" int savfi_sepconv_ws_watch(void) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= WS_MAX_DEV) return -1;
  if (ws_watch_word[dev]) return SAVFI_OK;
  unsigned* hp = nullptr;
  if (hipHostMalloc((void**)&hp, 64, hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess) return -1;
  *hp = 0u;
  unsigned* dp = nullptr;
  if (hipHostGetDevicePointer((void**)&dp, hp, 0) != hipSuccess) return -1;
  ws_watch_word[dev] = hp;
  ws_watch_dev[dev] = dp;
  return SAVFI_OK;
}
extern "C" int savfi_sepconv_ws_errors_peek(void) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= WS_MAX_DEV || !ws_watch_word[dev]) return -1;
  return (int)(__atomic_load_n(ws_watch_word[dev], __ATOMIC_ACQUIRE) & 0x7fffffffu);
}
// Clears the current device's mapped word and returns the count it held (or -1: not armed).  A caller that has HANDLED a reported time-out
// (dropped the iteration, restored its state) continues from zero; without it every later check of the process would raise again.  The
// synchronising counter of savfi_sepconv_ws_errors() keeps counting (it is the process's total).
extern "C" int savfi_sepconv_ws_errors_reset(void) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= WS_MAX_DEV || !ws_watch_word[dev]) return -1;
  return (int)(__atomic_exchange_n(ws_watch_word[dev], 0u, __ATOMIC_ACQ_REL) & 0x7fffffffu);
}
// Test hook: the spin limit of the kernels' bounded waits (default 1 << 19 spins of s_sleep 2; a NEGATIVE limit makes every wait that does
// not find its flag at once give up -- how tests/ provoke the error path).  The previous limit goes to *previous (may be NULL).
extern "C" int savfi_sepconv_ws_debug_spin_limit(int limit, int* previous) {
  if (previous) *previous = ws_spin_limit_host;
  ws_spin_limit_host = limit;
  return SAVFI_OK;
}
