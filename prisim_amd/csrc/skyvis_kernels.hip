// skyvis_kernels.hip -- hand-written CDNA4 (gfx950) kernels for the PRISim per-baseline sky-sum.
//
// Reference path replaced: prisim/interferometry.py:6255-6376 (InterferometerArray.observe):
//   V[b,f] = sum_s pbflux[s,f] * w[s,b,f] * exp(-2 pi i f (tau[s,b] - taupc[b]))
//   tau = dc . bl^T / c               (prisim/baseline_delay_horizon.py:240)
//   w   = exp(-1/2 (u_perp/sigma)^2)  (prisim/interferometry.py:6265-6283)
//
// MI355X mapping:
//   * lanes = baselines (64 consecutive baselines per wavefront), so pbflux[s,f] and the source
//     direction are wave-uniform: they are fetched with scalar loads and used as SGPR operands
//     (no LDS, no barriers in the source loop); no cross-lane reduction is needed.
//   * each thread owns CT consecutive channels of one baseline: 2*CT accumulators in VGPRs.
//   * the nsrc x nbl x nchan phase matrix of the reference is never materialised: per (source,
//     baseline, channel tile) one range-reduced seed phasor and one step phasor are formed
//     (fp64 phase reduction), then the phasor is advanced along frequency by a complex rotation
//     (3-4 VALU) and accumulated (2 FMA).  The tile is seeded at its centre channel and walked in
//     both directions (two independent dependency chains, half the drift).
//   * rows of the packed pbflux ([tile][source][CT]) stream through the scalar cache, requested
//     one piece ahead of their use; an LDS-DMA touch 12 sources ahead keeps L2 warm for skies
//     whose slab outgrows it.
//   * block id -> (pbflux slab, baseline group) is XCD-aware: all blocks resident on one XCD
//     read the same pbflux slab, which therefore stays in that XCD's 4 MiB L2.
//   * fp32 mode accumulates in fp32 registers and adds them into the fp64 cube every flush_src
//     sources (transposed through LDS so the stores are coalesced), so the summation error
//     does not grow with nsrc.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "skyvis_kernels.h"

namespace prisim {


__device__ __forceinline__ float fma_(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
__device__ __forceinline__ double fma_(double a, double b, double c) { return __builtin_fma(a, b, c); }

// ------------------------------------------------------------------------------------------
// sincos of an angle given in QUARTER CYCLES (a4 = 4 * phase[cycles], any magnitude < 2^31):
// returns (cos, sin) of 2 pi * phase.  The quadrant q = rint(a4) and the residual
// y = (a4 - q)/4 in [-1/8, 1/8] cycle are formed in fp64 (exact), so no separate reduction modulo
// one cycle is needed: q mod 4 selects the quadrant.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ void sincos_qcycles(double a4, float& c, float& s) {
  const double q = __builtin_rint(a4);
  const float y = (float)((a4 - q) * 0.25);
  const int qi = (int)q;
  const float y2 = y * y;
  // sin(2 pi y) = y * (2pi + y^2 * (-(2pi)^3/3! + y^2 * ((2pi)^5/5! + y^2 * (-(2pi)^7/7! + y^2 * (2pi)^9/9!))))
  float ps = 42.058693944897655f;                          // (2pi)^9/9!
  ps = __builtin_fmaf(ps, y2, -76.70585975306136f);        // -(2pi)^7/7!
  ps = __builtin_fmaf(ps, y2, 81.60524927607504f);         // (2pi)^5/5!
  ps = __builtin_fmaf(ps, y2, -41.341702240399755f);       // -(2pi)^3/3!
  float sy = __builtin_fmaf(y * y2, ps, y * 6.2831855f);   // fl32(2pi) = 6.28318548
  sy = __builtin_fmaf(y, -1.7484555e-7f, sy);              // + y * (2pi - fl32(2pi))
  // cos(2 pi y) = 1 + y^2 * (-(2pi)^2/2 + y^2 * ((2pi)^4/4! + y^2 * (-(2pi)^6/6! + y^2 * ((2pi)^8/8! - y^2 (2pi)^10/10!))))
  float pc = -26.42625678337438f;                          // -(2pi)^10/10!
  pc = __builtin_fmaf(pc, y2, 60.24464137187666f);         // (2pi)^8/8!
  pc = __builtin_fmaf(pc, y2, -85.45681720669373f);        // -(2pi)^6/6!
  pc = __builtin_fmaf(pc, y2, 64.93939402266829f);         // (2pi)^4/4!
  pc = __builtin_fmaf(pc, y2, -19.739208802178716f);       // -(2pi)^2/2
  const float cy = __builtin_fmaf(pc, y2, 1.0f);
  // rotate by q quarter turns: q=0:(c,s) 1:(-s,c) 2:(-c,-s) 3:(s,-c)
  const bool swap = (qi & 1) != 0;
  const float cc = swap ? sy : cy;
  const float ss = swap ? cy : sy;
  c = (((qi + 1) & 2) != 0) ? -cc : cc;    // q mod 4 in {1,2}
  s = ((qi & 2) != 0) ? -ss : ss;          // q mod 4 in {2,3}
}

// fp64 (sin x, cos x) for |x| <= pi/4 (fdlibm __kernel_sin / __kernel_cos minimax coefficients), no reduction, no quadrant logic
__device__ __forceinline__ void sincos_kernel_f64(double x, double& sn, double& cs) {
  const double z = x * x;
  double ps = 1.58969099521155010221e-10;
  ps = __builtin_fma(ps, z, -2.50507602534068634195e-08);
  ps = __builtin_fma(ps, z, 2.75573137070700676789e-06);
  ps = __builtin_fma(ps, z, -1.98412698298579493134e-04);
  ps = __builtin_fma(ps, z, 8.33333333332248946124e-03);
  ps = __builtin_fma(ps, z, -1.66666666666666324348e-01);
  sn = __builtin_fma(x * z, ps, x);
  double pc = -1.13596475577881948265e-11;
  pc = __builtin_fma(pc, z, 2.08757232129817482790e-09);
  pc = __builtin_fma(pc, z, -2.75573143513906633035e-07);
  pc = __builtin_fma(pc, z, 2.48015872894767294178e-05);
  pc = __builtin_fma(pc, z, -1.38888888888741095749e-03);
  pc = __builtin_fma(pc, z, 4.16666666666666019037e-02);
  cs = __builtin_fma(z * z, pc, __builtin_fma(-0.5, z, 1.0));
}

// a / c for c in [0.7, 1]: v_rcp_f64 seed + two Newton steps + one residual correction (the generic fp64 division is ~14 instructions
// of scaling and fix-up that this range does not need)
__device__ __forceinline__ double div_unit_range_f64(double a, double c) {
  double r = __builtin_amdgcn_rcp(c);
  r = __builtin_fma(r, __builtin_fma(-c, r, 1.0), r);
  r = __builtin_fma(r, __builtin_fma(-c, r, 1.0), r);
  const double q = a * r;
  return __builtin_fma(r, __builtin_fma(-c, q, a), q);
}

// (cos, sin) of 2 pi a for a phase in CYCLES of any magnitude: fp64 reduction to [-1/2, 1/2] cycle, then the hardware
// v_cos_f32 / v_sin_f32 (input in revolutions; measured max abs error 1.25e-7 on [-1/2, 1/2], tools/microbench_trig.hip).
// Two quarter-rate instructions instead of ~25 for the polynomial + quadrant logic.  Only for phasors that are used once
// (the seed at the tile centre); the step phasor, whose error is multiplied by the chain length, keeps the polynomial.
__device__ __forceinline__ void sincos_cycles_hw(double a, float& c, float& s) {
  const float y = (float)(a - __builtin_rint(a));
  c = __builtin_amdgcn_cosf(y);
  s = __builtin_amdgcn_sinf(y);
}

// sin(2 pi y) for |y| <= 1/8 cycle: y * (2pi_hi + (2pi_lo + y^2 P(y^2))), P = degree-2 minimax fit (1.8e-9 abs), so the
// result is within ~1 ulp in 6 instructions.  This is the lifting step's s = sin(alpha): its error is multiplied by the chain length.
__device__ __forceinline__ float sin_2pi_y(float y) {
  const float y2 = y * y;
  float ps = -75.36964416503906f;
  ps = __builtin_fmaf(ps, y2, 81.59197998046875f);
  ps = __builtin_fmaf(ps, y2, -41.3416633605957f);
  const float q = __builtin_fmaf(ps, y2, -1.7484555e-7f);               // + (2pi - fl32(2pi))
  return __builtin_fmaf(y, q, y * 6.2831855f);
}

// tan(pi y) for |y| <= 1/8 cycle, ~1 ulp: y * (pi_hi + (pi_lo + y^2 P(y^2))), P = degree-3 fit of (tan(pi y)/y - pi)/y^2
__device__ __forceinline__ float tan_pi_y(float y) {
  const float u = y * y;
  float pt = 769.7825317382812f;
  pt = __builtin_fmaf(pt, u, 161.2586212158203f);
  pt = __builtin_fmaf(pt, u, 40.81293869018555f);
  pt = __builtin_fmaf(pt, u, 10.335405349731445f);
  const float q = __builtin_fmaf(pt, u, -8.742278e-8f);                 // + (pi - fl32(pi))
  return __builtin_fmaf(y, q, y * 3.1415927410125732f);
}

// cos(2 pi y) for |y| <= 1/8 cycle (the cosine polynomial of sincos_qcycles without the quadrant logic)
__device__ __forceinline__ float cos_2pi_y(float y) {
  const float y2 = y * y;
  float pc = -26.42625678337438f;
  pc = __builtin_fmaf(pc, y2, 60.24464137187666f);
  pc = __builtin_fmaf(pc, y2, -85.45681720669373f);
  pc = __builtin_fmaf(pc, y2, 64.93939402266829f);
  pc = __builtin_fmaf(pc, y2, -19.739208802178716f);
  return __builtin_fmaf(pc, y2, 1.0f);
}

__device__ __forceinline__ void sincos_qcycles(double a4, double& c, double& s) {
  // fp64: residual angle t = 2 pi y, |t| <= pi/4; fdlibm __kernel_sin/__kernel_cos minimax coefficients
  const double q = __builtin_rint(a4);
  const double t = (a4 - q) * 1.5707963267948966192;       // (a4-q)/4 * 2pi
  const int qi = (int)q;
  const double z = t * t;
  double ps = 1.58969099521155010221e-10;
  ps = __builtin_fma(ps, z, -2.50507602534068634195e-08);
  ps = __builtin_fma(ps, z, 2.75573137070700676789e-06);
  ps = __builtin_fma(ps, z, -1.98412698298579493134e-04);
  ps = __builtin_fma(ps, z, 8.33333333332248946124e-03);
  ps = __builtin_fma(ps, z, -1.66666666666666324348e-01);
  const double sy = __builtin_fma(t * z, ps, t);
  double pc = -1.13596475577881948265e-11;
  pc = __builtin_fma(pc, z, 2.08757232129817482790e-09);
  pc = __builtin_fma(pc, z, -2.75573143513906633035e-07);
  pc = __builtin_fma(pc, z, 2.48015872894767294178e-05);
  pc = __builtin_fma(pc, z, -1.38888888888741095749e-03);
  pc = __builtin_fma(pc, z, 4.16666666666666019037e-02);
  const double cy = __builtin_fma(z * z, pc, __builtin_fma(-0.5, z, 1.0));
  const bool swap = (qi & 1) != 0;
  const double cc = swap ? sy : cy;
  const double ss = swap ? cy : sy;
  c = (((qi + 1) & 2) != 0) ? -cc : cc;
  s = ((qi & 2) != 0) ? -ss : ss;
}


// fp64 (cos, sin) of 2 pi a from a table of kTabN unit phasors in LDS: the caller passes aN = a * kTabN (any magnitude below 2^31);
// k = rint(aN) picks the table entry (k mod kTabN), the residual angle |t| <= pi / kTabN = 3.1e-3 rad needs only
// sin t = t - t^3/6 (next term 2.3e-15) and cos t = 1 - t^2/2 + t^4/24 (next term 1.2e-18).  16 VALU instructions and one
// ds_read_b128 instead of ~30 for the two minimax polynomials with quadrant logic: the fp64 kernels issue v_fma_f64 back to back
// (SQ busy ~ 100 % of the fp64 rate, profiles/r02a_fp64), so every seed instruction saved is throughput.
constexpr int kTabN = 1024;
struct TabPhase { double r; double2 e; };      // residual in table steps (|r| <= 1/2) and the table phasor of one lookup
__device__ __forceinline__ TabPhase sincos_tab_front(double aN, const double2* tab) {
  // k = rint(aN) through the 1.5 * 2^52 shift: the low mantissa word of the sum is k modulo 2^32, so no conversion is needed
  const double shifted = aN + 6755399441055744.0;
  const double k = shifted - 6755399441055744.0;
  TabPhase r;
  r.r = aN - k;
  r.e = tab[__double2loint(shifted) & (kTabN - 1)];
  return r;
}
__device__ __forceinline__ void sincos_tab_back(const TabPhase& q, double& c, double& s) {
  // t = w r with w = 2 pi / kTabN folded into the coefficients: sin t = r (w - w^3/6 r^2), cos t = 1 - w^2/2 r^2 + w^4/24 r^4
  constexpr double w = 6.283185307179586476925 / kTabN;
  const double z = q.r * q.r;
  const double sl = q.r * __builtin_fma(z, -(w * w * w) / 6.0, w);
  const double cl = __builtin_fma(z, __builtin_fma(z, (w * w * w * w) / 24.0, -0.5 * w * w), 1.0);
  c = __builtin_fma(q.e.x, cl, -(q.e.y * sl));
  s = __builtin_fma(q.e.y, cl, q.e.x * sl);
}

// each thread of the block fills kTabN / kBlockThreads entries (ocml sincospi: exact at the octant points); the caller barriers
__device__ __forceinline__ void fill_phasor_table(double2* tab) {
  for (int i = threadIdx.x; i < kTabN; i += kBlockThreads) {
    double sn, cs;
    sincospi((double)(2 * i) / (double)kTabN, &sn, &cs);
    tab[i] = make_double2(cs, sn);
  }
}

// a / c for c in [0.7, 1] to ~1e-15 relative: v_rcp_f64 seed (~2^-26) and one Newton step; the quotient is not corrected further
// (the fp64 tolerance is 1e-11 of sum|pbflux|, and an error of tan(alpha/2) only shifts the step angle by ~alpha * 1e-15)
__device__ __forceinline__ double div_unit_range_fast_f64(double a, double c) {
  double r = __builtin_amdgcn_rcp(c);
  r = __builtin_fma(r, __builtin_fma(-c, r, 1.0), r);
  return a * r;
}

// exp(x) for x <= 0 down to the underflow threshold from a table of 2^(i / kExpTabN) in LDS, split like sincos_tab so that the LDS read
// can be issued a piece ahead of its use: k = rint(x kExpTabN / ln2) by the 1.5 * 2^52 shift (its low mantissa word is k in two's
// complement), residual r = x - k ln2 / kExpTabN in two pieces (|k| < 2^21, the high piece carries 32 bits: the product is exact),
// |r| <= 3.4e-4 so exp(r) = 1 + r + r^2/2 + r^3/6 + r^4/24 (next term 4e-20).  12 fp64 instructions against ~40 for the library call.
constexpr int kExpTabN = 1024;
struct ExpPhase { double r, t; int e; };
__device__ __forceinline__ ExpPhase exp_tab_front(double x, const double* etab) {
  x = __builtin_fmax(x, -800.0);
  const double shifted = __builtin_fma(x, 1477.3197218702985, 6755399441055744.0);
  const double k = shifted - 6755399441055744.0;
  const int lo = __double2loint(shifted);
  ExpPhase f;
  f.r = __builtin_fma(k, 4.1024561256651217e-14, __builtin_fma(k, -0x1.62e42ff000000p-11, x));
  f.t = etab[lo & (kExpTabN - 1)];
  f.e = lo >> 10;
  return f;
}
__device__ __forceinline__ double exp_tab_back(const ExpPhase& f) {
  double q = __builtin_fma(f.r, 4.16666666666666666667e-02, 1.66666666666666666667e-01);
  q = __builtin_fma(q, f.r, 0.5);
  q = __builtin_fma(q, f.r, 1.0);
  q = __builtin_fma(q, f.r, 1.0);
  return __builtin_ldexp(f.t * q, f.e);
}
__device__ __forceinline__ void fill_exp_table(double* etab) {
  for (int i = threadIdx.x; i < kExpTabN; i += kBlockThreads) etab[i] = exp2((double)i / (double)kExpTabN);
}

// exp(x) for |x| <= 0.1: Taylor series to x^9 (next term 2.8e-17)
__device__ __forceinline__ double exp_series9(double x) {
  double q = 2.75573192239858906526e-06;                                       // 1/9!
  q = __builtin_fma(q, x, 2.48015873015873015873e-05);
  q = __builtin_fma(q, x, 1.98412698412698412698e-04);
  q = __builtin_fma(q, x, 1.38888888888888888889e-03);
  q = __builtin_fma(q, x, 8.33333333333333333333e-03);
  q = __builtin_fma(q, x, 4.16666666666666666667e-02);
  q = __builtin_fma(q, x, 1.66666666666666666667e-01);
  q = __builtin_fma(q, x, 0.5);
  q = __builtin_fma(q, x, 1.0);
  return __builtin_fma(q, x, 1.0);
}

// fp64 source-shape taper along frequency as a second-order multiplicative recurrence of the Gaussian w = exp(-g f^2):
//   w_{k+1} = w_k q_k,  q_{k+1} = q_k h,   h = exp(-2 g df^2);   w_{k-1} = w_k q'_k, q'_{k-1} = q'_k h
// Returns the weights of channels HC (wu) and HC-1 (wd) of the tile centred on fc and the ratios that advance them.
__device__ __forceinline__ void taper_seed_f64(double gq, double fc, double df, double& wu, double& wd, double& qu, double& qd, double& h) {
  const double x1 = -2.0 * gq * fc * df, x2 = -gq * df * df;
  double e1, e1inv, e2;
  if (__builtin_amdgcn_ballot_w64(!(__builtin_fabs(x1) < 0.1 && __builtin_fabs(x2) < 1e-3)) == 0) {
    // the usual case (df << f): exp(+-x1) from the even and odd series to x^9 (next term 3e-17 at |x1| = 0.1) and
    // exp(x2) from five terms, instead of two library exponentials and an fp64 division -- wave-uniform choice
    const double z = x1 * x1;
    double ce = 2.48015873015873015873e-05;                                    // 1/8!
    ce = __builtin_fma(ce, z, 1.38888888888888888889e-03);
    ce = __builtin_fma(ce, z, 4.16666666666666666667e-02);
    ce = __builtin_fma(ce, z, 0.5);
    ce = __builtin_fma(ce, z, 1.0);
    double so = 2.75573192239858906526e-06;                                    // 1/9!
    so = __builtin_fma(so, z, 1.98412698412698412698e-04);
    so = __builtin_fma(so, z, 8.33333333333333333333e-03);
    so = __builtin_fma(so, z, 1.66666666666666666667e-01);
    so = __builtin_fma(so, z, 1.0) * x1;
    e1 = ce + so;
    e1inv = ce - so;
    double p2 = 4.16666666666666666667e-02;
    p2 = __builtin_fma(p2, x2, 1.66666666666666666667e-01);
    p2 = __builtin_fma(p2, x2, 0.5);
    p2 = __builtin_fma(p2, x2, 1.0);
    e2 = __builtin_fma(p2, x2, 1.0);
  } else {
    e1 = exp(x1);
    e1inv = 1.0 / e1;
    e2 = exp(x2);
  }
  h = e2 * e2;
  wu = exp(-gq * fc * fc);                  // channel HC
  qu = e1 * e2;                             // w_{HC+1}/w_{HC}
  qd = e2 * e1inv;                          // w_{HC-1}/w_{HC}
  wd = wu * qd;                             // channel HC-1
  qd *= h;                                  // w_{HC-2}/w_{HC-1}
}

// ------------------------------------------------------------------------------------------
// Recurrence kernel
// ------------------------------------------------------------------------------------------
// waves per SIMD the register allocator is asked to leave room for (hipcc otherwise spends
// up to 256 VGPRs on scheduling freedom and drops to 1-2 waves/SIMD, too few to keep the
// 2-cycle fp32 VALU issue slots filled -- see tools/microbench_valu.hip results).
template <typename T, int CT, bool TAPER> struct WavesPerEU {
  // fp64: 56 KiB of LDS per block (phasor table, flush buffer, prefetch area) admit 2 blocks per CU whatever the tile
  static constexpr int value = (sizeof(T) == 4) ? (CT <= 32 ? 4 : 2) : 2;
};

// Wave-local ordering point between LDS writes and LDS reads of other lanes of the same wave (DS operations of one wave
// execute in order; this only stops the compiler from moving them across and makes it wait for the counters).
__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Rows of the per-wave transpose buffer used by the flush: a thread owns CT channels of ONE baseline, so storing straight
// from registers writes 16-byte pieces 16 KiB apart (measured 0.5 TB/s: 2 ms of the 59 ms launch at config 3, 10 % of a
// 1/8 baseline shard).  Instead each wave transposes kFlushCh channels at a time through LDS (unused otherwise) and writes
// runs of 128 contiguous bytes per 8 lanes.
template <typename T> struct FlushCfg;
template <> struct FlushCfg<float> { static constexpr int ch = 16; using vec = float2; };     // 64 x 17 x 8 B = 8.5 KiB per wave
template <> struct FlushCfg<double> { static constexpr int ch = 8; using vec = double2; };    // 64 x 9 x 16 B = 9 KiB per wave
template <typename T> constexpr int flush_lds_bytes() { return (kBlockThreads / 64) * 64 * (FlushCfg<T>::ch + 1) * (int)sizeof(typename FlushCfg<T>::vec); }

// L2 warm-up for the scalar row stream.  A scalar load is requested one piece (~300 cycles) ahead of its use, which covers the
// scalar cache and L2 but not HBM: when a tile's pbflux slab (nsrc x 256 B) outgrows the 4 MiB L2 (config 5: 100 MB) every row
// would stall the wave.  So every 4th source each wave also touches 1 KiB of rows (4 rows of 256 B) and 8 directions
// kPrefetchAhead sources ahead with LDS-DMA loads (no VGPR, never waited for) that land in a dummy area of LDS behind the flush buffer.
constexpr int kPrefetchAhead = 12;
constexpr int kPrefetchWaveBytes = 1024;
constexpr int kPrefetchLdsBytes = (kBlockThreads / 64) * kPrefetchWaveBytes;
typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;


// XCD-aware, load-balanced block -> (slab, baseline group) map.  Blocks are dealt round-robin to the 8 XCDs (blockIdx % 8 shares an
// XCD).  The work items (slab, baseline group), slab-major, are cut into 8 equal contiguous ranges, one per XCD: the blocks resident
// on one XCD walk the baseline groups of one slab before the next, so that slab's pbflux rows stay in the XCD's 4 MiB L2, and every
// XCD gets the same number of blocks whatever the slab count (pinning whole slabs to XCDs left 2 of 8 XCDs with 2 instead of 3
// tiles at 22 tiles: 9 % of the launch).  Returns false for the padding blocks past the last item.
__device__ __forceinline__ bool block_item(const SkyvisParams& p, int& slab, int& bg) {
  const int total = p.ntiles * p.nsplit * p.nbgroups;
  const int per_xcd = (total + 7) >> 3;
  const int item = (int)(blockIdx.x & 7) * per_xcd + (int)(blockIdx.x >> 3);
  slab = item / p.nbgroups;
  bg = item - slab * p.nbgroups;
  return (int)(blockIdx.x >> 3) < per_xcd && item < total;
}

// LIFT: lifting (three-shear) form of the step rotation, see skyvis_rec_f32pk_body below; chosen per baseline group by the host.
// TAPER: 0 = none; 1 = exact second-order amplitude recurrence on the pbflux operand (8 instructions per term);
//   2 (fp64 only, k_skyvis_taper_f64) = the GROUPED form of the packed fp32 kernels carried over to fp64, made exact: ONE chain from the
//   tile's first channel, zeta_j = w_j z_j advanced by a complex factor rho_t = r q_t that is held over a group of 8 steps at the group's
//   geometric-mean amplitude ratio q_t = (w_{8t+8} / w_{8t})^(1/8) -- exact at the group ends, low by exp(nu m (8 - m)), nu = g df^2, at
//   step m inside, and that factor (four per-lane values X^7, X^12, X^15, X^16 formed once per (source, baseline, tile)) is put back on
//   the wave-uniform pbflux operand: 7 instructions per term (1 correction multiply, 2 accumulate FMAs, 4 for zeta * rho) and
//   rho_{t+1} = rho_t exp(-16 nu) per group.  Nothing is truncated: every factor is formed to fp64 accuracy (short series, or the library
//   exp on a wave-uniform slow path for steps that are not small).  Rows are packed in natural channel order, one group = one 64-byte piece.
// WITEM (k_skyvis_taper_f64_wave, arrays of at most 256 baselines): the unit of work is a WAVEFRONT, not a block -- item g = 4 bg + wave
//   is (baseline wave g mod nbw, source split g / nbw), nbw = ceil(nbl / 64), p.wave_nsplit splits.  A 171-baseline array (config 2)
//   fills 3 of a block's 4 wavefronts: with block items one SIMD of every CU idles and the 4th wave costs nothing when added (256
//   baselines take the time of 171, tools/config2_fullwave_probe.py); with wave items all four SIMDs carry sources.
// WITEM = 2 (k_skyvis_taper_f64_wave_batch): wave items over a BATCH of snapshots -- item g is (snapshot g / (nbw nsplit), split, baseline
//   wave); every snapshot has its own source range in the concatenated packed rows / prepared directions, its own phase centre and its
//   own output (cube slot or partial cubes), read from the wave-uniform table p.wave_snaps.  A whole observing run of a small array
//   (HERA-19: 3 baseline waves) fills the chip in ONE launch instead of one 50 us launch per snapshot.
template <typename T, int CT, int TAPER, bool LIFT, int WITEM = 0>
__device__ __forceinline__ void skyvis_rec_body(const SkyvisParams& p, unsigned char* flush_lds, unsigned char* pf_area, const double2* tab,
                                                const double* etab = nullptr) {
  static_assert(CT % 8 == 0, "channel tile must be a multiple of 8");
  static_assert(WITEM == 0 || TAPER == 2, "wave items: the grouped fp64 taper kernel only");
  static_assert(TAPER != 2 || (sizeof(T) == 8 && CT >= 16 && !LIFT), "the grouped fp64 taper: 16- or 32-channel tiles, folded (no lifting)");
  constexpr bool GROUPED = TAPER == 2;
  constexpr int HC = CT / 2;                       // channels per chain (here the taper multiplies pbflux, so z stays a pure rotation
                                                   // and the lifting form applies with or without it)
  // The row is fetched in NPART pieces of at most 64 bytes (16 SGPRs) through two SGPR buffers: with half rows (2 x 32 SGPRs at
  // CT = 32 fp64) the buffers did not fit beside the kernel's other wave-uniform state and the compiler moved 103-134 SGPRs
  // through v_readlane / v_writelane around the source loop.
  constexpr int NPART = (CT * (int)sizeof(T) / 64) > 2 ? (CT * (int)sizeof(T) / 64) : 2;
  constexpr int NH = CT / NPART;                   // elements per piece
  static_assert(NPART % 2 == 0 && NH % 2 == 0, "pieces hold whole (up, down) pairs and alternate between two buffers");
  typedef const __attribute__((address_space(4))) T* crow_p;
  typedef const volatile __attribute__((address_space(4))) double* cdir_p;    // volatile: see the packed kernel (keeps the load where it is written)
  typedef const __attribute__((address_space(4))) float* cfsq_p;

  int slab, bg;
  if (!block_item(p, slab, bg)) return;
  const int tile = slab % p.ntiles;
  int split = slab / p.ntiles;
  int bwave = 0;                                    // WITEM: this wave's 64 baselines start at 64 bwave
  bool item_ok = true;
  // the source range, phase centre and output of this item's snapshot (WITEM = 2: from the batch table; else the launch's own)
  int64_t src_lo = p.src_lo, src_hi = p.src_hi, src_per_split = p.src_per_split;
  double pc_x = p.pc_x, pc_y = p.pc_y, pc_z = p.pc_z;
  double* out_base = p.out;
  if constexpr (WITEM != 0) {
    const int g = bg * (kBlockThreads / 64) + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    if constexpr (WITEM == 2) {
      const int per_snap = p.wave_nbw * p.wave_nsplit;
      int snap = g / per_snap;
      const int r = g - snap * per_snap;
      split = r / p.wave_nbw;
      bwave = r - split * p.wave_nbw;
      item_ok = snap < p.wave_nsnap;
      snap = __builtin_amdgcn_readfirstlane(item_ok ? snap : 0);
      const BatchSnap* sn = p.wave_snaps + snap;
      src_lo = sn->row0; src_hi = sn->row0 + sn->nsrc; src_per_split = sn->src_per_split;
      pc_x = sn->pc[0]; pc_y = sn->pc[1]; pc_z = sn->pc[2];
      out_base = sn->out;
    } else {
      split = g / p.wave_nbw;                       // (block_item sees p.nsplit = 1 and p.nbgroups = quads of items)
      bwave = g - split * p.wave_nbw;
      item_ok = split < p.wave_nsplit;
    }
  }
  const int nsp = WITEM != 0 ? p.wave_nsplit : p.nsplit;
  const int cg = WITEM != 0 ? 0 : bg;               // baseline group of the culling table (WITEM: one group)

  int64_t s_begin = (int64_t)split * p.src_per_split;
  int64_t s_end = s_begin + p.src_per_split;
  if (s_end > p.nsrc) s_end = p.nsrc;
  // (TAPER == 1 bodies: no source ranges, no taper culling -- a variable source range cost k_skyvis_rec<double,32,true> 47 more SGPR
  // spills, 33 of them as lane moves inside its source loop; the grouped fp64 kernel has both, the fp64 body without the taper takes a
  // source range and accumulates: it sums the point-source runs of a mixed sky)
  constexpr bool RANGED = GROUPED || (sizeof(T) == 8 && TAPER == 0);
  if constexpr (RANGED && !GROUPED) {
    s_begin = src_lo + (int64_t)split * src_per_split;
    s_end = s_begin + src_per_split;
    if (s_end > src_hi) s_end = src_hi;
  }
  if constexpr (GROUPED) {
    // sources [src_lo, src_hi) of the sky (a run of one source size when the host walks the sky run by run), cut into nsplit pieces;
    // taper culling as in the packed fp32 kernels: the group's leading sources are provably below the tolerance (capi.cpp) and what is
    // left is cut into nsplit equal pieces again
    s_begin = src_lo + (int64_t)split * src_per_split;
    s_end = s_begin + src_per_split;
    if (s_end > src_hi) s_end = src_hi;
    if (WITEM != 2 && p.src_first != nullptr) {
      const int64_t f = p.src_first[cg];
      if (f > src_lo) {
        const int64_t per = (src_hi - f + nsp - 1) / nsp;
        s_begin = f + (int64_t)split * per;
        s_end = s_begin + per < src_hi ? s_begin + per : src_hi;
      }
    }
  }

  const int tid = threadIdx.x;
  const int64_t b_raw = WITEM != 0 ? (int64_t)bwave * 64 + (tid & 63) : (int64_t)bg * kBlockThreads + tid;
  const bool b_valid = b_raw < p.nbl;
  const int64_t b = b_valid ? b_raw : (p.nbl - 1);
  // a wavefront whose first baseline is out of range does no arithmetic (wave-uniform)
  const bool wave_active = WITEM != 0 ? item_ok : ((int64_t)bg * kBlockThreads + (tid & ~63)) < p.nbl;

  const double bx = p.bl_x[b], by = p.bl_y[b], bz = p.bl_z[b];
  const int k0 = tile * CT;
  const double fc = p.f0 + (double)(k0 + (GROUPED ? 0 : HC)) * p.df;   // frequency of the seed channel (the tile's centre; GROUPED: its first)
  const double df = p.df;
  const double fc4 = 4.0 * fc, df4 = 4.0 * df;         // quarter-cycle scaling for sincos_qcycles
  const double fcN = fc * kTabN, dfN = df * kTabN, dfN_half = df * (0.5 * kTabN);   // table scaling for sincos_tab (fp64)

  // taper per-lane constants
  double bl2_c2 = 0.0, bpc = 0.0;
  if (TAPER) {
    bl2_c2 = (bx * bx + by * by + bz * bz) * (p.inv_c * p.inv_c);
    bpc = (bx * pc_x + by * pc_y + bz * pc_z) * p.inv_c;   // tau_pc: un-offset delay = d + bpc
  }

  T acc_re[CT], acc_im[CT];
#pragma unroll
  for (int k = 0; k < CT; ++k) { acc_re[k] = (T)0; acc_im[k] = (T)0; }

  // pbflux rows and directions are wave-uniform: scalar loads into SGPRs (see k_skyvis_rec_f32pk for the pipeline)
  const crow_p gp = (crow_p)(uintptr_t)(reinterpret_cast<const T*>(p.pb_packed) + (size_t)tile * (size_t)p.nsrc_pad * CT);
  const cdir_p gd = (cdir_p)(uintptr_t)p.dirs_prep;
  const cfsq_p gfq = (cfsq_p)(uintptr_t)(p.fsq_pairs ? p.fsq_pairs + (size_t)tile * CT : nullptr);   // fp32 taper only

  double* const out = out_base + ((size_t)split * p.nbl * p.nchan) * 2;   // partial buffer of this split
  bool first_flush = !RANGED || p.accumulate == 0;     // accumulate: an earlier launch (another source run) already wrote this slot

  using FV = typename FlushCfg<T>::vec;
  constexpr int FCH = FlushCfg<T>::ch < CT ? FlushCfg<T>::ch : CT;
  FV* const wbuf = reinterpret_cast<FV*>(flush_lds) + (tid >> 6) * (64 * (FCH + 1));
  const int lane = tid & 63;
  const int64_t bw0 = WITEM != 0 ? (int64_t)bwave * 64 : (int64_t)bg * kBlockThreads + (tid & ~63);      // first baseline of this wave

  auto flush = [&]() {
    if (wave_active) {
      int lane_o = lane;
      asm volatile("" : "+v"(lane_o));        // opaque: keeps the 64 store addresses from being hoisted out of the segment loop and spilled
#pragma unroll
      for (int pz = 0; pz < CT / FCH; ++pz) {
#pragma unroll
        for (int c = 0; c < FCH; ++c) {
          FV v; v.x = acc_re[pz * FCH + c]; v.y = acc_im[pz * FCH + c];
          wbuf[lane * (FCH + 1) + c] = v;
          acc_re[pz * FCH + c] = (T)0; acc_im[pz * FCH + c] = (T)0;     // dead during the stores below
        }
        wave_lds_sync();
        const int c = lane_o % FCH;
        const int k = k0 + pz * FCH + c;
#pragma unroll
        for (int r = 0; r < FCH; ++r) {                          // 64 / FCH baselines per store instruction
          const int bi = r * (64 / FCH) + lane_o / FCH;
          const FV a = wbuf[bi * (FCH + 1) + c];
          const int64_t bb = bw0 + bi;
          if (bb < p.nbl && k < p.nchan) {
            if (sizeof(T) == 4 && p.out_f32) {                     // complex64 partial of a source split: written once, no read-modify-write
              reinterpret_cast<float2*>(p.out)[((size_t)split * p.nbl + (size_t)bb) * p.nchan + k] = make_float2((float)a.x, (float)a.y);
            } else {
              double2* o = reinterpret_cast<double2*>(out) + (size_t)bb * p.nchan + k;
              double2 v = make_double2((double)a.x, (double)a.y);
              if (!first_flush) { const double2 old = *o; v.x += old.x; v.y += old.y; }
              *o = v;
            }
          }
          __builtin_amdgcn_sched_barrier(0);                       // one store at a time: the flush must fit beside 128 accumulators
        }
        wave_lds_sync();
      }
    }
#pragma unroll
    for (int k = 0; k < CT; ++k) { acc_re[k] = (T)0; acc_im[k] = (T)0; }
    first_flush = false;
  };

  // Sources are walked in segments of flush_src (fp32: the partial sums of a segment are then added into the fp64 cube;
  // fp64: one segment).  One flush site, 32-bit source indices (the host keeps nsrc below 2^31).
  const int n_loc = s_end > s_begin ? (int)(s_end - s_begin) : 0;
  const int seg_len = (sizeof(T) == 4 && p.flush_src > 0) ? p.flush_src : 0x7fffffff;
  const crow_p gps = gp + (size_t)s_begin * CT;
  const cdir_p gds = gd + (size_t)s_begin * 4;
  const float* const pf_rows = reinterpret_cast<const float*>(reinterpret_cast<const T*>(p.pb_packed) + (size_t)tile * (size_t)p.nsrc_pad * CT +
                                                              (size_t)s_begin * CT);
  const float* const pf_dirs = reinterpret_cast<const float*>(p.dirs_prep) + (size_t)s_begin * 8;
  const lptr_t pf_lds = (lptr_t)(pf_area + (tid >> 6) * kPrefetchWaveBytes);
  constexpr int kRowDwords = CT * (int)sizeof(T) / 4;                    // <= 96
  constexpr int kReqRows = 256 / kRowDwords > 8 ? 256 / kRowDwords : 8;  // sources one warm-up request covers (1 KiB of rows, 8 directions)
  const bool pf_on = n_loc >= 2 * kReqRows;          // requests are clamped kReqRows before the end of the wave's own range
  int seg0 = 0;
  do {
  const int seg1 = (n_loc - seg0 > seg_len) ? seg0 + seg_len : n_loc;
  if (wave_active && seg1 > seg0) {
    T ra[NH], rb[NH];                                  // [2j] = channel HC+j (up), [2j+1] = channel HC-1-j (down)
    double sv[4] = {0.0, 0.0, 0.0, 0.0};
    // The seed of source s+1 is started one piece early (software pipeline): its delay d, the taper's kappa and -- fp64 -- the two
    // phasor-table reads are formed right after the wait of phase 1 of source s, so the LDS latency hides behind a piece of pair
    // arithmetic and no s_waitcnt ever has a table read and a fresh scalar request outstanding together (LDS and SMEM share
    // lgkmcnt and SMEM returns out of order: the only usable wait is lgkmcnt(0), which would otherwise expose a scalar-cache round
    // trip at the top of every source).  The directions are therefore requested TWO sources ahead.
    struct Pre { double d, kap; TabPhase pc, ps; ExpPhase ex; };
    auto front = [&](Pre& q) {
      q.d = __builtin_fma(bx, sv[0], __builtin_fma(by, sv[1], bz * sv[2]));   // seconds
      if constexpr (GROUPED) {
        // g = kappa_s (|b|^2/c^2 - tau^2), tau = d + b.s_pc/c the un-offset delay; the table read of w_0 = exp(-g f_0^2) goes out here
        const double tau = q.d + bpc;
        double gq = sv[3] * __builtin_fma(-tau, tau, bl2_c2);
        gq = __builtin_fmax(gq, 0.0);              // |b|^2 >= (b.s)^2 up to rounding
        q.kap = gq;
        q.ex = exp_tab_front(-gq * (fc * fc), etab);
      } else if constexpr (TAPER != 0) {
        double kap = sv[3];
        asm volatile("" : "+v"(kap));            // carried in a VGPR: two more live SGPR pairs tipped the taper bodies into lane spills
        q.kap = kap;
      } else {
        q.kap = 0.0;
      }
      if constexpr (sizeof(T) == 8) {
        q.pc = sincos_tab_front(q.d * fcN, tab);                                // phase at the centre channel
        q.ps = sincos_tab_front(q.d * (LIFT ? dfN_half : dfN), tab);            // step (LIFT: its half angle)
      }
    };
    Pre pre, pre_next;
    {
      const cdir_p d0 = gds + (size_t)seg0 * 4;
      sv[0] = d0[0]; sv[1] = d0[1]; sv[2] = d0[2];
      if (TAPER) sv[3] = d0[3];
      front(pre);                                                      // waits for the direction
      __builtin_amdgcn_sched_barrier(0);
      const crow_p r0 = gps + (size_t)seg0 * CT;
#pragma unroll
      for (int i = 0; i < NH; ++i) ra[i] = r0[i];
      const cdir_p d1 = gds + (size_t)((seg0 + 1 < seg1) ? seg0 + 1 : seg0) * 4;
      sv[0] = d1[0]; sv[1] = d1[1]; sv[2] = d1[2];
      if (TAPER) sv[3] = d1[3];
      __builtin_amdgcn_sched_barrier(0);
    }
    for (int s = seg0; s < seg1; ++s) {
      const crow_p row = gps + (size_t)s * CT;
      const int sn = (s + 1 < seg1) ? s + 1 : s;                    // the last source is simply fetched again
      const int sn2 = (s + 2 < seg1) ? s + 2 : seg1 - 1;
      if (pf_on && ((s - seg0) & 3) == 0) {
        // L2 warm-up, every 4th source: 1 KiB of rows and 8 directions kPrefetchAhead sources ahead (see k_skyvis_rec_f32pk)
        const int spf = (s + kPrefetchAhead < n_loc - kReqRows) ? s + kPrefetchAhead : n_loc - kReqRows;
        int lane_pf = lane;
        asm volatile("" : "+v"(lane_pf));                            // formed here from the lane id: no per-lane pointers kept live across the loop
        __builtin_amdgcn_global_load_lds((gptr_t)(pf_rows + (size_t)spf * kRowDwords + lane_pf * 4), pf_lds, 16, 0, 0);
        __builtin_amdgcn_global_load_lds((gptr_t)(pf_dirs + (size_t)spf * 8 + lane_pf), pf_lds, 4, 0, 0);
      }
      // top of source s: the first piece of its row, the direction of s+1 and the table reads of its own seed are in flight
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_waitcnt(0xC07F);            // lgkmcnt(0)
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < NH; ++i) rb[i] = row[NH + i];
      __builtin_amdgcn_sched_barrier(0);
      const double d = pre.d;

      T zc, zs, rc = (T)1, rs;
      const double th4 = d * df4;
      T tl = (T)0;                         // tan(alpha/2), alpha = -2 pi theta the step angle (LIFT groups only)
      if constexpr (sizeof(T) == 8) {
        sincos_tab_back(pre.pc, zc, zs);
        if constexpr (LIFT) {
          // |theta| <= 1/4 cycle is guaranteed for this baseline group: half angle beta = pi theta in [-pi/4, pi/4] from the table,
          // tan(beta) by a short reciprocal (cos beta >= 0.7), sin of the step by doubling; the lifting form never needs cos alpha
          double sb, cb;
          sincos_tab_back(pre.ps, cb, sb);
          rs = 2.0 * sb * cb;                                                       // sin 2 beta
          tl = -div_unit_range_fast_f64(sb, cb);                                    // alpha = -2 beta
        } else {
          sincos_tab_back(pre.ps, rc, rs);                       // phase step per channel
        }
      } else {
        sincos_qcycles(d * fc4, zc, zs);
        sincos_qcycles(th4, rc, rs);
      }
      // exp(-2 pi i phi): z = (cos, -sin)
      T ur = zc, ui = -zs;            // up chain: channel HC + j
      const T rr = rc, ri = -rs;      // step forward; step backward is conj(r)
      if constexpr (LIFT && sizeof(T) == 4) tl = -tan_pi_y((float)(0.25 * th4));      // |theta| <= 1/8 cycle guaranteed
      T dr, di;                            // z * conj(r): channel HC-1
      if constexpr (GROUPED) {
        dr = rr; di = ri;                  // (set below: the step factor rho)
      } else if constexpr (LIFT) {
        // one inverse lifting step (t -> -t, s -> -s)
        const T xd = fma_(tl, ui, ur);
        di = fma_(-ri, xd, ui);
        dr = fma_(tl, di, xd);
      } else {
        dr = fma_(ur, rr, ui * ri);
        di = fma_(ui, rr, -(ur * ri));
      }
      // source-shape taper  w = exp(-g f^2),  g = kappa_s * (|b|^2/c^2 - tau^2),  tau = d + b.s_pc/c
      double gq = 0.0;
      float g2 = 0.f;
      double wu = 1.0, wd = 1.0, qu = 1.0, qd = 1.0, h = 1.0;
      double cm[4] = {1.0, 1.0, 1.0, 1.0}, hg = 1.0;      // GROUPED: X^7, X^12, X^15, X^16 (X = exp(g df^2)) and exp(-16 g df^2)
      if constexpr (GROUPED) {
        // ln w_j = -g (f_0 + j df)^2.  Group t (steps 8t .. 8t+8) holds the amplitude ratio q_t = exp(-a - nu (16 t + 8)), a = 2 g df f_0,
        // nu = g df^2; zeta_0 = w_0 z_0, rho_0 = q_0 r, rho_{t+1} = rho_t exp(-16 nu); step m of a group is low by X^(m (8 - m)).
        const double gq2 = pre.kap;
        const double a = gq2 * (2.0 * df * fc);
        const double u = gq2 * (df * df);
        double E, y8;
        if (__builtin_amdgcn_ballot_w64(!(__builtin_fabs(a) < 4.0e-3 && u < 1.0e-6)) == 0) {
          // HERA-size exponents (config 3 / 5: |a| <= 1.3e-3, u <= 4.2e-7): exp(-a) to a^5 (next term 5.7e-18), X to u^2 (1.7e-19),
          // exp(-8u) to (8u)^3 with the factor folded into the coefficients (1.7e-22): 7 instructions fewer than the general short series
          const double na = -a;
          double q5 = __builtin_fma(na, 8.33333333333333333333e-03, 4.16666666666666666667e-02);
          q5 = __builtin_fma(q5, na, 1.66666666666666666667e-01);
          q5 = __builtin_fma(q5, na, 0.5);
          q5 = __builtin_fma(q5, na, 1.0);
          E = __builtin_fma(q5, na, 1.0);
          const double X = __builtin_fma(u, __builtin_fma(u, 0.5, 1.0), 1.0);
          const double X2 = X * X, X3 = X2 * X, X5 = X3 * X2;
          cm[0] = X5 * X2;
          cm[1] = cm[0] * X5;
          cm[2] = cm[1] * X3;
          cm[3] = cm[2] * X;
          y8 = __builtin_fma(u, __builtin_fma(u, __builtin_fma(u, -8.53333333333333333333e+01, 32.0), -8.0), 1.0);
        } else if (__builtin_amdgcn_ballot_w64(!(__builtin_fabs(a) < 0.1 && u < 3.0e-5)) == 0) {
          // the usual case (df << f): all by short series -- exp(-a) to a^9 (next term 2.8e-17), X = exp(u) to u^3 (next 3e-20), the
          // powers by 7 multiplications, exp(-8 u) to (8u)^4 (next 2.7e-20)
          E = exp_series9(-a);
          const double X = __builtin_fma(u, __builtin_fma(u, __builtin_fma(u, 1.66666666666666666667e-01, 0.5), 1.0), 1.0);
          const double X2 = X * X, X3 = X2 * X, X5 = X3 * X2;
          cm[0] = X5 * X2;
          cm[1] = cm[0] * X5;
          cm[2] = cm[1] * X3;
          cm[3] = cm[2] * X;
          const double v = -8.0 * u;
          y8 = __builtin_fma(v, __builtin_fma(v, __builtin_fma(v, __builtin_fma(v, 4.16666666666666666667e-02, 1.66666666666666666667e-01), 0.5), 1.0), 1.0);
        } else {
          // wave-uniform slow path: coarse channel grids or very long baselines over large sources
          E = exp(-a);
          cm[0] = exp(7.0 * u); cm[1] = exp(12.0 * u); cm[2] = exp(15.0 * u); cm[3] = exp(16.0 * u);
          y8 = exp(-8.0 * u);
        }
        hg = y8 * y8;
        const double w0 = exp_tab_back(pre.ex);
        const double q0 = E * y8;
        ur = w0 * zc; ui = -(w0 * zs);                 // zeta_0
        dr = q0 * rr; di = q0 * ri;                    // rho_0 (the down-chain registers carry the step factor in this form)
      } else if constexpr (TAPER != 0) {
        const double tau = d + bpc;
        gq = pre.kap * (bl2_c2 - tau * tau);
        gq = gq > 0.0 ? gq : 0.0;      // |b|^2 >= (b.s)^2 up to rounding
        if constexpr (sizeof(T) == 4) {
          g2 = -(float)(gq * p.fsq_scale);   // fp32: direct exp2 per term (no error accumulation)
        } else {
          taper_seed_f64(gq, fc, df, wu, wd, qu, qd, h);
        }
      }

      auto pairs = [&](const T (&r)[NH], int jbase) {
#pragma unroll
        for (int jj = 0; jj < NH / 2; ++jj) {
          const int j = jbase + jj;
          const int ku = HC + j, kd = HC - 1 - j;
          T pu = r[2 * jj], pd = r[2 * jj + 1];
          if constexpr (TAPER != 0) {
            if constexpr (sizeof(T) == 4) {
              pu *= __builtin_amdgcn_exp2f(g2 * gfq[2 * j]);
              pd *= __builtin_amdgcn_exp2f(g2 * gfq[2 * j + 1]);
            } else {
              pu *= wu; pd *= wd;
              wu *= qu; qu *= h; wd *= qd; qd *= h;
            }
          }
          acc_re[ku] = fma_(pu, ur, acc_re[ku]);
          acc_im[ku] = fma_(pu, ui, acc_im[ku]);
          acc_re[kd] = fma_(pd, dr, acc_re[kd]);
          acc_im[kd] = fma_(pd, di, acc_im[kd]);
          if constexpr (LIFT) {
            // x1 = x - t y, y1 = y + s x1, x2 = x1 - t y1 with t = tan(alpha/2), s = sin(alpha) = ri (up), -t, -s (down)
            const T xu = fma_(-tl, ui, ur);
            const T yu = fma_(ri, xu, ui);
            ur = fma_(-tl, yu, xu); ui = yu;
            const T xd = fma_(tl, di, dr);
            const T yd = fma_(-ri, xd, di);
            dr = fma_(tl, yd, xd); di = yd;
          } else {
            const T nur = fma_(ur, rr, -(ui * ri));
            const T nui = fma_(ur, ri, ui * rr);
            const T ndr = fma_(dr, rr, di * ri);
            const T ndi = fma_(di, rr, -(dr * ri));
            ur = nur; ui = nui; dr = ndr; di = ndi;
          }
        }
      };
      // GROUPED: one piece = one group of 8 channels in natural order (7 instructions per term)
      auto group = [&](const T (&r)[NH], int t) {
        if constexpr (GROUPED) {
          constexpr int kmap[8] = {0, 0, 1, 2, 3, 2, 1, 0};            // m (8 - m) = 7, 12, 15, 16, 15, 12, 7 for m = 1..7
#pragma unroll
          for (int m = 0; m < 8; ++m) {
            const int k = 8 * t + m;
            T pc = r[m];
            if (m != 0) pc *= (T)cm[kmap[m]];
            acc_re[k] = fma_(pc, ur, acc_re[k]);
            acc_im[k] = fma_(pc, ui, acc_im[k]);
            if (!(t == NPART - 1 && m == 7)) {
              const T t0 = ui * di, t1 = ur * di;
              const T nr = fma_(ur, dr, -t0);
              const T ni = fma_(ui, dr, t1);
              ur = nr; ui = ni;
            }
          }
          if (t + 1 < NPART) { dr *= (T)hg; di *= (T)hg; }
        }
      };
      if constexpr (GROUPED) group(ra, 0);
      else pairs(ra, 0);
#pragma unroll
      for (int ph = 1; ph < NPART; ++ph) {
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_waitcnt(0xC07F);          // lgkmcnt(0): piece ph has landed before the next request goes out
        __builtin_amdgcn_sched_barrier(0);
        if (ph == 1) {
          front(pre_next);                           // direction of s+1 (requested one source ago): d, kappa, table reads
          __builtin_amdgcn_sched_barrier(0);
        }
        if (ph + 1 < NPART) {
          // piece ph + 1 of this source into the buffer piece ph - 1 has just left
#pragma unroll
          for (int i = 0; i < NH; ++i) {
            if (ph & 1) ra[i] = row[(ph + 1) * NH + i];
            else rb[i] = row[(ph + 1) * NH + i];
          }
        } else {
          // first piece of the next source (NPART is even: it goes to ra) + the direction of the one after it
          const crow_p rn = gps + (size_t)sn * CT;
#pragma unroll
          for (int i = 0; i < NH; ++i) ra[i] = rn[i];
          const cdir_p dn = gds + (size_t)sn2 * 4;
          sv[0] = dn[0]; sv[1] = dn[1]; sv[2] = dn[2];
          if (TAPER) sv[3] = dn[3];
        }
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (GROUPED) {
          if (ph & 1) group(rb, ph);
          else group(ra, ph);
        } else {
          if (ph & 1) pairs(rb, ph * (NH / 2));
          else pairs(ra, ph * (NH / 2));
        }
      }
      pre = pre_next;
    }
  }
  flush();
  seg0 = seg1;
  } while (seg0 < n_loc);
}

template <typename T, int CT, bool TAPER>
__global__ __launch_bounds__(kBlockThreads) __attribute__((amdgpu_waves_per_eu(WavesPerEU<T, CT, TAPER>::value)))
void k_skyvis_rec(const SkyvisParams p) {
  // three separate LDS objects: the LDS-DMA prefetch writes (global_load_lds) are tracked by vmcnt, and the compiler makes every
  // LDS read that may alias their target wait for vmcnt(0) -- with one shared array the phasor-table reads of the fp64 seed
  // waited out an HBM round trip every 4th source
  __shared__ __attribute__((aligned(16))) unsigned char flush_lds[flush_lds_bytes<T>()];
  __shared__ __attribute__((aligned(16))) unsigned char pf_area[kPrefetchLdsBytes];
  __shared__ double2 tab_lds[sizeof(T) == 8 ? kTabN : 1];
  const double2* tab = nullptr;
  if constexpr (sizeof(T) == 8) {
    fill_phasor_table(tab_lds);
    __syncthreads();                                             // the only block barrier of the kernel: before any early exit
    tab = tab_lds;
  }
  int slab_, bg;
  const bool in_range = block_item(p, slab_, bg);                  // padding blocks read flag 0 and leave inside the body
  if (in_range && p.lift_flags != nullptr && p.lift_flags[bg] != 0) {   // block-uniform; the two bodies share no live state
    skyvis_rec_body<T, CT, TAPER ? 1 : 0, true>(p, flush_lds, pf_area, tab);
    return;
  }
  skyvis_rec_body<T, CT, TAPER ? 1 : 0, false>(p, flush_lds, pf_area, tab);
}

// fp64 sky-sum with the source-shape taper in the grouped form (TAPER = 2 bodies above): interferometry.py:6257-6283, 6332-6335 at the
// reference's default precision.  Sources [src_lo, src_hi), taper culling through src_first, accumulate for a later source run.
template <int CT>
__global__ __launch_bounds__(kBlockThreads) __attribute__((amdgpu_waves_per_eu(2)))
void k_skyvis_taper_f64(const SkyvisParams p) {
  __shared__ __attribute__((aligned(16))) unsigned char flush_lds[flush_lds_bytes<double>()];
  __shared__ __attribute__((aligned(16))) unsigned char pf_area[kPrefetchLdsBytes];
  __shared__ double2 tab_lds[kTabN];
  __shared__ double etab_lds[kExpTabN];
  fill_phasor_table(tab_lds);
  fill_exp_table(etab_lds);
  __syncthreads();                                               // the only block barrier of the kernel: before any early exit
  skyvis_rec_body<double, CT, 2, false>(p, flush_lds, pf_area, tab_lds, etab_lds);
}

// The same for arrays of at most 256 baselines whose sources are split: wave items (WITEM above).  p.nbgroups = ceil(nbw wave_nsplit / 4)
// quads of items, p.nsplit = 1, partial cube `split` of p.wave_nsplit.
template <int CT>
__global__ __launch_bounds__(kBlockThreads) __attribute__((amdgpu_waves_per_eu(2)))
void k_skyvis_taper_f64_wave(const SkyvisParams p) {
  __shared__ __attribute__((aligned(16))) unsigned char flush_lds[flush_lds_bytes<double>()];
  __shared__ __attribute__((aligned(16))) unsigned char pf_area[kPrefetchLdsBytes];
  __shared__ double2 tab_lds[kTabN];
  __shared__ double etab_lds[kExpTabN];
  fill_phasor_table(tab_lds);
  fill_exp_table(etab_lds);
  __syncthreads();
  skyvis_rec_body<double, CT, 2, false, 1>(p, flush_lds, pf_area, tab_lds, etab_lds);
}

// Wave items over a batch of snapshots (WITEM = 2 above): p.wave_snaps[p.wave_nsnap], p.nbgroups = ceil(nsnap nbw wave_nsplit / 4).
template <int CT>
__global__ __launch_bounds__(kBlockThreads) __attribute__((amdgpu_waves_per_eu(2)))
void k_skyvis_taper_f64_wave_batch(const SkyvisParams p) {
  __shared__ __attribute__((aligned(16))) unsigned char flush_lds[flush_lds_bytes<double>()];
  __shared__ __attribute__((aligned(16))) unsigned char pf_area[kPrefetchLdsBytes];
  __shared__ double2 tab_lds[kTabN];
  __shared__ double etab_lds[kExpTabN];
  fill_phasor_table(tab_lds);
  fill_exp_table(etab_lds);
  __syncthreads();
  skyvis_rec_body<double, CT, 2, false, 2>(p, flush_lds, pf_area, tab_lds, etab_lds);
}

// ------------------------------------------------------------------------------------------
// Fused visibility + baseline-gradient kernel (interferometry.py:6330, 6338, 6343):
//   V[b,f] and G_k[b,f] = sum_s dircos[s,k] * (summand), k = 0..2, in ONE pass over the sources.
//
// The four sums share the phasor; they differ by a per-source coefficient c_i(s) = (1, l_s, m_s, n_s).  That IS a small matrix
// product -- out[i][b] += sum_s c_i(s) p(s,f) * z(s,b,f) -- and it maps exactly onto v_mfma_f64_4x4x4_4b_f64
// (layout probed by tools/mfma_layout_test.hip: A[i][k] <- lane 16 k + 4 blk + i, B[k][j] <- lane 16 k + 4 blk + j,
// D[i][j] -> lane 16 i + 4 blk + j):
//   * a wavefront owns 16 baselines (4 blocks x 4 columns j) and walks the sources FOUR at a time: lane 16 k + q follows baseline q
//     of the wave and source 4 g + k, forms its phasor z by the same centre-seeded lifting recurrence as k_skyvis_rec (B operand),
//     and supplies A = c_(lane % 4)(s_k) * p(s_k, f) for its own source;
//   * one MFMA then adds, for all 16 baselines, the four sources into the four sums: 256 FMAs in ~17 cycles.  Lane 16 i + q ends with
//     sum i of baseline q: 2 CT doubles per lane.
// What it buys (profiles/r02_microbench_mfma.txt): on MI355X the fp64 MFMA runs on the SAME datapath as v_fma_f64 -- an MFMA loop and a
// VALU loop of two waves on one SIMD take the sum of their times, and 256 FMAs / 17 cycles is the VALU's 64 FMAs / 4 cycles -- so the
// matrix instruction adds no throughput; it removes instructions: per pair of channels 4 MFMAs (= 16 v_fma_f64 of work) + 8 VALU
// operations, against 4 passes x 10 VALU operations.  Measured on config 3 (tools/grad_timing.py): 287 ms against 476 ms for the four
// fp64 passes and 118 ms for a plain fp64 pass: V + gradient = 2.4 x a plain pass.  The accumulate FMAs alone (4 sums x re/im) are
// 1.6 x a plain pass's whole inner loop, so nothing on this datapath gets the gradient under ~2.2 x.  (fp32 requests run the GRAD
// bodies of the packed kernel, k_skyvis_grad_f32pk: plain packed FMAs -- an fp32 MFMA 4x4x1 form would share the fp32 datapath likewise
// and deliver 256 FMAs per ~9 cycles against v_pk_fma_f32's 128 per 4.)
// pbflux rows and directions are per lane group here (4 sources per wavefront), so they come through vector loads (4 distinct
// 16-byte addresses per instruction).
// ------------------------------------------------------------------------------------------
// STAGE (32-channel tiles without the taper): the four sources' rows (4 x 256 B) and directions (4 x 64 B) of the NEXT group are brought
// into a per-wave LDS area by two LDS-DMA loads (global_load_lds, no VGPRs) issued right after the current group's operands have been read
// out of the other half of that area, so the MFMA loop of one group hides the memory latency of the next (round 3: the per-lane vector
// loads at the loop top were waited for in place, SQ_WAIT_ANY 14 % of the wave cycles).
constexpr int kGradStageWaveBytes = 2 * (1024 + 256);
template <int CT, bool TAPER, bool LIFT>
__device__ __forceinline__ void skyvis_grad_f64_body(const SkyvisParams& p, const double2* tab, unsigned char* stage) {
  constexpr int HC = CT / 2;
  constexpr bool STAGE = CT == 32 && !TAPER;
  int slab, bg;
  if (!block_item(p, slab, bg)) return;          // p.nbgroups counts groups of 64 baselines for this kernel
  const int tile = slab % p.ntiles;              // the host plans nsplit = 1: every block walks all sources

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int k = lane >> 4;                        // source phase (B / A role), sum index i (D role)
  const int x = lane & 3;                         // A role: which coefficient this lane supplies
  const int64_t bw0 = (int64_t)bg * 64 + (tid >> 6) * 16;     // first baseline of this wave
  const int64_t b_raw = bw0 + (lane & 15);
  const bool b_valid = b_raw < p.nbl;
  const int64_t b = b_valid ? b_raw : (p.nbl - 1);
  if (bw0 >= p.nbl) return;                       // wave-uniform: whole wave out of range

  const double bx = p.bl_x[b], by = p.bl_y[b], bz = p.bl_z[b];
  const int k0 = tile * CT;
  const double fc = p.f0 + (double)(k0 + HC) * p.df;
  const double df = p.df;
  const double fcN = fc * kTabN, dfN = df * kTabN, dfN_half = df * (0.5 * kTabN);
  double bl2_c2 = 0.0, bpc = 0.0;
  if (TAPER) {
    bl2_c2 = (bx * bx + by * by + bz * bz) * (p.inv_c * p.inv_c);
    bpc = (bx * p.pc_x + by * p.pc_y + bz * p.pc_z) * p.inv_c;
  }

  double a_ur[HC], a_ui[HC], a_dr[HC], a_di[HC];    // sum (lane >> 4) of baseline (lane & 15): channels HC + j (up), HC - 1 - j (down)
#pragma unroll
  for (int j = 0; j < HC; ++j) { a_ur[j] = 0.0; a_ui[j] = 0.0; a_dr[j] = 0.0; a_di[j] = 0.0; }

  const double* const rows = reinterpret_cast<const double*>(p.pb_packed) + (size_t)tile * (size_t)p.nsrc_pad * CT;
  const double4* const prep = reinterpret_cast<const double4*>(p.dirs_prep);
  const double4* const raw = reinterpret_cast<const double4*>(p.dirs);
  const int64_t ns_pad = p.nsrc_pad;             // a multiple of 4 (the source chunk); rows and prepared directions past nsrc are zero

  // STAGE: this wave's two staging halves; lane L brings 16 bytes of row (L >> 4) and 4 bytes of the directions (lanes 0-31: prepared,
  // 32-63: raw, clamped to the last real source like the direct loads)
  unsigned char* const wst = stage + (tid >> 6) * kGradStageWaveBytes;
  auto stage_issue = [&](int64_t s0n, int half) {
    if constexpr (STAGE) {
      int ln = lane;
      asm volatile("" : "+v"(ln));
      const char* const grow = reinterpret_cast<const char*>(rows + (size_t)(s0n + (ln >> 4)) * CT) + (ln & 15) * 16;
      const int64_t sd = s0n + ((ln & 31) >> 3);
      const char* const gdir = (ln < 32) ? reinterpret_cast<const char*>(prep + sd) + (ln & 7) * 4
                                         : reinterpret_cast<const char*>(raw + (sd < p.nsrc ? sd : p.nsrc - 1)) + (ln & 7) * 4;
      __builtin_amdgcn_global_load_lds((gptr_t)grow, (lptr_t)(wst + half * 1280), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((gptr_t)gdir, (lptr_t)(wst + half * 1280 + 1024), 4, 0, 0);
    }
  };
  if (STAGE && ns_pad > 0) stage_issue(0, 0);
  for (int64_t s0 = 0; s0 < ns_pad; s0 += 4) {
    const int64_t s = s0 + k;
    double4 sv, rw;
    double2 pr[HC];
    if constexpr (STAGE) {
      const int half = (int)((s0 >> 2) & 1);
      __builtin_amdgcn_s_waitcnt(0x0F70);                                  // vmcnt(0): this group's two DMA loads have landed
      wave_lds_sync();
      const unsigned char* const hb = wst + half * 1280;
      sv = *reinterpret_cast<const double4*>(hb + 1024 + k * 32);
      rw = *reinterpret_cast<const double4*>(hb + 1024 + 128 + k * 32);
      const double2* const lrow = reinterpret_cast<const double2*>(hb + k * 256);
#pragma unroll
      for (int j = 0; j < HC; ++j) pr[j] = lrow[j];
      wave_lds_sync();                                                      // operands are in registers: the other half may be overwritten
      if (s0 + 4 < ns_pad) stage_issue(s0 + 4, half ^ 1);
    } else {
      sv = prep[s];
      rw = raw[s < p.nsrc ? s : p.nsrc - 1];
      const double2* const row = reinterpret_cast<const double2*>(rows + (size_t)s * CT);     // (up, down) pairs
#pragma unroll
      for (int j = 0; j < HC; ++j) pr[j] = row[j];
    }
    const double cx = (x == 0) ? 1.0 : (x == 1 ? rw.x : (x == 2 ? rw.y : rw.z));
    const double d = __builtin_fma(bx, sv.x, __builtin_fma(by, sv.y, bz * sv.z));

    double zc, zs, rc = 1.0, rs, tl = 0.0;
    {
      const TabPhase pc = sincos_tab_front(d * fcN, tab);
      const TabPhase ps = sincos_tab_front(d * (LIFT ? dfN_half : dfN), tab);
      sincos_tab_back(pc, zc, zs);
      if constexpr (LIFT) {
        double sb, cb;
        sincos_tab_back(ps, cb, sb);
        rs = 2.0 * sb * cb;
        tl = -div_unit_range_fast_f64(sb, cb);
      } else {
        sincos_tab_back(ps, rc, rs);
      }
    }
    double ur = zc, ui = -zs;
    const double rr = rc, ri = -rs;
    double dr, di;
    if constexpr (LIFT) {
      const double xd = __builtin_fma(tl, ui, ur);
      di = __builtin_fma(-ri, xd, ui);
      dr = __builtin_fma(tl, di, xd);
    } else {
      dr = __builtin_fma(ur, rr, ui * ri);
      di = __builtin_fma(ui, rr, -(ur * ri));
    }
    // The taper weight w(s, b, f) belongs to the (source, baseline) pair, i.e. to the B operand (the A operand of a lane serves all
    // four baselines of its block): B = w z, advanced with the weight's own recurrence.
    double wu = 1.0, wd = 1.0, qu = 1.0, qd = 1.0, h = 1.0;
    if constexpr (TAPER) {
      const double tau = d + bpc;
      double gq = sv.w * (bl2_c2 - tau * tau);
      gq = gq > 0.0 ? gq : 0.0;
      taper_seed_f64(gq, fc, df, wu, wd, qu, qd, h);
    }
#pragma unroll
    for (int j = 0; j < HC; ++j) {
      const double au = cx * pr[j].x, ad = cx * pr[j].y;
      double bur = ur, bui = ui, bdr = dr, bdi = di;
      if constexpr (TAPER) {
        bur *= wu; bui *= wu; bdr *= wd; bdi *= wd;
        wu *= qu; qu *= h; wd *= qd; qd *= h;
      }
      a_ur[j] = __builtin_amdgcn_mfma_f64_4x4x4f64(au, bur, a_ur[j], 0, 0, 0);
      a_ui[j] = __builtin_amdgcn_mfma_f64_4x4x4f64(au, bui, a_ui[j], 0, 0, 0);
      a_dr[j] = __builtin_amdgcn_mfma_f64_4x4x4f64(ad, bdr, a_dr[j], 0, 0, 0);
      a_di[j] = __builtin_amdgcn_mfma_f64_4x4x4f64(ad, bdi, a_di[j], 0, 0, 0);
      if constexpr (LIFT) {
        const double xu = __builtin_fma(-tl, ui, ur);
        const double yu = __builtin_fma(ri, xu, ui);
        ur = __builtin_fma(-tl, yu, xu); ui = yu;
        const double xd = __builtin_fma(tl, di, dr);
        const double yd = __builtin_fma(-ri, xd, di);
        dr = __builtin_fma(tl, yd, xd); di = yd;
      } else {
        const double nur = __builtin_fma(ur, rr, -(ui * ri));
        const double nui = __builtin_fma(ur, ri, ui * rr);
        const double ndr = __builtin_fma(dr, rr, di * ri);
        const double ndi = __builtin_fma(di, rr, -(dr * ri));
        ur = nur; ui = nui; dr = ndr; di = ndi;
      }
    }
  }
  // lane 16 i + q holds sum i (0: visibility, 1-3: gradient components) of baseline bw0 + q
  if (b_valid) {
    const int i = k;
    double2* const dst = (i == 0) ? reinterpret_cast<double2*>(p.out) : reinterpret_cast<double2*>(p.grad_out) + (size_t)(i - 1) * p.nbl * p.nchan;
    double2* const o = dst + (size_t)b * p.nchan;
#pragma unroll
    for (int j = 0; j < HC; ++j) {
      const int ku = k0 + HC + j, kd = k0 + HC - 1 - j;
      if (ku < p.nchan) o[ku] = make_double2(a_ur[j], a_ui[j]);
      if (kd < p.nchan) o[kd] = make_double2(a_dr[j], a_di[j]);
    }
  }
}

template <int CT, bool TAPER>
__global__ __launch_bounds__(kBlockThreads) __attribute__((amdgpu_waves_per_eu(2, 2)))
void k_skyvis_grad_f64(const SkyvisParams p) {
  __shared__ double2 tab_lds[kTabN];
  // (a separate LDS object: reads that might alias an LDS-DMA target are made to wait for vmcnt(0) -- the phasor-table reads must not)
  __shared__ __attribute__((aligned(16))) unsigned char stage_lds[(CT == 32 && !TAPER) ? (kBlockThreads / 64) * kGradStageWaveBytes : 16];
  fill_phasor_table(tab_lds);
  __syncthreads();
  int slab_, bg;
  const bool in_range = block_item(p, slab_, bg);
  if (in_range && p.lift_flags != nullptr && p.lift_flags[bg >> 2] != 0) {      // flags are kept per 256 baselines
    skyvis_grad_f64_body<CT, TAPER, true>(p, tab_lds, stage_lds);
    return;
  }
  skyvis_grad_f64_body<CT, TAPER, false>(p, tab_lds, stage_lds);
}

// ------------------------------------------------------------------------------------------
// Packed-fp32 recurrence kernel (the headline fp32 path).
//
// Same mapping as k_skyvis_rec (lanes = baselines, CT channels per thread, tile seeded at its centre), but
//   * every inner-loop instruction is a packed v_pk_fma_f32 on an (up-chain, down-chain) pair: register pair .x = channel
//     HC+j, .y = channel HC-1-j.  A packed instruction occupies the SIMD for 4 cycles and does two lanes' worth of work
//     (tools/microbench_valu.hip), which is what lets a thread own CT = 64 channels (128 accumulator VGPRs);
//   * pbflux rows are stored interleaved by k_pack (up_0, down_0, up_1, down_1, ...).  pbflux[s, tile] and the source
//     direction are wave-uniform, so they are fetched with SCALAR loads (s_load_dwordx16 through the scalar cache) straight
//     into SGPRs and used as the SGPR-pair operand of v_pk_fma_f32 (full rate, tools/microbench_trig.hip: 12.8e12 terms/s
//     for the bare lifting loop against 11.1-12.5e12 with LDS broadcast reads).  No LDS, no barriers, no VGPR staging of
//     the rows: waves run free and the no-taper kernel fits 3 waves per SIMD (157 VGPRs).
// Scalar loads return out of order, so the only wait is lgkmcnt(0); the half rows are requested one phase ahead:
//   top of source s:  wait (first half row + direction of s are there)  ->  request second half row of s
//                     seed arithmetic, pairs 0..HC/2-1
//   middle:           wait (second half row)                             ->  request first half row + direction of s+1
//                     pairs HC/2..HC-1
//
// LIFT: the step rotation uses the lifting (three-shear) form
//     x1 = x - t y,  y1 = y + s x1,  x2 = x1 - t y1,   t = tan(alpha/2), s = sin(alpha)
// = 3 dependent packed FMAs instead of 2 mul + 2 fma, i.e. 5 instead of 6 packed instructions per pair of terms
// (tools/microbench_inner.hip: 12.5e12 vs 11.1e12 terms/s for the bare loop).  With inexact (t, s) the map is still
// area-preserving and advances the phase by beta with cos(beta) = 1 - t s, so the angle error per step is ~ alpha * eps:
// it is only used where |alpha| <= pi/4 is GUARANTEED for every source (the host sets lift_flags[bg] when
// max|b| * max_s|s - s_pc| * |df| / c <= 1/8 cycle for the baseline group), all other groups take the 4-instruction rotation.
// ------------------------------------------------------------------------------------------
// waves per SIMD the packed kernels are built for: at 3 (168 VGPRs) the compiler spills around the flush (and, with the taper,
// inside the source loop); 2 waves measured 1-1.5 % faster on the same box and leave no scratch use at all
#ifndef PK_WAVES
#define PK_WAVES 2
#endif
typedef float f32x2 __attribute__((ext_vector_type(2)));


__device__ __forceinline__ f32x2 pkfma(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }

typedef const __attribute__((address_space(4))) float* cfloat_p;
typedef const __attribute__((address_space(4))) double* cdouble_p;
typedef const volatile __attribute__((address_space(4))) double* cvdouble_p;
typedef float f32x8 __attribute__((ext_vector_type(8)));
typedef const __attribute__((address_space(4))) f32x8* cf32x8_p;

// Tried and dropped (round 3, tools/ab_lib.py on one box): evaluating the seed's polynomial PAIRS -- (sin 2 pi y, tan pi y) of the lifting
// bodies, (cos, sin) of the small-step taper bodies, the two exp2m1_small of the grouped recurrence -- as the halves of packed FMAs.
// Bit-identical results and 6 + 4 + 8 fewer VALU instructions per (source, baseline, tile), but 61.5 instead of 60.9 ms on the headline
// launch and 82.8 instead of 82.4 ms with the taper: the constant pairs cost SGPRs (48 -> 48 spills, 112 -> 120 lane moves around the
// loops) and under the package power limit a packed FMA is not cheaper than the two scalar ones it replaces.
// exp2(D) - 1 to ~1e-8 absolute for the small per-step taper exponents of the grouped form: 5-term series in x = D ln2, clamped
// to |x| <= 1.  The host selects the grouped form only for df/f_min <= 3.4e-3, where |x| >= 1/8 implies a taper weight below
// 1.5e-8 on every channel of the tile, so the loss of accuracy (and the clamp) out there is immaterial.
__device__ __forceinline__ float exp2m1_small(float D) {
  const float x = __builtin_fminf(__builtin_fmaxf(D * 0.6931471805599453f, -1.0f), 1.0f);
  float e = __builtin_fmaf(x, 8.3333333e-3f, 4.1666668e-2f);
  e = __builtin_fmaf(e, x, 0.16666667f);
  e = __builtin_fmaf(e, x, 0.5f);
  e = __builtin_fmaf(e, x, 1.0f);
  return e * x;
}

// REANCHOR (taper bodies): the rounding of the step factor rho is the same at every step of a chain, so the error of zeta grows
// linearly with the step count.  Measured per term at the chain ends (32 steps, one source, tools/fuzz_parity.py and a single-source
// sweep on HERA-350 baselines): 3.0e-6 on the up chain but 5.0-6.9e-6 on the down chain, whose amplitude ratio is > 1 (floats just
// above 1 carry half the relative precision of floats just below), and more where the step angle is large (|alpha| ~ 1 rad on long
// baselines).  5e-6 is the tolerance and a sky may be dominated by one source, so the chains are re-formed exactly at their
// midpoint (hardware sin/cos of the fp64-reduced phase + one exp2): 1 = the down chain only (every taper run, ~4 % of its time),
// 2 = both chains (baseline groups whose step angle is not guaranteed <= pi/4).
// GRAD: visibility AND the three baseline-gradient sums G_k = sum_s dircos[s,k] * (summand) (interferometry.py:6330) in one pass: the
// term t = p * zeta is formed once (2 packed multiplies per pair) and added into four accumulator sets, the gradient ones through
// packed FMAs whose coefficient (l, l), (m, m), (n, n) is an SGPR-pair operand -- 13 packed instructions per pair of terms against
// 4 passes x 5.  CT = 16 (128 accumulator VGPRs); MFMA would not help: it shares the FMA datapath.
// TGROUP: 0 = exact per-step amplitude recurrence; 1 = grouped (8 steps at the group's geometric-mean ratio + parabola correction);
//   2, 3 = SPLIT forms of the grouped recurrence for a source range with ONE source size (kappa0, the case of every HEALPix sky:
//   FWHM = nside2resol for all pixels, run_prisim.py:1230-1246).  The taper exponent kappa (|b|^2 - (b.s)^2) f^2/c^2 splits into a
//   source-independent part, exp(-kappa0 |b|^2 f^2/c^2), applied ONCE per flush to the fp32 partial sums (exactly, per baseline and
//   channel), and exp(+kappa0 (b.s)^2 f^2/c^2), carried by the recurrence.  For a sky seen through a beam the second exponent is
//   small wherever the flux is ((b.s)^2 <= |b|^2 sin^2 theta), so the parabola the grouped form leaves inside a group --
//   16 kappa (b.s)^2 df^2/c^2 at most, relative to the term -- can be BOUNDED from beam-weighted moments of the sky (the host does,
//   per baseline group, capi.cpp:taper_split_plan) and, where that bound is below 2e-7 of sum|pbflux|, not corrected at all:
//   3 = no parabola correction (6.375 packed instructions per pair of terms instead of 7.25), 2 = corrected (groups that fail the bound),
//   4 = no correction and groups of 16 steps (the parabola is 4x larger: groups whose bound passes with that factor; 6.25 per pair).
//   GPK (GRAD bodies without the taper): the rows arrive PRE-MULTIPLIED by the gradient coefficients -- k_pack_grad writes, per source and
//   16-channel tile, the four operand rows p, p l, p m, p n interleaved per pair as (set, up / down), 64 floats = one 256-byte row like a
//   64-channel tile's -- so every accumulator set takes its own SGPR-pair operand straight into v_pk_fma_f32: 8 accumulate FMAs + the
//   3-instruction lifting rotation = 11 packed instructions per pair of terms instead of 13 (term = p zeta first, then four adds / FMAs), and
//   the per-source coefficient load disappears.
template <int CT, bool TAPER, bool LIFT, int TGROUP = 0, int REANCHOR = 0, bool GRAD = false, bool GPK = false>
__device__ __forceinline__ void skyvis_rec_f32pk_body(const SkyvisParams& p, unsigned char* flush_lds) {
  static_assert(!GPK || (GRAD && !TAPER), "pre-multiplied rows: the gradient bodies without the taper");
  constexpr int NR = GRAD ? 4 : 1;                   // accumulator sets: V (+ G_l, G_m, G_n)
  constexpr bool SPLIT = TGROUP >= 2;
  constexpr bool PARABOLA = TGROUP == 1 || TGROUP == 2;
  constexpr int GS = TGROUP == 4 ? 16 : 8;            // steps per group of the grouped forms
  static_assert(!(TAPER && LIFT), "the taper-folded recurrence is a scaled rotation: no lifting form");
  static_assert(TAPER || !TGROUP, "TGROUP is a taper variant");
  static_assert(!(SPLIT && GRAD), "the split taper form is built for the plain sky-sum");
  static_assert(TAPER || REANCHOR == 0, "REANCHOR is a taper variant");
  constexpr int HC = CT / 2;
  // pieces per row: halves (2 x 32 SGPRs at CT = 64) without the taper; quarters with it, whose extra wave-uniform state would
  // otherwise push the row buffers out of the ~100 SGPRs (48 v_readlane/v_writelane per source in the loop)
  constexpr int NPART = (TAPER && CT >= 64) ? 4 : 2;
  constexpr int ROWF = GPK ? 4 * CT : CT;            // floats per (source, tile) row
  constexpr int NP = ROWF / NPART;                   // floats per piece

  int slab, bg;
  if (!block_item(p, slab, bg)) return;
  const int tile = slab % p.ntiles;
  const int split = slab / p.ntiles;

  // sources [src_lo, src_hi) of the sky (the whole sky unless the host walks it in ranges of one source size), cut into nsplit pieces
  int64_t s_begin = p.src_lo + (int64_t)split * p.src_per_split;
  int64_t s_end = s_begin + p.src_per_split;
  if (s_end > p.src_hi) s_end = p.src_hi;
  if (TAPER && p.src_first != nullptr) {
    // taper culling: the group's leading sources are provably below the tolerance (capi.cpp).  What is left is cut into nsplit EQUAL
    // pieces again, so that every split of the group shrinks alike (the XCD map deals whole slabs to XCDs: skipping only the first
    // split's sources would idle one XCD and leave the launch as long as before)
    const int64_t f = p.src_first[bg];
    if (f > p.src_lo) {
      const int64_t per = (p.src_hi - f + p.nsplit - 1) / p.nsplit;
      s_begin = f + (int64_t)split * per;
      s_end = s_begin + per < p.src_hi ? s_begin + per : p.src_hi;
    }
  }

  const int tid = threadIdx.x;
  const int64_t b_raw = (int64_t)bg * kBlockThreads + tid;
  const bool b_valid = b_raw < p.nbl;
  const int64_t b = b_valid ? b_raw : (p.nbl - 1);
  const bool wave_active = ((int64_t)bg * kBlockThreads + (tid & ~63)) < p.nbl;

  const double bx = p.bl_x[b], by = p.bl_y[b], bz = p.bl_z[b];
  const int k0 = tile * CT;
  const double fc_hz = p.f0 + (double)(k0 + HC) * p.df;
  const double df4 = 4.0 * p.df;
  double bl2_c2 = 0.0, bpc = 0.0;
  if (TAPER) {
    bl2_c2 = (bx * bx + by * by + bz * bz) * (p.inv_c * p.inv_c);
    bpc = (bx * p.pc_x + by * p.pc_y + bz * p.pc_z) * p.inv_c;
  }
  // log2 w = A + B j + C j^2 per (source, baseline): A = gq kA, B = gq kB, C = gq kC with gq = kappa (|b|^2/c^2 - tau^2)
  const double kA = -1.4426950408889634 * fc_hz * fc_hz;
  const float kBf = (float)(-2.0 * 1.4426950408889634 * fc_hz * p.df);
  const float kCf = (float)(-1.4426950408889634 * p.df * p.df);

  f32x2 acc_re[NR][HC], acc_im[NR][HC];
#pragma unroll
  for (int r = 0; r < NR; ++r)
#pragma unroll
    for (int j = 0; j < HC; ++j) { acc_re[r][j] = (f32x2)(0.f); acc_im[r][j] = (f32x2)(0.f); }

  const cfloat_p gp = (cfloat_p)(uintptr_t)(reinterpret_cast<const float*>(p.pb_packed) + (size_t)tile * (size_t)p.nsrc_pad * ROWF);
  const cdouble_p gd = (cdouble_p)(uintptr_t)p.dirs_prep;
  double2* const out = reinterpret_cast<double2*>(p.out) + (size_t)split * p.nbl * p.nchan;
  bool first_flush = p.accumulate == 0;              // accumulate: an earlier launch (another source range) already wrote this slot
  float2* const wbuf = reinterpret_cast<float2*>(flush_lds) + (tid >> 6) * (64 * 17);
  const int lane = tid & 63;
  const int64_t bw0 = (int64_t)bg * kBlockThreads + (tid & ~63);      // first baseline of this wave
  // SPLIT: log2 of the source-independent taper factor of a baseline at frequency f is kE f^2, kE = -log2(e) kappa0 |b|^2/c^2.  It is
  // applied in the flush's STORE loop (few live registers there), so every lane parks its kE in LDS for the lanes that store its row.
  double* const ke_lds = reinterpret_cast<double*>(flush_lds + flush_lds_bytes<float>() + kPrefetchLdsBytes) + (tid & ~63);
  auto escale = [&](double ke, int kchan) -> double {
    const double f = p.f0 + (double)kchan * p.df;
    const double x = ke * f * f;                     // <= 0; integer part by ldexp, fraction by the hardware exp2: ~1e-7 relative
    const double xi = __builtin_rint(x);
    return (double)__builtin_ldexpf(__builtin_amdgcn_exp2f((float)(x - xi)), (int)__builtin_fmax(xi, -300.0));
  };

  // fp32 partial sums -> fp64 cube (read-modify-write after the first flush), transposed through LDS 16 channels at a time:
  // piece pz holds the pairs j = 8 pz .. 8 pz + 7, i.e. channels HC+8pz .. HC+8pz+7 (columns 0-7) and HC-1-8pz .. HC-8-8pz
  // (columns 8-15); 16 lanes write the 2 x 128 contiguous bytes of one baseline, 4 baselines per store instruction.
  auto flush = [&]() {
    if (wave_active) {
      int lane_o = lane;
      asm volatile("" : "+v"(lane_o));        // opaque: keeps the 64 store addresses from being hoisted out of the segment loop and spilled
      if constexpr (SPLIT) {
        ke_lds[lane_o] = -1.4426950408889634 * p.kappa0 * bl2_c2;
        wave_lds_sync();
      }
#pragma unroll
      for (int rs = 0; rs < NR; ++rs) {
      // destination of accumulator set rs: the visibility slot, or gradient component rs - 1 of this slot
      double2* const outr = (rs == 0) ? out : reinterpret_cast<double2*>(p.grad_out) + (size_t)(rs - 1) * p.nbl * p.nchan;
#pragma unroll
      for (int pz = 0; pz < HC / 8; ++pz) {
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) {
          const int j = 8 * pz + jj;
          wbuf[lane * 17 + jj] = make_float2(acc_re[rs][j].x, acc_im[rs][j].x);
          wbuf[lane * 17 + 8 + jj] = make_float2(acc_re[rs][j].y, acc_im[rs][j].y);
          acc_re[rs][j] = (f32x2)(0.f); acc_im[rs][j] = (f32x2)(0.f);   // dead during the stores below
        }
        wave_lds_sync();
        const int c = lane_o & 15;
        const int k = k0 + ((c < 8) ? (HC + 8 * pz + c) : (HC - 1 - 8 * pz - (c - 8)));
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int bi = r * 4 + (lane_o >> 4);
          const float2 a = wbuf[bi * 17 + c];
          const int64_t bb = bw0 + bi;
          if (bb < p.nbl && k < p.nchan) {
            if (!GRAD && p.out_f32) {                              // complex64 partial of a source split: written once, no read-modify-write
              float2 a2 = a;
              if constexpr (SPLIT) {                               // every partial carries the flush factor (the sum of the partials is linear in it)
                const float e = (float)escale(ke_lds[bi], k);
                a2.x *= e; a2.y *= e;
              }
              reinterpret_cast<float2*>(p.out)[((size_t)split * p.nbl + (size_t)bb) * p.nchan + k] = a2;
            } else {
              double2* o = outr + (size_t)bb * p.nchan + k;
              double2 v = make_double2((double)a.x, (double)a.y);
              if constexpr (SPLIT) {
                const double e = escale(ke_lds[bi], k);            // the source-independent half of the taper, exact per (baseline, channel)
                v.x *= e; v.y *= e;
              }
              if (!first_flush) { const double2 old = *o; v.x += old.x; v.y += old.y; }
              *o = v;
            }
          }
          __builtin_amdgcn_sched_barrier(0);                       // one store at a time: the flush must fit beside 128 accumulators
        }
        wave_lds_sync();
      }
      }
    }
#pragma unroll
    for (int r = 0; r < NR; ++r)
#pragma unroll
      for (int j = 0; j < HC; ++j) { acc_re[r][j] = (f32x2)(0.f); acc_im[r][j] = (f32x2)(0.f); }
    first_flush = false;
  };

  // Sources are walked in segments of flush_src: the fp32 partial sums of a segment are then added into the fp64 cube.
  // One flush site, 32-bit source indices (the host keeps nsrc below 2^31).
  const int n_loc = s_end > s_begin ? (int)(s_end - s_begin) : 0;
  const int seg_len = p.flush_src > 0 ? p.flush_src : 0x7fffffff;
  const cfloat_p gps = gp + (size_t)s_begin * ROWF;
  const cdouble_p gds = gd + (size_t)s_begin * 4;
  const float* const pf_rows = reinterpret_cast<const float*>(p.pb_packed) + (size_t)tile * (size_t)p.nsrc_pad * ROWF + (size_t)s_begin * ROWF;
  const float* const pf_dirs = reinterpret_cast<const float*>(p.dirs_prep) + (size_t)s_begin * 8;
  const lptr_t pf_lds = (lptr_t)(flush_lds + flush_lds_bytes<float>() + (tid >> 6) * kPrefetchWaveBytes);
  const bool pf_on = n_loc >= 64;                    // 4 (CT = 64) or 8 rows per request, clamped 8 rows before the end
  int seg0 = 0;
  do {
  const int seg1 = (n_loc - seg0 > seg_len) ? seg0 + seg_len : n_loc;
  if (wave_active && seg1 > seg0) {
    // the row is fetched in NPART pieces through two SGPR buffers (piece k in buffer k & 1), one piece ahead of its use
    float ra[NP], rb[NP];
    double sv[4] = {0.0, 0.0, 0.0, 0.0};
    f32x8 cs = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};      // GRAD: (l, l, m, m, n, n, 0, 0) of the source: three SGPR pairs, one s_load_dwordx8
    const cf32x8_p gcs = (cf32x8_p)(uintptr_t)((GRAD && !GPK) ? p.dirs_c32 + (size_t)s_begin * 8 : nullptr);
    {
      if constexpr (GRAD && !GPK) cs = gcs[seg0];
      const cfloat_p r0 = gps + (size_t)seg0 * ROWF;
#pragma unroll
      for (int i = 0; i < NP; ++i) ra[i] = r0[i];
      // volatile: keeps instcombine from folding phi(load before the loop, load in the loop) into one load of a phi'd address at
      // the loop TOP -- right before the seed that needs it, where every source would wait out a scalar-cache round trip
      const cvdouble_p d0 = gds + (size_t)seg0 * 4;
      sv[0] = d0[0]; sv[1] = d0[1]; sv[2] = d0[2];
      if (TAPER) sv[3] = d0[3];
    }
    for (int s = seg0; s < seg1; ++s) {
      const cfloat_p row = gps + (size_t)s * ROWF;
      const int sn = (s + 1 < seg1) ? s + 1 : s;                    // the last source is simply fetched again
      if (pf_on && ((s - seg0) & 3) == 0) {
        // every 4th source: the next 4 rows (64 lanes x 16 B) and 8 directions (64 lanes x 4 B), kPrefetchAhead sources ahead
        constexpr int kPfRows = (256 / ROWF) > 8 ? (256 / ROWF) : 8;      // rows one 1 KiB request covers (>= the 8 directions)
        const int spf = (s + kPrefetchAhead < n_loc - kPfRows) ? s + kPrefetchAhead : n_loc - kPfRows;
        int lane_pf = lane;
        asm volatile("" : "+v"(lane_pf));                            // formed here from the lane id: no per-lane pointers kept live across the loop
        __builtin_amdgcn_global_load_lds((gptr_t)(pf_rows + (size_t)spf * ROWF + lane_pf * 4), pf_lds, 16, 0, 0);
        __builtin_amdgcn_global_load_lds((gptr_t)(pf_dirs + (size_t)spf * 8 + lane_pf), pf_lds, 4, 0, 0);
      }
      // the first use of sv waits for everything in flight (first piece + direction); only then ask for the second piece
      const double d = __builtin_fma(bx, sv[0], __builtin_fma(by, sv[1], bz * sv[2]));
      const f32x2 CL = {cs[0], cs[1]}, CM = {cs[2], cs[3]}, CN = {cs[4], cs[5]};     // this source's coefficients (GRAD)
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < NP; ++i) rb[i] = row[NP + i];
      __builtin_amdgcn_sched_barrier(0);
      float zc, zs;
      sincos_cycles_hw(d * fc_hz, zc, zs);
      const float ur0 = zc, ui0 = -zs;
      float rr = 1.f, ri, tpy = 0.f, dr0, di0;
      if (LIFT) {
        const float yth = (float)(d * p.df);
        ri = -sin_2pi_y(yth);
        tpy = tan_pi_y(yth);
        const float x1 = __builtin_fmaf(-tpy, ui0, ur0);
        di0 = __builtin_fmaf(-ri, x1, ui0);
        dr0 = __builtin_fmaf(-tpy, di0, x1);
      } else {
        if (TAPER && REANCHOR != 2) {
          // taper bodies of baseline groups whose |theta| <= 1/8 cycle is guaranteed (the host's lift flag): no quadrant logic
          const float yth = (float)(d * p.df);
          rr = cos_2pi_y(yth);
          ri = -sin_2pi_y(yth);
        } else {
          float rc, rs;
          sincos_qcycles(d * df4, rc, rs);
          rr = rc; ri = -rs;
        }
        dr0 = __builtin_fmaf(ur0, rr, ui0 * ri);
        di0 = __builtin_fmaf(ui0, rr, -(ur0 * ri));
      }
      const f32x2 NT = {tpy, -tpy};                    // (-t_up, -t_down)
      const f32x2 SS = {ri, -ri};                      // (sin alpha_up, sin alpha_down)
      f32x2 zre = {ur0, dr0};
      f32x2 zim = {ui0, di0};
      const f32x2 RR = {rr, rr};
      const f32x2 RI = {-ri, ri};                      // re' = re*rr + im*RI ;  im' = im*rr - re*RI
      // Source-shape taper folded into the recurrence.  log2 w at channel HC+j is L(j) = A + B j + C j^2 with
      //   A = -G fc^2, B = -2 G fc df, C = -G df^2, G = kappa_s (|b|^2/c^2 - tau^2) log2(e)     (interferometry.py:6265-6283)
      // so zeta_j = w_j z_j advances by the complex factor rho_j = r * exp2(L(j+1)-L(j)) (up) / conj(r) * exp2(L(-2-j)-L(-1-j)) (down).
      //  * exact form (TGROUP = false): rho_{j+1} = rho_j * exp2(2C), 2 more packed instructions per pair of terms instead of two
      //    v_exp_f32; rho is re-formed exactly every RESEED steps so that its rounding error cannot random-walk into zeta's phase.
      //  * grouped form (TGROUP): within a group of 8 steps rho is held at the group's geometric-mean ratio
      //    rho_g = r * exp2(B + C (16 g + 8)) (up), conj(r) * exp2(-B + C (16 g + 10)) (down), which is exact at the group ends and
      //    low by exp2(C m (8 - m)) at step m inside; that known parabola is put back on the (wave-uniform) pbflux operand,
      //    p_eff = p + p * (-ln2 C m (8 - m)): 1 packed FMA on 7 of 8 steps instead of 2 on every step, and rho_g moves to the next
      //    group with 2.  Residual: (ln2 C m(8-m))^2 / 2 relative to the term; the host only
      //    selects this form when that is < 1e-8 of sum|pbflux| for every possible source (df / f_min <= 3.4e-3).
      const f32x2 RIC = {ri, -ri};                     // imaginary part of (r, conj r)
      f32x2 rho_re = RR, rho_im = RIC, HM = {0.f, 0.f}, EQ = {0.f, 0.f};
      f32x2 EK[4] = {{0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}};
      float tB = 0.f, tC = 0.f, tA_keep = 0.f;
      if (TAPER) {
        const double tau = d + bpc;
        double gq;
        if constexpr (SPLIT) {
          gq = -(sv[3] * tau * tau);                     // only the source-dependent part rides the recurrence (an amplitude that GROWS with f)
        } else {
          gq = sv[3] * (bl2_c2 - tau * tau);
          gq = gq > 0.0 ? gq : 0.0;                      // |b|^2 >= (b.s)^2 up to rounding
        }
        const float tA = (float)(gq * kA);               // |A| can reach tens of octaves: fp64 product; B and C are small
        const float gqf = (float)gq;
        tB = gqf * kBf;
        tC = gqf * kCf;
        tA_keep = tA;
        float w_u0, w_d0;
        if constexpr (SPLIT) {
          // the in-loop weight exp2(A) can be LARGE here (it is cancelled by the flush factor), so a float exponent of tens of octaves
          // would cost 1e-6 of a term with w ~ 1: integer part by ldexp, fraction by the hardware exp2; the other channels hang on it
          const double tAd = gq * kA;
          const double ti = __builtin_rint(tAd);
          w_u0 = __builtin_ldexpf(__builtin_amdgcn_exp2f((float)(tAd - ti)), (int)ti);
          w_d0 = w_u0 * __builtin_amdgcn_exp2f(tC - tB);
          tA_keep = w_u0;                                 // REANCHOR re-forms amplitudes as w_u0 * exp2(B j + C j^2)
        } else {
          w_u0 = __builtin_amdgcn_exp2f(tA);                           // channel HC
          w_d0 = __builtin_amdgcn_exp2f(tA - tB + tC);                 // channel HC-1
        }
        zre = zre * (f32x2){w_u0, w_d0};
        zim = zim * (f32x2){w_u0, w_d0};
        const float th = (TGROUP ? 2.0f * GS : 2.0f) * tC * 0.6931471805599453f;   // exp2(2C) - 1 (exp2(2 GS C) - 1) = th + th^2/2 + ...
        const float hm = __builtin_fmaf(0.5f * th, th, th);
        HM = (f32x2){hm, hm};
        if (PARABOLA) {
          const float c = -0.6931471805599453f * tC;                     // >= 0 (SPLIT: <= 0, the held ratio is then HIGH inside a group)
          EK[0] = (f32x2){7.f * c, 7.f * c};
          EK[1] = (f32x2){12.f * c, 12.f * c};
          EK[2] = (f32x2){15.f * c, 15.f * c};
          EK[3] = (f32x2){16.f * c, 16.f * c};
        }
      }
      constexpr int RESEED = 8;

      auto pairs = [&](const float (&r)[NP], int jbase) {
        if constexpr (GPK) {
          // 8 floats per pair: (set 0 .. 3) x (up, down) of p c_set; the piece holds NP / 8 pairs
#pragma unroll
          for (int jj = 0; jj < NP / 8; ++jj) {
            const int j = jbase + jj;
#pragma unroll
            for (int rs = 0; rs < 4; ++rs) {
              const f32x2 pr = {r[8 * jj + 2 * rs], r[8 * jj + 2 * rs + 1]};
              acc_re[rs][j] = pkfma(pr, zre, acc_re[rs][j]);
              acc_im[rs][j] = pkfma(pr, zim, acc_im[rs][j]);
            }
            if (LIFT) {
              const f32x2 x1 = pkfma(NT, zim, zre);
              const f32x2 y1 = pkfma(SS, x1, zim);
              zre = pkfma(NT, y1, x1);
              zim = y1;
            } else {
              const f32x2 t0 = zim * RI;
              const f32x2 t1 = zre * RI;
              const f32x2 nre = pkfma(zre, RR, t0);
              const f32x2 nim = pkfma(zim, RR, -t1);
              zre = nre; zim = nim;
            }
          }
          return;
        }
#pragma unroll
        for (int jj = 0; jj < NP / 2; ++jj) {
          const int j = jbase + jj;
          f32x2 pp = {r[2 * jj], r[2 * jj + 1]};
          if (REANCHOR != 0 && HC >= 32 && j == HC / 2) {
            // exact zeta at channel HC - 1 - j (down) and, REANCHOR == 2, HC + j (up): phase d f, amplitude exp2(L(-1-j)) / exp2(L(j))
            float cd, sd;
            sincos_cycles_hw(d * (fc_hz - (double)(j + 1) * p.df), cd, sd);
            const float wd = SPLIT ? tA_keep * __builtin_amdgcn_exp2f(__builtin_fmaf(tC, (float)((j + 1) * (j + 1)), tB * -(float)(j + 1)))
                                   : __builtin_amdgcn_exp2f(__builtin_fmaf(tC, (float)((j + 1) * (j + 1)), __builtin_fmaf(tB, -(float)(j + 1), tA_keep)));
            zre.y = cd * wd;
            zim.y = -(sd * wd);
            if (REANCHOR == 2) {
              float cu, su;
              sincos_cycles_hw(d * (fc_hz + (double)j * p.df), cu, su);
              const float wu = SPLIT ? tA_keep * __builtin_amdgcn_exp2f(__builtin_fmaf(tC, (float)(j * j), tB * (float)j))
                                     : __builtin_amdgcn_exp2f(__builtin_fmaf(tC, (float)(j * j), __builtin_fmaf(tB, (float)j, tA_keep)));
              zre.x = cu * wu;
              zim.x = -(su * wu);
            }
          }
          if (PARABOLA) {
            constexpr int kmap[8] = {0, 0, 1, 2, 3, 2, 1, 0};          // m (8 - m) = 7, 12, 15, 16, 15, 12, 7 for m = 1..7
            const int m = j % 8;
            if (m != 0) pp = pkfma(pp, EK[kmap[m]], pp);
          }
          if constexpr (GRAD) {
            const f32x2 tre = pp * zre, tim = pp * zim;
            acc_re[0][j] += tre; acc_im[0][j] += tim;
            acc_re[1][j] = pkfma(tre, CL, acc_re[1][j]); acc_im[1][j] = pkfma(tim, CL, acc_im[1][j]);
            acc_re[2][j] = pkfma(tre, CM, acc_re[2][j]); acc_im[2][j] = pkfma(tim, CM, acc_im[2][j]);
            acc_re[3][j] = pkfma(tre, CN, acc_re[3][j]); acc_im[3][j] = pkfma(tim, CN, acc_im[3][j]);
          } else {
            acc_re[0][j] = pkfma(pp, zre, acc_re[0][j]);
            acc_im[0][j] = pkfma(pp, zim, acc_im[0][j]);
          }
          if (LIFT) {
            // x1 = x - t y, y1 = y + s x1, x2 = x1 - t y1
            const f32x2 x1 = pkfma(NT, zim, zre);
            const f32x2 y1 = pkfma(SS, x1, zim);
            zre = pkfma(NT, y1, x1);
            zim = y1;
          } else if (!TAPER) {
            const f32x2 t0 = zim * RI;
            const f32x2 t1 = zre * RI;
            const f32x2 nre = pkfma(zre, RR, t0);
            const f32x2 nim = pkfma(zim, RR, -t1);
            zre = nre; zim = nim;
          } else {
            if (TGROUP) {
              // rho_g = r * q_g is applied 8 times with the SAME rounding error, and v_exp_f32's 1 ulp (6e-8 of a number just below
              // 1, 1.2e-7 just above) plus a product rounding per group put 3.7e-6 / 5.0e-6 of one term on the ends of the up / down
              // chains.  So q_g - 1 =: e_g is carried instead (accurate to ~1e-8: series at the first group, e_{g+1} = e_g + h + e_g h
              // after that) and rho_g = r + r e_g is one FMA: what is left is the rounding of r and of rho itself.
              if (j == 0) {
                EQ = (f32x2){exp2m1_small(__builtin_fmaf(tC, (float)GS, tB)), exp2m1_small(__builtin_fmaf(tC, (float)(GS + 2), -tB))};
              } else if ((j % GS) == 0) {
                EQ = pkfma(EQ, HM, EQ + HM);                 // mean ratio of the next group: (1 + e)(1 + h) - 1, h = exp2(2 GS C) - 1
              }
              if ((j % GS) == 0) {
                rho_re = pkfma(RR, EQ, RR);
                rho_im = pkfma(RIC, EQ, RIC);
              }
            } else if ((j % RESEED) == 0) {
              // exact per-step ratio at step j
              const float qu = __builtin_amdgcn_exp2f(__builtin_fmaf(tC, (float)(2 * j + 1), tB));
              const float qd = __builtin_amdgcn_exp2f(__builtin_fmaf(tC, (float)(2 * j + 3), -tB));
              rho_re = (f32x2){qu * rr, qd * rr};
              rho_im = (f32x2){qu * ri, -(qd * ri)};
            }
            const f32x2 t0 = zim * rho_im;
            const f32x2 t1 = zre * rho_im;
            const f32x2 nre = pkfma(zre, rho_re, -t0);
            const f32x2 nim = pkfma(zim, rho_re, t1);
            zre = nre; zim = nim;
            if (!TGROUP) {
              rho_re = pkfma(rho_re, HM, rho_re);
              rho_im = pkfma(rho_im, HM, rho_im);
            }
          }
        }
      };
      pairs(ra, 0);
#pragma unroll
      for (int ph = 1; ph < NPART; ++ph) {
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_waitcnt(0xC07F);          // lgkmcnt(0): piece ph has landed before the next request goes out
        __builtin_amdgcn_sched_barrier(0);
        if (ph + 1 < NPART) {
          // piece ph + 1 of this source into the buffer piece ph - 1 has just left
#pragma unroll
          for (int i = 0; i < NP; ++i) {
            if (ph & 1) ra[i] = row[(ph + 1) * NP + i];
            else rb[i] = row[(ph + 1) * NP + i];
          }
        } else {
          // first piece + direction of the next source (NPART is even: it goes to ra)
          const cfloat_p rn = gps + (size_t)sn * ROWF;
#pragma unroll
          for (int i = 0; i < NP; ++i) ra[i] = rn[i];
          const cvdouble_p dn = gds + (size_t)sn * 4;
          sv[0] = dn[0]; sv[1] = dn[1]; sv[2] = dn[2];
          if (TAPER) sv[3] = dn[3];
          if constexpr (GRAD && !GPK) cs = gcs[sn];
        }
        __builtin_amdgcn_sched_barrier(0);
        if (ph & 1) pairs(rb, ph * (GPK ? NP / 8 : NP / 2));
        else pairs(ra, ph * (GPK ? NP / 8 : NP / 2));
      }
    }
  }
  flush();
  seg0 = seg1;
  } while (seg0 < n_loc);
}

// ------------------------------------------------------------------------------------------
// V + baseline gradient in fp64 WITH the source-shape taper, in the grouped single-chain form of k_skyvis_taper_f64 (round 5).
// Round 3's form (skyvis_grad_f64_body<16, true>: exact second-order amplitude recurrence beside the lifting rotation, 16-channel tiles
// because its per-lane state did not fit beside 128 accumulator VGPRs, a 111-instruction seed with a library exp) ran the reference's
// default-precision gradient on a diffuse sky at 0.27 of the 16-flop contract.  Here: 32-channel tiles, ONE chain from the tile's first
// channel, zeta = w z advanced by the complex factor rho = r q_t held over a group of 8 steps (4 instructions), the parabola the held
// ratio leaves put back on the B operand -- it depends on (source, baseline) through tau, and the A operand c_i(s) p(s, f) serves all
// four baselines of its MFMA block -- and the four sources' rows and directions staged through LDS by two DMA loads a group ahead,
// read out eight channels at a time (the row operands never occupy more than 16 VGPRs).
// BATCH (k_skyvis_grad_taper_f64_batch, arrays of at most 256 baselines): wave items over a batch of snapshots, as WITEM = 2 of
// skyvis_rec_body -- item g = 4 bg + wave is (snapshot, source split, wave of 16 baselines); the snapshot's rows, directions, phase
// centre and the destinations of its four sums (cube / gradient slot, or its partial cubes when the sources are split) come from the
// wave-uniform table p.wave_snaps.  p.wave_nbw counts waves of SIXTEEN baselines here.
template <bool BATCH>
__device__ __forceinline__ void skyvis_grad_taper_f64_body(const SkyvisParams& p, const double2* tab, const double* etab, unsigned char* stage_lds) {
  constexpr int CT = 32;
  int slab, bg;
  if (!block_item(p, slab, bg)) return;          // p.nbgroups counts groups of 64 baselines for this kernel (BATCH: quads of wave items)
  const int tile = slab % p.ntiles;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int k = lane >> 4;                        // source of the group of four this lane follows (B / A role), sum index (D role)
  const int x = lane & 3;                         // A role: which coefficient this lane supplies
  int64_t bw0 = (int64_t)bg * 64 + (tid >> 6) * 16;
  // the snapshot's source rows [s_lo, s_hi) (multiples of four), directions, phase centre and destinations
  int64_t s_lo = 0, s_hi = p.nsrc_pad, dir0 = 0, dir_last = p.nsrc - 1;
  double pc_x = p.pc_x, pc_y = p.pc_y, pc_z = p.pc_z;
  double* out_v = p.out;
  double* out_g = p.grad_out;
  if constexpr (BATCH) {
    const int g = bg * (kBlockThreads / 64) + __builtin_amdgcn_readfirstlane((int)(tid >> 6));
    const int per_snap = p.wave_nbw * p.wave_nsplit;
    int snap = g / per_snap;
    if (snap >= p.wave_nsnap) return;             // wave-uniform: a padding item of the last block
    snap = __builtin_amdgcn_readfirstlane(snap);
    const int r = g - snap * per_snap;
    const int split = r / p.wave_nbw;
    bw0 = (int64_t)(r - split * p.wave_nbw) * 16;
    const BatchSnap* sn = p.wave_snaps + snap;
    const int64_t nrow = sn->nsrc > 0 ? sn->nrow : 0;      // (an empty region of interest: nothing is read, zeros are written)
    s_lo = (int64_t)split * sn->src_per_split;
    s_hi = s_lo + sn->src_per_split < nrow ? s_lo + sn->src_per_split : nrow;
    if (s_lo > s_hi) s_lo = s_hi;
    s_lo += sn->row0; s_hi += sn->row0;
    dir0 = sn->dir0 - sn->row0;                   // raw direction of packed row s: dirs[dir0 + s]
    dir_last = sn->dir0 + sn->nsrc - 1;
    pc_x = sn->pc[0]; pc_y = sn->pc[1]; pc_z = sn->pc[2];
    out_v = sn->out + (size_t)split * (size_t)p.nbl * p.nchan * 2;
    out_g = sn->gout + (size_t)split * (size_t)p.nbl * p.nchan * 6;
  }
  const int64_t b_raw = bw0 + (lane & 15);
  const bool b_valid = b_raw < p.nbl;
  const int64_t b = b_valid ? b_raw : (p.nbl - 1);
  if (bw0 >= p.nbl) return;

  const double bx = p.bl_x[b], by = p.bl_y[b], bz = p.bl_z[b];
  const int k0 = tile * CT;
  const double fc = p.f0 + (double)k0 * p.df;     // the chain starts at the tile's first channel
  const double df = p.df;
  const double fcN = fc * kTabN, dfN = df * kTabN;
  const double bl2_c2 = (bx * bx + by * by + bz * bz) * (p.inv_c * p.inv_c);
  const double bpc = (bx * pc_x + by * pc_y + bz * pc_z) * p.inv_c;

  double a_r[CT], a_i[CT];                        // sum (lane >> 4) of baseline (lane & 15), channel k0 + j
#pragma unroll
  for (int j = 0; j < CT; ++j) { a_r[j] = 0.0; a_i[j] = 0.0; }

  const double* const rows = reinterpret_cast<const double*>(p.pb_packed) + (size_t)tile * (size_t)p.nsrc_pad * CT;      // natural channel order
  const double4* const prep = reinterpret_cast<const double4*>(p.dirs_prep);
  const double4* const raw = reinterpret_cast<const double4*>(p.dirs);
  const int64_t ns_pad = s_hi;

  unsigned char* const wst = stage_lds + (tid >> 6) * kGradStageWaveBytes;
  auto stage_issue = [&](int64_t s0n, int half) {
    int ln = lane;
    asm volatile("" : "+v"(ln));
    const char* const grow = reinterpret_cast<const char*>(rows + (size_t)(s0n + (ln >> 4)) * CT) + (ln & 15) * 16;
    const int64_t sd = s0n + ((ln & 31) >> 3);
    const int64_t sr = BATCH ? dir0 + sd : sd;
    const char* const gdir = (ln < 32) ? reinterpret_cast<const char*>(prep + sd) + (ln & 7) * 4
                                       : reinterpret_cast<const char*>(raw + (sr < dir_last ? sr : dir_last)) + (ln & 7) * 4;
    __builtin_amdgcn_global_load_lds((gptr_t)grow, (lptr_t)(wst + half * 1280), 16, 0, 0);
    __builtin_amdgcn_global_load_lds((gptr_t)gdir, (lptr_t)(wst + half * 1280 + 1024), 4, 0, 0);
  };
  if (ns_pad > s_lo) stage_issue(s_lo, 0);
  for (int64_t s0 = s_lo; s0 < ns_pad; s0 += 4) {
    const int half = (int)(((s0 - s_lo) >> 2) & 1);
    __builtin_amdgcn_s_waitcnt(0x0F70);                                    // vmcnt(0): this group's two DMA loads have landed
    wave_lds_sync();
    const unsigned char* const hb = wst + half * 1280;
    const double4 sv = *reinterpret_cast<const double4*>(hb + 1024 + k * 32);
    const double4 rw = *reinterpret_cast<const double4*>(hb + 1024 + 128 + k * 32);
    const double2* const lrow = reinterpret_cast<const double2*>(hb + k * 256);       // this lane's source: 32 channels
    // the next group goes into the other half; this half is read until the end of the iteration and rewritten one iteration later
    if (s0 + 4 < ns_pad) stage_issue(s0 + 4, half ^ 1);
    const double cx = (x == 0) ? 1.0 : (x == 1 ? rw.x : (x == 2 ? rw.y : rw.z));
    const double d = __builtin_fma(bx, sv.x, __builtin_fma(by, sv.y, bz * sv.z));
    const double tau = d + bpc;
    double gq = sv.w * __builtin_fma(-tau, tau, bl2_c2);
    gq = __builtin_fmax(gq, 0.0);
    const ExpPhase ex = exp_tab_front(-gq * (fc * fc), etab);
    const TabPhase pc = sincos_tab_front(d * fcN, tab);
    const TabPhase ps = sincos_tab_front(d * dfN, tab);
    double zc, zs, rc, rs;
    sincos_tab_back(pc, zc, zs);
    sincos_tab_back(ps, rc, rs);
    // ln w_j = -g (f_0 + j df)^2: group t holds the ratio q_t = exp(-a - nu (16 t + 8)), a = 2 g df f_0, nu = g df^2 (skyvis_rec_body, GROUPED)
    const double a = gq * (2.0 * df * fc);
    const double u = gq * (df * df);
    double E, y8, cm[4];
    if (__builtin_amdgcn_ballot_w64(!(__builtin_fabs(a) < 0.1 && u < 3.0e-5)) == 0) {
      E = exp_series9(-a);
      const double X = __builtin_fma(u, __builtin_fma(u, __builtin_fma(u, 1.66666666666666666667e-01, 0.5), 1.0), 1.0);
      const double X2 = X * X, X3 = X2 * X, X5 = X3 * X2;
      cm[0] = X5 * X2;
      cm[1] = cm[0] * X5;
      cm[2] = cm[1] * X3;
      cm[3] = cm[2] * X;
      const double v = -8.0 * u;
      y8 = __builtin_fma(v, __builtin_fma(v, __builtin_fma(v, __builtin_fma(v, 4.16666666666666666667e-02, 1.66666666666666666667e-01), 0.5), 1.0), 1.0);
    } else {
      E = exp(-a);
      cm[0] = exp(7.0 * u); cm[1] = exp(12.0 * u); cm[2] = exp(15.0 * u); cm[3] = exp(16.0 * u);
      y8 = exp(-8.0 * u);
    }
    const double hg = y8 * y8;
    const double w0 = exp_tab_back(ex);
    const double q0 = E * y8;
    double ur = w0 * zc, ui = -(w0 * zs);            // zeta_0 = w_0 z_0, z = exp(-2 pi i phi)
    double dr = q0 * rc, di = -(q0 * rs);            // rho_0
#pragma unroll
    for (int t = 0; t < CT / 8; ++t) {
      double2 pr[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) pr[j] = lrow[4 * t + j];
      constexpr int kmap[8] = {0, 0, 1, 2, 3, 2, 1, 0};
#pragma unroll
      for (int m = 0; m < 8; ++m) {
        const int kk = 8 * t + m;
        const double pv = (m & 1) ? pr[m >> 1].y : pr[m >> 1].x;
        const double au = cx * pv;
        double br = ur, bi = ui;
        if (m != 0) { br *= cm[kmap[m]]; bi *= cm[kmap[m]]; }
        a_r[kk] = __builtin_amdgcn_mfma_f64_4x4x4f64(au, br, a_r[kk], 0, 0, 0);
        a_i[kk] = __builtin_amdgcn_mfma_f64_4x4x4f64(au, bi, a_i[kk], 0, 0, 0);
        if (!(t == CT / 8 - 1 && m == 7)) {
          const double t0 = ui * di, t1 = ur * di;
          const double nr = __builtin_fma(ur, dr, -t0);
          const double ni = __builtin_fma(ui, dr, t1);
          ur = nr; ui = ni;
        }
      }
      if (t + 1 < CT / 8) { dr *= hg; di *= hg; }
    }
  }
  if (b_valid) {
    const int i = k;
    double2* const dst = (i == 0) ? reinterpret_cast<double2*>(out_v) : reinterpret_cast<double2*>(out_g) + (size_t)(i - 1) * p.nbl * p.nchan;
    double2* const o = dst + (size_t)b * p.nchan;
#pragma unroll
    for (int j = 0; j < CT; ++j)
      if (k0 + j < p.nchan) o[k0 + j] = make_double2(a_r[j], a_i[j]);
  }
}

__global__ __launch_bounds__(kBlockThreads) __attribute__((amdgpu_waves_per_eu(2, 2)))
void k_skyvis_grad_taper_f64(const SkyvisParams p) {
  __shared__ double2 tab_lds[kTabN];
  __shared__ double etab_lds[kExpTabN];
  __shared__ __attribute__((aligned(16))) unsigned char stage_lds[(kBlockThreads / 64) * kGradStageWaveBytes];
  fill_phasor_table(tab_lds);
  fill_exp_table(etab_lds);
  __syncthreads();
  skyvis_grad_taper_f64_body<false>(p, tab_lds, etab_lds, stage_lds);
}

// V + baseline gradient of a whole chunk of snapshots of a small array in one launch (interferometry.py:6330, 6338, 6343 per snapshot)
__global__ __launch_bounds__(kBlockThreads) __attribute__((amdgpu_waves_per_eu(2, 2)))
void k_skyvis_grad_taper_f64_batch(const SkyvisParams p) {
  __shared__ double2 tab_lds[kTabN];
  __shared__ double etab_lds[kExpTabN];
  __shared__ __attribute__((aligned(16))) unsigned char stage_lds[(kBlockThreads / 64) * kGradStageWaveBytes];
  fill_phasor_table(tab_lds);
  fill_exp_table(etab_lds);
  __syncthreads();
  skyvis_grad_taper_f64_body<true>(p, tab_lds, etab_lds, stage_lds);
}

template <int CT, bool TAPER>
__global__ __launch_bounds__(kBlockThreads) __attribute__((amdgpu_waves_per_eu(PK_WAVES, PK_WAVES)))
void k_skyvis_rec_f32pk(const SkyvisParams p) {
  __shared__ __attribute__((aligned(16))) unsigned char flush_lds[flush_lds_bytes<float>() + kPrefetchLdsBytes];
  if constexpr (!TAPER) {
    // block-uniform choice made by the host per baseline group; the two bodies share no live state
    int slab_, bg;
    if (!block_item(p, slab_, bg)) return;
    if (p.lift_flags != nullptr && p.lift_flags[bg] != 0) {
      skyvis_rec_f32pk_body<CT, false, true>(p, flush_lds);
      return;
    }
  } else {
    // lift_flags[bg] = 1: |step angle| <= pi/4 for every source of this baseline group; 0: re-anchor the chains at their midpoint
    int slab_, bg;
    if (!block_item(p, slab_, bg)) return;
    const bool small_step = p.lift_flags != nullptr && p.lift_flags[bg] != 0;
    if (p.taper_group) {                      // launch-uniform, chosen by the host from df / f_min
      if (small_step) skyvis_rec_f32pk_body<CT, true, false, 1, 0>(p, flush_lds);
      else skyvis_rec_f32pk_body<CT, true, false, 1, 2>(p, flush_lds);
    } else {
      if (small_step) skyvis_rec_f32pk_body<CT, true, false, 0, 1>(p, flush_lds);
      else skyvis_rec_f32pk_body<CT, true, false, 0, 2>(p, flush_lds);
    }
    return;
  }
  skyvis_rec_f32pk_body<CT, TAPER, false>(p, flush_lds);
}

// Packed fp32 sky-sum of ONE source range whose sources share a size (kappa0): the split taper form (see skyvis_rec_f32pk_body).
// split_flags[bg]: bit 0 = |step angle| <= pi/4 guaranteed for the group (no re-anchoring), bit 1 = the uncorrected parabola could not be
// bounded below 2e-7 for this group: keep the correction, bit 2 = even four times the bound passes: groups of 16 steps.
template <int CT>
__global__ __launch_bounds__(kBlockThreads) __attribute__((amdgpu_waves_per_eu(PK_WAVES, PK_WAVES)))
void k_skyvis_rec_f32pk_split(const SkyvisParams p) {
  __shared__ __attribute__((aligned(16))) unsigned char flush_lds[flush_lds_bytes<float>() + kPrefetchLdsBytes + kBlockThreads * sizeof(double)];
  int slab_, bg;
  if (!block_item(p, slab_, bg)) return;
  const int fl = p.split_flags[bg];
  if (fl & 2) {
    if (fl & 1) skyvis_rec_f32pk_body<CT, true, false, 2, 0>(p, flush_lds);
    else skyvis_rec_f32pk_body<CT, true, false, 2, 2>(p, flush_lds);
  } else if (fl & 4) {
    if (fl & 1) skyvis_rec_f32pk_body<CT, true, false, 4, 0>(p, flush_lds);
    else skyvis_rec_f32pk_body<CT, true, false, 4, 2>(p, flush_lds);
  } else {
    if (fl & 1) skyvis_rec_f32pk_body<CT, true, false, 3, 0>(p, flush_lds);
    else skyvis_rec_f32pk_body<CT, true, false, 3, 2>(p, flush_lds);
  }
}

// Beam-weighted moments of a source range per channel, for the host's bound on the split taper's uncorrected parabola:
//   out[0][k] = sum_s p[s,k],  out[1][k] = sum_s p rho^2,  out[2][k] = sum_s p rho |n|,  out[3][k] = sum_s p n^2,   rho^2 = l^2 + m^2,
// (l, m, n) the source direction: (b.s)^2 <= (|b_h| rho + |b_z| |n|)^2.  |p| is summed (the tolerance is relative to sum|pbflux|).
// Deterministic: every block (a chunk of 1024 sources) writes its own partial sums, k_moments_reduce adds them in chunk order -- the
// bound decides which body a baseline group runs, and a sum whose order changes from run to run (atomics) could flip a group that sits near
// a limit, making fp32 results differ at 1e-7 of sum|pbflux| between runs of one input.
__global__ void k_taper_moments(const double* __restrict__ pb, const double* __restrict__ dirs, int64_t s_lo, int64_t s_hi, int64_t nchan,
                                double* __restrict__ part /*[gridDim.y][4][nchan]*/) {
  const int kc = threadIdx.x & 63, sl = threadIdx.x >> 6;                 // 64 channels x 4 source lanes
  const int64_t k = (int64_t)blockIdx.x * 64 + kc;
  const int64_t chunk = 1024;
  const int64_t s0 = s_lo + (int64_t)blockIdx.y * chunk;
  const int64_t s1 = (s0 + chunk < s_hi) ? s0 + chunk : s_hi;
  double m0 = 0.0, m1 = 0.0, m2 = 0.0, m3 = 0.0;
  if (k < nchan) {
    for (int64_t s = s0 + sl; s < s1; s += 4) {
      const double4 d = reinterpret_cast<const double4*>(dirs)[s];
      const double v = __builtin_fabs(pb[(size_t)s * nchan + k]);
      const double r2 = d.x * d.x + d.y * d.y, an = __builtin_fabs(d.z);
      m0 += v; m1 += v * r2; m2 += v * __builtin_sqrt(r2) * an; m3 += v * an * an;
    }
  }
  __shared__ double red[4][4][64];
  red[0][sl][kc] = m0; red[1][sl][kc] = m1; red[2][sl][kc] = m2; red[3][sl][kc] = m3;
  __syncthreads();
  if (sl == 0 && k < nchan) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const double v = red[q][0][kc] + red[q][1][kc] + red[q][2][kc] + red[q][3][kc];
      part[((size_t)blockIdx.y * 4 + q) * nchan + k] = v;
    }
  }
}

__global__ void k_moments_reduce(const double* __restrict__ part, int64_t nchunks, int64_t nchan, double* __restrict__ out /*[4][nchan]*/) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;          // (q, k)
  if (i >= 4 * nchan) return;
  double a = 0.0;
  for (int64_t c = 0; c < nchunks; ++c) a += part[(size_t)c * 4 * nchan + i];
  out[i] = a;
}

// Per-group body choice of the split taper kernel from the moments, ON THE DEVICE (no host round trip: a download here would make every
// snapshot's launch wait for the previous snapshot's sky-sum).  One block.  bound_g = c16 (H^2 M2 + 2 H Z M11 + Z^2 M02) with the
// moments' per-channel ratios maximised over the channels; flags[g] = (small step angle ? 1 : 0) | (bound_g <= limit ? 0 : 2) |
// (4 bound_g <= limit ? 4 : 0); *count += groups that run uncorrected.
__global__ void k_split_flags(const double* __restrict__ mom /*[4][nchan]*/, int64_t nchan, const double* __restrict__ grp_h,
                              const double* __restrict__ grp_z, const int32_t* __restrict__ lift_flags, int nbg, double c16, double limit,
                              int32_t* __restrict__ flags, int32_t* __restrict__ count) {
  __shared__ double sm[3][256];
  double a = 0.0, b = 0.0, c = 0.0;
  for (int64_t k = threadIdx.x; k < nchan; k += blockDim.x) {
    const double s0 = mom[k];
    if (s0 > 0.0) {
      a = fmax(a, mom[nchan + k] / s0); b = fmax(b, mom[2 * nchan + k] / s0); c = fmax(c, mom[3 * nchan + k] / s0);
    }
  }
  sm[0][threadIdx.x] = a; sm[1][threadIdx.x] = b; sm[2][threadIdx.x] = c;
  __syncthreads();
  for (int st = 128; st > 0; st >>= 1) {
    if ((int)threadIdx.x < st) {
      sm[0][threadIdx.x] = fmax(sm[0][threadIdx.x], sm[0][threadIdx.x + st]);
      sm[1][threadIdx.x] = fmax(sm[1][threadIdx.x], sm[1][threadIdx.x + st]);
      sm[2][threadIdx.x] = fmax(sm[2][threadIdx.x], sm[2][threadIdx.x + st]);
    }
    __syncthreads();
  }
  const double m2 = sm[0][0], m11 = sm[1][0], m02 = sm[2][0];
  int mine = 0;
  for (int g = threadIdx.x; g < nbg; g += blockDim.x) {
    const double H = grp_h[g], Z = grp_z[g];
    const double bound = c16 * (H * H * m2 + 2.0 * H * Z * m11 + Z * Z * m02);
    int32_t fl = lift_flags[g] ? 1 : 0;
    if (!(bound <= limit)) {
      fl |= 2;
    } else {
      ++mine;
      if (4.0 * bound <= limit) fl |= 4;              // the parabola of a 16-step group is 4x that of an 8-step one
    }
    flags[g] = fl;
  }
  if (mine) atomicAdd(count, mine);
}

// fp32 visibility + baseline gradient in one pass (GRAD bodies of the packed kernel, 16-channel tiles, no source split)
template <bool TAPER>
__global__ __launch_bounds__(kBlockThreads) __attribute__((amdgpu_waves_per_eu(PK_WAVES, PK_WAVES)))
void k_skyvis_grad_f32pk(const SkyvisParams p) {
  __shared__ __attribute__((aligned(16))) unsigned char flush_lds[flush_lds_bytes<float>() + kPrefetchLdsBytes];
  int slab_, bg;
  if (!block_item(p, slab_, bg)) return;
  const bool small_step = p.lift_flags != nullptr && p.lift_flags[bg] != 0;
  if constexpr (!TAPER) {
    // rows pre-multiplied by (1, l, m, n) (k_pack_grad): the GPK bodies
    if (small_step) skyvis_rec_f32pk_body<16, false, true, 0, 0, true, true>(p, flush_lds);
    else skyvis_rec_f32pk_body<16, false, false, 0, 0, true, true>(p, flush_lds);
  } else {
    // 8 steps per chain: one group of the grouped recurrence and no mid-chain re-anchoring (HC < 32); the REANCHOR = 2 bodies are the ones
    // that seed the step phasor for any step angle (groups without the |theta| <= 1/8 cycle guarantee)
    if (p.taper_group) {
      if (small_step) skyvis_rec_f32pk_body<16, true, false, 1, 0, true>(p, flush_lds);
      else skyvis_rec_f32pk_body<16, true, false, 1, 2, true>(p, flush_lds);
    } else {
      if (small_step) skyvis_rec_f32pk_body<16, true, false, 0, 0, true>(p, flush_lds);
      else skyvis_rec_f32pk_body<16, true, false, 0, 2, true>(p, flush_lds);
    }
  }
}

// ------------------------------------------------------------------------------------------
// Direct kernel: one (baseline, channel) output per thread, one sincospi per term, fp64 only.
// Works for any channel grid (no uniform-spacing assumption).  Slow; used as the on-device
// cross-check and as the fallback for non-uniform channel arrays.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlockThreads)
void k_skyvis_direct(const SkyvisParams p, const double* __restrict__ freqs,
                     const double* __restrict__ pb /* [nsrc][nchan] */, const double* __restrict__ scale) {
  // block = 64 baselines x 4 channels
  __shared__ double4 sd[64];
  __shared__ double sp[64][4];
  const int lane = threadIdx.x & 63;
  const int w = threadIdx.x >> 6;
  const int64_t nchan4 = (p.nchan + 3) / 4;
  const int64_t bgi = blockIdx.x / nchan4;
  const int64_t kq = blockIdx.x % nchan4;
  const int64_t b_raw = bgi * 64 + lane;
  const bool valid_b = b_raw < p.nbl;
  const int64_t b = valid_b ? b_raw : p.nbl - 1;
  const int64_t k_raw = kq * 4 + w;
  const bool valid_k = k_raw < p.nchan;
  const int64_t k = valid_k ? k_raw : p.nchan - 1;
  const double bx = p.bl_x[b], by = p.bl_y[b], bz = p.bl_z[b];
  const double f = freqs[k];
  const double bl2_c2 = (bx * bx + by * by + bz * bz) * (p.inv_c * p.inv_c);
  const double bpc = (bx * p.pc_x + by * p.pc_y + bz * p.pc_z) * p.inv_c;
  const double4* gd = reinterpret_cast<const double4*>(p.dirs);
  double are = 0.0, aim = 0.0;
  for (int64_t s0 = 0; s0 < p.nsrc; s0 += 64) {
    __syncthreads();
    if (threadIdx.x < 64) {
      const int64_t s = s0 + threadIdx.x;
      double4 v = make_double4(0, 0, 0, 0);
      if (s < p.nsrc) v = gd[s];
      sd[threadIdx.x] = make_double4((v.x - p.pc_x) * p.inv_c, (v.y - p.pc_y) * p.inv_c, (v.z - p.pc_z) * p.inv_c, v.w);
    }
    {
      const int ss = threadIdx.x >> 2, kk = threadIdx.x & 3;
      const int64_t s = s0 + ss;
      const int64_t kc = kq * 4 + kk;
      double v = 0.0;
      if (s < p.nsrc && kc < p.nchan) {
        v = pb[(size_t)s * p.nchan + kc];
        if (scale) v *= scale[(size_t)s * 4 + p.scale_comp];
      }
      sp[ss][kk] = v;
    }
    __syncthreads();
    const int ns = (int)((p.nsrc - s0) < 64 ? (p.nsrc - s0) : 64);
    for (int s = 0; s < ns; ++s) {
      const double4 sv = sd[s];
      const double d = __builtin_fma(bx, sv.x, __builtin_fma(by, sv.y, bz * sv.z));
      double ph = d * f;
      ph -= __builtin_rint(ph);
      double sn, cs;
      sincospi(2.0 * ph, &sn, &cs);
      double a = sp[s][w];
      if (p.taper) {
        const double tau = d + bpc;
        double gq = sv.w * (bl2_c2 - tau * tau);
        gq = gq > 0.0 ? gq : 0.0;
        a *= exp(-gq * f * f);
      }
      are = __builtin_fma(a, cs, are);
      aim = __builtin_fma(a, -sn, aim);
    }
  }
  if (valid_b && valid_k) {
    double2* o = reinterpret_cast<double2*>(p.out);
    o[(size_t)b * p.nchan + k] = make_double2(are, aim);
  }
}

// ------------------------------------------------------------------------------------------
// pack: pb[nsrc][nchan] (double) -> packed[ntiles][nsrc][CT] (T), optional per-source scale
// (gradient passes multiply the rows by a direction-cosine component, interferometry.py:6338).
// Channels beyond nchan in the last tile are zero-filled.
// ------------------------------------------------------------------------------------------
// interleave != 0: element 2j of a row holds channel HC+j, element 2j+1 channel HC-1-j (HC = ct/2),
// the (up, down) operand pairs of k_skyvis_rec_f32pk.
// Only the rows of sources [s_lo, s_hi) are written (the whole padded range: 0, nsrc_pad): a run of a mixed sky can be re-packed in the
// layout its kernel wants.
template <typename T>
__device__ __forceinline__ void pack_rows(const double* __restrict__ pb, T* __restrict__ packed, int64_t nsrc, int64_t nsrc_pad,
                                          int64_t nchan, int ct, int ntiles, const double* __restrict__ dirs, int scale_comp,
                                          int interleave, int64_t s_lo, int64_t s_hi, unsigned bid, unsigned nblocks) {
  const int64_t nrow = s_hi - s_lo;
  const int64_t total = (int64_t)ntiles * nrow * ct;
  for (int64_t i = (int64_t)bid * blockDim.x + threadIdx.x; i < total; i += (int64_t)nblocks * blockDim.x) {
    const int cpos = (int)(i % ct);
    int c = cpos;
    if (interleave) c = (c & 1) ? (ct / 2 - 1 - (c >> 1)) : (ct / 2 + (c >> 1));
    const int64_t s = s_lo + (i / ct) % nrow;
    const int tile = (int)(i / ((int64_t)ct * nrow));
    const int64_t k = (int64_t)tile * ct + c;
    double v = 0.0;
    if (k < nchan && s < nsrc) {
      v = pb[(size_t)s * nchan + k];
      if (scale_comp >= 0) v *= dirs[(size_t)s * 4 + scale_comp];
    }
    packed[((size_t)tile * nsrc_pad + (size_t)s) * ct + cpos] = (T)v;
  }
}

template <typename T>
__global__ void k_pack(const double* __restrict__ pb, T* __restrict__ packed, int64_t nsrc, int64_t nsrc_pad,
                       int64_t nchan, int ct, int ntiles, const double* __restrict__ dirs, int scale_comp,
                       int interleave, int64_t s_lo, int64_t s_hi) {
  pack_rows<T>(pb, packed, nsrc, nsrc_pad, nchan, ct, ntiles, dirs, scale_comp, interleave, s_lo, s_hi, blockIdx.x, gridDim.x);
}

// Rows of the fused fp32 gradient kernel without the taper (GPK bodies, 16-channel tiles): packed[tile][s][pair j][set r][up / down] =
// pb[s][k] c_r(s), c = (1, l, m, n), k = tile 16 + 8 + j (up) / tile 16 + 7 - j (down); 64 floats per (source, tile); zero rows past nsrc.
__global__ void k_pack_grad(const double* __restrict__ pb, float* __restrict__ packed, int64_t nsrc, int64_t nsrc_pad, int64_t nchan, int ntiles,
                            const double* __restrict__ dirs) {
  const int64_t total = (int64_t)ntiles * nsrc_pad * 64;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int e = (int)(i & 63);
    const int j = e >> 3, rs = (e >> 1) & 3, ud = e & 1;
    const int64_t s = (i >> 6) % nsrc_pad;
    const int tile = (int)((i >> 6) / nsrc_pad);
    const int64_t k = (int64_t)tile * 16 + (ud ? 7 - j : 8 + j);
    double v = 0.0;
    if (k < nchan && s < nsrc) {
      v = pb[(size_t)s * nchan + k];
      if (rs > 0) v *= dirs[(size_t)s * 4 + (rs - 1)];
    }
    packed[i] = (float)v;
  }
}

// dirs_prep[s] = ((l,m,n) - s_pc)/c, kappa   for s < nsrc;  zeros for nsrc <= s < nsrc_pad
// c32 (optional): [nsrc_pad][8] floats (l, l, m, m, n, n, 0, 0): the gradient coefficients as ready SGPR-pair operands
__device__ __forceinline__ void prep_dirs(const double* __restrict__ dirs, double* __restrict__ prep, float* __restrict__ c32, int64_t nsrc,
                                          int64_t nsrc_pad, double pcx, double pcy, double pcz, double inv_c, unsigned bid, unsigned nblocks) {
  for (int64_t s = (int64_t)bid * blockDim.x + threadIdx.x; s < nsrc_pad; s += (int64_t)nblocks * blockDim.x) {
    double4 v = make_double4(0, 0, 0, 0);
    float4 c0 = make_float4(0.f, 0.f, 0.f, 0.f), c1 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (s < nsrc) {
      const double4 r = reinterpret_cast<const double4*>(dirs)[s];
      v = make_double4((r.x - pcx) * inv_c, (r.y - pcy) * inv_c, (r.z - pcz) * inv_c, r.w);
      c0 = make_float4((float)r.x, (float)r.x, (float)r.y, (float)r.y);
      c1 = make_float4((float)r.z, (float)r.z, 0.f, 0.f);
    }
    reinterpret_cast<double4*>(prep)[s] = v;
    if (c32) {
      reinterpret_cast<float4*>(c32)[2 * s] = c0;
      reinterpret_cast<float4*>(c32)[2 * s + 1] = c1;
    }
  }
}

__global__ void k_prep_dirs(const double* __restrict__ dirs, double* __restrict__ prep, float* __restrict__ c32, int64_t nsrc,
                            int64_t nsrc_pad, double pcx, double pcy, double pcz, double inv_c) {
  prep_dirs(dirs, prep, c32, nsrc, nsrc_pad, pcx, pcy, pcz, inv_c, blockIdx.x, gridDim.x);
}

// Both per-snapshot pre-passes in ONE launch (the first gpack blocks pack the rows, the rest prepare the directions; neither reads what
// the other writes): a launch and its gap are ~6 us of a small problem's snapshot (config 2: 81 us).
template <typename T>
__global__ void k_pack_prep(const double* __restrict__ pb, T* __restrict__ packed, int64_t nsrc, int64_t nsrc_pad, int64_t nchan, int ct,
                            int ntiles, int interleave, const double* __restrict__ dirs, double* __restrict__ prep, double pcx, double pcy,
                            double pcz, double inv_c, unsigned gpack) {
  if (blockIdx.x < gpack) pack_rows<T>(pb, packed, nsrc, nsrc_pad, nchan, ct, ntiles, nullptr, -1, interleave, 0, nsrc_pad, blockIdx.x, gpack);
  else prep_dirs(dirs, prep, nullptr, nsrc, nsrc_pad, pcx, pcy, pcz, inv_c, blockIdx.x - gpack, gridDim.x - gpack);
}

// The same two pre-passes for a BATCH of snapshots in one launch (blockIdx.y = snapshot): rows of snapshot t go to rows
// [row0, row0 + nrow) of every tile's slab (pitch = all snapshots' rows), its beam x flux block starts at row pb0 of pb, its
// directions at row dir0 of dirs; rows past nsrc are zero.  Natural channel order (the grouped fp64 taper kernel's layout).
__global__ void k_pack_prep_batch(const double* __restrict__ pb, double* __restrict__ packed, int64_t pitch, int64_t nchan, int ct, int ntiles,
                                  const double* __restrict__ dirs, double* __restrict__ prep, double inv_c, unsigned gpack,
                                  const BatchSnap* __restrict__ snaps) {
  const BatchSnap sn = snaps[blockIdx.y];
  if (blockIdx.x < gpack) {
    const int64_t total = (int64_t)ntiles * sn.nrow * ct;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gpack * blockDim.x) {
      const int c = (int)(i % ct);
      const int64_t s = (i / ct) % sn.nrow;
      const int tile = (int)(i / ((int64_t)ct * sn.nrow));
      const int64_t k = (int64_t)tile * ct + c;
      double v = 0.0;
      if (k < nchan && s < sn.nsrc) v = pb[(size_t)(sn.pb0 + s) * nchan + k];
      packed[((size_t)tile * pitch + (size_t)(sn.row0 + s)) * ct + c] = v;
    }
  } else {
    const unsigned bid = blockIdx.x - gpack, nb = gridDim.x - gpack;
    for (int64_t s = (int64_t)bid * blockDim.x + threadIdx.x; s < sn.nrow; s += (int64_t)nb * blockDim.x) {
      double4 v = make_double4(0, 0, 0, 0);
      if (s < sn.nsrc) {
        const double4 r = reinterpret_cast<const double4*>(dirs)[sn.dir0 + s];
        v = make_double4((r.x - sn.pc[0]) * inv_c, (r.y - sn.pc[1]) * inv_c, (r.z - sn.pc[2]) * inv_c, r.w);
      }
      reinterpret_cast<double4*>(prep)[sn.row0 + s] = v;
    }
  }
}

// k_reduce_partials for a batch: blockIdx.y = snapshot, its nsplit partial cubes -> its (consecutive) cube slot
__global__ void k_reduce_partials_batch(const double* __restrict__ part, double* __restrict__ out, int64_t n2, int nsplit) {
  const double* pp = part + (size_t)blockIdx.y * (size_t)nsplit * (size_t)n2;
  double* oo = out + (size_t)blockIdx.y * (size_t)n2;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n2; i += (int64_t)gridDim.x * blockDim.x) {
    double a = 0.0;
    for (int sp = 0; sp < nsplit; ++sp) a += pp[(size_t)sp * n2 + i];
    oo[i] = a;
  }
}

// sum nsplit partial cubes [nsplit][n] (complex as 2 doubles, or 2 floats for the fp32 kernels' single-flush partials) -> out[n];
// deterministic order, fp64 sum.
template <typename TP>
__global__ void k_reduce_partials(const TP* __restrict__ part, double* __restrict__ out, int64_t n2, int nsplit) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n2; i += (int64_t)gridDim.x * blockDim.x) {
    double a = 0.0;
    for (int sp = 0; sp < nsplit; ++sp) a += (double)part[(size_t)sp * n2 + i];
    out[i] = a;
  }
}

__global__ void k_f32_to_f64(const float* __restrict__ in, double* __restrict__ out, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    out[i] = (double)in[i];
}

__global__ void k_fsq(const double* __restrict__ freqs, float* __restrict__ fsq, int64_t nchan, int64_t npad,
                      double scale) {
  // fsq[k] = (f_k * scale)^2 * log2(e) in fp32; scale keeps g*fsq in fp32 range
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < npad; i += (int64_t)gridDim.x * blockDim.x) {
    double f = freqs[i < nchan ? i : nchan - 1] * scale;
    fsq[i] = (float)(f * f * 1.4426950408889634);
  }
}

// ------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------
template <typename T, int CT>
static hipError_t launch_rec_ct(const SkyvisParams& p, hipStream_t stream) {
  const int64_t items = (int64_t)p.ntiles * p.nsplit * p.nbgroups;          // see block_item()
  if (items <= 0 || items > 0x3fffffffLL) return hipErrorInvalidValue;
  const unsigned grid = 8u * (unsigned)((items + 7) / 8);
  if (p.taper)
    hipLaunchKernelGGL((k_skyvis_rec<T, CT, true>), dim3(grid), dim3(kBlockThreads), 0, stream, p);
  else
    hipLaunchKernelGGL((k_skyvis_rec<T, CT, false>), dim3(grid), dim3(kBlockThreads), 0, stream, p);
  return hipGetLastError();
}

template <int CT>
static hipError_t launch_rec_pk_ct(const SkyvisParams& p, hipStream_t stream) {
  const int64_t items = (int64_t)p.ntiles * p.nsplit * p.nbgroups;          // see block_item()
  if (items <= 0 || items > 0x3fffffffLL) return hipErrorInvalidValue;
  const unsigned grid = 8u * (unsigned)((items + 7) / 8);
  if (p.taper)
    hipLaunchKernelGGL((k_skyvis_rec_f32pk<CT, true>), dim3(grid), dim3(kBlockThreads), 0, stream, p);
  else
    hipLaunchKernelGGL((k_skyvis_rec_f32pk<CT, false>), dim3(grid), dim3(kBlockThreads), 0, stream, p);
  return hipGetLastError();
}

hipError_t launch_skyvis_rec_f32pk(const SkyvisParams& p, int ct, hipStream_t stream) {
  switch (ct) {
    case 32: return launch_rec_pk_ct<32>(p, stream);
    case 64: return launch_rec_pk_ct<64>(p, stream);
  }
  return hipErrorInvalidValue;
}

hipError_t launch_skyvis_rec_f32pk_split(const SkyvisParams& p, int ct, hipStream_t stream) {
  const int64_t items = (int64_t)p.ntiles * p.nsplit * p.nbgroups;
  if (items <= 0 || items > 0x3fffffffLL || !p.split_flags || ct != 64) return hipErrorInvalidValue;
  const unsigned grid = 8u * (unsigned)((items + 7) / 8);
  hipLaunchKernelGGL((k_skyvis_rec_f32pk_split<64>), dim3(grid), dim3(kBlockThreads), 0, stream, p);
  return hipGetLastError();
}

int64_t taper_moments_chunks(int64_t s_lo, int64_t s_hi) { return (s_hi - s_lo + 1023) / 1024; }

hipError_t launch_taper_moments(const double* pb, const double* dirs, int64_t s_lo, int64_t s_hi, int64_t nchan, double* part, double* out,
                                hipStream_t stream) {
  if (s_hi <= s_lo || nchan <= 0 || !part) return hipErrorInvalidValue;
  const int64_t gy = taper_moments_chunks(s_lo, s_hi);
  if (gy > 65535) return hipErrorInvalidValue;
  hipLaunchKernelGGL(k_taper_moments, dim3((unsigned)((nchan + 63) / 64), (unsigned)gy), dim3(256), 0, stream, pb, dirs, s_lo, s_hi, nchan, part);
  hipLaunchKernelGGL(k_moments_reduce, dim3((unsigned)((4 * nchan + 255) / 256)), dim3(256), 0, stream, part, gy, nchan, out);
  return hipGetLastError();
}

hipError_t launch_split_flags(const double* mom, int64_t nchan, const double* grp_h, const double* grp_z, const int32_t* lift_flags, int nbg,
                              double c16, double limit, int32_t* flags, int32_t* count, hipStream_t stream) {
  if (nbg <= 0 || nchan <= 0) return hipErrorInvalidValue;
  hipLaunchKernelGGL(k_split_flags, dim3(1), dim3(256), 0, stream, mom, nchan, grp_h, grp_z, lift_flags, nbg, c16, limit, flags, count);
  return hipGetLastError();
}

hipError_t launch_skyvis_rec(const SkyvisParams& p, bool f32, int ct, hipStream_t stream) {
  if (f32) {
    switch (ct) {
      case 8: return launch_rec_ct<float, 8>(p, stream);
      case 16: return launch_rec_ct<float, 16>(p, stream);
      case 32: case 64: return launch_skyvis_rec_f32pk(p, ct, stream);   // packed kernel owns the wide tiles
    }
  } else {
    switch (ct) {
      case 8: return launch_rec_ct<double, 8>(p, stream);
      case 16: return launch_rec_ct<double, 16>(p, stream);
      case 32: return launch_rec_ct<double, 32>(p, stream);
    }
  }
  return hipErrorInvalidValue;
}

hipError_t launch_skyvis_taper_f64(const SkyvisParams& p, int ct, hipStream_t stream) {
  const int64_t items = (int64_t)p.ntiles * p.nsplit * p.nbgroups;          // see block_item()
  if (items <= 0 || items > 0x3fffffffLL || !p.taper) return hipErrorInvalidValue;
  const unsigned grid = 8u * (unsigned)((items + 7) / 8);
  if (p.wave_nbw > 0) {
    // wave items: one baseline group, p.nbgroups quads of (baseline wave, split) items
    if (p.nsplit != 1 || p.wave_nsplit < 1 || p.nbl > kBlockThreads || (int64_t)p.wave_nbw * 64 < p.nbl ||
        (int64_t)p.nbgroups * (kBlockThreads / 64) < (int64_t)p.wave_nbw * p.wave_nsplit)
      return hipErrorInvalidValue;
    switch (ct) {
      case 16: hipLaunchKernelGGL((k_skyvis_taper_f64_wave<16>), dim3(grid), dim3(kBlockThreads), 0, stream, p); break;
      case 32: hipLaunchKernelGGL((k_skyvis_taper_f64_wave<32>), dim3(grid), dim3(kBlockThreads), 0, stream, p); break;
      default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
  }
  switch (ct) {
    case 16: hipLaunchKernelGGL((k_skyvis_taper_f64<16>), dim3(grid), dim3(kBlockThreads), 0, stream, p); break;
    case 32: hipLaunchKernelGGL((k_skyvis_taper_f64<32>), dim3(grid), dim3(kBlockThreads), 0, stream, p); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

// many snapshots in one launch: p.wave_snaps / p.wave_nsnap set, p.nbgroups = ceil(nsnap nbw wave_nsplit / 4), p.nsplit = 1
hipError_t launch_skyvis_taper_f64_wave_batch(const SkyvisParams& p, int ct, hipStream_t stream) {
  const int64_t items = (int64_t)p.ntiles * p.nbgroups;
  if (items <= 0 || items > 0x3fffffffLL || !p.taper || p.nsplit != 1 || p.wave_nsplit < 1 || p.wave_nsnap < 1 || !p.wave_snaps || p.nbl > kBlockThreads ||
      (int64_t)p.wave_nbw * 64 < p.nbl || (int64_t)p.nbgroups * (kBlockThreads / 64) < (int64_t)p.wave_nsnap * p.wave_nbw * p.wave_nsplit)
    return hipErrorInvalidValue;
  const unsigned grid = 8u * (unsigned)((items + 7) / 8);
  switch (ct) {
    case 16: hipLaunchKernelGGL((k_skyvis_taper_f64_wave_batch<16>), dim3(grid), dim3(kBlockThreads), 0, stream, p); break;
    case 32: hipLaunchKernelGGL((k_skyvis_taper_f64_wave_batch<32>), dim3(grid), dim3(kBlockThreads), 0, stream, p); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

// V + gradient of many snapshots in one launch: p.wave_nbw = waves of 16 baselines, p.nbgroups = ceil(nsnap wave_nbw wave_nsplit / 4), 32-channel tiles
hipError_t launch_skyvis_grad_taper_f64_batch(const SkyvisParams& p, hipStream_t stream) {
  const int64_t items = (int64_t)p.ntiles * p.nbgroups;
  if (items <= 0 || items > 0x3fffffffLL || !p.taper || p.nsplit != 1 || p.wave_nsplit < 1 || p.wave_nsnap < 1 || !p.wave_snaps || p.nbl > kBlockThreads ||
      (int64_t)p.wave_nbw * 16 < p.nbl || (int64_t)p.nbgroups * (kBlockThreads / 64) < (int64_t)p.wave_nsnap * p.wave_nbw * p.wave_nsplit)
    return hipErrorInvalidValue;
  const unsigned grid = 8u * (unsigned)((items + 7) / 8);
  hipLaunchKernelGGL(k_skyvis_grad_taper_f64_batch, dim3(grid), dim3(kBlockThreads), 0, stream, p);
  return hipGetLastError();
}

hipError_t launch_pack_prep_batch(const double* pb, double* packed, int64_t pitch, int64_t max_nrow, int64_t nchan, int ct, int ntiles, const double* dirs,
                                  double* prep, double inv_c, const BatchSnap* snaps, int nsnap, hipStream_t stream) {
  if (nsnap <= 0 || max_nrow <= 0) return hipSuccess;
  int64_t gp = ((int64_t)ntiles * max_nrow * ct + 255) / 256, gd = (max_nrow + 255) / 256;
  if (gp > 2048) gp = 2048;
  if (gd > 64) gd = 64;
  hipLaunchKernelGGL(k_pack_prep_batch, dim3((unsigned)(gp + gd), (unsigned)nsnap), dim3(256), 0, stream, pb, packed, pitch, nchan, ct, ntiles, dirs, prep,
                     inv_c, (unsigned)gp, snaps);
  return hipGetLastError();
}

hipError_t launch_reduce_partials_batch(const double* part, double* out, int64_t n2, int nsplit, int nsnap, hipStream_t stream) {
  if (nsnap <= 0) return hipSuccess;
  int64_t g = (n2 + 255) / 256;
  if (g > 4096) g = 4096;
  hipLaunchKernelGGL(k_reduce_partials_batch, dim3((unsigned)g, (unsigned)nsnap), dim3(256), 0, stream, part, out, n2, nsplit);
  return hipGetLastError();
}

hipError_t launch_skyvis_grad_f32(const SkyvisParams& p, hipStream_t stream) {
  // p.nbgroups = groups of 256 baselines, p.nsplit = 1, 16-channel tiles, p.dirs_c32 and p.grad_out set
  const int64_t items = (int64_t)p.ntiles * p.nbgroups;
  if (items <= 0 || items > 0x3fffffffLL || p.nsplit != 1 || (p.taper && !p.dirs_c32) || !p.grad_out) return hipErrorInvalidValue;
  const unsigned grid = 8u * (unsigned)((items + 7) / 8);
  if (p.taper) hipLaunchKernelGGL((k_skyvis_grad_f32pk<true>), dim3(grid), dim3(kBlockThreads), 0, stream, p);
  else hipLaunchKernelGGL((k_skyvis_grad_f32pk<false>), dim3(grid), dim3(kBlockThreads), 0, stream, p);
  return hipGetLastError();
}

hipError_t launch_skyvis_grad_f64(const SkyvisParams& p, int ct, hipStream_t stream) {
  // p.nbgroups = groups of 64 baselines, p.nsplit = 1
  const int64_t items = (int64_t)p.ntiles * p.nbgroups;
  if (items <= 0 || items > 0x3fffffffLL || p.nsplit != 1) return hipErrorInvalidValue;
  const unsigned grid = 8u * (unsigned)((items + 7) / 8);
  if (ct == 32 && p.taper) {            // grouped single-chain form, rows in natural channel order (launch_pack interleave = 0)
    hipLaunchKernelGGL(k_skyvis_grad_taper_f64, dim3(grid), dim3(kBlockThreads), 0, stream, p);
  } else if (ct == 32) {                // (the exact taper recurrence's per-lane state does not fit beside 128 accumulator VGPRs at 32 channels)
    hipLaunchKernelGGL((k_skyvis_grad_f64<32, false>), dim3(grid), dim3(kBlockThreads), 0, stream, p);
  } else if (ct == 16) {
    if (p.taper) hipLaunchKernelGGL((k_skyvis_grad_f64<16, true>), dim3(grid), dim3(kBlockThreads), 0, stream, p);
    else hipLaunchKernelGGL((k_skyvis_grad_f64<16, false>), dim3(grid), dim3(kBlockThreads), 0, stream, p);
  } else {
    return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

hipError_t launch_skyvis_direct(const SkyvisParams& p, const double* freqs, const double* pb, const double* scale,
                                hipStream_t stream) {
  const int64_t nchan4 = (p.nchan + 3) / 4;
  const int64_t nbg = (p.nbl + 63) / 64;
  const int64_t grid = nbg * nchan4;
  if (grid <= 0 || grid > 0x7fffffffLL) return hipErrorInvalidValue;
  hipLaunchKernelGGL(k_skyvis_direct, dim3((unsigned)grid), dim3(kBlockThreads), 0, stream, p, freqs, pb, scale);
  return hipGetLastError();
}

static unsigned grid_for(int64_t n) {
  int64_t g = (n + 255) / 256;
  if (g > 8192) g = 8192;
  if (g < 1) g = 1;
  return (unsigned)g;
}

hipError_t launch_pack(const double* pb, void* packed, bool f32, int64_t nsrc, int64_t nsrc_pad, int64_t nchan, int ct,
                       int ntiles, const double* dirs, int scale_comp, int interleave, hipStream_t stream, int64_t s_lo, int64_t s_hi) {
  if (s_hi < 0) { s_lo = 0; s_hi = nsrc_pad; }
  if (s_lo < 0 || s_hi > nsrc_pad || s_hi < s_lo) return hipErrorInvalidValue;
  const int64_t total = (int64_t)ntiles * (s_hi - s_lo) * ct;
  if (total == 0) return hipSuccess;
  if (f32)
    hipLaunchKernelGGL(k_pack<float>, dim3(grid_for(total)), dim3(256), 0, stream, pb, (float*)packed, nsrc, nsrc_pad,
                       nchan, ct, ntiles, dirs, scale_comp, interleave, s_lo, s_hi);
  else
    hipLaunchKernelGGL(k_pack<double>, dim3(grid_for(total)), dim3(256), 0, stream, pb, (double*)packed, nsrc, nsrc_pad,
                       nchan, ct, ntiles, dirs, scale_comp, interleave, s_lo, s_hi);
  return hipGetLastError();
}

hipError_t launch_pack_prep(const double* pb, void* packed, bool f32, int64_t nsrc, int64_t nsrc_pad, int64_t nchan, int ct, int ntiles,
                            int interleave, const double* dirs, double* prep, double pcx, double pcy, double pcz, double inv_c,
                            hipStream_t stream) {
  const int64_t total = (int64_t)ntiles * nsrc_pad * ct;
  if (total == 0) return hipSuccess;
  const unsigned gpack = grid_for(total), gprep = grid_for(nsrc_pad);
  if (f32)
    hipLaunchKernelGGL(k_pack_prep<float>, dim3(gpack + gprep), dim3(256), 0, stream, pb, (float*)packed, nsrc, nsrc_pad, nchan, ct, ntiles,
                       interleave, dirs, prep, pcx, pcy, pcz, inv_c, gpack);
  else
    hipLaunchKernelGGL(k_pack_prep<double>, dim3(gpack + gprep), dim3(256), 0, stream, pb, (double*)packed, nsrc, nsrc_pad, nchan, ct, ntiles,
                       interleave, dirs, prep, pcx, pcy, pcz, inv_c, gpack);
  return hipGetLastError();
}

hipError_t launch_pack_grad(const double* pb, float* packed, int64_t nsrc, int64_t nsrc_pad, int64_t nchan, int ntiles, const double* dirs,
                            hipStream_t stream) {
  const int64_t total = (int64_t)ntiles * nsrc_pad * 64;
  if (total == 0) return hipSuccess;
  hipLaunchKernelGGL(k_pack_grad, dim3(grid_for(total)), dim3(256), 0, stream, pb, packed, nsrc, nsrc_pad, nchan, ntiles, dirs);
  return hipGetLastError();
}

hipError_t launch_reduce_partials(const void* part, bool part_f32, double* out, int64_t n2, int nsplit, hipStream_t stream) {
  if (part_f32)
    hipLaunchKernelGGL(k_reduce_partials<float>, dim3(grid_for(n2)), dim3(256), 0, stream, (const float*)part, out, n2, nsplit);
  else
    hipLaunchKernelGGL(k_reduce_partials<double>, dim3(grid_for(n2)), dim3(256), 0, stream, (const double*)part, out, n2, nsplit);
  return hipGetLastError();
}

hipError_t launch_f32_to_f64(const float* in, double* out, int64_t n, hipStream_t stream) {
  if (n == 0) return hipSuccess;
  hipLaunchKernelGGL(k_f32_to_f64, dim3(grid_for(n)), dim3(256), 0, stream, in, out, n);
  return hipGetLastError();
}

// fsq_pairs[tile][2j], [2j+1] = fsq of channels tile*ct+HC+j, tile*ct+HC-1-j (operand order of the packed kernel)
__global__ void k_fsq_pairs(const float* __restrict__ fsq, float* __restrict__ pairs, int ct, int ntiles) {
  const int total = ct * ntiles;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const int tile = i / ct, c = i % ct;
    const int k = (c & 1) ? (ct / 2 - 1 - (c >> 1)) : (ct / 2 + (c >> 1));
    pairs[i] = fsq[tile * ct + k];
  }
}

hipError_t launch_fsq_pairs(const float* fsq, float* pairs, int ct, int ntiles, hipStream_t stream) {
  hipLaunchKernelGGL(k_fsq_pairs, dim3(grid_for((int64_t)ct * ntiles)), dim3(256), 0, stream, fsq, pairs, ct, ntiles);
  return hipGetLastError();
}

hipError_t launch_prep_dirs(const double* dirs, double* prep, float* c32, int64_t nsrc, int64_t nsrc_pad, double pcx, double pcy,
                            double pcz, double inv_c, hipStream_t stream) {
  if (nsrc_pad == 0) return hipSuccess;
  hipLaunchKernelGGL(k_prep_dirs, dim3(grid_for(nsrc_pad)), dim3(256), 0, stream, dirs, prep, c32, nsrc, nsrc_pad, pcx, pcy,
                     pcz, inv_c);
  return hipGetLastError();
}

hipError_t launch_fsq(const double* freqs, float* fsq, int64_t nchan, int64_t npad, double scale, hipStream_t stream) {
  hipLaunchKernelGGL(k_fsq, dim3(grid_for(npad)), dim3(256), 0, stream, freqs, fsq, nchan, npad, scale);
  return hipGetLastError();
}

}  // namespace prisim
