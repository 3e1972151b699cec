// delay_kernels.hip -- fused delay transform for channel counts 256 R, R in {1, 2, 3, 4, 8, 16} (gfx950).
//
// Reference path replaced: prisim/interferometry.py:8114-8134 (InterferometerArray.delay_transform):
//   x = V * bp * bp_wts  ->  zero-pad to N' = N (1 + pad)  ->  ifft * N' * df  ->  fftshift  ->  keep every (1 + pad)-th lag.
// For an integer 1 + pad = F and even N the kept samples are   out[j] = df * sum_n x[n] exp(+2 pi i n k / N),  k = (j + N/2) mod N
// (index F (j + N/2) mod (N F) of the padded transform is index (j + N/2) mod N of the unpadded one): the zero padding and the
// F-fold longer FFT are pure overhead.  The rocFFT pipeline (k_dt_prepare -> rocFFT -> k_dt_finish) moves
// 16 + 32 + 64 + 32 + 16 = 160 bytes per visibility through HBM for pad = 1; this kernel reads each visibility once and writes each
// lag once (32 bytes, 24 with power only), which is the algorithmic minimum of a stage that is HBM-bound by nature
// (5 N log2 N flops against 32 N bytes: 1.6 flop/byte at N = 1024).
//
// One row (baseline, snapshot) = N complex128 = N/16 threads x 16 points, three register stages with two LDS exchanges:
//   N = 16 * 16 * R,  R = N / 256 in {1, 2, 3, 4, 8, 16}   (768 = the MWA's 24 x 32 fine channels)
//   A: thread j: 16-point DFT over n = j + (N/16) q, twiddle W_N^(j p), to LDS [p][j]
//   B: thread (p, a): 16-point DFT over j = a + R q', twiddle W_N^(16 a p'), to LDS [a][p + 16 p']
//   C: thread L: R-point DFTs over a for c = p + 16 p' = L + (N/16) m; output index k = c + 256 r
// Both exchanges are laid out so that every group of 8 lanes of a 16-byte access hits 8 different bank slots (pitches M + R, 256 + 8/R).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "skyvis_kernels.h"

namespace prisim {

namespace {

__device__ __forceinline__ double2 cadd(double2 a, double2 b) { return make_double2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ double2 csub(double2 a, double2 b) { return make_double2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ double2 cmul(double2 a, double wr, double wi) {
  return make_double2(__builtin_fma(a.x, wr, -(a.y * wi)), __builtin_fma(a.x, wi, a.y * wr));
}
__device__ __forceinline__ double2 cmul(double2 a, double2 w) { return cmul(a, w.x, w.y); }
__device__ __forceinline__ double2 mul_i(double2 a) { return make_double2(-a.y, a.x); }     // a * (+i)

// W_16^m = exp(+2 pi i m / 16), m = 0 ... 9 (all the products j p with j, p < 4 that the 4 x 4 split of the 16-point DFT needs)
__device__ __forceinline__ double2 tw16(int m) {
  constexpr double c = 0.92387953251128673848, s = 0.38268343236508978178, r = 0.70710678118654752440;
  switch (m) {
    case 0: return make_double2(1.0, 0.0);
    case 1: return make_double2(c, s);
    case 2: return make_double2(r, r);
    case 3: return make_double2(s, c);
    case 4: return make_double2(0.0, 1.0);
    case 5: return make_double2(-s, c);
    case 6: return make_double2(-r, r);
    case 7: return make_double2(-c, s);
    case 8: return make_double2(-1.0, 0.0);
    default: return make_double2(-c, -s);          // 9
  }
}

// In-register DFTs with the inverse sign, X[k] = sum_n v[n] exp(+2 pi i n k / R), natural order in and out.
__device__ __forceinline__ void dft2(double2& a, double2& b) {
  const double2 t = a;
  a = cadd(t, b);
  b = csub(t, b);
}
__device__ __forceinline__ void dft4(double2& v0, double2& v1, double2& v2, double2& v3) {
  const double2 t0 = cadd(v0, v2), t1 = csub(v0, v2), t2 = cadd(v1, v3), t3 = mul_i(csub(v1, v3));
  v0 = cadd(t0, t2);
  v1 = cadd(t1, t3);
  v2 = csub(t0, t2);
  v3 = csub(t1, t3);
}

template <int R> __device__ __forceinline__ void dft_small(double2 (&v)[16]);
template <> __device__ __forceinline__ void dft_small<2>(double2 (&v)[16]) { dft2(v[0], v[1]); }
template <> __device__ __forceinline__ void dft_small<3>(double2 (&v)[16]) {
  // w = exp(+2 pi i / 3) = (-1/2, +sqrt(3)/2):  X1 = a + w b + w^2 c,  X2 = a + w^2 b + w c
  const double2 a = v[0], sbc = cadd(v[1], v[2]), dbc = csub(v[1], v[2]);
  const double2 m = make_double2(__builtin_fma(-0.5, sbc.x, a.x), __builtin_fma(-0.5, sbc.y, a.y));       // a - (b + c)/2
  const double2 q = make_double2(-0.86602540378443864676 * dbc.y, 0.86602540378443864676 * dbc.x);          // i sqrt(3)/2 (b - c)
  v[0] = cadd(a, sbc);
  v[1] = cadd(m, q);
  v[2] = csub(m, q);
}
template <> __device__ __forceinline__ void dft_small<4>(double2 (&v)[16]) { dft4(v[0], v[1], v[2], v[3]); }
template <> __device__ __forceinline__ void dft_small<8>(double2 (&v)[16]) {
  // n = j + 2 q (j < 2, q < 4), k = p + 4 r:  X[p + 4 r] = sum_j W_8^(j p) W_2^(j r) [ sum_q v[j + 2 q] W_4^(q p) ]
  dft4(v[0], v[2], v[4], v[6]);                      // j = 0: y0[p] in v[2 p]
  dft4(v[1], v[3], v[5], v[7]);                      // j = 1: y1[p] in v[2 p + 1]
#pragma unroll
  for (int p = 1; p < 4; ++p) v[2 * p + 1] = cmul(v[2 * p + 1], tw16(2 * p));
  double2 o[8];
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    o[p] = cadd(v[2 * p], v[2 * p + 1]);             // r = 0
    o[p + 4] = csub(v[2 * p], v[2 * p + 1]);         // r = 1
  }
#pragma unroll
  for (int k = 0; k < 8; ++k) v[k] = o[k];
}
template <> __device__ __forceinline__ void dft_small<16>(double2 (&v)[16]) {
  // n = j + 4 q, k = p + 4 r:  X[p + 4 r] = sum_j W_16^(j p) W_4^(j r) [ sum_q v[j + 4 q] W_4^(q p) ]
#pragma unroll
  for (int j = 0; j < 4; ++j) dft4(v[j], v[j + 4], v[j + 8], v[j + 12]);       // y_j[p] in v[j + 4 p]
#pragma unroll
  for (int j = 1; j < 4; ++j)
#pragma unroll
    for (int p = 1; p < 4; ++p) v[j + 4 * p] = cmul(v[j + 4 * p], tw16(j * p));
#pragma unroll
  for (int p = 0; p < 4; ++p) dft4(v[4 * p], v[4 * p + 1], v[4 * p + 2], v[4 * p + 3]);   // over j: X[p + 4 r] in v[4 p + r]
  double2 o[16];
#pragma unroll
  for (int p = 0; p < 4; ++p)
#pragma unroll
    for (int r = 0; r < 4; ++r) o[p + 4 * r] = v[4 * p + r];
#pragma unroll
  for (int k = 0; k < 16; ++k) v[k] = o[k];
}

}  // namespace

// tw: [N / 2] W_N^m = exp(+2 pi i m / N), m < N / 2 (W_N^(m + N/2) = -W_N^m)
// WMODE 0: no window; 1: one window [N] for every row (each thread keeps its 16 weights in registers for the whole launch);
//       2: a window per baseline, bpwts [nbl][N], fetched with the row.
// The 16 loads of a row are issued back to back, and the rows of the NEXT pass are requested before the current pass is
// transformed (register double buffer): the kernel is HBM-bound only if enough bytes are in flight per CU.
template <int N, int WMODE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(N <= 1024 ? 2 : 1, 2))) void k_delay_fft(const double2* __restrict__ cube, const double* __restrict__ bpwts,
                                                    const double2* __restrict__ tw_g, double2* __restrict__ out,
                                                    double* __restrict__ out_pow, int64_t nrows, int64_t nbl, double scale,
                                                    double power_scale) {
  static_assert(N % 256 == 0 && N >= 256 && N <= 4096, "N = 256 R");
  constexpr int M = N / 16;             // threads per row = points per first-stage column
  constexpr int R = N / 256;            // radix of the last stage: 1, 2, 3 (768 channels), 4, 8, 16
  static_assert(R == 1 || R == 2 || R == 3 || R == 4 || R == 8 || R == 16, "last-stage radix");
  constexpr int RPB = 256 / M;          // rows per block pass (R = 3: 5 rows, 16 of the 256 threads only keep the barriers company)
  constexpr int NC = (256 + M - 1) / M; // stage-C passes of a thread over the 256 (p, p') columns
  // LDS exchange pitches.  A 64-lane 16-byte access is served 8 lanes per clock (128 B/clk): it is conflict-free when every group of 8
  // consecutive lanes touches 8 different 16-byte slots modulo 128 B.  Stage-B lanes are (p, a) = (lane / R, lane % R) and read
  // [p * PA + a + R q]: PA = M + R puts consecutive p at slot offsets R apart; they write [a * PB + p + 16 p']: PB = 256 + 8 / R does the
  // same for consecutive a (R = 3: 259; any odd pitch when R >= 8, where 8 lanes share one p).  Measured with the first layout
  // (pitches M + 1 and 257): SQ_LDS_BANK_CONFLICT = 50 % of SQ_LDS_IDX_ACTIVE.
  constexpr int PA = M + R;
  constexpr int PB = 256 + (R >= 8 ? 1 : (R == 3 ? 3 : (R >= 2 ? 8 / R : 1)));
  constexpr int PITCH = (16 * PA > R * PB) ? 16 * PA : R * PB;      // elements of LDS per row
  static_assert(16 * PA <= PITCH && R * PB <= PITCH, "exchange layouts fit the row buffer");
  __shared__ double2 tw[N / 2];
  __shared__ double2 xbuf[RPB * PITCH];

  for (int i = threadIdx.x; i < N / 2; i += 256) tw[i] = tw_g[i];
  constexpr bool ALLT = (RPB * M == 256);                          // every thread has a row (all but R = 3)
  constexpr bool POW2 = (N & (N - 1)) == 0;
  const bool tactive = ALLT || (int)threadIdx.x < RPB * M;         // R = 3: threads 240 ... 255 have no row
  const int rl = (ALLT || tactive) ? threadIdx.x / M : 0;          // row of this pass the thread works on
  const int j = threadIdx.x % M;        // thread within the row
  double2* const xb = xbuf + rl * PITCH;
  auto twid = [&](int m) {              // W_N^m for 0 <= m < N
    if constexpr (POW2) {
      const double2 w = tw[m & (N / 2 - 1)];
      return (m & (N / 2)) ? make_double2(-w.x, -w.y) : w;
    } else {
      const double2 w = tw[m >= N / 2 ? m - N / 2 : m];
      return (m >= N / 2) ? make_double2(-w.x, -w.y) : w;
    }
  };
  auto shifted = [&](int k) {           // (k + N/2) mod N
    if constexpr (POW2) return (k + N / 2) & (N - 1);
    else return k >= N / 2 ? k - N / 2 : k + N / 2;
  };
  double wfix[16];
  if constexpr (WMODE == 1) {
#pragma unroll
    for (int q = 0; q < 16; ++q) wfix[q] = bpwts[j + M * q];
  }
  __syncthreads();

  const int64_t npass = (nrows + RPB - 1) / RPB;
  double2 vn[16];
  double wn[16];
  auto fetch = [&](int64_t pass) {      // x[j + M q], q = 0 ... 15: consecutive j are consecutive addresses
    const int64_t row = pass * RPB + rl;
    const int64_t rr = row < nrows ? row : nrows - 1;            // rows past the end re-read the last one and are never stored
    const double2* src = cube + rr * N;
#pragma unroll
    for (int q = 0; q < 16; ++q) vn[q] = src[j + M * q];
    if constexpr (WMODE == 2) {
      const double* wsrc = bpwts + (rr % nbl) * N;
#pragma unroll
      for (int q = 0; q < 16; ++q) wn[q] = wsrc[j + M * q];
    }
  };
  if ((int64_t)blockIdx.x < npass) fetch(blockIdx.x);
  for (int64_t pass = blockIdx.x; pass < npass; pass += gridDim.x) {
    const int64_t row = pass * RPB + rl;
    const bool live = tactive && row < nrows;
    double2 v[16];
    // ---- stage A: window, 16-point DFT over q, twiddle W_N^(j p)
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      v[q] = vn[q];
      if constexpr (WMODE == 1) { v[q].x *= wfix[q]; v[q].y *= wfix[q]; }
      if constexpr (WMODE == 2) { v[q].x *= wn[q]; v[q].y *= wn[q]; }
    }
    if (pass + gridDim.x < npass) fetch(pass + gridDim.x);       // in flight while this pass is transformed
    dft_small<16>(v);
    if (tactive) {
#pragma unroll
      for (int p = 0; p < 16; ++p) xb[p * PA + j] = (p == 0) ? v[0] : cmul(v[p], twid(j * p));
    }
    __syncthreads();
    // ---- stage B: thread (p, a): 16-point DFT over j = a + R q'
    const int p = j / R, a = j % R;
#pragma unroll
    for (int q = 0; q < 16; ++q) v[q] = xb[p * PA + a + R * q];
    dft_small<16>(v);
    if constexpr (R == 1) {
      // k = p + 16 p': done.  out[(k + N/2) mod N]
      __syncthreads();                  // the row buffer is rewritten by the next pass
      if (live) {
#pragma unroll
        for (int pp = 0; pp < 16; ++pp) {
          const int k = p + 16 * pp;
          const int64_t o = row * N + shifted(k);
          const double2 r = make_double2(v[pp].x * scale, v[pp].y * scale);
          if (out) out[o] = r;
          if (out_pow) out_pow[o] = (r.x * r.x + r.y * r.y) * power_scale;
        }
      }
    } else {
      __syncthreads();                  // every thread of the row has read its stage-A values
      if (tactive) {
#pragma unroll
        for (int pp = 0; pp < 16; ++pp) xb[a * PB + p + 16 * pp] = (pp == 0) ? v[0] : cmul(v[pp], twid(16 * a * pp));
      }
      __syncthreads();
      // ---- stage C: thread L = j: R-point DFTs over a for the columns c = L + M m < 256; k = c + 256 r
#pragma unroll
      for (int m = 0; m < NC; ++m) {
        const int c = j + M * m;
        const bool cok = (256 % M == 0) || c < 256;
        double2 u[16];
#pragma unroll
        for (int aa = 0; aa < R; ++aa) u[aa] = xb[aa * PB + (cok ? c : 0)];
        dft_small<R>(u);
        if (live && cok) {
#pragma unroll
          for (int r = 0; r < R; ++r) {
            const int k = c + 256 * r;
            const int64_t o = row * N + shifted(k);
            const double2 res = make_double2(u[r].x * scale, u[r].y * scale);
            if (out) out[o] = res;
            if (out_pow) out_pow[o] = (res.x * res.x + res.y * res.y) * power_scale;
          }
        }
      }
      __syncthreads();                  // the row buffer is rewritten by the next pass
    }
  }
}

bool delay_fft_supported(int64_t nchan) {
  return nchan == 256 || nchan == 512 || nchan == 768 || nchan == 1024 || nchan == 2048 || nchan == 4096;
}

template <int N>
static hipError_t launch_delay_fft_n(const double* cube, const double* bpwts, int wts_rows, const double* tw, double* out, double* out_pow,
                                     int64_t nrows, int64_t nbl, double scale, double power_scale, int cu_count, hipStream_t stream) {
  constexpr int RPB = 256 / (N / 16);
  const int64_t npass = (nrows + RPB - 1) / RPB;
  int64_t grid = (int64_t)(cu_count > 0 ? cu_count : 256) * 2;          // persistent blocks: what is resident (2 per CU: 75 KB of LDS each)
  if (grid > npass) grid = npass;
  if (grid < 1) grid = 1;
  const dim3 g((unsigned)grid), b(256);
  const double2* c2 = reinterpret_cast<const double2*>(cube);
  const double2* t2 = reinterpret_cast<const double2*>(tw);
  double2* o2 = reinterpret_cast<double2*>(out);
  if (!bpwts)
    hipLaunchKernelGGL((k_delay_fft<N, 0>), g, b, 0, stream, c2, bpwts, t2, o2, out_pow, nrows, nbl, scale, power_scale);
  else if (wts_rows == 1)
    hipLaunchKernelGGL((k_delay_fft<N, 1>), g, b, 0, stream, c2, bpwts, t2, o2, out_pow, nrows, nbl, scale, power_scale);
  else
    hipLaunchKernelGGL((k_delay_fft<N, 2>), g, b, 0, stream, c2, bpwts, t2, o2, out_pow, nrows, nbl, scale, power_scale);
  return hipGetLastError();
}

// bpwts: device window, [N] (wts_rows == 1) or [nbl][N] (wts_rows == nbl), or NULL
hipError_t launch_delay_fft(const double* cube, const double* bpwts, int64_t wts_rows, const double* tw, double* out, double* out_pow,
                            int64_t nrows, int64_t nbl, int64_t nchan, double scale, double power_scale, int cu_count, hipStream_t stream) {
  if (nrows == 0) return hipSuccess;
  const int wr = wts_rows == 1 ? 1 : 2;
  switch (nchan) {
    case 256: return launch_delay_fft_n<256>(cube, bpwts, wr, tw, out, out_pow, nrows, nbl, scale, power_scale, cu_count, stream);
    case 512: return launch_delay_fft_n<512>(cube, bpwts, wr, tw, out, out_pow, nrows, nbl, scale, power_scale, cu_count, stream);
    case 768: return launch_delay_fft_n<768>(cube, bpwts, wr, tw, out, out_pow, nrows, nbl, scale, power_scale, cu_count, stream);
    case 1024: return launch_delay_fft_n<1024>(cube, bpwts, wr, tw, out, out_pow, nrows, nbl, scale, power_scale, cu_count, stream);
    case 2048: return launch_delay_fft_n<2048>(cube, bpwts, wr, tw, out, out_pow, nrows, nbl, scale, power_scale, cu_count, stream);
    case 4096: return launch_delay_fft_n<4096>(cube, bpwts, wr, tw, out, out_pow, nrows, nbl, scale, power_scale, cu_count, stream);
  }
  return hipErrorInvalidValue;
}

}  // namespace prisim
