// aux_kernels.hip -- fused analytic primary beam x power-law flux, delay-transform pre/post
// passes around rocFFT, and the gathered-cube checksum.  gfx950 only.
//
// Reference statements restated:
//   Gaussian power beam   prisim/primary_beams.py:716-728
//   Airy power beam       prisim/primary_beams.py:609-623   (HERA preset D = 14 m, :239-247)
//   pbfluxes = pb*fluxes  prisim/interferometry.py:6254
//   delay transform       prisim/interferometry.py:8114-8134
//   delay power           prisim/delay_spectrum.py:3992-3993
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <algorithm>
#include "skyvis_kernels.h"
#include "../../include/prisim_hip.h"

namespace prisim {

static constexpr double kC = 299792458.0;
static constexpr double kPi = 3.14159265358979323846;

// thread per (source, channel); channel fastest (coalesced store of pb_out[s][f]).  The launchers make the total thread count a
// multiple of nchan, so a thread keeps ONE channel over its whole grid-stride loop and everything that depends on the frequency alone
// -- the wavenumber, the on-axis normalisation 2 J1(x0) / x0 of the Airy pattern (a second Bessel function per element otherwise), the
// Gaussian's width, the ground plane's denominator -- is formed once per thread (config 2 x 64 snapshots: 1.40 -> see DESIGN 4.4).
// One instance per element pattern (KIND), and per pattern one without array factor, beamformer and ground plane (EXTRAS = false: delta,
// Gaussian, Airy -- the analytic beams of the BASELINE configurations).  ONE body with every pattern behind run-time branches needed 286
// VGPRs: one wavefront per SIMD, and nothing to hide the index -> flux loads behind (config 2 x 64 snapshots spent 0.45 ms in it with a
// UNIFORM beam, 0.65 with the Airy pattern; now 0.06 / 0.27).  The statements are the same ones; the dead patterns are compiled out.
template <int KIND, bool EXTRAS>
__device__ __forceinline__ void beam_flux_body(BeamParams p) {
  constexpr int beam_kind = KIND;
  if (p.batch != nullptr) {            // one snapshot of a batch per blockIdx.y: its rows of the geometry set and of pb, its beam pointing
    const BatchSnap sn = p.batch[blockIdx.y];
    p.dirs += sn.dir0 * 4;
    if (p.src_index) p.src_index += sn.dir0;
    p.pb_out += sn.pb0 * p.nchan;
    p.nsrc = sn.nsrc;
    p.bpc_x = sn.bpc[0]; p.bpc_y = sn.bpc[1]; p.bpc_z = sn.bpc[2];
  }
  const int64_t total = p.nsrc * p.nchan;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;            // a multiple of nchan (launch_beam_flux)
  const int64_t i0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i0 >= total) return;
  const int64_t k = i0 % p.nchan;
  const int64_t ds = stride / p.nchan;                               // sources per step of the loop
  const double f = p.freqs[k];
  const double kk = 2.0 * kPi * f / kC;                              // :609, :950
  const double lam = kC / f;
  // Airy: small-angle floor and on-axis value (:611-618)
  const double sin_tol = sin(1e-10);
  const double a0 = kk * 0.5 * p.diameter * sin_tol;
  const double mx_airy = beam_kind == PRISIM_BEAM_AIRY ? 2.0 * j1(a0) / a0 : 1.0;
  // Gaussian (:717-721)
  const double sigma_aprtr = p.diameter / (2.0 * sqrt(2.0 * log(2.0))) / lam;
  const double sigma_dircos = 1.0 / (2.0 * kPi * sigma_aprtr);
  const double kh = kk * 0.5 * p.diameter;                           // dipole: k h, h = L/2 (:1203-1204)
  const double gp_den = (EXTRAS && p.gp_height > 0.0) ? 2.0 * sin(kk * p.gp_height) : 1.0;      // :965-966
  const double lg_fr = log2(f / p.ref_freq);                         // S = S0 (f / f_ref)^alpha = S0 2^(alpha log2(f / f_ref)): the logarithm once per thread
  int64_t s = i0 / p.nchan;
  for (int64_t i = i0; i < total; i += stride, s += ds) {
    const double4 d = reinterpret_cast<const double4*>(p.dirs)[s];
    // angle to the beam pointing centre: cos = s.p, sin = |s x p|
    const double cx = d.y * p.bpc_z - d.z * p.bpc_y;
    const double cy = d.z * p.bpc_x - d.x * p.bpc_z;
    const double cz = d.x * p.bpc_y - d.y * p.bpc_x;
    const double sinx = sqrt(cx * cx + cy * cy + cz * cz);
    const double cosx = d.x * p.bpc_x + d.y * p.bpc_y + d.z * p.bpc_z;
    // blank beyond the horizon of the dish or of the sky (primary_beams.py:607, 714)
    const bool blank = (cosx <= 0.0) || (d.z <= 0.0);
    // element FIELD pattern
    double ep = 1.0;
    if (beam_kind == PRISIM_BEAM_GAUSSIAN) {
      const double r = sinx / sigma_dircos;
      ep = blank ? 0.0 : exp(-0.5 * r * r);                                               // :724-725
    } else if (beam_kind == PRISIM_BEAM_AIRY) {
      const double sx = sinx < sin_tol ? sin_tol : sinx;                                  // :611-612 (x >= tol)
      const double a = kk * 0.5 * p.diameter * sx;
      const double pat = 2.0 * j1(a) / a;                                                 // :614
      ep = blank ? 0.0 : pat / mx_airy;                                                   // :616, :623
    } else if (beam_kind == PRISIM_BEAM_DIPOLE) {
      double dot = p.dip_x * d.x + p.dip_y * d.y + p.dip_z * d.z;                         // :1205
      dot = dot > 1.0 ? 1.0 : (dot < -1.0 ? -1.0 : dot);
      const double ang = acos(dot);                                                       // :1206
      const bool zero_ang = fabs(fabs(dot) - 1.0) < 1e-10;                                // :1209
      if (p.dipole_mode == PRISIM_DIPOLE_SHORT) {
        ep = sin(ang);                                                                    // :1215
      } else {
        double mx = 1.0;
        if (p.dipole_mode == PRISIM_DIPOLE_HALFWAVE) {
          ep = cos(0.5 * kPi * cos(ang)) / sin(ang);                                      // :1219
        } else {
          mx = 1.0 - cos(kh);                                                             // :1222
          ep = (cos(kh * cos(ang)) - cos(kh)) / sin(ang);                                 // :1223
        }
        if (zero_ang) ep = kh * sin(kh * cos(ang)) * tan(ang);                            // :1226 (L'Hospital)
        ep /= mx;                                                                         // :1230-1232
      }
    }
    // isotropic-radiator array factor (:1436-1475)
    double af = 1.0;
    if (EXTRAS && p.nax1 > 0) {
      const double rx = (p.rot_c * d.x + p.rot_s * d.y) - (p.rot_c * p.apc_x + p.rot_s * p.apc_y);      // :1443-1449
      const double ry = (-p.rot_s * d.x + p.rot_c * d.y) - (-p.rot_s * p.apc_x + p.rot_c * p.apc_y);
      const double phi = 2.0 * kPi * p.sep1 * rx / lam;                                   // :1458
      const double psi = 2.0 * kPi * p.sep2 * ry / lam;                                   // :1459
      const double n1 = (double)p.nax1, n2 = (double)p.nax2;
      const double t1 = fabs(phi) < 1e-10 ? cos(0.5 * n1 * phi) / cos(0.5 * phi) : sin(0.5 * n1 * phi) / sin(0.5 * phi) / n1;   // :1465-1467
      const double t2 = fabs(psi) < 1e-10 ? cos(0.5 * n2 * psi) / cos(0.5 * psi) : sin(0.5 * n2 * psi) / sin(0.5 * psi) / n2;   // :1469-1471 (nax2)
      af = t1 * t2;
    }
    double pb = (ep * af) * (ep * af);                                                    // :317 / :349 / :416
    if (EXTRAS && p.bf_nelem > 0) {
      // phased-array beamformer (:1728-1746): field of the elements with compensation delays and gains, power averaged over the
      // jitter realisations (:317, :416).  fp64 here; the reference forms the same sum in float32 / complex64.
      double acc = 0.0;
      for (int r = 0; r < p.bf_nrand; ++r) {
        double fre = 0.0, fi = 0.0;
        for (int e = 0; e < p.bf_nelem; ++e) {
          const double geo = -(p.bf_pos[3 * e] * d.x + p.bf_pos[3 * e + 1] * d.y + p.bf_pos[3 * e + 2] * d.z) / kC;
          double ph = f * (geo + p.bf_delays[(size_t)e * p.bf_nrand + r]);               // cycles
          ph -= rint(ph);
          double sn, cs;
          sincospi(2.0 * ph, &sn, &cs);
          const double g = p.bf_gains[(size_t)e * p.bf_nrand + r];
          fre = fma(g, cs, fre);
          fi = fma(g, sn, fi);
        }
        acc += fre * fre + fi * fi;
      }
      const double n2 = (double)p.bf_nelem * (double)p.bf_nelem;
      pb = ep * ep * acc / (n2 * (double)p.bf_nrand);
    }
    if (beam_kind == PRISIM_BEAM_POLY) {
      // VLA / GMRT polynomial in x = (zenith angle [deg] * 60 * f [GHz])^2 (:503, :508-509 / :796, :801); no blanking, no pointing
      const double th = atan2(sqrt(d.x * d.x + d.y * d.y), d.z) * (180.0 / kPi);      // zenith angle; well conditioned near the axis
      const double u = th * 60.0 * (f * 1e-9);
      const double x = u * u;
      pb = 1.0 + p.poly[0] * x / 1e3 + p.poly[1] * (x * x) / 1e7 + p.poly[2] * (x * x * x) / 1e10 + p.poly[3] * (x * x * x * x) / 1e13;
      if (pb != pb) atomicOr(p.flag, 2);
      else if (pb >= 1.01) atomicOr(p.flag, 1);
    }
    if (EXTRAS && p.gp_height > 0.0) {                                                    // ground plane (:950-966)
      const double nz = d.z < -1.0 ? -1.0 : (d.z > 1.0 ? 1.0 : d.z);                      // sin(alt) = n
      double gp = 2.0 * sin(kk * p.gp_height * nz);                                       // :953
      if (p.gp_modify & 1) {
        double val = 1.0 / sqrt(fabs(d.z));                                               // :957
        if (p.gp_modify & 2) val *= p.gp_scale;
        if (p.gp_modify & 4) val = val < 0.0 ? 0.0 : (val > p.gp_max ? p.gp_max : val);   // :960-961
        gp *= val;
      }
      gp /= gp_den;                                                                       // :965-966
      pb *= gp * gp;                                                                      // :439
    }
    // (catalogue path: the flux vectors / spectra stay in catalogue order and are read through the compacted index list)
    const int64_t cs = p.src_index ? (int64_t)p.src_index[s] : s;
    const double flux = p.flux_spec ? p.flux_spec[cs * p.nchan + k] : p.flux_ref[cs] * exp2(p.spindex[cs] * lg_fr);
    p.pb_out[i] = pb * flux;
  }
}


template <int KIND>
__global__ __launch_bounds__(256)
void k_beam_flux(BeamParams p) { beam_flux_body<KIND, true>(p); }

template <int KIND>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3)))
void k_beam_flux_plain(BeamParams p) { beam_flux_body<KIND, false>(p); }

// grid of a beam launch: as many blocks as the work wants, rounded to a whole number of channel periods (blocks x 256 threads a
// multiple of nchan) so that every thread keeps its channel
static unsigned beam_grid(int64_t total, int64_t nchan, int64_t cap, int64_t rows) {
  // a thread amortises its per-channel constants over ~16 sources as long as `rows` launches of this grid (snapshots of a batch)
  // still put >= 1024 blocks on the chip
  int64_t g = (total + 255) / 256;
  const int64_t want = std::max<int64_t>((g + 15) / 16, std::min<int64_t>(g, (1024 + rows - 1) / rows));
  g = std::min(want, cap);
  int64_t a = nchan, b = 256;
  while (b) { const int64_t t = a % b; a = b; b = t; }               // gcd(nchan, 256)
  const int64_t period = nchan / a;                                  // blocks per channel period
  g = (g + period - 1) / period * period;
  return (unsigned)g;
}

static void launch_beam_kernel(const BeamParams& p, dim3 grid, hipStream_t stream) {
  const bool plain = p.nax1 <= 0 && p.bf_nelem <= 0 && !(p.gp_height > 0.0);
  if (plain && p.beam_kind == PRISIM_BEAM_AIRY) hipLaunchKernelGGL((k_beam_flux_plain<PRISIM_BEAM_AIRY>), grid, dim3(256), 0, stream, p);
  else if (plain && p.beam_kind == PRISIM_BEAM_GAUSSIAN) hipLaunchKernelGGL((k_beam_flux_plain<PRISIM_BEAM_GAUSSIAN>), grid, dim3(256), 0, stream, p);
  else if (plain && p.beam_kind == PRISIM_BEAM_DELTA) hipLaunchKernelGGL((k_beam_flux_plain<PRISIM_BEAM_DELTA>), grid, dim3(256), 0, stream, p);
  else {
    switch (p.beam_kind) {
      case PRISIM_BEAM_DELTA: hipLaunchKernelGGL((k_beam_flux<PRISIM_BEAM_DELTA>), grid, dim3(256), 0, stream, p); break;
      case PRISIM_BEAM_GAUSSIAN: hipLaunchKernelGGL((k_beam_flux<PRISIM_BEAM_GAUSSIAN>), grid, dim3(256), 0, stream, p); break;
      case PRISIM_BEAM_AIRY: hipLaunchKernelGGL((k_beam_flux<PRISIM_BEAM_AIRY>), grid, dim3(256), 0, stream, p); break;
      case PRISIM_BEAM_DIPOLE: hipLaunchKernelGGL((k_beam_flux<PRISIM_BEAM_DIPOLE>), grid, dim3(256), 0, stream, p); break;
      default: hipLaunchKernelGGL((k_beam_flux<PRISIM_BEAM_POLY>), grid, dim3(256), 0, stream, p); break;
    }
  }
}

hipError_t launch_beam_flux(const BeamParams& p, hipStream_t stream) {
  const int64_t total = p.nsrc * p.nchan;
  if (total == 0) return hipSuccess;
  launch_beam_kernel(p, dim3(beam_grid(total, p.nchan, 16384, 1)), stream);
  return hipGetLastError();
}

// p.batch[nsnap] set, p.nsrc = the largest source count of the batch (sizes the grid)
hipError_t launch_beam_flux_batch(const BeamParams& p, int nsnap, hipStream_t stream) {
  const int64_t total = p.nsrc * p.nchan;
  if (total == 0 || nsnap <= 0 || !p.batch) return hipSuccess;
  launch_beam_kernel(p, dim3(beam_grid(total, p.nchan, 4096, nsnap), (unsigned)nsnap), stream);
  return hipGetLastError();
}

__global__ void k_mul_inplace(double* __restrict__ a, const double* __restrict__ b, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) a[i] *= b[i];
}

hipError_t launch_mul_inplace(double* a, const double* b, int64_t n, hipStream_t stream) {
  if (n == 0) return hipSuccess;
  int64_t g = (n + 255) / 256;
  if (g > 16384) g = 16384;
  hipLaunchKernelGGL(k_mul_inplace, dim3((unsigned)g), dim3(256), 0, stream, a, b, n);
  return hipGetLastError();
}

// ---- external HEALPix beam (scripts/run_prisim.py:2091-2103) ---------------------------------------------
// table[p][c] = sum_j M[c][j] * log10(beam[p][j])
__global__ void k_extbeam_table(const double* __restrict__ beam, const double* __restrict__ interp, double* __restrict__ table,
                                int64_t npix, int64_t nfreq, int64_t nchan) {
  const int64_t total = npix * nchan;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t pix = i / nchan, c = i - pix * nchan;
    double a = 0.0;
    for (int64_t j = 0; j < nfreq; ++j) {
      const double m = interp[c * nfreq + j];
      if (m != 0.0) a = fma(m, log10(beam[pix * nfreq + j]), a);
    }
    table[i] = a;
  }
}

struct RingInfo { int64_t start; int64_t nr; double theta; int shift; };

// HEALPix RING: start pixel, pixels in ring, colatitude and phi-shift flag of ring ir in [1, 4 nside - 1]
// (Healpix_Base::get_ring_info2)
__device__ __forceinline__ RingInfo ring_info(int64_t nside, int64_t ir) {
  const int64_t npix = 12 * nside * nside, ncap = 2 * nside * (nside - 1);
  const double fact2 = 4.0 / (double)npix, fact1 = (double)(2 * nside) * fact2;
  const bool south = ir > 2 * nside;
  const int64_t north = south ? 4 * nside - ir : ir;
  RingInfo r;
  if (north < nside) {
    const double tmp = (double)(north * north) * fact2;
    r.theta = atan2(sqrt(tmp * (2.0 - tmp)), 1.0 - tmp);
    r.nr = 4 * north;
    r.start = 2 * north * (north - 1);
    r.shift = 1;
  } else {
    double ct = (double)(2 * nside - north) * fact1;
    ct = ct > 1.0 ? 1.0 : (ct < -1.0 ? -1.0 : ct);
    r.theta = acos(ct);
    r.nr = 4 * nside;
    r.start = ncap + (north - nside) * 4 * nside;
    r.shift = (((north - nside) & 1) == 0) ? 1 : 0;
  }
  if (south) { r.theta = kPi - r.theta; r.start = npix - r.start - r.nr; }
  return r;
}

// work[s][c] = bilinear HEALPix interpolation of table[.][c] at the direction of source s
// (Healpix_Base::get_interpol / healpy.get_interp_val).  One block per source; thread 0 forms the 4 pixels + weights.
// batch != NULL: blockIdx.y = snapshot of a chunk (its directions in the geometry set, its rows of work; the external-beam kernels below alike)
__global__ __launch_bounds__(256)
void k_extbeam_gather(const double* __restrict__ table, int nside_i, const double* __restrict__ dirs, double* __restrict__ work,
                      int64_t nsrc, int64_t nchan, const BatchSnap* __restrict__ batch) {
  __shared__ int64_t spix[4];
  __shared__ double swgt[4];
  if (batch != nullptr) {
    const BatchSnap sn = batch[blockIdx.y];
    dirs += sn.dir0 * 4;
    work += sn.pb0 * nchan;
    nsrc = sn.nsrc;
  }
  const int64_t nside = nside_i;
  for (int64_t s = blockIdx.x; s < nsrc; s += gridDim.x) {
    __syncthreads();
    if (threadIdx.x == 0) {
      const double4 d = reinterpret_cast<const double4*>(dirs)[s];
      double z = d.z;                                   // cos(theta), theta = zenith angle
      z = z > 1.0 ? 1.0 : (z < -1.0 ? -1.0 : z);
      const double theta = acos(z);
      double phi = atan2(d.x, d.y);                     // azimuth from North through East
      if (phi < 0.0) phi += 2.0 * kPi;
      if (phi >= 2.0 * kPi) phi -= 2.0 * kPi;           // (-1e-17 + 2 pi rounds to 2 pi: a source due North to the last bit belongs to phi = 0,
                                                        //  not to a pixel index one past its ring)
      const int64_t npix = 12 * nside * nside;
      const double az = fabs(z);
      int64_t ir1;
      if (az <= 2.0 / 3.0) ir1 = (int64_t)((double)nside * (2.0 - 1.5 * z));
      else { const int64_t irr = (int64_t)((double)nside * sqrt(3.0 * (1.0 - az))); ir1 = z > 0.0 ? irr : 4 * nside - irr - 1; }
      const int64_t ir2 = ir1 + 1;
      int64_t pix[4] = {0, 0, 0, 0};
      double wgt[4] = {0, 0, 0, 0};
      double theta1 = 0.0, theta2 = 0.0;
      if (ir1 > 0) {
        const RingInfo r = ring_info(nside, ir1);
        const double dphi = 2.0 * kPi / (double)r.nr;
        const double tmp = phi / dphi - 0.5 * r.shift;
        int64_t i1 = tmp < 0.0 ? (int64_t)tmp - 1 : (int64_t)tmp;
        const double w1 = (phi - ((double)i1 + 0.5 * r.shift) * dphi) / dphi;
        int64_t i2 = i1 + 1;
        if (i1 < 0) i1 += r.nr;
        if (i2 >= r.nr) i2 -= r.nr;
        pix[0] = r.start + i1; pix[1] = r.start + i2; wgt[0] = 1.0 - w1; wgt[1] = w1; theta1 = r.theta;
      }
      if (ir2 < 4 * nside) {
        const RingInfo r = ring_info(nside, ir2);
        const double dphi = 2.0 * kPi / (double)r.nr;
        const double tmp = phi / dphi - 0.5 * r.shift;
        int64_t i1 = tmp < 0.0 ? (int64_t)tmp - 1 : (int64_t)tmp;
        const double w1 = (phi - ((double)i1 + 0.5 * r.shift) * dphi) / dphi;
        int64_t i2 = i1 + 1;
        if (i1 < 0) i1 += r.nr;
        if (i2 >= r.nr) i2 -= r.nr;
        pix[2] = r.start + i1; pix[3] = r.start + i2; wgt[2] = 1.0 - w1; wgt[3] = w1; theta2 = r.theta;
      }
      if (ir1 == 0) {                                   // north polar cap
        const double wth = theta / theta2;
        wgt[2] *= wth; wgt[3] *= wth;
        const double fac = (1.0 - wth) * 0.25;
        wgt[0] = fac; wgt[1] = fac; wgt[2] += fac; wgt[3] += fac;
        pix[0] = (pix[2] + 2) & 3; pix[1] = (pix[3] + 2) & 3;
      } else if (ir2 == 4 * nside) {                    // south polar cap
        const double wth = (theta - theta1) / (kPi - theta1);
        wgt[0] *= (1.0 - wth); wgt[1] *= (1.0 - wth);
        const double fac = wth * 0.25;
        wgt[0] += fac; wgt[1] += fac; wgt[2] = fac; wgt[3] = fac;
        pix[2] = ((pix[0] + 2) & 3) + npix - 4; pix[3] = ((pix[1] + 2) & 3) + npix - 4;
      } else {
        const double wth = (theta - theta1) / (theta2 - theta1);
        wgt[0] *= (1.0 - wth); wgt[1] *= (1.0 - wth); wgt[2] *= wth; wgt[3] *= wth;
      }
      for (int k = 0; k < 4; ++k) { spix[k] = pix[k]; swgt[k] = wgt[k]; }
    }
    __syncthreads();
    for (int64_t c = threadIdx.x; c < nchan; c += blockDim.x) {
      double a = 0.0;
#pragma unroll
      for (int k = 0; k < 4; ++k) a += table[spix[k] * nchan + c] * swgt[k];      // same summation order as the restatement
      work[s * nchan + c] = a;
    }
  }
}

// partial[blk][c] = nan-ignoring max over the sources handled by block blk; fixed block count => deterministic
__global__ __launch_bounds__(256)
void k_colmax_partial(const double* __restrict__ work, double* __restrict__ partial, int64_t nsrc, int64_t nchan,
                      const BatchSnap* __restrict__ batch) {
  if (batch != nullptr) {            // (a maximum does not depend on how the rows are dealt to blocks: the same bits as the single launch)
    const BatchSnap sn = batch[blockIdx.y];
    work += sn.pb0 * nchan;
    partial += (int64_t)blockIdx.y * gridDim.x * nchan;
    nsrc = sn.nsrc;
  }
  for (int64_t c = threadIdx.x; c < nchan; c += blockDim.x) {
    double m = -INFINITY;
    bool any = false;
    for (int64_t s = blockIdx.x; s < nsrc; s += gridDim.x) {
      const double v = work[s * nchan + c];
      if (!isnan(v)) { m = any ? (v > m ? v : m) : v; any = true; }
    }
    partial[(int64_t)blockIdx.x * nchan + c] = any ? m : NAN;
  }
}

// colmax[c] = max(nan-ignoring maximum of the partial rows, 0).  32 channels x 8 row lanes per block: the partial rows are read
// coalesced along the channel and 8 at a time (one thread per channel walking 1024 rows took 0.28 ms per snapshot at config 4).
__global__ __launch_bounds__(256)
void k_colmax_final(const double* __restrict__ partial, double* __restrict__ colmax, int nblk, int64_t nchan) {
  __shared__ double sm[8][32];
  __shared__ int sa[8][32];
  partial += (int64_t)blockIdx.y * nblk * nchan;          // batch: snapshot blockIdx.y (a single launch has gridDim.y = 1)
  colmax += (int64_t)blockIdx.y * nchan;
  const int cx = threadIdx.x & 31, py = threadIdx.x >> 5;
  const int64_t c = (int64_t)blockIdx.x * 32 + cx;
  double m = -INFINITY;
  bool any = false;
  if (c < nchan) {
    for (int b = py; b < nblk; b += 8) {
      const double v = partial[(int64_t)b * nchan + c];
      if (!isnan(v)) { m = any ? (v > m ? v : m) : v; any = true; }
    }
  }
  sm[py][cx] = m; sa[py][cx] = any ? 1 : 0;
  __syncthreads();
  if (py == 0 && c < nchan) {
    for (int q = 1; q < 8; ++q)
      if (sa[q][cx]) { m = any ? (sm[q][cx] > m ? sm[q][cx] : m) : sm[q][cx]; any = true; }
    m = any ? m : NAN;
    colmax[c] = (m <= 0.0) ? 0.0 : m;                   // run_prisim.py:2099-2100 (NaN stays NaN, as numpy's comparison leaves it)
  }
}

// pb_out[s][c] = float32( 10 ** (work - colmax[c]) ) * flux[s][c];  flux from the table `fluxes`, or, when that is NULL, the power law
// flux_ref[s] * (f_c / ref_freq) ** spindex[s] formed here (SkyModel.generate_spectrum of a 'func' sky model)
__global__ void k_extbeam_finish(const double* __restrict__ work, const double* __restrict__ colmax, const double* __restrict__ fluxes,
                                 const double* __restrict__ flux_ref, const double* __restrict__ spindex, const double* __restrict__ freqs,
                                 double inv_ref_freq, double* __restrict__ pb_out, int64_t nsrc, int64_t nchan,
                                 const int32_t* __restrict__ src_index, const BatchSnap* __restrict__ batch) {
  if (batch != nullptr) {
    const BatchSnap sn = batch[blockIdx.y];
    work += sn.pb0 * nchan;
    pb_out += sn.pb0 * nchan;
    colmax += (int64_t)blockIdx.y * nchan;
    if (src_index) src_index += sn.dir0;
    nsrc = sn.nsrc;
  }
  const int64_t total = nsrc * nchan;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t c = i % nchan;
    const double pb = (double)(float)exp10(work[i] - colmax[c]);      // :2101-2102 ; interferometry.py:4466 (float32 storage)
    double fl;
    const int64_t srow = i / nchan;
    const int64_t sidx = src_index ? (int64_t)src_index[srow] : srow;        // catalogue path: fluxes stay in catalogue order
    if (fluxes) {
      fl = fluxes[sidx * nchan + c];
    } else {
      fl = flux_ref[sidx] * pow(freqs[c] * inv_ref_freq, spindex[sidx]);
    }
    pb_out[i] = pb * fl;                                               // interferometry.py:6254
  }
}

static unsigned grid_for_(int64_t n) {
  int64_t g = (n + 255) / 256;
  if (g > 16384) g = 16384;
  if (g < 1) g = 1;
  return (unsigned)g;
}

hipError_t launch_extbeam_table(const double* beam, const double* interp, double* table, int64_t npix, int64_t nfreq,
                                int64_t nchan, hipStream_t stream) {
  hipLaunchKernelGGL(k_extbeam_table, dim3(grid_for_(npix * nchan)), dim3(256), 0, stream, beam, interp, table, npix, nfreq, nchan);
  return hipGetLastError();
}

hipError_t launch_extbeam_sky(const double* table, int nside, const double* dirs, const double* fluxes, const double* flux_ref,
                              const double* spindex, const double* freqs, double ref_freq, double* work,
                              double* colmax_scratch, double* pb_out, int64_t nsrc, int64_t nchan, hipStream_t stream,
                              const int32_t* src_index) {
  if (nsrc == 0) return hipSuccess;
  const unsigned gs = (unsigned)(nsrc < 16384 ? nsrc : 16384);
  hipLaunchKernelGGL(k_extbeam_gather, dim3(gs), dim3(256), 0, stream, table, nside, dirs, work, nsrc, nchan, (const BatchSnap*)nullptr);
  const int nblk = (int)(nsrc < 256 ? nsrc : 256);
  double* partial = colmax_scratch;
  double* colmax = colmax_scratch + (size_t)1024 * nchan;
  hipLaunchKernelGGL(k_colmax_partial, dim3(nblk), dim3(256), 0, stream, work, partial, nsrc, nchan, (const BatchSnap*)nullptr);
  hipLaunchKernelGGL(k_colmax_final, dim3((unsigned)((nchan + 31) / 32)), dim3(256), 0, stream, partial, colmax, nblk, nchan);
  hipLaunchKernelGGL(k_extbeam_finish, dim3(grid_for_(nsrc * nchan)), dim3(256), 0, stream, work, colmax, fluxes, flux_ref, spindex, freqs,
                     1.0 / ref_freq, pb_out, nsrc, nchan, src_index, (const BatchSnap*)nullptr);
  return hipGetLastError();
}

// The same for the snapshots of a chunk (batch[nsnap]; dirs / src_index: the chunk's geometry set; work, pb_out: the concatenated
// [sum nsrc][nchan] blocks; nsrc_max sizes the grids): four launches for the whole chunk.  colmax_scratch: nsnap * (kExtBatchBlocks + 1) * nchan.
hipError_t launch_extbeam_sky_batch(const double* table, int nside, const double* dirs, const double* fluxes, const double* flux_ref,
                                    const double* spindex, const double* freqs, double ref_freq, double* work, double* colmax_scratch,
                                    double* pb_out, int64_t nsrc_max, int64_t nchan, const int32_t* src_index, const BatchSnap* batch, int nsnap,
                                    hipStream_t stream) {
  if (nsrc_max == 0 || nsnap <= 0) return hipSuccess;
  const unsigned gs = (unsigned)std::min<int64_t>(nsrc_max, std::max<int64_t>(1, 16384 / nsnap));
  hipLaunchKernelGGL(k_extbeam_gather, dim3(gs, (unsigned)nsnap), dim3(256), 0, stream, table, nside, dirs, work, nsrc_max, nchan, batch);
  const int nblk = kExtBatchBlocks;
  double* partial = colmax_scratch;
  double* colmax = colmax_scratch + (size_t)nsnap * nblk * nchan;
  hipLaunchKernelGGL(k_colmax_partial, dim3(nblk, (unsigned)nsnap), dim3(256), 0, stream, work, partial, nsrc_max, nchan, batch);
  hipLaunchKernelGGL(k_colmax_final, dim3((unsigned)((nchan + 31) / 32), (unsigned)nsnap), dim3(256), 0, stream, partial, colmax, nblk, nchan);
  const unsigned gf = (unsigned)std::min<int64_t>(grid_for_(nsrc_max * nchan), std::max<int64_t>(1, 65536 / nsnap));
  hipLaunchKernelGGL(k_extbeam_finish, dim3(gf, (unsigned)nsnap), dim3(256), 0, stream, work, colmax, fluxes, flux_ref, spindex, freqs,
                     1.0 / ref_freq, pb_out, nsrc_max, nchan, src_index, batch);
  return hipGetLastError();
}

// ---- delay transform -----------------------------------------------------------------------
// work[row][n] (complex128, nfft per row) = cube[row][n] * w[b][n] for n < nchan, else 0.
__global__ void k_dt_prepare(const double2* __restrict__ cube, const double* __restrict__ bpwts, int64_t wts_rows,
                             double2* __restrict__ work, int64_t nrows, int64_t nbl, int64_t nchan, int64_t nfft) {
  const int64_t total = nrows * nfft;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t row = i / nfft;
    const int64_t n = i - row * nfft;
    double2 v = make_double2(0.0, 0.0);
    if (n < nchan) {
      v = cube[row * nchan + n];
      if (bpwts) {
        const double w = bpwts[(wts_rows == 1 ? 0 : (row % nbl)) * nchan + n];      // one window for every baseline, or one per baseline
        v.x *= w; v.y *= w;
      }
    }
    work[i] = v;
  }
}

// out[row][j] = scale * shifted[row][j*factor] with linear interpolation for non-integer factor,
// shifted = fftshift(work) (zero lag moved to index nfft/2).
__global__ void k_dt_finish(const double2* __restrict__ work, double2* __restrict__ out, double* __restrict__ out_power,
                            int64_t nrows, int64_t nfft, int64_t nout, double factor, double scale,
                            double power_scale) {
  const int64_t total = nrows * nout;
  const int64_t half = nfft / 2;   // numpy fftshift: shifted[i] = x[(i + nfft - half) % nfft]  (half = floor(n/2))
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t row = i / nout;
    const int64_t j = i - row * nout;
    const double pos = (double)j * factor;
    int64_t i0 = (int64_t)floor(pos);
    double frac = pos - (double)i0;
    if (i0 >= nfft - 1) { i0 = nfft - 1; frac = 0.0; }
    const int64_t src0 = (i0 + nfft - half) % nfft;
    double2 v = work[row * nfft + src0];
    if (frac != 0.0) {
      const int64_t src1 = (i0 + 1 + nfft - half) % nfft;
      const double2 v1 = work[row * nfft + src1];
      v.x += frac * (v1.x - v.x);
      v.y += frac * (v1.y - v.y);
    }
    v.x *= scale; v.y *= scale;
    if (out) out[i] = v;
    if (out_power) out_power[i] = (v.x * v.x + v.y * v.y) * power_scale;
  }
}

static unsigned grid_for(int64_t n) {
  int64_t g = (n + 255) / 256;
  if (g > 16384) g = 16384;
  if (g < 1) g = 1;
  return (unsigned)g;
}

hipError_t launch_dt_prepare(const double* cube, const double* bpwts, int64_t wts_rows, double* work, int64_t nrows, int64_t nbl,
                             int64_t nchan, int64_t nfft, hipStream_t stream) {
  hipLaunchKernelGGL(k_dt_prepare, dim3(grid_for(nrows * nfft)), dim3(256), 0, stream,
                     reinterpret_cast<const double2*>(cube), bpwts, wts_rows, reinterpret_cast<double2*>(work), nrows, nbl, nchan,
                     nfft);
  return hipGetLastError();
}

hipError_t launch_dt_finish(const double* work, double* out, double* out_power, int64_t nrows, int64_t nfft,
                            int64_t nout, double factor, double scale, double power_scale, hipStream_t stream) {
  hipLaunchKernelGGL(k_dt_finish, dim3(grid_for(nrows * nout)), dim3(256), 0, stream,
                     reinterpret_cast<const double2*>(work), reinterpret_cast<double2*>(out), out_power, nrows, nfft,
                     nout, factor, scale, power_scale);
  return hipGetLastError();
}

// ---- phase-centre rotation: V *= exp(-2 pi i f (b . diff_t)/c)   (interferometry.py:7871-7877) ------------
__global__ void k_phase_rotate(double2* __restrict__ cube, const double* __restrict__ blx, const double* __restrict__ bly,
                               const double* __restrict__ blz, const double* __restrict__ freqs, const double* __restrict__ diff,
                               int64_t nt, int64_t nbl, int64_t nchan) {
  const int64_t total = nt * nbl * nchan;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t f = i % nchan;
    const int64_t b = (i / nchan) % nbl;
    const int64_t t = i / (nchan * nbl);
    const double bdl = blx[b] * diff[3 * t] + bly[b] * diff[3 * t + 1] + blz[b] * diff[3 * t + 2];   // :7872
    double ph = bdl / kC * freqs[f];                    // cycles
    ph -= rint(ph);
    double sn, cs;
    sincospi(2.0 * ph, &sn, &cs);
    const double2 v = cube[i];
    cube[i] = make_double2(v.x * cs + v.y * sn, v.y * cs - v.x * sn);                                 // v * exp(-i phi), :7877
  }
}

hipError_t launch_phase_rotate(double* cube, const double* blx, const double* bly, const double* blz, const double* freqs,
                               const double* diff, int64_t nt, int64_t nbl, int64_t nchan, hipStream_t stream) {
  hipLaunchKernelGGL(k_phase_rotate, dim3(grid_for(nt * nbl * nchan)), dim3(256), 0, stream, reinterpret_cast<double2*>(cube), blx,
                     bly, blz, freqs, diff, nt, nbl, nchan);
  return hipGetLastError();
}

// ---- thermal noise: Philox-4x32-10 counter-based normals (Salmon et al. 2011) ---------------------------------
__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1,
                                              uint32_t out[4]) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
    const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
    const uint32_t n1 = (uint32_t)p1;
    const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
    const uint32_t n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

// one snapshot: out[b][f] = rms[b][f]/sqrt(2) * (n1 + i n2); counter = (f, global baseline, snapshot, 0)
__global__ void k_noise(const double* __restrict__ rms, double2* __restrict__ out, int64_t nbl, int64_t nchan, int64_t t,
                        int64_t bl_offset, uint64_t seed) {
  const int64_t total = nbl * nchan;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t b = i / nchan, f = i - b * nchan;
    uint32_t r[4];
    philox4x32_10((uint32_t)f, (uint32_t)(bl_offset + b), (uint32_t)t, (uint32_t)((uint64_t)(bl_offset + b) >> 32),
                  (uint32_t)seed, (uint32_t)(seed >> 32), r);
    // two 52-bit uniforms in (0,1], Box-Muller
    const double u1 = ((double)(((uint64_t)r[0] << 20) | (r[1] >> 12)) + 1.0) * (1.0 / 4503599627370496.0);
    const double u2 = ((double)(((uint64_t)r[2] << 20) | (r[3] >> 12))) * (1.0 / 4503599627370496.0);
    const double rad = sqrt(-2.0 * log(u1));
    double sn, cs;
    sincospi(2.0 * u2, &sn, &cs);
    const double sc = rms[i] * 0.70710678118654752440;                      // :6692 sqrt(2) split
    out[i] = make_double2(sc * rad * cs, sc * rad * sn);
  }
}

hipError_t launch_noise(const double* rms, double* out, int64_t nbl, int64_t nchan, int64_t t, int64_t bl_offset, uint64_t seed,
                        hipStream_t stream) {
  hipLaunchKernelGGL(k_noise, dim3(grid_for(nbl * nchan)), dim3(256), 0, stream, rms, reinterpret_cast<double2*>(out), nbl, nchan, t,
                     bl_offset, seed);
  return hipGetLastError();
}

// ---- deterministic checksum: fixed 1024-block partial sums, then one block ------------------
template <typename T>
__global__ __launch_bounds__(256)
void k_checksum_partial(const T* __restrict__ data, int64_t n, double* __restrict__ partial) {
  __shared__ double red[256];
  double a = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) a += (double)data[i];
  red[threadIdx.x] = a;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) partial[blockIdx.x] = red[0];
}

__global__ __launch_bounds__(256)
void k_checksum_final(const double* __restrict__ partial, int np, double* __restrict__ out) {
  __shared__ double red[256];
  double a = 0.0;
  for (int i = threadIdx.x; i < np; i += 256) a += partial[i];
  red[threadIdx.x] = a;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[0] = red[0];
}

__global__ void k_f64_to_f32(const double* __restrict__ in, float* __restrict__ out, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    out[i] = (float)in[i];
}

hipError_t launch_f64_to_f32(const double* in, float* out, int64_t n, hipStream_t stream) {
  if (n == 0) return hipSuccess;
  hipLaunchKernelGGL(k_f64_to_f32, dim3(grid_for(n)), dim3(256), 0, stream, in, out, n);
  return hipGetLastError();
}

// Un-deal one gathered snapshot: stage [nranks][planes][nbl_shard][row words] (rank blocks as the all-gather leaves them; shards were dealt
// round-robin in groups of baselines and padded, prisim_amd/sharding.py) -> out [planes][nbl_total][row words] in the GLOBAL baseline order
// of the unsharded array -- the order of the reference's rank-0 concatenate (scripts/run_prisim.py:2233-2242); map[r * nbl_shard + j] =
// global baseline of local row j of rank r, -1 = padding (dropped).  A pure copy: every word is read once and written once, rows are
// contiguous on both sides (coalesced 8-byte words; one block per (rank row, plane)).
__global__ __launch_bounds__(256)
void k_undeal(const uint64_t* __restrict__ stage, uint64_t* __restrict__ out, const int64_t* __restrict__ map, int64_t nbl_shard, int64_t nbl_total,
              int planes, int64_t row_words, int64_t nrows) {
  for (int64_t rj = blockIdx.x; rj < nrows; rj += gridDim.x) {
    const int64_t g = map[rj];
    if (g < 0) continue;
    const int64_t r = rj / nbl_shard, j = rj - r * nbl_shard;
    const int k = blockIdx.y;
    const uint64_t* src = stage + ((size_t)(r * planes + k) * (size_t)nbl_shard + (size_t)j) * (size_t)row_words;
    uint64_t* dst = out + ((size_t)k * (size_t)nbl_total + (size_t)g) * (size_t)row_words;
    for (int64_t w = threadIdx.x; w < row_words; w += blockDim.x) dst[w] = src[w];
  }
}

hipError_t launch_undeal(const void* stage, void* out, const int64_t* map, int nranks, int64_t nbl_shard, int64_t nbl_total, int planes,
                         int64_t row_words, hipStream_t stream) {
  const int64_t nrows = (int64_t)nranks * nbl_shard;
  if (nrows == 0 || row_words == 0) return hipSuccess;
  const unsigned gx = (unsigned)std::min<int64_t>(nrows, 65535);
  hipLaunchKernelGGL(k_undeal, dim3(gx, (unsigned)planes), dim3(256), 0, stream, (const uint64_t*)stage, (uint64_t*)out, map, nbl_shard, nbl_total,
                     planes, row_words, nrows);
  return hipGetLastError();
}


hipError_t launch_checksum(const void* data, bool is_f32, int64_t n, double* out /* [1025] device scratch: out[0]=result */,
                           hipStream_t stream) {
  if (is_f32)
    hipLaunchKernelGGL(k_checksum_partial<float>, dim3(1024), dim3(256), 0, stream, (const float*)data, n, out + 1);
  else
    hipLaunchKernelGGL(k_checksum_partial<double>, dim3(1024), dim3(256), 0, stream, (const double*)data, n, out + 1);
  hipLaunchKernelGGL(k_checksum_final, dim3(1), dim3(256), 0, stream, out + 1, 1024, out);
  return hipGetLastError();
}

}  // namespace prisim
