// catalog_kernels.hip -- per-snapshot sky geometry of a device-resident catalogue.  gfx950 only.
//
// The reference re-derives the whole sky at every snapshot on the host (prisim/interferometry.py:6171-6180 hadec/radec -> alt-az,
// :6204-6219 region of interest, :6263 direction cosines) from a sky model that does not change over a run
// (scripts/run_prisim.py:2165-2207 loops observe() over n_acc with one skymod).  Here the catalogue is uploaded once
// (prisim_hip_set_catalog) as unit vectors in its own frame, and every snapshot's frame -> (l, m, n), horizon / ROI mask, STABLE
// compaction (the compacted index list is obs_catalog_indices, :6377), the altitude ordering the taper culling wants and the cull table
// itself are formed on the device.  The arithmetic is geometry.frame_dircos() of the host mirror operation by operation with FMA
// contraction off: index lists and direction cosines agree with the host path bit for bit.
//
// All kernels take a snapshot index from blockIdx.y: a batch of K snapshots is K rows of one launch.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>

#include "skyvis_kernels.h"

namespace prisim {

static constexpr int kCatBlock = 256;

// One catalogue source in one snapshot's frame: s = normalise(R (u + beta)) and the region-of-interest flag
// (interferometry.py:6174-6180 astropy FK5 -> AltAz / GEOM.hadec2altaz, :6263 altaz2dircos, :6204-6216 region of interest -- 'zenith':
// altitude >= 90 - roi_radius <=> n >= sin(90 - roi_radius); 'pointing_center': angle to it <= roi_radius <=> s . s_pc >= cos(roi_radius)).
// The frame (rotation + aberration vector) comes with the snapshot: prisim_snapshot.cel2enu / aberr_beta, or the plain sidereal rotation
// the library builds from lst and latitude.  Only +, *, sqrt and / with contraction off, in the order of geometry.frame_dircos() on the
// host: both sides round identically, so the index lists agree by construction and not by the luck of two maths libraries.
__device__ __forceinline__ bool cat_source(const CatGeomParams& p, const CatSnap& sn, int64_t i, double& l, double& m, double& n) {
#pragma clang fp contract(off)
  const double t0 = p.ux[i] + sn.beta[0], t1 = p.uy[i] + sn.beta[1], t2 = p.uz[i] + sn.beta[2];
  const double v0 = (sn.rot[0] * t0 + sn.rot[1] * t1) + sn.rot[2] * t2;
  const double v1 = (sn.rot[3] * t0 + sn.rot[4] * t1) + sn.rot[5] * t2;
  const double v2 = (sn.rot[6] * t0 + sn.rot[7] * t1) + sn.rot[8] * t2;
  const double nrm = sqrt((v0 * v0 + v1 * v1) + v2 * v2);
  l = v0 / nrm;
  m = v1 / nrm;
  n = v2 / nrm;
  if (p.roi_center == 0) return n >= p.sin_alt_min;
  const double cosd = (l * sn.roi_pc[0] + m * sn.roi_pc[1]) + n * sn.roi_pc[2];
  return cosd >= p.cos_radius;
}

// unit vectors of a catalogue given as (longitude, latitude) degrees, once per catalogue (a caller that wants the host's bits passes
// prisim_catalog.unitvec instead: sin / cos of two maths libraries may differ in the last place)
__global__ void k_cat_prepare(const double* __restrict__ lon_deg, const double* __restrict__ lat_deg, int coords, double* __restrict__ ux,
                              double* __restrict__ uy, double* __restrict__ uz, int64_t n) {
#pragma clang fp contract(off)
  constexpr double kD2R = 3.141592653589793238462643383279502884 / 180.0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const double a = lon_deg[i] * kD2R, d = lat_deg[i] * kD2R;
    if (coords == PRISIM_CAT_ALTAZ) {           // (alt, az): East-North-Up direction cosines, geometry.altaz2dircos
      const double ca = cos(a);
      ux[i] = ca * sin(d);
      uy[i] = ca * cos(d);
      uz[i] = sin(a);
    } else {
      const double cd = cos(d);
      ux[i] = cd * cos(a);
      uy[i] = cd * sin(a);
      uz[i] = sin(d);
    }
  }
}

// pass 1: ROI sources per block of 256 catalogue sources
__global__ __launch_bounds__(kCatBlock)
void k_cat_count(const CatGeomParams p) {
  const int snap = blockIdx.y;
  const CatSnap sn = p.snaps[snap];
  const int64_t i = (int64_t)blockIdx.x * kCatBlock + threadIdx.x;
  bool f = false;
  if (i < p.n) {
    double l, m, n;
    f = cat_source(p, sn, i, l, m, n);
  }
  const int c = __syncthreads_count(f ? 1 : 0);
  if (threadIdx.x == 0) p.block_off[(size_t)snap * p.nblocks + blockIdx.x] = c;
}

// pass 2: exclusive scan of a snapshot's block counts (one block per snapshot), total -> out[snap].nsrc
__global__ __launch_bounds__(1024)
void k_cat_scan(const CatGeomParams p) {
  __shared__ int64_t wsum[16];
  __shared__ int64_t carry_s;
  const int snap = blockIdx.x;
  int32_t* off = p.block_off + (size_t)snap * p.nblocks;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (threadIdx.x == 0) carry_s = 0;
  __syncthreads();
  for (int64_t base = 0; base < p.nblocks; base += 1024) {
    const int64_t j = base + threadIdx.x;
    const int64_t v = j < p.nblocks ? off[j] : 0;
    int64_t x = v;                                   // inclusive scan inside the wave
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const int64_t y = __shfl_up(x, d, 64);
      if (lane >= d) x += y;
    }
    if (lane == 63) wsum[wave] = x;
    __syncthreads();
    int64_t wbase = 0;
    for (int w = 0; w < wave; ++w) wbase += wsum[w];
    const int64_t carry = carry_s;
    if (j < p.nblocks) off[j] = (int32_t)(carry + wbase + x - v);
    __syncthreads();
    if (threadIdx.x == 1023) carry_s = carry + wbase + x;
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    p.out[snap].nsrc = carry_s;
    p.out[snap].dmax2_bits = 0;
    for (int r = 0; r <= PRISIM_CAT_MAX_RUNS; ++r) p.out[snap].run_start[r] = carry_s;      // runs with no catalogue source at all
    if (p.batch) {
      // the snapshot's entry of the batched launch's table (catalog.cpp run_wave_batch lays the blocks out; wave_split_sources there)
      const CatSnap& sn = p.snaps[snap];
      BatchSnap e;
      const int64_t N = carry_s;
      e.dir0 = (int64_t)snap * p.n; e.nsrc = N; e.pb0 = (int64_t)snap * p.n; e.row0 = (int64_t)snap * p.batch_npad;
      const int64_t n1 = N > 1 ? N : 1;
      e.nrow = (n1 + 3) / 4 * 4;
      const int64_t sw = p.batch_nsplit > 1 ? p.batch_nsplit : 1;
      e.src_per_split = ((n1 + sw - 1) / sw + 3) / 4 * 4;
      for (int i = 0; i < 3; ++i) { e.pc[i] = sn.pc[i]; e.bpc[i] = sn.bpc[i]; }
      e.out = p.batch_out + (size_t)snap * (size_t)(p.batch_nsplit > 1 ? p.batch_nsplit : 1) * (size_t)p.batch_slot_elems;
      e.gout = p.batch_gout ? p.batch_gout + (size_t)snap * (size_t)sw * 3 * (size_t)p.batch_slot_elems : nullptr;
      p.batch[snap] = e;
    }
  }
}

// pass 3: stable scatter -- compacted catalogue indices, directions (l, m, n, kappa), the first compacted source of every catalogue
// run, max |s - s_pc|^2, and (want_keys) the altitude keys of the culling order
__global__ __launch_bounds__(kCatBlock)
void k_cat_scatter(const CatGeomParams p) {
  __shared__ int wcount[kCatBlock / 64];
  __shared__ double wmax[kCatBlock / 64];
  const int snap = blockIdx.y;
  const CatSnap sn = p.snaps[snap];
  const int64_t i = (int64_t)blockIdx.x * kCatBlock + threadIdx.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  bool f = false;
  double l = 0.0, m = 0.0, n = 0.0;
  if (i < p.n) f = cat_source(p, sn, i, l, m, n);
  const uint64_t bal = __ballot(f ? 1 : 0);
  const int before = __popcll(bal & ((1ull << lane) - 1ull));
  if (lane == 0) wcount[wave] = __popcll(bal);
  __syncthreads();
  int wbase = 0;
  for (int w = 0; w < wave; ++w) wbase += wcount[w];
  const int64_t rank = (int64_t)p.block_off[(size_t)snap * p.nblocks + blockIdx.x] + wbase + before;     // ROI sources before catalogue source i
  const size_t row0 = (size_t)snap * (size_t)p.n;
  double e2 = 0.0;
  if (i < p.n) {
    const int run = p.run_id ? p.run_id[i] : 0;
    if (p.run_id && (i == 0 || p.run_id[i - 1] != run)) p.out[snap].run_start[run] = rank;
    if (f) {
      p.idx[row0 + rank] = (int32_t)i;
      reinterpret_cast<double4*>(p.dirs)[row0 + rank] = make_double4(l, m, n, p.kappa ? p.kappa[i] : 0.0);
      const double ex = l - sn.pc[0], ey = m - sn.pc[1], ez = n - sn.pc[2];
      e2 = ex * ex + ey * ey + ez * ez;
      if (p.want_keys) {
        // runs stay together (high bits); inside a run decreasing altitude = increasing 1 - n, 2^-28 steps
        double q = (1.0 - n) * 134217728.0;                      // (1 - n) / 2 * 2^28
        q = q < 0.0 ? 0.0 : (q > 268435455.0 ? 268435455.0 : q);
        p.keys[row0 + rank] = ((uint32_t)run << 28) | (uint32_t)q;
        p.pos[row0 + rank] = (uint32_t)rank;
      }
    }
  }
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) e2 = fmax(e2, __shfl_xor(e2, d, 64));
  if (lane == 0) wmax[wave] = e2;
  __syncthreads();
  if (threadIdx.x == 0) {
    double mx = wmax[0];
    for (int w = 1; w < kCatBlock / 64; ++w) mx = fmax(mx, wmax[w]);
    if (mx > 0.0) atomicMax((unsigned long long*)&p.out[snap].dmax2_bits, (unsigned long long)__double_as_longlong(mx));   // non-negative doubles order like their bits
  }
}

// Small catalogues (n <= kCatSmallMax) of small arrays: count, scan and scatter of a snapshot in ONE block of 256 threads (grid = snapshots) --
// thread t owns the contiguous sources [t m, (t + 1) m), m = ceil(n / 256) <= 64: it counts its region-of-interest sources (flags kept as a
// bit mask), the block scans the 256 counts, and the thread walks its sources again writing them at their ranks.  (A 1024-thread block
// was tried first: beside a large array's sky-sum grid it waited 36 ms for a whole CU to drain -- small blocks slip in as sky-sum blocks
// retire, and large arrays keep the three-pass form anyway.)  The same cat_source(), the same
// stable order, the same records as the three-pass form -- but one launch instead of three dependent ones, the snapshot's inputs in the
// kernel arguments instead of behind a host-to-device copy, and the result record written straight into page-locked host memory instead of
// through a device-to-host copy: a single snapshot's geometry costs the host one launch latency, not five (observe() on HERA-19: 45 us).
__global__ __launch_bounds__(256)
void k_cat_small(const CatGeomParams p) {
  __shared__ int64_t wsum[4];
  __shared__ double wmax[4];
  __shared__ int64_t total_s;
  const int snap = blockIdx.x;
  const CatSnap sn = p.inline_snap ? p.snap0 : p.snaps[snap];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t m = (p.n + 255) / 256;
  const int64_t i0 = (int64_t)tid * m < p.n ? (int64_t)tid * m : p.n, i1 = i0 + m < p.n ? i0 + m : p.n;
  uint64_t mask = 0;
  for (int64_t i = i0; i < i1; ++i) {
    double l, mm, n;
    if (cat_source(p, sn, i, l, mm, n)) mask |= 1ull << (i - i0);
  }
  const int64_t cnt = __popcll(mask);
  int64_t x = cnt;                                     // inclusive scan inside the wave, then over the 16 waves
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int64_t y = __shfl_up(x, d, 64);
    if (lane >= d) x += y;
  }
  if (lane == 63) wsum[wave] = x;
  __syncthreads();
  int64_t wbase = 0;
  for (int w = 0; w < wave; ++w) wbase += wsum[w];
  if (tid == 255) total_s = wbase + x;
  int64_t rank = wbase + x - cnt;                      // region-of-interest sources before source i0
  const size_t row0 = (size_t)snap * (size_t)p.n;
  double e2 = 0.0;
  for (int64_t i = i0; i < i1; ++i) {
    const int run = p.run_id ? p.run_id[i] : 0;
    if (p.run_id && i > 0 && p.run_id[i - 1] != run) p.out[snap].run_start[run] = rank;      // (run 0 starts at 0: written below)
    if (mask & (1ull << (i - i0))) {
      double l, mm, n;
      (void)cat_source(p, sn, i, l, mm, n);
      p.idx[row0 + rank] = (int32_t)i;
      reinterpret_cast<double4*>(p.dirs)[row0 + rank] = make_double4(l, mm, n, p.kappa ? p.kappa[i] : 0.0);
      const double ex = l - sn.pc[0], ey = mm - sn.pc[1], ez = n - sn.pc[2];
      e2 = fmax(e2, ex * ex + ey * ey + ez * ez);
      if (p.want_keys) {
        double q = (1.0 - n) * 134217728.0;                      // as k_cat_scatter
        q = q < 0.0 ? 0.0 : (q > 268435455.0 ? 268435455.0 : q);
        p.keys[row0 + rank] = ((uint32_t)run << 28) | (uint32_t)q;
        p.pos[row0 + rank] = (uint32_t)rank;
      }
      rank += 1;
    }
  }
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) e2 = fmax(e2, __shfl_xor(e2, d, 64));
  if (lane == 0) wmax[wave] = e2;
  __syncthreads();
  if (tid == 0) {
    const int64_t N = total_s;
    double mx = wmax[0];
    for (int w = 1; w < 4; ++w) mx = fmax(mx, wmax[w]);
    p.out[snap].nsrc = N;
    p.out[snap].dmax2_bits = mx > 0.0 ? (uint64_t)__double_as_longlong(mx) : 0;
    // runs no source of the catalogue starts (none exist past the last run id) keep "start = N"; run 0 starts at 0
    if (p.run_id) {
      const int last_run = p.run_id[p.n - 1];
      for (int r = last_run + 1; r <= PRISIM_CAT_MAX_RUNS; ++r) p.out[snap].run_start[r] = N;
      p.out[snap].run_start[0] = 0;
    } else {
      for (int r = 0; r <= PRISIM_CAT_MAX_RUNS; ++r) p.out[snap].run_start[r] = N;
    }
    if (p.batch) {
      BatchSnap e;
      e.dir0 = (int64_t)snap * p.n; e.nsrc = N; e.pb0 = (int64_t)snap * p.n; e.row0 = (int64_t)snap * p.batch_npad;
      const int64_t n1 = N > 1 ? N : 1;
      e.nrow = (n1 + 3) / 4 * 4;
      const int64_t sw = p.batch_nsplit > 1 ? p.batch_nsplit : 1;
      e.src_per_split = ((n1 + sw - 1) / sw + 3) / 4 * 4;
      for (int i = 0; i < 3; ++i) { e.pc[i] = sn.pc[i]; e.bpc[i] = sn.bpc[i]; }
      e.out = p.batch_out + (size_t)snap * (size_t)sw * (size_t)p.batch_slot_elems;
      e.gout = p.batch_gout ? p.batch_gout + (size_t)snap * (size_t)sw * 3 * (size_t)p.batch_slot_elems : nullptr;
      p.batch[snap] = e;
    }
  }
}

// after the altitude sort: directions and indices in the sorted order
__global__ void k_cat_gather(const uint32_t* __restrict__ perm, const double* __restrict__ dirs, const int32_t* __restrict__ idx,
                             double* __restrict__ dirs_out, int32_t* __restrict__ idx_out, int64_t n) {
  for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += (int64_t)gridDim.x * blockDim.x) {
    const uint32_t s = perm[j];
    reinterpret_cast<double4*>(dirs_out)[j] = reinterpret_cast<const double4*>(dirs)[s];
    idx_out[j] = idx[s];
  }
}

// Taper culling table on the device (capi.cpp upload_common has the derivation): first[(prec * nruns + r) * ng + g] = first source
// of run r that baseline group g still has to sum -- the leading sources whose exponent bound kappa max(Hmin |n| - Z rho, 0)^2 fmin^2/c^2
// is >= T (28 for fp64, 18 for fp32) are skipped.  One block per (group, run); the walk stops at the first source that fails, like the
// host's.  culled[prec] accumulates (skipped sources) x (baselines of the group) as integers (deterministic).
__global__ __launch_bounds__(256)
void k_cull_first(const CullParams p) {
  __shared__ int64_t first_s;
  const int g = blockIdx.x, r = blockIdx.y;
  const int64_t lo = p.run_lo[r], hi = p.run_hi[r];
  const double kap = p.run_kappa[r];
  const double H = p.grp_minh[g], Z = p.grp_maxz[g];
  for (int pr = 1; pr >= 0; --pr) {                     // fp32 (threshold 18) first; the fp64 walk (28) cannot go further
    const double thr = pr == 1 ? 18.0 : 28.0;
    int64_t first = lo;
    if (kap > 0.0 && kap * H * H * p.fc2 >= thr) {
      for (int64_t base = lo; base < hi; base += 256) {
        const int64_t s = base + threadIdx.x;
        bool fails = true;
        if (s < hi) {
          const double4 d = reinterpret_cast<const double4*>(p.dirs)[s];
          const double rho = sqrt(d.x * d.x + d.y * d.y), an = fabs(d.z);
          const double perp = H * an - Z * rho;
          fails = !(perp > 0.0 && kap * perp * perp * p.fc2 >= thr);
        }
        if (threadIdx.x == 0) first_s = -1;
        __syncthreads();
        if (fails && s < hi) atomicMin((unsigned long long*)&first_s, (unsigned long long)s);      // (-1 is the largest unsigned value)
        __syncthreads();
        const int64_t got = first_s;
        __syncthreads();
        if (got >= 0) { first = got; break; }
        first = base + 256 < hi ? base + 256 : hi;
      }
    }
    if (threadIdx.x == 0) {
      p.first[((size_t)pr * p.nruns + r) * p.ng + g] = (int32_t)first;
      if (first > lo) {
        const int64_t nb = p.nbl - (int64_t)g * 256 < 256 ? p.nbl - (int64_t)g * 256 : 256;
        atomicAdd((unsigned long long*)&p.culled[pr], (unsigned long long)((first - lo) * nb));
      }
    }
  }
}

// lifting-rotation flags of the baseline groups: |step phase| <= limit cycles guaranteed for every source of the sky
// (capi.cpp prisim_hip_compute: k = max_s |s - s_pc| |df| / c)
__global__ void k_lift_flags(const double* __restrict__ grp_maxlen, double k, double limit, int32_t* __restrict__ flags, int ng) {
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g < ng) flags[g] = (grp_maxlen[g] * k <= limit) ? 1 : 0;
}

hipError_t launch_lift_flags(const double* grp_maxlen, double k, double limit, int32_t* flags, int ng, hipStream_t stream) {
  if (ng <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_lift_flags, dim3((unsigned)((ng + 255) / 256)), dim3(256), 0, stream, grp_maxlen, k, limit, flags, ng);
  return hipGetLastError();
}

static unsigned grid1d(int64_t n) {
  int64_t g = (n + 255) / 256;
  if (g > 16384) g = 16384;
  if (g < 1) g = 1;
  return (unsigned)g;
}

hipError_t launch_cat_prepare(const double* lon_deg, const double* lat_deg, int coords, double* ux, double* uy, double* uz, int64_t n, hipStream_t stream) {
  if (n == 0) return hipSuccess;
  hipLaunchKernelGGL(k_cat_prepare, dim3(grid1d(n)), dim3(256), 0, stream, lon_deg, lat_deg, coords, ux, uy, uz, n);
  return hipGetLastError();
}

int64_t cat_blocks(int64_t n) { return (n + kCatBlock - 1) / kCatBlock; }

hipError_t launch_cat_geometry(const CatGeomParams& p, int nsnap, hipStream_t stream) {
  if (p.n == 0 || nsnap == 0) return hipSuccess;
  if (p.small_form) {
    hipLaunchKernelGGL(k_cat_small, dim3((unsigned)nsnap), dim3(256), 0, stream, p);
    return hipGetLastError();
  }
  hipLaunchKernelGGL(k_cat_count, dim3((unsigned)p.nblocks, (unsigned)nsnap), dim3(kCatBlock), 0, stream, p);
  hipLaunchKernelGGL(k_cat_scan, dim3((unsigned)nsnap), dim3(1024), 0, stream, p);
  hipLaunchKernelGGL(k_cat_scatter, dim3((unsigned)p.nblocks, (unsigned)nsnap), dim3(kCatBlock), 0, stream, p);
  return hipGetLastError();
}

size_t cat_sort_temp_bytes(int64_t n) {
  size_t bytes = 0;
  (void)rocprim::radix_sort_pairs(nullptr, bytes, (const uint32_t*)nullptr, (uint32_t*)nullptr, (const uint32_t*)nullptr, (uint32_t*)nullptr,
                                  (size_t)n, 0, 32, (hipStream_t)0);
  return bytes;
}

// stable sort of the n compacted sources of one snapshot by key, then directions / indices in that order
hipError_t launch_cat_sort(void* temp, size_t temp_bytes, const uint32_t* keys, uint32_t* keys_out, const uint32_t* pos, uint32_t* perm,
                           const double* dirs, const int32_t* idx, double* dirs_out, int32_t* idx_out, int64_t n, hipStream_t stream) {
  if (n == 0) return hipSuccess;
  hipError_t e = rocprim::radix_sort_pairs(temp, temp_bytes, keys, keys_out, pos, perm, (size_t)n, 0, 32, stream);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(k_cat_gather, dim3(grid1d(n)), dim3(256), 0, stream, perm, dirs, idx, dirs_out, idx_out, n);
  return hipGetLastError();
}

hipError_t launch_cull_first(const CullParams& p, hipStream_t stream) {
  if (p.nruns == 0 || p.ng == 0) return hipSuccess;
  hipLaunchKernelGGL(k_cull_first, dim3((unsigned)p.ng, (unsigned)p.nruns), dim3(256), 0, stream, p);
  return hipGetLastError();
}

}  // namespace prisim
