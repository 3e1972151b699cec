/*
 * prisim_hip.h -- C-ABI of the MI355X-native PRISim sky-sum library (libprisim_hip.so).
 *
 * The reference (nithyanandan/PRISim) has NO native / FFI / plugin interface for this
 * path: the sky-sum is inline numpy inside InterferometerArray.observe()
 * (prisim/interferometry.py:6255-6376).  This header therefore declares the boundary a
 * maintainer would bind with ctypes; every entry point cites the
 * reference statements it replaces.  Plain C types only -- no torch, no C++ exceptions.
 *
 * Conventions
 *   - all host arrays are C-contiguous, caller-owned, and only read/written during the call;
 *   - every function returns 0 on success or a negative PRISIM_E* code; the message for the
 *     last failure on a context is available from prisim_hip_last_error();
 *   - one context <-> one GPU <-> one HIP stream; calls on one context must be serialised
 *     by the caller, distinct contexts are independent;
 *   - complex arrays are interleaved (re, im) pairs.
 */
#ifndef PRISIM_HIP_H
#define PRISIM_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct prisim_ctx prisim_ctx;

enum {
  PRISIM_OK = 0,
  PRISIM_EINVAL = -1,      /* bad argument (maps to ValueError / TypeError in the Python shim) */
  PRISIM_ENODEV = -2,      /* no usable HIP device / HIP runtime error */
  PRISIM_ENOMEM = -3,      /* device allocation failed (MemoryError) */
  PRISIM_ESTATE = -4,      /* call order violated (e.g. compute before set_array) */
  PRISIM_ELIB = -5,        /* optional library (rocFFT / RCCL) unavailable */
  PRISIM_EINTERNAL = -6
};

/* precision of the accumulate / phasor arithmetic */
enum {
  PRISIM_FP64 = 0,         /* reference default path, interferometry.py:6332-6343 */
  PRISIM_FP32 = 1          /* reference "memsave" path, :6323-6330; here the phase is range-reduced
                              in fp64 before the fp32 phasor recurrence */
};

/* kernel selection for prisim_hip_compute() */
enum {
  PRISIM_KERNEL_AUTO = 0,        /* channel-recurrence kernel when the channel grid is uniform, else direct */
  PRISIM_KERNEL_RECURRENCE = 1,  /* phasor recurrence along frequency (needs uniform channel spacing) */
  PRISIM_KERNEL_DIRECT = 2       /* one sincospi per term; any channel grid; fp64 only; slow cross-check */
};

/* ---- lifetime -------------------------------------------------------------------------- */

/* Create a context on HIP device `device` (index after HIP_VISIBLE_DEVICES). */
int prisim_hip_create(int device, prisim_ctx** out);
void prisim_hip_destroy(prisim_ctx* ctx);
/* Message of the last error on ctx (ctx may be NULL: last error of a failed create). */
const char* prisim_hip_last_error(const prisim_ctx* ctx);
/* Library version string "prisim_hip <major>.<minor> gfx950"; the minor number changes with every change of a struct or a signature
   in this header (0.2: prisim_timing carries the delay-stage fields; 0.3: prisim_comm_stats, self-test, gradient gather, asynchronous downloads;
   0.4: the device-resident catalogue -- prisim_catalog / prisim_obs / prisim_snapshot and their four entries; 0.5: the snapshot carries its
   frame -- prisim_snapshot.frame_given / cel2enu / aberr_beta, prisim_catalog.unitvec -- and the shard map / global-order gather).
   A binding should refuse a library that reports another one. */
const char* prisim_hip_version(void);

/* ---- array: baselines + channels, resident across snapshots ---------------------------- */

/* Replaces the per-call use of self.baselines / self.channels in observe()
 * (interferometry.py:6151, 6332).  bl_enu: [nbl][3] metres, local East-North-Up
 * (baseline_coords='localenu').  freqs_hz: [nchan] Hz.  nt_max: number of snapshot slots to
 * allocate in the device visibility cube (layout [nt][nbl][nchan] complex128). */
int prisim_hip_set_array(prisim_ctx* ctx, const double* bl_enu, int64_t nbl,
                         const double* freqs_hz, int64_t nchan, int64_t nt_max);

/* ---- sky for one snapshot --------------------------------------------------------------- */

typedef struct prisim_sky {
  int64_t nsrc;              /* sources inside the region of interest (may be 0, :6378-6382) */
  const double* dircos;      /* [nsrc][3] ENU direction cosines (GEOM.altaz2dircos of skypos_altaz_roi, :6255/:6263) */
  const void* pbflux;        /* [nsrc][nchan] beam x flux = pb * fluxes (:6254); dtype per pbflux_is_f32 */
  int32_t pbflux_is_f32;     /* 0: float64, 1: float32 (external beams are stored float32, :4466) */
  const double* pc_dircos;   /* [3] phase-centre direction cosines (:6164) */
  const double* fwhm_deg;    /* [nsrc] sqrt(maj*min) of skymodel.src_shape in degrees (:6267), or NULL:
                                no source-shape taper (skymodel.src_shape is None, :6258) */
  const double* fluxes;      /* optional [nsrc][nchan] float64: when non-NULL `pbflux` holds the beam pb only and the
                                product pb * fluxes (:6254) is formed on the device; NULL: pbflux is already the product */
} prisim_sky;

/* Upload the snapshot's sky to the device (pbflux is repacked on the device into per-channel-tile
 * slabs).  Replaces nothing numerically; it is the H2D half of :6254-6255. */
int prisim_hip_set_sky(prisim_ctx* ctx, const prisim_sky* sky);

/* Compute the snapshot from device-resident inputs into cube slot `slot`:
 *   V[b,f] = sum_s pbflux[s,f] * w[s,b,f] * exp(-2 pi i f (tau[s,b] - taupc[b]))     (:6320-6343)
 *   tau = dc . bl^T / c (baseline_delay_horizon.py:240),  taupc (:6165),  w (:6257-6283).
 * want_grad != 0 additionally computes G_k[b,f] = sum_s dircos[s,k] * (summand), k=0..2 (:6338,:6343)
 * into the gradient cube slot.  Asynchronous on the context stream. */
int prisim_hip_compute(prisim_ctx* ctx, int precision, int kernel, int want_grad, int64_t slot);

/* Copy slot `slot` to the host.  out_is_c64 = 0: complex128 [nbl][nchan]; 1: complex64 (memsave dtype, :6183).
 * grad (may be NULL): [3][nbl][nchan], same dtype.  Synchronises the stream. */
int prisim_hip_get_vis(prisim_ctx* ctx, int64_t slot, void* vis, void* grad, int out_is_c64);

/* Upload a host visibility snapshot [nbl][nchan] complex128 into cube slot `slot` (used to delay-transform
 * cubes that live on the host, e.g. vis_freq / vis_noise_freq of interferometry.py:8116-8118). */
int prisim_hip_set_vis(prisim_ctx* ctx, int64_t slot, const double* vis);

/* One-shot drop-in for interferometry.py:6255-6376: set_sky + compute + get_vis. */
int prisim_hip_skyvis(prisim_ctx* ctx, const prisim_sky* sky, int precision, int kernel,
                      void* vis, void* grad, int out_is_c64);

/* ---- fused analytic primary beams on the device ------------------------ */

enum {
  PRISIM_BEAM_DELTA = 0,     /* pb = 1 (telescope shape 'delta', primary_beams.py:357-359) */
  PRISIM_BEAM_GAUSSIAN = 1,  /* primary_beams.py:716-728 (power pattern) */
  PRISIM_BEAM_AIRY = 2,      /* primary_beams.py:609-623 (power pattern, HERA D=14 m preset :239-247) */
  PRISIM_BEAM_DIPOLE = 3,    /* primary_beams.py:1207-1235 (field pattern squared); needs prisim_beam_ext */
  PRISIM_BEAM_POLY = 4       /* VLA / GMRT polynomial power beams (:445-513, :734-808); needs prisim_beam_ext.poly_coef */
};

enum { PRISIM_DIPOLE_GENERAL = 0, PRISIM_DIPOLE_SHORT = 1, PRISIM_DIPOLE_HALFWAVE = 2 };

/* Optional factors multiplying the element pattern (primary_beams.py:282-286, 317, 416-439):
 *   power = | element_field x array_factor |^2 x ground_plane_field^2 */
typedef struct prisim_beam_ext {
  double dipole_dircos[3];   /* dipole axis (ENU direction cosines), PRISIM_BEAM_DIPOLE only (:1207) */
  int32_t dipole_mode;       /* PRISIM_DIPOLE_* (:1214-1224) */
  int32_t array_nax1;        /* isotropic-radiator array factor (:1460-1475); 0 = no array factor */
  int32_t array_nax2;
  int32_t ground_modify;     /* bit 0: apply 1/sqrt(|n|) modifier (:956-958); bit 1: scale given; bit 2: max given */
  double array_sep1, array_sep2;     /* element separations in metres along axis 1 / 2 */
  double array_east2ax1_deg;         /* rotation of axis 1 anti-clockwise from East (:1436-1441) */
  double array_pc_dircos[3];         /* array pointing centre (zenith = {0,0,1}) */
  double ground_height;      /* ground-plane height in metres (:950-966); <= 0 = no ground plane */
  double ground_scale, ground_max;
  /* Phased-array beamformer over isotropic elements (array_field_pattern, primary_beams.py:1482-1754, called from
   * primary_beam_generator :288-317 and :385-416 when pointing_info is given).  bf_nelem = 0: none.  Otherwise
   *   F[s,f,r] = 1/N sum_i g[i][r] exp(2 pi i f (-pos_i . s / c + delay[i][r])),  power *= mean_r |F[s,f,r]|^2   (:317, :416)
   * and the analytic array_* factor above must be off.  Delays are the beamformer's compensation delays (seconds) including any
   * jitter realisations, gains likewise: the host draws them (the reference uses numpy's global RNG, :1655, :1665). */
  int32_t bf_nelem;          /* 0 ... 4096 */
  int32_t bf_nrand;          /* realisations, 1 ... 256 */
  const double* bf_pos;      /* host [bf_nelem][3] ENU metres */
  const double* bf_delays;   /* host [bf_nelem][bf_nrand] seconds */
  const double* bf_gains;    /* host [bf_nelem][bf_nrand] */
  /* PRISIM_BEAM_POLY: power = 1 + c0 x/1e3 + c1 x^2/1e7 + c2 x^3/1e10 + c3 x^4/1e13 with x = (zenith angle [deg] * 60 * f [GHz])^2
   * (:508-509, :801); like the reference the call fails (PRISIM_EINVAL) if any value is NaN or >= 1.01 (:510-512, :802-807). */
  double poly_coef[4];
} prisim_beam_ext;

typedef struct prisim_beam_sky {
  int64_t nsrc;
  const double* dircos;        /* [nsrc][3] ENU */
  const double* flux_ref;      /* [nsrc] flux density at ref_freq_hz (Jy) */
  const double* spindex;       /* [nsrc] spectral index: S(f) = flux_ref * (f/ref_freq)^spindex */
  const double* flux_spectrum; /* optional [nsrc][nchan] float64 spectra (SkyModel.generate_spectrum output, :6249);
                                  when non-NULL it replaces the power law and flux_ref/spindex may be NULL */
  double ref_freq_hz;
  int32_t beam_kind;           /* PRISIM_BEAM_* */
  double diameter_m;           /* dish diameter / Gaussian FWHM aperture size */
  const double* beam_pc_dircos;/* [3] beam pointing centre (ENU); zenith = {0,0,1} */
  const double* pc_dircos;     /* [3] phase centre */
  const double* fwhm_deg;      /* [nsrc] or NULL */
  const prisim_beam_ext* ext;  /* optional dipole / array factor / ground plane; NULL = none */
} prisim_beam_sky;

/* Build pbflux[s,f] = beam(s,f) * flux(s,f) on the device (no nsrc x nchan host array) and make it the
 * current sky.  Replaces primary_beam_generator + generate_spectrum(power law) + :6254 for these leaves. */
int prisim_hip_set_sky_analytic(prisim_ctx* ctx, const prisim_beam_sky* sky);

/* External (tabulated) primary beam, resident across snapshots (scripts/run_prisim.py:489-494, 2091-2103).
 * beam: [npix][nfreq] > 0, HEALPix RING in the local frame (theta = zenith angle, phi = azimuth N->E).
 * interp_matrix: [nchan][nfreq] linear operator of the spectral interpolation onto the channel grid
 * (scipy interp1d of kind beam.spec_interp applied to unit vectors; one row with a single 1 = achromatic
 * nearest-frequency selection, :2096-2097).  The device forms table[p][c] = sum_j M[c][j] log10(beam[p][j]). */
int prisim_hip_set_external_beam(prisim_ctx* ctx, const double* beam, int64_t npix, int64_t nfreq,
                                 const double* interp_matrix);

/* Sky for one snapshot with the external beam: sky->pbflux must be NULL and sky->fluxes [nsrc][nchan] given.
 * On the device: bilinear HEALPix interpolation of the table at each source (healpy get_interp_val),
 * subtraction of max(per-channel maximum over the sources, 0) (:2098-2101), 10**, rounding to float32
 * (the reference stores supplied beams as float32, interferometry.py:4466), times fluxes (:6254). */
int prisim_hip_set_sky_external(prisim_ctx* ctx, const prisim_sky* sky);

/* The same with the flux spectra formed on the device: S = flux_ref * (f / ref_freq_hz)^spindex (the spectrum
 * SkyModel.generate_spectrum returns for a power-law sky model, interferometry.py:6249), or sky->flux_spectrum when given.
 * Only nsrc-sized vectors cross PCIe per snapshot (a drift scan's 32 accumulations re-send no nsrc x nchan table);
 * beam_kind, diameter_m, beam_pc_dircos and ext are ignored.  Like every set_sky_* call it allocates nothing after the first
 * snapshot of a given size and does not synchronise the stream: inputs are copied to pinned staging before it returns. */
int prisim_hip_set_sky_external_analytic(prisim_ctx* ctx, const prisim_beam_sky* sky);

/* Read back the device pbflux (float64 [nsrc][nchan]) -- for parity tests of the fused beams. */
int prisim_hip_get_pbflux(prisim_ctx* ctx, double* out);

/* ---- device-resident catalogue: the sky of a whole run stays in HBM (ABI 0.4) ------------------------------
 *
 * The reference re-derives the sky at every snapshot from a sky model that does not change over a run
 * (scripts/run_prisim.py:2165-2207 loops observe() over n_acc with ONE skymod; interferometry.py:6223-6247
 * store_prev_skymodel_file exists because that is expensive).  Here the catalogue is uploaded once, as unit vectors u in its own frame,
 * and every snapshot's
 *   catalogue frame -> local East-North-Up   s = normalise(R (u + beta))      interferometry.py:6174-6180 (radec), :6176-6177 (hadec)
 *   region of interest                       :6204-6219  (zenith: n >= sin(90 - roi_radius); pointing centre: s . s_pc >= cos(roi_radius))
 *   direction cosines                        :6263       (GEOM.altaz2dircos: they ARE s)
 *   obs_catalog_indices (STABLE compaction)  :6377
 *   flux spectra of the ROI sources, pb * fluxes   :6249-6254
 * is formed on the device; the host reads back one small record per snapshot (source count, run boundaries).
 *
 * WHAT THE FRAME IS.  For skycoords 'radec' the reference goes FK5(equinox = skymodel.epoch) -> FK5(equinox = obstime) -> AltAz(obstime,
 * location) through astropy (:6174-6180): precession, nutation, annual (and diurnal) aberration, Earth rotation, polar motion.  All of that
 * is ONE rotation R (catalogue axes -> local East, North, Up) and ONE vector beta (observer velocity / c in the catalogue frame; first-order
 * aberration, exact to 1e-3 arcsec), which is what prisim_snapshot carries (frame_given = 1).  The library does no astrometry of its own
 * beyond the fall-back below; who fills R and beta decides what is modelled:
 *   - prisim_amd/frames.py (the Python host mirror, default): IAU 2006 precession angles + truncated IAU 1980 nutation + annual
 *     aberration + rotation by the caller's apparent LST + latitude tilt; NOT modelled: FK5/ICRS frame bias, light deflection, diurnal
 *     aberration, polar motion (together < 1 arcsec; list with sizes in that module).  Parity with astropy is UNPINNED (not installable here);
 *   - a PRISim-side binding that has astropy fills them from astropy itself and reproduces :6174-6180 exactly.
 * frame_given = 0 (fall-back): R = tilt(latitude) . rot_z(lst_deg), beta = 0, i.e. hour angle = LST - RA with NO precession, nutation or
 * aberration -- correct only for a catalogue already in the true equator and equinox of the snapshot; it is NOT what :6174-6180 computes.
 * For 'hadec' catalogues the reference itself uses the plain rotation (GEOM.hadec2altaz, :6176-6177) and for 'altaz' none; frame_given = 0
 * is exact for those. */
enum { PRISIM_COORDS_RADEC = 0, PRISIM_COORDS_HADEC = 1, PRISIM_COORDS_ALTAZ = 2 };

typedef struct prisim_catalog {
  int64_t nsrc;
  int32_t coords;               /* PRISIM_COORDS_*: what `location` holds (InterferometerArray.skycoords, :5862-5865) */
  int32_t reserved_;
  const double* location;       /* [nsrc][2] degrees: (RA, Dec) | (HA, Dec) | (alt, az)   (skymodel.location) */
  const double* flux_ref;       /* [nsrc] flux density at ref_freq_hz, with spindex: S = flux_ref (f / ref_freq)^spindex ... */
  const double* spindex;        /* [nsrc] */
  double ref_freq_hz;
  const double* flux_spectrum;  /* ... or [nsrc][nchan] spectra on the channel grid (SkyModel.generate_spectrum of the whole catalogue, :6249);
                                   when non-NULL it replaces the power law */
  const double* fwhm_deg;       /* [nsrc] sqrt(maj * min) of skymodel.src_shape (:6267), or NULL (no source-shape taper) */
  const double* unitvec;        /* optional [nsrc][3]: the catalogue's unit vectors in its own frame -- RA-Dec / HA-Dec (cos d cos a, cos d sin a, sin d),
                                   alt-az East-North-Up direction cosines.  NULL: the device forms them from `location` (its sin / cos may
                                   differ from the host's in the last place); given: `location` may be NULL, and a host that applies
                                   s = normalise(R (u + beta)) with the same operation order selects the same sources bit for bit */
} prisim_catalog;

/* Upload the catalogue (once per run; set_array drops it: the spectra are per channel grid).  Synchronises the stream. */
int prisim_hip_set_catalog(prisim_ctx* ctx, const prisim_catalog* cat);

/* What the snapshots of a run share (keywords of observe(), :5874-5880) */
typedef struct prisim_obs {
  double latitude_deg;          /* self.latitude */
  double roi_radius_deg;        /* roi_radius (default 90, :6207) */
  int32_t roi_center;           /* 0: 'zenith' (:6214-6216), 1: 'pointing_center' (:6210-6213, the snapshot's pc_dircos) */
  int32_t use_external_beam;    /* 1: the table of prisim_hip_set_external_beam; beam_kind, diameter_m, ext are then ignored */
  int32_t beam_kind;            /* PRISIM_BEAM_* */
  int32_t reserved_;
  double diameter_m;
  const prisim_beam_ext* ext;   /* optional dipole / array factor / ground plane / beamformer; NULL = none */
} prisim_obs;

typedef struct prisim_snapshot {
  double lst_deg;               /* local sidereal time of the snapshot (:6113); used only when frame_given = 0 and the catalogue is RA-Dec */
  double pc_dircos[3];          /* pointing = phase centre, ENU direction cosines (:6155-6167) */
  double beam_pc_dircos[3];     /* beam pointing centre (zenith = {0,0,1}) */
  int32_t frame_given;          /* 1: cel2enu / aberr_beta below are the snapshot's frame; 0: the fall-back rotation described above */
  int32_t reserved_;
  double cel2enu[9];            /* row-major rotation: catalogue-frame unit vectors -> local East, North, Up (must be orthonormal to 1e-9) */
  double aberr_beta[3];         /* observer velocity / c in the CATALOGUE frame (|beta| < 0.01); zeros = no aberration */
} prisim_snapshot;

/* Make snapshot `snap` of the catalogue the current sky: geometry, compaction, (when long baselines can resolve sources out) the
 * altitude ordering and the cull table, flux spectra and pb * fluxes -- all on the device.  The geometry runs on a second,
 * high-priority stream into one of two buffer sets, so that it proceeds beside the previous snapshot's sky-sum; the call waits for
 * ITS small result record only, never for the compute stream.  *nsrc_roi (may be NULL) = sources inside the region of interest.
 * Follow with prisim_hip_compute. */
int prisim_hip_set_sky_from_catalog(prisim_ctx* ctx, const prisim_obs* obs, const prisim_snapshot* snap, int64_t* nsrc_roi);

/* The region of interest of an arbitrary snapshot, for class state that is read rarely (obs_catalog_indices :6377, geometric_delays
 * :6287-6291): indices [cap] int64 into the catalogue in catalogue order and dircos [cap][3] (either may be NULL); *nsrc_roi = the
 * count (fails with PRISIM_EINVAL when cap is smaller).  Runs in its own buffer set and synchronises only the geometry stream. */
int prisim_hip_catalog_roi(prisim_ctx* ctx, const prisim_obs* obs, const prisim_snapshot* snap, int64_t* nsrc_roi, int64_t* indices,
                           double* dircos, int64_t cap);

/* What follows every snapshot of prisim_hip_observe_catalog, queued behind that snapshot's sky-sum (on the copy / communication stream,
 * i.e. under the next snapshot's sky-sum). */
typedef struct prisim_post {
  void* host_vis;               /* page-locked host cube [nt_max][nbl][nchan] (prisim_hip_host_alloc) or NULL: slot t is downloaded into
                                   host_vis + t * nbl * nchan * (host_is_c64 ? 8 : 16) bytes (prisim_hip_get_vis_async) */
  int32_t host_is_c64;
  int32_t gather;               /* 1: prisim_hip_allgather_slot_async(slot, gather_as_c64) */
  int32_t gather_as_c64;
  int32_t reserved_;
} prisim_post;

/* nsnap snapshots in one call: the geometry of all of them first (chunks of snapshots, ONE readback per chunk), then for snapshot t
 * sky + compute into cube slot slot0 + t, queued back to back without any host synchronisation on the compute stream.  Arrays of at
 * most 256 baselines (uniform channel grid; analytic Gaussian / Airy / delta / dipole beams with or without array
 * factor and ground plane, or the external HEALPix beam; catalogues with or without source shapes, unless the taper culling could
 * shorten something) put the beam x flux, the packing and the sky-sums of a whole chunk of up to 256 snapshots into ONE launch each -- the
 * sky-sum's work item is (snapshot, baseline wave, channel tile, source split) -- and ONE reduction; the launch's per-snapshot table is
 * written by the geometry on the device, so such a chunk (a single snapshot included: nsnap = 1 is a chunk of one) is queued without the
 * host having seen a count, and the counts are read when everything is queued.  All three modes of interferometry.py:6320-6343 take that
 * launch: fp64; want_grad (visibilities + the three baseline-gradient sums, wave items of 16 baselines x 4 sources on the fp64 matrix
 * instruction); and precision = PRISIM_FP32 -- on arrays this small the ARITHMETIC of an fp32 request is done by the same fp64 launch (a
 * snapshot costs its launches, not its flops: HERA-19 39 us per snapshot against 106 through a per-snapshot fp32 chain), so the result
 * is the fp64 one, inside the fp32 tolerance by nine orders of magnitude; PRISIM_HIP_BATCH_FP32_AS_FP64=0 keeps fp32 arithmetic.  Replaces the loop of
 * interferometry.py:6641-6647 / scripts/run_prisim.py:2165-2207.  nsrc_roi: [nsnap] or NULL; post: NULL = nothing. */
int prisim_hip_observe_catalog(prisim_ctx* ctx, const prisim_obs* obs, const prisim_snapshot* snaps, int64_t nsnap, int precision,
                               int want_grad, int64_t slot0, int64_t* nsrc_roi, const prisim_post* post);

/* ---- delay transform (follow-on stage, interferometry.py:8052-8137) ---------------------- */

/* For slots [0, nt): x = V * bpwts (bpwts: [nbl][nchan] real weights = bp*bp_wts, or NULL = 1),
 * zero-pad by npad = int(nchan*pad) at the high-frequency end (:8123-8125), inverse FFT along
 * frequency (rocFFT), fftshift, scale by (nchan+npad)*df (:8125), keep every (1+pad)-th lag (:8132).
 * out: host complex128 [nt][nbl][nchan_out]; lags_out: [nchan_out] seconds (may be NULL).
 * power != 0 writes |.|^2 * power_scale into out_power [nt][nbl][nchan_out] float64
 * (delay_spectrum.py:3992-3993) instead of / in addition to out (either may be NULL). */
int prisim_hip_delay_transform(prisim_ctx* ctx, int64_t nt, const double* bpwts, double pad,
                               double* out, double* lags_out, double* out_power, double power_scale);

/* The same transform with the results left in HBM: lag spectra [nt][nbl][nchan_out] complex128 (want_lag) and/or delay power
 * float64 (want_power) of all nt snapshots stay resident in the context until read back (whole, or selected baselines) or
 * exchanged with prisim_hip_allgather_lags -- config 5's 120 x 61075 spectra never cross PCIe unless asked for.
 * bpwts: host window [wts_rows][nchan] with wts_rows = 1 (one window for every baseline: 8 KB instead of 500 MB at HERA-350)
 * or nbl, or NULL.  lags_out [nchan] / nout_out may be NULL.  Asynchronous on the context stream.
 * When 1 + pad is an integer and nchan is 256 R with R in {1, 2, 3, 4, 8, 16} the stage is ONE kernel (delay_kernels.hip): every
 * visibility is read once and every lag written once -- the kept samples of the zero-padded transform are exactly the
 * nchan-point transform; other shapes run window/pad -> rocFFT -> shift/decimate. */
int prisim_hip_delay_transform_device(prisim_ctx* ctx, int64_t nt, const double* bpwts, int64_t wts_rows, double pad, int want_lag, int want_power,
                                      double power_scale, double* lags_out, int64_t* nout_out);
/* Read back snapshots [t0, t0 + nt) of the resident spectra: out [nt][nrows][nchan_out] (complex128 / float64) for the baselines
 * listed in rows[nrows], or [nt][nbl][nchan_out] when rows is NULL.  Synchronises the stream. */
int prisim_hip_get_lags(prisim_ctx* ctx, int64_t t0, int64_t nt, const int64_t* rows, int64_t nrows, double* out);
int prisim_hip_get_delay_power(prisim_ctx* ctx, int64_t t0, int64_t nt, const int64_t* rows, int64_t nrows, double* out);

/* ---- phase-centre rotation (interferometry.py:7871-7877) --------------------------- */

/* For slots [0, nt): V[t][b][f] *= exp(-2 pi i f (b . diff_dircos[t]) / c), diff = current - new phase-centre
 * direction cosines per snapshot ([nt][3]).  In place on the device cube. */
int prisim_hip_phase_rotate(prisim_ctx* ctx, int64_t nt, const double* diff_dircos);

/* ---- thermal noise (interferometry.py:6661-6693) ----------------------------------- */

/* noise[t][b][f] = rms[t][b][f] / sqrt(2) * (n1 + i n2), n1, n2 ~ N(0,1) (interferometry.py:6692).  The normals come from
 * a counter-based generator (Philox-4x32-10 keyed by `seed`, counter = (t, bl_offset + b, f)), so a baseline-sharded run
 * draws exactly the numbers of the unsharded run.  rms: host [nt][nbl][nchan] float64 (vis_rms_freq, :6685-6689);
 * out: host complex128 [nt][nbl][nchan]. */
int prisim_hip_noise(prisim_ctx* ctx, int64_t nt, const double* rms, uint64_t seed, int64_t bl_offset, double* out);
/* The same for a shard whose baselines are not one contiguous range of the whole array (shards dealt round-robin in groups,
 * prisim_amd/sharding.py): bl_index[nbl] = global index of every local baseline. */
int prisim_hip_noise_indexed(prisim_ctx* ctx, int64_t nt, const double* rms, uint64_t seed, const int64_t* bl_index, double* out);

/* ---- multi-GPU: baseline shards + one RCCL all-gather ---------------------- */

/* 128-byte RCCL unique id; rank 0 creates it, the launcher distributes it out of band. */
int prisim_hip_comm_unique_id(char id[128]);
/* "librccl <major>.<minor>.<patch> (<path it was loaded from>)" of the RCCL this library dlopen()ed -- for run records. */
int prisim_hip_comm_version(char out[128]);
int prisim_hip_comm_init(prisim_ctx* ctx, const char id[128], int nranks, int rank);
/* For run records and for a watchdog around the two calls above (ncclCommInitRank is a collective: it blocks until every rank has
 * arrived, for ever when one never does): the PCI bus id of HIP device `device`, and the text of librccl's last error / warning
 * (ncclGetLastError).  prisim_hip_comm_last_error touches neither a context nor the HIP runtime and may be called from another thread
 * while prisim_hip_comm_init is still inside RCCL -- the one exception to "calls on one context are serialised" (it takes no context). */
int prisim_hip_device_pci(int device, char out[64]);
int prisim_hip_comm_last_error(char out[512]);
/* All-gather the local cube (equal-sized baseline shards, [nt][nbl_shard][nchan]) into a device cube
 * [nt][nranks][nbl_shard][nchan] held by the context (snapshot-major: every snapshot's full baseline set is
 * contiguous, rank blocks in rank order).  as_c64 = 0: complex128 on the wire; 1: the
 * shard is first rounded to complex64 on the device (the reference's memsave dtype, :6183) and
 * half the bytes cross xGMI.  Replaces the reference's per-rank _part_i.hdf5 files + rank-0
 * concatenate (scripts/run_prisim.py:2207, 2233-2242).  Asynchronous on the context stream. */
int prisim_hip_allgather(prisim_ctx* ctx, int64_t nt, int as_c64);
/* Same exchange for ONE snapshot slot, enqueued on a second HIP stream behind everything issued so far on the
 * compute stream: the RCCL transfer of snapshot t overlaps the sky-sum of snapshot t+1 (xGMI copies beside VALU work).
 * prisim_hip_sync / _get_gathered / _gathered_checksum wait for it. */
int prisim_hip_allgather_slot_async(prisim_ctx* ctx, int64_t slot, int as_c64);
/* Copy the gathered cube to the host: out [nt][nranks][nbl_shard][nchan] -- or [nt][nbl_total][nchan] in global baseline order once a
 * shard map is set (prisim_hip_set_shard_map) --, complex128 or complex64 according to the as_c64 of the last allgather. */
int prisim_hip_get_gathered(prisim_ctx* ctx, int64_t nt, void* out);
/* All-gather of the RESIDENT lag spectra (prisim_hip_delay_transform_device with want_lag) of nt snapshots, device to device:
 * the FFT runs along frequency, so every rank transforms its own baseline shard and the spectra are exchanged like the
 * visibilities; the gathered cube then holds [nt][nranks][nbl_shard][nchan_out] complex128. */
int prisim_hip_allgather_lags(prisim_ctx* ctx, int64_t nt);
/* Checksum (sum of all re,im accumulated in double, fixed reduction order) of the gathered cube,
 * computed on the device. */
int prisim_hip_gathered_checksum(prisim_ctx* ctx, int64_t nt, double* out);
/* The baseline-gradient cubes of a sharded run (gradient_mode='baseline'; the reference concatenates them over baseline chunks,
 * interferometry.py:8349-8350): gathers [nt][3][nbl_shard][nchan] into [nt][nranks][3][nbl_shard][nchan]; read with
 * prisim_hip_get_gathered (its rows are then 3 nchan long). */
int prisim_hip_allgather_grad(prisim_ctx* ctx, int64_t nt, int as_c64);
/* The gathered cube in the REFERENCE'S baseline order, on the device.  Shards need not be contiguous ranges of the array (this
 * repository deals groups of baselines round-robin so that every GPU gets its share of the long, expensive baselines, prisim_amd/sharding.py)
 * and are padded to equal size for the exchange.  bl_index: [nranks][nbl_shard] = global baseline of local row j of rank r, negative =
 * padding; every baseline in [0, nbl_total) must occur exactly once.  After this call every gather (allgather, allgather_slot_async,
 * allgather_lags, allgather_grad) lands in a staging block and a copy kernel behind it, on the same stream, writes the gathered cube as
 *   [nt][nbl_total][row]            (gradients: [nt][3][nbl_total][nchan])
 * -- the unsharded array's order, padding dropped: what the reference's rank-0 concatenate of its _part_i files leaves
 * (scripts/run_prisim.py:2233-2242).  prisim_hip_get_gathered / _gathered_checksum then speak of that cube.  bl_index = NULL goes back to
 * the rank-major layout.  Needs set_array (and comm_init when nranks > 1) first; set_array with another shard size drops the map. */
int prisim_hip_set_shard_map(prisim_ctx* ctx, const int64_t* bl_index, int64_t nbl_total);
/* Who receives the gathered cubes of every LATER gather call (allgather, allgather_slot_async, allgather_lags, allgather_grad):
 * root = -1 (default) every rank (ncclAllGather: each GPU ends up with the whole cube, 60-120 GB at config 5); root = r only rank r
 * (grouped ncclSend / ncclRecv: SURVEY 8(e) `gather_to_root`, the layout of the reference's rank-0 concatenate, run_prisim.py:2233-2242) --
 * the other ranks then allocate no gathered cube, and prisim_hip_get_gathered / _gathered_checksum fail there with PRISIM_ESTATE. */
int prisim_hip_set_gather_root(prisim_ctx* ctx, int root);
/* All-gather of `bytes` per rank filled with a rank-dependent pattern, read back and verified on the host: run once before any
 * timed or production exchange so that a communicator that cannot move data fails HERE (PRISIM_ELIB + message), not as a wrong cube.
 * Also valid on a 1-rank context without a communicator (device copy). */
int prisim_hip_comm_selftest(prisim_ctx* ctx, int64_t bytes);

/* What the overlapped per-snapshot gathers (prisim_hip_allgather_slot_async) cost, from hipEvent pairs on the communication stream. */
typedef struct prisim_comm_stats {
  int64_t n_gathers;                     /* gathers measured since the last reset */
  int64_t bytes_per_peer;                /* bytes of this rank's shard that every peer receives per gather (= bytes it receives from each) */
  double sum_gather_ms;                  /* start -> end on the communication stream (complex64 rounding kernel included), summed */
  double last_gather_ms, max_gather_ms;
  double last_gather_after_compute_ms;   /* end of the LAST gather minus end of the sky-sum it followed: the part no later compute hid */
  int32_t stream_priority;               /* priority the communication stream was created with (numerically lower = higher) ... */
  int32_t stream_priority_lowest;        /* ... and the lowest the device offers (the compute stream runs at default priority 0) */
  int32_t nranks;
  int32_t reserved_;
  double sum_undeal_ms;                  /* of sum_gather_ms: the un-deal copy kernels behind the gathers (0 without a shard map) */
  double last_undeal_ms;
} prisim_comm_stats;
int prisim_hip_get_comm_stats(prisim_ctx* ctx, prisim_comm_stats* out, int reset);

/* ---- host-visible results without a serial PCIe tail (interferometry.py:6384-6393: skyvis_freq lives on the host) ---------------- */

/* Page-locked host memory for asynchronous downloads (hipHostMalloc / hipHostFree). */
int prisim_hip_host_alloc(int64_t bytes, void** out);
int prisim_hip_host_free(void* p);
/* Enqueue the download of cube slot `slot` (and of its gradient block when grad != NULL) into caller memory that must stay valid
 * until prisim_hip_wait_downloads / prisim_hip_sync returns: the copy waits for everything issued so far on the compute stream and
 * runs on a copy stream, i.e. the PCIe transfer of snapshot t overlaps the sky-sum of snapshot t+1.  out_is_c64: rounded to
 * complex64 on the device first (half the bytes).  Memory from prisim_hip_host_alloc gives a true overlap; pageable memory works
 * but is staged by the runtime. */
int prisim_hip_get_vis_async(prisim_ctx* ctx, int64_t slot, void* vis, void* grad, int out_is_c64);
int prisim_hip_wait_downloads(prisim_ctx* ctx);

/* ---- timing / introspection -------------------------------------------------------------- */

typedef struct prisim_timing {
  double last_kernel_ms;     /* hipEvent duration of the dominant sky-sum kernel of the last compute() */
  double last_compute_ms;    /* hipEvent duration of the whole last compute() (pack + sum + reduce) */
  double sum_kernel_ms;      /* accumulated over compute() calls since the last reset */
  int64_t n_kernel;          /* number of sky-sum kernel launches accumulated */
  int64_t last_terms;        /* nbl*nchan*nsrc of the last compute() */
  int32_t last_kernel_id;    /* PRISIM_KERNEL_* actually used */
  int32_t last_chan_tile;    /* channels per thread of the recurrence kernel */
  int32_t last_nsplit;       /* source split factor */
  int32_t last_lift_groups;  /* baseline groups (of 256) that ran the lifting (three-shear) rotation; 0 for the packed fp32 taper kernel */
  int32_t last_taper_group;  /* 1: the packed fp32 taper kernel ran its grouped recurrence (df/f_min <= 3.4e-3), 0: exact per-step form */
  int32_t last_delay_fused;  /* 1: the last delay_transform_device ran the fused LDS FFT kernel, 0: the rocFFT pipeline */
  double last_delay_ms;      /* hipEvent duration of the last prisim_hip_delay_transform_device (all batches) */
  int32_t last_taper_split;  /* > 0: the packed fp32 taper ran its split form over this many source runs of one source size each */
  int32_t last_split_uncorrected_groups;   /* (source run, baseline group) pairs whose parabola bound allowed the uncorrected body */
  double last_culled_fraction;             /* share of the snapshot's (source, baseline) pairs the taper culling skipped (packed fp32
                                              kernels, grouped fp64 taper kernel): their summed contribution is below exp(-18) (fp32) /
                                              exp(-28) (fp64) of sum|pbflux| (0: none) */
  int32_t last_batch_snapshots;            /* snapshots whose sky-sums shared the last launch (prisim_hip_observe_catalog on arrays of at most
                                              256 baselines: the whole chunk in one launch; last_terms and the kernel times then cover all of them) */
  int32_t reserved_;
} prisim_timing;

int prisim_hip_sync(prisim_ctx* ctx);
int prisim_hip_get_timing(prisim_ctx* ctx, prisim_timing* out, int reset);
/* Device properties: CU count and clock (kHz) used to re-derive the VALU peak on the box. */
int prisim_hip_device_info(prisim_ctx* ctx, int* cu_count, int* clock_khz, char name[64]);
/* Tuning knobs (0 = library default): channels per thread, source padding / split granularity, source split factor. */
int prisim_hip_set_tuning(prisim_ctx* ctx, int chan_tile, int src_chunk, int nsplit);

#ifdef __cplusplus
}
#endif
#endif /* PRISIM_HIP_H */
