// skyvis_kernels.h -- internal launcher interface between the C-ABI layer (capi.cpp) and the
// HIP kernels (skyvis_kernels.hip, beam_kernels.hip).  Not part of the public ABI.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace prisim {

static constexpr int kBlockThreads = 256;   // 4 wavefronts; lanes = baselines

// One snapshot of a batched pass (many snapshots per launch; device table, one entry per snapshot)
struct BatchSnap {
  int64_t dir0;              // first row of the snapshot's directions / catalogue indices in the geometry set
  int64_t nsrc;              // sources inside its region of interest
  int64_t pb0;               // first row of its beam x flux block in pb
  int64_t row0;              // first row of its block in the packed rows / prepared directions
  int64_t nrow;              // rows of that block (nsrc rounded up; zero rows past nsrc)
  int64_t src_per_split;     // wave items: sources per split
  double pc[3];              // phase centre
  double bpc[3];             // beam pointing centre
  double* out;               // its sums: the cube slot (one split) or its nsplit partial cubes
  double* gout;              // gradient batches: its three gradient sums [3][nbl][nchan] -- the gradient slot or nsplit partial sets
};

struct SkyvisParams {
  // array (resident)
  const double* bl_x;        // [nbl] metres, East
  const double* bl_y;        // [nbl] North
  const double* bl_z;        // [nbl] Up
  int64_t nbl;
  int64_t nchan;
  double f0;                 // Hz, channel 0 (recurrence kernel: uniform grid f0 + k*df)
  double df;                 // Hz
  double inv_c;              // 1 / 299792458
  // sky (per snapshot)
  const double* dirs;        // [nsrc][4]: l, m, n, kappa = ln2 * (2 sin(fwhm/2))^2   (raw)
  const double* dirs_prep;   // [nsrc_pad][4]: (l-lpc)/c, (m-mpc)/c, (n-npc)/c, kappa; zero rows past nsrc
  const void* pb_packed;     // [ntiles][nsrc_pad][CT] of T, zero rows past nsrc
  const float* fsq;          // [npad] (f_k*1e-8)^2*log2(e) (fp32 taper)
  const float* fsq_pairs;    // [ntiles][CT] the same, in the (up,down) pair order of k_skyvis_rec_f32pk
  const int32_t* lift_flags; // [nbgroups] 1: the lifting rotation is safe for this baseline group (or nullptr)
  double fsq_scale;          // 1e16
  int64_t nsrc;
  int64_t nsrc_pad;          // nsrc rounded up to a multiple of src_chunk
  double pc_x, pc_y, pc_z;   // phase-centre direction cosines
  int32_t taper;
  // decomposition
  int32_t ntiles;            // channel tiles of CT channels
  int32_t nbgroups;          // baseline groups of kBlockThreads
  int32_t nsplit;            // source split factor (partials reduced afterwards)
  int64_t src_per_split;
  int32_t src_chunk;         // granularity of the zero padding / of the source split (host side; the kernels do not use it)
  int32_t flush_src;         // fp32: flush accumulators into the fp64 cube every this many sources
  int32_t scale_comp;        // direct kernel: gradient component or -1
  int32_t taper_group;       // packed fp32 taper: 1 = grouped recurrence (valid when df/f_min <= 3.4e-3), 0 = exact per-step form
  double* out;               // [nsplit][nbl][nchan] complex128 (nsplit==1: the cube slot itself)
  int32_t out_f32;           // 1: out is [nsplit][nbl][nchan] complex64 partial sums (fp32 run, every split flushes exactly once)
  int32_t pad2_;
  double* grad_out;          // fused gradient kernels: [3][nbl][nchan] complex128 of this slot
  const float* dirs_c32;     // fused fp32 gradient kernel: [nsrc_pad][8] (l, l, m, m, n, n, 0, 0)
  // packed fp32 kernels: the source range of this launch (the whole sky: 0, nsrc) and whether an earlier launch already wrote the slot
  int64_t src_lo, src_hi;
  int32_t accumulate;        // 1: the first flush adds to what the output slot holds (a later source range of the same snapshot)
  int32_t pad3_;
  // split taper form (k_skyvis_rec_f32pk_split): one source size for the whole range
  double kappa0;             // ln2 (2 sin(fwhm/2))^2 of every source in [src_lo, src_hi)
  const int32_t* split_flags;// [nbgroups] bit 0: small step angle, bit 1: keep the parabola correction
  // taper culling (recurrence kernels with the taper): the sources [range start, src_first[bg]) contribute less than the precision's
  // cull threshold to every baseline of group bg and are skipped; nullptr = none
  const int32_t* src_first;
  // wave items (k_skyvis_taper_f64_wave, nbl <= 256): wave_nbw = ceil(nbl / 64) baseline waves x wave_nsplit source splits, four items
  // per block; 0 = block items
  int32_t wave_nbw;
  int32_t wave_nsplit;
  // wave items over a batch of snapshots (k_skyvis_taper_f64_wave_batch): item g = (snapshot, split, baseline wave)
  const BatchSnap* wave_snaps;
  int32_t wave_nsnap;
  int32_t pad4_;
};

hipError_t launch_skyvis_rec(const SkyvisParams& p, bool f32, int ct, hipStream_t stream);
hipError_t launch_skyvis_rec_f32pk(const SkyvisParams& p, int ct, hipStream_t stream);
// split taper form for a source range of one source size (ct = 64; p.src_lo/src_hi, p.kappa0, p.split_flags, p.accumulate)
hipError_t launch_skyvis_rec_f32pk_split(const SkyvisParams& p, int ct, hipStream_t stream);
// fp64 sky-sum with the taper in the grouped form (ct = 16 or 32; rows packed in NATURAL channel order: launch_pack interleave = 0;
// p.src_lo/src_hi, p.src_first, p.accumulate as for the packed fp32 kernels)
hipError_t launch_skyvis_taper_f64(const SkyvisParams& p, int ct, hipStream_t stream);
hipError_t launch_skyvis_taper_f64_wave_batch(const SkyvisParams& p, int ct, hipStream_t stream);
hipError_t launch_skyvis_grad_taper_f64_batch(const SkyvisParams& p, hipStream_t stream);
hipError_t launch_pack_prep_batch(const double* pb, double* packed, int64_t pitch, int64_t max_nrow, int64_t nchan, int ct, int ntiles, const double* dirs,
                                  double* prep, double inv_c, const BatchSnap* snaps, int nsnap, hipStream_t stream);
hipError_t launch_reduce_partials_batch(const double* part, double* out, int64_t n2, int nsplit, int nsnap, hipStream_t stream);
// beam-weighted sky moments of sources [s_lo, s_hi) per channel into out[4][nchan] (device), see k_taper_moments
// flags of the split taper kernel's baseline groups from those moments, on the device (k_split_flags); *count += uncorrected groups
hipError_t launch_split_flags(const double* mom, int64_t nchan, const double* grp_h, const double* grp_z, const int32_t* lift_flags, int nbg,
                              double c16, double limit, int32_t* flags, int32_t* count, hipStream_t stream);
// part: scratch [taper_moments_chunks(s_lo, s_hi)][4][nchan] doubles (per-chunk partial sums, added in chunk order: deterministic)
int64_t taper_moments_chunks(int64_t s_lo, int64_t s_hi);
hipError_t launch_taper_moments(const double* pb, const double* dirs, int64_t s_lo, int64_t s_hi, int64_t nchan, double* part, double* out,
                                hipStream_t stream);
// V + baseline gradient in one pass (fp64, MFMA 4x4x4): p.nbgroups = groups of 64 baselines, p.nsplit = 1, ct = 16 or 32
hipError_t launch_skyvis_grad_f64(const SkyvisParams& p, int ct, hipStream_t stream);
// the same in packed fp32 (VALU; p.nbgroups = groups of 256 baselines, 16-channel tiles, p.nsplit = 1; with the taper: rows from
// launch_pack + p.dirs_c32; without: rows from launch_pack_grad)
hipError_t launch_skyvis_grad_f32(const SkyvisParams& p, hipStream_t stream);
hipError_t launch_fsq_pairs(const float* fsq, float* pairs, int ct, int ntiles, hipStream_t stream);
hipError_t launch_skyvis_direct(const SkyvisParams& p, const double* freqs, const double* pb, const double* scale,
                                hipStream_t stream);
// (s_lo, s_hi): only the rows of these sources are written (default: all nsrc_pad rows)
hipError_t launch_pack(const double* pb, void* packed, bool f32, int64_t nsrc, int64_t nsrc_pad, int64_t nchan, int ct,
                       int ntiles, const double* dirs, int scale_comp, int interleave, hipStream_t stream, int64_t s_lo = 0, int64_t s_hi = -1);
// rows of the fused fp32 gradient kernel without the taper: [ntiles(16 ch)][nsrc_pad][64] floats, pre-multiplied by (1, l, m, n)
// launch_pack (whole sky, no scaling) + launch_prep_dirs (no c32) in one launch
hipError_t launch_pack_prep(const double* pb, void* packed, bool f32, int64_t nsrc, int64_t nsrc_pad, int64_t nchan, int ct, int ntiles,
                            int interleave, const double* dirs, double* prep, double pcx, double pcy, double pcz, double inv_c,
                            hipStream_t stream);
hipError_t launch_pack_grad(const double* pb, float* packed, int64_t nsrc, int64_t nsrc_pad, int64_t nchan, int ntiles, const double* dirs,
                            hipStream_t stream);
hipError_t launch_prep_dirs(const double* dirs, double* prep, float* c32 /*[nsrc_pad][8] or NULL*/, int64_t nsrc, int64_t nsrc_pad, double pcx,
                            double pcy, double pcz, double inv_c, hipStream_t stream);
hipError_t launch_reduce_partials(const void* part, bool part_f32, double* out, int64_t n2, int nsplit, hipStream_t stream);
hipError_t launch_f32_to_f64(const float* in, double* out, int64_t n, hipStream_t stream);
hipError_t launch_fsq(const double* freqs, float* fsq, int64_t nchan, int64_t npad, double scale, hipStream_t stream);

// beam_kernels.hip
struct BeamParams {
  const double* dirs;        // [nsrc][4]
  const double* flux_ref;    // [nsrc]
  const double* spindex;     // [nsrc]
  const double* flux_spec;   // [nsrc][nchan] or nullptr (then power law)
  const double* freqs;       // [nchan]
  double ref_freq;
  int32_t beam_kind;
  double diameter;
  double bpc_x, bpc_y, bpc_z;  // beam pointing centre
  // optional factors (prisim_beam_ext)
  double dip_x, dip_y, dip_z; int32_t dipole_mode;
  int32_t nax1, nax2; double sep1, sep2, rot_c, rot_s; double apc_x, apc_y, apc_z;
  double gp_height; int32_t gp_modify; double gp_scale, gp_max;
  int32_t bf_nelem, bf_nrand;  // beamformer (device arrays): positions [n][3], delays [n][nrand], gains [n][nrand]
  const double* bf_pos; const double* bf_delays; const double* bf_gains;
  double poly[4]; int32_t* flag;   // PRISIM_BEAM_POLY coefficients; flag bit 0: value >= 1.01, bit 1: NaN
  int64_t nsrc, nchan;
  double* pb_out;            // [nsrc][nchan]
  const int32_t* src_index;  // catalogue path: flux_ref / spindex / flux_spec rows are read at src_index[s] (nullptr: at s)
  const BatchSnap* batch;    // many snapshots in one launch (blockIdx.y): directions / indices from row dir0, output rows from pb0, nsrc and the
                             // beam pointing of that snapshot
};
hipError_t launch_beam_flux(const BeamParams& p, hipStream_t stream);
hipError_t launch_beam_flux_batch(const BeamParams& p, int nsnap, hipStream_t stream);
hipError_t launch_mul_inplace(double* a, const double* b, int64_t n, hipStream_t stream);
// external HEALPix beam (aux_kernels.hip)
hipError_t launch_extbeam_table(const double* beam, const double* interp, double* table, int64_t npix, int64_t nfreq,
                                int64_t nchan, hipStream_t stream);
hipError_t launch_extbeam_sky(const double* table, int nside, const double* dirs, const double* fluxes /*[nsrc][nchan] or NULL*/,
                              const double* flux_ref /*[nsrc]*/, const double* spindex /*[nsrc]*/, const double* freqs, double ref_freq,
                              double* work /*[nsrc][nchan]*/, double* colmax_scratch /*[1024*nchan + nchan]*/, double* pb_out,
                              int64_t nsrc, int64_t nchan, hipStream_t stream, const int32_t* src_index = nullptr);
constexpr int kExtBatchBlocks = 64;      // row blocks of the per-snapshot column maximum in a batched external-beam launch
hipError_t launch_extbeam_sky_batch(const double* table, int nside, const double* dirs, const double* fluxes, const double* flux_ref,
                                    const double* spindex, const double* freqs, double ref_freq, double* work, double* colmax_scratch,
                                    double* pb_out, int64_t nsrc_max, int64_t nchan, const int32_t* src_index, const BatchSnap* batch, int nsnap,
                                    hipStream_t stream);

// device-resident catalogue: per-snapshot geometry (catalog_kernels.hip)
static constexpr int PRISIM_CAT_RADEC = 0, PRISIM_CAT_HADEC = 1, PRISIM_CAT_ALTAZ = 2;     // = PRISIM_COORDS_* of the public header
static constexpr int PRISIM_CAT_MAX_RUNS = 8;
struct CatSnap {             // per-snapshot inputs (device array, one entry per snapshot of a batch)
  double rot[9];             // catalogue frame -> local East-North-Up, row-major (prisim_snapshot.cel2enu, or built from lst / latitude)
  double beta[3];            // aberration vector in the catalogue frame (observer velocity / c; zeros = none)
  double roi_pc[3];          // centre of the region of interest (roi_center = pointing centre)
  double pc[3];              // phase centre (for max |s - s_pc|)
  double bpc[3];             // beam pointing centre (the batched launch's per-snapshot table is built on the device from this)
};
struct CatOut {              // per-snapshot results the host reads back (pinned memory)
  int64_t nsrc;              // sources inside the region of interest
  int64_t run_start[PRISIM_CAT_MAX_RUNS + 1];   // first compacted source of every catalogue run; [nruns] = nsrc
  uint64_t dmax2_bits;       // max |s - s_pc|^2, bit pattern of a non-negative double
};
struct CatGeomParams {
  const double* ux;          // [n] catalogue unit vectors in the catalogue's own frame (RA-Dec / HA-Dec: cos d cos a, cos d sin a, sin d;
  const double* uy;          //     alt-az: East-North-Up direction cosines)
  const double* uz;
  const double* kappa;       // [n] ln2 (2 sin(fwhm/2))^2, or nullptr (no source shapes)
  const uint8_t* run_id;     // [n] catalogue run of every source (runs of one source size), or nullptr
  int64_t n;
  int64_t nblocks;           // ceil(n / 256)
  int32_t roi_center;        // 0 zenith, 1 pointing centre
  int32_t want_keys;         // also write the altitude keys of the culling order
  double sin_alt_min;        // sin(90 - roi_radius): 'zenith' keeps n >= this         (interferometry.py:6216)
  double cos_radius;         // cos(roi_radius): 'pointing_center' keeps s . s_pc >= this  (:6211)
  const CatSnap* snaps;      // [nsnap] device
  int32_t* block_off;        // [nsnap][nblocks] scratch
  int32_t* idx;              // [nsnap][n] compacted catalogue indices (catalogue order)
  double* dirs;              // [nsnap][n][4] compacted l, m, n, kappa
  uint32_t* keys;            // [nsnap][n] (want_keys)
  uint32_t* pos;             // [nsnap][n] (want_keys) 0, 1, 2, ...
  CatOut* out;               // [nsnap] device
  // The per-snapshot table of a batched launch (BatchSnap, small arrays), written by the scan pass from the counts it has just formed --
  // so that beam x flux, packing and the sky-sums of the chunk can be queued WITHOUT the host having seen the counts.  Every snapshot owns
  // a fixed-size block: directions at t * n, beam x flux rows at t * n, packed rows at t * batch_npad (n rounded up to 4).
  BatchSnap* batch;          // [nsnap] device, or nullptr
  int64_t batch_npad;
  int32_t batch_nsplit;      // source splits of the sky-sum launch
  int32_t pad2_;
  double* batch_out;         // nsplit == 1: cube slot of snapshot 0 of the chunk; else the partial cubes [nsnap][nsplit][slot]
  int64_t batch_slot_elems;  // doubles per snapshot slot (nbl * nchan * 2)
  double* batch_gout;        // gradient batches: gradient slot of snapshot 0 of the chunk / the gradient partial sets [nsnap][nsplit][3 slot]; else nullptr
  int32_t inline_snap;       // 1: the (single) snapshot's inputs are `snap0` below, not snaps[0] -- no host-to-device copy in front of the kernel
  int32_t small_form;        // 1: k_cat_small (one block per snapshot, records written to page-locked host memory); 0: the three passes
  CatSnap snap0;
};
static constexpr int64_t kCatSmallMax = 16384;     // catalogues up to this size (of arrays of at most 256 baselines): a snapshot's geometry in ONE block
struct CullParams {
  const double* dirs;        // [nsrc][4] in upload order
  int64_t run_lo[PRISIM_CAT_MAX_RUNS], run_hi[PRISIM_CAT_MAX_RUNS];
  double run_kappa[PRISIM_CAT_MAX_RUNS];
  int32_t nruns, ng;
  const double* grp_minh;    // [ng] smallest horizontal baseline length of the group
  const double* grp_maxz;    // [ng] largest |b_z|
  double fc2;                // (f_min / c)^2
  int64_t nbl;
  int32_t* first;            // [2][nruns][ng]
  uint64_t* culled;          // [2] skipped (source, baseline) pairs per precision
};
int64_t cat_blocks(int64_t n);
hipError_t launch_cat_prepare(const double* lon_deg, const double* lat_deg, int coords, double* ux, double* uy, double* uz, int64_t n, hipStream_t stream);
hipError_t launch_cat_geometry(const CatGeomParams& p, int nsnap, hipStream_t stream);
size_t cat_sort_temp_bytes(int64_t n);
hipError_t launch_cat_sort(void* temp, size_t temp_bytes, const uint32_t* keys, uint32_t* keys_out, const uint32_t* pos, uint32_t* perm,
                           const double* dirs, const int32_t* idx, double* dirs_out, int32_t* idx_out, int64_t n, hipStream_t stream);
hipError_t launch_cull_first(const CullParams& p, hipStream_t stream);
hipError_t launch_lift_flags(const double* grp_maxlen, double k, double limit, int32_t* flags, int ng, hipStream_t stream);

// delay transform helpers (delay_kernels.hip)
hipError_t launch_dt_prepare(const double* cube, const double* bpwts /*device [wts_rows][nchan] or NULL*/, int64_t wts_rows /*1 or nbl*/,
                             double* work, int64_t nrows, int64_t nbl, int64_t nchan, int64_t nfft, hipStream_t stream);
hipError_t launch_dt_finish(const double* work, double* out, double* out_power, int64_t nrows, int64_t nfft,
                            int64_t nout, double factor, double scale, double power_scale, hipStream_t stream);
// fused delay transform for power-of-two channel counts and integer 1 + pad (delay_kernels.hip)
bool delay_fft_supported(int64_t nchan);
hipError_t launch_delay_fft(const double* cube, const double* bpwts /*device [wts_rows][nchan] or NULL*/, int64_t wts_rows /*1 or nbl*/,
                            const double* tw /*[nchan/2] complex*/, double* out, double* out_pow, int64_t nrows, int64_t nbl, int64_t nchan,
                            double scale, double power_scale, int cu_count, hipStream_t stream);
hipError_t launch_phase_rotate(double* cube, const double* blx, const double* bly, const double* blz, const double* freqs,
                               const double* diff /*[nt][3] device*/, int64_t nt, int64_t nbl, int64_t nchan, hipStream_t stream);
hipError_t launch_noise(const double* rms, double* out, int64_t nbl, int64_t nchan, int64_t t, int64_t bl_offset, uint64_t seed,
                        hipStream_t stream);
hipError_t launch_checksum(const void* data, bool is_f32, int64_t n, double* out, hipStream_t stream);
hipError_t launch_f64_to_f32(const double* in, float* out, int64_t n, hipStream_t stream);
hipError_t launch_undeal(const void* stage, void* out, const int64_t* map /*[nranks][nbl_shard] device*/, int nranks, int64_t nbl_shard,
                         int64_t nbl_total, int planes, int64_t row_words, hipStream_t stream);

}  // namespace prisim
