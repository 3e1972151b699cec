// catalog.cpp -- the device-resident catalogue path of libprisim_hip.so (include/prisim_hip.h, "device-resident catalogue").
//
// prisim_hip_set_catalog uploads a run's sky model once; prisim_hip_set_sky_from_catalog / prisim_hip_observe_catalog then form
// every snapshot's sky on the device (catalog_kernels.hip) -- what InterferometerArray.observe() of the reference does on the host at
// every call (prisim/interferometry.py:6171-6180, 6204-6219, 6249-6254, 6263).  Host work per snapshot: a handful of launches and one
// small read-back (source count, run boundaries, max |s - s_pc|) that waits for the GEOMETRY stream only.
#include <cmath>
#include <chrono>

#include "ctx_internal.h"

namespace {

using Clock = std::chrono::steady_clock;

// PRISIM_HIP_TRACE_ALLOC (development hook, with the allocation trace of ensure()): host time of the steps of a catalogue-path call
struct HostSpan {
  const char* what;
  Clock::time_point t0;
  explicit HostSpan(const char* w) : what(w), t0(Clock::now()) {}
  ~HostSpan() {
    static const bool trace = getenv("PRISIM_HIP_TRACE_ALLOC") != nullptr;
    if (trace) fprintf(stderr, "[prisim_hip host] %s: %.1f us\n", what, 1e6 * std::chrono::duration<double>(Clock::now() - t0).count());
  }
};

constexpr double kCullThr[2] = {28.0, 18.0};      // index = precision (PRISIM_FP64 = 0, PRISIM_FP32 = 1); see upload_common (capi.cpp)

}  // namespace

namespace pint {
// The streams, events and the first pinned block of the catalogue path.  prisim_hip_create calls this (two priority streams cost
// 2.5 ms each to create -- 5 ms that used to sit in front of a run's first snapshot); cat_runtime calls it again if that failed.
int catalog_streams(prisim_ctx* ctx) {
  auto& C = ctx->cat;
  if (C.gstream) return PRISIM_OK;
  HostSpan sp0("catalog_streams: two priority streams, events, first pinned block");
  int least = 0, greatest = 0;
  if (hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess) { least = greatest = 0; (void)hipGetLastError(); }
  if (hipStreamCreateWithPriority(&C.gstream, hipStreamNonBlocking, greatest) != hipSuccess) {
    (void)hipGetLastError();
    C.gstream = nullptr;
    HIPCHK(ctx, hipStreamCreateWithFlags(&C.gstream, hipStreamNonBlocking));
  }
  // the preparation stream is a stream of its own (same priority): the geometry of snapshot t+1, whose small record the host waits for,
  // does not queue behind the preparation of snapshot t, which a sky-sum in progress can hold up for milliseconds
  if (!ctx->prep_stream && hipStreamCreateWithPriority(&ctx->prep_stream, hipStreamNonBlocking, greatest) != hipSuccess) {
    (void)hipGetLastError();
    ctx->prep_stream = nullptr;
    HIPCHK(ctx, hipStreamCreateWithFlags(&ctx->prep_stream, hipStreamNonBlocking));
  }
  for (SkyBufs& k : ctx->skb) {
    if (!k.ev_prep) HIPCHK(ctx, hipEventCreateWithFlags(&k.ev_prep, hipEventDisableTiming));
    if (!k.ev_sum) HIPCHK(ctx, hipEventCreateWithFlags(&k.ev_sum, hipEventDisableTiming));
  }
  if (!C.ev_geom) HIPCHK(ctx, hipEventCreateWithFlags(&C.ev_geom, hipEventDisableTiming));
  if (!C.ev_join) HIPCHK(ctx, hipEventCreateWithFlags(&C.ev_join, hipEventDisableTiming));
  for (auto& e : C.ev_tab) { if (!e) HIPCHK(ctx, hipEventCreateWithFlags(&e, hipEventDisableTiming)); }
  for (auto& e : C.ev_tabfree) { if (!e) HIPCHK(ctx, hipEventCreateWithFlags(&e, hipEventDisableTiming)); }
  for (auto& s : C.set) {
    if (!s.ev_free) HIPCHK(ctx, hipEventCreateWithFlags(&s.ev_free, hipEventDisableTiming));
    if (!s.ev_prepared) HIPCHK(ctx, hipEventCreateWithFlags(&s.ev_prepared, hipEventDisableTiming));
  }
  if (!C.culled_host) {
    if (hipHostMalloc((void**)&C.culled_host, 2 * sizeof(uint64_t), hipHostMallocDefault) != hipSuccess) {
      C.culled_host = nullptr;
      return fail(ctx, PRISIM_ENOMEM, "hipHostMalloc for the catalogue path failed");
    }
    C.culled_host[0] = C.culled_host[1] = 0;
  }
  return PRISIM_OK;
}
}  // namespace pint

namespace {

int cat_runtime(prisim_ctx* ctx, int64_t nsnap) {
  auto& C = ctx->cat;
  if (!C.gstream || !ctx->prep_stream || !C.culled_host) {
    if (C.gstream && (!ctx->prep_stream || !C.culled_host)) { (void)hipStreamDestroy(C.gstream); C.gstream = nullptr; }      // a half-made set: again
    int rc = catalog_streams(ctx);
    if (rc) return rc;
  }
  if (nsnap > C.cap_snaps) {
    HostSpan sp2("cat_runtime: per-snapshot pinned records + device tables");
    if (C.out_host) { (void)hipHostFree(C.out_host); C.out_host = nullptr; }
    if (C.snaps_host) { (void)hipHostFree(C.snaps_host); C.snaps_host = nullptr; }
    C.cap_snaps = 0;
    const int64_t cap = std::max<int64_t>(nsnap, 16);
    if (hipHostMalloc((void**)&C.out_host, (size_t)cap * sizeof(CatOut), hipHostMallocDefault) != hipSuccess ||
        hipHostMalloc((void**)&C.snaps_host, (size_t)cap * sizeof(CatSnap), hipHostMallocDefault) != hipSuccess)
      return fail(ctx, PRISIM_ENOMEM, "hipHostMalloc for the catalogue path failed");
    int rc;
    if ((rc = ensure(ctx, C.snaps, (size_t)cap * sizeof(CatSnap))) || (rc = ensure(ctx, C.out_dev, (size_t)cap * sizeof(CatOut)))) return rc;
    C.cap_snaps = cap;
  }
  return PRISIM_OK;
}

// Does the altitude ordering pay?  The condition of InterferometerArray._cull_order (prisim_amd/interferometry.py): source shapes in at
// most 8 runs of one size each and kappa_max (|b|_max f_min / c)^2 >= 18 -- otherwise no baseline group can cull anything.
bool cat_sort_wanted(const prisim_ctx* ctx) {
  const auto& C = ctx->cat;
  if (!C.have_shape || C.runs.empty() || C.n < 2) return false;
  if (const char* env = getenv("PRISIM_HIP_TAPER_CULL")) { if (atoi(env) == 0) return false; }
  double lmax = 0.0;
  for (double v : ctx->grp_maxlen) lmax = std::max(lmax, v);
  const double fmin = ctx->h_freqs.empty() ? 0.0 : std::min(std::fabs(ctx->h_freqs.front()), std::fabs(ctx->h_freqs.back()));
  const double x = lmax * fmin / kC;
  return C.kappa_max * x * x >= 18.0;
}

// The frame of a snapshot the caller gave no frame for (frame_given = 0): the plain rotation -- RA-Dec: tilt(latitude) . rot_z(lst), i.e. hour
// angle = LST - RA (a catalogue in the true equator and equinox of date); HA-Dec: the tilt alone on (cos d cos H, cos d sin H, sin d)
// (GEOM.hadec2altaz, interferometry.py:6176-6177); alt-az: identity.  prisim_amd/frames.py equatorial_to_enu / hadec_to_enu are the same matrices.
void fallback_frame(int coords, double lst_deg, double latitude_deg, double rot[9], double beta[3]) {
  beta[0] = beta[1] = beta[2] = 0.0;
  const double lat = latitude_deg * (M_PI / 180.0);
  const double sl = std::sin(lat), cl = std::cos(lat);
  if (coords == PRISIM_COORDS_ALTAZ) {
    const double id[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    memcpy(rot, id, sizeof(id));
  } else if (coords == PRISIM_COORDS_HADEC) {
    const double m[9] = {0.0, -1.0, 0.0, -sl, 0.0, cl, cl, 0.0, sl};
    memcpy(rot, m, sizeof(m));
  } else {
    const double a = lst_deg * (M_PI / 180.0);
    const double c = std::cos(a), s = std::sin(a);
    // tilt . r3(a):  r3 = [[c, s, 0], [-s, c, 0], [0, 0, 1]],  tilt = [[0, 1, 0], [-sl, 0, cl], [cl, 0, sl]]
    const double m[9] = {-s, c, 0.0, -sl * c, -sl * s, cl, cl * c, cl * s, sl};
    memcpy(rot, m, sizeof(m));
  }
}

int check_frame(prisim_ctx* ctx, const prisim_snapshot& sn) {
  if (!sn.frame_given) return PRISIM_OK;
  for (int i = 0; i < 9; ++i) if (!std::isfinite(sn.cel2enu[i])) return fail(ctx, PRISIM_EINVAL, "non-finite cel2enu");
  double b2 = 0.0;
  for (int i = 0; i < 3; ++i) { if (!std::isfinite(sn.aberr_beta[i])) return fail(ctx, PRISIM_EINVAL, "non-finite aberr_beta"); b2 += sn.aberr_beta[i] * sn.aberr_beta[i]; }
  if (b2 >= 1e-4) return fail(ctx, PRISIM_EINVAL, "aberr_beta must be a velocity / c with |beta| < 0.01");
  for (int r = 0; r < 3; ++r)
    for (int q = r; q < 3; ++q) {
      double d = 0.0;
      for (int k = 0; k < 3; ++k) d += sn.cel2enu[3 * r + k] * sn.cel2enu[3 * q + k];
      if (std::fabs(d - (r == q ? 1.0 : 0.0)) > 1e-9) return fail(ctx, PRISIM_EINVAL, "cel2enu is not a rotation matrix (rows must be orthonormal to 1e-9)");
    }
  return PRISIM_OK;
}

int check_obs(prisim_ctx* ctx, const prisim_obs* obs, const prisim_snapshot* snaps, int64_t nsnap) {
  if (!obs || !snaps) return fail(ctx, PRISIM_EINVAL, "obs / snapshot is NULL");
  if (!ctx->array_set) return fail(ctx, PRISIM_ESTATE, "set_array must be called before the catalogue path");
  if (!ctx->cat.loaded) return fail(ctx, PRISIM_ESTATE, "set_catalog must be called first (set_array drops the catalogue)");
  if (!std::isfinite(obs->latitude_deg) || !std::isfinite(obs->roi_radius_deg)) return fail(ctx, PRISIM_EINVAL, "non-finite latitude / roi_radius");
  if (obs->roi_center != 0 && obs->roi_center != 1) return fail(ctx, PRISIM_EINVAL, "roi_center must be 0 (zenith) or 1 (pointing centre)");
  for (int64_t t = 0; t < nsnap; ++t) {
    if (!snaps[t].frame_given && !std::isfinite(snaps[t].lst_deg)) return fail(ctx, PRISIM_EINVAL, "non-finite LST");
    if (int rc = check_frame(ctx, snaps[t])) return rc;
    for (int i = 0; i < 3; ++i)
      if (!std::isfinite(snaps[t].pc_dircos[i]) || !std::isfinite(snaps[t].beam_pc_dircos[i])) return fail(ctx, PRISIM_EINVAL, "non-finite pointing / phase centre");
  }
  if (obs->use_external_beam) {
    if (ctx->ext_nside <= 0) return fail(ctx, PRISIM_ESTATE, "set_external_beam must be called before use_external_beam");
    return PRISIM_OK;
  }
  return check_beam_spec(ctx, obs->beam_kind, obs->diameter_m, snaps[0].beam_pc_dircos, obs->ext);
}

// A sky is about to be prepared on the preparation stream.  Between two such skies the events of the buffer sets order everything
// (ev_prep / ev_sum); but when the previous sky was prepared in line on the compute stream -- an uploaded sky, a batched chunk, the A/B
// hook -- nothing does: its sums may still be reading the set, the external-beam work area or the sort's temporaries this preparation is
// about to write.  So the first preparation after such a sky waits for everything queued on the compute stream so far.
int enter_prep_stream(prisim_ctx* ctx) {
  auto& C = ctx->cat;
  if (!ctx->prep_async) {
    HIPCHK(ctx, hipEventRecord(C.ev_join, ctx->stream));
    HIPCHK(ctx, hipStreamWaitEvent(ctx->prep_stream, C.ev_join, 0));
  }
  ctx->sk = &ctx->skb[ctx->sk_next];
  ctx->sk_next ^= 1;
  ctx->prep_async = true;
  if (ctx->sk->sum_recorded) HIPCHK(ctx, hipStreamWaitEvent(ctx->prep_stream, ctx->sk->ev_sum, 0));
  return PRISIM_OK;
}

// What a batched launch (small arrays, run_wave_batch) tells the geometry so that the scan pass can write the launch's per-snapshot table
struct BatchLayout {
  BatchSnap* tab = nullptr;
  int64_t npad = 0;
  int32_t nsplit = 1;
  double* out = nullptr;
  int64_t slot_elems = 0;
  double* gout = nullptr;      // gradient batches (BatchSnap::gout)
};

// Geometry of nsnap snapshots into buffer set b, queued on the geometry stream; ev_geom is recorded behind the copy of the per-snapshot
// records into C.out_host.  geometry_wait() makes them readable; device work that needs the geometry waits for ev_geom in its own stream.
int geometry_enqueue(prisim_ctx* ctx, const prisim_obs* obs, const prisim_snapshot* snaps, int64_t nsnap, int b, bool want_keys,
                     const BatchLayout* batch = nullptr) {
  auto& C = ctx->cat;
  int rc;
  { HostSpan sp("cat_runtime"); if ((rc = cat_runtime(ctx, nsnap))) return rc; }
  auto& S = C.set[b];
  const size_t rows = (size_t)nsnap * (size_t)std::max<int64_t>(C.n, 1);
  const int64_t nblocks = cat_blocks(C.n);
  if ((rc = ensure(ctx, S.idx, rows * sizeof(int32_t))) || (rc = ensure(ctx, S.dirs, rows * 4 * sizeof(double))) ||
      (want_keys && ((rc = ensure(ctx, S.keys, rows * sizeof(uint32_t))) || (rc = ensure(ctx, S.pos, rows * sizeof(uint32_t))))) ||
      (rc = ensure(ctx, C.block_off, (size_t)nsnap * (size_t)std::max<int64_t>(nblocks, 1) * sizeof(int32_t))))
    return rc;
  for (int64_t t = 0; t < nsnap; ++t) {
    CatSnap& s = C.snaps_host[t];
    if (snaps[t].frame_given) {
      memcpy(s.rot, snaps[t].cel2enu, sizeof(s.rot));
      memcpy(s.beta, snaps[t].aberr_beta, sizeof(s.beta));
    } else {
      fallback_frame(C.coords, snaps[t].lst_deg, obs->latitude_deg, s.rot, s.beta);
    }
    for (int i = 0; i < 3; ++i) { s.roi_pc[i] = snaps[t].pc_dircos[i]; s.pc[i] = snaps[t].pc_dircos[i]; s.bpc[i] = snaps[t].beam_pc_dircos[i]; }
  }
  C.geom_pending = false;
  C.geom_nsnap = nsnap; C.geom_set = b;
  if (C.n == 0) {
    for (int64_t t = 0; t < nsnap; ++t) {
      C.out_host[t].nsrc = 0; C.out_host[t].dmax2_bits = 0;
      for (auto& v : C.out_host[t].run_start) v = 0;
    }
    return PRISIM_OK;
  }
  CatGeomParams p{};
  p.ux = (const double*)C.ux.p; p.uy = (const double*)C.uy.p; p.uz = (const double*)C.uz.p;
  p.kappa = C.have_shape ? (const double*)C.kappa.p : nullptr;
  p.run_id = C.runs.empty() ? nullptr : (const uint8_t*)C.run_id.p;
  p.n = C.n; p.nblocks = nblocks;
  p.roi_center = obs->roi_center; p.want_keys = want_keys ? 1 : 0;
  p.sin_alt_min = std::sin((90.0 - obs->roi_radius_deg) * (M_PI / 180.0));      // interferometry.py:6216 on the direction cosine n
  p.cos_radius = std::cos(obs->roi_radius_deg * (M_PI / 180.0));               // :6211 on s . s_pc  (geometry.roi_thresholds)
  p.snaps = (const CatSnap*)C.snaps.p;
  p.block_off = (int32_t*)C.block_off.p;
  p.idx = (int32_t*)S.idx.p; p.dirs = (double*)S.dirs.p;
  p.keys = want_keys ? (uint32_t*)S.keys.p : nullptr; p.pos = want_keys ? (uint32_t*)S.pos.p : nullptr;
  p.out = (CatOut*)C.out_dev.p;
  if (batch) {
    p.batch = batch->tab; p.batch_npad = batch->npad; p.batch_nsplit = batch->nsplit; p.batch_out = batch->out; p.batch_slot_elems = batch->slot_elems;
    p.batch_gout = batch->gout;
  }
  // the set may still be read by sky-sums queued earlier on the compute stream
  if (S.ev_recorded) HIPCHK(ctx, hipStreamWaitEvent(C.gstream, S.ev_free, 0));
  // ... or by the preparation of a sky nobody summed (two set_sky_from_catalog calls in a row, an empty region of interest)
  if (S.prep_recorded) HIPCHK(ctx, hipStreamWaitEvent(C.gstream, S.ev_prepared, 0));
  const auto t0 = Clock::now();
  // the one-block form is for the latency of small problems (HERA-19 x nside-16: a snapshot's geometry in one launch, no copies); a large
  // array's geometry runs beside a sky-sum grid that fills the chip and must consist of many small blocks
  const bool small = C.n <= kCatSmallMax && ctx->nbl <= kBlockThreads;
  p.small_form = small ? 1 : 0;
  if (small && nsnap == 1) {
    p.inline_snap = 1;                       // the one snapshot's inputs travel in the kernel arguments
    p.snap0 = C.snaps_host[0];
  } else {
    HIPCHK(ctx, hipMemcpyAsync(C.snaps.p, C.snaps_host, (size_t)nsnap * sizeof(CatSnap), hipMemcpyHostToDevice, C.gstream));
  }
  // small catalogues: one block per snapshot writes its record straight into the page-locked host array (device-visible); otherwise
  // the three passes write device records and a copy follows
  if (small) p.out = C.out_host;
  HIPCHK(ctx, launch_cat_geometry(p, (int)nsnap, C.gstream));
  if (!small) HIPCHK(ctx, hipMemcpyAsync(C.out_host, C.out_dev.p, (size_t)nsnap * sizeof(CatOut), hipMemcpyDeviceToHost, C.gstream));
  HIPCHK(ctx, hipEventRecord(C.ev_geom, C.gstream));
  C.geom_pending = true;
  C.geom_t0 = t0;
  return PRISIM_OK;
}

// The per-snapshot records of the last geometry_enqueue are in C.out_host when this returns.
int geometry_wait(prisim_ctx* ctx) {
  auto& C = ctx->cat;
  if (C.geom_pending) {
    HIPCHK(ctx, hipEventSynchronize(C.ev_geom));
    C.geom_pending = false;
    C.geom_ms_sum += std::chrono::duration<double, std::milli>(Clock::now() - C.geom_t0).count();
    C.geom_calls += 1;
  }
  if (C.geom_set != 2) {
    C.chunk_nmax = 0;
    for (int64_t t = 0; t < C.geom_nsnap; ++t) C.chunk_nmax = std::max(C.chunk_nmax, C.out_host[t].nsrc);
  }
  return PRISIM_OK;
}

int geometry_run(prisim_ctx* ctx, const prisim_obs* obs, const prisim_snapshot* snaps, int64_t nsnap, int b, bool want_keys) {
  int rc = geometry_enqueue(ctx, obs, snaps, nsnap, b, want_keys);
  return rc ? rc : geometry_wait(ctx);
}

// Make snapshot t of buffer set b the current sky: runs, altitude order, cull table, pb * flux.  Everything is queued on the compute
// stream; nothing is waited for.
int activate_snapshot(prisim_ctx* ctx, const prisim_obs* obs, const prisim_snapshot& snap, int b, int64_t t, bool want_keys) {
  auto& C = ctx->cat;
  auto& S = C.set[b];
  const CatOut& o = C.out_host[t];
  const int64_t N = o.nsrc;
  if (N < 0 || N > C.n) return fail(ctx, PRISIM_EINTERNAL, "catalogue geometry returned an impossible source count");
  int rc;
  ctx->sky_set = false;
  // Preparation stream and buffer set.  The sky of snapshot t+1 (altitude sort, cull table, beam x flux, and compute()'s packing) is
  // prepared on the geometry stream into the set the sky-sum of snapshot t is NOT reading, so that it runs under that sky-sum; a
  // beamformer's per-snapshot host arrays go through the compute stream's staging area, so such skies stay on the one stream.
  bool async = beamformer_doubles(obs->use_external_beam ? nullptr : obs->ext) == 0;
  if (const char* env = getenv("PRISIM_HIP_PREP_ASYNC")) async = async && atoi(env) != 0;      // A/B hook
  if (async) {
    if ((rc = enter_prep_stream(ctx))) return rc;
  } else {
    if (ctx->prep_async) HIPCHK(ctx, hipStreamSynchronize(ctx->prep_stream));
    ctx->prep_async = false;
    ctx->sk = &ctx->skb[0];
  }
  const hipStream_t ps = pstream(ctx);
  ctx->nsrc = N;
  ctx->taper = C.have_shape;
  for (int i = 0; i < 3; ++i) ctx->pc[i] = snap.pc_dircos[i];
  double d2 = 0.0;
  memcpy(&d2, &o.dmax2_bits, sizeof(double));
  ctx->dmax = std::sqrt(d2);
  ctx->kappa_runs.clear();
  const int ncr = (int)C.runs.size();
  for (int r = 0; r < ncr; ++r) {
    const int64_t lo = o.run_start[r], hi = r + 1 < ncr ? o.run_start[r + 1] : N;
    if (lo < 0 || hi < lo || hi > N) return fail(ctx, PRISIM_EINTERNAL, "catalogue geometry returned inconsistent run boundaries");
    if (hi > lo) ctx->kappa_runs.push_back({lo, hi, C.runs[(size_t)r].kappa, r});
  }
  const size_t off = (size_t)t * (size_t)C.n;
  const double* dirs = (const double*)S.dirs.p + off * 4;
  const int32_t* idx = (const int32_t*)S.idx.p + off;
  if (want_keys && N > 1) {
    // every run of one source size by decreasing altitude (stable radix sort on (run, 1 - n) keys): the leading sources of a run are then
    // the ones a long-baseline group can cull
    const size_t nn = (size_t)C.n;
    if (!C.sort_tmp.p &&
        ((rc = ensure(ctx, C.sort_tmp, std::max<size_t>(cat_sort_temp_bytes(C.n), 16))) || (rc = ensure(ctx, C.keys_out, nn * sizeof(uint32_t))) ||
         (rc = ensure(ctx, C.perm, nn * sizeof(uint32_t)))))
      return rc;
    if ((rc = ensure(ctx, ctx->sk->idx_sorted, nn * sizeof(int32_t))) || (rc = ensure(ctx, ctx->sk->dirs_sorted, nn * 4 * sizeof(double)))) return rc;
    HIPCHK(ctx, launch_cat_sort(C.sort_tmp.p, C.sort_tmp.bytes, (const uint32_t*)S.keys.p + off, (uint32_t*)C.keys_out.p, (const uint32_t*)S.pos.p + off,
                                (uint32_t*)C.perm.p, dirs, idx, (double*)ctx->sk->dirs_sorted.p, (int32_t*)ctx->sk->idx_sorted.p, N, ps));
    ctx->dirs_p = (const double*)ctx->sk->dirs_sorted.p;
    ctx->src_index = (const int32_t*)ctx->sk->idx_sorted.p;
  } else {
    ctx->dirs_p = dirs;
    ctx->src_index = idx;
  }
  // taper culling: the table of upload_common's host walk, built by k_cull_first (one row per CATALOGUE run)
  ctx->cull_any[0] = ctx->cull_any[1] = false;
  ctx->cull_frac[0] = ctx->cull_frac[1] = 0.0;
  ctx->cull_nruns = 0;
  const size_t ng = ctx->grp_maxlen.size();
  const char* cull_env = getenv("PRISIM_HIP_TAPER_CULL");
  if (!ctx->kappa_runs.empty() && ng > 0 && N > 0 && !(cull_env && atoi(cull_env) == 0)) {
    const double fmin = std::min(std::fabs(ctx->h_freqs.front()), std::fabs(ctx->h_freqs.back()));
    const double fc2 = (fmin / kC) * (fmin / kC);
    double hmax = 0.0;
    for (size_t g = 0; g < ng; ++g) hmax = std::max(hmax, ctx->grp_minh[g]);
    bool possible[2] = {false, false};
    for (const auto& run : ctx->kappa_runs)
      for (int pr = 0; pr < 2; ++pr) possible[pr] = possible[pr] || (run.kappa > 0.0 && run.kappa * hmax * hmax * fc2 >= kCullThr[pr]);
    if (possible[0] || possible[1]) {
      CullParams cp{};
      cp.dirs = ctx->dirs_p;
      for (int r = 0; r < ncr; ++r) {
        cp.run_lo[r] = o.run_start[r];
        cp.run_hi[r] = r + 1 < ncr ? o.run_start[r + 1] : N;
        cp.run_kappa[r] = C.runs[(size_t)r].kappa;
      }
      cp.nruns = ncr; cp.ng = (int32_t)ng;
      cp.grp_maxz = (const double*)ctx->grp_hz.p + ng;
      cp.grp_minh = (const double*)ctx->grp_hz.p + 3 * ng;
      cp.fc2 = fc2;
      cp.nbl = ctx->nbl;
      if ((rc = ensure(ctx, ctx->sk->cull_first, 2 * (size_t)ncr * ng * sizeof(int32_t))) || (rc = ensure(ctx, C.culled, 2 * sizeof(uint64_t)))) return rc;
      cp.first = (int32_t*)ctx->sk->cull_first.p;
      cp.culled = (uint64_t*)C.culled.p;
      HIPCHK(ctx, hipMemsetAsync(C.culled.p, 0, 2 * sizeof(uint64_t), ps));
      HIPCHK(ctx, launch_cull_first(cp, ps));
      HIPCHK(ctx, hipMemcpyAsync(C.culled_host, C.culled.p, 2 * sizeof(uint64_t), hipMemcpyDeviceToHost, ps));
      ctx->cull_any[0] = possible[0]; ctx->cull_any[1] = possible[1];
      ctx->cull_nruns = ncr;
    }
  }
  // pb * fluxes (:6249-6254); the flux vectors / spectra stay in catalogue order and are read through the index list.  Sized for the
  // largest region of interest of the chunk (known from the geometry) plus a quarter, never beyond the whole catalogue: a horizon ROI sees
  // at most half of an all-sky catalogue, and sizing both buffer sets for all of it was 4x what the uploaded path ever allocated (a large
  // catalogue that ran before must not run out of memory here); the headroom keeps a drift scan's growing ROI from re-allocating -- a
  // device-wide synchronisation -- at every snapshot.
  if ((rc = ensure_grow(ctx, ctx->sk->pb, (size_t)std::max<int64_t>(std::max(C.chunk_nmax, N) * ctx->nchan, 1) * sizeof(double),
                        (size_t)std::max<int64_t>(C.n * ctx->nchan, 1) * sizeof(double))))
    return rc;
  if (N > 0) {
    const double* fr = C.have_spec ? nullptr : (const double*)C.flux_ref.p;
    const double* sp = C.have_spec ? nullptr : (const double*)C.spindex.p;
    const double* fs = C.have_spec ? (const double*)C.spec.p : nullptr;
    if (obs->use_external_beam) {
      if ((rc = extbeam_sky(ctx, N, fs, fr, sp, C.have_spec ? 1.0 : C.ref_freq, ctx->src_index))) return rc;
    } else {
      const size_t bfb = beamformer_doubles(obs->ext) * sizeof(double);
      if (bfb && (rc = stage_begin(ctx, bfb + 4096))) return rc;
      rc = sky_beam_flux(ctx, N, obs->beam_kind, obs->diameter_m, snap.beam_pc_dircos, obs->ext, fr, sp, fs, C.ref_freq, ctx->src_index);
      if (bfb) stage_end(ctx);
      if (rc) return rc;
      if (obs->beam_kind == PRISIM_BEAM_POLY && (rc = check_poly_beam_flag(ctx))) return rc;
    }
  }
  HIPCHK(ctx, hipEventRecord(S.ev_prepared, ps));
  S.prep_recorded = true;
  C.cur = b;
  ctx->sky_set = true;
  return PRISIM_OK;
}


// ---- many snapshots of a small array in ONE launch -------------------------------------------------------------------------------
// Arrays of at most 256 baselines (HERA-19: 171 = 3 baseline waves) cannot fill the chip with one snapshot: config 2's sky-sum is a
// 50 us launch at 0.17 of the fp64 roofline, three more launches per snapshot, and ~0.2 ms of host time around them.  With the
// catalogue resident the snapshots of a chunk are independent work items: their beam x flux, their packing and their sky-sums each go
// into ONE launch over (snapshot, ...) and one reduction -- the work item of the sky-sum is (snapshot, baseline wave, channel tile,
// source split).  Eligible: fp64 with the source-shape taper (the grouped kernel), a uniform channel grid,
// nothing the taper culling could skip, an analytic beam without a beamformer or the external HEALPix beam, no gradient.  Everything else
// takes the per-snapshot loop.
bool wave_batch_eligible(const prisim_ctx* ctx, const prisim_obs* obs, int precision, int want_grad, int64_t kc) {
  const auto& C = ctx->cat;
  if (const char* env = getenv("PRISIM_HIP_WAVE_BATCH")) { if (atoi(env) == 0) return false; }
  if (const char* env = getenv("PRISIM_HIP_WAVE_ITEMS")) { if (atoi(env) == 0) return false; }
  // precision: a request for fp32 arithmetic (PRISim's memsave) on an array this small is served by the SAME fp64 launch -- the cost of
  // a small array's snapshot is its launches, not its arithmetic (HERA-19: 26 us per snapshot batched in fp64 against a 45-50 us launch
  // chain per snapshot in either precision), and the result is the fp64 one (well inside the fp32 tolerance).  PRISIM_HIP_BATCH_FP32_AS_FP64=0:
  // fp32 requests keep the per-snapshot fp32 chain.
  if (precision != PRISIM_FP64) { const char* env = getenv("PRISIM_HIP_BATCH_FP32_AS_FP64"); if (env && atoi(env) == 0) return false; }
  // want_grad: the grouped fp64 gradient kernel on wave items of 16 baselines (k_skyvis_grad_taper_f64_batch), 32-channel tiles
  if (kc < 1 || !ctx->uniform || ctx->nbl > kBlockThreads || ctx->nchan < 16) return false;
  if (want_grad) {
    if (!grad_taper_grouped()) return false;
    if (const char* env = getenv("PRISIM_HIP_WAVE_BATCH_GRAD")) { if (atoi(env) == 0) return false; }
    if (const char* env = getenv("PRISIM_HIP_FUSED_GRAD")) { if (atoi(env) == 0) return false; }
  }
  // one snapshot per call (observe() on a small array) takes the batched launch too -- 16-channel tiles, one round of wave slots, prepared
  // on the preparation stream under the previous snapshot's sky-sum, queued without a count (run_wave_batch): 100 us per observe() of
  // HERA-19 against 115 through the per-snapshot chain (tools/observe_single_ab.py).  PRISIM_HIP_WAVE_BATCH_SINGLE=0: that chain (A/B).
  if (kc < 2) { const char* env = getenv("PRISIM_HIP_WAVE_BATCH_SINGLE"); if (env && atoi(env) == 0) return false; }
  // (sizes may vary from source to source and the sky may consist of several runs -- point sources + a diffuse map: every source
  // carries its own kappa, and with nothing to cull the runs need no separate launches; point sources then pay the taper kernel's 9.7
  // instead of the plain kernel's 6.2 instructions per term, which a launch per run and snapshot would cost many times over)
  // (a catalogue without source shapes is the same with kappa = 0 everywhere: the weight is exactly 1, the sums those of the plain kernel
  // to rounding)
  if (!taper_f64_grouped_enabled()) return false;
  if (!obs->use_external_beam && (beamformer_doubles(obs->ext) != 0 || obs->beam_kind == PRISIM_BEAM_POLY)) return false;
  if (ctx->tune_chunk) return false;
  if (cat_sort_wanted(ctx)) return false;
  // nothing to cull for any precision (the cull table is per snapshot)
  {
    const double fmin = std::min(std::fabs(ctx->h_freqs.front()), std::fabs(ctx->h_freqs.back()));
    double hmax = 0.0;
    for (double v : ctx->grp_minh) hmax = std::max(hmax, v);
    if (C.kappa_max * hmax * hmax * (fmin / kC) * (fmin / kC) >= kCullThr[1]) return false;
  }
  return true;
}

// The geometry of the chunk is queued from in here: the scan pass writes the launch's per-snapshot table on the device (BatchSnap: counts,
// split sizes, phase / beam centres, output pointers), every snapshot owning a fixed-size block of the direction, beam x flux and packed-row
// buffers -- so beam x flux, packing, the sky-sums and the reduction are queued without the host having seen a single count, and the
// compute stream waits for the geometry by event.  The host reads the counts (nsrc_roi, timing) at the END, when they have long arrived:
// a chunk costs no round trip to the device any more (a single observe() of HERA-19 spent 45 of its 118 us waiting for one).
int run_wave_batch(prisim_ctx* ctx, const prisim_obs* obs, const prisim_snapshot* snaps, int b, int64_t kc, int64_t slot0, int64_t* nsrc_roi,
                   bool want_keys, bool want_grad) {
  auto& C = ctx->cat;
  auto& S = C.set[b];
  int rc;
  if ((rc = cat_runtime(ctx, kc))) return rc;
  // buffer set and preparation stream, as activate_snapshot -- but by default the chunk is prepared on the COMPUTE stream: its preparation
  // is two large compute-bound launches (beam x flux, packing), and under the previous chunk's sky-sum they only take its CUs away
  // (config 2 x 64 snapshots, 8 chunks queued: 1.975 ms per chunk on the preparation stream, the sky-sum slowed from 1.46 to 1.84 ms;
  // 1.885 ms in line).  The host is ahead either way: the geometry of the next chunk runs on its own stream.
  // ... A SINGLE snapshot per call (observe() on a small array) is the opposite case: its preparation is a few latency-bound launches, and
  // on the preparation stream (two sky-buffer sets) it runs under the previous snapshot's sky-sum.
  bool async = kc == 1;
  if (const char* env = getenv("PRISIM_HIP_PREP_ASYNC_BATCH")) async = atoi(env) != 0;
  if (async) {
    if ((rc = enter_prep_stream(ctx))) return rc;
  } else {
    if (ctx->prep_async) HIPCHK(ctx, hipStreamSynchronize(ctx->prep_stream));
    ctx->prep_async = false;
    ctx->sk = &ctx->skb[0];
  }
  const hipStream_t ps = pstream(ctx);
  SkyBufs& K = *ctx->sk;
  // plan: 32-channel tiles (the seed of a (source, baseline, tile) triple is 26 % of a 32-channel tile's instructions, 41 % of a
  // 16-channel tile's), source splits so that the grid is about three rounds of two wavefronts per SIMD.  Everything here is a function
  // of the array, the catalogue and the chunk length -- never of the counts, which the host has not seen.
  // (one snapshot alone: 16-channel tiles and one round of wave slots -- the cut the single-launch planner arrives at for HERA-19)
  // (V + gradient: the MFMA kernel's wave follows 16 baselines x 4 sources on 32-channel tiles)
  int ct = ctx->tune_ct ? ctx->tune_ct : ((ctx->nchan >= 32 && kc > 1) ? 32 : 16);
  if (ct != 16 && ct != 32) ct = 32;
  if (want_grad) ct = 32;
  const int ntiles = (int)((ctx->nchan + ct - 1) / ct);
  const int nbw = want_grad ? (int)((ctx->nbl + 15) / 16) : (int)((ctx->nbl + 63) / 64);
  const int64_t npad = round_up(std::max<int64_t>(C.n, 1), 4);
  int64_t nsplit = ctx->tune_nsplit;
  if (nsplit == 0) {
    const int64_t slots = 2LL * 4 * std::max(ctx->cu_count, 1);
    const int64_t items0 = kc * nbw * ntiles;
    const int64_t lo = std::max<int64_t>(1, ((kc > 1 ? 5 * slots / 2 : slots) + items0 - 1) / items0);
    // ... and of those counts the one whose last round is fullest (config 2 x 64 snapshots: 3 splits = 2.25 rounds 1.66 ms, 4 = 3.0 rounds
    // 1.52 ms, 5 = 3.75 rounds 1.58 ms; tools/config2_batch_sweep.py)
    double best = 1e30;
    nsplit = lo;
    for (int64_t n = lo; n <= 2 * lo; ++n) {
      const double r = (double)(items0 * n) / (double)slots;
      const double waste = std::ceil(r) / r + 0.002 * (double)(n - lo);
      if (waste < best - 1e-12) { best = waste; nsplit = n; }
    }
    if (kc == 1) nsplit = std::max<int64_t>(1, slots / items0);              // one snapshot alone: ONE round of wave slots, never a second
    nsplit = std::min<int64_t>(nsplit, std::max<int64_t>(1, C.n / 32));      // (a horizon region of interest holds about half the catalogue)
    nsplit = std::min<int64_t>(nsplit, 64);
  }
  const size_t slot_elems = (size_t)ctx->nbl * ctx->nchan * 2;
  const int64_t pitch = kc * npad;                     // packed rows / prepared directions: snapshot t owns rows [t npad, (t + 1) npad)
  const size_t pb_rows = (size_t)kc * (size_t)std::max<int64_t>(C.n, 1);
  const int ti = C.tab_next;
  C.tab_next = (ti + 1) & 3;
  DevBuf& tabbuf = C.batch_tabs[ti];
  // partial cubes: [kc][nsplit][slot] of V, then (gradient) [kc][nsplit][3 slot]
  const size_t part_v = (size_t)kc * (size_t)nsplit * slot_elems;
  if (want_grad && !ctx->grad.p) {
    const size_t gbytes = (size_t)ctx->nt_max * 3 * slot_elems * sizeof(double);
    if ((rc = ensure(ctx, ctx->grad, gbytes))) return rc;
    HIPCHK(ctx, hipMemsetAsync(ctx->grad.p, 0, gbytes, ctx->stream));
  }
  if ((rc = ensure(ctx, tabbuf, (size_t)std::max<int64_t>(kc, 64) * sizeof(BatchSnap))) ||
      (nsplit > 1 && (rc = ensure(ctx, ctx->partial, part_v * (want_grad ? 4 : 1) * sizeof(double)))) ||
      (rc = ensure(ctx, K.pb, pb_rows * ctx->nchan * sizeof(double))) ||
      (rc = ensure(ctx, K.packed, (size_t)ntiles * (size_t)pitch * ct * sizeof(double))) ||
      (rc = ensure(ctx, K.dirs_prep, (size_t)pitch * 4 * sizeof(double))) || (rc = ensure(ctx, ctx->sky_flag, sizeof(int32_t))))
    return rc;
  if (obs->use_external_beam &&
      ((rc = ensure(ctx, ctx->ext_work, pb_rows * ctx->nchan * sizeof(double))) ||
       (rc = ensure(ctx, ctx->ext_colmax, std::max<size_t>((size_t)kc * (kExtBatchBlocks + 1), 1025) * ctx->nchan * sizeof(double)))))
    return rc;
  BatchLayout lay;
  lay.tab = (BatchSnap*)tabbuf.p;
  lay.npad = npad;
  lay.nsplit = (int32_t)nsplit;
  lay.slot_elems = (int64_t)slot_elems;
  lay.out = nsplit > 1 ? (double*)ctx->partial.p : (double*)ctx->cube.p + (size_t)slot0 * slot_elems;
  if (want_grad) lay.gout = nsplit > 1 ? (double*)ctx->partial.p + part_v : (double*)ctx->grad.p + (size_t)slot0 * 3 * slot_elems;
  // the table is one of a ring of four: the launches of the chunk that used this one four chunks ago read it on the compute / preparation
  // stream, the geometry stream writes it -- order the write behind them (the set's geometry buffers are ordered by ev_free / ev_prepared)
  if (C.tabfree_rec[ti]) HIPCHK(ctx, hipStreamWaitEvent(C.gstream, C.ev_tabfree[ti], 0));
  { HostSpan sp("geometry_enqueue"); if ((rc = geometry_enqueue(ctx, obs, snaps, kc, b, want_keys, &lay))) return rc; }
  HIPCHK(ctx, hipStreamWaitEvent(ps, C.ev_geom, 0));
  const int64_t nmax = C.n;
  // beam x flux of all snapshots (:6249-6254), then rows + prepared directions of all snapshots
  if (obs->use_external_beam) {
    // external HEALPix beam (run_prisim.py:2091-2103): gather, per-snapshot column maximum, 10 ** (.) x flux -- four launches for the chunk
    HIPCHK(ctx, launch_extbeam_sky_batch((const double*)ctx->ext_table.p, ctx->ext_nside, (const double*)S.dirs.p,
                                         C.have_spec ? (const double*)C.spec.p : nullptr, C.have_spec ? nullptr : (const double*)C.flux_ref.p,
                                         C.have_spec ? nullptr : (const double*)C.spindex.p, (const double*)ctx->freqs.p, C.have_spec ? 1.0 : C.ref_freq,
                                         (double*)ctx->ext_work.p, (double*)ctx->ext_colmax.p, (double*)K.pb.p, nmax, ctx->nchan,
                                         (const int32_t*)S.idx.p, (const BatchSnap*)tabbuf.p, (int)kc, ps));
  } else {
    BeamParams bp{};
    bp.dirs = (const double*)S.dirs.p;
    bp.src_index = (const int32_t*)S.idx.p;
    bp.flux_ref = C.have_spec ? nullptr : (const double*)C.flux_ref.p;
    bp.spindex = C.have_spec ? nullptr : (const double*)C.spindex.p;
    bp.flux_spec = C.have_spec ? (const double*)C.spec.p : nullptr;
    bp.freqs = (const double*)ctx->freqs.p;
    bp.ref_freq = C.have_spec ? 1.0 : C.ref_freq;
    bp.beam_kind = obs->beam_kind;
    bp.diameter = obs->diameter_m;
    if (obs->ext) {
      const prisim_beam_ext* x = obs->ext;
      bp.dip_x = x->dipole_dircos[0]; bp.dip_y = x->dipole_dircos[1]; bp.dip_z = x->dipole_dircos[2];
      bp.dipole_mode = x->dipole_mode;
      bp.nax1 = x->array_nax1; bp.nax2 = x->array_nax2; bp.sep1 = x->array_sep1; bp.sep2 = x->array_sep2;
      const double ang = x->array_east2ax1_deg * M_PI / 180.0;
      bp.rot_c = std::cos(ang); bp.rot_s = std::sin(ang);
      bp.apc_x = x->array_pc_dircos[0]; bp.apc_y = x->array_pc_dircos[1]; bp.apc_z = x->array_pc_dircos[2];
      bp.gp_height = x->ground_height; bp.gp_modify = x->ground_modify; bp.gp_scale = x->ground_scale; bp.gp_max = x->ground_max;
    }
    bp.flag = (int32_t*)ctx->sky_flag.p;
    bp.nsrc = nmax; bp.nchan = ctx->nchan;
    bp.pb_out = (double*)K.pb.p;
    bp.batch = (const BatchSnap*)tabbuf.p;
    HIPCHK(ctx, launch_beam_flux_batch(bp, (int)kc, ps));
  }
  HIPCHK(ctx, launch_pack_prep_batch((const double*)K.pb.p, (double*)K.packed.p, pitch, npad, ctx->nchan, ct, ntiles, (const double*)S.dirs.p,
                                     (double*)K.dirs_prep.p, 1.0 / kC, (const BatchSnap*)tabbuf.p, (int)kc, ps));
  if ((rc = join_prep(ctx))) return rc;
  // the sky-sums of the whole chunk: ONE launch, ONE reduction
  harvest_timing(ctx, ctx->ring_pending >= prisim_ctx::kTimingRing ? 1 : 0);
  SkyvisParams p{};
  p.bl_x = (const double*)ctx->blx.p; p.bl_y = (const double*)ctx->bly.p; p.bl_z = (const double*)ctx->blz.p;
  p.nbl = ctx->nbl; p.nchan = ctx->nchan;
  p.f0 = ctx->f0; p.df = ctx->df; p.inv_c = 1.0 / kC;
  p.dirs = (const double*)S.dirs.p;
  p.dirs_prep = (const double*)K.dirs_prep.p;
  p.pb_packed = K.packed.p;
  p.fsq = (const float*)ctx->fsq.p;
  p.fsq_scale = 1e16;
  p.nsrc = pitch; p.nsrc_pad = pitch;                       // (the slab pitch: every snapshot's rows are a range of it)
  p.taper = 1;
  p.ntiles = ntiles; p.nsplit = 1; p.src_per_split = 0; p.src_chunk = 4;
  p.flush_src = 16384; p.scale_comp = -1;
  p.wave_nbw = nbw; p.wave_nsplit = (int32_t)nsplit;
  p.wave_snaps = (const BatchSnap*)tabbuf.p; p.wave_nsnap = (int32_t)kc;
  p.nbgroups = (int32_t)((kc * nbw * nsplit + kBlockThreads / 64 - 1) / (kBlockThreads / 64));
  p.out = (double*)ctx->cube.p;
  const int ri = ctx->ring_head;
  HIPCHK(ctx, hipEventRecord(ctx->ev_c0[ri], ctx->stream));
  HIPCHK(ctx, hipEventRecord(ctx->ev_k0[ri], ctx->stream));
  if (want_grad) HIPCHK(ctx, launch_skyvis_grad_taper_f64_batch(p, ctx->stream));
  else HIPCHK(ctx, launch_skyvis_taper_f64_wave_batch(p, ct, ctx->stream));
  HIPCHK(ctx, hipEventRecord(ctx->ev_k1[ri], ctx->stream));
  if (nsplit > 1) {
    HIPCHK(ctx, launch_reduce_partials_batch((const double*)ctx->partial.p, (double*)ctx->cube.p + (size_t)slot0 * slot_elems, (int64_t)slot_elems, (int)nsplit,
                                             (int)kc, ctx->stream));
    if (want_grad)
      HIPCHK(ctx, launch_reduce_partials_batch((const double*)ctx->partial.p + part_v, (double*)ctx->grad.p + (size_t)slot0 * 3 * slot_elems,
                                               3 * (int64_t)slot_elems, (int)nsplit, (int)kc, ctx->stream));
  }
  HIPCHK(ctx, hipEventRecord(ctx->ev_c1[ri], ctx->stream));
  HIPCHK(ctx, hipEventRecord(K.ev_sum, ctx->stream));      // (in line too: a later preparation-stream user of this set waits for these sums)
  K.sum_recorded = true;
  HIPCHK(ctx, hipEventRecord(C.ev_tabfree[ti], ctx->stream));
  C.tabfree_rec[ti] = true;
  C.cur = b;
  catalog_after_compute(ctx);
  ctx->ring_head = (ctx->ring_head + 1) % prisim_ctx::kTimingRing;
  ctx->ring_pending += 1;
  // only now the counts: they arrived while the launches above were being queued
  { HostSpan sp("geometry_wait"); if ((rc = geometry_wait(ctx))) return rc; }
  int64_t ntot = 0;
  for (int64_t t = 0; t < kc; ++t) {
    const int64_t N = C.out_host[t].nsrc;
    if (N < 0 || N > C.n) return fail(ctx, PRISIM_EINTERNAL, "catalogue geometry returned an impossible source count");
    if (nsrc_roi) nsrc_roi[t] = N;
    ntot += N;
  }
  ctx->timing.last_terms = ctx->nbl * ctx->nchan * ntot;
  ctx->timing.last_kernel_id = PRISIM_KERNEL_RECURRENCE;
  ctx->timing.last_chan_tile = ct;
  ctx->timing.last_nsplit = (int32_t)nsplit;
  ctx->timing.last_lift_groups = 0;
  ctx->timing.last_taper_group = 0;
  ctx->timing.last_taper_split = 0;
  ctx->timing.last_split_uncorrected_groups = 0;
  ctx->timing.last_culled_fraction = 0.0;
  ctx->timing.last_batch_snapshots = (int32_t)kc;
  // no single snapshot is "the current sky" afterwards
  ctx->sky_set = false;
  ctx->nsrc = C.out_host[kc - 1].nsrc;
  return PRISIM_OK;
}

// the download / gather of a finished slot, behind its sky-sum
int post_snapshot(prisim_ctx* ctx, const prisim_post* post, int64_t slot) {
  if (!post) return PRISIM_OK;
  int rc;
  if (post->host_vis) {
    const size_t bytes = (size_t)ctx->nbl * (size_t)ctx->nchan * (post->host_is_c64 ? 8 : 16);
    if ((rc = prisim_hip_get_vis_async(ctx, slot, (char*)post->host_vis + (size_t)slot * bytes, nullptr, post->host_is_c64))) return rc;
  }
  if (post->gather && (rc = prisim_hip_allgather_slot_async(ctx, slot, post->gather_as_c64))) return rc;
  return PRISIM_OK;
}

}  // namespace

namespace pint {

void catalog_after_compute(prisim_ctx* ctx) {
  auto& C = ctx->cat;
  if (C.cur < 0 || C.cur > 1) return;
  if (hipEventRecord(C.set[C.cur].ev_free, ctx->stream) == hipSuccess) C.set[C.cur].ev_recorded = true;
}

void catalog_destroy(prisim_ctx* ctx) {
  auto& C = ctx->cat;
  if (C.gstream) (void)hipStreamSynchronize(C.gstream);
  for (DevBuf* b : {&C.lon, &C.lat, &C.ux, &C.uy, &C.uz, &C.kappa, &C.run_id, &C.flux_ref, &C.spindex, &C.spec, &C.block_off, &C.snaps, &C.out_dev,
                    &C.sort_tmp, &C.keys_out, &C.perm, &C.culled})
    release(*b);
  for (auto& s : C.set) {
    for (DevBuf* b : {&s.idx, &s.dirs, &s.keys, &s.pos}) release(*b);
    if (s.ev_free) (void)hipEventDestroy(s.ev_free);
    if (s.ev_prepared) (void)hipEventDestroy(s.ev_prepared);
    s.ev_free = s.ev_prepared = nullptr;
  }
  if (C.ev_geom) (void)hipEventDestroy(C.ev_geom);
  if (C.ev_join) (void)hipEventDestroy(C.ev_join);
  for (auto& e : C.ev_tab) { if (e) (void)hipEventDestroy(e); e = nullptr; }
  for (auto& e : C.ev_tabfree) { if (e) (void)hipEventDestroy(e); e = nullptr; }
  for (auto& b : C.batch_tabs) release(b);
  C.tab_recorded[0] = C.tab_recorded[1] = false;
  if (C.out_host) (void)hipHostFree(C.out_host);
  if (C.snaps_host) (void)hipHostFree(C.snaps_host);
  if (C.culled_host) (void)hipHostFree(C.culled_host);
  if (C.batch_host) (void)hipHostFree(C.batch_host);
  if (C.gstream) (void)hipStreamDestroy(C.gstream);
  if (ctx->prep_stream) { (void)hipStreamSynchronize(ctx->prep_stream); (void)hipStreamDestroy(ctx->prep_stream); }
  ctx->prep_stream = nullptr;
  ctx->prep_async = false;
  C = prisim_ctx::Catalog();
}

}  // namespace pint

extern "C" {

int prisim_hip_set_catalog(prisim_ctx* ctx, const prisim_catalog* cat) {
  return guarded(ctx, [&]() -> int {
  if (!ctx) return PRISIM_EINVAL;
  if (!cat) return fail(ctx, PRISIM_EINVAL, "cat is NULL");
  if (!ctx->array_set) return fail(ctx, PRISIM_ESTATE, "set_array must be called before set_catalog (the spectra are per channel grid)");
  const int64_t n = cat->nsrc;
  if (n < 0 || n > (int64_t)0x7fff0000) return fail(ctx, PRISIM_EINVAL, "nsrc must be in [0, 2^31)");
  if (cat->coords < PRISIM_COORDS_RADEC || cat->coords > PRISIM_COORDS_ALTAZ) return fail(ctx, PRISIM_EINVAL, "unknown catalogue coordinates");
  if (n > 0 && !cat->location && !cat->unitvec) return fail(ctx, PRISIM_EINVAL, "location and unitvec are both NULL");
  const bool have_spec = cat->flux_spectrum != nullptr;
  if (n > 0 && !have_spec && (!cat->flux_ref || !cat->spindex)) return fail(ctx, PRISIM_EINVAL, "flux_ref / spindex is NULL and no flux_spectrum given");
  if (!have_spec && !(cat->ref_freq_hz > 0.0)) return fail(ctx, PRISIM_EINVAL, "ref_freq_hz must be positive");
  // ---- every input is validated BEFORE the resident catalogue is touched: a rejected call leaves the previous sky in place ----
  std::vector<double> lon, lat, uvec[3], kap;
  if (cat->unitvec) {
    for (auto& v : uvec) v.resize((size_t)n);
    for (int64_t i = 0; i < n; ++i) {
      double r2 = 0.0;
      for (int k = 0; k < 3; ++k) {
        const double v = cat->unitvec[3 * i + k];
        if (!std::isfinite(v)) return fail(ctx, PRISIM_EINVAL, "non-finite catalogue unit vector");
        uvec[k][(size_t)i] = v;
        r2 += v * v;
      }
      if (std::fabs(r2 - 1.0) > 1e-9) return fail(ctx, PRISIM_EINVAL, "catalogue unit vectors must have unit length (to 1e-9)");
    }
  } else {
    lon.resize((size_t)n); lat.resize((size_t)n);
    for (int64_t i = 0; i < n; ++i) {
      lon[(size_t)i] = cat->location[2 * i]; lat[(size_t)i] = cat->location[2 * i + 1];
      if (!std::isfinite(lon[(size_t)i]) || !std::isfinite(lat[(size_t)i])) return fail(ctx, PRISIM_EINVAL, "non-finite catalogue position");
    }
  }
  std::vector<prisim_ctx::Catalog::Run> runs;
  double kappa_max = 0.0;
  std::vector<uint8_t> run_id;
  if (cat->fwhm_deg && n > 0) {
    kap.resize((size_t)n);
    for (int64_t i = 0; i < n; ++i) {
      const double fw = cat->fwhm_deg[i];
      if (!std::isfinite(fw) || fw < 0.0) return fail(ctx, PRISIM_EINVAL, "invalid source FWHM");
      const double fd = 2.0 * std::sin(0.5 * fw * M_PI / 180.0);          // interferometry.py:6268-6283, as upload_common
      kap[(size_t)i] = M_LN2 * fd * fd;
      kappa_max = std::max(kappa_max, kap[(size_t)i]);
    }
    run_id.assign((size_t)n, 0);
    int64_t lo = 0;
    bool ok = true;
    for (int64_t s = 1; s <= n && ok; ++s) {
      if (s == n || kap[(size_t)s] != kap[(size_t)lo]) {
        if (runs.size() == (size_t)PRISIM_CAT_MAX_RUNS) { ok = false; break; }
        for (int64_t j = lo; j < s; ++j) run_id[(size_t)j] = (uint8_t)runs.size();
        runs.push_back({lo, s, kap[(size_t)lo]});
        lo = s;
      }
    }
    if (!ok) runs.clear();                                                   // sizes vary source by source: no runs (like the uploaded path)
  }
  if (n > 0 && have_spec) {
    for (size_t i = 0; i < (size_t)n * (size_t)ctx->nchan; ++i)
      if (!std::isfinite(cat->flux_spectrum[i])) return fail(ctx, PRISIM_EINVAL, "non-finite flux spectrum");
  } else if (n > 0) {
    for (int64_t i = 0; i < n; ++i)
      if (!std::isfinite(cat->flux_ref[i]) || !std::isfinite(cat->spindex[i])) return fail(ctx, PRISIM_EINVAL, "non-finite flux_ref / spindex");
  }
  HIPCHK(ctx, hipSetDevice(ctx->device));
  auto& C = ctx->cat;
  // a set of the previous catalogue may still be read by queued work
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  if (C.gstream) HIPCHK(ctx, hipStreamSynchronize(C.gstream));
  if (ctx->prep_stream) HIPCHK(ctx, hipStreamSynchronize(ctx->prep_stream));
  C.loaded = false;
  ctx->sky_set = ctx->sky_set && C.cur < 0;       // a sky that lives in the old catalogue's buffers is gone
  C.cur = -1;
  C.runs = runs;
  C.kappa_max = kappa_max;
  int rc;
  const size_t nb = (size_t)std::max<int64_t>(n, 1) * sizeof(double);
  if ((rc = ensure(ctx, C.ux, nb)) || (rc = ensure(ctx, C.uy, nb)) || (rc = ensure(ctx, C.uz, nb))) return rc;
  if (n > 0) {
    if (cat->unitvec) {
      HIPCHK(ctx, hipMemcpy(C.ux.p, uvec[0].data(), (size_t)n * sizeof(double), hipMemcpyHostToDevice));
      HIPCHK(ctx, hipMemcpy(C.uy.p, uvec[1].data(), (size_t)n * sizeof(double), hipMemcpyHostToDevice));
      HIPCHK(ctx, hipMemcpy(C.uz.p, uvec[2].data(), (size_t)n * sizeof(double), hipMemcpyHostToDevice));
    } else {
      if ((rc = ensure(ctx, C.lon, nb)) || (rc = ensure(ctx, C.lat, nb))) return rc;
      HIPCHK(ctx, hipMemcpy(C.lon.p, lon.data(), (size_t)n * sizeof(double), hipMemcpyHostToDevice));
      HIPCHK(ctx, hipMemcpy(C.lat.p, lat.data(), (size_t)n * sizeof(double), hipMemcpyHostToDevice));
      HIPCHK(ctx, launch_cat_prepare((const double*)C.lon.p, (const double*)C.lat.p, cat->coords, (double*)C.ux.p, (double*)C.uy.p, (double*)C.uz.p, n, ctx->stream));
    }
    if (!kap.empty()) {
      if ((rc = ensure(ctx, C.kappa, nb)) || (rc = ensure(ctx, C.run_id, (size_t)n))) return rc;
      HIPCHK(ctx, hipMemcpy(C.kappa.p, kap.data(), (size_t)n * sizeof(double), hipMemcpyHostToDevice));
      if (!C.runs.empty()) HIPCHK(ctx, hipMemcpy(C.run_id.p, run_id.data(), (size_t)n, hipMemcpyHostToDevice));
    }
    if (have_spec) {
      const size_t sb = (size_t)n * (size_t)ctx->nchan * sizeof(double);
      if ((rc = ensure(ctx, C.spec, sb))) return rc;
      HIPCHK(ctx, hipMemcpy(C.spec.p, cat->flux_spectrum, sb, hipMemcpyHostToDevice));
    } else {
      if ((rc = ensure(ctx, C.flux_ref, nb)) || (rc = ensure(ctx, C.spindex, nb))) return rc;
      HIPCHK(ctx, hipMemcpy(C.flux_ref.p, cat->flux_ref, (size_t)n * sizeof(double), hipMemcpyHostToDevice));
      HIPCHK(ctx, hipMemcpy(C.spindex.p, cat->spindex, (size_t)n * sizeof(double), hipMemcpyHostToDevice));
    }
  }
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  C.n = n;
  C.coords = cat->coords;
  C.have_shape = cat->fwhm_deg != nullptr;
  C.have_spec = have_spec;
  C.ref_freq = have_spec ? 1.0 : cat->ref_freq_hz;
  release(C.sort_tmp);                 // sized per catalogue
  for (auto& s : C.set) s.ev_recorded = s.prep_recorded = false;
  for (auto& r : C.tabfree_rec) r = false;
  C.loaded = true;
  return PRISIM_OK;
  });
}

int prisim_hip_set_sky_from_catalog(prisim_ctx* ctx, const prisim_obs* obs, const prisim_snapshot* snap, int64_t* nsrc_roi) {
  return guarded(ctx, [&]() -> int {
  if (!ctx) return PRISIM_EINVAL;
  int rc;
  if ((rc = check_obs(ctx, obs, snap, 1))) return rc;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  auto& C = ctx->cat;
  const bool want_keys = cat_sort_wanted(ctx);
  const int b = C.next;
  ctx->sky_set = false;
  if ((rc = geometry_run(ctx, obs, snap, 1, b, want_keys))) return rc;
  if ((rc = activate_snapshot(ctx, obs, *snap, b, 0, want_keys))) return rc;
  C.next = b ^ 1;
  if (nsrc_roi) *nsrc_roi = ctx->nsrc;
  return PRISIM_OK;
  });
}

int prisim_hip_catalog_roi(prisim_ctx* ctx, const prisim_obs* obs, const prisim_snapshot* snap, int64_t* nsrc_roi, int64_t* indices, double* dircos,
                           int64_t cap) {
  return guarded(ctx, [&]() -> int {
  if (!ctx) return PRISIM_EINVAL;
  if (!obs || !snap) return fail(ctx, PRISIM_EINVAL, "obs / snapshot is NULL");
  if (!ctx->array_set || !ctx->cat.loaded) return fail(ctx, PRISIM_ESTATE, "set_array and set_catalog must be called first");
  if (!std::isfinite(obs->latitude_deg) || !std::isfinite(obs->roi_radius_deg) || (!snap->frame_given && !std::isfinite(snap->lst_deg)))
    return fail(ctx, PRISIM_EINVAL, "non-finite latitude / roi_radius / LST");
  if (int rcf = check_frame(ctx, *snap)) return rcf;
  if (obs->roi_center != 0 && obs->roi_center != 1) return fail(ctx, PRISIM_EINVAL, "roi_center must be 0 (zenith) or 1 (pointing centre)");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  auto& C = ctx->cat;
  int rc;
  if ((rc = geometry_run(ctx, obs, snap, 1, 2, false))) return rc;
  const int64_t N = C.out_host[0].nsrc;
  if (nsrc_roi) *nsrc_roi = N;
  if ((indices || dircos) && cap < N) return fail(ctx, PRISIM_EINVAL, "cap is smaller than the region of interest");
  if (N > 0 && indices) {
    std::vector<int32_t> h((size_t)N);
    HIPCHK(ctx, hipMemcpy(h.data(), C.set[2].idx.p, (size_t)N * sizeof(int32_t), hipMemcpyDeviceToHost));
    for (int64_t i = 0; i < N; ++i) indices[i] = h[(size_t)i];
  }
  if (N > 0 && dircos) {
    std::vector<double> h((size_t)N * 4);
    HIPCHK(ctx, hipMemcpy(h.data(), C.set[2].dirs.p, (size_t)N * 4 * sizeof(double), hipMemcpyDeviceToHost));
    for (int64_t i = 0; i < N; ++i)
      for (int k = 0; k < 3; ++k) dircos[3 * i + k] = h[(size_t)(4 * i + k)];
  }
  return PRISIM_OK;
  });
}

int prisim_hip_observe_catalog(prisim_ctx* ctx, const prisim_obs* obs, const prisim_snapshot* snaps, int64_t nsnap, int precision, int want_grad,
                               int64_t slot0, int64_t* nsrc_roi, const prisim_post* post) {
  return guarded(ctx, [&]() -> int {
  if (!ctx) return PRISIM_EINVAL;
  if (nsnap <= 0) return fail(ctx, PRISIM_EINVAL, "nsnap must be positive");
  int rc;
  if ((rc = check_obs(ctx, obs, snaps, nsnap))) return rc;
  if (precision != PRISIM_FP64 && precision != PRISIM_FP32) return fail(ctx, PRISIM_EINVAL, "unknown precision");
  if (slot0 < 0 || slot0 + nsnap > ctx->nt_max) return fail(ctx, PRISIM_EINVAL, "slot range outside the device cube (set_array nt_max)");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  auto& C = ctx->cat;
  const bool want_keys = cat_sort_wanted(ctx);
  // chunks of snapshots whose geometry (44 bytes per catalogue source and snapshot) stays under 512 MiB
  const int64_t per_snap = 44 * std::max<int64_t>(C.n, 1);
  int64_t chunk = std::max<int64_t>(1, std::min<int64_t>(64, ((int64_t)512 << 20) / per_snap));
  if (nsnap > 64 && wave_batch_eligible(ctx, obs, precision, want_grad, nsnap)) {
    // small arrays whose snapshots share one launch: up to 256 snapshots per chunk (a 64-snapshot launch of HERA-19 lasts 1.7 ms and
    // starts from the idle clock; 256 snapshots fill the grid without source splits -- no partial cubes, no reduction pass) while the
    // chunk's beam x flux and packed rows (2 x 8 B per catalogue source and channel, at most) stay under 4 GiB
    int64_t big = 256;
    if (const char* env = getenv("PRISIM_HIP_BATCH_CHUNK")) big = std::max<int64_t>(1, atoll(env));
    const int64_t per_batch = 16 * std::max<int64_t>(C.n, 1) * std::max<int64_t>(ctx->nchan, 1) + per_snap;
    chunk = std::max<int64_t>(chunk, std::min<int64_t>(big, ((int64_t)4 << 30) / per_batch));
  }
  for (int64_t c0 = 0; c0 < nsnap; c0 += chunk) {
    const int64_t kc = std::min(chunk, nsnap - c0);
    const int b = C.next;
    ctx->sky_set = false;
    if (C.n > 0 && wave_batch_eligible(ctx, obs, precision, want_grad, kc)) {
      if ((rc = run_wave_batch(ctx, obs, snaps + c0, b, kc, slot0 + c0, nsrc_roi ? nsrc_roi + c0 : nullptr, want_keys, want_grad != 0))) return rc;
      for (int64_t t = 0; t < kc; ++t)
        if ((rc = post_snapshot(ctx, post, slot0 + c0 + t))) return rc;
      C.next = b ^ 1;
      continue;
    }
    { HostSpan sp("geometry_run"); if ((rc = geometry_run(ctx, obs, snaps + c0, kc, b, want_keys))) return rc; }
    for (int64_t t = 0; t < kc; ++t) {
      { HostSpan sp("activate_snapshot"); if ((rc = activate_snapshot(ctx, obs, snaps[c0 + t], b, t, want_keys))) return rc; }
      if (nsrc_roi) nsrc_roi[c0 + t] = ctx->nsrc;
      const int64_t slot = slot0 + c0 + t;
      { HostSpan sp("compute"); if ((rc = prisim_hip_compute(ctx, precision, PRISIM_KERNEL_AUTO, want_grad, slot))) return rc; }
      if ((rc = post_snapshot(ctx, post, slot))) return rc;
    }
    C.next = b ^ 1;
  }
  return PRISIM_OK;
  });
}

}  // extern "C"
