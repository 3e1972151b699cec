// capi.cpp -- C-ABI of libprisim_hip.so (declared in include/prisim_hip.h).
//
// Owns the per-GPU context: HIP stream, resident array (baselines, channels), per-snapshot sky,
// the device visibility cube, and lazily dlopen()ed rocFFT / RCCL handles.  No C++ exception
// crosses the ABI; every export returns 0 or a negative PRISIM_E* code.
#include "ctx_internal.h"

namespace pint {

std::string g_create_error;
RcclApi g_rccl;
RocfftApi g_rocfft;

bool load_rccl(std::string& err) {
  if (g_rccl.handle) return true;
  // The ROCm RCCL this library was compiled against (<rccl/rccl.h> of /opt/rocm) first, by absolute path: a bare soname would
  // resolve to whatever librccl the process already carries (a python process that imported torch carries torch's bundled one).
  void* h = nullptr;
  std::string path;
  if (const char* env = getenv("PRISIM_RCCL_LIB")) { h = dlopen(env, RTLD_NOW | RTLD_LOCAL); if (h) path = env; }
  for (const char* cand : {"/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so", "librccl.so.1", "librccl.so"}) {
    if (h) break;
    h = dlopen(cand, RTLD_NOW | RTLD_LOCAL);
    if (h) path = cand;
  }
  if (!h) { err = std::string("cannot load librccl: ") + dlerror(); return false; }
  bool ok = load_sym(h, "ncclGetUniqueId", g_rccl.GetUniqueId) && load_sym(h, "ncclCommInitRank", g_rccl.CommInitRank) &&
            load_sym(h, "ncclAllGather", g_rccl.AllGather) && load_sym(h, "ncclCommDestroy", g_rccl.CommDestroy) &&
            load_sym(h, "ncclGetErrorString", g_rccl.GetErrorString);
  if (!ok) { err = "librccl lacks a required symbol"; dlclose(h); return false; }
  if (!(load_sym(h, "ncclSend", g_rccl.Send) && load_sym(h, "ncclRecv", g_rccl.Recv) && load_sym(h, "ncclGroupStart", g_rccl.GroupStart) &&
        load_sym(h, "ncclGroupEnd", g_rccl.GroupEnd)))
    g_rccl.Send = nullptr;                              // gather-to-root then reports PRISIM_ELIB
  (void)load_sym(h, "ncclGetVersion", g_rccl.GetVersion);
  (void)load_sym(h, "ncclGetLastError", g_rccl.GetLastError);
  g_rccl.path = path;
  g_rccl.handle = h;
  return true;
}

bool load_rocfft(std::string& err) {
  if (g_rocfft.handle) return true;
  void* h = dlopen("librocfft.so.0", RTLD_NOW | RTLD_LOCAL);
  if (!h) h = dlopen("librocfft.so", RTLD_NOW | RTLD_LOCAL);
  if (!h) h = dlopen("/opt/rocm/lib/librocfft.so.0", RTLD_NOW | RTLD_LOCAL);
  if (!h) { err = std::string("cannot load librocfft: ") + dlerror(); return false; }
  RocfftApi& a = g_rocfft;
  bool ok = load_sym(h, "rocfft_setup", a.setup) && load_sym(h, "rocfft_plan_create", a.plan_create) &&
            load_sym(h, "rocfft_plan_destroy", a.plan_destroy) &&
            load_sym(h, "rocfft_plan_get_work_buffer_size", a.plan_get_work_buffer_size) &&
            load_sym(h, "rocfft_execution_info_create", a.execution_info_create) &&
            load_sym(h, "rocfft_execution_info_destroy", a.execution_info_destroy) &&
            load_sym(h, "rocfft_execution_info_set_stream", a.execution_info_set_stream) &&
            load_sym(h, "rocfft_execution_info_set_work_buffer", a.execution_info_set_work_buffer) &&
            load_sym(h, "rocfft_execute", a.execute);
  if (!ok) { err = "librocfft lacks a required symbol"; dlclose(h); return false; }
  a.handle = h;
  return true;
}

Plan make_plan(const prisim_ctx* ctx, int precision, int kernel) {
  Plan pl{};
  pl.f32 = (precision == PRISIM_FP32);
  pl.kernel = kernel;
  if (kernel == PRISIM_KERNEL_AUTO) pl.kernel = ctx->uniform ? PRISIM_KERNEL_RECURRENCE : PRISIM_KERNEL_DIRECT;
  const int64_t nbl = ctx->nbl, nchan = ctx->nchan, nsrc = ctx->nsrc;
  pl.nbgroups = (int)((nbl + kBlockThreads - 1) / kBlockThreads);
  // channel tile: the cheapest tile (seed + 5 instructions per term, tiles past nchan are wasted work) that still yields enough
  // blocks to fill 256 CUs x 4 blocks
  // (fp64 48-channel tiles were tried: 5.98 instead of 6.47 instructions per term, but 22 tiles x 239 groups = 10.3 rounds of resident
  // blocks end in a 7 % tail against 14.9 rounds at 32 channels -- measured 128 ms against 125 ms on the same box)
  int max_ct = pl.f32 ? 64 : 32;
  {
    // coarse channel grids with the taper run the exact per-step amplitude recurrence (the grouped form needs df/f_min <= 3.4e-3,
    // run_pass): its rounding grows with the chain length (5.8e-6 of one term at 32 steps and df/f = 3 %, tools/fuzz_parity.py),
    // so such runs use 32-channel tiles (16 steps from the seed)
    const double fmin = std::min(std::fabs(ctx->f0), std::fabs(ctx->f0 + ctx->df * (double)(ctx->nchan - 1)));
    if (pl.f32 && ctx->taper && !(fmin > 0.0 && std::fabs(ctx->df) <= 3.4e-3 * fmin)) max_ct = 32;
  }
  int ct = ctx->tune_ct;
  if (ct == 0) {
    const int64_t want_blocks = 1024;
    const int64_t max_split = std::max<int64_t>(1, nsrc / 64);
    const double seed = pl.f32 ? 30.0 : 48.0;             // instructions per (source, baseline, tile) outside the pair loop (ISA census)
    const double per_term = pl.f32 ? 2.5 : 5.0;
    double best = 0.0;
    ct = 8;
    // Small problems (config 2: 171 baselines = 3 wavefronts) cannot fill the chip whatever the tiling; what matters then is that
    // every SIMD has work and that the seed is amortised over as many channels as the wave count allows.  Measured on config 2
    // (tools/small_problem_census.py, profiles/r03_small_problem_census.jsonl): fp64 16-channel tiles 75.6 us against 85.8 at 8 and
    // 81.0 at 32 (fp64 wants ~2 waves per SIMD to cover its dependent-issue latency), packed fp32 32-channel tiles 39.8 us against 54.2
    // at 8 (a single wave issues v_pk_fma_f32 at near the full rate, the 2-cycle v_fma_f32 of the narrow tiles does not).  So a wider
    // tile is admitted as soon as it still yields 1024 (fp64) / 512 (fp32) active wavefronts, even if that is fewer than 1024 blocks.
    const int64_t waves_per_group = std::min<int64_t>(kBlockThreads / 64, (nbl + 63) / 64);
    const int64_t active_waves_per_tile = (int64_t)(pl.nbgroups - 1) * (kBlockThreads / 64) +
                                          std::min<int64_t>(kBlockThreads / 64, (nbl - (int64_t)(pl.nbgroups - 1) * kBlockThreads + 63) / 64);
    (void)waves_per_group;
    const int64_t want_waves = pl.f32 ? 512 : 1024;
    for (int cand : {64, 32, 16, 8}) {
      if (cand > max_ct) continue;
      const int64_t tiles = (nchan + cand - 1) / cand;
      if (cand > 8 && tiles * pl.nbgroups * max_split < want_blocks && tiles * active_waves_per_tile * max_split < want_waves) continue;
      const double cost = (double)tiles * (seed + per_term * cand);
      if (best == 0.0 || cost < best * (1.0 - 1e-9)) { best = cost; ct = cand; }
    }
  }
  if (ct > max_ct) ct = max_ct;          // a tuning request never overrides the accuracy cap (or the register budget) above
  pl.ct = ct;
  pl.pk = pl.f32 && (ct == 32 || ct == 64) && pl.kernel == PRISIM_KERNEL_RECURRENCE;
  pl.ntiles = (int)((nchan + ct - 1) / ct);
  // source chunk: granularity of the zero padding and of the source split (the kernels stream rows through scalar loads)
  const int esz = pl.f32 ? 4 : 8;
  int chunk = ctx->tune_chunk ? ctx->tune_chunk : 64;
  const int max_chunk = 16384 / (ct * esz);
  if (chunk > max_chunk) chunk = max_chunk;
  if (chunk > 256) chunk = 256;
  if (chunk < 1) chunk = 1;
  pl.chunk = chunk;
  pl.nsrc_pad = round_up(std::max<int64_t>(nsrc, 1), chunk);
  // source split so that small problems still fill the GPU; each split is a multiple of the chunk
  int nsplit = ctx->tune_nsplit;
  if (nsplit == 0) {
    const int64_t base = (int64_t)pl.ntiles * pl.nbgroups;
    // resident blocks per CU = waves per SIMD the kernels are built for (PK_WAVES / WavesPerEU in skyvis_kernels.hip)
    // (fp64: the 16 KiB phasor table + 36 KiB flush buffer + prefetch area = 56 KiB of LDS per block allow 2 blocks per CU)
    const int per_cu = pl.pk ? 2 : (pl.f32 ? 4 : 2);
    const int64_t slots = (int64_t)std::max(ctx->cu_count, 1) * per_cu;
    // The grid runs in rounds of `slots` resident blocks.  Measured on 1/2, 1/4 and 1/8 baseline shards of config 3
    // (tools/shard_nsplit_sweep.py, candidates alternating): the step time is lowest when the sources are split so that the grid
    // is again about 7.5 rounds deep, like the unsplit full problem (nsplit = 2 / 4 / 8: 1.00 / 1.02 / 1.03 x the ideal T1/N
    // against 1.03 / 1.06 / 1.10 x unsplit) -- blocks then start and finish out of step and the tail is short; every extra split
    // costs partial-cube traffic, so much deeper grids lose again.  Problems that are already >= 6 rounds deep keep nsplit = 1
    // (alternating A/B on the full config 3, tools/full_nsplit.py: 1 / 2 / 4 splits within 0.4 % of each other).
    nsplit = 1;
    int64_t want = 1;
    if (base * 10 < slots * 60) want = (slots * 15 / 2 + base / 2) / base;       // round(7.5 * slots / base)
    want = std::min<int64_t>(want, std::max<int64_t>(1, nsrc / 32));             // keep >= 32 sources per split
    {
      // ... and every split writes a partial cube that k_reduce_partials reads again: keep that traffic (~3 TB/s effective) under
      // ~3 % of the sky-sum's own time, but never below one round of resident blocks.  One rank's share of config 4 at N = 8
      // (1016 bl x 768 ch x 24 576 sources): 2.86 ms at 16 splits against 3.13 ms at the 64 the round rule alone asks for.
      const double rate = pl.f32 ? (ctx->taper ? 7.0e12 : 1.0e13) : (ctx->taper ? 2.8e12 : 5.0e12);       // terms/s of the big kernels
      const double t_compute = (double)nbl * (double)nchan * (double)nsrc / rate;
      const double t_split = 2.0 * (double)nbl * (double)nchan * (pl.f32 ? 8.0 : 16.0) / 3.0e12;
      const int64_t cap = std::max<int64_t>((slots + base - 1) / std::max<int64_t>(base, 1), (int64_t)(0.03 * t_compute / t_split));
      // (splits up to 16 keep the measured round rule.)  Deeper ones: the sweep's optimum sits where the grid is ~1.5 rounds of
      // resident blocks -- the second, half-empty round runs one block per CU, and a lone wave per SIMD issues packed FMAs at 3/4 of
      // the full rate -- 2.53 ms at 16 splits against 3.28 at 19 and 2.75 at 10 or 24 on the config-4 share.
      if (want > 16) want = std::max<int64_t>(16, std::min<int64_t>((3 * slots / 2 + base / 2) / std::max<int64_t>(base, 1), cap));
    }
    nsplit = (int)std::max<int64_t>(1, std::min<int64_t>(want, 64));
  }
  if (!pl.f32 && ctx->taper && (ct == 16 || ct == 32) && pl.kernel == PRISIM_KERNEL_RECURRENCE && nbl <= kBlockThreads && !ctx->tune_chunk &&
      taper_f64_grouped_enabled()) {
    // One baseline group, fp64 with the taper: wave items (run_pass) -- two wavefronts on every SIMD, and at least 16 sources per
    // item.  Config 2 (3 baseline waves x 16 tiles): 42 splits of 36 sources = 2016 wavefronts in 512 blocks.  A requested split
    // count (set_tuning) is cut by the same rule: a single launch then partitions a sky exactly as a batched launch
    // (prisim_hip_observe_catalog) with that split count does.
    const int64_t nbw = (nbl + 63) / 64;
    int64_t s_want;
    if (ctx->tune_nsplit) {
      s_want = ctx->tune_nsplit;
    } else {
      const int64_t waves = 2LL * 4 * std::max(ctx->cu_count, 1);
      s_want = std::max<int64_t>(1, waves / (pl.ntiles * nbw));
      s_want = std::min<int64_t>(s_want, std::max<int64_t>(1, nsrc / 16));
    }
    if (s_want > 1) {
      const int64_t per = wave_split_sources(nsrc, s_want);
      pl.chunk = 4;
      pl.nsrc_pad = round_up(std::max<int64_t>(nsrc, 1), 16);
      pl.src_per_split = per;
      pl.nsplit = (int)std::max<int64_t>(1, (nsrc + per - 1) / per);
      return pl;
    }
  }
  if (!ctx->tune_chunk && (!pl.f32 || ctx->tune_nsplit)) {
    // (fp64; the packed fp32 kernel's time barely moves with the split count -- 40.3 us at 47 splits against 42.2 at 24 on config 2 -- and
    // every split costs its share of k_reduce_partials: 65.7 us per snapshot against 61.7, r04_cfg2_probe_f32.jsonl.)
    // A small sky cut into many pieces (config 2: 1504 sources, 32 splits fill the 512 block slots exactly): 64-source chunks would cap
    // the split count at nsrc / 64 = 24 -- 62 us against 76 with 16-source chunks and 32 splits (tools/config2_fullwave_probe.py)
    auto realised = [&](int c) {          // split count a chunk size allows: whole chunks per split
      const int64_t nch = round_up(std::max<int64_t>(nsrc, 1), c) / c;
      const int64_t per = (nch + nsplit - 1) / std::max(nsplit, 1);
      return (nch + per - 1) / std::max<int64_t>(per, 1);
    };
    while (chunk > 16 && realised(chunk) < nsplit) chunk /= 2;
    pl.chunk = chunk;
    pl.nsrc_pad = round_up(std::max<int64_t>(nsrc, 1), chunk);
  }
  const int64_t nchunks = pl.nsrc_pad / chunk;
  if (nsplit > nchunks) nsplit = (int)nchunks;
  if (nsplit < 1) nsplit = 1;
  const int64_t chunks_per_split = (nchunks + nsplit - 1) / nsplit;
  pl.src_per_split = chunks_per_split * chunk;
  pl.nsplit = (int)((nchunks + chunks_per_split - 1) / chunks_per_split);
  return pl;
}

}  // namespace pint


extern "C" {

const char* prisim_hip_version(void) { return "prisim_hip 0.5 gfx950"; }     // 0.5: the snapshot carries its frame; shard map

const char* prisim_hip_last_error(const prisim_ctx* ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

int prisim_hip_create(int device, prisim_ctx** out) {
  return guarded(nullptr, [&]() -> int {
  if (!out) return fail(nullptr, PRISIM_EINVAL, "out is NULL");
  *out = nullptr;
  int ndev = 0;
  hipError_t e = hipGetDeviceCount(&ndev);
  if (e != hipSuccess || ndev <= 0)
    return fail(nullptr, PRISIM_ENODEV, std::string("no HIP device available: ") + hipGetErrorString(e));
  if (device < 0 || device >= ndev)
    return fail(nullptr, PRISIM_EINVAL, "device index out of range (" + std::to_string(ndev) + " visible)");
  prisim_ctx* ctx = new (std::nothrow) prisim_ctx();
  if (!ctx) return fail(nullptr, PRISIM_ENOMEM, "out of host memory");
  ctx->device = device;
  e = hipSetDevice(device);
  if (e == hipSuccess) e = hipStreamCreate(&ctx->stream);
  for (int i = 0; i < prisim_ctx::kTimingRing && e == hipSuccess; ++i) {
    if ((e = hipEventCreate(&ctx->ev_c0[i])) != hipSuccess || (e = hipEventCreate(&ctx->ev_c1[i])) != hipSuccess ||
        (e = hipEventCreate(&ctx->ev_k0[i])) != hipSuccess || (e = hipEventCreate(&ctx->ev_k1[i])) != hipSuccess)
      break;
  }
  if (e != hipSuccess) {
    std::string m = std::string("HIP context setup failed: ") + hipGetErrorString(e);
    prisim_hip_destroy(ctx);
    return fail(nullptr, PRISIM_ENODEV, m);
  }
  // the catalogue path's two priority streams and their events belong to the context too (5 ms to create: not in front of a run's first
  // snapshot); a failure here is not fatal -- the path tries again when it is first used and reports it then
  if (catalog_streams(ctx) != PRISIM_OK) { ctx->err.clear(); (void)hipGetLastError(); }
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device) == hipSuccess) {
    snprintf(ctx->devname, sizeof(ctx->devname), "%s (%s)", prop.name, prop.gcnArchName);
    ctx->cu_count = prop.multiProcessorCount;
    ctx->clock_khz = prop.clockRate;
  }
  *out = ctx;
  return PRISIM_OK;
  });
}

void prisim_hip_destroy(prisim_ctx* ctx) {
  if (!ctx) return;
  (void)hipSetDevice(ctx->device);
  if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
  catalog_destroy(ctx);
  if (ctx->comm_stream) (void)hipStreamSynchronize(ctx->comm_stream);
  if (ctx->comm && g_rccl.CommDestroy) g_rccl.CommDestroy(ctx->comm);
  if (ctx->copy_stream) (void)hipStreamSynchronize(ctx->copy_stream);
  if (ctx->ev_slot_done) (void)hipEventDestroy(ctx->ev_slot_done);
  for (int i = 0; i < prisim_ctx::kCommRing; ++i)
    for (hipEvent_t ev : {ctx->ev_gc[i], ctx->ev_g0[i], ctx->ev_g1[i], ctx->ev_gu[i]})
      if (ev) (void)hipEventDestroy(ev);
  if (ctx->comm_stream) (void)hipStreamDestroy(ctx->comm_stream);
  if (ctx->ev_copy_ready) (void)hipEventDestroy(ctx->ev_copy_ready);
  if (ctx->copy_stream) (void)hipStreamDestroy(ctx->copy_stream);
  if (ctx->fft_plan && g_rocfft.plan_destroy) g_rocfft.plan_destroy(ctx->fft_plan);
  if (ctx->fft_info && g_rocfft.execution_info_destroy) g_rocfft.execution_info_destroy(ctx->fft_info);
  for (DevBuf* b : {&ctx->blx, &ctx->bly, &ctx->blz, &ctx->freqs, &ctx->fsq, &ctx->fsq_pairs, &ctx->cube, &ctx->grad, &ctx->dirs,
                    &ctx->partial, &ctx->scratch, &ctx->gathered, &ctx->sendbuf, &ctx->shard_map, &ctx->stage_main, &ctx->stage_comm, &ctx->ext_table,
                    &ctx->ext_work, &ctx->ext_colmax, &ctx->sky_flux, &ctx->sky_sp, &ctx->sky_bf, &ctx->sky_flag,
                    &ctx->dl_stage, &ctx->grp_hz, &ctx->fft_work, &ctx->fft_buf, &ctx->dt_out, &ctx->dt_pow, &ctx->dt_wts, &ctx->dt_lag_all, &ctx->dt_pow_all, &ctx->dt_tw})
    release(*b);
  for (SkyBufs& k : ctx->skb) {
    for (DevBuf* b : {&k.pb, &k.packed, &k.dirs_prep, &k.dirs_c32, &k.lift_flags, &k.split_flags, &k.moments, &k.moments_part, &k.split_count, &k.cull_first,
                      &k.dirs_sorted, &k.idx_sorted, &k.batch_tab})
      release(*b);
    if (k.ev_prep) (void)hipEventDestroy(k.ev_prep);
    if (k.ev_sum) (void)hipEventDestroy(k.ev_sum);
  }
  for (int i = 0; i < prisim_ctx::kTimingRing; ++i)
    for (hipEvent_t ev : {ctx->ev_c0[i], ctx->ev_c1[i], ctx->ev_k0[i], ctx->ev_k1[i]})
      if (ev) (void)hipEventDestroy(ev);
  if (ctx->ev_stage) (void)hipEventDestroy(ctx->ev_stage);
  if (ctx->ev_d0) (void)hipEventDestroy(ctx->ev_d0);
  if (ctx->ev_d1) (void)hipEventDestroy(ctx->ev_d1);
  if (ctx->h_stage) (void)hipHostFree(ctx->h_stage);
  if (ctx->h_split_count) (void)hipHostFree(ctx->h_split_count);
  if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
  delete ctx;
}

int prisim_hip_set_array(prisim_ctx* ctx, const double* bl_enu, int64_t nbl, const double* freqs_hz, int64_t nchan,
                         int64_t nt_max) {
  return guarded(ctx, [&]() -> int {
  if (!ctx) return PRISIM_EINVAL;
  if (!bl_enu || !freqs_hz) return fail(ctx, PRISIM_EINVAL, "bl_enu / freqs_hz is NULL");
  if (nbl <= 0 || nchan <= 0 || nt_max <= 0) return fail(ctx, PRISIM_EINVAL, "nbl, nchan and nt_max must be positive");
  if (nbl > (int64_t)1 << 30 || nchan > (int64_t)1 << 24) return fail(ctx, PRISIM_EINVAL, "nbl or nchan too large");
  for (int64_t i = 0; i < nchan; ++i)
    if (!std::isfinite(freqs_hz[i])) return fail(ctx, PRISIM_EINVAL, "non-finite channel frequency");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  ctx->array_set = false;
  ctx->sky_set = false;
  ctx->ext_nside = 0;   // the external-beam table is per channel grid
  if (ctx->cat.gstream) HIPCHK(ctx, hipStreamSynchronize(ctx->cat.gstream));
  if (ctx->prep_stream) HIPCHK(ctx, hipStreamSynchronize(ctx->prep_stream));
  ctx->prep_async = false;
  ctx->sk = &ctx->skb[0];
  ctx->cat.loaded = false;   // ... and so are the catalogue's spectra
  ctx->cat.cur = -1;
  std::vector<double> x(nbl), y(nbl), z(nbl);
  for (int64_t b = 0; b < nbl; ++b) {
    x[b] = bl_enu[3 * b]; y[b] = bl_enu[3 * b + 1]; z[b] = bl_enu[3 * b + 2];
    if (!std::isfinite(x[b]) || !std::isfinite(y[b]) || !std::isfinite(z[b]))
      return fail(ctx, PRISIM_EINVAL, "non-finite baseline component");
  }
  int rc;
  const size_t bb = (size_t)nbl * sizeof(double);
  if ((rc = ensure(ctx, ctx->blx, bb)) || (rc = ensure(ctx, ctx->bly, bb)) || (rc = ensure(ctx, ctx->blz, bb))) return rc;
  ctx->nchan_pad = round_up(nchan, 64);
  if ((rc = ensure(ctx, ctx->freqs, (size_t)nchan * sizeof(double)))) return rc;
  if ((rc = ensure(ctx, ctx->fsq, (size_t)ctx->nchan_pad * sizeof(float)))) return rc;
  const size_t cube_bytes = (size_t)nt_max * nbl * nchan * 2 * sizeof(double);
  if ((rc = ensure(ctx, ctx->cube, cube_bytes))) return rc;
  HIPCHK(ctx, hipMemcpyAsync(ctx->blx.p, x.data(), bb, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(ctx, hipMemcpyAsync(ctx->bly.p, y.data(), bb, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(ctx, hipMemcpyAsync(ctx->blz.p, z.data(), bb, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(ctx, hipMemcpyAsync(ctx->freqs.p, freqs_hz, (size_t)nchan * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(ctx, hipMemsetAsync(ctx->cube.p, 0, cube_bytes, ctx->stream));
  HIPCHK(ctx, launch_fsq((const double*)ctx->freqs.p, (float*)ctx->fsq.p, nchan, ctx->nchan_pad, 1e-8, ctx->stream));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  ctx->h_freqs.assign(freqs_hz, freqs_hz + nchan);
  ctx->grp_maxlen.assign((size_t)((nbl + kBlockThreads - 1) / kBlockThreads), 0.0);
  ctx->grp_maxh.assign(ctx->grp_maxlen.size(), 0.0);
  ctx->grp_maxz.assign(ctx->grp_maxlen.size(), 0.0);
  ctx->grp_minh.assign(ctx->grp_maxlen.size(), 1e300);
  ctx->fsq_pairs_ct = ctx->fsq_pairs_ntiles = -1;
  for (SkyBufs& k : ctx->skb) k.lift_key_k = -1.0;             // the cached lifting flags belong to the previous array
  for (int64_t b = 0; b < nbl; ++b) {
    const double len = std::sqrt(x[b] * x[b] + y[b] * y[b] + z[b] * z[b]);
    const size_t g = (size_t)(b / kBlockThreads);
    if (len > ctx->grp_maxlen[g]) ctx->grp_maxlen[g] = len;
    const double hl = std::sqrt(x[b] * x[b] + y[b] * y[b]);
    if (hl < ctx->grp_minh[g]) ctx->grp_minh[g] = hl;
    if (hl > ctx->grp_maxh[g]) ctx->grp_maxh[g] = hl;
    if (std::fabs(z[b]) > ctx->grp_maxz[g]) ctx->grp_maxz[g] = std::fabs(z[b]);
  }
  {
    const size_t ng = ctx->grp_maxh.size();
    std::vector<double> hz(4 * ng);
    for (size_t g = 0; g < ng; ++g) { hz[g] = ctx->grp_maxh[g]; hz[ng + g] = ctx->grp_maxz[g]; hz[2 * ng + g] = ctx->grp_maxlen[g]; hz[3 * ng + g] = ctx->grp_minh[g]; }
    if ((rc = ensure(ctx, ctx->grp_hz, 4 * ng * sizeof(double)))) return rc;
    HIPCHK(ctx, hipMemcpy(ctx->grp_hz.p, hz.data(), 4 * ng * sizeof(double), hipMemcpyHostToDevice));
  }
  ctx->nbl = nbl; ctx->nchan = nchan; ctx->nt_max = nt_max;
  // uniform channel grid?  f_k = f0 + k*df to within 1e-7 Hz (phase error <= 1e-13 cycles at 1 us delay)
  ctx->f0 = freqs_hz[0];
  ctx->df = nchan > 1 ? (freqs_hz[nchan - 1] - freqs_hz[0]) / (double)(nchan - 1) : 0.0;
  ctx->uniform = true;
  for (int64_t k = 0; k < nchan; ++k)
    if (std::fabs(freqs_hz[k] - (ctx->f0 + (double)k * ctx->df)) > 1e-7) { ctx->uniform = false; break; }
  release(ctx->grad);
  release(ctx->gathered);
  release(ctx->sendbuf);
  release(ctx->stage_main); release(ctx->stage_comm);
  if (ctx->nbl_total > 0 && (int64_t)ctx->shard_map_h.size() != (int64_t)ctx->nranks * nbl) {   // a shard map is per shard size
    ctx->nbl_total = 0; ctx->shard_map_h.clear(); release(ctx->shard_map);
  }
  release(ctx->dt_lag_all);
  release(ctx->dt_pow_all);
  ctx->dt_nt = ctx->dt_nout = 0;
  ctx->dt_have_lag = ctx->dt_have_pow = false;
  ctx->array_set = true;
  return PRISIM_OK;
  });
}

// Directions + source-shape constants of one snapshot -> ctx->dirs, through the pinned staging area (no synchronisation).
// `extra_stage_bytes`: staging the caller will use for its own small uploads in the same group (it calls stage_end()).
}  // extern "C"
namespace pint {
int upload_common(prisim_ctx* ctx, int64_t nsrc, const double* dircos, const double* pc_dircos,
                         const double* fwhm_deg, size_t extra_stage_bytes) {
  if (nsrc < 0) return fail(ctx, PRISIM_EINVAL, "nsrc must be non-negative");
  if (nsrc > (int64_t)0x7fff0000) return fail(ctx, PRISIM_EINVAL, "nsrc must be below 2^31 (the kernels index sources with 32 bits)");
  if (nsrc > 0 && !dircos) return fail(ctx, PRISIM_EINVAL, "dircos is NULL");
  if (!pc_dircos) return fail(ctx, PRISIM_EINVAL, "pc_dircos is NULL");
  for (int i = 0; i < 3; ++i)
    if (!std::isfinite(pc_dircos[i])) return fail(ctx, PRISIM_EINVAL, "non-finite pc_dircos");
  const size_t d4_bytes = (size_t)std::max<int64_t>(nsrc, 1) * 4 * sizeof(double);
  const size_t cull_bytes = 2 * 8 * ctx->grp_maxlen.size() * sizeof(int32_t) + 512;      // [precision][<= 8 runs][groups]
  int rc;
  if ((rc = stage_begin(ctx, d4_bytes + extra_stage_bytes + cull_bytes + 4096))) return rc;
  double* d4 = (double*)stage_alloc(ctx, d4_bytes);
  if (!d4) return fail(ctx, PRISIM_EINTERNAL, "staging area too small");
  d4[0] = d4[1] = d4[2] = d4[3] = 0.0;
  double dmax2 = 0.0;
  for (int64_t s = 0; s < nsrc; ++s) {
    {
      const double ex = dircos[3 * s] - pc_dircos[0], ey = dircos[3 * s + 1] - pc_dircos[1], ez = dircos[3 * s + 2] - pc_dircos[2];
      const double e2 = ex * ex + ey * ey + ez * ez;
      if (e2 > dmax2) dmax2 = e2;
    }
    for (int i = 0; i < 3; ++i) {
      const double v = dircos[3 * s + i];
      if (!std::isfinite(v)) return fail(ctx, PRISIM_EINVAL, "non-finite direction cosine");
      d4[4 * s + i] = v;
    }
    double kappa = 0.0;
    if (fwhm_deg) {
      const double fw = fwhm_deg[s];
      if (!std::isfinite(fw) || fw < 0.0) return fail(ctx, PRISIM_EINVAL, "invalid source FWHM");
      // interferometry.py:6268-6283:  w = exp(-ln2 * (2 sin(FWHM/2))^2 * (|b|^2 - (b.s)^2) * f^2 / c^2)
      const double fd = 2.0 * std::sin(0.5 * fw * M_PI / 180.0);
      kappa = M_LN2 * fd * fd;
    }
    d4[4 * s + 3] = kappa;
  }
  ctx->kappa_runs.clear();
  if (fwhm_deg && nsrc > 0) {
    constexpr size_t kMaxRuns = 8;
    int64_t lo = 0;
    for (int64_t s = 1; s <= nsrc; ++s) {
      if (s == nsrc || d4[4 * s + 3] != d4[4 * lo + 3]) {
        if (ctx->kappa_runs.size() == kMaxRuns) { ctx->kappa_runs.clear(); break; }     // sizes vary source by source: no runs
        ctx->kappa_runs.push_back({lo, s, d4[4 * lo + 3], (int)ctx->kappa_runs.size()});
        lo = s;
      }
    }
  }
  if ((rc = ensure(ctx, ctx->dirs, d4_bytes))) return rc;
  HIPCHK(ctx, stage_send(ctx, ctx->dirs.p, d4, d4_bytes));
  if (ctx->prep_async) {             // the previous sky was prepared on the preparation stream: let it drain, then back to one stream
    HIPCHK(ctx, hipStreamSynchronize(ctx->prep_stream));
    ctx->prep_async = false;
  }
  ctx->sk = &ctx->skb[0];
  ctx->dirs_p = static_cast<const double*>(ctx->dirs.p);       // an uploaded sky: no catalogue indirection
  ctx->src_index = nullptr;
  ctx->cat.cur = -1;
  // Taper culling.  w = exp(-kappa (|b|^2 - (b.s)^2) f^2/c^2), and for a baseline of horizontal length h and height z and a source at
  // (rho, n) = (sin, cos) of the zenith angle, (b.s)^2 <= (h rho + |z| |n|)^2, i.e. |b|^2 - (b.s)^2 >= (h |n| - |z| rho)^2 when
  // h |n| >= |z| rho (the identity h^2 + z^2 - (h rho + |z||n|)^2 = (h|n| - |z| rho)^2).  So for every baseline of a group (smallest
  // horizontal length Hmin, largest |b_z| Z) and every channel the exponent of source s is at least
  //   x_s = kappa_s max(Hmin |n_s| - Z rho_s, 0)^2 fmin^2/c^2 .
  // The leading sources of a run whose x_s >= T contribute together at most exp(-T) sum|pbflux|: T = 18 (1.5e-8) for fp32 requests --
  // far inside the 5e-6 tolerance -- and the packed fp32 kernels start the group's source loop behind them; T = 28 (7e-13 against
  // fp64's 1e-11) for fp64 requests, used by the grouped fp64 kernel k_skyvis_taper_f64 (run_pass).
  // Long baselines over coarse diffuse pixels (config 4: MWA to 2.5 km, nside 64) shed the sources nearest the zenith this way when
  // the caller lists a run's sources by decreasing altitude (InterferometerArray.observe does); unordered skies just cull little.
  ctx->cull_any[0] = ctx->cull_any[1] = false;
  ctx->cull_frac[0] = ctx->cull_frac[1] = 0.0;
  ctx->cull_nruns = 0;
  {
    const char* env = getenv("PRISIM_HIP_TAPER_CULL");
    const size_t ng = ctx->grp_maxlen.size();
    const size_t nruns = ctx->kappa_runs.size();
    if (nruns > 0 && ng > 0 && !(env && atoi(env) == 0)) {
      const double fmin = std::min(std::fabs(ctx->h_freqs.front()), std::fabs(ctx->h_freqs.back()));
      const double fc2 = (fmin / kC) * (fmin / kC);
      int32_t* tab = (int32_t*)stage_alloc(ctx, 2 * nruns * ng * sizeof(int32_t));
      if (tab) {
        const double thr[2] = {28.0, 18.0};                      // index = precision (PRISIM_FP64 = 0, PRISIM_FP32 = 1)
        double culled[2] = {0.0, 0.0};
        // Serial host work per snapshot, so kept small: a run is only looked at when its longest-baseline group could cull anything at
        // all, (rho, |n|) of its sources are formed once (not per group and precision), and the fp64 walk (the higher threshold) never
        // goes past where the fp32 walk of the same group stopped.
        double hmax = 0.0;
        for (size_t g = 0; g < ng; ++g) hmax = std::max(hmax, ctx->grp_minh[g]);
        std::vector<double>& rho = ctx->cull_rho;
        std::vector<double>& an = ctx->cull_an;
        for (size_t r = 0; r < nruns; ++r) {
          const auto& run = ctx->kappa_runs[r];
          const bool any_possible = run.kappa > 0.0 && run.kappa * hmax * hmax * fc2 >= thr[1];
          int64_t nlead = 0;                                       // leading sources whose (rho, |n|) are formed so far
          if (any_possible) { rho.clear(); an.clear(); }
          for (size_t g = 0; g < ng; ++g) {
            int64_t first32 = run.lo;
            for (int pr = 1; pr >= 0; --pr) {                      // fp32 (threshold 18) first: the fp64 walk (28) stops no later
              int64_t sfirst = run.lo;
              const int64_t limit = pr == 1 ? run.hi : first32;
              if (any_possible && run.kappa * ctx->grp_minh[g] * ctx->grp_minh[g] * fc2 >= thr[pr]) {
                const double H = ctx->grp_minh[g], Z = ctx->grp_maxz[g];
                while (sfirst < limit) {
                  const int64_t i = sfirst - run.lo;
                  if (i >= nlead) {
                    const double l = d4[4 * sfirst], m = d4[4 * sfirst + 1];
                    rho.push_back(std::sqrt(l * l + m * m));
                    an.push_back(std::fabs(d4[4 * sfirst + 2]));
                    nlead = i + 1;
                  }
                  const double perp = H * an[(size_t)i] - Z * rho[(size_t)i];
                  if (!(perp > 0.0 && run.kappa * perp * perp * fc2 >= thr[pr])) break;
                  ++sfirst;
                }
              }
              if (pr == 1) first32 = sfirst;
              tab[(pr * nruns + r) * ng + g] = (int32_t)sfirst;
              if (sfirst > run.lo) {
                ctx->cull_any[pr] = true;
                const int64_t nb = std::min<int64_t>(kBlockThreads, ctx->nbl - (int64_t)g * kBlockThreads);
                culled[pr] += (double)(sfirst - run.lo) * (double)nb;
              }
            }
          }
        }
        if (ctx->cull_any[0] || ctx->cull_any[1]) {
          if ((rc = ensure(ctx, ctx->sk->cull_first, 2 * nruns * ng * sizeof(int32_t)))) return rc;
          HIPCHK(ctx, stage_send(ctx, ctx->sk->cull_first.p, tab, 2 * nruns * ng * sizeof(int32_t)));
          ctx->cull_nruns = (int)nruns;
          for (int pr = 0; pr < 2; ++pr) ctx->cull_frac[pr] = culled[pr] / ((double)nsrc * (double)ctx->nbl);
        }
      }
    }
  }
  ctx->nsrc = nsrc;
  ctx->dmax = std::sqrt(dmax2);
  ctx->taper = fwhm_deg != nullptr;
  for (int i = 0; i < 3; ++i) ctx->pc[i] = pc_dircos[i];
  return PRISIM_OK;
}

// Upload a caller array of `bytes`: staged (asynchronous) when small, otherwise straight from the caller's memory followed by
// a stream synchronisation (the caller may reuse the array as soon as the call returns).
int upload_any(prisim_ctx* ctx, void* dst, const void* src, size_t bytes, bool* synced) {
  if (bytes <= kStageMaxBytes && ctx->h_stage_used + bytes + 256 <= ctx->h_stage_bytes) {
    HIPCHK(ctx, stage_upload(ctx, dst, src, bytes));
    return PRISIM_OK;
  }
  HIPCHK(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  if (synced) *synced = true;
  return PRISIM_OK;
}
}  // namespace pint
extern "C" {

int prisim_hip_set_sky(prisim_ctx* ctx, const prisim_sky* sky) {
  return guarded(ctx, [&]() -> int {
  if (!ctx) return PRISIM_EINVAL;
  if (!sky) return fail(ctx, PRISIM_EINVAL, "sky is NULL");
  if (!ctx->array_set) return fail(ctx, PRISIM_ESTATE, "set_array must be called before set_sky");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  ctx->sky_set = false;
  if (sky->nsrc > 0 && !sky->pbflux) return fail(ctx, PRISIM_EINVAL, "pbflux is NULL");
  const int64_t n = sky->nsrc * ctx->nchan;
  const size_t pb_bytes = (size_t)std::max<int64_t>(n, 0) * (sky->pbflux_is_f32 ? sizeof(float) : sizeof(double));
  const size_t fl_bytes = sky->fluxes ? (size_t)std::max<int64_t>(n, 0) * sizeof(double) : 0;
  const size_t stage_extra = (pb_bytes <= kStageMaxBytes ? pb_bytes : 0) + (fl_bytes <= kStageMaxBytes ? fl_bytes : 0) + 1024;
  int rc = upload_common(ctx, sky->nsrc, sky->dircos, sky->pc_dircos, sky->fwhm_deg, stage_extra);
  if (rc) return rc;
  if ((rc = ensure(ctx, ctx->sk->pb, (size_t)std::max<int64_t>(n, 1) * sizeof(double)))) return rc;
  if (n > 0) {
    if (sky->pbflux_is_f32) {
      if ((rc = ensure(ctx, ctx->sky_flux, (size_t)n * sizeof(float)))) return rc;
      if ((rc = upload_any(ctx, ctx->sky_flux.p, sky->pbflux, (size_t)n * sizeof(float), nullptr))) return rc;
      HIPCHK(ctx, launch_f32_to_f64((const float*)ctx->sky_flux.p, (double*)ctx->sk->pb.p, n, ctx->stream));
    } else {
      if ((rc = upload_any(ctx, ctx->sk->pb.p, sky->pbflux, (size_t)n * sizeof(double), nullptr))) return rc;
    }
    if (sky->fluxes) {   // pbfluxes = pb * fluxes (:6254) on the device
      if ((rc = ensure(ctx, ctx->sky_sp, (size_t)n * sizeof(double)))) return rc;
      if ((rc = upload_any(ctx, ctx->sky_sp.p, sky->fluxes, (size_t)n * sizeof(double), nullptr))) return rc;
      HIPCHK(ctx, launch_mul_inplace((double*)ctx->sk->pb.p, (const double*)ctx->sky_sp.p, n, ctx->stream));
    }
  }
  stage_end(ctx);
  ctx->sky_set = true;
  return PRISIM_OK;
  });
}

}  // extern "C"
namespace pint {

// argument checks of a fused analytic beam (prisim_beam_sky / prisim_obs)
int check_beam_spec(prisim_ctx* ctx, int beam_kind, double diameter_m, const double* beam_pc_dircos, const prisim_beam_ext* ext) {
  if (beam_kind < PRISIM_BEAM_DELTA || beam_kind > PRISIM_BEAM_POLY) return fail(ctx, PRISIM_EINVAL, "unknown beam_kind");
  if (beam_kind == PRISIM_BEAM_DIPOLE && !ext) return fail(ctx, PRISIM_EINVAL, "PRISIM_BEAM_DIPOLE needs a prisim_beam_ext (dipole axis)");
  if (beam_kind == PRISIM_BEAM_POLY && !ext) return fail(ctx, PRISIM_EINVAL, "PRISIM_BEAM_POLY needs a prisim_beam_ext (poly_coef)");
  if (ext) {
    const prisim_beam_ext* x = ext;
    if (x->dipole_mode < PRISIM_DIPOLE_GENERAL || x->dipole_mode > PRISIM_DIPOLE_HALFWAVE)
      return fail(ctx, PRISIM_EINVAL, "unknown dipole_mode");
    if (x->array_nax1 < 0 || x->array_nax2 < 0 || (x->array_nax1 > 0) != (x->array_nax2 > 0))
      return fail(ctx, PRISIM_EINVAL, "array_nax1 and array_nax2 must both be positive or both zero");
    if (x->array_nax1 > 0 && !(x->array_sep1 > 0.0 && x->array_sep2 > 0.0))
      return fail(ctx, PRISIM_EINVAL, "array element separations must be positive");
    if (!std::isfinite(x->ground_height) || !std::isfinite(x->array_east2ax1_deg))
      return fail(ctx, PRISIM_EINVAL, "non-finite beam extension parameter");
    if (x->bf_nelem < 0 || x->bf_nelem > 4096) return fail(ctx, PRISIM_EINVAL, "bf_nelem must be in 0 ... 4096");
    if (x->bf_nelem > 0) {
      if (x->bf_nrand < 1 || x->bf_nrand > 256) return fail(ctx, PRISIM_EINVAL, "bf_nrand must be in 1 ... 256");
      if (!x->bf_pos || !x->bf_delays || !x->bf_gains) return fail(ctx, PRISIM_EINVAL, "beamformer positions / delays / gains is NULL");
      if (x->array_nax1 > 0) return fail(ctx, PRISIM_EINVAL, "the beamformer replaces the analytic array factor: set array_nax1 = array_nax2 = 0");
      for (int64_t i = 0; i < (int64_t)x->bf_nelem * 3; ++i)
        if (!std::isfinite(x->bf_pos[i])) return fail(ctx, PRISIM_EINVAL, "non-finite beamformer element position");
      for (int64_t i = 0; i < (int64_t)x->bf_nelem * x->bf_nrand; ++i)
        if (!std::isfinite(x->bf_delays[i]) || !std::isfinite(x->bf_gains[i]))
          return fail(ctx, PRISIM_EINVAL, "non-finite beamformer delay / gain");
    }
  }
  if (beam_kind != PRISIM_BEAM_DELTA && beam_kind != PRISIM_BEAM_POLY && !(diameter_m > 0.0))
    return fail(ctx, PRISIM_EINVAL, "diameter_m must be positive");
  if (!beam_pc_dircos) return fail(ctx, PRISIM_EINVAL, "beam_pc_dircos is NULL");
  return PRISIM_OK;
}

size_t beamformer_doubles(const prisim_beam_ext* ext) {
  const int bf_n = ext ? ext->bf_nelem : 0, bf_r = bf_n > 0 ? ext->bf_nrand : 0;
  return (size_t)bf_n * 3 + 2 * (size_t)bf_n * bf_r;
}

// pbflux[s][f] = beam(dirs_p[s], f) x flux(src_index ? src_index[s] : s, f) of the current sky (ns sources) into ctx->sk->pb.  The flux
// pointers are device arrays (power law: d_flux_ref / d_spindex; tabulated: d_flux_spec).  A beamformer's element arrays (host, in ext)
// are sent through the OPEN staging group (the caller ran stage_begin with room for beamformer_doubles(ext) doubles and ends it).
int sky_beam_flux(prisim_ctx* ctx, int64_t ns, int beam_kind, double diameter_m, const double* beam_pc_dircos, const prisim_beam_ext* ext,
                  const double* d_flux_ref, const double* d_spindex, const double* d_flux_spec, double ref_freq, const int32_t* src_index) {
  int rc;
  const int bf_n = ext ? ext->bf_nelem : 0, bf_r = bf_n > 0 ? ext->bf_nrand : 0;
  if ((bf_n > 0 && (rc = ensure(ctx, ctx->sky_bf, beamformer_doubles(ext) * sizeof(double)))) || (rc = ensure(ctx, ctx->sky_flag, sizeof(int32_t))))
    return rc;
  if (bf_n > 0) {
    double* b = (double*)ctx->sky_bf.p;
    if ((rc = upload_any(ctx, b, ext->bf_pos, (size_t)bf_n * 3 * sizeof(double), nullptr)) ||
        (rc = upload_any(ctx, b + (size_t)bf_n * 3, ext->bf_delays, (size_t)bf_n * bf_r * sizeof(double), nullptr)) ||
        (rc = upload_any(ctx, b + (size_t)bf_n * 3 + (size_t)bf_n * bf_r, ext->bf_gains, (size_t)bf_n * bf_r * sizeof(double), nullptr)))
      return rc;
  }
  BeamParams bp{};
  bp.dirs = ctx->dirs_p;
  bp.flux_ref = d_flux_ref;
  bp.spindex = d_spindex;
  bp.flux_spec = d_flux_spec;
  bp.src_index = src_index;
  bp.freqs = (const double*)ctx->freqs.p;
  bp.ref_freq = d_flux_spec ? 1.0 : ref_freq;
  bp.beam_kind = beam_kind;
  bp.diameter = diameter_m;
  bp.bpc_x = beam_pc_dircos[0]; bp.bpc_y = beam_pc_dircos[1]; bp.bpc_z = beam_pc_dircos[2];
  if (ext) {
    const prisim_beam_ext* x = ext;
    bp.dip_x = x->dipole_dircos[0]; bp.dip_y = x->dipole_dircos[1]; bp.dip_z = x->dipole_dircos[2];
    bp.dipole_mode = x->dipole_mode;
    bp.nax1 = x->array_nax1; bp.nax2 = x->array_nax2; bp.sep1 = x->array_sep1; bp.sep2 = x->array_sep2;
    const double ang = x->array_east2ax1_deg * M_PI / 180.0;
    bp.rot_c = std::cos(ang); bp.rot_s = std::sin(ang);
    bp.apc_x = x->array_pc_dircos[0]; bp.apc_y = x->array_pc_dircos[1]; bp.apc_z = x->array_pc_dircos[2];
    bp.gp_height = x->ground_height; bp.gp_modify = x->ground_modify; bp.gp_scale = x->ground_scale; bp.gp_max = x->ground_max;
    if (bf_n > 0) {
      bp.bf_nelem = bf_n; bp.bf_nrand = bf_r;
      bp.bf_pos = (const double*)ctx->sky_bf.p;
      bp.bf_delays = bp.bf_pos + (size_t)bf_n * 3;
      bp.bf_gains = bp.bf_delays + (size_t)bf_n * bf_r;
    }
  }
  if (beam_kind == PRISIM_BEAM_POLY)
    for (int i = 0; i < 4; ++i) bp.poly[i] = ext->poly_coef[i];
  bp.flag = (int32_t*)ctx->sky_flag.p;
  bp.nsrc = ns; bp.nchan = ctx->nchan;
  bp.pb_out = (double*)ctx->sk->pb.p;
  HIPCHK(ctx, hipMemsetAsync(ctx->sky_flag.p, 0, sizeof(int32_t), pstream(ctx)));
  HIPCHK(ctx, launch_beam_flux(bp, pstream(ctx)));
  return PRISIM_OK;
}

// only the polynomial beams can trip the reference's validity checks (:510-512, :802-807): the one case that reads back
int check_poly_beam_flag(prisim_ctx* ctx) {
  int32_t hflag = 0;
  HIPCHK(ctx, hipMemcpyAsync(&hflag, ctx->sky_flag.p, sizeof(int32_t), hipMemcpyDeviceToHost, pstream(ctx)));
  HIPCHK(ctx, hipStreamSynchronize(pstream(ctx)));
  if (hflag & 2)
    return fail(ctx, PRISIM_EINVAL, "Primary beam values were found to be NaN in some case(s). Check if the polynomial equations are valid for the frequencies specified.");
  if (hflag & 1)
    return fail(ctx, PRISIM_EINVAL, "Primary beam exceeds unity by a significant amount. Check the validity of the Primary beam equation for the angles specified.");
  return PRISIM_OK;
}

}  // namespace pint
extern "C" {

int prisim_hip_set_sky_analytic(prisim_ctx* ctx, const prisim_beam_sky* sky) {
  return guarded(ctx, [&]() -> int {
  if (!ctx) return PRISIM_EINVAL;
  if (!sky) return fail(ctx, PRISIM_EINVAL, "sky is NULL");
  if (!ctx->array_set) return fail(ctx, PRISIM_ESTATE, "set_array must be called before set_sky_analytic");
  int rc;
  if ((rc = check_beam_spec(ctx, sky->beam_kind, sky->diameter_m, sky->beam_pc_dircos, sky->ext))) return rc;
  const bool have_spec = sky->flux_spectrum != nullptr;
  if (sky->nsrc > 0 && !have_spec && (!sky->flux_ref || !sky->spindex))
    return fail(ctx, PRISIM_EINVAL, "flux_ref / spindex is NULL and no flux_spectrum given");
  if (!have_spec && !(sky->ref_freq_hz > 0.0)) return fail(ctx, PRISIM_EINVAL, "ref_freq_hz must be positive");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  ctx->sky_set = false;
  const int64_t ns = sky->nsrc;
  const int64_t n = ns * ctx->nchan;
  const size_t frb = have_spec ? (size_t)n * sizeof(double) : (size_t)ns * sizeof(double);
  const size_t stage_extra = (frb <= kStageMaxBytes ? frb : 0) + (size_t)ns * sizeof(double) + beamformer_doubles(sky->ext) * sizeof(double) + 4096;
  rc = upload_common(ctx, ns, sky->dircos, sky->pc_dircos, sky->fwhm_deg, stage_extra);
  if (rc) return rc;
  if ((rc = ensure(ctx, ctx->sk->pb, (size_t)std::max<int64_t>(n, 1) * sizeof(double)))) return rc;
  if (ns > 0) {
    if ((rc = ensure(ctx, ctx->sky_flux, frb)) || (rc = ensure(ctx, ctx->sky_sp, (size_t)ns * sizeof(double)))) return rc;
    if ((rc = upload_any(ctx, ctx->sky_flux.p, have_spec ? sky->flux_spectrum : sky->flux_ref, frb, nullptr))) return rc;
    if (!have_spec && (rc = upload_any(ctx, ctx->sky_sp.p, sky->spindex, (size_t)ns * sizeof(double), nullptr))) return rc;
    if ((rc = sky_beam_flux(ctx, ns, sky->beam_kind, sky->diameter_m, sky->beam_pc_dircos, sky->ext, have_spec ? nullptr : (const double*)ctx->sky_flux.p,
                            have_spec ? nullptr : (const double*)ctx->sky_sp.p, have_spec ? (const double*)ctx->sky_flux.p : nullptr,
                            sky->ref_freq_hz, nullptr)))
      return rc;
    stage_end(ctx);
    if (sky->beam_kind == PRISIM_BEAM_POLY && (rc = check_poly_beam_flag(ctx))) return rc;
  } else {
    stage_end(ctx);
  }
  ctx->sky_set = true;
  return PRISIM_OK;
  });
}

int prisim_hip_set_external_beam(prisim_ctx* ctx, const double* beam, int64_t npix, int64_t nfreq, const double* interp_matrix) {
  return guarded(ctx, [&]() -> int {
  if (!ctx) return PRISIM_EINVAL;
  if (!ctx->array_set) return fail(ctx, PRISIM_ESTATE, "set_array must be called before set_external_beam");
  if (!beam || !interp_matrix) return fail(ctx, PRISIM_EINVAL, "beam / interp_matrix is NULL");
  if (npix < 12 || nfreq < 1) return fail(ctx, PRISIM_EINVAL, "npix must be 12*nside^2 and nfreq >= 1");
  const int64_t nside = (int64_t)std::llround(std::sqrt((double)npix / 12.0));
  if (12 * nside * nside != npix || nside > (1 << 13)) return fail(ctx, PRISIM_EINVAL, "npix is not 12*nside^2");
  for (int64_t i = 0; i < npix * nfreq; ++i)
    if (!(beam[i] > 0.0) || !std::isfinite(beam[i]))
      return fail(ctx, PRISIM_EINVAL, "external beam values must be finite and > 0 (log10 is interpolated, run_prisim.py:2094)");
  for (int64_t i = 0; i < ctx->nchan * nfreq; ++i)
    if (!std::isfinite(interp_matrix[i])) return fail(ctx, PRISIM_EINVAL, "non-finite spectral interpolation weight");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  ctx->ext_nside = 0;
  DevBuf dbeam, dm;
  int rc;
  if ((rc = ensure(ctx, dbeam, (size_t)npix * nfreq * sizeof(double))) || (rc = ensure(ctx, dm, (size_t)ctx->nchan * nfreq * sizeof(double))) ||
      (rc = ensure(ctx, ctx->ext_table, (size_t)npix * ctx->nchan * sizeof(double)))) {
    release(dbeam); release(dm);
    return rc;
  }
  hipError_t e = hipMemcpyAsync(dbeam.p, beam, (size_t)npix * nfreq * sizeof(double), hipMemcpyHostToDevice, ctx->stream);
  if (e == hipSuccess) e = hipMemcpyAsync(dm.p, interp_matrix, (size_t)ctx->nchan * nfreq * sizeof(double), hipMemcpyHostToDevice, ctx->stream);
  if (e == hipSuccess) e = launch_extbeam_table((const double*)dbeam.p, (const double*)dm.p, (double*)ctx->ext_table.p, npix, nfreq, ctx->nchan, ctx->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
  release(dbeam); release(dm);
  HIPCHK(ctx, e);
  ctx->ext_nside = (int)nside;
  return PRISIM_OK;
  });
}

// beam table -> pbflux for the current directions; fluxes: device [nsrc][nchan] table, or NULL with flux_ref / spindex (device [nsrc])
}  // extern "C"
namespace pint {
int extbeam_sky(prisim_ctx* ctx, int64_t nsrc, const double* d_fluxes, const double* d_flux_ref, const double* d_spindex,
                double ref_freq, const int32_t* src_index) {
  const int64_t n = nsrc * ctx->nchan;
  int rc;
  // (grown with headroom: the region of interest of a drift scan grows by a few sources per snapshot, and an exact fit would re-allocate
  // -- and so drain the queue of -- almost every snapshot)
  if ((rc = ensure_grow(ctx, ctx->ext_work, (size_t)n * sizeof(double))) ||
      (rc = ensure(ctx, ctx->ext_colmax, (size_t)1025 * ctx->nchan * sizeof(double))))
    return rc;
  HIPCHK(ctx, launch_extbeam_sky((const double*)ctx->ext_table.p, ctx->ext_nside, ctx->dirs_p, d_fluxes, d_flux_ref, d_spindex,
                                 (const double*)ctx->freqs.p, ref_freq, (double*)ctx->ext_work.p, (double*)ctx->ext_colmax.p,
                                 (double*)ctx->sk->pb.p, nsrc, ctx->nchan, pstream(ctx), src_index));
  return PRISIM_OK;
}
}  // namespace pint
extern "C" {

int prisim_hip_set_sky_external(prisim_ctx* ctx, const prisim_sky* sky) {
  return guarded(ctx, [&]() -> int {
  if (!ctx) return PRISIM_EINVAL;
  if (!sky) return fail(ctx, PRISIM_EINVAL, "sky is NULL");
  if (!ctx->array_set) return fail(ctx, PRISIM_ESTATE, "set_array must be called before set_sky_external");
  if (ctx->ext_nside <= 0) return fail(ctx, PRISIM_ESTATE, "set_external_beam must be called before set_sky_external");
  if (sky->pbflux) return fail(ctx, PRISIM_EINVAL, "pbflux must be NULL: the beam comes from the external table");
  if (sky->nsrc > 0 && !sky->fluxes) return fail(ctx, PRISIM_EINVAL, "fluxes is NULL");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  ctx->sky_set = false;
  const int64_t n = sky->nsrc * ctx->nchan;
  const size_t fl_bytes = (size_t)std::max<int64_t>(n, 0) * sizeof(double);
  int rc = upload_common(ctx, sky->nsrc, sky->dircos, sky->pc_dircos, sky->fwhm_deg, (fl_bytes <= kStageMaxBytes ? fl_bytes : 0) + 1024);
  if (rc) return rc;
  if ((rc = ensure(ctx, ctx->sk->pb, (size_t)std::max<int64_t>(n, 1) * sizeof(double)))) return rc;
  if (n > 0) {
    if ((rc = ensure(ctx, ctx->sky_flux, fl_bytes))) return rc;
    if ((rc = upload_any(ctx, ctx->sky_flux.p, sky->fluxes, fl_bytes, nullptr))) return rc;
    if ((rc = extbeam_sky(ctx, sky->nsrc, (const double*)ctx->sky_flux.p, nullptr, nullptr, 1.0, nullptr))) return rc;
  }
  stage_end(ctx);
  ctx->sky_set = true;
  return PRISIM_OK;
  });
}

int prisim_hip_set_sky_external_analytic(prisim_ctx* ctx, const prisim_beam_sky* sky) {
  return guarded(ctx, [&]() -> int {
  if (!ctx) return PRISIM_EINVAL;
  if (!sky) return fail(ctx, PRISIM_EINVAL, "sky is NULL");
  if (!ctx->array_set) return fail(ctx, PRISIM_ESTATE, "set_array must be called before set_sky_external_analytic");
  if (ctx->ext_nside <= 0) return fail(ctx, PRISIM_ESTATE, "set_external_beam must be called before set_sky_external_analytic");
  const bool have_spec = sky->flux_spectrum != nullptr;
  if (sky->nsrc > 0 && !have_spec && (!sky->flux_ref || !sky->spindex))
    return fail(ctx, PRISIM_EINVAL, "flux_ref / spindex is NULL and no flux_spectrum given");
  if (!have_spec && !(sky->ref_freq_hz > 0.0)) return fail(ctx, PRISIM_EINVAL, "ref_freq_hz must be positive");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  ctx->sky_set = false;
  const int64_t ns = sky->nsrc;
  const int64_t n = ns * ctx->nchan;
  const size_t frb = have_spec ? (size_t)std::max<int64_t>(n, 0) * sizeof(double) : (size_t)std::max<int64_t>(ns, 0) * sizeof(double);
  int rc = upload_common(ctx, ns, sky->dircos, sky->pc_dircos, sky->fwhm_deg,
                         (frb <= kStageMaxBytes ? frb : 0) + (size_t)std::max<int64_t>(ns, 0) * sizeof(double) + 2048);
  if (rc) return rc;
  if ((rc = ensure(ctx, ctx->sk->pb, (size_t)std::max<int64_t>(n, 1) * sizeof(double)))) return rc;
  if (ns > 0) {
    if ((rc = ensure(ctx, ctx->sky_flux, frb)) || (rc = ensure(ctx, ctx->sky_sp, (size_t)ns * sizeof(double)))) return rc;
    if ((rc = upload_any(ctx, ctx->sky_flux.p, have_spec ? sky->flux_spectrum : sky->flux_ref, frb, nullptr))) return rc;
    if (!have_spec && (rc = upload_any(ctx, ctx->sky_sp.p, sky->spindex, (size_t)ns * sizeof(double), nullptr))) return rc;
    if ((rc = extbeam_sky(ctx, ns, have_spec ? (const double*)ctx->sky_flux.p : nullptr, have_spec ? nullptr : (const double*)ctx->sky_flux.p,
                          have_spec ? nullptr : (const double*)ctx->sky_sp.p, have_spec ? 1.0 : sky->ref_freq_hz, nullptr)))
      return rc;
  }
  stage_end(ctx);
  ctx->sky_set = true;
  return PRISIM_OK;
  });
}

int prisim_hip_get_pbflux(prisim_ctx* ctx, double* out) {
  return guarded(ctx, [&]() -> int {
  if (!ctx) return PRISIM_EINVAL;
  if (!ctx->sky_set) return fail(ctx, PRISIM_ESTATE, "no sky set");
  if (!out) return fail(ctx, PRISIM_EINVAL, "out is NULL");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  const int64_t n = ctx->nsrc * ctx->nchan;
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  if (ctx->prep_async) HIPCHK(ctx, hipStreamSynchronize(ctx->prep_stream));
  if (n > 0) HIPCHK(ctx, hipMemcpy(out, ctx->sk->pb.p, (size_t)n * sizeof(double), hipMemcpyDeviceToHost));
  return PRISIM_OK;
  });
}

// kernel parameters common to every sky-sum launch of the current array / sky / plan
static void fill_params(prisim_ctx* ctx, const Plan& pl, SkyvisParams& p) {
  p.bl_x = (const double*)ctx->blx.p; p.bl_y = (const double*)ctx->bly.p; p.bl_z = (const double*)ctx->blz.p;
  p.nbl = ctx->nbl; p.nchan = ctx->nchan;
  p.f0 = ctx->f0; p.df = ctx->df; p.inv_c = 1.0 / kC;
  p.dirs = ctx->dirs_p;
  p.dirs_prep = (const double*)ctx->sk->dirs_prep.p;
  p.pb_packed = ctx->sk->packed.p;
  p.fsq = (const float*)ctx->fsq.p;
  p.fsq_pairs = (const float*)ctx->fsq_pairs.p;
  p.lift_flags = ctx->sk->lift_flags.p ? (const int32_t*)ctx->sk->lift_flags.p : nullptr;   // taper kernels: which groups need no re-anchoring
  p.fsq_scale = 1e16;
  p.nsrc = ctx->nsrc; p.nsrc_pad = pl.nsrc_pad;
  p.pc_x = ctx->pc[0]; p.pc_y = ctx->pc[1]; p.pc_z = ctx->pc[2];
  p.taper = ctx->taper ? 1 : 0;
  p.ntiles = pl.ntiles; p.nbgroups = pl.nbgroups; p.nsplit = pl.nsplit; p.src_per_split = pl.src_per_split;
  p.src_chunk = pl.chunk;
  // fp32 accumulators are flushed into the fp64 cube every flush_src sources: the rounding error of a sequential
  // fp32 sum grows like eps/2*sqrt(n/3) relative to sum|pbflux| in the fully coherent worst case (1.1e-6 at n = 16384,
  // tolerance 5e-6), and every flush is a read-modify-write pass over the whole cube, so flush as rarely as that allows.
  p.flush_src = 16384;
  if (const char* env = getenv("PRISIM_HIP_FLUSH_SRC")) {        // test hook: exercise the read-modify-write flush on small skies
    const long v = atol(env);
    if (v > 0 && v < (1L << 30)) p.flush_src = (int32_t)v;
  }
  p.scale_comp = -1;
  p.src_lo = 0; p.src_hi = ctx->nsrc; p.accumulate = 0; p.kappa0 = 0.0; p.split_flags = nullptr; p.src_first = nullptr;
  {
    // grouped taper recurrence (skyvis_kernels.hip): second-order residual (11.09 (df/f)^2)^2 * 0.565 <= 1e-8 of sum|pbflux|
    const double fmin = std::min(std::fabs(ctx->f0), std::fabs(ctx->f0 + ctx->df * (double)(ctx->nchan - 1)));
    p.taper_group = (ctx->taper && fmin > 0.0 && std::fabs(ctx->df) <= 3.4e-3 * fmin) ? 1 : 0;
    if (const char* env = getenv("PRISIM_HIP_TAPER_GROUP")) p.taper_group = (atoi(env) != 0 && ctx->taper) ? 1 : 0;   // A/B hook
    ctx->timing.last_taper_group = (pl.pk && p.taper_group) ? 1 : 0;
  }
}

// Decide whether this packed fp32 taper pass runs in the split form and prepare its per-run, per-group flags.  Needs: the packed
// 64-channel kernel, the grouped recurrence's channel-grid condition, sources in <= 8 runs of one size each (exactly one when the
// sources are split into partial cubes),
// in-loop exponents kappa (|b| f / c)^2 <= 30 with a per-step exponent <= 1/8 (fp32 range and the series of exp2m1_small), and --
// per baseline group -- a bound on the parabola the uncorrected grouped form leaves: relative to a term it is at most
// 16 kappa (b.s)^2 df^2 / c^2 with (b.s)^2 <= (H rho_s + Z |n_s|)^2 (H, Z: the group's largest horizontal length and |b_z|), so
// relative to sum|pbflux| it is at most 16 kappa df^2/c^2 (H^2 M2 + 2 H Z M11 + Z^2 M02) with the beam-weighted moments
// M = sum_s |p_s| (.) / sum_s |p_s| of the run's sources, maximised over the channels (k_taper_moments: one pass over pbflux;
// k_split_flags turns them into the per-group flags ON THE DEVICE, so nothing is downloaded and a snapshot's launches never wait for
// the previous snapshot's sky-sum).  Groups above 2e-7 keep the correction (flag bit 1).
static bool taper_split_plan(prisim_ctx* ctx, const Plan& pl, const SkyvisParams& p) {
  ctx->timing.last_taper_split = 0;
  ctx->timing.last_split_uncorrected_groups = 0;
  if (!(pl.pk && ctx->taper && pl.ct == 64 && p.taper_group && !ctx->kappa_runs.empty())) return false;
  // (with a source split every run of the sky writes its own set of partial cubes: run_pass)
  if (ctx->kappa_runs.size() > (size_t)kMaxRunSets) return false;
  if (const char* env = getenv("PRISIM_HIP_TAPER_SPLIT")) { if (atoi(env) == 0) return false; }      // A/B hook
  const double fmax = std::max(std::fabs(ctx->f0), std::fabs(ctx->f0 + ctx->df * (double)(ctx->nchan - 1)));
  const double fmin = std::min(std::fabs(ctx->f0), std::fabs(ctx->f0 + ctx->df * (double)(ctx->nchan - 1)));
  double lmax = 0.0;
  for (double v : ctx->grp_maxlen) lmax = std::max(lmax, v);
  double kmax = 0.0;
  bool any_taper = false;
  for (const auto& r : ctx->kappa_runs) { kmax = std::max(kmax, r.kappa); any_taper = any_taper || r.kappa > 0.0; }
  if (!any_taper) return false;
  const double umax = kmax * (lmax * fmax / kC) * (lmax * fmax / kC);
  if (!(umax <= 30.0) || !(2.0 * umax * std::fabs(ctx->df) <= 0.125 * fmin)) return false;
  const int64_t nchan = ctx->nchan;
  const size_t nruns = ctx->kappa_runs.size();
  const size_t ng = ctx->grp_maxh.size();
  if ((size_t)pl.nbgroups != ng || !ctx->sk->lift_flags.p) return false;
  if (ensure(ctx, ctx->sk->moments, (size_t)4 * nchan * sizeof(double) * nruns) != PRISIM_OK) return false;
  if (ensure(ctx, ctx->sk->split_flags, nruns * ng * sizeof(int32_t)) != PRISIM_OK) return false;
  if (ensure(ctx, ctx->sk->split_count, 8 * sizeof(int32_t)) != PRISIM_OK) return false;
  if (!ctx->h_split_count && hipHostMalloc((void**)&ctx->h_split_count, 8 * sizeof(int32_t), hipHostMallocDefault) != hipSuccess) {
    ctx->h_split_count = nullptr;
    return false;
  }
  // moments -> flags entirely on the stream: no download, so a snapshot's launches never wait for the previous snapshot's sky-sum
  if (hipMemsetAsync(ctx->sk->split_count.p, 0, 8 * sizeof(int32_t), pstream(ctx)) != hipSuccess) return false;
  for (size_t r = 0; r < nruns; ++r) {
    const auto& run = ctx->kappa_runs[r];
    if (run.kappa <= 0.0) continue;
    double* mom = (double*)ctx->sk->moments.p + r * (size_t)4 * nchan;
    if (ensure(ctx, ctx->sk->moments_part, (size_t)taper_moments_chunks(run.lo, run.hi) * 4 * nchan * sizeof(double)) != PRISIM_OK) return false;
    if (launch_taper_moments((const double*)ctx->sk->pb.p, ctx->dirs_p, run.lo, run.hi, nchan, (double*)ctx->sk->moments_part.p, mom,
                             pstream(ctx)) != hipSuccess)
      return false;
    const double c16 = 16.0 * run.kappa * (ctx->df / kC) * (ctx->df / kC);
    if (launch_split_flags(mom, nchan, (const double*)ctx->grp_hz.p, (const double*)ctx->grp_hz.p + ng, (const int32_t*)ctx->sk->lift_flags.p, (int)ng, c16,
                           2.0e-7, (int32_t*)ctx->sk->split_flags.p + r * ng, (int32_t*)ctx->sk->split_count.p + r, pstream(ctx)) != hipSuccess)
      return false;
  }
  // the counts travel to pinned host memory behind the flags kernels; get_timing reads them after the compute's events have completed
  if (hipMemcpyAsync(ctx->h_split_count, ctx->sk->split_count.p, 8 * sizeof(int32_t), hipMemcpyDeviceToHost, pstream(ctx)) != hipSuccess) return false;
  ctx->split_count_runs = (int)nruns;
  return true;
}

// one sky-sum pass into `dst` ([nbl][nchan] complex128); scale_comp >= 0 multiplies pbflux rows by dircos[:,comp]
// Wave items (k_skyvis_taper_f64_wave): the grouped fp64 taper kernel on an array of one baseline group whose sources are split.
// PRISIM_HIP_WAVE_ITEMS=0: block items (the A/B baseline).
static bool wave_items(const prisim_ctx* ctx, const Plan& pl) {
  bool on = !pl.f32 && ctx->taper && (pl.ct == 16 || pl.ct == 32) && pl.kernel == PRISIM_KERNEL_RECURRENCE && pl.nsplit > 1 &&
            ctx->nbl <= kBlockThreads;
  if (const char* env = getenv("PRISIM_HIP_WAVE_ITEMS")) on = on && atoi(env) != 0;
  return on;
}

static int run_pass(prisim_ctx* ctx, const Plan& pl, double* dst, int scale_comp, bool timed, bool prep = false) {
  SkyvisParams p{};
  fill_params(ctx, pl, p);
  p.scale_comp = scale_comp;
  if (pl.kernel == PRISIM_KERNEL_DIRECT) {
    p.out = dst;
    if (prep) { int rcj = join_prep(ctx); if (rcj) return rcj; }
    if (timed) HIPCHK(ctx, hipEventRecord(ctx->ev_k0[ctx->ring_head], ctx->stream));
    HIPCHK(ctx, launch_skyvis_direct(p, (const double*)ctx->freqs.p, (const double*)ctx->sk->pb.p,
                                     scale_comp >= 0 ? ctx->dirs_p : nullptr, ctx->stream));
    if (timed) HIPCHK(ctx, hipEventRecord(ctx->ev_k1[ctx->ring_head], ctx->stream));
    return PRISIM_OK;
  }
  // fp64 with the source-shape taper (the reference's default precision on every run_prisim.py sky): the grouped kernel
  // k_skyvis_taper_f64 on 16- / 32-channel tiles, rows in natural channel order (PRISIM_HIP_TAPER_F64_GROUP=0: the exact second-order
  // form, k_skyvis_rec<double, CT, true> -- the A/B baseline and the 8-channel tiles' kernel)
  const bool g64 = !pl.f32 && ctx->taper && (pl.ct == 16 || pl.ct == 32) && taper_f64_grouped_enabled();
  if (prep && scale_comp < 0)      // the snapshot's first pass: rows and directions in one launch
    HIPCHK(ctx, launch_pack_prep((const double*)ctx->sk->pb.p, ctx->sk->packed.p, pl.f32, ctx->nsrc, pl.nsrc_pad, ctx->nchan, pl.ct, pl.ntiles,
                                 g64 ? 0 : 1, ctx->dirs_p, (double*)ctx->sk->dirs_prep.p, ctx->pc[0], ctx->pc[1], ctx->pc[2],
                                 1.0 / kC, pstream(ctx)));
  else
    HIPCHK(ctx, launch_pack((const double*)ctx->sk->pb.p, ctx->sk->packed.p, pl.f32, ctx->nsrc, pl.nsrc_pad, ctx->nchan, pl.ct,
                            pl.ntiles, ctx->dirs_p, scale_comp, g64 ? 0 : 1, ctx->stream));
  // Packed fp32 taper on a sky whose sources come in a few runs of one size each (every HEALPix sky; point sources + diffuse): the
  // split form, run by run (skyvis_kernels.hip: TGROUP 2 / 3) -- size-0 runs take the plain (no-taper) bodies.
  const bool split = scale_comp < 0 && taper_split_plan(ctx, pl, p);
  if (prep) { int rcj = join_prep(ctx); if (rcj) return rcj; }      // rows, directions, flags are ready: the sums run on the compute stream
  // Run by run under a SOURCE SPLIT (baseline shards of a mixed sky: point sources + a diffuse map): every run is cut into nsplit pieces
  // of its own and writes its own set of nsplit partial cubes; k_reduce_partials then sums nruns x nsplit of them, in fixed order.
  // (Before, a split sky of several runs fell back to ONE launch of the unsplit-form kernels over the whole sky: 510 ms against
  // 453 ideal for one rank's half of config 3 + diffuse, profiles/r04_shard_balance.json.)
  const size_t nruns_sky = ctx->kappa_runs.size();
  const bool by_run = (split || g64) && nruns_sky > 1 && nruns_sky <= (size_t)kMaxRunSets;
  const int nsets = (pl.nsplit > 1 && by_run) ? (int)nruns_sky : 1;
  auto run_per_split = [&](const prisim_ctx::KappaRun& run) {
    return round_up(((run.hi - run.lo) + pl.nsplit - 1) / pl.nsplit, pl.chunk);
  };
  int64_t max_per = pl.src_per_split;
  if (nsets > 1) {
    max_per = 0;
    for (const auto& run : ctx->kappa_runs) max_per = std::max(max_per, run_per_split(run));
  }
  p.out = pl.nsplit > 1 ? (double*)ctx->partial.p : dst;
  // fp32 kernels whose splits each flush exactly once store their partial sums as complex64: half the partial traffic
  const bool part_f32 = pl.nsplit > 1 && pl.f32 && pl.kernel == PRISIM_KERNEL_RECURRENCE && max_per <= (int64_t)p.flush_src;
  p.out_f32 = part_f32 ? 1 : 0;
  const size_t set_reals = (size_t)pl.nsplit * (size_t)ctx->nbl * (size_t)ctx->nchan * 2;       // reals of one run's partial cubes
  auto set_out = [&](size_t r) -> double* {
    if (nsets == 1) return p.out;
    return part_f32 ? (double*)((float*)ctx->partial.p + r * set_reals) : (double*)ctx->partial.p + r * set_reals;
  };
  // taper culling: per baseline group the first source it still has to sum (tables staged by set_sky_*); a launch over the whole sky
  // can only skip the leading sources of the FIRST run
  const int cpr = pl.f32 ? 1 : 0;
  const bool cull = (pl.pk || g64) && ctx->taper && ctx->cull_any[cpr] && ctx->sk->cull_first.p && ctx->cull_nruns > 0 && !ctx->kappa_runs.empty() &&
                    (size_t)pl.nbgroups == ctx->grp_maxlen.size();             // (the packed fp32 kernels and the grouped fp64 kernel)
  auto cull_table = [&](size_t r) {
    return (const int32_t*)ctx->sk->cull_first.p + ((size_t)cpr * ctx->cull_nruns + (size_t)ctx->kappa_runs[r].tab_row) * (size_t)pl.nbgroups;
  };
  if (cull && !split) p.src_first = cull_table(0);
  ctx->timing.last_culled_fraction = cull ? ctx->cull_frac[cpr] : 0.0;
  ctx->cull_frac_pending = (cull && ctx->cat.cur >= 0) ? cpr + 1 : 0;      // catalogue path: the device's count is read once the events are in
  if (timed) HIPCHK(ctx, hipEventRecord(ctx->ev_k0[ctx->ring_head], ctx->stream));
  if (split) {
    int launches = 0;
    for (size_t r = 0; r < ctx->kappa_runs.size(); ++r) {
      const prisim_ctx::KappaRun& run = ctx->kappa_runs[r];
      SkyvisParams q = p;
      q.src_lo = run.lo; q.src_hi = run.hi;
      if (pl.nsplit == 1) q.src_per_split = round_up(run.hi - run.lo, pl.chunk);
      else if (nsets > 1) q.src_per_split = run_per_split(run);                        // (a split sky of ONE run: the plan's pieces stand)
      q.accumulate = (launches > 0 && nsets == 1) ? 1 : 0;
      q.out = set_out(r);
      if (run.kappa > 0.0) {
        q.kappa0 = run.kappa;
        if (cull) q.src_first = cull_table(r);
        q.split_flags = (const int32_t*)ctx->sk->split_flags.p + r * (size_t)pl.nbgroups;
        HIPCHK(ctx, launch_skyvis_rec_f32pk_split(q, pl.ct, ctx->stream));
      } else {
        q.taper = 0;                                   // point sources: w = 1 (:6270 sigma = inf), the lifting / plain bodies
        HIPCHK(ctx, launch_skyvis_rec_f32pk(q, pl.ct, ctx->stream));
      }
      ++launches;
    }
    ctx->timing.last_taper_split = launches;
  } else if (pl.pk) {
    HIPCHK(ctx, launch_skyvis_rec_f32pk(p, pl.ct, ctx->stream));
  } else if (g64) {
    // run by run when the sky comes in runs of one source size (so that every run's leading sources can be culled); with a source
    // split (partial cubes, written once per split) only a sky that is one run -- otherwise one launch over the whole sky
    const size_t nruns = ctx->kappa_runs.size();
    if (by_run) {
      for (size_t r = 0; r < nruns; ++r) {
        const prisim_ctx::KappaRun& run = ctx->kappa_runs[r];
        SkyvisParams q = p;
        q.src_lo = run.lo; q.src_hi = run.hi;
        q.src_per_split = pl.nsplit == 1 ? round_up(run.hi - run.lo, pl.chunk) : run_per_split(run);
        q.accumulate = (r > 0 && nsets == 1) ? 1 : 0;
        q.out = set_out(r);
        q.src_first = cull ? cull_table(r) : nullptr;
        if (run.kappa > 0.0) {
          if (wave_items(ctx, pl)) {
            q.wave_nbw = (int32_t)((ctx->nbl + 63) / 64);
            q.wave_nsplit = pl.nsplit;
            q.nsplit = 1;
            q.nbgroups = (q.wave_nbw * pl.nsplit + kBlockThreads / 64 - 1) / (kBlockThreads / 64);
          }
          HIPCHK(ctx, launch_skyvis_taper_f64(q, pl.ct, ctx->stream));
        } else {
          // point sources (w = 1, :6270 sigma = inf): the fp64 kernel without the taper (6.2 instead of 9.8 instructions per term); its
          // rows are (up, down) pairs: this run's rows are re-packed in that layout
          HIPCHK(ctx, launch_pack((const double*)ctx->sk->pb.p, ctx->sk->packed.p, false, ctx->nsrc, pl.nsrc_pad, ctx->nchan, pl.ct, pl.ntiles,
                                  ctx->dirs_p, scale_comp, 1, ctx->stream, run.lo, run.hi));
          q.taper = 0;
          q.src_first = nullptr;
          HIPCHK(ctx, launch_skyvis_rec(q, false, pl.ct, ctx->stream));
        }
      }
    } else {
      SkyvisParams q = p;
      if (wave_items(ctx, pl)) {
        // an array of at most 256 baselines with split sources: the unit of work is a wavefront (baseline wave, split), four per block
        q.wave_nbw = (int32_t)((ctx->nbl + 63) / 64);
        q.wave_nsplit = pl.nsplit;
        q.nsplit = 1;
        q.nbgroups = (q.wave_nbw * pl.nsplit + kBlockThreads / 64 - 1) / (kBlockThreads / 64);
      }
      HIPCHK(ctx, launch_skyvis_taper_f64(q, pl.ct, ctx->stream));
    }
  } else {
    HIPCHK(ctx, launch_skyvis_rec(p, pl.f32, pl.ct, ctx->stream));
  }
  if (timed) HIPCHK(ctx, hipEventRecord(ctx->ev_k1[ctx->ring_head], ctx->stream));
  if (pl.nsplit > 1)
    HIPCHK(ctx, launch_reduce_partials(ctx->partial.p, part_f32, dst, ctx->nbl * ctx->nchan * 2, pl.nsplit * nsets, ctx->stream));
  return PRISIM_OK;
}

// V + the three baseline-gradient sums of one snapshot in one pass: dst [nbl][nchan], gdst [3][nbl][nchan] complex128.
// fp64: k_skyvis_grad_f64 (MFMA 4x4x4, groups of 64 baselines); fp32: the GRAD bodies of the packed kernel (16-channel tiles).
static int run_grad_pass(prisim_ctx* ctx, const Plan& pl, double* dst, double* gdst) {
  SkyvisParams p{};
  fill_params(ctx, pl, p);
  p.nsplit = 1; p.src_per_split = pl.nsrc_pad;
  p.out = dst;
  p.grad_out = gdst;
  p.out_f32 = 0;
  if (pl.f32 && !ctx->taper) {
    // rows pre-multiplied by the gradient coefficients (1, l, m, n): 64 floats per (source, 16-channel tile)
    int rc2;
    if ((rc2 = ensure(ctx, ctx->sk->packed, (size_t)pl.ntiles * pl.nsrc_pad * 64 * sizeof(float)))) return rc2;
    p.pb_packed = ctx->sk->packed.p;
    HIPCHK(ctx, launch_pack_grad((const double*)ctx->sk->pb.p, (float*)ctx->sk->packed.p, ctx->nsrc, pl.nsrc_pad, ctx->nchan, pl.ntiles,
                                 ctx->dirs_p, pstream(ctx)));
  } else {
    // (the grouped fp64 taper kernel takes its rows in natural channel order, everything else (up, down) pairs)
    HIPCHK(ctx, launch_pack((const double*)ctx->sk->pb.p, ctx->sk->packed.p, pl.f32, ctx->nsrc, pl.nsrc_pad, ctx->nchan, pl.ct, pl.ntiles,
                            ctx->dirs_p, -1, (!pl.f32 && ctx->taper && pl.ct == 32) ? 0 : 1, pstream(ctx)));
  }
  { int rcj = join_prep(ctx); if (rcj) return rcj; }
  HIPCHK(ctx, hipEventRecord(ctx->ev_k0[ctx->ring_head], ctx->stream));
  if (pl.f32) {
    p.dirs_c32 = (const float*)ctx->sk->dirs_c32.p;
    HIPCHK(ctx, launch_skyvis_grad_f32(p, ctx->stream));
  } else {
    p.nbgroups = (int)((ctx->nbl + 63) / 64);      // the MFMA kernel's blocks own 64 baselines; lift flags stay per 256 (it reads [group >> 2])
    HIPCHK(ctx, launch_skyvis_grad_f64(p, pl.ct, ctx->stream));
  }
  HIPCHK(ctx, hipEventRecord(ctx->ev_k1[ctx->ring_head], ctx->stream));
  return PRISIM_OK;
}

int prisim_hip_compute(prisim_ctx* ctx, int precision, int kernel, int want_grad, int64_t slot) {
  return guarded(ctx, [&]() -> int {
  if (!ctx) return PRISIM_EINVAL;
  if (!ctx->array_set || !ctx->sky_set) return fail(ctx, PRISIM_ESTATE, "set_array and set_sky must precede compute");
  if (precision != PRISIM_FP64 && precision != PRISIM_FP32) return fail(ctx, PRISIM_EINVAL, "unknown precision");
  if (kernel < PRISIM_KERNEL_AUTO || kernel > PRISIM_KERNEL_DIRECT) return fail(ctx, PRISIM_EINVAL, "unknown kernel id");
  if (slot < 0 || slot >= ctx->nt_max) return fail(ctx, PRISIM_EINVAL, "slot out of range");
  if (kernel == PRISIM_KERNEL_RECURRENCE && !ctx->uniform)
    return fail(ctx, PRISIM_EINVAL, "recurrence kernel needs a uniform channel grid");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  harvest_timing(ctx, /*max_wait=*/ctx->ring_pending >= prisim_ctx::kTimingRing ? 1 : 0);   // a full ring waits for its oldest entry only
  const size_t slot_elems = (size_t)ctx->nbl * ctx->nchan * 2;
  double* dst = (double*)ctx->cube.p + (size_t)slot * slot_elems;
  int rc;
  if (want_grad) {
    const size_t gbytes = (size_t)ctx->nt_max * 3 * slot_elems * sizeof(double);
    if (!ctx->grad.p) {
      if ((rc = ensure(ctx, ctx->grad, gbytes))) return rc;
      HIPCHK(ctx, hipMemsetAsync(ctx->grad.p, 0, gbytes, ctx->stream));
    }
  }
  if (ctx->nsrc == 0) {   // interferometry.py:6378-6382: visibilities stay zero
    HIPCHK(ctx, hipMemsetAsync(dst, 0, slot_elems * sizeof(double), ctx->stream));
    if (want_grad)
      HIPCHK(ctx, hipMemsetAsync((double*)ctx->grad.p + (size_t)slot * 3 * slot_elems, 0, 3 * slot_elems * sizeof(double), ctx->stream));
    ctx->timing.last_terms = 0;
    catalog_after_compute(ctx);      // (an empty region of interest still releases the catalogue's buffer set in stream order)
    return PRISIM_OK;
  }
  Plan pl = make_plan(ctx, precision, kernel);
  // Visibility + baseline gradient on a uniform channel grid: ONE fused pass -- fp64: the MFMA kernel k_skyvis_grad_f64 (2.4 x a plain
  // fp64 pass instead of 4 x); fp32: the GRAD bodies of the packed kernel on 16-channel tiles (13 packed instructions per pair of terms
  // against 4 passes x 5).  PRISIM_HIP_FUSED_GRAD=0: the four-pass form (the A/B baseline; non-uniform grids use the direct kernel).
  bool fused_grad = want_grad && pl.kernel == PRISIM_KERNEL_RECURRENCE;
  if (const char* env = getenv("PRISIM_HIP_FUSED_GRAD")) fused_grad = fused_grad && atoi(env) != 0;
  if (fused_grad) {
    const bool f32 = pl.f32;
    pl = make_plan(ctx, f32 ? PRISIM_FP32 : PRISIM_FP64, PRISIM_KERNEL_RECURRENCE);
    // fp32: 4 x 2 x 16 packed accumulators = 128 VGPRs; fp64: the taper's per-lane recurrence state does not fit beside 128 at 32
    // fp64 with the taper: the grouped single-chain kernel on 32-channel tiles (k_skyvis_grad_taper_f64); PRISIM_HIP_GRAD_TAPER_GROUP=0 =
    // round 3's exact form on 16-channel tiles (the A/B baseline)
    pl.ct = f32 ? 16 : ((ctx->taper && !grad_taper_grouped()) ? 16 : 32);
    pl.pk = f32;
    pl.ntiles = (int)((ctx->nchan + pl.ct - 1) / pl.ct);
    pl.nsplit = 1;
    pl.nsrc_pad = round_up(pl.nsrc_pad, 4);       // the fp64 kernel walks the sources four at a time (zero rows past nsrc)
    pl.src_per_split = pl.nsrc_pad;
  }
  ctx->timing.last_lift_groups = 0;
  ctx->timing.last_taper_group = 0;
  ctx->timing.last_taper_split = 0;
  ctx->timing.last_split_uncorrected_groups = 0;
  ctx->timing.last_culled_fraction = 0.0;
  ctx->timing.last_batch_snapshots = 1;
  if (pl.kernel == PRISIM_KERNEL_RECURRENCE) {
    const size_t pbytes = (size_t)pl.ntiles * pl.nsrc_pad * pl.ct * (pl.f32 ? 4 : 8);
    if ((rc = ensure(ctx, ctx->sk->packed, pbytes))) return rc;
    if ((rc = ensure(ctx, ctx->sk->dirs_prep, (size_t)pl.nsrc_pad * 4 * sizeof(double)))) return rc;
    // (a sky of several runs of one source size writes one set of partial cubes per run: run_pass)
    const size_t part_sets = (ctx->taper && ctx->kappa_runs.size() > 1 && ctx->kappa_runs.size() <= (size_t)kMaxRunSets) ? ctx->kappa_runs.size() : 1;
    if (pl.nsplit > 1 && (rc = ensure(ctx, ctx->partial, part_sets * (size_t)pl.nsplit * slot_elems * sizeof(double)))) return rc;
    {
      // lifting rotation is used for a baseline group only when |step phase| <= 1/8 cycle (fp32; 1/4 cycle in fp64, where the
      // angle error alpha*eps is irrelevant and only tan(alpha/2) must stay bounded) is guaranteed for every source:
      // |theta| = |b . (s - s_pc)| |df| / c <= max|b| * max_s|s - s_pc| * |df| / c
      const double k = ctx->dmax * std::fabs(ctx->df) / kC;
      const double lift_limit = (pl.f32 ? 0.125 : 0.25) * (1.0 - 1e-9);
      if (k != ctx->sk->lift_key_k || (int)pl.f32 != ctx->sk->lift_key_f32 || ctx->sk->lift_groups != pl.nbgroups) {
        // formed on the device from the groups' longest baselines (resident since set_array): max|s - s_pc| changes with every snapshot
        // of a drift scan, and a host-side table would need a stream synchronisation before it could be rewritten
        if ((rc = ensure(ctx, ctx->sk->lift_flags, (size_t)pl.nbgroups * sizeof(int32_t)))) return rc;
        HIPCHK(ctx, launch_lift_flags((const double*)ctx->grp_hz.p + 2 * ctx->grp_maxlen.size(), k, lift_limit, (int32_t*)ctx->sk->lift_flags.p,
                                      pl.nbgroups, pstream(ctx)));
        ctx->sk->lift_key_k = k;
        ctx->sk->lift_key_f32 = (int)pl.f32;
        ctx->sk->lift_groups = pl.nbgroups;
      }
      int nlift = 0;
      for (int g = 0; g < pl.nbgroups; ++g) nlift += (ctx->grp_maxlen[(size_t)g] * k <= lift_limit) ? 1 : 0;
      // the packed taper kernel folds the amplitude into the phasor (a scaled rotation: no lifting there, the flags only select its
      // re-anchored body); every other kernel lifts the flagged groups
      ctx->timing.last_lift_groups = (ctx->taper && pl.pk) ? 0 : nlift;
    }
    if (pl.f32 && ctx->taper) {
      if ((rc = ensure(ctx, ctx->fsq_pairs, (size_t)pl.ntiles * pl.ct * sizeof(float)))) return rc;
      if (ctx->fsq_pairs_ct != pl.ct || ctx->fsq_pairs_ntiles != pl.ntiles) {      // (a function of the tiling only: once per plan)
        HIPCHK(ctx, launch_fsq_pairs((const float*)ctx->fsq.p, (float*)ctx->fsq_pairs.p, pl.ct, pl.ntiles, ctx->stream));
        ctx->fsq_pairs_ct = pl.ct; ctx->fsq_pairs_ntiles = pl.ntiles;
      }
    }
  }
  HIPCHK(ctx, hipEventRecord(ctx->ev_c0[ctx->ring_head], ctx->stream));
  if (pl.kernel == PRISIM_KERNEL_RECURRENCE && fused_grad) {
    float* c32 = nullptr;
    if (pl.f32) {
      if ((rc = ensure(ctx, ctx->sk->dirs_c32, (size_t)pl.nsrc_pad * 8 * sizeof(float)))) return rc;
      c32 = (float*)ctx->sk->dirs_c32.p;
    }
    HIPCHK(ctx, launch_prep_dirs(ctx->dirs_p, (double*)ctx->sk->dirs_prep.p, c32, ctx->nsrc, pl.nsrc_pad, ctx->pc[0],
                                 ctx->pc[1], ctx->pc[2], 1.0 / kC, pstream(ctx)));
  }
  if (fused_grad) {
    if ((rc = run_grad_pass(ctx, pl, dst, (double*)ctx->grad.p + (size_t)slot * 3 * slot_elems))) return rc;
  } else if ((rc = run_pass(ctx, pl, dst, -1, true, /*prep=*/true))) {      // (prepares the directions with its first launch)
    return rc;
  }
  if (want_grad && !fused_grad) {
    for (int comp = 0; comp < 3; ++comp) {
      double* gdst = (double*)ctx->grad.p + ((size_t)slot * 3 + comp) * slot_elems;
      if ((rc = run_pass(ctx, pl, gdst, comp, false))) return rc;
    }
  }
  HIPCHK(ctx, hipEventRecord(ctx->ev_c1[ctx->ring_head], ctx->stream));
  if (ctx->prep_async) {
    HIPCHK(ctx, hipEventRecord(ctx->sk->ev_sum, ctx->stream));
    ctx->sk->sum_recorded = true;
  }
  catalog_after_compute(ctx);
  ctx->ring_head = (ctx->ring_head + 1) % prisim_ctx::kTimingRing;
  ctx->ring_pending += 1;
  ctx->timing.last_terms = ctx->nbl * ctx->nchan * ctx->nsrc;
  ctx->timing.last_kernel_id = pl.kernel;
  ctx->timing.last_chan_tile = pl.kernel == PRISIM_KERNEL_RECURRENCE ? pl.ct : 1;
  ctx->timing.last_nsplit = pl.kernel == PRISIM_KERNEL_RECURRENCE ? pl.nsplit : 1;
  return PRISIM_OK;
  });
}

static void to_c64(const double* in, float* out, size_t n2) {
  for (size_t i = 0; i < n2; ++i) out[i] = (float)in[i];
}

int prisim_hip_get_vis(prisim_ctx* ctx, int64_t slot, void* vis, void* grad, int out_is_c64) {
  return guarded(ctx, [&]() -> int {
  if (!ctx) return PRISIM_EINVAL;
  if (!ctx->array_set) return fail(ctx, PRISIM_ESTATE, "set_array has not been called");
  if (slot < 0 || slot >= ctx->nt_max) return fail(ctx, PRISIM_EINVAL, "slot out of range");
  if (!vis) return fail(ctx, PRISIM_EINVAL, "vis is NULL");
  if (grad && !ctx->grad.p) return fail(ctx, PRISIM_ESTATE, "no gradient has been computed");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  harvest_timing(ctx);
  const size_t slot_elems = (size_t)ctx->nbl * ctx->nchan * 2;
  const double* src = (const double*)ctx->cube.p + (size_t)slot * slot_elems;
  const double* gsrc = grad ? (const double*)ctx->grad.p + (size_t)slot * 3 * slot_elems : nullptr;
  if (!out_is_c64) {
    HIPCHK(ctx, hipMemcpy(vis, src, slot_elems * sizeof(double), hipMemcpyDeviceToHost));
    if (grad) HIPCHK(ctx, hipMemcpy(grad, gsrc, 3 * slot_elems * sizeof(double), hipMemcpyDeviceToHost));
  } else {
    std::vector<double> tmp;
    try { tmp.resize(grad ? 3 * slot_elems : slot_elems); } catch (...) { return fail(ctx, PRISIM_ENOMEM, "host staging"); }
    HIPCHK(ctx, hipMemcpy(tmp.data(), src, slot_elems * sizeof(double), hipMemcpyDeviceToHost));
    to_c64(tmp.data(), (float*)vis, slot_elems);
    if (grad) {
      HIPCHK(ctx, hipMemcpy(tmp.data(), gsrc, 3 * slot_elems * sizeof(double), hipMemcpyDeviceToHost));
      to_c64(tmp.data(), (float*)grad, 3 * slot_elems);
    }
  }
  return PRISIM_OK;
  });
}

int prisim_hip_set_vis(prisim_ctx* ctx, int64_t slot, const double* vis) {
  return guarded(ctx, [&]() -> int {
  if (!ctx) return PRISIM_EINVAL;
  if (!ctx->array_set) return fail(ctx, PRISIM_ESTATE, "set_array has not been called");
  if (slot < 0 || slot >= ctx->nt_max) return fail(ctx, PRISIM_EINVAL, "slot out of range");
  if (!vis) return fail(ctx, PRISIM_EINVAL, "vis is NULL");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  const size_t slot_elems = (size_t)ctx->nbl * ctx->nchan * 2;
  HIPCHK(ctx, hipMemcpyAsync((double*)ctx->cube.p + (size_t)slot * slot_elems, vis, slot_elems * sizeof(double),
                             hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  return PRISIM_OK;
  });
}

int prisim_hip_skyvis(prisim_ctx* ctx, const prisim_sky* sky, int precision, int kernel, void* vis, void* grad,
                      int out_is_c64) {
  return guarded(ctx, [&]() -> int {
  int rc = prisim_hip_set_sky(ctx, sky);
  if (rc) return rc;
  if ((rc = prisim_hip_compute(ctx, precision, kernel, grad != nullptr, 0))) return rc;
  return prisim_hip_get_vis(ctx, 0, vis, grad, out_is_c64);
  });
}

int prisim_hip_sync(prisim_ctx* ctx) {
  return guarded(ctx, [&]() -> int {
  if (!ctx) return PRISIM_EINVAL;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  if (ctx->prep_stream) HIPCHK(ctx, hipStreamSynchronize(ctx->prep_stream));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  if (ctx->comm_stream && ctx->comm_pending) { HIPCHK(ctx, hipStreamSynchronize(ctx->comm_stream)); ctx->comm_pending = false; }
  if (ctx->copy_stream && ctx->copy_pending) { HIPCHK(ctx, hipStreamSynchronize(ctx->copy_stream)); ctx->copy_pending = false; }
  harvest_timing(ctx);
  harvest_comm(ctx, true);
  return PRISIM_OK;
  });
}

int prisim_hip_get_timing(prisim_ctx* ctx, prisim_timing* out, int reset) {
  return guarded(ctx, [&]() -> int {
  if (!ctx) return PRISIM_EINVAL;
  if (!out) return fail(ctx, PRISIM_EINVAL, "out is NULL");
  if (ctx->ring_pending > 0) {
    HIPCHK(ctx, hipSetDevice(ctx->device));
    harvest_timing(ctx);
  }
  if (ctx->ev_d1 && ctx->timing.last_delay_ms < 0.0) {
    HIPCHK(ctx, hipSetDevice(ctx->device));
    float ms = 0.f;
    if (hipEventSynchronize(ctx->ev_d1) == hipSuccess && hipEventElapsedTime(&ms, ctx->ev_d0, ctx->ev_d1) == hipSuccess)
      ctx->timing.last_delay_ms = ms;
  }
  if (ctx->cull_frac_pending && ctx->cat.culled_host && ctx->ring_pending == 0) {
    // (catalogue path: the cull table was built on the device; its count copy was queued before the compute whose events are now in)
    const int pr = ctx->cull_frac_pending - 1;
    const double pairs = (double)ctx->nsrc * (double)ctx->nbl;
    ctx->cull_frac[pr] = pairs > 0.0 ? (double)ctx->cat.culled_host[pr] / pairs : 0.0;
    ctx->timing.last_culled_fraction = ctx->cull_frac[pr];
    ctx->cull_frac_pending = 0;
  }
  if (ctx->timing.last_taper_split > 0 && ctx->h_split_count && ctx->ring_pending == 0) {
    // (every compute's events have completed: the count copy queued before them has landed)
    int n = 0;
    for (int r = 0; r < ctx->split_count_runs && r < 8; ++r) n += ctx->h_split_count[r];
    ctx->timing.last_split_uncorrected_groups = n;
  }
  *out = ctx->timing;
  if (reset) { ctx->timing.sum_kernel_ms = 0.0; ctx->timing.n_kernel = 0; }
  return PRISIM_OK;
  });
}

int prisim_hip_device_info(prisim_ctx* ctx, int* cu_count, int* clock_khz, char name[64]) {
  return guarded(ctx, [&]() -> int {
  if (!ctx) return PRISIM_EINVAL;
  if (cu_count) *cu_count = ctx->cu_count;
  if (clock_khz) *clock_khz = ctx->clock_khz;
  if (name) { memcpy(name, ctx->devname, 64); name[63] = 0; }
  return PRISIM_OK;
  });
}

int prisim_hip_set_tuning(prisim_ctx* ctx, int chan_tile, int src_chunk, int nsplit) {
  return guarded(ctx, [&]() -> int {
  if (!ctx) return PRISIM_EINVAL;
  if (chan_tile != 0 && chan_tile != 8 && chan_tile != 16 && chan_tile != 32 && chan_tile != 64)
    return fail(ctx, PRISIM_EINVAL, "chan_tile must be 0, 8, 16, 32 or 64");
  if (src_chunk < 0 || src_chunk > 256 || nsplit < 0 || nsplit > 4096)
    return fail(ctx, PRISIM_EINVAL, "src_chunk must be in [0,256], nsplit in [0,4096]");
  ctx->tune_ct = chan_tile; ctx->tune_chunk = src_chunk; ctx->tune_nsplit = nsplit;
  return PRISIM_OK;
  });
}

// ---- delay transform ------------------------------------------------------------------------

namespace {

struct DelayGeom {
  int64_t npad, nfft, nout;
  double factor;
  bool fused;          // integer 1 + pad and a power-of-two channel count: one LDS kernel, no padding, no rocFFT
};

DelayGeom delay_geom(const prisim_ctx* ctx, double pad) {
  DelayGeom g{};
  const int64_t nchan = ctx->nchan;
  g.npad = (int64_t)((double)nchan * pad);                                // :8123
  g.nfft = nchan + g.npad;
  g.factor = 1.0 + pad;                                                   // :8131
  g.nout = (int64_t)std::ceil((double)g.nfft / g.factor - 1e-12);         // len(arange(0, nfft, factor))
  const double fr = std::round(g.factor);
  g.fused = delay_fft_supported(nchan) && std::fabs(g.factor - fr) == 0.0 && fr >= 1.0 && g.nfft == nchan * (int64_t)fr && g.nout == nchan;
  if (const char* env = getenv("PRISIM_HIP_DT_FUSED")) g.fused = g.fused && atoi(env) != 0;      // A/B hook
  return g;
}

// Upload the window [nbl][nchan] (or none) and make sure the fused kernel's twiddle table / the rocFFT plan exist.
int delay_prepare(prisim_ctx* ctx, const DelayGeom& g, const double* bpwts, int64_t wts_rows, int64_t nrows_batch) {
  int rc;
  const int64_t nchan = ctx->nchan, nbl = ctx->nbl;
  if (bpwts) {
    if (wts_rows != 1 && wts_rows != nbl) return fail(ctx, PRISIM_EINVAL, "wts_rows must be 1 or nbl");
    if ((rc = ensure(ctx, ctx->dt_wts, (size_t)wts_rows * nchan * sizeof(double)))) return rc;
    HIPCHK(ctx, hipMemcpyAsync(ctx->dt_wts.p, bpwts, (size_t)wts_rows * nchan * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));     // bpwts is caller-owned
  }
  if (g.fused) {
    if (ctx->dt_tw_n != nchan) {
      std::vector<double> tw((size_t)nchan);            // nchan / 2 complex values W_N^m = exp(+2 pi i m / N)
      for (int64_t m = 0; m < nchan / 2; ++m) {
        const double a = 2.0 * M_PI * (double)m / (double)nchan;
        tw[(size_t)2 * m] = std::cos(a);
        tw[(size_t)2 * m + 1] = std::sin(a);
      }
      if ((rc = ensure(ctx, ctx->dt_tw, tw.size() * sizeof(double)))) return rc;
      HIPCHK(ctx, hipMemcpy(ctx->dt_tw.p, tw.data(), tw.size() * sizeof(double), hipMemcpyHostToDevice));
      ctx->dt_tw_n = nchan;
    }
    return PRISIM_OK;
  }
  std::string lerr;
  if (!load_rocfft(lerr)) return fail(ctx, PRISIM_ELIB, lerr);
  RocfftApi& F = g_rocfft;
  if (!F.setup_done) {
    if (F.setup() != rocfft_status_success) return fail(ctx, PRISIM_ELIB, "rocfft_setup failed");
    F.setup_done = true;
  }
  if ((rc = ensure(ctx, ctx->fft_buf, (size_t)nrows_batch * g.nfft * 2 * sizeof(double)))) return rc;
  return PRISIM_OK;
}

// Transform `nrows` = ntc * nbl rows starting at snapshot slot t0 into the device buffers d_out ([nrows][nout] complex128) and/or
// d_pow ([nrows][nout] float64).  Asynchronous on the context stream.
int delay_batch(prisim_ctx* ctx, const DelayGeom& g, int64_t t0, int64_t nrows, bool have_wts, int64_t wts_rows, double* d_out, double* d_pow,
                double power_scale) {
  const int64_t nchan = ctx->nchan, nbl = ctx->nbl;
  const size_t slot_elems = (size_t)nbl * nchan * 2;
  const double* src = (const double*)ctx->cube.p + (size_t)t0 * slot_elems;
  const double* wts = have_wts ? (const double*)ctx->dt_wts.p : nullptr;
  // rocFFT's inverse is unnormalised: sum_n x[n] e^{+2 pi i k n / N'}.  The reference forms ifft(x) * N' * df (:8125) = that sum times df.
  const double scale = ctx->df;
  if (g.fused) {
    HIPCHK(ctx, launch_delay_fft(src, wts, wts_rows, (const double*)ctx->dt_tw.p, d_out, d_pow, nrows, nbl, nchan, scale, power_scale,
                                 ctx->cu_count, ctx->stream));
    return PRISIM_OK;
  }
  RocfftApi& F = g_rocfft;
  int rc;
  if (!ctx->fft_plan || ctx->fft_len != (size_t)g.nfft || ctx->fft_batch != (size_t)nrows) {
    if (ctx->fft_plan) { F.plan_destroy(ctx->fft_plan); ctx->fft_plan = nullptr; }
    size_t len = (size_t)g.nfft;
    if (F.plan_create(&ctx->fft_plan, rocfft_placement_inplace, rocfft_transform_type_complex_inverse,
                      rocfft_precision_double, 1, &len, (size_t)nrows, nullptr) != rocfft_status_success) {
      ctx->fft_plan = nullptr;
      return fail(ctx, PRISIM_ELIB, "rocfft_plan_create failed");
    }
    ctx->fft_len = (size_t)g.nfft; ctx->fft_batch = (size_t)nrows;
    if (!ctx->fft_info && F.execution_info_create(&ctx->fft_info) != rocfft_status_success)
      return fail(ctx, PRISIM_ELIB, "rocfft_execution_info_create failed");
    if (F.execution_info_set_stream(ctx->fft_info, ctx->stream) != rocfft_status_success)
      return fail(ctx, PRISIM_ELIB, "rocfft_execution_info_set_stream failed");
    size_t wbytes = 0;
    F.plan_get_work_buffer_size(ctx->fft_plan, &wbytes);
    if (wbytes) {
      if ((rc = ensure(ctx, ctx->fft_work, wbytes))) return rc;
      if (F.execution_info_set_work_buffer(ctx->fft_info, ctx->fft_work.p, wbytes) != rocfft_status_success)
        return fail(ctx, PRISIM_ELIB, "rocfft_execution_info_set_work_buffer failed");
    }
  }
  HIPCHK(ctx, launch_dt_prepare(src, wts, wts_rows, (double*)ctx->fft_buf.p, nrows, nbl, nchan, g.nfft, ctx->stream));
  void* bufs[1] = {ctx->fft_buf.p};
  if (F.execute(ctx->fft_plan, bufs, nullptr, ctx->fft_info) != rocfft_status_success)
    return fail(ctx, PRISIM_ELIB, "rocfft_execute failed");
  HIPCHK(ctx, launch_dt_finish((const double*)ctx->fft_buf.p, d_out, d_pow, nrows, g.nfft, g.nout, g.factor, scale, power_scale, ctx->stream));
  return PRISIM_OK;
}

// snapshots per batch of the rocFFT pipeline so that its padded work buffer stays <= 4 GiB whatever nt is
// (config 5: 120 x 61075 rows of 2048 would be 240 GB at once); the fused kernel needs no work buffer
int64_t delay_batch_snapshots(const prisim_ctx* ctx, const DelayGeom& g, int64_t nt) {
  if (g.fused && !getenv("PRISIM_HIP_DT_BATCH_BYTES")) return nt;
  const int64_t row_bytes = g.nfft * 2 * (int64_t)sizeof(double);
  int64_t budget = (int64_t)4 << 30;
  if (const char* env = getenv("PRISIM_HIP_DT_BATCH_BYTES")) {   // test hook: force several batches on small cubes
    const long long v = atoll(env);
    if (v > 0) budget = v;
  }
  int64_t nt_b = std::max<int64_t>(1, budget / (ctx->nbl * row_bytes));
  return std::min(nt_b, nt);
}

void delay_lags(const prisim_ctx* ctx, double* lags_out) {
  // DSP.spectral_axis(nchan, delx=df, shift=True) (:8114) == fftshift(fftfreq(nchan, df))
  for (int64_t i = 0; i < ctx->nchan; ++i) {
    const int64_t k = i - ctx->nchan / 2;
    lags_out[i] = (double)k / ((double)ctx->nchan * ctx->df);
  }
}

int delay_check(prisim_ctx* ctx, int64_t nt, double& pad) {
  if (!ctx->array_set) return fail(ctx, PRISIM_ESTATE, "set_array has not been called");
  if (nt <= 0 || nt > ctx->nt_max) return fail(ctx, PRISIM_EINVAL, "nt out of range");
  if (!(pad >= 0.0) || !std::isfinite(pad)) pad = 0.0;   // interferometry.py:8091-8092
  if (!ctx->uniform || ctx->nchan < 2) return fail(ctx, PRISIM_EINVAL, "delay transform needs >= 2 uniformly spaced channels");
  return PRISIM_OK;
}

}  // namespace

int prisim_hip_delay_transform(prisim_ctx* ctx, int64_t nt, const double* bpwts, double pad, double* out,
                               double* lags_out, double* out_power, double power_scale) {
  return guarded(ctx, [&]() -> int {
  if (!ctx) return PRISIM_EINVAL;
  int rc;
  if ((rc = delay_check(ctx, nt, pad))) return rc;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  const DelayGeom g = delay_geom(ctx, pad);
  const int64_t nbl = ctx->nbl, nout = g.nout;
  // the host-output form always goes batch by batch through two staging buffers
  int64_t nt_b = delay_batch_snapshots(ctx, g, nt);
  if (g.fused) nt_b = std::min<int64_t>(nt, std::max<int64_t>(1, ((int64_t)2 << 30) / (nbl * nout * 16)));
  if ((rc = delay_prepare(ctx, g, bpwts, nbl, nt_b * nbl))) return rc;
  if (out && (rc = ensure(ctx, ctx->dt_out, (size_t)(nt_b * nbl) * nout * 2 * sizeof(double)))) return rc;
  if (out_power && (rc = ensure(ctx, ctx->dt_pow, (size_t)(nt_b * nbl) * nout * sizeof(double)))) return rc;
  for (int64_t t0 = 0; t0 < nt; t0 += nt_b) {
    const int64_t ntc = std::min(nt_b, nt - t0);
    const int64_t nrows = ntc * nbl;
    if ((rc = delay_batch(ctx, g, t0, nrows, bpwts != nullptr, nbl, out ? (double*)ctx->dt_out.p : nullptr,
                          out_power ? (double*)ctx->dt_pow.p : nullptr, power_scale)))
      return rc;
    if (out)
      HIPCHK(ctx, hipMemcpyAsync(out + (size_t)t0 * nbl * nout * 2, ctx->dt_out.p, (size_t)nrows * nout * 2 * sizeof(double),
                                 hipMemcpyDeviceToHost, ctx->stream));
    if (out_power)
      HIPCHK(ctx, hipMemcpyAsync(out_power + (size_t)t0 * nbl * nout, ctx->dt_pow.p, (size_t)nrows * nout * sizeof(double),
                                 hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));     // the staging buffers are reused by the next batch
  }
  if (lags_out) delay_lags(ctx, lags_out);
  return PRISIM_OK;
  });
}

int prisim_hip_delay_transform_device(prisim_ctx* ctx, int64_t nt, const double* bpwts, int64_t wts_rows, double pad, int want_lag, int want_power,
                                      double power_scale, double* lags_out, int64_t* nout_out) {
  return guarded(ctx, [&]() -> int {
  if (!ctx) return PRISIM_EINVAL;
  int rc;
  if ((rc = delay_check(ctx, nt, pad))) return rc;
  if (!want_lag && !want_power) return fail(ctx, PRISIM_EINVAL, "nothing to compute: want_lag and want_power are both 0");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  const DelayGeom g = delay_geom(ctx, pad);
  const int64_t nbl = ctx->nbl, nout = g.nout;
  const int64_t nt_b = delay_batch_snapshots(ctx, g, nt);
  if ((rc = delay_prepare(ctx, g, bpwts, wts_rows, nt_b * nbl))) return rc;
  ctx->dt_have_lag = ctx->dt_have_pow = false;
  if (want_lag && (rc = ensure(ctx, ctx->dt_lag_all, (size_t)nt * nbl * nout * 2 * sizeof(double)))) return rc;
  if (want_power && (rc = ensure(ctx, ctx->dt_pow_all, (size_t)nt * nbl * nout * sizeof(double)))) return rc;
  if (!ctx->ev_d0) {
    HIPCHK(ctx, hipEventCreate(&ctx->ev_d0));
    HIPCHK(ctx, hipEventCreate(&ctx->ev_d1));
  }
  HIPCHK(ctx, hipEventRecord(ctx->ev_d0, ctx->stream));
  for (int64_t t0 = 0; t0 < nt; t0 += nt_b) {
    const int64_t ntc = std::min(nt_b, nt - t0);
    double* d_out = want_lag ? (double*)ctx->dt_lag_all.p + (size_t)t0 * nbl * nout * 2 : nullptr;
    double* d_pow = want_power ? (double*)ctx->dt_pow_all.p + (size_t)t0 * nbl * nout : nullptr;
    if ((rc = delay_batch(ctx, g, t0, ntc * nbl, bpwts != nullptr, wts_rows, d_out, d_pow, power_scale))) return rc;
  }
  HIPCHK(ctx, hipEventRecord(ctx->ev_d1, ctx->stream));
  ctx->dt_nt = nt; ctx->dt_nout = nout;
  ctx->dt_have_lag = want_lag != 0; ctx->dt_have_pow = want_power != 0;
  ctx->timing.last_delay_fused = g.fused ? 1 : 0;
  ctx->timing.last_delay_ms = -1.0;         // filled in by get_timing once the events have completed
  if (lags_out) delay_lags(ctx, lags_out);
  if (nout_out) *nout_out = nout;
  return PRISIM_OK;
  });
}

// Copy rows of the resident spectra to the host.  rows == NULL: all nbl baselines.
static int get_resident(prisim_ctx* ctx, const DevBuf& buf, bool have, int reals, int64_t t0, int64_t nt, const int64_t* rows, int64_t nrows,
                        double* out) {
  if (!have || !buf.p) return fail(ctx, PRISIM_ESTATE, "delay_transform_device has not produced this quantity");
  if (!out) return fail(ctx, PRISIM_EINVAL, "out is NULL");
  if (t0 < 0 || nt <= 0 || t0 + nt > ctx->dt_nt) return fail(ctx, PRISIM_EINVAL, "snapshot range outside the resident spectra");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  const int64_t nbl = ctx->nbl, nout = ctx->dt_nout;
  const size_t row_bytes = (size_t)nout * reals * sizeof(double);
  const char* base = (const char*)buf.p;
  if (!rows) {
    HIPCHK(ctx, hipMemcpy(out, base + (size_t)t0 * nbl * row_bytes, (size_t)nt * nbl * row_bytes, hipMemcpyDeviceToHost));
    return PRISIM_OK;
  }
  if (nrows <= 0) return fail(ctx, PRISIM_EINVAL, "nrows must be positive when rows is given");
  for (int64_t i = 0; i < nrows; ++i)
    if (rows[i] < 0 || rows[i] >= nbl) return fail(ctx, PRISIM_EINVAL, "row index out of range");
  for (int64_t t = 0; t < nt; ++t)
    for (int64_t i = 0; i < nrows; ++i)
      HIPCHK(ctx, hipMemcpyAsync((char*)out + ((size_t)t * nrows + i) * row_bytes, base + ((size_t)(t0 + t) * nbl + rows[i]) * row_bytes, row_bytes,
                                 hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  return PRISIM_OK;
}

int prisim_hip_get_lags(prisim_ctx* ctx, int64_t t0, int64_t nt, const int64_t* rows, int64_t nrows, double* out) {
  return guarded(ctx, [&]() -> int {
  if (!ctx) return PRISIM_EINVAL;
  return get_resident(ctx, ctx->dt_lag_all, ctx->dt_have_lag, 2, t0, nt, rows, nrows, out);
  });
}

int prisim_hip_get_delay_power(prisim_ctx* ctx, int64_t t0, int64_t nt, const int64_t* rows, int64_t nrows, double* out) {
  return guarded(ctx, [&]() -> int {
  if (!ctx) return PRISIM_EINVAL;
  return get_resident(ctx, ctx->dt_pow_all, ctx->dt_have_pow, 1, t0, nt, rows, nrows, out);
  });
}

int prisim_hip_phase_rotate(prisim_ctx* ctx, int64_t nt, const double* diff_dircos) {
  return guarded(ctx, [&]() -> int {
  if (!ctx) return PRISIM_EINVAL;
  if (!ctx->array_set) return fail(ctx, PRISIM_ESTATE, "set_array has not been called");
  if (nt <= 0 || nt > ctx->nt_max) return fail(ctx, PRISIM_EINVAL, "nt out of range");
  if (!diff_dircos) return fail(ctx, PRISIM_EINVAL, "diff_dircos is NULL");
  for (int64_t i = 0; i < 3 * nt; ++i)
    if (!std::isfinite(diff_dircos[i])) return fail(ctx, PRISIM_EINVAL, "non-finite phase-centre offset");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  int rc;
  if ((rc = ensure(ctx, ctx->scratch, std::max<size_t>(1025, (size_t)3 * nt) * sizeof(double)))) return rc;
  HIPCHK(ctx, hipMemcpyAsync(ctx->scratch.p, diff_dircos, (size_t)3 * nt * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(ctx, launch_phase_rotate((double*)ctx->cube.p, (const double*)ctx->blx.p, (const double*)ctx->bly.p, (const double*)ctx->blz.p,
                                  (const double*)ctx->freqs.p, (const double*)ctx->scratch.p, nt, ctx->nbl, ctx->nchan, ctx->stream));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));   // diff_dircos is caller-owned
  return PRISIM_OK;
  });
}

int prisim_hip_noise(prisim_ctx* ctx, int64_t nt, const double* rms, uint64_t seed, int64_t bl_offset, double* out) {
  return guarded(ctx, [&]() -> int {
  if (!ctx) return PRISIM_EINVAL;
  if (!ctx->array_set) return fail(ctx, PRISIM_ESTATE, "set_array has not been called");
  if (nt <= 0 || !rms || !out || bl_offset < 0) return fail(ctx, PRISIM_EINVAL, "bad noise arguments");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  const size_t n = (size_t)ctx->nbl * ctx->nchan;
  for (size_t i = 0; i < n * (size_t)nt; ++i)
    if (!(rms[i] >= 0.0) || !std::isfinite(rms[i])) return fail(ctx, PRISIM_EINVAL, "noise rms must be finite and non-negative");
  DevBuf drms, dout;
  int rc;
  if ((rc = ensure(ctx, drms, n * sizeof(double))) || (rc = ensure(ctx, dout, n * 2 * sizeof(double)))) {
    release(drms); release(dout);
    return rc;
  }
  hipError_t e = hipSuccess;
  for (int64_t t = 0; t < nt && e == hipSuccess; ++t) {
    e = hipMemcpyAsync(drms.p, rms + (size_t)t * n, n * sizeof(double), hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = launch_noise((const double*)drms.p, (double*)dout.p, ctx->nbl, ctx->nchan, t, bl_offset, seed, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(out + (size_t)t * n * 2, dout.p, n * 2 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
  }
  release(drms); release(dout);
  HIPCHK(ctx, e);
  return PRISIM_OK;
  });
}

int prisim_hip_noise_indexed(prisim_ctx* ctx, int64_t nt, const double* rms, uint64_t seed, const int64_t* bl_index, double* out) {
  return guarded(ctx, [&]() -> int {
  if (!ctx) return PRISIM_EINVAL;
  if (!ctx->array_set) return fail(ctx, PRISIM_ESTATE, "set_array has not been called");
  if (nt <= 0 || !rms || !out || !bl_index) return fail(ctx, PRISIM_EINVAL, "bad noise arguments");
  for (int64_t b = 0; b < ctx->nbl; ++b)
    if (bl_index[b] < 0) return fail(ctx, PRISIM_EINVAL, "negative global baseline index");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  const size_t n = (size_t)ctx->nbl * ctx->nchan;
  for (size_t i = 0; i < n * (size_t)nt; ++i)
    if (!(rms[i] >= 0.0) || !std::isfinite(rms[i])) return fail(ctx, PRISIM_EINVAL, "noise rms must be finite and non-negative");
  DevBuf drms, dout;
  int rc;
  if ((rc = ensure(ctx, drms, n * sizeof(double))) || (rc = ensure(ctx, dout, n * 2 * sizeof(double)))) {
    release(drms); release(dout);
    return rc;
  }
  hipError_t e = hipSuccess;
  for (int64_t t = 0; t < nt && e == hipSuccess; ++t) {
    e = hipMemcpyAsync(drms.p, rms + (size_t)t * n, n * sizeof(double), hipMemcpyHostToDevice, ctx->stream);
    // one launch per run of consecutive global indices: the generator's counter is (channel, global baseline, snapshot)
    for (int64_t b0 = 0; b0 < ctx->nbl && e == hipSuccess;) {
      int64_t b1 = b0 + 1;
      while (b1 < ctx->nbl && bl_index[b1] == bl_index[b1 - 1] + 1) ++b1;
      e = launch_noise((const double*)drms.p + (size_t)b0 * ctx->nchan, (double*)dout.p + (size_t)b0 * ctx->nchan * 2, b1 - b0, ctx->nchan, t,
                       bl_index[b0], seed, ctx->stream);
      b0 = b1;
    }
    if (e == hipSuccess) e = hipMemcpyAsync(out + (size_t)t * n * 2, dout.p, n * 2 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
  }
  release(drms); release(dout);
  HIPCHK(ctx, e);
  return PRISIM_OK;
  });
}

// ---- multi-GPU ------------------------------------------------------------------------------

int prisim_hip_comm_unique_id(char id[128]) {
  return guarded(nullptr, [&]() -> int {
  if (!id) return PRISIM_EINVAL;
  std::string lerr;
  if (!load_rccl(lerr)) return fail(nullptr, PRISIM_ELIB, lerr);
  ncclUniqueId uid;
  static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is expected to be 128 bytes");
  ncclResult_t r = g_rccl.GetUniqueId(&uid);
  if (r != ncclSuccess) return fail(nullptr, PRISIM_ELIB, std::string("ncclGetUniqueId: ") + g_rccl.GetErrorString(r));
  memcpy(id, &uid, 128);
  return PRISIM_OK;
  });
}

int prisim_hip_comm_version(char out[128]) {
  return guarded(nullptr, [&]() -> int {
  if (!out) return PRISIM_EINVAL;
  std::string lerr;
  if (!load_rccl(lerr)) return fail(nullptr, PRISIM_ELIB, lerr);
  int v = 0;
  if (g_rccl.GetVersion) (void)g_rccl.GetVersion(&v);
  // NCCL_VERSION_CODE: major * 10000 + minor * 100 + patch (2.9 and later)
  snprintf(out, 128, "librccl %d.%d.%d (%s)", v / 10000, (v / 100) % 100, v % 100, g_rccl.path.c_str());
  return PRISIM_OK;
  });
}

int prisim_hip_comm_last_error(char out[512]) {
  // Deliberately NOT through guarded() and without touching any context or the HIP runtime: this is what a watchdog thread calls while
  // the main thread sits inside ncclCommInitRank -- it must not wait for anything that thread holds.
  if (!out) return PRISIM_EINVAL;
  out[0] = 0;
  if (!g_rccl.handle) { snprintf(out, 512, "librccl not loaded"); return PRISIM_OK; }
  const char* msg = g_rccl.GetLastError ? g_rccl.GetLastError(nullptr) : nullptr;
  snprintf(out, 512, "%s", (msg && msg[0]) ? msg : "(librccl reports no error text; NCCL_DEBUG=WARN prints its warnings on stderr)");
  return PRISIM_OK;
}

int prisim_hip_device_pci(int device, char out[64]) {
  return guarded(nullptr, [&]() -> int {
  if (!out) return PRISIM_EINVAL;
  out[0] = 0;
  if (hipDeviceGetPCIBusId(out, 64, device) != hipSuccess) { (void)hipGetLastError(); snprintf(out, 64, "unknown"); }
  return PRISIM_OK;
  });
}

int prisim_hip_comm_init(prisim_ctx* ctx, const char id[128], int nranks, int rank) {
  return guarded(ctx, [&]() -> int {
  if (!ctx) return PRISIM_EINVAL;
  if (!id || nranks <= 0 || rank < 0 || rank >= nranks) return fail(ctx, PRISIM_EINVAL, "bad communicator arguments");
  std::string lerr;
  if (!load_rccl(lerr)) return fail(ctx, PRISIM_ELIB, lerr);
  HIPCHK(ctx, hipSetDevice(ctx->device));
  if (ctx->comm) { g_rccl.CommDestroy(ctx->comm); ctx->comm = nullptr; }
  ncclUniqueId uid;
  memcpy(&uid, id, 128);
  ncclResult_t r = g_rccl.CommInitRank(&ctx->comm, nranks, uid, rank);
  if (r != ncclSuccess) {
    ctx->comm = nullptr;
    return fail(ctx, PRISIM_ELIB, std::string("ncclCommInitRank: ") + g_rccl.GetErrorString(r));
  }
  if (nranks != ctx->nranks && ctx->nbl_total > 0) {
    // a shard map is a table of nranks x nbl_shard rows: one made for another communicator size must not be walked with this one
    ctx->nbl_total = 0; ctx->shard_map_h.clear(); release(ctx->shard_map); release(ctx->gathered);
  }
  ctx->nranks = nranks; ctx->rank = rank;
  return PRISIM_OK;
  });
}

// Gather snapshot `slot` of every rank on stream `st`.  src_all: this rank's [nt][planes][nbl][row] complex128 (the visibility cube, the
// resident lag spectra, or the gradient cube with planes = 3).  Without a shard map the snapshot lands in gathered[slot][rank][planes][b][row]
// (rank blocks as the all-gather leaves them); with one (prisim_hip_set_shard_map) it lands in a staging block of that stream and the
// un-deal kernel behind it writes gathered[slot][planes][global baseline][row] -- the reference's order (run_prisim.py:2233-2242).
static int gather_one_slot(prisim_ctx* ctx, const double* src_all, int64_t row, int planes, int64_t slot, int as_c64, hipStream_t st, int ring = -1) {
  const size_t shard = (size_t)ctx->nbl * row * 2 * (size_t)planes;   // reals per snapshot shard
  const size_t esz = as_c64 ? sizeof(float) : sizeof(double);
  const double* src = src_all + (size_t)slot * shard;
  const void* send = src;
  if (as_c64) {
    float* sb = (float*)ctx->sendbuf.p + (size_t)slot * shard;
    HIPCHK(ctx, launch_f64_to_f32(src, sb, (int64_t)shard, st));
    send = sb;
  }
  const bool receiver = ctx->gather_root < 0 || ctx->gather_root == ctx->rank;
  const bool ordered = ctx->nbl_total > 0;
  if (ordered && ctx->shard_map_h.size() != (size_t)ctx->nranks * (size_t)ctx->nbl)
    return fail(ctx, PRISIM_ESTATE, "the shard map does not match the communicator size and the shard size (set it after comm_init and set_array)");
  char* dst = nullptr;
  if (receiver) {
    if (ordered) {
      DevBuf& stage = (st == ctx->comm_stream && ctx->comm_stream) ? ctx->stage_comm : ctx->stage_main;
      int rc;
      if ((rc = ensure(ctx, stage, shard * (size_t)ctx->nranks * esz))) return rc;
      dst = (char*)stage.p;
    } else {
      dst = (char*)ctx->gathered.p + (size_t)slot * shard * (size_t)ctx->nranks * esz;
    }
  }
  if (ctx->nranks == 1 && !ctx->comm) {
    HIPCHK(ctx, hipMemcpyAsync(dst, send, shard * esz, hipMemcpyDeviceToDevice, st));
  } else {
    if (!ctx->comm) return fail(ctx, PRISIM_ESTATE, "comm_init has not been called");
    const ncclDataType_t ty = as_c64 ? ncclFloat : ncclDouble;
    if (ctx->gather_root < 0) {
      ncclResult_t r = g_rccl.AllGather(send, dst, shard, ty, ctx->comm, st);
      if (r != ncclSuccess) return fail(ctx, PRISIM_ELIB, std::string("ncclAllGather: ") + g_rccl.GetErrorString(r));
    } else {
      // gather to ONE rank (SURVEY 8(e) `gather_to_root`): the other GPUs keep no copy of the whole cube -- 120 GB at config 5.  One
      // grouped call: the root posts a receive per peer into that peer's block, every other rank one send; the root's own block is a copy.
      if (!g_rccl.Send) return fail(ctx, PRISIM_ELIB, "this librccl has no ncclSend / ncclRecv: gather to a root is unavailable");
      ncclResult_t r = g_rccl.GroupStart();
      if (r == ncclSuccess) {
        if (receiver) {
          for (int q = 0; q < ctx->nranks && r == ncclSuccess; ++q)
            if (q != ctx->rank) r = g_rccl.Recv(dst + (size_t)q * shard * esz, shard, ty, q, ctx->comm, st);
        } else {
          r = g_rccl.Send(send, shard, ty, ctx->gather_root, ctx->comm, st);
        }
        const ncclResult_t r2 = g_rccl.GroupEnd();
        if (r == ncclSuccess) r = r2;
      }
      if (r != ncclSuccess) return fail(ctx, PRISIM_ELIB, std::string("ncclSend/ncclRecv (gather to root): ") + g_rccl.GetErrorString(r));
      if (receiver) HIPCHK(ctx, hipMemcpyAsync(dst + (size_t)ctx->rank * shard * esz, send, shard * esz, hipMemcpyDeviceToDevice, st));
    }
  }
  if (ordered && receiver) {
    if (ring >= 0) HIPCHK(ctx, hipEventRecord(ctx->ev_gu[ring], st));
    const size_t row_words = (size_t)row * 2 * esz / 8;               // a row of complex numbers in 8-byte words
    char* out = (char*)ctx->gathered.p + (size_t)slot * (size_t)planes * (size_t)ctx->nbl_total * row_words * 8;
    HIPCHK(ctx, launch_undeal(dst, out, (const int64_t*)ctx->shard_map.p, ctx->nranks, ctx->nbl, ctx->nbl_total, planes, (int64_t)row_words, st));
  }
  return PRISIM_OK;
}

// bytes of the gathered cube for nt snapshots of rows of `row` complex numbers (x planes)
static size_t gathered_bytes(const prisim_ctx* ctx, int64_t nt, int64_t row_total, bool c64) {
  const size_t esz = c64 ? sizeof(float) : sizeof(double);
  const size_t rows = ctx->nbl_total > 0 ? (size_t)ctx->nbl_total : (size_t)ctx->nbl * (size_t)ctx->nranks;
  return (size_t)nt * rows * (size_t)row_total * 2 * esz;
}

static int ensure_gather_buffers(prisim_ctx* ctx, int64_t row, int as_c64) {
  const size_t shard = (size_t)ctx->nbl * row * 2;
  int rc;
  if (ctx->gathered.p && (ctx->gathered_c64 != (as_c64 != 0) || ctx->gathered_row != row)) release(ctx->gathered);
  const bool receiver = ctx->gather_root < 0 || ctx->gather_root == ctx->rank;       // only receivers hold the whole cube
  if ((rc = ensure(ctx, ctx->gathered, receiver ? gathered_bytes(ctx, ctx->nt_max, row, as_c64 != 0) : 16))) return rc;
  if (as_c64 && (rc = ensure(ctx, ctx->sendbuf, shard * (size_t)ctx->nt_max * sizeof(float)))) return rc;
  ctx->gathered_c64 = as_c64 != 0;
  ctx->gathered_row = row;
  return PRISIM_OK;
}

int prisim_hip_allgather(prisim_ctx* ctx, int64_t nt, int as_c64) {
  return guarded(ctx, [&]() -> int {
  if (!ctx) return PRISIM_EINVAL;
  if (!ctx->array_set) return fail(ctx, PRISIM_ESTATE, "set_array has not been called");
  if (nt <= 0 || nt > ctx->nt_max) return fail(ctx, PRISIM_EINVAL, "nt out of range");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  int rc;
  if (ctx->comm_stream && ctx->comm_pending) { HIPCHK(ctx, hipStreamSynchronize(ctx->comm_stream)); ctx->comm_pending = false; }
  if ((rc = ensure_gather_buffers(ctx, ctx->nchan, as_c64))) return rc;
  for (int64_t t = 0; t < nt; ++t)
    if ((rc = gather_one_slot(ctx, (const double*)ctx->cube.p, ctx->nchan, 1, t, as_c64, ctx->stream))) return rc;
  return PRISIM_OK;
  });
}

int prisim_hip_allgather_lags(prisim_ctx* ctx, int64_t nt) {
  return guarded(ctx, [&]() -> int {
  if (!ctx) return PRISIM_EINVAL;
  if (!ctx->array_set) return fail(ctx, PRISIM_ESTATE, "set_array has not been called");
  if (!ctx->dt_have_lag || !ctx->dt_lag_all.p) return fail(ctx, PRISIM_ESTATE, "delay_transform_device(want_lag) must be called first");
  if (nt <= 0 || nt > ctx->dt_nt || nt > ctx->nt_max) return fail(ctx, PRISIM_EINVAL, "nt does not match the resident lag spectra");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  int rc;
  if (ctx->comm_stream && ctx->comm_pending) { HIPCHK(ctx, hipStreamSynchronize(ctx->comm_stream)); ctx->comm_pending = false; }
  if ((rc = ensure_gather_buffers(ctx, ctx->dt_nout, 0))) return rc;
  for (int64_t t = 0; t < nt; ++t)
    if ((rc = gather_one_slot(ctx, (const double*)ctx->dt_lag_all.p, ctx->dt_nout, 1, t, 0, ctx->stream))) return rc;
  return PRISIM_OK;
  });
}

int prisim_hip_allgather_slot_async(prisim_ctx* ctx, int64_t slot, int as_c64) {
  return guarded(ctx, [&]() -> int {
  if (!ctx) return PRISIM_EINVAL;
  if (!ctx->array_set) return fail(ctx, PRISIM_ESTATE, "set_array has not been called");
  if (slot < 0 || slot >= ctx->nt_max) return fail(ctx, PRISIM_EINVAL, "slot out of range");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  int rc;
  if ((rc = ensure_comm_stream(ctx))) return rc;
  if (!ctx->gathered.p || ctx->gathered_c64 != (as_c64 != 0) || ctx->gathered_row != ctx->nchan) {
    // (re)allocation must not race with gathers in flight
    HIPCHK(ctx, hipStreamSynchronize(ctx->comm_stream));
    if ((rc = ensure_gather_buffers(ctx, ctx->nchan, as_c64))) return rc;
  }
  if (ctx->cring_pending >= prisim_ctx::kCommRing) {          // a full ring waits for its oldest entry only
    const int o = (ctx->cring_head - ctx->cring_pending + 2 * prisim_ctx::kCommRing) % prisim_ctx::kCommRing;
    HIPCHK(ctx, hipEventSynchronize(ctx->ev_g1[o]));
  }
  harvest_comm(ctx, false);
  const int ri = ctx->cring_head;
  // the gather of slot t waits for everything enqueued so far on the compute stream (i.e. compute(slot t)) ...
  HIPCHK(ctx, hipEventRecord(ctx->ev_gc[ri], ctx->stream));
  HIPCHK(ctx, hipStreamWaitEvent(ctx->comm_stream, ctx->ev_gc[ri], 0));
  // ... and runs on the (highest-priority) communication stream, overlapping the next snapshot's compute
  HIPCHK(ctx, hipEventRecord(ctx->ev_g0[ri], ctx->comm_stream));
  if ((rc = gather_one_slot(ctx, (const double*)ctx->cube.p, ctx->nchan, 1, slot, as_c64, ctx->comm_stream, ri))) return rc;
  HIPCHK(ctx, hipEventRecord(ctx->ev_g1[ri], ctx->comm_stream));
  ctx->cring_head = (ctx->cring_head + 1) % prisim_ctx::kCommRing;
  ctx->cring_pending += 1;
  ctx->cstats.bytes_per_peer = (int64_t)((size_t)ctx->nbl * ctx->nchan * 2 * (as_c64 ? sizeof(float) : sizeof(double)));
  ctx->cstats.nranks = ctx->nranks;
  ctx->comm_pending = true;
  return PRISIM_OK;
  });
}

int prisim_hip_get_gathered(prisim_ctx* ctx, int64_t nt, void* out) {
  return guarded(ctx, [&]() -> int {
  if (!ctx) return PRISIM_EINVAL;
  if (!out) return fail(ctx, PRISIM_EINVAL, "out is NULL");
  if (!ctx->gathered.p) return fail(ctx, PRISIM_ESTATE, "allgather has not been called");
  if (ctx->gather_root >= 0 && ctx->gather_root != ctx->rank) return fail(ctx, PRISIM_ESTATE, "the cube was gathered to another rank (set_gather_root)");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  const size_t bytes = gathered_bytes(ctx, nt, ctx->gathered_row, ctx->gathered_c64);
  if (nt <= 0 || bytes > ctx->gathered.bytes) return fail(ctx, PRISIM_EINVAL, "nt does not match the gathered cube");
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  if (ctx->comm_stream) { HIPCHK(ctx, hipStreamSynchronize(ctx->comm_stream)); ctx->comm_pending = false; }
  HIPCHK(ctx, hipMemcpy(out, ctx->gathered.p, bytes, hipMemcpyDeviceToHost));
  return PRISIM_OK;
  });
}

int prisim_hip_gathered_checksum(prisim_ctx* ctx, int64_t nt, double* out) {
  return guarded(ctx, [&]() -> int {
  if (!ctx) return PRISIM_EINVAL;
  if (!out) return fail(ctx, PRISIM_EINVAL, "out is NULL");
  if (!ctx->gathered.p) return fail(ctx, PRISIM_ESTATE, "allgather has not been called");
  if (ctx->gather_root >= 0 && ctx->gather_root != ctx->rank) return fail(ctx, PRISIM_ESTATE, "the cube was gathered to another rank (set_gather_root)");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  const size_t esz = ctx->gathered_c64 ? sizeof(float) : sizeof(double);
  const int64_t n = (int64_t)(gathered_bytes(ctx, nt, ctx->gathered_row, ctx->gathered_c64) / esz);
  if (nt <= 0 || (size_t)n * esz > ctx->gathered.bytes) return fail(ctx, PRISIM_EINVAL, "nt does not match the gathered cube");
  int rc;
  if ((rc = ensure(ctx, ctx->scratch, 1025 * sizeof(double)))) return rc;
  if (ctx->comm_stream) { HIPCHK(ctx, hipStreamSynchronize(ctx->comm_stream)); ctx->comm_pending = false; }
  HIPCHK(ctx, launch_checksum(ctx->gathered.p, ctx->gathered_c64, n, (double*)ctx->scratch.p, ctx->stream));
  HIPCHK(ctx, hipMemcpyAsync(out, ctx->scratch.p, sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  return PRISIM_OK;
  });
}

int prisim_hip_allgather_grad(prisim_ctx* ctx, int64_t nt, int as_c64) {
  return guarded(ctx, [&]() -> int {
  if (!ctx) return PRISIM_EINVAL;
  if (!ctx->array_set) return fail(ctx, PRISIM_ESTATE, "set_array has not been called");
  if (!ctx->grad.p) return fail(ctx, PRISIM_ESTATE, "no gradient has been computed");
  if (nt <= 0 || nt > ctx->nt_max) return fail(ctx, PRISIM_EINVAL, "nt out of range");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  int rc;
  if (ctx->comm_stream && ctx->comm_pending) { HIPCHK(ctx, hipStreamSynchronize(ctx->comm_stream)); ctx->comm_pending = false; }
  // a snapshot's gradient block is [3][nbl][nchan]: the same exchange with rows of 3 nchan -> gathered [nt][nranks][3][nbl][nchan]
  if ((rc = ensure_gather_buffers(ctx, 3 * ctx->nchan, as_c64))) return rc;
  for (int64_t t = 0; t < nt; ++t)
    if ((rc = gather_one_slot(ctx, (const double*)ctx->grad.p, ctx->nchan, 3, t, as_c64, ctx->stream))) return rc;
  return PRISIM_OK;
  });
}

int prisim_hip_set_shard_map(prisim_ctx* ctx, const int64_t* bl_index, int64_t nbl_total) {
  return guarded(ctx, [&]() -> int {
  if (!ctx) return PRISIM_EINVAL;
  if (!ctx->array_set) return fail(ctx, PRISIM_ESTATE, "set_array (and comm_init on more than one rank) must be called before set_shard_map");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  const size_t n = (size_t)ctx->nranks * (size_t)ctx->nbl;
  std::vector<int64_t> map;
  if (bl_index) {
    if (nbl_total <= 0 || (size_t)nbl_total > n) return fail(ctx, PRISIM_EINVAL, "nbl_total must be in [1, nranks x nbl_shard]");
    // every global baseline exactly once; anything else is padding (-1)
    std::vector<uint8_t> seen((size_t)nbl_total, 0);
    map.assign(bl_index, bl_index + n);
    for (size_t i = 0; i < n; ++i) {
      if (map[i] < 0) { map[i] = -1; continue; }
      if (map[i] >= nbl_total) return fail(ctx, PRISIM_EINVAL, "shard map: global baseline index out of range");
      if (seen[(size_t)map[i]]) return fail(ctx, PRISIM_EINVAL, "shard map: a global baseline is listed twice (padding rows must be -1)");
      seen[(size_t)map[i]] = 1;
    }
    for (int64_t g = 0; g < nbl_total; ++g)
      if (!seen[(size_t)g]) return fail(ctx, PRISIM_EINVAL, "shard map: a global baseline is missing");
  }
  // gathers in flight use the old layout
  if (ctx->comm_stream && ctx->comm_pending) { HIPCHK(ctx, hipStreamSynchronize(ctx->comm_stream)); ctx->comm_pending = false; }
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  release(ctx->gathered);                  // its layout follows the map
  if (!bl_index) {
    ctx->nbl_total = 0; ctx->shard_map_h.clear(); release(ctx->shard_map);
    return PRISIM_OK;
  }
  int rc;
  if ((rc = ensure(ctx, ctx->shard_map, n * sizeof(int64_t)))) return rc;
  HIPCHK(ctx, hipMemcpy(ctx->shard_map.p, map.data(), n * sizeof(int64_t), hipMemcpyHostToDevice));
  ctx->shard_map_h.swap(map);
  ctx->nbl_total = nbl_total;
  return PRISIM_OK;
  });
}

int prisim_hip_set_gather_root(prisim_ctx* ctx, int root) {
  return guarded(ctx, [&]() -> int {
  if (!ctx) return PRISIM_EINVAL;
  if (root < -1 || root >= ctx->nranks) return fail(ctx, PRISIM_EINVAL, "root must be -1 (all ranks) or a rank of the communicator");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  if (ctx->comm_stream && ctx->comm_pending) { HIPCHK(ctx, hipStreamSynchronize(ctx->comm_stream)); ctx->comm_pending = false; }
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  if (root != ctx->gather_root) release(ctx->gathered);          // its size depends on who receives
  ctx->gather_root = root;
  return PRISIM_OK;
  });
}

int prisim_hip_comm_selftest(prisim_ctx* ctx, int64_t bytes) {
  return guarded(ctx, [&]() -> int {
  if (!ctx) return PRISIM_EINVAL;
  if (bytes < 4 || bytes > ((int64_t)1 << 30)) return fail(ctx, PRISIM_EINVAL, "selftest size must be in [4 B, 1 GiB]");
  if (ctx->nranks > 1 && !ctx->comm) return fail(ctx, PRISIM_ESTATE, "comm_init has not been called");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  const size_t n = (size_t)bytes / 4;
  const int nr = ctx->nranks;
  std::vector<uint32_t> h(n * (size_t)nr);
  auto pattern = [](int r, size_t i) { return (uint32_t)(r + 1) * 0x9E3779B1u ^ (uint32_t)(i * 0x85EBCA6Bu + 0x1234567u); };
  for (size_t i = 0; i < n; ++i) h[i] = pattern(ctx->rank, i);
  DevBuf dsend, drecv;
  int rc;
  if ((rc = ensure(ctx, dsend, n * 4)) || (rc = ensure(ctx, drecv, n * 4 * (size_t)nr))) { release(dsend); release(drecv); return rc; }
  hipError_t e = hipMemcpyAsync(dsend.p, h.data(), n * 4, hipMemcpyHostToDevice, ctx->stream);
  if (e == hipSuccess) e = hipMemsetAsync(drecv.p, 0, n * 4 * (size_t)nr, ctx->stream);
  std::string nerr;
  if (e == hipSuccess) {
    if (ctx->comm) {
      ncclResult_t r = g_rccl.AllGather(dsend.p, drecv.p, n, ncclUint32, ctx->comm, ctx->stream);
      if (r != ncclSuccess) nerr = std::string("ncclAllGather (self-test): ") + g_rccl.GetErrorString(r);
    } else {
      e = hipMemcpyAsync(drecv.p, dsend.p, n * 4, hipMemcpyDeviceToDevice, ctx->stream);
    }
  }
  if (e == hipSuccess && nerr.empty()) e = hipMemcpyAsync(h.data(), drecv.p, n * 4 * (size_t)nr, hipMemcpyDeviceToHost, ctx->stream);
  if (e == hipSuccess && nerr.empty()) e = hipStreamSynchronize(ctx->stream);
  release(dsend); release(drecv);
  if (!nerr.empty()) return fail(ctx, PRISIM_ELIB, nerr);
  HIPCHK(ctx, e);
  for (int r = 0; r < nr; ++r)
    for (size_t i = 0; i < n; ++i)
      if (h[(size_t)r * n + i] != pattern(r, i))
        return fail(ctx, PRISIM_ELIB, "RCCL self-test: block of rank " + std::to_string(r) + " arrived corrupted at word " + std::to_string(i) +
                                          " on rank " + std::to_string(ctx->rank));
  return PRISIM_OK;
  });
}

int prisim_hip_get_comm_stats(prisim_ctx* ctx, prisim_comm_stats* out, int reset) {
  return guarded(ctx, [&]() -> int {
  if (!ctx) return PRISIM_EINVAL;
  if (!out) return fail(ctx, PRISIM_EINVAL, "out is NULL");
  if (ctx->cring_pending > 0) {
    HIPCHK(ctx, hipSetDevice(ctx->device));
    harvest_comm(ctx, true);
  }
  ctx->cstats.nranks = ctx->nranks;
  *out = ctx->cstats;
  if (reset) {
    ctx->cstats.n_gathers = 0;
    ctx->cstats.sum_gather_ms = ctx->cstats.max_gather_ms = ctx->cstats.last_gather_ms = ctx->cstats.last_gather_after_compute_ms = 0.0;
    ctx->cstats.sum_undeal_ms = ctx->cstats.last_undeal_ms = 0.0;
  }
  return PRISIM_OK;
  });
}

// ---- host-visible results without a serial PCIe tail --------------------------------------------

int prisim_hip_host_alloc(int64_t bytes, void** out) {
  return guarded(nullptr, [&]() -> int {
  if (!out) return fail(nullptr, PRISIM_EINVAL, "out is NULL");
  *out = nullptr;
  if (bytes <= 0) return fail(nullptr, PRISIM_EINVAL, "bytes must be positive");
  void* p = nullptr;
  hipError_t e = hipHostMalloc(&p, (size_t)bytes, hipHostMallocDefault);
  if (e != hipSuccess) return fail(nullptr, e == hipErrorOutOfMemory ? PRISIM_ENOMEM : PRISIM_ENODEV,
                                   std::string("hipHostMalloc(") + std::to_string(bytes) + " B): " + hipGetErrorString(e));
  *out = p;
  return PRISIM_OK;
  });
}

int prisim_hip_host_free(void* p) {
  return guarded(nullptr, [&]() -> int {
  if (p && hipHostFree(p) != hipSuccess) return fail(nullptr, PRISIM_EINVAL, "hipHostFree: not a prisim_hip_host_alloc pointer");
  return PRISIM_OK;
  });
}

int prisim_hip_get_vis_async(prisim_ctx* ctx, int64_t slot, void* vis, void* grad, int out_is_c64) {
  return guarded(ctx, [&]() -> int {
  if (!ctx) return PRISIM_EINVAL;
  if (!ctx->array_set) return fail(ctx, PRISIM_ESTATE, "set_array has not been called");
  if (slot < 0 || slot >= ctx->nt_max) return fail(ctx, PRISIM_EINVAL, "slot out of range");
  if (!vis) return fail(ctx, PRISIM_EINVAL, "vis is NULL");
  if (grad && !ctx->grad.p) return fail(ctx, PRISIM_ESTATE, "no gradient has been computed");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  int rc;
  if ((rc = ensure_copy_stream(ctx))) return rc;
  const size_t slot_elems = (size_t)ctx->nbl * ctx->nchan * 2;
  const double* src = (const double*)ctx->cube.p + (size_t)slot * slot_elems;
  const double* gsrc = grad ? (const double*)ctx->grad.p + (size_t)slot * 3 * slot_elems : nullptr;
  // the download waits for everything enqueued so far on the compute stream (the snapshot's sky-sum) and then runs on the copy stream,
  // under the next snapshot's compute
  HIPCHK(ctx, hipEventRecord(ctx->ev_copy_ready, ctx->stream));
  HIPCHK(ctx, hipStreamWaitEvent(ctx->copy_stream, ctx->ev_copy_ready, 0));
  if (!out_is_c64) {
    HIPCHK(ctx, hipMemcpyAsync(vis, src, slot_elems * sizeof(double), hipMemcpyDeviceToHost, ctx->copy_stream));
    if (grad) HIPCHK(ctx, hipMemcpyAsync(grad, gsrc, 3 * slot_elems * sizeof(double), hipMemcpyDeviceToHost, ctx->copy_stream));
  } else {
    // rounded to complex64 on the device (the reference's memsave dtype, :6183): half the PCIe bytes.  One staging buffer is enough --
    // conversion and copy of slot t+1 follow the copy of slot t in stream order.
    if (ctx->dl_stage.bytes < 4 * slot_elems * sizeof(float)) {
      HIPCHK(ctx, hipStreamSynchronize(ctx->copy_stream));      // a copy in flight may still read the old buffer
      if ((rc = ensure(ctx, ctx->dl_stage, 4 * slot_elems * sizeof(float)))) return rc;
    }
    float* st = (float*)ctx->dl_stage.p;
    HIPCHK(ctx, launch_f64_to_f32(src, st, (int64_t)slot_elems, ctx->copy_stream));
    HIPCHK(ctx, hipMemcpyAsync(vis, st, slot_elems * sizeof(float), hipMemcpyDeviceToHost, ctx->copy_stream));
    if (grad) {
      HIPCHK(ctx, launch_f64_to_f32(gsrc, st + slot_elems, (int64_t)(3 * slot_elems), ctx->copy_stream));
      HIPCHK(ctx, hipMemcpyAsync(grad, st + slot_elems, 3 * slot_elems * sizeof(float), hipMemcpyDeviceToHost, ctx->copy_stream));
    }
  }
  ctx->copy_pending = true;
  return PRISIM_OK;
  });
}

int prisim_hip_wait_downloads(prisim_ctx* ctx) {
  return guarded(ctx, [&]() -> int {
  if (!ctx) return PRISIM_EINVAL;
  if (ctx->copy_stream && ctx->copy_pending) {
    HIPCHK(ctx, hipSetDevice(ctx->device));
    HIPCHK(ctx, hipStreamSynchronize(ctx->copy_stream));
    ctx->copy_pending = false;
  }
  return PRISIM_OK;
  });
}

}  // extern "C"
