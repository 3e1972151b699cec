// ctx_internal.h -- the per-GPU context and the helpers shared by the translation units of libprisim_hip.so (capi.cpp: the C-ABI of
// the array / sky / compute / delay / gather entries; catalog.cpp: the device-resident catalogue path).  Not part of the public ABI.
#pragma once
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <rccl/rccl.h>
#include <rocfft/rocfft.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "../../include/prisim_hip.h"
#include "skyvis_kernels.h"

using namespace prisim;


namespace pint {

constexpr double kC = 299792458.0;   // scipy.constants.c (baseline_delay_horizon.py:236)

extern std::string g_create_error;

struct RcclApi {
  void* handle = nullptr;
  decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
  decltype(&ncclCommInitRank) CommInitRank = nullptr;
  decltype(&ncclAllGather) AllGather = nullptr;
  decltype(&ncclSend) Send = nullptr;                   // optional (gather to one root): absent in very old RCCLs
  decltype(&ncclRecv) Recv = nullptr;
  decltype(&ncclGroupStart) GroupStart = nullptr;
  decltype(&ncclGroupEnd) GroupEnd = nullptr;
  decltype(&ncclCommDestroy) CommDestroy = nullptr;
  decltype(&ncclGetErrorString) GetErrorString = nullptr;
  decltype(&ncclGetVersion) GetVersion = nullptr;         // optional
  const char* (*GetLastError)(ncclComm_t) = nullptr;      // optional (NCCL >= 2.13): text of the last error / warning
  std::string path;                                       // what dlopen() took
};
extern RcclApi g_rccl;

struct RocfftApi {
  void* handle = nullptr;
  bool setup_done = false;
  decltype(&rocfft_setup) setup = nullptr;
  decltype(&rocfft_plan_create) plan_create = nullptr;
  decltype(&rocfft_plan_destroy) plan_destroy = nullptr;
  decltype(&rocfft_plan_get_work_buffer_size) plan_get_work_buffer_size = nullptr;
  decltype(&rocfft_execution_info_create) execution_info_create = nullptr;
  decltype(&rocfft_execution_info_destroy) execution_info_destroy = nullptr;
  decltype(&rocfft_execution_info_set_stream) execution_info_set_stream = nullptr;
  decltype(&rocfft_execution_info_set_work_buffer) execution_info_set_work_buffer = nullptr;
  decltype(&rocfft_execute) execute = nullptr;
};
extern RocfftApi g_rocfft;

template <typename F>
bool load_sym(void* h, const char* name, F& out) {
  out = reinterpret_cast<F>(dlsym(h, name));
  return out != nullptr;
}

bool load_rccl(std::string& err);
bool load_rocfft(std::string& err);

struct DevBuf {
  void* p = nullptr;
  size_t bytes = 0;
};

}  // namespace pint

using namespace pint;

// What one snapshot's sky occupies on the device between its preparation (beam x flux, packing, direction prep, flags and tables) and
// its sky-sum.  The context holds TWO sets: on the catalogue path the preparation of snapshot t+1 runs on the preparation stream into
// one set while the sky-sum of snapshot t reads the other (ev_prep: the set is ready; ev_sum: its sky-sum has run and it may be rewritten).
struct SkyBufs {
  DevBuf pb;                 // [nsrc][nchan] float64 beam x flux
  DevBuf packed;             // [ntiles][nsrc_pad][CT] rows of the kernels
  DevBuf dirs_prep;          // [nsrc_pad][4] (s - s_pc)/c, kappa
  DevBuf dirs_c32;           // fused fp32 gradient kernel
  DevBuf lift_flags;         // [groups] lifting-rotation flags and what they were formed for
  double lift_key_k = -1.0;
  int lift_key_f32 = -1;
  int lift_groups = -1;
  DevBuf split_flags, moments, moments_part, split_count;      // split taper form: per-run moments and per-group flags
  // Taper culling: cull_first[prec][run][group] = first source of the run that group still has to sum (device, int32); the sources before
  // it contribute < exp(-18) (fp32) / exp(-28) (fp64) of sum|pbflux| to every baseline of the group.
  DevBuf cull_first;
  DevBuf batch_tab;                    // many snapshots per launch: the BatchSnap table of the chunk
  DevBuf dirs_sorted, idx_sorted;      // catalogue path: directions / catalogue indices in the altitude order of the taper culling
  hipEvent_t ev_prep = nullptr, ev_sum = nullptr;
  bool sum_recorded = false;
};

struct prisim_ctx {
  SkyBufs skb[2];
  SkyBufs* sk = &skb[0];     // the set of the current sky
  int sk_next = 0;           // catalogue path: the set the next snapshot's preparation writes
  hipStream_t prep_stream = nullptr;      // preparation stream (catalogue path, highest priority); nullptr: everything on `stream`
  bool prep_async = false;   // the current sky was prepared on prep_stream: compute() hands its packing to that stream too
  int fsq_pairs_ct = -1, fsq_pairs_ntiles = -1;      // tiling the fsq_pairs table was last formed for
  int device = 0;
  hipStream_t stream = nullptr;
  std::string err;
  char devname[64] = {0};
  int cu_count = 0, clock_khz = 0;

  // array
  bool array_set = false;
  int64_t nbl = 0, nchan = 0, nt_max = 0;
  DevBuf blx, bly, blz, freqs, fsq, fsq_pairs, cube, grad;
  std::vector<double> grp_maxlen;     // max |b| per group of kBlockThreads baselines (lifting-rotation guarantee)
  std::vector<double> grp_maxh, grp_maxz;   // max horizontal length / max |b_z| per group (bound of the split taper's parabola)
  std::vector<double> grp_minh;             // min horizontal length per group (taper culling)
  // Taper culling: cull_first[prec][run][group] = first source of the run that group still has to sum (device, int32); the sources before
  // it contribute < exp(-18) (fp32) / exp(-28) (fp64) of sum|pbflux| to every baseline of the group.  cull_frac[prec]: culled share of
  // the snapshot's terms; cull_any[prec]: anything culled at all.
  std::vector<double> cull_rho, cull_an;    // scratch of the cull-table walk: sin / |cos| of the zenith angle of a run's leading sources
  bool cull_any[2] = {false, false};
  double cull_frac[2] = {0.0, 0.0};
  int cull_nruns = 0;
  // runs of consecutive sources with one source size kappa (HEALPix skies: one run; point sources + diffuse: two): the packed fp32
  // taper kernel walks such skies run by run in its split form.  Empty: sizes vary from source to source (or no taper).
  struct KappaRun { int64_t lo, hi; double kappa; int tab_row; };      // tab_row: this run's row of the cull table
  std::vector<KappaRun> kappa_runs;
  DevBuf grp_hz;      // grp_hz: [4][groups] (max horizontal length, max |b_z|, max length, min horizontal length) on the device
  int32_t* h_split_count = nullptr;                      // pinned: uncorrected-group counts of the last split launch, per run (read after a sync)
  int split_count_runs = 0;
  double dmax = 2.0;                  // max_s |s - s_pc| of the current sky
  std::vector<double> h_freqs;
  bool uniform = false;
  double f0 = 0.0, df = 0.0;
  int64_t nchan_pad = 0;

  // sky
  bool sky_set = false;
  int64_t nsrc = 0;
  bool taper = false;
  double pc[3] = {0, 0, 1};
  DevBuf dirs, partial, scratch;
  // per-snapshot sky inputs (flux_ref / spindex or a flux table, beamformer elements, validity flag): owned by the context so that
  // a set_sky_* call allocates nothing after the first snapshot
  DevBuf sky_flux, sky_sp, sky_bf, sky_flag;
  // pinned host staging for the small per-snapshot uploads (directions, flux_ref, spindex ...): the caller's arrays are copied here
  // and sent with hipMemcpyAsync, so set_sky_* returns without a stream synchronisation; ev_stage marks the last upload that
  // reads the area
  void* h_stage = nullptr;
  size_t h_stage_bytes = 0, h_stage_used = 0;
  hipEvent_t ev_stage = nullptr;
  bool stage_pending = false;
  bool stage_open = false;        // copies of the current group have been queued and stage_end() has not run yet
  // directions of the current sky ([nsrc][4] l, m, n, kappa): dirs.p when the sky was uploaded (set_sky_*), a catalogue geometry set
  // (or its altitude-sorted copy) on the catalogue path; src_index: that path's compacted catalogue indices in the same order
  const double* dirs_p = nullptr;
  const int32_t* src_index = nullptr;
  // device-resident catalogue (catalog.cpp): uploaded once per run, every snapshot's geometry formed on the device
  struct Catalog {
    bool loaded = false;
    int64_t n = 0;
    int coords = 0;                     // PRISIM_COORDS_*
    bool have_shape = false, have_spec = false;
    double ref_freq = 1.0;
    double kappa_max = 0.0;
    DevBuf lon, lat, ux, uy, uz, kappa, run_id, flux_ref, spindex, spec;      // lon / lat: staging of `location` for k_cat_prepare
    struct Run { int64_t lo, hi; double kappa; };
    std::vector<Run> runs;              // runs of one source size in catalogue order (<= 8); empty: no shapes, or sizes vary source by source
    // geometry outputs.  Sets 0 / 1 alternate so that the geometry of the next snapshot (or chunk of snapshots) runs on the geometry
    // stream under the sky-sum of the current one; set 2 is the scratch of prisim_hip_catalog_roi.
    struct Set {
      DevBuf idx, dirs, keys, pos;
      hipEvent_t ev_free = nullptr;     // recorded on the compute stream behind the last kernel that reads the set
      bool ev_recorded = false;
      hipEvent_t ev_prepared = nullptr; // recorded behind the preparation (sort, cull table, beam x flux) that reads the set: a second
      bool prep_recorded = false;       // set_sky_from_catalog without a compute() in between must not scatter under it
    } set[3];
    DevBuf block_off, snaps, out_dev, sort_tmp, keys_out, perm, culled;
    CatOut* out_host = nullptr;         // pinned [cap_snaps]
    CatSnap* snaps_host = nullptr;      // pinned [cap_snaps]
    uint64_t* culled_host = nullptr;    // pinned [2]
    BatchSnap* batch_host = nullptr;    // pinned [2][cap_batch_host]: per-snapshot layout of a batched chunk
    int64_t cap_batch_host = 0;
    hipEvent_t ev_tab[2] = {nullptr, nullptr};      // the upload that last read half h of batch_host has run
    bool tab_recorded[2] = {false, false};
    int tab_half = 0;
    int64_t cap_snaps = 0;
    int64_t chunk_nmax = 0;             // largest region of interest among the snapshots of the chunk whose geometry was formed last
    // per-snapshot tables of the batched launches: a ring of four, so that the geometry that WRITES the table of chunk c never waits for
    // the sums of chunk c - 1 that still read theirs (one event per table: recorded behind the sums that read it)
    DevBuf batch_tabs[4];
    hipEvent_t ev_tabfree[4] = {nullptr, nullptr, nullptr, nullptr};
    bool tabfree_rec[4] = {false, false, false, false};
    int tab_next = 0;
    bool geom_pending = false;          // a geometry is queued whose records nobody has waited for yet (geometry_enqueue / geometry_wait)
    int64_t geom_nsnap = 0;
    int geom_set = 0;
    std::chrono::steady_clock::time_point geom_t0;
    hipStream_t gstream = nullptr;      // geometry stream (highest priority: a few small kernels beside a sky-sum grid)
    hipEvent_t ev_geom = nullptr;
    hipEvent_t ev_join = nullptr;       // "everything queued on the compute stream so far" (first preparation-stream sky after an in-line one)
    int cur = -1;                       // set the current sky lives in; -1: the current sky was uploaded
    int next = 0;                       // set the next geometry call of the product path writes
    double geom_ms_sum = 0.0;           // host wall time spent waiting for geometry results (prisim_cat_stats)
    int64_t geom_calls = 0;
  } cat;
  int cull_frac_pending = 0;            // precision + 1 whose culled-pair count (catalogue path) get_timing still has to read
  // external beam
  DevBuf ext_table, ext_work, ext_colmax;
  int ext_nside = 0;

  // events / timing
  // hipEvent timing of compute(): a ring of event quadruples so that back-to-back compute() calls queue on the stream without
  // a host synchronisation; completed entries are harvested in order (lazily, or at sync / get_timing)
  static constexpr int kTimingRing = 16;
  hipEvent_t ev_c0[kTimingRing] = {}, ev_c1[kTimingRing] = {}, ev_k0[kTimingRing] = {}, ev_k1[kTimingRing] = {};
  int ring_head = 0;            // next entry to record
  int ring_pending = 0;         // recorded, not yet harvested (oldest = head - pending)
  // lifting flags of the last compute (host copy stays alive for the asynchronous upload) and what they were computed for
  prisim_timing timing{};

  // tuning overrides
  int tune_ct = 0, tune_chunk = 0, tune_nsplit = 0;

  // comm
  ncclComm_t comm = nullptr;
  hipStream_t comm_stream = nullptr;     // all-gathers overlapped with the next snapshot's compute
  hipEvent_t ev_slot_done = nullptr;
  bool comm_pending = false;
  int nranks = 1, rank = 0;
  int gather_root = -1;                  // -1: every rank receives the gathered cube (all-gather); r: only rank r does (ncclSend / ncclRecv)
  DevBuf gathered, sendbuf;
  bool gathered_c64 = false;
  // gather timing: a ring of (compute-stream marker, gather start, gather end) events per overlapped gather, harvested in order
  static constexpr int kCommRing = 32;
  hipEvent_t ev_gc[kCommRing] = {}, ev_g0[kCommRing] = {}, ev_g1[kCommRing] = {};
  int cring_head = 0, cring_pending = 0;
  prisim_comm_stats cstats{};
  // asynchronous downloads (prisim_hip_get_vis_async): copy stream behind an event on the compute stream
  hipStream_t copy_stream = nullptr;
  hipEvent_t ev_copy_ready = nullptr;
  bool copy_pending = false;
  DevBuf dl_stage;                       // complex64 staging of one slot (+ its three gradient slots)

  // fft
  rocfft_plan fft_plan = nullptr;
  rocfft_execution_info fft_info = nullptr;
  size_t fft_len = 0, fft_batch = 0;
  DevBuf fft_work, fft_buf, dt_out, dt_pow, dt_wts;
  // device-resident delay spectra of all snapshots (prisim_hip_delay_transform_device): [nt][nbl][nout] complex128 / float64
  DevBuf dt_lag_all, dt_pow_all, dt_tw;
  int64_t dt_tw_n = 0;              // channel count the twiddle table was built for
  int64_t dt_nt = 0, dt_nout = 0;   // shape of the resident spectra
  bool dt_have_lag = false, dt_have_pow = false;
  hipEvent_t ev_d0 = nullptr, ev_d1 = nullptr;
  int64_t gathered_row = 0;         // row length of the gathered cube (nchan for visibilities, nout for delay spectra)
  // shard map (prisim_hip_set_shard_map): with it every gather lands in a per-stream staging block and is un-dealt into the gathered
  // cube in GLOBAL baseline order, [nt][planes][nbl_total][row]; without it the cube keeps the rank-major layout of the all-gather
  std::vector<int64_t> shard_map_h; // [nranks][nbl] global baseline of every (rank, local row); -1 = padding
  DevBuf shard_map, stage_main, stage_comm;
  int64_t nbl_total = 0;            // > 0: the map is set
  hipEvent_t ev_gu[32] = {};        // between a gather and its un-deal kernel (communication-stream ring, as ev_g0 / ev_g1)
};

namespace pint {

inline int fail(prisim_ctx* ctx, int code, const std::string& msg) {
  try {
    if (ctx) ctx->err = msg; else g_create_error = msg;
  } catch (...) {      // the message itself could not be stored: the code still says what happened
  }
  return code;
}

// Every extern "C" entry runs its body through this: no C++ exception crosses the ABI; a failed host allocation
// becomes PRISIM_ENOMEM, anything else PRISIM_EINTERNAL with the exception's text.
template <typename F>
inline int guarded(prisim_ctx* ctx, F&& body) noexcept {
  try {
    return body();
  } catch (const std::bad_alloc&) {
    return fail(ctx, PRISIM_ENOMEM, "out of host memory");
  } catch (const std::exception& e) {
    const char* w = e.what();
    try { return fail(ctx, PRISIM_EINTERNAL, std::string("C++ exception: ") + (w ? w : "?")); } catch (...) { return PRISIM_EINTERNAL; }
  } catch (...) {
    return fail(ctx, PRISIM_EINTERNAL, "unknown C++ exception");
  }
}

#define HIPCHK(ctx, call)                                                                      \
  do {                                                                                         \
    hipError_t e_ = (call);                                                                    \
    if (e_ != hipSuccess) {                                                                    \
      return fail(ctx, e_ == hipErrorOutOfMemory ? PRISIM_ENOMEM : PRISIM_ENODEV,              \
                  std::string(#call) + ": " + hipGetErrorString(e_));                         \
    }                                                                                          \
  } while (0)

inline int ensure(prisim_ctx* ctx, DevBuf& b, size_t bytes) {
  if (bytes == 0) bytes = 16;
  if (b.bytes >= bytes && b.p) return PRISIM_OK;
  static const bool trace = getenv("PRISIM_HIP_TRACE_ALLOC") != nullptr;      // development hook: every (re)allocation with its cost, on stderr
  const auto t0 = std::chrono::steady_clock::now();
  const size_t had = b.bytes;
  if (b.p) { (void)hipFree(b.p); b.p = nullptr; b.bytes = 0; }
  hipError_t e = hipMalloc(&b.p, bytes);
  if (trace)
    fprintf(stderr, "[prisim_hip alloc] %zu B (had %zu): %.1f us\n", bytes, had,
            1e6 * std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
  if (e != hipSuccess) {
    b.p = nullptr;
    return fail(ctx, PRISIM_ENOMEM, std::string("hipMalloc(") + std::to_string(bytes) + " B): " + hipGetErrorString(e));
  }
  b.bytes = bytes;
  return PRISIM_OK;
}

// A buffer whose size follows something that drifts (the region of interest of a drift scan grows and shrinks by a few sources per
// snapshot): grow with a quarter of headroom, never beyond `cap` bytes, so that re-allocations -- each one a device-wide
// synchronisation -- stay rare without sizing for the worst case (a whole catalogue x nchan) up front.
inline int ensure_grow(prisim_ctx* ctx, DevBuf& b, size_t bytes, size_t cap = (size_t)-1) {
  if (b.bytes >= bytes && b.p) return PRISIM_OK;
  size_t want = bytes + bytes / 4;
  if (want > cap) want = cap;
  if (want < bytes) want = bytes;
  return ensure(ctx, b, want);
}

inline void release(DevBuf& b) {
  if (b.p) (void)hipFree(b.p);
  b.p = nullptr;
  b.bytes = 0;
}

inline int64_t round_up(int64_t v, int64_t m) { return (v + m - 1) / m * m; }

// Uploads above this size go straight from the caller's (pageable) memory and are waited for; smaller ones are staged.
constexpr size_t kStageMaxBytes = (size_t)64 << 20;

// Start a group of staged uploads needing `bytes` of pinned memory in total: waits until the previous group has left the area.
inline int stage_begin(prisim_ctx* ctx, size_t bytes) {
  if (ctx->stage_pending) {
    if (hipEventSynchronize(ctx->ev_stage) != hipSuccess) return fail(ctx, PRISIM_ENODEV, "hipEventSynchronize(staging) failed");
    ctx->stage_pending = false;
  }
  if (ctx->stage_open) {
    // the previous group was abandoned on an error path after some of its copies had been queued: let them leave the area first
    if (hipStreamSynchronize(ctx->stream) != hipSuccess) return fail(ctx, PRISIM_ENODEV, "hipStreamSynchronize(staging) failed");
    ctx->stage_open = false;
  }
  if (!ctx->ev_stage && hipEventCreateWithFlags(&ctx->ev_stage, hipEventDisableTiming) != hipSuccess)
    return fail(ctx, PRISIM_ENODEV, "hipEventCreate(staging) failed");
  bytes += 4096;
  if (bytes > ctx->h_stage_bytes) {
    if (ctx->h_stage) { (void)hipHostFree(ctx->h_stage); ctx->h_stage = nullptr; ctx->h_stage_bytes = 0; }
    const size_t want = std::max(bytes, (size_t)1 << 20);
    if (hipHostMalloc(&ctx->h_stage, want, hipHostMallocDefault) != hipSuccess) {
      ctx->h_stage = nullptr;
      return fail(ctx, PRISIM_ENOMEM, "hipHostMalloc(" + std::to_string(want) + " B) for the upload staging area failed");
    }
    ctx->h_stage_bytes = want;
  }
  ctx->h_stage_used = 0;
  return PRISIM_OK;
}

// Reserve `bytes` of the staging area (256-byte aligned); the caller fills it and then calls stage_send.
inline void* stage_alloc(prisim_ctx* ctx, size_t bytes) {
  const size_t off = (ctx->h_stage_used + 255) & ~(size_t)255;
  if (off + bytes > ctx->h_stage_bytes) return nullptr;
  ctx->h_stage_used = off + bytes;
  return (char*)ctx->h_stage + off;
}

inline hipError_t stage_send(prisim_ctx* ctx, void* dst, const void* staged, size_t bytes) {
  ctx->stage_open = true;
  return hipMemcpyAsync(dst, staged, bytes, hipMemcpyHostToDevice, ctx->stream);
}

// Copy a caller array into the staging area and send it.
inline hipError_t stage_upload(prisim_ctx* ctx, void* dst, const void* src, size_t bytes) {
  void* h = stage_alloc(ctx, bytes);
  if (!h) return hipErrorOutOfMemory;
  memcpy(h, src, bytes);
  return stage_send(ctx, dst, h, bytes);
}

inline void stage_end(prisim_ctx* ctx) {
  if (hipEventRecord(ctx->ev_stage, ctx->stream) == hipSuccess) {
    ctx->stage_pending = true;
    ctx->stage_open = false;         // otherwise stays set: the next group then waits for the whole stream
  }
}

// Collect the hipEvent timings of finished compute() calls, oldest first.  max_wait: how many of the pending entries may be
// waited for (hipEventSynchronize); the rest are taken only if already complete.  -1: wait for all of them.
inline void harvest_timing(prisim_ctx* ctx, int max_wait = -1) {
  while (ctx->ring_pending > 0) {
    const int i = (ctx->ring_head - ctx->ring_pending + 2 * prisim_ctx::kTimingRing) % prisim_ctx::kTimingRing;
    if (max_wait != 0) {
      if (hipEventSynchronize(ctx->ev_c1[i]) != hipSuccess) { ctx->ring_pending = 0; return; }
      if (max_wait > 0) --max_wait;
    } else if (hipEventQuery(ctx->ev_c1[i]) != hipSuccess) {
      return;
    }
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, ctx->ev_c0[i], ctx->ev_c1[i]) == hipSuccess) ctx->timing.last_compute_ms = ms;
    if (hipEventElapsedTime(&ms, ctx->ev_k0[i], ctx->ev_k1[i]) == hipSuccess) {
      ctx->timing.last_kernel_ms = ms;
      ctx->timing.sum_kernel_ms += ms;
      ctx->timing.n_kernel += 1;
    }
    ctx->ring_pending -= 1;
  }
}

// Collect finished gather timings (oldest first).  wait_all: hipEventSynchronize every pending entry; otherwise only take what is complete.
inline void harvest_comm(prisim_ctx* ctx, bool wait_all) {
  while (ctx->cring_pending > 0) {
    const int i = (ctx->cring_head - ctx->cring_pending + 2 * prisim_ctx::kCommRing) % prisim_ctx::kCommRing;
    if (wait_all) {
      if (hipEventSynchronize(ctx->ev_g1[i]) != hipSuccess) { ctx->cring_pending = 0; return; }
    } else if (hipEventQuery(ctx->ev_g1[i]) != hipSuccess) {
      return;
    }
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, ctx->ev_g0[i], ctx->ev_g1[i]) == hipSuccess) {
      ctx->cstats.last_gather_ms = ms;
      ctx->cstats.sum_gather_ms += ms;
      if (ms > ctx->cstats.max_gather_ms) ctx->cstats.max_gather_ms = ms;
      ctx->cstats.n_gathers += 1;
    }
    // what the overlap did not hide of THIS gather: its end against the compute-stream marker recorded when it was enqueued
    // (= the end of the snapshot's own sky-sum); only the last harvested entry is kept -- the gathers before it ran under later compute
    if (hipEventElapsedTime(&ms, ctx->ev_gc[i], ctx->ev_g1[i]) == hipSuccess) ctx->cstats.last_gather_after_compute_ms = ms;
    if (ctx->nbl_total > 0 && hipEventElapsedTime(&ms, ctx->ev_gu[i], ctx->ev_g1[i]) == hipSuccess) {
      ctx->cstats.last_undeal_ms = ms;
      ctx->cstats.sum_undeal_ms += ms;
    }
    ctx->cring_pending -= 1;
  }
}

// The communication stream gets the HIGHEST priority the device offers: its RCCL kernels are few blocks that must be scheduled
// beside a sky-sum grid occupying every CU; with equal priority they would only start as sky-sum blocks drain.
inline int ensure_comm_stream(prisim_ctx* ctx) {
  if (ctx->comm_stream) return PRISIM_OK;
  int least = 0, greatest = 0;
  if (hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess) { least = greatest = 0; (void)hipGetLastError(); }
  if (hipStreamCreateWithPriority(&ctx->comm_stream, hipStreamNonBlocking, greatest) != hipSuccess) {
    (void)hipGetLastError();
    ctx->comm_stream = nullptr;
    HIPCHK(ctx, hipStreamCreateWithFlags(&ctx->comm_stream, hipStreamNonBlocking));
    greatest = 0; least = 0;
  }
  ctx->cstats.stream_priority = greatest;
  ctx->cstats.stream_priority_lowest = least;
  HIPCHK(ctx, hipEventCreateWithFlags(&ctx->ev_slot_done, hipEventDisableTiming));
  for (int i = 0; i < prisim_ctx::kCommRing; ++i) {
    HIPCHK(ctx, hipEventCreate(&ctx->ev_gc[i]));
    HIPCHK(ctx, hipEventCreate(&ctx->ev_g0[i]));
    HIPCHK(ctx, hipEventCreate(&ctx->ev_g1[i]));
    HIPCHK(ctx, hipEventCreate(&ctx->ev_gu[i]));
  }
  return PRISIM_OK;
}

inline int ensure_copy_stream(prisim_ctx* ctx) {
  if (ctx->copy_stream) return PRISIM_OK;
  HIPCHK(ctx, hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking));
  HIPCHK(ctx, hipEventCreateWithFlags(&ctx->ev_copy_ready, hipEventDisableTiming));
  return PRISIM_OK;
}

constexpr int kMaxRunSets = 8;       // runs of one source size a split sky may have and still be summed run by run (partial-cube sets)

struct Plan {
  int kernel;      // PRISIM_KERNEL_*
  bool f32;
  int ct;
  int chunk;
  int nsplit;
  int64_t src_per_split;
  int64_t nsrc_pad;
  int ntiles;
  int nbgroups;
  bool pk;         // packed-fp32 kernel (k_skyvis_rec_f32pk) with interleaved pbflux pairs
};

// wave items (arrays of at most 256 baselines, grouped fp64 taper kernel): sources per split when nsrc sources are cut into s_want pieces
inline int64_t wave_split_sources(int64_t nsrc, int64_t s_want) {
  return round_up((std::max<int64_t>(nsrc, 1) + s_want - 1) / std::max<int64_t>(s_want, 1), 4);
}

// PRISIM_HIP_TAPER_F64_GROUP=0 (A/B hook): fp64 taper requests run the exact second-order kernel instead of the grouped one
inline bool taper_f64_grouped_enabled() {
  const char* env = getenv("PRISIM_HIP_TAPER_F64_GROUP");
  return !(env && atoi(env) == 0);
}

// PRISIM_HIP_GRAD_TAPER_GROUP=0 (A/B hook): fp64 gradients of a tapered sky run round 3's exact form on 16-channel tiles
inline bool grad_taper_grouped() {
  const char* env = getenv("PRISIM_HIP_GRAD_TAPER_GROUP");
  return !(env && atoi(env) == 0);
}

// the stream a sky is prepared on: the preparation stream when the current sky came from the resident catalogue, else the compute stream
inline hipStream_t pstream(const prisim_ctx* ctx) { return ctx->prep_async ? ctx->prep_stream : ctx->stream; }

// the compute stream waits for everything queued so far on the preparation stream (the current sky's set is ready)
inline int join_prep(prisim_ctx* ctx) {
  if (!ctx->prep_async) return PRISIM_OK;
  HIPCHK(ctx, hipEventRecord(ctx->sk->ev_prep, ctx->prep_stream));
  HIPCHK(ctx, hipStreamWaitEvent(ctx->stream, ctx->sk->ev_prep, 0));
  return PRISIM_OK;
}

// capi.cpp
Plan make_plan(const prisim_ctx* ctx, int precision, int kernel);
int upload_common(prisim_ctx* ctx, int64_t nsrc, const double* dircos, const double* pc_dircos, const double* fwhm_deg, size_t extra_stage_bytes);
int upload_any(prisim_ctx* ctx, void* dst, const void* src, size_t bytes, bool* synced);
int check_beam_spec(prisim_ctx* ctx, int beam_kind, double diameter_m, const double* beam_pc_dircos, const prisim_beam_ext* ext);
size_t beamformer_doubles(const prisim_beam_ext* ext);
int sky_beam_flux(prisim_ctx* ctx, int64_t ns, int beam_kind, double diameter_m, const double* beam_pc_dircos, const prisim_beam_ext* ext,
                  const double* d_flux_ref, const double* d_spindex, const double* d_flux_spec, double ref_freq, const int32_t* src_index);
int check_poly_beam_flag(prisim_ctx* ctx);
int extbeam_sky(prisim_ctx* ctx, int64_t nsrc, const double* d_fluxes, const double* d_flux_ref, const double* d_spindex, double ref_freq,
                const int32_t* src_index);
// catalog.cpp
void catalog_destroy(prisim_ctx* ctx);
int catalog_streams(prisim_ctx* ctx);             // the catalogue path's streams and events (created with the context)
void catalog_after_compute(prisim_ctx* ctx);        // marks the current geometry set free once the compute just enqueued has run

}  // namespace pint
