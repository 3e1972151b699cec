#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X-native PRISim sky-sum.

Metric (BASELINE.json): visibility-terms/s = nbl * nchan * nsrc * nt / wall, on the synthetic
HERA-350 x 1024-channel x 1e4-source workload (SURVEY.md 8(d) config 3, fp32 with tolerance check).

  python bench.py --gpus N --steps K --warmup W            (N > 1 launched bare: starts its own N ranks, prisim_amd.launch -- no torch)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W        (the driver's launcher: only its RANK / WORLD_SIZE are used)

A "step" is one snapshot: one pass of the hot path (prep + pack + sky-sum kernel) over the whole
sky with all inputs already resident in HBM.  With N > 1 the baselines are sharded (one process per GPU, the reference's pp.key='bl'
model, scripts/run_prisim.py:1775-1791; groups of 256 baselines dealt round-robin so that every rank gets its share of the long ones),
each rank writes snapshot t into slot t of its shard of the visibility cube, and the RCCL all-gather
of the cube (issued per snapshot on a second HIP stream so that it overlaps the next snapshot's compute) is
INSIDE the timed region.  The total workload is fixed as N grows ("scaling": "strong").

No torch anywhere: the launcher only provides RANK / WORLD_SIZE / LOCAL_RANK; the out-of-band exchange of the 128-byte RCCL id,
the barriers and the max-over-ranks of the timing go through prisim_amd.rendezvous (loopback sockets), all GPU work through
libprisim_hip.so.  If the RCCL communicator cannot be built at N > 1, or its 1 MiB self-test all-gather does not deliver every
rank's pattern, the run FAILS before the timed region (exit 3, RCCL's own warnings on stderr): there is no host-side gather.
At N > 1 the line also carries `gather` (bytes per peer, per-snapshot ms on the communication stream, the exposed tail after the last
sky-sum, GB/s per xGMI link against 153), the slowest / fastest rank's kernel ms and `value_n1_equiv` = value / N.

Rank 0 prints ONE JSON line.  Beside the contract's keys it carries (N = 1 only, all outside the timed region):
  roofline / roofline_hbm   the dominant kernel against the VALU issue roofline (the binding one) and the HBM figure BASELINE words
  cpu_baseline              C/OpenMP port of interferometry.py:6332-6340 on this box's host cores (bounded sample) + parity check
  cpu_baseline_ref          the reference FORMULATION itself: numpy restatement, one process, slabbed like :6348-6376 (bounded sample)
  cpu_baseline_ref_xN       the reference's parallel model: N such processes side by side over baseline chunks (mpirun, run_prisim.py:1749-1791)
  e2e                       the same snapshots through InterferometerArray.observe(), host geometry and sky staging included
  e2e_host                  the same with every snapshot landing in (pinned) HOST memory: downloads overlapped with the next sky-sum
  delay_ps                  delay power spectra of the K resident snapshots (interferometry.py:8114-8134 + delay_spectrum.py:3992):
                            device time, FFT count, achieved algorithmic GB/s against the 8 TB/s HBM roofline
  other_kernels             min / median of 5 timed launches each of the fp64 kernel (headline sky), the fused gradient kernels, and -- on config 3
                            with its nside-128 diffuse half -- the packed fp32 taper kernel and the grouped fp64 taper kernel: the kernels the
                            headline workload itself does not run
  shard_estimate            rank 0's share of this workload at N = 1, 2, 4, 8 measured on this GPU (queued snapshots, wall clock): the kernel-side
                            strong-scaling efficiency a multi-GPU run starts from
  config2                   BASELINE config 2 (HERA-19 x 256 ch x nside-16 diffuse, fp64): device time of one snapshot, min / median of 20
"""
import argparse
import hashlib
import json
import os
import sys
import time

import numpy as NP

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from prisim_amd import _abi, launch, rendezvous, sharding, workloads as W   # noqa: E402

FLOPS_PER_TERM = 10.0       # rotate (4 mul + 2 add) + accumulate (2 mul + 2 add) = 6 VALU slots
# Taper contract: the amplitude rides on the step factor, rho = r * q, which is still ONE complex multiply (6 flop) + the
# accumulate (4 flop); what the taper adds is advancing the ratio, rho_{k+1} = rho_k * h: 2 real multiplies per term = 12 flop, 8 VALU slots.
FLOPS_PER_TERM_TAPER = 12.0
PEAK_TFLOPS = {'f32': 157.3, 'f64': 78.6}     # MI355X_MICROARCH.md chip table: vector FP32 157.3 TF; FP64 = half
# What the instruction the kernels are built from actually sustains on this chip: back-to-back v_pk_fma_f32 on every SIMD,
# tools/microbench_valu.hip (profiles/r01_microbench_valu.txt): 126.6 TFLOP/s at 4 waves per SIMD (the chip does not hold the 2.4 GHz of the datasheet figure
# under a full-width FMA stream); fp64 v_fma_f64 64.4.  Printed beside the datasheet fraction, never instead of it.
MEASURED_PEAK_TFLOPS = {'f32': 126.6, 'f64': 64.4}
HBM_PEAK_GBS = 8000.0
XGMI_LINK_GBS = 153.0          # MI355X_MICROARCH.md: 7 point-to-point links of ~153 GB/s per GPU


def shard_range(nbl, world, rank):
    """CONTIGUOUS equal-size baseline blocks (the reference's chunks) -- kept for tools/shard_balance.py, which measures what they cost:
    the product shards through prisim_amd.sharding (groups of baselines dealt round-robin)."""
    per = (nbl + world - 1) // world
    lo = min(rank * per, nbl)
    hi = min(lo + per, nbl)
    return per, lo, hi


def shard_baselines_contiguous(bl, world, rank):
    per, lo, hi = shard_range(bl.shape[0], world, rank)
    mine = bl[lo:hi]
    if mine.shape[0] < per:
        pad = NP.repeat(bl[-1:], per - mine.shape[0], axis=0)
        mine = NP.vstack((mine, pad))
    return mine, hi - lo


def shard_baselines(bl, world, rank):
    """This rank's baselines, padded to the common shard size, and how many of them are real (prisim_amd.sharding: groups of 256
    baselines dealt round-robin, so that every rank gets its share of the long -- more expensive -- baselines)."""
    mine, idx, n_real = sharding.shard_rows(bl, world, rank)
    return mine, n_real


def csrc_hash():
    """Identity of the kernel sources a profile was taken with: sha1 over prisim_amd/csrc (names + contents).  tools/profile_round.sh
    stores it beside the counters; a PMC summary of other sources is not quoted as this build's traffic."""
    h = hashlib.sha1()
    d = os.path.join(ROOT, 'prisim_amd', 'csrc')
    for name in sorted(os.listdir(d)):
        if name.endswith(('.hip', '.cpp', '.h')):
            h.update(name.encode())
            with open(os.path.join(d, name), 'rb') as f:
                h.update(f.read())
    return h.hexdigest()


def host_cpu_info():
    """The host this process may run on: CPU model, logical CPUs, the affinity set, the physical cores inside it, any cgroup quota."""
    info = {'cpu_model': None, 'nproc': os.cpu_count()}
    cores = {}
    try:
        cur = {}
        with open('/proc/cpuinfo') as f:
            for line in f:
                if ':' in line:
                    k, v = [x.strip() for x in line.split(':', 1)]
                    cur[k] = v
                elif cur:
                    if info['cpu_model'] is None:
                        info['cpu_model'] = cur.get('model name')
                    cores[int(cur.get('processor', -1))] = (cur.get('physical id', '0'), cur.get('core id', cur.get('processor')))
                    cur = {}
            if cur:
                cores[int(cur.get('processor', -1))] = (cur.get('physical id', '0'), cur.get('core id', cur.get('processor')))
                if info['cpu_model'] is None:
                    info['cpu_model'] = cur.get('model name')
    except Exception:
        pass
    try:
        aff = sorted(os.sched_getaffinity(0))
    except Exception:
        aff = list(range(os.cpu_count() or 1))
    info['affinity_cpus'] = len(aff)
    phys = {cores[c] for c in aff if c in cores}
    info['physical_cores'] = len(phys) if phys else len(aff)
    info['sockets'] = len({p[0] for p in phys}) if phys else None
    quota = None
    try:
        with open('/sys/fs/cgroup/cpu.max') as f:
            q, per = f.read().split()
            if q != 'max':
                quota = float(q) / float(per)
    except Exception:
        pass
    info['cgroup_cpu_quota'] = quota
    usable = info['physical_cores']
    if quota:
        usable = max(1, min(usable, int(quota)))
    info['usable_cores'] = usable
    return info


def cpu_baseline(cfg, pbflux_sample_fn, target_seconds=8.0):
    """Time the C oracle (oracle/skyvis_oracle.c, the checker) on a bounded baseline sample of the same
    workload, on ALL the physical cores this process may use on this box (one OpenMP thread per core).  The sample is sized from a
    calibration pass so that the leg takes about target_seconds.  Reported baseline only -- never the thing measured above."""
    from oracle import c_oracle as CO
    CO.use_native_build()          # -march=native for THIS host's cores (falls back to the shipped portable build)
    cpu = host_cpu_info()
    threads = max(1, cpu['usable_cores'])
    bl, ch, sky = cfg['baselines'], cfg['channels'], cfg['sky']
    nsrc, nchan = sky['dircos'].shape[0], ch.size
    pb = pbflux_sample_fn()
    zen = NP.array([0.0, 0.0, 1.0])
    fw = sky['fwhm_deg'] if cfg['taper'] else None
    ncal = min(bl.shape[0], threads)
    cal_sel = NP.linspace(0, bl.shape[0] - 1, ncal).astype(int)
    CO.skyvis(bl[cal_sel], ch, sky['dircos'], pb, zen, fwhm_deg=fw, nthreads=threads)     # warm-up: threads, page faults
    t0 = time.perf_counter()
    CO.skyvis(bl[cal_sel], ch, sky['dircos'], pb, zen, fwhm_deg=fw, nthreads=threads)     # calibration: one baseline per thread
    t_cal = max(time.perf_counter() - t0, 1e-4)
    rounds = int(max(1, min(target_seconds / t_cal, bl.shape[0] // max(ncal, 1))))
    nbl_s = min(bl.shape[0], ncal * rounds)
    stride = max(1, bl.shape[0] // nbl_s)
    bls = NP.ascontiguousarray(bl[::stride][:nbl_s])
    t0 = time.perf_counter()
    ref = CO.skyvis(bls, ch, sky['dircos'], pb, zen, fwhm_deg=fw, nthreads=threads)
    dt = time.perf_counter() - t0
    terms = float(bls.shape[0]) * nchan * nsrc
    return {'value': terms / dt, 'unit': 'terms/s', 'cores': threads, 'threads': threads, 'kind': 'port',
            'cpu_model': cpu['cpu_model'], 'nproc': cpu['nproc'], 'physical_cores': cpu['physical_cores'], 'sockets': cpu['sockets'],
            'affinity_cpus': cpu['affinity_cpus'], 'cgroup_cpu_quota': cpu['cgroup_cpu_quota'],
            'sample': '%d of %d baselines (every %d-th) x %d ch x %d src = %.3g terms in %.1f s on %d threads (one per physical core this '
                      'process may use), C/OpenMP libm-sincos port of interferometry.py:6332-6340, %s'
                      % (bls.shape[0], bl.shape[0], stride, nchan, nsrc, terms, dt, threads, CO.flavour)}, bls, ref, stride


def cpu_baseline_ref_xn(cfg, pb, t_one_baseline, max_ranks=32, target_seconds=8.0, slab_bytes=128 << 20):
    """The reference's own parallel model on this box: N independent single-threaded numpy processes, each simulating a contiguous chunk
    of baselines with the statements of interferometry.py:6320-6376 (oracle/ref_rank.py; `mpirun -n N run_prisim.py`,
    scripts/run_prisim.py:1749-1791, README.rst:93-99).  N = the physical cores this process may use, capped at max_ranks (each rank
    holds a slab_bytes phase-matrix slab plus three temporaries of that size); the ranks start together on a go-file and the rate is
    all their terms over the SLOWEST rank's time, as a job's wall clock would be."""
    import subprocess
    import tempfile
    import shutil
    cpu = host_cpu_info()
    n = int(max(1, min(max_ranks, cpu['usable_cores'])))
    bl, ch, sky = cfg['baselines'], cfg['channels'], cfg['sky']
    nsrc, nchan = sky['dircos'].shape[0], ch.size
    per = int(max(1, min(16, target_seconds / max(t_one_baseline, 1e-3))))       # baselines per rank
    sel = NP.linspace(0, bl.shape[0] - 1, n * per).astype(int)
    tmp = tempfile.mkdtemp(prefix='prisim_refxn_')
    try:
        inputs = os.path.join(tmp, 'inputs.npz')
        arrs = dict(bl=bl[sel], ch=ch, dircos=sky['dircos'], pb=pb, pc=NP.array([0.0, 0.0, 1.0]))
        if cfg['taper']:
            arrs['fwhm'] = sky['fwhm_deg']
        NP.savez(inputs, **arrs)
        prefix = os.path.join(tmp, 'run')
        env = dict(os.environ, OMP_NUM_THREADS='1', OPENBLAS_NUM_THREADS='1', MKL_NUM_THREADS='1')
        procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, 'oracle', 'ref_rank.py'), inputs, str(r), str(n), prefix, str(slab_bytes)],
                                  env=env, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE) for r in range(n)]
        t_wait = time.time()
        while not all(os.path.exists('%s.ready%d' % (prefix, r)) for r in range(n)):
            if time.time() - t_wait > 120.0 or any(p.poll() not in (None, 0) for p in procs):
                for p_ in procs:
                    if p_.poll() is None:
                        p_.kill()
                raise RuntimeError('reference ranks did not start')
            time.sleep(0.01)
        t0 = time.perf_counter()
        open(prefix + '.go', 'w').close()
        for p_ in procs:
            try:
                p_.wait(timeout=300)
            except subprocess.TimeoutExpired:
                for q in procs:
                    if q.poll() is None:
                        q.kill()
                raise RuntimeError('a reference rank did not finish within 300 s')
        wall = time.perf_counter() - t0
        if any(p_.returncode != 0 for p_ in procs):
            raise RuntimeError('a reference rank failed: ' + (procs[0].stderr.read().decode()[-300:] if procs[0].stderr else ''))
        vis = NP.zeros((sel.size, nchan), dtype=NP.complex128)
        secs = []
        for r in range(n):
            with NP.load('%s.rank%d.npz' % (prefix, r)) as f:
                vis[int(f['lo']):int(f['hi'])] = f['vis']
                secs.append(float(f['seconds']))
        terms = float(sel.size) * nchan * nsrc
        return {'value': terms / max(secs), 'unit': 'terms/s', 'cores': n, 'ranks': n, 'kind': 'reference-formulation x N ranks (oracle/ref_rank.py)',
                'cpu_model': cpu['cpu_model'], 'nproc': cpu['nproc'], 'physical_cores': cpu['physical_cores'],
                'slowest_rank_s': max(secs), 'fastest_rank_s': min(secs), 'wall_incl_startup_s': wall,
                'sample': '%d ranks x %d baselines x %d ch x %d src = %.3g terms; rate = all terms / slowest rank (%.1f s); one numpy process '
                          'per rank, %d MiB source slabs (interferometry.py:6348-6376), the reference\'s mpirun model (run_prisim.py:1749-1791)'
                          % (n, per, nchan, nsrc, terms, max(secs), slab_bytes >> 20)}, sel, vis
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def cpu_baseline_reference_formulation(cfg, pb, target_seconds=8.0):
    """The reference's own formulation -- oracle/skyvis_oracle.py, the line-by-line numpy restatement of interferometry.py:6320-6376
    (nsrc x nbl x nchan phase matrix, exp, multiply, sum over sources; slabbed over sources when it does not fit, :6348-6376) --
    in ONE process, which is what one PRISim MPI rank executes.  Bounded sample: a few baselines of the same workload."""
    from oracle import skyvis_oracle as O
    bl, ch, sky = cfg['baselines'], cfg['channels'], cfg['sky']
    nsrc, nchan = sky['dircos'].shape[0], ch.size
    zen = NP.array([0.0, 0.0, 1.0])
    fw = sky['fwhm_deg'] if cfg['taper'] else None
    sel = NP.linspace(0, bl.shape[0] - 1, 4).astype(int)
    t0 = time.perf_counter()
    O.skyvis(bl[sel[:1]], ch, sky['dircos'], pb, zen, fwhm_deg=fw)                  # one baseline: sizes the sample
    t1 = time.perf_counter() - t0
    nb = int(max(2, min(64, target_seconds / max(t1, 1e-3))))
    sel = NP.linspace(0, bl.shape[0] - 1, nb).astype(int)
    t0 = time.perf_counter()
    ref = O.skyvis(bl[sel], ch, sky['dircos'], pb, zen, fwhm_deg=fw)
    dt = time.perf_counter() - t0
    terms = float(nb) * nchan * nsrc
    return {'value': terms / dt, 'unit': 'terms/s', 'cores': 1, 'kind': 'reference-formulation (numpy restatement, oracle/skyvis_oracle.py)',
            'sample': '%d of %d baselines x %d ch x %d src = %.3g terms in %.1f s, one process, the statements of '
                      'interferometry.py:6320-6343 (source slabs as :6348-6376 when the phase matrix does not fit)' % (nb, bl.shape[0], nchan, nsrc, terms, dt),
            'seconds_per_baseline': dt / nb}, sel, ref


def profiled_traffic(kernel_tag):
    """HBM bytes per launch of the dominant kernel from a committed rocprofv3 PMC summary (profiles/*/pmc_summary.json: separate
    --pmc passes of this same command, 2*FETCH_SIZE + WRITE_SIZE in KiB with the gfx950 FETCH correction of MI355X_MICROARCH.md)
    -- only if that summary was taken with THIS build's kernel sources (csrc_hash), else None: counters cannot be read inside
    an ordinary run, and a stale figure is worse than none."""
    import glob
    mine = csrc_hash()
    best = None
    for path in sorted(glob.glob(os.path.join(ROOT, 'profiles', '*', 'pmc_summary.json'))):
        try:
            with open(path) as f:
                d = json.load(f)
            if d.get('_csrc_hash') != mine:
                continue
            if kernel_tag in d.get('_kernel', {}).get('Kernel_Name', '') and 'FETCH_SIZE' in d and 'WRITE_SIZE' in d:
                best = ((2.0 * d['FETCH_SIZE']['mean_per_launch'] + d['WRITE_SIZE']['mean_per_launch']) * 1024.0,
                        os.path.relpath(path, ROOT))
        except Exception:
            pass
    return best


def other_kernels(ctx, cfg, zen, nlaunch=5):
    """Driver-run figures for the sky-sum kernels the headline workload does not exercise (same array, same context; every figure is the
    MINIMUM and the MEDIAN of `nlaunch` timed launches after one warm-up): the fp64 kernel on the headline sky, the fused gradient
    kernels, and -- on BASELINE config 3 as worded, 1e4 point sources + nside=128 diffuse (108 048 sources above the horizon) -- the packed
    fp32 taper kernel (every diffuse configuration's kernel) and the grouped fp64 taper kernel (the reference's default precision on such a
    sky, interferometry.py:6332-6335), each against both contracts (10 flop per term, and the taper's 12)."""
    bl, ch, sky = cfg['baselines'], cfg['channels'], cfg['sky']
    res = {'launches_per_figure': nlaunch}

    def timed(prec, want_grad=False, n=nlaunch):
        ctx.compute(precision=prec, want_grad=want_grad, slot=0)          # warm-up (first use of a kernel, table uploads)
        ctx.sync()
        ks, cs, tm = [], [], None
        for rep in range(n):
            ctx.compute(precision=prec, want_grad=want_grad, slot=0)
            ctx.sync()
            tm = ctx.timing()
            ks.append(tm['last_kernel_ms'])
            cs.append(tm['last_compute_ms'])
        return {'kernel_ms_min': min(ks), 'kernel_ms_median': float(NP.median(ks)), 'compute_ms_min': min(cs),
                'compute_ms_median': float(NP.median(cs))}, tm

    def frac(terms, ms, flops, key):
        return terms * flops / (ms * 1e-3) / 1e12 / PEAK_TFLOPS[key]

    p32, _ = timed(_abi.PRISIM_FP32)
    p64, tm = timed(_abi.PRISIM_FP64)
    terms = float(tm['last_terms'])
    res['fp64'] = dict(p64, kernel='k_skyvis_rec<double,%d>' % tm['last_chan_tile'], terms_per_s=terms / (p64['kernel_ms_min'] * 1e-3),
                       roofline_frac=frac(terms, p64['kernel_ms_min'], FLOPS_PER_TERM, 'f64'),
                       roofline_frac_median=frac(terms, p64['kernel_ms_median'], FLOPS_PER_TERM, 'f64'), workload=cfg['name'])
    g64, _ = timed(_abi.PRISIM_FP64, want_grad=True)
    g32, _ = timed(_abi.PRISIM_FP32, want_grad=True)
    res['gradient'] = {'what': 'visibility + baseline gradient (interferometry.py:6330-6343) in one fused pass; ms per snapshot incl. pack/prep; '
                               'roofline against 16 flop per term (10 + 6 for the three extra accumulations)',
                       'fp64_ms': g64['compute_ms_min'], 'fp64_ms_median': g64['compute_ms_median'],
                       'fp64_over_plain_pass': g64['compute_ms_min'] / p64['compute_ms_min'],
                       'fp64_roofline_frac_16flop': frac(terms, g64['kernel_ms_min'], 16.0, 'f64'),
                       'fp32_ms': g32['compute_ms_min'], 'fp32_ms_median': g32['compute_ms_median'],
                       'fp32_over_plain_pass': g32['compute_ms_min'] / p32['compute_ms_min'],
                       'fp32_roofline_frac_16flop': frac(terms, g32['kernel_ms_min'], 16.0, 'f32'),
                       'kernels': 'k_skyvis_grad_f64 (MFMA 4x4x4) / k_skyvis_grad_f32pk'}
    cfg_d = W.config3(with_diffuse=True)
    sky_d = cfg_d['sky']
    ctx.set_sky_analytic(sky_d['dircos'], sky_d['flux_ref'], sky_d['spindex'], sky_d['ref_freq'], _abi.PRISIM_BEAM_AIRY, cfg_d['diameter'], zen, zen,
                         fwhm_deg=sky_d['fwhm_deg'])
    t32, tm = timed(_abi.PRISIM_FP32)
    terms = float(tm['last_terms'])
    res['fp32_taper'] = dict(t32, kernel='k_skyvis_rec_f32pk<%d,taper>' % tm['last_chan_tile'], terms_per_s=terms / (t32['kernel_ms_min'] * 1e-3),
                             grouped_recurrence=bool(tm['last_taper_group']),
                             roofline_frac_vs_no_taper_contract=frac(terms, t32['kernel_ms_min'], FLOPS_PER_TERM, 'f32'),
                             roofline_frac_vs_taper_contract=frac(terms, t32['kernel_ms_min'], FLOPS_PER_TERM_TAPER, 'f32'),
                             roofline_frac_vs_taper_contract_median=frac(terms, t32['kernel_ms_median'], FLOPS_PER_TERM_TAPER, 'f32'),
                             taper_split_runs=tm.get('last_taper_split', 0), taper_uncorrected_groups=tm.get('last_split_uncorrected_groups', 0),
                             workload=cfg_d['name'] + ' (taper on)')
    t64, tm = timed(_abi.PRISIM_FP64)
    res['fp64_taper'] = dict(t64, kernel='k_skyvis_taper_f64<%d> (grouped single-chain form)' % tm['last_chan_tile'],
                             terms_per_s=terms / (t64['kernel_ms_min'] * 1e-3),
                             roofline_frac_vs_no_taper_contract=frac(terms, t64['kernel_ms_min'], FLOPS_PER_TERM, 'f64'),
                             roofline_frac_vs_taper_contract=frac(terms, t64['kernel_ms_min'], FLOPS_PER_TERM_TAPER, 'f64'),
                             roofline_frac_vs_no_taper_contract_median=frac(terms, t64['kernel_ms_median'], FLOPS_PER_TERM, 'f64'),
                             workload=cfg_d['name'] + ' (taper on), fp64')
    return res


def config2_step(nlaunch=20, nbatch=64):
    """BASELINE config 2 as worded (HERA-19, 256 channels, nside-16 diffuse, fp64, taper).
    `single`: device time of one snapshot's compute (prep + pack + kernel + partial reduce), minimum and median of `nlaunch` launches --
    a 6.6e7-term problem: launch-bound.  `batch`: the same array over `nbatch` LSTs of a drift scan as ONE call of
    prisim_hip_observe_catalog -- geometry of all snapshots on the device, one beam launch, one packing launch, ONE sky-sum launch
    (work item = snapshot x baseline wave x channel tile x source split) and ONE reduction; per snapshot: the sky-sum kernel, the
    sum + reduction on the compute stream ("compute"), and the wall clock of the whole call with the queue drained at both ends."""
    from prisim_amd import geometry as GEOM
    cfg = W.config2()
    zen = NP.array([0.0, 0.0, 1.0])
    sky = cfg['sky']
    dev = int(os.environ.get('PRISIM_BENCH_DEVICE', os.environ.get('LOCAL_RANK', '0')))
    out = {'workload': cfg['name']}
    with _abi.Context(dev) as c2:
        c2.set_array(cfg['baselines'], cfg['channels'], nt_max=1)
        c2.set_sky_analytic(sky['dircos'], sky['flux_ref'], sky['spindex'], sky['ref_freq'], _abi.PRISIM_BEAM_AIRY, cfg['diameter'], zen, zen,
                            fwhm_deg=sky['fwhm_deg'])
        c2.compute(precision=_abi.PRISIM_FP64)
        c2.sync()
        ks, cs, tm = [], [], None
        for rep in range(nlaunch):
            c2.compute(precision=_abi.PRISIM_FP64)
            c2.sync()
            tm = c2.timing()
            ks.append(tm['last_kernel_ms'])
            cs.append(tm['last_compute_ms'])
        terms = float(tm['last_terms'])
        out.update({'terms': terms, 'chan_tile': tm['last_chan_tile'], 'nsplit': tm['last_nsplit'],
                    'kernel_us_min': 1e3 * min(ks), 'kernel_us_median': 1e3 * float(NP.median(ks)),
                    'compute_us_min': 1e3 * min(cs), 'compute_us_median': 1e3 * float(NP.median(cs)),
                    'roofline_frac_10flop': terms * FLOPS_PER_TERM / (min(ks) * 1e-3) / 1e12 / PEAK_TFLOPS['f64'],
                    'roofline_frac_10flop_whole_compute': terms * FLOPS_PER_TERM / (min(cs) * 1e-3) / 1e12 / PEAK_TFLOPS['f64'],
                    'terms_per_s_whole_compute': terms / (min(cs) * 1e-3)})
    lat, lst0 = -30.7224, 30.0
    hadec = GEOM.altaz2hadec(sky['altaz'], lat, units='degrees')
    radec = NP.stack(((lst0 - hadec[:, 0]) % 360.0, hadec[:, 1]), axis=1)
    lsts = lst0 + 0.25 * NP.arange(nbatch)
    with _abi.Context(dev) as c2:
        c2.set_array(cfg['baselines'], cfg['channels'], nt_max=nbatch)
        c2.set_catalog(radec, 'radec', flux_ref=sky['flux_ref'], spindex=sky['spindex'], ref_freq_hz=sky['ref_freq'], fwhm_deg=sky['fwhm_deg'])
        obs = c2.make_obs(lat, beam_kind=_abi.PRISIM_BEAM_AIRY, diameter_m=cfg['diameter'])
        def batch_case(ls, reps):
            recs = []
            for rep in range(reps + 1):
                c2.sync()
                c2.timing(reset=True)
                t0 = time.perf_counter()
                counts = c2.observe_catalog(obs, ls, zen, precision=_abi.PRISIM_FP64)
                c2.sync()
                wall = time.perf_counter() - t0
                tm = c2.timing()
                nb = len(ls)
                bterms = float(cfg['baselines'].shape[0]) * cfg['channels'].size * float(NP.sum(counts))
                # (the library keeps the compute time -- sky-sum + reduction -- of the LAST launch only: with several launches per call and no
                # source splits there is no reduction pass and compute = kernel; with splits the figure is not available)
                kern_ms = tm['sum_kernel_ms']
                comp_ms = tm['last_compute_ms'] if tm['n_kernel'] == 1 else (kern_ms if tm['last_nsplit'] == 1 else float('nan'))
                if rep > 0:                                       # the first call carries the allocations
                    recs.append({'snapshots': nb, 'snapshots_per_launch': int(tm['last_batch_snapshots']), 'terms': bterms, 'chan_tile': tm['last_chan_tile'],
                                 'nsplit': tm['last_nsplit'], 'launches': int(tm['n_kernel']),
                                 'kernel_us_per_snapshot': 1e3 * kern_ms / nb, 'compute_us_per_snapshot': 1e3 * comp_ms / nb,
                                 'call_us_per_snapshot': 1e6 * wall / nb,
                                 'roofline_frac_10flop': bterms * FLOPS_PER_TERM / (kern_ms * 1e-3) / 1e12 / PEAK_TFLOPS['f64'],
                                 'roofline_frac_10flop_whole_compute': bterms * FLOPS_PER_TERM / (comp_ms * 1e-3) / 1e12 / PEAK_TFLOPS['f64'],
                                 'roofline_frac_10flop_whole_call': bterms * FLOPS_PER_TERM / wall / 1e12 / PEAK_TFLOPS['f64']})
            mid = sorted(recs, key=lambda r: r['call_us_per_snapshot'])[len(recs) // 2]       # the median call; its own kernel / compute figures
            mid = dict(mid, passes=len(recs),
                       whole_compute_spread=spread([r['roofline_frac_10flop_whole_compute'] for r in recs]),
                       whole_call_spread=spread([r['roofline_frac_10flop_whole_call'] for r in recs]),
                       what='median of %d calls (by whole-call time) after one that carries the allocations; whole_compute = sky-sum + reduction by '
                            'hipEvents on the compute stream; whole_call = wall clock of prisim_hip_observe_catalog (geometry, beam x flux, packing, '
                            'sky-sum, reduction), queue drained at both ends' % len(recs))
            return mid
        out['batch'] = batch_case(lsts, 5)
        # the same chunk with an external HEALPix beam (what HERA-sized runs use): gather + column maximum + 10 ** (.) x flux of all 64 skies
        # in four launches, then the same sky-sum launch
        from prisim_amd import primary_beams as PBM
        bfreq = NP.linspace(float(cfg['channels'][0]) - 5e6, float(cfg['channels'][-1]) + 5e6, 21)
        c2.set_external_beam(W.synthetic_healpix_beam(32, bfreq), PBM.spectral_interp_matrix(bfreq, cfg['channels'], kind='cubic', chromatic=True, select_freq=None))
        obs_x = c2.make_obs(lat, use_external_beam=True)
        bx = None
        for rep in range(4):
            c2.sync()
            c2.timing(reset=True)
            t0 = time.perf_counter()
            c2.observe_catalog(obs_x, lsts, zen, precision=_abi.PRISIM_FP64)
            c2.sync()
            wall = time.perf_counter() - t0
            tm = c2.timing()
            if rep > 0 and (bx is None or wall < bx[0]):
                bx = (wall, int(tm['last_batch_snapshots']))
        out['batch_external_beam'] = {'call_us_per_snapshot': 1e6 * bx[0] / nbatch, 'snapshots_per_launch': bx[1]}
    # a long run: 1024 LSTs in one call = four launches of 256 snapshots (no source splits at that length, the clock is up)
    nlong = 1024
    with _abi.Context(dev) as c2:
        c2.set_array(cfg['baselines'], cfg['channels'], nt_max=nlong)
        c2.set_catalog(radec, 'radec', flux_ref=sky['flux_ref'], spindex=sky['spindex'], ref_freq_hz=sky['ref_freq'], fwhm_deg=sky['fwhm_deg'])
        obs = c2.make_obs(lat, beam_kind=_abi.PRISIM_BEAM_AIRY, diameter_m=cfg['diameter'])
        out['batch_long'] = batch_case(lst0 + 0.05 * NP.arange(nlong), 3)
        # the other two modes of interferometry.py:6320-6343 through the same launch, 256 LSTs in one call: an fp32 request (memsave; served by
        # the fp64 launch on arrays this small -- include/prisim_hip.h) and visibilities + baseline gradients (16-flop contract)
        def mode_case(prec, grad, flop):
            ls = lst0 + 0.05 * NP.arange(256)
            recs = []
            for rep in range(4):
                c2.sync()
                c2.timing(reset=True)
                t0 = time.perf_counter()
                counts = c2.observe_catalog(obs, ls, zen, precision=prec, want_grad=grad)
                c2.sync()
                wall = time.perf_counter() - t0
                tm = c2.timing()
                bterms = float(cfg['baselines'].shape[0]) * cfg['channels'].size * float(NP.sum(counts))
                if rep > 0:
                    recs.append({'snapshots_per_launch': int(tm['last_batch_snapshots']), 'launches': int(tm['n_kernel']), 'chan_tile': tm['last_chan_tile'],
                                 'nsplit': tm['last_nsplit'], 'kernel_us_per_snapshot': 1e3 * tm['sum_kernel_ms'] / ls.size,
                                 'call_us_per_snapshot': 1e6 * wall / ls.size, 'flop_per_term_fp64_contract': flop,
                                 'roofline_frac_kernel': bterms * flop / (tm['sum_kernel_ms'] * 1e-3) / 1e12 / PEAK_TFLOPS['f64'],
                                 'roofline_frac_whole_call': bterms * flop / wall / 1e12 / PEAK_TFLOPS['f64']})
            mid = sorted(recs, key=lambda r: r['call_us_per_snapshot'])[len(recs) // 2]
            return dict(mid, passes=len(recs), call_us_spread=spread([r['call_us_per_snapshot'] for r in recs]))
        out['batch_modes'] = {'fp32_request': dict(mode_case(_abi.PRISIM_FP32, False, FLOPS_PER_TERM), arithmetic='fp64 (the batched launch serves fp32 requests of small arrays)'),
                              'gradient': mode_case(_abi.PRISIM_FP64, True, 16.0),
                              'what': '256 LSTs of config 2 in one call; median of 3 calls after a warm-up; fractions of the fp64 vector peak'}
    # ... and through the class: InterferometerArray.observe() per snapshot / observe_batch() on the (RA, Dec) sky model, 64 snapshots queued
    # (the second 64 of an instance: its catalogue, streams and buffers resident; `fresh` = the first 64, which carry those once); median of 3
    cls = {}
    for mode in ('observe', 'batch'):
        rs = [product_loop_case(2, 1, nbatch, False, mode, True, device=dev, reps=1, passes=2) for _ in range(3)]
        r = sorted(rs, key=lambda x: x['wall_ms_per_snapshot_resident'])[1]
        cls[mode] = {'us_per_snapshot': 1e3 * r['wall_ms_per_snapshot_resident'], 'fresh_us_per_snapshot': 1e3 * r['wall_ms_per_snapshot'],
                     'host_us_per_snapshot': 1e3 * r['host_ms_per_snapshot'], 'snapshots_per_launch': r['snapshots_per_launch'],
                     'us_per_snapshot_spread': spread([1e3 * x['wall_ms_per_snapshot_resident'] for x in rs])}
    # ... and the other two modes of the sum through the class (memsave = an fp32 request, gradient_mode='baseline'): second pass of a
    # resident instance, median of 3 instances
    from prisim_amd import interferometry as RI
    skymod = radec_skymodel(cfg, lat, 30.0)
    tel = {'id': 'hera', 'shape': 'dish', 'size': 14.0, 'ocoords': 'altaz', 'orientation': NP.array([[90.0, 270.0]]), 'groundplane': None}
    def class_mode(mode, kw):
        vals = []
        for rep in range(3):
            ia = RI.InterferometerArray(['b%d' % i for i in range(cfg['baselines'].shape[0])], cfg['baselines'], cfg['channels'], telescope=tel,
                                        latitude=lat, skycoords='radec', pointing_coords='hadec', device=dev)
            ia.reserve(2 * nbatch)
            times = [(2455000.0 + j * 60.0 / 86400.0, 30.0 + 0.25 * j) for j in range(2 * nbatch)]
            for ps in range(2):
                ia._ctx.sync()
                t0 = time.perf_counter()
                if mode == 'observe':
                    for j in range(ps * nbatch, (ps + 1) * nbatch):
                        ia.observe(times[j], {'Tnet': 100.0}, NP.ones(cfg['channels'].size), NP.array([0.0, lat]), skymod, 60.0, **kw)
                else:
                    ia.observe_batch(times[ps * nbatch:(ps + 1) * nbatch], {'Tnet': 100.0}, NP.ones(cfg['channels'].size), NP.array([0.0, lat]), skymod, 60.0, **kw)
                ia._ctx.sync()
                wall = time.perf_counter() - t0
            per_launch = int(ia._ctx.timing().get('last_batch_snapshots', 1))
            ia.close()
            vals.append(1e6 * wall / nbatch)
        return {'us_per_snapshot': float(NP.median(vals)), 'us_per_snapshot_spread': spread(vals), 'snapshots_per_launch': per_launch}
    for name, kw in (('memsave', {'memsave': True}), ('gradient', {'gradient_mode': 'baseline'})):
        for mode in ('observe', 'batch'):
            cls['%s_%s' % (mode, name)] = class_mode(mode, kw)
    out['through_class'] = cls
    return out


def shard_estimate(cfg, zen, prec, device, ranks=(1, 2, 4, 8), nqueue=6):
    """What one rank of an N-GPU run of this workload would spend per snapshot before any communication, measured on THIS GPU: rank 0's
    share of the baselines (prisim_amd/sharding.py), `nqueue` snapshots queued back to back, wall clock per snapshot (sky-sum + pack +
    partial-cube reduction), against the N = 1 figure of the same loop divided by N.  The strong-scaling curve of the driver's multi-GPU
    run cannot be better than this; the all-gather is overlapped on top."""
    bl, ch, sky = cfg['baselines'], cfg['channels'], cfg['sky']
    res = {}
    with _abi.Context(device) as c:
        for n in ranks:
            mine = shard_baselines(bl, n, 0)[0]
            c.set_array(mine, ch, nt_max=1)
            c.set_sky_analytic(sky['dircos'], sky['flux_ref'], sky['spindex'], sky['ref_freq'], _abi.PRISIM_BEAM_AIRY, cfg['diameter'], zen, zen,
                               fwhm_deg=(sky['fwhm_deg'] if cfg['taper'] else None))
            for rep in range(3):                           # clock ramp, first-use costs
                c.compute(precision=prec)
            c.sync()
            best = None
            for rep in range(3):
                t0 = time.perf_counter()
                for k in range(nqueue):
                    c.compute(precision=prec)
                c.sync()
                dt = (time.perf_counter() - t0) / nqueue * 1e3
                best = dt if best is None else min(best, dt)
            tm = c.timing()
            res[str(n)] = {'nbl': int(mine.shape[0]), 'ms_per_snapshot': best, 'nsplit': tm['last_nsplit'], 'chan_tile': tm['last_chan_tile']}
    t1 = res[str(ranks[0])]['ms_per_snapshot'] * ranks[0]
    for n in ranks:
        res[str(n)]['over_ideal'] = res[str(n)]['ms_per_snapshot'] / (t1 / n)
        res[str(n)]['kernel_side_efficiency'] = (t1 / n) / res[str(n)]['ms_per_snapshot']
    res['what'] = ('rank 0 of N, measured on one GPU: wall ms per snapshot of %d queued back to back; ideal = the N = 1 figure / N; no communication in it' % nqueue)
    return res


def _smi_sample():
    """(sclk MHz, package W) from one rocm-smi call, None where it cannot be read."""
    import re
    import subprocess
    try:
        out = subprocess.run(['rocm-smi', '--showclocks', '--showpower', '-d', '0'], capture_output=True, text=True, timeout=20).stdout
    except Exception:
        return None, None
    m = re.search(r'sclk clock level:\s*\d+:\s*\((\d+)Mhz\)', out)
    w = re.search(r'Package Power \(W\):\s*([0-9.]+)', out)
    return (int(m.group(1)) if m else None), (float(w.group(1)) if w else None)


def power_clock(ctx, cfg, zen, prec):
    """What the chip does under the dominant kernel: ~2 s of back-to-back launches on the workload's sky with rocm-smi sampled in the middle,
    then the same launch on a sky whose sources all sit at the phase centre (same kernel, grid and instruction stream; every phasor 1, every
    rotation the identity).  The second holds the spec clock at less power and finishes sooner: the kernel is limited by data-dependent
    switching power (profiles/r02_power_data_probe.txt).  The caller restores the workload's sky afterwards."""
    import threading
    sky = cfg['sky']
    res = {}
    for name, dc in (('workload_sky', sky['dircos']), ('constant_operands', NP.repeat(zen[None, :], sky['dircos'].shape[0], axis=0))):
        ctx.set_sky_analytic(dc, sky['flux_ref'], sky['spindex'], sky['ref_freq'], _abi.PRISIM_BEAM_AIRY, cfg['diameter'], zen, zen,
                             fwhm_deg=(sky['fwhm_deg'] if cfg['taper'] else None))
        for i in range(10):
            ctx.compute(precision=prec, slot=0)
        ctx.sync()
        ctx.timing(reset=True)
        sample = []
        th = threading.Thread(target=lambda: sample.append(_smi_sample()))
        n = 40
        for i in range(n):
            ctx.compute(precision=prec, slot=0)
            if i == 4:
                th.start()                  # the queue is ~2 s deep by now: the sample falls inside the run
        ctx.sync()
        th.join()
        tm = ctx.timing()
        res[name] = {'kernel_ms': tm['sum_kernel_ms'] / max(1, tm['n_kernel']), 'launches': n,
                     'sclk_mhz': sample[0][0] if sample else None, 'package_w': sample[0][1] if sample else None}
    res['what'] = ('dominant kernel back to back; constant_operands = every source at the phase centre (same instruction stream, operands that '
                   'never change)')
    return res


def radec_skymodel(cfg, lat, lst0):
    """The workload's sky (defined in the local frame at LST lst0) as a (RA, Dec) sky model -- what prisim_amd.driver.build_skymodel hands
    to observe()."""
    from prisim_amd import geometry as GEOM, skymodel as SM
    sky = cfg['sky']
    n = sky['dircos'].shape[0]
    hadec = GEOM.altaz2hadec(sky['altaz'], lat, units='degrees')
    radec = NP.stack(((lst0 - hadec[:, 0]) % 360.0, hadec[:, 1]), axis=1)
    return SM.SkyModel(location=radec, flux_ref=sky['flux_ref'], spindex=sky['spindex'], ref_freq=sky['ref_freq'],
                       src_shape=(NP.stack((sky['fwhm_deg'], sky['fwhm_deg'], NP.zeros(n)), axis=1) if cfg['taper'] else None),
                       epoch=None).freeze()  # (laid out in the local frame: coordinates of date; frozen as driver.run freezes its model)


def e2e_observe(cfg, n_snap, device, memsave, to_host=False, batch=False):
    """The same workload through the reference's entry point for the path, InterferometerArray.observe() (interferometry.py:5874), on a
    (RA, Dec) sky model with the LST advancing from snapshot to snapshot -- a drift scan as scripts/run_prisim.py:2165-2207 runs it.
    Per snapshot: the sky geometry (hadec -> altaz -> direction cosines of every source, horizon cut), flux spectra, the fused beam x
    flux, prep, pack and the sky-sum; since round 5 all of it on the device, from the sky model kept resident there (round 4 timed an
    alt-az sky here, which never paid the per-snapshot transform).  batch: observe_batch() -- all snapshots in one call.
    Returns terms/s over n_snap snapshots, wall clock around the loop + final sync."""
    from prisim_amd import interferometry as RI
    bl, ch = cfg['baselines'], cfg['channels']
    lat, lst0, dlst = -30.7224, 40.0, 10.7 * 360.0 * 1.00273790935 / 86400.0
    skymod = radec_skymodel(cfg, lat, lst0)
    tsys, bp, pc = {'Tnet': 100.0}, NP.ones(ch.size), [0.0, lat]

    def when(j):
        return (2457000.5 + j * 10.7 / 86400.0, lst0 + j * dlst)

    ia = RI.InterferometerArray(['b%d' % i for i in range(bl.shape[0])], bl, ch, telescope={'id': 'hera', 'orientation': [90.0, 270.0], 'ocoords': 'altaz'},
                                latitude=lat, skycoords='radec', pointing_coords='hadec', device=device)
    ia.reserve(n_snap + 1, host_staging=to_host)
    ia.observe(when(0), tsys, bp, pc, skymod, 10.7, memsave=memsave)     # warm-up snapshot: catalogue upload, allocations, first-launch costs
    ia._ctx.sync()
    t0 = time.perf_counter()
    if batch:
        ia.observe_batch([when(j) for j in range(1, n_snap + 1)], tsys, bp, pc, skymod, 10.7, memsave=memsave)
    else:
        for j in range(1, n_snap + 1):
            ia.observe(when(j), tsys, bp, pc, skymod, 10.7, memsave=memsave)
    if to_host:
        cube = ia.skyvis_freq_snapshots()      # (n_acc, nbl, nchan) in page-locked host memory: waits for the last snapshot's copy only
        assert cube.shape == (n_snap + 1, bl.shape[0], ch.size)
    ia._ctx.sync()
    dt = time.perf_counter() - t0
    terms = float(bl.shape[0]) * ch.size * float(sum(int(e.size) for e in ia.obs_catalog_indices[1:]))
    res = {'value': terms / dt, 'unit': 'terms/s', 'snapshots': n_snap, 'ms_per_snapshot': dt / n_snap * 1e3, 'skycoords': 'radec',
           'sources_first_last': [int(ia.obs_catalog_indices[1].size), int(ia.obs_catalog_indices[-1].size)],
           'catalogue_resident': type(ia.obs_catalog_indices[-1]).__name__ == '_CatalogROI'}
    if to_host:
        res['staged'] = bool(ia._stage and ia._host_cube is not None)
        res['host_bytes_per_snapshot'] = int(bl.shape[0]) * int(ch.size) * (8 if memsave else 16)
        res['path'] = ('InterferometerArray.observe() with reserve(host_staging=True): as e2e, plus every snapshot copied into a page-locked host '
                       'cube on a copy stream under the next snapshot\'s sky-sum; the clock stops when the snapshot-major cube (skyvis_freq_snapshots) is '
                       'readable on the host')
    else:
        res['path'] = ('InterferometerArray.%s on a (RA, Dec) sky model: geometry + ROI + beam x flux on the device from the resident catalogue, '
                       'sky-sum, cube left resident on the device' % ('observe_batch()' if batch else 'observe()'))
    ia._ctx.close()
    return res


def product_loop_case(cfgno, nranks, n_acc, memsave, mode, catalog, device=0, reps=2, passes=1):
    """Rank 0's share of a BASELINE configuration through the PRODUCT loop: InterferometerArray.observe() / observe_batch()
    on a (RA, Dec) sky model as prisim_amd.driver.run drives it, wall per snapshot with the queue kept full (one synchronisation at the end),
    beside the kernel-only figure (the same shard's compute() with the last sky resident, queued back to back).  catalog False =
    PRISIM_CATALOG=0: every snapshot's sky formed on the host and uploaded (rounds 1-4).  The second repetition is reported.  passes > 1: the
    same instance goes on observing (n_acc more snapshots per pass, the queue drained in between) -- `wall_ms_per_snapshot` stays the FIRST
    pass of a fresh instance (catalogue upload, every first allocation, pinned buffers, streams), `wall_ms_per_snapshot_resident` is the last
    pass: what a snapshot costs once the run's state is resident."""
    from prisim_amd import geometry as GEOM, interferometry as RI
    os.environ['PRISIM_CATALOG'] = '1' if catalog else '0'
    try:
        if cfgno == 4:
            cfg = W.config4(n_acc=n_acc)
            tel = {'id': 'custom', 'shape': 'delta', 'size': 1.0, 'ocoords': 'altaz', 'orientation': NP.array([[90.0, 270.0]]), 'groundplane': None}
        else:
            cfg = W.config2() if cfgno == 2 else W.config3(with_diffuse=False)
            cfg.update(latitude=-30.7224, t_acc=(60.0 if cfgno == 2 else 10.7))
            tel = {'id': 'hera', 'shape': 'dish', 'size': 14.0, 'ocoords': 'altaz', 'orientation': NP.array([[90.0, 270.0]]), 'groundplane': None}
        lat, lst0 = cfg['latitude'], 30.0
        skymod = radec_skymodel(cfg, lat, lst0)
        bl = sharding.shard_rows(cfg['baselines'], nranks, 0)[0]
        ch = cfg['channels']
        dlst = cfg['t_acc'] * 360.0 * 1.00273790935 / 86400.0
        out = None
        for rep in range(reps):
            ia = RI.InterferometerArray(['b%d' % i for i in range(bl.shape[0])], bl, ch, telescope=tel, latitude=lat, skycoords='radec',
                                        pointing_coords='hadec', device=device)
            ia.reserve(n_acc * passes)
            if cfg['beam'] == 'external':
                ia.set_external_beam(cfg['beam_table'], cfg['beam_freqs'])
            lsts = lst0 + NP.arange(n_acc * passes) * dlst
            times = [(2455000.0 + j * cfg['t_acc'] / 86400.0, float(lsts[j])) for j in range(n_acc * passes)]
            tsys, bp, pc = {'Tnet': 100.0}, NP.ones(ch.size), NP.array([0.0, lat])
            walls = []
            for ps in range(passes):
                ia._ctx.sync()
                ia._ctx.timing(reset=True)
                t0 = time.perf_counter()
                if mode == 'observe':
                    for j in range(ps * n_acc, (ps + 1) * n_acc):
                        ia.observe(times[j], tsys, bp, pc, skymod, cfg['t_acc'], memsave=memsave)
                else:
                    ia.observe_batch(times[ps * n_acc:(ps + 1) * n_acc], tsys, bp, pc, skymod, cfg['t_acc'], memsave=memsave)
                if ps == 0:
                    t_host = time.perf_counter() - t0          # the host is done queueing
                ia._ctx.sync()
                walls.append(time.perf_counter() - t0)
                if ps == 0:
                    tm = ia._ctx.timing()
            wall = walls[0]
            nsrc = [int(e.size) for e in ia.obs_catalog_indices]
            prec = _abi.PRISIM_FP32 if memsave else _abi.PRISIM_FP64
            batched = tm.get('last_batch_snapshots', 1)
            pcd = GEOM.altaz2dircos(GEOM.hadec2altaz(pc, lat, units='degrees'), 'degrees').ravel()

            def kernel_only(ps):
                """compute() alone, n_acc times back to back, on the sky of the MIDDLE snapshot of pass ps (the sky drifts: fewer sources and
                more culling later in config 4's scan) -- after every pass, so both figures are taken at the same clock state"""
                if catalog:      # (a batched launch leaves no single current sky either)
                    ia._ctx.set_sky_from_catalog(ia._catalog_obs_cache[1], float(lsts[ps * n_acc + n_acc // 2]), pcd)
                ia._ctx.sync()
                ia._ctx.timing(reset=True)
                t1 = time.perf_counter()
                for j in range(n_acc):
                    ia._ctx.compute(precision=prec, slot=j)
                ia._ctx.sync()
                return time.perf_counter() - t1, ia._ctx.timing()

            wall_k, tmk = kernel_only(0)
            out = {'config': cfgno, 'nranks': nranks, 'shard_baselines': int(bl.shape[0]), 'nchan': int(ch.size), 'n_acc': n_acc,
                   'precision': 'fp32' if memsave else 'fp64', 'mode': mode, 'catalog': bool(catalog),
                   'nsrc_roi_first_last': [nsrc[0], nsrc[-1]], 'wall_ms_per_snapshot': 1e3 * wall / n_acc, 'wall_ms_total': 1e3 * wall,
                   'host_ms_per_snapshot': 1e3 * t_host / n_acc, 'kernel_ms_per_snapshot': tm['sum_kernel_ms'] / max(tm['n_kernel'], 1) / batched,
                   'kernel_only_wall_ms_per_snapshot': 1e3 * wall_k / n_acc,
                   'kernel_only_kernel_ms_per_snapshot': tmk['sum_kernel_ms'] / max(tmk['n_kernel'], 1),
                   'ratio_wall_over_kernel_only_wall': (wall / n_acc) / (wall_k / n_acc), 'chan_tile': tm['last_chan_tile'], 'nsplit': tm['last_nsplit'],
                   'culled_fraction_last': tm['last_culled_fraction'], 'snapshots_per_launch': batched}
            if passes > 1:
                wall_kr = kernel_only(passes - 1)[0]
                out['wall_ms_per_snapshot_resident'] = 1e3 * walls[-1] / n_acc
                out['kernel_only_wall_ms_per_snapshot_resident_sky'] = 1e3 * wall_kr / n_acc
                out['resident_over_kernel_only_wall'] = walls[-1] / wall_kr
                out['first_pass_extra_ms_total'] = 1e3 * ((wall - wall_k) - (walls[-1] - wall_kr))
            ia._ctx.close()
            del ia
        return out
    finally:
        os.environ.pop('PRISIM_CATALOG', None)


def e2e_shard_estimate(device, n_acc=32, ranks=(1, 8)):
    """rank 0's share of BASELINE config 4 (MWA-128T, 768 channels, nside-64 diffuse, external HEALPix beam, drift
    scan) at N = 1 and 8 THROUGH THE PRODUCT LOOP (observe_batch on the (RA, Dec) sky model) -- wall per snapshot, queued -- beside the
    kernel-only figure.  `marginal` takes the difference of a 3 n_acc and an n_acc run: what one more snapshot costs once the catalogue
    is resident and the clock is up (the first snapshots of any run carry the catalogue upload, the first allocations and ~30 ms of
    clock ramp)."""
    res = {'what': 'rank 0 of N on one GPU, config 4, fp32; wall ms per snapshot through InterferometerArray.observe_batch (driver.run\'s loop) '
                   'against compute() alone on the sky of the pass\'s middle snapshot, queued back to back after all passes (same clock '
                   'state); wall / over_kernel_only = the first n_acc snapshots of a FRESH instance (catalogue upload, first allocations, the first '
                   'snapshot\'s preparation that nothing overlaps, clock ramp; the context\'s streams exist since its creation), resident = the next n_acc of '
                   'the same instance; first_pass_extra_ms_total = what the fresh pass spends beyond the resident one, once per run; '
                   'marginal = (wall(3 n) - wall(n)) / 2n of two fresh instances', 'n_acc': n_acc}
    for n in ranks:
        runs = [product_loop_case(4, n, n_acc, True, 'batch', True, device=device, reps=1, passes=2) for _ in range(3)]
        order = sorted(range(3), key=lambda i: runs[i]['resident_over_kernel_only_wall'])
        a = runs[order[1]]                                            # the pass whose resident ratio is the median: every figure below is its
        b = product_loop_case(4, n, 3 * n_acc, True, 'batch', True, device=device, reps=1)
        marg = (b['wall_ms_total'] - a['wall_ms_total']) / (2 * n_acc)
        res[str(n)] = {'shard_baselines': a['shard_baselines'], 'wall_ms_per_snapshot': a['wall_ms_per_snapshot'], 'marginal_ms_per_snapshot': marg,
                       'kernel_only_wall_ms_per_snapshot': a['kernel_only_wall_ms_per_snapshot'], 'kernel_ms_per_snapshot': a['kernel_ms_per_snapshot'],
                       'over_kernel_only': a['wall_ms_per_snapshot'] / a['kernel_only_wall_ms_per_snapshot'],
                       'resident_ms_per_snapshot': a['wall_ms_per_snapshot_resident'], 'resident_over_kernel_only': a['resident_over_kernel_only_wall'],
                       'kernel_only_resident_sky_ms_per_snapshot': a['kernel_only_wall_ms_per_snapshot_resident_sky'],
                       'first_pass_extra_ms_total': a['first_pass_extra_ms_total'],
                       'marginal_over_kernel_only': marg / a['kernel_only_wall_ms_per_snapshot'], 'host_ms_per_snapshot': a['host_ms_per_snapshot'],
                       'culled_fraction': a['culled_fraction_last'], 'nsplit': a['nsplit'],
                       'over_kernel_only_spread': spread([r['wall_ms_per_snapshot'] / r['kernel_only_wall_ms_per_snapshot'] for r in runs]),
                       'resident_over_kernel_only_spread': spread([r['resident_over_kernel_only_wall'] for r in runs])}
    return res


def spread(values):
    """{'median', 'min', 'max', 'n'} of a list of numbers: what a quoted secondary figure carries."""
    v = sorted(float(x) for x in values)
    return {'median': float(NP.median(v)), 'min': v[0], 'max': v[-1], 'n': len(v)}


def gather_rehearsal(cfg, zen, prec, device, nsnap=12):
    """librccl in the driver-observed N = 1 record.  A 1-rank communicator (its own unique id), the self-test, then
    `nsnap` snapshots of rank 0's 1/8 shard of the headline workload with prisim_hip_allgather_slot_async on the highest-priority
    stream under the next sky-sum -- RCCL's kernels sharing the CUs with the sky-sum grid on real hardware.  It cannot measure xGMI (one
    rank: the gather is a device-local copy through RCCL's own kernel), it does prove load, ABI and stream interplay of the box's RCCL."""
    bl, ch, sky = cfg['baselines'], cfg['channels'], cfg['sky']
    mine = shard_baselines(bl, 8, 0)[0]
    c64 = prec == _abi.PRISIM_FP32
    res = {'what': '1-rank RCCL communicator on the bench box: rank 0\'s 1/8 shard, ncclAllGather of every snapshot on the priority stream under the '
                   'next sky-sum (no xGMI in it)', 'shard_baselines': int(mine.shape[0]), 'snapshots': nsnap, 'wire_dtype': 'complex64' if c64 else 'complex128'}
    res['librccl'] = _abi.Context.comm_version()
    with _abi.Context(device) as c:
        c.set_array(mine, ch, nt_max=nsnap)
        c.set_sky_analytic(sky['dircos'], sky['flux_ref'], sky['spindex'], sky['ref_freq'], _abi.PRISIM_BEAM_AIRY, cfg['diameter'], zen, zen,
                           fwhm_deg=(sky['fwhm_deg'] if cfg['taper'] else None))
        c.comm_init(_abi.Context.comm_unique_id(), 1, 0)
        c.comm_selftest(1 << 20)
        res['selftest'] = 'ok'
        # a shard map as an N > 1 run sets one (prisim_hip_set_shard_map): here the 1-rank map is a fixed PERMUTATION of the rows with the last
        # three rows declared padding, so the staging buffer and the un-deal kernel run behind librccl's gather on real hardware
        nbl_tot = int(mine.shape[0]) - 3
        perm = NP.random.default_rng(20).permutation(nbl_tot).astype(NP.int64)
        smap = NP.full((1, mine.shape[0]), -1, dtype=NP.int64)
        smap[0, :nbl_tot] = perm
        c.set_shard_map(smap, nbl_tot)
        res['order'] = 'a fixed permutation of the shard\'s rows as the 1-rank shard map (3 padding rows dropped): staging + un-deal kernel behind every gather'

        def loop(gather):
            for t in range(3):
                c.compute(precision=prec, slot=t)
            c.sync()
            c.timing(reset=True)
            c.comm_stats(reset=True)
            t0 = time.perf_counter()
            for t in range(nsnap):
                c.compute(precision=prec, slot=t)
                if gather:
                    c.allgather_slot_async(t, complex64=c64)
            c.sync()
            dt = (time.perf_counter() - t0) / nsnap * 1e3
            tm = c.timing()
            return dt, tm['sum_kernel_ms'] / max(tm['n_kernel'], 1)
        passes = [(loop(False), loop(True)) for _ in range(3)]          # three pairs, the quoted ratio is their median
        ratios = [b[0] / a[0] for a, b in passes]
        mid = sorted(range(3), key=lambda i: ratios[i])[1]
        (wall0, kern0), (wall1, kern1) = passes[mid]
        st = c.comm_stats()
        res.update({'compute_ms_without_gathers': wall0, 'compute_ms_with_gathers': wall1, 'kernel_ms_without_gathers': kern0, 'kernel_ms_with_gathers': kern1,
                    'compute_slowdown': wall1 / wall0, 'compute_slowdown_spread': spread(ratios),
                    'per_snapshot_ms': st['sum_gather_ms'] / max(st['n_gathers'], 1), 'max_gather_ms': st['max_gather_ms'],
                    'exposed_ms': st['last_gather_after_compute_ms'], 'bytes_per_gather': st['bytes_per_peer'], 'gathers_measured': st['n_gathers'],
                    'comm_stream_priority': st['stream_priority'], 'lowest_priority': st['stream_priority_lowest'],
                    'undeal_ms_per_snapshot': st['sum_undeal_ms'] / max(st['n_gathers'], 1)})
        cs = c.gathered_checksum(nsnap, complex64=c64)
        g = c.get_gathered(1, 1)[0]                                        # (nbl_tot, nchan) in the map's order
        v = c.get_vis(slot=0, complex64=c64)
        res['gather_ok'] = bool(NP.isfinite(cs) and g.shape[0] == nbl_tot and NP.array_equal(g[perm], v[:nbl_tot]))
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=5)
    ap.add_argument('--warmup', type=int, default=1)
    ap.add_argument('--precision', choices=('fp32', 'fp64'), default='fp32')
    ap.add_argument('--nsrc', type=int, default=10000)
    ap.add_argument('--workload', choices=('cfg3', 'cfg3d', 'cfg5'), default='cfg3',
                    help='cfg3: the headline workload (BASELINE config 3, 1e4 point sources); cfg3d: config 3 with its nside=128 diffuse half '
                         '(source-shape taper on); cfg5: one LST of config 5 (nside=256 diffuse sky, taper on)')
    ap.add_argument('--no-cpu-baseline', action='store_true', help='skip the CPU baselines, the e2e and the delay-stage extras (profiling runs)')
    ap.add_argument('--extras', choices=('core', 'all'), default='core',
                    help="N = 1 only.  core (default): the CPU baselines and the extras the documents quote -- config 2, the config-4 shard through the "
                         "product loop, the 1-rank RCCL rehearsal, the delay stage, observe() end to end -- each a median of 3 passes; all: also "
                         "the reference formulation x N ranks, the power / clock probe, the other kernels and the kernel-only shard estimate "
                         "(kept under profiles/ once per round: they add a minute)")
    ap.add_argument('--chan-tile', type=int, default=0, help='A/B hook: force the channel tile of the recurrence kernels (0 = planned)')
    ap.add_argument('--want-grad', action='store_true', help='profiling hook: every step also computes the baseline gradient (fused kernel)')
    args = ap.parse_args()

    if 'WORLD_SIZE' not in os.environ and args.gpus > 1:
        # launched bare: become the launcher.  N rank processes are started BEFORE anything here touches the GPU (this process never
        # does), with RANK / LOCAL_RANK / WORLD_SIZE set; rank 0's JSON line goes to the inherited stdout; the exit code is the job's.
        script = os.environ.get('PRISIM_BENCH_SELF', os.path.abspath(__file__))
        sys.exit(launch.spawn_ranks(args.gpus, [sys.executable, script] + sys.argv[1:]))

    # Libraries loaded later (RCCL prints a version banner on its first communicator) write to fd 1: keep the real stdout for
    # the ONE JSON line of the contract and send everything else to stderr.
    sys.stdout.flush()
    json_out = os.fdopen(os.dup(1), 'w')
    os.dup2(2, 1)

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        args.gpus = world                                      # the launcher's world size is the truth
    os.environ.setdefault('NCCL_SOCKET_IFNAME', 'lo')          # single node: RCCL bootstraps over loopback (one node is all this bench does)
    if world > 1:
        os.environ.setdefault('NCCL_DEBUG', 'WARN')            # a communicator that cannot be built says why, on stderr
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')   # dmabuf IPC (what RCCL's peer-to-peer set-up needs on this driver); set before HIP loads, as prisim_amd.launch does
    rdzv = rendezvous.Rendezvous(rank, world)                  # sockets only: before any GPU call

    if args.workload == 'cfg5':
        cfg = W.config5(n_acc=1)
    else:
        cfg = W.config3(nsrc=args.nsrc, with_diffuse=(args.workload == 'cfg3d'))
    bl_all, ch, sky = cfg['baselines'], cfg['channels'], cfg['sky']
    nbl_total, nchan, nsrc = bl_all.shape[0], ch.size, sky['dircos'].shape[0]
    bl_mine, n_real = shard_baselines(bl_all, world, rank)
    prec = _abi.PRISIM_FP32 if args.precision == 'fp32' else _abi.PRISIM_FP64
    dtype = 'f32' if args.precision == 'fp32' else 'f64'
    K, Wm = max(1, args.steps), max(0, args.warmup)
    zen = NP.array([0.0, 0.0, 1.0])

    # PRISIM_BENCH_DEVICE: rehearsal hook (several ranks on one GPU to exercise the multi-process flow on a 1-GPU box)
    device = int(os.environ.get('PRISIM_BENCH_DEVICE', local_rank))
    ctx = _abi.Context(device)
    ctx.set_array(bl_mine, ch, nt_max=K)
    if args.chan_tile:
        ctx.set_tuning(args.chan_tile, 0, 0)
    # inputs resident in HBM before the timed region: directions, reference fluxes, spectral indices;
    # beam x flux (nsrc x nchan) is built on the device (fused Airy beam x power law)
    ctx.set_sky_analytic(sky['dircos'], sky['flux_ref'], sky['spindex'], sky['ref_freq'], _abi.PRISIM_BEAM_AIRY,
                         cfg['diameter'], zen, zen, fwhm_deg=(sky['fwhm_deg'] if cfg['taper'] else None))
    c64 = (prec == _abi.PRISIM_FP32)
    if world > 1:
        # RCCL communicator: rank 0 creates the 128-byte id, the rendezvous carries it to the other ranks.  Any failure is fatal.  The whole
        # block runs under one deadline (PRISIM_COMM_TIMEOUT_S, default 120 s; prisim_amd/watchdog.py): ncclCommInitRank and the exchanges
        # around it wait for every rank, and a rank that never arrives must end the job with a message, not with the driver's timeout.
        from prisim_amd import watchdog
        err = None
        uid = b''
        with watchdog.for_context(rank, device, _abi) as deadline:
            if rank == 0:
                try:
                    uid = _abi.Context.comm_unique_id()
                except Exception as exc:
                    err = 'RCCL unique id failed: %r' % (exc,)
            deadline.step('rendezvous: broadcast of the RCCL unique id')
            uid = rdzv.broadcast_bytes(uid)
            if len(uid) == 128:
                try:
                    deadline.step('ncclCommInitRank (prisim_hip_comm_init)')
                    ctx.comm_init(uid, world, rank)
                except Exception as exc:
                    err = 'RCCL comm_init failed: %r' % (exc,)
            elif err is None:
                err = 'no RCCL unique id received'
            if err is None:
                try:
                    deadline.step('self-test all-gather (prisim_hip_comm_selftest)')
                    ctx.comm_selftest(1 << 20)     # 1 MiB per rank through ncclAllGather, every rank's pattern verified on the host
                except Exception as exc:
                    err = 'RCCL self-test failed: %r' % (exc,)
            if err is None:
                try:
                    # every rank's rows in the global numbering (-1 = padding): each gathered snapshot is put into the reference's baseline
                    # order by a copy kernel behind the gather, on the communication stream (prisim_hip_set_shard_map)
                    smap = NP.full((world, bl_mine.shape[0]), -1, dtype=NP.int64)
                    for r in range(world):
                        idx_r = sharding.shard_index(nbl_total, world, r)
                        smap[r, :idx_r.size] = idx_r
                    ctx.set_shard_map(smap, nbl_total)
                except Exception as exc:
                    err = 'shard map refused: %r' % (exc,)
            deadline.step('rendezvous: exchange of the set-up outcomes')
            errs = [e for e in rdzv.allgather(err) if e]
        if errs:
            sys.stderr.write('rank %d: %s\n' % (rank, '; '.join(errs)))
            rdzv.close()
            sys.exit(3)        # no host-side gather fallback: a value printed without RCCL would not be the metric

    wg = bool(args.want_grad)
    for i in range(Wm):
        ctx.compute(precision=prec, slot=i % K, want_grad=wg)
        if world > 1:
            ctx.allgather_slot_async(i % K, complex64=c64)
    ctx.sync()
    ctx.timing(reset=True)
    if world > 1:
        ctx.comm_stats(reset=True)

    rdzv.barrier()
    ctx.sync()
    t0 = time.perf_counter()
    for t in range(K):
        ctx.compute(precision=prec, slot=t, want_grad=wg)
        if world > 1:
            # RCCL all-gather of snapshot t on the communication stream, overlapped with the sky-sum of snapshot t+1;
            # only the last snapshot's exchange is exposed.  Still inside the timed region.
            ctx.allgather_slot_async(t, complex64=c64)
    ctx.sync()
    rdzv.barrier()
    t1 = time.perf_counter()
    elapsed = rdzv.allreduce_max(t1 - t0)

    tm = ctx.timing()
    gather_ok = None
    my_kern_ms = tm['sum_kernel_ms'] / max(1, tm['n_kernel'])
    rank_kern_ms = rdzv.allgather(my_kern_ms)
    # what every rank's planner chose for its shard (a slow N > 1 line then says whether one rank ran another decomposition) and the whole
    # compute() per step beside the kernel alone (prep + pack + kernel + partial-cube reduction)
    rank_plan = rdzv.allgather({'chan_tile': int(tm['last_chan_tile']), 'nsplit': int(tm['last_nsplit']), 'lift_groups': int(tm.get('last_lift_groups', 0)),
                                'nbl_shard': int(bl_mine.shape[0]), 'last_compute_ms': float(tm.get('last_compute_ms', 0.0))})
    gstats = None
    if world > 1:
        cs_ = ctx.comm_stats()
        per_snap = cs_['sum_gather_ms'] / max(1, cs_['n_gathers'])
        mine_g = {'per_snapshot_ms': per_snap, 'max_ms': cs_['max_gather_ms'], 'exposed_ms': cs_['last_gather_after_compute_ms'],
                  'bytes_per_peer': cs_['bytes_per_peer'], 'n': cs_['n_gathers'], 'prio': [cs_['stream_priority'], cs_['stream_priority_lowest']],
                  'undeal_ms': cs_['sum_undeal_ms'] / max(1, cs_['n_gathers'])}
        allg = rdzv.allgather(mine_g)
        worst = max(allg, key=lambda g: g['per_snapshot_ms'])
        gbps = (worst['bytes_per_peer'] / (worst['per_snapshot_ms'] * 1e-3) / 1e9) if worst['per_snapshot_ms'] > 0 else None
        gstats = {'what': 'ncclAllGather of one snapshot per step on the highest-priority communication stream, overlapped with the next sky-sum',
                  'bytes_per_peer': worst['bytes_per_peer'], 'wire_dtype': 'complex64' if c64 else 'complex128',
                  'per_snapshot_ms': worst['per_snapshot_ms'], 'per_snapshot_ms_all_ranks': [g['per_snapshot_ms'] for g in allg],
                  'max_ms': max(g['max_ms'] for g in allg),
                  'exposed_ms': max(g['exposed_ms'] for g in allg),
                  'exposed_ms_what': 'end of the LAST snapshot\'s gather minus end of its sky-sum, slowest rank: the part no compute hides',
                  'GBps_per_link': gbps, 'link_peak_GBps': XGMI_LINK_GBS, 'link_frac': (gbps / XGMI_LINK_GBS) if gbps else None,
                  'GBps_per_link_what': 'one shard to each of N-1 peers over its own xGMI link, while the sky-sum grid occupies the CUs (slowest rank)',
                  'gathers_measured': worst['n'], 'comm_stream_priority': worst['prio'][0], 'lowest_priority': worst['prio'][1],
                  'order': 'global', 'order_what': 'every receiving GPU holds [nt][nbl_total][nchan] in the unsharded array\'s baseline order '
                                                   '(run_prisim.py:2233-2242), put there by a copy kernel behind each gather; its time is part of per_snapshot_ms',
                  'undeal_ms_per_snapshot': max(g['undeal_ms'] for g in allg)}
    if world > 1:
        # every rank must hold the same gathered cube (device checksums agree) AND the rows of this rank's baselines in it -- global
        # order -- must be this rank's own shard, element for element (a plain checksum would not see swapped blocks)
        cs = ctx.gathered_checksum(K, complex64=c64)
        allcs = rdzv.allgather(cs)
        gather_ok = bool(all(abs(c - allcs[0]) <= 1e-9 * max(1.0, abs(allcs[0])) for c in allcs))
        g = ctx.get_gathered(1, world)[0]                                  # snapshot 0: [global baseline][f]
        mine = ctx.get_vis(slot=0, complex64=c64)
        idx_mine = sharding.shard_index(nbl_total, world, rank)
        gather_ok = gather_ok and g.shape[0] == nbl_total
        gather_ok = gather_ok and bool(all(rdzv.allgather(bool(NP.array_equal(g[idx_mine], mine[:n_real])))))

    if rank == 0:
        terms_total = float(nbl_total) * nchan * nsrc * K
        value = terms_total / elapsed
        kern_ms = tm['sum_kernel_ms'] / max(1, tm['n_kernel'])
        # terms one launch of the dominant kernel processes on this rank (padded shard)
        terms_launch = float(bl_mine.shape[0]) * nchan * nsrc
        flops_term = FLOPS_PER_TERM + (6.0 if wg else 0.0)      # the gradient adds 6 flops per term (three more real x complex FMAs)
        ach_tflops = terms_launch * flops_term / (kern_ms * 1e-3) / 1e12
        wp = 4 if prec == _abi.PRISIM_FP32 else 8
        alg_bytes = nsrc * nchan * wp + 24 * nsrc + 24 * bl_mine.shape[0] + 8 * nchan + 2 * 8 * bl_mine.shape[0] * nchan
        ach_gbs = alg_bytes / (kern_ms * 1e-3) / 1e9
        traffic = traffic_src = None
        if world == 1 and nsrc == 10000 and args.workload == 'cfg3' and not wg:
            tr = profiled_traffic('f32pk<64, false>' if dtype == 'f32' else 'k_skyvis_rec<double, 32, false>')
            if tr is not None:
                traffic, traffic_src = tr
        out = {
            'metric': 'visibility-terms/sec', 'value': value, 'unit': 'terms/s', 'n_gpus': world, 'steps': K, 'warmup': Wm,
            'ms_per_step': elapsed / K * 1e3, 'higher_is_better': True, 'scaling': 'strong', 'vs_baseline': None,
            'dtype': dtype, 'data': 'synthetic',
            'config': {'workload': cfg['name'], 'nbl': nbl_total, 'nchan': nchan, 'nsrc': nsrc, 'nt': K,
                       'beam': 'airy D=14 m fused on device',
                       'sharding': 'baselines/%d (groups of %d dealt round-robin) + 1 RCCL all-gather' % (world, sharding.group_size(nbl_total, world)),
                       'kernel': 'recurrence ct=%d nsplit=%d' % (tm['last_chan_tile'], tm['last_nsplit']),
                       'taper_split_runs': tm.get('last_taper_split', 0), 'taper_uncorrected_groups': tm.get('last_split_uncorrected_groups', 0)},
            'roofline': {'bound': 'valu', 'achieved': ach_tflops, 'peak': PEAK_TFLOPS[dtype], 'unit': 'TFLOP/s',
                         'frac': ach_tflops / PEAK_TFLOPS[dtype], 'traffic': traffic, 'traffic_unit': 'bytes/launch',
                         'traffic_source': traffic_src,
                         'frac_vs_measured_peak': ach_tflops / MEASURED_PEAK_TFLOPS[dtype], 'measured_peak': MEASURED_PEAK_TFLOPS[dtype],
                         'measured_peak_what': 'back-to-back v_pk_fma_f32 (fp64: v_fma_f64) on every SIMD, profiles/r01_microbench_valu.txt',
                         'kernel': 'k_skyvis_grad' if wg else 'k_skyvis_rec', 'avg_kernel_ms': kern_ms, 'flops_per_term': flops_term,
                         'terms_per_launch': terms_launch},
            'roofline_hbm': {'bound': 'hbm', 'achieved': ach_gbs, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                             'frac': ach_gbs / HBM_PEAK_GBS, 'traffic': traffic, 'traffic_unit': 'bytes/launch',
                             'algorithmic_bytes_per_launch': alg_bytes},
            'csrc_hash': csrc_hash(),
        }
        if cfg['taper'] and not wg:
            out['roofline']['flops_per_term_taper_contract'] = FLOPS_PER_TERM_TAPER
            out['roofline']['frac_vs_taper_contract'] = terms_launch * FLOPS_PER_TERM_TAPER / (kern_ms * 1e-3) / 1e12 / PEAK_TFLOPS[dtype]
            out['roofline']['frac_what'] = ('frac charges the no-taper 10 flop/term; frac_vs_taper_contract charges 12 (scaled rotation 6 + accumulate 4 + '
                                            'ratio update 2)')
        out['kernel_ms_per_rank'] = {'min': min(rank_kern_ms), 'max': max(rank_kern_ms), 'all': rank_kern_ms}
        out['plan_per_rank'] = rank_plan
        out['value_n1_equiv'] = value / world           # whole-job rate per GPU: what to hold against the N = 1 line
        if world > 1:
            out['gather'] = gstats
            out['gather_ok'] = gather_ok
            out['launcher'] = 'torch.distributed.run env' if 'TORCHELASTIC_RUN_ID' in os.environ else 'prisim_amd.launch'
        if world == 1 and not args.no_cpu_baseline:
            # Everything below is reported beside the contract line, never instead of it: an extra that raises is named in `extras_failed`
            # (and keeps its {'error': ...} entry), so a regression in one of them is visible in the record; `extras_seconds` says what each
            # cost.  --extras core (the default) keeps the whole run inside ~90 s of driver time; --extras all is the once-per-round record.
            failed, seconds = [], {}

            def extra(key, fn, on_error=None):
                t_x = time.perf_counter()
                try:
                    out[key] = fn()
                except Exception as exc:       # the extra is a report, never a reason to lose the bench line
                    failed.append(key)
                    out[key] = dict(on_error or {}, error=repr(exc))
                seconds[key] = round(time.perf_counter() - t_x, 2)
            full = args.extras == 'all'
            state = {'pb_host': None, 't_one': None}

            def x_cpu_baseline():
                state['pb_host'] = ctx.get_pbflux()
                cb, bls, ref, stride = cpu_baseline(cfg, lambda: state['pb_host'])
                # parity spot-check of the timed GPU result against the checker on the same sample
                vis = ctx.get_vis(slot=K - 1)
                gpu = vis[::stride][:bls.shape[0]]
                scale = NP.sum(NP.abs(state['pb_host']), axis=0)[None, :]
                out['parity_max_err_rel_sumflux'] = float(NP.max(NP.abs(gpu - ref) / scale))
                return cb
            extra('cpu_baseline', x_cpu_baseline, {'value': None, 'unit': 'terms/s', 'cores': 0, 'kind': 'port', 'sample': 'failed'})

            def x_cpu_ref():
                cbr, sel, ref2 = cpu_baseline_reference_formulation(cfg, state['pb_host'])
                cbr['parity_of_gpu_vs_this_max_err_rel_sumflux'] = float(NP.max(NP.abs(ctx.get_vis(slot=K - 1)[sel] - ref2)
                                                                                / NP.sum(NP.abs(state['pb_host']), axis=0)[None, :]))
                state['t_one'] = cbr.pop('seconds_per_baseline')
                return cbr
            extra('cpu_baseline_ref', x_cpu_ref, {'value': None, 'unit': 'terms/s', 'cores': 1, 'kind': 'reference-formulation', 'sample': 'failed'})

            def x_cpu_ref_xn():
                if state['t_one'] is None:
                    raise RuntimeError('the one-process leg failed')
                cbx, selx, refx = cpu_baseline_ref_xn(cfg, state['pb_host'], state['t_one'])
                cbx['parity_of_gpu_vs_this_max_err_rel_sumflux'] = float(NP.max(NP.abs(ctx.get_vis(slot=K - 1)[selx] - refx)
                                                                                / NP.sum(NP.abs(state['pb_host']), axis=0)[None, :]))
                return cbx
            if full:
                extra('cpu_baseline_ref_xN', x_cpu_ref_xn, {'value': None, 'unit': 'terms/s', 'cores': 0, 'kind': 'reference-formulation x N', 'sample': 'failed'})

            def x_delay_ps():
                # delay power spectra of the K resident snapshots, one window for all baselines, pad = 1 (run_prisim.py:954, 2284)
                # in K^2 (Mpc/h)^3: abs(.)^2 * jacobian1 * jacobian2 * Jy2K^2 (delay_spectrum.py:3659-3663, 3992) -- redshift, comoving
                # distances, and the beam volume of the Airy pattern on a HEALPix nside-32 grid (evaluated on the device) on the host
                from prisim_amd import delay_spectrum as DSM
                win = NP.blackman(nchan) + 0.01
                pconst = DSM.power_constants(ch, {'id': 'hera'}, freq_wts=win, device=device)
                ms = []
                for rep in range(4):
                    ctx.delay_transform_device(K, bpwts=win, pad=1.0, want_lag=False, want_power=True, power_scale=pconst['factor'])
                    ctx.sync()
                    if rep:
                        ms.append(ctx.timing()['last_delay_ms'])
                tmd = ctx.timing()
                med = float(NP.median(ms))
                nrow = K * bl_mine.shape[0]
                dbytes = float(nrow) * nchan * (16 + 8)            # each visibility read once, each power sample written once
                gbs = dbytes / (med * 1e-3) / 1e9
                return {'device_ms': med, 'device_ms_spread': spread(ms), 'ffts': nrow, 'fft_length_kept': nchan, 'pad': 1.0,
                        'power_scale_K2_Mpc3_per_Jy2Hz2': pconst['factor'], 'z': pconst['z'], 'omega_bw_SrHz': float(pconst['omega_bw'][0]),
                        'rz_los_Mpc_h': pconst['rz_los'], 'drz_los_Mpc_h': pconst['drz_los'], 'cosmology': pconst['cosmology'],
                        'fused_lds_kernel': bool(tmd['last_delay_fused']),
                        'roofline': {'bound': 'hbm', 'achieved': gbs, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': gbs / HBM_PEAK_GBS,
                                     'algorithmic_bytes': dbytes}}
            extra('delay_ps', x_delay_ps, {'device_ms': None})
            if full:
                extra('power_clock', lambda: power_clock(ctx, cfg, zen, prec))

                def x_other():
                    ctx.set_sky_analytic(sky['dircos'], sky['flux_ref'], sky['spindex'], sky['ref_freq'], _abi.PRISIM_BEAM_AIRY,
                                         cfg['diameter'], zen, zen, fwhm_deg=(sky['fwhm_deg'] if cfg['taper'] else None))
                    return other_kernels(ctx, cfg, zen)
                extra('other_kernels', x_other)
            ctx.close()
            extra('config2', config2_step)
            if full:
                extra('shard_estimate', lambda: shard_estimate(cfg, zen, prec, device))
            extra('gather_rehearsal', lambda: gather_rehearsal(cfg, zen, prec, device))
            extra('e2e_shard_estimate', lambda: e2e_shard_estimate(device))
            memsave = prec == _abi.PRISIM_FP32

            def x_e2e(**kw):
                # three fresh instances; the one whose wall per snapshot is the median is reported, with the spread of all three
                rs = [e2e_observe(cfg, 8, device, memsave=memsave, **kw) for _ in range(3)]
                r = sorted(rs, key=lambda x: x['ms_per_snapshot'])[1]
                r['ms_per_snapshot_spread'] = spread([x['ms_per_snapshot'] for x in rs])
                return r

            def x_e2e_plain():
                r = x_e2e()
                r['over_step'] = r['ms_per_snapshot'] / (elapsed / K * 1e3)
                r['over_step_spread'] = {k: (v / (elapsed / K * 1e3) if k != 'n' else v) for k, v in r['ms_per_snapshot_spread'].items()}
                return r
            extra('e2e', x_e2e_plain, {'value': None})
            extra('e2e_batch', lambda: x_e2e(batch=True), {'value': None})

            def x_e2e_host():
                r = x_e2e(to_host=True)
                if out.get('e2e', {}).get('ms_per_snapshot'):
                    r['over_e2e'] = r['ms_per_snapshot'] / out['e2e']['ms_per_snapshot']
                return r
            extra('e2e_host', x_e2e_host, {'value': None})
            out['extras'] = args.extras
            out['extras_failed'] = failed
            out['extras_seconds'] = seconds
        json_out.write(json.dumps(out) + '\n')
        json_out.flush()
    ctx.close()
    rdzv.barrier()
    rdzv.close()


if __name__ == '__main__':
    main()
