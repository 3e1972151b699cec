#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X-native PRISim sky-sum.

Metric (BASELINE.json): visibility-terms/s = nbl * nchan * nsrc * nt / wall, on the synthetic
HERA-350 x 1024-channel x 1e4-source workload (SURVEY.md 8(d) config 3, fp32 with tolerance check).

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one snapshot: one pass of the hot path (prep + pack + sky-sum kernel) over the whole
sky with all inputs already resident in HBM.  With N > 1 the baselines are sharded in contiguous
blocks (one process per GPU, the reference's pp.key='bl' model, scripts/run_prisim.py:1775-1791),
each rank writes snapshot t into slot t of its shard of the visibility cube, and the RCCL all-gather
of the cube (issued per snapshot on a second HIP stream so that it overlaps the next snapshot's compute) is
INSIDE the timed region.  The total workload is
fixed as N grows ("scaling": "strong").  torch is used only for the multi-process rendezvous
(gloo barrier / max-reduce / unique-id broadcast); all GPU work goes through libprisim_hip.so.

Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

import numpy as NP

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from prisim_amd import _abi, workloads as W   # noqa: E402

FLOPS_PER_TERM = 10.0       # rotate (4 mul + 2 add) + accumulate (2 mul + 2 add) = 6 VALU slots
PEAK_TFLOPS = {'f32': 157.3, 'f64': 78.6}     # MI355X_MICROARCH.md chip table: vector FP32 157.3 TF; FP64 = half
HBM_PEAK_GBS = 8000.0


def shard_range(nbl, world, rank):
    """Contiguous equal-size baseline blocks, ceil(nbl/world) each (last ones padded by repeating
    the final baseline so that every rank -- and the all-gather -- has the same shard size)."""
    per = (nbl + world - 1) // world
    lo = min(rank * per, nbl)
    hi = min(lo + per, nbl)
    return per, lo, hi


def shard_baselines(bl, world, rank):
    per, lo, hi = shard_range(bl.shape[0], world, rank)
    mine = bl[lo:hi]
    if mine.shape[0] < per:
        pad = NP.repeat(bl[-1:], per - mine.shape[0], axis=0)
        mine = NP.vstack((mine, pad))
    return mine, hi - lo


def cpu_baseline(cfg, pbflux_sample_fn, target_terms=1.0e10):
    """Time the C oracle (oracle/skyvis_oracle.c, the checker) on a bounded baseline sample of the same
    workload, on this box's host cores.  Reported baseline only -- never the thing measured above."""
    from oracle import c_oracle as CO
    CO.use_native_build()          # -march=native for THIS host's cores (falls back to the shipped portable build)
    threads = max(1, min(16, os.cpu_count() or 1, CO.max_threads()))
    bl, ch, sky = cfg['baselines'], cfg['channels'], cfg['sky']
    nsrc, nchan = sky['dircos'].shape[0], ch.size
    nbl_s = int(max(threads, min(bl.shape[0], target_terms // (nsrc * nchan))))
    stride = max(1, bl.shape[0] // nbl_s)
    bls = NP.ascontiguousarray(bl[::stride][:nbl_s])
    pb = pbflux_sample_fn()
    zen = NP.array([0.0, 0.0, 1.0])
    fw = sky['fwhm_deg'] if cfg['taper'] else None
    CO.skyvis(bls[:threads], ch, sky['dircos'], pb, zen, fwhm_deg=fw, nthreads=threads)     # warm-up
    t0 = time.perf_counter()
    ref = CO.skyvis(bls, ch, sky['dircos'], pb, zen, fwhm_deg=fw, nthreads=threads)
    dt = time.perf_counter() - t0
    terms = float(bls.shape[0]) * nchan * nsrc
    return {'value': terms / dt, 'unit': 'terms/s', 'cores': threads, 'kind': 'port',
            'sample': '%d of %d baselines (every %d-th) x %d ch x %d src = %.3g terms in %.1f s, C/OpenMP libm-sincos port of '
                      'interferometry.py:6332-6340, %s' % (bls.shape[0], bl.shape[0], stride, nchan, nsrc, terms, dt, CO.flavour)}, bls, ref, stride


def profiled_traffic(kernel_tag):
    """HBM bytes per launch of the dominant kernel, from the newest committed rocprofv3 PMC summary
    (profiles/*/pmc_summary.json: separate --pmc passes of this same command, 2*FETCH_SIZE + WRITE_SIZE in KiB with the
    gfx950 FETCH correction of MI355X_MICROARCH.md).  Counters cannot be read inside an ordinary run, so this is the
    committed measurement for the same kernel/workload, or None."""
    import glob
    best = None
    for path in sorted(glob.glob(os.path.join(ROOT, 'profiles', '*', 'pmc_summary.json'))):
        try:
            with open(path) as f:
                d = json.load(f)
            if kernel_tag in d.get('_kernel', {}).get('Kernel_Name', '') and 'FETCH_SIZE' in d and 'WRITE_SIZE' in d:
                best = ((2.0 * d['FETCH_SIZE']['mean_per_launch'] + d['WRITE_SIZE']['mean_per_launch']) * 1024.0,
                        os.path.relpath(path, ROOT))
        except Exception:
            pass
    return best


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
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--chan-tile', type=int, default=0, help='A/B hook: force the channel tile of the recurrence kernels (0 = planned)')
    args = ap.parse_args()

    # Libraries loaded later (RCCL prints a version banner on its first communicator) write to fd 1: keep the real stdout for
    # the ONE JSON line of the contract and send everything else to stderr.
    sys.stdout.flush()
    json_out = os.fdopen(os.dup(1), 'w')
    os.dup2(2, 1)

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit('bench.py --gpus %d must be launched with torch.distributed.run --nproc-per-node %d' % (args.gpus, args.gpus))
        args.gpus = world
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('NCCL_SOCKET_IFNAME', 'lo')      # single node: RCCL bootstraps over loopback
        dist.init_process_group(backend='gloo', rank=rank, world_size=world)

    def barrier():
        if dist is not None:
            dist.barrier()

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
    ctx = _abi.Context(int(os.environ.get('PRISIM_BENCH_DEVICE', local_rank)))
    ctx.set_array(bl_mine, ch, nt_max=K)
    if args.chan_tile:
        ctx.set_tuning(args.chan_tile, 0, 0)
    # inputs resident in HBM before the timed region: directions, reference fluxes, spectral indices;
    # beam x flux (nsrc x nchan) is built on the device (fused Airy beam x power law)
    ctx.set_sky_analytic(sky['dircos'], sky['flux_ref'], sky['spindex'], sky['ref_freq'], _abi.PRISIM_BEAM_AIRY,
                         cfg['diameter'], zen, zen, fwhm_deg=(sky['fwhm_deg'] if cfg['taper'] else None))
    c64 = (prec == _abi.PRISIM_FP32)
    rccl_ok = True
    if world > 1:
        # RCCL communicator: rank 0 creates the 128-byte id, gloo carries it to the other ranks
        try:
            uid = [_abi.Context.comm_unique_id() if rank == 0 else None]
        except Exception as exc:                      # keep the ranks in step even if librccl cannot be loaded
            uid = [None]
            sys.stderr.write('rank %d: RCCL unique id failed: %r\n' % (rank, exc))
        dist.broadcast_object_list(uid, src=0)
        ok = 0
        if uid[0] is not None:
            try:
                ctx.comm_init(uid[0], world, rank)
                ok = 1
            except Exception as exc:
                sys.stderr.write('rank %d: RCCL comm_init failed: %r\n' % (rank, exc))
        import torch
        flag = torch.tensor([ok], dtype=torch.int32)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        rccl_ok = bool(flag.item())

    def host_gather(nt):
        """Fallback when RCCL cannot be initialised: device -> host, gloo all_gather, kept only so that the bench still
        reports a (slow) whole-job number instead of crashing; flagged in the JSON line."""
        import torch
        shard = NP.stack([ctx.get_vis(slot=t, complex64=c64) for t in range(nt)])
        tsr = torch.from_numpy(NP.ascontiguousarray(shard.view(NP.float32 if c64 else NP.float64)))
        parts = [torch.empty_like(tsr) for _ in range(world)]
        dist.all_gather(parts, tsr)
        return parts

    for i in range(Wm):
        ctx.compute(precision=prec, slot=i % K)
        if world > 1 and rccl_ok:
            ctx.allgather_slot_async(i % K, complex64=c64)
    ctx.sync()
    ctx.timing(reset=True)

    barrier()
    ctx.sync()
    t0 = time.perf_counter()
    for t in range(K):
        ctx.compute(precision=prec, slot=t)
        if world > 1 and rccl_ok:
            # RCCL all-gather of snapshot t on the communication stream, overlapped with the sky-sum of snapshot t+1;
            # only the last snapshot's exchange is exposed.  Still inside the timed region.
            ctx.allgather_slot_async(t, complex64=c64)
    ctx.sync()
    if world > 1 and not rccl_ok:
        host_gather(K)
    barrier()
    t1 = time.perf_counter()
    elapsed = t1 - t0
    if dist is not None:
        import torch
        tt = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    tm = ctx.timing()
    gather_ok = None
    if world > 1 and rccl_ok:
        # every rank must hold the same gathered cube: compare device checksums
        cs = ctx.gathered_checksum(K, complex64=c64)
        import torch
        allcs = [None] * world
        dist.all_gather_object(allcs, cs)
        gather_ok = bool(all(abs(c - allcs[0]) <= 1e-9 * max(1.0, abs(allcs[0])) for c in allcs))

    if rank == 0:
        terms_total = float(nbl_total) * nchan * nsrc * K
        value = terms_total / elapsed
        kern_ms = tm['sum_kernel_ms'] / max(1, tm['n_kernel'])
        # terms one launch of the dominant kernel processes on this rank (padded shard)
        terms_launch = float(bl_mine.shape[0]) * nchan * nsrc
        ach_tflops = terms_launch * FLOPS_PER_TERM / (kern_ms * 1e-3) / 1e12
        wp = 4 if prec == _abi.PRISIM_FP32 else 8
        alg_bytes = nsrc * nchan * wp + 24 * nsrc + 24 * bl_mine.shape[0] + 8 * nchan + 2 * 8 * bl_mine.shape[0] * nchan
        ach_gbs = alg_bytes / (kern_ms * 1e-3) / 1e9
        traffic = traffic_src = None
        if world == 1 and nsrc == 10000 and args.workload == 'cfg3':
            tr = profiled_traffic('f32pk<64, false>' if dtype == 'f32' else 'k_skyvis_rec<double, 32, false>')
            if tr is not None:
                traffic, traffic_src = tr
        out = {
            'metric': 'visibility-terms/sec', 'value': value, 'unit': 'terms/s', 'n_gpus': world, 'steps': K, 'warmup': Wm,
            'ms_per_step': elapsed / K * 1e3, 'higher_is_better': True, 'scaling': 'strong', 'vs_baseline': None,
            'dtype': dtype, 'data': 'synthetic',
            'config': {'workload': cfg['name'], 'nbl': nbl_total, 'nchan': nchan, 'nsrc': nsrc, 'nt': K,
                       'beam': 'airy D=14 m fused on device', 'sharding': 'baselines/%d + 1 RCCL all-gather' % world,
                       'kernel': 'recurrence ct=%d nsplit=%d' % (tm['last_chan_tile'], tm['last_nsplit'])},
            'roofline': {'bound': 'valu', 'achieved': ach_tflops, 'peak': PEAK_TFLOPS[dtype], 'unit': 'TFLOP/s',
                         'frac': ach_tflops / PEAK_TFLOPS[dtype], 'traffic': traffic, 'traffic_unit': 'bytes/launch',
                         'traffic_source': traffic_src,
                         'kernel': 'k_skyvis_rec', 'avg_kernel_ms': kern_ms, 'flops_per_term': FLOPS_PER_TERM,
                         'terms_per_launch': terms_launch},
            'roofline_hbm': {'bound': 'hbm', 'achieved': ach_gbs, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                             'frac': ach_gbs / HBM_PEAK_GBS, 'traffic': traffic, 'traffic_unit': 'bytes/launch',
                             'algorithmic_bytes_per_launch': alg_bytes},
        }
        if world > 1:
            out['gather'] = 'rccl-allgather (per snapshot, overlapped)' if rccl_ok else 'host-gloo-fallback (RCCL init failed)'
        if gather_ok is not None:
            out['gather_ok'] = gather_ok
        if world == 1 and not args.no_cpu_baseline:
            try:
                def pb_sample():
                    return ctx.get_pbflux()
                cb, bls, ref, stride = cpu_baseline(cfg, pb_sample)
                out['cpu_baseline'] = cb
                # parity spot-check of the timed GPU result against the checker on the same sample
                vis = ctx.get_vis(slot=K - 1)
                gpu = vis[::stride][:bls.shape[0]]
                scale = NP.sum(NP.abs(pb_sample()), axis=0)[None, :]
                out['parity_max_err_rel_sumflux'] = float(NP.max(NP.abs(gpu - ref) / scale))
            except Exception as exc:   # the baseline is a report, never a reason to lose the bench line
                out['cpu_baseline'] = {'value': None, 'unit': 'terms/s', 'cores': 0, 'kind': 'port', 'sample': 'failed: %r' % (exc,)}
        json_out.write(json.dumps(out) + '\n')
        json_out.flush()
    ctx.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
