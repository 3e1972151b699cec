"""Worker for tests/test_distributed_cpu.py (and its GPU twin): one process per rank, started by prisim_amd.launch (no torch).

Runs the PRODUCT's multi-rank code -- prisim_amd.rendezvous for the out-of-band id exchange / barrier / max-reduce, and
prisim_amd.driver.run(parms, rank=r, world=N) with its baseline sharding (padded last shard), InterferometerArray.observe into
reserved slots, allgather(), per-shard delay_transform() and allgather_lags() -- and checks the gathered cube and the gathered
delay spectra against a world-1 run of the same driver, element for element.  RCCL needs one GPU per rank, so the context is
replaced at the `_abi.Context` seam by a stand-in from tests/fake_context.py whose exchange goes through the product's own socket
rendezvous (host arrays):
  dist_worker.py oracle   no GPU (CPU suite)
  dist_worker.py gpu      real HIP context, both ranks on device 0, host exchange
A second, shorter run with processing.gradient_mode = 'baseline' checks the gathered gradient cube (interferometry.py:8349-8350) and the
HDF5 file rank 0 writes from it against the unsharded run.
"""
import os
import sys

import numpy as NP

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import bench                                                 # noqa: E402
from prisim_amd import _abi, driver, rendezvous              # noqa: E402
import fake_context                                          # noqa: E402


def parms_for_test():
    p = driver.deep_merge(driver.DEFAULTS, {
        'array': {'layout': 'HERA-19', 'redundant': False},                       # 171 baselines: 86 per rank, the last shard is padded
        'telescope': {'id': 'custom', 'latitude': -30.7224},
        'antenna': {'shape': 'delta', 'size': 1.0},
        'bandpass': {'freq': 150e6, 'freq_resolution': 1e6, 'nchan': 16},
        'obsparm': {'n_acc': 3, 't_acc': 600.0, 'obs_mode': 'drift'},
        'pointing': {'lst_init': 1.0, 'drift_init': {'ha': 0.0, 'dec': -30.7224}},
        'skyparm': {'model': 'ptsrc_random', 'n_src': 40, 'seed': 7, 'custom_reffreq': 0.150, 'spindex': -0.8},
        'processing': {'delay_transform': True, 'f_pad': 1.0, 'bpass_shape': 'bhw', 'noise_seed': 20261004},
        'phasing': {'center': [75.0, 40.0], 'coords': 'altaz'},                   # away from the pointing: every rank re-centres its own shard
    })
    return p


def _hdf5_of_sharded_run_equals_unsharded(parms, out_sharded, out_single):
    """Rank 0 of the sharded run writes PRISim's HDF5 file of the WHOLE array (driver.assemble_full_array: its own shard object + the
    gathered cubes, where the reference concatenates part files); it must hold what the unsharded run's file holds, dataset by dataset."""
    import tempfile
    from prisim_amd import hdf5io
    try:
        hdf5io._load()
    except hdf5io.HDF5Unavailable:
        return True
    files = []
    for tag, out in (('sharded', out_sharded), ('single', out_single)):
        p = driver.deep_merge(parms, {'dirstruct': {'rootdir': tempfile.mkdtemp() + '/', 'project': 'p', 'simid': tag},
                                      'save_formats': {'npz': True, 'hdf5': True}})
        driver.save(out, p)
        files.append(os.path.join(p['dirstruct']['rootdir'], 'p', tag, 'simdata', 'simvis.hdf5'))

    def content(fname):
        found = {}
        with hdf5io.File(fname, 'r') as f:
            def walk(group):
                for name in f.list(group or '/'):
                    path = (group + '/' + name) if group else name
                    try:
                        f.list(path)
                        walk(path)
                    except KeyError:
                        found[path] = f.read(path)
            walk('')
        return found
    a, b = content(files[0]), content(files[1])
    if sorted(a) != sorted(b):
        print('HDF5 objects differ:', sorted(set(a) ^ set(b)))
        return False
    for key in a:
        va, vb = a[key], b[key]
        if isinstance(va, str) or isinstance(vb, str):
            same = va == vb
        else:
            va, vb = NP.asarray(va), NP.asarray(vb)
            if va.shape != vb.shape:
                same = False
            elif va.dtype.kind in 'fc':
                scale = float(NP.max(NP.abs(vb))) if vb.size else 0.0
                same = bool(NP.all(NP.isnan(va) == NP.isnan(vb))) and float(NP.max(NP.abs(NP.nan_to_num(va - vb)), initial=0.0)) <= 1e-11 * max(scale, 1e-300)
            else:
                same = bool(NP.array_equal(va, vb))
        if not same:
            print('HDF5 dataset differs:', key)
            return False
    return len(a) > 30


def ref_bl_for_shard(nbl, world, rank):
    """The baselines rank `rank` must hold: its round-robin groups of the whole (length-sorted) list."""
    from prisim_amd import sharding
    bl_all = driver.baseline_info(parms_for_test())[0]
    return bl_all[sharding.shard_index(nbl, world, rank)]


def main():
    mode = sys.argv[1] if len(sys.argv) > 1 else 'oracle'
    rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
    # the product's own rendezvous (what bench.py and the driver's main() use): id broadcast, barrier, reductions
    rdzv = rendezvous.Rendezvous(rank, world)
    uid = rdzv.broadcast_bytes(bytes(range(128)) if rank == 0 else b'')
    assert uid == bytes(range(128))
    assert rdzv.allreduce_max(float(rank + 1)) == float(world) and rdzv.allreduce_min(float(rank + 1)) == 1.0
    assert rdzv.allgather({'rank': rank}) == [{'rank': r} for r in range(world)]
    rdzv.barrier()
    fake_context.RDZV = rdzv                                   # carries the stand-in communicator's data (RCCL's place)
    _abi.Context = fake_context.OracleContext if mode == 'oracle' else fake_context.HostCommContext

    parms = parms_for_test()
    out = driver.run(parms, rank=rank, world=world, device=0, comm_uid=uid, verbose=False)
    from prisim_amd import sharding
    per = sharding.shard_size(171, world)
    assert out['ia'].baselines.shape[0] == per                                  # equal shards (groups dealt round-robin), padded
    assert NP.array_equal(out['ia'].baselines[:sharding.shard_index(171, world, rank).size], ref_bl_for_shard(171, world, rank))
    assert out['skyvis_freq'].shape == (171, 16, 3) and out['skyvis_lag'].shape == (171, 16, 3)   # padding rows dropped
    ref = driver.run(parms, rank=0, world=1, device=0, verbose=False)          # the same driver, unsharded, in this process
    tol = 0.0 if mode == 'oracle' else 1e-11 * float(NP.max(NP.abs(ref['skyvis_freq'])))
    err_v = float(NP.max(NP.abs(out['skyvis_freq'] - ref['skyvis_freq'])))
    err_l = float(NP.max(NP.abs(out['skyvis_lag'] - ref['skyvis_lag'])))
    lag_scale = float(NP.max(NP.abs(ref['skyvis_lag'])))
    ok = err_v <= tol and err_l <= max(tol, 1e-12 * lag_scale) and out['labels'] == ref['labels'] and NP.array_equal(out['bl'], ref['bl'])
    # thermal noise: every shard draws what the unsharded run draws for its baselines; re-centred like the visibilities, gathered with them
    noise_scale = float(NP.max(NP.abs(ref['vis_noise_freq'])))
    err_n = float(NP.max(NP.abs(out['vis_noise_freq'] - ref['vis_noise_freq'])))
    err_s = float(NP.max(NP.abs(out['vis_freq'] - ref['vis_freq'])))
    ok = ok and noise_scale > 0 and err_n <= 1e-12 * noise_scale and err_s <= 1e-12 * noise_scale + tol
    # what the YAML entry point does (driver.main): only rank 0 pulls the gathered cube and spectra to the host
    out_root = driver.run(parms, rank=rank, world=world, device=0, comm_uid=uid, verbose=False, host_copy='root')
    if rank == 0:
        ok = ok and NP.array_equal(out_root['skyvis_freq'], out['skyvis_freq']) and NP.array_equal(out_root['skyvis_lag'], out['skyvis_lag']) \
            and NP.array_equal(out_root['vis_noise_freq'], out['vis_noise_freq'])
    else:
        ok = ok and out_root['skyvis_freq'] is None and out_root['skyvis_lag'] is None and out_root['vis_freq'] is None
    if rank == 0:
        ok = ok and _hdf5_of_sharded_run_equals_unsharded(parms, out_root, ref)
    # baseline gradients of a sharded run: gathered like the visibilities, equal to the unsharded run's, and in rank 0's HDF5 file
    gparms = driver.deep_merge(parms, {'processing': {'gradient_mode': 'baseline', 'delay_transform': False, 'add_noise': False},
                                       'obsparm': {'n_acc': 2}, 'pp': {'gather': 'root'}})       # and only rank 0 receives (gather to root)
    gout = driver.run(gparms, rank=rank, world=world, device=0, comm_uid=uid, verbose=False, host_copy='root')
    if rank == 0:
        gref = driver.run(gparms, rank=0, world=1, device=0, verbose=False)
        g, gr = gout['gradient']['baseline'], gref['gradient']['baseline']
        gscale = float(NP.max(NP.abs(gr)))
        err_g = float(NP.max(NP.abs(g - gr))) if g.shape == gr.shape else float('inf')
        ok = ok and g.shape == (3, 171, 16, 2) and gscale > 0 and err_g <= (0.0 if mode == 'oracle' else 1e-11 * gscale)
        ok = ok and _hdf5_of_sharded_run_equals_unsharded(gparms, gout, gref)
    else:
        ok = ok and gout['gradient'] is None
    all_ok = all(rdzv.allgather(bool(ok)))
    rdzv.barrier()
    rdzv.close()
    if not all_ok:
        print('RANK %d MISMATCH vis %g lag %g' % (rank, err_v, err_l))
        sys.exit(1)
    print('RANK %d OK per=%d err_vis=%.2e err_lag=%.2e' % (rank, per, err_v, err_l))


if __name__ == '__main__':
    main()
