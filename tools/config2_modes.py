"""BASELINE config 2 (HERA-19 x 256 ch x nside-16 diffuse) as a run of K LSTs through prisim_hip_observe_catalog in the modes of
interferometry.py:6320-6343 other than the plain fp64 sum: fp32 requests (memsave) and visibilities + baseline gradients -- each through
the batched launch and through the per-snapshot chain it replaces (the A/B switches PRISIM_HIP_BATCH_FP32_AS_FP64 / PRISIM_HIP_WAVE_BATCH_GRAD).
Whole call by wall clock with the queue drained at both ends (median of 5 after a warm-up), kernel time by hipEvents; roofline fractions
against the contract of the arithmetic that ran (fp64: 10 flop per term plain, 16 with the gradient; 78.6 TF).
usage: python tools/config2_modes.py [K ...]"""
import json
import os
import statistics
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as NP

from prisim_amd import _abi, geometry as GEOM, workloads as W

PEAK_F64, PEAK_F32 = 78.6e12, 157.3e12

MODES = [
    # name, precision, want_grad, env
    ('fp64', _abi.PRISIM_FP64, False, {}),
    ('fp32_request_batched_as_fp64', _abi.PRISIM_FP32, False, {}),
    ('fp32_request_per_snapshot_chain', _abi.PRISIM_FP32, False, {'PRISIM_HIP_BATCH_FP32_AS_FP64': '0'}),
    ('grad_batched', _abi.PRISIM_FP64, True, {}),
    ('grad_per_snapshot_chain', _abi.PRISIM_FP64, True, {'PRISIM_HIP_WAVE_BATCH_GRAD': '0'}),
]


def main():
    ks = [int(x) for x in sys.argv[1:]] or [256]
    cfg = W.config2()
    lat, lst0 = -30.7224, 30.0
    sky = cfg['sky']
    hadec = GEOM.altaz2hadec(sky['altaz'], lat, units='degrees')
    radec = NP.stack(((lst0 - hadec[:, 0]) % 360.0, hadec[:, 1]), axis=1)
    zen = NP.array([0.0, 0.0, 1.0])
    nbl, nchan = cfg['baselines'].shape[0], cfg['channels'].size
    for k in ks:
        lsts = lst0 + 0.25 * NP.arange(k)
        for name, prec, grad, env in MODES:
            for key, val in env.items():
                os.environ[key] = val
            try:
                with _abi.Context(0) as ctx:
                    ctx.set_array(cfg['baselines'], cfg['channels'], nt_max=k)
                    ctx.set_catalog(radec, 'radec', flux_ref=sky['flux_ref'], spindex=sky['spindex'], ref_freq_hz=sky['ref_freq'],
                                    fwhm_deg=sky['fwhm_deg'])
                    obs = ctx.make_obs(lat, beam_kind=_abi.PRISIM_BEAM_AIRY, diameter_m=14.0)
                    walls, kern = [], []
                    reps = 6 if (k <= 64 or not env) else 3
                    for rep in range(reps):
                        ctx.sync()
                        ctx.timing(reset=True)
                        t0 = time.perf_counter()
                        counts = ctx.observe_catalog(obs, lsts, zen, precision=prec, want_grad=grad)
                        ctx.sync()
                        wall = time.perf_counter() - t0
                        tm = ctx.timing()
                        if rep > 0:
                            walls.append(wall)
                            kern.append(tm['sum_kernel_ms'] * 1e-3)
                    terms = float(nbl) * nchan * float(NP.sum(counts))
                    flop = 16.0 if grad else 10.0
                    wall, kt = statistics.median(walls), statistics.median(kern)
                    print(json.dumps({'K': k, 'mode': name, 'us_per_snapshot': round(1e6 * wall / k, 2), 'min_max_us': [round(1e6 * min(walls) / k, 2), round(1e6 * max(walls) / k, 2)],
                                      'kernel_us_per_snapshot': round(1e6 * kt / k, 2), 'launches': tm['n_kernel'], 'batch_snapshots': tm['last_batch_snapshots'],
                                      'chan_tile': tm['last_chan_tile'], 'nsplit': tm['last_nsplit'], 'terms_per_s_whole_call': terms / wall,
                                      'fp64_contract_flop_per_term': flop, 'roofline_whole_call_fp64': terms * flop / wall / PEAK_F64,
                                      'roofline_kernel_only_fp64': terms * flop / kt / PEAK_F64}), flush=True)
            finally:
                for key in env:
                    del os.environ[key]


if __name__ == '__main__':
    main()
