"""Profiling target for the delay stage (tools/profile_round.sh with PROFILE_CMD): config-5-sized rows (HERA-350 x 1024 channels, a few
LSTs) through prisim_hip_delay_transform_device -- the fused LDS FFT kernel -- a few times; prints one JSON line with the device
time and the algorithmic bytes so that tools/summarize_pmc.py can put the counters beside them."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as NP

from prisim_amd import _abi, workloads as W

nt = int(sys.argv[1]) if len(sys.argv) > 1 else 4
cfg = W.config4(n_acc=nt) if (len(sys.argv) > 2 and sys.argv[2] == 'cfg4') else W.config5(n_acc=nt)
bl, ch = cfg['baselines'], cfg['channels']
ctx = _abi.Context(0)
ctx.set_array(bl, ch, nt_max=nt)
rng = NP.random.default_rng(0)
snap = (rng.normal(size=(bl.shape[0], ch.size)) + 1j * rng.normal(size=(bl.shape[0], ch.size)))
for t in range(nt):
    ctx.set_vis(snap, slot=t)
win = NP.blackman(ch.size) + 0.01
ms = []
for rep in range(5):
    ctx.delay_transform_device(nt, bpwts=win, pad=1.0, want_lag=False, want_power=True)
    ctx.sync()
    ms.append(ctx.timing()['last_delay_ms'])
rows = nt * bl.shape[0]
nbytes = rows * ch.size * 24.0
print(json.dumps({'delay_rows': rows, 'fft_length_kept': int(ch.size), 'device_ms': ms, 'algorithmic_bytes': nbytes,
                  'achieved_GBps_best': nbytes / (min(ms) * 1e-3) / 1e9,
                  'roofline': {'terms_per_launch': rows * ch.size, 'avg_kernel_ms': sum(ms[1:]) / len(ms[1:])},
                  'roofline_hbm': {'algorithmic_bytes_per_launch': nbytes}}))
ctx.close()
