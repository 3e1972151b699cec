import sys, os, json, struct
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as NP
from prisim_amd import _abi, workloads as W
cfg = W.config3()
bl, ch, sky = cfg['baselines'], cfg['channels'], cfg['sky']
zen = NP.array([0.0, 0.0, 1.0])
ctx = _abi.Context(0)
ctx.set_array(bl, ch, nt_max=1)
ctx.set_sky_analytic(sky['dircos'], sky['flux_ref'], sky['spindex'], sky['ref_freq'], _abi.PRISIM_BEAM_AIRY, 14.0, zen, zen)
os.environ['PRISIM_HIP_BALANCED_COST_NOLIFT'] = '90'
ctx.compute(precision=_abi.PRISIM_FP32); ctx.sync()
path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'gpurun_out', 'bal_trace.bin')
os.environ['PRISIM_HIP_BALANCED_TRACE'] = path
ctx.compute(precision=_abi.PRISIM_FP32); ctx.sync()
print('kernel ms', ctx.timing()['last_kernel_ms'])
raw = open(path, 'rb').read()
nb, mp, nbg, nt = struct.unpack('4i', raw[:16])
tr = NP.frombuffer(raw[16:16 + 8 * nb * (2 + mp)], dtype=NP.uint64).reshape(nb, 2 + mp).astype(NP.float64)
pcs = NP.frombuffer(raw[16 + 8 * nb * (2 + mp):], dtype=NP.int32).reshape(nb, mp, 4)
t0 = tr[:, 0].min()
print('blocks', nb, 'max_pieces', mp)
dur = []
for b in range(nb):
    seg = (b & 7) * (nb >> 3) + (b >> 3)
    stamps = tr[b, 1:]
    n = int(NP.argmax(stamps == 0)) if (stamps == 0).any() else mp
    end = stamps[n - 1] if n > 0 else tr[b, 0]
    dur.append(((end - tr[b, 0]) / 1e5, (tr[b, 0] - t0) / 1e5, n, seg))
dur = NP.array(dur)
print('block duration ms: min %.2f median %.2f max %.2f; start offset max %.3f ms' % (dur[:, 0].min(), NP.median(dur[:, 0]), dur[:, 0].max(), dur[:, 1].max()))
order = NP.argsort(-dur[:, 0])[:12]
for b in order:
    seg = int(dur[b, 3]); n = int(dur[b, 2])
    st = NP.concatenate(([tr[b, 0]], tr[b, 1:1 + n]))
    per = NP.diff(st) / 1e5
    items = [(int(pcs[seg, i, 0]) % nbg, int(pcs[seg, i, 2] - pcs[seg, i, 1])) for i in range(n)]
    print('block %d seg %d xcd %d: %.2f ms  pieces (bg, nsrc, ms): %s' % (b, seg, b & 7, dur[b, 0], ' '.join('(%d,%d,%.2f)' % (it[0], it[1], p_) for it, p_ in zip(items, per))))
print('per-XCD mean duration', [round(float(dur[NP.arange(nb) % 8 == x, 0].mean()), 2) for x in range(8)])
# cost per source by group class
lift_ms, nolift_ms = [], []
for b in range(nb):
    seg = (b & 7) * (nb >> 3) + (b >> 3); n = int(dur[b, 2])
    st = NP.concatenate(([tr[b, 0]], tr[b, 1:1 + n])); per = NP.diff(st) / 1e5
    for i in range(n):
        ns = int(pcs[seg, i, 2] - pcs[seg, i, 1]); bgi = int(pcs[seg, i, 0]) % nbg
        if ns >= 5000:
            (nolift_ms if bgi >= 225 else lift_ms).append(per[i] / ns * 1e4)
print('ms per 1e4 sources: lift median %.3f (n=%d), no-lift median %.3f (n=%d)' % (NP.median(lift_ms), len(lift_ms), NP.median(nolift_ms), len(nolift_ms)))
