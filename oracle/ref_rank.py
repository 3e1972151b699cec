"""One "MPI rank" of the reference's parallel model for the CPU baseline (TEST INFRASTRUCTURE / bench.py's cpu_baseline leg only).

The reference runs `mpirun -n N run_prisim.py` (README.rst:93-99): N independent single-threaded numpy processes, each simulating a
contiguous chunk of baselines (scripts/run_prisim.py:1749-1791) with the statements of interferometry.py:6320-6376.  bench.py starts N
of these workers at once on a bounded sample; each loads the shared inputs, evaluates oracle/skyvis_oracle.py (the line-by-line numpy
restatement) on ITS baselines with the reference's source slabs, and writes its visibilities and its own wall time.

    python oracle/ref_rank.py <inputs.npz> <rank> <nranks> <out_prefix> [slab_bytes]
"""
import os
import sys
import time

os.environ.setdefault('OMP_NUM_THREADS', '1')           # one rank = one core, like an mpirun slot
os.environ.setdefault('OPENBLAS_NUM_THREADS', '1')
os.environ.setdefault('MKL_NUM_THREADS', '1')

import numpy as NP          # noqa: E402

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import skyvis_oracle as O       # noqa: E402


def main():
    path, rank, nranks, prefix = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
    slab = int(sys.argv[5]) if len(sys.argv) > 5 else (128 << 20)
    with NP.load(path) as f:
        bl, ch, dircos, pb, pc = f['bl'], f['ch'], f['dircos'], f['pb'], f['pc']
        fw = f['fwhm'] if 'fwhm' in f.files else None
    per = (bl.shape[0] + nranks - 1) // nranks                                   # run_prisim.py:1775-1791: contiguous chunks
    lo, hi = min(rank * per, bl.shape[0]), min((rank + 1) * per, bl.shape[0])
    # start together: every rank waits for the go file so that the N processes really run side by side
    go = prefix + '.go'
    open('%s.ready%d' % (prefix, rank), 'w').close()
    t_wait = time.time()
    while not os.path.exists(go) and time.time() - t_wait < 120.0:
        time.sleep(0.002)
    t0 = time.perf_counter()
    vis = O.skyvis(bl[lo:hi], ch, dircos, pb, pc, fwhm_deg=fw, slab_bytes=slab) if hi > lo else NP.zeros((0, ch.size), dtype=NP.complex128)
    dt = time.perf_counter() - t0
    tmp = '%s.rank%d.tmp.npz' % (prefix, rank)
    NP.savez(tmp, vis=vis, lo=lo, hi=hi, seconds=dt)
    os.replace(tmp, '%s.rank%d.npz' % (prefix, rank))


if __name__ == '__main__':
    main()
