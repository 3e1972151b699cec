"""Where a snapshot of the PRODUCT loop goes: one rank's share of a BASELINE configuration through
InterferometerArray.observe() / observe_batch() on a (RA, Dec) sky model, as prisim_amd.driver.run drives it -- wall per snapshot with the
queue kept full (one synchronisation at the end) -- beside the kernel-only figure (the same shard's compute() with the sky already
resident, queued back to back), with the device-resident catalogue on and off (PRISIM_CATALOG=0: the sky of every snapshot formed on the
host and uploaded, the path of rounds 1-4).

    python tools/product_loop.py [--config 4|2|3] [--nranks 8] [--n-acc 32] [--precision fp32|fp64]
prints one JSON line per (mode, catalogue on / off)."""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as NP


import bench


def run_case(cfgno, nranks, n_acc, memsave, mode, catalog, reps=2):
    return bench.product_loop_case(cfgno, nranks, n_acc, memsave, mode, catalog, reps=reps)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--config', type=int, default=4)
    ap.add_argument('--nranks', type=int, default=8)
    ap.add_argument('--n-acc', type=int, default=32)
    ap.add_argument('--precision', default=None)
    ap.add_argument('--modes', default='observe,batch')
    args = ap.parse_args()
    prec = args.precision or ('fp64' if args.config == 2 else 'fp32')
    for mode in args.modes.split(','):
        for catalog in (True, False):
            if mode == 'batch' and not catalog:
                continue
            print(json.dumps(run_case(args.config, args.nranks, args.n_acc, prec == 'fp32', mode, catalog)), flush=True)


if __name__ == '__main__':
    main()
