"""One rank of a multi-rank rehearsal of bench.py (tests/test_distributed_cpu.py): bench.main() itself runs -- argument handling, the
socket rendezvous, the hand-over of the communicator id, the timed loop with its per-snapshot gathers, the gather check and the ONE
JSON line of the contract -- with a stand-in for the GPU context that has no kernels behind it: compute() writes a pattern that is a
function of the rank's own baseline shard, and the exchange goes through files in a directory every rank can see.  What this does not
cover is exactly what needs the hardware: the HIP kernels (tests -m gpu) and RCCL itself."""
import os
import sys
import time

import numpy as NP

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from prisim_amd import _abi  # noqa: E402

XDIR = os.environ['BENCH_REHEARSAL_DIR']


def _wait_for(path, timeout=120.0):
    t0 = time.time()
    while not os.path.exists(path):
        if time.time() - t0 > timeout:
            raise TimeoutError(path)
        time.sleep(0.005)


class PatternContext(object):
    """Same call surface as prisim_amd._abi.Context as far as bench.py uses it with --no-cpu-baseline."""
    mode = os.environ.get('BENCH_REHEARSAL_MODE', 'ok')

    def __init__(self, device=0):
        self.device = device
        self._n = 0
        self._gathered = {}

    def close(self):
        pass

    def set_array(self, baselines, freqs_hz, nt_max=1):
        self.bl = NP.asarray(baselines, dtype=NP.float64)                        # every baseline (the shard map speaks of rows), a thinned
        self.ch = NP.asarray(freqs_hz, dtype=NP.float64)[::64]                   # channel axis: what is exchanged is 64 times smaller
        self.nbl = self.bl.shape[0]
        self.nt_max = int(nt_max)
        self.cube = NP.zeros((self.nt_max, self.bl.shape[0], self.ch.size), dtype=NP.complex64)

    def set_sky_analytic(self, dircos, *args, **kwargs):
        self.nsrc = NP.asarray(dircos).shape[0]

    @staticmethod
    def comm_unique_id():
        if PatternContext.mode == 'no_uid':
            raise RuntimeError('stand-in: librccl cannot be loaded')
        return bytes(range(128))

    def comm_init(self, uid, nranks, rank):
        assert uid == bytes(range(128))
        if self.mode == 'init_fails_on_1' and rank == 1:
            raise RuntimeError('stand-in: ncclCommInitRank failed')
        self.nranks, self.rank = int(nranks), int(rank)

    def set_shard_map(self, bl_index, nbl_total):
        m = NP.asarray(bl_index, dtype=NP.int64)
        assert m.shape == (self.nranks, self.nbl) and NP.array_equal(NP.sort(m[m >= 0]), NP.arange(nbl_total))
        self._smap, self._nbl_total = m, int(nbl_total)

    def compute(self, precision=0, kernel=0, want_grad=False, slot=0):
        self._n += 1
        self.cube[slot] = (self.bl[:, 0:1] * (1 + slot) + 1j * (self.bl[:, 1:2] + self.ch[None, :] * 1e-9)).astype(NP.complex64)

    def allgather_slot_async(self, slot, complex64=False):
        assert complex64
        tag = '%d_%d' % (self._n, slot)                                          # one exchange per compute, in order
        tmp = os.path.join(XDIR, 'x_%s_%d.tmp.npy' % (tag, self.rank))
        NP.save(tmp, self.cube[slot])
        os.replace(tmp, os.path.join(XDIR, 'x_%s_%d.npy' % (tag, self.rank)))
        parts = []
        for r in range(self.nranks):
            p = os.path.join(XDIR, 'x_%s_%d.npy' % (tag, r))
            _wait_for(p)
            parts.append(NP.load(p))
        out = NP.empty((self._nbl_total, self.ch.size), dtype=NP.complex64)      # un-dealt: [global baseline][f], padding dropped
        for r in range(self.nranks):
            keep = self._smap[r] >= 0
            out[self._smap[r][keep]] = parts[r][keep]
        self._gathered[slot] = out

    def comm_selftest(self, nbytes=1 << 20):
        if self.mode == 'selftest_fails_on_0' and self.rank == 0:
            raise RuntimeError('stand-in: the self-test all-gather delivered a corrupted block')

    def comm_stats(self, reset=False):
        return {'n_gathers': self._n, 'bytes_per_peer': int(self.cube[0].nbytes), 'sum_gather_ms': 0.25 * self._n, 'last_gather_ms': 0.25,
                'max_gather_ms': 0.3, 'last_gather_after_compute_ms': 0.2, 'stream_priority': -1, 'stream_priority_lowest': 1,
                'nranks': self.nranks, 'sum_undeal_ms': 0.01 * self._n, 'last_undeal_ms': 0.01}

    def sync(self):
        pass

    def timing(self, reset=False):
        return {'sum_kernel_ms': 1.0 * self._n, 'n_kernel': self._n, 'last_chan_tile': 64, 'last_nsplit': 1, 'last_kernel_ms': 1.0,
                'last_terms': 0, 'last_taper_group': 0, 'last_delay_ms': 0.0, 'last_delay_fused': 0}

    def gathered_checksum(self, nt, complex64=False):
        g = NP.stack([self._gathered[t] for t in range(nt)])
        if self.mode == 'rank1_differs' and self.rank == 1:
            return float(NP.sum(g.real.astype(NP.float64))) + 1.0
        return float(NP.sum(g.real.astype(NP.float64)) + NP.sum(g.imag.astype(NP.float64)))

    def get_gathered(self, nt, nranks=None, row=None):
        return NP.stack([self._gathered[t] for t in range(nt)])

    def get_vis(self, slot=0, want_grad=False, complex64=False):
        return self.cube[slot].copy()


_abi.Context = PatternContext
os.environ['PRISIM_BENCH_SELF'] = os.path.abspath(__file__)        # a bare `--gpus N` launch spawns THIS file as its ranks
import bench  # noqa: E402

if __name__ == '__main__':
    sys.argv = ['bench.py'] + sys.argv[1:]
    bench.main()
