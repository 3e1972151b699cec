"""Worker for tests/test_distributed_cpu.py: one process per rank, gloo backend, no GPU.

Exercises the N>1 plumbing bench.py uses -- equal-size contiguous baseline shards, out-of-band broadcast
of the 128-byte communicator id, gather of the shards, max-reduce of timings -- with the oracle standing in
for the GPU kernel and a gloo all_gather standing in for the RCCL one (RCCL needs one GPU per rank)."""
import os
import sys

import numpy as NP
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench                                      # noqa: E402
from oracle import skyvis_oracle as O             # noqa: E402
from prisim_amd import workloads as W             # noqa: E402


def main():
    rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
    dist.init_process_group(backend='gloo', rank=rank, world_size=world)
    cfg = W.subsample(W.config2(), bl_stride=3, ch_count=24, src_stride=9)     # 57 baselines: not divisible by 2
    bl, ch, sky = cfg['baselines'], cfg['channels'], cfg['sky']
    pb = sky['flux_ref'][:, None] * (ch[None, :] / sky['ref_freq']) ** sky['spindex'][:, None]
    zen = NP.array([0.0, 0.0, 1.0])

    uid = [bytes(range(128)) if rank == 0 else None]
    dist.broadcast_object_list(uid, src=0)
    assert uid[0] == bytes(range(128))

    mine, n_real = bench.shard_baselines(bl, world, rank)
    per = mine.shape[0]
    shard = O.skyvis(mine, ch, sky['dircos'], pb, zen, fwhm_deg=sky['fwhm_deg'])
    t = torch.from_numpy(NP.ascontiguousarray(shard.view(NP.float64)))
    parts = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(parts, t)
    gathered = NP.stack([p.numpy().view(NP.complex128) for p in parts])           # [world][per][nchan]
    full = O.skyvis(bl, ch, sky['dircos'], pb, zen, fwhm_deg=sky['fwhm_deg'])
    flat = gathered.reshape(world * per, -1)[:bl.shape[0]]                         # drop the padding rows
    err = NP.max(NP.abs(flat - full))
    tt = torch.tensor([float(rank + 1)], dtype=torch.float64)
    dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    assert tt.item() == float(world)
    dist.barrier()
    if err > 1e-9:
        print('RANK %d MISMATCH %g' % (rank, err))
        sys.exit(1)
    print('RANK %d OK per=%d n_real=%d err=%.2e' % (rank, per, n_real, err))
    dist.destroy_process_group()


if __name__ == '__main__':
    main()
