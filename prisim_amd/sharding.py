"""Baseline shards of a one-process-per-GPU run (the reference's `pp.key: 'bl'` chunks, scripts/run_prisim.py:1775-1791).

The reference cuts the baseline list into contiguous chunks.  The arrays here list baselines by LENGTH (getBaselineInfo sorts them,
interferometry.py:1878, 1904), and the kernels' cost depends on it: baseline groups whose step angle rules out the lifting rotation run
the 4-instruction rotation (+17 %), taper runs re-anchor their chains there, and the taper culling shortens exactly the long-baseline
groups.  Contiguous shards would hand all of that to the last rank and the job would run at its pace: measured on the headline workload at
N = 8 the last contiguous shard takes 9.08 ms against 7.7 ms for the others (slowest / mean = 1.15), dealt-out shards 8.0-8.1 ms each
(1.007; tools/shard_balance.py, profiles/r03_shard_balance.json).  So groups of consecutive baselines are dealt ROUND-ROBIN: group g goes
to rank g % world.  Groups are 256 baselines (one kernel block) for large arrays and smaller for small ones so that every rank gets work.
Shares differ by at most one group and are padded to equal size (repeating the last baseline) because the all-gather moves equal shards;
`unshard_rows` puts a gathered, rank-major cube back into the global baseline order.
"""
import numpy as NP


def group_size(nbl, world):
    return int(max(1, min(256, -(-nbl // (4 * max(world, 1))))))


def shard_index(nbl, world, rank, group=None):
    """Global indices (increasing) of the baselines of rank `rank`."""
    group = group_size(nbl, world) if group is None else int(group)
    ngroups = -(-nbl // group)
    parts = [NP.arange(g * group, min((g + 1) * group, nbl)) for g in range(rank, ngroups, world)]
    return NP.concatenate(parts).astype(NP.int64) if parts else NP.zeros(0, dtype=NP.int64)


def shard_size(nbl, world, group=None):
    """Baselines per rank after padding: the largest share."""
    return int(max(shard_index(nbl, world, r, group).size for r in range(world)))


def shard_rows(rows, world, rank, group=None):
    """(this rank's rows of a per-baseline array padded to the common shard size with copies of the last row, its global indices, real count)."""
    rows = NP.asarray(rows)
    idx = shard_index(rows.shape[0], world, rank, group)
    per = shard_size(rows.shape[0], world, group)
    mine = rows[idx]
    if mine.shape[0] < per:
        mine = NP.concatenate((mine, NP.repeat(rows[-1:], per - mine.shape[0], axis=0)), axis=0)
    return mine, idx, int(idx.size)


def unshard_rows(gathered, nbl, world, group=None):
    """`gathered`: (world * per, ...) rows in rank-major order as an all-gather of padded shards leaves them -> (nbl, ...) in global order."""
    gathered = NP.asarray(gathered)
    per = shard_size(nbl, world, group)
    if gathered.shape[0] != world * per:
        raise ValueError('gathered cube has {0} rows, expected {1} ranks x {2}'.format(gathered.shape[0], world, per))
    out = NP.empty((nbl,) + gathered.shape[1:], dtype=gathered.dtype)
    for r in range(world):
        idx = shard_index(nbl, world, r, group)
        out[idx] = gathered[r * per:r * per + idx.size]
    return out
