"""Antenna layouts and baseline generation (host side, numpy).

Restates the recipes of the reference that the BASELINE configs need:
  * hexagon_generator          prisim/interferometry.py:857-989
  * baseline_generator (j>i)   prisim/interferometry.py:1354
  * orientation fold + sort    prisim/interferometry.py:1866-1883 (getBaselineInfo)
  * HERA-N presets             prisim/interferometry.py:1808-1827 (14.6 m spacing)
"""
import numpy as NP


def hexagon_generator(spacing, n_total=None, n_side=None, orientation=None, center=None):
    """Hexagonally packed antenna positions (interferometry.py:857-989).

    Returns (xy [n_total,2], labels [list of str]).  Same argument names, validation and error
    types as the reference.
    """
    if spacing is None:
        raise NameError('No spacing provided.')
    if not isinstance(spacing, (int, float)):
        raise TypeError('spacing must be scalar value')
    if orientation is not None and not isinstance(orientation, (int, float)):
        raise TypeError('orientation must be a scalar')
    if center is not None:
        center = NP.asarray(center, dtype=float)
        if center.size != 2:
            raise ValueError('center should be a 2-element vector')
        center = center.reshape(1, -1)
    if (n_total is None) and (n_side is None):
        raise NameError('n_total or n_side must be provided')
    elif (n_total is not None) and (n_side is not None):
        raise ValueError('Only one of n_total or n_side must be specified.')
    elif n_total is not None:
        if not isinstance(n_total, (int, NP.integer)):
            raise TypeError('n_total must be an integer')
        if n_total <= 0:
            raise ValueError('n_total must be positive')
    else:
        if not isinstance(n_side, (int, NP.integer)):
            raise TypeError('n_side must be an integer')
        if n_side <= 0:
            raise ValueError('n_side must be positive')

    if n_total is not None:
        sqroots = NP.roots([3.0, -3.0, 1.0 - n_total])                       # :945
        valid = NP.logical_and(sqroots.real >= 1, sqroots.imag == 0.0)
        if not NP.any(valid):
            raise ValueError('No valid root found for the quadratic equation with the specified n_total')
        n_side = int(NP.round(sqroots[valid].real)[0])
        if 3 * n_side ** 2 - 3 * n_side + 1 != n_total:
            raise ValueError('n_total is not a valid number for a hexagonal array')
    else:
        n_total = 3 * n_side ** 2 - 3 * n_side + 1

    xref = NP.arange(2 * n_side - 1, dtype=float)                             # :958
    xloc, yloc = [], []
    for i in range(1, n_side):                                                # :960-965
        x = xref[:-i] + i * NP.cos(NP.pi / 3)
        y = i * NP.sin(NP.pi / 3) * NP.ones(2 * n_side - 1 - i)
        xloc += x.tolist() * 2
        yloc += y.tolist()
        yloc += (-y).tolist()
    xloc += xref.tolist()                                                     # :967-968
    yloc += [0.0] * int(2 * n_side - 1)
    xy = NP.asarray(list(zip(xloc, yloc)))
    if xy.shape[0] != n_total:
        raise ValueError('Sizes of x- and y-locations do not agree with n_total')
    xy = xy - NP.mean(xy, axis=0, keepdims=True)                              # :978
    if orientation is not None:
        angle = NP.radians(orientation)
        rot = NP.asarray([[NP.cos(angle), -NP.sin(angle)], [NP.sin(angle), NP.cos(angle)]])
        xy = NP.dot(xy, rot.T)
    xy = xy * spacing
    if center is not None:
        xy = xy + center
    return xy, [str(i) for i in range(n_total)]


def baseline_generator(antenna_locations, auto=False, conjugate=False):
    """All antenna pairs as baseline vectors pos[j]-pos[i], i outer loop, j>i (interferometry.py:1354).

    Returns (baselines [nbl,3], ids [nbl,2] as (j, i)).
    """
    pos = NP.asarray(antenna_locations, dtype=float)
    if pos.ndim != 2 or pos.shape[1] not in (2, 3):
        raise ValueError('antenna_locations must be an N x 2 or N x 3 array')
    if pos.shape[1] == 2:
        pos = NP.hstack((pos, NP.zeros((pos.shape[0], 1))))
    n = pos.shape[0]
    ii, jj = NP.triu_indices(n, k=0 if auto else 1)      # row-major: i outer, j inner with j>=i / j>i
    bl = pos[jj] - pos[ii]
    ids = NP.stack((jj, ii), axis=1)
    if conjugate:
        i2, j2 = NP.tril_indices(n, k=-1)                # i outer, j<i
        bl = NP.vstack((bl, pos[j2] - pos[i2]))
        ids = NP.vstack((ids, NP.stack((j2, i2), axis=1)))
    return bl, ids


def fold_and_sort_baselines(bl, ids=None):
    """Orientation fold to (-67.5, 112.5] deg and stable sort by length (interferometry.py:1866-1883)."""
    bl = NP.array(bl, dtype=float, copy=True)
    blo = NP.angle(bl[:, 0] + 1j * bl[:, 1], deg=True)
    neg = (blo < -67.5) | (blo > 112.5)
    bl[neg, :] = -1.0 * bl[neg, :]
    if ids is not None:
        ids = NP.array(ids, copy=True)
        ids[neg] = ids[neg][:, ::-1]
    length = NP.sqrt(NP.sum(bl ** 2, axis=1))
    order = NP.argsort(length, kind='mergesort')
    if ids is not None:
        return bl[order], ids[order]
    return bl[order]


def uniq_baselines(baseline_locations, redundant=None):
    """Unique / redundant / non-redundant baselines of a set (interferometry.py:1373-1461).

    Two baselines are the same when their length (to 0.01 m), zenith angle and folded orientation in [0, 180) deg (both to
    0.001 arcsec) agree -- the reference's string key '{len:.2f}_{za:.3f}_{orientation:.3f}', kept verbatim because the order
    of the returned unique baselines is the lexicographic order of those keys.
    redundant=None: every distinct baseline; True: only those occurring more than once; False: only those occurring once.
    Returns (baselines [n,3], index of the first occurrence of each, counts, list of index lists of all occurrences).
    """
    if not isinstance(baseline_locations, NP.ndarray):
        raise TypeError('baseline_locations must be a numpy array')
    if redundant is not None and not isinstance(redundant, bool):
        raise TypeError('keyword "redundant" must be set to None or a boolean value')
    bl = NP.asarray(baseline_locations, dtype=float)
    if bl.shape[1] > 3:
        bl = bl[:, :3]
    elif bl.shape[1] < 3:
        bl = NP.hstack((bl, NP.zeros((bl.shape[0], 3 - bl.shape[1]))))
    orient = NP.angle(bl[:, 0] + 1j * bl[:, 1], deg=True)
    orient[orient >= 180.0] -= 180.0
    orient[orient < 0.0] += 180.0
    length = NP.sqrt(NP.sum(bl ** 2, axis=1))
    za = NP.degrees(NP.arccos(bl[:, 2] / length))
    keys = NP.array(['%.2f_%.3f_%.3f' % (l, 3.6e3 * z, 3.6e3 * o) for l, z, o in zip(length, za, orient)])
    _, first, inverse, counts = NP.unique(keys, return_index=True, return_inverse=True, return_counts=True)
    if redundant is None:
        sel = NP.arange(first.size)
    elif redundant:
        sel = NP.flatnonzero(counts > 1)
    else:
        sel = NP.flatnonzero(counts == 1)
    order = NP.argsort(inverse, kind='mergesort')
    starts = NP.concatenate(([0], NP.cumsum(counts)))
    occurrences = [order[starts[g]:starts[g + 1]].tolist() for g in sel]
    return bl[first[sel], :], first[sel], counts[sel], occurrences


_HERA_PRESETS = (7, 19, 37, 61, 91, 127, 169, 217, 271, 331)


def array_layout(name):
    """Antenna positions (ENU metres, [n,3]) for the layout names the BASELINE configs use.

    'HERA-N' for N in the reference's preset list (interferometry.py:1808-1827); 'HERA-350' is
    not a reference preset (its largest is HERA-331): here it is the 331-element hex
    plus 19 outriggers evenly spaced on a circle of twice the hex radius (deterministic).
    """
    if not isinstance(name, str) or not name.startswith('HERA-'):
        raise ValueError('unknown array layout {0!r}'.format(name))
    n = int(name.split('-')[1])
    if n in _HERA_PRESETS:
        xy, _ = hexagon_generator(14.6, n_total=n)
    elif n == 350:
        xy, _ = hexagon_generator(14.6, n_total=331)
        radius = 2.0 * NP.max(NP.sqrt(NP.sum(xy ** 2, axis=1)))
        ang = 2 * NP.pi * (NP.arange(19) + 0.25) / 19.0
        xy = NP.vstack((xy, radius * NP.stack((NP.cos(ang), NP.sin(ang)), axis=1)))
    else:
        raise ValueError('no preset for {0!r}'.format(name))
    return NP.hstack((xy, NP.zeros((xy.shape[0], 1))))


def layout_baselines(name):
    """Baselines of a named layout the way getBaselineInfo builds them (all pairs j>i, folded, sorted)."""
    pos = array_layout(name)
    bl, ids = baseline_generator(pos)
    return fold_and_sort_baselines(bl, ids)
