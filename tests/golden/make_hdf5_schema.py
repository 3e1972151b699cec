"""Generates tests/golden/hdf5_schema.json: the on-disk layout InterferometerArray.save(fmt='HDF5') of the reference produces
(prisim/interferometry.py:8722-8854), obtained by EXECUTING those statements of the reference -- they are plain h5py calls --
against a recording stand-in for the h5py file object, with `self` a representative observed array (two snapshots, baseline
gradient, noise, delay spectra, redundancy groups).  No reference text is stored: the script reads the cited line range from
/root/reference at generation time (build container only) and writes paths, kinds, dtypes, ranks and attributes.

  python tests/golden/make_hdf5_schema.py        # needs /root/reference; rewrites tests/golden/hdf5_schema.json

What h5py stores for a python value is restated in `describe` (python str -> variable-length string scalar, float -> float64
scalar, int -> int64 scalar, list -> array, numpy array -> its dtype, complex -> compound {r, i}).
"""
import json
import os
import sys
import textwrap
import types

import numpy

HERE = os.path.dirname(os.path.abspath(__file__))
REF = '/root/reference/prisim/interferometry.py'
FIRST, LAST = 8723, 8854          # body of `with h5py.File(filename, write_str) as fileobj:` in save() (:8722)


def describe(value):
    if isinstance(value, str):
        return {'kind': 'scalar', 'dtype': 'str'}
    if isinstance(value, bytes):
        return {'kind': 'scalar', 'dtype': 'bytes'}
    if isinstance(value, bool):
        return {'kind': 'scalar', 'dtype': 'bool'}
    if isinstance(value, int):
        return {'kind': 'scalar', 'dtype': 'int64'}
    if isinstance(value, float):
        return {'kind': 'scalar', 'dtype': 'float64'}
    arr = numpy.asarray(value)
    if arr.dtype.names:
        dt = 'compound(' + ','.join('%s:%s' % (n, arr.dtype[n].kind) for n in arr.dtype.names) + ')'
    elif arr.dtype.kind in 'SU':
        dt = 'string'
    else:
        dt = arr.dtype.name
    return {'kind': 'scalar' if arr.ndim == 0 else 'array', 'dtype': dt, 'ndim': int(arr.ndim)}


class Dataset(object):
    def __init__(self, rec, path, value):
        self.attrs = {}
        self.value = value
        rec[path] = dict(describe(value), attrs=self.attrs)


class Group(object):
    def __init__(self, rec, path):
        self._rec, self._path, self._items = rec, path, {}
        if path:
            rec[path] = {'kind': 'group'}

    def _join(self, name):
        return (self._path + '/' + name) if self._path else name

    def create_group(self, name):
        g = Group(self._rec, self._join(name))
        self._items[name] = g
        return g

    def __setitem__(self, name, value):
        self._items[name] = Dataset(self._rec, self._join(name), value)

    def __getitem__(self, name):
        return self._items[name]


def representative_array():
    """An observed InterferometerArray as the reference's observe()/add_noise()/delay_transform() leave it (types per
    interferometry.py:5665-5870, 6384-6399, 6685-6692, 8114-8134): 3 baselines, 4 channels, 2 snapshots."""
    nbl, nchan, nt = 3, 4, 2
    rng = numpy.random.default_rng(0)
    cplx = lambda *s: rng.normal(size=s) + 1j * rng.normal(size=s)          # noqa: E731
    s = types.SimpleNamespace()
    s.flux_unit = 'JY'
    s.latitude, s.longitude, s.altitude = -30.7, 21.4, 1050.0
    s.telescope = {'id': 'hera', 'shape': 'dish', 'size': 14.0, 'ocoords': 'altaz', 'orientation': numpy.array([[90.0, 270.0]]), 'groundplane': 0.3}
    s.freq_resolution = 1.0e5
    s.channels = 150e6 + 1e5 * numpy.arange(nchan)
    s.lags = numpy.arange(nchan) * 1e-6
    s.bp = numpy.ones((nbl, nchan, nt))
    s.bp_wts = numpy.ones((nbl, nchan, nt))
    s.simparms_file = '/path/to/simparms.yaml'
    s.layout = {'positions': rng.normal(size=(3, 3)), 'coords': 'ENU', 'labels': numpy.array([b'0', b'1', b'2']), 'ids': numpy.arange(3)}
    s.t_obs, s.n_acc, s.t_acc = 21.4, nt, [10.7, 10.7]
    s.timestamp = [2457000.5, 2457000.6]
    s.pointing_coords = s.phase_center_coords = 'hadec'
    s.skycoords = 'radec'
    s.lst = [10.0, 10.1]
    s.pointing_center = numpy.zeros((nt, 2))
    s.phase_center = numpy.zeros((nt, 2))
    s.labels = [(b'1', b'0'), (b'2', b'0'), (b'2', b'1')]
    s.baselines = rng.normal(size=(nbl, 3))
    s.baseline_coords = 'localenu'
    s.projected_baselines = rng.normal(size=(nbl, 3, nt))
    s.A_eff = numpy.full((nbl, nchan), 154.0)
    s.eff_Q = numpy.full((nbl, nchan), 0.96)
    s.Tsysinfo = [{'Trx': 100.0, 'Tant': {'T0': 200.0, 'f0': 150e6, 'spindex': -2.5}, 'Tnet': None}] * nt
    s.Tsys = numpy.full((nbl, nchan, nt), 300.0)
    s.vis_rms_freq = numpy.ones((nbl, nchan, nt))
    s.vis_freq, s.skyvis_freq, s.vis_noise_freq = cplx(nbl, nchan, nt), cplx(nbl, nchan, nt), cplx(nbl, nchan, nt)
    s.vis_lag, s.skyvis_lag, s.vis_noise_lag = cplx(nbl, nchan, nt), cplx(nbl, nchan, nt), cplx(nbl, nchan, nt)
    s.gradient_mode = 'baseline'
    s.gradient = {'baseline': cplx(3, nbl, nchan, nt)}
    s.gaininfo = None
    s.blgroups = {(b'1', b'0'): numpy.array([(b'1', b'0'), (b'2', b'1')])}
    s.bl_reversemap = {(b'1', b'0'): (b'1', b'0'), (b'2', b'1'): (b'1', b'0')}
    return s


def main():
    with open(REF) as f:
        lines = f.readlines()[FIRST - 1:LAST]
    block = textwrap.dedent(''.join(lines))
    rec = {}
    np_proxy = types.SimpleNamespace(**{k: getattr(numpy, k) for k in ('asarray',)}, float=float)        # NP.float left numpy in 1.24
    env = {'fileobj': Group(rec, ''), 'self': representative_array(), 'NP': np_proxy, 'outfile': 'out',
           'astroutils': types.SimpleNamespace(__githash__='githash'), 'prisim': types.SimpleNamespace(__githash__='githash')}
    exec(compile(block, 'reference save() HDF5 block', 'exec'), env)
    out = {'_source': 'prisim/interferometry.py:%d-%d executed against a recording h5py stand-in (tests/golden/make_hdf5_schema.py)' % (FIRST, LAST),
           'objects': {k: rec[k] for k in sorted(rec)}}
    with open(os.path.join(HERE, 'hdf5_schema.json'), 'w') as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print('wrote %d objects' % len(rec))


if __name__ == '__main__':
    main()
