"""prisim_amd/hdf5io.py: the ctypes HDF5 writer behind InterferometerArray.save (PRISim's on-disk layout, interferometry.py:8717-8846)."""
import os
import shutil
import subprocess

import numpy as NP
import pytest

from prisim_amd import hdf5io

try:
    hdf5io._load()
    HAVE = True
except hdf5io.HDF5Unavailable:
    HAVE = False

needs_hdf5 = pytest.mark.skipif(not HAVE, reason='the HDF5 C library is not installed')


@needs_hdf5
def test_roundtrip_of_every_value_kind(tmp_path):
    rng = NP.random.default_rng(0)
    vis = rng.normal(size=(3, 4, 2)) + 1j * rng.normal(size=(3, 4, 2))
    vis32 = vis.astype(NP.complex64)
    labels = NP.asarray([('a1', 'a0'), ('a2', 'a0'), ('a3', 'a1')], dtype=[('A2', 'S4'), ('A1', 'S4')])
    path = str(tmp_path / 't.hdf5')
    with hdf5io.File(path, 'w') as f:
        f.create_group('header')
        f.write('header/flux_unit', 'JY')
        f.write('telescope_parms/latitude', -30.72, attrs={'units': 'deg'})
        f.write('timing/n_acc', 2)
        f.write('spectral_info/freqs', NP.arange(4) * 1e5 + 150e6, attrs={'units': 'Hz', 'scale': 2.5})
        f.write('visibilities/freq_spectrum/skyvis', vis, attrs={'units': 'Jy'})
        f.write('visibilities/freq_spectrum/skyvis32', vis32)
        f.write('array/labels', labels)
        f.write('layout/labels', NP.asarray(['a0', 'a1', 'a22']))
        f.write('layout/ids', NP.arange(3, dtype=NP.int32))
        f.write('empty', NP.zeros((0, 3)))
    with open(path, 'rb') as fh:
        assert fh.read(8) == b'\x89HDF\r\n\x1a\n'
    with hdf5io.File(path, 'r') as f:
        assert f.read('header/flux_unit') == 'JY'
        assert f.read('telescope_parms/latitude') == -30.72 and f.read_attr('telescope_parms/latitude', 'units') == 'deg'
        assert f.read('timing/n_acc') == 2 and f.read('timing/n_acc').dtype == NP.int64
        assert NP.array_equal(f.read('spectral_info/freqs'), NP.arange(4) * 1e5 + 150e6)
        assert f.read_attr('spectral_info/freqs', 'scale') == 2.5
        got = f.read('visibilities/freq_spectrum/skyvis')
        assert got.dtype == NP.complex128 and NP.array_equal(got, vis)
        assert NP.array_equal(f.read('visibilities/freq_spectrum/skyvis32'), vis32)
        assert NP.array_equal(f.read('array/labels'), labels) and f.read('array/labels').dtype.names == ('A2', 'A1')
        assert f.read('layout/labels').tolist() == [b'a0', b'a1', b'a22']
        assert f.read('layout/ids').dtype == NP.int32 and f.read('empty').shape == (0, 3)
        assert f.exists('visibilities/freq_spectrum') and not f.exists('nope')
        with pytest.raises(KeyError):
            f.read('nope')
    with pytest.raises(IOError):
        hdf5io.File(path, 'w-')                               # exists: the reference's overwrite=False behaviour
    with pytest.raises(TypeError):
        with hdf5io.File(str(tmp_path / 'u.hdf5'), 'w') as f:
            f.write('bad', NP.asarray([object()]))


@needs_hdf5
def test_file_is_what_h5py_would_write(tmp_path):
    """Independent check with the HDF5 tools when they are installed: complex numbers are the compound {r, i}, python strings are
    variable-length UTF-8 scalars, python numbers are scalar datasets -- h5py's conventions, which PRISim's readers rely on."""
    h5dump = shutil.which('h5dump') or ('/opt/conda/bin/h5dump' if os.path.exists('/opt/conda/bin/h5dump') else None)
    if h5dump is None:
        pytest.skip('h5dump not installed')
    path = str(tmp_path / 'v.hdf5')
    with hdf5io.File(path, 'w') as f:
        f.write('header/flux_unit', 'JY')
        f.write('timing/t_obs', 120.5)
        f.write('visibilities/freq_spectrum/skyvis', NP.ones((2, 3, 1), dtype=NP.complex128), attrs={'units': 'Jy'})
    text = subprocess.run([h5dump, '-H', path], capture_output=True, text=True, timeout=60).stdout
    flat = ' '.join(text.split())
    assert 'GROUP "visibilities" { GROUP "freq_spectrum" { DATASET "skyvis"' in flat
    assert 'H5T_COMPOUND { H5T_IEEE_F64LE "r"; H5T_IEEE_F64LE "i"; }' in flat and 'DATASPACE SIMPLE { ( 2, 3, 1 ) / ( 2, 3, 1 ) }' in flat
    assert 'DATASET "flux_unit" { DATATYPE H5T_STRING { STRSIZE H5T_VARIABLE;' in flat and 'CSET H5T_CSET_UTF8' in flat
    assert 'DATASET "t_obs" { DATATYPE H5T_IEEE_F64LE DATASPACE SCALAR' in flat
    assert 'ATTRIBUTE "units"' in flat
