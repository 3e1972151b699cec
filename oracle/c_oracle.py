"""ctypes wrapper of oracle/skyvis_oracle.c (TEST INFRASTRUCTURE / CPU baseline only)."""
import ctypes as C
import os
import subprocess

import numpy as NP

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, 'liboracle_skyvis.so')
_LIB_NATIVE = os.path.join(_HERE, 'liboracle_skyvis_native.so')
_lib = None
flavour = 'portable (x86-64-v3)'


def build():
    subprocess.check_call(['make', '-s', '-C', _HERE])


def use_native_build():
    """Compile the oracle with -march=native on THIS machine and switch to it (bench.py's cpu_baseline leg: the baseline is
    timed on the host cores of the GPU box, so it gets that CPU's full instruction set).  Falls back silently to the
    portable build that ships with the snapshot."""
    global _lib, flavour
    try:
        subprocess.check_call(['make', '-s', '-B', '-C', _HERE, 'native'], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        lib = C.CDLL(_LIB_NATIVE)
    except Exception:
        return False
    _lib = None
    _declare(lib)
    _lib = lib
    flavour = 'native (-march=native on this host)'
    return True


def _declare(lib):
    lib.oracle_skyvis_f64.restype = C.c_int
    lib.oracle_skyvis_f64.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p,
                                      C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
    lib.oracle_max_threads.restype = C.c_int


def _load():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB):
            build()
        lib = C.CDLL(_LIB)
        _declare(lib)
        _lib = lib
    return _lib


def max_threads():
    return _load().oracle_max_threads()


def skyvis(baselines, channels, dircos, pbfluxes, pc_dircos, fwhm_deg=None, nthreads=0):
    lib = _load()
    bl = NP.ascontiguousarray(baselines, dtype=NP.float64).reshape(-1, 3)
    fr = NP.ascontiguousarray(channels, dtype=NP.float64).ravel()
    dc = NP.ascontiguousarray(dircos, dtype=NP.float64).reshape(-1, 3)
    pb = NP.ascontiguousarray(pbfluxes, dtype=NP.float64)
    pc = NP.ascontiguousarray(pc_dircos, dtype=NP.float64).ravel()
    fw = None if fwhm_deg is None else NP.ascontiguousarray(fwhm_deg, dtype=NP.float64).ravel()
    out = NP.empty((bl.shape[0], fr.size), dtype=NP.complex128)
    p = lambda a: None if a is None else a.ctypes.data_as(C.c_void_p)
    rc = lib.oracle_skyvis_f64(p(bl), bl.shape[0], p(fr), fr.size, p(dc), p(pb), dc.shape[0], p(pc), p(fw), p(out),
                               int(nthreads))
    if rc != 0:
        raise ValueError('oracle_skyvis_f64 rejected its arguments')
    return out
