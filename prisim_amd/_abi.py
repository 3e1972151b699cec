"""ctypes binding of libprisim_hip.so (include/prisim_hip.h) -- numpy + ctypes only, no torch.

The product path FAILS LOUDLY when the HIP library or a GPU is missing: there is no CPU
fallback anywhere in ``prisim_amd``.
"""
import ctypes as C
import os

import numpy as NP

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('PRISIM_HIP_LIB') or os.path.join(_HERE, 'lib', 'libprisim_hip.so')      # PRISIM_HIP_LIB: A/B another build of the same ABI
ABI_VERSION = 'prisim_hip 0.5 gfx950'       # prisim_hip_version(): bumped whenever a struct or a signature of include/prisim_hip.h changes

PRISIM_OK = 0
PRISIM_EINVAL, PRISIM_ENODEV, PRISIM_ENOMEM, PRISIM_ESTATE, PRISIM_ELIB, PRISIM_EINTERNAL = -1, -2, -3, -4, -5, -6
PRISIM_FP64, PRISIM_FP32 = 0, 1
PRISIM_KERNEL_AUTO, PRISIM_KERNEL_RECURRENCE, PRISIM_KERNEL_DIRECT = 0, 1, 2
PRISIM_BEAM_DELTA, PRISIM_BEAM_GAUSSIAN, PRISIM_BEAM_AIRY, PRISIM_BEAM_DIPOLE, PRISIM_BEAM_POLY = 0, 1, 2, 3, 4
PRISIM_DIPOLE_GENERAL, PRISIM_DIPOLE_SHORT, PRISIM_DIPOLE_HALFWAVE = 0, 1, 2
PRISIM_COORDS = {'radec': 0, 'hadec': 1, 'altaz': 2}

# every symbol include/prisim_hip.h declares (tests check the library exports all of them)
EXPORTS = (
    'prisim_hip_create', 'prisim_hip_destroy', 'prisim_hip_last_error', 'prisim_hip_version',
    'prisim_hip_set_array', 'prisim_hip_set_sky', 'prisim_hip_compute', 'prisim_hip_get_vis',
    'prisim_hip_skyvis', 'prisim_hip_set_vis', 'prisim_hip_set_sky_analytic',
    'prisim_hip_set_external_beam', 'prisim_hip_set_sky_external', 'prisim_hip_set_sky_external_analytic', 'prisim_hip_get_pbflux',
    'prisim_hip_delay_transform', 'prisim_hip_delay_transform_device', 'prisim_hip_get_lags', 'prisim_hip_get_delay_power',
    'prisim_hip_allgather_lags', 'prisim_hip_phase_rotate', 'prisim_hip_noise', 'prisim_hip_noise_indexed', 'prisim_hip_comm_unique_id',
    'prisim_hip_comm_init',
    'prisim_hip_allgather', 'prisim_hip_allgather_slot_async', 'prisim_hip_get_gathered', 'prisim_hip_gathered_checksum',
    'prisim_hip_sync', 'prisim_hip_get_timing', 'prisim_hip_device_info', 'prisim_hip_set_tuning',
    'prisim_hip_allgather_grad', 'prisim_hip_comm_selftest', 'prisim_hip_get_comm_stats', 'prisim_hip_set_gather_root', 'prisim_hip_set_shard_map', 'prisim_hip_device_pci', 'prisim_hip_comm_last_error',
    'prisim_hip_host_alloc', 'prisim_hip_host_free', 'prisim_hip_get_vis_async', 'prisim_hip_wait_downloads',
    'prisim_hip_set_catalog', 'prisim_hip_set_sky_from_catalog', 'prisim_hip_catalog_roi', 'prisim_hip_observe_catalog',
    'prisim_hip_comm_version',
)


class PrisimSky(C.Structure):
    _fields_ = [('nsrc', C.c_int64), ('dircos', C.c_void_p), ('pbflux', C.c_void_p),
                ('pbflux_is_f32', C.c_int32), ('pc_dircos', C.c_void_p), ('fwhm_deg', C.c_void_p), ('fluxes', C.c_void_p)]


class PrisimBeamExt(C.Structure):
    _fields_ = [('dipole_dircos', C.c_double * 3), ('dipole_mode', C.c_int32), ('array_nax1', C.c_int32),
                ('array_nax2', C.c_int32), ('ground_modify', C.c_int32), ('array_sep1', C.c_double), ('array_sep2', C.c_double),
                ('array_east2ax1_deg', C.c_double), ('array_pc_dircos', C.c_double * 3), ('ground_height', C.c_double),
                ('ground_scale', C.c_double), ('ground_max', C.c_double), ('bf_nelem', C.c_int32), ('bf_nrand', C.c_int32),
                ('bf_pos', C.c_void_p), ('bf_delays', C.c_void_p), ('bf_gains', C.c_void_p), ('poly_coef', C.c_double * 4)]


def make_beam_ext(ext):
    """dict -> PrisimBeamExt.  Keys: dipole_dircos, dipole_mode, array (dict nax1, nax2, sep1, sep2, east2ax1, pointing_dircos),
    ground (dict height, modifier{scale,max}), beamformer (dict positions [n,3], delays [n] or [n,nrand], gains likewise).
    The returned struct keeps the beamformer arrays alive (attribute _keep)."""
    if ext is None:
        return None
    x = PrisimBeamExt()
    dd = NP.asarray(ext.get('dipole_dircos', (1.0, 0.0, 0.0)), dtype=NP.float64).ravel()
    if dd.size != 3:
        raise ValueError('dipole_dircos must have 3 elements')
    x.dipole_dircos[:] = dd.tolist()
    x.dipole_mode = int(ext.get('dipole_mode', PRISIM_DIPOLE_GENERAL))
    arr = ext.get('array', None)
    if arr is not None:
        x.array_nax1, x.array_nax2 = int(arr['nax1']), int(arr['nax2'])
        x.array_sep1, x.array_sep2 = float(arr['sep1']), float(arr['sep2'])
        x.array_east2ax1_deg = float(arr.get('east2ax1', 0.0) or 0.0)
        pc = NP.asarray(arr.get('pointing_dircos', (0.0, 0.0, 1.0)), dtype=NP.float64).ravel()
        if pc.size != 3:
            raise ValueError('array pointing_dircos must have 3 elements')
        x.array_pc_dircos[:] = pc.tolist()
    gnd = ext.get('ground', None)
    if gnd is not None:
        x.ground_height = float(gnd['height'])
        mod = gnd.get('modifier', None)
        if isinstance(mod, dict):
            x.ground_modify = 1
            if 'scale' in mod:
                x.ground_modify |= 2
                x.ground_scale = float(mod['scale'])
            if 'max' in mod:
                x.ground_modify |= 4
                x.ground_max = float(mod['max'])
    bf = ext.get('beamformer', None)
    if bf is not None:
        pos = NP.ascontiguousarray(bf['positions'], dtype=NP.float64)
        if pos.ndim != 2 or pos.shape[1] != 3:
            raise ValueError('beamformer positions must have shape (nelem, 3)')
        nel = pos.shape[0]
        delays = NP.asarray(bf.get('delays', NP.zeros(nel)), dtype=NP.float64).reshape(nel, -1)
        gains = NP.asarray(bf.get('gains', NP.ones(nel)), dtype=NP.float64).reshape(nel, -1)
        nrand = max(delays.shape[1], gains.shape[1])
        delays = NP.ascontiguousarray(NP.broadcast_to(delays, (nel, nrand)))
        gains = NP.ascontiguousarray(NP.broadcast_to(gains, (nel, nrand)))
        x.bf_nelem, x.bf_nrand = nel, nrand
        x.bf_pos, x.bf_delays, x.bf_gains = pos.ctypes.data, delays.ctypes.data, gains.ctypes.data
        x._keep = (pos, delays, gains)
    poly = ext.get('poly', None)
    if poly is not None:
        c = NP.zeros(4)
        pc = NP.asarray(poly, dtype=NP.float64).ravel()
        if pc.size < 1 or pc.size > 4:
            raise ValueError('poly must have 1 to 4 coefficients')
        c[:pc.size] = pc
        x.poly_coef[:] = c.tolist()
    return x


class PrisimBeamSky(C.Structure):
    _fields_ = [('nsrc', C.c_int64), ('dircos', C.c_void_p), ('flux_ref', C.c_void_p), ('spindex', C.c_void_p),
                ('flux_spectrum', C.c_void_p), ('ref_freq_hz', C.c_double), ('beam_kind', C.c_int32), ('diameter_m', C.c_double),
                ('beam_pc_dircos', C.c_void_p), ('pc_dircos', C.c_void_p), ('fwhm_deg', C.c_void_p), ('ext', C.c_void_p)]


class PrisimCatalog(C.Structure):
    _fields_ = [('nsrc', C.c_int64), ('coords', C.c_int32), ('reserved_', C.c_int32), ('location', C.c_void_p), ('flux_ref', C.c_void_p),
                ('spindex', C.c_void_p), ('ref_freq_hz', C.c_double), ('flux_spectrum', C.c_void_p), ('fwhm_deg', C.c_void_p),
                ('unitvec', C.c_void_p)]


class PrisimObs(C.Structure):
    _fields_ = [('latitude_deg', C.c_double), ('roi_radius_deg', C.c_double), ('roi_center', C.c_int32), ('use_external_beam', C.c_int32),
                ('beam_kind', C.c_int32), ('reserved_', C.c_int32), ('diameter_m', C.c_double), ('ext', C.c_void_p)]


class PrisimSnapshot(C.Structure):
    _fields_ = [('lst_deg', C.c_double), ('pc_dircos', C.c_double * 3), ('beam_pc_dircos', C.c_double * 3), ('frame_given', C.c_int32),
                ('reserved_', C.c_int32), ('cel2enu', C.c_double * 9), ('aberr_beta', C.c_double * 3)]


class PrisimPost(C.Structure):
    _fields_ = [('host_vis', C.c_void_p), ('host_is_c64', C.c_int32), ('gather', C.c_int32), ('gather_as_c64', C.c_int32), ('reserved_', C.c_int32)]


class PrisimTiming(C.Structure):
    _fields_ = [('last_kernel_ms', C.c_double), ('last_compute_ms', C.c_double), ('sum_kernel_ms', C.c_double),
                ('n_kernel', C.c_int64), ('last_terms', C.c_int64), ('last_kernel_id', C.c_int32),
                ('last_chan_tile', C.c_int32), ('last_nsplit', C.c_int32), ('last_lift_groups', C.c_int32),
                ('last_taper_group', C.c_int32), ('last_delay_fused', C.c_int32), ('last_delay_ms', C.c_double),
                ('last_taper_split', C.c_int32), ('last_split_uncorrected_groups', C.c_int32), ('last_culled_fraction', C.c_double),
                ('last_batch_snapshots', C.c_int32), ('reserved_', C.c_int32)]


class PrisimCommStats(C.Structure):
    _fields_ = [('n_gathers', C.c_int64), ('bytes_per_peer', C.c_int64), ('sum_gather_ms', C.c_double), ('last_gather_ms', C.c_double),
                ('max_gather_ms', C.c_double), ('last_gather_after_compute_ms', C.c_double), ('stream_priority', C.c_int32),
                ('stream_priority_lowest', C.c_int32), ('nranks', C.c_int32), ('reserved_', C.c_int32), ('sum_undeal_ms', C.c_double),
                ('last_undeal_ms', C.c_double)]


class PrisimHipError(RuntimeError):
    """Raised when libprisim_hip.so is missing/unloadable or no GPU is usable."""


_lib = None


def load_library():
    """Load libprisim_hip.so and declare the prototypes.  Raises PrisimHipError if it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise PrisimHipError('HIP extension not built: {0} is missing. Run `python -c "import __graft_entry__ as g; '
                             'g.build()"` (or make -C prisim_amd/csrc). There is no CPU fallback.'.format(LIB_PATH))
    try:
        lib = C.CDLL(LIB_PATH)
    except OSError as exc:
        raise PrisimHipError('cannot load {0}: {1}'.format(LIB_PATH, exc))
    vp, i64, i32, dbl = C.c_void_p, C.c_int64, C.c_int, C.c_double
    lib.prisim_hip_create.argtypes = [i32, C.POINTER(vp)]
    lib.prisim_hip_destroy.argtypes = [vp]
    lib.prisim_hip_destroy.restype = None
    lib.prisim_hip_last_error.argtypes = [vp]
    lib.prisim_hip_last_error.restype = C.c_char_p
    lib.prisim_hip_version.argtypes = []
    lib.prisim_hip_version.restype = C.c_char_p
    have = (lib.prisim_hip_version() or b'').decode()
    if have != ABI_VERSION:
        # a library of another round next to this binding: the structs below (prisim_timing, prisim_beam_sky ...) would be read wrongly
        raise PrisimHipError('{0} reports {1!r}, this binding is written for {2!r}: rebuild it (python -c "import __graft_entry__ as g; '
                             'g.build()")'.format(LIB_PATH, have, ABI_VERSION))
    lib.prisim_hip_set_array.argtypes = [vp, vp, i64, vp, i64, i64]
    lib.prisim_hip_set_sky.argtypes = [vp, C.POINTER(PrisimSky)]
    lib.prisim_hip_compute.argtypes = [vp, i32, i32, i32, i64]
    lib.prisim_hip_get_vis.argtypes = [vp, i64, vp, vp, i32]
    lib.prisim_hip_set_vis.argtypes = [vp, i64, vp]
    lib.prisim_hip_skyvis.argtypes = [vp, C.POINTER(PrisimSky), i32, i32, vp, vp, i32]
    lib.prisim_hip_set_sky_analytic.argtypes = [vp, C.POINTER(PrisimBeamSky)]
    lib.prisim_hip_set_external_beam.argtypes = [vp, vp, i64, i64, vp]
    lib.prisim_hip_set_sky_external.argtypes = [vp, C.POINTER(PrisimSky)]
    lib.prisim_hip_set_sky_external_analytic.argtypes = [vp, C.POINTER(PrisimBeamSky)]
    lib.prisim_hip_get_pbflux.argtypes = [vp, vp]
    lib.prisim_hip_delay_transform.argtypes = [vp, i64, vp, dbl, vp, vp, vp, dbl]
    lib.prisim_hip_delay_transform_device.argtypes = [vp, i64, vp, i64, dbl, i32, i32, dbl, vp, C.POINTER(i64)]
    lib.prisim_hip_get_lags.argtypes = [vp, i64, i64, vp, i64, vp]
    lib.prisim_hip_get_delay_power.argtypes = [vp, i64, i64, vp, i64, vp]
    lib.prisim_hip_allgather_lags.argtypes = [vp, i64]
    lib.prisim_hip_phase_rotate.argtypes = [vp, i64, vp]
    lib.prisim_hip_noise.argtypes = [vp, i64, vp, C.c_uint64, i64, vp]
    lib.prisim_hip_noise_indexed.argtypes = [vp, i64, vp, C.c_uint64, vp, vp]
    lib.prisim_hip_comm_unique_id.argtypes = [C.c_char_p]
    lib.prisim_hip_comm_init.argtypes = [vp, C.c_char_p, i32, i32]
    lib.prisim_hip_comm_version.argtypes = [C.c_char_p]
    lib.prisim_hip_allgather.argtypes = [vp, i64, i32]
    lib.prisim_hip_allgather_slot_async.argtypes = [vp, i64, i32]
    lib.prisim_hip_get_gathered.argtypes = [vp, i64, vp]
    lib.prisim_hip_gathered_checksum.argtypes = [vp, i64, C.POINTER(dbl)]
    lib.prisim_hip_sync.argtypes = [vp]
    lib.prisim_hip_get_timing.argtypes = [vp, C.POINTER(PrisimTiming), i32]
    lib.prisim_hip_device_info.argtypes = [vp, C.POINTER(i32), C.POINTER(i32), C.c_char_p]
    lib.prisim_hip_set_tuning.argtypes = [vp, i32, i32, i32]
    lib.prisim_hip_allgather_grad.argtypes = [vp, i64, i32]
    lib.prisim_hip_comm_selftest.argtypes = [vp, i64]
    lib.prisim_hip_set_gather_root.argtypes = [vp, i32]
    lib.prisim_hip_set_shard_map.argtypes = [vp, vp, i64]
    lib.prisim_hip_device_pci.argtypes = [i32, C.c_char_p]
    lib.prisim_hip_comm_last_error.argtypes = [C.c_char_p]
    lib.prisim_hip_get_comm_stats.argtypes = [vp, C.POINTER(PrisimCommStats), i32]
    lib.prisim_hip_host_alloc.argtypes = [i64, C.POINTER(vp)]
    lib.prisim_hip_host_free.argtypes = [vp]
    lib.prisim_hip_get_vis_async.argtypes = [vp, i64, vp, vp, i32]
    lib.prisim_hip_wait_downloads.argtypes = [vp]
    lib.prisim_hip_set_catalog.argtypes = [vp, C.POINTER(PrisimCatalog)]
    lib.prisim_hip_set_sky_from_catalog.argtypes = [vp, C.POINTER(PrisimObs), C.POINTER(PrisimSnapshot), C.POINTER(i64)]
    lib.prisim_hip_catalog_roi.argtypes = [vp, C.POINTER(PrisimObs), C.POINTER(PrisimSnapshot), C.POINTER(i64), vp, vp, i64]
    lib.prisim_hip_observe_catalog.argtypes = [vp, C.POINTER(PrisimObs), C.POINTER(PrisimSnapshot), i64, i32, i32, i64, vp, C.POINTER(PrisimPost)]
    for name in EXPORTS:
        fn = getattr(lib, name)
        if name not in ('prisim_hip_destroy', 'prisim_hip_last_error', 'prisim_hip_version'):
            fn.restype = C.c_int
    _lib = lib
    return lib


def _ptr(arr):
    return None if arr is None else arr.ctypes.data_as(C.c_void_p)


def _raise(code, msg):
    """Map the C-ABI error codes onto the exception types the reference raises inline."""
    if code == PRISIM_EINVAL:
        raise ValueError(msg)
    if code == PRISIM_ENOMEM:
        raise MemoryError(msg)
    if code == PRISIM_ESTATE:
        raise RuntimeError(msg)
    raise PrisimHipError(msg)


def host_empty(shape, dtype):
    """numpy array over page-locked host memory (prisim_hip_host_alloc): the destination of asynchronous downloads.  Freed when the
    last view of it is garbage-collected."""
    import weakref
    lib = load_library()
    dtype = NP.dtype(dtype)
    nbytes = int(NP.prod(shape, dtype=NP.int64)) * dtype.itemsize
    p = C.c_void_p()
    rc = lib.prisim_hip_host_alloc(max(nbytes, 1), C.byref(p))
    if rc != PRISIM_OK:
        _raise(rc, 'prisim_hip_host_alloc({0} B) failed: {1}'.format(nbytes, lib.prisim_hip_last_error(None).decode()))
    buf = (C.c_char * max(nbytes, 1)).from_address(p.value)
    weakref.finalize(buf, lib.prisim_hip_host_free, C.c_void_p(p.value))
    return NP.frombuffer(buf, dtype=dtype, count=int(NP.prod(shape, dtype=NP.int64))).reshape(shape)


class Context(object):
    """One GPU <-> one context <-> one HIP stream (include/prisim_hip.h)."""

    def __init__(self, device=0):
        self._lib = load_library()
        h = C.c_void_p()
        rc = self._lib.prisim_hip_create(int(device), C.byref(h))
        if rc != PRISIM_OK:
            msg = self._lib.prisim_hip_last_error(None).decode()
            _raise(rc, 'prisim_hip_create(device={0}) failed: {1}'.format(device, msg))
        self._h = h
        self.device = int(device)
        self.nbl = self.nchan = self.nt_max = 0
        self.nsrc = 0

    def close(self):
        if getattr(self, '_h', None):
            self._lib.prisim_hip_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def _check(self, rc, what):
        if rc != PRISIM_OK:
            msg = self._lib.prisim_hip_last_error(self._h).decode()
            _raise(rc, '{0} failed: {1}'.format(what, msg))

    # ---- array ----
    def set_array(self, baselines, freqs_hz, nt_max=1):
        bl = NP.ascontiguousarray(baselines, dtype=NP.float64).reshape(-1, 3)
        fr = NP.ascontiguousarray(freqs_hz, dtype=NP.float64).ravel()
        self._check(self._lib.prisim_hip_set_array(self._h, _ptr(bl), bl.shape[0], _ptr(fr), fr.size, int(nt_max)),
                    'prisim_hip_set_array')
        if getattr(self, 'nbl_total', 0) and bl.shape[0] != self.nbl:
            self.nbl_total = 0                       # (the library drops a shard map made for another shard size)
        self.nbl, self.nchan, self.nt_max = bl.shape[0], fr.size, int(nt_max)

    # ---- sky ----
    def _sky_struct(self, dircos, pbflux, pc_dircos, fwhm_deg, fluxes=None):
        dc = NP.ascontiguousarray(dircos, dtype=NP.float64).reshape(-1, 3)
        nsrc = dc.shape[0]
        pbflux = NP.asarray(pbflux)
        is_f32 = pbflux.dtype == NP.float32
        pb = NP.ascontiguousarray(pbflux, dtype=NP.float32 if is_f32 else NP.float64)
        if nsrc > 0 and pb.size != nsrc * self.nchan:
            raise ValueError('pbflux must have shape (nsrc, nchan) = ({0}, {1}), got {2}'.format(nsrc, self.nchan, pb.shape))
        pc = NP.ascontiguousarray(pc_dircos, dtype=NP.float64).ravel()
        if pc.size != 3:
            raise ValueError('pc_dircos must have 3 elements')
        fw = None
        if fwhm_deg is not None:
            fw = NP.ascontiguousarray(fwhm_deg, dtype=NP.float64).ravel()
            if fw.size != nsrc:
                raise ValueError('fwhm_deg must have nsrc elements')
        fl = None
        if fluxes is not None:
            fl = NP.ascontiguousarray(fluxes, dtype=NP.float64)
            if fl.size != nsrc * self.nchan:
                raise ValueError('fluxes must have shape (nsrc, nchan)')
        sky = PrisimSky(nsrc, _ptr(dc), _ptr(pb), 1 if is_f32 else 0, _ptr(pc), _ptr(fw), _ptr(fl))
        return sky, (dc, pb, pc, fw, fl)

    def set_sky(self, dircos, pbflux, pc_dircos, fwhm_deg=None, fluxes=None):
        """pbflux: beam x flux (nsrc, nchan); or, with `fluxes` given, the beam alone (product formed on the device)."""
        sky, keep = self._sky_struct(dircos, pbflux, pc_dircos, fwhm_deg, fluxes)
        self._check(self._lib.prisim_hip_set_sky(self._h, C.byref(sky)), 'prisim_hip_set_sky')
        self.nsrc = sky.nsrc

    def set_sky_analytic(self, dircos, flux_ref, spindex, ref_freq_hz, beam_kind, diameter_m, beam_pc_dircos,
                         pc_dircos, fwhm_deg=None, flux_spectrum=None, ext=None):
        """Fused beam x flux on the device.  Flux is the power law flux_ref*(f/ref)^spindex, or, when
        flux_spectrum (nsrc, nchan) is given, that tabulated spectrum."""
        dc = NP.ascontiguousarray(dircos, dtype=NP.float64).reshape(-1, 3)
        nsrc = dc.shape[0]
        fs = fr = sp = None
        if flux_spectrum is not None:
            fs = NP.ascontiguousarray(flux_spectrum, dtype=NP.float64)
            if fs.size != nsrc * self.nchan:
                raise ValueError('flux_spectrum must have shape (nsrc, nchan)')
            ref_freq_hz = 1.0 if ref_freq_hz is None else ref_freq_hz
        else:
            fr = NP.ascontiguousarray(flux_ref, dtype=NP.float64).ravel()
            sp = NP.ascontiguousarray(spindex, dtype=NP.float64).ravel()
            if fr.size != nsrc or sp.size != nsrc:
                raise ValueError('flux_ref and spindex must have nsrc elements')
        bpc = NP.ascontiguousarray(beam_pc_dircos, dtype=NP.float64).ravel()
        pc = NP.ascontiguousarray(pc_dircos, dtype=NP.float64).ravel()
        fw = None
        if fwhm_deg is not None:
            fw = NP.ascontiguousarray(fwhm_deg, dtype=NP.float64).ravel()
            if fw.size != nsrc:
                raise ValueError('fwhm_deg must have nsrc elements')
        xs = make_beam_ext(ext)
        sky = PrisimBeamSky(nsrc, _ptr(dc), _ptr(fr), _ptr(sp), _ptr(fs), float(ref_freq_hz), int(beam_kind), float(diameter_m),
                            _ptr(bpc), _ptr(pc), _ptr(fw), None if xs is None else C.cast(C.pointer(xs), C.c_void_p))
        self._check(self._lib.prisim_hip_set_sky_analytic(self._h, C.byref(sky)), 'prisim_hip_set_sky_analytic')
        self.nsrc = nsrc

    def set_external_beam(self, beam, interp_matrix):
        """beam (npix, nfreq) HEALPix RING, local frame; interp_matrix (nchan, nfreq) spectral interpolation operator."""
        b = NP.ascontiguousarray(beam, dtype=NP.float64)
        if b.ndim != 2:
            raise ValueError('beam must be a (npix, nfreq) array')
        m = NP.ascontiguousarray(interp_matrix, dtype=NP.float64)
        if m.shape != (self.nchan, b.shape[1]):
            raise ValueError('interp_matrix must have shape (nchan, nfreq) = ({0}, {1})'.format(self.nchan, b.shape[1]))
        self._check(self._lib.prisim_hip_set_external_beam(self._h, _ptr(b), b.shape[0], b.shape[1], _ptr(m)),
                    'prisim_hip_set_external_beam')

    def set_sky_external(self, dircos, fluxes, pc_dircos, fwhm_deg=None):
        dc = NP.ascontiguousarray(dircos, dtype=NP.float64).reshape(-1, 3)
        nsrc = dc.shape[0]
        fl = NP.ascontiguousarray(fluxes, dtype=NP.float64)
        if fl.size != nsrc * self.nchan:
            raise ValueError('fluxes must have shape (nsrc, nchan)')
        pc = NP.ascontiguousarray(pc_dircos, dtype=NP.float64).ravel()
        if pc.size != 3:
            raise ValueError('pc_dircos must have 3 elements')
        fw = None
        if fwhm_deg is not None:
            fw = NP.ascontiguousarray(fwhm_deg, dtype=NP.float64).ravel()
            if fw.size != nsrc:
                raise ValueError('fwhm_deg must have nsrc elements')
        sky = PrisimSky(nsrc, _ptr(dc), None, 0, _ptr(pc), _ptr(fw), _ptr(fl))
        self._check(self._lib.prisim_hip_set_sky_external(self._h, C.byref(sky)), 'prisim_hip_set_sky_external')
        self.nsrc = nsrc

    def set_sky_external_analytic(self, dircos, flux_ref, spindex, ref_freq_hz, pc_dircos, fwhm_deg=None, flux_spectrum=None):
        """External-beam sky whose flux spectra are formed on the device from the power law flux_ref*(f/ref)^spindex (only nsrc-sized
        vectors are uploaded per snapshot), or from flux_spectrum (nsrc, nchan) when given."""
        dc = NP.ascontiguousarray(dircos, dtype=NP.float64).reshape(-1, 3)
        nsrc = dc.shape[0]
        fs = fr = sp = None
        if flux_spectrum is not None:
            fs = NP.ascontiguousarray(flux_spectrum, dtype=NP.float64)
            if fs.size != nsrc * self.nchan:
                raise ValueError('flux_spectrum must have shape (nsrc, nchan)')
            ref_freq_hz = 1.0 if ref_freq_hz is None else ref_freq_hz
        else:
            fr = NP.ascontiguousarray(flux_ref, dtype=NP.float64).ravel()
            sp = NP.ascontiguousarray(spindex, dtype=NP.float64).ravel()
            if fr.size != nsrc or sp.size != nsrc:
                raise ValueError('flux_ref and spindex must have nsrc elements')
        pc = NP.ascontiguousarray(pc_dircos, dtype=NP.float64).ravel()
        if pc.size != 3:
            raise ValueError('pc_dircos must have 3 elements')
        fw = None
        if fwhm_deg is not None:
            fw = NP.ascontiguousarray(fwhm_deg, dtype=NP.float64).ravel()
            if fw.size != nsrc:
                raise ValueError('fwhm_deg must have nsrc elements')
        sky = PrisimBeamSky(nsrc, _ptr(dc), _ptr(fr), _ptr(sp), _ptr(fs), float(ref_freq_hz), PRISIM_BEAM_DELTA, 1.0,
                            _ptr(pc), _ptr(pc), _ptr(fw), None)
        self._check(self._lib.prisim_hip_set_sky_external_analytic(self._h, C.byref(sky)), 'prisim_hip_set_sky_external_analytic')
        self.nsrc = nsrc

    # ---- device-resident catalogue ----
    def set_catalog(self, location, coords, flux_ref=None, spindex=None, ref_freq_hz=None, flux_spectrum=None, fwhm_deg=None, unitvec='host'):
        """Upload a run's sky model once (prisim_hip_set_catalog): location (nsrc, 2) degrees in `coords` ('radec' | 'hadec' | 'altaz'),
        the power law flux_ref (f / ref_freq_hz)^spindex or flux_spectrum (nsrc, nchan), source sizes fwhm_deg (or None).
        unitvec: 'host' (default) -- the catalogue's unit vectors are formed here (geometry.catalog_unitvec) and uploaded, so that the host
        mirror of the snapshot geometry (geometry.frame_dircos) and the device select the same sources bit for bit; 'device' -- only
        `location` crosses and the device forms them; or an (nsrc, 3) array."""
        loc = NP.ascontiguousarray(location, dtype=NP.float64).reshape(-1, 2)
        nsrc = loc.shape[0]
        if coords not in PRISIM_COORDS:
            raise ValueError('coords must be "radec", "hadec" or "altaz"')
        uv = None
        if isinstance(unitvec, str):
            if unitvec == 'host':
                from . import geometry as _G
                uv = NP.ascontiguousarray(_G.catalog_unitvec(loc, coords))
            elif unitvec != 'device':
                raise ValueError("unitvec must be 'host', 'device' or an (nsrc, 3) array")
        elif unitvec is not None:
            uv = NP.ascontiguousarray(unitvec, dtype=NP.float64).reshape(-1, 3)
            if uv.shape[0] != nsrc:
                raise ValueError('unitvec must have shape (nsrc, 3)')
        fs = fr = sp = fw = None
        if flux_spectrum is not None:
            fs = NP.ascontiguousarray(flux_spectrum, dtype=NP.float64)
            if fs.size != nsrc * self.nchan:
                raise ValueError('flux_spectrum must have shape (nsrc, nchan)')
            ref_freq_hz = 1.0
        else:
            fr = NP.ascontiguousarray(flux_ref, dtype=NP.float64).ravel()
            sp = NP.ascontiguousarray(spindex, dtype=NP.float64).ravel()
            if fr.size != nsrc or sp.size != nsrc:
                raise ValueError('flux_ref and spindex must have nsrc elements')
        if fwhm_deg is not None:
            fw = NP.ascontiguousarray(fwhm_deg, dtype=NP.float64).ravel()
            if fw.size != nsrc:
                raise ValueError('fwhm_deg must have nsrc elements')
        cat = PrisimCatalog(nsrc, PRISIM_COORDS[coords], 0, _ptr(loc), _ptr(fr), _ptr(sp), float(ref_freq_hz), _ptr(fs), _ptr(fw), _ptr(uv))
        self._check(self._lib.prisim_hip_set_catalog(self._h, C.byref(cat)), 'prisim_hip_set_catalog')
        self.ncat = nsrc
        self.cat_coords = coords

    @staticmethod
    def make_obs(latitude_deg, roi_radius_deg=90.0, roi_center='zenith', beam_kind=PRISIM_BEAM_DELTA, diameter_m=1.0, ext=None,
                 use_external_beam=False):
        """prisim_obs of a run (keeps the beam extension alive as attribute _keep)."""
        xs = make_beam_ext(ext)
        obs = PrisimObs(float(latitude_deg), float(roi_radius_deg), 1 if roi_center == 'pointing_center' else 0, 1 if use_external_beam else 0,
                        int(beam_kind), 0, float(diameter_m), None if xs is None else C.cast(C.pointer(xs), C.c_void_p))
        obs._keep = xs
        return obs

    @staticmethod
    def _fill_snapshot(sn, lst_deg, pc_dircos, beam_pc_dircos, frame=None):
        """frame: None -- the library's fall-back rotation from lst and latitude (hour angle = LST - RA); or (R (3, 3), beta (3,)), the
        snapshot's catalogue-frame -> East-North-Up rotation and aberration vector (prisim_amd/frames.py snapshot_frame)."""
        sn.lst_deg = float(lst_deg)
        sn.pc_dircos[0], sn.pc_dircos[1], sn.pc_dircos[2] = float(pc_dircos[0]), float(pc_dircos[1]), float(pc_dircos[2])
        b = pc_dircos if beam_pc_dircos is None else beam_pc_dircos
        sn.beam_pc_dircos[0], sn.beam_pc_dircos[1], sn.beam_pc_dircos[2] = float(b[0]), float(b[1]), float(b[2])
        if frame is None:
            sn.frame_given = 0
        else:
            rot, beta = frame
            sn.frame_given = 1
            sn.cel2enu[:] = NP.asarray(rot, dtype=NP.float64).reshape(9).tolist()
            sn.aberr_beta[:] = NP.asarray(beta, dtype=NP.float64).reshape(3).tolist()
        return sn

    @classmethod
    def _snapshot(cls, lst_deg, pc_dircos, beam_pc_dircos, frame=None):
        return cls._fill_snapshot(PrisimSnapshot(), lst_deg, pc_dircos, beam_pc_dircos, frame)

    def set_sky_from_catalog(self, obs, lst_deg, pc_dircos, beam_pc_dircos=None, frame=None):
        """Snapshot geometry, region of interest, beam x flux of the resident catalogue on the device; returns the ROI source count."""
        sn = self._snapshot(lst_deg, pc_dircos, beam_pc_dircos, frame)
        n = C.c_int64()
        self._check(self._lib.prisim_hip_set_sky_from_catalog(self._h, C.byref(obs), C.byref(sn), C.byref(n)), 'prisim_hip_set_sky_from_catalog')
        self.nsrc = int(n.value)
        return self.nsrc

    def catalog_roi(self, obs, lst_deg, pc_dircos, want_indices=True, want_dircos=True, frame=None):
        """(indices int64 [n], dircos [n, 3]) of the region of interest of one snapshot, catalogue order (prisim_hip_catalog_roi)."""
        sn = self._snapshot(lst_deg, pc_dircos, None, frame)
        n = C.c_int64()
        self._check(self._lib.prisim_hip_catalog_roi(self._h, C.byref(obs), C.byref(sn), C.byref(n), None, None, 0), 'prisim_hip_catalog_roi')
        cnt = int(n.value)
        idx = NP.empty(cnt, dtype=NP.int64) if want_indices else None
        dc = NP.empty((cnt, 3), dtype=NP.float64) if want_dircos else None
        if cnt > 0 and (want_indices or want_dircos):
            self._check(self._lib.prisim_hip_catalog_roi(self._h, C.byref(obs), C.byref(sn), C.byref(n), _ptr(idx), _ptr(dc), cnt),
                        'prisim_hip_catalog_roi')
        return idx, dc

    def observe_catalog(self, obs, lst_deg, pc_dircos, beam_pc_dircos=None, precision=PRISIM_FP64, want_grad=False, slot0=0,
                        host_cube=None, gather=None, frames=None):
        """K snapshots of the resident catalogue in one call (prisim_hip_observe_catalog): lst_deg (K,), pc_dircos (K, 3) or (3,),
        beam_pc_dircos likewise (default: pc_dircos).  Results land in cube slots slot0 ... slot0 + K - 1; returns the ROI counts (K,).
        host_cube: page-locked (nt_max, nbl, nchan) complex128 / complex64 array (host_empty) every finished slot is downloaded into,
        behind its sky-sum; gather: None, or 'c128' / 'c64' -- every finished slot is all-gathered on the communication stream.
        frames: None (the library's fall-back rotation from lst and latitude), or K pairs (R (3, 3), beta (3,)) -- see _fill_snapshot.
        Arrays of at most 256 baselines take ONE launch per chunk of up to 256 snapshots in every mode (fp64, want_grad, and
        precision=PRISIM_FP32, whose arithmetic is then the fp64 launch's)."""
        lst = NP.asarray(lst_deg, dtype=NP.float64).ravel()
        k = lst.size
        if frames is not None and len(frames) != k:
            raise ValueError('frames must hold one (R, beta) pair per snapshot')
        # the K prisim_snapshot structs are filled through numpy views of their memory (20 doubles each: lst, pc[3], beam_pc[3],
        # {frame_given, reserved}, cel2enu[9], aberr_beta[3]) -- a per-field ctypes fill costs more than the C call of a small array
        cache = self.__dict__.get('_snap_cache')
        if cache is None or cache[0] != k:
            snaps = (PrisimSnapshot * k)()
            cache = self.__dict__['_snap_cache'] = (k, snaps, NP.frombuffer(snaps, dtype=NP.float64).reshape(k, 20),
                                                    NP.frombuffer(snaps, dtype=NP.int32).reshape(k, 40))
        _, snaps, fv, iv = cache
        fv[:, 0] = lst
        fv[:, 1:4] = NP.asarray(pc_dircos, dtype=NP.float64).reshape(-1, 3)
        fv[:, 4:7] = fv[:, 1:4] if beam_pc_dircos is None else NP.asarray(beam_pc_dircos, dtype=NP.float64).reshape(-1, 3)
        if frames is None:
            iv[:, 14] = 0
        else:
            iv[:, 14] = 1
            for t in range(k):
                fv[t, 8:17] = NP.asarray(frames[t][0], dtype=NP.float64).reshape(9)
                fv[t, 17:20] = frames[t][1]
        counts = NP.zeros(k, dtype=NP.int64)
        post = None
        if host_cube is not None or gather is not None:
            post = PrisimPost()
            if host_cube is not None:
                if host_cube.shape[1:] != (self.nbl, self.nchan) or host_cube.dtype not in (NP.complex128, NP.complex64) or not host_cube.flags['C_CONTIGUOUS']:
                    raise ValueError('host_cube must be a C-contiguous (nt, nbl, nchan) complex128 / complex64 array')
                if host_cube.shape[0] < slot0 + k:
                    raise ValueError('host_cube has fewer snapshots than slot0 + K')
                post.host_vis = host_cube.ctypes.data
                post.host_is_c64 = 1 if host_cube.dtype == NP.complex64 else 0
            if gather is not None:
                post.gather, post.gather_as_c64 = 1, (1 if gather == 'c64' else 0)
                self._gathered_c64 = gather == 'c64'
        self._check(self._lib.prisim_hip_observe_catalog(self._h, C.byref(obs), snaps, k, int(precision), 1 if want_grad else 0, int(slot0),
                                                         _ptr(counts), None if post is None else C.byref(post)), 'prisim_hip_observe_catalog')
        self.nsrc = int(counts[-1]) if k else 0
        return counts

    def get_pbflux(self):
        out = NP.empty((self.nsrc, self.nchan), dtype=NP.float64)
        self._check(self._lib.prisim_hip_get_pbflux(self._h, _ptr(out)), 'prisim_hip_get_pbflux')
        return out

    # ---- compute ----
    def compute(self, precision=PRISIM_FP64, kernel=PRISIM_KERNEL_AUTO, want_grad=False, slot=0):
        self._check(self._lib.prisim_hip_compute(self._h, int(precision), int(kernel), 1 if want_grad else 0, int(slot)),
                    'prisim_hip_compute')

    def get_vis(self, slot=0, want_grad=False, complex64=False):
        ctype = NP.complex64 if complex64 else NP.complex128
        vis = NP.empty((self.nbl, self.nchan), dtype=ctype)
        grad = NP.empty((3, self.nbl, self.nchan), dtype=ctype) if want_grad else None
        self._check(self._lib.prisim_hip_get_vis(self._h, int(slot), _ptr(vis), _ptr(grad), 1 if complex64 else 0),
                    'prisim_hip_get_vis')
        return (vis, grad) if want_grad else vis

    def skyvis(self, dircos, pbflux, pc_dircos, fwhm_deg=None, precision=PRISIM_FP64, kernel=PRISIM_KERNEL_AUTO,
               want_grad=False, complex64=False, fluxes=None):
        """One-shot drop-in for interferometry.py:6255-6376."""
        sky, keep = self._sky_struct(dircos, pbflux, pc_dircos, fwhm_deg, fluxes)
        ctype = NP.complex64 if complex64 else NP.complex128
        vis = NP.empty((self.nbl, self.nchan), dtype=ctype)
        grad = NP.empty((3, self.nbl, self.nchan), dtype=ctype) if want_grad else None
        self._check(self._lib.prisim_hip_skyvis(self._h, C.byref(sky), int(precision), int(kernel), _ptr(vis), _ptr(grad),
                                                1 if complex64 else 0), 'prisim_hip_skyvis')
        self.nsrc = sky.nsrc
        return (vis, grad) if want_grad else vis

    # ---- delay transform ----
    def delay_transform(self, nt, bpwts=None, pad=1.0, want_power=False, power_scale=1.0, want_lag=True):
        pad = max(float(pad), 0.0)
        nchan = self.nchan
        nfft = nchan + int(nchan * pad)
        nout = int(NP.arange(0, nfft, 1.0 + pad).size)
        w = None
        if bpwts is not None:
            w = NP.ascontiguousarray(bpwts, dtype=NP.float64).reshape(self.nbl, nchan)
        out = NP.empty((nt, self.nbl, nout), dtype=NP.complex128) if want_lag else None
        pw = NP.empty((nt, self.nbl, nout), dtype=NP.float64) if want_power else None
        lags = NP.empty(nchan, dtype=NP.float64)
        self._check(self._lib.prisim_hip_delay_transform(self._h, int(nt), _ptr(w), pad, _ptr(out), _ptr(lags), _ptr(pw),
                                                         float(power_scale)), 'prisim_hip_delay_transform')
        return out, lags, pw

    def delay_transform_device(self, nt, bpwts=None, pad=1.0, want_lag=True, want_power=False, power_scale=1.0):
        """Delay-transform slots [0, nt) and leave the spectra in HBM (prisim_hip_delay_transform_device).  Returns (lags, nout);
        read the results with get_lags / get_delay_power or exchange them with allgather_lags."""
        pad = max(float(pad), 0.0)
        w, wrows = None, 0
        if bpwts is not None:
            w = NP.ascontiguousarray(bpwts, dtype=NP.float64)
            if w.size == self.nchan:
                w, wrows = w.reshape(1, self.nchan), 1                   # one window for every baseline
            else:
                w, wrows = w.reshape(self.nbl, self.nchan), self.nbl
                if self.nbl > 1 and NP.array_equal(w, NP.broadcast_to(w[:1], w.shape)):
                    w, wrows = NP.ascontiguousarray(w[:1]), 1
        lags = NP.empty(self.nchan, dtype=NP.float64)
        nout = C.c_int64()
        self._check(self._lib.prisim_hip_delay_transform_device(self._h, int(nt), _ptr(w), wrows, pad, 1 if want_lag else 0, 1 if want_power else 0,
                                                                float(power_scale), _ptr(lags), C.byref(nout)),
                    'prisim_hip_delay_transform_device')
        self._dt_nout = int(nout.value)
        # every transform overwrites the context's ONE resident spectrum buffer: holders of a lazily fetched result (DelaySpectrum with
        # action='store', InterferometerArray.skyvis_lag) remember the generation they produced and transform again when it has moved on
        self._dt_generation = getattr(self, '_dt_generation', 0) + 1
        return lags, self._dt_nout

    def _get_resident(self, fn, what, dtype, t0, nt, rows):
        r = None
        nrow = self.nbl
        if rows is not None:
            r = NP.ascontiguousarray(rows, dtype=NP.int64).ravel()
            nrow = r.size
        out = NP.empty((nt, nrow, self._dt_nout), dtype=dtype)
        self._check(fn(self._h, int(t0), int(nt), _ptr(r), 0 if r is None else r.size, _ptr(out)), what)
        return out

    def get_lags(self, t0, nt, rows=None):
        """(nt, nrows | nbl, nout) complex128 lag spectra of snapshots [t0, t0 + nt) from the device-resident result."""
        return self._get_resident(self._lib.prisim_hip_get_lags, 'prisim_hip_get_lags', NP.complex128, t0, nt, rows)

    def get_delay_power(self, t0, nt, rows=None):
        return self._get_resident(self._lib.prisim_hip_get_delay_power, 'prisim_hip_get_delay_power', NP.float64, t0, nt, rows)

    def allgather_lags(self, nt):
        """RCCL all-gather of the resident lag spectra, device to device; read with get_gathered(nt, nranks, row=nout)."""
        self._check(self._lib.prisim_hip_allgather_lags(self._h, int(nt)), 'prisim_hip_allgather_lags')
        self._gathered_c64 = False

    def set_vis(self, vis, slot=0):
        v = NP.ascontiguousarray(vis, dtype=NP.complex128)
        if v.shape != (self.nbl, self.nchan):
            raise ValueError('vis must have shape (nbl, nchan)')
        self._check(self._lib.prisim_hip_set_vis(self._h, int(slot), _ptr(v)), 'prisim_hip_set_vis')

    def delay_transform_host(self, vis, bpwts, pad):
        """Delay-transform one host snapshot (nbl, nchan): upload into slot 0, transform, download."""
        self.set_vis(vis, 0)
        out, lags, pw = self.delay_transform(1, bpwts=bpwts, pad=pad)
        return out[0], lags, pw

    def phase_rotate(self, nt, diff_dircos):
        """cube[t] *= exp(-2 pi i f (b . diff[t])/c) for t < nt, in place on the device (interferometry.py:7871-7877)."""
        d = NP.ascontiguousarray(diff_dircos, dtype=NP.float64).reshape(-1, 3)
        if d.shape[0] != nt:
            raise ValueError('diff_dircos must have one row per snapshot')
        self._check(self._lib.prisim_hip_phase_rotate(self._h, int(nt), _ptr(d)), 'prisim_hip_phase_rotate')

    def noise(self, rms, seed, bl_offset=0, bl_index=None):
        """Complex Gaussian noise (nt, nbl, nchan) with per-element rms (interferometry.py:6692), Philox counter-based draws
        on the device: identical for sharded and unsharded runs when bl_offset is the shard's first global baseline -- or, for
        shards that are not one contiguous range, bl_index holds the global index of every local baseline."""
        r = NP.ascontiguousarray(rms, dtype=NP.float64)
        if r.ndim != 3 or r.shape[1:] != (self.nbl, self.nchan):
            raise ValueError('rms must have shape (nt, nbl, nchan)')
        out = NP.empty(r.shape, dtype=NP.complex128)
        if bl_index is not None:
            ix = NP.ascontiguousarray(bl_index, dtype=NP.int64).ravel()
            if ix.size != self.nbl:
                raise ValueError('bl_index must have one entry per baseline')
            self._check(self._lib.prisim_hip_noise_indexed(self._h, r.shape[0], _ptr(r), int(seed) & 0xFFFFFFFFFFFFFFFF, _ptr(ix), _ptr(out)),
                        'prisim_hip_noise_indexed')
            return out
        self._check(self._lib.prisim_hip_noise(self._h, r.shape[0], _ptr(r), int(seed) & 0xFFFFFFFFFFFFFFFF, int(bl_offset), _ptr(out)),
                    'prisim_hip_noise')
        return out

    # ---- multi-GPU ----
    @staticmethod
    def comm_unique_id():
        lib = load_library()
        buf = C.create_string_buffer(128)
        rc = lib.prisim_hip_comm_unique_id(buf)
        if rc != PRISIM_OK:
            _raise(rc, 'prisim_hip_comm_unique_id failed: ' + lib.prisim_hip_last_error(None).decode())
        return buf.raw

    @staticmethod
    def comm_version():
        """'librccl <version> (<path>)' of the RCCL the library loads."""
        lib = load_library()
        buf = C.create_string_buffer(128)
        rc = lib.prisim_hip_comm_version(buf)
        if rc != PRISIM_OK:
            _raise(rc, 'prisim_hip_comm_version failed: ' + lib.prisim_hip_last_error(None).decode())
        return buf.value.decode()

    @staticmethod
    def device_pci(device):
        lib = load_library()
        buf = C.create_string_buffer(64)
        lib.prisim_hip_device_pci(int(device), buf)
        return buf.value.decode()

    @staticmethod
    def comm_last_error():
        """Text of librccl's last error / warning; safe to call from a watchdog thread while comm_init blocks in another."""
        lib = load_library()
        buf = C.create_string_buffer(512)
        lib.prisim_hip_comm_last_error(buf)
        return buf.value.decode()

    def comm_init(self, uid, nranks, rank):
        if len(uid) != 128:
            raise ValueError('unique id must be 128 bytes')
        self._check(self._lib.prisim_hip_comm_init(self._h, uid, int(nranks), int(rank)), 'prisim_hip_comm_init')
        if getattr(self, 'nranks', 1) != int(nranks):
            self.nbl_total = 0                       # (the library drops a shard map made for another communicator size)
        self.nranks = int(nranks)

    def allgather(self, nt, complex64=False):
        self._check(self._lib.prisim_hip_allgather(self._h, int(nt), 1 if complex64 else 0), 'prisim_hip_allgather')
        self._gathered_c64 = bool(complex64)

    def allgather_slot_async(self, slot, complex64=False):
        """Gather one snapshot on the communication stream, overlapping later compute() calls."""
        self._check(self._lib.prisim_hip_allgather_slot_async(self._h, int(slot), 1 if complex64 else 0),
                    'prisim_hip_allgather_slot_async')
        self._gathered_c64 = bool(complex64)

    def set_shard_map(self, bl_index, nbl_total):
        """bl_index (nranks, nbl_shard): global baseline of every local row of every rank (negative = padding), or None to go back to the
        rank-major layout.  Afterwards every gather leaves the gathered cube in the global baseline order of the unsharded array,
        (nt, nbl_total, row), on the device (prisim_hip_set_shard_map)."""
        if bl_index is None:
            self._check(self._lib.prisim_hip_set_shard_map(self._h, None, 0), 'prisim_hip_set_shard_map')
            self.nbl_total = 0
            return
        m = NP.ascontiguousarray(bl_index, dtype=NP.int64)
        if m.size != getattr(self, 'nranks', 1) * self.nbl:
            raise ValueError('bl_index must have shape (nranks, nbl_shard)')
        self._check(self._lib.prisim_hip_set_shard_map(self._h, _ptr(m), int(nbl_total)), 'prisim_hip_set_shard_map')
        self.nbl_total = int(nbl_total)

    def get_gathered(self, nt, nranks=None, row=None):
        """(nt, nranks, nbl_shard, row): snapshot-major, rank blocks in rank order; row = nchan (visibilities) or nout (delay spectra).
        With a shard map set: (nt, nbl_total, row) in global baseline order."""
        nranks = getattr(self, 'nranks', 1) if nranks is None else nranks
        row = self.nchan if row is None else int(row)
        dtype = NP.complex64 if getattr(self, '_gathered_c64', False) else NP.complex128
        if getattr(self, 'nbl_total', 0) > 0:
            out = NP.empty((nt, self.nbl_total, row), dtype=dtype)
            self._check(self._lib.prisim_hip_get_gathered(self._h, int(nt), _ptr(out)), 'prisim_hip_get_gathered')
            return out
        out = NP.empty((nt, nranks, self.nbl, row), dtype=dtype)
        self._check(self._lib.prisim_hip_get_gathered(self._h, int(nt), _ptr(out)), 'prisim_hip_get_gathered')
        return out

    def allgather_grad(self, nt, complex64=False):
        """Gather the baseline-gradient cube of nt snapshots; read with get_gathered_grad(nt, nranks)."""
        self._check(self._lib.prisim_hip_allgather_grad(self._h, int(nt), 1 if complex64 else 0), 'prisim_hip_allgather_grad')
        self._gathered_c64 = bool(complex64)

    def get_gathered_grad(self, nt, nranks=None):
        """(nt, nranks, 3, nbl_shard, nchan) after allgather_grad; with a shard map set (nt, 3, nbl_total, nchan) in global order."""
        nranks = getattr(self, 'nranks', 1) if nranks is None else nranks
        if getattr(self, 'nbl_total', 0) > 0:
            return self.get_gathered(nt, nranks, row=3 * self.nchan).reshape(nt, 3, self.nbl_total, self.nchan)
        g = self.get_gathered(nt, nranks, row=3 * self.nchan)                  # rows of 3*nchan: the block is [3][nbl][nchan] per rank
        return g.reshape(nt, nranks, 3, self.nbl, self.nchan)

    def set_gather_root(self, root=None):
        """Later gathers deliver to rank `root` only (None: to every rank); the other ranks then hold no gathered cube."""
        self._check(self._lib.prisim_hip_set_gather_root(self._h, -1 if root is None else int(root)), 'prisim_hip_set_gather_root')

    def comm_selftest(self, nbytes=1 << 20):
        """All-gather of a rank-dependent pattern, verified on the host; raises PrisimHipError when the communicator cannot move data."""
        self._check(self._lib.prisim_hip_comm_selftest(self._h, int(nbytes)), 'prisim_hip_comm_selftest')

    def comm_stats(self, reset=False):
        st = PrisimCommStats()
        self._check(self._lib.prisim_hip_get_comm_stats(self._h, C.byref(st), 1 if reset else 0), 'prisim_hip_get_comm_stats')
        return {k: getattr(st, k) for k, _ in PrisimCommStats._fields_ if k != 'reserved_'}

    # ---- asynchronous downloads ----
    def get_vis_async(self, slot, out, grad_out=None):
        """Enqueue the download of slot `slot` into `out` (nbl, nchan) complex128 / complex64 -- ideally an array from host_empty() --
        on the copy stream, behind the compute issued so far.  The arrays must stay alive until wait_downloads() / sync()."""
        if out.shape != (self.nbl, self.nchan) or out.dtype not in (NP.complex128, NP.complex64) or not out.flags['C_CONTIGUOUS']:
            raise ValueError('out must be a C-contiguous (nbl, nchan) complex128 / complex64 array')
        c64 = out.dtype == NP.complex64
        if grad_out is not None and (grad_out.shape != (3, self.nbl, self.nchan) or grad_out.dtype != out.dtype or not grad_out.flags['C_CONTIGUOUS']):
            raise ValueError('grad_out must be a C-contiguous (3, nbl, nchan) array of the dtype of out')
        self._check(self._lib.prisim_hip_get_vis_async(self._h, int(slot), _ptr(out), _ptr(grad_out), 1 if c64 else 0), 'prisim_hip_get_vis_async')

    def wait_downloads(self):
        self._check(self._lib.prisim_hip_wait_downloads(self._h), 'prisim_hip_wait_downloads')

    def gathered_checksum(self, nt, complex64=None):
        v = C.c_double()
        self._check(self._lib.prisim_hip_gathered_checksum(self._h, int(nt), C.byref(v)), 'prisim_hip_gathered_checksum')
        return v.value

    # ---- misc ----
    def sync(self):
        self._check(self._lib.prisim_hip_sync(self._h), 'prisim_hip_sync')

    def timing(self, reset=False):
        t = PrisimTiming()
        self._check(self._lib.prisim_hip_get_timing(self._h, C.byref(t), 1 if reset else 0), 'prisim_hip_get_timing')
        return {k: getattr(t, k) for k, _ in PrisimTiming._fields_ if k != 'reserved_'}

    def device_info(self):
        cu, clk = C.c_int(), C.c_int()
        name = C.create_string_buffer(64)
        self._check(self._lib.prisim_hip_device_info(self._h, C.byref(cu), C.byref(clk), name), 'prisim_hip_device_info')
        return {'name': name.value.decode(), 'cu_count': cu.value, 'clock_khz': clk.value}

    def set_tuning(self, chan_tile=0, src_chunk=0, nsplit=0):
        self._check(self._lib.prisim_hip_set_tuning(self._h, int(chan_tile), int(src_chunk), int(nsplit)),
                    'prisim_hip_set_tuning')
