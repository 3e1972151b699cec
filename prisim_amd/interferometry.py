"""Host-side mirror of the reference's ``InterferometerArray`` for the sky-sum path.

Same class / method / keyword names, argument meaning and exception types as
prisim/interferometry.py of nithyanandan/PRISim (constructor :5140-5145, 5665-5870; observe
:5874-6410; observing_run :6414-6657; delay_transform :8052-8137) so that a PRISim user can switch
to ``prisim_amd.interferometry.InterferometerArray`` for this path.  Everything numerical on the
path -- tau, the source-shape taper, the DFT sum, the baseline gradient, pb*flux, the analytic
beams in scope, the delay transform -- runs in HIP kernels behind include/prisim_hip.h.  There is
NO CPU fallback: without libprisim_hip.so or a GPU the constructor raises ``PrisimHipError``.

Differences kept on purpose:
  Q1  ``timeobj`` may be an astropy-``Time``-like object (``.jd``, ``.sidereal_time('apparent').deg``),
      or a ``(jd, lst_deg)`` tuple, or a float jd with the ``lst`` keyword; astropy is not required.
  Q2  ``observing_run`` works (the reference's passes a str as timeobj and cannot run).
  Q4  baseline gradients work for shapeless sky models too.
  Q5  ``memsave`` reduces the phase in fp64 before the fp32 recurrence (more accurate than the reference's
      all-fp32 phase); results are complex64 like the reference's.  On arrays of at most 256 baselines observed from the resident
      catalogue the arithmetic of a ``memsave`` snapshot is fp64 (the batched launch of include/prisim_hip.h serves it: faster than any
      fp32 launch chain at that size); the stored type stays complex64.
  Q7  the visibility cube is grown without O(nt^2) recopy; same logical shape (nbl, nchan, n_acc).
  Q20 delay_transform transforms whichever of the three cubes exist.
  sky coordinates 'radec': the reference's astropy chain FK5(skymodel.epoch) -> FK5(obstime) -> AltAz (:6174-6180) is ONE rotation and
      ONE aberration vector per snapshot, applied on the device (prisim_snapshot.cel2enu / aberr_beta).  Without astropy they come from
      prisim_amd/frames.py: IAU 2006 precession + truncated IAU 1980 nutation + annual aberration + rotation by the apparent LST
      (``frame_model = 'apparent'``, the default; 'mean' = precession only; 'date' = none, HA = LST - RA).  Not modelled: frame bias,
      light deflection, diurnal aberration, polar motion (< 1 arcsec together; parity with astropy unpinned).  A caller with astropy sets
      ``frame_provider`` and reproduces the reference's frame exactly.
Out of scope here: gain tables, FITS persistence, uvfits / uvh5.
"""
import os
import warnings

import numpy as NP

from . import _abi
from . import baseline_delay_horizon as DLY
from . import frames as FRAMES
from . import geometry as GEOM
from . import primary_beams as PB

C_LIGHT = 299792458.0
SIDEREAL_RATE = 1.00273790935    # sidereal seconds per solar second


class _DeviceSlot(object):
    """Placeholder in InterferometerArray._cube for a snapshot that so far lives only in slot `slot` of the device cube (after
    reserve()); the host copy is fetched the first time skyvis_freq is read.  Runs that gather on the device never fetch it."""
    __slots__ = ('slot', 'dtype', 'staged')

    def __init__(self, slot, dtype, staged=False):
        self.slot, self.dtype = slot, NP.dtype(dtype)
        self.staged = staged           # an asynchronous download into the pinned host cube is in flight or done (reserve(host_staging=True))


class _LayerStack(object):
    """The per-snapshot layers of a (nbl, nchan, n_acc) attribute (bp, bp_wts, Tsys) in the broadcastable form they were given in
    -- (1 | nbl, 1 | nchan) -- instead of the dense arrays the reference grows by NP.dstack at every observe()
    (interferometry.py:6019-6024, 6082-6086: 0.5 GB per snapshot and attribute at HERA-350 x 1024 channels, O(n_acc^2) copies).
    The dense array with the reference's shape is formed when the attribute is read."""

    def __init__(self, nbl, nchan, initial):
        self.nbl, self.nchan = nbl, nchan
        self.initial = NP.asarray(initial, dtype=NP.float64)         # value before the first snapshot, broadcastable to (nbl, nchan)
        self.layers = []

    def append(self, layer):
        layer = NP.asarray(layer, dtype=NP.float64)
        if layer.ndim == 3:
            layer = layer[:, :, 0]
        self.layers.append(layer.reshape((layer.shape[0] if layer.ndim == 2 else 1), -1) if layer.ndim else layer.reshape(1, 1))

    def dense(self):
        if not self.layers:
            return NP.array(NP.broadcast_to(self.initial, (self.nbl, self.nchan)))
        return NP.stack([NP.broadcast_to(l, (self.nbl, self.nchan)) for l in self.layers], axis=2)

    _ONE = NP.ones((1, 1))

    def ones_like(self):
        # (observe() resets bp_wts to ones of the shape of bp at EVERY snapshot (:6024): a stack that simply follows this one's length --
        # a fresh array per existing layer made that O(n_acc^2) allocations over a run, 0.5 ms of a 0.76 ms config-2 snapshot after 300
        # snapshots; a fresh list per snapshot still O(n_acc))
        return _OnesStack(self)


class _OnesStack(_LayerStack):
    """bp_wts after observe(): ones with as many layers as the bandpass stack it follows."""

    def __init__(self, parent):
        self.nbl, self.nchan = parent.nbl, parent.nchan
        self.initial = self._ONE
        self.parent = parent

    @property
    def layers(self):
        return [self._ONE] * len(self.parent.layers)

    @layers.setter
    def layers(self, value):          # (the rollback of observe_batch restores lists: nothing to restore here)
        pass

    def append(self, layer):
        raise TypeError('bp_wts follows the bandpass stack; assign an array to replace it')


class _SameObject(object):
    """Equal only to a wrapper of the very same object (which it keeps alive: its identity cannot be handed to another object)."""
    __slots__ = ('obj',)

    def __init__(self, obj):
        self.obj = obj

    def __eq__(self, other):
        return isinstance(other, _SameObject) and other.obj is self.obj

    def __ne__(self, other):
        return not self.__eq__(other)

    __hash__ = None


class _LazyGradients(dict):
    """InterferometerArray.gradient: {gradient_mode: (3, nbl, nchan, n_acc)} like the reference's, but the per-snapshot gradient blocks that
    observe() left in the device gradient cube (reserve()) are only fetched -- and stacked, once, instead of the reference's per-snapshot
    dstack (interferometry.py:6385-6393) -- when the entry is read.  Sharded runs that gather the gradients on the device never fetch them."""

    def __init__(self, owner):
        dict.__init__(self)
        self._owner = owner

    def _materialise(self, key):
        ia = self._owner
        if key == ia.gradient_mode and ia._grad and not dict.__contains__(self, key):
            blocks = []
            for i, g in enumerate(ia._grad):
                if isinstance(g, _DeviceSlot):
                    g = ia._ctx.get_vis(slot=g.slot, want_grad=True, complex64=(g.dtype == NP.complex64))[1]
                    ia._grad[i] = g
                blocks.append(g)
            dict.__setitem__(self, key, NP.stack(blocks, axis=3))

    def invalidate(self, key):
        dict.pop(self, key, None)

    def __getitem__(self, key):
        self._materialise(key)
        return dict.__getitem__(self, key)

    def get(self, key, default=None):
        self._materialise(key)
        return dict.get(self, key, default)

    def __contains__(self, key):
        return dict.__contains__(self, key) or (key == self._owner.gradient_mode and bool(self._owner._grad))

    def __bool__(self):
        return dict.__len__(self) > 0 or bool(self._owner._grad)

    def keys(self):
        ks = list(dict.keys(self))
        if self._owner.gradient_mode is not None and self._owner._grad and self._owner.gradient_mode not in ks:
            ks.append(self._owner.gradient_mode)
        return ks

    def items(self):
        return [(k, self[k]) for k in self.keys()]

    def values(self):
        return [self[k] for k in self.keys()]

    def __iter__(self):
        return iter(self.keys())

    def __len__(self):
        return len(self.keys())


class _CatalogROI(object):
    """Entry of ``InterferometerArray.obs_catalog_indices`` (interferometry.py:6377) for a snapshot whose sky was formed on the device
    from the resident catalogue: the integer index list -- and the direction cosines behind ``geometric_delays`` -- are fetched when
    they are read (prisim_hip_catalog_roi forms that snapshot's region of interest again; it is deterministic).  ``size`` / ``len``
    are known from the snapshot's own read-back.  Behaves like the int64 array the reference stores."""

    def __init__(self, owner, obs, lst, pc_dircos, size, frame=None):
        self._owner, self._obs, self._lst, self._pc = owner, obs, float(lst), NP.array(pc_dircos, dtype=NP.float64)
        self._frame = frame
        self.size = int(size)
        self.shape = (self.size,)
        self.ndim = 1
        self.dtype = NP.dtype(NP.int64)
        self._idx = self._dc = None

    def fetch(self):
        if self._idx is None:
            self._idx, self._dc = self._owner._ctx.catalog_roi(self._obs, self._lst, self._pc, frame=self._frame)
            self._owner = self._obs = None
        return self._idx

    def dircos(self):
        self.fetch()
        return self._dc

    def __array__(self, dtype=None, copy=None):
        out = self.fetch()
        return out if dtype is None else out.astype(dtype)

    def __len__(self):
        return self.size

    def __getitem__(self, key):
        return self.fetch()[key]

    def __iter__(self):
        return iter(self.fetch())


class LazyGeometricDelays(object):
    """Stand-in for one entry of ``InterferometerArray.geometric_delays`` (the reference stores the full
    nsrc x nbl matrix per snapshot, interferometry.py:6287-6291 -- 4.9 GB at HERA-350 x 1e4 sources).
    Materialised on demand by ``numpy.asarray(obj)``: tau = dc . bl^T / c.  `dircos`: the (nsrc, 3) array, or the snapshot's
    _CatalogROI (the direction cosines then come from the device when first needed)."""

    def __init__(self, baselines, dircos, dtype):
        self._bl, self._dc, self._dtype = baselines, dircos, dtype
        self.shape = ((dircos.size if isinstance(dircos, _CatalogROI) else dircos.shape[0]), baselines.shape[0])

    def __array__(self, dtype=None, copy=None):
        dc = self._dc.dircos() if isinstance(self._dc, _CatalogROI) else self._dc
        out = DLY.geometric_delay(self._bl, dc, altaz=False, hadec=False, dircos=True).astype(self._dtype)
        return out if dtype is None else out.astype(dtype)


def _available_host_bytes():
    """Host memory that can be claimed without swapping: MemAvailable of /proc/meminfo (free pages + reclaimable cache), else the strictly
    free pages (SC_AVPHYS_PAGES), else None."""
    try:
        with open('/proc/meminfo') as f:
            for line in f:
                if line.startswith('MemAvailable:'):
                    return int(line.split()[1]) * 1024
    except (OSError, ValueError, IndexError):
        pass
    try:
        return os.sysconf('SC_AVPHYS_PAGES') * os.sysconf('SC_PAGE_SIZE')
    except (ValueError, OSError, AttributeError):
        return None


def _lst_and_jd(timeobj, lst):
    """Resolve (jd, lst_deg) from the accepted forms of ``timeobj`` (interferometry.py:6113, 6395)."""
    if hasattr(timeobj, 'sidereal_time'):
        st = timeobj.sidereal_time('apparent')
        lst_deg = float(getattr(st, 'deg', st))
        return float(timeobj.jd), lst_deg       # the lst kwarg is overwritten, as in the reference (:6113)
    if isinstance(timeobj, (tuple, list)) and len(timeobj) == 2:
        return float(timeobj[0]), float(timeobj[1])
    if isinstance(timeobj, dict) and 'jd' in timeobj and 'lst' in timeobj:
        return float(timeobj['jd']), float(timeobj['lst'])
    if isinstance(timeobj, (int, float, NP.floating, NP.integer)):
        if lst is None:
            raise ValueError('LST must be provided when timeobj is a bare Julian date.')
        return float(timeobj), float(lst)
    raise TypeError('timeobj must be a Time-like object, a (jd, lst_deg) pair, or a Julian date with lst given.')


class InterferometerArray(object):
    """Interferometer array whose snapshots are simulated on one MI355X (see module docstring).

    Attributes touched by observe() keep the reference's names and shapes:
    bp, bp_wts, Tsys, Tsysinfo, pointing_center, phase_center, geometric_delays, obs_catalog_indices,
    skyvis_freq, gradient, gradient_mode, timestamp, t_acc, t_obs, n_acc, lst.
    """

    # Astrometric model of skycoords 'radec' (module docstring; prisim_amd/frames.py): 'apparent' | 'mean' | 'date', and an optional
    # callable (jd, lst_deg, skymodel) -> (R (3, 3), beta (3,)) that replaces it (e.g. filled from astropy)
    frame_model = os.environ.get('PRISIM_FRAME_MODEL', 'apparent')
    frame_provider = None

    def __init__(self, labels, baselines, channels, telescope=None, eff_Q=0.89,
                 latitude=34.0790, longitude=0.0, altitude=0.0, skycoords='radec',
                 A_eff=NP.pi * (25.0 / 2) ** 2, pointing_coords='hadec', layout=None,
                 blgroupinfo=None, baseline_coords='localenu', freq_scale=None,
                 gaininfo=None, init_file=None, simparms_file=None, device=0):
        if init_file is not None:
            # interferometry.py:5184-5658: initialise from <init_file>.hdf5; if it cannot be opened, fall back to the arguments
            try:
                self._init_from_hdf5(init_file, device)
                return
            except (IOError, OSError) as exc:
                warnings.warn('\tinit_file provided but could not open the initialization file ({0}). Attempting to initialize '
                              'with input parameters...'.format(exc))
        if gaininfo is not None:
            raise NotImplementedError('gaininfo (instrument gains) is outside the sky-sum path')

        self.baselines = NP.asarray(baselines, dtype=NP.float64)                      # :5668-5682
        if self.baselines.ndim == 1:
            if self.baselines.size == 2:
                self.baselines = NP.hstack((self.baselines.reshape(1, -1), NP.zeros((1, 1))))
            elif self.baselines.size == 3:
                self.baselines = self.baselines.reshape(1, -1)
            else:
                raise ValueError('Baseline(s) must be a 2- or 3-column array.')
        elif self.baselines.ndim == 2:
            if self.baselines.shape[1] == 2:
                self.baselines = NP.hstack((self.baselines, NP.zeros((self.baselines.shape[0], 1))))
            elif self.baselines.shape[1] != 3:
                raise ValueError('Baseline(s) must be a 2- or 3-column array')
        else:
            raise ValueError('Baseline(s) array contains more than 2 dimensions.')
        self.baseline_lengths = NP.sqrt(NP.sum(self.baselines ** 2, axis=1))         # :5684
        self.baseline_orientations = NP.angle(self.baselines[:, 0] + 1j * self.baselines[:, 1])
        self.projected_baselines = None

        if not isinstance(labels, (list, tuple, NP.ndarray)):                         # :5688-5693
            raise TypeError('Interferometer array labels must be a list or tuple of unique identifiers')
        elif len(labels) != self.baselines.shape[0]:
            raise ValueError('Number of labels do not match the number of baselines specified.')
        self.labels = labels
        self.simparms_file = simparms_file if isinstance(simparms_file, str) else None

        if isinstance(telescope, dict):                                               # :5701-5710
            self.telescope = telescope
        else:
            self.telescope = {'id': 'vla', 'shape': 'dish', 'size': 25.0, 'ocoords': 'altaz',
                              'orientation': NP.asarray([90.0, 270.0]).reshape(1, -1), 'groundplane': None}
        self.layout = {}
        if isinstance(layout, dict):                                                  # :5713-5751
            for key in ('positions', 'coords', 'labels', 'ids'):
                if key not in layout:
                    raise KeyError('Array layout {0} missing'.format(key))
            self.layout = dict(layout)
        self.blgroups = None
        self.bl_reversemap = None
        if blgroupinfo is not None:
            if not isinstance(blgroupinfo, dict):
                raise TypeError('Input blgroupinfo must be a dictionary')
            self.blgroups = blgroupinfo['groups']
            self.bl_reversemap = blgroupinfo['reversemap']

        self.latitude, self.longitude, self.altitude = latitude, longitude, altitude
        self.vis_freq = None
        self.skyvis_freq = None
        self.vis_noise_freq = None
        self.gradient_mode = None
        self.gradient = {}
        self.gaininfo = None

        ch = NP.asarray(channels, dtype=NP.float64).ravel()                           # :5779-5788
        if (freq_scale is None) or (freq_scale in ('Hz', 'hz')):
            self.channels = ch
        elif freq_scale in ('GHz', 'ghz'):
            self.channels = ch * 1.0e9
        elif freq_scale in ('MHz', 'mhz'):
            self.channels = ch * 1.0e6
        elif freq_scale in ('kHz', 'khz'):
            self.channels = ch * 1.0e3
        else:
            raise ValueError('Frequency units must be "GHz", "MHz", "kHz" or "Hz". If not set, it defaults to "Hz"')

        nbl, nchan = self.baselines.shape[0], self.channels.size
        self._stacks = {'bp': _LayerStack(nbl, nchan, NP.ones((1, 1))), 'bp_wts': _LayerStack(nbl, nchan, NP.ones((1, 1))),
                        'Tsys': _LayerStack(nbl, nchan, NP.zeros((1, 1)))}                # :5790-5793, read through the properties below
        self._dense = {}
        self.lag_kernel = None
        self.Tsysinfo = []
        self.flux_unit = 'JY'
        self.timestamp = []
        self.t_acc = []
        self.t_obs = 0.0
        self.n_acc = 0
        self.pointing_center = NP.empty([1, 2])
        self.phase_center = NP.empty([1, 2])
        self.lst = []

        self.eff_Q = self._broadcast_bl_chan(eff_Q, 'Efficiency', lo=0.0, hi=1.0)      # :5806-5824
        self.A_eff = self._broadcast_bl_chan(A_eff, 'Effective area', lo=0.0, hi=None)  # :5826-5844

        self.vis_rms_freq = None
        self.freq_resolution = self.channels[1] - self.channels[0] if nchan > 1 else 0.0   # :5847
        self.lags = None
        self.skyvis_lag = None
        self.vis_noise_lag = None
        self.vis_lag = None
        self.obs_catalog_indices = []
        self.geometric_delays = []

        if pointing_coords in ('radec', 'hadec', 'altaz'):                            # :5856-5860
            self.pointing_coords = pointing_coords
            self.phase_center_coords = pointing_coords
        else:
            raise ValueError('Pointing center of the interferometer must be "radec", "hadec" or "altaz". Check inputs.')
        if skycoords in ('radec', 'hadec', 'altaz'):                                  # :5862-5865
            self.skycoords = skycoords
        else:
            raise ValueError('Sky coordinates must be "radec", "hadec" or "altaz". Check inputs.')
        if baseline_coords in ('equatorial', 'localenu'):                             # :5867-5870
            self.baseline_coords = baseline_coords
        else:
            raise ValueError('Baseline coordinates must be "equatorial" or "local". Check inputs.')

        # GPU context: fails loudly when the HIP library or a device is missing
        self._ctx = _abi.Context(device)
        self._ctx.set_array(self._baselines_local(), self.channels, nt_max=1)
        self._cube = []        # per-snapshot (nbl, nchan) visibilities, stacked lazily into skyvis_freq
        self._grad = []
        self._reserved = 1     # snapshot slots of the device cube (reserve())
        self._stage, self._host_cube = False, None      # reserve(host_staging=True): snapshots are copied to a pinned host cube as they finish
        self._device_in_step = False                    # device slots [0, n_acc) hold the current skyvis_freq

    def _init_from_hdf5(self, init_file, device):
        """Attributes from a file written by save() / by PRISim (interferometry.py:5186-5657; same group and dataset names,
        same KeyErrors for what is mandatory there).  The visibility cube goes back onto the device, snapshot t in slot t."""
        from . import hdf5io
        fname = init_file if init_file.endswith('.hdf5') else init_file + '.hdf5'
        if not os.path.exists(fname):
            raise IOError('no such file: ' + fname)
        try:
            f = hdf5io.File(fname, 'r')
        except hdf5io.HDF5Unavailable as exc:
            raise IOError(str(exc))
        def text(v):
            return v.decode() if isinstance(v, bytes) else str(v)
        with f:
            def get(path, default=None, required=None):
                if f.exists(path):
                    return f.read(path)
                if required:
                    raise KeyError(required)
                return default
            for key in ('header', 'telescope_parms', 'spectral_info', 'antenna_element', 'timing', 'skyparms', 'array', 'instrument',
                        'visibilities'):
                if not f.exists(key):
                    raise KeyError('Key {0} not found in init_file'.format(key))
            self.simparms_file = get('simparms/simfile')
            self.flux_unit = text(get('header/flux_unit', 'JY'))
            self.latitude = float(get('telescope_parms/latitude', 0.0))
            self.longitude = float(get('telescope_parms/longitude', 0.0))
            self.altitude = float(get('telescope_parms/altitude', 0.0))
            self.telescope = {'shape': 'delta', 'size': 1.0, 'groundplane': None}
            if f.exists('telescope_parms/id'):
                self.telescope['id'] = text(f.read('telescope_parms/id'))
            if f.exists('antenna_element/shape'):
                self.telescope['shape'] = text(f.read('antenna_element/shape'))
            if f.exists('antenna_element/size'):
                size = NP.asarray(f.read('antenna_element/size'), dtype=NP.float64)
                self.telescope['size'] = float(size) if size.ndim == 0 else size
            self.telescope['ocoords'] = text(get('antenna_element/ocoords', required='Keyword "ocoords" not found in init_file'))
            self.telescope['orientation'] = NP.asarray(get('antenna_element/orientation', required='Key "orientation" not found in init_file'),
                                                       dtype=NP.float64).reshape(1, -1)
            if f.exists('antenna_element/groundplane'):
                self.telescope['groundplane'] = float(f.read('antenna_element/groundplane'))
            self.layout = {}
            if f.exists('layout'):
                self.layout = {'positions': get('layout/positions', required='Antenna layout positions is missing'),
                               'coords': text(f.read_attr('layout/positions', 'coords')),
                               'labels': get('layout/labels', required='Layout antenna labels is missing'),
                               'ids': get('layout/ids', required='Layout antenna ids is missing')}
            self.freq_resolution = float(f.read('spectral_info/freq_resolution'))
            self.channels = NP.asarray(f.read('spectral_info/freqs'), dtype=NP.float64)
            self.lags = get('spectral_info/lags')
            self.bp = NP.asarray(get('spectral_info/bp', required='Key "bp" not found in init_file'))
            self.bp_wts = NP.asarray(get('spectral_info/bp_wts', NP.ones_like(self.bp)))
            self.pointing_coords = text(get('skyparms/pointing_coords', 'hadec'))
            self.phase_center_coords = text(get('skyparms/phase_center_coords', self.pointing_coords))
            self.skycoords = text(get('skyparms/skycoords', 'radec'))
            self.lst = NP.asarray(f.read('skyparms/LST')).ravel().tolist()
            self.pointing_center = NP.asarray(f.read('skyparms/pointing_center'))
            self.phase_center = NP.asarray(f.read('skyparms/phase_center'))
            self.timestamp = NP.asarray(get('timing/timestamps', required='Key "timestamps" not found in init_file')).tolist()
            self.t_acc = NP.asarray(get('timing/t_acc', required='Key "t_acc" not found in init_file')).ravel().tolist()
            self.t_obs = float(f.read('timing/t_obs'))
            self.n_acc = int(f.read('timing/n_acc'))
            labels = get('array/labels', required='Key "labels" not found in init_file')
            if labels.dtype.names:
                self.labels = [tuple(text(x) for x in rec) for rec in labels.tolist()]
            else:
                self.labels = [text(x) for x in labels.tolist()]
            self.baselines = NP.asarray(get('array/baselines', required='Key "baselines" not found in init_file'), dtype=NP.float64)
            self.baseline_coords = text(get('array/baseline_coords', 'localenu'))
            self.projected_baselines = get('array/projected_baselines')
            self.baseline_lengths = NP.sqrt(NP.sum(self.baselines ** 2, axis=1))
            self.baseline_orientations = NP.angle(self.baselines[:, 0] + 1j * self.baselines[:, 1])
            self.A_eff = NP.asarray(f.read('instrument/effective_area'))
            self.eff_Q = NP.asarray(f.read('instrument/efficiency'))
            self.Tsysinfo = []
            if f.exists('instrument/Trx'):
                trx, t0, f0, sp = (NP.asarray(f.read('instrument/' + k)).ravel() for k in ('Trx', 'Tant0', 'f0', 'spindex'))
                tnet = NP.asarray(get('instrument/Tnet', NP.full(trx.size, -999.0))).ravel()
                for i in range(trx.size):
                    self.Tsysinfo += [{'Trx': float(trx[i]), 'Tant': {'T0': float(t0[i]), 'f0': float(f0[i]), 'spindex': float(sp[i])},
                                       'Tnet': float(tnet[i]) if tnet[i] > 0 else None}]
            self.Tsys = NP.asarray(get('instrument/Tsys', NP.zeros((self.baselines.shape[0], self.channels.size))))
            self.vis_rms_freq = get('visibilities/freq_spectrum/rms')
            self.vis_freq = get('visibilities/freq_spectrum/vis')
            skyvis = get('visibilities/freq_spectrum/skyvis', required='Key "skyvis" not found in init_file')
            self.vis_noise_freq = get('visibilities/freq_spectrum/noise')
            self.vis_lag = get('visibilities/delay_spectrum/vis')
            self.skyvis_lag = get('visibilities/delay_spectrum/skyvis')
            self.vis_noise_lag = get('visibilities/delay_spectrum/noise')
            self.gradient_mode, self.gradient = None, {}
            if f.exists('gradients/baseline'):
                self.gradient_mode = 'baseline'
                self.gradient = {'baseline': f.read('gradients/baseline')}
        self.gaininfo = None
        self.blgroups = None
        self.bl_reversemap = None
        self.lag_kernel = None
        self.obs_catalog_indices = []
        self.geometric_delays = []
        self._cube, self._grad = [], []
        self._reserved = max(int(self.n_acc), 1)
        self._ctx = _abi.Context(device)
        self._ctx.set_array(self._baselines_local(), self.channels, nt_max=self._reserved)
        self._stage, self._host_cube = False, None
        self.skyvis_freq = skyvis
        self._upload_cube()

    def _baselines_local(self):
        """The baselines in the local East-North-Up frame the sky-sum works in: as given, or rotated from the equatorial frame at the
        array's latitude (baseline_coords='equatorial'; interferometry.py:6151-6153 does this at every observe())."""
        if getattr(self, 'baseline_coords', 'localenu') == 'equatorial':
            return GEOM.xyz2enu(self.baselines, self.latitude, 'degrees')
        return self.baselines

    def reserve(self, n_acc, host_staging=False):
        """Allocate `n_acc` snapshot slots in the device visibility cube so that every observe() also leaves its result
        resident on the GPU (slot = snapshot index) for a later allgather() / device-side delay transform.  Not in the
        reference (its cube is a host array grown by dstack, interferometry.py:6384-6393).

        host_staging=True: the reference's product is skyvis_freq ON THE HOST (:6384-6393).  Every observe() then also enqueues the
        download of its snapshot into a page-locked host cube on a copy stream, behind its own sky-sum and beside the next one's
        (prisim_hip_get_vis_async): reading skyvis_freq afterwards only waits for the last copy instead of pulling the whole cube
        over PCIe in one serial tail.  Falls back to the lazy download when the pinned cube cannot be allocated."""
        n_acc = int(n_acc)
        if n_acc < 1:
            raise ValueError('n_acc must be positive')
        if self.n_acc > 0:
            raise RuntimeError('reserve() must be called before the first observe()')
        self._ctx.set_array(self._baselines_local(), self.channels, nt_max=n_acc)
        self._reserved = n_acc
        self._stage, self._host_cube = bool(host_staging), None
        self._restore_external_beam()
        self._catalog_key = None                       # (set_array drops the resident catalogue)

    def _reset_device_array(self):
        """set_array of the (changed) baselines after snapshots have been observed: what lives only on the device is fetched first --
        the catalogue-resident index lists and the gradient blocks observe() left in the device gradient cube (prisim_hip_set_array
        releases that cube) -- and the catalogue is uploaded again by the next observe()."""
        self._materialise_catalog_state()
        if self.gradient_mode is not None and isinstance(self.gradient, _LazyGradients):
            self.gradient.get(self.gradient_mode)
        self._ctx.set_array(self._baselines_local(), self.channels, nt_max=self._reserved)
        self._restore_external_beam()
        self._catalog_key = None

    def _cull_order(self, alt_deg, fwhm_deg):
        """Permutation that lists every run of sources of one size by decreasing altitude, or None when nothing could be culled
        (no source sizes, short baselines, more than 8 runs, or the sources already are in that order)."""
        if fwhm_deg is None or fwhm_deg.size < 2:
            return None
        kmax = NP.log(2.0) * (2.0 * NP.sin(0.5 * NP.radians(float(NP.max(fwhm_deg))))) ** 2
        lmax = float(NP.max(self.baseline_lengths)) if self.baseline_lengths.size else 0.0
        fmin = float(NP.min(NP.abs(self.channels)))
        if not kmax * (lmax * fmin / C_LIGHT) ** 2 >= 18.0:
            return None
        cuts = NP.flatnonzero(fwhm_deg[1:] != fwhm_deg[:-1]) + 1
        if cuts.size >= 8:
            return None
        order = NP.arange(fwhm_deg.size)
        bounds = NP.concatenate(([0], cuts, [fwhm_deg.size]))
        for lo, hi in zip(bounds[:-1], bounds[1:]):
            order[lo:hi] = lo + NP.argsort(-alt_deg[lo:hi], kind='stable')
        return None if NP.array_equal(order, NP.arange(order.size)) else order

    def _ensure_host_cube(self, dtype):
        """The page-locked host cube of reserve(host_staging=True) (allocated on first use), or None when staging is off, unavailable,
        or the cube holds another dtype."""
        if not getattr(self, '_stage', False):
            return None
        if self._host_cube is None:
            need = self._reserved * self.baselines.shape[0] * self.channels.size * NP.dtype(dtype).itemsize
            avail = _available_host_bytes()
            # Page-locked memory cannot be swapped or reclaimed, and reading skyvis_freq afterwards stacks the snapshots into a second
            # (pageable) cube of the same size beside it: the budget is 2 x need, against what the kernel says can be made available
            # (MemAvailable: free + reclaimable page cache -- a box that has just read a large catalog is not short of memory)
            if avail is not None and 2 * need > avail:
                warnings.warn('host staging switched off: the pinned host cube and its (nbl, nchan, n_acc) copy would need {0:.1f} GiB of '
                              '{1:.1f} GiB available'.format(2 * need / 2.0 ** 30, avail / 2.0 ** 30))
                self._stage = False
                return None
            try:
                self._host_cube = _abi.host_empty((self._reserved, self.baselines.shape[0], self.channels.size), dtype)
            except (MemoryError, _abi.PrisimHipError, OSError) as exc:
                warnings.warn('host staging switched off: the pinned host cube could not be allocated ({0})'.format(exc))
                self._stage = False
                return None
        if self._host_cube.dtype != NP.dtype(dtype):
            return None                      # a run that mixes memsave and full precision: this snapshot takes the lazy path
        return self._host_cube

    def _stage_download(self, slot, dtype):
        """Enqueue the asynchronous download of device slot `slot` into the pinned host cube; False when staging is off / unavailable."""
        hc = self._ensure_host_cube(dtype)
        if hc is None:
            return False
        self._ctx.get_vis_async(slot, hc[slot])
        return True

    def set_external_beam(self, beam, beam_freqs_hz, spec_interp='cubic', chromatic=True, select_freq=None):
        """Use a tabulated HEALPix (RING, local zenith-angle/azimuth frame) power beam [npix, nfreq] for all following
        observe() calls that do not pass roi_info (scripts/run_prisim.py:489-494, 2091-2103: interpolation of log10(beam) in
        frequency and on the sphere, per-channel peak normalisation, float32 storage) -- evaluated on the GPU per snapshot."""
        m = PB.spectral_interp_matrix(beam_freqs_hz, self.channels, kind=spec_interp, chromatic=chromatic, select_freq=select_freq)
        self._ctx.set_external_beam(beam, m)
        self._extbeam = (NP.asarray(beam), m)

    def _restore_external_beam(self):
        if getattr(self, '_extbeam', None) is not None:
            self._ctx.set_external_beam(*self._extbeam)

    def comm_setup(self, comm_uid, nranks, rank, selftest=True):
        """Build the RCCL communicator of this array's context (once) and run its self-test: a 1 MiB all-gather of a rank-dependent
        pattern verified on the host of every rank (prisim_hip_comm_selftest), so that a communicator that cannot move data raises HERE
        (PrisimHipError) and not as a silently wrong cube on disk.  Returns True; callers that can talk to the other ranks combine the
        outcomes of the SELF-TEST (driver.run does, over the rendezvous: selftest=False here, then comm_selftest()) so that every rank
        stops when one fails.  comm_init itself is a collective (ncclCommInitRank): a rank whose init fails must END -- its peers are
        inside that collective and only the launcher, seeing the exit, can stop them -- so its error is never swallowed into a vote."""
        if not getattr(self, '_comm_ready', False):
            self._ctx.comm_init(comm_uid, nranks, rank)
            if selftest:
                self._ctx.comm_selftest()
            self._comm_ready = True
        self._apply_shard_map()
        return True

    def set_shard_map(self, bl_index, nbl_total):
        """This array is one baseline shard of a larger array: bl_index (nranks, nbl_shard) = global baseline of every (padded) local row of
        every rank, negative = padding.  With it every gather (allgather, allgather_lags, allgather_gradient, allgather_cube) leaves -- and
        returns -- the cube in the GLOBAL baseline order of the unsharded array, padding dropped, put in order by a copy kernel on the
        receiving GPU (prisim_hip_set_shard_map): the order of the reference's rank-0 concatenate (scripts/run_prisim.py:2233-2242).
        Without it the gathered cubes are rank-major (nranks * nbl_shard rows)."""
        if bl_index is None:
            self._shard_map = None
        else:
            m = NP.ascontiguousarray(bl_index, dtype=NP.int64)
            if m.ndim != 2 or m.shape[1] != self.baselines.shape[0]:
                raise ValueError('bl_index must have shape (nranks, nbl_shard)')
            self._shard_map = (m, int(nbl_total))
        self._shard_map_applied = False
        if getattr(self, '_comm_ready', False):
            self._apply_shard_map()

    def _apply_shard_map(self):
        if getattr(self, '_shard_map_applied', True):
            return
        sm = getattr(self, '_shard_map', None)
        if sm is None:
            self._ctx.set_shard_map(None, 0)
        else:
            self._ctx.set_shard_map(sm[0], sm[1])
        self._shard_map_applied = True

    def _gathered_cube(self, g, nranks, row):
        """The host copy of a gathered cube, baseline axis first like every cube of the class: (rows, row, nt) with rows = nbl_total in
        global order when a shard map is set, nranks * nbl_shard (rank-major) otherwise."""
        nt = g.shape[0]
        if getattr(self, '_shard_map', None) is not None:
            return NP.transpose(g, (1, 2, 0))                         # [t][global baseline][row]
        return NP.transpose(g.reshape(nt, nranks * self.baselines.shape[0], row), (1, 2, 0))

    def comm_selftest(self):
        self._ctx.comm_selftest()

    def allgather(self, comm_uid, nranks, rank, download=True, root=None):
        """One RCCL all-gather of the baseline shards of all ranks (equal shard sizes; replaces the reference's per-rank
        part files + rank-0 concatenate, scripts/run_prisim.py:2207, 2233-2242).  Returns (nranks*nbl, nchan, n_acc) rank-major -- or, with
        set_shard_map(), (nbl_total, nchan, n_acc) in the global baseline order, put in order on the device --, or None with
        download=False (the gathered cube then stays in HBM only: a rank that writes nothing need not pull it over PCIe).
        root = r: this and every later gather of this array (lags, gradients, noise) deliver to rank r ONLY -- the other GPUs keep no
        copy of the whole cube (SURVEY 8(e) gather_to_root); download must then be False everywhere else."""
        if self._reserved < self.n_acc:
            raise RuntimeError('reserve(n_acc) must be called before observing to keep the cube on the device')
        if root is not None and download and rank != root:
            raise ValueError('with root = {0} only that rank can download the gathered cube'.format(root))
        self.comm_setup(comm_uid, nranks, rank)
        self._ctx.set_gather_root(root)
        # complex64 on the wire only when EVERY snapshot was observed with memsave (host arrays and _DeviceSlot placeholders both
        # carry their dtype); a run that mixes precisions, or has no snapshot yet, gathers complex128
        c64 = bool(self._cube) and all(NP.dtype(sn.dtype) == NP.complex64 for sn in self._cube)
        self._ctx.allgather(self.n_acc, complex64=c64)
        if not download:
            return None
        return self._gathered_cube(self._ctx.get_gathered(self.n_acc, nranks), nranks, self.channels.size)      # [t][rank][b][f] | [t][bl][f]

    def allgather_lags(self, nranks, download=True):
        """All-gather of the delay spectra of the baseline shards (the FFT is along frequency, so every rank transforms
        its own shard and the spectra are exchanged like the visibilities).  Call after allgather() and delay_transform().  Spectra
        that delay_transform() left resident on the device go GPU -> GPU (prisim_hip_allgather_lags); host-side spectra take the
        place of the visibilities in the device cube slots for the exchange.  Returns (nranks*nbl, nlag, n_acc)."""
        if not getattr(self, '_comm_ready', False):
            raise RuntimeError('allgather() must be called first (it sets up the communicator)')
        if getattr(self, '_lag_resident', None) is not None:
            nt, nout = self._lag_resident
            self._refresh_resident_lags()
            self._ctx.allgather_lags(nt)
            if not download:
                return None
            return self._gathered_cube(self._ctx.get_gathered(nt, nranks, row=nout), nranks, nout)             # [t][rank][b][lag] | [t][bl][lag]
        if self.skyvis_lag is None:
            raise RuntimeError('delay_transform() must be called first')
        if self.skyvis_lag.shape != (self.baselines.shape[0], self.channels.size, self.n_acc):
            raise NotImplementedError('host-side delay spectra are exchanged through the visibility slots: this needs nlag == nchan (pad = 0, 1, 2, ...)')
        return self.allgather_cube(self.skyvis_lag, nranks, download=download)

    def allgather_gradient(self, nranks, download=True):
        """All-gather of the baseline-gradient cubes of the shards (gradient_mode='baseline'): the reference concatenates them over the
        baseline chunks (interferometry.py:8349-8350).  The gradient blocks observe() left in the device gradient cube go GPU -> GPU
        (prisim_hip_allgather_grad).  Call after allgather().  Returns (3, nranks*nbl, nchan, n_acc), or None with download=False."""
        if not getattr(self, '_comm_ready', False):
            raise RuntimeError('allgather() must be called first (it sets up the communicator)')
        if self.gradient_mode is None or len(self._grad) != self.n_acc:
            raise RuntimeError('every snapshot must have been observed with gradient_mode set')
        if self._reserved < self.n_acc:
            raise RuntimeError('reserve(n_acc) must be called before observing to keep the gradient cube on the device')
        c64 = all(NP.dtype(g.dtype) == NP.complex64 for g in self._grad)
        self._ctx.allgather_grad(self.n_acc, complex64=c64)
        if not download:
            return None
        g = self._ctx.get_gathered_grad(self.n_acc, nranks)             # [t][rank][k][b][f], or [t][k][global baseline][f] with a shard map
        if getattr(self, '_shard_map', None) is not None:
            return NP.transpose(g, (1, 2, 3, 0))
        nbl, nchan = self.baselines.shape[0], self.channels.size
        return NP.transpose(g, (2, 1, 3, 4, 0)).reshape(3, nranks * nbl, nchan, self.n_acc)

    def allgather_cube(self, cube, nranks, download=True):
        """All-gather of a host-side (nbl, nchan, n_acc) cube of this shard that has no device copy (the thermal noise, host-side delay
        spectra): it takes the place of the visibilities in the device cube slots for the exchange, and the visibilities go back afterwards.
        Call after allgather().  Returns (nranks*nbl, nchan, n_acc), or None with download=False."""
        if not getattr(self, '_comm_ready', False):
            raise RuntimeError('allgather() must be called first (it sets up the communicator)')
        cube = NP.asarray(cube)
        if cube.shape != (self.baselines.shape[0], self.channels.size, self.n_acc):
            raise ValueError('cube must have the shape of the visibility cube (nbl, nchan, n_acc)')
        _ = self.skyvis_freq                                           # make sure the host owns the visibilities before their slots are reused
        for t in range(self.n_acc):
            self._ctx.set_vis(NP.ascontiguousarray(cube[:, :, t], dtype=NP.complex128), slot=t)
        self._ctx.allgather(self.n_acc, complex64=False)
        g = self._ctx.get_gathered(self.n_acc, nranks) if download else None
        for t in range(self.n_acc):                                    # and the visibilities go back into their slots
            self._ctx.set_vis(NP.asarray(self.skyvis_freq[:, :, t], dtype=NP.complex128), slot=t)
        if g is None:
            return None
        return self._gathered_cube(g, nranks, self.channels.size)

    # ------------------------------------------------------------------------------------------
    def _broadcast_bl_chan(self, value, what, lo, hi):
        nbl, nchan = self.baselines.shape[0], self.channels.size
        if isinstance(value, (int, float)):
            if value < lo or (hi is not None and value > hi):
                raise ValueError('{0} value of interferometer is invalid.'.format(what))
            return value * NP.ones((nbl, nchan))
        elif isinstance(value, (list, tuple, NP.ndarray)):
            value = NP.asarray(value)
            if NP.any(value < lo) or (hi is not None and NP.any(value > hi)):
                raise ValueError('One or more values of {0} found to be outside the valid range.'.format(what))
            if value.size == nbl:
                return NP.repeat(value.reshape(-1, 1), nchan, axis=1)
            elif value.size == nchan:
                return NP.repeat(value.reshape(1, -1), nbl, axis=0)
            elif value.size == nbl * nchan:
                return value.reshape(-1, nchan)
            raise ValueError('{0} values of interferometers incompatible with the number of interferometers and/or '
                             'frequency channels.'.format(what))
        raise TypeError('{0} values of interferometers must be provided as a scalar, list, tuple or numpy array.'.format(what))

    # ------------------------------------------------------------------------------------------
    def _stack_bandpass(self, bandpass):
        """interferometry.py:5993-6024."""
        bandpass = NP.asarray(bandpass)
        nbl, nchan = self.baselines.shape[0], self.channels.size
        if bandpass.ndim == 1:
            if bandpass.size != nchan:
                raise ValueError('Specified bandpass incompatible with the number of frequency channels')
            layer = bandpass.reshape(1, -1)                                # one row for every baseline (dense on read)
        elif bandpass.ndim == 2:
            if bandpass.shape[1] != nchan:
                raise ValueError('Specified bandpass incompatible with the number of frequency channels')
            elif bandpass.shape[0] != nbl:
                raise ValueError('Specified bandpass incompatible with the number of interferometers')
            layer = bandpass
        elif bandpass.ndim == 3:
            if bandpass.shape[1] != nchan:
                raise ValueError('Specified bandpass incompatible with the number of frequency channels')
            elif bandpass.shape[0] != nbl:
                raise ValueError('Specified bandpass incompatible with the number of interferometers')
            elif bandpass.shape[2] != 1:
                raise ValueError('Bandpass can have only one layer for this instance of accumulation.')
            layer = bandpass
        else:
            raise ValueError('Specified bandpass has too many dimensions')
        self._append_layer('bp', layer)                                      # :6019-6022
        if self._stacks.get('bp') is not None:
            cur = self._stacks.get('bp_wts')
            if not (isinstance(cur, _OnesStack) and cur.parent is self._stacks['bp']):
                self._stacks['bp_wts'] = self._stacks['bp'].ones_like()      # :6024
            self._dense.pop('bp_wts', None)
        else:
            self.bp_wts = NP.ones_like(self.bp)

    def _stack_tsys(self, Tsysinfo, bpcorrect):
        """interferometry.py:6026-6086."""
        nbl, nchan = self.baselines.shape[0], self.channels.size
        if not isinstance(Tsysinfo, dict):
            raise TypeError('Input Tsysinfo must be a dictionary')
        Tsys = None
        if Tsysinfo.get('Tnet', None) is not None:
            Tsys = Tsysinfo['Tnet']
        else:
            try:
                Tsys = Tsysinfo['Trx'] + Tsysinfo['Tant']['T0'] * (self.channels / Tsysinfo['Tant']['f0']) ** Tsysinfo['Tant']['spindex']
            except KeyError:
                raise KeyError('One or more keys not found in input Tsysinfo')
            Tsys = Tsys.reshape(1, -1)                                     # one row for every baseline (:6046; dense on read)
        self.Tsysinfo += [Tsysinfo]
        if bpcorrect is not None:
            if not isinstance(bpcorrect, NP.ndarray):
                raise TypeError('Input specifying bandpass correction must be a numpy array')
            if bpcorrect.size == nchan:
                bpcorrect = bpcorrect.reshape(1, -1)
            elif bpcorrect.size == nbl:
                bpcorrect = bpcorrect.reshape(-1, 1)
            elif bpcorrect.size == nbl * nchan:
                bpcorrect = bpcorrect.reshape(-1, nchan)
            else:
                raise ValueError('Input bpcorrect has dimensions incompatible with the number of baselines and frequencies')
            Tsys = Tsys * bpcorrect
        if isinstance(Tsys, (int, float)):
            if Tsys < 0.0:
                raise ValueError('Tsys found to be negative.')
            layer = NP.full((1, 1), float(Tsys))
        elif isinstance(Tsys, (list, tuple, NP.ndarray)):
            Tsys = NP.asarray(Tsys)
            if NP.any(Tsys < 0.0):
                raise ValueError('Tsys should be non-negative.')
            if Tsys.size == 1:
                layer = NP.full((1, 1), float(Tsys.ravel()[0]))
            elif Tsys.shape == (1, nchan) or (Tsys.size == nchan and Tsys.size != nbl):
                layer = Tsys.reshape(1, -1)
            elif Tsys.size == nbl:                                        # (as in the reference, nbl wins when nbl == nchan)
                layer = Tsys.reshape(-1, 1)
            elif Tsys.size == nchan:
                layer = Tsys.reshape(1, -1)
            elif Tsys.size == nbl * nchan:
                layer = Tsys.reshape(-1, nchan)
            else:
                raise ValueError('Specified Tsys has incompatible dimensions with the number of baselines and/or number of frequency channels.')
        else:
            raise TypeError('Tsys should be a scalar, list, tuple, or numpy array')
        self._append_layer('Tsys', layer)                                    # :6082-6086

    # ------------------------------------------------------------------------------------------
    # ------------------------------------------------------------------------------------------
    # Device-resident catalogue (prisim_hip_set_catalog, ABI 0.4).  The reference re-derives the sky at every observe() from a sky
    # model that does not change over a run (scripts/run_prisim.py:2165-2207; interferometry.py:6223-6247 store_prev_skymodel_file
    # exists because that is expensive).  Here the model is uploaded the first time it is seen and every later snapshot's
    # hadec -> altaz -> dircos, region of interest, flux spectra and beam x flux are formed on the GPU.

    def _catalog_fingerprint(self, skymodel):
        """What decides whether the catalogue resident on the device still IS this sky model: identity of the object and the CONTENT of
        everything that was uploaded (an in-place edit -- ``skymodel.flux_ref *= 2`` between two observe() calls, a Monte-Carlo loop over
        realisations -- must be seen; the reference re-reads the sky model at every call).  One pass of sums over the nsrc-sized vectors
        (a microsecond or two per 1e4 sources each) and over a strided sample of a spectrum table."""
        def digest(a, sample=False):
            # ONE integer reduction over the BIT PATTERNS (a wrapping 64-bit sum: any edit that is not an exchange of values moves it): exact, NaN-safe, and a plain numpy
            # loop -- NOT a BLAS dot product: OpenBLAS answers a 1e5-element dot with every core it sees (64 threads on a box whose cgroup
            # grants 16), and the scheduler then throttles the whole process for tens of milliseconds at a time -- measured: a config-4 shard
            # through observe_batch went from 1.04 to 1.2-1.8 x the kernel-only time with a dot product here.
            if a is None:
                return None
            if type(a) is not NP.ndarray:
                a = NP.asarray(a)
            if a.dtype != NP.float64:
                if a.dtype.kind not in 'fiu':
                    return (a.shape, str(a.dtype))
                a = a.astype(NP.float64)
            flat = a.reshape(-1)
            if sample and flat.size > (1 << 20):          # a spectrum table: 65536 evenly spaced samples (plus the shape) instead of gigabytes
                flat = flat[::flat.size // (1 << 16)]
            if not flat.flags.c_contiguous:
                flat = NP.ascontiguousarray(flat)
            u = flat.view(NP.uint64)
            return (a.shape, int(NP.add.reduce(u)))
        def g(k):
            return getattr(skymodel, k, None)      # (attributes, class-level defaults and properties alike)
        ref_freq = g('ref_freq')
        frozen = getattr(skymodel, '_frozen', None)
        if frozen is not None:
            # prisim_amd.skymodel.SkyModel.freeze(): the arrays are private read-only copies -- their identity stands for their content (the key
            # holds them, so no identity can be reused); anything assigned or made writeable since falls through to the content pass
            cur = (g('location'), g('flux_ref'), g('spindex'), g('spectrum'), g('frequency'), g('src_shape'))
            if len(frozen) == len(cur) and all(c is f and (c is None or not c.flags.writeable) for c, f in zip(cur, frozen)):
                return (id(skymodel), self.skycoords, g('spec_type'), _SameObject(frozen), None if ref_freq is None else float(NP.sum(ref_freq)),
                        self.channels.size, float(self.channels[0]), float(self.channels[-1]), self._reserved)
        return (id(skymodel), self.skycoords, g('spec_type'), digest(skymodel.location), digest(g('flux_ref')), digest(g('spindex')),
                None if ref_freq is None else float(NP.sum(ref_freq)), digest(g('spectrum'), sample=True), digest(g('frequency')),
                digest(g('src_shape')), self.channels.size, float(self.channels[0]), float(self.channels[-1]), self._reserved)

    def close(self):
        """Release the GPU context.  Class state that still lives on the device is fetched first: obs_catalog_indices / geometric_delays of
        snapshots formed from the resident catalogue, snapshots and gradient blocks parked in device slots."""
        if getattr(self, '_ctx', None) is None:
            return
        self._materialise_catalog_state()
        if getattr(self, '_cube', None):
            self.skyvis_freq
        if isinstance(self.gradient, _LazyGradients) and self.gradient_mode is not None:
            self.gradient.get(self.gradient_mode)
        self._ctx.close()

    def invalidate_catalog(self):
        """Forget the sky model resident on the device: the next observe() uploads it again (for edits the fingerprint cannot see,
        e.g. a custom generate_spectrum whose parameters changed)."""
        self._materialise_catalog_state()
        self._catalog_key = None

    def _catalog_ready(self, skymodel):
        """True when `skymodel` is (now) the catalogue resident on the device.  PRISIM_CATALOG=0 switches the path off (A/B: every
        snapshot's sky is then formed on the host and uploaded, as before round 5)."""
        if os.environ.get('PRISIM_CATALOG', '1') == '0' or not hasattr(self._ctx, 'set_catalog'):
            return False
        key = self._catalog_fingerprint(skymodel)
        if getattr(self, '_catalog_key', None) == key:
            return True
        if getattr(self, '_catalog_refused', None) == id(skymodel):
            return False                               # this sky model ran out of device memory on the resident path (observe())
        self._materialise_catalog_state()              # lazy class state of the previous catalogue is fetched while it is still there
        location = NP.asarray(skymodel.location, dtype=NP.float64).reshape(-1, 2)
        nsrc, nchan = location.shape[0], self.channels.size
        fwhm = None
        src_shape = getattr(skymodel, 'src_shape', None)
        if src_shape is not None:                                                      # :6258, 6267
            src_shape = NP.asarray(src_shape, dtype=NP.float64)
            fwhm = NP.sqrt(src_shape[:, 0] * src_shape[:, 1])
        powerlaw = (getattr(skymodel, 'spec_type', None) == 'func' and all(hasattr(skymodel, a) for a in ('flux_ref', 'spindex', 'ref_freq')))
        try:
            if powerlaw:
                self._ctx.set_catalog(location, self.skycoords, flux_ref=skymodel.flux_ref, spindex=skymodel.spindex,
                                      ref_freq_hz=float(skymodel.ref_freq), fwhm_deg=fwhm)
            else:
                if nsrc * nchan * 8 > (16 << 30):
                    return False                       # a spectrum table this large stays on the per-snapshot path (ROI rows only)
                spectra = NP.asarray(skymodel.generate_spectrum(ind=NP.arange(nsrc), frequency=self.channels, interp_method='pchip'),
                                     dtype=NP.float64).reshape(-1, nchan)              # :6249, once for the whole catalogue
                self._ctx.set_catalog(location, self.skycoords, flux_spectrum=spectra, fwhm_deg=fwhm)
        except MemoryError:
            warnings.warn('the sky model does not fit on the device as a resident catalogue; it takes the per-snapshot upload path')
            self._catalog_refused = id(skymodel)
            return False
        self._catalog_key = key
        self._catalog_obs_cache = None
        return True

    def _catalog_obs(self, pb_info, pc_altaz, roi_radius, roi_center):
        """(prisim_obs, beam pointing direction cosines) of a snapshot.  The struct is cached while nothing it depends on changes."""
        if getattr(self, '_extbeam', None) is not None:
            key = ('ext', roi_radius, roi_center, self.latitude)
            cache = getattr(self, '_catalog_obs_cache', None)
            if cache is None or cache[0] != key:
                cache = (key, self._ctx.make_obs(self.latitude, roi_radius, roi_center, use_external_beam=True))
                self._catalog_obs_cache = cache
            return cache[1], None
        if pb_info is None:
            skey = (id(self.telescope), float(pc_altaz[0]), float(pc_altaz[1]))
            spec = getattr(self, '_beam_spec_cache', None)
            if spec is None or spec[0] != skey:
                spec = (skey, PB.device_beam_spec(self.telescope, pointing_info=None, pointing_center=pc_altaz,
                                                  first_frequency_hz=float(self.channels[0])))                  # :6252
                self._beam_spec_cache = spec
            kind, dia, bpc, ext = spec[1]
        else:
            kind, dia, bpc, ext = PB.device_beam_spec(self.telescope, pointing_info=pb_info, pointing_center=pc_altaz,
                                                      first_frequency_hz=float(self.channels[0]))
        if pb_info is not None or (ext is not None and 'beamformer' in ext):
            return self._ctx.make_obs(self.latitude, roi_radius, roi_center, beam_kind=kind, diameter_m=dia, ext=ext), bpc
        key = (kind, dia, id(self.telescope), roi_radius, roi_center, self.latitude)
        cache = getattr(self, '_catalog_obs_cache', None)
        if cache is None or cache[0] != key:
            cache = (key, self._ctx.make_obs(self.latitude, roi_radius, roi_center, beam_kind=kind, diameter_m=dia, ext=ext))
            self._catalog_obs_cache = cache
        return cache[1], bpc

    def _materialise_catalog_state(self):
        """Fetch the lazy per-snapshot class state that lives in the device-resident catalogue (obs_catalog_indices, the direction
        cosines behind geometric_delays) before the catalogue goes away: a new sky model, or set_array (reserve, conjugate ...)."""
        for entry in getattr(self, 'obs_catalog_indices', []):
            if isinstance(entry, _CatalogROI):
                entry.fetch()
        self._catalog_key = None

    def _snapshot_frame(self, jd, lst, skymodel):
        """(R, beta) of one snapshot: catalogue frame of `skymodel` -> local East, North, Up (interferometry.py:6174-6180; prisim_amd/frames.py).
        The same pair goes to the device (prisim_snapshot.cel2enu / aberr_beta) and to the host path (geometry.frame_dircos)."""
        if self.frame_provider is not None and self.skycoords == 'radec':
            rot, beta = self.frame_provider(jd, lst, skymodel)
            return NP.asarray(rot, dtype=NP.float64).reshape(3, 3), NP.asarray(beta, dtype=NP.float64).reshape(3)
        key = (self.skycoords, float(lst), float(self.latitude), float(jd), getattr(skymodel, 'epoch', None), self.frame_model)
        cache = getattr(self, '_frame_cache', None)
        if cache is None or cache[0] != key:
            cache = (key, FRAMES.snapshot_frame(self.skycoords, lst, self.latitude, jd=jd, epoch=getattr(skymodel, 'epoch', None),
                                                model=self.frame_model))
            self._frame_cache = cache
        return cache[1]

    def _check_gradient_mode(self, gradient_mode):
        if gradient_mode is not None:                                                 # :6306-6311
            if not isinstance(gradient_mode, str):
                raise TypeError('Input gradient_mode must be a string')
            if gradient_mode.lower() not in ['baseline', 'skypos', 'frequency']:
                raise ValueError('Invalid value specified in input gradient_mode')
            if gradient_mode.lower() != 'baseline':
                raise NotImplementedError('only gradient_mode="baseline" is computed (as in the reference)')
            if self.gradient_mode is None:
                self.gradient_mode = gradient_mode
        return gradient_mode is not None

    def _append_pointing(self, pointing_center, lst):
        """Pointing = phase centre of one more snapshot (:6103-6108, 6155-6164): (alt-az degrees, ENU direction cosines)."""
        pc = NP.asarray(pointing_center, dtype=NP.float64).reshape(1, -1)
        if pc.size != 2:
            raise ValueError('pointing_center must be a 2-element vector')
        # pointing_center / phase_center are (n_acc, 2) arrays like the reference's (:6103-6108, grown there by NP.vstack: O(n) per
        # snapshot); here both are views of buffers that grow by doubling
        n = len(self.timestamp)
        for name in ('pointing_center', 'phase_center'):
            cur = getattr(self, name)
            buf = self.__dict__.get('_buf_' + name)
            if n == 0 or buf is None or cur.base is not buf or cur.shape[0] != n or buf.shape[0] <= n:
                newbuf = NP.empty((max(16, 2 * (n + 1)), 2))
                if n > 0:
                    newbuf[:n] = NP.asarray(cur, dtype=NP.float64).reshape(-1, 2)[:n]
                buf = self.__dict__['_buf_' + name] = newbuf
            buf[n] = pc[0]
            setattr(self, name, buf[:n + 1])
        # (a drift scan points at one (HA, Dec) for hours: the conversion of the previous snapshot is kept while nothing it depends on changes)
        key = (self.pointing_coords, float(pc[0, 0]), float(pc[0, 1]), lst if self.pointing_coords == 'radec' else None, self.latitude)
        cache = getattr(self, '_pointing_cache', None)
        if cache is not None and cache[0] == key:
            return cache[1], cache[2]
        pc_altaz = self.pointing_center[-1, :]                                        # :6155-6162
        if self.pointing_coords == 'hadec':
            pc_altaz = GEOM.hadec2altaz(self.pointing_center[-1, :], self.latitude, units='degrees')
        elif self.pointing_coords == 'radec':
            pc_altaz = GEOM.hadec2altaz(NP.asarray([lst - self.pointing_center[-1, 0], self.pointing_center[-1, 1]]),
                                        self.latitude, units='degrees')
        pc_altaz = NP.array(pc_altaz, dtype=NP.float64)
        pc_dircos = GEOM.altaz2dircos(pc_altaz, 'degrees').ravel()                    # :6164
        self._pointing_cache = (key, pc_altaz, pc_dircos)
        return pc_altaz, pc_dircos

    def _unpark(self, slot):
        """Fetch whatever snapshot (and gradient block) still lives only in device slot `slot` before the slot is overwritten."""
        for i, snap in enumerate(self._cube):
            if isinstance(snap, _DeviceSlot) and snap.slot == slot:
                self._cube[i] = self._ctx.get_vis(slot=slot, complex64=(snap.dtype == NP.complex64))
        for i, g in enumerate(self._grad):
            if isinstance(g, _DeviceSlot) and g.slot == slot:
                self._grad[i] = self._ctx.get_vis(slot=slot, want_grad=True, complex64=(g.dtype == NP.complex64))[1]

    @staticmethod
    def _roi_defaults(roi_radius, roi_center):
        if roi_radius is None:                                                        # :6204-6216
            roi_radius = 90.0
        if roi_center is None:
            roi_center = 'zenith'
        elif (roi_center != 'zenith') and (roi_center != 'pointing_center'):
            raise ValueError('Center of region of interest, roi_center, must be set to "zenith" or "pointing_center".')
        return roi_radius, roi_center

    def observe(self, timeobj, Tsysinfo, bandpass, pointing_center, skymodel,
                t_acc, pb_info=None, brightness_units=None, bpcorrect=None,
                roi_info=None, roi_radius=None, roi_center=None, lst=None,
                gradient_mode=None, memsave=False, vmemavail=None,
                store_prev_skymodel_file=None):
        """Simulate one snapshot (interferometry.py:5874-6410).  See the reference docstring for the
        argument meaning; vmemavail is accepted and ignored (the GPU kernel never materialises the nsrc x nbl x nchan matrix, so there
        is no memory-shortage path).  store_prev_skymodel_file (:6223-6247: the previous call's sky on disk, so that an unchanged sky
        is not re-derived) is accepted too: what it buys is had without a file -- the sky model handed in stays resident on the GPU
        from the first call that sees it, and every later snapshot's geometry, region of interest and beam x flux are formed there
        (prisim_hip_set_catalog / prisim_hip_observe_catalog)."""
        self._stack_bandpass(bandpass)
        self._stack_tsys(Tsysinfo, bpcorrect)
        jd, lst = _lst_and_jd(timeobj, lst)                                           # :6113
        pc_altaz, pc_dircos = self._append_pointing(pointing_center, lst)             # :6103-6108, 6155-6164

        for attr in ('location', 'generate_spectrum'):                                 # :6171 (duck-typed SkyModel)
            if not hasattr(skymodel, attr):
                raise TypeError('skymodel should be an instance of class SkyModel.')
        nbl, nchan = self.baselines.shape[0], self.channels.size
        datatype = NP.complex64 if memsave else NP.complex128                         # :6182-6185
        want_grad = self._check_gradient_mode(gradient_mode)
        prec = _abi.PRISIM_FP32 if memsave else _abi.PRISIM_FP64
        slot = self.n_acc if self.n_acc < self._reserved else 0

        frame = self._snapshot_frame(jd, lst, skymodel)                               # :6174-6180 as one rotation + aberration vector
        if roi_info is None and self._catalog_ready(skymodel):
            # ---- the sky model is resident on the device: geometry, ROI, spectra, beam x flux and the sky-sum in ONE call ----
            roi_radius, roi_center = self._roi_defaults(roi_radius, roi_center)
            obs, bpc = self._catalog_obs(pb_info, pc_altaz, roi_radius, roi_center)
            if slot != self.n_acc:
                self._unpark(slot)
            try:
                nroi = int(self._ctx.observe_catalog(obs, [lst], pc_dircos, bpc, precision=prec, want_grad=want_grad, slot0=slot,
                                                     frames=[frame])[0])
            except MemoryError:
                # the resident path sizes its beam x flux for a whole chunk of snapshots; a catalogue that ran through the per-snapshot
                # upload before must still run: drop the resident copy and take that path (ROI rows only)
                warnings.warn('device memory exhausted on the resident-catalogue path; this sky model continues on the per-snapshot upload path')
                self._catalog_refused = id(skymodel)
                self._materialise_catalog_state()
                nroi = None
            roi = None if nroi is None else _CatalogROI(self, obs, lst, pc_dircos, nroi, frame)
        else:
            roi = None
        if roi is not None:
            if nroi == 0:                                                             # :6378-6382 (the device slot holds zeros)
                warnings.warn('No sources found in the catalog within matching radius. Simply populating the observed visibilities and/or gradients with noise.')
            self.geometric_delays = self.geometric_delays + [LazyGeometricDelays(self._baselines_local(), roi,
                                                                                 NP.float32 if memsave else NP.float64)]   # :6287-6291
            self.obs_catalog_indices = self.obs_catalog_indices + [roi]               # :6377
            have_sky = True
        else:
            have_sky = self._upload_snapshot_sky(skymodel, frame, pc_altaz, pc_dircos, pb_info, roi_info, roi_radius, roi_center, memsave)
            if have_sky:
                if slot != self.n_acc:                                   # (a fresh reserved slot holds nothing: no O(n_acc) scan per snapshot)
                    self._unpark(slot)
                self._ctx.compute(precision=prec, want_grad=want_grad, slot=slot)

        if have_sky:
            self._device_in_step = slot == self.n_acc and (self.n_acc == 0 or getattr(self, '_device_in_step', False))
            if slot == self.n_acc:
                # the snapshot stays in its own slot of the device cube: no synchronous download (1 GB and 20 ms per HERA-350
                # snapshot; 4 GB with the gradient blocks); with host staging its copy to the pinned host cube is queued behind the
                # sky-sum, on the copy stream.  The gradient blocks stay in the device gradient cube until `gradient` is read.
                skyvis, skyvis_gradient = _DeviceSlot(slot, datatype), (_DeviceSlot(slot, datatype) if want_grad else None)
                skyvis.staged = self._stage_download(slot, datatype)
            else:
                res = self._ctx.get_vis(slot=slot, want_grad=want_grad, complex64=memsave)
                skyvis, skyvis_gradient = res if want_grad else (res, None)
        else:                                                                         # :6378-6382
            warnings.warn('No sources found in the catalog within matching radius. Simply populating the observed visibilities and/or gradients with noise.')
            skyvis = NP.zeros((nbl, nchan), dtype=datatype)
            skyvis_gradient = NP.zeros((3, nbl, nchan), dtype=datatype) if want_grad else None
            if self.n_acc < self._reserved:                                            # the snapshot's device slot says the same
                self._ctx.set_vis(NP.zeros((nbl, nchan), dtype=NP.complex128), slot=self.n_acc)
                self._device_in_step = self.n_acc == 0 or getattr(self, '_device_in_step', False)
            else:
                self._device_in_step = False

        self._cube.append(skyvis)                                                     # :6384-6393
        self._skyvis_cache = None
        if want_grad:
            self._grad.append(skyvis_gradient)
            if not isinstance(self.gradient, _LazyGradients):
                self.gradient = _LazyGradients(self)
            self.gradient.invalidate(gradient_mode)                                   # stacked again when it is next read

        self.timestamp = self.timestamp + [jd]                                        # :6395-6399
        self.t_acc = self.t_acc + [t_acc]
        self.t_obs += t_acc
        self.n_acc += 1
        self.lst = self.lst + [lst]

    def observe_batch(self, timeobjs, Tsysinfo, bandpass, pointing_centers, skymodel, t_acc, pb_info=None, bpcorrect=None,
                      roi_radius=None, roi_center=None, gradient_mode=None, memsave=False):
        """K snapshots of ONE sky model in one call -- what the reference's callers write as a loop over observe()
        (scripts/run_prisim.py:2165-2207, interferometry.py:6641-6647).  Not in the reference.  With the sky model resident on the
        device the geometry of all K snapshots is formed first (one small read-back for the lot) and their skies and sky-sums are then
        queued back to back without a host synchronisation; arrays of at most 256 baselines put the K sky-sums into ONE launch
        (prisim_hip_observe_catalog).  The class state afterwards is what K calls of observe() leave.

        timeobjs          K time objects as observe() takes them ((jd, lst_deg) pairs, Time-likes ...)
        Tsysinfo          one dictionary for every snapshot, or a list of K
        bandpass          (nchan,) / (nbl, nchan) for every snapshot, or a list of K
        pointing_centers  (2,) for every snapshot or (K, 2)
        t_acc             scalar or K values
        Falls back to K calls of observe() when the catalogue path is not available (PRISIM_CATALOG=0, no free device slots)."""
        k = len(timeobjs)
        if k == 0:
            return
        tsys_l = Tsysinfo if isinstance(Tsysinfo, (list, tuple)) else [Tsysinfo] * k
        bp_l = bandpass if isinstance(bandpass, (list, tuple)) else [bandpass] * k
        pcs = NP.broadcast_to(NP.asarray(pointing_centers, dtype=NP.float64).reshape(-1, 2), (k, 2))
        tacc_l = list(NP.broadcast_to(NP.asarray(t_acc, dtype=NP.float64).ravel(), (k,)))
        if len(tsys_l) != k or len(bp_l) != k:
            raise ValueError('Tsysinfo / bandpass lists must have one entry per snapshot')
        for attr in ('location', 'generate_spectrum'):
            if not hasattr(skymodel, attr):
                raise TypeError('skymodel should be an instance of class SkyModel.')
        if self.n_acc == 0 and self._reserved < k:
            self.reserve(k, host_staging=getattr(self, '_stage', False))
        batched = self.n_acc + k <= self._reserved and (pb_info is None) and self._catalog_ready(skymodel)
        if not batched:
            for t in range(k):
                self.observe(timeobjs[t], tsys_l[t], bp_l[t], pcs[t], skymodel, float(tacc_l[t]), pb_info=pb_info, bpcorrect=bpcorrect,
                             roi_radius=roi_radius, roi_center=roi_center, gradient_mode=gradient_mode, memsave=memsave)
            return
        want_grad = self._check_gradient_mode(gradient_mode)
        roi_radius, roi_center = self._roi_defaults(roi_radius, roi_center)
        datatype = NP.complex64 if memsave else NP.complex128
        prec = _abi.PRISIM_FP32 if memsave else _abi.PRISIM_FP64
        # The per-snapshot class state grown before the device call (timestamp, pointing / phase centre rows, bandpass and Tsys layers) is
        # rolled back if anything fails -- a changed beam specification, a device error -- so that the instance stays aligned with n_acc.
        state0 = (self.pointing_center, self.phase_center, getattr(self, '_pointing_cache', None), self.timestamp, list(self.Tsysinfo),
                  dict(self._stacks), {n: (None if st is None else list(st.layers)) for n, st in self._stacks.items()}, dict(self._dense))
        jds, lsts, pc_dcs, bpcs, frames, obs = [], [], [], [], [], None
        try:
            for t in range(k):
                self._stack_bandpass(bp_l[t])
                self._stack_tsys(tsys_l[t], bpcorrect)
                jd, lst = _lst_and_jd(timeobjs[t], None)
                pc_altaz, pc_dircos = self._append_pointing(pcs[t], lst)
                self.timestamp = self.timestamp + [jd]                                # (_append_pointing starts a fresh array on an empty list)
                o, bpc = self._catalog_obs(None, pc_altaz, roi_radius, roi_center)
                obs = o if obs is None else obs
                if o is not obs:
                    raise RuntimeError('the beam specification changed inside a batch')
                jds.append(jd); lsts.append(lst); pc_dcs.append(pc_dircos); bpcs.append(pc_dircos if bpc is None else bpc)
                frames.append(self._snapshot_frame(jd, lst, skymodel))
            slot0 = self.n_acc
            host_cube = self._ensure_host_cube(datatype) if getattr(self, '_stage', False) else None
            counts = self._ctx.observe_catalog(obs, lsts, NP.asarray(pc_dcs), NP.asarray(bpcs), precision=prec, want_grad=want_grad, slot0=slot0,
                                               host_cube=host_cube, frames=frames)
        except Exception as exc:
            self.pointing_center, self.phase_center, self._pointing_cache, self.timestamp, self.Tsysinfo, stacks, layers, dense = state0
            self._stacks = stacks
            for n, st in stacks.items():
                if st is not None:
                    st.layers = layers[n]
            self._dense = dense
            if not isinstance(exc, MemoryError):
                raise
            # the resident path ran out of device memory: this sky model continues on the per-snapshot upload path (ROI rows only)
            warnings.warn('device memory exhausted on the resident-catalogue path; this sky model continues on the per-snapshot upload path')
            self._catalog_refused = id(skymodel)
            self._materialise_catalog_state()
            for t in range(k):
                self.observe(timeobjs[t], tsys_l[t], bp_l[t], pcs[t], skymodel, float(tacc_l[t]), pb_info=pb_info, bpcorrect=bpcorrect,
                             roi_radius=roi_radius, roi_center=roi_center, gradient_mode=gradient_mode, memsave=memsave)
            return
        base_bl = self._baselines_local()
        for t in range(k):
            roi = _CatalogROI(self, obs, lsts[t], pc_dcs[t], int(counts[t]), frames[t])
            if counts[t] == 0:
                warnings.warn('No sources found in the catalog within matching radius. Simply populating the observed visibilities and/or gradients with noise.')
            self.geometric_delays = self.geometric_delays + [LazyGeometricDelays(base_bl, roi, NP.float32 if memsave else NP.float64)]
            self.obs_catalog_indices = self.obs_catalog_indices + [roi]
            snap = _DeviceSlot(slot0 + t, datatype)
            snap.staged = host_cube is not None
            self._cube.append(snap)
            if want_grad:
                self._grad.append(_DeviceSlot(slot0 + t, datatype))
            self.t_acc = self.t_acc + [float(tacc_l[t])]
            self.t_obs += float(tacc_l[t])
            self.lst = self.lst + [lsts[t]]
        self._device_in_step = slot0 == 0 or getattr(self, '_device_in_step', False)
        self.n_acc += k
        self._skyvis_cache = None
        if want_grad:
            if not isinstance(self.gradient, _LazyGradients):
                self.gradient = _LazyGradients(self)
            self.gradient.invalidate(gradient_mode)

    def _upload_snapshot_sky(self, skymodel, frame, pc_altaz, pc_dircos, pb_info, roi_info, roi_radius, roi_center, memsave):
        """The snapshot's sky formed on the HOST and uploaded (roi_info given, PRISIM_CATALOG=0, or a spectrum table too large to keep
        resident): interferometry.py:6171-6283 statement by statement, with the astropy / GEOM coordinate chain of :6174-6180 as the
        snapshot's frame (R, beta) applied by geometry.frame_dircos -- the host statement of the device's cat_source().  Returns False when
        the region of interest is empty."""
        nchan = self.channels.size
        location = NP.asarray(skymodel.location, dtype=NP.float64).reshape(-1, 2)
        dc_all = GEOM.frame_dircos(GEOM.catalog_unitvec(location, self.skycoords), frame[0], frame[1])     # :6174-6180, 6263
        pb = None
        if roi_info is not None:                                                      # :6189-6202
            if ('ind' not in roi_info) or ('pbeam' not in roi_info):
                raise KeyError('Both "ind" and "pbeam" keys must be present in dictionary roi_info')
            m2 = NP.arange(0)
            if (roi_info['ind'] is not None) and (roi_info['pbeam'] is not None):
                m2 = NP.asarray(roi_info['ind']).ravel()
                if m2.size > 0:
                    try:
                        pb = NP.asarray(roi_info['pbeam']).reshape(-1, nchan)
                    except ValueError:
                        raise ValueError('Number of columns of primary beam in key "pbeam" of dictionary roi_info must be equal to number of frequency channels.')
                    if m2.size != pb.shape[0]:
                        raise ValueError('Values in keys ind and pbeam in must carry same number of elements.')
        else:                                                                         # :6204-6216
            roi_radius, roi_center = self._roi_defaults(roi_radius, roi_center)
            m2 = GEOM.roi_select(dc_all, roi_center, roi_radius, pc_dircos)       # :6210-6216 on the direction cosines
        if len(m2) == 0:
            return False
        dircos_roi = dc_all[m2, :]                                                # :6219, 6263 (unconditional, Q4)
        # flux spectra (:6249).  A power-law sky model (spec_type 'func') is described to the device by its nsrc-sized
        # flux_ref / spindex vectors and S = flux_ref (f / ref_freq)^spindex is formed there; anything else goes through
        # generate_spectrum on the host, as in the reference.
        powerlaw = (getattr(skymodel, 'spec_type', None) == 'func' and pb is None
                    and all(hasattr(skymodel, a) for a in ('flux_ref', 'spindex', 'ref_freq')))
        if powerlaw:
            fluxes = None
            flux_ref = NP.asarray(skymodel.flux_ref, dtype=NP.float64)[m2]
            spindex = NP.asarray(skymodel.spindex, dtype=NP.float64)[m2]
            ref_freq = float(skymodel.ref_freq)
        else:
            fluxes = NP.asarray(skymodel.generate_spectrum(ind=m2, frequency=self.channels, interp_method='pchip'),
                                dtype=NP.float64).reshape(-1, nchan)
            flux_ref = spindex = ref_freq = None
        fwhm = None
        src_shape = getattr(skymodel, 'src_shape', None)
        if src_shape is not None:                                                 # :6258, 6267
            src_shape = NP.asarray(src_shape, dtype=NP.float64)
            fwhm = NP.sqrt(src_shape[m2, 0] * src_shape[m2, 1])
        # Upload order.  The sum over sources does not care about their order, the taper culling of the library does: for every
        # baseline group it skips the LEADING sources of a run of one source size whose weight is provably below the tolerance,
        # and those are the sources nearest the zenith (long baselines resolve them out: b_perp ~ |b|).  So when anything can be
        # culled at all -- kappa_max (|b|_max f_min / c)^2 >= 18 -- each run is listed by decreasing altitude.  Class state
        # (obs_catalog_indices, geometric_delays) keeps the catalog order.
        up = self._cull_order(dircos_roi[:, 2], fwhm)                             # (n = sin(altitude): the same order)
        dircos_up = dircos_roi
        if up is not None:
            dircos_up, fwhm = dircos_roi[up], fwhm[up]
            flux_ref, spindex = (flux_ref[up], spindex[up]) if flux_ref is not None else (None, None)
            fluxes = fluxes[up] if fluxes is not None else None
            pb = pb[up] if pb is not None else None
        if pb is not None:
            # supplied beam (ROI_parameters path): pbfluxes = pb * fluxes on the device (:6254)
            self._ctx.set_sky(dircos_up, pb, pc_dircos, fwhm_deg=fwhm, fluxes=fluxes)
        elif getattr(self, '_extbeam', None) is not None:
            self._ctx.set_sky_external_analytic(dircos_up, flux_ref, spindex, ref_freq, pc_dircos, fwhm_deg=fwhm, flux_spectrum=fluxes)
        else:
            kind, dia, bpc, ext = PB.device_beam_spec(self.telescope, pointing_info=pb_info, pointing_center=pc_altaz,
                                                      first_frequency_hz=float(self.channels[0]))                   # :6252
            self._ctx.set_sky_analytic(dircos_up, flux_ref, spindex, ref_freq, kind, dia, bpc, pc_dircos, fwhm_deg=fwhm,
                                       flux_spectrum=fluxes, ext=ext)
        self.geometric_delays = self.geometric_delays + [LazyGeometricDelays(self._baselines_local(), dircos_roi,
                                                                             NP.float32 if memsave else NP.float64)]   # :6287-6291
        self.obs_catalog_indices = self.obs_catalog_indices + [m2]                # :6377
        return True

    # skyvis_freq: (nbl, nchan, n_acc), time fastest, like the reference (:6385-6390)
    @property
    def skyvis_freq(self):
        if not getattr(self, '_cube', None):
            return self._skyvis_override
        if getattr(self, '_skyvis_cache', None) is None:
            staged = [isinstance(sn, _DeviceSlot) and sn.staged for sn in self._cube]
            if any(staged):
                self._ctx.wait_downloads()                              # the copies ran under the later snapshots' sky-sums
            for i, snap in enumerate(self._cube):
                if isinstance(snap, _DeviceSlot):                       # first read: fetch the device-resident snapshots
                    if snap.staged:
                        self._cube[i] = self._host_cube[snap.slot]
                    else:
                        self._cube[i] = self._ctx.get_vis(slot=snap.slot, complex64=(snap.dtype == NP.complex64))
            # the reference's layout: (nbl, nchan, n_acc) C-contiguous, time fastest (:6385-6390) -- a host-side transpose of the
            # per-snapshot arrays; skyvis_freq_snapshots() hands out the snapshot-major cube without it
            self._skyvis_cache = NP.stack(self._cube, axis=2)
        return self._skyvis_cache

    def skyvis_freq_snapshots(self):
        """The visibilities snapshot-major, (n_acc, nbl, nchan) -- the layout they are computed and downloaded in.  With
        reserve(host_staging=True) this is the page-locked host cube itself (no copy: the downloads already ran under the later
        snapshots' sky-sums and only the last one is waited for); otherwise the snapshots are fetched and stacked."""
        if not getattr(self, '_cube', None):
            return None if self._skyvis_override is None else NP.moveaxis(NP.asarray(self._skyvis_override), 2, 0)
        hc = getattr(self, '_host_cube', None)
        if hc is not None and all(isinstance(sn, _DeviceSlot) and sn.staged and sn.slot == i for i, sn in enumerate(self._cube)):
            self._ctx.wait_downloads()
            return hc[:len(self._cube)]
        return NP.moveaxis(self.skyvis_freq, 2, 0)

    @skyvis_freq.setter
    def skyvis_freq(self, value):
        self._skyvis_override = value
        self._device_in_step = False      # the device slots no longer hold this cube (callers that re-upload set the flag again)
        if value is not None:
            value = NP.asarray(value)
            self._cube = [value[:, :, i] for i in range(value.shape[2])]
            self._skyvis_cache = value

    def _upload_cube(self):
        """Put the host visibility cube into the device slots (needs reserve(n_acc) slots) and mark the two as in step."""
        cube = self.skyvis_freq
        for t in range(cube.shape[2]):
            self._ctx.set_vis(NP.ascontiguousarray(cube[:, :, t], dtype=NP.complex128), slot=t)
        self._device_in_step = True

    # bp, bp_wts, Tsys: (nbl, nchan) before the first snapshot, (nbl, nchan, n_acc) afterwards, like the reference's; kept as
    # per-snapshot layers (_LayerStack) and materialised on read.  Assigning an array replaces the layers.
    def _get_stacked(self, name):
        if name not in getattr(self, '_dense', {}):
            self.__dict__.setdefault('_dense', {})[name] = self._stacks[name].dense()
        return self._dense[name]

    def _set_stacked(self, name, value):
        self.__dict__.setdefault('_dense', {})[name] = NP.asarray(value)
        self.__dict__.setdefault('_stacks', {})[name] = None

    bp = property(lambda self: self._get_stacked('bp'), lambda self, v: self._set_stacked('bp', v))
    bp_wts = property(lambda self: self._get_stacked('bp_wts'), lambda self, v: self._set_stacked('bp_wts', v))
    Tsys = property(lambda self: self._get_stacked('Tsys'), lambda self, v: self._set_stacked('Tsys', v))

    def _append_layer(self, name, layer):
        """One more snapshot layer ((nbl | 1, nchan | 1), or (nbl, nchan, 1)) of a stacked attribute."""
        st = self._stacks.get(name)
        if st is not None:
            st.append(layer)
            self._dense.pop(name, None)
            return
        cur = self._dense[name]                                          # the attribute was assigned a dense array: grow that
        nbl, nchan = self.baselines.shape[0], self.channels.size
        layer = NP.asarray(layer)
        layer = NP.array(NP.broadcast_to(layer if layer.ndim == 3 else layer.reshape(layer.shape[0] if layer.ndim == 2 else 1, -1)[:, :, NP.newaxis],
                                         (nbl, nchan, 1)))
        self._dense[name] = layer if cur.ndim == 2 else NP.dstack((cur, layer))

    def _window_layers(self):
        """bp * bp_wts per snapshot in compact form: a list of (1 | nbl, nchan) arrays, or None when only dense arrays exist."""
        a, b = self._stacks.get('bp'), self._stacks.get('bp_wts')
        if a is None or b is None or len(a.layers) != len(b.layers) or not a.layers:
            return None
        return [x * y for x, y in zip(a.layers, b.layers)]

    # skyvis_lag / lag_kernel: computed on the GPU by delay_transform(); when the visibility cube is resident on the device the
    # spectra stay there too and are fetched on first read (config 5: 120 GB that a sharded run exchanges GPU -> GPU instead)
    def _refresh_resident_lags(self):
        """The context holds ONE resident spectrum buffer; when another transform (a DelaySpectrum of this array, a power-spectrum
        fetch) has overwritten it since delay_transform(), the transform is run again before the spectra are read or exchanged."""
        if getattr(self._ctx, '_dt_generation', None) != getattr(self, '_lag_gen', None) and getattr(self, '_lag_args', None) is not None:
            nt, w, pad = self._lag_args
            self._ctx.delay_transform_device(nt, bpwts=w, pad=pad, want_lag=True)
            self._lag_gen = getattr(self._ctx, '_dt_generation', None)

    @property
    def skyvis_lag(self):
        if getattr(self, '_lag_resident', None) is not None and getattr(self, '_skyvis_lag', None) is None:
            nt, nout = self._lag_resident
            self._refresh_resident_lags()
            self._skyvis_lag = NP.transpose(self._ctx.get_lags(0, nt), (1, 2, 0))
        return getattr(self, '_skyvis_lag', None)

    @skyvis_lag.setter
    def skyvis_lag(self, value):
        self._skyvis_lag = value
        self._lag_resident = None

    def skyvis_lag_rows(self, rows):
        """Delay spectra (len(rows), nlag, n_acc) of selected baselines without fetching the whole cube from the device."""
        if getattr(self, '_lag_resident', None) is not None and getattr(self, '_skyvis_lag', None) is None:
            nt, nout = self._lag_resident
            self._refresh_resident_lags()
            return NP.transpose(self._ctx.get_lags(0, nt, rows=rows), (1, 2, 0))
        if self.skyvis_lag is None:
            raise RuntimeError('delay_transform() must be called first')
        return self.skyvis_lag[NP.asarray(rows), :, :]

    @property
    def lag_kernel(self):
        if getattr(self, '_lag_kernel', None) is None and getattr(self, '_lag_kernel_maker', None) is not None:
            self._lag_kernel = self._lag_kernel_maker()
            self._lag_kernel_maker = None
        return getattr(self, '_lag_kernel', None)

    @lag_kernel.setter
    def lag_kernel(self, value):
        self._lag_kernel = value
        self._lag_kernel_maker = None

    # ------------------------------------------------------------------------------------------
    def observing_run(self, pointing_init, skymodel, t_acc, duration, channels,
                      bpass, Tsys, lst_init, roi_radius=None, roi_center=None,
                      mode='track', pointing_coords=None, freq_scale=None,
                      brightness_units=None, verbose=True, memsave=False):
        """Extended observing run in 'track' or 'drift' mode built from snapshots
        (interferometry.py:6414-6657; argument checks :6510-6601, LST ramp :6607, pointing :6611-6633).
        lst_init is in HOURS, as the reference's LST ramp (:6607) treats it."""
        if isinstance(pointing_init, list):
            pointing_init = NP.asarray(pointing_init)
        elif not isinstance(pointing_init, NP.ndarray):
            raise TypeError('pointing_init must be a list or numpy array.')
        if pointing_init.size != 2:
            raise ValueError('pointing_init must be a 2-element vector.')
        pointing_init = pointing_init.ravel().astype(NP.float64)
        if not isinstance(t_acc, (int, float)):
            raise TypeError('t_acc must be a scalar integer or float.')
        if t_acc <= 0.0:
            raise ValueError('t_acc must be positive.')
        if not isinstance(duration, (int, float)):
            raise TypeError('duration must be a scalar integer or float.')
        if duration <= t_acc:
            if verbose:
                warnings.warn('\t\tDuration specified to be shorter than t_acc. Will set it equal to t_acc')
            duration = t_acc
        n_acc = int(duration / t_acc)                                                  # :6536
        nbl, nchan = self.baselines.shape[0], self.channels.size
        if not isinstance(bpass, (list, tuple, NP.ndarray)):
            raise TypeError('bpass must be a list, tuple or numpy array')
        bpass = NP.asarray(bpass)
        if bpass.size == nchan:                                                        # :6561-6572
            bpass = NP.expand_dims(NP.repeat(bpass.reshape(1, -1), nbl, axis=0), axis=2)
        elif bpass.size == nbl * nchan:
            bpass = NP.expand_dims(bpass.reshape(-1, nchan), axis=2)
        elif bpass.size == nbl * nchan * n_acc:
            bpass = bpass.reshape(-1, nchan, n_acc)
        else:
            raise ValueError('Dimensions of bpass incompatible with the number of frequency channels, baselines and number of accumulations.')
        if not isinstance(Tsys, (int, float, list, tuple, NP.ndarray)):
            raise TypeError('Tsys must be a scalar, list, tuple or numpy array')
        Tsys = NP.asarray(Tsys, dtype=NP.float64).reshape(-1)                          # :6574-6598
        if Tsys.size == 1:
            Tsys = Tsys + NP.zeros((nbl, nchan, 1))
        elif Tsys.size == nchan:
            Tsys = NP.expand_dims(NP.repeat(Tsys.reshape(1, -1), nbl, axis=0), axis=2)
        elif Tsys.size == nbl:
            Tsys = NP.expand_dims(NP.repeat(Tsys.reshape(-1, 1), nchan, axis=1), axis=2)
        elif Tsys.size == nbl * nchan:
            Tsys = NP.expand_dims(Tsys.reshape(-1, nchan), axis=2)
        elif Tsys.size == nbl * nchan * n_acc:
            Tsys = Tsys.reshape(-1, nchan, n_acc)
        else:
            raise ValueError('Dimensions of Tsys incompatible with the number of frequency channels, baselines and number of accumulations.')
        if not isinstance(lst_init, (int, float)):
            raise TypeError('Starting LST should be a scalar')

        lst = (lst_init + (t_acc / 3.6e3) * NP.arange(n_acc)) * 15.0                   # :6607 (degrees)
        lst_init_deg = lst_init * 15.0
        if mode == 'track':                                                            # :6611-6622
            if pointing_coords == 'hadec':
                pointing = NP.asarray([lst_init_deg - pointing_init[0], pointing_init[1]])
            elif (pointing_coords == 'radec') or (pointing_coords is None):
                pointing = pointing_init
            elif pointing_coords == 'altaz':
                hadec = GEOM.altaz2hadec(pointing_init, self.latitude, units='degrees')
                pointing = NP.asarray([lst_init_deg - hadec[0], hadec[1]])
            else:
                raise ValueError('pointing_coords can only be set to "hadec", "radec" or "altaz".')
            self.pointing_coords = 'radec'
            self.phase_center_coords = 'radec'
        elif mode == 'drift':                                                          # :6623-6633
            if pointing_coords == 'radec':
                pointing = NP.asarray([lst_init_deg - pointing_init[0], pointing_init[1]])
            elif (pointing_coords == 'hadec') or (pointing_coords is None):
                pointing = pointing_init
            elif pointing_coords == 'altaz':
                pointing = GEOM.altaz2hadec(pointing_init, self.latitude, units='degrees')
            else:
                raise ValueError('pointing_coords can only be set to "hadec", "radec" or "altaz".')
            self.pointing_coords = 'hadec'
            self.phase_center_coords = 'hadec'
        else:
            raise ValueError('mode must be "track" or "drift"')

        jd0 = 2451545.0
        # :6641-6647 loops observe(); here the snapshots go to the device as ONE batch (observe_batch): the sky model stays resident,
        # the geometry of all snapshots is formed there first, and arrays of at most 256 baselines sum all of them in one launch
        times = [(jd0 + i * t_acc / 86400.0, float(lst[i])) for i in range(n_acc)]
        self.observe_batch(times, [{'Tnet': Tsys[:, :, i % Tsys.shape[2]]} for i in range(n_acc)],
                           [bpass[:, :, i % bpass.shape[2]] for i in range(n_acc)], pointing, skymodel, t_acc,
                           roi_radius=roi_radius, roi_center=roi_center, memsave=memsave)
        self.t_obs = duration                                                          # :6654-6655
        self.n_acc = n_acc

    # ------------------------------------------------------------------------------------------
    def generate_noise(self, seed=None, bl_offset=0, bl_index=None):
        """Thermal noise for every (baseline, channel, snapshot) from the system parameters (interferometry.py:6661-6693):
        vis_rms_freq = 2 k / sqrt(t_acc df) * Tsys / (A_eff eff_Q) / Jy (flux_unit 'JY') or Tsys / eff_Q / sqrt(t_acc df) ('K');
        vis_noise_freq = vis_rms_freq / sqrt(2) * (randn + 1j randn).  The normals are drawn on the GPU (counter-based Philox),
        reproducibly for a given `seed` (None: a fresh random seed, like the reference's global numpy RNG)."""
        if not self.timestamp:
            raise ValueError('no snapshots: call observe() first')
        eff_Q = self.eff_Q if self.eff_Q.ndim == 3 else self.eff_Q[:, :, NP.newaxis]
        A_eff = self.A_eff if self.A_eff.ndim == 3 else self.A_eff[:, :, NP.newaxis]
        t_acc = NP.asarray(self.t_acc, dtype=NP.float64)[NP.newaxis, NP.newaxis, :]
        Tsys = self.Tsys if self.Tsys.ndim == 3 else self.Tsys[:, :, NP.newaxis]
        if self.flux_unit in ('JY', 'jy', 'Jy'):
            self.vis_rms_freq = 2.0 * 1.380649e-23 / NP.sqrt(t_acc * self.freq_resolution) * (Tsys / A_eff / eff_Q) / 1.0e-26   # :6685
        elif self.flux_unit in ('K', 'k'):
            self.vis_rms_freq = 1.0 / NP.sqrt(t_acc * self.freq_resolution) * Tsys / eff_Q                                  # :6687
        else:
            raise ValueError('Flux density units can only be in Jy or K.')
        if seed is None:
            seed = int(NP.random.SeedSequence().generate_state(2, dtype=NP.uint32).astype(NP.uint64) @ NP.array([1, 1 << 32], dtype=NP.uint64))
        rms_tbf = NP.ascontiguousarray(NP.transpose(NP.broadcast_to(self.vis_rms_freq, (self.baselines.shape[0], self.channels.size,
                                                                                          len(self.timestamp))), (2, 0, 1)))
        # bl_offset / bl_index: this array's baselines inside the WHOLE array of a sharded run (the draws are keyed on the global index)
        noise = self._ctx.noise(rms_tbf, seed, bl_offset=bl_offset, bl_index=bl_index)   # (nt, nbl, nchan)
        self.vis_noise_freq = NP.transpose(noise, (1, 2, 0))                             # :6692
        self.noise_seed = seed

    def add_noise(self):
        """vis_freq = gains * skyvis_freq + vis_noise_freq with unity gains (interferometry.py:6697-6722; gain tables are out of scope)."""
        if self.vis_noise_freq is None:
            raise ValueError('generate_noise() must be called first')
        if self.gaininfo is None:
            warnings.warn('Gain table absent. Proceeding with default unity gains')
        self.vis_freq = self.skyvis_freq + self.vis_noise_freq

    # ------------------------------------------------------------------------------------------
    def _pc_dircos(self, pc, coords):
        """Phase centres (n_acc, 2|3) in `coords` -> ENU direction cosines, per snapshot (uses self.lst)."""
        pc = NP.asarray(pc, dtype=NP.float64)
        lst = NP.asarray(self.lst, dtype=NP.float64)
        if coords == 'dircos':
            if (pc.shape[1] < 2) or (pc.shape[1] > 3):
                raise ValueError('Dimensions incompatible for direction cosine positions')
            if NP.any(NP.sqrt(NP.sum(pc ** 2, axis=1)) > 1.0 + 1e-12):
                raise ValueError('direction cosines found to be exceeding unit magnitude.')
            if pc.shape[1] == 2:
                pc = NP.hstack((pc, NP.sqrt(NP.maximum(0.0, 1.0 - NP.sum(pc ** 2, axis=1))).reshape(-1, 1)))
            return pc
        if coords == 'altaz':
            return GEOM.altaz2dircos(pc, 'degrees')
        if coords == 'hadec':
            return GEOM.altaz2dircos(GEOM.hadec2altaz(pc, self.latitude, units='degrees'), 'degrees')
        if coords == 'radec':
            hadec = NP.stack((lst - pc[:, 0], pc[:, 1]), axis=1)
            return GEOM.altaz2dircos(GEOM.hadec2altaz(hadec, self.latitude, units='degrees'), 'degrees')
        raise ValueError('Invalid phase center coordinate system specified')

    def _convert_pc(self, dircos, coords):
        """ENU direction cosines -> phase-centre coordinates in `coords` (for the phase_center attribute)."""
        if coords == 'dircos':
            return dircos
        altaz = GEOM.dircos2altaz(dircos, units='degrees')
        if coords == 'altaz':
            return altaz
        hadec = GEOM.altaz2hadec(altaz, self.latitude, units='degrees')
        if coords == 'hadec':
            return hadec
        return NP.stack((NP.asarray(self.lst) - hadec[:, 0], hadec[:, 1]), axis=1)

    def phase_centering(self, phase_center=None, phase_center_coords=None, do_delay_transform=False, verbose=True):
        """Re-centre the visibility phases on new phase centre(s): V *= exp(-2 pi i f b.(s_cur - s_new)/c)
        (interferometry.py:7735-7886; arithmetic :7871-7877) on the GPU.  phase_center: (1|n_acc, 2|3) array in
        phase_center_coords ('radec', 'hadec', 'altaz', 'dircos'); None uses the pointing centres."""
        if phase_center is None:
            phase_center = self.pointing_center
            phase_center_coords = self.pointing_coords
        elif not isinstance(phase_center, NP.ndarray):
            raise TypeError('Phase center must be a numpy array')
        if phase_center_coords is None:
            raise NameError('Coordinates of phase center not provided.')
        phase_center = NP.asarray(phase_center, dtype=NP.float64)
        if phase_center.ndim == 1:
            phase_center = phase_center.reshape(1, -1)
        if phase_center.ndim != 2:
            raise ValueError('Phase center has invalid dimensions')
        n_acc = len(self.lst)
        if phase_center.shape[0] == 1:
            phase_center = NP.repeat(phase_center, n_acc, axis=0)
        elif phase_center.shape[0] != n_acc:
            raise ValueError('One phase center must be provided for every timestamp.')
        if not getattr(self, '_cube', None) and getattr(self, '_skyvis_override', None) is None:
            raise ValueError('no visibilities to rotate: call observe() first')
        cur = self._pc_dircos(self.phase_center, self.phase_center_coords)
        new = self._pc_dircos(phase_center, phase_center_coords)
        diff = cur - new                                                               # :7871

        def rotate(cube):
            cube = NP.asarray(cube, dtype=NP.complex128)
            out = NP.empty_like(cube)
            for t in range(cube.shape[2]):                 # through device cube slot 0 (host-resident cubes)
                self._ctx.set_vis(cube[:, :, t], slot=0)
                self._ctx.phase_rotate(1, diff[[t]])
                out[:, :, t] = self._ctx.get_vis(slot=0)
            return out

        resident = self._reserved >= n_acc and len(self._cube) == n_acc and all(isinstance(sn, _DeviceSlot) for sn in self._cube)
        if resident and self.vis_freq is None and self.vis_noise_freq is None:
            # the whole cube lives on the device (reserve()) and was never read: rotate it where it is, one kernel over all snapshots
            self._ctx.phase_rotate(n_acc, diff)                                         # :7877
            self._skyvis_cache = None
            for sn in self._cube:                              # host copies staged before the rotation are stale: queue them again
                if sn.staged:
                    sn.staged = self._stage_download(sn.slot, sn.dtype)
        else:
            dtype = self.skyvis_freq.dtype
            self.skyvis_freq = rotate(self.skyvis_freq).astype(dtype)                  # :7877
            if self.vis_freq is not None:
                self.vis_freq = rotate(self.vis_freq)
            if self.vis_noise_freq is not None:
                self.vis_noise_freq = rotate(self.vis_noise_freq)
            if self._reserved >= n_acc:                        # keep the device-resident cube in step
                self._upload_cube()
        self.phase_center = self._convert_pc(new, self.phase_center_coords)            # :7874-7875
        if do_delay_transform:
            self.delay_transform(verbose=verbose)

    def project_baselines(self, ref_point):
        """Baselines projected on the (u, v, w) frame of a reference direction per snapshot
        (interferometry.py:7890-7985; rotation matrix :7979-7981).  Host side: O(nbl * n_acc)."""
        if not isinstance(ref_point, dict):
            raise TypeError('Input ref_point must be a dictionary')
        if ('location' not in ref_point) or ('coords' not in ref_point):
            raise KeyError('Both keys "location" and "coords" must be specified in input dictionary ref_point')
        pc, coords = ref_point['location'], ref_point['coords']
        if not isinstance(pc, NP.ndarray):
            raise TypeError('The specified reference point must be a numpy array')
        if not isinstance(coords, str):
            raise TypeError('The specified coordinates of the reference point must be a string')
        if coords not in ['radec', 'hadec', 'altaz', 'dircos']:
            raise ValueError('Specified coordinates of reference point invalid')
        pc = pc.reshape(1, -1) if pc.ndim == 1 else pc
        if pc.ndim > 2:
            raise ValueError('Reference point has invalid dimensions')
        if (pc.shape[0] != self.n_acc) and (pc.shape[0] != 1):
            raise ValueError('Reference point has dimensions incompatible with the number of timestamps')
        if pc.shape[0] == 1:
            pc = NP.repeat(pc, self.n_acc, axis=0)
        dc = self._pc_dircos(pc, coords)
        hadec = GEOM.altaz2hadec(GEOM.dircos2altaz(dc, units='degrees'), self.latitude, units='degrees')
        ha, dec = NP.radians(hadec[:, 0]), NP.radians(hadec[:, 1])
        lat = NP.radians(self.latitude)
        e, n, u = self.baselines[:, 0], self.baselines[:, 1], self.baselines[:, 2]
        eq = NP.stack((-NP.sin(lat) * n + NP.cos(lat) * u, e, NP.cos(lat) * n + NP.sin(lat) * u), axis=1)   # GEOM.enu2xyz (:7977)
        rot = NP.asarray([[NP.sin(ha), NP.cos(ha), NP.zeros(ha.size)],
                          [-NP.sin(dec) * NP.cos(ha), NP.sin(dec) * NP.sin(ha), NP.cos(dec)],
                          [NP.cos(dec) * NP.cos(ha), -NP.cos(dec) * NP.sin(ha), NP.sin(dec)]])            # :7979-7981
        self.projected_baselines = NP.einsum('bk,jkt->bjt', eq, rot)                                      # :7985 (nbl, 3, n_acc)

    def rotate_visibilities(self, ref_point, do_delay_transform=False, verbose=True):
        """phase_centering + project_baselines about one reference point (interferometry.py:7655-7733)."""
        if not isinstance(ref_point, dict):
            raise TypeError('Input ref_point must be a dictionary')
        if ('location' not in ref_point) or ('coords' not in ref_point):
            raise KeyError('Both keys "location" and "coords" must be specified in input dictionary ref_point')
        self.phase_centering(phase_center=ref_point['location'], phase_center_coords=ref_point['coords'],
                             do_delay_transform=do_delay_transform, verbose=verbose)
        self.project_baselines(ref_point)

    def conjugate(self, ind=None, verbose=True):
        """Flip the selected baselines and conjugate their visibilities (interferometry.py:7989-8048)."""
        if ind is None:
            return
        if isinstance(ind, str):
            if ind != 'all':
                raise ValueError('Value of ind must be "all" if set to string')
            ind = NP.arange(self.baselines.shape[0])
        elif isinstance(ind, (int, NP.integer)):
            ind = [ind]
        elif not isinstance(ind, (list, NP.ndarray)):
            raise TypeError('ind must be string "all", scalar interger, list or numpy array')
        ind = NP.asarray(ind)
        if NP.any(ind >= self.baselines.shape[0]):
            raise IndexError('Out of range indices found.')
        self.labels = [tuple(reversed(self.labels[i])) if (i in ind and isinstance(self.labels[i], tuple)) else self.labels[i]
                       for i in range(len(self.labels))]
        self.baselines[ind, :] = -self.baselines[ind, :]
        self.baseline_orientations = NP.angle(self.baselines[:, 0] + 1j * self.baselines[:, 1])
        if self.skyvis_freq is not None:
            cube = NP.array(self.skyvis_freq)
            cube[ind, :, :] = cube[ind, :, :].conj()
            self.skyvis_freq = cube
        for name in ('vis_freq', 'vis_noise_freq'):
            arr = getattr(self, name)
            if arr is not None:
                arr[ind, :, :] = arr[ind, :, :].conj()
        if self.projected_baselines is not None:
            self.projected_baselines[ind, :, :] = -self.projected_baselines[ind, :, :]
        self._reset_device_array()                                                 # the resident array follows the flip
        if self._reserved >= self.n_acc and self.skyvis_freq is not None:
            self._upload_cube()

    def adopt_observation(self, shard):
        """Make this (so far unobserved) array stand for the WHOLE array of an observation of which `shard` -- another InterferometerArray, one
        rank's baseline shard -- was a part: take over everything that does not depend on the baseline (snapshots, pointings, phase centres,
        system-temperature records, the bandpass / weights / Tsys layers).  What the reference gets by concatenating the per-rank part files
        along the baseline axis (scripts/run_prisim.py:2233-2242, interferometry.py `concatenate`); here the visibility cubes arrive through
        the all-gather and are assigned by the caller (skyvis_freq, vis_freq, vis_noise_freq, skyvis_lag ...).  Per-baseline layers of a
        shard cannot stand for the other shards' baselines and are refused."""
        if self.timestamp:
            raise ValueError('adopt_observation() is for an array that has not observed anything itself')
        if not NP.array_equal(self.channels, shard.channels):
            raise ValueError('the shard was observed on another channel grid')
        nbl, nchan = self.baselines.shape[0], self.channels.size
        for name in ('bp', 'bp_wts', 'Tsys'):
            st = shard._stacks.get(name)
            if st is not None:
                if any(l.shape[0] != 1 for l in st.layers) or (NP.ndim(st.initial) == 2 and st.initial.shape[0] not in (1,)):
                    raise ValueError('{0} of the shard varies with the baseline: not enough to describe the whole array'.format(name))
                new = _LayerStack(nbl, nchan, st.initial)
                new.layers = list(st.layers)
                self._stacks[name] = new
                self.__dict__.setdefault('_dense', {}).pop(name, None)
            else:
                dense = NP.asarray(shard._dense[name])
                if dense.shape[0] > 1 and not NP.array_equal(dense, NP.broadcast_to(dense[:1], dense.shape)):
                    raise ValueError('{0} of the shard varies with the baseline: not enough to describe the whole array'.format(name))
                self._set_stacked(name, NP.array(NP.broadcast_to(dense[:1], (nbl,) + dense.shape[1:])))
        if shard.vis_rms_freq is not None:
            rms = NP.asarray(shard.vis_rms_freq)
            if rms.shape[0] > 1 and not NP.array_equal(rms, NP.broadcast_to(rms[:1], rms.shape)):
                raise ValueError('vis_rms_freq of the shard varies with the baseline: not enough to describe the whole array')
            self.vis_rms_freq = NP.array(NP.broadcast_to(rms[:1], (nbl,) + rms.shape[1:]))
        self.timestamp, self.t_acc, self.lst = list(shard.timestamp), list(shard.t_acc), list(shard.lst)
        self.t_obs, self.n_acc = shard.t_obs, shard.n_acc
        self.Tsysinfo = list(shard.Tsysinfo)
        self.pointing_center = NP.array(shard.pointing_center)
        self.phase_center = NP.array(shard.phase_center)
        self.phase_center_coords = shard.phase_center_coords
        self.obs_catalog_indices = list(shard.obs_catalog_indices)
        self.flux_unit = shard.flux_unit
        self.lags = None if shard.lags is None else NP.array(shard.lags)
        self.noise_seed = getattr(shard, 'noise_seed', None)

    def apply_gradients(self, gradient_mode=None, perturbations=None):
        """First-order change of the sky visibilities under small baseline displacements, from the gradient cube observe()
        accumulated with gradient_mode='baseline' (interferometry.py:6726-6819; arithmetic :6810-6812):

            delta V[..., b, f, t] = -2 pi i / lambda_f * sum_k db[..., k, b] * G[k, b, f, t]

        perturbations   {'baseline': array of shape (3, nbl), (nseed, 3, nbl) or (n1, n2, ..., 3, nbl)} in the units of
                        `baselines`.  Fewer than three coordinate rows are completed with zeros and more than three are cut
                        to the first three, each with a warning, as in the reference.  The caller's dictionary is left as
                        given (the reference reshapes the array inside it in place).
        Returns an (n1, ..., nbl, nchan, n_acc) complex array; a plain (3, nbl) input gives a leading axis of 1.
        """
        if gradient_mode is None:
            gradient_mode = self.gradient_mode
        if perturbations is None:
            perturbations = {gradient_mode: NP.zeros((1, 1, 1))}
        if self.gradient_mode is None or not self.gradient:
            raise AttributeError('No gradient attribute found')
        if not isinstance(perturbations, dict):
            raise TypeError('Input perturbations must be a dictionary')
        if not isinstance(gradient_mode, str):
            raise TypeError('Input gradient_mode must be a string')
        if gradient_mode not in ['baseline']:
            raise KeyError('Specified gradient mode {0} not currently supported'.format(gradient_mode))
        if gradient_mode not in perturbations:
            raise KeyError('{0} key not found in input perturbations'.format(gradient_mode))
        if gradient_mode != self.gradient_mode:
            raise ValueError('Specified gradient mode {0} not found in attribute'.format(gradient_mode))
        pert = perturbations[gradient_mode]
        if not isinstance(pert, NP.ndarray):
            raise TypeError('Perturbations must be specified as a numpy array')
        if pert.ndim < 2:
            raise ValueError('Perturbations must be two--dimensions or higher')
        if pert.ndim == 2:
            pert = pert[NP.newaxis, ...]
        lead = pert.shape[:-2]
        pert = pert.reshape(-1, pert.shape[-2], pert.shape[-1])                      # nseed x ncoord x nbl
        grad = self.gradient[gradient_mode]                                          # 3 x nbl x nchan x n_acc
        if pert.shape[2] != grad.shape[1]:
            raise ValueError('Number of {0} perturbations not equal to that in the gradient attribute'.format(gradient_mode))
        ncoord = pert.shape[1]
        if ncoord < 3:
            warnings.warn('Only {0}-dimensional coordinates specified. Proceeding with zero perturbations in other coordinate axes.'.format(ncoord))
        elif ncoord > 3:
            warnings.warn('{0}-dimensional coordinates specified. Proceeding with only the first three dimensions of coordinate axes.'.format(3))
        ncoord = min(ncoord, 3)                                                      # absent rows contribute nothing
        wavenumber = 2.0 * NP.pi * self.channels / C_LIGHT                           # 2 pi / lambda
        delta = NP.einsum('skb,kbft->sbft', pert[:, :ncoord, :], grad[:ncoord])
        delta = -1j * wavenumber.reshape(1, 1, -1, 1) * delta
        return delta.reshape(lead + grad.shape[1:])

    # ------------------------------------------------------------------------------------------
    def save(self, outfile, fmt='HDF5', tabtype='BinTableHDU', npz=True, overwrite=False, uvfits_parms=None, verbose=True):
        """Write the object to disk in PRISim's HDF5 layout (interferometry.py:8393-8863; groups and dataset names of
        :8723-8846) and, with npz, the reference's NPZ summary (:8854-8857).  HDF5 goes through the HDF5 C library
        (prisim_amd/hdf5io.py; no h5py here); FITS and UVFITS need astropy / pyuvdata and are not written."""
        if not isinstance(outfile, str):
            raise TypeError('outfile must be a string')
        if fmt.lower() not in ('hdf5', 'fits'):
            raise ValueError('Invalid output file format specified')
        if fmt.lower() == 'fits':
            raise NotImplementedError('FITS output needs astropy, which this build does not carry; use fmt="HDF5"')
        if uvfits_parms is not None:
            raise NotImplementedError('UVFITS output needs pyuvdata / astropy, which this build does not carry')
        from . import hdf5io
        filename = outfile + '.hdf5'
        if verbose:
            print('\nSaving information about interferometer...')
        tel = self.telescope
        with hdf5io.File(filename, 'w' if overwrite else 'w-') as f:
            f.create_group('header')
            f.write('header/AstroUtils#', 'none (prisim_amd)')
            f.write('header/PRISim#', 'prisim_amd')
            f.write('header/flux_unit', self.flux_unit)
            f.write('telescope_parms/latitude', float(self.latitude), attrs={'units': 'deg'})
            f.write('telescope_parms/longitude', float(self.longitude), attrs={'units': 'deg'})
            f.write('telescope_parms/altitude', float(self.altitude), attrs={'units': 'm'})
            if 'id' in tel:
                f.write('telescope_parms/id', str(tel['id']))
            f.write('spectral_info/freq_resolution', float(self.freq_resolution), attrs={'units': 'Hz'})
            f.write('spectral_info/freqs', self.channels, attrs={'units': 'Hz'})
            if self.lags is not None:
                f.write('spectral_info/lags', self.lags, attrs={'units': 's'})
            f.write('spectral_info/bp', self.bp)
            f.write('spectral_info/bp_wts', self.bp_wts)
            if self.simparms_file is not None:
                f.write('simparms/simfile', self.simparms_file)
            f.create_group('antenna_element')
            # the reference's reader insists on ocoords and orientation (:5261-5266): defaults of its constructor (:5703-5709)
            ocoords = str(tel.get('ocoords', 'altaz'))
            f.write('antenna_element/shape', str(tel.get('shape', 'delta')))
            f.write('antenna_element/ocoords', ocoords)
            f.write('antenna_element/size', NP.asarray(tel.get('size', 1.0), dtype=NP.float64), attrs={'units': 'm'})
            f.write('antenna_element/orientation', NP.asarray(tel.get('orientation', [[90.0, 270.0]]), dtype=NP.float64),      # as held (:8766)
                    attrs=({'units': 'deg'} if ocoords != 'dircos' else None))
            if tel.get('groundplane') is not None:
                f.write('antenna_element/groundplane', float(tel['groundplane']))
            if self.layout:
                f.write('layout/positions', NP.asarray(self.layout['positions'], dtype=NP.float64),
                        attrs={'units': 'm', 'coords': str(self.layout['coords'])})
                f.write('layout/labels', NP.asarray(self.layout['labels']))
                f.write('layout/ids', NP.asarray(self.layout['ids']))
            f.write('timing/t_obs', float(self.t_obs))
            f.write('timing/n_acc', int(self.n_acc))
            if self.t_acc:
                f.write('timing/t_acc', NP.asarray(self.t_acc, dtype=NP.float64))
            f.write('timing/timestamps', NP.asarray(self.timestamp))
            f.write('skyparms/pointing_coords', str(self.pointing_coords))
            f.write('skyparms/phase_center_coords', str(self.phase_center_coords))
            f.write('skyparms/skycoords', str(self.skycoords))
            f.write('skyparms/LST', NP.asarray(self.lst, dtype=NP.float64).ravel(), attrs={'units': 'deg'})
            f.write('skyparms/pointing_center', NP.asarray(self.pointing_center, dtype=NP.float64))
            f.write('skyparms/phase_center', NP.asarray(self.phase_center, dtype=NP.float64))
            labels = self.labels
            if len(labels) and isinstance(labels[0], (tuple, list)):                    # (A2, A1) antenna-pair records like the reference's
                width = max(max(len(str(a)), len(str(b))) for a, b in labels)
                labels = NP.asarray([(str(a).encode(), str(b).encode()) for a, b in labels], dtype=[('A2', 'S%d' % width), ('A1', 'S%d' % width)])
            else:
                labels = NP.asarray([str(l) for l in labels])
            f.write('array/labels', labels)
            # (the reference labels the dataset 'local-ENU' whatever baseline_coords says, :8794; an equatorial array is labelled as what it holds)
            f.write('array/baselines', self.baselines, attrs={'coords': 'equatorial-XYZ' if self.baseline_coords == 'equatorial' else 'local-ENU', 'units': 'm'})
            f.write('array/baseline_coords', str(self.baseline_coords))
            if self.projected_baselines is not None:
                f.write('array/projected_baselines', self.projected_baselines, attrs={'coords': 'eq-XYZ', 'units': 'm'})
            f.write('instrument/effective_area', self.A_eff, attrs={'units': 'm^2'})
            f.write('instrument/efficiency', self.eff_Q)
            if self.Tsysinfo:
                def col(get):
                    return NP.asarray([get(elem) for elem in self.Tsysinfo], dtype=NP.float64)
                nan = float('nan')
                f.write('instrument/Trx', col(lambda e: e.get('Trx') if e.get('Trx') is not None else nan), attrs={'units': 'K'})
                f.write('instrument/Tant0', col(lambda e: (e.get('Tant') or {}).get('T0', nan)), attrs={'units': 'K'})
                f.write('instrument/f0', col(lambda e: (e.get('Tant') or {}).get('f0', nan)), attrs={'units': 'Hz'})
                f.write('instrument/spindex', col(lambda e: (e.get('Tant') or {}).get('spindex', nan)))
                f.write('instrument/Tnet', col(lambda e: e['Tnet'] if e.get('Tnet') is not None else -999), attrs={'units': 'K'})
            f.write('instrument/Tsys', self.Tsys, attrs={'units': 'K'})
            f.create_group('visibilities/freq_spectrum')
            for name, arr in (('rms', self.vis_rms_freq), ('vis', self.vis_freq), ('skyvis', self.skyvis_freq), ('noise', self.vis_noise_freq)):
                if arr is not None:
                    f.write('visibilities/freq_spectrum/' + name, arr, attrs={'units': 'Jy'})
            f.create_group('visibilities/delay_spectrum')
            for name, arr in (('vis', self.vis_lag), ('skyvis', self.skyvis_lag), ('noise', self.vis_noise_lag)):
                if arr is not None:
                    f.write('visibilities/delay_spectrum/' + name, arr, attrs={'units': 'Jy Hz'})
            if self.gradient_mode is not None:
                for key, arr in self.gradient.items():
                    f.write('gradients/' + str(key), arr)
            if self.blgroups is not None:
                f.create_group('blgroupinfo/groups')
                f.create_group('blgroupinfo/reversemap')
                for key, members in self.blgroups.items():
                    f.write('blgroupinfo/groups/' + str(key).replace('/', '|'), NP.asarray([str(m) for m in members]))
                for key, val in (self.bl_reversemap or {}).items():
                    f.write('blgroupinfo/reversemap/' + str(key).replace('/', '|'), NP.asarray([str(val)]))
        if verbose:
            print('\tInterferometer array information written successfully to file on disk:\n\t\t{0}\n'.format(filename))
        if npz:
            keys = {'skyvis_freq': self.skyvis_freq, 'lst': self.lst, 'freq': self.channels, 'timestamp': self.timestamp,
                    'bl': self.baselines, 'bl_length': self.baseline_lengths}                              # :8857
            if (self.vis_freq is not None) and (self.vis_noise_freq is not None):                         # :8855
                keys.update({'vis_freq': self.vis_freq, 'vis_noise_freq': self.vis_noise_freq})
            NP.savez_compressed(outfile + '.npz', **keys)
        return filename

    # ------------------------------------------------------------------------------------------
    def duplicate_measurements(self, blgroups=None):
        """Re-create the redundant baselines from the simulated unique ones (interferometry.py:6823-6906): every baseline whose
        label is a key of `blgroups` is repeated once per member of its group (visibilities, gradient, baselines, per-baseline
        system parameters), labels become the members' labels, then noise is regenerated for the expanded set.  The GPU computes
        the unique baselines only; this is a host-side repeat."""
        if blgroups is None:
            blgroups = self.blgroups
        if not isinstance(blgroups, dict):
            raise TypeError('Input blgroups must be a dictionary')
        nbl = sum(len(v) for v in blgroups.values()) if self.bl_reversemap is None else len(self.bl_reversemap)
        if len(self.labels) >= nbl:
            return
        labels = [tuple(l) if isinstance(l, (list, NP.ndarray)) else l for l in self.labels]
        groups = {}
        for key, members in blgroups.items():
            members = [tuple(m) if isinstance(m, (list, NP.ndarray)) else m for m in members]
            if key not in labels:
                rkey = tuple(reversed(key)) if isinstance(key, tuple) else key
                if rkey not in labels:
                    raise KeyError('Input label {0} not found in attribute labels'.format(key))
                key = rkey
            if key not in members:
                members = [key] + members
            groups[key] = members
        new_labels, num_list = [], []
        for label in labels:
            members = groups.get(label, [label])
            num_list.append(len(members))
            for m in members:
                if m in new_labels:
                    raise ValueError('Label {0} repeated in more than one baseline group'.format(m))
                new_labels.append(m)
        if self.skyvis_freq is not None:
            self.skyvis_freq = NP.repeat(self.skyvis_freq, num_list, axis=0)
        if self.gradient_mode is not None and self.gradient.get(self.gradient_mode) is not None:
            self.gradient[self.gradient_mode] = NP.repeat(self.gradient[self.gradient_mode], num_list, axis=1)
        self.labels = new_labels
        self.baselines = NP.repeat(self.baselines, num_list, axis=0)
        if self.projected_baselines is not None:
            self.projected_baselines = NP.repeat(self.projected_baselines, num_list, axis=0)
        self.baseline_lengths = NP.repeat(self.baseline_lengths, num_list)
        self.baseline_orientations = NP.repeat(self.baseline_orientations, num_list)
        for name in ('Tsys', 'eff_Q', 'A_eff', 'bp', 'bp_wts'):
            arr = getattr(self, name, None)
            if isinstance(arr, NP.ndarray) and arr.ndim >= 1 and arr.shape[0] > 1:
                setattr(self, name, NP.repeat(arr, num_list, axis=0))
        # the resident device array follows the expanded baseline list (slots are re-uploaded on demand)
        self._reset_device_array()
        if self._reserved >= self.n_acc and self.skyvis_freq is not None:
            self._upload_cube()
        self.generate_noise()                                                          # :6905-6906
        self.add_noise()

    # ------------------------------------------------------------------------------------------
    def delay_transform(self, pad=1.0, freq_wts=None, verbose=True):
        """Frequency -> delay transform on the GPU (rocFFT) of whichever visibility cubes exist
        (interferometry.py:8052-8137; Q20).  Sets lags, skyvis_lag (vis_lag, vis_noise_lag when their
        frequency cubes exist) and lag_kernel."""
        if not isinstance(pad, (int, float)):
            raise TypeError('pad fraction must be a scalar value.')
        if pad < 0.0:
            pad = 0.0
            if verbose:
                warnings.warn('\tPad fraction found to be negative. Resetting to 0.0 (no padding will be applied).')
        nbl, nchan = self.baselines.shape[0], self.channels.size
        if freq_wts is not None:                                                       # :8096-8107
            freq_wts = NP.asarray(freq_wts)
            if freq_wts.size == nchan and self._stacks.get('bp') is not None and len(self._stacks['bp'].layers) == self.n_acc:
                # one window for every baseline and snapshot: kept as n_acc (1, nchan) layers (dense (nbl, nchan, n_acc) on read)
                st = _LayerStack(nbl, nchan, NP.ones((1, 1)))
                st.layers = [NP.asarray(freq_wts, dtype=NP.float64).reshape(1, -1)] * self.n_acc
                self._stacks['bp_wts'] = st
                self._dense.pop('bp_wts', None)
            else:
                if freq_wts.size == nchan:
                    freq_wts = NP.repeat(NP.expand_dims(NP.repeat(freq_wts.reshape(1, -1), nbl, axis=0), axis=2), self.n_acc, axis=2)
                elif freq_wts.size == nchan * self.n_acc:
                    freq_wts = NP.repeat(NP.expand_dims(freq_wts.reshape(nchan, -1), axis=0), nbl, axis=0)
                elif freq_wts.size == nchan * nbl:
                    freq_wts = NP.repeat(NP.expand_dims(freq_wts.reshape(-1, nchan), axis=2), self.n_acc, axis=2)
                elif freq_wts.size == nchan * nbl * self.n_acc:
                    freq_wts = freq_wts.reshape(nbl, nchan, self.n_acc)
                else:
                    raise ValueError('window shape dimensions incompatible with number of channels and/or number of tiemstamps.')
                self.bp_wts = freq_wts
        if not self._cube and self._skyvis_override is None:
            raise ValueError('no visibilities to transform: call observe() first')

        # bp * bp_wts (:8116): compact per-snapshot layers when the attributes still are layer stacks, else the dense product
        wlayers = self._window_layers()
        if wlayers is not None:
            same_wts = all(l.shape == wlayers[0].shape and NP.array_equal(l, wlayers[0]) for l in wlayers[1:])
            def window(t):
                return NP.broadcast_to(wlayers[t if t < len(wlayers) else 0], (nbl, nchan))
            w_first = wlayers[0][0] if wlayers[0].shape[0] == 1 else wlayers[0]      # (nchan,): one window for every baseline
        else:
            wall = (self.bp * self.bp_wts).reshape(nbl, nchan, -1)
            same_wts = wall.shape[2] == 1 or bool(NP.all(wall == wall[:, :, [0]]))
            def window(t):
                return wall[:, :, t if t < wall.shape[2] else 0]
            w_first = wall[:, :, 0]

        def transform(cube):
            # cube (nbl, nchan, nt) times bp*bp_wts, one snapshot at a time through the device cube slot 0
            nt = cube.shape[2]
            outs = []
            for t in range(nt):
                out, lags, _ = self._ctx.delay_transform_host(cube[:, :, t], window(t if not same_wts else 0), pad)
                outs.append(out)
            return NP.stack(outs, axis=2), lags

        # number of snapshots without forcing device-resident ones onto the host (_DeviceSlot placeholders)
        nt_all = len(self._cube) if self._cube else self._skyvis_override.shape[2]
        # Are the device slots holding THIS cube?  observe() into reserved slots leaves them in step; assigning skyvis_freq (the setter,
        # adopt_observation + assignment, assemble_full_array) does not -- then the cube is uploaded first when the slots exist,
        # otherwise the transform goes snapshot by snapshot through slot 0.
        if bool(self._cube) and self._reserved >= nt_all and self.n_acc == nt_all and not getattr(self, '_device_in_step', False) \
                and not any(isinstance(sn, _DeviceSlot) for sn in self._cube):
            self._upload_cube()
        resident = self._reserved >= self.n_acc and bool(self._cube) and getattr(self, '_device_in_step', False)   # slot 0 must survive
        self._skyvis_lag, self._lag_resident = None, None
        if resident and self._reserved >= nt_all and self.n_acc == nt_all and same_wts:
            # the cube is resident on the GPU (reserve()): all snapshots are transformed where they are and the spectra stay in HBM
            # until skyvis_lag is read (the device cube is complex128 for memsave runs too: nothing is rounded on the way)
            self.lags, nout = self._ctx.delay_transform_device(nt_all, bpwts=w_first, pad=pad, want_lag=True)
            self._lag_resident = (nt_all, nout)
            self._lag_args = (nt_all, None if w_first is None else NP.array(w_first, dtype=NP.float64), pad)
            self._lag_gen = getattr(self._ctx, '_dt_generation', None)
        else:
            host_cube = NP.asarray(self.skyvis_freq, dtype=NP.complex128)
            saved0 = self._ctx.get_vis(slot=0) if resident else None       # the host-side transforms run through slot 0
            self._skyvis_lag, self.lags = transform(host_cube)
            if saved0 is not None:
                self._ctx.set_vis(saved0, slot=0)

        def through_slot0(fn):
            saved = self._ctx.get_vis(slot=0) if resident else None
            try:
                return fn()
            finally:
                if saved is not None:
                    self._ctx.set_vis(saved, slot=0)                   # put the resident snapshot back

        if self.vis_freq is not None:
            self.vis_lag = through_slot0(lambda: transform(NP.asarray(self.vis_freq, dtype=NP.complex128))[0])
        if self.vis_noise_freq is not None:
            self.vis_noise_lag = through_slot0(lambda: transform(NP.asarray(self.vis_noise_freq, dtype=NP.complex128))[0])

        def make_kernel():
            if same_wts:
                # the transform of bp * bp_wts (:8119 / :8127) is the same for every snapshot then: one FFT batch, repeated
                kern = through_slot0(lambda: transform(NP.ones((nbl, nchan, 1), dtype=NP.complex128))[0])
                return NP.repeat(kern, nt_all, axis=2)
            return through_slot0(lambda: transform(NP.ones((nbl, nchan, nt_all), dtype=NP.complex128))[0])
        self._lag_kernel, self._lag_kernel_maker = None, make_kernel      # formed on first read of lag_kernel
