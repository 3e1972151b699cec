"""Delay spectra and delay power spectra of simulated visibilities on the GPU.

Mirrors the part of prisim/delay_spectrum.py that follows the sky-sum in a run: ``DelaySpectrum(ia).delay_transform(pad, freq_wts,
downsample, action)`` (:1224-1342) and ``DelayPowerSpectrum(ds).compute_power_spectrum()`` (:3605-3678, 3982-3995) with the same names,
keywords, attributes and exception types.  The transform itself (window, zero-pad, inverse FFT, shift, decimate) and
``abs(.)**2 * jacobian1 * jacobian2 * Jy2K**2`` run in libprisim_hip.so (prisim_hip_delay_transform*, one HBM-bound kernel for
power-of-two channel counts); the host computes the scalars: redshift, comoving distances, the beam volume, the Jy -> K factor.

Not here (out of scope): delay CLEAN, sub-band transforms, closure-phase spectra, FITS persistence.

Cosmology.  The reference takes ``astropy.cosmology.Planck15.clone(H0=100)`` (:34-35); astropy is not in this image, so ``cosmo100`` here
is this module's own flat LambdaCDM with Planck15's Om0 = 0.3075, Tcmb0 = 2.7255 K, Neff = 3.046 (photons + massless neutrinos in the
radiation term; Planck15's single 0.06 eV neutrino is not modelled: E(z) differs by ~1e-3 at z ~ 8) -- PARITY UNPINNED against astropy.
Any object with astropy's interface (``H0.value``, ``efunc(z)``, ``comoving_distance(z).to('Mpc').value``,
``comoving_transverse_distance(z)``) is accepted in its place.
"""
import warnings

import numpy as NP
import scipy.constants as FCNST

from . import _abi
from . import geometry as GEOM
from . import primary_beams as PB

REST_FREQ_HI = 1420405751.77      # Hz (astroutils.constants.rest_freq_HI, used at :3642, 3707)
JY = 1.0e-26                      # W m^-2 Hz^-1 (astroutils.constants.Jy, :3663)


class _Quantity(object):
    """The two accessors of an astropy Quantity the reference uses: ``.value`` and ``.to('Mpc').value``."""

    def __init__(self, value):
        self.value = value

    def to(self, unit):
        if unit != 'Mpc':
            raise ValueError('only Mpc is supported')
        return self


class FlatLambdaCDM(object):
    """Flat LambdaCDM with radiation: E(z)^2 = Om0 (1+z)^3 + Or0 (1+z)^4 + (1 - Om0 - Or0); comoving distances by quadrature."""

    def __init__(self, H0=100.0, Om0=0.3075, Tcmb0=2.7255, Neff=3.046, name=None):
        self.name = name
        self.H0 = _Quantity(float(H0))
        self.Om0 = float(Om0)
        self.Tcmb0, self.Neff = float(Tcmb0), float(Neff)
        h100 = float(H0) * 1e3 / (1e6 * FCNST.parsec)                              # s^-1
        rho_crit = 3.0 * h100 ** 2 / (8.0 * NP.pi * FCNST.G)                       # kg m^-3
        rho_gamma = 4.0 * FCNST.Stefan_Boltzmann * self.Tcmb0 ** 4 / FCNST.c ** 3  # kg m^-3
        self.Ogamma0 = rho_gamma / rho_crit
        self.Or0 = self.Ogamma0 * (1.0 + 0.22710731766 * self.Neff)                # 7/8 (4/11)^(4/3) per massless species
        self.Ode0 = 1.0 - self.Om0 - self.Or0

    def efunc(self, z):
        zp1 = 1.0 + NP.asarray(z, dtype=NP.float64)
        return NP.sqrt(self.Om0 * zp1 ** 3 + self.Or0 * zp1 ** 4 + self.Ode0)

    def comoving_distance(self, z):
        from scipy.integrate import quad
        dh = FCNST.c / 1e3 / self.H0.value                                         # Mpc
        zs = NP.atleast_1d(NP.asarray(z, dtype=NP.float64))
        out = NP.array([quad(lambda x: 1.0 / float(self.efunc(x)), 0.0, zi, epsabs=0.0, epsrel=1e-12)[0] for zi in zs]) * dh
        return _Quantity(out.reshape(NP.shape(z)) if NP.ndim(z) else float(out[0]))

    def comoving_transverse_distance(self, z):
        return self.comoving_distance(z)                                           # flat


cosmo100 = FlatLambdaCDM(H0=100.0, Om0=0.3075, name='flat LambdaCDM, Planck 2015 Om0, h = 1.0 (:34-35)')


def _is_cosmology(c):
    return hasattr(c, 'H0') and hasattr(c, 'efunc') and hasattr(c, 'comoving_distance') and hasattr(c, 'comoving_transverse_distance')


def dkprll_deta(redshift, cosmo=cosmo100):
    """Jacobian delay -> k_parallel (h/Mpc per second), :357-391."""
    if not isinstance(redshift, (int, float, list, NP.ndarray)):
        raise TypeError('redshift must be a scalar, list or numpy array')
    redshift = NP.asarray(redshift)
    if NP.any(redshift < 0.0):
        raise ValueError('redshift(s) must be non-negative')
    if not _is_cosmology(cosmo):
        raise TypeError('Input cosmology must be a cosmology class defined in Astropy')
    return 2 * NP.pi * cosmo.H0.value * REST_FREQ_HI * cosmo.efunc(redshift) / FCNST.c / (1 + redshift) ** 2 * 1e3      # :389


def beam3Dvol(beam, freqs, freq_wts=None, hemisphere=True):
    """Integral of the squared power pattern over solid angle and frequency, in Sr Hz (:395-489).  beam (npix, nchan | 1) on a
    HEALPix RING grid in the local frame (theta = zenith angle), peak-normalised."""
    if not isinstance(beam, NP.ndarray):
        raise TypeError('Input beam must be a numpy array')
    if not isinstance(freqs, (list, NP.ndarray)):
        raise TypeError('Input freqs must be a list or numpy array')
    freqs = NP.asarray(freqs).astype(NP.float64).reshape(-1)
    if freqs.size < 2:
        raise ValueError('Input freqs does not have enough elements to determine frequency resolution')
    if beam.ndim > 2:
        raise ValueError('Invalid dimensions for beam')
    elif beam.ndim == 2:
        if beam.shape[1] != 1 and beam.shape[1] != freqs.size:
            raise ValueError('Dimensions of beam do not match the number of frequency channels')
    elif beam.ndim == 1:
        beam = beam.reshape(-1, 1)
    else:
        raise ValueError('Invalid dimensions for beam')
    if freq_wts is not None:
        if not isinstance(freq_wts, NP.ndarray):
            raise TypeError('Input freq_wts must be a numpy array')
        if freq_wts.ndim > 2:
            raise ValueError('Input freq_wts must be of shape nwin x nchan')
        freq_wts = NP.asarray(freq_wts).astype(NP.float64).reshape(-1, freqs.size)
    else:
        freq_wts = NP.ones(freqs.size, dtype=NP.float64).reshape(1, -1)
    eps = 1e-10
    if beam.max() > 1.0 + eps:
        raise ValueError('Input beam maximum exceeds unity. Input beam should be normalized to peak of unity')
    npix = beam.shape[0]
    nside = int(round(NP.sqrt(npix / 12.0)))
    if 12 * nside * nside != npix:
        raise ValueError('beam does not have a HEALPix number of pixels')
    domega = 4.0 * NP.pi / npix
    df = freqs[1] - freqs[0]
    bw = df * freqs.size
    theta, _ = GEOM.healpix_pix2ang_ring(nside)
    ind = NP.where(theta <= NP.pi / 2)[0] if hemisphere else NP.arange(npix)
    b2 = beam[ind, :] ** 2                                                               # (npix', nchan | 1)
    # sum over pixels and channels of (beam * wts)^2 for every window (:484), without the (npix, nwin, nchan) temporary
    if b2.shape[1] == 1:
        omega_bw = domega * df * NP.nansum(b2) * NP.sum(freq_wts ** 2, axis=1)
    else:
        omega_bw = domega * df * (freq_wts ** 2).dot(NP.nansum(b2, axis=0))
    if NP.any(omega_bw > 4 * NP.pi * bw):
        raise ValueError('3D volume estimated from beam exceeds the upper limit. Check normalization of the input beam')
    return omega_bw


def healpix_power_pattern(channels, telescope, nside=32, extbeam=None, device=0):
    """(npix, nchan) power pattern at the pixel centres of a HEALPix RING grid in the local frame (theta = zenith angle, phi = azimuth;
    zero below the horizon), evaluated on the GPU: the analytic beam of ``telescope`` (:3956-3963), or an external beam
    ``extbeam = (table (npix_beam, nfreq), spectral interpolation matrix (nchan, nfreq))`` log-interpolated and peak-normalised per channel
    as in a run (:3934-3953; the grid is then no finer than the table's, :3931-3932)."""
    f = NP.asarray(channels, dtype=NP.float64)
    if extbeam is not None:
        beam_nside = int(round(NP.sqrt(extbeam[0].shape[0] / 12.0)))
        if beam_nside < nside:
            nside = beam_nside
    theta, phi = GEOM.healpix_pix2ang_ring(nside)
    up = theta <= NP.pi / 2
    altaz_up = NP.hstack(((90.0 - NP.degrees(theta[up])).reshape(-1, 1), NP.degrees(phi[up]).reshape(-1, 1)))
    beam = NP.zeros((theta.size, f.size))
    if extbeam is not None:
        dc = GEOM.altaz2dircos(altaz_up, 'degrees')
        n = dc.shape[0]
        with _abi.Context(device) as ctx:
            ctx.set_array(NP.zeros((1, 3)), f, nt_max=1)
            ctx.set_external_beam(*extbeam)
            ctx.set_sky_external_analytic(dc, NP.ones(n), NP.zeros(n), 1.0, NP.array([0.0, 0.0, 1.0]))
            beam[up] = ctx.get_pbflux()
    else:
        beam[up] = PB.primary_beam_generator(altaz_up, f, telescope, freq_scale='Hz', skyunits='altaz', east2ax1=0.0, pointing_info=None,
                                             pointing_center=None, device=device)
    return beam


def power_constants(channels, telescope, freq_wts=None, cosmo=cosmo100, nside=32, extbeam=None, device=0):
    """The scalars of DelayPowerSpectrum.__init__ (:3640-3663) for callers that hold a device cube but no InterferometerArray
    (bench.py, tools/): f0, wl0, z, bw, drz_los, rz_los, omega_bw, jacobian1, jacobian2, Jy2K and factor = jacobian1 jacobian2 Jy2K^2
    (a float when freq_wts is one window)."""
    f = NP.asarray(channels, dtype=NP.float64)
    df = f[1] - f[0]
    f0 = f[int(f.size / 2)]
    wl0 = FCNST.c / f0
    z = REST_FREQ_HI / f0 - 1
    bw = df * f.size
    drz_los = (FCNST.c / 1e3) * bw * (1 + z) ** 2 / REST_FREQ_HI / cosmo.H0.value / float(cosmo.efunc(z))
    rz_los = cosmo.comoving_distance(z).to('Mpc').value
    omega_bw = beam3Dvol(healpix_power_pattern(f, telescope, nside=nside, extbeam=extbeam, device=device), f,
                         freq_wts=None if freq_wts is None else NP.asarray(freq_wts, dtype=NP.float64))
    jacobian1 = 1 / omega_bw
    jacobian2 = rz_los ** 2 * drz_los / bw
    Jy2K = wl0 ** 2 * JY / (2 * FCNST.k)
    factor = jacobian1 * jacobian2 * Jy2K ** 2
    return {'f0': f0, 'wl0': wl0, 'z': z, 'bw': bw, 'drz_los': drz_los, 'rz_los': rz_los, 'omega_bw': omega_bw, 'jacobian1': jacobian1,
            'jacobian2': jacobian2, 'Jy2K': Jy2K, 'factor': float(factor[0]) if factor.size == 1 else factor,
            'cosmology': getattr(cosmo, 'name', None) or type(cosmo).__name__}


class DelaySpectrum(object):
    """Delay spectra of an InterferometerArray's visibilities (prisim/delay_spectrum.py:493-1342, the argument path of __init__ and
    delay_transform()).  Attributes as in the reference: ia, f, df, n_acc, bp, bp_wts, pad, lags, lag_kernel, skyvis_lag, vis_lag,
    vis_noise_lag, horizon_delay_limits; the CLEAN / sub-band attributes exist and stay None / empty."""

    def __init__(self, interferometer_array=None, init_file=None):
        if init_file is not None:
            raise NotImplementedError('DelaySpectrum(init_file=...): FITS persistence is out of scope (SURVEY.md 2.1 row 17)')
        from .interferometry import InterferometerArray
        if not isinstance(interferometer_array, InterferometerArray):
            raise TypeError('Input interferometer_array must be an instance of class InterferometerArray')
        self.ia = interferometer_array
        self.f = interferometer_array.channels
        self.df = interferometer_array.freq_resolution
        self.n_acc = interferometer_array.n_acc
        self.horizon_delay_limits = self.get_horizon_delay_limits()
        self.pad = 0.0
        self.lags = NP.fft.fftshift(NP.fft.fftfreq(self.f.size, self.df))               # DSP.spectral_axis(N, delx=df, shift=True), :1191
        self._bp_wts_override = None
        self._lag_kernel, self._lag_kernel_maker = None, None
        self._skyvis_lag, self._lag_resident = None, None
        self.vis_lag = None
        self.vis_noise_lag = None
        self.clean_window_buffer = 1.0
        for name in ('cc_lags', 'cc_freq', 'cc_lag_kernel', 'cc_skyvis_lag', 'cc_skyvis_res_lag', 'cc_vis_lag', 'cc_vis_res_lag',
                     'cc_skyvis_net_lag', 'cc_vis_net_lag', 'cc_skyvis_freq', 'cc_skyvis_res_freq', 'cc_vis_freq', 'cc_vis_res_freq',
                     'cc_skyvis_net_freq', 'cc_vis_net_freq'):
            setattr(self, name, None)
        self.subband_delay_spectra = {}
        self.subband_delay_spectra_resampled = {}

    # bp / bp_wts: the array's own attributes (dense (nbl, nchan, n_acc) on read, as in the reference) unless delay_transform(action='store')
    # replaced the weights
    @property
    def bp(self):
        return self.ia.bp

    @property
    def bp_wts(self):
        return self._bp_wts_override if self._bp_wts_override is not None else self.ia.bp_wts

    @bp_wts.setter
    def bp_wts(self, value):
        self._bp_wts_override = value

    def _refresh_resident(self):
        """The stored delay spectra live in the context's single resident buffer and are fetched when read.  Any later transform on that
        context (another window or pad without action='store', InterferometerArray.delay_transform, a power-spectrum fetch) overwrites the
        buffer: the transform is then simply run again (milliseconds) before the read -- a stored result never changes, as in the
        reference, whose store is a host copy."""
        ctx = self.ia._ctx
        if getattr(ctx, '_dt_generation', None) != getattr(self, '_lag_gen', None):
            nt, w0, pad = self._resident_args
            ctx.delay_transform_device(nt, bpwts=w0, pad=pad, want_lag=True)
            self._lag_gen = getattr(ctx, '_dt_generation', None)

    @property
    def skyvis_lag(self):
        if self._lag_resident is not None and self._skyvis_lag is None:
            nt, _ = self._lag_resident
            self._refresh_resident()
            self._skyvis_lag = NP.transpose(self.ia._ctx.get_lags(0, nt), (1, 2, 0))
        return self._skyvis_lag

    @skyvis_lag.setter
    def skyvis_lag(self, value):
        self._skyvis_lag, self._lag_resident = value, None

    @property
    def lag_kernel(self):
        if self._lag_kernel is None and self._lag_kernel_maker is not None:
            self._lag_kernel = self._lag_kernel_maker()
            self._lag_kernel_maker = None
        return self._lag_kernel

    @lag_kernel.setter
    def lag_kernel(self, value):
        self._lag_kernel, self._lag_kernel_maker = value, None

    def get_horizon_delay_limits(self, phase_center=None, phase_center_coords=None):
        """(n_phase_centres, nbl, 2): min / max delay of the horizon for every baseline, shifted by the phase centre (:2976-3030 and
        baseline_delay_horizon.py:100-129)."""
        if phase_center is None:
            phase_center = self.ia.phase_center
            phase_center_coords = self.ia.phase_center_coords
        if phase_center_coords not in ['hadec', 'altaz', 'dircos']:
            raise ValueError('Phase center coordinates must be "altaz", "hadec" or "dircos"')
        pc = NP.asarray(phase_center, dtype=NP.float64)
        pc = pc.reshape(-1, 3 if phase_center_coords == 'dircos' else 2)
        if phase_center_coords == 'hadec':
            pc_dircos = GEOM.altaz2dircos(GEOM.hadec2altaz(pc, self.ia.latitude, units='degrees'), units='degrees')
        elif phase_center_coords == 'altaz':
            pc_dircos = GEOM.altaz2dircos(pc, units='degrees')
        else:
            pc_dircos = pc
        bl = NP.asarray(self.ia.baselines, dtype=NP.float64)
        dmax = NP.sqrt(NP.sum(bl ** 2, axis=1)).reshape(1, -1) / FCNST.c                # baseline_delay_horizon.py:94
        shift = pc_dircos.dot(bl.T) / FCNST.c                                            # :95
        return NP.dstack((-dmax - shift, dmax - shift))                                  # :127-128

    def set_horizon_delay_limits(self):
        self.horizon_delay_limits = self.get_horizon_delay_limits()

    # ------------------------------------------------------------------------------------------
    def _window_source(self, freq_wts):
        """Per-snapshot windows bp * freq_wts in the most compact form available: (layers, same) with layers[t] of shape
        (1 | nbl, nchan) and same = every snapshot carries the same window; plus the freq_wts to report (a zero-copy broadcast
        view when one window serves every baseline and snapshot)."""
        ia = self.ia
        nbl, nchan, nt = ia.baselines.shape[0], self.f.size, self.n_acc
        stacks = getattr(ia, '_stacks', {})

        def layers_of(name):
            st = stacks.get(name)
            if st is not None and len(st.layers) == nt and nt > 0:
                return list(st.layers)
            dense = NP.asarray(getattr(ia, name))
            if dense.ndim == 2:
                return [dense] * max(nt, 1)
            return [dense[:, :, t] for t in range(dense.shape[2])]

        bp_layers = layers_of('bp')
        if freq_wts is not None:
            if freq_wts.size == nchan:                                                   # :1275-1276
                w = NP.asarray(freq_wts, dtype=NP.float64).reshape(1, -1)
                w_layers = [w] * max(nt, 1)
                report = NP.broadcast_to(w.reshape(1, -1, 1), (nbl, nchan, nt))
            elif freq_wts.size == nchan * nt:                                            # :1277-1278
                w2 = NP.asarray(freq_wts, dtype=NP.float64).reshape(nchan, -1)
                w_layers = [w2[:, t].reshape(1, -1) for t in range(nt)]
                report = NP.broadcast_to(w2[NP.newaxis, :, :], (nbl, nchan, nt))
            elif freq_wts.size == nchan * nbl:                                           # :1279-1280
                w2 = NP.asarray(freq_wts, dtype=NP.float64).reshape(-1, nchan)
                w_layers = [w2] * max(nt, 1)
                report = NP.broadcast_to(w2[:, :, NP.newaxis], (nbl, nchan, nt))
            elif freq_wts.size == nchan * nbl * nt:                                      # :1281-1282
                report = NP.asarray(freq_wts, dtype=NP.float64).reshape(nbl, nchan, nt)
                w_layers = [report[:, :, t] for t in range(nt)]
            else:
                raise ValueError('window shape dimensions incompatible with number of channels and/or number of tiemstamps.')
        else:
            if self._bp_wts_override is not None:
                report = NP.asarray(self._bp_wts_override)
                w_layers = [report[:, :, t] for t in range(report.shape[2])] if report.ndim == 3 else [report] * max(nt, 1)
            else:
                w_layers = layers_of('bp_wts')
                report = None                                                            # the array's own (dense on read)
        n = min(len(bp_layers), len(w_layers))
        layers = [NP.asarray(bp_layers[t]) * NP.asarray(w_layers[t]) for t in range(n)]
        same = all(l.shape == layers[0].shape and NP.array_equal(l, layers[0]) for l in layers[1:])
        return layers, same, report

    def _transform_full(self, cube_t, window_t, pad):
        """One snapshot without the final decimation: the zero-padded product is uploaded as the single snapshot of a temporary context
        whose channel grid is the padded one, and transformed there with pad = 0 -- same FFT length, same (npad + N) df scale (:1316-1321)."""
        nbl, nchan = cube_t.shape
        npad = int(nchan * pad)
        x = NP.zeros((nbl, nchan + npad), dtype=NP.complex128)
        x[:, :nchan] = cube_t * window_t
        grid = self.f[0] + self.df * NP.arange(nchan + npad)
        with _abi.Context(getattr(self.ia._ctx, 'device', 0)) as tmp:
            tmp.set_array(NP.asarray(self.ia.baselines, dtype=NP.float64), grid, nt_max=1)
            out, _, _ = tmp.delay_transform_host(x, None, 0.0)
        return out

    def delay_transform(self, pad=1.0, freq_wts=None, downsample=True, action=None, verbose=True):
        """IFFT of visibilities * bandpass * window along frequency on the GPU (:1224-1342): skyvis_lag, vis_lag, vis_noise_lag (for
        the cubes that exist), lag_kernel, lags, freq_wts, pad.  With the visibility cube resident in HBM (InterferometerArray.reserve)
        and one window for every snapshot, the spectra stay on the device until ``skyvis_lag`` is read."""
        if verbose:
            print('Preparing to compute delay transform...\n\tChecking input parameters for compatibility...')
        if not isinstance(pad, (int, float)):
            raise TypeError('pad fraction must be a scalar value.')
        if pad < 0.0:
            pad = 0.0
            if verbose:
                print('\tPad fraction found to be negative. Resetting to 0.0 (no padding will be applied).')
        if freq_wts is not None:
            freq_wts = NP.asarray(freq_wts)
        layers, same, report = self._window_source(freq_wts)
        if verbose:
            print('\tFrequency window weights assigned.')
        if not isinstance(downsample, bool):
            raise TypeError('Input downsample must be of boolean type')
        ia = self.ia
        nbl, nchan, nt = ia.baselines.shape[0], self.f.size, self.n_acc
        if nt == 0:
            raise ValueError('no visibilities to transform: call observe() first')
        ctx = ia._ctx
        result = {'freq_wts': report if report is not None else self.bp_wts, 'pad': pad}
        nfft = int(nchan * (1 + pad))
        result['lags'] = NP.fft.fftshift(NP.fft.fftfreq(nfft, self.df))                 # :1303
        decimate = downsample or pad == 0.0

        def window(t):
            return NP.broadcast_to(layers[t if (not same and t < len(layers)) else 0], (nbl, nchan))

        def transform(cube):
            outs = []
            for t in range(cube.shape[2]):
                if decimate:
                    out, _, _ = ctx.delay_transform_host(cube[:, :, t], window(t), pad)
                else:
                    out = self._transform_full(cube[:, :, t], window(t), pad)
                outs.append(out)
            return NP.stack(outs, axis=2)

        # the sky visibilities: on the device where they already are, when they are
        if hasattr(ia, '_cube') and bool(ia._cube) and ia._reserved >= nt and not getattr(ia, '_device_in_step', False) \
                and not any(type(sn).__name__ == '_DeviceSlot' for sn in ia._cube):
            ia._upload_cube()
        resident = bool(getattr(ia, '_cube', None)) and ia._reserved >= nt and getattr(ia, '_device_in_step', False)
        lag_resident, skyvis_lag, resident_args = None, None, None
        if resident and same and decimate:
            w0 = layers[0][0] if layers[0].shape[0] == 1 else layers[0]
            _, nout = ctx.delay_transform_device(nt, bpwts=w0, pad=pad, want_lag=True)
            lag_resident = (nt, nout)
            resident_args = (nt, NP.array(w0, dtype=NP.float64), pad)
        else:
            saved0 = ctx.get_vis(slot=0) if resident else None                          # the host-side transforms run through slot 0
            skyvis_lag = transform(NP.asarray(ia.skyvis_freq, dtype=NP.complex128))
            if saved0 is not None:
                ctx.set_vis(saved0, slot=0)

        def through_slot0(fn):
            saved = ctx.get_vis(slot=0) if resident else None
            try:
                return fn()
            finally:
                if saved is not None:
                    ctx.set_vis(saved, slot=0)

        # (the reference multiplies vis_freq / vis_noise_freq unconditionally and fails on a noiseless object, SURVEY Q20: here the
        # cubes that exist are transformed)
        vis_lag = vis_noise_lag = None
        if ia.vis_freq is not None:
            vis_lag = through_slot0(lambda: transform(NP.asarray(ia.vis_freq, dtype=NP.complex128)))
        if ia.vis_noise_freq is not None:
            vis_noise_lag = through_slot0(lambda: transform(NP.asarray(ia.vis_noise_freq, dtype=NP.complex128)))

        def make_kernel():
            if same:
                kern = through_slot0(lambda: transform(NP.ones((nbl, nchan, 1), dtype=NP.complex128)))
                return NP.repeat(kern, nt, axis=2)
            return through_slot0(lambda: transform(NP.ones((nbl, nchan, nt), dtype=NP.complex128)))

        if decimate and pad > 0.0:
            result['lags'] = result['lags'][NP.arange(0, nfft, 1 + pad).astype(int)] if float(1 + pad).is_integer() else \
                NP.interp(NP.arange(0, nfft, 1 + pad), NP.arange(nfft), result['lags'])    # DSP.downsampler(lags, 1 + pad), :1329
            result['lags'] = result['lags'].flatten()
            if verbose:
                print('\tDelay transform products downsampled by factor of {0:.1f}'.format(1 + pad))
                print('delay_transform() completed successfully.')

        if action == 'store':
            self.pad = pad
            self.lags = result['lags']
            if report is not None:
                self._bp_wts_override = report
            self._skyvis_lag, self._lag_resident = skyvis_lag, lag_resident
            if lag_resident is not None:               # (only a STORED result keeps a claim on the resident buffer)
                self._resident_args, self._lag_gen = resident_args, getattr(ctx, '_dt_generation', None)
            self.vis_lag = vis_lag
            self.vis_noise_lag = vis_noise_lag
            self._lag_kernel, self._lag_kernel_maker = None, make_kernel
        if lag_resident is not None:
            result['skyvis_lag'] = NP.transpose(ctx.get_lags(0, nt), (1, 2, 0)) if action != 'store' else _Deferred(lambda: self.skyvis_lag)
        else:
            result['skyvis_lag'] = skyvis_lag
        result['vis_lag'] = vis_lag
        result['vis_noise_lag'] = vis_noise_lag
        result['lag_kernel'] = _Deferred(make_kernel) if action == 'store' else make_kernel()
        return _LazyDict(result)


class _Deferred(object):
    def __init__(self, fn):
        self.fn = fn


class _LazyDict(dict):
    """dict whose _Deferred values are materialised on first access (a 120 GB spectrum cube is only pulled over PCIe when read)."""

    def __getitem__(self, key):
        v = dict.__getitem__(self, key)
        if isinstance(v, _Deferred):
            v = v.fn()
            dict.__setitem__(self, key, v)
        return v

    def get(self, key, default=None):
        return self[key] if key in self else default

    def items(self):
        return [(k, self[k]) for k in self.keys()]

    def values(self):
        return [self[k] for k in self.keys()]


class DelayPowerSpectrum(object):
    """Delay power spectra in K^2 (Mpc/h)^3 from delay spectra in Jy Hz (prisim/delay_spectrum.py:3355-3678, 3982-3995)."""

    def __init__(self, dspec, cosmo=cosmo100):
        if not isinstance(dspec, DelaySpectrum):
            raise TypeError('Input dspec must be an instance of class DelaySpectrum')
        if not _is_cosmology(cosmo):
            raise TypeError('Input cosmology must be a cosmology class defined in Astropy')
        self.cosmo = cosmo
        self.ds = dspec
        self.f = self.ds.f
        self.lags = self.ds.lags
        self.cc_lags = self.ds.cc_lags
        self.bl = self.ds.ia.baselines
        self.bl_length = self.ds.ia.baseline_lengths
        self.df = self.ds.df
        self.f0 = self.f[int(self.f.size / 2)]                                          # :3640
        self.wl0 = FCNST.c / self.f0
        self.z = REST_FREQ_HI / self.f0 - 1                                             # :3642
        self.bw = self.df * self.f.size
        self.kprll = self.k_parallel(self.lags, redshift=self.z, action='return')       # h/Mpc
        self.kperp = self.k_perp(self.bl_length, redshift=self.z, action='return')      # h/Mpc
        self.horizon_kprll_limits = self.k_parallel(self.ds.horizon_delay_limits, redshift=self.z, action='return')
        self.drz_los = self.comoving_los_depth(self.bw, self.z, action='return')        # Mpc/h
        self.rz_transverse = self.comoving_transverse_distance(self.z, action='return')
        self.rz_los = self.comoving_los_distance(self.z, action='return')
        omega_bw = self.beam3Dvol(freq_wts=self._first_window())                        # :3655 freq_wts = ds.bp_wts[0,:,0]
        self.jacobian1 = 1 / omega_bw                                                   # :3656
        self.jacobian2 = self.rz_los ** 2 * self.drz_los / self.bw                      # :3658
        self.Jy2K = self.wl0 ** 2 * JY / (2 * FCNST.k)                                  # :3659
        self.K2Jy = 1 / self.Jy2K
        self.dps = {}
        for key in ('skyvis', 'vis', 'noise', 'cc_skyvis', 'cc_vis', 'cc_skyvis_res', 'cc_vis_res', 'cc_skyvis_net', 'cc_vis_net'):
            self.dps[key] = None
        self.subband_delay_power_spectra = {}
        self.subband_delay_power_spectra_resampled = {}

    def _first_window(self):
        """ds.bp_wts[0, :, 0] without forming the dense (nbl, nchan, n_acc) array."""
        ds = self.ds
        if ds._bp_wts_override is not None:
            w = NP.asarray(ds._bp_wts_override)
            return NP.array(w[0, :, 0] if w.ndim == 3 else w[0, :], dtype=NP.float64)
        st = getattr(ds.ia, '_stacks', {}).get('bp_wts')
        if st is not None and st.layers:
            return NP.array(NP.broadcast_to(st.layers[0], (st.layers[0].shape[0], self.f.size))[0], dtype=NP.float64)
        w = NP.asarray(ds.ia.bp_wts)
        return NP.array(w[0, :, 0] if w.ndim == 3 else w[0, :], dtype=NP.float64)

    def comoving_los_depth(self, bw, redshift, action=None):
        drz_los = (FCNST.c / 1e3) * bw * (1 + redshift) ** 2 / REST_FREQ_HI / self.cosmo.H0.value / self.cosmo.efunc(redshift)   # :3707
        if action is None:
            self.z = redshift
            self.drz_los = drz_los
            return
        return drz_los

    def comoving_transverse_distance(self, redshift, action=None):
        rz_transverse = self.cosmo.comoving_transverse_distance(redshift).to('Mpc').value        # :3741
        if action is None:
            self.z = redshift
            self.rz_transverse = rz_transverse
            return
        return rz_transverse

    def comoving_los_distance(self, redshift, action=None):
        rz_los = self.cosmo.comoving_distance(redshift).to('Mpc').value                          # :3775
        if action is None:
            self.z = redshift
            self.rz_los = rz_los
            return
        return rz_los

    def k_parallel(self, lags, redshift, action=None):
        kprll = dkprll_deta(redshift, cosmo=self.cosmo) * lags                                   # :3813-3814
        if action is None:
            self.z = redshift
            self.kprll = kprll
            return
        return kprll

    def k_perp(self, baseline_length, redshift, action=None):
        kperp = 2 * NP.pi * (baseline_length / self.wl0) / self.comoving_transverse_distance(redshift, action='return')   # :3853
        if action is None:
            self.z = redshift
            self.kperp = kperp
            return
        return kperp

    def beam3Dvol(self, freq_wts=None, nside=32):
        """Omega x bandwidth of the array's power pattern (:3864-3978): the pattern is evaluated on the GPU at the pixel centres of a
        HEALPix grid in the local frame -- the analytic beam of ia.telescope, or the external beam the array was given
        (InterferometerArray.set_external_beam: log-interpolated in frequency, peak-normalised per channel) -- and squared and summed
        over the upper hemisphere and the band."""
        ia = self.ds.ia
        beam = healpix_power_pattern(self.f, ia.telescope, nside=nside, extbeam=getattr(ia, '_extbeam', None),
                                     device=getattr(ia._ctx, 'device', 0))
        return beam3Dvol(beam, self.f, freq_wts=freq_wts, hemisphere=True)

    def power_scale(self):
        """jacobian1 * jacobian2 * Jy2K**2 (:3992) as a scalar (one spectral window)."""
        return float(NP.ravel(self.jacobian1 * self.jacobian2 * self.Jy2K ** 2)[0])

    def compute_power_spectrum(self):
        """dps['skyvis' | 'vis' | 'noise'] = abs(lag spectrum)**2 * jacobian1 * jacobian2 * Jy2K**2 (:3982-3995).  When the delay
        spectra are resident on the device the product is formed there (prisim_hip_delay_transform_device with power_scale = the factor)
        and fetched when dps['skyvis'] is read."""
        ds = self.ds
        factor = self.jacobian1 * self.jacobian2 * self.Jy2K ** 2
        dps = _LazyDict()
        if ds._lag_resident is not None and ds._skyvis_lag is None:
            nt, w0, pad = ds._resident_args
            ctx = ds.ia._ctx
            k = self.power_scale()

            def fetch():
                ctx.delay_transform_device(nt, bpwts=w0, pad=pad, want_lag=True, want_power=True, power_scale=k)
                ds._lag_gen = getattr(ctx, '_dt_generation', None)       # (the same spectra again, with their power beside them)
                return NP.transpose(ctx.get_delay_power(0, nt), (1, 2, 0))
            dps['skyvis'] = _Deferred(fetch)
        elif ds.skyvis_lag is not None:
            dps['skyvis'] = NP.abs(ds.skyvis_lag) ** 2 * factor
        if ds.vis_lag is not None:
            dps['vis'] = NP.abs(ds.vis_lag) ** 2 * factor
        if ds.vis_noise_lag is not None:
            dps['noise'] = NP.abs(ds.vis_noise_lag) ** 2 * factor
        self.dps = dps
        if ds.subband_delay_spectra or ds.subband_delay_spectra_resampled:
            warnings.warn('sub-band delay power spectra are not on the accelerated path')
