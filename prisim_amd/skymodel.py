"""Minimal sky model with the duck-typed surface ``InterferometerArray.observe`` uses
(``location``, ``epoch``, ``src_shape``, ``generate_spectrum(ind, frequency, interp_method)``,
``subset``), standing in for ``astroutils.catalog.SkyModel`` which is not part of the reference tree
(PARITY UNPINNED at this boundary).  Host side, numpy only.
"""
import numpy as NP


class SkyModel(object):
    """Point sources / pixels with either a power-law spectrum (spec_type='func') or tabulated spectra
    (spec_type='spectrum').

    location   (nsrc, 2) degrees in the frame named by the consuming InterferometerArray.skycoords
    flux_ref   (nsrc,) flux density (Jy) at ref_freq     [func]
    spindex    (nsrc,) spectral index                     [func]
    spectrum   (nsrc, nf) tabulated flux densities at frequency (nf,)   [spectrum]
    src_shape  (nsrc, 3) major axis, minor axis (degrees FWHM), position angle; or None
    epoch      equinox of `location` for 'radec' skies, 'J2000' style (interferometry.py:6174); None or 'date': the coordinates are
               already those of date (apparent place), nothing is precessed (prisim_amd/frames.py)
    """

    def __init__(self, name=None, location=None, flux_ref=None, spindex=None, ref_freq=None, frequency=None,
                 spectrum=None, src_shape=None, epoch='J2000', spec_type=None):
        self.location = NP.asarray(location, dtype=NP.float64).reshape(-1, 2)
        nsrc = self.location.shape[0]
        self.name = NP.asarray(name) if name is not None else NP.arange(nsrc).astype(str)
        self.epoch = epoch
        if spec_type is None:
            spec_type = 'spectrum' if spectrum is not None else 'func'
        if spec_type not in ('func', 'spectrum'):
            raise ValueError('spec_type must be "func" or "spectrum"')
        self.spec_type = spec_type
        if spec_type == 'func':
            self.flux_ref = NP.asarray(flux_ref, dtype=NP.float64).ravel()
            self.spindex = NP.asarray(spindex, dtype=NP.float64).ravel() if spindex is not None else NP.zeros(nsrc)
            if self.spindex.size == 1:
                self.spindex = NP.full(nsrc, float(self.spindex[0]))
            self.ref_freq = float(ref_freq)
            if self.flux_ref.size != nsrc or self.spindex.size != nsrc:
                raise ValueError('flux_ref and spindex must have one element per source')
            self.frequency = None
            self.spectrum = None
        else:
            self.frequency = NP.asarray(frequency, dtype=NP.float64).ravel()
            self.spectrum = NP.asarray(spectrum, dtype=NP.float64).reshape(nsrc, self.frequency.size)
        self.src_shape = None if src_shape is None else NP.asarray(src_shape, dtype=NP.float64).reshape(nsrc, 3)

    _ARRAYS = ('location', 'flux_ref', 'spindex', 'spectrum', 'frequency', 'src_shape')

    def freeze(self):
        """Promise that the model will not be edited any more: every array becomes a private read-only copy.  InterferometerArray then
        recognises the model resident on the device by the identity of these arrays instead of a pass over their contents at every
        observe() (12 us of a 100 us snapshot on HERA-19; prisim_amd.driver.run freezes the model it builds).  Assigning a new array to
        an attribute afterwards is seen (the identity changes); returns self."""
        held = []
        for k in self._ARRAYS:
            a = getattr(self, k, None)
            if a is not None:
                a = NP.array(a, dtype=NP.float64, copy=True, order='C')       # owns its memory: no earlier view can write to it
                a.setflags(write=False)
                setattr(self, k, a)
            held.append(a)
        self._frozen = tuple(held)
        return self

    def generate_spectrum(self, ind=None, frequency=None, interp_method='linear'):
        """(len(ind), nfreq) flux densities at `frequency` (Hz).  Tabulated spectra are interpolated
        linearly, or with PCHIP when interp_method='pchip' (what observe() asks for, :6249)."""
        ind = NP.arange(self.location.shape[0]) if ind is None else NP.asarray(ind).ravel()
        frequency = NP.asarray(frequency, dtype=NP.float64).ravel()
        if self.spec_type == 'func':
            return self.flux_ref[ind, None] * (frequency[None, :] / self.ref_freq) ** self.spindex[ind, None]
        if self.frequency.size == frequency.size and NP.allclose(self.frequency, frequency, rtol=0, atol=1e-6):
            return self.spectrum[ind, :]
        if interp_method == 'pchip':
            from scipy.interpolate import PchipInterpolator
            return PchipInterpolator(self.frequency, self.spectrum[ind, :], axis=1, extrapolate=True)(frequency)
        out = NP.empty((ind.size, frequency.size))
        for i, s in enumerate(ind):
            out[i] = NP.interp(frequency, self.frequency, self.spectrum[s])
        return out

    def subset(self, indices, axis='position'):
        indices = NP.asarray(indices).ravel()
        if axis != 'position':
            raise ValueError('only axis="position" is supported')
        kw = dict(name=self.name[indices], location=self.location[indices], epoch=self.epoch, spec_type=self.spec_type,
                  src_shape=None if self.src_shape is None else self.src_shape[indices])
        if self.spec_type == 'func':
            kw.update(flux_ref=self.flux_ref[indices], spindex=self.spindex[indices], ref_freq=self.ref_freq)
        else:
            kw.update(frequency=self.frequency, spectrum=self.spectrum[indices])
        return SkyModel(**kw)
