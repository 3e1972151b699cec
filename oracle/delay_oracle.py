"""numpy restatement of InterferometerArray.delay_transform (TEST INFRASTRUCTURE).

Follows prisim/interferometry.py:8114-8134.  The DSP.* helpers it calls live in astroutils (not in
the reference tree, no version pin): PARITY UNPINNED.  Working reading, asserted by the
known-answer test KAT-8 (tests/test_oracle_kats.py) rather than assumed:
  DSP.FT1D(x, ax=1, inverse=True, shift=True)   = fftshift(ifft(x, axis=1), axes=1)
  DSP.downsampler(x, factor, axis=1)            = linear interpolation of x at arange(0, N, factor)
                                                  (= every factor-th sample for integer factor)
  DSP.spectral_axis(N, delx=df, shift=True)     = fftshift(fftfreq(N, df))
"""
import numpy as NP


def downsampler(x, factor, axis=1):
    n = x.shape[axis]
    pos = NP.arange(0, n, factor, dtype=NP.float64)
    i0 = NP.floor(pos).astype(int)
    frac = pos - i0
    i0 = NP.minimum(i0, n - 1)
    i1 = NP.minimum(i0 + 1, n - 1)
    x0 = NP.take(x, i0, axis=axis)
    x1 = NP.take(x, i1, axis=axis)
    shape = [1] * x.ndim
    shape[axis] = -1
    return x0 + frac.reshape(shape) * (x1 - x0)


def delay_transform(skyvis_freq, bp, bp_wts, freq_resolution, pad=1.0):
    """skyvis_freq, bp, bp_wts: (nbl, nchan, nt).  Returns (skyvis_lag (nbl, nout, nt), lags (nchan,))."""
    nchan = skyvis_freq.shape[1]
    if pad < 0.0:
        pad = 0.0                                                                     # :8091-8092
    lags = NP.fft.fftshift(NP.fft.fftfreq(nchan, freq_resolution))                    # :8114
    x = skyvis_freq * bp * bp_wts
    if pad == 0.0:
        out = NP.fft.fftshift(NP.fft.ifft(x, axis=1), axes=1) * nchan * freq_resolution      # :8117
    else:
        npad = int(nchan * pad)                                                       # :8123
        xp = NP.pad(x, ((0, 0), (0, npad), (0, 0)), mode='constant')                  # :8125
        out = NP.fft.fftshift(NP.fft.ifft(xp, axis=1), axes=1) * (npad + nchan) * freq_resolution
        out = downsampler(out, 1 + pad, axis=1)                                       # :8132
    return out, lags


def delay_power(skyvis_lag, scale):
    """abs(V~)^2 * scale  (prisim/delay_spectrum.py:3992-3993 with the jacobians folded into `scale`)."""
    return NP.abs(skyvis_lag) ** 2 * scale
