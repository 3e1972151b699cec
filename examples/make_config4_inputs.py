#!/usr/bin/env python3
"""Writes the input files examples/config4.yaml names (BASELINE config 4; the reference's own data directory, prisim/data, is not in its tree):

  config4_mwa128_layout.txt   128 tile positions (Label East North Up), prisim_amd.workloads.mwa128_layout(seed=4): a 2-D Gaussian of
                              sigma 400 m -- committed, regenerated bit-identically by this script
  config4_beam.hdf5           external power beam, HEALPix nside 32 RING in the local (zenith angle, azimuth) frame at 21 frequencies
                              165-205 MHz, in the layout the reference's FEKO converter writes (scripts/FEKO_beam_to_healpix.py:161-198):
                              gain_info/X  (nfreq, npix),  spectral_info/freqs (Hz),  header/gainunit

usage: python examples/make_config4_inputs.py [outdir]"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import numpy as NP                                   # noqa: E402
from prisim_amd import workloads as W, hdf5io       # noqa: E402


def main(outdir=HERE):
    pos = W.mwa128_layout()
    with open(os.path.join(outdir, 'config4_mwa128_layout.txt'), 'w') as f:
        f.write('# synthetic MWA-128T: 128 tiles from a 2-D Gaussian (sigma 400 m), prisim_amd.workloads.mwa128_layout(seed=4)\n')
        f.write('Label East North Up\n')
        for i, p in enumerate(pos):
            f.write('T%03d %r %r %r\n' % (i, float(p[0]), float(p[1]), float(p[2])))
    freqs = NP.linspace(165e6, 205e6, 21)
    beam = W.synthetic_healpix_beam(32, freqs)       # (npix, nfreq)
    path = os.path.join(outdir, 'config4_beam.hdf5')
    with hdf5io.File(path, 'w') as f:
        f.write('gain_info/X', NP.ascontiguousarray(beam.T))
        f.write('spectral_info/freqs', freqs)
        f.write('header/gainunit', '')
    return path


if __name__ == '__main__':
    print(main(sys.argv[1] if len(sys.argv) > 1 else HERE))
