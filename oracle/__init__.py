"""CPU oracle for the PRISim sky-sum hot path -- TEST INFRASTRUCTURE ONLY.

Nothing under ``oracle/`` is part of the shipped product.  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import or execute it, and there only as the checker / reported CPU baseline --
never as the thing measured or shipped.  ``prisim_amd`` never imports it.

Pinning status (see DESIGN.md, section "Oracle"):
  * PINNED against the reference's own statements executed under Python 3 on
    seeded inputs (tests/golden/make_golden.py reads the cited line ranges of
    /root/reference at generation time, in this container only; the sky-sum statements on
    two cases: baselines <= 150 m and an MWA-scale one, <= 2.5 km / ~1600 cycles of phase):
      - geometric delay (dircos path)        baseline_delay_horizon.py:236-240
      - fp64 DFT sum                          interferometry.py:6332, 6340
      - fp32 ("memsave") DFT sum              interferometry.py:6323, 6327
      - source-shape taper                    interferometry.py:6259-6283
      - baseline gradient                     interferometry.py:6338, 6343
      - apply_gradients (consumer of it)      interferometry.py:6726-6819 (the method itself, on a stand-in object)
      - Gaussian / Airy beams (zenith)        primary_beams.py:609-623, 716-728
      - dipole, 4x4 array factor, presets     primary_beams.py:975-1235, 1239-1478, 9-441
      - VLA / GMRT polynomial beams           primary_beams.py:445-513, 734-808
      - phased-array beamformer               primary_beams.py:1482-1754 (seeded jitter draws included)
      - phase-centre rotation                 interferometry.py:7871-7877
      - HDF5 layout of save()                 interferometry.py:8723-8854 (tests/golden/make_hdf5_schema.py: the statements
                                              executed against a recording h5py stand-in -> tests/golden/hdf5_schema.json)
  * PARITY UNPINNED (un-vendored, un-pinned third-party dependency
    ``astroutils``; the reference has no tests or golden vectors for them):
      - altaz<->dircos / hadec->altaz geometry (convention taken from in-tree
        docstrings, primary_beams.py:122-123, 255-258, 275-278)
      - DSP.FT1D / DSP.downsampler / DSP.spectral_axis (delay transform)
      - OPS.healpix_interp_along_axis (external HEALPix beams; restated from the published HEALPix algorithm)
      - GEOM.sphdist / dircos2altaz (off-zenith pointing, ground plane)
      - SkyModel.generate_spectrum
"""
