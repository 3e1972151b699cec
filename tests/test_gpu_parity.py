"""GPU (-m gpu): parity of the HIP sky-sum, called through the C-ABI, against the golden vectors
(reference statements), the oracle on seeded inputs, analytic KATs, and size-independent properties at
BASELINE.json's full headline size.

Tolerances (stated relative to S_f = sum_s |pbflux[s,f]|):
    fp64:  max |dV| <= 1e-11 * S_f        fp32:  max |dV| <= 5e-6 * S_f
"""
import numpy as NP
import pytest

from oracle import skyvis_oracle as O, c_oracle as CO
from prisim_amd import _abi, workloads as W

pytestmark = pytest.mark.gpu

TOL = {_abi.PRISIM_FP64: 1e-11, _abi.PRISIM_FP32: 5e-6}
C = 299792458.0


def relerr(v, ref, pb):
    return float(NP.max(NP.abs(v - ref) / O.abs_flux_sum(pb)[None, :]))


def _fwhm(g):
    return NP.sqrt(g['src_shape'][:, 0] * g['src_shape'][:, 1])


# ---------------------------------------------------------------- golden vectors (reference statements)
@pytest.mark.parametrize('prec', [_abi.PRISIM_FP64, _abi.PRISIM_FP32])
@pytest.mark.parametrize('kernel', [_abi.PRISIM_KERNEL_RECURRENCE, _abi.PRISIM_KERNEL_DIRECT])
def test_golden_fp64_reference_path(ctx, golden_skyvis, prec, kernel):
    g = golden_skyvis
    ctx.set_array(g['baselines'], g['channels'])
    ctx.set_tuning(0, 0, 0)
    v, gr = ctx.skyvis(g['dircos'], g['pbfluxes'], g['pc_dircos'], precision=prec, kernel=kernel, want_grad=True)
    assert relerr(v, g['skyvis_f64'], g['pbfluxes']) <= TOL[prec]
    assert max(relerr(gr[k], g['grad_f64'][k], g['pbfluxes']) for k in range(3)) <= TOL[prec]
    v, gr = ctx.skyvis(g['dircos'], g['pbfluxes'], g['pc_dircos'], fwhm_deg=_fwhm(g), precision=prec, kernel=kernel, want_grad=True)
    assert relerr(v, g['skyvis_f64_taper'], g['pbfluxes']) <= TOL[prec]
    assert max(relerr(gr[k], g['grad_f64_taper'][k], g['pbfluxes']) for k in range(3)) <= TOL[prec]


@pytest.mark.parametrize('flush', [None, '16'])
@pytest.mark.parametrize('nsplit', [1, 3])
@pytest.mark.parametrize('ct', [8, 16, 32, 64])
def test_golden_vectors_through_every_tile_split_and_flush(ctx, golden_skyvis, ct, nsplit, flush, monkeypatch):
    """the vectors the reference's own statements produced (tests/golden, 24- and 32-channel grids: ragged in every tile
    wider than 8) straight through the WIDE-tile kernels instead of only the planner's narrow choice for a 9 x 24 x 37 problem --
    k_skyvis_rec_f32pk<32|64> (packed fp32, with and without the taper), k_skyvis_rec<double, 16|32>, the grouped fp64 taper kernel
    k_skyvis_taper_f64<16|32> (as wave items when the sources are split), the fused gradient kernels (MFMA fp64, packed fp32) -- with the
    sources split into partial cubes and with the fp32 accumulators flushed every 16 sources (read-modify-write flushes)."""
    g = golden_skyvis
    if flush is not None:
        monkeypatch.setenv('PRISIM_HIP_FLUSH_SRC', flush)
    ctx.set_array(g['baselines'], g['channels'])
    ctx.set_tuning(ct, 0, nsplit)
    try:
        for prec in (_abi.PRISIM_FP64, _abi.PRISIM_FP32):
            for fw, vkey, gkey in ((None, 'skyvis_f64', 'grad_f64'), (_fwhm(g), 'skyvis_f64_taper', 'grad_f64_taper')):
                v = ctx.skyvis(g['dircos'], g['pbfluxes'], g['pc_dircos'], fwhm_deg=fw, precision=prec)
                tm = ctx.timing()
                assert tm['last_chan_tile'] == (min(ct, 32) if prec == _abi.PRISIM_FP64 else ct), tm       # the tile asked for did run
                assert relerr(v, g[vkey], g['pbfluxes']) <= TOL[prec], (prec, fw is not None, tm)
                v2, gr = ctx.skyvis(g['dircos'], g['pbfluxes'], g['pc_dircos'], fwhm_deg=fw, precision=prec, want_grad=True)
                assert relerr(v2, g[vkey], g['pbfluxes']) <= TOL[prec]
                assert max(relerr(gr[k], g[gkey][k], g['pbfluxes']) for k in range(3)) <= TOL[prec]
    finally:
        ctx.set_tuning(0, 0, 0)


def test_golden_memsave_reference_path(ctx, golden_skyvis):
    """The reference's own fp32 path (:6323) forms the phase in fp32 and is ~1e-5 S_f away from its fp64 path;
    our fp32 mode must be at least as close to the reference fp32 result as that intrinsic error."""
    g = golden_skyvis
    ctx.set_array(g['baselines'], g['channels'])
    v = ctx.skyvis(g['dircos'], g['pbfluxes'], g['pc_dircos'], precision=_abi.PRISIM_FP32, complex64=True)
    assert v.dtype == NP.complex64
    ref_gap = relerr(g['skyvis_f32'], g['skyvis_f64'], g['pbfluxes'])
    assert relerr(v, g['skyvis_f32'], g['pbfluxes']) <= ref_gap + 5e-6
    assert relerr(v, g['skyvis_f64'], g['pbfluxes']) <= 5e-6


# ---------------------------------------------------------------- oracle on seeded inputs, all kernel variants
def _random_case(seed, nbl, nchan, nsrc, taper, maxbl=300.0, df=97656.25):
    rng = NP.random.default_rng(seed)
    bl = rng.uniform(-maxbl, maxbl, size=(nbl, 3))
    bl[:, 2] *= 0.01
    ch = 100e6 + NP.arange(nchan) * df
    alt = NP.degrees(NP.arcsin(rng.uniform(NP.sin(NP.radians(5.0)), 1.0, nsrc)))
    dc = O.altaz2dircos(NP.stack((alt, rng.uniform(0, 360, nsrc)), axis=1))
    pb = rng.uniform(0.0, 10.0, size=(nsrc, 1)) * (ch / 150e6).reshape(1, -1) ** -0.8 * rng.uniform(0.2, 1.0, size=(nsrc, nchan))
    pc = O.altaz2dircos(NP.array([[80.0, 30.0]]))[0]
    fw = None
    if taper:
        fw = rng.uniform(0.0, 1.5, nsrc)
        fw[::5] = 0.0
    return bl, ch, dc, pb, pc, fw


@pytest.mark.parametrize('shape', [(3, 64, 100), (171, 256, 600), (300, 100, 777), (1, 1, 1), (65, 9, 2), (257, 33, 65)])
@pytest.mark.parametrize('taper', [False, True])
def test_oracle_parity_all_variants(ctx, shape, taper):
    nbl, nchan, nsrc = shape
    bl, ch, dc, pb, pc, fw = _random_case(11, nbl, nchan, nsrc, taper)
    ref = O.skyvis(bl, ch, dc, pb, pc, fwhm_deg=fw)
    ctx.set_array(bl, ch)
    ctx.set_sky(dc, pb, pc, fwhm_deg=fw)
    for prec in (_abi.PRISIM_FP64, _abi.PRISIM_FP32):
        cts = [0, 8, 16, 32] + ([64] if prec == _abi.PRISIM_FP32 else [])
        for ct in cts:
            for nsplit in (0, 1, 3):
                ctx.set_tuning(ct, 0, nsplit)
                ctx.compute(precision=prec, kernel=_abi.PRISIM_KERNEL_RECURRENCE)
                assert relerr(ctx.get_vis(), ref, pb) <= TOL[prec], (shape, taper, prec, ct, nsplit)
        for chunk in (1, 7, 64, 256):
            ctx.set_tuning(0, chunk, 0)
            ctx.compute(precision=prec, kernel=_abi.PRISIM_KERNEL_RECURRENCE)
            assert relerr(ctx.get_vis(), ref, pb) <= TOL[prec], (shape, taper, prec, 'chunk', chunk)
    ctx.set_tuning(0, 0, 0)
    ctx.compute(precision=_abi.PRISIM_FP64, kernel=_abi.PRISIM_KERNEL_DIRECT)
    assert relerr(ctx.get_vis(), ref, pb) <= TOL[_abi.PRISIM_FP64]


@pytest.mark.parametrize('shape', [(3, 64, 100), (171, 256, 601), (70, 100, 33), (1, 1, 1), (257, 33, 66)])
@pytest.mark.parametrize('taper', [False, True])
def test_fused_mfma_gradient_matches_oracle_and_four_pass_form(ctx, monkeypatch, shape, taper):
    """interferometry.py:6330, 6338, 6343.  V and the three gradient sums come out of ONE pass (fp64: k_skyvis_grad_f64, the four sums are
    a 4 x 4 x 4 fp64 MFMA per channel with lanes = 16 baselines x 4 sources; fp32: k_skyvis_grad_f32pk, four accumulator sets fed by one
    packed term) -- against the numpy oracle, and against the four separate passes they replace; ragged sizes (baselines not a multiple of 16, sources not a multiple of 4, channels not a multiple
    of the tile), long baselines (no lifting), source chunks that do not divide by four, both precisions' entry."""
    nbl, nchan, nsrc = shape
    bl, ch, dc, pb, pc, fw = _random_case(23, nbl, nchan, nsrc, taper)
    bl[::3] *= 12.0                                                   # some baseline groups beyond the lifting guarantee
    ref, gref = O.skyvis(bl, ch, dc, pb, pc, fwhm_deg=fw, gradient=True)
    ctx.set_array(bl, ch)
    ctx.set_sky(dc, pb, pc, fwhm_deg=fw)
    for chunk in (0, 7):
        ctx.set_tuning(0, chunk, 0)
        for prec in (_abi.PRISIM_FP64, _abi.PRISIM_FP32):               # fp64: MFMA kernel; fp32: GRAD bodies of the packed kernel
            ctx.compute(precision=prec, want_grad=True)
            v, g = ctx.get_vis(want_grad=True)
            assert relerr(v, ref, pb) <= TOL[prec] and max(relerr(g[k], gref[k], pb) for k in range(3)) <= TOL[prec], (shape, taper, prec, chunk)
    ctx.compute(precision=_abi.PRISIM_FP64, want_grad=True)
    v, g = ctx.get_vis(want_grad=True)
    ctx.set_tuning(0, 0, 0)
    monkeypatch.setenv('PRISIM_HIP_FUSED_GRAD', '0')
    ctx.compute(precision=_abi.PRISIM_FP64, want_grad=True)
    v4, g4 = ctx.get_vis(want_grad=True)
    ctx.compute(precision=_abi.PRISIM_FP32, want_grad=True)
    v4s, g4s = ctx.get_vis(want_grad=True)
    monkeypatch.delenv('PRISIM_HIP_FUSED_GRAD')
    assert relerr(v4, v, pb) <= 1e-12 and max(relerr(g4[k], g[k], pb) for k in range(3)) <= 1e-12
    monkeypatch.setenv('PRISIM_HIP_FLUSH_SRC', '16')                 # fp32: several read-modify-write flushes of all four cubes
    ctx.compute(precision=_abi.PRISIM_FP32, want_grad=True)
    vs, gs = ctx.get_vis(want_grad=True)
    assert relerr(vs, ref, pb) <= TOL[_abi.PRISIM_FP32] and max(relerr(gs[k], gref[k], pb) for k in range(3)) <= TOL[_abi.PRISIM_FP32]
    assert relerr(vs, v4s, pb) <= 2 * TOL[_abi.PRISIM_FP32]


@pytest.mark.parametrize('taper', [False, True])
def test_fp32_periodic_flush_read_modify_write(ctx, monkeypatch, taper):
    """fp32 partial sums are added into the fp64 cube every flush_src sources (16384 in production, config 5 crosses it 24
    times per launch); the test hook forces a flush every 37 sources so segments, the transposed stores and the
    read-modify-write path are exercised on a sky the oracle finishes in seconds, for every fp32 tile width and a source split."""
    bl, ch, dc, pb, pc, fw = _random_case(23, 300, 200, 500, taper)
    ref = O.skyvis(bl, ch, dc, pb, pc, fwhm_deg=fw)
    monkeypatch.setenv('PRISIM_HIP_FLUSH_SRC', '37')
    ctx.set_array(bl, ch)
    ctx.set_sky(dc, pb, pc, fwhm_deg=fw)
    for ct in (8, 16, 32, 64):
        for nsplit in (1, 2):
            ctx.set_tuning(ct, 0, nsplit)
            ctx.compute(precision=_abi.PRISIM_FP32, kernel=_abi.PRISIM_KERNEL_RECURRENCE)
            assert relerr(ctx.get_vis(), ref, pb) <= TOL[_abi.PRISIM_FP32], (taper, ct, nsplit)
    ctx.set_tuning(0, 0, 0)


def test_fp32_taper_grouped_and_exact_recurrences(ctx, monkeypatch):
    """The packed fp32 taper kernel has two forms of the amplitude recurrence: exact per step, and grouped (8 steps at the group's
    mean ratio + the known parabola put back on pbflux), which the host selects only when df/f_min <= 3.4e-3.  Strong taper
    (degree-scale sources on km baselines: w spans 1 ... 1e-6 across a tile), both forms against the oracle and each other."""
    rng = NP.random.default_rng(31)
    nbl, nchan, nsrc = 256, 128, 300
    bl = rng.uniform(-1200.0, 1200.0, size=(nbl, 3)); bl[:, 2] *= 0.01
    bl[:8] *= 0.02                                                                     # a few short ones: w ~ 1
    ch = 120e6 + NP.arange(nchan) * 97656.25
    alt = NP.degrees(NP.arcsin(rng.uniform(NP.sin(NP.radians(8.0)), 1.0, nsrc)))
    dc = O.altaz2dircos(NP.stack((alt, rng.uniform(0, 360, nsrc)), axis=1))
    pb = rng.uniform(0.5, 10.0, size=(nsrc, 1)) * rng.uniform(0.5, 1.0, size=(nsrc, nchan))
    pc = NP.array([0.0, 0.0, 1.0])
    fw = rng.uniform(0.02, 0.4, nsrc)
    ref = O.skyvis(bl, ch, dc, pb, pc, fwhm_deg=fw)
    ctx.set_array(bl, ch)
    ctx.set_sky(dc, pb, pc, fwhm_deg=fw)
    res = {}
    for ct in (32, 64):
        ctx.set_tuning(ct, 0, 1)
        for form in (0, 1):
            monkeypatch.setenv('PRISIM_HIP_TAPER_GROUP', str(form))
            ctx.compute(precision=_abi.PRISIM_FP32, kernel=_abi.PRISIM_KERNEL_RECURRENCE)
            assert ctx.timing()['last_taper_group'] == form
            res[(ct, form)] = ctx.get_vis()
            assert relerr(res[(ct, form)], ref, pb) <= TOL[_abi.PRISIM_FP32], (ct, form)
        assert relerr(res[(ct, 1)], res[(ct, 0)], pb) <= 1e-6
    monkeypatch.delenv('PRISIM_HIP_TAPER_GROUP')
    ctx.compute(precision=_abi.PRISIM_FP32, kernel=_abi.PRISIM_KERNEL_RECURRENCE)
    assert ctx.timing()['last_taper_group'] == 1                                       # df/f_min = 8.1e-4: grouped form by default
    # coarse channels (df/f_min = 1.3e-2): the host must fall back to the exact form
    ch2 = 30e6 + NP.arange(nchan) * 390625.0
    ref2 = O.skyvis(bl[:64], ch2, dc, pb, pc, fwhm_deg=fw)
    ctx.set_array(bl[:64], ch2)
    ctx.set_sky(dc, pb, pc, fwhm_deg=fw)
    ctx.set_tuning(64, 0, 1)
    ctx.compute(precision=_abi.PRISIM_FP32, kernel=_abi.PRISIM_KERNEL_RECURRENCE)
    assert ctx.timing()['last_taper_group'] == 0 and ctx.timing()['last_chan_tile'] == 32     # and onto 32-channel tiles, whatever was asked
    assert relerr(ctx.get_vis(), ref2, pb) <= TOL[_abi.PRISIM_FP32]
    ctx.set_tuning(0, 0, 0)


def test_fp32_split_taper_runs_of_one_source_size(ctx, monkeypatch):
    """Skies whose sources come in runs of ONE size each (every HEALPix sky; point sources followed by a diffuse map) run the packed fp32
    taper in its split form: exp(-kappa |b|^2 f^2/c^2) once per flush, exp(+kappa (b.s)^2 f^2/c^2) in the recurrence, and -- where the
    host's beam-weighted bound allows -- no parabola correction inside a group.  Point-source runs (size 0) take the no-taper bodies.
    Checked against the oracle and against the unsplit kernel, on a sky concentrated near the zenith (bound passes: uncorrected
    bodies) and on one spread to the horizon with large sources (bound fails: corrected bodies); flushes every 97 sources so that
    the per-flush factor and the read-modify-write across runs are exercised."""
    rng = NP.random.default_rng(77)
    nbl, nchan = 520, 128
    bl = rng.uniform(-280.0, 280.0, size=(nbl, 3)); bl[:, 2] *= 0.005
    bl[-40:] *= 3.0                                                       # a last group with longer baselines (re-anchored bodies)
    ch = 150e6 + (NP.arange(nchan) - 64) * 97656.25
    pc = NP.array([0.0, 0.0, 1.0])
    monkeypatch.setenv('PRISIM_HIP_FLUSH_SRC', '97')
    ctx.set_array(bl, ch)
    ctx.set_tuning(64, 0, 1)
    for name, alt_lo, fw_d, want_uncorrected in (('zenith', 82.0, 0.229, True), ('horizon', 5.0, 0.5, False)):
        n_pt, n_df = 150, 450
        alt = NP.degrees(NP.arcsin(rng.uniform(NP.sin(NP.radians(alt_lo)), 1.0, n_pt + n_df)))
        dc = O.altaz2dircos(NP.stack((alt, rng.uniform(0, 360, n_pt + n_df)), axis=1))
        pb = rng.uniform(0.5, 10.0, size=(n_pt + n_df, 1)) * rng.uniform(0.5, 1.0, size=(n_pt + n_df, nchan))
        fw = NP.concatenate((NP.zeros(n_pt), NP.full(n_df, fw_d)))
        ref = CO.skyvis(bl, ch, dc, pb, pc, fwhm_deg=fw)
        ctx.set_sky(dc, pb, pc, fwhm_deg=fw)
        ctx.compute(precision=_abi.PRISIM_FP32)
        tm = ctx.timing()
        assert tm['last_chan_tile'] == 64 and tm['last_taper_split'] == 2, (name, tm)
        assert (tm['last_split_uncorrected_groups'] > 0) == want_uncorrected, (name, tm)
        v_split = ctx.get_vis()
        assert relerr(v_split, ref, pb) <= TOL[_abi.PRISIM_FP32], name
        monkeypatch.setenv('PRISIM_HIP_TAPER_SPLIT', '0')
        ctx.compute(precision=_abi.PRISIM_FP32)
        assert ctx.timing()['last_taper_split'] == 0
        assert relerr(ctx.get_vis(), v_split, pb) <= 1.5e-6, name
        monkeypatch.delenv('PRISIM_HIP_TAPER_SPLIT')
    # a sky that is ONE run also takes the split form when its sources are cut into partial cubes (baseline shards, config 4):
    # every partial carries the flush factor
    fw1 = NP.full(n_pt + n_df, 0.229)
    ref1 = CO.skyvis(bl, ch, dc, pb, pc, fwhm_deg=fw1)
    ctx.set_sky(dc, pb, pc, fwhm_deg=fw1)
    for nsplit in (1, 2, 5):
        ctx.set_tuning(64, 0, nsplit)
        ctx.compute(precision=_abi.PRISIM_FP32)
        tm = ctx.timing()
        assert tm['last_taper_split'] == 1 and tm['last_nsplit'] == nsplit, tm
        assert relerr(ctx.get_vis(), ref1, pb) <= TOL[_abi.PRISIM_FP32], nsplit
    # two runs + a source split (a baseline shard of a mixed sky): still run by run, every run into its own set of partial cubes
    ref2 = CO.skyvis(bl, ch, dc, pb, pc, fwhm_deg=fw)
    ctx.set_sky(dc, pb, pc, fwhm_deg=fw)
    for nsplit in (2, 3):
        ctx.set_tuning(64, 0, nsplit)
        ctx.compute(precision=_abi.PRISIM_FP32)
        tm = ctx.timing()
        assert tm['last_taper_split'] == 2 and tm['last_nsplit'] == nsplit, tm
        assert relerr(ctx.get_vis(), ref2, pb) <= TOL[_abi.PRISIM_FP32], nsplit
    monkeypatch.delenv('PRISIM_HIP_FLUSH_SRC')                            # ... and with single-flush (complex64) partials
    ctx.compute(precision=_abi.PRISIM_FP32)
    assert ctx.timing()['last_taper_split'] == 2 and relerr(ctx.get_vis(), ref2, pb) <= TOL[_abi.PRISIM_FP32]
    monkeypatch.setenv('PRISIM_HIP_FLUSH_SRC', '97')
    ctx.set_tuning(64, 0, 1)
    # sizes that vary from source to source: no runs, the unsplit kernel
    fw = rng.uniform(0.05, 0.4, n_pt + n_df)
    ctx.set_sky(dc, pb, pc, fwhm_deg=fw)
    ctx.compute(precision=_abi.PRISIM_FP32)
    assert ctx.timing()['last_taper_split'] == 0
    assert relerr(ctx.get_vis(), CO.skyvis(bl, ch, dc, pb, pc, fwhm_deg=fw), pb) <= TOL[_abi.PRISIM_FP32]
    ctx.set_tuning(0, 0, 0)


def test_taper_culling_skips_only_what_is_below_the_tolerance(ctx, monkeypatch):
    """Long baselines resolve out diffuse pixels: w = exp(-kappa |b_perp|^2 f^2/c^2) underflows for the sources nearest the zenith.  With
    a run's sources listed by decreasing altitude the library starts every baseline group's source loop behind the leading sources whose
    summed weight is below exp(-18) of sum|pbflux| (packed fp32 kernels; exp(-28) for the grouped fp64 kernel).  MWA-like baselines (to 2.5 km, sorted by length like the
    driver's) over degree-size pixels: a good share of the (source, baseline) pairs goes, the result stays inside the tolerances against
    the oracle that sums everything, and equals the unculled result to the cull bound."""
    rng = NP.random.default_rng(91)
    nbl, nchan, nsrc = 1024, 64, 1500
    xy = rng.normal(0.0, 800.0, size=(nbl, 2))
    bl = NP.hstack((xy, rng.normal(0.0, 0.5, size=(nbl, 1))))
    bl = bl[NP.argsort(NP.sqrt(NP.sum(bl ** 2, axis=1)))]
    ch = 170e6 + NP.arange(nchan) * 40e3
    alt = NP.sort(NP.degrees(NP.arcsin(rng.uniform(NP.sin(NP.radians(3.0)), 1.0, nsrc))))[::-1]       # decreasing altitude
    dc = O.altaz2dircos(NP.stack((alt, rng.uniform(0, 360, nsrc)), axis=1))
    pb = rng.uniform(0.5, 10.0, size=(nsrc, 1)) * rng.uniform(0.5, 1.0, size=(nsrc, nchan))
    fw = NP.full(nsrc, 0.916)                                                                          # nside-64 pixels
    pc = NP.array([0.0, 0.0, 1.0])
    ref = CO.skyvis(bl, ch, dc, pb, pc, fwhm_deg=fw)
    ctx.set_array(bl, ch)
    ctx.set_tuning(0, 0, 0)
    for prec, bound in ((_abi.PRISIM_FP32, 2e-8), (_abi.PRISIM_FP64, 1e-12)):
        res = {}
        for cull in ('1', '0'):
            monkeypatch.setenv('PRISIM_HIP_TAPER_CULL', cull)
            ctx.set_sky(dc, pb, pc, fwhm_deg=fw)
            ctx.compute(precision=prec)
            res[cull] = ctx.get_vis()
            frac = ctx.timing()['last_culled_fraction']
            # (the packed fp32 kernels cull below exp(-18), the grouped fp64 kernel below exp(-28) of sum|pbflux|)
            assert (frac > (0.20 if prec == _abi.PRISIM_FP32 else 0.10)) if cull == '1' else (frac == 0.0), (prec, cull, frac)
            assert relerr(res[cull], ref, pb) <= TOL[prec], (prec, cull)
        assert relerr(res['1'], res['0'], pb) <= bound, prec
    monkeypatch.delenv('PRISIM_HIP_TAPER_CULL')
    # sources in no particular order: little or nothing to skip, same answer
    perm = rng.permutation(nsrc)
    ctx.set_sky(dc[perm], pb[perm], pc, fwhm_deg=fw)
    ctx.compute(precision=_abi.PRISIM_FP32)
    assert ctx.timing()['last_culled_fraction'] < 0.05 and relerr(ctx.get_vis(), ref, pb) <= TOL[_abi.PRISIM_FP32]


def test_fp64_grouped_taper_runs_culling_and_slow_path(ctx, monkeypatch):
    """fp64 requests with the source-shape taper (the reference's default precision, interferometry.py:6332-6335) run the grouped kernel
    k_skyvis_taper_f64 on 16- / 32-channel tiles: one chain per tile, the amplitude ratio held over groups of 8 channels and the known
    in-group factor put back exactly on the pbflux operand.  Checked against the C oracle (1e-11 S_f) and against the exact second-order
    kernel (PRISIM_HIP_TAPER_F64_GROUP=0) on: a sky of three source runs (point sources + two pixel sizes; run by run with accumulation
    at nsplit = 1, one launch under a source split), sizes that vary source by source, a coarse descending channel grid with very
    large sources (the wave-uniform library-exp path), and long baselines whose second run sheds its leading sources (culling)."""
    rng = NP.random.default_rng(404)
    nbl = 600
    bl = rng.uniform(-290.0, 290.0, size=(nbl, 3)); bl[:, 2] *= 0.01
    pc = O.altaz2dircos(NP.array([[84.0, 200.0]]))[0]
    nsrc = 900
    alt = NP.degrees(NP.arcsin(rng.uniform(NP.sin(NP.radians(4.0)), 1.0, nsrc)))
    dc = O.altaz2dircos(NP.stack((alt, rng.uniform(0, 360, nsrc)), axis=1))

    def both(bl_, ch_, dc_, pb_, fw_, tunings, expect_group=True):
        ref = CO.skyvis(bl_, ch_, dc_, pb_, pc, fwhm_deg=fw_)
        ctx.set_array(bl_, ch_)
        ctx.set_sky(dc_, pb_, pc, fwhm_deg=fw_)
        for ct, nsplit in tunings:
            ctx.set_tuning(ct, 0, nsplit)
            monkeypatch.setenv('PRISIM_HIP_TAPER_F64_GROUP', '1')
            ctx.compute(precision=_abi.PRISIM_FP64)
            vg = ctx.get_vis()
            assert relerr(vg, ref, pb_) <= TOL[_abi.PRISIM_FP64], (ct, nsplit)
            monkeypatch.setenv('PRISIM_HIP_TAPER_F64_GROUP', '0')
            ctx.compute(precision=_abi.PRISIM_FP64)
            assert relerr(ctx.get_vis(), vg, pb_) <= 1e-12, (ct, nsplit)
        monkeypatch.delenv('PRISIM_HIP_TAPER_F64_GROUP')
        ctx.set_tuning(0, 0, 0)

    ch = 150e6 + (NP.arange(100) - 50) * 97656.25                        # 100 channels: a ragged last tile at 16 and at 32
    pb = rng.uniform(0.5, 10.0, size=(nsrc, 1)) * rng.uniform(0.5, 1.0, size=(nsrc, ch.size))
    fw_runs = NP.concatenate((NP.zeros(200), NP.full(400, 0.458), NP.full(300, 0.916)))
    both(bl, ch, dc, pb, fw_runs, [(32, 1), (16, 1), (32, 3), (16, 2), (0, 0)])
    both(bl, ch, dc, pb, rng.uniform(0.0, 1.2, nsrc), [(32, 1), (16, 0)])
    # coarse, descending grid + degree-scale sources on 2.5 km baselines: |2 g df f| and g df^2 leave the series' range
    ch2 = 200e6 - NP.arange(40) * 3.0e6
    bl2 = bl * 8.0
    pb2 = rng.uniform(0.5, 10.0, size=(nsrc, 1)) * rng.uniform(0.5, 1.0, size=(nsrc, ch2.size))
    both(bl2, ch2, dc, pb2, NP.full(nsrc, 2.0), [(32, 1), (16, 1)])
    # culling in the SECOND run: point sources, then nside-64-size pixels by decreasing altitude under MWA-like baselines
    xy = rng.normal(0.0, 800.0, size=(1024, 2))
    bl3 = NP.hstack((xy, rng.normal(0.0, 0.5, size=(1024, 1))))
    bl3 = bl3[NP.argsort(NP.sqrt(NP.sum(bl3 ** 2, axis=1)))]
    ch3 = 170e6 + NP.arange(64) * 40e3
    alt3 = NP.concatenate((alt[:100], NP.sort(alt[100:])[::-1]))
    dc3 = O.altaz2dircos(NP.stack((alt3, rng.uniform(0, 360, nsrc)), axis=1))
    pb3 = rng.uniform(0.5, 10.0, size=(nsrc, 1)) * rng.uniform(0.5, 1.0, size=(nsrc, ch3.size))
    fw3 = NP.concatenate((NP.zeros(100), NP.full(nsrc - 100, 0.916)))
    ref3 = CO.skyvis(bl3, ch3, dc3, pb3, pc, fwhm_deg=fw3)
    ctx.set_array(bl3, ch3)
    ctx.set_sky(dc3, pb3, pc, fwhm_deg=fw3)
    for ct in (32, 16):
        ctx.set_tuning(ct, 0, 1)
        ctx.compute(precision=_abi.PRISIM_FP64)
        assert ctx.timing()['last_culled_fraction'] > 0.05, ctx.timing()
        assert relerr(ctx.get_vis(), ref3, pb3) <= TOL[_abi.PRISIM_FP64], ct
    ctx.set_tuning(0, 0, 0)


def test_fp64_taper_wave_items_on_small_arrays(ctx, monkeypatch):
    """Arrays of at most 256 baselines whose sources are split run the grouped fp64 taper kernel with WAVE items
    (k_skyvis_taper_f64_wave: item = (baseline wave, source split), four per block) so that every SIMD carries sources when the array
    fills 1-3 wavefronts (config 1: 3 baselines, config 2: 171).  Against the C oracle and against block items
    (PRISIM_HIP_WAVE_ITEMS=0) for 3 / 64 / 65 / 171 / 256 baselines, item counts that are and are not multiples of four, splits that do
    not divide the sky, the planner's own choice, and a sky whose leading sources are culled."""
    rng = NP.random.default_rng(271)
    pc = O.altaz2dircos(NP.array([[86.0, 40.0]]))[0]
    nsrc = 333
    alt = NP.degrees(NP.arcsin(rng.uniform(NP.sin(NP.radians(5.0)), 1.0, nsrc)))
    dc = O.altaz2dircos(NP.stack((alt, rng.uniform(0, 360, nsrc)), axis=1))
    ch = 150e6 + (NP.arange(72) - 36) * 390625.0                        # ragged last tile at 16 and at 32
    pb = rng.uniform(0.5, 10.0, size=(nsrc, 1)) * rng.uniform(0.5, 1.0, size=(nsrc, ch.size))
    fw = NP.full(nsrc, 3.66)
    for nbl in (3, 64, 65, 171, 256):
        bl = rng.uniform(-60.0, 60.0, size=(nbl, 3)); bl[:, 2] *= 0.02
        ref = CO.skyvis(bl, ch, dc, pb, pc, fwhm_deg=fw)
        ctx.set_array(bl, ch)
        ctx.set_sky(dc, pb, pc, fwhm_deg=fw)
        for ct, nsplit in ((16, 2), (16, 7), (32, 5), (16, 20), (0, 0)):
            ctx.set_tuning(ct, 0, nsplit)
            ctx.compute(precision=_abi.PRISIM_FP64)
            vw = ctx.get_vis()
            t = ctx.timing()
            assert relerr(vw, ref, pb) <= TOL[_abi.PRISIM_FP64], (nbl, ct, nsplit, t)
            if (ct, nsplit) == (0, 0):
                assert t['last_nsplit'] > 1, t               # the planner splits a small array's sources (and so takes this path)
            monkeypatch.setenv('PRISIM_HIP_WAVE_ITEMS', '0')
            ctx.compute(precision=_abi.PRISIM_FP64)
            assert relerr(ctx.get_vis(), vw, pb) <= 1e-13, (nbl, ct, nsplit)
            monkeypatch.delenv('PRISIM_HIP_WAVE_ITEMS')
    # a mixed sky (point sources, then two pixel sizes) on 171 baselines with split sources: run by run, every run into its own set of
    # partial cubes -- the taper runs as wave items, the point-source run through the fp64 kernel without the taper (block items)
    fw_runs = NP.concatenate((NP.zeros(90), NP.full(143, 3.66), NP.full(100, 1.83)))
    bl = rng.uniform(-60.0, 60.0, size=(171, 3)); bl[:, 2] *= 0.02
    ref = CO.skyvis(bl, ch, dc, pb, pc, fwhm_deg=fw_runs)
    ctx.set_array(bl, ch)
    ctx.set_sky(dc, pb, pc, fwhm_deg=fw_runs)
    for ct, nsplit in ((16, 4), (32, 3), (0, 0)):
        ctx.set_tuning(ct, 0, nsplit)
        ctx.compute(precision=_abi.PRISIM_FP64)
        assert relerr(ctx.get_vis(), ref, pb) <= TOL[_abi.PRISIM_FP64], (ct, nsplit, ctx.timing())
    # culling: large pixels by decreasing altitude under 1.5 km baselines, sources split
    ang, rad = rng.uniform(0, 2 * NP.pi, 200), rng.uniform(1200.0, 1500.0, 200)       # every baseline long: the one group sheds sources
    bl = NP.stack((rad * NP.cos(ang), rad * NP.sin(ang), rng.normal(0.0, 0.5, size=200)), axis=1)
    ch3 = 170e6 + NP.arange(64) * 40e3
    dc3 = O.altaz2dircos(NP.stack((NP.sort(alt)[::-1], rng.uniform(0, 360, nsrc)), axis=1))
    pb3 = rng.uniform(0.5, 10.0, size=(nsrc, 1)) * rng.uniform(0.5, 1.0, size=(nsrc, ch3.size))
    fw3 = NP.full(nsrc, 0.916)
    ref3 = CO.skyvis(bl, ch3, dc3, pb3, pc, fwhm_deg=fw3)
    ctx.set_array(bl, ch3)
    ctx.set_sky(dc3, pb3, pc, fwhm_deg=fw3)
    for ct, nsplit in ((16, 6), (32, 3)):
        ctx.set_tuning(ct, 0, nsplit)
        ctx.compute(precision=_abi.PRISIM_FP64)
        assert relerr(ctx.get_vis(), ref3, pb3) <= TOL[_abi.PRISIM_FP64], (ct, nsplit, ctx.timing())
        assert ctx.timing()['last_culled_fraction'] > 0.05, ctx.timing()
    ctx.set_tuning(0, 0, 0)


@pytest.mark.parametrize('taper', [False, True])
def test_fp32_single_source_worst_case_per_term(ctx, taper):
    """One source, so nothing averages: the error of every (baseline, channel) term against the fp64 oracle must stay inside the
    5e-6 tolerance on its own, on short (lifting / small-step) and long (re-anchored) HERA-350 baselines, at the ends of the
    64-channel chains.  The recurrences' systematic per-step rounding is what this pins (3.0e-6 measured; 6.9e-6 before the
    amplitude ratio of the taper was carried as q - 1 and long-baseline chains were re-anchored at their midpoint)."""
    cfg = W.config3()
    bl = NP.vstack((cfg['baselines'][::37][-600:], cfg['baselines'][-200:]))
    ch = cfg['channels']
    rng = NP.random.default_rng(3)
    ctx.set_array(bl, ch)
    ctx.set_tuning(64, 0, 1)                       # the tile width the full array runs with (a small array would get 8)
    worst = 0.0
    for trial in range(6):
        dc = O.altaz2dircos(NP.array([[rng.uniform(8, 80), rng.uniform(0, 360)]]))
        pb = rng.uniform(0.5, 2.0, size=(1, ch.size))
        fw = NP.array([rng.uniform(0.05, 0.3)]) if taper else None
        ref = CO.skyvis(bl, ch, dc, pb, NP.array([0.0, 0.0, 1.0]), fwhm_deg=fw)
        ctx.set_sky(dc, pb, NP.array([0.0, 0.0, 1.0]), fwhm_deg=fw)
        ctx.compute(precision=_abi.PRISIM_FP32)
        assert ctx.timing()['last_chan_tile'] == 64
        worst = max(worst, float(NP.max(NP.abs(ctx.get_vis() - ref) / NP.abs(pb))))
    ctx.set_tuning(0, 0, 0)
    assert worst <= 4.0e-6, worst


def test_config2_fp64_full(ctx):
    """BASELINE config 2: HERA-19 x 256 ch x nside-16 diffuse sky, Airy beam, taper ON, fp64 -- full size."""
    cfg = W.config2()
    bl, ch, sky = cfg['baselines'], cfg['channels'], cfg['sky']
    zen = NP.array([0.0, 0.0, 1.0])
    ctx.set_array(bl, ch)
    ctx.set_tuning(0, 0, 0)
    ctx.set_sky_analytic(sky['dircos'], sky['flux_ref'], sky['spindex'], sky['ref_freq'], _abi.PRISIM_BEAM_AIRY, 14.0, zen, zen,
                         fwhm_deg=sky['fwhm_deg'])
    pb = ctx.get_pbflux()
    ref = CO.skyvis(bl, ch, sky['dircos'], pb, zen, fwhm_deg=sky['fwhm_deg'])
    for prec in (_abi.PRISIM_FP64, _abi.PRISIM_FP32):
        ctx.compute(precision=prec)
        assert relerr(ctx.get_vis(), ref, pb) <= TOL[prec]


def test_config3_subsample_fp32_and_fp64(ctx):
    """1/64 sub-sample of the headline config 3 (every 8th baseline, every 8th source... of HERA-350 x 1024 ch)."""
    cfg = W.subsample(W.config3(), bl_stride=64, src_stride=8)
    bl, ch, sky = cfg['baselines'], cfg['channels'], cfg['sky']
    zen = NP.array([0.0, 0.0, 1.0])
    ctx.set_array(bl, ch)
    ctx.set_tuning(0, 0, 0)
    ctx.set_sky_analytic(sky['dircos'], sky['flux_ref'], sky['spindex'], sky['ref_freq'], _abi.PRISIM_BEAM_AIRY, 14.0, zen, zen)
    pb = ctx.get_pbflux()
    ref = CO.skyvis(bl, ch, sky['dircos'], pb, zen)
    for prec in (_abi.PRISIM_FP64, _abi.PRISIM_FP32):
        ctx.compute(precision=prec)
        assert relerr(ctx.get_vis(), ref, pb) <= TOL[prec]


def test_lifting_rotation_groups_and_fallback(ctx):
    """fp32 / no taper: baseline groups whose step phase is guaranteed within +-1/8 cycle use the 5-instruction lifting
    rotation, the others the 4-instruction one; both must meet the tolerance, also inside one launch."""
    rng = NP.random.default_rng(77)
    nchan, nsrc = 128, 500
    ch = 150e6 + NP.arange(nchan) * 97656.25
    alt = NP.degrees(NP.arcsin(rng.uniform(NP.sin(NP.radians(10.0)), 1.0, nsrc)))
    dc = O.altaz2dircos(NP.stack((alt, rng.uniform(0, 360, nsrc)), axis=1))
    pb = rng.uniform(0.0, 5.0, (nsrc, nchan))
    zen = NP.array([0.0, 0.0, 1.0])
    ang = rng.uniform(0, 2 * NP.pi, 768)
    length = NP.concatenate((rng.uniform(5.0, 250.0, 512), rng.uniform(400.0, 3000.0, 256)))     # 2 short groups + 1 long group
    bl = NP.stack((length * NP.cos(ang), length * NP.sin(ang), NP.zeros(768)), axis=1)
    ref = CO.skyvis(bl, ch, dc, pb, zen)
    ctx.set_array(bl, ch)
    ctx.set_sky(dc, pb, zen)
    for ct in (64, 32):
        ctx.set_tuning(ct, 0, 1)
        ctx.compute(precision=_abi.PRISIM_FP32)
        vis = ctx.get_vis()
        assert ctx.timing()['last_lift_groups'] == 2
        assert relerr(vis[:512], ref[:512], pb) <= 5e-6          # lifting groups
        assert relerr(vis[512:], ref[512:], pb) <= 5e-6          # step phase up to ~1 cycle: standard rotation
    # single bright source at the edge of the lifting range, worst case for the per-step angle error
    s1 = O.altaz2dircos(NP.array([[10.0, 45.0]]))
    p1 = NP.ones((1, nchan))
    ctx.set_sky(s1, p1, zen)
    ctx.set_tuning(64, 0, 1)
    ctx.compute(precision=_abi.PRISIM_FP32)
    assert ctx.timing()['last_lift_groups'] == 2
    assert relerr(ctx.get_vis()[:512], CO.skyvis(bl[:512], ch, s1, p1, zen), p1) <= 5e-6
    # taper on: the packed fp32 kernel folds the amplitude into the phasor (a scaled rotation, no lifting); the generic kernels
    # (fp64, narrow fp32 tiles) put the taper on pbflux, so their phasor is still a pure rotation and the flagged groups lift
    fw = NP.full(nsrc, 0.3)
    reft = CO.skyvis(bl, ch, dc, pb, zen, fwhm_deg=fw)
    ctx.set_sky(dc, pb, zen, fwhm_deg=fw)
    ctx.compute(precision=_abi.PRISIM_FP32)
    assert ctx.timing()['last_lift_groups'] == 0 and ctx.timing()['last_chan_tile'] == 64
    assert relerr(ctx.get_vis(), reft, pb) <= 5e-6
    for prec, ct, tol in ((_abi.PRISIM_FP64, 32, 1e-11), (_abi.PRISIM_FP64, 16, 1e-11), (_abi.PRISIM_FP32, 16, 5e-6)):
        ctx.set_tuning(ct, 0, 1)
        ctx.compute(precision=prec)
        assert ctx.timing()['last_lift_groups'] == 2
        assert relerr(ctx.get_vis(), reft, pb) <= tol, (prec, ct)
    ctx.set_tuning(0, 0, 0)


# ---------------------------------------------------------------- analytic known answers on the GPU
BL = NP.array([[14.6, 0.0, 0.0], [7.3, 12.644, 0.0], [-250.0, 120.0, 1.5], [0.0, 0.0, 0.0]])
CH = 150e6 + (NP.arange(48) - 24) * 390625.0


@pytest.mark.parametrize('prec', [_abi.PRISIM_FP64, _abi.PRISIM_FP32])
def test_kat_phase_centre_and_closed_form(ctx, prec):
    ctx.set_array(BL, CH)
    ctx.set_tuning(0, 0, 0)
    pc = O.altaz2dircos([[70.0, 123.0]])[0]
    p = NP.linspace(1.0, 2.0, CH.size)[None, :]
    v = ctx.skyvis(pc[None, :], p, pc, precision=prec)
    assert relerr(v, NP.broadcast_to(p, v.shape), p) <= TOL[prec]                     # KAT-1
    s = O.altaz2dircos([[40.0, 250.0]])[0]
    zen = NP.array([0.0, 0.0, 1.0])
    p3 = NP.full((1, CH.size), 3.0)
    v = ctx.skyvis(s[None, :], p3, zen, precision=prec)
    expected = 3.0 * NP.exp(-2j * NP.pi * CH[None, :] * (BL @ (s - zen))[:, None] / C)
    assert relerr(v, expected, p3) <= TOL[prec]                                       # KAT-2


@pytest.mark.parametrize('prec', [_abi.PRISIM_FP64, _abi.PRISIM_FP32])
def test_kat_hermitian_linearity_additivity(ctx, prec):
    rng = NP.random.default_rng(5)
    s = O.altaz2dircos(NP.stack((rng.uniform(10, 90, 300), rng.uniform(0, 360, 300)), 1))
    p = rng.uniform(0, 5, (300, CH.size))
    pc = O.altaz2dircos([[85.0, 0.0]])[0]
    ctx.set_array(BL, CH)
    ctx.set_tuning(0, 0, 0)
    full = ctx.skyvis(s, p, pc, precision=prec)
    ctx.set_array(-BL, CH)
    assert relerr(ctx.skyvis(s, p, pc, precision=prec), NP.conj(full), p) <= 2 * TOL[prec]        # KAT-4
    ctx.set_array(BL, CH)
    assert relerr(ctx.skyvis(s, 2.5 * p, pc, precision=prec), 2.5 * full, 2.5 * p) <= 2 * TOL[prec]   # KAT-5 linear
    parts = ctx.skyvis(s[:101], p[:101], pc, precision=prec) + ctx.skyvis(s[101:], p[101:], pc, precision=prec)
    assert relerr(parts, full, p) <= 2 * TOL[prec]                                    # KAT-5 additive (slab invariance)
    # zero baseline sees the total flux
    assert NP.max(NP.abs(full[3] - p.sum(0)) / p.sum(0)) <= TOL[prec]


@pytest.mark.parametrize('prec', [_abi.PRISIM_FP64, _abi.PRISIM_FP32])
def test_kat3_two_equal_sources_envelope(ctx, prec):
    """KAT-3: two equal sources of flux p give |V| = 2 p |cos(pi f b.(s1 - s2)/c)| -- the beat envelope, a closed form
    no oracle is involved in."""
    zen = NP.array([0.0, 0.0, 1.0])
    s = O.altaz2dircos([[60.0, 10.0], [75.0, 200.0]])
    p = NP.full((2, CH.size), 2.0)
    ctx.set_array(BL, CH)
    ctx.set_tuning(0, 0, 0)
    v = ctx.skyvis(s, p, zen, precision=prec)
    dphi = NP.pi * CH[None, :] * (BL @ (s[0] - s[1]))[:, None] / C
    assert NP.max(NP.abs(NP.abs(v) - NP.abs(4.0 * NP.cos(dphi))) / 4.0) <= TOL[prec]
    # and the phase is that of the mean direction: V = 2 p cos(dphi) exp(-2 pi i f b.((s1 + s2)/2 - zen)/c)
    mean = 4.0 * NP.cos(dphi) * NP.exp(-2j * NP.pi * CH[None, :] * (BL @ (0.5 * (s[0] + s[1]) - zen))[:, None] / C)
    assert relerr(v, mean, p) <= TOL[prec]


def test_kat_taper_limits(ctx):
    zen = NP.array([0.0, 0.0, 1.0])
    s = O.altaz2dircos([[50.0, 90.0]])
    p = NP.ones((1, CH.size))
    ctx.set_array(BL, CH)
    ctx.set_tuning(0, 0, 0)
    a = ctx.skyvis(s, p, zen, fwhm_deg=[0.0])
    b = ctx.skyvis(s, p, zen)
    assert NP.max(NP.abs(a - b)) <= 1e-13                                             # FWHM = 0 -> w = 1
    ctx.set_array(NP.vstack((100.0 * s, BL[:1])), CH)
    v = ctx.skyvis(s, p, zen, fwhm_deg=[1.0])
    assert NP.max(NP.abs(NP.abs(v[0]) - 1.0)) <= 1e-9                                 # b || s -> u_perp = 0 -> w = 1
    assert NP.all(NP.abs(v[1]) < 1.0)


# ---------------------------------------------------------------- edge cases
def test_empty_sky_gives_zeros(ctx):
    ctx.set_array(BL, CH)
    v, g = ctx.skyvis(NP.zeros((0, 3)), NP.zeros((0, CH.size)), [0, 0, 1.0], want_grad=True)
    assert v.shape == (4, CH.size) and NP.all(v == 0) and NP.all(g == 0)


def test_nonuniform_channels_use_direct_kernel(ctx):
    rng = NP.random.default_rng(9)
    ch = NP.sort(rng.uniform(100e6, 200e6, 37))
    bl, _, dc, pb, pc, fw = _random_case(3, 70, 37, 90, True)
    pb = pb[:, :37]
    ctx.set_array(bl, ch)
    ctx.set_tuning(0, 0, 0)
    v = ctx.skyvis(dc, pb, pc, fwhm_deg=fw)
    assert ctx.timing()['last_kernel_id'] == _abi.PRISIM_KERNEL_DIRECT
    assert relerr(v, O.skyvis(bl, ch, dc, pb, pc, fwhm_deg=fw), pb) <= 1e-11
    with pytest.raises(ValueError):
        ctx.compute(kernel=_abi.PRISIM_KERNEL_RECURRENCE)


def test_float32_pbflux_and_device_side_product(ctx):
    bl, ch, dc, pb, pc, fw = _random_case(21, 40, 64, 120, False)
    ctx.set_array(bl, ch)
    ctx.set_tuning(0, 0, 0)
    pb32 = pb.astype(NP.float32)                         # external beams are stored float32 (interferometry.py:4466)
    v = ctx.skyvis(dc, pb32, pc)
    assert relerr(v, O.skyvis(bl, ch, dc, pb32.astype(NP.float64), pc), pb) <= 1e-11
    beam = NP.random.default_rng(2).uniform(0, 1, pb.shape)
    v = ctx.skyvis(dc, beam, pc, fluxes=pb)              # pbfluxes = pb * fluxes formed on the device (:6254)
    assert relerr(v, O.skyvis(bl, ch, dc, beam * pb, pc), beam * pb) <= 1e-11
    v = ctx.skyvis(dc, beam.astype(NP.float32), pc, fluxes=pb)
    assert relerr(v, O.skyvis(bl, ch, dc, beam.astype(NP.float32).astype(NP.float64) * pb, pc), beam * pb) <= 1e-11


def test_argument_errors_map_to_reference_exception_types(ctx):
    with pytest.raises(ValueError):
        ctx.set_array(NP.zeros((0, 3)), CH)
    ctx.set_array(BL, CH, nt_max=2)
    with pytest.raises(ValueError):
        ctx.set_sky(NP.zeros((2, 3)), NP.zeros((2, 5)), [0, 0, 1.0])                  # wrong nchan
    with pytest.raises(ValueError):
        ctx.set_sky(NP.full((2, 3), NP.nan), NP.zeros((2, CH.size)), [0, 0, 1.0])
    with pytest.raises(ValueError):
        ctx.set_sky(NP.zeros((2, 3)), NP.zeros((2, CH.size)), [0, 0, 1.0], fwhm_deg=[-1.0, 0.0])
    ctx.set_sky(O.altaz2dircos([[50.0, 0.0]]), NP.ones((1, CH.size)), [0, 0, 1.0])
    with pytest.raises(ValueError):
        ctx.compute(slot=2)
    with pytest.raises(ValueError):
        ctx.compute(precision=7)
    with pytest.raises(ValueError):
        ctx.set_tuning(12, 0, 0)
    fresh = _abi.Context(0)
    with pytest.raises(RuntimeError):
        fresh.compute()                                                               # compute before set_array
    fresh.close()


def test_snapshot_slots_are_independent(ctx):
    bl, ch, dc, pb, pc, fw = _random_case(31, 50, 32, 64, False)
    ctx.set_array(bl, ch, nt_max=3)
    ctx.set_tuning(0, 0, 0)
    refs = []
    for t in range(3):
        ctx.set_sky(dc, pb * (t + 1), pc)
        ctx.compute(slot=t)
        refs.append(O.skyvis(bl, ch, dc, pb * (t + 1), pc))
    for t in range(3):
        assert relerr(ctx.get_vis(slot=t), refs[t], pb * (t + 1)) <= 1e-11


# ---------------------------------------------------------------- full headline size: size-independent properties
def test_full_size_headline_properties(ctx):
    """HERA-350 x 1024 ch x 1e4 sources (6.25e11 terms per pass), fp32: the oracle cannot run this size in
    seconds, so check (i) a 1/977 baseline sample against the C oracle, (ii) additivity over two disjoint source
    halves, (iii) the zero-spacing identity sum_b... (b = 0 is not in the array, so use V(-b) = conj V(b))."""
    cfg = W.config3()
    bl, ch, sky = cfg['baselines'], cfg['channels'], cfg['sky']
    zen = NP.array([0.0, 0.0, 1.0])
    ctx.set_array(bl, ch)
    ctx.set_tuning(0, 0, 0)
    ctx.set_sky_analytic(sky['dircos'], sky['flux_ref'], sky['spindex'], sky['ref_freq'], _abi.PRISIM_BEAM_AIRY, 14.0, zen, zen)
    pb = ctx.get_pbflux()
    ctx.compute(precision=_abi.PRISIM_FP32)
    full = ctx.get_vis()
    assert NP.all(NP.isfinite(full.view(NP.float64)))
    sel = NP.arange(0, bl.shape[0], 977)
    ref = CO.skyvis(bl[sel], ch, sky['dircos'], pb, zen)
    assert relerr(full[sel], ref, pb) <= 5e-6
    half = pb.shape[0] // 2
    acc = NP.zeros_like(full)
    for lo, hi in ((0, half), (half, pb.shape[0])):
        ctx.set_sky(sky['dircos'][lo:hi], pb[lo:hi], zen)
        ctx.compute(precision=_abi.PRISIM_FP32)
        acc += ctx.get_vis()
    assert relerr(acc, full, pb) <= 1e-5
    ctx.set_array(-bl[::61], ch)
    ctx.set_sky(sky['dircos'], pb, zen)
    ctx.compute(precision=_abi.PRISIM_FP32)
    assert relerr(ctx.get_vis(), NP.conj(full[::61]), pb) <= 1e-5


def test_full_size_config5_snapshot_properties(ctx):
    """One snapshot of BASELINE config 5 (HERA-350 x 1024 ch x nside-256 diffuse sky above the horizon = 392 704 sources, taper ON,
    fp32: 2.46e13 terms).  The sky is 24 flush intervals long and its pbflux slab is 100 MB per channel tile (25x the L2), so this
    is the size at which the periodic fp32 -> fp64 flush, the L2 warm-up and the grouped taper recurrence all work for a living.
    Checks: finite; 4 baselines against the C oracle; additivity over two disjoint source sets (slab invariance, :6348-6376)."""
    cfg = W.config5(n_acc=1)
    bl, ch, sky = cfg['baselines'], cfg['channels'], cfg['sky']
    zen = NP.array([0.0, 0.0, 1.0])
    n = sky['dircos'].shape[0]
    assert n > 390000
    ctx.set_array(bl, ch)
    ctx.set_tuning(0, 0, 0)

    def run(sel):
        ctx.set_sky_analytic(sky['dircos'][sel], sky['flux_ref'][sel], sky['spindex'][sel], sky['ref_freq'], _abi.PRISIM_BEAM_AIRY, 14.0,
                             zen, zen, fwhm_deg=sky['fwhm_deg'][sel])
        ctx.compute(precision=_abi.PRISIM_FP32)
        return ctx.get_vis()

    full = run(slice(None))
    tm = ctx.timing()
    assert tm['last_chan_tile'] == 64 and tm['last_taper_group'] == 1
    assert NP.all(NP.isfinite(full.view(NP.float64)))
    pb = ctx.get_pbflux()
    from conftest import body_class_sample
    sel_bl, lift = body_class_sample(bl, ch, sky['dircos'], zen, f32=True)      # every class of kernel body, not an even sprinkle
    assert sel_bl.size >= 16 and lift.any() and (~lift).any()
    ref = CO.skyvis(bl[sel_bl], ch, sky['dircos'], pb, zen, fwhm_deg=sky['fwhm_deg'])
    assert relerr(full[sel_bl], ref, pb) <= 5e-6
    odd = NP.arange(n) % 3 == 1
    parts = run(odd) + run(~odd)
    assert relerr(parts, full, pb) <= 1e-5


# ---------------------------------------------------------------- the binding INTEGRATION.md shows to a PRISim maintainer
def test_integration_md_stub_runs_as_written(golden_skyvis):
    """The ctypes stub of INTEGRATION.md section 1 is executed as written (only the library path is made absolute) and must reproduce the
    reference's own fp64 and memsave results on the golden inputs, gradient included."""
    import os
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with open(os.path.join(root, 'INTEGRATION.md')) as f:
        text = f.read()
    code = re.search(r'```python\n(.*?)```', text, re.S).group(1)
    assert "C.CDLL('libprisim_hip.so')" in code
    code = code.replace("C.CDLL('libprisim_hip.so')", 'C.CDLL(%r)' % _abi.LIB_PATH)
    ns = {}
    exec(compile(code, 'INTEGRATION.md#1', 'exec'), ns)
    g = golden_skyvis
    hip = ns['HipSkySum'](g['baselines'], g['channels'])
    v, gr = hip.skyvis(g['dircos'], g['pbfluxes'], g['pc_dircos'], src_fwhm_deg=_fwhm(g), gradient=True)
    assert v.dtype == NP.complex128
    assert relerr(v, g['skyvis_f64_taper'], g['pbfluxes']) <= 1e-11
    assert max(relerr(gr[k], g['grad_f64_taper'][k], g['pbfluxes']) for k in range(3)) <= 1e-11
    v32 = hip.skyvis(g['dircos'], g['pbfluxes'], g['pc_dircos'], memsave=True)
    assert v32.dtype == NP.complex64 and relerr(v32, g['skyvis_f64'], g['pbfluxes']) <= 5e-6
    with pytest.raises(ValueError):
        hip.skyvis(g['dircos'][:5] * NP.nan, g['pbfluxes'][:5], g['pc_dircos'])           # PRISIM_EINVAL -> ValueError, as the stub maps it
