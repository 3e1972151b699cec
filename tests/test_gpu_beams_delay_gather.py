"""GPU (-m gpu): fused device beams, rocFFT delay transform, single-rank all-gather, and the
InterferometerArray drop-in -- all through the C-ABI."""
import os

import numpy as NP
import pytest

from oracle import skyvis_oracle as O, beams_oracle as BO, delay_oracle as DO
from prisim_amd import _abi, workloads as W
from prisim_amd import interferometry as RI, skymodel as SM, primary_beams as PB

pytestmark = pytest.mark.gpu


def test_device_beams_match_golden_reference_functions(ctx, golden_beams):
    g = golden_beams
    sp, f = g['skypos_altaz'], g['freq_hz']
    dc = O.altaz2dircos(sp)
    zen = NP.array([0.0, 0.0, 1.0])
    ctx.set_array(NP.zeros((1, 3)), f)
    n = sp.shape[0]
    for kind, key in ((_abi.PRISIM_BEAM_AIRY, 'airy_power_d14'), (_abi.PRISIM_BEAM_GAUSSIAN, 'gauss_power_d14'),
                      (_abi.PRISIM_BEAM_DELTA, 'pbg_delta')):
        ctx.set_sky_analytic(dc, NP.ones(n), NP.zeros(n), 150e6, kind, 14.0, zen, zen)
        pb = ctx.get_pbflux()
        assert NP.max(NP.abs(pb - g[key])) <= 1e-12, key
    # host dispatcher mirror (primary_beam_generator) evaluates on the GPU as well
    pb = PB.primary_beam_generator(sp, f / 1e9, {'shape': 'dish', 'size': 14.0}, freq_scale='GHz', skyunits='altaz')
    assert NP.max(NP.abs(pb - g['pbg_dish_d14'])) <= 1e-12
    pb = PB.primary_beam_generator(sp, f / 1e9, {'id': 'hera', 'orientation': [90.0, 270.0], 'ocoords': 'altaz'},
                                   freq_scale='GHz', skyunits='altaz')
    assert NP.max(NP.abs(pb - g['airy_power_d14'])) <= 1e-12


def test_device_beam_times_power_law_and_offzenith_pointing(ctx):
    cfg = W.config1()
    ch, sky = cfg['channels'], cfg['sky']
    ctx.set_array(cfg['baselines'], ch)
    pc_altaz = NP.array([75.0, 200.0])
    bpc = O.altaz2dircos(pc_altaz[None, :])[0]
    zen = NP.array([0.0, 0.0, 1.0])
    flux = sky['flux_ref'][:, None] * (ch[None, :] / sky['ref_freq']) ** sky['spindex'][:, None]
    for kind, fn in ((_abi.PRISIM_BEAM_GAUSSIAN, BO.gaussian_beam), (_abi.PRISIM_BEAM_AIRY, BO.airy_disk_pattern)):
        ctx.set_sky_analytic(sky['dircos'], sky['flux_ref'], sky['spindex'], sky['ref_freq'], kind, 14.0, bpc, zen)
        ref = fn(14.0, sky['altaz'], ch, pointing_altaz=pc_altaz) * flux
        assert NP.max(NP.abs(ctx.get_pbflux() - ref)) <= 1e-11 * NP.max(ref)
        # tabulated spectra instead of the power law
        ctx.set_sky_analytic(sky['dircos'], None, None, None, kind, 14.0, bpc, zen, flux_spectrum=flux)
        assert NP.max(NP.abs(ctx.get_pbflux() - ref)) <= 1e-11 * NP.max(ref)


@pytest.mark.parametrize('pad', [0.0, 1.0, 0.5])
def test_delay_transform_matches_numpy_restatement(ctx, pad):
    rng = NP.random.default_rng(12)
    nbl, nchan, nt = 13, 64, 3
    ch = 150e6 + NP.arange(nchan) * 1.0e5
    ctx.set_array(rng.uniform(-100, 100, (nbl, 3)), ch, nt_max=nt)
    cube = rng.normal(size=(nt, nbl, nchan)) + 1j * rng.normal(size=(nt, nbl, nchan))
    for t in range(nt):
        ctx.set_vis(cube[t], slot=t)
    wts = rng.uniform(0.5, 1.0, (nbl, nchan))
    out, lags, pw = ctx.delay_transform(nt, bpwts=wts, pad=pad, want_power=True, power_scale=2.5)
    vis_bft = NP.transpose(cube, (1, 2, 0))                                    # (nbl, nchan, nt)
    ref, ref_lags = DO.delay_transform(vis_bft, wts[:, :, None], NP.ones((nbl, nchan, 1)), 1.0e5, pad=pad)
    ref = NP.transpose(ref, (2, 0, 1))
    assert out.shape == ref.shape
    assert NP.max(NP.abs(out - ref)) <= 1e-10 * NP.max(NP.abs(ref))            # delay spectra <= 1e-10 rel
    assert NP.allclose(lags, ref_lags, rtol=0, atol=1e-18)
    assert NP.max(NP.abs(pw - DO.delay_power(ref, 2.5))) <= 1e-9 * NP.max(NP.abs(ref)) ** 2 * 2.5


@pytest.mark.parametrize('nchan', [256, 512, 768, 1024, 2048, 4096])
def test_fused_lds_delay_fft_matches_numpy_and_rocfft_pipeline(ctx, monkeypatch, nchan):
    """Channel counts 256 R (R = 1, 2, 3, 4, 8, 16; 768 = the MWA grid) with an integer 1 + pad run ONE kernel (delay_kernels.hip: three register stages, two LDS
    exchanges): against the numpy restatement (zero padding, fftshift and decimation included), against the rocFFT pipeline on the
    same device cube, host-output and device-resident forms, with and without a window, rows not a multiple of the rows per block."""
    rng = NP.random.default_rng(nchan)
    nbl, nt, df = 37, 2, 97656.25
    ch = 150e6 + NP.arange(nchan) * df
    ctx.set_array(rng.uniform(-100, 100, (nbl, 3)), ch, nt_max=nt)
    cube = rng.normal(size=(nt, nbl, nchan)) + 1j * rng.normal(size=(nt, nbl, nchan))
    for t in range(nt):
        ctx.set_vis(cube[t], slot=t)
    wts = rng.uniform(0.2, 1.0, (nbl, nchan))
    vis_bft = NP.transpose(cube, (1, 2, 0))
    for pad, w in ((1.0, wts), (0.0, None), (2.0, wts), (1.0, wts[3])):          # windows: per baseline, none, one for all baselines
        wfull = None if w is None else NP.broadcast_to(w, (nbl, nchan))
        wr = NP.ones((nbl, nchan, 1)) if w is None else wfull[:, :, None]
        ref, ref_lags = DO.delay_transform(vis_bft, wr, NP.ones((nbl, nchan, 1)), df, pad=pad)
        ref = NP.transpose(ref, (2, 0, 1))
        out, lags, pw = ctx.delay_transform(nt, bpwts=wfull, pad=pad, want_power=True, power_scale=0.5)
        assert out.shape == ref.shape == (nt, nbl, nchan)
        assert NP.max(NP.abs(out - ref)) <= 1e-12 * NP.max(NP.abs(ref))
        assert NP.max(NP.abs(pw - DO.delay_power(ref, 0.5))) <= 1e-11 * NP.max(NP.abs(ref)) ** 2
        assert NP.allclose(lags, ref_lags, rtol=0, atol=1e-18)
        lags2, nout = ctx.delay_transform_device(nt, bpwts=w, pad=pad, want_lag=True, want_power=True, power_scale=0.5)
        assert nout == nchan and ctx.timing()['last_delay_fused'] == 1 and ctx.timing()['last_delay_ms'] > 0.0
        assert NP.array_equal(ctx.get_lags(0, nt), out) and NP.array_equal(ctx.get_delay_power(0, nt), pw)
        rows = NP.array([nbl - 1, 0, 5])
        assert NP.array_equal(ctx.get_lags(1, 1, rows=rows), out[1:2][:, rows])
        monkeypatch.setenv('PRISIM_HIP_DT_FUSED', '0')
        out_r, _, pw_r = ctx.delay_transform(nt, bpwts=wfull, pad=pad, want_power=True, power_scale=0.5)
        ctx.delay_transform_device(nt, bpwts=w, pad=pad)
        assert ctx.timing()['last_delay_fused'] == 0
        assert NP.array_equal(ctx.get_lags(0, nt), out_r)             # (the one-row window goes to the rocFFT pipeline as one row too)
        monkeypatch.delenv('PRISIM_HIP_DT_FUSED')
        assert NP.max(NP.abs(out - out_r)) <= 1e-12 * NP.max(NP.abs(ref))
    with pytest.raises(RuntimeError):
        ctx.get_delay_power(0, nt)                                    # the last device transform produced lags only
    with pytest.raises(ValueError):
        ctx.get_lags(0, nt + 1)


def test_delay_transform_in_snapshot_batches(ctx, monkeypatch):
    """Large cubes are transformed a few snapshots at a time (4 GiB work buffer); force 1-snapshot batches here."""
    rng = NP.random.default_rng(13)
    nbl, nchan, nt = 7, 32, 5
    ch = 150e6 + NP.arange(nchan) * 1.0e5
    ctx.set_array(rng.uniform(-100, 100, (nbl, 3)), ch, nt_max=nt)
    cube = rng.normal(size=(nt, nbl, nchan)) + 1j * rng.normal(size=(nt, nbl, nchan))
    for t in range(nt):
        ctx.set_vis(cube[t], slot=t)
    ref, _, refp = ctx.delay_transform(nt, pad=1.0, want_power=True)
    monkeypatch.setenv('PRISIM_HIP_DT_BATCH_BYTES', str(2 * nbl * 2 * nchan * 16))      # two snapshots per batch -> 3 batches
    out, _, pw = ctx.delay_transform(nt, pad=1.0, want_power=True)
    assert NP.array_equal(out, ref) and NP.array_equal(pw, refp)


def test_delay_transform_single_tone_kat8(ctx):
    nchan, df = 128, 97656.25
    ch = 150e6 + NP.arange(nchan) * df
    ctx.set_array(NP.zeros((1, 3)), ch)
    k0 = -9
    tau0 = k0 / (nchan * df)
    ctx.set_vis((3.0 * NP.exp(-2j * NP.pi * (ch - ch[0]) * tau0))[None, :])
    out, lags, _ = ctx.delay_transform(1, pad=1.0)
    ipk = int(NP.argmax(NP.abs(out[0, 0])))
    assert abs(lags[ipk] - tau0) <= 1e-15 and abs(abs(out[0, 0, ipk]) - 3.0 * nchan * df) <= 1e-6 * nchan * df


def test_single_rank_allgather_and_checksum(ctx):
    rng = NP.random.default_rng(3)
    nbl, nchan, nt = 17, 40, 2
    ch = 150e6 + NP.arange(nchan) * 1.0e5
    ctx.set_array(rng.uniform(-100, 100, (nbl, 3)), ch, nt_max=nt)
    cube = rng.normal(size=(nt, nbl, nchan)) + 1j * rng.normal(size=(nt, nbl, nchan))
    for t in range(nt):
        ctx.set_vis(cube[t], slot=t)
    ctx.allgather(nt)
    g = ctx.get_gathered(nt, 1)                                    # [t][rank][b][f]
    assert g.shape == (nt, 1, nbl, nchan) and NP.array_equal(g[:, 0], cube)
    assert abs(ctx.gathered_checksum(nt) - (cube.real.sum() + cube.imag.sum())) <= 1e-9
    ctx.allgather(nt, complex64=True)
    g = ctx.get_gathered(nt, 1)
    assert g.dtype == NP.complex64 and NP.array_equal(g[:, 0], cube.astype(NP.complex64))
    # per-slot gathers on the communication stream
    for t in range(nt):
        ctx.allgather_slot_async(t)
    ctx.sync()
    assert NP.array_equal(ctx.get_gathered(nt, 1)[:, 0], cube)


def test_rccl_communicator_of_one_rank(ctx):
    """RCCL itself (dlopen + ncclCommInitRank + ncclAllGather) on the one GPU of the box."""
    rng = NP.random.default_rng(4)
    ch = 150e6 + NP.arange(16) * 1.0e5
    c = _abi.Context(0)
    try:
        c.set_array(rng.uniform(-100, 100, (5, 3)), ch, nt_max=1)
        v = rng.normal(size=(5, 16)) + 1j * rng.normal(size=(5, 16))
        c.set_vis(v)
        c.comm_init(_abi.Context.comm_unique_id(), 1, 0)
        c.allgather(1)
        assert NP.array_equal(c.get_gathered(1, 1)[0, 0], v)
        c.allgather_slot_async(0, complex64=True)
        c.sync()
        assert NP.array_equal(c.get_gathered(1, 1)[0, 0], v.astype(NP.complex64))
    finally:
        c.close()


def test_rccl_selftest_gather_stats_and_gradient_gather_on_one_rank(ctx):
    """What bench.py and the driver do around RCCL at N > 1, on the one GPU a test box has: the pattern self-test through
    ncclAllGather, the overlapped per-snapshot gather on the high-priority stream with its event timing, and the gradient-cube gather."""
    rng = NP.random.default_rng(41)
    ch = 150e6 + NP.arange(32) * 1.0e5
    bl = rng.uniform(-100, 100, (70, 3))
    s = O.altaz2dircos(NP.stack((rng.uniform(20, 90, 50), rng.uniform(0, 360, 50)), 1))
    p = rng.uniform(0, 3, (50, ch.size))
    zen = NP.array([0.0, 0.0, 1.0])
    c = _abi.Context(0)
    try:
        c.set_array(bl, ch, nt_max=3)
        c.comm_selftest(1 << 16)                                  # no communicator yet: device-copy form of the same check
        c.comm_init(_abi.Context.comm_unique_id(), 1, 0)
        c.comm_selftest(1 << 20)
        c.set_sky(s, p, zen)
        c.comm_stats(reset=True)
        for t in range(3):
            c.compute(want_grad=True, slot=t)
            c.allgather_slot_async(t)
        c.sync()
        st = c.comm_stats()
        assert st['n_gathers'] == 3 and st['bytes_per_peer'] == 70 * 32 * 16 and st['nranks'] == 1
        assert 0.0 < st['last_gather_ms'] <= st['max_gather_ms'] and st['sum_gather_ms'] >= st['max_gather_ms']
        assert st['last_gather_after_compute_ms'] > 0.0
        assert st['stream_priority'] <= 0 <= st['stream_priority_lowest']     # created with the highest priority the device offers
        g = c.get_gathered(3, 1)
        for t in range(3):
            v, grad = c.get_vis(slot=t, want_grad=True)
            assert NP.array_equal(g[t, 0], v)
        c.set_gather_root(0)                                      # gather to one root (ncclSend / ncclRecv path; one rank: its own block)
        c.allgather(3)
        assert NP.array_equal(c.get_gathered(3, 1), g)
        with pytest.raises(ValueError):
            c.set_gather_root(1)                                  # not a rank of this communicator
        c.set_gather_root(None)
        c.allgather_grad(3)
        gg = c.get_gathered_grad(3, 1)
        assert gg.shape == (3, 1, 3, 70, 32)
        ref_v, ref_g = O.skyvis(bl, ch, s, p, zen, gradient=True)
        for t in range(3):
            assert NP.array_equal(gg[t, 0], c.get_vis(slot=t, want_grad=True)[1])
            assert NP.max(NP.abs(gg[t, 0] - ref_g)) <= 1e-11 * NP.sum(NP.abs(p), axis=0).max()
    finally:
        c.close()


def test_asynchronous_downloads_into_pinned_host_memory(ctx):
    """prisim_hip_get_vis_async: the copy-stream download of every slot (complex128 and device-rounded complex64, with and without the
    gradient block) equals the synchronous get_vis."""
    rng = NP.random.default_rng(43)
    ch = 150e6 + NP.arange(48) * 1.0e5
    bl = rng.uniform(-200, 200, (300, 3))
    s = O.altaz2dircos(NP.stack((rng.uniform(20, 90, 64), rng.uniform(0, 360, 64)), 1))
    zen = NP.array([0.0, 0.0, 1.0])
    c = _abi.Context(0)
    try:
        c.set_array(bl, ch, nt_max=4)
        host128 = _abi.host_empty((4, 300, 48), NP.complex128)
        host64 = _abi.host_empty((4, 300, 48), NP.complex64)
        hostg = _abi.host_empty((4, 3, 300, 48), NP.complex64)
        pageable = NP.empty((300, 48), dtype=NP.complex128)
        for t in range(4):
            c.set_sky(s, rng.uniform(0, 3, (64, ch.size)), zen)
            c.compute(want_grad=True, slot=t)
            c.get_vis_async(t, host128[t])                        # queued behind slot t's sky-sum, runs beside slot t+1's
            c.get_vis_async(t, host64[t], grad_out=hostg[t])
        c.get_vis_async(2, pageable)                              # pageable destination: staged by the runtime, same result
        c.wait_downloads()
        for t in range(4):
            v, g = c.get_vis(slot=t, want_grad=True)
            assert NP.array_equal(host128[t], v)
            assert NP.array_equal(host64[t], v.astype(NP.complex64)) and NP.array_equal(hostg[t], g.astype(NP.complex64))
        assert NP.array_equal(pageable, c.get_vis(slot=2))
        with pytest.raises(ValueError):
            c.get_vis_async(0, NP.empty((3, 3), dtype=NP.complex128))
        del host128, host64, hostg                                # pinned memory goes back (finalizers)
    finally:
        c.close()


def test_host_staging_through_the_class_matches_the_lazy_download():
    """reserve(host_staging=True): every observe() queues its snapshot's download; skyvis_freq is then a view of the pinned cube, equal to
    what the lazy path fetches -- also after a re-centring on the device, which re-queues the copies."""
    cfg = W.config2()
    bl, ch, sky = cfg['baselines'], cfg['channels'][:64], cfg['sky']
    n = sky['dircos'].shape[0]
    skymod = SM.SkyModel(location=sky['altaz'], flux_ref=sky['flux_ref'], spindex=sky['spindex'], ref_freq=sky['ref_freq'],
                         src_shape=NP.stack((sky['fwhm_deg'], sky['fwhm_deg'], NP.zeros(n)), axis=1))
    labels = ['b%d' % i for i in range(bl.shape[0])]
    cubes = []
    for staging in (True, False):
        for memsave in (False, True):
            ia = RI.InterferometerArray(labels, bl, ch, telescope={'id': 'hera', 'orientation': [90.0, 270.0], 'ocoords': 'altaz'},
                                        latitude=-30.7224, skycoords='altaz', pointing_coords='hadec')
            ia.reserve(3, host_staging=staging)
            for j in range(3):
                ia.observe((2457000.5 + j * 1e-3, 5.0 * j), {'Tnet': 100.0}, NP.ones(ch.size), [0.0, -30.7224], skymod, 10.0, memsave=memsave)
            assert all(isinstance(sn, RI._DeviceSlot) and sn.staged == staging for sn in ia._cube)
            ia.phase_centering(phase_center=NP.array([[80.0, 120.0]]), phase_center_coords='altaz', verbose=False)
            if staging:
                snaps = ia.skyvis_freq_snapshots()                # the pinned cube itself, snapshot-major, no copy
                assert snaps.shape == (3, bl.shape[0], ch.size) and NP.shares_memory(snaps, ia._host_cube)
            cube = ia.skyvis_freq
            assert cube.dtype == (NP.complex64 if memsave else NP.complex128) and cube.flags['C_CONTIGUOUS']
            if staging:
                assert NP.array_equal(NP.moveaxis(snaps, 0, 2), cube)
            cubes.append(NP.array(cube))
            ia._ctx.close()
    assert NP.array_equal(cubes[0], cubes[2]) and NP.array_equal(cubes[1], cubes[3])
    assert NP.max(NP.abs(cubes[0])) > 0


# ---------------------------------------------------------------- InterferometerArray drop-in
class FakeTime(object):
    """astropy.time.Time-like: observe() only needs .jd and .sidereal_time('apparent').deg (:6113, :6395)."""
    def __init__(self, jd, lst_deg):
        self.jd, self._lst = jd, lst_deg

    def sidereal_time(self, kind):
        class _A(object):
            pass
        a = _A()
        a.deg = self._lst
        return a


def test_interferometer_array_observe_matches_oracle():
    cfg = W.config2()
    bl, ch, sky = cfg['baselines'][::3], cfg['channels'][:64], cfg['sky']
    lat = -30.7224
    n = sky['dircos'].shape[0]
    skymod = SM.SkyModel(location=sky['altaz'], flux_ref=sky['flux_ref'], spindex=sky['spindex'], ref_freq=sky['ref_freq'],
                         src_shape=NP.stack((sky['fwhm_deg'], sky['fwhm_deg'], NP.zeros(n)), axis=1))
    labels = ['b%d' % i for i in range(bl.shape[0])]
    ia = RI.InterferometerArray(labels, bl, ch, telescope={'id': 'hera', 'orientation': [90.0, 270.0], 'ocoords': 'altaz'},
                                latitude=lat, skycoords='altaz', pointing_coords='hadec')
    tsys = {'Trx': 100.0, 'Tant': {'f0': 150e6, 'T0': 200.0, 'spindex': -2.5}, 'Tnet': None}
    for j, memsave in enumerate((False, True)):
        ia2 = ia if j == 0 else RI.InterferometerArray(labels, bl, ch, telescope=ia.telescope, latitude=lat, skycoords='altaz')
        ia2.observe(FakeTime(2457000.5 + j, 30.0), tsys, NP.ones(ch.size), [0.0, lat], skymod, 10.0, gradient_mode='baseline',
                    memsave=memsave, roi_radius=90.0)
        pb = BO.airy_disk_pattern(14.0, sky['altaz'], ch, pointing_altaz=[90.0, 270.0]) \
            * skymod.generate_spectrum(frequency=ch)
        zen = O.altaz2dircos(NP.array([[90.0, 0.0]]))[0]
        ref, gref = O.skyvis(bl, ch, sky['dircos'], pb, zen, fwhm_deg=sky['fwhm_deg'], gradient=True)
        tol = 5e-6 if memsave else 1e-11
        scale = O.abs_flux_sum(pb)[None, :]
        assert ia2.skyvis_freq.shape == (bl.shape[0], ch.size, 1)
        assert ia2.skyvis_freq.dtype == (NP.complex64 if memsave else NP.complex128)
        assert NP.max(NP.abs(ia2.skyvis_freq[:, :, 0] - ref) / scale) <= tol
        assert NP.max(NP.abs(ia2.gradient['baseline'][:, :, :, 0] - gref) / scale[None]) <= tol
        # apply_gradients (interferometry.py:6726-6819) on the cube the fused kernel produced: first-order prediction of mm-level
        # baseline errors against the oracle at the displaced baselines (phase-centre delays held, as the gradient holds them)
        db = NP.random.default_rng(5 + j).normal(scale=1e-3, size=(3, bl.shape[0]))
        dv = ia2.apply_gradients(perturbations={'baseline': db})[0, :, :, 0]
        assert NP.max(NP.abs(dv - (-2j * NP.pi * ch[None, :] / 299792458.0) * NP.einsum('kb,kbf->bf', db, gref)) / scale) <= 10 * tol
        moved = O.skyvis(bl + db.T, ch, sky['dircos'], pb, zen, fwhm_deg=sky['fwhm_deg']) \
            * NP.exp(-2j * NP.pi * ch[None, :] * db[2][:, None] / 299792458.0)
        assert NP.max(NP.abs(moved - ref - dv) / scale) <= 0.05 * NP.max(NP.abs(moved - ref) / scale) + 10 * tol
        assert ia2.n_acc == 1 and ia2.t_obs == 10.0 and ia2.lst == [30.0] and ia2.timestamp == [2457000.5 + j]
        assert ia2.bp.shape == (bl.shape[0], ch.size, 1) and ia2.Tsys.shape == (bl.shape[0], ch.size, 1)
        assert NP.asarray(ia2.geometric_delays[0]).shape == (n, bl.shape[0])
        assert NP.array_equal(ia2.obs_catalog_indices[0], NP.arange(n))
    # second snapshot appends along the last (time) axis; roi_info path with a supplied float32 beam
    pb32 = BO.airy_disk_pattern(14.0, sky['altaz'], ch).astype(NP.float32)
    ia.observe((2457000.6, 31.0), tsys, NP.ones(ch.size), [0.0, lat], skymod, 10.0,
               roi_info={'ind': NP.arange(n), 'pbeam': pb32})
    assert ia.skyvis_freq.shape == (bl.shape[0], ch.size, 2) and ia.n_acc == 2
    ref2 = O.skyvis(bl, ch, sky['dircos'], pb32.astype(NP.float64) * skymod.generate_spectrum(frequency=ch), zen,
                    fwhm_deg=sky['fwhm_deg'])
    assert NP.max(NP.abs(ia.skyvis_freq[:, :, 1] - ref2) / O.abs_flux_sum(pb)[None, :]) <= 1e-11
    # delay transform of the accumulated cube on the GPU
    ia.delay_transform(pad=1.0, verbose=False)
    ref_lag, ref_lags = DO.delay_transform(ia.skyvis_freq, ia.bp, ia.bp_wts, ia.freq_resolution, pad=1.0)
    assert ia.skyvis_lag.shape == ref_lag.shape
    assert NP.max(NP.abs(ia.skyvis_lag - ref_lag)) <= 1e-10 * NP.max(NP.abs(ref_lag))
    assert NP.allclose(ia.lags, ref_lags)


def test_interferometer_array_validation_and_empty_sky():
    ch = 150e6 + NP.arange(8) * 1e5
    with pytest.raises(ValueError):
        RI.InterferometerArray(['a'], NP.zeros((2, 3)), ch)
    with pytest.raises(TypeError):
        RI.InterferometerArray('a', NP.zeros((1, 3)), ch)
    with pytest.raises(ValueError):
        RI.InterferometerArray(['a'], NP.zeros((1, 3)), ch, freq_scale='THz')
    with pytest.raises(ValueError):
        RI.InterferometerArray(['a'], NP.zeros((1, 3)), ch, skycoords='galactic')
    ia = RI.InterferometerArray(['a'], NP.array([[10.0, 0.0]]), ch / 1e6, freq_scale='MHz', skycoords='altaz',
                                telescope={'shape': 'delta'})
    assert ia.baselines.shape == (1, 3) and NP.allclose(ia.channels, ch)
    skymod = SM.SkyModel(location=[[-10.0, 0.0]], flux_ref=[1.0], spindex=[0.0], ref_freq=150e6)   # below the horizon
    with pytest.warns(UserWarning):
        ia.observe((2457000.5, 0.0), {'Tnet': 50.0}, NP.ones(8), [0.0, 34.079], skymod, 1.0)
    assert NP.all(ia.skyvis_freq == 0) and ia.skyvis_freq.shape == (1, 8, 1)
    with pytest.raises(ValueError):
        ia.observe((2457000.5, 0.0), {'Tnet': 50.0}, NP.ones(7), [0.0, 34.079], skymod, 1.0)
    with pytest.raises(TypeError):
        ia.observe((2457000.5, 0.0), 50.0, NP.ones(8), [0.0, 34.079], skymod, 1.0)
    with pytest.raises(KeyError):
        ia.observe((2457000.5, 0.0), {'Tnet': 50.0}, NP.ones(8), [0.0, 34.079], skymod, 1.0, roi_info={'ind': None})


def test_observing_run_drift():
    ch = 150e6 + NP.arange(16) * 1e5
    bl = NP.array([[14.6, 0.0, 0.0], [0.0, 14.6, 0.0]])
    lat = -30.7224
    skymod = SM.SkyModel(location=[[10.0, lat], [355.0, lat + 5.0]], flux_ref=[1.0, 2.0], spindex=[0.0, -0.7], ref_freq=150e6)
    ia = RI.InterferometerArray(['a', 'b'], bl, ch, telescope={'shape': 'gaussian', 'size': 14.0}, latitude=lat, skycoords='radec')
    ia.frame_model = 'date'        # the catalogue is taken to be in the equinox of date: HA = LST - RA, what the oracle lines below write out
    ia.observing_run([0.0, lat], skymod, 60.0, 180.0, ch, NP.ones(16), 100.0, 0.5, mode='drift', verbose=False)
    assert ia.skyvis_freq.shape == (2, 16, 3) and ia.n_acc == 3 and ia.t_obs == 180.0
    assert NP.allclose(ia.lst, (0.5 + 60.0 / 3600 * NP.arange(3)) * 15.0)
    # snapshot 1 against the oracle: HA = LST - RA, zenith phase centre
    lst1 = ia.lst[1]
    hadec = NP.stack((lst1 - skymod.location[:, 0], skymod.location[:, 1]), 1)
    altaz = O.hadec2altaz(hadec, lat)
    pb = BO.gaussian_beam(14.0, altaz, ch, pointing_altaz=O.hadec2altaz([[0.0, lat]], lat)[0]) * skymod.generate_spectrum(frequency=ch)
    ref = O.skyvis(bl, ch, O.altaz2dircos(altaz), pb, O.altaz2dircos(O.hadec2altaz([[0.0, lat]], lat))[0])
    assert NP.max(NP.abs(ia.skyvis_freq[:, :, 1] - ref)) <= 1e-10


def test_device_dipole_array_ground_beams_match_reference_golden(ctx):
    import os
    from conftest import GOLDEN
    g = dict(NP.load(os.path.join(GOLDEN, 'golden_beams_ext.npz')))
    dc, f = g['dircos'], g['freq_hz']
    n = dc.shape[0]
    # through the host dispatcher mirror, evaluated on the GPU
    for tel, key in (({'id': 'mwa'}, 'pbg_mwa'), ({'id': 'mwa_dipole'}, 'pbg_mwa_dipole'), ({'id': 'paper'}, 'pbg_paper'),
                     ({'shape': 'dipole', 'size': 1.5, 'ocoords': 'dircos', 'orientation': g['tilt']}, 'pbg_shape_dipole')):
        pb = PB.primary_beam_generator(dc, f / 1e9, tel, freq_scale='GHz', skyunits='dircos', east2ax1=0.0)
        assert NP.max(NP.abs(pb - g[key])) <= 1e-12, key
    # field-level pieces through the C-ABI: rotated + pointed array factor, dipole approximations, ground plane
    zen = NP.array([0.0, 0.0, 1.0])
    ctx.set_array(NP.zeros((1, 3)), f)
    one, zero = NP.ones(n), NP.zeros(n)
    ctx.set_sky_analytic(dc, one, zero, 150e6, _abi.PRISIM_BEAM_DELTA, 0.0, zen, zen,
                         ext={'array': {'nax1': 4, 'nax2': 4, 'sep1': 1.1, 'sep2': 1.1, 'east2ax1': 30.0, 'pointing_dircos': g['array_pc']}})
    assert NP.max(NP.abs(ctx.get_pbflux() - g['irap_4x4_rot30_pointed'] ** 2)) <= 1e-12
    for mode, key in ((_abi.PRISIM_DIPOLE_SHORT, 'dipole_field_short'), (_abi.PRISIM_DIPOLE_HALFWAVE, 'dipole_field_halfwave'),
                      (_abi.PRISIM_DIPOLE_GENERAL, 'dipole_field_general')):
        ctx.set_sky_analytic(dc, one, zero, 150e6, _abi.PRISIM_BEAM_DIPOLE, 0.74, zen, zen, ext={'dipole_dircos': [1, 0, 0], 'dipole_mode': mode})
        assert NP.max(NP.abs(ctx.get_pbflux() - g[key] ** 2)) <= 1e-12, key
    mod = {'scale': 0.8, 'max': 2.5}
    ctx.set_sky_analytic(dc, one, zero, 150e6, _abi.PRISIM_BEAM_GAUSSIAN, 14.0, zen, zen, ext={'ground': {'height': 0.3, 'modifier': mod}})
    ref = BO.composite_power_beam(dc, f, element='gaussian', size=14.0, ground={'height': 0.3, 'modifier': mod})
    assert NP.max(NP.abs(ctx.get_pbflux() - ref)) <= 1e-12
    # every element pattern has a kernel instance of its own with the extras (k_beam_flux<KIND>) and, for delta / Gaussian / Airy, one
    # without (k_beam_flux_plain<KIND>): the Airy element under an array factor and a ground plane, and the three plain ones
    arr = {'nax1': 2, 'nax2': 3, 'sep1': 0.9, 'sep2': 1.3, 'east2ax1': 20.0, 'pointing_dircos': g['array_pc']}
    ctx.set_sky_analytic(dc, one, zero, 150e6, _abi.PRISIM_BEAM_AIRY, 6.0, zen, zen, ext={'array': arr, 'ground': {'height': 0.4}})
    ref = BO.composite_power_beam(dc, f, element='dish', size=6.0, array=arr, ground={'height': 0.4})
    assert NP.max(NP.abs(ctx.get_pbflux() - ref)) <= 1e-12
    for kind, element, size in ((_abi.PRISIM_BEAM_DELTA, 'delta', 0.0), (_abi.PRISIM_BEAM_GAUSSIAN, 'gaussian', 6.0), (_abi.PRISIM_BEAM_AIRY, 'dish', 6.0)):
        ctx.set_sky_analytic(dc, one, zero, 150e6, kind, size, zen, zen)
        assert NP.max(NP.abs(ctx.get_pbflux() - BO.composite_power_beam(dc, f, element=element, size=size))) <= 1e-12, element
    with pytest.raises(ValueError):
        ctx.set_sky_analytic(dc, one, zero, 150e6, _abi.PRISIM_BEAM_DIPOLE, 0.74, zen, zen)       # dipole without ext
    with pytest.raises(ValueError):
        ctx.set_sky_analytic(dc, one, zero, 150e6, _abi.PRISIM_BEAM_DELTA, 0.0, zen, zen, ext={'array': {'nax1': 4, 'nax2': 0, 'sep1': 1.1, 'sep2': 1.1}})


def _synthetic_external_beam(nside, bf):
    from prisim_amd import geometry as GEOM
    th, ph = GEOM.healpix_pix2ang_ring(nside)
    rng = NP.random.default_rng(40)
    base = NP.cos(NP.clip(th, 0, NP.pi / 2)) ** 2 * (1 + 0.2 * NP.cos(2 * ph)) + 1e-4
    return base[:, None] * (1 + 0.3 * (bf[None, :] / 150e6 - 1)) * NP.exp(0.05 * rng.normal(size=(th.size, bf.size)))


@pytest.mark.parametrize('kind,chromatic', [('cubic', True), ('linear', True), ('cubic', False)])
def test_device_external_healpix_beam_matches_restatement(ctx, kind, chromatic):
    from oracle import healpix_oracle as H
    cfg = W.config2()
    ch, sky = cfg['channels'][::4], cfg['sky']
    bf = NP.linspace(100e6, 200e6, 21)
    beam = _synthetic_external_beam(8, bf)
    theta = NP.pi / 2 - NP.radians(sky['altaz'][:, 0])
    phi = NP.radians(sky['altaz'][:, 1])
    flux = sky['flux_ref'][:, None] * (ch[None, :] / sky['ref_freq']) ** sky['spindex'][:, None]
    ctx.set_array(cfg['baselines'], ch)
    ctx.set_external_beam(beam, PB.spectral_interp_matrix(bf, ch, kind=kind, chromatic=chromatic, select_freq=152e6))
    zen = NP.array([0.0, 0.0, 1.0])
    ctx.set_sky_external(sky['dircos'], flux, zen, fwhm_deg=sky['fwhm_deg'])
    pb_ref = H.external_beam(beam, bf, theta, phi, ch, kind=kind, chromatic=chromatic, select_freq=152e6)
    got = ctx.get_pbflux()
    # float32 quantisation of the beam (interferometry.py:4466): values within one float32 ulp of the restatement
    assert NP.max(NP.abs(got - pb_ref * flux) / (pb_ref * flux + 1e-300)) <= 1.3e-7
    ctx.compute()
    ref = O.skyvis(cfg['baselines'], ch, sky['dircos'], got, zen, fwhm_deg=sky['fwhm_deg'])
    assert NP.max(NP.abs(ctx.get_vis() - ref) / O.abs_flux_sum(got)[None, :]) <= 1e-11
    with pytest.raises(ValueError):
        ctx.set_external_beam(-beam, NP.zeros((ch.size, bf.size)))
    with pytest.raises(ValueError):
        ctx.set_external_beam(beam[:100], NP.zeros((ch.size, bf.size)))


def test_interferometer_array_with_external_beam():
    from oracle import healpix_oracle as H
    cfg = W.config2()
    bl, ch, sky = cfg['baselines'][::5], cfg['channels'][:32], cfg['sky']
    n = sky['dircos'].shape[0]
    bf = NP.linspace(120e6, 180e6, 13)
    beam = _synthetic_external_beam(8, bf)
    skymod = SM.SkyModel(location=sky['altaz'], flux_ref=sky['flux_ref'], spindex=sky['spindex'], ref_freq=sky['ref_freq'])
    ia = RI.InterferometerArray(['b%d' % i for i in range(bl.shape[0])], bl, ch, telescope={'id': 'mwa'}, latitude=-26.7, skycoords='altaz')
    ia.reserve(2)
    ia.set_external_beam(beam, bf, spec_interp='cubic')
    for j in range(2):
        ia.observe((2457000.5 + j, 10.0 * j), {'Tnet': 100.0}, NP.ones(ch.size), [0.0, -26.7], skymod, 112.0)
    pb = H.external_beam(beam, bf, NP.pi / 2 - NP.radians(sky['altaz'][:, 0]), NP.radians(sky['altaz'][:, 1]), ch) * skymod.generate_spectrum(frequency=ch)
    ref = O.skyvis(bl, ch, sky['dircos'], pb, NP.array([0.0, 0.0, 1.0]))
    scale = O.abs_flux_sum(pb)[None, :]
    assert ia.skyvis_freq.shape == (bl.shape[0], 32, 2)
    # the float32-rounded beam differs from the double restatement by <= 6e-8 relative per source
    assert NP.max(NP.abs(ia.skyvis_freq[:, :, 0] - ref) / scale) <= 2e-7
    assert NP.array_equal(ia.skyvis_freq[:, :, 0], ia.skyvis_freq[:, :, 1])          # sky fixed in the local frame (skycoords='altaz')


def test_phase_rotate_matches_reference_statements(ctx, golden_skyvis):
    g = golden_skyvis
    cube = g['phase_cube_in']                           # (nbl, nchan, nt)
    nt = cube.shape[2]
    ctx.set_array(g['baselines'], g['channels'], nt_max=nt)
    for t in range(nt):
        ctx.set_vis(cube[:, :, t], slot=t)
    ctx.phase_rotate(nt, g['phase_pc_cur'] - g['phase_pc_new'])
    out = NP.stack([ctx.get_vis(slot=t) for t in range(nt)], axis=2)
    assert NP.max(NP.abs(out - g['phase_cube_out'])) <= 1e-11 * NP.max(NP.abs(cube))
    with pytest.raises(ValueError):
        ctx.phase_rotate(nt + 1, NP.zeros((nt + 1, 3)))


def test_interferometer_array_phase_centering_project_conjugate():
    ch = 150e6 + NP.arange(16) * 1e5
    bl = NP.array([[14.6, 0.0, 0.0], [0.0, 29.2, 0.0], [-50.0, 20.0, 1.0]])
    lat = -30.7224
    skymod = SM.SkyModel(location=[[80.0, 100.0], [60.0, 250.0]], flux_ref=[1.0, 2.0], spindex=[0.0, -0.7], ref_freq=150e6)
    ia = RI.InterferometerArray([('a', 'b'), ('a', 'c'), ('b', 'c')], bl, ch, telescope={'shape': 'delta'}, latitude=lat, skycoords='altaz',
                                pointing_coords='altaz')
    ia.reserve(2)
    for j in range(2):
        ia.observe((2457000.5 + j, 15.0 * j), {'Tnet': 100.0}, NP.ones(16), [90.0, 270.0], skymod, 10.0)
    before = NP.array(ia.skyvis_freq)
    # re-centre on source 0: its contribution becomes a real constant (flux) on every baseline / channel
    ia.phase_centering(phase_center=NP.array([[80.0, 100.0]]), phase_center_coords='altaz')
    dc = O.altaz2dircos(skymod.location)
    pb = skymod.generate_spectrum(frequency=ch)
    ref = O.skyvis(bl, ch, dc, pb, dc[0])
    assert NP.max(NP.abs(ia.skyvis_freq[:, :, 0] - ref)) <= 1e-10
    assert NP.allclose(ia.phase_center, [[80.0, 100.0]] * 2) and ia.phase_center_coords == 'altaz'
    assert NP.array_equal(ia._ctx.get_vis(slot=1), ia.skyvis_freq[:, :, 1])            # device cube follows
    # a cube that never left the device is rotated where it is (one kernel over all snapshots), nothing is downloaded
    ib = RI.InterferometerArray([('a', 'b'), ('a', 'c'), ('b', 'c')], bl, ch, telescope={'shape': 'delta'}, latitude=lat, skycoords='altaz',
                                pointing_coords='altaz')
    ib.reserve(2)
    for j in range(2):
        ib.observe((2457000.5 + j, 15.0 * j), {'Tnet': 100.0}, NP.ones(16), [90.0, 270.0], skymod, 10.0)
    ib.phase_centering(phase_center=NP.array([[80.0, 100.0]]), phase_center_coords='altaz')
    assert all(isinstance(sn, RI._DeviceSlot) for sn in ib._cube)
    assert NP.max(NP.abs(ib.skyvis_freq - ia.skyvis_freq)) <= 1e-12
    # rotating back restores the original
    ia.phase_centering(phase_center=NP.array([[90.0, 270.0]]), phase_center_coords='altaz')
    assert NP.max(NP.abs(ia.skyvis_freq - before)) <= 1e-10
    # projected baselines toward the zenith: w = Up component, |uvw| = |b|
    ia.project_baselines({'location': NP.array([90.0, 270.0]), 'coords': 'altaz'})
    assert ia.projected_baselines.shape == (3, 3, 2)
    assert NP.allclose(NP.linalg.norm(ia.projected_baselines[:, :, 0], axis=1), NP.linalg.norm(bl, axis=1))
    assert NP.allclose(ia.projected_baselines[:, 2, 0], bl[:, 2], atol=1e-9)
    ia.conjugate(ind=[2])
    assert NP.allclose(ia.baselines[2], [50.0, -20.0, -1.0]) and NP.allclose(ia.skyvis_freq[2], before[2].conj())
    assert ia.labels[2] == ('c', 'b')
    with pytest.raises(IndexError):
        ia.conjugate(ind=[7])
    with pytest.raises(KeyError):
        ia.rotate_visibilities({'location': NP.array([90.0, 270.0])})


def test_observe_radec_sky_and_roi_selection():
    """skycoords='radec' (HA = LST - RA), pointing_coords='radec', roi_center='pointing_center' with a finite radius."""
    rng = NP.random.default_rng(21)
    lat, lst = -30.7224, 47.0
    nsrc = 400
    ra = rng.uniform(0, 360, nsrc)
    dec = NP.degrees(NP.arcsin(rng.uniform(-1, 0.6, nsrc)))
    skymod = SM.SkyModel(location=NP.stack((ra, dec), 1), flux_ref=rng.uniform(1, 5, nsrc), spindex=rng.uniform(-1, 0, nsrc), ref_freq=150e6)
    ch = 150e6 + (NP.arange(24) - 12) * 2e5
    bl = rng.uniform(-100, 100, (11, 3)); bl[:, 2] = 0
    ia = RI.InterferometerArray(list(range(11)), bl, ch, telescope={'shape': 'dish', 'size': 14.0}, latitude=lat, skycoords='radec',
                                pointing_coords='radec')
    ia.frame_model = 'date'        # (catalogue in the equinox of date; the apparent-place frame is tests/test_gpu_catalog.py's)
    pc_radec = [lst - 20.0, lat + 10.0]                    # HA = 20 deg, Dec = lat + 10
    ia.observe((2457000.5, lst), {'Tnet': 100.0}, NP.ones(24), pc_radec, skymod, 10.0, roi_radius=40.0, roi_center='pointing_center')
    altaz = O.hadec2altaz(NP.stack((lst - ra, dec), 1), lat)
    dc = O.altaz2dircos(altaz)
    pc_altaz = O.hadec2altaz([[20.0, lat + 10.0]], lat)[0]
    pc_dc = O.altaz2dircos(pc_altaz[None, :])[0]
    sel = NP.where(NP.degrees(NP.arccos(NP.clip(dc @ pc_dc, -1, 1))) <= 40.0)[0]
    assert NP.array_equal(ia.obs_catalog_indices[0], sel) and 0 < sel.size < nsrc
    pb = BO.airy_disk_pattern(14.0, altaz[sel], ch, pointing_altaz=pc_altaz) * skymod.generate_spectrum(ind=sel, frequency=ch)
    ref = O.skyvis(bl, ch, dc[sel], pb, pc_dc)
    assert NP.max(NP.abs(ia.skyvis_freq[:, :, 0] - ref) / O.abs_flux_sum(pb)[None, :]) <= 1e-11
    # default region of interest: the visible hemisphere (alt >= 0)
    ia2 = RI.InterferometerArray(list(range(11)), bl, ch, telescope={'shape': 'delta'}, latitude=lat, skycoords='radec', pointing_coords='radec')
    ia2.frame_model = 'date'
    ia2.observe((2457000.5, lst), {'Tnet': 100.0}, NP.ones(24), pc_radec, skymod, 10.0)
    up = NP.where(altaz[:, 0] >= 0.0)[0]
    assert NP.array_equal(ia2.obs_catalog_indices[0], up)
    ref2 = O.skyvis(bl, ch, dc[up], skymod.generate_spectrum(ind=up, frequency=ch), pc_dc)
    assert NP.max(NP.abs(ia2.skyvis_freq[:, :, 0] - ref2)) <= 1e-10


def test_device_noise_statistics_determinism_and_sharding(ctx):
    rng = NP.random.default_rng(5)
    nbl, nchan, nt = 96, 128, 4
    ch = 150e6 + NP.arange(nchan) * 1e5
    bl = rng.uniform(-100, 100, (nbl, 3))
    ctx.set_array(bl, ch, nt_max=1)
    rms = rng.uniform(0.5, 2.0, (nt, nbl, nchan))
    a = ctx.noise(rms, seed=1234)
    assert a.shape == (nt, nbl, nchan) and NP.array_equal(a, ctx.noise(rms, seed=1234))        # deterministic for a seed
    assert not NP.array_equal(a, ctx.noise(rms, seed=1235))
    z = a / rms                                                   # unit-variance complex normals: var(re) = var(im) = 1/2
    n = z.size
    assert abs(z.real.mean()) < 4 / NP.sqrt(2 * n) and abs(z.imag.mean()) < 4 / NP.sqrt(2 * n)
    assert abs(z.real.var() - 0.5) < 0.02 and abs(z.imag.var() - 0.5) < 0.02
    assert abs(NP.mean(z.real * z.imag)) < 4 / NP.sqrt(n) / 2                                   # re, im uncorrelated
    assert abs(NP.mean(z[:, :, 1:].real * z[:, :, :-1].real)) < 0.01                            # adjacent channels uncorrelated
    assert abs(NP.mean(NP.abs(z) ** 4) - 2.0) < 0.1                                             # Gaussian kurtosis: E|z|^4 = 2 for CN(0,1)
    # a baseline-sharded run draws exactly the numbers of the unsharded run
    ctx.set_array(bl[40:], ch, nt_max=1)
    assert NP.array_equal(ctx.noise(rms[:, 40:], seed=1234, bl_offset=40), a[:, 40:])
    with pytest.raises(ValueError):
        ctx.noise(-rms[:, 40:], seed=1)


def test_interferometer_array_generate_and_add_noise():
    ch = 150e6 + NP.arange(32) * 1e5
    bl = NP.array([[14.6, 0.0, 0.0], [0.0, 29.2, 0.0]])
    skymod = SM.SkyModel(location=[[80.0, 100.0]], flux_ref=[1.0], spindex=[0.0], ref_freq=150e6)
    ia = RI.InterferometerArray(['a', 'b'], bl, ch, telescope={'shape': 'delta'}, skycoords='altaz', pointing_coords='altaz', A_eff=154.0, eff_Q=0.96)
    for j in range(3):
        ia.observe((2457000.5 + j, 0.0), {'Tnet': 300.0}, NP.ones(32), [90.0, 270.0], skymod, 60.0)
    ia.generate_noise(seed=99)
    rms = 2.0 * 1.380649e-23 / NP.sqrt(60.0 * 1e5) * 300.0 / 154.0 / 0.96 / 1e-26               # interferometry.py:6685
    assert ia.vis_rms_freq.shape[:2] == (2, 32) and NP.allclose(ia.vis_rms_freq, rms)
    assert ia.vis_noise_freq.shape == (2, 32, 3)
    assert abs(NP.std(ia.vis_noise_freq.real) / (rms / NP.sqrt(2)) - 1.0) < 0.2
    with pytest.warns(UserWarning):
        ia.add_noise()
    assert NP.array_equal(ia.vis_freq, ia.skyvis_freq + ia.vis_noise_freq)
    first = ia.vis_noise_freq.copy()
    ia.generate_noise(seed=99)
    assert NP.array_equal(first, ia.vis_noise_freq)
    ia.delay_transform(pad=0.0, verbose=False)                     # all three cubes are transformed now (Q20)
    assert ia.vis_lag.shape == ia.skyvis_lag.shape == ia.vis_noise_lag.shape == (2, 32, 3)


def test_duplicate_measurements_expands_redundant_groups():
    """interferometry.py:6823-6906: unique baselines are simulated, redundant ones re-created by repetition, noise regenerated."""
    ch = 150e6 + NP.arange(16) * 1e5
    bl = NP.array([[14.6, 0.0, 0.0], [0.0, 29.2, 0.0], [7.3, 12.644, 0.0]])
    labels = [('a1', 'a0'), ('a2', 'a0'), ('a3', 'a0')]
    groups = {('a1', 'a0'): [('a1', 'a0'), ('a5', 'a4'), ('a7', 'a6')], ('a3', 'a0'): [('a9', 'a8')]}      # key itself is added if missing
    rev = {m: k for k, v in groups.items() for m in v}
    rev[('a2', 'a0')] = ('a2', 'a0'); rev[('a3', 'a0')] = ('a3', 'a0')
    skymod = SM.SkyModel(location=[[80.0, 100.0], [60.0, 200.0]], flux_ref=[1.0, 2.0], spindex=[0.0, -0.7], ref_freq=150e6)
    ia = RI.InterferometerArray(labels, bl, ch, telescope={'shape': 'delta'}, skycoords='altaz', pointing_coords='altaz', A_eff=154.0,
                                eff_Q=0.96, blgroupinfo={'groups': groups, 'reversemap': rev})
    for j in range(2):
        ia.observe((2457000.5 + j, 0.0), {'Tnet': 300.0}, NP.ones(16), [90.0, 270.0], skymod, 60.0)
    before = ia.skyvis_freq.copy()
    with pytest.warns(UserWarning):
        ia.duplicate_measurements()
    assert ia.labels == [('a1', 'a0'), ('a5', 'a4'), ('a7', 'a6'), ('a2', 'a0'), ('a3', 'a0'), ('a9', 'a8')]
    assert ia.skyvis_freq.shape == (6, 16, 2) and ia.baselines.shape == (6, 3) and ia.baseline_lengths.shape == (6,)
    assert NP.array_equal(ia.skyvis_freq, before[[0, 0, 0, 1, 2, 2]]) and NP.array_equal(ia.baselines, bl[[0, 0, 0, 1, 2, 2]])
    assert ia.Tsys.shape[0] == 6 and ia.vis_noise_freq.shape == (6, 16, 2) and ia.vis_freq.shape == (6, 16, 2)
    assert not NP.array_equal(ia.vis_noise_freq[0], ia.vis_noise_freq[1])            # redundant baselines get independent noise
    ia.delay_transform(pad=0.0, verbose=False)                                        # device state follows the expanded list
    assert ia.skyvis_lag.shape == (6, 16, 2) and NP.allclose(ia.skyvis_lag[0], ia.skyvis_lag[2])
    with pytest.raises(TypeError):
        ia.duplicate_measurements(blgroups=[1, 2])


def test_device_beamformer_matches_reference_golden_and_oracle(ctx):
    """Phased-array beamformer on the device (primary_beams.py:1482-1754 via :288-317 and :385-416): fp64 on the GPU against the
    fp64 form of the oracle (tight) and against the reference's own complex64 output (float32-level)."""
    import os
    from conftest import GOLDEN
    g = dict(NP.load(os.path.join(GOLDEN, 'golden_beamformer.npz')))
    dc, f, tile, irr = g['dircos'], g['freq_hz'], g['tile'], g['irregular']
    n = dc.shape[0]
    zen = NP.array([0.0, 0.0, 1.0])
    one, zero = NP.ones(n), NP.zeros(n)
    ctx.set_array(NP.zeros((1, 3)), f)
    # field level through the C-ABI: delta element x beamformer -> |F|^2, one realisation and several
    NP.random.seed(5)
    cases = [({'delays': g['delays'], 'gains': g['gains']}, tile, 'field_delays_gains'),
             (None, irr, 'field_irregular_none'),
             ({'pointing_center': g['pc'], 'pointing_coords': 'dircos', 'delayerr': 0.3e-9, 'gainerr': 0.5, 'nrand': 3}, tile, 'field_jitter_seed5')]
    for info, locs, key in cases:
        d, gn = PB.beamformer_settings(locs, info)
        ctx.set_sky_analytic(dc, one, zero, 150e6, _abi.PRISIM_BEAM_DELTA, 0.0, zen, zen, ext={'beamformer': {'positions': locs, 'delays': d, 'gains': gn}})
        got = ctx.get_pbflux()
        ref64 = NP.mean(NP.abs(BO.array_field_pattern(locs, dc, 299792458.0 / f, d, gn, single=False)) ** 2, axis=2)
        assert NP.max(NP.abs(got - ref64)) <= 1e-12, key
        assert NP.max(NP.abs(got - NP.mean(NP.abs(g[key]) ** 2, axis=2))) <= 2e-6, key
    # through the host dispatcher mirror: MWA tile with explicit delays, seeded jitter, custom dipole elements with a pointing centre
    pb = PB.primary_beam_generator(dc, f / 1e9, {'id': 'mwa'}, freq_scale='GHz', skyunits='dircos', pointing_info={'delays': g['delays'], 'gains': g['gains']})
    assert NP.max(NP.abs(pb - g['pbg_mwa_delays'])) <= 1e-6
    NP.random.seed(6)
    pb = PB.primary_beam_generator(dc, f / 1e9, {'id': 'mwa'}, freq_scale='GHz', skyunits='dircos',
                                   pointing_info={'pointing_center': g['pc'], 'pointing_coords': 'dircos', 'delayerr': 0.2e-9, 'gainerr': 0.3, 'nrand': 4})
    assert NP.max(NP.abs(pb - g['pbg_mwa_jitter_seed6'])) <= 1e-6
    tel = {'shape': 'dipole', 'size': 1.5, 'ocoords': 'dircos', 'orientation': g['tilt'], 'element_locs': irr}
    pb = PB.primary_beam_generator(dc, f / 1e9, tel, freq_scale='GHz', skyunits='dircos', pointing_info={'pointing_center': g['pc'], 'pointing_coords': 'dircos'})
    assert NP.max(NP.abs(pb - g['pbg_dipole_elements_pointed'])) <= 1e-6
    # zero delays on the regular tile = the analytic 4x4 array factor pointed at the zenith
    pb_bf = PB.primary_beam_generator(dc, f / 1e9, {'id': 'mwa'}, freq_scale='GHz', skyunits='dircos', pointing_info={'delays': NP.zeros(16)})
    pb_af = PB.primary_beam_generator(dc, f / 1e9, {'id': 'mwa'}, freq_scale='GHz', skyunits='dircos')
    assert NP.max(NP.abs(pb_bf - pb_af)) <= 1e-12
    # argument checks at the ABI
    with pytest.raises(ValueError):
        ctx.set_sky_analytic(dc, one, zero, 150e6, _abi.PRISIM_BEAM_DELTA, 0.0, zen, zen,
                             ext={'beamformer': {'positions': tile, 'delays': NP.full(16, NP.nan)}})
    with pytest.raises(ValueError):
        ctx.set_sky_analytic(dc, one, zero, 150e6, _abi.PRISIM_BEAM_DELTA, 0.0, zen, zen,
                             ext={'array': {'nax1': 4, 'nax2': 4, 'sep1': 1.1, 'sep2': 1.1}, 'beamformer': {'positions': tile}})


def test_observe_with_mwa_beamformer_pointing():
    """observe(pb_info=...) with the MWA tile beamformer steered off zenith: the visibilities equal the oracle's sky-sum of
    (dipole x beamformed tile) beam x flux (interferometry.py:6252-6254)."""
    ch = 170e6 + NP.arange(24) * 1.28e6
    bl = NP.array([[30.0, 5.0, 0.0], [-120.0, 80.0, 0.5], [400.0, -310.0, 1.0]])
    rng = NP.random.default_rng(12)
    alt = NP.degrees(NP.arcsin(rng.uniform(0.2, 1.0, 40)))
    altaz = NP.stack((alt, rng.uniform(0, 360, 40)), axis=1)
    skymod = SM.SkyModel(location=altaz, flux_ref=rng.uniform(1, 5, 40), spindex=NP.full(40, -0.8), ref_freq=185e6)
    pinfo = {'pointing_center': NP.array([75.0, 150.0]), 'pointing_coords': 'altaz'}
    ia = RI.InterferometerArray(['a', 'b', 'c'], bl, ch, telescope={'id': 'mwa'}, skycoords='altaz', pointing_coords='altaz')
    ia.observe((2457000.5, 0.0), {'Tnet': 200.0}, NP.ones(24), [90.0, 270.0], skymod, 112.0, pb_info=pinfo)
    dc = O.altaz2dircos(altaz)
    tile = PB.mwa_tile_element_locs()
    d, gn = BO.beamformer_settings(tile, {'pointing_center': O.altaz2dircos(pinfo['pointing_center'].reshape(1, 2))[0], 'pointing_coords': 'dircos'})
    pb = BO.composite_power_beam(dc, ch, element='dipole', size=0.74, element_dircos=(1, 0, 0), beamformer={'positions': tile, 'delays': d, 'gains': gn})
    pbflux = pb * skymod.generate_spectrum(frequency=ch)
    ref = O.skyvis(bl, ch, dc, pbflux, NP.array([0.0, 0.0, 1.0]))
    assert NP.max(NP.abs(ia.skyvis_freq[:, :, 0] - ref) / O.abs_flux_sum(pbflux)[None, :]) <= 1e-11
    # the steered tile is brighter towards the pointing centre than the zenith-pointed one
    ia0 = RI.InterferometerArray(['a', 'b', 'c'], bl, ch, telescope={'id': 'mwa'}, skycoords='altaz', pointing_coords='altaz')
    ia0.observe((2457000.5, 0.0), {'Tnet': 200.0}, NP.ones(24), [90.0, 270.0], skymod, 112.0)
    assert not NP.allclose(ia0.skyvis_freq, ia.skyvis_freq)


def test_device_polynomial_dish_beams_match_reference_golden():
    """id 'vla' / 'gmrt' / 'ugmrt' on the device (PRISIM_BEAM_POLY) against the reference functions' output, and the reference's
    ValueError when the polynomial leaves its range of validity (:510-512) or has no coefficients (:802-803)."""
    import os
    from conftest import GOLDEN
    g = dict(NP.load(os.path.join(GOLDEN, 'golden_polybeams.npz')))
    for name, tid in (('vla_L', 'vla'), ('vla_P', 'vla'), ('gmrt_610', 'gmrt'), ('ugmrt_325', 'ugmrt')):
        pb = PB.primary_beam_generator(g['altaz_' + name], g['freq_' + name] / 1e9, {'id': tid}, freq_scale='GHz', skyunits='altaz')
        assert NP.max(NP.abs(pb - g['pbg_' + name])) <= 1e-11, name       # the angle goes alt -> direction cosines -> angle
    far = NP.array([[90.0, 0.0], [87.0, 10.0]])                    # 3 deg off axis at 610 MHz: the quartic term has taken over (1.6e4)
    with pytest.raises(ValueError, match='exceeds unity'):
        PB.primary_beam_generator(far, NP.array([0.61]), {'id': 'gmrt'}, freq_scale='GHz', skyunits='altaz')
    with pytest.raises(ValueError, match='NaN'):
        PB.primary_beam_generator(NP.array([[89.9, 0.0]]), NP.array([0.235]), {'id': 'ugmrt'}, freq_scale='GHz', skyunits='altaz')


def test_interferometer_array_save_hdf5_layout(tmp_path):
    """InterferometerArray.save: PRISim's HDF5 groups / datasets (interferometry.py:8723-8846) through the ctypes HDF5 writer."""
    from prisim_amd import hdf5io
    try:
        hdf5io._load()
    except hdf5io.HDF5Unavailable:
        pytest.skip('the HDF5 C library is not installed')
    ch = 150e6 + NP.arange(16) * 1e5
    bl = NP.array([[14.6, 0.0, 0.0], [0.0, 29.2, 0.0]])
    skymod = SM.SkyModel(location=[[80.0, 100.0], [50.0, 10.0]], flux_ref=[1.0, 3.0], spindex=[0.0, -0.7], ref_freq=150e6)
    ia = RI.InterferometerArray([('a1', 'a0'), ('a2', 'a0')], bl, ch, telescope={'id': 'hera', 'shape': 'dish', 'size': 14.0, 'orientation': [90.0, 270.0], 'ocoords': 'altaz'},
                                skycoords='altaz', pointing_coords='altaz', A_eff=154.0, eff_Q=0.96, latitude=-30.72)
    for j in range(2):
        ia.observe((2457000.5 + j, 15.0 * j), {'Tnet': 300.0}, NP.ones(16), [90.0, 270.0], skymod, 60.0)
    ia.generate_noise(seed=3)
    with pytest.warns(UserWarning):
        ia.add_noise()
    ia.delay_transform(pad=0.0, verbose=False)
    out = str(tmp_path / 'simvis')
    fname = ia.save(out, fmt='HDF5', npz=True, overwrite=False, verbose=False)
    assert fname == out + '.hdf5' and os.path.exists(out + '.npz')
    with hdf5io.File(fname, 'r') as f:
        assert f.read('header/flux_unit') == 'JY' and f.read('telescope_parms/id') == 'hera'
        assert f.read('telescope_parms/latitude') == -30.72 and f.read_attr('telescope_parms/latitude', 'units') == 'deg'
        assert NP.array_equal(f.read('spectral_info/freqs'), ch) and NP.array_equal(f.read('spectral_info/lags'), ia.lags)
        assert f.read('timing/n_acc') == 2 and NP.array_equal(f.read('timing/t_acc'), [60.0, 60.0])
        assert NP.array_equal(f.read('skyparms/LST'), [0.0, 15.0]) and f.read('skyparms/skycoords') == 'altaz'
        lab = f.read('array/labels')
        assert lab.dtype.names == ('A2', 'A1') and lab[1][0] == b'a2'
        assert NP.array_equal(f.read('array/baselines'), bl) and f.read_attr('array/baselines', 'units') == 'm'
        for name, arr in (('skyvis', ia.skyvis_freq), ('vis', ia.vis_freq), ('noise', ia.vis_noise_freq), ('rms', ia.vis_rms_freq)):
            assert NP.array_equal(f.read('visibilities/freq_spectrum/' + name), arr), name
        assert NP.array_equal(f.read('visibilities/delay_spectrum/skyvis'), ia.skyvis_lag)
        assert f.read_attr('visibilities/delay_spectrum/skyvis', 'units') == 'Jy Hz'
        assert NP.array_equal(f.read('instrument/Tnet'), [300.0, 300.0]) and f.read('instrument/Tsys').shape == ia.Tsys.shape
    z = NP.load(out + '.npz')
    assert NP.array_equal(z['skyvis_freq'], ia.skyvis_freq) and NP.array_equal(z['vis_freq'], ia.vis_freq)
    # init_file: the object comes back (interferometry.py:5184-5658) with its cube resident on the device again
    ib = RI.InterferometerArray(None, None, None, init_file=out)
    assert ib.labels == [('a1', 'a0'), ('a2', 'a0')] and NP.array_equal(ib.baselines, bl) and NP.array_equal(ib.channels, ch)
    assert ib.n_acc == 2 and ib.t_acc == [60.0, 60.0] and ib.lst == [0.0, 15.0] and ib.telescope['id'] == 'hera'
    assert ib.skycoords == 'altaz' and ib.flux_unit == 'JY' and ib.Tsysinfo[0]['Tnet'] == 300.0 and ib.latitude == -30.72
    assert NP.array_equal(ib.skyvis_freq, ia.skyvis_freq) and NP.array_equal(ib.vis_noise_freq, ia.vis_noise_freq)
    assert NP.array_equal(ib._ctx.get_vis(slot=1), ia.skyvis_freq[:, :, 1])
    ib.delay_transform(pad=0.0, verbose=False)                                     # device-side work on the reloaded cube
    assert NP.allclose(ib.skyvis_lag, ia.skyvis_lag, rtol=0, atol=1e-12 * NP.abs(ia.skyvis_lag).max())
    ib.observe((2457002.5, 30.0), {'Tnet': 300.0}, NP.ones(16), [90.0, 270.0], skymod, 60.0)     # and it keeps observing
    assert ib.skyvis_freq.shape == (2, 16, 3) and ib.n_acc == 3
    with pytest.warns(UserWarning, match='could not open'):
        ic = RI.InterferometerArray(['x'], [[1.0, 2.0, 0.0]], ch, init_file=str(tmp_path / 'missing'), skycoords='altaz', pointing_coords='altaz')
    assert ic.n_acc == 0 and ic.labels == ['x']
    with pytest.raises(IOError):
        ia.save(out, fmt='HDF5', npz=False, overwrite=False, verbose=False)          # exists
    with pytest.raises(NotImplementedError):
        ia.save(out, fmt='FITS', verbose=False)
    with pytest.raises(ValueError):
        ia.save(out, fmt='CSV', verbose=False)


def test_observe_keeps_reserved_snapshots_on_the_device_until_read():
    """After reserve(n) every snapshot stays in its own slot of the device cube and the host copy is only made when
    skyvis_freq is first read (a sharded run that gathers on the device never makes it); without reserve() the single slot
    is reused, so a parked snapshot is fetched before the next one overwrites it."""
    ch = 150e6 + NP.arange(16) * 1e5
    bl = NP.array([[14.6, 0.0, 0.0], [0.0, 29.2, 0.0], [100.0, -40.0, 0.0]])
    skymod = SM.SkyModel(location=[[80.0, 100.0], [50.0, 10.0], [30.0, 250.0]], flux_ref=[1.0, 3.0, 2.0], spindex=[0.0, -0.7, -1.0], ref_freq=150e6)
    def run(reserve):
        ia = RI.InterferometerArray(['a', 'b', 'c'], bl, ch, telescope={'shape': 'delta'}, skycoords='altaz', pointing_coords='altaz')
        if reserve:
            ia.reserve(3)
        for j in range(3):
            ia.observe((2457000.5 + j, 20.0 * j), {'Tnet': 100.0}, NP.ones(16), [90.0 - 5.0 * j, 270.0], skymod, 60.0)
        return ia
    ia = run(True)
    assert all(isinstance(s, RI._DeviceSlot) for s in ia._cube)                    # nothing downloaded yet
    for t in range(3):
        assert ia._ctx.get_vis(slot=t).shape == (3, 16)
    cube = ia.skyvis_freq                                                          # first read fetches the three slots
    assert cube.shape == (3, 16, 3) and not any(isinstance(s, RI._DeviceSlot) for s in ia._cube)
    ib = run(False)                                                                # one slot, reused: same numbers
    assert not any(isinstance(s, RI._DeviceSlot) for s in ib._cube)                # the parked first one was fetched before its slot was reused
    assert NP.array_equal(ib.skyvis_freq, cube)
    assert NP.abs(cube[:, :, 0] - cube[:, :, 1]).max() > 0                         # the snapshots do differ


def test_sharded_run_gathers_visibilities_then_delay_spectra():
    """The multi-GPU driver's sequence on a one-rank RCCL communicator: observe into reserved device slots (nothing downloaded),
    all-gather the shards, delay-transform the local shard on the device, all-gather the delay spectra through the same slots."""
    ch = 150e6 + NP.arange(32) * 1e5
    bl = NP.array([[14.6, 0.0, 0.0], [0.0, 29.2, 0.0], [100.0, -40.0, 0.0], [55.0, 60.0, 0.0]])
    skymod = SM.SkyModel(location=[[80.0, 100.0], [50.0, 10.0], [30.0, 250.0]], flux_ref=[1.0, 3.0, 2.0], spindex=[0.0, -0.7, -1.0], ref_freq=150e6)
    ia = RI.InterferometerArray(['a', 'b', 'c', 'd'], bl, ch, telescope={'shape': 'delta'}, skycoords='altaz', pointing_coords='altaz')
    ia.reserve(2)
    for j in range(2):
        ia.observe((2457000.5 + j, 20.0 * j), {'Tnet': 100.0}, NP.ones(32), [90.0, 270.0], skymod, 60.0)
    with pytest.raises(RuntimeError):
        ia.allgather_lags(1)                                           # needs the communicator of allgather()
    gathered = ia.allgather(_abi.Context.comm_unique_id(), 1, 0)
    assert all(isinstance(s, RI._DeviceSlot) for s in ia._cube)        # the shard itself never went to the host
    w = NP.hanning(32) + 0.1
    ia.delay_transform(pad=1.0, freq_wts=w, verbose=False)
    lags = ia.allgather_lags(1)
    assert lags.shape == (4, 32, 2) and NP.array_equal(lags, ia.skyvis_lag)
    assert NP.array_equal(gathered, ia.skyvis_freq)
    ref_lag, _ = DO.delay_transform(ia.skyvis_freq, ia.bp, ia.bp_wts, ia.freq_resolution, pad=1.0)
    assert NP.max(NP.abs(lags - ref_lag)) <= 1e-10 * NP.max(NP.abs(ref_lag))
    for t in range(2):                                                 # the visibilities are back in their slots afterwards
        assert NP.array_equal(ia._ctx.get_vis(slot=t), ia.skyvis_freq[:, :, t])


def test_shard_map_puts_every_gathered_cube_into_global_baseline_order(ctx):
    """prisim_hip_set_shard_map on one rank (with the box's RCCL): the shard's rows are a shuffled, padded listing of a 61-baseline array;
    after every kind of gather -- whole cube, complex64 on the wire, per-slot on the communication stream, to a root, delay spectra,
    gradients -- the gathered cube on the device is [nt][nbl_total][row] in the array's own order with the padding rows dropped
    (scripts/run_prisim.py:2233-2242: what the rank-0 concatenate leaves).  Multi-rank: tests/test_gpu_multirank_standin.py."""
    rng = NP.random.default_rng(17)
    nbl_total, nbl, nchan, nt = 61, 70, 32, 3
    ch = 150e6 + NP.arange(nchan) * 1.0e5
    smap = NP.full(nbl, -1, dtype=NP.int64)
    rows = rng.permutation(nbl)[:nbl_total]                       # the local rows that are real; the other 9 are padding
    smap[rows] = rng.permutation(nbl_total)
    real = smap >= 0
    bl = rng.uniform(-100, 100, (nbl, 3))
    s = O.altaz2dircos(NP.stack((rng.uniform(20, 90, 40), rng.uniform(0, 360, 40)), 1))
    p = rng.uniform(0, 3, (40, nchan))
    zen = NP.array([0.0, 0.0, 1.0])

    def ordered(local):                                           # (..., nbl, row) -> (..., nbl_total, row)
        out = NP.empty(local.shape[:-2] + (nbl_total, local.shape[-1]), dtype=local.dtype)
        out[..., smap[real], :] = local[..., real, :]
        return out
    c = _abi.Context(0)
    try:
        c.set_array(bl, ch, nt_max=nt)
        c.comm_init(_abi.Context.comm_unique_id(), 1, 0)
        with pytest.raises(ValueError):
            c.set_shard_map(NP.zeros((1, nbl), dtype=NP.int64), nbl_total)           # a baseline listed twice
        bad = smap.copy(); bad[rows[0]] = -1
        with pytest.raises(ValueError):
            c.set_shard_map(bad[None, :], nbl_total)                                  # a baseline missing
        c.set_shard_map(smap[None, :], nbl_total)
        c.set_sky(s, p, zen)
        c.comm_stats(reset=True)
        for t in range(nt):
            c.compute(want_grad=True, slot=t)
            c.allgather_slot_async(t)                             # gather + un-deal on the communication stream, under the next sky-sum
        c.sync()
        vis = NP.stack([c.get_vis(slot=t) for t in range(nt)])
        grad = NP.stack([c.get_vis(slot=t, want_grad=True)[1] for t in range(nt)])
        g = c.get_gathered(nt)
        assert g.shape == (nt, nbl_total, nchan) and NP.array_equal(g, ordered(vis))
        st = c.comm_stats()
        assert st['n_gathers'] == nt and 0.0 < st['sum_undeal_ms'] < st['sum_gather_ms'] and st['last_undeal_ms'] > 0.0
        assert abs(c.gathered_checksum(nt) - (g.real.sum() + g.imag.sum())) <= 1e-9 * NP.abs(g).sum()
        c.allgather(nt, complex64=True)
        g32 = c.get_gathered(nt)
        assert g32.dtype == NP.complex64 and NP.array_equal(g32, ordered(vis.astype(NP.complex64)))
        c.set_gather_root(0)
        c.allgather(nt)
        assert NP.array_equal(c.get_gathered(nt), ordered(vis))
        c.set_gather_root(None)
        c.allgather_grad(nt)
        gg = c.get_gathered_grad(nt)
        assert gg.shape == (nt, 3, nbl_total, nchan) and NP.array_equal(gg, ordered(grad))
        c.delay_transform_device(nt, pad=1.0, want_lag=True)
        lags = c.get_lags(0, nt)
        c.allgather_lags(nt)
        gl = c.get_gathered(nt, row=lags.shape[2])
        assert NP.array_equal(gl, ordered(lags))
        # back to the rank-major layout
        c.set_shard_map(None, 0)
        c.allgather(nt)
        assert NP.array_equal(c.get_gathered(nt, 1)[:, 0], vis)
    finally:
        c.close()
