"""TEST INFRASTRUCTURE: stand-ins injected at the `_abi.Context` seam so that the PRODUCT's multi-rank code
(prisim_amd.driver.run -> InterferometerArray.observe / allgather / delay_transform / allgather_lags) runs where RCCL cannot:

  OracleContext     no GPU at all (CPU suite): compute() is the numpy oracle, the delay transform its numpy restatement, and the
                    exchange an all-gather of host arrays over the product's own socket rendezvous (prisim_amd.rendezvous).
  HostCommContext   the real Context (HIP kernels, device cube, device-resident delay spectra) with only the communicator
                    replaced by that host exchange -- two ranks can then share the one GPU of a test box, which RCCL refuses.

No torch: the workers are started by prisim_amd.launch and set `fake_context.RDZV` to their Rendezvous before the first exchange.

Only tests/ imports this module."""
import numpy as NP

from oracle import skyvis_oracle as O, beams_oracle as BO, delay_oracle as DO
from prisim_amd import _abi


RDZV = None          # the worker's prisim_amd.rendezvous.Rendezvous: carries the stand-in communicator's data (RCCL's place)


def _host_allgather(arr):
    """Every rank's array (same shape and dtype on all ranks), in rank order."""
    a = NP.ascontiguousarray(arr)
    parts = RDZV.allgather_bytes(a.tobytes())
    return [NP.frombuffer(p, dtype=a.dtype).reshape(a.shape).copy() for p in parts]


class _HostShardMap(object):
    """Host statement of prisim_hip_set_shard_map + the un-deal kernel for the stand-in contexts: gathered cubes in the global baseline
    order of the unsharded array, padding rows (negative entries) dropped."""
    nbl_total = 0
    _smap = None

    def set_shard_map(self, bl_index, nbl_total):
        if bl_index is None:
            self._smap, self.nbl_total = None, 0
            return
        m = NP.asarray(bl_index, dtype=NP.int64)
        assert m.ndim == 2 and m.shape == (getattr(self, 'nranks', 1), self.nbl), (m.shape, getattr(self, 'nranks', 1), self.nbl)
        real = m[m >= 0]
        assert NP.array_equal(NP.sort(real), NP.arange(nbl_total))
        self._smap, self.nbl_total = m, int(nbl_total)

    def _undeal(self, g, bl_axis):
        """g: [t][rank]...[b]...: rank axis 1, local baseline axis `bl_axis` -> the rank axis dropped, baseline axis global."""
        if self._smap is None:
            return g
        shape = list(g.shape)
        del shape[1]
        shape[bl_axis - 1] = self.nbl_total
        out = NP.empty(shape, dtype=g.dtype)
        for r in range(self._smap.shape[0]):
            keep = self._smap[r] >= 0
            src = NP.take(g[:, r], NP.flatnonzero(keep), axis=bl_axis - 1)
            idx = [slice(None)] * out.ndim
            idx[bl_axis - 1] = self._smap[r][keep]
            out[tuple(idx)] = src
        return out


class OracleContext(_HostShardMap):
    def __init__(self, device=0):
        self.nbl = self.nchan = self.nt_max = self.nsrc = 0
        self._timing = {'last_terms': 0, 'last_delay_ms': 0.0, 'last_delay_fused': 0, 'last_chan_tile': 0, 'last_nsplit': 1,
                        'sum_kernel_ms': 0.0, 'n_kernel': 0, 'last_kernel_ms': 0.0, 'last_taper_group': 0}

    def close(self):
        pass

    def set_array(self, baselines, freqs_hz, nt_max=1):
        self.bl = NP.asarray(baselines, dtype=NP.float64).reshape(-1, 3)
        self.ch = NP.asarray(freqs_hz, dtype=NP.float64).ravel()
        self.nbl, self.nchan, self.nt_max = self.bl.shape[0], self.ch.size, int(nt_max)
        self.cube = NP.zeros((self.nt_max, self.nbl, self.nchan), dtype=NP.complex128)
        self._grad = {}                               # (prisim_hip_set_array releases the gradient cube)
        self._cat = None                              # ... and drops the resident catalogue

    def set_sky_analytic(self, dircos, flux_ref, spindex, ref_freq_hz, beam_kind, diameter_m, beam_pc_dircos, pc_dircos, fwhm_deg=None,
                         flux_spectrum=None, ext=None):
        self.dircos = NP.asarray(dircos, dtype=NP.float64).reshape(-1, 3)
        self.nsrc = self.dircos.shape[0]
        flux = NP.asarray(flux_spectrum) if flux_spectrum is not None else \
            NP.asarray(flux_ref)[:, None] * (self.ch[None, :] / ref_freq_hz) ** NP.asarray(spindex)[:, None]
        if beam_kind == _abi.PRISIM_BEAM_DELTA:
            pb = NP.ones_like(flux)
        elif beam_kind == _abi.PRISIM_BEAM_AIRY:
            pb = BO.airy_disk_pattern(diameter_m, _altaz(self.dircos), self.ch)
        elif beam_kind == _abi.PRISIM_BEAM_GAUSSIAN:
            pb = BO.gaussian_beam(diameter_m, _altaz(self.dircos), self.ch, power=True)
        else:
            raise NotImplementedError('OracleContext: beam kind %d' % beam_kind)
        self.pb = pb * flux
        self.pc = NP.asarray(pc_dircos, dtype=NP.float64).ravel()
        self.fwhm = None if fwhm_deg is None else NP.asarray(fwhm_deg, dtype=NP.float64)

    # ---- device-resident catalogue (ABI 0.5): the host statements the device kernels restate (prisim_amd/geometry.py) ----
    def set_catalog(self, location, coords, flux_ref=None, spindex=None, ref_freq_hz=None, flux_spectrum=None, fwhm_deg=None, unitvec='host'):
        self._cat = {'loc': NP.asarray(location, dtype=NP.float64).reshape(-1, 2), 'coords': coords,
                     'flux_ref': None if flux_ref is None else NP.asarray(flux_ref, dtype=NP.float64),
                     'spindex': None if spindex is None else NP.asarray(spindex, dtype=NP.float64), 'ref_freq': ref_freq_hz,
                     'spec': None if flux_spectrum is None else NP.asarray(flux_spectrum, dtype=NP.float64).reshape(-1, self.nchan),
                     'fwhm': None if fwhm_deg is None else NP.asarray(fwhm_deg, dtype=NP.float64)}
        self.ncat = self._cat['loc'].shape[0]

    @staticmethod
    def make_obs(latitude_deg, roi_radius_deg=90.0, roi_center='zenith', beam_kind=_abi.PRISIM_BEAM_DELTA, diameter_m=1.0, ext=None,
                 use_external_beam=False):
        if use_external_beam:
            raise NotImplementedError('OracleContext: analytic beams only')
        # (beam extensions are ignored, like set_sky_analytic of this stand-in does)
        return {'lat': float(latitude_deg), 'roi_radius': float(roi_radius_deg), 'roi_center': roi_center, 'kind': beam_kind, 'dia': diameter_m}

    def _roi(self, obs, lst, pc_dircos, frame=None):
        """geometry.frame_dircos / roi_select: the host statement of cat_source() (catalog_kernels.hip); frame None = the library's
        fall-back rotation (hour angle = LST - RA)."""
        from prisim_amd import geometry as GEOM, frames as FR
        cat = self._cat
        if cat is None:
            raise RuntimeError('set_catalog must be called first (set_array drops the catalogue)')
        if frame is None:
            frame = FR.snapshot_frame(cat['coords'], lst, obs['lat'], model='date')
        dc_all = GEOM.frame_dircos(GEOM.catalog_unitvec(cat['loc'], cat['coords']), frame[0], frame[1])
        m2 = GEOM.roi_select(dc_all, obs['roi_center'], obs['roi_radius'], pc_dircos)
        return m2, dc_all[m2]

    def catalog_roi(self, obs, lst_deg, pc_dircos, want_indices=True, want_dircos=True, frame=None):
        m2, dc = self._roi(obs, float(lst_deg), NP.asarray(pc_dircos, dtype=NP.float64), frame)
        return m2.astype(NP.int64), dc

    def observe_catalog(self, obs, lst_deg, pc_dircos, beam_pc_dircos=None, precision=0, want_grad=False, slot0=0, host_cube=None, gather=None,
                        frames=None):
        lst = NP.asarray(lst_deg, dtype=NP.float64).ravel()
        k = lst.size
        pc = NP.broadcast_to(NP.asarray(pc_dircos, dtype=NP.float64).reshape(-1, 3), (k, 3))
        bpc = pc if beam_pc_dircos is None else NP.broadcast_to(NP.asarray(beam_pc_dircos, dtype=NP.float64).reshape(-1, 3), (k, 3))
        counts = NP.zeros(k, dtype=NP.int64)
        cat = self._cat
        for t in range(k):
            m2, dc = self._roi(obs, lst[t], pc[t], None if frames is None else frames[t])
            counts[t] = m2.size
            if m2.size == 0:
                self.cube[slot0 + t] = 0.0
                if want_grad:
                    self._grad = getattr(self, '_grad', {})
                    self._grad[slot0 + t] = NP.zeros((3, self.nbl, self.nchan), dtype=NP.complex128)
            else:
                fw = None if cat['fwhm'] is None else cat['fwhm'][m2]
                if cat['spec'] is not None:
                    self.set_sky_analytic(dc, None, None, None, obs['kind'], obs['dia'], bpc[t], pc[t], fwhm_deg=fw, flux_spectrum=cat['spec'][m2])
                else:
                    self.set_sky_analytic(dc, cat['flux_ref'][m2], cat['spindex'][m2], cat['ref_freq'], obs['kind'], obs['dia'], bpc[t], pc[t], fwhm_deg=fw)
                self.compute(precision=precision, want_grad=want_grad, slot=slot0 + t)
            if host_cube is not None:
                host_cube[slot0 + t] = self.cube[slot0 + t].astype(host_cube.dtype)
            if gather is not None:
                raise NotImplementedError('OracleContext: per-snapshot gathers')
        self.nsrc = int(counts[-1]) if k else 0
        return counts

    def compute(self, precision=0, kernel=0, want_grad=False, slot=0):
        if want_grad:
            self.cube[slot], g = O.skyvis(self.bl, self.ch, self.dircos, self.pb, self.pc, fwhm_deg=self.fwhm, gradient=True)
            self._grad = getattr(self, '_grad', {})
            self._grad[slot] = g
        else:
            self.cube[slot] = O.skyvis(self.bl, self.ch, self.dircos, self.pb, self.pc, fwhm_deg=self.fwhm)
        self._timing['last_terms'] = self.nbl * self.nchan * self.nsrc

    def get_vis(self, slot=0, want_grad=False, complex64=False):
        ctype = NP.complex64 if complex64 else NP.complex128
        v = self.cube[slot].astype(ctype)
        return (v, self._grad[slot].astype(ctype)) if want_grad else v

    def noise(self, rms, seed, bl_offset=0, bl_index=None):
        # like the device generator, keyed on the GLOBAL baseline index: a shard draws what the unsharded run draws for its baselines
        r = NP.asarray(rms, dtype=NP.float64)                                  # (nt, nbl, nchan)
        out = NP.empty(r.shape, dtype=NP.complex128)
        for b in range(r.shape[1]):
            rng = NP.random.default_rng([int(seed) & 0xFFFFFFFF, (int(bl_offset) + b) if bl_index is None else int(bl_index[b])])
            z = rng.standard_normal((r.shape[0], r.shape[2], 2))
            out[:, b, :] = r[:, b, :] / NP.sqrt(2.0) * (z[..., 0] + 1j * z[..., 1])
        return out

    def set_vis(self, vis, slot=0):
        self.cube[slot] = vis

    def sync(self):
        pass

    def phase_rotate(self, nt, diff_dircos):
        d = NP.asarray(diff_dircos, dtype=NP.float64).reshape(-1, 3)
        for t in range(nt):
            self.cube[t] = self.cube[t] * NP.exp(-2j * NP.pi * self.ch[None, :] * (self.bl @ d[t])[:, None] / 299792458.0)

    def timing(self, reset=False):
        return dict(self._timing)

    # ---- communicator: the socket rendezvous stands in for RCCL ----
    @staticmethod
    def comm_unique_id():
        return bytes(range(128))

    def comm_init(self, uid, nranks, rank):
        assert len(uid) == 128
        self.nranks, self.rank = int(nranks), int(rank)

    def set_gather_root(self, root=None):
        self._root = root           # the host stand-in delivers to every rank whatever the root: only who READS differs

    def allgather(self, nt, complex64=False):
        parts = _host_allgather(self.cube[:nt])                               # [rank][t][b][f]
        self._gathered = NP.stack(parts, axis=1)                              # [t][rank][b][f]

    def get_gathered(self, nt, nranks=None, row=None):
        return self._undeal(self._gathered[:nt], 2)

    def allgather_grad(self, nt, complex64=False):
        mine = NP.stack([self._grad[t] for t in range(nt)])                   # [t][k][b][f]
        self._gathered_grad = NP.stack(_host_allgather(mine), axis=1)         # [t][rank][k][b][f]

    def get_gathered_grad(self, nt, nranks=None):
        return self._undeal(self._gathered_grad[:nt], 3)

    def comm_selftest(self, nbytes=1 << 20):
        got = _host_allgather(NP.full(4, self.rank, dtype=NP.int32))
        assert [int(g[0]) for g in got] == list(range(self.nranks))

    def comm_stats(self, reset=False):
        return {'n_gathers': 0, 'bytes_per_peer': 0, 'sum_gather_ms': 0.0, 'last_gather_ms': 0.0, 'max_gather_ms': 0.0,
                'last_gather_after_compute_ms': 0.0, 'stream_priority': 0, 'stream_priority_lowest': 0, 'nranks': getattr(self, 'nranks', 1)}

    # ---- downloads ----
    def get_vis_async(self, slot, out, grad_out=None):
        out[...] = self.cube[slot].astype(out.dtype)
        if grad_out is not None:
            grad_out[...] = self._grad[slot].astype(out.dtype)

    def wait_downloads(self):
        pass

    # ---- delay transform ----
    def delay_transform_device(self, nt, bpwts=None, pad=1.0, want_lag=True, want_power=False, power_scale=1.0):
        w = NP.ones((self.nbl, self.nchan)) if bpwts is None else NP.broadcast_to(NP.asarray(bpwts, dtype=NP.float64).reshape(-1, self.nchan),
                                                                                    (self.nbl, self.nchan))
        vis = NP.transpose(self.cube[:nt], (1, 2, 0))
        lag, lags = DO.delay_transform(vis, w[:, :, None], NP.ones((self.nbl, self.nchan, 1)), self.ch[1] - self.ch[0], pad=pad)
        self._lag = NP.ascontiguousarray(NP.transpose(lag, (2, 0, 1)))        # [t][b][lag]
        self._pow = NP.abs(self._lag) ** 2 * power_scale if want_power else None
        self._dt_nout = self._lag.shape[2]
        self._dt_generation = getattr(self, '_dt_generation', 0) + 1
        return lags, self._dt_nout

    def get_delay_power(self, t0, nt, rows=None):
        out = self._pow[t0:t0 + nt]
        return out if rows is None else out[:, NP.asarray(rows)]

    def get_pbflux(self):
        return NP.array(self.pb)

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def get_lags(self, t0, nt, rows=None):
        out = self._lag[t0:t0 + nt]
        return out if rows is None else out[:, NP.asarray(rows)]

    def allgather_lags(self, nt):
        self._gathered = NP.stack(_host_allgather(self._lag[:nt]), axis=1)

    def delay_transform_host(self, vis, bpwts, pad):
        w = NP.ones((self.nbl, self.nchan)) if bpwts is None else NP.asarray(bpwts)
        lag, lags = DO.delay_transform(NP.asarray(vis)[:, :, None], w[:, :, None], NP.ones((self.nbl, self.nchan, 1)), self.ch[1] - self.ch[0], pad=pad)
        return lag[:, :, 0], lags, None


def _altaz(dircos):
    alt = NP.degrees(NP.arcsin(NP.clip(dircos[:, 2], -1.0, 1.0)))
    az = NP.degrees(NP.arctan2(dircos[:, 0], dircos[:, 1])) % 360.0
    return NP.stack((alt, az), axis=1)


class HostCommContext(_HostShardMap, _abi.Context):
    """The real GPU context; only the exchange goes through the host (socket rendezvous), so that two ranks can share one GPU."""

    def comm_init(self, uid, nranks, rank):
        assert len(uid) == 128
        self.nranks, self.rank = int(nranks), int(rank)

    def set_gather_root(self, root=None):
        self._root = root

    def allgather(self, nt, complex64=False):
        mine = NP.stack([self.get_vis(slot=t) for t in range(nt)])
        self._gathered_host = NP.stack(_host_allgather(mine), axis=1)

    def allgather_lags(self, nt):
        self._gathered_host = NP.stack(_host_allgather(self.get_lags(0, nt)), axis=1)

    def allgather_grad(self, nt, complex64=False):
        mine = NP.stack([self.get_vis(slot=t, want_grad=True)[1] for t in range(nt)])
        self._gathered_grad_host = NP.stack(_host_allgather(mine), axis=1)

    def get_gathered_grad(self, nt, nranks=None):
        return self._undeal(self._gathered_grad_host[:nt], 3)

    def comm_selftest(self, nbytes=1 << 20):
        got = _host_allgather(NP.full(4, self.rank, dtype=NP.int32))
        assert [int(g[0]) for g in got] == list(range(self.nranks))

    def get_gathered(self, nt, nranks=None, row=None):
        return self._undeal(self._gathered_host[:nt], 2)
