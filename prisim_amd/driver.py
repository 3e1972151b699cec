"""YAML-driven simulation loop -- the accelerated counterpart of scripts/run_prisim.py of the reference
for the keys the BASELINE configs use (SURVEY.md 8(a) A10, section 5 "config / flags").

Kept from the reference driver (scripts/run_prisim.py): the `-i parms.yaml` entry, the parameter-file schema
(prisim/examples/simparms/defaultparms.yaml) with `preload.template` deep merge (:67-101), the channel grid (:900),
the antenna layout presets / baseline selection (getBaselineInfo, interferometry.py:1465-2013), drift / track
pointing schedules (:709-733), the custom catalog columns RA DEC F_INT SPINDEX MAJAX MINAX PA and its flux cut
(:1645-1684), baseline sharding across processes (`pp.key: 'bl'`, :1775-1791) and the NPZ output keys
(interferometry.py:8859-8863).

Replaced: `mpirun` ranks + per-rank part files + rank-0 concatenate become one process per GPU (`run_prisim.py -n N` starts them
itself through prisim_amd.launch; any launcher that sets RANK / WORLD_SIZE / LOCAL_RANK works too) and a single RCCL all-gather of the
visibility cube;
the rank-0 ROI/beam precompute through FITS files disappears (beams are fused on the device).
Not offered (out of scope): survey catalogs (SUMSS/NVSS/GLEAM/GSM need prisim/data, absent),
gains, uvfits/uvh5 writers, plots, resource monitor (`pp.key: 'freq' | 'src'` are accepted and run on baseline shards).  After the snapshots: thermal noise, re-centring on
phasing.center, delay transform and the npz / HDF5 files, in the reference's order (:2278-2286); in a sharded run every rank does the
per-baseline steps on its own shard and rank 0 puts the whole array together from the gathered cubes (assemble_full_array).  Two synthetic sky models
are added because the reference's catalogs are not available offline: skyparm.model 'ptsrc_random' and 'healpix_synthetic'.
Added key: processing.snapshot_batch (default 64) -- accumulations sent to the device per observe_batch() call (one geometry
read-back per batch; arrays of at most 256 baselines put the whole batch into one sky-sum launch).
Time: without astropy the LST ramp is lst_init + t * 15.0410686 deg/h (mean sidereal rate), jd from jd_init or obs_date.
"""
import copy
import hashlib
import datetime
import os
import time
import warnings

import numpy as NP
import yaml

from . import _abi
from . import frames as FRAMES
from . import watchdog as WATCHDOG
from . import geometry as GEOM
from . import interferometry as RI
from . import layouts as LAY
from . import sharding as SH
from . import skymodel as SM
from . import workloads as W

SIDEREAL_DEG_PER_SEC = 360.0 * 1.00273790935 / 86400.0

DEFAULTS = {
    'preload': {'template': None},
    'dirstruct': {'rootdir': './', 'project': 'prisim_amd_run', 'simid': None},
    'telescope': {'label_prefix': '', 'id': 'custom', 'latitude': -30.7224, 'longitude': 21.4278, 'altitude': 0.0,
                  'A_eff': 154, 'eff_Q': 0.96, 'Trx': 50.0, 'Tant_freqref': 150e6, 'Tant_spindex': -2.55, 'Tant_ref': 200.0,
                  'Tsys': None},
    'array': {'redundant': True, 'layout': 'HERA-19', 'file': None, 'seed': 200},
    'baseline': {'min': None, 'max': None, 'direction': None},
    'antenna': {'shape': 'dish', 'size': 14.0, 'orientation': [90.0, 270.0], 'ocoords': 'altaz', 'phased_array': False,
                'ground_plane': None},
    'beam': {'use_external': False, 'file': None, 'filefmt': 'npz', 'chromatic': True, 'select_freq': None, 'spec_interp': 'cubic'},
    'bandpass': {'freq': 150e6, 'freq_resolution': 390625.0, 'nchan': 256},
    'obsparm': {'obs_date': '2015/11/23', 'obs_mode': 'drift', 't_obs': None, 'n_acc': 2, 't_acc': 1080.0},
    'pointing': {'file': None, 'jd_init': None, 'lst_init': 0.0,
                 'drift_init': {'alt': None, 'az': None, 'ha': 0.0, 'dec': -30.7224},
                 'track_init': {'ra': 0.0, 'dec': -30.7224, 'ha': 0.0, 'epoch': '2000'}},
    'skyparm': {'model': 'custom', 'epoch': '2000', 'nside': 16, 'flux_unit': 'Jy', 'custom_reffreq': 0.150, 'flux_min': 0.0,
                'flux_max': None, 'fluxcut_reffreq': None, 'spindex': -0.83, 'roi_radius': None, 'n_src': 100, 'seed': 1},
    'catalog': {'custom_file': 'custom_catalog.txt'},
    'processing': {'gradient_mode': None, 'f_pad': 1.0, 'bpass_shape': 'bhw', 'delay_transform': False, 'memsave': False,
                   'add_noise': None, 'noise_seed': None},
    'phasing': {'center': [90.0, 270.0], 'coords': 'altaz'},
    'pp': {'key': 'bl', 'eqvol': True, 'gather': None},      # gather: 'all' | 'root'; None = 'root' when only rank 0 keeps a host copy, else 'all'
    'save_redundant': True,
    'save_formats': {'npz': True, 'hdf5': False, 'npz_compress': True},
    'diagnosis': {'wait_after_run': False},
}


def deep_merge(base, override):
    """Recursively overlay `override` on `base` (the reference merges three levels by hand, run_prisim.py:67-101)."""
    out = copy.deepcopy(base)
    for key, val in (override or {}).items():
        if isinstance(val, dict) and isinstance(out.get(key), dict):
            out[key] = deep_merge(out[key], val)
        else:
            out[key] = copy.deepcopy(val)
    return out


def load_parms(infile):
    with open(infile, 'r') as f:
        parms = yaml.safe_load(f) or {}
    template = (parms.get('preload') or {}).get('template')
    if template is not None:
        if not os.path.isabs(template):
            template = os.path.join(os.path.dirname(os.path.abspath(infile)), template)
        with open(template, 'r') as f:
            parms = deep_merge(yaml.safe_load(f) or {}, parms)
    return deep_merge(DEFAULTS, parms)


def telescope_dict(parms):
    """run_prisim.py:103-230 condensed: telescope id + antenna element description."""
    tel, ant = parms['telescope'], parms['antenna']
    out = {'latitude': tel['latitude'], 'longitude': tel['longitude'], 'altitude': tel['altitude']}
    if tel['id'] not in (None, 'custom'):
        out['id'] = tel['id']
        if ant.get('orientation') is not None:
            out['orientation'], out['ocoords'] = ant['orientation'], ant['ocoords']
    else:
        out['shape'] = ant['shape']
        out['size'] = ant['size']
        out['orientation'], out['ocoords'] = ant['orientation'], ant['ocoords']
    out['groundplane'] = ant.get('ground_plane')
    return out


def read_layout_file(path):
    """Whitespace table with a header naming East / North / Up columns (array.parser defaults, defaultparms.yaml)."""
    with open(path) as f:
        lines = [ln for ln in f if ln.strip() and not ln.lstrip().startswith('#')]
    header = lines[0].split()
    cols = {name.lower(): i for i, name in enumerate(header)}
    for need in ('east', 'north'):
        if need not in cols:
            raise KeyError('layout file must have East and North columns')
    rows = [ln.split() for ln in lines[1:]]
    east = NP.array([float(r[cols['east']]) for r in rows])
    north = NP.array([float(r[cols['north']]) for r in rows])
    up = NP.array([float(r[cols['up']]) for r in rows]) if 'up' in cols else NP.zeros(east.size)
    return NP.stack((east, north, up), axis=1)


def baseline_info(parms):
    """Antenna positions -> baselines (all pairs j>i, folded, length-sorted), length selection, redundancy handling."""
    arr = parms['array']
    if arr.get('file'):
        pos = read_layout_file(arr['file'])
    else:
        pos = LAY.array_layout(arr['layout'])
    bl, ids = LAY.baseline_generator(pos)
    bl, ids = LAY.fold_and_sort_baselines(bl, ids)
    length = NP.sqrt(NP.sum(bl ** 2, axis=1))
    keep = NP.ones(length.size, dtype=bool)
    if parms['baseline']['min'] is not None:
        keep &= length >= parms['baseline']['min']
    if parms['baseline']['max'] is not None:
        keep &= length <= parms['baseline']['max']
    prefix = parms['telescope'].get('label_prefix') or ''
    all_labels = NP.array(['{0}{1:d}-{0}{2:d}'.format(prefix, int(a), int(b)) for a, b in ids])
    if arr.get('redundant', True):
        # array.redundant (default true) = look for redundancy and simulate one baseline per redundant group
        # (interferometry.py:1890-1891); the others are re-created at save time when save_redundant is set (:6823-6906)
        ubl, first, counts, occ = LAY.uniq_baselines(bl)
        order = NP.argsort(NP.sqrt(NP.sum(ubl ** 2, axis=1)), kind='mergesort')         # :1904-1905
        ubl, first, counts, occ = ubl[order], first[order], counts[order], [occ[i] for i in order]
    else:
        ubl, first, counts, occ = bl, NP.arange(bl.shape[0]), NP.ones(bl.shape[0], dtype=int), [[i] for i in range(bl.shape[0])]
    sel = keep[first]                                                                  # length selection, :1958
    ubl, first, counts, occ = ubl[sel], first[sel], counts[sel], [occ[i] for i in NP.flatnonzero(sel)]
    labels = all_labels[first].tolist()
    groups = {labels[i]: all_labels[NP.asarray(occ[i])].tolist() for i in range(len(labels))}
    return ubl, labels, pos, groups


def read_custom_catalog(path):
    with open(path) as f:
        lines = [ln for ln in f if ln.strip() and not ln.lstrip().startswith('#')]
    header = lines[0].split()
    data = NP.array([[float(x) for x in ln.split()] for ln in lines[1:]])
    return {name: data[:, i] for i, name in enumerate(header)}


def build_skymodel(parms, infile_dir):
    """Sky model in (RA, Dec) degrees.  'custom' follows run_prisim.py:1645-1684."""
    sp = parms['skyparm']
    model = sp['model']
    freq = parms['bandpass']['freq']
    fluxcut_freq = sp['fluxcut_reffreq'] if sp['fluxcut_reffreq'] is not None else freq         # run_prisim.py:902-903
    if model == 'custom':
        path = parms['catalog']['custom_file']
        if not os.path.isabs(path):
            path = os.path.join(infile_dir, path)
        cat = read_custom_catalog(path)
        ra, dec, fint, spindex = cat['RA'], cat['DEC'], cat['F_INT'], cat['SPINDEX']
        majax, minax = cat['MAJAX'], cat['MINAX']
        ref = sp['custom_reffreq'] * 1e9                                                         # :1655
        # the thresholds are given at fluxcut_freq and moved to the catalog frequency with EACH SOURCE's own catalog spectral
        # index: `spindex = catdata['SPINDEX'].data` is assigned (:1649) before the cut (:1657-1660)
        scale = (ref / fluxcut_freq) ** spindex
        sel = fint >= sp['flux_min'] * scale
        if sp['flux_max'] is not None:
            sel &= fint <= sp['flux_max'] * scale
        if NP.sum(sel) == 0:
            raise IndexError('No sources in the catalog found satisfying flux threshold criteria')
        return SM.SkyModel(location=NP.stack((ra[sel], dec[sel]), axis=1), flux_ref=fint[sel], spindex=spindex[sel], ref_freq=ref,
                           src_shape=NP.stack((majax[sel], minax[sel], NP.zeros(int(sel.sum()))), axis=1),
                           epoch='J' + str(sp['epoch']))
    lst0 = (parms['pointing']['lst_init'] or 0.0) * 15.0
    lat = parms['telescope']['latitude']

    def synthetic(spc):
        if spc['model'] == 'ptsrc_random':
            return W.point_source_sky(int(spc['n_src']), int(spc['seed']), f_ref=spc['custom_reffreq'] * 1e9, spindex=spc['spindex'])
        if spc['model'] == 'healpix_synthetic':
            return W.diffuse_sky(int(spc['nside']), int(spc['seed']), f_ref=spc['custom_reffreq'] * 1e9, spindex=spc['spindex'])
        raise NotImplementedError('skyparm.model {0!r}: survey catalogs need prisim/data (absent); use custom, ptsrc_random, '
                                  'healpix_synthetic or synthetic_mix'.format(spc['model']))
    if model == 'synthetic_mix':
        # several synthetic components in one sky, in the order given (the reference's combined models -- 'csm' = NVSS + SUMSS, 'asm' =
        # diffuse + point sources, run_prisim.py:1020-1686 -- need its catalogs): skyparm.components = [{model: ..., <keys of that
        # model>}, ...], every entry overlaid on skyparm.  Point sources first keeps the sky in runs of one source size each.
        comps = sp.get('components') or []
        if not comps:
            raise ValueError("skyparm.model 'synthetic_mix' needs a non-empty skyparm.components list")
        sky = W.concat_skies(*[synthetic(deep_merge({k: v for k, v in sp.items() if k != 'components'}, c)) for c in comps])
    else:
        sky = synthetic(sp)
    hadec = GEOM.altaz2hadec(sky['altaz'], lat, units='degrees')         # local frame at lst_init -> (RA, Dec)
    radec = NP.stack(((lst0 - hadec[:, 0]) % 360.0, hadec[:, 1]), axis=1)
    n = radec.shape[0]
    # (a synthetic sky is laid out in the LOCAL frame at lst_init: its (RA, Dec) are coordinates of date by construction -- epoch None,
    # nothing to precess; skyparm.epoch applies to catalogues that come with an equinox)
    return SM.SkyModel(location=radec, flux_ref=sky['flux_ref'], spindex=sky['spindex'], ref_freq=sky['ref_freq'],
                       src_shape=NP.stack((sky['fwhm_deg'], sky['fwhm_deg'], NP.zeros(n)), axis=1), epoch=None)


def load_external_beam(parms, infile_dir):
    """External beam file -> (beam [npix, nfreq], freqs_hz).  The reference reads FITS / HDF5 / UVBeam files
    (run_prisim.py:489-520).  Read here: the HDF5 layout its own converter writes (scripts/FEKO_beam_to_healpix.py:161-198:
    gain_info/<pol> = nfreq x npix, spectral_info/freqs in Hz; beam.pol picks the dataset, default the first one) through the
    HDF5 C library, and the same content as .npz: keys 'beam' (npix, nfreq) or 'gain_info' (nfreq, npix), and 'freqs' (Hz).
    FITS and UVBeam need astropy / pyuvdata: not offered."""
    bm = parms['beam']
    fmt = str(bm.get('filefmt', 'npz')).lower()
    if fmt not in ('hdf5', 'h5', 'npz'):
        raise NotImplementedError('beam.filefmt {0!r}: HDF5 (gain_info/<pol>) and npz can be read; FITS / UVBeam need astropy / pyuvdata'.format(bm.get('filefmt')))
    path = bm['file']
    if path is None:
        raise ValueError('beam.file must be given when beam.use_external is true')
    if not os.path.isabs(path):
        path = os.path.join(infile_dir, path)
    if fmt in ('hdf5', 'h5'):
        from . import hdf5io
        with hdf5io.File(path, 'r') as f:
            pols = f.list('gain_info')
            if not pols:
                raise KeyError('gain_info group of the external beam file holds no polarisation dataset')
            pol = bm.get('pol') or pols[0]
            if pol not in pols:
                raise KeyError('polarisation {0!r} not in the external beam file (has {1})'.format(pol, pols))
            beam = NP.asarray(f.read('gain_info/' + pol), dtype=NP.float64).T          # file: nfreq x npix (run_prisim.py:492-494)
            freqs = NP.asarray(f.read('spectral_info/freqs'), dtype=NP.float64)
        return NP.ascontiguousarray(beam), freqs
    with NP.load(path) as f:
        beam = f['beam'] if 'beam' in f.files else f['gain_info'].T
        freqs = f['freqs']
    return NP.asarray(beam, dtype=NP.float64), NP.asarray(freqs, dtype=NP.float64)


def window(nchan, shape):
    """Frequency window for the delay transform (DSP.windowing, astroutils: unpinned).  area-normalised to mean 1."""
    n = NP.arange(nchan)
    if nchan < 2:
        if shape not in ('rect', 'RECT', None, 'bhw', 'BHW', 'bnw', 'BNW'):
            raise ValueError('bpass_shape must be "rect", "bhw" or "bnw"')
        return NP.ones(nchan)                     # a one-channel band has nothing to taper (and the cosine terms would divide by zero)
    if shape in ('rect', 'RECT', None):
        w = NP.ones(nchan)
    elif shape in ('bhw', 'BHW'):
        a = (0.35875, 0.48829, 0.14128, 0.01168)
        x = 2 * NP.pi * n / (nchan - 1)
        w = a[0] - a[1] * NP.cos(x) + a[2] * NP.cos(2 * x) - a[3] * NP.cos(3 * x)
    elif shape in ('bnw', 'BNW'):
        a = (0.3635819, 0.4891775, 0.1365995, 0.0106411)
        x = 2 * NP.pi * n / (nchan - 1)
        w = a[0] - a[1] * NP.cos(x) + a[2] * NP.cos(2 * x) - a[3] * NP.cos(3 * x)
    else:
        raise ValueError('bpass_shape must be "rect", "bhw" or "bnw"')
    return w * nchan / NP.sum(w)


def julian_date(obs_date):
    y, m, d = [int(x) for x in str(obs_date).replace('-', '/').split('/')]
    return datetime.date(y, m, d).toordinal() + 1721424.5


def schedule(parms):
    """(jd, lst_deg, pointing_hadec) per accumulation (run_prisim.py:684-733)."""
    ob, pt = parms['obsparm'], parms['pointing']
    t_acc = float(ob['t_acc'])
    n_acc = int(ob['n_acc']) if ob.get('t_obs') is None else int(ob['t_obs'] / t_acc)
    mode = ob['obs_mode'] or 'track'
    if mode not in ('track', 'drift'):
        raise ValueError('Invalid specification for obs_mode')
    lst_init = (pt['lst_init'] or 0.0) * 15.0
    jd0 = pt['jd_init'] if pt.get('jd_init') is not None else julian_date(ob['obs_date'])
    t = NP.arange(n_acc) * t_acc
    lst = (lst_init + t * SIDEREAL_DEG_PER_SEC) % 360.0
    jd = jd0 + t / 86400.0
    lat = parms['telescope']['latitude']
    if mode == 'drift':
        di = pt['drift_init']
        if di.get('alt') is None or di.get('az') is None:
            if di.get('ha') is None or di.get('dec') is None:
                raise ValueError('One of alt-az or ha-dec pairs must be specified')
            hadec0 = NP.asarray([di['ha'], di['dec']], dtype=float)
        else:
            hadec0 = GEOM.altaz2hadec(NP.asarray([di['alt'], di['az']], dtype=float), lat, units='degrees')
        hadec = NP.repeat(hadec0.reshape(1, -1), n_acc, axis=0)
    else:
        ti = pt['track_init']
        ha0 = lst_init - ti['ra']
        hadec = NP.stack((ha0 + t * SIDEREAL_DEG_PER_SEC, ti['dec'] + NP.zeros(n_acc)), axis=1)
    return jd, lst, hadec, t_acc, n_acc


def run(parms, infile_dir='.', rank=0, world=1, device=0, comm_uid=None, verbose=True, host_copy='all', rdzv=None):
    """Simulate the observation described by `parms` on this rank's GPU.  Returns a dict with the (gathered) visibility
    cube (nbl, nchan, n_acc), baselines, labels, channels, lst, timestamps and timing.  host_copy (sharded runs): 'all' = every
    rank downloads the gathered cube and spectra, 'root' = rank 0 only (the others return None for them; every GPU still holds
    the gathered data unless pp.gather is 'root').  rdzv (sharded runs): the ranks' rendezvous -- the outcome of the communicator
    self-test is combined over it, so that every rank stops (SystemExit 3) when any rank's RCCL cannot move data."""
    if host_copy not in ('all', 'root'):
        raise ValueError("host_copy must be 'all' or 'root'")
    download = host_copy == 'all' or rank == 0
    extbeam = None
    if parms['beam'].get('use_external'):
        extbeam = load_external_beam(parms, infile_dir)
    pp_key = parms['pp'].get('key')
    if pp_key in ('freq', 'src'):
        # run_prisim.py:1858-1995 ('freq': chunks of channels, the shipped default, defaultparms.yaml:939) / :1996-2080 ('src': chunks of
        # sources): the key only chooses how the work is cut, never the result -- every chunk is a sub-block of the same sum.  On GPUs the
        # baselines are the natural shard axis (each shard keeps whole channel recurrences and sums every source), so the run proceeds on
        # baseline shards; the visibilities are identical by construction.
        if rank == 0:
            warnings.warn("pp.key = {0!r}: the work is cut over baselines instead (the partition key changes how the sum is split, not "
                          "its result)".format(pp_key))
    elif pp_key != 'bl':
        raise ValueError("pp.key must be 'bl', 'freq' or 'src' (scripts/run_prisim.py:1775-2080); got {0!r}".format(pp_key))
    bp = parms['bandpass']
    chans = W.channel_grid(float(bp['freq']), float(bp['freq_resolution']), int(bp['nchan']))      # run_prisim.py:900
    bl, labels, antpos, blgroups = baseline_info(parms)
    nbl_total = bl.shape[0]
    # :1775-1791 cuts contiguous chunks; here groups of baselines are dealt round-robin (prisim_amd/sharding.py): the list is sorted by
    # length and long baselines cost more (and are what the taper culling shortens), so every rank gets its share of them; shards are
    # padded to equal size for the all-gather and the gathered cubes are put back into the global order on the receiving GPU (shard map)
    bl_mine, idx_mine, n_real = SH.shard_rows(bl, world, rank)
    per = bl_mine.shape[0]
    labels_mine = [labels[i] for i in idx_mine] + ['pad'] * (per - n_real)
    idx_padded = NP.concatenate((idx_mine, NP.full(per - n_real, nbl_total - 1, dtype=NP.int64)))     # padding rows repeat the last baseline

    # every rank's rows of the padded shards in the global numbering (-1 = padding): with it the receiving GPU puts each gathered cube
    # into the reference's baseline order itself (prisim_hip_set_shard_map; rank-0 concatenate order of run_prisim.py:2233-2242)
    shard_map = NP.full((world, per), -1, dtype=NP.int64)
    for r in range(world):
        idx_r = SH.shard_index(nbl_total, world, r)
        shard_map[r, :idx_r.size] = idx_r

    def unshard(gathered):
        return gathered                         # already (nbl_total, ...) in global order
    tel = telescope_dict(parms)
    skymod = build_skymodel(parms, infile_dir)
    jd, lst, hadec, t_acc, n_acc = schedule(parms)
    if getattr(skymod, 'epoch', None) is not None:
        # "Precess Sky model to observing epoch" (scripts/run_prisim.py:1688-1692): once, to the first timestamp; every observe() then
        # applies what is left up to ITS timestamp plus nutation and aberration (interferometry.py:6174-6180, prisim_amd/frames.py).  The
        # two precessions compose exactly (both go through J2000), so the visibilities are those of one step from the catalogue's equinox.
        to_epoch = FRAMES.jyear_of_jd(jd[0])
        skymod.location = FRAMES.precess_radec(skymod.location, skymod.epoch, to_epoch)
        skymod.epoch = 'J{0:.12f}'.format(to_epoch)
    if hasattr(skymod, 'freeze'):
        skymod.freeze()                         # nothing edits the model from here on: observe() recognises it by identity (no content pass)
    proc = parms['processing']
    ia_kwargs = dict(telescope=tel, eff_Q=parms['telescope']['eff_Q'], latitude=tel['latitude'], longitude=tel['longitude'],
                     altitude=tel['altitude'], skycoords='radec', A_eff=parms['telescope']['A_eff'], pointing_coords='hadec', device=device,
                     blgroupinfo={'groups': blgroups, 'reversemap': {m: k for k, v in blgroups.items() for m in v}})
    ia = RI.InterferometerArray(labels_mine, bl_mine, chans, **ia_kwargs)
    # unsharded runs hand the cube to the host (files, the caller): each snapshot's download is queued under the next one's sky-sum;
    # sharded runs gather on the device and never copy their own shard
    ia.reserve(n_acc, host_staging=(world == 1))
    if world > 1:
        ia.set_shard_map(shard_map, nbl_total)
    if world > 1:
        if comm_uid is None:
            raise ValueError('comm_uid is needed when world > 1')
        # communicator + self-test BEFORE the snapshots: a RCCL / xGMI setup that cannot move data must stop the job here, not write a
        # wrong cube (prisim_hip_comm_selftest: 1 MiB all-gather of a rank-dependent pattern, verified on every rank's host)
        # comm_init is a collective: if it fails on this rank the exception ends the process (non-zero) and the launcher stops the peers,
        # which sit inside ncclCommInitRank -- voting on it over the rendezvous would hang them all.  Only the self-test,
        # which every rank reaches, is voted on.
        # All of it under one deadline (PRISIM_COMM_TIMEOUT_S, default 120 s): these calls wait for every rank, for ever when one never
        # arrives; on expiry the rank says who and where it is and exits non-zero (prisim_amd/watchdog.py), and the launcher stops the rest.
        with WATCHDOG.for_context(rank, device, _abi) as deadline:
            deadline.step('ncclCommInitRank (prisim_hip_comm_init)')
            ia.comm_setup(comm_uid, world, rank, selftest=False)
            ok, why = True, ''
            try:
                deadline.step('self-test all-gather (prisim_hip_comm_selftest)')
                ia.comm_selftest()
            except _abi.PrisimHipError as exc:
                ok, why = False, str(exc)
                if rdzv is None:
                    raise
            outcomes = None
            if rdzv is not None:
                deadline.step('rendezvous: exchange of the self-test outcomes')
                outcomes = rdzv.allgather([bool(ok), why])
        if outcomes is not None:
            bad = [(r, o[1]) for r, o in enumerate(outcomes) if not o[0]]
            if bad:
                if rank == 0 or not ok:
                    import sys
                    sys.stderr.write('RCCL communicator self-test failed on rank(s) {0}: {1}\n'.format([r for r, _ in bad], bad[0][1]))
                raise SystemExit(3)
    if extbeam is not None:
        bm = parms['beam']
        ia.set_external_beam(extbeam[0], extbeam[1], spec_interp=bm.get('spec_interp', 'cubic'), chromatic=bool(bm.get('chromatic', True)),
                             select_freq=bm.get('select_freq'))
    tp = parms['telescope']
    if tp.get('Tsys') is not None:
        tsysinfo = {'Tnet': float(tp['Tsys'])}
    else:
        tsysinfo = {'Trx': tp['Trx'], 'Tant': {'f0': tp['Tant_freqref'], 'T0': tp['Tant_ref'], 'spindex': tp['Tant_spindex']}, 'Tnet': None}
    roi_radius = parms['skyparm'].get('roi_radius')
    t0 = time.time()
    # run_prisim.py:2180-2198 loops observe() over the accumulations.  Here they go to the device in batches (observe_batch): the sky
    # model is uploaded once, every batch's geometry is formed on the GPU with one small read-back, and its snapshots are then queued
    # back to back -- no nsrc-sized host array is touched per snapshot and nothing synchronises the compute stream inside the loop.
    batch = max(1, int(proc.get('snapshot_batch') or 64))
    for j0 in range(0, n_acc, batch):
        j1 = min(n_acc, j0 + batch)
        ia.observe_batch([(float(jd[j]), float(lst[j])) for j in range(j0, j1)], tsysinfo, NP.ones(chans.size), hadec[j0:j1], skymod, t_acc,
                         roi_radius=roi_radius, roi_center='zenith', gradient_mode=proc.get('gradient_mode'), memsave=bool(proc.get('memsave')))
        if verbose and rank == 0:
            for j in range(j0, j1):
                print('snapshot {0}/{1}: lst = {2:.4f} deg, {3} sources'.format(j + 1, n_acc, lst[j], ia.obs_catalog_indices[j].size
                                                                                  if len(ia.obs_catalog_indices) > j else 0))
    t_sim = time.time() - t0
    # After the snapshots the reference adds thermal noise and re-centres the phases on phasing.center (run_prisim.py:2278-2282); the same
    # here through the class methods.  The noise stage makes host-side cubes of the size of the visibility cube (vis_noise_freq, vis_freq,
    # as in the reference): processing.add_noise = null (default) runs it up to 4 GiB per cube and says so when it skips, true / false
    # force it.  Both steps are per baseline: every rank does them on its own shard, before the exchange (the noise cube is gathered too).
    noise_done = False
    want_noise = proc.get('add_noise')
    cube_bytes = 16.0 * nbl_total * chans.size * n_acc
    if want_noise is None:
        want_noise = cube_bytes <= 4.0 * 2 ** 30
        if not want_noise and verbose and rank == 0:
            print('thermal noise left out: the cube is {0:.1f} GiB (processing.add_noise: true forces it)'.format(cube_bytes / 2 ** 30))
    if want_noise:
        # counter-based draws keyed on the GLOBAL baseline index: a shard draws exactly what the unsharded run draws for its baselines, given
        # the same seed -- processing.noise_seed, else one derived from the communicator id every rank holds (single process: a fresh one)
        seed = proc.get('noise_seed')
        if seed is None and world > 1:
            seed = int.from_bytes(hashlib.sha1(bytes(comm_uid or b'')).digest()[:8], 'little')
        ia.generate_noise(seed=seed, bl_index=idx_padded)
        ia.add_noise()
        noise_done = True
    ph = parms.get('phasing') or {}
    if ph.get('center') is not None:
        ref_point = {'coords': ph.get('coords', 'altaz'), 'location': NP.asarray(ph['center'], dtype=float).reshape(1, -1)}
        ia.rotate_visibilities(ref_point, do_delay_transform=False, verbose=False)
    if world > 1:
        # shards go GPU -> GPU; the host never sees this rank's own cube.  pp.gather: 'all' = every GPU ends up with the whole cube
        # (ncclAllGather), 'root' = only rank 0 does (ncclSend / ncclRecv; the other GPUs keep nothing: 120 GB less at config 5)
        gather = parms['pp'].get('gather')
        gather = ('root' if host_copy == 'root' else 'all') if gather is None else str(gather).lower()
        if gather not in ('all', 'root'):
            raise ValueError("pp.gather must be 'all' or 'root'")
        if gather == 'root':
            download = rank == 0
        cube = unshard(ia.allgather(comm_uid, world, rank, download=download, root=(0 if gather == 'root' else None)))
        labels_all, bl_all = labels, bl
        noise_all = None
        if noise_done:
            noise_all = unshard(ia.allgather_cube(ia.vis_noise_freq, world, download=download))
        grad_all = None
        if ia.gradient_mode is not None:
            # the gradient cubes are per baseline too: gathered like the visibilities (interferometry.py:8349-8350 concatenates them)
            g = ia.allgather_gradient(world, download=download)
            grad_all = {ia.gradient_mode: g} if g is not None else None       # (3, nbl_total, nchan, n_acc), global order
    else:
        cube, labels_all, bl_all = ia.skyvis_freq[:nbl_total], labels, bl
        grad_all = {k: v[:, :nbl_total] for k, v in ia.gradient.items()} if ia.gradient_mode is not None else None
    out = {'skyvis_freq': cube, 'bl': bl_all, 'labels': labels_all, 'freq': chans, 'lst': NP.asarray(ia.lst),
           'timestamp': NP.asarray(ia.timestamp), 'bl_length': NP.sqrt(NP.sum(bl_all ** 2, axis=1)), 't_sim': t_sim,
           'antpos': antpos, 'ia': ia, 'blgroups': blgroups, 'world': world, 'ia_kwargs': ia_kwargs}
    if ia.gradient_mode is not None:
        out['gradient_mode'], out['gradient'] = ia.gradient_mode, grad_all
    if noise_done and world == 1:
        out['vis_freq'], out['vis_noise_freq'] = ia.vis_freq[:nbl_total], ia.vis_noise_freq[:nbl_total]
    elif noise_done:
        out['vis_noise_freq'] = noise_all
        out['vis_freq'] = (cube + noise_all) if (cube is not None and noise_all is not None) else None
    if proc.get('delay_transform'):
        # every rank transforms its own shard on its GPU (the FFT runs along frequency); sharded runs then exchange the spectra
        ia.delay_transform(pad=float(proc.get('f_pad', 1.0)), freq_wts=window(chans.size, proc.get('bpass_shape', 'bhw')), verbose=False)
        if world > 1:
            out['skyvis_lag'] = unshard(ia.allgather_lags(world, download=download))
            if noise_done and ia.vis_lag is not None:
                if ia.vis_lag.shape == (bl_mine.shape[0], chans.size, n_acc):
                    # the spectra of the noisy and of the noise cube (host-side on every shard) travel like the noise cube did
                    for key, arr in (('vis_lag', ia.vis_lag), ('vis_noise_lag', ia.vis_noise_lag)):
                        out[key] = unshard(ia.allgather_cube(arr, world, download=download))
                else:
                    # nlag != nchan (a fractional f_pad): these host-side spectra cannot ride in the visibility slots.  Nothing is
                    # dropped: whoever assembles the whole array (assemble_full_array, rank 0) transforms the gathered vis_freq /
                    # vis_noise_freq cubes there.
                    out['noisy_lags_deferred'] = True
                    if verbose and rank == 0:
                        print('vis_lag / vis_noise_lag (nlag = {0} != nchan = {1}) are formed from the gathered cubes on rank 0'.format(
                            ia.vis_lag.shape[1], chans.size))
        else:
            out['skyvis_lag'] = ia.skyvis_lag
        out['lags'] = ia.lags
    return out


def assemble_full_array(out, parms):
    """The InterferometerArray of the WHOLE array from what rank 0 of a baseline-sharded run holds: its own shard object (everything that
    does not depend on the baseline) and the gathered cubes.  Stands where the reference concatenates the per-rank part files
    (run_prisim.py:2233-2242)."""
    if out.get('skyvis_freq') is None:
        raise ValueError('this rank did not download the gathered cube (host_copy)')
    shard = out['ia']
    full = RI.InterferometerArray(out['labels'], out['bl'], out['freq'], **out['ia_kwargs'])
    full.adopt_observation(shard)
    full.skyvis_freq = out['skyvis_freq']
    if shard.gradient_mode is not None:
        if out.get('gradient') is None:
            raise ValueError('this rank did not download the gathered gradient cube (host_copy)')
        full.gradient_mode, full.gradient = shard.gradient_mode, dict(out['gradient'])
    if out.get('vis_freq') is not None:
        full.vis_freq, full.vis_noise_freq = out['vis_freq'], out['vis_noise_freq']
    if out.get('skyvis_lag') is not None:
        full.skyvis_lag = out['skyvis_lag']
    if out.get('vis_lag') is not None:
        full.vis_lag, full.vis_noise_lag = out['vis_lag'], out['vis_noise_lag']
    proc = parms.get('processing') or {}
    if out.get('skyvis_lag') is not None:
        if out.get('noisy_lags_deferred') and out.get('vis_freq') is not None:
            # the noisy cubes' spectra could not be exchanged (nlag != nchan): transform the gathered cubes here, on this rank's GPU
            lag_gathered = full.skyvis_lag
            full.delay_transform(pad=float(proc.get('f_pad', 1.0)), freq_wts=window(len(out['freq']), proc.get('bpass_shape', 'bhw')), verbose=False)
            full.skyvis_lag = lag_gathered
        elif shard.lag_kernel is not None:
            # lag_kernel = transform of bp * bp_wts (:8119): with one window for every baseline a shard's rows all equal the whole array's
            kern = NP.asarray(shard.lag_kernel)
            if kern.shape[0] >= 1 and NP.array_equal(kern, NP.broadcast_to(kern[:1], kern.shape)):
                full.lag_kernel = NP.array(NP.broadcast_to(kern[:1], (full.baselines.shape[0],) + kern.shape[1:]))
    ph = parms.get('phasing') or {}
    if ph.get('center') is not None:                  # the shards were re-centred there before the exchange: projected baselines to match
        full.project_baselines({'coords': ph.get('coords', 'altaz'), 'location': NP.asarray(ph['center'], dtype=float).reshape(1, -1)})
    return full


def save(out, parms, infile=None):
    ds = parms['dirstruct']
    simid = ds['simid'] or time.strftime('%Y-%m-%d-%H-%M-%S')
    outdir = os.path.join(ds['rootdir'], ds['project'], simid, 'simdata')
    os.makedirs(outdir, exist_ok=True)
    path = os.path.join(outdir, 'simvis')
    out_hdf5 = out                                    # the HDF5 path re-creates the redundant baselines itself (duplicate_measurements)
    if parms.get('save_redundant', True) and out.get('blgroups'):
        # re-create the redundant baselines from the simulated unique ones (run_prisim.py:2325-2326, interferometry.py:6889-6895)
        counts = [len(out['blgroups'].get(lbl, [lbl])) for lbl in out['labels']]
        if any(c > 1 for c in counts):
            out = dict(out)
            out['labels'] = [m for lbl in out['labels'] for m in out['blgroups'].get(lbl, [lbl])]
            for key, axis in (('skyvis_freq', 0), ('vis_freq', 0), ('vis_noise_freq', 0), ('bl', 0), ('bl_length', 0), ('skyvis_lag', 0)):
                if out.get(key) is not None:
                    out[key] = NP.repeat(out[key], counts, axis=axis)
    if parms['save_formats'].get('npz', True):
        keys = {k: out[k] for k in ('skyvis_freq', 'lst', 'freq', 'timestamp', 'bl', 'bl_length')}          # interferometry.py:8862
        keys['labels'] = NP.asarray(out['labels'])
        for extra in ('vis_freq', 'vis_noise_freq', 'skyvis_lag', 'lags'):                                   # :8860-8861 when noise was added
            if out.get(extra) is not None:
                keys[extra] = out[extra]
        # (compressed like the reference's, interferometry.py:8859-8863; save_formats.npz_compress: false writes the arrays as they are --
        # visibility cubes are noise-like and gigabytes of them deflate slowly)
        (NP.savez_compressed if parms['save_formats'].get('npz_compress', True) else NP.savez)(path + '.npz', **keys)
    if parms['save_formats'].get('hdf5', False) and out.get('ia') is not None:
        # PRISim's HDF5 layout (interferometry.py:8717-8846) of the InterferometerArray, redundant baselines re-created first when asked
        # for (run_prisim.py:2325-2326).
        # a baseline-sharded run first puts the whole array together from rank 0's shard object and the gathered cubes
        ia = assemble_full_array(out_hdf5, parms) if out.get('world', 1) > 1 else out['ia']
        if parms.get('save_redundant', True) and ia.blgroups and len(ia.labels) < sum(len(v) for v in ia.blgroups.values()):
            ia.duplicate_measurements()
        ia.save(path, fmt='HDF5', npz=False, overwrite=True, verbose=False)
    metadir = os.path.join(ds['rootdir'], ds['project'], simid, 'metainfo')
    os.makedirs(metadir, exist_ok=True)
    with open(os.path.join(metadir, 'simparms.yaml'), 'w') as f:                                            # run_prisim.py:2213-2220
        yaml.safe_dump(parms, f, default_flow_style=False)
    return path + '.npz'


def main(argv=None):
    import argparse
    parser = argparse.ArgumentParser(description='Program to simulate interferometer array data (MI355X path)')
    parser.add_argument('-i', '--infile', dest='infile', required=True, type=str, help='File specifying input parameters')
    parser.add_argument('-n', '--nranks', dest='nranks', type=int, default=1,
                        help='number of GPUs: starts that many rank processes itself (the `mpirun -n N` of README.rst:93-99; no torch, no MPI)')
    args = parser.parse_args(argv)
    if 'WORLD_SIZE' not in os.environ and args.nranks > 1:
        # become the launcher before anything touches the GPU: N children run this same entry with RANK / WORLD_SIZE set
        import sys
        from . import launch
        if argv is None:
            cmd = [sys.executable, os.path.abspath(sys.argv[0])] + list(sys.argv[1:])
        else:
            # called programmatically (a host program, a test): the ranks run this module, not whatever sys.argv[0] happens to be
            cmd = [sys.executable, '-m', 'prisim_amd.driver'] + list(argv)
        pkg_parent = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        env = dict(os.environ)
        env['PYTHONPATH'] = pkg_parent + (os.pathsep + env['PYTHONPATH'] if env.get('PYTHONPATH') else '')
        return launch.spawn_ranks(args.nranks, cmd, env=env)
    parms = load_parms(args.infile)
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    uid = None
    from . import rendezvous
    rdzv = rendezvous.Rendezvous(rank, world)      # loopback sockets; no torch, nothing touches the GPU before this returns
    if world > 1:
        os.environ.setdefault('NCCL_SOCKET_IFNAME', 'lo')
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')       # (as prisim_amd.launch sets it: ranks started by another launcher need it too)
        with WATCHDOG.CommDeadline(rank, local_rank) as deadline:
            deadline.step('rendezvous: broadcast of the RCCL unique id')
            uid = rdzv.broadcast_bytes(_abi.Context.comm_unique_id() if rank == 0 else b'')
    device = int(os.environ.get('PRISIM_DEVICE', local_rank))       # PRISIM_DEVICE: rehearsal hook (several ranks on the one GPU of a test box)
    out = run(parms, infile_dir=os.path.dirname(os.path.abspath(args.infile)), rank=rank, world=world, device=device, comm_uid=uid,
              host_copy='root', rdzv=rdzv)         # only rank 0 writes: by default (pp.gather: null) only its GPU receives the cube
    if rank == 0:
        path = save(out, parms, args.infile)
        print('simulated {0} baselines x {1} channels x {2} snapshots in {3:.3f} s -> {4}'.format(
            out['skyvis_freq'].shape[0], out['skyvis_freq'].shape[1], out['skyvis_freq'].shape[2], out['t_sim'], path))
    rdzv.barrier()
    rdzv.close()
    return 0


if __name__ == '__main__':
    import sys
    sys.exit(main())
