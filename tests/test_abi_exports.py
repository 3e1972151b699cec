"""CPU: the C-ABI library loads and exports every symbol include/prisim_hip.h declares; the ctypes
structures match the header; the product path fails loudly without a GPU (no CPU fallback)."""
import ctypes as C
import os
import sys
import re

import pytest

from prisim_amd import _abi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, 'include', 'prisim_hip.h')


def _declared_symbols():
    txt = open(HEADER).read()
    txt = re.sub(r'/\*.*?\*/', '', txt, flags=re.S)
    return sorted(set(re.findall(r'\b(prisim_hip_\w+)\s*\(', txt)))


def test_library_exports_every_declared_symbol():
    lib = _abi.load_library()
    names = _declared_symbols()
    assert len(names) >= 20
    for name in names:
        assert hasattr(lib, name), 'libprisim_hip.so does not export ' + name
    assert sorted(_abi.EXPORTS) == names


def test_version_and_error_string_without_context():
    lib = _abi.load_library()
    assert lib.prisim_hip_version().decode().startswith('prisim_hip ')
    assert lib.prisim_hip_version().decode() == _abi.ABI_VERSION            # the binding refuses any other library (stale build)
    assert isinstance(lib.prisim_hip_last_error(None), bytes)


def test_struct_layouts_match_header():
    # prisim_sky: int64, 2 ptr, int32 (+pad), 3 ptr  -> 56 bytes on LP64
    assert C.sizeof(_abi.PrisimSky) == 56
    assert _abi.PrisimSky.pbflux_is_f32.offset == 24 and _abi.PrisimSky.fluxes.offset == 48
    # prisim_beam_sky: int64, 4 ptr, double, int32 (+pad), double, 4 ptr -> 96 bytes
    assert C.sizeof(_abi.PrisimBeamSky) == 96 and _abi.PrisimBeamSky.ext.offset == 88
    # prisim_beam_ext: 3 double, 4 int32, 3 double, 3 double, 3 double, 2 int32, 3 pointers, 4 double -> 176 bytes
    assert C.sizeof(_abi.PrisimBeamExt) == 176 and _abi.PrisimBeamExt.array_sep1.offset == 40 and _abi.PrisimBeamExt.bf_pos.offset == 120
    assert _abi.PrisimBeamSky.beam_kind.offset == 48 and _abi.PrisimBeamSky.diameter_m.offset == 56
    # prisim_timing: 3 double, 2 int64, 6 int32, 1 double, 2 int32, 1 double, 2 int32 -> 96 bytes
    assert C.sizeof(_abi.PrisimTiming) == 96 and _abi.PrisimTiming.last_batch_snapshots.offset == 88 and _abi.PrisimTiming.last_delay_ms.offset == 64 and _abi.PrisimTiming.last_taper_split.offset == 72
    assert _abi.PrisimTiming.last_culled_fraction.offset == 80
    # prisim_comm_stats: 2 int64, 4 double, 4 int32, 2 double -> 80 bytes
    assert C.sizeof(_abi.PrisimCommStats) == 80 and _abi.PrisimCommStats.stream_priority.offset == 48 and _abi.PrisimCommStats.sum_undeal_ms.offset == 64
    # prisim_catalog: int64, 2 int32, 3 ptr, double, 3 ptr -> 72 bytes; prisim_obs: 2 double, 4 int32, double, ptr -> 48;
    # prisim_snapshot: 7 double, 2 int32, 12 double -> 160
    assert C.sizeof(_abi.PrisimCatalog) == 72 and _abi.PrisimCatalog.location.offset == 16 and _abi.PrisimCatalog.unitvec.offset == 64
    assert C.sizeof(_abi.PrisimObs) == 48 and _abi.PrisimObs.beam_kind.offset == 24 and _abi.PrisimObs.ext.offset == 40
    assert C.sizeof(_abi.PrisimSnapshot) == 160 and _abi.PrisimSnapshot.frame_given.offset == 56 and _abi.PrisimSnapshot.cel2enu.offset == 64


def test_struct_layouts_against_the_compiled_header(tmp_path):
    """The same by construction: include/prisim_hip.h compiled by gcc reports sizeof / offsetof of every field of every struct, and the
    ctypes mirrors must agree field by field."""
    import subprocess
    pairs = {'prisim_sky': _abi.PrisimSky, 'prisim_beam_ext': _abi.PrisimBeamExt, 'prisim_beam_sky': _abi.PrisimBeamSky,
             'prisim_catalog': _abi.PrisimCatalog, 'prisim_obs': _abi.PrisimObs, 'prisim_snapshot': _abi.PrisimSnapshot,
             'prisim_post': _abi.PrisimPost, 'prisim_timing': _abi.PrisimTiming, 'prisim_comm_stats': _abi.PrisimCommStats}
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "prisim_hip.h"', 'int main(void) {']
    for cname, cls in pairs.items():
        lines.append('  printf("%s %zu\\n", "{0}", sizeof({0}));'.format(cname))
        for fname, _ in cls._fields_:
            lines.append('  printf("%s.%s %zu\\n", "{0}", "{1}", offsetof({0}, {1}));'.format(cname, fname))
    lines += ['  return 0;', '}']
    src = tmp_path / 'layout.c'
    src.write_text('\n'.join(lines))
    exe = tmp_path / 'layout'
    subprocess.check_call(['gcc', '-I', os.path.join(ROOT, 'include'), str(src), '-o', str(exe)])
    got = dict(ln.split() for ln in subprocess.check_output([str(exe)]).decode().splitlines())
    for cname, cls in pairs.items():
        assert int(got[cname]) == C.sizeof(cls), cname
        for fname, _ in cls._fields_:
            assert int(got[cname + '.' + fname]) == getattr(cls, fname).offset, (cname, fname)


def test_every_export_is_guarded_against_cpp_exceptions():
    """no C++ exception crosses the ABI.  Every `int prisim_hip_*` entry of capi.cpp runs its body through guarded()
    (and of catalog.cpp) (bad_alloc -> PRISIM_ENOMEM, anything else -> PRISIM_EINTERNAL); the void / const char* entries cannot throw
    by construction."""
    src = ''.join(open(os.path.join(ROOT, 'prisim_amd', 'csrc', f)).read() for f in ('capi.cpp', 'catalog.cpp'))
    entries = re.findall(r'^int (prisim_hip_\w+)\(', src, flags=re.M)
    assert sorted(entries) == sorted(n for n in _abi.EXPORTS if n not in ('prisim_hip_destroy', 'prisim_hip_last_error', 'prisim_hip_version'))
    for name in entries:
        if name == 'prisim_hip_comm_last_error':
            # a watchdog thread calls it while another thread is inside RCCL: one snprintf into the caller's buffer, no allocation, no
            # context, no HIP call -- nothing in it can throw, and it must not wait for anything
            body = src[src.index('int %s(' % name):]
            body = body[:body.index('\n}\n')]
            assert 'snprintf' in body and 'std::string' not in body and 'hip' not in body.replace('prisim_hip', '')
            continue
        body = src[src.index('int %s(' % name):]
        head = body[:body.index('{') + 200]
        assert 'return guarded(' in head.split('\n', 3)[-1] or 'return guarded(' in head, name
    # host allocations of caller-sized vectors run inside the guard: a failed one comes back as a code, e.g. from host_alloc
    lib = _abi.load_library()
    p = C.c_void_p()
    assert lib.prisim_hip_host_alloc(-5, C.byref(p)) == _abi.PRISIM_EINVAL and not p.value


def test_null_context_is_rejected_not_crashing():
    lib = _abi.load_library()
    assert lib.prisim_hip_sync(None) == _abi.PRISIM_EINVAL
    assert lib.prisim_hip_compute(None, 0, 0, 0, 0) == _abi.PRISIM_EINVAL
    assert lib.prisim_hip_create(0, None) == _abi.PRISIM_EINVAL


def test_no_silent_cpu_fallback():
    """Without a GPU, creating a context raises PrisimHipError (on a GPU box it simply succeeds)."""
    try:
        ctx = _abi.Context(0)
    except _abi.PrisimHipError as exc:
        assert 'no HIP device' in str(exc) or 'HIP' in str(exc)
    else:
        ctx.close()


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    monkeypatch.setattr(_abi, '_lib', None)
    monkeypatch.setattr(_abi, 'LIB_PATH', str(tmp_path / 'nope.so'))
    with pytest.raises(_abi.PrisimHipError):
        _abi.load_library()


def test_sky_sum_kernels_use_no_scratch(tmp_path):
    """Build guard: every k_skyvis_rec* kernel must keep its accumulators in registers (no private segment).  A build of the packed
    taper kernel that spilled inside the source loop (3 waves/SIMD, 168 VGPRs) produced NaNs on partly filled wavefronts; the
    kernels are built for occupancies at which the compiler does not spill at all, and this test keeps it that way."""
    import re
    import shutil
    import subprocess
    hipcc = shutil.which('hipcc') or '/opt/rocm/bin/hipcc'
    if not os.path.exists(hipcc):
        pytest.skip('hipcc not available')
    src = os.path.join(ROOT, 'prisim_amd', 'csrc', 'skyvis_kernels.hip')
    out = tmp_path / 'skyvis_kernels.s'
    res = subprocess.run([hipcc, '-O3', '-std=c++17', '--offload-arch=gfx950', '-I/opt/rocm/include', '-S', '--cuda-device-only', src,
                          '-o', str(out)], capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr
    text = out.read_text()
    sys.path.insert(0, os.path.join(ROOT, 'tools'))
    import kernel_meta as KM
    found = 0
    for row in KM.kernel_meta(text):
        name = row['name']
        if 'k_skyvis_' not in name or 'direct' in name:
            continue
        found += 1
        assert row['scratch'] == 0, name
        assert row['vgpr_spill'] == 0, name
        if 'grad_f64' in name or 'grad_taper_f64' in name:
            continue                            # the MFMA kernels stream rows through vector loads: no scalar-row source loop to inspect
        # SGPR spills (v_readlane / v_writelane through a spare VGPR) are tolerated in the prologue and around the flush, never in
        # the source loops: round 1's fp64 kernels moved 103-134 SGPRs per source through lanes because their half-row operand
        # buffers (2 x 32 SGPRs) did not fit
        body = KM.kernel_body(text, name)
        lines, loops = KM.loops(body)
        src_loops = [(a, b) for a, b in loops
                     if (lambda c: c.get('lds', 0) <= 4 and c.get('s_load', 0) and (c.get('v_pk', 0) + c.get('v_f64', 0) + c.get('v_other', 0)) > 60)(
                         KM.census('\n'.join(lines[a:b + 1])))]
        assert src_loops, name
        # the compiler lays the every-4th-source prefetch branch out as overlapping variants of one loop: judge each cluster of
        # overlapping loops by its tightest member (the path 3 of 4 sources take); the widest may hold the branch's few lane moves
        clusters = []
        for a, b in sorted(src_loops):
            if clusters and a <= clusters[-1][-1][1]:
                clusters[-1].append((a, b))
            else:
                clusters.append([(a, b)])
        for cl in clusters:
            a, b = min(cl, key=lambda ab: ab[1] - ab[0])
            assert KM.census('\n'.join(lines[a:b + 1])).get('v_lane', 0) <= 1, (name, a, b)
    assert found >= 19


def test_binding_refuses_a_library_of_another_abi_version(monkeypatch):
    """A stale libprisim_hip.so beside a newer binding (or the reverse) would have its structs misread: loading must fail loudly."""
    monkeypatch.setattr(_abi, '_lib', None)
    monkeypatch.setattr(_abi, 'ABI_VERSION', 'prisim_hip 9.9 gfx950')
    with pytest.raises(_abi.PrisimHipError, match='rebuild'):
        _abi.load_library()
    monkeypatch.undo()
    assert _abi.load_library() is not None
