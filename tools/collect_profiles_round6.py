"""Copy the summaries tools/profile_all_round6.sh left under gpurun_out/prof_r06_*/ into profiles/r06_*/, write profiles/r06_MANIFEST.json
(every directory must carry the hash of the sources that are in the tree) and print the figures DESIGN.md 4.0 quotes.
    python tools/collect_profiles_round6.py <manifest of the first call> [<manifest of the second call> ...]"""
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench

dirs, what = [], None
for m in sys.argv[1:]:
    d = json.load(open(m))
    assert d['csrc_hash'] == bench.csrc_hash(), (m, d['csrc_hash'], bench.csrc_hash())
    dirs += d['dirs']
    what = d['what']
for d in dirs:
    src, dst = os.path.join(ROOT, 'gpurun_out', 'prof_' + d), os.path.join(ROOT, 'profiles', d)
    os.makedirs(dst, exist_ok=True)
    for f in ('kernel_stats.csv', 'pmc_summary.json', 'bench_under_rocprof.json'):
        shutil.copy(os.path.join(src, f), os.path.join(dst, f))
    s = json.load(open(os.path.join(dst, 'pmc_summary.json')))
    assert s['_csrc_hash'] == bench.csrc_hash(), d
    der = s['_derived']
    # the record of the profiled launch (tools/summarize_pmc.py: a batched call's own line, not the per-snapshot chain printed after it)
    recs = [json.loads(l) for l in open(os.path.join(dst, 'bench_under_rocprof.json')).read().splitlines() if l.startswith('{')]
    rec = ([r for r in recs if str(r.get('mode', '')).startswith('batch')] or recs)[-1] if recs else None
    if rec and 'roofline' in rec and rec['roofline'].get('terms_per_launch'):
        der['algorithmic_bytes_per_launch'] = rec['roofline_hbm']['algorithmic_bytes_per_launch']
        der['valu_wave_instructions_per_wave_term'] = s['SQ_INSTS_VALU']['mean_per_launch'] / (rec['roofline']['terms_per_launch'] / 64.0)
        der['hipEvent_avg_kernel_ms_under_rocprof'] = rec['roofline']['avg_kernel_ms']
        json.dump(s, open(os.path.join(dst, 'pmc_summary.json'), 'w'), indent=1)
    ms = der['avg_kernel_ms (rocprofv3 --kernel-trace --stats)']
    clock = der['clock_GHz (GRBM_GUI_ACTIVE/8/avg kernel time)']
    hbm = der['hbm_bytes_per_launch (2*FETCH_SIZE + WRITE_SIZE, KiB->B, gfx950 FETCH correction)']
    alg = der['algorithmic_bytes_per_launch']
    busy = s['SQ_INSTS_VALU']['mean_per_launch'] * 4 / (1024 * ms * 1e-3 * clock * 1e9)
    wait = s['SQ_WAIT_ANY']['mean_per_launch'] / s['SQ_WAVE_CYCLES']['mean_per_launch']
    terms = None
    if rec:
        terms = rec.get('roofline', {}).get('terms_per_launch')
    print('%-26s ms %.5g  clock %.3f  valu/wave-term %.3f  hbm %.4g B = %.2f x alg  wait %.3f  valu-issue %.3f' % (
        d, ms, clock, der['valu_wave_instructions_per_wave_term'], hbm, hbm / alg, wait, busy), end='')
    if terms:
        print('  terms/s %.4g  10-flop frac f32 %.3f f64 %.3f' % (terms / (ms * 1e-3), terms * 10 / (ms * 1e-3) / 157.3e12, terms * 10 / (ms * 1e-3) / 78.6e12), end='')
    mf = s.get('SQ_VALU_MFMA_BUSY_CYCLES', {}).get('mean_per_launch')
    if mf:
        print('  mfma-busy %.3f of SIMD cycles' % (mf / (1024 * ms * 1e-3 * clock * 1e9)), end='')
    print()
json.dump({'csrc_hash': bench.csrc_hash(), 'dirs': dirs, 'what': what}, open(os.path.join(ROOT, 'profiles', 'r06_MANIFEST.json'), 'w'), indent=1)
print('manifest written:', len(dirs), 'directories')
