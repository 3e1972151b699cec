"""Compile skyvis_kernels.hip to gfx950 assembly and print, per k_skyvis_rec* kernel, the register budget the compiler ended up
with (VGPRs, SGPRs, spills, scratch) and a static instruction census of the kernel body.  Runs without a GPU.

  python tools/kernel_meta.py [-DNAME=VALUE ...] [--filter substr] [--keep out.s]
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, 'prisim_amd', 'csrc', 'skyvis_kernels.hip')


def demangle(names):
    try:
        out = subprocess.run(['/opt/rocm/lib/llvm/bin/llvm-cxxfilt'] + names, capture_output=True, text=True).stdout.split('\n')
        return dict(zip(names, out))
    except Exception:
        return {n: n for n in names}


def compile_asm(defs=(), keep=None):
    out = keep or os.path.join(tempfile.mkdtemp(), 'k.s')
    cmd = ['/opt/rocm/bin/hipcc', '-O3', '-std=c++17', '--offload-arch=gfx950', '-I/opt/rocm/include', '-S', '--cuda-device-only',
           SRC, '-o', out] + list(defs)
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        sys.exit(res.stderr)
    with open(out) as f:
        return f.read()


def kernel_meta(text):
    meta = text[text.index('amdhsa.kernels:'):]
    rows = []
    for block in meta.split('- .agpr_count:')[1:]:
        def num(key):
            m = re.search(r'\.%s:\s+(\d+)' % key, block)
            return int(m.group(1)) if m else -1
        rows.append({'name': re.search(r'\.name:\s+(\S+)', block).group(1), 'vgpr': num('vgpr_count'), 'sgpr': num('sgpr_count'),
                     'vgpr_spill': num('vgpr_spill_count'), 'sgpr_spill': num('sgpr_spill_count'),
                     'scratch': num('private_segment_fixed_size'), 'lds': num('group_segment_fixed_size')})
    return rows


def kernel_body(text, name):
    start = text.index('\n' + name + ':')
    end = text.index('.end_amdhsa_kernel', start) if '.end_amdhsa_kernel' in text[start:] else len(text)
    end2 = text.find('\n\t.section', start)
    return text[start:min(end, end2 if end2 > 0 else end)]


def loops(body):
    """(first line, last line) of every backward branch, largest span first: the source loops of the kernel bodies."""
    lines = body.split('\n')
    labels = {}
    for i, ln in enumerate(lines):
        m = re.match(r'(\.LBB\d+_\d+):', ln)
        if m:
            labels[m.group(1)] = i
    out = []
    for i, ln in enumerate(lines):
        m = re.match(r'\s+s_c?branch\S*\s+(\.LBB\d+_\d+)', ln)
        if m and m.group(1) in labels and labels[m.group(1)] < i:
            out.append((labels[m.group(1)], i))
    out.sort(key=lambda ab: ab[0] - ab[1])
    return lines, out


def census(body):
    """Instruction counts by class."""
    cls = {}
    for line in body.split('\n'):
        line = line.strip()
        m = re.match(r'([sv]_[a-z0-9_]+|ds_[a-z0-9_]+|global_[a-z0-9_]+|buffer_[a-z0-9_]+|scratch_[a-z0-9_]+)', line)
        if not m:
            continue
        op = m.group(1)
        key = ('v_pk' if op.startswith('v_pk_') else 'v_f64' if (op.startswith('v_') and '_f64' in op) else
               'v_lane' if op in ('v_readlane_b32', 'v_writelane_b32', 'v_readfirstlane_b32') else
               'v_trans' if re.match(r'v_(exp|log|sin|cos|rcp|rsq|sqrt)_', op) else
               'v_other' if op.startswith('v_') else 's_load' if op.startswith('s_load') else 's_other' if op.startswith('s_') else
               'lds' if op.startswith('ds_') else 'vmem')
        cls[key] = cls.get(key, 0) + 1
    return cls


def main():
    defs = [a for a in sys.argv[1:] if a.startswith('-D')]
    flt = None
    keep = None
    args = sys.argv[1:]
    if '--filter' in args:
        flt = args[args.index('--filter') + 1]
    if '--keep' in args:
        keep = args[args.index('--keep') + 1]
    nloops = int(args[args.index('--loops') + 1]) if '--loops' in args else 0
    text = compile_asm(defs, keep)
    rows = [r for r in kernel_meta(text) if 'k_skyvis_' in r['name'] and 'direct' not in r['name']]
    dm = demangle([r['name'] for r in rows])
    for r in rows:
        pretty = dm[r['name']].replace('prisim::', '').replace('(prisim::SkyvisParams)', '')
        if flt and flt not in pretty:
            continue
        body = kernel_body(text, r['name'])
        c = census(body)
        print('%-52s vgpr %3d sgpr %3d  spill v %3d s %3d  scratch %4d | %s' % (
            pretty, r['vgpr'], r['sgpr'], r['vgpr_spill'], r['sgpr_spill'], r['scratch'],
            ' '.join('%s=%d' % kv for kv in sorted(c.items()))))
        lines, lp = loops(body)
        for a, b in (lp if nloops else []):
            if b - a < 60:
                continue
            cc = census('\n'.join(lines[a:b + 1]))
            if cc.get('lds', 0) > 4 or not cc.get('s_load', 0):
                continue                      # the flush / segment loops: only the source loops are of interest
            valu = sum(v for k, v in cc.items() if k.startswith('v_'))
            print('      loop lines %5d-%5d: VALU %4d | %s' % (a, b, valu, ' '.join('%s=%d' % kv for kv in sorted(cc.items()))))


if __name__ == '__main__':
    main()
