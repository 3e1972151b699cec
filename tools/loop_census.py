"""Static census of the basic blocks of one kernel of skyvis_kernels.hip that hold matrix instructions (or, with --all, of every block):
instructions per class.  Used for DESIGN 4.2's closure of k_skyvis_grad_taper_f64: the loop body of one group (4 sources x 16 baselines x
32 channels = 32 wave-terms) is 64 v_mfma_f64_4x4x4 + 262 other VALU instructions; at 16 datapath cycles per MFMA and 4 per fp64 VALU
instruction on the shared datapath that is 76 cycles per wave-term with the seed, against the contract's 32.

  python tools/loop_census.py grad_taper_f64 [--all]      (compiles to assembly first; no GPU needed)"""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def cls(op):
    if op.startswith('v_mfma'):
        return 'mfma'
    if op.startswith('v_fma_f64') or op.startswith('v_fmac_f64'):
        return 'v_fma_f64'
    if op.startswith('v_mul_f64'):
        return 'v_mul_f64'
    if op.startswith('v_add_f64'):
        return 'v_add_f64'
    if op.startswith('v_pk_'):
        return 'v_pk'
    if op.startswith('v_'):
        return 'v_other'
    if op.startswith('ds_'):
        return 'lds'
    if op.startswith('s_waitcnt'):
        return 's_waitcnt'
    if op.startswith('s_load') or op.startswith('s_buffer_load'):
        return 's_load'
    if op.startswith('s_'):
        return 'salu'
    if op.startswith(('global_', 'buffer_', 'flat_', 'scratch_')):
        return 'vmem'
    return 'other'


def main():
    want = sys.argv[1] if len(sys.argv) > 1 else 'grad_taper_f64'
    show_all = '--all' in sys.argv
    out = os.path.join(tempfile.mkdtemp(), 'k.s')
    subprocess.check_call(['/opt/rocm/bin/hipcc', '-O3', '-std=c++17', '--offload-arch=gfx950', '-I/opt/rocm/include', '-S', '--cuda-device-only',
                           os.path.join(ROOT, 'prisim_amd', 'csrc', 'skyvis_kernels.hip'), '-o', out])
    text = open(out).read()
    names = [m.group(1) for m in re.finditer(r'^(_ZN6prisim\w*%s\w*):' % re.escape(want), text, flags=re.M)]
    for name in names:
        i = text.index(name + ':')
        body = text[i:text.index('.Lfunc_end', i)].split('\n')
        blocks, cur = [], ('entry', [])
        for ln in body:
            s = ln.strip()
            m = re.match(r'^(\.LBB\d+_\d+):', s)
            if m:
                blocks.append(cur)
                cur = (m.group(1), [])
                continue
            if not s or s.startswith(';') or s.startswith('.') or s.endswith(':'):
                continue
            cur[1].append(s.split(';')[0].strip())
        blocks.append(cur)
        print(name)
        for lab, ins in blocks:
            c = collections.Counter(cls(x.split()[0]) for x in ins if x)
            if show_all or c.get('mfma', 0) > 0:
                print('  %-10s %4d instructions  %s' % (lab, len(ins), dict(sorted(c.items()))))


if __name__ == '__main__':
    main()
