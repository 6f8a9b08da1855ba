#!/usr/bin/env python3
"""Per-kernel resources of a hipcc object file or shared library (gfx950 code object): VGPRs, SGPRs, scratch bytes, spilled VGPRs / SGPRs, LDS.
Usage: python tools/kernel_resources.py dualsuperreslearningforsemseg_amd/csrc/conv_sk.o [--scratch-only] [--grep SUBSTR]
Used by the spill guard in tests/test_abi_and_host.py."""
import os
import re
import shutil
import subprocess
import sys
import tempfile

LLVM = '/opt/rocm/lib/llvm/bin'


def kernels(path):
    """[{name, vgpr, sgpr, scratch, vgpr_spill, sgpr_spill, lds}] of the gfx950 code object bundled in `path`."""
    with tempfile.TemporaryDirectory() as tmp:
        local = os.path.join(tmp, 'obj')
        shutil.copy(path, local)
        subprocess.run([os.path.join(LLVM, 'llvm-objdump'), '--offloading', local], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, cwd=tmp)
        dev = [p for p in os.listdir(tmp) if 'gfx950' in p]
        if not dev:
            return []               # host-only object (runtime.hip has no kernels)
        out = []
        for d in dev:
            notes = subprocess.run([os.path.join(LLVM, 'llvm-readelf'), '--notes', os.path.join(tmp, d)], check=True, capture_output=True, text=True).stdout
            cur = {}
            for ln in notes.splitlines():
                m = re.match(r'\s*-?\s*\.(\w+):\s+(.*)$', ln)
                if not m:
                    continue
                k, v = m.group(1), m.group(2).strip()
                if k == 'agpr_count' and cur.get('name'):     # first key of a kernel record
                    out.append(cur); cur = {}
                if k in ('name', 'vgpr_count', 'sgpr_count', 'private_segment_fixed_size', 'vgpr_spill_count', 'sgpr_spill_count', 'group_segment_fixed_size', 'agpr_count'):
                    if k == 'name' and 'name' in cur and 'vgpr' in cur:
                        out.append(cur); cur = {}
                    key = {'vgpr_count': 'vgpr', 'sgpr_count': 'sgpr', 'private_segment_fixed_size': 'scratch', 'vgpr_spill_count': 'vgpr_spill',
                           'sgpr_spill_count': 'sgpr_spill', 'group_segment_fixed_size': 'lds', 'agpr_count': 'agpr', 'name': 'name'}[k]
                    cur[key] = v.strip("'") if key == 'name' else int(v)
            if cur.get('name'):
                out.append(cur)
        return [k for k in out if 'vgpr' in k]


def demangle(names):
    tool = shutil.which('c++filt') or os.path.join(LLVM, 'llvm-cxxfilt')
    try:
        p = subprocess.run([tool], input='\n'.join(names), capture_output=True, text=True)
    except OSError:
        return names
    return p.stdout.splitlines() if p.returncode == 0 else names


if __name__ == '__main__':
    args = [a for a in sys.argv[1:] if not a.startswith('--')]
    only = '--scratch-only' in sys.argv
    pat = sys.argv[sys.argv.index('--grep') + 1] if '--grep' in sys.argv else ''
    if pat in args:
        args.remove(pat)
    for path in args:
        ks = kernels(path)
        names = demangle([k['name'] for k in ks])
        for k, n in zip(ks, names):
            if only and not (k.get('scratch', 0) or k.get('vgpr_spill', 0)):
                continue
            if pat and pat not in n:
                continue
            print(f"vgpr {k.get('vgpr', 0):3d} agpr {k.get('agpr', 0):3d} sgpr {k.get('sgpr', 0):3d} scratch {k.get('scratch', 0):5d} B  spills v{k.get('vgpr_spill', 0)} s{k.get('sgpr_spill', 0)}  lds {k.get('lds', 0):6d}  {n[:230]}")
