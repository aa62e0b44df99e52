"""Scans the gfx950 code objects inside libfss_gfx950.so for packed-f32 VALU instructions
(v_pk_fma_f32 / v_pk_add_f32 / v_pk_mul_f32).

The library must not contain them: with hipcc's SLP vectorizer on, pyramid_add_kernel's blends
became v_pk_fma_f32 + v_pk_add_f32 (halves exchanged through op_sel), and the results of one
half were lost in lanes 48-63 of a wave from time to time whenever the kernel shared compute
units with the uint8 stem convolution on a second stream (eager launches and graph replays
alike; never when it ran alone).  The same source built with -fno-slp-vectorize has no packed
f32 math and never showed it.  Exit status 1 lists the offending kernels.

    python tools/check_isa.py [path/to/libfss_gfx950.so]
"""
import os
import re
import shutil
import subprocess
import sys
import tempfile

LLVM = '/opt/rocm/lib/llvm/bin'
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEFAULT = os.path.join(ROOT, 'fast-semantic-segmentation_amd', 'libfss_gfx950.so')
PACKED = re.compile(r'\bv_pk_(fma|add|mul)_f32\b')


def packed_f32_kernels(lib_path):
    """{kernel symbol: number of packed-f32 instructions} over every device code object."""
    found = {}
    tmp = tempfile.mkdtemp(prefix='fss_isa_')
    try:
        so = os.path.join(tmp, 'lib.so')
        shutil.copy(lib_path, so)
        subprocess.run([os.path.join(LLVM, 'llvm-objdump'), '--offloading', so], check=True,
                       stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, cwd=tmp)
        objects = [f for f in os.listdir(tmp) if 'amdgcn' in f]
        if not objects:
            raise RuntimeError('no gfx950 code object found in %s' % lib_path)
        for f in objects:
            dis = subprocess.run([os.path.join(LLVM, 'llvm-objdump'), '-d', '--mcpu=gfx950',
                                  os.path.join(tmp, f)], check=True, capture_output=True, text=True).stdout
            sym = None
            for line in dis.splitlines():
                m = re.match(r'^[0-9a-f]+ <([^>]+)>:', line)
                if m:
                    sym = m.group(1)
                elif PACKED.search(line):
                    found[sym] = found.get(sym, 0) + 1
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    return found


if __name__ == '__main__':
    bad = packed_f32_kernels(sys.argv[1] if len(sys.argv) > 1 else DEFAULT)
    for k, v in sorted(bad.items()):
        print('%6d packed-f32 instructions in %s' % (v, k))
    print('packed-f32 VALU: %s' % ('FOUND' if bad else 'none'))
    sys.exit(1 if bad else 0)
