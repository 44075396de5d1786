"""The built library holds no packed-fp32 arithmetic with an op_sel source swizzle.

On MI355X `v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32 ... op_sel:[0,1]` (the low result lane takes the HIGH register of src1 -- how the
compiler broadcasts a scalar that sits in the odd register of a pair) occasionally computes with the wrong half while a wave of the
split-bf16 MLP kernel is resident on the same SIMD: tools/exp/pkopsel reproduces it with a single instruction, the sampler of
mvpnet_amd/csrc/fps.hip returned wrong indices beside the MLP kernels because of it.  The forms without `op_sel`
(plain, op_sel_hi:[1,0], op_sel_hi:[0,1]) and v_pk_mov_b32 were exact in the same experiment.  This test disassembles the gfx950 code
objects inside libmvp_hip.so and fails if a kernel contains the bad form, whatever put it there (a {v, v} splat folded by
instruction selection, the SLP vectoriser pairing scalar code)."""
import glob
import os
import re
import shutil
import subprocess

import pytest

from mvpnet_amd import _lib

OBJDUMP = '/opt/rocm/lib/llvm/bin/llvm-objdump'


def disassemble(tmp_path):
    lib = os.path.join(str(tmp_path), 'libmvp_hip.so')
    shutil.copy(_lib.LIB_PATH, lib)
    subprocess.run([OBJDUMP, '--offloading', lib], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, cwd=str(tmp_path))
    objs = sorted(glob.glob(lib + '.*gfx950'))
    assert objs, 'no gfx950 code object inside %s' % _lib.LIB_PATH
    text = ''
    for o in objs:
        text += subprocess.run([OBJDUMP, '-d', '--mcpu=gfx950', o], check=True, capture_output=True, text=True).stdout
    return text


@pytest.mark.skipif(not os.path.exists(OBJDUMP), reason='llvm-objdump of the ROCm toolchain is not installed')
def test_no_packed_fp32_arithmetic_with_an_op_sel_swizzle(tmp_path):
    if not os.path.exists(_lib.LIB_PATH):
        pytest.skip('libmvp_hip.so is not built')
    text = disassemble(tmp_path)
    kernel, bad, packed = None, {}, 0
    for line in text.splitlines():
        if line.endswith('>:'):
            m = re.match(r'^[0-9a-f]+ <(.*)>:$', line)
            if m:
                kernel = m.group(1)
            continue
        if 'v_pk_' in line and re.search(r'\bv_pk_(add|mul|fma)_f32\b', line):
            packed += 1
            if 'op_sel:' in line:
                bad.setdefault(kernel, []).append(line.strip())
    assert packed > 1000, 'the disassembly holds the kernels (packed fp32 ops found: %d)' % packed
    assert not bad, 'packed fp32 arithmetic with op_sel in: ' + ', '.join('%s (%d)' % (k, len(v)) for k, v in sorted(bad.items())[:8])
