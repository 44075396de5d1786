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


READELF = '/opt/rocm/lib/llvm/bin/llvm-readelf'
CXXFILT = shutil.which('c++filt') or '/opt/rocm/lib/llvm/bin/llvm-cxxfilt'


def kernel_metadata(tmp_path):
    """{demangled kernel name without its argument list: (vgpr_count, vgpr_spill_count, sgpr_spill_count, lds bytes)} of every gfx950
    kernel in libmvp_hip.so, from the code objects' AMDGPU metadata notes (what the loader reads)."""
    lib = os.path.join(str(tmp_path), 'libmvp_hip.so')
    shutil.copy(_lib.LIB_PATH, lib)
    subprocess.run([OBJDUMP, '--offloading', lib], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, cwd=str(tmp_path))
    recs = []
    for o in sorted(glob.glob(lib + '.*gfx950')):
        notes = subprocess.run([READELF, '--notes', o], check=True, capture_output=True, text=True).stdout
        for blk in re.split(r'\n\s+- \.agpr_count', notes)[1:]:
            f = lambda k: re.search(r'\.' + k + r':\s+(\S+)', blk)
            if f('name') and f('vgpr_spill_count'):
                recs.append((f('name').group(1), int(f('vgpr_count').group(1)), int(f('vgpr_spill_count').group(1)), int(f('sgpr_spill_count').group(1)),
                             int(f('group_segment_fixed_size').group(1))))
    dem = subprocess.run([CXXFILT], input='\n'.join(r[0] for r in recs), check=True, capture_output=True, text=True).stdout.split('\n')
    out = {}
    for r, d in zip(recs, dem):
        d = d.replace('void ', '').replace('(anonymous namespace)::', '')
        depth, name = 0, ''
        for ch in d:
            depth += ch == '<'
            depth -= ch == '>'
            if ch == '(' and depth == 0:
                break
            name += ch
        out[name] = r[1:]
    return out


@pytest.mark.skipif(not (os.path.exists(OBJDUMP) and os.path.exists(READELF)), reason='llvm-objdump / llvm-readelf of the ROCm toolchain are not installed')
def test_the_kernels_of_the_training_step_do_not_spill(tmp_path):
    """No kernel instance that the B = 32 training step launches (tests/golden/step_kernels_b32.txt) keeps registers in scratch memory
    (mlp_bwd_wide_kernel<2, 2, 128> -- the segmentation head -- spilled 10 VGPRs at 256 registers and two waves per SIMD)."""
    if not os.path.exists(_lib.LIB_PATH):
        pytest.skip('libmvp_hip.so is not built')
    meta = kernel_metadata(tmp_path)
    assert len(meta) > 300, 'kernel metadata found for %d kernels' % len(meta)
    with open(os.path.join(os.path.dirname(__file__), 'golden', 'step_kernels_b32.txt')) as f:
        step = [l.strip() for l in f if l.strip() and not l.startswith('#')]
    assert len(step) > 50
    missing = [k for k in step if k not in meta]
    assert not missing, 'kernels of the step that the library no longer holds (regenerate the list): %s' % missing[:5]
    # (vector registers only: a spilled SCALAR register lives in a lane of a vector register -- v_writelane / v_readlane --, not in memory;
    # sa_train_bwd_kernel<2, 2, *> and seg_loss_kernel hold 30 - 106 of those)
    spilled = {k: meta[k] for k in step if meta[k][1] > 0}
    assert not spilled, 'kernels of the training step with vector registers in scratch memory (vgprs, vgpr spill, sgpr spill, lds): %s' % spilled
