#!/usr/bin/env python3
"""Count packed-fp32 instructions with an op_sel source swizzle (`v_pk_{add,mul,fma}_f32 ... op_sel:[..]`) in the gfx950
code objects of OTHER people's kernels that run in the training step -- ATen's elementwise / reduce / scatter-gather kernels out of
libtorch_hip.so -- and in this repository's library.  VERDICT r3 next #7(a).

    python tools/pkopsel_scan.py [--lib PATH ...] [--names FILE_OR_PATTERN ...] [--out profiles/r04_pkopsel_scan.json]

For every shared library: the `.hip_fatbin` section is cut into its (compressed, "CCOB") clang offload bundles, each bundle's gfx950
code object is extracted with clang-offload-bundler and disassembled with llvm-objdump; per kernel symbol the packed-fp32 instructions
are counted by form.  --names restricts the report to kernels whose (demangled) name contains one of the given substrings; default: the
ATen kernels of profiles/r03_step_kernel_stats.csv (the step's non-library launches).  Runs on the CPU container (no GPU needed)."""
import argparse
import collections
import json
import os
import re
import struct
import subprocess
import sys
import tempfile

LLVM = '/opt/rocm/lib/llvm/bin'
TARGET = 'hipv4-amdgcn-amd-amdhsa--gfx950'
PK = re.compile(r'\b(v_pk_(?:add|mul|fma)_f32)\b(.*)')
DEFAULT_NAMES = ['FillFunctor<float>', 'FillFunctor<double>', 'CUDAFunctor_add<float>', 'sum_functor<float', '_scatter_gather_elementwise_kernel',
                 'direct_copy_kernel_cuda', 'MulFunctor<bool>', 'AbsFunctor<float>', 'linspace']


def fatbin(path):
    with open(path, 'rb') as f:
        eh = f.read(64)
        shoff = struct.unpack_from('<Q', eh, 0x28)[0]
        shentsize, shnum, shstrndx = struct.unpack_from('<HHH', eh, 0x3A)
        f.seek(shoff)
        sh = f.read(shentsize * shnum)
        secs = [struct.unpack_from('<IIQQQQIIQQ', sh, i * shentsize) for i in range(shnum)]
        f.seek(secs[shstrndx][4])
        strtab = f.read(secs[shstrndx][5])
        for s in secs:
            name = strtab[s[0]:strtab.index(b'\0', s[0])].decode()
            if name == '.hip_fatbin':
                f.seek(s[4])
                return f.read(s[5])
    return b''


def bundles(blob):
    """-> byte strings, one per offload bundle (compressed CCOB v2 / v3 or plain __CLANG_OFFLOAD_BUNDLE__)."""
    i = 0
    while True:
        j = blob.find(b'CCOB', i)
        k = blob.find(b'__CLANG_OFFLOAD_BUNDLE__', i)
        if j < 0 and k < 0:
            return
        if k < 0 or (0 <= j < k):
            ver = struct.unpack_from('<H', blob, j + 4)[0]
            total = struct.unpack_from('<Q' if ver >= 3 else '<I', blob, j + 8)[0]
            if total <= 16 or j + total > len(blob):
                i = j + 4
                continue
            yield blob[j:j + total]
            i = j + total
        else:
            n = struct.unpack_from('<Q', blob, k + 24)[0]
            end = k
            off = k + 32
            for _ in range(n):
                o, sz, ts = struct.unpack_from('<QQQ', blob, off)
                end = max(end, k + o + sz)
                off += 24 + ts
            yield blob[k:end]
            i = max(end, k + 24)


def demangle(names):
    if not names:
        return {}
    out = subprocess.run(['c++filt'], input='\n'.join(names), capture_output=True, text=True).stdout.split('\n')
    return dict(zip(names, out))


def scan_code_object(path):
    """-> {kernel symbol: Counter(form -> count)} for one gfx950 code object."""
    txt = subprocess.run([os.path.join(LLVM, 'llvm-objdump'), '-d', '--mcpu=gfx950', path], capture_output=True, text=True).stdout
    res, cur = {}, None
    for line in txt.split('\n'):
        if line.endswith('>:') and '<' in line:
            cur = line[line.index('<') + 1:-2]
            res.setdefault(cur, collections.Counter())
            continue
        if cur is None:
            continue
        m = PK.search(line)
        if m:
            mods = ' '.join(re.findall(r'op_sel(?:_hi)?:\[[0-9,]+\]', m.group(2)))
            res[cur][(m.group(1) + ' ' + mods).strip()] += 1
    return res


def is_bad(form):
    """the misexecuting family of DESIGN.md 4.10: an op_sel (not op_sel_hi) modifier that takes the HIGH register of a later source for the low lane"""
    m = re.search(r'op_sel:\[([0-9,]+)\]', form)
    return bool(m) and any(x == '1' for x in m.group(1).split(',')[1:])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--lib', action='append', default=[])
    ap.add_argument('--names', action='append', default=[])
    ap.add_argument('--out', default='')
    ap.add_argument('--max-bundles', type=int, default=0)
    args = ap.parse_args()
    libs = args.lib
    if not libs:
        import torch
        libs = [os.path.join(os.path.dirname(torch.__file__), 'lib', 'libtorch_hip.so'),
                os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'mvpnet_amd', 'libmvp_hip.so')]
    names = args.names or DEFAULT_NAMES
    report = {'target': TARGET, 'name_filters': names, 'libraries': {}}
    for lib in libs:
        blob = fatbin(lib)
        per_kernel = {}
        n_b = n_co = 0
        with tempfile.TemporaryDirectory() as tmp:
            for bi, b in enumerate(bundles(blob)):
                if args.max_bundles and bi >= args.max_bundles:
                    break
                n_b += 1
                bp, cp = os.path.join(tmp, 'b.bin'), os.path.join(tmp, 'c.co')
                with open(bp, 'wb') as f:
                    f.write(b)
                lst = subprocess.run([os.path.join(LLVM, 'clang-offload-bundler'), '--type=o', '--input=' + bp, '--list'], capture_output=True, text=True).stdout
                if TARGET not in lst:
                    continue
                r = subprocess.run([os.path.join(LLVM, 'clang-offload-bundler'), '--type=o', '--input=' + bp, '--targets=' + TARGET,
                                    '--output=' + cp, '--unbundle'], capture_output=True, text=True)
                if r.returncode != 0 or not os.path.exists(cp) or os.path.getsize(cp) == 0:
                    continue
                n_co += 1
                for sym, forms in scan_code_object(cp).items():
                    if forms:
                        per_kernel.setdefault(sym, collections.Counter()).update(forms)
                    else:
                        per_kernel.setdefault(sym, collections.Counter())
                os.remove(cp)
        dm = demangle(list(per_kernel))
        own = os.path.basename(lib).startswith('libmvp_hip')
        rows = []
        tot_kernels = tot_pk = tot_bad = matched = 0
        for sym, forms in per_kernel.items():
            name = dm.get(sym, sym)
            tot_kernels += 1
            n_pk = sum(forms.values())
            n_bad = sum(c for f, c in forms.items() if is_bad(f))
            tot_pk += n_pk
            tot_bad += n_bad
            if own or any(s in name for s in names):
                matched += 1
                if n_pk:
                    rows.append({'kernel': name[:240], 'packed_fp32': n_pk, 'op_sel_forms_of_4.10': n_bad, 'forms': dict(forms)})
        rows.sort(key=lambda r: (-r['op_sel_forms_of_4.10'], -r['packed_fp32']))
        # the kernel FAMILIES (name up to the first template bracket) that carry the form at all, whether or not they run in the step
        fam = collections.Counter()
        fam_k = collections.Counter()
        for sym, forms in per_kernel.items():
            n_bad = sum(c for f, c in forms.items() if is_bad(f))
            if n_bad:
                base = re.sub(r'^void ', '', dm.get(sym, sym)).split('<')[0].split('(')[0]
                fam[base] += n_bad
                fam_k[base] += 1
        report['libraries'][os.path.basename(lib)] = {
            'bundles': n_b, 'gfx950_code_objects': n_co, 'kernels': tot_kernels, 'packed_fp32_instructions': tot_pk,
            'op_sel_forms_of_4.10_all_kernels': tot_bad, 'kernels_matching_filters': matched,
            'families_with_the_form': [{'family': f, 'kernels': fam_k[f], 'instructions': c} for f, c in fam.most_common(40)],
            'matching_kernels_with_packed_fp32': rows[:60]}
        print('{}: {} bundles, {} gfx950 code objects, {} kernels, {} packed fp32 instructions, {} with the op_sel form; {} kernels match the filters, {} of them use '
              'packed fp32, {} carry the form'.format(os.path.basename(lib), n_b, n_co, tot_kernels, tot_pk, tot_bad, matched, len(rows),
                                                      sum(1 for r in rows if r['op_sel_forms_of_4.10'])), file=sys.stderr)
    txt = json.dumps(report, indent=1)
    if args.out:
        with open(args.out, 'w') as f:
            f.write(txt)
    else:
        print(txt)


if __name__ == '__main__':
    main()
