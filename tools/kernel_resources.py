"""Per-kernel register / LDS / occupancy report of one .hip source (hipcc -Rpass-analysis=kernel-resource-usage).
usage: python tools/kernel_resources.py mvpnet_amd/csrc/mlp.hip [filter]"""
import re
import subprocess
import sys

src = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ''
cmd = ['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-ffp-contract=off', '-munsafe-fp-atomics',
       '-fvisibility=hidden', '-fno-slp-vectorize', '-c', src, '-o', '/dev/null', '-Rpass-analysis=kernel-resource-usage']
txt = subprocess.run(cmd, capture_output=True, text=True).stderr
OCC, LDS = r'Occupancy \[waves/SIMD\]', r'LDS Size \[bytes/block\]'
for b in re.split(r'Function Name: ', txt)[1:]:
    name = b.split('\n')[0].split()[0]
    dn = subprocess.run(['c++filt', name], capture_output=True, text=True).stdout.strip()
    dn = dn.replace('(anonymous namespace)::', '').split('(')[0]
    if flt and flt not in dn:
        continue

    def g(k):
        m = re.search(k + r': (\d+)', b)
        return m.group(1) if m else '?'
    print('{:64s} VGPR {:>4} AGPR {:>4} spill {:>3} occ {:>2} LDS {:>6}'.format(dn[:64], g('VGPRs'), g('AGPRs'), g('VGPRs Spill'), g(OCC), g(LDS)))
