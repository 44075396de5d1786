#!/bin/bash
# Where a tile of the one-pass wide backward spends its time: builds the library with -DMVP_WIDE_PROF (phase timestamps per workgroup,
# csrc/mlp_bwd_wide.hip) as tools/exp/libmvp_wideprof.so and runs the kernel alone on the 262144 x 128 -> 128 shape.
#   bash tools/exp/wide_prof.sh   (on the GPU box)
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd $root/mvpnet_amd/csrc
objs=""
for f in *.hip; do
  o=build/${f%.hip}.o
  if [ "$f" = "mlp_bwd_wide.hip" ]; then
    o=/tmp/mlp_bwd_wide_prof.o
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -munsafe-fp-atomics -fvisibility=hidden -fno-slp-vectorize -DMVP_WIDE_PROF ${MVP_WIDE_EXP:+-DMVP_WIDE_EXP=$MVP_WIDE_EXP} -c $f -o $o || exit 1
  fi
  objs="$objs $o"
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $root/tools/exp/libmvp_wideprof.so $objs || exit 1
cd $root
MVP_LIBRARY=$root/tools/exp/libmvp_wideprof.so python tools/exp/wide_prof.py
