#!/bin/bash
# What the closing atomics of the one-pass layer backward cost: the library built with plain stores instead (MVP_WIDE_EXP=5: the column sums,
# 6: dW, 7: both -- wrong results, timing only), the kernel alone (tools/exp/wide_time.py)
root=${GRAFT_REPO_ROOT:-$(pwd)}
for x in 0 5 6 7; do
  cd $root/mvpnet_amd/csrc
  objs=""
  for f in *.hip; do
    o=build/${f%.hip}.o
    if [ "$f" = "mlp_bwd_wide.hip" ] && [ $x != 0 ]; then
      o=/tmp/wide_exp$x.o
      /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -munsafe-fp-atomics -fvisibility=hidden -fno-slp-vectorize -DMVP_WIDE_EXP=$x -c $f -o $o || exit 1
    fi
    objs="$objs $o"
  done
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/libmvp_wexp$x.so $objs || exit 1
  cd $root
  echo "== MVP_WIDE_EXP=$x"
  MVP_LIBRARY=/tmp/libmvp_wexp$x.so python tools/exp/wide_time.py 2>&1 | grep "262144.*mode"
done
