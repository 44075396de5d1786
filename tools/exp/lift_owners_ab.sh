#!/bin/bash
# In-step time of the lifting launch (bench roofline.ms_per_launch) with 1 / 2 / 4 owners per round trip in the whole-wave path of the
# 256-point workgroups (the B = 32 launch went 95.8 -> 100 us with round 4's batched tasks).
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd $root/mvpnet_amd/csrc
for own in 1 2 4; do
  objs=""
  for f in *.hip; do
    o=build/${f%.hip}.o
    if [ "$f" = "lift_fused.hip" ]; then
      o=/tmp/lift_fused_own$own.o
      /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -munsafe-fp-atomics -fvisibility=hidden -fno-slp-vectorize -DMVP_LIFT_OWNERS_BIG=$own -c $f -o $o || exit 1
    fi
    objs="$objs $o"
  done
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/libmvp_own$own.so $objs || exit 1
done
cd $root
for rep in 1 2 3; do
  for own in 1 2 4; do
    MVP_LIBRARY=/tmp/libmvp_own$own.so python bench.py --no-cpu-baseline --train-only --steps 60 --warmup 10 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('owners $own: lifting', d['roofline']['ms_per_launch'], 'ms in-step, frac', d['roofline']['frac'], 'step', d['ms_per_step'])"
  done
done
