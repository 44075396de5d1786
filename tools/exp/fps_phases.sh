#!/bin/bash
# Cycles of wave 0 per phase of fps_rounds_kernel (update / best + publish + barrier / resolver scan / resolver picks / hand-over barrier):
# the library built with -DMVP_FPS_PHASES as tools/exp/libmvp_fpsphase.so (build HERE or on the box), then tools/exp/fps_phases.py per setting
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd $root/mvpnet_amd/csrc
if [ ! -f $root/tools/exp/libmvp_fpsphase.so ] || [ fps.hip -nt $root/tools/exp/libmvp_fpsphase.so ]; then
  objs=""
  for f in *.hip; do
    o=build/${f%.hip}.o
    if [ "$f" = "fps.hip" ]; then
      o=/tmp/fps_phase.o
      /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -munsafe-fp-atomics -fvisibility=hidden -fno-slp-vectorize -DMVP_FPS_PHASES -c $f -o $o || exit 1
    fi
    objs="$objs $o"
  done
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $root/tools/exp/libmvp_fpsphase.so $objs || exit 1
fi
cd $root
[ "$1" = "build" ] && exit 0
for rl in ${FPS_RLS:-16 1}; do
  MVP_FPS_RL=$rl MVP_LIBRARY=$root/tools/exp/libmvp_fpsphase.so python tools/exp/fps_phases.py 2>&1 | grep "RL="
  MVP_FPS_DEBUG=2 MVP_FPS_RL=$rl MVP_LIBRARY=$root/tools/exp/libmvp_fpsphase.so python tools/exp/fps_phases.py 2>&1 | grep "RL="
done
