for so in tools/exp/liblift_*.so; do v=$(basename $so .so); echo "== $v"; timeout 120 python tools/exp/run_lift_sched.py $v.so 0 2>&1 | grep -E "sched  |sched 0:|Error|error" | tail -3; done
