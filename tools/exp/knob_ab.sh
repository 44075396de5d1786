# Same-box A/B of one environment knob:  bash tools/exp/knob_ab.sh NAME=VALUE [NAME=VALUE ...]   (each is compared with the default)
cd /root/repo
one() { python bench.py --no-cpu-baseline --train-only --steps 60 --warmup 10 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"; }
for rep in 1 2 3; do
echo "default  $(one)"
for kv in "$@"; do echo "$kv  $(env $kv bash -c "$(declare -f one); one")"; done
done
