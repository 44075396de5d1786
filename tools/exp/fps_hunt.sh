cd /root/repo
for v in splat splat intmin; do
TAG=$v MVP_LIBRARY=/root/repo/tools/exp/variants/lib_$v.so timeout 300 python tools/exp/fps_hunt.py 2>&1 | tail -1
done
