cd /root/repo
for v in sa_nosb_noslp sa_sb_slp; do echo == $v; MVP_LIBRARY=/root/repo/tools/exp/variants/lib_$v.so timeout 600 python tools/exp/sa_fused_time.py 2>&1 | tail -3; done
echo == product; timeout 600 python tools/exp/sa_fused_time.py 2>&1 | tail -3
for v in sa_nosb_noslp; do echo == $v; MVP_LIBRARY=/root/repo/tools/exp/variants/lib_$v.so timeout 600 python tools/exp/sa_fused_time.py 2>&1 | tail -3; done
