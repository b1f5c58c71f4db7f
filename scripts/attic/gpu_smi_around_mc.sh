echo "== before"; rocm-smi --showperflevel --showprofile 2>/dev/null | grep -v "^$\|====" | head -20
timeout -k 10 200 python3 scripts/attic/gpu_mc_variants.py $1 2>&1 | tail -1 | cut -c1-60
echo "== after"; rocm-smi --showperflevel --showprofile --showclocks 2>/dev/null | grep -v "^$\|====" | head -30
cat /sys/class/drm/card*/device/pp_power_profile_mode 2>/dev/null | head -20
cat /sys/module/amdgpu/parameters/sched_policy /sys/module/amdgpu/parameters/mes 2>/dev/null
