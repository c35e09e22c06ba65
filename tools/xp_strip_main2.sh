#!/bin/bash
# Timing experiment: the strip kernel (dense conv5) cut behind conv6 -- conv6's rows as global stores (2 688 B per site), the whole next strip
# arriving while conv6 runs, no conv7 / conv8 -- against the full kernel; results of the cut build are garbage, its stamps are not.
# Build: see DESIGN.md 11 (hm_tail_p.hip with -DHM_TRUNK_STAMP -DHM_XP_MAIN_ONLY, hm_tail_fc.hip with -DHM_XP_X8_SCALE=6).
for v in "" _xp_main; do
  echo "== libhifimeth_hip_stamp$v.so"
  HM_LIB_PATH=$PWD/hifimeth_amd/libhifimeth_hip_stamp$v.so timeout -k 10 200 python tools/tailp_stamps.py 2>&1 | grep -v amdgpu.ids | sed -n 1,16p
done
