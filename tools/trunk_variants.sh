#!/bin/bash
# A/B of trunk_kernel build variants (stamped diagnostic libraries hifimeth_amd/libhm_var_<name>.so) in one GPU session
for v in "$@"; do
  echo "=== $v"
  HM_LIB_PATH=$PWD/hifimeth_amd/libhm_var_$v.so timeout -k 10 120 python tools/trunk_stamps.py 2>&1 | grep -v amdgpu.ids || exit 1
done
