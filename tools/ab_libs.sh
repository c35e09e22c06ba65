#!/bin/bash
# same-session A/B of library build variants hifimeth_amd/libhm_var_<name>.so with tools/ab_trunk.py (trunk / edge / tail device ms)
for v in "$@"; do
  echo "=== $v"
  HM_LIB_PATH=$PWD/hifimeth_amd/libhm_var_$v.so timeout -k 10 120 python tools/ab_trunk.py 2>&1 | grep "impl 1" | tail -2 || exit 1
done
