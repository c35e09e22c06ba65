#!/bin/bash
# Where do conv5's / conv6's ticks go?  tools/tailp_stamps.py on stamped builds of the strip kernel with parts of PConv's stream removed
# (make stampxp XP="-DHM_XP_NO_EPI" XPNAME=noepi, ... ; results are garbage, the stamps are not): no epilogue stores / no epilogue at all /
# no operand reads / neither.
for v in "" _xp_nostore _xp_noepi _xp_noreads _xp_noreads_noepi; do
  echo "== libhifimeth_hip_stamp$v.so"
  HM_LIB_PATH=$PWD/hifimeth_amd/libhifimeth_hip_stamp$v.so timeout -k 10 200 python tools/tailp_stamps.py 2>&1 | grep -v amdgpu.ids | sed -n 1,16p
done
