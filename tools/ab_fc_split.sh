#!/bin/bash
# A/B of the fc split on one box: the strip tail with fc1 .. softmax inside the pass (libhifimeth_hip_fcin.so, `make fcin`) against
# tail_kernel_p + tail_fc_kernel (the default library); interleaved, twice.  Byte identity of the calls is tests/test_gpu_parity.py's.
for rep in 1 2; do for lib in libhifimeth_hip_fcin.so libhifimeth_hip.so; do
  HM_LIB_PATH=$PWD/hifimeth_amd/$lib python bench.py --steps 8 --warmup 3 --no-extras --no-cpu-baseline --no-e2e 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); t=d['device_ms_timed_region']; print('$lib', round(d['value']/1e6,2),'M sites/s', {k:round(v/d['steps'],1) for k,v in t.items() if k.endswith('_ms') and v>0})"
done; done
