#!/bin/bash
# A/B of engine builds in ONE gpurun call (same box, same clocks): tools/ab_bench.sh lib1.so lib2.so ... [-- bench args]
# prints sites/s and the per-kernel device ms of every build; box-to-box differences exceed most single changes.
libs=(); while [ $# -gt 0 ] && [ "$1" != "--" ]; do libs+=("$1"); shift; done; [ "$1" == "--" ] && shift
for rep in 1 2; do for l in "${libs[@]}"; do
  HM_LIB_PATH=$PWD/$l python bench.py --steps 6 --warmup 2 --no-extras --no-cpu-baseline "$@" 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); t=d['device_ms_timed_region']; print('$l', round(d['value']/1e6,2),'M sites/s', {k:round(v/d['steps'],1) for k,v in t.items() if k.endswith('_ms') and v>0})"
done; done
