#!/bin/bash
# A/B of engine OPTIONS in one gpurun call (same box): tools/ab_opts2.sh "tail_impl=1" "tail_impl=3" ... ; every option set twice, interleaved
for rep in 1 2; do for o in "$@"; do
  args=(); for kv in $o; do args+=(--opt "$kv"); done
  python bench.py --steps 6 --warmup 2 --no-extras --no-cpu-baseline --no-e2e "${args[@]}" 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); t=d['device_ms_timed_region']; print('$o', round(d['value']/1e6,2),'M sites/s', {k:round(v/d['steps'],1) for k,v in t.items() if k.endswith('_ms') and v>0})"
done; done
