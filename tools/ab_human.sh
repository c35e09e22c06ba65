#!/bin/bash
# human_slice (CpG ~1 % of the bases): CpG on the per-site kernels (default: trunk = 2 picks per context by site density) against everything on the dense trunk
for rep in 1 2; do for o in "trunk=2" "trunk=1"; do
  python bench.py --workload human_slice --steps 12 --warmup 2 --no-extras --no-cpu-baseline --no-e2e --opt $o 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); t=d['device_ms_timed_region']; print('$o', round(d['value']/1e6,2),'M sites/s', d['config']['kernel_path_by_context'], {k:round(v/d['steps'],1) for k,v in t.items() if k.endswith('_ms') and v>0})"
done; done
