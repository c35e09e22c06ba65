#!/bin/bash
# A/B of engine LIBRARIES on one box (tools/ab_libs.sh libA.so libB.so ...; names under hifimeth_amd/): bench.py --steps 8, interleaved, twice;
# then byte identity of the calls of every library against the first (tools/ab_tail.py's slab, CHH on the strip tail)
for rep in 1 2; do for lib in "$@"; do
  HM_LIB_PATH=$PWD/hifimeth_amd/$lib python bench.py --steps 8 --warmup 3 --no-extras --no-cpu-baseline --no-e2e 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); t=d['device_ms_timed_region']; print('$lib', round(d['value']/1e6,2),'M sites/s', {k:round(v/d['steps'],1) for k,v in t.items() if k.endswith('_ms') and v>0})"
done; done
for lib in "$@"; do
  HM_LIB_PATH=$PWD/hifimeth_amd/$lib python - <<PY
import sys, hashlib
sys.path.insert(0, ".")
from hifimeth_amd import MethylationCaller
from hifimeth_amd.synth import synth_reads
reads = synth_reads(300, seed=5)
with MethylationCaller(device=0) as m:
    m.set_option("trunk", 1)
    c = m.call(reads)
    print("$lib", len(c), "calls, md5", hashlib.md5(c.tobytes()).hexdigest())
PY
done
