#!/bin/bash
# experiment: the lo halves (activations: build with -DHM_XP_LO_BITS=n; weights: HM_XP_WLO_MASK_BITS=n) rounded to 10 - n mantissa bits -- does the
# chip hold a higher clock on operands with fewer significant bits, and what does it cost in |dp|?  Same box, interleaved, in-run parity of 465 k sites.
for rep in 1 2; do
  for n in 0 4 6 8; do
    if [ $n = 0 ]; then L=hifimeth_amd/libhifimeth_hip.so; E=""; else L=hifimeth_amd/lib_xp_lo$n.so; E="HM_XP_WLO_MASK_BITS=$n"; fi
    env HM_LIB_PATH=$PWD/$L $E python bench.py --steps 6 --warmup 2 --no-extras --no-e2e 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); t=d['device_ms_timed_region']; p=d['parity']; print('lo bits dropped $n:', round(d['value']/1e6,2),'M sites/s', {k:round(x/d['steps'],1) for k,x in t.items() if k in ('trunk_ms','edge_ms','tail_ms')}, 'max|dp| %.2e mean %.2e p99.9 %.2e ml+-1 %d' % (p['max_abs_dp_vs_oracle'], p['mean_abs_dp'], p['p999_abs_dp'], p['ml_bytes_off_by_1lsb']))"
  done
done
