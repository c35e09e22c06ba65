#!/bin/bash
# one GPU call: byte identity of the tails, the fc-split A/B, kernel times (tools/fc_check.sh)
R=${GRAFT_REPO_ROOT:-/root/repo}
timeout -k 10 500 python -m pytest tests/test_gpu_parity.py -x -q -k "strip or resident or human or full_size or ragged" > gpurun_out/fc_parity.txt 2>&1; tail -2 gpurun_out/fc_parity.txt
tools/ab_fc_split.sh > gpurun_out/fc_ab.txt 2>&1; cat gpurun_out/fc_ab.txt
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/fc_stats
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/fc_stats -o s -- python3 $R/bench.py --steps 6 --warmup 2 --no-extras --no-cpu-baseline --no-e2e > $R/gpurun_out/fc_stats.json 2> $R/gpurun_out/fc_stats.log
grep -i "tail_fc\|tail_kernel_p\|class_" $R/gpurun_out/fc_stats/s_kernel_stats.csv | awk -F, '{print substr($1,1,30), $(NF-6), $(NF-5), $(NF-4), $(NF-3)}'
