#!/bin/bash
# Round-3 evidence, run on the GPU box (bash tools/prof_r03.sh [stats|pmc|all]): everything lands under gpurun_out/prof_r03/;
# the summaries are then copied into profiles/ (tracked) by hand.
#   stats: rocprofv3 --kernel-trace --stats of the driver's bench command (secondary measurements off)  -> kernel_stats.csv
#   pmc  : PMC passes, each in its own run (only --kernel-trace beside --pmc)                            -> pmc_summary.txt, pmc_bench.json
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/prof_r03
mkdir -p $O
what=${1:-all}
if [ "$what" = stats ] || [ "$what" = all ]; then
  timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o s -- python3 $R/bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline > $O/stats_bench.json 2> $O/stats.log || echo "stats run failed"
  cp $(find $O/stats -name '*kernel_stats.csv' | head -1) $O/kernel_stats.csv 2>/dev/null
  head -12 $O/kernel_stats.csv
fi
if [ "$what" = pmc ] || [ "$what" = all ]; then
  i=0
  for grp in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_MFMA" \
             "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" \
             "SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" \
             "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE"; do
    i=$((i+1))
    timeout -k 10 300 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $O/pmc/g$i -o p -- python3 $R/bench.py --steps 1 --warmup 1 --reads 2000 --no-extras --no-cpu-baseline > $O/pmc_bench_g$i.json 2> $O/pmc_g$i.log || echo "pmc group $i failed"
  done
  cp $O/pmc_bench_g6.json $O/pmc_bench.json
  python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: [0.0, 0])
for f in sorted(glob.glob("$O/pmc/g*/**/*counter_collection.csv", recursive=True)):
    for row in csv.DictReader(open(f)):
        n = row["Kernel_Name"]
        k = "tail_kernel_r" if "tail_kernel_r" in n else n.split("(")[0].split("<")[0][-28:] + ("<13" if "<13" in n or "ILi13" in n else "")
        a = agg[(k, row["Counter_Name"])]
        a[0] += float(row["Counter_Value"]); a[1] += 1
with open("$O/pmc_summary.txt", "w") as o:
    for (k, c), (v, n) in sorted(agg.items()):
        o.write(f"{k}\t{c}\tmean_per_launch\t{v / n:.6g}\tlaunches\t{n}\n")
print(open("$O/pmc_summary.txt").read()[:3000])
PY
fi
