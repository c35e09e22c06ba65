#!/bin/bash
# Round-5 evidence, run on the GPU box (bash tools/prof_r05.sh [stats|pmc|traffic|power|all] [tag]): everything lands under
# gpurun_out/prof_r05<tag>/; the summaries are then copied into profiles/ (tracked) by hand.  Every pass runs the DRIVER's
# configuration (bench.py defaults: 11 700-read slabs, the engine's default trunk groups), not a reduced one.
#   stats  : rocprofv3 --kernel-trace --stats of the driver's bench command (secondary measurements off)  -> kernel_stats.csv
#   pmc    : SQ counter passes, each in its own run (only --kernel-trace beside --pmc), 2 steps              -> pmc_summary.txt
#   traffic: FETCH_SIZE / WRITE_SIZE passes at full slab size                                              -> pmc_summary.txt
#   power  : rocm-smi samples (power, sclk, the cap in force) beside an un-profiled bench run              -> power_sample.txt
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
what=${1:-all}
O=$R/gpurun_out/prof_r05$2
mkdir -p $O
if [ "$what" = stats ] || [ "$what" = all ]; then
  timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o s -- python3 $R/bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline > $O/stats_bench.json 2> $O/stats.log || echo "stats run failed"
  cp $(find $O/stats -name '*kernel_stats.csv' | head -1) $O/kernel_stats.csv 2>/dev/null
  head -12 $O/kernel_stats.csv
fi
pmc_pass() {  # name, counters...
  n=$1; shift
  timeout -k 10 400 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $O/pmc/$n -o p -- python3 $R/bench.py --steps 2 --warmup 1 --no-extras --no-cpu-baseline > $O/pmc_bench_$n.json 2> $O/pmc_$n.log || echo "pmc pass $n failed"
}
if [ "$what" = pmc ] || [ "$what" = all ]; then
  pmc_pass g1 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_MFMA
  pmc_pass g2 SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD
  pmc_pass g3 SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY
  pmc_pass g6 GRBM_GUI_ACTIVE
fi
if [ "$what" = traffic ] || [ "$what" = all ]; then
  pmc_pass g4 FETCH_SIZE
  pmc_pass g5 WRITE_SIZE
fi
if [ "$what" != stats ] && [ "$what" != power ]; then
  python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: [0.0, 0])
for f in sorted(glob.glob("$O/pmc/g*/**/*counter_collection.csv", recursive=True)):
    for row in csv.DictReader(open(f)):
        n = row["Kernel_Name"]
        k = n.split("(")[0].split("<")[0][-28:] + ("<13" if "<13" in n or "ILi13" in n else "")
        for t in ("tail_kernel_p", "tail_fc_kernel", "class_count_kernel", "class_write_kernel", "class_mark_kernel", "class_scan_kernel", "tail_kernel_r", "tail_main_kernel", "tail_head_kernel", "trunk3_kernel", "trunk2_kernel", "rowlist3_kernel", "edge2_kernel"):
            if t in n:
                k = t + ("<13>" if ("<13" in n or "ILi13" in n) else "<11>" if ("<11" in n or "ILi11" in n) else "")
        a = agg[(k, row["Counter_Name"])]
        a[0] += float(row["Counter_Value"]); a[1] += 1
with open("$O/pmc_summary.txt", "w") as o:
    for (k, c), (v, n) in sorted(agg.items()):
        o.write(f"{k}\t{c}\tmean_per_launch\t{v / n:.6g}\tlaunches\t{n}\n")
print(open("$O/pmc_summary.txt").read()[:3000])
PY
fi
if [ "$what" = power ] || [ "$what" = all ]; then
  {
    echo "# rocm-smi beside: python3 bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline (un-profiled)"
    echo "# cap in force (rocm-smi --showmaxpower):"; rocm-smi --showmaxpower 2>&1 | grep -i -E 'power|GPU\[' | head -4
    echo "# sysfs power1_cap (uW):"; cat /sys/class/drm/card*/device/hwmon/hwmon*/power1_cap 2>/dev/null | head -2
  } > $O/power_sample.txt
  python3 $R/bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline > $O/power_bench.json 2> $O/power_bench.log &
  BP=$!
  for i in $(seq 1 40); do
    sleep 1
    echo "t=$i $(rocm-smi --showpower --showclocks 2>/dev/null | grep -E 'Socket Graphics Package Power|Average Graphics Package Power|sclk' | sed 's/^GPU\[0\][[:space:]]*: //' | tr '\n' ' ')" >> $O/power_sample.txt
    kill -0 $BP 2>/dev/null || break
  done
  wait $BP
  tail -5 $O/power_sample.txt
fi
