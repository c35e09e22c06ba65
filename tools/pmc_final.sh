#!/bin/bash
# PMC passes over bench.py for the final front kernel: MFMA busy, instruction mix, HBM bytes (each group in its own run:
# --pmc only with --kernel-trace).  usage (on the GPU box): bash tools/pmc_final.sh -> gpurun_out/pmc_final/g<i>/
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
i=0
for grp in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_MFMA" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" \
           "SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM_RD" \
           "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $R/gpurun_out/pmc_final/g$i -o p -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/pmc_final_g$i.log 2>&1 || echo "group $i failed"
done
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: [0.0, 0])
for f in sorted(glob.glob("$R/gpurun_out/pmc_final/g*/**/*counter_collection.csv", recursive=True)):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0][-40:]
        a = agg[(k, row["Counter_Name"])]
        a[0] += float(row["Counter_Value"]); a[1] += 1
with open("$R/gpurun_out/pmc_final/summary.txt", "w") as o:
    for (k, c), (v, n) in sorted(agg.items()):
        o.write(f"{k}\t{c}\tmean_per_launch\t{v / n:.6g}\tlaunches\t{n}\n")
print(open("$R/gpurun_out/pmc_final/summary.txt").read())
PY
