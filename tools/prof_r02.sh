#!/bin/bash
# Round-2 evidence, run on the GPU box (bash tools/prof_r02.sh): everything lands under gpurun_out/prof_r02/ and the
# summaries are then copied into profiles/ by hand.
#   1. rocprofv3 --kernel-trace --stats of the streamed bench            -> kernel_stats.csv
#   2. kernel + memory-copy trace of the same command                    -> overlap.txt (H2D / D2H time hidden under kernels)
#   3. PMC passes, each in its own run (only --kernel-trace beside --pmc) -> pmc_summary.txt
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/prof_r02
mkdir -p $O
BENCH="python3 $R/bench.py --steps 4 --warmup 1 --reads 2000 --no-extras --no-cpu-baseline"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o s -- $BENCH > $O/stats.log 2>&1 || echo "stats run failed"
cp $(find $O/stats -name '*kernel_stats.csv' | head -1) $O/kernel_stats.csv 2>/dev/null
timeout -k 10 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $O/trace -o t -- $BENCH > $O/trace.log 2>&1 || echo "trace run failed"
python3 - <<PY > $O/overlap.txt
import csv, glob
k = glob.glob("$O/trace/**/*kernel_trace.csv", recursive=True)
m = glob.glob("$O/trace/**/*memory_copy_trace.csv", recursive=True)
if k and m:
    ker = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in csv.DictReader(open(k[0])))
    t0, t1 = ker[0][0], ker[-1][1]
    merged = []
    for a, b in ker:
        if merged and a <= merged[-1][1]:
            merged[-1][1] = max(merged[-1][1], b)
        else:
            merged.append([a, b])
    busy = sum(b - a for a, b in merged)
    print(f"kernel span {1e-6 * (t1 - t0):.1f} ms, kernels busy {1e-6 * busy:.1f} ms ({100.0 * busy / (t1 - t0):.1f} % of the span)")
    import bisect
    starts = [a for a, _ in merged]
    tot = {}
    for r in csv.DictReader(open(m[0])):
        a, b = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        d = r.get("Direction", r.get("Kind", "?"))
        if b <= t0 or a >= t1:
            continue   # copies before the first / after the last kernel (model upload, tear-down)
        ov = 0
        i = max(0, bisect.bisect_right(starts, a) - 1)
        while i < len(merged) and merged[i][0] < b:
            ov += max(0, min(b, merged[i][1]) - max(a, merged[i][0]))
            i += 1
        t = tot.setdefault(d, [0, 0, 0, 0])
        t[0] += 1; t[1] += b - a; t[2] += ov; t[3] += int(r.get("Bytes", 0) or 0) if "Bytes" in r else 0
    for d, (n, dur, ov, by) in sorted(tot.items()):
        print(f"{d}: {n} copies, {1e-6 * dur:.2f} ms in flight, {1e-6 * ov:.2f} ms ({100.0 * ov / max(1, dur):.1f} %) of it while kernels of other batches run")
else:
    print("trace files not found", k, m)
PY
i=0
for grp in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_MFMA" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" \
           "SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" \
           "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $O/pmc/g$i -o p -- python3 $R/bench.py --steps 1 --warmup 1 --reads 2000 --no-extras --no-cpu-baseline > $O/pmc_g$i.log 2>&1 || echo "pmc group $i failed"
done
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: [0.0, 0])
for f in sorted(glob.glob("$O/pmc/g*/**/*counter_collection.csv", recursive=True)):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0].split("<")[0][-28:] + ("<13" if "<13" in row["Kernel_Name"] or "ILi13" in row["Kernel_Name"] else "")
        a = agg[(k, row["Counter_Name"])]
        a[0] += float(row["Counter_Value"]); a[1] += 1
with open("$O/pmc_summary.txt", "w") as o:
    for (k, c), (v, n) in sorted(agg.items()):
        o.write(f"{k}\t{c}\tmean_per_launch\t{v / n:.6g}\tlaunches\t{n}\n")
print(open("$O/pmc_summary.txt").read()[:6000])
PY
cat $O/overlap.txt
head -30 $O/kernel_stats.csv
