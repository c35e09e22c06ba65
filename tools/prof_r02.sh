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
BENCH="python3 $R/bench.py --steps 4 --warmup 2 --reads 4000 --no-extras --no-cpu-baseline"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o s -- $BENCH > $O/stats.log 2>&1 || echo "stats run failed"
cp $(find $O/stats -name '*kernel_stats.csv' | head -1) $O/kernel_stats.csv 2>/dev/null
timeout -k 10 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $O/trace -o t -- $BENCH > $O/trace.log 2>&1 || echo "trace run failed"
python3 - <<PY > $O/overlap.txt
# Copies vs compute.  On this runtime pinned H2D / D2H copies mostly run as blit kernels (__amd_rocclr_copyBuffer) on
# their own hardware queue, a few as SDMA transfers (memory-copy trace): both are matched against the union of the
# engine's compute kernels (hm::*).
import csv, glob, bisect
k = glob.glob("$O/trace/**/*kernel_trace.csv", recursive=True)
m = glob.glob("$O/trace/**/*memory_copy_trace.csv", recursive=True)
rows = list(csv.DictReader(open(k[0]))) if k else []
comp = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows if "hm::" in r["Kernel_Name"] or "_ZN2hm" in r["Kernel_Name"])
copies = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "blit kernel") for r in rows if "copyBuffer" in r["Kernel_Name"]]
if m:
    copies += [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Direction"].replace("MEMORY_COPY_", "sdma ")) for r in csv.DictReader(open(m[0]))]
if comp:
    merged = []
    for a, b in comp:
        if merged and a <= merged[-1][1]:
            merged[-1][1] = max(merged[-1][1], b)
        else:
            merged.append([a, b])
    t0, t1 = comp[0][0], comp[-1][1]
    busy = sum(b - a for a, b in merged)
    print(f"compute span {1e-6 * (t1 - t0):.1f} ms, compute kernels busy {1e-6 * busy:.1f} ms ({100.0 * busy / (t1 - t0):.1f} % of the span; the run is traced, the host side is slower than in the plain bench)")
    starts = [a for a, _ in merged]
    tot = ovl = 0
    print("copies of >= 0.5 ms inside the compute span (start ms, duration ms, overlapped with compute ms):")
    for a, b, kind in sorted(copies):
        if a < t0 or a > t1:
            continue
        o = 0
        i = max(0, bisect.bisect_right(starts, a) - 1)
        while i < len(merged) and merged[i][0] < b:
            o += max(0, min(b, merged[i][1]) - max(a, merged[i][0]))
            i += 1
        tot += b - a
        ovl += o
        if b - a >= 500000:
            print(f"  {kind:24s} {1e-6 * (a - t0):9.1f} {1e-6 * (b - a):7.2f} {1e-6 * o:7.2f}")
    print(f"all copies inside the span: {1e-6 * tot:.1f} ms, {1e-6 * ovl:.1f} ms ({100.0 * ovl / max(1, tot):.0f} %) under compute kernels of other batches")
    print("(copies that do not overlap sit where the host had nothing queued: the warm-up / timed boundary -- a barrier -- and the gaps the tracer's host overhead opens)")
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

# 4. host side alone (no GPU work): BGZF inflate + record parsing + staging copy, one rank, and the CLI end to end
python3 - <<PY
import os, sys, time, subprocess, json
R = "$R"
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import bamutil
from hifimeth_amd.synth import synth_slab
t = time.time(); reads = synth_slab(6000, seed=5); src = "/tmp/prof_r02_in.bam"; bamutil.reads_to_bam(src, reads, level=1)
mb, mbases = os.path.getsize(src) / 1e6, sum(r.l_qseq for r in reads) / 1e6
out = [f"synthetic BAM: {len(reads)} reads, {mbases:.1f} Mbases, {mb:.1f} MB (level-1 BGZF), built in {time.time() - t:.1f} s"]
cli = os.path.join(R, "hifimeth_amd", "bin", "hifimeth-hip")
for th in (1, 4, 8, 16):
    out.append("stagebench " + subprocess.check_output([cli, "stagebench", "-t", str(th), src]).decode().strip())
out.append("stagebench rank 1 of 2: " + subprocess.check_output([cli, "stagebench", "-t", "8", "-R", "1/2", src]).decode().strip())
for b in ("1000", "3000"):
    t = time.time()
    p = subprocess.run([cli, "call", "-b", b, "-t", "16", src, "/tmp/prof_r02_out.bam"], stderr=subprocess.PIPE, text=True)
    dt = time.time() - t
    tail = [l.strip() for l in p.stderr.splitlines() if "##" in l]
    out.append(f"call -b {b} -t 16: exit {p.returncode}, {dt:.2f} s wall (engine start-up included); " + " | ".join(tail))
t = time.time()
p = subprocess.run([cli, "call", "-c", "cpg", "-b", "3000", "-t", "16", src, "/tmp/prof_r02_out.bam"], stderr=subprocess.PIPE, text=True)
out.append(f"call -c cpg -b 3000: exit {p.returncode}, {time.time() - t:.2f} s wall; " + " | ".join(l.strip() for l in p.stderr.splitlines() if "##" in l))
open("$O/host_e2e.txt", "w").write("\n".join(out) + "\n")
print("\n".join(out))
PY
