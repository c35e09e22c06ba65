#!/bin/bash
# WRITE_SIZE per byte written, by store width and pattern (run on the GPU box): bash tools/write_calib.sh > profiles/r05_write_size_calibration.txt
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/write_calib; mkdir -p $O
hipcc --offload-arch=gfx950 -O2 $R/tools/micro/write_calib.hip -o $O/write_calib || exit 1
timeout -k 10 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc -o w -- $O/write_calib > $O/run.log 2>&1 || { echo "rocprofv3 failed"; tail -5 $O/run.log; exit 1; }
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(list)
for f in glob.glob("$O/pmc/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if row["Counter_Name"] == "WRITE_SIZE":
            agg[row["Kernel_Name"].split("(")[0]].append(float(row["Counter_Value"]))
B = float(1 << 30)
print("# WRITE_SIZE (rocprofv3 --pmc WRITE_SIZE, gfx950) against the bytes a kernel really writes: 1 GiB per launch, three launches each")
print("# tools/micro/write_calib.hip; the counter's unit follows the tool's derived-metric definition (KB: x 1024 = bytes)")
for k, v in sorted(agg.items()):
    m = sum(v) / len(v)
    print(f"{k:14s} WRITE_SIZE mean {m:.6g}  -> x1024 / bytes = {m * 1024 / B:.4f}   (raw / bytes = {m / B:.6f})")
PY
