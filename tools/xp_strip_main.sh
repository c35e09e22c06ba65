cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
echo "== main-only (conv5 + conv6, timing only) vs full strip tail vs tail_impl 2 (split: main + head)"
HM_LIB_PATH=$R/hifimeth_amd/lib_xp_mainonly.so timeout -k 10 200 python tools/ab_tail.py 1200 3 2>&1 | grep "tail_impl" | tail -2
timeout -k 10 200 python tools/ab_tail.py 1200 3,2 2>&1 | grep "tail_impl" | tail -3
mkdir -p gpurun_out/xp_head
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/xp_head -o h -- python3 tools/ab_tail.py 600 2 > /dev/null 2>&1
python3 - <<PY
import csv,glob
for f in glob.glob("gpurun_out/xp_head/**/*kernel_stats.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "tail_" in row["Name"]: print(row["Name"][:60], row["Calls"], row["TotalDurationNs"], row["AverageNs"])
PY
