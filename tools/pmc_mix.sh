cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
i=0
for grp in "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAVE_CYCLES" "SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_BRANCH"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $R/gpurun_out/pmc2/g$i -o p -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/pmc2_g$i.log 2>&1 || echo "group $i failed"
done
