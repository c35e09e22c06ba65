#!/bin/bash
# the edge kernel's layer split, measured on ONE stage (tools/micro/edge_stage.hip): bash tools/edge_stage.sh > profiles/r05_edge_stage_microbench.txt
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/edge_stage; mkdir -p $O
hipcc --offload-arch=gfx950 -O3 -std=c++20 -I $R/hifimeth_amd/csrc $R/tools/micro/edge_stage.hip -o $O/edge_stage 2> $O/build.log || { echo "build failed"; tail -5 $O/build.log; exit 1; }
timeout -k 10 120 $O/edge_stage 4194304
