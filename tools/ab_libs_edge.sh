#!/bin/bash
# same-box A/B of engine builds on the resident-slab edge test: tools/ab_libs_edge.sh lib1.so lib2.so ...  (edge ms, byte identity vs edge_kernel)
for rep in 1 2; do for l in "$@"; do echo "== $l"; HM_LIB_PATH=$PWD/$l timeout -k 10 200 python tools/ab_edge.py 1200 2>&1 | grep -E 'edge_impl 1|identical|differing' | tail -3 || exit 1; done; done
