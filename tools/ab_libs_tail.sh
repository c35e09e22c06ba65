#!/bin/bash
# same-box A/B of engine builds on the resident-slab tail test: tools/ab_libs_tail.sh lib1.so lib2.so ...   (tail ms by context, byte identity vs tail_impl 0)
for rep in 1 2; do for l in "$@"; do echo "== $l"; HM_LIB_PATH=$PWD/$l python tools/ab_tail.py 1200 2>&1 | grep -E 'tail_impl 1|identical|differing' | tail -3; done; done
