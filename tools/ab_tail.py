"""Same-process A/B of the gathering tails (engine option tail_impl: 0 = tail_kernel_h, weights streamed per pass;
1 = tail_kernel_r, weights resident; 2 = the split tail, hm_tail_s.hip): device ms per resident-slab run, and byte identity of the calls."""
import sys

sys.path.insert(0, ".")
from hifimeth_amd import MethylationCaller  # noqa: E402
from hifimeth_amd.synth import synth_reads  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1200
IMPLS = tuple(int(x) for x in sys.argv[2].split(",")) if len(sys.argv) > 2 else (0, 1, 2)
reads = synth_reads(n, seed=5)
mc = MethylationCaller(device=0, timing=True)
mc.set_option("trunk", 1)
mc.submit_all(reads)
mc.upload()
out = {}
for rep in range(3):
    for impl in IMPLS:
        mc.set_option("tail_impl", impl)
        mc.run(); mc.sync()
        if rep == 0:
            out[impl] = mc.fetch().copy()
        mc.timing(reset=True)
        for _ in range(3):
            mc.run()
        mc.sync()
        tm = mc.timing()
        print(f"tail_impl {impl}: trunk {sum(tm['trunk_ms']) / 3:8.2f} ms  edge {sum(tm['edge_ms']) / 3:7.2f}  tail {sum(tm['tail_ms']) / 3:7.2f}  "
              f"by context {[round(x / 3, 2) for x in tm['tail_ms']]}   sites {mc.num_sites(3)}", flush=True)
ok = True
for impl in IMPLS[1:]:
    a, b = out[IMPLS[0]], out[impl]
    same = a.tobytes() == b.tobytes()
    print(f"calls of tail_impl {impl} byte-identical to tail_impl {IMPLS[0]}:", same, len(a))
    if not same:
        import numpy as np
        ok = False
        if len(a) != len(b):
            print("record counts differ", len(a), len(b))
            continue
        d = np.abs(a["p"] - b["p"])
        bad = np.nonzero(a["p"] != b["p"])[0]
        print("differing records:", len(bad), "max |dp|", float(d.max()), "first:", bad[:10], a[bad[:5]], b[bad[:5]])
sys.exit(0 if ok else 1)
