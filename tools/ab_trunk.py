"""Same-process A/B of the trunk kernels (engine option trunk_impl: 0 = 8-wave ConvH form, 1 = streaming 4-wave, 2 = streaming 8-wave,
3 = sliding window): device ms of trunk / edge / tail per resident slab run, and byte identity of the calls.
python tools/ab_trunk.py [reads] [impls, e.g. 1,3]"""
import sys

sys.path.insert(0, ".")
from hifimeth_amd import MethylationCaller  # noqa: E402
from hifimeth_amd.synth import synth_reads  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1200
IMPLS = tuple(int(x) for x in sys.argv[2].split(",")) if len(sys.argv) > 2 else (1, 3)
reads = synth_reads(n, seed=5)
mc = MethylationCaller(device=0, timing=True)
mc.set_option("trunk", 1)
mc.submit_all(reads)
mc.upload()
out = {}
for rep in range(3):
    for impl in IMPLS:
        mc.set_option("trunk_impl", impl)
        mc.run(); mc.sync()
        if rep == 0:
            out[impl] = mc.fetch().copy()
        mc.timing(reset=True)
        for _ in range(3):
            mc.run()
        mc.sync()
        tm = mc.timing()
        print(f"impl {impl}: trunk {sum(tm['trunk_ms']) / 3:8.2f} ms  edge {sum(tm['edge_ms']) / 3:7.2f}  tail {sum(tm['tail_ms']) / 3:7.2f}   sites {mc.num_sites(3)}"
              f"  steps: all {[x // 336 for x in tm['trunk_positions']]} listed {[x // 3 for x in tm['trunk_list_steps']]} constant {[x // 3 for x in tm.get('trunk_const_steps', [0, 0, 0])]}", flush=True)
ok = True
for impl in IMPLS[1:]:
    a, b = out[IMPLS[0]], out[impl]
    same = a.tobytes() == b.tobytes()
    print(f"calls of trunk_impl {impl} byte-identical to trunk_impl {IMPLS[0]}:", same, len(a))
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
