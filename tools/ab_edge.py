"""Same-process A/B of the two edge kernels (engine option edge_impl: 0 = edge_kernel, hm_trunk.hip;
1 = edge2_kernel, hm_edge2.hip): device ms per resident-slab run, and byte identity of the calls."""
import sys

sys.path.insert(0, ".")
from hifimeth_amd import MethylationCaller  # noqa: E402
from hifimeth_amd.synth import synth_reads  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1200
reads = synth_reads(n, seed=5)
mc = MethylationCaller(device=0, timing=True)
mc.set_option("trunk", 1)
mc.submit_all(reads)
mc.upload()
out = {}
for rep in range(3):
    for impl in (0, 1):
        mc.set_option("edge_impl", impl)
        mc.run(); mc.sync()
        if rep == 0:
            out[impl] = mc.fetch().copy()
        mc.timing(reset=True)
        for _ in range(3):
            mc.run()
        mc.sync()
        tm = mc.timing()
        print(f"edge_impl {impl}: trunk {sum(tm['trunk_ms']) / 3:8.2f} ms  edge {sum(tm['edge_ms']) / 3:7.2f}  tail {sum(tm['tail_ms']) / 3:7.2f}  "
              f"by context {[round(x / 3, 2) for x in tm['tail_ms']]}   sites {mc.num_sites(3)}", flush=True)
same = out[0].tobytes() == out[1].tobytes()
print("calls byte-identical:", same, len(out[0]))
if not same:
    import numpy as np
    d = np.abs(out[0]["p"] - out[1]["p"])
    bad = np.nonzero(out[0]["p"] != out[1]["p"])[0]
    print("differing records:", len(bad), "max |dp|", float(d.max()), "first:", bad[:10], out[0][bad[:5]], out[1][bad[:5]])
    sys.exit(1)
