"""Same-process A/B of the two trunk kernels (engine option trunk_impl): device ms of trunk / edge / tail per resident slab run."""
import sys

sys.path.insert(0, ".")
from hifimeth_amd import MethylationCaller  # noqa: E402
from hifimeth_amd.synth import synth_reads  # noqa: E402

reads = synth_reads(1200, seed=5)
mc = MethylationCaller(device=0, timing=True)
mc.set_option("trunk", 1)
mc.submit_all(reads)
mc.upload()
for rep in range(3):
    for impl in (0, 1, 2):
        mc.set_option("trunk_impl", impl)
        mc.run(); mc.sync()
        mc.timing(reset=True)
        for _ in range(3):
            mc.run()
        mc.sync()
        tm = mc.timing()
        print(f"impl {impl}: trunk {sum(tm['trunk_ms']) / 3:8.2f} ms  edge {sum(tm['edge_ms']) / 3:7.2f}  tail {sum(tm['tail_ms']) / 3:7.2f}   sites {mc.num_sites(3)}")
