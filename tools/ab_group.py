"""Same-process sweep of the engine option group_bases (bases per trunk read group): device ms per resident slab run."""
import sys
import time

sys.path.insert(0, ".")
import torch  # noqa: E402
from hifimeth_amd import MethylationCaller  # noqa: E402
from hifimeth_amd.synth import synth_reads  # noqa: E402

reads = synth_reads(2400, seed=5)
for gb in (1 << 21, 1 << 22, 1 << 23, 1 << 24, 1 << 21):
    mc = MethylationCaller(device=0, timing=True)
    mc.set_option("trunk", 1)
    mc.set_option("group_bases", gb)
    mc.submit_all(reads)
    mc.upload()
    mc.run(); mc.sync()
    mc.timing(reset=True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(3):
        mc.run()
    mc.sync(); dt = (time.perf_counter() - t0) / 3
    tm = mc.timing()
    print(f"group_bases {gb:8d}: wall {dt * 1e3:7.2f} ms  trunk {sum(tm['trunk_ms']) / 3:7.2f}  edge {sum(tm['edge_ms']) / 3:6.2f}  tail {sum(tm['tail_ms']) / 3:6.2f}  launches {sum(tm['trunk_launches']) // 3}")
    mc.close()
