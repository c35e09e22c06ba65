"""Experiment: does one GPU do more with TWO engines side by side, each on half the CUs, than with one engine on all of them?
(Different kernels then overlap: while one engine's trunk keeps the matrix pipes of its CUs -- and the chip's power budget --
busy, the other's edge / tail kernels, which leave the pipes idle two thirds of the time, run beside it.)
    python tools/ab_two_engines.py [reads per slab] [slabs]"""
import sys
import threading
import time

sys.path.insert(0, ".")
from hifimeth_amd import MethylationCaller  # noqa: E402
from hifimeth_amd.caller import ReadBlock  # noqa: E402
from hifimeth_amd.synth import synth_slab  # noqa: E402

n_reads = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
n_slabs = int(sys.argv[2]) if len(sys.argv) > 2 else 8
blocks = [ReadBlock(synth_slab(n_reads, seed=100 + i)) for i in range(4)]


def run(engines, cus):
    mcs = []
    for _ in range(engines):
        m = MethylationCaller(device=0)
        m.set_option("num_cu", cus)
        m.stage_threads = 4
        mcs.append(m)
    sites = [0] * engines

    def work(i, k):
        def on_batch(_k, b, calls):
            sites[i] += len(calls)
        mcs[i].stream((blocks[j % 4] for j in range(k)), on_batch=on_batch)

    for i in range(engines):
        work(i, 1)   # warm-up: buffers
    sites = [0] * engines
    t0 = time.perf_counter()
    ts = [threading.Thread(target=work, args=(i, n_slabs // engines)) for i in range(engines)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    dt = time.perf_counter() - t0
    for m in mcs:
        m.close()
    return sum(sites) / dt


for engines, cus in ((1, 256), (2, 128), (2, 256), (1, 256), (2, 128), (3, 86), (2, 160)):
    print(f"{engines} engine(s) x {cus:3d} workgroups per launch: {run(engines, cus) / 1e6:6.2f} M sites/s", flush=True)
