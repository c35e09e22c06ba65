"""Diagnostic: where a pass of the split tail's two kernels goes (stamped build: make -C hifimeth_amd/csrc stamp;
HM_LIB_PATH=hifimeth_amd/libhifimeth_hip_stamp.so python tools/tails_stamps.py).  s_memtime ticks per pass of workgroup 0 (main: 8
sites, head: 16 sites), one column per wave; shares of a pass, not clock cycles (the tick rate depends on the load)."""
import ctypes as C
import sys

import numpy as np

sys.path.insert(0, ".")
from hifimeth_amd import MethylationCaller, _lib  # noqa: E402
from hifimeth_amd.synth import synth_reads  # noqa: E402

reads = synth_reads(int(sys.argv[1]) if len(sys.argv) > 1 else 400, seed=5)
mc = MethylationCaller(device=0, timing=True)
mc.set_option("trunk", 1)
mc.set_option("tail_impl", 2)
mc.submit_all(reads)
mc.upload()
mc.run()
mc.sync()
fn = C.CDLL(_lib.LIB_PATH).hm_debug_tails_stamps
fn.argtypes = [C.c_void_p, C.c_int]
fn(None, 1)
mc.timing(reset=True)
for _ in range(3):
    mc.run()
mc.sync()
buf = np.zeros((2, 4, 16), np.uint64)
assert fn(buf.ctypes.data, 0) == 0
names = [["loop-top barrier", "conv5", "table", "conv6 (+gather)", "drain"],
         ["loop-top barrier", "conv7 (+input DMA)", "barrier", "conv8 (+input DMA)", "barrier", "fc1", "barrier", "fc2 + softmax", "drain"]]
for k, kn in enumerate(("main (8 sites per pass)", "head (16 sites per pass)")):
    n = float(buf[k, 0, 15])
    print(f"{kn}: passes of workgroup 0: {int(n)}   sites {mc.num_sites(3)}")
    tot = np.zeros(4)
    for i, nm in enumerate(names[k]):
        v = buf[k, :, i].astype(float) / max(n, 1)
        tot += v
        print(f"  {nm:28s} " + " ".join(f"{x:7.0f}" for x in v))
    print(f"  {'sum':28s} " + " ".join(f"{x:7.0f}" for x in tot))
    print(f"  in-kernel clock of workgroup 0's pass loops: {float(buf[k, 0, 13]) / max(float(buf[k, 0, 14]), 1) * 0.1:.3f} GHz (s_memtime / s_memrealtime x 100 MHz)")
tm = mc.timing()
print("tail_ms per run", [round(x / 3, 2) for x in tm["tail_ms"]])
