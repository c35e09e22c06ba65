"""Diagnostic: where an edge2_kernel pass's ticks go.  Needs the stamped build (make -C hifimeth_amd/csrc stamp) and
HM_LIB_PATH=hifimeth_amd/libhifimeth_hip_stamp.so.  Prints the mean s_memtime ticks per pass (32 sites) that the waves of
workgroup 0 spend in each phase -- shares of a pass, not a fixed unit of time."""
import ctypes as C
import sys

import numpy as np

sys.path.insert(0, ".")
from hifimeth_amd import MethylationCaller, _lib  # noqa: E402
from hifimeth_amd.synth import synth_reads  # noqa: E402

reads = synth_reads(400, seed=5)
mc = MethylationCaller(device=0, timing=True)
mc.set_option("trunk", 1)
mc.submit_all(reads)
mc.upload()
mc.run()
mc.sync()
fn = C.CDLL(_lib.LIB_PATH).hm_debug_edge2_stamps
fn.argtypes = [C.c_void_p, C.c_int]
buf = np.zeros((8, 16), np.uint64)
fn(None, 1)
mc.timing(reset=True)
for _ in range(3):
    mc.run()
mc.sync()
assert fn(buf.ctypes.data, 0) == 0
n = buf[:, 10].astype(float)
print("passes of workgroup 0:", int(n[0]), " sites", mc.num_sites(3))
names = ["conv1 (+ next descriptors)", "barrier", "conv2 (+ conv3's DMAs, weights)", "drain + barrier", "epilogue + barrier",
         "conv3 (+ conv4's DMAs, weights)", "drain + barrier", "epilogue + barrier", "conv4 | next pass's rows + DMAs", "drain + top barrier"]
tot = np.zeros(8)
for i, nm in enumerate(names):
    v = buf[:, i].astype(float) / n
    tot += v
    print(f"{nm:34s}" + " ".join(f"{x:7.0f}" for x in v))
print(f"{'sum':34s}" + " ".join(f"{x:7.0f}" for x in tot))
print("edge_ms per run", [round(x / 3, 2) for x in mc.timing()["edge_ms"]])
