"""Diagnostic: where a tile of tail_head_p_kernel (hm_tail_head.hip: conv7 .. softmax of 16 sites) goes (stamped build: make -C hifimeth_amd/csrc stamp;
HM_LIB_PATH=hifimeth_amd/libhifimeth_hip_stamp.so python tools/tailhead_stamps.py [reads]).  s_memtime ticks per tile of workgroup 0, one column per wave."""
import ctypes as C
import sys

import numpy as np

sys.path.insert(0, ".")
from hifimeth_amd import MethylationCaller, _lib  # noqa: E402
from hifimeth_amd.synth import synth_reads  # noqa: E402

reads = synth_reads(int(sys.argv[1]) if len(sys.argv) > 1 else 400, seed=5)
mc = MethylationCaller(contexts="chh", device=0, timing=True)
mc.set_option("trunk", 1)
mc.submit_all(reads)
mc.upload()
mc.run()
mc.sync()
fn = C.CDLL(_lib.LIB_PATH).hm_debug_tailhead_stamps
fn.argtypes = [C.c_void_p, C.c_int]
fn(None, 1)
mc.timing(reset=True)
for _ in range(3):
    mc.run()
mc.sync()
buf = np.zeros((4, 16), np.uint64)
assert fn(buf.ctypes.data, 0) == 0
n = float(buf[0, 11])
print(f"tail_head_p_kernel: tiles of workgroup 0: {int(n)}; CHH sites {mc.num_sites(2)}")
names = ["DMA issue + conv7", "barrier", "conv8", "barrier", "fc1", "wait for the next tile's rows", "barrier", "fc2 + softmax + stores"]
tot = np.zeros(4)
for i, nm in enumerate(names):
    v = buf[:, i].astype(float) / max(n, 1)
    tot += v
    print(f"{nm:32s} " + " ".join(f"{x:7.0f}" for x in v))
print(f"{'sum (ticks per tile)':32s} " + " ".join(f"{x:7.0f}" for x in tot))
print(f"{'whole loop, ticks per tile':32s} " + " ".join(f"{float(x) / max(n, 1):7.0f}" for x in buf[:, 13]))
print(f"in-kernel clock: {float(buf[0, 13]) / max(float(buf[0, 14]), 1) * 0.1:.3f} GHz")
print("tail_ms per run", [round(x / 3, 3) for x in mc.timing()["tail_ms"]])
mc.close()
