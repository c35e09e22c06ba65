"""Diagnostic: where a step of the sliding-window trunk (trunk3_kernel) goes, and the clock it runs at.  Needs the stamped build
(make -C hifimeth_amd/csrc stamp) and HM_LIB_PATH=hifimeth_amd/libhifimeth_hip_stamp.so.  s_memtime ticks per step of workgroup 0, one
column per wave: per layer the time from its barrier to the return of its streaming-conv call, and the wait at the next barrier."""
import ctypes as C
import sys

import numpy as np

sys.path.insert(0, ".")
from hifimeth_amd import MethylationCaller, _lib  # noqa: E402
from hifimeth_amd.synth import synth_reads  # noqa: E402

reads = synth_reads(int(sys.argv[1]) if len(sys.argv) > 1 else 1200, seed=5)
mc = MethylationCaller(device=0, timing=True)
mc.set_option("trunk", 1)
mc.set_option("trunk_impl", 3)
mc.submit_all(reads)
mc.upload()
mc.run()
mc.sync()
fn = C.CDLL(_lib.LIB_PATH).hm_debug_trunk3_stamps
fn.argtypes = [C.c_void_p, C.c_int]
fn(None, 1)
mc.timing(reset=True)
for _ in range(3):
    mc.run()
mc.sync()
buf = np.zeros((8, 24), np.uint64)
assert fn(buf.ctypes.data, 0) == 0
n = float(buf[0, 16])
print("steps of workgroup 0 (3 runs x 3 contexts, CHH two views; warm-up steps included):", int(n), " sites", mc.num_sites(3))
tot = np.zeros(4)
for l in range(4):
    for ph, nm in ((0, "layer"), (3, "barrier")):
        v = buf[:4, 4 * l + ph].astype(float) / n
        tot += v
        print(f"conv{l + 1} {nm:9s} " + " ".join(f"{x:7.0f}" for x in v))
print("sum            " + " ".join(f"{x:7.0f}" for x in tot))
if buf[0, 18]:
    print("constant steps of workgroup 0:", int(buf[0, 18]), " ticks each " + " ".join(f"{float(buf[w, 17]) / float(buf[w, 18]):7.0f}" for w in range(4)))
print(f"in-kernel clock of workgroup 0's step loops: {float(buf[0, 22]) / max(float(buf[0, 23]), 1) * 0.1:.3f} GHz (s_memtime / s_memrealtime x 100 MHz)")
tm = mc.timing()
print("trunk_ms per run", [round(x / 3, 2) for x in tm["trunk_ms"]], "listed-row steps", tm["trunk_list_steps"], "tiles", [p // 112 for p in tm["trunk_positions"]])
