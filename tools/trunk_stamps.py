"""Diagnostic: where a trunk_kernel tile's cycles go.  Needs the stamped build (make -C hifimeth_amd/csrc stamp) and
HM_LIB_PATH=hifimeth_amd/libhifimeth_hip_stamp.so.  Prints, per layer, the mean shader-clock cycles per tile that
workgroup 0's waves spend in: prologue (barrier -> first loads issued), k-loop, epilogue, waiting at the next barrier."""
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
L = _lib.load() if hasattr(_lib, "load") else C.CDLL(_lib.LIB_PATH)
fn = C.CDLL(_lib.LIB_PATH).hm_debug_trunk_stamps
fn.argtypes = [C.c_void_p, C.c_int]
buf = np.zeros((8, 24), np.uint64)
fe = C.CDLL(_lib.LIB_PATH).hm_debug_edge_stamps
fe.argtypes = [C.c_void_p, C.c_int]
fe(None, 1)
ft = C.CDLL(_lib.LIB_PATH).hm_debug_tail_stamps
ft.argtypes = [C.c_void_p, C.c_int]
ft(None, 1)
fn(None, 1)
for _ in range(3):
    mc.run()
mc.sync()
assert fn(buf.ctypes.data, 0) == 0
n = float(buf[0, 16])
print("tiles of workgroup 0:", int(n), " sites", mc.num_sites(3))
streaming = not buf[:, 1].any()  # trunk2_kernel stamps a layer as a whole (slot 0) and the wait at its barrier (slot 3)
names = ["layer", "", "", "barrier"] if streaming else ["prologue", "k-loop", "epilogue", "barrier"]
print("s_memtime ticks per tile (shares, not clock cycles), one column per wave of workgroup 0" + (" (streaming form: 4 waves)" if streaming else ""))
tot = np.zeros(8)
for l in range(4):
    for ph in range(4):
        v = buf[:, 4 * l + ph].astype(float) / n
        tot += v
        if names[ph]:
            print(f"conv{l + 1} {names[ph]:9s} " + " ".join(f"{x:7.0f}" for x in v))
if buf[:, 17:22].any():
    print("L1 parts [prologue reads, stream, final epilogue]:", (buf[:4, 17:20].astype(float) / n).round(0).tolist(), 'group starts', (buf[:4, 20:22].astype(float) / n).round(0).tolist())
print("sum            " + " ".join(f"{x:7.0f}" for x in tot))
tm = mc.timing()
print("trunk_ms", tm["trunk_ms"], "positions", tm["trunk_positions"])

eb = np.zeros((8, 12), np.uint64)
assert fe(eb.ctypes.data, 0) == 0
ne = float(eb[0, 9])
print("edge passes of workgroup 0:", int(ne))
for i, nm in enumerate(["req2+conv1+req34", "barrier", "stage2+bar", "conv2", "barrier", "stage3+bar", "conv3+bar", "stage4+bar", "conv4|prepare"]):
    print(f"edge {nm:18s} " + " ".join(f"{x:7.0f}" for x in eb[:, i].astype(float) / ne))
print("edge sum           " + " ".join(f"{x:7.0f}" for x in eb[:, :9].astype(float).sum(1) / ne))

tb = np.zeros((8, 16), np.uint64)
assert ft(tb.ctypes.data, 0) == 0
nt = float(tb[0, 10])
print("tail passes of workgroup 0:", int(nt), " fc batches", int(tb[0, 9]))
for i, nm in enumerate(["conv5", "barrier", "conv6|stage", "barrier", "conv7|stage", "barrier", "conv8|stage"]):
    print(f"tail {nm:14s} " + " ".join(f"{x:7.0f}" for x in tb[:, i].astype(float) / nt))
print("tail conv5: barrier -> prologue loads issued " + " ".join(f"{x:7.0f}" for x in tb[:, 11].astype(float) / nt))
print("tail conv5: k-loop                          " + " ".join(f"{x:7.0f}" for x in tb[:, 12].astype(float) / nt))
print("tail fc1+fc2 per pass " + " ".join(f"{x:7.0f}" for x in tb[:, 8].astype(float) / nt))
print("tail sum            " + " ".join(f"{x:7.0f}" for x in (tb[:, :7].astype(float).sum(1) + tb[:, 8]) / nt))
