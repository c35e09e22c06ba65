"""Diagnostic: where a pass of tail_kernel_p (the strip tail, hm_tail_p.hip) goes (stamped build: make -C hifimeth_amd/csrc stamp;
HM_LIB_PATH=hifimeth_amd/libhifimeth_hip_stamp.so python tools/tailp_stamps.py [reads]).  s_memtime ticks per pass of workgroup 0
(up to 16 sites), one column per wave; shares of a pass, not clock cycles (the tick rate depends on the load).  Context CHH only."""
import ctypes as C
import sys

import numpy as np

sys.path.insert(0, ".")
from hifimeth_amd import MethylationCaller, _lib  # noqa: E402
from hifimeth_amd.synth import synth_reads  # noqa: E402

reads = synth_reads(int(sys.argv[1]) if len(sys.argv) > 1 else 400, seed=5)
for impl in (3, 1):
    mc = MethylationCaller(contexts="chh", device=0, timing=True)
    mc.set_option("trunk", 1)
    mc.set_option("tail_impl", impl)
    mc.submit_all(reads)
    mc.upload()
    mc.run()
    mc.sync()
    fn = getattr(C.CDLL(_lib.LIB_PATH), "hm_debug_tailp_stamps" if impl == 3 else "hm_debug_tailr_stamps")
    fn.argtypes = [C.c_void_p, C.c_int]
    fn(None, 1)
    mc.timing(reset=True)
    for _ in range(3):
        mc.run()
    mc.sync()
    buf = np.zeros((4, 16), np.uint64)
    assert fn(buf.ctypes.data, 0) == 0
    n = float(buf[0, 11])
    tm = mc.timing()
    if impl == 3:
        print(f"tail_kernel_p: passes of workgroup 0: {int(n)}, sites taken {int(buf[0, 12])} = {float(buf[0, 12]) / max(n, 1):.2f} per pass of 16 slots; CHH sites {mc.num_sites(2)}")
        names = ["plan + conv5", "barrier", "conv6 (+early DMA)", "barrier", "conv7", "drain + barrier", "late DMA + conv8", "fc loads + barrier", "fc1", "fc2 + softmax"]   # (the last three: 0 since fc1 .. softmax left for tail_fc_kernel)
        tot = np.zeros(4)
        for i, nm in enumerate(names):
            v = buf[:, i].astype(float) / n
            tot += v
            print(f"{nm:30s} " + " ".join(f"{x:7.0f}" for x in v))
        print(f"{'sum (ticks per pass)':30s} " + " ".join(f"{x:7.0f}" for x in tot))
        print(f"{'ticks per site':30s} " + " ".join(f"{x * n / max(float(buf[0, 12]), 1):7.0f}" for x in tot))
        print(f"{'whole loop, ticks per pass':30s} " + " ".join(f"{float(x) / n:7.0f}" for x in buf[:, 13]) + "   (with the drain and the wait at the loop's top barrier, which no phase above holds)")
    else:
        tot = buf[:, :10].astype(float).sum(axis=1) / n
        print(f"tail_kernel_r: passes of workgroup 0: {int(n)} (8 sites each); ticks per pass " + " ".join(f"{x:7.0f}" for x in tot)
              + "; per site " + " ".join(f"{x / 8:7.0f}" for x in tot))
    print(f"in-kernel clock of workgroup 0's pass loop: {float(buf[0, 13]) / max(float(buf[0, 14]), 1) * 0.1:.3f} GHz (s_memtime / s_memrealtime x 100 MHz)")
    print("tail_ms per run", [round(x / 3, 3) for x in tm["tail_ms"]], "\n")
    mc.close()
