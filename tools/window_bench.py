import sys
sys.path.insert(0, ".")
from hifimeth_amd import MethylationCaller
from hifimeth_amd.synth import synth_reads
reads = synth_reads(96, seed=3)
with MethylationCaller(device=0, timing=True) as mc:
    mc.submit_all(reads); mc.upload(); mc.run(); mc.sync()
    sc = [mc.num_sites(c) for c in range(3)]
    big = max(range(3), key=lambda c: sc[c])
    for g in (None,):
        mc.windows(big, 0, sc[big], fetch=False)
        mc.timing(reset=True)
        for _ in range(5):
            mc.windows(big, 0, sc[big], fetch=False)
        tw = mc.timing()
        per_site = 401 * 8 * 4 + 401 * 5 + 12
        print("sites", sc[big], "ms", tw["window_ms"] / tw["window_launches"], "GB/s", tw["window_sites"] * per_site / (tw["window_ms"] * 1e-3) / 1e9)
