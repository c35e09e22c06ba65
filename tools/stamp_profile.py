#!/usr/bin/env python3
"""Diagnostic: where the front kernel's wave-cycles go (in-kernel s_memtime stamps, 8-wave build)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from hifimeth_amd import MethylationCaller
from hifimeth_amd.synth import synth_reads

names = ["window build", "barrier"] + [f"conv{l} {w}" for l in (1, 2, 3, 4) for w in ("prologue", "k-loop", "epilogue", "pad+barrier")]
reads = synth_reads(int(sys.argv[1]) if len(sys.argv) > 1 else 48, seed=20250220)
with MethylationCaller(timing=True) as mc:
    mc.set_option("front_waves", 8)
    mc.set_option("precision", int(os.environ.get("HM_PRECISION", "1")))
    mc.submit_all(reads); mc.upload(); mc.run(); mc.sync()
    mc.set_option("stamps", 1)
    mc.run(); mc.sync()
    st = mc.stamps()
    ns = len(st) // 8
    per_wave = [st[w * ns:(w + 1) * ns] for w in range(8)]
    nsite_wg = mc.num_sites(3) / 256
    tot = sum(st)
    print(f"{'phase':18s} {'all %':>7s} {'cyc/site':>9s} | waves 0-3 | waves 4-7")
    for i, n in enumerate(names):
        v = sum(pw[i] for pw in per_wave)
        lo = sum(pw[i] for pw in per_wave[:4]) / 4 / 256 / nsite_wg
        hi = sum(pw[i] for pw in per_wave[4:]) / 4 / 256 / nsite_wg
        print(f"{n:18s} {100.0 * v / tot:6.2f}  {v / (256 * 8) / nsite_wg:9.0f} | {lo:9.0f} | {hi:9.0f}")
    print("sites", mc.num_sites(3), "cycles/site/wave", tot / (256 * 8) / nsite_wg)
