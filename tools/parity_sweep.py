#!/usr/bin/env python3
"""One-off larger parity sweep: GPU calls vs the CPU oracle over N synthetic reads (all contexts), several seeds.
usage: parity_sweep.py [reads_per_seed] [seeds] [engine option=value ...]   (e.g. trunk=0 precision=2)"""
import os, sys, time
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
from hifimeth_amd import MethylationCaller
from hifimeth_amd.synth import synth_reads
from oracle import hm_oracle as O

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
seeds = int(sys.argv[2]) if len(sys.argv) > 2 else 3
O.build()
models = [O.Model(os.path.join(ROOT, "hifimeth_amd", "weights", c + ".hmw")) for c in ("CpG", "CHG", "CHH")]
worst, total, nml, t0 = 0.0, 0, 0, time.time()
hist = np.zeros(8, np.int64)   # |dp| decades 1e-9..1e-2
opts = [a.split("=") for a in sys.argv[3:]]
print("engine options:", opts or "defaults (dense trunk, split-half fp16x3)")
with MethylationCaller(device=0) as mc:
    for k, v in opts:
        mc.set_option(k, int(v))
    for seed in range(seeds):
        reads = synth_reads(n, seed=1000 + seed, gc=(0.36, 0.41, 0.5)[seed % 3])
        calls = mc.call(reads)
        for rid, rd in enumerate(reads):
            if not rd.has_kinetics() or rd.l_qseq < 1000:
                continue
            want = O.call_read(models, 7, rd, nthreads=16)
            got = calls[calls["read_id"] == rid]
            order = np.lexsort((want["qoff"], want["strand"]))
            assert len(got) == len(order) and np.array_equal(got["qoff"], want["qoff"][order])
            d = np.abs(got["p"] - want["p"][order])
            worst = max(worst, float(d.max(initial=0)))
            nml += int((got["scaled_prob"] != want["ml"][order]).sum())
            total += len(got)
            hist += np.histogram(np.log10(np.maximum(d, 1e-12)), bins=[-13, -8, -7, -6, -5, -4, -3, -2, 0])[0]
        print(f"seed {seed}: {total} sites so far, max|dp| {worst:.2e}, ML bytes off by one {nml}, {time.time() - t0:.0f} s", flush=True)
print("decades (<1e-8, 1e-8..1e-7, ..1e-6, ..1e-5, ..1e-4, ..1e-3, ..1e-2, >1e-2):", hist.tolist())
