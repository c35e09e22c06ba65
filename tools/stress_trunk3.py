import sys, numpy as np
sys.path.insert(0, ".")
from hifimeth_amd import MethylationCaller
from hifimeth_amd.synth import synth_reads
rng = np.random.default_rng(7)
bad = 0
for trial in range(14):
    n = int(rng.integers(1, 40))
    med = int(rng.choice([1100, 1500, 2500, 6000, 15000]))
    reads = synth_reads(n, seed=300 + trial, median_len=med, sigma=float(rng.choice([0.05, 0.3, 0.8])), frac_wide=0.2, frac_short=0, frac_missing=0,
                        gc=float(rng.choice([0.2, 0.36, 0.6])))
    spec = str(rng.choice(["cpg,chg,chh", "chh", "cpg", "chg,chh"]))
    ncu = int(rng.choice([1, 2, 13, 64, 256, 999]))
    gb = int(rng.choice([0, 4096, 65536]))
    out = []
    for impl in (1, 3):
        with MethylationCaller(contexts=spec, device=0) as m:
            m.set_option("trunk", 1); m.set_option("trunk_impl", impl); m.set_option("num_cu", ncu)
            if gb: m.set_option("group_bases", gb)
            a = m.call(reads).copy(); b = m.call(reads).copy()
            assert a.tobytes() == b.tobytes()
            out.append(a)
    same = out[0].tobytes() == out[1].tobytes()
    bad += not same
    print(trial, n, med, spec, ncu, gb, len(out[0]), "same" if same else "DIFF", flush=True)
print("bad", bad)
sys.exit(1 if bad else 0)
