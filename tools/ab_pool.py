"""Worker pool vs a thread start per batch in the BGZF code: the same `hifimeth-hip call` with both binaries, alternating
(python tools/ab_pool.py [reads]; the second binary, built from the commit before the pool, is hifimeth_amd/bin/hifimeth-hip-nopool)."""
import os, subprocess, sys, tempfile, time
sys.path.insert(0, ".")
from hifimeth_amd.synth import synth_slab, write_unaligned_bam
n = int(sys.argv[1]) if len(sys.argv) > 1 else 96000
reads = synth_slab(min(n, 12000), seed=3)
reads = [reads[i % len(reads)] for i in range(n)]
d = tempfile.mkdtemp()
src, dst = os.path.join(d, "in.bam"), os.path.join(d, "out.bam")
write_unaligned_bam(src, reads, level=1, threads=16)
print(f"input {os.path.getsize(src) / 1e6:.0f} MB, {n} reads")
for rep in range(3):
    for exe in ("hifimeth-hip-nopool", "hifimeth-hip"):
        for extra in ([], ["-Z"]):
            t0 = time.perf_counter()
            p = subprocess.run(["hifimeth_amd/bin/" + exe, "call"] + extra + [src, dst], stderr=subprocess.PIPE, text=True)
            print(f"rep {rep} {exe:20s} {' '.join(extra):3s} wall {time.perf_counter() - t0:.3f} s exit {p.returncode}", flush=True)
