"""Where the CLI's fixed costs go: HM_CLI_TIMING=1 hifimeth-hip call on a synthetic BAM (python tools/cli_timing.py [reads])."""
import os, subprocess, sys, tempfile, time
sys.path.insert(0, ".")
from hifimeth_amd.synth import synth_slab, write_unaligned_bam
n = int(sys.argv[1]) if len(sys.argv) > 1 else 24000
reads = synth_slab(min(n, 12000), seed=3)
reads = [reads[i % len(reads)] for i in range(n)]
d = tempfile.mkdtemp()
src, dst = os.path.join(d, "in.bam"), os.path.join(d, "out.bam")
write_unaligned_bam(src, reads, level=1, threads=16)
for rep in range(2):
    t0 = time.perf_counter()
    p = subprocess.run(["hifimeth_amd/bin/hifimeth-hip", "call", src, dst], env=dict(os.environ, HM_CLI_TIMING="1"), stderr=subprocess.PIPE, text=True)
    print(f"run {rep}: wall {time.perf_counter() - t0:.3f} s, exit {p.returncode}")
    print("\n".join(l for l in p.stderr.splitlines() if " t=" in l or "Wall time" in l))
