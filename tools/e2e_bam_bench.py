#!/usr/bin/env python3
"""End-to-end rate of `hifimeth-hip call` (BAM decode -> GPU -> BAM encode) on a synthetic HiFi BAM.
Reported separately from bench.py's HBM-resident figure."""
import os, subprocess, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bamutil
from hifimeth_amd.synth import synth_reads

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
threads = sys.argv[2] if len(sys.argv) > 2 else "16"
contexts = sys.argv[3] if len(sys.argv) > 3 else "cpg,chg,chh"
tmp = os.environ.get("TMPDIR", "/tmp")
src, dst = os.path.join(tmp, "e2e_in.bam"), os.path.join(tmp, "e2e_out.bam")
t = time.time(); reads = synth_reads(n, seed=20250220); bamutil.reads_to_bam(src, reads, level=1)
print(f"synthetic BAM: {n} reads, {sum(r.l_qseq for r in reads)/1e6:.1f} Mbases, {os.path.getsize(src)/1e6:.1f} MB, built in {time.time()-t:.1f} s", flush=True)
cli = os.path.join(ROOT, "hifimeth_amd", "bin", "hifimeth-hip")
for b in ("250", "1000"):
    t = time.time()
    p = subprocess.run([cli, "call", "-c", contexts, "-b", b, "-t", threads, src, dst], stderr=subprocess.PIPE, text=True)
    dt = time.time() - t
    tail = [l for l in p.stderr.splitlines() if "##" in l]
    print(f"-b {b}: exit {p.returncode}, {dt:.2f} s wall;", " | ".join(x.strip() for x in tail), flush=True)
print("output size MB:", os.path.getsize(dst) / 1e6)
