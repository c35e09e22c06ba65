#!/usr/bin/env python3
"""End-to-end rate of `hifimeth-hip pileup` (BGZF/BAM decode, MM/ML parsing, GPU projection + counting, BED text) on a
synthetic aligned mod-BAM.  usage: e2e_pileup_bench.py [genome_mb] [coverage] [threads]"""
import os, subprocess, sys, time
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bamutil
from hifimeth_amd.synth import AlignedRead, revcomp

gmb = float(sys.argv[1]) if len(sys.argv) > 1 else 5
cov = float(sys.argv[2]) if len(sys.argv) > 2 else 10
threads = sys.argv[3] if len(sys.argv) > 3 else "16"
tmp = os.environ.get("TMPDIR", "/tmp")
rng = np.random.default_rng(2)
G, L = int(gmb * 1e6), 15000
chrom = np.frombuffer(b"ACGT", np.uint8)[rng.choice(4, G, p=[0.32, 0.18, 0.18, 0.32])].tobytes().decode()
genome = [("chr1", chrom)]
t = time.time()
reads = []
for i, s in enumerate(np.sort(rng.integers(0, G - L, int(G * cov / L)))):
    seq = chrom[s:s + L]
    rev = bool(rng.random() < 0.5)
    fwd = np.frombuffer((revcomp(seq) if rev else seq).encode(), np.uint8)
    parts, mls = [], []
    for base, head in ((67, "C+m"), (71, "G-m")):          # every C / G called: an upper bound on the tag volume
        n = int((fwd == base).sum())
        parts.append(head + ",0" * n + ";")
        mls.append(np.where(rng.random(n) < 0.5, rng.integers(0, 60, n), rng.integers(196, 256, n)).astype(np.uint8))
    reads.append(AlignedRead(f"r{i}", 16 if rev else 0, 0, int(s), 60, [("=", L)], seq, "".join(parts), np.concatenate(mls)))
bam, fa, prefix = os.path.join(tmp, "pu_in.bam"), os.path.join(tmp, "pu_ref.fa"), os.path.join(tmp, "pu_out")
bamutil.aligned_to_bam(bam, genome, reads, level=1)
bamutil.write_fasta(fa, genome)
print(f"synthetic mod-BAM: {len(reads)} reads, {len(reads) * L / 1e6:.1f} Mbases aligned, {os.path.getsize(bam) / 1e6:.1f} MB, "
      f"built in {time.time() - t:.1f} s", flush=True)
cli = os.path.join(ROOT, "hifimeth_amd", "bin", "hifimeth-hip")
for b in ("512", "4096"):
    t = time.time()
    p = subprocess.run([cli, "pileup", "-t", threads, "-b", b, fa, bam, prefix], stderr=subprocess.PIPE, text=True)
    dt = time.time() - t
    rows = sum(1 for c in ("CpG", "CHG", "CHH") for _ in open(f"{prefix}.{c}.cov.bed"))
    print(f"-b {b}: exit {p.returncode}, {dt:.2f} s wall, {len(reads) * L / dt / 1e6:.1f} M aligned bases/s, {rows} BED rows", flush=True)
    for l in p.stderr.splitlines():
        if "##" in l:
            print("   ", l.strip())
