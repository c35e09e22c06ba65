#!/usr/bin/env python3
"""How evenly does a read-sharded `call` job spread its WORK (cytosine sites) over the ranks?  CPU only.

A synthetic HiFi BAM whose GC fraction and read length drift along the file (as they do along a real run: SMRT cells,
libraries and organelle reads are not shuffled) is cut (a) into `world` equal BYTE ranges at BGZF boundaries -- the static
split `hifimeth-hip call -R r/w` -- and (b) into the many small parts of the shared work queue (`-Q`, what
hifimeth_amd.call_dist launches), which ranks pull as they finish (simulated: a rank's time for a part = its sites).
Prints max / mean sites per rank for both; > 1.05 means the slowest rank holds the job back by more than 5 %.

    python tools/shard_balance.py [reads] [world]
"""
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
from hifimeth_amd.synth import synth_reads, unpack_codes, write_unaligned_bam  # noqa: E402

CLI = os.path.join(ROOT, "hifimeth_amd", "bin", "hifimeth-hip")


def sites_of(read) -> int:
    """CpG + CHG (forward strand) + CHH (both strands), as the scanners count them (eval_kmer_features.cpp:67-126)."""
    if not read.has_kinetics() or read.l_qseq < 1000:
        return 0
    c = unpack_codes(read.seq4, read.l_qseq).astype(np.int8)
    a, b, d = c[:-2], c[1:-1], c[2:]
    h1, h2 = (b != 2) & (b != 4), (d != 2) & (d != 4)
    cpg = int(((c[:-1] == 1) & (c[1:] == 2)).sum())
    chg = int(((a == 1) & h1 & (d == 2)).sum())
    fwd = (a == 1) & h1 & h2
    rev = ~fwd & (a != 1) & (a != 4) & (b != 1) & (b != 4) & (d == 2)
    return cpg + chg + int(fwd.sum()) + int(rev.sum())


def drifting_reads(n, seed=11):
    """n reads in 16 consecutive blocks: GC 0.30 -> 0.45 and median length 10 kb -> 20 kb from the head of the file to its tail."""
    out = []
    for k in range(16):
        f = k / 15.0
        out += synth_reads(n // 16, seed=seed + k, gc=0.30 + 0.15 * f, median_len=int(10000 + 10000 * f), sigma=0.3)
    return out


def records_per_part(src, parts):
    counts = []
    with tempfile.TemporaryDirectory() as tmp:
        for r in range(parts):
            p = subprocess.run([CLI, "bamcopy", "-R", f"{r}/{parts}", src, os.path.join(tmp, "o.bam")], capture_output=True, text=True, check=True)
            counts.append(int(p.stderr.split("wrote")[1].split()[0]))
    return counts


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1600
    world = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    reads = drifting_reads(n)
    sites = np.array([sites_of(r) for r in reads], np.int64)
    with tempfile.TemporaryDirectory() as tmp:
        src = os.path.join(tmp, "drift.bam")
        write_unaligned_bam(src, reads, level=1, threads=8)
        size = os.path.getsize(src)
        print(f"{len(reads)} reads, {sum(r.l_qseq for r in reads) / 1e6:.1f} Mbases, {sites.sum() / 1e6:.2f} M sites, {size / 1e6:.1f} MB BAM; "
              f"sites per base {sites[:n // 16].sum() / sum(r.l_qseq for r in reads[:n // 16]):.3f} at the head, "
              f"{sites[-(n // 16):].sum() / sum(r.l_qseq for r in reads[-(n // 16):]):.3f} at the tail")
        for label, parts in (("static byte ranges (-R r/w)", world), ("work queue (-Q), 4 parts per rank", 4 * world),
                             ("work queue (-Q), 16 parts per rank", 16 * world)):
            cnt = records_per_part(src, parts)
            assert sum(cnt) == len(reads)
            edges = np.concatenate([[0], np.cumsum(cnt)])
            work = np.array([sites[edges[i]:edges[i + 1]].sum() for i in range(parts)], np.float64)
            if parts == world:
                per_rank = work
            else:   # ranks pull the next part when they are free
                busy = np.zeros(world)
                for w in work:
                    busy[np.argmin(busy)] += w
                per_rank = busy
            print(f"{label:38s}: {parts:4d} parts, max / mean sites per rank = {per_rank.max() / per_rank.mean():.3f}")


if __name__ == "__main__":
    main()
