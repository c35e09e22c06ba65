#!/usr/bin/env python3
"""Throughput of the `pileup` device path on one GPU, synthetic data.

A random genome, error-free reads (one '=' CIGAR op each, both strands) with call-like MM/ML tags; the MM/ML lists are
parsed once on the host and the staged batches are replayed, so the timed region is what the GPU does per batch:
H2D of the staged records, plane memset, mods_kernel, project_kernel, then count_kernel and the covered-loci
compaction at the end.  Prints one JSON object; `--check` verifies that the per-locus counters add up to the number of projected calls.

    python tools/pileup_bench.py --genome-mb 20 --coverage 10
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

from hifimeth_amd.pileup import MOD_DTYPE, MethylationPileup  # noqa: E402
from hifimeth_amd.synth import pack_codes  # noqa: E402

_ASCII = np.frombuffer(b"ACGT", np.uint8)
_COMP = np.zeros(256, np.uint8)
_COMP[[65, 67, 71, 84]] = [84, 71, 67, 65]


def call_like_mods(fwd: np.ndarray, rng) -> np.ndarray:
    """C+m on CpG/CHG/CHH cytosines, G-m on the G of [AGT][AGT]G (what `hifimeth call` writes), already parsed"""
    L = len(fwd)
    C_, G_ = 67, 71
    n1 = np.concatenate([fwd[1:], [0]])
    n2 = np.concatenate([fwd[2:], [0, 0]])
    p1 = np.concatenate([[0], fwd[:-1]])
    p2 = np.concatenate([[0, 0], fwd[:-2]])
    isH = lambda x: (x == 65) | (x == 67) | (x == 84)  # noqa: E731
    isD = lambda x: (x == 65) | (x == 71) | (x == 84)  # noqa: E731
    fc = np.nonzero((fwd == C_) & ((n1 == G_) | (isH(n1) & ((n2 == G_) | isH(n2)))))[0]
    rg = np.nonzero((fwd == G_) & isD(p1) & isD(p2) & (np.arange(L) >= 2))[0]
    m = np.zeros(len(fc) + len(rg), MOD_DTYPE)
    m["qoff"] = np.concatenate([fc, rg])
    m["strand"][len(fc):] = 1
    m["unmod_base"][:len(fc)] = b"C"
    m["unmod_base"][len(fc):] = b"G"
    m["code"] = b"m"
    m["prob"] = np.where(rng.random(len(m)) < 0.5, rng.integers(0, 60, len(m)), rng.integers(196, 256, len(m)))
    return m


def cpu_baseline(genome, chrom, starts, staged, read_len, n_sample=600, repeat=12):
    """The reference's BamQuerySequence::init + BamMapInfo::init + extract_{cpg,chg,chh}_mapped_samples
    (src/corelib/bam_info.cpp:169-439, 5mc_motif_finder.cpp), built from its sources by oracle/ref_build, one thread."""
    import subprocess
    import tempfile
    exe = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "oracle", "_ref", "ref_align")
    if not os.path.exists(exe):
        return None
    n = min(n_sample, len(staged))
    with tempfile.TemporaryDirectory() as d:
        fa = os.path.join(d, "g.fa")
        with open(fa, "w") as f:
            f.write(">chr1\n" + genome[0][1] + "\n")
        recs = "".join(f"{staged[i][0]} 0 {int(starts[i])} {read_len}= {chrom[int(starts[i]):int(starts[i]) + read_len].tobytes().decode()}\n"
                       for i in range(n))
        r = subprocess.run([exe, "-t", str(repeat), fa], input=recs.encode(), capture_output=True, check=True)
    t = r.stdout.decode().split()
    cols, secs = int(t[4]), float(t[-1])
    return dict(value=round(cols / secs), unit="aligned columns/s", cores=1, kind="reference",
                sample=f"{n} reads x {repeat} passes = {cols} columns through the reference's alignment projection "
                       f"(ref_align -t), {secs:.1f} s")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--genome-mb", type=float, default=20)
    ap.add_argument("--coverage", type=float, default=10)
    ap.add_argument("--read-len", type=int, default=15000)
    ap.add_argument("--batch", type=int, default=512, help="reads per hm_pileup_run")
    ap.add_argument("--repeat", type=int, default=3, help="timed passes over the staged read set")
    ap.add_argument("--check", action="store_true")
    ap.add_argument("--cpu-baseline", action="store_true",
                    help="time the reference's own projection code (oracle/_ref/ref_align -t) on a bounded sample")
    a = ap.parse_args()

    rng = np.random.default_rng(1)
    G = int(a.genome_mb * 1e6)
    gc = 0.36
    codes = rng.choice(4, G, p=[(1 - gc) / 2, gc / 2, gc / 2, (1 - gc) / 2]).astype(np.uint8)
    chrom = _ASCII[codes]
    genome = [("chr1", chrom.tobytes().decode())]
    n_reads = int(G * a.coverage / a.read_len)
    starts = np.sort(rng.integers(0, G - a.read_len, n_reads))
    t0 = time.perf_counter()
    staged = []
    lut = np.zeros(256, np.uint8)
    lut[[65, 67, 71, 84]] = [0, 1, 2, 3]
    for i, s in enumerate(starts):
        seq = chrom[s:s + a.read_len]
        rev = bool(rng.random() < 0.5)
        fwd = _COMP[seq][::-1] if rev else seq
        mods = call_like_mods(fwd, rng)
        staged.append((16 if rev else 0, int(s), pack_codes(lut[seq]), np.array([(a.read_len << 4) | 7], np.uint32), mods))
    t_prep = time.perf_counter() - t0
    n_mods = sum(len(x[4]) for x in staged)

    pu = MethylationPileup(genome)
    L = pu._L

    def one_pass():
        for i, (flag, pos, seq4, cig, mods) in enumerate(staged):
            rc = L.hm_pileup_submit_read(pu._h, i, flag, 0, pos, 60, a.read_len, seq4.ctypes.data_as(C.c_void_p), 1,
                                         cig.ctypes.data_as(C.c_void_p), len(mods), mods.ctypes.data_as(C.c_void_p))
            assert rc == 1
            if (i + 1) % a.batch == 0:
                pu.flush()
        pu.flush()

    one_pass()                                     # warm-up (allocations)
    recs_per_pass = pu.num_records()
    pu.count([128, 128, 128])
    t0 = time.perf_counter()
    for _ in range(a.repeat):
        one_pass()
    t_project = time.perf_counter() - t0
    t0 = time.perf_counter()
    pu.count([128, 128, 128])
    t_count = time.perf_counter() - t0
    t0 = time.perf_counter()
    loci = pu.loci()
    t_loci = time.perf_counter() - t0
    cols = n_reads * a.read_len
    out = dict(genome_bases=G, reads=n_reads, aligned_columns=cols, mods=n_mods, records_per_pass=recs_per_pass,
               host_prep_s=round(t_prep, 2),
               stage_and_project_s_per_pass=round(t_project / a.repeat, 4),
               aligned_columns_per_s=round(cols * a.repeat / t_project),
               count_s=round(t_count, 4), records_counted=recs_per_pass * a.repeat,
               records_per_s_count=round(recs_per_pass * a.repeat / t_count),
               covered_loci=int(len(loci)), loci_fetch_s=round(t_loci, 4),
               loci_scan_bases_per_s=round(G / t_loci))
    if a.check:                                    # every pass (and the warm-up) adds the same records
        total = int((loci["pcov"].astype(np.int64) + loci["ncov"]).sum())
        out["check_total_records"] = total == recs_per_pass * (a.repeat + 1)
    if a.cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(genome, chrom, starts, staged, a.read_len)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
