"""GPU parity of the `pileup` row: histograms, projected calls, per-locus counters and BED
text of the HIP path (through the hm_pileup_* C ABI) against the CPU oracle (oracle/pileup_oracle.py), bit-exact."""
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu
CLI = os.path.join(ROOT, "hifimeth_amd", "bin", "hifimeth-hip")


@pytest.fixture(scope="module")
def P():
    from oracle import pileup_oracle
    return pileup_oracle


def _as_dict(r):
    return dict(flag=r.flag, tid=r.tid, pos=r.pos, mapq=r.mapq, cigar=r.cigar, seq=r.seq, mm=r.mm, ml=r.ml)


def _data(n=60, eqx=True, seed=7, err=0.01, median_len=1500, length=12000):
    from hifimeth_amd.synth import synth_alignments, synth_genome
    genome = synth_genome(n_chr=3, length=length, seed=seed)
    return genome, synth_alignments(genome, n, seed=seed + 1, median_len=median_len, err=err, eqx=eqx)


def _run(genome, reads, batch=16, **kw):
    from hifimeth_amd.pileup import MethylationPileup
    pu = MethylationPileup(genome, **kw)
    for i, r in enumerate(reads):
        pu.add(r)
        if (i + 1) % batch == 0:
            pu.flush()                    # several batches: records accumulate in HBM across runs
    pu.flush()
    return pu


def _check(P, genome, reads, pu, want, thresholds=None):
    off = pu.offsets
    bins = pu.histograms()
    assert (bins == want["bins"]).all()
    g, p, m, o = pu.records()
    order = {id(r): i for i, r in enumerate(reads)}
    exp = sorted((int(off[sid] + soff), prob, motif) for sid, soff, prob, motif in want["records"])
    assert sorted(zip(g.tolist(), p.tolist(), m.tolist())) == exp
    thr = pu.resolve_thresholds(bins) if thresholds is None else thresholds
    assert thr == want["thresholds"]
    pu.count(thr)
    assert pu.num_records() == 0
    loci = pu.loci()
    assert [(int(l["gpos"]), int(l["pcov"]), int(l["ncov"]), int(l["motif"])) for l in loci] == \
        [(int(off[sid] + soff), pc, nc, mo) for sid, soff, pc, nc, mo in want["loci"]]
    assert pu.bed(loci) == want["bed"]
    return loci


@pytest.mark.parametrize("eqx", [True, False])
def test_pileup_matches_oracle(P, eqx):
    genome, reads = _data(eqx=eqx)
    want = P.pileup([_as_dict(r) for r in reads], genome)
    assert len(want["records"]) > 10000 and (want["bins"].sum(1) > 0).all()
    pu = _run(genome, reads)
    loci = _check(P, genome, reads, pu, want)
    # per-sequence fetch (what the CLI does) concatenates to the whole
    parts = [pu.loci(int(pu.offsets[s]), int(pu.offsets[s + 1])) for s in range(len(genome))]
    assert (np.concatenate(parts) == loci).all()
    pu.close()


def test_pileup_filters(P):
    genome, reads = _data(n=40, err=0.03)
    recs = [_as_dict(r) for r in reads]
    for kw in (dict(min_mapq=30), dict(min_pi=97.5), dict(min_mapq=10, min_pi=96.9)):
        want = P.pileup(recs, genome, **kw)
        base = P.pileup(recs, genome)
        assert 0 < len(want["records"]) < len(base["records"])
        assert (want["bins"] == base["bins"]).all()          # the filters come after the histograms (pileup.cpp:274)
        pu = _run(genome, reads, **kw)
        _check(P, genome, reads, pu, want)
        pu.close()


def test_pileup_explicit_thresholds_and_low_error(P):
    genome, reads = _data(n=30, err=0.0, seed=19)
    want = P.pileup([_as_dict(r) for r in reads], genome, thresholds=[200, 60, 128])
    pu = _run(genome, reads, batch=1000)
    _check(P, genome, reads, pu, want, thresholds=[200, 60, 128])
    pu.close()


def test_pileup_motif_conflict_follows_bam_order(P):
    """reference 'CGG': forward reads vote CpG at the C, reverse reads vote CpG at the C and CHG (their own CCG) at
    the same locus; the locus' file is decided by its last record in BAM order (the reference's own result depends on
    thread timing there)."""
    from hifimeth_amd.synth import AlignedRead, revcomp
    chrom = ("AT" * 20) + "ACGGTTACGGA" + ("TA" * 20)
    genome = [("c", chrom)]
    L = len(chrom)

    def read(flag, name):
        seq = chrom
        fwd = revcomp(seq) if flag & 16 else seq
        cs = [i for i, ch in enumerate(fwd) if ch == "C"]
        mm = "C+m" + "".join(",0" for _ in cs) + ";"
        return AlignedRead(name, flag, 0, 0, 60, [("M", L)], seq, mm, np.full(len(cs), 250, np.uint8))
    for flags in ((0, 16), (16, 0), (16, 16, 0, 0)):
        reads = [read(f, f"r{i}") for i, f in enumerate(flags)]
        want = P.pileup([_as_dict(r) for r in reads], genome)
        conflict = [l for l in want["loci"] if chrom[l[1]:l[1] + 3] == "CGG"]
        assert conflict and all(l[4] == (1 if flags[-1] == 16 else 0) for l in conflict)
        pu = _run(genome, reads)
        _check(P, genome, reads, pu, want)
        pu.close()


def test_pileup_submit_errors():
    from hifimeth_amd import HifimethError
    from hifimeth_amd.pileup import MethylationPileup
    from hifimeth_amd.synth import AlignedRead
    genome = [("c", "ACGT" * 50)]
    pu = MethylationPileup(genome)
    ok = AlignedRead("a", 0, 0, 0, 60, [("M", 40)], "ACGT" * 10, "C+m,0;", np.array([200], np.uint8))
    assert pu.add(ok) == 1
    assert pu.add(AlignedRead("u", 4, -1, -1, 0, [], "ACGT" * 10, "C+m,0;", np.array([200], np.uint8))) == 0
    assert pu.add(AlignedRead("n", 0, 0, 0, 60, [("M", 40)], "ACGT" * 10, None, None)) == 0
    with pytest.raises(HifimethError, match="past the end of the reference"):
        pu.add(AlignedRead("b", 0, 0, 180, 60, [("M", 40)], "ACGT" * 10, "C+m,0;", np.array([200], np.uint8)))
    with pytest.raises(HifimethError, match="more bases than SEQ"):
        pu.add(AlignedRead("c", 0, 0, 0, 60, [("M", 44)], "ACGT" * 10, "C+m,0;", np.array([200], np.uint8)))
    with pytest.raises(HifimethError, match="sequence index"):
        pu.add(AlignedRead("d", 0, 3, 0, 60, [("M", 40)], "ACGT" * 10, "C+m,0;", np.array([200], np.uint8)))
    pu.flush()
    assert pu.num_records() == 1
    pu.close()


def test_pileup_external_planes_and_rank_slices(P):
    """the multi-GPU layout on one GPU: count into caller-owned (torch) planes padded to world * chunk, then read each
    rank's slice back through hm_pileup_fetch_loci with plane_base -- what every rank does after the reduce-scatter."""
    import torch
    from hifimeth_amd.pileup import locus_ranges, reduce_scatter_planes
    genome, reads = _data(n=30)
    want = P.pileup([_as_dict(r) for r in reads], genome)
    n_loci = sum(len(s) for _, s in genome)
    world = 4
    ranges = locus_ranges(n_loci, world)
    chunk = ranges[0][1] - ranges[0][0]
    planes = [torch.zeros(world * chunk, dtype=torch.int32, device="cuda") for _ in range(3)]
    pu = _run(genome, reads, planes=planes)
    pu.count(pu.resolve_thresholds(pu.histograms()))
    torch.cuda.synchronize()
    rows = []
    for r, (lo, hi) in enumerate(ranges):
        sl = [t[r * chunk:(r + 1) * chunk] for t in planes]
        got = pu.loci(0, hi - lo, planes=sl, plane_base=lo)
        rows += [(int(l["gpos"]), int(l["pcov"]), int(l["ncov"]), int(l["motif"])) for l in got]
    off = pu.offsets
    assert rows == [(int(off[sid] + soff), pc, nc, mo) for sid, soff, pc, nc, mo in want["loci"]]

    class _One:                      # world of one: the exchange is the identity
        @staticmethod
        def is_initialized():
            return False
    pc, nc, key, base = reduce_scatter_planes(_One, *planes)
    assert base == 0 and pc.data_ptr() == planes[0].data_ptr()
    pu.close()


def test_cli_pileup_end_to_end(P, tmp_path):
    from bamutil import aligned_to_bam, write_fasta
    genome, reads = _data(n=80, seed=31)
    bam, fa, prefix = str(tmp_path / "mod.bam"), str(tmp_path / "ref.fa"), str(tmp_path / "out")
    aligned_to_bam(bam, genome, reads)
    write_fasta(fa, genome)
    r = subprocess.run([CLI, "pileup", "-t", "4", "-b", "25", fa, bam, prefix], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    want = P.pileup([_as_dict(x) for x in reads], genome)
    for c in ("CpG", "CHG", "CHH"):
        assert open(f"{prefix}.{c}.cov.bed").read() == want["bed"][c]
        assert f"{c} samples: " in r.stderr
    # filters on the command line
    r = subprocess.run([CLI, "pileup", "-q", "20", "-f", "98.5", fa, bam, prefix + "2"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    want = P.pileup([_as_dict(x) for x in reads], genome, min_mapq=20, min_pi=98.5)
    for c in ("CpG", "CHG", "CHH"):
        assert open(f"{prefix}2.{c}.cov.bed").read() == want["bed"][c]
    # unsorted / unmapped input is refused like the reference does (pileup.cpp:438-459)
    aligned_to_bam(bam, genome, reads, sort_order="unknown")
    r = subprocess.run([CLI, "pileup", fa, bam, prefix + "3"], capture_output=True, text=True)
    assert r.returncode == 1 and "BAM is not sorted" in r.stderr


@pytest.mark.parametrize("force_collectives", [False, True])
def test_pileup_dist_driver(P, tmp_path, force_collectives):
    """python -m hifimeth_amd.pileup_dist on one GPU: plain, and as a world of one over RCCL (all-reduce of the
    histograms, reduce-scatter of the planes) -- the code path every rank of an N-GPU job runs."""
    import sys
    from bamutil import aligned_to_bam, write_fasta
    genome, reads = _data(n=50, seed=53)
    bam, fa, prefix = str(tmp_path / "mod.bam"), str(tmp_path / "ref.fa"), str(tmp_path / "out")
    aligned_to_bam(bam, genome, reads)
    write_fasta(fa, genome)
    env = dict(os.environ, PYTHONPATH=ROOT)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    if force_collectives:
        env.update(HM_FORCE_COLLECTIVES="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29541")
    r = subprocess.run([sys.executable, "-m", "hifimeth_amd.pileup_dist", "--slab", "7", fa, bam, prefix],
                       capture_output=True, text=True, env=env, cwd=ROOT, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    want = P.pileup([_as_dict(x) for x in reads], genome)
    for c in ("CpG", "CHG", "CHH"):
        assert open(f"{prefix}.{c}.cov.bed").read() == want["bed"][c]


def test_pileup_high_error_alignments(P):
    """many short match runs (10 % substitutions / insertions / deletions): motifs straddling op boundaries, runs of
    length 1-2, soft clips on both ends"""
    for eqx in (True, False):
        genome, reads = _data(n=40, eqx=eqx, err=0.10, seed=71, median_len=900)
        want = P.pileup([_as_dict(r) for r in reads], genome)
        assert len(want["records"]) > 1000
        pu = _run(genome, reads, batch=9)
        _check(P, genome, reads, pu, want)
        pu.close()


def test_pileup_shard_invariance_and_linearity():
    """size-independent properties at a larger size (no oracle): two engines over disjoint halves of the records,
    planes summed / max-ed as the reduce-scatter does, equal one engine over everything; counters add up to the
    number of projected calls; thresholds only move calls between pcov and ncov."""
    import torch
    from hifimeth_amd.pileup import MethylationPileup
    genome, reads = _data(n=400, seed=91, median_len=2500, length=60000)
    n_loci = sum(len(s) for _, s in genome)

    def run(sel, thr, planes=None):
        pu = MethylationPileup(genome, planes=planes)
        for i in sel:
            pu.add(reads[i], order=i)
        pu.flush()
        n = pu.num_records()
        bins = pu.histograms()
        pu.count(thr)
        return pu, n, bins

    whole, n_all, bins_all = run(range(len(reads)), [128, 128, 128])
    loci = whole.loci()
    assert n_all > 100000 and int(loci["pcov"].sum() + loci["ncov"].sum()) == n_all
    planes = [[torch.zeros(n_loci, dtype=torch.int32, device="cuda") for _ in range(3)] for _ in range(2)]
    a, n_a, bins_a = run(range(0, len(reads), 2), [128, 128, 128], planes[0])
    b, n_b, bins_b = run(range(1, len(reads), 2), [128, 128, 128], planes[1])
    torch.cuda.synchronize()
    assert n_a + n_b == n_all and (bins_a + bins_b == bins_all).all()
    merged = [planes[0][0] + planes[1][0], planes[0][1] + planes[1][1], torch.maximum(planes[0][2], planes[1][2])]
    got = a.loci(0, n_loci, planes=merged, plane_base=0)
    assert (got == loci).all()
    lo, _, _ = run(range(len(reads)), [1, 1, 1])
    hi, _, _ = run(range(len(reads)), [255, 255, 255])
    l1, l2 = lo.loci(), hi.loci()
    assert (l1["gpos"] == l2["gpos"]).all() and (l1["pcov"] + l1["ncov"] == l2["pcov"] + l2["ncov"]).all()
    assert (l1["pcov"] >= l2["pcov"]).all() and l1["pcov"].sum() > l2["pcov"].sum()
    for x in (whole, a, b, lo, hi):
        x.close()


def test_cli_pileup_fasta_order_differs_from_bam_header(P, tmp_path):
    """records carry BAM tids; the reference resolves them by NAME in the FASTA (HbnDatabase::seq_name2id) and writes the
    chromosomes in FASTA order (pileup.cpp:514-595) -- here the FASTA lists them in the opposite order, gzipped"""
    import gzip
    from bamutil import aligned_to_bam
    genome, reads = _data(n=60, seed=61)
    bam, fa, prefix = str(tmp_path / "mod.bam"), str(tmp_path / "ref.fa.gz"), str(tmp_path / "out")
    aligned_to_bam(bam, genome, reads)
    with gzip.open(fa, "wt") as f:
        for n, s in reversed(genome):
            f.write(f">{n}\n{s}\n")
    r = subprocess.run([CLI, "pileup", fa, bam, prefix], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    want = P.pileup([_as_dict(x) for x in reads], genome)
    for c in ("CpG", "CHG", "CHH"):
        rows = want["bed"][c].splitlines(keepends=True)
        by_chr = {n: [l for l in rows if l.split("\t")[0] == n] for n, _ in genome}
        assert open(f"{prefix}.{c}.cov.bed").read() == "".join("".join(by_chr[n]) for n, _ in reversed(genome))
    # a reference without one of the chromosomes is an error, as in the reference
    with gzip.open(fa, "wt") as f:
        f.write(f">{genome[0][0]}\n{genome[0][1]}\n")
    r = subprocess.run([CLI, "pileup", fa, bam, prefix + "x"], capture_output=True, text=True)
    assert r.returncode != 0 and "does not exist" in r.stderr


def test_cli_eval_counts_and_sample_files(P, tmp_path):
    """hifimeth-hip eval (src/app/hifimeth/eval.cpp): thresholds and the counts[context][label][scaled_prob] table
    behind the sample files equal the oracle's; the files themselves (random draws in the reference too) are checked
    through their invariants: 5 files x (100 000 positives + 100 000 negatives), predictions follow the threshold,
    every drawn value exists in the table, and the seed makes the draw repeatable."""
    import json
    from bamutil import aligned_to_bam, write_fasta
    from hifimeth_amd.synth import AlignedRead
    genome, reads = _data(n=90, seed=61)
    # an unmapped primary record with calls: counts for the thresholds, gives no samples
    donor = next(r for r in reads if r.mm and r.flag == 0)
    reads.append(AlignedRead("unmapped1", 4, -1, -1, 0, [], donor.seq, donor.mm, donor.ml))
    recs = [_as_dict(x) for x in reads]
    rng = np.random.default_rng(5)
    rows = []
    for name, sq in genome:
        for i in range(len(sq)):
            if sq[i] in "CG" and rng.random() < 0.7:
                kind = rng.integers(0, 4)
                pc, nc = [(int(rng.integers(10, 40)), 0), (0, int(rng.integers(10, 40))), (int(rng.integers(1, 20)), int(rng.integers(1, 20))),
                          (int(rng.integers(0, 5)), 0)][kind]
                rows.append(f"{name}\t{i}\t{i + 1}\t{100.0 * pc / max(1, pc + nc):g}\t{pc}\t{nc}")
    bed_text = "\n".join(rows) + "\n"
    bam, fa, bed, prefix, dump = (str(tmp_path / x) for x in ("mod.bam", "ref.fa", "truth.bed", "ev", "counts.json"))
    aligned_to_bam(bam, genome, reads)
    write_fasta(fa, genome)
    open(bed, "w").write(bed_text)
    r = subprocess.run([CLI, "eval", "-s", "11", "-d", dump, fa, bed, bam, prefix], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    labels = P.bismark_labels(genome, bed_text)
    bins, thr, cnt = P.eval_counts(recs, genome, labels)
    got = json.load(open(dump))
    assert got["thresholds"] == thr
    assert (np.array(got["counts"], np.uint64).reshape(3, 2, 256) == cnt).all()
    assert cnt[:, 0].sum() > 0 and cnt[:, 1].sum() > 0
    n_pos, n_neg = sum(v == 1 for v in labels.values()), sum(v == 0 for v in labels.values())
    assert f"Load {n_pos} methylated sites and {n_neg} unmethylated sites" in r.stderr
    for c, name in enumerate(("CpG", "CHG", "CHH")):
        if cnt[c, 0].sum() == 0 or cnt[c, 1].sum() == 0:
            assert not os.path.exists(f"{prefix}.{name}.0")
            continue
        assert f"Original {name} positive samples: {int(cnt[c, 1].sum())}" in r.stderr      # small set: replicated
        for i in range(5):
            rows_ = [l.split("\t") for l in open(f"{prefix}.{name}.{i}").read().strip().split("\n")]
            assert len(rows_) == 200000
            lab = np.array([int(x[0]) for x in rows_])
            pred = np.array([int(x[1]) for x in rows_])
            prob = np.rint(np.array([float(x[2]) for x in rows_]) * 255).astype(int)
            assert (lab[:100000] == 1).all() and (lab[100000:] == 0).all()
            assert (pred == (prob >= thr[c])).all()
            assert set(prob[:100000]) <= set(np.nonzero(cnt[c, 1])[0]) and set(prob[100000:]) <= set(np.nonzero(cnt[c, 0])[0])
    first = open(f"{prefix}.CpG.0").read()
    r2 = subprocess.run([CLI, "eval", "-s", "11", fa, bed, bam, prefix + "b"], capture_output=True, text=True)
    assert r2.returncode == 0 and open(f"{prefix}b.CpG.0").read() == first
    r3 = subprocess.run([CLI, "eval", "-s", "12", fa, bed, bam, prefix + "c"], capture_output=True, text=True)
    assert r3.returncode == 0 and open(f"{prefix}c.CpG.0").read() != first
    assert subprocess.run([CLI, "eval", fa, bed, bam], capture_output=True).returncode == 1


def test_label_histograms_binding(P):
    """hm_pileup_label_histograms through the Python mirror: records joined with labels == the oracle's table"""
    genome, reads = _data(n=40, seed=67)
    pu = _run(genome, reads)
    off = pu.offsets
    rng = np.random.default_rng(2)
    lab = rng.integers(-1, 2, int(off[-1])).astype(np.int8)
    labels = {}
    for sid in range(len(genome)):
        for soff in range(len(genome[sid][1])):
            v = int(lab[off[sid] + soff])
            if v >= 0:
                labels[(sid, soff)] = v
    _, _, cnt = P.eval_counts([_as_dict(x) for x in reads], genome, labels)
    assert (pu.label_histograms(lab) == cnt).all() and cnt.sum() > 100
    assert pu.num_records() > 0                     # the records stay resident
    with pytest.raises(Exception):
        pu.label_histograms(lab[:-1])
