"""The BAM front end (`hifimeth-hip`): BGZF/BAM codec and the MM/ML/MN writer, on CPU (no GPU needed:
sub-commands bamcopy / tagtest), against the restated reference tag builder (oracle/modtags.py)."""
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT
from hifimeth_amd.caller import CALL_DTYPE
from hifimeth_amd.synth import synth_reads

import bamutil

CLI = os.path.join(ROOT, "hifimeth_amd", "bin", "hifimeth-hip")


def _reads():
    return synth_reads(7, seed=3, median_len=1800, sigma=0.3, frac_wide=0.3, frac_short=0.2, frac_missing=0.2)


def _oracle_calls(oracle, oracle_models, reads, mask=7):
    recs = []
    for i, rd in enumerate(reads):
        if not rd.has_kinetics() or rd.l_qseq < 1000:
            continue
        r = oracle.call_read(oracle_models, mask, rd)
        order = np.lexsort((r["qoff"], r["strand"]))
        c = np.zeros(len(order), CALL_DTYPE)
        c["read_id"], c["qoff"], c["strand"], c["ctx"] = i, r["qoff"][order], r["strand"][order], r["ctx"][order]
        c["scaled_prob"], c["p"] = r["ml"][order], r["p"][order]
        recs.append(c)
    return np.concatenate(recs)


def test_bamcopy_roundtrip(tmp_path):
    reads = _reads()
    src, dst = str(tmp_path / "in.bam"), str(tmp_path / "out.bam")
    bamutil.reads_to_bam(src, reads)
    subprocess.check_call([CLI, "bamcopy", src, dst])
    t0, r0 = bamutil.read_bam(src)
    t1, r1 = bamutil.read_bam(dst)
    assert t0 == t1 and len(r0) == len(r1) == len(reads)
    assert all(a["raw"] == b["raw"] for a, b in zip(r0, r1))
    # many blocks, multi-threaded inflate/deflate path: a larger file
    big = synth_reads(40, seed=8, frac_missing=0)
    bamutil.reads_to_bam(src, big)
    subprocess.check_call([CLI, "bamcopy", src, dst])
    assert [r["raw"] for r in bamutil.read_bam(src)[1]] == [r["raw"] for r in bamutil.read_bam(dst)[1]]
    assert open(dst, "rb").read()[-28:] == bamutil._EOF


def test_bad_input_is_rejected(tmp_path):
    bad = tmp_path / "bad.bam"
    bad.write_bytes(b"not a bam file at all")
    assert subprocess.call([CLI, "bamcopy", str(bad), str(tmp_path / "o.bam")], stderr=subprocess.DEVNULL) != 0
    assert subprocess.call([CLI, "call", "-c", "cpg,foo", "a.bam", "b.bam"], stderr=subprocess.DEVNULL) != 0
    assert subprocess.call([CLI, "call", "only-one-arg.bam"], stderr=subprocess.DEVNULL) != 0


@pytest.mark.parametrize("keep", [False, True])
def test_tag_writer_matches_reference_rules(tmp_path, oracle, oracle_models, keep):
    from oracle.modtags import expected_tags
    reads = _reads()
    calls = _oracle_calls(oracle, oracle_models, reads)
    src, dst, cb = str(tmp_path / "in.bam"), str(tmp_path / "out.bam"), str(tmp_path / "calls.bin")
    # records 0 and 1 carry stale MM/ML tags that must disappear
    stale = lambda i, r: (bamutil.aux_Z("MM", "C+m,1;") + bamutil.aux_B("ML", np.array([7], np.uint8))) if i < 2 else b""
    bamutil.reads_to_bam(src, reads, extra_aux=stale)
    calls.tofile(cb)
    subprocess.check_call([CLI, "tagtest", src, cb, dst] + (["-k"] if keep else []))
    text, recs = bamutil.read_bam(dst)
    assert "@PG\tID:hifimeth-hip" in text and text.startswith("@HD")
    assert len(recs) == len(reads)
    for i, (rd, rec) in enumerate(zip(reads, recs)):
        assert rec["name"] == rd.name and rec["l_seq"] == rd.l_qseq and np.array_equal(rec["seq4"], rd.seq4)
        tags = bamutil.parse_aux(rec["aux"])
        names = [t[0] for t in tags]
        assert names[:3] == ["np", "rq", "RG"] and "zm" in names          # untouched tags keep their order
        for k in ("fi", "fp", "ri", "rp"):
            present = getattr(rd, k) is not None
            assert (k in names) == (keep and present)
        mine = calls[calls["read_id"] == i]
        want = expected_tags(oracle.decode(rd), mine["qoff"], mine["strand"], mine["scaled_prob"]) if len(mine) else None
        d = {t[0]: t for t in tags}
        if want is None:
            assert "MM" not in d and "ML" not in d and "MN" not in d       # build_mod_bam.cpp:129-130 returns early
            continue
        assert names[-3:] == ["MM", "ML", "MN"]
        assert d["MM"][1] == "Z" and d["MM"][2] == want["MM"]
        assert d["ML"][1] == "BC" and np.array_equal(d["ML"][2], want["ML"])
        assert d["MN"][2] == want["MN"] and d["MN"][1] == ("S" if rd.l_qseq <= 0xffff else "I")
        assert names.count("MM") == 1 and names.count("ML") == 1


def test_tag_writer_reverse_flag_read(tmp_path, oracle, oracle_models):
    """flag 0x10: MM deltas count C/G on the FORWARD strand (get_bam_fwd_strand_base, bam_info.cpp:224-233)."""
    from oracle.modtags import expected_tags
    rd = synth_reads(1, seed=2, median_len=1300, sigma=0.05, frac_short=0, frac_missing=0, frac_wide=0)[0]
    rd.flag = 16
    calls = _oracle_calls(oracle, oracle_models, [rd], mask=1)
    src, dst, cb = str(tmp_path / "in.bam"), str(tmp_path / "out.bam"), str(tmp_path / "calls.bin")
    bamutil.reads_to_bam(src, [rd])
    calls.tofile(cb)
    subprocess.check_call([CLI, "tagtest", src, cb, dst])
    tags = {t[0]: t for t in bamutil.parse_aux(bamutil.read_bam(dst)[1][0]["aux"])}
    want = expected_tags(oracle.decode(rd), calls["qoff"], calls["strand"], calls["scaled_prob"])
    assert tags["MM"][2] == want["MM"] and np.array_equal(tags["ML"][2], want["ML"])


def test_tag_writer_literal_known_answers(tmp_path):
    """MM / ML / MN strings worked out by hand from build_mod_bam.cpp:134-176 (tests/golden/modtags_known_answers.json):
    the product's writer AND the restated rules (oracle/modtags.py) must both reproduce them, so the two cannot drift
    together.  (The reference's own tag code needs htslib and holds no fixtures: parity is otherwise unpinned here.)"""
    import json
    from oracle.modtags import expected_tags
    from hifimeth_amd.synth import read_from_ascii
    cases = json.load(open(os.path.join(ROOT, "tests", "golden", "modtags_known_answers.json")))["cases"]
    comp = bytes.maketrans(b"ACGT", b"TGCA")
    reads, calls = [], []
    for i, c in enumerate(cases):
        seq = c["stored_seq"].encode()
        L = len(seq)
        k = np.zeros(L, np.uint8)
        reads.append(read_from_ascii(seq, k, k, k, k, flag=c["flag"], name=c["name"]))
        rec = np.zeros(len(c["calls"]), CALL_DTYPE)
        for j, (q, st, ml) in enumerate(c["calls"]):
            rec[j] = (i, q, st, 0, ml, 0, ml / 255.0)
        calls.append(rec)
        fwd = seq.translate(comp)[::-1] if c["flag"] & 16 else seq
        want = expected_tags(fwd, rec["qoff"], rec["strand"], rec["scaled_prob"])
        assert want["MM"] == c["MM"] and want["ML"].tolist() == c["ML"] and want["MN"] == c["MN"], c["name"]
    src, dst, cb = str(tmp_path / "in.bam"), str(tmp_path / "out.bam"), str(tmp_path / "calls.bin")
    bamutil.reads_to_bam(src, reads)
    np.concatenate(calls).tofile(cb)
    subprocess.check_call([CLI, "tagtest", src, cb, dst])
    _, recs = bamutil.read_bam(dst)
    for c, rec in zip(cases, recs):
        d = {t[0]: t for t in bamutil.parse_aux(rec["aux"])}
        assert d["MM"][2] == c["MM"] and d["ML"][2].tolist() == c["ML"] and d["MN"][2] == c["MN"], c["name"]


def _literal_pins():
    import json
    return json.load(open(os.path.join(ROOT, "tests", "golden", "literal_pins.json")))


def test_tag_strip_rules_and_mn_literal_known_answers(tmp_path):
    """tests/golden/literal_pins.json, tag_strip: which tags `call` removes, keeps and adds, IN WHICH ORDER, and how MN is typed --
    written by hand from build_mod_bam.cpp:87-109,125-130,176-224 and htslib's documented bam_aux_update_* semantics (the
    reference calls htslib inline here and holds no fixtures): fi/ri/fp/rp go unless -k, old MM / ML always go, MM / ML are
    appended, MN = l_qseq is appended -- or, if the read already carries one, updated where it stands, as wide as it was if
    the value fits -- and a read without calls gets the stripping only."""
    from hifimeth_amd.synth import read_from_ascii
    rng = np.random.default_rng(3)
    cases = _literal_pins()["tag_strip"]

    def aux_of(spec, L):
        out = b""
        for t in spec:
            name, _, rest = t.partition(":")
            typ, _, val = rest.partition("=")
            if typ == "i":
                out += bamutil.aux_i(name, int(val or 11))
            elif typ == "f":
                out += bamutil.aux_f(name, 0.5)
            elif typ == "Z":
                out += bamutil.aux_Z(name, val or "x")
            elif typ == "C":
                out += name.encode() + b"C" + bytes([int(val)])
            elif typ == "BC":
                out += bamutil.aux_B(name, np.full(int(val), 7, np.uint8) if val else rng.integers(0, 255, L).astype(np.uint8))
            else:
                raise AssertionError(t)
        return out

    for keep in (False, True):
        sub = [c for c in cases if c["keep_kinetics"] == keep]
        reads, calls = [], []
        for i, c in enumerate(sub):
            L = c["l_qseq"]
            seq = bytes(rng.choice(np.frombuffer(b"ACGT", np.uint8), L).tobytes())
            rd = read_from_ascii(seq, None, None, None, None, name=c["name"])
            reads.append(rd)
            if c["calls"]:
                cpos = [k for k, b in enumerate(seq) if b == ord("C")][:3]
                rec = np.zeros(len(cpos), CALL_DTYPE)
                for j, q in enumerate(cpos):
                    rec[j] = (i, q, 0, 0, 100 + j, 0, 0.4)
                calls.append(rec)
        src, dst, cb = str(tmp_path / f"in{keep}.bam"), str(tmp_path / f"out{keep}.bam"), str(tmp_path / f"calls{keep}.bin")
        # records carry exactly the fixture's tags, in the fixture's order
        parts = [b"BAM\1" + (0).to_bytes(4, "little") + (0).to_bytes(4, "little")]
        for c, rd in zip(sub, reads):
            parts.append(bamutil.record(rd.name, rd.flag, rd.seq4, rd.l_qseq, aux_of(c["in"], rd.l_qseq)))
        bamutil.write_bgzf(src, b"".join(parts))
        np.concatenate(calls).tofile(cb)
        subprocess.check_call([CLI, "tagtest", src, cb, dst] + (["-k"] if keep else []))
        _, recs = bamutil.read_bam(dst)
        for c, rec in zip(sub, recs):
            tags = bamutil.parse_aux(rec["aux"])
            assert [f"{t[0]}:{t[1]}" for t in tags] == c["out"], (c["name"], [f"{t[0]}:{t[1]}" for t in tags])
            mn = [t[2] for t in tags if t[0] == "MN"]
            assert (mn[0] if mn else None) == c["MN"], c["name"]


def test_bed_rows_literal_known_answers():
    """tests/golden/literal_pins.json, bed_rows: the text of a *.cov.bed row, by hand from pileup.cpp:562-590 (an ostringstream's
    default floating-point format): the oracle's rows and the host mirror's must be these strings (the CLI's rows are compared
    with the oracle's byte for byte in tests/test_gpu_pileup.py)."""
    from hifimeth_amd.pileup import LOCUS_DTYPE, MethylationPileup
    from oracle import pileup_oracle as P
    rows = _literal_pins()["bed_rows"]
    for r in rows:
        assert "%s\t%d\t%d\t%g\t%d\t%d\n" % (r["chr"], r["k"], r["k"] + 1, 100.0 * r["pcov"] / (r["pcov"] + r["ncov"]), r["pcov"], r["ncov"]) == r["row"]
    # the oracle's BED writer on a genome / call set built to produce exactly these loci
    names = sorted({r["chr"] for r in rows}, key=[r["chr"] for r in rows].index)
    text = P.bed_text([(r["chr"], r["k"], r["pcov"], r["ncov"], 0) for r in rows])["CpG"]
    assert text == "".join(r["row"] for r in rows)
    # the host mirror formats loci the same way (no device needed: bed() is pure formatting)
    pu = MethylationPileup.__new__(MethylationPileup)
    lens = {n: max(r["k"] for r in rows if r["chr"] == n) + 2 for n in names}
    pu.names = names
    pu.offsets = np.concatenate([[0], np.cumsum([lens[n] for n in names])]).astype(np.int64)
    loci = np.zeros(len(rows), LOCUS_DTYPE)
    for i, r in enumerate(rows):
        loci[i]["gpos"] = pu.offsets[names.index(r["chr"])] + r["k"]
        loci[i]["pcov"], loci[i]["ncov"], loci[i]["motif"] = r["pcov"], r["ncov"], 0
    assert pu.bed(loci)["CpG"] == "".join(r["row"] for r in rows)


def test_tag_writer_rejects_unsorted_calls(tmp_path, oracle, oracle_models):
    reads = _reads()
    calls = _oracle_calls(oracle, oracle_models, reads)
    calls[[0, 1]] = calls[[1, 0]]                                          # per-strand order violated
    src, cb = str(tmp_path / "in.bam"), str(tmp_path / "calls.bin")
    bamutil.reads_to_bam(src, reads)
    calls.tofile(cb)
    assert subprocess.call([CLI, "tagtest", src, cb, str(tmp_path / "o.bam")], stderr=subprocess.DEVNULL) != 0


@pytest.mark.gpu
@pytest.mark.parametrize("ctx,mask", [("cpg,chg,chh", 7), ("cpg", 1)])
def test_cli_call_end_to_end(tmp_path, oracle, oracle_models, ctx, mask):
    """hifimeth-hip call on a synthetic BAM; MM strings exact, ML within 1 LSB of the oracle's."""
    from oracle.modtags import expected_tags
    reads = synth_reads(25, seed=44, median_len=2500, sigma=0.4, frac_wide=0.1, frac_short=0.1, frac_missing=0.1)
    src, dst = str(tmp_path / "in.bam"), str(tmp_path / "out.bam")
    bamutil.reads_to_bam(src, reads)
    # small -b: several batches in flight; "-d 0,0" drives the multi-device round-robin (4 engines) on the one GPU
    subprocess.check_call([CLI, "call", "-c", ctx, "-b", "7", "-t", "4", "-d", "0,0" if mask == 7 else "0", src, dst])
    _, recs = bamutil.read_bam(dst)
    assert [r["name"] for r in recs] == [r.name for r in reads]           # input order preserved
    called = 0
    for rd, rec in zip(reads, recs):
        tags = {t[0]: t for t in bamutil.parse_aux(rec["aux"])}
        assert not ({"fi", "fp", "ri", "rp"} & set(tags))
        if not rd.has_kinetics() or rd.l_qseq < 1000:
            assert "MM" not in tags
            continue
        r = oracle.call_read(oracle_models, mask, rd)
        want = expected_tags(oracle.decode(rd), r["qoff"], r["strand"], r["ml"])
        assert tags["MM"][2] == want["MM"] and tags["MN"][2] == rd.l_qseq
        assert np.abs(tags["ML"][2].astype(int) - want["ML"].astype(int)).max() <= 1
        called += 1
    assert called >= 15


@pytest.mark.gpu
def test_cli_call_sharded_over_ranks_matches_single_rank(tmp_path):
    """`call -R r/w`: two ranks (run one after the other on this box's one GPU) each call their byte range of the input;
    the merged BAM holds exactly the records of the single-rank run, in input order, and `-h` / `-v` exit 0
    (mod_options.cpp:62-71)."""
    reads = synth_reads(30, seed=61, median_len=4000, sigma=0.5, frac_wide=0.1, frac_short=0.1, frac_missing=0.1)
    src, one, two = str(tmp_path / "in.bam"), str(tmp_path / "one.bam"), str(tmp_path / "two.bam")
    bamutil.reads_to_bam(src, reads, level=1)
    subprocess.check_call([CLI, "call", "-b", "9", "-t", "4", src, one], stderr=subprocess.DEVNULL)
    for r in range(2):
        subprocess.check_call([CLI, "call", "-b", "9", "-t", "4", "-R", f"{r}/2", src, two], stderr=subprocess.DEVNULL)
    subprocess.check_call([CLI, "merge", two, "2"])
    _, a = bamutil.read_bam(one)
    _, b = bamutil.read_bam(two)
    assert len(a) == len(b) == len(reads)
    for x, y in zip(a, b):
        assert x["name"] == y["name"] and x["aux"] == y["aux"] and np.array_equal(x["seq4"], y["seq4"])
    assert sum(1 for x in a if b"MM" in x["aux"]) >= 15
    assert subprocess.call([CLI, "call", "-h"], stderr=subprocess.DEVNULL) == 0
    assert subprocess.call([CLI, "call", "-v"], stderr=subprocess.DEVNULL) == 0
    assert subprocess.call([CLI, "call", "--nonsense"], stderr=subprocess.DEVNULL) != 0


@pytest.mark.gpu
def test_cli_default_flags_cut_batches_into_slabs_without_changing_the_output(tmp_path):
    """`-b` keeps the reference's meaning (reads per outer batch, default 10000: mod_options.cpp:10-17) but no longer sets the
    granularity of the GPU pipeline: a batch is cut into engine slabs of <= 6 Mi bases (-S).  2 000 reads (~30 Mbases) with
    the default flags, with -b 250 and with tiny slabs must give byte-identical BAM payloads -- the calls do not depend
    on the batch cut (mod_main.cpp:330-362) -- and the default run must have used more than one slab."""
    import gzip
    reads = synth_reads(2000, seed=20250221)
    src = str(tmp_path / "in.bam")
    bamutil.reads_to_bam(src, reads, level=1)
    outs = []
    for tag, extra in (("default", []), ("b250", ["-b", "250"]), ("tiny", ["-S", "300000", "-b", "9999"])):
        dst = str(tmp_path / f"{tag}.bam")
        r = subprocess.run([CLI, "call", "-t", "8"] + extra + [src, dst], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-2000:]
        slabs = r.stderr.count("reads done")
        outs.append((tag, slabs, gzip.open(dst, "rb").read()))
    assert outs[0][1] >= 2 and outs[1][1] == 8 and outs[2][1] > 50, [(t, n) for t, n, _ in outs]
    # the @PG line records the command line (mod_main.cpp:101-117): compare everything behind the header text
    def body(raw):
        l_text = int.from_bytes(raw[4:8], "little")
        return raw[8 + l_text:]
    assert body(outs[0][2]) == body(outs[1][2]) == body(outs[2][2])
    _, recs = bamutil.read_bam(str(tmp_path / "default.bam"))
    assert len(recs) == len(reads) and sum(1 for x in recs if b"MM" in x["aux"]) > 1900


@pytest.mark.gpu
def test_call_dist_two_ranks_launched_like_a_multi_gpu_job(tmp_path):
    """python -m torch.distributed.run --nproc-per-node 2 -m hifimeth_amd.call_dist: one process per rank, each calls its
    BGZF-offset shard through the native front end (both on this box's one GPU; gloo for the barrier), rank 0 merges.
    Records equal the single-process run."""
    import socket
    import sys
    reads = synth_reads(24, seed=77, median_len=5000, sigma=0.4, frac_wide=0.1, frac_short=0.1, frac_missing=0.1)
    src, one, two = str(tmp_path / "in.bam"), str(tmp_path / "one.bam"), str(tmp_path / "two.bam")
    bamutil.reads_to_bam(src, reads, level=1)
    subprocess.check_call([CLI, "call", "-t", "4", src, one], stderr=subprocess.DEVNULL)
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    # launched exactly as documented: no HM_DIST_BACKEND override (the ranks' barrier is a CPU collective: gloo by default)
    env = dict(os.environ, PYTHONPATH=ROOT)
    env.pop("HM_DIST_BACKEND", None)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                        "127.0.0.1", "--master-port", str(port), "-m", "hifimeth_amd.call_dist", "-t", "4", src, two],
                       capture_output=True, text=True, env=env, cwd=ROOT, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    _, a = bamutil.read_bam(one)
    _, b = bamutil.read_bam(two)
    assert len(a) == len(b) == len(reads)
    for x, y in zip(a, b):
        assert x["name"] == y["name"] and x["aux"] == y["aux"]


def _modstats(path):
    import json
    return json.loads(subprocess.check_output([CLI, "modstats", path]))


def test_mm_ml_parser_roundtrip_and_histograms(tmp_path, oracle, oracle_models):
    """tagtest writes MM/ML from oracle calls; modstats parses them back (bam_mod_parser.cpp:231-286) and builds the
    per-context probability histograms of pileup.cpp:237-272 -- they must equal the histograms of the calls."""
    from oracle.modtags import resolve_threshold
    reads = synth_reads(9, seed=6, median_len=2200, sigma=0.3, frac_wide=0.2, frac_short=0.1, frac_missing=0.1)
    calls = _oracle_calls(oracle, oracle_models, reads)
    src, dst, cb = str(tmp_path / "in.bam"), str(tmp_path / "out.bam"), str(tmp_path / "calls.bin")
    bamutil.reads_to_bam(src, reads)
    calls.tofile(cb)
    subprocess.check_call([CLI, "tagtest", src, cb, dst])
    st = _modstats(dst)
    assert st["reads"] == len(reads) and st["calls"] == len(calls)
    assert st["reads_with_mods"] == len(np.unique(calls["read_id"]))
    for c, name in enumerate(("CpG", "CHG", "CHH")):
        want = np.bincount(calls["scaled_prob"][calls["ctx"] == c], minlength=256)
        assert st[name]["bins"] == want.tolist(), name
        thr, n = resolve_threshold(want)
        assert st[name]["threshold"] == thr and st[name]["samples_in_window"] == n
    # the untouched input has no MM/ML at all
    st0 = _modstats(src)
    assert st0["calls"] == 0 and st0["reads_with_mods"] == 0 and st0["CpG"]["threshold"] == 128


def _modparse_golden():
    import json
    return json.load(open(os.path.join(ROOT, "tests", "golden", "modparse.json")))["records"]


def test_mm_ml_parsers_match_the_reference_parser(tmp_path):
    """tests/golden/modparse.json: the REFERENCE's own parser core (s_parse_one_mod_list, bam_mod_parser.cpp:136-229, compiled
    in place: oracle/ref_build/ref_modparse_driver.cpp) on our writer's strings (both strands, a flag-16 read), the hand-worked
    known answers and other accepted dialects.  The oracle's parser, the host mirror and the CLI's C++ parser must list the
    same (qoff, strand, base, code, probability) in the same order."""
    from hifimeth_amd.pileup import parse_mods
    from hifimeth_amd.synth import read_from_ascii
    from oracle import pileup_oracle as P
    recs = _modparse_golden()
    assert len(recs) >= 15 and sum(len(r["mods"]) for r in recs) > 2000
    comp = bytes.maketrans(b"ACGTN", b"TGCAN")
    reads = []
    for i, r in enumerate(recs):
        want = [tuple(m) for m in r["mods"]]
        seq = r["seq"].encode()
        fwd = (seq.translate(comp)[::-1] if r["flag"] & 16 else seq).decode()
        assert [tuple(m) for m in P.parse_mods(fwd, r["mm"], r["ml"])] == want, i
        got = parse_mods(r["seq"], r["flag"], r["mm"], np.array(r["ml"], np.uint8))
        assert [(int(m["qoff"]), int(m["strand"]), m["unmod_base"].decode(), m["code"].decode(), int(m["prob"])) for m in got] == want, i
        k = np.zeros(len(seq), np.uint8)
        reads.append(read_from_ascii(seq, k, k, k, k, flag=r["flag"], name=f"r{i}"))
    src = str(tmp_path / "mods.bam")
    bamutil.reads_to_bam(src, reads, extra_aux=lambda i, rd: bamutil.aux_Z("MM", recs[i]["mm"]) + bamutil.aux_B("ML", np.array(recs[i]["ml"], np.uint8)))
    out = subprocess.run([CLI, "modlist", src], capture_output=True, text=True, check=True).stdout.split("\n")
    li = 0
    for i, r in enumerate(recs):
        n = int(out[li]); li += 1
        got = []
        for _ in range(n):
            q, st, ub, code, pr = out[li].split(); li += 1
            got.append((int(q), int(st), ub, code, int(pr)))
        assert got == [tuple(m) for m in r["mods"]], i


def test_tag_writer_output_is_what_the_reference_parser_reads_back(tmp_path):
    """the fixture's first records carry the calls their MM/ML were written from, and the reference's parser maps those strings
    back to exactly these calls (asserted when the fixture was made): the product writer must produce the same strings."""
    from hifimeth_amd.synth import read_from_ascii
    recs = [r for r in _modparse_golden() if "calls" in r]
    assert len(recs) >= 6
    reads, calls = [], []
    for i, r in enumerate(recs):
        seq = r["seq"].encode()
        k = np.zeros(len(seq), np.uint8)
        reads.append(read_from_ascii(seq, k, k, k, k, flag=r["flag"], name=f"w{i}"))
        rec = np.zeros(len(r["calls"]), CALL_DTYPE)
        for j, (q, st, mlb) in enumerate(r["calls"]):
            rec[j] = (i, q, st, 0, mlb, 0, mlb / 255.0)
        calls.append(rec)
    src, dst, cb = str(tmp_path / "in.bam"), str(tmp_path / "out.bam"), str(tmp_path / "calls.bin")
    bamutil.reads_to_bam(src, reads)
    np.concatenate(calls).tofile(cb)
    subprocess.check_call([CLI, "tagtest", src, cb, dst])
    _, out = bamutil.read_bam(dst)
    for r, rec in zip(recs, out):
        d = {t[0]: t for t in bamutil.parse_aux(rec["aux"])}
        assert d["MM"][2] == r["mm"] and d["ML"][2].tolist() == r["ml"]


def test_threshold_rule_known_answers():
    from oracle.modtags import resolve_threshold
    flat = [1000] * 256
    assert resolve_threshold(flat) == (20, 216000)            # first minimum of a flat window
    valley = [5000 - abs(i - 140) * 0 for i in range(256)]
    valley[140] = 11
    assert resolve_threshold(valley)[0] == 140
    few = [30] * 256                                            # 216 bins x 30 < 10000 samples -> fallback
    assert resolve_threshold(few) == (128, 6480)
    narrow = [0] * 256
    for i in range(100, 140):
        narrow[i] = 100000                                      # window narrower than 50 bins -> fallback
    assert resolve_threshold(narrow) == (128, 0)
    bimodal = [int(20000 * (np.exp(-((i - 30) / 18.0) ** 2) + np.exp(-((i - 225) / 15.0) ** 2))) + 12 for i in range(256)]
    thr, n = resolve_threshold(bimodal)
    assert thr == 20 + int(np.argmin(bimodal[20:236])) and 60 <= thr <= 200 and n > 10000   # first minimum of the valley


def test_parser_accepts_foreign_mm_dialects_and_rejects_garbage(tmp_path):
    """ChEBI code, '?' / '.' flags, multi-code lists (bam_mod_parser.cpp:36-77,150-160); malformed tags are errors."""
    rd = synth_reads(1, seed=9, median_len=1200, sigma=0.05, frac_short=0, frac_missing=0, frac_wide=0)[0]
    seq = rd.ascii()
    cpos = [i for i, b in enumerate(seq) if b == ord("C")]

    def bam_with(mm, ml):
        path = str(tmp_path / "x.bam")
        extra = lambda i, r: bamutil.aux_Z("MM", mm) + bamutil.aux_B("ML", np.array(ml, np.uint8))
        bamutil.reads_to_bam(path, [rd], extra_aux=extra)
        return path

    st = _modstats(bam_with("C+27551,0,1;", [200, 17]))                 # ChEBI 27551 = 5mC
    assert st["reads_with_mods"] == 1 and st["calls"] <= 2              # counted only where a context exists
    st = _modstats(bam_with("C+m?,0;C+h.,0;", [9, 8]))                  # flags skipped; two lists share ML in order
    assert st["reads_with_mods"] == 1
    st = _modstats(bam_with("C+mh,2;", [11, 12]))                       # two codes per position consume two ML entries
    assert st["reads_with_mods"] == 1
    for mm, ml in (("C+m,0", [1]), ("C+m,99999;", [1]), ("C+m,0,0;", [1]), ("X+m,0;", [1]), ("C+m;0;", [1])):
        p = bam_with(mm, ml)
        assert subprocess.call([CLI, "modstats", p], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL) != 0, mm


def test_cli_survives_corrupted_records(tmp_path):
    """mutation fuzz of the BAM record / aux / MM-ML parsers (CPU sub-commands): a corrupted file must end in exit
    code 0 or 1 with a message, never in a crash.  (The same loop was run once with an ASan+UBSan build: no reports.)"""
    import gzip
    import random
    from bamutil import aligned_to_bam, write_bgzf
    from hifimeth_amd.synth import synth_alignments, synth_genome
    genome = synth_genome(n_chr=1, length=3000)
    aligned_to_bam(str(tmp_path / "a.bam"), genome, synth_alignments(genome, 6, median_len=400))
    payload = bytearray(gzip.open(str(tmp_path / "a.bam"), "rb").read())
    start = payload.index(b"chr1\0") + 9
    rng = random.Random(7)
    mut = str(tmp_path / "m.bam")
    for _ in range(60):
        p = bytearray(payload)
        for _ in range(rng.choice([1, 1, 2, 4, 8])):
            p[rng.randrange(start, len(p))] = rng.randrange(256)
        write_bgzf(mut, bytes(p))
        for cmd in (["modstats", mut], ["bamcopy", mut, str(tmp_path / "o.bam")]):
            r = subprocess.run([CLI] + cmd, capture_output=True, timeout=60)
            assert r.returncode in (0, 1), (cmd[0], r.returncode, r.stderr[-300:])


def test_cli_sample_properties(tmp_path):
    """hifimeth-hip sample = src/app/hifimeth/subsample_bam.cpp.  The reference shuffles with a random_device seed, so
    the checks are its invariants: only reads with >= 5000 bases and all four kinetics arrays come out, in input
    order, byte-identical; the base total reaches coverage x reference size, and dropping the last pick would not."""
    if not os.path.exists(CLI):
        pytest.skip("CLI not built")
    reads = synth_reads(60, seed=5, median_len=7000, min_len=1000, max_len=12000, frac_missing=0.1, frac_short=0.05)
    src, fa = str(tmp_path / "in.bam"), str(tmp_path / "ref.fa")
    bamutil.reads_to_bam(src, reads)
    bamutil.write_fasta(fa, [("chr1", "ACGT" * 5000), ("chr2", "TTGCA" * 2000)])     # 30 000 bases
    usable = {r.name: r.l_qseq for r in reads
              if r.l_qseq >= 5000 and all(getattr(r, t) is not None and len(getattr(r, t)) == r.l_qseq for t in ("fi", "fp", "ri", "rp"))}
    assert 10 < len(usable) < len(reads)
    _, inp = bamutil.read_bam(src)
    raw = {d["name"]: d["raw"] for d in inp}
    order = [d["name"] for d in inp]
    picks = []
    for seed in (1, 2):
        dst = str(tmp_path / f"out{seed}.bam")
        r = subprocess.run([CLI, "sample", "-s", str(seed), fa, src, "3", dst], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        assert "DB size: 29.3KB" in r.stderr and "target size: 87.9KB" in r.stderr
        hdr, out = bamutil.read_bam(dst)
        names = [d["name"] for d in out]
        assert all(n in usable for n in names)
        assert names == [n for n in order if n in set(names)]                  # input order kept
        assert all(d["raw"] == raw[d["name"]] for d in out)                    # records untouched
        total = sum(usable[n] for n in names)
        assert total >= 90_000 and total - min(usable[n] for n in names) < 90_000 + max(usable.values())
        assert f"Extracted reads: {len(names)} " in r.stderr
        picks.append(tuple(names))
    assert picks[0] != picks[1]                                                # the seed changes the draw
    # a target beyond the file: every usable read, nothing else
    dst = str(tmp_path / "all.bam")
    assert subprocess.run([CLI, "sample", fa, src, "1000", dst], capture_output=True).returncode == 0
    assert [d["name"] for d in bamutil.read_bam(dst)[1]] == [n for n in order if n in usable]
    assert subprocess.run([CLI, "sample", fa, src, "3"], capture_output=True).returncode == 1
