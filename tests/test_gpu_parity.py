"""Parity of the HIP path (through the C ABI) against the CPU oracle and the golden fixtures.
Runs on the GPU box only:  python -m pytest tests -m gpu -x -q
Bars: site lists and windows bit-exact; probabilities |dp| <= 1e-4 (fp32 path, BASELINE.json
north_star); ML bytes within 1 LSB.  Ordered so that a failure localises: scan -> windows ->
CNN layers -> logits -> whole path."""
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN, WEIGHTS
from hifimeth_amd.synth import Read, read_from_ascii, synth_reads

pytestmark = pytest.mark.gpu

DP_TOL = 1e-4  # north_star: |dp| <= 1e-4 vs the fp32 CPU path


@pytest.fixture(scope="module", params=[(1, 1), (1, 0), (0, 1), (0, 0)],
                ids=["trunk-f16x3", "persite-f16x3", "trunk-fp32", "persite-fp32"])
def mc(request):
    """Every parity test runs in four modes: the default (conv1..conv4 as a dense trunk over every read position,
    split-half fp16 MFMA with fp32 accumulation), the same arithmetic with conv1..conv4 per site (front kernels), and
    both layouts again on plain fp32 MFMA.  All must meet the same |dp| <= 1e-4 bar."""
    from hifimeth_amd import MethylationCaller
    m = MethylationCaller(device=0, timing=True)
    m.set_option("precision", request.param[0])
    m.set_option("trunk", request.param[1])
    m.precision = request.param[0]
    yield m
    m.close()


def _kin(L, rng, wide=False):
    if wide:
        return [np.clip(np.rint(rng.gamma(2.0, 30.0, L)), 0, 2000).astype(np.uint16) for _ in range(4)]
    return [np.clip(np.rint(rng.gamma(2.0, 12.0, L)), 0, 255).astype(np.uint8) for _ in range(4)]


def _mixed_reads():
    rng = np.random.default_rng(17)
    reads = synth_reads(5, seed=31, median_len=2600, sigma=0.3, frac_wide=0.4, frac_short=0, frac_missing=0, frac_n=0.003)
    # exact-minimum length, a chunk-boundary length, stored-as-reverse read, homopolymers
    for L, flag in ((1000, 4), (1024, 4), (1025, 16), (2049, 0)):
        seq = "".join("ACGT"[i] for i in rng.choice(4, L))
        reads.append(read_from_ascii(seq.encode(), *_kin(L, rng, wide=(L == 1025)), flag=flag))
    reads.append(read_from_ascii(b"C" * 1100, *_kin(1100, rng)))
    reads.append(read_from_ascii(b"G" * 1030, *_kin(1030, rng)))
    reads.append(read_from_ascii(b"CG" * 600, *_kin(1200, rng)))
    reads.append(read_from_ascii((b"ACGTNCGNNCCGCANGCTGGGNAAGNTTGCNAACCNGG" * 30), *_kin(38 * 30, rng)))
    return reads


def test_scan_sites_match_oracle(mc, oracle):
    reads = _mixed_reads()
    mc.clear()
    assert mc.submit_all(reads) == len(reads)
    mc.upload()
    mc.run()
    total = 0
    for c in range(3):
        rid, qoff, strand = mc.scan_sites(c)
        want_r, want_q, want_s = [], [], []
        for i, rd in enumerate(reads):
            fwd = oracle.decode(rd)
            offs = np.sort(oracle.scan(fwd, c))       # device order is (read, qoff); reference emission differs for CHH
            want_r += [i] * len(offs)
            want_q += offs.tolist()
            want_s += [1 if fwd[o:o + 1] == b"G" else 0 for o in offs]
        assert np.array_equal(rid, want_r) and np.array_equal(qoff, want_q) and np.array_equal(strand, want_s), c
        assert mc.num_sites(c) == len(want_q)
        total += len(want_q)
    assert mc.num_sites(3) == total > 0


def test_scan_golden_reference_lists(mc):
    """Device scanner against the lists the reference's own C++ scanner produced (tests/golden/scan.json)."""
    rng = np.random.default_rng(3)
    recs = [r for r in json.load(open(os.path.join(GOLDEN, "scan.json")))]
    mc.set_option("min_read_size", 1)
    try:
        mc.clear()
        for i, r in enumerate(recs):
            L = len(r["seq"])
            assert mc.submit(i, read_from_ascii(r["seq"].encode(), *_kin(L, rng), flag=r["flag"]))
        mc.upload()
        mc.run()
        for c, key in enumerate(("cpg", "chg", "chh")):
            rid, qoff, _ = mc.scan_sites(c)
            for i, r in enumerate(recs):
                assert qoff[rid == i].tolist() == sorted(r[key]), (key, i)
    finally:
        mc.set_option("min_read_size", 1000)
        mc.clear()


def test_windows_bit_exact_vs_oracle(mc, oracle):
    reads = _mixed_reads()
    mc.clear()
    mc.submit_all(reads)
    mc.upload()
    mc.run()
    for c in range(3):
        rid, qoff, strand = mc.scan_sites(c)
        n = len(qoff)
        pick = np.unique(np.concatenate([np.arange(min(n, 40)), np.arange(max(0, n - 40), n),
                                         np.random.default_rng(c).integers(0, n, 60)]))
        got_all = mc.windows(c)
        assert got_all.shape == (n, 401, 8)
        for i in pick:
            rd = reads[rid[i]]
            w, s = oracle.window(rd, oracle.decode(rd), int(qoff[i]))
            assert s == strand[i]
            assert np.array_equal(got_all[i], w), (c, i, rid[i], qoff[i])


def test_windows_golden_reference_python(mc):
    """Device windows against windows assembled by the reference's training/sample_dataset.py."""
    z = np.load(os.path.join(GOLDEN, "windows.npz"))
    reads = [Read("g", int(z[f"len_{i}"]), 4, z[f"seq4_{i}"], z[f"fi_{i}"], z[f"fp_{i}"], z[f"ri_{i}"], z[f"rp_{i}"])
             for i in range(int(z["n_reads"]))]
    mc.set_option("min_read_size", 1)
    try:
        mc.clear()
        mc.submit_all(reads)
        mc.upload()
        mc.run()
        found = 0
        for c in range(3):
            rid, qoff, strand = mc.scan_sites(c)
            if len(qoff) == 0:
                continue
            wins = mc.windows(c)
            key = {(int(r), int(q)): k for k, (r, q) in enumerate(zip(rid, qoff))}
            for g in range(len(z["qoff"])):
                k = key.get((int(z["read"][g]), int(z["qoff"][g])))
                if k is None:
                    continue
                assert strand[k] == z["strand"][g]
                assert np.array_equal(wins[k], z["windows"][g])
                found += 1
        # every golden site is a C or G; those that are a site of some context must all have matched
        assert found >= len(z["qoff"]) // 2
    finally:
        mc.set_option("min_read_size", 1000)
        mc.clear()


@pytest.mark.parametrize("ctx,name", [(0, "CpG"), (1, "CHG"), (2, "CHH")])
def test_cnn_layers_vs_oracle(mc, oracle, ctx, name):
    z = np.load(os.path.join(GOLDEN, f"cnn_{name}.npz"))
    om = oracle.Model(os.path.join(WEIGHTS, name + ".hmw"))
    for widx in (0, 5, 61):
        w = z["windows"][widx]
        for layer in range(1, 9):
            want = om.layer(w, layer)
            got = mc.debug_layer(ctx, w, layer)
            assert got.shape == want.shape, (layer, got.shape, want.shape)
            err = np.abs(got - want).max()
            assert err <= 1e-4 * max(1.0, np.abs(want).max()), (name, widx, layer, err)


@pytest.mark.parametrize("ctx,name", [(0, "CpG"), (1, "CHG"), (2, "CHH")])
def test_cnn_logits_vs_reference_torchscript(mc, oracle, ctx, name):
    z = np.load(os.path.join(GOLDEN, f"cnn_{name}.npz"))
    lg, p, ml = mc.cnn_logits(ctx, z["windows"])
    assert np.abs(lg - z["logits"]).max() < 5e-5
    pr, mlr = oracle.softmax(z["logits"])
    assert np.abs(p - pr).max() <= DP_TOL
    assert np.abs(ml.astype(int) - mlr.astype(int)).max() <= 1


def test_cnn_logits_chg_and_ragged_batches(mc, oracle, oracle_models):
    rng = np.random.default_rng(5)
    for n in (1, 7, 8, 9, 300):
        w = np.zeros((n, 401, 8), np.float32)
        w[:, :, :4] = np.eye(4, dtype=np.float32)[rng.integers(0, 4, (n, 401))]
        w[:, :, 4:] = rng.gamma(2.0, 0.03, (n, 401, 4)).astype(np.float32)
        lg, p, ml = mc.cnn_logits(1, w)
        want = oracle_models[1].logits(w)
        assert np.abs(lg - want).max() < 5e-5
        pw, _ = oracle.softmax(want)
        assert np.abs(p - pw).max() <= DP_TOL


def _check_calls(calls, reads, oracle, oracle_models, mask):
    worst, nml, n = 0.0, 0, 0
    for rid, rd in enumerate(reads):
        got = calls[calls["read_id"] == rid]
        if not rd.has_kinetics() or rd.l_qseq < 1000:
            assert len(got) == 0
            continue
        want = oracle.call_read(oracle_models, mask, rd)
        order = np.lexsort((want["qoff"], want["strand"]))     # (strand, qoff): FWD calls then REV calls
        assert len(got) == len(order)
        assert np.array_equal(got["qoff"], want["qoff"][order])
        assert np.array_equal(got["strand"], want["strand"][order])
        assert np.array_equal(got["ctx"], want["ctx"][order])
        dp = np.abs(got["p"] - want["p"][order])
        worst = max(worst, float(dp.max(initial=0)))
        dml = np.abs(got["scaled_prob"].astype(int) - want["ml"][order].astype(int))
        assert dml.max(initial=0) <= 1
        nml += int((dml > 0).sum())
        n += len(got)
    assert worst <= DP_TOL, worst
    return n, nml, worst


def test_end_to_end_all_contexts(mc, oracle, oracle_models):
    reads = _mixed_reads() + synth_reads(4, seed=77, median_len=1800, sigma=0.2, frac_short=0.5, frac_missing=0.3)
    calls = mc.call(reads)
    n, nml, worst = _check_calls(calls, reads, oracle, oracle_models, 7)
    assert n > 3000
    # per-strand calls strictly increasing in qoff (asserted downstream by build_mod_bam.cpp:138,156)
    for rid in np.unique(calls["read_id"]):
        for s in (0, 1):
            q = calls["qoff"][(calls["read_id"] == rid) & (calls["strand"] == s)]
            assert (np.diff(q) > 0).all()
    print(f"e2e: {n} sites, max|dp|={worst:.2e}, ML bytes differing by 1: {nml}")


@pytest.mark.parametrize("spec,mask", [("cpg", 1), ("chg,chh", 6)])
def test_context_masks(oracle, oracle_models, spec, mask):
    from hifimeth_amd import MethylationCaller
    reads = synth_reads(3, seed=5, median_len=1500, sigma=0.1, frac_short=0, frac_missing=0)
    with MethylationCaller(contexts=spec) as m:
        calls = m.call(reads)
        assert set(np.unique(calls["ctx"]).tolist()) <= {c for c in range(3) if mask >> c & 1}
        _check_calls(calls, reads, oracle, oracle_models, mask)


@pytest.mark.parametrize("waves", [4, 8])
def test_front_wave_configs(mc, oracle, oracle_models, waves):
    reads = synth_reads(2, seed=11, median_len=1600, sigma=0.1, frac_short=0, frac_missing=0)
    mc.set_option("front_waves", waves)
    try:
        calls = mc.call(reads)
        n, _, worst = _check_calls(calls, reads, oracle, oracle_models, 7)
        assert n > 500
    finally:
        mc.set_option("front_waves", 4)


def test_split_half_precision_mode(mc, oracle, oracle_models):
    """Option precision=1: conv1..conv4 on fp16 MFMA with hi+lo split operands and fp32 accumulation.
    Must hold the SAME bar as the fp32 path: |dp| <= 1e-4, ML within 1 LSB."""
    reads = _mixed_reads()[:8] + synth_reads(2, seed=12, median_len=4000, sigma=0.1, frac_short=0, frac_missing=0)
    z = np.load(os.path.join(GOLDEN, "cnn_CpG.npz"))
    om = oracle.Model(os.path.join(WEIGHTS, "CpG.hmw"))
    mc.set_option("precision", 1)
    try:
        for layer in (1, 2, 3, 4):
            want = om.layer(z["windows"][5], layer)
            got = mc.debug_layer(0, z["windows"][5], layer)
            assert got.shape == want.shape
            assert np.abs(got - want).max() <= 2e-5 * max(1.0, np.abs(want).max()), layer
        lg, p, ml = mc.cnn_logits(0, z["windows"])
        pr, mlr = oracle.softmax(z["logits"])
        assert np.abs(p - pr).max() <= DP_TOL
        calls = mc.call(reads)
        n, nml, worst = _check_calls(calls, reads, oracle, oracle_models, 7)
        assert n > 3000
        print(f"f16x3: {n} sites, max|dp|={worst:.2e}, ML bytes differing by 1: {nml}")
    finally:
        mc.set_option("precision", mc.precision)


def test_fp16_weight_mode_holds_its_bar_and_the_literal_config_stays_closed(oracle, oracle_models):
    """BASELINE.json configs[4] (plain fp16 CNN weights, |dp| <= 1e-3).  AS WRITTEN -- fp16 weights in every layer -- it was measured in
    rounds 1-2 at 2.5e-3 and stays closed (`precision` 3 is an error, not a silent fallback).  What holds the bar with margin is offered
    as `precision` = 2: plain fp16 weights (the w_lo x_hi product dropped) in conv8 and fc1 -- tail_kernel_r<true> for CpG / CHG, the strip kernel's
    run-time w16 conv8 + tail_fc_kernel<true> for CHH, byte-identical to tail_kernel_r<true> --;
    activations stay hi + lo (profiles/r05_parity_sweep_precision2.txt: the multi-million-site sweep; conv3 alone -- the layer VERDICT r04
    proposed -- reaches 1.3e-3 there: profiles/r05_parity_sweep_conv3_w16.txt).  Here: the mode is accepted, it is
    a different arithmetic from mode 1 (some p differ), and its calls hold 1e-3 against the fp32 oracle on configs[2]-like reads (GC 0.36)
    and on human-like ones (GC 0.41, CpG depleted), site lists and order unchanged."""
    from hifimeth_amd import HifimethError, MethylationCaller
    from hifimeth_amd.synth import synth_slab
    with MethylationCaller(device=0) as m:
        with pytest.raises(HifimethError):
            m.set_option("precision", 3)
    for gc, oe in ((0.36, 1.0), (0.41, 0.24)):
        reads = synth_slab(8, seed=411, gc=gc, cpg_oe=oe, median_len=7000, sigma=0.4, frac_wide=0.2)
        with MethylationCaller(device=0) as m:
            m.set_option("trunk", 1)
            ref = m.call(reads).copy()
        with MethylationCaller(device=0, timing=True) as m:
            m.set_option("trunk", 1)
            m.set_option("precision", 2)
            got = m.call(reads).copy()
            assert m.timing()["tail_strip_passes"] > 0, "CHH runs on the strip tail in this mode too (w16: a run-time parameter of the kernel)"
        with MethylationCaller(device=0) as m:   # the same mode on tail_kernel_r<true> for every context: the same bytes
            m.set_option("trunk", 1)
            m.set_option("precision", 2)
            m.set_option("tail_impl", 1)
            assert m.call(reads).tobytes() == got.tobytes()
        assert len(got) == len(ref) > 5000
        for f in ("read_id", "qoff", "strand", "ctx"):
            assert np.array_equal(got[f], ref[f])
        assert (got["p"] != ref["p"]).mean() > 0.5 and float(np.abs(got["p"] - ref["p"]).max()) < 1e-3
        worst = 0.0
        for rid, rd in enumerate(reads):
            if not rd.has_kinetics() or rd.l_qseq < 1000:
                continue
            want = oracle.call_read(oracle_models, 7, rd)
            g = got[got["read_id"] == rid]
            order = np.lexsort((want["qoff"], want["strand"]))
            assert len(g) == len(order) and np.array_equal(g["qoff"], want["qoff"][order])
            worst = max(worst, float(np.abs(g["p"] - want["p"][order]).max()))
            assert int(np.abs(g["scaled_prob"].astype(int) - want["ml"][order].astype(int)).max()) <= 1
        assert worst <= 1e-3, worst   # the tolerance BASELINE.json configs[4] states


def test_group_size_option_is_bounded():
    """group_bases: 0 = sized from free device memory; an explicit size is capped at 48 Mi bases (190 GB of maps; the edge kernel addresses
    a group's map rows in 27 bits).  Out-of-range values are errors, and the engine keeps working afterwards."""
    from hifimeth_amd import HifimethError, MethylationCaller
    with MethylationCaller(device=0, timing=True) as m:
        for bad in (-1, (48 << 20) + 1, 1 << 40):
            with pytest.raises(HifimethError):
                m.set_option("group_bases", bad)
        m.set_option("group_bases", 48 << 20)
        m.set_option("group_bases", 0)
        reads = synth_reads(3, seed=5, median_len=2000, sigma=0.2)
        assert len(m.call(reads)) > 100
        assert 0 < m.timing()["group_bases"] <= 16 << 20


@pytest.mark.parametrize("spec,tag", [("cpg", "cpg"), ("cpg,chg,chh", "all")])
def test_config1_committed_goldens(spec, tag):
    """BASELINE.json configs[1]: CpG-only (and all-context) calls of a fixed read set against the COMMITTED CPU-path
    outputs (tests/golden/config1_calls.npz, generated by tools/make_golden.py) -- no oracle needed at run time."""
    from hifimeth_amd import MethylationCaller
    z = np.load(os.path.join(GOLDEN, "config1_calls.npz"))
    reads = synth_reads(int(z["n_reads"]), seed=20250220, gc=0.36, median_len=2400, sigma=0.35, frac_wide=0.2,
                        frac_short=0.1, frac_missing=0.1)
    with MethylationCaller(contexts=spec) as m:
        calls = m.call(reads)
    assert np.array_equal(calls["read_id"], z[f"{tag}_read"]) and np.array_equal(calls["qoff"], z[f"{tag}_qoff"])
    assert np.array_equal(calls["strand"], z[f"{tag}_strand"]) and np.array_equal(calls["ctx"], z[f"{tag}_ctx"])
    assert np.abs(calls["p"] - z[f"{tag}_p"]).max() <= DP_TOL
    assert np.abs(calls["scaled_prob"].astype(int) - z[f"{tag}_ml"].astype(int)).max() <= 1


def test_very_long_read_and_many_small_reads(mc, oracle, oracle_models):
    """A 70 kb read (69 scan chunks, l_qseq > 65535) next to a crowd of minimum-length reads."""
    rng = np.random.default_rng(23)
    L = 70001
    long_read = read_from_ascii("".join("ACGT"[i] for i in rng.choice(4, L, p=[0.32, 0.18, 0.18, 0.32])).encode(), *_kin(L, rng))
    small = [read_from_ascii("".join("ACGT"[i] for i in rng.choice(4, 1000)).encode(), *_kin(1000, rng)) for _ in range(40)]
    reads = small[:20] + [long_read] + small[20:]
    calls = mc.call(reads)
    n, _, worst = _check_calls(calls, reads, oracle, oracle_models, 7)
    assert n > 30000 and (calls["read_id"] == 20).sum() > 15000


def test_empty_and_skipped(mc):
    mc.clear()
    mc.upload()
    mc.run()
    assert mc.num_sites(3) == 0 and len(mc.fetch()) == 0
    rng = np.random.default_rng(1)
    short = read_from_ascii(b"ACGT" * 100, *_kin(400, rng))
    missing = read_from_ascii(b"ACGT" * 300, *_kin(1200, rng))
    missing.rp = None
    badlen = read_from_ascii(b"ACGT" * 300, *_kin(1200, rng))
    badlen.fi = badlen.fi[:-1]
    mc.clear()
    assert not mc.submit(0, short) and not mc.submit(1, missing) and not mc.submit(2, badlen)
    mc.upload()
    mc.run()
    assert len(mc.fetch()) == 0
    mc.clear()


def test_illegal_base_is_reported(mc):
    from hifimeth_amd import HifimethError
    rng = np.random.default_rng(2)
    rd = read_from_ascii(b"ACGT" * 300, *_kin(1200, rng))
    rd.seq4 = rd.seq4.copy()
    rd.seq4[10] = 0x31  # nibble 3 ('M') is not A/C/G/T/N: the reference aborts (bam_info.cpp:100-121)
    mc.clear()
    mc.submit(0, rd)
    mc.upload()
    mc.run()
    with pytest.raises(HifimethError):
        mc.sync()
    mc.clear()


def test_idempotent_rerun_and_sub_batches(mc):
    """Same resident batch run twice, and with a tiny sub-batch size, gives identical bytes."""
    reads = synth_reads(3, seed=9, median_len=2000, sigma=0.1, frac_short=0, frac_missing=0)
    mc.clear()
    mc.submit_all(reads)
    mc.upload()
    mc.run()
    a = mc.fetch().copy()
    mc.run()
    b = mc.fetch().copy()
    mc.set_option("sub_batch_sites", 64)
    mc.run()
    c = mc.fetch().copy()
    mc.set_option("sub_batch_sites", 65536)
    mc.clear()
    assert a.tobytes() == b.tobytes() == c.tobytes() and len(a) > 0


def test_full_size_properties(mc):
    """BASELINE-size batch (too big for the oracle): size-independent properties.
    count identity, strand/ctx consistency with the sequence, sortedness, probability range, and
    equality of results between one big batch and the same reads split over two batches."""
    reads = synth_reads(48, seed=123, frac_missing=0.02)
    calls = mc.call(reads)
    assert len(calls) > 100_000
    assert np.isfinite(calls["p"]).all() and (calls["p"] >= 0).all() and (calls["p"] <= 1).all()
    assert (calls["scaled_prob"] == np.minimum(255, (255 * calls["p"]).astype(np.int64))).all()
    key = calls["read_id"].astype(np.int64) * 4 + calls["strand"]
    assert (np.diff(key) >= 0).all()
    same = np.diff(key) == 0
    assert (np.diff(calls["qoff"])[same] > 0).all()
    for rid in (0, 17, 47):
        rd = reads[rid]
        sub = calls[calls["read_id"] == rid]
        if not rd.has_kinetics():
            assert len(sub) == 0
            continue
        seq = np.frombuffer(rd.ascii(), np.uint8)
        assert (seq[sub["qoff"][sub["strand"] == 0]] == ord("C")).all()
        assert (seq[sub["qoff"][sub["strand"] == 1]] == ord("G")).all()
        assert (sub["ctx"][sub["strand"] == 1] == 2).all()
        nxt = seq[np.minimum(sub["qoff"][sub["ctx"] == 0] + 1, len(seq) - 1)]
        assert (nxt == ord("G")).all()
    half = len(reads) // 2
    a = mc.call(reads[:half], first_id=0)
    b = mc.call(reads[half:], first_id=half)
    assert np.concatenate([a, b]).tobytes() == calls.tobytes()


def test_extreme_kinetics_and_homopolymers(mc, oracle, oracle_models):
    """saturated (255 = 952 frames), zero and alternating kinetics codes, u16 frame arrays at and above the 952 cap,
    on poly-C / CG-repeat / random sequence: the largest activations the network can see must stay finite in the
    split-half planes and within tolerance of the fp32 oracle."""
    from hifimeth_amd.synth import read_from_ascii
    rng = np.random.default_rng(5)
    L = 1400
    seqs = [b"C" * L, b"CG" * (L // 2), bytes(rng.choice(np.frombuffer(b"ACGT", np.uint8), L).tobytes()),
            b"CCA" * (L // 3) + b"CC"]
    reads = []
    for i, sq in enumerate(seqs):
        for j, fill in enumerate((255, 0, None)):
            if fill is None:
                k = [np.where(np.arange(len(sq)) % 2 == 0, 255, 0).astype(np.uint8) for _ in range(4)]
            else:
                k = [np.full(len(sq), fill, np.uint8) for _ in range(4)]
            reads.append(read_from_ascii(sq, *k, flag=4 if (i + j) % 2 else 20, name=f"x{i}_{j}"))
        wide = [np.full(len(sq), v, np.uint16) for v in (952, 2000, 65535, 447)]
        reads.append(read_from_ascii(sq, *wide, flag=4, name=f"w{i}"))
    calls = mc.call(reads)
    assert np.isfinite(calls["p"]).all()
    n, nml, worst = _check_calls(calls, reads, oracle, oracle_models, 7)
    assert n > 5000
    print(f"extreme kinetics: {n} sites, max|dp|={worst:.2e}")


def test_batch_pipeline_matches_synchronous_calls(mc):
    """The asynchronous batch pipeline (hm_batch_*: staging of slab k+1 overlapping the compute of slab k, site counts
    consumed on the device, one packed D2H) returns byte-identical records to the synchronous calls."""
    slabs = [synth_reads(n, seed=900 + i, gc=0.36 + 0.02 * i) for i, n in enumerate((9, 1, 14, 6, 11, 3))]
    want = [mc.call(s).copy() for s in slabs]
    got = []
    total = mc.stream(slabs, on_batch=lambda k, b, calls: got.append((calls.copy(), [b.num_sites(c) for c in range(4)])))
    assert total == sum(len(w) for w in want) and len(got) == len(slabs)
    for w, (g, ns) in zip(want, got):
        assert len(w) > 0 and w.tobytes() == g.tobytes()
        assert ns[3] == len(w) and sum(ns[:3]) == ns[3] and [int((w["ctx"] == c).sum()) for c in range(3)] == ns[:3]


def test_path_follows_site_density_of_the_first_batch(oracle, oracle_models):
    """Option trunk=2 (default): per context, conv1..conv4 run as the dense trunk or per site, whichever is cheaper at the
    site density of the engine's FIRST batch -- counted on the host when that batch is queued and then fixed, so the same
    input always takes the same kernels whatever the host timing (the reference's output is deterministic).  A CpG-only run
    on a CpG-poor genome (0.3 % sites per base) takes the per-site kernels from its first batch on, and keeps them when a
    CpG-rich batch follows; "trunk_mask" lets a front end make the choice itself; both paths hold the 1e-4 bar and agree to
    ~1e-5."""
    from hifimeth_amd import MethylationCaller
    from hifimeth_amd.caller import ReadBlock, trunk_mask_for_reads
    reads = synth_reads(6, seed=123, gc=0.11, median_len=5000, sigma=0.2, frac_short=0, frac_missing=0, frac_wide=0)
    rich = synth_reads(3, seed=124, gc=0.5, median_len=3000, sigma=0.2, frac_short=0, frac_missing=0, frac_wide=0)
    assert trunk_mask_for_reads(ReadBlock(reads), 1) == 0 and trunk_mask_for_reads(ReadBlock(rich), 1) == 1
    with MethylationCaller(contexts="cpg", timing=True) as m:
        b = m.call(reads).copy()
        t1 = m.timing(reset=True)
        b2 = m.call(reads).copy()
        m.call(rich)
        t2 = m.timing(reset=True)
        assert sum(t1["trunk_launches"]) == 0 and sum(t1["front_launches"]) > 0      # decided before the first launch
        assert sum(t2["trunk_launches"]) == 0 and sum(t2["front_launches"]) > 0      # and kept
        assert b.tobytes() == b2.tobytes()
        m.set_option("trunk_mask", 1)
        a = m.call(reads).copy()
        t3 = m.timing()
        assert sum(t3["trunk_launches"]) > 0 and sum(t3["front_launches"]) == 0
    assert len(a) == len(b) > 20 and np.array_equal(a["qoff"], b["qoff"]) and np.abs(a["p"] - b["p"]).max() < 2e-5
    for calls in (a, b):
        n, nml, worst = _check_calls(calls, reads, oracle, oracle_models, 1)
        assert n == len(calls) and worst <= DP_TOL


def test_trunk_groups_do_not_change_the_calls(oracle, oracle_models):
    """The dense trunk cuts a batch into read groups that reuse one set of map buffers (engine option group_bases; default: sized
    from free device memory, at most 16 Mi bases -- bench.py's 182-Mbase steps run ~15 groups per context).  With group_bases = 32 Ki the same reads fall into many groups
    -- a read that ends a group, a read that starts one, a 70 kb read that is a group of its own, a crowd of minimum-length
    reads -- and the calls must be byte-identical to the one-group run and within 1e-4 of the oracle: results may not
    depend on the batch cut (mod_main.cpp:330-362), in the split-half and in the strict-fp32 arithmetic."""
    from hifimeth_amd import MethylationCaller
    rng = np.random.default_rng(41)

    def rnd(L, wide=False):
        return read_from_ascii("".join("ACGT"[i] for i in rng.choice(4, L, p=[0.32, 0.18, 0.18, 0.32])).encode(), *_kin(L, rng, wide))

    # group boundaries at >= 32768 bases: [A 20000, B 13000] | [C 70001] | [D 32768] | [E 1000 x 33] | [F 9000, G 24000] | ...
    reads = [rnd(20000), rnd(13000), rnd(70001), rnd(32768), *[rnd(1000) for _ in range(33)], rnd(9000), rnd(24000, wide=True)]
    reads += synth_reads(6, seed=43, median_len=12000, sigma=0.4, frac_wide=0.3, frac_short=0, frac_missing=0)
    assert len(reads) >= 12
    for precision in (1, 0):
        with MethylationCaller(device=0, timing=True) as m:
            m.set_option("precision", precision)
            m.set_option("trunk", 1)
            one = m.call(reads).copy()
            t_one = m.timing(reset=True)
            m.set_option("group_bases", 32768)
            many = m.call(reads).copy()
            t_many = m.timing(reset=True)
            # the streamed form of the same thing: slabs of the asynchronous pipeline, several groups each
            got = []
            m.stream([reads[:4], reads[4:]], on_batch=lambda k, b, c: got.append(c.copy()))
        assert sum(t_one["trunk_launches"]) == 3 and sum(t_many["trunk_launches"]) >= 3 * 8, (t_one["trunk_launches"], t_many["trunk_launches"])
        assert len(one) > 60000 and one.tobytes() == many.tobytes()
        second = got[1].copy()
        second["read_id"] += 4
        assert np.concatenate([got[0], second]).tobytes() == one.tobytes()
        n, nml, worst = _check_calls(many, reads, oracle, oracle_models, 7)
        print(f"precision {precision}: {sum(t_many['trunk_launches'])} trunk launches, {n} sites, max|dp|={worst:.2e}, ML off by one: {nml}")


def test_bulk_submit_matches_read_by_read(mc):
    """hm_batch_submit_reads (one call per slab, copies on several host threads) stages exactly what hm_batch_submit_read
    stages read by read: same calls, byte for byte, including skipped reads (short / missing tag / B:S arrays)."""
    from hifimeth_amd.caller import ReadBlock
    reads = _mixed_reads() + synth_reads(12, seed=71, median_len=2200, sigma=0.5, frac_wide=0.3, frac_short=0.2, frac_missing=0.2)
    want = mc.call(reads).copy()
    blk = ReadBlock(reads)
    for threads in (1, 5):
        b = mc.begin_batch()
        n = b.submit_block(blk, threads)
        assert n == sum(1 for r in reads if r.has_kinetics() and r.l_qseq >= 1000)
        b.enqueue()
        got = b.wait().copy()
        b.release()
        assert got.tobytes() == want.tobytes()


def test_batches_staged_from_several_threads(mc):
    """Different batches may be staged by different host threads (the reference's workers pull from a shared queue,
    sam_batch.hpp:38-54); an empty batch and a batch of skipped reads go through the pipeline too."""
    import threading
    slabs = [synth_reads(7, seed=950 + i) for i in range(6)] + [[], synth_reads(4, seed=99, min_len=200, max_len=900)]
    want = [mc.call(s).copy() for s in slabs]
    got = [None] * len(slabs)
    errs = []

    def worker(ids):
        try:
            for i in ids:
                b = mc.begin_batch()
                b.submit_all(slabs[i])
                b.enqueue()
                got[i] = b.wait().copy()
                b.release()
        except Exception as ex:  # noqa: BLE001
            errs.append(ex)

    ts = [threading.Thread(target=worker, args=(range(k, len(slabs), 3),)) for k in range(3)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errs, errs
    for w, g in zip(want, got):
        assert w.tobytes() == g.tobytes()
    assert len(want[-1]) == 0 and len(want[-2]) == 0


def test_streaming_trunk_is_byte_identical_to_the_8_wave_form():
    """Engine option trunk_impl: the streaming 4-wave trunk kernel (weights resident in registers, positions streamed in
    tile groups, map rows copied out through row lists) keeps every accumulator's order of products, so its calls are
    byte-identical to the 8-wave kernel's -- including reads shorter than a tile, homopolymers (every row flagged) and
    wide kinetics."""
    from hifimeth_amd import MethylationCaller
    reads = _mixed_reads() + synth_reads(6, seed=77, median_len=5000, sigma=0.5, frac_wide=0.3)
    out = []
    for impl in (0, 1, 2, 3):  # 8-wave ConvH form, streaming on 4 waves, streaming on 8 waves, sliding window (default)
        with MethylationCaller(device=0) as m:
            m.set_option("trunk", 1)
            m.set_option("trunk_impl", impl)
            out.append(m.call(reads).copy())
    assert len(out[0]) == len(out[1]) == len(out[2]) == len(out[3]) > 1000
    assert out[0].tobytes() == out[1].tobytes() == out[2].tobytes() == out[3].tobytes()


@pytest.mark.gpu
def test_sliding_window_trunk_is_byte_identical_whatever_the_runs_of_tiles():
    """trunk_impl = 3 (hm_trunk3.hip): a workgroup walks a contiguous run of tiles and keeps every layer's right-hand rows for the next
    tile; kept rows are rebuilt by a warm-up step at the start of a run (at a read's start they are the constant rows).  Where the runs are cut depends on the number
    of workgroups (num_cu) and on the read groups: 1, 3, 7, 256 workgroups and many small groups -- runs that start in the middle of
    a read, runs of a single tile, reads of a single tile, more workgroups than tiles -- all give the streaming trunk's calls byte for
    byte, second batches through the same engine included."""
    from hifimeth_amd import MethylationCaller
    reads = _mixed_reads() + synth_reads(9, seed=91, median_len=4000, sigma=0.7, frac_wide=0.3)
    with MethylationCaller(device=0) as m:
        m.set_option("trunk", 1)
        m.set_option("trunk_impl", 1)
        ref = m.call(reads).copy()
    assert len(ref) > 1000
    for num_cu, group_bases in ((1, 0), (3, 0), (7, 32768), (256, 0), (1024, 4096)):
        with MethylationCaller(device=0) as m:
            m.set_option("trunk", 1)
            m.set_option("trunk_impl", 3)
            m.set_option("num_cu", num_cu)
            if group_bases:
                m.set_option("group_bases", group_bases)
            a = m.call(reads).copy()
            b = m.call(reads[::-1]).copy()     # other reads first: stale kept rows / maps of the previous batch must not leak
            c = m.call(reads).copy()
        assert a.tobytes() == ref.tobytes() == c.tobytes(), (num_cu, group_bases)
        assert len(b) == len(ref)


@pytest.mark.gpu
def test_constant_steps_store_what_a_computed_step_would():
    """hm_trunk3.hip, constant steps: a read's warm-up step, its first tile (u = -200) and the tiles behind its end (u >= len) lie where
    no receptive field reaches the read; the kernel stores the constant rows of a calibration step there instead of computing them.
    The calls stay byte-identical to the streaming trunk's (every position computed) -- with sites in the first 27 and the last
    positions of a read (homopolymers: their edge chains read E1 .. E3 rows of the warm-up step and of the tiles behind the end), reads
    of a few tiles, one workgroup and many -- and the kernel's own count of constant tiles is the one the tile plan gives:
    per read and strand view the first tile + the tiles at u >= len."""
    from hifimeth_amd import MethylationCaller
    reads = _mixed_reads() + synth_reads(7, seed=93, median_len=3000, sigma=0.6, frac_wide=0.3, frac_short=0, frac_missing=0)
    with MethylationCaller(device=0) as m:
        m.set_option("trunk", 1)
        m.set_option("trunk_impl", 1)
        ref = m.call(reads).copy()
    want = 0
    for rd in reads:
        L = rd.l_qseq
        ntile = (L + 400 + 111) // 112
        want += 1 + sum(1 for t in range(ntile) if -200 + 112 * t >= L)
    for num_cu in (1, 5, 256):
        with MethylationCaller(device=0, timing=True) as m:
            m.set_option("trunk", 1)
            m.set_option("trunk_impl", 3)
            m.set_option("num_cu", num_cu)
            got = m.call(reads).copy()
            tm = m.timing()
        assert got.tobytes() == ref.tobytes(), num_cu
        assert list(tm["trunk_const_steps"]) == [want, want, 2 * want], (num_cu, tm["trunk_const_steps"], want)
        tiles = [p // 112 for p in tm["trunk_positions"]]
        assert all(0 < c < t // 4 for c, t in zip(tm["trunk_const_steps"], tiles))


def test_edge2_is_byte_identical_to_the_staging_edge_kernel():
    """Engine option edge_impl: edge2_kernel (hm_edge2.hip: taps read in place, taps on the zero padding skipped, map rows by
    LDS-DMA and weights a layer ahead) keeps every accumulator's order of products over its live k-blocks, and the blocks it
    skips were exact zeros: its edge rows, hence the calls, are byte-identical to edge_kernel's -- for site counts that are no
    multiple of 32, lists shorter than the grid (one read, CpG only), both K1 geometries (CpG / CHG: 11, CHH: 13), sites whose
    windows hang over the read's ends, and many trunk groups."""
    from hifimeth_amd import MethylationCaller
    reads = _mixed_reads() + synth_reads(8, seed=79, median_len=6000, sigma=0.5, frac_wide=0.3)
    cases = [("cpg,chg,chh", reads, None), ("cpg,chg,chh", reads[:1], None), ("cpg", reads[:3], None), ("chg", reads, None), ("chh", reads, 32768)]
    for spec, rs, group_bases in cases:
        out = []
        for impl in (0, 1):
            with MethylationCaller(contexts=spec, device=0, timing=True) as m:
                m.set_option("trunk", 1)
                m.set_option("edge_impl", impl)
                if group_bases:
                    m.set_option("group_bases", group_bases)
                out.append(m.call(rs).copy())
                out.append(m.call(rs).copy())      # a second batch through the same engine (buffers reused)
        assert len(out[0]) == len(out[2]) > 50, (spec, len(out[0]))
        assert out[0].tobytes() == out[1].tobytes() == out[2].tobytes() == out[3].tobytes(), spec


def test_resident_tail_is_byte_identical_to_the_streaming_tail():
    """Engine option tail_impl: the tail with conv5..conv7's weights resident in registers (hm_tail_r.hip: four waves, the
    6 x 7 tile pairs of conv5 dealt as pairs + singles, row-aligned LDS-DMA gather through a per-pass row table, conv5 in two
    parts around the late rows) keeps every accumulator's order of products: its calls are byte-identical to tail_kernel_h's
    -- for site counts that are no multiple of 8, lists shorter than the grid (one read, CpG only), homopolymers, wide
    kinetics and many trunk groups."""
    from hifimeth_amd import MethylationCaller
    reads = _mixed_reads() + synth_reads(8, seed=78, median_len=6000, sigma=0.5, frac_wide=0.3)
    cases = [("cpg,chg,chh", reads, None), ("cpg,chg,chh", reads[:1], None), ("cpg", reads[:3], None), ("chh", reads, 32768)]
    for spec, rs, group_bases in cases:
        out = []
        for impl in (0, 1, 2):   # 2 = the split tail (hm_tail_s.hip: conv5 + conv6 | conv7 .. softmax over 16 sites per pass)
            with MethylationCaller(contexts=spec, device=0, timing=True) as m:
                m.set_option("trunk", 1)
                m.set_option("tail_impl", impl)
                if group_bases:
                    m.set_option("group_bases", group_bases)
                out.append(m.call(rs).copy())
                out.append(m.call(rs).copy())      # a second batch through the same engine (buffers reused)
        assert len(out[0]) == len(out[2]) == len(out[4]) > 50, (spec, len(out[0]))
        assert out[0].tobytes() == out[1].tobytes() == out[2].tobytes() == out[3].tobytes() == out[4].tobytes() == out[5].tobytes(), spec
    # the split tail in launches of 64 sites (engine option tail_slice): many launch pairs per context, the hand-off buffer reused
    with MethylationCaller(device=0) as m:
        m.set_option("trunk", 1)
        m.set_option("tail_impl", 0)
        ref = m.call(reads).copy()
    with MethylationCaller(device=0) as m:
        m.set_option("trunk", 1)
        m.set_option("tail_impl", 2)
        m.set_option("tail_slice", 64)
        got = m.call(reads).copy()
    assert got.tobytes() == ref.tobytes()


@pytest.mark.gpu
def test_strip_tail_is_byte_identical_to_the_resident_tail():
    """Engine option tail_impl = 3: CHH's sites are visited in (first E4 map row mod 16, map row) order and a pass takes up to 16
    consecutive sites of one residue class that share one strip of 144 lattice rows in LDS (hm_tail_p.hip; hm_convp.h: the sites along
    the rows of the MFMA tiles, zero-padding taps skipped).  Every accumulator keeps its order of products, so the calls are
    byte-identical to tail_kernel_r's (and tail_kernel_h's) -- for one read, many reads, homopolymers (no CHH site at all / one every
    position), wide kinetics, reads stored reversed, trunk groups cut inside the slab, a second batch through the same engine, and grids
    of 1 / 7 / 256 workgroups (a workgroup's range of the sorted list cuts passes at arbitrary places)."""
    from hifimeth_amd import MethylationCaller
    reads = _mixed_reads() + synth_reads(12, seed=178, median_len=7000, sigma=0.5, frac_wide=0.3)
    cases = [("cpg,chg,chh", reads, None, None), ("chh", reads[:1], None, None), ("chh", reads, 32768, None), ("chh", reads, None, 1),
             ("chh", reads, 65536, 7)]
    for spec, rs, group_bases, ncu in cases:
        out = []
        for impl in (1, 3):
            with MethylationCaller(contexts=spec, device=0, timing=True) as m:
                m.set_option("trunk", 1)
                m.set_option("tail_impl", impl)
                if group_bases:
                    m.set_option("group_bases", group_bases)
                if ncu:
                    m.set_option("num_cu", ncu)
                out.append(m.call(rs).copy())
                out.append(m.call(rs[::-1]).copy())      # a second, different batch through the same engine (buffers reused)
        assert len(out[0]) == len(out[2]) > 50, (spec, len(out[0]))
        assert out[0].tobytes() == out[2].tobytes() and out[1].tobytes() == out[3].tobytes(), (spec, group_bases, ncu)
    # site densities from sparse to dense (GC 0.2 ... 0.7: CHH 0.16 ... 0.27 sites per base; the strip holds 11 ... 16 sites per pass) and
    # CpG-depleted, human-like reads: passes of every fill, classes that run dry inside a strip
    from hifimeth_amd.synth import synth_slab
    for gc, oe in ((0.2, 1.0), (0.5, 1.0), (0.7, 1.0), (0.41, 0.24)):
        rs = synth_slab(6, seed=int(1000 * gc), gc=gc, cpg_oe=oe, median_len=9000, sigma=0.5, frac_wide=0.2)
        out = []
        for impl in (1, 3):
            with MethylationCaller(contexts="chh", device=0) as m:
                m.set_option("trunk", 1)
                m.set_option("tail_impl", impl)
                out.append(m.call(rs).copy())
        assert len(out[0]) > 5000 and out[0].tobytes() == out[1].tobytes(), (gc, oe)


@pytest.mark.gpu
def test_fc_kernel_tiles_of_16_with_ragged_ends():
    """tail_fc_kernel (hm_tail_fc.hip) takes the class-sorted list in tiles of 16 positions, whatever pass a site was taken in: reads
    built to hold EXACTLY k CHH sites (poly-A with k isolated C: no G, so nothing on the reverse strand) for k around the tile size --
    one site, one short of a tile, a full tile, one more, two tiles and one -- give k calls, byte-identical to tail_kernel_r's, alone and
    together in one batch (launches of 1 .. 82 sites: last tiles of every fill; the strip kernel's stand-in slots write nothing new)."""
    from hifimeth_amd import MethylationCaller
    from hifimeth_amd.synth import read_from_ascii
    rng = np.random.default_rng(11)

    def read_with(k, L=2400):
        seq = bytearray(b"A" * L)
        for i in range(k):
            seq[300 + 53 * i] = ord("C")   # C A A: a CHH site; 53 apart: its own window, every residue class mod 16 in turn
        kin = [np.clip(np.rint(rng.gamma(2.0, 12.0, L)), 0, 255).astype(np.uint8) for _ in range(4)]
        return read_from_ascii(bytes(seq), *kin, name=f"k{k}")

    ks = [1, 15, 16, 17, 33]
    reads = [read_with(k) for k in ks]
    for rs, want in [([r], k) for r, k in zip(reads, ks)] + [(reads, sum(ks))]:
        out = []
        for impl in (1, 3):
            with MethylationCaller(contexts="chh", device=0) as m:
                m.set_option("trunk", 1)
                m.set_option("tail_impl", impl)
                out.append(m.call(rs).copy())
        assert len(out[0]) == len(out[1]) == want, (want, len(out[0]), len(out[1]))
        assert out[0].tobytes() == out[1].tobytes(), want


@pytest.mark.gpu
def test_human_like_reads_mix_the_per_site_and_the_trunk_path_in_one_engine(oracle, oracle_models):
    """BASELINE.json configs[3] statistics (bench.py --workload human_slice: GC 0.41, CpG depleted to observed / expected 0.24 -> ~1 %
    of the bases): with the default options CpG takes the per-site kernels while CHG and CHH take the dense trunk, in the same engine
    and the same batch -- the calls of all three contexts hold the 1e-4 bar against the oracle, in the order the reference writes
    them (mod_main.cpp:217-251), also when a second slab follows through the batch pipeline."""
    from hifimeth_amd import MethylationCaller
    from hifimeth_amd.synth import synth_slab
    reads = synth_slab(10, seed=97, gc=0.41, cpg_oe=0.24, median_len=6000, sigma=0.4, frac_wide=0.2)
    with MethylationCaller(device=0, timing=True) as m:
        calls = m.call(reads).copy()
        t = m.timing()
        again = m.call(reads).copy()
    assert t["front_launches"][0] > 0 and t["trunk_launches"][0] == 0, "CpG (1 % of the bases) takes the per-site kernels"
    assert t["trunk_launches"][1] > 0 and t["trunk_launches"][2] > 0 and t["front_launches"][1] == 0 and t["front_launches"][2] == 0
    assert calls.tobytes() == again.tobytes()
    n, nml, worst = _check_calls(calls, reads, oracle, oracle_models, 7)
    by_ctx = [int((calls["ctx"] == c).sum()) for c in range(3)]
    assert n == len(calls) and min(by_ctx) > 100 and by_ctx[0] < 0.02 * sum(r.l_qseq for r in reads), by_ctx
    print(f"human-like mix: {by_ctx} sites per context, max|dp|={worst:.2e}, ML bytes off by one: {nml}")


@pytest.mark.gpu
def test_two_engines_on_one_device_fit_at_the_default_group_size():
    """VERDICT r03 #6 / ADVICE r03: the default group size is taken from the device's FREE memory (a quarter of it at most for a group's
    maps, edge rows and row lists, at most 16 Mi bases), so two engines on one device -- or ranks sharing a GPU -- cannot run it out of
    memory; hm_get_timing reports the size in force and the bytes held."""
    import torch
    from hifimeth_amd import MethylationCaller
    reads = synth_reads(12, seed=93, median_len=5000, sigma=0.4)
    free0, total = torch.cuda.mem_get_info(0)
    with MethylationCaller(device=0, timing=True) as a, MethylationCaller(device=0, timing=True) as b:
        ca = a.call(reads).copy()
        cb = b.call(reads).copy()
        ta, tb = a.timing(), b.timing()
    assert ca.tobytes() == cb.tobytes() and len(ca) > 1000
    for t in (ta, tb):
        assert (1 << 20) <= t["group_bases"] <= (16 << 20) and t["group_bases"] % (1 << 20) == 0, t["group_bases"]
        assert 0 < t["group_bytes"] <= free0 // 4 + (64 << 20), (t["group_bytes"], free0)
        assert t["group_bases"] * 5800 <= free0 // 4 + (1 << 30), (t["group_bases"], free0)
