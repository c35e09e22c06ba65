"""The CPU oracle against the fixtures generated from the reference itself (tools/make_golden.py).

scan.json    <- reference C++ scanner (eval_kmer_features.cpp:67-126) built by oracle/ref_build
windows.npz  <- reference Python assembler (training/sample_dataset.py:84-139)
cnn_*.npz    <- reference TorchScript models models/CpG.pt, models/CHH.pt; models/CHG.onnx via torch functional ops
softmax_ml.json <- reference s_logits_to_methy_probs (mod_batch.cpp:46-64) compiled in place by oracle/ref_build
"""
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN, WEIGHTS
from hifimeth_amd.synth import Read, read_from_ascii, synth_reads


def _scan_records():
    return json.load(open(os.path.join(GOLDEN, "scan.json")))


def test_scan_matches_reference_golden(oracle):
    recs = _scan_records()
    assert len(recs) >= 10
    for r in recs:
        rd = read_from_ascii(r["seq"].encode(), None, None, None, None, flag=r["flag"])
        fwd = oracle.decode(rd)
        assert fwd.decode() == r["fwd"]
        for c, key in enumerate(("cpg", "chg", "chh")):
            assert oracle.scan(fwd, c).tolist() == r[key], (key, r["seq"][:20])


def test_scan_matches_reference_binary_live(oracle):
    """Where oracle/_ref/ref_scan exists (build container), fuzz the oracle against it."""
    if not oracle.ref_scan_available():
        pytest.skip("oracle/_ref/ref_scan not built (reference absent)")
    reads = synth_reads(12, seed=99, median_len=3000, frac_n=0.002, frac_missing=0)
    ref = oracle.ref_scan([(r.flag, r.ascii().decode()) for r in reads])
    for r, d in zip(reads, ref):
        fwd = oracle.decode(r)
        assert fwd.decode() == d["fwd"]
        for c, key in enumerate(("cpg", "chg", "chh")):
            assert oracle.scan(fwd, c).tolist() == d[key]


def test_chh_emission_is_not_monotonic_but_set_is_right(oracle):
    # rev hit at i yields i+2 before a fwd hit at i+1 (eval_kmer_features.cpp:67-87)
    offs = oracle.scan(b"AAGCAA", 2).tolist()   # AAG (rev, i=0 -> 2), CAA (fwd, i=3)
    assert offs == [2, 3]
    offs = oracle.scan(b"TTGCTTGAGCAT", 2).tolist()
    assert sorted(offs) != offs or len(offs) > 0


def test_windows_match_reference_python(oracle):
    z = np.load(os.path.join(GOLDEN, "windows.npz"))
    n = 0
    for ri in range(int(z["n_reads"])):
        rd = Read("g", int(z[f"len_{ri}"]), 4, z[f"seq4_{ri}"], z[f"fi_{ri}"], z[f"fp_{ri}"], z[f"ri_{ri}"], z[f"rp_{ri}"])
        fwd = oracle.decode(rd)
        sel = np.nonzero(z["read"] == ri)[0]
        w, s = oracle.windows(rd, fwd, z["qoff"][sel])
        assert np.array_equal(s, z["strand"][sel])
        assert np.array_equal(w, z["windows"][sel])          # bit-exact
        n += len(sel)
    assert n == len(z["windows"]) >= 85
    # fixture must exercise both clipped ends and both strands
    assert (z["windows"][:, 0, :].sum(axis=1) == 0).any() and (z["windows"][:, 400, :].sum(axis=1) == 0).any()
    assert set(z["strand"].tolist()) == {0, 1}
    # ... and every codev1 code, in all four kinetic channels, on both strands, at both read ends (the reference assembler divides in
    # float64 and casts, inference divides in fp32: a code whose quotients differed would break the bit-exact comparison above)
    lut = (oracle.codev1_table().astype(np.float32) / np.float32(952))
    assert len(set(lut.tolist())) == 256
    W = z["windows"]
    left, right = W[:, 0, :].sum(axis=1) == 0, W[:, 400, :].sum(axis=1) == 0   # clipped at the read's start / end
    for strand in (0, 1):
        for end in (left, right):
            sel = (z["strand"] == strand) & end
            for ch in range(4, 8):
                seen = set(np.unique(W[sel][:, :, ch]).tolist())
                assert set(lut.tolist()) <= seen, (strand, ch, len(seen))


def test_codec_tables(oracle):
    t = oracle.codev1_table()
    assert t[0] == 0 and t[63] == 63 and t[64] == 64 and t[127] == 190 and t[128] == 192
    assert t[191] == 444 and t[192] == 448 and t[255] == 952
    # encode(decode(code)) == code ; encode clamps at 952 ; lossy in between
    assert all(oracle.encode_frames(int(t[c])) == c for c in range(256))
    assert oracle.encode_frames(5000) == 255 and oracle.encode_frames(65) == 64 and oracle.encode_frames(195) == 128


def test_codec_matches_the_reference_functions(oracle):
    """tests/golden/codec.json: the reference's own decode table (BamKinetics ctor) and s_encode_signal_value
    (bam_info.cpp:455-478, 568-576; compiled in place, oracle/ref_build/ref_codec_driver.cpp)"""
    g = json.load(open(os.path.join(GOLDEN, "codec.json")))
    assert oracle.codev1_table().tolist() == g["decode"]
    assert [oracle.encode_frames(s) for s in range(1200)] == g["encode_0_1199"]
    for s, c in g["encode_big"].items():
        assert oracle.encode_frames(int(s)) == c


def test_u16_kinetics_equal_reencoded_u8(oracle):
    rd = synth_reads(1, seed=3, median_len=1200, sigma=0.05, frac_wide=1.0, frac_missing=0, frac_short=0)[0]
    assert rd.fi.dtype == np.uint16
    enc = lambda a: np.array([oracle.encode_frames(int(v)) for v in a], np.uint8)
    rd8 = Read("x", rd.l_qseq, rd.flag, rd.seq4, enc(rd.fi), enc(rd.fp), enc(rd.ri), enc(rd.rp))
    fwd = oracle.decode(rd)
    offs = oracle.scan(fwd, 0)[:16]
    w16, _ = oracle.windows(rd, fwd, offs)
    w8, _ = oracle.windows(rd8, fwd, offs)
    assert np.array_equal(w16, w8)


@pytest.mark.parametrize("ctx", ["CpG", "CHG", "CHH"])
def test_cnn_matches_reference_torchscript(oracle, ctx):
    # CpG / CHH: outputs of the reference's TorchScript models; CHG: models/CHG.onnx evaluated with torch functional ops
    # over an independent minimal ONNX parse (CHG.pt holds another checkpoint) AND, since round 5, by the reference's own TorchScript
    # graph (models/CpG.pt: the same architecture) with CHG.onnx's tensors in the place of its constants (`logits_ts`) --
    # tools/make_golden.py:make_cnn_chg
    z = np.load(os.path.join(GOLDEN, f"cnn_{ctx}.npz"))
    m = oracle.Model(os.path.join(WEIGHTS, ctx + ".hmw"))
    lg = m.logits(z["windows"])
    assert np.abs(lg - z["logits"]).max() < 2e-5
    if ctx == "CHG":
        assert np.abs(z["logits_ts"] - z["logits"]).max() < 1e-5 and np.abs(lg - z["logits_ts"]).max() < 2e-5
    p, ml = oracle.softmax(lg)
    pr, _ = oracle.softmax(z["logits"])
    assert np.abs(p - pr).max() < 1e-5


def test_zero_window_known_answer(oracle):
    # known logits for an all-zero window (from models/CpG.pt and CHH.pt)
    z = np.zeros((1, 401, 8), np.float32)
    got = oracle.Model(os.path.join(WEIGHTS, "CpG.hmw")).logits(z)[0]
    assert np.allclose(got, [-0.2388, 0.2470], atol=2e-4)
    got = oracle.Model(os.path.join(WEIGHTS, "CHH.hmw")).logits(z)[0]
    assert np.allclose(got, [0.3964, -0.3984], atol=2e-4)


def test_softmax_ml_byte(oracle):
    lg = np.array([[0, 0], [-50, 50], [50, -50], [0.3, -0.2]], np.float32)
    p, ml = oracle.softmax(lg)
    assert ml.tolist() == [127, 255, 0, int(255 * p[3])]
    assert abs(p[0] - 0.5) < 1e-7


def test_softmax_ml_bytes_match_the_reference_function(oracle):
    """tests/golden/softmax_ml.json was made by the REFERENCE's own s_logits_to_methy_probs (mod_batch.cpp:46-64, compiled in
    place: oracle/ref_build/ref_softmax_driver.cpp, tools/make_golden.py softmax): 2 786 logit pairs incl. both sides of every
    255 * p = k truncation boundary.  The oracle's a10 must give the same byte for every pair (same float arithmetic, same libm)."""
    g = json.load(open(os.path.join(GOLDEN, "softmax_ml.json")))
    lg = np.array([[int(h[:8], 16), int(h[8:], 16)] for h in g["logits_hex"]], np.uint32).view(np.float32)
    p, ml = oracle.softmax(lg)
    assert ml.tolist() == g["ml"]
    assert len(set(g["ml"])) == 256
    # and the float probability is the value the byte was truncated from
    q = np.minimum((255 * p).astype(np.int32), 255)
    assert q.tolist() == g["ml"]


def test_reference_softmax_build_reproduces_the_fixture():
    import subprocess
    exe = os.path.join(os.path.dirname(GOLDEN), "..", "oracle", "_ref", "ref_softmax")
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref/ref_softmax not built (reference absent)")
    g = json.load(open(os.path.join(GOLDEN, "softmax_ml.json")))
    txt = str(len(g["ml"])) + "\n" + "\n".join(h[:8] + " " + h[8:] for h in g["logits_hex"]) + "\n"
    out = subprocess.run([exe], input=txt, capture_output=True, text=True, check=True).stdout.split()
    assert [int(x) for x in out] == g["ml"]


def test_geometry(oracle_models):
    assert [m.k1 for m in oracle_models] == [11, 11, 13]


def test_call_read_composition(oracle, oracle_models):
    rd = synth_reads(1, seed=21, median_len=1500, sigma=0.05, frac_wide=0, frac_missing=0, frac_short=0)[0]
    res = oracle.call_read(oracle_models, 0b111, rd)
    fwd = oracle.decode(rd)
    n = sum(len(oracle.scan(fwd, c)) for c in range(3))
    assert len(res["qoff"]) == n
    # CpG / CHG are forward-strand only, CHH both
    assert (res["strand"][res["ctx"] < 2] == 0).all() and (res["strand"][res["ctx"] == 2] == 1).any()
    # skipped when shorter than -l (mod_main.cpp:189-192)
    assert len(oracle.call_read(oracle_models, 0b111, rd, min_len=10 ** 6)["qoff"]) == 0


def test_config1_goldens_match_live_oracle(oracle, oracle_models):
    """The committed CPU-path outputs of the configs[0] stand-in still equal what the oracle computes now."""
    z = np.load(os.path.join(GOLDEN, "config1_calls.npz"))
    reads = synth_reads(int(z["n_reads"]), seed=20250220, gc=0.36, median_len=2400, sigma=0.35, frac_wide=0.2,
                        frac_short=0.1, frac_missing=0.1)
    k = 0
    for i, rd in enumerate(reads):
        if not rd.has_kinetics() or rd.l_qseq < 1000:
            continue
        r = oracle.call_read(oracle_models, 1, rd)
        order = np.lexsort((r["qoff"], r["strand"]))
        n = len(order)
        assert np.array_equal(z["cpg_qoff"][k:k + n], r["qoff"][order]) and (z["cpg_read"][k:k + n] == i).all()
        assert np.abs(z["cpg_p"][k:k + n] - r["p"][order]).max() < 1e-6
        k += n
    assert k == len(z["cpg_qoff"]) > 300
