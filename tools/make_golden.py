#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ from the REFERENCE itself.

Runs only in the build container (needs /root/reference); the fixtures are committed so the
GPU box never needs the reference.  Three sources, each the reference's own code or data:

  scan.json      site lists from the reference's C++ scanner, compiled from its sources by
                 oracle/ref_build (-> oracle/_ref/ref_scan):
                 src/app/hifimeth/eval_kmer_features.cpp:67-126, src/corelib/bam_info.cpp:169-222
  windows.npz    401x8 windows from the reference's Python training-time assembler
                 training/sample_dataset.py:84-139 (imported, not copied)
  cnn_<ctx>.npz  logits of the reference's shipped TorchScript models models/CpG.pt, CHH.pt
                 (torch.jit.load on CPU).  CHG.pt holds a different checkpoint than CHG.onnx
                  and is not used.

  align.json     alignment columns, identity and mapped CpG/CHG/CHH samples from the reference's BamMapInfo and
                 5mc_motif_finder.cpp (oracle/_ref/ref_align) -- pins the pileup oracle's projection

usage: python tools/make_golden.py [/root/reference [align]]
"""
import json
import struct
import os
import sys

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
sys.dont_write_bytecode = True
sys.path.insert(0, os.path.join(REF, "training"))

import torch  # noqa: E402

from hifimeth_amd.synth import Read, pack_codes, synth_reads  # noqa: E402
from oracle import hm_oracle as O  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
os.makedirs(GOLD, exist_ok=True)


def make_scan():
    rng = np.random.default_rng(7)
    recs = []

    def add(seq, flag=4):
        recs.append((flag, seq))

    for L, gc in ((1500, 0.36), (1200, 0.5), (1003, 0.7), (64, 0.5), (3, 0.5), (2, 0.5), (1, 0.5)):
        p = [(1 - gc) / 2, gc / 2, gc / 2, (1 - gc) / 2]
        add("".join("ACGT"[i] for i in rng.choice(4, L, p=p)))
    add("CG" * 40)
    add("C" * 50 + "G" * 50)
    add("CCGCAGCTGCGGCHHH".replace("H", "A") * 5)
    add("ACGTNCGNNCCGCANGCTGGGNAAGNTTGCNAACCNGG" * 8)           # N breaks motifs
    add("".join("ACGT"[i] for i in rng.choice(4, 1100)), flag=16)  # stored as reverse strand
    add("".join("ACGT"[i] for i in rng.choice(4, 1100)), flag=0)
    out = O.ref_scan(recs)
    js = [dict(flag=f, seq=s, **d) for (f, s), d in zip(recs, out)]
    json.dump(js, open(os.path.join(GOLD, "scan.json"), "w"))
    print(f"scan.json: {len(js)} records, {sum(len(d['cpg']) + len(d['chg']) + len(d['chh']) for d in js)} sites")


def make_windows():
    import sample_dataset as SD  # the reference's module

    rng = np.random.default_rng(11)
    reads = [r for r in synth_reads(3, seed=5, median_len=1400, sigma=0.1, frac_wide=0, frac_short=0, frac_missing=0)]
    # one very short read so that both window ends are clipped at once
    L = 260
    codes = rng.choice(4, L).astype(np.uint8)
    kin = [np.clip(np.rint(rng.gamma(2.0, 12.0, L)), 0, 255).astype(np.uint8) for _ in range(4)]
    kin[0][:8] = [0, 63, 64, 127, 128, 191, 192, 255]  # all four codec segments incl. boundaries
    reads.append(Read("short", L, 4, pack_codes(codes), *kin))
    # codec sweep (round 4): every codev1 code 0..255 in each of the four kinetics arrays within 256 bases of BOTH
    # read ends, and C and G sites 100 bases from each end -- so that windows of both strands, at both ends, hold every code in all
    # four kinetic channels.  The training assembler divides in float64 and casts (sample_dataset.py:49), inference divides in fp32
    # (eval_kmer_features.cpp:46-60): this is where a code whose two quotients differ would show.
    L = 700
    codes = rng.choice(4, L).astype(np.uint8)
    for q, c in ((96, 1), (104, 2), (L - 105, 1), (L - 97, 2), (3, 1), (5, 2), (L - 6, 1), (L - 4, 2)):
        codes[q] = c
    kin = []
    for ph in (0, 67, 131, 199):
        a = np.zeros(L, np.uint8)
        a[:256] = (np.arange(256) + ph) % 256
        a[L - 256:] = (np.arange(256)[::-1] + 3 * ph) % 256
        a[256:L - 256] = rng.integers(0, 256, L - 512)
        kin.append(a)
    reads.append(Read("codec_sweep", L, 4, pack_codes(codes), *kin))
    sweep_sites = {len(reads) - 1: [96, 104, L - 105, L - 97, 3, 5, L - 6, L - 4]}
    packs = dict(n_reads=len(reads))
    all_w, all_s, all_q, all_r = [], [], [], []
    for ri_, rd in enumerate(reads):
        fwd = rd.ascii()
        L = rd.l_qseq
        codes = np.frombuffer(fwd, np.uint8)
        lut = np.zeros(256, np.uint8)
        lut[[65, 67, 71, 84]] = [0, 1, 2, 3]
        codes = lut[codes]
        # dataset layout (sample_dataset.py:90-96): seq | fipd | fpw | ripd | rpw, all in FORWARD
        # coordinates; the BAM tags ri/rp are stored in reverse-strand order (bam_info.cpp:520-548
        # indexed with L-1-i at eval_kmer_features.cpp:54-60) -> flip them.
        feats = np.concatenate([codes, rd.fi, rd.fp, rd.ri[::-1], rd.rp[::-1]]).astype(np.uint8)
        offsets = np.array([(0, 0, L, -1, -1)], dtype=[('offset', np.int64), ('id', np.int32), ('size', np.int32),
                                                        ('fn', np.int32), ('rn', np.int32)])
        cg = np.nonzero((codes == 1) | (codes == 2))[0]
        pick = np.unique(np.concatenate([cg[:6], cg[-6:], rng.choice(cg, 6, replace=False), np.array(sweep_sites.get(ri_, []), np.int64)]))
        samples = np.array([[0, q, 1] for q in pick], dtype=np.uint64)
        for i in range(len(pick)):
            F, _ = SD.assemble_one_sample_features(feats, samples, offsets, 401, i)
            all_w.append(F.numpy())
            all_s.append(0 if codes[pick[i]] == 1 else 1)
            all_q.append(int(pick[i]))
            all_r.append(ri_)
        packs[f"seq4_{ri_}"] = rd.seq4
        packs[f"len_{ri_}"] = np.int32(L)
        for nm in ("fi", "fp", "ri", "rp"):
            packs[f"{nm}_{ri_}"] = getattr(rd, nm)
    packs.update(windows=np.stack(all_w).astype(np.float32), strand=np.array(all_s, np.uint8),
                 qoff=np.array(all_q, np.int32), read=np.array(all_r, np.int32))
    np.savez_compressed(os.path.join(GOLD, "windows.npz"), **packs)
    print(f"windows.npz: {len(all_w)} windows from {len(reads)} reads")
    # the reference's LUT (float64 divide, then cast) vs the inference path's fp32 divide
    lut64 = np.array(SD.codev1_to_frame_table, np.float64).astype(np.float32)
    lut32 = O.codev1_table().astype(np.float32) / np.float32(952)
    print("  LUT fp64-vs-fp32 divide mismatches:", int((lut64 != lut32).sum()))
    return np.stack(all_w).astype(np.float32)


def make_cnn(wins):
    rng = np.random.default_rng(13)
    extra = np.zeros((4, 401, 8), np.float32)          # all-zero window + random windows
    extra[1:, :, :4] = np.eye(4, dtype=np.float32)[rng.integers(0, 4, (3, 401))]
    extra[1:, :, 4:] = rng.gamma(2.0, 0.03, (3, 401, 4)).astype(np.float32)
    x = np.concatenate([wins[:60], extra])
    report = {}
    for ctx in ("CpG", "CHH"):
        m = torch.jit.load(os.path.join(REF, "models", ctx + ".pt"), map_location="cpu")
        with torch.no_grad():
            lg = m(torch.from_numpy(x)).numpy().astype(np.float32)
        np.savez_compressed(os.path.join(GOLD, f"cnn_{ctx}.npz"), windows=x, logits=lg)
        om = O.Model(os.path.join(ROOT, "hifimeth_amd", "weights", ctx + ".hmw"))
        d = float(np.abs(om.logits(x) - lg).max())
        report[ctx + ".pt_vs_oracle_max_dlogit"] = d
        print(f"cnn_{ctx}.npz: {len(x)} windows, oracle(ONNX weights) vs {ctx}.pt max |dlogit| = {d:.3g}")

    # structural check against the Python definition (training/model_cnn.py) with random weights:
    # fold its BatchNorms the way the ONNX export did and run the oracle on the folded weights.
    import model_cnn
    from hifimeth_amd.onnx_weights import ModelWeights, save_hmw
    torch.manual_seed(3)
    net = model_cnn.DNAModNet(401)
    with torch.no_grad():
        for mod in net.modules():
            if isinstance(mod, torch.nn.BatchNorm1d):
                mod.running_mean.normal_(0, 0.3)
                mod.running_var.uniform_(0.5, 1.5)
                mod.weight.uniform_(0.5, 1.5)
                mod.bias.normal_(0, 0.2)
    net.eval()
    cw, cb = [], []
    seq = list(net.convs)
    for i in range(8):
        conv, bn = seq[3 * i], seq[3 * i + 1]
        s = (bn.weight / torch.sqrt(bn.running_var + bn.eps)).detach()
        cw.append((conv.weight.detach() * s[:, None, None]).numpy())
        cb.append((bn.bias - bn.running_mean * s).detach().numpy())
    mw = ModelWeights(13, float(net.bn0.eps), net.bn0.weight.detach().numpy(), net.bn0.bias.detach().numpy(),
                      net.bn0.running_mean.numpy(), net.bn0.running_var.numpy(), cw, cb,
                      net.fc1.weight.detach().numpy(), net.fc1.bias.detach().numpy(),
                      net.fc2.weight.detach().numpy(), net.fc2.bias.detach().numpy())
    tmp = "/tmp/_dnamodnet_random.hmw"
    save_hmw(mw, tmp)
    with torch.no_grad():
        want = net(torch.from_numpy(x)).numpy()
    got = O.Model(tmp).logits(x)
    d = float(np.abs(want - got).max())
    report["DNAModNet_random_vs_oracle_max_dlogit"] = d
    print(f"DNAModNet(random weights, BN folded) vs oracle max |dlogit| = {d:.3g} (logit scale {np.abs(want).max():.3g})")
    os.remove(tmp)
    json.dump(report, open(os.path.join(GOLD, "cnn_report.json"), "w"), indent=1)


def _mini_onnx(path):
    """A second, MINIMAL reading of an ONNX file, written independently of hifimeth_amd/onnx_weights.py and of the C++
    reader (hm_weights.cpp): a flat protobuf wire walk that returns (initializers by name, [(op_type, inputs, attrs)]).
    Only what models/CHG.onnx uses (opset 17: initializers with raw_data, Conv / Relu / BatchNormalization / Gemm ...)."""
    buf = open(path, "rb").read()

    def varint(b, i):
        v = s = 0
        while True:
            c = b[i]; i += 1
            v |= (c & 127) << s; s += 7
            if c < 128:
                return v, i

    def walk(b):
        i = 0
        while i < len(b):
            key, i = varint(b, i)
            f, wt = key >> 3, key & 7
            if wt == 0:
                v, i = varint(b, i)
            elif wt == 2:
                n, i = varint(b, i); v = b[i:i + n]; i += n
            elif wt == 5:
                v = b[i:i + 4]; i += 4
            elif wt == 1:
                v = b[i:i + 8]; i += 8
            else:
                raise ValueError("wire type")
            yield f, wt, v

    graph = next(v for f, _, v in walk(buf) if f == 7)
    inits, nodes = {}, []
    for f, _, v in walk(graph):
        if f == 5:  # TensorProto: dims=1 data_type=2 name=8 raw_data=9
            dims, name, raw = [], None, None
            for g, wt, x in walk(v):
                if g == 1:
                    dims += [x] if wt == 0 else [t for t in _unpack(x)]
                elif g == 8:
                    name = x.decode()
                elif g == 9:
                    raw = x
            inits[name] = np.frombuffer(raw, "<f4").reshape(dims).copy()
        elif f == 1:  # NodeProto: input=1 output=2 op_type=4 attribute=5
            ins, op, attrs = [], None, {}
            for g, wt, x in walk(v):
                if g == 1:
                    ins.append(x.decode())
                elif g == 4:
                    op = x.decode()
                elif g == 5:
                    an, ai, af, ints = None, None, None, []
                    for h, wt2, y in walk(x):
                        if h == 1:
                            an = y.decode()
                        elif h == 3:
                            ai = y
                        elif h == 2:
                            af = struct.unpack("<f", y)[0]
                        elif h == 8:
                            ints += [y] if wt2 == 0 else list(_unpack(y))
                    attrs[an] = ints if ints else (ai if ai is not None else af)
            nodes.append((op, ins, attrs))
    return inits, nodes


def _unpack(b):
    i = 0
    while i < len(b):
        v = s = 0
        while True:
            c = b[i]; i += 1
            v |= (c & 127) << s; s += 7
            if c < 128:
                break
        yield v


def make_cnn_chg(wins):
    """models/CHG.onnx is the model the reference's CPU path loads (mod_main.cpp:85); its TorchScript twin CHG.pt holds a
    DIFFERENT checkpoint, so the fixture cannot come from torch.jit.load.  Instead: the ONNX graph is
    evaluated node by node with torch functional ops on tensors from the independent mini reader above -- an arithmetic
    path that shares no code with the repo's readers, oracle or kernels.  Labelled as such in cnn_report.json."""
    import torch.nn.functional as F
    rng = np.random.default_rng(13)
    extra = np.zeros((4, 401, 8), np.float32)
    extra[1:, :, :4] = np.eye(4, dtype=np.float32)[rng.integers(0, 4, (3, 401))]
    extra[1:, :, 4:] = rng.gamma(2.0, 0.03, (3, 401, 4)).astype(np.float32)
    x = np.concatenate([wins[:60], extra])
    inits, nodes = _mini_onnx(os.path.join(REF, "models", "CHG.onnx"))
    T = {k: torch.from_numpy(v) for k, v in inits.items()}
    h = torch.from_numpy(x)
    nconv = 0
    with torch.no_grad():
        for op, ins, at in nodes:
            if op == "Transpose":
                h = h.permute(*at["perm"])
            elif op == "BatchNormalization":
                h = F.batch_norm(h, T[ins[3]], T[ins[4]], T[ins[1]], T[ins[2]], False, 0.0, at.get("epsilon", 1e-5))
            elif op == "Conv":
                assert at["strides"] == [2] and at["pads"] == [1, 1] and at.get("dilations", [1]) == [1]
                h = F.conv1d(h, T[ins[1]], T[ins[2]], stride=2, padding=1)
                nconv += 1
            elif op == "Relu":
                h = F.relu(h)
            elif op in ("Flatten", "Reshape"):
                h = h.flatten(1)
            elif op == "Gemm":
                assert at.get("transB", 0) == 1
                h = F.linear(h, T[ins[1]], T[ins[2]])
            elif op in ("Constant", "Shape", "Gather", "Unsqueeze", "Concat"):
                continue  # shape bookkeeping of the export around Reshape
            else:
                raise SystemExit(f"CHG.onnx: unexpected op {op}")
    assert nconv == 8 and tuple(h.shape) == (len(x), 2)
    lg = h.numpy().astype(np.float32)
    # ... and EXECUTED BY THE REFERENCE'S OWN GRAPH (round 5): models/CpG.pt is the TorchScript export of the same architecture (first kernel
    # 11, BN folded), its 24 weight tensors are prim::Constant nodes of its forward graph -- CHG.onnx's tensors are put in their place
    # (batch_norm: weight, bias, mean, var; the eight conv1d; the two matmul / add pairs, Gemm's [out, in] weights transposed) and the
    # reference's graph is run on the same windows.  What CHG.onnx CONTAINS still comes through a parser of ours (two independent ones agree);
    # what is COMPUTED from it is now the reference's exported program, not a re-evaluation of ours.
    ts = torch.jit.load(os.path.join(REF, "models", "CpG.pt"), map_location="cpu")
    tconst = lambda n: [i.node() for i in n.inputs() if i.node().kind() == "prim::Constant" and i.type().kind() == "TensorType"]  # noqa: E731
    o_bn = [ins for op, ins, _ in nodes if op == "BatchNormalization"][0]
    o_conv = [ins for op, ins, _ in nodes if op == "Conv"]
    o_gemm = [ins for op, ins, _ in nodes if op == "Gemm"]
    ci = gi = 0
    for n in ts.graph.nodes():
        if n.kind() == "aten::batch_norm":
            for c, name in zip(tconst(n), (o_bn[1], o_bn[2], o_bn[3], o_bn[4])):
                c.t_("value", T[name].clone())
        elif n.kind() == "aten::conv1d":
            c = tconst(n)
            assert tuple(c[0].t("value").shape) == tuple(T[o_conv[ci][1]].shape)
            c[0].t_("value", T[o_conv[ci][1]].clone())
            c[1].t_("value", T[o_conv[ci][2]].clone())
            ci += 1
        elif n.kind() == "aten::matmul":
            tconst(n)[0].t_("value", T[o_gemm[gi][1]].t().contiguous().clone())
        elif n.kind() == "aten::add" and tconst(n):
            tconst(n)[0].t_("value", T[o_gemm[gi][2]].clone())
            gi += 1
    assert ci == 8 and gi == 2
    with torch.no_grad():
        lg_ts = ts(torch.from_numpy(x)).numpy().astype(np.float32)
    d_ts = float(np.abs(lg_ts - lg).max())
    assert d_ts < 1e-5, d_ts
    np.savez_compressed(os.path.join(GOLD, "cnn_CHG.npz"), windows=x, logits=lg, logits_ts=lg_ts)
    om = O.Model(os.path.join(ROOT, "hifimeth_amd", "weights", "CHG.hmw"))
    d = float(np.abs(om.logits(x) - lg).max())
    rp = os.path.join(GOLD, "cnn_report.json")
    report = json.load(open(rp)) if os.path.exists(rp) else {}
    report["CHG.onnx_torch_functional_over_independent_mini_parse_vs_oracle_max_dlogit"] = d
    report["CHG.onnx_weights_run_by_the_reference_TorchScript_graph_vs_torch_functional_max_dlogit"] = d_ts
    report["CHG.onnx_weights_run_by_the_reference_TorchScript_graph_vs_oracle_max_dlogit"] = float(np.abs(om.logits(x) - lg_ts).max())
    report["CHG_fixture_source"] = ("models/CHG.onnx evaluated (a) with torch functional ops over tools/make_golden.py:_mini_onnx (second, minimal ONNX "
                                    "parse) and (b) by the REFERENCE'S OWN TorchScript graph -- models/CpG.pt, the same architecture -- with CHG.onnx's "
                                    "tensors put in the place of its 24 constants (`logits_ts`); CHG.pt is a different checkpoint and was not used")
    json.dump(report, open(rp, "w"), indent=1)
    print(f"cnn_CHG.npz: {len(x)} windows, oracle(ONNX weights) vs torch-functional(CHG.onnx, mini parse) max |dlogit| = {d:.3g}; "
          f"the reference's TorchScript graph over CHG.onnx's weights vs torch-functional: {d_ts:.3g}")


def make_config_goldens():
    """BASELINE.json configs[0]/[1] stand-in (the P.patens tutorial BAM is an external download): a small CpG-only
    read set with the CPU path's outputs.  These vectors come from the ORACLE (which is pinned against the reference
    by the fixtures above), not from the reference directly -- they freeze the CPU path so that the GPU tests can
    be checked against committed numbers as well as against the live oracle."""
    reads = synth_reads(10, seed=20250220, gc=0.36, median_len=2400, sigma=0.35, frac_wide=0.2, frac_short=0.1,
                        frac_missing=0.1)
    models = [O.Model(os.path.join(ROOT, "hifimeth_amd", "weights", n + ".hmw")) for n in ("CpG", "CHG", "CHH")]
    out = dict(n_reads=len(reads))
    for mask, tag in ((1, "cpg"), (7, "all")):
        rid, qoff, strand, ctx, p, ml = [], [], [], [], [], []
        for i, rd in enumerate(reads):
            if not rd.has_kinetics() or rd.l_qseq < 1000:
                continue
            r = O.call_read(models, mask, rd)
            order = np.lexsort((r["qoff"], r["strand"]))
            rid += [i] * len(order)
            qoff.append(r["qoff"][order]); strand.append(r["strand"][order]); ctx.append(r["ctx"][order])
            p.append(r["p"][order]); ml.append(r["ml"][order])
        out.update({f"{tag}_read": np.array(rid, np.int32), f"{tag}_qoff": np.concatenate(qoff),
                    f"{tag}_strand": np.concatenate(strand), f"{tag}_ctx": np.concatenate(ctx),
                    f"{tag}_p": np.concatenate(p), f"{tag}_ml": np.concatenate(ml)})
        print(f"config golden ({tag}): {len(rid)} sites")
    np.savez_compressed(os.path.join(GOLD, "config1_calls.npz"), **out)


def make_align():
    """align.json: the reference's own BamMapInfo::init / cigar_to_alignment (src/corelib/bam_info.cpp:262-439) and
    extract_{cpg,chg,chh}_mapped_samples (src/corelib/5mc_motif_finder.cpp), built by oracle/ref_build into
    oracle/_ref/ref_align, run on synthetic alignments (=/X and M CIGARs, soft clips, both strands, an N op, a
    leading hard clip, an unmapped record)."""
    from hifimeth_amd.synth import AlignedRead, synth_alignments, synth_genome
    from oracle import pileup_oracle as P
    genome = synth_genome(n_chr=2, length=2500, seed=17)
    reads = synth_alignments(genome, 8, seed=23, median_len=600, err=0.03, eqx=True, frac_unmapped=0.13)
    reads += synth_alignments(genome, 6, seed=29, median_len=500, err=0.03, eqx=False, frac_unmapped=0)
    chrom = genome[0][1]
    reads.append(AlignedRead("skipN", 0, 0, 100, 60, [("H", 7), ("M", 50), ("N", 40), ("M", 60), ("P", 2), ("M", 5), ("S", 4)],
                             chrom[100:150] + chrom[190:250] + chrom[250:255] + "ACGT", None, None))
    fa = "/tmp/_golden_align.fa"
    with open(fa, "w") as f:
        f.write("! comment line\n")
        for n, sq in genome:
            f.write(f">{n} some description\n")
            for i in range(0, len(sq), 70):
                f.write(sq[i:i + 70].lower() if (i // 70) % 5 == 0 else sq[i:i + 70])
                f.write("\n")
    assert P.load_fasta(fa) == genome
    out = P.ref_align(fa, [(r.flag, max(r.tid, 0), max(r.pos, 0), r.cigar_string() if r.cigar else "1M", r.seq) for r in reads])
    recs = []
    for r, d in zip(reads, out):
        recs.append(dict(flag=r.flag, tid=r.tid, pos=r.pos, cigar=r.cigar_string(), seq=r.seq, ref=d))
    json.dump(dict(genome=genome, reads=recs), open(os.path.join(GOLD, "align.json"), "w"))
    os.remove(fa)
    print(f"align.json: {len(recs)} records, {sum(d is not None for d in out)} mapped, "
          f"{sum(len(d['chh']) for d in out if d)} CHH samples")


def make_helpers():
    """helpers.json: the reference's own `cov2bed` and `corr` subcommands (src/app/hifimeth/cov_to_bed.cpp,
    pileup_correlation.cpp), built by oracle/ref_build into oracle/_ref/ref_tools, run on a synthetic genome with
    Bismark-style coverage rows on every C and G (both strands, rows on the partner strand only, rows in non-motif
    positions, a chromosome that is visited twice) and on two BED files with partially shared loci."""
    import subprocess
    from hifimeth_amd.synth import synth_genome
    tools = os.path.join(ROOT, "oracle", "_ref", "ref_tools")
    genome = synth_genome(n_chr=3, length=1200, seed=41)
    rng = np.random.default_rng(43)
    fa = "/tmp/_golden_helpers.fa"
    with open(fa, "w") as f:
        for n, sq in genome:
            f.write(f">{n}\n")
            for i in range(0, len(sq), 60):
                f.write(sq[i:i + 60] + "\n")
    out = dict(genome=genome, cov2bed={}, corr=[])
    order = [0, 1, 2, 0]                                   # chromosome 0 comes back after the others
    for ctx in ("CpG", "CHG", "CHH"):
        rows = []
        for visit, ci in enumerate(order):
            name, sq = genome[ci]
            lo, hi = (0, len(sq) // 2) if (ci == 0 and visit == 0) else (len(sq) // 2, len(sq)) if ci == 0 else (0, len(sq))
            for i in range(max(lo, 3), min(hi, len(sq) - 3)):   # the reference reads neighbours without bounds checks
                if sq[i] in "CG" and rng.random() < 0.8 or rng.random() < 0.02:
                    pc, nc = int(rng.integers(0, 30)), int(rng.integers(0, 30))
                    if pc + nc == 0:
                        pc = 1
                    rows.append(f"{name}\t{i + 1}\t{i + 1}\t{100.0 * pc / (pc + nc):.6f}\t{pc}\t{nc}")
        cov, bed = f"/tmp/_golden_{ctx}.cov", f"/tmp/_golden_{ctx}.bed"
        open(cov, "w").write("\n".join(rows) + "\n")
        r = subprocess.run([tools, "cov2bed", fa, ctx.lower() if ctx == "CHG" else ctx, cov, bed], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        out["cov2bed"][ctx] = dict(cov=open(cov).read(), bed=open(bed).read(),
                                   stderr=[l for l in r.stderr.split("\n") if l.startswith("forward-strand-sites")][0])
        print(f"cov2bed {ctx}: {len(rows)} rows -> {out['cov2bed'][ctx]['bed'].count(chr(10))} loci; {out['cov2bed'][ctx]['stderr']}")
        os.remove(cov)
    # corr: the CpG BED above against a perturbed copy (some loci dropped, counts jittered, chromosome order swapped)
    a = out["cov2bed"]["CpG"]["bed"]
    lines = [l.split("\t") for l in a.strip().split("\n")]
    pert = []
    for l in lines:
        if rng.random() < 0.15:
            continue
        pc = max(0, int(l[4]) + int(rng.integers(-3, 4)))
        nc = max(0, int(l[5]) + int(rng.integers(-3, 4)))
        if pc + nc == 0:
            nc = 1
        pert.append((l[0], int(l[1]), pc, nc))
    pert.sort(key=lambda t: (-[n for n, _ in genome].index(t[0]), t[1]))
    b = "".join(f"{c}\t{s}\t{s + 1}\t{100.0 * pc / (pc + nc):g}\t{pc}\t{nc}\tCG\n" for c, s, pc, nc in pert)
    pa, pb = "/tmp/_golden_a.bed", "/tmp/_golden_b.bed"
    open(pa, "w").write(a)
    open(pb, "w").write(b)
    for args in ([], ["-c", "1"], ["-c", "20"], ["-c", "200"]):
        r = subprocess.run([tools, "corr"] + args + [pa, pb], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        corr = [l for l in r.stderr.split("\n") if l.startswith("correlation:")]
        out["corr"].append(dict(args=args, stdout=r.stdout, correlation=corr[0] if corr else None,
                                skipped="Skip computation" in r.stderr))
        print("corr", args, r.stdout.strip(), corr)
    out["corr_beds"] = [a, b]
    json.dump(out, open(os.path.join(GOLD, "helpers.json"), "w"))
    for f_ in (fa, pa, pb, "/tmp/_golden_CpG.bed", "/tmp/_golden_CHG.bed", "/tmp/_golden_CHH.bed"):
        os.remove(f_)


def make_pileup_thresholds():
    """pileup_thresholds.json: the reference's own s_resolve_scaled_prob_threshold (src/app/hifimeth/pileup.cpp:355-436;
    compiled in place into oracle/_ref/ref_pileup), run on histogram triples that cover its branches: bimodal (the valley is
    the threshold), too few samples (< 10 000 in the window -> 128), a window narrower than 50 bins, ties for the minimum
    (the first one wins), counts below 10 at the window's edges, empty histograms, mass only outside [20, 236)."""
    import subprocess
    rng = np.random.default_rng(20250220)
    cases = []

    def bimodal(n, lo_c, hi_c, w=18.0, floor=0):
        x = np.arange(256)
        h = n * (0.6 * np.exp(-0.5 * ((x - lo_c) / w) ** 2) + 0.4 * np.exp(-0.5 * ((x - hi_c) / w) ** 2)) + floor
        return rng.poisson(h).astype(np.int64)

    for _ in range(10):
        cases.append([bimodal(rng.integers(2000, 60000), rng.integers(5, 60), rng.integers(180, 250), rng.uniform(8, 40),
                              rng.integers(0, 30)) for _ in range(3)])
    flat = np.full(256, 100, np.int64)                      # every bin ties: the first bin of the window wins
    few = bimodal(300, 30, 220, 25.0, 12)                   # window wide enough, fewer than 10 000 samples -> 128
    narrow = np.zeros(256, np.int64); narrow[100:140] = 5000   # en - st < 50 -> 128
    edges = np.full(256, 400, np.int64); edges[:40] = 3; edges[200:] = 9; edges[120] = 17; edges[150] = 17  # trimmed edges, tie
    outside = np.zeros(256, np.int64); outside[:20] = 10 ** 6; outside[236:] = 10 ** 6    # nothing inside the window
    empty = np.zeros(256, np.int64)
    big = bimodal(4 * 10 ** 9, 20, 235, 30.0, 1000)         # counts beyond 32 bits
    step = np.concatenate([np.full(128, 900, np.int64), np.full(128, 50, np.int64)])
    cases += [[flat, few, narrow], [edges, outside, empty], [big, step, flat[::-1].copy()], [few, few, few]]
    for _ in range(6):  # random sparse / noisy histograms
        cases.append([rng.integers(0, rng.integers(2, 400), 256).astype(np.int64) * (rng.random(256) < rng.uniform(0.3, 1.0)) for _ in range(3)])
    txt = [str(len(cases))]
    for c in cases:
        for h in c:
            txt.append(" ".join(str(int(v)) for v in h))
    exe = os.path.join(ROOT, "oracle", "_ref", "ref_pileup")
    out = subprocess.run([exe], input="\n".join(txt) + "\n", capture_output=True, text=True, check=True).stdout.split()
    thr = [[int(out[3 * i + k]) for k in range(3)] for i in range(len(cases))]
    json.dump(dict(cases=[[[int(v) for v in h] for h in c] for c in cases], thresholds=thr),
              open(os.path.join(GOLD, "pileup_thresholds.json"), "w"))
    print(f"pileup_thresholds.json: {len(cases)} histogram triples; thresholds {sorted(set(t for r in thr for t in r))}")


def make_softmax():
    """softmax_ml.json: the reference's own s_logits_to_methy_probs (src/app/hifimeth/mod_batch.cpp:46-64; compiled in place
    into oracle/_ref/ref_softmax) on logit pairs: random, equal, far apart, and pairs placed on both sides of every
    255 * p = k boundary (where the truncation to the ML byte flips)."""
    import subprocess
    rng = np.random.default_rng(20250221)
    v0 = [rng.normal(0, 4, 1500).astype(np.float32)]
    v1 = [rng.normal(0, 4, 1500).astype(np.float32)]
    k = np.arange(1, 255, dtype=np.float64)
    for eps in (-3e-6, -3e-7, 0.0, 3e-7, 3e-6):          # p = k / 255 +- eps: logit difference log(p / (1 - p))
        pt = np.clip(k / 255.0 + eps, 1e-9, 1 - 1e-9)
        base = rng.normal(0, 3, len(k))
        v0.append(base.astype(np.float32))
        v1.append((base.astype(np.float32).astype(np.float64) + np.log(pt / (1 - pt))).astype(np.float32))
    ext = np.array([[0, 0], [1, 1], [-7.5, -7.5], [0, 50], [50, 0], [0, 100], [100, 0], [-80, 80], [3e4, -3e4], [0, 5.54], [0, 5.55],
                    [1e-30, -1e-30], [0, 16.7], [0, 88.0], [0, -88.0], [12.25, 12.25]], np.float32)
    v0.append(ext[:, 0]); v1.append(ext[:, 1])
    lg = np.stack([np.concatenate(v0), np.concatenate(v1)], 1).astype(np.float32)
    txt = str(len(lg)) + "\n" + "\n".join(f"{a:08x} {b:08x}" for a, b in lg.view(np.uint32))
    exe = os.path.join(ROOT, "oracle", "_ref", "ref_softmax")
    out = subprocess.run([exe], input=txt + "\n", capture_output=True, text=True, check=True).stdout.split()
    ml = [int(x) for x in out]
    assert len(ml) == len(lg)
    json.dump(dict(logits_hex=[f"{a:08x}{b:08x}" for a, b in lg.view(np.uint32)], ml=ml), open(os.path.join(GOLD, "softmax_ml.json"), "w"))
    print(f"softmax_ml.json: {len(ml)} logit pairs, bytes {min(ml)}..{max(ml)}, {len(set(ml))} distinct")


def make_codec():
    """codec.json: the reference's own codev1 decode table (BamKinetics ctor, src/corelib/bam_info.cpp:568-576) and lossy
    u16 -> u8 encoder s_encode_signal_value (:455-478), compiled in place into oracle/_ref/ref_codec."""
    import subprocess
    out = subprocess.run([os.path.join(ROOT, "oracle", "_ref", "ref_codec")], capture_output=True, text=True, check=True).stdout.splitlines()
    dec = [int(x) for x in out[0].split()]
    enc = [int(x) for x in out[1].split()]
    big = [int(x) for x in out[2].split()]
    assert len(dec) == 256 and len(enc) == 1200 and len(big) == 3
    json.dump(dict(decode=dec, encode_0_1199=enc, encode_big={"2000": big[0], "4095": big[1], "65535": big[2]}),
              open(os.path.join(GOLD, "codec.json"), "w"))
    print(f"codec.json: decode table 0..{dec[-1]}, encode(0..1199) -> 0..{max(enc)}")


def make_modparse():
    """modparse.json: the reference's own MM/ML parser core (s_parse_one_mod_list, src/corelib/bam_mod_parser.cpp:136-229;
    compiled in place into oracle/_ref/ref_modparse) on (flag, stored SEQ, MM, ML) records:
      * the MM/ML strings our WRITER rules produce for oracle calls on synthetic reads (both strands, a flag-16 read) -- the
        reference parser must give back exactly those calls (asserted here), which pins the writer's semantics to it;
      * the hand-worked known answers of modtags_known_answers.json;
      * other dialects the parser accepts: '?' / '.' flags, two codes per position, several lists, other bases."""
    import subprocess
    from oracle.modtags import expected_tags
    recs = []
    reads = synth_reads(6, seed=41, median_len=1500, sigma=0.2, frac_short=0, frac_missing=0, frac_wide=0.3)
    reads[1].flag = 16
    reads[4].flag = 16
    models = [O.Model(os.path.join(ROOT, "hifimeth_amd", "weights", c + ".hmw")) for c in ("CpG", "CHG", "CHH")]
    comp = bytes.maketrans(b"ACGTN", b"TGCAN")
    for rd in reads:
        c = O.call_read(models, 7, rd)
        order = np.lexsort((c["qoff"], c["strand"]))
        fwd = O.decode(rd)
        tags = expected_tags(fwd, c["qoff"][order], c["strand"][order], c["ml"][order])
        recs.append(dict(flag=int(rd.flag), seq=rd.ascii().decode(), mm=tags["MM"], ml=[int(v) for v in tags["ML"]],
                         calls=[[int(q), int(s), int(m)] for q, s, m in zip(c["qoff"][order], c["strand"][order], c["ml"][order])]))
    for k in json.load(open(os.path.join(GOLD, "modtags_known_answers.json")))["cases"]:
        recs.append(dict(flag=k["flag"], seq=k["stored_seq"], mm=k["MM"], ml=k["ML"], calls=[list(x) for x in k["calls"]]))
    s0 = reads[0].ascii().decode()
    # (no ChEBI case: the reference's numeric-code path cannot succeed -- s_chebi_to_iupac_code:44 compares the ',' behind the
    #  number with the list's LENGTH, and where that passes the caller asserts on the character behind the ',' it skipped
    #  (bam_mod_parser.cpp:183); our parsers accept ChEBI codes, a superset that no reference-written file exercises)
    recs += [dict(flag=0, seq=s0, mm="C+m?,0;C+h.,0;", ml=[9, 8]),
             dict(flag=0, seq=s0, mm="C+mh,2;", ml=[11, 12]), dict(flag=16, seq=s0, mm="G-m,3,0,0;C+m,1;", ml=[1, 2, 3, 4]),
             dict(flag=0, seq=s0, mm="A+a,0,5;T-a,2;", ml=[5, 6, 7])]
    txt = [str(len(recs))]
    for r in recs:
        txt.append(f"{r['flag']} {r['seq']} {r['mm']} {len(r['ml'])} " + " ".join(str(v) for v in r["ml"]))
    out = subprocess.run([os.path.join(ROOT, "oracle", "_ref", "ref_modparse")], input="\n".join(txt) + "\n", capture_output=True,
                         text=True, check=True).stdout.split("\n")
    li = 0
    for r in recs:
        n = int(out[li]); li += 1
        mods = []
        for _ in range(n):
            q, st, ub, code, pr = out[li].split(); li += 1
            mods.append([int(q), int(st), ub, code, int(pr)])
        r["mods"] = mods
        if "calls" in r:  # the writer's strings mean exactly the calls they were written from, by the reference's own parser
            got = sorted((m[0], m[1], m[4]) for m in mods)
            assert got == sorted(tuple(x) for x in r["calls"]), "writer / reference parser disagree"
    json.dump(dict(records=recs), open(os.path.join(GOLD, "modparse.json"), "w"))
    print(f"modparse.json: {len(recs)} records, {sum(len(r['mods']) for r in recs)} mods parsed by the reference")


if __name__ == "__main__":
    if not O.ref_scan_available():
        raise SystemExit("build oracle/_ref first: make -C oracle")
    if len(sys.argv) > 2 and sys.argv[2] == "align":      # only the pileup fixtures
        make_align()
        raise SystemExit(0)
    if len(sys.argv) > 2 and sys.argv[2] == "helpers":    # only the cov2bed / corr fixtures
        make_helpers()
        raise SystemExit(0)
    if len(sys.argv) > 2 and sys.argv[2] == "modparse":    # only the MM/ML parser fixture
        make_modparse()
        raise SystemExit(0)
    if len(sys.argv) > 2 and sys.argv[2] == "codec":       # only the kinetics codec fixture
        make_codec()
        raise SystemExit(0)
    if len(sys.argv) > 2 and sys.argv[2] == "softmax":     # only the logits -> ML byte fixture
        make_softmax()
        raise SystemExit(0)
    if len(sys.argv) > 2 and sys.argv[2] == "thresholds":  # only the pileup threshold fixture
        make_pileup_thresholds()
        raise SystemExit(0)
    if len(sys.argv) > 2 and sys.argv[2] == "windows":    # only the window fixture (the CNN fixtures keep the windows they were made from)
        make_windows()
        raise SystemExit(0)
    if len(sys.argv) > 2 and sys.argv[2] == "chg":        # only the CHG CNN fixture (windows are those of cnn_CpG.npz)
        make_cnn_chg(np.load(os.path.join(GOLD, "cnn_CpG.npz"))["windows"])
        raise SystemExit(0)
    make_scan()
    w = make_windows()
    make_cnn(w)
    make_cnn_chg(w)
    make_config_goldens()
    make_align()
    make_helpers()
    make_pileup_thresholds()
    make_softmax()
    make_codec()
    make_modparse()
