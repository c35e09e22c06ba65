"""CPU restatement of `hifimeth pileup` (TEST INFRASTRUCTURE ONLY -- never imported by the product path).

Pure-Python loops, sized for test inputs.  Each function cites the reference lines it follows.

Pin status: `cigar_to_alignment`, `map_info` and `chh_mapped_samples` are checked against the reference's own code
(oracle/ref_build builds BamMapInfo::init + 5mc_motif_finder.cpp into oracle/_ref/ref_align; fixtures in
tests/golden/align.json); `resolve_threshold` against the reference's own s_resolve_scaled_prob_threshold (pileup.cpp is
compiled in place into oracle/_ref/ref_pileup; fixtures in tests/golden/pileup_thresholds.json); `parse_mods` against the
reference's own parser core s_parse_one_mod_list (bam_mod_parser.cpp compiled in place into oracle/_ref/ref_modparse;
fixtures in tests/golden/modparse.json).  The record/count/BED stages of pileup.cpp are inline in a function that calls the
htslib library, which is not in this image, so those stages are a restatement by reading: PARITY UNPINNED.
`cov_to_bed` is pinned against the reference's whole `cov2bed` subcommand (oracle/_ref/ref_tools, tests/golden/helpers.json).

Where the reference's output depends on thread timing (two motif classes landing on one locus, see `count`), the
restatement fixes the order to "BAM order, classes CpG < CHG < CHH within a read", i.e. what one reference thread
with a stable sort would produce.
"""
import numpy as np

GAP = "-"
_DEC = {1: "A", 2: "C", 4: "G", 8: "T", 15: "N"}           # s_decode_bam_query_base, bam_info.cpp:100-121
_COMP = {"A": "T", "C": "G", "G": "C", "T": "A", "N": "N"}  # completement_residue, bam_info.cpp:146-167
CIGAR_OPS = "MIDNSHP=XB"

FWD_CHG = ("CAG", "CCG", "CTG")                              # 5mc_context.cpp:4-5
FWD_CHH = tuple(a + b + c for a in "C" for b in "ACT" for c in "ACT")   # 5mc_context.cpp:9
REV_CHH = tuple(a + b + c for a in "AGT" for b in "AGT" for c in "G")   # 5mc_context.cpp:10


# ---- FASTA ------------------------------------------------------------------------------------------------------
def load_fasta(path):
    """HbnDatabase::HbnDatabase (hbn_seqdb.cpp:37-95): -> [(name, SEQUENCE)] upper-cased.
    A line is a header when it starts with '>' OR holds a digit or '|' among its first 33 characters
    (s_IsSeqID, hbn_seqdb.cpp:7-16); lines starting with ! # ; are comments; the name ends at the first blank."""
    import gzip
    opener = gzip.open if open(path, "rb").read(2) == b"\x1f\x8b" else open
    seqs, name, parts = [], None, []
    with opener(path, "rt") as f:
        for raw in f:
            line = raw.strip()
            if not line or line[0] in "!#;":
                continue
            is_id = any(ch in "0123456789|" for ch in line[:33])
            if is_id or line[0] == ">":
                if name:
                    seqs.append((name, "".join(parts).upper()))
                s = line[1:] if line[0] == ">" else line
                name, parts = s.split()[0] if s.split() else "", []
            else:
                parts.append(line)
    if name:
        seqs.append((name, "".join(parts).upper()))
    return seqs


# ---- read sequence ------------------------------------------------------------------------------------------------
def stored_seq(seq4, l_qseq):
    s4 = np.asarray(seq4, np.uint8)
    codes = np.empty(2 * len(s4), np.uint8)
    codes[0::2], codes[1::2] = s4 >> 4, s4 & 15
    return "".join(_DEC[int(c)] for c in codes[:l_qseq])


def fwd_rev(stored, flag):
    """BamQuerySequence::init (bam_info.cpp:169-222): (fwd_rqs, rev_rqs) from SEQ as stored."""
    rc = "".join(_COMP[c] for c in reversed(stored))
    return (rc, stored) if flag & 16 else (stored, rc)


# ---- MM / ML ---------------------------------------------------------------------------------------------------
_CHEBI = {27551: "m", 76792: "h", 76794: "f", 76793: "c", 16964: "g", 80961: "e", 17477: "b", 28871: "a",
          44605: "o", 18107: "n"}


def parse_mods(fwd, mm, ml):
    """extract_bam_base_mods (bam_mod_parser.cpp:231-286) -> [(qoff, strand, unmod_base, code, prob)];
    `fwd` is the forward-strand sequence (the MM deltas count bases of it, bam_mod_parser.cpp:197-228)."""
    mods = []
    if ml is None or len(ml) == 0 or mm is None:
        return mods
    assert mm.endswith(";")
    pi = 0
    for series in mm[:-1].split(";"):
        series += ";"
        assert len(series) >= 4
        ub, strand = series[0], 0 if series[1] == "+" else 1
        assert ub in "CGTAUN" and series[1] in "+-"
        si = 2
        if series[2].isdigit():
            while series[si].isdigit():
                si += 1
            codes = _CHEBI[int(series[2:si])]
        else:
            codes = ""
            while series[si] not in ",;":
                if series[si] not in ".?":
                    codes += series[si]
                si += 1
        deltas = [int(x) for x in series[si:-1].split(",") if x != ""]
        q = 0
        for d in deltas:
            cnt = 0
            while cnt < d:
                if fwd[q] == ub:
                    cnt += 1
                q += 1
            while fwd[q] != ub:
                q += 1
            for code in codes:
                mods.append((q, strand, ub, code, int(ml[pi])))
                pi += 1
            q += 1
    return mods


def mod_context(fwd, q):
    """context of a call for the histograms (pileup.cpp:237-272): 0 CpG, 1 CHG, 2 CHH, -1 none"""
    L = len(fwd)
    if fwd[q] == "C":
        if q + 1 < L and fwd[q + 1] == "G":
            return 0
        if q + 2 < L and fwd[q:q + 3] in FWD_CHG:
            return 1
        if q + 2 < L and fwd[q:q + 3] in FWD_CHH:
            return 2
        return -1
    if q - 2 >= 0 and fwd[q - 2:q + 1] in REV_CHH:
        return 2
    return -1


# ---- alignment ---------------------------------------------------------------------------------------------------
def cigar_to_alignment(query, subject, cigar):
    """cigar_to_alignment (bam_info.cpp:262-371).  `subject` starts at the alignment's POS; cigar = [(op_char, len)].
    Only cigar[0] is examined for the leading clip; S/H/P elsewhere add nothing (and do not advance the query)."""
    qas, sas, qpos, spos = [], [], [], []
    opi, qb = 0, 0
    if cigar and cigar[0][0] == "S":
        qb, opi = cigar[0][1], 1
    elif cigar and cigar[0][0] == "H":
        opi = 1
    qi, si = qb - 1, -1
    for op, num in cigar[opi:]:
        if op in "M=X":
            for _ in range(num):
                qi += 1; si += 1
                qas.append(query[qi]); sas.append(subject[si]); qpos.append(qi); spos.append(si)
        elif op == "I":
            for _ in range(num):
                qi += 1
                qas.append(query[qi]); sas.append(GAP); qpos.append(qi); spos.append(si)
        elif op in "DN":
            for _ in range(num):
                si += 1
                qas.append(GAP); sas.append(subject[si]); qpos.append(qi); spos.append(si)
        elif op in "SHP":
            continue
        else:
            raise ValueError(f"Unrecognised CIGAR operation '{op}'")
    return dict(qas="".join(qas), sas="".join(sas), qpos=qpos, spos=spos, qb=qb, qe=qi, sb=0, se=si)


def map_info(flag, pos, cigar, stored, chr_seq):
    """BamMapInfo::init (bam_info.cpp:373-439): None for unmapped records; the query is SEQ as stored for both
    strands (fwd_rqs when forward, rev_rqs when reverse)."""
    if flag & 4:
        return None
    a = cigar_to_alignment(stored, chr_seq[pos:], cigar)
    a["spos"] = [p + pos for p in a["spos"]]
    a["sb"] += pos
    a["se"] += pos + 1
    a["qe"] += 1
    n = len(a["qas"])
    a["as_size"] = n
    a["qdir"] = 1 if flag & 16 else 0
    a["pi"] = 0.0 if n == 0 else 100.0 * sum(1 for x, y in zip(a["qas"], a["sas"]) if x == y) / n   # :11-23
    return a


def cpg_records(a, L):
    """pileup.cpp:292-304 -> [(qoff, soff)] (REV: the C of the read's own strand, recorded at the reference C)"""
    out = []
    qas, sas = a["qas"], a["sas"]
    for i in range(a["as_size"] - 1):
        if qas[i:i + 2] != "CG" or sas[i:i + 2] != "CG":
            continue
        out.append((a["qpos"][i] if a["qdir"] == 0 else L - 1 - (a["qpos"][i] + 1), a["spos"][i]))
    return out


def chg_records(a, L):
    """pileup.cpp:306-335: forward reads CCG/CAG/CTG, reverse reads CGG/CAG/CTG; soff is always the motif start"""
    out = []
    qas, sas = a["qas"], a["sas"]
    pats = ("CCG", "CAG", "CTG") if a["qdir"] == 0 else ("CGG", "CAG", "CTG")
    for i in range(a["as_size"] - 2):
        if qas[i:i + 3] in pats and sas[i:i + 3] == qas[i:i + 3]:
            out.append((a["qpos"][i] if a["qdir"] == 0 else L - 1 - (a["qpos"][i] + 2), a["spos"][i]))
    return out


def chh_mapped_samples(a, L):
    """extract_chh_mapped_samples (5mc_motif_finder.cpp:104-144): forward motifs first, then reverse motifs"""
    out = []
    qas, sas = a["qas"], a["sas"]
    for i in range(a["as_size"] - 2):
        if qas[i:i + 3] in FWD_CHH and sas[i:i + 3] == qas[i:i + 3]:
            out.append((a["qpos"][i] if a["qdir"] == 0 else L - 1 - a["qpos"][i], a["spos"][i]))
    for i in range(a["as_size"] - 2):
        if qas[i:i + 3] in REV_CHH and sas[i:i + 3] == qas[i:i + 3]:
            out.append((a["qpos"][i] + 2 if a["qdir"] == 0 else L - 1 - (a["qpos"][i] + 2), a["spos"][i] + 2))
    return out


# ---- the pileup --------------------------------------------------------------------------------------------------
def resolve_threshold(bins):
    """s_resolve_scaled_prob_threshold for one context (pileup.cpp:355-436) -> (threshold, samples_in_window)"""
    a = [int(x) for x in bins]
    st, en = 20, 256 - 20
    while st < 256 and a[st] < 10:
        st += 1
    while en and a[en - 1] < 10:
        en -= 1
    total, min_i, min_cnt = 0, -1, None
    if en - st >= 50:
        for i in range(st, en):
            total += a[i]
            if min_cnt is None or min_cnt > a[i]:
                min_cnt, min_i = a[i], i
    return (128 if total < 10000 or min_i == -1 else min_i), total


def read_contribution(rec, chrs, min_mapq=0, min_pi=0.0):
    """One iteration of s_genomic_methy_freq_thread (pileup.cpp:230-347).
    rec: dict(flag, tid, pos, mapq, cigar=[(op,len)], seq (stored), mm, ml);  chrs: [(name, seq)] indexed by tid.
    -> (hist[3][256] increments as a list of (ctx, prob), records [(sid, soff, prob, motif)])"""
    stored = rec["seq"]
    L = len(stored)
    fwd, _ = fwd_rev(stored, rec["flag"])
    mods = parse_mods(fwd, rec.get("mm"), rec.get("ml"))
    if not mods:
        return [], []
    a = map_info(rec["flag"], rec["pos"], rec["cigar"], stored, chrs[rec["tid"]][1]) if not rec["flag"] & 4 else None
    if a is None:
        return [], []
    hist = []
    if not rec["flag"] & 0x900:
        for q, _s, ub, _code, prob in mods:
            if ub not in "CG":
                continue
            c = mod_context(fwd, q)
            if c >= 0:
                hist.append((c, prob))
    if rec["mapq"] < min_mapq or a["pi"] < min_pi:
        return hist, []
    read_mods = {}
    for q, _s, _ub, code, prob in mods:
        if code == "m":
            read_mods[q] = prob
    recs = []
    for motif, pairs in ((0, cpg_records(a, L)), (1, chg_records(a, L)), (2, chh_mapped_samples(a, L))):
        for qoff, soff in pairs:
            if qoff in read_mods:
                recs.append((rec["tid"], soff, read_mods[qoff], motif))
    return hist, recs


def pileup(records, chrs, min_mapq=0, min_pi=0.0, thresholds=None):
    """s_compute_methy_freq (pileup.cpp:461-606) -> dict(bins, thresholds, loci, bed)
    loci: sorted [(sid, soff, pcov, ncov, motif)]; bed: {"CpG"|"CHG"|"CHH": text of <prefix>.<ctx>.cov.bed}.
    The motif of a locus hit by two classes is that of its last record in BAM order (CpG < CHG < CHH within a read)."""
    bins = np.zeros((3, 256), np.uint64)
    allrec = []
    for rec in records:
        hist, recs = read_contribution(rec, chrs, min_mapq, min_pi)
        for c, p in hist:
            bins[c, p] += 1
        allrec += recs
    thr = list(thresholds) if thresholds is not None else [resolve_threshold(bins[c])[0] for c in range(3)]
    cov = {}
    for sid, soff, prob, motif in allrec:
        e = cov.setdefault((sid, soff), [0, 0, motif])
        e[2] = motif                                  # base_motifs[soff] = motif (pileup.cpp:533-552)
        e[0 if prob >= thr[motif] else 1] += 1
    loci = sorted((sid, soff, e[0], e[1], e[2]) for (sid, soff), e in cov.items())
    bed = bed_text([(chrs[sid][0], soff, p, n, motif) for sid, soff, p, n, motif in loci])
    return dict(bins=bins, thresholds=thr, records=allrec, loci=loci, bed=bed)


def bed_text(loci):
    """(chromosome name, offset, pcov, ncov, motif) in output order -> the text of the three *.cov.bed files (pileup.cpp:562-590:
    chr TAB k TAB k+1 TAB freq TAB pcov TAB ncov, freq = 100.0 * pcov / cov through an ostringstream = %g).  Pinned by the
    literal rows of tests/golden/literal_pins.json."""
    bed = {"CpG": [], "CHG": [], "CHH": []}
    for name, soff, p, n, motif in loci:
        bed[("CpG", "CHG", "CHH")[motif]].append("%s\t%d\t%d\t%g\t%d\t%d\n" % (name, soff, soff + 1, 100.0 * p / (p + n), p, n))
    return {k: "".join(v) for k, v in bed.items()}


# ---- helpers: cov2bed (src/app/hifimeth/cov_to_bed.cpp) ---------------------------------------------------------
def cov_to_bed(chrs, context, cov_text):
    """`hifimeth cov2bed reference context bismark-call bed` -> (bed text, forward sites, reverse sites).
    Pinned against the reference's own subcommand (oracle/_ref/ref_tools; fixtures in tests/golden/helpers.json).
    CpG  : cov_to_bed.cpp:36-142   C+G starts a locus, a row on the G of CG adds to the locus one base left
    CHG  : cov_to_bed.cpp:144-294  CCG / CAG / CTG start a locus; a G closing CGG stays in place (named CCG); a G
                                   closing CAG / CTG adds to the locus two bases left
    CHH  : cov_to_bed.cpp:296-394  9 forward motifs on C; a G closing one of the 9 reverse motifs stays in place and
                                   is named by the forward spelling of the same index (5mc_context.cpp:9-10)
    A chromosome's loci are written out when the input moves to another chromosome (:59-67)."""
    names = [n for n, _ in chrs]
    ctx = context.lower()
    out, fs, rs = [], 0, 0
    cur, loci = None, None

    def dump():
        if cur is None:
            return
        for i in sorted(loci):
            pc, nc, motif = loci[i]
            assert pc + nc > 0
            out.append("%s\t%d\t%d\t%g\t%d\t%d\t%s\n" % (names[cur], i, i + 1, 100.0 * pc / (pc + nc), pc, nc, motif))

    for line in cov_text.split("\n"):
        if not line:
            continue
        col = line.split("\t")
        sid = names.index(col[0])
        if sid != cur:
            dump()
            cur, loci = sid, {}
        sq = chrs[sid][1]
        soff, pc, nc = int(col[1]) - 1, int(col[4]), int(col[5])
        assert int(col[2]) - 1 == soff
        b = sq[soff]
        nxt, prv = sq[soff:soff + 3], sq[max(soff - 2, 0):soff + 1]
        if ctx == "cpg":
            if b == "C" and nxt[:2] == "CG":
                loci[soff] = [pc, nc, "CG"]; fs += 1
            if b == "G" and prv[-2:] == "CG":
                e = loci.setdefault(soff - 1, [0, 0, "CG"]); e[0] += pc; e[1] += nc; e[2] = "CG"; rs += 1
        elif ctx == "chg":
            if b == "C" and nxt in FWD_CHG:
                loci[soff] = [pc, nc, nxt]; fs += 1
            if b == "G" and prv == "CGG":
                loci[soff] = [pc, nc, "CCG"]; rs += 1
            if b == "G" and prv in ("CAG", "CTG"):
                e = loci.setdefault(soff - 2, [0, 0, prv]); e[0] += pc; e[1] += nc; rs += 1
        elif ctx == "chh":
            if b == "C" and nxt in FWD_CHH:
                loci[soff] = [pc, nc, nxt]; fs += 1
            elif b == "G" and prv in REV_CHH:
                loci[soff] = [pc, nc, "".join(_COMP[c] for c in reversed(prv))]; rs += 1   # kRev[i] = revcomp(kFwd[i])
        else:
            raise ValueError("Illegal 5mc context: " + context)
    dump()
    return "".join(out), fs, rs


# ---- helpers: eval (src/app/hifimeth/eval.cpp) -- restated by reading, PARITY UNPINNED (htslib-bound, and its output
# files are drawn with random_device seeds); only the deterministic part is restated: thresholds and sample counts
def bismark_labels(chrs, bed_text):
    """s_fill_chr_base_label_with_bismark (eval.cpp:42-114): {(sid, soff): 0 | 1} from 0-based BED rows
    chr, start, end, freq, pcov, ncov -- rows with < 10 reads or mixed calls carry no label"""
    names = [n for n, _ in chrs]
    lab = {}
    for line in bed_text.split("\n"):
        if not line:
            continue
        col = line.split("\t")
        sid, soff, send, pc, nc = names.index(col[0]), int(col[1]), int(col[2]), int(col[4]), int(col[5])
        assert send - soff == 1
        if pc + nc < 10:
            continue
        if pc == 0:
            lab[(sid, soff)] = 0
        elif nc == 0:
            lab[(sid, soff)] = 1
    return lab


def eval_counts(records, chrs, labels):
    """-> (bins[3][256] of s_prob_bin_thread (eval.cpp:153-211: every primary record with calls, mapped or not),
           thresholds, counts[3][2][256] of s_read_level_sample_thread (eval.cpp:469-560) before the CHH thinning)"""
    bins = np.zeros((3, 256), np.uint64)
    cnt = np.zeros((3, 2, 256), np.uint64)
    for rec in records:
        fwd, _ = fwd_rev(rec["seq"], rec["flag"])
        mods = parse_mods(fwd, rec.get("mm"), rec.get("ml"))
        if not mods:
            continue
        if not rec["flag"] & 0x900:
            for q, _s, _ub, _code, prob in mods:
                c = mod_context(fwd, q)
                if c >= 0:
                    bins[c, prob] += 1
        if rec["flag"] & 4:
            continue
        _h, recs = read_contribution(rec, chrs)           # no mapQ / identity filter in eval (:484-489)
        for sid, soff, prob, motif in recs:
            lab = labels.get((sid, soff))
            if lab is not None:
                cnt[motif, lab, prob] += 1
    return bins, [resolve_threshold(bins[c])[0] for c in range(3)], cnt


# ---- the reference's own alignment code (oracle/_ref/ref_align), used to pin the functions above ---------------
def ref_align_available():
    import os
    return os.path.exists(os.path.join(os.path.dirname(os.path.abspath(__file__)), "_ref", "ref_align"))


def ref_align(fasta_path, recs):
    """recs: [(flag, tid, pos, cigar_string, stored_seq)] -> per record None (unmapped) or dict"""
    import os
    import subprocess
    exe = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_ref", "ref_align")
    inp = "".join(f"{f} {t} {p} {c} {s}\n" for f, t, p, c, s in recs)
    out = subprocess.run([exe, fasta_path], input=inp.encode(), capture_output=True, check=True).stdout.decode()
    lines = out.split("\n")
    res, i = [], 0
    while i < len(lines) and lines[i]:
        if lines[i] == "unmapped":
            res.append(None)
            i += 1
            continue
        h = lines[i].split()
        d = dict(qdir=int(h[1]), qb=int(h[2]), qe=int(h[3]), sid=int(h[4]), sb=int(h[5]), se=int(h[6]),
                 as_size=int(h[7]), pi=float(h[8]))
        d["qas"], d["sas"] = lines[i + 1][4:], lines[i + 2][4:]
        d["qpos"] = [int(x) for x in lines[i + 3].split()[1:]]
        d["spos"] = [int(x) for x in lines[i + 4].split()[1:]]
        for k, tag in enumerate(("cpg", "chg", "chh")):
            t = lines[i + 5 + k].split()
            assert t[0] == tag
            d[tag] = [tuple(int(v) for v in x.split(":")) for x in t[2:]]
        res.append(d)
        i += 8
    return res


def parse_cigar(s):
    if s == "*":
        return []
    out, n = [], ""
    for ch in s:
        if ch.isdigit():
            n += ch
        else:
            out.append((ch, int(n)))
            n = ""
    return out
