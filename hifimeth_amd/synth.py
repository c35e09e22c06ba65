"""Synthetic PacBio HiFi reads with kinetics (stand-in for the unavailable tutorial / 30x BAMs).

Follows SURVEY.md section 8(d): i.i.d. genome with GC fraction g, reads with log-normal length
(median 15 kb, clipped to [1 kb, 30 kb]), random strand, unaligned (flag 4), per-base codev1
kinetics bytes `fi,fp,ri,rp` (IPD ~ round(Gamma(2,12)), PW ~ round(Gamma(3,5)), clipped to 0..255),
a small fraction of reads with `B:S` u16 frame arrays, a few short reads and reads with a missing tag
(the pass-through paths of reference src/app/hifimeth/mod_main.cpp:189-196).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List, Optional

import numpy as np

_NIB = np.array([1, 2, 4, 8, 15], np.uint8)  # A C G T N  (BAM seq_nt16 codes)
_ASCII = np.frombuffer(b"ACGTN", np.uint8)


@dataclass
class Read:
    """One unaligned read as the BAM record holds it."""
    name: str
    l_qseq: int
    flag: int
    seq4: np.ndarray                 # uint8[(L+1)//2], high nibble first
    fi: Optional[np.ndarray]         # uint8[L] (B:C) or uint16[L] (B:S); None = tag missing
    fp: Optional[np.ndarray]
    ri: Optional[np.ndarray]
    rp: Optional[np.ndarray]

    def has_kinetics(self) -> bool:
        return all(x is not None and len(x) == self.l_qseq for x in (self.fi, self.fp, self.ri, self.rp))

    def ascii(self) -> bytes:
        """Sequence as stored (not strand-normalised)."""
        return _ASCII[unpack_codes(self.seq4, self.l_qseq)].tobytes()


def pack_codes(codes: np.ndarray) -> np.ndarray:
    """codes 0..4 (A,C,G,T,N) -> BAM 4-bit packed."""
    nib = _NIB[codes]
    if len(nib) & 1:
        nib = np.concatenate([nib, np.zeros(1, np.uint8)])
    return ((nib[0::2] << 4) | nib[1::2]).astype(np.uint8)


def unpack_codes(seq4: np.ndarray, L: int) -> np.ndarray:
    nib = np.empty(2 * len(seq4), np.uint8)
    nib[0::2] = seq4 >> 4
    nib[1::2] = seq4 & 15
    lut = np.full(16, 4, np.uint8)
    lut[[1, 2, 4, 8]] = [0, 1, 2, 3]
    return lut[nib[:L]]


def read_from_ascii(seq: bytes, fi, fp, ri, rp, flag: int = 4, name: str = "r") -> Read:
    lut = np.full(256, 4, np.uint8)
    lut[[65, 67, 71, 84]] = [0, 1, 2, 3]
    codes = lut[np.frombuffer(seq, np.uint8)]
    return Read(name, len(seq), flag, pack_codes(codes), fi, fp, ri, rp)


def _kinetics(rng, L: int, wide: bool):
    if not wide:
        return [np.clip(np.rint(rng.gamma(k, th, L)), 0, 255).astype(np.uint8)
                for k, th in ((2.0, 12.0), (3.0, 5.0), (2.0, 12.0), (3.0, 5.0))]
    # B:S arrays hold raw frame counts (may exceed 952; re-encoded lossily, bam_info.cpp:455-478)
    return [np.clip(np.rint(rng.gamma(2.0, s, L)), 0, 2000).astype(np.uint16) for s in (30.0, 12.0, 30.0, 12.0)]


def synth_reads(n_reads: int, seed: int = 20250220, gc: float = 0.36, median_len: int = 15000,
                sigma: float = 0.35, min_len: int = 1000, max_len: int = 30000, frac_wide: float = 0.01,
                frac_short: float = 0.005, frac_missing: float = 0.001, frac_n: float = 0.0,
                genome_len: int = 4_000_000) -> List[Read]:
    rng = np.random.default_rng(seed)
    p = np.array([(1 - gc) / 2, gc / 2, gc / 2, (1 - gc) / 2])
    genome = rng.choice(4, size=genome_len, p=p).astype(np.uint8)
    out = []
    for i in range(n_reads):
        L = int(np.clip(rng.lognormal(np.log(median_len), sigma), min_len, max_len))
        if rng.random() < frac_short:
            L = int(rng.integers(50, min_len))
        L = min(L, genome_len)
        st = int(rng.integers(0, genome_len - L + 1))
        codes = genome[st:st + L].copy()
        if rng.random() < 0.5:  # read from the reverse strand, still stored as-is (unaligned, flag 4)
            codes = (3 - codes)[::-1].copy()
        if frac_n > 0:
            codes[rng.random(L) < frac_n] = 4
        fi, fp, ri, rp = _kinetics(rng, L, rng.random() < frac_wide)
        if rng.random() < frac_missing:
            rp = None
        out.append(Read(f"m0/{i}/ccs", L, 4, pack_codes(codes), fi, fp, ri, rp))
    return out


def write_unaligned_bam(path: str, reads, level: int = 1, threads: int = 8,
                        header_text: str = "@HD\tVN:1.6\tSO:unknown\tpb:5.0.0\n") -> int:
    """PacBio-style unaligned BAM of `reads` (fi / fp / ri / rp as B:C or B:S, plus np / rq / RG / zm), BGZF blocks deflated
    on `threads` threads (zlib releases the GIL) -- the same records tests/bamutil.reads_to_bam writes, fast enough for the
    GB-sized inputs of the end-to-end benchmark.  Returns the number of payload bytes."""
    import struct
    import zlib
    from concurrent.futures import ThreadPoolExecutor

    def block(data: bytes) -> bytes:
        co = zlib.compressobj(level, zlib.DEFLATED, -15)
        comp = co.compress(data) + co.flush()
        return (b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff\x06\x00BC\x02\x00" + struct.pack("<H", 18 + len(comp) + 8 - 1) + comp
                + struct.pack("<II", zlib.crc32(data) & 0xffffffff, len(data)))

    def aux_b(tag: bytes, a) -> bytes:
        a = np.ascontiguousarray(a)
        return tag + (b"BC" if a.dtype.itemsize == 1 else b"BS") + struct.pack("<I", len(a)) + a.astype(a.dtype.newbyteorder("<")).tobytes()

    BLK = 0xff00
    total = 0
    with open(path, "wb") as f, ThreadPoolExecutor(max_workers=max(1, threads)) as ex:
        buf = bytearray(b"BAM\1" + struct.pack("<I", len(header_text)) + header_text.encode() + struct.pack("<I", 0))

        def drain(final: bool):
            nonlocal buf, total
            n = len(buf) if final else len(buf) // BLK * BLK
            if n == 0:
                return
            mv = bytes(buf[:n])
            for comp in ex.map(block, (mv[i:i + BLK] for i in range(0, n, BLK))):
                f.write(comp)
            total += n
            del buf[:n]

        for i, r in enumerate(reads):
            aux = b"npi" + struct.pack("<i", 10 + i) + b"rqf" + struct.pack("<f", 0.999) + b"RGZrg0\0"
            for tag in ("fi", "fp", "ri", "rp"):
                a = getattr(r, tag)
                if a is not None:
                    aux += aux_b(tag.encode(), a)
            aux += b"zmi" + struct.pack("<i", i)
            qn = r.name.encode() + b"\0"
            core = struct.pack("<iiBBHHHiiii", -1, -1, len(qn), 255, 4680, 0, r.flag, r.l_qseq, -1, -1, 0)
            body_len = len(core) + len(qn) + len(r.seq4) + r.l_qseq + len(aux)
            buf += struct.pack("<I", body_len) + core + qn + bytes(r.seq4) + b"\xff" * r.l_qseq + aux
            if len(buf) >= (64 << 20):
                drain(False)
        drain(True)
        f.write(bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000"))
    return total


_QTAB = {}


def _quantile_table(k: float, theta: float, hi: int, dtype):
    """65536-entry quantile table of clip(rint(Gamma(k, theta)), 0, hi): one table lookup per value turns
    16 random bits into a draw of the same distribution (2^-16 resolution) -- ~20x faster than rng.gamma."""
    key = (k, theta, hi)
    if key not in _QTAB:
        rng = np.random.default_rng(977)
        v = np.sort(np.clip(np.rint(rng.gamma(k, theta, 1 << 20)), 0, hi))
        _QTAB[key] = v[8::16].astype(dtype)
    return _QTAB[key]


def synth_slab(n_reads: int, seed: int = 20250220, gc: float = 0.36, median_len: int = 15000, sigma: float = 0.35,
               min_len: int = 1000, max_len: int = 30000, frac_wide: float = 0.01, frac_short: float = 0.005,
               frac_missing: float = 0.001, genome_len: int = 4_000_000, cpg_oe: float = 1.0) -> List[Read]:
    """Same statistics as synth_reads, generated slab-at-a-time for the streaming benchmark: read
    placement, strands and the kinetics of ALL reads come from a few vectorised RNG calls; the reads are views.
    cpg_oe < 1 depletes CpG the way vertebrate genomes are (observed / expected CpG ~ 0.2 - 0.25 in human): that fraction of the
    i.i.d. genome's CG dinucleotides stays, the others become TG or CA (the deamination products) -- at GC 0.41 and cpg_oe 0.24
    CpG sites are ~1 % of the bases, below the density at which a context takes the dense trunk (hm_engine.cpp)."""
    rng = np.random.default_rng(seed)
    p = np.array([(1 - gc) / 2, gc / 2, gc / 2, (1 - gc) / 2])
    genome = rng.choice(4, size=genome_len, p=p).astype(np.uint8)
    if cpg_oe < 1.0:
        cg = np.nonzero((genome[:-1] == 1) & (genome[1:] == 2))[0]
        kill = cg[rng.random(len(cg)) >= cpg_oe]
        half = rng.random(len(kill)) < 0.5
        genome[kill[half]] = 3          # CG -> TG
        genome[kill[~half] + 1] = 0     # CG -> CA
    L = np.clip(rng.lognormal(np.log(median_len), sigma, n_reads), min_len, max_len).astype(np.int64)
    short = rng.random(n_reads) < frac_short
    L[short] = rng.integers(50, min_len, int(short.sum()))
    L = np.minimum(L, genome_len)
    st = (rng.random(n_reads) * (genome_len - L + 1)).astype(np.int64)
    rev = rng.random(n_reads) < 0.5
    wide = rng.random(n_reads) < frac_wide
    missing = rng.random(n_reads) < frac_missing
    tot = int(L.sum())
    off = np.concatenate([[0], np.cumsum(L)])
    bits = rng.integers(0, 1 << 16, size=(4, tot), dtype=np.uint16)
    kin8 = [_quantile_table(k, th, 255, np.uint8)[bits[i]] for i, (k, th) in enumerate(((2.0, 12.0), (3.0, 5.0), (2.0, 12.0), (3.0, 5.0)))]
    out = []
    for i in range(n_reads):
        a, b = int(off[i]), int(off[i + 1])
        codes = genome[st[i]:st[i] + L[i]]
        if rev[i]:
            codes = (3 - codes)[::-1]
        if wide[i]:  # B:S arrays hold raw frame counts (may exceed 952; re-encoded lossily, bam_info.cpp:455-478)
            arrs = [_quantile_table(2.0, s, 2000, np.uint16)[bits[j, a:b]] for j, s in enumerate((30.0, 12.0, 30.0, 12.0))]
        else:
            arrs = [kin8[j][a:b] for j in range(4)]
        out.append(Read(f"m0/{i}/ccs", int(L[i]), 4, pack_codes(np.ascontiguousarray(codes)), arrs[0], arrs[1], arrs[2],
                        None if missing[i] else arrs[3]))
    return out


def expected_sites_per_base(gc: float) -> float:
    """CpG + CHG (fwd only) + CHH (both strands) density for i.i.d. bases."""
    c = gc / 2
    return c * c + c * c * (1 - c) + 2 * c * (1 - c) ** 2


# ---- aligned, modification-tagged reads for `pileup` ------------------------------------------------------------
@dataclass
class AlignedRead:
    """One mapped mod-BAM record (what `hifimeth call` + pbmm2 hand to `hifimeth pileup`)."""
    name: str
    flag: int            # 0 / 16, optionally | 0x100 / 0x800; 4 = unmapped
    tid: int
    pos: int
    mapq: int
    cigar: list          # [(op_char, len)]
    seq: str             # SEQ as stored (reference orientation)
    mm: Optional[str]
    ml: Optional[np.ndarray]

    @property
    def l_qseq(self) -> int:
        return len(self.seq)

    @property
    def seq4(self) -> np.ndarray:
        lut = np.full(256, 4, np.uint8)
        lut[[65, 67, 71, 84]] = [0, 1, 2, 3]
        return pack_codes(lut[np.frombuffer(self.seq.encode(), np.uint8)])

    def cigar_string(self) -> str:
        return "".join(f"{n}{op}" for op, n in self.cigar) or "*"

    def cigar_u32(self) -> np.ndarray:
        return np.array([(n << 4) | "MIDNSHP=XB".index(op) for op, n in self.cigar], np.uint32)


_RC = bytes.maketrans(b"ACGTN", b"TGCAN")


def revcomp(s: str) -> str:
    return s.encode().translate(_RC)[::-1].decode()


def synth_genome(n_chr: int = 2, length: int = 20000, gc: float = 0.4, seed: int = 3, n_frac: float = 0.001):
    """-> [(name, SEQ)]; a few N runs so that alignments cross non-ACGT reference bases."""
    rng = np.random.default_rng(seed)
    p = [(1 - gc) / 2, gc / 2, gc / 2, (1 - gc) / 2]
    out = []
    for c in range(n_chr):
        L = int(length * (1 + 0.3 * c))
        s = _ASCII[rng.choice(4, L, p=p)].copy()
        for _ in range(int(L * n_frac / 5) + 1):
            a = int(rng.integers(0, L - 5))
            s[a:a + 5] = ord("N")
        out.append((f"chr{c + 1}", s.tobytes().decode()))
    return out


def _call_like_mods(fwd: str, rng, level, frac_called: float = 0.97):
    """MM/ML as `hifimeth call` writes them (build_mod_bam.cpp:125-248): C+m on CpG / CHG / CHH cytosines of the
    forward strand, G-m on the G of [AGT][AGT]G; `level(k)` = methylation level of forward offset k."""
    s = np.frombuffer(fwd.encode(), np.uint8)
    L = len(s)
    C, G = ord("C"), ord("G")
    n1 = np.concatenate([s[1:], [0]])
    n2 = np.concatenate([s[2:], [0, 0]])
    p1 = np.concatenate([[0], s[:-1]])
    p2 = np.concatenate([[0, 0], s[:-2]])
    isH = lambda x: (x == 65) | (x == 67) | (x == 84)  # noqa: E731
    isD = lambda x: (x == 65) | (x == 71) | (x == 84)  # noqa: E731
    fwd_c = (s == C) & ((n1 == G) | (isH(n1) & ((n2 == G) | isH(n2))))
    rev_g = (s == G) & isD(p1) & isD(p2) & (np.arange(L) >= 2)
    parts, mls = [], []
    for mask, base, head in ((fwd_c, C, "C+m"), (rev_g, G, "G-m")):
        q = np.nonzero(mask & (rng.random(L) < frac_called))[0]
        if len(q) == 0:
            continue
        cnt = np.concatenate([[0], np.cumsum(s == base)])
        last = np.concatenate([[0], q[:-1] + 1])
        parts.append(head + "".join(f",{int(d)}" for d in cnt[q] - cnt[last]) + ";")
        meth = rng.random(len(q)) < np.array([level(int(k)) for k in q])
        pr = np.where(meth, 255 - rng.gamma(1.2, 18, len(q)), rng.gamma(1.2, 18, len(q)))
        mls.append(np.clip(np.rint(pr), 0, 255).astype(np.uint8))
    if not parts:
        return None, None
    return "".join(parts), np.concatenate(mls)


def synth_alignments(genome, n_reads: int = 40, seed: int = 5, median_len: int = 3000, err: float = 0.01,
                     eqx: bool = True, frac_unmapped: float = 0.05, frac_supp: float = 0.05,
                     frac_no_mods: float = 0.05) -> List[AlignedRead]:
    """Reads sampled from `genome` with substitutions / insertions / deletions at rate `err`, soft clips on some
    reads, both strands, coordinate-sorted.  `eqx`: CIGAR with =/X (pbmm2 default) instead of M.
    Per-locus methylation level is a deterministic function of the position, so the pileup has structure."""
    rng = np.random.default_rng(seed)
    reads = []
    for i in range(n_reads):
        tid = int(rng.integers(0, len(genome)))
        chrom = genome[tid][1]
        L = int(np.clip(rng.lognormal(np.log(median_len), 0.3), 300, len(chrom) - 10))
        pos = int(rng.integers(0, len(chrom) - L))
        ops, q = [], []
        for k in range(pos, pos + L):
            r = rng.random()
            b = chrom[k]
            if r < err / 3:
                ops.append("D")
            elif r < 2 * err / 3:
                q.append("ACGT"[int(rng.integers(0, 4))]); ops.append("I")
                q.append(b); ops.append("=" if eqx else "M")
            elif r < err:
                nb = "ACGT"[int(rng.integers(0, 4))]
                q.append(nb); ops.append(("=" if nb == b else "X") if eqx else "M")
            else:
                q.append(b if b != "N" else "ACGT"[int(rng.integers(0, 4))])
                ops.append(("=" if b != "N" else "X") if eqx else "M")
        while ops and ops[0] in "DI":      # alignments start and end on an aligned pair
            if ops[0] == "I":
                q.pop(0)
            else:
                pos += 1
            ops.pop(0)
        while ops and ops[-1] in "DI":
            if ops[-1] == "I":
                q.pop()
            ops.pop()
        cig = []
        for o in ops:
            if cig and cig[-1][0] == o:
                cig[-1][1] += 1
            else:
                cig.append([o, 1])
        lead = int(rng.integers(0, 30)) if rng.random() < 0.3 else 0
        trail = int(rng.integers(0, 30)) if rng.random() < 0.3 else 0
        clip = lambda n: "".join("ACGT"[int(x)] for x in rng.integers(0, 4, n))  # noqa: E731
        seq = clip(lead) + "".join(q) + clip(trail)
        cigar = ([("S", lead)] if lead else []) + [(o, n) for o, n in cig] + ([("S", trail)] if trail else [])
        rev = bool(rng.random() < 0.5)
        flag = 16 if rev else 0
        u = rng.random()
        if u < frac_supp:
            flag |= 0x800 if rng.random() < 0.5 else 0x100
        fwd = revcomp(seq) if rev else seq
        Lq = len(seq)
        # methylation level of forward offset k: depends on the reference neighbourhood (blocks of 400 bp)
        def level(k, pos=pos, rev=rev, Lq=Lq, tid=tid):
            g = pos + (Lq - 1 - k if rev else k)
            return (0.85, 0.1, 0.5)[((g // 400) + tid) % 3]
        mm, ml = (None, None) if rng.random() < frac_no_mods else _call_like_mods(fwd, rng, level)
        reads.append(AlignedRead(f"aln{i}", flag, tid, pos, int(rng.integers(0, 61)), cigar, seq, mm, ml))
    reads.sort(key=lambda r: (r.tid, r.pos))
    n_un = int(n_reads * frac_unmapped)
    for i in range(n_un):                  # unmapped records carry tags too; pileup ignores them
        s = "".join("ACGT"[int(x)] for x in rng.integers(0, 4, 500))
        mm, ml = _call_like_mods(s, rng, lambda k: 0.5)
        reads.append(AlignedRead(f"un{i}", 4, -1, -1, 0, [], s, mm, ml))
    return reads
