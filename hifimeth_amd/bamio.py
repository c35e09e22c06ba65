"""Minimal streaming BAM reader for the Python host paths (SAMv1 sections 4.1-4.2; BGZF = concatenated gzip members,
which the gzip module reads natively).  The C++ CLI has its own multi-threaded codec (csrc/hm_bam.cpp); this one only
serves `python -m hifimeth_amd.pileup_dist`, where every rank scans the file and keeps its own slabs of records."""
from __future__ import annotations

import gzip
import struct
from dataclasses import dataclass
from typing import Iterator, List, Optional, Tuple

import numpy as np

_NT16 = "=ACMGRSVTWYHKDBN"
_DEC = np.frombuffer(b"=ACNGNNNTNNNNNNN", np.uint8)   # 1,2,4,8 -> ACGT; everything else reads as N here


@dataclass
class MappedRecord:
    """the fields of one BAM record that `pileup` uses; attribute names match synth.AlignedRead"""
    name: str
    flag: int
    tid: int
    pos: int
    mapq: int
    cigar: np.ndarray        # uint32, as stored
    seq4: np.ndarray         # uint8, 4-bit packed as stored
    l_qseq: int
    mm: Optional[str]
    ml: Optional[np.ndarray]

    def cigar_u32(self) -> np.ndarray:
        return self.cigar

    @property
    def seq(self) -> str:
        nib = np.empty(2 * len(self.seq4), np.uint8)
        nib[0::2], nib[1::2] = self.seq4 >> 4, self.seq4 & 15
        return _DEC[nib[:self.l_qseq]].tobytes().decode()


def _aux_tags(aux: bytes, want=(b"MM", b"ML", b"Mm", b"Ml")):
    out, p, n = {}, 0, len(aux)
    size = {"A": 1, "c": 1, "C": 1, "s": 2, "S": 2, "i": 4, "I": 4, "f": 4}
    while p + 3 <= n:
        tag, t = aux[p:p + 2], chr(aux[p + 2])
        p += 3
        if t in "ZH":
            e = aux.index(b"\0", p)
            if tag in want:
                out[tag] = aux[p:e].decode()
            p = e + 1
        elif t == "B":
            sub = chr(aux[p])
            cnt = struct.unpack_from("<I", aux, p + 1)[0]
            p += 5
            w = size[sub]
            if tag in want:
                dt = {"C": "<u1", "c": "<i1", "S": "<u2", "s": "<i2", "I": "<u4", "i": "<i4", "f": "<f4"}[sub]
                out[tag] = np.frombuffer(aux, dt, cnt, p).copy()
            p += w * cnt
        else:
            p += size[t]
    return out


def read_bam(path: str) -> Tuple[str, List[Tuple[str, int]], Iterator[MappedRecord]]:
    """-> (header text, [(reference name, length)], iterator over records)"""
    f = gzip.open(path, "rb")
    if f.read(4) != b"BAM\1":
        raise ValueError(f"{path}: not a BAM file")
    l_text = struct.unpack("<I", f.read(4))[0]
    text = f.read(l_text).decode(errors="replace").rstrip("\0")
    refs = []
    for _ in range(struct.unpack("<I", f.read(4))[0]):
        ln = struct.unpack("<I", f.read(4))[0]
        nm = f.read(ln)[:-1].decode()
        refs.append((nm, struct.unpack("<I", f.read(4))[0]))

    def records():
        while True:
            h = f.read(4)
            if len(h) < 4:
                break
            body = f.read(struct.unpack("<I", h)[0])
            tid, pos, l_rn, mapq, _bin, n_cig, flag, l_seq = struct.unpack_from("<iiBBHHHi", body, 0)
            o = 32
            name = body[o:o + l_rn - 1].decode()
            o += l_rn
            cigar = np.frombuffer(body, "<u4", n_cig, o).copy()
            o += 4 * n_cig
            seq4 = np.frombuffer(body, np.uint8, (l_seq + 1) // 2, o).copy()
            o += (l_seq + 1) // 2 + l_seq
            tags = _aux_tags(body[o:])
            mm = tags.get(b"MM", tags.get(b"Mm"))
            ml = tags.get(b"ML", tags.get(b"Ml"))
            yield MappedRecord(name, flag, tid, pos, mapq, cigar, seq4, l_seq, mm, ml)
        f.close()
    return text, refs, records()


def is_coordinate_sorted(header_text: str) -> bool:
    for line in header_text.split("\n"):
        if line.startswith("@HD"):
            return any(fld == "SO:coordinate" for fld in line.split("\t")[1:])
    return False


def load_fasta(path: str) -> List[Tuple[str, str]]:
    """HbnDatabase (reference src/corelib/hbn_seqdb.cpp:37-95): header = '>' line or a line with a digit or '|' among
    its first 33 characters; ! # ; comment lines; name up to the first blank; bases upper-cased."""
    with open(path, "rb") as probe:
        opener = gzip.open if probe.read(2) == b"\x1f\x8b" else open
    seqs, name, parts = [], None, []
    with opener(path, "rt") as f:
        for raw in f:
            line = raw.strip()
            if not line or line[0] in "!#;":
                continue
            if line[0] == ">" or any(ch in "0123456789|" for ch in line[:33]):
                if name:
                    seqs.append((name, "".join(parts).upper()))
                fields = (line[1:] if line[0] == ">" else line).split()
                name, parts = (fields[0] if fields else ""), []
            elif name:
                parts.append(line)
    if name:
        seqs.append((name, "".join(parts).upper()))
    return seqs
