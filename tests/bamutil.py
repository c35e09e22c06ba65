"""Tiny pure-Python BAM/BGZF writer + reader for the tests (SAMv1 sections 4.1-4.2). Test infrastructure only."""
import gzip
import struct
import zlib

import numpy as np

_EOF = bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000")


def _bgzf_block(data: bytes, level: int = 6) -> bytes:
    co = zlib.compressobj(level, zlib.DEFLATED, -15)
    comp = co.compress(data) + co.flush()
    bsize = 18 + len(comp) + 8 - 1
    return (b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff\x06\x00BC\x02\x00" + struct.pack("<H", bsize) + comp
            + struct.pack("<II", zlib.crc32(data) & 0xffffffff, len(data)))


def write_bgzf(path, payload: bytes, block=0xff00, level: int = 6):
    with open(path, "wb") as f:
        for i in range(0, len(payload), block):
            f.write(_bgzf_block(payload[i:i + block], level))
        f.write(_EOF)


def aux_B(tag: str, arr: np.ndarray) -> bytes:
    sub = {np.dtype("uint8"): b"C", np.dtype("uint16"): b"S"}[arr.dtype]
    return tag.encode() + b"B" + sub + struct.pack("<I", len(arr)) + arr.astype(arr.dtype.newbyteorder("<")).tobytes()


def aux_Z(tag: str, s: str) -> bytes:
    return tag.encode() + b"Z" + s.encode() + b"\0"


def aux_i(tag: str, v: int) -> bytes:
    return tag.encode() + b"i" + struct.pack("<i", v)


def aux_f(tag: str, v: float) -> bytes:
    return tag.encode() + b"f" + struct.pack("<f", v)


def record(name: str, flag: int, seq4: np.ndarray, l_seq: int, aux: bytes) -> bytes:
    qn = name.encode() + b"\0"
    core = struct.pack("<iiBBHHHiiii", -1, -1, len(qn), 255, 4680, 0, flag, l_seq, -1, -1, 0)
    body = core + qn + bytes(seq4) + b"\xff" * l_seq + aux
    return struct.pack("<I", len(body)) + body


def aligned_to_bam(path, genome, reads, sort_order="coordinate", level: int = 6):
    """hifimeth_amd.synth.AlignedRead objects -> mapped mod-BAM (MM/ML/MN tags), reference list from `genome`."""
    text = f"@HD\tVN:1.6\tSO:{sort_order}\n" + "".join(f"@SQ\tSN:{n}\tLN:{len(s)}\n" for n, s in genome)
    parts = [b"BAM\1" + struct.pack("<I", len(text)) + text.encode() + struct.pack("<I", len(genome))]
    for n, s in genome:
        nm = n.encode() + b"\0"
        parts.append(struct.pack("<I", len(nm)) + nm + struct.pack("<I", len(s)))
    for r in reads:
        aux = aux_Z("RG", "rg0")
        if r.mm is not None:
            aux += aux_Z("MM", r.mm) + aux_B("ML", np.asarray(r.ml, np.uint8)) + aux_i("MN", r.l_qseq)
        qn = r.name.encode() + b"\0"
        cig = r.cigar_u32()
        core = struct.pack("<iiBBHHHiiii", r.tid, r.pos, len(qn), r.mapq, 4680, len(cig), r.flag, r.l_qseq, -1, -1, 0)
        body = core + qn + cig.astype("<u4").tobytes() + bytes(r.seq4) + b"\xff" * r.l_qseq + aux
        parts.append(struct.pack("<I", len(body)) + body)
    write_bgzf(path, b"".join(parts), level=level)


def write_fasta(path, genome, width: int = 60):
    with open(path, "w") as f:
        for n, s in genome:
            f.write(f">{n} synthetic\n")
            for i in range(0, len(s), width):
                f.write(s[i:i + width] + "\n")


def reads_to_bam(path, reads, header_text="@HD\tVN:1.6\tSO:unknown\tpb:5.0.0\n", extra_aux=None, level: int = 6):
    """reads: hifimeth_amd.synth.Read objects -> unaligned PacBio-style BAM with fi/fp/ri/rp (+ a few other tags)."""
    parts = [b"BAM\1" + struct.pack("<I", len(header_text)) + header_text.encode() + struct.pack("<I", 0)]
    for i, r in enumerate(reads):
        aux = aux_i("np", 10 + i) + aux_f("rq", 0.999) + aux_Z("RG", "rg0")
        for tag in ("fi", "fp", "ri", "rp"):
            a = getattr(r, tag)
            if a is not None:
                aux += aux_B(tag, np.asarray(a))
        aux += aux_i("zm", i)
        if extra_aux:
            aux += extra_aux(i, r)
        parts.append(record(r.name, r.flag, r.seq4, r.l_qseq, aux))
    write_bgzf(path, b"".join(parts), level=level)


def parse_aux(aux: bytes):
    """-> ordered list of (tag, type, value); B arrays as numpy, Z as str, ints as int."""
    out = []
    p = 0
    sizes = {"A": 1, "c": 1, "C": 1, "s": 2, "S": 2, "i": 4, "I": 4, "f": 4}
    fmt = {"c": "<b", "C": "<B", "s": "<h", "S": "<H", "i": "<i", "I": "<I", "f": "<f"}
    while p < len(aux):
        tag, t = aux[p:p + 2].decode(), chr(aux[p + 2])
        p += 3
        if t == "Z":
            e = aux.index(b"\0", p)
            out.append((tag, t, aux[p:e].decode()))
            p = e + 1
        elif t == "B":
            sub = chr(aux[p])
            n = struct.unpack_from("<I", aux, p + 1)[0]
            p += 5
            dt = {"C": "<u1", "c": "<i1", "S": "<u2", "s": "<i2", "I": "<u4", "i": "<i4", "f": "<f4"}[sub]
            arr = np.frombuffer(aux, dt, n, p).copy()
            out.append((tag, "B" + sub, arr))
            p += arr.nbytes
        elif t == "A":
            out.append((tag, t, chr(aux[p])))
            p += 1
        else:
            out.append((tag, t, struct.unpack_from(fmt[t], aux, p)[0]))
            p += sizes[t]
    return out


def read_bam(path):
    """-> (header_text, [dict(name, flag, l_seq, seq4, aux_bytes, raw)])"""
    data = gzip.open(path, "rb").read()
    assert data[:4] == b"BAM\1"
    l_text = struct.unpack_from("<I", data, 4)[0]
    text = data[8:8 + l_text].decode()
    p = 8 + l_text
    n_ref = struct.unpack_from("<I", data, p)[0]
    p += 4
    for _ in range(n_ref):
        ln = struct.unpack_from("<I", data, p)[0]
        p += 4 + ln + 4
    recs = []
    while p < len(data):
        bs = struct.unpack_from("<I", data, p)[0]
        body = data[p + 4:p + 4 + bs]
        p += 4 + bs
        l_rn, n_cig, flag, l_seq = body[8], struct.unpack_from("<H", body, 12)[0], struct.unpack_from("<H", body, 14)[0], \
            struct.unpack_from("<i", body, 16)[0]
        o = 32
        name = body[o:o + l_rn - 1].decode()
        o += l_rn + 4 * n_cig
        seq4 = np.frombuffer(body, np.uint8, (l_seq + 1) // 2, o).copy()
        o += (l_seq + 1) // 2 + l_seq
        recs.append(dict(name=name, flag=flag, l_seq=l_seq, seq4=seq4, aux=body[o:], raw=body))
    return text, recs
