"""Host-side mirror of `hifimeth pileup` (reference src/app/hifimeth/pileup.cpp:461-606) over the hm_pileup_* C ABI.

    pu = MethylationPileup(genome)            # [(name, SEQUENCE)]   <- HbnDatabase
    for rec in aligned_reads: pu.add(rec)     # one iteration of s_genomic_methy_freq_thread
    pu.flush()
    bins = pu.histograms()                    # 3 x 256
    thr = pu.resolve_thresholds(bins)         # s_resolve_scaled_prob_threshold
    pu.count(thr)
    loci = pu.loci()                          # rows of the three BED files
    text = pu.bed(loci)

Multi-GPU (one process per GPU, records dealt to ranks in slabs): `reduce_over_ranks` sums the histograms with an
all-reduce before the thresholds are resolved, and after counting reduce-scatters the per-locus planes (sum for
pcov / ncov, max for the motif key) so that every rank ends up owning one contiguous range of loci.
There is no CPU fallback: construction fails without the HIP library and a gfx950 device.
"""
from __future__ import annotations

import ctypes as C
from typing import List, Optional, Sequence, Tuple

import numpy as np

from ._lib import lib
from .caller import HifimethError

MOD_DTYPE = np.dtype([("qoff", "<i4"), ("strand", "u1"), ("unmod_base", "S1"), ("code", "S1"), ("prob", "u1")])
LOCUS_DTYPE = np.dtype([("gpos", "<i8"), ("pcov", "<i4"), ("ncov", "<i4"), ("motif", "<u4"), ("reserved", "<u4")])
CTX_NAMES = ("CpG", "CHG", "CHH")
_CHEBI = {27551: "m", 76792: "h", 76794: "f", 76793: "c", 16964: "g", 80961: "e", 17477: "b", 28871: "a",
          44605: "o", 18107: "n"}
_RC = bytes.maketrans(b"ACGTN", b"TGCAN")


def parse_mods(stored_seq: str, flag: int, mm: Optional[str], ml) -> np.ndarray:
    """extract_bam_base_mods (src/corelib/bam_mod_parser.cpp:231-286), vectorised: the k-th entry of an edit
    series sits on the (sum of (delta+1))-th occurrence of the unmodified base in the forward-strand sequence."""
    if mm is None or ml is None or len(ml) == 0:
        return np.zeros(0, MOD_DTYPE)
    if not mm.endswith(";"):
        raise HifimethError("The MM aux tag must end with ';'")
    s = stored_seq.encode()
    fwd = np.frombuffer(s.translate(_RC)[::-1] if flag & 16 else s, np.uint8)
    ml = np.asarray(ml, np.uint8)
    out, pi = [], 0
    for series in mm[:-1].split(";"):
        if len(series) < 3 or series[0] not in "CGTAUN" or series[1] not in "+-":
            raise HifimethError(f"Corrupted edit series {series};")
        head, _, rest = series.partition(",")
        codes = head[2:]
        codes = _CHEBI[int(codes)] if codes[:1].isdigit() else codes.replace(".", "").replace("?", "")
        deltas = np.array([int(x) for x in rest.split(",")] if rest else [], np.int64)
        where = np.nonzero(fwd == ord(series[0]))[0]
        nth = np.cumsum(deltas + 1) - 1
        if len(nth) and nth[-1] >= len(where):
            raise HifimethError(f"edit series runs past the read end: {series};")
        q = where[nth]
        n = len(q) * len(codes)
        if pi + n > len(ml):
            raise HifimethError("ML is shorter than the MM edit lists")
        m = np.zeros(n, MOD_DTYPE)
        m["qoff"] = np.repeat(q, len(codes))
        m["strand"] = 0 if series[1] == "+" else 1
        m["unmod_base"] = series[0].encode()
        m["code"] = np.tile(np.frombuffer(codes.encode(), "S1"), len(q))
        m["prob"] = ml[pi:pi + n]
        pi += n
        out.append(m)
    return np.concatenate(out) if out else np.zeros(0, MOD_DTYPE)


def resolve_threshold(bins) -> Tuple[int, int]:
    """s_resolve_scaled_prob_threshold for one context (pileup.cpp:355-436) -> (threshold, samples in window)"""
    a = np.asarray(bins, np.uint64)
    st, en = 20, 236
    while st < 256 and a[st] < 10:
        st += 1
    while en and a[en - 1] < 10:
        en -= 1
    if en - st < 50:
        return 128, 0
    w = a[st:en]
    total = int(w.sum())
    return (128 if total < 10000 else st + int(np.argmin(w))), total


class MethylationPileup:
    def __init__(self, genome: Sequence[Tuple[str, str]], device: int = 0, min_mapq: int = 0, min_pi: float = 0.0,
                 planes=None):
        """planes: optional (pcov, ncov, key) torch CUDA tensors (int32, int32, int32-as-bits) of total genome length
        that the engine counts into -- used when a collective consumes them afterwards."""
        self._L = lib()
        self._h = C.c_void_p()
        if self._L.hm_pileup_create(C.byref(self._h), device) != 0:
            raise HifimethError(self._L.hm_pileup_last_error(None).decode())
        self.names = [n for n, _ in genome]
        self.lengths = np.array([len(s) for _, s in genome], np.int64)
        self.offsets = np.concatenate([[0], np.cumsum(self.lengths)])
        self._planes = planes
        self._order = 0
        self._check(self._L.hm_pileup_set_option(self._h, b"min_mapq", float(min_mapq)))
        self._check(self._L.hm_pileup_set_option(self._h, b"min_pi", float(min_pi)))
        if planes is not None:
            self._check(self._L.hm_pileup_use_planes(self._h, *(C.c_void_p(t.data_ptr()) for t in planes)))
        bases = "".join(s for _, s in genome).upper().encode()
        self._check(self._L.hm_pileup_set_reference(self._h, len(genome), self.lengths.ctypes.data_as(C.c_void_p),
                                                    C.c_char_p(bases)))

    def close(self):
        if self._h:
            self._L.hm_pileup_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc < 0:
            raise HifimethError(self._L.hm_pileup_last_error(self._h).decode())
        return rc

    @property
    def n_loci(self) -> int:
        return int(self.offsets[-1])

    def add(self, read, order: Optional[int] = None) -> int:
        """read: synth.AlignedRead-like (flag, tid, pos, mapq, cigar_u32(), seq, seq4, mm, ml).  -> 1 staged / 0 skipped"""
        if order is None:
            order = self._order
        self._order = order + 1
        mods = parse_mods(read.seq, read.flag, read.mm, read.ml)
        if len(mods) == 0 or read.flag & 4:
            return 0
        seq4 = np.ascontiguousarray(read.seq4, np.uint8)
        cig = np.ascontiguousarray(read.cigar_u32(), np.uint32)
        return self._check(self._L.hm_pileup_submit_read(
            self._h, order, read.flag, read.tid, read.pos, read.mapq, len(read.seq), seq4.ctypes.data_as(C.c_void_p),
            len(cig), cig.ctypes.data_as(C.c_void_p), len(mods), mods.ctypes.data_as(C.c_void_p)))

    def flush(self):
        self._check(self._L.hm_pileup_run(self._h))

    def num_records(self) -> int:
        return int(self._L.hm_pileup_num_records(self._h))

    def histograms(self) -> np.ndarray:
        b = np.zeros(768, np.uint64)
        self._check(self._L.hm_pileup_histograms(self._h, b.ctypes.data_as(C.c_void_p)))
        return b.reshape(3, 256)

    def records(self):
        """projected calls still resident (unordered): (gpos, prob, motif, order)"""
        n = self.num_records()
        g, p, m, o = np.zeros(n, np.int64), np.zeros(n, np.uint8), np.zeros(n, np.uint8), np.zeros(n, np.uint32)
        self._check(self._L.hm_pileup_fetch_records(self._h, *(x.ctypes.data_as(C.c_void_p) for x in (g, p, m, o)), n))
        return g, p, m, o

    def label_histograms(self, labels: np.ndarray) -> np.ndarray:
        """`hifimeth eval` (eval.cpp:469-560): resident records joined with per-locus truth labels (int8 over the
        concatenated reference: -1 none, 0 unmethylated, 1 methylated) -> counts[motif, label, scaled_prob]"""
        lab = np.ascontiguousarray(labels, np.int8)
        b = np.zeros(1536, np.uint64)
        self._check(self._L.hm_pileup_label_histograms(self._h, lab.ctypes.data_as(C.c_void_p), lab.size,
                                                       b.ctypes.data_as(C.c_void_p)))
        return b.reshape(3, 2, 256)

    @staticmethod
    def resolve_thresholds(bins) -> List[int]:
        return [resolve_threshold(bins[c])[0] for c in range(3)]

    def count(self, thresholds: Sequence[int]):
        t = np.asarray(thresholds, np.uint8)
        self._check(self._L.hm_pileup_count(self._h, t.ctypes.data_as(C.c_void_p)))

    def loci(self, lo: int = 0, hi: Optional[int] = None, planes=None, plane_base: int = 0) -> np.ndarray:
        """covered loci of [lo, hi) (plane coordinates) in ascending order; planes = torch tensors or None (own)"""
        hi = self.n_loci if hi is None else hi
        ptrs = [None, None, None] if planes is None else [C.c_void_p(t.data_ptr()) for t in planes]
        n = self._check(self._L.hm_pileup_fetch_loci(self._h, *ptrs, plane_base, lo, hi, None, 0))
        out = np.zeros(n, LOCUS_DTYPE)
        if n:
            self._check(self._L.hm_pileup_fetch_loci(self._h, *ptrs, plane_base, lo, hi, out.ctypes.data_as(C.c_void_p), n))
        return out

    def bed(self, loci: np.ndarray) -> dict:
        """the text of <prefix>.{CpG,CHG,CHH}.cov.bed (pileup.cpp:562-590)"""
        sid = np.searchsorted(self.offsets, loci["gpos"], side="right") - 1
        soff = loci["gpos"] - self.offsets[sid]
        rows = {k: [] for k in CTX_NAMES}
        for s, k, p, n, m in zip(sid, soff, loci["pcov"], loci["ncov"], loci["motif"]):
            rows[CTX_NAMES[int(m)]].append("%s\t%d\t%d\t%g\t%d\t%d\n" % (self.names[s], k, k + 1, 100.0 * p / (p + n), p, n))
        return {k: "".join(v) for k, v in rows.items()}


# ---- multi-GPU exchange: histograms all-reduced, per-locus planes reduce-scattered ------------
def locus_ranges(n_loci: int, world: int) -> List[Tuple[int, int]]:
    """contiguous, equal-size (padded) ranges: rank r owns [r*chunk, min(n_loci, (r+1)*chunk))"""
    chunk = (n_loci + world - 1) // world
    return [(min(n_loci, r * chunk), min(n_loci, (r + 1) * chunk)) for r in range(world)]


def allreduce_histograms(dist, bins: np.ndarray, device: str = "cpu") -> np.ndarray:
    import torch
    t = torch.from_numpy(bins.astype(np.int64).reshape(-1)).to(device)
    if dist.is_initialized():
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t.cpu().numpy().astype(np.uint64).reshape(3, 256)


def reduce_scatter_planes(dist, pcov, ncov, key, force: bool = False):
    """-> (pcov, ncov, key, base): this rank's slice of the job-wide planes, `base` = its first locus.
    pcov / ncov are summed, key (order << 2 | motif; hm_pileup_submit_read keeps order < 2^29, so the key is < 2^31 and
    compares as int32 exactly like the device's uint32 atomicMax) takes the maximum.
    Planes must be padded to world * chunk elements.  RCCL reduce-scatters; gloo (CPU tests) all-reduces and slices."""
    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    n = pcov.numel()
    assert n % world == 0
    chunk = n // world
    if world == 1 and not (force and dist.is_initialized()):
        return pcov, ncov, key, 0
    import torch
    outs = []
    for t, op in ((pcov, dist.ReduceOp.SUM), (ncov, dist.ReduceOp.SUM), (key, dist.ReduceOp.MAX)):
        if t.is_cuda:
            o = torch.empty(chunk, dtype=t.dtype, device=t.device)
            dist.reduce_scatter_tensor(o, t, op=op)
        else:
            dist.all_reduce(t, op=op)
            o = t[rank * chunk:(rank + 1) * chunk].clone()
        outs.append(o)
    return outs[0], outs[1], outs[2], rank * chunk
