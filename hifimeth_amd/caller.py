"""Host-side mirror of the reference's hot-path classes on top of the C ABI.

The reference drives the path through three C++ classes inside its worker thread
(src/app/hifimeth/mod_main.cpp:145-262): `ModModels` (model load), `EvalKmerFeaturesGenerator`
(init / extract_*_samples / get_next_sample_features, eval_kmer_features.hpp:13-49) and `ModBatch`
(call_mods_for_one_read / call_current_batch, mod_batch.hpp:12-43).  `MethylationCaller` offers the
same verbs, batched: reads are staged, then one device pass scans all sites, builds their windows
on chip and runs the CNN.  Everything below is plumbing around libhifimeth_hip.so; there is no CPU
implementation here.
"""
from __future__ import annotations

import ctypes as C
from typing import Iterable, Optional

import numpy as np

from . import _lib

CPG, CHG, CHH = 0, 1, 2
CTX_NAMES = ("CpG", "CHG", "CHH")
KMER, FEATS = 401, 8

# hm_read_t as a numpy record (pointers as integers): descriptors of a whole slab for Batch.submit_block
READ_DTYPE = np.dtype([("read_id", "<i4"), ("l_qseq", "<i4"), ("flag", "<i4"), ("width", "u1", 4), ("seq4", "<u8"),
                       ("kin", "<u8", 4)], align=True)
assert READ_DTYPE.itemsize == C.sizeof(_lib.hm_read_t)

CALL_DTYPE = np.dtype([("read_id", "<i4"), ("qoff", "<i4"), ("strand", "u1"), ("ctx", "u1"),
                       ("scaled_prob", "u1"), ("reserved", "u1"), ("p", "<f4")])
assert CALL_DTYPE.itemsize == C.sizeof(_lib.hm_call_t) == 16


class HifimethError(RuntimeError):
    pass


def parse_contexts(spec: str) -> int:
    """The reference's `-c cpg,chg,chh` (mod_options.cpp:61-134) -> ctx_mask."""
    mask = 0
    for tok in spec.lower().split(","):
        tok = tok.strip()
        if tok not in ("cpg", "chg", "chh"):
            raise ValueError(f"unknown methylation context '{tok}'")
        mask |= 1 << ("cpg", "chg", "chh").index(tok)
    return mask


def _vp(a: Optional[np.ndarray]):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class MethylationCaller:
    def __init__(self, model_dir: Optional[str] = None, contexts: str = "cpg,chg,chh", device: int = 0,
                 min_read_size: int = 1000, timing: bool = False):
        self._L = _lib.lib()
        self._h = C.c_void_p()
        self.ctx_mask = parse_contexts(contexts)
        rc = self._L.hm_create(C.byref(self._h), (model_dir or _lib.WEIGHTS_DIR).encode(), self.ctx_mask, device)
        if rc < 0:
            self._h = None
            raise HifimethError(f"hm_create failed ({rc}): {self._L.hm_last_error(None).decode()}")
        self.set_option("min_read_size", min_read_size)
        self.set_option("timing", int(timing))

    # -- lifetime ------------------------------------------------------------------------------
    def close(self):
        if getattr(self, "_h", None):
            self._L.hm_destroy(self._h)
            self._h = None

    __del__ = close

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def _check(self, rc, what):
        if rc < 0:
            raise HifimethError(f"{what} failed ({rc}): {self._L.hm_last_error(self._h).decode()}")
        return rc

    def set_option(self, key: str, value: int):
        self._check(self._L.hm_set_option(self._h, key.encode(), int(value)), f"hm_set_option({key})")

    def new_sample(self):
        """Forget the kernel-path choice of the previous input.  With the default option trunk = 2 the engine decides ONCE, from the
        site density of the first non-empty batch it is given, which contexts run conv1..conv4 as the dense trunk and which per site,
        and keeps that for its lifetime (so that the calls of one input never depend on batch cuts or host timing: the two paths
        agree to ~1e-5 in p, not bit for bit).  A caller that reuses one engine for several samples calls this between them --
        a CpG-poor first sample would otherwise leave the per-site kernels in force for a dense one.  (Same as setting "trunk" to 2
        again; front ends that shard ONE input over several engines set "trunk_mask" from the head of the file instead.)"""
        self.set_option("trunk", 2)

    # -- staging (EvalKmerFeaturesGenerator::init) ------------------------------------------------
    @staticmethod
    def _read_args(read):
        arrs, widths = [], []
        for nm in ("fi", "fp", "ri", "rp"):
            a = getattr(read, nm)
            if a is not None:
                if len(a) != read.l_qseq:  # bam_auxB_len != l_qseq -> init() == false (bam_info.cpp:447)
                    a = None
                else:
                    a = np.ascontiguousarray(a)
                    if a.dtype.itemsize not in (1, 2):
                        raise ValueError("kinetics arrays must be uint8 (B:C) or uint16 (B:S)")
            arrs.append(a)
            widths.append(1 if a is None else a.dtype.itemsize)
        seq4 = np.ascontiguousarray(read.seq4, np.uint8)
        return (read.l_qseq, read.flag, _vp(seq4), _vp(arrs[0]), widths[0], _vp(arrs[1]), widths[1],
                _vp(arrs[2]), widths[2], _vp(arrs[3]), widths[3]), (seq4, arrs)

    def submit(self, read_id: int, read) -> bool:
        """`read`: object with l_qseq, flag, seq4 and fi/fp/ri/rp (uint8 or uint16 arrays, or None).
        True if accepted, False if passed through uncalled (short read / missing kinetics)."""
        args, _keep = self._read_args(read)
        return bool(self._check(self._L.hm_submit_read(self._h, read_id, *args), "hm_submit_read"))

    def submit_all(self, reads: Iterable, first_id: int = 0) -> int:
        n = 0
        for i, r in enumerate(reads):
            n += self.submit(first_id + i, r)
        return n

    def clear(self):
        self._check(self._L.hm_clear(self._h), "hm_clear")

    # -- execution -------------------------------------------------------------------------------
    def upload(self):
        self._check(self._L.hm_upload(self._h), "hm_upload")

    def run(self):
        self._check(self._L.hm_run(self._h), "hm_run")

    def sync(self):
        self._check(self._L.hm_sync(self._h), "hm_sync")

    def num_sites(self, ctx: int = 3) -> int:
        return self._check(self._L.hm_num_sites(self._h, ctx), "hm_num_sites")

    def fetch(self) -> np.ndarray:
        self.sync()
        n = self.num_sites(3)
        out = np.empty(n, CALL_DTYPE)
        got = self._check(self._L.hm_fetch(self._h, _vp(out), n), "hm_fetch")
        return out[:got]

    def call(self, reads: Iterable, first_id: int = 0) -> np.ndarray:
        """Stage, run and fetch in one go; returns CALL_DTYPE records ordered by (read, strand, qoff)."""
        self.clear()
        self.submit_all(reads, first_id)
        self.upload()
        self.run()
        out = self.fetch()
        self.clear()
        return out

    # -- asynchronous batch pipeline (hm_batch_*) --------------------------------------------------------
    def begin_batch(self) -> "Batch":
        """A free slot of the engine's pipeline (blocks while all slots are staged or in flight)."""
        h = self._L.hm_batch_begin(self._h)
        if not h:
            raise HifimethError(f"hm_batch_begin failed: {self._L.hm_last_error(self._h).decode()}")
        return Batch(self, h)

    def stream(self, slabs: Iterable, on_batch=None) -> int:
        """Runs every slab (an iterable of reads) through the pipeline: slab k+1 is staged and uploaded while slab k
        computes; results are collected in order.  `on_batch(k, batch, calls)` sees each slab's CALL_DTYPE records (a
        view of pinned memory that is only valid inside the callback).  Returns the total number of calls."""
        inflight, total, k_done = [], 0, 0

        def collect(b):
            nonlocal total, k_done
            calls = b.wait()
            total += len(calls)
            if on_batch is not None:
                on_batch(k_done, b, calls)
            k_done += 1
            b.release()

        for slab in slabs:
            b = self.begin_batch()
            if isinstance(slab, ReadBlock):
                b.submit_block(slab, self.stage_threads)
            else:
                b.submit_all(slab)
            b.enqueue()
            inflight.append(b)
            while len(inflight) > 1 and (inflight[0].done() or len(inflight) >= self.max_inflight):
                collect(inflight.pop(0))
        while inflight:
            collect(inflight.pop(0))
        return total

    max_inflight = 3
    stage_threads = 4

    # -- seams -------------------------------------------------------------------------------------
    def scan_sites(self, ctx: int):
        """extract_{cpg,chg,chh}_samples of every staged read: (read_id, qoff, strand) in (read, qoff) order."""
        self.sync()
        n = self.num_sites(ctx)
        rid, qoff, strand = np.empty(n, np.int32), np.empty(n, np.int32), np.empty(n, np.uint8)
        self._check(self._L.hm_scan_sites(self._h, ctx, _vp(rid), _vp(qoff), _vp(strand), n), "hm_scan_sites")
        return rid, qoff, strand

    def windows(self, ctx: int, first: int = 0, n: Optional[int] = None, fetch: bool = True):
        """get_next_sample_features for sites [first, first+n) of context ctx -> float32 [n, 401, 8]."""
        if n is None:
            n = self.num_sites(ctx) - first
        out = np.empty((n, KMER, FEATS), np.float32) if fetch else None
        self._check(self._L.hm_windows(self._h, ctx, first, n, _vp(out)), "hm_windows")
        return out

    def cnn_logits(self, ctx: int, windows: np.ndarray):
        """ModBatch::call_current_batch on a [n, 401, 8] batch -> (logits [n,2], p [n], ml [n])."""
        w = np.ascontiguousarray(windows, np.float32).reshape(-1, KMER, FEATS)
        n = w.shape[0]
        lg, p, ml = np.empty((n, 2), np.float32), np.empty(n, np.float32), np.empty(n, np.uint8)
        self._check(self._L.hm_cnn_logits(self._h, ctx, _vp(w), n, _vp(lg), _vp(p), _vp(ml)), "hm_cnn_logits")
        return lg, p, ml

    def debug_layer(self, ctx: int, window: np.ndarray, layer: int) -> np.ndarray:
        w = np.ascontiguousarray(window, np.float32).reshape(KMER, FEATS)
        out = np.empty(197 * 128, np.float32)
        n = self._check(self._L.hm_debug_layer(self._h, ctx, _vp(w), layer, _vp(out), out.size), "hm_debug_layer")
        chans = (8, 128, 128, 128, 96, 96, 96, 64, 64)[layer]
        return out[:n].reshape(-1, chans).copy()

    def stamps(self) -> list:
        buf = (C.c_uint64 * 256)()
        n = self._check(self._L.hm_get_stamps(self._h, buf, 256), "hm_get_stamps")
        return [int(buf[i]) for i in range(n)]

    def timing(self, reset: bool = False) -> dict:
        t = _lib.hm_timing_t()
        self._check(self._L.hm_get_timing(self._h, C.byref(t)), "hm_get_timing")
        d = {k: (list(getattr(t, k)) if hasattr(getattr(t, k), "__len__") else getattr(t, k)) for k, _ in t._fields_}
        if reset:
            self._L.hm_reset_timing(self._h)
        return d


def trunk_mask_for_reads(block: "ReadBlock", ctx_mask: int = 7) -> int:
    """hm_trunk_mask_for_reads: the per-context kernel-path choice of engine option trunk = 2 for a sample of reads (host
    only).  Pass the result to set_option("trunk_mask", m) on every engine that shares one input."""
    rc = _lib.lib().hm_trunk_mask_for_reads(block.desc.ctypes.data_as(C.c_void_p), len(block.desc), ctx_mask)
    if rc < 0:
        raise HifimethError(f"hm_trunk_mask_for_reads failed ({rc})")
    return rc


class ReadBlock:
    """Descriptors (hm_read_t) of a list of reads: what a BAM decoder hands to the engine.  Keeps the arrays alive."""

    def __init__(self, reads, first_id: int = 0):
        self.reads = list(reads)
        self.desc = np.zeros(len(self.reads), READ_DTYPE)
        self._keep = []
        for i, r in enumerate(self.reads):
            d = self.desc[i]
            d["read_id"], d["l_qseq"], d["flag"] = first_id + i, r.l_qseq, r.flag
            seq4 = np.ascontiguousarray(r.seq4, np.uint8)
            d["seq4"] = seq4.ctypes.data
            self._keep.append(seq4)
            for k, nm in enumerate(("fi", "fp", "ri", "rp")):
                a = getattr(r, nm)
                if a is None or len(a) != r.l_qseq:   # bam_auxB_len != l_qseq -> init() == false (bam_info.cpp:447)
                    d["kin"][k], d["width"][k] = 0, 1
                    continue
                a = np.ascontiguousarray(a)
                if a.dtype.itemsize not in (1, 2):
                    raise ValueError("kinetics arrays must be uint8 (B:C) or uint16 (B:S)")
                d["kin"][k], d["width"][k] = a.ctypes.data, a.dtype.itemsize
                self._keep.append(a)

    def __len__(self):
        return len(self.reads)


class Batch:
    """One slot of the engine's pipeline: stage -> enqueue (returns at once) -> wait -> release."""

    def __init__(self, mc: MethylationCaller, handle):
        self._mc, self._L, self._h = mc, mc._L, C.c_void_p(handle)

    def submit(self, read_id: int, read) -> bool:
        args, _keep = MethylationCaller._read_args(read)
        return bool(self._mc._check(self._L.hm_batch_submit_read(self._h, read_id, *args), "hm_batch_submit_read"))

    def submit_all(self, reads: Iterable, first_id: int = 0) -> int:
        n = 0
        for i, r in enumerate(reads):
            n += self.submit(first_id + i, r)
        return n

    def submit_block(self, block: "ReadBlock", threads: int = 4) -> int:
        """All reads of a prepared ReadBlock in ONE call (hm_batch_submit_reads): the copies into the pinned slab run on
        `threads` host threads.  Same result as submit() read by read."""
        rc = self._L.hm_batch_submit_reads(self._h, block.desc.ctypes.data_as(C.c_void_p), len(block.desc), threads, None)
        return self._mc._check(rc, "hm_batch_submit_reads")

    def staged_bases(self) -> int:
        return self._L.hm_batch_staged_bases(self._h)

    def enqueue(self):
        self._mc._check(self._L.hm_batch_enqueue(self._h), "hm_batch_enqueue")

    def done(self) -> bool:
        return bool(self._mc._check(self._L.hm_batch_done(self._h), "hm_batch_done"))

    def num_sites(self, ctx: int = 3) -> int:
        return self._mc._check(self._L.hm_batch_num_sites(self._h, ctx), "hm_batch_num_sites")

    def wait(self) -> np.ndarray:
        """The batch's calls as a CALL_DTYPE view of the slot's pinned result buffer (valid until release())."""
        ptr = C.c_void_p()
        n = self._mc._check(self._L.hm_batch_wait(self._h, C.byref(ptr)), "hm_batch_wait")
        if n == 0:
            return np.empty(0, CALL_DTYPE)
        buf = (C.c_char * (n * CALL_DTYPE.itemsize)).from_address(ptr.value)
        return np.frombuffer(buf, CALL_DTYPE, n)

    def release(self):
        if self._h:
            self._L.hm_batch_release(self._h)
            self._h = None
