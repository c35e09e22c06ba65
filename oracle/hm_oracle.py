"""ctypes binding of the CPU oracle (oracle/libhm_oracle.so).

TEST INFRASTRUCTURE ONLY -- importable from tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg; never from hifimeth_amd/ (the product path).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libhm_oracle.so")
_LIB_AVX512 = os.path.join(_HERE, "libhm_oracle_avx512.so")   # same source, -march=x86-64-v4: ~2x faster, same results
REF_SCAN = os.path.join(_HERE, "_ref", "ref_scan")

KMER, FEATS = 401, 8
CTX_NAMES = ("CpG", "CHG", "CHH")


class _Read(C.Structure):
    _fields_ = [("l_qseq", C.c_int32), ("flag", C.c_int32), ("seq4", C.c_void_p),
                ("fi", C.c_void_p), ("fp", C.c_void_p), ("ri", C.c_void_p), ("rp", C.c_void_p),
                ("fi_w", C.c_int32), ("fp_w", C.c_int32), ("ri_w", C.c_int32), ("rp_w", C.c_int32)]


def build(force: bool = False) -> None:
    if force or not os.path.exists(_LIB_PATH) or not os.path.exists(_LIB_AVX512):
        subprocess.check_call(["make", "-C", _HERE, "libhm_oracle.so", "libhm_oracle_avx512.so"], stdout=subprocess.DEVNULL)


def _host_has_avx512() -> bool:
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("flags"):
                    fl = line.split()
                    return all(x in fl for x in ("avx512f", "avx512bw", "avx512dq", "avx512vl", "avx512cd"))
    except OSError:
        pass
    return False


_lib = None
variant = "avx2"      # which build lib() loaded: "avx2" (x86-64-v3) or "avx512" (x86-64-v4)


def lib():
    global _lib
    if _lib is None:
        build()
        global variant
        use512 = _host_has_avx512() and os.path.exists(_LIB_AVX512) and not os.environ.get("HM_ORACLE_NO_AVX512")
        variant = "avx512" if use512 else "avx2"
        L = C.CDLL(_LIB_AVX512 if use512 else _LIB_PATH)
        L.hmo_decode_read.argtypes = [C.POINTER(_Read), C.c_char_p]
        L.hmo_scan.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_void_p]
        L.hmo_window.argtypes = [C.POINTER(_Read), C.c_char_p, C.c_int, C.c_void_p, C.POINTER(C.c_int)]
        L.hmo_model_load.restype = C.c_void_p
        L.hmo_model_load.argtypes = [C.c_char_p]
        L.hmo_model_free.argtypes = [C.c_void_p]
        L.hmo_model_k1.argtypes = [C.c_void_p]
        L.hmo_cnn_logits.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int]
        L.hmo_cnn_layer.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        L.hmo_softmax.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        L.hmo_codev1_table.argtypes = [C.c_void_p]
        L.hmo_encode_frames.argtypes = [C.c_int]
        L.hmo_call_read.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.POINTER(_Read), C.c_int, C.c_int,
                                    C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        _lib = L
    return _lib


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def _cread(rd):
    """rd: any object with l_qseq, flag, seq4, fi, fp, ri, rp numpy arrays (hifimeth_amd.synth.Read)."""
    keep = [np.ascontiguousarray(x) for x in (rd.seq4, rd.fi, rd.fp, rd.ri, rd.rp)]
    r = _Read(rd.l_qseq, rd.flag, *[a.ctypes.data for a in keep], *[a.dtype.itemsize for a in keep[1:]])
    return r, keep


def decode(rd) -> bytes:
    seq4 = np.ascontiguousarray(rd.seq4)
    r = _Read(rd.l_qseq, rd.flag, seq4.ctypes.data, None, None, None, None, 1, 1, 1, 1)
    buf = C.create_string_buffer(rd.l_qseq + 1)
    if lib().hmo_decode_read(C.byref(r), buf) != 0:
        raise ValueError("illegal BAM base nibble")
    return buf.raw[:rd.l_qseq]


def scan(fwd: bytes, ctx: int) -> np.ndarray:
    out = np.empty(len(fwd) + 1, np.int32)
    n = lib().hmo_scan(fwd, len(fwd), ctx, _ptr(out))
    return out[:n].copy()


def window(rd, fwd: bytes, qoff: int):
    w, s = windows(rd, fwd, [qoff])
    return w[0], int(s[0])


def windows(rd, fwd: bytes, offs) -> tuple:
    r, _k = _cread(rd)
    out = np.empty((len(offs), KMER, FEATS), np.float32)
    strands = np.empty(len(offs), np.uint8)
    s = C.c_int()
    for i, o in enumerate(offs):
        lib().hmo_window(C.byref(r), fwd, int(o), C.c_void_p(out[i].ctypes.data), C.byref(s))
        strands[i] = s.value
    return out, strands


def codev1_table() -> np.ndarray:
    t = np.empty(256, np.int32)
    lib().hmo_codev1_table(_ptr(t))
    return t


def encode_frames(s: int) -> int:
    return lib().hmo_encode_frames(int(s))


class Model:
    def __init__(self, hmw_path: str):
        self._h = lib().hmo_model_load(hmw_path.encode())
        if not self._h:
            raise IOError(f"cannot load {hmw_path}")
        self.k1 = lib().hmo_model_k1(self._h)

    def __del__(self):
        if getattr(self, "_h", None):
            lib().hmo_model_free(self._h)
            self._h = None

    def logits(self, win: np.ndarray, nthreads: int = 0) -> np.ndarray:
        win = np.ascontiguousarray(win, np.float32).reshape(-1, KMER, FEATS)
        out = np.empty((win.shape[0], 2), np.float32)
        lib().hmo_cnn_logits(self._h, _ptr(win), win.shape[0], _ptr(out), nthreads or (os.cpu_count() or 1))
        return out

    def layer(self, win: np.ndarray, layer: int) -> np.ndarray:
        """Post-ReLU channels-last output [Lout, Cout] of conv `layer` (1..8) for one window."""
        win = np.ascontiguousarray(win, np.float32).reshape(KMER, FEATS)
        out = np.empty(197 * 128, np.float32)
        n = lib().hmo_cnn_layer(self._h, _ptr(win), layer, _ptr(out))
        chans = (8, 128, 128, 128, 96, 96, 96, 64, 64)[layer]
        return out[:n].reshape(-1, chans).copy()


def softmax(logits: np.ndarray):
    logits = np.ascontiguousarray(logits, np.float32).reshape(-1, 2)
    p = np.empty(logits.shape[0], np.float32)
    ml = np.empty(logits.shape[0], np.uint8)
    lib().hmo_softmax(_ptr(logits), logits.shape[0], _ptr(p), _ptr(ml))
    return p, ml


def call_read(models, ctx_mask: int, rd, min_len: int = 1000, nthreads: int = 0):
    """models: indexable by ctx id -> Model (unused entries may be None). Returns dict of arrays in
    the reference's emission order (empty when the read is skipped)."""
    r, _k = _cread(rd)
    cap = 2 * rd.l_qseq + 8
    qoff = np.empty(cap, np.int32)
    strand = np.empty(cap, np.uint8)
    ctx = np.empty(cap, np.uint8)
    p = np.empty(cap, np.float32)
    ml = np.empty(cap, np.uint8)
    arr = (C.c_void_p * 3)(*[(models[c]._h if (ctx_mask >> c & 1) else None) for c in range(3)])
    n = lib().hmo_call_read(arr, ctx_mask, C.byref(r), min_len, cap, _ptr(qoff), _ptr(strand), _ptr(ctx),
                            _ptr(p), _ptr(ml), nthreads or (os.cpu_count() or 1))
    if n < 0:
        raise RuntimeError("hmo_call_read failed")
    return dict(qoff=qoff[:n].copy(), strand=strand[:n].copy(), ctx=ctx[:n].copy(), p=p[:n].copy(), ml=ml[:n].copy())


def ref_scan_available() -> bool:
    return os.access(REF_SCAN, os.X_OK)


def ref_scan(records):
    """Run the REFERENCE's scanner binary (oracle/_ref/ref_scan) on [(flag, seq_ascii_as_stored)].
    Returns a list of dicts {fwd, cpg, chg, chh}."""
    inp = "".join(f"{f} {s}\n" for f, s in records)
    out = subprocess.run([REF_SCAN], input=inp.encode(), stdout=subprocess.PIPE, check=True).stdout.decode().split("\n")
    res = []
    for i in range(len(records)):
        d = {"fwd": out[4 * i][4:]}
        for j, key in enumerate(("cpg", "chg", "chh")):
            tok = out[4 * i + 1 + j].split()
            assert tok[0] == key
            d[key] = [int(x) for x in tok[2:]]
            assert len(d[key]) == int(tok[1])
        res.append(d)
    return res
