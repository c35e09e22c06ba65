"""Minimal ONNX (protobuf wire-format) reader for the three shipped 5mC models.

The reference loads `<model_dir>/{CpG,CHG,CHH}.onnx` through OpenVINO
(reference: src/app/hifimeth/mod_main.cpp:32-98).  Neither `onnx` nor OpenVINO
exist in this image, so the graph is read straight from the protobuf wire
format.  Two dialects ship:

* CpG/CHG: opset 17, weights are graph *initializers*, FC = Gemm(transB=1)
  with [out,in] weights.
* CHH: opset 11, weights are `Constant` *nodes*, FC = MatMul([in,out]) + Add.

Both are normalised into one canonical tensor list (`CANONICAL_ORDER`) and can
be written to the flat `.hmw` container the HIP engine and the C oracle read.

`.hmw` layout (little endian):
    char[4]  magic "HMW1"
    int32    kmer (401), int32 features (8), int32 k1 (first conv kernel),
    int32    n_conv (8)
    int32[9] channels  (8,128,128,128,96,96,96,64,64)
    int32[8] kernel sizes
    float    bn_eps
    then fp32 tensors: bn0 gamma[8] beta[8] mean[8] var[8];
    per conv i: W[Cout][Cin][k] (ONNX OIW order), bias[Cout];
    fc1 W[256][128] ([out][in]), b[256]; fc2 W[2][256], b[2].
"""
from __future__ import annotations

import struct
from dataclasses import dataclass, field
from typing import Dict, List, Tuple

import numpy as np

KMER = 401
FEATURES = 8
CHANNELS = (8, 128, 128, 128, 96, 96, 96, 64, 64)
FC1_OUT = 256
N_CLASSES = 2


# ----------------------------------------------------------------------------
# protobuf wire format
# ----------------------------------------------------------------------------
def _varint(buf: bytes, pos: int) -> Tuple[int, int]:
    out = 0
    shift = 0
    while True:
        b = buf[pos]
        pos += 1
        out |= (b & 0x7F) << shift
        if not b & 0x80:
            return out, pos
        shift += 7


def _fields(buf: bytes):
    """Yield (field_number, wire_type, value) for one message."""
    pos = 0
    n = len(buf)
    while pos < n:
        key, pos = _varint(buf, pos)
        fno, wt = key >> 3, key & 7
        if wt == 0:
            val, pos = _varint(buf, pos)
        elif wt == 1:
            val = buf[pos:pos + 8]
            pos += 8
        elif wt == 2:
            ln, pos = _varint(buf, pos)
            val = buf[pos:pos + ln]
            pos += ln
        elif wt == 5:
            val = buf[pos:pos + 4]
            pos += 4
        else:
            raise ValueError(f"unsupported protobuf wire type {wt}")
        yield fno, wt, val


def _packed_ints(val, wt) -> List[int]:
    if wt == 0:
        return [val]
    out = []
    pos = 0
    while pos < len(val):
        v, pos = _varint(val, pos)
        out.append(v)
    return out


def _tensor(buf: bytes) -> Tuple[str, np.ndarray]:
    dims: List[int] = []
    dtype = 0
    name = ""
    raw = None
    floats: List[float] = []
    for fno, wt, val in _fields(buf):
        if fno == 1:
            dims += _packed_ints(val, wt)
        elif fno == 2:
            dtype = val
        elif fno == 4:
            if wt == 2:
                floats += list(np.frombuffer(val, dtype="<f4"))
            else:
                floats.append(struct.unpack("<f", val)[0])
        elif fno == 8:
            name = val.decode()
        elif fno == 9:
            raw = val
    if dtype != 1:
        return name, None  # only fp32 tensors matter here
    arr = np.frombuffer(raw, dtype="<f4") if raw is not None else np.asarray(floats, dtype=np.float32)
    return name, arr.reshape(dims).astype(np.float32)


@dataclass
class _Node:
    op: str = ""
    inputs: List[str] = field(default_factory=list)
    outputs: List[str] = field(default_factory=list)
    ints: Dict[str, List[int]] = field(default_factory=dict)
    floats: Dict[str, float] = field(default_factory=dict)
    tensor: np.ndarray = None


def _node(buf: bytes) -> _Node:
    nd = _Node()
    for fno, wt, val in _fields(buf):
        if fno == 1:
            nd.inputs.append(val.decode())
        elif fno == 2:
            nd.outputs.append(val.decode())
        elif fno == 4:
            nd.op = val.decode()
        elif fno == 5:
            aname = ""
            for f2, w2, v2 in _fields(val):
                if f2 == 1:
                    aname = v2.decode()
                elif f2 == 2:
                    nd.floats[aname] = struct.unpack("<f", v2)[0]
                elif f2 == 3:
                    nd.ints.setdefault(aname, []).append(v2)
                elif f2 == 5:
                    nd.tensor = _tensor(v2)[1]
                elif f2 == 8:
                    nd.ints.setdefault(aname, []).extend(_packed_ints(v2, w2))
    return nd


# ----------------------------------------------------------------------------
# canonical model
# ----------------------------------------------------------------------------
@dataclass
class ModelWeights:
    """Canonical fp32 parameters of one context model (BN folded into convs, bn0 explicit)."""
    k1: int
    bn_eps: float
    bn_gamma: np.ndarray
    bn_beta: np.ndarray
    bn_mean: np.ndarray
    bn_var: np.ndarray
    conv_w: List[np.ndarray]  # [Cout, Cin, k]
    conv_b: List[np.ndarray]
    fc1_w: np.ndarray  # [256, 128]  ([out, in])
    fc1_b: np.ndarray
    fc2_w: np.ndarray  # [2, 256]
    fc2_b: np.ndarray

    @property
    def kernels(self) -> Tuple[int, ...]:
        return tuple(int(w.shape[2]) for w in self.conv_w)

    def n_params(self) -> int:
        n = 4 * 8
        n += sum(w.size + b.size for w, b in zip(self.conv_w, self.conv_b))
        return n + self.fc1_w.size + self.fc1_b.size + self.fc2_w.size + self.fc2_b.size

    def macs_per_site(self) -> int:
        L = KMER
        macs = 0
        for w in self.conv_w:
            co, ci, k = w.shape
            L = (L + 2 - k) // 2 + 1
            macs += L * co * ci * k
        return macs + self.fc1_w.size + self.fc2_w.size


def load_onnx(path: str) -> ModelWeights:
    buf = open(path, "rb").read()
    graph = None
    for fno, _, val in _fields(buf):
        if fno == 7:
            graph = val
    if graph is None:
        raise ValueError(f"{path}: no GraphProto")
    inits: Dict[str, np.ndarray] = {}
    nodes: List[_Node] = []
    for fno, _, val in _fields(graph):
        if fno == 1:
            nodes.append(_node(val))
        elif fno == 5:
            name, arr = _tensor(val)
            if arr is not None:
                inits[name] = arr
    for nd in nodes:  # CHH dialect: weights live in Constant nodes
        if nd.op == "Constant" and nd.tensor is not None:
            inits[nd.outputs[0]] = nd.tensor

    bn = [n for n in nodes if n.op == "BatchNormalization"]
    convs = [n for n in nodes if n.op == "Conv"]
    if len(bn) != 1 or len(convs) != 8:
        raise ValueError(f"{path}: expected 1 BatchNormalization + 8 Conv, got {len(bn)} + {len(convs)}")
    g, b, m, v = (inits[x] for x in bn[0].inputs[1:5])
    eps = bn[0].floats.get("epsilon", 1e-5)
    cw, cb = [], []
    for i, c in enumerate(convs):
        w = inits[c.inputs[1]]
        bias = inits[c.inputs[2]] if len(c.inputs) > 2 else np.zeros(w.shape[0], np.float32)
        if c.ints.get("strides", [1]) != [2] or c.ints.get("pads", [0, 0]) != [1, 1]:
            raise ValueError(f"{path}: conv {i} is not stride 2 / pad 1")
        if c.ints.get("dilations", [1]) != [1] or c.ints.get("group", [1]) != [1]:
            raise ValueError(f"{path}: conv {i} has dilation/groups")
        if tuple(w.shape[:2]) != (CHANNELS[i + 1], CHANNELS[i]):
            raise ValueError(f"{path}: conv {i} weight shape {w.shape}")
        cw.append(np.ascontiguousarray(w))
        cb.append(np.ascontiguousarray(bias))

    fcs = []  # (W [out,in], b)
    gemms = [n for n in nodes if n.op == "Gemm"]
    if gemms:
        for gm in gemms:
            w = inits[gm.inputs[1]]
            if gm.ints.get("transB", [0]) != [1]:
                w = w.T
            if gm.floats.get("alpha", 1.0) != 1.0 or gm.floats.get("beta", 1.0) != 1.0:
                raise ValueError(f"{path}: Gemm alpha/beta != 1")
            fcs.append((np.ascontiguousarray(w), inits[gm.inputs[2]]))
    else:
        mms = [n for n in nodes if n.op == "MatMul"]
        adds = [n for n in nodes if n.op == "Add"]
        for mm in mms:
            w = inits[mm.inputs[1]]  # [in, out]
            add = next(a for a in adds if mm.outputs[0] in a.inputs)
            bias_name = [x for x in add.inputs if x != mm.outputs[0]][0]
            fcs.append((np.ascontiguousarray(w.T), inits[bias_name]))
    if len(fcs) != 2 or fcs[0][0].shape != (FC1_OUT, 128) or fcs[1][0].shape != (N_CLASSES, FC1_OUT):
        raise ValueError(f"{path}: unexpected FC stack {[f[0].shape for f in fcs]}")
    return ModelWeights(int(cw[0].shape[2]), float(eps), g, b, m, v, cw, cb,
                        fcs[0][0], fcs[0][1], fcs[1][0], fcs[1][1])


_MAGIC = b"HMW1"


def save_hmw(w: ModelWeights, path: str) -> None:
    with open(path, "wb") as f:
        f.write(_MAGIC)
        f.write(struct.pack("<4i", KMER, FEATURES, w.k1, 8))
        f.write(struct.pack("<9i", *CHANNELS))
        f.write(struct.pack("<8i", *w.kernels))
        f.write(struct.pack("<f", w.bn_eps))
        for t in (w.bn_gamma, w.bn_beta, w.bn_mean, w.bn_var):
            f.write(np.ascontiguousarray(t, "<f4").tobytes())
        for cw, cb in zip(w.conv_w, w.conv_b):
            f.write(np.ascontiguousarray(cw, "<f4").tobytes())
            f.write(np.ascontiguousarray(cb, "<f4").tobytes())
        for t in (w.fc1_w, w.fc1_b, w.fc2_w, w.fc2_b):
            f.write(np.ascontiguousarray(t, "<f4").tobytes())


def load_hmw(path: str) -> ModelWeights:
    buf = open(path, "rb").read()
    if buf[:4] != _MAGIC:
        raise ValueError(f"{path}: not an HMW1 file")
    pos = 4
    kmer, feats, k1, nconv = struct.unpack_from("<4i", buf, pos); pos += 16
    chans = struct.unpack_from("<9i", buf, pos); pos += 36
    kern = struct.unpack_from("<8i", buf, pos); pos += 32
    (eps,) = struct.unpack_from("<f", buf, pos); pos += 4
    if (kmer, feats, nconv) != (KMER, FEATURES, 8) or tuple(chans) != CHANNELS or kern[0] != k1:
        raise ValueError(f"{path}: unsupported geometry")

    def take(shape):
        nonlocal pos
        n = int(np.prod(shape))
        a = np.frombuffer(buf, "<f4", n, pos).reshape(shape).copy()
        pos += 4 * n
        return a

    g, b, m, v = take((8,)), take((8,)), take((8,)), take((8,))
    cw, cb = [], []
    for i in range(8):
        cw.append(take((chans[i + 1], chans[i], kern[i])))
        cb.append(take((chans[i + 1],)))
    f1w, f1b = take((FC1_OUT, 128)), take((FC1_OUT,))
    f2w, f2b = take((N_CLASSES, FC1_OUT)), take((N_CLASSES,))
    if pos != len(buf):
        raise ValueError(f"{path}: trailing bytes")
    return ModelWeights(k1, eps, g, b, m, v, cw, cb, f1w, f1b, f2w, f2b)
