#!/usr/bin/env python3
"""Which of the three split-half products could be dropped, layer by layer?  

The GPU computes  w*x ~= w_hi*x_hi + w_hi*x_lo + w_lo*x_hi  per layer.  Dropping `w_lo*x_hi` in a layer is the same as
rounding that layer's WEIGHTS to fp16; dropping `w_hi*x_lo` is the same as rounding its INPUT ACTIVATIONS to fp16.  This
tool emulates both on the CPU (torch fp32 functional graph over the ONNX-extracted weights, the oracle's definition of the
network), one layer and one term at a time, and prints max / mean |dp| against the all-fp32 result: a term that costs less
than ~2e-5 would be free throughput.  None is.

    python tools/term_error_table.py [n_reads] [ctx]
"""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from hifimeth_amd.onnx_weights import load_hmw  # noqa: E402
from hifimeth_amd.synth import synth_reads  # noqa: E402
from oracle import hm_oracle as O  # noqa: E402


def main():
    n_reads = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    ctx = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    name = ("CpG", "CHG", "CHH")[ctx]
    w = load_hmw(os.path.join(ROOT, "hifimeth_amd", "weights", name + ".hmw"))
    wins = []
    for rd in synth_reads(n_reads, seed=4242, gc=0.36, median_len=6000):
        if not rd.has_kinetics() or rd.l_qseq < 1000:
            continue
        fwd = O.decode(rd)
        wv, _ = O.windows(rd, fwd, O.scan(fwd, ctx))
        wins.append(wv)
    x0 = torch.from_numpy(np.concatenate(wins))
    g, b, m, v = (torch.tensor(t) for t in (w.bn_gamma, w.bn_beta, w.bn_mean, w.bn_var))
    cw = [torch.tensor(t) for t in w.conv_w]
    cb = [torch.tensor(t) for t in w.conv_b]
    f1w, f1b, f2w, f2b = (torch.tensor(t) for t in (w.fc1_w, w.fc1_b, w.fc2_w, w.fc2_b))
    r16 = lambda t: t.half().float()  # noqa: E731

    def run(w16=-1, x16=-1):
        """layer index 0..7 = conv1..conv8, 8 = fc1"""
        with torch.no_grad():
            h = ((x0 - m) / torch.sqrt(v + w.bn_eps) * g + b).permute(0, 2, 1)
            for i in range(8):
                hin = r16(h) if x16 == i else h
                h = F.relu(F.conv1d(hin, r16(cw[i]) if w16 == i else cw[i], cb[i], stride=2, padding=1))
            h = h.flatten(1)
            h = F.relu(F.linear(r16(h) if x16 == 8 else h, r16(f1w) if w16 == 8 else f1w, f1b))
            lg = F.linear(h, f2w, f2b)
            return torch.softmax(lg, 1)[:, 1].numpy()

    ref = run()
    names = [f"conv{i + 1}" for i in range(8)] + ["fc1"]
    print(f"{name}: {len(ref)} windows (p in (0.05, 0.95): {100 * ((ref > 0.05) & (ref < 0.95)).mean():.0f} %)")
    print(f"{'layer':<8}{'drop w_lo*x_hi: max |dp|':>26}{'mean':>10}{'drop w_hi*x_lo: max |dp|':>28}{'mean':>10}")
    for i, nm in enumerate(names):
        dw = np.abs(run(w16=i) - ref)
        dx = np.abs(run(x16=i) - ref)
        note = "   (conv1's operand is exact fp16 on the GPU: no x_lo term exists there)" if i == 0 else ""
        print(f"{nm:<8}{dw.max():>26.2e}{dw.mean():>10.2e}{dx.max():>28.2e}{dx.mean():>10.2e}{note}")


if __name__ == "__main__":
    main()
