#!/usr/bin/env python3
"""BASELINE.json configs[4] at scale: which layers can run with plain fp16 WEIGHTS (the w_lo*x_hi product dropped) inside |dp| <= 1e-3?
The per-layer table of profiles/r02_term_error_table.txt was taken on 7 k windows; maxima grow with the sample (conv3 alone: 5.4e-4 there,
1.3e-3 over 6.1 M sites on the GPU -- profiles/r05_parity_sweep_precision2.txt).  This emulates sets of layers on the CPU (torch fp32
functional graph over the ONNX-extracted weights = the oracle's definition of the network; fp16-rounded weights in the chosen layers) over
several hundred thousand windows of three GC contents, in chunks.

    python tools/w16_sweep.py [reads_per_gc] [ctx]
"""
import os
import sys
import time

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from hifimeth_amd.onnx_weights import load_hmw  # noqa: E402
from hifimeth_amd.synth import synth_reads  # noqa: E402
from oracle import hm_oracle as O  # noqa: E402

VARIANTS = {"conv3": (2,), "conv7": (6,), "conv8": (7,), "fc1": (8,), "conv8+fc1": (7, 8), "conv7+conv8+fc1": (6, 7, 8), "conv6": (5,)}


def main():
    n_reads = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    ctx = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    name = ("CpG", "CHG", "CHH")[ctx]
    w = load_hmw(os.path.join(ROOT, "hifimeth_amd", "weights", name + ".hmw"))
    g, b, m, v = (torch.tensor(t) for t in (w.bn_gamma, w.bn_beta, w.bn_mean, w.bn_var))
    cw = [torch.tensor(t) for t in w.conv_w]
    cb = [torch.tensor(t) for t in w.conv_b]
    f1w, f1b, f2w, f2b = (torch.tensor(t) for t in (w.fc1_w, w.fc1_b, w.fc2_w, w.fc2_b))
    r16 = lambda t: t.half().float()  # noqa: E731
    cw16 = [r16(t) for t in cw]
    f1w16 = r16(f1w)

    def run(x0, layers=()):
        with torch.no_grad():
            h = ((x0 - m) / torch.sqrt(v + w.bn_eps) * g + b).permute(0, 2, 1)
            for i in range(8):
                h = F.relu(F.conv1d(h, cw16[i] if i in layers else cw[i], cb[i], stride=2, padding=1))
            h = F.relu(F.linear(h.flatten(1), f1w16 if 8 in layers else f1w, f1b))
            return torch.softmax(F.linear(h, f2w, f2b), 1)[:, 1].numpy()

    worst = {k: 0.0 for k in VARIANTS}
    above = {k: [0, 0] for k in VARIANTS}   # sites above 7e-4, above 1e-3
    total, t0 = 0, time.time()
    for gi, gc in enumerate((0.36, 0.41, 0.5)):
        for rd in synth_reads(n_reads, seed=5200 + gi, gc=gc, median_len=9000):
            if not rd.has_kinetics() or rd.l_qseq < 1000:
                continue
            fwd = O.decode(rd)
            wv, _ = O.windows(rd, fwd, O.scan(fwd, ctx))
            for c0 in range(0, len(wv), 2048):
                x0 = torch.from_numpy(wv[c0:c0 + 2048])
                ref = run(x0)
                for k, layers in VARIANTS.items():
                    d = np.abs(run(x0, layers) - ref)
                    worst[k] = max(worst[k], float(d.max()))
                    above[k][0] += int((d > 7e-4).sum())
                    above[k][1] += int((d > 1e-3).sum())
            total += len(wv)
        print(f"# gc {gc}: {total} windows so far, {time.time() - t0:.0f} s: " + ", ".join(f"{k} {x:.2e}" for k, x in worst.items()), flush=True)
    print(f"{name}: {total} windows, three GC contents; fp16-rounded WEIGHTS in the named layers, everything else fp32 (CPU emulation)")
    print(f"{'layers':<20}{'max |dp|':>12}{'sites > 7e-4':>14}{'sites > 1e-3':>14}")
    for k in VARIANTS:
        print(f"{k:<20}{worst[k]:>12.2e}{above[k][0]:>14d}{above[k][1]:>14d}")


if __name__ == "__main__":
    main()
