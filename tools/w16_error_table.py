#!/usr/bin/env python3
"""Per-layer error table for BASELINE.json configs[4] (fp16 CNN weights, |dp| <= 1e-3 vs the CPU fp32 path).

Runs on the CPU with the oracle only (test infrastructure): the weights of a chosen set of layers are rounded to
fp16 (what dropping the w_lo MFMA pass does on the GPU) and the oracle is evaluated in fp32 on the windows of synthetic
reads.  Prints max / mean |dp| and the fraction above 1e-4 / 1e-3 for every single layer and for the cumulative sets
"round layers >= k", which is what decides where the GPU's precision=2 mode may drop the pass.

    python tools/w16_error_table.py [n_reads] [ctx]
"""
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from hifimeth_amd.onnx_weights import load_hmw, save_hmw  # noqa: E402
from hifimeth_amd.synth import synth_reads  # noqa: E402
from oracle import hm_oracle as O  # noqa: E402

LAYERS = ["conv1", "conv2", "conv3", "conv4", "conv5", "conv6", "conv7", "conv8", "fc1", "fc2"]


def rounded(w, which):
    import copy
    m = copy.deepcopy(w)
    r16 = lambda a: a.astype(np.float16).astype(np.float32)  # noqa: E731
    for i in which:
        if i < 8:
            m.conv_w[i] = r16(m.conv_w[i])
        elif i == 8:
            m.fc1_w = r16(m.fc1_w)
        else:
            m.fc2_w = r16(m.fc2_w)
    return m


def main():
    n_reads = int(sys.argv[1]) if len(sys.argv) > 1 else 12
    ctx = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    name = ("CpG", "CHG", "CHH")[ctx]
    w = load_hmw(os.path.join(ROOT, "hifimeth_amd", "weights", name + ".hmw"))
    reads = [r for r in synth_reads(n_reads, seed=4242, gc=0.36, median_len=6000) if r.has_kinetics() and r.l_qseq >= 1000]
    wins = []
    for rd in reads:
        fwd = O.decode(rd)
        offs = O.scan(fwd, ctx)
        wv, _ = O.windows(rd, fwd, offs)
        wins.append(wv)
    wins = np.concatenate(wins)
    tmp = tempfile.mkdtemp()

    def probs(which):
        path = os.path.join(tmp, "m.hmw")
        save_hmw(rounded(w, which), path)
        p, _ = O.softmax(O.Model(path).logits(wins))
        return p

    ref = probs([])
    print(f"{name}: {len(wins)} windows; p in (0.05, 0.95): {100 * ((ref > 0.05) & (ref < 0.95)).mean():.1f} %")
    print(f"{'fp16 weights in':<28}{'max |dp|':>10}{'mean':>10}{'>1e-4 %':>9}{'>1e-3 %':>9}")

    def row(label, which):
        d = np.abs(probs(which) - ref)
        print(f"{label:<28}{d.max():>10.2e}{d.mean():>10.2e}{100 * (d > 1e-4).mean():>9.2f}{100 * (d > 1e-3).mean():>9.3f}")

    for i, nm in enumerate(LAYERS[:9]):
        row(nm + " only", [i])
    for k in range(1, 9):
        row(f"{LAYERS[k]}..fc1", list(range(k, 9)))
    row("conv2..conv8 (not fc1)", list(range(1, 8)))
    row("conv3..conv8 (not fc1)", list(range(2, 8)))
    row("conv4..conv8 (not fc1)", list(range(3, 8)))
    row("conv5..conv8 (not fc1)", list(range(4, 8)))


if __name__ == "__main__":
    main()
