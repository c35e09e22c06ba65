#!/usr/bin/env python3
"""Convert the reference's shipped ONNX models to the flat .hmw container.

usage: python tools/onnx_to_hmw.py /root/reference/models hifimeth_amd/weights
(reference: models/{CpG,CHG,CHH}.onnx, loaded at src/app/hifimeth/mod_main.cpp:76,85,94)
"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from hifimeth_amd.onnx_weights import load_onnx, save_hmw  # noqa: E402


def main():
    src, dst = sys.argv[1], sys.argv[2]
    os.makedirs(dst, exist_ok=True)
    for ctx in ("CpG", "CHG", "CHH"):
        w = load_onnx(os.path.join(src, ctx + ".onnx"))
        out = os.path.join(dst, ctx + ".hmw")
        save_hmw(w, out)
        print(f"{ctx}: k1={w.k1} params={w.n_params()} MAC/site={w.macs_per_site()} -> {out}")


if __name__ == "__main__":
    main()
