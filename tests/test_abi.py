"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports every symbol that
include/hifimeth_hip.h declares; no compute call is made (no GPU here)."""
import ctypes
import os
import re

import numpy as np
import pytest

from conftest import ROOT


def _declared():
    src = open(os.path.join(ROOT, "include", "hifimeth_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(hm_[a-z_0-9]+)\s*\(", src)))


def test_header_symbols_exported():
    from hifimeth_amd import _lib
    L = _lib.lib()
    names = _declared()
    assert len(names) >= 18
    for n in names:
        assert hasattr(L, n), f"{n} declared in include/hifimeth_hip.h but not exported"
    assert set(names) == set(L._hm_symbols), set(names) ^ set(L._hm_symbols)


def test_struct_layout_matches_header():
    from hifimeth_amd import _lib
    from hifimeth_amd.caller import CALL_DTYPE
    assert ctypes.sizeof(_lib.hm_call_t) == 16 and CALL_DTYPE.itemsize == 16
    assert CALL_DTYPE.fields["p"][1] == 12 and CALL_DTYPE.fields["scaled_prob"][1] == 10


def test_timing_struct_matches_header(tmp_path):
    """hm_timing_t grows round by round (round 4: trunk_list_steps, trunk_const_steps, group_bases, group_bytes; round 5: tail_strip_passes): the
    ctypes mirror must keep the header's size and the offsets of its last fields -- checked against what the C compiler makes of
    include/hifimeth_hip.h -- and the LIBRARY reports the header version and struct size it was built with (a consumer built
    against another header must see the mismatch instead of having hm_get_timing write past its struct)."""
    import subprocess
    from hifimeth_amd import _lib
    src = tmp_path / "t.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "hifimeth_hip.h"\n'
                   'int main(void) { printf("%zu %zu %zu %zu %zu %d\\n", sizeof(hm_timing_t), offsetof(hm_timing_t, trunk_list_steps), '
                   'offsetof(hm_timing_t, group_bases), offsetof(hm_timing_t, group_bytes), offsetof(hm_timing_t, tail_strip_passes), HM_ABI_VERSION); return 0; }\n')
    exe = tmp_path / "t"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    size, o1, o2, o3, o4, ver = (int(x) for x in subprocess.check_output([str(exe)], text=True).split())
    T = _lib.hm_timing_t
    assert (size, o1, o2, o3, o4) == (ctypes.sizeof(T), T.trunk_list_steps.offset, T.group_bases.offset, T.group_bytes.offset, T.tail_strip_passes.offset)
    L = _lib.lib()
    assert L.hm_abi_version() == ver == _lib.HM_ABI_VERSION and L.hm_timing_size() == size


def test_no_cpu_fallback():
    """Without a GPU the product must fail loudly, never fall back to a CPU path."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from hifimeth_amd import HifimethError, MethylationCaller
    with pytest.raises(HifimethError):
        MethylationCaller()


def test_product_does_not_import_oracle():
    pkg = os.path.join(ROOT, "hifimeth_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h")):
                txt = open(os.path.join(dirpath, f)).read()
                assert "hm_oracle" not in txt and "oracle/" not in txt.replace("the oracle", ""), f


def test_parse_contexts():
    from hifimeth_amd import parse_contexts
    assert parse_contexts("cpg,chg,chh") == 7 and parse_contexts("CpG") == 1 and parse_contexts("chh,chg") == 6
    with pytest.raises(ValueError):
        parse_contexts("cpg,foo")


def test_onnx_reader_roundtrip(tmp_path):
    """Python ONNX/.hmw reader: .hmw round trip is lossless and geometry matches SURVEY.md 8(a9)."""
    from hifimeth_amd.onnx_weights import load_hmw, save_hmw
    w = load_hmw(os.path.join(ROOT, "hifimeth_amd", "weights", "CHH.hmw"))
    assert w.k1 == 13 and w.n_params() == 268866 and w.macs_per_site() == 11440640
    p = str(tmp_path / "x.hmw")
    save_hmw(w, p)
    assert open(p, "rb").read() == open(os.path.join(ROOT, "hifimeth_amd", "weights", "CHH.hmw"), "rb").read()
    w = load_hmw(os.path.join(ROOT, "hifimeth_amd", "weights", "CpG.hmw"))
    assert w.k1 == 11 and w.n_params() == 266818 and w.macs_per_site() == 11148800


def test_synth_density():
    from hifimeth_amd.synth import expected_sites_per_base, synth_reads
    from oracle import hm_oracle as O
    reads = synth_reads(6, seed=5, gc=0.36, frac_missing=0, frac_short=0)
    bases = sum(r.l_qseq for r in reads)
    sites = 0
    for r in reads:
        f = O.decode(r)
        sites += sum(len(O.scan(f, c)) for c in range(3))
    assert abs(sites / bases - expected_sites_per_base(0.36)) < 0.02


def test_cxx_onnx_reader_matches_python_reader(tmp_path):
    """The C++ model reader of the product (both ONNX dialects) against the committed .hmw files,
    which the Python reader produced from the same ONNX files.  Needs /root/reference (build container)."""
    ref = "/root/reference/models"
    if not os.path.isdir(ref):
        pytest.skip("reference models absent")
    from hifimeth_amd import _lib
    L = _lib.lib()
    for ctx in ("CpG", "CHG", "CHH"):
        out = str(tmp_path / (ctx + ".hmw"))
        assert L.hm_convert_model(os.path.join(ref, ctx + ".onnx").encode(), out.encode()) == 0, L.hm_last_error(None)
        assert open(out, "rb").read() == open(os.path.join(ROOT, "hifimeth_amd", "weights", ctx + ".hmw"), "rb").read()
    assert L.hm_convert_model(b"/nonexistent.onnx", b"/tmp/x.hmw") < 0
    assert b"cannot read" in L.hm_last_error(None)


def test_cxx_hmw_roundtrip_and_rejects_garbage(tmp_path):
    from hifimeth_amd import _lib
    L = _lib.lib()
    src = os.path.join(ROOT, "hifimeth_amd", "weights", "CHG.hmw")
    out = str(tmp_path / "o.hmw")
    assert L.hm_convert_model(src.encode(), out.encode()) == 0
    assert open(out, "rb").read() == open(src, "rb").read()
    bad = tmp_path / "bad.hmw"
    bad.write_bytes(open(src, "rb").read()[:1000])
    assert L.hm_convert_model(str(bad).encode(), out.encode()) < 0
    junk = tmp_path / "junk.onnx"
    junk.write_bytes(os.urandom(4096))
    assert L.hm_convert_model(str(junk).encode(), out.encode()) < 0


def test_trunk_mask_for_reads_is_a_host_function_of_the_sample():
    """hm_trunk_mask_for_reads (no device needed): which contexts take the dense trunk follows from the site density of the
    sample alone -- CpG-poor reads send CpG / CHG to the per-site kernels, CHH stays dense; ctx_mask masks the answer; the
    same sample always gives the same answer (the CLI passes the head of the file to every rank)."""
    from hifimeth_amd.caller import ReadBlock, trunk_mask_for_reads
    from hifimeth_amd.synth import expected_sites_per_base, synth_reads
    rich = ReadBlock(synth_reads(8, seed=3, gc=0.36, median_len=3000, frac_missing=0, frac_short=0))
    poor = ReadBlock(synth_reads(8, seed=4, gc=0.11, median_len=3000, frac_missing=0, frac_short=0))
    assert expected_sites_per_base(0.36) > 0.25
    assert trunk_mask_for_reads(rich) == 7 and trunk_mask_for_reads(rich, 1) == 1 and trunk_mask_for_reads(rich, 6) == 6
    assert trunk_mask_for_reads(poor) == 4 and trunk_mask_for_reads(poor, 3) == 0   # (0.055)^2 = 0.3 % CpG; CHH ~ 9 %
    assert trunk_mask_for_reads(ReadBlock([])) == 0
    assert trunk_mask_for_reads(poor) == trunk_mask_for_reads(ReadBlock(poor.reads))


def test_bench_gpus_flag_launches_or_fails_loudly():
    """`python bench.py --gpus N` is never silently a one-rank run: without N devices (this container has none) and
    without HM_DIST_BACKEND=gloo it exits non-zero BEFORE anything touches the GPU, and says why"""
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "HM_DIST_BACKEND")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "1"], capture_output=True, text=True,
                       env=env, timeout=300)
    import torch
    if torch.cuda.device_count() < 8:
        assert r.returncode != 0 and "--gpus 8" in r.stderr and "{" not in r.stdout
