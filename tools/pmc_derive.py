"""Derived per-kernel figures from a rocprofv3 --pmc summary written by tools/prof_r05.sh (prof_r04.sh) (pmc_summary.txt):
   python tools/pmc_derive.py gpurun_out/prof_r04/pmc_summary.txt [bench.json of a PMC pass] [traffic.json to write] [commit] > profiles/r04_pmc_derived.txt
With the optional arguments the HBM-side bytes of the trunk kernel are also written per VIEW POSITION (bytes per launch / the
positions a launch of that context covered in the profiled run): what bench.py gives as roofline.traffic_quoted.
GRBM_GUI_ACTIVE sums the 8 XCDs; the SQ counters sum all 1024 SIMDs -> MFMA busy = MFMA_BUSY / (GUI_ACTIVE / 8 * 1024).
FETCH_SIZE / WRITE_SIZE are in KB; FETCH_SIZE is doubled (gfx950 counts 128-byte requests as 64 B: MI355X_MICROARCH.md,
HBM section)."""
import collections
import sys

d = collections.defaultdict(dict)
for line in open(sys.argv[1]):
    f = line.rstrip("\n").split("\t")
    if len(f) >= 6:
        d[f[0]][f[1]] = (float(f[3]), int(f[5]))
print("# derived from", sys.argv[1].split("/")[-1], "(rocprofv3 --pmc, one counter group per run, over the DRIVER's configuration: bench.py --steps 2 --warmup 1, full-size slabs, default options)")
print("# GRBM_GUI_ACTIVE sums the 8 XCDs (8 x 2.1 GHz x launch time); SQ counters sum all 1024 SIMDs -> MFMA busy = MFMA_BUSY / (GUI_ACTIVE / 8 * 1024)")
print("# FETCH_SIZE / WRITE_SIZE in KB; FETCH_SIZE doubled (gfx950 counts 128-B requests as 64 B: MI355X_MICROARCH.md, HBM section)")
for k, c in sorted(d.items()):
    if "GRBM_GUI_ACTIVE" not in c or "SQ_INSTS_MFMA" not in c or c["SQ_INSTS_MFMA"][0] < 1e5:
        continue
    g = lambda n: c.get(n, (0.0, 0))[0]  # noqa: E731
    gui, mf = g("GRBM_GUI_ACTIVE"), g("SQ_INSTS_MFMA")
    wc = max(g("SQ_WAVE_CYCLES"), 1.0)
    k = k.replace("void hm::", "")
    k = "tail_kernel_r (resident weights)" if k == "tail_kernel_r" else "tail_kernel_p (strip tail, CHH)" if k == "tail_kernel_p" else "tail_fc_kernel (fc1 .. softmax, CHH)" if k == "tail_fc_kernel" else k
    print(f"{k:36s} launches {c['GRBM_GUI_ACTIVE'][1]:4d}  mean launch {gui / 8 / 2.1e6:7.3f} ms(@2.1GHz)  "
          f"MFMA busy {g('SQ_VALU_MFMA_BUSY_CYCLES') / (gui / 8 * 1024):.3f}  MFMA/launch {mf:.3g}  VALU/MFMA {g('SQ_INSTS_VALU') / mf:.2f}  "
          f"LDS/MFMA {g('SQ_INSTS_LDS') / mf:.2f}  VMEM_RD/MFMA {g('SQ_INSTS_VMEM_RD') / mf:.2f}  "
          f"HBM-side read {2 * g('FETCH_SIZE') * 1024 / 1e9:.3f} GB  write {g('WRITE_SIZE') * 1024 / 1e9:.3f} GB per launch  "
          f"wave cycles: waiting {g('SQ_WAIT_ANY') / wc:.2f} issue-stalled {(g('SQ_WAIT_INST_ANY') - g('SQ_WAIT_ANY')) / wc:.2f} issuing {g('SQ_ACTIVE_INST_ANY') / wc:.2f}")

if len(sys.argv) > 3:
    import json
    b = json.load(open(sys.argv[2]))
    dm = b["device_ms_timed_region"]
    pos = [dm["trunk_positions"][c] / max(1, dm["trunk_launches"][c]) for c in range(3)]   # positions per launch, per context
    TR = ("trunk3_kernel", "trunk2_kernel")
    kern = next((t for t in TR if any(t in k for k in d)), TR[0])
    out = {"_source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, --kernel-trace only) over `python bench.py --steps 2 --warmup 1 "
                      "--no-extras --no-cpu-baseline` (the driver's full-size slabs and default options); FETCH_SIZE doubled (gfx950 tallies 128-B "
                      "requests as 64 B: MI355X_MICROARCH.md, HBM); WRITE_SIZE as reported: exact for the trunk's 8-B-per-lane row stores as well "
                      "(profiles/r05_write_size_calibration.txt: counter / bytes = 1.0000 for 16-B and 8-B stores, contiguous and in the trunk's "
                      "pattern); Infinity-Cache hits are counted, not excluded",
           "kernel": kern, "command": "python bench.py --steps 2 --warmup 1 --no-extras --no-cpu-baseline", "commit": sys.argv[4] if len(sys.argv) > 4 else None,
           "trunk_kernel": {}}
    for key, tag, cs in (("k11", "", (0, 1)), ("k13", "<13", (2,))):
        c = next((v for k, v in d.items() if kern in k and ("<13" in k) == bool(tag)), None)
        if c is None:
            continue
        p = sum(pos[i] for i in cs) / len(cs)
        rd, wr = 2 * c.get("FETCH_SIZE", (0, 0))[0] * 1024, c.get("WRITE_SIZE", (0, 0))[0] * 1024
        out["trunk_kernel"][key] = {"read_B_per_launch": rd, "write_B_per_launch": wr, "positions_per_launch": p,
                                    "read_B_per_position": rd / p, "write_B_per_position": wr / p, "launches": c["GRBM_GUI_ACTIVE"][1] if "GRBM_GUI_ACTIVE" in c else 0}
    json.dump(out, open(sys.argv[3], "w"), indent=1)
