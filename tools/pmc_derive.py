"""Derived per-kernel figures from a rocprofv3 --pmc summary written by tools/prof_r02.sh (pmc_summary.txt):
   python tools/pmc_derive.py gpurun_out/prof_r02/pmc_summary.txt > profiles/r02_pmc_derived.txt
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
print("# derived from", sys.argv[1].split("/")[-1], "(rocprofv3 --pmc, one counter group per run, bench.py --steps 1 --warmup 1 --reads 2000)")
print("# GRBM_GUI_ACTIVE sums the 8 XCDs (8 x 2.1 GHz x launch time); SQ counters sum all 1024 SIMDs -> MFMA busy = MFMA_BUSY / (GUI_ACTIVE / 8 * 1024)")
print("# FETCH_SIZE / WRITE_SIZE in KB; FETCH_SIZE doubled (gfx950 counts 128-B requests as 64 B: MI355X_MICROARCH.md, HBM section)")
for k, c in sorted(d.items()):
    if "GRBM_GUI_ACTIVE" not in c or "SQ_INSTS_MFMA" not in c or c["SQ_INSTS_MFMA"][0] < 1e5:
        continue
    g = lambda n: c.get(n, (0.0, 0))[0]  # noqa: E731
    gui, mf = g("GRBM_GUI_ACTIVE"), g("SQ_INSTS_MFMA")
    wc = max(g("SQ_WAVE_CYCLES"), 1.0)
    k = "tail_kernel_h<0, GATHER>" if "PKDF16" in k else k.replace("void hm::", "")
    print(f"{k:36s} launches {c['GRBM_GUI_ACTIVE'][1]:4d}  mean launch {gui / 8 / 2.1e6:7.3f} ms(@2.1GHz)  "
          f"MFMA busy {g('SQ_VALU_MFMA_BUSY_CYCLES') / (gui / 8 * 1024):.3f}  MFMA/launch {mf:.3g}  VALU/MFMA {g('SQ_INSTS_VALU') / mf:.2f}  "
          f"LDS/MFMA {g('SQ_INSTS_LDS') / mf:.2f}  VMEM_RD/MFMA {g('SQ_INSTS_VMEM_RD') / mf:.2f}  "
          f"HBM-side read {2 * g('FETCH_SIZE') * 1024 / 1e9:.3f} GB  write {g('WRITE_SIZE') * 1024 / 1e9:.3f} GB per launch  "
          f"wave cycles: waiting {g('SQ_WAIT_ANY') / wc:.2f} issue-stalled {(g('SQ_WAIT_INST_ANY') - g('SQ_WAIT_ANY')) / wc:.2f} issuing {g('SQ_ACTIVE_INST_ANY') / wc:.2f}")
