#!/usr/bin/env python3
"""bench.py -- cytosine sites/sec (CpG+CHG+CHH) of the hifimeth `call` hot path on MI355X.

One "step" = one FRESH slab of synthetic HiFi reads (BASELINE.json configs[2] statistics: Arabidopsis-like GC 0.36,
read length log-normal around 15 kb, codev1 kinetics, all three contexts) going through the whole engine:

    hm_batch_submit_read (host -> pinned slab) -> async H2D -> site scan -> kinetics windows -> CNN -> probabilities
    -> packed D2H of the calls

through the engine's asynchronous batch pipeline, i.e. slab k+1 is staged and uploaded while slab k computes -- the
reference's outer batch loop (src/app/hifimeth/mod_main.cpp:330-362) with its GPU variant's pinned staging
(src/app-gpu/hifimeth-gpu/5mc_call_gpu.cpp:367).  Staging, both copies and the result fetch are INSIDE the timed region;
only the synthesis of the reads (the stand-in for BAM decode) happens before it.  Each rank drives one GPU with its own
slabs (reads are independent: no data-path collective, weak scaling); `value` is the whole-job sites/s = sum of sites
over ranks / max time.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--reads R] [--workload arabidopsis|human_slice] [--e2e-dist] [--no-extras] [--no-e2e] [--no-cpu-baseline]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

`python bench.py --gpus N` with N > 1 and no torchrun environment STARTS the N ranks itself (launch_ranks: fresh child processes
through torch.distributed.run, started before this process has made any HIP / torch.cuda call; never an exec), one device per rank,
RCCL for the barrier and the (sum of sites, max of time) reduction; rank 0 prints the one line with n_gpus = N.
BASELINE.json configs[3] (30x human-size over 8 GPUs) is `python bench.py --gpus 8 --workload human_slice --e2e-dist`: every rank
streams its eighth (64 steps = 11.8 Gbases), then the ranks hand the devices to `python -m hifimeth_amd.call_dist` (queue mode) over ONE
BAM file, so that the host feed (N ranks inflating, parsing, deflating side by side) is measured too.
"""
import argparse
import json
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# algorithmic work per site (BASELINE.md section 2 / SURVEY.md 8d), 2 FLOP per MAC
MAC_TOTAL = {0: 11148800, 1: 11148800, 2: 11440640}
# share of conv1..conv4 (the front of the network)
MAC_FRONT = {0: 197 * 128 * 88 + 99 * 128 * 384 + 50 * 128 * 384 + 25 * 96 * 384,
             1: 197 * 128 * 88 + 99 * 128 * 384 + 50 * 128 * 384 + 25 * 96 * 384,
             2: 196 * 128 * 104 + 98 * 128 * 384 + 49 * 128 * 384 + 25 * 96 * 384}
CONV1_MAC = {0: 197 * 128 * 88, 1: 197 * 128 * 88, 2: 196 * 128 * 104}
PEAK_FP32_MFMA_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32 / 32x32x2_f32 dense peak
PEAK_FP16_MFMA_TFLOPS = 2500.0  # MI355X_MICROARCH.md: BF16/FP16 MFMA ~2.5 PF dense (2:1-sparsity figures excluded)
CONFIG2_SITES = 1.1e9           # 30x Arabidopsis ~ 3.6 Gbases ~ 1.1 G sites (BASELINE.json configs[2])


def host_cores():
    """CPUs this process may actually use: affinity mask capped by the cgroup CPU quota."""
    env = os.environ.get("HM_CPU_THREADS")
    if env:
        return max(1, int(env))
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return n


def cpu_baseline(sample, gpu_calls=None, budget_s=25.0, tol=1e-4, what=""):
    """The CPU oracle (port of the reference path) timed on this node's host cores, bounded sample.
    `sample`: (read id in the slab, read) pairs; `gpu_calls`: {read id: the GPU's records of that read}.
    The oracle's results for that sample double as an in-run parity check of the GPU calls."""
    from oracle import hm_oracle as O
    O.build()
    cores = host_cores()
    models = [O.Model(os.path.join(ROOT, "hifimeth_amd", "weights", n + ".hmw")) for n in ("CpG", "CHG", "CHH")]
    spent = 0.0
    sites = 0
    nreads = 0
    worst, nml, ncheck = 0.0, 0, 0
    deltas = []
    for rid, rd in sample:
        if not rd.has_kinetics() or rd.l_qseq < 1000:
            continue
        t1 = time.perf_counter()
        want = O.call_read(models, 7, rd, nthreads=cores)
        spent += time.perf_counter() - t1
        sites += len(want["qoff"])
        nreads += 1
        if gpu_calls is not None:  # untimed: compare with the GPU's calls for the same read
            got = gpu_calls[rid]
            order = np.lexsort((want["qoff"], want["strand"]))
            assert len(got) == len(order) and np.array_equal(got["qoff"], want["qoff"][order]), "site lists differ"
            d = np.abs(got["p"] - want["p"][order])
            deltas.append(d)
            worst = max(worst, float(d.max(initial=0)))
            nml += int((got["scaled_prob"] != want["ml"][order]).sum())
            ncheck += len(got)
        if spent > budget_s:
            break
    out = {"value": sites / spent, "unit": "sites/s", "cores": cores, "kind": "port",
           "sample": f"{nreads} reads of the rank-0 slab 0 {what}({sites} sites, all contexts), "
                     f"oracle/hm_oracle.c fp32 ({O.variant} build), OpenMP over sites on all {cores} cores this job is "
                     f"granted ({os.cpu_count()} online), {spent:.1f} s",
           "extrapolated_48_threads": sites / spent * 48 / cores,
           "extrapolation_note": "linear in cores (the port scales over sites with no shared state); "
                                 "the north-star's 48-thread reference host is not this box"}
    parity = None
    if gpu_calls is not None:
        alld = np.concatenate(deltas) if deltas else np.zeros(1)
        parity = {"sample": what.strip() or "synchronous call", "sites_checked": ncheck, "max_abs_dp_vs_oracle": worst, "mean_abs_dp": float(alld.mean()),
                  "p999_abs_dp": float(np.quantile(alld, 0.999)), "ml_bytes_off_by_1lsb": nml, "tolerance": tol,
                  "frac_above_1e-4": float((alld > 1e-4).mean()), "frac_above_1e-3": float((alld > 1e-3).mean())}
    return out, parity


GROUP_BASES = 16 << 20   # bases per trunk read group: read back from the engine in main() (its default is sized from free device memory)


def trunk_groups(reads):
    """Group index of every accepted read of a slab, by the engine's own rule: a new group starts once the current one holds
    GROUP_BASES map ROWS -- ceil((len + 400) / 112) * 112 + 32 per read (hm_engine.cpp: add_read_tiles).  -1 for reads the engine passes through."""
    g, held, out = -1, GROUP_BASES, []
    for r in reads:
        if not r.has_kinetics() or r.l_qseq < 1000:
            out.append(-1)
            continue
        if held >= GROUP_BASES:
            g, held = g + 1, 0
        held += (r.l_qseq + 400 + 111) // 112 * 112 + 32
        out.append(g)
    return out


def parity_sample(reads, per_group=32):
    """Read ids of the oracle sample of a streamed slab: the first reads of trunk group 0, reads from the middle group and
    the LAST reads of the last group -- the map buffers are reused group after group, so a wrong map offset or a buffer
    reused one launch early would show in the later groups only."""
    grp = trunk_groups(reads)
    ng = max(grp) + 1
    if ng <= 0:
        return [], 0
    picks = []
    for g in sorted({0, ng // 2, ng - 1}):
        members = [i for i, x in enumerate(grp) if x == g]
        picks.append(members[-per_group:][::-1] if g == ng - 1 and ng > 1 else members[:per_group])
    # interleaved, so that a CPU time budget that runs out early has still seen every group
    ids = [p[k] for k in range(per_group) for p in picks if k < len(p)]
    return list(dict.fromkeys(ids)), ng


def quoted_traffic(tm, launches, sites):
    """HBM traffic of the trunk kernel.  HBM bytes come from PMC counters, which need rocprofv3 passes of their own (one counter
    group per run) and cannot be read from inside this process: `traffic` (measured in THIS run) is therefore null, and the figure
    of the round's kept full-size pass (profiles/r05_traffic.json, written by tools/pmc_derive.py from tools/prof_r05.sh traffic:
    this same command, gfx950 corrections applied) is given under `traffic_quoted`, with its source."""
    alg = 28.0 * sites / max(1, launches)   # ~16 B of raw input + 12 B of result per site
    out = {"traffic": None, "algorithmic_bytes": alg,
           "traffic_note": "HBM bytes need separate rocprofv3 --pmc passes (tools/prof_r05.sh traffic); see traffic_quoted"}
    path = next((q for q in (os.path.join(ROOT, "profiles", n) for n in ("r05_traffic.json", "r04_traffic.json")) if os.path.exists(q)), "")
    try:
        q = json.load(open(path))
        k = q["trunk_kernel"]
        per_pos = {0: k["k11"], 1: k["k11"], 2: k["k13"]}
        tot = sum(tm["trunk_positions"][c] * (per_pos[c]["read_B_per_position"] + per_pos[c]["write_B_per_position"]) for c in range(3))
        out["traffic_quoted"] = {"bytes_per_launch": tot / max(1, launches), "over_algorithmic": tot / max(1, launches) / alg,
                                 "source": "profiles/" + os.path.basename(path), "kernel": q.get("kernel"), "command": q.get("command"),
                                 "commit": q.get("commit"),
                                 "note": "HBM-side bytes per view position measured by rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of the "
                                         "full-size bench command (gfx950: FETCH_SIZE doubled), times this run's positions per launch; the "
                                         "maps E1..E4 the design parks in HBM are what exceeds the algorithmic bytes"}
    except (OSError, KeyError, ValueError):
        pass
    return out


def end_to_end(slabs, n_reads, ctx="cpg,chg,chh", extra_flags=(), cycles=2):
    """`hifimeth-hip call IN.bam OUT.bam` with the reference's default flags (mod_options.cpp:10-17) on a synthetic BAM of
    n_reads reads: BGZF inflate -> parse -> stage -> GPU -> MM/ML tags -> deflate, engine start-up included -- the quantity
    the reference's one published figure is about (README.md:31: wall-clock of the whole command)."""
    import shutil
    import subprocess
    import tempfile
    from hifimeth_amd.synth import write_unaligned_bam
    cli = os.path.join(ROOT, "hifimeth_amd", "bin", "hifimeth-hip")
    if not os.path.exists(cli):
        return {"error": "hifimeth_amd/bin/hifimeth-hip is not built"}
    pool = [r for s in slabs for r in s]
    n_reads = min(n_reads, cycles * len(pool))   # the pool cycles at most twice by default (a 5 GB file needs more reads than six slabs hold)
    reads = [pool[i % len(pool)] for i in range(n_reads)]
    tmp = tempfile.mkdtemp(prefix="hm_e2e_")
    try:
        src, dst = os.path.join(tmp, "in.bam"), os.path.join(tmp, "out.bam")
        t0 = time.perf_counter()
        write_unaligned_bam(src, reads, level=1, threads=host_cores())
        t_build = time.perf_counter() - t0
        t0 = time.perf_counter()
        p = subprocess.run([cli, "call", "-c", ctx, src, dst], stderr=subprocess.PIPE, stdout=subprocess.DEVNULL, text=True)
        wall = time.perf_counter() - t0
        if p.returncode != 0:
            return {"error": f"exit {p.returncode}: {p.stderr[-300:]}"}
        more = {}
        for fl in extra_flags:   # the same file again with one more flag (e.g. -Z: deflate the output with libdeflate)
            t1 = time.perf_counter()
            q = subprocess.run([cli, "call", "-c", ctx, fl, src, dst + fl], stderr=subprocess.PIPE, stdout=subprocess.DEVNULL, text=True)
            codec = [ln.split(":", 1)[1].strip() for ln in q.stderr.splitlines() if "## BGZF codec" in ln]
            more[fl] = {"wall_s": time.perf_counter() - t1, "exit": q.returncode, "bgzf_codec": codec[0] if codec else None,
                        "bam_out_MB": os.path.getsize(dst + fl) / 1e6 if q.returncode == 0 else None}
        st = {}
        for line in p.stderr.splitlines():
            if "##" in line and ":" in line:
                k, v = line.split("##", 1)[1].split(":", 1)
                st[k.strip()] = v.strip()
        sites = sum(int(st.get(f"{c} samples", "0")) for c in ("CpG", "CHG", "CHH"))
        return {"value": sites / wall, "unit": "sites/s", "wall_s": wall, "reads": len(reads), "bases": int(st.get("Bases", "0")),
                "sites": sites, "bam_in_MB": os.path.getsize(src) / 1e6, "bam_out_MB": os.path.getsize(dst) / 1e6,
                "host_threads": host_cores(), "flags": "defaults (-b 10000, -l 1000, all contexts, -z 6)", "bgzf_codec": st.get("BGZF codec"),
                "command": "hifimeth-hip call IN.bam OUT.bam", "bam_build_s": t_build,
                "with_flag": {k: dict(v, value=sites / v["wall_s"]) for k, v in more.items()},
                "note": "whole command incl. process and engine start-up; BGZF level-1 input, level-6 output"}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


TORCHRUN_ENV = ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "GROUP_RANK", "GROUP_WORLD_SIZE", "ROLE_RANK", "ROLE_WORLD_SIZE",
                "ROLE_NAME", "MASTER_ADDR", "MASTER_PORT", "TORCHELASTIC_RUN_ID", "TORCHELASTIC_RESTART_COUNT", "TORCHELASTIC_MAX_RESTARTS",
                "TORCHELASTIC_USE_AGENT_STORE", "TORCH_NCCL_ASYNC_ERROR_HANDLING", "TORCHELASTIC_ERROR_FILE", "OMP_NUM_THREADS")


def free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def clean_env():
    """This process's environment without what a torchrun parent put there (a nested launch must build its own world)."""
    env = {k: v for k, v in os.environ.items() if k not in TORCHRUN_ENV}
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # (this pool's driver supports dmabuf IPC only: RCCL needs it; already exported on the boxes)
    return env


def launch_ranks(n, argv):
    """`python bench.py --gpus N` outside torchrun: N ranks as fresh children of torch.distributed.run.  This process has made
    no HIP call and makes none (the device count comes from the KFD sysfs topology): it waits and passes the children's exit
    code on.  Fails loudly when the box has fewer devices than ranks, unless HM_DIST_BACKEND=gloo asks for
    the rehearsal in which the ranks share a card (RCCL refuses two ranks on one device)."""
    import subprocess
    from hifimeth_amd.dist import gpu_count
    ndev = gpu_count()   # KFD sysfs, not HIP: this process does not open the device
    if ndev < n and os.environ.get("HM_DIST_BACKEND", "nccl") != "gloo":
        sys.exit(f"bench.py: --gpus {n} but this box has {ndev} GPU(s); one rank per device (RCCL).  "
                 f"HM_DIST_BACKEND=gloo rehearses the multi-rank path with ranks sharing a device.")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.abspath(__file__)] + list(argv)
    print(f"[bench] starting {n} ranks: {' '.join(cmd[1:9])} bench.py ...", file=sys.stderr, flush=True)
    return subprocess.call(cmd, env=clean_env())


def end_to_end_dist(slabs, n_reads, ranks, cycles=2, ctx="cpg,chg,chh"):
    """BASELINE.json configs[3]'s host side: ONE BAM file called by `ranks` processes that pull parts of it from the shared work
    queue (python -m hifimeth_amd.call_dist: hifimeth-hip call -Q queue -C parts -d gpu per rank, rank-0 merge) -- the reference's
    reader queue (src/corelib/sam_batch.hpp:38-54) and ordered write (mod_main.cpp:330-362) stretched over processes.  Wall clock of the
    whole launch: torchrun + N engine start-ups + N x (inflate, parse, stage, GPU, tags, deflate) + merge."""
    import shutil
    import subprocess
    import tempfile
    from hifimeth_amd.synth import write_unaligned_bam
    from hifimeth_amd.dist import gpu_count
    pool = [r for s in slabs for r in s]
    n_reads = min(n_reads, cycles * len(pool))
    reads = [pool[i % len(pool)] for i in range(n_reads)]
    tmp = tempfile.mkdtemp(prefix="hm_e2ed_")
    try:
        src, dst = os.path.join(tmp, "in.bam"), os.path.join(tmp, "out.bam")
        t0 = time.perf_counter()
        write_unaligned_bam(src, reads, level=1, threads=host_cores())
        t_build = time.perf_counter() - t0
        env = clean_env()
        env.pop("HM_DIST_BACKEND", None)   # call_dist's ranks never touch the GPU themselves: gloo barrier, the native child owns the device
        threads = max(1, host_cores() // ranks)
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(ranks), "--master-addr", "127.0.0.1",
               "--master-port", str(free_port()), "-m", "hifimeth_amd.call_dist", "-c", ctx, "-t", str(threads), src, dst]
        t0 = time.perf_counter()
        p = subprocess.run(cmd, stderr=subprocess.PIPE, stdout=subprocess.DEVNULL, text=True, env=env, cwd=ROOT)
        wall = time.perf_counter() - t0
        if p.returncode != 0:
            return {"error": f"exit {p.returncode}: {p.stderr[-400:]}"}
        tot = {"CpG samples": 0, "CHG samples": 0, "CHH samples": 0, "Bases": 0, "Reads": 0}
        per_rank, parts_taken = [], []
        for line in p.stderr.splitlines():   # every rank's native child prints its own final stats
            if "##" in line and ":" in line:
                k, v = line.split("##", 1)[1].split(":", 1)
                k, v = k.strip(), v.strip()
                if k in tot:
                    tot[k] += int(v)
                elif k.startswith("Parts taken"):
                    parts_taken.append(int(v.split()[0]))
                elif k.startswith("Wall time"):
                    per_rank.append(float(v.split()[0]))
        sites = tot["CpG samples"] + tot["CHG samples"] + tot["CHH samples"]
        return {"value": sites / wall, "unit": "sites/s", "wall_s": wall, "ranks": ranks, "devices": gpu_count(),
                "reads": tot["Reads"], "reads_in_file": len(reads), "bases": tot["Bases"], "sites": sites,
                "bam_in_MB": os.path.getsize(src) / 1e6, "bam_out_MB": os.path.getsize(dst) / 1e6,
                "host_threads_per_rank": threads, "host_cores": host_cores(), "parts_taken_by_rank": parts_taken,
                "rank_wall_s": per_rank, "bam_build_s": t_build,
                "command": f"python -m torch.distributed.run --nproc-per-node {ranks} -m hifimeth_amd.call_dist -t {threads} IN.bam OUT.bam",
                "note": "queue mode: parts of ~256 MB of BAM (at least 4 per rank) claimed from a flock'ed counter; whole launch incl. "
                        "torchrun, every rank's engine start-up and the rank-0 merge; ranks beyond the box's devices share a card"}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


WORKLOADS = {
    # BASELINE.json configs[2]: 30x Arabidopsis-size, GC 0.36, i.i.d. bases; 20 steps of 11 700 reads cover it in full
    "arabidopsis": {"gc": 0.36, "cpg_oe": 1.0, "steps": 20, "total_sites": 1.1e9,
                    "what": "BASELINE.json configs[2] statistics (GC 0.36, i.i.d. bases, ~0.30 sites per base), all three contexts"},
    # BASELINE.json configs[3], ONE rank's share: 1/8 of a 30x human-size input = 11.6 Gbases; GC 0.41 and CpG depleted as in a vertebrate
    # genome (observed / expected 0.24: CpG ~1 % of the bases) -- so CpG takes the per-site kernels while CHG / CHH take the trunk
    "human_slice": {"gc": 0.41, "cpg_oe": 0.24, "steps": 64, "total_sites": None,
                    "what": "one rank's eighth of BASELINE.json configs[3] (30x human-size: 11.6 Gbases per rank; GC 0.41, CpG observed / "
                            "expected 0.24), all three contexts; NO 8-GPU curve exists: this is the per-rank slice on one GPU"},
}


def make_slabs(n, reads, seed, gc=0.36, cpg_oe=1.0):
    """n distinct slabs of `reads` synthetic reads each (threads: numpy releases the GIL in the RNG / table passes)."""
    from hifimeth_amd.synth import synth_slab
    with ThreadPoolExecutor(max_workers=max(1, min(n, host_cores()))) as ex:
        return list(ex.map(lambda i: synth_slab(reads, seed=seed + 7919 * i, gc=gc, cpg_oe=cpg_oe), range(n)))


def stream(mc, slabs, order, keep=None):
    """Runs slabs[order[0]], slabs[order[1]], ... through the batch pipeline; returns the sites per context.
    `slabs` hold ReadBlocks (descriptors = pointers to the reads' SEQ / kinetics arrays, what a BAM decoder hands over):
    a step stages its slab with ONE hm_batch_submit_reads call, whose copies into the pinned slab run on a few host threads.
    keep = (read ids, dict): the streamed records of those reads of the FIRST slab are copied into the dict."""
    sites = [0, 0, 0]
    calls_seen = 0

    def on_batch(k, batch, calls):
        nonlocal calls_seen
        for c in range(3):
            sites[c] += batch.num_sites(c)
        calls_seen += len(calls)  # the records are on the host here (pinned view of the packed D2H)
        if keep is not None and k == 0:   # records are ordered by read: two binary searches per wanted read
            rid = calls["read_id"]
            for i in keep[0]:
                lo, hi = np.searchsorted(rid, [i, i + 1])
                keep[1][i] = calls[lo:hi].copy()

    mc.stream((slabs[i] for i in order), on_batch=on_batch)
    assert calls_seen == sum(sites)
    return sites


def main():
    global GROUP_BASES
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="default: what covers the workload (20 | 64)")
    ap.add_argument("--workload", default="arabidopsis", choices=sorted(WORKLOADS))
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--reads", type=int, default=11700,
                    help="reads per slab = per step (~15 kb each); default: 20 steps cover BASELINE.json configs[2] (~1.1 G sites)")
    ap.add_argument("--pool", type=int, default=6, help="distinct slabs synthesised up front; steps cycle through them")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the secondary measurements (fp32 mode, resident slab, windows)")
    ap.add_argument("--no-e2e", action="store_true", help="skip the end-to-end CLI run (BAM in -> BAM out, default flags)")
    ap.add_argument("--e2e-reads", type=int, default=96000, help="reads of the synthetic BAM of the end-to-end run (96000 ~ 5 GB: start-up amortised)")
    ap.add_argument("--e2e-cycles", type=int, default=2, help="how often the end-to-end file may repeat the read pool (4 with --e2e-reads 234000: configs[2] in full)")
    ap.add_argument("--e2e-dist", action="store_true",
                    help="after the timed region: ONE BAM file called by all ranks through the shared work queue (python -m hifimeth_amd.call_dist)")
    ap.add_argument("--e2e-dist-ranks", type=int, default=0, help="ranks of the --e2e-dist leg (default: --gpus; more ranks than devices share a card)")
    ap.add_argument("--precision", type=int, default=1, choices=[0, 1, 2],
                    help="CNN arithmetic: 1 = split-half fp16x3 MFMA + fp32 accumulate (default), 0 = fp32 MFMA, 2 = as 1 with plain fp16 "
                         "weights in conv8 and fc1 (BASELINE.json configs[4]'s variant that holds 1e-3 with margin)")
    ap.add_argument("--opt", action="append", default=[], help="engine option key=value (e.g. front_waves=8)")
    args = ap.parse_args()
    wl = WORKLOADS[args.workload]
    if args.steps is None:
        args.steps = wl["steps"]

    if args.gpus < 1:
        ap.error("--gpus must be >= 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N`: this process only launches the N ranks (before anything here has touched the GPU) and waits
        raise SystemExit(launch_ranks(args.gpus, sys.argv[1:]))

    from hifimeth_amd import MethylationCaller
    from hifimeth_amd import dist as hmdist

    rank, local_rank, world = hmdist.env_world()
    if world != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but the launcher's WORLD_SIZE is {world}: one rank per GPU, start as many ranks as --gpus says")
    # HM_DIST_BACKEND=gloo lets the multi-rank path be rehearsed on a box with fewer GPUs than ranks
    backend = os.environ.get("HM_DIST_BACKEND", "nccl")
    import torch
    ndev = torch.cuda.device_count()
    if ndev < world and backend == "nccl":
        sys.exit(f"bench.py: {world} ranks but {ndev} GPU(s) on this box: RCCL needs one device per rank (HM_DIST_BACKEND=gloo rehearses with a shared card)")
    # HM_BENCH_FORCE_DIST=1 builds a world of one, so that the RCCL barrier / reductions run on a one-GPU box too
    force = os.environ.get("HM_BENCH_FORCE_DIST", "0") == "1"
    dist = hmdist.init_process_group(backend, force=force) if (world > 1 or force) else None
    on_gpu_collectives = dist is not None and backend == "nccl"

    n_pool = max(1, min(args.pool, args.steps + args.warmup))
    slabs = make_slabs(n_pool, args.reads, seed=20250220 + 104729 * rank, gc=wl["gc"], cpg_oe=wl["cpg_oe"])
    bases_slab = [sum(r.l_qseq for r in s if r.has_kinetics() and r.l_qseq >= 1000) for s in slabs]
    from hifimeth_amd.caller import ReadBlock
    blocks = [ReadBlock(s) for s in slabs]
    mc = MethylationCaller(device=local_rank % max(ndev, 1), timing=True)
    mc.set_option("precision", args.precision)
    mc.stage_threads = max(1, min(8, host_cores() // max(1, min(world, 8)) // 2))
    trunk_impl = 3
    for kv in args.opt:
        k, v = kv.split("=")
        mc.set_option(k, int(v))
        if k == "trunk_impl":
            trunk_impl = int(v)
    GROUP_BASES = int(mc.timing()["group_bases"])   # the engine's group size in force (default: sized from free device memory)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    order_w = [i % n_pool for i in range(args.warmup)]
    order_t = [(args.warmup + i) % n_pool for i in range(args.steps)]
    # the oracle sample: reads of slab 0 from its first, a middle and its last trunk group; their records are taken from the
    # STREAMED run of that slab (warm-up step 0: same pipeline, same multi-group launches as the timed steps)
    want_parity = rank == 0 and world == 1 and not args.no_cpu_baseline
    sample_ids, n_groups = parity_sample(slabs[0]) if want_parity else ([], 0)
    streamed = {}
    keep = (sample_ids, streamed) if sample_ids else None
    if order_w:
        stream(mc, blocks, order_w, keep=keep)   # untimed: buffers grow to their steady-state size here
    mc.timing(reset=True)
    barrier()
    t0 = time.perf_counter()
    sites_ctx = stream(mc, blocks, order_t)
    barrier()
    dt = time.perf_counter() - t0
    tm = mc.timing()

    sites_job = sum(sites_ctx)
    bases_job = sum(bases_slab[i] for i in order_t)
    sites_all, dt_max = hmdist.job_throughput(dist, sites_job, dt, device="cuda" if on_gpu_collectives else "cpu")
    # every rank's own (sites, seconds, device index), on every rank: the line shows the slowest rank, not only the sum
    by_rank = hmdist.gather_rank_stats(dist, [sites_job, dt, local_rank % max(ndev, 1), sum(tm["trunk_ms"]), sum(tm["edge_ms"]), sum(tm["tail_ms"])],
                                       device="cuda" if on_gpu_collectives else "cpu")

    extras = {}
    if keep is not None and not streamed:
        stream(mc, blocks, [0], keep=keep)   # --warmup 0: the sample's records from an extra, untimed streamed run of slab 0
    if rank == 0 and world == 1 and not args.no_extras:
        # ---- secondary measurements, all OUTSIDE the timed region -----------------------------------------------
        # (a) the same slab resident in HBM, re-run without staging / copies: the kernel-path rate of round 1
        small = slabs[0][:96]
        mc.clear()
        mc.submit_all(small)
        mc.upload()
        mc.run()
        mc.sync()
        t1 = time.perf_counter()
        for _ in range(5):
            mc.run()
        mc.sync()
        dtr = time.perf_counter() - t1
        extras["resident_slab"] = {"value": mc.num_sites(3) * 5 / dtr, "unit": "sites/s", "reads": len(small),
                                   "note": "96-read slab already in HBM, hm_run x5: no staging, no copies (round-1 workload)"}
        # (b) feature extraction standalone: the materialised 401x8 fp32 windows of the largest context, device only.
        # In the product path the window never leaves the chip; this is the HBM-bound kernel the north-star asks to be
        # priced against the HBM roofline.
        sc = [mc.num_sites(c) for c in range(3)]
        big = int(np.argmax(sc))
        mc.timing(reset=True)
        for _ in range(3):
            mc.windows(big, 0, sc[big], fetch=False)
        tw = mc.timing()
        if tw["window_ms"] > 0:
            # 12 832 B of window written + the 5 B per base of raw input AMORTISED over the context's sites + the 12-byte
            # site record (the kernel re-reads a 401 x 5 B slice per site, but from L2: not algorithmic bytes -- VERDICT r04)
            bases_in = sum(r.l_qseq for r in small if r.has_kinetics() and r.l_qseq >= 1000)
            per_site = 401 * 8 * 4 + 5.0 * bases_in / max(1, sc[big]) + 12
            feat = {"kernel": "window_kernel (401x8 fp32 windows to HBM, test/roofline seam)", "bound": "hbm",
                    "achieved": tw["window_sites"] * per_site / (tw["window_ms"] * 1e-3) / 1e9, "peak": 8000.0, "unit": "GB/s",
                    "sites_per_launch": sc[big], "avg_launch_ms": tw["window_ms"] / tw["window_launches"],
                    "algorithmic_bytes_per_site": per_site}
            feat["frac"] = feat["achieved"] / feat["peak"]
            extras["feature_extraction"] = feat
        mc.clear()
        # (c) strict-fp32 arithmetic (v_mfma_f32_16x16x4_f32), streamed the same way over fewer slabs
        if args.precision >= 1:
            mc.set_option("precision", 0)
            mc.timing(reset=True)
            k32 = max(1, min(3, args.steps))
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            s32 = stream(mc, blocks, [i % n_pool for i in range(k32)])
            torch.cuda.synchronize()
            d32 = time.perf_counter() - t1
            t32 = mc.timing()
            f_ms, tr_ms = sum(t32["front_ms"]), sum(t32["trunk_ms"])
            fp32 = {"value": sum(s32) / d32, "unit": "sites/s", "steps": k32, "dtype": "f32", "peak": PEAK_FP32_MFMA_TFLOPS,
                    "note": "same streamed workload, engine option precision=0"}
            if tr_ms > 0:
                # dense trunk in strict fp32 (hm_trunk_f32.hip): `achieved` as in the headline roofline = the reference's
                # conv1..conv4 FLOPs for the sites served / kernel time; `executed` = 2048 FLOP per v_mfma_f32_16x16x4_f32,
                # per 112-position tile 9 / 9 / 8 / 7 position tiles x 8 / 8 / 8 / 6 channel tiles x K / 4 steps
                k1g = {0: 24, 1: 24, 2: 28}
                mf = {c: 9 * 8 * k1g[c] + (9 + 8) * 8 * 96 + 7 * 6 * 96 for c in range(3)}
                ach = sum(2.0 * MAC_FRONT[c] * s32[c] for c in range(3)) / (tr_ms * 1e-3) / 1e12
                ex = sum(t32["trunk_positions"][c] / 112.0 * mf[c] * 2048.0 for c in range(3)) / (tr_ms * 1e-3) / 1e12
                # here `achieved` is the EXECUTED rate (the dense form issues ~5x fewer MFMA FLOPs than the reference's per-site
                # count, so the algorithmic figure exceeds the fp32 MFMA peak and says nothing about the kernel)
                fp32.update({"kernel": "trunk_kernel_f32 (v_mfma_f32_16x16x4_f32)", "achieved": ex, "frac": ex / PEAK_FP32_MFMA_TFLOPS,
                             "algorithmic": ach, "algorithmic_note": "reference conv1..conv4 FLOPs for the sites served / kernel time",
                             "device_ms": {"trunk_ms": tr_ms, "edge_ms": sum(t32["edge_ms"]), "tail_ms": sum(t32["tail_ms"])}})
            else:
                fl = sum(2.0 * MAC_FRONT[c] * t32["front_sites"][c] for c in range(3))
                ach = fl / (f_ms * 1e-3) / 1e12 if f_ms > 0 else 0.0
                fp32.update({"kernel": "front_kernel (v_mfma_f32_16x16x4_f32)", "achieved": ach, "frac": ach / PEAK_FP32_MFMA_TFLOPS})
            extras["fp32_mode"] = fp32
            mc.set_option("precision", args.precision)

    if rank == 0:
        split = args.precision >= 1
        peak = PEAK_FP16_MFMA_TFLOPS if split else PEAK_FP32_MFMA_TFLOPS
        front_ms = sum(tm["front_ms"])
        trunk_ms = sum(tm["trunk_ms"])
        flop_front_sites = lambda n: sum(2.0 * MAC_FRONT[c] * n[c] for c in range(3))  # noqa: E731
        if trunk_ms > 0:
            # Dominant kernel: trunk2_kernel (conv1..conv4 once per read position instead of once per site).
            # `achieved` prices it with the ALGORITHMIC work of the path: the conv1..conv4 FLOPs the
            # reference spends on the sites this kernel served; `executed` counts the MFMA work it really issued
            # (16 384 FLOP per v_mfma_f32_16x16x32_f16; per 112-position tile 9 / 9 / 8 / 7 position tiles of 16 rows, conv1
            # one stacked product over 6 | 7 k-blocks, conv2..conv4 three split-half products over 12 k-blocks).
            launches = sum(tm["trunk_launches"])
            served = [sites_ctx[c] if tm["trunk_ms"][c] > 0 else 0 for c in range(3)]   # (a sparse context may take the per-site kernels)
            algorithmic = flop_front_sites(served) / (trunk_ms * 1e-3) / 1e12
            # MFMAs per 112-position tile: position tiles of 16 rows x channel tiles x k-blocks (conv1: one stacked product over 6 | 7
            # k-blocks; conv2..conv4: three split-half products over 12 k-blocks).  trunk3 (sliding window): 7 position tiles in every
            # layer; trunk2: 9 / 9 / 8 / 7 (the halo recomputed per tile).
            rows = (7, 7, 7, 7) if trunk_impl == 3 else (9, 9, 8, 7)
            c3_products = 24 if any(kv == "conv3_w16=1" for kv in args.opt) else 36   # diagnostic option conv3_w16: conv3's w_lo x_hi product dropped
            mfma_tile = {c: rows[0] * (6 if c < 2 else 7) * 8 + rows[1] * 36 * 8 + rows[2] * c3_products * 8 + rows[3] * 36 * 6 for c in range(3)}
            # tiles whose conv4 ran over the listed (needed) rows only -- counted by the kernel -- issued 4 instead of 7 position tiles there
            listed = list(tm["trunk_list_steps"])
            # tiles stored as constant rows (a read's first tile, the tiles behind its end: no receptive field reaches the read) issued none;
            # a workgroup's calibration step and the warm-up step of a run that starts inside a read (< 0.3 %) are not counted
            const_tiles = list(tm.get("trunk_const_steps", [0, 0, 0]))
            n_mfma = sum((tm["trunk_positions"][c] / 112.0 - const_tiles[c]) * mfma_tile[c] - listed[c] * 3 * 36 * 6 for c in range(3))
            executed = n_mfma * 16384.0 / (trunk_ms * 1e-3) / 1e12
            kname = ("trunk3_kernel (sliding window over consecutive tiles: every layer computes 112 rows per tile, the right-hand rows kept in LDS)"
                     if trunk_impl == 3 else "trunk2_kernel")
            # `achieved` / `frac` are the EXECUTED figures -- MFMA FLOPs the kernel issued / its time, against the dense fp16 MFMA peak: the
            # hardware fraction.  The reference's per-site conv1..conv4 FLOPs for the sites served divided by the same time
            # is given as `algorithmic`: the dense form shares work between sites, so that figure can exceed the peak for a launch and is
            # not a roofline fraction.
            roof = {"bound": "mfma", "unit": "TFLOP/s", "achieved": executed, "peak": peak, "frac": executed / peak,
                    "avg_launch_ms": trunk_ms / launches, "launches": launches,
                    "executed": executed, "frac_executed": executed / peak, "utilisation": executed / peak,
                    "algorithmic": algorithmic, "algorithmic_over_peak": algorithmic / peak,
                    "algorithmic_flops_per_site": {"CpG": 2 * MAC_FRONT[0], "CHG": 2 * MAC_FRONT[1], "CHH": 2 * MAC_FRONT[2]},
                    "mfma_per_tile": mfma_tile, "tiles": [tm["trunk_positions"][c] // 112 for c in range(3)], "tiles_conv4_on_listed_rows": listed, "tiles_constant": const_tiles,
                    "positions_per_site": sum(tm["trunk_positions"]) / max(1, sites_job),
                    "sustained_peak_random_operands": 1880.0,
                    "frac_of_sustained": executed / 1880.0,
                    "kernel": kname + " -- feature rows + bn0 + conv1..conv4 as dense a-trous maps over every read position; streaming form: "
                              "4 waves, a layer's weights resident in registers, positions streamed in tile groups; "
                              "v_mfma_f32_16x16x32_f16 split-half x3, fp32 accumulate.  `achieved` = `executed` = MFMA FLOPs issued / kernel time; "
                              "`algorithmic` = the reference's conv1..conv4 FLOPs for the sites served / kernel time; "
                              "`sustained_peak_random_operands`: what a pure MFMA loop holds on this chip on random data (profiles/r04_mfma_shapes.txt)"}
            roof.update(quoted_traffic(tm, launches, sites_job))
            # ---- every MFMA kernel of the path, executed figures (VERDICT r04 #6a): MFMAs issued x 16 384 FLOP / the kernel's HIP-event time
            # edge2_kernel, per pass of 32 sites (64 pseudo-rows = 4 m-tiles): conv1 4 x 8 n-tiles x (6 | 7) stacked k-blocks; conv2 / conv3
            # (8 n-tiles) and conv4 (6): left chains (2 m-tiles) 8 live k-blocks x 3 products, right chains 8 where the window ends on the
            # padding (K1 = 11: conv2, conv3; K1 = 13: conv4), else 12
            edge_pass = {11: 4 * 8 * 6 + 2 * (2 * 8 * 8 * 3 + 2 * 8 * 8 * 3) + (2 * 6 * 8 * 3 + 2 * 6 * 12 * 3),
                         13: 4 * 8 * 7 + 2 * (2 * 8 * 8 * 3 + 2 * 8 * 12 * 3) + (2 * 6 * 8 * 3 + 2 * 6 * 8 * 3)}
            k1 = {0: 11, 1: 11, 2: 13}
            edge_mfma = sum(served[c] / 32.0 * edge_pass[k1[c]] for c in range(3))
            # tail_kernel_r, per pass of 8 sites: conv5 7 m-tiles x 6 x 9 x 3, conv6 4 x 6 x 9 x 3, conv7 2 x 4 x 9 x 3, conv8 1 x 4 x 6 x 3, fc1 per
            # 32 sites 2 x 16 x 4 x 3.  tail_kernel_p (strip tail, CHH), per pass of 16 site slots, zero-padding taps skipped: conv5 DENSE over the strip's rows, (9 row tiles x 9
            # + 2 site tiles x 6) k-blocks x 3 x 6 (site-major it was (13 x 9 - 6)), conv6 (7 x 9 - 6) x 3 x 6, conv7 (4 x 9 - 6) x 3 x 4, conv8 (2 x 6 - 2) x 3 x 4; the kernel counts its passes.  fc1 of those
            # sites runs in tail_fc_kernel on full tiles of 16 sites in list order: 16 n-tiles x 4 k-blocks x 3 per tile (the ragged last tile of a
            # launch -- one in ~ 160 000 -- is not counted)
            pr8 = 2 if args.precision == 2 else 3   # products per k-block in conv8 and fc1 (precision 2: plain fp16 weights there)
            tail_r_site = (7 * 6 * 27 + 4 * 6 * 27 + 2 * 4 * 27 + 4 * 6 * pr8) / 8.0 + 2 * 16 * 4 * pr8 / 32.0
            strip_pass = (9 * 9 + 2 * 6) * 18 + (7 * 9 - 6) * 18 + (4 * 9 - 6) * 12 + (2 * 6 - 2) * 4 * pr8
            fc_tile = 16 * 4 * pr8
            strip_passes = int(tm.get("tail_strip_passes", 0))
            strip_mfma = strip_passes * strip_pass + (served[2] / 16.0 * fc_tile if strip_passes > 0 else 0.0)
            tail_mfma = sum(served[c] * tail_r_site for c in range(3) if not (c == 2 and strip_passes > 0)) + strip_mfma
            edge_ms_t, tail_ms_t = sum(tm["edge_ms"]), sum(tm["tail_ms"])
            tf = lambda n, ms: n * 16384.0 / (ms * 1e-3) / 1e12 if ms > 0 else 0.0  # noqa: E731
            dev_ms = trunk_ms + edge_ms_t + tail_ms_t
            roof["by_kernel"] = {
                "trunk": {"mfma": n_mfma, "ms": trunk_ms, "executed": tf(n_mfma, trunk_ms), "frac": tf(n_mfma, trunk_ms) / peak, "share_of_device_ms": trunk_ms / dev_ms},
                "edge": {"mfma": edge_mfma, "ms": edge_ms_t, "executed": tf(edge_mfma, edge_ms_t), "frac": tf(edge_mfma, edge_ms_t) / peak, "share_of_device_ms": edge_ms_t / dev_ms,
                         "mfma_per_site": {"K1=11": edge_pass[11] / 32.0, "K1=13": edge_pass[13] / 32.0}},
                "tail": {"mfma": tail_mfma, "ms": tail_ms_t, "executed": tf(tail_mfma, tail_ms_t), "frac": tf(tail_mfma, tail_ms_t) / peak, "share_of_device_ms": tail_ms_t / dev_ms,
                         "strip_tail_passes": strip_passes, "strip_tail_sites_per_pass": served[2] / strip_passes if strip_passes else None,
                         "mfma_per_site": {"tail_kernel_r": tail_r_site, "tail_kernel_p + tail_fc_kernel": strip_mfma / served[2] if strip_passes and served[2] else None},
                         "note": "tail_ms of the strip tail = its class sort (memset + 4 small kernels per launch) + tail_kernel_p (conv5 .. conv8) + tail_fc_kernel (fc1, fc2, softmax)"},
                # this rank's device: all MFMAs of the timed region over its WALL time (copies, staging and scanner kernels included)
                "whole_device": {"mfma": n_mfma + edge_mfma + tail_mfma, "executed_over_timed_region": tf(n_mfma + edge_mfma + tail_mfma, dt * 1e3),
                                 "frac_over_timed_region": tf(n_mfma + edge_mfma + tail_mfma, dt * 1e3) / peak},
                "unit": "TFLOP/s", "peak": peak}
        else:
            front_launches = sum(tm["front_launches"])
            achieved = flop_front_sites(tm["front_sites"]) / (front_ms * 1e-3) / 1e12 if front_ms > 0 else 0.0
            # executed fp16 products per algorithmic MAC: 3 (hi*hi, hi*lo, lo*hi), except conv1, whose operand is exact
            # fp16 since bn0 is folded into its weights: 2 (1 stacked product over a doubled K)
            exec_macs = sum((3.0 * MAC_FRONT[c] - CONV1_MAC[c]) * tm["front_sites"][c] for c in range(3)) if split else 0.0
            products = exec_macs / max(1.0, sum(MAC_FRONT[c] * tm["front_sites"][c] for c in range(3))) if split else 1.0
            roof = {"bound": "mfma", "unit": "TFLOP/s", "achieved": achieved, "peak": peak, "frac": achieved / peak,
                    "avg_launch_ms": front_ms / front_launches if front_launches else None, "launches": front_launches,
                    "traffic": None, "utilisation": None,
                    "traffic_note": "HBM bytes need separate rocprofv3 --pmc passes; see profiles/ for the per-launch figures",
                    "kernel": ("front_kernel_h (window+bn0+conv1..conv4 per site, v_mfma_f32_16x16x32_f16 x3 split-half, fp32 accumulate)"
                               if split else "front_kernel (window+bn0+conv1..conv4 per site, v_mfma_f32_16x16x4_f32)")}
            if split:
                roof.update(executed=products * achieved, frac_executed=products * achieved / peak, utilisation=products * achieved / peak,
                            products_per_mac=products,
                            vs_fp32_mfma_peak=achieved / PEAK_FP32_MFMA_TFLOPS)
        gpu_ms = {k: tm[k] for k in ("prep_ms", "scan_ms", "emit_ms", "pack_ms", "empty_ms")}
        gpu_ms["front_ms"] = front_ms
        gpu_ms["trunk_ms"] = trunk_ms
        gpu_ms["edge_ms"] = sum(tm["edge_ms"])
        gpu_ms["tail_ms"] = sum(tm["tail_ms"])
        gpu_ms["empty_launches"] = tm["empty_launches"]
        gpu_ms["trunk_launches"] = list(tm["trunk_launches"])          # per context (CpG, CHG: one strand view; CHH: two)
        gpu_ms["trunk_positions"] = list(tm["trunk_positions"])        # view positions the trunk launches covered, per context
        out = {
            "metric": "cytosine sites/sec (CpG+CHG+CHH)",
            "value": sites_all / dt_max,
            "unit": "sites/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt_max / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": {0: "f32", 1: "f16x3+f32acc", 2: "f16x3+f32acc; conv8 and fc1 f16x2 (plain fp16 weights: w_lo x_hi dropped)"}[args.precision],
            "cnn_path": ("dense trunk (conv1..conv4 once per read position) + per-site edge rows + tail" if trunk_ms > 0 and front_ms == 0
                         else "per context: dense trunk + edge rows + tail, or per-site front + tail kernels (config.kernel_path_by_context)" if trunk_ms > 0
                         else "per site (front + tail kernels)"),
            "data": "synthetic",
            "config": {"workload": f"{args.workload}: streamed, every step stages a fresh slab of {args.reads} synthetic HiFi reads (~15 kb "
                                   f"log-normal, codev1 kinetics; {wl['what']}) through "
                                   "hm_batch_submit_read -> async H2D -> scan + windows + CNN -> packed D2H, double-buffered; "
                                   "staging and both copies inside the timed region",
                       "group_bases": GROUP_BASES, "group_bytes": int(tm["group_bytes"]),
                       "group_size_note": "group_bases = map rows per trunk read group, default a quarter of the device's FREE memory (12 Mi on an empty "
                                          "288 GB part: group_bytes of HBM held, per rank and per GPU -- not host memory); same-box A/B 2 Mi against the "
                                          "default: 60.65 / 61.10 against 61.60 / 61.53 M sites/s = +1.0 % (profiles/r05_ab_group.txt); option group_bases "
                                          "sets it by hand",
                       "kernel_path_by_context": {n: ("dense trunk" if tm["trunk_ms"][c] > 0 else "per site") for c, n in enumerate(("CpG", "CHG", "CHH"))},
                       "reads_per_step": args.reads, "distinct_slabs": n_pool, "staging_threads": mc.stage_threads,
                       "bases_per_gpu": int(bases_job), "sites_per_gpu": int(sites_job),
                       "sites_per_gpu_step": int(sites_job / max(1, args.steps)),
                       "timed_region_s": dt_max,
                       "fraction_of_configs2": sites_all / CONFIG2_SITES if args.workload == "arabidopsis" else None,
                       "fraction_of_configs3_rank_slice": bases_job / 11.6e9 if args.workload == "human_slice" else None,
                       "sites_by_context": {"CpG": sites_ctx[0], "CHG": sites_ctx[1], "CHH": sites_ctx[2]},
                       "parallelism": f"read-sharded x{world}, no collective",
                       "launch": ("one process per GPU via torch.distributed.run (bench.py --gpus N starts the ranks itself when no torchrun "
                                  "environment is present)" if world > 1 else "single process")},
            "ranks": {"world": world, "backend": (backend if dist is not None else None), "rccl_ranks": world if on_gpu_collectives else 0,
                      "sites_per_s": [r[0] / r[1] for r in by_rank], "seconds": [r[1] for r in by_rank], "device": [int(r[2]) for r in by_rank],
                      "device_ms": [{"trunk_ms": r[3], "edge_ms": r[4], "tail_ms": r[5]} for r in by_rank],
                      "slowest_over_mean_seconds": max(r[1] for r in by_rank) / (sum(r[1] for r in by_rank) / len(by_rank))},
            "roofline": roof,
            "device_ms_timed_region": gpu_ms,
            "effective_tflops_all_layers": sum(2.0 * MAC_TOTAL[c] * sites_ctx[c] for c in range(3)) / dt_max / 1e12,
        }
        out.update(extras)
        if world == 1 and not args.no_extras and not args.no_e2e:
            mc.close()   # the CLI is a process of its own on the same device
            out["end_to_end"] = end_to_end(slabs, args.e2e_reads, extra_flags=("-Z",) if args.e2e_cycles <= 2 else (), cycles=args.e2e_cycles)
        if want_parity and streamed:
            what = (f"(records taken from the STREAMED run of that slab: reads of trunk groups 0, {n_groups // 2} and {n_groups - 1} "
                    f"of its {n_groups} groups per context) ")
            out["cpu_baseline"], out["parity"] = cpu_baseline([(i, slabs[0][i]) for i in sample_ids], streamed,
                                                              tol=1e-3 if args.precision == 2 else 1e-4, what=what)
            out["parity"]["trunk_groups_in_slab"] = n_groups
            out["speedup_vs_cpu_baseline"] = out["value"] / out["cpu_baseline"]["value"]
            # external sanity bound, DERIVED not measured: the reference README's "~2 hours on 48
            # CPU threads" for 30x Arabidopsis, ~1.1e9 sites => ~1.5e5 sites/s; the north-star asks for >= 30x of it
            out["readme_derived_48_thread_baseline"] = {"value": 1.5e5, "unit": "sites/s", "label": "derived, not measured",
                                                        "ratio": out["value"] / 1.5e5}
    # ---- the host feed of configs[3]: one BAM, all ranks pulling parts from the work queue.  Every rank frees its device first; the
    # other ranks wait at the barrier (no timeout on the wait itself: the leg is bounded by the file, ~10 s per 5 GB and rank)
    if args.e2e_dist:
        mc.close()
        if dist is not None:
            dist.barrier()
        if rank == 0:
            try:
                out["end_to_end_dist"] = end_to_end_dist(slabs, args.e2e_reads * (2 if world >= 4 else 1), args.e2e_dist_ranks or world,
                                                         cycles=max(args.e2e_cycles, 3))
            except Exception as ex:  # noqa: BLE001  (a failed extra leg must not void the line)
                out["end_to_end_dist"] = {"error": repr(ex)}
        if dist is not None:
            dist.barrier()
    if rank == 0:
        print(json.dumps(out), flush=True)
    mc.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
