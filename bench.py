#!/usr/bin/env python3
"""bench.py -- cytosine sites/sec (CpG+CHG+CHH) of the hifimeth `call` hot path on MI355X.

One "step" = one pass of the hot path (site scan -> kinetics windows -> CNN -> probabilities)
over one batch of synthetic HiFi reads that is already resident in HBM (BASELINE.json configs[2]
statistics: Arabidopsis-like GC 0.36, read length log-normal around 15 kb, all three contexts).
Each rank drives one GPU with its own batch (reads are independent: no data-path collective,
weak scaling); `value` is the whole-job sites/s = sum of sites over ranks * steps / max time.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--reads R]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# algorithmic work per site (BASELINE.md section 2 / SURVEY.md 8d), 2 FLOP per MAC
MAC_TOTAL = {0: 11148800, 1: 11148800, 2: 11440640}
# share of the dominant kernel (front: conv1..conv4)
MAC_FRONT = {0: 197 * 128 * 88 + 99 * 128 * 384 + 50 * 128 * 384 + 25 * 96 * 384,
             1: 197 * 128 * 88 + 99 * 128 * 384 + 50 * 128 * 384 + 25 * 96 * 384,
             2: 196 * 128 * 104 + 98 * 128 * 384 + 49 * 128 * 384 + 25 * 96 * 384}
PEAK_FP32_MFMA_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32 / 32x32x2_f32 dense peak
PEAK_FP16_MFMA_TFLOPS = 2500.0  # MI355X_MICROARCH.md: BF16/FP16 MFMA ~2.5 PF dense (2:1-sparsity figures excluded)


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


def cpu_baseline(reads, gpu_calls=None, budget_s=15.0):
    """The CPU oracle (port of the reference path) timed on this node's host cores, bounded sample.
    The oracle's results for that sample double as an in-run parity check of the GPU calls."""
    from oracle import hm_oracle as O
    O.build()
    cores = host_cores()
    models = [O.Model(os.path.join(ROOT, "hifimeth_amd", "weights", n + ".hmw")) for n in ("CpG", "CHG", "CHH")]
    t0 = time.perf_counter()
    sites = 0
    nreads = 0
    worst, nml, ncheck = 0.0, 0, 0
    deltas = []
    for rid, rd in enumerate(reads):
        if not rd.has_kinetics() or rd.l_qseq < 1000:
            continue
        t1 = time.perf_counter()
        want = O.call_read(models, 7, rd, nthreads=cores)
        spent = time.perf_counter() - t1
        sites += len(want["qoff"])
        nreads += 1
        if gpu_calls is not None:  # untimed: compare with the GPU's calls for the same read
            t0 += 0.0
            tchk = time.perf_counter()
            got = gpu_calls[gpu_calls["read_id"] == rid]
            order = np.lexsort((want["qoff"], want["strand"]))
            assert len(got) == len(order) and np.array_equal(got["qoff"], want["qoff"][order]), "site lists differ"
            d = np.abs(got["p"] - want["p"][order])
            deltas.append(d)
            worst = max(worst, float(d.max(initial=0)))
            nml += int((got["scaled_prob"] != want["ml"][order]).sum())
            ncheck += len(got)
            t0 += time.perf_counter() - tchk  # keep the comparison out of the timed span
        if time.perf_counter() - t0 > budget_s:
            break
    dt = time.perf_counter() - t0
    out = {"value": sites / dt, "unit": "sites/s", "cores": cores, "kind": "port",
           "sample": f"first {nreads} reads of the rank-0 batch ({sites} sites, all contexts), "
                     f"oracle/hm_oracle.c fp32 ({O.variant} build), OpenMP over sites, {dt:.1f} s"}
    parity = None
    if gpu_calls is not None:
        alld = np.concatenate(deltas) if deltas else np.zeros(1)
        parity = {"sites_checked": ncheck, "max_abs_dp_vs_oracle": worst, "mean_abs_dp": float(alld.mean()),
                  "p999_abs_dp": float(np.quantile(alld, 0.999)), "ml_bytes_off_by_1lsb": nml, "tolerance": 1e-4}
    return out, parity


def roofline(precision, achieved, front_ms, front_launches, products=3.0):
    """Roofline of the dominant kernel. `achieved` = ALGORITHMIC TFLOP/s (2 FLOP per MAC of conv1..conv4).
    The split-half kernel issues three fp16 MFMAs per algorithmic MAC (hi*hi + hi*lo + lo*hi) -- two in conv1, whose
    operand is exact fp16 once bn0 is folded into its weights -- so its executed rate is `products` (~2.76) x the
    algorithmic one; both are given, the peak is the fp16 dense MFMA peak."""
    base = {"bound": "mfma", "unit": "TFLOP/s", "achieved": achieved,
            "avg_launch_ms": front_ms / front_launches if front_launches else None, "launches": front_launches,
            "traffic": None}
    # HBM bytes per launch come from separate rocprofv3 --pmc passes (cannot run inside this process);
    # the committed summary of the last such pass is quoted, with its source, when present
    try:
        pmc = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc_summary.json")))
        if precision >= 1:
            t = pmc["front_kernel_h<13> per launch (65536 sites)"]
            base["traffic"] = t["hbm_read_bytes_corrected"] + t["hbm_write_bytes"]
            base["traffic_source"] = "profiles/r01_pmc_summary.json (front_kernel_h<13>, 65536 sites per launch; FETCH_SIZE x2 corrected)"
    except (OSError, KeyError, ValueError):
        pass
    if precision >= 1:
        base.update(kernel="front_kernel_h (window+bn0+conv1..conv4, v_mfma_f32_16x16x32_f16 x3 split-half, fp32 accumulate)",
                    peak=PEAK_FP16_MFMA_TFLOPS, frac=achieved / PEAK_FP16_MFMA_TFLOPS,
                    executed=products * achieved, frac_executed=products * achieved / PEAK_FP16_MFMA_TFLOPS,
                    products_per_mac=products,
                    vs_fp32_mfma_peak=achieved / PEAK_FP32_MFMA_TFLOPS)
    else:
        base.update(kernel="front_kernel (window+bn0+conv1..conv4, v_mfma_f32_16x16x4_f32)",
                    peak=PEAK_FP32_MFMA_TFLOPS, frac=achieved / PEAK_FP32_MFMA_TFLOPS)
    return base


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--reads", type=int, default=96, help="reads per GPU batch (~15 kb each)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--precision", type=int, default=1, choices=[0, 1, 2],
                    help="front-kernel arithmetic: 1 = split-half fp16x3 MFMA + fp32 accumulate (default), 0 = fp32 MFMA")
    ap.add_argument("--opt", action="append", default=[], help="engine option key=value (e.g. front_waves=8)")
    args = ap.parse_args()

    from hifimeth_amd import MethylationCaller
    from hifimeth_amd import dist as hmdist
    from hifimeth_amd.synth import synth_reads

    rank, local_rank, world = hmdist.env_world()
    # HM_DIST_BACKEND=gloo lets the multi-rank path be rehearsed on a box with fewer GPUs than ranks
    backend = os.environ.get("HM_DIST_BACKEND", "nccl")
    # HM_BENCH_FORCE_DIST=1 builds a world of one, so that the RCCL barrier / reductions run on a one-GPU box too
    force = os.environ.get("HM_BENCH_FORCE_DIST", "0") == "1"
    dist = hmdist.init_process_group(backend, force=force) if (world > 1 or force) else None
    on_gpu_collectives = dist is not None and backend == "nccl"

    reads = synth_reads(args.reads, seed=20250220 + rank, gc=0.36)
    import torch
    ndev = torch.cuda.device_count()
    mc = MethylationCaller(device=local_rank % max(ndev, 1), timing=True)
    mc.set_option("precision", args.precision)
    for kv in args.opt:
        k, v = kv.split("=")
        mc.set_option(k, int(v))
    mc.submit_all(reads)
    mc.upload()          # inputs resident in HBM before the timed region
    mc.sync()

    def barrier():
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        mc.run()
        mc.sync()
    mc.timing(reset=True)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        mc.run()
    mc.sync()
    barrier()
    dt = time.perf_counter() - t0

    sites_ctx = [mc.num_sites(c) for c in range(3)]
    sites_step = sum(sites_ctx)
    tm = mc.timing()
    bases = sum(r.l_qseq for r in reads if r.has_kinetics() and r.l_qseq >= 1000)

    sites_all, dt_max = hmdist.job_throughput(dist, sites_step, dt, device="cuda" if on_gpu_collectives else "cpu")

    gpu_calls = mc.fetch() if (rank == 0 and world == 1 and not args.no_cpu_baseline) else None

    # feature extraction measured standalone (outside the timed region): the materialised 401x8 fp32 windows of
    # the largest context, device only.  In the product path the window never leaves LDS; this is the HBM-bound
    # kernel the north-star asks to be priced against the HBM roofline.
    feat = None
    if rank == 0:
        big = int(np.argmax(sites_ctx))
        mc.timing(reset=True)
        for _ in range(3):
            mc.windows(big, 0, sites_ctx[big], fetch=False)
        tw = mc.timing()
        if tw["window_ms"] > 0:
            wbytes = tw["window_sites"] * (401 * 8 * 4 + 401 * 5 + 12)   # window out + raw slice in + site record
            feat = {"kernel": "window_kernel (401x8 fp32 windows to HBM, test/roofline seam)", "bound": "hbm",
                    "achieved": wbytes / (tw["window_ms"] * 1e-3) / 1e9, "peak": 8000.0, "unit": "GB/s",
                    "sites_per_launch": sites_ctx[big], "avg_launch_ms": tw["window_ms"] / tw["window_launches"],
                    "algorithmic_bytes_per_site": 401 * 8 * 4 + 401 * 5 + 12}
            feat["frac"] = feat["achieved"] / feat["peak"]
            prep_b = bases * (4.5 + 1 + 4) + sites_step * 21 + bases            # raw in, packed out, site records, emit re-read
            feat["scan_kernels"] = {"prep_scan_emit_ms_per_step": (tm["prep_ms"] + tm["scan_ms"] + tm["emit_ms"]) / max(1, args.steps),
                                    "achieved_GBps": prep_b / ((tm["prep_ms"] + tm["scan_ms"] + tm["emit_ms"]) / max(1, args.steps) * 1e-3) / 1e9,
                                    "note": f"three launches over {bases / 1e6:.1f} Mbases; launch-latency sized for small batches"}
    if rank == 0:
        front_ms = sum(tm["front_ms"])
        front_launches = sum(tm["front_launches"])
        flop_front = sum(2.0 * MAC_FRONT[c] * tm["front_sites"][c] for c in range(3))
        achieved = flop_front / (front_ms * 1e-3) / 1e12 if front_ms > 0 else 0.0
        # executed fp16 products per algorithmic MAC: 3 (hi*hi, hi*lo, lo*hi), except conv1, whose operand is exact fp16 since
        # bn0 is folded into its weights: 2
        conv1_macs = {0: 197 * 128 * 88, 1: 197 * 128 * 88, 2: 196 * 128 * 104}
        exec_macs = sum((3.0 * MAC_FRONT[c] - (conv1_macs[c] if args.precision >= 1 else 0)) * tm["front_sites"][c]
                        for c in range(3))
        products = exec_macs / max(1.0, sum(MAC_FRONT[c] * tm["front_sites"][c] for c in range(3)))
        gpu_ms = {k: tm[k] for k in ("prep_ms", "scan_ms", "emit_ms")}
        gpu_ms["front_ms"] = front_ms
        gpu_ms["tail_ms"] = sum(tm["tail_ms"])
        out = {
            "metric": "cytosine sites/sec (CpG+CHG+CHH)",
            "value": sites_all * args.steps / dt_max,
            "unit": "sites/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt_max / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": {0: "f32", 1: "f16x3+f32acc", 2: "f16w/f16x2+f32acc"}[args.precision],
            "data": "synthetic",
            "config": {"workload": "synthetic 30x-style HiFi reads (GC 0.36, ~15 kb log-normal, codev1 kinetics), "
                                   "all three contexts, batch resident in HBM; BASELINE.json configs[2] statistics",
                       "reads_per_gpu": args.reads, "bases_per_gpu": int(bases), "sites_per_gpu_step": int(sites_step),
                       "sites_by_context": {"CpG": sites_ctx[0], "CHG": sites_ctx[1], "CHH": sites_ctx[2]},
                       "parallelism": f"read-sharded x{world}, no collective"},
            "roofline": roofline(args.precision, achieved, front_ms, front_launches, products),
            "feature_extraction": feat,
            "device_ms_timed_region": gpu_ms,
            "effective_tflops_all_layers": sum(2.0 * MAC_TOTAL[c] * sites_ctx[c] for c in range(3)) * args.steps / dt_max / 1e12,
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"], out["parity"] = cpu_baseline(reads, gpu_calls)
            out["speedup_vs_cpu_baseline"] = out["value"] / out["cpu_baseline"]["value"]
            # external sanity bound, DERIVED not measured: the reference README's "~2 hours on 48
            # CPU threads" for 30x Arabidopsis, ~1.1e9 sites => ~1.5e5 sites/s; the north-star asks for >= 30x of it
            out["readme_derived_48_thread_baseline"] = {"value": 1.5e5, "unit": "sites/s", "label": "derived, not measured",
                                                        "ratio": out["value"] / 1.5e5}
        print(json.dumps(out))
    mc.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
