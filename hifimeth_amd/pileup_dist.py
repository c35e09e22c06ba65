"""`hifimeth pileup` over N GPUs of one node, one process per GPU (the path's only exchange step).

    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \\
        -m hifimeth_amd.pileup_dist [-q mapQ] [-f identity] reference.fa mod.bam output-prefix

Records are dealt to the ranks in slabs of `--slab` records (round-robin, like the `call` path).  Each rank projects its
records and histograms them on its own GPU; then
  1. all-reduce(sum) of the 3 x 256 histograms (6 KB)           -> every rank resolves the same thresholds,
  2. per-rank counting into planes laid out as world x chunk loci,
  3. reduce-scatter(sum) of pcov / ncov and reduce-scatter(max) of the motif key over RCCL
                                                                 -> rank r owns loci [r*chunk, (r+1)*chunk),
  4. every rank compacts and formats its range; rank 0 concatenates the parts in rank order (= locus order).
A single process (no torchrun) runs the same code with the collectives skipped.
"""
from __future__ import annotations

import argparse
import os
import sys

import numpy as np

from . import dist as D
from .bamio import is_coordinate_sorted, load_fasta, read_bam
from .pileup import CTX_NAMES, MethylationPileup, allreduce_histograms, locus_ranges, reduce_scatter_planes, resolve_threshold


def run(reference: str, bam: str, prefix: str, min_mapq: int = 0, min_pi: float = 0.0, slab: int = 256,
        batch: int = 256, backend: str | None = None, log=sys.stderr):
    import torch
    rank, local_rank, world = D.env_world()
    dist = D.init_process_group(backend, force=bool(os.environ.get("HM_FORCE_COLLECTIVES")))
    on_gpu = dist is None or dist.get_backend() == "nccl"
    text, refs, records = read_bam(bam)
    def leave(code):
        if dist is not None:
            dist.destroy_process_group()
        return code

    if not refs or not is_coordinate_sorted(text):          # s_bam_is_mapped_and_sorted (pileup.cpp:438-459)
        if rank == 0:
            print("ERROR: Methylation frequency could not be computed due to the following errors:", file=log)
            if not refs:
                print("BAM is not mapped", file=log)
            if not is_coordinate_sorted(text):
                print("BAM is not sorted", file=log)
        return leave(1)                                      # every rank sees the same header: all leave together
    genome = load_fasta(reference)
    sid_of = {n: i for i, (n, _) in enumerate(genome)}
    missing = None
    n_loci = sum(len(s) for _, s in genome)
    ranges = locus_ranges(n_loci, world)
    chunk = max(1, (n_loci + world - 1) // world)
    ndev = max(torch.cuda.device_count(), 1)
    dev = torch.device("cuda", local_rank % ndev)
    torch.cuda.set_device(dev)
    planes = [torch.zeros(world * chunk, dtype=torch.int32, device=dev) for _ in range(3)]
    torch.cuda.synchronize()
    pu = MethylationPileup(genome, device=dev.index, min_mapq=min_mapq, min_pi=min_pi, planes=planes)
    staged = 0
    for order, rec in enumerate(records):
        if (order // slab) % world != rank or rec.flag & 4 or rec.mm is None:
            continue
        name = refs[rec.tid][0]
        if name not in sid_of:                               # only the rank that owns the record sees it: flag, do not exit
            missing = name
            break
        rec.tid = sid_of[name]
        staged += pu.add(rec, order=order)
        if staged >= batch:
            pu.flush()
            staged = 0
    pu.flush()
    # a failed rank must not leave the others waiting in the collectives below: agree on the error first
    bad = torch.tensor([1 if missing else 0], dtype=torch.int32, device=dev if on_gpu else "cpu")
    if dist is not None:
        dist.all_reduce(bad, op=dist.ReduceOp.MAX)
    if int(bad.item()):
        if missing:
            print(f"ERROR: Sequence name {missing} does not exist", file=log)
        pu.close()
        return leave(1)
    bins = allreduce_histograms(dist, pu.histograms(), device=str(dev) if on_gpu else "cpu") if dist is not None \
        else pu.histograms()
    thr = []
    for c in range(3):
        t, samples = resolve_threshold(bins[c])
        thr.append(t)
        if rank == 0:
            print(f"{CTX_NAMES[c]} samples: {samples}\n{CTX_NAMES[c]} scaled probability threshold: {t}", file=log)
    pu.count(thr)
    torch.cuda.synchronize()
    if dist is not None:
        if not on_gpu:                                      # gloo rehearsal: collectives on host copies
            host = [t.cpu() for t in planes]
            pc, nc, key, base = reduce_scatter_planes(dist, *host)
            pc, nc, key = (t.to(dev) for t in (pc, nc, key))
        else:
            pc, nc, key, base = reduce_scatter_planes(dist, *planes, force=True)
        torch.cuda.synchronize()
    else:
        pc, nc, key, base = planes[0], planes[1], planes[2], 0
    lo, hi = ranges[rank]
    loci = pu.loci(0, hi - lo, planes=(pc, nc, key), plane_base=base)
    part = pu.bed(loci)
    if dist is not None:
        parts = [None] * world if rank == 0 else None
        dist.gather_object(part, parts, dst=0)
    else:
        parts = [part]
    if rank == 0:
        for c in CTX_NAMES:
            with open(f"{prefix}.{c}.cov.bed", "w") as f:
                for p in parts:
                    f.write(p[c])
    pu.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    return 0


def main(argv=None):
    ap = argparse.ArgumentParser(prog="python -m hifimeth_amd.pileup_dist")
    ap.add_argument("-q", type=int, default=0, help="minimum mapping quality")
    ap.add_argument("-f", type=float, default=0.0, help="minimum alignment identity (percent)")
    ap.add_argument("--slab", type=int, default=256, help="records per slab dealt to a rank")
    ap.add_argument("--backend", default=None, help="nccl (RCCL, default on GPUs) or gloo")
    ap.add_argument("reference")
    ap.add_argument("mod_bam")
    ap.add_argument("output_prefix")
    a = ap.parse_args(argv)
    return run(a.reference, a.mod_bam, a.output_prefix, a.q, a.f, slab=a.slab, backend=a.backend)


if __name__ == "__main__":
    sys.exit(main())
