"""`hifimeth call` over N GPUs of one node, one process per GPU, reads sharded over the ranks.

    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \\
        -m hifimeth_amd.call_dist [--static] [hifimeth call options] in.bam out.bam

Every read is independent (reference mod_main.cpp:180-212), so `call` needs no data-path collective.  The input is cut by
COMPRESSED bytes at BGZF block boundaries into parts (one per 256 MB, at least four per rank); every rank runs the native
front end (`hifimeth-hip call -Q <queue> -C <parts> -d <local gpu>`) as a CHILD process, which PULLS parts from a shared
counter until none is left -- the reference's work queue (src/corelib/sam_batch.hpp:38-54: whichever worker is free takes the
next reads) stretched over processes: site density and read length drift along a real file, so equal byte ranges are not
equal work (tools/shard_balance.py: the slowest of 8 ranks carries 12 % more sites than the mean on a file whose GC drifts
from 0.30 to 0.45 and whose reads double in length; 3 % with the queue).  Part k is
written to out.bam.shard<k>; after a barrier rank 0 joins the parts in order, which is input order (the order the reference
writes in: mod_main.cpp:352-362).  `--static` gives every rank the one fixed part `-R rank/world` names instead.
The rank itself only needs torch.distributed for a CPU barrier -- gloo unless HM_DIST_BACKEND says otherwise -- and never
touches the GPU; the native program is never exec'ed.
`--copy` replaces `call` by `bamcopy` (decode + re-encode, no GPU): the sharding / queue / merge logic on a CPU-only box.
"""
from __future__ import annotations

import os
import subprocess
import sys

from . import dist as D

CLI = os.path.join(os.path.dirname(os.path.abspath(__file__)), "bin", "hifimeth-hip")


PART_BYTES = 256 << 20


def host_cores() -> int:
    """CPUs this process may use: affinity mask capped by the cgroup CPU quota (what the native front end's own -t default counts)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return n


def n_parts(path: str, world: int) -> int:
    """Parts of the queue: one per 256 MB of BAM, at least four per rank (every rank computes the same number)."""
    try:
        size = os.path.getsize(path)
    except OSError:
        size = 0
    return max(4 * world, (size + PART_BYTES - 1) // PART_BYTES, 1)


def run(argv, backend: str | None = None) -> int:
    copy = "--copy" in argv
    static = "--static" in argv
    argv = [a for a in argv if a not in ("--copy", "--static")]
    if len(argv) < 2:
        print(__doc__, file=sys.stderr)
        return 2
    src, out = argv[-2], argv[-1]
    rank, local_rank, world = D.env_world()
    # The ranks only exchange a failure flag (which doubles as the barrier before the merge): a CPU collective.  gloo
    # keeps this process off the GPU altogether -- the child that does the work owns the device -- and works on any box;
    # an explicit "nccl" is honoured (the flag then lives on this rank's device).
    backend = backend or "gloo"
    dist = D.init_process_group(backend)
    rc = 0
    try:
        queue, parts = out + ".queue", world
        if static or dist is None:
            shard = ["-R", f"{rank}/{world}"]
        else:
            parts = n_parts(src, world)
            if rank == 0 and os.path.exists(queue):
                os.remove(queue)   # a counter left behind by a killed job would make every rank skip the first parts
            dist.barrier()
            shard = ["-Q", queue, "-C", str(parts)]
        if copy:
            cmd = [CLI, "bamcopy"] + shard + argv[-2:]
        else:
            ndev = max(D.gpu_count(), 1)   # KFD sysfs: this rank never opens the device, its native child owns it
            opts = argv[:-2]
            if "-t" not in opts and world > 1:
                # the ranks of a node share its cores: each native child gets its share (the CLI's own default is every core it may use)
                opts = opts + ["-t", str(max(1, host_cores() // world))]
            cmd = [CLI, "call"] + opts + shard + ["-d", str(local_rank % ndev)] + argv[-2:]
        try:
            rc = subprocess.call(cmd)
        except OSError as ex:   # e.g. the native front end is missing: every rank must still reach the collective below
            print(f"[call_dist] rank {rank}: cannot run {cmd[0]}: {ex}", file=sys.stderr)
            rc = 127
        if dist is not None:
            import torch
            dev = torch.device("cuda", local_rank) if backend == "nccl" else torch.device("cpu")
            flag = torch.tensor([rc != 0], dtype=torch.int32, device=dev)
            dist.all_reduce(flag, op=dist.ReduceOp.MAX)   # doubles as the barrier before the merge
            rc = int(flag.item()) or rc
        if rc == 0 and rank == 0 and world > 1:
            rc = subprocess.call([CLI, "merge", out, str(parts)])
        if rank == 0 and os.path.exists(queue):
            os.remove(queue)
    finally:
        if dist is not None:
            dist.destroy_process_group()
    return rc


if __name__ == "__main__":
    raise SystemExit(run(sys.argv[1:], backend=os.environ.get("HM_DIST_BACKEND")))
