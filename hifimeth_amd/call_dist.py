"""`hifimeth call` over N GPUs of one node, one process per GPU, read-sharded by BGZF offset.

    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \\
        -m hifimeth_amd.call_dist [hifimeth call options] in.bam out.bam

Every read is independent (reference mod_main.cpp:180-212), so `call` needs no data-path collective: rank r runs the
native front end (`hifimeth-hip call -R r/N -d <local gpu>`) on ITS byte range of the input -- it inflates, stages and calls
only its own reads -- and writes out.bam.shard<r>; after a barrier rank 0 joins the shards in rank order, which is input
order (the order the reference writes in: mod_main.cpp:352-362).  The native program runs as a CHILD process of the rank
(the rank itself only needs torch.distributed for a CPU barrier -- gloo unless HM_DIST_BACKEND says otherwise -- and never
touches the GPU), never via exec.
`--copy` replaces `call` by `bamcopy` (decode + re-encode, no GPU): the sharding / merge logic on a CPU-only box.
"""
from __future__ import annotations

import os
import subprocess
import sys

from . import dist as D

CLI = os.path.join(os.path.dirname(os.path.abspath(__file__)), "bin", "hifimeth-hip")


def run(argv, backend: str | None = None) -> int:
    copy = "--copy" in argv
    argv = [a for a in argv if a != "--copy"]
    if len(argv) < 2:
        print(__doc__, file=sys.stderr)
        return 2
    out = argv[-1]
    rank, local_rank, world = D.env_world()
    # The ranks only exchange a failure flag (which doubles as the barrier before the merge): a CPU collective.  gloo
    # keeps this process off the GPU altogether -- the child that does the work owns the device -- and works on any box;
    # an explicit "nccl" is honoured (the flag then lives on this rank's device).
    backend = backend or "gloo"
    dist = D.init_process_group(backend)
    rc = 0
    try:
        shard = ["-R", f"{rank}/{world}"]
        if copy:
            cmd = [CLI, "bamcopy"] + shard + argv[-2:]
        else:
            ndev = 1
            try:
                import torch
                ndev = max(torch.cuda.device_count(), 1)   # counting devices does not initialise the GPU
            except Exception:  # noqa: BLE001
                pass
            cmd = [CLI, "call"] + argv[:-2] + shard + ["-d", str(local_rank % ndev)] + argv[-2:]
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
            rc = subprocess.call([CLI, "merge", out, str(world)])
    finally:
        if dist is not None:
            dist.destroy_process_group()
    return rc


if __name__ == "__main__":
    raise SystemExit(run(sys.argv[1:], backend=os.environ.get("HM_DIST_BACKEND")))
