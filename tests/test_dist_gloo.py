"""N>1 host logic on CPU: two gloo ranks shard read slabs, 'call' them (with the CPU oracle standing in
for the device so the test runs here), reduce the job throughput and gather the calls in input order."""
import os
import socket

import numpy as np
import torch.multiprocessing as mp

from conftest import ROOT, WEIGHTS


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _oracle_calls(reads, ids, models):
    from hifimeth_amd.caller import CALL_DTYPE
    from oracle import hm_oracle as O
    out = []
    for rid in ids:
        rd = reads[rid]
        if not rd.has_kinetics() or rd.l_qseq < 1000:
            continue
        r = O.call_read(models, 1, rd, nthreads=1)     # CpG only keeps the test fast
        order = np.lexsort((r["qoff"], r["strand"]))
        rec = np.zeros(len(order), CALL_DTYPE)
        rec["read_id"], rec["qoff"], rec["strand"] = rid, r["qoff"][order], r["strand"][order]
        rec["ctx"], rec["scaled_prob"], rec["p"] = r["ctx"][order], r["ml"][order], r["p"][order]
        out.append(rec)
    return np.concatenate(out) if out else np.empty(0, CALL_DTYPE)


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    import sys
    sys.path.insert(0, ROOT)
    from hifimeth_amd import dist as D
    from hifimeth_amd.caller import CALL_DTYPE
    from hifimeth_amd.synth import synth_reads
    from oracle import hm_oracle as O
    dist = D.init_process_group("gloo")
    reads = synth_reads(7, seed=4, median_len=1300, sigma=0.2, frac_short=0.2, frac_missing=0.1)
    models = [O.Model(os.path.join(WEIGHTS, "CpG.hmw")), None, None]
    slabs = D.make_slabs(len(reads), 2)
    mine = D.slab_assignment(len(slabs), rank, world)
    calls = [_oracle_calls(reads, slabs[s], models) for s in mine]
    sites, secs = D.job_throughput(dist, sum(len(c) for c in calls), 1.0 + rank)
    allc = D.gather_calls(dist, calls, mine, len(slabs), CALL_DTYPE)
    if rank == 0:
        q.put((sites, secs, allc.tobytes()))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharding_matches_single_process():
    from hifimeth_amd import dist as D
    from hifimeth_amd.caller import CALL_DTYPE
    from hifimeth_amd.synth import synth_reads
    from oracle import hm_oracle as O
    assert D.slab_assignment(5, 0, 2) == [0, 2, 4] and D.slab_assignment(5, 1, 2) == [1, 3]
    assert [list(r) for r in D.make_slabs(5, 2)] == [[0, 1], [2, 3], [4]]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    sites, secs, raw = q.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    reads = synth_reads(7, seed=4, median_len=1300, sigma=0.2, frac_short=0.2, frac_missing=0.1)
    models = [O.Model(os.path.join(WEIGHTS, "CpG.hmw")), None, None]
    single = _oracle_calls(reads, range(len(reads)), models)
    assert raw == single.tobytes()
    assert sites == len(single) and secs == 2.0      # SUM of sites, MAX of time
    # single-process path of the same helpers
    s1, t1 = D.job_throughput(None, 5, 0.5)
    assert (s1, t1) == (5.0, 0.5)
    g = D.gather_calls(None, [single[:3], single[3:]], [0, 1], 2, CALL_DTYPE)
    assert g.tobytes() == single.tobytes()
