"""N>1 host logic on CPU: two gloo ranks shard read slabs, 'call' them (with the CPU oracle standing in
for the device so the test runs here), reduce the job throughput and gather the calls in input order."""
import os
import socket

import numpy as np
import pytest
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
    rows = D.gather_rank_stats(dist, [sum(len(c) for c in calls), 1.0 + rank, rank])   # what bench.py's `ranks` block is made of
    allc = D.gather_calls(dist, calls, mine, len(slabs), CALL_DTYPE)
    if rank == 0:
        q.put((sites, secs, allc.tobytes(), rows))
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
    sites, secs, raw, rows = q.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # every rank's own (sites, seconds, rank) on rank 0, in rank order: the bench line's per-rank figures
    assert len(rows) == 2 and [r[2] for r in rows] == [0.0, 1.0] and [r[1] for r in rows] == [1.0, 2.0] and sum(r[0] for r in rows) == sites
    assert D.gather_rank_stats(None, [3, 0.5]) == [[3.0, 0.5]]
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


def test_gpu_count_without_opening_the_device(monkeypatch, tmp_path):
    """dist.gpu_count_no_init: KFD topology nodes with SIMDs, cut down by *_VISIBLE_DEVICES; -1 where there is no KFD sysfs (this container).
    A launcher (bench.py --gpus N) or a barrier-only rank (call_dist) must not become one more process that holds the GPU."""
    import glob as _glob
    from hifimeth_amd import dist as D
    nodes = []
    for i, simd in enumerate((0, 256, 256, 0, 256)):     # two CPU nodes, three GPUs
        d = tmp_path / str(i)
        d.mkdir()
        (d / "properties").write_text(f"cpu_cores_count {0 if simd else 64}\nsimd_count {simd}\nmem_banks_count 1\n")
        nodes.append(str(d / "properties"))
    monkeypatch.setattr(_glob, "glob", lambda pat: nodes if "kfd" in pat else [])
    for v in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        monkeypatch.delenv(v, raising=False)
    assert D.gpu_count_no_init() == 3
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "0,2")
    assert D.gpu_count_no_init() == 2
    monkeypatch.setattr(_glob, "glob", lambda pat: [])
    assert D.gpu_count_no_init() == -1


def _payload(path):
    """inflated content of a BGZF file (block layout aside, this is what two BAM files have to agree on)"""
    import gzip
    return gzip.open(path, "rb").read()


def test_bgzf_offset_sharding_and_merge_is_identity(tmp_path):
    """`-R r/w` splits the input by BGZF offset (a rank inflates only its own blocks and finds its first record without
    an index); the shards joined in rank order must equal the single-rank output -- for 2, 3 and 7 ranks, on a file whose
    ~50 KB records straddle block boundaries, and on a file with fewer records than ranks."""
    import subprocess
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import bamutil
    from hifimeth_amd.synth import synth_reads
    cli = os.path.join(ROOT, "hifimeth_amd", "bin", "hifimeth-hip")
    for tag, reads in (("big", synth_reads(60, seed=8, median_len=9000, sigma=0.5, frac_short=0.1, frac_missing=0.1, frac_wide=0.2)),
                       ("tiny", synth_reads(2, seed=9, median_len=1500, sigma=0.1))):
        src, one = str(tmp_path / f"{tag}.bam"), str(tmp_path / f"{tag}.one.bam")
        bamutil.reads_to_bam(src, reads, level=1)
        subprocess.check_call([cli, "bamcopy", src, one], stderr=subprocess.DEVNULL)
        want = _payload(one)
        for world in (2, 3, 7):
            out = str(tmp_path / f"{tag}.w{world}.bam")
            counts = []
            for r in range(world):
                p = subprocess.run([cli, "bamcopy", "-R", f"{r}/{world}", src, out], capture_output=True, text=True)
                assert p.returncode == 0, p.stderr
                counts.append(int(p.stderr.split("wrote")[1].split()[0]))
            assert sum(counts) == len(reads)
            if tag == "big":
                assert min(counts) > 0 and max(counts) < len(reads)          # every rank got a share
            subprocess.check_call([cli, "merge", out, str(world)])
            assert _payload(out) == want and not os.path.exists(out + ".shard0")


def _copy_worker(rank, world, port, src, out):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    import sys
    sys.path.insert(0, ROOT)
    from hifimeth_amd import call_dist
    rc = call_dist.run(["--copy", src, out], backend="gloo")
    assert rc == 0


def test_two_rank_call_driver_over_a_bam_file(tmp_path):
    """hifimeth_amd.call_dist as the driver launches it (one process per rank, gloo here): each rank shards the BAM by
    offset and runs the native front end as a child, rank 0 merges after the barrier; output = single-rank output."""
    import subprocess
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import bamutil
    from hifimeth_amd.synth import synth_reads
    cli = os.path.join(ROOT, "hifimeth_amd", "bin", "hifimeth-hip")
    reads = synth_reads(40, seed=18, median_len=6000, sigma=0.4)
    src, one, out = str(tmp_path / "in.bam"), str(tmp_path / "one.bam"), str(tmp_path / "two.bam")
    bamutil.reads_to_bam(src, reads, level=1)
    subprocess.check_call([cli, "bamcopy", src, one], stderr=subprocess.DEVNULL)
    port = _free_port()
    ctx = mp.get_context("spawn")
    ps = [ctx.Process(target=_copy_worker, args=(r, 2, port, src, out)) for r in range(2)]
    for p in ps:
        p.start()
    for p in ps:
        p.join(120)
        assert p.exitcode == 0
    assert _payload(out) == _payload(one)


def _missing_cli_worker(rank, world, port, src, out, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    os.environ.pop("HM_DIST_BACKEND", None)
    import sys
    sys.path.insert(0, ROOT)
    from hifimeth_amd import call_dist
    if rank == 1:
        call_dist.CLI = "/nonexistent/hifimeth-hip"      # this rank cannot start its child
    q.put((rank, call_dist.run(["--copy", src, out])))   # default backend: gloo, no GPU touched


def test_call_driver_rank_that_cannot_start_its_child_fails_the_job_without_a_hang(tmp_path):
    """One rank fails to launch the native front end (OSError): it must still reach the collective, so that the others do not
    block until the process-group timeout; every rank returns non-zero and nothing is merged.  Runs with the DEFAULT backend
    (no HM_DIST_BACKEND): the ranks' barrier is a CPU collective."""
    import subprocess
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import bamutil
    from hifimeth_amd.synth import synth_reads
    src, out = str(tmp_path / "in.bam"), str(tmp_path / "out.bam")
    bamutil.reads_to_bam(src, synth_reads(6, seed=3, median_len=2000, sigma=0.2), level=1)
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_missing_cli_worker, args=(r, 2, port, src, out, q)) for r in range(2)]
    for p in ps:
        p.start()
    for p in ps:
        p.join(120)
        assert p.exitcode == 0
    rcs = dict(q.get(timeout=5) for _ in range(2))
    assert rcs[0] != 0 and rcs[1] != 0 and not os.path.exists(out)


def _call_dist_worker(rank, world, port, src, out, extra, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    os.environ.pop("HM_DIST_BACKEND", None)
    import sys
    sys.path.insert(0, ROOT)
    from hifimeth_amd import call_dist
    q.put((rank, call_dist.run(["--copy"] + extra + [src, out])))


@pytest.mark.parametrize("mode", ["queue", "static"])
def test_eight_rank_call_driver(tmp_path, mode):
    """hifimeth_amd.call_dist with EIGHT ranks (gloo; `--copy`: decode + re-encode instead of the GPU), as a node of eight GPUs
    would run it.  Default: the ranks pull the parts of the input from the shared counter (32 parts here; a fast rank takes
    more) -- `--static`: every rank takes its one byte range.  Either way the merged output equals the single-process output,
    in input order, and the queue file is gone afterwards."""
    import subprocess
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import bamutil
    from hifimeth_amd.synth import synth_reads
    cli = os.path.join(ROOT, "hifimeth_amd", "bin", "hifimeth-hip")
    reads = synth_reads(150, seed=28, median_len=5000, sigma=0.5)
    src, one, out = str(tmp_path / "in.bam"), str(tmp_path / "one.bam"), str(tmp_path / "eight.bam")
    bamutil.reads_to_bam(src, reads, level=1)
    subprocess.check_call([cli, "bamcopy", src, one], stderr=subprocess.DEVNULL)
    if mode == "queue":
        open(out + ".queue", "w").write("7\n")    # a counter left behind by a killed job: must not make the ranks skip parts
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_call_dist_worker, args=(r, 8, port, src, out, ["--static"] if mode == "static" else [], q)) for r in range(8)]
    for p in ps:
        p.start()
    for p in ps:
        p.join(300)
        assert p.exitcode == 0
    assert all(rc == 0 for _, rc in (q.get(timeout=5) for _ in range(8)))
    assert _payload(out) == _payload(one)
    assert not os.path.exists(out + ".queue") and not os.path.exists(out + ".shard0")


def test_queue_parts_are_claimed_exactly_once(tmp_path):
    """`-Q file -C n`: three processes running at the same time claim the n parts of the input from one counter file (flock):
    every part exactly once, whoever is faster takes more; the merged parts are the input."""
    import subprocess
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import bamutil
    from hifimeth_amd.synth import synth_reads
    cli = os.path.join(ROOT, "hifimeth_amd", "bin", "hifimeth-hip")
    reads = synth_reads(90, seed=29, median_len=6000, sigma=0.5)
    src, one, out, qf = str(tmp_path / "in.bam"), str(tmp_path / "one.bam"), str(tmp_path / "q.bam"), str(tmp_path / "counter")
    bamutil.reads_to_bam(src, reads, level=1)
    subprocess.check_call([cli, "bamcopy", src, one], stderr=subprocess.DEVNULL)
    n = 13
    ps = [subprocess.Popen([cli, "bamcopy", "-Q", qf, "-C", str(n), src, out], stderr=subprocess.PIPE, text=True) for _ in range(3)]
    took, wrote = [], []
    for p in ps:
        err = p.communicate(timeout=120)[1]
        assert p.returncode == 0, err
        took.append(int(err.split("took")[1].split()[0]))
        wrote.append(int(err.split("wrote")[1].split()[0]))
    assert sum(took) == n and sum(wrote) == len(reads)
    assert int(open(qf).read()) == n + 3                    # every process made one claim past the end
    subprocess.check_call([cli, "merge", out, str(n)])
    assert _payload(out) == _payload(one)


def test_queue_failures_are_errors_and_a_single_part_keeps_its_shard_name(tmp_path):
    """a rank that cannot reach the counter file (here: a directory that does not exist) must fail, not exit 0 having
    done nothing; and a queue run writes OUT.shard<k> even when the queue has one part (`merge OUT 1` then finds it)."""
    import subprocess
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import bamutil
    from hifimeth_amd.synth import synth_reads
    cli = os.path.join(ROOT, "hifimeth_amd", "bin", "hifimeth-hip")
    reads = synth_reads(12, seed=31, median_len=3000, sigma=0.3)
    src, one, out = str(tmp_path / "in.bam"), str(tmp_path / "one.bam"), str(tmp_path / "q.bam")
    bamutil.reads_to_bam(src, reads, level=1)
    subprocess.check_call([cli, "bamcopy", src, one], stderr=subprocess.DEVNULL)
    p = subprocess.run([cli, "bamcopy", "-Q", str(tmp_path / "no_such_dir" / "counter"), "-C", "4", src, out], stderr=subprocess.PIPE, text=True)
    assert p.returncode != 0 and "work queue" in p.stderr, p.stderr
    subprocess.check_call([cli, "bamcopy", "-Q", str(tmp_path / "counter1"), "-C", "1", src, out], stderr=subprocess.DEVNULL)
    assert os.path.exists(out + ".shard0") and not os.path.exists(out)
    subprocess.check_call([cli, "merge", out, "1"])
    assert _payload(out) == _payload(one)


def test_many_queue_parts_per_process_scan_the_input_once(tmp_path):
    """a process that takes many parts of one input must not re-scan the file's BGZF blocks per part (one seek and
    two reads per block, times the parts: more than the GPU work of a large input).  One process takes all 40 parts: the output is the
    input and the blocks were scanned once."""
    import subprocess
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import bamutil
    from hifimeth_amd.synth import synth_reads
    cli = os.path.join(ROOT, "hifimeth_amd", "bin", "hifimeth-hip")
    reads = synth_reads(300, seed=33, median_len=5000, sigma=0.4)
    src, one, out, qf = str(tmp_path / "in.bam"), str(tmp_path / "one.bam"), str(tmp_path / "q.bam"), str(tmp_path / "counter")
    bamutil.reads_to_bam(src, reads, level=1)
    subprocess.check_call([cli, "bamcopy", src, one], stderr=subprocess.DEVNULL)
    n = 40
    p = subprocess.run([cli, "bamcopy", "-Q", qf, "-C", str(n), src, out], stderr=subprocess.PIPE, text=True)
    assert p.returncode == 0, p.stderr
    assert "took 40 of 40 parts" in p.stderr and "block scans 1" in p.stderr, p.stderr
    subprocess.check_call([cli, "merge", out, str(n)])
    assert _payload(out) == _payload(one)
