"""world_size-2 gloo test of the N>1 path: variable-length all-gather of packed (s, pi, z) records,
plus pack/unpack round trip of the device record layout."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from azsp.distributed import all_gather_variable, pack_records, record_bytes, unpack_records


def _records(rank: int, n: int):
    rng = np.random.RandomState(100 + rank)
    states = rng.randint(-1, 2, size=(n, 42)).astype(np.int8)
    counts = rng.randint(0, 30, size=(n, 7))
    counts[:, 0] += 1
    policies = counts / counts.sum(axis=1, keepdims=True)
    values = rng.randint(-1, 2, size=n).astype(np.int32)
    return states, policies, values


def _worker(rank: int, world: int, port: int, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        n = [5, 13][rank]
        local = torch.from_numpy(pack_records(*_records(rank, n)))
        parts = all_gather_variable(local, dist, world)
        ok = len(parts) == world
        for r, part in enumerate(parts):
            s, p, v = unpack_records(part.numpy(), 42, 7)
            es, ep, ev = _records(r, [5, 13][r])
            ok = ok and np.array_equal(s, es) and np.array_equal(p, ep) and np.array_equal(v, ev)
        empty = all_gather_variable(local[:0] if rank == 0 else local, dist, world)     # a rank with nothing
        ok = ok and empty[0].shape[0] == 0 and empty[1].shape[0] == 13
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


def _bench_worker(rank: int, world: int, port: int, q):
    """bench.py's own post-step exchange (combine_ranks) on the all-gathered records of two ranks."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        cpu = torch.device("cpu")
        n = [7, 11][rank]
        local = torch.from_numpy(pack_records(*_records(rank, n)))
        parts = all_gather_variable(local, dist, world)
        got = sum(int(p.shape[0]) for p in parts)
        out = bench.combine_ranks(dist, world, cpu, dt=1.0 + rank, episodes=3 + rank, samples=n, gathered_records=got, gather_s=0.01 * (rank + 1))
        ok = (out["dt"] == 2.0 and out["episodes"] == 7 and out["samples"] == 18 and out["per_rank_s"] == [1.0, 2.0]
              and out["allgather_s"] == [0.01, 0.02])
        try:                                   # a rank that did not receive everyone's records must stop the bench
            bench.combine_ranks(dist, world, cpu, dt=1.0, episodes=1, samples=n, gathered_records=got - 1, gather_s=0.0)
            ok = False
        except SystemExit:
            pass
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


def test_bench_rank_combination_gloo_world2():
    """The N > 1 leg of bench.py (max-over-ranks time, summed counts, per-rank times, all-gather completeness check)
    with two gloo ranks on the CPU; the single-rank form is the identity."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    one = bench.combine_ranks(None, 1, None, 2.5, 4, 40, 0, 0.0)
    assert one["dt"] == 2.5 and one["episodes"] == 4 and one["samples"] == 40 and one["per_rank_s"] == [2.5]
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_bench_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = dict(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert results == {0: True, 1: True}


def test_pmc_traffic_is_reported_only_for_the_measured_sources(tmp_path, monkeypatch):
    """roofline.traffic comes from the committed PMC summary; it must go null when the kernels changed since."""
    import json
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    sha = bench.kernel_source_sha()
    assert len(sha) == 16 and sha == bench.kernel_source_sha()
    (tmp_path / "profiles").mkdir()
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    monkeypatch.setattr(bench, "kernel_source_sha", lambda: sha)
    (tmp_path / "profiles" / "pmc_traffic.json").write_text(json.dumps({"_source_sha16": sha, "k": {"bytes_per_launch": 12.5}}))
    assert bench.pmc_traffic("k") == 12.5 and bench.pmc_traffic("other") is None
    (tmp_path / "profiles" / "pmc_traffic.json").write_text(json.dumps({"_source_sha16": "0" * 16, "k": {"bytes_per_launch": 12.5}}))
    assert bench.pmc_traffic("k") is None


def test_pack_unpack_roundtrip():
    s, p, v = _records(0, 17)
    rec = pack_records(s, p, v)
    assert rec.shape == (17, record_bytes(42, 7)) == (17, 104)
    s2, p2, v2 = unpack_records(rec, 42, 7)
    assert np.array_equal(s, s2) and np.array_equal(p, p2) and np.array_equal(v, v2)


def test_all_gather_variable_gloo_world2():
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = dict(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert results == {0: True, 1: True}


def _run_bench(args, env_extra, timeout=300):
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra)
    return subprocess.run([sys.executable, os.path.join(root, "bench.py"), *args], env=env, capture_output=True, text=True, timeout=timeout)


def test_bench_gpus2_without_a_launcher_starts_two_ranks_and_prints_one_line():
    """`python bench.py --gpus 2` with no launcher environment must not measure one GPU and call it two: it starts its
    own two rank processes (torch.distributed.run child, before anything touches a GPU) and rank 0 prints ONE JSON line
    with n_gpus 2.  Run here in rehearsal mode (AZSP_BENCH_REHEARSE=1: fabricated records, gloo, no engine), which goes
    through the real launcher, rendezvous, fences, variable-length all-gather and rank combination."""
    import json
    r = _run_bench(["--gpus", "2", "--steps", "2", "--warmup", "1", "--slots", "8"], {"AZSP_BENCH_REHEARSE": "1"})
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["rehearsal"] is True and out["steps"] == 2
    assert len(out["ranks"]["ms_per_step"]) == 2 and out["config"]["launched_by"].startswith("bench.py")
    assert out["config"]["episodes"] == 2 * 2 * 8            # both ranks' episodes, both steps
    assert out["value"] == 0.0 and "REHEARSAL" in out["data"]


def test_bench_never_reports_more_gpus_than_it_ran_on():
    """--gpus N is refused (non-zero exit, no JSON line) when fewer than N GPUs are visible, and when a launcher started
    a different number of ranks than --gpus names (the round-2 form silently measured one GPU)."""
    import torch
    if torch.cuda.device_count() < 2:
        r = _run_bench(["--gpus", "2"], {})
        assert r.returncode != 0 and "GPU(s) visible" in r.stderr and "{" not in r.stdout
    r = _run_bench(["--gpus", "2"], {"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "WORLD_SIZE=1" in r.stderr and "{" not in r.stdout
