#!/usr/bin/env python3
"""Self-play throughput benchmark (BASELINE.json: "self-play episodes/sec (Connect-4, 100 sims/move)").

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

`--gpus N` with N > 1 and no launcher environment (WORLD_SIZE unset) makes bench.py its own launcher: before
anything touches a GPU it starts `python -m torch.distributed.run --nproc-per-node N bench.py ...` as a child
process (N fresh rank processes), relays rank 0's JSON line and exits with the child's code; it refuses to run
when fewer than N GPUs are visible.  `--gpus` never silently measures fewer GPUs than it names.

A *step* is one complete `EpisodeGenerator.generate_episodes` call of the workload: `--slots`
concurrent Connect-4 games per GPU (default 4096 = BASELINE.json configs[1]), `--sims` MCTS
simulations per move, random-weight ResNet(`--blocks`, `--channels`), run until `slots` episodes have
completed (reference stop rule), examples copied to the host, and -- for N > 1 -- the packed
(s, pi, z) records all-gathered over RCCL.  Games shard over GPUs with no other traffic (weak scaling).

The headline (`value`, `roofline`) is measured with the inference kernel in exact fp32
(v_mfma_f32_32x32x2_f32, `dtype` "f32").  Unless `--single`, the same workload is then measured once more
with the split-fp16 inference kernel (fp32 operands carried as two fp16 numbers, fp32 accumulate; same
accuracy bound in the tests) and reported under `"f16x2"`; `--precision f16x2` makes that the headline.

After those, the same workload is measured with the evaluation cache on (`"eval_cache"`: every position evaluated
at most once per generate_episodes call -- identical episodes, tested; NOT the headline until a verdict accepts it).

The JSON line carries the MFMA roofline of the dominant kernel (timed with HIP events on the engine's
own streams inside the timed region) and a CPU baseline: the oracle's pure-Python/torch-CPU port of the
reference path timed on this box's host cores (rank 0, N=1 only)."""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "alphazero-implementation_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402
import torch  # noqa: E402

PEAK_TFLOPS = {"f32": 157.3,      # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
               "f16x2": 2500.0}   # MI355X_MICROARCH.md: fp16/bf16 MFMA, dense
DTYPE = {"f32": "f32", "f16x2": "f16x2 (fp32 operands split into two fp16, fp32 accumulate)"}
KERNEL = {"f32": "k_resnet_forward", "f16x2": "k_resnet_forward_f16"}


def resnet_flops_per_board(blocks: int, ch: int, cells: int = 42, actions: int = 7) -> float:
    """2*MAC over conv + linear layers of resnet.py:44-72 (25 254 012 for 4 blocks x 64 ch)."""
    conv = cells * (3 * 9 * ch + 2 * blocks * ch * ch * 9 + ch * 32 + ch * 9 * 3)
    fc = 32 * cells * actions + 3 * cells
    return 2.0 * (conv + fc)


def tree_report(st: dict, wall_s: float, n_actions: int = 7) -> dict:
    """Selection/expansion/backup kernels: HBM-bound graph work.  Algorithmic bytes per
    simulation with d levels selected and A children per node: select d*A*20 (N, W, P, child index per child) + leaf
    step 32 + expand A*20 + 24 + path/backup (d + 1)*24; achieved = bytes / summed kernel time (HIP events on the
    engine streams; the kernels of one slot group run concurrently with the other groups' inference)."""
    sims = int(st["evals"] + st["terminal_leaves"])
    depth = st["path_nodes"] / max(sims, 1) - 1.0
    bytes_per_sim = depth * n_actions * 20 + 32 + (n_actions * 20 + 24) + (depth + 1.0) * 24
    k_ms = st["select_ms"] + st["expand_ms"]
    traffic = pmc_traffic("k_expand_select")                       # bytes per launch = one slot group's round
    sims_per_launch = sims / st["select_launches"] if st.get("select_launches") else 0.0
    gbs = sims * bytes_per_sim / (k_ms * 1e-3) / 1e9 if k_ms > 0 else 0.0
    return {"select_ms": st["select_ms"], "expand_ms": st["expand_ms"], "play_ms": st["play_ms"],
            "rounds": int(st["rounds"]), "evals": int(st["evals"]), "terminal_leaves": int(st["terminal_leaves"]),
            "path_nodes": int(st["path_nodes"]), "sims_per_s": sims / wall_s, "evals_per_s": st["evals"] / wall_s,
            "mean_levels_selected": depth,
            "roofline": dict({"bound": "hbm", "kernels": "k_expand_select (expansion + backup of round r-1, descent of round r)",
                              "bytes_per_sim": bytes_per_sim,
                              "achieved": gbs, "peak": 8000.0, "unit": "GB/s", "frac": gbs / 8000.0,
                              "traffic": traffic},
                             **({"traffic_per_sim": traffic / sims_per_launch, "traffic_over_algorithmic": traffic / sims_per_launch / bytes_per_sim}
                                if traffic and sims_per_launch else {}))}


def kernel_source_sha() -> str:
    """Fingerprint of the HIP sources: ties the committed PMC summary to the kernels it was measured on."""
    import hashlib
    h = hashlib.sha256()
    csrc = os.path.join(ROOT, "alphazero-implementation_amd", "csrc")
    for name in sorted(os.listdir(csrc)):
        if name.endswith((".hip", ".h")):
            with open(os.path.join(csrc, name), "rb") as f:
                h.update(name.encode() + b"\0" + f.read())
    return h.hexdigest()[:16]


def pmc_traffic(kernel: str):
    """HBM bytes per launch of the kernel from the committed rocprofv3 PMC summary (separate --pmc passes,
    FETCH_SIZE doubled per the gfx950 correction; tools/pmc_summary.py).  None if no summary is present OR if it was
    measured on other kernel sources than the ones in the tree (the summary records their fingerprint)."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        with open(path) as f:
            d = json.load(f)
    except (OSError, ValueError):
        return None
    if d.get("_source_sha16") != kernel_source_sha():
        return None
    return d.get(kernel, {}).get("bytes_per_launch")


def launch_ranks(n: int, argv: list[str]) -> int:
    """`bench.py --gpus n` started without a launcher: this process has not touched a GPU yet (device_count() does not
    initialise one), so it may start the n rank processes itself -- fresh children through torch.distributed.run, one
    per GPU -- and hand their exit code on.  Rank 0's JSON line goes straight to our stdout."""
    backend = os.environ.get("AZSP_BENCH_BACKEND", "nccl")
    rehearse = os.environ.get("AZSP_BENCH_REHEARSE") == "1"
    if backend == "nccl" and not rehearse:
        n_dev = torch.cuda.device_count()
        if n_dev < n:
            print(f"bench.py: --gpus {n} but only {n_dev} GPU(s) visible: refusing to report a {n}-GPU number measured on fewer "
                  f"(AZSP_BENCH_BACKEND=gloo rehearses the N-rank path on fewer GPUs)", file=sys.stderr)
            return 2
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__), *argv]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env["AZSP_BENCH_SELF_LAUNCHED"] = "1"
    return subprocess.call(cmd, env=env)


class RehearsalGenerator:
    """AZSP_BENCH_REHEARSE=1: no engine, no GPU -- each step fabricates packed records so that the launcher, the
    rendezvous, the timing fences, the variable-length all-gather and combine_ranks can be run end to end on a CPU
    box (tests/test_distributed_cpu.py).  The line it produces says so and carries no throughput."""

    def __init__(self, slots: int, rank: int):
        self.slots, self.rank = slots, rank

    def step(self, seed: int, dist, world: int, coll_dev):
        from azsp.distributed import all_gather_variable, pack_records
        rng = np.random.RandomState(seed)
        n = self.slots * 3 + self.rank
        counts = rng.randint(0, 30, size=(n, 7)) + 1
        rec = torch.from_numpy(pack_records(rng.randint(-1, 2, size=(n, 42)).astype(np.int8), counts / counts.sum(axis=1, keepdims=True),
                                            rng.randint(-1, 2, size=n).astype(np.int32)))
        tg = time.perf_counter()
        gathered = all_gather_variable(rec, dist, world) if dist is not None else None
        return self.slots, n, gathered, time.perf_counter() - tg


def combine_ranks(dist, world: int, coll_dev, dt: float, episodes: int, samples: int, gathered_records: int,
                  gather_s: float) -> dict:
    """What the ranks exchange after the timed region (also exercised by the 2-rank gloo test): the step time is the
    MAX over ranks, counts are summed, every rank's own time and all-gather time travel along so that a scaling run
    explains itself, and the all-gather must have delivered every rank's samples to everyone."""
    if dist is None:
        return {"dt": dt, "episodes": episodes, "samples": samples, "per_rank_s": [dt], "allgather_s": [gather_s]}
    mine = torch.tensor([dt, gather_s], device=coll_dev, dtype=torch.float64)
    flat = torch.zeros(world * 2, device=coll_dev, dtype=torch.float64)     # (gloo wants the concatenated, flat form)
    dist.all_gather_into_tensor(flat, mine)
    allt = flat.view(world, 2)
    tot = torch.tensor([episodes, samples], device=coll_dev, dtype=torch.int64)
    dist.all_reduce(tot)
    episodes, samples = int(tot[0]), int(tot[1])
    if gathered_records != samples:
        raise SystemExit(f"rank {dist.get_rank()}: all-gather delivered {gathered_records} records, expected {samples}")
    allt = allt.cpu()
    return {"dt": float(allt[:, 0].max()), "episodes": episodes, "samples": samples,
            "per_rank_s": [float(x) for x in allt[:, 0]], "allgather_s": [float(x) for x in allt[:, 1]]}


def cpu_baseline(sims: int, slots: int, state_dict) -> dict:
    """The oracle's port of the reference's Python/torch-CPU path (oracle/az_oracle.py + oracle/resnet_ref.py),
    plus, as a second figure, the plain-C tree search of oracle/az_oracle.c driving the same torch-CPU network."""
    from oracle import az_oracle as oz
    from oracle import c_oracle as co
    from oracle.resnet_ref import ResNetRef
    cores = os.cpu_count() or 1
    threads = min(cores, 16)
    torch.set_num_threads(threads)
    net = ResNetRef(state_dict)
    np.random.seed(0)
    t0 = time.perf_counter()
    eps = list(oz.generate_episodes(oz.C4Rules(), net.predict, sims, slots))
    dt = time.perf_counter() - t0
    out = {"value": len(eps) / dt, "unit": "episodes/s", "cores": threads, "kind": "port",
           "sample": f"{slots} concurrent games x {sims} sims, first {len(eps)} completions, {dt:.1f} s, "
                     f"{net.boards_seen} net evals, host has {cores} logical cores"}
    try:
        net2 = ResNetRef(state_dict)
        cb = co.make_predict_cb(net2.predict, (6, 7), 7)
        t0 = time.perf_counter()
        ref = co.generate("connect4", sims, slots, seed=0, predict_cb=cb, trace=False)
        dt2 = time.perf_counter() - t0
        out["c_tree_torch_net"] = {"value": ref["n_episodes"] / dt2, "unit": "episodes/s", "seconds": dt2,
                                   "note": "oracle/az_oracle.c tree search + the same torch-CPU network"}
    except Exception as exc:                # noqa: BLE001  the second figure is optional
        out["c_tree_torch_net"] = {"error": repr(exc)}
    return out


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--slots", type=int, default=4096, help="concurrent games per GPU")
    ap.add_argument("--sims", type=int, default=100)
    ap.add_argument("--blocks", type=int, default=4)
    ap.add_argument("--channels", type=int, default=64)
    ap.add_argument("--cpu-slots", type=int, default=48, help="concurrent games of the CPU baseline sample")
    ap.add_argument("--groups", type=int, default=0, help="concurrent slot groups / HIP streams per GPU (0 = auto)")
    ap.add_argument("--precision", choices=["f32", "f16x2"], default="f32", help="inference arithmetic of the headline")
    ap.add_argument("--single", action="store_true", help="measure only --precision")
    ap.add_argument("--no-clock-probe", action="store_true", help="skip the in-kernel clock measurement after the timed region")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--timing-sample", type=int, default=4,
                    help="evaluation-cache runs: HIP events around the kernels of 32 consecutive simulation rounds out of every 32 * N "
                         "(1 = every launch, as the runs without the cache always do)")
    ap.add_argument("--no-eval-cache", action="store_true", help="skip the evaluation-cache measurements")
    ap.add_argument("--eval-cache-headline", action="store_true", help="diagnostics: make the evaluation-cache run the top-level line")
    ap.add_argument("--game", choices=["connect4", "gomoku9"], default="connect4",
                    help="gomoku9 = BASELINE.json configs[4]'s game (not a headline config; no CPU baseline, no evaluation cache)")
    args = ap.parse_args()
    if args.game == "gomoku9":
        args.no_cpu_baseline = True

    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # no launcher: become one (nothing in this process has initialised a GPU)
        raise SystemExit(launch_ranks(args.gpus, sys.argv[1:]))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} rank(s); start it as "
                         f"`python bench.py --gpus {args.gpus}` (bench.py launches its own ranks) or "
                         f"`python -m torch.distributed.run --nproc-per-node {args.gpus} bench.py --gpus {args.gpus}`")
    rehearse = os.environ.get("AZSP_BENCH_REHEARSE") == "1"
    if rehearse:
        args.single, args.no_cpu_baseline, args.no_clock_probe, args.no_eval_cache = True, True, True, True
    elif not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the self-play engine has no CPU path")
    # one process per GPU; AZSP_BENCH_BACKEND=gloo is a rehearsal mode for boxes with fewer GPUs than ranks
    backend = "gloo" if rehearse else os.environ.get("AZSP_BENCH_BACKEND", "nccl")
    n_dev = 0 if rehearse else torch.cuda.device_count()
    if backend == "nccl" and world > n_dev:
        raise SystemExit(f"bench.py: {world} ranks but only {n_dev} GPU(s) visible: one process per GPU over RCCL needs one device "
                         f"per rank (AZSP_BENCH_BACKEND=gloo rehearses the collectives with several ranks on fewer GPUs)")
    if not rehearse:
        local_rank %= n_dev                    # (only the gloo rehearsal mode can have more ranks than devices)
        torch.cuda.set_device(local_rank)
    coll_dev = torch.device("cuda", local_rank) if backend == "nccl" else torch.device("cpu")
    dist = None
    if world > 1 or os.environ.get("AZSP_BENCH_FORCE_DIST") == "1":   # the latter: 1-rank rehearsal of the RCCL calls
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")           # (only the single-process rehearsal runs without a launcher)
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=coll_dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    import azsp
    from azsp.distributed import all_gather_records

    torch.manual_seed(0)                       # same random-weight net on every rank
    game_cls, board, n_act = (azsp.Connect4Game, (6, 7), 7) if args.game == "connect4" else (azsp.GomokuGame, (9, 9), 81)
    net = azsp.ResNet(board, n_act, args.blocks, args.channels)
    net.eval()
    cfg = azsp.AlphaZeroConfig(batch_size=32, num_iterations=1, num_simulations=args.sims, num_episodes=args.slots,
                               num_iters_for_train_history=1, epochs=1)
    cells = board[0] * board[1]
    fpb = resnet_flops_per_board(args.blocks, args.channels, cells, n_act)

    def fence():
        if dist is not None:
            dist.barrier()
        if not rehearse:
            torch.cuda.synchronize()

    def measure(precision: str, eval_cache: bool = False) -> dict:
        if rehearse:
            reh = RehearsalGenerator(args.slots, rank)
            gen = None
        else:
            gen = azsp.EpisodeGenerator(game=game_cls(), config=cfg, device=local_rank, timing=True, n_groups=args.groups,
                                        f16_split=precision == "f16x2", eval_cache=eval_cache,
                                        # sampled event windows for the cache runs only (a round there is launch latency);
                                        # the runs without the cache time every launch
                                        timing_sample=args.timing_sample if eval_cache else 1)

        def step(seed: int):
            if rehearse:
                return reh.step(seed, dist, world, coll_dev)
            np.random.seed(seed)               # independent replica per rank (SURVEY 8e mode B)
            eps = list(gen.generate_episodes(net))
            tg = time.perf_counter()
            gathered = all_gather_records(gen, dist, world, coll_dev) if dist is not None else None
            if gathered is not None and coll_dev.type == "cuda":
                torch.cuda.synchronize()
            return len(eps), sum(len(e) for e in eps), gathered, time.perf_counter() - tg

        for w in range(args.warmup):
            step(1000 * (w + 1) + rank)
        eng = None if rehearse else gen.engine_for(net)
        if eng is not None:
            eng.reset_stats()
        fence()
        t0 = time.perf_counter()
        episodes = samples = gathered_records = 0
        gather_s = 0.0
        for k in range(args.steps):
            e, s, g, tg = step(10 * k + rank)
            episodes += e
            samples += s
            gather_s += tg
            if g is not None:
                gathered_records += sum(int(x.shape[0]) for x in g)
        fence()
        dt = time.perf_counter() - t0
        st = eng.stats() if eng is not None else None
        # gathered_records counts every rank's records as seen here; samples is this rank's own until combined
        comb = combine_ranks(dist, world, coll_dev, dt, episodes, samples, gathered_records, gather_s)
        dt, episodes, samples = comb["dt"], comb["episodes"], comb["samples"]
        ranks = {"ms_per_step": [1e3 * x / max(args.steps, 1) for x in comb["per_rank_s"]],
                 "allgather_ms_per_step": [1e3 * x / max(args.steps, 1) for x in comb["allgather_s"]]}
        if rehearse:
            return {"value": 0.0, "unit": "episodes/s", "ms_per_step": 1e3 * dt / max(args.steps, 1), "dtype": "none (rehearsal)",
                    "episodes": episodes, "samples": samples, "ranks": ranks, "roofline": None, "tree": None}
        eng.close()
        launches = max(int(st["infer_launches"]), 1)
        # Evaluation-cache runs: the events sit on the launches of 32 consecutive simulation rounds out of every 32 *
        # --timing-sample (timestamps on a dispatch cost a few microseconds each -- 5-9 % of the step where a round is launch
        # latency); durations and overlap are measured on those launches and the busy time scaled to all of them (the union
        # of a window's intervals misses the untimed neighbours at its two ends: with two groups ~1 % too much busy time).
        timed = max(int(st["infer_launches_timed"]), 1)
        avg_ms = st["infer_ms"] / timed
        # Launches of different slot groups run concurrently on separate streams, so the rate is taken over the
        # union of the launch intervals: flops per launch / (average launch duration / overlap factor).
        union_timed = st["infer_union_ms"] if st["infer_union_ms"] > 0 else st["infer_ms"]
        overlap = st["infer_ms"] / union_timed if union_timed > 0 else 1.0
        union_ms = union_timed * launches / timed
        peak = PEAK_TFLOPS[precision]
        boards = st["infer_boards"]                  # positions the kernel evaluated (== evals unless the evaluation cache is on)
        # EXECUTED matrix-pipe FLOP, summed on the device: every inference launch adds (workgroups that had boards) x (what
        # one workgroup of that kernel variant executes: the padding rows of its tile included, statically skipped
        # off-board taps excluded, x 3 for f16x2's hi*hi + hi*lo + lo*hi)
        exec_flops = st["infer_exec_flops"]
        efpb = exec_flops / max(boards, 1)
        tf = lambda flop, ms: flop / (ms * 1e-3) / 1e12 if ms > 0 else 0.0      # noqa: E731
        achieved = tf(exec_flops, union_ms)
        algorithmic = tf(fpb * boards, union_ms)
        return {
            "value": episodes / dt, "unit": "episodes/s", "ms_per_step": 1e3 * dt / max(args.steps, 1), "dtype": DTYPE[precision],
            "episodes": episodes, "samples": samples, "ranks": ranks,
            "evals": int(st["evals"]), "evals_executed": int(st["evals_executed"]),
            **({"eval_cache": {"entries": int(st["eval_cache_entries"]),
                               "hit_rate": 1.0 - st["evals_executed"] / max(st["evals"], 1),
                               "reference_evals_per_executed": st["evals"] / max(st["evals_executed"], 1)}} if eval_cache else {}),
            # frac = EXECUTED matrix FLOP / time with >= 1 inference launch running / peak: <= 1 by construction.
            # algorithmic_* = the same with SURVEY 8d's dense FLOP count per evaluated board (the kernel skips
            # exact-zero taps, so this one can exceed 1); frac_over_wall = executed FLOP / whole step time / peak.
            "roofline": {"bound": "mfma", "kernel": KERNEL[precision], "achieved": achieved, "peak": peak, "unit": "TFLOP/s",
                         "frac": achieved / peak, "traffic": pmc_traffic(KERNEL[precision]),
                         "executed_flops_per_board": efpb, "flops_per_board": fpb,
                         "algorithmic_achieved": algorithmic, "algorithmic_frac": algorithmic / peak,
                         "frac_over_wall": tf(exec_flops, dt * 1e3) / peak,
                         "boards_per_launch": boards / launches, "avg_launch_ms": avg_ms, "launches": launches, "launches_timed": timed,
                         "concurrent_launches": overlap, "busy_ms": union_ms,
                         # the same without the overlap factor: one launch's executed flops / its own duration (the
                         # launches of the slot groups share the machine: a per-launch share, not the machine's rate)
                         "per_launch_frac": tf(exec_flops / launches, avg_ms) / peak if avg_ms > 0 else None,
                         "infer_share_of_wall": union_ms * 1e-3 / dt if dt > 0 else None},
            "tree": tree_report(st, dt, n_act),
        }

    def probe_clock(result, precision):
        """The peaks assume the nominal 2.4 GHz; what the chip holds under the kernel is measured right after the timed
        region with the stamped diagnostic instantiation: 2000 untraced launches (about 1.5 s) of 3072 mid-game boards
        (two workgroups per CU, like the overlapping launches of the run), then one traced launch."""
        rl = result["roofline"]
        try:
            from azsp.engine import Engine
            from azsp.weights import resnet_blob
            rng = np.random.RandomState(0)
            pb = np.zeros((3072, 6, 7), np.int8)
            for b in pb:
                for c in range(7):
                    h = rng.randint(0, 6)
                    b[6 - h:, c] = rng.choice([-1, 1], size=h)
            with Engine("connect4", 3072, 1, device=local_rank, res_blocks=args.blocks, channels=args.channels,
                        f16_split=(precision == "f16x2")) as pe:
                pe.load_resnet(resnet_blob(net)[0])
                for _ in range(2000):
                    pe.predict(pb)
                ghz, span = pe.probe_clock(pb)
            rl["held_clock_ghz"] = ghz
            rl["peak_at_held_clock"] = rl["peak"] * ghz / 2.4
            rl["frac_at_held_clock"] = rl["achieved"] / rl["peak_at_held_clock"]      # executed FLOP against the clock the chip held
        except Exception as exc:        # noqa: BLE001  the probe is an extra, never a reason to lose the bench line
            rl["held_clock_ghz"] = None
            rl["held_clock_error"] = str(exc)

    probe_ok = args.game == "connect4" and 32 < args.channels <= 64 and rank == 0 and not args.no_clock_probe
    cache_ok = args.game == "connect4" and not args.no_eval_cache
    head = measure(args.precision, eval_cache=args.eval_cache_headline and cache_ok)
    if probe_ok:
        probe_clock(head, args.precision)
    other = None
    if not args.single:
        other_precision = "f16x2" if args.precision == "f32" else "f32"
        other = measure(other_precision)
        if probe_ok:
            probe_clock(other, other_precision)
    # the evaluation cache (identical episodes, fewer executed evaluations): reported beside the headline, never as it
    cached = {}
    if cache_ok and not args.eval_cache_headline:
        for prec in ([args.precision] if args.single else ["f32", "f16x2"]):
            cached[prec] = measure(prec, eval_cache=True)

    if rank == 0:
        keys = ("value", "unit", "ms_per_step", "dtype", "evals", "evals_executed", "roofline", "tree")
        out = {
            "metric": "self-play episodes/sec (Connect-4, 100 sims/move)" if (args.sims == 100 and args.game == "connect4") else
                      f"self-play episodes/sec ({args.game}, {args.sims} sims/move)",
            "value": head["value"], "unit": "episodes/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": head["ms_per_step"], "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": head["dtype"],
            "data": "REHEARSAL: fabricated records, no engine, no GPU (launcher / collective plumbing only)" if rehearse else
                    "synthetic (empty-board starts, random-weight net, seeded RNG)",
            "config": {"workload": f"{args.game} self-play, ResNet {args.channels}ch/{args.blocks}blk random weights, "
                                   f"{args.sims} sims/move, {args.slots} concurrent games per GPU, first "
                                   f"{args.slots} completions per GPU per step"
                                   + (", evaluation cache ON" if args.eval_cache_headline and cache_ok else ""),
                       "slots_per_gpu": args.slots, "sims": args.sims, "parallelism": f"game-sharded x{world}",
                       "episodes": head["episodes"], "samples": head["samples"],
                       "mean_plies_per_episode": head["samples"] / max(head["episodes"], 1),
                       "launched_by": "bench.py (own torch.distributed.run child)" if os.environ.get("AZSP_BENCH_SELF_LAUNCHED") == "1"
                                      else ("external launcher" if "WORLD_SIZE" in os.environ else "single process")},
            "ranks": head["ranks"],
            "roofline": head["roofline"], "tree": head["tree"],
        }
        if rehearse:
            out["rehearsal"] = True
        for k in ("evals", "evals_executed", "eval_cache"):
            if k in head:
                out[k] = head[k]
        if other is not None:
            key = "f16x2" if args.precision == "f32" else "f32"
            out[key] = {k: other[k] for k in keys}
        if cached:
            out["eval_cache"] = {"note": "AZSP_FLAG_EVAL_CACHE: every position evaluated at most once per generate_episodes call; "
                                         "episodes identical to the headline's (tests), roofline on EXECUTED evaluations only",
                                 **{prec: {k: r[k] for k in keys + ("eval_cache",)} for prec, r in cached.items()}}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.sims, args.cpu_slots, net.state_dict())
        if "cpu_baseline" in out:
            out["cpu_baseline"]["gpu_over_cpu"] = out["value"] / out["cpu_baseline"]["value"]
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
