#!/usr/bin/env python3
"""Self-play throughput benchmark (BASELINE.json: "self-play episodes/sec (Connect-4, 100 sims/move)").

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A *step* is one complete `EpisodeGenerator.generate_episodes` call of the workload: `--slots`
concurrent Connect-4 games per GPU (default 4096 = BASELINE.json configs[1]), `--sims` MCTS
simulations per move, random-weight ResNet(`--blocks`, `--channels`), run until `slots` episodes have
completed (reference stop rule), examples copied to the host, and -- for N > 1 -- the packed
(s, pi, z) records all-gathered over RCCL.  Games shard over GPUs with no other traffic (weak scaling).

The headline (`value`, `roofline`) is measured with the inference kernel in exact fp32
(v_mfma_f32_32x32x2_f32, `dtype` "f32").  Unless `--single`, the same workload is then measured once more
with the split-fp16 inference kernel (fp32 operands carried as two fp16 numbers, fp32 accumulate; same
accuracy bound in the tests) and reported under `"f16x2"`; `--precision f16x2` makes that the headline.

The JSON line carries the MFMA roofline of the dominant kernel (timed with HIP events on the engine's
own streams inside the timed region) and a CPU baseline: the oracle's pure-Python/torch-CPU port of the
reference path timed on this box's host cores (rank 0, N=1 only)."""
from __future__ import annotations

import argparse
import json
import os
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
    ap.add_argument("--game", choices=["connect4", "gomoku9"], default="connect4",
                    help="gomoku9 = BASELINE.json configs[4]'s game (not a headline config; fp32 only, no CPU baseline)")
    args = ap.parse_args()
    if args.game == "gomoku9":
        args.single, args.no_cpu_baseline, args.precision = True, True, "f32"

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the self-play engine has no CPU path")
    # one process per GPU; AZSP_BENCH_BACKEND=gloo is a rehearsal mode for boxes with fewer GPUs than ranks
    backend = os.environ.get("AZSP_BENCH_BACKEND", "nccl")
    n_dev = torch.cuda.device_count()
    if backend == "nccl" and world > n_dev:
        raise SystemExit(f"bench.py: {world} ranks but only {n_dev} GPU(s) visible: one process per GPU over RCCL needs one device "
                         f"per rank (AZSP_BENCH_BACKEND=gloo rehearses the collectives with several ranks on fewer GPUs)")
    local_rank %= n_dev                        # (only the gloo rehearsal mode can have more ranks than devices)
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
        torch.cuda.synchronize()

    def measure(precision: str) -> dict:
        gen = azsp.EpisodeGenerator(game=game_cls(), config=cfg, device=local_rank, timing=True,
                                    n_groups=args.groups, f16_split=precision == "f16x2")

        def step(seed: int):
            np.random.seed(seed)               # independent replica per rank (SURVEY 8e mode B)
            eps = list(gen.generate_episodes(net))
            tg = time.perf_counter()
            gathered = all_gather_records(gen, dist, world, coll_dev) if dist is not None else None
            if gathered is not None and coll_dev.type == "cuda":
                torch.cuda.synchronize()
            return len(eps), sum(len(e) for e in eps), gathered, time.perf_counter() - tg

        for w in range(args.warmup):
            step(1000 * (w + 1) + rank)
        eng = gen.engine_for(net)
        eng.reset_stats()
        fence()
        t0 = time.perf_counter()
        episodes = samples = gathered_records = 0
        gather_s = 0.0
        t_own = 0.0
        for k in range(args.steps):
            ts = time.perf_counter()
            e, s, g, tg = step(10 * k + rank)
            t_own += time.perf_counter() - ts
            episodes += e
            samples += s
            gather_s += tg
            if g is not None:
                gathered_records += sum(int(x.shape[0]) for x in g)
        fence()
        dt = time.perf_counter() - t0
        st = eng.stats()
        # gathered_records counts every rank's records as seen here; samples is this rank's own until combined
        comb = combine_ranks(dist, world, coll_dev, dt, episodes, samples, gathered_records, gather_s)
        dt, episodes, samples = comb["dt"], comb["episodes"], comb["samples"]
        eng.close()
        launches = max(int(st["infer_launches"]), 1)
        avg_ms = st["infer_ms"] / launches
        # Launches of different slot groups run concurrently on separate streams, so the rate is taken over the
        # union of the launch intervals: flops per launch / (average launch duration / overlap factor).
        union_ms = st["infer_union_ms"] if st["infer_union_ms"] > 0 else st["infer_ms"]
        overlap = st["infer_ms"] / union_ms if union_ms > 0 else 1.0
        achieved = fpb * st["infer_boards"] / (union_ms * 1e-3) / 1e12 if union_ms > 0 else 0.0
        peak = PEAK_TFLOPS[precision]
        return {
            "value": episodes / dt, "unit": "episodes/s", "ms_per_step": 1e3 * dt / max(args.steps, 1), "dtype": DTYPE[precision],
            "episodes": episodes, "samples": samples,
            "ranks": {"ms_per_step": [1e3 * x / max(args.steps, 1) for x in comb["per_rank_s"]],
                      "allgather_ms_per_step": [1e3 * x / max(args.steps, 1) for x in comb["allgather_s"]]},
            "roofline": {"bound": "mfma", "kernel": KERNEL[precision], "achieved": achieved, "peak": peak, "unit": "TFLOP/s",
                         "frac": achieved / peak, "traffic": pmc_traffic(KERNEL[precision]), "flops_per_board": fpb,
                         "boards_per_launch": st["infer_boards"] / launches, "avg_launch_ms": avg_ms, "launches": launches,
                         "concurrent_launches": overlap, "busy_ms": union_ms,
                         # the same without the overlap factor: one launch's flops / its own duration (the launches of the
                         # three slot groups share the machine, so this is a per-launch share, not the machine's rate)
                         "per_launch_frac": fpb * st["infer_boards"] / launches / (avg_ms * 1e-3) / 1e12 / peak if avg_ms > 0 else None,
                         "infer_share_of_wall": union_ms * 1e-3 / dt if dt > 0 else None,
                         # Connect-4 kernels skip the 3x3 taps that a whole 32-row tile lacks (exact zeros): 3 of 18
                         # tile-taps per residual conv.  `achieved` stays algorithmic; this is the executed share.
                         "executed_flops_per_board": fpb - (2.0 * cells * 2 * args.blocks * args.channels * args.channels * 9 / 6.0
                                                            if args.game == "connect4" else 0.0),
                         },
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
            rl["frac_at_held_clock"] = rl["achieved"] / rl["peak_at_held_clock"]
            # f16x2 executes three fp16 MFMAs per algorithmic multiply-add (hi*hi, hi*lo, lo*hi)
            rl["executed_frac_at_held_clock"] = (rl["frac_at_held_clock"] * rl["executed_flops_per_board"] / rl["flops_per_board"]
                                                 * (3.0 if precision == "f16x2" else 1.0))
        except Exception as exc:        # noqa: BLE001  the probe is an extra, never a reason to lose the bench line
            rl["held_clock_ghz"] = None
            rl["held_clock_error"] = str(exc)

    probe_ok = args.game == "connect4" and 32 < args.channels <= 64 and rank == 0 and not args.no_clock_probe
    head = measure(args.precision)
    if probe_ok:
        probe_clock(head, args.precision)
    other = None
    if not args.single:
        other_precision = "f16x2" if args.precision == "f32" else "f32"
        other = measure(other_precision)
        if probe_ok:
            probe_clock(other, other_precision)

    if rank == 0:
        out = {
            "metric": "self-play episodes/sec (Connect-4, 100 sims/move)" if (args.sims == 100 and args.game == "connect4") else
                      f"self-play episodes/sec ({args.game}, {args.sims} sims/move)",
            "value": head["value"], "unit": "episodes/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": head["ms_per_step"], "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": head["dtype"], "data": "synthetic (empty-board starts, random-weight net, seeded RNG)",
            "config": {"workload": f"{args.game} self-play, ResNet {args.channels}ch/{args.blocks}blk random weights, "
                                   f"{args.sims} sims/move, {args.slots} concurrent games per GPU, first "
                                   f"{args.slots} completions per GPU per step",
                       "slots_per_gpu": args.slots, "sims": args.sims, "parallelism": f"game-sharded x{world}",
                       "episodes": head["episodes"], "samples": head["samples"],
                       "mean_plies_per_episode": head["samples"] / max(head["episodes"], 1)},
            "ranks": head["ranks"],
            "roofline": head["roofline"], "tree": head["tree"],
        }
        if other is not None:
            key = "f16x2" if args.precision == "f32" else "f32"
            out[key] = {k: other[k] for k in ("value", "unit", "ms_per_step", "dtype", "roofline", "tree")}
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
