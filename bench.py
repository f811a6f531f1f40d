#!/usr/bin/env python3
"""
bench.py -- headline benchmark of the self-play rollout path on MI355X (contract: see the task statement).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

Metric (BASELINE.json): MCTS nodes/sec (+ self-play plies / games per hour) at 800 simulations per move
with the 10-block x 128-filter net.  node = one simulation = one iteration of the reference's loop
mcts.py:210, so nodes = games x plies x NUM_SIMULATIONS.
Workload at N=1: 256 concurrent games x 800 sims/move, MCTS_BATCH_SIZE 96, net 8 plain + 2 SE blocks x 128
(= the per-GPU shard of BASELINE.json configs[2]; configs[1] is the same with --sims 400).  One timed
"step" = one ply of every game (256 searches of 800 simulations + the moves).  Weak scaling: every rank
runs its own 256 games (game id -> rank by id mod world); the only collective is the record all-gather.

Also measured live, per the contract:
  roofline      PUCT-select kernel on the wide synthetic workload of SURVEY.md section 8d (HIP events on the
                kernel's stream; algorithmic bytes = measured levels x 392 B)
  roofline_step the in-loop tree-step kernel (latency-bound, cache-resident; reported for honesty)
  cpu_baseline  the CPU oracle (oracle/, a C port of the reference) + the same net under torch-CPU,
                timed on this box's host cores on a bounded sample
"""
from __future__ import annotations

import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np
import torch

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E peak (MI355X_MICROARCH.md)
NETS = {"4x64": (3, 1, 64), "10x128": (8, 2, 128), "20x256": (15, 5, 256)}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--games", type=int, default=256, help="concurrent games per GPU")
    ap.add_argument("--sims", type=int, default=800)
    ap.add_argument("--batch", type=int, default=96)
    ap.add_argument("--net", default="10x128", choices=list(NETS))
    ap.add_argument("--net-dtype", default="fp32", choices=["fp32", "fp16", "bf16"])
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--wide-trees", type=int, default=0, help="0 = 262144 if >= 150 GB of HBM is free, else 131072")
    ap.add_argument("--wide-nodes", type=int, default=800)
    ap.add_argument("--fast", action="store_true", help="FAST search mode (virtual loss; not the reference's semantics; not the headline)")
    ap.add_argument("--leaves", type=int, default=16, help="FAST mode: leaves per game per step")
    ap.add_argument("--max-game-moves", type=int, default=16384, help="config.MAX_GAME_MOVES (small values make games finish: record path)")
    ap.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"], help="gloo: rehearsal of the N>1 path on one GPU")
    ap.add_argument("--force-dist", action="store_true", help="initialise torch.distributed even with WORLD_SIZE=1")
    ap.add_argument("--share-gpu", action="store_true", help="rehearsal: every rank uses cuda:0")
    return ap.parse_args()


def make_net(name, device, dtype, batch=256):
    from betaone_amd import dropin

    dropin.install()
    import config
    import network

    config.RESIDUAL_BLOCKS, config.SE_RESIDUAL_BLOCKS, config.CONV_FILTERS = NETS[name]
    torch.manual_seed(0)
    net = network.PolicyValueNet().eval()
    td = {"fp32": torch.float32, "fp16": torch.float16, "bf16": torch.bfloat16}[dtype]
    if str(device) == "cpu":
        return net, net.for_inference(dtype=td, channels_last=False)
    from betaone_amd.nn_tune import best_inference_copy

    return net, best_inference_copy(net, batch, device, td)


class CastIn(torch.nn.Module):
    """fp16/bf16 evaluate stage: the engine writes float32 planes; cast once in front of the net."""

    def __init__(self, net, dtype):
        super().__init__()
        self.net, self.dtype = net, dtype

    def forward(self, x):
        if getattr(self.net, "wants_float32_input", False):  # the fp16 tower converts the planes itself
            return self.net(x)
        return self.net(x.to(self.dtype))


def select_roofline(args, device):
    """HBM roofline of the PUCT-select kernel on the SURVEY section 8d wide workload."""
    from betaone_amd import select_wide as SW

    n_trees = args.wide_trees
    if n_trees <= 0:
        free, _total = torch.cuda.mem_get_info(device)
        n_trees = 262144 if free > 150 * (1 << 30) else (131072 if free > 75 * (1 << 30) else 32768)
    args.wide_trees = n_trees
    w = SW.build(n_trees, args.wide_nodes, seed=0, device=device)
    flush = torch.empty(512 << 20, dtype=torch.uint8, device=device)  # evict the 256 MiB Infinity Cache between launches
    out = SW.run(w)
    torch.cuda.synchronize(device)
    levels = int(out[1].sum().item())
    alg_bytes = levels * SW.LEVEL_BYTES
    reps, ms = 20, []
    for _ in range(reps):
        flush.fill_(1)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        SW.run(w, out=out)
        e1.record()
        e1.synchronize()
        ms.append(e0.elapsed_time(e1))
    t = float(np.mean(ms)) * 1e-3
    ach = alg_bytes / t / 1e9
    # HBM bytes per launch from the committed PMC pass (rocprofv3 --pmc FETCH_SIZE, x2 gfx950 correction) when it
    # was taken on this very workload; counters cannot be read from inside the process.
    traffic, traffic_src = None, None
    try:
        pm = json.load(open(os.path.join(ROOT, "profiles", "r01_select_wide_pmc.json")))
        if pm["n_trees"] == n_trees and pm["nodes"] == args.wide_nodes and pm["levels_per_launch"] == levels:
            traffic, traffic_src = pm["hbm_read_bytes_per_launch_corrected"], "profiles/r01_select_wide_pmc.md"
    except Exception:
        pass
    return {"bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4),
            "traffic": traffic, "traffic_source": traffic_src, "kernel": "bo_k_select_wide",
            "workload": f"{args.wide_trees} trees x {args.wide_nodes} nodes x 32 children (512 B child blocks, "
                        f"{w['blocks'].numel() * 4 / 1e9:.1f} GB resident in HBM, 512 MiB written between launches to evict the Infinity Cache), "
                        f"one PUCT descent per tree per launch; bytes = levels x (12 B x 32 children + 8 B), the kernel moves 512 B per level",
            "levels_per_launch": levels, "alg_bytes_per_launch": alg_bytes, "avg_launch_ms": round(t * 1e3, 4)}


def nn_roofline(net, batch, device):
    """MFMA roofline of the evaluate stage's tower kernel (csrc/bo_tower_wg.h / bo_tower.h), timed with events on the
    stream it is launched on.  `achieved` counts the fp32 MFMA flops the kernel executes (Winograd F(2x2,3x3): 16
    multiplies per 2x2 output tile and input channel, input conv padded to 128 channels); `algorithmic` is the direct
    3x3 convolution's flop count for the same layers (what MIOpen / the reference's net would be charged)."""
    conv = getattr(net, "conv", None)
    if conv not in ("tower", "tower_wg"):
        return None
    C, n_conv = net.c, 1 + 2 * len(net.blocks)
    x = torch.rand((batch, 120, 8, 8), device=device)
    with torch.no_grad():
        for _ in range(5):
            net._tower_forward(x, heads=conv == "tower_wg")
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 50
        e0.record()
        for _ in range(reps):
            net._tower_forward(x, heads=conv == "tower_wg")
        e1.record()
        e1.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    per_mac = 16 * 16 if conv == "tower_wg" else 9 * 64       # multiplies per (c_in, c_out) pair and board
    executed = 2.0 * per_mac * C * (128 + (n_conv - 1) * C) * batch
    algorithmic = 2.0 * 9 * 64 * C * (120 + (n_conv - 1) * C) * batch
    peak = 157.3  # TFLOP/s dense fp32 MFMA: 256 CUs x 4 SIMDs x 64 flop/clk x 2.4 GHz (MI355X_MICROARCH.md)
    return {"bound": "mfma", "kernel": "bo_k_tower_wg" if conv == "tower_wg" else "bo_k_tower", "achieved": round(executed / us / 1e6, 1), "peak": peak,
            "unit": "TFLOP/s", "frac": round(executed / us / 1e6 / peak, 4), "traffic": None, "avg_launch_us": round(us, 1),
            "boards_per_launch": batch, "conv_layers": n_conv,
            "algorithmic_direct_conv_tflops": round(algorithmic / us / 1e6, 1),
            "note": "fp32 v_mfma_f32_16x16x4_f32; one workgroup per board, activations LDS-resident for the whole tower"}


def step_roofline(ro, n_steps_timed):
    """The in-loop tree step kernel: algorithmic bytes from the engine's own counters."""
    st = ro.eng.status()
    return {"levels": int(st["levels"].sum()), "children_scanned": int(st["children_scanned"].sum()),
            "evals": int(st["evals"].sum()), "flushes": int(st["flushes"].sum())}


def cpu_baseline(args):
    """The reference's algorithm on the host CPU: oracle/ (C port of mcts.py/self_play.py + python-chess
    rules) with the same net under torch-CPU fp32."""
    from oracle import oracle as O

    net, _ = make_net(args.net, "cpu", "fp32")
    threads = max(1, min(os.cpu_count() or 1, 16))
    torch.set_num_threads(threads)

    def eval_fn(planes):
        with torch.no_grad():
            logits, value = net(torch.from_numpy(np.ascontiguousarray(planes)))
            return torch.softmax(logits, dim=1).numpy(), value.reshape(-1).numpy()

    cfg = O.default_config(num_simulations=args.sims, batch_size=args.batch)
    t0, sims, plies, games, evals = time.perf_counter(), 0, 0, 0, 0
    while time.perf_counter() - t0 < args.cpu_seconds:
        g = O.self_play(eval_fn, np.random.RandomState(games), cfg, max_plies=6)
        sims += g["n_sims"]; plies += len(g["moves"]); evals += g["n_evals"]; games += 1
    dt = time.perf_counter() - t0
    cpu_model = "unknown CPU"
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                cpu_model = ln.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return {"value": round(sims / dt, 1), "unit": "nodes/s", "cores": threads, "kind": "port", "cpu": cpu_model,
            "sample": f"{games} game prefixes x 6 plies ({plies} searches of {args.sims} sims, {evals} unique NN evals) "
                      f"in {dt:.1f} s; oracle/ C port single-threaded, net {args.net} fp32 under torch-CPU with {threads} threads; "
                      f"the port evaluates each unique leaf once (the Python reference evaluates up to 96 duplicate rows per batch)"}


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if args.share_gpu:
        local = 0
    if world > 1 or args.force_dist:
        import torch.distributed as dist

        torch.cuda.set_device(local)
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device(f"cuda:{local}"))
        else:
            dist.init_process_group("gloo")
    device = torch.device(f"cuda:{local}")
    torch.cuda.set_device(device)

    from betaone_amd import engine as E
    from betaone_amd import records
    from betaone_amd.rollout import Rollout

    E.load_hip_library()
    _, net = make_net(args.net, device, args.net_dtype, args.games * (args.leaves if args.fast else 1))
    net_layout = getattr(net, "layout", "nchw")
    if args.net_dtype != "fp32":
        net = CastIn(net, {"fp16": torch.float16, "bf16": torch.bfloat16}[args.net_dtype])
    G = args.games
    ro = Rollout(net, G, num_simulations=args.sims, mcts_batch_size=args.batch, device=str(device), use_graph=not args.no_graph,
                 rng_mode="native", max_game_moves=args.max_game_moves, fast=args.fast, leaves_per_step=args.leaves)
    ids = [rank + world * s for s in range(G)]  # game id -> rank = id mod world; RandomState(seed = game id) streams
    ro.start_games(list(range(G)), ids, ids)
    next_id = [rank + world * G]
    finished_batch, n_finished, finished_timed, plies_finished, timing = [], [0], [0], [0], [False]

    def on_finished(fin):
        finished_batch.append(fin)
        n_finished[0] += 1
        if timing[0]:
            finished_timed[0] += 1
            plies_finished[0] += len(fin.moves)

    def refill(_slot):
        i = next_id[0]
        next_id[0] += world
        return i, i, None

    exchange = records.LaggedGameExchange(device) if dist is not None else None

    def one_step():
        ro.play_ply(on_finished=on_finished, refill=refill)
        if exchange is not None:  # the path's only exchange step: finished games' records to every rank (RCCL over xGMI);
            exchange.push(finished_batch)  # the size all-gather completes while the next ply runs
        finished_batch.clear()

    for _ in range(args.warmup):
        one_step()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize(device)
    s0, p0, f0, h0 = ro.n_sims, ro.n_plies, ro.n_forward, ro.host_seconds
    timing[0] = True
    t0 = time.perf_counter()
    for _ in range(args.steps):
        one_step()
    if exchange is not None:
        exchange.flush()  # the last step's records are delivered inside the timed region
    torch.cuda.synchronize(device)
    if dist is not None:
        dist.barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        rdev = device if args.dist_backend == "nccl" else torch.device("cpu")
        t = torch.tensor([dt], dtype=torch.float64, device=rdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        tot = torch.tensor([ro.n_sims - s0, ro.n_plies - p0, ro.n_forward - f0], dtype=torch.float64, device=rdev)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        sims, plies, fwd = [float(x) for x in tot.tolist()]
    else:
        sims, plies, fwd = float(ro.n_sims - s0), float(ro.n_plies - p0), float(ro.n_forward - f0)
    ro.eng.check_status()

    out = None
    if rank == 0:
        out = {
            "metric": "mcts_nodes_per_sec", "value": round(sims / dt, 1), "unit": "nodes/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": {"fp32": "f32", "fp16": "f16", "bf16": "bf16"}[args.net_dtype], "data": "synthetic",
            "config": {"workload": f"{G} concurrent self-play games per GPU x {args.sims} sims/move, MCTS_BATCH_SIZE {args.batch}, "
                                   f"net {args.net} ({'+'.join(map(str, NETS[args.net][:2]))} blocks x {NETS[args.net][2]} filters, "
                                   f"random init, {args.net_dtype}, BN folded), start position, per-game seeds = game id; "
                                   + ("BASELINE.json configs[4] per-GPU shard (f16 net, f32 tree)" if (args.net, args.net_dtype, G) == ("20x256", "fp16", 512)
                                      else "BASELINE.json configs[1]" if (args.net, args.net_dtype, G, args.sims) == ("10x128", "fp32", 256, 400)
                                      else "BASELINE.json configs[2] per-GPU shard" if (args.net, args.net_dtype, G, args.sims) == ("10x128", "fp32", 256, 800)
                                      else "custom configuration"),
                       "games_per_gpu": G, "sims_per_move": args.sims, "net": args.net, "net_dtype": args.net_dtype,
                       "hipgraph": not args.no_graph, "net_layout": net_layout,
                       "search_mode": ("fast: virtual loss, %d leaves/step, full-width expansion (NOT the reference's semantics)" % args.leaves)
                                      if args.fast else "reference semantics (bit-exact)", "parallelism": f"games sharded over {world} GPU(s), record all-gather only"},
            "plies_per_sec": round(plies / dt, 2), "nn_forwards_per_sec": round(fwd / dt / world, 2),
            "unique_nn_evals_per_sec": round(fwd * G * (args.leaves if args.fast else 1) / dt, 1), "games_finished": n_finished[0],
            "games_per_hour_at_100_plies": round(plies / dt * 3600 / 100.0, 1),
            "games_per_hour_measured": (round(finished_timed[0] * world * 3600.0 / dt, 1) if finished_timed[0] else None),
            "mean_plies_of_finished_games": (round(plies_finished[0] / finished_timed[0], 1) if finished_timed[0] else None),
            "host_fraction": round((ro.host_seconds - h0) / dt, 4),
        }
    if rank == 0 and not args.no_roofline:
        sr = step_roofline(ro, args.steps)
        out["roofline_step"] = {"bound": "latency", "note": "parity-mode trees are cache-resident (a few KB per game)",
                                "select_alg_bytes_total": sr["levels"] * 8 + sr["children_scanned"] * 12, **sr}
    ro.close()
    del ro
    torch.cuda.empty_cache()
    if rank == 0 and not args.no_roofline:
        out["roofline"] = select_roofline(args, device)
        if not args.fast and args.net_dtype == "fp32":
            rn = nn_roofline(net, G, device)
            if rn:
                out["roofline_nn"] = rn
    if rank == 0 and world == 1 and not args.no_cpu_baseline:  # "on rank 0 at N=1 only"
        out["cpu_baseline"] = cpu_baseline(args)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out))


if __name__ == "__main__":
    main()
