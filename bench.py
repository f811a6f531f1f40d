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

The timed region is STEADY-STATE self-play: before it, `--preroll` plies are played (untimed, staggered slot starts, same
engine, same settings) so that the resident games are at every stage of a game, finished games leave and new ones start
in every step, and games/hour is measured, not extrapolated.  The opening-only figure (all games at the start position) is
reported as the labelled extra `opening_phase`.

`--gpus N` with N > 1 and no WORLD_SIZE in the environment starts the N ranks itself (a torch.distributed.run child
process, started before this process touches a GPU) and relays rank 0's JSON line.

Also measured live, per the contract:
  roofline      the dominant kernel of the timed workload: the evaluate stage's residual-tower kernel (MFMA-bound, HIP events
                on its stream); with --fast the select + backup kernel on the trees the run grew (HBM, engine byte counters)
  roofline_select_wide_synthetic
                PUCT-select on the wide synthetic workload of SURVEY.md section 8d (algorithmic bytes = measured levels x 392 B)
  roofline_step the in-loop tree-step kernel (latency-bound, cache-resident; reported for honesty)
  cpu_baseline  the CPU oracle (oracle/, a C port of the reference) + the same net under torch-CPU: one game per core on
                the box's host cores (the reference's mp.Pool, main.py:168), bounded sample; per-core and x cores
"""
from __future__ import annotations

import argparse
import ctypes
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import betaone_amd  # noqa: F401  (first: sets GPU_MAX_HW_QUEUES before the HIP runtime starts -- betaone_amd/__init__.py)
import numpy as np
import torch

C_INT32 = ctypes.c_int32
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E peak (MI355X_MICROARCH.md)
NETS = {"4x64": (3, 1, 64), "10x128": (8, 2, 128), "20x256": (15, 5, 256)}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--games", type=int, default=None, help="concurrent games per GPU (default 256; --fast: 32768)")
    ap.add_argument("--sims", type=int, default=None, help="simulations per move (default 800; --fast: 128)")
    ap.add_argument("--batch", type=int, default=96)
    ap.add_argument("--net", default=None, choices=list(NETS), help="default 10x128")
    ap.add_argument("--net-dtype", default=None, choices=["fp32", "fp16", "bf16"], help="default fp32; --fast: fp16")
    ap.add_argument("--f32-tower", default=None, choices=["split", "fp32"],
                    help="float32 nets: 'split' (default, or BETAONE_F32_TOWER) = the tower on the fp16 matrix pipe with (hi, lo) operand pairs "
                         "(csrc/bo_tower_s.h); 'fp32' = the fp32-MFMA Winograd tower of rounds 1-2 (csrc/bo_tower_wg.h)")
    ap.add_argument("--cohorts", type=int, default=None,
                    help="the resident games as K phase-shifted cohorts, each with its own engine, HIP stream and captured graphs "
                         "(betaone_amd.rollout.CohortRollout); results per game are identical for every K; default: 4 from 256 games up, 2 from 128, else 1")
    ap.add_argument("--cu-masks", default=None, choices=["auto", "off", "contiguous", "interleaved", "full"],
                    help="cohort streams confined to disjoint 1/K shares of the compute units (hipExtStreamCreateWithCUMask; default: "
                         "BETAONE_COHORT_CU_MASK or off; with 4 cohorts 'contiguous' is what keeps their towers off each other's CUs)")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--cpu-cores", type=int, default=0, help="0 = the box's CPU share (affinity mask, at most 16)")
    ap.add_argument("--preroll", type=int, default=None, help="untimed plies played before warm-up to reach steady state (0 = opening phase; default 640, --fast: 10)")
    ap.add_argument("--opening-steps", type=int, default=20, help="plies of the labelled opening-only extra measurement (0 = skip)")
    ap.add_argument("--softmax", default="torch", choices=["torch", "engine"],
                    help="policy softmax: in the evaluate stage (torch: torch.softmax as in the reference, or the stage's own head kernel bo_k_heads_rows: the "
                         "seam the oracle replays are recorded at) or in the tree kernel that consumes the row (engine; with the hand-written float32 heads the "
                         "step kernel then finishes the value head too, bo_step_heads -- same bits, same speed: profiles/r05_device_turn_and_tiles.md section 6)")
    ap.add_argument("--exchange-every", type=int, default=16, help="N>1: plies per record-exchange period")
    ap.add_argument("--dump-games", default="", help="write the move lists of finished games to this JSON file")
    ap.add_argument("--wide-trees", type=int, default=0, help="0 = 262144 if >= 150 GB of HBM is free, else 131072")
    ap.add_argument("--wide-nodes", type=int, default=800)
    ap.add_argument("--fast", action="store_true", help="FAST search mode (virtual loss; not the reference's semantics; not the headline)")
    ap.add_argument("--leaves", type=int, default=4, help="FAST mode: leaves per game per step")
    ap.add_argument("--select-games-per-halfwave", type=int, default=None, help="FAST mode: bo_fast_options games_per_halfwave (2 or 4)")
    ap.add_argument("--select-flags", type=int, default=None, help="FAST mode: bo_fast_options select_flags (1 nt, 2 root in registers, 4 dense)")
    ap.add_argument("--select-sweep", action="store_true", help="FAST mode: time every variant of the select + backup kernel on this run's trees")
    ap.add_argument("--roofline-steps", type=int, default=48, help="FAST mode: evaluate -> step iterations run eagerly for the select kernel's event timing")
    ap.add_argument("--arena-granules-per-expansion", type=int, default=12, help="FAST mode: arena size per game = this x (sims + leaves + 2) granules of 128 B")
    ap.add_argument("--max-game-moves", type=int, default=None, help="config.MAX_GAME_MOVES (small values make games finish: record path; default 16384, --fast: 510)")
    ap.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"], help="gloo: rehearsal of the N>1 path on one GPU")
    ap.add_argument("--force-dist", action="store_true", help="initialise torch.distributed even with WORLD_SIZE=1")
    ap.add_argument("--share-gpu", action="store_true", help="rehearsal: every rank uses cuda:0")
    ap.add_argument("--long-run-steps", type=int, default=None,
                    help="plies of the confirmation leg run AFTER the timed steps (same workload, same timing brackets; reported as long_run); "
                         "default 2000 at N=1 with the default workload (~5.5 s of GPU work: longer than a 5-second utilisation sampler's period), 0 elsewhere")
    args = ap.parse_args()
    # --fast is priced on its own workload: enough resident games for the select + backup kernel to be
    # bandwidth-bound (a descent is a chain of dependent reads), a small fp16 net so that the evaluate stage does not starve it
    d_ref = dict(games=256, sims=800, net="10x128", net_dtype="fp32", preroll=640, max_game_moves=16384, cohorts=None)
    d_fast = dict(games=32768, sims=800, net="10x128", net_dtype="fp16", preroll=3, max_game_moves=510, cohorts=1)
    for k, v in (d_fast if args.fast else d_ref).items():
        if getattr(args, k) is None:
            setattr(args, k, v)
    if args.cohorts is None:  # phase-shifted cohorts of >= 64 games (measured at 256 games: two +4..7 % per ply, four on disjoint CU sets +2.7 % more)
        args.cohorts = next((k for k in (4, 2) if args.games % k == 0 and args.games // k >= 64), 1)
    if args.cohorts < 1 or args.games % args.cohorts or (args.fast and args.cohorts > 1):
        ap.error("--cohorts must divide --games (and is 1 with --fast)")
    if args.fast and args.steps == 20 and args.warmup == 3:
        args.steps, args.warmup = 1, 0  # (a ply of 32768 games x 800 simulations is 26 M evaluations: ~12 s)
    if args.fast and args.opening_steps == 20:
        args.opening_steps = 0          # (the opening-only extra would be twenty such plies)
    if args.long_run_steps is None:     # ~5.5 s of the default workload; off for other workloads, N > 1 and with measurement switches off
        default_workload = (not args.fast and args.gpus == 1 and (args.games, args.sims, args.net, args.net_dtype) == (256, 800, "10x128", "fp32"))
        args.long_run_steps = 2000 if (default_workload and args.preroll > 0 and not args.no_roofline) else 0
    return args


def make_net(name, device, dtype, batch=256, f32_pipe=None):
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

    return net, best_inference_copy(net, batch, device, td, f32_pipe=f32_pipe)


class CastIn(torch.nn.Module):
    """fp16/bf16 evaluate stage: the engine writes float32 planes; cast once in front of the net."""

    def __init__(self, net, dtype):
        super().__init__()
        self.net, self.dtype = net, dtype

    def forward(self, x):
        if getattr(self.net, "wants_float32_input", False):  # the fp16 tower converts the planes itself
            return self.net(x)
        return self.net(x.to(self.dtype))

    def tail_supported(self, batch):  # (logits, value_fc1 partial sums) for Engine.step_heads: passed through to a net that takes float32 planes
        return bool(getattr(self.net, "wants_float32_input", False) and hasattr(self.net, "forward_tail") and self.net.tail_supported(batch))

    def forward_tail(self, x):
        return self.net.forward_tail(x)

    def tail_params(self):
        return self.net.tail_params()

    def forward_probs(self, x):  # (softmax(logits), value) where the net has its own fused form
        if hasattr(self.net, "forward_probs") and getattr(self.net, "wants_float32_input", False):
            return self.net.forward_probs(x)
        logits, value = self.forward(x)
        return torch.softmax(logits.float(), dim=1), value


def softmax_site(net, args, rows):
    """Where the policy softmax of mcts.py:185,287 actually runs in this configuration (the label follows the code path)."""
    if args.softmax != "torch":
        if not args.fast and hasattr(net, "forward_tail") and net.tail_supported(rows) and os.environ.get("BETAONE_STEP_TAIL", "1") != "0":
            return ("bo_k_step (bo_step_heads: the wave that consumes a row finishes it -- softmax and the value head's last layer, bit for bit what "
                    "bo_k_heads_rows writes; the evaluate stage ends behind bo_k_heads_tiles)")
        return "bo_k_step / bo_k_fw_apply (the tree kernel's own softmax over the row)"
    inner = getattr(net, "net", net)
    conv = getattr(inner, "conv", None)
    if getattr(inner, "fused_heads", False) and hasattr(inner, "forward_probs") and (conv != "tower_f16" or rows <= 1024) \
            and (args.net_dtype == "fp32" or getattr(inner, "wants_float32_input", False)):
        return "bo_k_heads_rows (hand-written head kernel csrc/bo_heads.h: the row's softmax in registers, float32)"
    return "torch.softmax(logits.float(), dim=1) inside the captured graph"


def select_roofline(args, device):
    """HBM roofline of the PUCT-select kernel on the SURVEY section 8d wide workload."""
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    import select_wide_lab as SW  # lab harness of the synthetic wide-tree kernel (include/betaone_lab.h), not part of the package

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


# HBM-side bytes per launch come from separate `rocprofv3 --pmc` passes (counters cannot be read from inside the process): the line
# carries `traffic: null` and names the committed pass; no constant from an earlier run is printed as if this run had measured it.
TOWER_PMC_SOURCE = {"tower_split": "profiles/r03_tower_split_pmc.md (separate rocprofv3 --pmc pass of this kernel, 256 boards of the 10x128 net: 2 x FETCH_SIZE + WRITE_SIZE per launch)",
                    "tower_wg": "profiles/r02_tower_wg_pmc.md (separate rocprofv3 --pmc pass of this kernel, 256 boards x 128 filters)"}
# This round's pass of the split-precision tower at the bench's own launch shape (scripts/gpu_round.sh pmcsplit64 -> profiles/r05_tower_pmc.json:
# FETCH_SIZE x 2 (the gfx950 correction of MI355X_MICROARCH.md) + WRITE_SIZE per launch, per kernel name and boards per launch)
TOWER_PMC_JSON = os.path.join(ROOT, "profiles", "r05_tower_pmc.json")
# what one compute unit can take in from its XCD's L2 (scripts/weight_stream_lab.hip, profiles/r05_tower_bound.md): the bound of a kernel that
# keeps one board per workgroup and therefore streams the whole tower's weights through every CU
L2_PORT_LAB_GBPS_PER_CU, L2_PORT_LAB_SOURCE = 70.2, "profiles/r05_tower_bound.md (scripts/weight_stream_lab.hip: 29.4 B/clk per CU at 2.39 GHz with 48 KiB in flight per wave, 64 and 256 CUs alike; 27.0 B/clk with 24 KiB)"


def tower_pmc_traffic(kernel, boards):
    """(bytes per launch, source) from the committed PMC pass of `kernel` at `boards` boards per launch, or (None, why not)."""
    try:
        rec = json.load(open(TOWER_PMC_JSON))
    except Exception:
        return None, TOWER_PMC_SOURCE.get("tower_split")
    for e in rec.get("launch_shapes", []):
        if e.get("kernel") == kernel and int(e.get("boards", -1)) == int(boards):
            return float(e["hbm_side_bytes_per_launch"]), f"profiles/r05_tower_pmc.json ({e.get('source', 'rocprofv3 --pmc')}; ratio to algorithmic bytes {e.get('ratio_to_algorithmic')})"
    return None, f"profiles/r05_tower_pmc.json has no pass of {kernel} at {boards} boards per launch"


def tower_timings(parts, seq0, khz):
    """Durations (us) and [start, end) intervals (us, common clock) of the tower launches each Rollout made since `seq0` -- noted by the
    kernel itself (bo_nn_tower_forward_timed), so launches inside the captured graphs of the timed region are measured, not re-runs."""
    dur, iv = [], []
    for p, s0 in zip(parts, seq0):
        t = p.tower_timing.cpu().numpy().astype(np.uint64)
        s1, cap = int(t[0]), 4096
        lo = max(s0, s1 - cap)
        for k in range(lo, s1):
            a, b = int(t[2 + k % cap]), int(t[2 + cap + k % cap])
            if b > a:
                dur.append((b - a) / khz * 1e3)
                iv.append((a / khz * 1e3, b / khz * 1e3))
    return dur, iv


def _union(iv):
    tot, cs, ce = 0.0, None, None
    for a, b in sorted(iv):
        if ce is None or a > ce:
            if ce is not None:
                tot += ce - cs
            cs, ce = a, b
        else:
            ce = max(ce, b)
    return tot + (ce - cs if ce is not None else 0.0)


def _tower_back_to_back_us(net, batch, device, heads, per_graph=20, replays=4):
    """The tower kernel alone on the chip at `batch` boards: `per_graph` consecutive graph nodes, `replays` replays, microseconds per launch."""
    x = torch.rand((batch, 120, 8, 8), device=device)
    with torch.no_grad():
        side = torch.cuda.Stream(device)
        side.wait_stream(torch.cuda.current_stream(device))
        with torch.cuda.stream(side):
            for _ in range(3):
                net._tower_forward(x, heads=heads)
        torch.cuda.current_stream(device).wait_stream(side)
        torch.cuda.synchronize(device)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, capture_error_mode="thread_local"):  # (the RCCL watchdog thread may query events meanwhile)
            keep = [net._tower_forward(x, heads=heads) for _ in range(per_graph)]
        g.replay()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(replays):
            g.replay()
        e1.record()
        e1.synchronize()
        del keep
    return e0.elapsed_time(e1) * 1e3 / (per_graph * replays)


def nn_roofline(net, batch, device, timed=None, wall_us=None, all_boards=None, ply_us=None, launches_per_ply=None):
    """MFMA roofline of the evaluate stage's tower kernel.  `achieved` / `frac` = ALGORITHMIC flops per launch (SURVEY.md section 8d:
    2 x MACs of the tower's direct 3x3 convolutions) / the kernel's average launch duration IN THE TIMED REGION, which the kernel notes
    itself (first workgroup's start, last workgroup's end, constant-rate device clock: `timed` = (durations us, intervals) from
    tower_timings) -- event pairs cannot sit between the nodes of the captured graphs the timed region replays; rocprofv3's per-kernel
    average of the same command is the check (profiles/).  With cohorts several launches of a share of the boards overlap: `concurrency` =
    sum of durations / time with at least one launch running; `achieved` prices a launch against the chip's time per launch (see below).
    `back_to_back_us`: the same kernel alone on the chip, replayed as 20 consecutive graph nodes.
    `towers_only`: the same for ONE launch over all `all_boards` games of the GPU (every CU on the matrix pipe, nothing else on the chip)
    x the evaluations a ply needs = the ply if the chip did nothing but towers, against the measured ply (`ply_us`)."""
    conv = getattr(net, "conv", None)
    if conv not in ("tower", "tower_wg", "tower_split"):
        return None
    heads = conv in ("tower_wg", "tower_split")
    C, n_conv = net.c, 1 + 2 * len(net.blocks)
    net.__dict__.pop("tower_timing_buf", None)
    per_graph, replays = 20, 4
    b2b = _tower_back_to_back_us(net, batch, device, heads, per_graph, replays)
    towers_only = None
    if all_boards and ply_us and launches_per_ply:
        t_all = _tower_back_to_back_us(net, all_boards, device, heads, per_graph, replays)
        towers_only = {"boards": all_boards, "lone_launch_us": round(t_all, 1), "evaluations_per_ply": round(launches_per_ply, 2),
                       "ply_ms_if_towers_only": round(t_all * launches_per_ply / 1e3, 3), "ply_ms_measured": round(ply_us / 1e3, 3),
                       "frac": round(t_all * launches_per_ply / ply_us, 3),
                       "note": "one launch of this kernel over ALL the GPU's games, alone on the chip, back to back (every CU on the matrix pipe: the "
                               "clock the chip holds then) x the evaluations one game's ply needs, against the ply of the timed region: what is left "
                               "for better overlap of tree steps, heads and gaps -- the rest is the tower's own rate"}
    us, n_timed, how, conc = b2b, per_graph * replays, "20 consecutive graph nodes x 4 replays (no in-kernel timings for this kernel)", None
    if timed and timed[0]:
        dur, iv = timed
        us, n_timed = float(np.mean(dur)), len(dur)
        conc = float(sum(dur) / max(_union(iv), 1e-9))
        how = ("every launch of the timed region, noted by the kernel itself (first workgroup's start -> last workgroup's end, wall_clock64; "
               "bo_nn_tower_forward_timed)")
    per_mac = 16 * 16 if conv == "tower_wg" else 3 * 9 * 64 if conv == "tower_split" else 9 * 64  # multiplies per (c_in, c_out) pair and board
    executed = 2.0 * per_mac * C * (128 + (n_conv - 1) * C) * batch
    # SURVEY.md section 8d's per-unit figure: 2 x MACs of the direct 3x3 convolutions of the tower, per board (input conv: 120 planes)
    algorithmic = 2.0 * 9 * 64 * C * (120 + (n_conv - 1) * C) * batch
    # `achieved` / `frac` = ALGORITHMIC flops per launch / the measured launch duration against the dense peak of the pipe the kernel
    # multiplies on; what the kernel EXECUTES on that pipe (three fp16 MFMAs per float32 product; Winograd: 16 multiplies per 2x2
    # tile; the input conv padded to 128 channels) is reported beside it as `pipe_utilisation`.
    if conv == "tower_split":
        t16 = getattr(net, "split_tile", 32) == 16
        peak, kernel = 2500.0, ("bo_k_tower_s16" if t16 else "bo_k_tower_s")   # fp16 MFMA dense (MI355X_MICROARCH.md)
        note = ("float32 planes in and out; every float32 operand a (hi, lo) fp16 pair, every product three "
                + ("v_mfma_f32_16x16x32_f16" if t16 else "v_mfma_f32_32x32x16_f16") + " with "
                "float32 accumulation (direct 3x3 form); one workgroup per board, activations LDS-resident for the whole tower")
    else:
        peak, kernel = 157.3, ("bo_k_tower_wg" if conv == "tower_wg" else "bo_k_tower")  # fp32 MFMA dense: 256 CUs x 4 SIMDs x 64 flop/clk x 2.4 GHz
        note = "fp32 v_mfma_f32_16x16x4_f32; one workgroup per board, activations LDS-resident for the whole tower"
    # Cohorts launch this kernel for a share of the boards on a share of the CUs, several launches at a time: the time the CHIP spends per
    # launch is (time with at least one launch running) / launches = average duration / concurrency, and `achieved` / `frac` price the
    # algorithmic flops of a launch against THAT (with one launch at a time it is the launch's own duration).  `per_launch` keeps the
    # single launch against the whole chip and against the CUs it can occupy (one workgroup per board: boards / CUs of the chip).
    n_cu = torch.cuda.get_device_properties(device).multi_processor_count
    eff_us = us / conc if conc else us
    ach, ach1 = algorithmic / eff_us / 1e6, algorithmic / us / 1e6
    share = min(1.0, batch / float(n_cu))
    traffic, traffic_src = (tower_pmc_traffic(kernel, batch) if conv == "tower_split" else (None, TOWER_PMC_SOURCE.get(conv)))
    l2_port = None
    if conv == "tower_split":  # one board per workgroup: every CU streams the whole tower's (hi, lo) weight pairs -- what actually bounds a launch
        w_bytes = 4.0 * 9 * C * (128 + (n_conv - 1) * C)
        l2_port = {"bound": "per-CU L2 port (the weight stream of one board per workgroup)", "weight_bytes_per_cu_per_launch": w_bytes,
                   "achieved_GBps_per_cu": round(w_bytes / us / 1e3, 1), "alone_GBps_per_cu": round(w_bytes / b2b / 1e3, 1),
                   "lab_peak_GBps_per_cu": L2_PORT_LAB_GBPS_PER_CU, "frac": round(w_bytes / us / 1e3 / L2_PORT_LAB_GBPS_PER_CU, 3),
                   "frac_alone": round(w_bytes / b2b / 1e3 / L2_PORT_LAB_GBPS_PER_CU, 3), "lab": L2_PORT_LAB_SOURCE,
                   "note": "the port delivers bytes per CLOCK: `achieved` (launches of the timed region, several cohorts' towers in flight, clock "
                           "lowered by the matrix load) sits below `alone` by the clock ratio; `alone` is the lone launch's rate"}
    return {"bound": "mfma", "kernel": kernel, "achieved": round(ach, 1), "peak": peak, "unit": "TFLOP/s", "frac": round(ach / peak, 4),
            "traffic": traffic, "traffic_source": traffic_src, "l2_port": l2_port,
            "basis": "algorithmic flops = 2 x MACs of the tower's direct 3x3 convolutions (2*9*64*C*(120 + (layers-1)*C) per board) x boards per launch, "
                     "/ the chip's time per launch = time with >= 1 launch of this kernel running / launches (= avg_launch_us / concurrency)",
            "alg_flops_per_launch": algorithmic, "executed_mfma_flops_per_launch": executed,
            "pipe_utilisation": round(executed / eff_us / 1e6 / peak, 4),
            "executed_tflops": round(executed / eff_us / 1e6, 1),
            "chip_us_per_launch": round(eff_us, 1),
            "avg_launch_us": round(us, 1), "launches_timed": n_timed, "timing": how, "back_to_back_us": round(b2b, 1), "towers_only": towers_only,
            "launch_us_p10_p50_p90": ([round(float(v), 1) for v in np.percentile(timed[0], [10, 50, 90])] if timed and timed[0] else None),
            "concurrency": (round(conc, 3) if conc else None),
            "share_of_wall_time": (round(_union(timed[1]) / wall_us, 4) if (conc and wall_us) else None),  # time with >= 1 tower launch running / the timed region
            "per_launch": {"achieved": round(ach1, 1), "frac_of_whole_chip": round(ach1 / peak, 4), "cu_share": round(share, 4),
                           "frac_of_cu_share": round(ach1 / (peak * share), 4), "pipe_utilisation_of_cu_share": round(executed / us / 1e6 / (peak * share), 4)},
            "boards_per_launch": batch, "conv_layers": n_conv, "note": note}


def _fast_counters(eng):
    st, fs = eng.status(), eng.fast_stats()
    return dict(levels=int(st["levels"].astype(np.int64).sum()), kids=int(st["children_scanned"].astype(np.int64).sum()),
                gran=int(fs["granules_read"].sum()), pnodes=int(fs["path_nodes"].sum()), arena=int(fs["arena_granules"].astype(np.int64).sum()))


def fast_select_measure(ro, drv, steps):
    """`steps` more evaluate -> step iterations of the searches in progress, launched eagerly (same engine, same games, no graph)
    with HIP events around every launch of the select + backup kernel on its stream; bytes from the engine's own counters over the
    same launches.  (The searches were begun by the last ply's turn; the next ply simply finds them that much further along.  When
    they would run out of simulations before `steps` launches, a ply is played first.)"""
    if getattr(drv, "extra_steps", 0) + steps > ro.expected_evals - 8:
        drv.step()
        drv.extra_steps = 0
    drv.extra_steps = getattr(drv, "extra_steps", 0) + steps
    eng = ro.eng
    c0 = _fast_counters(eng)
    prof = os.environ.get("BO_SELECT_PROFILE", "0") not in ("", "0")
    if prof:
        eng.profile(0, read=False)
        eng.profile(1, read=False)  # (0 -> 1 clears the counters)
    eng.fast_stats(time_select=1)
    for _ in range(steps):
        ro._eval_and_step_eager()
    torch.cuda.synchronize(ro.device)
    fs1 = eng.fast_stats(time_select=0)
    if prof:  # per-wave shader cycles of the kernel's phases (csrc/bo_fastw.h), averaged over the waves of these launches
        pr = eng.profile(0).astype(np.float64).sum(axis=0)
        w = max(1.0, pr[5])
        print("[select profile] waves/launch %.0f  cycles per wave: ctl+backup %.0f | wait for the backup's stores %.0f | descents %.0f | tail %.0f | "
              "level-loop iterations %.1f (%.0f cycles each)" % (w / max(1, steps), pr[0] / w, pr[1] / w, pr[2] / w, pr[3] / w, pr[4] / w, pr[2] / max(1.0, pr[4])),
              file=sys.stderr, flush=True)
    c1 = _fast_counters(eng)
    launches = int(fs1["select_launches"])
    if launches == 0:
        return None
    # an event pair around ONE launch also measures the pair itself (~6 us on this part, 10 % of this kernel): calibrated on an empty
    # kernel on the same stream and subtracted; rocprofv3's per-dispatch durations of the same command are the check (profiles/)
    ov = eng.event_pair_overhead_ms(32, ro._stream()) * 1e-3
    d = {k: c1[k] - c0[k] for k in ("levels", "kids", "gran", "pnodes")}
    alg = 12 * d["kids"] + 8 * d["levels"] + 16 * d["pnodes"]
    moved = eng.GRANULE_BYTES * d["gran"] + 16 * d["pnodes"]           # what the kernel requests: whole record granules + the backup's (n, w) pairs
    t_raw = fs1["select_ms"] * 1e-3 / launches
    t = max(t_raw - ov, 0.5 * t_raw)
    return dict(launches=launches, t=t, t_raw=t_raw, overhead=ov, alg=alg / launches, moved=moved / launches, levels=d["levels"] / launches,
                kids=d["kids"] / launches, pnodes=d["pnodes"] / launches, arena_bytes=c1["arena"] * eng.GRANULE_BYTES)


FAST_PMC_JSON = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r05_fast_select_pmc.json")


def fast_pmc_traffic(games, leaves):
    """The committed PMC pass of bo_k_fw_select at this launch shape: (HBM-side bytes per launch, ratio to that pass's algorithmic bytes, source)."""
    try:
        for e in json.load(open(FAST_PMC_JSON)).get("launch_shapes", []):
            if int(e.get("games", -1)) == int(games) and int(e.get("leaves_per_step", -1)) == int(leaves):
                return float(e["hbm_side_bytes_per_launch"]), float(e["ratio_to_algorithmic"]), f"profiles/r05_fast_select_pmc.json ({e.get('source')})"
    except Exception:
        pass
    return None, None, "no PMC pass of bo_k_fw_select at this launch shape is committed (scripts/gpu_round.sh pmcfast pmcfastw)"


def fast_select_roofline(ro, drv, steps, label=""):
    """FAST mode: the select + backup kernel (csrc/bo_fastw.h: bo_k_fw_select) on the trees the searches of this run grew --
    no synthetic topology; virtual loss and backup included (12 B per child scanned + 8 B per level;
    backup 16 B per path node)."""
    m = fast_select_measure(ro, drv, steps)
    if m is None:
        return None
    ach = m["alg"] / m["t"] / 1e9
    traffic, ratio, tsrc = fast_pmc_traffic(ro.G, ro.L)
    return {"bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4),
            "traffic": traffic, "traffic_source": tsrc,
            # what the counters say the kernel moves per second at this run's launch duration (the PMC pass's ratio of HBM-side to algorithmic
            # bytes x this run's algorithmic rate): the north star's "rocprof-reported HBM GB/s" -- beside `frac`, never instead of it
            "hbm_side": ({"GBps": round(ach * ratio, 1), "frac_of_peak": round(ach * ratio / HBM_PEAK_GBS, 4), "ratio_to_algorithmic": ratio,
                          "note": "FETCH_SIZE x 2 + WRITE_SIZE of the committed PMC pass over that pass's algorithmic bytes, applied to this run's rate: "
                                  "the kernel keeps this much of the HBM busy; `frac` counts only the bytes the algorithm needs"} if ratio else None),
            "kernel": "bo_k_fw_select" + label,
            "workload": f"the search trees of this run ({ro.G} games x {ro.L} descents per launch under a virtual loss, backup of the previous "
                        f"launch's simulations in the same kernel; arenas of 128-byte granules, {m['arena_bytes'] / 1e9:.2f} GB live); "
                        f"bytes = 12 B x children scanned + 8 B x levels + 16 B x path nodes; the kernel requests whole 128-byte record granules",
            "launches_timed": m["launches"], "avg_launch_us": round(m["t"] * 1e6, 2),
            "timing": f"HIP event pair around every launch ({m['t_raw'] * 1e6:.2f} us) minus the pair's own cost calibrated on empty kernels on the same "
                      f"stream ({m['overhead'] * 1e6:.2f} us = 2 x pair(1 launch) - pair(2 launches), medians of 32); rocprofv3 per-dispatch durations of the same command: profiles/r03_fast_select_rocprof.md",
            "alg_bytes_per_launch": int(m["alg"]),
            "moved_bytes_per_launch": int(m["moved"]), "moved_over_algorithmic": round(m["moved"] / m["alg"], 3),
            "levels_per_launch": int(m["levels"]), "children_per_level": round(m["kids"] / max(1.0, m["levels"]), 2),
            "levels_per_descent": round(m["levels"] / max(1.0, m["pnodes"] - m["levels"]), 2),
            "moved_GBps": round(m["moved"] / m["t"] / 1e9, 1)}


def fast_select_sweep(ro, drv, steps=12):
    """Every variant of the select + backup kernel over `steps` more launches each on this run's trees (they keep growing meanwhile;
    when the searches run out of simulations a ply is played in between)."""
    out = []
    variants = ((2, 32), (2, 36), (2, 16)) if ro.G > 40000 else ((2, 32), (2, 36), (2, 34), (2, 18), (2, 16), (2, 19), (2, 0), (4, 8))
    for ut, fl in variants:  # (flags 32 / 34: four lanes per game; 16..19: eight; 0..7: half a wave; 8 / 9: one lane)
        if True:
            ro.eng.fast_options(games_per_halfwave=ut, select_flags=fl)
            r = fast_select_roofline(ro, drv, steps, f" u{ut} flags{fl}")
            if r:
                out.append({k: r[k] for k in ("kernel", "frac", "achieved", "avg_launch_us", "moved_GBps", "moved_over_algorithmic", "launches_timed",
                                              "levels_per_descent", "children_per_level")})
                print("[sweep]", json.dumps(out[-1]), file=sys.stderr, flush=True)
    return out


def step_roofline(ro, n_steps_timed):
    """The in-loop tree step kernel: algorithmic bytes from the engine's own counters."""
    st = ro.eng.status()
    return {"levels": int(st["levels"].sum()), "children_scanned": int(st["children_scanned"].sum()),
            "evals": int(st["evals"].sum()), "flushes": int(st["flushes"].sum())}


def _cpu_model():
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown CPU"


def _cpu_worker(job):
    """One host core = one self-play game at a time, like one worker of the reference's mp.Pool (main.py:39-65, :168):
    the CPU oracle (C port of mcts.py / self_play.py + python-chess rules) with the same net under torch-CPU, 1 thread."""
    net_name, sims, batch, seconds, worker, threads = job
    torch.set_num_threads(threads)
    from oracle import oracle as O

    net, _ = make_net(net_name, "cpu", "fp32")

    def eval_fn(planes):
        with torch.no_grad():
            logits, value = net(torch.from_numpy(np.ascontiguousarray(planes)))
            return torch.softmax(logits, dim=1).numpy(), value.reshape(-1).numpy()

    cfg = O.default_config(num_simulations=sims, batch_size=batch)
    O.self_play(eval_fn, np.random.RandomState(10_000 + worker), cfg, max_plies=1)  # warm-up (first torch call, oracle load)
    t0, n_sims, plies, games, evals = time.perf_counter(), 0, 0, 0, 0
    while time.perf_counter() - t0 < seconds:
        g = O.self_play(eval_fn, np.random.RandomState(1000 * worker + games), cfg, max_plies=6)
        n_sims += g["n_sims"]; plies += len(g["moves"]); evals += g["n_evals"]; games += 1
    return n_sims, plies, evals, games, time.perf_counter() - t0


def cpu_baseline(args):
    """The reference's algorithm on the host CPU, shaped like the reference's own parallelism: one
    game per core x cores, plus the single-game figure with the net on all those threads."""
    import multiprocessing as mp

    cores = args.cpu_cores or max(1, min(len(os.sched_getaffinity(0)), 16))
    os.environ["BETAONE_DEVICE"] = "cpu"  # the helper processes never open the GPU (config.py would probe it at import)
    ctx = mp.get_context("spawn")
    with ctx.Pool(cores) as pool:
        res = pool.map(_cpu_worker, [(args.net, args.sims, args.batch, args.cpu_seconds, w, 1) for w in range(cores)])
    del os.environ["BETAONE_DEVICE"]
    rates = [r[0] / r[4] for r in res]
    total = float(sum(rates))
    sims, plies, evals, games = (sum(r[k] for r in res) for k in range(4))
    one = _cpu_worker((args.net, args.sims, args.batch, min(8.0, args.cpu_seconds), 999, cores))
    return {"value": round(total, 1), "unit": "nodes/s", "cores": cores, "kind": "port", "cpu": _cpu_model(),
            "per_core": round(total / cores, 1), "per_core_min_max": [round(min(rates), 1), round(max(rates), 1)],
            "single_game_all_threads": round(one[0] / one[4], 1),
            "sample": f"{cores} processes x 1 thread, one game at a time each (the reference's mp.Pool shape, main.py:168), {args.cpu_seconds:.0f} s: "
                      f"{games} game prefixes x 6 plies ({plies} searches of {args.sims} sims, {evals} unique NN evals); oracle/ C port + "
                      f"net {args.net} fp32 under torch-CPU; the port evaluates each unique leaf once (the Python reference evaluates up to 96 "
                      f"duplicate rows per batch and runs the rules in Python); single_game_all_threads = 1 game, net on {cores} threads"}


def comm_census(dist, args, device, rank, world):
    """Did the process group the record exchange runs on really form over `world` ranks on `world` different GPUs?  One all-reduce of
    ones and one all-gather of (host, device index, device UUID / PCI bus id) over that group, before the timed region -- the N > 1
    line then answers "did RCCL see N ranks?" by itself; the gloo rehearsal runs the same code."""
    import socket

    rdev = device if args.dist_backend == "nccl" else torch.device("cpu")
    one = torch.ones(1, dtype=torch.float64, device=rdev)
    dist.all_reduce(one, op=dist.ReduceOp.SUM)
    prop = torch.cuda.get_device_properties(device)
    ident = str(getattr(prop, "uuid", "")) or f"pci:{getattr(prop, 'pci_bus_id', '?')}:{getattr(prop, 'pci_device_id', '?')}"
    mine = (socket.gethostname(), int(device.index or 0), ident)
    every = [None] * world
    dist.all_gather_object(every, mine)
    distinct = len({(h, u) for h, _, u in every})
    torch.cuda.synchronize(device)  # (the census is over before anything else is enqueued)
    return {"backend": dist.get_backend(), "world_size": world, "world_size_seen": int(round(float(one.item()))),
            "ranks_on_distinct_devices": distinct == world, "distinct_devices": distinct,
            "devices": [{"rank": r, "host": h, "device_index": i, "device_id": u} for r, (h, i, u) in enumerate(every)],
            "collectives_used_for_this_check": ["all_reduce(SUM) of ones", "all_gather_object"]}


def self_launch(args):
    """`python bench.py --gpus N` (N > 1, not under torchrun): start the N ranks as a child torch.distributed.run job -- before
    this process has touched a GPU -- and exit with its code; rank 0 of the child prints the JSON line."""
    from betaone_amd.selfplay_main import launch_ranks

    sys.stdout.flush()
    return launch_ranks(args.gpus, [os.path.abspath(__file__)] + sys.argv[1:])


class Driver:
    """The step loop around one Rollout: finished games are counted (and handed to the exchange), their slots refilled."""

    def __init__(self, ro, rank, world, exchange, dump=None):
        self.ro, self.rank, self.world, self.exchange, self.dump = ro, rank, world, exchange, dump
        self.next_id = rank
        self.batch, self.n_finished, self.plies_finished, self.lengths = [], 0, 0, []
        self.n_received = 0
        self.exchange_seconds = 0.0

    def new_id(self):
        i = self.next_id
        self.next_id += self.world
        return i

    def on_finished(self, fin):
        self.batch.append(fin)
        self.n_finished += 1
        self.plies_finished += len(fin.moves)
        self.lengths.append(len(fin.moves))
        if self.dump is not None:
            self.dump.append({"game_id": fin.game_id, "terminal": fin.terminal, "moves": [int(m) for m in fin.moves]})

    def refill(self, _slot):
        i = self.new_id()
        return i, i, None  # game id -> rank = id mod world; RandomState(seed = game id) stream

    def step(self):
        self.extra_steps = 0  # (evaluate -> step iterations run by a measurement since the last ply)
        self.ro.play_ply(on_finished=self.on_finished, refill=self.refill, while_searching=self.hand_over)

    def hand_over(self):
        """The path's only exchange step: finished games' records to every rank (RCCL over xGMI).  Called by play_ply once the
        ply's searches are enqueued: packing the records (~0.2 ms per game on the host) overlaps the search."""
        if self.exchange is not None:
            t0 = time.perf_counter()
            self.n_received += len(self.exchange.push(self.batch))  # pipelined: nothing here waits for a collective
            self.exchange_seconds += time.perf_counter() - t0
        self.batch.clear()

    def preroll(self, plies, G):
        """Untimed: bring the resident games to every stage of a game.  Slot s enters play at step s * (plies / 2) / G, so after
        `plies` steps every slot has been playing for at least plies / 2 steps and the starts are spread evenly."""
        ro, window = self.ro, max(1, plies // 2)
        started = 0
        for k in range(plies):
            upto = G if plies < 2 else min(G, (k + 1) * G // window + 1)
            if upto > started:
                slots = list(range(started, upto))
                ids = [self.new_id() for _ in slots]
                ro.start_games(slots, ids, ids)
                started = upto
            self.step()
        if started < G:
            slots = list(range(started, G))
            ids = [self.new_id() for _ in slots]
            ro.start_games(slots, ids, ids)


def main():
    args = parse()
    if args.share_gpu and args.gpus > 1 and args.dist_backend == "nccl":
        # More ranks than devices: RCCL (like NCCL) rejects a communicator with two ranks on one device.  Refused here, before
        # any rank is started or any GPU call is made -- not found out by starting ranks that are expected to fail.
        print("bench.py: --share-gpu puts every rank on cuda:0 and RCCL does not form a communicator with two ranks on one device; "
              "use --dist-backend gloo for the one-GPU rehearsal of the N>1 path", file=sys.stderr)
        sys.exit(2)
    env_world = os.environ.get("WORLD_SIZE")
    if args.gpus > 1 and not args.share_gpu and torch.cuda.device_count() < args.gpus:
        # (device_count() does not initialise the GPU: this is decided before any rank touches one)
        print(f"bench.py: --gpus {args.gpus} but this node has {torch.cuda.device_count()} GPU(s); one rank per GPU "
              "(--share-gpu --dist-backend gloo rehearses the N>1 path on one GPU)", file=sys.stderr)
        sys.exit(2)
    if args.gpus > 1 and env_world is None:
        sys.exit(self_launch(args))
    world = int(env_world or "1")
    if world != args.gpus and not args.force_dist:
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}", file=sys.stderr)
        sys.exit(2)
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if args.share_gpu:
        local = 0
        if args.cu_masks is None and "BETAONE_COHORT_CU_MASK" not in os.environ:
            args.cu_masks = "off"  # (ranks sharing one GPU would all confine their cohort k to the same CUs)
    if world > 1 or args.force_dist:
        import torch.distributed as dist

        torch.cuda.set_device(local)
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device(f"cuda:{local}"))
        else:
            dist.init_process_group("gloo")
    device = torch.device(f"cuda:{local}")
    torch.cuda.set_device(device)
    comm = comm_census(dist, args, device, rank, world) if dist is not None else None

    from betaone_amd import engine as E
    from betaone_amd import records
    from betaone_amd.rollout import CohortRollout, Rollout

    E.load_hip_library()
    _, net = make_net(args.net, device, args.net_dtype, (args.games // args.cohorts) * (args.leaves if args.fast else 1),
                      f32_pipe=None if args.f32_tower is None else args.f32_tower == "fp32")
    net_layout = getattr(net, "layout", "nchw")
    if args.net_dtype != "fp32":
        net = CastIn(net, {"fp16": torch.float16, "bf16": torch.bfloat16}[args.net_dtype])
    G = args.games
    ro_kw = dict(num_simulations=args.sims, mcts_batch_size=args.batch, device=str(device), use_graph=not args.no_graph,
                 rng_mode="native", max_game_moves=args.max_game_moves, fast=args.fast, leaves_per_step=args.leaves,
                 fast_arena_granules=(args.arena_granules_per_expansion * (args.sims + args.leaves + 2) if args.fast else 0),
                 policy_kind="probs" if args.softmax == "torch" else "logits",
                 time_tower=(not args.no_roofline and not args.fast and args.net_dtype == "fp32"))
    ro = CohortRollout(net, G, cohorts=args.cohorts, cu_masks=args.cu_masks, **ro_kw) if args.cohorts > 1 else Rollout(net, G, **ro_kw)
    if args.fast and G * args.leaves > 65536:
        ro.MAX_GRAPH_ITERATIONS = 4 if G * args.leaves <= 131072 else 2  # (every iteration of a captured graph keeps its own logits / probabilities: 6.5 GB at 131072 rows)
    if args.fast and (args.select_games_per_halfwave is not None or args.select_flags is not None):
        ro.eng.fast_options(games_per_halfwave=args.select_games_per_halfwave, select_flags=args.select_flags)
    exchange = records.PeriodicGameExchange(device, every=args.exchange_every) if dist is not None else None
    dump = [] if args.dump_games else None
    drv = Driver(ro, rank, world, exchange, dump)

    t_pre = time.perf_counter()
    if args.preroll > 0:
        drv.preroll(args.preroll, G)
    else:
        ids = [drv.new_id() for _ in range(G)]
        ro.start_games(list(range(G)), ids, ids)
    import gc

    gc.collect()  # (before the warm-up steps, not between them and the timed region: a full collection walks the heap and leaves the first
    gc.disable()  # timed step with cold caches -- +0.2 ms on its first cohort; no cyclic-GC pause inside a 0.1 s timed region: up to 5 % of 20 steps)
    for _ in range(args.warmup):
        drv.step()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize(device)
    t_pre = time.perf_counter() - t_pre
    parts = getattr(ro, "parts", [ro])
    timing_on = parts[0].tower_timing is not None
    seq0 = [int(p.tower_timing[0].item()) for p in parts] if timing_on else None
    fin_pre = drv.n_finished
    s0, p0, f0, h0, n0, pf0 = ro.n_sims, ro.n_plies, ro.n_forward, ro.host_seconds, drv.n_finished, drv.plies_finished
    t0 = time.perf_counter()
    step_end = []
    step_wait = []  # CohortRollout: the host's time without a ready cohort, per step
    for _ in range(args.steps):
        w0 = getattr(ro, "wait_seconds", 0.0)
        drv.step()
        step_end.append(time.perf_counter())
        step_wait.append(getattr(ro, "wait_seconds", 0.0) - w0)
    torch.cuda.synchronize(device)
    if dist is not None:
        dist.barrier()
    dt = time.perf_counter() - t0
    if os.environ.get("BO_PLY_PROFILE") == "trace":  # the host's phases inside the first two timed steps, in order
        for a_, b_, who, name in Rollout.PLY_TRACE:
            if t0 <= b_ <= step_end[min(1, len(step_end) - 1)]:
                print(f"[ply trace] {(a_ - t0) * 1e6:8.0f} -> {(b_ - t0) * 1e6:8.0f} us  cohort {who}  {name}", file=sys.stderr, flush=True)
    gc.enable()
    timed = None
    if timing_on:  # the tower launches of the timed region, timed by the kernel itself
        khz = C_INT32(0)
        E.load_hip_library().bo_device_wall_clock_khz(device.index or 0, ctypes.byref(khz))
        timed = tower_timings(parts, seq0, float(khz.value or 100000))
    # The exchange pipeline is drained after the K timed steps: every step did its own exchange tick (records of earlier plies
    # arrived during the timed region exactly as these will in the steps after it); the drain is an artefact of stopping.
    t_flush = time.perf_counter()
    if exchange is not None:
        drv.n_received += len(exchange.flush())
    t_flush = time.perf_counter() - t_flush
    mine = [ro.n_sims - s0, ro.n_plies - p0, ro.n_forward - f0, drv.n_finished - n0, drv.plies_finished - pf0,
            drv.n_finished, drv.plies_finished]
    per_rank_ms = None
    if dist is not None:
        rdev = device if args.dist_backend == "nccl" else torch.device("cpu")
        t = torch.tensor([dt], dtype=torch.float64, device=rdev)
        every = torch.zeros(world, dtype=torch.float64, device=rdev)
        dist.all_gather_into_tensor(every, t)   # each rank's own time over the K steps: stragglers show in the line
        per_rank_ms = [round(float(x) / args.steps * 1e3, 3) for x in every.cpu().tolist()]
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        blocked = torch.tensor([float(exchange.blocked_ticks if exchange is not None else 0)], dtype=torch.float64, device=rdev)
        dist.all_reduce(blocked, op=dist.ReduceOp.SUM)
        blocked_all = int(blocked.item())
        tot = torch.tensor(mine, dtype=torch.float64, device=rdev)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        mine = tot.tolist()
    sims, plies, fwd, fin_timed, fin_plies_timed, fin_all, fin_plies_all = [float(x) for x in mine]
    step_ms = np.diff(np.array([t0] + step_end)) * 1e3  # host-side period of each step on this rank (a step returns when its moves are played)
    arena_full = ro.eng.check_status()  # (raises on a fault; fast mode: slots whose arena was full at some expansion -- those searches ran narrower)
    ro.check_net()
    if getattr(ro, "ply_profile", None):  # BO_PLY_PROFILE=1: host seconds per phase of the native ply path over the whole run
        import collections
        if os.environ.get("BO_PLY_PROFILE", "0") not in ("", "0", "1"):  # a path: the raw list
            json.dump([[n, round(sec * 1e6, 1)] for n, sec in ro.ply_profile], open(os.environ["BO_PLY_PROFILE"], "w"))
        acc = collections.defaultdict(list)
        for name, sec in ro.ply_profile:
            acc[name].append(sec * 1e6)
        for name, v in acc.items():
            v.sort()
            print("[ply profile] %-18s n %5d  p50 %8.1f  p90 %8.1f  p99 %8.1f  max %8.1f  sum %10.1f us" % (name, len(v), v[len(v) // 2], v[int(len(v) * 0.9)], v[int(len(v) * 0.99)], v[-1], sum(v)),
                  file=sys.stderr, flush=True)
    host_frac = (ro.host_seconds - h0) / dt

    # Confirmation leg: the headline above is K = 20 plies from a drained pipeline (it pays the fill and the
    # drain, and a 60 ms window is over before a 5-second utilisation sampler looks); here the same workload goes on for `long_run_steps`
    # more plies under the same brackets.  Reported beside the headline, never instead of it.
    long_run = None
    if args.long_run_steps > 0 and dist is None:
        s1, p1, n1 = ro.n_sims, ro.n_plies, drv.n_finished
        gc.disable()
        torch.cuda.synchronize(device)
        tl = time.perf_counter()
        for _ in range(args.long_run_steps):
            drv.step()
        torch.cuda.synchronize(device)
        dl = time.perf_counter() - tl
        gc.enable()
        ro.eng.check_status()
        long_run = {"steps": args.long_run_steps, "ms_per_step": round(dl / args.long_run_steps * 1e3, 3), "nodes_per_sec": round((ro.n_sims - s1) / dl, 1),
                    "seconds": round(dl, 3), "games_finished": int(drv.n_finished - n1),
                    "games_per_hour": round((drv.n_finished - n1) * 3600.0 / dl, 1), "plies_per_sec": round((ro.n_plies - p1) / dl, 2),
                    "note": "same workload and brackets as the headline, continued from its end state; the headline stays the K-step window"}

    if hasattr(ro, "drain"):
        ro.drain()  # (cohorts: the plies still outstanding after the timed steps are ended; nothing new is begun)
    opening = None
    if rank == 0 and args.opening_steps > 0 and args.preroll > 0 and dist is None:
        # labelled extra: every game at the start position (what round 1 reported as the headline)
        ids = [drv.new_id() for _ in range(G)]
        ro.start_games(list(range(G)), ids, ids)
        for _ in range(3):
            drv.step()
        torch.cuda.synchronize(device)
        so, po = ro.n_sims, ro.n_plies
        to = time.perf_counter()
        for _ in range(args.opening_steps):
            drv.step()
        torch.cuda.synchronize(device)
        do = time.perf_counter() - to
        if hasattr(ro, "drain"):
            ro.drain()
        opening = {"nodes_per_sec": round((ro.n_sims - so) / do, 1), "ms_per_step": round(do / args.opening_steps * 1e3, 3),
                   "steps": args.opening_steps, "note": "all games within their first ~25 plies; not the headline"}

    fast_roof = rn = sweep = None
    if rank == 0 and args.fast and not args.no_roofline and dist is None:
        if args.select_sweep:
            sweep = fast_select_sweep(ro, drv)
            best = max(sweep, key=lambda r: r["frac"]) if sweep else None
            if best:  # (kernel label = " u<games per half-wave> flags<n>")
                ro.eng.fast_options(games_per_halfwave=int(best["kernel"].split(" u")[1][0]), select_flags=int(best["kernel"].split("flags")[1]))
        fast_roof = fast_select_roofline(ro, drv, args.roofline_steps)
        if fast_roof is not None and args.select_sweep:
            fast_roof["variants"] = sweep
    if rank == 0 and not args.no_roofline and not args.fast and args.net_dtype == "fp32":
        rn = nn_roofline(net, G // args.cohorts, device, timed, dt * 1e6, all_boards=G, launches_per_ply=fwd / args.steps / max(args.cohorts, 1),
                         ply_us=(long_run["ms_per_step"] * 1e3 if long_run else dt / args.steps * 1e6))

    out = None
    if rank == 0:
        mean_len = fin_plies_all / fin_all if fin_all else None
        cfg_name = ("BASELINE.json configs[4] per-GPU shard (f16 net, f32 tree)" if (args.net, args.net_dtype, G, args.sims) == ("20x256", "fp16", 512, 800)
                    else "BASELINE.json configs[1]" if (args.net, args.net_dtype, G, args.sims) == ("10x128", "fp32", 256, 400)
                    else "BASELINE.json configs[2] per-GPU shard" if (args.net, args.net_dtype, G, args.sims) == ("10x128", "fp32", 256, 800)
                    else "custom configuration")
        out = {
            "metric": "mcts_nodes_per_sec", "value": round(sims / dt, 1), "unit": "nodes/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": ("f32 (3 x f16 split, f32 accumulate)" if getattr(getattr(net, "net", net), "conv", None) == "tower_split" else {"fp32": "f32", "fp16": "f16", "bf16": "bf16"}[args.net_dtype]), "data": "synthetic",
            "config": {"workload": f"{G} concurrent self-play games per GPU x {args.sims} sims/move, MCTS_BATCH_SIZE {args.batch}, "
                                   f"net {args.net} ({'+'.join(map(str, NETS[args.net][:2]))} blocks x {NETS[args.net][2]} filters, "
                                   f"random init, {args.net_dtype}, BN folded), games from the start position, per-game seeds = game id, "
                                   + (f"steady state (games at every stage after {args.preroll} untimed pre-roll plies with staggered starts; finished games are "
                                      f"exported and their slots refilled inside the timed region); " if args.preroll > 0 else "opening phase (all games start together); ")
                                   + cfg_name,
                       "games_per_gpu": G, "cohorts": args.cohorts, "cohort_cu_masks": getattr(ro, "cu_masks", "off") if args.cohorts > 1 else None, "sims_per_move": args.sims, "net": args.net, "net_dtype": args.net_dtype,
                       "hipgraph": not args.no_graph, "net_layout": net_layout, "policy_softmax": softmax_site(net, args, G * (args.leaves if args.fast else 1)), "evaluate_stage": getattr(getattr(net, "net", net), "route", net_layout), "hw_queues": betaone_amd.hw_queues(), "preroll_plies": args.preroll,
                       "search_mode": ("fast: virtual loss, %d leaves/step, full-width expansion (NOT the reference's semantics)" % args.leaves)
                                      if args.fast else "reference semantics (bit-exact)",
                       "parallelism": f"games sharded over {world} GPU(s) by id; record all-gather every {args.exchange_every} plies, pipelined" if world > 1
                                      else "1 GPU"},
            "plies_per_sec": round(plies / dt, 2), "nn_forwards_per_sec": round(fwd / dt / world, 2),
            "unique_nn_evals_per_sec": round(fwd * (G // args.cohorts) * (args.leaves if args.fast else 1) / dt, 1),  # (a forward evaluates ONE cohort's boards)
            "arena_full_slots": (int(arena_full) if args.fast else None),
            "games_finished_in_timed_region": int(fin_timed),
            "games_per_hour_measured": (round(fin_timed * 3600.0 / dt, 1) if fin_timed else None),
            "mean_plies_of_finished_games": (round(fin_plies_timed / fin_timed, 1) if fin_timed else None),
            "step_ms_min_p50_p90_max": [round(float(x), 3) for x in (step_ms.min(), np.percentile(step_ms, 50), np.percentile(step_ms, 90), step_ms.max())],
            "step_ms_each": ([round(float(x), 2) for x in step_ms] if (len(step_ms) <= 64 or os.environ.get("BO_ALL_STEPS")) else None),  # (host-side period of every timed step, in order)
            "step_host_wait_ms_each": ([round(w * 1e3, 2) for w in step_wait] if (len(step_wait) <= 64 and hasattr(ro, "wait_seconds")) else None),  # (of which: no cohort was ready -- the host waited for the device)
            "games_finished_since_start": int(fin_all),
            "mean_plies_of_all_finished_games": (round(mean_len, 1) if mean_len else None),
            "games_per_hour_from_ply_rate": (round(plies / dt * 3600.0 / mean_len, 1) if mean_len else None),
            "untimed_setup_seconds": round(t_pre, 2),
            "host_fraction": round(host_frac, 4),
        }
        if long_run:
            out["long_run"] = long_run
        if opening:
            out["opening_phase"] = opening
        if comm is not None:
            out["rccl" if args.dist_backend == "nccl" else "process_group"] = comm
        if exchange is not None:
            out["record_exchange"] = {"size_gathers": exchange.n_size_gathers, "payload_gathers": exchange.n_payload_gathers,
                                      "ticks_that_blocked": exchange.blocked_ticks, "ticks_that_blocked_all_ranks": blocked_all,
                                      "records_received_rank0": drv.n_received,
                                      "host_seconds_in_exchange_rank0": round(drv.exchange_seconds, 4),
                                      "seconds_draining_the_pipeline_after_the_timed_steps_rank0": round(t_flush, 4),
                                      "backend": args.dist_backend,
                                      "hardware_coverage": "the builder's pool hands out one-GPU boxes: RCCL between DISTINCT GPUs has never run before this "
                                                           "command; covered so far: gloo world 2/4/8 on CPU, RCCL at N=1, two gloo ranks sharing one GPU"}
            out["per_rank_ms_per_step"] = {"min": min(per_rank_ms), "max": max(per_rank_ms), "ranks": per_rank_ms}
    if rank == 0 and not args.no_roofline:
        sr = step_roofline(ro, args.steps)
        out["roofline_step"] = {"bound": "latency", "note": "parity-mode trees are cache-resident (a few KB per game)",
                                "select_alg_bytes_total": sr["levels"] * 8 + sr["children_scanned"] * 12, **sr}
    ro.close()
    del ro
    torch.cuda.empty_cache()
    if dump is not None:
        with open(args.dump_games if world == 1 else f"{args.dump_games}.rank{rank}", "w") as f:
            json.dump(dump, f)
    if rank == 0 and not args.no_roofline:
        # `roofline` = the DOMINANT kernel of the timed workload.  Reference semantics: the evaluate stage's tower kernel
        # (~80 % of a ply, MFMA-bound).  Fast mode: its select + backup kernel on the trees this run grew (what SURVEY.md
        # section 8f asks to price).  The north star's select target on the section-8d synthetic wide workload
        # (bo_k_select_wide: the same child-block layout and arithmetic, 262144 static trees) is reported beside it.
        if fast_roof is not None:
            out["roofline"] = fast_roof
        else:
            # LAB kernel, run by no search: PUCT select alone over static synthetic trees (SURVEY.md section 8d's workload); the
            # product's select + backup kernel on grown trees is priced by `python bench.py --fast`
            # (N = 1 only, or when it is the only roofline there is: the other ranks of an N > 1 run wait in the closing barrier meanwhile)
            wide = select_roofline(args, device) if (world == 1 or not rn) else None
            if wide is not None:
                wide["kernel"] = "bo_k_select_wide (lab kernel: select only, static synthetic trees; no search runs it)"
                out["roofline_select_wide_synthetic"] = wide
            out["roofline"] = rn if rn else wide
    if rank == 0 and world == 1 and not args.no_cpu_baseline and not args.fast:  # "on rank 0 at N=1 only"
        out["cpu_baseline"] = cpu_baseline(args)
    elif rank == 0 and args.fast:
        out["cpu_baseline"] = None  # the fast mode is not the reference's algorithm: nothing of the reference to time beside it
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out))


if __name__ == "__main__":
    try:
        main()
    except Exception as ex:
        from betaone_amd.records import ExchangeError

        if not isinstance(ex, ExchangeError):
            raise
        # a peer rank died or hangs mid-period: say so and leave with a non-zero code at once (no waiting in the process group's
        # teardown for the peer; the launcher ends the other ranks)
        print(f"bench.py rank {os.environ.get('RANK', '0')}: {ex}", file=sys.stderr, flush=True)
        os._exit(3)
