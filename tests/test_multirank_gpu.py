"""N > 1 on a one-GPU box: bench.py's own multi-rank path (self-launch through torch.distributed.run, one process per rank,
game ids sharded by id mod world, PeriodicGameExchange on a side stream) with two ranks sharing cuda:0.

More ranks than devices is not a configuration RCCL forms a communicator for, so the collective backend here is gloo (host
staging) and bench.py refuses `--share-gpu --dist-backend nccl` up front with a message (asserted below; nothing is started
that is expected to fail).  What this box cannot show -- RCCL between distinct GPUs -- is stated as unmeasured in DESIGN.md."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(extra, timeout=420):
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--share-gpu", "--games", "32", "--sims", "100", "--steps", "24",
           "--warmup", "2", "--preroll", "16", "--max-game-moves", "12", "--exchange-every", "4", "--no-roofline", "--no-cpu-baseline",
           "--opening-steps", "0"] + extra
    p = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=timeout)
    line = next((ln for ln in reversed(p.stdout.splitlines()) if ln.startswith("{")), None)
    return p, (json.loads(line) if line else None)


def test_two_ranks_share_the_gpu_and_exchange_every_finished_game():
    p, out = _bench(["--dist-backend", "gloo"])
    assert p.returncode == 0 and out is not None, p.stderr[-2000:]
    assert out["n_gpus"] == 2 and out["scaling"] == "weak" and out["value"] > 0
    ex = out["record_exchange"]
    # games end every 12 plies: every period had records, each one delivered to rank 0 exactly once (both ranks' games)
    assert ex["payload_gathers"] >= 2 and ex["records_received_rank0"] >= out["games_finished_since_start"] - 2 * 32
    assert out["games_finished_in_timed_region"] >= 32          # both ranks' finished games are in the all-reduced count
    assert out["config"]["hw_queues"]["in_effect"] is True
    # the communicator census the N > 1 line carries (gloo rehearsal: "process_group"; RCCL: "rccl"): both ranks seen, ONE device between them
    pg = out["process_group"]
    assert pg["backend"] == "gloo" and pg["world_size_seen"] == 2 and pg["ranks_on_distinct_devices"] is False and pg["distinct_devices"] == 1
    assert [d["rank"] for d in pg["devices"]] == [0, 1] and pg["devices"][0]["device_id"] == pg["devices"][1]["device_id"]


def test_rccl_path_at_one_rank_with_graphs_captured_beside_the_watchdog():
    """The RCCL ("nccl") backend at N = 1 through bench.py's distributed path: process group on the GPU, the communicator census
    (all_reduce + all_gather_object), cohorts capturing their hipGraphs AFTERWARDS while the process group's watchdog thread polls its
    collectives' events, the record exchange's all-gathers on the side stream.  With hipGraph captures in the default (global) mode the
    watchdog's hipEventQuery killed the first capture ("operation not permitted when stream is capturing", round 5's rehearsal);
    rollout.CAPTURE_MODE = "thread_local" is what the N > 1 run on distinct GPUs needs on every rank."""
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29541")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--force-dist", "--games", "128", "--cohorts", "2", "--sims", "100", "--steps", "24", "--warmup", "2",
           "--preroll", "16", "--max-game-moves", "12", "--exchange-every", "4", "--no-roofline", "--no-cpu-baseline", "--opening-steps", "0"]
    p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=420)
    line = next((ln for ln in reversed(p.stdout.splitlines()) if ln.startswith("{")), None)
    assert p.returncode == 0 and line is not None, p.stderr[-3000:]
    out = json.loads(line)
    assert out["rccl"]["backend"] == "nccl" and out["rccl"]["world_size_seen"] == 1 and out["rccl"]["ranks_on_distinct_devices"] is True
    assert out["config"]["hipgraph"] is True and out["record_exchange"]["payload_gathers"] >= 2 and out["games_finished_in_timed_region"] >= 64


def test_share_gpu_with_the_rccl_backend_is_refused_before_any_rank_starts():
    """no run that is expected to fail on the GPU box.  The refusal comes from bench.py's argument check (exit
    code 2, a message naming the gloo rehearsal), before a rank is started or the GPU is touched."""
    p, out = _bench(["--dist-backend", "nccl"], timeout=120)
    assert p.returncode == 2 and out is None
    assert "--dist-backend gloo" in p.stderr and "two ranks on one device" in p.stderr
    assert "ChildFailedError" not in p.stderr and "torch.distributed" not in p.stderr
