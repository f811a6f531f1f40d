"""CPU tests of the N>1 path: games shard by id over ranks (no data-path collective), finished games
travel as compact records through ONE all-gather (gloo here, RCCL on the GPUs).  world_size = 2, 4 and 8."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir, every=0):
    sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from betaone_amd import records
    from betaone_amd.rollout import Rollout
    from engine_harness import emu_call
    from fake_model import FakeNet

    class Net(torch.nn.Module):
        def forward(self, x):
            return FakeNet(scale=0.0, salt=21)(x)

    n_total = 3 * world
    ids = records.shard_game_ids(n_total, rank, world)           # game id -> rank = id mod world
    assert ids == list(range(rank, n_total, world))
    # uneven finish times: rank r's games stop after 3 + r moves, so in most steps only some ranks (or none) have records
    ro = emu_call(Rollout, Net(), len(ids), num_simulations=30, mcts_batch_size=16, max_game_moves=3 + rank, device="cpu",
                  use_graph=False)
    ro.start_games(list(range(len(ids))), ids, [np.random.RandomState(i) for i in ids])
    fins = []
    gathered = []
    ex = records.PeriodicGameExchange(every=every) if every else None   # bench.py's form: pipelined, every `every` plies
    n_steps = 5 + world
    for _step in range(n_steps):   # every rank makes the SAME number of exchange steps (as bench.py does: one per step)
        batch = []
        if any(g is not None for g in ro.games):
            ro.play_ply(on_finished=batch.append)
        fins.extend(batch)
        gathered.extend(ex.push(batch) if every else records.all_gather_games(batch))   # the path's only exchange step
    if every:
        gathered.extend(ex.flush())
        ticks = n_steps // every
        assert ex.n_size_gathers == ticks + 1                     # one 8-byte all-gather per period (+ the flush's)
        assert ex.n_payload_gathers <= ticks + 1 and ex.n_payload_gathers >= 1
        if every > n_steps:  # (8, 16): the node's shape with the bench's default period -- everything travels in the final drain
            assert ex.n_size_gathers == 1 and ex.n_payload_gathers == 1
        if every == 1:  # periods in which NO rank finished a game took no payload step
            assert ex.n_payload_gathers <= world
    assert not any(g is not None for g in ro.games)
    ro.close()
    mine = {g.game_id: [m for m in g.moves] for g in fins}
    assert len(gathered) == n_total and len({g["game_id"] for g in gathered}) == n_total   # every record exactly once
    np.save(os.path.join(out_dir, f"rank{rank}.npy"),
            np.array([[g["game_id"], g["n_plies"], g["terminal"]] + list(g["moves"][:3]) for g in
                      sorted(gathered, key=lambda x: x["game_id"])], dtype=np.int64))
    for g in gathered:
        if g["game_id"] in mine:
            assert list(g["moves"]) == mine[g["game_id"]]
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,every", [(2, 0), (2, 1), (2, 3), (4, 2), (8, 1), (8, 16)])
def test_ranks_shard_games_and_all_gather_records(tmp_path, world, every):
    port = _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path), every), nprocs=world, join=True)
    a = np.load(tmp_path / "rank0.npy")
    for r in range(1, world):
        assert np.array_equal(a, np.load(tmp_path / f"rank{r}.npy"))   # every rank ends with every game's record
    assert sorted(a[:, 0].tolist()) == list(range(3 * world))
    for r in range(world):
        assert all(a[a[:, 0] % world == r][:, 1] == 3 + r)


def test_results_do_not_depend_on_the_sharding():
    """game id + seed travel together: the same game on a different rank/slot count gives the same record."""
    sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
    from betaone_amd.rollout import Rollout
    from engine_harness import emu_call
    from fake_model import FakeNet

    class Net(torch.nn.Module):
        def forward(self, x):
            return FakeNet(scale=0.0, salt=21)(x)

    def play(ids):
        ro = emu_call(Rollout, Net(), len(ids), num_simulations=30, mcts_batch_size=16, max_game_moves=4, device="cpu",
                      use_graph=False)
        ro.start_games(list(range(len(ids))), ids, [np.random.RandomState(i) for i in ids])
        fins = []
        while any(g is not None for g in ro.games):
            ro.play_ply(on_finished=fins.append)
        ro.close()
        return {f.game_id: f.moves for f in fins}

    one = play([0, 1, 2, 3])
    two = {**play([0, 2]), **play([1, 3])}
    assert one == two


def test_record_wire_format_roundtrip():
    sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
    from betaone_amd import engine as E, records
    from betaone_amd.rollout import FinishedGame

    pos = [E.BoPosition() for _ in range(3)]
    for i, p in enumerate(pos):
        p.bb[0] = 0xFF00 + i
        p.turn = (i + 1) % 2
        p.ep_square, p.ep_key, p.fullmove_number = -1, -1, 1 + i
    fin = FinishedGame(game_id=7, slot=0, moves=[796, 3364], positions=pos,
                       pis=[(np.array([5, 900], np.int32), np.array([0.25, 0.75], np.float32)),
                            (np.array([33], np.int32), np.array([1.0], np.float32))], outcome=1.0, terminal=1)
    blob = records.pack_game(fin) + records.pack_game(fin)
    games = records.unpack_games(blob)
    assert len(games) == 2 and games[1]["game_id"] == 7 and games[0]["outcome"] == 1.0
    assert list(games[0]["moves"]) == [796, 3364]
    assert games[0]["positions"][2].bb[0] == 0xFF02 and games[0]["positions"][1].turn == 0
    assert games[0]["pis"][0][0].tolist() == [5, 900] and games[0]["pis"][0][1].tolist() == [0.25, 0.75]
    assert len(blob) // 2 < 400     # ~100 B per ply instead of 49 KB per ply


def test_launch_ranks_starts_one_process_per_rank(tmp_path):
    """`bench.py --gpus N` / `selfplay_main --gpus N` start their ranks through selfplay_main.launch_ranks (a
    torch.distributed.run child on 127.0.0.1): every rank sees RANK / LOCAL_RANK / WORLD_SIZE and can rendezvous."""
    sys.path[:0] = [ROOT]
    from betaone_amd.selfplay_main import launch_ranks

    script = tmp_path / "child.py"
    script.write_text(
        "import os, sys\n"
        "import torch.distributed as dist\n"
        "dist.init_process_group('gloo')\n"
        "r, w = dist.get_rank(), dist.get_world_size()\n"
        "assert r == int(os.environ['RANK']) == int(os.environ['LOCAL_RANK']) and w == int(os.environ['WORLD_SIZE']) == 3\n"
        "import torch\n"
        "t = torch.tensor([r + 1.0]); dist.all_reduce(t)\n"
        "open(os.path.join(sys.argv[1], f'rank{r}.txt'), 'w').write(str(t.item()))\n"
        "dist.destroy_process_group()\n")
    rc = launch_ranks(3, [str(script), str(tmp_path)])
    assert rc == 0
    assert [open(tmp_path / f"rank{r}.txt").read() for r in range(3)] == ["6.0"] * 3
    # a failing rank is reported through the exit code
    bad = tmp_path / "bad.py"
    bad.write_text("import sys; sys.exit(3)\n")
    assert launch_ranks(2, [str(bad)]) != 0


_DYING_RANK = r"""
import os, sys, time
sys.path[:0] = [{root!r}]
import numpy as np
import torch.distributed as dist
rank = int(os.environ["RANK"])
dist.init_process_group("gloo", rank=rank, world_size=2)
from betaone_amd import records
ex = records.PeriodicGameExchange(every=1, timeout_s=float(sys.argv[1]))
ex.push([])                       # tick 0: both ranks start the 8-byte size gather
if rank == 1:
    if sys.argv[2] == "die":
        os._exit(7)               # dies mid-period, without a word to its peers
    time.sleep(600)               # ... or hangs
t0 = time.monotonic()
try:
    for _ in range(4):
        ex.push([])               # rank 0 carries on: its next collective has no partner
        time.sleep(0.05)
except records.ExchangeError as e:
    print("ExchangeError after %.1f s: %s" % (time.monotonic() - t0, e), flush=True)
    os._exit(3)                   # loud: a non-zero exit, no lingering in destroy_process_group
os._exit(0)
"""


@pytest.mark.parametrize("how", ["die", "hang"])
def test_a_rank_that_dies_or_hangs_mid_period_makes_its_peer_fail_loudly_not_wait(tmp_path, how):
    """PeriodicGameExchange must not hang when a rank dies (or stops answering) inside an exchange
    period.  Rank 1 goes away after the first size gather; rank 0's next ticks find their collective without a partner and
    raise records.ExchangeError within the timeout (gloo reports the closed connection at once; a hanging peer runs into
    `timeout_s`), and the process exits non-zero."""
    import subprocess
    import time

    script = tmp_path / "rank.py"
    script.write_text(_DYING_RANK.format(root=ROOT))
    port = _free_port()
    procs = []
    for r in range(2):
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(r), WORLD_SIZE="2")
        procs.append(subprocess.Popen([sys.executable, str(script), "4", how], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    t0 = time.monotonic()
    try:
        out0, _ = procs[0].communicate(timeout=90)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    assert procs[0].returncode == 3, out0[-2000:]
    assert "ExchangeError" in out0 and time.monotonic() - t0 < 60
