"""
betaone_amd/selfplay_main.py -- self-play orchestration for one training iteration.

Replaces the self-play part of /root/reference/main.py:142-191 (an mp.Pool of 6 one-game workers that each rebuild
the model on GPU 0 and exchange results through pickle files) with one process per GPU driving `--slots` resident
games.  Keeps the reference's contracts:
  * weights come from SAVE_DIR/best_model.pth (main.py:147-148; written by the training side);
  * results go to DATA_DIR/iter_{i}/game_{j}.pkl in save_game_data's format (self_play.py:220-231), which
    train.load_recent_data reads unchanged (train.py:187-219);
  * resume = skip game ids whose file already exists (main.py:26-36 check_existing_self_play_data).
Multi-GPU, one command like the reference's (main.py:166-175 starts its N workers itself): `--gpus N` starts N ranks,
one process per GPU (a `torch.distributed.run` child, started before this process touches a GPU); game id j runs on
rank j mod N (its RNG seed is derived from (iteration, j), so results do not depend on N).  Under an external
torchrun the RANK / LOCAL_RANK / WORLD_SIZE variables are honoured instead.

    python -m betaone_amd.selfplay_main --iteration 3 --games 1000 --slots 256 --gpus 8
"""
from __future__ import annotations

import argparse
import os
import subprocess
import sys
import time
from typing import Callable, Dict, List, Optional


def game_seed(iteration: int, game_id: int) -> int:
    return (iteration * 1_000_003 + game_id * 7919 + 12345) & 0xFFFFFFFF


def pending_game_ids(data_dir: str, iteration: int, n_games: int, records: str = "pickle") -> List[int]:
    """Game ids of the iteration whose result is not on disk yet (main.py:26-36) in the form(s) `records` asks for: "pickle" =
    no game_{j}.pkl; "compact" = no record in any games_rank*.bog; "both" = missing in EITHER form (a run killed between a
    game's two writes plays that game again and writes only the form that is missing)."""
    d = os.path.join(data_dir, f"iter_{iteration}")
    compact = set()
    if records in ("compact", "both"):
        from betaone_amd import records as R

        compact = R.game_ids_on_disk(data_dir, iteration)

    def pickled(j):
        return os.path.exists(os.path.join(d, f"game_{j}.pkl"))

    if records == "compact":
        return [j for j in range(n_games) if j not in compact]
    if records == "both":
        return [j for j in range(n_games) if not (j in compact and pickled(j))]
    return [j for j in range(n_games) if not pickled(j)]


class ModelFileWatcher:
    """Weights hot-swap inside a living process (main.py:147-148 saves best_model.pth at the top of every iteration and each of
    its one-game workers loads it, main.py:47-49): poll() returns a freshly loaded PolicyValueNet when the file's (mtime, size)
    changed since the last look, else None.  A file caught in the middle of being written fails to load and is retried at the
    next poll."""

    def __init__(self, path: str, make_model: Callable[[], object], device, every: int = 8):
        self.path, self.make_model, self.device, self.every = path, make_model, device, max(1, int(every))
        self._calls, self.n_reloads = 0, 0
        self._seen = self._stamp()

    def _stamp(self):
        try:
            st = os.stat(self.path)
            return (st.st_mtime_ns, st.st_size)
        except OSError:
            return None

    def poll(self):
        self._calls += 1
        if self._calls % self.every:
            return None
        now = self._stamp()
        if now is None or now == self._seen:
            return None
        import torch

        try:
            sd = torch.load(self.path, map_location=self.device)
            model = self.make_model().to(self.device)
            model.load_state_dict(sd)
        except Exception as ex:  # half-written file: try again next time
            print(f"[selfplay] {self.path} changed but could not be loaded yet ({type(ex).__name__})")
            return None
        self._seen = now
        self.n_reloads += 1
        return model.eval()


def run_iteration(model, iteration: int, n_games: int, n_slots: int, rank: int = 0, world: int = 1,
                  log: Callable[[str], None] = print, records: str = "pickle", reload_model=None) -> Dict[int, int]:
    """Play the games of `iteration` that are not on disk yet.  records: "pickle" = the reference's one pickle per game
    (self_play.py:220-231), "compact" = ~100 B/ply records appended to DATA_DIR/iter_{i}/games_rank{rank}.bog
    (betaone_amd.records, read back by records.CompactDataset with ChessDataset's item contract), "both".
    Returns {game_id: plies}."""
    from betaone_amd import dropin

    dropin.install()
    import config
    import self_play
    from betaone_amd import records as R

    assert records in ("pickle", "compact", "both")
    todo = [j for j in pending_game_ids(config.DATA_DIR, iteration, n_games, records) if j % world == rank]
    if not todo:
        log(f"[rank {rank}] iteration {iteration}: nothing to do")
        return {}
    t0 = time.time()
    done: Dict[int, int] = {}
    path = R.compact_path(config.DATA_DIR, iteration, rank)

    have_compact = R.game_ids_on_disk(config.DATA_DIR, iteration) if records == "both" else set()

    def on_game(fin):  # one append per finished game: a killed run keeps every game it finished
        done[fin.game_id] = len(fin.pis)
        if records != "pickle" and len(fin.pis) and fin.game_id not in have_compact:
            R.save_games(path, [fin])

    def on_records(game_id, data):  # the reference's pickle, written the moment the game's dense tuples exist (not after the
        if data:                    # whole iteration: a killed run keeps these too, and the tuples need not stay in memory)
            self_play.save_game_data(data, iteration, game_id)
        return []

    results = self_play.run_self_play_games(model, todo, seeds=[game_seed(iteration, j) for j in todo],
                                            n_slots=min(n_slots, len(todo)), on_game=on_game, dense=records != "compact",
                                            reload_model=reload_model, on_records=on_records if records != "compact" else None)
    for j, data in results.items():
        if data is None:
            done.pop(j, None)  # aborted game (self_play.py:167): no record
    done = {j: n for j, n in done.items() if n}
    dt = time.time() - t0
    plies = sum(done.values())
    log(f"[rank {rank}] iteration {iteration}: {len(done)} games, {plies} plies in {dt:.1f} s "
        f"({plies * config.NUM_SIMULATIONS / max(dt, 1e-9):.0f} nodes/s)")
    return done


def launch_ranks(n: int, child_args: List[str], env: Optional[dict] = None) -> int:
    """Start `n` ranks of `python <child_args>` on this node through torch.distributed.run (one process per GPU, rendezvous on
    127.0.0.1) as a CHILD process and return its exit code.  Call before the parent has touched a GPU."""
    import socket

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port)] + child_args
    e = dict(os.environ if env is None else env)
    e.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    pkg_parent = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    e["PYTHONPATH"] = pkg_parent + (os.pathsep + e["PYTHONPATH"] if e.get("PYTHONPATH") else "")
    return subprocess.call(cmd, env=e)


def main(argv: Optional[List[str]] = None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--iteration", type=int, required=True)
    ap.add_argument("--games", type=int, default=None, help="games in this iteration (default config.GAMES_MINIMUM)")
    ap.add_argument("--slots", type=int, default=256, help="concurrent games per GPU")
    ap.add_argument("--model", default=None, help="state_dict file (default SAVE_DIR/best_model.pth)")
    ap.add_argument("--gpus", type=int, default=1, help="ranks to start on this node (ignored under torchrun)")
    ap.add_argument("--records", default="pickle", choices=["pickle", "compact", "both"],
                    help="pickle: the reference's game_{id}.pkl; compact: games_rank{r}.bog (~100 B/ply, records.CompactDataset); both")
    ap.add_argument("--watch-model", action="store_true", help="reload the state_dict file whenever it changes on disk (checked every 8 plies)")
    args = ap.parse_args(argv)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args.gpus, ["-m", "betaone_amd.selfplay_main"] + list(sys.argv[1:] if argv is None else argv)))
    import torch

    from betaone_amd import dropin

    dropin.install()
    import config
    import network

    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if torch.cuda.is_available():
        torch.cuda.set_device(local)
        config.DEVICE = f"cuda:{local}"  # explicit: engine, NN rows and model of this rank on its own GPU
    model = network.PolicyValueNet().to(config.DEVICE)
    path = args.model or os.path.join(config.SAVE_DIR, "best_model.pth")
    if os.path.exists(path):
        model.load_state_dict(torch.load(path, map_location=config.DEVICE))
    else:
        print(f"[rank {rank}] warning: {path} not found, playing with a randomly initialised net")
    model.eval()
    watcher = ModelFileWatcher(path, network.PolicyValueNet, config.DEVICE) if args.watch_model else None
    run_iteration(model, args.iteration, args.games or config.GAMES_MINIMUM, args.slots, rank, world, records=args.records,
                  reload_model=watcher.poll if watcher else None)
    if watcher:
        print(f"[rank {rank}] weights reloaded {watcher.n_reloads} time(s) from {path}")


if __name__ == "__main__":
    main()
