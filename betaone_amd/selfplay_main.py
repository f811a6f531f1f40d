"""
betaone_amd/selfplay_main.py -- self-play orchestration for one training iteration.

Replaces the self-play part of /root/reference/main.py:142-191 (an mp.Pool of 6 one-game workers that each rebuild
the model on GPU 0 and exchange results through pickle files) with one process per GPU driving `--slots` resident
games.  Keeps the reference's contracts:
  * weights come from SAVE_DIR/best_model.pth (main.py:147-148; written by the training side);
  * results go to DATA_DIR/iter_{i}/game_{j}.pkl in save_game_data's format (self_play.py:220-231), which
    train.load_recent_data reads unchanged (train.py:187-219);
  * resume = skip game ids whose file already exists (main.py:26-36 check_existing_self_play_data).
Multi-GPU: launch with `python -m torch.distributed.run --nproc-per-node N -m betaone_amd.selfplay_main ...`;
game id j runs on rank j mod N (its RNG seed is derived from (iteration, j), so results do not depend on N).

    python -m betaone_amd.selfplay_main --iteration 3 --games 1000 --slots 256
"""
from __future__ import annotations

import argparse
import os
import time
from typing import Callable, Dict, List, Optional


def game_seed(iteration: int, game_id: int) -> int:
    return (iteration * 1_000_003 + game_id * 7919 + 12345) & 0xFFFFFFFF


def pending_game_ids(data_dir: str, iteration: int, n_games: int) -> List[int]:
    d = os.path.join(data_dir, f"iter_{iteration}")
    return [j for j in range(n_games) if not os.path.exists(os.path.join(d, f"game_{j}.pkl"))]


def run_iteration(model, iteration: int, n_games: int, n_slots: int, rank: int = 0, world: int = 1,
                  log: Callable[[str], None] = print) -> Dict[int, int]:
    """Play the games of `iteration` that are not on disk yet and save each as the reference's pickle.
    Returns {game_id: plies}."""
    from betaone_amd import dropin

    dropin.install()
    import config
    import self_play

    todo = [j for j in pending_game_ids(config.DATA_DIR, iteration, n_games) if j % world == rank]
    if not todo:
        log(f"[rank {rank}] iteration {iteration}: nothing to do")
        return {}
    t0 = time.time()
    results = self_play.run_self_play_games(model, todo, seeds=[game_seed(iteration, j) for j in todo],
                                            n_slots=min(n_slots, len(todo)))
    done = {}
    for j, data in results.items():
        if data:
            self_play.save_game_data(data, iteration, j)
            done[j] = len(data)
    dt = time.time() - t0
    plies = sum(done.values())
    log(f"[rank {rank}] iteration {iteration}: {len(done)} games, {plies} plies in {dt:.1f} s "
        f"({plies * config.NUM_SIMULATIONS / max(dt, 1e-9):.0f} nodes/s)")
    return done


def main(argv: Optional[List[str]] = None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--iteration", type=int, required=True)
    ap.add_argument("--games", type=int, default=None, help="games in this iteration (default config.GAMES_MINIMUM)")
    ap.add_argument("--slots", type=int, default=256, help="concurrent games per GPU")
    ap.add_argument("--model", default=None, help="state_dict file (default SAVE_DIR/best_model.pth)")
    args = ap.parse_args(argv)
    import torch

    from betaone_amd import dropin

    dropin.install()
    import config
    import network

    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if torch.cuda.is_available():
        torch.cuda.set_device(local)
    model = network.PolicyValueNet().to(config.DEVICE)
    path = args.model or os.path.join(config.SAVE_DIR, "best_model.pth")
    if os.path.exists(path):
        model.load_state_dict(torch.load(path, map_location=config.DEVICE))
    else:
        print(f"[rank {rank}] warning: {path} not found, playing with a randomly initialised net")
    model.eval()
    run_iteration(model, args.iteration, args.games or config.GAMES_MINIMUM, args.slots, rank, world)


if __name__ == "__main__":
    main()
