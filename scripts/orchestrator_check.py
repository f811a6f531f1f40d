import os, sys, tempfile, pickle, glob
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from betaone_amd import dropin
dropin.install()
import config
tmp = tempfile.mkdtemp()
config.DATA_DIR = os.path.join(tmp, "data"); config.SAVE_DIR = os.path.join(tmp, "save")
config.NUM_SIMULATIONS, config.MCTS_BATCH_SIZE, config.MAX_GAME_MOVES = 64, 16, 12
config.RESIDUAL_BLOCKS, config.SE_RESIDUAL_BLOCKS, config.CONV_FILTERS = 3, 1, 64
import torch, network
os.makedirs(config.SAVE_DIR, exist_ok=True)
torch.manual_seed(0)
torch.save(network.PolicyValueNet().state_dict(), os.path.join(config.SAVE_DIR, "best_model.pth"))
sys.argv = ["selfplay_main", "--iteration", "1", "--games", "24", "--slots", "8"]
from betaone_amd import selfplay_main
selfplay_main.main()
files = sorted(glob.glob(os.path.join(config.DATA_DIR, "iter_1", "game_*.pkl")))
print("files", len(files))
d = pickle.load(open(files[0], "rb"))
print(type(d), len(d), d[0][0].shape, d[0][1].shape, d[0][2])
