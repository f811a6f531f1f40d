#!/usr/bin/env python3
"""scripts/select_sweep.py -- achieved algorithmic GB/s of bo_k_select_wide vs. trees per launch and grid size."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import select_wide_lab as SW

dev = "cuda:0"
flush = torch.empty(512 << 20, dtype=torch.uint8, device=dev)
for n_trees in [int(x) for x in (sys.argv[1:] or ["32768", "131072"])]:
    w = SW.build(n_trees, 800, seed=0, device=dev)
    for grid in (0, 4096, 8192, 12288, 16384, 32768):
        out = SW.run(w, grid_blocks=grid)
        torch.cuda.synchronize()
        levels = int(out[1].sum().item())
        ms = []
        for _ in range(8):
            flush.fill_(1)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); SW.run(w, grid_blocks=grid, out=out); e1.record(); e1.synchronize()
            ms.append(e0.elapsed_time(e1))
        t = np.median(ms) * 1e-3
        print(f"trees={n_trees} grid={(grid & 0xFFFFF) or 'auto'} variant={grid >> 20} levels={levels} ({levels/n_trees:.2f}/tree) "
              f"t={t*1e6:.1f}us alg={levels*SW.LEVEL_BYTES/t/1e9:.0f} GB/s actual512={levels*512/t/1e9:.0f} GB/s", flush=True)
    del w
    torch.cuda.empty_cache()
