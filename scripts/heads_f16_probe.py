#!/usr/bin/env python3
"""scripts/heads_f16_probe.py -- LAB: bo_nn_heads_f16 against the library path it replaces (fp16 GEMM + widening copy + softmax, value
GEMM + ReLU + GEMM + tanh) at fast mode's row counts.  usage: heads_f16_probe.py [rows ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from betaone_amd import engine as E

lib = E.load_hip_library()
g = torch.Generator().manual_seed(1)
wp = (torch.randn((4672, 128), generator=g) / 11.0).cuda().half(); bp = torch.randn(4672, generator=g).cuda()
w1 = (torch.randn((256, 2048), generator=g) / 45.0).cuda().half(); b1 = torch.randn(256, generator=g).cuda()
w2 = (torch.randn((1, 256), generator=g) / 16.0).cuda(); b2 = torch.randn(1, generator=g).cuda()
st = torch.cuda.current_stream().cuda_stream


def timed(fn, n=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


for B in [int(x) for x in sys.argv[1:]] or [4096, 16384, 65536, 131072]:
    p = torch.rand((B, 128), generator=g).cuda().half(); v = torch.rand((B, 2048), generator=g).cuda().half()
    out = torch.empty((B, 4672), device="cuda"); val = torch.empty((B, 1), device="cuda"); scr = torch.empty(20 * B, device="cuda")
    ours = timed(lambda: lib.bo_nn_heads_f16(p.data_ptr(), v.data_ptr(), wp.data_ptr(), bp.data_ptr(), w1.data_ptr(), b1.data_ptr(), w2.data_ptr(),
                                             b2.data_ptr(), out.data_ptr(), val.data_ptr(), scr.data_ptr(), B, 1, st))
    bph, b1h, w2h, b2h = bp.half(), b1.half(), w2.half(), b2.half()
    libt = timed(lambda: (torch.softmax(F.linear(p, wp, bph).float(), dim=1), torch.tanh(F.linear(F.relu(F.linear(v, w1, b1h)), w2h, b2h))))
    pol = timed(lambda: lib.bo_nn_heads_f16(p.data_ptr(), v.data_ptr(), wp.data_ptr(), bp.data_ptr(), w1.data_ptr(), b1.data_ptr(), w2.data_ptr(),
                                            b2.data_ptr(), out.data_ptr(), val.data_ptr(), scr.data_ptr(), min(B, 32), 1, st))  # (launch floor of the pair)
    gb = B * 4672 * 4 / 1e9
    print(f"{B:7d} rows: bo_nn_heads_f16 {ours:8.1f} us ({gb / ours * 1e6:6.0f} GB/s of probabilities written)   library path {libt:8.1f} us   pair at 32 rows {pol:6.1f} us")
