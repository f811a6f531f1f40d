"""CPU test: the product library loads and exports every symbol include/betaone_engine.h declares
(no compute calls -- there is no GPU here)."""
import ctypes as C
import os
import re

from betaone_amd import build, engine as E

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols(header="betaone_engine.h"):
    src = open(os.path.join(ROOT, "include", header)).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(bo_[a-z_0-9]+)\s*\(", src)))


def test_header_and_binding_agree():
    assert declared_symbols() == sorted(E._SYMBOLS)
    assert declared_symbols("betaone_lab.h") == sorted(E._LAB_SYMBOLS)
    # the boundary does not lean on the lab header: no lab or debug entry point is declared in it, and it includes nothing of ours
    engine_h = open(os.path.join(ROOT, "include", "betaone_engine.h")).read()
    assert not re.search(r"bo_debug_|bo_select_wide|bo_event_pair|_profile\s*\(|betaone_lab", re.sub(r"/\*.*?\*/", "", engine_h, flags=re.S))


def test_hip_library_builds_loads_and_exports_the_abi():
    lib_path = build.build()
    lib = C.CDLL(lib_path)
    for name in declared_symbols() + declared_symbols("betaone_lab.h"):
        assert hasattr(lib, name), name
    E.bind(lib)
    header = open(os.path.join(ROOT, "include", "betaone_engine.h")).read()
    assert lib.bo_abi_version() == E.ABI_VERSION == int(re.search(r"#define BO_ABI_VERSION (\d+)", header).group(1))
    lab_header = open(os.path.join(ROOT, "include", "betaone_lab.h")).read()
    assert E.PROF_SLOTS == int(re.search(r"#define BO_PROF_SLOTS (\d+)", lab_header).group(1))


def test_product_loader_has_no_fallback(monkeypatch, tmp_path):
    import pytest

    monkeypatch.setattr(E, "HIP_LIB_PATH", str(tmp_path / "missing.so"))
    monkeypatch.setattr(E, "_hip_lib", None)
    with pytest.raises(E.EngineError):
        E.load_hip_library()


def test_tower_descriptors_are_validated_before_any_device_call():
    """bo_nn_tower_create / bo_nn_conv3x3 / bo_nn_tower_forward reject malformed arguments on the host (no GPU here:
    a descriptor that passed would go on to hipMalloc and fail differently)."""
    import numpy as np

    lib = E.bind(C.CDLL(build.build()))
    lib.bo_last_error.restype = C.c_char_p
    c, per_direct, per_wg = 128, 9 * 16 * 128 * 2 * 4, 32 * 8 * 4 * 64 * 4  # floats per 128->128 layer
    w = np.zeros(3 * per_wg, np.float32)
    p = np.zeros(4096, np.float32)
    out = C.c_void_p()

    def create(rows, algo, channels=c, head=None, n_w=None):
        t = np.ascontiguousarray(np.array(rows, dtype=np.int32))
        return lib.bo_nn_tower_create(t.ctypes.data, len(rows), w.ctypes.data, w.size if n_w is None else n_w, p.ctypes.data, p.size, channels, algo,
                                      head.ctypes.data if head is not None else None, 0, C.byref(out))

    good_direct = [[0, 16, 0, 0, 0, 0, 0, 0], [per_direct // 4, 16, 128, 1, 0, 0, 0, 0], [2 * per_direct // 4, 16, 256, 2, 0, 0, 0, 1]]
    bad = [
        (lambda: create(good_direct, 0, channels=96), E.BO_E_CONFIG if hasattr(E, "BO_E_CONFIG") else -3),          # unsupported filter count
        (lambda: create(good_direct, 7), -1),                                                                        # unknown algo
        (lambda: create([[0, 15, 0, 0, 0, 0, 0, 0]] + good_direct[1:], 0), -1),                                       # wrong K-step count
        (lambda: create([good_direct[1]] + good_direct[1:], 0), -1),                                                  # first layer must be the input conv
        (lambda: create(good_direct[:2], 0), -1),                                                                     # a first conv without its second conv
        (lambda: create(good_direct, 0, n_w=per_direct), -1),                                                         # weights offset beyond the buffer
        (lambda: create(good_direct[:2] + [[2 * per_direct // 4, 16, 4000, 2, 0, 0, 0, 1]], 0), -1),                  # bias offset beyond params
        (lambda: create(good_direct[:2] + [[2 * per_direct // 4, 16, 256, 3, 0, 0, 17, 1]], 0), -3),                  # SE hidden width > 16
        (lambda: create(good_direct[:2] + [[2 * per_direct // 4, 16, 256, 2, 0, 0, 0, 0]], 0), -1),                   # nobody stores the output
        (lambda: create(good_direct, 0, head=np.array([34, 2, 0, 0], np.int32)), -3),                                 # fused head needs Winograd / fp16
        (lambda: create(good_direct, 2, channels=256), -1),                                                           # fp16 tower without head
    ]
    for i, (call, code) in enumerate(bad):
        rc = call()
        assert rc == code and lib.bo_last_error(), (i, rc, code, lib.bo_last_error())
    assert lib.bo_nn_conv3x3(1, 1, 1, None, 1, 4, 24, 24, 0, None) == -3      # no instantiation for 24 filters
    assert lib.bo_nn_conv3x3(1, 1, 1, None, 1, 4, 128, 128, 2, None) == -1    # residual epilogue without a residual
    assert lib.bo_nn_tower_forward(None, 1, 1, None, None, 4, None) == -1
    assert lib.bo_nn_value_tail(None, 1, 1, 1, 4, 8, None) == -1


def test_evaluate_stage_route_is_a_shape_rule():
    """nn_tune.kernel_route: which hand-written evaluate stage a (filters, batch, dtype) runs on; None = library path (warned)."""
    import torch
    from betaone_amd.nn_tune import kernel_route as R

    assert R(128, 256, torch.float32) == R(128, 100000, torch.float32) == R(256, 17, torch.float32) == "tower_split"
    assert R(128, 256, torch.float32, f32_pipe=True) == "tower_wg" and R(64, 17, torch.float32) == "tower_wg"
    assert R(256, 1, torch.float32) == R(256, 4, torch.float32) == R(128, 8, torch.float32) == R(64, 16, torch.float32) == "tower_b1"  # resident grids
    assert R(256, 5, torch.float32) == R(256, 16, torch.float32) == R(128, 9, torch.float32) == "mfma_small"
    assert R(256, 17, torch.float32, f32_pipe=True) == "mfma"
    assert R(128, 4096, torch.float16) == R(256, 512, torch.float16) == R(256, 1, torch.float16) == "tower_f16"
    assert R(64, 4096, torch.float16) is None and R(96, 32, torch.float32) is None and R(128, 256, torch.bfloat16) is None


def test_lds_chunk_swizzle_of_the_fp16_pipe_towers_is_conflict_free():
    """csrc/bo_tower_h.h bo_sw (restated here): chunk j of a cell of the channels-last LDS image is stored at position
    j ^ ((cell % 10 + 8 * (cell // 10)) & 15).  A ds_read_b128 is served per group of 16 lanes
    ({0-3,12-15,20-27}, {4-11,16-19,28-31} of each wave half); for every tap, both position halves, both channel counts and
    every chunk the 16 cells of a group must fall on 16 different bank quads -- and the linear pitch of C + 8 halves the
    kernels used before put three cells of a group on one quad."""
    import re
    src = open(os.path.join(ROOT, "betaone_amd", "csrc", "bo_tower_h.h")).read()
    assert re.search(r"bo_sw\(int cell\) \{ return \(\(cell % 10\) \+ 8 \* \(cell / 10\)\) & 15; \}", src), "restatement out of date"
    groups = [[0, 1, 2, 3, 12, 13, 14, 15] + list(range(20, 28)), list(range(4, 12)) + [16, 17, 18, 19, 28, 29, 30, 31]]
    sw = lambda cell: ((cell % 10) + 8 * (cell // 10)) & 15
    cell0 = lambda n: ((n >> 3) + 1) * 10 + (n & 7) + 1
    for C in (128, 256):
        for t in (0, 1):
            for tap in range(9):
                off = (tap // 3 - 1) * 10 + (tap % 3 - 1)
                for g in groups:
                    for j in range(C // 8):
                        quads = set()
                        for n in g:
                            cell = cell0(n) + 40 * t + off
                            quads.add((cell * (C // 8) + ((j & ~15) | ((j ^ sw(cell)) & 15))) % 16)
                        assert len(quads) == 16, (C, t, tap, j)
        old = [(cell0(n) * ((C + 8) // 8)) % 16 for n in groups[0]]
        assert max(old.count(q) for q in old) == 3


def test_split_precision_weight_packing_reconstructs_float32_and_follows_the_documented_layout():
    """fused_net.pack_conv_weight_split (BO_TOWER_SPLIT_F16, include/betaone_engine.h): element (step, mt, hl, lane, i) of a layer is the
    hi / lo fp16 half of s * W[32*mt + (lane & 31)][16*(step % (c_in/16)) + 8*(lane >> 5) + i][tap = step / (c_in/16)]; hi + lo gives
    s * W back to 2^-22 relative (the lo half of any weight above 2^-18 of the largest is a normal fp16 number), s is a power of two
    that puts the largest |s * W| in [2^14, 2^15)."""
    import numpy as np
    import torch
    from betaone_amd.fused_net import pack_conv_weight_split, split_scale, split_f16

    torch.manual_seed(3)
    co, ci = 64, 32
    w = torch.randn(co, ci, 3, 3) * 0.05
    w[5, 7, 1, 2] = 0.0
    s = split_scale(w)
    assert s == 2.0 ** round(np.log2(s)) and 2.0 ** 14 <= float(w.abs().max()) * s < 2.0 ** 15
    p = pack_conv_weight_split(w, s)
    assert p.dtype == torch.float16 and tuple(p.shape) == (9, ci // 16, co // 32, 2, 2, 32, 8)
    flat = p.reshape(9 * (ci // 16), co // 32, 2, 64, 8).double()  # [step][mt][hl][lane][i]
    rs = np.random.RandomState(0)
    for _ in range(200):
        step, mt, lane, i = rs.randint(9 * (ci // 16)), rs.randint(co // 32), rs.randint(64), rs.randint(8)
        oc, ic, tap = 32 * mt + (lane & 31), 16 * (step % (ci // 16)) + 8 * (lane >> 5) + i, step // (ci // 16)
        want = float(w[oc, ic, tap // 3, tap % 3]) * s
        got = float(flat[step, mt, 0, lane, i] + flat[step, mt, 1, lane, i])
        assert abs(got - want) <= abs(want) * 2.0 ** -21 + 2.0 ** -24, (step, mt, lane, i)
    x = torch.tensor([1.0, 1.0 + 2.0 ** -12, 3.14159265, -1234.567, 6.0e-5]).float().double()
    hi, lo = split_f16(x)
    back = hi.double() + lo.double()
    assert bool(((back - x).abs() <= x.abs() * 2.0 ** -21 + 2.0 ** -24).all())
    assert split_scale(torch.zeros(4, 4)) == 1.0


def test_cu_partition_masks_are_disjoint_equal_shares_of_the_compute_units():
    """engine.cu_partition_masks (the CU sets of CohortRollout's masked streams, bo_stream_create_cu_mask): K disjoint sets of
    n_cu // K CUs each, bit i = CU i; "contiguous" = consecutive CU numbers, "interleaved" = every K-th."""
    import numpy as np
    from betaone_amd.engine import cu_partition_masks

    for n_cu, K in ((256, 2), (256, 4), (256, 8), (304, 4), (256, 3)):
        for layout in ("contiguous", "interleaved"):
            ms = cu_partition_masks(n_cu, K, layout)
            bits = [np.unpackbits(m.view(np.uint8), bitorder="little")[:n_cu].astype(bool) for m in ms]
            assert len(ms) == K and all(m.dtype == np.uint32 and len(m) == (n_cu + 31) // 32 for m in ms)
            assert all(int(b.sum()) == n_cu // K for b in bits)
            assert int(np.sum(bits, axis=0).max()) == 1  # disjoint
            first = np.nonzero(bits[1])[0][:2]
            assert tuple(first) == ((n_cu // K, n_cu // K + 1) if layout == "contiguous" else (1, 1 + K))
    import pytest
    with pytest.raises(ValueError):
        cu_partition_masks(4, 8)


def test_lds_chunk_swizzle_of_the_16x16x32_split_tower_is_conflict_free():
    """csrc/bo_tower_s16.h bo_sw16 (restated here): chunk j of a cell sits at position j ^ 2*((cell % 10) & 7).  A B operand of
    v_mfma_f32_16x16x32_f16 is read per lane as one ds_read_b128: lane l = position tile's column l & 15, channel octet l >> 4 of the
    K-step's 32 channels (chunk 4*group + (l >> 4)).  ds_read_b128 is served in four groups of 16 lanes; for every tap, position
    tile and channel group the 16 lanes of a group must fall on 16 different bank quads."""
    import re
    src = open(os.path.join(ROOT, "betaone_amd", "csrc", "bo_tower_s16.h")).read()
    assert re.search(r"bo_sw16\(int cell\) \{ return 2 \* \(\(cell % 10\) & 7\); \}", src), "restatement out of date"
    half = [[0, 1, 2, 3, 12, 13, 14, 15] + list(range(20, 28)), list(range(4, 12)) + [16, 17, 18, 19, 28, 29, 30, 31]]
    groups = half + [[l + 32 for l in g] for g in half]
    sw16 = lambda cell: 2 * ((cell % 10) & 7)
    for pt in range(4):
        for tap in range(9):
            off = (tap // 3 - 1) * 10 + (tap % 3 - 1)
            for cg in range(4):
                for g in groups:
                    quads = set()
                    for l in g:
                        n16, kb = l & 15, l >> 4
                        cell = ((n16 >> 3) + 1) * 10 + (n16 & 7) + 1 + 20 * pt + off
                        assert 0 <= cell < 100
                        quads.add((cell * 16 + (((4 * cg + kb) ^ sw16(cell)) & 15)) % 16)
                    assert len(quads) == 16, (pt, tap, cg, g)
