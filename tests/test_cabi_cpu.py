"""CPU-only: the C-ABI shared library loads and exports every symbol include/dose_hip.h declares; host-side logic
(header parsing, pitched-row detection, error paths that need no GPU)."""
import ctypes
import os

import pytest
import torch


def test_build_and_exports():
    import __graft_entry__ as g
    lib_path = g.build()
    assert os.path.exists(lib_path)
    from dose_prediction_amd import _lib
    L = ctypes.CDLL(lib_path)
    assert len(_lib.PROTOS) >= 33
    for name in _lib.PROTOS:
        assert hasattr(L, name), f"{name} declared in include/dose_hip.h but not exported"
    assert _lib.lib().dp_version() >= 100
    assert _lib.lib().dp_stats_nblk(5000) == 129      # 39 rows per block


def test_header_prototypes_parse():
    from dose_prediction_amd import _lib
    restype, argtypes, names = _lib.PROTOS["dp_gemm_nt"]
    assert restype is ctypes.c_int and len(argtypes) == 23
    assert names[:4] == ["A", "lda", "sa0", "sa1"]
    assert _lib.PROTOS["dp_last_error"][0] is ctypes.c_char_p


def test_rows_ld_detection():
    from dose_prediction_amd import ops
    t = torch.zeros(2, 3, 4, 5, 24)
    assert ops.rows_ld(t) == (120, 24, 24)
    s = t[..., 8:16]
    assert ops.rows_ld(s) == (120, 8, 24)
    with pytest.raises(ValueError):
        ops.rows_ld(t.permute(0, 4, 1, 2, 3))
    assert ops.as_rows(t.permute(0, 2, 1, 3, 4)).is_contiguous()


def test_no_cpu_fallback():
    """The product path must fail loudly on CPU tensors instead of silently computing elsewhere."""
    from dose_prediction_amd import ops, _lib
    x = torch.zeros(1, 4, 4, 4, 8)
    w = torch.zeros(8, 8, 3, 3, 3)
    with pytest.raises(_lib.DoseHipError):
        ops.conv3d(x, w, None, 1, 1, 1)


def test_constructor_errors_match_reference():
    """dose_pyfer.py:43-47 / oar_transeg.py:67-71 raise ValueError for bad dropout / head counts."""
    from dose_prediction_amd.models import dose_pyfer, oar_transeg
    with pytest.raises(ValueError):
        dose_pyfer.ViTEncoder(in_channels=4, img_size=32, hidden_size=50, num_heads=6)
    with pytest.raises(ValueError):
        dose_pyfer.ViTEncoder(in_channels=4, img_size=32, dropout_rate=1.5)
    with pytest.raises(ValueError):
        oar_transeg.Model(1, 8, (32, 32, 32), hidden_size=50, num_heads=12, pos_embed="perceptron")


def test_state_dict_contract():
    """Key names / order / shapes equal the reference's (recorded in the golden fixtures by make_golden.keyinfo)."""
    import numpy as np
    from dose_prediction_amd.models import dose_pyfer, oar_transeg
    here = os.path.dirname(os.path.abspath(__file__))
    g = np.load(os.path.join(here, "golden", "g7_pyfer_model.npz"))
    m = dose_pyfer.Model(9, 1, [-1, 4, 8, 8, 16, 16], feature_size=4, img_size=(32, 16, 16), num_layers=4, num_heads=6)
    sd = m.state_dict()
    assert list(sd.keys()) == list(g["keys"])
    for k, s in zip(g["keys"], g["shapes"]):
        assert tuple(sd[k].shape) == tuple(int(t) for t in s.split(",") if t), k
    for tag, cls in (("new", oar_transeg.Model), ("old", oar_transeg.TRANSEG)):
        g = np.load(os.path.join(here, "golden", f"g7_transeg_{tag}.npz"))
        t = cls(1, 8, (32, 16, 16), feature_size=4, hidden_size=48, mlp_dim=96, num_heads=12, pos_embed="perceptron")
        assert list(t.state_dict().keys()) == list(g["keys"])
    # frozen-net_A filtering by substring works as in train_light_pyfer.py:85-88
    frozen = [k for k, _ in m.named_parameters() if "net_A" in k or "conv_out_A" in k]
    assert len(frozen) == 86      # 84 net_A tensors + conv_out_A.{weight,bias}


def test_ticket_fences_do_not_touch_the_l2(tmp_path):
    """The folded statistics finalize (csrc/norm.hip ticket_is_last) orders row stores -> ticket -> row loads with WORKGROUP-scope
    release / acquire fences (compiler-level ordering) around device-scope atomic accesses.  On gfx950 those fences must
    lower to s_waitcnt only: an agent-scope fence is `buffer_wbl2 sc1` / `buffer_inv sc1`, a write-back / invalidate of the XCD's whole L2
    under the convolutions running beside these kernels (measured: 23.4 -> 28.8 ms per step).  Checked in the ISA of the shipped library:
    the code object that holds the statistics kernels contains the tickets' atomic adds and no L2 maintenance instruction."""
    import glob
    import shutil
    import subprocess
    from dose_prediction_amd import _lib
    objdump = "/opt/rocm/lib/llvm/bin/llvm-objdump"
    if not os.path.exists(objdump):
        pytest.skip("llvm-objdump not available")
    so = shutil.copy(_lib.LIB_PATH, tmp_path / "lib.so")
    subprocess.run([objdump, "--offloading", str(so)], check=True, capture_output=True, cwd=tmp_path)
    found = False
    for co in sorted(glob.glob(str(tmp_path / "lib.so.*gfx950"))):
        syms = subprocess.run([objdump, "-t", co], capture_output=True, text=True).stdout
        if "k_stats_partial" not in syms:
            continue
        found = True
        isa = subprocess.run([objdump, "-d", co], capture_output=True, text=True, check=True).stdout
        assert isa.count("global_atomic_add") >= 5, "the ticket draws (three storage types of k_stats_partial + the backward partial kernels)"
        assert "buffer_wbl2" not in isa and "buffer_inv" not in isa, "an L2 write-back / invalidate crept into the normalisation kernels"
    assert found, "no gfx950 code object with k_stats_partial in the library"


def test_fast_binding_is_generated_from_the_header_and_agrees_with_ctypes():
    """The METH_FASTCALL module (dose_prediction_amd/_fastgen.py, round 6) is generated from include/dose_hip.h: one wrapper per declared
    prototype, same results as the ctypes binding on the entry points that need no GPU, argument-count and type errors raised as TypeError
    (not a crash), NULL for None, and `bind` refusing a library that lacks a declared symbol."""
    import __graft_entry__ as g
    g.build()
    from dose_prediction_amd import _lib, _fastgen
    L = _lib.lib()
    assert _lib.BINDING == "fastcall" and os.path.exists(_fastgen.SO_PATH)
    from dose_prediction_amd import _dose_fastcall as F
    protos = _fastgen.parse(_lib.HEADER)
    assert {n for n, _, _ in protos} == set(_lib.PROTOS)
    for name, rk, args in protos:
        assert hasattr(F, name), name
        assert len(args) == len(_lib.PROTOS[name][1]), name
    C = L._cdll
    for args in ((32, 16, 7, 1, 3, 1, 128), (16, 16, 3, 1, 1, 1, 128), (64, 32, 7, 1, 3, 1, 64), (3, 8, 3, 2, 1, 1, 32)):
        assert L.dp_conv3d_tiled_weight_elems(*args) == C.dp_conv3d_tiled_weight_elems(*args)
    for v in (1, 5000, 128 ** 3, 192 * 192 * 128):
        assert L.dp_stats_nblk(v) == C.dp_stats_nblk(v)
    assert L.dp_conv3d_tiled_ws_elems(2, 16, 8, 8, 128, 64, 7) == C.dp_conv3d_tiled_ws_elems(2, 16, 8, 8, 128, 64, 7)
    assert L.dp_version() == C.dp_version() and isinstance(L.dp_last_error(), bytes)
    with pytest.raises(TypeError):
        L.dp_stats_nblk()
    with pytest.raises(TypeError):
        L.dp_stats_nblk(1.5)
    with pytest.raises(TypeError):
        L.dp_stats_nblk("7")
    with pytest.raises(OverflowError):
        L.dp_conv3d_tiled_weight_elems(1 << 40, 16, 7, 1, 3, 1, 128)
    # a launch entry point with a NULL pointer argument and no GPU: the library reports an error status, the binding does not crash
    with pytest.raises(_lib.DoseHipError):
        _lib.call("dp_stats_partial", None, 8, 1, 64, 5000, None, 0, None)       # C > 8 * NT: rejected before any launch
    with pytest.raises(AttributeError):
        F.bind("/usr/lib/x86_64-linux-gnu/libm.so.6")
    F.bind(_lib.LIB_PATH)       # (restore the binding for the tests that follow)
