"""The C-ABI boundary: libqpx_hip.so loads and exports exactly what include/qpx.h declares
(no compute calls -- those need a GPU and live in the -m gpu tests)."""
import os
import re
import subprocess

import pytest

from conftest import ROOT

HEADER = os.path.join(ROOT, "include", "qpx.h")
HIP_SO = os.path.join(ROOT, "qpth_amd", "libqpx_hip.so")


def declared_symbols():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(qpx_[a-z_0-9]+)\s*\(", src)))


@pytest.fixture(scope="module")
def built():
    import __graft_entry__ as g
    g.build()          # hipcc cross-compiles gfx950 without a GPU
    return HIP_SO


def test_header_and_binding_agree():
    from qpth_amd import _lib
    assert sorted(_lib.ABI_SYMBOLS) == declared_symbols()


def test_hip_library_exports_every_declared_symbol(built):
    out = subprocess.check_output(["nm", "-D", "--defined-only", built]).decode()
    exported = set(re.findall(r" T (qpx_[a-z_0-9]+)", out))
    assert set(declared_symbols()) <= exported, set(declared_symbols()) - exported


def test_hip_library_loads_and_answers_metadata(built):
    from qpth_amd import _lib
    lib = _lib.QpxLib(built)
    assert lib.dll.qpx_abi_version() == 8
    assert lib.dll.qpx_max_dim() == 1024
    # factor blob per QP: -K, (G K)^T and ONE register image of R -- 217 KB at C2 in f64 (<= 250 KB),
    # 5.5 GB for all 65 536 QPs of C5 (<= 6 GB)
    assert 100 * 100 * 2 < lib.factor_elems(_lib.QPX_F64, 100, 100, 0) * 8 <= 250 * 1024
    assert lib.factor_elems(_lib.QPX_F64, 64, 64, 0) * 8 * 65536 <= 6 * 2 ** 30
    assert lib.dll.qpx_can_share_factors(_lib.QPX_F64, 100, 100, 0) == 1      # C2: tile kernels, the blob is read-only
    assert lib.dll.qpx_can_share_factors(_lib.QPX_F64, 500, 500, 0) == 0      # C4: per-QP work matrices live in it
    assert lib.dll.qpx_kernel_family(_lib.QPX_F64, 100, 100, 0) == _lib.FAMILY_TILE
    assert lib.dll.qpx_kernel_family(_lib.QPX_F64, 500, 500, 0) == _lib.FAMILY_BIG
    assert b"not supported" in lib.dll.qpx_strerror(-2)
    # the A/B knob: retired bits and values are dropped (set-then-get shows a script that its knob no longer exists)
    old = lib.dll.qpx_set_ipm_variant(3 | (2 << 16) | (1 << 25) | (1 << 30))
    try:
        assert lib.dll.qpx_get_ipm_variant() == 3 | (2 << 16)
        lib.dll.qpx_set_ipm_variant(1)                   # the workgroup kernels of round 1: deleted in v7
        assert lib.dll.qpx_get_ipm_variant() == 0
    finally:
        lib.dll.qpx_set_ipm_variant(old)


def test_hip_library_contains_gfx950_code(built):
    blob = open(built, "rb").read()
    assert b"hipv4-amdgcn-amd-amdhsa--gfx950" in blob          # the fat binary targets MI355X
    assert b"gfx942" not in blob and not re.search(rb"sm_\d\d", blob)          # ... and nothing else (no other AMD target, no CUDA sm_NN)


def test_missing_extension_fails_loudly(tmp_path):
    from qpth_amd import _lib
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        _lib.QpxLib(str(tmp_path / "libqpx_hip.so"))


def test_cpu_tensors_are_refused():
    """The product path never falls back to a CPU implementation."""
    import torch
    from qpth_amd.qp import QPFunction
    Q = torch.eye(3, dtype=torch.float64).unsqueeze(0)
    p = torch.zeros(1, 3, dtype=torch.float64)
    G = torch.ones(1, 2, 3, dtype=torch.float64)
    h = torch.ones(1, 2, dtype=torch.float64)
    e = torch.empty(0, dtype=torch.float64)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        QPFunction(verbose=-1)(Q, p, G, h, e, e)
