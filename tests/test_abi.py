"""CPU: the C-ABI library loads, exports every symbol include/qpnet_hip.h declares, validates
geometry, and refuses to compute without a GPU (no CPU fallback)."""
import ctypes as C
import os
import re

import pytest

from qpnet_amd import _lib
from qpnet_amd.config import TINY, PAPER, DEFAULT, QPNetConfig

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_header_symbols_all_exported():
    hdr = open(os.path.join(ROOT, "include", "qpnet_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(qpn_[a-z0-9_]+)\s*\(", hdr))
    assert declared, "no declarations parsed"
    L = _lib.lib()
    for name in declared:
        assert hasattr(L, name), "missing export " + name
    assert declared == {n for n, _, _ in _lib.SYMBOLS}, "ctypes table out of sync with the header"


def test_no_torch_types_in_abi():
    hdr = open(os.path.join(ROOT, "include", "qpnet_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)      # signatures only, not the prose
    assert "torch" not in hdr.lower() and "at::" not in hdr and "#include <hip" not in hdr


@pytest.mark.parametrize("cfg", [TINY, PAPER, DEFAULT])
def test_param_count(cfg):
    L = _lib.lib()
    assert L.qpn_param_count(C.byref(_lib.make_config(cfg))) == cfg.n_params


def test_param_counts_match_survey():
    assert (TINY.n_params, PAPER.n_params, DEFAULT.n_params) == (52591, 504495, 24151151)


def test_bad_geometry_rejected():
    L = _lib.lib()
    hp = C.c_void_p()
    bad = QPNetConfig(kernel_size=3)
    assert L.qpn_create(C.byref(_lib.make_config(bad)), C.byref(hp)) == -1
    assert b"kernel_size" in L.qpn_last_error()
    assert L.qpn_param_count(C.byref(_lib.make_config(QPNetConfig(n_resch=0)))) == -1


def test_compute_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    L = _lib.lib()
    hp = C.c_void_p()
    assert L.qpn_create(C.byref(_lib.make_config(TINY)), C.byref(hp)) == 0   # geometry-only handle
    buf = (C.c_float * 4)()
    rc = L.qpn_set_weights(hp, C.addressof(buf), TINY.n_params, None)
    assert rc == -2 and b"no CPU fallback" in L.qpn_last_error()
    L.qpn_destroy(hp)


def test_module_refuses_cpu_tensors():
    import torch
    from qpnet_amd.qpnet import QPNet
    m = QPNet(**TINY.kwargs())
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m.batch_fast_generate(torch.zeros(1, 1, dtype=torch.long), torch.zeros(1, 39, 4), [10],
                              __import__("numpy").ones((1, 440)), mode="argmax")
