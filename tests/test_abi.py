"""CPU tests of the drop-in boundary: the C-ABI library loads and exports every symbol that
include/hoqp.h declares; argument validation that needs no GPU."""
import ctypes as C
import os
import re

import pytest

from quadruped_locomotion_amd import hoqp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "hoqp.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(hoqp_[a-z_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    lib = hoqp.load_library()
    names = _declared()
    assert len(names) >= 12
    for name in names:
        assert hasattr(lib, name), f"{name} declared in include/hoqp.h but not exported"
    assert set(hoqp.EXPORTS) == set(names)


def test_default_config_matches_reference_constants():
    cfg = hoqp.default_config()
    assert cfg.semantics == hoqp.SEMANTICS_REFERENCE
    assert cfg.eps == 1e-12                      # ho_qp_problem.hpp:56
    assert (cfg.mu, cfg.tau_max, cfg.tau_min) == (0.5, 100.0, -100.0)  # ho_qp_controller.hpp:54-56
    assert hoqp.default_config("paper").semantics == hoqp.SEMANTICS_PAPER


def test_struct_sizes_match_header():
    # 3 ints + 2*8 ints + 4*8 pointers + 3 pointers, with natural alignment
    assert C.sizeof(hoqp.TaskBatch) == 4 * 3 + 4 * 16 + 4 + 8 * 32 + 8 * 3
    assert C.sizeof(hoqp.ResultBatch) == 8 * 12
    assert C.sizeof(hoqp.WbcInputs) == 8 + 8 * 11
    assert C.sizeof(hoqp.WbcOutputs) == 8 * 13


def test_lds_budget_for_wbc_extents():
    b = hoqp.lds_bytes(30, hoqp.WBC_M, hoqp.WBC_P)
    assert 0 < b <= 160 * 1024
    assert hoqp.lds_bytes(33, [1], [0]) == 0      # n > HOQP_MAX_VARS is rejected


def test_no_device_no_fallback():
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(RuntimeError):
        hoqp.HoQpSolver()
    lib = hoqp.load_library()
    cfg = hoqp.default_config()
    h = C.c_void_p()
    assert lib.hoqp_create(C.byref(cfg), C.byref(h)) < 0   # fails loudly, never a CPU path
    assert lib.hoqp_create(None, None) == -1


def test_product_never_touches_oracle():
    """The product tree must not reference the oracle (a routed-through checker voids parity)."""
    pkg = os.path.join(ROOT, "quadruped_locomotion_amd")
    for dirpath, _, files in os.walk(pkg):
        for fn in files:
            if fn.endswith((".py", ".hip", ".hpp", ".h", ".cpp")):
                txt = open(os.path.join(dirpath, fn), errors="ignore").read()
                assert "oracle" not in txt.lower(), f"{fn} mentions the oracle"
    for fn in ("hoqp.h", "hoqp.hpp"):
        assert "oracle" not in open(os.path.join(ROOT, "include", fn)).read().lower()
