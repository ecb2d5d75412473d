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


def test_abi_under_sanitizers():
    """SURVEY.md section 5 (host tests under ASan / UBSan): the C ABI's argument-validation paths that run without
    a GPU, host side compiled with -fsanitize=address,undefined (tests/Makefile.asan, about 3 minutes of hipcc:
    built on demand -- set HOQP_BUILD_ASAN=1 to (re)build here -- and run whenever the binary is present)."""
    import os
    import subprocess

    import pytest

    here = os.path.dirname(os.path.abspath(__file__))
    exe = os.path.join(here, "abi_asan_check")
    if os.environ.get("HOQP_BUILD_ASAN") == "1":
        subprocess.check_call(["make", "-s", "-C", here, "-f", "Makefile.asan", "abi_asan_check"])
    if not os.path.exists(exe):
        pytest.skip("tests/abi_asan_check not built (make -C tests -f Makefile.asan)")
    res = subprocess.run([exe], capture_output=True, text=True, timeout=300,
                         env=dict(os.environ, ASAN_OPTIONS="detect_leaks=0", UBSAN_OPTIONS="print_stacktrace=1"))
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    assert "abi_asan_check: ok" in res.stdout


def test_shim_validates_in_the_constructor():
    """include/hoqp.hpp: HoQpProblem solves lazily, but malformed tasks (A/b or D/f row mismatch, too many variables
    or rows, a column count that differs from the parent's) must fail at CONSTRUCTION, where the reference's
    constructor would have asserted -- checked by tests/shim_check --validate, which needs no GPU."""
    import subprocess

    exe = os.path.join(ROOT, "tests", "shim_check")
    if not os.path.exists(exe):
        import pytest

        pytest.skip("tests/shim_check not built (__graft_entry__.build())")
    res = subprocess.run([exe, "--validate"], capture_output=True, text=True, timeout=60)
    assert res.returncode == 0 and "validate caught 5 of 5" in res.stdout, res.stdout + res.stderr
