"""CPU-side checks of the C-ABI boundary: the library builds for gfx950, loads, exports every
symbol include/radex_emcee_amd.h declares, and refuses to compute without a GPU (no fallback)."""
import ctypes as C
import os
import re

import pytest

from radex_emcee_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    _lib.build()
    return _lib.load()


def _declared_functions():
    hdr = open(os.path.join(ROOT, "include", "radex_emcee_amd.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(rx_[a-z_]+)\s*\(", hdr)))


def test_header_and_export_list_agree():
    assert _declared_functions() == sorted(_lib.EXPORTS)


def test_library_exports_every_declared_symbol(lib):
    for name in _declared_functions():
        assert hasattr(lib, name), name
    assert lib.rx_abi_version() == 7


def test_header_cites_reference_interfaces():
    hdr = open(os.path.join(ROOT, "include", "radex_emcee_amd.h")).read()
    for cite in ("emcee/emcee_radex.py:177-181", "emcee/emcee_radex.py:120-130",
                 "emcee/emcee_radex.py:104-117", "core.py:845-854", "lubksb_"):
        assert cite in hdr, cite


def test_bad_arguments_do_not_crash(lib):
    err = C.create_string_buffer(256)
    assert not lib.rx_create(b"/nonexistent/co.dat", 2, 1.0, 0, err, 256)
    assert b"cannot open" in err.value
    assert not lib.rx_create(None, 2, 1.0, 0, err, 256)
    assert not lib.rx_create(b"x", 7, 1.0, 0, err, 256)
    assert lib.rx_nlev(None) < 0
    lib.rx_destroy(None)


def test_no_cpu_fallback(lib, co_path):
    """Without a HIP device the engine must fail loudly instead of computing on the CPU."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    err = C.create_string_buffer(256)
    h = lib.rx_create(co_path.encode(), 2, 1.0, 0, err, 256)
    assert not h
    assert b"no usable HIP device" in err.value
    from radex_emcee_amd.engine import Engine, EngineError
    with pytest.raises(EngineError):
        Engine(co_path)


def test_molecule_limits_are_reported(lib, tmp_path):
    """More than 64 levels cannot map to one wavefront: rx_create says so (RX_E_UNSUPP path)."""
    from radex_emcee_amd.molecule import synth_co_text
    p = tmp_path / "big.dat"
    p.write_text(synth_co_text(nlev=70))
    err = C.create_string_buffer(256)
    assert not lib.rx_create(str(p).encode(), 2, 1.0, 0, err, 256)
    assert b"exceeds kernel limits" in err.value
    q = tmp_path / "bad.dat"
    q.write_text("!MOLECULE\nX\n!W\n1.0\n!N\n2\n!L\n1 0.0 1.0\n")
    assert not lib.rx_create(str(q).encode(), 2, 1.0, 0, err, 256)
    assert b"malformed" in err.value


def test_product_never_touches_the_oracle():
    """Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may use oracle/."""
    pkg = os.path.join(ROOT, "radex_emcee_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".inc", ".h", ".cpp")):
                txt = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in txt, (dirpath, f)
