"""CPU: the C-ABI library loads and exports every symbol include/sfamd.h declares (no compute calls), the ctypes
signature table covers exactly that set, and the product loader refuses to fall back when the library is absent."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "sfamd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(sf_[a-z0-9_]+)\s*\(", text)))


def test_header_matches_ctypes_table():
    from slowfast_amd import lib
    assert declared_symbols() == sorted(lib.EXPORTED_SYMBOLS)
    text = open(os.path.join(ROOT, "include", "sfamd.h")).read()
    assert int(re.search(r"#define SF_ABI_VERSION (\d+)", text).group(1)) == lib.ABI_VERSION


def test_gfx950_library_exports_every_symbol():
    from slowfast_amd import build_ext
    path = build_ext.build_hip()          # hipcc cross-compiles without a GPU
    dll = ctypes.CDLL(path)
    for name in declared_symbols():
        assert hasattr(dll, name), f"{name} missing from {path}"
    dll.sf_backend.restype = ctypes.c_char_p
    assert dll.sf_backend() == b"gfx950"
    assert dll.sf_abi_version() == __import__("slowfast_amd.lib", fromlist=["x"]).ABI_VERSION


def test_hostsim_library_exports_every_symbol(hostsim_path):
    dll = ctypes.CDLL(hostsim_path)
    for name in declared_symbols():
        assert hasattr(dll, name)
    dll.sf_backend.restype = ctypes.c_char_p
    assert dll.sf_backend() == b"hostsim"


def test_missing_library_fails_loudly(tmp_path, monkeypatch):
    from slowfast_amd import lib
    monkeypatch.setenv("SFAMD_LIBRARY", str(tmp_path / "nope.so"))
    lib.reset_lib()
    with pytest.raises(lib.SfError, match="no CPU/PyTorch fallback"):
        lib.get_lib()
    lib.reset_lib()


def test_stale_binary_is_refused(tmp_path, monkeypatch, hostsim_path):
    """Build provenance: every binary carries the hash of the sources it was compiled from (sf_build_id); a binary whose id
    differs from the csrc/ + include/ beside it is refused at load, and build_ext treats it as stale whatever its mtime."""
    from slowfast_amd import build_ext, lib
    for path, sim in ((build_ext.build_hip(), False), (hostsim_path, True)):
        want = build_ext.source_id(sim)
        dll = ctypes.CDLL(path)
        dll.sf_build_id.restype = ctypes.c_char_p
        assert dll.sf_build_id().decode() == want == build_ext._built_id(path)
        assert not build_ext._stale(path, sim)
    assert build_ext.source_id(False) != build_ext.source_id(True), "the test shim is hashed into the simulator builds only"
    blob = open(hostsim_path, "rb").read()
    tag = b"sfamd-build-id:" + want.encode()
    assert blob.count(tag) == 1
    old = tmp_path / "libsfamd_old.so"
    old.write_bytes(blob.replace(tag, b"sfamd-build-id:" + b"0123456789abcdef"))
    os.utime(old, (2e9, 2e9))                                   # newer than every source: mtimes do not vouch for a binary
    assert build_ext._stale(str(old), True)
    monkeypatch.setenv("SFAMD_LIBRARY", str(old))
    lib.reset_lib()
    with pytest.raises(lib.SfError, match="built from other sources"):
        lib.get_lib()
    monkeypatch.setenv("SF_ALLOW_STALE_LIBRARY", "1")
    lib.reset_lib()
    assert lib.get_lib().build_id == "0123456789abcdef"
    lib.reset_lib()


def test_cpu_tensor_with_gfx950_library_is_rejected(monkeypatch):
    """The product library never computes on host tensors."""
    import torch
    from slowfast_amd import lib, ops
    monkeypatch.delenv("SFAMD_LIBRARY", raising=False)
    lib.reset_lib()
    try:
        assert lib.get_lib().backend == "gfx950"
        with pytest.raises(lib.SfError, match="need CUDA/HIP tensors"):
            ops.ncthw_to_cl(torch.zeros(1, 3, 1, 4, 4))
    finally:
        lib.reset_lib()


def test_error_reporting_through_c_abi(sim):
    from slowfast_amd import lib, ops
    import torch
    geom = ops.ConvGeom((1, 12, 1, 4, 4), 16, (1, 1, 1))
    with pytest.raises(lib.SfError, match="multiples of 8"):
        ops.prep_weights(torch.zeros(16, 12, 1, 1, 1), geom)
    with pytest.raises(lib.SfError, match="not channels-last"):
        ops.cl_ld(torch.zeros(1, 16, 1, 4, 4, dtype=lib.act_dtype()))


def test_captured_step_is_kernel_nodes_only():
    """libsfamd issues kernel launches only: a hipMemsetAsync in front of the rel-pos scatter became a memset node under stream
    capture whose fill did not take effect before its readers from the second replay on (profiles/r4/r4_v13_graph_memset.md)."""
    import glob
    import re
    csrc = os.path.join(ROOT, "slowfast_amd", "csrc")
    for path in sorted(glob.glob(os.path.join(csrc, "*"))):
        with open(path) as f:
            code = re.sub(r"//[^\n]*", "", f.read())
        assert not re.search(r"\bhipMem(set|cpy)\w*\s*\(", code), path
