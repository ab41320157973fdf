"""The C-ABI library loads (no GPU needed) and exports every symbol that include/miq.h declares."""
import ctypes

from miq import _lib


def test_header_symbols_exported():
    names = _lib.declared_symbols()
    assert "miq_version" in names and "miq_rowquant_uniform_fwd" in names
    lib = _lib.lib()
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, f"declared in miq.h but not exported by libmiq.so: {missing}"


def test_version_and_error_text():
    lib = _lib.lib()
    assert lib.miq_version() >= 100
    assert isinstance(lib.miq_last_error(), bytes)


def test_bad_arguments_are_rejected_without_a_gpu():
    # argument validation happens before any HIP call, so it can be exercised on the CPU box
    lib = _lib.lib()
    rc = lib.miq_rowquant_uniform_fwd(None, None, None, None, None, None, None, None, ctypes.c_int64(4),
                                      ctypes.c_int64(8), ctypes.c_int64(0), 4, 0, 0, 0, None)
    assert rc == -22
    assert b"null" in lib.miq_last_error()
