"""ctypes binding of libmiq.so (the C ABI declared in include/miq.h).

The library is the only implementation of the GPU path: if it cannot be loaded, every op raises.
There is no PyTorch fallback for CUDA/HIP tensors.
"""
from __future__ import annotations

import ctypes
import os
import re

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MIQ_LIB") or os.path.join(_HERE, "libmiq.so")
HEADER_PATH = os.path.normpath(os.path.join(_HERE, "..", "..", "include", "miq.h"))

F32, F16 = 0, 1
MIN_VERSION = 303  # include/miq.h: what each interface version added
_DT = {torch.float32: F32, torch.float16: F16}

_lib = None


class MiqError(RuntimeError):
    pass


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise MiqError(
                f"{LIB_PATH} is missing: build it with `make -C llm-quantization_amd/csrc` "
                "(or python -c 'import __graft_entry__ as g; g.build()'); there is no fallback path")
        _lib = ctypes.CDLL(LIB_PATH)
        _lib.miq_last_error.restype = ctypes.c_char_p
        _lib.miq_version.restype = ctypes.c_int
        if _lib.miq_version() < MIN_VERSION:
            # entry points keep their names when an argument changes meaning: only the version tells a stale build
            raise MiqError(f"{LIB_PATH} implements interface version {_lib.miq_version()}, this package needs "
                           f">= {MIN_VERSION}: rebuild it with `make -C llm-quantization_amd/csrc`")
        _lib.miq_act_workspace.restype = ctypes.c_int64
        _lib.miq_act_static_capacity.restype = ctypes.c_int64
        _lib.miq_act_fresh_workspace.restype = ctypes.c_int64
        _lib.miq_act_fresh_bwd_workspace.restype = ctypes.c_int64
        _lib.miq_let_workspace.restype = ctypes.c_int64
        _lib.miq_rmsnorm_workspace.restype = ctypes.c_int64
        _lib.miq_gather_layernorm_bwd_workspace.restype = ctypes.c_int64
        _lib.miq_col_sum_workspace.restype = ctypes.c_int64
        _lib.miq_mse_workspace.restype = ctypes.c_int64
    return _lib


def source_digest() -> str:
    """16 hex digits over the kernel sources next to the library (csrc/*.hip, *.h): measurement files record it so that a
    reader can tell whether they were taken on the build at hand."""
    import glob
    import hashlib
    h = hashlib.sha256()
    src = os.path.normpath(os.path.join(_HERE, "..", "csrc"))
    for path in sorted(glob.glob(os.path.join(src, "*.hip")) + glob.glob(os.path.join(src, "*.h"))):
        h.update(os.path.basename(path).encode())
        with open(path, "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def declared_symbols():
    """Function names declared in include/miq.h (used by the ABI completeness test)."""
    with open(HEADER_PATH) as f:
        text = f.read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(miq_[a-z0-9_]+)\s*\(", text)))


def dtype_code(t: torch.Tensor) -> int:
    try:
        return _DT[t.dtype]
    except KeyError:
        raise MiqError(f"unsupported dtype {t.dtype}: the kernels handle float32 and float16") from None


def ptr(t):
    if t is None:
        return ctypes.c_void_p(0)
    return ctypes.c_void_p(t.data_ptr())


def stream_of(t: torch.Tensor):
    # the raw handle of torch's current stream on t's device (no Stream object: this runs once per kernel launch)
    return ctypes.c_void_p(torch._C._cuda_getCurrentRawStream(t.device.index))


class _NoGuard:
    def __enter__(self):
        return None

    def __exit__(self, *exc):
        return False


_NO_GUARD = _NoGuard()


def on_device(dev):
    """Context manager that makes ``dev`` the current device for a launch; free when it already is (the usual
    case: one process per GPU), since torch.cuda.device() costs several microseconds per call."""
    if dev is None or dev.index is None or dev.index == torch.cuda.current_device():
        return _NO_GUARD
    return torch.cuda.device(dev)


def check_gpu(*tensors):
    dev = None
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise MiqError("libmiq kernels need GPU tensors; CPU tensors are not supported by the product path")
        if not t.is_contiguous():
            raise MiqError("libmiq kernels need contiguous tensors")
        if dev is None:
            dev = t.device
        elif t.device != dev:
            raise MiqError("all tensors of one call must live on the same device")
    return dev


def call(name: str, *args):
    fn = getattr(lib(), name)
    rc = fn(*args)
    if rc != 0:
        raise MiqError(f"{name} failed ({rc}): {lib().miq_last_error().decode()}")


i64 = ctypes.c_int64
i32 = ctypes.c_int
f32 = ctypes.c_float
