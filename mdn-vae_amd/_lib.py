"""ctypes binding of libmdnvae_hip.so (the C ABI declared in include/mdnvae_hip.h).

There is no CPU fallback: if the library is missing or a tensor is not a contiguous
fp32 tensor on a HIP device, the call raises."""
import ctypes
import os
import re

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libmdnvae_hip.so")
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "mdnvae_hip.h")

_lib = None

_CTYPES = {
    "int": ctypes.c_int, "long": ctypes.c_long, "float": ctypes.c_float, "double": ctypes.c_double,
    "size_t": ctypes.c_size_t, "unsigned long long": ctypes.c_ulonglong,
}


def declared_functions():
    """[(name, return type, [argument C types])] parsed from the header."""
    text = open(HEADER_PATH).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    out = []
    for m in re.finditer(r"\b(int|size_t|const char\*)\s+(mdnvae_\w+)\s*\(([^)]*)\)\s*;", text):
        ret, name, args = m.group(1), m.group(2), m.group(3).strip()
        argt = []
        if args and args != "void":
            for a in args.split(","):
                a = a.strip()
                if "*" in a:
                    argt.append("ptr")
                else:
                    argt.append(" ".join(a.split()[:-1]).replace("const ", "").strip())
        out.append((name, ret, argt))
    return out


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(f"{LIB_PATH} is missing: build it with `make` (hipcc --offload-arch=gfx950) "
                           "before using mdnvae_amd; there is no fallback path")
    lib = ctypes.CDLL(LIB_PATH)
    for name, ret, argt in declared_functions():
        fn = getattr(lib, name)
        fn.restype = {"int": ctypes.c_int, "size_t": ctypes.c_size_t, "const char*": ctypes.c_char_p}[ret]
        fn.argtypes = [ctypes.c_void_p if a == "ptr" else _CTYPES[a] for a in argt]
    _lib = lib
    return lib


def ptr(t, name="tensor", dtype=torch.float32):
    """Device pointer of a tensor the kernels can read as-is (None passes through as NULL)."""
    if t is None:
        return None
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{name}: expected a torch.Tensor, got {type(t).__name__}")
    if not t.is_cuda:
        raise RuntimeError(f"{name}: tensor is on {t.device}; the HIP path needs device memory (no CPU fallback)")
    if t.dtype != dtype:
        raise RuntimeError(f"{name}: dtype {t.dtype}, expected {dtype}")
    if not t.is_contiguous():
        raise RuntimeError(f"{name}: tensor must be contiguous")
    return t.data_ptr()


def stream():
    return torch.cuda.current_stream().cuda_stream


def call(fname, *args):
    lib = load()
    rc = getattr(lib, fname)(*args)
    if rc != 0:
        msg = lib.mdnvae_last_error().decode(errors="replace")
        raise RuntimeError(f"{fname} failed (code {rc}): {msg}")


def workspace_bytes(query):
    return int(getattr(load(), query)())


def workspace_bytes_f(query, *args):
    """Workspace size in FLOATS from a *_workspace_floats(...) query."""
    return int(getattr(load(), query)(*args))
