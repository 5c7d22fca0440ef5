"""Loader for libx264hip.so (the C-ABI product library).

There is no CPU fallback: if the shared object is missing or no MI355X is
visible, loading fails loudly.  Nothing here imports or links oracle/.
"""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
SO_PATH = os.path.join(_HERE, "libx264hip.so")
_lib = None


class X264HipError(RuntimeError):
    pass


class Cfg(C.Structure):
    _fields_ = [("device", C.c_int), ("arena_bytes", C.c_size_t)]


def build(verbose=False):
    """Compile csrc/*.hip for gfx950 into libx264hip.so (in-tree)."""
    cmd = ["make", "-C", os.path.join(_HERE, "csrc"), "-j8"]
    if not verbose:
        cmd.insert(1, "-s")
    subprocess.check_call(cmd)
    return SO_PATH


def build_examples():
    """examples/encode_chain: a C99 program that drives the slice level through include/x264hip.h (gcc, no HIP headers)."""
    root = os.path.dirname(_HERE)
    exe = os.path.join(root, "examples", "encode_chain")
    subprocess.check_call(["gcc", "-std=c99", "-O2", "-Wall", "-I" + os.path.join(root, "include"), os.path.join(root, "examples", "encode_chain.c"),
                           "-o", exe, "-L" + _HERE, "-lx264hip", "-lm", "-Wl,-rpath,$ORIGIN/../x264_vs2008_amd"])
    return exe


def open_library():
    """dlopen only (no device needed): used to check exported symbols."""
    if not os.path.exists(SO_PATH):
        raise X264HipError("libx264hip.so not built: run `python -c 'import __graft_entry__ as g; g.build()'`")
    lib = C.CDLL(SO_PATH)
    lib.x264hip_last_error.restype = C.c_char_p
    return lib


def load(device=None, arena_bytes=0):
    """dlopen + x264hip_init on the given device (default LOCAL_RANK or 0)."""
    global _lib
    if _lib is not None:
        return _lib
    lib = open_library()
    if device is None:
        device = int(os.environ.get("LOCAL_RANK", "0"))
    cfg = Cfg(device, arena_bytes)
    rc = lib.x264hip_init(C.byref(cfg))
    if rc != 0:
        raise X264HipError("x264hip_init failed (%d): %s" % (rc, lib.x264hip_last_error().decode()))
    _lib = lib
    return lib
