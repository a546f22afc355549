"""The C-ABI library must build, load without a GPU and export every symbol include/avsim.h declares.
(No compute calls here: those need a device and live in the -m gpu tests.)"""
import ctypes as C
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    txt = open(os.path.join(ROOT, "include", "avsim.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(avsim_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    from av_aloha_amd.build import build_hip
    so = build_hip()
    L = C.CDLL(so)
    syms = declared_symbols()
    assert len(syms) >= 18
    for s in syms:
        assert hasattr(L, s), f"libavsim.so lacks {s}"


def test_create_fails_loudly_without_device():
    import torch
    if torch.cuda.is_available():
        return
    from av_aloha_amd._ffi import AvsimError, Handle
    from avsim_test_util import blob
    try:
        Handle(blob(), 4)
    except AvsimError as e:
        assert "no usable HIP device" in str(e) or "HIP" in str(e)
    else:
        raise AssertionError("avsim_create must not succeed without a GPU (no CPU fallback)")
