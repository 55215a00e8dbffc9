"""Loads feta_tmlr_amd/libfeta_hip.so (built by ``python -m feta_tmlr_amd.build``).

There is deliberately no CPU fallback: every op of this package runs in the HIP
kernels or raises.  The oracle under ``oracle/`` is test infrastructure and is never
imported from here.
"""
import ctypes
import os

import torch

from ._abi import Abi, FetaError, bind

_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'libfeta_hip.so')
_ABI = None


def library_path():
    return _PATH


def abi() -> Abi:
    global _ABI
    if _ABI is None:
        if not os.path.exists(_PATH):
            raise FetaError('%s is missing: build it with `python -m feta_tmlr_amd.build` '
                            '(hipcc, gfx950). There is no CPU fallback.' % _PATH)
        _ABI = bind(ctypes.CDLL(_PATH))
    return _ABI


_TEST_ABI = None


class override_for_tests:
    """TEST HOOK (tests/ only): route the ops to an explicitly supplied Abi - the host
    SIMT emulation of the SAME kernel sources - so that the autograd / module wiring can be
    exercised without a GPU.  Nothing in the package installs it; without it CPU tensors raise."""

    def __init__(self, test_abi):
        self.abi = test_abi

    def __enter__(self):
        global _TEST_ABI
        self.prev, _TEST_ABI = _TEST_ABI, self.abi
        return self

    def __exit__(self, *exc):
        global _TEST_ABI
        _TEST_ABI = self.prev


def backend(*tensors):
    """-> (abi, stream handle) for an op on these tensors; raises for non-GPU tensors."""
    if _TEST_ABI is not None:
        return _TEST_ABI, None
    require_cuda(*tensors)
    return abi(), stream_handle()


def stream_handle():
    """hipStream_t of torch's current stream (kernels are enqueued there, so torch ops,
    torch.cuda.Event timing and hipGraph capture all see them)."""
    return torch.cuda.current_stream().cuda_stream


def require_cuda(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise FetaError('feta_tmlr_amd ops run on the MI355X only (got a %s tensor); '
                            'there is no CPU fallback' % t.device)
