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
        _load_gemm_tuning()
    return _ABI


_GEMM_TUNING = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'gemm_tuning_gfx950.csv')


def _load_gemm_tuning():
    """The fp32 path keeps LIBRARY GEMMs for the C x C linear of the coefficient generator; the
    rocBLAS / hipBLASLt solutions PyTorch's TunableOp selected for those shapes on an MI355X (tools/tune_gemm.py) ship
    with the package and are handed to TunableOp with tuning disabled - shapes that are not in the file run the default
    heuristic as before.  FETA_TUNED_GEMM=0 leaves TunableOp alone."""
    if os.environ.get('FETA_TUNED_GEMM', '1') == '0' or not os.path.exists(_GEMM_TUNING):
        return
    try:
        if torch.cuda.is_available() and not torch.cuda.tunable.is_enabled():
            torch.cuda.tunable.enable(True)
            torch.cuda.tunable.tuning_enable(False)
            torch.cuda.tunable.read_file(_GEMM_TUNING)
    except Exception:      # (a library tuning file is never a reason to fail)
        pass


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
