import ctypes
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session')
def emu():
    """ABI bound to the host SIMT emulation of the kernel sources (tools/simt)."""
    from feta_tmlr_amd import _abi
    so = os.path.join(ROOT, 'tools', 'simt', 'libfeta_emu.so')
    srcs = [os.path.join(ROOT, 'feta_tmlr_amd', 'csrc', f)
            for f in os.listdir(os.path.join(ROOT, 'feta_tmlr_amd', 'csrc'))]
    srcs += [os.path.join(ROOT, 'tools', 'simt', f) for f in ('simt_runtime.cpp', 'feta_device.h')]
    srcs += [os.path.join(ROOT, 'include', 'feta_hip.h'), os.path.join(ROOT, 'tools', 'simt', 'hip', 'hip_runtime.h')]
    if not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call([os.path.join(ROOT, 'tools', 'simt', 'build.sh')])
    return _abi.bind(ctypes.CDLL(so))


@pytest.fixture(scope='session')
def hip():
    """(abi, device, stream) of the real library on cuda:0."""
    import torch
    from feta_tmlr_amd import _lib
    assert torch.cuda.is_available(), 'gpu tests need a GPU'
    return _lib.abi(), torch.device('cuda:0'), torch.cuda.current_stream().cuda_stream
