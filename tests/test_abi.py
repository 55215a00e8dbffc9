"""The C-ABI library loads without a GPU and exports every symbol include/feta_hip.h declares
(no compute calls here).  Also checks that the header and the ctypes binding list the same names."""
import ctypes
import os
import re

import pytest

from feta_tmlr_amd import _abi, _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_functions():
    txt = open(os.path.join(ROOT, 'include', 'feta_hip.h')).read()
    txt = re.sub(r'/\*.*?\*/', '', txt, flags=re.S)
    return sorted(set(re.findall(r'\b(feta_[a-z0-9_]+)\s*\(', txt)))


def test_header_and_binding_agree():
    assert _header_functions() == sorted(_abi.SIGNATURES)


def test_libfeta_hip_exports_every_symbol():
    path = _lib.library_path()
    if not os.path.exists(path):
        from feta_tmlr_amd import build
        build.build(verbose=False)
    lib = ctypes.CDLL(path)
    for name in _header_functions():
        assert hasattr(lib, name), name
    abi = _abi.bind(lib)
    assert abi.lib.feta_version() == _abi.ABI_VERSION
    assert abi.coeff_bwd_groups(128, 4) == 128


def test_emulation_exports_the_same_abi(emu):
    for name in _header_functions():
        assert hasattr(emu.lib, name), name


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    monkeypatch.setattr(_lib, '_PATH', str(tmp_path / 'libfeta_hip.so'))
    monkeypatch.setattr(_lib, '_ABI', None)
    with pytest.raises(_abi.FetaError):
        _lib.abi()
