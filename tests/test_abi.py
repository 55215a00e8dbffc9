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


def _header_structs():
    """struct name -> [(field name, 'ptr' | 'int' | 'float' | 'int64')] in declaration order, from include/feta_hip.h"""
    txt = open(os.path.join(ROOT, 'include', 'feta_hip.h')).read()
    txt = re.sub(r'/\*.*?\*/', '', txt, flags=re.S)
    out = {}
    for m in re.finditer(r'typedef struct (\w+) \{(.*?)\} \1;', txt, flags=re.S):
        fields = []
        for decl in m.group(2).split(';'):
            decl = ' '.join(decl.split())
            if not decl:
                continue
            ptr = '*' in decl
            base = decl.replace('const ', '').replace('*', ' ').split()
            kind = 'ptr' if ptr else {'int': 'int', 'float': 'float', 'int64_t': 'int64', 'int32_t': 'int'}[base[0]]
            for name in ' '.join(base[1:]).split(','):
                fields.append((name.strip(), kind))
        out[m.group(1)] = fields
    return out


@pytest.mark.parametrize('cname,pyname', [('feta_attn_block', 'AttnBlock'), ('feta_ffn', 'Ffn'),
                                          ('feta_attn_block_grad', 'AttnBlockGrad'), ('feta_ffn_grad', 'FfnGrad'),
                                          ('feta_rowlin_ex', 'RowLinEx'), ('feta_colsum_seg', 'ColsumSeg'),
                                          ('feta_coeff_fwd_role', 'CoeffFwdRole'), ('feta_coeff_bwd_role', 'CoeffBwdRole'),
                                          ('feta_spec_cat', 'SpecCat')])
def test_descriptor_layouts_agree(cname, pyname):
    """Every descriptor struct of the header has the same fields, in the same order and of the same kind, as its ctypes
    mirror - a field added on one side only would shift every pointer behind it."""
    kinds = {ctypes.c_void_p: 'ptr', ctypes.c_int: 'int', ctypes.c_float: 'float', ctypes.c_int64: 'int64'}
    got = [(n.rstrip('_'), kinds[t]) for n, t in getattr(_abi, pyname)._fields_]     # (in_: `in` is a Python keyword)
    assert got == _header_structs()[cname]
