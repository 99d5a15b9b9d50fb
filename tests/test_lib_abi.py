"""CPU-side checks of the C ABI: the shared library loads and exports every symbol that
include/m3p_hip.h declares, and the ctypes table in m3p_amd/lib.py covers them all.
(No compute calls: there is no GPU here.)"""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, 'include', 'm3p_hip.h')).read()
    return re.findall(r'M3P_API\s+[\w\s\*]+?\b(m3p_\w+)\s*\(', src)


def _ensure_built():
    so = os.path.join(ROOT, 'm3p_amd', 'libm3p_hip.so')
    if not os.path.exists(so):
        import __graft_entry__ as g
        g.build()
    return so


def test_header_symbols_exported():
    names = _declared()
    assert len(names) >= 5
    lib = ctypes.CDLL(_ensure_built())
    for n in names:
        assert hasattr(lib, n), 'libm3p_hip.so does not export %s' % n


def test_ctypes_table_matches_header():
    from m3p_amd import lib as L
    assert set(L.SIGNATURES) == set(_declared())
    _ensure_built()
    L.load()
    assert L.load().m3p_version().decode().startswith('m3p_hip')


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from m3p_amd import lib as L
    monkeypatch.setattr(L, '_lib', None)
    monkeypatch.setattr(L, 'LIB_PATH', str(tmp_path / 'nope.so'))
    with pytest.raises(L.M3PError):
        L.load()


def test_header_is_plain_c(tmp_path):
    """include/m3p_hip.h is the contract a C / cgo / JNI binding compiles against: it must parse as C on its own (it once
    relied on a C++ translation unit having pulled in size_t)."""
    import shutil
    import subprocess
    gcc = shutil.which('gcc')
    if gcc is None:
        pytest.skip('no gcc')
    src = tmp_path / 'use.c'
    src.write_text('#include "m3p_hip.h"\nint probe(void) { M3PEpilogue ep = {0}; return (int)sizeof(ep) + M3P_EPI_BIAS_LSE; }\n')
    res = subprocess.run([gcc, '-std=c99', '-Wall', '-Werror', '-fsyntax-only', '-I', os.path.join(ROOT, 'include'), str(src)],
                         capture_output=True, text=True)
    assert res.returncode == 0, res.stderr
