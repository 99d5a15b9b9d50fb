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


def test_epilogue_struct_has_the_headers_size_and_field_offsets(tmp_path):
    """The M3PEpilogue the Python side (and INTEGRATION.md's stub) hands over is copied BY VALUE: its size and every field's
    offset must be the C compiler's for include/m3p_hip.h (round 5 review: the stub in INTEGRATION.md had stopped two fields short)."""
    import ctypes
    import re
    import shutil
    import subprocess
    from m3p_amd import lib as L
    gcc = shutil.which('gcc')
    if gcc is None:
        pytest.skip('no gcc')
    fields = [f[0] for f in L.Epilogue._fields_]
    body = ''.join('  printf("%s %%zu\\n", offsetof(M3PEpilogue, %s));\n' % (f, f) for f in fields)
    src = tmp_path / 'probe.c'
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "m3p_hip.h"\nint main(void) {\n  printf("sizeof %zu\\n", sizeof(M3PEpilogue));\n'
                   + body + '  return 0;\n}\n')
    exe = tmp_path / 'probe'
    res = subprocess.run([gcc, '-std=c99', '-I', os.path.join(ROOT, 'include'), str(src), '-o', str(exe)], capture_output=True, text=True)
    assert res.returncode == 0, res.stderr
    out = dict(line.split() for line in subprocess.run([str(exe)], capture_output=True, text=True).stdout.splitlines())
    assert int(out['sizeof']) == ctypes.sizeof(L.Epilogue)
    for f in fields:
        assert int(out[f]) == getattr(L.Epilogue, f).offset, f
    # the condensed stub of INTEGRATION.md names the same fields in the same order
    doc = open(os.path.join(ROOT, 'INTEGRATION.md')).read()
    stub = doc[doc.index('class M3PEpilogue(C.Structure)'):]
    stub = stub[:stub.index('lib.m3p_gemm_nt_bf16.restype')]
    assert re.findall(r'\("(\w+)", C\.', stub) == fields
    assert 'C.sizeof(M3PEpilogue) == %d' % ctypes.sizeof(L.Epilogue) in stub
