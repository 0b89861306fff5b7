# coding=utf-8
"""The C-ABI library builds, loads without a GPU, and exports every symbol include/duet_ef.h declares."""
import ctypes
import os
import re

import pytest

from duet_amd import _lib

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions(header='duet_ef.h'):
    with open(os.path.join(REPO, 'include', header)) as f:
        text = re.sub(r'/\*.*?\*/', '', f.read(), flags=re.S)
    return sorted(set(re.findall(r'\b(duet_[a-z_0-9]+)\s*\(', text)))


def test_header_and_binding_agree():
    assert declared_functions() == sorted(_lib.EXPORTS)


def test_library_exports_every_declared_symbol():
    import __graft_entry__
    __graft_entry__.build()
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared_functions():
        assert hasattr(lib, name), name
    lib.duet_abi_version.restype = ctypes.c_int
    assert lib.duet_abi_version() == 1


def exported_duet_symbols(path):
    import subprocess
    out = subprocess.check_output(['nm', '-D', '--defined-only', path]).decode()
    return sorted(set(l.split()[-1] for l in out.splitlines() if l.split() and l.split()[-1].startswith('duet_')))


def test_library_exports_nothing_but_the_declared_symbols():
    """-fvisibility=hidden + DUET_API: internal cross-unit helpers (duet_ef_upload, duet_ef_run_planned_on_device, ...)
    are not part of the ABI and must not be visible."""
    import __graft_entry__
    from duet_amd import native
    __graft_entry__.build()
    assert exported_duet_symbols(_lib.LIB_PATH) == declared_functions()
    assert exported_duet_symbols(native.LIB_PATH) == declared_functions('duet_ingest.h')


def test_ingest_library_exports_every_declared_symbol():
    from duet_amd import native
    import __graft_entry__
    __graft_entry__.build()
    assert declared_functions('duet_ingest.h') == sorted(native.EXPORTS)
    lib = ctypes.CDLL(native.LIB_PATH)
    for name in declared_functions('duet_ingest.h'):
        assert hasattr(lib, name), name


def test_struct_layout_matches_header():
    # 4 x u32, 9 pointers, 2 x u32 -> 16 + 72 + 8 = 96 bytes on LP64
    assert ctypes.sizeof(_lib.EfProblem) == 96
    assert _lib.EfProblem.cand_ctg_off.offset == 16
    assert _lib.EfProblem.svlen_thres.offset == 88
    assert ctypes.sizeof(_lib.EfStats) == 8 + 4 + 4 + 12 + 4


def test_no_gpu_means_loud_failure():
    """Without a device the product path must raise, not fall back to anything."""
    import torch
    if torch.cuda.is_available():
        pytest.skip('a GPU is present')
    with pytest.raises(_lib.DuetLibraryError):
        _lib.Context(0)
    from duet_amd import engine
    from tests import soa_fuzz
    with pytest.raises(_lib.DuetLibraryError):
        engine.run_ef(soa_fuzz.random_soa(1), 50, 2)


def test_product_never_imports_oracle():
    bad = []
    for root, _, files in os.walk(os.path.join(REPO, 'duet_amd')):
        for n in files:
            if n.endswith('.py'):
                with open(os.path.join(root, n)) as f:
                    src = f.read()
                if re.search(r'^\s*(from|import)\s+oracle\b', src, flags=re.M):
                    bad.append(n)
    assert not bad, bad
