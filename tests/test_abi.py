"""CPU-only: the C-ABI library builds, loads, and exports every symbol include/nafae_hip.h declares
(no compute calls -- there is no GPU here)."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from nafae_amd import build, _lib
    build.build()
    return _lib


def _declared():
    hdr = open(os.path.join(ROOT, "include", "nafae_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(nafae_[a-z0-9_]+)\s*\(", hdr)))


def test_every_declared_symbol_is_exported_and_bound(lib):
    names = _declared()
    assert len(names) >= 20
    l = lib.lib()
    for n in names:
        assert hasattr(l, n), "missing export: " + n
        assert n in lib.SIGNATURES, "missing ctypes signature: " + n
    assert sorted(lib.SIGNATURES) == names


def test_version_and_argument_errors(lib):
    assert lib.version().startswith("nafae_hip")
    l = lib.lib()
    # argument validation happens before any launch, so it is callable without a GPU
    assert l.nafae_gemm_nt(None, 4, None, 4, None, 4, None, 4, 4, 4, 1.0, 0, None) == -1
    assert l.nafae_loss_workspace_bytes(8, 8, 128, 16, 512) > 0
    assert l.nafae_loss_workspace_bytes(0, 8, 128, 16, 512) < 0


def test_ops_fail_loudly_without_gpu():
    import torch
    from nafae_amd import ops
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(ops.NafaeOpError):
        ops.gemm_nt(torch.zeros(4, 4), torch.zeros(4, 4))


def test_product_does_not_import_oracle():
    """The oracle is test infrastructure: nothing under nafae_amd/ may import it."""
    pkg = os.path.join(ROOT, "nafae_amd")
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith(".py"):
                src = open(os.path.join(dp, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f
                assert "/root/reference" not in src, f
