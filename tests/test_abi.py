"""The C-ABI libraries load and export every symbol include/vo_hip.h and include/myslam_c.h declare.
No compute calls are made on the HIP library here (no GPU in this environment)."""
import os
import re

import pytest

from oracle import ORACLE_LIB
from rgbd_visualodometry_amd import capi, system

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared(header, prefix):
    txt = open(os.path.join(ROOT, "include", header)).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(%s[a-z_0-9]+)\s*\(" % prefix, txt)))


def test_header_symbol_lists_are_complete():
    assert declared("vo_hip.h", "vo_") == sorted(capi.SYMBOLS)
    assert declared("myslam_c.h", "myslam_") == sorted(system.SYMBOLS)


@pytest.mark.parametrize("path", [capi.HIP_LIB, ORACLE_LIB])
def test_vo_abi_exports(path):
    L = capi.load(path)                     # raises if the library or any declared symbol is missing
    assert L.backend in ("hip-gfx950", "cpu-oracle")
    p = L.default_params()
    assert (p.width, p.height, p.n_features, p.n_levels) == (640, 480, 500, 8)
    t = L.default_track_params()
    assert t.n_hyp == 100 and abs(t.huber_delta ** 2 - 7.815) < 1e-12 and t.passes == 2


@pytest.mark.parametrize("path", [system.HOST_LIB, ORACLE_LIB])
def test_host_abi_exports(path):
    lib = system._load(path)
    assert lib.myslam_backend_name().decode() in ("hip-gfx950", "cpu-oracle")


def test_product_path_fails_loudly_without_gpu():
    """The product library must not fall back to the CPU: without a HIP device context creation errors."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    L = capi.load(capi.HIP_LIB)
    with pytest.raises(capi.VoError):
        L.context(L.default_params())
    with pytest.raises(RuntimeError):
        system.VoSystem(system.HOST_LIB)


def test_product_sources_do_not_reference_the_oracle():
    bad = []
    for base in ("rgbd_visualodometry_amd/csrc", "rgbd_visualodometry_amd/host"):
        for dp, _, files in os.walk(os.path.join(ROOT, base)):
            for f in files:
                if f.endswith((".hip", ".h", ".cpp")):
                    txt = open(os.path.join(dp, f)).read()
                    if re.search(r'#include\s+"([^"]*/)?(o_[a-z]+\.h|oracle/[^"]*)"', txt):
                        bad.append(f)
    assert not bad
