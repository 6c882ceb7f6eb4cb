"""CPU checks of the drop-in boundary: the C-ABI library loads and exports every symbol include/*.h declares
(no compute call without a GPU), and the product path fails loudly when the library is missing."""
import ctypes
import os
import re
import subprocess

import pytest

ROOT = os.path.join(os.path.dirname(__file__), "..")


def _declared(header):
    src = open(os.path.join(ROOT, "include", header)).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(dd_[a-z0-9_]+)\s*\(", src)))


def test_library_builds_and_exports_every_declared_symbol():
    import __graft_entry__ as g
    g.build()
    from distdiff_amd import _lib
    lib = _lib.lib()
    names = _declared("distdiff_hip.h") + _declared("distdiff_hip_ops.h")
    assert len(names) > 40
    for n in names:
        assert hasattr(lib, n), "symbol %s declared in include/ but not exported" % n
    out = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True).stdout
    exported = set(re.findall(r" T (dd_[a-z0-9_]+)", out))
    assert set(names) <= exported
    # and the Python side binds exactly the declared surface
    assert set(_lib.OPS_SYMBOLS) == set(_declared("distdiff_hip_ops.h"))
    assert set(_lib.ENGINE_SYMBOLS) == set(_declared("distdiff_hip.h"))


def test_struct_mirrors_match_c_layout():
    """ctypes mirrors must have the size the C compiler gives the POD parameter blocks."""
    src = r'''
    #include <stdio.h>
    #include "distdiff_amd/csrc/kernels.h"
    #include "include/distdiff_hip.h"
    int main(){ printf("%zu %zu %zu %zu %zu %zu %zu %zu\n", sizeof(ConvGemmParams), sizeof(GroupNormParams), sizeof(LayerNormParams),
                       sizeof(AttnParams), sizeof(dd_config), sizeof(dd_sampler_params), sizeof(dd_expand_args), sizeof(ConvF32Params));
                return 0; }
    '''
    exe = "/tmp/dd_sizes"
    cpp = "/tmp/dd_sizes.cpp"
    open(cpp, "w").write(src)
    r = subprocess.run(["/opt/rocm/bin/hipcc", "-x", "hip", "--offload-arch=gfx950", "-I", os.path.abspath(ROOT), "-o", exe, cpp], text=True,
                       capture_output=True, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    sizes = [int(x) for x in subprocess.run([exe], capture_output=True, text=True).stdout.split()]
    from distdiff_amd import _lib, engine
    got = [ctypes.sizeof(x) for x in (_lib.ConvGemmParams, _lib.GroupNormParams, _lib.LayerNormParams, _lib.AttnParams,
                                      engine.DDConfig, engine.DDSamplerParams, engine.DDExpandArgs, _lib.ConvF32Params)]
    assert got == sizes, (got, sizes)


def test_missing_library_fails_loudly(monkeypatch):
    from distdiff_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libdistdiff_hip.so")
    with pytest.raises(_lib.DistDiffLibraryError):
        _lib.lib()


def test_weight_packing_host_side():
    """dd_pack_conv_weight is pure host code: forward and dgrad packing agree with a numpy restatement."""
    import numpy as np
    import torch
    from distdiff_amd import _lib
    lib = _lib.lib()
    w = torch.randn(5, 3, 3, 3)
    out4 = (ctypes.c_int * 4)()
    lib.dd_pack_conv_weight(ctypes.c_void_p(w.data_ptr()), 5, 3, 3, 3, 1, 0, 0, None, None, out4)
    N, K, cin, nt = list(out4)
    assert (N, K, cin, nt) == (5, 128, 8, 9)
    wp = np.zeros((N, K), np.uint16)
    tt = np.zeros(nt, np.int32)
    lib.dd_pack_conv_weight(ctypes.c_void_p(w.data_ptr()), 5, 3, 3, 3, 1, 0, 0, wp.ctypes.data_as(ctypes.c_void_p),
                            tt.ctypes.data_as(ctypes.c_void_p), out4)
    got = torch.from_numpy(wp.view(np.int16)).view(torch.bfloat16).float().reshape(5, 128)[:, :72].reshape(5, 9, 8)[:, :, :3]
    ref = w.to(torch.bfloat16).float().permute(0, 2, 3, 1).reshape(5, 9, 3)
    assert torch.equal(got, ref)
    assert [((t >> 6) & 63) - 32 for t in tt] == [-1, -1, -1, 0, 0, 0, 1, 1, 1]
    assert [(t & 63) - 32 for t in tt] == [-1, 0, 1] * 3
    # dgrad packing: [Cin][tap][Cout] with flipped offsets
    lib.dd_pack_conv_weight(ctypes.c_void_p(w.data_ptr()), 5, 3, 3, 3, 1, 1, 0, None, None, out4)
    assert list(out4) == [3, 128, 8, 9]
    wp = np.zeros((3, 128), np.uint16)
    lib.dd_pack_conv_weight(ctypes.c_void_p(w.data_ptr()), 5, 3, 3, 3, 1, 1, 0, wp.ctypes.data_as(ctypes.c_void_p),
                            tt.ctypes.data_as(ctypes.c_void_p), out4)
    got = torch.from_numpy(wp.view(np.int16)).view(torch.bfloat16).float().reshape(3, 128)[:, :72].reshape(3, 9, 8)[:, :, :5]
    ref = w.to(torch.bfloat16).float().permute(1, 2, 3, 0).reshape(3, 9, 5)
    assert torch.equal(got, ref)
    assert [((t >> 6) & 63) - 32 for t in tt] == [1, 1, 1, 0, 0, 0, -1, -1, -1]


def test_abi_version_is_checked():
    """include/distdiff_hip.h DD_ABI_VERSION == what the library reports == what the ctypes mirrors speak; dd_create refuses a config block
    whose abi_version field is anything else (a caller compiled against an older header passes 0 or garbage there).  No GPU call."""
    from distdiff_amd import _lib, engine
    lib = _lib.lib()
    hdr = open(os.path.join(ROOT, "include", "distdiff_hip.h")).read()
    ver = int(re.search(r"#define\s+DD_ABI_VERSION\s+(\d+)", hdr).group(1))
    assert lib.dd_abi_version() == ver == _lib.ABI_VERSION == engine.DD_ABI_VERSION
    cfg = engine.DDConfig()
    cfg.max_batch = 1
    h = ctypes.c_void_p()
    assert lib.dd_create(ctypes.byref(cfg), ctypes.byref(h)) == -1 and not h.value          # DD_ERR_ARG: abi_version 0
    cfg.abi_version = ver
    assert lib.dd_create(ctypes.byref(cfg), ctypes.byref(h)) == 0 and h.value
    lib.dd_destroy(h)
