"""CPU: the C-ABI library loads and exports every symbol include/gsplat_hip.h declares, the
header-derived ctypes prototypes are sane, and the product fails loudly (no fallback) when the
library is missing or when it is handed CPU tensors.  No compute calls are made here."""
import ctypes
import os
import re
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from gscodec_studio_amd import _backend as B

    protos = B.parse_header()
    assert len(protos) >= 24
    lib = ctypes.CDLL(B.LIB_PATH)
    for name in protos:
        assert hasattr(lib, name), f"{name} declared in include/gsplat_hip.h but not exported"
    # and nothing with the gs_ prefix is exported without being declared
    out = subprocess.check_output(["nm", "-D", "--defined-only", B.LIB_PATH], text=True)
    exported = set(re.findall(r"\bT (gs_\w+)", out))
    assert exported == set(protos), exported ^ set(protos)


def test_version_and_error_string():
    from gscodec_studio_amd import _backend as B

    L = B.lib()
    assert L.gs_version() == B.header_abi_version() == 5
    assert L.gs_header_hash() == B.header_hash()  # the library was compiled against THIS header
    assert isinstance(L.gs_last_error(), bytes)
    assert B.query("gs_sort_temp_bytes", 1000) >= 1000 * 12
    assert B.query("gs_cumsum_scratch_bytes", 10_000) >= 8
    from gscodec_studio_amd import _wrapper as W

    plan, sb = W._raster_plan(8160, 4_000_000, 3)
    assert sb > 4_000_000 // 256 * 4 * 256 * 4  # checkpoint planes every 256 entries
    import struct

    magic, n_tiles, n_isects, channels, seg, solo = struct.unpack_from("<IIIIii", plan, 0)
    assert (n_tiles, n_isects, channels, seg, solo) == (8160, 4_000_000, 3, 256, 2048)
    prev = W.set_raster_tuning(raster_seg=100, raster_solo_min=0)  # rounded up to a multiple of 64
    try:
        plan2, sb2 = W._raster_plan(8160, 4_000_000, 3)
        assert struct.unpack_from("<ii", plan2, 16) == (128, 0) and sb2 > sb
    finally:
        W.set_raster_tuning(**prev)


def test_stale_header_or_library_is_refused(tmp_path):
    """A header that does not match the library (other version, or same version but different text) must fail the import,
    not be called through shifted argument lists."""
    from gscodec_studio_amd import _backend as B

    src = open(B.HEADER_PATH).read()
    for name, text, expect in (("ver.h", re.sub(r"#define GS_ABI_VERSION \d+", "#define GS_ABI_VERSION 2", src), "ABI version mismatch"),
                               ("txt.h", src.replace("gs_stream_t stream);", "gs_stream_t  stream);", 1), "different gsplat_hip.h")):
        h = tmp_path / name
        h.write_text(text)
        code = "import gscodec_studio_amd._backend as B; B.lib()"
        env = dict(os.environ, GSPLAT_HIP_HEADER=str(h), PYTHONPATH=ROOT)
        r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True)
        assert r.returncode != 0 and expect in r.stderr, (name, r.stderr[-500:])


def test_argument_validation_happens_before_any_launch():
    """Status codes + messages for bad arguments (no GPU needed: the checks precede the launch)."""
    from gscodec_studio_amd import _backend as B

    with pytest.raises(RuntimeError, match="exactly one of covars"):
        B.call("gs_projection_fwd", 1, 1, 1, None, None, None, 1, 1, 10, 10, 0.3, 0.01, 1e10, 0.0, 0, 1, 1, 1, 1, None, None)
    with pytest.raises(RuntimeError, match="degree must be <= 4"):
        B.call("gs_sh_fwd", 1, 1, 36, 5, 1, 1, 0, None, 1, None)
    with pytest.raises(RuntimeError, match="unsupported number of colour channels"):
        B.call("gs_rasterize_fwd", 1, 1, 0, 600, None, None, None, None, None, None, None, 16, 16, 16, 1, 1, 1, None, 1, 1, 1, None, None, None, 0, None)
    with pytest.raises(RuntimeError, match="tile_size must be in"):
        B.call("gs_rasterize_fwd", 1, 1, 0, 3, None, None, None, None, None, None, None, 16, 16, 32, 1, 1, 1, None, 1, 1, 1, None, None, None, 0, None)
    with pytest.raises(RuntimeError, match="plan and scratch go together"):
        B.call("gs_rasterize_fwd", 1, 1, 0, 3, None, None, None, None, None, None, None, 16, 16, 16, 1, 1, 1, None, 1, 1, 1, None, 1, None, 0, None)
    with pytest.raises(RuntimeError, match="temp too small"):
        B.call("gs_sort_pairs_u64_i32", 10, 1, 1, 1, 1, 0, 40, None, 0, None)


def test_missing_library_fails_loudly():
    code = "import gscodec_studio_amd._backend as B; B.lib()"
    env = dict(os.environ, GSPLAT_HIP_LIB="/nonexistent/libgsplat_hip.so", PYTHONPATH=ROOT)
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True)
    assert r.returncode != 0 and "native library not found" in r.stderr and "no CPU fallback" in r.stderr.replace("There is no", "no")


def test_cpu_tensors_are_rejected_not_emulated():
    import gscodec_studio_amd as g
    from gscodec_studio_amd.compression_simulation import fake_quantize_ste

    N = 8
    means, quats, scales = torch.randn(N, 3), torch.randn(N, 4), torch.rand(N, 3)
    viewmats, Ks = torch.eye(4)[None], torch.eye(3)[None]
    with pytest.raises(RuntimeError, match="no CPU"):
        g.fully_fused_projection(means, None, quats, scales, viewmats, Ks, 32, 32)
    with pytest.raises(RuntimeError, match="no CPU"):
        g.spherical_harmonics(0, torch.randn(N, 3), torch.randn(N, 1, 3))
    with pytest.raises(RuntimeError, match="no CPU"):
        g.rasterization(means, quats, scales, torch.rand(N), torch.rand(N, 3), viewmats, Ks, 32, 32)
    with pytest.raises(RuntimeError, match="no CPU"):
        fake_quantize_ste(torch.randn(10), -1, 1, 8, "round")


def test_host_struct_mirrors_match_the_library_layout():
    """The two host structs that cross the boundary by pointer (gs_step, gs_quant_desc) are mirrored by hand in ctypes: the
    library reports its own sizeof / offsetof (gs_step_layout, gs_quant_desc_layout) and the mirrors must agree; a mirror that
    drifts is refused before a descriptor is ever handed over."""
    import ctypes

    from gscodec_studio_amd import _step
    from gscodec_studio_amd.compression_simulation import ops

    _step.check_layout()
    ops.check_desc_layout()
    assert int(_step.B.query("gs_step_layout", None, 0)) == 1 + len(_step._LAYOUT_FIELDS)

    class Drift(ctypes.Structure):  # a field too many in the middle: everything behind it moves
        _fields_ = _step._Step._fields_[:5] + [("extra", ctypes.c_uint64)] + _step._Step._fields_[5:]

    real = _step._Step
    _step._Step = Drift
    try:
        with pytest.raises(ImportError, match="struct layout"):
            _step.check_layout()
    finally:
        _step._Step = real


def test_product_never_imports_the_oracle():
    """The oracle is test infrastructure: nothing under gscodec_studio_amd/ may reference it."""
    pkg = os.path.join(ROOT, "gscodec_studio_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle", src, flags=re.M), os.path.join(dirpath, f)
                assert "gs_oracle" not in src, os.path.join(dirpath, f)
