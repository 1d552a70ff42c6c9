"""The drop-in boundary: libdnsplat.so loads on a machine without a GPU and exports every entry point
include/dnsplat.h declares; the ctypes mirrors of the structs have the C layout; error strings work.
No compute entry point is called here (that needs the MI355X: tests/test_gpu_parity.py)."""
import ctypes
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "dnsplat.h")


def _declared_functions():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(dnsplat_[a-z0-9_]+)\s*\(", src)))


@pytest.fixture(scope="module")
def built(dns):
    return dns.build_library()


def test_library_exports_every_declared_symbol(dns, built):
    from dn_splatter_amd import _lib

    decl = _declared_functions()
    assert len(decl) >= 11
    L = ctypes.CDLL(str(built))
    missing = [f for f in decl if not hasattr(L, f)]
    assert not missing, missing
    assert sorted(_lib.EXPORTS) == decl, "the ctypes binding and the header disagree on the entry points"


def test_abi_version_and_strerror(dns, built):
    L = dns.load_library()
    assert L.dnsplat_abi_version() == 15
    msgs = {L.dnsplat_strerror(c).decode() for c in (0, -1, -2, -3, -4)}
    assert len(msgs) == 5 and "ok" in msgs
    assert "unknown" in L.dnsplat_strerror(-99).decode()


def test_workspace_query_is_host_only(dns, built):
    L = dns.load_library()
    small = L.dnsplat_bin_workspace_bytes(1000, 10_000, 256)
    big = L.dnsplat_bin_workspace_bytes(1_000_000, 30_000_000, 8160)
    assert 0 < small < big
    # 5 N-sized u32 arrays + 4 capacity-sized u32 arrays + histogram tables
    assert big >= 4 * (5 * 1_000_000 + 4 * 30_000_000)
    assert L.dnsplat_bin_workspace_bytes(-1, 0, 1) == 0


def test_invalid_arguments_return_codes_without_touching_the_gpu(dns, built):
    from dn_splatter_amd import _lib

    L = dns.load_library()
    assert L.dnsplat_project_fwd(None, None, None, None) == -1
    assert L.dnsplat_raster_fwd(None, None) == -1
    assert L.dnsplat_raster_bwd(None, None) == -1
    assert L.dnsplat_bin_prepare(None, None) == -1
    a = _lib.RasterArgs()
    a.tile_size, a.D, a.width, a.height = 8, 3, 16, 16
    assert L.dnsplat_raster_fwd(ctypes.byref(a), None) == -4          # only 16x16 tiles
    a.tile_size, a.D = 16, 9
    assert L.dnsplat_raster_fwd(ctypes.byref(a), None) == -4          # more than 8 channels
    a.D = 3
    assert L.dnsplat_raster_fwd(ctypes.byref(a), None) == -1          # null buffers
    assert L.dnsplat_pack_splats(-1, None, None, None, None, 0, None, None) == -1
    assert L.dnsplat_pack_splats(0, None, None, None, None, 0, None, None) == 0


def test_struct_layouts_match_the_c_compiler(dns, tmp_path):
    """sizeof/offsetof of every ABI struct as gcc sees the header == the ctypes mirror."""
    from dn_splatter_amd import _lib

    structs = {"dnsplat_scene": _lib.Scene, "dnsplat_camera": _lib.Camera, "dnsplat_proj_out": _lib.ProjOut,
               "dnsplat_bin_args": _lib.BinArgs, "dnsplat_raster_args": _lib.RasterArgs,
               "dnsplat_proj_grads": _lib.ProjGrads, "dnsplat_dn_post": _lib.DnPost, "dnsplat_dn_loss_args": _lib.DnLossArgs,
               "dnsplat_densify_args": _lib.DensifyArgs, "dnsplat_det_args": _lib.DetArgs}
    lines = ['#include <stdio.h>', '#include <stddef.h>', f'#include "{HEADER}"', 'int main(void){']
    for cname, cls in structs.items():
        lines.append(f'printf("{cname} %zu\\n", sizeof({cname}));')
        for fname, _ in cls._fields_:
            lines.append(f'printf("{cname}.{fname} %zu\\n", offsetof({cname}, {fname}));')
    lines.append('return 0;}')
    src = tmp_path / "layout.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "layout"
    subprocess.run(["gcc", str(src), "-o", str(exe)], check=True)
    got = dict(l.split() for l in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.splitlines())
    for cname, cls in structs.items():
        assert int(got[cname]) == ctypes.sizeof(cls), cname
        for fname, _ in cls._fields_:
            assert int(got[f"{cname}.{fname}"]) == getattr(cls, fname).offset, f"{cname}.{fname}"


def test_product_refuses_cpu_tensors_and_never_imports_the_oracle(dns):
    import torch

    from dn_splatter_amd import DnsplatError

    N = 8
    with pytest.raises(DnsplatError, match="no CPU fallback"):
        dns.rasterization(torch.zeros(N, 3), torch.ones(N, 4), torch.ones(N, 3), torch.ones(N), torch.ones(N, 3),
                          torch.eye(4)[None], torch.eye(3)[None], 16, 16)
    # static check: nothing under the package imports oracle/
    pkg = os.path.join(ROOT, "dn-splatter_amd")
    for fn in os.listdir(pkg):
        if fn.endswith(".py"):
            text = open(os.path.join(pkg, fn)).read()
            assert not re.search(r"^\s*(from|import)\s+oracle", text, flags=re.M), fn
    code = "import sys; sys.path.insert(0, %r); import dn_splatter_amd; assert not any(m == 'oracle' or m.startswith('oracle.') for m in sys.modules), 'oracle imported'" % ROOT
    subprocess.run([sys.executable, "-c", code], check=True)


def test_hand_placed_lds_loads_are_not_touched_before_their_wait():
    """raster_bwd.hip issues ds_read_b128 by hand and waits later (see row_issue/row_wait): the generated ISA
    must not read or copy the destination registers in between."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import check_asm_hazards

    assert check_asm_hazards.check() == 0


def test_integration_md_ctypes_stub_matches_the_binding(dns):
    """The ctypes stub INTEGRATION.md shows a maintainer (section C) is this ABI version's dnsplat_raster_args: same fields, same
    layout as the binding's mirror (which test_struct_layouts_match_gcc holds to the header) — a stale stub passes a short struct."""
    import ctypes
    import re

    from dn_splatter_amd import _lib

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    text = open(os.path.join(root, "INTEGRATION.md")).read()
    m = re.search(r"^class RasterArgs\(ctypes\.Structure\):.*?\]\s*(#[^\n]*)?\n\n", text, re.S | re.M)
    assert m, "stub not found"
    ns = {"ctypes": ctypes}
    exec(m.group(0), ns)
    stub = ns["RasterArgs"]
    assert [f[0] for f in stub._fields_] == [f[0] for f in _lib.RasterArgs._fields_]
    assert ctypes.sizeof(stub) == ctypes.sizeof(_lib.RasterArgs)
    for name, *_ in stub._fields_:
        assert getattr(stub, name).offset == getattr(_lib.RasterArgs, name).offset, name
    v = re.search(r"dnsplat_abi_version\(\) == (\d+)", text)
    assert v and int(v.group(1)) == _lib.ABI_VERSION


def test_integration_md_names_every_exported_symbol(dns):
    """INTEGRATION.md section C is the maintainer's map from the reference's call sites to the C ABI: every symbol include/dnsplat.h
    declares appears in it."""
    import os

    from dn_splatter_amd import _lib

    text = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "INTEGRATION.md")).read()
    assert [e for e in _lib.EXPORTS if e not in text] == []
