"""CPU-only: the C-ABI library loads and exports exactly what include/hyteg_hip.h declares
(no compute calls without a GPU), and the product never reaches into oracle/."""
import ctypes
import re
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent


def _declared_symbols():
    text = (ROOT / "include" / "hyteg_hip.h").read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(hyteg_hip_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_the_hot_path():
    syms = _declared_symbols()
    for needed in ("hyteg_hip_p1_apply_cell", "hyteg_hip_p1_jacobi_cell", "hyteg_hip_p1_sor_cell",
                   "hyteg_hip_p1_assign_cell", "hyteg_hip_p1_add_cell", "hyteg_hip_p1_mult_cell",
                   "hyteg_hip_p1_dot_cell", "hyteg_hip_p1_restrict_cell", "hyteg_hip_p1_prolongate_cell"):
        assert needed in syms


def test_library_exports_every_declared_symbol():
    from hyteg_amd import capi

    if not capi.lib_path().exists():
        import __graft_entry__ as g

        g.build()
    raw = ctypes.CDLL(str(capi.lib_path()))
    for sym in _declared_symbols():
        assert hasattr(raw, sym), f"{sym} declared in include/hyteg_hip.h but not exported"
    # binding table and header agree
    assert sorted(capi.SIGNATURES) == _declared_symbols()
    capi.lib()
    assert b"gfx950" in capi.lib().hyteg_hip_version()


def test_host_side_layout_helpers_match_the_oracle():
    from hyteg_amd import capi
    from oracle import p1_oracle as po

    for level in range(2, 9):
        assert capi.cell_size(level) == po.cell_size(level)
        assert capi.cell_inner_size(level) == po.cell_inner_size(level)
        assert capi.cell_width(level) == po.width(level)
    for (x, y, z) in [(0, 0, 0), (1, 1, 1), (3, 2, 1), (0, 0, 8), (1, 1, 5)]:
        assert capi.cell_index(3, x, y, z) == po.cell_index(3, x, y, z)


def test_argument_validation_happens_before_any_gpu_work():
    """Bad arguments are rejected on the host with EINVAL (the reference would WALBERLA_ABORT)."""
    from hyteg_amd import capi

    w = [0.0] * 15
    with pytest.raises(capi.HytegHipError, match="null pointer"):
        capi.p1_apply_cell(None, None, 4, w)
    with pytest.raises(capi.HytegHipError, match="level out of range"):
        capi.p1_apply_cell(4096, 8192, 1, w)
    with pytest.raises(capi.HytegHipError, match="level out of range"):
        capi.p1_apply_cell(4096, 8192, 12, w)
    with pytest.raises(capi.HytegHipError, match="alias"):
        capi.p1_apply_cell(4096, 4096, 4, w)
    with pytest.raises(capi.HytegHipError, match="bad update"):
        capi.p1_apply_cell(4096, 8192, 4, w, update=7)
    with pytest.raises(capi.HytegHipError, match="zero centre"):
        capi.p1_sor_cell(4096, 8192, 4, w, 1.0)
    with pytest.raises(capi.HytegHipError, match="nsrc"):
        capi.p1_assign_cell(4096, [1.0] * 5, [4096] * 5, 4)


def test_product_never_touches_the_oracle():
    """hyteg_amd/ and include/ must not import, link or mention oracle/ (tier rule 3)."""
    bad = []
    for p in list((ROOT / "hyteg_amd").rglob("*")) + list((ROOT / "include").rglob("*")):
        if p.is_file() and p.suffix in (".py", ".hip", ".hpp", ".h", ".cpp", ".c"):
            txt = p.read_text(errors="replace")
            if re.search(r"\boracle\b|p1_oracle|libhyteg_ref", txt):
                bad.append(str(p.relative_to(ROOT)))
    assert not bad, f"product files referencing the oracle: {bad}"
