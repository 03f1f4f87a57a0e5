"""The C-ABI library loads and exports every function declared in include/ (no compute calls)."""
import re
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
DECL = re.compile(r"^\s*(?:const\s+)?(?:unsigned\s+)?(?:int|void|double|return_t|answer_t|char)\s*\*?\s*(\w+)\s*\(", re.M)


def declared_functions(header: Path):
    text = header.read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    names = set(DECL.findall(text))
    # macro-generated accessors of tree_qp_common.h
    for m in re.finditer(r"TQ_MAT_ACCESSORS\(\w+,\s*(\w+),\s*(\w+)\)", text):
        kind, name = m.groups()
        if kind != "KIND":
            names |= {f"tree_qp_in_set_{kind}_{name}_colmajor", f"tree_qp_in_get_{kind}_{name}_colmajor"}
    for m in re.finditer(r"TQ_VEC_ACCESSORS\((\w+),\s*\w+,\s*(\w+),\s*(\w+)\)", text):
        pfx, kind, name = m.groups()
        if pfx != "PFX":
            names |= {f"{pfx}_set_{kind}_{name}", f"{pfx}_get_{kind}_{name}"}
    return {n for n in names if not n.startswith("TQ_") and n not in ("defined",)}


HEADERS = sorted(p for p in (ROOT / "include").rglob("*.h"))


@pytest.mark.parametrize("header", HEADERS, ids=lambda p: str(p.relative_to(ROOT / "include")))
def test_every_declared_symbol_is_exported(capi, header):
    L = capi.lib()
    missing = [n for n in sorted(declared_functions(header)) if not hasattr(L, n)]
    assert not missing, f"{header.name}: not exported: {missing}"


def test_key_entry_points_present(capi):
    L = capi.lib()
    for n in ("treeqp_tdunes_solve", "treeqp_tdunes_create", "treeqp_tdunes_calculate_size",
              "treeqp_tdunes_set_dual_initialization", "tree_qp_in_fill_lti_data_diag_weights",
              "tree_qp_out_max_KKT_res", "tqgpu_create", "tqgpu_solve", "tqgpu_get_solution",
              "write_solution_to_txt", "timers_print", "blasfeo_print_tran_dvec", "calculate_number_of_nodes"):
        assert hasattr(L, n), n
    assert b"gfx950" in L.tqgpu_version()


def test_no_device_fails_loudly(capi):
    """On a box without a GPU the product must refuse to solve (no CPU fallback)."""
    if capi.device_count() > 0:
        pytest.skip("a HIP device is visible")
    from treeqp_amd import problems as P
    f = P.thesis_example()
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        capi.TqGpu(f.nk, f.nx, f.nu)
    qp = capi.TreeQp(f.nx, f.nu, f.nk).set_flat(f)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        capi.TdunesSolver(qp).solve()


def test_product_does_not_link_oracle():
    import subprocess
    from treeqp_amd import capi as c
    out = subprocess.run(["nm", "-D", str(c.library_path())], capture_output=True, text=True).stdout
    assert "oracle_" not in out


def test_reference_examples_link_unchanged():
    """Drop-in boundary: the reference's own drivers compile and link UNCHANGED against our headers
    and library (build container only: the sources do not travel)."""
    ref = Path("/root/reference")
    if not ref.exists():
        pytest.skip("/root/reference is not present on this box")
    from treeqp_amd import build
    exes = build.build_reference_dropins()
    names = {e.name for e in exes}
    assert {"spring_mass_tdunes", "thesis_example"} <= names
