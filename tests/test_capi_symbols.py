"""The C-ABI library loads without a GPU and exports every symbol include/sfm_fusion.h
declares (no compute calls here)."""
import ctypes
import os
import re

from sfm_amd import fusion

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "sfm_fusion.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(sfm_fusion_\w+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    names = declared_symbols()
    assert len(names) >= 17
    lib = ctypes.CDLL(fusion.LIB_PATH)
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/sfm_fusion.h but not exported"
    assert set(fusion.EXPORTS) == set(names)


def test_default_params_match_the_reference_constants():
    p = fusion.default_params()       # dense_reconstruction.hpp:44-56, .cpp:23-25,85,229,309
    assert (p.base_line, p.T_cov, p.voxel_size, p.T_dist, p.min_dist_filter) == (0.5, 0.5, 0.05, 0.5, 0.0)
    assert (p.dir_dot, p.depth_div, p.close_frac) == (0.30, 1000.0, 0.5)
    assert list(p.S) == [0.05, 0.05, 10.0]
    assert (p.table_capacity, p.max_points) == (0, 0)


def test_struct_sizes_match_the_header():
    assert ctypes.sizeof(fusion.FusionParams) == 11 * 8 + 2 * 8
    assert ctypes.sizeof(fusion.FusionStats) == 11 * 8 + 2 * 8 + 8 + 8 + 8 + 8 + 8 + 8 + 8
    assert fusion.SFM_RECORD_BYTES == 48


def test_version_string():
    assert b"gfx950" in fusion.load_library().sfm_fusion_version()


def test_no_product_import_of_the_oracle():
    # the product path must never route through the CPU oracle
    bad = re.compile(r"(^\s*(from|import)\s+oracle\b)|fusion_oracle|liboracle|oracle/|oracle\.py", re.M)
    for dirpath, _, files in os.walk(os.path.join(ROOT, "sfm_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".hpp", ".cpp")):
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                assert not bad.search(text), os.path.join(dirpath, f)
