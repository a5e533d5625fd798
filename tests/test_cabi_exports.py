"""The C-ABI library loads and exports every symbol include/tchgeo.h declares (no GPU needed)."""
import ctypes as C
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "tch-geometric_amd", "lib", "libtchgeo_hip.so")
HEADER = os.path.join(ROOT, "include", "tchgeo.h")


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(LIB):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "tch-geometric_amd"), "-s"])
    return C.CDLL(LIB)


def declared_symbols():
    src = open(HEADER).read()
    return sorted(set(re.findall(r"TG_API\s+[\w\s\*]+?\b(tg_\w+)\s*\(", src)))


def test_every_declared_symbol_is_exported(lib):
    names = declared_symbols()
    assert len(names) >= 9 and "tg_ns_homo_batched" in names
    for n in names:
        assert hasattr(lib, n), "missing export " + n


def test_python_binding_lists_the_same_exports(lib):
    import importlib.util
    spec = importlib.util.spec_from_file_location(
        "_cabi_probe", os.path.join(ROOT, "tch-geometric_amd", "tch_geometric", "_cabi.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    assert sorted(mod.EXPORTS) == declared_symbols()


def test_version_and_capacity(lib):
    lib.tg_version.restype = C.c_char_p
    assert lib.tg_version().startswith(b"tchgeo-gfx950")
    cn, ce = C.c_int64(), C.c_int64()
    fan = (C.c_int64 * 2)(15, 10)
    assert lib.tg_ns_homo_capacity(C.c_int64(1024), fan, 2, C.byref(cn), C.byref(ce)) == 0
    assert (cn.value, ce.value) == (1024 + 15360 + 153600, 15360 + 153600)   # S <= 168 960
    fan0 = (C.c_int64 * 2)(15, 0)
    assert lib.tg_ns_homo_capacity(C.c_int64(4), fan0, 2, C.byref(cn), C.byref(ce)) == 1   # TG_ERR_INVALID


def test_argument_errors_do_not_touch_the_gpu(lib):
    lib.tg_last_error.restype = C.c_char_p
    assert lib.tg_ns_homo_batched(None, None, C.c_int64(1), C.c_int64(1), None, 0, None, None, None, None) == 1
    assert b"null graph" in lib.tg_last_error()
    assert lib.tg_rmat_edges(0, C.c_int64(4), C.c_uint64(0), None, None, None) == 1


def test_header_is_plain_c(tmp_path):
    """the boundary must be bindable from C / cgo / Rust bindgen: no C++ or HIP types in include/tchgeo.h"""
    src = tmp_path / "hdr.c"
    src.write_text('#include "tchgeo.h"\nint main(void) { tg_graph g; tg_rng r; (void)g; (void)r; return TG_OK; }\n')
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", os.path.dirname(HEADER),
                           "-fsyntax-only", str(src)])
    text = open(HEADER).read()
    assert "hipStream_t stream" not in text and "at::" not in text and "torch" not in text.replace("no torch", "")


def test_new_entry_points_reject_bad_arguments(lib):
    lib.tg_last_error.restype = C.c_char_p
    st = C.c_int32(0)
    # negative sizes / short stride are refused before any launch
    assert lib.tg_gather_rows(None, C.c_int64(4), C.c_int64(16), C.c_int64(8), None, C.c_int64(1), None, C.byref(st), None) == 1
    assert b"stride" in lib.tg_last_error()
    assert lib.tg_gather_rows(None, C.c_int64(0), C.c_int64(16), C.c_int64(16), None, C.c_int64(0), None, None, None) == 0
    nbytes = C.c_int64(-1)
    assert lib.tg_biased_walk_workspace_bytes(C.c_int64(10), C.c_int64(100), C.c_int32(1), C.byref(nbytes)) == 0
    assert nbytes.value == 0                               # rows of <= 1024 edges sort in LDS
    assert lib.tg_biased_walk_workspace_bytes(C.c_int64(10), C.c_int64(5000), C.c_int32(1), C.byref(nbytes)) == 0
    assert nbytes.value == 3 * 4 * 2 * 5000 * 8            # 3 workgroups x 4 wavefronts x two radix buffers of 5000 keys
    assert lib.tg_biased_walk_workspace_bytes(C.c_int64(10), C.c_int64(5000), C.c_int32(2), C.byref(nbytes)) == 0
    assert nbytes.value == 0
    assert lib.tg_biased_tempo_random_walk(None, None, None, None, None, C.c_int64(1), C.c_int64(2), C.c_int32(7),
                                           C.c_int32(1), C.c_int64(1), C.c_int64(0), None, None, None, None, None,
                                           C.c_int64(0), None) == 1


def test_round2_entry_points_reject_bad_arguments(lib):
    """the window-ordered launch, the partitioned protocol and the heterogeneous step helpers refuse bad sizes / null
    buffers before any launch (no GPU needed)"""
    lib.tg_last_error.restype = C.c_char_p
    nbytes = C.c_int64(-1)
    fan = (C.c_int64 * 2)(15, 10)
    assert lib.tg_ns_homo_workspace_bytes(C.c_int64(4096), C.c_int64(1024), fan, C.c_int32(2), C.byref(nbytes)) == 0
    assert nbytes.value > 4096 * 15360 * 16 * 2            # room for the widest hop's items, unsorted and sorted
    bad = (C.c_int64 * 2)(15, 300)
    assert lib.tg_ns_homo_workspace_bytes(C.c_int64(8), C.c_int64(8), bad, C.c_int32(2), C.byref(nbytes)) == 1
    assert b"fanout" in lib.tg_last_error()
    assert lib.tg_ns_homo_workspace_bytes(C.c_int64(-1), C.c_int64(8), fan, C.c_int32(2), C.byref(nbytes)) == 1
    # tg_ns_homo_batched_ws: a workspace size without a workspace, an unknown form
    assert lib.tg_ns_homo_batched_ws(None, None, C.c_int64(1), C.c_int64(1), fan, C.c_int32(2), None, None, None, None,
                                     C.c_int64(64), C.c_int32(0), None) == 1
    assert lib.tg_ns_homo_batched_ws(None, None, C.c_int64(1), C.c_int64(1), fan, C.c_int32(2), None, None, None, None,
                                     C.c_int64(0), C.c_int32(9), None) == 1
    assert b"mode" in lib.tg_last_error()
    # partitioned protocol
    assert lib.tg_part_workspace_bytes(C.c_int64(4), C.c_int64(1 << 20), C.c_int32(65), C.byref(nbytes)) == 1
    assert lib.tg_part_workspace_bytes(C.c_int64(4), C.c_int64(1 << 33), C.c_int32(8), C.byref(nbytes)) == 1
    assert lib.tg_part_workspace_bytes(C.c_int64(4), C.c_int64(1 << 20), C.c_int32(8), C.byref(nbytes)) == 0
    assert nbytes.value > (1 << 20) * (16 + 8 + 4 + 8)     # requests, their states, positions, reply offsets
    assert lib.tg_part_requests(None, C.c_int64(1), C.c_int64(1), C.c_int64(1), C.c_int32(1), None, None, None, None,
                                None) == 1
    assert lib.tg_part_emit(None, C.c_int64(1), C.c_int64(1), C.c_int64(1), C.c_int64(1), C.c_int32(1), C.c_int32(5),
                            C.c_int32(0), C.c_int32(2), None, None, None, None, C.c_int32(2), None) == 1
    assert lib.tg_part_pack(None, None, None, C.c_int64(0), C.c_int64(0), C.c_int32(1), None, None, None, C.c_int32(2),
                            None, None) == 1
    assert lib.tg_part_sample(None, C.c_int64(0), C.c_int64(0), None, None, C.c_int64(0), C.c_int32(1), None, None,
                              C.c_int32(4), C.c_int32(0), C.c_uint64(0), None, None, None, C.c_int32(1), None) == 1
    # an unknown reply format is refused by pack and emit alike
    assert lib.tg_part_emit(None, C.c_int64(1), C.c_int64(1), C.c_int64(1), C.c_int64(1), C.c_int32(1), C.c_int32(5),
                            C.c_int32(0), C.c_int32(2), None, None, None, None, C.c_int32(7), None) == 1
    # heterogeneous step helpers
    words = C.c_int64(0)
    assert lib.tg_het_meta_words(C.c_int32(3), C.c_int32(5), C.c_int32(2), C.byref(words)) == 0
    assert words.value == 3 * 3 + 5 + 5 * 2 * 3 + (3 + 2 * 5)   # len, fbeg, fend | ne | layer offsets | snapshots
    assert lib.tg_het_meta_words(C.c_int32(0), C.c_int32(5), C.c_int32(2), C.byref(words)) == 1
    assert lib.tg_het_step_begin(None, None, None, C.c_int32(3), C.c_int32(5), C.c_int32(2), C.c_int32(0), C.c_int32(0),
                                 C.c_int32(0), C.c_int32(0), C.c_int64(8), None, None, None, None) == 1
    assert lib.tg_het_hop_end(None, C.c_int32(3), C.c_int32(5), C.c_int32(2), None) == 1
    # all relations of a hop at once: no entries, too many segments
    assert lib.tg_het_hop_begin_all(None, C.c_int32(0), None, C.c_int32(3), C.c_int32(5), C.c_int32(2), C.c_int64(0), None,
                                    None, None, None, None) == 1
    assert lib.tg_ns_hop_segments(None, C.c_int32(9), None, None, None, None, None, None, None, None, C.c_int64(0),
                                  C.c_int64(1), None) == 1
    assert lib.tg_het_hop_end_all(None, C.c_int32(1), None, None, C.c_int64(0), C.c_int64(0), None, None, C.c_int32(3),
                                  C.c_int32(5), C.c_int32(2), C.c_int32(0), C.c_int32(1), None, None) == 1
