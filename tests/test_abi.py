"""No-GPU checks of the drop-in boundary: libtrt_hip.so loads, exports every symbol that
include/trt_hip.h declares, and its pure-host helpers behave.  No compute call is made."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import support as T
from terminalraytracer_amd import hip


def _declared_symbols(header="trt_hip.h"):
    text = open(os.path.join(T.ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    names = re.findall(r"\b(project_scene|render_frame|trt_[a-z0-9_]+)\s*\(", text)
    return sorted(set(names))


def test_library_exports_every_declared_symbol():
    """include/trt_hip.h is the product boundary, include/trt_hip_diag.h the bench-top (counters, read-backs, self-tests, probes,
    test hooks): the library exports every symbol either declares, the binding knows exactly those, and nothing of the
    bench-top has crept back into the product header."""
    dll = hip.lib()
    product, diag = _declared_symbols(), _declared_symbols("trt_hip_diag.h")
    assert "project_scene" in product and "render_frame" in product and len(product) >= 20
    assert not set(product) & set(diag)
    assert not [n for n in product if re.search(r"selftest|probe|read_(path_tables|light_grid|counters|diagnostics|loop|sweep|shading)|allow_rccl|pool_words", n)], product
    for header, declared in (("trt_hip.h", product), ("trt_hip_diag.h", diag)):
        for name in declared:
            assert hasattr(dll, name), f"{name} declared in include/{header} but not exported"
    assert set(product) | set(diag) == set(hip.SYMBOLS), (set(product) | set(diag)) ^ set(hip.SYMBOLS)


def test_library_exports_every_host_side_symbol():
    """include/trt_host.h: camera orbit, PPM / cubemap loader, emitter, frame fingerprint -- host C in the same library"""
    from terminalraytracer_amd import host
    text = open(os.path.join(T.ROOT, "include", "trt_host.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    declared = sorted(set(re.findall(r"\b(trt_[a-z0-9_]+)\s*\(", text)))
    dll = host.lib()
    assert len(declared) >= 15
    for name in declared:
        assert hasattr(dll, name), f"{name} declared in include/trt_host.h but not exported"
    assert set(declared) == set(host.HOST_SYMBOLS), set(declared) ^ set(host.HOST_SYMBOLS)
    # the fingerprint is FNV-1a-64 as SURVEY 8c states it
    assert host.fnv1a64(np.frombuffer(b"", dtype=np.uint8)) == f"{1469598103934665603:016x}"
    h = 1469598103934665603  # SURVEY's offset (not the textbook 14695981039346656037): the goldens are recorded with it
    for byte in b"TerminalRayTracer":
        h = ((h ^ byte) * 1099511628211) % (1 << 64)
    assert host.fnv1a64(np.frombuffer(b"TerminalRayTracer", dtype=np.uint8)) == f"{h:016x}"


def test_version_and_error_strings():
    assert b"gfx950" in hip.lib().trt_version()
    assert isinstance(hip.lib().trt_last_error(), bytes)


@pytest.mark.parametrize("w,h,tile,world", [(1920, 1080, 8, 8), (3840, 2160, 8, 8), (67, 13, 4, 3), (5, 1, 8, 2),
                                             (16, 9, 1, 4), (160, 48, 48, 1)])
def test_rowsets_partition_the_frame(w, h, tile, world):
    lib = hip.lib()
    seen = []
    for rank in range(world):
        rs = hip.RowSet.shard(w, h, rank, world, tile)
        n = lib.trt_rowset_rows(C.byref(rs))
        rows = [lib.trt_rowset_frame_row(C.byref(rs), i) for i in range(n)]
        assert rows == sorted(rows) and all(0 <= r < h for r in rows)
        assert lib.trt_rowset_frame_row(C.byref(rs), n) == -1
        seen += rows
    assert sorted(seen) == list(range(h))  # every row exactly once


@pytest.mark.parametrize("w,h,tile,world", [(1920, 1080, 8, 8), (1920, 1080, 8, 3), (3840, 2160, 8, 8), (67, 13, 4, 3), (5, 2, 8, 4), (16, 9, 1, 4)])
def test_dist_assembly_map_puts_every_ranks_rows_in_frame_order(w, h, tile, world):
    """trt_dist's root gathers the ranks' compact shards rank-major (each padded to the largest shard) and one kernel copies
    row source_row[r] of that buffer to frame row r.  Here the ranks' shards are simulated on the CPU: shard rows carry their
    frame row number, and the map must put them back in order (what test_gpu_parity checks on the GPU for one rank)."""
    lib = hip.lib()
    source = (C.c_int * h)()
    max_rows = lib.trt_dist_source_rows(w, h, tile, world, source)
    assert max_rows > 0
    gathered = np.full(world * max_rows, -1, dtype=np.int64)
    for rank in range(world):
        rs = hip.RowSet.shard(w, h, rank, world, tile)
        n = lib.trt_rowset_rows(C.byref(rs))
        assert n <= max_rows
        for i in range(n):
            gathered[rank * max_rows + i] = lib.trt_rowset_frame_row(C.byref(rs), i)  # what rank `rank` sends as its row i
    frame = gathered[np.array(list(source))]
    assert np.array_equal(frame, np.arange(h))
    assert lib.trt_dist_source_rows(w, h, 0, world, source) < 0


def test_rowset_rejects_nonsense():
    lib = hip.lib()
    for bad in (hip.RowSet(0, 10, 1, 0, 1), hip.RowSet(10, 10, 0, 0, 1), hip.RowSet(10, 10, 1, -1, 1),
                hip.RowSet(10, 10, 1, 0, 0)):
        assert lib.trt_rowset_rows(C.byref(bad)) == 0
    assert lib.trt_rowset_rows(None) == 0


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    """There is no CPU or PyTorch fallback: without libtrt_hip.so the binding raises, it does not degrade."""
    hip.lib.cache_clear()
    monkeypatch.setattr(hip, "LIB_PATH", str(tmp_path / "libtrt_hip.so"))
    try:
        with pytest.raises(ImportError, match="no fallback"):
            hip.lib()
    finally:
        monkeypatch.undo()
        hip.lib.cache_clear()
        hip.lib()


def test_rccl_is_bound_by_name_and_may_be_overridden_for_tests(tmp_path):
    """csrc/trt_dist.hip binds its eight RCCL entry points at run time; TRT_RCCL_LIB names another library with the same entry
    points (the tests' stand-in for several ranks on one GPU, tests/rccl_stub.cpp).  It is a test hook: honoured only by a process
    that called trt_dist_allow_rccl_override(1) before its first use of RCCL -- and then strictly: a library that cannot be
    loaded is an error, not a reason to fall back to librccl -- and IGNORED by every other process (which then binds RCCL itself
    and says so).  Once RCCL is bound the switch refuses.  No GPU is needed to make an id."""
    import subprocess
    import sys
    stub = os.path.join(T.ROOT, "tests", "_build", "librccl_stub.so")
    if not os.path.exists(stub):
        rc = subprocess.run(["make", "-C", T.ROOT, "stub"], capture_output=True, text=True)
        if rc.returncode != 0 or not os.path.exists(stub):
            pytest.skip("the RCCL stand-in could not be built here")
    code = ("import sys; sys.path.insert(0, %r)\n"
            "from terminalraytracer_amd import hip\n"
            "if len(sys.argv) > 1:\n    hip.dist_allow_rccl_override(True)\n"
            "try:\n    print('ID', hip.dist_unique_id()[:15])\nexcept hip.TrtError as e:\n    print('ERR', e.code)\n"
            "print('LIB', hip.dist_rccl_library())\n"
            "try:\n    hip.dist_allow_rccl_override(True)\n    print('LATE ok')\nexcept hip.TrtError as e:\n    print('LATE', e.code)\n") % T.ROOT
    out = subprocess.run([sys.executable, "-c", code, "allow"], capture_output=True, text=True, env=dict(os.environ, TRT_RCCL_LIB=stub), timeout=120)
    assert "ID b'/trt_rccl_stub_" in out.stdout and "LIB STAND-IN (TRT_RCCL_LIB): " + stub in out.stdout, (out.stdout, out.stderr[-300:])
    assert "LATE -5" in out.stdout, out.stdout  # the switch refuses once RCCL is bound
    out = subprocess.run([sys.executable, "-c", code, "allow"], capture_output=True, text=True,
                         env=dict(os.environ, TRT_RCCL_LIB=str(tmp_path / "no_such_library.so")), timeout=120)
    assert "ERR -5" in out.stdout, (out.stdout, out.stderr[-300:])
    # without the switch the variable is ignored: the process binds RCCL itself (or fails for want of it), never the stand-in
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=dict(os.environ, TRT_RCCL_LIB=stub), timeout=120)
    assert "trt_rccl_stub_" not in out.stdout and "STAND-IN" not in out.stdout, (out.stdout, out.stderr[-300:])
    assert "LIB librccl" in out.stdout or "LIB /opt/rocm/lib/librccl" in out.stdout or "ERR -5" in out.stdout, out.stdout
