"""The C-ABI library loads on a GPU-less box and exports every symbol include/dccn.h declares
(no compute calls here); host-side argument validation that needs no device."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from dl_ofdm_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    return _lib.load()


def header_symbols():
    src = open(os.path.join(ROOT, "include", "dccn.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(dccn_[a-z0-9_]+)\s*\(", src)))


def test_every_declared_symbol_is_exported_and_bound(lib):
    from dl_ofdm_amd import _lib
    names = header_symbols()
    assert len(names) >= 30
    for n in names:
        assert hasattr(lib, n), "libdccn.so does not export %s" % n
    assert sorted(_lib.SIGNATURES) == names, set(names) ^ set(_lib.SIGNATURES)


def test_struct_sizes_match_header(lib):
    from dl_ofdm_amd import _lib
    assert C.sizeof(_lib.Metrics) == 64
    assert C.sizeof(_lib.AdamState) == 16
    assert C.sizeof(_lib.AdamHParams) == 24
    assert C.sizeof(_lib.RxShape) == 24
    # 18 pointers/sizes + x_next + x_prenormalised (padded) + x_norm_next + (norm_slot, keep_dense_grad) + reg_uniform_dense (padded)
    assert C.sizeof(_lib.RxBuffers) == 26 * 8          # ... + x_next_ready + gen_next + tuning
    assert C.sizeof(_lib.GenStatic) == 200             # (static_assert'ed against the C struct in csrc/dccn_abi.hip)
    assert C.sizeof(_lib.GenProfile) == 32



def test_host_side_queries(lib):
    from dl_ofdm_amd import _lib
    assert lib.dccn_version() >= 100
    assert lib.dccn_strerror(0) == b"ok" and b"workspace" in lib.dccn_strerror(-2)
    assert [lib.dccn_tail_param_count(b) for b in (1, 2, 3, 4)] == [16, 40, 90, 200]
    assert lib.dccn_tail_param_count(5) < 0
    sh = _lib.RxShape(1170, 7, 80, 64, 320, 2)
    offs = (C.c_longlong * 6)()
    assert lib.dccn_rx_param_offsets(C.byref(sh), offs) == 0
    assert list(offs) == [0, 10240, 10368, 583808, 584448, 584488]        # 584 488 live params
    assert lib.dccn_rx_workspace_size(C.byref(sh), 1) > lib.dccn_rx_workspace_size(C.byref(sh), 0) > 0
    bad = _lib.RxShape(1170, 7, 80, 64, 320, 5)
    assert lib.dccn_rx_param_offsets(C.byref(bad), offs) == -1
    assert lib.dccn_rx_workspace_size(C.byref(bad), 1) == 0
    # argument validation happens before any device work
    assert lib.dccn_dense_fwd(None, None, None, None, 4, 4, 4, None) == -1
    assert lib.dccn_cconv_gemm_fwd(None, None, None, None, 0, 80, 64, None) == -1


def test_eq_workspace_tensor_lookup(lib):
    """dccn_eq_workspace_tensor: a host-side query (no device work): named intermediates of the fused equaliser step lie
    inside the sized workspace on 256-byte boundaries; unknown names and training-only tensors of an evaluation workspace
    are refused."""
    from dl_ofdm_amd import _lib
    sh = _lib.EqShape(12, 7, 64, 16, 1, 64, 320, 2, 16, 8)
    off, cnt = C.c_size_t(), C.c_size_t()
    assert lib.dccn_eq_workspace_tensor(C.byref(sh), 1, b"y", C.byref(off), C.byref(cnt)) == 0 and cnt.value == 12 * 7 * 128
    total = lib.dccn_eq_workspace_size(C.byref(sh), 1)
    assert off.value % 256 == 0 and off.value + 4 * cnt.value <= total
    seen = set()
    for name in (b"x_norm", b"ln", b"t1", b"d1", b"d2", b"d3", b"d4", b"eq", b"corr", b"cat", b"dz", b"dout", b"deqc", b"dcorc",
                 b"deq", b"dcorr", b"dy", b"dh", b"dd4", b"dd3", b"dd2", b"dflat", b"dt1"):
        assert lib.dccn_eq_workspace_tensor(C.byref(sh), 1, name, C.byref(off), C.byref(cnt)) == 0, name
        assert off.value % 256 == 0 and cnt.value > 0 and off.value + 4 * cnt.value <= total and off.value not in seen
        seen.add(off.value)
    assert lib.dccn_eq_workspace_tensor(C.byref(sh), 0, b"dh", C.byref(off), C.byref(cnt)) == -1          # training only
    assert lib.dccn_eq_workspace_tensor(C.byref(sh), 0, b"y", C.byref(off), C.byref(cnt)) == 0
    assert lib.dccn_eq_workspace_tensor(C.byref(sh), 1, b"nope", C.byref(off), C.byref(cnt)) == -1
    assert lib.dccn_eq_workspace_tensor(C.byref(sh), 1, None, C.byref(off), C.byref(cnt)) == -1


def test_ops_refuse_cpu_tensors():
    import torch
    from dl_ofdm_amd import _lib, ops
    with pytest.raises(_lib.DccnError):
        ops.batch_moment_norm(torch.zeros(4, 7, 80, 2))
    with pytest.raises(_lib.DccnError):
        ops.dense(torch.zeros(4, 8), torch.zeros(8, 8), None)
    from dl_ofdm_amd.engine import RxDims, RxEngine
    with pytest.raises(_lib.DccnError):
        RxEngine(RxDims(7, 80, 64, 320, 2), 4, device="cpu")


def test_bench_reports_traffic_only_for_the_build_it_was_measured_on(tmp_path):
    """bench.traffic_for: the PMC counters kept in profiles/pmc_traffic.json are stamped with the id of the library they were
    collected on (dccn_build_id); for any other build the bench line carries null + traffic_stale instead of an old number."""
    import json
    import sys
    ROOT_ = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if ROOT_ not in sys.path:
        sys.path.insert(0, ROOT_)
    import bench
    from dl_ofdm_amd import _lib
    bid = _lib.load().dccn_build_id().decode()
    assert len(bid) == 16 and int(bid, 16) >= 0
    f = tmp_path / "pmc_traffic.json"
    f.write_text(json.dumps({"c2": {"rx_backward": 7.0e7}, "build_id": bid}))
    assert bench.traffic_for(str(f), "c2", "rx_backward", bid) == (7.0e7, False, bid)
    f.write_text(json.dumps({"c2": {"rx_backward": 7.0e7}, "build_id": "0123456789abcdef"}))
    assert bench.traffic_for(str(f), "c2", "rx_backward", bid) == (None, True, "0123456789abcdef")
    f.write_text(json.dumps({"c2": {"rx_backward": 7.0e7}}))                      # a file from before the stamp existed
    assert bench.traffic_for(str(f), "c2", "rx_backward", bid) == (None, True, None)
    assert bench.traffic_for(str(tmp_path / "missing.json"), "c2", "rx_backward", bid) == (None, False, None)
