"""The C-ABI library builds for gfx950, loads without a GPU and exports every symbol include/dmm_match.h declares.
(No compute calls here: those are the -m gpu tests.)"""
import ctypes
import os
import re

from conftest import ROOT
from dmm_net_amd import _lib


def header_symbols():
    txt = open(os.path.join(ROOT, "include", "dmm_match.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(dmm_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    if not os.path.exists(_lib.LIB_PATH):
        _lib.build()
    L = ctypes.CDLL(_lib.LIB_PATH)
    syms = header_symbols()
    assert len(syms) >= 10
    for s in syms:
        assert hasattr(L, s), f"{s} declared in include/dmm_match.h but not exported"
    assert sorted(_lib.SYMBOLS) == syms, "dmm_net_amd/_lib.py binds a different symbol set than the header declares"


def test_load_and_status_strings():
    if not os.path.exists(_lib.LIB_PATH):
        _lib.build()
    L = _lib.load()
    assert L.dmm_abi_version() == 2
    assert L.dmm_status_string(0) == b"ok"
    assert b"gfx950" in L.dmm_build_info() and L.dmm_build_info().endswith(b"abi %d" % L.dmm_abi_version())
    assert L.dmm_workspace_bytes(4, 50, 10, 512) > 4 * (50 + 10) * 512 * 4
    assert L.dmm_workspace_bytes(0, 50, 10, 512) == 0


def test_argument_validation_without_gpu():
    """Bad arguments are rejected before anything touches the device."""
    if not os.path.exists(_lib.LIB_PATH):
        _lib.build()
    L = _lib.load()
    assert L.dmm_iou_counts(None, None, 0, 1, 4, 2, 16, 64, 16, 32, 16, None, None, None, None, None, None) == 1
    assert L.dmm_iou_counts(None, None, 0, -1, 4, 2, 16, 64, 16, 32, 16, None, None, None, None, None, None) == 1
    assert L.dmm_iou_counts(None, None, 0, 0, 4, 2, 16, 64, 16, 32, 16, None, None, None, None, None, None) == 0
    # M beyond the compiled solver envelope
    one = ctypes.c_void_p(8)
    assert L.dmm_relax_solve_f32(one, 1, 33, 40, None, None, 1, 1, 0.1, one, one, None, one, None) == 2
    assert L.dmm_relax_solve_f32(one, 1, 3, 257, None, None, 1, 1, 0.1, one, one, None, one, None) == 2


def test_training_entries_validate_their_arguments_without_gpu():
    """(5d) / (5e) / (1e): workspace sizes are positive and grow with the batch, null pointers / negative sizes answer
    DMM_ERR_BAD_ARG, tables outside the fast kernels' envelope answer DMM_ERR_UNSUPPORTED (the caller then takes the granular
    entries) -- all before anything touches the device."""
    if not os.path.exists(_lib.LIB_PATH):
        _lib.build()
    L = _lib.load()
    assert 0 < L.dmm_match_train_forward_workspace_bytes(1, 50, 5, 512) < L.dmm_match_train_forward_workspace_bytes(4, 50, 5, 512)
    assert L.dmm_match_train_forward_workspace_bytes(0, 50, 5, 512) == 0
    assert L.dmm_match_train_backward_workspace_bytes(1, 50, 5, 512, 10, 5) > 4 * (50 + 5) * 512
    # the solver's tape: R + 10 x 5 sweep records of 64 x 8 bytes + 10 sweep counts per frame; none for tables wider than a wave
    assert L.dmm_match_train_tape_bytes(1, 50, 5, 10, 5) >= 4 * 5 * 50 + 50 * 512 + 40
    assert L.dmm_match_train_tape_bytes(4, 50, 5, 10, 5) > 3 * L.dmm_match_train_tape_bytes(1, 50, 5, 10, 5)
    assert L.dmm_match_train_tape_bytes(1, 200, 20, 10, 5) == 0 and L.dmm_match_train_tape_bytes(1, 50, 5, 0, 5) == 0
    one = ctypes.c_void_p(8)
    base = lambda B, N, M: (one, one, one, 0, one, one, one, B, N, M, 64, 512, 3200, 64, 320, 64, 320, 64, None, None, 0.3,
                            10, 5, 0.1, 0, one, one, one, one, one, one, one, one, one, one, 1 << 30, None, 0, None, None)
    assert L.dmm_match_train_forward(*base(-1, 50, 5)) == 1
    assert L.dmm_match_train_forward(*base(0, 50, 5)) == 0                       # nothing to do
    assert L.dmm_match_train_forward(*base(1, 0, 5)) == 1
    assert L.dmm_match_train_forward(*base(1, 50, 33)) == 2                      # more than 32 templates
    assert L.dmm_match_train_forward(*base(1, 300, 5)) == 2                      # more than 256 solver columns
    args = list(base(1, 50, 5))
    args[0] = None
    assert L.dmm_match_train_forward(*args) == 1
    args = list(base(1, 50, 5))
    args[28] = None                                                              # cost_loss missing although targets are given
    assert L.dmm_match_train_forward(*args) == 1
    assert L.dmm_matching_loss_f32(None, None, None, None, 2, 50, 5, None, None, None, None, None) == 1
    assert L.dmm_matching_loss_f32(one, one, one, one, 2, 9000, 5, None, None, one, one, None) == 2
    assert L.dmm_launch_count() >= 0


def test_product_does_not_import_oracle():
    """The product package must never reach into oracle/ (CPU fallback would void parity)."""
    pkg = os.path.join(ROOT, "dmm_net_amd")
    for dp, _, fns in os.walk(pkg):
        for fn in fns:
            if fn.endswith((".py", ".hip", ".h")):
                src = open(os.path.join(dp, fn)).read()
                assert not re.search(r"^\s*(import|from)\s+oracle\b", src, flags=re.M), fn
                assert not re.search(r"#\s*include\s*[<\"][^>\"]*oracle", src), fn
                assert "libdmm_oracle" not in src and "dmmo_" not in src.replace("dmmo_check_div_by_const", ""), fn


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    """No HIP library -> DmmError with build instructions; nothing falls back to a CPU path."""
    import pytest
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "libdmm_match.so"))
    with pytest.raises(_lib.DmmError, match="no CPU fallback"):
        _lib.load()
    # and the tensor-level ops refuse CPU tensors outright
    import torch
    from dmm_net_amd import ops
    monkeypatch.undo()
    with pytest.raises(_lib.DmmError, match="no CPU fallback"):
        ops.iou_counts(torch.zeros(1, 2, 4, 4), torch.zeros(1, 1, 4, 4))
    from dmm_net_amd.match_model import MatchModel
    m = MatchModel({"matching": {"algo": "relax"}, "relax_max_iter": 2, "relax_proj_iter": 2,
                    "relax_learning_rate": 0.1, "score_weight": 0.3}, 1)
    with pytest.raises(_lib.DmmError):
        m(torch.zeros(2, 8), torch.zeros(2, 4, 4), [torch.zeros(1, 8)], torch.zeros(1, 4, 4), torch.zeros(2))


def test_workspace_sizes_cover_the_general_solver_outside_the_envelope():
    """Host-side size functions (no GPU): beyond 32 templates / 256 solver columns the fused forward's workspace also holds
    the general solver's state -- 9 tables of M x Pp floats per frame -- and dmm_relax_any_scratch_bytes says what the
    granular any-size entry needs."""
    L = _lib.load()
    B, D = 3, 64
    inside, wide_n, wide_m = L.dmm_workspace_bytes(B, 200, 20, D), L.dmm_workspace_bytes(B, 300, 20, D), \
        L.dmm_workspace_bytes(B, 20, 40, D)
    assert wide_n - inside > 9 * B * 20 * 300 * 4 and wide_m > 9 * B * 40 * 41 * 4
    for (N, M) in [(300, 40), (20, 50), (50, 10)]:
        Pp = max(N, M + 1)
        need = L.dmm_relax_any_scratch_bytes(B, N, M)
        assert 9 * B * M * Pp * 4 <= need <= 10 * B * M * Pp * 4 + B * 4096
    assert L.dmm_relax_any_scratch_bytes(0, 300, 40) == 0
    # argument validation of the any-size entry happens before any launch
    assert L.dmm_relax_match_any_f32(None, None, None, None, None, -1, 300, 40, None, None, 0.3, 2, 2, 0.1, 1, None, None, None,
                                     None, None, None, None, None, 0, None) == 1


def test_dispatch_options_go_through_the_abi_and_not_through_the_environment(monkeypatch):
    """VERDICT r3 hygiene: kernel choices / tuning values are integers set with dmm_set_option (include/dmm_match.h (0));
    the library contains no getenv call, so a stray variable cannot change what production dispatches."""
    L = _lib.load()
    names = re.findall(r"\b(DMM_OPT_[A-Z0-9_]+)\s*=\s*(\d+)", open(os.path.join(ROOT, "include", "dmm_match.h")).read())
    count = dict(names).pop("DMM_OPT_COUNT")
    assert int(count) == len(_lib.OPTIONS) == len(names) - 1
    for name, k in names:
        if name != "DMM_OPT_COUNT":
            assert _lib.OPTIONS[name[len("DMM_OPT_"):]] == int(k), name
    defaults = {k: _lib.get_option(k) for k in _lib.OPTIONS}
    assert defaults["COST_KERNEL"] == -1 and defaults["COST_TINY_FRAMES"] == 8 and defaults["FORCE_WIDE"] == 0
    monkeypatch.setenv("DMM_WIDE", "1")                           # what used to flip the dispatch
    monkeypatch.setenv("DMM_COST_KERNEL", "1")
    assert {k: _lib.get_option(k) for k in _lib.OPTIONS} == defaults
    inside = L.dmm_workspace_bytes(3, 200, 20, 64)
    with _lib.options(FORCE_WIDE=1, COST_KERNEL=1):
        assert _lib.get_option("FORCE_WIDE") == 1 and _lib.get_option("COST_KERNEL") == 1
        assert L.dmm_workspace_bytes(3, 200, 20, 64) > inside     # sizes and dispatch use ONE test
    assert {k: _lib.get_option(k) for k in _lib.OPTIONS} == defaults
    assert L.dmm_set_option(99, 1) == 1 and L.dmm_set_option(_lib.OPTIONS["COST_KERNEL"], 7) == 1
    assert L.dmm_set_option(_lib.OPTIONS["MIX_ALIGN"], 48) == 1 and L.dmm_get_option(99) == -2 ** 31
    _lib.set_option("MIX_WGS", 1000)
    assert L.dmm_reset_options() == 0 and _lib.get_option("MIX_WGS") == defaults["MIX_WGS"]
    for fn in os.listdir(_lib.CSRC):
        if fn.endswith((".hip", ".h")):
            assert "getenv" not in open(os.path.join(_lib.CSRC, fn)).read(), fn


def test_small_tables_share_one_staging_buffer():
    """``_lib.small_to_device_many``: several short host lists -> views of ONE buffer (one pinned copy on a GPU), each
    8-byte aligned, dtypes and values kept."""
    import torch
    specs = [([3, 1, 2], torch.int32), ([], torch.int32), ([2 ** 40 + 8, 7], torch.int64), ([0.5, -2.0, 1e-8], torch.float32)]
    got = _lib.small_to_device_many(specs, "cpu")
    assert [t.dtype for t in got] == [s[1] for s in specs]
    assert got[0].tolist() == [3, 1, 2] and got[1].numel() == 0 and got[2].tolist() == [2 ** 40 + 8, 7]
    assert got[3].tolist() == torch.tensor([0.5, -2.0, 1e-8]).tolist()
    base = got[0].untyped_storage().data_ptr()
    assert all(t.untyped_storage().data_ptr() == base for t in got if t.numel())
    assert all(t.data_ptr() % 8 == 0 for t in got if t.numel())
