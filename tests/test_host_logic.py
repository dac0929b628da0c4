"""Host-side logic that needs no GPU."""
import numpy as np

import oracle
from dmm_net_amd import ops, synth


def test_div_by_const_is_ieee_division():
    # the solver divides by the constant row / column counts with q = fma(fma(-a*r, b, a), r, a*r), r = RN(1/b);
    # it must reproduce IEEE a / b for every b the kernels accept (DMM_MAX_PROPOSALS = 256)
    assert oracle.check_div_by_const(256, 200000) == 0


def test_padded_width_rule():
    # match_model.py:109-113: pad to O+1 columns when P <= O
    assert ops.padded_width(50, 10) == 50
    assert ops.padded_width(3, 5) == 6
    assert ops.padded_width(5, 5) == 6
    assert ops.padded_width(6, 5) == 6
    assert ops.padded_width(1, 1) == 2


def test_synth_is_deterministic_and_structured():
    a = synth.make_config_frame(1, kind="structured", with_targets=True)
    b = synth.make_config_frame(1, kind="structured", with_targets=True)
    assert a.checksum() == b.checksum()
    assert a.proposed_mask.dtype == np.float32 and a.proposed_mask.shape == (8, 64, 64)
    assert set(np.unique(a.targets)) <= {0.0, 1.0}
    # the planted assignment is what the oracle recovers
    o = oracle.match_forward(a.proposed_mask, a.mask_last_occurence, a.proposed_feature, a.template_feature,
                             a.proposal_score, max_iter=20, proj_iter=5, is_test=1, want_outmask=False)
    assert np.array_equal(o["R"].argmax(1), a.perm)


def test_small_host_helpers_on_cpu():
    """_lib.small_to_device / device_guard degrade to plain tensors / a null context off the GPU, and the frame loop's
    output helpers keep the nesting of an encoder's output."""
    import contextlib

    import torch

    from dmm_net_amd import _lib, video
    t = _lib.small_to_device([3, 1, 2], torch.int32, torch.device("cpu"))
    assert t.dtype == torch.int32 and t.tolist() == [3, 1, 2] and t.device.type == "cpu"
    assert isinstance(_lib.device_guard(torch.device("cpu")), contextlib.nullcontext)
    out = {"a": (torch.arange(12).view(6, 2), torch.arange(6)), "b": [torch.arange(6).view(6, 1)], "c": "tag"}
    s = video._slice_batch(out, 2, 4)
    assert s["a"][0].tolist() == [[4, 5], [6, 7]] and s["a"][1].tolist() == [2, 3] and s["b"][0].tolist() == [[2], [3]]
    assert isinstance(s["a"], tuple) and isinstance(s["b"], list) and s["c"] == "tag"


def test_valid_layout_is_read_once_per_clip_tensor():
    """``DMM_Model._valid_layout_of_clip``: the reference hands the SAME ``tplt_valid_batch`` tensor to every frame step of a
    clip (trainer.py:113-121) -- its layout (live templates per video, non-prefix scale) is read from the device once per
    tensor object and version; an in-place edit or another tensor is read again; nothing ends up in the module's __dict__
    (deepcopy / pickle of the model keep working)."""
    import copy
    import pickle
    import torch
    from dmm_net_amd import dmm_model
    cfg = {"matching": {"algo": "relax"}, "relax_max_iter": 10, "relax_proj_iter": 5, "relax_learning_rate": 0.1,
           "score_weight": 0.3}
    m = dmm_model.DMM_Model(cfg, 0)
    v = torch.tensor([[1., 1., 0.], [1., 0., 0.]])
    reads = []
    real = dmm_model.DMM_Model._valid_layout
    m._valid_layout = lambda t, *a: (reads.append(1), real(t, *a))[1]
    a = m._valid_layout_of_clip(v)
    assert a[0] == [2, 1] and a[1] is None
    a[0].append(99)                                # a caller editing its answer does not edit the cache
    assert m._valid_layout_of_clip(v)[0] == [2, 1] and len(reads) == 1
    v[1, 1] = 1                                   # in-place edit: version moves, the flags are read again
    assert m._valid_layout_of_clip(v)[0] == [2, 2]
    w = torch.tensor([[1., 0., 1.], [0., 0., 0.]])  # another clip, templates not a prefix
    n_tplt, scale = m._valid_layout_of_clip(w)
    assert n_tplt == [2, 0] and scale.tolist() == [[1.0, 0.0, 0.0], [0.0, 0.0, 0.0]]
    assert m._valid_layout_of_clip(v)[0] == [2, 2]  # (one entry per model: v is read again, correctly)
    del m._valid_layout                           # (the counting stand-in of this test is not part of the model)
    copy.deepcopy(m)
    pickle.dumps(m)
    assert not any(k.startswith("_valid") for k in m.__dict__)


def test_bench_compacts_the_other_workloads_lines():
    """``bench.compact``: what the default line's ``other_configs`` keeps of a workload's own line -- the drop-in's cases
    (wall / device / launches per call, where the host threads sat) and the training form's per-kernel rows."""
    import sys
    from conftest import ROOT
    sys.path.insert(0, ROOT)
    import bench
    drop = {"metric": "m", "value": 9.0, "unit": "calls/s", "ms_per_step": 0.1, "steps": 3,
            "config": {"workload": "w", "host_threads": "pinned to the CPUs of NUMA node 1 (the GPU's)",
                       "cases": {"eval_forward_50x5": {"wall_us": 100.0, "device_us": 102.0, "library_launches": 3,
                                                       "wall_over_device": 0.98, "what": "...", "calls": 200}}}}
    c = bench.compact(drop)
    assert c["cases"] == {"eval_forward_50x5": {"wall_us": 100.0, "device_us": 102.0, "library_launches": 3,
                                                "wall_over_device": 0.98}}
    assert c["host_threads"].startswith("pinned") and c["workload"] == "w" and "roofline" not in c
    train = {"metric": "m", "value": 1.0, "unit": "frames/s", "ms_per_step": 4.4, "steps": 10,
             "roofline": {"bound": "hbm", "kernel": "k", "achieved": 6.4e3, "peak": 8e3, "unit": "GB/s", "frac": 0.8, "traffic": 1},
             "config": {"workload": "t", "by_batch": {"64": {"fwd_bwd_ms": 0.73, "selected_planes_per_frame": 47.8,
                                                              "kernels": {"mask_mix_bwd": {"ms": 0.175, "frac": 0.686, "bound": "hbm"},
                                                                          "relax_match_bwd": {"ms": 0.157}}, "note": "..."}}}}
    c = bench.compact(train)
    assert c["roofline"] == {"bound": "hbm", "kernel": "k", "achieved": 6.4e3, "peak": 8e3, "unit": "GB/s", "frac": 0.8}
    assert c["by_batch"]["64"]["kernels"] == {"mask_mix_bwd": {"ms": 0.175, "frac": 0.686}, "relax_match_bwd": {"ms": 0.157}}


def test_importing_the_matching_layer_leaves_the_environment_alone(tmp_path):
    """The one-line swap imports ``dmm_net_amd.match_model`` only: that must neither touch ``os.environ`` (MIOpen's
    MIOPEN_USER_DB_PATH is set when an ENCODER is constructed, not at import) nor write under the home directory."""
    import json
    import os
    import subprocess
    import sys
    from conftest import ROOT
    code = ("import os, json, sys; sys.path.insert(0, %r); e0 = dict(os.environ); import dmm_net_amd.match_model; "
            "import dmm_net_amd; e1 = dict(os.environ); h1 = sorted(os.listdir(os.environ['HOME'])); "
            "from dmm_net_amd.encoder import FeatureEncoder; FeatureEncoder('resnet34'); e2 = dict(os.environ); "
            "print(json.dumps({'same': e0 == e1, 'home_after_import': h1, 'home': sorted(os.listdir(os.environ['HOME'])), "
            "'db_after_encoder': e2.get('MIOPEN_USER_DB_PATH')}))" % ROOT)
    home = tmp_path / "home"
    home.mkdir()
    env = {k: v for k, v in os.environ.items() if k != "MIOPEN_USER_DB_PATH"}
    env["HOME"] = str(home)
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-800:]
    out = json.loads(r.stdout.strip().splitlines()[-1])
    assert out["same"], "importing the package changed os.environ"
    assert out["home_after_import"] == [], "importing the package wrote under the home directory"
    # after the import nothing was written; the encoder's constructor is what seeds the find-db copy
    assert out["db_after_encoder"] and out["db_after_encoder"].startswith(str(home))
    assert out["home"] == [".cache"]
