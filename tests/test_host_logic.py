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
