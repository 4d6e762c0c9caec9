"""End-to-end GPU parity: the whole KD forward + distillation loss (R50 -> pixel decoder -> video decoder ->
GT criterion -> KD targets -> KD criterion) on HIP vs the CPU oracle, same seeded weights, same injected points."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _check(hip, ref, B, NL):
    # mask logits and class logits of all 10 prediction heads: 1e-3 relative (north star)
    for k in ("s_logits", "s_masks"):
        b = ref[k].astype(np.float64)
        np.testing.assert_allclose(hip[k], b, rtol=1e-3, atol=1e-3 * np.abs(b).max())
    assert hip["kd_counts"] == ref["kd_counts"]
    iq, it, nm = (x.cpu().numpy() for x in hip["model"].criterion.last_indices)   # KD pass ran last
    order = [NL - 1] + list(range(NL - 1))
    for li, layer in enumerate(order):
        for b in range(B):
            ri, rj = ref["idx_kd"][li][b]
            prob = layer * B + b
            assert nm[prob] == len(ri)
            np.testing.assert_array_equal(iq[prob, :len(ri)], ri)      # Hungarian indices bit-exact
            np.testing.assert_array_equal(it[prob, :len(ri)], rj)
    assert sorted(hip["losses"]) == sorted(ref["losses"])
    for k, v in ref["losses"].items():
        np.testing.assert_allclose(hip["losses"][k], float(v), rtol=1e-3, atol=1e-6, err_msg=k)


def test_kd_forward_loss_small(oracle):
    from tests.parity import run_case
    hip, ref = run_case(oracle, seed=3, B=2, T=2, H0=60, W0=90, Q=16, P=256, ns=(3, 4))
    assert len(hip["losses"]) == 42
    _check(hip, ref, 2, 10)


def test_kd_forward_loss_config1_plumbing(oracle):
    """BASELINE config 1 shape: 1 x 256 x 256 frame, 10 queries, forward + matcher (+ losses)"""
    from tests.parity import run_case
    hip, ref = run_case(oracle, seed=4, B=1, T=1, H0=256, W0=256, Q=10, P=1024, ns=(3,))
    _check(hip, ref, 1, 10)
