"""The backward gathers each stage's slice of the concatenated HR feature gradient from its producers' small dPre maps in two launches
(KBPN._gather_conv) instead of letting every producer accumulate into a whole-gradient buffer as autograd does
(/root/reference kbpn.py:355-372 forward: torch.cat of the stage outputs feeds sr_reconst / down.conv / output_conv).  Both orders are the same
sum; the gathered one rounds to fp16 once per slice instead of once per producer, so the two agree to fp16 noise -- and each is
checked against the reference fixtures by the other tests (the default is the gathered order)."""
import pytest
import torch

from golden_utils import load_golden
from test_joint_gpu import build_model
from test_determinism_gpu import _step

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("case", ["e2e_pspnet_it40000", "e2e_blurskip_x8_it40000", "e2e_pspnet_it1"])
def test_gathered_and_accumulated_concat_gradient_agree(case):
    g = load_golden(case)
    m, _ = build_model(g, 8)
    sd0 = {k: v.detach().clone() for k, v in m.state_dict().items()}
    kb = m._runtime()["kbpn"]
    assert kb.gather
    a = _step(m, g)
    m.load_state_dict(sd0)
    kb.gather = False
    b = _step(m, g)
    kb.gather = True
    worst = 0.0
    n = 0
    for k in a:
        if not k.startswith("grad."):
            assert torch.equal(a[k], b[k]), k            # the forward is untouched
            continue
        x, y = a[k].double(), b[k].double()
        if x.numel() > 1 and float(y.norm()) > 0:
            worst = max(worst, float((x - y).norm() / y.norm()))
            n += 1
    print(f"{case}: {n} gradient tensors, gathered vs accumulated order: worst relative L2 difference {worst:.2e}")
    assert n >= 20 and worst < 2e-2
