"""GPU: spf_adam_step (FlatAdam) against torch.nn.utils.clip_grad_norm_ + torch.optim.Adam on the same gradients —
the reference's step tail (spurfies/train.py:359-363, 548-564)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _make(seed, shapes=((700, 64), (256, 103), (256,), (3, 256), (1,))):
    g = torch.Generator().manual_seed(seed)
    return [torch.nn.Parameter((torch.randn(s, generator=g) * 0.1).cuda()) for s in shapes]


@pytest.mark.parametrize("max_norm", [0.0, 1.0])
def test_flat_adam_matches_torch_adam(max_norm):
    from spurfies_amd.dist import FlatGrads
    from spurfies_amd.optim import FlatAdam

    ours, ref = _make(0), _make(0)
    flat = FlatGrads(ours)
    opt = FlatAdam([{"params": [], "lr": 1e-2}, {"params": ours, "lr": 5e-4}], flat_grads=flat)
    topt = torch.optim.Adam([{"params": [], "lr": 1e-2}, {"params": ref, "lr": 5e-4}])
    sched = torch.optim.lr_scheduler.CosineAnnealingLR(opt, T_max=100, eta_min=3e-4)
    tsched = torch.optim.lr_scheduler.CosineAnnealingLR(topt, T_max=100, eta_min=3e-4)
    g = torch.Generator().manual_seed(1)
    for it in range(6):
        scale = 10.0 if it % 2 else 0.01                       # alternately above / below the clip threshold
        grads = [torch.randn(p.shape, generator=g).cuda() * scale for p in ours]
        flat.zero_()
        for p, q, gr in zip(ours, ref, grads):
            p.grad.copy_(gr)
            q.grad = gr.clone()
        st = opt.step(max_norm=max_norm)
        if max_norm > 0:
            torch.nn.utils.clip_grad_norm_(ref, max_norm)
        topt.step()
        sched.step()
        tsched.step()
        assert float(st[0]) == it + 1 and float(st[1]) == 0
        for p, q in zip(ours, ref):
            np.testing.assert_allclose(p.detach().cpu().numpy(), q.detach().cpu().numpy(), rtol=2e-6, atol=1e-8)
            np.testing.assert_allclose(p.grad.cpu().numpy(), q.grad.cpu().numpy(), rtol=1e-5, atol=1e-9)   # clipped in place
    for p, q in zip(ours, ref):
        np.testing.assert_allclose(opt.state[p]["exp_avg_sq"].cpu().numpy(), topt.state[q]["exp_avg_sq"].cpu().numpy(), rtol=1e-5, atol=1e-12)


def test_non_finite_gradient_skips_the_update_and_state_dict_roundtrips():
    from spurfies_amd.dist import FlatGrads
    from spurfies_amd.optim import FlatAdam

    ours = _make(3)
    flat = FlatGrads(ours)
    opt = FlatAdam([{"params": [], "lr": 1e-2}, {"params": ours, "lr": 5e-4}], flat_grads=flat)
    flat.buffer.normal_()
    opt.step(max_norm=1.0)
    before = [p.detach().clone() for p in ours]
    m_before = opt.state[ours[0]]["exp_avg"].clone()
    flat.buffer.normal_()
    flat.buffer[12345] = float("nan")
    st = opt.step(max_norm=1.0)
    assert float(st[0]) == 1 and float(st[1]) == 1              # step count unchanged, one skipped step
    for p, b in zip(ours, before):
        assert torch.equal(p.detach(), b)
    assert torch.equal(opt.state[ours[0]]["exp_avg"], m_before)
    # the reference's checkpoint layout: loads into torch.optim.Adam and back
    sd = opt.state_dict()
    assert set(sd) == {"state", "param_groups"} and set(sd["state"][0]) == {"step", "exp_avg", "exp_avg_sq"}
    ref = _make(3)
    topt = torch.optim.Adam([{"params": [], "lr": 1e-2}, {"params": ref, "lr": 5e-4}])
    topt.load_state_dict(sd)
    assert float(topt.state[ref[0]]["step"]) == 1
    opt2_params = _make(3)
    flat2 = FlatGrads(opt2_params)
    opt2 = FlatAdam([{"params": [], "lr": 1e-2}, {"params": opt2_params, "lr": 5e-4}], flat_grads=flat2)
    opt2.load_state_dict(topt.state_dict())
    assert float(opt2._flat["state"][0]) == 1
    assert torch.equal(opt2.state[opt2_params[1]]["exp_avg"], opt.state[ours[1]]["exp_avg"])
    assert opt2.state[opt2_params[1]]["exp_avg"].data_ptr() >= opt2._flat["m"].data_ptr()      # still views of the flat buffer
