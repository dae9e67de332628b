"""Row f2: the fused clip + AdamW step of libamdnuwa against torch.nn.utils.clip_grad_norm_ + torch.optim.AdamW (the reference
trainer's train_nuwa.py:253-255 with optimizer.py:6-31's parameter groups) on identical parameters and gradients."""
import pytest
import torch
from torch import nn

pytestmark = pytest.mark.gpu
DEV = 'cuda'


def _model():
    torch.manual_seed(0)
    return nn.Sequential(nn.Linear(37, 130), nn.LayerNorm(130), nn.Linear(130, 70000 // 130), nn.Linear(70000 // 130, 11)).to(DEV)


@pytest.mark.parametrize('max_norm', [None, 0.5, 1e6])
def test_fused_clip_adamw_matches_torch(max_norm):
    if not torch.cuda.is_available():
        pytest.skip('no GPU')
    from nuwa_pytorch_amd.optimizer import get_optimizer, separate_weight_decayable_params
    a, b = _model(), _model()
    wd_p, no_wd_p = separate_weight_decayable_params(list(b.parameters()))
    ref = torch.optim.AdamW([{'params': wd_p}, {'params': no_wd_p, 'weight_decay': 0}], lr=3e-3, weight_decay=0.1)
    opt = get_optimizer(a.parameters(), lr=3e-3, wd=0.1, filter_by_requires_grad=True)
    g = torch.Generator().manual_seed(1)
    for step in range(4):
        x = torch.randn(16, 37, generator=g).to(DEV)
        for m in (a, b):
            m.zero_grad(set_to_none=True)
            (m(x).square().mean() * 50).backward()
        if step == 2:                                  # a parameter without gradient this step is skipped by both
            a[3].bias.grad = None
            b[3].bias.grad = None
        if max_norm is not None:
            n_ref = torch.nn.utils.clip_grad_norm_(b.parameters(), max_norm)
        ref.step()
        opt.step(max_grad_norm=max_norm)
        if max_norm is not None:
            torch.testing.assert_close(opt._norm[0], n_ref, rtol=1e-5, atol=1e-7)
        for (n, pa), pb in zip(a.named_parameters(), b.parameters()):
            # (fp32 rounding of the update differs between implementations where sqrt(v) ~ eps: allow 1 % of one lr-sized step)
            torch.testing.assert_close(pa, pb, rtol=2e-5, atol=3e-5, msg=lambda s, n=n: f'step {step} {n}: {s}')


def test_clip_grad_norm_in_place_and_weight_cache_epoch():
    if not torch.cuda.is_available():
        pytest.skip('no GPU')
    from nuwa_pytorch_amd.optimizer import FusedAdamW, clip_grad_norm_
    from nuwa_pytorch_amd import ops
    a, b = _model(), _model()
    x = torch.randn(8, 37, device=DEV)
    for m in (a, b):
        (m(x).square().mean() * 100).backward()
    opt = FusedAdamW(a.parameters())
    n1 = clip_grad_norm_(opt, 0.25)
    n2 = torch.nn.utils.clip_grad_norm_(b.parameters(), 0.25)
    torch.testing.assert_close(n1, n2, rtol=1e-5, atol=1e-7)
    for pa, pb in zip(a.parameters(), b.parameters()):
        torch.testing.assert_close(pa.grad, pb.grad, rtol=1e-5, atol=1e-8)
    e0 = ops.WeightCache.EPOCH
    opt.step()
    assert ops.WeightCache.EPOCH == e0 + 1             # cached bf16 weight copies are rebuilt after the update


def test_fused_adamw_is_a_torch_optimizer_with_resumable_state():
    """FusedAdamW is a torch.optim.Optimizer: LR schedulers drive param_groups, state_dict()/load_state_dict() resume the moments and the
    per-parameter step counts (a resumed run continues bit for bit), and the in-place update bumps the parameters' version counter"""
    if not torch.cuda.is_available():
        pytest.skip('no GPU')
    from nuwa_pytorch_amd.optimizer import FusedAdamW
    a, b = _model(), _model()
    opt = FusedAdamW(a.parameters(), lr=1e-2, weight_decay=0.1)
    assert isinstance(opt, torch.optim.Optimizer) and len(opt.param_groups) == 2
    assert opt.param_groups[1]['weight_decay'] == 0 and all(p.ndim < 2 for p in opt.param_groups[1]['params'])
    sched = torch.optim.lr_scheduler.StepLR(opt, step_size=1, gamma=0.5)
    g = torch.Generator().manual_seed(3)
    xs = [torch.randn(16, 37, generator=g).to(DEV) for _ in range(4)]

    def run(model, optim, x, scheduler=None):
        model.zero_grad(set_to_none=True)
        (model(x).square().mean() * 50).backward()
        optim.step(max_grad_norm=1.0)
        if scheduler is not None:
            scheduler.step()

    v0 = a[0].weight._version
    run(a, opt, xs[0], sched)
    run(a, opt, xs[1], sched)
    assert a[0].weight._version > v0
    assert abs(opt.lr - 1e-2 * 0.25) < 1e-12                       # the scheduler's edits reach the fused launch
    sd = opt.state_dict()
    assert len(sd['state']) == len(list(a.parameters())) and float(sd['state'][0]['step']) == 2.0
    # resume into a fresh optimiser on a copy of the model
    b.load_state_dict(a.state_dict())
    opt2 = FusedAdamW(b.parameters(), lr=1e-2, weight_decay=0.1)
    opt2.load_state_dict(sd)
    assert abs(opt2.lr - 1e-2 * 0.25) < 1e-12 and opt2.param_steps == opt.param_steps
    run(a, opt, xs[2])
    run(b, opt2, xs[2])
    for (n, pa), pb in zip(a.named_parameters(), b.parameters()):
        assert torch.equal(pa, pb), n
    for ma, mb in zip(opt.state_m, opt2.state_m):
        assert torch.equal(ma, mb)


def test_fused_adamw_rejects_late_param_groups():
    """the fused chunk table is built once: add_param_group after construction raises instead of being silently ignored"""
    if not torch.cuda.is_available():
        pytest.skip('no GPU')
    from nuwa_pytorch_amd.optimizer import FusedAdamW
    a = torch.nn.Linear(8, 8).cuda()
    opt = FusedAdamW(a.parameters(), lr=1e-3)
    with pytest.raises(RuntimeError, match='cannot be added after construction'):
        opt.add_param_group({'params': [torch.nn.Parameter(torch.zeros(4, device='cuda'))]})
