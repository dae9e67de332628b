import json
import os

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LOG = os.path.join(ROOT, 'gpurun_out', 'parity_log.jsonl')


def rel_err(a, b):
    """max-abs error relative to the reference's max-abs"""
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp(min=1e-30))


def rel_l2(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).norm() / b.norm().clamp(min=1e-30))


def report(name, got, ref, tol):
    """records the error of `got` vs `ref` (max-abs / max-ref) in gpurun_out/parity_log.jsonl and asserts it"""
    assert got.shape == ref.shape, f'{name}: shape {tuple(got.shape)} vs {tuple(ref.shape)}'
    e, l2 = rel_err(got, ref), rel_l2(got, ref)
    finite = bool(torch.isfinite(got).all())
    try:
        os.makedirs(os.path.dirname(LOG), exist_ok=True)
        with open(LOG, 'a') as f:
            f.write(json.dumps(dict(name=name, rel_max=e, rel_l2=l2, tol=tol, finite=finite)) + '\n')
    except OSError:
        pass
    assert finite, f'{name}: non-finite values'
    assert e <= tol, f'{name}: rel err {e:.3e} (l2 {l2:.3e}) > tol {tol:.1e}'
    return e


def record(name, rel_max, l2, tol):
    """log an already computed error pair (same file / format as report())"""
    try:
        os.makedirs(os.path.dirname(LOG), exist_ok=True)
        with open(LOG, 'a') as f:
            f.write(json.dumps(dict(name=name, rel_max=rel_max, rel_l2=l2, tol=tol, finite=True)) + '\n')
    except OSError:
        pass


def bf_round(t):
    return t.to(torch.bfloat16).float()


def to_bf_pair(t, lo):
    """fp32 tensor -> kernels.BF(hi, lo) on the same device"""
    from nuwa_pytorch_amd.kernels import BF
    hi = t.to(torch.bfloat16)
    return BF(hi.contiguous(), (t - hi.float()).to(torch.bfloat16).contiguous() if lo else None)


def bf_value(p):
    v = p.hi.float()
    return v + p.lo.float() if p.lo is not None else v
