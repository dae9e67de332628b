"""Optimiser step of the reference trainer on the MI355X.

Mirrors nuwa_pytorch/optimizer.py (get_optimizer: AdamW, no weight decay on the ndim < 2 parameters, lr 3e-4) and the
clip-then-step of train_nuwa.py:253-255, as ONE fused pass over all parameters: libamdnuwa's multi-tensor kernels read a device table
of chunks, the global gradient norm is reduced in a fixed order and never leaves the device, and the AdamW update applies the clip
coefficient on the fly.

    opt = get_optimizer(nuwa.parameters(), lr=3e-4, wd=0.01, filter_by_requires_grad=True)
    loss.backward(); opt.step(max_grad_norm=0.5); opt.zero_grad()
    torch.save(opt.state_dict(), path)          # moments + per-parameter step counts; load_state_dict() resumes them
"""
import ctypes as C
import math

import numpy as np
import torch

from . import _lib
from . import ops

CHUNK = 65536


class _Chunk(C.Structure):                       # == amdnuwa_adamw_chunk
    _fields_ = [('p', C.c_void_p), ('g', C.c_void_p), ('m', C.c_void_p), ('v', C.c_void_p), ('n', C.c_longlong),
                ('weight_decay', C.c_float), ('bias_correction1', C.c_float), ('bias_correction2', C.c_float)]


def separate_weight_decayable_params(params):
    """optimizer.py:6-9"""
    no_wd = [p for p in params if p.ndim < 2]
    wd = [p for p in params if p.ndim >= 2]
    return wd, no_wd


class FusedAdamW(torch.optim.Optimizer):
    """torch.optim.AdamW semantics (betas (0.9, 0.999), eps 1e-8, decoupled decay) for fp32 parameters on one HIP device.

    A regular `torch.optim.Optimizer`: two parameter groups (matrices with weight decay, ndim < 2 without -- optimizer.py:6-9), so
    learning-rate schedulers attach to `param_groups`; `state[p]` holds `step`, `exp_avg`, `exp_avg_sq`, so `state_dict()` /
    `load_state_dict()` checkpoint and resume the moments and the per-parameter step counts.  The moments are the SAME storage the
    fused kernel updates through its chunk table; parameters are written through raw pointers and their autograd version counter is
    bumped afterwards, so saved-tensor checks see the update."""

    def __init__(self, params, lr=3e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2):
        seen, flat = set(), []
        for p in params:
            if id(p) not in seen:
                seen.add(id(p))
                flat.append(p)
        assert flat and all(p.is_cuda and p.dtype == torch.float32 for p in flat), \
            'FusedAdamW runs through libamdnuwa: fp32 parameters on an MI355X device'
        wd_p, no_wd_p = separate_weight_decayable_params(flat)
        groups = [g for g in ({'params': wd_p, 'weight_decay': weight_decay}, {'params': no_wd_p, 'weight_decay': 0.}) if g['params']]
        super().__init__(groups, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self.params = [p for g in self.param_groups for p in g['params']]
        self.device = self.params[0].device
        self.state_m = [torch.zeros_like(p, memory_format=torch.contiguous_format) for p in self.params]
        self.state_v = [torch.zeros_like(p, memory_format=torch.contiguous_format) for p in self.params]
        self.param_steps = [0] * len(self.params)       # torch counts updates per parameter (one without a gradient is skipped)
        self._bind_state()
        # host image of the chunk table (numpy structured array == amdnuwa_adamw_chunk); everything except the gradient pointers,
        # the decay and the per-parameter bias corrections is filled once
        self._dtype = np.dtype([('p', '<u8'), ('g', '<u8'), ('m', '<u8'), ('v', '<u8'), ('n', '<i8'), ('weight_decay', '<f4'),
                                ('bias_correction1', '<f4'), ('bias_correction2', '<f4')], align=True)
        assert self._dtype.itemsize == C.sizeof(_Chunk)
        counts = np.array([-(-p.numel() // CHUNK) for p in self.params])
        self.nchunks = int(counts.sum())
        self._owner = np.repeat(np.arange(len(self.params)), counts)                 # chunk -> parameter index
        first = np.concatenate(([0], np.cumsum(counts)[:-1]))
        self._off = (np.arange(self.nchunks) - np.repeat(first, counts)).astype(np.uint64) * np.uint64(4 * CHUNK)   # byte offset inside the tensor
        numel = np.array([p.numel() for p in self.params], dtype=np.int64)
        H = np.zeros(self.nchunks, dtype=self._dtype)
        H['p'] = np.array([p.data_ptr() for p in self.params], dtype=np.uint64)[self._owner] + self._off
        H['m'] = np.array([t.data_ptr() for t in self.state_m], dtype=np.uint64)[self._owner] + self._off
        H['v'] = np.array([t.data_ptr() for t in self.state_v], dtype=np.uint64)[self._owner] + self._off
        H['n'] = np.minimum(CHUNK, numel[self._owner] - (self._off // np.uint64(4)).astype(np.int64))
        self._host = H
        self._pptr = [p.data_ptr() for p in self.params]
        self._table = torch.empty(self.nchunks * self._dtype.itemsize, dtype=torch.uint8, device=self.device)
        self._partials = torch.empty(max(self.nchunks, 1), dtype=torch.float32, device=self.device)
        self._norm = torch.ones(2, dtype=torch.float32, device=self.device)      # [total norm, clip coefficient]
        self._built = True

    # -- hyper-parameters live in param_groups (schedulers edit them there); the fused launch takes ONE lr / betas / eps
    def _hyper(self):
        g0 = self.param_groups[0]
        for g in self.param_groups[1:]:
            if (g['lr'], tuple(g['betas']), g['eps']) != (g0['lr'], tuple(g0['betas']), g0['eps']):
                raise RuntimeError('FusedAdamW: lr / betas / eps must agree across the parameter groups (one fused launch)')
        return float(g0['lr']), tuple(g0['betas']), float(g0['eps'])

    lr = property(lambda self: self._hyper()[0])
    betas = property(lambda self: self._hyper()[1])
    eps = property(lambda self: self._hyper()[2])

    def _bind_state(self):
        """state[p] views the buffers the kernel updates (exp_avg / exp_avg_sq) and the host-side step count"""
        for i, p in enumerate(self.params):
            self.state[p] = {'step': torch.tensor(float(self.param_steps[i])), 'exp_avg': self.state_m[i], 'exp_avg_sq': self.state_v[i]}

    def add_param_group(self, param_group):
        """the chunk table of the fused launch is built once, in __init__ (torch's own constructor calls this for the initial groups)"""
        if getattr(self, '_built', False):
            raise RuntimeError('FusedAdamW: parameter groups cannot be added after construction (the fused chunk table is fixed); '
                               'build a new optimiser over all parameters instead')
        super().add_param_group(param_group)

    def state_dict(self):
        for i, p in enumerate(self.params):
            self.state[p]['step'] = torch.tensor(float(self.param_steps[i]))
        return super().state_dict()

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)              # (torch copies the loaded tensors: move their values into OUR storage)
        with torch.no_grad():
            for i, p in enumerate(self.params):
                st = self.state.get(p, {})
                if 'exp_avg' in st:
                    self.state_m[i].copy_(st['exp_avg'])
                    self.state_v[i].copy_(st['exp_avg_sq'])
                    self.param_steps[i] = int(float(st['step']))
                else:
                    self.state_m[i].zero_(); self.state_v[i].zero_(); self.param_steps[i] = 0
        self._bind_state()

    def _upload(self, advance=False):
        b1, b2 = self.betas
        gp = np.zeros(len(self.params), dtype=np.uint64)
        for i, p in enumerate(self.params):
            g = p.grad
            assert p.data_ptr() == self._pptr[i], 'parameter storage moved after the optimiser was built'
            if g is not None:
                assert g.dtype == torch.float32 and g.is_contiguous()
                gp[i] = g.data_ptr()
                if advance:
                    self.param_steps[i] += 1
        t = np.maximum(np.array(self.param_steps, dtype=np.float64), 1.)
        H = self._host
        has = gp[self._owner] != 0
        H['g'] = np.where(has, gp[self._owner] + self._off, np.uint64(0))
        H['weight_decay'] = np.array([g['weight_decay'] for g in self.param_groups for _ in g['params']], dtype=np.float32)[self._owner]
        H['bias_correction1'] = (1. - b1 ** t).astype(np.float32)[self._owner]
        H['bias_correction2'] = (1. - b2 ** t).astype(np.float32)[self._owner]
        self._table.copy_(torch.from_numpy(H.view(np.uint8).reshape(-1)), non_blocking=False)

    def grad_norm(self, max_norm=0.):
        """device tensor [norm, clip coefficient] of the current gradients (no host synchronisation)"""
        L = _lib.lib()
        self._upload()
        st = torch.cuda.current_stream().cuda_stream
        _lib.check(L.amdnuwa_grad_norm(self._table.data_ptr(), self.nchunks, float(max_norm), self._partials.data_ptr(),
                                       self._norm.data_ptr(), st), 'amdnuwa_grad_norm')
        return self._norm

    @torch.no_grad()
    def step(self, closure=None, max_grad_norm=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        L = _lib.lib()
        st = torch.cuda.current_stream().cuda_stream
        clip = None
        lr, (b1, b2), eps = self._hyper()
        self._upload(advance=True)
        if max_grad_norm is not None:
            _lib.check(L.amdnuwa_grad_norm(self._table.data_ptr(), self.nchunks, float(max_grad_norm), self._partials.data_ptr(),
                                           self._norm.data_ptr(), st), 'amdnuwa_grad_norm')
            clip = self._norm.data_ptr() + 4
        _lib.check(L.amdnuwa_adamw_step(self._table.data_ptr(), self.nchunks, lr, b1, b2, eps, clip, st), 'amdnuwa_adamw_step')
        torch.autograd.graph.increment_version([p for p in self.params if p.grad is not None])   # written behind autograd's back
        ops.WeightCache.EPOCH += 1               # the bf16 operand copies of the weights are stale now
        return loss

    def zero_grad(self, set_to_none=True):
        for p in self.params:
            if p.grad is not None:
                if set_to_none:
                    p.grad = None
                else:
                    p.grad.zero_()


def clip_grad_norm_(optimizer, max_norm):
    """torch.nn.utils.clip_grad_norm_ for the parameters of a FusedAdamW: scales the gradients in place, returns the norm (device)"""
    L = _lib.lib()
    nrm = optimizer.grad_norm(max_norm)
    _lib.check(L.amdnuwa_scale_grads(optimizer._table.data_ptr(), optimizer.nchunks, nrm.data_ptr() + 4,
                                     torch.cuda.current_stream().cuda_stream), 'amdnuwa_scale_grads')
    return nrm[0]


def get_optimizer(params, lr=3e-4, wd=1e-1, filter_by_requires_grad=False):
    """optimizer.py:11-31: AdamW with `wd` on the weight matrices only (Adam when wd == 0)"""
    params = list(params)
    if filter_by_requires_grad:
        params = [p for p in params if p.requires_grad]
    return FusedAdamW(params, lr=lr, weight_decay=wd)
