"""Data-parallel gradient reduction for the NUWA decoder over RCCL / xGMI (one process per GPU).

The path shards over the batch only: every rank holds a full replica, and the one
exchange step per optimiser step is a SUM all-reduce of the parameter gradients (divided by the world
size).  The reference has no distributed code at all; this is new.

Design for 8 x MI355X (xGMI is point-to-point, 7 links per GPU):
  * gradients live in a few large FLAT bucket buffers (`p.grad` is a view into its bucket, so autograd
    accumulates straight into the bucket -- no flatten/unflatten copies);
  * one bucket per decoder layer (+ one for embeddings/logits, + the text encoder), filled in reverse
    layer order by the backward pass; a per-parameter post-accumulate hook counts arrivals and, when a
    bucket is complete, enqueues its reduction on a dedicated communication stream that waits on an event
    recorded on the compute stream -- so layer l's reduction overlaps the backward of layers < l;
  * a handful of >= 8 MB messages (fp32: ~16.8 MB per cfg-3 decoder layer) instead of hundreds of small
    ones keeps RCCL in its bandwidth regime on every xGMI link;
  * `collective='allreduce'` (default) is one RCCL all-reduce per bucket; `collective='rs_ag'` is
    reduce-scatter + all-gather on the same flat buffer (padded to a multiple of the world size), the
    pair a direct all-to-all transport over the 7 links favours and the form a sharded optimiser would cut
    in the middle of;
  * `collective='native'` / `'native_rs_ag'` put the same two forms straight on libamdnuwa's communicator (`amdnuwa_comm_*`,
    include/amdnuwa.h: RCCL opened at run time, one communicator per process, ncclAvg instead of a division pass); the
    128-byte id travels over the torch process group once, at construction.  With the torch collectives the mean uses
    ReduceOp.AVG on the RCCL backend (gloo has no AVG: there the store is divided first);
  * gradient accumulation (the reference trainer runs 8 micro-steps per optimiser step, train_nuwa.py:243):
    micro-steps run under `no_sync()` -- their gradients only accumulate locally in the flat buffers -- and
    the LAST backward of the step, outside `no_sync()`, counts arrivals and launches the reductions.  A second
    counted backward without `zero_grad()` raises instead of silently mixing reduced and local gradients;
  * frozen parameters (the VAE copy inside NUWA never receives gradients -- quirk Q16) are excluded.
Works with the `gloo` backend on CPU tensors too (used by the world_size-2 tests).
"""
import atexit
import contextlib
import ctypes as C
import weakref

import torch
import torch.distributed as dist


def _layer_key(name):
    parts = name.split('.')
    for i, p in enumerate(parts):
        if p == 'layers' and i + 1 < len(parts) and parts[i + 1].isdigit():
            return '.'.join(parts[:i + 2])
    return parts[0] if parts[0] in ('text_transformer', 'video_transformer') else '_embeddings_logits'


def broadcast_parameters(module, src=0, process_group=None):
    """make every replica identical to rank `src`'s with ONE collective per dtype: parameters and buffers are packed into a
    flat tensor, broadcast, and scattered back (hundreds of per-tensor broadcasts cost a launch + a ring set-up each)."""
    if not dist.is_initialized() or dist.get_world_size(process_group) == 1:
        return 0
    by_type, seen = {}, set()
    for t in list(module.parameters()) + list(module.buffers()):
        if id(t) in seen:
            continue
        seen.add(id(t))
        by_type.setdefault((t.dtype, t.device), []).append(t)
    sent = 0
    with torch.no_grad():
        for (dtype, device), ts in by_type.items():
            wire = torch.uint8 if dtype == torch.bool else dtype
            flat = torch.cat([t.detach().reshape(-1).to(wire) for t in ts])
            dist.broadcast(flat, src=src, group=process_group)
            off = 0
            for t in ts:
                t.copy_(flat[off:off + t.numel()].view_as(t).to(dtype))
                off += t.numel()
            sent += flat.numel() * flat.element_size()
    return sent


class NativeComm:
    """libamdnuwa's RCCL communicator for this process (include/amdnuwa.h, amdnuwa_comm_*): rank 0 draws the id, the torch process
    group carries it to the other ranks, every rank joins on its current HIP device."""

    def __init__(self, process_group=None, device=None):
        from . import _lib
        self._lib, self._check = _lib.lib(), _lib.check
        if not self._lib.amdnuwa_comm_available():
            raise RuntimeError('libamdnuwa: librccl could not be opened (amdnuwa_comm_available() == 0)')
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(process_group) if dist.is_initialized() else 0
        self.device = torch.cuda.current_device() if device is None else int(device)
        ident = C.create_string_buffer(128)
        if self.rank == 0:
            self._check(self._lib.amdnuwa_comm_unique_id(ident, 128), 'amdnuwa_comm_unique_id')
        box = [bytes(ident.raw)]
        if self.world > 1:
            dist.broadcast_object_list(box, src=dist.get_global_rank(process_group, 0) if process_group is not None else 0,
                                       group=process_group)
        self._h = C.c_void_p()
        self._rc(self._lib.amdnuwa_comm_init(C.byref(self._h), box[0], 128, self.rank, self.world, self.device), 'amdnuwa_comm_init')
        # the communicator is torn down explicitly (GradReducer.remove() / close()) or at interpreter exit while HIP and the process
        # group are still alive -- not from __del__ during interpreter teardown
        ref = weakref.ref(self)
        atexit.register(lambda: ref() is not None and ref().close())

    def _rc(self, rc, what):
        if rc == -4:
            raise RuntimeError(f'{what}: {self._lib.amdnuwa_comm_last_error().decode()}')
        self._check(rc, what)

    def allreduce(self, t, average=True, stream=None):
        assert t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()
        st = (stream or torch.cuda.current_stream()).cuda_stream
        self._rc(self._lib.amdnuwa_comm_allreduce(self._h, t.data_ptr(), t.numel(), int(average), st), 'amdnuwa_comm_allreduce')

    def reduce_scatter_allgather(self, t, average=True, stream=None):
        assert t.is_cuda and t.dtype == torch.float32 and t.is_contiguous() and t.numel() % self.world == 0
        st = (stream or torch.cuda.current_stream()).cuda_stream
        self._rc(self._lib.amdnuwa_comm_reduce_scatter_allgather(self._h, t.data_ptr(), t.numel() // self.world, int(average), st),
                 'amdnuwa_comm_reduce_scatter_allgather')

    def broadcast(self, t, root=0, stream=None):
        assert t.is_cuda and t.is_contiguous()
        st = (stream or torch.cuda.current_stream()).cuda_stream
        self._rc(self._lib.amdnuwa_comm_broadcast(self._h, t.data_ptr(), t.numel() * t.element_size(), int(root), st), 'amdnuwa_comm_broadcast')

    def close(self):
        if self._h:
            self._lib.amdnuwa_comm_destroy(self._h)
            self._h = C.c_void_p()



class _Enqueued:
    """work handle of a collective libamdnuwa enqueued on the communication stream: finish() orders the compute stream after it"""

    def wait(self):
        return True


class GradReducer:
    def __init__(self, module, process_group=None, bucket_fn=_layer_key, average=True, collective='allreduce', always_reduce=False):
        """always_reduce: run the bucket collectives even in a world of one rank (they are then identities) -- lets a 1-GPU box
        exercise the exact RCCL call sequence, in-place shard aliasing included, that N ranks would run"""
        if collective not in ('allreduce', 'rs_ag', 'native', 'native_rs_ag'):
            raise ValueError(collective)
        self.always_reduce = bool(always_reduce)
        self.pg = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(process_group) if dist.is_initialized() else 0
        self.average = average
        self.collective = collective
        self._sync = True
        seen, groups = set(), {}
        for name, p in module.named_parameters():          # named_parameters() de-duplicates shared params
            if not p.requires_grad or id(p) in seen or name.startswith('vae.'):
                continue
            seen.add(id(p))
            groups.setdefault(bucket_fn(name), []).append(p)
        self.buckets = []
        for key, ps in groups.items():
            n = sum(p.numel() for p in ps)
            padded = -(-n // self.world) * self.world                      # rs_ag: equal shards
            store = torch.zeros(padded, dtype=ps[0].dtype, device=ps[0].device)
            flat = store[:n]
            off = 0
            for p in ps:
                p.grad = flat[off:off + p.numel()].view_as(p)
                off += p.numel()
            self.buckets.append(dict(key=key, params=ps, flat=flat, store=store, pending=len(ps), work=None, launched=False))
        self._by_param = {id(p): b for b in self.buckets for p in b['params']}
        self.cuda = bool(self.buckets) and self.buckets[0]['flat'].is_cuda
        self.comm_stream = torch.cuda.Stream() if self.cuda else None
        self.native = None
        if collective.startswith('native') and (self.world > 1 or self.always_reduce):
            if not self.cuda or any(b['flat'].dtype != torch.float32 for b in self.buckets):
                raise RuntimeError("GradReducer(collective='native'): fp32 gradients on a HIP device")
            self.native = NativeComm(process_group, device=self.buckets[0]['flat'].device.index)     # the device that HOLDS the buckets
        self._avg_op = bool(self.cuda and dist.is_initialized() and dist.get_backend(process_group) == 'nccl')
        self._hooks = []
        for b in self.buckets:
            for p in b['params']:
                self._hooks.append(p.register_post_accumulate_grad_hook(self._on_grad))

    # ------------------------------------------------------------------------------------------
    def zero_grad(self):
        """reset buckets for the next optimiser step (grads stay views into the flat buffers)"""
        for b in self.buckets:
            b['store'].zero_()
            b['pending'] = len(b['params'])
            b['work'] = None
            b['launched'] = False
            off = 0
            for p in b['params']:
                if p.grad is None or p.grad.data_ptr() != b['flat'].data_ptr() + off * b['flat'].element_size():
                    p.grad = b['flat'][off:off + p.numel()].view_as(p)
                off += p.numel()

    @contextlib.contextmanager
    def no_sync(self):
        """gradient accumulation: backward passes inside this context only add into the local flat buffers; run the step's
        last micro-batch OUTSIDE it, then finish():

            red.zero_grad()
            for i, mb in enumerate(micro_batches):
                with (red.no_sync() if i + 1 < len(micro_batches) else contextlib.nullcontext()):
                    (loss_fn(mb) / len(micro_batches)).backward()
            red.finish()
        """
        prev, self._sync = self._sync, False
        try:
            yield self
        finally:
            self._sync = prev

    def _on_grad(self, p):
        b = self._by_param[id(p)]
        if b['launched']:          # (checked under no_sync() too: accumulating on top of already reduced = averaged buffers is the same mistake)
            raise RuntimeError(f'GradReducer: bucket {b["key"]!r} received a gradient after it was reduced -- run every micro-batch '
                               f'but the last under no_sync(), and call zero_grad() between optimiser steps')
        if not self._sync:
            return
        b['pending'] -= 1
        if b['pending'] == 0:
            self._launch(b)

    def _reduce(self, b):
        if self.native is not None:                       # (already on the communication stream)
            if self.collective == 'native':
                self.native.allreduce(b['flat'], self.average)
            else:
                self.native.reduce_scatter_allgather(b['store'], self.average)
            return _Enqueued()
        op = dist.ReduceOp.SUM
        if self.average:
            if self._avg_op:
                op = dist.ReduceOp.AVG                    # RCCL divides inside the collective: no extra pass over the bucket
            else:
                b['store'].div_(self.world)
        if self.collective in ('allreduce', 'native'):
            return dist.all_reduce(b['flat'], op=op, group=self.pg, async_op=True)
        shard = b['store'].numel() // self.world
        mine = b['store'][self.rank * shard:(self.rank + 1) * shard]
        dist.reduce_scatter_tensor(mine, b['store'], op=op, group=self.pg)
        return dist.all_gather_into_tensor(b['store'], mine, group=self.pg, async_op=True)

    def _launch(self, b):
        b['launched'] = True
        if self.world == 1 and not (self.always_reduce and (self.native is not None or dist.is_initialized())):
            return
        if self.cuda:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream())
            with torch.cuda.stream(self.comm_stream):
                self.comm_stream.wait_event(ev)
                b['work'] = self._reduce(b)
        else:
            b['work'] = self._reduce(b)

    def finish(self):
        """call after the step's last backward(): launches buckets whose params got no grad this step (unused
        parameters stay zero) and makes the compute stream wait for all reductions."""
        if not self._sync:
            raise RuntimeError('GradReducer.finish() inside no_sync(): the last micro-batch must run outside the context')
        for b in self.buckets:
            if not b['launched']:
                self._launch(b)
        for b in self.buckets:
            if b['work'] is not None:
                b['work'].wait()
        if self.cuda and (self.world > 1 or self.always_reduce):
            torch.cuda.current_stream().wait_stream(self.comm_stream)

    def total_bytes(self):
        return sum(b['flat'].numel() * b['flat'].element_size() for b in self.buckets)

    def remove(self):
        for h in self._hooks:
            h.remove()
        self._hooks = []
        if self.native is not None:
            if self.cuda:
                self.comm_stream.synchronize()
            self.native.close()
            self.native = None
