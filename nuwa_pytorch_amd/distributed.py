"""Data-parallel gradient reduction for the NUWA decoder over RCCL / xGMI (one process per GPU).

The path shards over the batch only: every rank holds a full replica, and the one
exchange step per optimiser step is a SUM all-reduce of the parameter gradients (divided by the world
size).  The reference has no distributed code at all; this is new.

Design for 8 x MI355X (xGMI is point-to-point, 7 links per GPU):
  * gradients live in a few large FLAT bucket buffers (`p.grad` is a view into its bucket, so autograd
    accumulates straight into the bucket -- no flatten/unflatten copies);
  * one bucket per decoder layer (+ one for embeddings/logits, + the text encoder), filled in reverse
    layer order by the backward pass; a per-parameter post-accumulate hook counts arrivals and, when a
    bucket is complete, enqueues `all_reduce` on a dedicated communication stream that waits on an event
    recorded on the compute stream -- so layer l's all-reduce overlaps the backward of layers < l;
  * a handful of >= 8 MB messages (fp32: ~16.8 MB per cfg-3 decoder layer) instead of hundreds of small
    ones keeps RCCL in its bandwidth regime on every xGMI link;
  * frozen parameters (the VAE copy inside NUWA never receives gradients -- quirk Q16) are excluded.
Works with the `gloo` backend on CPU tensors too (used by the world_size-2 tests).
"""
import torch
import torch.distributed as dist


def _layer_key(name):
    parts = name.split('.')
    for i, p in enumerate(parts):
        if p == 'layers' and i + 1 < len(parts) and parts[i + 1].isdigit():
            return '.'.join(parts[:i + 2])
    return parts[0] if parts[0] in ('text_transformer', 'video_transformer') else '_embeddings_logits'


class GradReducer:
    def __init__(self, module, process_group=None, bucket_fn=_layer_key, average=True):
        self.pg = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.average = average
        seen, groups = set(), {}
        for name, p in module.named_parameters():          # named_parameters() de-duplicates shared params
            if not p.requires_grad or id(p) in seen or name.startswith('vae.'):
                continue
            seen.add(id(p))
            groups.setdefault(bucket_fn(name), []).append(p)
        self.buckets = []
        for key, ps in groups.items():
            n = sum(p.numel() for p in ps)
            flat = torch.zeros(n, dtype=ps[0].dtype, device=ps[0].device)
            off = 0
            for p in ps:
                p.grad = flat[off:off + p.numel()].view_as(p)
                off += p.numel()
            self.buckets.append(dict(key=key, params=ps, flat=flat, pending=len(ps), work=None, event=None))
        self._by_param = {id(p): b for b in self.buckets for p in b['params']}
        self.cuda = bool(self.buckets) and self.buckets[0]['flat'].is_cuda
        self.comm_stream = torch.cuda.Stream() if self.cuda else None
        self._hooks = []
        for b in self.buckets:
            for p in b['params']:
                self._hooks.append(p.register_post_accumulate_grad_hook(self._on_grad))

    # ------------------------------------------------------------------------------------------
    def zero_grad(self):
        """reset buckets for the next step (grads stay views into the flat buffers)"""
        for b in self.buckets:
            b['flat'].zero_()
            b['pending'] = len(b['params'])
            b['work'] = None
            off = 0
            for p in b['params']:
                if p.grad is None or p.grad.data_ptr() != b['flat'].data_ptr() + off * b['flat'].element_size():
                    p.grad = b['flat'][off:off + p.numel()].view_as(p)
                off += p.numel()

    def _on_grad(self, p):
        b = self._by_param[id(p)]
        b['pending'] -= 1
        if b['pending'] == 0:
            self._launch(b)

    def _launch(self, b):
        if self.world == 1:
            return
        if self.cuda:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream())
            with torch.cuda.stream(self.comm_stream):
                self.comm_stream.wait_event(ev)
                if self.average:
                    b['flat'].div_(self.world)
                b['work'] = dist.all_reduce(b['flat'], op=dist.ReduceOp.SUM, group=self.pg, async_op=True)
        else:
            if self.average:
                b['flat'].div_(self.world)
            b['work'] = dist.all_reduce(b['flat'], op=dist.ReduceOp.SUM, group=self.pg, async_op=True)

    def finish(self):
        """call after backward(): launches buckets whose params got no grad this step (unused
        parameters stay zero) and makes the compute stream wait for all reductions."""
        for b in self.buckets:
            if b['pending'] > 0 and b['work'] is None:
                self._launch(b)
        for b in self.buckets:
            if b['work'] is not None:
                b['work'].wait()
        if self.cuda and self.world > 1:
            torch.cuda.current_stream().wait_stream(self.comm_stream)

    def total_bytes(self):
        return sum(b['flat'].numel() * b['flat'].element_size() for b in self.buckets)

    def remove(self):
        for h in self._hooks:
            h.remove()
        self._hooks = []
