"""world_size-2 CPU tests (gloo) of the data-parallel gradient reducer: bucketed flat gradients,
hook-driven async all-reduce (and the reduce-scatter + all-gather form), frozen / unused parameters, equality
with the single-process gradient of the full batch, gradient accumulation under no_sync() (the reference trainer
runs several micro-batches per optimiser step, train_nuwa.py:243), the one-shot flat parameter broadcast."""
import os
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp
from torch import nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


class Toy(nn.Module):
    def __init__(self):
        super().__init__()
        self.video_transformer = nn.Module()
        self.video_transformer.layers = nn.ModuleList([nn.ModuleList([nn.Linear(8, 8), nn.Linear(8, 8)]) for _ in range(3)])
        self.to_logits = nn.Linear(8, 5, bias=False)
        self.unused = nn.Linear(8, 8)                 # never gets a gradient
        self.vae = nn.Linear(4, 4)                    # frozen copy: excluded from the buckets (quirk Q16)

    def forward(self, x):
        for a, b in self.video_transformer.layers:
            x = x + b(torch.tanh(a(x)))
        return self.to_logits(x)


def _worker(rank, world, port, q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from nuwa_pytorch_amd.distributed import GradReducer
    torch.manual_seed(0)
    m = Toy()
    red = GradReducer(m)
    assert len(red.buckets) == 4                       # 3 layers + (logits, unused): the last one completes in finish()
    assert all(not k['key'].startswith('vae') for k in red.buckets)
    torch.manual_seed(1)
    X, Y = torch.randn(8, 8), torch.randint(0, 5, (8,))
    outs = []
    for step in range(2):                              # two steps: buckets must reset correctly
        red.zero_grad()
        xs, ys = X[rank * 4:(rank + 1) * 4], Y[rank * 4:(rank + 1) * 4]
        loss = nn.functional.cross_entropy(m(xs), ys)
        loss.backward()
        red.finish()
        outs.append({n: p.grad.detach().numpy().copy() for n, p in m.named_parameters() if p.grad is not None})
    q.put((rank, outs))
    dist.barrier()
    dist.destroy_process_group()


def test_reducer_world2_matches_full_batch():
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in range(2))
    res = {r: [{n: torch.from_numpy(v) for n, v in st.items()} for st in outs] for r, outs in res.items()}
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    torch.manual_seed(0)
    m = Toy()
    torch.manual_seed(1)
    X, Y = torch.randn(8, 8), torch.randint(0, 5, (8,))
    nn.functional.cross_entropy(m(X), Y).backward()     # mean over the full batch == mean of the two half-batch means
    for step in range(2):
        for n, p in m.named_parameters():
            if n.startswith('vae') or n.startswith('unused'):
                continue
            for r in (0, 1):
                torch.testing.assert_close(res[r][step][n], p.grad, rtol=1e-5, atol=1e-6)
        assert torch.equal(res[0][step]['to_logits.weight'], res[1][step]['to_logits.weight'])
        assert float(res[0][step]['unused.weight'].abs().max()) == 0


def _accum_worker(rank, world, port, q, collective):
    import contextlib
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from nuwa_pytorch_amd.distributed import GradReducer, broadcast_parameters
    torch.manual_seed(100 + rank)                      # replicas start DIFFERENT: the flat broadcast must make them equal
    m = Toy()
    sent = broadcast_parameters(m, src=0)
    assert sent >= sum(p.numel() * 4 for p in m.parameters())
    w0 = {n: p.detach().clone() for n, p in m.named_parameters()}
    red = GradReducer(m, collective=collective)
    torch.manual_seed(1)
    X, Y = torch.randn(16, 8), torch.randint(0, 5, (16,))
    outs = []
    for step in range(2):
        red.zero_grad()
        micro = 2                                      # rank r, micro-batch i sees rows [8 r + 4 i, 8 r + 4 i + 4)
        for i in range(micro):
            xs, ys = X[8 * rank + 4 * i:8 * rank + 4 * i + 4], Y[8 * rank + 4 * i:8 * rank + 4 * i + 4]
            with (red.no_sync() if i + 1 < micro else contextlib.nullcontext()):
                (nn.functional.cross_entropy(m(xs), ys) / micro).backward()
        red.finish()
        outs.append({n: p.grad.detach().numpy().copy() for n, p in m.named_parameters() if p.grad is not None})
    # a counted backward after the reduction (forgot no_sync / zero_grad) must raise, not mix reduced and local gradients
    raised = False
    try:
        nn.functional.cross_entropy(m(X[:4]), Y[:4]).backward()
    except RuntimeError as e:
        raised = 'after it was reduced' in str(e)
    # ... and so must a micro-batch under no_sync() that follows finish() without zero_grad(): it would accumulate local gradients
    # on top of the already averaged buffers (round-2 advisor finding)
    raised2 = False
    try:
        with red.no_sync():
            nn.functional.cross_entropy(m(X[:4]), Y[:4]).backward()
    except RuntimeError as e:
        raised2 = 'after it was reduced' in str(e)
    raised = raised and raised2
    q.put((rank, outs, {n: v.numpy() for n, v in w0.items()}, raised))
    dist.barrier()
    dist.destroy_process_group()


def _run_accum(collective, port_base):
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = port_base + (os.getpid() % 2000)
    procs = [ctx.Process(target=_accum_worker, args=(r, 2, port, q, collective)) for r in range(2)]
    for p in procs:
        p.start()
    got = [q.get(timeout=120) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    res = {r: [{n: torch.from_numpy(v) for n, v in st.items()} for st in outs] for r, outs, _, _ in got}
    w = {r: w0 for r, _, w0, _ in got}
    assert all(raised for _, _, _, raised in got)
    for n in w[0]:
        assert (w[0][n] == w[1][n]).all(), f'{n}: replicas differ after broadcast_parameters'
    torch.manual_seed(100)                             # rank 0's initial weights = everyone's
    m = Toy()
    torch.manual_seed(1)
    X, Y = torch.randn(16, 8), torch.randint(0, 5, (16,))
    nn.functional.cross_entropy(m(X), Y).backward()     # mean over 16 rows == mean over (2 ranks x 2 micro-batches) of 4-row means
    for step in range(2):
        for n, p in m.named_parameters():
            if n.startswith('vae') or n.startswith('unused'):
                continue
            for r in (0, 1):
                torch.testing.assert_close(res[r][step][n], p.grad, rtol=1e-5, atol=1e-6)
        assert torch.equal(res[0][step]['to_logits.weight'], res[1][step]['to_logits.weight'])


def test_reducer_gradient_accumulation_no_sync_allreduce():
    _run_accum('allreduce', 31500)


def test_reducer_gradient_accumulation_no_sync_reduce_scatter_all_gather():
    _run_accum('rs_ag', 33500)


def _native_worker(rank, world, port, q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from nuwa_pytorch_amd.distributed import GradReducer
    try:
        GradReducer(Toy(), collective='native')
        q.put((rank, 'no error'))
    except RuntimeError as e:
        q.put((rank, str(e)))
    try:
        GradReducer(Toy(), collective='ring')
        q.put((rank, 'no error'))
    except ValueError:
        q.put((rank, 'ValueError'))
    dist.destroy_process_group()


def test_native_collective_refuses_cpu_buckets():
    """collective='native' puts the buckets on libamdnuwa's RCCL communicator: fp32 gradients on a HIP device only -- CPU replicas
    get a RuntimeError at construction, not a silent switch to another transport; an unknown name is a ValueError"""
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    ps = [ctx.Process(target=_native_worker, args=(r, 2, 29650 + (os.getpid() % 40), q)) for r in range(2)]
    for p in ps:
        p.start()
    got = [q.get(timeout=120) for _ in range(4)]
    for p in ps:
        p.join(timeout=60)
    msgs = sorted(m for _, m in got)
    assert msgs.count('ValueError') == 2 and sum('HIP device' in m for m in msgs) == 2, msgs
