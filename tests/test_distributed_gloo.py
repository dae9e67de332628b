"""world_size-2 CPU test (gloo) of the data-parallel gradient reducer: bucketed flat gradients,
hook-driven async all-reduce, frozen / unused parameters, equality with the single-process gradient
of the full batch."""
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
