"""Data-parallel step on the real HIP path: two processes (gloo, both on cuda:0 -- the collective library is not the point
here, the reducer's bucket / hook / stream logic around the libamdnuwa autograd Functions is) each run the tiny NUWA
decoder step on their half of the batch; the all-reduced gradients must equal the full-batch gradient of one process."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))


def _model(A):
    torch.manual_seed(0)
    vae = A.VQGanVAE(dim=32, image_size=16, num_layers=2, vq_codebook_size=64, vq_codebook_dim=32, use_vgg_and_gan=False)
    nuwa = A.NUWA(vae=vae, dim=32, text_num_tokens=50, text_max_seq_len=8, max_video_frames=3, text_enc_depth=2,
                  dec_depth=3, enc_reversible=True, dec_heads=2, dec_dim_head=32, text_enc_heads=2, text_enc_dim_head=16,
                  sparse_3dna_kernel_size=3, sparse_3dna_dilation=(1, 2))
    return nuwa.cuda().train()


def _data():
    g = torch.Generator().manual_seed(5)
    text = torch.randint(1, 50, (4, 8), generator=g)
    vid = torch.randint(0, 64, (4, 3, 4, 4), generator=g)
    return text, vid


def _grads(nuwa):
    return {n: p.grad.detach().float().cpu().numpy().copy() for n, p in nuwa.named_parameters()
            if p.grad is not None and not n.startswith('vae.')}


def _worker(rank, world, port, q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    import nuwa_pytorch_amd as A
    from nuwa_pytorch_amd.distributed import GradReducer
    A.set_precision('bf16x3')
    nuwa = _model(A)
    red = GradReducer(nuwa)
    text, vid = _data()
    sl = slice(rank * 2, rank * 2 + 2)
    out = []
    for step in range(2):
        red.zero_grad()
        loss = nuwa(text=text[sl].cuda(), video=vid[sl].cuda(), return_loss=True, cond_dropout_prob=0.)
        loss.backward()
        red.finish()
        torch.cuda.synchronize()
        out.append(_grads(nuwa))
    q.put((rank, out, len(red.buckets)))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_step_equals_full_batch():
    if not torch.cuda.is_available():
        pytest.skip('no GPU')
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 29600 + (os.getpid() % 300)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = {}
    for _ in range(2):
        r, out, nb = q.get(timeout=300)
        res[r] = out
        assert nb >= 4                     # 3 decoder layers + embeddings/logits (+ text encoder)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    import nuwa_pytorch_amd as A
    A.set_precision('bf16x3')
    try:
        nuwa = _model(A)
        text, vid = _data()
        nuwa(text=text.cuda(), video=vid.cuda(), return_loss=True, cond_dropout_prob=0.).backward()
        ref = _grads(nuwa)
    finally:
        A.set_precision('bf16')
    import numpy as np
    checked = 0
    for step in range(2):
        for n, g in ref.items():
            a, b = res[0][step][n], res[1][step][n]
            assert np.array_equal(a, b), f'{n}: ranks disagree after the all-reduce'
            scale = max(float(np.abs(g).max()), 1e-12)
            assert float(np.abs(a - g).max()) <= 2e-3 * scale + 1e-9, f'{n}: step {step} rel err {float(np.abs(a - g).max()) / scale:.2e}'
            checked += 1
    assert checked > 80


@pytest.mark.parametrize('launcher', ['self', 'torch.distributed.run'])
def test_bench_py_two_ranks_end_to_end(launcher):
    """the N > 1 code path of bench.py ITSELF (launch contract, flat parameter broadcast, GradReducer around the timed steps, max over
    ranks, one JSON line from rank 0): two ranks, gloo backend, both on cuda:0 -- started the way the driver starts N = 1
    (`python bench.py --gpus 2`: bench.py re-executes itself under torch.distributed.run) and through the launcher."""
    if not torch.cuda.is_available():
        pytest.skip('no GPU')
    import json
    import subprocess
    port = 29900 + (os.getpid() % 90)
    tail = ['--gpus', '2', '--steps', '2', '--warmup', '1', '--batch', '2', '--side-batch', '1', '--backend', 'gloo', '--single-device',
            '--no-tokenizer', '--no-cpu-baseline']
    if launcher == 'self':
        cmd = [sys.executable, os.path.join(ROOT, 'bench.py')] + tail
    else:
        cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
               '--master-port', str(port), os.path.join(ROOT, 'bench.py')] + tail
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK')}
    env.update(MASTER_ADDR='127.0.0.1', OMP_NUM_THREADS='4')
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith('{"metric"')]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out['n_gpus'] == 2 and out['steps'] == 2 and out['scaling'] == 'weak'
    assert out['config']['global_batch'] == 4 and out['config']['parallelism'] == 'dp2'
    assert out['value'] > 0 and abs(out['value'] - 2 * out['per_gpu_value']) < 1e-6 * out['value']
    assert abs(out['value'] - 2 * 2 * 2560 * 2 / (out['ms_per_step'] * 2e-3)) < 1e-3 * out['value']
    assert out['roofline']['families']['gemm_nt']['launches'] > 0 and 0 < out['roofline']['frac'] < 1 and \
        abs(out['roofline']['frac'] - out['step_mfma_frac']) < 1e-12      # the line's roofline IS the whole step
    assert out['precision_mode'] == 'bf16x3-fwd' and out['fast_mode']['dtype'] == 'bf16' and out['fast_mode']['value'] > 0
    assert 'single fp16 MFMA on' in out['dtype'] and out['fp16_forward_parts'] == {'cores': True, 'ff': True, 'qkv': True, 'two_mfma_products': 'oq'} and \
        '2 fp16 MFMAs (fp16 activation x fp16 hi+lo weight) on to_out x2, cross-attention q projection' in out['dtype']
    assert 8.5 < out['config']['loss'] < 10.0                      # ~ln(8192) at random init


def test_bench_py_refuses_more_gpus_than_the_box_has():
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK')}
    n = torch.cuda.device_count() + 1
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', str(n), '--steps', '1', '--warmup', '0'], capture_output=True,
                       text=True, timeout=600, cwd=ROOT, env=env)
    assert r.returncode != 0 and f'--gpus {n}' in (r.stderr + r.stdout)


def _rccl_worker(q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(29700 + (os.getpid() % 200))
    torch.cuda.set_device(0)
    dist.init_process_group('nccl', rank=0, world_size=1)
    import nuwa_pytorch_amd as A
    from nuwa_pytorch_amd.distributed import GradReducer
    A.set_precision('bf16x3')
    nuwa = _model(A)
    text, vid = _data()
    nuwa(text=text.cuda(), video=vid.cuda(), return_loss=True, cond_dropout_prob=0.).backward()
    torch.cuda.synchronize()
    ref = _grads(nuwa)
    out = {}
    for coll in ('allreduce', 'rs_ag', 'native', 'native_rs_ag'):
        nuwa.zero_grad(set_to_none=True)
        red = GradReducer(nuwa, collective=coll, always_reduce=True)
        red.zero_grad()
        nuwa(text=text.cuda(), video=vid.cuda(), return_loss=True, cond_dropout_prob=0.).backward()
        red.finish()
        torch.cuda.synchronize()
        got = _grads(nuwa)
        import numpy as np
        # (not bit-equal: the text encoder's 16-wide heads and the text embedding run PyTorch-ROCm ops whose backward uses atomics)
        worst = max(float(np.abs(got[n] - ref[n]).max()) / max(float(np.abs(ref[n]).max()), 1e-12) for n in ref)
        out[coll] = (worst, red.native is not None, len(red.buckets))
        red.remove()
    q.put(out)
    dist.destroy_process_group()


def test_reducer_collectives_on_rccl_one_rank():
    """every exchange form of the reducer on the REAL backend (RCCL through torch.distributed and through libamdnuwa's own
    communicator), one rank on the one device a test box has: with always_reduce the bucket collectives run (AVG all-reduce; the
    in-place reduce-scatter + all-gather on the padded store, shard aliasing included) on the side stream from the gradient hooks
    and must hand back exactly the local gradients."""
    if not torch.cuda.is_available():
        pytest.skip('no GPU')
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    p = ctx.Process(target=_rccl_worker, args=(q,))
    p.start()
    out = q.get(timeout=600)
    p.join(timeout=120)
    assert p.exitcode == 0
    for coll, (worst, native, nb) in out.items():
        assert worst <= 1e-5, f'{coll}: reduced gradients differ from the local ones in a world of one (rel {worst:.2e})'
        assert native == coll.startswith('native') and nb >= 4


def test_native_comm_single_rank():
    """libamdnuwa's own RCCL communicator (amdnuwa_comm_*): id, init on this device, the three collectives the reducer uses, destroy.
    One rank is all a 1-GPU box allows (RCCL refuses two ranks on one device): a world of 1 must leave the buffers as they are."""
    from nuwa_pytorch_amd.distributed import NativeComm
    comm = NativeComm()
    assert (comm.rank, comm.world) == (0, 1)
    assert comm._lib.amdnuwa_comm_rank(comm._h) == 0 and comm._lib.amdnuwa_comm_world(comm._h) == 1
    x = torch.randn(1 << 20, device='cuda')
    ref = x.clone()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    comm.allreduce(x, average=True, stream=side)
    comm.reduce_scatter_allgather(x, average=False, stream=side)
    comm.broadcast(x, root=0, stream=side)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    assert torch.equal(x, ref)
    with pytest.raises(RuntimeError):
        comm._rc(comm._lib.amdnuwa_comm_broadcast(comm._h, x.data_ptr(), 16, 3, None), 'amdnuwa_comm_broadcast')      # root outside the world
    comm.close()


def _nccl_avg_worker(port, q):
    try:
        os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY='0')
        torch.cuda.set_device(0)
        dist.init_process_group('nccl', rank=0, world_size=1)
        x = torch.randn(1 << 16, device='cuda')
        ref = x.clone()
        dist.all_reduce(x, op=dist.ReduceOp.AVG, async_op=True).wait()
        out = torch.empty_like(x)
        dist.reduce_scatter_tensor(out, x, op=dist.ReduceOp.AVG)
        dist.all_gather_into_tensor(x, out)
        torch.cuda.synchronize()
        q.put(bool(torch.equal(x, ref)))
        dist.destroy_process_group()
    except Exception as e:      # noqa: BLE001
        q.put(repr(e))


def test_rccl_backend_accepts_the_average_op():
    """GradReducer averages INSIDE the collective on the RCCL backend (ReduceOp.AVG for all_reduce and reduce_scatter_tensor): a
    one-rank process group on the real backend must accept both ops and leave the data unchanged"""
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    p = ctx.Process(target=_nccl_avg_worker, args=(29700 + (os.getpid() % 90), q))
    p.start()
    res = q.get(timeout=300)
    p.join(timeout=60)
    assert res is True, res
