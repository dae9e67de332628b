#!/usr/bin/env python
"""Benchmark of the hot path BASELINE.json names: video-tokens/sec, NUWA 3DNA decoder forward+backward
at 10x16x16 video tokens (BASELINE cfg 3: dim 512, depth 24, 8 heads, 3DNA kernel (5,3,3), dilation
cycle (1,2,4), 256 text tokens of context), bf16 MFMA operands, synthetic data, random-init weights.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One "step" = embedding assemble -> 24 decoder layers -> StableLayerNorm -> to_logits -> cross entropy ->
backward to all decoder parameter gradients (+ the RCCL gradient all-reduce for N > 1), on `--batch`
samples per GPU (weak scaling).  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

CFGS = {
    'cfg3': dict(dim=512, dec_depth=24, heads=8, dim_head=64, frames=10, fmap=16, kernel=(5, 3, 3), dilation=(1, 2, 4),
                 text_len=256, codebook=8192, vae=dict(dim=64, image_size=256, num_layers=4)),
    'cfg4': dict(dim=512, dec_depth=64, heads=8, dim_head=64, frames=10, fmap=16, kernel=(5, 3, 3), dilation=(1, 2, 4),
                 text_len=256, codebook=8192, vae=dict(dim=64, image_size=256, num_layers=4), reversible=True),
    'cfg2': dict(dim=256, dec_depth=6, heads=8, dim_head=64, frames=4, fmap=16, kernel=(3, 3, 3), dilation=(1,),
                 text_len=256, codebook=512, vae=dict(dim=64, image_size=64, num_layers=2)),
}
PEAK_BF16_TFLOPS = 2500.0      # dense bf16 MFMA peak, MI355X_MICROARCH.md


def fwd_flops_per_sample(c):
    """ALGORITHMIC forward FLOPs per sample: GEMMs + attention cores (valid 3DNA taps only).
    Reversible stacks (cfg 4, np.py:1246-1277) carry a FeedForward after the 3DNA block AND after the cross-attention block."""
    D, h, d, T, C = c['dim'], c['heads'], c['dim_head'], c['text_len'], c['codebook']
    n = c['frames'] * c['fmap'] ** 2
    inner = h * d
    ffi = (D * 4 * 2) // 3
    proj3 = 2 * n * D * inner + 2 * (n + 1) * D * 2 * inner + 2 * n * inner * D
    projx = 2 * n * D * inner + 2 * T * D * 2 * inner + 2 * n * inner * D
    corex = 4 * h * n * (T + 1) * d + 2 * h * h * n * (T + 1)
    ff = 2 * n * D * 2 * ffi + 2 * n * ffi * D
    from nuwa_pytorch_amd.nuwa_pytorch import causal_neighbor_mask
    tot = 0
    for l in range(c['dec_depth']):
        dl = c['dilation'][l % len(c['dilation'])]
        m = causal_neighbor_mask((c['frames'], c['fmap'], c['fmap']), c['kernel'], (dl, dl, dl))
        valid = int((~m[:n - 1]).sum())
        tot += proj3 + (4 * h * d + 2 * h * h) * valid + projx + corex + ff * (2 if c.get('reversible') else 1)
    return tot + 2 * n * D * C


def build_model(c, device):
    import nuwa_pytorch_amd as A
    torch.manual_seed(0)
    vae = A.VQGanVAE(dim=c['vae']['dim'], image_size=c['vae']['image_size'], num_layers=c['vae']['num_layers'],
                     vq_codebook_size=c['codebook'], use_vgg_and_gan=False)
    nuwa = A.NUWA(vae=vae, dim=c['dim'], max_video_frames=c['frames'], text_max_seq_len=c['text_len'], text_enc_depth=1,
                  enc_reversible=True, dec_reversible=bool(c.get('reversible')), dec_depth=c['dec_depth'], dec_heads=c['heads'], dec_dim_head=c['dim_head'],
                  sparse_3dna_kernel_size=c['kernel'], sparse_3dna_dilation=c['dilation'], shift_video_tokens=True)
    return nuwa.to(device).train()


def decoder_params(nuwa):
    ps = list(nuwa.video_transformer.parameters()) + [nuwa.to_logits.weight, nuwa.video_bos] + \
        list(nuwa.image_embedding.parameters()) + list(nuwa.video_pos_emb.parameters())
    seen, out = set(), []
    for p in ps:
        if id(p) not in seen:
            seen.add(id(p)); out.append(p)
    return out


def decoder_step(nuwa, ids, ctx, mask):
    x = nuwa.embed_video(ids[:, :-1])
    h = nuwa.decode_hidden(x, ctx, mask)
    loss = nuwa._final(h, ids)
    loss.backward()
    return loss


def cpu_baseline(c, budget_layers=3):
    """the oracle (fp32 CPU restatement of the reference algorithm) on the host cores: b = 1, full cfg
    geometry; times `budget_layers` decoder layers (one per dilation) + embed/final-norm/logits/CE
    forward+backward and scales the layer time to the full depth."""
    from oracle import nuwa_oracle as O
    torch.set_num_threads(min(os.cpu_count() or 1, 32))     # beyond ~32 threads the fp32 oracle gets slower, not faster
    cores = torch.get_num_threads()
    D, h, d, T, C = c['dim'], c['heads'], c['dim_head'], c['text_len'], c['codebook']
    N = c['frames'] * c['fmap'] ** 2
    L = min(budget_layers, c['dec_depth'])
    import nuwa_pytorch_amd.nuwa_pytorch as M
    torch.manual_seed(0)
    tr = M.Transformer(dim=D, depth=L, causal=True, heads=h, dim_head=d, cross_attend=True, sparse_3dna_attn=True,
                       sparse_3dna_kernel_size=c['kernel'], sparse_3dna_video_shape=(c['frames'], c['fmap'], c['fmap']),
                       sparse_3dna_dilations=c['dilation'], shift_video_tokens=True)
    P = {k: v.clone().requires_grad_(v.is_floating_point()) for k, v in tr.state_dict().items()}
    cfg = dict(video_shape=(c['frames'], c['fmap'], c['fmap']), kernel_size=c['kernel'], dilations=c['dilation'], heads=h, depth=L, shift=True)
    g = torch.Generator().manual_seed(1234)
    x = torch.randn(1, N, D, generator=g, requires_grad=True)
    ctx = torch.randn(1, T, D, generator=g)
    mask = torch.ones(1, T, dtype=torch.bool)
    t0 = time.perf_counter()
    y = x
    for l in range(L):
        y = O.decoder_layer(y, O.sub(P, f'layers.{l}'), cfg, l, ctx, mask)
    y.sum().backward()
    t_layers = time.perf_counter() - t0
    # embedding + final norm + logits + CE
    E = {'image_embedding.embed.weight': torch.randn(C, D, requires_grad=True), 'video_bos': torch.randn(D, requires_grad=True),
         'video_pos_emb.axial1': torch.randn(c['frames'], D, requires_grad=True), 'video_pos_emb.axial2': torch.randn(c['fmap'], D, requires_grad=True),
         'video_pos_emb.axial3': torch.randn(c['fmap'], D, requires_grad=True)}
    wl = torch.randn(C, D, requires_grad=True)
    nw, nb = torch.ones(D, requires_grad=True), torch.zeros(D, requires_grad=True)
    ids = torch.randint(0, C, (1, N), generator=g)
    t0 = time.perf_counter()
    e = O.embed_assemble(ids[:, :-1], E)
    hn = O.stable_layer_norm(e, nw, nb)
    loss = torch.nn.functional.cross_entropy((hn @ wl.t()).reshape(-1, C), ids.reshape(-1))
    loss.backward()
    t_rest = time.perf_counter() - t0
    t_step = t_layers * c['dec_depth'] / L + t_rest
    return dict(value=N / t_step, unit='video-tokens/s', cores=cores, kind='port',
                sample=f'oracle fp32, b=1, {L} of {c["dec_depth"]} decoder layers fwd+bwd ({t_layers:.1f}s) scaled x{c["dec_depth"] / L:.0f} '
                       f'+ embed/norm/logits/CE ({t_rest:.1f}s) -> {t_step:.1f}s per step')


def traffic_from_profiles(config, b):
    """HBM bytes per NT-GEMM launch from the committed PMC passes (profiles/traffic.json, written by tools/pmc_summary.py
    from separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE runs of this same command); None when no pass matches."""
    try:
        with open(os.path.join(ROOT, 'profiles', 'traffic.json')) as f:
            t = json.load(f)
        e = t.get(f'{config}_b{b}')
        return e if e and 'bytes_per_launch' in e else None
    except (OSError, ValueError):
        return None


def tokenizer_rate(nuwa, c, b, dev):
    """frozen VQGanVAE tokenizer (get_video_indices, exact-fp32 HIP kernels) on b videos of raw frames; reported beside the
    decoder metric, never inside its timed region (the decoder bench feeds token ids, as a cached-token trainer would)."""
    v = c['vae']
    video = torch.rand(b, c['frames'], 3, v['image_size'], v['image_size'], device=dev)
    nuwa.vae.get_video_indices(video)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        ids = nuwa.vae.get_video_indices(video)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 3
    return {'frames_per_s': b * c['frames'] / dt, 'video_tokens_per_s': ids.numel() / dt, 'ms': dt * 1e3, 'videos': b, 'dtype': 'f32'}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--batch', type=int, default=64, help='samples per GPU (weak scaling)')
    ap.add_argument('--config', default='cfg3', choices=list(CFGS))
    ap.add_argument('--precision', default='bf16', choices=['bf16', 'bf16x3'])
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--backend', default='nccl', choices=['nccl', 'gloo'], help='nccl = RCCL over xGMI (default); gloo only for functional checks')
    ap.add_argument('--single-device', action='store_true', help='functional check only: every rank uses cuda:0 (with --backend gloo)')
    ap.add_argument('--no-tokenizer', action='store_true', help='skip the (untimed, separately reported) frozen-VAE tokenizer rate')
    args = ap.parse_args()

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs a HIP device (the decoder path has no CPU fallback)')
    if args.single_device:
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group(args.backend, rank=rank, world_size=world)
    import nuwa_pytorch_amd as A
    from nuwa_pytorch_amd import kernels as K
    from nuwa_pytorch_amd.distributed import GradReducer
    A.set_precision(args.precision)
    c = CFGS[args.config]
    nuwa = build_model(c, dev)
    if world > 1:
        for p in nuwa.parameters():
            dist.broadcast(p.data, src=0)
    params = decoder_params(nuwa)
    for p in nuwa.parameters():
        p.requires_grad_(False)
    for p in params:
        p.requires_grad_(True)
    reducer = GradReducer(nuwa) if world > 1 else None

    b = args.batch
    N = c['frames'] * c['fmap'] ** 2
    g = torch.Generator(device='cpu').manual_seed(1234 + rank)
    ids = torch.randint(0, c['codebook'], (b, N), generator=g).to(dev)
    ctx = torch.randn(b, c['text_len'], c['dim'], generator=g).to(dev)
    mask = torch.ones(b, c['text_len'], dtype=torch.bool)
    mask[:, -64:] = torch.rand(b, 64, generator=g) > 0.5        # exercise the key mask
    mask = mask.to(dev)

    def step():
        if reducer is not None:
            reducer.zero_grad()
        else:
            for p in params:
                p.grad = None
        loss = decoder_step(nuwa, ids, ctx, mask)
        if reducer is not None:
            reducer.finish()
        return loss

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    if rank == 0:
        K.timer_arm(True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    gemm_ms, gemm_launches, gemm_flops, gemm_bytes = (0.0, 0, 0.0, 0.0)
    if rank == 0:
        gemm_ms, gemm_launches, gemm_flops, gemm_bytes = K.timer_collect()
        K.timer_arm(False)
    tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())

    if rank == 0:
        tokens = world * b * N * args.steps
        value = tokens / dt
        fl = fwd_flops_per_sample(c)
        step_flops = 3.0 * fl * b
        ach = gemm_flops / (gemm_ms * 1e-3) / 1e12 if gemm_ms > 0 else 0.0
        out = {
            'metric': 'video-tokens/sec, 3DNA decoder fwd+bwd @ 10x16x16' if args.config == 'cfg3' else 'video-tokens/sec, 3DNA decoder fwd+bwd',
            'value': value, 'unit': 'video-tokens/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': dt / args.steps * 1e3, 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': 'bf16' if args.precision == 'bf16' else 'bf16x3', 'data': 'synthetic',
            'config': {'workload': f'BASELINE {args.config}: NUWA decoder dim={c["dim"]} depth={c["dec_depth"]} heads={c["heads"]}, '
                                   f'{c["frames"]}x{c["fmap"]}x{c["fmap"]} video tokens, 3DNA kernel {c["kernel"]} dilation {c["dilation"]}, '
                                   f'{c["text_len"]} text tokens, codebook {c["codebook"]}',
                       'per_gpu_batch': b, 'global_batch': b * world, 'tokens_per_sample': N, 'parallelism': f'dp{world}',
                       'loss': float(loss.detach())},
            'per_gpu_value': value / world,
            'peak_hbm_gb': torch.cuda.max_memory_allocated(dev) / 2 ** 30,
            'step_tflops_per_gpu': step_flops / (dt / args.steps) / 1e12,
            'step_mfma_frac': step_flops / (dt / args.steps) / 1e12 / PEAK_BF16_TFLOPS,
            'roofline': {'bound': 'mfma', 'kernel': 'gemm_nt_* (bf16 MFMA NT GEMM; every amdnuwa_gemm_nt launch of the timed region, HIP events on the launch stream)',
                         'achieved': ach, 'peak': PEAK_BF16_TFLOPS, 'unit': 'TFLOP/s', 'frac': ach / PEAK_BF16_TFLOPS,
                         'traffic': None, 'algorithmic_bytes_per_launch': gemm_bytes / max(gemm_launches, 1),
                         'flops_per_launch': gemm_flops / max(gemm_launches, 1), 'launches': gemm_launches, 'avg_launch_us': gemm_ms * 1e3 / max(gemm_launches, 1),
                         'share_of_step': gemm_ms * 1e-3 / dt},
        }
        tr = traffic_from_profiles(args.config, b)
        if tr is not None:
            out['roofline']['traffic'] = tr['bytes_per_launch']
            out['roofline']['traffic_source'] = tr['source']
        if not args.no_tokenizer and world == 1:          # (side measurement, single-GPU runs only: keeps the ranks in step)
            out['vae_tokenizer'] = tokenizer_rate(nuwa, c, min(b, 8), dev)
        if world == 1 and not args.no_cpu_baseline:
            try:
                out['cpu_baseline'] = cpu_baseline(c)
            except Exception as e:      # the baseline is a report, never a reason to lose the GPU number
                out['cpu_baseline'] = {'value': None, 'unit': 'video-tokens/s', 'cores': os.cpu_count(), 'kind': 'port',
                                       'sample': f'failed: {type(e).__name__}: {e}'}
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
