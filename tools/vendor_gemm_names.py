#!/usr/bin/env python
"""Which kernels does the vendor library (hipBLASLt through torch.matmul) pick for the decoder's NT shapes?  Run under
rocprofv3 --kernel-trace --stats: the kernel NAMES carry the macro tile, MFMA shape and staging options.  Comparison point only."""
import sys
import torch
b = int(sys.argv[1]) if len(sys.argv) > 1 else 128
M = b * 2560
shapes = [('qkv', 1536, 512), ('to_out/ff2-like', 512, 512), ('ff1', 2752, 512), ('ff2', 512, 1376), ('dgrad_qkv', 512, 1536), ('dgrad_ff1', 512, 2752),
          ('logits', 8192, 512), ('dgrad_logits', 512, 8192)]
for name, N, K in shapes:
    a = torch.randn(M, K, device='cuda', dtype=torch.bfloat16)
    w = torch.randn(N, K, device='cuda', dtype=torch.bfloat16)
    for _ in range(3):
        c = a @ w.t()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(5):
        c = a @ w.t()
    e.record()
    torch.cuda.synchronize()
    t = s.elapsed_time(e) / 5
    print(f'{name:18s} M={M} N={N} K={K}: {t * 1e3:8.1f} us  {2.0 * M * N * K / t / 1e9:7.1f} TFLOP/s', flush=True)
    del a, w, c
