python - <<'PY'
import torch, time, subprocess, threading
a = torch.randn(8192, 8192, device='cuda', dtype=torch.bfloat16); b = torch.randn(8192, 8192, device='cuda', dtype=torch.bfloat16)
for n in (8192, 16384):
    a = torch.randn(n, 8192, device='cuda', dtype=torch.bfloat16); b = torch.randn(8192, 8192, device='cuda', dtype=torch.bfloat16)
    for _ in range(5): torch.matmul(a, b.t())
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    out = []
    def smi():
        time.sleep(0.3)
        out.append(subprocess.run(['rocm-smi', '--showclocks', '--showpower'], capture_output=True, text=True).stdout)
    th = threading.Thread(target=smi); th.start()
    s.record()
    iters = 400
    for _ in range(iters): torch.matmul(a, b.t())
    e.record(); torch.cuda.synchronize(); th.join()
    t = s.elapsed_time(e) / iters * 1e-3
    print(f'hipBLASLt bf16 {n}x8192x8192 NT: {t*1e6:.1f} us  {2*n*8192*8192/t/1e12:.0f} TF/s')
    print('\n'.join(l for l in out[0].splitlines() if 'sclk' in l or 'Power' in l or 'mclk' in l))
PY
