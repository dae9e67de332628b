#!/bin/bash
# round 6, call a: xattn6 forward -- phase probes (tuning key 18) and two SQ counter passes of tools/xattn6_bench.py
mkdir -p gpurun_out; export PYTHONUNBUFFERED=1 TMPDIR=/tmp
R=$PWD
python - <<'PY' > gpurun_out/r06a_probe.txt 2>&1
import sys, torch
sys.path.insert(0, '.')
from nuwa_pytorch_amd import kernels as K, _lib
L = _lib.lib()
b, n, heads, dh, T = 128, 2560, 8, 64, 256
g = K.x_geom(b, n, T, heads, dh)
q16 = torch.randn(b * n, 512, device='cuda').half(); kv16 = torch.randn(b * T, 1024, device='cuda').half()
nk, nv = torch.randn(8, 64, device='cuda'), torch.randn(8, 64, device='cuda')
wth = (torch.randn(8, 8, device='cuda') * 0.3 + torch.eye(8, device='cuda')).contiguous()
mask = (torch.rand(b, T, device='cuda') > 0.2).to(torch.uint8)
pk = K.xattn6_pack(g, kv16, mask)
def bench(fn, it=10):
    for _ in range(2): fn()
    torch.cuda.synchronize(); s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it): fn()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e) / it * 1e3
for rnd in range(2):
    row = []
    for v, name in ((0, 'full'), (1, 'no pass 1'), (2, 'no pass 2'), (3, 'neither')):
        L.amdnuwa_set_tuning(18, v)
        row.append(f'{name} {bench(lambda: K.xattn6_fwd(g, q16, pk, nk, nv, wth, o_f16=True)):7.1f}')
    L.amdnuwa_set_tuning(18, 0)
    print('xattn6_fwd: ' + ' | '.join(row))
PY
cat gpurun_out/r06a_probe.txt
for pass in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS" "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_MISC SQ_INST_LEVEL_LDS"; do
  n=$(echo $pass | cut -d' ' -f1)
  O=$R/gpurun_out/pmc_r06a_$n; rm -rf $O; mkdir -p $O
  ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc $pass -d $O -o pmc --output-format csv -- python $R/tools/xattn6_bench.py --batch 128 --iters 3 ) > gpurun_out/pmc_r06a_$n.log 2>&1
  echo "pmc pass [$pass] rc=$?"
done
python tools/pmc_summary.py gpurun_out/pmc_r06a_*/pmc_counter_collection.csv > gpurun_out/r06a_pmc.txt 2>&1
cat gpurun_out/r06a_pmc.txt | cut -c1-400
find gpurun_out/pmc_r06a_* -name "*.csv" -size +8M -delete
