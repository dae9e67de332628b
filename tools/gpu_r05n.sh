#!/bin/bash
# round 5: cfg 4 (depth 64, reversible, b = 64) had lost 4 % against round 4 (1411 -> 1467 ms): one amax pass per reversible block for a gradient
# scale nobody read.  Fixed; and the reconstruction x2 = y2 - g(y1) subtracts inside the post-norm kernel (four negation passes per block gone).
mkdir -p gpurun_out; export PYTHONUNBUFFERED=1 TMPDIR=/tmp
R=$PWD
timeout 1500 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_modules.py tests/test_gpu_decode.py -q --tb=short -k "layernorm or ln_ or reversible or g6 or g9 or g11 or dual or memory" 2>&1 | tail -n 5
line() { timeout 900 python bench.py --config cfg4 --batch 64 --no-cpu-baseline --no-tokenizer --no-parity --steps 3 --warmup 1 2>/dev/null | tail -n 1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1:', round(d['ms_per_step'],1), 'ms/step', round(d['value']), d['unit'])"; }
line "cfg 4, this tree             "
AMDNUWA_BWD_F16=0 line "cfg 4, fp16-gradient backward off"
timeout 600 python tools/cfg5_step.py --batch 64 2>&1 | tail -n 1
( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prof_n -o st --output-format csv -- python $R/bench.py --config cfg4 --batch 64 --steps 2 --warmup 1 --no-cpu-baseline --no-tokenizer --no-parity ) > /tmp/prof_n.log 2>&1
f=$(find /tmp/prof_n -name "*kernel_stats.csv" | head -n 1); [ -n "$f" ] && python -c "
import csv
rows=[r for r in csv.reader(open('$f'))][1:]
tot=sum(float(r[2]) for r in rows)
for r in rows[:24]: print('%-70s calls %6s total %9.1f ms avg %9.1f us  %5.2f %%' % (r[0].replace('(anonymous namespace)::','')[:70], r[1], float(r[2])/1e6, float(r[3])/1e3, 100*float(r[2])/tot))
"
