#!/bin/bash
# round 5: kernel stats of the side configurations (cfg 5 dual decoder b = 64, cfg 2 b = 512): are framework element-wise kernels hiding in them as they were in cfg 4?
mkdir -p gpurun_out; export PYTHONUNBUFFERED=1 TMPDIR=/tmp
R=$PWD
show() { f=$(find $1 -name "*kernel_stats.csv" | head -n 1); [ -n "$f" ] && python -c "
import csv
rows=[r for r in csv.reader(open('$f'))][1:]
tot=sum(float(r[2]) for r in rows)
fw=sum(float(r[2]) for r in rows if 'at::' in r[0] or 'rocclr' in r[0])
print('  total %.1f ms, framework kernels %.1f ms (%.1f %%)' % (tot/1e6, fw/1e6, 100*fw/tot))
for r in rows[:18]: print('  %-72s calls %6s total %9.1f ms avg %9.1f us  %5.2f %%' % (r[0].replace('(anonymous namespace)::','')[:72], r[1], float(r[2])/1e6, float(r[3])/1e3, 100*float(r[2])/tot))
"; }
echo "== cfg 5 (tools/cfg5_step.py --batch 64)"
( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prof_p5 -o st --output-format csv -- python $R/tools/cfg5_step.py --batch 64 ) > /tmp/prof_p5.log 2>&1; tail -n 1 /tmp/prof_p5.log; show /tmp/prof_p5
echo "== cfg 2 (bench.py --config cfg2 --batch 512)"
( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prof_p2 -o st --output-format csv -- python $R/bench.py --config cfg2 --batch 512 --steps 3 --warmup 1 --no-cpu-baseline --no-tokenizer --no-parity ) > /tmp/prof_p2.log 2>&1; show /tmp/prof_p2
