#!/bin/bash
# one GPU-box session: parity tests, smoke, bench.  Everything is logged under gpurun_out/.
mkdir -p gpurun_out
rm -f gpurun_out/parity_log.jsonl
export PYTHONUNBUFFERED=1
( rocminfo | grep -E "Marketing Name|Compute Unit|gfx" | head -6; nproc; free -g | head -2 ) > gpurun_out/box.txt 2>&1
timeout ${T_TESTS:-1200} python -m pytest tests -m gpu -q --timeout 300 -p no:cacheprovider ${PYTEST_ARGS:-} > gpurun_out/pytest_gpu.log 2>&1
echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
tail -n 60 gpurun_out/pytest_gpu.log
timeout 300 python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/smoke.log; tail -n 5 gpurun_out/smoke.log
timeout ${T_BENCH:-600} python bench.py ${BENCH_ARGS:---steps 5 --warmup 2 --batch 4} > gpurun_out/bench.log 2>&1; echo "bench rc=$?" >> gpurun_out/bench.log; tail -n 8 gpurun_out/bench.log
