#!/bin/bash
# per-change check on the MI355X: the whole GPU suite + the default bench line (TAG names the logs under gpurun_out/)
TAG=${TAG:-r02}
mkdir -p gpurun_out; export PYTHONUNBUFFERED=1 TMPDIR=/tmp
rm -f gpurun_out/parity_log.jsonl gpurun_out/named_size.json
timeout ${PYTEST_TIMEOUT:-1500} python -m pytest tests -m gpu -q --tb=short ${PYTEST_ARGS:-} > gpurun_out/pytest_$TAG.txt 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_$TAG.txt
tail -n 25 gpurun_out/pytest_$TAG.txt
timeout 900 python bench.py ${BENCH_ARGS:-} > gpurun_out/bench_$TAG.log 2>&1; echo "bench rc=$?"
tail -n 1 gpurun_out/bench_$TAG.log | cut -c1-3000
