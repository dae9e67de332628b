#!/bin/bash
# round 4 side measurements at batches that fill the GPU: cfg 2 / cfg 4 bench lines with their roofline objects, cfg 5 and
# the whole reference step (text encoder + tokenizer + decoder + clip / AdamW) in the package default mode ('bf16x3-fwd')
TAG=${TAG:-r04}
mkdir -p gpurun_out; export PYTHONUNBUFFERED=1 TMPDIR=/tmp
ARGS="--no-cpu-baseline --no-tokenizer --no-parity --steps 5 --warmup 2"
{ echo "# python bench.py --config cfg4 --batch 64 $ARGS"; timeout 900 python bench.py --config cfg4 --batch 64 $ARGS 2>&1 | tail -n 1
  echo "# python bench.py --config cfg4 --batch 96 $ARGS"; timeout 900 python bench.py --config cfg4 --batch 96 $ARGS 2>&1 | tail -n 1
  echo "# python bench.py --config cfg2 --batch 256 $ARGS"; timeout 600 python bench.py --config cfg2 --batch 256 $ARGS 2>&1 | tail -n 1
  echo "# python bench.py --config cfg2 --batch 512 $ARGS"; timeout 600 python bench.py --config cfg2 --batch 512 $ARGS 2>&1 | tail -n 1; } > gpurun_out/side_${TAG}_bench_configs.txt 2>&1
{ echo "# python tools/cfg5_step.py --batch 64"; timeout 600 python tools/cfg5_step.py --batch 64 2>&1 | tail -n 2
  echo "# python tools/cfg5_step.py --batch 32 --reversible"; timeout 600 python tools/cfg5_step.py --batch 32 --reversible 2>&1 | tail -n 2
  echo "# python tools/full_step.py --batch 128"; timeout 900 python tools/full_step.py --batch 128 2>&1 | tail -n 2
  echo "# python tools/full_step.py --batch 96 --optimizer"; timeout 900 python tools/full_step.py --batch 96 --optimizer 2>&1 | tail -n 2; } > gpurun_out/side_${TAG}_steps.txt 2>&1
cut -c1-420 gpurun_out/side_${TAG}_bench_configs.txt gpurun_out/side_${TAG}_steps.txt
