#!/bin/bash
# the side measurements DESIGN.md quotes, one log each under gpurun_out/ (copy the ones to keep into profiles/)
TAG=${TAG:-r02}
mkdir -p gpurun_out; export PYTHONUNBUFFERED=1 TMPDIR=/tmp
ARGS="--no-cpu-baseline --no-tokenizer --no-parity --steps 5 --warmup 2"
{ echo "# python bench.py --config cfg2 --batch 256 $ARGS"; python bench.py --config cfg2 --batch 256 $ARGS 2>&1 | tail -n 1
  echo "# python bench.py --config cfg4 --batch 8 $ARGS"; python bench.py --config cfg4 --batch 8 $ARGS 2>&1 | tail -n 1
  echo "# python bench.py --config cfg3 --precision bf16x3 --batch 16 $ARGS"; python bench.py --precision bf16x3 --batch 16 $ARGS 2>&1 | tail -n 1
  echo "# python bench.py --batch 128 $ARGS"; python bench.py --batch 128 $ARGS 2>&1 | tail -n 1; } > gpurun_out/side_${TAG}_bench_configs.txt 2>&1
{ echo "# python tools/cfg5_step.py --batch 32"; python tools/cfg5_step.py --batch 32 2>&1 | tail -n 2
  echo "# python tools/cfg5_step.py --batch 16 --reversible"; python tools/cfg5_step.py --batch 16 --reversible 2>&1 | tail -n 2
  echo "# python tools/full_step.py --batch 16"; python tools/full_step.py --batch 16 2>&1 | tail -n 3
  echo "# python tools/full_step.py --batch 16 --optimizer"; python tools/full_step.py --batch 16 --optimizer 2>&1 | tail -n 3; } > gpurun_out/side_${TAG}_steps.txt 2>&1
python tools/gen_bench.py --batch 4 > gpurun_out/side_${TAG}_gen_bench.txt 2>&1
python tools/attn_bench.py --batch 64 > gpurun_out/side_${TAG}_attn_bench_b64.txt 2>&1
python tools/gemm_probe.py 64 7 > gpurun_out/side_${TAG}_gemm_probe_b64.txt 2>&1
python tools/ew_bench.py 64 > gpurun_out/side_${TAG}_ew_bench_b64.txt 2>&1
python tools/vae_bench.py > gpurun_out/side_${TAG}_vae_bench.txt 2>&1
grep -h "value\|tokens/s\|ms" gpurun_out/side_${TAG}_bench_configs.txt gpurun_out/side_${TAG}_steps.txt | cut -c1-260
