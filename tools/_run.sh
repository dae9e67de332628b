timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_modules.py -x -q -k "s3 or sparse or 3dna or Sparse" 2>&1 | tail -3
R=$PWD; export TMPDIR=/tmp; OUT=/tmp/prof; mkdir -p $OUT
( cd /tmp && rocprofv3 --kernel-trace --stats -d $OUT -o ab -- python $R/tools/attn_bench.py --batch 64 ) > /tmp/log 2>&1
python tools/rocpd_stats.py $OUT/ab_results.db 2>&1 | grep "s3_bwd_fin\|xattn_unpack\|xattn_pack\|colsum\|partial_red\|splitk" | head -8 | cut -c1-150
