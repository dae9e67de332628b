R=$PWD; export TMPDIR=/tmp
for t in "9=0" "9=8"; do
OUT=/tmp/prof_$t; mkdir -p $OUT
( cd /tmp && AMDNUWA_TUNING="$t" rocprofv3 --kernel-trace --stats -d $OUT -o ab -- python $R/tools/attn_bench.py --batch 64 ) > /tmp/log 2>&1
echo "== $t"; python tools/rocpd_stats.py $OUT/ab_results.db 2>&1 | grep "s3_bwd" | head -8 | cut -c1-150
done
