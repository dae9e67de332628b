for t in "13=1" "14=-1" "14=-1,7=8" "13=1,7=8" "14=0"; do
  echo "== $t"; AMDNUWA_TUNING="$t" python bench.py --no-cpu-baseline --no-tokenizer --no-parity --steps 6 --warmup 2 2>&1 | tail -1 | cut -c1-160
done
