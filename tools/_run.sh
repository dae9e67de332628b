for t in "11=0" "11=1" "11=0" "11=1"; do echo "== $t"; AMDNUWA_TUNING="$t" python bench.py --no-cpu-baseline --no-tokenizer --no-parity --steps 8 --warmup 2 2>&1 | tail -1 | cut -c60-200; done
