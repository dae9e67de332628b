export PYTHONUNBUFFERED=1 TMPDIR=/tmp
python -m pytest tests/test_gpu_modules.py tests/test_gpu_named_size.py tests/test_gpu_decode.py -q -k "audio or g9 or cfg5 or dual or g4 or 2dna" --tb=short 2>&1 | tail -25
python tools/cfg5_step.py 2>&1 | tail -3
