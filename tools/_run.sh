export PYTHONUNBUFFERED=1 TMPDIR=/tmp
python -m pytest tests/test_gpu_modules.py -q -k "noncausal or g11 or g1_ or sketch" --tb=short 2>&1 | tail -25
