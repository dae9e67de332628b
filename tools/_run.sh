export PYTHONUNBUFFERED=1 TMPDIR=/tmp
python -m pytest tests/test_gpu_modules.py -q -k "sparse_cross_2dna or g11 or sketch" --tb=short 2>&1 | tail -40
