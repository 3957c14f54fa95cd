for i in 1 2 3 4; do python -m pytest tests/test_gpu_round2.py -q -x -k alternative 2>&1 | tail -4 | head -3; done
