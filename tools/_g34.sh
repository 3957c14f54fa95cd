set -e
O=gpurun_out/r2; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q -k "side_by_side" > $O/pytest_sbs.log 2>&1 || { tail -40 $O/pytest_sbs.log; exit 1; }
tail -2 $O/pytest_sbs.log
