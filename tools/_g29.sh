set -e
O=gpurun_out/r2; mkdir -p $O
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $O/pytest_full.log 2>&1 || { tail -40 $O/pytest_full.log; exit 1; }
tail -3 $O/pytest_full.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1 || { tail -20 $O/smoke.log; exit 1; }
tail -2 $O/smoke.log
