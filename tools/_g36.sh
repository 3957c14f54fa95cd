set -e
O=gpurun_out/r2; mkdir -p $O
echo "== default"; timeout -k 10 300 python tools/_lzf_race.py 2>&1 | grep -v amdgpu
echo "== lanes off"; CW_LZF_LANES=0 timeout -k 10 300 python tools/_lzf_race.py 2>&1 | grep -v amdgpu
timeout -k 10 900 python -m pytest tests -m gpu -x -q -k "side_by_side or lzf" > $O/pytest_sbs.log 2>&1 || { tail -40 $O/pytest_sbs.log; exit 1; }
tail -2 $O/pytest_sbs.log
