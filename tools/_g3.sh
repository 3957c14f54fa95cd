mkdir -p gpurun_out/r2
python tests/debug_lz4_diff.py 65536 > gpurun_out/r2/diff64k.log 2>&1; cat gpurun_out/r2/diff64k.log | tail -12
CW_LZ4_HEADW=64 python tests/debug_lz4_diff.py 65536 > gpurun_out/r2/diff64k_h64.log 2>&1; tail -4 gpurun_out/r2/diff64k_h64.log
CW_LZ4_PARSE=v2 python tests/debug_lz4_diff.py 65536 2>&1 | tail -2
python tests/debug_lz4_diff.py 16384 alice29.txt 2>&1 | tail -4
