set -e
O=gpurun_out/r2; mkdir -p $O
timeout -k 10 300 tools/random_line 4096 > $O/random_line.txt 2>&1
cat $O/random_line.txt
