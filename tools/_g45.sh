set -e
O=gpurun_out/r2; mkdir -p $O
timeout -k 10 500 python tools/soak_side_by_side.py > $O/soak_a.txt 2> $O/soak_a.err || { tail -5 $O/soak_a.err; tail -3 $O/soak_a.txt; exit 1; }
CW_LZ4_LANES=0 CW_LZF_LANES=0 timeout -k 10 500 python tools/soak_side_by_side.py > $O/soak_b.txt 2> $O/soak_b.err || { tail -5 $O/soak_b.err; exit 1; }
if diff $O/soak_a.txt $O/soak_b.txt > $O/soak_diff.txt; then echo "soak: $(wc -l < $O/soak_a.txt) cases identical"; else echo "soak: DIFFERENCES"; cat $O/soak_diff.txt; fi
grep -c lanes $O/soak_a.err || true
