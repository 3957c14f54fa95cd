set -e
O=gpurun_out/r2; mkdir -p $O
L=$O/lzf_round.log; rm -f $L
P="timeout -k 10 200 python tools/perf_probe.py --alg none --iters 2 --comp lzf --bs 4096"
for nb in 1048576 262144 98304; do
for d in text mixed random; do
for rr in "8192 16384" "16384 16384" "32768 16384" "32768 8192" "32768 32768"; do set -- $rr
echo "== lzf $d 4K x $nb round $1 reserve $2" >> $L
CW_LZF_ROUND=$1 CW_LANES_RESERVE=$2 $P --data $d --nb $nb >> $L 2>&1
done; done; done
grep -v amdgpu.ids $L | sed 's/lib=libcwhc.so alg=none //; s/ | kernel ms.*//' | grep -A1 "^==" | grep -v "^--" | paste - - | sed 's/comp=.*pass//; s/ratio.*//'
