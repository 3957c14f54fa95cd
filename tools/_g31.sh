set -e
O=gpurun_out/r2; mkdir -p $O
L=$O/decode2.log; rm -f $L
for bs in 4096 8192 16384; do
echo "== $bs B blocks, 2 GiB: wavefront / lanes" >> $L
CW_DECODE_LANES=0 timeout -k 10 300 python tools/decode_probe.py 2048 $bs >> $L 2>&1
CW_DECODE_LANES=1 CW_DECODE_LANES_BYTES=1 timeout -k 10 300 python tools/decode_probe.py 2048 $bs >> $L 2>&1
done
for mib in 128 256 512 1024; do
echo "== 64 KiB blocks, $mib MiB: wavefront / lanes" >> $L
CW_DECODE_LANES=0 timeout -k 10 300 python tools/decode_probe.py $mib 65536 >> $L 2>&1
CW_DECODE_LANES=1 timeout -k 10 300 python tools/decode_probe.py $mib 65536 >> $L 2>&1
done
grep -v amdgpu.ids $L
