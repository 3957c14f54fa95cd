set -e
O=gpurun_out/r2; mkdir -p $O
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "decoder or round_trip or lane" > $O/pytest_dec.log 2>&1 || { tail -40 $O/pytest_dec.log; exit 1; }
tail -2 $O/pytest_dec.log
L=$O/decode.log; rm -f $L
echo "== wavefront decoders, 4 GiB of text" >> $L
CW_DECODE_LANES=0 timeout -k 10 300 python tools/decode_probe.py 4096 65536 32768 >> $L 2>&1
echo "== lane decoders (wpc 8 / 4 / 16)" >> $L
timeout -k 10 300 python tools/decode_probe.py 4096 65536 32768 >> $L 2>&1
CW_LANES_WPC=4 timeout -k 10 300 python tools/decode_probe.py 4096 65536 >> $L 2>&1
CW_LANES_WPC=16 timeout -k 10 300 python tools/decode_probe.py 4096 65536 >> $L 2>&1
grep -v amdgpu.ids $L
