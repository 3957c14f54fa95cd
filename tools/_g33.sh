set -e
O=gpurun_out/r2; mkdir -p $O
SECONDS=0; python bench.py > $O/bench_rt.json 2> $O/bench_rt.err || { tail -20 $O/bench_rt.err; exit 1; }
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r2/bench_rt.json').read().strip().splitlines()[-1])
print(d['value'], d['roundtrip'])
for l in d['legs']: print(l['leg'], l['value'], l['roundtrip'])
PY
echo "bench wall: $SECONDS s"
