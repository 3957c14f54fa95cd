set -e
O=gpurun_out/r2; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/pytest_full.log 2>&1 || { tail -40 $O/pytest_full.log; exit 1; }
tail -2 $O/pytest_full.log
python bench.py > $O/bench_default.json 2> $O/bench_default.err || { tail -20 $O/bench_default.err; exit 1; }
python - <<'PY'
import json
d=json.load(open("gpurun_out/r2/bench_default.json"))
print("HEADLINE", d["value"], d["ms_per_step"], d["roofline"]["frac"], d.get("cpu_baseline",{}).get("value"))
for l in d.get("legs",[]): print(l["leg"], l["value"], "GB/s ratio", l["compression_ratio"], l.get("corpus_ratios"), "| dom", l["roofline"]["kernel"], l["roofline"]["frac"], "| cpu", l.get("cpu_baseline",{}).get("value"))
print("HOST", d.get("host_path"))
PY
python - <<'PY'
root = "tests/golden/corpus/canterbury/"
t = b"".join(open(root + f, "rb").read() for f in ("lcet10.txt", "kennedy.xls", "ptt5"))
with open("/tmp/corpus4g.bin", "wb") as f:
    n = 0
    while n < 4 << 30:
        f.write(t); n += len(t)
import numpy as np
rng = np.random.default_rng(1)
blk = [rng.integers(0, 256, 64 << 20, dtype=np.uint8).tobytes() for _ in range(4)]
with open("/tmp/random16g.bin", "wb") as f:
    for i in range(256): f.write(blk[i % 4])
PY
L=$O/hostpath.log; rm -f $L
H=./compute_war_amd/host/hashandcompress
for c in 1 2 4; do echo "== random16g -c $c skein512+lz4 64K" >> $L; $H -v -g true -c $c -r 1 --block-size=65536 -H skein512 -C lz4 /tmp/random16g.bin >> $L 2>&1; done
for c in 2 4; do echo "== corpus4g -c $c skein512+lz4 64K" >> $L; $H -v -g true -c $c -r 1 --block-size=65536 -H skein512 -C lz4 /tmp/corpus4g.bin >> $L 2>&1; done
echo "== corpus4g -c 4 skein+lz4 4K" >> $L; $H -v -g true -c 4 -r 8 -H skein -C lz4 /tmp/corpus4g.bin >> $L 2>&1
echo "== corpus4g -c 4 sha256mb+lzf 4K" >> $L; $H -v -g true -c 4 -r 8 -H sha256mb -C lzf /tmp/corpus4g.bin >> $L 2>&1
echo "== random16g -c 4 sha256mb+lzf 4K" >> $L; $H -v -g true -c 4 -r 8 -H sha256mb -C lzf /tmp/random16g.bin >> $L 2>&1
cat $L
rm -f /tmp/corpus4g.bin /tmp/random16g.bin
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d $O/prof_headline -o p -f csv -- python3 bench.py --no-legs --no-cpu-baseline > $O/prof_headline.json 2> $O/prof_headline.err
for leg in "mixed skein512 lz4 65536 mixed" "corpus_skein512_lz4 skein512 lz4 65536 corpus" "corpus_sha256_lzf_4k sha256mb lzf 4096 corpus" "corpus_sha256_lzf_64k sha256mb lzf 65536 corpus"; do
  set -- $leg
  rocprofv3 --kernel-trace --stats -d $O/prof_$1 -o p -f csv -- python3 bench.py --no-legs --no-cpu-baseline --hash $2 --comp $3 --block-bytes $4 --data $5 --blocks-per-gpu $(( (4<<30) / $4 )) --steps 3 --warmup 1 > $O/prof_$1.json 2> $O/prof_$1.err
done
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace -d $O/pmc_headline_$c -o p -f csv -- python3 bench.py --no-legs --no-cpu-baseline --steps 1 --warmup 0 > $O/pmc_headline_$c.log 2>&1
  rocprofv3 --pmc $c --kernel-trace -d $O/pmc_corpus_skein512_lz4_$c -o p -f csv -- python3 bench.py --no-legs --no-cpu-baseline --data corpus --blocks-per-gpu 65536 --steps 1 --warmup 0 > $O/pmc_corpus_skein512_lz4_$c.log 2>&1
  rocprofv3 --pmc $c --kernel-trace -d $O/pmc_corpus_sha256_lzf_4k_$c -o p -f csv -- python3 bench.py --no-legs --no-cpu-baseline --hash sha256mb --comp lzf --block-bytes 4096 --data corpus --blocks-per-gpu 1048576 --steps 1 --warmup 0 > $O/pmc_corpus_sha256_lzf_4k_$c.log 2>&1
done
ls $O | head -50
