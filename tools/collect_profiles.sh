#!/bin/bash
# Runs on the GPU box (gpurun): the default bench line, rocprofv3 kernel statistics of the headline command and of each extra
# leg as its own command, and FETCH_SIZE / WRITE_SIZE passes (separate --pmc runs, kernel trace only -- MI355X_MICROARCH.md).
# Raw output goes to gpurun_out/<tag>/; tools/profile_summary.py turns it into the tracked files under profiles/.
#   gpurun -- 'bash tools/collect_profiles.sh r03'            (everything; ~20 minutes -- or in two calls:)
#   gpurun -- 'bash tools/collect_profiles.sh r03 stats'      (default bench + kernel statistics)
#   gpurun -- 'bash tools/collect_profiles.sh r03 pmc'        (FETCH_SIZE / WRITE_SIZE passes)
set -e
TAG=${1:-r03}
PART=${2:-all}
O=gpurun_out/$TAG; mkdir -p $O
# leg name, hash, codec, block bytes, input, blocks (must agree with bench.py's plan and tools/profile_summary.py's LEGS)
LEGS=("mixed skein512 lz4 65536 mixed 65536" "corpus_skein512_lz4 skein512 lz4 65536 corpus 65536" "corpus_skein512_lz4_16g skein512 lz4 65536 corpus 262144" "corpus_skein512_lz4_3233 skein512 lz4 65536 corpus 3233"
      "corpus_skein512_lz4_51728 skein512 lz4 4096 corpus 51728" "corpus_skein256_lz4_4k skein lz4 4096 corpus 1048576"
      "corpus_sha256_lzf_4k sha256mb lzf 4096 corpus 1048576" "corpus_sha256_lzf_64k sha256mb lzf 65536 corpus 65536" "corpus_sha256_lzf_64k_16g sha256mb lzf 65536 corpus 262144")
if [ "$PART" != pmc ]; then
python bench.py > $O/bench_default.json 2> $O/bench_default.err || { tail -20 $O/bench_default.err; exit 1; }
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d $O/prof_headline -o p -f csv -- python3 bench.py --no-legs --no-cpu-baseline > $O/prof_headline.json 2> $O/prof_headline.err
for leg in "${LEGS[@]}"; do
  set -- $leg
  rocprofv3 --kernel-trace --stats -d $O/prof_$1 -o p -f csv -- python3 bench.py --no-legs --no-cpu-baseline --hash $2 --comp $3 --block-bytes $4 --data $5 --blocks-per-gpu $6 --steps 3 --warmup 1 > $O/prof_$1.json 2> $O/prof_$1.err
done
fi
if [ "$PART" = stats ]; then echo collected: $(ls $O | wc -l) entries in $O; exit 0; fi
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace -d $O/pmc_headline_$c -o p -f csv -- python3 bench.py --no-legs --no-cpu-baseline --no-roundtrip --steps 1 --warmup 0 > $O/pmc_headline_$c.log 2>&1
  for leg in "${LEGS[@]}"; do
    set -- $leg
    rocprofv3 --pmc $c --kernel-trace -d $O/pmc_$1_$c -o p -f csv -- python3 bench.py --no-legs --no-cpu-baseline --hash $2 --comp $3 --block-bytes $4 --data $5 --blocks-per-gpu $6 --steps 1 --warmup 0 > $O/pmc_$1_$c.log 2>&1
  done
done
echo collected: $(ls $O | wc -l) entries in $O
