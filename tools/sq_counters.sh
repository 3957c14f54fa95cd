#!/bin/bash
# SQ counter passes (rocprofv3 --pmc, kernel trace only, one counter set per run: tools/pmc_parse.sh) over the kernels the review
# asked about, summarised per kernel and launch into gpurun_out/sq_counters.txt (tracked copy: profiles/<round>_sq_counters.txt).
# Counter collection serialises kernels, so "beside" regimes cannot be measured this way: each kernel is measured alone.
#   gpurun -- 'bash tools/sq_counters.sh'        (round 2's kernels)
#   gpurun -- 'bash tools/sq_counters.sh vtab'   (round 3: the register-table LZ4 parsers)
O=gpurun_out/sq_counters.txt
: > $O
run() { # tag, kernel filter, blocks, perf_probe args...
    local tag=$1 filt=$2 nb=$3; shift 3
    bash tools/pmc_parse.sh $tag --nb $nb "$@" > /dev/null 2>&1 || { echo "$tag: collection failed" >> $O; return; }
    echo "== $tag: perf_probe.py --nb $nb $* (kernel filter: $filt; averages per launch, and per block where it makes sense)" >> $O
    python3 tools/pmc_summary.py $tag $filt $nb >> $O
    echo >> $O
    rm -rf gpurun_out/pmc_${tag}_*
}
if [ "$1" = "vtab" ]; then # round 3: the register-table parsers, each taking the whole queue (CW_LZ4_VTAB=1), 16 wavefronts per CU
    CW_LZ4_VTAB=1 CW_VTAB_GEN=3 CW_LZ4_LANES=0 run vtab_gen3 lz4_vtab3_kernel  8192 --alg none --comp lz4 --bs 65536 --data text
    CW_LZ4_VTAB=1 CW_VTAB_GEN=2 CW_LZ4_LANES=0 run vtab_gen2_4k lz4_vtab2_kernel 65536 --alg none --comp lz4 --bs 4096 --data text
    cat $O
    exit 0
fi
run parse_wave   lz4_parse_kernel      8192   --alg none --comp lz4 --bs 65536 --data text
run lanes_k2     lz4_lanes_ring_kernel 16384  --alg none --comp lz4 --bs 65536 --data text
run lanes_k1     lz4_lanes_ring_kernel 65536  --alg none --comp lz4 --bs 65536 --data text
run lzf_lanes    lzf_lanes_kernel      65536  --alg none --comp lzf --bs 65536 --data text
run parse_4k     lz4_parse_kernel      32768  --alg none --comp lz4 --bs 4096  --data text
run skein_alone  skein_slice_kernel    262144 --alg skein512 --bs 65536 --data random
run scan_alone   lz4_scan_span_kernel  262144 --alg none --comp lz4 --bs 65536 --data random
cat $O
