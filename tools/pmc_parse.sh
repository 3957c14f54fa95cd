#!/bin/bash
# usage: tools/pmc_parse.sh <tag> <perf_probe args...>   -- SQ counter passes over one perf_probe run (on the GPU box)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=$1; shift
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
           "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INSTS_BRANCH" \
           "SQ_LDS_UNALIGNED_STALL SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM" \
           "SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_IFETCH SQ_IFETCH_LEVEL SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"; do
    i=$((i+1))
    timeout -k 10 240 rocprofv3 --pmc $set --kernel-trace -d gpurun_out/pmc_${tag}_$i -o p -f csv -- python3 tools/perf_probe.py "$@" > gpurun_out/pmc_${tag}_$i.log 2>&1 || exit 1
done
grep "lib=" gpurun_out/pmc_${tag}_1.log
