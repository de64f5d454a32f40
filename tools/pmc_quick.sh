#!/bin/bash
# Two quick PMC passes (instruction counts and lane utilisation) over the bench's launch; knobs come from the environment.
# usage: tools/pmc_quick.sh <tag> [bench.py arguments]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
TAG=${1:-q}; shift
OUT=gpurun_out/pmc/$TAG
rm -rf $OUT; mkdir -p $OUT
run() { name=$1; shift; rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$name -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline $EXTRA > $OUT/$name.log 2>&1; }
EXTRA="$*"
run a SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY
run b SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_FMA_F32 SQ_THREAD_CYCLES_VALU SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS
python3 tools/pmc_summary.py $OUT > $OUT/summary.txt; grep -v "^  FETCH\|^  WRITE" $OUT/summary.txt
