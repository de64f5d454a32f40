#!/bin/bash
# PMC passes over the large-scene launch (tools/c5_launch.py): instruction counts, lane utilisation, waits, HBM traffic of kajo_render_*_biglist.
# usage: tools/pmc_c5.sh <tag> [strict] [nolists]      results under gpurun_out/pmc/<tag>/, summary printed
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
TAG=${1:-c5}; shift
OUT=gpurun_out/pmc/$TAG
mkdir -p $OUT
run() { name=$1; shift; rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$name -- python3 tools/c5_launch.py $EXTRA > $OUT/$name.log 2>&1; echo "pass $name done"; }
EXTRA="$*"
run a SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE
run b SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32 SQ_THREAD_CYCLES_VALU SQ_LDS_BANK_CONFLICT
run e SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_IOPS SQ_ACTIVE_INST_VALU2 SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS
run c FETCH_SIZE
run d WRITE_SIZE
python3 tools/pmc_summary.py $OUT c5 | tee $OUT/summary.txt
