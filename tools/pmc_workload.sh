#!/bin/bash
# `rocprofv3 --kernel-trace --stats` and the six PMC passes (counters only ever with --kernel-trace, each group a run of its own) over
# tools/launch_workload.py: instruction counts, lane utilisation, waits, HBM traffic of the kernel a workload runs.
# usage: tools/pmc_workload.sh <tag> c2|c4|c5 [fast|strict|exact] [nolists]     results under gpurun_out/pmc/<tag>/, summary printed
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
TAG=${1:?tag}; shift
OUT=gpurun_out/pmc/$TAG
mkdir -p $OUT
EXTRA="$*"
run() { name=$1; shift; rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$name -- python3 tools/launch_workload.py $EXTRA > $OUT/$name.log 2>&1; echo "pass $name done"; }
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 tools/launch_workload.py $EXTRA launches=8 > $OUT/stats.log 2>&1; echo "stats done"
run a SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE
run b SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32 SQ_THREAD_CYCLES_VALU SQ_LDS_BANK_CONFLICT
run e SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_IOPS SQ_ACTIVE_INST_VALU2 SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS
run f SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_MISC SQ_INST_LEVEL_LDS SQ_INSTS_VALU_FLOPS_FP32 SQ_INSTS_VALU_FLOPS_FP32_TRANS
run c FETCH_SIZE
run d WRITE_SIZE
python3 tools/pmc_summary.py $OUT $(echo $EXTRA | cut -d' ' -f1) | tee $OUT/summary.txt
python3 tools/steady_stats.py $OUT/stats | tee -a $OUT/summary.txt
