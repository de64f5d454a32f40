#!/bin/bash
# PMC passes over the bench (separate runs; --pmc only ever with --kernel-trace): results under gpurun_out/pmc/<tag>/,
# summary printed and written to gpurun_out/pmc/<tag>/summary.txt + counters.json (tools/pmc_summary.py).
# usage: tools/pmc.sh <tag> [extra bench.py arguments, e.g. --strict]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
TAG=${1:-v}; shift
OUT=gpurun_out/pmc/$TAG
mkdir -p $OUT
run() { name=$1; shift; rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$name -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --sustain-seconds 0 $EXTRA > $OUT/$name.log 2>&1; echo "pass $name done"; }
EXTRA="$*"
run a SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE
run b SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32 SQ_THREAD_CYCLES_VALU SQ_LDS_BANK_CONFLICT
run e SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_IOPS SQ_ACTIVE_INST_VALU2 SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS
run f SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_MISC SQ_INST_LEVEL_LDS SQ_INSTS_VALU_FLOPS_FP32 SQ_INSTS_VALU_FLOPS_FP32_TRANS
run c FETCH_SIZE
run d WRITE_SIZE
python3 tools/pmc_summary.py $OUT | tee $OUT/summary.txt
