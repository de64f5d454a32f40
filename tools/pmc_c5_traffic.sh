#!/bin/bash
# HBM traffic only (FETCH_SIZE, WRITE_SIZE: separate passes) of the large-scene launch. usage: tools/pmc_c5_traffic.sh <tag> [strict|exact] [nolists]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
TAG=${1:-c5t}; shift
OUT=gpurun_out/pmc/$TAG
mkdir -p $OUT
EXTRA="$*"
run() { name=$1; shift; rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$name -- python3 tools/c5_launch.py $EXTRA > $OUT/$name.log 2>&1; echo "pass $name done"; }
run c FETCH_SIZE
run d WRITE_SIZE
python3 tools/pmc_summary.py $OUT c5 | tee $OUT/summary.txt
