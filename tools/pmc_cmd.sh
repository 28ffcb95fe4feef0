#!/bin/bash
# GPU box: one rocprofv3 --pmc pass (kernel trace only) of a command; usage: tools/pmc_cmd.sh <tag> "<counters>" <command...>
tag=$1; counters=$2; shift 2
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc $counters --output-format csv -d /tmp/pmc_$tag -o run -- "$@" > /tmp/pmc_$tag.log 2>&1
mkdir -p $R/gpurun_out/pmc_$tag && cp /tmp/pmc_$tag/run_counter_collection.csv $R/gpurun_out/pmc_$tag/ && tail -2 /tmp/pmc_$tag.log
