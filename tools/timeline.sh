#!/bin/bash
# usage: tools/timeline.sh name first count program args...   (runs on the GPU box): kernel timeline (start order, gaps,
# queue) of dispatches first .. first+count of `python3 program args...` under rocprofv3 --kernel-trace
# -> gpurun_out/timeline_<name>.txt   (tools/rocpd_stats.py --timeline)
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
out=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $out
name=$1; first=$2; count=$3; shift 3
timeout 600 rocprofv3 --kernel-trace -d $out/tl_$name -o t -- python3 "$@" > $out/tl_$name.log 2>&1 < /dev/null
python3 tools/rocpd_stats.py $out/tl_$name/t_results.db --timeline $first $count > $out/timeline_$name.txt 2>&1
rm -rf $out/tl_$name
