#!/bin/bash
# usage: tools/r6_hiptrace.sh program args...  (GPU box): HIP API + kernel trace of the program -> gpurun_out/hiptrace/ (rocpd db)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
out=gpurun_out/hiptrace
rm -rf $out; mkdir -p $out
timeout 300 rocprofv3 --hip-runtime-trace --kernel-trace -d $out -o t -- python3 "$@" > $out/run.log 2>&1 < /dev/null
ls -la $out
python3 - <<'PY'
import sqlite3, glob
db = glob.glob('gpurun_out/hiptrace/*.db')[0]
con = sqlite3.connect(db)
names = [r[0] for r in con.execute("select name from sqlite_master where type in ('table','view') order by name")]
print(names)
PY
