#!/bin/bash
# rocprofv3 kernel trace + stats of one command; prints the top rows of the kernel summary.
#   tools/prof_top.sh <tag> <rows> -- <command...>
TAG=$1; ROWS=$2; shift 3
cd /tmp && export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_$TAG
rm -rf $OUT; mkdir -p $OUT
(cd $REPO && rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o $TAG -- "$@" > $OUT/cmd.log 2>&1)
f=$(find $OUT -name "*kernel_stats.csv" | head -1)
python3 - "$f" "$ROWS" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[: int(sys.argv[2])]:
    print(f"{r['Name'][:96]:96s} {int(r['Calls']):5d} {float(r['AverageNs']) / 1e3:9.1f} us {float(r['Percentage']):6.2f} %")
PY
