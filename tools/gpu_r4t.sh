#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
O=gpurun_out/r4t
rm -rf $O; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 600 python bench.py > $O/full.json 2> $O/full.err; echo "full: rc $?"
grep "\[bench\]\|abort\|Abort\|error" $O/full.err | cut -c1-300
AMD_SERIALIZE_KERNEL=3 timeout 600 python bench.py --no-cpu-baseline > $O/full2.json 2> $O/full2.err; echo "full serialized: rc $?"
grep "\[bench\]\|abort\|Abort\|error" $O/full2.err | cut -c1-300
