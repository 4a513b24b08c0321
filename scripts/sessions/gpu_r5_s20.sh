#!/bin/bash
# round 5, session 20: where a small scene's frame goes, launch by launch (rocprofv3 kernel trace, one frame at a time), fused and not
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r5s20
mkdir -p $O
for w in circle tiger image_sampling; do
  for m in fused unfused; do
    rm -rf /tmp/tl; mkdir -p /tmp/tl
    timeout 200 rocprofv3 --kernel-trace --output-format csv -d /tmp/tl -o tl -- python scripts/small_scene_trace.py $w $m 200 > /dev/null 2>&1
    f=$(find /tmp/tl -name "*kernel_trace.csv" | head -1)
    echo "== $w $m"; python scripts/frame_timeline.py "$f"
  done
done 2>&1 | tee $O/frame_timeline.txt
