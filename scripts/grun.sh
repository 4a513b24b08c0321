#!/bin/bash
# gpurun with a commit stamp: the GPU box receives a snapshot without .git, so bench.py and the PMC scripts read
# .commit_stamp ("<short hash>" or "<short hash>-dirty") to say which sources they measured.
cd /root/repo
s=$(git rev-parse --short HEAD)
git diff --quiet HEAD -- . ':!profiles' ':!*.md' || s="$s-dirty"
echo "$s" > .commit_stamp
exec /usr/local/graft/bin/gpurun "$@"
