#!/bin/bash
# round 5, session 23 (second sitting): same-box A/B of coarse's byte-interleaved tile bits (C against P), of the tiles zeroed beside
# k_flatten_light (A against C), of lane streams on CU partitions; the new GPU tests first
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r5s23
mkdir -p $O
cp .commit_stamp $O/commit.txt 2>/dev/null || true
(timeout 240 python -m pytest tests/test_gpu_parity.py -x -q -k "tiles_zeroed or config_c3 or config_c2 or clip_blend or many_bins or test_reference_test_scenes" 2>&1 | tail -8) > $O/new_tests.log; tail -3 $O/new_tests.log
timeout 420 python scripts/experiments/round5b_ab.py 3 > $O/round5b_ab.jsonl 2> $O/round5b_ab.txt; tail -40 $O/round5b_ab.txt
