#!/bin/bash
# round 5, session 19: k_front (the stages of a small scene as shared launches) -- its GPU tests (grid barrier
# across XCDs, four frames in flight), then fused against unfused on the small workloads, same box, interleaved
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r5s19
mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "front_fusion" 2>&1 | tail -5 | tee $O/tests.txt
timeout 600 python scripts/small_scene_latency.py 3 2>&1 | grep -v amdgpu.ids | tee $O/small_scene_latency.jsonl
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "smoke or tiger or clip or blend or random or brush or gradient or catalogue or fuzz" 2>&1 | tail -3 | tee $O/tests_more.txt
