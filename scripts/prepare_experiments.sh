#!/bin/bash
# Builds what scripts/gpu_experiments.sh measures: the phase-timer build and one library per patch of scripts/experiments/
# (ab_tmp/ is git-ignored and travels with gpurun).  Run on the CPU box before the GPU call.
set -euo pipefail
cd "$(dirname "$0")/.."
scripts/build_prof.sh
scripts/variant.sh fine_merged_restore B
scripts/variant.sh fine_slot_b128 C
scripts/variant.sh fine_graded_prio D
scripts/variant.sh path_count_chunk8 E
ls -la ab_tmp
