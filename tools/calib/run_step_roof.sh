#!/bin/bash
# GPU box: builds and runs tools/calib/step_roof.hip -> gpurun_out/step_roof.jsonl (copied to profiles/r3_step_roof.jsonl)
set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -o gpurun_out/step_roof tools/calib/step_roof.hip
timeout -k 10 120 gpurun_out/step_roof | tee gpurun_out/step_roof.jsonl
