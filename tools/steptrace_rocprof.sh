#!/bin/bash
# per-step GPU timeline of the Welch metric step (kernels + gaps): rocprofv3 kernel trace of tools/kbench.py
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/prof_step
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_step -- python3 tools/kbench.py --reps 40 > gpurun_out/prof_step.log 2>&1
python3 tools/trace_gaps.py gpurun_out/prof_step "k_welch_carry<4096, true, 8, true>" 15
