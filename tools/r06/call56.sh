#!/bin/bash
# round 6, call 56: counters of the aggregated configurations re-collected (their exponent launch is a new kernel): PMC passes only, cfg3 / cfg4
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
PMC_CFGS="cfg3 cfg4" ONLY_PMC=1 bash tools/collect_profiles.sh r06c > gpurun_out/collect_r06c.log 2>&1; tail -4 gpurun_out/collect_r06c.log | cut -c1-600
ls gpurun_out/prof_r06c
