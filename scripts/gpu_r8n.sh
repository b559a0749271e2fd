#!/bin/bash
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd $R
bash scripts/gpu_netprof.sh r8n 2>&1 | tail -45 | cut -c1-175
