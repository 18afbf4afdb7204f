#!/bin/bash
# run 36: what a 20-step timed region loses against steady state
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
for i in 1 2; do timeout -k 10 300 python3 tools/short_run.py 2>/dev/null | tail -1; done
