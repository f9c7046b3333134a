#!/bin/bash
# Power, clock and time of the headline walk under the launch variants, one process each, on one box:
# default (two, then one), one evaluation per lane throughout, two throughout.   tools/ab_power.sh [seconds]
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"
export PYTHONPATH=. FSS_AMD_TUNE=1
S="${1:-3}"
for rep in 1 2; do
  echo "== default";            python3 tools/power_probe.py "$S" "one key x 2^24" 2>/dev/null | grep "one key"
  echo "== FSS_AMD_ILP=1";      FSS_AMD_ILP=1 python3 tools/power_probe.py "$S" "one key x 2^24" 2>/dev/null | grep "one key"
  echo "== ILP=2, no mix";      FSS_AMD_ILP=2 FSS_AMD_MIX_MID_PCT=0 FSS_AMD_MIX_TAIL_UNITS=0 python3 tools/power_probe.py "$S" "one key x 2^24" 2>/dev/null | grep "one key"
done
