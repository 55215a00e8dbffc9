#!/bin/bash
# gpurun_out/prof_<tag> (tools/profile_config.sh) -> profiles/<prefix>_<tag>_kernel_stats.csv, profiles/<prefix>_<tag>_pmc.json
# usage: tools/collect_config.sh <tag> <prefix>
set -e
IN=gpurun_out/prof_$1; PRE=profiles/$2_$1
one() { ls -t $1/*/*_$2.csv 2>/dev/null | head -1; }
cp "$(one $IN/stats kernel_stats)" ${PRE}_kernel_stats.csv
python tools/pmc_step.py "$(one $IN/fetch counter_collection)" "$(one $IN/write counter_collection)" "$(one $IN/sq counter_collection)" ${PRE}_pmc.json
ls -la ${PRE}_*
