#!/bin/bash
# gpurun_out/<tag> (written by tools/profile_round.sh on the GPU box) -> the tracked summaries under profiles/:
#   <prefix>_bench_kernel_stats.csv, <prefix>_bf16_bench_kernel_stats.csv          rocprofv3 --kernel-trace --stats of bench.py
#   <prefix>_kernel_bench_b{128,16384}_{f32,bf16}.txt / _kernel_stats.csv           the per-kernel micro-benchmark
#   <prefix>_traffic_b*.json (tools/pmc_summary.py), <prefix>_pmc_mfma_b*.json (tools/pmc_mfma.py)
# usage: tools/collect_profiles.sh gpurun_out/r03f r03_end
set -e
IN=$1; PRE=profiles/$2
one() { ls -t $1/*/*_$2.csv 2>/dev/null | head -1; }
cp "$(one $IN/bench kernel_stats)" ${PRE}_bench_kernel_stats.csv
cp "$(one $IN/bench_bf16 kernel_stats)" ${PRE}_bf16_bench_kernel_stats.csv
for CFG in "128 f32 20" "128 bf16 20" "16384 f32 5"; do
  set -- $CFG; B=$1; DT=$2; IT=$3; T=${B}_${DT}
  SUF=""; [ $DT = bf16 ] && SUF="_bf16"
  cp $IN/kb_${T}.txt ${PRE}_kernel_bench_b${B}_${DT}.txt
  cp "$(one $IN/kb_${T}_trace kernel_stats)" ${PRE}_kernel_bench_b${B}_${DT}_kernel_stats.csv
  python tools/pmc_summary.py "$(one $IN/kb_${T}_fetch counter_collection)" "$(one $IN/kb_${T}_write counter_collection)" $IT ${PRE}_traffic_b${B}${SUF}.json > /dev/null
  python tools/pmc_mfma.py "$(one $IN/kb_${T}_mfma counter_collection)" ${PRE}_pmc_mfma_b${B}${SUF}.json > /dev/null
done
ls -la ${PRE}_*
