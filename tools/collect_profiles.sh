#!/bin/bash
# copies what tools/profile_r3.sh left under OUT into profiles/r03_* (the files profiles/README.md lists)
#   bash tools/collect_profiles.sh gpurun_out/r3q
set -e -o pipefail
R=${1:?run directory}
cp $R/bench.json profiles/r03_bench_n1.json
cp $R/bench_prof.json profiles/r03_bench_n1_under_rocprof.json
cp "$(ls -t $R/bench_trace/vm/*_kernel_stats.csv | head -1)" profiles/r03_bench_n1_kernel_stats.csv
cp $R/bench_trace.json profiles/r03_bench_n1_trace.json
cp $R/bench_hbm_traffic.json profiles/r03_bench_n1_hbm_traffic.json
cp $R/bench_sq_insts.json profiles/r03_bench_n1_sq_insts.json
cp $R/bench_kernels.json profiles/r03_bench_kernels.json
cp "$(ls -t $R/kernels_trace/vm/*_kernel_stats.csv | head -1)" profiles/r03_bench_kernels_kernel_stats.csv
cp $R/kernels_trace.json profiles/r03_bench_kernels_trace.json
cp $R/kernels_hbm_traffic.json profiles/r03_bench_kernels_hbm_traffic.json
