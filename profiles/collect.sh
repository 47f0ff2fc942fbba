#!/bin/bash
# Collects a round's evidence into gpurun_out/final/ (copied to profiles/ afterwards):
#   gpurun --timeout 1200 -- 'bash profiles/collect.sh r04'
# Order matters: the PMC passes come FIRST, so that the bench lines written afterwards look their `roofline.traffic` up
# in the summaries of this very build (round 2 ran the bench first and its JSON quoted an older summary).
set -euo pipefail
: "${GRAFT_REPO_ROOT:?run through gpurun (GRAFT_REPO_ROOT is the repo copy on the GPU box)}"
TAG=${1:-r04}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/final; mkdir -p "$O"
cd "$R"
bash profiles/pmc_passes.sh ${TAG}final > "$O/pmc.log" 2>&1
cp gpurun_out/pmc_${TAG}final.csv "$O/${TAG}_pmc_summary_serial.csv"
cp "$O/${TAG}_pmc_summary_serial.csv" profiles/      # (on the box only: bench.py below reads it)
bash profiles/pmc_passes.sh ${TAG}final_hd --frames=64 --workload hd64-bf16 > "$O/pmc_hd.log" 2>&1
cp gpurun_out/pmc_${TAG}final_hd.csv "$O/${TAG}_pmc_summary_serial_bf16.csv"
cp "$O/${TAG}_pmc_summary_serial_bf16.csv" profiles/
python bench.py > "$O/${TAG}_bench_default.json" 2> "$O/bench_default.err"
python bench.py --workload hd64-bf16 > "$O/${TAG}_bench_hd64_bf16.json" 2> "$O/bench_hd.err"
python bench.py --workload qvga32-magicpoint > "$O/${TAG}_bench_qvga32_magicpoint.json" 2> "$O/bench_qvga.err"
python bench.py --arch vgg > "$O/${TAG}_bench_vgg_f32.json" 2> "$O/bench_vgg.err"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/kt" -o p -- python3 "$R/bench.py" --steps 20 --warmup 5 --no-cpu-baseline --no-host-fed --no-alt-pass --no-serial-pass --no-steady-state --no-latency --no-other-workloads > "$O/${TAG}_bench_under_rocprof.json" 2> "$O/kt.err"
cp "$(find "$O/kt" -name "*kernel_stats.csv" | head -1)" "$O/${TAG}_bench_kernel_stats.csv"
# the kernels ALONE on the GPU (one context, one stream, NMS in line): the pass whose durations `roofline.frac` is priced on.
# rocprofv3's AverageNs per kernel (CSV) and the HIP-event durations of the line written by the same run (JSON) must agree.
export FPC_STREAMS=1 FPC_NMS_ASIDE=0
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/kt_serial" -o p -- python3 "$R/bench.py" --steps 20 --warmup 5 --only-timed --contexts 1 > "$O/${TAG}_bench_serial_under_rocprof.json" 2> "$O/kt_serial.err"
cp "$(find "$O/kt_serial" -name "*kernel_stats.csv" | head -1)" "$O/${TAG}_bench_serial_kernel_stats.csv"
unset FPC_STREAMS FPC_NMS_ASIDE
rm -rf "$O/kt_serial"
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/kt_hd" -o p -- python3 "$R/bench.py" --workload hd64-bf16 --steps 10 --warmup 3 --no-cpu-baseline --no-serial-pass --no-steady-state --no-latency > "$O/${TAG}_bench_hd64_bf16_under_rocprof.json" 2> "$O/kt_hd.err"
cp "$(find "$O/kt_hd" -name "*kernel_stats.csv" | head -1)" "$O/${TAG}_bench_hd64_bf16_kernel_stats.csv"
rm -rf "$O/kt" "$O/kt_hd"
ls -la "$O"
