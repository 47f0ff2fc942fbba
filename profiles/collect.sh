#!/bin/bash
# Collects the round's evidence into gpurun_out/final/ (copied to profiles/ afterwards)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/final; mkdir -p $O
cd $R
python bench.py > $O/r02_bench_default.json 2> $O/bench_default.err
python bench.py --workload hd64-bf16 > $O/r02_bench_hd64_bf16.json 2> $O/bench_hd.err
python bench.py --workload qvga32-magicpoint > $O/r02_bench_qvga32_magicpoint.json 2> $O/bench_qvga.err
python bench.py --arch vgg > $O/r02_bench_vgg_f32.json 2> $O/bench_vgg.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o p -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-host-fed --no-alt-pass --no-serial-pass --no-steady-state --no-latency > $O/r02_bench_under_rocprof.json 2> $O/kt.err
cp $(find $O/kt -name "*kernel_stats.csv" | head -1) $O/r02_bench_kernel_stats.csv
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_hd -o p -- python3 $R/bench.py --workload hd64-bf16 --steps 10 --warmup 3 --no-cpu-baseline --no-serial-pass --no-steady-state --no-latency > $O/r02_bench_hd64_bf16_under_rocprof.json 2> $O/kt_hd.err
cp $(find $O/kt_hd -name "*kernel_stats.csv" | head -1) $O/r02_bench_hd64_bf16_kernel_stats.csv
cd $R
bash profiles/pmc_passes.sh r02final > /dev/null 2>&1; cp gpurun_out/pmc_r02final.csv $O/r02_pmc_summary_serial.csv
TAG=r02final_hd; export FPC_STREAMS=1 FPC_NMS_ASIDE=0
cd /tmp
for pass in "mfma:SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "fetch:FETCH_SIZE" "write:WRITE_SIZE"; do
  name=${pass%%:*}; ctr=${pass#*:}
  rocprofv3 --pmc $ctr --output-format csv -d $R/gpurun_out/pmc_$TAG/$name -o p -- python3 $R/bench.py --workload hd64-bf16 --steps 3 --warmup 1 --only-timed --no-timing-events > /dev/null 2>&1
done
python3 $R/profiles/summarize_pmc.py --frames=64 $(find $R/gpurun_out/pmc_$TAG -name "*counter_collection.csv") > $O/r02_pmc_summary_serial_bf16.csv
rm -rf $O/kt $O/kt_hd
ls -la $O
