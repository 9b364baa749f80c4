#!/bin/bash
# Everything profiles/ holds for one round, in two GPU calls:
#   gpurun --timeout 1500 -- 'bash tools/round_profiles.sh r02 prof'    rocprofv3 kernel trace + PMC passes per workload (tools/profile_gpu.sh)
#   python tools/collect_round_profiles.py r02 prof                      (here: summaries -> profiles/, PMC traffic under the source hash)
#   gpurun --timeout 1500 -- 'bash tools/round_profiles.sh r02 bench'   the bench lines (with CPU baseline and the recorded traffic)
#   python tools/collect_round_profiles.py r02 bench
TAG=${1:-r02}
PHASE=${2:-prof}
run() {  # name, bench args...
    local name=$1; shift
    if [ $PHASE = prof ]; then
        bash tools/profile_gpu.sh ${TAG}_$name "$@" > gpurun_out/prof_${TAG}_$name.out 2>&1
    else
        timeout 900 python bench.py "$@" >> gpurun_out/${TAG}_bench_lines.new 2>gpurun_out/bench_$name.err
    fi
}
rm -f gpurun_out/${TAG}_bench_lines.new
run c3_full --workload c3                                   # the headline: configs[2], 65,536 QPs in one launch
run c3 --workload c3 --global-batch 0                       # the 8,192-QP shard one of eight GPUs solves
run c3_default --workload c3 --global-batch 0 --mode default
run c3_sqp --workload c3 --global-batch 0 --mode sqp
run c2 --workload c2
run c5 --workload c5 --steps 5
run c5_sp --workload c5 --steps 5 --p-density 0.03          # the same shape with P in compressed columns (read in place)
run c5_default --workload c5 --steps 3 --mode default        # config 5 under the reference defaults / the SQP driver's settings
run c5_sqp --workload c5 --steps 5 --mode sqp
run c5_sp_default --workload c5 --steps 2 --p-density 0.03 --mode default
run c5_sp_sqp --workload c5 --steps 5 --p-density 0.03 --mode sqp
run c3_fixed98 --workload c3 --global-batch 0 --iters 98      # (what the SQP-settings call of the C3 shard would stream without its checks: 98 iterations, no check)
run lane --n 2 --m 3 --batch-per-gpu 65536
[ $PHASE = prof ] && exit 0
timeout 600 python bench.py --global-batch 65536 --steps 5 --mode default >> gpurun_out/${TAG}_bench_lines.new 2>/dev/null
timeout 600 python bench.py --n 2 --m 3 --batch-per-gpu 65536 --dtype f32 --f32-arith >> gpurun_out/${TAG}_bench_lines.new 2>/dev/null
timeout 600 python bench.py --n 4 --m 6 --batch-per-gpu 65536 >> gpurun_out/${TAG}_bench_lines.new 2>/dev/null
timeout 600 python bench.py --n 4 --m 6 --batch-per-gpu 65536 --dtype f32 --f32-arith >> gpurun_out/${TAG}_bench_lines.new 2>/dev/null
timeout 600 python bench.py --n 200 --m 400 --batch-per-gpu 512 --steps 5 >> gpurun_out/${TAG}_bench_lines.new 2>/dev/null   # dense beyond the tiled shapes (csr_dense.hip)
mv gpurun_out/${TAG}_bench_lines.new gpurun_out/${TAG}_bench_lines.jsonl
