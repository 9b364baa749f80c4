# the round's evidence in one GPU call: the GPU suite, smoke, profiles (kernel trace + PMC), bench lines, config 4, fp32 accuracy
python -m pytest tests -m gpu -x -q 2>&1 | tail -12 > gpurun_out/r04_gputest.log; tail -3 gpurun_out/r04_gputest.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -5
bash tools/round_profiles.sh r04 prof > gpurun_out/r04_prof.log 2>&1
bash tools/round_profiles.sh r04 bench > gpurun_out/r04_benchphase.log 2>&1
python bench.py --spawn --steps 5 --no-cpu-baseline > gpurun_out/r04_spawn_line.json 2>gpurun_out/r04_spawn_line.err
bash tools/xp/quad_gpu.sh > gpurun_out/r04_quad3.txt 2>&1
python tools/xp/f32_err.py > gpurun_out/r04_f32_err.txt 2>&1
python tools/xp/check_cost.py > gpurun_out/r04_check_cost.txt 2>&1
wc -l gpurun_out/r04_bench_lines.jsonl
