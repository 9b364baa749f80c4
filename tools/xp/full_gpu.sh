# full GPU check of the shipped library: the GPU suite, smoke, the default bench line, the fp32-product lines
python -m pytest tests -m gpu -x -q 2>&1 | tail -15
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -6
python bench.py > gpurun_out/r04_default_line.json 2> gpurun_out/r04_default_line.err; tail -c 600 gpurun_out/r04_default_line.json; tail -5 gpurun_out/r04_default_line.err
python bench.py --workload c3 --dtype f32 --f32-arith --no-extra > gpurun_out/r04_c3_f32_line.json 2>/dev/null
python bench.py --workload c2 --dtype f32 --f32-arith --no-extra > gpurun_out/r04_c2_f32_line.json 2>/dev/null
python bench.py --spawn --steps 5 --no-cpu-baseline > gpurun_out/r04_spawn_line.json 2>gpurun_out/r04_spawn_line.err; tail -c 900 gpurun_out/r04_spawn_line.json
