# full GPU check of the shipped library: the GPU suite, smoke, the default bench line
python -m pytest tests -m gpu -x -q 2>&1 | tail -15
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -6
python bench.py > gpurun_out/r04_default_line.json 2> gpurun_out/r04_default_line.err; tail -c 3000 gpurun_out/r04_default_line.json; tail -5 gpurun_out/r04_default_line.err
