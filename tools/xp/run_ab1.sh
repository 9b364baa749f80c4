export PYTHONPATH=$PWD
bash tools/xp/ab_csb.sh xp_libs/base.so xp_libs/new.so > gpurun_out/ab1.txt 2>&1
for L in base new; do SQPH_LIB=xp_libs/$L.so python tools/bench_csr.py --steps 2 --check 8 2>/dev/null | tail -1 >> gpurun_out/ab1.txt; done
for L in base_pt new_pt; do echo "== $L" >> gpurun_out/ab1_pt.txt; SQPH_LIB=xp_libs/$L.so python tools/phase_timing_csb.py 200 400 2048 >> gpurun_out/ab1_pt.txt 2>&1; SQPH_PT_WAVE=7 SQPH_LIB=xp_libs/$L.so python tools/phase_timing_csb.py 200 400 2048 >> gpurun_out/ab1_pt.txt 2>&1; done
