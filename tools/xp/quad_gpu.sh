# the quad variant of the lane kernel on the GPU: the lane / small-shape GPU tests, then BASELINE config 4 (tests/cpp/sqp_batch_test.cpp)
python -m pytest tests/test_gpu_parity.py -x -q -k "lane or small_shapes or setup_solve_reuse or fused_call or api_sequence" 2>&1 | tail -8
python -c "
import sys; sys.path.insert(0,'tests')
import test_cpp_sqp as t, subprocess
p = subprocess.run([t.build()], capture_output=True, text=True, timeout=900)
open('gpurun_out/r04_sqp_parity_log.txt','w').write(p.stdout + p.stderr)
print('rc', p.returncode)
print('\n'.join(l for l in p.stdout.splitlines() if 'wall' in l or 'passed' in l or 'strict' in l)[:6000])
"
