# baseline of a round: set-up / iteration phase ticks of the C3 kernel and the bench lines of the three modes (slim builds, tools/slim_build.sh)
L=$PWD/sqp_solver_amd/lib
SQPH_LIB=$L/libsqp_hip_slimsetup.so python tools/setup_timing.py 50 100 8192
SQPH_LIB=$L/libsqp_hip_slimphase.so python tools/phase_timing.py 50 100 8192
for mode in fixed default sqp; do for i in 1 2; do
echo -n "$mode: "; SQPH_LIB=$L/libsqp_hip_slimA.so python bench.py --no-cpu-baseline --steps 40 --mode $mode 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read()); print(r['ms_per_step'], r['roofline']['kernel_ms_avg'], r['config']['admm_iters_per_qp'], r['config']['kernel'])"
done; done
