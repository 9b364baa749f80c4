# A/B of two slim builds (tools/slim_build.sh): set-up / iteration phase ticks of variant B, parity of B at the C3 shapes, then the
# bench lines of A and B in the three modes, alternating.   usage: tools/xp/ab_round.sh [A] [B]
A=${1:-A}; B=${2:-B}
L=$PWD/sqp_solver_amd/lib
[ -f $L/libsqp_hip_slimsetup$B.so ] && SQPH_LIB=$L/libsqp_hip_slimsetup$B.so python tools/setup_timing.py 50 100 8192
[ -f $L/libsqp_hip_slimphase$B.so ] && SQPH_LIB=$L/libsqp_hip_slimphase$B.so python tools/phase_timing.py 50 100 8192
SQPH_LIB=$L/libsqp_hip_slim$B.so timeout 300 python tools/xp/parity_c3.py 2>&1 | tail -25
for i in 1 2; do for mode in fixed default sqp; do for v in $A $B; do
echo -n "$mode $v: "; SQPH_LIB=$L/libsqp_hip_slim$v.so python bench.py --no-cpu-baseline --steps 40 --mode $mode 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read()); print(r['ms_per_step'], r['roofline']['kernel_ms_avg'], r['config']['admm_iters_per_qp'], r['config']['kernel'])"
done; done; done
