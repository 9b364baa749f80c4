L=$PWD/sqp_solver_amd/lib
for v in A B C A B C; do echo "== $v"; SQPH_LIB=$L/libsqp_hip_slimW4$v.so timeout 600 python tools/xp/w4_timing.py 2>&1 | tail -14; done
