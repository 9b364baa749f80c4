# tools/xp/shape_ab.sh "<shape specs>" A B ... : shape_timing.py under slim builds libsqp_hip_slim<V>.so, alternating twice
L=$PWD/sqp_solver_amd/lib
SPECS=$1; shift
for i in 1 2; do for v in "$@"; do echo "== $v"; SQPH_LIB=$L/libsqp_hip_slim$v.so timeout 900 python tools/xp/shape_timing.py $SPECS 2>&1 | grep -v amdgpu.ids; done; done
