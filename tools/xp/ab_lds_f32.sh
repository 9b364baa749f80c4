# A/B of two slim builds, fp32-product kernels (float interface, --f32-arith): kernel times alternating, parity columns, LDS counters
A=${1:-A}; B=${2:-B}
L=$PWD/sqp_solver_amd/lib
export TMPDIR=/tmp
for i in 1 2 3; do for wl in c3 c2; do for mode in fixed default; do for v in $A $B; do
echo -n "$wl $mode $v: "; SQPH_LIB=$L/libsqp_hip_slim$v.so python bench.py --workload $wl --dtype f32 --f32-arith --no-extra --steps 40 --mode $mode 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read()); print(r['ms_per_step'], r['roofline']['kernel_ms_avg'], r['config']['kernel'], r['cpu_baseline'].get('parity_max_rel_err_x'), r['cpu_baseline'].get('parity_max_rel_err_y'))"
done; done; done; done
for wl in c3 c2; do for v in $A $B; do
rm -rf gpurun_out/pmc_ab_$v; SQPH_LIB=$L/libsqp_hip_slim$v.so rocprofv3 --kernel-trace --output-format csv --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_INSTS_LDS -d gpurun_out/pmc_ab_$v -o pmc -- python bench.py --workload $wl --dtype f32 --f32-arith --steps 10 --warmup 3 --no-cpu-baseline --no-extra > /dev/null 2>&1
python - <<PY
import csv,glob
acc={}
for f in glob.glob("gpurun_out/pmc_ab_$v/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "admm" in r["Kernel_Name"]: acc[r["Counter_Name"]]=acc.get(r["Counter_Name"],0)+float(r["Counter_Value"])
print("$wl $v", {k: "%.4g"%v for k,v in acc.items()}, "conflict/active = %.3f" % (acc["SQ_LDS_BANK_CONFLICT"]/acc["SQ_LDS_IDX_ACTIVE"]))
PY
rm -rf gpurun_out/pmc_ab_$v
done; done
