# A/B of two slim builds: bench lines in the three modes (alternating) and the LDS counters of the fixed-200 run of each
A=${1:-D}; B=${2:-E}
L=$PWD/sqp_solver_amd/lib
export TMPDIR=/tmp
for i in 1 2 3; do for mode in fixed default sqp; do for v in $A $B; do
echo -n "$mode $v: "; SQPH_LIB=$L/libsqp_hip_slim$v.so python bench.py --no-cpu-baseline --no-extra --steps 40 --mode $mode 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read()); print(r['ms_per_step'], r['roofline']['kernel_ms_avg'], r['config']['admm_iters_per_qp'], r['config']['kernel'])"
done; done; done
for v in $A $B; do
rm -rf gpurun_out/pmc_ab_$v; SQPH_LIB=$L/libsqp_hip_slim$v.so rocprofv3 --kernel-trace --output-format csv --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_INSTS_LDS -d gpurun_out/pmc_ab_$v -o pmc -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extra > /dev/null 2>&1
python - <<PY
import csv,glob
acc={}
for f in glob.glob("gpurun_out/pmc_ab_$v/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "admm" in r["Kernel_Name"]: acc[r["Counter_Name"]]=acc.get(r["Counter_Name"],0)+float(r["Counter_Value"])
print("$v", {k: "%.4g"%v for k,v in acc.items()}, "conflict/active = %.3f" % (acc["SQ_LDS_BANK_CONFLICT"]/acc["SQ_LDS_IDX_ACTIVE"]))
PY
rm -rf gpurun_out/pmc_ab_$v
done
