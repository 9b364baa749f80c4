# kernel time of the quad (batch <= 16384) and lane (beyond) variants at the SQP driver's shape, device-resident inputs
for mode in fixed sqp default; do for b in 1024 16384 16385 65536; do
echo -n "$mode batch $b: "; python bench.py --n 2 --m 3 --batch-per-gpu $b --mode $mode --no-cpu-baseline --steps 50 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read()); print(r['ms_per_step'], r['roofline']['kernel_ms_avg'], r['config']['admm_iters_per_qp'], r['config']['kernel'])"
done; done
