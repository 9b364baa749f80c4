# per-iteration and fixed cost of the one-QP-per-lane / quad kernels: kernel time against the iteration count
for b in 1024 4096; do for it in 1 50 100 200 400 800; do
echo -n "batch $b iters $it: "; python bench.py --n 2 --m 3 --batch-per-gpu $b --mode fixed --iters $it --no-cpu-baseline --steps 50 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read()); print(r['ms_per_step'], r['roofline']['kernel_ms_avg'], r['config']['kernel'])"
done; done
