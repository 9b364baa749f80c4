for i in $(seq 1 25); do SQPH_SOAK_SEED=1 python tools/xp/repro_soak.py 40 132 2>&1 | grep -A25 "FAILED" | head -40; done; echo loop done
