import sys,json
for line in sys.stdin:
    line=line.strip()
    if not line.startswith('{'): continue
    d=json.loads(line); print(d["config"]["kernel"], d["config"]["mode"], "ms/step", round(d["ms_per_step"],3), "QP/s", int(d["value"]), "iters/s %.3g"%d["admm_iters_per_sec"], "TF %.2f"%d["roofline"]["onchip_f64_tflops"], "hbm frac %.4f"%d["roofline"]["frac"])
