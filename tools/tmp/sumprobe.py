import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
p = d["roofline"]["probe_step_all_gemm_templates"]
tot = sum(v["avg_us"] * v["launches_per_step"] for v in p.values())
fd = sum(v["avg_us"] * v["launches_per_step"] for k, v in p.items() if "wgrad" not in k)
ws4 = sum(v["avg_us"] * v["launches_per_step"] for k, v in p.items() if "ws4" in k)
print(sys.argv[1], "ms/step", d["ms_per_step"], "serial", d["roofline"]["measured"][60:80], "gemm us", round(tot), "fwd+dgrad us", round(fd), "ws4 us", round(ws4))
