import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print(sys.argv[1], "ms/step", d["ms_per_step"], "windows/s", d["value"], "host_enqueue", d["config"].get("host_enqueue_ms_per_step"))
