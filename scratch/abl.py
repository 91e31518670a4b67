import os, sys, subprocess, json
for mask in (0, 1, 2, 4, 8, 16, 32, 64, 127):
    env = dict(os.environ, QTOS_DBG=str(mask))
    out = subprocess.run([sys.executable, "bench.py", "--steps", "3", "--warmup", "1", "--cpu-sample", "0"], env=env, capture_output=True, text=True).stdout.strip().split("\n")[-1]
    try:
        d = json.loads(out)
        print(mask, d["roofline"]["avg_launch_ms"], d["ms_per_step"], d["config"]["iterations_mean"])
    except Exception as e:
        print(mask, "ERR", out[-200:])
