import os, sys, subprocess, json, csv, glob, shutil
for mask in (0, 256, 512, 1024, 1792):
    env = dict(os.environ, QTOS_DBG=str(mask), TMPDIR="/tmp")
    d = "/tmp/prof_%d" % mask
    shutil.rmtree(d, ignore_errors=True)
    subprocess.run(["rocprofv3", "--kernel-trace", "--stats", "--output-format", "csv", "-d", d, "--", sys.executable, "bench.py", "--steps", "3", "--warmup", "1", "--cpu-sample", "0"], env=env, capture_output=True, text=True, cwd=os.getcwd())
    f = glob.glob(d + "/*/*kernel_stats.csv")[0]
    rows = {r["Name"].split("(")[0]: float(r["AverageNs"]) / 1e6 for r in csv.DictReader(open(f)) if r["Name"].startswith("qtos")}
    print(mask, {k.replace("qtos::", ""): round(v, 3) for k, v in rows.items()})
