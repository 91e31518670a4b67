"""Copy the artefacts of scratch/final_measure6.sh (gpurun_out/*_<tag>*) into profiles/ as the round's set: one bench line per
configuration, then -- through scratch/collect_kernel_evidence.py -- kernel stats, per-launch HBM traffic and the SQ-counter summary
of the KKT kernel for the trot (the default command's main leg: profiles/r06_trot_*) and the walk (profiles/r06_*).
usage: python scratch/collect_profiles6.py <tag> [r06]"""
import json, os, shutil, subprocess, sys
tag = sys.argv[1]
R = sys.argv[2] if len(sys.argv) > 2 else "r06"
G = "gpurun_out/"
names = {"default": "bench", "driver_cmd": "bench_driver_cmd", "trot": "bench_trot", "walk": "bench_walk", "trot_no_pattern": "bench_trot_no_pattern", "walk_no_pattern": "bench_walk_no_pattern",
         "compat": "bench_reference_compat", "compat_trot": "bench_reference_compat_trot", "exp5": "bench_exp5", "mixed": "bench_mixed",
         "exp5_no_pattern": "bench_exp5_no_pattern", "mixed_no_pattern": "bench_mixed_no_pattern",
         "tol1e-3": "bench_tol1e-3", "batch512": "bench_batch512", "batch1024": "bench_batch1024", "knots200": "bench_knots200",
         "mpc": "bench_knots200_mpc_random", "mpc_no_pattern": "bench_knots200_mpc_random_no_pattern", "mpc_1set": "bench_knots200_mpc_random_one_set", "table": "bench_init_table",
         "superlinear_mu_mpc": "bench_superlinear_mu_knots200_mpc_random",
         "nochord": "bench_no_chord_step", "torchrun1": "bench_torchrun_1rank", "full_system": "bench_full_system",
         "exp5_lanes3": "bench_exp5_lanes3", "exp5_batch1024": "bench_exp5_batch1024", "mixed_batch1024": "bench_mixed_batch1024", "mixed_lanes3": "bench_mixed_lanes3", "lanes2": "bench_flat_lanes2", "steps500": "bench_steps500",
         "order0_walk": "bench_order0_walk", "order0_trot": "bench_order0_trot", "order0_compat": "bench_order0_reference_compat", "order0_compat_trot": "bench_order0_reference_compat_trot", "order0_exp5": "bench_order0_exp5", "order0_mixed": "bench_order0_mixed", "order0_knots200": "bench_order0_knots200",
         "order0_mpc": "bench_order0_knots200_mpc_random",
         "kkt5_walk": "bench_kkt5_walk", "kkt5_trot": "bench_kkt5_trot", "kkt2_trot": "bench_kkt2_trot", "kkt2_walk": "bench_kkt2_walk"}
for src, dst in names.items():
    f = G + "bench_%s_%s.json" % (tag, src)
    if not os.path.exists(f):
        print("missing", f); continue
    lines = open(f).read().strip().splitlines()
    if not lines:
        print("empty", f); continue
    json.loads(lines[-1])
    open("profiles/%s_%s.json" % (R, dst), "w").write(lines[-1] + "\n")
here = os.path.dirname(os.path.abspath(__file__))
for t, pre in ((tag + "trot", R + "_trot"), (tag + "walk", R)):
    if not os.path.exists(G + "prof_%s/runc_kernel_stats.csv" % t):
        print("no kernel evidence for", t); continue
    subprocess.check_call([sys.executable, os.path.join(here, "collect_kernel_evidence.py"), t, pre + "_tmp"])
    for suf in ("bench.json", "kernel_stats.csv", "pmc_hbm.json", "pmc_sq.json"):
        src = "profiles/%s_tmp_%s" % (pre, suf)
        if not os.path.exists(src):
            continue
        dst = "profiles/%s_%s" % (pre, {"bench.json": "evidence_bench.json", "kernel_stats.csv": "walk_kernel_stats.csv" if pre == R else "kernel_stats.csv"}.get(suf, suf))
        shutil.move(src, dst)
if os.path.exists(G + "prof_%sdefault/runc_kernel_stats.csv" % tag):   # the default command (both gaits) under rocprofv3
    shutil.copy(G + "prof_%sdefault/runc_kernel_stats.csv" % tag, "profiles/%s_bench_kernel_stats.csv" % R)
d = json.loads(open("profiles/%s_bench.json" % R).read())
print(d["value"], d["ms_per_step"], d["roofline"], d.get("walk", {}).get("value"), d["cpu_baseline"]["value"])
