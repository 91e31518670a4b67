"""Copy the artefacts of scratch/final_measure5.sh (gpurun_out/*_<tag>*) into profiles/ as the round's set: one bench line per
configuration, then -- through scratch/collect_kernel_evidence.py -- kernel stats, per-launch HBM traffic and the SQ-counter summary
of the KKT kernel for the default command (walk), the trot and k_kkt5 (QTOS_KKT=6).
usage: python scratch/collect_profiles.py <tag> [r05]"""
import json, os, shutil, subprocess, sys
tag = sys.argv[1]
R = sys.argv[2] if len(sys.argv) > 2 else "r05"
G = "gpurun_out/"
names = {"default": "bench", "compat": "bench_reference_compat", "exp5": "bench_exp5", "mixed": "bench_mixed", "trot": "bench_trot",
         "tol1e-3": "bench_tol1e-3", "batch512": "bench_batch512", "batch1024": "bench_batch1024", "knots200": "bench_knots200",
         "mpc": "bench_knots200_mpc_random", "mpc_1set": "bench_knots200_mpc_random_one_set", "table": "bench_init_table",
         "nochord": "bench_no_chord_step", "torchrun1": "bench_torchrun_1rank", "full_system": "bench_full_system",
         "exp5_lanes3": "bench_exp5_lanes3", "exp5_batch1024": "bench_exp5_batch1024", "mixed_batch1024": "bench_mixed_batch1024", "mixed_lanes3": "bench_mixed_lanes3", "lanes2": "bench_flat_lanes2", "steps500": "bench_steps500",
         "kkt5_walk": "bench_kkt5_walk", "kkt5_trot": "bench_kkt5_trot", "kkt5_compat": "bench_kkt5_reference_compat", "kkt5_knots200": "bench_kkt5_knots200",
         "kkt2_trot": "bench_kkt2_trot", "kkt2_walk": "bench_kkt2_walk", "no_swing": "bench_no_reduce_swing", "no_swing_trot": "bench_no_reduce_swing_trot",
         "r4_system": "bench_round4_system", "r4_system_trot": "bench_round4_system_trot", "no_short": "bench_no_short_stages", "no_short_trot": "bench_no_short_stages_trot",
         "plain_mu": "bench_plain_mu", "plain_mu_trot": "bench_plain_mu_trot", "plain_mu_exp5": "bench_plain_mu_exp5", "plain_mu_mixed": "bench_plain_mu_mixed",
         "plain_mu_knots200": "bench_plain_mu_knots200", "superlinear_mu_mpc": "bench_superlinear_mu_knots200_mpc_random"}
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
for t, pre in ((tag, R), (tag + "trot", R + "_trot"), (tag + "k5", R + "_kkt5")):
    if not os.path.exists(G + "prof_%s/runc_kernel_stats.csv" % t):
        print("no kernel evidence for", t); continue
    subprocess.check_call([sys.executable, os.path.join(here, "collect_kernel_evidence.py"), t, pre])
shutil.copy("profiles/%s_kernel_stats.csv" % R, "profiles/%s_bench_kernel_stats.csv" % R)   # (the name of rounds 1 - 4)
os.remove("profiles/%s_kernel_stats.csv" % R)
d = json.loads(open("profiles/%s_bench.json" % R).read())
print(d["value"], d["ms_per_step"], d["roofline"], d["cpu_baseline"])
