"""Copy the artefacts of scratch/final_measure.sh (gpurun_out/*_<tag>*) into profiles/ as the round's
final set and derive the per-launch HBM traffic of k_kkt from the two PMC passes."""
import csv, collections, json, shutil, sys
tag = sys.argv[1]
G = "gpurun_out/"
pairs = {"bench_%s.json": "r01_final_bench.json", "bench_%s_compat.json": "r01_final_bench_reference_compat.json",
         "bench_%s_exp5.json": "r01_final_bench_exp5.json", "bench_%s_mixed.json": "r01_final_bench_mixed.json",
         "bench_%s_mixed_inflight3.json": "r01_final_bench_mixed_inflight3.json",
         "bench_%s_flat_inflight2.json": "r01_final_bench_flat_inflight2.json",
         "bench_%s_knots200.json": "r01_final_bench_knots200.json",
         "bench_%s_mpc200_iter6.json": "r01_final_bench_knots200_mpc_random_maxiter6.json",
         "bench_%s_mpc200.json": "r01_final_bench_knots200_mpc_random.json",
         "bench_%s_table.json": "r01_final_bench_init_table.json",
         "bench_%s_table_compat.json": "r01_final_bench_init_table_reference_compat.json",
         "bench_%s_table_knots200.json": "r01_final_bench_init_table_knots200.json"}
for src, dst in pairs.items():
    line = open(G + src % tag).read().strip().splitlines()[-1]
    json.loads(line)
    open("profiles/" + dst, "w").write(line + "\n")
shutil.copy(G + "prof_%s/runc_kernel_stats.csv" % tag, "profiles/r01_final_bench_kernel_stats.csv")
out = {}
for name, f in (("FETCH_SIZE", G + "pmc_fetch_%s/runc_counter_collection.csv" % tag), ("WRITE_SIZE", G + "pmc_write_%s/runc_counter_collection.csv" % tag)):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "qtos::" in r["Kernel_Name"]:
            agg[r["Kernel_Name"].split("(")[0].replace("void ", "")].append(float(r["Counter_Value"]))
    out[name] = {k: {"dispatches": len(v), "mean_KB": sum(v) / len(v)} for k, v in agg.items()}
kk = [k for k in out["FETCH_SIZE"] if "k_kkt" in k][0]
f, w = out["FETCH_SIZE"][kk]["mean_KB"], out["WRITE_SIZE"][kk]["mean_KB"]
out["note"] = ("rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE, separate passes of `python3 bench.py --steps 3 --warmup 1 "
               "--cpu-sample 0` (batch 256, knots100). KB per dispatch. gfx950: FETCH_SIZE reports 1/2 of the bytes of wide coalesced "
               "reads (calibrated for 16 B/lane; k_kkt reads 8-16 B/lane) -> HBM bytes per k_kkt launch between (F+W)*1024 and "
               "(2F+W)*1024; bench.py reports the larger.")
out["k_kkt_traffic_bytes_per_launch"] = {"raw": (f + w) * 1024, "fetch_x2": (2 * f + w) * 1024}
json.dump(out, open("profiles/r01_final_pmc_hbm.json", "w"), indent=1)
print(json.dumps(out["k_kkt_traffic_bytes_per_launch"]))
for l in open("profiles/r01_final_bench_kernel_stats.csv").read().splitlines()[:5]:
    print(l[:150])
d = json.loads(open("profiles/r01_final_bench.json").read())
print(d["value"], d["ms_per_step"], d["roofline"], d["cpu_baseline"])
