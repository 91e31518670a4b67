"""Kernel evidence of ONE bench command into profiles/: rocprofv3 kernel stats + the PMC passes (HBM traffic, SQ counters) that
scratch/r5_trot_prof.sh (or final_measure*.sh) left under gpurun_out/*_<tag>.
usage: python scratch/collect_kernel_evidence.py <tag> <prefix>      e.g.  r5trot r05_trot"""
import csv, collections, json, os, shutil, sys
tag, R = sys.argv[1], sys.argv[2]
G = "gpurun_out/"
f = G + "bench_%s.json" % tag
if os.path.exists(f):
    lines = open(f).read().strip().splitlines()
    json.loads(lines[-1])
    open("profiles/%s_bench.json" % R, "w").write(lines[-1] + "\n")
shutil.copy(G + "prof_%s/runc_kernel_stats.csv" % tag, "profiles/%s_kernel_stats.csv" % R)

def per_kernel(f):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        if "qtos::" in r["Kernel_Name"]:
            agg[r["Kernel_Name"].split("(")[0].replace("void ", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return agg

out = {}
for name, f in (("FETCH_SIZE", G + "pmc_fetch_%s/runc_counter_collection.csv" % tag), ("WRITE_SIZE", G + "pmc_write_%s/runc_counter_collection.csv" % tag)):
    agg = per_kernel(f)
    out[name] = {k: {"dispatches": len(v[name]), "mean_KB": sum(v[name]) / len(v[name])} for k, v in agg.items()}
kk = [k for k in out["FETCH_SIZE"] if "k_kkt" in k][0]
f_, w_ = out["FETCH_SIZE"][kk]["mean_KB"], out["WRITE_SIZE"][kk]["mean_KB"]
out["note"] = ("rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE, separate passes of the bench command with --steps 3 --warmup 1 "
               "(batch 256). KB per dispatch. gfx950: FETCH_SIZE reports 1/2 of the bytes of wide coalesced reads -> HBM bytes per "
               "launch between (F+W)*1024 and (2F+W)*1024; bench.py reports the larger.")
out["k_kkt_kernel"] = kk
out["k_kkt_traffic_bytes_per_launch"] = {"raw": (f_ + w_) * 1024, "fetch_x2": (2 * f_ + w_) * 1024}
kc = [k for k in out["FETCH_SIZE"] if "k_chord" in k]
if kc:
    f2, w2 = out["FETCH_SIZE"][kc[0]]["mean_KB"], out["WRITE_SIZE"][kc[0]]["mean_KB"]
    out["k_chord_traffic_bytes_per_launch"] = {"raw": (f2 + w2) * 1024, "fetch_x2": (2 * f2 + w2) * 1024}
json.dump(out, open("profiles/%s_pmc_hbm.json" % R, "w"), indent=1)
print(json.dumps(out["k_kkt_traffic_bytes_per_launch"]), json.dumps(out.get("k_chord_traffic_bytes_per_launch")))
sq = {}
for sub in ("pmc_sq_%s" % tag, "pmc_sq2_%s" % tag):
    f = G + sub + "/runc_counter_collection.csv"
    if not os.path.exists(f):
        continue
    for k, v in per_kernel(f).items():
        d = sq.setdefault(k, {})
        for c, x in v.items():
            d[c] = sum(x) / len(x)
            d["dispatches"] = len(x)
summ = {"note": "rocprofv3 --pmc passes of the bench command with --steps 3 --warmup 1; means per dispatch, summed over the chip by the "
                "profiler. SQ_*_CYCLES / SQ_WAIT* / SQ_ACTIVE_INST_* count quad-cycles (x4 = cycles); SQ_VALU_MFMA_BUSY_CYCLES counts "
                "cycles per SIMD. 256 CUs x 4 SIMDs.", "raw": sq}
for k, d in sq.items():
    if "k_kkt" in k and "SQ_WAVE_CYCLES" in d:
        waves = d.get("SQ_WAVES", 4096.0)
        wave_cycles = 4.0 * d["SQ_WAVE_CYCLES"] / waves
        e = {"kernel": k, "cycles_per_launch": round(wave_cycles), "waves": waves}
        if "SQ_VALU_MFMA_BUSY_CYCLES" in d:
            e["mfma_busy_frac_of_simd_cycles"] = round(d["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024.0 / wave_cycles, 4)
        if "SQ_ACTIVE_INST_VALU" in d:
            e["valu_busy_frac_of_simd_cycles"] = round(4.0 * d["SQ_ACTIVE_INST_VALU"] / 1024.0 / wave_cycles, 4)
        if "SQ_ACTIVE_INST_LDS" in d:
            e["lds_inst_busy_frac_of_cu_cycles"] = round(4.0 * d["SQ_ACTIVE_INST_LDS"] / 256.0 / wave_cycles, 4)
        if "SQ_WAIT_INST_ANY" in d:
            e["wave_cycles_waiting_on_instructions_frac"] = round(d["SQ_WAIT_INST_ANY"] / d["SQ_WAVE_CYCLES"], 4)
            e["wave_cycles_waiting_any_frac"] = round(d["SQ_WAIT_ANY"] / d["SQ_WAVE_CYCLES"], 4)
        if "SQ_LDS_BANK_CONFLICT" in d and "SQ_LDS_IDX_ACTIVE" in d:
            e["lds_bank_conflict_frac_of_lds_active"] = round(d["SQ_LDS_BANK_CONFLICT"] / max(d["SQ_LDS_IDX_ACTIVE"], 1.0), 4)
        if "SQ_INSTS_VALU" in d:
            e["valu_instructions_per_problem"] = round(d["SQ_INSTS_VALU"] / 256.0)
            e["lds_instructions_per_problem"] = round(d.get("SQ_INSTS_LDS", 0) / 256.0)
        if "SQ_INSTS_VALU_MFMA_MOPS_F64" in d:
            e["mfma_mflop_per_problem"] = round(d["SQ_INSTS_VALU_MFMA_MOPS_F64"] * 512 / 256.0 / 1e6, 2)
        summ["k_kkt"] = e
json.dump(summ, open("profiles/%s_pmc_sq.json" % R, "w"), indent=1)
print(json.dumps(summ.get("k_kkt")))
for l in open("profiles/%s_kernel_stats.csv" % R).read().splitlines()[:6]:
    print(l[:160])
