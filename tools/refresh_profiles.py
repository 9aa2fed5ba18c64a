"""Copy the judged summaries of one measurement run from gpurun_out/ (scratch) into profiles/<round> (tracked).
usage: python tools/refresh_profiles.py <bench dir under gpurun_out> <prof tag> <round dir, e.g. r02>"""
import collections, csv, datetime, glob, json, os, shutil, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
bench_dir, tag, rnd = sys.argv[1], sys.argv[2], sys.argv[3]
dst = os.path.join(R, "profiles", rnd)
src = os.path.join(R, "gpurun_out", f"prof_{tag}")
os.makedirs(dst, exist_ok=True)
# files of an earlier call that this run does not produce again stay (profiles/README.md says which call each file is from)
rows = list(csv.reader(open(glob.glob(src + "/trace/*/*_kernel_stats.csv")[0])))
with open(dst + "/c4_n1_kernel_stats.csv", "w", newline="") as f:
    w = csv.writer(f, quoting=csv.QUOTE_ALL)
    for r in rows:
        w.writerow([r[0][:140]] + r[1:])
shutil.copy(src + "/trace_bench.json", dst + "/c4_n1_bench_under_rocprof.json")
out = {}
for sub, cn in (("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE")):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(glob.glob(f"{src}/{sub}/*/*_counter_collection.csv")[0])):
        if r["Counter_Name"] == cn and "rdx" in r["Kernel_Name"]:
            agg[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        out.setdefault(k, {})[cn + "_avg_per_launch_KB"] = round(sum(v) / len(v), 1)
        out[k]["launches_" + cn] = len(v)
for k, v in out.items():
    if "FETCH_SIZE_avg_per_launch_KB" in v and "WRITE_SIZE_avg_per_launch_KB" in v:
        v["hbm_side_bytes_per_launch"] = int(2 * v["FETCH_SIZE_avg_per_launch_KB"] * 1024 + v["WRITE_SIZE_avg_per_launch_KB"] * 1024)
_tb = json.load(open(src + "/trace_bench.json"))
out["_note"] = (f"separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of `python3 bench.py --steps {_tb['steps']} --warmup {_tb['warmup']} --no-cpu`; bytes = "
                "2*FETCH_SIZE*1024 + WRITE_SIZE*1024 (gfx950: FETCH_SIZE counts half of wide coalesced reads, MI355X_MICROARCH.md HBM section)")
json.dump(out, open(dst + "/c4_n1_pmc_summary.json", "w"), indent=1)
main = [v for k, v in out.items() if k.startswith("void rdx::k_scan<256, 1, false, false, false, false")][0]["hbm_side_bytes_per_launch"]
json.dump({"c4_n1": {"bytes_per_launch": main,
                     "source": f"profiles/{rnd}/c4_n1_pmc_summary.json (main scan k_scan<256,1,false,false,...>, rocprofv3 --pmc run of "
                               f"{datetime.date.today().isoformat()})"}},
          open(R + "/profiles/traffic.json", "w"), indent=1)
for f in glob.glob(os.path.join(R, "gpurun_out", bench_dir, "*.json")):
    shutil.copy(f, os.path.join(dst, os.path.basename(f).replace(".json", "_n1_bench.json")))
for extra in glob.glob(os.path.join(R, "gpurun_out", bench_dir, "*.txt")) + glob.glob(os.path.join(R, "gpurun_out", bench_dir, "*_kernel_stats.csv")):
    shutil.copy(extra, os.path.join(dst, os.path.basename(extra)))
print("traffic", main, sorted(os.listdir(dst)))
