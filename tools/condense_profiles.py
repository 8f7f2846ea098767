#!/usr/bin/env python3
"""Condenses the output of tools/collect_profiles.sh (DIR) into the small files kept under profiles/:
   tools/condense_profiles.py DIR TAG"""
import collections, csv, glob, json, os, re, shutil, sys
src, tag = sys.argv[1], sys.argv[2]
os.makedirs("profiles", exist_ok=True)
def one(pattern):
    f = glob.glob(pattern, recursive=True)
    return f[0] if f else None
for d in sorted(glob.glob(f"{src}/stats_*")):
    if not os.path.isdir(d):
        continue
    wl = os.path.basename(d)[len("stats_"):]
    f = one(f"{d}/**/*kernel_stats.csv")
    if f:
        shutil.copy(f, f"profiles/{tag}_{wl}_kernel_stats.csv")
def kname(full):
    m = re.search(r"zg::(k_\w+(?:<\d+>)?)", full)
    return m.group(1) if m else full.split("(")[0]
def pmc(d):
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    calls = collections.Counter()
    seen = set()
    for f in glob.glob(f"{d}/**/*_counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = kname(r["Kernel_Name"])
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
            key = (k, r["Dispatch_Id"])
            if key not in seen:
                seen.add(key)
                calls[k] += 1
    return agg, calls
traffic = {}
for d in sorted(glob.glob(f"{src}/fetch_*")):
    if not os.path.isdir(d):
        continue
    wl = os.path.basename(d)[len("fetch_"):]
    fa, fc = pmc(d)
    wa, wc = pmc(f"{src}/write_{wl}")
    t = {}
    for k in fa:
        n = max(1, fc[k])
        fetch_kb = fa[k].get("FETCH_SIZE", 0.0) / n
        write_kb = wa.get(k, {}).get("WRITE_SIZE", 0.0) / max(1, wc.get(k, 1))
        # counters are in KiB-like units of 1024 B; on gfx950 FETCH_SIZE tallies 128-byte requests at 64 bytes
        t[k] = {"launches": n, "fetch_size_kb_per_launch": fetch_kb, "write_size_kb_per_launch": write_kb,
                "hbm_bytes_per_launch": 2 * fetch_kb * 1024 + write_kb * 1024}
    traffic[wl] = t
for wl, t in traffic.items():  # mean kernel durations of the same workload, for a check against a later run
    f = one(f"{src}/stats_{wl}/**/*kernel_stats.csv")
    if not f:
        continue
    for r in csv.DictReader(open(f)):
        k = kname(r["Name"])
        if k in t:
            t[k]["avg_ms_in_stats_run"] = float(r["AverageNs"]) / 1e6
json.dump({"note": "per launch, one bench step; hbm_bytes = 2 x FETCH_SIZE + WRITE_SIZE (FETCH_SIZE counts wide reads at half "
                   "their size on gfx950); separate rocprofv3 --pmc passes", "workloads": traffic},
          open(f"profiles/{tag}_traffic.json", "w"), indent=1)
for wl in ("mixed8g", "text64m", "records4g"):
    if not os.path.isdir(f"{src}/sqa_{wl}"):
        continue
    sa, _ = pmc(f"{src}/sqa_{wl}")
    sb, _ = pmc(f"{src}/sqb_{wl}")
    with open(f"profiles/{tag}_{wl}_sq_counters.txt", "w") as o:
        for k in sorted(set(sa) | set(sb)):
            v = {**sa.get(k, {}), **sb.get(k, {})}
            if v.get("SQ_LDS_IDX_ACTIVE"):
                v["lds_conflict_frac"] = v.get("SQ_LDS_BANK_CONFLICT", 0.0) / v["SQ_LDS_IDX_ACTIVE"]
            o.write(k + " " + json.dumps({a: float(f"{b:.4g}") for a, b in sorted(v.items())}) + "\n")
# the bench lines themselves, one file per workload, and the kernels' mean durations beside the traffic figures
for f in sorted(glob.glob(f"{src}/bench_*.json")):
    wl = os.path.basename(f)[len("bench_"):-len(".json")]
    try:
        line = json.loads(open(f).read().strip().splitlines()[-1])
    except (ValueError, IndexError):
        continue
    json.dump(line, open(f"profiles/{tag}_bench_{wl}.json", "w"), indent=1)
print(open(f"profiles/{tag}_traffic.json").read()[:3000])
