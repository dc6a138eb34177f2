#!/usr/bin/env python3
"""Copy one profiles/collect.sh run from gpurun_out/prof_<tag>/ into profiles/ (replacing the previous set of the same
round) and stamp profiles/traffic.json -- only if the PMC run's kernel-source hash is the current source's.
    python tools/install_profile.py <tag> [<old tag to remove>]"""
import glob, json, os, shutil, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench

tag = sys.argv[1]
old = sys.argv[2] if len(sys.argv) > 2 else None
src = f"gpurun_out/prof_{tag}"
d = json.load(open(f"{src}/summary.json"))
h = bench.kernel_source_hash()
assert d["traffic_json"]["kernel_source_sha256"] == h, ("profile is of another source", d["traffic_json"], h)
shutil.copy(f"{src}/summary.json", f"profiles/{tag}_pmc_summary.json")
shutil.copy(glob.glob(f"{src}/trace/*/*kernel_stats.csv")[0], f"profiles/{tag}_kernel_stats.csv")
for name, dst in ((f"gpurun_out/bench_{tag}.json", f"profiles/{tag}_bench.json"),
                  (f"gpurun_out/bench_{tag}_d90.json", f"profiles/{tag}_bench_configs2_d90_k5.json")):
    if os.path.isfile(name):
        open(dst, "w").write(open(name).read().strip().splitlines()[-1] + "\n")
t = json.load(open("profiles/traffic.json"))
import datetime
stamp = datetime.datetime.fromtimestamp(os.path.getmtime(f"{src}/summary.json")).strftime("%Y-%m-%d %H:%M")
t.update(source=f"profiles/{tag}_pmc_summary.json", kernel_source_sha256=h, cosmetics_d64=d["traffic_json"]["hbm_bytes_per_hop"],
         measured_on=f"rocprofv3 PMC passes of profiles/collect.sh, run '{tag}' on a gpurun MI355X box (summary written {stamp}); "
                     "not the process that prints the bench line")
json.dump(t, open("profiles/traffic.json", "w"), indent=1)
d90 = f"gpurun_out/prof_{tag}_d90"
if os.path.isfile(f"{d90}/summary.json"):          # the D=90 / K=5 configuration, collected with HOPS=40
    e = json.load(open(f"{d90}/summary.json"))
    assert e["traffic_json"]["kernel_source_sha256"] == h
    shutil.copy(f"{d90}/summary.json", f"profiles/{tag}_d90_pmc_summary.json")
    shutil.copy(glob.glob(f"{d90}/trace/*/*kernel_stats.csv")[0], f"profiles/{tag}_d90_kernel_stats.csv")
    t = json.load(open("profiles/traffic.json"))
    t["cosmetics_d90"] = e["traffic_json"]["hbm_bytes_per_hop"]
    json.dump(t, open("profiles/traffic.json", "w"), indent=1)
if old:
    for f in glob.glob(f"profiles/{old}_*"):
        os.remove(f)
b = json.loads(open(f"profiles/{tag}_bench.json").read())
print("installed", tag, "hop", b["roofline"]["launch_ms"], "frac", b["roofline"]["frac"], "traffic", t["cosmetics_d64"])
