import csv,glob,sys
for d in sys.argv[1:]:
    for path in glob.glob(d+"/**/*kernel_stats.csv", recursive=True):
        print(d)
        for r in csv.DictReader(open(path)):
            n=r["Name"].replace("(anonymous namespace)::","").replace("void ","").split("(")[0]
            if n.startswith("k_"): print(f"   {n:28s} calls {r['Calls']:>4s} avg {float(r['AverageNs'])/1e3:8.1f} us  min {float(r['MinNs'])/1e3:8.1f}")
