"""Dev tool: per-kernel means of arbitrary rocprofv3 --pmc counters (csv output dirs given on the command line)."""
import collections, csv, os, sys
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for d in sys.argv[1:]:
    for root, _, files in os.walk(d):
        for f in files:
            if f.endswith("counter_collection.csv"):
                for x in csv.DictReader(open(os.path.join(root, f))):
                    k = x["Kernel_Name"].split("(")[0].replace("void ", "").replace("es::", "")
                    acc[(k, int(x["Grid_Size"]))][x["Counter_Name"]].append(float(x["Counter_Value"]))
names = sorted({c for v in acc.values() for c in v})
print("kernel grid n " + " ".join(names))
for key in sorted(acc, key=lambda k: -sum(acc[k].get("SQ_WAVE_CYCLES", [0]))):
    if not key[0].startswith("k_"):
        continue
    c = acc[key]
    print(key[0], key[1], len(next(iter(c.values()))), " ".join("%.4g" % (sum(c[n]) / len(c[n])) if n in c else "-" for n in names))
