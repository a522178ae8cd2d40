import csv, collections, sys, os, glob
d = sys.argv[1]
f = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)[0]
rows = list(csv.DictReader(open(f)))
agg = collections.defaultdict(lambda: collections.defaultdict(float)); dur = collections.defaultdict(float); n = collections.Counter()
seen = set()
for r in rows:
    k = r['Kernel_Name'][:70]
    agg[k][r['Counter_Name']] += float(r['Counter_Value'])
    key = (r['Dispatch_Id'])
    if key not in seen:
        seen.add(key); dur[k] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])); n[k] += 1
for k, v in agg.items():
    if dur[k] < 2e5: continue
    print(f"{k}  launches={n[k]} total_us={dur[k]/1e3:.1f}")
    for c in sorted(v): print(f"     {c:30s} {v[c]:16.0f}   per_us={v[c]/(dur[k]/1e3):12.1f}")
