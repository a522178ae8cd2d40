"""Condense rocprofv3 outputs of scripts/profile_round.sh into small text/JSON summaries under gpurun_out/
(copied to profiles/ and committed)."""
import collections, csv, glob, json, os, sys

R = sys.argv[1] if len(sys.argv) > 1 else "r01"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")


def first(pattern):
    f = glob.glob(os.path.join(OUT, pattern), recursive=True)
    return f[0] if f else None


def kernel_stats():
    f = first(f"prof_{R}/**/*kernel_stats.csv")
    if not f:
        return
    rows = list(csv.DictReader(open(f)))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    with open(os.path.join(OUT, f"{R}_kernel_stats.txt"), "w") as w:
        w.write(f"# rocprofv3 --kernel-trace --stats -- python bench.py --no-cpu-baseline  (round {R})\n")
        w.write(f"# total kernel time {tot/1e6:.2f} ms over all dispatches (warm-up + timed + 1 instrumented step)\n")
        w.write(f"{'kernel':100s} {'calls':>7s} {'total_ms':>10s} {'avg_us':>10s} {'min_us':>9s} {'max_us':>9s} {'pct':>6s}\n")
        for r in rows[:40]:
            w.write(f"{r['Name'][:100]:100s} {r['Calls']:>7s} {float(r['TotalDurationNs'])/1e6:10.3f} {float(r['AverageNs'])/1e3:10.2f} "
                    f"{float(r['MinNs'])/1e3:9.2f} {float(r['MaxNs'])/1e3:9.2f} {float(r['Percentage']):6.2f}\n")


def pmc(name, counter):
    f = first(f"pmc_{name}_{R}/**/*counter_collection.csv")
    if not f:
        return {}
    agg = collections.defaultdict(lambda: [0.0, 0, 0.0])
    seen = set()
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != counter:
            continue
        k = r["Kernel_Name"]
        a = agg[k]
        a[0] += float(r["Counter_Value"])
        if r["Dispatch_Id"] not in seen:
            seen.add(r["Dispatch_Id"])
            a[1] += 1
            a[2] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    return {k: {"sum": v[0], "launches": v[1], "ns": v[2]} for k, v in agg.items()}


def traffic():
    fe, wr = pmc("fetch", "FETCH_SIZE"), pmc("write", "WRITE_SIZE")
    if not fe and not wr:
        return
    out = {}
    for k in sorted(set(fe) | set(wr), key=lambda k: -(fe.get(k, {}).get("ns", 0))):
        f, w = fe.get(k), wr.get(k)
        if not f or not w or f["launches"] == 0:
            continue
        # counters are in KiB; gfx950 FETCH_SIZE reads exactly half of a wide coalesced streaming read
        # (MI355X_MICROARCH.md, HBM section) -> doubled; WRITE_SIZE is exact for 16-B-per-lane stores.
        fetch_b = 2.0 * f["sum"] * 1024 / f["launches"]
        write_b = w["sum"] * 1024 / max(1, w["launches"])
        out[k] = {"launches": f["launches"], "avg_us": f["ns"] / f["launches"] / 1e3,
                  "fetch_bytes_per_launch_corrected": fetch_b, "write_bytes_per_launch": write_b,
                  "hbm_bytes_per_launch": fetch_b + write_b}
    json.dump({"round": R, "note": "FETCH_SIZE (x2 gfx950 correction) and WRITE_SIZE from two separate rocprofv3 --pmc passes of bench.py",
               "kernels": dict(list(out.items())[:16])}, open(os.path.join(OUT, f"{R}_hbm_traffic.json"), "w"), indent=1)


kernel_stats()
traffic()
