"""Summarises the rocprofv3 output of tests/run_profile.sh into one small text file (copied into profiles/)."""
import csv, glob, os, sys, collections
out = sys.argv[1]
lines = []
for f in sorted(glob.glob(os.path.join(out, "trace", "**", "*kernel_stats.csv"), recursive=True)):
    lines.append("== kernel stats: " + os.path.relpath(f, out))
    with open(f) as fh:
        for i, row in enumerate(csv.reader(fh)):
            if i < 8:
                lines.append(", ".join(row))
for d in ("pmc1", "pmc2", "pmc3", "pmc4"):
    for f in sorted(glob.glob(os.path.join(out, d, "**", "*counter_collection.csv"), recursive=True)):
        acc = collections.defaultdict(lambda: [0.0, 0])
        with open(f) as fh:
            for row in csv.DictReader(fh):
                k = (row.get("Kernel_Name", "?")[:60], row.get("Counter_Name", "?"))
                acc[k][0] += float(row.get("Counter_Value", 0) or 0)
                acc[k][1] += 1
        lines.append("== counters (mean per dispatch): " + os.path.relpath(f, out))
        for (kn, cn), (s, n) in sorted(acc.items()):
            lines.append("{:<62s} {:<24s} {:>16.1f}  (n={})".format(kn, cn, s / max(n, 1), n))
for f in sorted(glob.glob(os.path.join(out, "bench_*.log"))):
    with open(f) as fh:
        js = [l for l in fh if l.startswith("{")]
    lines.append("== " + os.path.basename(f) + ": " + (js[-1].strip() if js else "(no json line)"))
open(os.path.join(out, "summary.txt"), "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
