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
# per-dispatch durations from the kernel trace, restricted to the TIMED window of the bench run (the last `steps` launches of
# the step kernel: bench.py --eager --steps 300 --warmup 50 => the first 50 + lazy-initialisation launches are left out), so
# that the mean is comparable with the bench line's ms_per_step
for tdir in ("trace", "trace_steady"):
    for f in sorted(glob.glob(os.path.join(out, tdir, "**", "*kernel_trace.csv"), recursive=True)):
        per = collections.defaultdict(list)
        with open(f) as fh:
            for row in csv.DictReader(fh):
                per[row.get("Kernel_Name", "?")[:60]].append((int(row["Start_Timestamp"]), int(row["End_Timestamp"])))
        lines.append("== timed window ({}): step-kernel dispatches in launch order, last 300".format(tdir))
        for kn, spans in sorted(per.items()):
            if "k_step" not in kn:
                continue
            spans.sort()
            w = spans[-300:]
            dur = sorted((b - a) / 1e3 for a, b in w)
            gaps = [(w[i + 1][0] - w[i][1]) / 1e3 for i in range(len(w) - 1)]
            lines.append("{:<62s} n={} mean {:.2f} us  median {:.2f}  min {:.2f}  p90 {:.2f}  max {:.2f} | mean gap to the next launch {:.2f} us".format(
                kn, len(w), sum(dur) / len(dur), dur[len(dur) // 2], dur[0], dur[int(0.9 * len(dur))], dur[-1], sum(gaps) / max(len(gaps), 1)))
for d in ("pmc1", "pmc2", "pmc3", "pmc4", "pmc5"):
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
