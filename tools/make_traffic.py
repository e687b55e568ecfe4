#!/usr/bin/env python3
"""profiles/traffic.json from the rocprofv3 PMC summaries (tests/run_profile.sh -> tests/summarize_prof.py): HBM-side bytes
per k_step launch = FETCH_SIZE [KB] x 1024 x 2 (gfx950: the counter reports half of a wide coalesced stream,
MI355X_MICROARCH.md "HBM") + WRITE_SIZE [KB] x 1024, each from its own --pmc pass, tagged with the hash of the kernel
sources the profile was taken on (bench.py reports the figure only for the same build)."""
import json, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench


def parse(path, kernel):
    vals = {}
    with open(path) as f:
        for line in f:
            m = re.match(r"^(.*?)\s{2,}(FETCH_SIZE|WRITE_SIZE)\s+([0-9.]+)\s+\(n=(\d+)\)", line)
            if m and kernel in m.group(1):
                vals[m.group(2)] = float(m.group(3))
    return vals


out = {}
TAG = sys.argv[1] if len(sys.argv) > 1 else "v15"
RND = sys.argv[2] if len(sys.argv) > 2 else "r04"
for key, fn, kernel in (("c3_log", "{}_c3_rocprofv3_summary_{}.txt".format(RND, TAG), "k_step2<true, 6>"),
                        ("c3", "{}_c3_dense_rocprofv3_summary_{}.txt".format(RND, TAG), "k_step2<true, 4>")):
    p = os.path.join(ROOT, "profiles", fn)
    if not os.path.exists(p):
        continue
    v = parse(p, kernel)
    b = int(v["FETCH_SIZE"] * 1024 * 2 + v["WRITE_SIZE"] * 1024)
    out[key] = {"bytes_per_launch": b, "source_hash": bench.source_hash(),
                "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes (profiles/{}): mean per {} dispatch FETCH_SIZE {:.1f} KB "
                          "x 1024 x 2 (gfx950 counter correction) + WRITE_SIZE {:.1f} KB x 1024; 65536 envs; algorithmic 61.4 MB (937 B per env-step)".format(
                              fn, kernel, v["FETCH_SIZE"], v["WRITE_SIZE"])}
with open(os.path.join(ROOT, "profiles", "traffic.json"), "w") as f:
    json.dump(out, f, indent=2)
print(json.dumps(out, indent=1))
