#!/usr/bin/env python3
"""Do the oracle tests notice a known bug?  Puts the terminal-observation bugs that lived in the kernels through rounds 1-5 back into
a COPY of csrc/ (the fixes were commits b1ce333 and 8b54a4a, and round 6's log_plane fix), builds it, and runs the round-6 coverage
tests against the mutant: every mutant must be caught ("What's weak" 2 / "Next round" 1).

    python tools/mutation_check.py emu                 # host emulation: the steady-state sampled test and the fuzzer, here
    python tools/mutation_check.py hip OUTDIR          # gfx950 libraries of the mutants (tools/devlib.py), for a gpurun session:
                                                       # FWGYM_MUTANT_LIB=... python -m pytest tests/test_gpu_oracle_coverage.py

Never the product: the mutated sources live in a temporary directory."""
import os
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "fixed-wing-gym_amd", "csrc")

MUTANTS = {
    # b1ce333 reverted: after a FAILED step under turbulence the air data are re-derived with the failed step's gust
    "airdata": [("fwgym.hip", """                if (TURB) {   // (air data: as the last committed step left them -- derived with its gust, see store_sim)
                    const float4 q = CGROUP(A.S, A.N, (c.L.derived >> 2) + 1, e);
                    E.d.alpha = q.x; E.d.beta = q.y; E.d.Va = q.z;
                }
""", "")],
    # 8b54a4a reverted: a foreseen end whose last step fails is installed by the physics wave all the same in the dense layout
    # (only the row log's partner_rows rule holds it back)
    "install_on_failed_last_step": [
        ("fwgym.hip", "        if (pre_install) pre_rows = pre_rows && fail == 0;", "        if (partner_rows) pre_rows = pre_rows && fail == 0;"),
        ("fwgym.hip", "            end_p = pre_install && valid && f2u(w.z) != 0u && fail == 0;",
         "            end_p = pre_install && valid && f2u(w.z) != 0u && !(partner_rows && fail != 0);")],
    # round 6's own find reverted (log_plane: a record one further back than the carried rows is at its home plane): the oldest
    # row of a failed step's terminal observation on the row log's wrap step, obs_step 1 (the shipped cnn configuration)
    "log_plane_past_the_log": [
        ("fwgym_env.h", "    return (long long)p * L + home + ((wraps > 0 && home <= len - 2) ? (long long)P * wraps : 0);",
         "    return (long long)p * L + home + (long long)P * wraps;")],
}


def mutated_tree(name):
    d = tempfile.mkdtemp(prefix="fwg_mut_{}_".format(name))
    dst = os.path.join(d, "fixed-wing-gym_amd", "csrc")
    shutil.copytree(CSRC, dst)
    shutil.copytree(os.path.join(ROOT, "include"), os.path.join(d, "include"))
    for fn, old, new in MUTANTS[name]:
        p = os.path.join(dst, fn)
        s = open(p).read()
        assert s.count(old) == 1, "mutation site of {} not found exactly once in {}: the source moved, update tools/mutation_check.py".format(name, fn)
        open(p, "w").write(s.replace(old, new))
    return d, dst


def _hip_mutant(job):
    name, layout, outdir = job
    sys.path[:0] = [ROOT, os.path.join(ROOT, "fixed-wing-gym_amd"), os.path.join(ROOT, "tests")]
    import copy
    from gym_fixed_wing import jit, presets
    d, dst = mutated_tree(name)
    cache = tempfile.mkdtemp(prefix="fwg_mutcache_")
    jit.CSRC, jit.CACHE = dst, cache
    cfg, ckw, skw, _, _ = presets.workload("c3")
    ckw = dict(copy.deepcopy(ckw), steps_max=45, simulator={"states": {6: {"constraint_min": -60, "constraint_max": 60}}})
    lib = jit.prebuild(cfg, ckw, skw, derived_views=False, obs_log_rows=presets.OBS_LOG_ROWS if layout == "row_log" else 0)
    assert lib is not None, (name, layout)
    out = os.path.join(outdir, "libfwgym_mut_{}_{}.so".format(name, layout))
    shutil.move(lib, out)
    shutil.rmtree(d), shutil.rmtree(cache)
    return out


def main():
    mode = sys.argv[1] if len(sys.argv) > 1 else "emu"
    if mode == "hip":
        # the fail-prone steady-state configuration of tests/test_gpu_oracle_coverage.py, frozen (jit.py's build recipe) from the
        # mutated sources, row log and dense: libfwgym_mut_<name>_<layout>.so under OUTDIR
        outdir = os.path.abspath(sys.argv[2])
        os.makedirs(outdir, exist_ok=True)
        sys.path[:0] = [ROOT, os.path.join(ROOT, "fixed-wing-gym_amd"), os.path.join(ROOT, "tests")]
        import copy
        from concurrent.futures import ProcessPoolExecutor
        jobs = [(name, layout, outdir) for name in ("airdata", "install_on_failed_last_step") for layout in ("row_log", "dense")]
        with ProcessPoolExecutor(max_workers=4) as pool:
            for out in pool.map(_hip_mutant, jobs):
                print("built", out)
        return 0
    sys.path[:0] = [ROOT, os.path.join(ROOT, "fixed-wing-gym_amd"), os.path.join(ROOT, "tests")]
    caught = {}
    for name in MUTANTS:
        d, dst = mutated_tree(name)
        env = dict(os.environ, FWGYM_MUTANT_SRC=dst, FWGYM_MUTANT_TAG="_mut_" + name)
        r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-p", "no:cacheprovider", "-k", "not frozen_kernel_through",
                            os.path.join(ROOT, "tests", "test_emu_coverage.py"), os.path.join(ROOT, "tests", "test_emu_fuzz.py")],
                           env=env, capture_output=True, text=True)
        caught[name] = r.returncode != 0
        tail = [l for l in r.stdout.splitlines() if "Mismatch" in l or "FAILED" in l or "passed" in l or "failed" in l][-6:]
        print("mutant {:32s} {}".format(name, "CAUGHT" if caught[name] else "SURVIVED"))
        for l in tail:
            print("    " + l[:220])
        shutil.rmtree(d)
        for f in os.listdir(os.path.join(ROOT, "tests", "emu")):
            if "_mut_" in f:
                os.remove(os.path.join(ROOT, "tests", "emu", f))
    return 0 if all(caught.values()) else 1


if __name__ == "__main__":
    sys.exit(main())
