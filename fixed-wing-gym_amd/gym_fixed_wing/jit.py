"""Run-time kernel specialisation for configurations that are not among the build-time presets.

The generic kernel interprets the configuration (scalar loads, config-driven branches, LDS tables) and is 2-3x slower
than a specialised instance.  `specialised_library(...)` lowers the configuration through the already-loaded library
(fwg_dump_spec), freezes it into a one-entry specs file and compiles a private copy of libfwgym with hipcc
(--offload-arch=gfx950, ~20 s once per distinct configuration, cached by content hash).  Default whenever
hipcc is on the machine (FixedWingVecEnv(..., specialize=False) / FWGYM_JIT=0 opt out); failures fall back to the generic
kernel of the stock library with a warning (still the HIP path, never a CPU path)."""
import hashlib
import os
import shutil
import subprocess
import warnings

from . import _native as nat
from . import specialize

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(os.path.dirname(_HERE), "csrc")
INCLUDE = os.path.join(os.path.dirname(os.path.dirname(_HERE)), "include")
CACHE = os.environ.get("FWGYM_JIT_CACHE", os.path.join(_HERE, "_jit"))


def _cache_dir():
    """The package directory when it is writable (the built .so then travels with an in-tree checkout), otherwise a
    per-user cache directory (installed, read-only packages)."""
    for d in (CACHE, os.path.join(os.path.expanduser("~"), ".cache", "fwgym_jit")):
        try:
            os.makedirs(d, exist_ok=True)
            probe = os.path.join(d, ".w{}".format(os.getpid()))
            with open(probe, "w"):
                pass
            os.remove(probe)
            return d
        except OSError:
            continue
    return None


def _source_digest():
    h = hashlib.sha256()
    for f in sorted(os.listdir(CSRC)):
        p = os.path.join(CSRC, f)
        if os.path.isfile(p):
            with open(p, "rb") as fh:
                h.update(fh.read())
    with open(os.path.join(INCLUDE, "fwgym.h"), "rb") as fh:
        h.update(fh.read())
    return h.hexdigest()[:12]


def hipcc_path():
    p = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    return p if os.path.exists(p) else None


def _key(env_config, auto_reset, store_derived, base_lib, obs_log_rows):
    lib = base_lib or nat.load_library()
    words = specialize.spec_words(lib, env_config, auto_reset=auto_reset, store_derived=store_derived, obs_log_rows=obs_log_rows)
    return words, hashlib.sha256((_source_digest() + ",".join("%08x" % w for w in words)).encode()).hexdigest()[:16]


def is_cached(env_config, auto_reset=True, store_derived=True, base_lib=None, obs_log_rows=0):
    """True when specialised_library() would return at once (the library for this configuration is built already)."""
    cache = _cache_dir()
    if cache is None:
        return False
    _, key = _key(env_config, auto_reset, store_derived, base_lib, obs_log_rows)
    return os.path.exists(os.path.join(cache, "libfwgym_{}.so".format(key)))


def specialised_library(env_config, auto_reset=True, store_derived=True, base_lib=None, obs_log_rows=0):
    """Returns the path of a libfwgym build whose spec 0 is this configuration (building it when missing), or None."""
    words, key = _key(env_config, auto_reset, store_derived, base_lib, obs_log_rows)
    cache = _cache_dir()
    if cache is None:
        warnings.warn("no writable cache directory for specialised kernels: using the generic kernel")
        return None
    out = os.path.join(cache, "libfwgym_{}.so".format(key))
    if os.path.exists(out):
        return out
    hipcc = hipcc_path()
    if hipcc is None:
        warnings.warn("hipcc not found: using the generic kernel")
        return None
    # ONE build per key on a machine: the ranks of a multi-GPU job all ask for the same library at the same moment; the first
    # takes the lock and compiles (one to two minutes of hipcc), the others wait for it and find the file
    lock = None
    try:
        import fcntl
        lock = open(os.path.join(cache, "libfwgym_{}.lock".format(key)), "w")
        fcntl.flock(lock, fcntl.LOCK_EX)
    except (ImportError, OSError):
        lock = None
    try:
        if os.path.exists(out):
            return out
        return _compile(hipcc, cache, key, words, out)
    finally:
        if lock is not None:
            try:
                import fcntl
                fcntl.flock(lock, fcntl.LOCK_UN)
                lock.close()
            except OSError:
                pass


def _compile(hipcc, cache, key, words, out):
    # per-process file names, published with os.replace: a half-written file is never read
    inc = os.path.join(cache, "specs_{}.{}.inc".format(key, os.getpid()))
    # The configuration is frozen TWICE (specs 0 and 1, identical; the handle picks 0).  k_step2 sits at the register limit
    # (255 VGPRs, ~40 scalar registers spilled to vector lanes); compiled as the only instance of the translation unit the
    # allocator ran out of lanes and put nine of those spills into scratch memory -- 36 bytes per lane, ~100 scratch instructions
    # on the episode-end paths: 14.5 instead of 12.8 us per step in the steady state of a long run (C3, same-box A/B,
    # tools/devlib.py) --, with a second instance beside it it does not (as in the product library with its ten presets).
    # Costs the compile time of the second instance (~50 s, once per configuration).
    lines = ["// GENERATED by gym_fixed_wing/jit.py"]
    for k in (0, 1):
        lines.append("static constexpr SpecWords kSpecWords{} = {{{{".format(k))
        for j in range(0, len(words), 8):
            lines.append("    " + ", ".join("0x{:08x}u".format(w) for w in words[j:j + 8]) + ",")
        lines.append("}};")
    lines += ["#define FWG_SPEC_LIST(X) X(0) X(1)"]
    with open(inc, "w") as f:
        f.write("\n".join(lines) + "\n")
    tmp = out + ".tmp{}".format(os.getpid())
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-ffp-contract=fast", "-fno-slp-vectorize",
           "-Rpass-analysis=kernel-resource-usage",
           "-I" + INCLUDE, "-I" + CSRC, '-DFWG_SPECS_FILE="{}"'.format(inc), "-o", tmp, os.path.join(CSRC, "fwgym.hip")]
    try:
        r = subprocess.run(cmd, check=True, capture_output=True, text=True)
        # (the register report of the build: a specialised step kernel with a scratch frame or spilled vector registers is a
        # slower kernel -- say so instead of letting a profile find out; tests/test_kernel_resources.py gates the presets)
        try:
            bad = [(k, v.get("ScratchSize [bytes/lane]"), v.get("VGPRs Spill")) for k, v in specialize.parse_resource_remarks(r.stderr).items()
                   if ("k_step2<" in k or "k_rollout<" in k) and ", -1" not in k
                   and (v.get("ScratchSize [bytes/lane]", 0) or v.get("VGPRs Spill", 0))]
            if bad:
                warnings.warn("run-time specialised kernels with a scratch frame / spilled registers: {}".format(bad[:4]), RuntimeWarning)
        except Exception:
            pass
        os.replace(tmp, out)
    except (subprocess.CalledProcessError, OSError) as e:
        warnings.warn("kernel specialisation failed ({}): using the generic kernel".format(getattr(e, "stderr", e)))
        return None
    finally:
        for f in (inc, tmp):
            try:
                os.remove(f)
            except OSError:
                pass
    return out


def prebuild(config, config_kw=None, sim_config_kw=None, derived_views=True, obs_log_rows=None, auto_reset=True):
    """Compiles (or finds) the specialised library FixedWingVecEnv(config, ..., specialize=True) would ask for, without
    creating an env -- hipcc cross-compiles without a GPU, and the cached .so travels with an in-tree checkout."""
    from .config import EnvConfig
    from . import presets
    ec = EnvConfig(config, None, None, config_kw, sim_config_kw)
    if obs_log_rows is None:
        ob = ec.cfg["observation"]
        noise = ob.get("noise", None)
        noisy = noise is not None and (noise.get("var", 0) != 0 or noise.get("mean", 0) != 0)
        applies = int(ob.get("length", 1)) > 1 and not noisy and len(ob["states"]) % 4 == 0
        obs_log_rows = presets.OBS_LOG_ROWS if applies else 0
    return specialised_library(ec, auto_reset=auto_reset, store_derived=derived_views, obs_log_rows=int(obs_log_rows))
