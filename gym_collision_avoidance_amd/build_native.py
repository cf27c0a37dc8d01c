"""Build libcagpu.so (the HIP hot path) in-tree for gfx950.  hipcc cross-compiles without a GPU."""
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
SRC = os.path.join(HERE, "csrc", "cagpu.hip")
OUT = os.path.join(HERE, "libcagpu.so")
# -ffp-contract=off: a fused multiply-add inside `dx*dx + dy*dy <= r*r` would change discrete events
# (collision / at-goal masks) relative to the reference; see DESIGN.md "Numerics".
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-Wno-pass-failed", "-mllvm", "-disable-machine-licm",
         "-fhip-fp32-correctly-rounded-divide-sqrt", "-fPIC", "-shared",
         "-I" + os.path.join(REPO, "include")]


def needs_build():
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    deps = [SRC, os.path.join(HERE, "csrc", "cagpu_grouplp.inc"), os.path.join(HERE, "csrc", "cagpu_scan.inc"),
            os.path.join(HERE, "csrc", "cagpu_ga3c.inc"), os.path.join(HERE, "csrc", "cagpu_gen.inc"), os.path.join(HERE, "csrc", "cagpu_pipe.inc"), os.path.join(HERE, "csrc", "cagpu_big.inc"),
            os.path.join(REPO, "include", "cagpu.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False, variant=None):
    """variant None: the product library libcagpu.so.  variant "knobs" / "ablate": an EXPERIMENT build next to it
    (libcagpu_knobs.so / libcagpu_ablate.so, -DCAGPU_KNOBS / -DCAGPU_ABLATE: geometry overrides from CAGPU_* environment
    variables, in-kernel phase timers), loaded only by scratch/ scripts through CAGPU_LIB; never the product file."""
    if variant is None and not force and not needs_build():
        return OUT
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    out = OUT if variant is None else os.path.join(HERE, "libcagpu_%s.so" % variant)
    # "ablate_fast" / "knobs_fast": the same with only the N = 10 unstaged instantiations compiled (quick iterations)
    # "exp<mask>_fast": compile-time experiment switches, -DCAGPU_EXP=<mask> (see EXP() in cagpu.hip)
    # and "dKEY=VAL" -> -DCAGPU_KEY=VAL; tokens are separated by "_", or by "," when a KEY itself contains "_"
    # (e.g. "dPIPE_WT=0,fast" -> -DCAGPU_PIPE_WT=0 -DCAGPU_FAST)
    # "fFLAG" tokens pass an -mllvm option through (compiler experiments), e.g. famdgpu-enable-max-ilp-scheduling-strategy
    extra = [] if variant is None else [("-DCAGPU_EXP=%s" % v[3:]) if v.startswith("exp") else
                                        ("-mllvm=-%s" % v[1:]) if (v.startswith("f") and v not in ("fast",)) else
                                        ("-DCAGPU_%s" % v[1:]) if (v.startswith("d") and "=" in v) else
                                        "-DCAGPU_%s" % v.upper() for v in
                                        (variant.split(",") if "," in variant or "=" in variant else variant.split("_"))]
    extra = [y for x in extra for y in (["-mllvm", x[len("-mllvm="):]] if x.startswith("-mllvm=") else [x])]
    cmd = [hipcc] + FLAGS + extra + [SRC, "-o", out]
    if variant is not None:
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
        return out
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    write_build_info()
    return OUT


INFO = os.path.join(HERE, "_build_info.json")


def source_digest():
    """sha256 over the kernel sources + the header, in a fixed order: names the source tree a library was built from"""
    import hashlib
    h = hashlib.sha256()
    for f in sorted(os.listdir(os.path.join(HERE, "csrc"))) + [os.path.join(REPO, "include", "cagpu.h")]:
        path = f if os.path.isabs(f) else os.path.join(HERE, "csrc", f)
        h.update(os.path.basename(path).encode() + b"\0" + open(path, "rb").read())
    return h.hexdigest()


def file_sha256(path):
    import hashlib
    h = hashlib.sha256()
    with open(path, "rb") as f:
        for chunk in iter(lambda: f.read(1 << 20), b""):
            h.update(chunk)
    return h.hexdigest()


def write_build_info():
    """_build_info.json beside the product library (git-ignored, travels with it to the GPU box, where there is no .git):
    the commit the tree stood at when the library was built, whether the tree was dirty, the digest of the kernel sources
    and of the library itself.  bench.py stamps every line with it, profiles/make_r0N.py refuses to file artefacts that
    carry different library hashes (a profile cannot go stale unnoticed: VERDICT r05 weak-3 / weak-13)."""
    import json
    def git(*a):
        try:
            return subprocess.run(["git", "-C", REPO] + list(a), capture_output=True, text=True, timeout=20).stdout.strip()
        except Exception:  # noqa: BLE001
            return ""
    info = {"git_sha": git("rev-parse", "HEAD"), "git_dirty": bool(git("status", "--porcelain", "--", "gym_collision_avoidance_amd/csrc", "include")),
            "source_sha256": source_digest(), "lib_sha256": file_sha256(OUT), "flags": " ".join(FLAGS)}
    json.dump(info, open(INFO, "w"), indent=1)
    return info


def build_info():
    """what write_build_info() recorded, checked against the library that is actually there -> dict (lib_sha256 is always
    the hash of the file on disk; `stale` says the record was written for another file)"""
    import json
    info = {}
    if os.path.exists(INFO):
        try:
            info = json.load(open(INFO))
        except Exception:  # noqa: BLE001
            info = {}
    have = file_sha256(OUT) if os.path.exists(OUT) else None
    info["stale"] = bool(info.get("lib_sha256") and info["lib_sha256"] != have)
    info["lib_sha256"] = have
    return info


if __name__ == "__main__":
    import sys
    build(force=True, verbose=True, variant=sys.argv[1] if len(sys.argv) > 1 else None)
