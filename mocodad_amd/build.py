"""Builds libmocodad_hip.so for gfx950 from mocodad_amd/csrc: mcd_api.hip (C ABI, packers, dispatch, the runtime-shape kernels)
plus mcd_inst.hip once per unit of kernel instantiations (csrc/mcd_instances.hpp), compiled in parallel and linked with hipcc.

    python -m mocodad_amd.build                       # the shipped library (mocodad_amd/libmocodad_hip.so)
    python -m mocodad_amd.build --profile             # + -DMCD_PROFILE -> libmocodad_hip_prof.so (tools/stage_profile.py)
    python -m mocodad_amd.build --fast-t 3 -o /tmp/x.so -D MCD_STASH=0    # developer build: one trajectory kernel only

Objects are cached under csrc/_obj/<tag>/ (git-ignored) and rebuilt when a source, the public header or the flag set is newer /
different; the library is relinked when any object changed."""
import argparse
import concurrent.futures as cf
import hashlib
import os
import re
import subprocess
import sys
import time
from typing import Iterable, List, Optional

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
HEADER = os.path.join(ROOT, "include", "mocodad_hip.h")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
DEFAULT_OUT = os.path.join(HERE, "libmocodad_hip.so")
BASE_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden", "-Wno-pass-failed"]


def n_units() -> int:
    txt = open(os.path.join(CSRC, "mcd_instances.hpp")).read()
    return int(re.search(r"#define\s+MCD_INST_UNITS\s+(\d+)", txt).group(1))


def unit_flags() -> dict:
    """Per-unit extra compile flags: `#define MCD_UNIT_FLAGS_<n> "..."` lines of csrc/mcd_instances.hpp."""
    txt = open(os.path.join(CSRC, "mcd_instances.hpp")).read()
    return {int(u): f.split() for u, f in re.findall(r'#define\s+MCD_UNIT_FLAGS_(\d+)\s+"([^"]*)"', txt)}


def shipped_shape(t: int):
    """(unit, NB, MINW) of the shipped score_kernel<T, ...> (the non-layer-test row of MCD_SCORE_INSTANCES), or None."""
    txt = open(os.path.join(CSRC, "mcd_instances.hpp")).read()
    for u, tt, nb, minw in re.findall(r"X\((\d+),\s*(\d+),\s*(\d+),\s*(\d+),\s*false\)", txt.split("MCD_SCORE_INSTANCES(X)")[1].split("MCD_COND_FAST_INSTANCES")[0]):
        if int(tt) == t:
            return int(u), int(nb), int(minw)
    return None


_MLLVM_OK = {}


def usable_flags(flags: List[str]) -> List[str]:
    """`-mllvm <opt>` pairs name INTERNAL LLVM options (the scheduler strategy of units 3 / 23): probe each pair once on an empty
    device file and drop it, with a warning, if this hipcc does not know it -- the library then builds with the default strategy
    (same results, the tuned kernels a few per cent slower) instead of not building at all."""
    out, i = [], 0
    while i < len(flags):
        if flags[i] == "-mllvm" and i + 1 < len(flags):
            opt = flags[i + 1]
            if opt not in _MLLVM_OK:
                p = subprocess.run([HIPCC, "--offload-arch=gfx950", "--cuda-device-only", "-x", "hip", "-c", "/dev/null", "-o", os.devnull,
                                    "-mllvm", opt], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
                _MLLVM_OK[opt] = p.returncode == 0
                if not _MLLVM_OK[opt]:
                    print(f"warning: {HIPCC} rejects '-mllvm {opt}': building without it", file=sys.stderr)
            if _MLLVM_OK[opt]:
                out += ["-mllvm", opt]
            i += 2
        else:
            out.append(flags[i]); i += 1
    return out


def toolchain_id() -> str:
    """One line naming the compiler that builds the library (`hipcc --version`: HIP version + clang version lines)."""
    p = subprocess.run([HIPCC, "--version"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    keep = [l.strip() for l in p.stdout.splitlines() if l.startswith(("HIP version", "AMD clang version", "clang version"))]
    return " | ".join(keep) or p.stdout.strip()[:200]


def buildinfo_path(out: str) -> str:
    return out + ".buildinfo"


def _compile(cmd: List[str], obj: str) -> float:
    """Compile into a private temporary and rename it into place: concurrent builds (several ranks calling build() at once)
    never see, or link, a half-written object."""
    tmp = obj + ".tmp%d" % os.getpid()
    try:
        dt = _run(cmd + ["-o", tmp])
        os.replace(tmp, obj)
    finally:
        if os.path.exists(tmp):
            os.remove(tmp)
    return dt


def sources() -> List[str]:
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".hpp"))) + [HEADER]


def _run(cmd: List[str]) -> float:
    t0 = time.perf_counter()
    p = subprocess.run(cmd, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if p.returncode != 0:
        raise RuntimeError("command failed: " + " ".join(cmd) + "\n" + p.stdout[-8000:])
    if p.stdout.strip():
        print(p.stdout, file=sys.stderr)
    return time.perf_counter() - t0


def build_library(out: str = DEFAULT_OUT, defines: Iterable[str] = (), extra_flags: Iterable[str] = (), force: bool = False,
                  jobs: Optional[int] = None, obj_dir: Optional[str] = None, verbose: bool = True) -> bool:
    """Compile + link; returns True when anything was rebuilt.  `defines`: "NAME" or "NAME=VALUE" strings."""
    defines = list(defines)
    flags = BASE_FLAGS + ["-D" + d for d in defines] + list(extra_flags)
    tag = hashlib.sha1(" ".join(flags).encode()).hexdigest()[:10]
    obj_dir = os.path.join(obj_dir, tag) if obj_dir else os.path.join(CSRC, "_obj", tag)      # (the flag set is part of the path)
    os.makedirs(obj_dir, exist_ok=True)
    # an object depends on its own .hip, every header of csrc and the public header (not on the other .hip)
    hdr_m = max(os.path.getmtime(p) for p in sources() if not p.endswith(".hip"))
    newest = lambda src: max(hdr_m, os.path.getmtime(src))
    fast = any(d.split("=")[0] == "MCD_FAST_T" for d in defines)
    units = [1] if fast else list(range(1, n_units() + 1))
    jobs_l = [("mcd_api.o", os.path.join(CSRC, "mcd_api.hip"), [])]
    uf = {} if fast else {u: usable_flags(f) for u, f in unit_flags().items()}      # (developer builds: command line / main())
    jobs_l += [(f"mcd_inst_{u}.o", os.path.join(CSRC, "mcd_inst.hip"), [f"-DMCD_INST_UNIT={u}"] + uf.get(u, [])) for u in units]
    todo = [(o, s, f) for o, s, f in jobs_l
            if force or not os.path.exists(os.path.join(obj_dir, o)) or os.path.getmtime(os.path.join(obj_dir, o)) < newest(s)]
    t0 = time.perf_counter()
    if todo:
        workers = jobs or min(len(todo), os.cpu_count() or 4)
        if verbose:
            print(f"+ {HIPCC} {' '.join(flags)} -c  x {len(todo)} translation units, {workers} at a time", flush=True)
        with cf.ThreadPoolExecutor(max_workers=workers) as ex:
            # -cuid: clang derives a compilation-unit id from the command line INCLUDING the output path and bakes it into the object
            # (names of the fat-binary handles): with the private temporaries above every build had its own library hash, and the
            # profile manifests (tools/profile_set.sh, bench.py's PMC check) could never match a rebuilt library.  A fixed id per
            # (object, flag set) makes the library a function of the sources and flags alone.
            cuid = lambda o: "-cuid=mcd_" + tag + "_" + re.sub(r"[^A-Za-z0-9]", "_", o)
            futs = {ex.submit(_compile, [HIPCC] + flags + f + [cuid(o), "-c", s], os.path.join(obj_dir, o)): o for o, s, f in todo}
            for fu in cf.as_completed(futs):
                dt = fu.result()
                if verbose:
                    print(f"  {futs[fu]:16s} {dt:6.1f} s", flush=True)
    objs = [os.path.join(obj_dir, o) for o, _, _ in jobs_l]
    relink = bool(todo) or not os.path.exists(out) or os.path.getmtime(out) < max(os.path.getmtime(o) for o in objs)
    if relink:
        tmp = out + ".tmp%d" % os.getpid()
        _run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", tmp] + objs)
        os.replace(tmp, out)        # (atomic: a concurrent reader never maps a half-written library)
        # what built it, next to it (git-ignored like the library, travels with it): tests/test_build_gpu.py compares a fresh
        # build byte for byte with the shipped file only when the box's toolchain is this one and no per-unit flag was dropped
        want = unit_flags()
        dropped = sorted({" ".join(f) for u, f in want.items() if not fast and uf.get(u, []) != f})
        import json
        with open(buildinfo_path(out) + ".tmp%d" % os.getpid(), "w") as fh:
            json.dump({"toolchain": toolchain_id(), "flags": flags, "dropped_unit_flags": dropped}, fh)
        os.replace(buildinfo_path(out) + ".tmp%d" % os.getpid(), buildinfo_path(out))
        if verbose:
            print(f"+ linked {os.path.relpath(out, ROOT)}  ({time.perf_counter() - t0:.1f} s)", flush=True)
    return relink


def main() -> None:
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("-o", "--out", default=None)
    ap.add_argument("-D", dest="defines", action="append", default=[], help="extra macro (NAME or NAME=VALUE)")
    ap.add_argument("--profile", action="store_true", help="-DMCD_PROFILE build (default output: libmocodad_hip_prof.so)")
    ap.add_argument("--fast-t", type=int, default=None, help="developer build holding only score_kernel<T, ...> (+ its encoders)")
    ap.add_argument("-X", dest="xflags", action="append", default=[], help="extra raw compiler flag (repeatable), e.g. -X=-mllvm -X=-amdgpu-sched-strategy=max-ilp")
    ap.add_argument("--force", action="store_true")
    ap.add_argument("-j", "--jobs", type=int, default=None)
    a = ap.parse_args()
    defs = list(a.defines)
    if a.profile:
        defs.append("MCD_PROFILE")
    fast_x = []
    if a.fast_t is not None:
        defs.append(f"MCD_FAST_T={a.fast_t}")
        # the shipped shape of that frame count (chains per workgroup, waves per SIMD, its unit's flags) unless given
        sh = shipped_shape(a.fast_t)
        given = {d.split("=")[0] for d in defs}
        if sh:
            unit, nb, minw = sh
            for fl in unit_flags().get(unit, []):
                if fl.startswith("-D"):
                    if fl[2:].split("=")[0] not in given:
                        defs.append(fl[2:])
                elif not a.xflags:            # the unit's other flags (e.g. its scheduler strategy) unless -X gives a set
                    fast_x.append(fl)
            if "MCD_FAST_NB" not in given:
                defs.append(f"MCD_FAST_NB={nb}")
            if "MCD_FAST_MINW" not in given and not any(d.split("=")[0] == "MCD_NWAVES" for d in a.defines):
                defs.append(f"MCD_FAST_MINW={minw}")
    out = a.out or (os.path.join(HERE, "libmocodad_hip_prof.so") if a.profile else DEFAULT_OUT)
    build_library(out, defs, extra_flags=usable_flags(list(a.xflags) + (fast_x if a.fast_t is not None else [])), force=a.force, jobs=a.jobs)


if __name__ == "__main__":
    main()
