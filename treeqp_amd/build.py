"""Build the native pieces of treeqp_amd, in-tree.

* ``libtreeqp_amd.so``  -- the product: host C layer (gcc) + HIP device path for gfx950 (hipcc),
  one C-ABI shared library (include/treeqp_amd.h + the reference-compatible treeqp headers).
* ``oracle/liboracle.so`` -- the CPU oracle (test infrastructure only, never linked into the product).
* ``oracle/_ref/*``      -- only when /root/reference is present: the reference's own example
  drivers, compiled UNCHANGED from where they lie and linked against libtreeqp_amd.so (drop-in
  proof for the boundary; the binaries travel to the GPU box, the sources do not).

hipcc cross-compiles gfx950 without a GPU, so this runs on the CPU-only build container.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
CSRC = ROOT / "treeqp_amd" / "csrc"
LIBDIR = ROOT / "treeqp_amd" / "lib"
OBJDIR = ROOT / "build" / "obj"
INCLUDE = ROOT / "include"
REFERENCE = Path("/root/reference")

HOST_SOURCES = ["blasfeo_compat.c", "tree_topology.c", "host_utils.c", "qp_container.c", "tdunes_host.c"]
DEVICE_SOURCES = ["tdunes_device.hip"]

HOST_CFLAGS = ["-O2", "-g", "-fPIC", "-std=gnu99", "-Wall", "-Wno-unused-variable", "-Wno-unused-but-set-variable"]
# max-ilp: with one or two waves per SIMD there is no occupancy to protect; scheduling the latency-bound chains for ILP measured 1.3 % faster on C2
HIP_FLAGS = ["--offload-arch=gfx950", "-O3", "-mllvm", "-amdgpu-sched-strategy=" + __import__("os").environ.get("TQ_SCHED", "max-ilp"),      # (TQ_SCHED: experiment builds with another scheduling strategy; the default scheduler does not compile this file on ROCm 7.2)
              *(["-DTQ_HOSTPROF"] if __import__("os").environ.get("TQ_HOSTPROF") else []), *(["-DTQ_FINE_STAMPS"] if __import__("os").environ.get("TQ_FINE_STAMPS") else []), *__import__("os").environ.get("TQ_DEFS", "").split(), "-fPIC", "-std=c++17", "-Wall", "-Wno-unused-function"]


def _run(cmd, **kw):
    print("[build]", " ".join(str(c) for c in cmd), flush=True)
    subprocess.run([str(c) for c in cmd], check=True, **kw)


def _newer(target: Path, sources) -> bool:
    if not target.exists():
        return True
    t = target.stat().st_mtime
    return any(Path(s).stat().st_mtime > t for s in sources)


def hipcc_path() -> str:
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and Path(cand).exists():
            return cand
    raise RuntimeError("hipcc not found: the treeqp_amd device path cannot be built")


def product_library() -> Path:
    return LIBDIR / "libtreeqp_amd.so"


# The device path is ONE source file compiled once per part, all parts in parallel (csrc/device/tdunes_parts.hpp):
# name -> (-DTQ_PARTS mask, extra defines).  The persistent family is sliced by shape (its kernels are most of the compile time).
PERSIST_SLICES = 9
DEVICE_PARTS = {
    "host": ("TQP_HOST", []),
    "gpersist": ("TQP_GP", []),
    "wide": ("TQP_WIDE", []),
    "wide3": ("TQP_W3", []),
    "tiered": ("TQP_TIER", []),
    **{f"persist{k}": ("TQP_PERSIST", [f"-DTQ_PERSIST_NSLICES={PERSIST_SLICES}", f"-DTQ_PERSIST_SLICE={k}"]) for k in range(PERSIST_SLICES)},
    "shard": ("TQP_SHARD", ["-DTQ_LD_SCOPE=__HIP_MEMORY_SCOPE_SYSTEM"]),
    "shard_ag": ("TQP_SHARD_AG", []),
    "batch": ("TQP_BATCH", []),
}
JOBS = max(1, min(len(DEVICE_PARTS), (os.cpu_count() or 4)))


def _part_cmd(part: str, obj: Path, defs=()):
    mask, extra = DEVICE_PARTS[part]
    src = CSRC / "device" / DEVICE_SOURCES[0]
    return [hipcc_path(), *HIP_FLAGS, f"-DTQ_PARTS={mask}", *extra, *defs, f"-I{INCLUDE}", f"-I{CSRC / 'device'}", "-c", src, "-o", obj]


def _compile_parts(jobs):
    """jobs: list of (part, obj, defs); run the compilers in parallel, longest parts first"""
    import concurrent.futures as cf
    import time
    order = sorted(jobs, key=lambda j: (not j[0].startswith("persist"), j[0]))
    t0 = time.time()

    def one(job):
        part, obj, defs = job
        cmd = _part_cmd(part, obj, defs)
        print("[build]", " ".join(str(c) for c in cmd), flush=True)
        t1 = time.time()
        r = subprocess.run([str(c) for c in cmd], capture_output=True, text=True)
        if r.returncode != 0:
            sys.stderr.write(r.stdout + r.stderr)
            raise RuntimeError(f"hipcc failed for part {part}")
        if r.stderr.strip():
            sys.stderr.write(r.stderr)
        print(f"[build] part {part}: {time.time() - t1:.0f} s", flush=True)

    with cf.ThreadPoolExecutor(max_workers=JOBS) as ex:
        list(ex.map(one, order))
    print(f"[build] device parts: {time.time() - t0:.0f} s wall", flush=True)


def build_variant(name: str, defs, parts=None) -> Path:
    """Experiment build: the listed parts of the device code (default: all) with extra -D flags, linked with the product's other
    objects into treeqp_amd/lib_var/<name>/libtreeqp_amd.so (loaded when TREEQP_AMD_LIB points at it; never the default)."""
    build_product()
    out = ROOT / "treeqp_amd" / "lib_var" / name
    out.mkdir(parents=True, exist_ok=True)
    parts = list(DEVICE_PARTS) if not parts else parts
    for q in parts:
        if q not in DEVICE_PARTS:
            raise SystemExit(f"unknown part {q}; parts: {' '.join(DEVICE_PARTS)}")
    objs = [OBJDIR / (n + ".o") for n in HOST_SOURCES]
    _compile_parts([(q, out / f"device_{q}.o", list(defs)) for q in parts])
    objs += [(out if q in parts else OBJDIR) / f"device_{q}.o" for q in DEVICE_PARTS]
    lib = out / "libtreeqp_amd.so"
    _run([hipcc_path(), "-shared", "-fPIC", "--offload-arch=gfx950", *objs, "-o", lib, "-lm"])
    return lib


def build_product(force: bool = False) -> Path:
    LIBDIR.mkdir(parents=True, exist_ok=True)
    OBJDIR.mkdir(parents=True, exist_ok=True)
    headers = list(INCLUDE.rglob("*.h"))
    objs = []
    for name in HOST_SOURCES:
        src = CSRC / "host" / name
        obj = OBJDIR / (name + ".o")
        if force or _newer(obj, [src] + headers):
            _run(["gcc", *HOST_CFLAGS, f"-I{INCLUDE}", "-c", src, "-o", obj])
        objs.append(obj)
    hipcc = hipcc_path()
    dev_src = [CSRC / "device" / n for n in DEVICE_SOURCES] + list((CSRC / "device").glob("*.h")) + list((CSRC / "device").glob("*.hpp"))
    jobs = []
    for part in DEVICE_PARTS:
        obj = OBJDIR / f"device_{part}.o"
        if force or _newer(obj, dev_src + headers):
            jobs.append((part, obj, []))
        objs.append(obj)
    if jobs:
        _compile_parts(jobs)
    lib = product_library()
    if force or _newer(lib, objs):
        _run([hipcc, "-shared", "-fPIC", "--offload-arch=gfx950", *objs, "-o", lib, "-lm"])
    # the JSON front end (qp_in.json -> qp_out.json), a small C program on top of the library
    tool_src = CSRC / "host" / "solve_qp_json.c"
    tool = LIBDIR / "treeqp_solve_json"
    if force or _newer(tool, [tool_src, lib] + headers):
        _run(["gcc", "-O2", "-std=gnu99", "-Wall", f"-I{INCLUDE}", tool_src, "-o", tool,
              f"-L{LIBDIR}", "-ltreeqp_amd", "-Wl,-rpath,$ORIGIN", "-lm"])
    return lib


def oracle_library() -> Path:
    return ROOT / "oracle" / "liboracle.so"


def build_oracle(force: bool = False, native: bool = False) -> Path:
    src = ROOT / "oracle" / "tdunes_oracle.c"
    hdr = ROOT / "oracle" / "tdunes_oracle.h"
    lib = oracle_library()
    if force or _newer(lib, [src, hdr]):
        # portable flags: the .so built here travels to the GPU box (different host CPU)
        arch = ["-march=native"] if native else ["-mavx2", "-mfma"]
        _run(["gcc", "-O3", *arch, "-fopenmp", "-fPIC", "-std=gnu99", "-Wall", "-Wno-unused-function",
              "-shared", "-o", lib, src, "-lm"])
    return lib


REF_EXAMPLES = {
    # output name -> (source relative to the reference root, extra defines)
    "spring_mass_tdunes": ("examples/spring_mass_dual_newton_tree.c", ["-DNREP=20", "-DPRINT_LEVEL=1", "-DPROFILE=0"]),
    "thesis_example": ("examples/thesis_example.c", ["-DNREP=1", "-DPRINT_LEVEL=1", "-DPROFILE=0"]),
    # the reference's unit test with golden vectors (cmake: test_random_qp_DATA<i>): dense Q, S != 0, unconstrained;
    # its own asserts compare with xopt / uopt of random_qp_utils/data0<i>.c to 1e-12
    **{f"random_qp_data0{i}": ("examples/random_qp.c", [f"-DDATA={i}", "-DPRINT_LEVEL=0", "-DPROFILE=0"]) for i in range(6)},
}


def build_reference_dropins(force: bool = False):
    """Compile the reference's UNCHANGED example drivers against our headers + library."""
    if not REFERENCE.exists():
        return []
    outdir = ROOT / "oracle" / "_ref"
    outdir.mkdir(parents=True, exist_ok=True)
    lib = product_library()
    built = []
    for name, (rel, defs) in REF_EXAMPLES.items():
        src = REFERENCE / rel
        exe = outdir / name
        if not src.exists():
            continue
        if force or _newer(exe, [src, lib]):
            # -I<reference> is needed only for the driver's `#include "examples/.../data.c"`;
            # our include dir comes first so every treeqp/blasfeo header resolves to ours.
            _run(["gcc", "-O2", "-std=gnu99", *defs, f"-I{INCLUDE}", f"-I{REFERENCE}", src, "-o", exe,
                  f"-L{LIBDIR}", "-ltreeqp_amd", f"-Wl,-rpath,$ORIGIN/../../treeqp_amd/lib", "-lm"])
        built.append(exe)
    return built


def build_all(force: bool = False):
    lib = build_product(force)
    orc = build_oracle(force)
    refs = build_reference_dropins(force)
    return lib, orc, refs


if __name__ == "__main__":
    if "--variant" in sys.argv:          # python treeqp_amd/build.py --variant NAME [--parts persist0,shard] -DFOO -DBAR=1
        i = sys.argv.index("--variant")
        rest = sys.argv[i + 2:]
        parts = None
        if "--parts" in rest:
            k = rest.index("--parts")
            parts = rest[k + 1].split(",")
            rest = rest[:k] + rest[k + 2:]
        print("[build] variant:", build_variant(sys.argv[i + 1], rest, parts))
        sys.exit(0)
    out = build_all(force="--force" in sys.argv)
    print("[build] done:", *out[:2], *out[2])
