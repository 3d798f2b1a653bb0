"""Build recipe of the native parts (no torch, no cmake: plain hipcc / g++ / make).

  phi_amd/libphi_amd.so   HIP kernels for gfx950 + the C ABI of include/phi_amd.h
  phi_amd/libphi_host.so  host-side graph/reads readers (C ABI of include/phi_host.h)
  phi_amd/PHI             the command-line driver (same CLI as the reference's ./PHI)

The CPU checker under oracle/ is test infrastructure and is NOT built from here: __graft_entry__.build()
and tests/conftest.py run its Makefile.

hipcc cross-compiles for gfx950 without a GPU present.
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "phi_amd", "csrc")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
ARCH = "gfx950"

HIP_SOURCES = ["sketch.hip", "sketch_pooled.hip", "table.hip", "anchors.hip", "contexts.hip", "dp.hip", "dp_events.hip", "phi_abi.hip", "phi_solve.hip", "solve_dev.hip", "phi_comm.hip", "phi_ipc.hip", "reads_text.hip", "walk_text.hip"]
HIP_HEADERS = ["phi_dev.h", "phi_kernels.h", "phi_ctx.h", "sketch.hip", "sketch_phases.inc", os.path.join("..", "..", "include", "phi_amd.h")]


def _run(cmd, cwd=None):
    print("+", " ".join(cmd), flush=True)
    subprocess.check_call(cmd, cwd=cwd)


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


# sketch_pooled.hip: the read kernel's instances that loop over a wave's chunks, at exactly 80 VGPRs (six waves per SIMD);
# hoisting constants and addresses out of that loop (machine LICM) keeps them in registers through every turn and pushes
# others into scratch
EXTRA_FLAGS = {"sketch_pooled.hip": ["-mllvm", "-disable-machine-licm"]}


def build_device(force=False):
    objdir = os.path.join(ROOT, "build", "obj")
    os.makedirs(objdir, exist_ok=True)
    hdrs = [os.path.join(CSRC, h) for h in HIP_HEADERS]
    objs = []
    for src in HIP_SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(objdir, src.replace(".hip", ".o"))
        if force or _stale(o, [s] + hdrs):
            _run([HIPCC, f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"]
                 + EXTRA_FLAGS.get(src, []) + ["-c", s, "-o", o])
        objs.append(o)
    lib = os.path.join(ROOT, "phi_amd", "libphi_amd.so")
    if force or _stale(lib, objs):
        tmp = f"{lib}.{os.getpid()}.tmp"                    # other ranks wait for `lib` to appear: never half-written
        _run([HIPCC, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", tmp] + objs + ["-ldl"])
        os.replace(tmp, lib)
    return lib


def build_host(force=False):
    hostdir = os.path.join(CSRC, "host")
    if not os.path.isdir(hostdir):
        return None
    _run(["make", "-C", hostdir] + (["-B"] if force else []))
    return os.path.join(ROOT, "phi_amd", "libphi_host.so")


def build_all(force=False):
    build_device(force)
    build_host(force)


if __name__ == "__main__":
    build_all(force="--force" in sys.argv)
