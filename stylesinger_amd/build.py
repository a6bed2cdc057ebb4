"""Build libstylesinger_hip.so (gfx950) with hipcc — in-tree, incremental, parallel.

`python -m stylesinger_amd.build` or `stylesinger_amd.build.build()`.  hipcc cross-compiles without a
GPU; the .so lands next to this file so that it travels with the tree to the GPU box.
"""
import concurrent.futures as cf
import hashlib
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "_obj")
LIB = os.path.join(HERE, "libstylesinger_hip.so")
ARCH = "gfx950"
FLAGS = ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-Wno-unused-result"] + os.environ.get("SS_EXTRA_HIPCC_FLAGS", "").split()


def _hipcc():
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: libstylesinger_hip.so cannot be built")


def _deps_hash(src):
    h = hashlib.sha1()
    h.update(" ".join(FLAGS).encode())
    for f in [src] + sorted(
        os.path.join(d, x)
        for d in (CSRC, os.path.join(os.path.dirname(HERE), "include"))
        for x in os.listdir(d)
        if x.endswith(".h")
    ):
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


def _compile(src):
    obj = os.path.join(OBJ, os.path.basename(src) + ".o")
    stamp = obj + ".sha1"
    want = _deps_hash(src)
    if os.path.exists(obj) and os.path.exists(stamp) and open(stamp).read() == want:
        return obj, False
    cmd = [_hipcc()] + FLAGS + ["-c", src, "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("hipcc failed for %s:\n%s" % (src, r.stderr[-6000:]))
    with open(stamp, "w") as fh:
        fh.write(want)
    return obj, True


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def build_clean(out_dir, jobs=None):
    """From-clean build of every source into `out_dir` (no object cache): proves the tree compiles on this box as it is."""
    os.makedirs(out_dir, exist_ok=True)
    srcs = sources()

    def one(src):
        obj = os.path.join(out_dir, os.path.basename(src) + ".o")
        r = subprocess.run([_hipcc()] + FLAGS + ["-c", src, "-o", obj], capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed for %s:\n%s" % (src, r.stderr[-4000:]))
        return obj
    with cf.ThreadPoolExecutor(jobs or min(len(srcs), os.cpu_count() or 4)) as ex:
        objs = list(ex.map(one, srcs))
    lib = os.path.join(out_dir, "libstylesinger_hip.so")
    r = subprocess.run([_hipcc(), "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", lib] + objs, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("link failed:\n" + r.stderr[-4000:])
    return lib


def build(verbose=True, jobs=None):
    os.makedirs(OBJ, exist_ok=True)
    srcs = sources()
    jobs = jobs or min(len(srcs), os.cpu_count() or 4)
    with cf.ThreadPoolExecutor(jobs) as ex:
        res = list(ex.map(_compile, srcs))
    objs = [o for o, _ in res]
    rebuilt = [o for o, r in res if r]
    if rebuilt or not os.path.exists(LIB):
        if os.path.exists(LIB + ".ok"):
            os.remove(LIB + ".ok")
        cmd = [_hipcc(), "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", LIB] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n" + r.stderr[-6000:])
    # a shared library links with undefined symbols; make sure this one resolves when it is loaded (e.g. a kernel whose host launch
    # stub the compiler dropped shows up only here). In a CHILD process: loading it here, before torch, would bring the system HIP runtime
    # into this process ahead of the one torch ships, and a later torch.cuda in the same process then finds no device.
    if not os.path.exists(LIB + ".ok"):
        r = subprocess.run([sys.executable, "-c", "import ctypes, sys; ctypes.CDLL(sys.argv[1])", LIB], capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("libstylesinger_hip.so does not load:\n" + (r.stderr or r.stdout)[-4000:])
        open(LIB + ".ok", "w").close()
    if verbose:
        print("[stylesinger_amd.build] %d sources, %d recompiled -> %s" % (len(srcs), len(rebuilt), LIB))
    return LIB


if __name__ == "__main__":
    build()
    sys.exit(0)
