"""Builds libkmerlr_hip.so (gfx950) in-tree with hipcc.  No JIT cache: the .so stays under
kmerlr_amd/_lib/ so it travels with the source tree to the GPU box."""
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "_build")
LIBDIR = os.path.join(HERE, "_lib")
LIB = os.path.join(LIBDIR, "libkmerlr_hip.so")

SOURCES = ["klr_ctx.hip", "klr_mat.hip", "klr_sweep.hip", "klr_select.hip", "klr_solve.hip",
           "klr_synth.cpp", "klr_driver.cpp", "klr_ingest.hip", "klr_kmers.cpp", "klr_shm.hip"]
HEADERS = [os.path.join(CSRC, "klr_internal.h"), os.path.join(HERE, "..", "include", "kmerlr.h"),
           os.path.join(HERE, "..", "include", "kmerlr_driver.h"), os.path.join(CSRC, "klr_kmers.h")]

FLAGS = ["-O3", "--offload-arch=gfx950", "-fPIC", "-std=c++17", "-ffp-contract=off", "-Wall",
         "-Wno-unused-result", "-fno-gpu-rdc"]
# KLR_ABLATIONS=1 python -m kmerlr_amd.build --force: a development build that also holds the A/B variants and the
# timing-only ablations of the sweep kernels (tools/ab_sweep.py, tools/ab_formats.py); never the shipped library
if os.environ.get("KLR_ABLATIONS"):
    FLAGS.append("-DKLR_ABLATIONS")
if os.environ.get("KLR_BUILD_DEFINES"):                     # development: extra -D switches, e.g. KLR_BUILD_DEFINES="KLR_GRAM_RB=4"
    FLAGS += ["-D" + d for d in os.environ["KLR_BUILD_DEFINES"].split()]


def hipcc():
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found; libkmerlr_hip.so cannot be built")


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.exists(d) and os.path.getmtime(d) > t for d in deps)


def kernel_hash():
    """Hash of the sources that decide what the roofline kernel does and reads (the sweep, and the ingest that lays the
    stream out): klr_version() reports it, profiles/ record it, bench.py refuses PMC numbers of another build."""
    import hashlib
    h = hashlib.sha256()
    for f in ("klr_sweep.hip", "klr_ingest.hip"):
        h.update(open(os.path.join(CSRC, f), "rb").read())
    return h.hexdigest()[:12]


def _write_hash_header():
    path = os.path.join(CSRC, "klr_kernel_hash.inc")
    text = '#define KLR_KERNEL_HASH "%s"\n' % kernel_hash()
    if not os.path.exists(path) or open(path).read() != text:
        open(path, "w").write(text)
    return path


def build(force=False, verbose=False):
    os.makedirs(OBJ, exist_ok=True)
    os.makedirs(LIBDIR, exist_ok=True)
    HEADERS.append(_write_hash_header()) if os.path.join(CSRC, "klr_kernel_hash.inc") not in HEADERS else _write_hash_header()
    cc = hipcc()
    srcs = [s for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    jobs = []
    for s in srcs:
        src = os.path.join(CSRC, s)
        obj = os.path.join(OBJ, os.path.splitext(s)[0] + ".o")
        if force or _stale(obj, [src] + HEADERS):
            cmd = [cc] + FLAGS + (["-x", "hip"] if s.endswith(".cpp") else []) + ["-c", src, "-o", obj]
            jobs.append(cmd)

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed:\n%s\n%s" % (" ".join(cmd), r.stderr[-4000:]))
        return r.stderr

    if jobs:
        with ThreadPoolExecutor(max_workers=min(6, len(jobs))) as ex:
            for w in ex.map(run, jobs):
                if verbose and w.strip():
                    print(w)
    objs = [os.path.join(OBJ, os.path.splitext(s)[0] + ".o") for s in srcs]
    if force or jobs or _stale(LIB, objs):
        run([cc, "-shared", "-fPIC", "--offload-arch=gfx950", "-o", LIB] + objs +
            ["-ldl", "-lpthread", "-Wl,-rpath,/opt/rocm/lib"])
    # command line (plain C++ against the C ABI)
    cli_src = os.path.join(HERE, "cli", "kmerlr_main.cpp")
    cli = os.path.join(LIBDIR, "kmerlr")
    if os.path.exists(cli_src) and (force or jobs or _stale(cli, [cli_src, LIB] + HEADERS)):
        run(["g++", "-O2", "-std=c++17", cli_src, "-o", cli, "-L" + LIBDIR, "-lkmerlr_hip",
             "-Wl,-rpath,$ORIGIN", "-Wl,-rpath,/opt/rocm/lib"])
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
