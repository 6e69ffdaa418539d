"""Build libqpnet_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU).

Every .hip file is compiled to an object of its own (in parallel, cached by modification time under csrc/_obj/) and the objects are
linked into qpnet_amd/libqpnet_hip.so.  `testing=True` builds the same sources with -DQPN_TESTING into
qpnet_amd/libqpnet_hip_testing.so: the only difference is the two fault-injection hooks the give-up tests need
(QPN_TEST_STACK_GIVES_UP, QPN_TEST_PIPE_GIVES_UP); the product library does not contain them.  `extra=[...]` + `name=` builds a dev
variant (stamps, removal experiments) into build_variants/libqpnet_<name>.so -- select it with QPN_LIB=<path>."""
import glob
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.dirname(HERE)
ROOT = os.path.dirname(PKG)
OUT = os.path.join(PKG, "libqpnet_hip.so")
OUT_TESTING = os.path.join(PKG, "libqpnet_hip_testing.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=off",
         "-fno-fast-math", "-Wall", "-Wno-unused-function"]


def sources():
    return sorted(glob.glob(os.path.join(HERE, "*.hip")))


def _headers():
    return glob.glob(os.path.join(HERE, "*.h")) + glob.glob(os.path.join(ROOT, "include", "*.h"))


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def needs_build(out=OUT):
    return _stale(out, sources() + _headers() + [os.path.abspath(__file__)])


def _compile(src, obj, flags, verbose):
    cmd = [HIPCC] + FLAGS + flags + ["-c", src, "-o", obj]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)


def build(force=False, verbose=False, testing=False, extra=(), name=None, jobs=None):
    """-> path of the shared library (built only when a source or header is newer)."""
    extra = list(extra)
    if name:
        out = os.path.join(ROOT, "build_variants", "libqpnet_%s.so" % name)
        tag = "v_" + name
    elif testing:
        out, tag = OUT_TESTING, "testing"
        extra = extra + ["-DQPN_TESTING"]
    else:
        out, tag = OUT, "product"
    if not force and not needs_build(out):
        return out
    objdir = os.path.join(HERE, "_obj", tag)
    os.makedirs(objdir, exist_ok=True)
    os.makedirs(os.path.dirname(out), exist_ok=True)
    hdrs = _headers() + [os.path.abspath(__file__)]
    todo, objs = [], []
    for src in sources():
        obj = os.path.join(objdir, os.path.basename(src)[:-4] + ".o")
        objs.append(obj)
        if force or _stale(obj, [src] + hdrs):
            todo.append((src, obj))
    jobs = jobs or min(len(todo) or 1, max(1, (os.cpu_count() or 2) - 1), 8)
    with ThreadPoolExecutor(jobs) as ex:
        list(ex.map(lambda so: _compile(so[0], so[1], extra, verbose), todo))
    cmd = [HIPCC, "-shared", "-fPIC", "--offload-arch=gfx950"] + objs + ["-o", out]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return out


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    force = "--force" in sys.argv
    if "--variant" in sys.argv:          # python build.py --variant <name> -DFLAG ...
        i = sys.argv.index("--variant")
        print(build(force=True, verbose=True, name=sys.argv[i + 1], extra=[a for a in sys.argv[i + 2:] if a.startswith("-")]))
    else:
        print(build(force=force, verbose=True))
        print(build(force=force, verbose=True, testing=True))
