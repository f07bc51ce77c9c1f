"""Build libnnr.so (gfx950) in-tree:  python nope-nerf_amd/csrc/build.py

hipcc cross-compiles without a GPU.  Objects go to csrc/build/, the library to nope-nerf_amd/nnr/libnnr.so
(git-ignored, but shipped to the GPU box by gpurun)."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
OUT_DIR = os.path.join(HERE, "build")
LIB = os.path.join(os.path.dirname(HERE), "nnr", "libnnr.so")
SOURCES = ["nnr_api.cpp", "nnr_pack.hip", "nnr_mlp_fwd.hip", "nnr_mlp_dgrad.hip", "nnr_mlp_fwd_bf16.hip", "nnr_mlp_dgrad_bf16.hip", "nnr_wgrad.hip", "nnr_wgrad_bf16.hip", "nnr_composite.hip", "nnr_camera.hip", "nnr_pointcloud.hip", "nnr_aux.hip", "nnr_randperm.hip"]
HEADERS = ["nnr_layout.h", "nnr_device.h", "nnr_kernels.h", os.path.join("..", "..", "include", "nnr.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics", "-I" + HERE,
         "-I" + os.path.join(HERE, "..", "..", "include"), "-x", "hip"]


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build_variant(name, defines):
    """Profiling-only library with compile-time ablations (e.g. -DNNR_ABLATE_NO_STASH); results are NOT valid.
    Written to nnr/libnnr_<name>.so and selected with NNR_LIB=<path> (see nnr/lib.py)."""
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    out = os.path.join(os.path.dirname(HERE), "nnr", "libnnr_%s.so" % name)
    cmd = [hipcc] + [f for f in FLAGS if f not in ("-x", "hip")] + ["-D" + d for d in defines] + ["-shared", "-x", "hip"] + \
          [os.path.join(HERE, s) for s in SOURCES] + ["-o", out]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(r.stderr)
    return out


def build(force=False, verbose=False):
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    os.makedirs(OUT_DIR, exist_ok=True)
    hdrs = [os.path.join(HERE, h) for h in HEADERS]
    jobs = []
    for src in SOURCES:
        obj = os.path.join(OUT_DIR, os.path.splitext(src)[0] + ".o")
        if force or _stale(obj, [os.path.join(HERE, src)] + hdrs):
            jobs.append([hipcc] + FLAGS + ["-c", os.path.join(HERE, src), "-o", obj])

    def run(cmd):
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed: %s\n%s" % (" ".join(cmd), r.stderr))
        if verbose and r.stderr:
            print(r.stderr, file=sys.stderr)

    with ThreadPoolExecutor(max_workers=min(6, max(1, len(jobs)))) as ex:
        list(ex.map(run, jobs))
    objs = [os.path.join(OUT_DIR, os.path.splitext(s)[0] + ".o") for s in SOURCES]
    if force or jobs or _stale(LIB, objs):
        run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs)
    return LIB


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--variant":      # build.py --variant nostash NNR_ABLATE_NO_STASH [...]
        print(build_variant(sys.argv[2], sys.argv[3:]))
    else:
        print(build(force="--force" in sys.argv, verbose=True))
