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
# (source, defines): the two fp32 MLP kernels are compiled one template instantiation per translation unit -- each is minutes of hipcc
# time (straight-line code of ~8 000 MFMAs), in one unit the forward alone took 8.5 minutes; the longest unit first
SOURCES = [("nnr_wgrad.hip", ()), ("nnr_wgrad.hip", ("NNR_WGRAD_F16_TU=1",)), ("nnr_mlp_fwd_f16.hip", ("NNR_FWD_D=256", "NNR_FWD_TRAIN=1")), ("nnr_mlp_dgrad_f16.hip", ("NNR_DGRAD_D=256",)),
           ("nnr_mlp_fwd_f16.hip", ("NNR_FWD_D=256", "NNR_FWD_TRAIN=0")),
           ("nnr_mlp_fwd_f16.hip", ("NNR_FWD_D=128", "NNR_FWD_TRAIN=1")), ("nnr_mlp_dgrad_f16.hip", ("NNR_DGRAD_D=128",)),
           ("nnr_mlp_fwd_f16.hip", ("NNR_FWD_D=128", "NNR_FWD_TRAIN=0")),
           ("nnr_mlp_fwd.hip", ("NNR_FWD_D=256", "NNR_FWD_TRAIN=1", "NNR_FWD_MODE=2")), ("nnr_mlp_dgrad.hip", ("NNR_DGRAD_D=256", "NNR_DGRAD_MODE=2")),
           ("nnr_mlp_fwd.hip", ("NNR_FWD_D=256", "NNR_FWD_TRAIN=0", "NNR_FWD_MODE=2")),
           ("nnr_mlp_fwd.hip", ("NNR_FWD_D=128", "NNR_FWD_TRAIN=1", "NNR_FWD_MODE=2")), ("nnr_mlp_dgrad.hip", ("NNR_DGRAD_D=128", "NNR_DGRAD_MODE=2")),
           ("nnr_mlp_fwd.hip", ("NNR_FWD_D=128", "NNR_FWD_TRAIN=0", "NNR_FWD_MODE=2")),
           ("nnr_mlp_fwd.hip", ("NNR_FWD_D=256", "NNR_FWD_TRAIN=1")), ("nnr_mlp_fwd.hip", ("NNR_FWD_D=256", "NNR_FWD_TRAIN=0")),
           ("nnr_mlp_dgrad.hip", ("NNR_DGRAD_D=256",)), ("nnr_mlp_fwd.hip", ("NNR_FWD_D=128", "NNR_FWD_TRAIN=1")),
           ("nnr_mlp_fwd.hip", ("NNR_FWD_D=128", "NNR_FWD_TRAIN=0")), ("nnr_mlp_dgrad.hip", ("NNR_DGRAD_D=128",)),
           ("nnr_mlp_fwd_bf16.hip", ()), ("nnr_mlp_dgrad_bf16.hip", ()), ("nnr_mlp_fwd.hip", ()), ("nnr_mlp_dgrad.hip", ()),
           ("nnr_api.cpp", ()), ("nnr_pack.hip", ()), ("nnr_wgrad_bf16.hip", ()), ("nnr_composite.hip", ()),
           ("nnr_camera.hip", ()), ("nnr_pointcloud.hip", ()), ("nnr_aux.hip", ()), ("nnr_randperm.hip", ()), ("nnr_optim.hip", ())]


def _obj_name(src, defines):
    tag = "".join("_" + d.split("=")[0].lower().replace("nnr_", "") + d.split("=")[1] for d in defines)
    return os.path.splitext(src)[0] + tag + ".o"
SPLIT2_ONLY = ["nnr_split2.h"]      # included by the two fp16-term kernels only: touching it does not rebuild the rest (minutes per unit)
HEADERS = ["nnr_layout.h", "nnr_device.h", "nnr_kernels.h", "nnr_mlp_bf16.h", "nnr_split.h", os.path.join("..", "..", "include", "nnr.h")]
# -pragma-unroll-threshold: the MLP kernels are straight-line code by construction (every `#pragma unroll` loop must unroll fully, or
# the register arrays they index fall back to scratch memory).  LLVM caps `#pragma unroll` at 16 K instructions per loop; one GEMM part
# of the fp32 input-gradient kernel sat right at that cap, and an unrelated clean-up pushed it over: the kernel compiled without a
# warning, kept passing every parity test and ran 8x slower (832 bytes of scratch per lane).  Hence the raised cap AND the check below.
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-mllvm", "-pragma-unroll-threshold=1048576",
         "-Rpass-analysis=kernel-resource-usage", "-I" + HERE, "-I" + os.path.join(HERE, "..", "..", "include"), "-x", "hip"]
# Scratch bytes per lane a hot kernel may use, by substring of the MANGLED name (the longest matching substring decides).  In the bf16
# training kernels a spill reload is not just a slow load: stores are always in flight there, hipcc waits for any load next to pending
# stores with vmcnt(0), and every reload drains the stash-store queue with the matrix pipe idle -- the TRAINING kernels may only keep a handful of
# prologue values in scratch, reloaded at pass start (where the pass waits for its inputs anyway), nothing inside a pass; the inference forward (no stores in flight) may spill its composite carry.
# (three-term kernels: only the D = 256 training forward keeps a few pass-start values in scratch, 32 bytes)
SCRATCH_LIMIT = {"18mlp_fwd_f16_kernelI": 0, "18mlp_fwd_f16_kernelILi256ELb0E": 48, "18mlp_fwd_f16_kernelILi128ELb0E": 48,      # (inference: prologue values and the composite carry, outside the MFMA streams)
                 "20mlp_dgrad_f16_kernelI": 0, "14mlp_fwd_kernelI": 0, "14mlp_fwd_kernelILi256ELb1ELi2E": 48, "16mlp_dgrad_kernelI": 0, "12wgrad_kernelI": 0, "14wgrad_b_kernelE": 0,
                 "19mlp_fwd_bf16_kernelI": 48, "19mlp_fwd_bf16_kernelILi256ELb1E": 16, "19mlp_fwd_bf16_kernelILi128ELb1E": 16,
                 "21mlp_dgrad_bf16_kernelI": 64, "20composite_fwd_kernelE": 0, "20composite_bwd_kernelE": 0}


def check_resources(remarks, what):
    """Parse hipcc's kernel-resource-usage remarks and fail the build when a hot kernel went to scratch memory."""
    import re
    name, bad = None, []
    for line in remarks.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            name = m.group(1)
        m = re.search(r"ScratchSize \[bytes/lane\]: (\d+)", line)
        if m and name:
            keys = [k for k in SCRATCH_LIMIT if ("3nnr" + k) in name]
            if keys:
                key = max(keys, key=len)
                if int(m.group(1)) > SCRATCH_LIMIT[key]:
                    bad.append("%s: %s bytes of scratch per lane (limit %d)" % (name, m.group(1), SCRATCH_LIMIT[key]))
    if bad:
        raise RuntimeError("%s: a hot kernel uses scratch memory -- a loop did not unroll or registers spilled:\n  " % what + "\n  ".join(bad))


WORKERS = max(2, min(12, (os.cpu_count() or 4)))


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build_variant(name, defines, only=None):
    """Profiling-only library nnr/libnnr_<name>.so with compile-time switches (-DNNR_ABLATE=<bits>, -DNNR_TIMELINE: nnr_device.h); results
    of such a build are NOT valid.  `only`: a substring of the source names to recompile (e.g. "_f16": the fp16-term kernels, minutes instead
    of the whole library); every other object comes from the main build.  Selected at run time with NNR_LIB=<path> (nnr/lib.py)."""
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    out = os.path.join(os.path.dirname(HERE), "nnr", "libnnr_%s.so" % name)
    tmp = os.path.join(OUT_DIR, "variant_" + name)
    os.makedirs(tmp, exist_ok=True)
    mine = [(src, d) for src, d in SOURCES if only is None or only in src]
    jobs = [[hipcc] + FLAGS + ["-D" + d for d in list(defines) + list(d0)] + ["-c", os.path.join(HERE, src), "-o", os.path.join(tmp, _obj_name(src, d0))]
            for src, d0 in mine]

    def run(cmd):
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(r.stderr)

    with ThreadPoolExecutor(max_workers=WORKERS) as ex:
        list(ex.map(run, jobs))
    objs = [os.path.join(tmp if (src, d) in mine else OUT_DIR, _obj_name(src, d)) for src, d in SOURCES]
    run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs)
    return out


LAST_BUILD = {"compiled": [], "reused": [], "linked": False}      # what the newest build() call did (printed by __graft_entry__.build())


def build(force=False, verbose=False):
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    os.makedirs(OUT_DIR, exist_ok=True)
    hdrs = [os.path.join(HERE, h) for h in HEADERS]
    jobs = []
    for src, defines in SOURCES:
        obj = os.path.join(OUT_DIR, _obj_name(src, defines))
        own = [os.path.join(HERE, h) for h in SPLIT2_ONLY] if "_f16" in src else []
        if force or _stale(obj, [os.path.join(HERE, src)] + hdrs + own):
            jobs.append([hipcc] + FLAGS + ["-D" + d for d in defines] + ["-c", os.path.join(HERE, src), "-o", obj])

    def run(cmd):
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed: %s\n%s" % (" ".join(cmd), r.stderr))
        try:
            check_resources(r.stderr, cmd[-3] if len(cmd) > 3 else "link")
        except RuntimeError:
            if os.path.exists(cmd[-1]) and cmd[-1].endswith(".o"):
                os.remove(cmd[-1])       # so that the next build compiles (and checks) this file again
            raise
        if verbose:
            other = "\n".join(l for l in r.stderr.splitlines() if "-Rpass-analysis=kernel-resource-usage" not in l and l.strip())
            if other:
                print(other, file=sys.stderr)

    with ThreadPoolExecutor(max_workers=min(WORKERS, max(1, len(jobs)))) as ex:
        list(ex.map(run, jobs))
    objs = [os.path.join(OUT_DIR, _obj_name(src, defines)) for src, defines in SOURCES]
    relink = bool(force or jobs or _stale(LIB, objs))
    if relink:
        run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs)
    compiled = [os.path.basename(j[-1]) for j in jobs]
    LAST_BUILD.update(compiled=compiled, reused=[os.path.basename(o) for o in objs if os.path.basename(o) not in compiled], linked=relink)
    return LIB


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--variant":      # build.py --variant noside [--only _f16] NNR_ABLATE=1
        rest = sys.argv[3:]
        only = None
        if rest and rest[0] == "--only":
            only, rest = rest[1], rest[2:]
        print(build_variant(sys.argv[2], rest, only))
    else:
        print(build(force="--force" in sys.argv, verbose=True))
