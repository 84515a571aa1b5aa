"""Build libsdslam_hip.so (HIP kernels + C ABI) in-tree with hipcc for gfx950.

    python -m sdslam_amd.build            # incremental
    python -m sdslam_amd.build --force

hipcc cross-compiles without a GPU.  -ffp-contract=off is REQUIRED: keypoint angles and
descriptor sampling positions are defined by single IEEE float operations (the oracle is built
the same way); the only fused multiply-adds are the explicit ones in sd_sincosf.h.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.environ.get("SD_OUT") or os.path.join(HERE, "libsdslam_hip.so")   # SD_OUT: instrumented builds beside the product (tools/)
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
ARCH = "gfx950"

FLAGS = ["-O3", "-std=c++17", "-fPIC", "-shared", f"--offload-arch={ARCH}", "-ffp-contract=off",
         "-fno-fast-math", "-fno-slp-vectorize", "-Wall", "-Wno-unused-function", "-Wno-unused-variable", "-Wno-unused-result"]


def sources():
    src = [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith((".hip", ".cpp"))]
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(HERE, "..", "include", "sdslam_hip.h")]
    return src, deps


def build(force: bool = False, verbose: bool = False) -> str:
    src, deps = sources()
    if not force and os.path.exists(OUT):
        t = os.path.getmtime(OUT)
        if all(os.path.getmtime(d) <= t for d in deps if os.path.exists(d)):
            return OUT
    cmd = [HIPCC] + FLAGS + os.environ.get("SD_EXTRA_FLAGS", "").split() + ["-x", "hip"] + src + ["-o", OUT]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
