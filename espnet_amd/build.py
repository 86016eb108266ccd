"""Build libespnet_amd.so (all HIP kernels + the C ABI of include/espnet_amd.h) for gfx950.

    python -m espnet_amd.build            # incremental
    python -m espnet_amd.build --force

hipcc cross-compiles without a GPU, so this runs in the CPU-only build container; the resulting
in-tree .so (git-ignored) travels to the GPU box with the repo snapshot.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

HERE = Path(__file__).resolve().parent
CSRC = HERE / "csrc"
OUT = HERE / "lib"
LIB = OUT / "libespnet_amd.so"
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-gpu-rdc",
         "-Wno-unused-result"]
# per-file extras.  block.hip: its kernels live at the 256-VGPR line with MFMA accumulators carried around loops; in
# hipcc's default "AGPR form" every loop trip copies the accumulators VGPR <-> AGPR (~100 v_accvgpr moves per FFN
# iteration, as much issue time as the MFMAs).  VGPR form keeps them where the VALU epilogues need them.
EXTRA_FLAGS = {"block.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form"]}


def _sources():
    return sorted(CSRC.glob("*.hip")) + sorted(CSRC.glob("*.cpp"))  # .cpp: host-only code of the C ABI


def _deps_mtime():
    hdrs = list(CSRC.glob("*.h")) + list(CSRC.glob("*.inc")) + [HERE.parent / "include" / "espnet_amd.h"]
    return max(p.stat().st_mtime for p in hdrs)


def build(force: bool = False, verbose: bool = True) -> Path:
    OUT.mkdir(exist_ok=True)
    hdr_m = _deps_mtime()
    jobs = []
    objs = []
    for src in _sources():
        obj = OUT / (src.stem + ".o")
        objs.append(obj)
        if force or not obj.exists() or obj.stat().st_mtime < max(src.stat().st_mtime, hdr_m):
            jobs.append((src, obj))

    def cc(job):
        src, obj = job
        flags = FLAGS + EXTRA_FLAGS.get(src.name, []) if src.suffix == ".hip" else ["-O3", "-std=c++17", "-fPIC", "-pthread"]
        cmd = [HIPCC, *flags, "-c", str(src), "-o", str(obj)]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {src.name}:\n{r.stderr}")
        if verbose:
            print(f"[espnet_amd.build] compiled {src.name}", flush=True)

    if jobs:
        with ThreadPoolExecutor(max_workers=min(8, len(jobs))) as ex:
            list(ex.map(cc, jobs))
    if jobs or not LIB.exists():
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-pthread", "-Wl,-z,defs", "-o", str(LIB), *map(str, objs)]  # -z defs: an undefined symbol fails HERE, not at the first call on the GPU box
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stderr}")
        if verbose:
            print(f"[espnet_amd.build] linked {LIB}", flush=True)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
