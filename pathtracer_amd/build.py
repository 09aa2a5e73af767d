"""Builds the native libraries in-tree with hipcc / g++ (no JIT cache: the .so files travel with the tree).

  pathtracer_amd/lib/libpt_hip.so    HIP kernels + C ABI (include/pt_hip.h), gfx950 only
  pathtracer_amd/lib/libpt_host.so   C++ host side mirroring the reference's surface (Shader, Framebuffer,
                                     camera, OBJ/MTL loading, render loop) on top of the C ABI

The device flags are part of the numerics contract: no fp contraction, denormals flushed, IEEE divide/sqrt.
-fno-slp-vectorize is a speed flag: packed fp32 (v_pk_mul_f32/v_pk_add_f32) runs at half the instruction rate on
gfx950, so pairing scalar operations into packed ones only adds register shuffles (measured: 1.3x slower with them).
"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
HOST = os.path.join(HERE, "host")
LIB = os.path.join(HERE, "lib")

ROCM = os.environ.get("ROCM_PATH", "/opt/rocm")
HIPCC = os.environ.get("HIPCC", os.path.join(ROCM, "bin", "hipcc"))
DEVICE_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
                "-fgpu-flush-denormals-to-zero", "-fno-fast-math", "-fno-slp-vectorize", "-fvisibility=hidden", "-Wall", "-Wextra",
                "-Wno-unused-parameter"]


def _newer(target, sources):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in sources)


def _run(cmd):
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        sys.stderr.write(" ".join(cmd) + "\n" + r.stdout + r.stderr)
        raise RuntimeError("build failed: " + os.path.basename(cmd[-1]))
    return r


def build_hip(force=False, save_temps=False, extra_flags=(), tag=""):
    """tag/extra_flags build an experimental variant next to the product library (tools/ab_bench.py)."""
    os.makedirs(LIB, exist_ok=True)
    out = os.path.join(LIB, f"libpt_hip{('__' + tag) if tag else ''}.so")
    srcs = [os.path.join(CSRC, f) for f in ("pt_kernels.hip", "pt_bvh.hip", "pt_grid.hip", "pt_capi.hip")]
    deps = srcs + [os.path.join(CSRC, "pt_internal.h"), os.path.join(ROOT, "include", "pt_hip.h"), __file__]
    if not force and not _newer(out, deps):
        return out
    if shutil.which(HIPCC) is None:
        raise RuntimeError(f"{HIPCC} not found: the HIP library cannot be built")
    cmd = [HIPCC] + DEVICE_FLAGS + list(extra_flags) + ["-shared", "-o", out] + srcs
    if save_temps:
        tmp = os.path.join(LIB, "temps")
        os.makedirs(tmp, exist_ok=True)
        cmd = [HIPCC] + DEVICE_FLAGS + ["-save-temps=obj", "-c", "-o", os.path.join(tmp, "pt_kernels.o"), srcs[0]]
    _run(cmd)
    return out


def build_host(force=False):
    os.makedirs(LIB, exist_ok=True)
    out = os.path.join(LIB, "libpt_host.so")
    if not os.path.isdir(HOST):
        return None
    srcs = sorted(os.path.join(HOST, f) for f in os.listdir(HOST) if f.endswith(".cpp") and not f.endswith("_main.cpp"))
    hdrs = [os.path.join(HOST, f) for f in os.listdir(HOST) if f.endswith(".h")]
    if not srcs:
        return None
    if not force and not _newer(out, srcs + hdrs + [os.path.join(ROOT, "include", "pt_hip.h"), __file__]):
        return out
    # (MultiGpu.cpp calls the HIP runtime for streams, events and copies and takes RCCL's types from its header; librccl itself is
    # opened at run time)
    cmd = ["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-Wall", "-Wextra", "-pthread", "-D__HIP_PLATFORM_AMD__",
           f"-I{os.path.join(ROOT, 'include')}", f"-I{ROCM}/include", "-o", out] + srcs + \
          [f"-L{LIB}", "-lpt_hip", f"-L{ROCM}/lib", "-lamdhip64", "-ldl", "-Wl,-rpath,$ORIGIN", f"-Wl,-rpath,{ROCM}/lib"]
    _run(cmd)
    return out


def build_cli(force=False):
    """pathtracer_amd/lib/pt_render: headless counterpart of the reference executable."""
    out = os.path.join(LIB, "pt_render")
    src = os.path.join(HOST, "pt_render_main.cpp")
    if not os.path.exists(src):
        return None
    if not force and not _newer(out, [src, os.path.join(LIB, "libpt_host.so")]):
        return out
    _run(["g++", "-O2", "-std=c++17", "-ffp-contract=off", f"-I{os.path.join(ROOT, 'include')}", "-o", out, src,
          f"-L{LIB}", "-lpt_host", "-lpt_hip", "-Wl,-rpath,$ORIGIN"])
    return out


def build_all(force=False):
    return [build_hip(force), build_host(force), build_cli(force)]


def build_sanitize():
    """The host's scene code under AddressSanitizer + UndefinedBehaviorSanitizer (host compiler only), in the test executable
    that tests/native/Makefile describes and tests/test_sanitizers.py runs. Returns nothing the product loads."""
    _run(["make", "-C", os.path.join(ROOT, "tests", "native"), "-s", "asan"])


if __name__ == "__main__":
    if "--sanitize" in sys.argv:
        build_sanitize()
        sys.exit(0)
    for p in build_all(force="--force" in sys.argv):
        print(p)
