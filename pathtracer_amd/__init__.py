"""MI355X-native path-tracing core: HIP kernels behind a C ABI (include/pt_hip.h), a C++ host that mirrors the
reference's Shader / Framebuffer / render-loop surface (pathtracer_amd/host), and thin Python plumbing for tests,
benchmarks and multi-GPU runs. No CPU fallback exists: loading fails loudly when the HIP library is missing."""
from . import capi  # noqa: F401
from .capi import Tracer, PtError  # noqa: F401
