"""Build libechohip.so (hipcc, gfx950) in-tree next to this file.  No CPU fallback is ever built."""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libechohip.so")
SOURCES = ["gemm_pp.hip", "gemm.hip", "gemm_f32.hip", "attention.hip", "elementwise.hip", "dac.hip", "postproc.hip", "engine.hip"]
# attention.hip: the one-wave-per-SIMD kernel places every VALU instruction in an MFMA gap by hand; SLP vectorisation would turn its
# scalar fp32 adds into v_pk_add_f32 plus the v_mov shuffles that feed them (cdna_hip_programming.md Appendix B, pitfalls)
EXTRA_FLAGS = {"attention.hip": ["-fno-slp-vectorize"]}
# No floating-point contraction: hipcc's default (fast-honor-pragmas) fuses a * c - b * s into an fma wherever it likes, and it liked different
# places in two copies of the same epilogue (the interior-tile and edge-tile forms of gemm_pp's fused QKV tail): a token's RoPE output then depended
# on which tile of the launch it sat in (1-4 elements per launch one bf16 ulp apart; caught by the row-permutation test).  Every rounding point of
# the reference's arithmetic (a * c - b * s, the Euler update, x * rs * w, ...) is a separate instruction now, in every kernel; the fused
# multiply-adds that are wanted are written as fmaf().
COMMON_FLAGS = ["-ffp-contract=off"]


def _hipcc() -> str:
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", shutil.which("hipcc")):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found (ROCm toolchain required)")


def needs_build() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(HERE, "..", "include", "echo_hip.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = True) -> str:
    if not force and not needs_build():
        return LIB
    objs = []
    os.makedirs(os.path.join(HERE, "build"), exist_ok=True)
    procs = []
    for src in SOURCES:
        obj = os.path.join(HERE, "build", src.replace(".hip", ".o"))
        cmd = [_hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC"] + COMMON_FLAGS + EXTRA_FLAGS.get(src, []) + ["-c", os.path.join(CSRC, src), "-o", obj]
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
        objs.append(obj)
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{out}")
        if verbose and out.strip():
            print(out, file=sys.stderr)
    cmd = [_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError("link failed:\n" + r.stdout)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
