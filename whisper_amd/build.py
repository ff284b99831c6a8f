"""Builds the native libraries in-tree (hipcc cross-compiles gfx950 without a GPU).

    python -m whisper_amd.build            # libwhisper_hip.so (+ libWhisper.so host API when its sources exist)

The .so files land in whisper_amd/lib/ so they travel with the source snapshot to the GPU box.
"""
from __future__ import annotations

import os
import subprocess
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
HOST = os.path.join(HERE, "host")
LIB_DIR = os.path.join(HERE, "lib")
HIP_LIB = os.path.join(LIB_DIR, "libwhisper_hip.so")
HOST_LIB = os.path.join(LIB_DIR, "libWhisper.so")
CLI_BIN = os.path.join(LIB_DIR, "whisper-main")

HIP_SOURCES = ["gemm.hip", "decode1.hip", "attn_enc.hip", "attn_dec.hip", "elementwise.hip", "mel.hip", "exact.hip", "runtime.hip"]
# exact.hip restates the reference CPU path's summation order: a fused multiply-add only where the source says fma()
EXTRA_FLAGS = {"exact.hip": ["-ffp-contract=off"]}
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
ARCH = "gfx950"


def _newer(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _run(cmd):
    print("+", " ".join(cmd), flush=True)
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        sys.stderr.write(r.stdout)
        raise RuntimeError("build step failed: " + " ".join(cmd))
    return r.stdout


def build_hip(force: bool = False) -> str:
    os.makedirs(LIB_DIR, exist_ok=True)
    objs = []
    headers = [os.path.join(CSRC, h) for h in ("common.h", "kernels.h", "epilogue.h", "exact_ops.h")] + [os.path.join(ROOT, "include", "whisper_hip.h")]
    obj_dir = os.path.join(LIB_DIR, "obj")
    os.makedirs(obj_dir, exist_ok=True)
    for s in HIP_SOURCES:
        src = os.path.join(CSRC, s)
        obj = os.path.join(obj_dir, s.replace(".hip", ".o"))
        if force or _newer(obj, [src] + headers):
            # WH_PROBES=1: the tile-shape experiments and ablation instances of tools/*probe* (not in the shipped objects)
            probes = ["-DWH_PROBES"] if os.environ.get("WH_PROBES", "") not in ("", "0") else []
            _run([HIPCC, "--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden", "-Wno-unused-result"] + probes + EXTRA_FLAGS.get(s, []) +
                 ["-I" + os.path.join(ROOT, "include"), "-c", src, "-o", obj])
        objs.append(obj)
    if force or _newer(HIP_LIB, objs):
        _run([HIPCC, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", HIP_LIB] + objs + ["-ldl"])
    return HIP_LIB


def build_host(force: bool = False):
    """libWhisper.so: the COM-style iModel/iContext host API (plain C++, no HIP headers)."""
    if not os.path.isdir(HOST):
        return None
    srcs = sorted(os.path.join(HOST, f) for f in os.listdir(HOST) if f.endswith(".cpp"))
    if not srcs:
        return None
    hdrs = [os.path.join(HOST, f) for f in os.listdir(HOST) if f.endswith(".h")]
    hdrs += [os.path.join(ROOT, "include", f) for f in os.listdir(os.path.join(ROOT, "include"))]
    if force or _newer(HOST_LIB, srcs + hdrs + [HIP_LIB]):
        _run(["g++", "-std=c++17", "-O2", "-fPIC", "-shared", "-fvisibility=hidden", "-I" + os.path.join(ROOT, "include"), "-I" + HOST,
              "-o", HOST_LIB] + srcs + ["-L" + LIB_DIR, "-lwhisper_hip", "-Wl,-rpath,$ORIGIN", "-lpthread"])
    return HOST_LIB


def build_cli(force: bool = False):
    """whisper-main: the command-line tool over libWhisper.so (counterpart of the reference's Examples/main)."""
    cli = os.path.join(HOST, "cli")
    if not os.path.isdir(cli):
        return None
    srcs = sorted(os.path.join(cli, f) for f in os.listdir(cli) if f.endswith(".cpp"))
    hdrs = [os.path.join(cli, f) for f in os.listdir(cli) if f.endswith(".h")] + [os.path.join(ROOT, "include", "whisperApi.h")]
    if force or _newer(CLI_BIN, srcs + hdrs + [HOST_LIB]):
        _run(["g++", "-std=c++17", "-O2", "-Wall", "-I" + os.path.join(ROOT, "include"), "-I" + cli, "-o", CLI_BIN] + srcs +
             ["-L" + LIB_DIR, "-lWhisper", "-Wl,-rpath,$ORIGIN"])
    return CLI_BIN


MGPU_BIN = os.path.join(LIB_DIR, "whisper-mgpu")


def build_mgpu(force: bool = False):
    """whisper-mgpu: one process per GPU over libWhisper.so + the RCCL broadcast of libwhisper_hip.so (plain C++, no torch)."""
    src = os.path.join(HOST, "mgpu", "whisperMgpu.cpp")
    if not os.path.exists(src):
        return None
    hdrs = [os.path.join(ROOT, "include", "whisperApi.h"), os.path.join(ROOT, "include", "whisper_hip.h")]
    if force or _newer(MGPU_BIN, [src, HOST_LIB, HIP_LIB] + hdrs):
        _run(["g++", "-std=c++17", "-O2", "-Wall", "-I" + os.path.join(ROOT, "include"), "-o", MGPU_BIN, src,
              "-L" + LIB_DIR, "-lWhisper", "-lwhisper_hip", "-Wl,-rpath,$ORIGIN", "-lpthread"])
    return MGPU_BIN


ABI_SRC = os.path.join(ROOT, "tests", "abi_caller", "caller.cpp")
ABI_REF_BIN = os.path.join(LIB_DIR, "abi-caller-ref")      # built from the REFERENCE's own headers (only where /root/reference exists)
ABI_OUR_BIN = os.path.join(LIB_DIR, "abi-caller-our")      # the same source against include/whisperApi.h
REFERENCE_ROOT = "/root/reference"


def build_abi_callers(force: bool = False):
    """The boundary proof of tests/test_abi_reference_headers.py: one caller source compiled against the reference's public
    headers where they lie (never copied; the GPU box only gets the binary) and against ours, both linking libWhisper.so."""
    if not os.path.exists(ABI_SRC):
        return None
    link = ["-L" + LIB_DIR, "-lWhisper", "-Wl,-rpath,$ORIGIN"]
    if force or _newer(ABI_OUR_BIN, [ABI_SRC, HOST_LIB, os.path.join(ROOT, "include", "whisperApi.h")]):
        _run(["g++", "-std=c++20", "-O1", "-I" + os.path.join(ROOT, "include"), ABI_SRC, "-o", ABI_OUR_BIN] + link)
    ref_hdr = os.path.join(REFERENCE_ROOT, "Whisper", "API", "whisperComLight.h")
    if os.path.exists(ref_hdr) and (force or _newer(ABI_REF_BIN, [ABI_SRC, HOST_LIB])):
        _run(["g++", "-std=c++20", "-O1", "-D__stdcall=", "-D__cdecl=", "-DUSE_REFERENCE_HEADERS", "-I" + REFERENCE_ROOT, ABI_SRC,
              "-o", ABI_REF_BIN] + link)
    return ABI_OUR_BIN


BATCH_CALLER_SRC = os.path.join(ROOT, "tests", "abi_caller", "batch_caller.cpp")
BATCH_CALLER_BIN = os.path.join(LIB_DIR, "batch-caller")


def build_batch_caller(force: bool = False):
    """tests/abi_caller/batch_caller.cpp: runFullBatch with callbacks against K sequential runFull calls (tests/test_batch_api.py)."""
    if not os.path.exists(BATCH_CALLER_SRC):
        return None
    if force or _newer(BATCH_CALLER_BIN, [BATCH_CALLER_SRC, HOST_LIB, os.path.join(ROOT, "include", "whisperApi.h")]):
        _run(["g++", "-std=c++20", "-O1", "-Wall", "-I" + os.path.join(ROOT, "include"), BATCH_CALLER_SRC, "-o", BATCH_CALLER_BIN,
              "-L" + LIB_DIR, "-lWhisper", "-Wl,-rpath,$ORIGIN"])
    return BATCH_CALLER_BIN


def build_all(force: bool = False):
    t = time.time()
    build_hip(force)
    build_host(force)
    build_cli(force)
    build_mgpu(force)
    build_abi_callers(force)
    build_batch_caller(force)
    print("native build ok in %.1fs" % (time.time() - t), flush=True)


if __name__ == "__main__":
    build_all(force="--force" in sys.argv)
