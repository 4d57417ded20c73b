"""Build libconv3p_hip.so (the C-ABI library of include/conv3p.h) for gfx950, in-tree.

    python -m pointwise_amd.build            # build if sources are newer than the library
    python -m pointwise_amd.build --force

hipcc cross-compiles without a GPU.  Flags:
  -ffp-contract=off       the tap arithmetic must not be contracted (see conv3p_device.hpp);
                          FMAs in the accumulation loops are written explicitly
  -munsafe-fp-atomics     hardware float atomics (ds_add_f32 / global_atomic_add_f32) instead of
                          CAS loops; all buffers are ordinary coarse-grained device memory
hipcc's default -fhip-fp32-correctly-rounded-divide-sqrt keeps `/` IEEE-exact on the device.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(CSRC, "libconv3p_hip.so")
def sources():
    """Every file the library is compiled from: the one translation unit plus all headers beside it."""
    return sorted(f for f in os.listdir(CSRC) if f.endswith((".hip", ".hpp", ".h")))


HEADER = os.path.join(os.path.dirname(HERE), "include", "conv3p.h")
ARCH = "gfx950"

FLAGS = ["-O3", "-std=c++17", "-ffp-contract=off", "-munsafe-fp-atomics", "-fPIC", "-shared",
         "-Wall", "-Wextra", "-Wno-unused-parameter"]


def source_hash():
    """sha256 over the library's sources (csrc/*.hip, *.hpp, include/conv3p.h), 16 hex digits: what the counter files
    under profiles/ are stamped with, so that bench.py can tell when they were collected for other kernels."""
    import hashlib
    h = hashlib.sha256()
    for f in [os.path.join(CSRC, s) for s in sources()] + [HEADER]:
        h.update(os.path.basename(f).encode())
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in sources()] + [HEADER]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    if not force and not stale():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc, "--offload-arch=" + ARCH] + FLAGS + ["-o", LIB, os.path.join(CSRC, "conv3p_abi.hip")]
    if verbose:
        print(" ".join(cmd))
    out = subprocess.run(cmd, capture_output=True, text=True)
    if out.returncode != 0:
        raise RuntimeError("hipcc failed:\n" + out.stdout + out.stderr)
    return LIB


def kernel_resources(lib=LIB):
    """[(mangled kernel name, vgprs, vgpr spills, sgpr spills, scratch bytes per lane)] from the notes of the gfx950 code
    object inside the built library (llvm-objcopy / clang-offload-bundler / llvm-readelf of the ROCm toolchain)."""
    import re
    import tempfile
    llvm = os.environ.get("ROCM_LLVM_BIN", "/opt/rocm/lib/llvm/bin")
    with tempfile.TemporaryDirectory() as d:
        fat, co = os.path.join(d, "fat.bin"), os.path.join(d, "co.elf")
        subprocess.run([os.path.join(llvm, "llvm-objcopy"), "-O", "binary", "--only-section=.hip_fatbin", lib, fat], check=True)
        subprocess.run([os.path.join(llvm, "clang-offload-bundler"), "--unbundle", "--type=o", "--input=" + fat,
                        "--targets=hipv4-amdgcn-amd-amdhsa--" + ARCH, "--output=" + co], check=True)
        notes = subprocess.run([os.path.join(llvm, "llvm-readelf"), "--notes", co], check=True, capture_output=True,
                               text=True).stdout
    out = []
    for blk in notes.split("- .agpr_count:")[1:]:
        get = lambda k: re.search(r"\.%s:\s*(\S+)" % k, blk).group(1)
        out.append((get("name"), int(get("vgpr_count")), int(get("vgpr_spill_count")), int(get("sgpr_spill_count")),
                    int(get("private_segment_fixed_size"))))
    return out


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
