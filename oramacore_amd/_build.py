"""In-tree build of liborama_hip.so (hipcc, gfx950 only).

`build_native()` compiles every .hip translation unit under oramacore_amd/csrc with
`hipcc --offload-arch=gfx950` and links them into oramacore_amd/csrc/liborama_hip.so.  The shared
object is git-ignored but travels to the GPU box with the repo snapshot.  hipcc cross-compiles
without a GPU, so this is also the CPU-side "does it build" check.
"""
from __future__ import annotations

import hashlib
import json
import os
import re
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
CSRC = ROOT / "oramacore_amd" / "csrc"
INCLUDE = ROOT / "include"
LIB = CSRC / "liborama_hip.so"
OBJ_DIR = CSRC / "build"
# the comparison flavour (ORAMA_COMPARISON_KERNELS=1 in the environment) is a SEPARATE library with its own object directory:
# building or loading it never touches the product library
LIB_CMP = CSRC / "liborama_hip_cmp.so"
OBJ_DIR_CMP = CSRC / "build_cmp"

ARCH = "gfx950"
COMMON_FLAGS = [
    f"--offload-arch={ARCH}",
    "-O3",
    "-std=c++17",
    "-fPIC",
    "-Wall",
    "-Wno-unused-result",
    # one remark block per kernel (registers, scratch, LDS, occupancy): parsed into csrc/build/<tu>.resources.json, which
    # tests/test_kernel_resources.py reads — a scan kernel that spills is a build regression, not a profile finding
    "-Rpass-analysis=kernel-resource-usage",
    f"-I{INCLUDE}",
    f"-I{CSRC}",
]
# Translation units whose f32 arithmetic must round exactly like the scalar reference
# (BM25F scores must be bit-identical to the scalar f32 evaluation): no FMA contraction.
EXACT_FP = {"fulltext.hip", "bm25_kernels.hip", "bm25_ranges.hip", "bm25_ranges_fast.hip", "bm25_ranges_merge.hip", "hybrid_tail.hip"}
# Superseded kernels kept for A/B measurements only — K2c (round-1 wide fp16 scan), K2h (K-split wave pairs, 6 % slower than
# K2q) and the round-3 merge-tree form of K3r's scoring launch.  They are compiled and linked only when the environment says
# ORAMA_COMPARISON_KERNELS=1 at build time; the product library does not contain them.
# Round 6: + hybrid_tail.hip, the device form of the one-call hybrid search's tail — built, bit-identical and 40-70 us slower than
# the host tail (profiles/r05_hybrid_device_tail_ab.log): it left the product library instead of shipping as a dead path.
#          + bm25_ranges_fast.hip, K3r's scoring launch with rank-free singletons and lists under the published floor left unscored —
# bit-identical, 14 % fewer vector instructions, 66-80 % of the wave iterations unscored, and 3 % SLOWER (profiles/r06_k3r_fast_body.md).
COMPARISON_UNITS = {"vec_f16_wide.hip", "vec_f16_kh.hip", "bm25_ranges_merge.hip", "hybrid_tail.hip", "bm25_ranges_fast.hip"}


def comparison_build() -> bool:
    return os.environ.get("ORAMA_COMPARISON_KERNELS", "0") == "1"


def lib_path() -> Path:
    return LIB_CMP if comparison_build() else LIB


def obj_dir() -> Path:
    return OBJ_DIR_CMP if comparison_build() else OBJ_DIR


def _flags() -> list[str]:
    extra = [f"-DORAMA_K3R_WG={os.environ['ORAMA_K3R_WG']}"] if os.environ.get("ORAMA_K3R_WG") else []  # (experiments: 512-thread scoring workgroups)
    return [*COMMON_FLAGS, f"-DORAMA_COMPARISON_KERNELS={1 if comparison_build() else 0}", *extra]


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and Path(cand).exists():
            return cand
    raise RuntimeError("hipcc not found — liborama_hip.so cannot be built (no CPU fallback exists)")


def _sources() -> list[Path]:
    return sorted(p for p in CSRC.glob("*.hip") if comparison_build() or p.name not in COMPARISON_UNITS)


def _fingerprint() -> str:
    h = hashlib.sha256()
    for p in sorted(list(CSRC.glob("*.hip")) + list(CSRC.glob("*.hpp")) + list(CSRC.glob("*.inc")) + list(INCLUDE.glob("*.h"))):
        h.update(p.name.encode())
        h.update(p.read_bytes())
    # flags without the checkout-specific include paths: the tree is shipped to the GPU box under another root, and a
    # path in the fingerprint would force a rebuild there on every run
    h.update(" ".join(f for f in _flags() if not f.startswith("-I")).encode())
    return h.hexdigest()


_REMARK = re.compile(r"remark: (?:[^:]+:\d+:\d+: )?\s*(Function Name|SGPRs|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Dynamic Stack|"
                     r"Occupancy \[waves/SIMD\]|SGPRs Spill|VGPRs Spill|LDS Size \[bytes/block\]): (\S+)")
_KEYS = {"SGPRs": "sgprs", "VGPRs": "vgprs", "AGPRs": "agprs", "ScratchSize [bytes/lane]": "scratch_bytes_per_lane",
         "Dynamic Stack": "dynamic_stack", "Occupancy [waves/SIMD]": "occupancy_waves_per_simd", "SGPRs Spill": "sgpr_spills",
         "VGPRs Spill": "vgpr_spills", "LDS Size [bytes/block]": "lds_bytes_per_block"}


def parse_resource_remarks(stderr: str) -> tuple[list[dict], str]:
    """(kernels, everything else): the -Rpass-analysis=kernel-resource-usage blocks of one compile as records
    {name (mangled), vgprs, agprs, sgprs, scratch_bytes_per_lane, dynamic_stack, occupancy_waves_per_simd, ...}."""
    kernels: list[dict] = []
    rest = []
    in_remark = False
    lines = stderr.splitlines()
    for li, line in enumerate(lines):
        m = _REMARK.search(line)
        if not m and line.startswith("In file included from") and li + 1 < len(lines) and (
                _REMARK.search(lines[li + 1]) or lines[li + 1].startswith("In file included from")):
            continue  # the include chain printed in front of a remark about a kernel of an included file (select_wide.hip)
        if not m:
            if "[-Rpass-analysis=kernel-resource-usage]" in line:
                in_remark = True
            elif in_remark and re.match(r"\s*\d*\s*\|", line):
                pass  # the source line / caret a remark is printed with
            else:
                in_remark = False
                rest.append(line)
            continue
        in_remark = True
        key, val = m.group(1), m.group(2)
        if key == "Function Name":
            kernels.append({"name": val})
        elif kernels:
            kernels[-1][_KEYS[key]] = (val == "True") if key == "Dynamic Stack" else int(val)
    return kernels, "\n".join(rest)


def kernel_resources() -> dict[str, list[dict]]:
    """{translation unit: [kernel record, ...]} of the library as last built (csrc/build/*.resources.json)."""
    return {p.name[: -len(".resources.json")]: json.loads(p.read_text()) for p in sorted(obj_dir().glob("*.resources.json"))}


def native_is_fresh() -> bool:
    stamp = obj_dir() / "fingerprint"
    return lib_path().exists() and stamp.exists() and stamp.read_text() == _fingerprint()


def build_native(force: bool = False, verbose: bool = True) -> Path:
    LIB, OBJ_DIR = lib_path(), obj_dir()  # (of the flavour the environment names)
    if not force and native_is_fresh():
        return LIB
    hipcc = _hipcc()
    OBJ_DIR.mkdir(exist_ok=True)

    def compile_one(src: Path) -> Path:
        obj = OBJ_DIR / (src.stem + ".o")
        flags = _flags()
        flags.append("-ffp-contract=off" if src.name in EXACT_FP else "-ffp-contract=fast")
        cmd = [hipcc, *flags, "-c", str(src), "-o", str(obj)]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {src.name}:\n{r.stdout}\n{r.stderr}")
        kernels, rest = parse_resource_remarks(r.stderr)
        (OBJ_DIR / (src.stem + ".resources.json")).write_text(json.dumps(kernels, indent=1))
        if verbose and rest.strip():
            sys.stderr.write(rest + "\n")
        return obj

    keep = {src.stem for src in _sources()}
    for stale in list(OBJ_DIR.glob("*.o")) + list(OBJ_DIR.glob("*.resources.json")):  # units of the other build flavour
        if stale.name.split(".")[0] not in keep:
            stale.unlink()
    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as ex:
        objs = list(ex.map(compile_one, _sources()))
    cmd = [hipcc, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", str(LIB), *map(str, objs), "-lpthread", "-ldl"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    (OBJ_DIR / "fingerprint").write_text(_fingerprint())
    return LIB


if __name__ == "__main__":
    print(build_native(force="--force" in sys.argv))
