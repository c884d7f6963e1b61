"""CPU tests of the drop-in boundary: liborama_hip.so loads, exports every symbol that
include/orama_hip.h declares, and refuses to compute without a HIP device (no CPU fallback)."""
import ctypes as C
import re

import pytest

from oramacore_amd import _build, _native as N


def test_library_builds_and_loads():
    lib = N.load()
    assert lib.orama_abi_version() == 2
    assert _build.lib_path().exists()


def test_every_declared_symbol_is_exported():
    lib = N.load()
    names = N.declared_symbols()
    assert len(names) >= 25
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, f"declared in include/orama_hip.h but not exported: {missing}"


def test_header_cites_reference_interfaces():
    text = N.HEADER.read_text()
    for needle in ("embedding_field.rs", "token_score.rs", "bm25.rs", "sort.rs", "search.rs"):
        assert needle in text
    assert 'extern "C"' in text


def test_signatures_are_plain_c():
    """No C++/torch types in the boundary: the header must compile as C."""
    import subprocess
    import tempfile

    with tempfile.NamedTemporaryFile("w", suffix=".c") as f:
        f.write('#include "orama_hip.h"\nint main(void){return (int)sizeof(orama_bm25_params);}\n')
        f.flush()
        r = subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-fsyntax-only", f"-I{N.HEADER.parent}", f.name],
                           capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


def test_no_cpu_fallback_without_gpu(has_gpu):
    if has_gpu:
        pytest.skip("GPU present")
    lib = N.load()
    h = C.c_void_p()
    st = lib.orama_ctx_create(0, C.byref(h))
    assert st == N.ORAMA_ERR_HIP
    assert b"no CPU fallback" in lib.orama_last_error()


def test_product_never_imports_the_oracle():
    """oracle/ is test infrastructure: nothing under oramacore_amd/ (or include/) may reference it."""
    root = _build.ROOT
    offenders = []
    for p in list((root / "oramacore_amd").rglob("*.py")) + list((root / "oramacore_amd" / "csrc").glob("*.h*")):
        t = p.read_text(errors="replace")
        if re.search(r"\boracle\b|orama_oracle|liborama_oracle", t):
            offenders.append(str(p))
    assert not offenders, offenders


def test_integration_doc_mentions_every_entry_point():
    """INTEGRATION.md shows the binding (or says why none is needed) for every symbol the header declares."""
    from oramacore_amd import _native

    doc = (_build.ROOT / "INTEGRATION.md").read_text()
    missing = [s for s in _native.declared_symbols() if s not in doc]
    assert not missing, missing


def test_product_library_reads_only_the_deployment_environment():
    """46 ORAMA_* switches — debug hooks, ablations, one that could make answers wrong — were read by
    liborama_hip.so.  The product library's environment is the deployment allow-list; everything else goes through
    orama::dev_env (nullptr unless built as the comparison flavour) or orama_ctx_set_option."""
    allow = {"ORAMA_RCCL_LIB", "ORAMA_SCRATCH_POOL_MIB", "ORAMA_MAX_INFLIGHT", "ORAMA_ACQUIRE_TIMEOUT_MS", "ORAMA_SHARD_LANES",
             "ORAMA_TWO_STAGE", "ORAMA_VMM"}
    seen, offenders = set(), []
    for p in sorted(list(_build.CSRC.glob("*.hip")) + list(_build.CSRC.glob("*.hpp")) + list(_build.CSRC.glob("*.inc"))):
        text = p.read_text(errors="replace")
        for m in re.finditer(r"(?<![A-Za-z_:])(?:std::)?getenv\(([^)]*)\)", text):
            arg = m.group(1).strip()
            line = text[: m.start()].count("\n") + 1
            ctx_line = text.splitlines()[line - 1]
            if p.name == "common.hpp" and "dev_env" in text[max(0, m.start() - 300): m.start()]:
                continue  # dev_env itself
            name = arg.strip('"')
            if arg.startswith('"') and name in allow:
                seen.add(name)
                continue
            offenders.append(f"{p.name}:{line}: {ctx_line.strip()}")
    assert not offenders, offenders
    assert seen == allow, (seen ^ allow)
    # and INTEGRATION.md documents exactly that list as the deployment environment
    doc = (_build.ROOT / "INTEGRATION.md").read_text()
    for name in allow:
        assert f"`{name}`" in doc, name
