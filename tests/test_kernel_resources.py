"""No scan kernel that ships may touch scratch.

A spilled value's reload — like every scratch access — waits on the in-order vmcnt counter, i.e. on the whole ring of
LDS-DMA prefetches in flight, so a few bytes of scratch in a streaming kernel cost far more than the instruction count
suggests (DESIGN.md §4 "register budget").  The build asks hipcc for its per-kernel resource remarks
(-Rpass-analysis=kernel-resource-usage) and keeps them as csrc/build/<tu>.resources.json; this test parses that record:
the bar is ScratchSize 0, no dynamic stack and no register spills for EVERY kernel of EVERY translation unit — which
covers each instantiation reachable from orama_vec_search / orama_vec_search_device / the shard group and the BM25
entry points, whatever the dimension, batch width and f16_wide mode.
"""
from oramacore_amd import _build

SCAN_UNITS = {"vec_kernels", "vec_multi", "vec_f16", "vec_f16_prep", "vec_f16_pc", "vec_f16_qs", "bm25_kernels",
              "bm25_ranges", "select", "facets"}
# comparison builds (ORAMA_COMPARISON_KERNELS=1 at build time) add the superseded kernels; the product library has none of them
COMPARISON_UNITS = {"vec_f16_wide", "vec_f16_kh", "bm25_ranges_merge"}


def test_remark_parser_reads_hipcc_blocks():
    text = """\
a.hip:61:1: remark: Function Name: _ZN5orama4scanILi2EEEvv [-Rpass-analysis=kernel-resource-usage]
   61 | __global__ void scan() {
      | ^
a.hip:61:1: remark:     SGPRs: 40 [-Rpass-analysis=kernel-resource-usage]
a.hip:61:1: remark:     VGPRs: 173 [-Rpass-analysis=kernel-resource-usage]
a.hip:61:1: remark:     AGPRs: 0 [-Rpass-analysis=kernel-resource-usage]
a.hip:61:1: remark:     ScratchSize [bytes/lane]: 188 [-Rpass-analysis=kernel-resource-usage]
a.hip:61:1: remark:     Dynamic Stack: False [-Rpass-analysis=kernel-resource-usage]
a.hip:61:1: remark:     Occupancy [waves/SIMD]: 2 [-Rpass-analysis=kernel-resource-usage]
a.hip:61:1: remark:     SGPRs Spill: 0 [-Rpass-analysis=kernel-resource-usage]
a.hip:61:1: remark:     VGPRs Spill: 47 [-Rpass-analysis=kernel-resource-usage]
a.hip:61:1: remark:     LDS Size [bytes/block]: 65536 [-Rpass-analysis=kernel-resource-usage]
a.hip:70:5: warning: unused variable 'x' [-Wunused-variable]
"""
    kernels, rest = _build.parse_resource_remarks(text)
    assert kernels == [{"name": "_ZN5orama4scanILi2EEEvv", "sgprs": 40, "vgprs": 173, "agprs": 0, "scratch_bytes_per_lane": 188,
                        "dynamic_stack": False, "occupancy_waves_per_simd": 2, "sgpr_spills": 0, "vgpr_spills": 47,
                        "lds_bytes_per_block": 65536}]
    assert rest.strip() == "a.hip:70:5: warning: unused variable 'x' [-Wunused-variable]"


def test_no_kernel_uses_scratch():
    _build.build_native()  # no-op when the library matches the sources
    per_unit = _build.kernel_resources()
    assert SCAN_UNITS <= set(per_unit), sorted(SCAN_UNITS - set(per_unit))
    assert (COMPARISON_UNITS <= set(per_unit)) if _build.comparison_build() else not (COMPARISON_UNITS & set(per_unit))
    n = 0
    offenders = []
    for unit, kernels in per_unit.items():
        if unit in SCAN_UNITS:
            assert kernels, f"{unit}: the build recorded no kernels (remark format changed?)"
        for k in kernels:
            n += 1
            for field in ("vgprs", "scratch_bytes_per_lane", "dynamic_stack", "vgpr_spills", "sgpr_spills", "occupancy_waves_per_simd"):
                assert field in k, (unit, k)
            if k["scratch_bytes_per_lane"] or k["dynamic_stack"] or k["vgpr_spills"]:
                offenders.append((unit, k["name"], k["scratch_bytes_per_lane"], k["vgpr_spills"]))
    assert n > 400  # 475 instantiations in the product build, ~600 with the comparison kernels
    assert not offenders, offenders


def test_wide_scan_kernels_keep_their_occupancy_plan():
    """The asynchronous wide kernels are written for a fixed number of resident waves: K2q / K2h one block
    of 8 waves per CU (2 per SIMD, <= 256 registers), K2d 16 or 12 waves per CU (<= 128 / <= 168 registers)."""
    per_unit = _build.kernel_resources()
    for unit, floor in (("vec_f16_qs", 2), ("vec_f16_kh", 2), ("vec_f16_pc", 3)):
        for k in per_unit.get(unit, []) if unit in COMPARISON_UNITS else per_unit[unit]:
            if "_kernel" in k["name"] and "prep" not in k["name"]:
                assert k["occupancy_waves_per_simd"] >= floor, (unit, k)


def test_register_budgets_that_buy_a_resident_workgroup():
    """Three kernels owe a resident workgroup per CU to a register count (DESIGN §4 K3r form v6, §4 K4): the plain instantiation of
    K3r's scoring launch (62 VGPRs: the eighth workgroup of 256 threads beside 19.9 KB of LDS) and the two reduction kernels of the
    selections (1 024 threads: two workgroups per CU need <= 64).  A change that costs them is a performance regression the
    parity tests cannot see."""
    _build.build_native()
    per_unit = _build.kernel_resources()
    by_name = {k["name"]: k for unit in ("bm25_ranges", "select") for k in per_unit[unit]}

    def the(fragment):
        hits = [k for name, k in by_name.items() if fragment in name]
        assert hits, fragment
        return hits

    for k in the("range_score_kernelILb0ELb0ELb1E") + the("range_score_compact_kernel"):  # (round 5: the compact-list form too)
        assert k["vgprs"] <= 64 and k["lds_bytes_per_block"] * 8 <= 160 * 1024 and k["scratch_bytes_per_lane"] == 0, k
    for fragment in ("pairs_reduce_kernel", "keys_reduce_kernel", "keys_final_kernel"):
        for k in the(fragment):
            assert k["vgprs"] <= 64 and k["lds_bytes_per_block"] * 2 <= 160 * 1024, k


def test_the_rounds_new_selection_kernels_keep_their_own_code_object():
    """DESIGN §4 K4 "a code object of their own" (profiles/r05_select_unit_split.log): with select_tiny_kernel and
    pairs_reduce_wide_kernel inside select.hip's code object the key-list kernels of that unit ran 20 % longer (same source, same
    instruction counts).  A performance fact no parity test sees: the unit layout is asserted here."""
    _build.build_native()
    per_unit = _build.kernel_resources()
    names = lambda unit: [k["name"] for k in per_unit[unit]]
    assert not any("select_tiny_kernel" in n or "pairs_reduce_wide_kernel" in n for n in names("select"))
    wide = names("select_wide")
    assert len(wide) == 2 and any("select_tiny_kernel" in n for n in wide) and any("pairs_reduce_wide_kernel" in n for n in wide)
    for k in per_unit["select_wide"]:
        assert k["scratch_bytes_per_lane"] == 0 and k["lds_bytes_per_block"] <= 80 * 1024, k
