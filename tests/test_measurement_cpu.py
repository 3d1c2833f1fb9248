"""The numbers bench.py does not measure live must not go stale silently: the VALU instruction
counts it prices paths_kernel with are re-derived here from the kernels as they compile now, and
the PMC traffic table must name where each figure came from."""
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))


@pytest.fixture(scope="module")
def asm(tmp_path_factory):
    import isa_loop_count as I
    return I.emit_asm(str(tmp_path_factory.mktemp("isa") / "smmc_kernels.s"))


@pytest.fixture(scope="module")
def ref_asm(tmp_path_factory):  # the reference-stream kernels: 40 s of hipcc, once for the module
    import isa_loop_count as I
    return I.emit_asm(str(tmp_path_factory.mktemp("isa_ref") / "smmc_ref_kernels.s"), "smmc_ref_kernels.hip")


@pytest.mark.parametrize("mode", ["gaussian", "table"])
def test_valu_instruction_counts_match_the_built_kernels(asm, mode):
    import bench
    import isa_loop_count as I
    variant, periods = I.VARIANTS[mode]
    c = I.count(asm, variant)
    assert I.valu(c) == pytest.approx(bench.VALU_INSTS_PER_STEP[mode] * periods), dict(c)
    # the range-checked divide adds its two compares per Philox block and nothing else
    chk = I.count(asm, I.VARIANTS[mode + "_checked"][0])
    assert I.valu(chk) - I.valu(c) == pytest.approx(bench.VALU_CHECK_PER_STEP[mode] * periods)
    # one LDS gather per period: a table entry, or (Gaussian) a b128 radius row and a b64 trig pair per two periods
    assert I.lds(c) == periods
    # the gfx950 forms the loop is built on are really there
    assert c["v_bitop3_b32"] >= 15 and c["v_mad_u64_u32"] >= 16


def test_ref_stream_instruction_counts_match_the_built_kernel(ref_asm):
    """bench.py --stream ref prices ref_windowed_kernel with three counts: per step of the seed run-up, per
    output below output 227 (two seed chains) and from 227 on (three chains, the earlier output generated
    again).  Re-derived from the kernel as it compiles now: the two written-out 4-output loops are the two
    largest innermost loops, the run-up is the 8-step loop of plain seed steps."""
    import bench
    import isa_loop_count as I
    import isa_loops as L
    lines = open(ref_asm).read().splitlines()
    body = L.kernel_body(lines, "ref_windowed_kernelILi0ELb0E")
    found = sorted((L.summary(c)[0], c) for _, c in L.loops(body))
    (hi, c_hi), (lo, c_lo) = found[-1], found[-2]
    assert hi == pytest.approx(bench.REF_VALU["output_hi"] * 4) and lo == pytest.approx(bench.REF_VALU["output_lo"] * 4)
    for c, chains in ((c_lo, 2), (c_hi, 3)):
        # a chain step is lshr, xor and ONE v_mad_u64_u32 (hipcc's own v_mul_lo_u32 + v_add was 18 % slower overall);
        # four more of them are the Lemire products
        assert c["ds_read_b32"] == 4 and c["v_mad_u64_u32"] == 4 * chains + 4 and "v_mul_lo_u32" not in c
        assert c["v_bfi_b32"] == 4 * (chains - 1) and c["v_cmp_gt_u32_e32"] == 4  # the rejection test: one v_cmp
        assert "v_cndmask_b32_e32" not in c and "v_cndmask_b32_e64" not in c      # no per-lane select in the loop
    # the run-up: the innermost loop of eight bare chain steps (no LDS, nothing but lshr / xor / mad)
    runups = [c for _, c in L.loops(body) if c.get("v_mad_u64_u32") == 8 and "ds_read_b32" not in c]
    assert len(runups) == 1
    assert L.summary(runups[0])[0] == pytest.approx(bench.REF_VALU["runup_step"] * 8)
    assert bench.ref_valu_per_path(360) == pytest.approx(397 * 3 + 227 * 25 + 133 * 32)
    assert 0 < bench.ref_valu_per_path(0) < bench.ref_valu_per_path(1)
    # ref_tree_kernel (455 .. 1816 periods): one written-out 4-output loop per stretch, in program order; the first two
    # are the windowed kernel's own counts; every later stretch has one more made word (v_bfi_b32: one per twist) and,
    # every second or third stretch, one more live seed chain (v_mad_u64_u32) -- streams are shared between users
    tree = L.kernel_body(lines, "ref_tree_kernelILi0ELb0ELi1816E")
    big = [c for _, c in L.loops(tree) if c.get("ds_read_b32") == 4]
    assert [L.summary(c)[0] for c in big] == [pytest.approx(4 * per) for _, per in bench.REF_TREE_STRETCHES]
    assert [c["v_mad_u64_u32"] // 4 - 1 for c in big] == [2, 3, 4, 4, 5, 5, 6, 6, 7, 7, 7, 8, 8, 8, 9, 9, 9]   # live seed chains
    assert [c["v_bfi_b32"] // 4 for c in big] == list(range(1, 18))                                       # made words per output
    assert all("v_mul_lo_u32" not in c and "scratch_load_dword" not in c for c in big)
    # the instantiation for paths of up to 1077 periods has the same loops for its seven stretches
    short = [c for _, c in L.loops(L.kernel_body(lines, "ref_tree_kernelILi0ELb0ELi1077E")) if c.get("ds_read_b32") == 4]
    assert [L.summary(c)[0] for c in short] == [L.summary(c)[0] for c in big[:7]]
    # the checked-divide variant (what the bundled table gets at 1000 periods) adds, at every check site, two float
    # compares, a select and the flag's v_cmp (bench.REF_CHECK_VALU per 8 periods) and nothing else on the VALU
    from collections import Counter
    def whole(sym):
        body = L.kernel_body(lines, sym)
        return Counter(x.split()[0] for x in body if x.strip() and x.strip()[0] not in ";." and not x.startswith("_Z"))
    fast, chk = whole("ref_tree_kernelILi0ELb0ELi1077E"), whole("ref_tree_kernelILi2ELb0ELi1077E")
    sites = chk["v_cmp_ngt_f32_e64"] - fast["v_cmp_ngt_f32_e64"]
    assert sites > 0 and bench.REF_CHECK_VALU == 4
    spill = ("v_mov_b32_e32", "v_readlane_b32", "v_writelane_b32")  # copies and scalar spills between the stretches, none in a loop
    extra = {k: chk[k] - fast[k] for k in set(chk) | set(fast) if k.startswith("v_") and chk[k] != fast[k] and k not in spill}
    assert not any(c.get("v_readlane_b32") or c.get("v_writelane_b32") for _, c in L.loops(tree))
    assert extra == {"v_cmp_ngt_f32_e64": sites, "v_cmp_nlt_f32_e32": sites, "v_cndmask_b32_e64": sites, "v_cmp_ne_u32_e32": sites}, extra
    assert bench.ref_valu_per_path(1000) == pytest.approx(397 * 3 + 227 * 25 + 227 * 32 + 169 * 40 + 58 * 45 + 169 * 52 + 58 * 57 + 92 * 65)
    assert bench.ref_valu_per_path(360) == pytest.approx(397 * 3 + 227 * 25 + 133 * 32)


def test_pmc_traffic_table_names_its_sources():
    import bench
    table = json.load(open(bench.PMC_TRAFFIC_FILE))
    assert "gaussian|100000000|360|all" in table
    for key, rec in table.items():
        mode, n, periods, outputs = key.split("|")
        assert mode in ("gaussian", "table", "ref") and outputs in ("all", "final", "stats", "host")
        src = rec["source"].split(" ")[0]
        assert src.startswith("profiles/") and os.path.exists(os.path.join(ROOT, src)), src
        # algorithmic bytes: 4 B per path (+ 8 B per 256-path chunk): traffic must not be below them
        if outputs in ("all", "final"):
            assert rec["bytes"] >= 4.0 * int(n)
            assert rec["bytes"] <= 1.25 * (4.0 + 8.0 / 256.0 * 8) * int(n)  # chunk stats are counted as 32-byte writes
    assert bench.pmc_traffic("gaussian", 12345, 360, "all") == (None, None)


def test_pmc_traffic_belongs_to_the_kernels_as_they_compile_now(asm, ref_asm, tmp_path):
    """Every entry of profiles/pmc_traffic.json carries the fingerprint of the kernel that was profiled (its
    instruction mnemonics in program order) and a digest of the kernel sources + compiler flags.  A kernel
    change without a new PMC pass fails here; bench.py itself reports `traffic: null` for a stale entry."""
    import bench
    import isa_loop_count as I
    table = json.load(open(bench.PMC_TRAFFIC_FILE))
    for key, rec in table.items():
        mode = I.traffic_kernel_of(key)
        src, kernel, variant = I.TRAFFIC_KERNELS[mode]
        assert rec["kernel"] == kernel + variant
        path = asm
        if src != "smmc_kernels.hip":
            assert src == "smmc_ref_kernels.hip"
            path = ref_asm
        assert rec["isa_fingerprint"] == I.fingerprint(path, variant, kernel), (
            f"{key}: the kernel changed since the PMC pass ({rec['source']}): run tools/profile_r04.sh again")
        assert rec["source_sha256"] == I.source_digest(mode), f"{key}: kernel sources or flags changed since the PMC pass"
    b, src = bench.pmc_traffic("gaussian", 100_000_000, 360, "all")
    assert b and src and src.startswith("profiles/")
    # a stale entry is not quoted
    stale = dict(table)
    stale["gaussian|100000000|360|all"] = dict(stale["gaussian|100000000|360|all"], source_sha256="0" * 64)
    p = tmp_path / "stale.json"
    p.write_text(json.dumps(stale))
    old = bench.PMC_TRAFFIC_FILE
    try:
        bench.PMC_TRAFFIC_FILE = str(p)
        b, why = bench.pmc_traffic("gaussian", 100_000_000, 360, "all")
        assert b is None and why.startswith("stale")
    finally:
        bench.PMC_TRAFFIC_FILE = old


def test_weighted_valu_model_prices_the_loops_as_they_compile_now(asm):
    """bench.py's valu.weighted_frac = (the period loop priced with this round's measured per-opcode issue costs) /
    (measured time at the held clock).  The priced loop is stored beside the PMC traffic of the same build; here it is
    re-derived from the kernels as they compile now and from profiles/r04/ubench_ops.jsonl: no opcode of the two hot
    loops is priced by assumption, the stored figure is the current one, and the table says what DESIGN.md section 5
    says it says (half-rate multiplies, conversions, SDWA; an SGPR source alone halves the rate; full-rate v_bitop3)."""
    import bench
    import isa_loop_count as I
    import valu_model as V
    ops = V.load_table()
    stored = json.load(open(bench.PMC_TRAFFIC_FILE))
    for mode, key in (("gaussian", "gaussian|100000000|360|all"), ("table", "table|100000000|360|all")):
        variant, periods = I.VARIANTS[mode]
        rows, assumed = V.price(V.loop_lines(asm, variant), ops)
        assert not assumed, assumed
        m = V.model(rows)
        assert m["valu_insts"] == pytest.approx(bench.VALU_INSTS_PER_STEP[mode] * periods)
        assert m["model_clk"] == m["pipe_clk"] > m["sgpr_port_clk"]  # the scalar-operand port is not what binds these loops
        rec = stored[key]["valu"]
        assert rec["model_clk_per_block"] == pytest.approx(m["model_clk"], rel=1e-9) and rec["periods_per_block"] == periods
        assert rec["class_clk_per_block"] == m["class_clk"] == 2.0 * m["valu_insts"] + 2.0 * m["half_rate_insts"]
        assert m["class_clk"] < m["model_clk"] < 1.12 * m["class_clk"]  # the probes' overhead: a few per cent
        assert os.path.exists(os.path.join(ROOT, rec["weights_source"].split(" ")[0]))
    full, half = 2.0, 4.0
    def clk(probe, operands):
        return ops[(probe, operands)]
    assert all(abs(clk(*k) - full) < 0.35 for k in (("v_xor_b32", "vgpr"), ("v_mul_f32", "vgpr"), ("v_add_u32", "vgpr"),
                                                    ("v_fma_f32", "three distinct vgprs"), ("v_lshrrev_b32", "inline"),
                                                    ("v_bitop3_b32", "explicit registers, three banks")))
    assert all(abs(clk(*k) - half) < 0.35 for k in (("v_mad_u64_u32", "vgpr, vcc carry"), ("v_mul_lo_u32", "vgpr"),
                                                    ("v_cvt_f32_i32", "vgpr"), ("v_and_b32_sdwa", "vgpr"), ("v_and_or_b32", "vgpr"),
                                                    ("v_lshlrev_b32", "inline"), ("v_xor_b32", "sgpr"), ("v_mul_f32", "sgpr"),
                                                    ("v_bitop3_b32", "sgpr")))
    # the port: one SGPR reader among VGPR-only instructions costs nothing extra; a simple integer operation next to a
    # multiply costs a multiply's time; a binary32 FMA next to a multiply does not
    assert clk("mix: v_xor sgpr + 3 vgpr-only", "per GROUP of 4: count x 4") < 4 * full + 1.0
    assert clk("mix: v_mad_u64_u32 + v_xor vgpr", "per GROUP of 2: count x 2") > 2 * half - 0.5
    assert clk("order: 32 v_mad_u64_u32 + 32 v_fma_f32", "runs of 1 (alternating)") < 3.3
