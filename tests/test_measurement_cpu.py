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


def test_pmc_traffic_table_names_its_sources():
    import bench
    table = json.load(open(bench.PMC_TRAFFIC_FILE))
    assert "gaussian|100000000|360|all" in table
    for key, rec in table.items():
        mode, n, periods, outputs = key.split("|")
        assert mode in ("gaussian", "table") and outputs in ("all", "final", "stats", "host")
        src = rec["source"].split(" ")[0]
        assert src.startswith("profiles/") and os.path.exists(os.path.join(ROOT, src)), src
        # algorithmic bytes: 4 B per path (+ 8 B per 256-path chunk): traffic must not be below them
        if outputs in ("all", "final"):
            assert rec["bytes"] >= 4.0 * int(n)
            assert rec["bytes"] <= 1.25 * (4.0 + 8.0 / 256.0 * 8) * int(n)  # chunk stats are counted as 32-byte writes
    b, src = bench.pmc_traffic("gaussian", 100_000_000, 360, "all")
    assert b and src
    assert bench.pmc_traffic("gaussian", 12345, 360, "all") == (None, None)
