"""Prices the period loop of a kernel with this round's measured per-opcode issue costs.

Inputs: the kernel as it compiles now (tools/isa_loop_count.py: the innermost loop's instructions) and
profiles/r04/ubench_ops.jsonl (tools/ubench_ops.hip: clocks per wave-instruction per SIMD by opcode and operand
kind, 8 waves per SIMD).  Two costs per VALU instruction:

  clk      the opcode's measured clocks with VGPR / inline / literal operands -- 2.1-2.2 for the full-rate ones
           (v_xor, v_add, v_mul_f32, v_fma / v_fmac with sources in different banks, v_mov, shifts, v_bitop3 with
           three VGPRs), 4.1 for the half-rate ones (every multiply, conversions, compares, SDWA, VOP3-only integer
           operations, packed and double-precision arithmetic);
  sgpr     1 when the instruction reads an SGPR source.  Alone, such an instruction issues every 4.1 clocks whatever
           its opcode; in a mix it behaves like a PORT that is busy for four clocks while VGPR-only instructions go
           on issuing (ubench_ops' mix probes; and the A/B of round 4 that moved 15 round keys of the Gaussian loop
           from SGPRs to VGPRs: 1.5 %, profiles/r04/ab_operands.txt).

Two prices per block: `class_clk` -- every instruction at its class cost, 2 clocks (full rate) or 4 (half rate), a true lower
bound of the issue time -- and `model_clk` = max( sum of the measured clk, 4.1 x number of SGPR readers ), which carries
the probes' own overhead (2.1-2.3 and 4.07-4.2 per instruction).  bench.py's valu.weighted_frac = class_clk x blocks per
SIMD / (kernel time x the clock the chip held); valu.weighted_frac_measured_costs the same with model_clk.  Opcodes that have no row in the table are listed as `assumed` (priced by encoding class), never silently.

usage: valu_model.py [--json] gaussian|table|gaussian_checked|table_checked|ref
"""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import isa_loop_count as I  # noqa: E402

OPS_TABLE = os.path.join(ROOT, "profiles", "r04", "ubench_ops.jsonl")
WAVES = 8
SGPR_PORT_CLK = 4.1
# the row of the table that prices an opcode when its operands are VGPRs / inline constants / literals
ROW = {
    "v_xor_b32": ("v_xor_b32", "vgpr"), "v_and_b32": ("v_and_b32", "literal"), "v_or_b32": ("v_or_b32", "inline 1.0"),
    "v_add_u32": ("v_add_u32", "vgpr"), "v_sub_u32": ("v_sub_u32", "vgpr"), "v_lshrrev_b32": ("v_lshrrev_b32", "inline"),
    "v_lshlrev_b32": ("v_lshlrev_b32", "inline"), "v_mov_b32": ("v_mov_b32", "vgpr"),
    "v_bitop3_b32": ("v_bitop3_b32", "explicit registers, three banks"), "v_and_or_b32": ("v_and_or_b32", "vgpr"),
    "v_lshl_add_u32": ("v_lshl_add_u32", "vgpr"), "v_add3_u32": ("v_add3_u32", "vgpr"), "v_or3_b32": ("v_or3_b32", "vgpr"),
    "v_xad_u32": ("v_xad_u32", "vgpr"), "v_bfe_u32": ("v_bfe_u32", "inline"), "v_bfe_i32": ("v_bfe_u32", "inline"),
    "v_bfi_b32": ("v_bfi_b32", "vgpr"), "v_alignbit_b32": ("v_alignbit_b32", "vgpr+inline"), "v_perm_b32": ("v_perm_b32", "vgpr"),
    "v_and_b32_sdwa": ("v_and_b32_sdwa", "vgpr"), "v_mad_u64_u32": ("v_mad_u64_u32", "vgpr, vcc carry"),
    "v_mul_lo_u32": ("v_mul_lo_u32", "vgpr"), "v_mul_hi_u32": ("v_mul_hi_u32", "vgpr"), "v_mad_u32_u24": ("v_mad_u32_u24", "vgpr"),
    "v_fma_f32": ("v_fma_f32", "explicit registers, three banks"), "v_fmac_f32": ("v_fmac_f32", "literal"),
    "v_fmamk_f32": ("v_fmamk_f32", "literal"), "v_mul_f32": ("v_mul_f32", "vgpr"), "v_add_f32": ("v_add_f32", "vgpr"),
    "v_max_f32": ("v_max_f32", "vgpr"), "v_cvt_f32_i32": ("v_cvt_f32_i32", "vgpr"), "v_cvt_f32_u32": ("v_cvt_f32_u32", "vgpr"),
    "v_cmp_lt_u32": ("v_cmp_lt_u32", "vgpr -> vcc"), "v_cmp_gt_u32": ("v_cmp_lt_u32", "vgpr -> vcc"),
    "v_cmp_gt_f32": ("v_cmp_lt_u32", "vgpr -> vcc"), "v_cmp_lt_f32": ("v_cmp_lt_u32", "vgpr -> vcc"),
    "v_pk_fma_f32": ("v_pk_fma_f32", "vgpr"), "v_pk_mul_f32": ("v_pk_mul_f32", "vgpr"), "v_fma_f64": ("v_fma_f64", "vgpr"),
    "v_add_f64": ("v_add_f64", "vgpr"), "v_lshlrev_b64": ("v_lshlrev_b64", "inline"), "v_cndmask_b32": ("v_cndmask_b32", "vgpr, sgpr-pair mask"),
}
HALF_BY_ENCODING = 4.1  # an opcode without a row: VOP3-only / multiply / conversion class
FULL_BY_ENCODING = 2.15


def load_table(path=OPS_TABLE):
    t = {}
    for line in open(path):
        if not line.startswith("{"):
            continue
        try:
            r = json.loads(line)
        except ValueError:
            continue
        if r["waves_per_simd"] == WAVES:
            t[(r["probe"], r["operands"])] = r["clk_per_inst"]
    return t


def loop_lines(asm_path, variant, kernel="paths_kernel", which="inner"):
    lines = open(asm_path).read().splitlines()
    sym = f"_ZN4smmc12_GLOBAL__N_1{len(kernel)}{kernel}{variant}"
    beg = [i for i, l in enumerate(lines) if l.startswith(sym)][0]
    fin = [i for i, l in enumerate(lines) if i > beg and "s_endpgm" in l][0]
    body = lines[beg:fin]
    if which == "inner":
        start = [i for i, l in enumerate(body) if "Inner Loop Header: Depth=2" in l][0]
    else:  # the longest loop of the kernel (the reference-stream kernels: the four-output trips)
        heads = [i for i, l in enumerate(body) if "Loop Header" in l]
        ends = [[j for j, l in enumerate(body) if j > h and "s_cbranch" in l][0] for h in heads]
        start = max(zip(heads, ends), key=lambda he: sum(1 for l in body[he[0]:he[1]] if l.strip().startswith("v_")))[0]
    end = [i for i, l in enumerate(body) if i > start and "s_cbranch" in l][0]
    return [l.strip() for l in body[start:end + 1] if l.strip() and l.strip()[0] not in ";."]


def price(lines, table):
    rows, assumed = [], set()
    for l in lines:
        op = l.split()[0]
        if not op.startswith("v_"):
            continue
        base = re.sub(r"_e(32|64)$", "", op)
        operands = l[len(op):].split(",")
        srcs = operands[1:]
        if base in ("v_mad_u64_u32", "v_add_co_u32"):  # second operand is the carry OUT
            srcs = operands[2:]
        reads_sgpr = any(re.search(r"(^|[\s\[-])s\d+|s\[\d+:\d+\]", x.strip()) for x in srcs)
        key = ROW.get(base)
        if key and key in table:
            clk = table[key]
        else:
            vop3_only = base.endswith(("_u64_u32", "3_b32", "3_u32")) or "_sdwa" in base or "cvt" in base or "mul_" in base or "cmp" in base
            clk = HALF_BY_ENCODING if vop3_only else FULL_BY_ENCODING
            assumed.add(base)
        rows.append((base, clk, reads_sgpr))
    return rows, sorted(assumed)


def class_clk(clk):
    """The issue CLASS a measured cost falls in: 2 clocks (a wave64 instruction at full rate), 4 (half rate), or what was
    measured when it is worse than that.  Probe streams measure 2.1-2.3 and 4.07-4.2 -- their own loop overhead and the
    arbitration of eight waves on top of the class cost -- so a loop priced with the measured costs can come out a few
    per cent ABOVE what a kernel needs; priced with the class costs it is a true lower bound of the issue time."""
    return 2.0 if clk < 3.2 else 4.0 if clk < 6.0 else clk


def model(rows):
    total = sum(c for _, c, _ in rows)
    readers = sum(1 for _, _, s in rows if s)
    class_total = sum(class_clk(c) for _, c, _ in rows)
    by = {}
    for op, c, s in rows:
        d = by.setdefault(op, {"count": 0, "clk": c, "sgpr_readers": 0})
        d["count"] += 1
        d["sgpr_readers"] += 1 if s else 0
    return {"valu_insts": len(rows), "pipe_clk": total, "class_clk": max(class_total, readers * 4.0),
            "half_rate_insts": sum(1 for _, c, _ in rows if class_clk(c) >= 4.0),
            "sgpr_readers": readers, "sgpr_port_clk": readers * SGPR_PORT_CLK,
            "model_clk": max(total, readers * SGPR_PORT_CLK), "slots": max(total, readers * SGPR_PORT_CLK) / 2.0, "by_opcode": by}


def kernel_model(mode):
    """mode: a key of isa_loop_count.VARIANTS (paths_kernel), or "ref" (ref_windowed_kernel's first stretch)."""
    table = load_table()
    if mode == "ref":
        asm = I.emit_asm(f"/tmp/valu_model_{os.getpid()}_ref.s", "smmc_ref_kernels.hip")
        lines = loop_lines(asm, "ILi0ELb0E", "ref_windowed_kernel", which="longest")
        per = 4  # outputs per trip
    else:
        variant, per = I.VARIANTS[mode]
        asm = I.emit_asm(f"/tmp/valu_model_{os.getpid()}.s")
        lines = loop_lines(asm, variant)
    rows, assumed = price(lines, table)
    m = model(rows)
    m.update({"kernel": mode, "periods_per_block": per, "assumed": assumed, "weights_source": os.path.relpath(OPS_TABLE, ROOT),
              "waves_per_simd_of_the_weights": WAVES})
    return m


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if a != "--json"]
    m = kernel_model(args[0] if args else "gaussian")
    if "--json" in sys.argv:
        print(json.dumps(m))
    else:
        print(f"{m['kernel']}: {m['valu_insts']} VALU per {m['periods_per_block']} periods ({m['half_rate_insts']} half rate: "
              f"{m['class_clk']:.0f} clk by class cost); measured costs: pipe {m['pipe_clk']:.1f} clk, "
              f"{m['sgpr_readers']} SGPR readers x {SGPR_PORT_CLK} = {m['sgpr_port_clk']:.1f} clk -> model {m['model_clk']:.1f} clk "
              f"= {m['slots']:.1f} slots per block ({m['slots'] / m['periods_per_block']:.2f} per period); assumed: {m['assumed']}")
        for op, d in sorted(m["by_opcode"].items(), key=lambda kv: -kv[1]["count"] * kv[1]["clk"]):
            print(f"   {op:18s} x {d['count']:3d}  {d['clk']:.2f} clk  ({d['sgpr_readers']} read an SGPR)")
