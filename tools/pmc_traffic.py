"""Writes profiles/pmc_traffic.json: HBM bytes per paths_kernel launch, from rocprofv3 --pmc passes.

bench.py's `roofline.traffic` is read from that file (bench.py cannot profile itself), so the
figure is refreshed by the round's PMC pass instead of living in the source as a literal.

Inputs: one directory per pass (WRITE_SIZE pass, FETCH_SIZE pass), the workload key and the
provenance string stored beside the number.  Corrections follow MI355X_MICROARCH.md, HBM section:
WRITE_SIZE (KiB) is exact for 16-byte-per-lane streaming stores; FETCH_SIZE (KiB) tallies 128-byte
requests at 64 bytes on gfx950 and is doubled.

Each entry also records which build it belongs to: `isa_fingerprint` (sha256 of the profiled kernel's
instruction mnemonics, tools/isa_loop_count.py -- tests/test_measurement_cpu.py fails when the kernel as
it compiles now differs) and `source_sha256` (kernel sources + compiler flags -- bench.py checks it at run
time and reports `traffic: null` for a stale entry).

usage: pmc_traffic.py --key "gaussian|100000000|360|all" --write DIR --fetch DIR --source TEXT
"""
import argparse
import csv
import glob
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def build_identity(mode):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import isa_loop_count as I
    src, kernel, variant = I.TRAFFIC_KERNELS[mode]
    asm = I.emit_asm(f"/tmp/pmc_traffic_{os.getpid()}_{src}.s", src)
    return {"kernel": kernel + variant, "isa_fingerprint": I.fingerprint(asm, variant, kernel),
            "source_sha256": I.source_digest(mode)}


def valu_model_of(mode):
    """The kernel's period loop priced with the round's per-opcode issue costs (tools/valu_model.py): what bench.py's
    valu.weighted_frac is computed from.  Belongs to the build like the traffic figure (same fingerprint)."""
    import valu_model as V
    name = {"gaussian": "gaussian", "table": "table", "ref": "ref"}.get(mode)
    if name is None:
        return None
    m = V.kernel_model(name)
    return {"class_clk_per_block": m["class_clk"], "half_rate_insts_per_block": m["half_rate_insts"],
            "model_clk_per_block": m["model_clk"], "pipe_clk": m["pipe_clk"], "sgpr_readers": m["sgpr_readers"],
            "valu_insts_per_block": m["valu_insts"], "periods_per_block": m["periods_per_block"], "assumed_opcodes": m["assumed"],
            "weights_source": m["weights_source"] + " (tools/ubench_ops.hip, %d waves per SIMD; tools/valu_model.py)" % m["waves_per_simd_of_the_weights"]}


def mean_counter(directory, counter, kernel="paths_kernel"):
    vals = []
    for f in glob.glob(os.path.join(directory, "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                if r["Counter_Name"] == counter and kernel in r["Kernel_Name"]:
                    vals.append(float(r["Counter_Value"]))
    if not vals:
        raise SystemExit(f"no {counter} rows for {kernel} under {directory}")
    return sum(vals) / len(vals), len(vals)


def refresh_valu(out):
    """Re-prices the loops of the entries that belong to the kernels as they compile now (no GPU needed: the ISA and
    the committed issue-cost table): for a change of tools/valu_model.py or of the table, not of a kernel."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import isa_loop_count as I
    table = json.load(open(out))
    for key, rec in table.items():
        mode = I.traffic_kernel_of(key)
        ident = build_identity(mode)
        if ident["isa_fingerprint"] != rec["isa_fingerprint"] or ident["source_sha256"] != rec["source_sha256"]:
            raise SystemExit(f"{key}: the kernel changed since the PMC pass: profile again (tools/profile_r04.sh)")
        rec["valu"] = valu_model_of(mode)
    with open(out, "w") as fh:
        json.dump(table, fh, indent=1, sort_keys=True)
        fh.write("\n")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--key")
    ap.add_argument("--write")
    ap.add_argument("--fetch")
    ap.add_argument("--source")
    ap.add_argument("--refresh-valu", action="store_true")
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "pmc_traffic.json"))
    a = ap.parse_args()
    if a.refresh_valu:
        return refresh_valu(a.out)
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import isa_loop_count as I
    mode = I.traffic_kernel_of(a.key)
    kernel = I.TRAFFIC_KERNELS[mode][1]
    w, nw = mean_counter(a.write, "WRITE_SIZE", kernel)
    f, nf = mean_counter(a.fetch, "FETCH_SIZE", kernel)
    try:
        table = json.load(open(a.out))
    except (OSError, ValueError):
        table = {}
    table[a.key] = {"bytes": (w + 2.0 * f) * 1024.0, "write_size_kib": w, "fetch_size_kib": f,
                    "dispatches": [nw, nf], "source": a.source, **build_identity(mode), "valu": valu_model_of(mode),
                    "correction": "WRITE_SIZE KiB x 1024 + 2 x FETCH_SIZE KiB x 1024 (gfx950 read counter tallies "
                                  "128-byte requests at 64 bytes)"}
    with open(a.out, "w") as fh:
        json.dump(table, fh, indent=1, sort_keys=True)
        fh.write("\n")
    print(a.key, table[a.key])


if __name__ == "__main__":
    main()
