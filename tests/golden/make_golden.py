"""Regenerates the fixtures in tests/golden/ (run from the repo root, CPU only).

  libstdcxx_random.json   system libstdc++ <random> outputs + whole reference-style
                          paths, produced by oracle/pin/pin_libstdcxx.cpp (the real
                          library the reference calls at src/simulations.cpp:245-250).
                          This pins the oracle's hand-written mt19937 / Lemire map.
  counter_stream_v2.json  frozen outputs of the oracle's counter-stream engine (C):
  counter_stream_v3.json  regression vectors for the HIP kernels that also travel to
                          the GPU box (v2: round 1's stream, selected with SMMC_FLAG_STREAM_V2;
                          v3: the default).  The stream number sets the Philox counter layout in
                          BOTH modes, so the table-mode values of counter_stream_v3.json are NOT
                          those of round 1 (counter_stream_v2.json holds them, bit-unchanged).
                          Pinned to the oracle itself only.

The reference implementation could not be built here (see oracle/Makefile), so no
fixture in this directory comes from an executed reference.
"""
import json
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402

HERE = os.path.join(ROOT, "tests", "golden")


def load_table():
    vals = []
    with open(os.path.join(ROOT, "data", "SP500_monthly_returns.csv")) as f:
        header = f.readline().strip().split(",")
        col = header.index("returns")
        for line in f:
            cell = line.rstrip("\n").split(",")[col]
            if cell != "":
                vals.append(np.float32(cell))
    return np.array(vals, dtype=np.float32)


def main():
    table = load_table()
    assert table.size == 1127
    # --- libstdc++ pin -----------------------------------------------------
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "pin/pin_libstdcxx"])
    tab_txt = os.path.join(HERE, "_table.txt")
    with open(tab_txt, "w") as f:
        for v in table:
            f.write(f"{float(v):.9g}\n")
    out = subprocess.check_output([os.path.join(ROOT, "oracle", "pin", "pin_libstdcxx"), tab_txt])
    os.remove(tab_txt)
    pin = json.loads(out)
    pin["table_bits_fnv1a"] = int(fnv1a(table.view(np.uint32)))
    with open(os.path.join(HERE, "libstdcxx_random.json"), "w") as f:
        json.dump(pin, f, indent=1)
    # --- counter stream regression vectors (v2 = round 1's Gaussian draw, v3 = the default) -------
    for stream in (2, 3):
        cases = []
        for mode, name in ((O.MODE_TABLE, "table"), (O.MODE_GAUSSIAN, "gaussian")):
            for P in (1, 3, 4, 5, 360, 1000):
                for first in (0, (1 << 32) - 100):
                    p = O.make_params(mode, P, 300, seed=0x5EED0123456789AB, first_path=first, table=table,
                                      n_bins=100, hist_lo=0.0, hist_hi=20000.0, stream=stream)
                    r = O.counter_mc(p)
                    st = r["stats"]
                    cases.append({
                        "mode": name, "n_periods": P, "first_path": first, "n_paths": 300,
                        "seed": 0x5EED0123456789AB, "initial_capital": 1000.0,
                        "gauss_mean": 0.5, "gauss_std": float(np.float32(0.83333)),
                        "n_bins": 100, "hist_lo": 0.0, "hist_hi": 20000.0, "below_threshold": 1000.0,
                        "final_bits": [int(x) for x in r["final"].view(np.uint32)],
                        "hist": [int(x) for x in r["hist"]],
                        "below": int(st.below), "underflow": int(st.underflow), "overflow": int(st.overflow),
                        "sum": st.sum, "sumsq": st.sumsq, "min": float(st.min), "max": float(st.max),
                    })
        with open(os.path.join(HERE, f"counter_stream_v{stream}.json"), "w") as f:
            json.dump({"table_bits_fnv1a": int(fnv1a(table.view(np.uint32))), "stream": stream, "cases": cases}, f)
    print("wrote fixtures")


def fnv1a(words):
    h = 0xCBF29CE484222325
    for w in np.asarray(words, dtype=np.uint32).tobytes():
        h = ((h ^ w) * 0x100000001B3) & 0xFFFFFFFFFFFFFFFF
    return h


if __name__ == "__main__":
    main()
