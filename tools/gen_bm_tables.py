"""Generates the Box-Muller tables of counter streams v2 and v3.

  radius table  2 sides x 33 octaves x 16 sub-intervals, cubic in x in [-0.5, 0.5):
                r = sqrt(-2 ln U) for U = fl(2 w + 1) / 2^33 (fl = round to binary32), binned
                geometrically from whichever end U is closer to (side 0: by U itself, side 1:
                by 1 - U), so both the log singularity at 0 and the sqrt singularity at 1 sit
                at the small end of an octave ladder.  Octave e = exponent of fl(2 w + 1) =
                0 .. 32 (32 only for the value 2^32 itself), bin value
                u = 2^(e - 33) (1 + (sub + 0.5 + x) / 16).
  trig table    256 x (cos, sin) of 2 pi i / 256.

The same numbers must be compiled into the HIP kernels and into the CPU oracle, so the
script writes the identical .inc file to both trees (tests/test_numerics_cpu.py compares
them).  Coefficients are least-squares fits at 64 Chebyshev nodes in double precision,
rounded once to binary32 and printed as exact hex floats.
"""
import os
import struct

import numpy as np
from numpy.polynomial import chebyshev as Ch
from numpy.polynomial import polynomial as Po

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SUB = 16
OCT = 33


def radius(u, side):
    u = np.minimum(np.asarray(u, dtype=np.float64), 1.0 - 2.0 ** -40)  # octave 32 formally reaches u = 1
    return np.sqrt(-2.0 * np.log(u)) if side == 0 else np.sqrt(-2.0 * np.log1p(-u))


def fit_bin(side, c, j):
    nodes = 0.5 * np.cos(np.pi * (np.arange(64) + 0.5) / 64)  # Chebyshev nodes in [-0.5, 0.5]
    u = 2.0 ** (c - 33) * (1.0 + (j + 0.5 + nodes) / SUB)
    coef = Ch.chebfit(nodes * 2.0, radius(u, side), 3)          # T_k(2x)
    p = Ch.cheb2poly(coef)                                      # power basis in y = 2x
    p = np.array([p[k] * 2.0 ** k for k in range(4)])           # power basis in x
    p32 = p.astype(np.float32)
    xs = np.linspace(-0.5, 0.5, 257)
    us = 2.0 ** (c - 33) * (1.0 + (j + 0.5 + xs) / SUB)
    err = np.abs(Po.polyval(xs, p32.astype(np.float64)) - radius(us, side)).max()
    return p32, err


# ---- counter stream v3 -------------------------------------------------------------------------
# radius: f = fl(w | 1), w the 31-bit distance of the uniform from the nearer end of (0, 1), u = f / 2^32
# in (0, 1/2].  Exponent 127 .. 158: 32 octaves; table index = (bits >> 19) & 511 = (exponent << 4 | sub)
# & 511, i.e. the octaves are stored rotated (exponent 128 first, exponent 127 last) so that the index is
# one bit-field extract; side 1 (distance from 1) follows at +512.  The cubic's argument is
# x' = as_float(0x3f800000 | low 19 mantissa bits) - (1 + 1/32), in [-1/32, 1/32): no shift needed.
OCT3 = 32


def fit_bin3(side, c, j):
    nodes = 0.5 * np.cos(np.pi * (np.arange(64) + 0.5) / 64)  # Chebyshev nodes in [-0.5, 0.5]
    u = 2.0 ** (c - 32) * (1.0 + (j + 0.5 + nodes) / SUB)
    coef = Ch.chebfit(nodes * 2.0, radius(u, side), 3)          # T_k(2x), x in [-0.5, 0.5]
    p = Ch.cheb2poly(coef)
    p = np.array([p[k] * 2.0 ** k for k in range(4)])           # power basis in x
    p = np.array([p[k] * 16.0 ** k for k in range(4)])          # power basis in x' = x / 16
    p32 = p.astype(np.float32)
    xs = np.linspace(-0.5, 0.5, 257)
    us = 2.0 ** (c - 32) * (1.0 + (j + 0.5 + xs) / SUB)
    err = np.abs(Po.polyval(xs / 16.0, p32.astype(np.float64)) - radius(us, side)).max()
    return p32, err


def v3_tables():
    rows, worst = [], 0.0
    for side in (0, 1):
        for t in range(OCT3 * SUB):           # table index within the side
            e = 128 + (t >> 4) if (t >> 4) < 31 else 127   # exponent stored at this index
            c, j = e - 127, t & 15
            # octave 31 (exponent 158) holds the single value f = 2^31 (u = 1/2, sub 0, x' = -1/32): its
            # other sub-intervals are never read and repeat sub-interval 0
            p, err = fit_bin3(side, c, 0 if c == OCT3 - 1 else j)
            worst = max(worst, err)
            rows.append(p)
    # angles at the MIDDLE of each of the 512 sectors: the index is then ub >> 23 with no rounding add,
    # and the residual angle (low 23 bits - 2^22) 2 pi / 2^32 lies in [-pi/512, pi/512)
    trig = [(np.cos(2 * np.pi * (i + 0.5) / 512), np.sin(2 * np.pi * (i + 0.5) / 512)) for i in range(512)]
    return rows, trig, worst


def hexf(v):
    return float(np.float32(v)).hex() + "f"


def main():
    rows, worst = [], 0.0
    for side in (0, 1):
        for c in range(OCT):
            for j in range(SUB):
                # octave 32 holds the single value fl(2 w + 1) = 2^32 (sub-interval 0, x = -0.5);
                # its other sub-intervals are never read and repeat sub-interval 0
                p, err = fit_bin(side, c, 0 if c == OCT - 1 else j)
                worst = max(worst, err)
                rows.append(p)
    trig = [(np.cos(2 * np.pi * i / 256), np.sin(2 * np.pi * i / 256)) for i in range(256)]
    # exact zeros / ones where the true value is one
    out = ["// generated by tools/gen_bm_tables.py -- do not edit; counter stream v2 Box-Muller tables",
           f"// radius: [side][octave][sub] x {{c0, c1, c2, c3}}, r = c0 + x (c1 + x (c2 + x c3)); max fit error {worst:.3g}",
           "#define SMMC_BM_RADIUS_ENTRIES %d" % len(rows),
           "#define SMMC_BM_TRIG_ENTRIES 256",
           "static const float smmc_bm_radius[SMMC_BM_RADIUS_ENTRIES][4] = {"]
    for p in rows:
        out.append("  {" + ", ".join(hexf(v) for v in p) + "},")
    out.append("};")
    out.append("static const float smmc_bm_trig[SMMC_BM_TRIG_ENTRIES][2] = {")
    for c, s in trig:
        c = 0.0 if abs(c) < 1e-15 else c
        s = 0.0 if abs(s) < 1e-15 else s
        out.append("  {" + hexf(c) + ", " + hexf(s) + "},")
    out.append("};")
    rows3, trig3, worst3 = v3_tables()
    out += ["// counter stream v3: radius [side][(exponent << 4 | sub) & 511] x {c0..c3} in x' (see tools/gen_bm_tables.py); "
            f"max fit error {worst3:.3g}",
            "#define SMMC_BM3_RADIUS_ENTRIES %d" % len(rows3),
            "#define SMMC_BM3_TRIG_ENTRIES 512",
            "static const float smmc_bm3_radius[SMMC_BM3_RADIUS_ENTRIES][4] = {"]
    for p in rows3:
        out.append("  {" + ", ".join(hexf(v) for v in p) + "},")
    out.append("};")
    out.append("static const float smmc_bm3_trig[SMMC_BM3_TRIG_ENTRIES][2] = {")
    for c, s in trig3:
        c = 0.0 if abs(c) < 1e-15 else c
        s = 0.0 if abs(s) < 1e-15 else s
        out.append("  {" + hexf(c) + ", " + hexf(s) + "},")
    out.append("};")
    text = "\n".join(out) + "\n"
    for path in (os.path.join(ROOT, "stock_market_monte_carlo_amd", "csrc", "smmc_bm_tables.inc"),
                 os.path.join(ROOT, "oracle", "smmc_bm_tables.inc")):
        with open(path, "w") as f:
            f.write(text)
    print(f"v2: {len(rows)} radius bins, worst abs fit error {worst:.3g}; v3: {len(rows3)} bins, {worst3:.3g}")


if __name__ == "__main__":
    main()
