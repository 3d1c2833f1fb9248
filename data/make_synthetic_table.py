"""Writes data/SP500_monthly_returns.csv -- a SYNTHETIC stand-in for the file the
reference's python/get_data.py:59-69 downloads (network; unavailable here).

Same shape: an index column plus a column named `returns`, monthly returns in
PERCENT, 1128 rows of which the first is empty (pct_change of the first month is
NaN there) -> 1127 usable values, the table length the reference hard-codes
(src/simulations.cu:123).  Values: N(0.6, 4.3) clipped to (-30, 42), seed 42,
rounded to 6 decimals so the text file alone defines the float32 table.
"""
import os
import numpy as np

def main():
    rng = np.random.Generator(np.random.PCG64(42))
    v = np.clip(rng.normal(0.6, 4.3, 1127), -29.9, 41.9)
    here = os.path.dirname(os.path.abspath(__file__))
    with open(os.path.join(here, "SP500_monthly_returns.csv"), "w") as f:
        f.write("Date,returns\n")
        f.write("1928-01-31,\n")
        y, m = 1928, 2
        for x in v:
            f.write(f"{y:04d}-{m:02d}-28,{x:.6f}\n")
            m += 1
            if m == 13:
                y, m = y + 1, 1

if __name__ == "__main__":
    main()
