#!/usr/bin/env python3
"""The experiment VERDICT r1 item 4(iii) asked for: a per-handle minimax fit of sin(C atan(B s)) that would
replace the atan -> (co)sine double Horner chain of csrc/vdyn_packed.hpp, for the reference's C = 1.5047.

On t = min(x, 1/x), x = B s, the two branches are different functions: x <= 1 is odd in t (t P(t^2)), x > 1 is
neither odd nor even (Q(t)).  Lawson minimax fits, fp32 Horner evaluation against float64:

    python3 tools/fit_direct_pacejka.py

Result (recorded in DESIGN.md section 8): 2e-7 absolute needs degree 8 in t^2 for x <= 1 (10 instructions) and
degree 10 in t for x > 1 (10 instructions).  The lanes of a wave -- and the two halves of a packed pair -- sit in
different branches, so both chains run (20 + a blend) or every coefficient is selected per lane; the chain in the
kernels costs 14 (degree-7 atan 8, phase 2, cosine 4) at 3e-7.  Not adopted."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from fit_polys import horner32, lawson  # noqa: E402

if __name__ == "__main__":
    C = 1.5047
    t = np.linspace(1e-6, 1.0, 200001)
    t32 = t.astype(np.float32)
    print("x <= 1: sin(C atan t) = t P(t^2)")
    for deg in (5, 6, 7, 8):
        coef, err = lawson(t * t, np.sin(C * np.arctan(t)) / t, deg, weight=t)
        got = (t32 * horner32(coef, (t32 * t32).astype(np.float32))).astype(np.float64)
        e32 = np.max(np.abs(got - np.sin(C * np.arctan(t32.astype(np.float64)))))
        print(f"  degree {deg} in t^2: fit {err:.2e}, fp32 evaluation {e32:.2e} absolute, {deg + 2} instructions")
    print("x > 1: sin(C (pi/2 - atan t)) = Q(t), t = 1/x")
    for deg in (8, 10, 12):
        coef, err = lawson(t, np.sin(C * (np.pi / 2 - np.arctan(t))), deg)
        got = horner32(coef, t32).astype(np.float64)
        e32 = np.max(np.abs(got - np.sin(C * (np.pi / 2 - np.arctan(t32.astype(np.float64))))))
        print(f"  degree {deg} in t: fit {err:.2e}, fp32 evaluation {e32:.2e} absolute, {deg} instructions")
