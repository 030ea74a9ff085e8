"""Generate tests/golden/fresnel_reference.npy from the reference's own Python restatement of Schlick's
approximation and of the dielectric F0 (reference tests/fresnel_compare.py:23-24 `Shlick`, :38 `F0`; the
C++ they restate is src/ray_program.cc:40-45 `fresnel_schlick` and src/brdf.cc:70-72 `f0_dielectric`).

Columns, one row per angle of incidence 0..90 degrees (double precision):
    0 angle in radians     1 Shlick(F0, angle)     2 exact unpolarised Fresnel air -> glass     3 F0 = ((n1-n2)/(n1+n2))^2, n1 = 1, n2 = 1.5

Run in the authoring container only (it reads /root/reference):
    MPLBACKEND=Agg python tests/golden/make_fresnel_golden.py
The reference script plots at import time; the Agg backend turns mp.show() into a no-op.
"""
import os
import runpy

import numpy as np

os.environ.setdefault("MPLBACKEND", "Agg")
ns = runpy.run_path("/root/reference/tests/fresnel_compare.py")
X = np.asarray(ns["X"], dtype=np.float64)
table = np.stack([X, np.asarray(ns["vf_shlick"](X), dtype=np.float64), np.asarray(ns["Y1"], dtype=np.float64),
                  np.full_like(X, float(ns["F0"]))], axis=1)
assert table.shape == (91, 4)
np.save(os.path.join(os.path.dirname(os.path.abspath(__file__)), "fresnel_reference.npy"), table)
print("wrote", table.shape, table[:2], table[-1])
