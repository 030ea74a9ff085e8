"""Generate tests/golden/fibonacci_reference_256.npy from the reference's own Python restatement of
fibonacci2D (reference tests/fibonacci_sequence.py:17-43, double precision, 256 points).

Run in the authoring container only (it reads /root/reference):
    MPLBACKEND=Agg python tests/golden/make_fibonacci_golden.py
The reference script plots at import time; the Agg backend turns plt.show() into a no-op.
"""
import os
import runpy

import numpy as np

os.environ.setdefault("MPLBACKEND", "Agg")
ns = runpy.run_path("/root/reference/tests/fibonacci_sequence.py")
samples = np.asarray(ns["samples"], dtype=np.float64)
assert samples.shape == (256, 2)
np.save(os.path.join(os.path.dirname(os.path.abspath(__file__)), "fibonacci_reference_256.npy"), samples)
print("wrote", samples.shape, samples[:3])
