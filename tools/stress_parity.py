"""One-off randomised stress: the seeded parity tests of tests/test_gpu_parity.py over many more seeds (GPU only).
Found the colliding-completion bug of the blocked QR (exactly dependent column blocks) in round 1."""
import sys, subprocess
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import pytest, types
import test_gpu_parity as t
fails = 0
for seed in range(12, 150):
    try:
        t.test_random_trains_vs_oracle(seed)
    except Exception as e:
        fails += 1; print("TRAIN seed", seed, "FAILED:", str(e)[:300].replace("\n", " "))
for seed in range(6, 60):
    try:
        t.test_random_dense_and_tucker_vs_oracle(seed)
    except Exception as e:
        fails += 1; print("DENSE seed", seed, "FAILED:", str(e)[:300].replace("\n", " "))
print("stress done, failures:", fails)
