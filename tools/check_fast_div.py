"""Exhaustive check of the GPU's three-instruction division by sigma = 0.01f (cgic_entropy_dev.h: div_by_sigma001) against the IEEE
quotient, over EVERY fp32 with 2^-100 <= |x| <= 8, both signs (2 x 864 026 625 values; a few seconds on 16 threads).
Last run (round 4): 0 mismatches.  tests/test_oracle_golden.py samples the same function in the CPU suite."""
import ctypes, os, struct, sys
from concurrent.futures import ThreadPoolExecutor
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import cgic_oracle as orc
f = orc.lib().cgic_oracle_check_fast_div
f.restype = ctypes.c_long
f.argtypes = [ctypes.c_uint, ctypes.c_uint, ctypes.c_uint, ctypes.POINTER(ctypes.c_uint)]
bits = lambda v: struct.unpack("<I", struct.pack("<f", v))[0]
lo, hi = bits(2.0 ** -100), bits(8.0) + 1
T = 16
cuts = [lo + (hi - lo) * k // T for k in range(T + 1)]
with ThreadPoolExecutor(T) as ex:
    bad = sum(ex.map(lambda k: f(cuts[k], cuts[k + 1], 1, None), range(T)))
print(f"{hi - lo} magnitudes x 2 signs in [2^-100, 8]: {bad} mismatches")
sys.exit(1 if bad else 0)
