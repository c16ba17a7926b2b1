#!/usr/bin/env python3
"""Generate the hex-float constants used by the 'crd' (compute-in-double, round once
to float) transcendental spec shared by oracle/sq_oracle.c and csrc/sq_math.h.
Pure integer arithmetic (Machin pi), no libm involved.  Run: python tools/gen_math_consts.py
"""
from fractions import Fraction
import math

def pi_fraction(digits=80):
    # Machin: pi = 16 atan(1/5) - 4 atan(1/239), integer arithmetic
    scale = 10 ** (digits + 10)
    def atan_inv(n):
        s = 0; term = scale // n; k = 0; n2 = n * n
        while term:
            s += term // (2 * k + 1) if k % 2 == 0 else -(term // (2 * k + 1))
            term //= n2; k += 1
        return s
    return Fraction(16 * atan_inv(5) - 4 * atan_inv(239), scale)

def rn_double(fr):
    """Round a Fraction to nearest double (ties-even) exactly."""
    f = float(fr)  # CPython's Fraction->float is correctly rounded (int/int true division)
    return f

PI = pi_fraction()
print("pi         ", rn_double(PI).hex(), repr(rn_double(PI)))
print("2pi        ", rn_double(2*PI).hex())
print("pi/2       ", rn_double(PI/2).hex())
print("2/pi       ", rn_double(2/PI).hex(), repr(rn_double(2/PI)))
# pio2_1: first 33 bits of pi/2 ; pio2_1t = pi/2 - pio2_1
p = PI/2
e = 0  # pi/2 in [1,2) -> 33 significant bits => multiply by 2^32
p1 = Fraction(int(p * 2**32), 2**32)
print("pio2_1     ", float(p1).hex(), repr(float(p1)), "exact" if Fraction(float(p1)) == p1 else "INEXACT")
print("pio2_1t    ", rn_double(p - p1).hex(), repr(rn_double(p - p1)))
fact = 1
for n in range(2, 20):
    fact *= n
    print(f"1/{n}!", rn_double(Fraction(1, fact)).hex())
for k in range(1, 16):
    print(f"1/{2*k+1}", rn_double(Fraction(1, 2*k+1)).hex())
# float32 pi as GHC sees it: pi :: Float = 3.141592653589793238 rounded to float
import struct
def rn_float(fr):
    d = float(fr)
    f = struct.unpack('f', struct.pack('f', d))[0]
    return f
print("pi_f32", rn_float(PI).hex(), repr(rn_float(PI)))

# ---- atan(k/8) table and asin Taylor coefficients ----
def atan_frac(num, den, digits=60):
    scale = 10 ** (digits + 10)
    if num == den:
        return PI / 4
    x = Fraction(num, den)
    s = 0; k = 0
    term = scale * num // den
    n2, d2 = num * num, den * den
    while abs(term) > 0:
        s += term // (2 * k + 1) if k % 2 == 0 else -(term // (2 * k + 1))
        term = term * n2 // d2; k += 1
    return Fraction(s, scale)
print("ATAN_TAB (atan(k/8), k=0..8):")
for k in range(9):
    v = atan_frac(k, 8) if k else Fraction(0)
    print("   ", rn_double(v).hex(), "/* %.17g */" % rn_double(v))
print("ASIN Taylor a_k = C(2k,k)/(4^k (2k+1)), k=1..24:")
from math import comb
for k in range(1, 25):
    v = Fraction(comb(2 * k, k), 4 ** k * (2 * k + 1))
    print("   ", rn_double(v).hex(), "/* k=%d */" % k)
print("pi/4", rn_double(PI/4).hex())
