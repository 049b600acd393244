"""glibc's (fdlibm's) float atanf / atan2f in numpy float32 scalars — the THIRD statement of the algorithm next to the C library itself (which the
oracle calls) and csrc/fdlibm_f32.h (which the device calls).  Used by the literal Python transcription of scan registration
(tests/test_oracle_pipeline.py) so that it reproduces the oracle's intensity / scan line bit for bit, and checked against the C library
in tests/test_fdlibm_f32.py.  Written from the published algorithm (sysdeps/ieee754/flt-32/s_atanf.c, e_atan2f.c; Sun Microsystems 1993,
float conversion by Ian Lance Taylor): break points 7/16, 11/16, 19/16, 39/16; an 11-term odd polynomial as two Horner chains."""
import numpy as np

f32 = np.float32


def _bits(u):
    return np.array([u], dtype=np.uint32).view(np.float32)[0]


def _word(x):
    return int(np.array([x], dtype=np.float32).view(np.int32)[0])


_HI = [_bits(0x3eed6338), _bits(0x3f490fda), _bits(0x3f7b985e), _bits(0x3fc90fda)]
_LO = [_bits(0x31ac3769), _bits(0x33222168), _bits(0x33140fb4), _bits(0x33a22168)]
_AT = [_bits(u) for u in (0x3eaaaaab, 0xbe4ccccd, 0x3e124925, 0xbde38e38, 0x3dba2e6e, 0xbd9d8795, 0x3d886b35, 0xbd6ef16b, 0x3d4bda59, 0xbd15a221, 0x3c8569d7)]
_ONE, _TWO, _ONEP5 = f32(1.0), f32(2.0), f32(1.5)


def atanf(x):
    x = f32(x)
    hx = _word(x)
    ix = hx & 0x7fffffff
    if ix >= 0x4c000000:
        if ix > 0x7f800000:
            return f32(x + x)
        return f32(_HI[3] + _LO[3]) if hx > 0 else f32(f32(-_HI[3]) - _LO[3])
    if ix < 0x3ee00000:
        if ix < 0x31000000:
            return x
        idx = -1
    else:
        x = f32(abs(x))
        if ix < 0x3f980000:
            if ix < 0x3f300000:
                idx, x = 0, f32(f32(f32(_TWO * x) - _ONE) / f32(_TWO + x))
            else:
                idx, x = 1, f32(f32(x - _ONE) / f32(x + _ONE))
        elif ix < 0x401c0000:
            idx, x = 2, f32(f32(x - _ONEP5) / f32(_ONE + f32(_ONEP5 * x)))
        else:
            idx, x = 3, f32(f32(-1.0) / x)
    z = f32(x * x)
    w = f32(z * z)
    a = _AT
    s1 = f32(z * f32(a[0] + f32(w * f32(a[2] + f32(w * f32(a[4] + f32(w * f32(a[6] + f32(w * f32(a[8] + f32(w * a[10])))))))))))
    s2 = f32(w * f32(a[1] + f32(w * f32(a[3] + f32(w * f32(a[5] + f32(w * f32(a[7] + f32(w * a[9])))))))))
    if idx < 0:
        return f32(x - f32(x * f32(s1 + s2)))
    r = f32(_HI[idx] - f32(f32(f32(x * f32(s1 + s2)) - _LO[idx]) - x))
    return f32(-r) if hx < 0 else r


_TINY, _PI_O_4, _PI_O_2, _PI, _PI_LO = f32(1.0e-30), _bits(0x3f490fdb), _bits(0x3fc90fdb), _bits(0x40490fdb), _bits(0xb3bbbd2e)


def atan2f(y, x):
    y, x = f32(y), f32(x)
    hx, hy = _word(x), _word(y)
    ix, iy = hx & 0x7fffffff, hy & 0x7fffffff
    if ix > 0x7f800000 or iy > 0x7f800000:
        return f32(x + y)
    if hx == 0x3f800000:
        return atanf(y)
    m = ((hy >> 31) & 1) | ((hx >> 30) & 2)
    if iy == 0:
        return y if m < 2 else (f32(_PI + _TINY) if m == 2 else f32(f32(-_PI) - _TINY))
    if ix == 0:
        return f32(f32(-_PI_O_2) - _TINY) if hy < 0 else f32(_PI_O_2 + _TINY)
    if ix == 0x7f800000:
        if iy == 0x7f800000:
            return [f32(_PI_O_4 + _TINY), f32(f32(-_PI_O_4) - _TINY), f32(f32(f32(3.0) * _PI_O_4) + _TINY), f32(f32(f32(-3.0) * _PI_O_4) - _TINY)][m]
        return [f32(0.0), f32(-0.0), f32(_PI + _TINY), f32(f32(-_PI) - _TINY)][m]
    if iy == 0x7f800000:
        return f32(f32(-_PI_O_2) - _TINY) if hy < 0 else f32(_PI_O_2 + _TINY)
    k = (iy - ix) >> 23
    if k > 60:
        z = f32(_PI_O_2 + f32(f32(0.5) * _PI_LO))
    elif hx < 0 and k < -60:
        z = f32(0.0)
    else:
        with np.errstate(all="ignore"):
            z = atanf(f32(abs(f32(y / x))))
    if m == 0:
        return z
    if m == 1:
        return f32(-z)
    if m == 2:
        return f32(_PI - f32(z - _PI_LO))
    return f32(f32(z - _PI_LO) - _PI)
