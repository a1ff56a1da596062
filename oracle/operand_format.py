"""TEST INFRASTRUCTURE (numpy, CPU): the arithmetic of the fp16 x 2 operand format of csrc/stc_x3_frag.h (FmtH2), restated.

An fp32 operand a is carried as two fp16 pieces, a s = h + l with s a power of two chosen per operand class; a product sum takes the
three piece products l.h' + h.l' + h.h' (each exact in the fp32 accumulator) and is unscaled afterwards.  ``pow2_scale`` is the rule the
kernels use for s (tables: maximum into [1/2, 1); gradient operands: into [2^3, 2^4)).  Used by tests/test_operand_format.py to pin the
format's error claims on the CPU; the product path never imports this module."""
import numpy as np


def pow2_scale(amax: float, t: int) -> float:
    """2^k with amax * 2^k in [2^(t-1), 2^t); 1 for zero / subnormal / non-finite amax; |k| <= 100 (as the device function)."""
    bits = np.float32(amax).view(np.uint32)
    e = int((bits >> 23) & 255)
    if e == 0 or e == 255:
        return 1.0
    k = max(-100, min(100, t - (e - 126)))
    return float(2.0 ** k)


def split_f16x2(a: np.ndarray):
    """(h, l) as float32 arrays: h = fp16(a), l = fp16(a - h), round to nearest even, subnormals kept, overflow to inf."""
    a = np.asarray(a, np.float32)
    with np.errstate(over='ignore'):
        h = a.astype(np.float16).astype(np.float32)
        l = (a - h).astype(np.float32).astype(np.float16).astype(np.float32)
    return h, l


def dot_f16x2(A: np.ndarray, B: np.ndarray, sa: float = 1.0, sb: float = 1.0) -> np.ndarray:
    """sum_k A[..., k] B[..., k] the way the matrix-core kernels take it: operands scaled by the powers of two sa, sb, split, three piece
    products accumulated in fp32 (smallest first), result unscaled."""
    ah, al = split_f16x2(np.asarray(A, np.float32) * np.float32(sa))
    bh, bl = split_f16x2(np.asarray(B, np.float32) * np.float32(sb))
    acc = np.zeros(np.broadcast(ah[..., 0], bh[..., 0]).shape, np.float32)
    for P, Q in ((al, bh), (ah, bl), (ah, bh)):
        for k in range(P.shape[-1]):
            acc = (acc + (P[..., k] * Q[..., k]).astype(np.float32)).astype(np.float32)
    return acc / np.float32(sa * sb)
