"""The fp16 x 2 operand format of the split-operand matrix-core kernels (csrc/stc_x3_frag.h, FmtH2), pinned on the CPU through its numpy
restatement (oracle/operand_format.py): what two fp16 pieces represent, what three piece products lose, and that the power-of-two scales
(tables normalised to [1/2, 1), gradient operands taken into [8, 16) from the launch's maximum) keep that error at the fp32 level for
gradients of any magnitude -- the claims DESIGN.md section 3.3 makes; the kernels themselves are checked against the goldens on the GPU."""
import numpy as np
import pytest

from oracle.operand_format import dot_f16x2, pow2_scale, split_f16x2


def test_two_pieces_represent_22_bits_above_the_subnormal_floor():
    rng = np.random.default_rng(1)
    a = (rng.uniform(-1, 1, 200000) * 2.0 ** rng.integers(-12, 4, 200000)).astype(np.float32)
    h, l = split_f16x2(a)
    err = np.abs(h.astype(np.float64) + l - a)
    assert np.all(err <= np.maximum(2.0 ** -22 * np.abs(a), 2.0 ** -25))          # relative 2^-22, or the absolute floor of fp16 subnormals
    big = np.abs(a) >= 0.25                                                        # low piece still normal: full relative precision
    assert np.all(err[big] <= 2.0 ** -22 * np.abs(a[big]))
    with np.errstate(invalid='ignore'):
        h, l = split_f16x2(np.float32([7.0e4]))                                    # beyond fp16: infinity, i.e. a NaN result -- never a silently wrong number
        assert np.isinf(h[0]) and not np.isfinite(h[0] + l[0])


@pytest.mark.parametrize('t', [0, 4])
def test_pow2_scale_brings_the_maximum_into_its_binade(t):
    for amax in (3e-12, 1.7e-8, 0.49, 0.5, 1.0, 3.3, 65000.0, 2e9):
        s = pow2_scale(amax, t)
        assert np.log2(s) == round(np.log2(s)) and 2.0 ** (t - 1) <= amax * s < 2.0 ** t
    assert pow2_scale(0.0, t) == 1.0 and pow2_scale(float('inf'), t) == 1.0 and pow2_scale(float('nan'), t) == 1.0


@pytest.mark.parametrize('grad_scale', [1e-12, 1e-8, 1e-3, 1.0, 1e4])
def test_three_products_with_scales_stay_at_the_fp32_level(grad_scale):
    """32-term dot products gradient x weight and gradient x activation against float64, max-norm: < 1e-6 whatever the gradient's magnitude
    once the operands carry their scales (an fp32 fmaf chain on the same data: ~2e-7); without the gradient scale small gradients underflow."""
    rng = np.random.default_rng(7)
    M = 2048
    grad = (rng.standard_normal((M, 32)) * grad_scale).astype(np.float32)
    w = (rng.standard_normal((M, 32)) * 0.1).astype(np.float32)
    act = rng.uniform(-1, 1, (M, 32)).astype(np.float32)
    sg, sw = pow2_scale(np.abs(grad).max(), 4), pow2_scale(np.abs(w).max(), 0)
    for other, so in ((w, sw), (act, 1.0)):
        ref = (grad.astype(np.float64) * other).sum(-1)
        got = dot_f16x2(grad, other, sg, so)
        assert np.abs(got - ref).max() / np.abs(ref).max() < 1e-6
    if grad_scale <= 1e-8:
        ref = (grad.astype(np.float64) * w).sum(-1)
        assert np.abs(dot_f16x2(grad, w) - ref).max() / np.abs(ref).max() > 1e-3          # the scale is not optional
