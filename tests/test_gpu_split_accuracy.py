"""Adversarial accuracy gate of the split-precision arithmetic (three exact bf16 planes per fp32 operand, six partial products on
v_mfma_f32_32x32x16_bf16, fp32 accumulation): is it "not narrower than fp32"?

A bare GEMM out = dA^T X (es_gemm_atb: the weight-gradient kernels' contraction; the register-resident chain kernels use the same
split_pair / six-product arithmetic) is run on the fp32 matrix pipes and in split precision on operands chosen to hurt the split:
magnitudes spread over the fp32 range, sums with 2^20 cancellation, operands next to the largest finite bf16.  Both results are compared
with the fp64 product; the error is measured per output entry relative to sum_m |dA[m][n]| |X[m][k]| (the scale the rounding errors of a
dot product live on), and the split result must stay within 1.5x the fp32-MFMA kernel's own error (+ one ulp of that scale).

The regimes where the split IS narrower than fp32 are pinned down as well (tests below):
  * |x| > 3.3895e38 (the largest finite bf16, 2^127 (2 - 2^-7)): the high plane rounds to infinity.  fp32 itself has only 0.4 % of headroom
    left there; every product with |w| > 1 overflows in fp32 too.
  * |x| < 2^-110: the low planes (2^-8, 2^-16 of the operand) drop below the smallest normal bf16 / fp32 (2^-126) and are flushed: the
    product keeps 8-16 of its 24 bits.  Absolute error < 2^-118 |w| per product, far below the rounding noise of any sum that also
    contains an O(1e-30) term; network activations and weights are O(1e-6 .. 1e2)."""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
LOG = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
M = 1024


def _gemm(X, dA, split):
    from endosurf_amd import _lib
    from endosurf_amd.engine import Engine
    eng = Engine("cuda")
    eng.deterministic = True
    out = torch.zeros(256, 256, device="cuda")
    _lib.check(eng.lib.es_gemm_atb(_lib.ptr(X), _lib.ptr(dA), X.shape[0], _lib.ptr(out), int(split), _lib.ptr(eng.wg_scratch()), eng.st()), "es_gemm_atb")
    torch.cuda.synchronize()
    return out


def _errors(X, dA):
    """(fp32-MFMA error, split error): max over the outputs of |out - exact| / sum_m |dA| |X|, exact = the fp64 product."""
    ref = dA.double().t() @ X.double()
    scale = dA.double().abs().t() @ X.double().abs()
    e = []
    for split in (0, 1):
        out = _gemm(X, dA, split).double()
        assert bool(torch.isfinite(out).all()), "non-finite output"
        e.append(float(((out - ref).abs() / scale.clamp_min(1e-300)).max()))
    return e[0], e[1]


def _log_uniform(rng, shape, lo, hi):
    return (rng.choice([-1.0, 1.0], size=shape) * np.exp2(rng.uniform(lo, hi, size=shape))).astype(np.float32)


CASES = {
    # magnitudes spread over 2^-40 .. 2^40 in both operands (products 2^-80 .. 2^80)
    "log_uniform_wide": lambda rng: (_log_uniform(rng, (M, 256), -40, 40), _log_uniform(rng, (M, 256), -40, 40)),
    # small operands down to the guard 2^-110 against O(1) partners
    "small_down_to_2^-110": lambda rng: (_log_uniform(rng, (M, 256), -110, -60), _log_uniform(rng, (M, 256), -2, 2)),
    # large operands up to the bf16 guard against tiny partners (products O(1))
    "large_up_to_bf16_max": lambda rng: ((rng.choice([-1.0, 1.0], size=(M, 256)) * np.exp2(127.0) * rng.uniform(1.0, 1.98, size=(M, 256))).astype(np.float32),
                                         _log_uniform(rng, (M, 256), -100, -96)),
    # network-like: unit-scale gaussians
    "gaussian": lambda rng: (rng.normal(size=(M, 256)).astype(np.float32), rng.normal(size=(M, 256)).astype(np.float32)),
}


def _cancelling(rng):
    """Rows in pairs (x, a), (x (1 + 2^-20 u), -a): every output is a sum whose terms cancel to ~2^-20 of their size."""
    X = rng.normal(size=(M // 2, 256)).astype(np.float32)
    A = rng.normal(size=(M // 2, 256)).astype(np.float32)
    X2 = (X.astype(np.float64) * (1.0 + 2.0 ** -20 * rng.uniform(-1, 1, size=X.shape))).astype(np.float32)
    Xc, Ac = np.empty((M, 256), np.float32), np.empty((M, 256), np.float32)
    Xc[0::2], Xc[1::2], Ac[0::2], Ac[1::2] = X, X2, A, -A
    return Xc, Ac


CASES["cancellation_2^20"] = _cancelling


@pytest.mark.parametrize("name", sorted(CASES))
def test_split_gemm_is_not_narrower_than_the_fp32_gemm(name):
    rng = np.random.default_rng(sorted(CASES).index(name))
    X, dA = (torch.from_numpy(a).cuda() for a in CASES[name](rng))
    e32, ex3 = _errors(X, dA)
    os.makedirs(LOG, exist_ok=True)
    with open(os.path.join(LOG, f"split_accuracy_{name.replace('^', '')}.json"), "w") as f:
        json.dump(dict(case=name, fp32_mfma_rel_err=e32, split_rel_err=ex3, ratio=ex3 / max(e32, 1e-300)), f)
    ulp = 2.0 ** -24
    assert e32 < 64 * ulp, e32                    # the fp32 kernel itself: a 1024-term dot product
    assert ex3 <= 1.5 * e32 + ulp, (name, e32, ex3)


def test_cancelled_sums_measured_on_the_result():
    """The cancellation case again, measured on the RESULT (forward error of residuals that are 2^-20 of their terms).  This is the one
    place where the split is measurably behind the fp32 MFMA: that instruction is a chain of FUSED multiply-adds (exact products, one
    rounding per accumulation), while a split product drops its three smallest partial products (<= 3 x 2^-24 of the product).  Relative to
    the terms both errors are ~4e-9 (test above: ratio 1.3); relative to a residual 2^-20 of the terms that is 2.5 % against 14 %
    (measured) -- the bound asserted here is 8x.  An fp32 GEMM without fused products (error 2^-24 per product) sits in between."""
    rng = np.random.default_rng(11)
    X, dA = (torch.from_numpy(a).cuda() for a in _cancelling(rng))
    ref = dA.double().t() @ X.double()
    big = ref.abs() > ref.abs().median()
    r32 = float((((_gemm(X, dA, 0).double() - ref).abs() / ref.abs())[big]).median())
    rx3 = float((((_gemm(X, dA, 1).double() - ref).abs() / ref.abs())[big]).median())
    assert rx3 <= 8.0 * r32 + 1e-4, (r32, rx3)


def test_documented_limits_of_the_split():
    """Above the largest finite bf16 the high plane overflows (guard: |x| <= 3.3895e38); below 2^-110 the low planes are flushed and the
    product keeps >= 8 bits: both regimes are outside what the networks produce and are documented, not hidden."""
    rng = np.random.default_rng(3)
    dA = torch.from_numpy(_log_uniform(rng, (M, 256), -125, -124)).cuda()
    X = torch.full((M, 256), 3.4e38, device="cuda")                       # > bf16 max: rounds to +inf in the high plane
    assert bool(torch.isfinite(_gemm(X, dA, 0)).all())                    # fp32 MFMA: products O(1), finite
    assert not bool(torch.isfinite(_gemm(X, dA, 1)).all())                # split: documented overflow of the high plane
    X = torch.from_numpy(_log_uniform(rng, (M, 256), -125, -118)).cuda()   # below the guard: low planes flushed
    dA = torch.from_numpy(_log_uniform(rng, (M, 256), 0, 4)).cuda()
    ref = dA.double().t() @ X.double()
    scale = dA.double().abs().t() @ X.double().abs()
    ex3 = float(((_gemm(X, dA, 1).double() - ref).abs() / scale).max())
    assert ex3 < 2.0 ** -7, ex3                                           # at least the high plane's 8 bits survive


@pytest.mark.parametrize("rows", [64, 128, 320, 1088, 6464 + 64])
def test_split_gemm_row_counts_that_do_not_fill_the_pipeline(rows):
    """The split-precision kernel processes a row chunk in groups of six 16-row stages (two register sets x three LDS buffers) and stages
    the rows past the end of a chunk as zeros: chunks of 4, 8, 20, 68 stages and a problem cut into several tasks must all give the plain sum."""
    rng = np.random.default_rng(rows)
    X = torch.from_numpy(rng.normal(size=(rows, 256)).astype(np.float32)).cuda()
    dA = torch.from_numpy(rng.normal(size=(rows, 256)).astype(np.float32)).cuda()
    ref = dA.double().t() @ X.double()
    scale = dA.double().abs().t() @ X.double().abs()
    for split in (1, 0):
        err = float(((_gemm(X, dA, split).double() - ref).abs() / scale).max())
        assert err < 1e-6, (rows, split, err)
