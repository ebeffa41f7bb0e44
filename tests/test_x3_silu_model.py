"""The split precision's SiLU (csrc/elem16.h x3_silu) restated in NumPy float32: its error class against float64, next to the float32
evaluation of the reference expression v / (1 + exp(-v)) that oracle/nets.py (torch / NumPy) uses.

CPU test: the device function cannot run here, so this pins the ALGORITHM (compensated product -> exp2 -> one Newton step on the
reciprocal) with the hardware steps modelled at their documented 1-ulp accuracy; tests/test_gpu_x3.py measures the device result
through whole layers."""
import numpy as np

F = np.float32


def _fma(a, b, c):   # float32 fma: the double product of two floats is exact, one rounding at the end (double rounding: < 2^-29 relative)
    return (a.astype(np.float64) * b.astype(np.float64) + c.astype(np.float64)).astype(F)


def x3_silu_model(v, rng, ulp_noise=True):
    L2E_HI, L2E_LO, LN2 = F(1.44269502162933349609375), F(1.925963033500011e-08), F(0.693147180559945)
    x = np.minimum(-v, F(88.0)).astype(F)
    t = (x * L2E_HI).astype(F)
    r = (_fma(x, np.full_like(x, L2E_HI), -t) + (x * L2E_LO).astype(F)).astype(F)
    e = np.exp2(t.astype(np.float64)).astype(F)                      # v_exp_f32
    if ulp_noise:
        e = (e * (1 + rng.choice([-1, 0, 1], e.shape) * F(2.0 ** -24))).astype(F)
    d = (F(1) + _fma((e * r).astype(F), np.full_like(x, LN2), e)).astype(F)
    with np.errstate(all="ignore"):
        q = (F(1) / d).astype(F)                                     # v_rcp_f32
        if ulp_noise:
            q = (q * (1 + rng.choice([-1, 0, 1], q.shape) * F(2.0 ** -24))).astype(F)
        q = _fma(_fma(-d, q, np.ones_like(d)), q, q)
        return (v * q).astype(F)


def test_x3_silu_is_in_the_float32_class_of_the_reference_expression():
    rng = np.random.default_rng(0)
    v = np.concatenate([rng.normal(0, 3, 600_000), rng.uniform(-90, 90, 300_000), rng.normal(0, 0.1, 100_000),
                        np.array([0.0, -0.0, 1e-30, -1e-30, 87.9, -87.9, 88.0, -88.0, 1e4, 6.5e4])]).astype(F)
    want = v.astype(np.float64) / (1.0 + np.exp(-v.astype(np.float64)))
    with np.errstate(over="ignore"):
        ref32 = (v / (F(1) + np.exp(-v).astype(F))).astype(F)        # the reference expression evaluated in float32
    ok = np.abs(want) > 1e-30
    for noise in (False, True):
        got = x3_silu_model(v, rng, noise)
        assert np.all(np.isfinite(got))
        rel = np.abs(got[ok] - want[ok]) / np.abs(want[ok])
        rel0 = np.abs(ref32[ok] - want[ok]) / np.abs(want[ok])
        print("x3_silu model (1-ulp exp2 / rcp noise: %s): max rel %.3e mean %.3e | float32 reference expression: max %.3e mean %.3e"
              % (noise, rel.max(), rel.mean(), rel0.max(), rel0.mean()))
        assert rel.max() < 4e-7 and rel.mean() < 6e-8
        assert rel.max() < 2 * rel0.max()
    # below v = -88 the clamp keeps e^x finite: the result is a (signed) zero-sized value, never NaN / inf
    tiny = x3_silu_model(np.array([-88.5, -100.0, -1e4, -6.5e4], F), rng, False)
    assert np.all(np.isfinite(tiny)) and np.all(np.abs(tiny) < 1e-33)
