"""CPU checks of the numeric claims the device kernels rely on; constants are parsed from the HIP
sources so that the claims and the code cannot drift apart."""
import os
import re

import numpy as np
from scipy.special import erf

CSRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "dino_tracker_amd", "csrc")


def _src(name):
    with open(os.path.join(CSRC, name)) as fh:
        return fh.read()


def test_gelu_polynomial_error_bound():
    """gelu2() in vit.hip: erf(s / sqrt 2) ~ s P(s^2) on |s| <= c, evaluated in fp32 exactly as the kernel does
    (Horner with fused multiply-adds ~ float32 arithmetic here); claimed |GELU error| < 6e-5 everywhere."""
    src = _src("vit.hip")
    body = src[src.index("__device__ __forceinline__ f2 gelu2(f2 x)"):]
    body = body[:body.index("return __builtin_elementwise_fma(hx, e, hx);")]
    clamp = float(re.search(r"const f2 c = \{([0-9.eE+-]+)f", body).group(1))
    lead = float(re.search(r"f2 p = \{([0-9.eE+-]+)f", body).group(1))
    rest = [float(m) for m in re.findall(r"fma\(p, u, f2\{([0-9.eE+-]+)f", body)]
    coefs = np.array([lead] + rest, dtype=np.float32)  # highest degree first
    assert len(coefs) == 9 and clamp == 4.25
    x = np.linspace(-12, 12, 2_000_001).astype(np.float32)
    s = np.clip(x, -clamp, clamp).astype(np.float32)
    u = (s * s).astype(np.float32)
    p = np.full_like(u, coefs[0])
    for c in coefs[1:]:
        p = (p * u + c).astype(np.float32)
    e = np.clip((s * p).astype(np.float32), -1, 1)
    hx = (x * np.float32(0.5)).astype(np.float32)
    got = (hx * e + hx).astype(np.float32)
    ref = 0.5 * x.astype(np.float64) * (1 + erf(x.astype(np.float64) / np.sqrt(2)))
    assert np.abs(got - ref).max() < 6e-5


def test_split_fp16_products_are_fp32_grade():
    """x = hi + lo with hi = fp16(x), lo = fp16(x - hi): representation error <= max(2^-22 |x|, 2^-25) (the second term
    is half the fp16 subnormal spacing: lo halves of |x| < 1/8 are subnormal); the three products
    xh.wh + xh.wl + xl.wh reproduce x.w to that accuracy (delta_dino.hip, vit.hip patch embedding; the weights carry
    2^8 so that THEIR lo halves stay normal)."""
    rng = np.random.default_rng(0)
    x = (rng.standard_normal(200_000) * rng.choice([1e-2, 1.0, 30.0], 200_000)).astype(np.float32)
    w = (rng.standard_normal(200_000) * 0.05 * 256).astype(np.float32)  # weights carry 2^8
    w = np.where(np.abs(w) < 0.125, np.float32(0.125), w)              # (a weight of |w| < 2^-11 before scaling)
    xh = x.astype(np.float16)
    xl = (x - xh.astype(np.float32)).astype(np.float16)
    wh = w.astype(np.float16)
    wl = (w - wh.astype(np.float32)).astype(np.float16)
    x64, w64 = x.astype(np.float64), w.astype(np.float64)
    rep_x = np.abs(xh.astype(np.float64) + xl.astype(np.float64) - x64)
    rep_w = np.abs(wh.astype(np.float64) + wl.astype(np.float64) - w64)
    assert (rep_x <= np.maximum(2.0 ** -22 * np.abs(x64), 2.0 ** -25)).all()
    assert (rep_w <= 2.0 ** -22 * np.abs(w64)).all()
    three = (xh.astype(np.float64) * wh.astype(np.float64) + xh.astype(np.float64) * wl.astype(np.float64)
             + xl.astype(np.float64) * wh.astype(np.float64))
    exact = x64 * w64
    # representation errors of both factors + the dropped xl.wl term (<= 2^-22 relative)
    tol = np.abs(w64) * np.maximum(2.0 ** -22 * np.abs(x64), 2.0 ** -25) + 2 * 2.0 ** -22 * np.abs(exact)
    assert (np.abs(three - exact) <= tol).all()


def test_candidate_band_covers_fp16_correlation_error():
    """corr_peaks keeps the cells within EPS_PK of the fp16 maximum; the exact arg-max is guaranteed to be among them
    iff EPS_PK >= 2 * sup|rho16 - rho| + accumulation + truncation.  The operand roundings give
    |rho16 - rho| <= (2u + u^2) sum|a_i b_i| <= 2^-10 (u = 2^-11, Cauchy-Schwarz); checked here on random and on
    adversarial (all components rounding the same way) unit vectors."""
    src = _src("track_mfma.hip")
    eps_pk = float(re.search(r"constexpr float EPS_PK = ([0-9.eE+-]+)f;", src).group(1))
    val_bits = int(re.search(r"constexpr int PK_VAL_BITS = (\d+);", src).group(1))
    src_scale = float(re.search(r"constexpr float PK_SRC_SCALE = ([0-9.eE+-]+)f;", src).group(1))
    fscale = float(re.search(r"constexpr float FSCALE = ([0-9.eE+-]+)f;", src).group(1))
    assert src_scale * fscale == 2.0 ** val_bits
    bound = 2.0 ** -10 + 2.0 ** -22            # operand roundings
    slack = 24 * 2.0 ** -24 + 2.0 ** -val_bits   # fp32 accumulation over 24 k-steps + truncation to fixed point
    assert eps_pk >= 2 * (bound + slack)
    rng = np.random.default_rng(1)
    C = 384
    worst = 0.0
    for trial in range(200):
        s = rng.standard_normal(C)
        f = s + rng.standard_normal(C) * rng.choice([0.05, 0.5, 2.0])
        if trial % 4 == 0:  # adversarial: mantissas just above a rounding boundary, same signs
            s = np.abs(s) * (1 + 2.0 ** -12 * 0.999)
            f = np.abs(f) * (1 + 2.0 ** -12 * 0.999)
        a = src_scale * s / np.linalg.norm(s)
        b = fscale * f / np.linalg.norm(f)
        rho = float(a @ b) / 2.0 ** val_bits
        rho16 = float(a.astype(np.float16).astype(np.float64) @ b.astype(np.float16).astype(np.float64)) / 2.0 ** val_bits
        worst = max(worst, abs(rho16 - rho))
    assert worst <= bound


def _certificate_bounds(head, a):
    """(zub_interior, zub_border) exactly as refine_head_kernel computes them (track_mfma.hip, `certified`)."""
    import torch

    from oracle import ref_algo as A

    w1 = A.normalized_conv_weight(head["cnn_refiner.0.weight"])[:, 0].reshape(16, 9).double()
    w2 = A.normalized_conv_weight(head["cnn_refiner.2.weight"])[0].reshape(16, 9).double()
    b1 = head["cnn_refiner.0.bias"].double()
    b2 = head["cnn_refiner.2.bias"].double()[0]
    p1, n1 = w1.clamp(min=0).sum(1), w1.clamp(max=0).sum(1)
    p2, n2 = w2.clamp(min=0).sum(1), w2.clamp(max=0).sum(1)
    zb = b2 + (p2 * torch.relu(b1 + p1 * a)).sum()
    zi = zb + (n2 * torch.relu(b1 + n1 * a)).sum()
    return zi, zb, w1


def test_refiner_upper_bound_of_the_certificate():
    """Tier 1 of the tracker skips the whole-map softmax statistics when  ln Z_ub - z_w < 18.42 - 0.1  with
        Z_ub = n_int exp(zi) + n_brd exp(zb),
        zb = b2 + sum_ch W2+ relu(b1 + P1 a)                 bound of the refiner at cells whose conv2 window touches the
                                                             zero padding (hidden = 0 there, so W2- earns nothing),
        zi = zb + sum_ch W2- relu(b1 + N1 a)                 bound at interior cells,
    (P1 / N1 = positive / negative tap sums of the NORMALISED conv1 kernels, W2+ / W2- of conv2).  Both must hold for
    ANY map with values in [0, a]; round 1 used zi for every cell, which border cells violate.  Checked
    against the oracle's refiner over many seeds, benign and ill-conditioned weights, a -> 0, random / saturated /
    empty / per-channel worst-case maps, border cells included."""
    import torch

    from dino_tracker_amd import synth
    from oracle import ref_algo as A

    torch.manual_seed(0)
    H, W = 12, 14
    interior = torch.zeros(H, W, dtype=torch.bool)
    interior[1:-1, 1:-1] = True
    n_int, n_brd = int(interior.sum()), int((~interior).sum())
    old_form_violations = 0
    for benign in (True, False):
        for seed in range(40):
            head = synth.synth_head_weights(seed, benign=benign)
            for a in (0.002, 0.05, 0.4, 1.0):
                zi, zb, w1 = _certificate_bounds(head, a)
                assert zi <= zb + 1e-12
                maps = [torch.rand(4, 1, H, W) * a, (torch.rand(4, 1, H, W) > 0.5).float() * a,
                        torch.full((1, 1, H, W), a), torch.zeros(1, 1, H, W)]
                # maps shaped to excite each channel: a where the conv1 tap is positive (negative), 0 elsewhere,
                # tiled over the whole map so that border cells see them too
                for ch in range(0, 16, 5):
                    for sign in (1, -1):
                        pat = ((sign * w1[ch].reshape(3, 3)) > 0).float().flip(0, 1) * a
                        maps.append(pat.repeat(H // 3 + 1, W // 3 + 1)[None, None, :H, :W])
                for m in maps:
                    z = A.head_refiner(m.float(), head).double()[:, 0]
                    tol_i, tol_b = 1e-4 * (1 + abs(zi)), 1e-4 * (1 + abs(zb))
                    assert z[:, interior].max() <= zi + tol_i, (benign, seed, a)
                    assert z.max() <= zb + tol_b, (benign, seed, a)
                    old_form_violations += int(z.max() > zi + tol_i)
                    # the quantity the certificate needs: ln sum exp(z) <= ln Z_ub
                    lse = torch.logsumexp(z.reshape(z.shape[0], -1), dim=1).max()
                    lzub = zb + torch.log(n_brd + n_int * torch.exp(zi - zb))
                    assert lse <= lzub + tol_b
    assert old_form_violations > 0  # the sweep does reach the border case the one-bound form missed


def test_single_candidate_shortcut_of_rescore():
    """rescore_kernel (round 5) skips the exact re-scoring of a source whose candidate list holds ONE cell and whose approximate
    maximum exceeds 2 EPS_C: (1) the exact arg-max is among the candidates (the band test above), so with one candidate it IS that
    cell; (2) the exact maximum is positive -- |rho16 - rho| <= EPS_PK / 2 < EPS_C -- so the `best > 0` test the full path makes
    cannot fail.  Checked on the constants of the source and on simulated maps: fp16-perturbed values within the proven bound,
    candidates = cells within EPS_PK of the approximate maximum."""
    src = _src("track_mfma.hip")
    eps_pk = float(re.search(r"constexpr float EPS_PK = ([0-9.eE+-]+)f;", src).group(1))
    eps_c = float(re.search(r"constexpr float EPS_C = ([0-9.eE+-]+)f;", src).group(1))
    assert "rc.ncand == 1 && rc.amax > 2.f * EPS_C" in src
    bound = 2.0 ** -10 + 2.0 ** -22 + 24 * 2.0 ** -24 + 2.0 ** -17    # sup |rho16 - rho| incl. accumulation and truncation
    assert eps_pk >= 2 * bound and 2 * eps_c - bound > 0                # amax > 2 EPS_C  =>  exact maximum > 2 EPS_C - bound > 0
    rng = np.random.default_rng(3)
    singles = 0
    for trial in range(400):
        n = 8107
        rho = np.clip(rng.normal(0.2, 0.15, n), -1, 1)
        k = rng.integers(0, n)
        rho[k] = rho.max() + rng.choice([1e-4, 1e-3, 3e-3, 2e-2])       # peaks from barely to clearly separated
        rho16 = rho + rng.uniform(-bound, bound, n)
        amax = rho16.max()
        cand = np.nonzero(rho16 >= amax - eps_pk)[0]
        assert int(rho.argmax()) in cand
        if len(cand) == 1 and amax > 2 * eps_c:
            singles += 1
            assert int(cand[0]) == int(rho.argmax()) and rho.max() > 0
    assert singles > 50


def test_fp32_tie_band_of_the_arbiter():
    """oracle.ref_algo.fp32_dot_band: 2 sqrt(C) 2^-24.  A float32 evaluation of a length-C dot product of unit vectors deviates
    from float64 by far less than sqrt(C) 2^-24 per evaluation in practice (any summation order numpy / torch uses), so two
    evaluations stay inside the band; the worst case C 2^-24 would be 20 x wider."""
    from oracle import ref_algo as A
    C = 384
    band = A.fp32_dot_band(C)
    assert abs(band - 2 * np.sqrt(C) * 2.0 ** -24) < 1e-12 and band < 0.11 * C * 2.0 ** -24 * 2
    rng = np.random.default_rng(5)
    worst = 0.0
    for _ in range(300):
        a = rng.standard_normal((64, C))
        b = a + 0.3 * rng.standard_normal((64, C))
        a /= np.linalg.norm(a, axis=1, keepdims=True)
        b /= np.linalg.norm(b, axis=1, keepdims=True)
        e64 = np.einsum("ij,ij->i", a, b)
        a32, b32 = a.astype(np.float32), b.astype(np.float32)
        fwd = np.einsum("ij,ij->i", a32, b32).astype(np.float64)
        rev = np.einsum("ij,ij->i", a32[:, ::-1].copy(), b32[:, ::-1].copy()).astype(np.float64)
        seq = np.array([float(np.float32(sum(np.float32(x) * np.float32(y) for x, y in zip(r, s)))) for r, s in zip(a32[:4], b32[:4])])
        exact32 = np.einsum("ij,ij->i", a32.astype(np.float64), b32.astype(np.float64))
        worst = max(worst, np.abs(fwd - exact32).max(), np.abs(rev - exact32).max(), np.abs(seq - exact32[:4]).max())
        worst = max(worst, np.abs(fwd - rev).max())
        del e64
    assert worst < band / 2, (worst, band)
