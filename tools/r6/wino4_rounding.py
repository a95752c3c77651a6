"""tools/r6/wino4_rounding.py -- fp32 rounding of Winograd F(4x4, 3x3) / F(2x2, 3x3) / the direct sum against float64, in numpy (CPU).

A transcription of the arithmetic of csrc/winograd4.h (transforms and the channel sum rounded to fp32 after every step), made before the
kernel was written: is the form's rounding inside north_star's budget?   python tools/r6/wino4_rounding.py
"""
import numpy as np

BT = np.array([[4, 0, -5, 0, 1, 0], [0, -4, -4, 1, 1, 0], [0, 4, -4, -1, 1, 0], [0, -2, -1, 2, 1, 0], [0, 2, -1, -2, 1, 0],
               [0, 4, 0, -5, 0, 1]], dtype=np.float64)
G = np.array([[1 / 4, 0, 0], [-1 / 6, -1 / 6, -1 / 6], [-1 / 6, 1 / 6, -1 / 6], [1 / 24, 1 / 12, 1 / 6], [1 / 24, -1 / 12, 1 / 6], [0, 0, 1]])
AT = np.array([[1, 1, 1, 1, 1, 0], [0, 1, -1, 2, -2, 0], [0, 1, 1, 4, 4, 0], [0, 1, -1, 8, -8, 1]], dtype=np.float64)
BT2 = np.array([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=np.float64)
G2 = np.array([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]])
AT2 = np.array([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=np.float64)


def winograd(x, w, AT_, G_, BT_, m, dt):
    """x [C, H, W], w [K, C, 3, 3] -> [K, H - 2, W - 2] with m x m output tiles; every intermediate rounded to dt."""
    C, H, W = x.shape
    K = w.shape[0]
    AT_, G_, BT_ = AT_.astype(dt), G_.astype(dt), BT_.astype(dt)
    U = np.einsum('ia,kcab,jb->kcij', G_, w.astype(dt), G_).astype(dt)
    out = np.zeros((K, H - 2, W - 2), dtype=dt)
    p = m + 2
    for ty in range(0, H - 2, m):
        for tx in range(0, W - 2, m):
            d = x[:, ty:ty + p, tx:tx + p].astype(dt)
            V = np.einsum('cib,jb->cij', np.einsum('ia,cab->cib', BT_, d).astype(dt), BT_).astype(dt)
            M = np.zeros((K, p, p), dtype=dt)
            for c in range(C):                    # an fmaf chain over the reduction channels, like the MFMA
                M = (M + U[:, c] * V[c][None]).astype(dt)
            out[:, ty:ty + m, tx:tx + m] = np.einsum('kib,jb->kij', np.einsum('ia,kab->kib', AT_, M).astype(dt), AT_).astype(dt)
    return out


def direct(x, w, dt):
    C, H, W = x.shape
    out = np.zeros((w.shape[0], H - 2, W - 2), dtype=dt)
    for c in range(C):
        for a in range(3):
            for b in range(3):
                out = (out + w[:, c, a, b].astype(dt)[:, None, None] * x[c, a:a + H - 2, b:b + W - 2].astype(dt)[None]).astype(dt)
    return out


if __name__ == '__main__':
    rng = np.random.default_rng(0)
    C, K, H = 64, 16, 34
    w = (rng.standard_normal((K, C, 3, 3)) / np.sqrt(9 * C)).astype(np.float32)
    for name, x in (("ReLU activations", np.maximum(rng.standard_normal((C, H, H)), 0).astype(np.float32)),
                    ("wide per-channel range", (rng.standard_normal((C, H, H)) * np.exp(rng.standard_normal((C, 1, 1)) * 2)).astype(np.float32))):
        ref = direct(x, w, np.float64)
        sc = np.abs(ref).max()
        assert np.abs(winograd(x, w, AT, G, BT, 4, np.float64) - ref).max() < 1e-12 * sc
        for form, got in (("direct fp32 sum", direct(x, w, np.float32)), ("F(2x2, 3x3) fp32", winograd(x, w, AT2, G2, BT2, 2, np.float32)),
                          ("F(4x4, 3x3) fp32", winograd(x, w, AT, G, BT, 4, np.float32))):
            e = got.astype(np.float64) - ref
            print("%-24s %-18s max %.2e  rms %.2e   (of max |result|)" % (name, form, np.abs(e).max() / sc, np.sqrt((e ** 2).mean()) / sc))
