"""The tile kernel's sRGB encoder, restated in numpy, against the oracle's definition.

The kernel (renderer_amd/csrc/rnd_device.h, srgb_encode) does not search the 255 thresholds: it brings the colour into
[2^-13, largest float below 1] by a median, reads the number of thresholds not above its bucket's lower edge from a
table of one bucket per exponent and eight mantissa bits, and adds one if the colour reaches the next threshold.  The
tables are built as rnd_device_create builds them (renderer_amd/csrc/rnd_abi.cpp).  The oracle counts the thresholds
not above the value (NaN: none).  Both must agree on every float."""
import numpy as np

SHIFT = 15
FIRST = 0x39000000 >> SHIFT
LAST = 0x3F800000 >> SHIFT


def tables():
    k = np.arange(1, 256, dtype=np.float64)
    s = (k - 0.5) / 255.0
    lin = np.where(s <= 0.04045, s / 12.92, ((s + 0.055) / 1.055) ** 2.4)
    thr = np.empty(257, dtype=np.float32)
    thr[0], thr[1:256], thr[256] = -np.inf, lin.astype(np.float32), np.inf
    edges = (np.arange(FIRST, LAST, dtype=np.uint32) << SHIFT).view(np.float32)
    lut = np.searchsorted(thr[1:256], edges, side="right").astype(np.uint32)  # thresholds <= the bucket's lower edge
    return thr, lut, edges


def kernel_encode(c, thr, lut):
    c = np.asarray(c, dtype=np.float32)
    lo, hi = np.float32(1.220703125e-4), np.float32(0.99999994)
    cc = np.where(np.isnan(c), lo, np.minimum(np.maximum(c, lo), hi))  # v_med3_f32: a NaN leaves it as the lower bound
    key = (cc.view(np.uint32) >> SHIFT) - FIRST
    k = lut[key]
    return k + (cc >= thr[k + 1]).astype(np.uint32)


def test_tables_meet_what_the_encoder_assumes():
    thr, lut, edges = tables()
    assert thr[1] > np.float32(1.220703125e-4) and thr[255] < np.float32(0.99999994)
    assert lut[0] == 0 and lut[-1] == 255
    # at most one threshold inside a bucket: the one compare after the table is enough
    nxt = np.append(edges[1:], np.float32(1.0))
    inside = np.searchsorted(thr[1:256], nxt, side="left") - np.searchsorted(thr[1:256], edges, side="right")
    assert inside.max() <= 1


def test_encoder_equals_the_oracle_on_every_kind_of_float(oracle):
    lib = oracle.load()
    thr, lut, edges = tables()
    rng = np.random.RandomState(7)
    ulp = lambda a, d: (a.view(np.uint32).astype(np.int64) + d).astype(np.uint32).view(np.float32)  # noqa: E731
    finite_thr = thr[1:256].copy()
    cases = np.concatenate([
        finite_thr, ulp(finite_thr, -1), ulp(finite_thr, 1), edges, ulp(edges, -1), ulp(edges, 1),
        np.array([0.0, -0.0, 1e-45, -1e-45, 1.17549435e-38, 1.2207031e-4, 1.2207032e-4, 0.99999994, 1.0, 1.0000001, 2.0, 255.0,
                  3.4028235e38, -3.4028235e38, np.inf, -np.inf, np.nan, -1.0, 0.5, 0.0031308], dtype=np.float32),
        rng.randint(0, 2 ** 32, size=200000, dtype=np.uint64).astype(np.uint32).view(np.float32),
        rng.rand(200000).astype(np.float32), (rng.rand(50000) * 0.01).astype(np.float32),
    ])
    got = kernel_encode(cases, thr, lut)
    want = np.array([lib.orc_srgb_encode(float(c)) for c in cases], dtype=np.uint32)
    bad = np.nonzero(got != want)[0]
    assert bad.size == 0, "first difference at %r: kernel rule %d, oracle %d" % (cases[bad[0]], got[bad[0]], want[bad[0]])
