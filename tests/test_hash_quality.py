"""hmum (tla_rust_amd/csrc/mc_common.h) — the per-element hash of raft's ADDITIVE fingerprint — as a hash: the engine trusts
that sums of hmum terms over the changed elements of a state identify the state as well as 64 random bits would.  The inputs
are what the lowering feeds it: narrow, structured words (small fields, high half often constant) under a fixed salt."""
import ctypes as C

import numpy as np

import helpers

SALT_M = 0x8F1BBCDC8F1BBCDC


def hmum(x, salt=SALT_M):
    lib = helpers.shim_lib()
    x = np.ascontiguousarray(x, dtype=np.uint64)
    out = np.empty_like(x)
    lib.shim_hmum.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p]
    lib.shim_hmum.restype = None
    lib.shim_hmum(x.ctypes.data, x.size, salt, out.ctypes.data)
    return out


def collisions(a):
    a = np.sort(a)
    return int(np.count_nonzero(a[1:] == a[:-1]))


def test_avalanche_on_narrow_and_wide_words():
    rng = np.random.default_rng(1)
    for mask, nbits in ((0xFFFFFF, 24), (0xFFFFFFFFFF, 40), (0xFFFFFFFFFFFFFFFF, 64)):
        x = rng.integers(0, 1 << 63, 40000, dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, 40000, dtype=np.uint64)
        x &= np.uint64(mask)
        h = hmum(x)
        for i in range(nbits):
            d = h ^ hmum(x ^ np.uint64(1 << i))
            p = np.array([np.count_nonzero(d >> np.uint64(j) & np.uint64(1)) for j in range(64)]) / x.size
            assert np.all(np.abs(p - 0.5) < 0.02), (mask, i, p)  # 40 000 samples: sigma = 0.0025


def test_no_collisions_and_birthday_rate_halves_on_structured_words():
    i, j, k = np.meshgrid(np.arange(1 << 13, dtype=np.uint64), np.arange(64, dtype=np.uint64), np.arange(16, dtype=np.uint64), indexing="ij")
    x = (i | (j << np.uint64(13)) | (k << np.uint64(29))).ravel()  # 2^23 words of three small fields, one in the high half
    h = hmum(x)
    assert collisions(h) == 0
    expect = x.size ** 2 / 2 / 2 ** 32
    for half in (h & np.uint64(0xFFFFFFFF), h >> np.uint64(32)):
        assert 0.9 * expect < collisions(half) < 1.1 * expect
    # the seen-set's bucket index (bits 3..) is uniform
    cnt = np.bincount(((h >> np.uint64(3)) & np.uint64(0xFFFF)).astype(np.int64), minlength=1 << 16)
    chi2 = float(np.sum((cnt - x.size / 65536) ** 2 / (x.size / 65536))) / 65535
    assert 0.97 < chi2 < 1.03


def test_sums_and_differences_of_terms_do_not_collide():
    """the additive use: H(a) + H(b) over pairs of structured words, multiplicities 1..3 (the message bag), differences"""
    for shift in (0, 2, 13, 32, 40):
        h = hmum(np.arange(4096, dtype=np.uint64) << np.uint64(shift))
        iu = np.triu_indices(4096)
        sums = h[iu[0]] + h[iu[1]]
        assert collisions(sums) == 0
        expect = sums.size ** 2 / 2 / 2 ** 32
        assert 0.9 * expect < collisions(sums >> np.uint64(32)) < 1.1 * expect
        assert 0.9 * expect < collisions(sums & np.uint64(0xFFFFFFFF)) < 1.1 * expect
        hh = h[:1500]
        diff = (hh[:, None] - hh[None, :])[~np.eye(1500, dtype=bool)]
        assert collisions(diff) == 0
        for c in (2, 3):  # (count + 1) * H(key)
            w = (np.uint64(c) * hh[:, None] + hh[None, :])[~np.eye(1500, dtype=bool)]
            assert collisions(np.concatenate([w, sums[:2_000_000]])) == 0
