"""Philox4x32-10 (Salmon, Moraes, Dror, Shaw: "Parallel random numbers: as easy as 1, 2, 3", SC'11) in plain Python: the checker of
es_uniform (csrc/step.hip).  Pinned by the Random123 known-answer vectors in tests/test_host_logic.py."""

M0, M1, W0, W1 = 0xD2511F53, 0xCD9E8D57, 0x9E3779B9, 0xBB67AE85


def philox4x32_10(ctr, key):
    c = list(ctr)
    k0, k1 = key
    for _ in range(10):
        p0, p1 = M0 * c[0], M1 * c[2]
        c = [(p1 >> 32) ^ c[1] ^ k0, p1 & 0xFFFFFFFF, (p0 >> 32) ^ c[3] ^ k1, p0 & 0xFFFFFFFF]
        k0, k1 = (k0 + W0) & 0xFFFFFFFF, (k1 + W1) & 0xFFFFFFFF
    return c


def uniform(n, seed, subsequence):
    """What es_uniform(out, n, seed, subsequence, NULL) writes: draw i = word i % 4 of block (i / 4, subsequence), (x >> 8) * 2^-24."""
    out = []
    for q in range((n + 3) // 4):
        w = philox4x32_10([q & 0xFFFFFFFF, q >> 32, subsequence & 0xFFFFFFFF, (subsequence >> 32) & 0xFFFFFFFF],
                          [seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF])
        out += [(x >> 8) / 16777216.0 for x in w]
    return out[:n]
