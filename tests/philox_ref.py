"""numpy restatement of Philox4x32-10 (Salmon, Moraes, Dror, Shaw: "Parallel random numbers: as easy as 1, 2, 3", SC'11),
the counter-based generator include/lfgc.h names for lfgc_lattice_sample_f32.  Test infrastructure."""
import numpy as np


def philox4x32_10(c, k):
    """numpy restatement of Philox4x32-10 (Salmon et al., SC'11; the generator include/lfgc.h names for
    lfgc_lattice_sample_f32): c (n,4) uint32 counters, k (2,) uint32 key -> (n,4) uint32."""
    c = c.astype(np.uint64).copy()
    k0, k1 = np.uint64(k[0]), np.uint64(k[1])
    m32 = np.uint64(0xFFFFFFFF)
    for _ in range(10):
        p0 = np.uint64(0xD2511F53) * c[:, 0]
        p1 = np.uint64(0xCD9E8D57) * c[:, 2]
        n0 = ((p1 >> np.uint64(32)) ^ c[:, 1] ^ k0) & m32
        n2 = ((p0 >> np.uint64(32)) ^ c[:, 3] ^ k1) & m32
        c = np.stack([n0, p1 & m32, n2, p0 & m32], 1)
        k0 = (k0 + np.uint64(0x9E3779B9)) & m32
        k1 = (k1 + np.uint64(0xBB67AE85)) & m32
    return c.astype(np.uint32)
