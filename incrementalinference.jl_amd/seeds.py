"""Counter-based RNG seeding.  Every op (one proposal or one product) carries a 64-bit Philox
key derived on the host from (base seed, clique, pass, step, factor); the kernels derive all
per-particle streams from it (DESIGN.md "RNG").  The same function is implemented on the device
for nbp_program_reseed."""
_M = (1 << 64) - 1


def splitmix64(x):
    x = (x + 0x9E3779B97F4A7C15) & _M
    z = x
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & _M
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & _M
    return z ^ (z >> 31)


def mix_seed(seed, salt):
    return splitmix64((seed ^ splitmix64(salt & _M)) & _M)


def op_seed(base, *ids):
    h = splitmix64(base & _M)
    for i in ids:
        h = splitmix64((h ^ (int(i) & _M)) & _M)
    return h
