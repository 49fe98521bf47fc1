"""pyref.py — independent Python big-int model of the hot path (TEST INFRASTRUCTURE ONLY).

Written from the mathematical definitions (not from the C oracle) so the two can check each
other: field ops are `%` on Python ints, the DFT is the O(n^2) definition, Poseidon is the
textbook round function (ARK -> S-box -> MDS, no "fast partial round" tables), the Merkle tree is
built level by level and only then mapped into the reference's digest layout through the closed
form used by `MerkleTree::prove` (plonky2/src/hash/merkle_tree.rs:424-435).

Only tests/ (including tests/golden/gen_golden.py) import this. Small sizes only — it is slow on purpose.
"""
import os
import re

P = 0xFFFFFFFF00000001
GENERATOR = 7  # MULTIPLICATIVE_GROUP_GENERATOR, field/src/goldilocks_field.rs:82
POWER_OF_TWO_GENERATOR = 1753635133440165772  # goldilocks_field.rs:89
TWO_ADICITY = 32


def root_of_unity(n_log):
    """Field::primitive_root_of_unity (field/src/types.rs:268-272)."""
    assert n_log <= TWO_ADICITY
    return pow(POWER_OF_TWO_GENERATOR, 1 << (TWO_ADICITY - n_log), P)


def reverse_bits(x, bits):
    r = 0
    for i in range(bits):
        r |= ((x >> i) & 1) << (bits - 1 - i)
    return r


def log2_strict(n):
    l = n.bit_length() - 1
    assert 1 << l == n
    return l


# ---------------------------------------------------------------- transforms (definitions)

def dft(coeffs):
    """values[k] = sum_j coeffs[j] * w^(jk): what fft() must equal (field/src/fft.rs:286-309)."""
    n = len(coeffs)
    w = root_of_unity(log2_strict(n))
    pw = [pow(w, k, P) for k in range(n)]
    return [sum(c * pw[(j * k) % n] for j, c in enumerate(coeffs)) % P for k in range(n)]


def idft(values):
    n = len(values)
    w_inv = pow(root_of_unity(log2_strict(n)), P - 2, P)
    n_inv = pow(n, P - 2, P)
    pw = [pow(w_inv, k, P) for k in range(n)]
    return [sum(v * pw[(j * k) % n] for k, v in enumerate(values)) * n_inv % P for j in range(n)]


def coset_lde(coeffs, rate_bits, shift=GENERATOR):
    """Evaluations of the polynomial on shift*H_{n<<rate_bits}, natural order
    (field/src/polynomial/mod.rs:205-207, 286-299)."""
    n_ext = len(coeffs) << rate_bits
    scaled = [c * pow(shift, i, P) % P for i, c in enumerate(coeffs)] + [0] * (n_ext - len(coeffs))
    return dft(scaled)


def coset_idft(values, shift=GENERATOR):
    c = idft(values)
    s_inv = pow(shift, P - 2, P)
    return [x * pow(s_inv, i, P) % P for i, x in enumerate(c)]


def fast_ntt(a, inverse=False):
    """O(n log n) recursive radix-2 for fixture sizes the O(n^2) definition cannot reach;
    checked against dft() in tests."""
    n = len(a)
    if n == 1:
        return list(a)
    w = root_of_unity(log2_strict(n))
    if inverse:
        w = pow(w, P - 2, P)

    def rec(x, w):
        m = len(x)
        if m == 1:
            return x
        ev = rec(x[0::2], w * w % P)
        od = rec(x[1::2], w * w % P)
        out = [0] * m
        t = 1
        for k in range(m // 2):
            u = od[k] * t % P
            out[k] = (ev[k] + u) % P
            out[k + m // 2] = (ev[k] - u) % P
            t = t * w % P
        return out

    r = rec([x % P for x in a], w)
    if inverse:
        n_inv = pow(n, P - 2, P)
        r = [x * n_inv % P for x in r]
    return r


def coset_idft_fast(values, shift=GENERATOR):
    """coset_ifft via the O(n log n) transform (for the quotient-polynomial fixtures)."""
    c = fast_ntt(values, inverse=True)
    s_inv = pow(shift, P - 2, P)
    out, r = [], 1
    for x in c:
        out.append(x * r % P)
        r = r * s_inv % P
    return out


# ---------------------------------------------------------------- Poseidon (textbook)

def _load_constants():
    """Parse the committed data header (oracle/poseidon_constants.h)."""
    text = open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "poseidon_constants.h")).read()
    out = {}
    for m in re.finditer(r"(?:static const|POSEIDON_CONST) uint64_t (\w+)\[(\d+)\] = \{(.*?)\};", text, re.S):
        out[m.group(1)] = [int(t.rstrip("UL"), 16) for t in re.findall(r"0x[0-9a-fA-F]+ULL", m.group(3))]
        assert len(out[m.group(1)]) == int(m.group(2))
    return out


_C = _load_constants()
ROUND_CONSTANTS = _C["POSEIDON_ALL_ROUND_CONSTANTS"]
MDS_CIRC = _C["POSEIDON_MDS_CIRC"]
MDS_DIAG = _C["POSEIDON_MDS_DIAG"]
WIDTH, HALF_FULL, N_PARTIAL = 12, 4, 22


def _mds(s):
    # row r of (circulant(MDS_CIRC) + diag(MDS_DIAG)); plonky2/src/hash/poseidon.rs:174-194
    return [
        (sum(s[(i + r) % WIDTH] * MDS_CIRC[i] for i in range(WIDTH)) + s[r] * MDS_DIAG[r]) % P
        for r in range(WIDTH)
    ]


def poseidon(state):
    s = [x % P for x in state]
    rc = 0
    for phase in ("full", "partial", "full"):
        for _ in range(HALF_FULL if phase == "full" else N_PARTIAL):
            s = [(x + ROUND_CONSTANTS[rc * WIDTH + i]) % P for i, x in enumerate(s)]
            if phase == "full":
                s = [pow(x, 7, P) for x in s]
            else:
                s[0] = pow(s[0], 7, P)
            s = _mds(s)
            rc += 1
    return s


def hash_no_pad(inputs):
    """hash_n_to_hash_no_pad (plonky2/src/hash/hashing.rs:81-108): overwrite-mode sponge, rate 8."""
    st = [0] * WIDTH
    for off in range(0, len(inputs), 8):
        chunk = inputs[off : off + 8]
        st[: len(chunk)] = [x % P for x in chunk]
        st = poseidon(st)
    return st[:4]


def hash_or_noop(inputs):
    """Hasher::hash_or_noop (plonky2/src/plonk/config.rs:56-67)."""
    if len(inputs) <= 4:
        return [x % P for x in inputs] + [0] * (4 - len(inputs))
    return hash_no_pad(inputs)


def two_to_one(l, r):
    return poseidon(list(l) + list(r) + [0, 0, 0, 0])[:4]


# ---------------------------------------------------------------- Merkle tree

def merkle_tree(leaves, cap_height):
    """Returns (digests, cap) with digests in the reference layout: inside each cap subtree the
    pair q of layer L sits at hash index 2*((q << (L+1)) + 2^L - 1) + parity
    (merkle_tree.rs:424-435); the subtree root goes to cap."""
    n = len(leaves)
    lg = log2_strict(n)
    assert cap_height <= lg
    n_cap = 1 << cap_height
    sub_leaves = n >> cap_height
    sub_digests = 2 * (sub_leaves - 1)
    digests = [None] * (n_cap * sub_digests)
    cap = []
    for c in range(n_cap):
        layer = [hash_or_noop(l) for l in leaves[c * sub_leaves : (c + 1) * sub_leaves]]
        L = 0
        while len(layer) > 1:
            for idx, d in enumerate(layer):
                q, parity = idx >> 1, idx & 1
                digests[c * sub_digests + 2 * ((q << (L + 1)) + (1 << L) - 1) + parity] = d
            layer = [two_to_one(layer[2 * i], layer[2 * i + 1]) for i in range(len(layer) // 2)]
            L += 1
        cap.append(layer[0])
    assert all(d is not None for d in digests)
    return digests, cap


def commit_from_values(values, rate_bits, cap_height):
    """PolynomialBatch::from_values (plonky2/src/fri/oracle.rs:709-731, 911-977), no blinding.
    values: list of columns. Returns (coeffs columns, leaves rows, digests, cap)."""
    coeffs = [fast_ntt(col, inverse=True) for col in values]
    n_ext = len(values[0]) << rate_bits
    lde = []
    for c in coeffs:
        scaled = [x * pow(GENERATOR, i, P) % P for i, x in enumerate(c)] + [0] * (n_ext - len(c))
        lde.append(fast_ntt(scaled))
    lg = log2_strict(n_ext)
    leaves = [[col[reverse_bits(i, lg)] for col in lde] for i in range(n_ext)]
    digests, cap = merkle_tree(leaves, cap_height)
    return coeffs, leaves, digests, cap


# ---------------------------------------------------------------- synthetic inputs

def splitmix64(seed):
    """SplitMix64 stream reduced into [0, p) by rejection."""
    mask = (1 << 64) - 1
    x = seed & mask
    while True:
        x = (x + 0x9E3779B97F4A7C15) & mask
        z = x
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & mask
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & mask
        z ^= z >> 31
        if z < P:
            yield z
