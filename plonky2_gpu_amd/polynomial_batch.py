"""Host-side mirror of PolynomialBatch (plonky2/src/fri/oracle.rs:112-120, 709-731, 911-1018)."""
import numpy as np

from . import _lib
from .device import DeviceBuffer
from .merkle_tree import MerkleTree

COSET_SHIFT = 7
SALT_SIZE = 4


class PolynomialBatch:
    """A batch of polynomials committed with a Poseidon Merkle cap; everything stays in HBM:
    `d_polynomials` (coefficients, [P][n]), `d_lde` (column-major bit-reversed LDE,
    [P+salt][n_ext]) and `merkle_tree` (digests, cap, optional leaf-major leaves)."""

    def __init__(self, ctx, d_polynomials, d_lde, merkle_tree, n_polys, degree_log, rate_bits, blinding):
        self.ctx = ctx
        self.d_polynomials = d_polynomials
        self.d_lde = d_lde
        self.merkle_tree = merkle_tree
        self.n_polys = n_polys
        self.degree_log = degree_log
        self.rate_bits = rate_bits
        self.blinding = blinding

    # -- constructors -----------------------------------------------------------------------
    @classmethod
    def _commit(cls, ctx, d_poly, from_values, n_polys, log_n, rate_bits, blinding, cap_height, salt, leaf_major):
        n = 1 << log_n
        n_ext = n << rate_bits
        salt_size = SALT_SIZE if blinding else 0
        if cap_height > log_n + rate_bits:
            raise ValueError(f"cap_height={cap_height} should be at most log2(leaves.len())={log_n + rate_bits}")
        cols = n_polys + salt_size
        d_lde = DeviceBuffer(ctx, cols * n_ext)
        if salt_size:
            # the reference draws the salt columns from OsRng (oracle.rs:998-1002); here the caller
            # supplies them so that commitments are reproducible
            s = np.ascontiguousarray(salt, dtype=np.uint64)
            if s.shape != (salt_size, n_ext):
                raise ValueError(f"blinding needs salt of shape ({salt_size}, {n_ext})")
            d_lde.upload(s, offset=n_polys * n_ext)
        d_leaves = DeviceBuffer(ctx, cols * n_ext) if leaf_major else None
        d_dig = DeviceBuffer(ctx, 4 * 2 * (n_ext - (1 << cap_height)))
        d_cap = DeviceBuffer(ctx, 4 << cap_height)
        _lib.call(
            "gl_commit_from_values" if from_values else "gl_commit_from_coeffs",
            d_poly.ptr, n_polys, log_n, rate_bits, cap_height, salt_size, COSET_SHIFT,
            d_lde.ptr, d_leaves.ptr if d_leaves else None, d_dig.ptr, d_cap.ptr, ctx.ptr,
        )
        tree = MerkleTree(ctx, n_ext, cols, cap_height, d_dig, d_cap, d_leaves, d_lde, n_ext)
        return cls(ctx, d_poly, d_lde, tree, n_polys, log_n, rate_bits, blinding)

    @classmethod
    def from_values(cls, ctx, values, rate_bits, blinding, cap_height, timing=None, fft_root_table=None, salt=None,
                    leaf_major=True):
        """PolynomialBatch::from_values (oracle.rs:709-731). values: [n_polys, n] evaluations on H
        (host array, uploaded) or a DeviceBuffer plus shape via from_values_device."""
        v = np.ascontiguousarray(values, dtype=np.uint64)
        n_polys, n = v.shape
        if n & (n - 1):
            raise ValueError("degree must be a power of two")
        d_poly = DeviceBuffer.from_host(ctx, v)
        return cls._commit(ctx, d_poly, True, n_polys, n.bit_length() - 1, rate_bits, blinding, cap_height, salt, leaf_major)

    @classmethod
    def from_coeffs(cls, ctx, polynomials, rate_bits, blinding, cap_height, timing=None, fft_root_table=None, salt=None,
                    leaf_major=True):
        """PolynomialBatch::from_coeffs (oracle.rs:911-977)."""
        c = np.ascontiguousarray(polynomials, dtype=np.uint64)
        n_polys, n = c.shape
        if n & (n - 1):
            raise ValueError("degree must be a power of two")
        d_poly = DeviceBuffer.from_host(ctx, c)
        return cls._commit(ctx, d_poly, False, n_polys, n.bit_length() - 1, rate_bits, blinding, cap_height, salt, leaf_major)

    @classmethod
    def from_values_device(cls, ctx, d_values, n_polys, log_n, rate_bits, blinding, cap_height, salt=None, leaf_major=True):
        """Same as from_values for a trace already resident in HBM (transformed in place)."""
        return cls._commit(ctx, d_values, True, n_polys, log_n, rate_bits, blinding, cap_height, salt, leaf_major)

    @classmethod
    def from_coeffs_device(cls, ctx, d_coeffs, n_polys, log_n, rate_bits, blinding, cap_height, salt=None, leaf_major=True):
        """Same as from_coeffs for coefficients already resident in HBM (kept as `d_polynomials`)."""
        return cls._commit(ctx, d_coeffs, False, n_polys, log_n, rate_bits, blinding, cap_height, salt, leaf_major)

    # -- accessors --------------------------------------------------------------------------
    @property
    def polynomials(self):
        n = 1 << self.degree_log
        return self.d_polynomials.download(0, self.n_polys * n).reshape(self.n_polys, n)

    def lde_column_major(self):
        n_ext = 1 << (self.degree_log + self.rate_bits)
        cols = self.merkle_tree.leaf_len
        return self.d_lde.download(0, cols * n_ext).reshape(cols, n_ext)

    def get_lde_values(self, index, step=1):
        """get_lde_values (oracle.rs:1007-1018): the LDE row at natural point index*step, salt removed."""
        index = index * step
        bits = self.degree_log + self.rate_bits
        rev = int(f"{index:0{bits}b}"[::-1], 2) if bits else 0
        row = self.merkle_tree.get(rev)
        return row[: len(row) - (SALT_SIZE if self.blinding else 0)]

    def eval_polynomials_ext2(self, points):
        """`c.polynomials.par_iter().map(|p| p.to_extension().eval(z))` of OpeningSet::new
        (plonky2/src/plonk/proof.rs:314-319) for each z in `points` (pairs (c0, c1) of F_p[X]/(X^2-7)).
        Returns an array [len(points), n_polys, 2]; the coefficients never leave HBM."""
        pts = np.ascontiguousarray(points, dtype=np.uint64).reshape(-1, 2)
        out = DeviceBuffer(self.ctx, pts.shape[0] * self.n_polys * 2)
        n = 1 << self.degree_log
        _lib.call("gl_eval_polys_ext2", self.d_polynomials.ptr, self.n_polys, self.degree_log, n, pts.ctypes.data, pts.shape[0],
                  out.ptr, self.ctx.ptr)
        res = out.download().reshape(pts.shape[0], self.n_polys, 2)
        out.free()
        return res
