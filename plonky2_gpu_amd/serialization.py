"""Host-side mirror of the proof wire format (plonky2/src/util/serialization.rs:492-700 Write,
:57-348 Read): little-endian canonical u64 per field element, extension elements as their two
base coefficients, hashes as 4 elements, Merkle proofs with a one-byte length prefix. Pure byte
shuffling — no field arithmetic."""
import struct

P = 0xFFFFFFFF00000001


class Buffer:
    def __init__(self, data=b""):
        self.data = bytearray(data)
        self.pos = 0

    # -- Write (serialization.rs:466-700)
    def write_u8(self, x):
        self.data += struct.pack("<B", x)

    def write_field(self, x):
        self.data += struct.pack("<Q", int(x) % P)

    def write_field_vec(self, v):
        for a in v:
            self.write_field(a)

    def write_field_ext_vec(self, v):
        for a, b in v:
            self.write_field(a)
            self.write_field(b)

    def write_hash(self, h):
        self.write_field_vec(h)

    def write_merkle_cap(self, cap):
        for h in cap:
            self.write_hash(h)

    def write_merkle_proof(self, siblings):
        if len(siblings) > 255:
            raise ValueError("Merkle proof length must fit in u8.")
        self.write_u8(len(siblings))
        for h in siblings:
            self.write_hash(h)

    # -- Read (serialization.rs:48-348)
    def remaining(self):
        return len(self.data) - self.pos

    def _take(self, n):
        if self.remaining() < n:
            raise EOFError("IoError: unexpected end of proof bytes")
        out = bytes(self.data[self.pos : self.pos + n])
        self.pos += n
        return out

    def read_u8(self):
        return self._take(1)[0]

    def read_field(self):
        (x,) = struct.unpack("<Q", self._take(8))
        if x >= P:
            raise ValueError("IoError: non-canonical field element")  # F::from_canonical_u64 asserts in debug
        return x

    def read_field_vec(self, n):
        return [self.read_field() for _ in range(n)]

    def read_field_ext_vec(self, n):
        return [(self.read_field(), self.read_field()) for _ in range(n)]

    def read_hash(self):
        return self.read_field_vec(4)

    def read_merkle_cap(self, cap_height):
        return [self.read_hash() for _ in range(1 << cap_height)]

    def read_merkle_proof(self):
        return [self.read_hash() for _ in range(self.read_u8())]


def _num_partial_products(num_routed, qdf):
    return -(-num_routed // qdf) - 1


def proof_to_bytes(proof):
    """write_proof_with_public_inputs (serialization.rs:674-689)"""
    b = Buffer()
    b.write_merkle_cap(proof["wires_cap"])
    b.write_merkle_cap(proof["plonk_zs_partial_products_cap"])
    b.write_merkle_cap(proof["quotient_polys_cap"])
    op = proof["openings"]  # write_opening_set :557-571
    for k in ("constants", "plonk_sigmas", "wires", "plonk_zs", "plonk_zs_next", "partial_products", "quotient_polys"):
        b.write_field_ext_vec(op[k])
    fp = proof["opening_proof"]  # write_fri_proof :641-656
    for cap in fp["commit_phase_merkle_caps"]:
        b.write_merkle_cap(cap)
    for rnd in fp["query_round_proofs"]:  # write_fri_query_rounds :621-638
        for evals, siblings in rnd["initial_trees_proof"]:
            b.write_field_vec(evals)
            b.write_merkle_proof(siblings)
        for step in rnd["steps"]:
            b.write_field_ext_vec(step["evals"])
            b.write_merkle_proof(step["merkle_proof"])
    b.write_field_ext_vec(fp["final_poly"])
    b.write_field(fp["pow_witness"])
    b.write_field_vec(proof["public_inputs"])
    return bytes(b.data)


def proof_from_bytes(data, common):
    """read_proof_with_public_inputs (serialization.rs:306-348). `common` carries the shape fields of
    CommonCircuitData: num_constants, num_routed_wires, num_wires, num_challenges,
    quotient_degree_factor, degree_bits, fri_params (with fri_params["hiding"]: salted leaves)."""
    get = (lambda k: common[k]) if isinstance(common, dict) else (lambda k: getattr(common, k))
    fp = get("fri_params")
    nch, qdf = get("num_challenges"), get("quotient_degree_factor")
    npp = _num_partial_products(get("num_routed_wires"), qdf)
    b = Buffer(data)
    h = fp["cap_height"]
    proof = dict(wires_cap=b.read_merkle_cap(h), plonk_zs_partial_products_cap=b.read_merkle_cap(h), quotient_polys_cap=b.read_merkle_cap(h))
    proof["openings"] = dict(
        constants=b.read_field_ext_vec(get("num_constants")), plonk_sigmas=b.read_field_ext_vec(get("num_routed_wires")),
        wires=b.read_field_ext_vec(get("num_wires")), plonk_zs=b.read_field_ext_vec(nch), plonk_zs_next=b.read_field_ext_vec(nch),
        partial_products=b.read_field_ext_vec(npp * nch), quotient_polys=b.read_field_ext_vec(qdf * nch))
    caps = [b.read_merkle_cap(h) for _ in fp["reduction_arity_bits"]]
    # read_fri_initial_proof (serialization.rs:196-239): with fri_params.hiding the three blinded oracles' leaves end in SALT_SIZE = 4
    # elements (salt_size, plonk/plonk_common.rs:46-52; constants/sigmas are never blinded)
    salt = 4 if fp.get("hiding") else 0
    leaf_lens = [get("num_constants") + get("num_routed_wires"), get("num_wires") + salt, nch * (1 + npp) + salt, nch * qdf + salt]
    rounds = []
    for _ in range(fp["num_query_rounds"]):
        initial = []
        for n in leaf_lens:
            evals = b.read_field_vec(n)
            initial.append((evals, b.read_merkle_proof()))
        steps = []
        for ab in fp["reduction_arity_bits"]:
            evals = b.read_field_ext_vec(1 << ab)
            steps.append(dict(evals=evals, merkle_proof=b.read_merkle_proof()))
        rounds.append(dict(initial_trees_proof=initial, steps=steps))
    final_len = 1 << (get("degree_bits") - sum(fp["reduction_arity_bits"]))  # FriParams::final_poly_len
    final = b.read_field_ext_vec(final_len)
    pow_witness = b.read_field()
    proof["opening_proof"] = dict(commit_phase_merkle_caps=caps, query_round_proofs=rounds, final_poly=final, pow_witness=pow_witness)
    proof["public_inputs"] = b.read_field_vec(b.remaining() // 8)
    return proof
