"""Host-side mirror of plonky2/src/iop/challenger.rs (Fiat-Shamir transcript). The sponge
permutation runs on the device (gl_poseidon_permute_batch on a 12-element buffer); the serial
glue — buffers, overwrite-mode duplexing — is host logic exactly like the reference's."""
import numpy as np

from . import _lib

SPONGE_RATE, SPONGE_WIDTH = 8, 12
P = 0xFFFFFFFF00000001


class Challenger:
    def __init__(self, ctx):
        self.ctx = ctx
        self.sponge_state = [0] * SPONGE_WIDTH
        self.input_buffer = []
        self.output_buffer = []

    def _absorb(self, state, blocks):
        """state after absorbing len(blocks)/8 full rate blocks (overwrite mode), one device call"""
        st = np.array(state, dtype=np.uint64)
        inp = np.array(blocks, dtype=np.uint64)
        _lib.call("gl_sponge_absorb", st, inp, inp.size // SPONGE_RATE, self.ctx.ptr)
        return [int(v) for v in st]

    def _permute(self, state):
        return self._absorb(state, state[:SPONGE_RATE])

    def observe_element(self, e):  # challenger.rs:43-53
        self.observe_elements([e])

    def observe_elements(self, es):
        """observe_element for each e (challenger.rs:43-59). Every time the input buffer fills the
        reference duplexes; the outputs of all but the last of those duplexings are discarded by the
        next observe, so all full blocks are absorbed by a single device call."""
        es = [int(e) % P for e in es]
        if not es:
            return
        buf = self.input_buffer + es
        full = len(buf) // SPONGE_RATE * SPONGE_RATE
        self.output_buffer = []
        if full:
            self.sponge_state = self._absorb(self.sponge_state, buf[:full])
            if full == len(buf):  # the last observe triggered the duplexing: its output is live
                self.output_buffer = list(self.sponge_state[:SPONGE_RATE])
        self.input_buffer = buf[full:]

    def observe_extension_elements(self, es):
        self.observe_elements([x for e in es for x in e])

    def observe_hash(self, h):
        self.observe_elements(h)

    def observe_cap(self, cap):  # challenger.rs:81-85
        self.observe_elements([x for h in cap for x in h])

    def get_challenge(self):  # challenger.rs:87-97
        if self.input_buffer or not self.output_buffer:
            self.duplexing()
        return self.output_buffer.pop()

    def get_n_challenges(self, n):
        return [self.get_challenge() for _ in range(n)]

    def get_extension_challenge(self):
        a, b = self.get_n_challenges(2)
        return (a, b)

    def duplexing(self):  # challenger.rs:131-149
        for i, x in enumerate(self.input_buffer):
            self.sponge_state[i] = x
        self.input_buffer = []
        self.sponge_state = self._permute(self.sponge_state)
        self.output_buffer = list(self.sponge_state[:SPONGE_RATE])


def hash_no_pad(ctx, inputs):
    """hash_n_to_hash_no_pad (plonky2/src/hash/hashing.rs:81-108) of a short host vector (public
    inputs, circuit digest parts); the permutation runs on the device."""
    c = Challenger(ctx)
    state = [0] * SPONGE_WIDTH
    inputs = [int(x) % P for x in inputs]
    full = len(inputs) // SPONGE_RATE * SPONGE_RATE
    if full:
        state = c._absorb(state, inputs[:full])
    if full < len(inputs):  # a short last chunk leaves the old lanes in place
        state[: len(inputs) - full] = inputs[full:]
        state = c._permute(state)
    return state[:4]
