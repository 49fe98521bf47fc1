"""Host-side mirror of plonky2/src/iop/challenger.rs (Fiat-Shamir transcript). The sponge
permutation runs on the device (gl_poseidon_permute_batch on a 12-element buffer); the serial
glue — buffers, overwrite-mode duplexing — is host logic exactly like the reference's."""
import numpy as np

from . import _lib
from .device import DeviceBuffer

SPONGE_RATE, SPONGE_WIDTH = 8, 12
P = 0xFFFFFFFF00000001


class Challenger:
    def __init__(self, ctx):
        self.ctx = ctx
        self.sponge_state = [0] * SPONGE_WIDTH
        self.input_buffer = []
        self.output_buffer = []
        self._buf = DeviceBuffer(ctx, SPONGE_WIDTH)

    def _permute(self, state):
        self._buf.upload(np.array(state, dtype=np.uint64))
        _lib.call("gl_poseidon_permute_batch", self._buf.ptr, 1, self.ctx.ptr)
        return [int(v) for v in self._buf.download()]

    def observe_element(self, e):  # challenger.rs:43-53
        self.output_buffer = []
        self.input_buffer.append(int(e) % P)
        if len(self.input_buffer) == SPONGE_RATE:
            self.duplexing()

    def observe_elements(self, es):
        for e in es:
            self.observe_element(e)

    def observe_extension_elements(self, es):
        for a, b in es:
            self.observe_element(a)
            self.observe_element(b)

    def observe_hash(self, h):
        self.observe_elements(h)

    def observe_cap(self, cap):  # challenger.rs:81-85
        for h in cap:
            self.observe_elements(h)

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
    for off in range(0, len(inputs), SPONGE_RATE):
        chunk = inputs[off : off + SPONGE_RATE]
        state[: len(chunk)] = chunk
        state = c._permute(state)
    return state[:4]
