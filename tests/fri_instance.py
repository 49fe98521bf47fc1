"""A small FRI opening instance (two committed oracles, openings at zeta and g*zeta)."""
import random

from oracle import fri_ref, plonk_ref, pyref

P = pyref.P


def make_fri_instance(degree_bits=4, polys_per_oracle=(3, 2), rate_bits=3, cap_height=1, arity_bits=(2, 1), pow_bits=3,
                      num_queries=3, seed=1):
    rng = random.Random(seed)
    n = 1 << degree_bits
    oracles = []
    for k in polys_per_oracle:
        vals = [[rng.randrange(P) for _ in range(n)] for _ in range(k)]
        coeffs, leaves, digests, cap = pyref.commit_from_values(vals, rate_bits, cap_height)
        oracles.append(dict(values=vals, polynomials=coeffs, leaves=leaves, digests=digests, cap=cap))
    zeta = (rng.randrange(P), rng.randrange(P))
    g = pyref.root_of_unity(degree_bits)
    all_polys = [(oi, pi) for oi, o in enumerate(oracles) for pi in range(len(o["polynomials"]))]
    zs_polys = [(1, 0)]
    instance = dict(batches=[(zeta, all_polys), (plonk_ref.ext2_mul((g, 0), zeta), zs_polys)])
    params = dict(rate_bits=rate_bits, cap_height=cap_height, reduction_arity_bits=list(arity_bits), proof_of_work_bits=pow_bits,
                  num_query_rounds=num_queries)
    openings = [[plonk_ref.eval_ext2(oracles[oi]["polynomials"][pi], pt) for oi, pi in polys] for pt, polys in instance["batches"]]
    return oracles, instance, params, openings


def transcript_before_fri(oracles, openings):
    """what the prover's challenger has seen before prove_openings: the caps, then the openings
    (prover.rs:92-100, 132-134, 181-206)."""
    ch = fri_ref.Challenger()
    for o in oracles:
        ch.observe_cap(o["cap"])
    for vals in openings:
        ch.observe_extension_elements(vals)
    return ch
