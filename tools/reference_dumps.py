#!/usr/bin/env python3
"""Feed this library the raw-u64 dumps the reference's Rust prover and CUDA test harness exchange, and compare.

The reference's GPU prover is developed against binary dumps of one ed25519 proof (little-endian u64 field elements,
no header). Their writers are the commented blocks in plonky2/src/plonk/circuit_builder.rs:1077-1117 (re-read at
:1119-1186), plonky2/src/fri/oracle.rs:743-753 and plonky2/src/plonk/prover.rs:829-877; their reader is
cuda/test.cu:129-136, 412-428. A Rust user un-comments those blocks, runs the ed25519 example once, and points this tool
at the directory:

    python tools/reference_dumps.py check <dir> [--degree-bits 18] [--public-inputs-hash h0,h1,h2,h3]

Files (P = polynomials, n = 2^degree_bits rows, n_ext = 8 n):
  values.bin                                   [234][n]   wire VALUES handed to PolynomialBatch::from_values (oracle.rs:743-753)
  zs_partial_products.bin                      [20][n]    Z / partial-product values (the same dump with compute_zs_partial_products)
  sigma_vecs.bin                               [80][n]    sigma values (circuit_builder.rs:1097-1099)
  <c>.polynomials.bin / .leaves.bin / .digests.bin / .caps.bin   for <c> in constants_sigmas_commitment,
                                               zs_partial_products_commitment (circuit_builder.rs:1103-1115, prover.rs:832-847);
                                               [P][n] coefficients, [n_ext][P] leaf-major LDE, [2(n_ext-16)][4], [16][4]
  k_is.bin [80]   alphas.bin betas.bin gammas.bin [2]   (prover.rs:849-864)
  quotient_values2.bin                         [2][n_ext] quotient coefficients (cuda/test.cu:553-566) — or the CPU's, same layout
  roots.bin roots2.bin powers.bin inv-powers.bin points.bin z_h_on_coset.*.bin forest.bin   accepted, not needed (tables
                                               the kernels derive themselves; the union-find of witness generation)
Extensions this tool also understands when present (written by the same kind of two-line dump on the Rust side):
  wires_commitment.{polynomials,leaves,digests,caps}.bin   the CPU's commitment of values.bin
  public_inputs.bin [k], public_inputs_hash.bin [4], proof.bin (ProofWithPublicInputs::to_bytes)

What `check` runs, through the reference's OWN FFI symbols where they exist (region contract of cuda/plonky2_gpu.cu):
  1. merkle_tree_from_values on values.bin                        -> vs wires_commitment.*
  2. merkle_tree_from_coeffs on constants_sigmas_commitment.polynomials.bin -> vs its .leaves/.digests/.caps
  3. merkle_tree_from_values on zs_partial_products.bin           -> vs zs_partial_products_commitment.*
  4. compute_quotient_polys (leaf buffers from the files)          -> vs quotient_values2.bin
  5. gl_circuit_create + gl_prove with the ed25519 gate table, constants recovered from the commitment's coefficients,
     sigmas from sigma_vecs.bin, witness = values.bin            -> vs proof.bin
Prints one JSON report; exit code 1 if any comparison that could be made failed. Needs an MI355X (no CPU fallback)."""
import argparse
import ctypes
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
P = 0xFFFFFFFF00000001


def load(d, name, shape=None):
    path = os.path.join(d, name)
    if not os.path.exists(path):
        return None
    a = np.fromfile(path, dtype="<u8")
    return a.reshape(shape) if shape is not None else a


def compare(report, what, got, exp):
    if exp is None:
        report[what] = "no file to compare with"
        return True
    got, exp = np.asarray(got).reshape(-1), np.asarray(exp).reshape(-1)
    if got.size != exp.size:
        report[what] = f"SIZE MISMATCH: device {got.size} elements, file {exp.size}"
        return False
    bad = np.flatnonzero(got != exp)
    if bad.size:
        i = int(bad[0])
        report[what] = f"MISMATCH at element {i}: device {int(got[i]):#x}, file {int(exp[i]):#x} ({bad.size} of {got.size} differ)"
        return False
    report[what] = f"equal ({got.size} elements)"
    return True


def commit_via_reference_symbol(pg, _lib, ctx, data, poly_num, log_n, rate_bits, cap_height, from_coeffs):
    """merkle_tree_from_values / merkle_tree_from_coeffs with the reference's region contract: returns
    (coefficients, leaf-major leaves, digests, cap, the device region) — the leaves stay on the device at region[0..pad)."""
    n, n_ext = 1 << log_n, 1 << (log_n + rate_bits)
    pad = poly_num * n_ext
    nd = 2 * (n_ext - (1 << cap_height))
    ext = pg.DeviceBuffer(ctx, 2 * pad + 4 * nd + (4 << cap_height))
    ext.upload(np.ascontiguousarray(data).reshape(-1), 0)
    if from_coeffs:
        _lib.call("merkle_tree_from_coeffs", ext.ptr, ext.ptr, poly_num, n, log_n, None, None, None, rate_bits, 0, cap_height, pad, ctx.ptr)
        coeffs = None
    else:
        n_inv = ctypes.c_uint64(P - ((P - 1) >> log_n))
        _lib.call("ifft", ext.ptr, poly_num, n, log_n, None, ctypes.addressof(n_inv), ctx.ptr)
        coeffs = ext.download(0, poly_num * n).reshape(poly_num, n)
        _lib.call("merkle_tree_from_coeffs", ext.ptr, ext.ptr, poly_num, n, log_n, None, None, None, rate_bits, 0, cap_height, pad, ctx.ptr)
    ctx.synchronize()
    leaves = ext.download(0, pad).reshape(n_ext, poly_num)
    dig = ext.download(2 * pad, 4 * nd).reshape(-1, 4)
    cap = ext.download(2 * pad + 4 * nd, 4 << cap_height).reshape(-1, 4)
    return coeffs, leaves, dig, cap, ext


def check(d, degree_bits, rate_bits, cap_height, pih):
    import plonky2_gpu_amd as pg
    from plonky2_gpu_amd import _lib
    from plonky2_gpu_amd import ed25519_circuit as ed

    report, ok = {"directory": d}, True
    ctx = pg.Context(0)
    values = load(d, "values.bin")
    if values is None:
        raise SystemExit("values.bin is missing: nothing to do")
    if degree_bits is None:
        degree_bits = (values.size // ed.NUM_WIRES).bit_length() - 1
    n, n_ext = 1 << degree_bits, 1 << (degree_bits + rate_bits)
    report["degree_bits"] = degree_bits
    if values.size != ed.NUM_WIRES * n:
        raise SystemExit(f"values.bin holds {values.size} elements, expected 234 x 2^{degree_bits}")
    files = lambda c, P_: dict(polynomials=load(d, c + ".polynomials.bin"), leaves=load(d, c + ".leaves.bin"),  # noqa: E731
                               digests=load(d, c + ".digests.bin"), caps=load(d, c + ".caps.bin"))

    # 1. wires commitment from values.bin
    w_coeffs, w_leaves, w_dig, w_cap, w_ext = commit_via_reference_symbol(pg, _lib, ctx, values, ed.NUM_WIRES, degree_bits, rate_bits,
                                                                          cap_height, False)
    f = files("wires_commitment", ed.NUM_WIRES)
    for k, got in (("polynomials", w_coeffs), ("leaves", w_leaves), ("digests", w_dig), ("caps", w_cap)):
        ok &= compare(report, f"1 wires_commitment.{k}", got, f[k])
    report["1 wires cap[0]"] = [hex(int(x)) for x in w_cap[0]]

    # 2. constants_sigmas commitment from its coefficients
    f = files("constants_sigmas_commitment", ed.CONSTANTS_SIGMAS_LEAF_LEN)
    cs_leaves_dev = None
    if f["polynomials"] is not None:
        _, cs_leaves, cs_dig, cs_cap, cs_ext = commit_via_reference_symbol(pg, _lib, ctx, f["polynomials"], ed.CONSTANTS_SIGMAS_LEAF_LEN,
                                                                           degree_bits, rate_bits, cap_height, True)
        for k, got in (("leaves", cs_leaves), ("digests", cs_dig), ("caps", cs_cap)):
            ok &= compare(report, f"2 constants_sigmas_commitment.{k}", got, f[k])
        cs_leaves_dev = cs_ext
    else:
        report["2 constants_sigmas_commitment"] = "polynomials.bin missing: skipped"

    # 3. zs_partial_products commitment from its values
    zs_values = load(d, "zs_partial_products.bin")
    f = files("zs_partial_products_commitment", ed.ZS_PARTIAL_PRODUCTS_LEAF_LEN)
    zs_leaves_dev = None
    if zs_values is not None:
        z_coeffs, z_leaves, z_dig, z_cap, z_ext = commit_via_reference_symbol(pg, _lib, ctx, zs_values, ed.ZS_PARTIAL_PRODUCTS_LEAF_LEN,
                                                                              degree_bits, rate_bits, cap_height, False)
        for k, got in (("polynomials", z_coeffs), ("leaves", z_leaves), ("digests", z_dig), ("caps", z_cap)):
            ok &= compare(report, f"3 zs_partial_products_commitment.{k}", got, f[k])
        zs_leaves_dev = z_ext
    elif f["leaves"] is not None:  # the commitment's leaves alone are enough for step 4
        zs_leaves_dev = pg.DeviceBuffer.from_host(ctx, f["leaves"])
        report["3 zs_partial_products_commitment"] = "zs_partial_products.bin missing: leaves taken from the file"
    else:
        report["3 zs_partial_products_commitment"] = "no input: skipped"

    # 4. the reference's compute_quotient_polys symbol
    k_is, alphas, betas, gammas = (load(d, x + ".bin") for x in ("k_is", "alphas", "betas", "gammas"))
    if None not in (cs_leaves_dev, zs_leaves_dev) and all(x is not None for x in (k_is, alphas, betas, gammas)):
        if pih is not None:
            pg.reference_set_public_inputs_hash(pih)
        up = lambda a: pg.DeviceBuffer.from_host(ctx, np.ascontiguousarray(a))  # noqa: E731
        out = pg.reference_compute_quotient_polys(ctx, w_ext, degree_bits, zs_leaves_dev, cs_leaves_dev, up(k_is), up(alphas), up(betas),
                                                  up(gammas))
        got = out.download().reshape(ed.NUM_CHALLENGES, n_ext)
        ok &= compare(report, "4 quotient_values2 (quotient polynomial coefficients)", got, load(d, "quotient_values2.bin"))
        if pih is not None:
            pg.reference_set_public_inputs_hash(None)
    else:
        report["4 compute_quotient_polys"] = "needs the two leaf buffers and k_is / alphas / betas / gammas: skipped"

    # 5. the whole proof
    sig = load(d, "sigma_vecs.bin")
    cs_poly = load(d, "constants_sigmas_commitment.polynomials.bin")
    pis = load(d, "public_inputs.bin")
    if sig is not None and cs_poly is not None and pis is not None:
        import synth_circuit as sc

        # constant VALUES = forward NTT of the first num_constants coefficient vectors of the commitment
        consts = pg.DeviceBuffer.from_host(ctx, cs_poly.reshape(ed.CONSTANTS_SIGMAS_LEAF_LEN, n)[: ed.NUM_CONSTANTS])
        _lib.call("gl_ntt_batch", consts.ptr, ed.NUM_CONSTANTS, degree_bits, n, 0, 0, ctx.ptr)
        circuit = dict(degree_bits=degree_bits, num_wires=ed.NUM_WIRES, num_routed_wires=ed.NUM_ROUTED_WIRES, num_constants=ed.NUM_CONSTANTS,
                       num_challenges=ed.NUM_CHALLENGES, quotient_degree_factor=ed.QUOTIENT_DEGREE_FACTOR,
                       k_is=[int(x) for x in (k_is if k_is is not None else [pow(7, j, P) for j in range(ed.NUM_ROUTED_WIRES)])],
                       constants=consts.download().reshape(ed.NUM_CONSTANTS, n), sigmas=sig.reshape(ed.NUM_ROUTED_WIRES, n),
                       gates=list(ed.GATES), selector_indices=list(ed.SELECTOR_INDICES), groups=list(ed.GROUPS),
                       num_gate_constraints=ed.NUM_GATE_CONSTRAINTS,
                       fri_params=dict(rate_bits=rate_bits, cap_height=cap_height, reduction_arity_bits=sc.constant_arity_bits(degree_bits, rate_bits, cap_height),
                                       proof_of_work_bits=16, num_query_rounds=28), circuit_digest=None)
        fp_file = os.path.join(d, "fri_params.json")  # extension: non-standard FRI parameters of a test circuit
        if os.path.exists(fp_file):
            circuit["fri_params"] = json.load(open(fp_file))
        nc = pg.NativeCircuit(ctx, circuit)
        if load(d, "constants_sigmas_commitment.caps.bin") is not None:
            ok &= compare(report, "5 preprocessed commitment cap (gl_circuit_create)", np.array(nc.constants_sigmas_cap, dtype=np.uint64),
                          load(d, "constants_sigmas_commitment.caps.bin"))
        data = nc.prove_bytes(values.reshape(ed.NUM_WIRES, n), [int(x) for x in pis])
        report["5 proof bytes"] = len(data)
        proof_file = os.path.join(d, "proof.bin")
        if os.path.exists(proof_file):
            exp = open(proof_file, "rb").read()
            same = data == exp
            report["5 proof.bin"] = "equal (%d bytes)" % len(data) if same else "MISMATCH (%d vs %d bytes)" % (len(data), len(exp))
            ok &= same
        else:
            report["5 proof.bin"] = "no file to compare with"
        nc.close()
    else:
        report["5 gl_prove"] = "needs sigma_vecs.bin, constants_sigmas_commitment.polynomials.bin and public_inputs.bin: skipped"
    report["ok"] = bool(ok)
    print(json.dumps(report, indent=1))
    return 0 if ok else 1


def main():
    ap = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    ap.add_argument("command", choices=["check"])
    ap.add_argument("directory")
    ap.add_argument("--degree-bits", type=int, default=None, help="default: from the size of values.bin (234 polynomials)")
    ap.add_argument("--rate-bits", type=int, default=3)
    ap.add_argument("--cap-height", type=int, default=4)
    ap.add_argument("--public-inputs-hash", default=None, help="h0,h1,h2,h3 (hex or decimal); default: public_inputs_hash.bin, else the "
                                                               "hash the reference's kernel hard-wires (cuda/plonky2_gpu.cu:686-689)")
    args = ap.parse_args()
    pih = None
    if args.public_inputs_hash:
        pih = [int(x, 0) for x in args.public_inputs_hash.split(",")]
    else:
        f = load(args.directory, "public_inputs_hash.bin")
        if f is not None:
            pih = [int(x) for x in f]
    sys.exit(check(args.directory, args.degree_bits, args.rate_bits, args.cap_height, pih))


if __name__ == "__main__":
    main()
