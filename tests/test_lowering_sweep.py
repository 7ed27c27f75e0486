"""Seeded random sweeps over the PARAMETERS of the hand lowerings (tla_rust_amd/csrc/spec_raft.h, spec_ssi.h, compiled for the host by
tests/_shim) against the C oracle: the fixed configurations of tests/test_lowering_vs_oracle.py pin the models the bench and the goldens
use; these walk the corners between them — server counts, term / log / message bounds, MaxMsgKeys, invariant masks, the textbook
variant and SYMMETRY of the SI spec — each to a budget of distinct states, comparing counters, verdict, depth and every per-level
count, and that the incrementally maintained fingerprint equals a recomputation on every state.  CPU only."""
import random

import pytest


def raft_config(seed):
    r = random.Random(1000 + seed)
    n = r.choice([2, 2, 3, 3, 5])
    mcr = r.randrange(1, 5)
    max_term = r.choice([2, 2, 3])
    max_log = r.choice([2, 3, 9])
    max_msgs = r.choice([1, 1, 2])
    inv = r.choice([1, 3])
    keys = r.choice([0, 0, 5, 6, 8, 10])
    dev = [n, mcr, max_term, max_log, max_msgs, inv] + ([keys, 0, 0, keys] if keys else [])
    return dev


@pytest.mark.parametrize("seed", range(40))
def test_raft_random_configuration_prefix(oracle, shim, seed):
    dev = raft_config(seed)
    o = oracle.oracle_run("raft", oracle.raft_oracle_params(dev), max_distinct=60000)
    s = shim.shim_run("raft", dev, max_distinct=60000)
    for k in ("distinct", "generated", "depth", "verdict", "levels"):
        assert o[k] == s[k], (k, dev)
    assert s["fp_mismatch"] == 0, dev


def ssi_config(seed):
    r = random.Random(2000 + seed)
    txns, keys = r.choice([(2, 1), (2, 2), (2, 3), (3, 1), (3, 2), (4, 1)])
    inv = r.choice([127, 127, 31, 1])
    textbook = r.choice([0, 0, 1])
    sym = r.choice([0, 0, 1, 2, 3])
    return [txns, keys, inv, 0, textbook, sym]


@pytest.mark.parametrize("seed", range(24))
def test_ssi_random_configuration_prefix(oracle, shim, seed):
    params = ssi_config(seed)
    o = oracle.oracle_run("ssi", params, max_distinct=40000)
    s = shim.shim_run("ssi", params, max_distinct=40000)
    for k in ("distinct", "generated", "depth", "verdict", "levels"):
        assert o[k] == s[k], (k, params)
    assert s["fp_mismatch"] == 0, params
