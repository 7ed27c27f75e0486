"""Host-side pieces of bench.py that run without a GPU: the workload table against the golden files, the kernel-source stamp of the
counter summary the contract line quotes, the N-rank line's HBM figure."""
import importlib.util
import json
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent


def _bench():
    spec = importlib.util.spec_from_file_location("bench_module", ROOT / "bench.py")
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_every_workload_names_a_golden_case():
    b = _bench()
    for name, w in b.WORKLOADS.items():
        g = json.loads((ROOT / "tests" / "golden" / w.get("golden_file", "raft_levels.json")).read_text())
        case = next(c for c in g["cases"] if c["name"] == w["golden"])
        assert case["distinct"] > 0 and len(case["levels"]) == case["depth"], name
        if "max_levels" in w:
            assert case["depth"] == w["max_levels"] and case["verdict"] == w["verdict"] == "budget", name


def test_the_quoted_counter_summary_is_stamped_with_the_kernel_sources_of_this_tree():
    """bench.py reads HBM traffic from the newest profiles/r*_pmc.json of the workload's SPEC and refuses it (traffic = null) unless it was
    collected on the kernel sources that are timed: a commit that touches the kernels without re-collecting the counters shows up here
    (as a SKIP naming the stale summary), not only in the driver's line"""
    import pytest
    b = _bench()
    stale = []
    for spec, stag, kn in (("raft", "SpecRaft<3>", ("k_expand_family",)), ("raft", "SpecRaft<5>", ("k_expand_family",)), ("ssi", "SpecSsi", ("k_expand_pairs",))):
        src, k = b.pmc_for(spec, stag, kn)
        if k is None:   # (between a kernel change and its next profile this is a state of work, not an error: the line then says traffic = null)
            stale.append(f"{stag}: {src}")
        else:
            assert k["launches"] > 0 and "FETCH_SIZE" in k and "WRITE_SIZE" in k, (stag, src)
    if stale:
        pytest.skip("; ".join(stale) + ": re-run profiles/collect.sh")


def test_n_rank_roofline_object():
    b = _bench()
    import tla_rust_amd as amd
    W = amd.state_bytes("raft", b.WORKLOADS["t3"]["params"])
    r = b.dist_roofline(W, 525782408, 6708500293, 8, 0.025)
    assert W == 136 and r["bound"] == "hbm" and r["peak"] == 8000.0 and r["traffic"] is None
    assert abs(r["alg_bytes_per_step_per_gpu"] - (2 * 136 * 525782408 + 8 * 6708500293) / 8) < 1 and abs(r["frac"] - r["achieved"] / 8000.0) < 1e-12
    assert 0.12 < r["frac"] < 0.13


def test_the_eight_rank_forms_of_configs_4_and_5_go_one_level_deeper():
    """`bench.py --gpus 8 --workload raft5 | ssi4x3` searches one BFS level beyond the one-GPU budget (eight
    devices hold it): config 5 against the oracle's 11-level golden, config 4 gated on the oracle's 18 levels with level 19 reported"""
    b = _bench()
    ssi, raft = b.DEEP["ssi4x3"], b.DEEP["raft5"]
    g = json.loads((ROOT / "tests" / "golden" / "ssi_levels.json").read_text())
    c = next(c for c in g["cases"] if c["name"] == ssi["golden"])
    assert ssi["max_levels"] == c["depth"] == 11 and c["distinct"] == 1184049193 and c["verdict"] == "budget"
    assert b.DEEP_TABLE_SLOTS["ssi4x3"] >= 3 * c["distinct"]                      # the sharded tables stay sparse (32-byte probes)
    assert raft["max_levels"] == 19 and raft["golden_prefix"] == 18 and raft["golden"] == "raft5_mcr6_t2_m1_levels18"
    b.WORKLOAD = raft
    G0 = b.golden()
    assert len(G0["prefix_levels"]) == 18 and G0["distinct"] == raft["expect_distinct"] > 924041864
    # --levels L (tests/test_gpu_sharded.py: the same command lines on one shared GPU): a prefix of the same golden, sizes follow
    b.WORKLOAD = dict(raft, max_levels=13, reduced_levels=13)
    G0 = b.golden()
    assert G0["prefix_levels"] == [1, 6, 40, 205, 775, 2851, 10000, 32015, 97215, 287510, 816406, 2225540, 5913945] and G0["distinct"] == sum(G0["prefix_levels"])
