"""Node-level liveness (spec.ModelSpec.node_liveness, the mirror of the plan compiler's): which nodes of which layer can reach the decoder.  Checked
against the oracle (fp64 restatement of the reference, pinned by the golden vectors): a parameter whose every use sits on dead nodes gets an EXACT
zero gradient there, and every parameter with a live use gets a non-zero one on random data -- the engine skips exactly the former."""
import random

import pytest
import torch

from morphsym_hgnn_amd import synth
from morphsym_hgnn_amd.spec import ModelSpec
from morphsym_hgnn_amd.topology import RobotTopology
from oracle import ms_hgnn_oracle as orc
from tests import helpers


def predicted_dead(spec):
    return spec.dead_parameters()


def oracle_zero_grads(spec, B=3, seed=5):
    n_y = spec.out_channels * spec.num_nodes[spec.out_type] if spec.regression else spec.num_nodes[spec.out_type]
    x_dict, y = synth.make_windows(seed, B, spec.num_nodes, spec.widths, n_y, classification=not spec.regression)
    params = synth.make_params(seed, spec.param_shapes())
    _, _, grads = orc.step(helpers.oracle_config(spec), params, x_dict, spec.topology.edge_index_dict(B), y, B)
    return {k for k, g in grads.items() if float(g.abs().max()) == 0.0}


@pytest.mark.parametrize("kind,topo,cfg,layers", [("c2", "a1-c2", "a1-c2", 1), ("c2", "a1-c2", "a1-c2", 3), ("c2", "a1-c2", "a1-c2", 5),
                                                  ("k4", "mini_cheetah-k4", "mini_cheetah-k4", 3), ("mi", "quadruped-mi", "", 2),
                                                  ("k4_com", "solo-k4-com", "solo-k4", 2)])
def test_dead_parameters_are_exactly_the_oracles_zero_gradients(kind, topo, cfg, layers):
    spec = helpers.make_spec(kind, topo, cfg, 128, layers, grf=3 if kind == "c2" else 1)
    assert predicted_dead(spec) == oracle_zero_grads(spec)


def test_a1_c2_at_three_layers_never_sees_its_base_nodes():
    """The headline configuration: the base (IMU) nodes are four hops from the feet, so at 3 layers nothing of them is live -- 49 of the 84 parameter
    tensors are dead, the encoder runs on 16 of 18 nodes and reads 10 808 of 14 408 input values' bytes per window; at 5 layers everything is live."""
    spec = helpers.make_spec("c2", "a1-c2", "a1-c2", 128, 3)
    live, need = spec.node_liveness()
    assert need[0] == {"base": [], "joint": list(range(12)), "foot": [0, 1, 2, 3]}
    assert live[0]["joint"] == [1, 2, 4, 5, 7, 8, 10, 11] and live[1]["joint"] == [2, 5, 8, 11] and live[2]["joint"] == []
    assert all(live[l]["base"] == [] for l in range(3)) and len(predicted_dead(spec)) == 49
    deep = helpers.make_spec("c2", "a1-c2", "a1-c2", 128, 5)
    assert deep.node_liveness()[1][0] == {"base": [0, 1], "joint": list(range(12)), "foot": [0, 1, 2, 3]}


@pytest.mark.parametrize("seed", [1, 2, 3, 4, 5, 6])
def test_liveness_on_random_topologies(seed):
    rng = random.Random(seed)
    n = {"base": rng.randint(1, 3), "joint": rng.randint(3, 9), "foot": rng.randint(1, 4)}
    rels = []
    for s_, d_ in (("base", "joint"), ("joint", "base"), ("joint", "joint"), ("foot", "joint"), ("joint", "foot")):
        pairs = [[rng.randrange(n[s_]), rng.randrange(n[d_])] for _ in range(rng.randint(0 if d_ != "foot" else 1, max(n[s_], n[d_])))]
        rels.append(((s_, "connect", d_), pairs))
    spec = ModelSpec(kind="mi", topology=RobotTopology(name=f"r{seed}", num_nodes=n, relations=rels), hidden=128, num_layers=rng.randint(1, 4),
                     widths={"base": 6, "joint": 5, "foot": 3}, regression=True, grf_dimension=1, group=None, num_timesteps=1)
    assert predicted_dead(spec) == oracle_zero_grads(spec, seed=seed)


def test_type_level_liveness_is_a_superset(monkeypatch):
    spec = helpers.make_spec("c2", "a1-c2", "a1-c2", 128, 3)
    fine = spec.node_liveness()
    monkeypatch.setenv("MSHGNN_PRUNE", "0")
    coarse = spec.node_liveness()
    for a, b in zip(fine[0] + fine[1], coarse[0] + coarse[1]):
        for t in a:
            assert set(a[t]) <= set(b[t])
    assert coarse[1][0] == {"base": [0, 1], "joint": list(range(12)), "foot": [0, 1, 2, 3]}


@pytest.mark.parametrize("hidden", [128, 512])
@pytest.mark.parametrize("seed", [1, 2, 3, 4, 5, 6, 7, 8])
def test_python_liveness_mirror_agrees_with_both_plan_compilers(seed, hidden):
    """`ddp.flat_data_parallel(live_only=True)` is exact only if spec.node_liveness (Python) and the C plan compilers (mshgnn_plan.hpp at hidden 128,
    mshgnn_gen_plan.hpp at 512) prune the same nodes.  Host-only cross-check on random topologies: the input bytes the plan reports as live
    (`mshgnn_info.bytes_in_live`, summed over the nodes ITS liveness keeps) == the same sum over the Python mirror's `need[0]`."""
    from morphsym_hgnn_amd import engine
    rng = random.Random(100 + seed)
    n = {"base": rng.randint(1, 3), "joint": rng.randint(3, 9), "foot": rng.randint(1, 4)}
    rels = []
    for s_, d_ in (("base", "joint"), ("joint", "base"), ("joint", "joint"), ("foot", "joint"), ("joint", "foot")):
        pairs = [[rng.randrange(n[s_]), rng.randrange(n[d_])] for _ in range(rng.randint(0 if d_ != "foot" else 1, max(n[s_], n[d_])))]
        rels.append(((s_, "connect", d_), pairs))
    widths = {"base": 6, "joint": 5, "foot": 3}
    spec = ModelSpec(kind="mi", topology=RobotTopology(name=f"x{seed}", num_nodes=n, relations=rels), hidden=hidden, num_layers=rng.randint(1, 4),
                     widths=widths, regression=True, grf_dimension=1, group=None, num_timesteps=1)
    info = engine.compile_plan_host(spec, "bf16")
    _, need = spec.node_liveness()
    assert info.bytes_in_live == 2 * sum(len(need[0][t]) * widths[t] for t in widths)
    assert info.bytes_in == 2 * sum(n[t] * widths[t] for t in widths)
