"""GPU parity tests (-m gpu) of the stand-alone operator surface: `morphsym_hgnn_amd.nn.{Linear, HeteroDictLinear, GraphConv,
HeteroConv}` called on their own (HIP operators of csrc/mshgnn_ops.hip through the C-ABI) against the oracle's restatement of the
torch_geometric==2.5.0 operators (oracle/pyg_restated, the module the reference's own model files were imported over when the golden
vectors were generated) in fp64 on the CPU: outputs and every gradient within 1e-4 relative (BASELINE.json north_star tolerance;
fp32 accumulation stays around 1e-6)."""
import os
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RTOL = 1e-4


def _ref_nn():
    sys.path.insert(0, os.path.join(ROOT, "oracle", "pyg_restated"))
    try:
        import torch_geometric.nn as ref
    finally:
        sys.path.pop(0)
    return ref


def _rel(a, b):
    return float((a.detach().double().cpu() - b.detach().double()).abs().max() / b.detach().double().abs().max().clamp(min=1e-30))


def _copy_params(dst, src):
    with torch.no_grad():
        for (kd, pd), (ks, ps) in zip(sorted(dst.named_parameters()), sorted(src.named_parameters())):
            assert kd == ks, (kd, ks)
            pd.copy_(ps)


def _require_gpu():
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a HIP device (there is no CPU fallback to fall through to)")


@pytest.mark.parametrize("M,K,N", [(1, 7, 5), (50, 450, 128), (3000, 900, 128), (257, 128, 3), (70000, 128, 128)])
def test_linear_matches_reference(M, K, N):
    _require_gpu()
    from morphsym_hgnn_amd import nn as pnn
    ref = _ref_nn()
    torch.manual_seed(M + K)
    r = ref.Linear(K, N).double()
    m = pnn.Linear(-1, N).double()
    x = torch.randn(M, K, dtype=torch.float64)
    xr = x.clone().requires_grad_(True)
    xg = x.cuda().requires_grad_(True)
    m.cuda()
    y = m(xg)                      # lazy weight materialises here
    assert tuple(m.weight.shape) == (N, K) and y.dtype == torch.float64
    _copy_params(m, r)
    y = m(xg)
    yr = r(xr)
    g = torch.randn(M, N, dtype=torch.float64)
    y.backward(g.cuda()); yr.backward(g)
    assert _rel(y, yr) < RTOL
    assert _rel(xg.grad, xr.grad) < RTOL and _rel(m.weight.grad, r.weight.grad) < RTOL and _rel(m.bias.grad, r.bias.grad) < RTOL


@pytest.mark.parametrize("aggr", ["add", "mean"])
@pytest.mark.parametrize("n_src,n_dst,E,H,O", [(12, 4, 4, 128, 128), (300, 170, 2000, 96, 40), (5, 9, 0, 16, 8), (40000, 40000, 120000, 128, 128)])
def test_graph_conv_matches_reference(aggr, n_src, n_dst, E, H, O):
    """Bipartite GraphConv on random multigraphs (repeated edges, isolated destinations, an empty edge set), 'add' and 'mean'."""
    _require_gpu()
    from morphsym_hgnn_amd import nn as pnn
    ref = _ref_nn()
    g = torch.Generator().manual_seed(E + n_src)
    r = ref.GraphConv((H, H), O, aggr=aggr).double()
    m = pnn.GraphConv((H, H), O, aggr=aggr).double().cuda()
    _copy_params(m, r)
    xs, xd = torch.randn(n_src, H, dtype=torch.float64, generator=g), torch.randn(n_dst, H, dtype=torch.float64, generator=g)
    ei = torch.stack([torch.randint(0, n_src, (E,), generator=g), torch.randint(0, max(1, n_dst - 1), (E,), generator=g)])     # the last destination has no in-edge
    a = [t.clone().requires_grad_(True) for t in (xs, xd)]
    b = [t.cuda().requires_grad_(True) for t in (xs, xd)]
    yr = r((a[0], a[1]), ei)
    y = m((b[0], b[1]), ei.cuda())
    go = torch.randn(n_dst, O, dtype=torch.float64, generator=g)
    yr.backward(go); y.backward(go.cuda())
    assert _rel(y, yr) < RTOL
    if E:
        assert _rel(b[0].grad, a[0].grad) < RTOL
        assert _rel(m.lin_rel.weight.grad, r.lin_rel.weight.grad) < RTOL
    else:
        assert float(b[0].grad.abs().max()) == 0.0 and float(m.lin_rel.weight.grad.abs().max()) == 0.0
    assert _rel(b[1].grad, a[1].grad) < RTOL
    assert _rel(m.lin_rel.bias.grad, r.lin_rel.bias.grad) < RTOL and _rel(m.lin_root.weight.grad, r.lin_root.weight.grad) < RTOL


def test_hetero_stack_matches_reference_on_the_a1_graph():
    """The reference's own layer stack (hgnn_c2.py:88-113,143-166 without the symmetry masks) assembled from the four operator modules on
    a PyG-batched A1-C2 graph: HeteroDictLinear -> relu -> 2 x HeteroConv{GraphConv} -> relu, forward and every parameter gradient."""
    _require_gpu()
    from morphsym_hgnn_amd import nn as pnn
    from morphsym_hgnn_amd import synth
    from tests import helpers
    ref = _ref_nn()
    spec = helpers.make_spec("c2", "a1-c2", "a1-c2", 128, 2)
    B, H = 5, 64
    x_dict, _ = synth.make_windows(21, B, spec.num_nodes, spec.widths, 12)
    x_dict = {k: v.double() for k, v in x_dict.items()}
    ei = spec.topology.edge_index_dict(B)
    types, rels = spec.topology.metadata()

    def build(nnmod):
        torch.manual_seed(3)
        enc = nnmod.HeteroDictLinear(-1, H, types)
        convs = torch.nn.ModuleList([nnmod.HeteroConv({tuple(r): nnmod.GraphConv(H, H, aggr="mean" if r[1] in ("gt", "gs") else "add") for r in rels}, aggr="sum")
                                     for _ in range(2)])
        return torch.nn.ModuleList([enc, convs]).double()

    def run(net, xd, eid):
        h = {k: v.relu() for k, v in net[0](xd).items()}
        for conv in net[1]:
            h = {k: v.relu() for k, v in conv(h, eid).items()}
        return h

    r = build(ref)
    m = build(pnn)
    hr = run(r, x_dict, ei)                                   # materialises the reference's lazy encoder
    m.cuda()
    xg = {k: v.cuda() for k, v in x_dict.items()}
    eg = {k: v.cuda() for k, v in ei.items()}
    run(m, xg, eg)                                            # ... and ours
    _copy_params(m, r)
    hm = run(m, xg, eg)
    assert set(hm) == set(hr)
    lr = sum((v ** 2).sum() for v in hr.values())
    lm = sum((v ** 2).sum() for v in hm.values())
    lr.backward(); lm.backward()
    for k in hr:
        assert _rel(hm[k], hr[k]) < RTOL, k
    for (kd, pd), (ks, ps) in zip(sorted(m.named_parameters()), sorted(r.named_parameters())):
        if ps.grad is None:
            assert pd.grad is None or float(pd.grad.abs().max()) == 0.0, kd
        else:
            assert _rel(pd.grad, ps.grad) < RTOL, kd


def test_csr_cache_never_serves_another_graph():
    """Fresh edge_index tensors every call (what PyG's collate produces): the cached CSR views are keyed on storage the cache keeps alive, so a new
    graph that happens to reuse a freed address cannot hit a stale entry."""
    _require_gpu()
    from morphsym_hgnn_amd import nn as pnn
    ref = _ref_nn()
    r = ref.GraphConv((8, 8), 8, aggr="add").double()
    m = pnn.GraphConv((8, 8), 8, aggr="add").double().cuda()
    _copy_params(m, r)
    g = torch.Generator().manual_seed(0)
    xs, xd = torch.randn(30, 8, dtype=torch.float64, generator=g), torch.randn(20, 8, dtype=torch.float64, generator=g)
    for it in range(40):
        ei = torch.stack([torch.randint(0, 30, (50,), generator=g), torch.randint(0, 20, (50,), generator=g)])
        eg = ei.cuda()                       # a new device tensor per iteration; the previous one is dropped
        y = m((xs.cuda(), xd.cuda()), eg)
        assert _rel(y, r((xs, xd), ei)) < RTOL, it
        del eg
