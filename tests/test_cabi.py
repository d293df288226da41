"""CPU tests: the C-ABI library loads, exports every symbol include/mshgnn.h declares, and the host-side plan
compiler (no GPU involved) produces the expected work counts and rejects bad descriptors."""
import os
import re

import pytest

from morphsym_hgnn_amd import engine
from tests import helpers

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    lib = engine.load_library()
    hdr = open(os.path.join(ROOT, "include", "mshgnn.h")).read()
    declared = set(re.findall(r"\b(mshgnn_[a-z_]+)\s*\(", hdr))
    assert declared, "no declarations found in include/mshgnn.h"
    for name in declared:
        assert hasattr(lib, name), f"libmshgnn.so does not export {name}"
    assert b"gfx950" in lib.mshgnn_version()


def test_plan_compiler_work_counts(monkeypatch):
    # A1-C2 h=128 L=3 with type-level liveness (MSHGNN_PRUNE=0, rounds 1-3): encoder 1 844 224 + (52 + 40 + 8) live node-linears * 32 768 + decoder 3 072 FLOP / window
    monkeypatch.setenv("MSHGNN_PRUNE", "0")
    info = engine.compile_plan_host(helpers.make_spec("c2", "a1-c2", "a1-c2", 128, 3), "f32")
    assert info.rows_per_tile == 16 and info.total_nodes == 18 and info.lds_bytes == 20 * 8192   # 18 node blocks + 2 base_transform scratch
    assert info.flops_fwd == 1844224 + 100 * 32768 + 3072
    assert info.bytes_in == 7204 * 4 and info.bytes_in_live == 7204 * 4
    # node-level liveness (the default): at 3 layers the base nodes are four hops from the feet -- the encoder runs on the 12 joints + 4 feet, layer 0
    # on thighs, knees and feet (8 + 12 + 8 node-linears), layer 1 on knees and feet (12 + 8), layer 2 on the feet (8)
    monkeypatch.delenv("MSHGNN_PRUNE")
    info = engine.compile_plan_host(helpers.make_spec("c2", "a1-c2", "a1-c2", 128, 3), "f32")
    assert info.flops_fwd == 2 * 128 * (12 * 450 + 4) + (28 + 20 + 8) * 32768 + 3072
    assert info.bytes_in == 7204 * 4 and info.bytes_in_live == (12 * 450 + 4) * 4
    # the paper's depth: everything is live in the first four layers
    info8 = engine.compile_plan_host(helpers.make_spec("c2", "a1-c2", "a1-c2", 128, 8), "f32")
    assert info8.bytes_in_live == info8.bytes_in == 7204 * 4
    info16 = engine.compile_plan_host(helpers.make_spec("c2", "a1-c2", "a1-c2", 128, 3), "bf16")
    assert info16.rows_per_tile == 16 and info16.bytes_in == 7204 * 2 and info16.lds_bytes == 20 * 4096
    k4 = engine.compile_plan_host(helpers.make_spec("k4", "mini_cheetah-k4", "mini_cheetah-k4", 128, 8, regression=False), "f32")
    assert k4.total_nodes == 20 and k4.lds_bytes == 160 * 1024   # fp32: no spare LDS, base_transform re-uses its own blocks


def test_plan_compiler_kernel_sets():
    """Which stack-kernel variants a plan allows: fp32 plans use the per-layer kernels; bf16 plans the fused stack kernels, and the
    slab variant (two 4-wave workgroups of <= 80 KB LDS per CU, <= 12 + 6 / 8 accumulator slots) where the topology fits."""
    c2 = helpers.make_spec("c2", "a1-c2", "a1-c2", 128, 3)
    assert engine.compile_plan_host(c2, "f32").kernel_sets == 0
    assert engine.compile_plan_host(c2, "bf16").kernel_sets == 3        # fused | slab
    assert engine.compile_plan_host(helpers.make_spec("c2", "a1-c2", "a1-c2", 128, 8), "bf16").kernel_sets & 3 == 3
    mi = helpers.make_spec("mi", "quadruped-mi", "", 128, 2, grf=1)
    assert engine.compile_plan_host(mi, "bf16").kernel_sets & 3 == 3
    k4 = helpers.make_spec("k4", "mini_cheetah-k4", "mini_cheetah-k4", 128, 8, regression=False)
    assert engine.compile_plan_host(k4, "bf16").kernel_sets & 3 == 3       # 20 nodes: the base_transform scratch aliases four joint blocks (80 KB, two workgroups per CU), group B of 8 slots
    for kind, topo, cfg in (("k4_com", "solo-k4-com", "solo-k4-com"), ("c2_com", "solo-c2-com", "solo-c2-com")):
        try:
            spec = helpers.make_spec(kind, topo, cfg, 128, 2)
        except Exception:
            continue
        assert engine.compile_plan_host(spec, "bf16").kernel_sets & 1


def test_plan_compiler_rejects_bad_descriptors():
    spec = helpers.make_spec("c2", "a1-c2", "a1-c2", 64, 2)   # hidden not a multiple of 128: neither engine takes it
    with pytest.raises(engine.MshgnnError, match="128"):
        engine.compile_plan_host(spec, "f32")


def test_wide_and_many_node_models_compile_on_the_generic_engine(monkeypatch):
    """hidden = 256 / 512, 129 nodes per window (BASELINE configs[4]): the specialised plan declines, the generic-width engine's compiler
    takes the descriptor (kernel_sets bit 2) for every arithmetic mode."""
    from morphsym_hgnn_amd import synth, topology
    from morphsym_hgnn_amd.spec import ModelSpec
    synth32 = ModelSpec(kind="mi", topology=topology.synthetic_limbs(32), hidden=512, num_layers=6, widths=synth.feature_widths("mi", True),
                        regression=True, grf_dimension=3)
    assert synth32.num_params() == 16438787
    for dt in ("bf16", "x3", "f32"):
        info = engine.compile_plan_host(synth32, dt)
        assert info.kernel_sets == 4 and info.total_nodes == 129
        assert engine.compile_plan_host(helpers.make_spec("c2", "a1-c2", "a1-c2", 256, 3), dt).kernel_sets == 4
    # algorithmic work of configs[4]: 2.31 GFLOP per window forward + backward with node-level liveness (hips and thighs are dead in the second-last
    # layer, hips in the one before, ...), 2.70 with whole types live (MSHGNN_PRUNE=0: the figure of rounds 2-3)
    assert abs((info.flops_fwd + info.flops_bwd) / 1e9 - 2.31) < 0.05
    monkeypatch.setenv("MSHGNN_PRUNE", "0")
    info = engine.compile_plan_host(synth32, "bf16")
    assert abs((info.flops_fwd + info.flops_bwd) / 1e9 - 2.70) < 0.05
    # the same many-node topologies at hidden = 128, where the specialised plan is tried first: it declines (its node tables hold 64 entries) instead of
    # writing past them, and the generic engine takes over
    for limbs in (32, 40, 70):
        many = ModelSpec(kind="mi", topology=topology.synthetic_limbs(limbs), hidden=128, num_layers=3, widths=synth.feature_widths("mi", True),
                         regression=True, grf_dimension=1)
        for dt in ("bf16", "x3"):
            info = engine.compile_plan_host(many, dt)
            assert info.kernel_sets == 4 and info.total_nodes == 1 + 4 * limbs


def test_engine_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        engine.Engine(helpers.make_spec("c2", "a1-c2", "a1-c2", 128, 1))


def test_abi_guard_reports_the_bindings_struct_sizes():
    """include/mshgnn.h MSHGNN_ABI_VERSION == the library's == engine.ABI_VERSION, and every struct that crosses the boundary has the size the ctypes
    mirror has (mshgnn_plan_info writes sizeof(mshgnn_info) bytes through the caller's pointer: a stale binding must be caught before that)."""
    import ctypes as C
    lib = engine.load_library()
    hdr = open(os.path.join(ROOT, "include", "mshgnn.h")).read()
    assert int(re.search(r"#define MSHGNN_ABI_VERSION (\d+)", hdr).group(1)) == lib.mshgnn_abi_version() == engine.ABI_VERSION
    for which, st in enumerate((engine.MshgnnDesc, engine.MshgnnInfo, engine.MshgnnWsLayout, engine.MshgnnWindowDesc, engine.MshgnnKernelStat)):
        assert lib.mshgnn_struct_size(which) == C.sizeof(st) > 0, st.__name__
    assert lib.mshgnn_struct_size(99) == 0


def test_comm_entry_points_reject_bad_arguments_without_touching_rccl():
    """The data-parallel collective's entry points validate their arguments before RCCL is even loaded (no GPU, no librccl needed for that)."""
    import ctypes as C
    lib = engine.load_library()
    out = C.c_void_p()
    assert lib.mshgnn_comm_create(None, None, 2, 0, C.byref(out)) == -1 and b"mshgnn_comm_create" in lib.mshgnn_last_error()
    ident = C.create_string_buffer(128)
    assert lib.mshgnn_comm_create(None, ident, 2, 5, C.byref(out)) == -1      # rank out of range
    assert lib.mshgnn_comm_unique_id(None, None) == -1
    assert lib.mshgnn_comm_allreduce_mean(None, None, 4, None) == -1
    lib.mshgnn_comm_destroy(None)      # a null communicator is a no-op
