"""ctypes binding of libmshgnn.so (include/mshgnn.h) and the host-side engine object.

PyTorch is used here for device memory, streams and dtype casts only; all arithmetic of the hot path
runs in the HIP kernels behind the C-ABI.  There is NO CPU fallback: if the library is missing or no GPU
is present, construction fails loudly.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

from .spec import ModelSpec, relation_aggr

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MSHGNN_LIB") or os.path.join(_HERE, "libmshgnn.so")      # MSHGNN_LIB: an alternative build of the same library (kernel experiments)

MAX_TYPES = 4
F32, BF16, BF16X3 = 0, 1, 2
DTYPE_CODES = {"f32": F32, "bf16": BF16, "x3": BF16X3}      # "x3": split-bf16 parity plan (include/mshgnn.h, MSHGNN_BF16X3)
FLAG_RESIDUAL, FLAG_BASE_MLP = 1, 2


class MshgnnDesc(C.Structure):
    _fields_ = [
        ("n_types", C.c_int32), ("hidden", C.c_int32), ("num_layers", C.c_int32), ("n_rel", C.c_int32),
        ("out_type", C.c_int32), ("out_channels", C.c_int32), ("mlp_type", C.c_int32), ("flags", C.c_uint32),
        ("dtype", C.c_int32),
        ("type_nodes", C.c_int32 * MAX_TYPES), ("type_width", C.c_int32 * MAX_TYPES),
        ("rel_src", C.POINTER(C.c_int32)), ("rel_dst", C.POINTER(C.c_int32)), ("rel_mean", C.POINTER(C.c_int32)),
        ("rel_edge_off", C.POINTER(C.c_int32)), ("edges", C.POINTER(C.c_int32)),
        ("in_mask", C.POINTER(C.c_float) * MAX_TYPES), ("out_mask", C.POINTER(C.c_float)),
        ("off_enc_w", C.POINTER(C.c_int64)), ("off_enc_b", C.POINTER(C.c_int64)),
        ("off_rel_w", C.POINTER(C.c_int64)), ("off_rel_b", C.POINTER(C.c_int64)), ("off_root_w", C.POINTER(C.c_int64)),
        ("off_mlp", C.c_int64 * 4), ("off_dec_w", C.c_int64), ("off_dec_b", C.c_int64), ("n_flat", C.c_int64),
    ]


class MshgnnInfo(C.Structure):
    _fields_ = [
        ("rows_per_tile", C.c_int32), ("total_nodes", C.c_int32), ("lds_bytes", C.c_int64),
        ("flops_fwd", C.c_double), ("flops_bwd", C.c_double), ("flops_exec_fwd", C.c_double), ("flops_exec_bwd", C.c_double),
        ("bytes_in", C.c_double), ("n_gradw_workgroups", C.c_int32), ("n_launches_fwd", C.c_int32), ("n_launches_bwd", C.c_int32),
        ("kernel_sets", C.c_int32), ("grad_split", C.c_int64), ("bytes_in_live", C.c_double),
    ]


class MshgnnWsLayout(C.Structure):
    _fields_ = [
        ("total", C.c_size_t), ("x", C.c_size_t * 17), ("dx", C.c_size_t * 17), ("dh", C.c_size_t * 16), ("dd", C.c_size_t * 16),
        ("mask", C.c_size_t * 16), ("hb", C.c_size_t * 16), ("t1", C.c_size_t * 16), ("du", C.c_size_t * 16),
        ("wpack", C.c_size_t), ("bias", C.c_size_t), ("dec_slabs", C.c_size_t), ("slabs", C.c_size_t), ("loss", C.c_size_t),
    ]


class MshgnnWindowDesc(C.Structure):
    _fields_ = [
        ("n_types", C.c_int32), ("dtype", C.c_int32), ("history", C.c_int32), ("normalize", C.c_int32),
        ("type_nodes", C.c_int32 * 4), ("type_width", C.c_int32 * 4),
        ("n_src", C.c_int32), ("n_runs", C.c_int32), ("n_rows", C.c_int32), ("fast_layout", C.c_int32), ("runs", C.c_void_p), ("rows", C.c_void_p),
        ("n_label", C.c_int32), ("label_src", C.c_int32), ("label_rotate", C.c_int32), ("quat_src", C.c_int32),
        ("label_cols", C.c_void_p), ("run_ptrs_ready", C.c_int32), ("reserved_", C.c_int32),
    ]


class MshgnnKernelStat(C.Structure):
    _fields_ = [
        ("name", C.c_char * 32), ("launches", C.c_int32), ("bound", C.c_int32), ("total_ms", C.c_float), ("_pad", C.c_float),
        ("flops_per_window", C.c_double), ("flops_exec_per_window", C.c_double), ("bytes_per_window", C.c_double),
    ]


EXPORTS = [
    "mshgnn_last_error", "mshgnn_version", "mshgnn_plan_create", "mshgnn_plan_destroy", "mshgnn_plan_info", "mshgnn_plan_specialised", "mshgnn_plan_attach_program",
    "mshgnn_plan_compile_host", "mshgnn_workspace_layout", "mshgnn_forward", "mshgnn_backward", "mshgnn_mse_loss", "mshgnn_ce_loss", "mshgnn_metrics_regression_step", "mshgnn_metrics_classification_step", "mshgnn_metrics_com_step",
    "mshgnn_profile_enable", "mshgnn_profile_read", "mshgnn_backward_mse", "mshgnn_adam_step", "mshgnn_adam_step_counted",
    "mshgnn_metrics_regression", "mshgnn_metrics_classification", "mshgnn_grf_body_to_world", "mshgnn_assemble_windows",
    "mshgnn_backward_ce", "mshgnn_step_mse", "mshgnn_step_mse_phase",
    "mshgnn_step_mse_series", "mshgnn_step_ce_series", "mshgnn_step_ce", "mshgnn_op_gemm", "mshgnn_op_gemm_workspace", "mshgnn_op_aggregate", "mshgnn_op_colsum", "mshgnn_op_colsum_workspace",
    "mshgnn_abi_version", "mshgnn_struct_size", "mshgnn_forward_src", "mshgnn_step_mse_src", "mshgnn_step_ce_src",
    "mshgnn_comm_unique_id", "mshgnn_comm_create", "mshgnn_comm_destroy", "mshgnn_comm_allreduce_mean", "mshgnn_comm_allreduce_sum",
]
ABI_VERSION = 6      # include/mshgnn.h MSHGNN_ABI_VERSION: the ctypes structures above mirror THAT header

_lib = None


def build_library(force: bool = False) -> str:
    """Compile libmshgnn.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
    src = os.path.join(_HERE, "csrc")
    if force and os.path.exists(LIB_PATH):
        os.remove(LIB_PATH)
    subprocess.run(["make", "-j", "6", "-C", src], check=True)
    return LIB_PATH


def load_library():
    """Load libmshgnn.so; raises if it has not been built (no silent fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                           "or `make -C morphsym_hgnn_amd/csrc` -- the MS-HGNN engine has no CPU fallback")
    lib = C.CDLL(LIB_PATH)
    lib.mshgnn_last_error.restype = C.c_char_p
    lib.mshgnn_version.restype = C.c_char_p
    lib.mshgnn_plan_create.argtypes = [C.POINTER(MshgnnDesc), C.POINTER(C.c_void_p)]
    lib.mshgnn_plan_destroy.argtypes = [C.c_void_p]
    lib.mshgnn_plan_destroy.restype = None
    lib.mshgnn_plan_info.argtypes = [C.c_void_p, C.POINTER(MshgnnInfo)]
    if hasattr(lib, "mshgnn_plan_attach_program"):
        lib.mshgnn_plan_attach_program.argtypes = [C.c_void_p, C.c_void_p]
        lib.mshgnn_plan_attach_program.restype = C.c_int
    if hasattr(lib, "mshgnn_plan_specialised"):      # (absent from older builds of the library handed over through MSHGNN_LIB for A/B runs)
        lib.mshgnn_plan_specialised.argtypes = [C.c_void_p]
        lib.mshgnn_plan_specialised.restype = C.c_char_p
    lib.mshgnn_plan_compile_host.argtypes = [C.POINTER(MshgnnDesc), C.POINTER(MshgnnInfo), C.POINTER(C.c_int32)]
    lib.mshgnn_workspace_layout.argtypes = [C.c_void_p, C.c_int64, C.c_int, C.POINTER(MshgnnWsLayout)]
    lib.mshgnn_forward.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_int64), C.c_void_p, C.c_void_p,
                                   C.c_void_p, C.c_int64, C.c_int, C.c_void_p]
    lib.mshgnn_backward.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_int64), C.c_void_p, C.c_void_p,
                                    C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]
    lib.mshgnn_backward_mse.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_int64), C.c_void_p, C.c_void_p, C.c_void_p,
                                        C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]
    lib.mshgnn_backward_ce.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_int64), C.c_void_p, C.c_void_p, C.c_void_p,
                                       C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]
    lib.mshgnn_step_mse.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_int64), C.c_void_p, C.c_void_p, C.c_void_p,
                                    C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]
    lib.mshgnn_step_ce.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_int64), C.c_void_p, C.c_void_p, C.c_void_p,
                                   C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]
    lib.mshgnn_step_mse_phase.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_int64), C.c_void_p, C.c_void_p, C.c_void_p,
                                          C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p]
    lib.mshgnn_adam_step.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_float, C.c_float,
                                     C.c_float, C.c_float, C.c_float, C.c_void_p]
    if hasattr(lib, "mshgnn_adam_step_counted"):
        lib.mshgnn_adam_step_counted.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_float, C.c_float,
                                                 C.c_float, C.c_float, C.c_float, C.c_void_p]
    lib.mshgnn_profile_enable.argtypes = [C.c_void_p, C.c_int]
    lib.mshgnn_profile_read.argtypes = [C.c_void_p, C.POINTER(MshgnnKernelStat), C.POINTER(C.c_int32)]
    lib.mshgnn_mse_loss.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.mshgnn_ce_loss.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.mshgnn_metrics_regression.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]
    lib.mshgnn_metrics_classification.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.mshgnn_metrics_regression_step.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.mshgnn_metrics_classification_step.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                                       C.c_void_p]
    lib.mshgnn_metrics_com_step.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_void_p, C.c_void_p,
                                            C.c_void_p, C.c_void_p]
    lib.mshgnn_grf_body_to_world.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]
    lib.mshgnn_assemble_windows.argtypes = [C.POINTER(MshgnnWindowDesc), C.POINTER(C.c_void_p), C.POINTER(C.c_int64), C.POINTER(C.c_int64),
                                            C.c_void_p, C.c_int64, C.POINTER(C.c_void_p), C.POINTER(C.c_int64), C.c_void_p, C.c_void_p,
                                            C.c_void_p]
    lib.mshgnn_step_mse_series.argtypes = [C.c_void_p, C.POINTER(MshgnnWindowDesc), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_int64),
                                           C.POINTER(C.c_int64), C.c_void_p, C.c_int64, C.POINTER(C.c_void_p), C.POINTER(C.c_int64), C.c_void_p, C.c_void_p,
                                           C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.mshgnn_step_ce_series.argtypes = list(lib.mshgnn_step_mse_series.argtypes)
    lib.mshgnn_op_gemm_workspace.restype = C.c_int64
    lib.mshgnn_op_gemm_workspace.argtypes = [C.c_int64, C.c_int64, C.c_int64, C.POINTER(C.c_int32)]
    lib.mshgnn_op_gemm.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_int64,
                                   C.c_int64, C.c_int64, C.c_int64, C.c_int, C.c_void_p, C.c_void_p]
    lib.mshgnn_op_aggregate.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int64,
                                        C.c_void_p]
    lib.mshgnn_op_colsum_workspace.restype = C.c_int64
    lib.mshgnn_op_colsum_workspace.argtypes = [C.c_int64, C.c_int64]
    lib.mshgnn_op_colsum.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p]
    src6 = [C.c_void_p, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_int64), C.POINTER(C.c_void_p), C.POINTER(C.c_int64)]
    lib.mshgnn_forward_src.argtypes = src6 + [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p]
    lib.mshgnn_step_mse_src.argtypes = src6 + [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]
    lib.mshgnn_step_ce_src.argtypes = src6 + [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]
    lib.mshgnn_struct_size.restype = C.c_size_t
    lib.mshgnn_struct_size.argtypes = [C.c_int]
    lib.mshgnn_comm_unique_id.argtypes = [C.c_char_p, C.c_void_p]
    lib.mshgnn_comm_create.argtypes = [C.c_char_p, C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_void_p)]
    lib.mshgnn_comm_destroy.argtypes = [C.c_void_p]
    lib.mshgnn_comm_destroy.restype = None
    lib.mshgnn_comm_allreduce_mean.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]
    lib.mshgnn_comm_allreduce_sum.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]
    # ABI guard: the structures of this binding and the library's must be the same header version and the same sizes (a stale .so, or a binding
    # edited without the header, would otherwise have the library write past a smaller struct)
    if lib.mshgnn_abi_version() != ABI_VERSION:
        raise RuntimeError(f"{LIB_PATH}: ABI version {lib.mshgnn_abi_version()} != {ABI_VERSION} of this binding -- rebuild (make -C morphsym_hgnn_amd/csrc)")
    for which, st in enumerate((MshgnnDesc, MshgnnInfo, MshgnnWsLayout, MshgnnWindowDesc, MshgnnKernelStat)):
        if lib.mshgnn_struct_size(which) != C.sizeof(st):
            raise RuntimeError(f"{LIB_PATH}: sizeof({st.__name__}) is {lib.mshgnn_struct_size(which)} in the library, {C.sizeof(st)} in engine.py")
    _lib = lib
    return lib


class MshgnnError(RuntimeError):
    pass


def _check(lib, rc: int, what: str):
    if rc != 0:
        raise MshgnnError(f"{what} failed ({rc}): {lib.mshgnn_last_error().decode()}")


class _DescHolder:
    """Keeps the numpy arrays a MshgnnDesc points into alive."""

    def __init__(self, spec: ModelSpec, dtype: int):
        self.keep: List[np.ndarray] = []
        d = MshgnnDesc()
        types = spec.node_types
        tix = {t: i for i, t in enumerate(types)}
        d.n_types = len(types)
        d.hidden = spec.hidden
        d.num_layers = spec.num_layers
        d.n_rel = len(spec.edge_types)
        d.out_type = tix[spec.out_type]
        d.out_channels = spec.out_channels
        d.mlp_type = tix["base"] if spec.has_base_transform else -1
        d.flags = (FLAG_RESIDUAL if spec.residual else 0) | (FLAG_BASE_MLP if spec.has_base_transform else 0)
        d.dtype = dtype
        for i, t in enumerate(types):
            d.type_nodes[i] = spec.num_nodes[t]
            d.type_width[i] = spec.widths[t]

        def arr(a, dt):
            a = np.ascontiguousarray(np.asarray(a, dtype=dt))
            self.keep.append(a)
            return a

        def iptr(a):
            return arr(a, np.int32).ctypes.data_as(C.POINTER(C.c_int32))

        def lptr(a):
            return arr(a, np.int64).ctypes.data_as(C.POINTER(C.c_int64))

        def fptr(a):
            return arr(a, np.float32).ctypes.data_as(C.POINTER(C.c_float))

        d.rel_src = iptr([tix[s] for s, _, _ in spec.edge_types])
        d.rel_dst = iptr([tix[t] for _, _, t in spec.edge_types])
        d.rel_mean = iptr([1 if relation_aggr(spec.kind, et) == "mean" else 0 for et in spec.edge_types])
        off, edges = [0], []
        for et in spec.edge_types:
            e = spec.topology.edges(et)
            for s_, d_ in e:
                edges += [s_, d_]
            off.append(off[-1] + len(e))
        d.rel_edge_off = iptr(off)
        d.edges = iptr(edges if edges else [0])
        masks = spec.input_masks()
        for i, t in enumerate(types):
            d.in_mask[i] = fptr(masks[t].numpy().reshape(-1))
        d.out_mask = fptr(spec.output_mask().numpy().reshape(-1))
        po = spec.param_offsets()
        from .spec import rel_key
        d.off_enc_w = lptr([po[f"encoder.lins.{t}.weight"][0] for t in types])
        d.off_enc_b = lptr([po[f"encoder.lins.{t}.bias"][0] for t in types])
        rw, rb, ro = [], [], []
        for l in range(spec.num_layers):
            for et in spec.edge_types:
                p = f"convs.{l}.convs.{rel_key(et)}."
                rw.append(po[p + "lin_rel.weight"][0]); rb.append(po[p + "lin_rel.bias"][0]); ro.append(po[p + "lin_root.weight"][0])
        d.off_rel_w, d.off_rel_b, d.off_root_w = lptr(rw), lptr(rb), lptr(ro)
        if spec.has_base_transform:
            for k, name in enumerate(["base_transform.0.weight", "base_transform.0.bias", "base_transform.2.weight", "base_transform.2.bias"]):
                d.off_mlp[k] = po[name][0]
        d.off_dec_w = po["decoder.weight"][0]
        d.off_dec_b = po["decoder.bias"][0]
        d.n_flat = spec.flat_size()
        self.desc = d


ROW_ALIGN = 128      # bytes; see row_pitch (a module constant since round 6: the A/B run it served as an environment switch was made in round 4)


def row_pitch(width: int, elem_bytes: int) -> int:
    """Row pitch (elements) of an input type in the engine's own layout.  Rows start 16-byte aligned (what the kernels' 16-byte loads need); rows
    longer than a cache line start ON a cache line (128 bytes): a 128-column K chunk of a row is then exactly two lines, and a line is never shared
    by two chunks that different workgroups read at different times (measured, A1-C2 8192 windows bf16: the weight-gradient launch 110 -> 101 us,
    step 0.298 -> 0.288 ms on the same box).  The pad columns are never read as data.  (ROW_ALIGN = 16 restores the 16-byte pitch of rounds 1-3.)"""
    align = ROW_ALIGN
    q = (align if (align > 16 and width * elem_bytes > align) else 16) // elem_bytes
    return (width + q - 1) // q * q


def accepted_pitches(width: int, elem_bytes: int):
    """The row pitches (elements) a device tensor of an input type may arrive at: the reference's dense width, the engine's own pitch (`row_pitch`)
    and the plain 16-byte-rounded width (what `ROW_ALIGN = 16` / the on-device window assembly of round 2 produced).  Anything else --
    e.g. a tensor with a few MORE real feature columns than the model was built for -- is a shape error, not padding."""
    q = 16 // elem_bytes
    return {width, row_pitch(width, elem_bytes), (width + q - 1) // q * q}


def compile_plan_host(spec: ModelSpec, dtype: str = "f32") -> MshgnnInfo:
    """Run the plan compiler only (no GPU needed) and return its work/traffic summary."""
    lib = load_library()
    h = _DescHolder(spec, DTYPE_CODES[dtype])
    info = MshgnnInfo()
    n = C.c_int32(0)
    _check(lib, lib.mshgnn_plan_compile_host(C.byref(h.desc), C.byref(info), C.byref(n)), "mshgnn_plan_compile_host")
    return info


def flatten_params(spec: ModelSpec, params: Dict[str, torch.Tensor], device=None) -> torch.Tensor:
    """state_dict-named tensors -> the flat fp32 parameter buffer of the C-ABI."""
    flat = torch.zeros(spec.flat_size(), dtype=torch.float32)
    for k, (off, n) in spec.param_offsets().items():
        flat[off:off + n] = params[k].detach().reshape(-1).to(torch.float32).cpu()
    return flat.to(device) if device is not None else flat


def unflatten(spec: ModelSpec, flat: torch.Tensor) -> Dict[str, torch.Tensor]:
    """Views of a flat parameter/gradient buffer keyed by state_dict names."""
    shapes = spec.param_shapes()
    return {k: flat[off:off + n].view(shapes[k]) for k, (off, n) in spec.param_offsets().items()}


class WideInputs(list):
    """What `Engine.cast_inputs` hands back for the reference's own device tensors (fp64, or fp32 at the dense pitch) on the bf16 / split plans: no cast
    pass has run.  The list holds the engine's plan-dtype row buffers (what the weight-gradient kernel and any later backward read -- the encoder of
    the next forward / step on this object WRITES them while it converts the source rows in registers: `mshgnn_*_src`); `.src` are the caller's
    tensors, `.pending` says the rows have not been materialised yet.  The row buffers belong to the engine (one set per batch size): a later
    `cast_inputs` of the same batch size reuses them, exactly as a later forward reuses the activation stash."""

    def __init__(self, rows, src, src_bytes):
        super().__init__(rows)
        self.src, self.src_bytes, self.pending = list(src), int(src_bytes), True
        self.gen = None      # generation of the engine's row buffers this object's rows were written as (set when the encoder materialises them)


class Engine:
    """One compiled plan (topology x model dims x precision) on one GPU."""

    def __init__(self, spec: ModelSpec, dtype: str = "f32", device: Optional[torch.device] = None):
        if dtype not in DTYPE_CODES:
            raise ValueError("dtype must be 'f32' (exact fp32 MFMA), 'bf16' (throughput) or 'x3' (split-bf16 parity plan)")
        if not torch.cuda.is_available():
            raise RuntimeError("the MS-HGNN engine needs a HIP device; there is no CPU fallback")
        self.lib = load_library()
        self.spec = spec
        self.dtype = dtype
        self.torch_dtype = torch.bfloat16 if dtype == "bf16" else torch.float32      # dtype of the INPUT tensors (x3 takes fp32)
        self.device = torch.device(device if device is not None else "cuda:0")
        self._holder = _DescHolder(spec, DTYPE_CODES[dtype])
        self._plan = C.c_void_p()
        with torch.cuda.device(self.device):
            _check(self.lib, self.lib.mshgnn_plan_create(C.byref(self._holder.desc), C.byref(self._plan)), "mshgnn_plan_create")
        self.info = MshgnnInfo()
        _check(self.lib, self.lib.mshgnn_plan_info(self._plan, C.byref(self.info)), "mshgnn_plan_info")
        # "" or the compile-time program of the one-call steps ("A1C2_L3", ...)
        self.specialised = self.lib.mshgnn_plan_specialised(self._plan).decode() if hasattr(self.lib, "mshgnn_plan_specialised") else ""
        # generic-width engine (hidden != 128, many nodes, ...: kernel_sets bit 2) and how activations are stored: "x3" = rows of
        # [hi | lo] bf16 halves (the split plan, and MSHGNN_F32 requests served by the generic engine's split arithmetic)
        self.generic = bool(self.info.kernel_sets & 4)
        self.storage = "x3" if (dtype == "x3" or (self.generic and dtype == "f32")) else dtype
        # MSHGNN_JIT=1: a topology the build has no compile-time program for gets one compiled now (hipcc, minutes, cached by table hash) -- morphsym_hgnn_amd/jit.py
        if os.environ.get("MSHGNN_JIT") == "1" and not self.specialised and dtype in ("bf16", "x3") and not self.generic and os.environ.get("MSHGNN_SPEC") != "0":
            from . import jit
            try:
                jit.attach_program(self)
            except Exception as ex:  # noqa: BLE001  (no hipcc, no sources: the interpreting kernels stay -- same results)
                import warnings
                warnings.warn(f"MSHGNN_JIT=1: no program compiled for this plan ({str(ex)[:300]}); the interpreting kernels are used")
        self._ws: Dict[Tuple[int, int], torch.Tensor] = {}
        self._tickets: Dict[int, int] = {}
        self._chunked: Dict[int, bool] = {}      # batch size -> the last training call on it was a one-call step (which the library may run as sub-steps)
        self._rows_gen: Dict[int, int] = {}      # batch size -> how many times the encoder has written the engine's input row buffers of that size
        self._lay: Dict[Tuple[int, int], MshgnnWsLayout] = {}
        self._rows: Dict[int, List[torch.Tensor]] = {}      # plan-dtype input rows the encoder materialises from wide source tensors (WideInputs)
        self._pending_wide: Optional[WideInputs] = None     # the last WideInputs handed out whose rows have not been materialised yet
        self.types = spec.node_types
        self.n_out = spec.num_nodes[spec.out_type]

    def __del__(self):
        try:
            if getattr(self, "_plan", None) is not None and self._plan.value:
                self.lib.mshgnn_plan_destroy(self._plan)
                self._plan = C.c_void_p()
        except Exception:
            pass

    # ---- memory ------------------------------------------------------------------------------
    def layout(self, B: int, training: bool = True) -> MshgnnWsLayout:
        key = (B, int(training))
        if key not in self._lay:
            lay = MshgnnWsLayout()
            _check(self.lib, self.lib.mshgnn_workspace_layout(self._plan, B, int(training), C.byref(lay)), "mshgnn_workspace_layout")
            self._lay[key] = lay
        return self._lay[key]

    MAX_WORKSPACES = 4      # distinct (batch size, training) workspaces kept alive; least recently used beyond that are released

    def workspace(self, B: int, training: bool = True) -> torch.Tensor:
        """The caller-owned activation stash of the C-ABI for batch size B (0.6 GB at B = 8192 on the bf16 plan).  One per (B, training),
        at most MAX_WORKSPACES of them: a training loop has one or two batch sizes (the last, ragged batch; validation), a sweep over
        batch sizes must not pin them all.  Evicting a workspace invalidates a forward whose backward has not run yet: its stash
        ticket is bumped, so that backward raises instead of reading freed memory."""
        key = (B, int(training))
        ws = self._ws.pop(key, None)
        if ws is None:
            lay = self.layout(B, training)
            ws = torch.empty(lay.total, dtype=torch.uint8, device=self.device)
            if os.environ.get("MSHGNN_POISON_WS") == "1":      # tests: a fresh workspace reads as NaNs / all-ones relu bytes, so that a kernel which reads what no launch
                ws.fill_(0xFF)                                 # wrote fails loudly instead of finding an earlier engine's values in recycled memory (tests/conftest.py sets it)
            while len(self._ws) >= self.MAX_WORKSPACES:
                old_key = next(iter(self._ws))
                del self._ws[old_key]
                if old_key[1] == 1:      # only a TRAINING workspace holds a stash some pending backward may still need
                    self._tickets[old_key[0]] = self._tickets.get(old_key[0], 0) + 1
        self._ws[key] = ws           # (re-inserted last: dict order = recency)
        return ws

    def padded_width(self, t: str) -> int:
        """Row pitch (elements) of input type t in engine layout (`row_pitch`: 16-byte aligned rows, long rows on a cache line)."""
        return row_pitch(self.spec.widths[t], 2 if self.dtype == "bf16" else 4)

    def cast_inputs(self, x_dict: Dict[str, torch.Tensor], pad: bool = True) -> List[torch.Tensor]:
        """Reference-convention inputs ([B*n_t, F_t], any float dtype) -> plan-dtype device tensors.  The cast writes
        rows at the engine's pitch (`row_pitch`: 450 -> 512 bf16 elements for A1 joints, 900 -> 960 for the base nodes), so the kernels
        stream them with 16-byte loads and every K chunk is whole cache lines; the pad columns are never read as data."""
        wide = self._wide_inputs(x_dict) if pad else None
        if wide is not None:
            return wide
        out = []
        for t in self.types:
            x = x_dict[t]
            F = self.spec.widths[t]
            P = self.padded_width(t) if pad else F
            if x.is_cuda and x.device == self.device and x.dtype == self.torch_dtype and x.is_contiguous() and x.dim() == 2 and x.shape[1] in accepted_pitches(F, x.element_size()):
                out.append(x)          # already what the kernels read (an unpadded width takes their element-wise loaders): no copy
            elif P == F:
                out.append(x.to(device=self.device, dtype=self.torch_dtype).contiguous())
            else:
                buf = torch.empty(x.shape[0], P, dtype=self.torch_dtype, device=self.device)
                buf[:, :F].copy_(x, non_blocking=True)       # one fused cast + re-pitch kernel (host tensors: plus the PCIe copy)
                buf[:, F:].zero_()                           # pad columns: never read as data by the kernels, and never NaN for a future reader
                out.append(buf)
        return out

    def _wide_inputs(self, x_dict) -> Optional[WideInputs]:
        """The no-cast route (bf16 and split plans of the LDS-resident kernels): every type arrives as a contiguous fp64 / fp32 tensor on this device
        that the kernels cannot take as it is (fp64; or fp32 whose dense rows are not whole 16-byte chunks) -> the encoder reads it directly
        (`mshgnn_forward_src` / `mshgnn_step_*_src`) and writes the plan-dtype rows into the engine's row buffers.  MSHGNN_WIDE_SRC=0: always cast."""
        if self.generic or self.dtype not in ("bf16", "x3") or os.environ.get("MSHGNN_WIDE_SRC", "1") == "0":
            return None
        xs = [x_dict[t] for t in self.types]
        dt = xs[0].dtype
        if dt not in (torch.float64, torch.float32):
            return None
        rows = None
        for t, x in zip(self.types, xs):
            F = self.spec.widths[t]
            if not (x.is_cuda and x.device == self.device and x.dtype == dt and x.dim() == 2 and x.is_contiguous() and x.shape[1] == F):
                return None
            if rows is None:
                rows = x.shape[0] // self.spec.num_nodes[t]
            need = 8 if (dt == torch.float64 or F % 2 == 0) else 4      # fp64 units, fp32 rows of an even width: 8-byte loads; odd fp32 widths: element loads
            if x.shape[0] != rows * self.spec.num_nodes[t] or x.data_ptr() % need:
                return None
        if dt == self.torch_dtype and all((self.spec.widths[t] * 4) % 16 == 0 for t in self.types):
            return None      # fp32 rows the kernels' 16-byte loaders take as they are
        bufs = self._row_buffers(rows)
        w = self._pending_wide
        if w is not None and w.pending and len(w) == len(bufs) and all(a.data_ptr() == b.data_ptr() for a, b in zip(w, bufs)):
            # a second batch of the same size before the first was run (prefetch, train + validation): both would name the same row buffers, and a forward on the
            # unpacked rows of the first could not be told from one on the second -- this one takes the cast pass (tensors of its own)
            return None
        self._pending_wide = WideInputs(bufs, xs, 8 if dt == torch.float64 else 4)
        return self._pending_wide

    def _rewrap(self, xs):
        """A caller that unpacked a pending WideInputs into its row-buffer tensors (autograd Functions take `*xs`) still means the wide route: the row
        buffers are the engine's own, recognised by address."""
        w = self._pending_wide
        if isinstance(xs, WideInputs) or w is None or not w.pending:
            return xs
        if len(xs) == len(w) and all(a.data_ptr() == b.data_ptr() for a, b in zip(xs, w)):
            return w
        return xs

    def _rows_written(self, xs: "WideInputs", B: int):
        """The encoder is about to (re)write the engine's row buffers of batch size B from `xs.src`: whatever was materialised there before -- an earlier
        WideInputs of this size, and the rows a pending autograd backward of this size would read for the encoder's weight gradients -- is gone.  The stash
        ticket is bumped whatever `training` is (an evaluation forward on other data between a training forward and its backward must make that backward
        raise, not read the wrong rows), and `xs` remembers the generation it was written as."""
        self._tickets[B] = self._tickets.get(B, 0) + 1
        self._rows_gen[B] = self._rows_gen.get(B, 0) + 1
        xs.gen = self._rows_gen[B]

    def _check_rows_fresh(self, xs, B: int):
        """A materialised WideInputs is only as good as the row buffers it points into: refuse it once another batch of the same size has been written there."""
        if isinstance(xs, WideInputs) and not xs.pending and xs.gen != self._rows_gen.get(B, 0):
            raise RuntimeError(f"these inputs were converted into the engine's row buffers for batch size {B}, which a later batch of the same size has since "
                               "overwritten -- call cast_inputs again (or keep your own cast: MSHGNN_WIDE_SRC=0)")

    def _row_buffers(self, B: int) -> List[torch.Tensor]:
        """Plan-dtype input rows at the engine's pitch for batch size B, owned by the engine (zeroed once: pad columns and the rows of nodes the plan
        never reads stay zero); kept for the batch sizes that have a workspace."""
        bufs = self._rows.get(B)
        if bufs is None:
            bufs = [torch.zeros(B * self.spec.num_nodes[t], self.padded_width(t), dtype=self.torch_dtype, device=self.device) for t in self.types]
            while len(self._rows) >= self.MAX_WORKSPACES:
                del self._rows[next(iter(self._rows))]
            self._rows[B] = bufs
        return bufs

    def _src_args(self, xs: "WideInputs"):
        n = len(xs.src)
        sp, spitch = (C.c_void_p * n)(), (C.c_int64 * n)()
        for i, x in enumerate(xs.src):
            sp[i], spitch[i] = x.data_ptr(), x.shape[1]
        return xs.src_bytes, sp, spitch

    def _xptrs(self, xs: Sequence[torch.Tensor], B: int):
        ptrs = (C.c_void_p * len(xs))()
        pitch = (C.c_int64 * len(xs))()
        for i, (t, x) in enumerate(zip(self.types, xs)):
            if x.dtype != self.torch_dtype or not x.is_cuda or not x.is_contiguous():
                raise ValueError(f"input '{t}' must be a contiguous {self.torch_dtype} device tensor")
            rows = B * self.spec.num_nodes[t]
            if x.numel() % rows or x.numel() // rows < self.spec.widths[t]:
                raise ValueError(f"input '{t}' has {x.numel()} elements, expected {rows} rows of >= {self.spec.widths[t]}")
            ptrs[i] = x.data_ptr()
            pitch[i] = x.numel() // rows
        return ptrs, pitch

    def _check_flat(self, flat: torch.Tensor, name: str):
        if flat.dtype != torch.float32 or not flat.is_cuda or not flat.is_contiguous() or flat.numel() != self.spec.flat_size():
            raise ValueError(f"{name} must be a contiguous fp32 device tensor of {self.spec.flat_size()} elements")

    # ---- hot path ------------------------------------------------------------------------------
    def forward(self, xs: Sequence[torch.Tensor], params_flat: torch.Tensor, B: int, training: bool = True,
                out: Optional[torch.Tensor] = None) -> torch.Tensor:
        xs = self._rewrap(xs)
        self._check_rows_fresh(xs, B)
        self._check_flat(params_flat, "params_flat")
        ptrs, pitch = self._xptrs(xs, B)
        if out is None:
            out = torch.empty(B * self.n_out, self.spec.out_channels, dtype=torch.float32, device=self.device)
        ws = self.workspace(B, training)
        if training:
            self._tickets[B] = self._tickets.get(B, 0) + 1; self._chunked[B] = False
        stream = torch.cuda.current_stream(self.device).cuda_stream
        with torch.cuda.device(self.device):
            if isinstance(xs, WideInputs) and xs.pending:      # the caller's fp64 / fp32 tensors: the encoder converts them and writes the plan-dtype rows `xs` holds
                self._rows_written(xs, B)
                sb, sp, spitch = self._src_args(xs)
                _check(self.lib, self.lib.mshgnn_forward_src(self._plan, sb, sp, spitch, ptrs, pitch, params_flat.data_ptr(), out.data_ptr(),
                                                             ws.data_ptr(), B, int(training), stream), "mshgnn_forward_src")
                xs.pending = False
            else:
                _check(self.lib, self.lib.mshgnn_forward(self._plan, ptrs, pitch, params_flat.data_ptr(), out.data_ptr(),
                                                         ws.data_ptr(), B, int(training), stream), "mshgnn_forward")
        return out

    def backward(self, xs: Sequence[torch.Tensor], params_flat: torch.Tensor, grad_out: torch.Tensor, B: int,
                 grad_flat: Optional[torch.Tensor] = None) -> torch.Tensor:
        self._check_rows_fresh(xs, B)
        self._check_flat(params_flat, "params_flat")
        ptrs, pitch = self._xptrs(xs, B)
        if grad_out.dtype != torch.float32 or not grad_out.is_contiguous() or grad_out.numel() != B * self.n_out * self.spec.out_channels:
            raise ValueError("grad_out must be contiguous fp32 with B*n_out*out_channels elements")
        if grad_flat is None:
            grad_flat = torch.empty(self.spec.flat_size(), dtype=torch.float32, device=self.device)
        else:
            self._check_flat(grad_flat, "grad_flat")
        ws = self.workspace(B, True)
        stream = torch.cuda.current_stream(self.device).cuda_stream
        with torch.cuda.device(self.device):
            _check(self.lib, self.lib.mshgnn_backward(self._plan, ptrs, pitch, params_flat.data_ptr(), grad_out.data_ptr(),
                                                  grad_flat.data_ptr(), ws.data_ptr(), B, stream), "mshgnn_backward")
        return grad_flat

    def backward_mse(self, xs: Sequence[torch.Tensor], params_flat: torch.Tensor, out: torch.Tensor, y: torch.Tensor, B: int,
                     grad_flat: Optional[torch.Tensor] = None, loss: Optional[torch.Tensor] = None):
        """Fused wrapper-MSE + backward (gnnLightning.py:633-639 + autograd): returns (loss[1], grad_flat)."""
        self._check_rows_fresh(xs, B)
        self._check_flat(params_flat, "params_flat")
        ptrs, pitch = self._xptrs(xs, B)
        n = B * self.n_out * self.spec.out_channels
        if out.dtype != torch.float32 or y.dtype != torch.float32 or out.numel() != n or y.numel() != n:
            raise ValueError("out and y must be fp32 with B*n_out*out_channels elements")
        if grad_flat is None:
            grad_flat = torch.empty(self.spec.flat_size(), dtype=torch.float32, device=self.device)
        if loss is None:
            loss = torch.empty(1, dtype=torch.float32, device=self.device)
        ws = self.workspace(B, True)
        stream = torch.cuda.current_stream(self.device).cuda_stream
        with torch.cuda.device(self.device):
            _check(self.lib, self.lib.mshgnn_backward_mse(self._plan, ptrs, pitch, params_flat.data_ptr(), out.data_ptr(), y.data_ptr(),
                                                      loss.data_ptr(), grad_flat.data_ptr(), ws.data_ptr(), B, stream), "mshgnn_backward_mse")
        return loss, grad_flat

    def step_mse(self, xs: Sequence[torch.Tensor], params_flat: torch.Tensor, y: torch.Tensor, B: int, out: Optional[torch.Tensor] = None,
                 grad_flat: Optional[torch.Tensor] = None, loss: Optional[torch.Tensor] = None):
        """One training step of the regression wrappers in one call (forward + MSE + backward, mshgnn_step_mse):
        returns (out, loss[1], grad_flat)."""
        xs = self._rewrap(xs)
        self._check_rows_fresh(xs, B)
        self._check_flat(params_flat, "params_flat")
        ptrs, pitch = self._xptrs(xs, B)
        n = B * self.n_out * self.spec.out_channels
        if y.dtype != torch.float32 or y.numel() != n or not y.is_contiguous():
            raise ValueError("y must be contiguous fp32 with B*n_out*out_channels elements")
        if out is None:
            out = torch.empty(B * self.n_out, self.spec.out_channels, dtype=torch.float32, device=self.device)
        if grad_flat is None:
            grad_flat = torch.empty(self.spec.flat_size(), dtype=torch.float32, device=self.device)
        if loss is None:
            loss = torch.empty(1, dtype=torch.float32, device=self.device)
        ws = self.workspace(B, True)
        self._tickets[B] = self._tickets.get(B, 0) + 1; self._chunked[B] = True      # the activation stash of this batch size is overwritten
        stream = torch.cuda.current_stream(self.device).cuda_stream
        with torch.cuda.device(self.device):
            if isinstance(xs, WideInputs) and xs.pending:
                self._rows_written(xs, B)
                sb, sp, spitch = self._src_args(xs)
                _check(self.lib, self.lib.mshgnn_step_mse_src(self._plan, sb, sp, spitch, ptrs, pitch, params_flat.data_ptr(), y.data_ptr(), out.data_ptr(),
                                                              loss.data_ptr(), grad_flat.data_ptr(), ws.data_ptr(), B, stream), "mshgnn_step_mse_src")
                xs.pending = False
            else:
                _check(self.lib, self.lib.mshgnn_step_mse(self._plan, ptrs, pitch, params_flat.data_ptr(), y.data_ptr(), out.data_ptr(), loss.data_ptr(),
                                                          grad_flat.data_ptr(), ws.data_ptr(), B, stream), "mshgnn_step_mse")
        return out, loss, grad_flat

    def step_ce(self, xs: Sequence[torch.Tensor], params_flat: torch.Tensor, labels: torch.Tensor, B: int, out: Optional[torch.Tensor] = None,
                grad_flat: Optional[torch.Tensor] = None, loss: Optional[torch.Tensor] = None):
        """One training step of the classification wrappers in one call (forward + cross entropy + backward, mshgnn_step_ce):
        returns (out, loss[1], grad_flat).  labels: int32 [B, n_out] in {0, 1}."""
        xs = self._rewrap(xs)
        self._check_rows_fresh(xs, B)
        self._check_flat(params_flat, "params_flat")
        ptrs, pitch = self._xptrs(xs, B)
        if labels.dtype != torch.int32 or labels.numel() != B * self.n_out or not labels.is_contiguous():
            raise ValueError("labels must be contiguous int32 with B*n_out elements")
        if out is None:
            out = torch.empty(B * self.n_out, self.spec.out_channels, dtype=torch.float32, device=self.device)
        if grad_flat is None:
            grad_flat = torch.empty(self.spec.flat_size(), dtype=torch.float32, device=self.device)
        if loss is None:
            loss = torch.empty(1, dtype=torch.float32, device=self.device)
        ws = self.workspace(B, True)
        self._tickets[B] = self._tickets.get(B, 0) + 1; self._chunked[B] = True
        stream = torch.cuda.current_stream(self.device).cuda_stream
        with torch.cuda.device(self.device):
            if isinstance(xs, WideInputs) and xs.pending:
                self._rows_written(xs, B)
                sb, sp, spitch = self._src_args(xs)
                _check(self.lib, self.lib.mshgnn_step_ce_src(self._plan, sb, sp, spitch, ptrs, pitch, params_flat.data_ptr(), labels.data_ptr(), out.data_ptr(),
                                                             loss.data_ptr(), grad_flat.data_ptr(), ws.data_ptr(), B, stream), "mshgnn_step_ce_src")
                xs.pending = False
            else:
                _check(self.lib, self.lib.mshgnn_step_ce(self._plan, ptrs, pitch, params_flat.data_ptr(), labels.data_ptr(), out.data_ptr(), loss.data_ptr(),
                                                         grad_flat.data_ptr(), ws.data_ptr(), B, stream), "mshgnn_step_ce")
        return out, loss, grad_flat

    def step_ce_series(self, store, starts: torch.Tensor, params_flat: torch.Tensor, out: Optional[torch.Tensor] = None,
                       grad_flat: Optional[torch.Tensor] = None, loss: Optional[torch.Tensor] = None, materialize: bool = True):
        """`step_mse_series` for the classification wrappers (mshgnn_step_ce_series): the labels are the contact flags of each window's last
        step, cross entropy over the per-foot logit pairs.  Returns (xs | None, labels int32 [B, n_out], out, loss[1], grad_flat) -- bit-identical
        to `store.assemble(starts)` followed by `step_ce`."""
        return self._step_series(True, store, starts, params_flat, out, grad_flat, loss, materialize)

    def step_mse_series(self, store, starts: torch.Tensor, params_flat: torch.Tensor, out: Optional[torch.Tensor] = None,
                        grad_flat: Optional[torch.Tensor] = None, loss: Optional[torch.Tensor] = None, materialize: bool = True):
        """One training step straight from a `windows.SequenceStore` (mshgnn_step_mse_series): the encoder gathers its raw inputs from the
        sequence's resident series and (materialize=True, the fast route) writes the batch's windows on the side for the weight-gradient pass;
        materialize=False keeps no windows at all -- the weight-gradient kernel gathers from the series too (same bits, no 118 MB buffer, but
        measured slower: its random 256-byte gathers stretch the raw-input lanes' steps, 0.385 vs 0.557 ms/step).  starts: device int64 window
        start rows.  Returns (xs | None, y, out, loss[1], grad_flat) -- bit-identical to `store.assemble(starts)` followed by `step_mse`.
        bf16 plan with the fused stack kernels (a bf16 store), or the split plan "x3" (an fp32 store: `SequenceStore(..., dtype="x3")`; its
        encoder gathers from the fp32 series and always materialises the windows)."""
        return self._step_series(False, store, starts, params_flat, out, grad_flat, loss, materialize)

    def _step_series(self, ce: bool, store, starts, params_flat, out, grad_flat, loss, materialize):
        self._check_flat(params_flat, "params_flat")
        if not starts.is_cuda or starts.dtype != torch.int64:
            raise ValueError("starts must be a device int64 tensor")
        B = int(starts.numel())
        r = store.recipe
        xs, y, q = store._buffers(B)
        if y is None:
            raise ValueError("the recipe has no labels")
        if out is None:
            out = torch.empty(B * self.n_out, self.spec.out_channels, dtype=torch.float32, device=self.device)
        if grad_flat is None:
            grad_flat = torch.empty(self.spec.flat_size(), dtype=torch.float32, device=self.device)
        if loss is None:
            loss = torch.empty(1, dtype=torch.float32, device=self.device)
        src16, run_ptrs = store.series_step_args(bf16=self.storage != "x3")
        if self.storage == "x3" and not materialize:
            raise ValueError("the split plan's weight-gradient kernel reads materialised windows: materialize=False is a bf16-plan option")
        xp = (C.c_void_p * len(xs))(*[x.data_ptr() for x in xs]) if materialize else None
        pitch = (C.c_int64 * len(xs))(*[x.shape[1] for x in xs]) if materialize else None
        if not materialize:
            xs = None
        ws = self.workspace(B, True)
        self._tickets[B] = self._tickets.get(B, 0) + 1; self._chunked[B] = True
        stream = torch.cuda.current_stream(self.device).cuda_stream
        # the runs' column pointers in the store's scratch depend only on the series' addresses and the element size: resolved by the first step of this
        # store on this stream at this storage, vouched for afterwards (mshgnn_window_desc.run_ptrs_ready: one launch less in front of every encoder)
        key = (self.storage == "x3", stream, tuple(a.data_ptr() for a in store.series))
        store.desc.run_ptrs_ready = 1 if getattr(store, "_run_ptrs_key", None) == key else 0
        store._run_ptrs_key = key
        with torch.cuda.device(self.device):
            if ce:
                labels = torch.empty(B, self.n_out, dtype=torch.int32, device=self.device)
                _check(self.lib, self.lib.mshgnn_step_ce_series(self._plan, C.byref(store.desc), store._src, src16, store._pitch, store._rows, starts.data_ptr(), B,
                                                            xp, pitch, y.data_ptr(), labels.data_ptr(), run_ptrs.data_ptr(),
                                                            params_flat.data_ptr(), out.data_ptr(), loss.data_ptr(), grad_flat.data_ptr(), ws.data_ptr(), stream),
                       "mshgnn_step_ce_series")
                return xs, labels, out, loss, grad_flat
            _check(self.lib, self.lib.mshgnn_step_mse_series(self._plan, C.byref(store.desc), store._src, src16, store._pitch, store._rows, starts.data_ptr(), B,
                                                         xp, pitch, y.data_ptr(), q.data_ptr() if q is not None else None, run_ptrs.data_ptr(),
                                                         params_flat.data_ptr(), out.data_ptr(), loss.data_ptr(), grad_flat.data_ptr(), ws.data_ptr(), stream),
                   "mshgnn_step_mse_series")
        return xs, y, out, loss, grad_flat

    def step_mse_phase(self, phase: int, xs: Sequence[torch.Tensor], params_flat: torch.Tensor, y: torch.Tensor, B: int, out: torch.Tensor,
                       grad_flat: torch.Tensor, loss: torch.Tensor):
        """mshgnn_step_mse in two calls (phase 0, then 1): after phase 0 grad_flat[self.info.grad_split:] and loss are final,
        after phase 1 the encoder's gradients grad_flat[:grad_split] -- the caller puts the all-reduce of the first region
        in between."""
        ptrs, pitch = self._xptrs(xs, B)
        ws = self.workspace(B, True)
        if phase == 0:
            self._tickets[B] = self._tickets.get(B, 0) + 1
        stream = torch.cuda.current_stream(self.device).cuda_stream
        with torch.cuda.device(self.device):
            _check(self.lib, self.lib.mshgnn_step_mse_phase(self._plan, ptrs, pitch, params_flat.data_ptr(), y.data_ptr(), out.data_ptr(),
                                                        loss.data_ptr(), grad_flat.data_ptr(), ws.data_ptr(), B, phase, stream), "mshgnn_step_mse_phase")

    def backward_ce(self, xs: Sequence[torch.Tensor], params_flat: torch.Tensor, out: torch.Tensor, labels: torch.Tensor, B: int,
                    grad_flat: Optional[torch.Tensor] = None, loss: Optional[torch.Tensor] = None):
        """Fused wrapper cross entropy + backward (gnnLightning.py:640-648 + autograd): labels int32 [B, n_out] in {0,1};
        returns (loss[1], grad_flat)."""
        self._check_rows_fresh(xs, B)
        self._check_flat(params_flat, "params_flat")
        ptrs, pitch = self._xptrs(xs, B)
        if out.dtype != torch.float32 or out.numel() != B * self.n_out * 2 or labels.dtype != torch.int32 or labels.numel() != B * self.n_out:
            raise ValueError("out must be fp32 [B*n_out, 2] and labels int32 [B, n_out]")
        if grad_flat is None:
            grad_flat = torch.empty(self.spec.flat_size(), dtype=torch.float32, device=self.device)
        if loss is None:
            loss = torch.empty(1, dtype=torch.float32, device=self.device)
        ws = self.workspace(B, True)
        stream = torch.cuda.current_stream(self.device).cuda_stream
        with torch.cuda.device(self.device):
            _check(self.lib, self.lib.mshgnn_backward_ce(self._plan, ptrs, pitch, params_flat.data_ptr(), out.data_ptr(), labels.data_ptr(),
                                                     loss.data_ptr(), grad_flat.data_ptr(), ws.data_ptr(), B, stream), "mshgnn_backward_ce")
        return loss, grad_flat

    def adam_step(self, params_flat: torch.Tensor, grad_flat: torch.Tensor, exp_avg: torch.Tensor, exp_avg_sq: torch.Tensor,
                  step: int, lr: float, betas=(0.9, 0.999), eps: float = 1e-8, grad_scale: float = 1.0):
        """torch.optim.Adam semantics on the flat buffers, in place (gnnLightning.py:258-265)."""
        for t, n in ((params_flat, "params"), (grad_flat, "grads"), (exp_avg, "exp_avg"), (exp_avg_sq, "exp_avg_sq")):
            self._check_flat(t, n)
        stream = torch.cuda.current_stream(self.device).cuda_stream
        with torch.cuda.device(self.device):
            _check(self.lib, self.lib.mshgnn_adam_step(params_flat.data_ptr(), grad_flat.data_ptr(), exp_avg.data_ptr(), exp_avg_sq.data_ptr(),
                                                   params_flat.numel(), step, lr, betas[0], betas[1], eps, grad_scale, stream), "mshgnn_adam_step")

    def mse_loss(self, out: torch.Tensor, y: torch.Tensor, want_grad: bool = True):
        """Wrapper loss (gnnLightning.py:633-639): returns (loss[1], dL/d out or None)."""
        n = out.numel()
        if y.numel() != n or y.dtype != torch.float32 or out.dtype != torch.float32:
            raise ValueError("out and y must be fp32 tensors with the same number of elements")
        loss = torch.empty(1, dtype=torch.float32, device=self.device)
        g = torch.empty_like(out) if want_grad else None
        stream = torch.cuda.current_stream(self.device).cuda_stream
        with torch.cuda.device(self.device):
            _check(self.lib, self.lib.mshgnn_mse_loss(out.data_ptr(), y.data_ptr(), n, loss.data_ptr(),
                                                  g.data_ptr() if g is not None else None, stream), "mshgnn_mse_loss")
        return loss, g

    def stash_ticket(self, B: int) -> int:
        """Monotonic id of the last training forward that wrote the activation stash for batch size B."""
        return self._tickets.get(B, 0)

    # ---- per-kernel timing -----------------------------------------------------------------------
    def profile(self, on: bool):
        _check(self.lib, self.lib.mshgnn_profile_enable(self._plan, int(on)), "mshgnn_profile_enable")

    def profile_read(self) -> List[dict]:
        """Per-kernel HIP-event timings accumulated since the last read, with each kernel's algorithmic work."""
        n = C.c_int32(64)
        arr = (MshgnnKernelStat * 64)()
        _check(self.lib, self.lib.mshgnn_profile_read(self._plan, arr, C.byref(n)), "mshgnn_profile_read")
        out = []
        for i in range(n.value):
            k = arr[i]
            out.append(dict(name=k.name.decode(), launches=k.launches, bound="mfma" if k.bound == 1 else "hbm",
                            total_ms=float(k.total_ms), flops_per_window=k.flops_per_window,
                            flops_exec_per_window=k.flops_exec_per_window, bytes_per_window=k.bytes_per_window))
        return out

    # ---- introspection (tests) -----------------------------------------------------------------
    def _act_tensor(self, off: int, B: int) -> torch.Tensor:
        """An activation tensor of the workspace as [B, NN, hidden] (node-major [NN, B, hidden] in memory; the split plan stores rows
        of [hi | lo] bf16 halves, returned as their fp32 sum)."""
        ws = self.workspace(B, True)
        nn_ = self.info.total_nodes
        n = B * nn_ * self.spec.hidden
        if self.storage == "x3":
            halves = ws[off:off + 4 * n].view(torch.bfloat16).view(nn_, B, 2, self.spec.hidden).float()
            return (halves[:, :, 0] + halves[:, :, 1]).permute(1, 0, 2)
        es = 4 if self.storage == "f32" else 2
        return ws[off:off + n * es].view(torch.float32 if self.storage == "f32" else torch.bfloat16).view(nn_, B, self.spec.hidden).permute(1, 0, 2)

    def _stash_is_whole(self, B: int):
        """One-call steps over >= 2 x MSHGNN_STEP_CHUNK windows (default 32 768) run as sub-steps on the FRONT of the workspace (include/mshgnn.h, StepChunk): what
        the stash holds afterwards is the last sub-step's rows in a sub-batch layout, not the batch's -- the introspection below would read garbage."""
        chunk = int(os.environ.get("MSHGNN_STEP_CHUNK", "32768"))
        if not self.generic and chunk > 0 and B >= 2 * chunk and self._chunked.get(B):
            raise RuntimeError(f"the last one-call step of {B} windows ran as sub-steps of >= {chunk}: the workspace holds only the last sub-step "
                               "(MSHGNN_STEP_CHUNK=0 keeps whole-batch steps)")

    def hidden_state(self, B: int, layer: int) -> torch.Tensor:
        """X_layer as [B, NN, hidden]."""
        self._stash_is_whole(B)
        return self._act_tensor(self.layout(B, True).x[layer], B)

    def grad_hidden(self, B: int, layer: int) -> torch.Tensor:
        self._stash_is_whole(B)
        return self._act_tensor(self.layout(B, True).dx[layer], B)


class PaddedEngine:
    """A model whose hidden width is NOT a multiple of 128 (the reference takes any `hidden_channels`, hgnn_c2.py:11; its scripts expose it as
    --hidden_size) on the fused engines: the same model at the next multiple of 128 with every added weight row / column and bias element zero.
    The added features are then exactly zero in every layer (zero rows + zero bias -> relu(0) = 0, the residual adds 0 to 0, base_transform maps 0
    to 0 through zero rows), contribute exact zeros to every product they enter, and receive exact-zero activation gradients (their columns of the
    next weight are zero and relu'(0) = 0): outputs, loss and the gradients of the REAL parameters are those of the unpadded model -- to fp32
    summation order -- on the same kernels and at the same parity as a 128-wide model.  The caller sees the true layout (state_dict order, true shapes,
    `spec.flat_size()` elements); the parameters are scattered into the padded flat buffer before a call and the gradient gathered out of the padded
    one after it (two index kernels each way, ~4 MB)."""

    padded = True

    def __init__(self, spec: ModelSpec, dtype: str = "f32", device: Optional[torch.device] = None):
        import dataclasses
        self.spec = spec
        hp = (spec.hidden + 127) // 128 * 128
        self.inner_spec = dataclasses.replace(spec, hidden=hp)
        self.inner = Engine(self.inner_spec, dtype=dtype, device=device)
        e = self.inner
        self.lib, self.device, self.dtype, self.torch_dtype, self.types, self.n_out = e.lib, e.device, e.dtype, e.torch_dtype, e.types, e.n_out
        self.generic, self.storage, self.info = e.generic, e.storage, e.info
        true_pos, pad_pos = [], []
        shp_t, shp_p, off_t, off_p = spec.param_shapes(), self.inner_spec.param_shapes(), spec.param_offsets(), self.inner_spec.param_offsets()
        for k, st in shp_t.items():
            sp, (ot, nt_), (op, _) = shp_p[k], off_t[k], off_p[k]
            true_pos.append(torch.arange(ot, ot + nt_, dtype=torch.int64))
            if len(st) == 1:
                pad_pos.append(op + torch.arange(st[0], dtype=torch.int64))
            else:
                pad_pos.append((op + torch.arange(st[0], dtype=torch.int64)[:, None] * sp[1] + torch.arange(st[1], dtype=torch.int64)[None, :]).reshape(-1))
        self._true_pos = torch.cat(true_pos).to(self.device)
        self._pad_pos = torch.cat(pad_pos).to(self.device)
        self._pflat = torch.zeros(self.inner_spec.flat_size(), dtype=torch.float32, device=self.device)      # everything outside _pad_pos stays zero for ever
        self._pgrad = torch.empty_like(self._pflat)
        self._tmp = torch.empty(self._true_pos.numel(), dtype=torch.float32, device=self.device)

    # ---- layout translation ----
    def _pad(self, params_flat: torch.Tensor) -> torch.Tensor:
        if params_flat.dtype != torch.float32 or not params_flat.is_cuda or params_flat.numel() != self.spec.flat_size():
            raise ValueError(f"params_flat must be a contiguous fp32 device tensor of {self.spec.flat_size()} elements")
        torch.index_select(params_flat, 0, self._true_pos, out=self._tmp)
        self._pflat.index_copy_(0, self._pad_pos, self._tmp)
        return self._pflat

    def _unpad_grad(self, grad_flat: Optional[torch.Tensor]) -> torch.Tensor:
        if grad_flat is None:
            grad_flat = torch.empty(self.spec.flat_size(), dtype=torch.float32, device=self.device)
        torch.index_select(self._pgrad, 0, self._pad_pos, out=self._tmp)
        grad_flat.zero_()
        grad_flat.index_copy_(0, self._true_pos, self._tmp)
        return grad_flat

    # ---- the Engine surface models.py / wrappers.py / tests use ----
    def cast_inputs(self, x_dict, pad: bool = True):
        return self.inner.cast_inputs(x_dict, pad)

    def workspace(self, B, training=True):
        return self.inner.workspace(B, training)

    def layout(self, B, training=True):
        return self.inner.layout(B, training)

    def stash_ticket(self, B):
        return self.inner.stash_ticket(B)

    def forward(self, xs, params_flat, B, training=True, out=None):
        return self.inner.forward(xs, self._pad(params_flat), B, training=training, out=out)

    def backward(self, xs, params_flat, grad_out, B, grad_flat=None):
        self.inner.backward(xs, self._pad(params_flat), grad_out, B, grad_flat=self._pgrad)
        return self._unpad_grad(grad_flat)

    def backward_mse(self, xs, params_flat, out, y, B, grad_flat=None, loss=None):
        loss, _ = self.inner.backward_mse(xs, self._pad(params_flat), out, y, B, grad_flat=self._pgrad, loss=loss)
        return loss, self._unpad_grad(grad_flat)

    def backward_ce(self, xs, params_flat, out, labels, B, grad_flat=None, loss=None):
        loss, _ = self.inner.backward_ce(xs, self._pad(params_flat), out, labels, B, grad_flat=self._pgrad, loss=loss)
        return loss, self._unpad_grad(grad_flat)

    def step_mse(self, xs, params_flat, y, B, out=None, grad_flat=None, loss=None):
        out, loss, _ = self.inner.step_mse(xs, self._pad(params_flat), y, B, out=out, grad_flat=self._pgrad, loss=loss)
        return out, loss, self._unpad_grad(grad_flat)

    def step_ce(self, xs, params_flat, labels, B, out=None, grad_flat=None, loss=None):
        out, loss, _ = self.inner.step_ce(xs, self._pad(params_flat), labels, B, out=out, grad_flat=self._pgrad, loss=loss)
        return out, loss, self._unpad_grad(grad_flat)

    def mse_loss(self, out, y, want_grad=True):
        return self.inner.mse_loss(out, y, want_grad)

    def adam_step(self, params_flat, grad_flat, exp_avg, exp_avg_sq, step, lr, betas=(0.9, 0.999), eps=1e-8, grad_scale=1.0):
        """Adam is elementwise: it runs on the caller's TRUE-size flat buffers as they are (the inner engine's own check would demand the padded size)."""
        n = self.spec.flat_size()
        for t, name in ((params_flat, "params"), (grad_flat, "grads"), (exp_avg, "exp_avg"), (exp_avg_sq, "exp_avg_sq")):
            if t.dtype != torch.float32 or not t.is_cuda or not t.is_contiguous() or t.numel() != n:
                raise ValueError(f"{name} must be a contiguous fp32 device tensor of {n} elements")
        stream = torch.cuda.current_stream(self.device).cuda_stream
        with torch.cuda.device(self.device):
            _check(self.inner.lib, self.inner.lib.mshgnn_adam_step(params_flat.data_ptr(), grad_flat.data_ptr(), exp_avg.data_ptr(), exp_avg_sq.data_ptr(),
                                                                   n, step, lr, betas[0], betas[1], eps, grad_scale, stream), "mshgnn_adam_step")

    def padded_width(self, t):
        return self.inner.padded_width(t)

    def step_mse_phase(self, phase, xs, params_flat, y, B, out, grad_flat, loss):
        """Two-phase step (multi-GPU overlap): phase 0 pads the parameters and runs the inner phase on the padded gradient buffer, phase 1 finishes it and unpads."""
        r = self.inner.step_mse_phase(phase, xs, self._pad(params_flat) if phase == 0 else self._pflat, y, B, out, self._pgrad, loss)
        if phase == 1:
            self._unpad_grad(grad_flat)
        return r

    def hidden_state(self, B, layer):
        return self.inner.hidden_state(B, layer)[..., :self.spec.hidden]

    def grad_hidden(self, B, layer):
        return self.inner.grad_hidden(B, layer)[..., :self.spec.hidden]

    def profile(self, on):
        return self.inner.profile(on)

    def profile_read(self):
        return self.inner.profile_read()


def make_engine(spec: ModelSpec, dtype: str = "f32", device: Optional[torch.device] = None):
    """Engine for any hidden width: multiples of 128 as they are, other widths zero-padded to the next one (PaddedEngine)."""
    return Engine(spec, dtype=dtype, device=device) if spec.hidden % 128 == 0 else PaddedEngine(spec, dtype=dtype, device=device)
