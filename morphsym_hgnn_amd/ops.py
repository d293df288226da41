"""Stand-alone operators of the hot path on the MI355X, through the C-ABI (include/mshgnn.h, `mshgnn_op_*`): what the four
`torch_geometric.nn` modules of nn.py compute when they are called on their own -- `Linear` / `HeteroDictLinear`
(hgnn_c2.py:88,131), `GraphConv` (hgnn_c2.py:100-112: aggregate over in-edges, `lin_rel`, `+ lin_root`), summed per
destination type by `HeteroConv` (hgnn_c2.py:113) -- with autograd, on arbitrary graphs and widths.

The model classes of models.py do not come through here (they hand a whole minibatch to the fused engine); this is the
path of a maintainer who swaps only the PyG import.  Arithmetic: fp32 operands, fp32 MFMA accumulation in HIP kernels
(csrc/mshgnn_ops.hip), deterministic (no float atomics).  Tensors of another floating dtype (the reference runs fp64,
gnnLightning.py:1183) are cast in and the results cast back.  There is no CPU implementation: tensors that are not on
a HIP device raise.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Tuple

import torch

from . import engine as eng


def _need_gpu(t: torch.Tensor, what: str):
    if not t.is_cuda:
        raise RuntimeError(f"{what}: the MS-HGNN operators run on a HIP device (tensor is on {t.device}); there is no CPU fallback")


def _stream(dev) -> C.c_void_p:
    return C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)


def _f32(t: torch.Tensor) -> torch.Tensor:
    return t.detach().to(torch.float32).contiguous()


def gemm(A: torch.Tensor, sAm: int, sAk: int, B: torch.Tensor, sBn: int, sBk: int, M: int, N: int, K: int,
         bias: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None, accumulate: bool = False) -> torch.Tensor:
    """out[m, n] (+)= sum_k A[m sAm + k sAk] B[n sBn + k sBk] (+ bias[n]) on fp32 device buffers (mshgnn_op_gemm)."""
    lib = eng.load_library()
    dev = A.device
    if out is None:
        out = torch.empty(M, N, dtype=torch.float32, device=dev)
    need = lib.mshgnn_op_gemm_workspace(M, N, K, None)
    ws = torch.empty(need, dtype=torch.uint8, device=dev) if need else None
    with torch.cuda.device(dev):
        eng._check(lib, lib.mshgnn_op_gemm(A.data_ptr(), sAm, sAk, B.data_ptr(), sBn, sBk, bias.data_ptr() if bias is not None else None,
                                           out.data_ptr(), out.stride(0) if out.dim() == 2 else N, M, N, K, int(accumulate),
                                           ws.data_ptr() if ws is not None else None, _stream(dev)), "mshgnn_op_gemm")
    return out


def colsum(X: torch.Tensor) -> torch.Tensor:
    """Column sums of a contiguous fp32 [M, N] device matrix (bias gradient), fixed summation order."""
    lib = eng.load_library()
    M, N = X.shape
    out = torch.empty(N, dtype=torch.float32, device=X.device)
    ws = torch.empty(max(1, lib.mshgnn_op_colsum_workspace(M, N)), dtype=torch.uint8, device=X.device)
    with torch.cuda.device(X.device):
        eng._check(lib, lib.mshgnn_op_colsum(X.data_ptr(), N, out.data_ptr(), M, N, ws.data_ptr(), _stream(X.device)), "mshgnn_op_colsum")
    return out


class _Linear(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias):
        _need_gpu(x, "Linear")
        _need_gpu(weight, "Linear")
        out_f, in_f = weight.shape
        if x.shape[-1] != in_f:
            raise RuntimeError(f"Linear: input has {x.shape[-1]} features, the weight expects {in_f}")
        x2, w = _f32(x.reshape(-1, in_f)), _f32(weight)
        b = _f32(bias) if bias is not None else None
        y = gemm(x2, in_f, 1, w, in_f, 1, x2.shape[0], out_f, in_f, bias=b)
        ctx.save_for_backward(x2, w)
        ctx.meta = (x.shape, x.dtype, weight.dtype, bias.dtype if bias is not None else None)
        return y.to(x.dtype).reshape(*x.shape[:-1], out_f)

    @staticmethod
    def backward(ctx, gy):
        x2, w = ctx.saved_tensors
        xshape, xdt, wdt, bdt = ctx.meta
        out_f, in_f = w.shape
        g = _f32(gy.reshape(-1, out_f))
        M = g.shape[0]
        gx = gw = gb = None
        if ctx.needs_input_grad[0]:
            gx = gemm(g, out_f, 1, w, 1, in_f, M, in_f, out_f).to(xdt).reshape(xshape)           # dx = dy W
        if ctx.needs_input_grad[1]:
            gw = gemm(g, 1, out_f, x2, 1, in_f, out_f, in_f, M).to(wdt)                           # dW = dy^T x
        if bdt is not None and ctx.needs_input_grad[2]:
            gb = colsum(g).to(bdt)
        return gx, gw, gb


def linear(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor] = None) -> torch.Tensor:
    """y = x W^T + b (PyG `Linear.forward`), W: [out, in]."""
    return _Linear.apply(x, weight, bias)


class Csr:
    """Both CSR views of one edge_index [2, E] (row 0 = source, row 1 = destination), built once per graph with device sorts:
    by destination for the forward aggregation, by source for its backward.  `scale`: 1 / max(in-degree, 1) per edge ('mean')."""

    def __init__(self, edge_index: torch.Tensor, n_src: int, n_dst: int, mean: bool):
        _need_gpu(edge_index, "GraphConv")
        src, dst = edge_index[0].long(), edge_index[1].long()
        if src.numel() and (int(src.max()) >= n_src or int(dst.max()) >= n_dst or int(src.min()) < 0 or int(dst.min()) < 0):
            raise IndexError("edge_index refers to nodes outside x")
        self.n_src, self.n_dst, self.mean = n_src, n_dst, mean
        deg = torch.bincount(dst, minlength=n_dst)
        o = torch.sort(dst, stable=True).indices
        self.f_rowptr = torch.cat([deg.new_zeros(1), deg.cumsum(0)]).to(torch.int32)
        self.f_col = src[o].to(torch.int32)
        e_scale = (1.0 / deg.clamp(min=1).to(torch.float32))[dst] if mean else None
        self.f_scale = e_scale[o].contiguous() if mean else None
        o2 = torch.sort(src, stable=True).indices
        sdeg = torch.bincount(src, minlength=n_src)
        self.b_rowptr = torch.cat([sdeg.new_zeros(1), sdeg.cumsum(0)]).to(torch.int32)
        self.b_col = dst[o2].to(torch.int32)
        self.b_scale = e_scale[o2].contiguous() if mean else None


def _aggregate(x: torch.Tensor, rowptr, col, scale, n_rows: int) -> torch.Tensor:
    lib = eng.load_library()
    H = x.shape[1]
    out = torch.empty(n_rows, H, dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        eng._check(lib, lib.mshgnn_op_aggregate(x.data_ptr(), H, rowptr.data_ptr(), col.data_ptr(), scale.data_ptr() if scale is not None else None,
                                                out.data_ptr(), H, n_rows, H, _stream(x.device)), "mshgnn_op_aggregate")
    return out


class _GraphConv(torch.autograd.Function):
    """out = (aggr_{j->i} x_src[j]) W_rel^T + b_rel + x_dst W_root^T   (PyG 2.5.0 GraphConv.forward: aggregate first, then lin_rel)."""

    @staticmethod
    def forward(ctx, x_src, x_dst, w_rel, b_rel, w_root, csr):
        for t in (x_src, x_dst, w_rel, w_root):
            _need_gpu(t, "GraphConv")
        out_f, in_s = w_rel.shape
        in_d = w_root.shape[1]
        if x_src.shape[1] != in_s or x_dst.shape[1] != in_d:
            raise RuntimeError(f"GraphConv: feature widths {x_src.shape[1]} / {x_dst.shape[1]} do not match the weights ({in_s} / {in_d})")
        xs, xd, wr, wo = _f32(x_src), _f32(x_dst), _f32(w_rel), _f32(w_root)
        agg = _aggregate(xs, csr.f_rowptr, csr.f_col, csr.f_scale, csr.n_dst)
        y = gemm(agg, in_s, 1, wr, in_s, 1, csr.n_dst, out_f, in_s, bias=_f32(b_rel) if b_rel is not None else None)
        gemm(xd, in_d, 1, wo, in_d, 1, csr.n_dst, out_f, in_d, out=y, accumulate=True)
        ctx.save_for_backward(agg, xd, wr, wo)
        ctx.csr = csr
        ctx.meta = (x_src.dtype, x_dst.dtype, w_rel.dtype, b_rel.dtype if b_rel is not None else None, w_root.dtype)
        return y.to(x_dst.dtype)

    @staticmethod
    def backward(ctx, gy):
        agg, xd, wr, wo = ctx.saved_tensors
        csr = ctx.csr
        dts, dtd, dtwr, dtb, dtwo = ctx.meta
        out_f, in_s = wr.shape
        in_d = wo.shape[1]
        g = _f32(gy)
        M = g.shape[0]
        gxs = gxd = gwr = gb = gwo = None
        if ctx.needs_input_grad[0]:
            gagg = gemm(g, out_f, 1, wr, 1, in_s, M, in_s, out_f)                                  # d agg = dy W_rel
            gxs = _aggregate(gagg, csr.b_rowptr, csr.b_col, csr.b_scale, csr.n_src).to(dts)        # transposed aggregation
        if ctx.needs_input_grad[1]:
            gxd = gemm(g, out_f, 1, wo, 1, in_d, M, in_d, out_f).to(dtd)
        if ctx.needs_input_grad[2]:
            gwr = gemm(g, 1, out_f, agg, 1, in_s, out_f, in_s, M).to(dtwr)
        if dtb is not None and ctx.needs_input_grad[3]:
            gb = colsum(g).to(dtb)
        if ctx.needs_input_grad[4]:
            gwo = gemm(g, 1, out_f, xd, 1, in_d, out_f, in_d, M).to(dtwo)
        return gxs, gxd, gwr, gb, gwo, None


_csr_cache: dict = {}


def csr_of(edge_index: torch.Tensor, n_src: int, n_dst: int, mean: bool) -> Csr:
    """CSR views of an edge_index, cached on the tensor's storage address, version and shape (a training loop presents the same graph every
    step).  The cache entry keeps a reference to the tensor, so its storage cannot be freed and handed to ANOTHER edge_index while the entry is
    alive -- an address match therefore means the same data (in-place edits bump the version)."""
    key = (edge_index.data_ptr(), edge_index._version, tuple(edge_index.shape), tuple(edge_index.stride()), n_src, n_dst, mean, str(edge_index.device))
    hit = _csr_cache.get(key)
    if hit is None:
        if len(_csr_cache) >= 64:
            _csr_cache.clear()
        hit = _csr_cache[key] = (edge_index, Csr(edge_index, n_src, n_dst, mean))
    return hit[1]


def graph_conv(x: "torch.Tensor | Tuple[torch.Tensor, torch.Tensor]", edge_index: torch.Tensor, w_rel: torch.Tensor,
               b_rel: Optional[torch.Tensor], w_root: torch.Tensor, aggr: str = "add") -> torch.Tensor:
    """PyG `GraphConv.forward(x | (x_src, x_dst), edge_index)`; returns [n_dst, out]."""
    x_src, x_dst = (x, x) if isinstance(x, torch.Tensor) else x
    if x_dst is None:
        raise ValueError("GraphConv needs destination features (x_dst) on this engine")
    if edge_index.dim() != 2 or edge_index.shape[0] != 2:
        raise ValueError("edge_index must be [2, num_edges]")
    csr = csr_of(edge_index, x_src.shape[0], x_dst.shape[0], aggr == "mean")
    return _GraphConv.apply(x_src, x_dst, w_rel, b_rel, w_root, csr)
