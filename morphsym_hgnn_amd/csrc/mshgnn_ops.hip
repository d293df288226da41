// Stand-alone operators behind the four torch_geometric.nn names of the hot path (hgnn_c2.py:3,88,100-113,131; PyG 2.5.0 semantics):
//   Linear / HeteroDictLinear          y = x W^T + b                                    -> mshgnn_op_gemm
//   GraphConv                          out = lin_rel(aggr_{j->i} x_src[j]) + lin_root(x_dst)   -> mshgnn_op_aggregate + mshgnn_op_gemm (accumulate)
//   their autograd backward            dX = dY W, dW = dY^T X (split-K, fixed-order sums), db = column sums of dY, transposed aggregation
// for arbitrary graphs / widths (the fused engine of mshgnn.hip needs the fixed per-robot topology; these do not).  fp32 operands and fp32
// MFMA (v_mfma_f32_32x32x2_f32: an exact fp32 fma chain), so a maintainer who swaps only the PyG import keeps the reference's numerics to
// fp32 rounding.  Deterministic: no float atomics; split-K partials and column sums are added in a fixed order.
#include "mshgnn_device.hpp"

namespace {

constexpr int OG_TM = 64, OG_TN = 64, OG_TK = 16, OG_LD = OG_TM + 1;      // workgroup tile 64 x 64, K tile 16; LDS rows padded by one float

struct OpGemmArgs {
    const float* A; int64_t sAm, sAk;      // A(m, k) = A[m sAm + k sAk]
    const float* B; int64_t sBn, sBk;      // B(n, k) = B[n sBn + k sBk]
    const float* bias;                     // [N] or null
    float* C; int64_t ldc;                 // C[m ldc + n]   (splits > 1: partial z goes to ws[z][M][N])
    float* ws;
    int M, N, K, kchunk, splits, accumulate;
};

// C[m][n] (+)= sum_k A(m, k) B(n, k) (+ bias[n]); one workgroup = one 64 x 64 tile of C over one K chunk, 4 waves x (32 x 32) MFMA tiles
__global__ __launch_bounds__(256) void k_op_gemm(OpGemmArgs a) {
    __shared__ float As[2][OG_TK * OG_LD], Bs[2][OG_TK * OG_LD];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, wr = wv >> 1, wc = wv & 1;
    const int m0 = blockIdx.y * OG_TM, n0 = blockIdx.x * OG_TN, z = blockIdx.z;
    const int k_begin = z * a.kchunk, k_end = min(a.K, k_begin + a.kchunk);
    // staging map: along whichever index is contiguous in memory (k: 16 consecutive threads read 64 bytes of a row; m / n: 64 consecutive rows)
    const bool a_kc = a.sAk == 1 || a.sAm != 1, b_kc = a.sBk == 1 || a.sBn != 1;
    f32x16 acc;
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[q] = 0.f;
    float ra[4], rb[4];
    auto fetch = [&](int k0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int kk = a_kc ? (tid & 15) : (tid >> 6) + 4 * i, mm = a_kc ? (tid >> 4) + 16 * i : (tid & 63);
            const int m = m0 + mm, k = k0 + kk;
            ra[i] = (m < a.M && k < k_end) ? a.A[(int64_t)m * a.sAm + (int64_t)k * a.sAk] : 0.f;
            const int kb = b_kc ? (tid & 15) : (tid >> 6) + 4 * i, nn = b_kc ? (tid >> 4) + 16 * i : (tid & 63);
            const int n = n0 + nn, k2 = k0 + kb;
            rb[i] = (n < a.N && k2 < k_end) ? a.B[(int64_t)n * a.sBn + (int64_t)k2 * a.sBk] : 0.f;
        }
    };
    auto stage = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int kk = a_kc ? (tid & 15) : (tid >> 6) + 4 * i, mm = a_kc ? (tid >> 4) + 16 * i : (tid & 63);
            As[buf][kk * OG_LD + mm] = ra[i];
            const int kb = b_kc ? (tid & 15) : (tid >> 6) + 4 * i, nn = b_kc ? (tid >> 4) + 16 * i : (tid & 63);
            Bs[buf][kb * OG_LD + nn] = rb[i];
        }
    };
    const int ntile = (k_end - k_begin + OG_TK - 1) / OG_TK;
    if (ntile > 0) { fetch(k_begin); stage(0); }
    __syncthreads();
    for (int t = 0; t < ntile; ++t) {
        const int buf = t & 1;
        if (t + 1 < ntile) fetch(k_begin + (t + 1) * OG_TK);      // the next K tile streams in under this tile's MFMAs
#pragma unroll
        for (int ks = 0; ks < OG_TK; ks += 2) {
            const float av = As[buf][(ks + (lane >> 5)) * OG_LD + wr * 32 + (lane & 31)];
            const float bv = Bs[buf][(ks + (lane >> 5)) * OG_LD + wc * 32 + (lane & 31)];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc, 0, 0, 0);
        }
        if (t + 1 < ntile) stage(buf ^ 1);      // (the other buffer: its readers finished before the previous barrier)
        __syncthreads();
    }
    float* out = a.splits > 1 ? a.ws + (size_t)z * a.M * a.N : a.C;
    const int64_t ld = a.splits > 1 ? a.N : a.ldc;
    const int n = n0 + wc * 32 + (lane & 31);
    if (n < a.N) {
        const float bn = (a.bias && a.splits == 1) ? a.bias[n] : 0.f;
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int m = m0 + wr * 32 + (q & 3) + ((q >> 2) << 3) + ((lane >> 5) << 2);
            if (m < a.M) {
                float v = acc[q] + bn;
                if (a.accumulate && a.splits == 1) v += out[(int64_t)m * ld + n];
                out[(int64_t)m * ld + n] = v;
            }
        }
    }
}

// C[m][n] (+)= sum_z ws[z][m][n] (+ bias[n]), z in ascending order
__global__ __launch_bounds__(256) void k_op_splitk_sum(const float* ws, const float* bias, float* C, int64_t ldc, int M, int N, int splits, int accumulate) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x, total = (int64_t)M * N;
    if (i >= total) return;
    const int m = (int)(i / N), n = (int)(i % N);
    float s = 0.f;
    for (int z = 0; z < splits; ++z) s += ws[(size_t)z * total + i];
    if (bias) s += bias[n];
    if (accumulate) s += C[(int64_t)m * ldc + n];
    C[(int64_t)m * ldc + n] = s;
}

// out[r][:] = sum_{e in [rowptr[r], rowptr[r+1])} scale[e] x[col[e]][:]   (scale null: 1); 32 lanes per row, 8 rows per workgroup, edges in CSR order
__global__ __launch_bounds__(256) void k_op_aggregate(const float* x, int64_t ldx, const int32_t* rowptr, const int32_t* col, const float* scale,
                                                      float* out, int64_t ldo, int n_rows, int H) {
    const int r = blockIdx.x * 8 + (threadIdx.x >> 5), l = threadIdx.x & 31;
    if (r >= n_rows) return;
    const int e0 = rowptr[r], e1 = rowptr[r + 1];
    for (int c = l; c < H; c += 32) {
        float s = 0.f;
        for (int e = e0; e < e1; ++e) {
            const float v = x[(int64_t)col[e] * ldx + c];
            s += scale ? scale[e] * v : v;
        }
        out[(int64_t)r * ldo + c] = s;
    }
}

// column sums of X[M][N] in two fixed-order stages: partial[b][n] over rows [b RB, (b + 1) RB), then over b
constexpr int OC_RB = 512;
__global__ __launch_bounds__(256) void k_op_colsum1(const float* X, int64_t ldx, float* partial, int M, int N) {
    const int r0 = blockIdx.x * OC_RB, r1 = min(M, r0 + OC_RB);
    for (int n = threadIdx.x; n < N; n += 256) {
        float s = 0.f;
        for (int r = r0; r < r1; ++r) s += X[(int64_t)r * ldx + n];
        partial[(size_t)blockIdx.x * N + n] = s;
    }
}
__global__ __launch_bounds__(256) void k_op_colsum2(const float* partial, float* out, int nb, int N) {
    const int n = blockIdx.x * 256 + threadIdx.x;
    if (n >= N) return;
    float s = 0.f;
    for (int b = 0; b < nb; ++b) s += partial[(size_t)b * N + n];
    out[n] = s;
}

}  // namespace

extern "C" int64_t mshgnn_op_gemm_workspace(int64_t M, int64_t N, int64_t K, int32_t* splits_out) {
    // split K where the output alone cannot fill the chip (weight gradients: a 128 x 128 output reduced over ~1e5 rows)
    const int64_t tiles = ((M + OG_TM - 1) / OG_TM) * ((N + OG_TN - 1) / OG_TN);
    int64_t splits = 1;
    if (tiles < 512 && K >= 2048) splits = std::min<int64_t>(std::min<int64_t>((1024 + tiles - 1) / tiles, K / 512), 256);
    if (splits < 1) splits = 1;
    if (splits_out) *splits_out = (int32_t)splits;
    return splits > 1 ? splits * M * N * (int64_t)sizeof(float) : 0;
}

extern "C" int mshgnn_op_gemm(const float* A, int64_t sAm, int64_t sAk, const float* B, int64_t sBn, int64_t sBk, const float* bias,
                              float* C, int64_t ldc, int64_t M, int64_t N, int64_t K, int accumulate, void* workspace, void* stream) {
    if (M < 0 || N < 0 || K < 0 || M > INT32_MAX || N > INT32_MAX || K > INT32_MAX) return set_err(MSHGNN_EINVAL, "mshgnn_op_gemm: bad shape");
    if (M == 0 || N == 0) return MSHGNN_OK;
    if (!A || !B || !C) return set_err(MSHGNN_EINVAL, "mshgnn_op_gemm: null operand");
    if (ldc < N) return set_err(MSHGNN_EINVAL, "mshgnn_op_gemm: ldc smaller than N");
    int32_t splits = 1;
    const int64_t need = mshgnn_op_gemm_workspace(M, N, K, &splits);
    if (need && !workspace) return set_err(MSHGNN_EINVAL, "mshgnn_op_gemm: this shape needs a split-K workspace (mshgnn_op_gemm_workspace)");
    OpGemmArgs a{A, sAm, sAk, B, sBn, sBk, bias, C, ldc, reinterpret_cast<float*>(workspace), (int)M, (int)N, (int)K, 0, splits, accumulate};
    a.kchunk = (int)((((K + splits - 1) / splits) + OG_TK - 1) / OG_TK * OG_TK);
    if (a.kchunk < OG_TK) a.kchunk = OG_TK;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const dim3 grid((unsigned)((N + OG_TN - 1) / OG_TN), (unsigned)((M + OG_TM - 1) / OG_TM), (unsigned)splits);
    if (grid.y > 65535) return set_err(MSHGNN_EINVAL, "mshgnn_op_gemm: more than 65535 row tiles (M > 4 194 240)");
    hipLaunchKernelGGL(k_op_gemm, grid, dim3(256), 0, st, a);
    if (splits > 1) {
        const int64_t total = M * N;
        hipLaunchKernelGGL(k_op_splitk_sum, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, a.ws, bias, C, ldc, (int)M, (int)N, splits, accumulate);
    }
    HIPCHK(hipGetLastError());
    return MSHGNN_OK;
}

extern "C" int mshgnn_op_aggregate(const float* x, int64_t ldx, const int32_t* rowptr, const int32_t* col, const float* edge_scale,
                                   float* out, int64_t ldo, int64_t n_rows, int64_t width, void* stream) {
    if (n_rows < 0 || width < 0 || n_rows > INT32_MAX || width > INT32_MAX) return set_err(MSHGNN_EINVAL, "mshgnn_op_aggregate: bad shape");
    if (n_rows == 0 || width == 0) return MSHGNN_OK;
    if (!rowptr || !out) return set_err(MSHGNN_EINVAL, "mshgnn_op_aggregate: null operand");      // (x / col may be null for a graph without edges: never read)
    hipLaunchKernelGGL(k_op_aggregate, dim3((unsigned)((n_rows + 7) / 8)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), x, ldx, rowptr, col, edge_scale,
                       out, ldo, (int)n_rows, (int)width);
    HIPCHK(hipGetLastError());
    return MSHGNN_OK;
}

extern "C" int64_t mshgnn_op_colsum_workspace(int64_t M, int64_t N) { return ((M + OC_RB - 1) / OC_RB) * N * (int64_t)sizeof(float); }

extern "C" int mshgnn_op_colsum(const float* X, int64_t ldx, float* out, int64_t M, int64_t N, void* workspace, void* stream) {
    if (M < 0 || N < 0 || M > INT32_MAX || N > INT32_MAX) return set_err(MSHGNN_EINVAL, "mshgnn_op_colsum: bad shape");
    if (N == 0) return MSHGNN_OK;
    if (!out || (M > 0 && (!X || !workspace))) return set_err(MSHGNN_EINVAL, "mshgnn_op_colsum: null operand");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int nb = (int)((M + OC_RB - 1) / OC_RB);
    if (nb > 0) hipLaunchKernelGGL(k_op_colsum1, dim3(nb), dim3(256), 0, st, X, ldx, reinterpret_cast<float*>(workspace), (int)M, (int)N);
    hipLaunchKernelGGL(k_op_colsum2, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, st, reinterpret_cast<const float*>(workspace), out, nb, (int)N);
    HIPCHK(hipGetLastError());
    return MSHGNN_OK;
}
