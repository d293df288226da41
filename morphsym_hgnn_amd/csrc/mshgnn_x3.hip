// Split-bf16 parity plan (MSHGNN_BF16X3) of the MS-HGNN engine: the same path as the bf16 plan of mshgnn.hip
// (GRF_HGNN_C2.forward hgnn_c2.py:133-182 and its autograd backward, lowered as DESIGN.md section 2 describes) at the
// north_star's tolerance (1e-4 relative) on the bf16 matrix cores.
//
// Every fp32 quantity x that feeds a product -- inputs, weights, activations, activation gradients -- travels as two bf16
// values, hi = bf16(x) and lo = bf16(x - hi): x = hi + lo to 16 mantissa bits.  A product is taken as
//     x w  =  x_hi w_hi + x_lo w_hi + x_hi w_lo          (the lo x lo term is 2^-18 relative: below fp32 resolution)
// with fp32 accumulation, i.e. three bf16 MFMAs per term instead of one fp32 MFMA at 1/16 of the rate.  Layout:
//   * inputs x[t]: fp32 in HBM (as the MSHGNN_F32 plan takes them); split in registers while they are staged;
//   * every activation tensor in the workspace: rows of 256 bf16, [hi 128 | lo 128] ([NN][B][2][128]): a window's hi and lo halves are
//     one contiguous 512-byte piece for every stream (stash writes, tile loads, the weight-gradient kernel's operand rows);
//   * LDS tile of the stack kernels: block n = hi plane of node n, block lo_blk + n = its lo plane (A1-C2: 40 blocks = 160 KB,
//     one 8-wave workgroup per CU); k_prep writes a hi and a lo image of every weight pack;
//   * the MAC loop of the stack kernels is the bf16 plan's: the plan compiler (split_segs, mshgnn_plan.hpp) turns each
//     segment into two -- the hi image with the hi and lo block of every source, the lo image with the hi block.
#include "mshgnn_device.hpp"

using T16 = __bf16;
using P16 = Prec<__bf16>;

// element index of the hi half of (window w, node) in a split-plan activation tensor; the lo half follows at + H
__device__ __forceinline__ size_t x3_idx(int w, int node, int B) { return ((size_t)node * B + w) * (2 * H); }

// ------------------------------------------------------------------------------------------------------
// k_prep_x3: hi and lo MFMA B-fragment images of every weight pack (root-sum, transpose) + bias sums
// ------------------------------------------------------------------------------------------------------
// (many packs: k_prep_tiled<__bf16, true>, mshgnn_device.hpp)
// output vector `idx` of this launch's pack range (hi and lo images), or (idx past the packs) one bias sum
__device__ __forceinline__ void prep_one_x3(const PrepArgs& a, int idx, bool with_bias) {
    constexpr int EPC = 8, NBV = 8;
    const int vec_per_pack = H * H / EPC;
    const int npk = a.pack_n < 0 ? a.n_packs : a.pack_n;
    const int total = npk * vec_per_pack;
    if (idx < total) {
        const int pack = a.pack0 + idx / vec_per_pack, r = idx % vec_per_pack;
        const int gid = pack * vec_per_pack + r;
        const int lane = r % 64, v = (r / 64) % NBV, wv = r / (64 * NBV);
        const PackDesc pd = a.packs[pack];
        float g[8][EPC];     // up to 8 source matrices (root-sum), every gather issued before the first add
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int x = 0; x < EPC; ++x) {
                // same element map as k_prep<__bf16>: MFMA row i = lane & 15 of the wave's 16-row block fb carries output feature 8 (i / 4) + 4 fb + i % 4
                const int i16 = lane & 15;
                const int k = 32 * (lane >> 4) + 8 * (v & 3) + x, col = wv * 32 + 8 * (i16 >> 2) + 4 * (v >> 2) + (i16 & 3);
                g[i][x] = 0.f;
                if (i < pd.n_src) {
                    if (pd.orient == 0) { if (k < pd.ncols) g[i][x] = a.params[pd.src[i] + (int64_t)col * pd.ld + pd.col0 + k]; }
                    else g[i][x] = a.params[pd.src[i] + (int64_t)k * pd.ld + col];
                }
            }
        float sum[EPC];
#pragma unroll
        for (int x = 0; x < EPC; ++x) {
            sum[x] = 0.f;
#pragma unroll
            for (int i = 0; i < 8; ++i) sum[x] += g[i][x];     // fixed order: same value every step
        }
        u32x4 hi, lo;
        split_oct(f32x4{sum[0], sum[1], sum[2], sum[3]}, f32x4{sum[4], sum[5], sum[6], sum[7]}, hi, lo);
        u32x4* dst = reinterpret_cast<u32x4*>(a.wpack);
        dst[gid] = hi;
        dst[(size_t)a.n_packs * vec_per_pack + gid] = lo;
    } else if (with_bias) {
        const int b = idx - total;
        if (b < a.n_biases * H) {
            const BiasDesc bd = a.biases[b / H];
            float s = 0.f;
            for (int i = 0; i < bd.n_src; ++i) s += a.params[bd.src[i] + (b % H)];
            a.bias[b] = s;
        }
    }
}
#ifndef MSHGNN_SPEC_SHARD
#define MSHGNN_SPEC_SHARD 0      // 1..7: this source compiled as one of the translation units that instantiate the compile-time programs' kernels (below, csrc/Makefile)
#endif
#if MSHGNN_SPEC_SHARD == 0
__global__ void k_prep_x3(PrepArgs a) { prep_one_x3(a, blockIdx.x * blockDim.x + threadIdx.x, true); }
#endif

// ------------------------------------------------------------------------------------------------------
// k_enc_x3: X_0[node] = relu((mask . x) W_enc^T + b) from fp32 inputs (hgnn_c2.py:143-147); one workgroup = 64 windows of ONE
// node, K streamed in chunks of 128 through LDS (hi blocks [0, 4), lo blocks [4, 8)), the next chunk prefetched in registers
// ------------------------------------------------------------------------------------------------------
// out[e] = e < n0 ? a[e] : b[e - n0] over the 8 fp32 elements (a0 | a1), (b0 | b1): the second piece of a chunk that straddles two runs of a
// window row (k_enc_x3<.., SERIES>); b is moved up by n0 elements in three conditional stages (4, 2, 1), n0 in [1, 7] is per thread
__device__ __forceinline__ void splice8f(u32x4& a0, u32x4& a1, const u32x4 b0, const u32x4 b1, int n0) {
    unsigned sft[8] = {b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2], b1[3]};
    if (n0 & 4) {
#pragma unroll
        for (int e = 7; e >= 4; --e) sft[e] = sft[e - 4];
    }
    if (n0 & 2) {
#pragma unroll
        for (int e = 7; e >= 2; --e) sft[e] = sft[e - 2];
    }
    if (n0 & 1) {
#pragma unroll
        for (int e = 7; e >= 1; --e) sft[e] = sft[e - 1];
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        a0[e] = e < n0 ? a0[e] : sft[e];
        a1[e] = e + 4 < n0 ? a1[e] : sft[e + 4];
    }
}

// SERIES (with ALIGNED): the fp32 inputs are gathered from the sequence's resident series like the bf16 encoder's (k_enc_fwd<.., SERIES>): element k
// of a node row = element starts[w] + k % T of run k / T -- two 4-byte-aligned 16-byte loads per (window, chunk), four and a splice where the chunk
// straddles two runs -- and the materialised window rows are written on the side for the weight-gradient pass (a.x: the window buffers).
// SRC (8 / 4, with ALIGNED): the rows come from the caller's own fp64 / fp32 tensors at their dense pitch (WideSrc, mshgnn_device.hpp) and are also written to
// a.x as fp32 rows at the engine's pitch for the weight-gradient kernel (mshgnn_*_src entry points)
template <bool ALIGNED, bool SERIES = false, int SRC = 0> __global__ __launch_bounds__(256) void k_enc_x3(EncArgs a, int n_img, SeriesSrc ser, WideSrc wsrc) {
    static_assert(!SERIES || ALIGNED, "the series gather writes aligned window buffers");
    static_assert(SRC == 0 || (ALIGNED && !SERIES), "wide source rows: aligned destination rows, no series gather");
    using P = P16;
    constexpr int MB = 4;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lab_blocks = SERIES ? (int)((ser.lab.B + 255) / 256) : 0;      // SERIES: the first workgroups compute the batch's window labels (as k_enc_fwd)
    if constexpr (SERIES) {
        if ((int)blockIdx.x < lab_blocks) {
            const int64_t b = (int64_t)blockIdx.x * 256 + tid;
            if (b < ser.lab.B) window_labels_one(ser.lab, b);
            return;
        }
    }
    const int bid = (int)blockIdx.x - lab_blocks;
    if (bid >= a.wg_prefix[a.n_types]) {      // a workgroup of the embedded layer-pack prep (EncArgs.prep)
        if constexpr (ALIGNED && !SERIES) prep_one_x3(a.prep, (bid - a.wg_prefix[a.n_types]) * 256 + tid, false);
        return;
    }
    int t = 0;
    while (t + 1 < a.n_types && bid >= a.wg_prefix[t + 1]) ++t;
    const int local = bid - a.wg_prefix[t];
    const int node = a.node_list[a.node_off[t] + (ENC_ORDER ? local % a.nodes[t] : local / a.tiles)], tile = ENC_ORDER ? local / a.nodes[t] : local % a.tiles;
    const bool skip = SERIES && ((a.skip_mask >> (a.tbase[t] + node)) & 1ull) != 0;      // window rows only: nobody reads this node's X_0 (uniform; EncArgs)
    const int w0 = tile * MB * P::ROWS;
    const float* x = reinterpret_cast<const float*>(a.x[t]);
    const int64_t pitch = a.pitch[t];
    const int F = a.width[t], nt = a.tbase[t + 1] - a.tbase[t], nkc = a.nkc[t], vb = a.vb[t];      // nt: nodes of the type in the input rows
    const uint8_t* sg = a.signs + a.sign_off[t] + (size_t)node * nkc * H;
    const T16* wpack = reinterpret_cast<const T16*>(a.wpack);
    const float* bias = a.bias + (size_t)max(a.bias_idx[t], 0) * H;

    P::Acc acc[MB];
#pragma unroll
    for (int m = 0; m < MB; ++m) acc_init_bias<T16>(acc[m], bias, wv, lane);
    const int c = tid & 15, r0 = tid >> 4;      // staging: thread = (row, 8-element chunk) of each of the MB row blocks
    P::BFrag bfh, bfl;
    P::AFrag af;
    const AOff<T16> ao(lane);
    u32x4 v[MB][2];                             // the 8 fp32 elements of the chunk (two 16-byte loads)
    u32x2 wv8[SRC ? MB : 1][SRC ? SRC : 1];     // SRC: the chunk's 8 source elements as 8-byte units, untouched until the staging pass
    const bool unit_ok = SRC == 8 || (F & 1) == 0;
    const int64_t spitch = SRC ? wsrc.pitch[t] : 0;
    int srow[SERIES ? MB : 1]; int rfirst = 0;  // SERIES: first series row of this thread's window rows, the node row's first run
    if constexpr (SERIES) {
#pragma unroll
        for (int m = 0; m < MB; ++m) srow[m] = (int)ser.starts[min(w0 + m * P::ROWS + r0, a.B - 1)];
        rfirst = ser.rows[2 * (ser.row0[t] + node)];
    }
    auto fetch = [&](int kc) {
        const int k0 = kc * H + c * 8;
        const int nv = F - k0;
        if constexpr (SERIES) {
            // elements [k0, k0 + 8) of the row: n0 of them from run j at time offset off, the rest from run j + 1 at offset 0
            const int j = k0 / ser.T, off = k0 - j * ser.T, n0 = min(8, ser.T - off);
            const bool second = min(nv, 8) > n0;
            const unsigned long long pa = nv > 0 ? ser.run_ptr[rfirst + j] : 0ull, pb = second ? ser.run_ptr[rfirst + j + 1] : 0ull;
            const u32x4 ones = u32x4{0x3f800000u, 0x3f800000u, 0x3f800000u, 0x3f800000u};      // the constant-1 run
#pragma unroll
            for (int m = 0; m < MB; ++m) {
                u32x4 a0 = nv > 0 ? ones : u32x4{0, 0, 0, 0}, a1 = a0;
                if (pa) {      // (4-byte aligned; the second load may run up to 7 elements past the window's last step: the columns' slack)
                    const float* sp = reinterpret_cast<const float*>(pa) + srow[m] + off;
                    a0 = *reinterpret_cast<const u32x4*>(sp); a1 = *reinterpret_cast<const u32x4*>(sp + 4);
                }
                if (second) {
                    u32x4 b0 = ones, b1 = ones;
                    if (pb) {
                        const float* sp = reinterpret_cast<const float*>(pb) + srow[m];
                        b0 = *reinterpret_cast<const u32x4*>(sp); b1 = *reinterpret_cast<const u32x4*>(sp + 4);
                    }
                    splice8f(a0, a1, b0, b1, n0);
                }
                v[m][0] = a0; v[m][1] = a1;
            }
            return;
        }
        if constexpr (SRC > 0) {
#pragma unroll
            for (int m = 0; m < MB; ++m) {
                const int w = w0 + m * P::ROWS + r0;
                const char* row = reinterpret_cast<const char*>(wsrc.p[t]) + ((size_t)min(w, a.B - 1) * nt + node) * spitch * SRC;
                wide_fetch<SRC>(wv8[m], row, k0, F, unit_ok, w < a.B);
            }
            return;
        }
#pragma unroll
        for (int m = 0; m < MB; ++m) {
            const int w = w0 + m * P::ROWS + r0;
            if constexpr (ALIGNED) {
                // unconditional raw loads (nothing uses them here): rows past the batch re-read the last row, halves past the row's end
                // re-read the K chunk's first elements -- the staging pass zeroes the latter, the former are never stored
                const float* src = x + ((size_t)min(w, a.B - 1) * nt + node) * pitch;
                v[m][0] = ld16<ENC_NT>(src + (nv > 0 ? k0 : kc * H));
                v[m][1] = ld16<ENC_NT>(src + (nv > 4 ? k0 + 4 : kc * H));
            } else {
                v[m][0] = u32x4{0, 0, 0, 0}; v[m][1] = u32x4{0, 0, 0, 0};
                if (w < a.B) {
                    const float* src = x + ((size_t)w * nt + node) * pitch + k0;
                    v[m][0] = load_chunk<float>(src, nv, vb);
                    v[m][1] = load_chunk<float>(src + 4, nv - 4, vb);
                }
            }
        }
    };
    fetch(0);
    for (int kc = 0; kc < nkc; ++kc) {
        const u32x4 sxa = sign_xor<float>(sg + kc * H + c * 8), sxb = sign_xor<float>(sg + kc * H + c * 8 + 4);   // apply_symmetry: +-1 mask as a sign-bit XOR
        const int nv = F - (kc * H + c * 8);
        __syncthreads();   // previous chunk's MFMAs are done reading LDS
#pragma unroll
        for (int m = 0; m < MB; ++m) {
            u32x4 fa = v[m][0], fb = v[m][1];
            if constexpr (SRC > 0) {      // fp64 -> fp32 (round to nearest even, as torch's .float()), or the fp32 units as they are; elements past the row: zero
                f32x4 lo4, hi4;
                wide_to_f32<SRC>(wv8[m], nv, lo4, hi4);
                fa = __builtin_bit_cast(u32x4, lo4); fb = __builtin_bit_cast(u32x4, hi4);
            } else
            if (kc + 1 == nkc) { fa = chunk_keep_first<float>(fa, nv); fb = chunk_keep_first<float>(fb, nv - 4); }     // only the last K chunk has pad columns
            if constexpr (SERIES || SRC > 0) {      // the materialised window row (raw values: the sign mask is applied by whoever reads it)
                const int w = w0 + m * P::ROWS + r0, k0 = kc * H + c * 8;
                if (x != nullptr && w < a.B) {
                    float* dst = const_cast<float*>(x) + ((size_t)w * nt + node) * pitch + k0;
                    if (k0 < (int)pitch) *reinterpret_cast<u32x4*>(dst) = fa;
                    if (k0 + 4 < (int)pitch) *reinterpret_cast<u32x4*>(dst + 4) = fb;
                }
            }
            fa ^= sxa; fb ^= sxb;
            u32x4 hi, lo;
            split_oct(__builtin_bit_cast(f32x4, fa), __builtin_bit_cast(f32x4, fb), hi, lo);
            *reinterpret_cast<u32x4*>(smem + lds_chunk<T16>(m, r0, c)) = hi;
            *reinterpret_cast<u32x4*>(smem + lds_chunk<T16>(MB + m, r0, c)) = lo;
        }
        __syncthreads();
        if (!skip) {
            load_bfrag<T16>(bfh, wpack, a.pack0[t] + kc, wv, lane);              // before the prefetch: vmcnt retires in order
            load_bfrag<T16>(bfl, wpack, n_img + a.pack0[t] + kc, wv, lane);
        }
        if (kc + 1 < nkc) fetch(kc + 1);   // the next K chunk streams from HBM under this chunk's MFMAs
        if (!skip) {
#pragma unroll
            for (int m = 0; m < MB; ++m) {
                if (w0 + m * P::ROWS < a.B) {   // uniform
                    load_afrag<T16>(af, smem, m, ao);
                    mac(acc[m], af, bfh);
                    mac(acc[m], af, bfl);
                    load_afrag<T16>(af, smem, MB + m, ao);
                    mac(acc[m], af, bfh);
                }
            }
        }
    }
    if (skip) return;
    T16* x0 = reinterpret_cast<T16*>(a.x0);
    const int gnode = a.tbase[t] + node;
#pragma unroll
    for (int m = 0; m < MB; ++m) {
        const int w = w0 + m * P::ROWS + c_win(lane);
        if (w0 + m * P::ROWS < a.B) {     // uniform: the 16-window block exists (rows past the batch land in the mask buffer's padding)
            const unsigned bits = relu_with_bits<T16>(acc[m]);
            if (a.mask0) a.mask0[relu_tile_base(gnode, a.B, (w0 + m * P::ROWS) >> 4, wv) + lane] = (uint8_t)bits;
        }
        if (w < a.B) {
            u32x4 hi, lo;
            split_oct(acc[m].c[0], acc[m].c[1], hi, lo);
            T16* q = x0 + x3_idx(w, gnode, a.B) + wv * 32 + c_oct(lane);
            *reinterpret_cast<u32x4*>(q) = hi;
            *reinterpret_cast<u32x4*>(q + H) = lo;
        }
    }
}

// stage the [hi | lo] rows of every node for which keep(n) into LDS: 512 threads = 16 rows x 32 chunks, one node per load, 10 in flight
// (an 18-node tile in two round trips to HBM: with one workgroup per CU nothing else hides them)
template <typename Keep>
__device__ __forceinline__ void stage_tile_x3(char* smem, const T16* src, int NN, int LO, int w0, int B, int tid, Keep keep) {
    const int row = tid >> 5, c = tid & 31;      // chunk c < 16: hi half, else lo half
    const int blk_off = c < 16 ? 0 : LO, cc = c & 15;
    constexpr int BATCH = 10;
    for (int nb = 0; nb < NN; nb += BATCH) {
        u32x4 v[BATCH];
#pragma unroll
        for (int i = 0; i < BATCH; ++i) {
            v[i] = u32x4{0, 0, 0, 0};
            if (nb + i < NN && keep(nb + i) && w0 + row < B) v[i] = *reinterpret_cast<const u32x4*>(src + x3_idx(w0 + row, nb + i, B) + c * 8);
        }
#pragma unroll
        for (int i = 0; i < BATCH; ++i)
            if (nb + i < NN && keep(nb + i)) *reinterpret_cast<u32x4*>(smem + lds_chunk<T16>(blk_off + nb + i, row, cc)) = v[i];
    }
}

// three-product block GEMM of the base_transform chain on this wave's <= 2 accumulators: acc[u] += LDS[blk] (hi + lo) . W (hi + lo)
template <int N> __device__ __forceinline__ void mlp_mac3(P16::Acc (&acc)[N], const char* smem, int blk0, int lo_blk, int nmlp, int wh, const P16::BFrag& wh_, const P16::BFrag& wl_, int lane) {
    P16::AFrag af;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int n = 2 * u + wh;
        if (n < nmlp) {
            load_afrag<T16>(af, smem, blk0 + n, lane);
            mac(acc[u], af, wh_);
            mac(acc[u], af, wl_);
            load_afrag<T16>(af, smem, lo_blk + blk0 + n, lane);
            mac(acc[u], af, wh_);
        }
    }
}

// ------------------------------------------------------------------------------------------------------
// k_stack_fwd_x3: the whole message-passing stack (+ decoder, + wrapper MSE and decoder backward under mshgnn_step_mse) of one
// 16-window tile in one 8-wave workgroup -- k_stack_fwd of the bf16 plan on hi/lo planes
// ------------------------------------------------------------------------------------------------------
// ALIAS: the base_transform scratch blocks are the blocks of the last nodes (topologies whose doubled tile leaves no room: MiniCheetah-K4)
// STEP: part of k_stack_step_x3 -- the decoder tail leaves dX_L (both planes) in the out-type nodes' LDS blocks for the backward sweep of the same launch
// one forward layer of the split plan's 8-wave workgroup.  FH / FP: the layer's header and this wave half's program, interpreted (FHdr / FProg) or compile-time
// (SHdr / SProg: specialised kernels); WHS: the wave half as a template constant (compile-time programs: the per-node header reads fold) or -1; mid(): what has
// to settle between the MACs and the stores
template <bool ALIAS, bool STEP, int WHS, class FH, class FP, class Mid>
__device__ __forceinline__ void x3_fwd_layer(const StackArgs& a, char* smem, const T16* wpack, int wn, int wh, int lane, int l, int L, const FH& fh, const FP& wp, Mid&& mid) {
    using T = T16; using P = P16;
    const int tid = threadIdx.x, w0 = blockIdx.x * P::ROWS, B = a.B, NN = a.NN, LO = a.lo_blk, SCR = a.scr0;
    const int whv = WHS >= 0 ? WHS : wh;
    const bool train = a.training != 0;
    P::Acc acc[FS_HS];
        const int nmlp = fh[FH_NMLP], flags = fh[FH_FLAGS];
#pragma unroll
        for (int u = 0; u < FS_HS; ++u) {
            const int n = 2 * u + whv;
            if (n < NN && fh[FH_KIND + n] != NK_DEAD) acc_init_bias<T>(acc[u], a.bias + (size_t)fh[FH_BIAS + n] * H, wn, lane);
            else acc_fill(acc[u], 0.f);
        }
        FS_STAMP(2 + 4 * l);
        fs_run<T>(wp, acc, smem, wpack, wn, lane);
        // lane constants of the epilogue rebuilt per layer from an opaque copy of the lane id (per-node addresses derived from them were hoisted out
        // of the layer loop and spilled; a scratch reload next to pending stores is a full vmcnt(0) drain)
        const int lq = opaque(lane);
        const int win = c_win(lq), w = w0 + win, col = wn * 32 + c_oct(lq);
        const bool w_ok = w < B;
        FS_STAMP(3 + 4 * l);
        __syncthreads();   // every wave is done reading X_l: the node blocks may be overwritten
        FS_STAMP(4 + 4 * l);

        u32x4 hph[2] = {}, hpl[2] = {}, tph[2] = {}, tpl[2] = {};
        u32x4 vrh[2] = {}, vrl[2] = {};      // ALIAS: residual octets of the victim nodes SCR + wh + 2 j this wave owns, read before the chain overwrites their blocks
        if constexpr (ALIAS) {
            if (nmlp > 0 && (flags & FF_RESIDUAL)) {
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int v = SCR + whv + 2 * j;
                    if (v < SCR + nmlp && fh[FH_KIND + v] != NK_DEAD) {
                        vrh[j] = *reinterpret_cast<const u32x4*>(smem + lds_chunk<T>(v, win, col / P::EPC));
                        vrl[j] = *reinterpret_cast<const u32x4*>(smem + lds_chunk<T>(LO + v, win, col / P::EPC));
                    }
                }
            }
        }
        if (nmlp > 0) {
            // base_transform: Y = W2 relu(W1 H + b1) + b2 on the first nmlp nodes (hgnn_c2.py:117-121,156); scratch blocks NN + i (hi),
            // LO + NN + i (lo).  The H and T1 stashes are kept packed in registers and stored after the chain.
            P::BFrag bfh, bfl;
            load_bfrag<T>(bfh, wpack, fh[FH_W1], wn, lane);
            load_bfrag<T>(bfl, wpack, a.n_img + fh[FH_W1], wn, lane);
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int n = 2 * u + whv;
                if (n < nmlp) {
                    split_oct(acc[u].c[0], acc[u].c[1], hph[u], hpl[u]);
                    *reinterpret_cast<u32x4*>(smem + lds_chunk<T>(SCR + n, win, col / P::EPC)) = hph[u];
                    *reinterpret_cast<u32x4*>(smem + lds_chunk<T>(LO + SCR + n, win, col / P::EPC)) = hpl[u];
                    acc_init_bias<T>(acc[u], a.bias + (size_t)fh[FH_B1] * H, wn, lane);
                }
            }
            __syncthreads();
            mlp_mac3(acc, smem, SCR, LO, nmlp, whv, bfh, bfl, lane);      // accumulators 0..1 = nodes 0..3
            load_bfrag<T>(bfh, wpack, fh[FH_W2], wn, lane);
            load_bfrag<T>(bfl, wpack, a.n_img + fh[FH_W2], wn, lane);
            __syncthreads();   // all reads of H done before T1 overwrites the scratch blocks
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int n = 2 * u + whv;
                if (n < nmlp) {
                    split_oct(relu4(acc[u].c[0]), relu4(acc[u].c[1]), tph[u], tpl[u]);
                    *reinterpret_cast<u32x4*>(smem + lds_chunk<T>(SCR + n, win, col / P::EPC)) = tph[u];
                    *reinterpret_cast<u32x4*>(smem + lds_chunk<T>(LO + SCR + n, win, col / P::EPC)) = tpl[u];
                    acc_init_bias<T>(acc[u], a.bias + (size_t)fh[FH_B2] * H, wn, lane);
                }
            }
            __syncthreads();
            mlp_mac3(acc, smem, SCR, LO, nmlp, whv, bfh, bfl, lane);      // accumulators 0..1 = nodes 0..3
        }
        FS_STAMP(16 + l);
        // every load issued so far has landed before the first store of the epilogue goes out
        __builtin_amdgcn_s_waitcnt((7 << 4) | (15 << 8));
        mid();      // (FProg::settle of the next header / program: no vmcnt(0) for them at the top of the next layer)
        if (nmlp > 0 && train && w_ok) {
            T* hb = reinterpret_cast<T*>(a.ws + a.hb_off[l]);
            T* t1 = reinterpret_cast<T*>(a.ws + a.t1_off[l]);
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int n = 2 * u + whv;
                if (n < nmlp) {
                    *reinterpret_cast<u32x4*>(hb + x3_idx(w, n, B) + col) = hph[u];
                    *reinterpret_cast<u32x4*>(hb + x3_idx(w, n, B) + H + col) = hpl[u];
                    *reinterpret_cast<u32x4*>(t1 + x3_idx(w, n, B) + col) = tph[u];
                    *reinterpret_cast<u32x4*>(t1 + x3_idx(w, n, B) + H + col) = tpl[u];
                }
            }
        }

        // X_{l+1}[n] = f(H[n]) (+ X_l[n]) for every live node, in place; stash + relu bits on the side
        T* xo = reinterpret_cast<T*>(a.ws + a.x_off[l + 1]);
        uint8_t* maskbytes = reinterpret_cast<uint8_t*>(a.ws + a.mask_off[l]);
        u32x4 resh[FS_HS], resl[FS_HS]; int kindv[FS_HS];     // the residual octets of every node, all LDS reads in flight together
#pragma unroll
        for (int u = 0; u < FS_HS; ++u) {
            const int n = 2 * u + whv;
            kindv[u] = n < NN ? fh[FH_KIND + n] : NK_DEAD;
            resh[u] = u32x4{0, 0, 0, 0}; resl[u] = u32x4{0, 0, 0, 0};
            if (kindv[u] != NK_DEAD && (flags & FF_RESIDUAL)) {
                if (ALIAS && nmlp > 0 && n >= SCR && n < SCR + nmlp) {      // a victim: its block holds T1 by now
                    resh[u] = ((n - SCR) >> 1) == 0 ? vrh[0] : vrh[1]; resl[u] = ((n - SCR) >> 1) == 0 ? vrl[0] : vrl[1];
                } else {
                    resh[u] = *reinterpret_cast<const u32x4*>(smem + lds_chunk<T>(n, win, col / P::EPC));
                    resl[u] = *reinterpret_cast<const u32x4*>(smem + lds_chunk<T>(LO + n, win, col / P::EPC));
                }
            }
        }
#pragma unroll
        for (int u = 0; u < FS_HS; ++u) {
            const int n = 2 * u + whv;
            if (n < NN) {
                const int kind = kindv[u];
                if (kind != NK_DEAD) {
                    if (kind == NK_RELU) {
                        const unsigned bits = relu_with_bits<T>(acc[u]);
                        if (train) maskbytes[relu_byte(n, B, w, col)] = (uint8_t)bits;
                    }
                    f32x4 y0 = acc[u].c[0], y1 = acc[u].c[1];
                    if (flags & FF_RESIDUAL) {
                        f32x4 r0, r1;
                        join_oct(resh[u], resl[u], r0, r1);
                        y0 += r0; y1 += r1;
                    }
                    u32x4 hi, lo;
                    split_oct(y0, y1, hi, lo);
                    *reinterpret_cast<u32x4*>(smem + lds_chunk<T>(n, win, col / P::EPC)) = hi;
                    *reinterpret_cast<u32x4*>(smem + lds_chunk<T>(LO + n, win, col / P::EPC)) = lo;
                    if (train && w_ok && !(STEP && l + 1 == L)) {      // (X_L of a one-launch step is read by nobody)
                        T* q = xo + x3_idx(w, n, B) + col;
                        stash_store(q, hi, a.stash_nt != 0);
                        stash_store(q + H, lo, a.stash_nt != 0);
                    }
                }
            }
        }
        __syncthreads();
        FS_STAMP(5 + 4 * l);
}
// the layers of a compile-time program SP for wave half WH, unrolled
template <bool ALIAS, bool STEP, class SP, int WH, int l = 0>
__device__ __forceinline__ void x3_fwd_layers_static(const StackArgs& a, char* smem, const T16* wpack, int wn, int lane) {
    if constexpr (l < SP::L) {
        x3_fwd_layer<ALIAS, STEP, WH>(a, smem, wpack, wn, WH, lane, l, SP::L, SHdr<SP, 0, l>{}, SProg<SP, 0, l, WH>{}, [] {});
        x3_fwd_layers_static<ALIAS, STEP, SP, WH, l + 1>(a, smem, wpack, wn, lane);
    }
}

template <bool ALIAS, bool STEP, class SP = void> __device__ __forceinline__ void stack_fwd_x3_body(const StackArgs& a, char* smem) {
    using T = T16; using P = P16;
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wv & 3, wh = wv >> 2;
    const int w0 = blockIdx.x * P::ROWS, B = a.B, NN = a.NN, LO = a.lo_blk;
    const T* wpack = reinterpret_cast<const T*>(a.wpack);

    FS_STAMP(0);
    stage_tile_x3(smem, reinterpret_cast<const T*>(a.tile_in), NN, LO, w0, B, tid, [](int) { return true; });
    __syncthreads();
    FS_STAMP(1);

    if constexpr (std::is_void<SP>::value) {
        FHdr fhn(a.tables + a.prog_off[0], lane);
        FProg wpn(a.tables + a.prog_off[0] + FH_SIZE + wh * FPROG_LEN, lane);
        fhn.settle(); wpn.settle();
        for (int l = 0; l < a.L; ++l) {
            const FHdr fh = fhn;
            const FProg wp = wpn;
            if (l + 1 < a.L) {    // the next layer's header and wave program stream in under this layer's MACs
                fhn = FHdr(a.tables + a.prog_off[l + 1], lane);
                wpn = FProg(a.tables + a.prog_off[l + 1] + FH_SIZE + wh * FPROG_LEN, lane);
            }
            x3_fwd_layer<ALIAS, STEP, -1>(a, smem, wpack, wn, wh, lane, l, a.L, fh, wp, [&] { fhn.settle(); wpn.settle(); });
        }
    } else if (wh == 0) x3_fwd_layers_static<ALIAS, STEP, SP, 0>(a, smem, wpack, wn, lane);      // (uniform per wave: each half runs its own straight-line program; the
    else x3_fwd_layers_static<ALIAS, STEP, SP, 1>(a, smem, wpack, wn, lane);                      //  barriers pair up by count)
    decoder_tail<T, LAYER_THREADS, true, STEP>(a, smem, tid, lane, wv, w0, B);
    FS_STAMP(30);
}
template <bool ALIAS> __global__ __launch_bounds__(LAYER_THREADS, 2) void k_stack_fwd_x3(StackArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    stack_fwd_x3_body<ALIAS, false>(a, smem);
}

// ------------------------------------------------------------------------------------------------------
// k_stack_bwd_x3: the L backward layers of a tile -- k_stack_bwd of the bf16 plan on hi/lo planes
// ------------------------------------------------------------------------------------------------------
// STEP: part of k_stack_step_x3 -- the forward's decoder tail of the same launch left the dX_L tile in LDS; the layers' programs are a.prog_off_b
// one backward layer (see x3_fwd_layer); acc carries nothing between layers: the residual term is re-read from the tile
template <int WHS, class FH, class FP, class Mid>
__device__ __forceinline__ void x3_bwd_layer(const StackArgs& a, char* smem, const T16* wpack, int wn, int wh, int lane, int l, const FH& bh, const FP& wp, Mid&& mid) {
    using T = T16; using P = P16;
    const int w0 = blockIdx.x * P::ROWS, B = a.B, NN = a.NN, LO = a.lo_blk;
    const int whv = WHS >= 0 ? WHS : wh;
    P::Acc acc[FS_HS];
        const int nmlp = bh[FH_NMLP], flags = bh[FH_FLAGS];
        const uint8_t* maskbytes = reinterpret_cast<const uint8_t*>(a.ws + a.mask_off[l]);
        // lane constants rebuilt per layer from an opaque copy of the lane id: the per-node 64-bit addresses derived from them were hoisted out of
        // the layer loop and spilled, and a scratch reload next to the epilogue's pending stores is a full vmcnt(0) drain
        const int lq = opaque(lane);
        const int win = c_win(lq), w = w0 + win, col = wn * 32 + c_oct(lq), g8 = (lq >> 4) << 3;
        const bool w_ok = w < B;

        // phase 1 (each lane on the octets it owns): the accumulator of node n starts at its residual term G_{l+1}[n]; relu nodes are
        // then masked in place (both planes) -> dH_l[n]
        {
            unsigned mword[FS_HS]; u32x4 rawh[FS_HS], rawl[FS_HS]; int kindv[FS_HS];
#pragma unroll
            for (int u = 0; u < FS_HS; ++u) {
                const int n = 2 * u + whv;
                kindv[u] = n < NN ? bh[FH_KIND + n] : NK_DEAD;
                mword[u] = 0u; rawh[u] = u32x4{0, 0, 0, 0}; rawl[u] = u32x4{0, 0, 0, 0};
                if (kindv[u] == NK_RELU && w_ok) mword[u] = maskbytes[relu_byte(n, B, w, wn * 32 + g8)];
                if (kindv[u] != NK_DEAD) {
                    rawh[u] = *reinterpret_cast<const u32x4*>(smem + lds_chunk<T>(n, win, col / P::EPC));
                    rawl[u] = *reinterpret_cast<const u32x4*>(smem + lds_chunk<T>(LO + n, win, col / P::EPC));
                }
            }
#pragma unroll
            for (int u = 0; u < FS_HS; ++u) {
                const int n = 2 * u + whv;
                acc_fill(acc[u], 0.f);
                if (kindv[u] != NK_DEAD) {
                    if (bh[FH_RES + n]) join_oct(rawh[u], rawl[u], acc[u].c[0], acc[u].c[1]);
                    if (kindv[u] == NK_RELU) {
                        *reinterpret_cast<u32x4*>(smem + lds_chunk<T>(n, win, col / P::EPC)) = chunk_mask_bits<T>(rawh[u], mword[u]);
                        *reinterpret_cast<u32x4*>(smem + lds_chunk<T>(LO + n, win, col / P::EPC)) = chunk_mask_bits<T>(rawl[u], mword[u]);
                    }
                }
            }
        }
        __syncthreads();

        if (nmlp > 0) {
            // dT1 = dY W2 ; dU = dT1 . (T1 > 0) ; dH = dU W1     (backward of base_transform, in place on nodes 0..nmlp-1)
            const T* t1 = reinterpret_cast<const T*>(a.ws + a.t1_off[l]);
            T* du = reinterpret_cast<T*>(a.ws + a.du_off[l]);
            T* dh = reinterpret_cast<T*>(a.ws + a.dh_off[l]);
            P::BFrag bfh, bfl;
            P::Acc tm[2];
            u32x4 traw[2];
            load_bfrag<T>(bfh, wpack, bh[FH_W2], wn, lane);
            load_bfrag<T>(bfl, wpack, a.n_img + bh[FH_W2], wn, lane);
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int n = 2 * u + whv;
                traw[u] = u32x4{0, 0, 0, 0};
                acc_fill(tm[u], 0.f);
                if (n < nmlp && w_ok) traw[u] = *reinterpret_cast<const u32x4*>(t1 + x3_idx(w, n, B) + col);     // the hi half carries the sign of T1
            }
            mlp_mac3(tm, smem, 0, LO, nmlp, whv, bfh, bfl, lane);
            load_bfrag<T>(bfh, wpack, bh[FH_W1], wn, lane);
            load_bfrag<T>(bfl, wpack, a.n_img + bh[FH_W1], wn, lane);
            __syncthreads();   // all reads of the dY blocks done
            u32x4 duh[2] = {}, dul[2] = {};
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int n = 2 * u + whv;
                if (n < nmlp) {
                    f32x4 t0, t1v, r0, r1; unpack_oct(traw[u], t0, t1v);
#pragma unroll
                    for (int j = 0; j < 4; ++j) { r0[j] = t0[j] > 0.f ? tm[u].c[0][j] : 0.f; r1[j] = t1v[j] > 0.f ? tm[u].c[1][j] : 0.f; }
                    split_oct(r0, r1, duh[u], dul[u]);
                    *reinterpret_cast<u32x4*>(smem + lds_chunk<T>(n, win, col / P::EPC)) = duh[u];
                    *reinterpret_cast<u32x4*>(smem + lds_chunk<T>(LO + n, win, col / P::EPC)) = dul[u];
                    acc_fill(tm[u], 0.f);
                }
            }
            __syncthreads();
            mlp_mac3(tm, smem, 0, LO, nmlp, whv, bfh, bfl, lane);
            __syncthreads();   // all reads of the dU blocks done
            __builtin_amdgcn_s_waitcnt((7 << 4) | (15 << 8));
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int n = 2 * u + whv;
                if (n < nmlp) {
                    u32x4 hh, hl;
                    split_oct(tm[u].c[0], tm[u].c[1], hh, hl);
                    *reinterpret_cast<u32x4*>(smem + lds_chunk<T>(n, win, col / P::EPC)) = hh;
                    *reinterpret_cast<u32x4*>(smem + lds_chunk<T>(LO + n, win, col / P::EPC)) = hl;
                    if (w_ok) {
                        *reinterpret_cast<u32x4*>(du + x3_idx(w, n, B) + col) = duh[u];
                        *reinterpret_cast<u32x4*>(du + x3_idx(w, n, B) + H + col) = dul[u];
                        *reinterpret_cast<u32x4*>(dh + x3_idx(w, n, B) + col) = hh;
                        *reinterpret_cast<u32x4*>(dh + x3_idx(w, n, B) + H + col) = hl;
                    }
                }
            }
            __syncthreads();
        }

        // phase 2: dX_l[j] = (residual) + dH_j W_rootsum + sum_r sum_{j->i} dH_i W_rel^r
        fs_run<T>(wp, acc, smem, wpack, wn, lane);
        __syncthreads();   // every wave is done reading dH_l

        T* dxo = reinterpret_cast<T*>(a.ws + a.dx_off[l]);
        const uint8_t* m0 = reinterpret_cast<const uint8_t*>(a.ws + a.mask0_off);
        mid();      // next header / program landed before the stores go out (FProg::settle)
        // layer 0: the encoder's relu bytes of every node are requested before the first store of the epilogue (one round trip; a load waited for
        // while stores are in flight drains them all)
        unsigned xbv[FS_HS];
#pragma unroll
        for (int u = 0; u < FS_HS; ++u) {
            const int n = 2 * u + whv;
            xbv[u] = 0xffu;
            if ((flags & FF_ENC_MASK) && w_ok && n < NN && bh[FH_OUT + n]) xbv[u] = m0[relu_byte(n, B, w, wn * 32 + g8)];
        }
#pragma unroll
        for (int u = 0; u < FS_HS; ++u) {
            const int n = 2 * u + whv;
            if (n < NN && bh[FH_OUT + n]) {
                f32x4 y0 = acc[u].c[0], y1 = acc[u].c[1];
                if ((flags & FF_ENC_MASK) && w_ok) {   // layer 0: x relu'(X_0): the encoder's relu byte of this lane
                    const unsigned xb = xbv[u];
#pragma unroll
                    for (int j = 0; j < 4; ++j) { y0[j] = ((xb >> j) & 1u) ? y0[j] : 0.f; y1[j] = ((xb >> (4 + j)) & 1u) ? y1[j] : 0.f; }
                }
                u32x4 hi, lo;
                split_oct(y0, y1, hi, lo);
                if (l > 0) {
                    *reinterpret_cast<u32x4*>(smem + lds_chunk<T>(n, win, col / P::EPC)) = hi;
                    *reinterpret_cast<u32x4*>(smem + lds_chunk<T>(LO + n, win, col / P::EPC)) = lo;
                }
                if (w_ok) {
                    T* q = dxo + x3_idx(w, n, B) + col;
                    stash_store(q, hi, a.stash_nt != 0);
                    stash_store(q + H, lo, a.stash_nt != 0);
                }
            }
        }
        __syncthreads();
}
template <class SP, int WH, int l>
__device__ __forceinline__ void x3_bwd_layers_static(const StackArgs& a, char* smem, const T16* wpack, int wn, int lane) {
    if constexpr (l >= 0) {
        x3_bwd_layer<WH>(a, smem, wpack, wn, WH, lane, l, SHdr<SP, 1, l>{}, SProg<SP, 1, l, WH>{}, [] {});
        x3_bwd_layers_static<SP, WH, l - 1>(a, smem, wpack, wn, lane);
    }
}

template <bool STEP, class SP = void> __device__ __forceinline__ void stack_bwd_x3_body(const StackArgs& a, char* smem) {
    using T = T16; using P = P16;
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wv & 3, wh = wv >> 2;
    const int w0 = blockIdx.x * P::ROWS, B = a.B, NN = a.NN, LO = a.lo_blk;
    const T* wpack = reinterpret_cast<const T*>(a.wpack);
    auto prog_of = [&](int l) { return STEP ? a.prog_off_b[l] : a.prog_off[l]; };

    // dX_L tile: only the nodes that are live in the last layer carry a gradient
    if constexpr (!STEP) {
        const FHdr bh(a.tables + prog_of(a.L - 1), lane);
        stage_tile_x3(smem, reinterpret_cast<const T*>(a.tile_in), NN, LO, w0, B, tid, [&](int n) { return bh[FH_KIND + n] != NK_DEAD; });
    }
    __syncthreads();

    if constexpr (std::is_void<SP>::value) {
        FHdr bhn(a.tables + prog_of(a.L - 1), lane);
        FProg wpn(a.tables + prog_of(a.L - 1) + FH_SIZE + wh * FPROG_LEN, lane);
        bhn.settle(); wpn.settle();
        for (int l = a.L - 1; l >= 0; --l) {
            const FHdr bh = bhn;
            const FProg wp = wpn;
            if (l > 0) {          // the next layer's header and wave program stream in under this layer's MACs
                bhn = FHdr(a.tables + prog_of(l - 1), lane);
                wpn = FProg(a.tables + prog_of(l - 1) + FH_SIZE + wh * FPROG_LEN, lane);
            }
            x3_bwd_layer<-1>(a, smem, wpack, wn, wh, lane, l, bh, wp, [&] { bhn.settle(); wpn.settle(); });
        }
    } else if (wh == 0) x3_bwd_layers_static<SP, 0, SP::L - 1>(a, smem, wpack, wn, lane);
    else x3_bwd_layers_static<SP, 1, SP::L - 1>(a, smem, wpack, wn, lane);
}
#if MSHGNN_SPEC_SHARD == 0
__global__ __launch_bounds__(LAYER_THREADS, 2) void k_stack_bwd_x3(StackArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    stack_bwd_x3_body<false>(a, smem);
}
#endif
// mshgnn_step_mse on the split plan: both sweeps of a tile in one launch (k_slab_step of mshgnn.hip: dX_L stays in LDS, no second launch, no tile reload)
// SP: void = the plan's tables are interpreted; else the compile-time program of one (topology, depth) on the split plan (mshgnn_spec_tables.inc, X3_*)
template <bool ALIAS, class SP = void> __global__ __launch_bounds__(LAYER_THREADS, 2) void k_stack_step_x3(StackArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    stack_fwd_x3_body<ALIAS, true, SP>(a, smem);
    __syncthreads();
    stack_bwd_x3_body<true, SP>(a, smem);
}
// the forward launch alone (evaluation / first call of the two-call training route: what the nn.Module surface -- default precision "x3" -- runs) and the backward launch alone over
// the same programs; predicated stores like the interpreters', so ragged batches take them too.  Same MACs, same order: the interpreters' bits.
template <bool ALIAS, class SP> __global__ __launch_bounds__(LAYER_THREADS, 2) void k_stack_fwd_x3_spec(StackArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    stack_fwd_x3_body<ALIAS, false, SP>(a, smem);
}
template <class SP> __global__ __launch_bounds__(LAYER_THREADS, 2) void k_stack_bwd_x3_spec(StackArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    stack_bwd_x3_body<false, SP>(a, smem);
}

// Specialised one-call step kernels of the split plan: k_stack_step_x3 over the compile-time program of one (topology, depth) -- mshgnn_spec_tables.inc (X3_*),
// generated from this library's own plan compiler.  A plan takes one only when its fused tables are exactly the ints the kernel was compiled from.  The kernels are
// instantiated in translation units of their own (this source with -DMSHGNN_SPEC_SHARD=1..7: csrc/Makefile), side by side with the rest.
#include "mshgnn_spec_tables.inc"
using StackKernelX3 = void (*)(StackArgs);
template <class SP> static bool spec_matches_x3(const HostPlan& hp) {
    if (!hp.split || !hp.fused || hp.L != SP::L || hp.NN != SP::NN || (hp.x3_alias ? 1 : 0) != SP::ALIAS) return false;
    for (int l = 0; l < SP::L; ++l) {
        if (hp.fs_fwd_off[l] + SP::ROW > (int)hp.tables.size() || hp.fs_bwd_off[l] + SP::ROW > (int)hp.tables.size()) return false;
        if (memcmp(hp.tables.data() + hp.fs_fwd_off[l], SP::fwd[l], sizeof(int32_t) * SP::ROW) != 0) return false;
        if (memcmp(hp.tables.data() + hp.fs_bwd_off[l], SP::bwd[l], sizeof(int32_t) * SP::ROW) != 0) return false;
    }
    return true;
}
#define X3_SHARD_LIST(X) X(1) X(2) X(3) X(4) X(5) X(6) X(7)      // one program per shard (tools/gen_spec_tables.py X3_SHARDS; a split-plan step kernel compiles for 75-95 s)
#define X3_SHARD_DECL(k) StackKernelX3 x3_spec_shard##k(const HostPlan& hp, int kind, const char** name);      // kind 0: one-call step, 1: forward alone, 2: backward alone
X3_SHARD_LIST(X3_SHARD_DECL)
#if MSHGNN_SPEC_SHARD != 0
#define MSHGNN_SPEC_TRY(SP) if (spec_matches_x3<SP>(hp)) { if (name) *name = #SP; \
        return kind == 0 ? k_stack_step_x3<SP::ALIAS != 0, SP> : (kind == 1 ? k_stack_fwd_x3_spec<SP::ALIAS != 0, SP> : k_stack_bwd_x3_spec<SP>); }
#define X3_CAT2(a, b) a##b
#define X3_CAT(a, b) X3_CAT2(a, b)
#if MSHGNN_SPEC_SHARD == 99      // a program compiled for one plan after the build (morphsym_hgnn_amd/jit.py; see shard 99 of mshgnn.hip)
#include MSHGNN_JIT_TABLES
#endif
StackKernelX3 X3_CAT(x3_spec_shard, MSHGNN_SPEC_SHARD)(const HostPlan& hp, int kind, const char** name) { X3_CAT(MSHGNN_SPEC_X3_LIST_, MSHGNN_SPEC_SHARD)(MSHGNN_SPEC_TRY) return nullptr; }
#if MSHGNN_SPEC_SHARD == 99
extern "C" StackKernelX3 mshgnn_jit_program_x3(const HostPlan& hp, int kind, const char** name) { return x3_spec_shard99(hp, kind, name); }
#endif
#undef MSHGNN_SPEC_TRY
#else      // MSHGNN_SPEC_SHARD == 0: the library proper, to the end of this file
using SpecSelectorX3 = StackKernelX3 (*)(const HostPlan& hp, int kind, const char** name);
static StackKernelX3 x3_spec_kernel(const HostPlan& hp, int kind, const char** name = nullptr) {
    if (hp.jit_prog) if (StackKernelX3 kk = reinterpret_cast<SpecSelectorX3>(hp.jit_prog)(hp, kind, name)) return kk;      // a program compiled for this plan after the build
#define X3_SHARD_TRY(k) if (StackKernelX3 kk = x3_spec_shard##k(hp, kind, name)) return kk;
    X3_SHARD_LIST(X3_SHARD_TRY)
#undef X3_SHARD_TRY
    return nullptr;
}
static StackKernelX3 x3_step_spec_kernel(const HostPlan& hp, const char** name = nullptr) { return x3_spec_kernel(hp, 0, name); }


// ------------------------------------------------------------------------------------------------------
// k_dec_bwd_x3: decoder backward (+ fused wrapper MSE / cross entropy) on the hi/lo planes of X_L -> dX_L planes
// (k_dec_bwd of mshgnn.hip; used by mshgnn_backward / _mse / _ce -- mshgnn_step_mse takes the fused tail of k_stack_fwd_x3)
// ------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_dec_bwd_x3(DecArgs a) {
    using T = T16;
    __shared__ float red[16][DEC_SLAB_FLOATS];
    const int c = threadIdx.x & 15, rg = threadIdx.x >> 4;
    const int64_t rows = (int64_t)a.B * a.n_out;
    const int64_t per = ((rows + gridDim.x - 1) / gridDim.x + 15) / 16 * 16;
    const int64_t r_begin = (int64_t)blockIdx.x * per, r_end = min(rows, r_begin + per);
    const T* xl = reinterpret_cast<const T*>(a.xl);
    T* dxl = reinterpret_cast<T*>(a.dxl);
    const float* W = a.params + a.off_w;
    float accw[8][8], accb[8];
#pragma unroll
    for (int d = 0; d < 8; ++d) { accb[d] = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) accw[d][e] = 0.f; }
    float lsum = 0.f;
    for (int64_t r = r_begin + rg; r < r_end; r += 16) {
        const int w = (int)(r / a.n_out), f = (int)(r % a.n_out);
        const size_t idx = x3_idx(w, a.node0 + f, a.B) + c * 8;
        float x[8], xlo[8], dx[8];
        load8<T>(xl + idx, x);
        load8<T>(xl + idx + H, xlo);
#pragma unroll
        for (int e = 0; e < 8; ++e) { x[e] += xlo[e]; dx[e] = 0.f; }
        float ce_g[2] = {0.f, 0.f};
        if (a.labels) {   // wrapper cross entropy fused (gnnLightning.py:640-648, mean over the batch * 4 feet): dL/dlogit = (p - onehot) / rows
            const float l0 = a.out[r * 2], l1 = a.out[r * 2 + 1];
            const float m = fmaxf(l0, l1), e0 = expf(l0 - m), e1 = expf(l1 - m), se = e0 + e1;
            const int lab = a.labels[r] != 0;
            ce_g[0] = (e0 / se - (lab ? 0.f : 1.f)) * a.inv_n; ce_g[1] = (e1 / se - (lab ? 1.f : 0.f)) * a.inv_n;
            if (c == 0) lsum += (m + logf(se)) - (lab ? l1 : l0);
        }
#pragma unroll
        for (int d = 0; d < 8; ++d) {
            if (d < a.dout) {
                float go;
                if (a.labels) go = ce_g[d & 1];
                else if (a.y) {   // wrapper MSE fused (gnnLightning.py:633-639): dL/dout = 2 (out - y) / n
                    const float dlt = a.out[r * a.dout + d] - a.y[r * a.dout + d];
                    go = 2.0f * dlt * a.inv_n;
                    if (c == 0) lsum += dlt * dlt;
                } else go = a.gout[r * a.dout + d];
                const float g = go * a.out_mask[f * a.dout + d];
                accb[d] += g;
#pragma unroll
                for (int e = 0; e < 8; ++e) { accw[d][e] += g * x[e]; dx[e] += g * W[d * H + c * 8 + e]; }
            }
        }
        u32x4 hi, lo;
        split_oct(f32x4{dx[0], dx[1], dx[2], dx[3]}, f32x4{dx[4], dx[5], dx[6], dx[7]}, hi, lo);
        *reinterpret_cast<u32x4*>(dxl + idx) = hi;
        *reinterpret_cast<u32x4*>(dxl + idx + H) = lo;
    }
#pragma unroll
    for (int d = 0; d < 8; ++d) {
#pragma unroll
        for (int e = 0; e < 8; ++e) red[rg][d * H + c * 8 + e] = accw[d][e];
        if (c == 0) red[rg][8 * H + d] = accb[d];
    }
    __syncthreads();
    float* slab = a.slabs + (size_t)blockIdx.x * DEC_SLAB_FLOATS;
    {   // per-block loss partial rides in the slab (summed in fixed order by k_finalize: no atomics, deterministic)
        __shared__ float lred[4];
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) lsum += __shfl_xor(lsum, m, 64);
        if ((threadIdx.x & 63) == 0) lred[threadIdx.x >> 6] = lsum;
        __syncthreads();
        if (threadIdx.x == 0) slab[8 * H + 8] = (lred[0] + lred[1]) + (lred[2] + lred[3]);
    }
    for (int i = threadIdx.x; i < 8 * H + 8; i += 256) {
        float s2 = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) s2 += red[r][i];
        slab[i] = s2;
    }
}

// ------------------------------------------------------------------------------------------------------
// k_gradw_x3: all weight gradients of the step as one split-K MFMA launch, dW[o][k] = sum_w P[w][o] Q[w][k] with
// P = P_hi + P_lo, Q = Q_hi + Q_lo: four staged tiles per 64-window step, P_hi Q_hi + P_hi Q_lo + P_lo Q_hi on the 32x32x16 bf16
// MFMA.  Workgroup = (lane = one item, window part); raw encoder inputs (fp32) are split while they are staged.
// ------------------------------------------------------------------------------------------------------
#ifndef GWX3_KW
#define GWX3_KW 32          // windows per staged step: 32 KB of tiles and three workgroups per CU (64-window steps at two per CU: 339 vs 301 us)
#endif
#ifndef GWX3_WPS
#define GWX3_WPS 3
#endif
#ifndef GWX3_NST
#define GWX3_NST 1          // register stages: global loads run NST steps ahead of the LDS writes
#endif
#ifdef MSHGNN_TUNING      // (the general split-plan weight-gradient kernel of round 2: A/B runs in tuning builds only; the lean kernel below is the product's)
__global__ __launch_bounds__(256, GWX3_WPS) void k_gradw_x3(GradwArgs a) {
    using T = T16;
    constexpr int KW = GWX3_KW, NP = KW / 16, NST = GWX3_NST;      // NP: staging passes (16 rows each)
    __shared__ __attribute__((aligned(16))) __bf16 Ph[KW * GWB_PITCH];
    __shared__ __attribute__((aligned(16))) __bf16 Pl[KW * GWB_PITCH];
    __shared__ __attribute__((aligned(16))) __bf16 Qh[KW * GWB_PITCH];
    __shared__ __attribute__((aligned(16))) __bf16 Ql[KW * GWB_PITCH];
    __shared__ __attribute__((aligned(16))) u32x4 mlut[256];       // relu byte -> AND mask of 8 bf16
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    mlut[tid] = chunk_mask_bits<__bf16>(u32x4{0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu}, (unsigned)tid);   // (visible after the first barrier)
    const int wr = wv >> 1, wc = wv & 1;
    const int ln = a.lane_order[blockIdx.x % a.n_pad], part = blockIdx.x / a.n_pad;
    if (ln < 0) return;
    const int* lh = a.lanes + ln * LANE_INTS;
    const int bias_flag = lh[3];
    const int nchunks = (a.B + KW - 1) / KW;
    const int ch0 = (int)((int64_t)part * nchunks / a.n_parts), ch1 = (int)((int64_t)(part + 1) * nchunks / a.n_parts);
    const int nsteps = ch1 - ch0;            // one item per lane on the split plan
    const int c = tid & 15, r0 = tid >> 4;   // staging: 16 chunks of 8 elements per row, 16 rows per pass

    // Every operand address is a workgroup-uniform base (scalar registers, from the item table) + one small per-thread offset: the
    // six 64-bit per-thread pointers this replaces were what pushed the kernel over its register budget.
    const int* im = a.items + lh[0] * ITEM_INTS;
    const T* p_hi = reinterpret_cast<const T*>(a.ws + a.buf_off[im[0]]) + x3_idx(0, im[2], a.B);      // rows of [hi 128 | lo 128]
    const T* p_lo = p_hi + H;
    const bool p_masked = im[9] >= 0;
    const uint8_t* mb = p_masked ? reinterpret_cast<const uint8_t*>(a.ws + a.buf_off[im[9]]) + (((size_t)im[2] * 4) * ((a.B + 15) >> 4) << 6) : nullptr;
    const int moff = (int)((((size_t)(c >> 2)) * ((a.B + 15) >> 4)) << 6) + ((c & 3) << 4) + r0;      // relu_byte(node, B, w, 8 c) = node base + moff + 64 (w >> 4)   (w & 15 == r0)
    const bool q_act = im[4] >= 0;           // Q is an activation stash (two bf16 planes) / a raw fp32 input
    const T* q_hi = nullptr; const T* q_lo = nullptr; const float* qf = nullptr;
    int64_t qstride = H; int qvalid = 8, qvb = 16;
    u32x4 sxa = u32x4{0, 0, 0, 0}, sxb = u32x4{0, 0, 0, 0};
    if (q_act) {
        q_hi = reinterpret_cast<const T*>(a.ws + a.buf_off[im[3]]) + x3_idx(0, im[5], a.B);
        q_lo = q_hi + H;
    } else {
        const int t = im[3] - BUF_IN;
        qf = reinterpret_cast<const float*>(a.x[t]) + (size_t)im[5] * a.pitch[t] + im[6];
        qstride = (int64_t)a.nodes[t] * a.pitch[t]; qvalid = im[7] - c * 8; qvb = a.vb[t];
        sxa = sign_xor<float>(a.signs + im[8] + c * 8); sxb = sign_xor<float>(a.signs + im[8] + c * 8 + 4);
    }
    const int loff = r0 * 2 * H + c * 8;                 // element offset of this thread inside a 16-row pass of an activation tensor
    const int64_t qoff = (int64_t)r0 * qstride + c * 8;  // same for a raw input (floats)
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.f;
    float bsum[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) bsum[e] = 0.f;

    struct Stage { u32x4 ph[NP], pl[NP], qa[NP], qb[NP]; unsigned mw[NP]; };     // qa / qb: the hi / lo plane rows, or the two fp32 halves of a raw row
    auto fetch = [&](Stage& st, int s) {
        const int w0 = (ch0 + s) * KW;       // uniform
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            const int wb = w0 + 16 * p, w = wb + r0;
            st.ph[p] = u32x4{0, 0, 0, 0}; st.pl[p] = u32x4{0, 0, 0, 0}; st.qa[p] = u32x4{0, 0, 0, 0}; st.qb[p] = u32x4{0, 0, 0, 0}; st.mw[p] = 0xffffffffu;
            if (w < a.B) {
                st.ph[p] = ld16<(GW_NT > 1)>(p_hi + (size_t)wb * 2 * H + loff);
                st.pl[p] = ld16<(GW_NT > 1)>(p_lo + (size_t)wb * 2 * H + loff);
                if (p_masked) st.mw[p] = (mb + ((size_t)(wb >> 4) << 6))[moff];
                if (q_act) {
                    st.qa[p] = ld16<(GW_NT > 1)>(q_hi + (size_t)wb * 2 * H + loff);
                    st.qb[p] = ld16<(GW_NT > 1)>(q_lo + (size_t)wb * 2 * H + loff);
                } else if (a.aligned) {      // raw: a use here would serialise the loads
                    const float* q = qf + (size_t)wb * qstride + qoff;
                    if (qvalid > 0) st.qa[p] = ld16<(GW_NT > 0)>(q);
                    if (qvalid > 4) st.qb[p] = ld16<(GW_NT > 0)>(q + 4);
                } else {
                    const float* q = qf + (size_t)wb * qstride + qoff;
                    st.qa[p] = load_chunk<float>(q, qvalid, qvb);
                    st.qb[p] = load_chunk<float>(q + 4, qvalid - 4, qvb);
                }
            }
        }
    };
    auto stage_to_lds = [&](const Stage& st) {
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            const int row = r0 + 16 * p;
            u32x4 ph = st.ph[p], pl = st.pl[p];
            if (p_masked) { const u32x4 m = mlut[st.mw[p] & 0xffu]; ph &= m; pl &= m; }     // dH = dX . relu bits
            *reinterpret_cast<u32x4*>(&Ph[gwb_elem(row, c * 8)]) = ph;
            *reinterpret_cast<u32x4*>(&Pl[gwb_elem(row, c * 8)]) = pl;
            u32x4 qh = st.qa[p], ql = st.qb[p];
            if (!q_act) {      // drop pad columns, symmetry sign mask, fp32 -> hi / lo
                const u32x4 fa = chunk_keep_first<float>(st.qa[p], qvalid) ^ sxa, fb = chunk_keep_first<float>(st.qb[p], qvalid - 4) ^ sxb;
                split_oct(__builtin_bit_cast(f32x4, fa), __builtin_bit_cast(f32x4, fb), qh, ql);
            }
            *reinterpret_cast<u32x4*>(&Qh[gwb_elem(row, c * 8)]) = qh;
            *reinterpret_cast<u32x4*>(&Ql[gwb_elem(row, c * 8)]) = ql;
            if (bias_flag) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    bsum[2 * e] += __builtin_bit_cast(float, ph[e] << 16) + __builtin_bit_cast(float, pl[e] << 16);
                    bsum[2 * e + 1] += __builtin_bit_cast(float, ph[e] & 0xffff0000u) + __builtin_bit_cast(float, pl[e] & 0xffff0000u);
                }
            }
        }
    };
    auto mfmas = [&]() {
#pragma unroll
        for (int ks = 0; ks < KW / 16; ++ks) {
            bf16x8 afh[2], afl[2], bqh[2], bql[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                afh[i] = tr_frag(Ph, ks * 16, wr * 64 + i * 32, lane);
                afl[i] = tr_frag(Pl, ks * 16, wr * 64 + i * 32, lane);
                bqh[i] = tr_frag(Qh, ks * 16, wc * 64 + i * 32, lane);
                bql[i] = tr_frag(Ql, ks * 16, wc * 64 + i * 32, lane);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afh[i], bqh[j], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afh[i], bql[j], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afl[i], bqh[j], acc[i][j], 0, 0, 0);
                }
        }
    };
    if constexpr (NST == 1) {
        Stage sa;
        if (nsteps > 0) fetch(sa, 0);
        for (int s = 0; s < nsteps; ++s) {
            __syncthreads();      // the previous MFMA phase of every wave is done reading the tiles
            stage_to_lds(sa);
            __syncthreads();
            if (s + 1 < nsteps) fetch(sa, s + 1);
            mfmas();
        }
    } else {
        // two register stages: the loads of steps s + 1 and s + 2 are in flight while step s is multiplied
        Stage sa, sb;
        if (nsteps > 0) fetch(sa, 0);
        if (nsteps > 1) fetch(sb, 1);
        for (int s = 0; s < nsteps; s += 2) {
            __syncthreads();
            stage_to_lds(sa);
            __syncthreads();
            if (s + 2 < nsteps) fetch(sa, s + 2);
            mfmas();
            if (s + 1 < nsteps) {
                __syncthreads();
                stage_to_lds(sb);
                __syncthreads();
                if (s + 3 < nsteps) fetch(sb, s + 3);
                mfmas();
            }
        }
    }
    float* slab = a.slabs + (size_t)(part * a.n_lanes + ln) * SLAB_FLOATS;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int o = wr * 64 + i * 32 + (q & 3) + ((q >> 2) << 3) + ((lane >> 5) << 2), k = wc * 64 + j * 32 + (lane & 31);
                slab[o * H + k] = acc[i][j][q];
            }
    if (bias_flag) {
        float* red = reinterpret_cast<float*>(Ph);   // 16 x 128 floats = 8 KB <= one tile
        __syncthreads();
#pragma unroll
        for (int e = 0; e < 8; ++e) red[r0 * H + c * 8 + e] = bsum[e];
        __syncthreads();
        if (tid < H) {
            float s2 = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) s2 += red[r * H + tid];
            slab[H * H + tid] = s2;
        }
    }
}
#endif

// k_gradw_x3_lean: the same 32-window step loop with the addressing of k_gradw_bf16_lean (mshgnn.hip): a thread owns TWO CONSECUTIVE windows of
// one 16-byte column chunk, so every global address is  workgroup-uniform stream pointer (scalar registers, advanced by scalar adds)  +  ONE
// per-thread 32-bit offset  +  an immediate (the [hi | lo] row layout puts the four loads of an operand at +0 / +256 / +512 / +768), the relu
// bytes of its two rows are one aligned 16-bit load, and full chunks carry no bound checks.  The general kernel above keeps six 64-bit
// per-thread pointers alive and spills three of them (28 B of scratch per lane, reloaded inside the step loop in front of the loads that
// need them).  A lane's items (all of one target) are swept one after the other over the lane's window part into the same accumulators (any
// number of items per lane: the plan picks the items-per-lane x window-parts split that fills the chip best).  ALIGNED: raw inputs in the
// engine's own 16-byte-aligned layout; the other instantiation reads them element-wise.
template <bool ALIGNED> __global__ __launch_bounds__(256, GWX3_WPS) void k_gradw_x3_lean(GradwArgs a) {
    using T = T16;
    constexpr int KW = 32;
    __shared__ __attribute__((aligned(16))) __bf16 Ph[KW * GWB_PITCH];
    __shared__ __attribute__((aligned(16))) __bf16 Pl[KW * GWB_PITCH];
    __shared__ __attribute__((aligned(16))) __bf16 Qh[KW * GWB_PITCH];
    __shared__ __attribute__((aligned(16))) __bf16 Ql[KW * GWB_PITCH];
    __shared__ __attribute__((aligned(16))) u32x4 mlut[256];       // relu byte -> AND mask of 8 bf16
    __shared__ __attribute__((aligned(16))) u32x4 qfix[16][2];            // raw-input items: per column chunk c the keep masks of its two fp32 halves
    __shared__ __attribute__((aligned(16))) u32x4 qsign_t[GW_IPL][16][2]; // ... and per (item, column chunk) the symmetry sign XORs
    __shared__ unsigned long long sbase[GW_IPL][4];                       // per item: stream bases of P, relu bytes, Q at the part's first window
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    mlut[tid] = chunk_mask_bits<__bf16>(u32x4{0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu}, (unsigned)tid);   // (visible after the first barrier)
    const int wr = wv >> 1, wc = wv & 1;
    const int ln = a.lane_order[blockIdx.x % a.n_pad], part = blockIdx.x / a.n_pad;
    if (ln < 0) return;
    const int* lh = a.lanes + ln * LANE_INTS;
    const int it0 = lh[0], nit = lh[1] - lh[0], bias_flag = lh[3];
    const int* im0 = a.items + it0 * ITEM_INTS;
    const int nchunks = (a.B + KW - 1) / KW;
    const int ch0 = (int)((int64_t)part * nchunks / a.n_parts), ch1 = (int)((int64_t)(part + 1) * nchunks / a.n_parts);
    const int nsteps = ch1 - ch0, total = nsteps * nit;      // the lane's items interleaved chunk by chunk (every workgroup sweeps the batch at the same pace), into the same accumulators
    const int c = tid & 15, r2 = (tid >> 4) * 2;      // rows r2, r2 + 1 of the 32-window step, columns [8c, 8c + 8)
    constexpr int ROWB = 2 * H * (int)sizeof(T);      // bytes of one [hi | lo] row

    // what every item of the lane shares (one target): operand kinds, the input type and column chunk of a raw Q
    const bool p_masked = im0[9] >= 0, q_act = im0[4] >= 0;
    const int ntile = (a.B + 15) >> 4;
    const unsigned voffP = (unsigned)(r2 * ROWB + c * 16);
    const unsigned voffM = (unsigned)(((c >> 2) * ntile + (r2 >> 4)) * 64 + (c & 3) * 16 + (r2 & 15));
    const int qt = q_act ? 0 : im0[3] - BUF_IN;
    const unsigned qsb = q_act ? (unsigned)ROWB : (unsigned)(a.nodes[qt] * a.pitch[qt]) * 4u;
    const int qn = q_act ? 8 : im0[7] - c * 8, qvb = q_act ? 16 : a.vb[qt];
    if (!q_act && tid < 16) {      // (these constants would cost 16 VGPRs per lane: over the budget of three workgroups per CU)
        const u32x4 ones = u32x4{0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu};
        qfix[tid][0] = chunk_keep_first<float>(ones, qn); qfix[tid][1] = chunk_keep_first<float>(ones, qn - 4);
    }
    // raw rows: two 16-byte halves of 8 floats; halves past the row's end re-read the chunk's first bytes (never used: the keep masks are 0)
    const unsigned voffQ0 = q_act ? voffP : (unsigned)r2 * qsb + (qn > 0 ? (unsigned)c * 32u : 0u);
    const unsigned qhalf = q_act ? 256u : (qn > 4 ? 16u : 0u);
    if (tid < nit) {     // thread k resolves item k's stream bases (at the part's first window)
        const int* im = a.items + (it0 + tid) * ITEM_INTS;
        sbase[tid][0] = (unsigned long long)(a.ws + a.buf_off[im[0]] + (x3_idx(0, im[2], a.B) + (size_t)ch0 * KW * 2 * H) * sizeof(T));
        sbase[tid][1] = p_masked ? (unsigned long long)(a.ws + a.buf_off[im[9]] + relu_byte(im[2], a.B, 0, 0) + (size_t)ch0 * (KW / 16) * 64) : 0ull;
        sbase[tid][2] = q_act ? (unsigned long long)(a.ws + a.buf_off[im[3]] + (x3_idx(0, im[5], a.B) + (size_t)ch0 * KW * 2 * H) * sizeof(T))
                              : (unsigned long long)(reinterpret_cast<const char*>(a.x[qt]) + ((size_t)im[5] * a.pitch[qt] + im[6]) * 4 + (size_t)ch0 * KW * qsb);
    }
    if (!q_act && tid < nit * 16) {
        const int* im = a.items + (it0 + (tid >> 4)) * ITEM_INTS;
        qsign_t[tid >> 4][c][0] = sign_xor<float>(a.signs + im[8] + c * 8); qsign_t[tid >> 4][c][1] = sign_xor<float>(a.signs + im[8] + c * 8 + 4);
    }
    __syncthreads();
    const char* pS = a.ws; const char* mS = a.ws; const char* qS = a.ws;
    auto ubase = [&](int k, int j, size_t off) -> const char* {      // workgroup-uniform 64-bit pointer out of the LDS table
        const unsigned long long v = sbase[k][j] + off;
        return reinterpret_cast<const char*>(((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(v >> 32)) << 32) |
                                             (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)v));
    };
    unsigned ldsw[2];
#pragma unroll
    for (int p = 0; p < 2; ++p) ldsw[p] = (unsigned)gwb_elem(r2 + p, c * 8);
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.f;
    float bsum[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) bsum[e] = 0.f;

    u32x4 ph[2], pl[2], qa[2], qb[2]; unsigned mw = 0xffffu;
    auto load_qa = [&](int p) -> u32x4 {
        if constexpr (ALIGNED) return *reinterpret_cast<const u32x4*>(qS + voffQ0 + p * qsb);
        else return q_act ? *reinterpret_cast<const u32x4*>(qS + voffQ0 + p * qsb) : load_chunk<float>(reinterpret_cast<const float*>(qS + (unsigned)(r2 + p) * qsb + (unsigned)c * 32u), qn, qvb);
    };
    auto load_qb = [&](int p) -> u32x4 {
        if constexpr (ALIGNED) return *reinterpret_cast<const u32x4*>(qS + voffQ0 + p * qsb + qhalf);
        else return q_act ? *reinterpret_cast<const u32x4*>(qS + voffQ0 + p * qsb + qhalf) : load_chunk<float>(reinterpret_cast<const float*>(qS + (unsigned)(r2 + p) * qsb + (unsigned)c * 32u) + 4, qn - 4, qvb);
    };
    int nk = 0, nc = 0;            // (item, chunk of the part) of the next fetch: items interleaved chunk by chunk
    auto fetch = [&]() {
        pS = ubase(nk, 0, (size_t)nc * KW * ROWB);
        if (p_masked) mS = ubase(nk, 1, (size_t)nc * (KW / 16) * 64);
        qS = ubase(nk, 2, (size_t)nc * KW * qsb);
        const int w0 = (ch0 + nc) * KW;
        if (w0 + KW <= a.B) {
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                ph[p] = *reinterpret_cast<const u32x4*>(pS + voffP + p * ROWB);
                pl[p] = *reinterpret_cast<const u32x4*>(pS + voffP + p * ROWB + 256);
                qa[p] = load_qa(p);
                qb[p] = load_qb(p);
            }
            if (p_masked) mw = *reinterpret_cast<const unsigned short*>(mS + voffM);
        } else {                   // last chunk of the batch: rows beyond B are zero
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                ph[p] = u32x4{0, 0, 0, 0}; pl[p] = u32x4{0, 0, 0, 0}; qa[p] = u32x4{0, 0, 0, 0}; qb[p] = u32x4{0, 0, 0, 0};
                if (w0 + r2 + p < a.B) {
                    ph[p] = *reinterpret_cast<const u32x4*>(pS + voffP + p * ROWB);
                    pl[p] = *reinterpret_cast<const u32x4*>(pS + voffP + p * ROWB + 256);
                    qa[p] = load_qa(p);
                    qb[p] = load_qb(p);
                }
            }
            mw = 0xffffu;
            if (p_masked && w0 + r2 < a.B) mw = *reinterpret_cast<const unsigned short*>(mS + voffM);     // (the 16-window tile of row r2 exists)
        }
        if (++nk == nit) { nk = 0; ++nc; }
    };
    int sk = 0;                    // item of the step being staged
    auto stage_to_lds = [&]() {
        const int ks = sk;
        if (++sk == nit) sk = 0;
        u32x4 mk[2];
        if (p_masked) {
#pragma unroll
            for (int p = 0; p < 2; ++p) mk[p] = mlut[(mw >> (8 * p)) & 0xffu];
        }
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            u32x4 h = ph[p], l = pl[p];
            if (p_masked) { h &= mk[p]; l &= mk[p]; }                   // dH = dX . relu bits
            *reinterpret_cast<u32x4*>(&Ph[ldsw[p]]) = h;
            *reinterpret_cast<u32x4*>(&Pl[ldsw[p]]) = l;
            u32x4 qh = qa[p], ql = qb[p];
            if (!q_act) {      // drop pad columns, symmetry sign mask, fp32 -> hi / lo
                const u32x4 fa = (qa[p] & qfix[c][0]) ^ qsign_t[ks][c][0], fb = (qb[p] & qfix[c][1]) ^ qsign_t[ks][c][1];
                split_oct(__builtin_bit_cast(f32x4, fa), __builtin_bit_cast(f32x4, fb), qh, ql);
            }
            *reinterpret_cast<u32x4*>(&Qh[ldsw[p]]) = qh;
            *reinterpret_cast<u32x4*>(&Ql[ldsw[p]]) = ql;
            if (bias_flag) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    bsum[2 * e] += __builtin_bit_cast(float, h[e] << 16) + __builtin_bit_cast(float, l[e] << 16);
                    bsum[2 * e + 1] += __builtin_bit_cast(float, h[e] & 0xffff0000u) + __builtin_bit_cast(float, l[e] & 0xffff0000u);
                }
            }
        }
    };
    if (total > 0) fetch();
    for (int s = 0; s < total; ++s) {
        __syncthreads();      // the previous MFMA phase of every wave is done reading the tiles
        stage_to_lds();
        __syncthreads();
        if (s + 1 < total) fetch();
#pragma unroll
        for (int ks = 0; ks < KW / 16; ++ks) {
            bf16x8 afh[2], afl[2], bqh[2], bql[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                afh[i] = tr_frag(Ph, ks * 16, wr * 64 + i * 32, lane);
                afl[i] = tr_frag(Pl, ks * 16, wr * 64 + i * 32, lane);
                bqh[i] = tr_frag(Qh, ks * 16, wc * 64 + i * 32, lane);
                bql[i] = tr_frag(Ql, ks * 16, wc * 64 + i * 32, lane);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afh[i], bqh[j], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afh[i], bql[j], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afl[i], bqh[j], acc[i][j], 0, 0, 0);
                }
        }
    }
    float* slab = a.slabs + (size_t)(part * a.n_lanes + ln) * SLAB_FLOATS;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int o = wr * 64 + i * 32 + (q & 3) + ((q >> 2) << 3) + ((lane >> 5) << 2), k = wc * 64 + j * 32 + (lane & 31);
                slab[o * H + k] = acc[i][j][q];
            }
    if (bias_flag) {
        float* red = reinterpret_cast<float*>(Ph);   // 16 x 128 floats = 8 KB <= one tile
        __syncthreads();
#pragma unroll
        for (int e = 0; e < 8; ++e) red[(tid >> 4) * H + c * 8 + e] = bsum[e];
        __syncthreads();
        if (tid < H) {
            float s2 = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) s2 += red[r * H + tid];
            slab[H * H + tid] = s2;
        }
    }
}

// ------------------------------------------------------------------------------------------------------
// launch sequences (mirrors of forward_impl / backward_impl in mshgnn.hip)
// ------------------------------------------------------------------------------------------------------
int x3_launch_prep(const PrepArgs& a, hipStream_t st) {
    const int64_t total = (int64_t)a.n_packs * (H * H / 8) + (int64_t)a.n_biases * H;
    if (prep_use_tiled(a.n_packs)) hipLaunchKernelGGL((k_prep_tiled<__bf16, true>), dim3(prep_tiled_grid(a.n_packs, a.n_biases)), dim3(256), 0, st, a);
    else hipLaunchKernelGGL(k_prep_x3, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, a);
    return MSHGNN_OK;
}

static int x3_lds_stack(const HostPlan& hp) { return 2 * hp.fs_blk * P16::BLK; }

// the kernels over this plan's compile-time program, if it has one (built in, or attached): dynamic-LDS attribute, name; else use_spec = false
static int x3_set_spec_attrs(mshgnn_plan* p) {
    int rc;
    const int flds = x3_lds_stack(p->hp);
    const char* nm = nullptr;
    if (StackKernelX3 k = x3_step_spec_kernel(p->hp, &nm)) {
        if ((rc = set_lds_attr(k, flds)) || (rc = set_lds_attr(x3_spec_kernel(p->hp, 1), flds)) || (rc = set_lds_attr(x3_spec_kernel(p->hp, 2), flds))) return rc;
        p->spec_name_buf = strncmp(nm, "spec::", 6) == 0 ? nm + 6 : nm;
        p->spec_name = p->spec_name_buf.c_str();
    }
    else p->use_spec = false;
    return MSHGNN_OK;
}
// mshgnn_plan_attach_program on a split plan: `selector` is the mshgnn_jit_program_x3 of a program compiled for this plan
int x3_attach_program(mshgnn_plan* p, void* selector) {
    void* const prev = p->hp.jit_prog;
    p->hp.jit_prog = selector;
    const char* nm = nullptr;
    if (!reinterpret_cast<SpecSelectorX3>(selector)(p->hp, 0, &nm) || !nm) { p->hp.jit_prog = prev; return set_err(MSHGNN_EINVAL, "the program's tables are not this plan's"); }
    p->use_spec = true;
    const int rc = x3_set_spec_attrs(p);
    if (rc || !p->use_spec) { p->hp.jit_prog = prev; p->use_spec = false; return rc ? rc : set_err(MSHGNN_EINVAL, "no kernel of the attached program could be used"); }
    return MSHGNN_OK;
}
int x3_set_attrs(mshgnn_plan* p) {
    int rc;
    const int flds = x3_lds_stack(p->hp);
    { const char* esp = getenv("MSHGNN_SPEC"); p->use_spec = !(esp && atoi(esp) == 0); }
    if (p->use_spec && (rc = x3_set_spec_attrs(p))) return rc;
    if ((rc = set_lds_attr(k_stack_fwd_x3<false>, flds)) || (rc = set_lds_attr(k_stack_fwd_x3<true>, flds)) || (rc = set_lds_attr(k_stack_bwd_x3, flds)) ||
        (rc = set_lds_attr(k_stack_step_x3<false>, flds)) || (rc = set_lds_attr(k_stack_step_x3<true>, flds)) ||
        (rc = set_lds_attr(k_enc_x3<true>, 8 * P16::BLK)) || (rc = set_lds_attr(k_enc_x3<false>, 8 * P16::BLK)) ||
        (rc = set_lds_attr(k_enc_x3<true, true>, 8 * P16::BLK)) || (rc = set_lds_attr(k_enc_x3<true, false, 8>, 8 * P16::BLK)) || (rc = set_lds_attr(k_enc_x3<true, false, 4>, 8 * P16::BLK))) return rc;
    return MSHGNN_OK;
}

static void x3_stack_args(const mshgnn_plan* p, const mshgnn_ws_layout& lay, char* ws, int B, StackArgs& a) {
    const HostPlan& hp = p->hp;
    a.ws = ws;
    for (int l = 0; l <= hp.L; ++l) { a.x_off[l] = lay.x[l]; a.dx_off[l] = lay.dx[l]; }
    for (int l = 0; l < hp.L; ++l) { a.mask_off[l] = lay.mask[l]; a.hb_off[l] = lay.hb[l]; a.t1_off[l] = lay.t1[l]; a.dh_off[l] = lay.dh[l]; a.du_off[l] = lay.du[l]; }
    a.wpack = ws + lay.wpack; a.bias = reinterpret_cast<const float*>(ws + lay.bias); a.tables = p->d_tables;
    a.B = B; a.NN = hp.NN; a.L = hp.L;
    a.lo_blk = hp.lo_blk; a.n_img = hp.n_img; a.scr0 = hp.x3_alias ? hp.NN - hp.n_mlp : hp.NN;
    a.stash_nt = stash_nt_for(B, stash_rows_of(hp), H * 2);      // (counted per row, not per byte: stash_nt_for)
}

int x3_forward(const mshgnn_plan* p, const void* const* x, const int64_t* x_pitch, const float* params, float* out, char* ws, int64_t batch,
               int training, hipStream_t st, const float* y_fused, const SeriesSrc* series, bool* stack_step_done, const int32_t* labels_fused) {
    const HostPlan& hp = p->hp;
    const mshgnn_desc& d = hp.d;
    mshgnn_ws_layout lay; layout_workspace(hp, batch, training, &lay);
    const int B = (int)batch;
    // 1. hi / lo weight images; few packs: only the encoder's packs + biases here, the layer packs under the encoder's tail (as forward_impl of mshgnn.hip)
    PrepArgs pa{params, ws + lay.wpack, reinterpret_cast<float*>(ws + lay.bias), p->d_packs, p->d_biases, hp.n_img, (int)hp.biases.size()};
    int enc_pack0 = hp.n_img;
    for (int t = 0; t < hp.NT; ++t) if (hp.pack_enc_base[t] >= 0) enc_pack0 = std::min(enc_pack0, hp.pack_enc_base[t]);
    static const bool embed_off = TUNE_ENV("MSHGNN_PREP_EMBED") && atoi(TUNE_ENV("MSHGNN_PREP_EMBED")) == 0;
    const bool embed = !prep_use_tiled(pa.n_packs) && enc_pack0 > 0 && !embed_off && !series;
    {
        PrepArgs a = pa;
        if (embed) { a.pack0 = enc_pack0; a.pack_n = pa.n_packs - enc_pack0; }
        const int64_t total = (int64_t)(embed ? a.pack_n : a.n_packs) * (H * H / 8) + (int64_t)hp.biases.size() * H;
        ProfScope ps(p, hp.ks_prep, st);
        if (prep_use_tiled(a.n_packs)) hipLaunchKernelGGL((k_prep_tiled<__bf16, true>), dim3(prep_tiled_grid(a.n_packs, a.n_biases)), dim3(256), 0, st, a);
        else hipLaunchKernelGGL(k_prep_x3, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, a);
    }
    const WideSrc* wide = series ? nullptr : g_wide_src;      // (mshgnn_*_src: the caller's fp64 / fp32 rows; x = the fp32 rows to materialise)
    {   // 2. encoder (fp32 inputs)
        EncArgs a{};
        a.n_types = hp.NT; a.B = B; a.NN = hp.NN; a.tiles = (B + 63) / 64;
        a.wg_prefix[0] = 0;
        for (int t = 0; t < hp.NT; ++t) {
            a.x[t] = x[t]; a.pitch[t] = x_pitch ? x_pitch[t] : d.type_width[t];
            if (a.pitch[t] < d.type_width[t]) return set_err(MSHGNN_EINVAL, "x_pitch smaller than the feature width");
            a.vb[t] = vec_bytes(x[t], a.pitch[t], 4);
            if (t == 0) a.aligned = 1;
            if (a.vb[t] != 16 || a.pitch[t] % 4) a.aligned = 0;
            a.width[t] = d.type_width[t]; a.tbase[t] = hp.type_base[t]; a.nkc[t] = hp.enc_nkc[t];
            a.pack0[t] = hp.pack_enc_base[t]; a.bias_idx[t] = hp.bias_enc[t]; a.sign_off[t] = hp.sign_off[t];
            // the launch's nodes of this type (EncArgs.node_list): those whose X_0 can reach the output; with window rows to materialise every node
            a.node_off[t] = t == 0 ? 0 : a.node_off[t - 1] + a.nodes[t - 1];
            a.nodes[t] = 0;
            for (int i = 0; i < d.type_nodes[t]; ++i) {
                const bool need = hp.need_n[0][hp.type_base[t] + i];
                if (need || (series != nullptr && x != nullptr)) a.node_list[a.node_off[t] + a.nodes[t]++] = (unsigned char)i;
                if (!need) a.skip_mask |= 1ull << (hp.type_base[t] + i);
            }
            a.wg_prefix[t + 1] = a.wg_prefix[t] + a.nodes[t] * a.tiles;
        }
        a.tbase[hp.NT] = hp.NN;
        a.wpack = ws + lay.wpack; a.bias = reinterpret_cast<const float*>(ws + lay.bias); a.signs = p->d_signs; a.x0 = ws + lay.x[0];
        a.mask0 = (training && lay.dd[0]) ? reinterpret_cast<uint8_t*>(ws + lay.dd[0]) : nullptr;
        unsigned enc_grid = (unsigned)a.wg_prefix[hp.NT];
        if (embed) {
            PrepArgs lp = pa; lp.pack0 = 0; lp.pack_n = enc_pack0;
            if (a.aligned) { a.prep = lp; a.prep_vecs = enc_pack0 * (H * H / 8); enc_grid += (unsigned)((a.prep_vecs + 255) / 256); }
            else {
                ProfScope ps(p, hp.ks_prep, st);
                hipLaunchKernelGGL(k_prep_x3, dim3((unsigned)(((int64_t)enc_pack0 * (H * H / 8) + 255) / 256)), dim3(256), 0, st, lp);
            }
        }
        ProfScope ps(p, hp.ks_enc, st);
        if (series) {      // inputs gathered from the sequence's series; x = the window buffers the rows are materialised into
            if (!a.aligned) return set_err(MSHGNN_EINVAL, "the series gather writes 16-byte-aligned window buffers whose pitch is a multiple of 4");
            enc_grid += (unsigned)((series->lab.B + 255) / 256);      // the label workgroups
            hipLaunchKernelGGL((k_enc_x3<true, true>), dim3(enc_grid), dim3(256), 8 * P16::BLK, st, a, hp.n_img, *series, WideSrc{});
        }
        else if (wide) {      // the caller's fp64 / fp32 rows: converted by the encoder, fp32 rows written to x on the side
            if (!a.aligned) return set_err(MSHGNN_EINVAL, "wide source rows: the fp32 rows need 16-byte alignment and a pitch that is a multiple of 4");
            if (wide->bytes == 8) hipLaunchKernelGGL((k_enc_x3<true, false, 8>), dim3(enc_grid), dim3(256), 8 * P16::BLK, st, a, hp.n_img, SeriesSrc{}, *wide);
            else hipLaunchKernelGGL((k_enc_x3<true, false, 4>), dim3(enc_grid), dim3(256), 8 * P16::BLK, st, a, hp.n_img, SeriesSrc{}, *wide);
        }
        else if (a.aligned) hipLaunchKernelGGL(k_enc_x3<true>, dim3(enc_grid), dim3(256), 8 * P16::BLK, st, a, hp.n_img, SeriesSrc{}, WideSrc{});
        else hipLaunchKernelGGL(k_enc_x3<false>, dim3(enc_grid), dim3(256), 8 * P16::BLK, st, a, hp.n_img, SeriesSrc{}, WideSrc{});
    }
    {   // 3. all layers + decoder (+ MSE and decoder backward when y_fused)
        StackArgs a{};
        x3_stack_args(p, lay, ws, B, a);
        a.tile_in = ws + lay.x[0]; a.training = training;
        for (int l = 0; l < hp.L; ++l) a.prog_off[l] = hp.fs_fwd_off[l];
        a.params = params; a.out_mask = p->d_out_mask; a.out = out; a.off_dec_w = d.off_dec_w; a.off_dec_b = d.off_dec_b;
        a.node0 = hp.type_base[d.out_type]; a.n_out = d.type_nodes[d.out_type]; a.dout = d.out_channels;
        if (y_fused) {
            a.y = y_fused; a.dec_slabs = reinterpret_cast<float*>(ws + lay.dec_slabs);
            a.inv_n = 1.0f / (float)(loss_windows(B) * a.n_out * a.dout);
        } else if (labels_fused) {      // mshgnn_step_ce: cross entropy over the per-foot logit pairs, mean over B * n_out rows (the tail is the bf16 plan's)
            a.labels = labels_fused; a.dec_slabs = reinterpret_cast<float*>(ws + lay.dec_slabs);
            a.inv_n = 1.0f / (float)(loss_windows(B) * a.n_out);
        }
        const int tiles = (B + P16::ROWS - 1) / P16::ROWS;
        a.stamps = stamp_ptr("MSHGNN_STAMPS");
        // one-call step: the backward sweep in the same launch.  The tail's reduction scratch (one decoder slab per wave) must not touch the out-type nodes' blocks
        // of either plane, which receive dX_L for the backward sweep: it sits in the hi plane's blocks in front of them, or (models whose out type comes first: the
        // centroidal-momentum ones) in the hi plane's blocks behind them -- as on the bf16 plan (forward_impl, mshgnn.hip)
        const size_t red_need = (size_t)(LAYER_THREADS / 64) * DEC_SLAB_FLOATS * sizeof(float);
        const size_t red_back = (size_t)(a.node0 + a.n_out) * P16::BLK, plane = (size_t)hp.fs_blk * P16::BLK;
        const bool red_front_ok = (size_t)a.node0 * P16::BLK >= red_need, red_back_ok = red_back + red_need <= plane;
        const bool step = stack_step_done && (y_fused || labels_fused) && p->use_step && (red_front_ok || red_back_ok);
        if (step && !red_front_ok) a.red_off = (int)red_back;
        ProfScope ps(p, step ? hp.ks_stack_step : hp.ks_stack_fwd, st);
        if (step) {
            a.mask0_off = lay.dd[0];
            for (int l = 0; l < hp.L; ++l) a.prog_off_b[l] = hp.fs_bwd_off[l];
            StackKernelX3 ks = p->use_spec ? x3_step_spec_kernel(hp) : nullptr;      // (predicated stores: ragged batches run it too)
            if (ks) hipLaunchKernelGGL(ks, dim3(tiles), dim3(LAYER_THREADS), x3_lds_stack(hp), st, a);
            else if (hp.x3_alias) hipLaunchKernelGGL(k_stack_step_x3<true>, dim3(tiles), dim3(LAYER_THREADS), x3_lds_stack(hp), st, a);
            else hipLaunchKernelGGL(k_stack_step_x3<false>, dim3(tiles), dim3(LAYER_THREADS), x3_lds_stack(hp), st, a);
            *stack_step_done = true;
        } else
        if (StackKernelX3 kf = p->use_spec ? x3_spec_kernel(hp, 1) : nullptr) hipLaunchKernelGGL(kf, dim3(tiles), dim3(LAYER_THREADS), x3_lds_stack(hp), st, a);
        else if (hp.x3_alias) hipLaunchKernelGGL(k_stack_fwd_x3<true>, dim3(tiles), dim3(LAYER_THREADS), x3_lds_stack(hp), st, a);
        else hipLaunchKernelGGL(k_stack_fwd_x3<false>, dim3(tiles), dim3(LAYER_THREADS), x3_lds_stack(hp), st, a);
    }
    HIPCHK(hipGetLastError());
    return MSHGNN_OK;
}

int x3_backward(const mshgnn_plan* p, const void* const* x, const int64_t* x_pitch, const float* params, const float* gout, float* gparams, char* ws,
                int64_t batch, hipStream_t st, const float* out, const float* y, float* loss, const int32_t* labels, bool dec_done, int gw_phase, bool stack_done) {
    const HostPlan& hp = p->hp;
    const mshgnn_desc& d = hp.d;
    mshgnn_ws_layout lay; layout_workspace(hp, batch, 1, &lay);
    const int B = (int)batch;
    if (!dec_done && gw_phase != 1) {
        DecArgs a{};
        a.xl = ws + lay.x[hp.L]; a.dxl = ws + lay.dx[hp.L]; a.params = params; a.out_mask = p->d_out_mask; a.gout = gout;
        a.slabs = reinterpret_cast<float*>(ws + lay.dec_slabs); a.off_w = d.off_dec_w; a.off_b = d.off_dec_b;
        a.B = B; a.NN = hp.NN; a.node0 = hp.type_base[d.out_type]; a.n_out = d.type_nodes[d.out_type]; a.dout = d.out_channels; a.slab0 = 0;
        if (y) { a.y = y; a.out = const_cast<float*>(out); a.loss = loss; a.inv_n = 1.0f / (float)(loss_windows(B) * a.n_out * a.dout); }
        if (labels) { a.labels = labels; a.out = const_cast<float*>(out); a.loss = loss; a.inv_n = 1.0f / (float)(loss_windows(B) * a.n_out); }
        ProfScope ps(p, hp.ks_dec_bwd, st);
        hipLaunchKernelGGL(k_dec_bwd_x3, dim3(NWG_DEC), dim3(256), 0, st, a);
    }
    if (gw_phase != 1 && !stack_done) {
        StackArgs a{};
        x3_stack_args(p, lay, ws, B, a);
        a.tile_in = ws + lay.dx[hp.L]; a.training = 1; a.mask0_off = lay.dd[0];
        for (int l = 0; l < hp.L; ++l) a.prog_off[l] = hp.fs_bwd_off[l];
        const int tiles = (B + P16::ROWS - 1) / P16::ROWS;
        ProfScope ps(p, hp.ks_stack_bwd, st);
        if (StackKernelX3 kb = p->use_spec ? x3_spec_kernel(hp, 2) : nullptr) hipLaunchKernelGGL(kb, dim3(tiles), dim3(LAYER_THREADS), x3_lds_stack(hp), st, a);
        else hipLaunchKernelGGL(k_stack_bwd_x3, dim3(tiles), dim3(LAYER_THREADS), x3_lds_stack(hp), st, a);
    }
    const int gw_parts = gw_parts_for(hp.n_parts, hp.n_lanes, hp.gw_ipl, B, 32, p->n_cu);      // window parts of this batch's weight-gradient launch (32-window steps)
    {
        GradwArgs a{};
        a.ws = ws;
        for (int l = 0; l <= hp.L; ++l) { a.buf_off[BUF_X + l] = lay.x[l]; a.buf_off[BUF_DX + l] = lay.dx[l]; }
        for (int l = 0; l < hp.L; ++l) a.buf_off[BUF_MASK + l] = lay.mask[l];
        for (int l = 0; l < hp.L; ++l) { a.buf_off[BUF_DH + l] = lay.dh[l]; a.buf_off[BUF_HB + l] = lay.hb[l]; a.buf_off[BUF_T1 + l] = lay.t1[l]; a.buf_off[BUF_DU + l] = lay.du[l]; }
        for (int t = 0; t < hp.NT; ++t) {
            a.x[t] = x[t]; a.pitch[t] = x_pitch ? x_pitch[t] : d.type_width[t]; a.nodes[t] = d.type_nodes[t];
            a.vb[t] = vec_bytes(x[t], a.pitch[t], 4);
            if (t == 0) a.aligned = 1;
            if (a.vb[t] != 16 || a.pitch[t] % 4) a.aligned = 0;
        }
        a.items = p->d_tables + hp.item_off; a.lanes = p->d_tables + hp.lane_off; a.lane_order = p->d_tables + hp.lane_order_off; a.n_pad = hp.n_lanes_pad;
        if (gw_phase >= 0) { a.lane_order = p->d_tables + hp.order_ph_off[gw_phase]; a.n_pad = hp.npad_ph[gw_phase]; }
        a.signs = p->d_signs; a.slabs = reinterpret_cast<float*>(ws + lay.slabs); a.B = B; a.n_lanes = hp.n_lanes; a.n_parts = gw_parts;
        ProfScope ps(p, hp.ks_gradw, st);
        [[maybe_unused]] static const bool gw_general = TUNE_ENV("MSHGNN_GRADW") && std::string(TUNE_ENV("MSHGNN_GRADW")) == "general";   // read once: the general kernel also where the lean one applies (A/B runs)
        if (a.n_pad > 0) {
#ifdef MSHGNN_TUNING
            if (gw_general && hp.gw_ipl == 1) hipLaunchKernelGGL(k_gradw_x3, dim3(a.n_pad * gw_parts), dim3(256), 0, st, a);
            else
#endif
            if (a.aligned) hipLaunchKernelGGL(k_gradw_x3_lean<true>, dim3(a.n_pad * gw_parts), dim3(256), 0, st, a);
            else hipLaunchKernelGGL(k_gradw_x3_lean<false>, dim3(a.n_pad * gw_parts), dim3(256), 0, st, a);
        }
    }
    return run_finalize(p, lay, ws, gparams, B, (y || labels) ? loss : nullptr, labels != nullptr, dec_done, gw_phase, st, gw_parts);
}
#endif      // MSHGNN_SPEC_SHARD == 0
