// libmshgnn: MI355X (gfx950 / CDNA4) MS-HGNN message-passing engine -- HIP kernels + C-ABI (include/mshgnn.h).
//
// Design (DESIGN.md has the long form):
//   * every window graph shares one tiny topology  => a minibatch is a dense [B][NN][128] tensor per layer;
//   * a workgroup owns a tile of ROWS windows (16 for fp32, 32 for bf16): all NN node blocks (ROWS x 128, 8 KB
//     each, XOR-swizzled) are staged into LDS once per layer, and every per-relation linear of the layer is an
//     MFMA block-GEMM  acc[dst node] += X_lds[src node] . W  with the weight fragment held in registers and reused
//     for every destination node of the relation (root weights pre-summed per destination type);
//   * wave w of the workgroup owns output columns [32w, 32w+32) of every node; MFMA operands use a K-permutation
//     (lane group g owns a contiguous K range) so each lane's A and B data are 128 contiguous bytes;
//   * bias, ReLU, base_transform MLP, residual, ReLU-bit stash fused in the epilogue; backward reuses the same
//     engine on the transposed graph with transposed weight images; weight gradients are one split-K MFMA launch
//     over all layers with deterministic slab reduction (no float atomics).
//
// No CPU fallback exists: every entry point launches HIP kernels or fails loudly.
#include "mshgnn_device.hpp"
#ifndef MSHGNN_SPEC_SHARD
#define MSHGNN_SPEC_SHARD 0      // 1..7: this source compiled as one of the translation units that only instantiate the compile-time programs' kernels (csrc/Makefile; see spec_shard1 below)
#endif
#if MSHGNN_SPEC_SHARD == 0
extern "C" const char* mshgnn_last_error(void) { return g_err.c_str(); }
extern "C" const char* mshgnn_version(void) { return "mshgnn 0.5 (gfx950)"; }
extern "C" int mshgnn_abi_version(void) { return MSHGNN_ABI_VERSION; }
extern "C" size_t mshgnn_struct_size(int which) {
    switch (which) {
        case 0: return sizeof(mshgnn_desc);
        case 1: return sizeof(mshgnn_info);
        case 2: return sizeof(mshgnn_ws_layout);
        case 3: return sizeof(mshgnn_window_desc);
        case 4: return sizeof(mshgnn_kernel_stat);
        default: return 0;
    }
}
#endif

// ------------------------------------------------------------------------------------------------------
// k_prep: pack weights into MFMA B-fragment images (root-sum, transpose, dtype) and sum biases
// ------------------------------------------------------------------------------------------------------

// Few packs (A1-C2 at L = 3: ~100): one thread per output vector, gathers straight from global memory -- more workgroups than packs, two memory
// round trips.  Many packs (K4 at L = 8, the generic-width engine): k_prep_tiled (mshgnn_device.hpp), one workgroup per half pack through LDS
// (measured: A1-C2 L=3 10.7 vs 17.1 us, K4 L=8 18.5 vs 16.2, synthetic 32-limb h=512 97 vs 52).
// output vector `idx` of this launch's pack range, or (idx past the packs) one bias sum
template <typename T> __device__ __forceinline__ void prep_one(const PrepArgs& a, int idx, bool with_bias) {
    constexpr int EPC = Prec<T>::EPC, NBV = Prec<T>::NBV;
    const int vec_per_pack = H * H / EPC;
    const int npk = a.pack_n < 0 ? a.n_packs : a.pack_n;
    const int total = npk * vec_per_pack;
    if (idx < total) {
        const int pack = a.pack0 + idx / vec_per_pack, r = idx % vec_per_pack;
        const int gid = pack * vec_per_pack + r;
        const int lane = r % 64, v = (r / 64) % NBV, wv = r / (64 * NBV);
        const PackDesc pd = a.packs[pack];
        T out[EPC];
        float g[8][EPC];     // up to 8 source matrices (root-sum), every gather issued before the first add
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int x = 0; x < EPC; ++x) {
                int k, col;
                // MFMA row i = lane & 15 of the wave's 16-row block fb carries output feature 8 (i / 4) + 4 fb + i % 4
                const int i16 = lane & 15;
                if constexpr (sizeof(T) == 4) { k = 32 * (lane >> 4) + 4 * (v & 7) + x; col = wv * 32 + 8 * (i16 >> 2) + 4 * (v >> 3) + (i16 & 3); }
                else { k = 32 * (lane >> 4) + 8 * (v & 3) + x; col = wv * 32 + 8 * (i16 >> 2) + 4 * (v >> 2) + (i16 & 3); }
                g[i][x] = 0.f;
                if (i < pd.n_src) {
                    if (pd.orient == 0) { if (k < pd.ncols) g[i][x] = a.params[pd.src[i] + (int64_t)col * pd.ld + pd.col0 + k]; }
                    else g[i][x] = a.params[pd.src[i] + (int64_t)k * pd.ld + col];
                }
            }
#pragma unroll
        for (int x = 0; x < EPC; ++x) {
            float sum = 0.f;
#pragma unroll
            for (int i = 0; i < 8; ++i) sum += g[i][x];     // fixed order: same value every step
            out[x] = from_f32<T>(sum);
        }
        T* dst = reinterpret_cast<T*>(a.wpack) + (size_t)gid * EPC;
#pragma unroll
        for (int x = 0; x < EPC; ++x) dst[x] = out[x];
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
template <typename T> __global__ void k_prep(PrepArgs a) { prep_one<T>(a, blockIdx.x * blockDim.x + threadIdx.x, true); }

// ------------------------------------------------------------------------------------------------------
// k_enc_fwd: X_0[node] = relu((mask . x) W_enc^T + b)      (hgnn_c2.py:143-147)
// one workgroup = MB row blocks (MB*ROWS windows) of ONE node; K streamed in chunks of 128 through LDS
// ------------------------------------------------------------------------------------------------------

// ALIGNED: every input row starts 16-byte aligned with a pitch of whole 16-byte chunks (the engine's own input layout): raw
// 16-byte loads only -- the general element-wise path is compiled out of this instantiation (a third of the kernel's code)
// SERIES (bf16, ALIGNED): the rows come out of the sequence's raw series (SeriesSrc) and are ALSO written to a.x as materialised windows (the
// weight-gradient kernel reads them later): window assembly fused into the encoder -- one unaligned 16-byte load per chunk (two where the chunk
// straddles two runs), no separate gather pass over 118 MB.
__device__ __forceinline__ u32x4 splice8(u32x4 A, u32x4 B, int n0) {      // bf16 elements A[0 .. n0) ++ B[0 .. 8 - n0), 0 < n0 < 8
    const unsigned __int128 a = ((unsigned __int128)(((unsigned long long)A[3] << 32) | A[2]) << 64) | (((unsigned long long)A[1] << 32) | A[0]);
    const unsigned __int128 b = ((unsigned __int128)(((unsigned long long)B[3] << 32) | B[2]) << 64) | (((unsigned long long)B[1] << 32) | B[0]);
    const int sh = 16 * n0;
    const unsigned __int128 r = (a & ((((unsigned __int128)1) << sh) - 1)) | (b << sh);
    return u32x4{(unsigned)r, (unsigned)(r >> 32), (unsigned)(r >> 64), (unsigned)(r >> 96)};
}
// SRC (bf16, ALIGNED; 8 / 4): the rows come from the caller's own fp64 / fp32 tensors at their dense pitch (WideSrc), are converted in registers and ALSO
// written to a.x as plan-dtype rows for the weight-gradient kernel -- the cast + re-pitch pass fused into the encoder (mshgnn_*_src entry points).
template <typename T, bool ALIGNED, bool SERIES = false, int SRC = 0> __global__ __launch_bounds__(256) void k_enc_fwd(EncArgs a, SeriesSrc ser, WideSrc wsrc) {
    using P = Prec<T>;
    static_assert(!SERIES || (sizeof(T) == 2 && ALIGNED), "the series gather is a bf16 path");
    static_assert(SRC == 0 || (sizeof(T) == 2 && ALIGNED && !SERIES), "wide source rows: bf16 plan, aligned destination rows");
    constexpr int MB = P::ENC_MB;                       // row blocks (of 16 windows) per workgroup
    constexpr int VPB = P::ROWS * P::CPR, NIT = VPB / 256 > 0 ? VPB / 256 : 1, BPP = 256 / VPB > 0 ? 256 / VPB : 1;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    // SERIES: the FIRST workgroups of the launch compute the batch's window labels (SeriesSrc.lab) -- chains of dependent round trips that finish
    // under the encoder's body (as the last workgroups they stretched its tail instead: measured)
    const int lab_blocks = SERIES ? (int)((ser.lab.B + 255) / 256) : 0;
    if constexpr (SERIES) {
        if ((int)blockIdx.x < lab_blocks) {
            const int64_t b = (int64_t)blockIdx.x * 256 + tid;
            if (b < ser.lab.B) window_labels_one(ser.lab, b);
            return;
        }
    }
    const int bid = (int)blockIdx.x - lab_blocks;
    if (bid >= a.wg_prefix[a.n_types]) {      // a workgroup of the embedded layer-pack prep (bf16 plan; see EncArgs.prep)
        if constexpr (sizeof(T) == 2 && ALIGNED && !SERIES) prep_one<T>(a.prep, (bid - a.wg_prefix[a.n_types]) * 256 + tid, false);
        return;
    }
    int t = 0;
    while (t + 1 < a.n_types && bid >= a.wg_prefix[t + 1]) ++t;
    const int local = bid - a.wg_prefix[t];
    const int node = a.node_list[a.node_off[t] + (ENC_ORDER ? local % a.nodes[t] : local / a.tiles)], tile = ENC_ORDER ? local / a.nodes[t] : local % a.tiles;
    const bool skip = SERIES && ((a.skip_mask >> (a.tbase[t] + node)) & 1ull) != 0;      // window rows only: nobody reads this node's X_0 (uniform)
    const int w0 = tile * MB * P::ROWS;
    const T* x = reinterpret_cast<const T*>(a.x[t]);
    const int64_t pitch = a.pitch[t];
    const int F = a.width[t], nt = a.tbase[t + 1] - a.tbase[t], nkc = a.nkc[t], vb = a.vb[t];      // nt: nodes of the type in the input rows (not the launch's list)
    const uint8_t* sg = a.signs + a.sign_off[t] + (size_t)node * nkc * H;
    const T* wpack = reinterpret_cast<const T*>(a.wpack);
    const float* bias = a.bias + (size_t)max(a.bias_idx[t], 0) * H;      // (a type whose X_0 nobody reads has no packs and no bias row: skip)

    typename P::Acc acc[MB];
#pragma unroll
    for (int m = 0; m < MB; ++m) acc_init_bias<T>(acc[m], bias, wv, lane);
    // staging map: fp32 block = 512 chunks -> 2 per thread; bf16 block = 256 chunks -> 1 per thread
    const int c = tid % P::CPR, r0 = (tid % VPB) / P::CPR, sub = tid / VPB;   // sub is 0 for fp32
    typename P::BFrag bf;
    typename P::AFrag af;
    const AOff<T> ao(lane);      // fragment offsets once per kernel (the generic load_afrag rebuilds them per call: 14 VALU instructions)
    u32x4 v[MB / BPP][NIT];
    u32x4 rawv[(SERIES || SRC) ? MB / BPP : 1][NIT];      // SERIES / SRC: the chunk being multiplied, kept until its window rows have been written
    u32x2 wv8[SRC ? MB / BPP : 1][SRC ? SRC : 1];          // SRC: the chunk's 8 source elements as 8-byte units, untouched until the staging pass
    const bool unit_ok = SRC == 8 || (F & 1) == 0;        // every 8-byte unit of a row is wholly valid or wholly past its end (uniform)
    const int64_t spitch = SRC ? wsrc.pitch[t] : 0;
    // SERIES: first series row of this thread's window rows, the node row's first run
    int srow[SERIES ? MB / BPP : 1];
    __shared__ unsigned long long rp_s[SERIES ? 16 : 1];      // SERIES: the column pointers of this node row's first 16 runs (a chunk takes its pieces from runs j, j + 1)
    int rfirst = 0, rend = 0;
    if constexpr (SERIES) {
#pragma unroll
        for (int mi = 0; mi < MB / BPP; ++mi) srow[mi] = (int)ser.starts[min(w0 + (mi * BPP + sub) * P::ROWS + r0, a.B - 1)];
        // one round trip for the row's run pointers instead of a dependent pointer load in front of every chunk's data loads
        rfirst = ser.rows[2 * (ser.row0[t] + node)]; rend = ser.rows[2 * (ser.row0[t] + node) + 1];
        if (tid < 16) rp_s[tid] = rfirst + tid < rend ? ser.run_ptr[rfirst + tid] : 0ull;
        __syncthreads();
    }
    auto fetch = [&](int kc) {
        const int k0 = kc * H + c * P::EPC;
        const int nvalid = min(P::EPC, F - k0);
        if constexpr (SERIES) {
            // elements [k0, k0 + 8) of the row: n0 of them from run j at time offset off, the rest from run j + 1 at offset 0
            const int j = k0 / ser.T, off = k0 - j * ser.T, n0 = min(P::EPC, ser.T - off);
            const bool second = nvalid > n0;
            // (runs past the 16th -- a node row of more than 16 T-long variables -- come from the global table: a dependent load, rare recipes only)
            auto run_ptr_of = [&](int jj) -> unsigned long long { return jj < 16 ? rp_s[jj] : (rfirst + jj < rend ? ser.run_ptr[rfirst + jj] : 0ull); };
            const unsigned long long pa = nvalid > 0 ? run_ptr_of(j) : 0ull, pb = second ? run_ptr_of(j + 1) : 0ull;
            const u32x4 ones = u32x4{0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};      // the constant-1 run (bf16 1.0)
#pragma unroll
            for (int mi = 0; mi < MB / BPP; ++mi) {
                u32x4 va = nvalid > 0 ? ones : u32x4{0, 0, 0, 0};
                if (pa) va = *reinterpret_cast<const u32x4*>(reinterpret_cast<const T*>(pa) + srow[mi] + off);       // 2-byte aligned: served at full rate
                if (second) {
                    u32x4 vb2 = ones;
                    if (pb) vb2 = *reinterpret_cast<const u32x4*>(reinterpret_cast<const T*>(pb) + srow[mi]);
                    va = splice8(va, vb2, n0);
                }
                v[mi][0] = va;
            }
            return;
        }
        if constexpr (SRC > 0) {
#pragma unroll
            for (int mi = 0; mi < MB / BPP; ++mi) {
                const int w = w0 + (mi * BPP + sub) * P::ROWS + r0;
                const char* row = reinterpret_cast<const char*>(wsrc.p[t]) + ((size_t)min(w, a.B - 1) * nt + node) * spitch * SRC;
                wide_fetch<SRC>(wv8[mi], row, k0, F, unit_ok, w < a.B);
            }
            return;
        }
#pragma unroll
        for (int mi = 0; mi < MB / BPP; ++mi)
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int m = mi * BPP + sub, row = r0 + it * (256 / P::CPR), w = w0 + m * P::ROWS + row;
                if constexpr (ALIGNED) {
                    // unconditional raw 16-byte load (nothing uses it here): rows past the batch re-read the last row, chunks past the
                    // row's end re-read the K chunk's first one -- the staging pass zeroes both
                    const T* src = x + ((size_t)min(w, a.B - 1) * nt + node) * pitch + (nvalid > 0 ? k0 : kc * H);
                    v[mi][it] = ld16<ENC_NT>(src);
                } else {
                    v[mi][it] = u32x4{0, 0, 0, 0};
                    if (w < a.B) v[mi][it] = load_chunk<T>(x + ((size_t)w * nt + node) * pitch + k0, nvalid, vb);
                }
            }
    };
    // (two K chunks in flight per thread were measured: no gain, ENC_DEEP of round 2; the plain loop keeps the kernel at 82 VGPRs = five workgroups per CU)
    fetch(0);
    for (int kc = 0; kc < nkc; ++kc) {
        const u32x4 sx = sign_xor<T>(sg + kc * H + c * P::EPC);   // apply_symmetry: +-1 mask as a sign-bit XOR
        const int nv = F - (kc * H + c * P::EPC);                  // valid elements of this thread's chunk (pad columns dropped)
        __syncthreads();   // previous chunk's MFMAs are done reading LDS
#pragma unroll
        for (int mi = 0; mi < MB / BPP; ++mi)
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                u32x4 raw;
                if constexpr (SRC > 0) {      // fp64 / fp32 -> (fp32 ->) bf16, round to nearest even twice as torch's .to(bfloat16) does; elements past the row: zero
                    f32x4 lo4, hi4;
                    wide_to_f32<SRC>(wv8[mi], nv, lo4, hi4);
                    raw = pack_oct(lo4, hi4);
                } else
                raw = kc + 1 == nkc ? chunk_keep_first<T>(v[mi][it], nv) : v[mi][it];      // only the last K chunk has pad columns
                *reinterpret_cast<u32x4*>(smem + lds_chunk<T>(mi * BPP + sub, r0 + it * (256 / P::CPR), c)) = raw ^ sx;
                if constexpr (SERIES || SRC > 0) rawv[mi][it] = raw;
            }
        __syncthreads();
        if (!skip) load_bfrag<T>(bf, wpack, a.pack0[t] + kc, wv, lane);   // before the prefetch: vmcnt retires in order
        if (kc + 1 < nkc) fetch(kc + 1);   // the next K chunk streams from HBM under this chunk's MFMAs
        if constexpr (SERIES || SRC > 0) {
            // the materialised window rows of THIS chunk (raw values: the sign mask is applied by whoever reads them) go out BEHIND the next chunk's
            // loads: vmcnt retires in issue order, so a load issued after stores can only be waited for together with them -- with the stores
            // youngest, the next chunk's wait leaves them in flight
#pragma unroll
            for (int mi = 0; mi < MB / BPP; ++mi)
#pragma unroll
                for (int it = 0; it < NIT; ++it) {
                    const int w = w0 + (mi * BPP + sub) * P::ROWS + r0 + it * (256 / P::CPR), k0 = kc * H + c * P::EPC;
                    if (x != nullptr && w < a.B && k0 < (int)pitch) *reinterpret_cast<u32x4*>(const_cast<T*>(x) + ((size_t)w * nt + node) * pitch + k0) = rawv[mi][it];
                }
        }
        if (!skip) {
#pragma unroll
            for (int m = 0; m < MB; ++m) {
                if (w0 + m * P::ROWS < a.B) {   // uniform
                    load_afrag<T>(af, smem, m, ao);
                    mac(acc[m], af, bf);
                }
            }
        }
    }
    if (skip) return;
    T* x0 = reinterpret_cast<T*>(a.x0);
    const int gnode = a.tbase[t] + node;
#pragma unroll
    for (int m = 0; m < MB; ++m) {
        const int w = w0 + m * P::ROWS + c_win(lane);
        if (w0 + m * P::ROWS < a.B) {     // uniform: the 16-window block exists (rows past the batch land in the mask buffer's padding)
            const unsigned bits = relu_with_bits<T>(acc[m]);
            if (a.mask0) a.mask0[relu_tile_base(gnode, a.B, (w0 + m * P::ROWS) >> 4, wv) + lane] = (uint8_t)bits;
        }
        if (w < a.B) store_oct(x0 + act_idx(w, gnode, a.B) + wv * 32 + c_oct(lane), acc[m].c[0], acc[m].c[1]);
    }
}

// ------------------------------------------------------------------------------------------------------
// k_layer_fwd: one HeteroConv layer + activation / base_transform + residual  (hgnn_c2.py:150-166)
// ------------------------------------------------------------------------------------------------------
struct LayerArgs {
    const void* x_in;      // fwd: X_l                    bwd: G_{l+1} = dX_{l+1} (read; written when it is materialised here)
    void* x_out;           // fwd: X_{l+1}                bwd: layer 0 only: dX_0 = relu'(X_0) . (G_1 + D_0)
    unsigned* maskbits;    // relu bits of this layer [NN][4][B]
    void* hb; void* t1;    // base_transform stash of this layer [B][n_mlp][128]
    void* dh; void* du;    // bwd: dH_l [B][NN][128], dU_l [B][n_mlp][128]
    const void* x_act;     // bwd layer 0: X_0 (encoder relu mask)
    const void* wpack; const float* bias; const int* prog;
    int B, NN, n_mlp;
    int dbg;               // ablation switches for timing experiments (MSHGNN_DBG): 1 no stage-in, 2 no MACs, 4 no W loads, 8 no group epilogues,
                           // backward: 16 no stage-1 loads/stores, 32 no base_transform chain, 64 no stage-2 epilogues
};

template <typename T>
__device__ __forceinline__ void store_relu_bits(const typename Prec<T>::Acc& acc, unsigned* maskbits, int NN, int n, int w0, int B, int wn, int lane) {
    unsigned bits = 0;
#pragma unroll
    for (int fb = 0; fb < 2; ++fb)
#pragma unroll
        for (int j = 0; j < 4; ++j) bits |= (acc.c[fb][j] > 0.f ? 1u : 0u) << (4 * fb + j);
    // rows past the batch inside the last tile land in the buffer's padding (it is sized for whole tiles)
    reinterpret_cast<uint8_t*>(maskbits)[relu_byte(n, B, w0 + c_win(lane), wn * 32 + c_oct(lane))] = (uint8_t)bits;
}

// The wave's MAC program of a group lives in two VGPRs (entry i in lane i); entries are fetched with v_readlane.
struct WaveProg {
    int p0, p1, pc;
    __device__ __forceinline__ WaveProg(const int* prog, int lane) : p0(prog[lane]), p1(prog[64 + lane]), pc(0) {}
    __device__ __forceinline__ int next() {
        const int i = pc++;
        return i < 64 ? __builtin_amdgcn_readlane(p0, i) : __builtin_amdgcn_readlane(p1, i - 64);
    }
};

// run every segment of the group for this wave: per segment load the packed weight fragment once, then walk the
// accumulators in static order, each with its run-time list of source blocks
template <typename T>
__device__ __forceinline__ void run_segments(WaveProg& wp, int pc0, typename Prec<T>::Acc (&acc)[Prec<T>::HS], const char* smem,
                                             const T* wpack, int wn, int lane, int dbg) {
    using P = Prec<T>;
    wp.pc = pc0;
    const int nseg = wp.next();
    typename P::BFrag bf;
    typename P::AFrag af;
    for (int s = 0; s < nseg; ++s) {
        const int pack = wp.next();
        if (!ABL(dbg & 4) || s == 0) load_bfrag<T>(bf, wpack, pack, wn, lane);
#pragma unroll
        for (int u = 0; u < P::HS; ++u) {
            const int cnt = wp.next();
            for (int k = 0; k < cnt; ++k) {
                const int blk = wp.next();
                if (!ABL(dbg & 2)) {
                    load_afrag<T>(af, smem, blk, lane);
                    mac(acc[u], af, bf);
                }
            }
        }
    }
}

// one extra block-GEMM of the base_transform chain: acc[u] = bias + LDS[blk(slot)] . W(pack) for this wave's slots
template <typename T>
__device__ __forceinline__ void mlp_gemm(typename Prec<T>::Acc (&acc)[Prec<T>::HS], const char* smem, const int* blks, int ns,
                                         const T* wpack, int pack, const float* bias, int wn, int wh, int lane) {
    using P = Prec<T>;
    typename P::BFrag bf;
    typename P::AFrag af;
    load_bfrag<T>(bf, wpack, pack, wn, lane);
#pragma unroll
    for (int u = 0; u < P::HS; ++u) {
        const int slot = 2 * u + wh;
        if (slot < ns) {
            acc_init_bias<T>(acc[u], bias, wn, lane);
            load_afrag<T>(af, smem, blks[slot], lane);
            mac(acc[u], af, bf);
        }
    }
}

template <typename T> __global__ __launch_bounds__(LAYER_THREADS, Prec<T>::WPS) void k_layer_fwd(LayerArgs a) {
    using P = Prec<T>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wv & 3, wh = wv >> 2;
    const int w0 = blockIdx.x * P::ROWS, B = a.B, NN = a.NN;
    const T* xin = reinterpret_cast<const T*>(a.x_in);
    T* xout = reinterpret_cast<T*>(a.x_out);
    const T* wpack = reinterpret_cast<const T*>(a.wpack);
    const int win = c_win(lane), w = w0 + win;
    const bool w_ok = w < B;

    const int* pg = a.prog;
    const int ngroups = pg[0];
    WaveProg wp(pg + 1 + ngroups * GH_SIZE + wh * WPROG_LEN, lane);   // the whole layer's MAC program of this wave, loaded once
    if (!ABL(a.dbg & 1)) stage_nodes<T>(smem, xin, NN, w0, B, tid);
    __syncthreads();

    typename P::Acc acc[P::HS];
    const int* gh = pg + 1;
    for (int g = 0; g < ngroups; ++g, gh += GH_SIZE) {
        const int kind = gh[GH_KIND], ns = gh[GH_NSLOTS], flags = gh[GH_FLAGS];
        const BiasQ bq = load_bias(a.bias + (size_t)gh[GH_BIAS] * H, wn, lane);
#pragma unroll
        for (int u = 0; u < P::HS; ++u) acc_fill(acc[u], 0.f);
        run_segments<T>(wp, gh[GH_PC0 + wh], acc, smem, wpack, wn, lane, a.dbg);
        if ABL(a.dbg & 8) continue;
#pragma unroll
        for (int u = 0; u < P::HS; ++u) { acc[u].c[0] += bq.b[0]; acc[u].c[1] += bq.b[1]; }
        if (kind == KIND_RELU) {
            const bool lds_epi = (flags & GF_LDS_EPI) != 0;
            // last group: every source block is dead once all waves are past their MACs -> X_new goes into the nodes'
            // own LDS blocks (residual read in place) and is then stored as whole rows
            if (lds_epi) __syncthreads();
#pragma unroll
            for (int u = 0; u < P::HS; ++u) {
                const int slot = 2 * u + wh;
                if (slot < ns) {
                    const int n = gh[GH_NODES + slot];
                    if (flags & GF_STORE_MASK) store_relu_bits<T>(acc[u], a.maskbits, NN, n, w0, B, wn, lane);
                    const int col = wn * 32 + c_oct(lane);
                    f32x4 y0 = relu4(acc[u].c[0]), y1 = relu4(acc[u].c[1]);
                    if (flags & GF_RESIDUAL) {
                        f32x4 r0, r1;
                        lds_load_oct<T>(smem, n, win, col, r0, r1);
                        y0 += r0; y1 += r1;
                    }
                    if (lds_epi) lds_store_oct<T>(smem, n, win, col, y0, y1);
                    else if (w_ok) store_oct(xout + act_idx(w, n, B) + col, y0, y1);
                }
            }
            if (lds_epi) {
                __syncthreads();
                const RowMap<T> m(tid);
                if (w0 + m.row < B)
                    for (int sl = m.sub; sl < ns; sl += RowMap<T>::NPB) {
                        const int n = gh[GH_NODES + sl];
                        const u32x4 v = *reinterpret_cast<const u32x4*>(smem + lds_chunk<T>(n, m.row, m.c));
                        *reinterpret_cast<u32x4*>(xout + act_idx(w0 + m.row, n, B) + m.c * P::EPC) = v;
                    }
            }
        } else {
            // base_transform: Y = W2 relu(W1 H + b1) + b2, X <- Y + X   (hgnn_c2.py:117-121,156,161-166)
            T* hb = reinterpret_cast<T*>(a.hb);
            T* t1 = reinterpret_cast<T*>(a.t1);
            const int* scr = gh + GH_SCR;
            const bool own = scr[0] == gh[GH_NODES];   // scratch aliases the nodes' own blocks (no spare LDS)
            __syncthreads();   // every wave is done reading the group's source blocks
#pragma unroll
            for (int u = 0; u < P::HS; ++u) {
                const int slot = 2 * u + wh;
                if (slot < ns) {
                    const int mi = gh[GH_MLPIDX + slot], col = wn * 32 + c_oct(lane);
                    lds_store_oct<T>(smem, scr[slot], win, col, acc[u].c[0], acc[u].c[1]);
                    if (w_ok) store_oct(hb + act_idx(w, mi, B) + col, acc[u].c[0], acc[u].c[1]);
                }
            }
            __syncthreads();
            mlp_gemm<T>(acc, smem, scr, ns, wpack, gh[GH_W1], a.bias + (size_t)gh[GH_B1] * H, wn, wh, lane);
            __syncthreads();   // all reads of H done before T1 overwrites the blocks
#pragma unroll
            for (int u = 0; u < P::HS; ++u) {
                const int slot = 2 * u + wh;
                if (slot < ns) {
                    const int mi = gh[GH_MLPIDX + slot], col = wn * 32 + c_oct(lane);
                    const f32x4 t0 = relu4(acc[u].c[0]), t1v = relu4(acc[u].c[1]);
                    lds_store_oct<T>(smem, scr[slot], win, col, t0, t1v);
                    if (w_ok) store_oct(t1 + act_idx(w, mi, B) + col, t0, t1v);
                }
            }
            __syncthreads();
            mlp_gemm<T>(acc, smem, scr, ns, wpack, gh[GH_W2], a.bias + (size_t)gh[GH_B2] * H, wn, wh, lane);
#pragma unroll
            for (int u = 0; u < P::HS; ++u) {
                const int slot = 2 * u + wh;
                if (slot < ns && w_ok) {
                    const int n = gh[GH_NODES + slot], col = wn * 32 + c_oct(lane);
                    f32x4 y0 = acc[u].c[0], y1 = acc[u].c[1];
                    if (flags & GF_RESIDUAL) {
                        f32x4 r0, r1;
                        if (own) load_oct(xin + act_idx(w, n, B) + col, r0, r1);
                        else lds_load_oct<T>(smem, n, win, col, r0, r1);
                        y0 += r0; y1 += r1;
                    }
                    store_oct(xout + act_idx(w, n, B) + col, y0, y1);
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------------
// k_layer_bwd: G_{l+1} = G_{l+2} + D_{l+1} (coalesced, materialised), dH_l from it (relu bits / base_transform
// chain), then D_l on the transposed graph.  Layer 0 finishes dX_0 = relu'(X_0) . (G_1 + D_0) itself.
// ------------------------------------------------------------------------------------------------------
template <typename T> __global__ __launch_bounds__(LAYER_THREADS, Prec<T>::WPS) void k_layer_bwd(LayerArgs a) {
    using P = Prec<T>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wv & 3, wh = wv >> 2;
    const int w0 = blockIdx.x * P::ROWS, B = a.B, NN = a.NN;
    const T* gtop = reinterpret_cast<const T*>(a.x_in);   // dX_{l+1}
    T* dxo = reinterpret_cast<T*>(a.x_out);               // dX_l
    T* dh = reinterpret_cast<T*>(a.dh);
    const T* xact = reinterpret_cast<const T*>(a.x_act);
    const T* wpack = reinterpret_cast<const T*>(a.wpack);
    const int* pg = a.prog;
    const int ngroups = pg[BH_NGROUPS], nmlp = pg[BH_NMLP], w2pack = pg[BH_W2], w1pack = pg[BH_W1];
    const int* node_kind = pg + BH_KIND;
    const int* mlp_nodes = pg + BH_MLPNODES;
    WaveProg wp(pg + BH_SIZE + ngroups * GH_SIZE + wh * WPROG_LEN, lane);   // this wave's MAC program of the layer, loaded once
    const int win = c_win(lane), w = w0 + win;
    const bool w_ok = w < B;

    // stage 1 (row-major, coalesced): G_{l+1} -> LDS; relu nodes masked (-> dH, also to global for k_gradw)
    {
        constexpr int EPC = P::EPC, NPB = RowMap<T>::NPB, BATCH = 5;
        const RowMap<T> m(tid);
        const int row = m.row, c = m.c, wr = w0 + row;
        for (int nb = m.sub; nb < NN; nb += NPB * BATCH) {
            u32x4 v[BATCH]; unsigned word[BATCH]; int nk[BATCH];
#pragma unroll
            for (int i = 0; i < BATCH; ++i) {       // issue every load of the batch first
                const int n = nb + i * NPB;
                nk[i] = n < NN ? node_kind[n] : NK_DEAD;
                v[i] = u32x4{0, 0, 0, 0}; word[i] = 0;
                if (nk[i] != NK_DEAD && wr < B && !ABL(a.dbg & 16)) {
                    v[i] = *reinterpret_cast<const u32x4*>(gtop + act_idx(wr, n, B) + c * EPC);
                    if (nk[i] == NK_RELU) word[i] = reinterpret_cast<const uint8_t*>(a.maskbits)[relu_byte(n, B, wr, c * EPC)];
                }
            }
#pragma unroll
            for (int i = 0; i < BATCH; ++i) {
                const int n = nb + i * NPB;
                if (nk[i] == NK_DEAD) continue;
                if (nk[i] == NK_RELU) {
                    v[i] = chunk_mask_bits<T>(v[i], word[i] >> ((c * EPC) % 8));   // dH (k_gradw recomputes it the same way)
                }
                *reinterpret_cast<u32x4*>(smem + lds_chunk<T>(n, row, c)) = v[i];
            }
        }
    }
    __syncthreads();

    typename P::Acc acc[P::HS];

    if (nmlp > 0 && !ABL(a.dbg & 32)) {
        // dT1 = dY W2 ; dU = dT1 . (T1 > 0) ; dH = dU W1     (backward of base_transform; re-uses the base blocks)
        const T* t1 = reinterpret_cast<const T*>(a.t1);
        T* du = reinterpret_cast<T*>(a.du);
        mlp_gemm<T>(acc, smem, mlp_nodes, nmlp, wpack, w2pack, nullptr, wn, wh, lane);
        __syncthreads();   // all reads of dY blocks done
#pragma unroll
        for (int u = 0; u < P::HS; ++u) {
            const int slot = 2 * u + wh;
            if (slot < nmlp) {
                const int col = wn * 32 + c_oct(lane);
                f32x4 r0 = f32x4{0, 0, 0, 0}, r1 = f32x4{0, 0, 0, 0};
                if (w_ok) {
                    f32x4 tv0, tv1;
                    load_oct(t1 + act_idx(w, slot, B) + col, tv0, tv1);
#pragma unroll
                    for (int j = 0; j < 4; ++j) { r0[j] = tv0[j] > 0.f ? acc[u].c[0][j] : 0.f; r1[j] = tv1[j] > 0.f ? acc[u].c[1][j] : 0.f; }
                    store_oct(du + act_idx(w, slot, B) + col, r0, r1);
                }
                lds_store_oct<T>(smem, mlp_nodes[slot], win, col, r0, r1);
            }
        }
        __syncthreads();
        mlp_gemm<T>(acc, smem, mlp_nodes, nmlp, wpack, w1pack, nullptr, wn, wh, lane);
        __syncthreads();   // all reads of dU blocks done
#pragma unroll
        for (int u = 0; u < P::HS; ++u) {
            const int slot = 2 * u + wh;
            if (slot < nmlp) {
                const int n = mlp_nodes[slot], col = wn * 32 + c_oct(lane);
                lds_store_oct<T>(smem, n, win, col, acc[u].c[0], acc[u].c[1]);
                if (w_ok) store_oct(dh + act_idx(w, n, B) + col, acc[u].c[0], acc[u].c[1]);
            }
        }
        __syncthreads();
    }

    // stage 2: D_l[j] = dH_j W_rootsum + sum_r sum_{j->i} dH_i W_rel^r
    const int* gh = pg + BH_SIZE;
    for (int g = 0; g < ngroups; ++g, gh += GH_SIZE) {
        const int ns = gh[GH_NSLOTS], flags = gh[GH_FLAGS];
#pragma unroll
        for (int u = 0; u < P::HS; ++u) acc_fill(acc[u], 0.f);
        run_segments<T>(wp, gh[GH_PC0 + wh], acc, smem, wpack, wn, lane, a.dbg);
        if ABL(a.dbg & 64) continue;
        if (flags & GF_LDS_EPI) {
            __syncthreads();   // last group: all dH blocks are dead -> stage D through LDS, store whole rows
#pragma unroll
            for (int u = 0; u < P::HS; ++u) {
                const int slot = 2 * u + wh;
                if (slot < ns) lds_store_oct<T>(smem, gh[GH_NODES + slot], win, wn * 32 + c_oct(lane), acc[u].c[0], acc[u].c[1]);
            }
            __syncthreads();
            const RowMap<T> m(tid);
            const int row = m.row, c = m.c;
            if (w0 + row < B)
                for (int sl0 = m.sub; sl0 < ns; sl0 += 2 * RowMap<T>::NPB) {
                    u32x4 v[2], g1[2], xa[2]; bool ok[2];
#pragma unroll
                    for (int i = 0; i < 2; ++i) {       // two nodes in flight per thread
                        const int sl = sl0 + i * RowMap<T>::NPB;
                        ok[i] = sl < ns;
                        const int n = gh[GH_NODES + (ok[i] ? sl : 0)];
                        const size_t idx = act_idx(w0 + row, n, B) + c * P::EPC;
                        v[i] = *reinterpret_cast<const u32x4*>(smem + lds_chunk<T>(n, row, c));
                        g1[i] = u32x4{0, 0, 0, 0}; xa[i] = u32x4{0, 0, 0, 0};
                        if (ok[i] && (flags & GF_RESIDUAL)) g1[i] = *reinterpret_cast<const u32x4*>(gtop + idx);
                        if (ok[i] && (flags & GF_ENC_MASK)) xa[i] = *reinterpret_cast<const u32x4*>(xact + idx);
                    }
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        if (!ok[i]) continue;
                        const int n = gh[GH_NODES + sl0 + i * RowMap<T>::NPB];
                        const size_t idx = act_idx(w0 + row, n, B) + c * P::EPC;
                        u32x4 r = v[i];
                        if (flags & GF_RESIDUAL) r = chunk_add<T>(r, g1[i]);             // dX_l = dX_{l+1} + D_l
                        if (flags & GF_ENC_MASK) r = chunk_mask_pos<T>(r, xa[i]);        // layer 0: x relu'(X_0)
                        *reinterpret_cast<u32x4*>(dxo + idx) = r;
                    }
                }
        } else {
#pragma unroll
            for (int u = 0; u < P::HS; ++u) {
                const int slot = 2 * u + wh;
                if (slot < ns && w_ok) {
                    const int n = gh[GH_NODES + slot];
                    const size_t idx = act_idx(w, n, B) + wn * 32 + c_oct(lane);
                    f32x4 g1[2] = {f32x4{0, 0, 0, 0}, f32x4{0, 0, 0, 0}}, xa[2] = {f32x4{1, 1, 1, 1}, f32x4{1, 1, 1, 1}};
                    if (flags & GF_RESIDUAL) load_oct(gtop + idx, g1[0], g1[1]);
                    if (flags & GF_ENC_MASK) load_oct(xact + idx, xa[0], xa[1]);
                    f32x4 y[2];
#pragma unroll
                    for (int fb = 0; fb < 2; ++fb) {
                        y[fb] = round_as<T>(acc[u].c[fb]) + g1[fb];
#pragma unroll
                        for (int j = 0; j < 4; ++j) y[fb][j] = xa[fb][j] > 0.f ? y[fb][j] : 0.f;
                    }
                    store_oct(dxo + idx, y[0], y[1]);
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------------
// FUSED STACK kernels (bf16 plan): the whole message-passing stack of one 16-window tile in ONE workgroup.
//   k_stack_fwd: X_0 tile -> LDS, L x (HeteroConv + relu / base_transform + residual) in place, decoder -> out.
//   k_stack_bwd: dX_L tile -> LDS, L x (relu' / base_transform backward, transposed HeteroConv, residual) in place.
// Every node owns one accumulator per wave for the whole layer (node n -> wave half n & 1, accumulator n >> 1), so a
// layer's epilogue runs once, after every wave has finished reading the previous activations, and overwrites them in
// place; the stashes the backward pass / k_gradw need are written on the side and never read back by this kernel.
// One workgroup of 8 waves per CU (<= 256 VGPRs per wave): all accumulators live + double-buffered weight fragments.

// STEP: part of k_stack_step -- the decoder tail leaves dX_L in the out-type nodes' LDS blocks for the backward sweep that follows in the same launch
template <typename T, bool STEP> __device__ __forceinline__ void stack_fwd_body(const StackArgs& a, char* smem) {
    using P = Prec<T>;
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wv & 3, wh = wv >> 2;
    const int w0 = blockIdx.x * P::ROWS, B = a.B, NN = a.NN;
    const T* wpack = reinterpret_cast<const T*>(a.wpack);
    const int win = c_win(lane), w = w0 + win, col = wn * 32 + c_oct(lane);
    const bool w_ok = w < B, train = a.training != 0;

    FS_STAMP(0);
    stage_nodes<T>(smem, reinterpret_cast<const T*>(a.tile_in), NN, w0, B, tid);
    __syncthreads();
    FS_STAMP(1);

    typename P::Acc acc[FS_HS];
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
        const int nmlp = fh[FH_NMLP], flags = fh[FH_FLAGS];
        // accumulators start at the bias row of their node's type (the loads hide under the first weight fragment)
#pragma unroll
        for (int u = 0; u < FS_HS; ++u) {
            const int n = 2 * u + wh;
            if (n < NN && fh[FH_KIND + n] != NK_DEAD) acc_init_bias<T>(acc[u], a.bias + (size_t)fh[FH_BIAS + n] * H, wn, lane);
            else acc_fill(acc[u], 0.f);
        }
        FS_STAMP(2 + 4 * l);
#ifdef MSHGNN_SEG_STAMPS
        // per-wave clock at every segment start of every layer: [layer][wave][16] in LDS behind the node blocks, dumped at the end
        long long* segclk = a.stamps ? reinterpret_cast<long long*>(smem + (NN + 4) * P::BLK) + (l * 8 + wv) * 16 : nullptr;
        fs_run<T>(wp, acc, smem, wpack, wn, lane, a.dbg, segclk);
#else
        if (!ABL(a.dbg & 2)) fs_run<T>(wp, acc, smem, wpack, wn, lane, a.dbg);
#endif
        FS_STAMP(3 + 4 * l);
        __syncthreads();   // every wave is done reading X_l: the node blocks may be overwritten
        FS_STAMP(4 + 4 * l);
        if ABL(a.dbg & 8) continue;

        u32x4 hpk[2] = {}, tpk[2] = {};
        if (nmlp > 0 && !ABL(a.dbg & 64)) {
            // base_transform: Y = W2 relu(W1 H + b1) + b2 on the first nmlp nodes (hgnn_c2.py:117-121,156); scratch blocks NN + i.
            // The H and T1 stashes are kept packed in registers and stored after the chain: a load waited for while stores are in
            // flight costs a full drain of those stores.
            static_assert(sizeof(T) == 2, "fused stack kernels are bf16");
            typename P::BFrag bf, bf2;
            typename P::AFrag af;
            load_bfrag<T>(bf, wpack, fh[FH_W1], wn, lane);
            load_bfrag<T>(bf2, wpack, fh[FH_W2], wn, lane);
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int n = 2 * u + wh;
                if (n < nmlp) {
                    hpk[u] = pack_oct(acc[u].c[0], acc[u].c[1]);
                    *reinterpret_cast<u32x4*>(smem + lds_chunk<T>(NN + n, win, col / P::EPC)) = hpk[u];
                    acc_init_bias<T>(acc[u], a.bias + (size_t)fh[FH_B1] * H, wn, lane);
                }
            }
            __syncthreads();
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int n = 2 * u + wh;
                if (n < nmlp) {
                    load_afrag<T>(af, smem, NN + n, lane);
                    mac(acc[u], af, bf);
                }
            }
            __syncthreads();   // all reads of H done before T1 overwrites the scratch blocks
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int n = 2 * u + wh;
                if (n < nmlp) {
                    tpk[u] = pack_oct(relu4(acc[u].c[0]), relu4(acc[u].c[1]));
                    *reinterpret_cast<u32x4*>(smem + lds_chunk<T>(NN + n, win, col / P::EPC)) = tpk[u];
                    acc_init_bias<T>(acc[u], a.bias + (size_t)fh[FH_B2] * H, wn, lane);
                }
            }
            __syncthreads();
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int n = 2 * u + wh;
                if (n < nmlp) {
                    load_afrag<T>(af, smem, NN + n, lane);
                    mac(acc[u], af, bf2);
                }
            }
        }
        FS_STAMP(16 + l);
        // every load issued so far (chain operands, the next layer's header and program) has landed before the first store of
        // the epilogue goes out: from here to the next layer's first weight fragment nothing waits on a load
        __builtin_amdgcn_s_waitcnt((7 << 4) | (15 << 8));
        fhn.settle(); wpn.settle();      // (and the compiler stops tracking them as pending: no vmcnt(0) at the top of the next layer, FProg::settle)
        if (nmlp > 0 && train && w_ok) {
            T* hb = reinterpret_cast<T*>(a.ws + a.hb_off[l]);
            T* t1 = reinterpret_cast<T*>(a.ws + a.t1_off[l]);
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int n = 2 * u + wh;
                if (n < nmlp) {
                    *reinterpret_cast<u32x4*>(hb + act_idx(w, n, B) + col) = hpk[u];
                    *reinterpret_cast<u32x4*>(t1 + act_idx(w, n, B) + col) = tpk[u];
                }
            }
        }

        // X_{l+1}[n] = f(H[n]) (+ X_l[n]) for every live node, in place; stash + relu bits on the side
        T* xo = reinterpret_cast<T*>(a.ws + a.x_off[l + 1]);
        unsigned* maskbits = reinterpret_cast<unsigned*>(a.ws + a.mask_off[l]);
        u32x4 resv[FS_HS]; int kindv[FS_HS];     // the residual octets of every node, all LDS reads in flight together
#pragma unroll
        for (int u = 0; u < FS_HS; ++u) {
            const int n = 2 * u + wh;
            kindv[u] = n < NN ? fh[FH_KIND + n] : NK_DEAD;
            resv[u] = u32x4{0, 0, 0, 0};
            if (kindv[u] != NK_DEAD && (flags & FF_RESIDUAL)) resv[u] = *reinterpret_cast<const u32x4*>(smem + lds_chunk<T>(n, win, col / P::EPC));
        }
#pragma unroll
        for (int u = 0; u < FS_HS; ++u) {
            const int n = 2 * u + wh;
            if (n < NN) {
                const int kind = kindv[u];
                if (kind != NK_DEAD) {
                    f32x4 y0 = acc[u].c[0], y1 = acc[u].c[1];
                    if (kind == NK_RELU) {
                        const unsigned bits = relu_with_bits<T>(acc[u]);
                        if (train) reinterpret_cast<uint8_t*>(maskbits)[relu_byte(n, B, w, col)] = (uint8_t)bits;
                        y0 = acc[u].c[0]; y1 = acc[u].c[1];
                    }
                    if (flags & FF_RESIDUAL) {
                        const u32x4 r = resv[u];
                        y0 += f32x4{__builtin_bit_cast(float, r[0] << 16), __builtin_bit_cast(float, r[0] & 0xffff0000u),
                                    __builtin_bit_cast(float, r[1] << 16), __builtin_bit_cast(float, r[1] & 0xffff0000u)};
                        y1 += f32x4{__builtin_bit_cast(float, r[2] << 16), __builtin_bit_cast(float, r[2] & 0xffff0000u),
                                    __builtin_bit_cast(float, r[3] << 16), __builtin_bit_cast(float, r[3] & 0xffff0000u)};
                    }
                    lds_store_oct<T>(smem, n, win, col, y0, y1);
                    if (train && w_ok && !ABL(a.dbg & 16) && !(STEP && l + 1 == a.L)) store_oct(xo + act_idx(w, n, B) + col, y0, y1);      // (X_L of a one-launch step is read by nobody: the decoder's gradients come from the tile in LDS)
                }
            }
        }
        __syncthreads();
        FS_STAMP(5 + 4 * l);
    }

    decoder_tail<T, LAYER_THREADS, false, STEP>(a, smem, tid, lane, wv, w0, B);
    FS_STAMP(30);
#ifdef MSHGNN_SEG_STAMPS
    if (a.stamps) {
        __syncthreads();
        const long long* sc = reinterpret_cast<const long long*>(smem + (NN + 4) * P::BLK);
        if (tid < 3 * 8 * 16) a.stamps[(size_t)gridDim.x * 32 + (size_t)blockIdx.x * 384 + tid] = sc[tid];
    }
#endif
}
template <typename T> __global__ __launch_bounds__(LAYER_THREADS, 2) void k_stack_fwd(StackArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    stack_fwd_body<T, false>(a, smem);
}

// ------------------------------------------------------------------------------------------------------
// Slab variant of the forward stack kernel: 4 waves per workgroup, two workgroups per CU.  Wave wn owns columns
// [32 wn, 32 wn + 32) of EVERY node, so each weight pack is pulled through the CU's vector L1 once per tile (the 8-wave
// kernel pulls it once per wave half, and its MAC phase is bound by that path), and the second workgroup's epilogues run
// under this one's MAC phases.  The destination nodes are processed in two groups (mshgnn_plan.hpp, SL_HA / SL_HB) so that
// the accumulators stay in registers; group A's new activations wait, packed, while group B is multiplied.
// ------------------------------------------------------------------------------------------------------
template <typename T, int HS, int Q0, int NM = 4, bool PRE = false, class A = StackArgs, class FH = FHdr, class FP = FProg>     // one group: slots q = Q0 + u of the slab header; NM: compile-time bound on the base_transform nodes; PRE: (*pre) = the run's first two weight fragments, already requested
__device__ __forceinline__ void slab_group_fwd(const A& a, const FH& fh, const FP& wp, char* smem, const T* wpack, int wn, int lane,
                                               int slot_base, int nmlp, bool residual, u32x4 (&keep)[HS], unsigned (&bits)[(HS + 3) / 4], typename Prec<T>::BFrag (*pre)[2] = nullptr) {
    using P = Prec<T>;
    const int win = c_win(lane), col = wn * 32 + c_oct(lane);
    if (fh[FH_FLAGS] & (Q0 == 0 ? FF_A_EMPTY : FF_B_EMPTY)) {      // no live node in this group in this layer (uniform): nothing to compute or keep
#pragma unroll
        for (int u = 0; u < HS; ++u) keep[u] = u32x4{0, 0, 0, 0};
#pragma unroll
        for (int i = 0; i < (HS + 3) / 4; ++i) bits[i] = 0;
        return;
    }
    typename P::Acc acc[HS];
    const unsigned boff = (unsigned)c_oct(opaque(lane)) * 4u;      // (compile-time programs) this lane's 8 floats in a wave's 32-float slice of a bias row
#pragma unroll
    for (int u = 0; u < HS; ++u) {
        if (fh[FH_KIND + Q0 + u] != NK_DEAD) {
            if constexpr (FH::is_static) acc_init_bias_u<T>(acc[u], a.bias + (size_t)fh[FH_BIAS + Q0 + u] * H, wn, boff);
            else acc_init_bias<T>(acc[u], a.bias + (size_t)fh[FH_BIAS + Q0 + u] * H, wn, lane);
        } else acc_fill(acc[u], 0.f);
    }
    if constexpr (PRE) fs_run_static<T, HS, (Q0 == 0 ? SL_CBA : SL_CBB), FP, true>(acc, smem, wpack, wn, lane, *pre);
    else fs_run<T, HS, (Q0 == 0 ? SL_CBA : SL_CBB)>(wp, acc, smem, wpack, wn, lane, a.dbg);      // (a.dbg: timing ablations, compiled out of the product build)
    if constexpr (Q0 > 0) if (nmlp > 0) {
        // base_transform: Y = W2 relu(W1 H + b1) + b2 on the first nmlp slots of this group (hgnn_c2.py:117-121,156); scratch
        // blocks NN + u.  The H / T1 stashes go out packed, behind the chain's last load.
        static_assert(sizeof(T) == 2, "fused stack kernels are bf16");
        typename P::BFrag bf, bf2;
        typename P::AFrag af;
        u32x4 hpk[NM], tpk[NM];
        // scratch blocks: behind the tile, or (FF_SCR_ALIAS) the blocks of the first group-A nodes -- every wave must then be done with ALL
        // its MACs (and group A's residual reads) before H lands in them
        const bool alias = (fh[FH_FLAGS] & FF_SCR_ALIAS) != 0;
        int scr[NM];
#pragma unroll
        for (int u = 0; u < NM; ++u) scr[u] = alias ? fh[FH_SLOTA + u] : a.NN + u;
        if (alias) __syncthreads();
        load_bfrag<T>(bf, wpack, fh[FH_W1], wn, lane);
        load_bfrag<T>(bf2, wpack, fh[FH_W2], wn, lane);
#pragma unroll
        for (int u = 0; u < NM && u < HS; ++u) {
            if (u < nmlp) {
                hpk[u] = pack_oct(acc[u].c[0], acc[u].c[1]);
                *reinterpret_cast<u32x4*>(smem + lds_chunk<T>(scr[u], win, col / P::EPC)) = hpk[u];
                acc_init_bias<T>(acc[u], a.bias + (size_t)fh[FH_B1] * H, wn, lane);
            }
        }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < NM && u < HS; ++u) {
            if (u < nmlp) { load_afrag<T>(af, smem, scr[u], lane); mac(acc[u], af, bf); }
        }
        __syncthreads();   // all reads of H done before T1 overwrites the scratch blocks
#pragma unroll
        for (int u = 0; u < NM && u < HS; ++u) {
            if (u < nmlp) {
                tpk[u] = pack_oct(relu4(acc[u].c[0]), relu4(acc[u].c[1]));
                *reinterpret_cast<u32x4*>(smem + lds_chunk<T>(scr[u], win, col / P::EPC)) = tpk[u];
                acc_init_bias<T>(acc[u], a.bias + (size_t)fh[FH_B2] * H, wn, lane);
            }
        }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < NM && u < HS; ++u) {
            if (u < nmlp) { load_afrag<T>(af, smem, scr[u], lane); mac(acc[u], af, bf2); }
        }
        __builtin_amdgcn_s_waitcnt((7 << 4) | (15 << 8));      // every load has landed before the first store
        if (a.training) {
            const int w = blockIdx.x * P::ROWS + win;
            if (A::full || w < a.B) {
                T* hb = reinterpret_cast<T*>(a.ws + a.hb_off[slot_base >> 8]);
                T* t1 = reinterpret_cast<T*>(a.ws + a.t1_off[slot_base >> 8]);
                const unsigned roff = (unsigned)((w * H + col) * (int)sizeof(T));
#pragma unroll
                for (int u = 0; u < NM && u < HS; ++u) {
                    if (u < nmlp) {
                        if constexpr (FH::is_static) {
                            gstore16(uniform_wptr(reinterpret_cast<char*>(hb + act_idx(0, u, a.B))), roff, hpk[u]);
                            gstore16(uniform_wptr(reinterpret_cast<char*>(t1 + act_idx(0, u, a.B))), roff, tpk[u]);
                        } else {
                            *reinterpret_cast<u32x4*>(hb + act_idx(w, u, a.B) + col) = hpk[u];
                            *reinterpret_cast<u32x4*>(t1 + act_idx(w, u, a.B) + col) = tpk[u];
                        }
                    }
                }
            }
        }
    }
    // X_{l+1}[n] = f(H[n]) (+ X_l[n]), kept packed in registers (X_l is still being read by the other waves)
#pragma unroll
    for (int i = 0; i < (HS + 3) / 4; ++i) bits[i] = 0;
#pragma unroll
    for (int u = 0; u < HS; ++u) {
        keep[u] = u32x4{0, 0, 0, 0};
        const int kind = fh[FH_KIND + Q0 + u];
        if (kind != NK_DEAD) {
            const int n = fh[(slot_base & 255) + u];
            if (kind == NK_RELU) bits[u >> 2] |= relu_with_bits<T>(acc[u]) << (8 * (u & 3));
            f32x4 y0 = acc[u].c[0], y1 = acc[u].c[1];
            if (residual) {
                const u32x4 r = *reinterpret_cast<const u32x4*>(smem + lds_chunk<T>(n, win, col / P::EPC));
                f32x4 r0, r1; unpack_oct(r, r0, r1);
                y0 += r0; y1 += r1;
            }
            keep[u] = pack_oct(y0, y1);
            pad_valu();
        }
    }
}
// write one group's new activations: LDS block, stash, relu bytes
template <typename T, int HS, int Q0, class A = StackArgs, class FH = FHdr>
__device__ __forceinline__ void slab_group_store(const A& a, const FH& fh, char* smem, int wn, int lane, int slot_arr, int l,
                                                 const u32x4 (&keep)[HS], const unsigned (&bits)[(HS + 3) / 4], bool stash_x = true) {
    using P = Prec<T>;
    const int win = c_win(lane), col = wn * 32 + c_oct(lane), w = blockIdx.x * P::ROWS + win;
    T* xo = reinterpret_cast<T*>(a.ws + a.x_off[l + 1]);
    uint8_t* maskbytes = reinterpret_cast<uint8_t*>(a.ws + a.mask_off[l]);
    if constexpr (FH::is_static) {
        // compile-time program: the LDS rows and relu bytes of every live slot first, then all stash rows under ONE lane predicate (one exec switch per group
        // instead of a compare + branch per node)
        // every global address = a scalar base (layer, node) + this lane's 32-bit offset, rebuilt from an opaque copy of the lane id per call: per-node 64-bit
        // addresses would be shared between the unrolled layers and live (two registers per node) through the whole kernel
        const int lq = opaque(lane);
        const unsigned roff = (unsigned)(((blockIdx.x * P::ROWS + c_win(lq)) * H + wn * 32 + c_oct(lq)) * (int)sizeof(T));      // this lane's octet in a [B][128] row block
#pragma unroll
        for (int u = 0; u < HS; ++u) {
            const int kind = fh[FH_KIND + Q0 + u];
            if (kind != NK_DEAD) {
                const int n = fh[slot_arr + u];
                *reinterpret_cast<u32x4*>(smem + lds_chunk<T>(n, win, col / P::EPC)) = keep[u];
                if (a.training && kind == NK_RELU) gstore1(uniform_wptr(reinterpret_cast<char*>(maskbytes) + relu_tile_base(n, a.B, blockIdx.x, wn)), (unsigned)lq, bits[u >> 2] >> (8 * (u & 3)));
            }
        }
        if (a.training && stash_x && (A::full || w < a.B)) {
#ifdef MSHGNN_STASH_ALIAS      // timing experiment (wrong results): every tile's stash rows land on the rows of 32 tiles, a footprint the L2 holds -- is the store phase bound by HBM writes?
            const unsigned soff = (unsigned)((((blockIdx.x & 31) * P::ROWS + c_win(lq)) * H + wn * 32 + c_oct(lq)) * (int)sizeof(T));
#else
            const unsigned soff = roff;
#endif
#pragma unroll
            for (int u = 0; u < HS; ++u)
                if (fh[FH_KIND + Q0 + u] != NK_DEAD) stash_store_u(a, uniform_wptr(reinterpret_cast<char*>(xo + act_idx(0, fh[slot_arr + u], a.B))), soff, keep[u]);
        }
        return;
    }
#pragma unroll
    for (int u = 0; u < HS; ++u) {
        const int kind = fh[FH_KIND + Q0 + u];
        if (kind != NK_DEAD) {
            const int n = fh[slot_arr + u];
            *reinterpret_cast<u32x4*>(smem + lds_chunk<T>(n, win, col / P::EPC)) = keep[u];
            if (a.training) {
                if (kind == NK_RELU) maskbytes[relu_tile_base(n, a.B, blockIdx.x, wn) + lane] = (uint8_t)(bits[u >> 2] >> (8 * (u & 3)));
                if (w < a.B && stash_x) stash_store(xo + act_idx(w, n, a.B) + col, keep[u], a.stash_nt != 0);
            }
        }
    }
}

// one forward layer of a slab workgroup: both groups' MAC passes, then the layer's stores.  FH / FP: the layer's header and wave programs, interpreted (FHdr /
// FProg: plan tables in VGPRs) or compile-time (SHdr / SProg: specialised kernels); mid(): what has to settle between the MACs and the stores
// PRE (compile-time programs): (*pre) holds the first two weight fragments of the layer's first pass, requested before the previous store phase; bits_out: the
// layer's relu bits (12 + HB bytes in 3 + 2 registers) for the backward sweep of the same launch
template <typename T, int NM, int HB, bool STEP, bool PRE = false, class A, class FH, class FP, class FPB, class Mid>
__device__ __forceinline__ void slab_fwd_layer(const A& a, char* smem, const T* wpack, int wn, int lane, int l, int L, const FH& fh, const FP& wa, const FPB& wb, Mid&& mid,
                                               typename Prec<T>::BFrag (*pre)[2] = nullptr, unsigned* bits_out = nullptr) {
    const int tid = threadIdx.x;
    const int nmlp = fh[FH_NMLP];
    const bool residual = (fh[FH_FLAGS] & FF_RESIDUAL) != 0;
    u32x4 keepA[SL_HA], keepB[HB]; unsigned bitsA[(SL_HA + 3) / 4], bitsB[(HB + 3) / 4];
    if constexpr (PRE) {      // the requested fragments belong to the first group with work
        constexpr bool a_first = !(FH{}[FH_FLAGS] & FF_A_EMPTY);
        slab_group_fwd<T, SL_HA, 0, 4, a_first>(a, fh, wa, smem, wpack, wn, lane, FH_SLOTA | (l << 8), 0, residual, keepA, bitsA, pre);
        FS_STAMP(2 + 4 * l);
        slab_group_fwd<T, HB, SL_HA, NM, !a_first>(a, fh, wb, smem, wpack, wn, lane, FH_SLOTB | (l << 8), nmlp, residual, keepB, bitsB, pre);
    } else {
        slab_group_fwd<T, SL_HA, 0>(a, fh, wa, smem, wpack, wn, lane, FH_SLOTA | (l << 8), 0, residual, keepA, bitsA);
        FS_STAMP(2 + 4 * l);
        slab_group_fwd<T, HB, SL_HA, NM>(a, fh, wb, smem, wpack, wn, lane, FH_SLOTB | (l << 8), nmlp, residual, keepB, bitsB);
    }
    if (bits_out) {
#pragma unroll
        for (int i = 0; i < (SL_HA + 3) / 4; ++i) bits_out[i] = bitsA[i];
#pragma unroll
        for (int i = 0; i < (HB + 3) / 4; ++i) bits_out[(SL_HA + 3) / 4 + i] = bitsB[i];
    }
    FS_STAMP(3 + 4 * l);
    __syncthreads();   // every wave is done reading X_l: the node blocks may be overwritten
    FS_STAMP(4 + 4 * l);
    mid();      // the next header / programs have landed before the stores go out (no drain at the top of the next layer)
    const bool stash_x = !(STEP && l + 1 == L);      // X_L of a one-launch step is read by nobody (the decoder's gradients come from the tile in LDS)
    slab_group_store<T, SL_HA, 0>(a, fh, smem, wn, lane, FH_SLOTA, l, keepA, bitsA, stash_x);
    slab_group_store<T, HB, SL_HA>(a, fh, smem, wn, lane, FH_SLOTB, l, keepB, bitsB, stash_x);
    __syncthreads();
    FS_STAMP(5 + 4 * l);
}
// the layers of a compile-time program SP, unrolled
// the first pass with work of forward (DIR 0) / backward (DIR 1) layer l of a compile-time program: request its first two weight fragments
template <typename T, class SP, int DIR, int l> __device__ __forceinline__ void slab_prefetch_static(typename Prec<T>::BFrag (&pre)[2], const T* wpack, int wn, int lane) {
    if constexpr (!(SHdr<SP, DIR, l>{}[FH_FLAGS] & FF_A_EMPTY)) fs_prefetch_static<T, SProg<SP, DIR, l, 0>>(pre, wpack, wn, lane);
    else fs_prefetch_static<T, SProg<SP, DIR, l, 1>>(pre, wpack, wn, lane);
}
// Between a layer's MACs and its stores a compile-time program requests what the phase AFTER the stores starts with -- the next layer's first two weight
// fragments, or (last layer) the decoder tail's operands -- so that phase begins under the store drain instead of behind it.
template <typename T, int NM, int HB, bool STEP, class SP, int l = 0, class A>
__device__ __forceinline__ void slab_fwd_layers_static(const A& a, char* smem, const T* wpack, int wn, int lane, typename Prec<T>::BFrag (&pre)[2],
                                                       unsigned* lastbits, DecOps<SP::DMAX, DEC_NPP_STATIC>& dops) {
    if constexpr (l < SP::L) {
        auto mid = [&] {
            if constexpr (l + 1 < SP::L) { if constexpr ((SP::PRE & 1) != 0) slab_prefetch_static<T, SP, 0, l + 1>(pre, wpack, wn, lane); }
            else if constexpr ((SP::PRE & 4) != 0) decoder_ops_load<SL_THREADS, SP::DMAX, DEC_NPP_STATIC>(args_of(a), threadIdx.x, blockIdx.x * Prec<T>::ROWS, a.B, threadIdx.x >> 8, true, dops);
        };
        slab_fwd_layer<T, NM, HB, STEP, (SP::PRE & 1) != 0>(a, smem, wpack, wn, lane, l, SP::L, SHdr<SP, 0, l>{}, SProg<SP, 0, l, 0>{}, SProg<SP, 0, l, 1>{}, mid, &pre,
                                                            l + 1 == SP::L ? lastbits : nullptr);
        slab_fwd_layers_static<T, NM, HB, STEP, SP, l + 1>(a, smem, wpack, wn, lane, pre, lastbits, dops);
    }
}

// STEP: part of k_slab_step -- the decoder tail leaves dX_L in the out-type nodes' LDS blocks for the backward sweep that follows in the same launch
// SP: void = the plan's tables are interpreted; else the compile-time program of one (topology, depth) (mshgnn_spec_tables.inc)
template <typename T, int NM, int HB, bool STEP, class SP = void, class A = StackArgs> __device__ __forceinline__ void slab_fwd_body(const A& a, char* smem, unsigned* lastbits = nullptr) {
    using P = Prec<T>;
    constexpr bool DYN = std::is_void<SP>::value;
    const int tid = threadIdx.x, lane = tid & 63, wn = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int w0 = blockIdx.x * P::ROWS, B = a.B, NN = a.NN;
    const T* wpack = reinterpret_cast<const T*>(a.wpack);
    stack_stagger(a);
    FS_STAMP(0);

    // layer 0's header and programs stream in under the tile load
    FHdr fhn; FProg wan, wbn;
    typename P::BFrag pre[2];
    if constexpr (DYN) {
        fhn = FHdr(a.tables + a.prog_off[0], lane);
        wan = FProg(a.tables + a.prog_off[0] + FH_SIZE, lane); wbn = FProg(a.tables + a.prog_off[0] + FH_SIZE + FPROG_LEN, lane);
    } else if constexpr ((SP::PRE & 1) != 0) slab_prefetch_static<T, SP, 0, 0>(pre, wpack, wn, lane);      // layer 0's first weight fragments stream in under the tile load
    {   // X_0 tile -> LDS: thread = (row, 16-byte chunk), one node per pass, 6 loads in flight
        const T* src = reinterpret_cast<const T*>(a.tile_in);
        const int row = tid >> 4, c = tid & 15;
        constexpr int BATCH = 6;
        for (int nb = 0; nb < NN; nb += BATCH) {
            u32x4 v[BATCH];
#pragma unroll
            for (int i = 0; i < BATCH; ++i) {
                v[i] = u32x4{0, 0, 0, 0};
                // (read once by this kernel: non-temporal -- 0.2026 -> 0.2004 ms/step over six alternating runs at 3 layers, nothing at 8)
                if (nb + i < NN && (A::full || w0 + row < B)) v[i] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(src + act_idx(w0 + row, nb + i, B) + c * P::EPC));
            }
#pragma unroll
            for (int i = 0; i < BATCH; ++i)
                if (nb + i < NN) *reinterpret_cast<u32x4*>(smem + lds_chunk<T>(nb + i, row, c)) = v[i];
        }
    }
    __syncthreads();
    FS_STAMP(1);

    if constexpr (DYN) {
        fhn.settle(); wan.settle(); wbn.settle();      // (waited for here, not by a vmcnt(0) at the top of every layer)
        for (int l = 0; l < a.L; ++l) {
            const FHdr fh = fhn;
            const FProg wa = wan, wb = wbn;
            if (l + 1 < a.L) {    // the next layer's header and wave programs stream in under this layer's MACs
                fhn = FHdr(a.tables + a.prog_off[l + 1], lane);
                wan = FProg(a.tables + a.prog_off[l + 1] + FH_SIZE, lane);
                wbn = FProg(a.tables + a.prog_off[l + 1] + FH_SIZE + FPROG_LEN, lane);
            }
            slab_fwd_layer<T, NM, HB, STEP>(a, smem, wpack, wn, lane, l, a.L, fh, wa, wb, [&] { fhn.settle(); wan.settle(); wbn.settle(); });
        }
        decoder_tail<T, SL_THREADS, false, STEP>(args_of(a), smem, tid, lane, wn, w0, B);
    } else {
        DecOps<SP::DMAX, DEC_NPP_STATIC> dops;
        slab_fwd_layers_static<T, NM, HB, STEP, SP>(a, smem, wpack, wn, lane, pre, lastbits, dops);
        decoder_tail_impl<T, SL_THREADS, SP::DMAX, false, STEP, DEC_NPP_STATIC, (SP::PRE & 4) != 0>(args_of(a), smem, tid, lane, wn, w0, B, &dops);
    }
    FS_STAMP(30);
}
template <typename T, int NM, int HB> __global__ __launch_bounds__(SL_THREADS, 2) void k_slab_fwd(StackArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    slab_fwd_body<T, NM, HB, false>(a, smem);
}

// STEP: part of k_stack_step -- the forward's decoder tail of the same launch left the dX_L tile in LDS; the layers' programs are a.prog_off_b
template <typename T, bool STEP> __device__ __forceinline__ void stack_bwd_body(const StackArgs& a, char* smem) {
    using P = Prec<T>;
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wv & 3, wh = wv >> 2;
    const int w0 = blockIdx.x * P::ROWS, B = a.B, NN = a.NN;
    const T* wpack = reinterpret_cast<const T*>(a.wpack);
    auto prog_of = [&](int l) { return STEP ? a.prog_off_b[l] : a.prog_off[l]; };

    FS_STAMP(0);
    // dX_L tile: only the nodes that are live in the last layer carry a gradient
    if constexpr (!STEP) {
        const FHdr bh(a.tables + prog_of(a.L - 1), lane);
        const T* src = reinterpret_cast<const T*>(a.tile_in);
        const RowMap<T> m(tid);
        for (int n = m.sub; n < NN; n += RowMap<T>::NPB) {
            if (bh[FH_KIND + n] == NK_DEAD) continue;
            u32x4 v = u32x4{0, 0, 0, 0};
            if (w0 + m.row < B) v = *reinterpret_cast<const u32x4*>(src + act_idx(w0 + m.row, n, B) + m.c * P::EPC);
            *reinterpret_cast<u32x4*>(smem + lds_chunk<T>(n, m.row, m.c)) = v;
        }
    }
    __syncthreads();
    FS_STAMP(1);

    typename P::Acc acc[FS_HS];
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
        const int nmlp = bh[FH_NMLP], flags = bh[FH_FLAGS];
        const unsigned* maskbits = reinterpret_cast<const unsigned*>(a.ws + a.mask_off[l]);
        // lane constants rebuilt per layer from an opaque copy of the lane id: the per-node 64-bit addresses derived from them were hoisted out of the
        // layer loop and spilled, and a scratch reload next to the epilogue's pending stores is a full vmcnt(0) drain
        const int lq = opaque(lane);
        const int win = c_win(lq), w = w0 + win, col = wn * 32 + c_oct(lq), g8 = (lq >> 4) << 3;
        const bool w_ok = w < B;

        // phase 1 (each lane on the octets it owns): the accumulator of node n starts at its residual term
        // G_{l+1}[n]; relu nodes are then masked in place -> dH_l[n].  Every relu-bit word and every LDS read is issued
        // before the first use: one memory latency per layer, not one per node.
        {
            unsigned mword[FS_HS]; u32x4 rawv[FS_HS]; int kindv[FS_HS];
#pragma unroll
            for (int u = 0; u < FS_HS; ++u) {
                const int n = 2 * u + wh;
                kindv[u] = n < NN ? bh[FH_KIND + n] : NK_DEAD;
                mword[u] = 0u; rawv[u] = u32x4{0, 0, 0, 0};
                if (kindv[u] == NK_RELU && w_ok) mword[u] = reinterpret_cast<const uint8_t*>(maskbits)[relu_byte(n, B, w, wn * 32 + g8)];
                if (kindv[u] != NK_DEAD) rawv[u] = *reinterpret_cast<const u32x4*>(smem + lds_chunk<T>(n, win, col / P::EPC));
            }
#pragma unroll
            for (int u = 0; u < FS_HS; ++u) {
                const int n = 2 * u + wh;
                acc_fill(acc[u], 0.f);
                if (kindv[u] != NK_DEAD) {
                    const u32x4 raw = rawv[u];
                    if (bh[FH_RES + n]) {
                        acc[u].c[0] = f32x4{__builtin_bit_cast(float, raw[0] << 16), __builtin_bit_cast(float, raw[0] & 0xffff0000u),
                                            __builtin_bit_cast(float, raw[1] << 16), __builtin_bit_cast(float, raw[1] & 0xffff0000u)};
                        acc[u].c[1] = f32x4{__builtin_bit_cast(float, raw[2] << 16), __builtin_bit_cast(float, raw[2] & 0xffff0000u),
                                            __builtin_bit_cast(float, raw[3] << 16), __builtin_bit_cast(float, raw[3] & 0xffff0000u)};
                    }
                    if (kindv[u] == NK_RELU)
                        *reinterpret_cast<u32x4*>(smem + lds_chunk<T>(n, win, col / P::EPC)) = chunk_mask_bits<T>(raw, mword[u]);
                }
            }
        }
        __syncthreads();
        FS_STAMP(2 + 5 * (a.L - 1 - l));

        if (nmlp > 0) {
            // dT1 = dY W2 ; dU = dT1 . (T1 > 0) ; dH = dU W1     (backward of base_transform, in place on nodes 0..nmlp-1)
            const T* t1 = reinterpret_cast<const T*>(a.ws + a.t1_off[l]);
            T* du = reinterpret_cast<T*>(a.ws + a.du_off[l]);
            T* dh = reinterpret_cast<T*>(a.ws + a.dh_off[l]);
            typename P::BFrag bf;
            typename P::AFrag af;
            typename P::Acc tm[2];
            f32x4 tv0[2], tv1[2];
            load_bfrag<T>(bf, wpack, bh[FH_W2], wn, lane);
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int n = 2 * u + wh;
                tv0[u] = f32x4{0, 0, 0, 0}; tv1[u] = f32x4{0, 0, 0, 0};
                if (n < nmlp) {
                    if (w_ok) load_oct(t1 + act_idx(w, n, B) + col, tv0[u], tv1[u]);
                    acc_fill(tm[u], 0.f);
                    load_afrag<T>(af, smem, n, lane);
                    mac(tm[u], af, bf);
                }
            }
            load_bfrag<T>(bf, wpack, bh[FH_W1], wn, lane);
            __syncthreads();   // all reads of the dY blocks done
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int n = 2 * u + wh;
                if (n < nmlp) {
                    f32x4 r0, r1;
#pragma unroll
                    for (int j = 0; j < 4; ++j) { r0[j] = tv0[u][j] > 0.f ? tm[u].c[0][j] : 0.f; r1[j] = tv1[u][j] > 0.f ? tm[u].c[1][j] : 0.f; }
                    if (w_ok) store_oct(du + act_idx(w, n, B) + col, r0, r1);
                    lds_store_oct<T>(smem, n, win, col, r0, r1);
                }
            }
            __syncthreads();
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int n = 2 * u + wh;
                if (n < nmlp) {
                    acc_fill(tm[u], 0.f);
                    load_afrag<T>(af, smem, n, lane);
                    mac(tm[u], af, bf);
                }
            }
            __syncthreads();   // all reads of the dU blocks done
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int n = 2 * u + wh;
                if (n < nmlp) {
                    lds_store_oct<T>(smem, n, win, col, tm[u].c[0], tm[u].c[1]);
                    if (w_ok) store_oct(dh + act_idx(w, n, B) + col, tm[u].c[0], tm[u].c[1]);
                }
            }
            __syncthreads();
        }

        // phase 2: dX_l[j] = (residual) + dH_j W_rootsum + sum_r sum_{j->i} dH_i W_rel^r
        FS_STAMP(3 + 5 * (a.L - 1 - l));
        fs_run<T>(wp, acc, smem, wpack, wn, lane);
        FS_STAMP(4 + 5 * (a.L - 1 - l));
        __syncthreads();   // every wave is done reading dH_l
        FS_STAMP(5 + 5 * (a.L - 1 - l));

        T* dxo = reinterpret_cast<T*>(a.ws + a.dx_off[l]);
        const T* xact = reinterpret_cast<const T*>(a.ws + a.x_off[0]);
        bhn.settle(); wpn.settle();      // next header / program landed before the stores go out (FProg::settle)
        // layer 0: the encoder's relu bytes of every node are requested before the first store of the epilogue (a load waited for while stores are
        // in flight drains them all: one round trip instead of one per node)
        unsigned xbv[FS_HS];
#pragma unroll
        for (int u = 0; u < FS_HS; ++u) {
            const int n = 2 * u + wh;
            xbv[u] = 0xffu;
            if ((flags & FF_ENC_MASK) && w_ok && a.mask0_off && n < NN && bh[FH_OUT + n])
                xbv[u] = reinterpret_cast<const uint8_t*>(a.ws + a.mask0_off)[relu_byte(n, B, w, wn * 32 + g8)];
        }
#pragma unroll
        for (int u = 0; u < FS_HS; ++u) {
            const int n = 2 * u + wh;
            if (n < NN && bh[FH_OUT + n]) {
                f32x4 y0 = acc[u].c[0], y1 = acc[u].c[1];
                if ((flags & FF_ENC_MASK) && w_ok) {   // layer 0: x relu'(X_0)  (encoder activation)
                    if (a.mask0_off) {      // the encoder's relu byte of this lane
                        const unsigned xb = xbv[u];
#pragma unroll
                        for (int j = 0; j < 4; ++j) { y0[j] = ((xb >> j) & 1u) ? y0[j] : 0.f; y1[j] = ((xb >> (4 + j)) & 1u) ? y1[j] : 0.f; }
                    } else {
                        f32x4 x0, x1;
                        load_oct(xact + act_idx(w, n, B) + col, x0, x1);
#pragma unroll
                        for (int j = 0; j < 4; ++j) { y0[j] = x0[j] > 0.f ? y0[j] : 0.f; y1[j] = x1[j] > 0.f ? y1[j] : 0.f; }
                    }
                }
                if (l > 0) lds_store_oct<T>(smem, n, win, col, y0, y1);
                if (w_ok) store_oct(dxo + act_idx(w, n, B) + col, y0, y1);
            }
        }
        __syncthreads();
        FS_STAMP(6 + 5 * (a.L - 1 - l));
    }
}
template <typename T> __global__ __launch_bounds__(LAYER_THREADS, 2) void k_stack_bwd(StackArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    stack_bwd_body<T, false>(a, smem);
}
// the one-call steps below the slab kernels' batch size: both sweeps of a tile in one launch (k_slab_step's scheme on the 8-wave kernels)
template <typename T> __global__ __launch_bounds__(LAYER_THREADS, 2) void k_stack_step(StackArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    stack_fwd_body<T, true>(a, smem);
    __syncthreads();
    stack_bwd_body<T, true>(a, smem);
}

// ------------------------------------------------------------------------------------------------------
// Slab variant of the backward stack kernel (see k_slab_fwd): wave wn owns columns [32 wn, 32 wn + 32) of every node; the
// nodes whose dX_l is produced are processed in two groups.  A group's accumulators start at the residual term, which is the
// packed dX_{l+1} row this wave produced one layer earlier and kept in registers (for the last layer: the decoder
// backward's dX_L, read once before the loop).
// ------------------------------------------------------------------------------------------------------
template <typename T, int HS, int Q0, bool PRE = false, class A = StackArgs, class FH = FHdr, class FP = FProg>
__device__ __forceinline__ void slab_group_bwd(const A& a, const FH& bh, const FP& wp, char* smem, const T* wpack, int wn, int lane,
                                               int slot_arr, const T* xact, bool enc_mask, u32x4 (&keep)[HS], typename Prec<T>::BFrag (*pre)[2] = nullptr) {
    using P = Prec<T>;
    if (bh[FH_FLAGS] & (Q0 == 0 ? FF_A_EMPTY : FF_B_EMPTY)) {      // no dX row of this group is produced in this layer (uniform)
#pragma unroll
        for (int u = 0; u < HS; ++u) keep[u] = u32x4{0, 0, 0, 0};
        return;
    }
    const int lq = opaque(lane);      // per-node global addresses are rebuilt per call: hoisted out of the layer loop they cost ~40 VGPRs and spill
    const int win = c_win(lq), col = wn * 32 + c_oct(lq), w = min(blockIdx.x * P::ROWS + win, a.B - 1);
    // The accumulators start at the residual term G_{l+1}[n]: `keep` holds it on entry -- the packed dX_{l+1} rows this wave
    // produced one layer earlier (or the decoder backward's dX_L), carried in registers from layer to layer.  (Re-reading them
    // from the stash instead cost 12 us per launch: 70 MB of fabric traffic and an exposed latency per group.)
    typename P::Acc acc[HS];
#pragma unroll
    for (int u = 0; u < HS; ++u) {
        if (bh[FH_KIND + Q0 + u] != NK_DEAD && bh[FH_RES + Q0 + u]) unpack_oct(keep[u], acc[u].c[0], acc[u].c[1]);
        else acc_fill(acc[u], 0.f);
    }
    if constexpr (PRE) fs_run_static<T, HS, (Q0 == 0 ? SL_CBA : SL_CBB), FP, true>(acc, smem, wpack, wn, lane, *pre);
    else fs_run<T, HS, (Q0 == 0 ? SL_CBA : SL_CBB)>(wp, acc, smem, wpack, wn, lane, a.dbg);      // (a.dbg: timing ablations, compiled out of the product build)
    // layer 0: x relu'(X_0) (encoder activation) from the encoder's relu bytes (one per lane, written by k_enc_fwd): a byte per
    // node instead of the 16-byte X_0 octet (38 MB per launch, 4 VGPRs per node), all requested back to back
    const uint8_t* m0 = reinterpret_cast<const uint8_t*>(a.ws + a.mask0_off);
    unsigned xb[HS];
    if (enc_mask) {
#pragma unroll
        for (int u = 0; u < HS; ++u) {
            xb[u] = 0;
            if (bh[FH_OUT + Q0 + u]) {
                if constexpr (FH::is_static) xb[u] = gload1(uniform_ptr(reinterpret_cast<const char*>(m0) + relu_tile_base(bh[slot_arr + u], a.B, blockIdx.x, wn)), (unsigned)lq);
                else xb[u] = m0[relu_tile_base(bh[slot_arr + u], a.B, blockIdx.x, wn) + lq];
            }
        }
    }
#pragma unroll
    for (int u = 0; u < HS; ++u) {
        keep[u] = u32x4{0, 0, 0, 0};
        if (bh[FH_OUT + Q0 + u]) {
            const u32x4 pk = pack_oct(acc[u].c[0], acc[u].c[1]);
            keep[u] = enc_mask ? chunk_mask_bits<T>(pk, xb[u]) : pk;
            pad_valu();
        }
    }
}

// one backward layer of a slab workgroup (FH / FP / mid: see slab_fwd_layer); keepA / keepB carry the packed dX rows from layer to layer
// PRE (compile-time programs): mbq holds this layer's relu bytes (requested before the previous store phase, or the forward's own bits for the last layer) and
// (*pre) the first two weight fragments of the layer's first pass with work; mid() requests the same for the layer that follows
template <typename T, int NM, int HB, bool PRE = false, bool PREM = false, class A, class FH, class FP, class FPB, class Mid>
__device__ __forceinline__ void slab_bwd_layer(const A& a, char* smem, const T* wpack, int wn, int lane, int l, int li, const FH& bh, const FP& wa, const FPB& wb,
                                               u32x4 (&keepA)[SL_HA], u32x4 (&keepB)[HB], Mid&& mid, unsigned* mbq = nullptr, typename Prec<T>::BFrag (*pre)[2] = nullptr) {
    using P = Prec<T>;
    const int tid = threadIdx.x, w0 = blockIdx.x * P::ROWS, B = a.B;
    const int nmlp = bh[FH_NMLP], flags = bh[FH_FLAGS];
    const uint8_t* maskbytes = reinterpret_cast<const uint8_t*>(a.ws + a.mask_off[l]);
    // lane constants rebuilt per layer from an opaque copy of the lane id: the per-node global addresses derived from them
    // would otherwise be hoisted out of the layer loop (dozens of VGPRs, spilled)
    const int lq = opaque(lane);
    const int win = c_win(lq), w = w0 + win, col = wn * 32 + c_oct(lq), wc = min(w, B - 1);
    const bool w_ok = A::full || w < B;

    // mask phase (each lane on the octets it owns): relu nodes are masked in place -> dH_l[n]; group A's accumulators start
    // at the residual term G_{l+1}[n].  Every relu byte and every LDS read is issued before the first use.
    {
        // (the unmasked rows are the carried `keep` registers: the packed dX_{l+1} this wave wrote into the blocks itself)
        unsigned mb[SL_HA + HB];
#pragma unroll
        for (int q = 0; q < SL_HA + HB; ++q) {
            mb[q] = 0xffu;
            if (bh[FH_KIND + q] == NK_RELU) {
                if constexpr (PREM) mb[q] = mbq[q];
                else if constexpr (FH::is_static) mb[q] = gload1(uniform_ptr(reinterpret_cast<const char*>(maskbytes) + relu_tile_base(bh[(q < SL_HA ? FH_SLOTA : FH_SLOTB - SL_HA) + q], B, blockIdx.x, wn)), (unsigned)lq);
                else mb[q] = maskbytes[relu_tile_base(bh[(q < SL_HA ? FH_SLOTA : FH_SLOTB - SL_HA) + q], B, blockIdx.x, wn) + lane];
            }
        }
#pragma unroll
        for (int q = 0; q < SL_HA + HB; ++q) {
            if (bh[FH_KIND + q] == NK_RELU) {
                const u32x4 raw = q < SL_HA ? keepA[q < SL_HA ? q : 0] : keepB[q < SL_HA ? 0 : q - SL_HA];
                *reinterpret_cast<u32x4*>(smem + lds_chunk<T>(bh[(q < SL_HA ? FH_SLOTA : FH_SLOTB - SL_HA) + q], win, col / P::EPC)) = chunk_mask_bits<T>(raw, mb[q]);
            }
        }
    }
    __syncthreads();
    FS_STAMP2(1 + 6 * li);

    if (nmlp > 0) {
        // dT1 = dY W2 ; dU = dT1 . (T1 > 0) ; dH = dU W1     (backward of base_transform, in place on nodes 0..nmlp-1 = the
        // first slots of group B); the dU / dH stashes go out behind the chain's last load
        const T* t1 = reinterpret_cast<const T*>(a.ws + a.t1_off[l]);
        T* du = reinterpret_cast<T*>(a.ws + a.du_off[l]);
        T* dh = reinterpret_cast<T*>(a.ws + a.dh_off[l]);
        // Two nodes at a time (K4 has four base_transform nodes: all four at once need 16 accumulator + 16 staging registers the kernel does not have next
        // to the carried residual rows; each pair pays its own three barriers -- the chain is a few hundred cycles, the spills were round trips)
        typename P::BFrag bf;      // one buffer for both weights: a carried residual (72 VGPRs) lives through this chain
        typename P::AFrag af;
        for (int u0 = 0; u0 < NM; u0 += 2) {
            if (u0 >= nmlp) break;      // (uniform)
            typename P::Acc tm[2];
            u32x4 traw[2], dupk[2];
            load_bfrag<T>(bf, wpack, bh[FH_W2], wn, lane);
#pragma unroll
            for (int v = 0; v < 2; ++v) {
                const int u = u0 + v;
                traw[v] = u32x4{0, 0, 0, 0};
                if (u < nmlp) {
                    if constexpr (FH::is_static) traw[v] = gload16(uniform_ptr(reinterpret_cast<const char*>(t1 + act_idx(0, u, B))), (unsigned)((wc * H + col) * (int)sizeof(T)));
                    else traw[v] = *reinterpret_cast<const u32x4*>(t1 + act_idx(wc, u, B) + col);
                    acc_fill(tm[v], 0.f);
                    load_afrag<T>(af, smem, u, lane);
                    mac(tm[v], af, bf);
                }
            }
            load_bfrag<T>(bf, wpack, bh[FH_W1], wn, lane);
            __syncthreads();   // all reads of the dY blocks done
#pragma unroll
            for (int v = 0; v < 2; ++v) {
                const int u = u0 + v;
                dupk[v] = u32x4{0, 0, 0, 0};
                if (u < nmlp) {
                    f32x4 t0, t1v, r0, r1; unpack_oct(traw[v], t0, t1v);
#pragma unroll
                    for (int j = 0; j < 4; ++j) { r0[j] = t0[j] > 0.f ? tm[v].c[0][j] : 0.f; r1[j] = t1v[j] > 0.f ? tm[v].c[1][j] : 0.f; }
                    dupk[v] = pack_oct(r0, r1);
                    *reinterpret_cast<u32x4*>(smem + lds_chunk<T>(u, win, col / P::EPC)) = dupk[v];
                }
            }
            __syncthreads();
#pragma unroll
            for (int v = 0; v < 2; ++v) {
                const int u = u0 + v;
                if (u < nmlp) { acc_fill(tm[v], 0.f); load_afrag<T>(af, smem, u, lane); mac(tm[v], af, bf); }
            }
            __syncthreads();   // all reads of the dU blocks done
#pragma unroll
            for (int v = 0; v < 2; ++v) {
                const int u = u0 + v;
                if (u < nmlp) {
                    const u32x4 hp = pack_oct(tm[v].c[0], tm[v].c[1]);
                    *reinterpret_cast<u32x4*>(smem + lds_chunk<T>(u, win, col / P::EPC)) = hp;
                    if (w_ok) {
                        if constexpr (FH::is_static) {
                            gstore16(uniform_wptr(reinterpret_cast<char*>(du + act_idx(0, u, B))), (unsigned)((w * H + col) * (int)sizeof(T)), dupk[v]);
                            gstore16(uniform_wptr(reinterpret_cast<char*>(dh + act_idx(0, u, B))), (unsigned)((w * H + col) * (int)sizeof(T)), hp);
                        } else {
                            *reinterpret_cast<u32x4*>(du + act_idx(w, u, B) + col) = dupk[v];
                            *reinterpret_cast<u32x4*>(dh + act_idx(w, u, B) + col) = hp;
                        }
                    }
                }
            }
        }
        __syncthreads();
    }

    // dX_l[j] = (residual) + dH_j W_rootsum + sum_r sum_{j->i} dH_i W_rel^r, group A then group B
    const T* xact = reinterpret_cast<const T*>(a.ws + a.x_off[0]);
    const bool enc_mask = (flags & FF_ENC_MASK) != 0;
    FS_STAMP2(2 + 6 * li);
    if constexpr (PRE) {
        constexpr bool a_first = !(FH{}[FH_FLAGS] & FF_A_EMPTY);
        slab_group_bwd<T, SL_HA, 0, a_first>(a, bh, wa, smem, wpack, wn, lane, FH_SLOTA, xact, enc_mask, keepA, pre);
        FS_STAMP2(3 + 6 * li);
        slab_group_bwd<T, HB, SL_HA, !a_first>(a, bh, wb, smem, wpack, wn, lane, FH_SLOTB, xact, enc_mask, keepB, pre);
    } else {
        slab_group_bwd<T, SL_HA, 0>(a, bh, wa, smem, wpack, wn, lane, FH_SLOTA, xact, enc_mask, keepA);
        FS_STAMP2(3 + 6 * li);
        slab_group_bwd<T, HB, SL_HA>(a, bh, wb, smem, wpack, wn, lane, FH_SLOTB, xact, enc_mask, keepB);
    }
    FS_STAMP2(4 + 6 * li);
    __syncthreads();   // every wave is done reading dH_l
    FS_STAMP2(5 + 6 * li);
    mid();      // next header / programs landed before the stores go out (FProg::settle)
    T* dxo = reinterpret_cast<T*>(a.ws + a.dx_off[l]);
    if constexpr (FH::is_static) {      // (as slab_group_store: LDS rows first, then every stash row under one lane predicate)
        if (l > 0) {
#pragma unroll
            for (int q = 0; q < SL_HA + HB; ++q)
                if (bh[FH_OUT + q])
                    *reinterpret_cast<u32x4*>(smem + lds_chunk<T>(bh[(q < SL_HA ? FH_SLOTA : FH_SLOTB - SL_HA) + q], win, col / P::EPC)) = q < SL_HA ? keepA[q < SL_HA ? q : 0] : keepB[q < SL_HA ? 0 : q - SL_HA];
        }
        if (w_ok) {
#ifdef MSHGNN_STASH_ALIAS
            const unsigned soff = (unsigned)((((blockIdx.x & 31) * P::ROWS + win) * H + col) * (int)sizeof(T));
#else
            const unsigned soff = (unsigned)((w * H + col) * (int)sizeof(T));
#endif
#pragma unroll
            for (int q = 0; q < SL_HA + HB; ++q)
                if (bh[FH_OUT + q])
                    stash_store_u(a, uniform_wptr(reinterpret_cast<char*>(dxo + act_idx(0, bh[(q < SL_HA ? FH_SLOTA : FH_SLOTB - SL_HA) + q], B))), soff, q < SL_HA ? keepA[q < SL_HA ? q : 0] : keepB[q < SL_HA ? 0 : q - SL_HA]);
        }
    } else {
#pragma unroll
        for (int q = 0; q < SL_HA + HB; ++q) {
            if (bh[FH_OUT + q]) {
                const int n = bh[(q < SL_HA ? FH_SLOTA : FH_SLOTB - SL_HA) + q];
                const u32x4 v = q < SL_HA ? keepA[q < SL_HA ? q : 0] : keepB[q < SL_HA ? 0 : q - SL_HA];
                if (l > 0) *reinterpret_cast<u32x4*>(smem + lds_chunk<T>(n, win, col / P::EPC)) = v;
                if (w_ok) stash_store(dxo + act_idx(w, n, B) + col, v, a.stash_nt != 0);
            }
        }
    }
    __syncthreads();
    FS_STAMP2(6 + 6 * li);
}
// the layers of a compile-time program SP, last to first, unrolled
template <typename T, int NM, int HB, class SP, int l, class A>
__device__ __forceinline__ void slab_bwd_layers_static(const A& a, char* smem, const T* wpack, int wn, int lane, u32x4 (&keepA)[SL_HA], u32x4 (&keepB)[HB],
                                                       unsigned (&mbq)[SL_HA + HB], typename Prec<T>::BFrag (&pre)[2]) {
    if constexpr (l >= 0) {
        auto mid = [&] {      // before layer l's stores: layer l - 1's relu bytes (written by the forward sweep long ago) and first weight fragments
            if constexpr (l > 0) {
                if constexpr ((SP::PRE & 2) != 0) {
                    constexpr SHdr<SP, 1, l - 1> nh{};
                    const uint8_t* mbn = reinterpret_cast<const uint8_t*>(a.ws + a.mask_off[l - 1]);
#pragma unroll
                    for (int q = 0; q < SL_HA + HB; ++q) {
                        mbq[q] = 0xffu;
                        if (nh[FH_KIND + q] == NK_RELU) mbq[q] = gload1(uniform_ptr(reinterpret_cast<const char*>(mbn) + relu_tile_base(nh[(q < SL_HA ? FH_SLOTA : FH_SLOTB - SL_HA) + q], a.B, blockIdx.x, wn)), (unsigned)lane);
                    }
                }
                if constexpr ((SP::PRE & 1) != 0) slab_prefetch_static<T, SP, 1, l - 1>(pre, wpack, wn, lane);
            }
        };
        // (the last layer's bytes are the forward's own bits whatever SP::PRE says)
        slab_bwd_layer<T, NM, HB, (SP::PRE & 1) != 0, (SP::PRE & 2) != 0 || l == SP::L - 1>(a, smem, wpack, wn, lane, l, SP::L - 1 - l, SHdr<SP, 1, l>{}, SProg<SP, 1, l, 0>{}, SProg<SP, 1, l, 1>{},
                                                                                              keepA, keepB, mid, mbq, &pre);
        slab_bwd_layers_static<T, NM, HB, SP, l - 1>(a, smem, wpack, wn, lane, keepA, keepB, mbq, pre);
    }
}
// the last layer's relu bytes of a one-launch step are the forward's own bits (same wave, same lane, same slots): no load
template <class SP, int HB> __device__ __forceinline__ void slab_bits_to_bytes(const unsigned* lastbits, unsigned (&mbq)[SL_HA + HB]) {
    constexpr SHdr<SP, 1, SP::L - 1> bh{};
    constexpr SHdr<SP, 0, SP::L - 1> fh{};
#pragma unroll
    for (int q = 0; q < SL_HA + HB; ++q) {
        mbq[q] = 0xffu;
        if (bh[FH_KIND + q] == NK_RELU) {
            const int u = q < SL_HA ? q : q - SL_HA;
            const unsigned word = lastbits[(q < SL_HA ? 0 : (SL_HA + 3) / 4) + (u >> 2)];
            mbq[q] = (word >> (8 * (u & 3))) & 0xffu;
        }
    }
}
template <class SP, int HB> constexpr bool slab_bits_consistent() {      // a node masked by the backward's last layer was a relu node of the forward's last layer, in the same slot
    for (int q = 0; q < SL_HA + HB; ++q)
        if (SP::bwd[SP::L - 1][FH_KIND + q] == NK_RELU && SP::fwd[SP::L - 1][FH_KIND + q] != NK_RELU) return false;
    for (int u = 0; u < 16; ++u)
        if (SP::bwd[SP::L - 1][FH_SLOTA + u] != SP::fwd[SP::L - 1][FH_SLOTA + u] || SP::bwd[SP::L - 1][FH_SLOTB + u] != SP::fwd[SP::L - 1][FH_SLOTB + u]) return false;
    return true;
}
// the carried rows at the start of the sweep: the decoder backward's dX_L, read back from the tile in LDS (rows past the batch are zero there)
template <typename T, int HB, class FH>
__device__ __forceinline__ void slab_bwd_keep_init(const FH& bh, const char* smem, int wn, int lane, u32x4 (&keepA)[SL_HA], u32x4 (&keepB)[HB]) {
    using P = Prec<T>;
    const int loffq = lds_chunk<T>(0, c_win(lane), (wn * 32 + c_oct(lane)) / P::EPC);
#pragma unroll
    for (int u = 0; u < SL_HA; ++u) {
        keepA[u] = u32x4{0, 0, 0, 0};
        const int n = bh[FH_SLOTA + u];
        if (n >= 0 && bh[FH_KIND + u] != NK_DEAD) keepA[u] = *reinterpret_cast<const u32x4*>(smem + n * P::BLK + loffq);
    }
#pragma unroll
    for (int u = 0; u < HB; ++u) {
        keepB[u] = u32x4{0, 0, 0, 0};
        const int n = bh[FH_SLOTB + u];
        if (n >= 0 && bh[FH_KIND + SL_HA + u] != NK_DEAD) keepB[u] = *reinterpret_cast<const u32x4*>(smem + n * P::BLK + loffq);
    }
}

// STEP: part of k_slab_step -- the forward's decoder tail of the same launch left the dX_L tile in LDS; the layers' programs are a.prog_off_b
// SP: void = the plan's tables are interpreted; else the compile-time program of one (topology, depth)
template <typename T, int NM, int HB, bool STEP, class SP = void, class A = StackArgs> __device__ __forceinline__ void slab_bwd_body(const A& a, char* smem, const unsigned* lastbits = nullptr) {
    using P = Prec<T>;
    constexpr bool DYN = std::is_void<SP>::value;
    const int tid = threadIdx.x, lane = tid & 63, wn = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int w0 = blockIdx.x * P::ROWS, B = a.B, NN = a.NN;
    const T* wpack = reinterpret_cast<const T*>(a.wpack);
    static_assert(sizeof(T) == 2, "fused stack kernels are bf16");
    auto prog_of = [&](int l) { return STEP ? a.prog_off_b[l] : a.prog_off[l]; };
    if constexpr (!STEP) stack_stagger(a);

    // the last layer's header and programs stream in under the tile load
    FHdr bhn; FProg wan, wbn;
    if constexpr (DYN) {
        bhn = FHdr(a.tables + prog_of(a.L - 1), lane);
        wan = FProg(a.tables + prog_of(a.L - 1) + FH_SIZE, lane); wbn = FProg(a.tables + prog_of(a.L - 1) + FH_SIZE + FPROG_LEN, lane);
    }
    // dX_L tile: only the nodes that are live in the last layer carry a gradient -- the output type's [node0, node0 + n_out), known from the arguments, so
    // the loads do not wait for the header; four nodes per round trip
    if constexpr (!STEP) {
        const T* src = reinterpret_cast<const T*>(a.tile_in);
        const int row = tid >> 4, c = tid & 15;
        for (int f0 = 0; f0 < a.n_out; f0 += 4) {
            u32x4 v[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                v[i] = u32x4{0, 0, 0, 0};
                if (f0 + i < a.n_out && w0 + row < B) v[i] = *reinterpret_cast<const u32x4*>(src + act_idx(w0 + row, a.node0 + f0 + i, B) + c * P::EPC);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (f0 + i < a.n_out) *reinterpret_cast<u32x4*>(smem + lds_chunk<T>(a.node0 + f0 + i, row, c)) = v[i];
        }
    }
    __syncthreads();

    // group A's residual term travels from layer to layer in registers (the packed dX rows of the previous epilogue); for the
    // last layer it is the decoder backward's dX_L
    u32x4 keepA[SL_HA], keepB[HB];
    if constexpr (DYN) {
        slab_bwd_keep_init<T, HB>(bhn, smem, wn, lane, keepA, keepB);
        bhn.settle(); wan.settle(); wbn.settle();
        FS_STAMP2(0);
        for (int l = a.L - 1; l >= 0; --l) {
            const FHdr bh = bhn;
            const FProg wa = wan, wb = wbn;
            if (l > 0) {
                bhn = FHdr(a.tables + prog_of(l - 1), lane);
                wan = FProg(a.tables + prog_of(l - 1) + FH_SIZE, lane);
                wbn = FProg(a.tables + prog_of(l - 1) + FH_SIZE + FPROG_LEN, lane);
            }
            slab_bwd_layer<T, NM, HB>(a, smem, wpack, wn, lane, l, a.L - 1 - l, bh, wa, wb, keepA, keepB, [&] { bhn.settle(); wan.settle(); wbn.settle(); });
        }
    } else {
        static_assert(slab_bits_consistent<SP, HB>(), "forward / backward tables of the last layer disagree");
        typename P::BFrag pre[2];
        if constexpr ((SP::PRE & 1) != 0) slab_prefetch_static<T, SP, 1, SP::L - 1>(pre, wpack, wn, lane);      // (nothing is stored between here and the first pass: the mask phase writes LDS only)
        slab_bwd_keep_init<T, HB>(SHdr<SP, 1, SP::L - 1>{}, smem, wn, lane, keepA, keepB);
        unsigned mbq[SL_HA + HB];
        if constexpr (STEP) slab_bits_to_bytes<SP, HB>(lastbits, mbq);
        else {      // the backward launch alone: the last layer's relu bytes come from the forward launch's stash, like every other layer's
            constexpr SHdr<SP, 1, SP::L - 1> nh{};
            const uint8_t* mbn = reinterpret_cast<const uint8_t*>(a.ws + a.mask_off[SP::L - 1]);
#pragma unroll
            for (int q = 0; q < SL_HA + HB; ++q) {
                mbq[q] = 0xffu;
                if (nh[FH_KIND + q] == NK_RELU) mbq[q] = gload1(uniform_ptr(reinterpret_cast<const char*>(mbn) + relu_tile_base(nh[(q < SL_HA ? FH_SLOTA : FH_SLOTB - SL_HA) + q], a.B, blockIdx.x, wn)), (unsigned)lane);
            }
        }
        FS_STAMP2(0);
        slab_bwd_layers_static<T, NM, HB, SP, SP::L - 1>(a, smem, wpack, wn, lane, keepA, keepB, mbq, pre);
    }
}
template <typename T, int NM, int HB> __global__ __launch_bounds__(SL_THREADS, 2) void k_slab_bwd(StackArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    slab_bwd_body<T, NM, HB, false>(a, smem);
}
// The backward launch alone (second call of the two-call training route: mshgnn_backward / _mse / _ce after mshgnn_forward(training = 1)) over the compile-time programs:
// whole tiles, NT = the stash store policy.  Same MACs, same order as k_slab_bwd: identical bits.
template <typename T, int NM, int HB, class SP, int NT> __global__ __launch_bounds__(SL_THREADS, 2) void k_slab_bwd_spec(StackArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const StackView<NT> v(a, true);
    slab_bwd_body<T, NM, HB, false, SP>(v, smem);
}
// One-call training step (mshgnn_step_mse / mshgnn_step_ce): the forward layers, decoder + loss + decoder backward and the backward layers of a tile in ONE
// launch.  The tail leaves dX_L in the node blocks, so the backward sweep starts without a launch boundary, without the header / tile round trips of
// k_slab_bwd's start and without re-reading dX_L (stamps: 16 k of its 163 k cycles).  Same code, same order of every accumulation: identical bits.
template <typename T, int NM, int HB, class SP = void, int NT = 0, bool FULL = true> __global__ __launch_bounds__(SL_THREADS, 2) void k_slab_step(StackArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    if constexpr (!std::is_void<SP>::value) {      // compile-time program; NT: the stash store policy; FULL: whole tiles only, unpredicated stores (else any batch size)
        const StackView<NT, FULL> v(a, true);
        unsigned lastbits[(SL_HA + 3) / 4 + (HB + 3) / 4];
        slab_fwd_body<T, NM, HB, true, SP>(v, smem, lastbits);
        __syncthreads();
        slab_bwd_body<T, NM, HB, true, SP>(v, smem, lastbits);
    } else {
        slab_fwd_body<T, NM, HB, true, SP>(a, smem);
        __syncthreads();      // the tile's dX_L rows are in the out-type blocks, the tail's reduction scratch has been read
        slab_bwd_body<T, NM, HB, true, SP>(a, smem);
    }
}

// the slab instantiation of a plan: NM = bound on the base_transform nodes (2 / 4), HB = group-B slots (6 / 8)
using StackKernel = void (*)(StackArgs);
// Specialised step kernels: k_slab_step over the COMPILE-TIME program of one (topology, depth) -- mshgnn_spec_tables.inc, generated from this library's own plan
// compiler by tools/gen_spec_tables.py.  A plan takes one only when its slab tables are exactly the ints the kernel was compiled from (same packs, same
// slots, same liveness), so a stale table file costs speed, never results; MSHGNN_SPEC=0 keeps the interpreting kernel (A/B runs, bit-identity tests).
#include "mshgnn_spec_tables.inc"
template <class SP> static bool spec_matches(const HostPlan& hp) {
    if (!hp.slab || hp.L != SP::L || hp.NN != SP::NN || hp.sl_hb != SP::HB || (hp.n_mlp <= 2 ? 2 : 4) != SP::NM || (hp.d.out_channels <= 4 ? 4 : 8) != SP::DMAX) return false;
    for (int l = 0; l < SP::L; ++l) {
        if (hp.sl_fwd_off[l] + SP::ROW > (int)hp.tables.size() || hp.sl_bwd_off[l] + SP::ROW > (int)hp.tables.size()) return false;
        if (memcmp(hp.tables.data() + hp.sl_fwd_off[l], SP::fwd[l], sizeof(int32_t) * SP::ROW) != 0) return false;
        if (memcmp(hp.tables.data() + hp.sl_bwd_off[l], SP::bwd[l], sizeof(int32_t) * SP::ROW) != 0) return false;
    }
    return true;
}
// The forward launch alone (mshgnn_forward: evaluation, or the first call of the two-call training route) over the same compile-time programs: whole tiles; TR (training:
// stashes and relu bytes written) and NT (their store policy) are template parameters like the step kernels' NT -- with them at run time the 8-layer programs keep their
// store addresses live across the unrolled layers and spill (256-492 B of scratch: slower than the interpreter at 8 192 windows).  The decoder tail runs without the fused
// loss.  Same MACs, same order: the interpreter's bits.
template <typename T, int NM, int HB, class SP, int TR, int NT, bool FULL = true> __global__ __launch_bounds__(SL_THREADS, 2) void k_slab_fwd_spec(StackArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    StackView<NT, FULL> v(a, false);
    v.training = TR;
    unsigned lastbits[(SL_HA + 3) / 4 + (HB + 3) / 4];
    slab_fwd_body<T, NM, HB, false, SP>(v, smem, lastbits);
}
// The kernels over the compile-time programs are instantiated in their own translation units -- this source compiled with -DMSHGNN_SPEC_SHARD=1..7 (csrc/Makefile:
// mshgnn_spec<k>.o), one program each (MSHGNN_SPEC_LIST_<k>), side by side with the rest of the library (shard 0: everything else).  A shard exports one
// selector: kind 0 = one-call step, 1 = forward alone (tr: training), 2 = backward alone; nt = the launch's stash store policy (stash_nt_for); name: the program's name.
#define SPEC_SHARD_LIST(X) X(1) X(2) X(3) X(4) X(5) X(6) X(7)      // one program per shard (tools/gen_spec_tables.py SHARDS)
#define SPEC_SHARD_DECL(k) StackKernel spec_shard##k(const HostPlan& hp, int kind, int tr, int nt, int full, const char** name);
SPEC_SHARD_LIST(SPEC_SHARD_DECL)
// full (the batch is whole 16-window tiles): the unpredicated kernels, both store policies.  Else: the predicated forms that exist -- the one-call step with plain stash stores
// (a ragged batch whose stash wants non-temporal stores keeps the interpreter: the weight-gradient launch behind plain stores loses more than the program wins at 8 layers)
// and the evaluation forward; the two-call training route of a ragged batch keeps the interpreters (nullptr).
#if MSHGNN_SPEC_SHARD != 0
#define MSHGNN_SPEC_TRY(SP) if (spec_matches<SP>(hp)) { \
        if (name) *name = #SP; \
        if (kind == 0 && !full) return nt ? nullptr : k_slab_step<__bf16, SP::NM, SP::HB, SP, 0, false>; \
        if (kind == 0) return nt ? k_slab_step<__bf16, SP::NM, SP::HB, SP, 1> : k_slab_step<__bf16, SP::NM, SP::HB, SP, 0>; \
        if (kind == 1 && !tr) return full ? k_slab_fwd_spec<__bf16, SP::NM, SP::HB, SP, 0, 0> : k_slab_fwd_spec<__bf16, SP::NM, SP::HB, SP, 0, 0, false>; \
        if (!full) return nullptr; \
        if (kind == 1) return nt ? k_slab_fwd_spec<__bf16, SP::NM, SP::HB, SP, 1, 1> : k_slab_fwd_spec<__bf16, SP::NM, SP::HB, SP, 1, 0>; \
        return nt ? k_slab_bwd_spec<__bf16, SP::NM, SP::HB, SP, 1> : k_slab_bwd_spec<__bf16, SP::NM, SP::HB, SP, 0>; }
#define SPEC_CAT2(a, b) a##b
#define SPEC_CAT(a, b) SPEC_CAT2(a, b)
#if MSHGNN_SPEC_SHARD == 99
// Shard 99 is not part of the library: it is this source compiled AFTER the build, for one plan's own tables (morphsym_hgnn_amd/jit.py renders them as MSHGNN_JIT_TABLES -- a
// struct spec::JIT_<hash> and MSHGNN_SPEC_LIST_99 -- and compiles a small shared library of the program's kernels), for topologies the build has no program for.  The library
// takes its selector through mshgnn_plan_attach_program and checks the tables like every shard does.
#include MSHGNN_JIT_TABLES
#endif
StackKernel SPEC_CAT(spec_shard, MSHGNN_SPEC_SHARD)(const HostPlan& hp, int kind, int tr, int nt, int full, const char** name) { SPEC_CAT(MSHGNN_SPEC_LIST_, MSHGNN_SPEC_SHARD)(MSHGNN_SPEC_TRY) return nullptr; }
#if MSHGNN_SPEC_SHARD == 99
extern "C" StackKernel mshgnn_jit_program(const HostPlan& hp, int kind, int tr, int nt, int full, const char** name) { return spec_shard99(hp, kind, tr, nt, full, name); }
#endif
#undef MSHGNN_SPEC_TRY
#else      // MSHGNN_SPEC_SHARD == 0: the library proper, to the end of this file
static StackKernel slab_fwd_kernel(const HostPlan& hp) {
    if (hp.sl_hb <= SL_HB) return hp.n_mlp <= 2 ? k_slab_fwd<__bf16, 2, SL_HB> : k_slab_fwd<__bf16, 4, SL_HB>;
    return hp.n_mlp <= 2 ? k_slab_fwd<__bf16, 2, SL_HB_MAX> : k_slab_fwd<__bf16, 4, SL_HB_MAX>;
}
using SpecSelector = StackKernel (*)(const HostPlan& hp, int kind, int tr, int nt, int full, const char** name);
static StackKernel spec_kernel(const HostPlan& hp, int kind, int tr, int nt, int full, const char** name = nullptr) {
    const char* nm = nullptr;      // (a shard that holds the plan's program sets the name even where it has no kernel of that kind: stop there)
    if (hp.jit_prog) {             // a program compiled for this plan after the build
        StackKernel kk = reinterpret_cast<SpecSelector>(hp.jit_prog)(hp, kind, tr, nt, full, &nm);
        if (nm) { if (name) *name = nm; return kk; }
    }
#define SPEC_SHARD_TRY(k) { StackKernel kk = spec_shard##k(hp, kind, tr, nt, full, &nm); if (nm) { if (name) *name = nm; return kk; } }
    SPEC_SHARD_LIST(SPEC_SHARD_TRY)
#undef SPEC_SHARD_TRY
    return nullptr;
}
inline bool whole_tiles(int64_t B) { return B > 0 && B % TILE_ROWS == 0; }
static StackKernel slab_step_spec_kernel(const HostPlan& hp, int nt, const char** name = nullptr, bool full = true) { return spec_kernel(hp, 0, 1, nt, full, name); }
static StackKernel slab_fwd_spec_kernel(const HostPlan& hp, int training, int nt, bool full = true) { return spec_kernel(hp, 1, training, nt, full); }
static StackKernel slab_bwd_spec_kernel(const HostPlan& hp, int nt, bool full = true) { return spec_kernel(hp, 2, 1, nt, full); }
// the step kernel of a launch: the specialised one where the plan has one and the batch is whole tiles (its stores are unpredicated), else the interpreter
static StackKernel slab_step_kernel(const HostPlan& hp, int64_t B = -1, int nt = 0, bool use_spec = false) {
    if (use_spec && B > 0) if (StackKernel k = slab_step_spec_kernel(hp, nt, nullptr, whole_tiles(B))) return k;
    if (hp.sl_hb <= SL_HB) return hp.n_mlp <= 2 ? k_slab_step<__bf16, 2, SL_HB> : k_slab_step<__bf16, 4, SL_HB>;
    return hp.n_mlp <= 2 ? k_slab_step<__bf16, 2, SL_HB_MAX> : k_slab_step<__bf16, 4, SL_HB_MAX>;
}
static StackKernel slab_bwd_kernel(const HostPlan& hp) {
    if (hp.sl_hb <= SL_HB) return hp.n_mlp <= 2 ? k_slab_bwd<__bf16, 2, SL_HB> : k_slab_bwd<__bf16, 4, SL_HB>;
    return hp.n_mlp <= 2 ? k_slab_bwd<__bf16, 2, SL_HB_MAX> : k_slab_bwd<__bf16, 4, SL_HB_MAX>;
}

// ------------------------------------------------------------------------------------------------------
// decoder forward / backward (hgnn_c2.py:176-189) and the wrapper's MSE (gnnLightning.py:633-639)
// ------------------------------------------------------------------------------------------------------
template <typename T> __global__ __launch_bounds__(256) void k_dec_fwd(DecArgs a) {
    const int c = threadIdx.x & 15;
    const int64_t row = (int64_t)blockIdx.x * 16 + (threadIdx.x >> 4);
    const int64_t rows = (int64_t)a.B * a.n_out;
    const bool ok = row < rows;
    const int w = ok ? (int)(row / a.n_out) : 0, f = ok ? (int)(row % a.n_out) : 0;
    float x[8];
    load8<T>(reinterpret_cast<const T*>(a.xl) + act_idx(w, a.node0 + f, a.B) + c * 8, x);
    const float* W = a.params + a.off_w;
    for (int d = 0; d < a.dout; ++d) {
        float s = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) s += x[e] * W[d * H + c * 8 + e];
#pragma unroll
        for (int m = 8; m >= 1; m >>= 1) s += __shfl_xor(s, m, 64);
        if (c == 0 && ok) a.out[row * a.dout + d] = (s + a.params[a.off_b + d]) * a.out_mask[f * a.dout + d];
    }
}

template <typename T> __global__ __launch_bounds__(256) void k_dec_bwd(DecArgs a) {
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
        const size_t idx = act_idx(w, a.node0 + f, a.B) + c * 8;
        float x[8], dx[8];
        load8<T>(xl + idx, x);
#pragma unroll
        for (int e = 0; e < 8; ++e) dx[e] = 0.f;
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
        store8<T>(dxl + idx, dx);
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

__global__ void k_mse(const float* out, const float* y, int64_t n, float* loss, float* gout) {
    __shared__ float red[4];
    float s = 0.f;
    const float inv = 1.0f / (float)n;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float dlt = out[i] - y[i];
        s += dlt * dlt;
        if (gout) gout[i] = 2.0f * dlt * inv;
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) s += __shfl_xor(s, m, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(loss, (red[0] + red[1] + red[2] + red[3]) * inv);
}

// ------------------------------------------------------------------------------------------------------
// k_gradw: all weight gradients of the step as one split-K MFMA launch.  dW[o][k] = sum_w P[w][o] Q[w][k]
// workgroup = (lane, part): the lane's <= GW_IPL items over window part `part`; slab index = blockIdx.x
// ------------------------------------------------------------------------------------------------------
constexpr int GW_KW = 32;       // fp32: windows per staged chunk
constexpr int GW_PITCH = 144;   // fp32: floats per LDS row (bank-conflict-free column reads)

__global__ __launch_bounds__(256) void k_gradw_f32(GradwArgs a) {
    using T = float;
    __shared__ __attribute__((aligned(16))) float Ps[GW_KW * GW_PITCH];
    __shared__ __attribute__((aligned(16))) float Qs[GW_KW * GW_PITCH];
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wv >> 1, wc = wv & 1;
    // blocks b and b+8 share an XCD (round-robin dispatch; speed only): lane_order puts the lanes that read the same
    // dH / X rows on one XCD so they share its L2
    const int ln = a.lane_order[blockIdx.x % a.n_pad], part = blockIdx.x / a.n_pad;
    if (ln < 0) return;
    const int* lh = a.lanes + ln * LANE_INTS;
    const int it0 = lh[0], it1 = lh[1], bias_flag = lh[3];
    const int nchunks = (a.B + GW_KW - 1) / GW_KW;
    const int ch0 = (int)((int64_t)part * nchunks / a.n_parts), ch1 = (int)((int64_t)(part + 1) * nchunks / a.n_parts);
    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0, 0, 0, 0};
    f32x4 bsum = f32x4{0, 0, 0, 0};
    const int c = tid & 31, r0 = tid >> 5;   // staging: 32 chunks of 4 floats per row, 8 rows per pass
    for (int ch = ch0; ch < ch1; ++ch) {
        const int w0 = ch * GW_KW;
        for (int it = it0; it < it1; ++it) {
            const int* im = a.items + it * ITEM_INTS;
            const T* pb = reinterpret_cast<const T*>(a.ws + a.buf_off[im[0]]);
            const int ps = im[1], po = im[2], qbuf = im[3], qs = im[4], qo = im[5], qc0 = im[6], qn = im[7], so = im[8];
            f32x4 pv[4], qv[4];
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                const int row = r0 + 8 * p, w = w0 + row;
                pv[p] = f32x4{0, 0, 0, 0}; qv[p] = f32x4{0, 0, 0, 0};
                if (w < a.B) {
                    pv[p] = *reinterpret_cast<const f32x4*>(pb + act_idx(w, po, a.B) + c * 4);
                    if (im[9] >= 0) {   // P = dX_{l+1} . relu bits
                        const unsigned word = reinterpret_cast<const uint8_t*>(a.ws + a.buf_off[im[9]])[relu_byte(po, a.B, w, c * 4)];
                        pv[p] = __builtin_bit_cast(f32x4, chunk_mask_bits<float>(__builtin_bit_cast(u32x4, pv[p]), word >> ((c * 4) % 8)));
                    }
                    if (qs >= 0) {
                        qv[p] = *reinterpret_cast<const f32x4*>(reinterpret_cast<const T*>(a.ws + a.buf_off[qbuf]) + act_idx(w, qo, a.B) + c * 4);
                    } else {
                        const int t = qbuf - BUF_IN;
                        const T* qq = reinterpret_cast<const T*>(a.x[t]) + ((size_t)w * a.nodes[t] + qo) * a.pitch[t] + qc0 + c * 4;
                        const u32x4 raw = load_chunk<T>(qq, qn - c * 4, a.vb[t]) ^ sign_xor<T>(a.signs + so + c * 4);
                        qv[p] = __builtin_bit_cast(f32x4, raw);
                    }
                }
            }
            __syncthreads();   // previous MFMA phase finished reading Ps/Qs
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                const int row = r0 + 8 * p;
                *reinterpret_cast<f32x4*>(&Ps[row * GW_PITCH + c * 4]) = pv[p];
                *reinterpret_cast<f32x4*>(&Qs[row * GW_PITCH + c * 4]) = qv[p];
                if (bias_flag) bsum += pv[p];
            }
            __syncthreads();
#pragma unroll
            for (int ks = 0; ks < GW_KW / 4; ++ks) {
                const int row = 4 * ks + (lane >> 4);
                float af[4], bq[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    af[i] = Ps[row * GW_PITCH + wr * 64 + i * 16 + (lane & 15)];
                    bq[i] = Qs[row * GW_PITCH + wc * 64 + i * 16 + (lane & 15)];
                }
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i], bq[j], acc[i][j], 0, 0, 0);
            }
        }
    }
    float* slab = a.slabs + (size_t)(part * a.n_lanes + ln) * SLAB_FLOATS;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int o = wr * 64 + i * 16 + ((lane >> 4) << 2) + q, k = wc * 64 + j * 16 + (lane & 15);
                slab[o * H + k] = acc[i][j][q];
            }
    if (bias_flag) {
        __syncthreads();
        *reinterpret_cast<f32x4*>(&Ps[r0 * GW_PITCH + c * 4]) = bsum;
        __syncthreads();
        if (tid < H) {
            float s = 0.f;
#pragma unroll
            for (int r = 0; r < 8; ++r) s += Ps[r * GW_PITCH + tid];
            slab[H * H + tid] = s;
        }
    }
}

#ifndef GW_WPS
#define GW_WPS 3
#endif
template <int IPL> __global__ __launch_bounds__(256, GW_WPS) void k_gradw_bf16(GradwArgs a) {
    using T = __bf16;
    constexpr int I1 = IPL - 1;       // index of a lane's second item (== 0 with one item per lane)
    // two LDS stages: the staging writes of step s+1 go to the other stage while step s's MFMAs read this one -> one
    // barrier per step and LDS writes overlap the MFMAs
#ifndef GW_DEEP
#define GW_DEEP 0   // measured: the shallow pipeline at 3 workgroups/CU (132 us) beats the deep one at 2 (153 us)
#endif
    __shared__ __attribute__((aligned(16))) __bf16 Pbuf[1 + GW_DEEP][GWB_KW * GWB_PITCH];
    __shared__ __attribute__((aligned(16))) __bf16 Qbuf[1 + GW_DEEP][GWB_KW * GWB_PITCH];
    __shared__ __attribute__((aligned(16))) u32x4 mlut[256];       // relu byte -> AND mask of 8 bf16
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    mlut[tid] = chunk_mask_bits<__bf16>(u32x4{0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu}, (unsigned)tid);   // (256 threads; visible after the first barrier)
    const int wr = wv >> 1, wc = wv & 1;
    // blocks b and b+8 share an XCD (round-robin dispatch; speed only): lane_order puts the lanes that read the same
    // dH / X rows on one XCD so they share its L2
    const int ln = a.lane_order[blockIdx.x % a.n_pad], part = blockIdx.x / a.n_pad;
    if (ln < 0) return;
    const int* lh = a.lanes + ln * LANE_INTS;
    const int it0 = lh[0], nit = lh[1] - lh[0], bias_flag = lh[3];
    const int nchunks = (a.B + GWB_KW - 1) / GWB_KW;
    const int ch0 = (int)((int64_t)part * nchunks / a.n_parts), ch1 = (int)((int64_t)(part + 1) * nchunks / a.n_parts);
    const int nsteps = (ch1 - ch0) * nit;    // step s -> chunk ch0 + s / nit, item s % nit
    const int c = tid & 15, r0 = tid >> 4;   // staging: 16 chunks of 8 bf16 per row, 16 rows per pass

    // the lane's (<= IPL) items are resolved ONCE into per-thread base pointers: no dependent scalar loads
    // inside the streaming loop
    const T* pbase[IPL]; const T* qbase[IPL]; int64_t qstride[IPL]; int qvalid[IPL], qvb[IPL]; u32x4 qsign[IPL];
    const uint8_t* mbase[IPL];    // relu-bit bytes of the P rows, at window 0 (nullptr: P is used as stored)
#pragma unroll
    for (int i = 0; i < IPL; ++i) {
        const int* im = a.items + (it0 + min(i, nit - 1)) * ITEM_INTS;
        pbase[i] = reinterpret_cast<const T*>(a.ws + a.buf_off[im[0]]) + act_idx(0, im[2], a.B) + c * 8;
        mbase[i] = im[9] >= 0 ? reinterpret_cast<const uint8_t*>(a.ws + a.buf_off[im[9]]) + relu_byte(im[2], a.B, 0, c * 8) : nullptr;
        qsign[i] = u32x4{0, 0, 0, 0};
        if (im[4] >= 0) {
            qbase[i] = reinterpret_cast<const T*>(a.ws + a.buf_off[im[3]]) + act_idx(0, im[5], a.B) + c * 8;
            qstride[i] = H; qvalid[i] = 8; qvb[i] = 16;
        } else {
            const int t = im[3] - BUF_IN;
            qbase[i] = reinterpret_cast<const T*>(a.x[t]) + (size_t)im[5] * a.pitch[t] + im[6] + c * 8;
            qstride[i] = (int64_t)a.nodes[t] * a.pitch[t]; qvalid[i] = im[7] - c * 8; qvb[i] = a.vb[t];
            qsign[i] = sign_xor<T>(a.signs + im[8] + c * 8);
        }
    }
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

    struct Stage { u32x4 pv[4], qv[4]; unsigned mw[4]; };
    auto fetch = [&](Stage& st, int s) {
        const int w0 = (ch0 + (nit == 2 ? s >> 1 : s)) * GWB_KW;
        const int it = (nit == 2) ? (s & 1) : 0;
        const T* pb = it ? pbase[I1] : pbase[0];
        const T* qb = it ? qbase[I1] : qbase[0];
        const uint8_t* mb = it ? mbase[I1] : mbase[0];
        const int64_t qs = it ? qstride[I1] : qstride[0];
        const int qn = it ? qvalid[I1] : qvalid[0], vb = it ? qvb[I1] : qvb[0];
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int w = w0 + r0 + 16 * p;
            st.pv[p] = u32x4{0, 0, 0, 0}; st.qv[p] = u32x4{0, 0, 0, 0}; st.mw[p] = 0xffffffffu;
            if (w < a.B && !ABL(a.dbg & 1)) {
                st.pv[p] = *reinterpret_cast<const u32x4*>(pb + (size_t)w * H);
                if (mb) st.mw[p] = mb[((size_t)(w >> 4) << 6) + (w & 15)];
                if (a.aligned) { if (qn > 0) st.qv[p] = *reinterpret_cast<const u32x4*>(qb + (size_t)w * qs); }   // raw: a use here would serialise the loads
                else st.qv[p] = load_chunk<T>(qb + (size_t)w * qs, qn, vb);
            }
        }
    };
    // Lane-uniform kinds (the items of a lane belong to one target): P needs the relu mask only for items that carry one, Q needs
    // pad-column clearing and the symmetry sign only when it is an encoder input.  The staging path is VALU-bound next to the
    // loads (SQ counters: 29 M VALU instructions per launch against 1.4 M MFMAs), so the generic transforms are skipped where
    // they are the identity, and the 8 relu bits expand to a 16-byte AND mask through a 256-entry table in LDS.
    const bool p_masked = mbase[0] != nullptr, q_raw_input = qvb[0] != 16 || qvalid[0] != 8 || (qsign[0][0] | qsign[0][1] | qsign[0][2] | qsign[0][3] | qsign[nit == 2 ? I1 : 0][0] | qsign[nit == 2 ? I1 : 0][1] | qsign[nit == 2 ? I1 : 0][2] | qsign[nit == 2 ? I1 : 0][3]) != 0 || qvalid[nit == 2 ? I1 : 0] != 8;
    auto stage_to_lds = [&](const Stage& st, const u32x4 sx, const int qn, __bf16* Ps, __bf16* Qs) {
        if ABL(a.dbg & 2) { asm volatile("" :: "v"(st.pv[0][0]), "v"(st.qv[3][3])); return; }
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int row = r0 + 16 * p;
            u32x4 pm = st.pv[p];
            if (p_masked) pm &= mlut[st.mw[p] & 0xffu];                 // dH = dX . relu bits
            *reinterpret_cast<u32x4*>(&Ps[gwb_elem(row, c * 8)]) = pm;
            u32x4 qm = st.qv[p];
            if (q_raw_input) qm = chunk_keep_first<T>(qm, qn) ^ sx;     // drop pad columns, symmetry sign mask of encoder inputs
            *reinterpret_cast<u32x4*>(&Qs[gwb_elem(row, c * 8)]) = qm;
            if (bias_flag) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    bsum[2 * e] += __builtin_bit_cast(float, pm[e] << 16);
                    bsum[2 * e + 1] += __builtin_bit_cast(float, pm[e] & 0xffff0000u);
                }
            }
        }
    };
    auto mfmas = [&](const __bf16* Ps, const __bf16* Qs) {
        if ABL(a.dbg & 4) return;
#pragma unroll
        for (int ks = 0; ks < GWB_KW / 16; ++ks) {
            bf16x8 af[2], bq[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                af[i] = tr_frag(Ps, ks * 16, wr * 64 + i * 32, lane);
                bq[i] = tr_frag(Qs, ks * 16, wc * 64 + i * 32, lane);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bq[j], acc[i][j], 0, 0, 0);
        }
    };
    // register stages sa / sb hold the global loads of steps s+1, s+2 (in flight under the MFMAs); even steps use LDS
    // stage 0, odd steps LDS stage 1.  Loop invariant at the top of step s: LDS[s&1] holds step s, registers hold s+1 (and
    // s+2 is being fetched).
    const u32x4 sx1 = nit == 2 ? qsign[I1] : qsign[0];
    const int qn1 = nit == 2 ? qvalid[I1] : qvalid[0];
#if GW_DEEP
    Stage sa, sb;
    if (nsteps > 0) fetch(sa, 0);
    if (nsteps > 1) fetch(sb, 1);
    if (nsteps > 0) stage_to_lds(sa, qsign[0], qvalid[0], Pbuf[0], Qbuf[0]);
    if (nsteps > 2) fetch(sa, 2);
    __syncthreads();
    for (int s = 0; s < nsteps; s += 2) {
        // step s (LDS 0): write step s+1 (registers sb) into LDS 1 while multiplying LDS 0
        if (s + 1 < nsteps) stage_to_lds(sb, sx1, qn1, Pbuf[1], Qbuf[1]);
        if (s + 3 < nsteps) fetch(sb, s + 3);
        mfmas(Pbuf[0], Qbuf[0]);
        __syncthreads();
        if (s + 1 < nsteps) {
            // step s+1 (LDS 1): write step s+2 (registers sa) into LDS 0
            if (s + 2 < nsteps) stage_to_lds(sa, qsign[0], qvalid[0], Pbuf[0], Qbuf[0]);
            if (s + 4 < nsteps) fetch(sa, s + 4);
            mfmas(Pbuf[1], Qbuf[1]);
            __syncthreads();
        }
    }
#else
    // shallow variant: one register stage, one LDS stage, two barriers per step -- fewer registers / less LDS, one more
    // resident workgroup per CU
    Stage sa;
    if (nsteps > 0) fetch(sa, 0);
#ifndef MSHGNN_GW_STAMPS
#define MSHGNN_GW_STAMPS 0      // build with EXTRA=-DMSHGNN_GW_STAMPS=1 for tools/stamps_gradw.py (costs 6 VGPRs)
#endif
#if MSHGNN_GW_STAMPS
    long long tph[5] = {0, 0, 0, 0, 0}, tprev = a.stamps ? clock64() : 0;
    auto lap = [&](int k) { if (a.stamps) { const long long t = clock64(); tph[k] += t - tprev; tprev = t; } };
#else
    auto lap = [](int) {};
#endif
    for (int s = 0; s < nsteps; ++s) {
        __syncthreads();
        lap(0);      // waited for the previous MFMA phase of every wave
        stage_to_lds(sa, (nit == 2 && (s & 1)) ? qsign[I1] : qsign[0], (nit == 2 && (s & 1)) ? qvalid[I1] : qvalid[0], Pbuf[0], Qbuf[0]);
        lap(1);      // global loads landed + LDS written
        __syncthreads();
        lap(2);
        if (s + 1 < nsteps) fetch(sa, s + 1);
        lap(3);      // next step's loads issued
        mfmas(Pbuf[0], Qbuf[0]);
        lap(4);
    }
#if MSHGNN_GW_STAMPS
    if (a.stamps && tid == 0) { for (int k = 0; k < 5; ++k) a.stamps[(size_t)blockIdx.x * 8 + k] = tph[k]; a.stamps[(size_t)blockIdx.x * 8 + 5] = nsteps; }
#endif
    (void)sx1; (void)qn1;
#endif
    float* slab = a.slabs + (size_t)(part * a.n_lanes + ln) * SLAB_FLOATS;
    if (!ABL(a.dbg & 8))
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
        float* red = reinterpret_cast<float*>(Pbuf[0]);   // 16 x 128 floats = 8 KB <= one LDS stage
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

// k_gradw_bf16_lean: the same split-K step loop with a hot path that issues almost nothing but loads, LDS traffic and MFMAs.  The general kernel above
// spends ~300 VALU instructions per wave and 64-window step on 64-bit row addresses, bound checks and per-row mask bytes (SQ counters: 29 M VALU
// instructions per launch against 1.4 M MFMAs).  Here a thread owns FOUR CONSECUTIVE windows of one 16-byte column chunk, so that
//   * every global address is  wave-uniform stream pointer (SGPRs, advanced by scalar adds)  +  per-thread 32-bit offset  +  immediate (256 p),
//   * the relu bytes of its four rows are ONE aligned 32-bit load (the byte layout keeps 16 consecutive windows together),
//   * full chunks (all but the last one of the batch) carry no bound checks.
// A lane holds up to GW_IPL items (all of one target: same operand kinds, same input column chunk), interleaved chunk by chunk into the same
// accumulators -- chunk-major, so that every workgroup of the launch sweeps the batch at the same pace and rows shared between lanes still meet in
// L2 (item-major order was measured: 109 -> 160 us, the sharing is lost; more than two items per lane lose it too, see GW_IPL_MAX).  The items'
// stream bases and sign masks sit in a small LDS table; a step reads its three 64-bit bases from it (broadcast reads, moved to scalar
// registers).  ALIGNED: raw inputs in the engine's own 16-byte-aligned layout; the other instantiation reads them element-wise.  Same MFMA
// sequence as the general kernel.
// SERIES (with ALIGNED): a raw Q operand is gathered from the sequence's series like the encoder's input (k_enc_fwd<.., SERIES>): element k of a
// node row = element starts[w] + k % T of run k / T -- one unaligned 16-byte load per (window, chunk), two and a splice where the chunk straddles
// two runs; the four window starts of a thread are fetched one step ahead.  No materialised windows are read.
template <bool ALIGNED, bool SERIES = false> __global__ __launch_bounds__(256, GW_WPS) void k_gradw_bf16_lean(GradwArgs a) {
    static_assert(!SERIES || ALIGNED, "the series gather replaces the aligned raw loads");
    using T = __bf16;
    __shared__ __attribute__((aligned(16))) __bf16 Ps[GWB_KW * GWB_PITCH];
    __shared__ __attribute__((aligned(16))) __bf16 Qs[GWB_KW * GWB_PITCH];
    __shared__ __attribute__((aligned(16))) u32x4 mlut[256];       // relu byte -> AND mask of 8 bf16
    __shared__ __attribute__((aligned(16))) u32x4 qkeep_t[16];            // raw Q: keep mask (pad columns) of column chunk c (registers are short)
    __shared__ __attribute__((aligned(16))) u32x4 qsign_t[GW_IPL][16];    // raw Q: symmetry sign XOR of (item, column chunk)
    __shared__ unsigned long long sbase[GW_IPL][4];                       // per item: stream bases of P, relu bytes, Q at the part's first window
    __shared__ unsigned long long sser[SERIES ? GW_IPL : 1][16][2];       // SERIES, raw Q: per (item, column chunk) the run pointers (at the chunk's time offset) of its two pieces
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    mlut[tid] = chunk_mask_bits<__bf16>(u32x4{0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu}, (unsigned)tid);   // (visible after the first barrier)
    const int wr = wv >> 1, wc = wv & 1;
    const int ln = a.lane_order[blockIdx.x % a.n_pad], part = blockIdx.x / a.n_pad;
    if (ln < 0) return;
    const int* lh = a.lanes + ln * LANE_INTS;
    const int it0 = lh[0], nit = lh[1] - lh[0], bias_flag = lh[3];
    const int* im0 = a.items + it0 * ITEM_INTS;
    const int nchunks = (a.B + GWB_KW - 1) / GWB_KW;
    const int ch0 = (int)((int64_t)part * nchunks / a.n_parts), ch1 = (int)((int64_t)(part + 1) * nchunks / a.n_parts);
    const int nsteps = ch1 - ch0, total = nsteps * nit;
    const int c = tid & 15, r4 = (tid >> 4) * 4;      // rows r4 .. r4 + 3 of the 64-window step, columns [8c, 8c + 8)

    // what every item of the lane shares (one target): operand kinds, the input type and column chunk of a raw Q
    const bool p_masked = im0[9] >= 0, q_raw = im0[4] < 0;
    const int ntile = (a.B + 15) >> 4;
    const unsigned voffP = (unsigned)(r4 * H + c * 8) * (unsigned)sizeof(T);
    const unsigned voffM = (unsigned)(((c >> 2) * ntile + (r4 >> 4)) * 64 + (c & 3) * 16 + (r4 & 15));
    const int qt = q_raw ? im0[3] - BUF_IN : 0;
    const unsigned qsb = q_raw ? (unsigned)(a.nodes[qt] * a.pitch[qt]) * (unsigned)sizeof(T) : (unsigned)(H * sizeof(T));
    const int qn = q_raw ? im0[7] - c * 8 : 8, qvb = q_raw ? a.vb[qt] : 16;
    const unsigned voffQ0 = (unsigned)r4 * qsb + (unsigned)c * 16u;
    unsigned ldsw[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) ldsw[p] = (unsigned)gwb_elem(r4 + p, c * 8);
    if (tid < 16) qkeep_t[tid] = chunk_keep_first<T>(u32x4{0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu}, qn);      // (tid == c)
    if (tid < nit) {     // thread k resolves item k's stream bases (at the part's first window)
        const int* im = a.items + (it0 + tid) * ITEM_INTS;
        sbase[tid][0] = (unsigned long long)(a.ws + a.buf_off[im[0]] + (act_idx(0, im[2], a.B) + (size_t)ch0 * GWB_KW * H) * sizeof(T));
        sbase[tid][1] = p_masked ? (unsigned long long)(a.ws + a.buf_off[im[9]] + relu_byte(im[2], a.B, 0, 0) + (size_t)ch0 * (GWB_KW / 16) * 64) : 0ull;
        sbase[tid][2] = q_raw ? (unsigned long long)(reinterpret_cast<const char*>(a.x[qt]) + ((size_t)im[5] * a.pitch[qt] + im[6]) * sizeof(T) + (size_t)ch0 * GWB_KW * qsb)
                              : (unsigned long long)(a.ws + a.buf_off[im[3]] + (act_idx(0, im[5], a.B) + (size_t)ch0 * GWB_KW * H) * sizeof(T));
    }
    if (q_raw && tid < nit * 16) {
        const int* im = a.items + (it0 + (tid >> 4)) * ITEM_INTS;
        qsign_t[tid >> 4][c] = sign_xor<T>(a.signs + im[8] + c * 8);
        if constexpr (SERIES) {      // elements [k0, k0 + 8) of the node row: n0 from run j at time offset off, the rest from run j + 1 at offset 0
            const int k0 = im[6] + c * 8, j = k0 / a.ser.T, off = k0 - j * a.ser.T, n0 = min(8, a.ser.T - off);
            const int rfirst = a.ser.rows[2 * (a.ser.row0[qt] + im[5])];
            const unsigned long long pa = qn > 0 ? a.ser.run_ptr[rfirst + j] : 0ull, pb = min(qn, 8) > n0 ? a.ser.run_ptr[rfirst + j + 1] : 0ull;
            sser[tid >> 4][c][0] = pa ? pa + (unsigned long long)off * sizeof(T) : 0ull;
            sser[tid >> 4][c][1] = pb;
        }
    }
    int s_n0 = 8; bool s_two = false;      // SERIES: this thread's chunk takes n0 elements from its first run; a second piece exists
    if constexpr (SERIES) { if (q_raw) { const int k0 = im0[6] + c * 8, off = k0 % a.ser.T; s_n0 = min(8, a.ser.T - off); s_two = min(qn, 8) > s_n0; } }
    __syncthreads();
    const char* pS = a.ws; const char* mS = a.ws; const char* qS = a.ws;
    auto ubase = [&](int k, int j, size_t off) -> const char* {      // wave-uniform 64-bit pointer out of the LDS table
        const unsigned long long v = sbase[k][j] + off;
        return reinterpret_cast<const char*>(((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(v >> 32)) << 32) |
                                             (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)v));
    };
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

    u32x4 pv[4], qv[4]; unsigned mw = 0xffffffffu;
    int nk = 0, nc = 0;            // (item, chunk of the part) of the next fetch: items interleaved chunk by chunk
    int wst[SERIES ? 4 : 1];      // SERIES: first series row of the four windows of the NEXT fetch (loaded one step ahead)
    auto load_starts = [&](int chunk) {
        if constexpr (SERIES) {
            if (q_raw) {
#pragma unroll
                for (int p = 0; p < 4; ++p) wst[p] = (int)a.ser.starts[min((ch0 + chunk) * GWB_KW + r4 + p, a.B - 1)];
            }
        }
    };
    auto load_q = [&](int p) -> u32x4 {
        if constexpr (SERIES) {
            if (q_raw) {
                const u32x4 ones = u32x4{0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};      // the constant-1 run (bf16 1.0)
                const unsigned long long pa = sser[nk][c][0];
                u32x4 va = ones;
                if (pa) va = *reinterpret_cast<const u32x4*>(reinterpret_cast<const T*>(pa) + wst[p]);
                if (s_two) {
                    const unsigned long long pb = sser[nk][c][1];
                    u32x4 vb2 = ones;
                    if (pb) vb2 = *reinterpret_cast<const u32x4*>(reinterpret_cast<const T*>(pb) + wst[p]);
                    va = splice8(va, vb2, s_n0);
                }
                return va;
            }
        }
        if constexpr (ALIGNED) return *reinterpret_cast<const u32x4*>(qS + voffQ0 + p * qsb);
        else return q_raw ? load_chunk<T>(reinterpret_cast<const T*>(qS + voffQ0 + p * qsb), qn, qvb) : *reinterpret_cast<const u32x4*>(qS + voffQ0 + p * qsb);
    };
    auto fetch = [&]() {
        pS = ubase(nk, 0, (size_t)nc * GWB_KW * H * sizeof(T));
        if (p_masked) mS = ubase(nk, 1, (size_t)nc * (GWB_KW / 16) * 64);
        qS = ubase(nk, 2, (size_t)nc * GWB_KW * qsb);
        const int w0 = (ch0 + nc) * GWB_KW;
        if (w0 + GWB_KW <= a.B) {
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                pv[p] = *reinterpret_cast<const u32x4*>(pS + voffP + p * (int)(H * sizeof(T)));
                qv[p] = u32x4{0, 0, 0, 0};
                if (qn > 0) qv[p] = load_q(p);
            }
            if (p_masked) mw = *reinterpret_cast<const unsigned*>(mS + voffM);
        } else {                   // last chunk of the batch: rows beyond B are zero
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                pv[p] = u32x4{0, 0, 0, 0}; qv[p] = u32x4{0, 0, 0, 0};
                if (w0 + r4 + p < a.B) {
                    pv[p] = *reinterpret_cast<const u32x4*>(pS + voffP + p * (int)(H * sizeof(T)));
                    if (qn > 0) qv[p] = load_q(p);
                }
            }
            mw = 0xffffffffu;
            if (p_masked && w0 + r4 < a.B) mw = *reinterpret_cast<const unsigned*>(mS + voffM);     // (the 16-window tile of row r4 exists)
        }
        if (++nk == nit) { nk = 0; ++nc; if (nc < nsteps) load_starts(nc); }      // (the next chunk's window starts: used by the fetch after this one)
    };
    int sk = 0;                    // item of the step being staged
    auto stage_to_lds = [&]() {
        u32x4 mk[4], qkeep, sx;
        if (q_raw) { qkeep = qkeep_t[c]; sx = qsign_t[sk][c]; }
        if (++sk == nit) sk = 0;
        if (p_masked) {              // the four table reads go out together (one LDS round trip, not four)
#pragma unroll
            for (int p = 0; p < 4; ++p) mk[p] = mlut[(mw >> (8 * p)) & 0xffu];
        }
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            u32x4 pm = pv[p];
            if (p_masked) pm &= mk[p];                                   // dH = dX . relu bits
            *reinterpret_cast<u32x4*>(&Ps[ldsw[p]]) = pm;
            u32x4 qm = qv[p];
            if (q_raw) qm = (qm & qkeep) ^ sx;                           // drop pad columns, symmetry sign mask of encoder inputs
            *reinterpret_cast<u32x4*>(&Qs[ldsw[p]]) = qm;
            if (bias_flag) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    bsum[2 * e] += __builtin_bit_cast(float, pm[e] << 16);
                    bsum[2 * e + 1] += __builtin_bit_cast(float, pm[e] & 0xffff0000u);
                }
            }
        }
    };
    if (total > 0) { load_starts(0); fetch(); }
    for (int s = 0; s < total; ++s) {
        __syncthreads();             // every wave is done with the previous step's tiles
        stage_to_lds();
        __syncthreads();
        if (s + 1 < total) fetch();
#pragma unroll
        for (int ks = 0; ks < GWB_KW / 16; ++ks) {
            bf16x8 af[2], bq[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                af[i] = tr_frag(Ps, ks * 16, wr * 64 + i * 32, lane);
                bq[i] = tr_frag(Qs, ks * 16, wc * 64 + i * 32, lane);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bq[j], acc[i][j], 0, 0, 0);
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
        float* red = reinterpret_cast<float*>(Ps);   // 16 x 128 floats = 8 KB <= one tile
        __syncthreads();
#pragma unroll
        for (int e = 0; e < 8; ++e) red[(r4 >> 2) * H + c * 8 + e] = bsum[e];
        __syncthreads();
        if (tid < H) {
            float s2 = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) s2 += red[r * H + tid];
            slab[H * H + tid] = s2;
        }
    }
}

// k_finalize: sum split-K slabs in fixed order into the flat gradient buffer (every parameter written once)

#ifndef FIN_LB
#define FIN_LB 4       // lanes per load batch of k_finalize (x 4 parts = 16 slab loads in flight per thread; 8 / 16 lanes measured: A1-C2 18.7 -> 18.6 / 17.9 us,
                       // MiniCheetah-K4 L=8 19.3 -> 20.5 / 23.6 us)
#endif
__global__ __launch_bounds__(256) void k_finalize(FinArgs a) {
    const int* f = a.fin + blockIdx.x * FIN_INTS;
    const int64_t dst = (int64_t)(unsigned)f[0] | ((int64_t)f[1] << 32);
    const int rows = f[2], cols = f[3], ld = f[4], tg = f[5], kind = f[6];
    const int* more = f[7] ? a.fin0 + f[7] : nullptr;      // further destinations of the same sum: {n, dst_lo, dst_hi, ...}
    const int RPB = (rows + gridDim.y - 1) / gridDim.y;
    const int r0 = blockIdx.y * RPB, r1 = min(rows, r0 + RPB);
    const int n = (r1 - r0) * cols;
    if (a.loss && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x < 64) {   // fused MSE: sum the per-block loss partials
        float l = 0.f;
        for (int b = threadIdx.x; b < a.n_dec; b += 64) l += a.dec_slabs[(size_t)b * DEC_SLAB_FLOATS + 8 * H + 8];
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) l += __shfl_xor(l, m, 64);
        if (threadIdx.x == 0) *a.loss = l * a.inv_n + (a.accumulate ? *a.loss : 0.f);      // (accumulate: a later sub-step of a chunked step, mshgnn_device.hpp StepChunk)
    }
    const bool is_mat = (kind == FIN_MATRIX || kind == FIN_DEC_W);
    if (kind == FIN_DEC_W || kind == FIN_DEC_B) {
        // decoder partials: NWG_DEC slabs per element -> one WAVE per element, 8 slabs per lane in flight, then a fixed
        // xor-shuffle tree (deterministic); elements are dealt round-robin to the (gridDim.y x 4) waves of this op
        const int lane = threadIdx.x & 63, wave = blockIdx.y * 4 + (threadIdx.x >> 6), nwaves = gridDim.y * 4;
        for (int e = wave; e < rows * cols; e += nwaves) {
            const int r = e / cols, cidx = e % cols;
            const int src = is_mat ? r * H + cidx : 8 * H + cidx;
            float sum = 0.f;
            for (int b0 = 0; b0 < a.n_dec; b0 += 512) {       // 8 slabs per lane in flight; lane-strided, fixed order
                float v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) { const int b = b0 + lane + 64 * u; v[u] = b < a.n_dec ? a.dec_slabs[(size_t)b * DEC_SLAB_FLOATS + src] : 0.f; }
                sum += ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
            }
#pragma unroll
            for (int m = 32; m >= 1; m >>= 1) sum += __shfl_xor(sum, m, 64);
            if (lane == 0) { float* gp = a.grad + dst + (int64_t)r * ld + cidx; *gp = a.accumulate ? *gp + sum : sum; }
        }
        return;
    }
    if (n <= 0) return;
    for (int i = threadIdx.x; i < n; i += 256) {
        const int r = r0 + i / cols, cidx = i % cols;
        float s = 0.f;
        if (kind == FIN_MATRIX || kind == FIN_BIAS) {
            const int src = is_mat ? r * H + cidx : H * H + cidx;
            const int l0 = a.targets[tg * TGT_INTS], nl = a.targets[tg * TGT_INTS + 1] - l0;
            // slab(part, j) = part * n_lanes + l0 + j; summed lane-major in batches of 16 loads -- fixed summation order
            const float* sp = a.slabs + (size_t)l0 * SLAB_FLOATS + src;
            const size_t pstride = (size_t)a.n_lanes * SLAB_FLOATS;
            for (int j0 = 0; j0 < nl; j0 += FIN_LB)
                for (int p0 = 0; p0 < a.n_parts; p0 += 4) {
                    float v[FIN_LB][4];
#pragma unroll
                    for (int jj = 0; jj < FIN_LB; ++jj)
#pragma unroll
                        for (int pp = 0; pp < 4; ++pp)
                            v[jj][pp] = (j0 + jj < nl && p0 + pp < a.n_parts) ? sp[(size_t)(j0 + jj) * SLAB_FLOATS + (size_t)(p0 + pp) * pstride] : 0.f;
#pragma unroll
                    for (int jj = 0; jj < FIN_LB; ++jj) s += (v[jj][0] + v[jj][1]) + (v[jj][2] + v[jj][3]);
                }
        }
        if (a.accumulate) s += a.grad[dst + (int64_t)r * ld + cidx];      // (the further destinations of a shared sum held the same value)
        a.grad[dst + (int64_t)r * ld + cidx] = s;
        if (more) {
            const int nm = more[0];
            for (int k = 0; k < nm; ++k) a.grad[((int64_t)(unsigned)more[1 + 2 * k] | ((int64_t)more[2 + 2 * k] << 32)) + (int64_t)r * ld + cidx] = s;
        }
    }
}

// ------------------------------------------------------------------------------------------------------
// plan object + C-ABI
// ------------------------------------------------------------------------------------------------------

// the kernels over this plan's compile-time program, if one exists (built in, or attached): their dynamic-LDS attribute, the program's name; else use_spec = false
static int set_spec_attrs(mshgnn_plan* p, int flds) {
    const HostPlan& hp = p->hp;
    int rc;
    const char* nm = nullptr;
    for (int nt = 0; nt < 2; ++nt)
        if (StackKernel k = slab_step_spec_kernel(hp, nt, &nm)) if ((rc = set_lds_attr(k, flds))) return rc;
    if (!nm) { p->use_spec = false; return MSHGNN_OK; }
    p->spec_name_buf = strncmp(nm, "spec::", 6) == 0 ? nm + 6 : nm;
    p->spec_name = p->spec_name_buf.c_str();
    for (int v = 0; v < 3; ++v)
        if (StackKernel k = slab_fwd_spec_kernel(hp, v > 0, v - 1)) if ((rc = set_lds_attr(k, flds))) return rc;
    for (int nt = 0; nt < 2; ++nt)
        if (StackKernel k = slab_bwd_spec_kernel(hp, nt)) if ((rc = set_lds_attr(k, flds))) return rc;
    if (StackKernel k = slab_step_spec_kernel(hp, 0, nullptr, false)) if ((rc = set_lds_attr(k, flds))) return rc;      // (ragged batches)
    if (StackKernel k = slab_fwd_spec_kernel(hp, 0, 0, false)) if ((rc = set_lds_attr(k, flds))) return rc;
    return MSHGNN_OK;
}

// A program compiled for this plan after the library was built (morphsym_hgnn_amd/jit.py: this source as shard 99 over the plan's own tables): `selector` is that
// library's mshgnn_jit_program.  Only LDS-resident bf16 plans whose slab kernels are in use take one; the tables are compared like a built-in program's.
extern "C" int mshgnn_plan_attach_program(mshgnn_plan* p, void* selector) {
    if (!p || !selector) return set_err(MSHGNN_EINVAL, "plan or selector is null");
    if (!p->gen && p->hp.d.dtype == MSHGNN_BF16X3) return x3_attach_program(p, selector);      // (the split plan: selector = mshgnn_jit_program_x3)
    if (p->gen || p->hp.d.dtype != MSHGNN_BF16 || !p->use_fused || !p->use_slab) return set_err(MSHGNN_EINVAL, "attached programs exist for LDS-resident bf16 / split-bf16 plans (bf16: on the slab kernels) only");
    void* const prev = p->hp.jit_prog;
    p->hp.jit_prog = selector;
    const char* nm = nullptr;
    if (!reinterpret_cast<SpecSelector>(selector)(p->hp, 0, 1, 0, 1, &nm) || !nm) {
        p->hp.jit_prog = prev;
        return set_err(MSHGNN_EINVAL, "the program's tables are not this plan's");
    }
    p->use_spec = true;
    const int rc = set_spec_attrs(p, (p->hp.fs_blk + FS_EXTRA_BLK) * Prec<__bf16>::BLK);
    if (rc || !p->use_spec) { p->hp.jit_prog = prev; p->use_spec = false; return rc ? rc : set_err(MSHGNN_EINVAL, "no kernel of the attached program could be used"); }
    return MSHGNN_OK;
}

extern "C" int mshgnn_plan_create(const mshgnn_desc* desc, mshgnn_plan** out) {
    if (!out) return set_err(MSHGNN_EINVAL, "plan_out is null");
    *out = nullptr;
    mshgnn_plan* p = new (std::nothrow) mshgnn_plan();
    if (!p) return set_err(MSHGNN_ENOMEM, "out of host memory");
    if (!desc) { delete p; return set_err(MSHGNN_EINVAL, "null descriptor"); }
    p->n_types = desc->n_types;
    if (const char* e = getenv("MSHGNN_STEP_CHUNK")) p->step_chunk = std::max<int64_t>(0, atoll(e));
    // Engine choice: the LDS-resident kernels (hidden == 128, <= 20 nodes, in-degree 1 on mean relations) where they apply, else the
    // generic-width engine of mshgnn_gen.hip.  MSHGNN_ENGINE=generic forces the latter (the GPU tests run the golden cases through both).
    const char* eng_env = getenv("MSHGNN_ENGINE");
    bool want_gen = eng_env && std::string(eng_env) == "generic";
    std::string why;
    if (!want_gen && !compile_plan(desc, p->hp)) {
        why = p->hp.err;
        const bool unsup = why.find("supports") != std::string::npos || why.find("not supported") != std::string::npos || why.find("too many") != std::string::npos;
        if (!unsup) { delete p; return set_err(MSHGNN_EINVAL, why); }
        want_gen = true;
    }
    if (want_gen) {
        int ndev0 = 0;
        if (hipGetDeviceCount(&ndev0) != hipSuccess || ndev0 == 0) { delete p; return set_err(MSHGNN_EHIP, "no HIP device: the MS-HGNN engine has no CPU fallback"); }
        const int rc = gen_create(p, desc);
        if (rc) {
            const std::string m = g_err; gen_destroy(p); delete p;
            return set_err(rc, why.empty() ? m : why + "; generic-width engine: " + m);
        }
#ifdef MSHGNN_ABLATE
        { const char* e = getenv("MSHGNN_DBG"); p->dbg = e ? atoi(e) : 0; }
#endif
        if (!p->hp.d.n_types) {      // forced generic engine: the specialised compiler never ran; entry points read the scalar fields from here
            p->hp.d = *desc;         // (its host pointers belong to the caller and are not read again)
        }
        *out = p;
        return MSHGNN_OK;
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) {
        delete p; return set_err(MSHGNN_EHIP, "no HIP device: the MS-HGNN engine has no CPU fallback");
    }
#ifdef MSHGNN_ABLATE
    { const char* e = getenv("MSHGNN_DBG"); p->dbg = e ? atoi(e) : 0; e = getenv("MSHGNN_DBG_GW"); p->dbg_gw = e ? atoi(e) : 0; }
#endif
    HostPlan& hp = p->hp;
    auto up = [&](void** dptr, const void* src, size_t bytes) -> int {
        HIPCHK(hipMalloc(dptr, std::max<size_t>(bytes, 16)));
        if (bytes) HIPCHK(hipMemcpy(*dptr, src, bytes, hipMemcpyHostToDevice));
        return MSHGNN_OK;
    };
    int rc;
    if ((rc = up((void**)&p->d_tables, hp.tables.data(), hp.tables.size() * 4)) != 0 ||
        (rc = up((void**)&p->d_signs, hp.signs.data(), hp.signs.size())) != 0 ||
        (rc = up((void**)&p->d_out_mask, hp.out_mask_f.data(), hp.out_mask_f.size() * 4)) != 0 ||
        (rc = up((void**)&p->d_packs, hp.packs.data(), hp.packs.size() * sizeof(PackDesc))) != 0 ||
        (rc = up((void**)&p->d_biases, hp.biases.data(), hp.biases.size() * sizeof(BiasDesc))) != 0) {
        mshgnn_plan_destroy(p); return rc;
    }
    const int lds = hp.n_blk * hp.blk_bytes;
    if (hp.d.dtype == MSHGNN_BF16X3) {
        p->use_fused = true;      // the split plan has only the fused 8-wave stack kernels (mshgnn_x3.hip)
        { const char* et = getenv("MSHGNN_STEP_KERNEL"); p->use_step = !(et && atoi(et) == 0); }      // read per plan, as on the bf16 plan
        if ((rc = x3_set_attrs(p))) { mshgnn_plan_destroy(p); return rc; }
    } else if (hp.d.dtype == MSHGNN_F32) {
        if ((rc = set_lds_attr(k_layer_fwd<float>, lds)) || (rc = set_lds_attr(k_layer_bwd<float>, lds)) ||
            (rc = set_lds_attr(k_enc_fwd<float, true>, Prec<float>::ENC_MB * Prec<float>::BLK)) || (rc = set_lds_attr(k_enc_fwd<float, false>, Prec<float>::ENC_MB * Prec<float>::BLK))) { mshgnn_plan_destroy(p); return rc; }
    } else {
        if ((rc = set_lds_attr(k_layer_fwd<__bf16>, lds)) || (rc = set_lds_attr(k_layer_bwd<__bf16>, lds)) ||
            (rc = set_lds_attr(k_enc_fwd<__bf16, true>, Prec<__bf16>::ENC_MB * Prec<__bf16>::BLK)) || (rc = set_lds_attr(k_enc_fwd<__bf16, true, true>, Prec<__bf16>::ENC_MB * Prec<__bf16>::BLK)) || (rc = set_lds_attr(k_enc_fwd<__bf16, true, false, 8>, Prec<__bf16>::ENC_MB * Prec<__bf16>::BLK)) || (rc = set_lds_attr(k_enc_fwd<__bf16, true, false, 4>, Prec<__bf16>::ENC_MB * Prec<__bf16>::BLK)) || (rc = set_lds_attr(k_enc_fwd<__bf16, false>, Prec<__bf16>::ENC_MB * Prec<__bf16>::BLK))) { mshgnn_plan_destroy(p); return rc; }
        const char* e = getenv("MSHGNN_FUSED");
        p->use_fused = hp.fused && !(e && atoi(e) == 0);
        if (p->use_fused) {
            const int flds = (hp.fs_blk + FS_EXTRA_BLK) * Prec<__bf16>::BLK;
            if ((rc = set_lds_attr(k_stack_fwd<__bf16>, flds)) || (rc = set_lds_attr(k_stack_bwd<__bf16>, flds)) || (rc = set_lds_attr(k_stack_step<__bf16>, flds))) { mshgnn_plan_destroy(p); return rc; }
            const char* es = getenv("MSHGNN_SLAB");
            p->use_slab = hp.slab && !(es && atoi(es) == 0);       // default on where the plan allows it; MSHGNN_SLAB=0 selects the 8-wave kernels
            p->slab_force = es && atoi(es) == 2;                   // MSHGNN_SLAB=2: also for batches that do not fill the chip
            { int dev = 0, cus = 256; if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev); p->n_cu = cus > 0 ? cus : 256; }
            if (p->use_slab && ((rc = set_lds_attr(slab_fwd_kernel(hp), flds)) || (rc = set_lds_attr(slab_bwd_kernel(hp), flds)) ||
                                (rc = set_lds_attr(slab_step_kernel(hp), flds)))) { mshgnn_plan_destroy(p); return rc; }
            { const char* esp = getenv("MSHGNN_SPEC"); p->use_spec = p->use_slab && !(esp && atoi(esp) == 0); }
            if (p->use_spec && (rc = set_spec_attrs(p, flds))) { mshgnn_plan_destroy(p); return rc; }
            { const char* et = getenv("MSHGNN_STEP_KERNEL"); p->use_step = !(et && atoi(et) == 0); }      // one-call steps: forward + backward sweep in one launch

            { const char* eg = TUNE_ENV("MSHGNN_STAGGER"); p->stagger = eg ? atoi(eg) : 0; }
        }
    }
    *out = p;
    return MSHGNN_OK;
}

extern "C" void mshgnn_plan_destroy(mshgnn_plan* p) {
    if (!p) return;
    gen_destroy(p);
    if (p->d_tables) (void)hipFree(p->d_tables);
    if (p->d_signs) (void)hipFree(p->d_signs);
    if (p->d_out_mask) (void)hipFree(p->d_out_mask);
    if (p->d_packs) (void)hipFree(p->d_packs);
    if (p->d_biases) (void)hipFree(p->d_biases);
    for (ProfRec& r : p->recs) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); }
    for (hipEvent_t e : p->free_events) (void)hipEventDestroy(e);
    delete p;
}

extern "C" int mshgnn_profile_enable(mshgnn_plan* p, int on) {
    if (!p) return set_err(MSHGNN_EINVAL, "null plan");
    p->prof = on != 0;
    return MSHGNN_OK;
}

extern "C" int mshgnn_profile_read(mshgnn_plan* p, mshgnn_kernel_stat* stats, int32_t* n_inout) {
    if (!p || !stats || !n_inout) return set_err(MSHGNN_EINVAL, "null argument");
    std::vector<mshgnn_kernel_stat> ks = p->gen ? *gen_kstats(p) : p->hp.kstats;
    for (const ProfRec& r : p->recs) {
        HIPCHK(hipEventSynchronize(r.b));
        float ms = 0.f;
        HIPCHK(hipEventElapsedTime(&ms, r.a, r.b));
        if (r.slot >= 0 && r.slot < (int)ks.size()) { ks[r.slot].total_ms += ms; ks[r.slot].launches += 1; }
        p->free_events.push_back(r.a); p->free_events.push_back(r.b);
    }
    p->recs.clear();
    const int n = std::min<int>(*n_inout, (int)ks.size());
    for (int i = 0; i < n; ++i) stats[i] = ks[i];
    *n_inout = n;
    return MSHGNN_OK;
}

extern "C" const char* mshgnn_plan_specialised(const mshgnn_plan* p) { return p && !p->gen && p->use_spec ? p->spec_name : ""; }
extern "C" int mshgnn_plan_info(const mshgnn_plan* p, mshgnn_info* info) {
    if (!p || !info) return set_err(MSHGNN_EINVAL, "null argument");
    *info = p->gen ? *gen_info(p) : p->hp.info;
    return MSHGNN_OK;
}

// host-only plan compilation (no GPU needed): used by the CPU test-suite to check the plan compiler
extern "C" int mshgnn_plan_compile_host(const mshgnn_desc* desc, mshgnn_info* info, int32_t* n_tables_out) {
    HostPlan hp;
    const char* eng_env = getenv("MSHGNN_ENGINE");
    if ((eng_env && std::string(eng_env) == "generic") || !compile_plan(desc, hp)) {
        const std::string why = hp.err;
        if (gen_host_compile(desc, info, n_tables_out) == MSHGNN_OK) return MSHGNN_OK;
        return set_err(MSHGNN_EINVAL, why.empty() ? g_err : why + "; generic-width engine: " + g_err);
    }
    if (info) *info = hp.info;
    if (n_tables_out) *n_tables_out = (int32_t)hp.tables.size();
    return MSHGNN_OK;
}

extern "C" int mshgnn_workspace_layout(const mshgnn_plan* p, int64_t batch, int training, mshgnn_ws_layout* out) {
    if (!p || !out || batch < 1) return set_err(MSHGNN_EINVAL, "bad argument to mshgnn_workspace_layout");
    if (p->gen) gen_layout(p, batch, training, out); else layout_workspace(p->hp, batch, training, out);
    return MSHGNN_OK;
}

int launch_prep(const PrepArgs& a, bool split, hipStream_t st) {
    if (split) return x3_launch_prep(a, st);
    const int64_t total = (int64_t)a.n_packs * (H * H / Prec<__bf16>::EPC) + (int64_t)a.n_biases * H;
    if (prep_use_tiled(a.n_packs)) hipLaunchKernelGGL((k_prep_tiled<__bf16, false>), dim3(prep_tiled_grid(a.n_packs, a.n_biases)), dim3(256), 0, st, a);
    else hipLaunchKernelGGL(k_prep<__bf16>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, a);
    return MSHGNN_OK;
}

int run_finalize(const mshgnn_plan* p, const mshgnn_ws_layout& lay, char* ws, float* gparams, int B, float* loss, bool is_ce, bool dec_done,
                 int gw_phase, hipStream_t st, int gw_parts) {
    const HostPlan& hp = p->hp;
    const mshgnn_desc& d = hp.d;
    FinArgs a{p->d_tables + hp.fin_off, p->d_tables + hp.fin_off, p->d_tables + hp.tgt_off, reinterpret_cast<const float*>(ws + lay.slabs),
              reinterpret_cast<const float*>(ws + lay.dec_slabs), gparams, hp.n_lanes, gw_parts, loss,
              1.0f / (float)(loss_windows(B) * d.type_nodes[d.out_type] * (is_ce ? 1 : d.out_channels)),
              dec_done ? (int)((B + TILE_ROWS - 1) / TILE_ROWS) : NWG_DEC, step_accumulates()};
    int f0 = 0, nf = hp.n_fin;
    if (gw_phase == 0) nf = hp.n_fin_ph0;
    if (gw_phase == 1) { f0 = hp.n_fin_ph0; nf = hp.n_fin - hp.n_fin_ph0; a.loss = nullptr; }
    a.fin += (size_t)f0 * FIN_INTS;
    ProfScope ps(p, hp.ks_fin, st);
    if (nf > 0) hipLaunchKernelGGL(k_finalize, dim3(nf, 64), dim3(256), 0, st, a);
    HIPCHK(hipGetLastError());
    return MSHGNN_OK;
}

template <typename T>
static int forward_impl(const mshgnn_plan* p, const void* const* x, const int64_t* x_pitch, const float* params, float* out,
                        char* ws, int64_t batch, int training, hipStream_t st, const float* y_fused = nullptr, const SeriesSrc* series = nullptr,
                        const int32_t* labels_fused = nullptr, bool* stack_step_done = nullptr) {
    // stack_step_done (one-call steps): where the slab kernels run and the loss is fused, the backward sweep of the stack runs in the same launch
    // (k_slab_step) and *stack_step_done tells backward_impl to skip its own
    const HostPlan& hp = p->hp;
    const mshgnn_desc& d = hp.d;
    mshgnn_ws_layout lay; layout_workspace(hp, batch, training, &lay);
    const int B = (int)batch;
    // 1. weight images.  bf16 plan with few packs: only the encoder's packs (the last ones of the list) + the biases here; the layer packs are packed
    //    by extra workgroups of the encoder launch, under its tail (EncArgs.prep) -- the whole-list launch in front of the encoder cost 10.9 us
    PrepArgs pa{params, ws + lay.wpack, reinterpret_cast<float*>(ws + lay.bias), p->d_packs, p->d_biases, (int)hp.packs.size(), (int)hp.biases.size()};
    int enc_pack0 = (int)hp.packs.size();
    for (int t = 0; t < hp.NT; ++t) if (hp.pack_enc_base[t] >= 0) enc_pack0 = std::min(enc_pack0, hp.pack_enc_base[t]);
    static const bool embed_off = TUNE_ENV("MSHGNN_PREP_EMBED") && atoi(TUNE_ENV("MSHGNN_PREP_EMBED")) == 0;      // (A/B runs)
    const bool embed = sizeof(T) == 2 && !series && !prep_use_tiled(pa.n_packs) && enc_pack0 > 0 && !embed_off;
    {
        PrepArgs a = pa;
        if (embed) { a.pack0 = enc_pack0; a.pack_n = pa.n_packs - enc_pack0; }
        const int64_t total = (int64_t)(embed ? a.pack_n : a.n_packs) * (H * H / Prec<T>::EPC) + (int64_t)hp.biases.size() * H;
        ProfScope ps(p, hp.ks_prep, st);
        if (prep_use_tiled(a.n_packs)) hipLaunchKernelGGL((k_prep_tiled<T, false>), dim3(prep_tiled_grid(a.n_packs, a.n_biases)), dim3(256), 0, st, a);
        else hipLaunchKernelGGL(k_prep<T>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, a);
    }
    // 2. encoder
    const WideSrc* wide = series ? nullptr : g_wide_src;      // (mshgnn_*_src: the caller's fp64 / fp32 rows; x = the plan-dtype rows to materialise, may be null)
    {
        EncArgs a{};
        a.n_types = hp.NT; a.B = B; a.NN = hp.NN; a.tiles = (B + Prec<T>::ENC_MB * Prec<T>::ROWS - 1) / (Prec<T>::ENC_MB * Prec<T>::ROWS);
        a.wg_prefix[0] = 0;
        for (int t = 0; t < hp.NT; ++t) {
            a.x[t] = x ? x[t] : nullptr; a.pitch[t] = x_pitch ? x_pitch[t] : d.type_width[t];
            if (a.pitch[t] < d.type_width[t]) return set_err(MSHGNN_EINVAL, "x_pitch smaller than the feature width");
            a.vb[t] = vec_bytes(a.x[t], a.pitch[t], (int)sizeof(T));
            if (t == 0) a.aligned = 1;
            if (a.vb[t] != 16 || a.pitch[t] % Prec<T>::EPC) a.aligned = 0;
            a.width[t] = d.type_width[t]; a.tbase[t] = hp.type_base[t]; a.nkc[t] = hp.enc_nkc[t];
            a.pack0[t] = hp.pack_enc_base[t]; a.bias_idx[t] = hp.bias_enc[t]; a.sign_off[t] = hp.sign_off[t];
            // the launch's nodes of this type: those whose X_0 can reach the output; with window rows to materialise (series route, x given) every node,
            // the others marked in skip_mask
            a.node_off[t] = t == 0 ? 0 : a.node_off[t - 1] + a.nodes[t - 1];
            a.nodes[t] = 0;
            const bool all_rows = series != nullptr && x != nullptr;
            for (int i = 0; i < d.type_nodes[t]; ++i) {
                const bool need = hp.need_n[0][hp.type_base[t] + i];
                if (need || all_rows) a.node_list[a.node_off[t] + a.nodes[t]++] = (unsigned char)i;
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
            if (a.aligned) { a.prep = lp; a.prep_vecs = enc_pack0 * (H * H / Prec<T>::EPC); enc_grid += (unsigned)((a.prep_vecs + 255) / 256); }
            else {      // the element-wise encoder has no embedded prep: pack the layer images in front of it after all
                ProfScope ps(p, hp.ks_prep, st);
                hipLaunchKernelGGL(k_prep<T>, dim3((unsigned)(((int64_t)enc_pack0 * (H * H / Prec<T>::EPC) + 255) / 256)), dim3(256), 0, st, lp);
            }
        }
        ProfScope ps(p, hp.ks_enc, st);
        if constexpr (sizeof(T) == 2) {
            if (series) {
                if (x && !a.aligned) return set_err(MSHGNN_EINVAL, "the fused window assembly needs 16-byte aligned window rows (pitch a multiple of 8)");
                enc_grid += (unsigned)((series->lab.B + 255) / 256);      // the label workgroups
                hipLaunchKernelGGL((k_enc_fwd<T, true, true>), dim3(enc_grid), dim3(256), Prec<T>::ENC_MB * Prec<T>::BLK, st, a, *series, WideSrc{});
            } else if (wide) {      // the caller's fp64 / fp32 rows: converted by the encoder, plan-dtype rows written to x on the side
                if (x && !a.aligned) return set_err(MSHGNN_EINVAL, "wide source rows: the plan-dtype rows need 16-byte alignment and a pitch that is a multiple of 8");
                if (wide->bytes == 8) hipLaunchKernelGGL((k_enc_fwd<T, true, false, 8>), dim3(enc_grid), dim3(256), Prec<T>::ENC_MB * Prec<T>::BLK, st, a, SeriesSrc{}, *wide);
                else hipLaunchKernelGGL((k_enc_fwd<T, true, false, 4>), dim3(enc_grid), dim3(256), Prec<T>::ENC_MB * Prec<T>::BLK, st, a, SeriesSrc{}, *wide);
            } else if (a.aligned) hipLaunchKernelGGL((k_enc_fwd<T, true>), dim3(enc_grid), dim3(256), Prec<T>::ENC_MB * Prec<T>::BLK, st, a, SeriesSrc{}, WideSrc{});
            else hipLaunchKernelGGL((k_enc_fwd<T, false>), dim3(enc_grid), dim3(256), Prec<T>::ENC_MB * Prec<T>::BLK, st, a, SeriesSrc{}, WideSrc{});
        } else {
            if (wide) return set_err(MSHGNN_EUNSUPPORTED, "wide source rows: not on the fp32 plan (cast the inputs)");
            if (a.aligned) hipLaunchKernelGGL((k_enc_fwd<T, true>), dim3(enc_grid), dim3(256), Prec<T>::ENC_MB * Prec<T>::BLK, st, a, SeriesSrc{}, WideSrc{});
            else hipLaunchKernelGGL((k_enc_fwd<T, false>), dim3(enc_grid), dim3(256), Prec<T>::ENC_MB * Prec<T>::BLK, st, a, SeriesSrc{}, WideSrc{});
        }
    }
    // 3. layers (+ decoder): one fused launch on the bf16 plan, else one kernel per layer and the decoder kernel
    const int tiles = (B + Prec<T>::ROWS - 1) / Prec<T>::ROWS;
    if constexpr (sizeof(T) == 2) {
        if (p->use_fused) {
            StackArgs a{};
            a.tile_in = ws + lay.x[0]; a.ws = ws;
            for (int l = 0; l <= hp.L; ++l) a.x_off[l] = lay.x[l];
            for (int l = 0; l < hp.L; ++l) { a.mask_off[l] = lay.mask[l]; a.hb_off[l] = lay.hb[l]; a.t1_off[l] = lay.t1[l]; a.prog_off[l] = hp.fs_fwd_off[l]; }
            a.wpack = ws + lay.wpack; a.bias = reinterpret_cast<const float*>(ws + lay.bias); a.tables = p->d_tables;
            a.B = B; a.NN = hp.NN; a.L = hp.L; a.training = training;
            a.dbg = p->dbg;
            a.params = params; a.out_mask = p->d_out_mask; a.out = out; a.off_dec_w = d.off_dec_w; a.off_dec_b = d.off_dec_b;
            a.node0 = hp.type_base[d.out_type]; a.n_out = d.type_nodes[d.out_type]; a.dout = d.out_channels;
            if (y_fused) {
                a.y = y_fused; a.dec_slabs = reinterpret_cast<float*>(ws + lay.dec_slabs); a.dx_off[hp.L] = lay.dx[hp.L];
                a.inv_n = 1.0f / (float)(loss_windows(B) * a.n_out * a.dout);
            } else if (labels_fused) {      // mshgnn_step_ce: cross entropy over the per-foot logit pairs, mean over B * n_out rows
                a.labels = labels_fused; a.dec_slabs = reinterpret_cast<float*>(ws + lay.dec_slabs); a.dx_off[hp.L] = lay.dx[hp.L];
                a.inv_n = 1.0f / (float)(loss_windows(B) * a.n_out);
            }
            a.stamps = stamp_ptr("MSHGNN_STAMPS");
            a.stagger = p->slab_for(tiles) && tiles > p->n_cu ? p->stagger : 0;
            a.stash_nt = training ? stash_nt_for(B, stash_rows_of(hp), H * (int)sizeof(T)) : 0;
            // the tail's reduction scratch (one decoder slab per wave) must not touch the out-type nodes' blocks, which receive dX_L for the backward sweep: it sits
            // in the blocks in front of them, or (models whose out type comes first: the centroidal-momentum ones) in the blocks behind them
            const bool want_step = stack_step_done && p->use_step && (y_fused || labels_fused);
            // Slab or 8-wave kernels: the 8-wave ones take batches of at most one tile per CU -- unless the one-call step has a compile-time program for this plan
            // (whole tiles): a single tile's chain on the specialised slab kernel is 37-42 us where the interpreting 8-wave kernel takes 57-62 (A1-C2, 32 .. 4 096 windows;
            // 8 layers: 140-166 against 202-209), so the specialised step runs at every batch size.  MSHGNN_SLAB=0 / MSHGNN_SPEC=0 keep the 8-wave kernels there.
            bool step_slab = p->slab_for(tiles);
            const bool spec_small = !step_slab && want_step && p->use_slab && p->use_spec && slab_step_spec_kernel(hp, a.stash_nt, nullptr, whole_tiles(B)) != nullptr;
            bool red_front_ok = false, red_back_ok = false; size_t red_back = 0;
            auto red_fits = [&](bool slab) {
                const size_t red_need = (size_t)((slab ? SL_THREADS : LAYER_THREADS) / 64) * DEC_SLAB_FLOATS * sizeof(float);
                const size_t lds_launch = (size_t)((slab ? hp.sl_blk : hp.fs_blk) + FS_EXTRA_BLK) * Prec<T>::BLK;
                red_back = (size_t)(a.node0 + a.n_out) * Prec<T>::BLK;
                red_front_ok = (size_t)a.node0 * Prec<T>::BLK >= red_need; red_back_ok = red_back + red_need <= lds_launch;
                return red_front_ok || red_back_ok; };
            if (spec_small && red_fits(true)) step_slab = true;
            const bool step = want_step && red_fits(step_slab);
            if (step && !red_front_ok) a.red_off = (int)red_back;
            // the forward launch alone (evaluation / two-call training) on its compile-time program, at every whole-tile batch size
            const StackKernel fwd_spec = (!step && p->use_slab && p->use_spec) ? slab_fwd_spec_kernel(hp, training, a.stash_nt, whole_tiles(B)) : nullptr;
            if (fwd_spec) step_slab = true;
            ProfScope ps(p, step ? hp.ks_stack_step : hp.ks_stack_fwd, st);
            if (step_slab) {
                for (int l = 0; l < hp.L; ++l) a.prog_off[l] = hp.sl_fwd_off[l];
                if (step) {
                    for (int l = 0; l <= hp.L; ++l) a.dx_off[l] = lay.dx[l];
                    for (int l = 0; l < hp.L; ++l) { a.dh_off[l] = lay.dh[l]; a.du_off[l] = lay.du[l]; a.prog_off_b[l] = hp.sl_bwd_off[l]; }
                    a.mask0_off = lay.dd[0];
                    hipLaunchKernelGGL(slab_step_kernel(hp, B, a.stash_nt, p->use_spec), dim3(tiles), dim3(SL_THREADS), (hp.sl_blk + FS_EXTRA_BLK) * Prec<T>::BLK, st, a);
                    *stack_step_done = true;
                } else
                hipLaunchKernelGGL(fwd_spec ? fwd_spec : slab_fwd_kernel(hp), dim3(tiles), dim3(SL_THREADS), (hp.sl_blk + FS_EXTRA_BLK) * Prec<T>::BLK, st, a);
            } else if (step) {
                for (int l = 0; l <= hp.L; ++l) a.dx_off[l] = lay.dx[l];
                for (int l = 0; l < hp.L; ++l) { a.dh_off[l] = lay.dh[l]; a.du_off[l] = lay.du[l]; a.prog_off_b[l] = hp.fs_bwd_off[l]; }
                a.mask0_off = lay.dd[0];
                hipLaunchKernelGGL(k_stack_step<T>, dim3(tiles), dim3(LAYER_THREADS), (hp.fs_blk + FS_EXTRA_BLK) * Prec<T>::BLK, st, a);
                *stack_step_done = true;
            } else
            hipLaunchKernelGGL(k_stack_fwd<T>, dim3(tiles), dim3(LAYER_THREADS), (hp.fs_blk + FS_EXTRA_BLK) * Prec<T>::BLK, st, a);
            HIPCHK(hipGetLastError());
            return MSHGNN_OK;
        }
    }
    for (int l = 0; l < hp.L; ++l) {
        LayerArgs a{};
        a.x_in = ws + lay.x[l]; a.x_out = ws + lay.x[l + 1]; a.maskbits = reinterpret_cast<unsigned*>(ws + lay.mask[l]);
        a.hb = ws + lay.hb[l]; a.t1 = ws + lay.t1[l]; a.wpack = ws + lay.wpack; a.bias = reinterpret_cast<const float*>(ws + lay.bias);
        a.prog = p->d_tables + hp.fwd_prog_off[l]; a.B = B; a.NN = hp.NN; a.n_mlp = std::max(1, hp.n_mlp);
        a.dbg = p->dbg;
        ProfScope ps(p, hp.ks_layer_fwd0 + l, st);
        hipLaunchKernelGGL(k_layer_fwd<T>, dim3(tiles), dim3(LAYER_THREADS), hp.n_blk * Prec<T>::BLK, st, a);
    }
    // 4. decoder
    {
        DecArgs a{};
        a.xl = ws + lay.x[hp.L]; a.params = params; a.out_mask = p->d_out_mask; a.out = out; a.off_w = d.off_dec_w; a.off_b = d.off_dec_b;
        a.B = B; a.NN = hp.NN; a.node0 = hp.type_base[d.out_type]; a.n_out = d.type_nodes[d.out_type]; a.dout = d.out_channels;
        const int64_t rows = (int64_t)B * a.n_out;
        ProfScope ps(p, hp.ks_dec_fwd, st);
        hipLaunchKernelGGL(k_dec_fwd<T>, dim3((unsigned)((rows + 15) / 16)), dim3(256), 0, st, a);
    }
    HIPCHK(hipGetLastError());
    return MSHGNN_OK;
}

template <typename T>
static int backward_impl(const mshgnn_plan* p, const void* const* x, const int64_t* x_pitch, const float* params, const float* gout,
                         float* gparams, char* ws, int64_t batch, hipStream_t st, const float* out = nullptr, const float* y = nullptr,
                         float* loss = nullptr, const int32_t* labels = nullptr, bool dec_done = false, int gw_phase = -1, const SeriesSrc* series = nullptr,
                         bool stack_done = false) {      // stack_done: k_slab_step already ran the backward sweep of the stack (forward_impl)
    // gw_phase: -1 = everything; 0 = backward sweep + the weight gradients of every parameter but the encoder's; 1 = only the
    // encoder's weight gradients (the sweep of phase 0 left dX_0 in the workspace)
    const HostPlan& hp = p->hp;
    const mshgnn_desc& d = hp.d;
    mshgnn_ws_layout lay; layout_workspace(hp, batch, 1, &lay);
    const int B = (int)batch;
    if (!dec_done && gw_phase != 1) {
        DecArgs a{};
        a.xl = ws + lay.x[hp.L]; a.dxl = ws + lay.dx[hp.L]; a.params = params; a.out_mask = p->d_out_mask; a.gout = gout;
        a.slabs = reinterpret_cast<float*>(ws + lay.dec_slabs); a.off_w = d.off_dec_w; a.off_b = d.off_dec_b;
        a.B = B; a.NN = hp.NN; a.node0 = hp.type_base[d.out_type]; a.n_out = d.type_nodes[d.out_type]; a.dout = d.out_channels; a.slab0 = 0;
        if (y) {
            a.y = y; a.out = const_cast<float*>(out); a.loss = loss; a.inv_n = 1.0f / (float)(loss_windows(B) * a.n_out * a.dout);
        }
        if (labels) {
            a.labels = labels; a.out = const_cast<float*>(out); a.loss = loss; a.inv_n = 1.0f / (float)(loss_windows(B) * a.n_out);
        }
        ProfScope ps(p, hp.ks_dec_bwd, st);
        hipLaunchKernelGGL(k_dec_bwd<T>, dim3(NWG_DEC), dim3(256), 0, st, a);
    }
    const int tiles = (B + Prec<T>::ROWS - 1) / Prec<T>::ROWS;
    bool fused_done = gw_phase == 1 || stack_done;
    if constexpr (sizeof(T) == 2) {
        if (p->use_fused && gw_phase != 1 && !stack_done) {
            StackArgs a{};
            a.tile_in = ws + lay.dx[hp.L]; a.ws = ws;
            for (int l = 0; l <= hp.L; ++l) { a.x_off[l] = lay.x[l]; a.dx_off[l] = lay.dx[l]; }
            for (int l = 0; l < hp.L; ++l) { a.mask_off[l] = lay.mask[l]; a.t1_off[l] = lay.t1[l]; a.dh_off[l] = lay.dh[l]; a.du_off[l] = lay.du[l]; a.prog_off[l] = hp.fs_bwd_off[l]; }
            a.wpack = ws + lay.wpack; a.bias = reinterpret_cast<const float*>(ws + lay.bias); a.tables = p->d_tables;
            a.B = B; a.NN = hp.NN; a.L = hp.L; a.training = 1;
            a.node0 = hp.type_base[d.out_type]; a.n_out = d.type_nodes[d.out_type];      // the nodes that carry dX_L (the only live type of the last layer)
            a.mask0_off = lay.dd[0];
            a.dbg = p->dbg;
            a.stamps = stamp_ptr("MSHGNN_STAMPS_BWD");
            a.stagger = p->slab_for(tiles) && tiles > p->n_cu ? p->stagger : 0;
            a.stash_nt = stash_nt_for(B, stash_rows_of(hp), H * (int)sizeof(T));
            ProfScope ps(p, hp.ks_stack_bwd, st);
            const StackKernel bwd_spec = (p->use_slab && p->use_spec) ? slab_bwd_spec_kernel(hp, a.stash_nt, whole_tiles(B)) : nullptr;      // (at every whole-tile batch size, like the forward)
            if (bwd_spec || p->slab_for(tiles)) {
                for (int l = 0; l < hp.L; ++l) a.prog_off[l] = hp.sl_bwd_off[l];
                hipLaunchKernelGGL(bwd_spec ? bwd_spec : slab_bwd_kernel(hp), dim3(tiles), dim3(SL_THREADS), (hp.sl_blk + FS_EXTRA_BLK) * Prec<T>::BLK, st, a);
            } else
            hipLaunchKernelGGL(k_stack_bwd<T>, dim3(tiles), dim3(LAYER_THREADS), (hp.fs_blk + FS_EXTRA_BLK) * Prec<T>::BLK, st, a);
            fused_done = true;
        }
    }
    for (int l = hp.L - 1; l >= 0 && !fused_done; --l) {
        LayerArgs a{};
        a.x_in = ws + lay.dx[l + 1]; a.x_out = ws + lay.dx[l]; a.maskbits = reinterpret_cast<unsigned*>(ws + lay.mask[l]);
        a.hb = ws + lay.hb[l]; a.t1 = ws + lay.t1[l]; a.dh = ws + lay.dh[l]; a.du = ws + lay.du[l]; a.x_act = ws + lay.x[0];
        a.wpack = ws + lay.wpack; a.bias = reinterpret_cast<const float*>(ws + lay.bias);
        a.prog = p->d_tables + hp.bwd_prog_off[l]; a.B = B; a.NN = hp.NN; a.n_mlp = std::max(1, hp.n_mlp);
        a.dbg = p->dbg;
        ProfScope ps(p, hp.ks_layer_bwd0 + (hp.L - 1 - l), st);
        hipLaunchKernelGGL(k_layer_bwd<T>, dim3(tiles), dim3(LAYER_THREADS), hp.n_blk * Prec<T>::BLK, st, a);
    }
    const int gw_parts = gw_parts_for(hp.n_parts, hp.n_lanes, hp.gw_ipl, B, sizeof(T) == 4 ? GW_KW : GWB_KW, p->n_cu);      // window parts of this batch's weight-gradient launch (<= the plan's)
    {
        GradwArgs a{};
        a.ws = ws;
        for (int l = 0; l <= hp.L; ++l) { a.buf_off[BUF_X + l] = lay.x[l]; a.buf_off[BUF_DX + l] = lay.dx[l]; }
        for (int l = 0; l < hp.L; ++l) a.buf_off[BUF_MASK + l] = lay.mask[l];
        for (int l = 0; l < hp.L; ++l) { a.buf_off[BUF_DH + l] = lay.dh[l]; a.buf_off[BUF_HB + l] = lay.hb[l]; a.buf_off[BUF_T1 + l] = lay.t1[l]; a.buf_off[BUF_DU + l] = lay.du[l]; }
        for (int t = 0; t < hp.NT; ++t) {
            a.x[t] = x ? x[t] : nullptr; a.pitch[t] = x_pitch ? x_pitch[t] : d.type_width[t]; a.nodes[t] = d.type_nodes[t];
            a.vb[t] = vec_bytes(a.x[t], a.pitch[t], (int)sizeof(T));
            if (t == 0) a.aligned = 1;
            if (a.vb[t] != 16 || a.pitch[t] % Prec<T>::EPC) a.aligned = 0;
        }
        if (series) a.ser = *series;
        a.items = p->d_tables + hp.item_off; a.lanes = p->d_tables + hp.lane_off; a.lane_order = p->d_tables + hp.lane_order_off; a.n_pad = hp.n_lanes_pad;
        if (gw_phase >= 0) { a.lane_order = p->d_tables + hp.order_ph_off[gw_phase]; a.n_pad = hp.npad_ph[gw_phase]; }
        a.signs = p->d_signs; a.slabs = reinterpret_cast<float*>(ws + lay.slabs); a.B = B; a.n_lanes = hp.n_lanes; a.n_parts = gw_parts;
        a.dbg = p->dbg_gw;
        a.stamps = stamp_ptr("MSHGNN_STAMPS_GW");
        ProfScope ps(p, hp.ks_gradw, st);
        auto launch_gradw = [&](const GradwArgs& ga, hipStream_t s_) {
            [[maybe_unused]] static const bool gw_general = TUNE_ENV("MSHGNN_GRADW") && std::string(TUNE_ENV("MSHGNN_GRADW")) == "general";   // read once: the general kernel also where the lean one applies (A/B runs)
            if (ga.n_pad <= 0) return;
            if constexpr (sizeof(T) == 4) hipLaunchKernelGGL(k_gradw_f32, dim3(ga.n_pad * gw_parts), dim3(256), 0, s_, ga);
#ifdef MSHGNN_TUNING      // (the general bf16 kernel of rounds 1-3: kept for A/B runs in tuning builds, not in the product binary)
            else if (gw_general && hp.gw_ipl == 1) hipLaunchKernelGGL(k_gradw_bf16<1>, dim3(ga.n_pad * gw_parts), dim3(256), 0, s_, ga);
            else if (gw_general && hp.gw_ipl == 2) hipLaunchKernelGGL(k_gradw_bf16<2>, dim3(ga.n_pad * gw_parts), dim3(256), 0, s_, ga);
#endif
            else if (series) hipLaunchKernelGGL((k_gradw_bf16_lean<true, true>), dim3(ga.n_pad * gw_parts), dim3(256), 0, s_, ga);      // raw operands from the series
            else if (ga.aligned) hipLaunchKernelGGL(k_gradw_bf16_lean<true>, dim3(ga.n_pad * gw_parts), dim3(256), 0, s_, ga);
            else hipLaunchKernelGGL(k_gradw_bf16_lean<false>, dim3(ga.n_pad * gw_parts), dim3(256), 0, s_, ga);
        };
        // (round 6, measured and not kept: phase 1's lanes on a side stream BESIDE phase 0's -- forked behind the stack launch, joined by phase 1's finalize.  On a
        //  1-rank RCCL group the two-phase step took 0.306 ms that way against 0.268 back to back and 0.186 for the plain step: two concurrent sweeps of the batch
        //  evict each other's shared rows.)
        launch_gradw(a, st);
    }
    return run_finalize(p, lay, ws, gparams, B, (y || labels) ? loss : nullptr, labels != nullptr, dec_done, gw_phase, st, gw_parts);
}

extern "C" int mshgnn_forward(const mshgnn_plan* p, const void* const* x, const int64_t* x_pitch, const float* params, float* out,
                              void* workspace, int64_t batch, int training, void* stream) {
    if (!p || !x || !params || !out || !workspace) return set_err(MSHGNN_EINVAL, "null argument to mshgnn_forward");
    if (batch < 1 || batch > (1 << 24)) return set_err(MSHGNN_EINVAL, "batch must be in [1, 2^24]");
    for (int t = 0; t < p->n_types; ++t) if (!x[t]) return set_err(MSHGNN_EINVAL, "null input tensor");
    if (p->gen) return gen_forward(p, x, x_pitch, params, out, (char*)workspace, batch, training, (hipStream_t)stream);
    if (p->hp.d.dtype == MSHGNN_BF16X3) return x3_forward(p, x, x_pitch, params, out, (char*)workspace, batch, training, (hipStream_t)stream, nullptr);
    if (p->hp.d.dtype == MSHGNN_F32) return forward_impl<float>(p, x, x_pitch, params, out, (char*)workspace, batch, training, (hipStream_t)stream);
    return forward_impl<__bf16>(p, x, x_pitch, params, out, (char*)workspace, batch, training, (hipStream_t)stream);
}

extern "C" int mshgnn_backward(const mshgnn_plan* p, const void* const* x, const int64_t* x_pitch, const float* params, const float* grad_out,
                               float* grad_params, void* workspace, int64_t batch, void* stream) {
    if (!p || !x || !params || !grad_out || !grad_params || !workspace) return set_err(MSHGNN_EINVAL, "null argument to mshgnn_backward");
    if (batch < 1 || batch > (1 << 24)) return set_err(MSHGNN_EINVAL, "batch must be in [1, 2^24]");
    if (p->gen) return gen_backward(p, x, x_pitch, params, grad_out, grad_params, (char*)workspace, batch, (hipStream_t)stream, nullptr, nullptr, nullptr, nullptr);
    if (p->hp.d.dtype == MSHGNN_BF16X3) return x3_backward(p, x, x_pitch, params, grad_out, grad_params, (char*)workspace, batch, (hipStream_t)stream, nullptr, nullptr, nullptr, nullptr, false, -1);
    if (p->hp.d.dtype == MSHGNN_F32) return backward_impl<float>(p, x, x_pitch, params, grad_out, grad_params, (char*)workspace, batch, (hipStream_t)stream);
    return backward_impl<__bf16>(p, x, x_pitch, params, grad_out, grad_params, (char*)workspace, batch, (hipStream_t)stream);
}

extern "C" int mshgnn_backward_mse(const mshgnn_plan* p, const void* const* x, const int64_t* x_pitch, const float* params, const float* out,
                                   const float* y, float* loss_out, float* grad_params, void* workspace, int64_t batch, void* stream) {
    if (!p || !x || !params || !out || !y || !loss_out || !grad_params || !workspace) return set_err(MSHGNN_EINVAL, "null argument to mshgnn_backward_mse");
    if (batch < 1 || batch > (1 << 24)) return set_err(MSHGNN_EINVAL, "batch must be in [1, 2^24]");
    if (p->gen) return gen_backward(p, x, x_pitch, params, nullptr, grad_params, (char*)workspace, batch, (hipStream_t)stream, out, y, loss_out, nullptr);
    if (p->hp.d.dtype == MSHGNN_BF16X3) return x3_backward(p, x, x_pitch, params, nullptr, grad_params, (char*)workspace, batch, (hipStream_t)stream, out, y, loss_out, nullptr, false, -1);
    if (p->hp.d.dtype == MSHGNN_F32) return backward_impl<float>(p, x, x_pitch, params, nullptr, grad_params, (char*)workspace, batch, (hipStream_t)stream, out, y, loss_out);
    return backward_impl<__bf16>(p, x, x_pitch, params, nullptr, grad_params, (char*)workspace, batch, (hipStream_t)stream, out, y, loss_out);
}

// ---- long batches as sub-steps (StepChunk, mshgnn_device.hpp) ----
// Windows per sub-step of a one-call step over `batch` windows, 0: the step runs whole.  Whole: batches under twice MSHGNN_STEP_CHUNK (read when the plan is created; default 32 768; 0 = always whole), the
// generic-width engine (its finalize kernel overwrites), wide-source calls (their source pointers are not offset here) and the sub-steps themselves.  Equal sub-steps of
// whole 16-window tiles.
static int64_t step_chunk_windows(const mshgnn_plan* p, int64_t batch) {
    const int64_t limit = p->step_chunk;
    if (limit <= 0 || batch < 2 * limit || p->gen || g_wide_src || g_step_chunk) return 0;
    const int64_t n = batch / limit;      // sub-steps of at least `limit` windows each (a step's cost per window is flat from there up; shorter ones cost ~5 % more)
    return ((batch + n - 1) / n + TILE_ROWS - 1) / TILE_ROWS * TILE_ROWS;
}
template <typename F> static int chunked_step(const mshgnn_plan* p, const void* const* x, const int64_t* x_pitch, int64_t batch, int64_t cw, F&& sub_step) {
    const mshgnn_desc& d = p->hp.d;
    const int64_t eb = d.dtype == MSHGNN_BF16 ? 2 : 4;      // input rows: bf16 on the bf16 plan, fp32 on the split and fp32 plans
    int idx = 0;
    for (int64_t w0 = 0; w0 < batch; w0 += cw, ++idx) {
        const void* xc[MSHGNN_MAX_TYPES] = {};
        for (int t = 0; t < p->n_types; ++t)
            xc[t] = static_cast<const char*>(x[t]) + w0 * d.type_nodes[t] * (x_pitch ? x_pitch[t] : (int64_t)d.type_width[t]) * eb;
        const StepChunk ck{batch, idx};
        g_step_chunk = &ck;
        const int rc = sub_step(xc, w0, std::min(cw, batch - w0));
        g_step_chunk = nullptr;
        if (rc) return rc;
    }
    return MSHGNN_OK;
}

extern "C" int mshgnn_step_mse(const mshgnn_plan* p, const void* const* x, const int64_t* x_pitch, const float* params, const float* y,
                               float* out, float* loss_out, float* grad_params, void* workspace, int64_t batch, void* stream) {
    if (!p || !x || !params || !y || !out || !loss_out || !grad_params || !workspace) return set_err(MSHGNN_EINVAL, "null argument to mshgnn_step_mse");
    if (batch < 1 || batch > (1 << 24)) return set_err(MSHGNN_EINVAL, "batch must be in [1, 2^24]");
    for (int t = 0; t < p->n_types; ++t) if (!x[t]) return set_err(MSHGNN_EINVAL, "null input tensor");
    if (const int64_t cw = step_chunk_windows(p, batch)) {      // a long batch: sub-steps over contiguous window ranges, one gradient (StepChunk)
        const int64_t orow = (int64_t)p->hp.d.type_nodes[p->hp.d.out_type] * p->hp.d.out_channels;
        return chunked_step(p, x, x_pitch, batch, cw, [&](const void* const* xc, int64_t w0, int64_t bc) {
            return mshgnn_step_mse(p, xc, x_pitch, params, y + w0 * orow, out + w0 * orow, loss_out, grad_params, workspace, bc, stream); });
    }
    hipStream_t st = (hipStream_t)stream;
    if (p->gen) {      // generic-width engine: the two-call sequence
        const int rc = gen_forward(p, x, x_pitch, params, out, (char*)workspace, batch, 1, st);
        if (rc) return rc;
        return gen_backward(p, x, x_pitch, params, nullptr, grad_params, (char*)workspace, batch, st, out, y, loss_out, nullptr);
    }
    if (p->hp.d.dtype == MSHGNN_BF16X3) {      // split plan: decoder, loss and decoder backward in the tail of its fused forward kernel as well
        bool stack_done = false;
        int rc = x3_forward(p, x, x_pitch, params, out, (char*)workspace, batch, 1, st, y, nullptr, &stack_done);
        if (rc) return rc;
        return x3_backward(p, x, x_pitch, params, nullptr, grad_params, (char*)workspace, batch, st, out, y, loss_out, nullptr, true, -1, stack_done);
    }
    if (p->hp.d.dtype == MSHGNN_F32 || !p->use_fused) {      // no fused stack kernels on this plan: the two-call sequence
        int rc = p->hp.d.dtype == MSHGNN_F32 ? forward_impl<float>(p, x, x_pitch, params, out, (char*)workspace, batch, 1, st)
                                              : forward_impl<__bf16>(p, x, x_pitch, params, out, (char*)workspace, batch, 1, st);
        if (rc) return rc;
        return p->hp.d.dtype == MSHGNN_F32 ? backward_impl<float>(p, x, x_pitch, params, nullptr, grad_params, (char*)workspace, batch, st, out, y, loss_out)
                                           : backward_impl<__bf16>(p, x, x_pitch, params, nullptr, grad_params, (char*)workspace, batch, st, out, y, loss_out);
    }
    bool stack_done = false;
    int rc = forward_impl<__bf16>(p, x, x_pitch, params, out, (char*)workspace, batch, 1, st, y, nullptr, nullptr, &stack_done);
    if (rc) return rc;
    return backward_impl<__bf16>(p, x, x_pitch, params, nullptr, grad_params, (char*)workspace, batch, st, out, y, loss_out, nullptr, true, -1, nullptr, stack_done);
}

extern "C" int mshgnn_step_ce(const mshgnn_plan* p, const void* const* x, const int64_t* x_pitch, const float* params, const int32_t* labels,
                              float* out, float* loss_out, float* grad_params, void* workspace, int64_t batch, void* stream) {
    if (!p || !x || !params || !labels || !out || !loss_out || !grad_params || !workspace) return set_err(MSHGNN_EINVAL, "null argument to mshgnn_step_ce");
    if (batch < 1 || batch > (1 << 24)) return set_err(MSHGNN_EINVAL, "batch must be in [1, 2^24]");
    for (int t = 0; t < p->n_types; ++t) if (!x[t]) return set_err(MSHGNN_EINVAL, "null input tensor");
    if (p->hp.d.out_channels != 2) return set_err(MSHGNN_EINVAL, "mshgnn_step_ce: the classification wrappers have two logits per foot");
    if (const int64_t cw = step_chunk_windows(p, batch)) {
        const int64_t n_out = p->hp.d.type_nodes[p->hp.d.out_type];
        return chunked_step(p, x, x_pitch, batch, cw, [&](const void* const* xc, int64_t w0, int64_t bc) {
            return mshgnn_step_ce(p, xc, x_pitch, params, labels + w0 * n_out, out + w0 * n_out * 2, loss_out, grad_params, workspace, bc, stream); });
    }
    hipStream_t st = (hipStream_t)stream;
    // bf16 plan with the fused stack kernels: decoder, cross entropy and decoder backward in the tail of the forward kernel; every other plan: the
    // two-call sequence (mshgnn_forward + mshgnn_backward_ce)
    if (!p->gen && p->hp.d.dtype == MSHGNN_BF16 && p->use_fused) {
        bool stack_done = false;
        int rc = forward_impl<__bf16>(p, x, x_pitch, params, out, (char*)workspace, batch, 1, st, nullptr, nullptr, labels, &stack_done);
        if (rc) return rc;
        return backward_impl<__bf16>(p, x, x_pitch, params, nullptr, grad_params, (char*)workspace, batch, st, out, nullptr, loss_out, labels, true, -1, nullptr, stack_done);
    }
    if (!p->gen && p->hp.d.dtype == MSHGNN_BF16X3) {      // split plan: decoder, cross entropy and decoder backward in the tail of its fused forward kernel as well
        bool stack_done = false;
        int rc = x3_forward(p, x, x_pitch, params, out, (char*)workspace, batch, 1, st, nullptr, nullptr, &stack_done, labels);
        if (rc) return rc;
        return x3_backward(p, x, x_pitch, params, nullptr, grad_params, (char*)workspace, batch, st, out, nullptr, loss_out, labels, true, -1, stack_done);
    }
    int rc = mshgnn_forward(p, x, x_pitch, params, out, workspace, batch, 1, stream);
    if (rc) return rc;
    return mshgnn_backward_ce(p, x, x_pitch, params, out, labels, loss_out, grad_params, workspace, batch, stream);
}

// ---- the caller's own fp64 / fp32 tensors as inputs (the reference's datasets produce fp64, gnnLightning.py:1183): the encoder converts in registers and
// writes the plan-dtype rows the weight-gradient kernel needs on the side -- no separate cast + re-pitch pass.  Thin wrappers: they publish the source
// descriptor to the encoder launch of the plain entry point they forward to (same thread, synchronous on the host).
namespace {
struct WideGuard {
    WideSrc w{};
    int rc = MSHGNN_OK;
    WideGuard(const mshgnn_plan* p, int src_bytes, const void* const* src, const int64_t* src_pitch, const char* who) {
        if (!p || !src) { rc = set_err(MSHGNN_EINVAL, std::string("null argument to ") + who); return; }
        if (src_bytes != 4 && src_bytes != 8) { rc = set_err(MSHGNN_EINVAL, std::string(who) + ": src_bytes must be 4 (fp32) or 8 (fp64)"); return; }
        if (p->gen || p->hp.d.dtype == MSHGNN_F32) { rc = set_err(MSHGNN_EUNSUPPORTED, std::string(who) + ": wide source rows run on the bf16 and split-bf16 plans of the LDS-resident kernels"); return; }
        w.bytes = src_bytes;
        for (int t = 0; t < p->n_types; ++t) {
            if (!src[t]) { rc = set_err(MSHGNN_EINVAL, std::string(who) + ": null source tensor"); return; }
            w.p[t] = src[t]; w.pitch[t] = src_pitch ? src_pitch[t] : p->hp.d.type_width[t];
            if (w.pitch[t] < p->hp.d.type_width[t]) { rc = set_err(MSHGNN_EINVAL, std::string(who) + ": src_pitch smaller than the feature width"); return; }
            if (((uintptr_t)src[t] % (src_bytes == 8 ? 8 : 4)) != 0) { rc = set_err(MSHGNN_EINVAL, std::string(who) + ": source tensor not aligned to its element size"); return; }
            // fp32 rows are read in 8-byte units where the width allows it: rows must then start on 8 bytes
            if (src_bytes == 4 && (p->hp.d.type_width[t] & 1) == 0 && ((((uintptr_t)src[t]) | (uintptr_t)(w.pitch[t] * 4)) & 7) != 0) {
                rc = set_err(MSHGNN_EINVAL, std::string(who) + ": fp32 source rows of an even width must start 8-byte aligned (base and pitch)"); return; }
        }
        g_wide_src = &w;
    }
    ~WideGuard() { g_wide_src = nullptr; }
};
}  // namespace
extern "C" int mshgnn_forward_src(const mshgnn_plan* p, int src_bytes, const void* const* src, const int64_t* src_pitch, void* const* x_rows, const int64_t* x_pitch,
                                  const float* params, float* out, void* workspace, int64_t batch, int training, void* stream) {
    if (!x_rows) return set_err(MSHGNN_EINVAL, "mshgnn_forward_src: x_rows is null (the plan-dtype rows the encoder materialises)");
    WideGuard g(p, src_bytes, src, src_pitch, "mshgnn_forward_src");
    if (g.rc) return g.rc;
    return mshgnn_forward(p, x_rows, x_pitch, params, out, workspace, batch, training, stream);
}
extern "C" int mshgnn_step_mse_src(const mshgnn_plan* p, int src_bytes, const void* const* src, const int64_t* src_pitch, void* const* x_rows, const int64_t* x_pitch,
                                   const float* params, const float* y, float* out, float* loss_out, float* grad_params, void* workspace, int64_t batch, void* stream) {
    if (!x_rows) return set_err(MSHGNN_EINVAL, "mshgnn_step_mse_src: x_rows is null (the plan-dtype rows the encoder materialises)");
    WideGuard g(p, src_bytes, src, src_pitch, "mshgnn_step_mse_src");
    if (g.rc) return g.rc;
    return mshgnn_step_mse(p, x_rows, x_pitch, params, y, out, loss_out, grad_params, workspace, batch, stream);
}
extern "C" int mshgnn_step_ce_src(const mshgnn_plan* p, int src_bytes, const void* const* src, const int64_t* src_pitch, void* const* x_rows, const int64_t* x_pitch,
                                  const float* params, const int32_t* labels, float* out, float* loss_out, float* grad_params, void* workspace, int64_t batch, void* stream) {
    if (!x_rows) return set_err(MSHGNN_EINVAL, "mshgnn_step_ce_src: x_rows is null (the plan-dtype rows the encoder materialises)");
    WideGuard g(p, src_bytes, src, src_pitch, "mshgnn_step_ce_src");
    if (g.rc) return g.rc;
    return mshgnn_step_ce(p, x_rows, x_pitch, params, labels, out, loss_out, grad_params, workspace, batch, stream);
}

extern "C" int mshgnn_step_mse_phase(const mshgnn_plan* p, const void* const* x, const int64_t* x_pitch, const float* params, const float* y,
                                     float* out, float* loss_out, float* grad_params, void* workspace, int64_t batch, int phase, void* stream) {
    if (!p || !x || !params || !y || !out || !loss_out || !grad_params || !workspace) return set_err(MSHGNN_EINVAL, "null argument to mshgnn_step_mse_phase");
    if (batch < 1 || batch > (1 << 24)) return set_err(MSHGNN_EINVAL, "batch must be in [1, 2^24]");
    if (phase != 0 && phase != 1) return set_err(MSHGNN_EINVAL, "phase must be 0 or 1");
    if (p->gen || p->hp.grad_split < 0) return set_err(MSHGNN_EUNSUPPORTED, "this plan has no two-phase gradient split");
    for (int t = 0; t < p->n_types; ++t) if (!x[t]) return set_err(MSHGNN_EINVAL, "null input tensor");
    hipStream_t st = (hipStream_t)stream;
    if (p->hp.d.dtype == MSHGNN_BF16X3) {
        if (phase == 0) { int rc = x3_forward(p, x, x_pitch, params, out, (char*)workspace, batch, 1, st, y); if (rc) return rc; }
        return x3_backward(p, x, x_pitch, params, nullptr, grad_params, (char*)workspace, batch, st, out, y, loss_out, nullptr, true, phase);
    }
    const bool f32 = p->hp.d.dtype == MSHGNN_F32, fused = !f32 && p->use_fused;
    bool stack_done = false;      // phase 0 on the fused plans: both sweeps of the stack in the one-launch step kernel, as mshgnn_step_mse (the specialised one where the plan has it)
    if (phase == 0) {
        int rc = f32 ? forward_impl<float>(p, x, x_pitch, params, out, (char*)workspace, batch, 1, st)
                     : forward_impl<__bf16>(p, x, x_pitch, params, out, (char*)workspace, batch, 1, st, fused ? y : nullptr, nullptr, nullptr, fused ? &stack_done : nullptr);
        if (rc) return rc;
    }
    return f32 ? backward_impl<float>(p, x, x_pitch, params, nullptr, grad_params, (char*)workspace, batch, st, out, y, loss_out, nullptr, false, phase)
               : backward_impl<__bf16>(p, x, x_pitch, params, nullptr, grad_params, (char*)workspace, batch, st, out, y, loss_out, nullptr, fused, phase, nullptr, stack_done);
}

extern "C" int mshgnn_backward_ce(const mshgnn_plan* p, const void* const* x, const int64_t* x_pitch, const float* params, const float* out,
                                  const int32_t* labels, float* loss_out, float* grad_params, void* workspace, int64_t batch, void* stream) {
    if (!p || !x || !params || !out || !labels || !loss_out || !grad_params || !workspace) return set_err(MSHGNN_EINVAL, "null argument to mshgnn_backward_ce");
    if (batch < 1 || batch > (1 << 24)) return set_err(MSHGNN_EINVAL, "batch must be in [1, 2^24]");
    if (p->hp.d.out_channels != 2) return set_err(MSHGNN_EINVAL, "mshgnn_backward_ce needs a 2-logit (contact classification) plan");
    if (p->gen) return gen_backward(p, x, x_pitch, params, nullptr, grad_params, (char*)workspace, batch, (hipStream_t)stream, out, nullptr, loss_out, labels);
    if (p->hp.d.dtype == MSHGNN_BF16X3) return x3_backward(p, x, x_pitch, params, nullptr, grad_params, (char*)workspace, batch, (hipStream_t)stream, out, nullptr, loss_out, labels, false, -1);
    if (p->hp.d.dtype == MSHGNN_F32) return backward_impl<float>(p, x, x_pitch, params, nullptr, grad_params, (char*)workspace, batch, (hipStream_t)stream, out, nullptr, loss_out, labels);
    return backward_impl<__bf16>(p, x, x_pitch, params, nullptr, grad_params, (char*)workspace, batch, (hipStream_t)stream, out, nullptr, loss_out, labels);
}

// ------------------------------------------------------------------------------------------------------
// Adam on the flat fp32 buffers (configure_optimizers: optim.Adam(self.parameters(), lr), gnnLightning.py:258-265;
// torch defaults beta=(0.9, 0.999), eps=1e-8, no weight decay, no amsgrad).  SURVEY.md section 8(f) row 2.
// ------------------------------------------------------------------------------------------------------
__global__ void k_adam(float* p, const float* g, float* m, float* v, int64_t n, float lr, float b1, float b2, float eps,
                       float bc1, float bc2_sqrt, float gscale) {
    for (int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < n; i += (int64_t)gridDim.x * blockDim.x * 4) {
        if (i + 4 <= n) {
            f32x4 pp = *reinterpret_cast<f32x4*>(p + i), gg = *reinterpret_cast<const f32x4*>(g + i) * gscale;
            f32x4 mm = *reinterpret_cast<f32x4*>(m + i), vv = *reinterpret_cast<f32x4*>(v + i);
            mm = b1 * mm + (1.f - b1) * gg;
            vv = b2 * vv + (1.f - b2) * gg * gg;
#pragma unroll
            for (int e = 0; e < 4; ++e) pp[e] -= lr / bc1 * mm[e] / (sqrtf(vv[e]) / bc2_sqrt + eps);
            *reinterpret_cast<f32x4*>(p + i) = pp; *reinterpret_cast<f32x4*>(m + i) = mm; *reinterpret_cast<f32x4*>(v + i) = vv;
        } else {
            for (int64_t k = i; k < n; ++k) {
                const float gg = g[k] * gscale;
                m[k] = b1 * m[k] + (1.f - b1) * gg; v[k] = b2 * v[k] + (1.f - b2) * gg * gg;
                p[k] -= lr / bc1 * m[k] / (sqrtf(v[k]) / bc2_sqrt + eps);
            }
        }
    }
}

extern "C" int mshgnn_adam_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int64_t n, int64_t step,
                                float lr, float beta1, float beta2, float eps, float grad_scale, void* stream) {
    if (!params || !grads || !exp_avg || !exp_avg_sq || n < 1 || step < 1) return set_err(MSHGNN_EINVAL, "bad argument to mshgnn_adam_step");
    if (((uintptr_t)params | (uintptr_t)grads | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq) & 15) return set_err(MSHGNN_EINVAL, "adam buffers must be 16-byte aligned");
    const float bc1 = 1.0f - powf(beta1, (float)step), bc2 = 1.0f - powf(beta2, (float)step);
    const int blocks = (int)std::min<int64_t>((n / 4 + 255) / 256 + 1, 2048);
    hipLaunchKernelGGL(k_adam, dim3(blocks), dim3(256), 0, (hipStream_t)stream, params, grads, exp_avg, exp_avg_sq, n, lr, beta1, beta2, eps,
                       bc1, sqrtf(bc2), grad_scale);
    HIPCHK(hipGetLastError());
    return MSHGNN_OK;
}

// The same update with the step count on the DEVICE (capturable in a HIP graph: nothing of the bias corrections is baked into the launch arguments).  The kernel
// reads t = *step_count + 1 and derives 1 - beta^t itself; a one-thread launch behind it stores t.  FlatAdam(graph_safe=True), wrappers.GraphedTrainingStep.
__global__ void k_adam_counted(float* p, const float* g, float* m, float* v, int64_t n, const int64_t* step_count, float lr, float b1, float b2, float eps, float gscale) {
    const float t = (float)(*step_count + 1);
    const float bc1 = 1.0f - powf(b1, t), bc2_sqrt = sqrtf(1.0f - powf(b2, t));
    for (int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < n; i += (int64_t)gridDim.x * blockDim.x * 4) {
        if (i + 4 <= n) {
            f32x4 pp = *reinterpret_cast<f32x4*>(p + i), gg = *reinterpret_cast<const f32x4*>(g + i) * gscale;
            f32x4 mm = *reinterpret_cast<f32x4*>(m + i), vv = *reinterpret_cast<f32x4*>(v + i);
            mm = b1 * mm + (1.f - b1) * gg;
            vv = b2 * vv + (1.f - b2) * gg * gg;
#pragma unroll
            for (int e = 0; e < 4; ++e) pp[e] -= lr / bc1 * mm[e] / (sqrtf(vv[e]) / bc2_sqrt + eps);
            *reinterpret_cast<f32x4*>(p + i) = pp; *reinterpret_cast<f32x4*>(m + i) = mm; *reinterpret_cast<f32x4*>(v + i) = vv;
        } else {
            for (int64_t k = i; k < n; ++k) {
                const float gg = g[k] * gscale;
                m[k] = b1 * m[k] + (1.f - b1) * gg; v[k] = b2 * v[k] + (1.f - b2) * gg * gg;
                p[k] -= lr / bc1 * m[k] / (sqrtf(v[k]) / bc2_sqrt + eps);
            }
        }
    }
}
__global__ void k_step_count_inc(int64_t* step_count) { *step_count += 1; }

extern "C" int mshgnn_adam_step_counted(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int64_t n, int64_t* step_count,
                                        float lr, float beta1, float beta2, float eps, float grad_scale, void* stream) {
    if (!params || !grads || !exp_avg || !exp_avg_sq || !step_count || n < 1) return set_err(MSHGNN_EINVAL, "bad argument to mshgnn_adam_step_counted");
    if (((uintptr_t)params | (uintptr_t)grads | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq) & 15) return set_err(MSHGNN_EINVAL, "adam buffers must be 16-byte aligned");
    const int blocks = (int)std::min<int64_t>((n / 4 + 255) / 256 + 1, 2048);
    hipLaunchKernelGGL(k_adam_counted, dim3(blocks), dim3(256), 0, (hipStream_t)stream, params, grads, exp_avg, exp_avg_sq, n, step_count, lr, beta1, beta2, eps, grad_scale);
    hipLaunchKernelGGL(k_step_count_inc, dim3(1), dim3(1), 0, (hipStream_t)stream, step_count);
    HIPCHK(hipGetLastError());
    return MSHGNN_OK;
}

extern "C" int mshgnn_mse_loss(const float* out, const float* y, int64_t n, float* loss_out, float* grad_out, void* stream) {
    if (!out || !y || !loss_out || n < 1) return set_err(MSHGNN_EINVAL, "bad argument to mshgnn_mse_loss");
    hipStream_t st = (hipStream_t)stream;
    HIPCHK(hipMemsetAsync(loss_out, 0, sizeof(float), st));
    const int blocks = (int)std::min<int64_t>((n + 255) / 256, 1024);
    hipLaunchKernelGGL(k_mse, dim3(blocks), dim3(256), 0, st, out, y, n, loss_out, grad_out);
    HIPCHK(hipGetLastError());
    return MSHGNN_OK;
}

// Stand-alone contact cross entropy of the classification wrappers (gnnLightning.py:640-648, customMetrics.py:6-25: CrossEntropyLoss over
// the [rows, 2] per-foot logits, batch value = sum / rows) with its gradient (softmax - onehot) / rows.  One thread per row.
__global__ void k_ce(const float* logits, const int32_t* labels, int64_t rows, float* loss, float* gout) {
    __shared__ float red[4];
    float s = 0.f;
    const float inv = 1.0f / (float)rows;
    for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < rows; r += (int64_t)gridDim.x * blockDim.x) {
        const float l0 = logits[2 * r], l1 = logits[2 * r + 1];
        const float m = fmaxf(l0, l1);
        const float e0 = expf(l0 - m), e1 = expf(l1 - m), z = e0 + e1;
        const int lab = labels[r] != 0;
        s += logf(z) + m - (lab ? l1 : l0);
        if (gout) {
            gout[2 * r] = (e0 / z - (lab ? 0.f : 1.f)) * inv;
            gout[2 * r + 1] = (e1 / z - (lab ? 1.f : 0.f)) * inv;
        }
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) s += __shfl_xor(s, m, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(loss, (red[0] + red[1] + red[2] + red[3]) * inv);
}

extern "C" int mshgnn_ce_loss(const float* logits, const int32_t* labels, int64_t rows, float* loss_out, float* grad_out, void* stream) {
    if (!logits || !labels || !loss_out || rows < 1) return set_err(MSHGNN_EINVAL, "bad argument to mshgnn_ce_loss");
    hipStream_t st = (hipStream_t)stream;
    HIPCHK(hipMemsetAsync(loss_out, 0, sizeof(float), st));
    const int blocks = (int)std::min<int64_t>((rows + 255) / 256, 1024);
    hipLaunchKernelGGL(k_ce, dim3(blocks), dim3(256), 0, st, logits, labels, rows, loss_out, grad_out);
    HIPCHK(hipGetLastError());
    return MSHGNN_OK;
}

// ------------------------------------------------------------------------------------------------------
// Step metrics of the Lightning wrappers, on device.  The reference keeps
// torchmetrics states that are plain sums across steps (gnnLightning.py:52-63, customMetrics.py:11-54); these kernels
// ADD one step's sums into caller-owned state buffers.  One workgroup, fixed reduction order: deterministic.
// ------------------------------------------------------------------------------------------------------
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
    return v;
}
__device__ __forceinline__ long long wave_sum(long long v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
    return v;
}

// Both kernels run on up to MET_BLOCKS workgroups: every workgroup leaves its partial sums in the caller's scratch, takes a ticket, and the
// workgroup that draws the last ticket adds the partials IN INDEX ORDER (bit-reproducible whatever the arrival order) and resets the
// ticket.  The one-workgroup entry points (mshgnn_metrics_regression / _classification, no scratch) run the same kernels with one block.
constexpr int MET_COUNTS = 18, MET_BLOCKS = 64, MET_THREADS = 256;
struct MetScratch {
    unsigned int ticket, pad;
    double f[MET_BLOCKS][2];
    long long c[MET_BLOCKS][MET_COUNTS];
};
static_assert(sizeof(MetScratch) <= MSHGNN_METRICS_SCRATCH_BYTES, "include/mshgnn.h promises this scratch size");

__device__ __forceinline__ void met_store(double* p, double v) { __hip_atomic_store(reinterpret_cast<long long*>(p), __double_as_longlong(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ double met_load(const double* p) { return __longlong_as_double(__hip_atomic_load(reinterpret_cast<const long long*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)); }
__device__ __forceinline__ void met_store(long long* p, long long v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ long long met_load(const long long* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// thread 0 of every workgroup, after its partials are stored: true in the workgroup that arrives last (sc == nullptr: a one-block launch)
__device__ __forceinline__ bool met_last_block(MetScratch* sc) {
    if (!sc) return true;
    __atomic_thread_fence(__ATOMIC_RELEASE);        // (agent scope: the partials reach memory every XCD's L2 sees)
    const unsigned int t = __hip_atomic_fetch_add(&sc->ticket, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
    if (t != gridDim.x - 1) return false;
    __atomic_thread_fence(__ATOMIC_ACQUIRE);
    return true;
}

// regression (calculate_losses_step, gnnLightning.py:124-130): sums of (pred - y)^2, |pred - y| and n; `batch` (nullable) receives this
// step's sums (overwritten), `epoch` (nullable) has them added; gout (nullable) = d mean((pred - y)^2) / d pred = 2 (pred - y) / n
template <int NT> __global__ __launch_bounds__(NT) void k_metrics_reg(const float* pred, const float* y, int64_t n, double* batch, double* epoch, float* gout,
                                                             MetScratch* sc) {
    __shared__ double r0[NT / 64], r1[NT / 64], pf[MET_BLOCKS][2];
    __shared__ int s_last;
    double s = 0.0, a = 0.0;
    const double inv2 = 2.0 / (double)n;
    for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < n; i += (int64_t)gridDim.x * NT) {
        const double dlt = (double)pred[i] - (double)y[i];
        s += dlt * dlt; a += fabs(dlt);
        if (gout) gout[i] = (float)(dlt * inv2);
    }
    s = wave_sum(s); a = wave_sum(a);
    if ((threadIdx.x & 63) == 0) { r0[threadIdx.x >> 6] = s; r1[threadIdx.x >> 6] = a; }
    __syncthreads();
    double ts = 0.0, ta = 0.0;
    if (threadIdx.x == 0) {
        for (int k = 0; k < NT / 64; ++k) { ts += r0[k]; ta += r1[k]; }
        if (sc) { met_store(&sc->f[blockIdx.x][0], ts); met_store(&sc->f[blockIdx.x][1], ta); }
        s_last = met_last_block(sc) ? 1 : 0;
    }
    if (sc) {
        __syncthreads();
        if (!s_last) return;
        __atomic_thread_fence(__ATOMIC_ACQUIRE);
        // the last workgroup: one partial per thread (all loads in flight at once), then thread 0 adds them in workgroup order
        if (threadIdx.x < gridDim.x) { pf[threadIdx.x][0] = met_load(&sc->f[threadIdx.x][0]); pf[threadIdx.x][1] = met_load(&sc->f[threadIdx.x][1]); }
        __syncthreads();
        if (threadIdx.x == 0) {
            ts = ta = 0.0;
            for (unsigned b = 0; b < gridDim.x; ++b) { ts += pf[b][0]; ta += pf[b][1]; }
            __hip_atomic_store(&sc->ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    if (threadIdx.x == 0) {
        if (batch) {      // the sums, then the step's published values: MSE, RMSE, L1 (gnnLightning.py:124-130)
            batch[0] = ts; batch[1] = ta; batch[2] = (double)n; batch[3] = ts / (double)n; batch[4] = sqrt(ts / (double)n); batch[5] = ta / (double)n;
            batch[6] = batch[7] = 0.0;
        }
        if (epoch) { epoch[0] += ts; epoch[1] += ta; epoch[2] += (double)n; }
    }
}

// classification (gnnLightning.py:132-151, 285-348): logits [B*4][2], labels [B][4] in {0,1}.
//   ce_state[0] += sum of per-foot cross entropies, ce_state[1] += 4 B                      (customMetrics.py:17-24)
//   counts[0] += B, counts[1] += windows whose 16-class argmax equals the label state       (Accuracy, 16 classes)
//   counts[2 + 4 k + {0,1,2,3}] += tp, fp, fn, tn of leg k                                   (BinaryF1Score)
// The 16-class probabilities are the reference's products (p or 1 - p per foot, ((f0 f1)(f2 f3)), first maximum wins).
// ce_b / counts_b (nullable): this step's sums, overwritten; ce_state / counts (nullable): added into; gout (nullable) [B*4][2] = d ce / d logits
// = (softmax - onehot) / (4 B)
template <int NT> __global__ __launch_bounds__(NT) void k_metrics_cls(const float* logits, const int32_t* y, int64_t B, double* ce_b, long long* counts_b,
                                                             double* ce_state, long long* counts, float* gout, MetScratch* sc) {
    __shared__ double rce[NT / 64];
    __shared__ long long rc[NT / 64][MET_COUNTS], pc[MET_BLOCKS][MET_COUNTS];
    __shared__ double pce[MET_BLOCKS];
    __shared__ int s_last;
    double ce = 0.0;
    long long c[MET_COUNTS];
#pragma unroll
    for (int k = 0; k < MET_COUNTS; ++k) c[k] = 0;
    for (int64_t w = (int64_t)blockIdx.x * NT + threadIdx.x; w < B; w += (int64_t)gridDim.x * NT) {
        double p1[4];
        int state = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const double l0 = (double)logits[(w * 4 + k) * 2], l1 = (double)logits[(w * 4 + k) * 2 + 1];
            const double m = fmax(l0, l1), e0 = exp(l0 - m), e1 = exp(l1 - m), se = e0 + e1;
            const int lab = y[w * 4 + k] != 0;
            ce += (m + log(se)) - (lab ? l1 : l0);
            const double p0 = e0 / se; p1[k] = e1 / se;
            if (gout) {
                const double inv = 1.0 / (double)(4 * B);
                gout[(w * 4 + k) * 2] = (float)((p0 - (lab ? 0.0 : 1.0)) * inv);
                gout[(w * 4 + k) * 2 + 1] = (float)((p1[k] - (lab ? 1.0 : 0.0)) * inv);
            }
            const int pred = p1[k] > p0 ? 1 : 0;             // argmax over (p0, p1): the first maximum wins
            const int cell = pred ? (lab ? 0 : 1) : (lab ? 2 : 3);      // tp, fp, fn, tn -- added by compare, not by a run-time index (the counters stay in registers)
#pragma unroll
            for (int j = 0; j < 4; ++j) c[2 + 4 * k + j] += (cell == j);
            state = state * 2 + lab;
        }
        int best = 0; double bestv = -1.0;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const double f0 = (j & 8) ? p1[0] : 1.0 - p1[0], f1 = (j & 4) ? p1[1] : 1.0 - p1[1];
            const double f2 = (j & 2) ? p1[2] : 1.0 - p1[2], f3 = (j & 1) ? p1[3] : 1.0 - p1[3];
            const double v = (f0 * f1) * (f2 * f3);
            if (v > bestv) { bestv = v; best = j; }
        }
        c[0] += 1; c[1] += (best == state);
    }
    ce = wave_sum(ce);
#pragma unroll
    for (int k = 0; k < MET_COUNTS; ++k) c[k] = wave_sum(c[k]);
    if ((threadIdx.x & 63) == 0) {
        rce[threadIdx.x >> 6] = ce;
#pragma unroll
        for (int k = 0; k < MET_COUNTS; ++k) rc[threadIdx.x >> 6][k] = c[k];
    }
    __syncthreads();
    // this workgroup's sums: thread 0 the cross entropy, threads 0..17 one count each
    double tce = 0.0; long long tc = 0;
    if (threadIdx.x == 0) for (int k = 0; k < NT / 64; ++k) tce += rce[k];
    if (threadIdx.x < MET_COUNTS) for (int k = 0; k < NT / 64; ++k) tc += rc[k][threadIdx.x];
    if (sc) {
        if (threadIdx.x == 0) met_store(&sc->f[blockIdx.x][0], tce);
        if (threadIdx.x < MET_COUNTS) met_store(&sc->c[blockIdx.x][threadIdx.x], tc);
        __syncthreads();                                    // every partial of this workgroup is stored before thread 0 takes the ticket
        if (threadIdx.x == 0) s_last = met_last_block(sc) ? 1 : 0;
        __syncthreads();
        if (!s_last) return;
        __atomic_thread_fence(__ATOMIC_ACQUIRE);
        // the last workgroup: thread b fetches workgroup b's partials (all loads in flight at once), then one thread per sum adds them in workgroup order
        if (threadIdx.x < gridDim.x) {
            pce[threadIdx.x] = met_load(&sc->f[threadIdx.x][0]);
#pragma unroll
            for (int k = 0; k < MET_COUNTS; ++k) pc[threadIdx.x][k] = met_load(&sc->c[threadIdx.x][k]);
        }
        __syncthreads();
        tce = 0.0; tc = 0;
        if (threadIdx.x == 0) for (unsigned b = 0; b < gridDim.x; ++b) tce += pce[b];
        if (threadIdx.x < MET_COUNTS) for (unsigned b = 0; b < gridDim.x; ++b) tc += pc[b][threadIdx.x];
        if (threadIdx.x == 0) __hip_atomic_store(&sc->ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (threadIdx.x == 0) {
        if (ce_b) { ce_b[0] = tce; ce_b[1] = (double)(4 * B); ce_b[2] = (double)(float)tce / (double)(4 * B); }      // [2]: this step's CE, `summed_loss.float() / total_num` (customMetrics.py:24)
        if (ce_state) { ce_state[0] += tce; ce_state[1] += (double)(4 * B); }
    }
    if (threadIdx.x < MET_COUNTS) {
        if (counts_b) counts_b[threadIdx.x] = tc;
        if (counts) counts[threadIdx.x] += tc;
        if (ce_b) pc[0][threadIdx.x] = tc;                  // (pc: free again -- every partial has been added)
    }
    if (ce_b) {      // the step's published values next to its sums: [3] 16-class accuracy, [4..7] F1 of leg 0..3 (customMetrics.py:51-54, 0/0 -> 0)
        __syncthreads();
        if (threadIdx.x == 0) ce_b[3] = (double)pc[0][1] / (double)pc[0][0];
        if (threadIdx.x >= 1 && threadIdx.x <= 4) {
            const int k = threadIdx.x - 1;
            const double tp = (double)pc[0][2 + 4 * k], fp = (double)pc[0][3 + 4 * k], fn = (double)pc[0][4 + 4 * k];
            const double precision = tp / (tp + fp), recall = tp / (tp + fn);
            const double f1 = 2.0 * (precision * recall) / (precision + recall);
            ce_b[4 + k] = f1 != f1 ? 0.0 : f1;
        }
    }
}

// centroidal-momentum wrappers (gnnLightning_com.py:96-121): y / y_pred [B][nb][6] = per base node (lin(3) | ang(3)), standardised.
//   state[0] += sum sq err of the lin halves, [1] += of the ang halves, [2] += 3 nb B, [3] += 3 nb B,
//   [4] += sum over windows of cos(lin_pred, lin) of base node 0 after un-standardising (v * y_std + y_mean), [5] += the same for ang,
//   [6] += B.  Cosine similarity as torch.nn.CosineSimilarity(dim=1, eps=1e-8): sum (a / max(|a|, eps)) (b / max(|b|, eps))
//   (customMetrics.py:56-95).  One thread per window; multi-workgroup with the ticket scheme above.
struct MetScratchCom { unsigned int ticket, pad; double f[MET_BLOCKS][4]; };
static_assert(sizeof(MetScratchCom) <= MSHGNN_METRICS_SCRATCH_BYTES, "include/mshgnn.h promises this scratch size");
struct ComStats { double mean[6], std[6]; };

__global__ __launch_bounds__(MET_THREADS) void k_metrics_com(const float* pred, const float* y, int64_t B, int nb, ComStats st, double* batch, double* epoch,
                                                             MetScratchCom* sc) {
    __shared__ double r[MET_THREADS / 64][4], pf[MET_BLOCKS][4];
    __shared__ int s_last;
    double a[4] = {0.0, 0.0, 0.0, 0.0};      // sq lin, sq ang, cos lin, cos ang
    for (int64_t w = (int64_t)blockIdx.x * MET_THREADS + threadIdx.x; w < B; w += (int64_t)gridDim.x * MET_THREADS) {
        for (int b = 0; b < nb; ++b)
#pragma unroll
            for (int k = 0; k < 6; ++k) {
                const double d = (double)pred[(w * nb + b) * 6 + k] - (double)y[(w * nb + b) * 6 + k];
                a[k < 3 ? 0 : 1] += d * d;
            }
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            double pp = 0.0, yy = 0.0, py = 0.0, pv[3], yv[3];
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                pv[k] = (double)pred[w * nb * 6 + 3 * h + k] * st.std[3 * h + k] + st.mean[3 * h + k];
                yv[k] = (double)y[w * nb * 6 + 3 * h + k] * st.std[3 * h + k] + st.mean[3 * h + k];
                pp += pv[k] * pv[k]; yy += yv[k] * yv[k];
            }
            const double pn = fmax(sqrt(pp), 1e-8), yn = fmax(sqrt(yy), 1e-8);
#pragma unroll
            for (int k = 0; k < 3; ++k) py += (pv[k] / pn) * (yv[k] / yn);
            a[2 + h] += py;
        }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) a[k] = wave_sum(a[k]);
    if ((threadIdx.x & 63) == 0)
#pragma unroll
        for (int k = 0; k < 4; ++k) r[threadIdx.x >> 6][k] = a[k];
    __syncthreads();
    double t[4] = {0.0, 0.0, 0.0, 0.0};
    if (threadIdx.x == 0) {
        for (int v = 0; v < MET_THREADS / 64; ++v)
#pragma unroll
            for (int k = 0; k < 4; ++k) t[k] += r[v][k];
        if (sc)
#pragma unroll
            for (int k = 0; k < 4; ++k) met_store(&sc->f[blockIdx.x][k], t[k]);
        s_last = 1;
        if (sc) {
            __atomic_thread_fence(__ATOMIC_RELEASE);
            const unsigned int tk = __hip_atomic_fetch_add(&sc->ticket, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
            s_last = tk == gridDim.x - 1;
        }
    }
    if (sc) {
        __syncthreads();
        if (!s_last) return;
        __atomic_thread_fence(__ATOMIC_ACQUIRE);
        if (threadIdx.x < gridDim.x)
#pragma unroll
            for (int k = 0; k < 4; ++k) pf[threadIdx.x][k] = met_load(&sc->f[threadIdx.x][k]);
        __syncthreads();
        if (threadIdx.x == 0) {
#pragma unroll
            for (int k = 0; k < 4; ++k) t[k] = 0.0;
            for (unsigned b = 0; b < gridDim.x; ++b)
#pragma unroll
                for (int k = 0; k < 4; ++k) t[k] += pf[b][k];
            __hip_atomic_store(&sc->ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    if (threadIdx.x == 0) {
        const double n3 = 3.0 * (double)nb * (double)B;
        const double v[8] = {t[0], t[1], n3, n3, t[2], t[3], (double)B, 0.0};
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            if (batch) batch[k] = v[k];
            if (epoch) epoch[k] += v[k];
        }
    }
}

// GRF body frame -> world frame (gnnLightning.py:663-676): quat = world->body rotation, scalar-last (x, y, z, w) as scipy's
// Rotation.from_quat takes it (normalised here as scipy does); world = R(quat)^-1 f for each of the 4 feet.
__global__ void k_grf_to_world(const float* quat, const float* body, float* world, int64_t B) {
    const int64_t w = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= B) return;
    double x = quat[w * 4], yq = quat[w * 4 + 1], z = quat[w * 4 + 2], s = quat[w * 4 + 3];
    const double nrm = sqrt(x * x + yq * yq + z * z + s * s);
    x /= nrm; yq /= nrm; z /= nrm; s /= nrm;
    // R = matrix of the unit quaternion; its inverse is the transpose
    const double R[3][3] = {{1 - 2 * (yq * yq + z * z), 2 * (x * yq - z * s), 2 * (x * z + yq * s)},
                            {2 * (x * yq + z * s), 1 - 2 * (x * x + z * z), 2 * (yq * z - x * s)},
                            {2 * (x * z - yq * s), 2 * (yq * z + x * s), 1 - 2 * (x * x + yq * yq)}};
    for (int f = 0; f < 4; ++f) {
        const double b0 = body[w * 12 + f * 3], b1 = body[w * 12 + f * 3 + 1], b2 = body[w * 12 + f * 3 + 2];
#pragma unroll
        for (int i = 0; i < 3; ++i) world[w * 12 + f * 3 + i] = (float)(R[0][i] * b0 + R[1][i] * b1 + R[2][i] * b2);
    }
}

static int met_blocks(int64_t items, int per_thread) {
    return (int)std::max<int64_t>(1, std::min<int64_t>(MET_BLOCKS, (items + per_thread * MET_THREADS - 1) / (per_thread * MET_THREADS)));
}

extern "C" int mshgnn_metrics_regression(const float* y_pred, const float* y, int64_t n, double* state, void* stream) {
    if (!y_pred || !y || !state || n < 1) return set_err(MSHGNN_EINVAL, "bad argument to mshgnn_metrics_regression");
    hipLaunchKernelGGL(k_metrics_reg<1024>, dim3(1), dim3(1024), 0, (hipStream_t)stream, y_pred, y, n, (double*)nullptr, state, (float*)nullptr, (MetScratch*)nullptr);
    HIPCHK(hipGetLastError());
    return MSHGNN_OK;
}

extern "C" int mshgnn_metrics_regression_step(const float* y_pred, const float* y, int64_t n, double* batch_state, double* epoch_state, float* grad_out,
                                              void* scratch, void* stream) {
    if (!y_pred || !y || (!batch_state && !epoch_state) || !scratch || n < 1) return set_err(MSHGNN_EINVAL, "bad argument to mshgnn_metrics_regression_step");
    hipLaunchKernelGGL(k_metrics_reg<MET_THREADS>, dim3(met_blocks(n, 16)), dim3(MET_THREADS), 0, (hipStream_t)stream, y_pred, y, n, batch_state, epoch_state, grad_out,
                       reinterpret_cast<MetScratch*>(scratch));
    HIPCHK(hipGetLastError());
    return MSHGNN_OK;
}

extern "C" int mshgnn_metrics_classification(const float* logits, const int32_t* y, int64_t batch, double* ce_state, int64_t* counts, void* stream) {
    if (!logits || !y || !ce_state || !counts || batch < 1) return set_err(MSHGNN_EINVAL, "bad argument to mshgnn_metrics_classification");
    hipLaunchKernelGGL(k_metrics_cls<1024>, dim3(1), dim3(1024), 0, (hipStream_t)stream, logits, y, batch, (double*)nullptr, (long long*)nullptr, ce_state,
                       reinterpret_cast<long long*>(counts), (float*)nullptr, (MetScratch*)nullptr);
    HIPCHK(hipGetLastError());
    return MSHGNN_OK;
}

extern "C" int mshgnn_metrics_classification_step(const float* logits, const int32_t* y, int64_t batch, double* batch_ce, int64_t* batch_counts,
                                                  double* epoch_ce, int64_t* epoch_counts, float* grad_out, void* scratch, void* stream) {
    if (!logits || !y || batch < 1 || !scratch || (!batch_ce != !batch_counts) || (!epoch_ce != !epoch_counts) || (!batch_ce && !epoch_ce))
        return set_err(MSHGNN_EINVAL, "bad argument to mshgnn_metrics_classification_step");
    hipLaunchKernelGGL(k_metrics_cls<MET_THREADS>, dim3(met_blocks(batch, 1)), dim3(MET_THREADS), 0, (hipStream_t)stream, logits, y, batch, batch_ce,
                       reinterpret_cast<long long*>(batch_counts), epoch_ce, reinterpret_cast<long long*>(epoch_counts), grad_out,
                       reinterpret_cast<MetScratch*>(scratch));
    HIPCHK(hipGetLastError());
    return MSHGNN_OK;
}

extern "C" int mshgnn_metrics_com_step(const float* y_pred, const float* y, int64_t batch, int n_bases, const double* y_mean, const double* y_std,
                                       double* batch_state, double* epoch_state, void* scratch, void* stream) {
    if (!y_pred || !y || !y_mean || !y_std || batch < 1 || n_bases < 1 || !scratch || (!batch_state && !epoch_state))
        return set_err(MSHGNN_EINVAL, "bad argument to mshgnn_metrics_com_step");
    ComStats st;
    for (int k = 0; k < 6; ++k) { st.mean[k] = y_mean[k]; st.std[k] = y_std[k]; }
    hipLaunchKernelGGL(k_metrics_com, dim3(met_blocks(batch, 1)), dim3(MET_THREADS), 0, (hipStream_t)stream, y_pred, y, batch, n_bases, st, batch_state,
                       epoch_state, reinterpret_cast<MetScratchCom*>(scratch));
    HIPCHK(hipGetLastError());
    return MSHGNN_OK;
}

extern "C" int mshgnn_grf_body_to_world(const float* quat, const float* grf_body, float* grf_world, int64_t batch, void* stream) {
    if (!quat || !grf_body || !grf_world || batch < 1) return set_err(MSHGNN_EINVAL, "bad argument to mshgnn_grf_body_to_world");
    hipLaunchKernelGGL(k_grf_to_world, dim3((unsigned)((batch + 255) / 256)), dim3(256), 0, (hipStream_t)stream, quat, grf_body, grf_world, batch);
    HIPCHK(hipGetLastError());
    return MSHGNN_OK;
}

// ------------------------------------------------------------------------------------------------------
// On-device window assembly: the raw time series of a sequence stay in HBM and a batch of
// windows [start, start + T) is gathered straight into the engine's input layout [B][n_t][pitch] at the plan dtype --
// what the reference does per window in Python (quadSDKDataset_Morph.py:304-369: axis-major flatten('F') of each
// variable, joint re-ordering, base tiling, all-ones feet) followed by PyG's collate.
// One wave per RUN = T consecutive features of one node row: feature f0 + t = src[start + t][col] (optionally
// standardised over the window like flexibleDataset.py:390-396), or the constant 1.
// ------------------------------------------------------------------------------------------------------
constexpr int WIN_MAX_SRC = 12;
constexpr int WIN_ROW_RUNS = 8;          // runs per node row handled with all loads in flight
struct WindowArgs {
    const float* src[WIN_MAX_SRC]; int64_t src_cstride[WIN_MAX_SRC];       // series are COLUMN-major: element (row, col) at col * cstride + row
    void* x[MSHGNN_MAX_TYPES]; int64_t x_pitch[MSHGNN_MAX_TYPES]; int nodes[MSHGNN_MAX_TYPES];
    const int* runs; int n_runs;            // per run: type, node, first feature, source (-1: ones) << 8 | column, length; sorted by (type, node)
    const int* rows; int n_rows;            // per node row: first run, end run
    const int64_t* starts; int64_t B; int T, normalize;
    const int* label_cols; int n_label, label_src, label_rotate, quat_src;
    float* y; float* quat;
};

// one WORKGROUP per window, its 4 waves take the node rows round-robin.  Lane r resolves run r ONCE (source pointer, destination
// offset, length) and the waves fetch those with v_readlane, so there is no dependent descriptor load per run; every run
// of a row is a contiguous stretch of a column-major series, read coalesced with all of the row's loads in flight.
template <typename T, int NSET> __global__ __launch_bounds__(256) void k_assemble_windows(WindowArgs a) {
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t b = blockIdx.x;
    const int64_t start = a.starts[b];
    // lane r & 63 of register set r >> 6 <- run r (n_runs <= 64 NSET, checked by the host)
    int v_lo[NSET], v_hi[NSET], v_doff[NSET], v_len[NSET], v_t[NSET];
#pragma unroll
    for (int set = 0; set < NSET; ++set) {
        const int* run = a.runs + (size_t)min(lane + 64 * set, a.n_runs - 1) * 5;
        const int t = run[0], node = run[1], f0 = run[2], sc = run[3];
        const float* sp = nullptr;
#pragma unroll
        for (int k = 0; k < WIN_MAX_SRC; ++k) if (sc >= 0 && (sc >> 8) == k) sp = a.src[k] + (size_t)(sc & 0xff) * a.src_cstride[k] + start;
        int nodes = 0; int64_t pitch = 0;
#pragma unroll
        for (int k = 0; k < MSHGNN_MAX_TYPES; ++k) if (t == k) { nodes = a.nodes[k]; pitch = a.x_pitch[k]; }
        v_lo[set] = (int)((uintptr_t)sp & 0xffffffffu); v_hi[set] = (int)((uintptr_t)sp >> 32);
        v_doff[set] = (int)(((size_t)b * nodes + node) * pitch + f0 - (size_t)b * nodes * pitch);     // offset inside the window's block of this type
        v_len[set] = run[4]; v_t[set] = t;
    }
    auto rl = [&](const int (&v)[NSET], int r) {
        if constexpr (NSET == 1) return __builtin_amdgcn_readlane(v[0], r);
        else return r < 64 ? __builtin_amdgcn_readlane(v[0], r & 63) : __builtin_amdgcn_readlane(v[NSET - 1], r & 63);
    };
    // lane handles the element pairs (2 lane + 128 j, +1), j = 0, 1: one packed store per pair (runs start at even features
    // and node rows are 16-byte aligned, so pairs are 4-byte (bf16) / 8-byte (fp32) aligned); lengths up to 256
    for (int row = wv; row < a.n_rows; row += 4) {
        const int r_begin = a.rows[2 * row], r_end = a.rows[2 * row + 1];
        for (int rb = r_begin; rb < r_end; rb += WIN_ROW_RUNS) {
            float v[WIN_ROW_RUNS][4];
#pragma unroll
            for (int i = 0; i < WIN_ROW_RUNS; ++i) {
                const int r = min(rb + i, r_end - 1);
                const float* sp = reinterpret_cast<const float*>((uintptr_t)(unsigned)rl(v_lo, r) | ((uintptr_t)(unsigned)rl(v_hi, r) << 32));
                const int len = rl(v_len, r);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int k = 2 * lane + 128 * (q >> 1) + (q & 1);
                    v[i][q] = 1.0f;
                    if (rb + i < r_end && sp != nullptr && k < len) v[i][q] = sp[k];
                }
            }
#pragma unroll
            for (int i = 0; i < WIN_ROW_RUNS; ++i) {
                if (rb + i >= r_end) break;
                const int r = rb + i;
                const int t = rl(v_t, r), len = rl(v_len, r), doff = rl(v_doff, r);
                const bool has_src = (rl(v_lo, r) | rl(v_hi, r)) != 0;
                T* dst = reinterpret_cast<T*>(a.x[t]) + (size_t)b * a.nodes[t] * a.x_pitch[t] + doff;
                if (a.normalize && has_src) {
                    // (x - mean) / std with the unbiased estimator, NaN -> 0 (flexibleDataset.py:390-396); fp64, two passes over registers
                    double s1 = 0.0;
#pragma unroll
                    for (int q = 0; q < 4; ++q) if (2 * lane + 128 * (q >> 1) + (q & 1) < len) s1 += (double)v[i][q];
                    const double mean = wave_sum(s1) / (double)len;
                    double qs = 0.0;
#pragma unroll
                    for (int q = 0; q < 4; ++q) if (2 * lane + 128 * (q >> 1) + (q & 1) < len) { const double dlt = (double)v[i][q] - mean; qs += dlt * dlt; }
                    const double sd = sqrt(wave_sum(qs) / (double)(len - 1));
#pragma unroll
                    for (int q = 0; q < 4; ++q) { const double z = ((double)v[i][q] - mean) / sd; v[i][q] = z == z ? (float)z : 0.0f; }
                }
                const bool even = ((doff | len) & 1) == 0;
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int k = 2 * lane + 128 * j;
                    if (even && k < len) {
                        if constexpr (sizeof(T) == 2) {
                            union { unsigned u; __bf16 e[2]; } pk; pk.e[0] = (__bf16)v[i][2 * j]; pk.e[1] = (__bf16)v[i][2 * j + 1];
                            *reinterpret_cast<unsigned*>(dst + k) = pk.u;
                        } else *reinterpret_cast<f32x2*>(dst + k) = f32x2{v[i][2 * j], v[i][2 * j + 1]};
                    } else {
                        if (k < len) dst[k] = from_f32<T>(v[i][2 * j]);
                        if (k + 1 < len) dst[k + 1] = from_f32<T>(v[i][2 * j + 1]);
                    }
                }
            }
        }
    }
}

// Fast path of the same gather (no standardisation, 16-byte aligned output rows whose runs all have one length -- what SequenceStore builds):
// a thread owns one 16-BYTE CHUNK of one node row -- EPC consecutive features, which live in at most two runs of that length -- gathers them
// with EPC scalar loads (consecutive lanes read consecutive elements of a column-major series) and writes ONE 16-byte store; all chunks of a
// window are independent, so every load of the window is in flight at once.  The general kernel above writes 4 bytes per lane and walks a row's
// runs in turn: 0.12 ms for 8192 A1 windows against 0.03-0.04 ms here.  Pad columns inside a row's last chunk are written as zeros.
constexpr int WIN_MAX_ROWS = 64, WIN_MAX_RUNS = 128;
template <typename T> __global__ __launch_bounds__(256) void k_assemble_windows_fast(WindowArgs a) {
    constexpr int EPC = 16 / (int)sizeof(T);
    __shared__ unsigned long long s_src[WIN_MAX_RUNS];                    // source pointer of run r at this window's first step (0: the constant 1)
    __shared__ int s_first[WIN_MAX_ROWS + 1];                             // chunk prefix per node row
    __shared__ int s_run0[WIN_MAX_ROWS], s_len[WIN_MAX_ROWS], s_width[WIN_MAX_ROWS];
    __shared__ unsigned long long s_dst[WIN_MAX_ROWS];                    // destination of the row's first element
    const int tid = threadIdx.x;
    const int64_t b = blockIdx.x, start = a.starts[b];
    if (tid < a.n_runs) {
        const int sc = a.runs[(size_t)tid * 5 + 3];
        const float* sp = nullptr;
#pragma unroll
        for (int k = 0; k < WIN_MAX_SRC; ++k) if (sc >= 0 && (sc >> 8) == k) sp = a.src[k] + (size_t)(sc & 0xff) * a.src_cstride[k] + start;
        s_src[tid] = (unsigned long long)sp;
    }
    if (tid < a.n_rows) {
        const int r0 = a.rows[2 * tid], r1 = a.rows[2 * tid + 1];
        const int* run = a.runs + (size_t)r0 * 5;
        const int t = run[0], node = run[1], len = run[4];
        int nodes = 0; int64_t pitch = 0; char* xb = nullptr;
#pragma unroll
        for (int k = 0; k < MSHGNN_MAX_TYPES; ++k) if (t == k) { nodes = a.nodes[k]; pitch = a.x_pitch[k]; xb = reinterpret_cast<char*>(a.x[k]); }
        s_run0[tid] = r0; s_len[tid] = len; s_width[tid] = (r1 - r0) * len;
        s_dst[tid] = (unsigned long long)(xb + (((size_t)b * nodes + node) * pitch + run[2]) * sizeof(T));
    }
    __syncthreads();
    if (tid == 0) {
        int acc = 0;
        for (int r = 0; r < a.n_rows; ++r) { s_first[r] = acc; acc += (s_width[r] + EPC - 1) / EPC; }
        s_first[a.n_rows] = acc;
    }
    __syncthreads();
    const int total = s_first[a.n_rows];
    int row = 0;
    for (int task = tid; task < total; task += 256) {
        while (task >= s_first[row + 1]) ++row;                            // (tasks of a thread ascend)
        const int j = task - s_first[row], len = s_len[row], width = s_width[row];
        const int k0 = j * EPC;
        int run = s_run0[row] + k0 / len, off = k0 % len;
        float v[EPC];
#pragma unroll
        for (int e = 0; e < EPC; ++e) {
            v[e] = 0.f;
            if (k0 + e < width) {
                const float* sp = reinterpret_cast<const float*>(s_src[run]);
                v[e] = sp ? sp[off] : 1.0f;
            }
            if (++off == len) { off = 0; ++run; }
        }
        T* dst = reinterpret_cast<T*>(s_dst[row]) + k0;
        if constexpr (sizeof(T) == 2) {
            union { u32x4 u; __bf16 e[8]; } pk;
#pragma unroll
            for (int e = 0; e < 8; ++e) pk.e[e] = (__bf16)v[e];
            *reinterpret_cast<u32x4*>(dst) = pk.u;
        } else *reinterpret_cast<f32x4*>(dst) = f32x4{v[0], v[1], v[2], v[3]};
    }
}

// labels of a window = the label row of its LAST time step (quadSDKDataset.py: grfs[-1]); with label_rotate the world-frame
// GRFs are taken into the body frame with the world->body quaternion of that step, R f per foot (the as_matrix() @ grfs_T
// branch of load_data_at_dataset_seq_3d); quat out = that quaternion (data.r_o, quadSDKDataset_Morph.py:365-367)
// (window_labels_one itself lives in mshgnn_device.hpp: the fused-gather encoders run it in extra workgroups of their own launch)
__device__ __forceinline__ LabelArgs label_args_of(const WindowArgs& a, int32_t* labels_int) {
    LabelArgs l{};
    l.lab = a.src[a.label_src]; l.lab_cs = a.src_cstride[a.label_src];
    l.quat_src = a.quat_src >= 0 ? a.src[a.quat_src] : nullptr; l.quat_cs = a.quat_src >= 0 ? a.src_cstride[a.quat_src] : 0;
    l.starts = a.starts; l.B = a.B; l.T = a.T; l.label_cols = a.label_cols; l.n_label = a.n_label; l.label_rotate = a.label_rotate;
    l.y = a.y; l.quat = a.quat; l.labels_int = labels_int;
    return l;
}

__global__ void k_window_labels(WindowArgs a) {
    const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b < a.B) window_labels_one(label_args_of(a, nullptr), b);
}

extern "C" int mshgnn_assemble_windows(const mshgnn_window_desc* d, const float* const* src, const int64_t* src_cstride, const int64_t* src_rows,
                                       const int64_t* starts, int64_t batch, void* const* x_out, const int64_t* x_pitch, float* y_out,
                                       float* quat_out, void* stream) {
    if (!d || !src || !src_cstride || !src_rows || !starts || !x_out || !x_pitch || batch < 1) return set_err(MSHGNN_EINVAL, "null argument to mshgnn_assemble_windows");
    if (d->n_types < 1 || d->n_types > MSHGNN_MAX_TYPES || d->n_src < 1 || d->n_src > WIN_MAX_SRC || d->n_runs < 1 || !d->runs || d->n_rows < 1 || !d->rows)
        return set_err(MSHGNN_EINVAL, "bad window descriptor");
    if (d->history < 1 || (d->normalize && d->history < 2)) return set_err(MSHGNN_EINVAL, "history must be >= 1 (>= 2 when normalising)");
    if (d->dtype != MSHGNN_F32 && d->dtype != MSHGNN_BF16 && d->dtype != MSHGNN_BF16X3) return set_err(MSHGNN_EINVAL, "dtype must be MSHGNN_F32, MSHGNN_BF16 or MSHGNN_BF16X3");
    if (d->n_label > 0 && (!d->label_cols || !y_out || d->label_src < 0 || d->label_src >= d->n_src)) return set_err(MSHGNN_EINVAL, "bad label description");
    if (d->label_rotate && (d->n_label % 3 != 0 || d->quat_src < 0)) return set_err(MSHGNN_EINVAL, "label rotation needs 3-D labels and a quaternion source");
    if (d->quat_src >= d->n_src) return set_err(MSHGNN_EINVAL, "quat_src out of range");
    WindowArgs a{};
    for (int i = 0; i < d->n_src; ++i) {
        if (!src[i] || src_cstride[i] < src_rows[i] || src_rows[i] < d->history) return set_err(MSHGNN_EINVAL, "bad source array");
        a.src[i] = src[i]; a.src_cstride[i] = src_cstride[i];
    }
    for (int t = 0; t < d->n_types; ++t) {
        if (!x_out[t] || d->type_nodes[t] < 1 || x_pitch[t] < d->type_width[t]) return set_err(MSHGNN_EINVAL, "bad output tensor");
        a.x[t] = x_out[t]; a.x_pitch[t] = x_pitch[t]; a.nodes[t] = d->type_nodes[t];
    }
    if (d->history > 256) return set_err(MSHGNN_EUNSUPPORTED, "history longer than 256 steps is not supported by this build");
    a.runs = d->runs; a.n_runs = d->n_runs; a.rows = d->rows; a.n_rows = d->n_rows; a.starts = starts; a.B = batch; a.T = d->history; a.normalize = d->normalize;
    a.label_cols = d->label_cols; a.n_label = d->n_label; a.label_src = d->label_src; a.label_rotate = d->label_rotate; a.quat_src = d->quat_src;
    a.y = y_out; a.quat = quat_out;
    hipStream_t st = (hipStream_t)stream;
    if (d->n_runs > 128) return set_err(MSHGNN_EUNSUPPORTED, "more than 128 feature runs per window are not supported by this build");
    // fast path: no standardisation, every row's runs of one length starting at the row's first feature 0, 16-byte aligned rows whose pitch covers whole chunks
    const bool f32 = d->dtype == MSHGNN_F32 || d->dtype == MSHGNN_BF16X3;      // the split plan takes fp32 inputs
    bool fast = !d->normalize && d->n_rows <= WIN_MAX_ROWS && d->n_runs <= WIN_MAX_RUNS && d->fast_layout != 0;
    for (int t = 0; t < d->n_types && fast; ++t) {
        const int epc = f32 ? 4 : 8;
        if (((uintptr_t)x_out[t] & 15) || x_pitch[t] % epc || x_pitch[t] < (d->type_width[t] + epc - 1) / epc * epc) fast = false;
    }
    if (fast) {
        if (f32) hipLaunchKernelGGL(k_assemble_windows_fast<float>, dim3((unsigned)batch), dim3(256), 0, st, a);
        else hipLaunchKernelGGL(k_assemble_windows_fast<__bf16>, dim3((unsigned)batch), dim3(256), 0, st, a);
    } else if (f32) {
        if (d->n_runs <= 64) hipLaunchKernelGGL((k_assemble_windows<float, 1>), dim3((unsigned)batch), dim3(256), 0, st, a);
        else hipLaunchKernelGGL((k_assemble_windows<float, 2>), dim3((unsigned)batch), dim3(256), 0, st, a);
    } else {
        if (d->n_runs <= 64) hipLaunchKernelGGL((k_assemble_windows<__bf16, 1>), dim3((unsigned)batch), dim3(256), 0, st, a);
        else hipLaunchKernelGGL((k_assemble_windows<__bf16, 2>), dim3((unsigned)batch), dim3(256), 0, st, a);
    }
    if (d->n_label > 0 || (quat_out && d->quat_src >= 0))
        hipLaunchKernelGGL(k_window_labels, dim3((unsigned)((batch + 255) / 256)), dim3(256), 0, st, a);
    HIPCHK(hipGetLastError());
    return MSHGNN_OK;
}

// ------------------------------------------------------------------------------------------------------
// mshgnn_step_mse_series: one training step straight from a sequence's resident raw series -- the window gather of mshgnn_assemble_windows
// fused into the encoder (k_enc_fwd<.., SERIES>), which also writes the materialised windows for the weight-gradient kernel; labels by
// k_window_labels.  bf16 plan with the fused stack kernels; everything after the encoder is mshgnn_step_mse.
// ------------------------------------------------------------------------------------------------------
// the one thing mshgnn_step_*_series needs before its encoder launch: the base pointer of every run's series column (one workgroup).  The windows'
// labels (and the int32 contact flags of the classification step) are computed by extra workgroups of the encoder launch itself (SeriesSrc.lab):
// a launch of their own cost 13.6 us of dependent round trips in front of the encoder, there they run under its tail.
__global__ void k_series_run_ptrs(const int* runs, int n_runs, WindowArgs wa, int elem_bytes, unsigned long long* run_ptr) {
    const int r = threadIdx.x;
    if (r >= n_runs) return;
    const int sc = runs[(size_t)r * 5 + 3];
    unsigned long long p = 0ull;
#pragma unroll
    for (int k = 0; k < WIN_MAX_SRC; ++k)
        if (sc >= 0 && (sc >> 8) == k) p = (unsigned long long)(reinterpret_cast<const char*>(wa.src[k]) + (size_t)(sc & 0xff) * wa.src_cstride[k] * elem_bytes);
    run_ptr[r] = p;
}

static int step_series(const mshgnn_plan* p, const mshgnn_window_desc* d, const float* const* src, const void* const* src_bf16,
                       const int64_t* src_cstride, const int64_t* src_rows, const int64_t* starts, int64_t batch,
                       void* const* x_out, const int64_t* x_pitch, float* y_out, float* quat_out, void* run_ptrs,
                       const float* params, float* out, float* loss_out, float* grad_params, void* workspace, void* stream, int32_t* labels_out) {
    const bool ce = labels_out != nullptr;      // classification wrappers: cross entropy over the per-foot logit pairs, labels = the window labels != 0
    if (!p || !d || !src || !src_cstride || !src_rows || !starts || !y_out || !run_ptrs || !params || !out || !loss_out ||
        !grad_params || !workspace || (x_out && !x_pitch)) return set_err(MSHGNN_EINVAL, "null argument to mshgnn_step_mse_series / mshgnn_step_ce_series");
    if (batch < 1 || batch > (1 << 24)) return set_err(MSHGNN_EINVAL, "batch must be in [1, 2^24]");
    // bf16 plan (fused stack kernels): bf16 copies of the series, optional materialisation.  Split plan (MSHGNN_BF16X3): the fp32 series
    // themselves, windows always materialised (fp32) for its weight-gradient kernel.
    const bool x3 = !p->gen && p->hp.d.dtype == MSHGNN_BF16X3;
    if (p->gen || (!x3 && (p->hp.d.dtype != MSHGNN_BF16 || !p->use_fused)))
        return set_err(MSHGNN_EUNSUPPORTED, "mshgnn_step_mse_series runs on the bf16 plan with the fused stack kernels or on the split plan; use mshgnn_assemble_windows + mshgnn_step_mse");
    if (!x3 && !src_bf16) return set_err(MSHGNN_EINVAL, "null argument to mshgnn_step_mse_series / mshgnn_step_ce_series");
    if (x3 && !x_out) return set_err(MSHGNN_EUNSUPPORTED, "the split plan's weight-gradient kernel reads materialised windows: x_out must be given");
    const mshgnn_desc& md = p->hp.d;
    const bool dtype_ok = x3 ? (d->dtype == MSHGNN_F32 || d->dtype == MSHGNN_BF16X3) : d->dtype == MSHGNN_BF16;
    if (d->n_types != md.n_types || !dtype_ok || d->normalize || !d->fast_layout || d->n_src < 1 || d->n_src > WIN_MAX_SRC || d->n_runs < 1 ||
        d->n_runs > WIN_MAX_RUNS || !d->runs || !d->rows || d->history < 1 || d->n_label < 1 || !d->label_cols || d->label_src < 0 || d->label_src >= d->n_src)
        return set_err(MSHGNN_EINVAL, "mshgnn_step_mse_series: the window descriptor must be a fast_layout, unstandardised recipe with labels at the plan's input dtype (bf16; split plan: fp32)");
    const int epc = x3 ? 4 : 8;      // elements per 16 bytes of the window buffers
    int n_rows = 0;
    for (int t = 0; t < d->n_types; ++t) {
        if (d->type_nodes[t] != md.type_nodes[t] || d->type_width[t] != md.type_width[t]) return set_err(MSHGNN_EINVAL, "window recipe and plan disagree on a node type");
        if (x_out && (!x_out[t] || ((uintptr_t)x_out[t] & 15) || x_pitch[t] % epc || x_pitch[t] < (d->type_width[t] + epc - 1) / epc * epc)) return set_err(MSHGNN_EINVAL, "bad window buffer");
        n_rows += d->type_nodes[t];
    }
    if (n_rows != d->n_rows) return set_err(MSHGNN_EINVAL, "window recipe: one node row per node expected");
    if (d->n_label != md.type_nodes[md.out_type] * (ce ? 1 : md.out_channels)) return set_err(MSHGNN_EINVAL, "window recipe: label count differs from the model's outputs");
    if (ce && md.out_channels != 2) return set_err(MSHGNN_EINVAL, "mshgnn_step_ce_series: the classification wrappers have two logits per foot");
    WindowArgs wa{};
    for (int i = 0; i < d->n_src; ++i) {
        // (the bf16 copies need 8 elements of slack behind every column: a chunk's 16-byte load may run past the window's last step)
        if (!src[i] || (!x3 && !src_bf16[i]) || src_rows[i] < d->history || src_cstride[i] < src_rows[i] + 8 || src_rows[i] >= (1ll << 31)) return set_err(MSHGNN_EINVAL, "bad source array (the gather needs cstride >= rows + 8)");
        wa.src[i] = x3 ? src[i] : reinterpret_cast<const float*>(src_bf16[i]); wa.src_cstride[i] = src_cstride[i];
    }
    hipStream_t st = (hipStream_t)stream;
    // labels (fp32 series)
    WindowArgs la{};
    for (int i = 0; i < d->n_src; ++i) { la.src[i] = src[i]; la.src_cstride[i] = src_cstride[i]; }
    la.starts = starts; la.B = batch; la.T = d->history; la.label_cols = d->label_cols; la.n_label = d->n_label; la.label_src = d->label_src;
    la.label_rotate = d->label_rotate; la.quat_src = d->quat_src; la.y = y_out; la.quat = quat_out;
    if (d->label_rotate && (d->n_label % 3 != 0 || d->quat_src < 0)) return set_err(MSHGNN_EINVAL, "label rotation needs 3-D labels and a quaternion source");
    static_assert(WIN_MAX_RUNS <= 256, "k_series_run_ptrs resolves the runs in one 256-thread workgroup");
    if (ce && d->label_rotate) return set_err(MSHGNN_EINVAL, "mshgnn_step_ce_series: contact labels are not rotated");
    if (!d->run_ptrs_ready)      // (the caller vouches for the scratch's contents otherwise: same descriptor, same source arrays as the call that filled it)
        hipLaunchKernelGGL(k_series_run_ptrs, dim3(1), dim3(256), 0, st, d->runs, d->n_runs, wa, x3 ? 4 : 2, reinterpret_cast<unsigned long long*>(run_ptrs));
    SeriesSrc ser{};
    {   // labels: extra workgroups of the encoder launch
        LabelArgs& l = ser.lab;
        l.lab = la.src[la.label_src]; l.lab_cs = la.src_cstride[la.label_src];
        l.quat_src = la.quat_src >= 0 ? la.src[la.quat_src] : nullptr; l.quat_cs = la.quat_src >= 0 ? la.src_cstride[la.quat_src] : 0;
        l.starts = starts; l.B = batch; l.T = d->history; l.label_cols = d->label_cols; l.n_label = d->n_label; l.label_rotate = d->label_rotate;
        l.y = y_out; l.quat = quat_out; l.labels_int = ce ? labels_out : nullptr;
    }
    ser.run_ptr = reinterpret_cast<const unsigned long long*>(run_ptrs); ser.rows = d->rows; ser.starts = starts; ser.T = d->history;
    { int r0 = 0; for (int t = 0; t < d->n_types; ++t) { ser.row0[t] = r0; r0 += d->type_nodes[t]; } }
    if (x3) {      // (the split plan fuses the MSE into its forward kernel's tail; cross entropy: forward, then the fused-loss backward)
        bool x3_stack_done = false;
        int rc = x3_forward(p, x_out, x_pitch, params, out, (char*)workspace, batch, 1, st, ce ? nullptr : y_out, &ser, &x3_stack_done, ce ? labels_out : nullptr);
        if (rc) return rc;
        return x3_backward(p, x_out, x_pitch, params, nullptr, grad_params, (char*)workspace, batch, st, out, ce ? nullptr : y_out, loss_out,
                           ce ? labels_out : nullptr, true, -1, x3_stack_done);
    }
    bool stack_done = false;
    int rc = forward_impl<__bf16>(p, x_out, x_pitch, params, out, (char*)workspace, batch, 1, st, ce ? nullptr : y_out, &ser, ce ? labels_out : nullptr, &stack_done);
    if (rc) return rc;
    // x_out == NULL: no materialised windows at all -- the weight-gradient kernel gathers its raw-input operands from the series as well
    return backward_impl<__bf16>(p, x_out, x_pitch, params, nullptr, grad_params, (char*)workspace, batch, st, out, ce ? nullptr : y_out, loss_out,
                                 ce ? labels_out : nullptr, true, -1, x_out ? nullptr : &ser, stack_done);
}

extern "C" int mshgnn_step_mse_series(const mshgnn_plan* p, const mshgnn_window_desc* d, const float* const* src, const void* const* src_bf16,
                                      const int64_t* src_cstride, const int64_t* src_rows, const int64_t* starts, int64_t batch,
                                      void* const* x_out, const int64_t* x_pitch, float* y_out, float* quat_out, void* run_ptrs,
                                      const float* params, float* out, float* loss_out, float* grad_params, void* workspace, void* stream) {
    return step_series(p, d, src, src_bf16, src_cstride, src_rows, starts, batch, x_out, x_pitch, y_out, quat_out, run_ptrs, params, out, loss_out, grad_params,
                       workspace, stream, nullptr);
}

extern "C" int mshgnn_step_ce_series(const mshgnn_plan* p, const mshgnn_window_desc* d, const float* const* src, const void* const* src_bf16,
                                     const int64_t* src_cstride, const int64_t* src_rows, const int64_t* starts, int64_t batch,
                                     void* const* x_out, const int64_t* x_pitch, float* y_out, int32_t* labels_out, void* run_ptrs,
                                     const float* params, float* out, float* loss_out, float* grad_params, void* workspace, void* stream) {
    if (!labels_out) return set_err(MSHGNN_EINVAL, "null argument to mshgnn_step_ce_series");
    return step_series(p, d, src, src_bf16, src_cstride, src_rows, starts, batch, x_out, x_pitch, y_out, nullptr, run_ptrs, params, out, loss_out, grad_params,
                       workspace, stream, labels_out);
}
#endif      // MSHGNN_SPEC_SHARD == 0
