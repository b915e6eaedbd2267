// Convolution + fused batch-norm / bias / activation epilogue for gfx950.
//
// Replaces, in ONE launch per layer, the reference's GPU sequence
//   fill_ongpu -> per image { im2col_ongpu ; gemm_ongpu } -> normalize_gpu ->
//   scale_bias_gpu -> add_bias_gpu -> activate_array_ongpu
// (src_yolo2/convolutional_kernels.cu:77-131, batchnorm_layer.c:194-197), whose
// CPU semantics are convolutional_layer.c:435-474 + blas.c:115-126.
//
// Kernels (fp32; the fp16-storage variants live in y2_conv_f16.hip):
//
//  * conv_first_kernel -- the 3-channel first layer (K = 27): no LDS, haloed input, 14 MFMAs per 32
//    pixels, optional fused 2x2 maxpool; HBM bound.
//
//  * conv_stem_kernel -- few-channel first layers of any size / stride (7x7/2, 11x11/4, ...): same shape, with the
//    weights and tap offsets in LDS.
//
//  * splitk_reduce_kernel -- second pass of layers whose K loop was cut across workgroups.
//
//  * conv_mfma_kernel -- implicit GEMM on the fp32 matrix cores
//    (v_mfma_f32_32x32x2_f32).  out[pixel][cout] = sum_k patch[pixel][k] *
//    W[cout][k], with the GEMM "M" dimension = B*H*W output pixels (NHWC, so
//    a pixel's channels are contiguous), "N" = filters and
//    k = (kh*size + kw)*Cin + ci.  No im2col buffer exists: each K-step
//    stages a [BM pixels][BK channels] slice of one filter tap straight from
//    the NHWC input (one 128-byte line per pixel, zero for padding taps via
//    the buffer-load range check) and a [BN filters][BK] slice of the packed
//    weights into LDS, double buffered, while the previous slice is consumed
//    by MFMAs.  LDS rows are padded to BK+4 floats so the ds_read_b128 fragment
//    reads (one per 4 MFMA k-steps) are bank-conflict free.  Workgroups are
//    persistent (tiles b, b+G, ...; staging runs one slice ahead across tile
//    boundaries), a following 2x2/2 maxpool is taken in the epilogue (GEMM rows in
//    pool-major order: a window = four registers of one lane), and grids too
//    small for 256 CUs are cut along K (split-K through an fp32 workspace).
//    The MFMA is an exact k-ordered fp32 fma chain, so results differ from the
//    CPU path (separately rounded mul+add, ci-major k order) only by ordinary
//    fp32 summation noise (~1e-6 relative).
//
//    Strides are free (the GEMM rows enumerate the output grid; taps are centred on input (oy*s, ox*s)); sizes 1, 3
//    and 5 are instantiated (one tap-validity mask bit per tap, pad = size/2).
//
//  * conv_direct_kernel -- plain VALU kernel for shapes the MFMA kernels do
//    not take (Cin not a multiple of 16 behind the first layer, even sizes) and for the strict mode:
//    it accumulates in the reference's exact order (ci, kh, kw ascending;
//    product and sum rounded separately, gemm.c:74-88) and is bit-identical to
//    the CPU path.
//
// The epilogue reproduces blas.c:122 / convolutional_layer.c:407-419 /
// activations.h:41 step by step, each step rounded to fp32 as the reference's
// separate passes do: (x - mean) * [1/(sqrt((double)var)+1e-6f)] in double,
// * scale, + bias, leaky as .1*x in double.
//
// This file is compiled with -ffp-contract=off: no mul+add below may fuse.
#include "y2_conv_shared.hpp"
#include <type_traits>

// ---------------------------------------------------------------------------
// MFMA implicit GEMM
// ---------------------------------------------------------------------------
__device__ int g_skh_timeouts = 0;     // hybrid stream-K: flag waits that gave up (a bug or a lost workgroup; results of that launch are wrong)
template <int BM, int BN, int BK, int KS, int WM, int WN, bool PIPE, bool XO = false, int SKMODE = 0>
__global__ __launch_bounds__(WM *WN * 64, 2) void conv_mfma_kernel(ConvK a)
{
    constexpr bool SKM = SKMODE == 1;     // stream-K over ALL tiles, pieces finished by sk_reduce_kernel (grids smaller than the machine)
    constexpr bool SKH = SKMODE == 2;     // hybrid: whole tiles + the last partial round cut along K, finished inside this launch
    constexpr int NT = WM * WN * 64;
    constexpr int LS = BK + 4;            // LDS row stride (floats); (LS/4) odd -> conflict-free b128 reads
    constexpr int CH = BK / 4;            // 16-byte chunks per staged row
    constexpr int RP = NT / CH;           // rows staged per pass
    constexpr int PA = (BM + RP - 1) / RP, PB = (BN + RP - 1) / RP;   // staging passes (last may be partial)
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
    static_assert(TM >= 1 && TN >= 1, "wave tile must hold at least one 32x32 MFMA tile");

    extern __shared__ __attribute__((aligned(16))) float smem[];
    // [2 buffers][BM + BN rows][LS]  (+ [waves][16][ES] epilogue scratch in the 8-wave tiles)
    constexpr int BUF = (BM + BN) * LS;
    // 8-wave tiles (one workgroup per CU, LDS to spare): outputs leave as 16-byte stores through a wave-private LDS
    // transpose.  A lane of the accumulator layout owns ONE filter of 16 pixels -- sixteen 4-byte stores per 32x32 tile,
    // 96 per lane and tile, each with its own 64-bit address; transposed, a lane stores 4 consecutive filters of a pixel.
    constexpr bool VST = true;
    constexpr bool ES_OWN = (WM * WN == 8);   // 8-wave tiles: 18 KB of LDS of their own; 4-wave tiles (two workgroups per
                                              // CU, no LDS to spare): the staging buffer the last K-step has just
                                              // released, and a barrier behind the epilogue before it is written again
    constexpr int ES = 36;                // scratch row stride (floats): 32 filters + 16 B

    const int t = threadIdx.x;
    const int lane = t & 63, wv = t >> 6;
    const int wm = wv / WN, wn = wv % WN;
    const int li = lane & 31, lh = lane >> 5;

    // Persistent workgroups: block b computes tiles b, b+grid, b+2*grid, ... (filter tile fastest).
    // The staging side runs one slice AHEAD of the MFMA side, across tile boundaries: while the last
    // K-step of tile T multiplies, the first slice of tile T+1 is already being fetched, so neither a
    // per-tile prologue latency nor a second launch wave is exposed.
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void *)a.x, 0, a.xbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc((void *)a.w, 0, a.wbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc((void *)a.y, 0, a.ybytes, 0x00020000);

    // staging role of this thread: chunk `sc` of rows `sr + q*RP`
    const int sc = t % CH, sr = t / CH;
    unsigned a_off[PA];      // byte offset of the pixel's channel 0 (+ this thread's chunk)
    unsigned a_msk[PA];      // validity of the KS*KS taps
    unsigned b_off[PB];
    // row metadata of the tile the STAGING side is working on (tiles past the end: everything masked)
    // A work item ("virtual tile") v is K-split ks = v % ksplit of output tile v / ksplit: it covers the
    // slices [ks*nk/ksplit, (ks+1)*nk/ksplit) of the K loop (split-K, for grids too small to fill 256 CUs).
    const int nk = KS * KS * (a.Cin / BK);
    int tap = 0, c0 = 0;      // position of the NEXT slice to load
    int s_k = 0, s_n = 0;     // staging cursor: slices of its work item already fetched / slices that item has
    // SKM (stream-K, grids smaller than the machine: batch 1 .. 8): the ntiles * nk K-steps of ALL output tiles are dealt in
    // equal contiguous shares to the sk_wgs workgroups (a share never exceeds one tile: ntiles <= sk_wgs), so a workgroup has
    // at most two work items = pieces (tile, K range); work item v in {0, 1} is piece v, its raw sums go to workspace slot
    // 2 * wg + v in tile-local layout [BM][BN], and sk_reduce_kernel adds a tile's pieces in ascending K order.  Against
    // the integer split-K above every workgroup gets the same number of K-steps whatever the tile count.
    int n_sk = 0, sk_t0 = 0, sk_kb0 = 0, sk_ke0 = 0, sk_ke1 = 0;
    if constexpr (SKM) {
        const long I = (long)a.sk_tiles * nk;
        const long lo = (long)blockIdx.x * I / a.sk_wgs, hi = (long)(blockIdx.x + 1) * I / a.sk_wgs;
        if (hi > lo) {
            const int t0 = (int)(lo / nk), k0 = (int)(lo - (long)t0 * nk), len = (int)(hi - lo);
            sk_t0 = t0; sk_kb0 = k0; sk_ke0 = k0 + len < nk ? k0 + len : nk; n_sk = 1;
            if (k0 + len > nk) { sk_ke1 = k0 + len - nk; n_sk = 2; }
        }
    }
    // SKH (hybrid stream-K, grids of several rounds: batch 32): the first ndp = ntiles - sk_tiles output tiles are walked whole
    // (b, b + G, ...); the K loops of the last sk_tiles tiles -- the partial last round -- are dealt in equal contiguous
    // shares to ALL G workgroups, again at most two pieces each.  A piece that does not reach the end of its tile's K loop
    // ("producer"; at most one per workgroup) leaves its raw sums in slot wg and raises that slot's flag; the piece that does ("finisher")
    // waits for the flags of the tile's earlier pieces, adds their sums to its accumulators and runs the ordinary epilogue.
    // A workgroup computes its producer piece FIRST, then its finisher piece, then its whole tiles: no producer ever waits,
    // so the hand-off cannot deadlock (workgroups are dispatched in order: a finisher's producers, all lower-numbered,
    // are on the machine before it), and by the time a finisher has computed its own piece its producers, who started at the
    // same moment with a piece no longer than the finisher's share, have published.  Work-item codes in SKH: 0 .. ndp - 1 a
    // whole tile, ntiles + v piece v of this workgroup (v = 0: in tile sk_t0, from sk_kb0; v = 1: the head of tile sk_t0 + 1).
    const int ndp = SKH ? a.ntiles - a.sk_tiles : 0;
    if constexpr (SKH) {
        const long I = (long)a.sk_tiles * nk;
        const long lo = (long)blockIdx.x * I / a.sk_wgs, hi = (long)(blockIdx.x + 1) * I / a.sk_wgs;
        if (hi > lo) {
            const int t0 = (int)(lo / nk), k0 = (int)(lo - (long)t0 * nk), len = (int)(hi - lo);
            sk_t0 = ndp + t0; sk_kb0 = k0; sk_ke0 = k0 + len < nk ? k0 + len : nk; n_sk = 1;
            if (k0 + len > nk) { sk_ke1 = k0 + len - nk; n_sk = 2; }
        }
    }
    const int END = SKM ? 2 : SKH ? a.ntiles + 2 : a.ntiles;       // work-item number that means "past the end"
    auto setup_tile = [&](int vtile) {
    const bool live = vtile < END;
    const bool piece = SKH && vtile >= a.ntiles;                   // (SKH) a stream-K piece of this workgroup
    const int tile = SKM ? sk_t0 + (vtile & 1) : piece ? sk_t0 + (vtile - a.ntiles) : vtile / a.ksplit;
    const int kb = SKM ? (vtile == 0 ? sk_kb0 : 0)
                 : SKH ? (piece && vtile == a.ntiles ? sk_kb0 : 0)
                       : ((vtile - tile * a.ksplit) * nk) / a.ksplit;     // first slice of this work item
    s_n = !live ? 0x40000000                                                                       // past the end: never hop again
          : SKM ? (vtile == 0 ? sk_ke0 - sk_kb0 : sk_ke1)
          : SKH ? (!piece ? nk : vtile == a.ntiles ? sk_ke0 - sk_kb0 : sk_ke1)
                : (((vtile - tile * a.ksplit) + 1) * nk) / a.ksplit - kb;
    c0 = (kb / (KS * KS)) * BK;
    tap = kb % (KS * KS);
    const int p0 = (tile / a.tiles_n) * BM, n0 = (tile % a.tiles_n) * BN;
    // One coordinate decode per thread and tile: this thread's rows are RP apart, so the image coordinates
    // of the following rows come from a carry update (unit grid = pooling windows, row r = 4*window + corner,
    // with the fused pool; pixels without), and the nine tap tests collapse into one product of a column and a
    // row pattern.  The setup runs on the VALU while the matrix pipe waits, so its length matters on short-K layers.
    // (a.out_h x a.out_w is the output grid the GEMM rows enumerate, a.H x a.W the input the taps are read from:
    // output (oy, ox) is centred on input (oy*stride, ox*stride) for the pad = size/2 shapes this kernel takes)
    const int Wu = a.pool ? a.out_w >> 1 : a.out_w, Hu = a.pool ? a.out_h >> 1 : a.out_h;
    const int r0 = p0 + sr;
    const int u0 = a.pool ? r0 >> 2 : r0, tc = r0 & 3;        // RP % 4 == 0: the corner is the same for every q
    int cn = u0 / (Hu * Wu);
    int cy = (u0 - cn * Hu * Wu) / Wu, cx = u0 - cn * Hu * Wu - cy * Wu;
    static_assert(RP % 4 == 0, "rows of one thread must keep their pooling-window corner");
#pragma unroll
    for (int q = 0; q < PA; ++q) {
        const int r = r0 + q * RP;                            // GEMM row
        const int py = (a.pool ? 2 * cy + (tc >> 1) : cy) * a.stride, px = (a.pool ? 2 * cx + (tc & 1) : cx) * a.stride;
        a_off[q] = ((unsigned)((cn * a.H + py) * a.W + px) * (unsigned)a.ldx + (unsigned)sc * 4u) * 4u;
        unsigned m = 0;
        if (live && r < a.npix && (BM % RP == 0 || sr + q * RP < BM)) {
            if (KS == 1) m = 1u;
            else if (KS == 3) {
                // bit kh*3+kw = tap inside the image: (column pattern) x (row pattern spread 3 bits apart), no carries
                const unsigned xm = (px > 0 ? 1u : 0u) | 2u | (px < a.W - 1 ? 4u : 0u);
                const unsigned ym = (py > 0 ? 1u : 0u) | 8u | (py < a.H - 1 ? 64u : 0u);
                m = xm * ym;
            } else {
                // same product for any odd size with KS*KS <= 32 taps (5x5: alexnet.cfg)
                static_assert(KS * KS <= 32, "one mask bit per tap");
                unsigned xm = 0, ym = 0;
#pragma unroll
                for (int k = 0; k < KS; ++k) {
                    if (px + k - KS / 2 >= 0 && px + k - KS / 2 < a.W) xm |= 1u << k;
                    if (py + k - KS / 2 >= 0 && py + k - KS / 2 < a.H) ym |= 1u << (k * KS);
                }
                m = xm * ym;
            }
        }
        a_msk[q] = m;
        cx += a.pool ? RP / 4 : RP;
        while (cx >= Wu) { cx -= Wu; if (++cy >= Hu) { cy = 0; ++cn; } }
    }
#pragma unroll
    for (int q = 0; q < PB; ++q) {
        const unsigned co = (unsigned)(n0 + sr + q * RP);
        // rows past Cout (or past the tile) land beyond wbytes and read as zero
        b_off[q] = (live && co < (unsigned)a.Cout && sr + q * RP < BN) ? (co * (unsigned)a.K + (unsigned)sc * 4u) * 4u : a.wbytes;
    }
    };

    f32x16 acc[TM][TN];
    f32x4 ra[PA], rb[PB];

    // i-th tile of this workgroup (a.ntiles = none): b, b+G, b+2G, ...  (Cutting the tile range into
    // one contiguous chunk per XCD, so that neighbouring pixel tiles share halo rows in one L2, was
    // measured slower on every layer: profiles/r01_notes.md.)
    int wgid = blockIdx.x;
    if (a.dbg & 64) {         // A/B (env Y2_CONV_XCD_REMAP): XCD-contiguous tile numbers (bijective for any grid)
        const int nwg = gridDim.x, xcd = wgid & 7, q = nwg >> 3, r = nwg & 7;
        wgid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (wgid >> 3);
    }
    auto tile_at = [&](int i) -> int {
        if constexpr (XO) {        // (a template parameter: as a run-time switch its scalar state costs the 192x256 kernel 88 B more scratch per lane)
            // work list of XCD x: for each block of pblk pixel tiles, for each of its filter tiles x, x + 8, ..., the block's
            // pixel tiles; the XCD's gridDim.x / 8 workgroups walk it with that stride
            const int xcd = wgid & 7, nf = (a.tiles_n - xcd + 7) >> 3;           // filter tiles of this XCD
            const long idx = (long)(wgid >> 3) + (long)i * (gridDim.x >> 3);
            if (idx >= (long)nf * a.tiles_m) return a.ntiles;
            const int full = a.tiles_m / a.pblk, per = nf * a.pblk;               // whole blocks, tiles per whole block
            int pb, fi, pi;
            if (idx < (long)full * per) {
                pb = (int)(idx / per);
                const int rem = (int)(idx - (long)pb * per);
                fi = rem / a.pblk; pi = rem - fi * a.pblk;
            } else {
                const int last = a.tiles_m - full * a.pblk, rem = (int)(idx - (long)full * per);
                pb = full; fi = rem / last; pi = rem - fi * last;
            }
            return (pb * a.pblk + pi) * a.tiles_n + xcd + 8 * fi;
        }
        if constexpr (SKM) return i < n_sk ? i : END;
        if constexpr (SKH) {
            if (i < n_sk) return a.ntiles + (n_sk == 2 ? 1 - i : 0);          // the producer piece (head of the next tile) first
            const long tl = (long)blockIdx.x + (long)(i - n_sk) * gridDim.x;
            return tl < ndp ? (int)tl : END;
        }
        const long tl = (long)wgid + (long)i * gridDim.x;
        return tl < a.ntiles ? (int)tl : a.ntiles;
    };
    int lti = 0;              // staging side: index of its tile in this workgroup's sequence
    setup_tile(tile_at(0));
    // After the last slice of the last tile the staging side keeps running one step into a
    // "tile" past the end whose rows are all masked: those loads are out-of-range buffer accesses
    // (zeros, no memory traffic) -- cheaper than a branch, which would split the scheduling region.
    auto load_slice = [&]() {
        int delta = 0;        // float offset of the tap relative to the centre pixel
        if (KS > 1) {
            const int kh = tap / KS, kw = tap - kh * KS;
            delta = ((kh - KS / 2) * a.W + (kw - KS / 2)) * a.ldx;
        }
        const unsigned add = (unsigned)((delta + c0) * 4);
#pragma unroll
        for (int q = 0; q < PA; ++q) {
            const bool ok = (a_msk[q] >> tap) & 1u;
            const unsigned off = ok ? a_off[q] + add : a.xbytes;
            ra[q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xr, off, 0, 0));
        }
        const unsigned kadd = (unsigned)((tap * a.Cin + c0) * 4);
#pragma unroll
        for (int q = 0; q < PB; ++q) {
            const unsigned off = (b_off[q] == a.wbytes) ? a.wbytes : b_off[q] + kadd;
            rb[q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(wr, off, 0, 0));
        }
        // K order: channel chunk outermost, the KS*KS taps innermost.  The nine taps of one
        // 32-channel chunk touch the same (neighbouring) 128-byte pixel lines, so eight of the
        // nine A-slice reads hit L1/L2 instead of going back to the Infinity Cache / HBM.
        if (++tap == KS * KS) { tap = 0; c0 += BK; }     // the hop to the next tile is done by the K loop
        ++s_k;
    };
    auto store_slice = [&](int buf) {
        float *As = smem + buf * BUF;
        float *Bs = As + BM * LS;
#pragma unroll
        for (int q = 0; q < PA; ++q)
            if (BM % RP == 0 || sr + q * RP < BM) *(f32x4 *)&As[(sr + q * RP) * LS + sc * 4] = ra[q];
#pragma unroll
        for (int q = 0; q < PB; ++q)
            if (BN % RP == 0 || sr + q * RP < BN) *(f32x4 *)&Bs[(sr + q * RP) * LS + sc * 4] = rb[q];
    };

#ifdef Y2_NO_EARLYB
    constexpr bool EB = false;
#else
    // (not the 64x64 tile: it is what batch-1 grids run, one short work item per workgroup, where the longer prologue of
    // the two-slices-ahead staging costs 1-2 us per layer and buys nothing)
    constexpr bool EB = PIPE && BK >= 32 && !(BM == 64 && BN == 64);
#endif
    load_slice();
    store_slice(0);
    if (EB) {
        // EB keeps the staging side TWO slices ahead when it issues its loads (one register set: the slice is loaded
        // under the last MFMA group of K-step k and written to LDS under the third group of step k+1, three groups of
        // latency cover instead of two -- storing one group earlier stalled on the loads, profiles/r02_notes.md)
        if (s_k == s_n) { setup_tile(tile_at(++lti)); s_k = 0; }
        load_slice();
        // raw barrier: only the LDS stores of slice 0 have to be complete -- __syncthreads() would also wait for the
        // second slice's loads just issued (one full memory latency per launch: 5 % of a batch-1 layer)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    } else {
        __syncthreads();
    }

    // EB ("early barrier", BK = 32 tiles): the slice fetched in this K-step is written to the other LDS buffer under the
    // SECOND-TO-LAST MFMA group and the workgroup barrier follows that group; the last group then multiplies from
    // fragments already in registers while the first fragments of the NEXT K-step are read from the buffer the barrier
    // has just published -- a K-step boundary no longer exposes an LDS read latency with the matrix pipe idle.  Still
    // one barrier per K-step: every read of the current buffer is issued (and, by the barrier's lgkmcnt(0), returned)
    // before the barrier, so the buffer may be overwritten one step later without a second one.
    f32x4 af[2][TM], bf[2][TN];      // operand fragments, two register sets (EB: live across K-steps and tiles)
    if (EB) {
        const float *As0 = smem + (wm * (BM / WM) + li) * LS + lh * 4;
        const float *Bs0 = smem + BM * LS + (wn * (BN / WN) + li) * LS + lh * 4;
#pragma unroll
        for (int i = 0; i < TM; ++i) af[0][i] = *(const f32x4 *)&As0[i * 32 * LS];
#pragma unroll
        for (int j = 0; j < TN; ++j) bf[0][j] = *(const f32x4 *)&Bs0[j * 32 * LS];
    }

    // One K-step = BK/8 groups of 4*TM*TN MFMAs.  Software pipeline (PIPE): the operand fragments
    // of group g+1 are read from LDS while group g multiplies (two register sets), the global
    // loads of the next slice are issued under group 0 and written to the other LDS buffer
    // before the last group, so that when the workgroup reaches the barrier only the barrier is
    // left -- the matrix pipe idles for one LDS read latency per K-step instead of one per group
    // plus the whole staging tail.  sched_barrier pins the phases against the compiler's scheduler.
    constexpr int NG = BK / 8;
    int cur = 0;
#ifdef Y2_F32_STAMPS
    // diagnostic: cycles per wave in [0] tile setup + accumulator clear, [1] K loop, [2] epilogue; [3] K-steps, [4] tiles
    unsigned long long st[7] = {0, 0, 0, 0, 0, 0, 0}, st_bar = 0, st_prev = __builtin_amdgcn_s_memtime();      // [5] producer publish, [6] finisher wait + gather (hybrid stream-K)
#define F32_STAMP(k) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); st[k] += t_ - st_prev; st_prev = t_; } while (0)
#else
#define F32_STAMP(k) do { } while (0)
#endif
    for (int cti = 0;; ++cti) {
    const int vt = tile_at(cti);
    if (vt >= END) break;
    const bool piece = SKH && vt >= a.ntiles;
    const int ct = SKM ? sk_t0 + vt : piece ? sk_t0 + (vt - a.ntiles) : vt / a.ksplit;
    const int ks = SKM ? 2 * (int)blockIdx.x + vt : piece ? (int)blockIdx.x : vt - ct * a.ksplit;       // SKM: piece slot; SKH: a workgroup has at most ONE producer piece, slot = workgroup
    const int kb = SKM ? (vt == 0 ? sk_kb0 : 0) : SKH ? (piece && vt == a.ntiles ? sk_kb0 : 0) : (ks * nk) / a.ksplit;
    const int ke = SKM ? (vt == 0 ? sk_ke0 : sk_ke1) : SKH ? (!piece ? nk : vt == a.ntiles ? sk_ke0 : sk_ke1) : ((ks + 1) * nk) / a.ksplit;
    const int p0 = (ct / a.tiles_n) * BM, n0 = (ct % a.tiles_n) * BN;
    if constexpr (SKH) {
        // Every vector-memory operation retired before a work item starts, as a wait the compiler's waitcnt pass SEES (an asm
        // wait is invisible to it): otherwise a register reload issued in front of the tile loop stays "pending" in the pass's
        // merged state at the K loop's header, and the K-step opens with vmcnt(4) / vmcnt(0) -- the slice loads issued under
        // the previous step's last MFMA group get 1 300 cycles of cover instead of 4 500 (K-step 12.8 k -> 16.2 k cycles).
        // Costs the tail of the previous item's stores once per work item.
        __builtin_amdgcn_s_waitcnt(0x0F70);       // vmcnt(0) expcnt(7) lgkmcnt(15)
    }
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    // a wave in its K loop outranks co-resident waves that are in their (VALU / store) epilogue or tile setup
    __builtin_amdgcn_s_setprio(1);
    F32_STAMP(0);
#ifdef Y2_F32_STAMPS
    st[3] += ke - kb; st[4] += 1;
#endif
    for (int kt = kb; kt < ke; ++kt) {
        if (EB ? (s_k == s_n) : (kt == ke - 1)) {
            // the slice fetched during this K-step is the first one of the block's next work item;
            // switching here, outside the K-step body, keeps that body a single scheduling region
            setup_tile(tile_at(++lti));
            s_k = 0;
        }
        const float *As = smem + cur * BUF + (wm * (BM / WM) + li) * LS + lh * 4;
        const float *Bs = smem + cur * BUF + BM * LS + (wn * (BN / WN) + li) * LS + lh * 4;
        if (!PIPE) {
            load_slice();                            // global loads in flight under the MFMAs
#pragma unroll
            for (int kg = 0; kg < NG; ++kg) {
                f32x4 af[TM], bf[TN];
#pragma unroll
                for (int i = 0; i < TM; ++i) af[i] = *(const f32x4 *)&As[i * 32 * LS + kg * 8];
#pragma unroll
                for (int j = 0; j < TN; ++j) bf[j] = *(const f32x4 *)&Bs[j * 32 * LS + kg * 8];
#pragma unroll
                for (int s = 0; s < 4; ++s)
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][s], bf[j][s], acc[i][j], 0, 0, 0);
            }
            store_slice(cur ^ 1);
        } else {
            if (!EB) {
#pragma unroll
                for (int i = 0; i < TM; ++i) af[0][i] = *(const f32x4 *)&As[i * 32 * LS];
#pragma unroll
                for (int j = 0; j < TN; ++j) bf[0][j] = *(const f32x4 *)&Bs[j * 32 * LS];
            }
            constexpr int SG = EB ? NG - 2 : NG - 1;      // group under which the fetched slice is written to LDS
            constexpr int BG = EB ? NG - 2 : NG - 1;                  // group behind which the workgroup barrier sits
            constexpr int LG = EB ? NG - 1 : 0;                       // group under which the next global loads are issued
#pragma unroll
            for (int kg = 0; kg < NG; ++kg) {
                const int c = kg & 1, n = c ^ 1;
                if (kg + 1 < NG) {
#pragma unroll
                    for (int i = 0; i < TM; ++i) af[n][i] = *(const f32x4 *)&As[i * 32 * LS + (kg + 1) * 8];
#pragma unroll
                    for (int j = 0; j < TN; ++j) bf[n][j] = *(const f32x4 *)&Bs[j * 32 * LS + (kg + 1) * 8];
                } else if (EB) {
                    // group 0 of the next K-step, from the buffer published by the barrier behind group NG-2
                    const float *An = smem + (cur ^ 1) * BUF + (wm * (BM / WM) + li) * LS + lh * 4;
                    const float *Bn = smem + (cur ^ 1) * BUF + BM * LS + (wn * (BN / WN) + li) * LS + lh * 4;
#pragma unroll
                    for (int i = 0; i < TM; ++i) af[n][i] = *(const f32x4 *)&An[i * 32 * LS];
#pragma unroll
                    for (int j = 0; j < TN; ++j) bf[n][j] = *(const f32x4 *)&Bn[j * 32 * LS];
                }
                if (kg == SG) store_slice(cur ^ 1);
                if (kg == LG) load_slice();               // (after the store when both fall into one group: same registers)
#pragma unroll
                for (int s = 0; s < 4; ++s)
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[c][i][s], bf[c][j][s], acc[i][j], 0, 0, 0);
                // issue order inside the group: one MFMA first (the pipe is busy from here on), then the
                // fragment reads of the next group, then the staging work one piece per MFMA
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                if (kg + 1 < NG || EB) __builtin_amdgcn_sched_group_barrier(0x100, TM + TN, 0);
                if (kg == LG) {
#pragma unroll
                    for (int q = 0; q < PA + PB; ++q) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x002, 6, 0);
                        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                    }
                }
                if (kg == SG) {
#pragma unroll
                    for (int q = 0; q < PA + PB; ++q) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
#ifdef Y2_F32_STAMPS
                if (EB && kg == BG) { const unsigned long long b0 = __builtin_amdgcn_s_memtime(); __syncthreads(); st_bar += __builtin_amdgcn_s_memtime() - b0; }
#else
                if (EB && kg == BG) {
                    // raw barrier: only this wave's LDS traffic has to be complete (its slice stores and its reads of the
                    // current buffer); __syncthreads() would also wait for the global loads that may already be in flight
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    __builtin_amdgcn_s_barrier();
                    __builtin_amdgcn_sched_barrier(0);
                }
#endif
            }
        }
        if (!EB || !PIPE) __syncthreads();
        cur ^= 1;
    }
    __builtin_amdgcn_s_setprio(0);
    F32_STAMP(1);

    // epilogue: lane holds column (cout) li of each 32x32 tile and 16 rows (pixels)
    // (`bn` and `act` are uniform, but tested per output value they are real branches -- 1439 s_cbranch in the 192x256
    // instantiation; the common batch-norm + leaky case is compiled with both as constants)
    auto epilogue_pass = [&](auto MODEC, auto VSTC) {
        constexpr int MODE = decltype(MODEC)::value;
        constexpr bool VS = decltype(VSTC)::value;            // 16-byte stores through the LDS transpose        // 0 run-time bn / act, 1 batch-norm + leaky, 2 batch-norm + linear
        const bool BN_ = MODE ? true : (bool)a.bn;
        const int ACT_ = MODE == 1 ? (int)Y2H_ACT_LEAKY : MODE == 2 ? (int)Y2H_ACT_LINEAR : a.act;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int co = n0 + wn * (BN / WN) + j * 32 + li;
            const bool cok = co < a.Cout;
            float mean = 0.f, scale = 1.f, bias = 0.f;
            double rinv = 1.0;
            if (cok) {
                bias = a.bias[co];
                if (BN_) { mean = a.mean[co]; rinv = a.rinv[co]; scale = a.scale[co]; }
            }
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int prow = p0 + wm * (BM / WM) + i * 32 + 4 * lh;
                if (a.ksplit > 1) {
                    // split-K: raw partial sums to the workspace [split][pixel][filter]; splitk_reduce_kernel
                    // adds the splits in a fixed order (reproducible) and applies the epilogue
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int p = prow + (r & 3) + 8 * (r >> 2);
                        if (cok && p < a.npix) a.ws[((size_t)ks * a.npix + p) * a.Cout + co] = acc[i][j][r];
                    }
                    continue;
                }
                if constexpr (VS) {
                    // LDS operations of one wave execute in order, and the scratch is this wave's own: no barrier.
                    // Stores go through the buffer descriptor of y with 32-bit offsets (rows past the end and filters
                    // past Cout get an out-of-range offset: the store is dropped, no exec masking, no 64-bit address math
                    // -- the scalar form spills 127 registers of hoisted 64-bit addresses in this tile shape).
                    float *es = (ES_OWN ? smem + 2 * BUF : smem + (cur ^ 1) * BUF) + wv * (16 * ES);
                    const int cb = n0 + wn * (BN / WN) + j * 32;          // first filter of this 32-wide tile
                    const int pb = p0 + wm * (BM / WM) + i * 32;          // first GEMM row of this tile
                    const int rrow = lane >> 3, rch = (lane & 7) * 4;     // read side: row 0..7 (+8), filters rch..rch+3
                    const bool fok = cb + rch < a.Cout;
                    if (a.pool) {
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            const float m = epilogue_f32(pool_pick(acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3],
                                                                 !BN_ || scale >= 0.f), BN_, mean, rinv, scale, bias, ACT_);
                            es[(2 * g + lh) * ES + li] = m;               // pooled row (pb + 8g + 4lh) / 4 - pb / 4
                        }
                        const u32x4 v = *(const u32x4 *)&es[rrow * ES + rch];
                        const int prow_ = (pb >> 2) + rrow;
                        const unsigned off = (fok && 4 * prow_ < a.npix) ? ((unsigned)prow_ * (unsigned)a.ldy + (unsigned)(cb + rch)) * 4u : 0xffffffffu;
                        __builtin_amdgcn_raw_buffer_store_b128(v, yr, off, 0, 0);
                    } else {
                        const unsigned base = ((unsigned)(pb + rrow) * (unsigned)a.ldy + (unsigned)(cb + rch)) * 4u;
#pragma unroll
                        for (int h2 = 0; h2 < 2; ++h2) {                  // rows 0..15, then 16..31 of the tile
#pragma unroll
                            for (int r = 0; r < 8; ++r)
                                es[((r & 3) + 8 * (r >> 2) + 4 * lh) * ES + li] =
                                    epilogue_f32(acc[i][j][8 * h2 + r], BN_, mean, rinv, scale, bias, ACT_);
#pragma unroll
                            for (int u = 0; u < 2; ++u) {
                                const u32x4 v = *(const u32x4 *)&es[(rrow + 8 * u) * ES + rch];
                                const int p = pb + 16 * h2 + rrow + 8 * u;
                                const unsigned off = (fok && p < a.npix) ? base + (unsigned)(16 * h2 + 8 * u) * (unsigned)a.ldy * 4u : 0xffffffffu;
                                __builtin_amdgcn_raw_buffer_store_b128(v, yr, off, 0, 0);
                            }
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);      // one tile at a time: interleaved, the six tiles' temporaries spill
                } else {
                if (a.pool) {
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int r0 = prow + 8 * g;              // first of the window's four rows
                        const float m = epilogue_f32(pool_pick(acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3],
                                                                 !BN_ || scale >= 0.f), BN_, mean, rinv, scale, bias, ACT_);
                        // buffer store, 32-bit offset, out-of-range = dropped (no exec masking, no 64-bit address per store)
                        const unsigned off = (cok && r0 < a.npix) ? ((unsigned)(r0 >> 2) * (unsigned)a.ldy + (unsigned)co) * 4u : 0xffffffffu;
                        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, m), yr, off, 0, 0);
                    }
                } else {
                    const unsigned base = ((unsigned)prow * (unsigned)a.ldy + (unsigned)co) * 4u;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int dr = (r & 3) + 8 * (r >> 2);
                        const float v = epilogue_f32(acc[i][j][r], BN_, mean, rinv, scale, bias, ACT_);
                        const unsigned off = (cok && prow + dr < a.npix) ? base + (unsigned)dr * (unsigned)a.ldy * 4u : 0xffffffffu;
                        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), yr, off, 0, 0);
                    }
                }
                }
            }
        }
    };
    // split-K: the raw partial sums of this K range go to their workspace slab [pixel][filter] as 16-byte stores through the
    // same wave-private LDS transpose (the scalar form below -- one 4-byte store with a 64-bit address per value, 96 per lane
    // of a 192x256 tile -- cost a batch-1 work item of 8-14 K-steps 6-12 k cycles: profiles/r02_notes.md)
    auto partial_pass = [&]() {
        // slab of this work item: split-K [pixel][filter] of K range ks; stream-K: slot ks, tile-local [BM][BN]
        const __amdgpu_buffer_rsrc_t wsr = SKM
            ? __builtin_amdgcn_make_buffer_rsrc((void *)(a.ws + (size_t)ks * (BM * BN)), 0, (unsigned)(BM * BN * 4), 0x00020000)
            : __builtin_amdgcn_make_buffer_rsrc((void *)(a.ws + (size_t)ks * a.npix * a.Cout), 0, (unsigned)((size_t)a.npix * a.Cout * 4), 0x00020000);
        const unsigned ldw = SKM ? (unsigned)BN : (unsigned)a.Cout;
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                float *es = (ES_OWN ? smem + 2 * BUF : smem + (cur ^ 1) * BUF) + wv * (16 * ES);
                const int cb = (SKM ? 0 : n0) + wn * (BN / WN) + j * 32, pb = (SKM ? 0 : p0) + wm * (BM / WM) + i * 32;
                const int rrow = lane >> 3, rch = (lane & 7) * 4;
                const bool fok = SKM || cb + rch < a.Cout;
                const unsigned base = ((unsigned)(pb + rrow) * ldw + (unsigned)(cb + rch)) * 4u;
#pragma unroll
                for (int h2 = 0; h2 < 2; ++h2) {
#pragma unroll
                    for (int r = 0; r < 8; ++r) es[((r & 3) + 8 * (r >> 2) + 4 * lh) * ES + li] = acc[i][j][8 * h2 + r];
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        const u32x4 v = *(const u32x4 *)&es[(rrow + 8 * u) * ES + rch];
                        const int p = pb + 16 * h2 + rrow + 8 * u;
                        const unsigned off = (fok && (SKM || p < a.npix)) ? base + (unsigned)(16 * h2 + 8 * u) * ldw * 4u : 0xffffffffu;
                        __builtin_amdgcn_raw_buffer_store_b128(v, wsr, off, 0, 0);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
    };
    if constexpr (SKH) {
        constexpr int NE = TM * TN * 4;                   // 16-byte groups of accumulators per lane
        if (piece && ke < nk) {
            // producer: the raw sums leave as the lanes hold them, NE coalesced 16-byte write-through (sc1) stores per lane into
            // [slot][e][thread]; every wave drains its stores, then ONE lane raises the flag (MI355X hand-off rules: sc1 payload,
            // vmcnt(0) in asm -- the compiler may drop a waitcnt it thinks redundant --, sc1 flag)
            const __amdgpu_buffer_rsrc_t pr = __builtin_amdgcn_make_buffer_rsrc((void *)(a.ws + (size_t)ks * (BM * BN)), 0, (unsigned)(BM * BN * 4), 0x00020000);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const f32x4 v = {acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]};
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), pr, (unsigned)((((i * TN + j) * 4 + q) * NT + t) * 16), 0, 16);
                    }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if (t == 0) __hip_atomic_store(a.sk_flags + ks, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            F32_STAMP(5);
            continue;
        }
        if (piece && kb > 0) {
            // finisher: the earlier pieces of this tile belong to the workgroups below this one whose shares reach into it
            const long I = (long)a.sk_tiles * nk, G = a.sk_wgs, tb = (long)(ct - ndp) * nk;
            long w0 = tb * G / I;
            while (w0 > 0 && w0 * I / G > tb) --w0;
            while (w0 + 1 < G && (w0 + 1) * I / G <= tb) ++w0;
            if (t == 0) {
                for (long w = w0; w < (long)blockIdx.x; ++w) {
                    const long lo = w * I / G, hi = (w + 1) * I / G;
                    if (hi <= lo) continue;
                    int *fl = a.sk_flags + w;
                    int spins = 0;
                    while (__hip_atomic_load(fl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 1) {
                        if (++spins > (1 << 22)) { atomicAdd(&g_skh_timeouts, 1); break; }        // seconds: counted (y2h_f32_stream_k_timeouts), never a hang
                        __builtin_amdgcn_s_sleep(4);
                    }
                }
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // (the polling lane's last load has returned; see above)
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");                          // no payload load may be scheduled in front of the barrier
            for (long w = w0; w < (long)blockIdx.x; ++w) {
                const long lo = w * I / G, hi = (w + 1) * I / G;
                if (hi <= lo) continue;
                const size_t slot = (size_t)w;
                const __amdgpu_buffer_rsrc_t pr = __builtin_amdgcn_make_buffer_rsrc((void *)(a.ws + slot * (BM * BN)), 0, (unsigned)(BM * BN * 4), 0x00020000);
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(pr, (unsigned)((((i * TN + j) * 4 + q) * NT + t) * 16), 0, 16));
#pragma unroll
                            for (int e = 0; e < 4; ++e) acc[i][j][4 * q + e] += v[e];
                        }
            }
            F32_STAMP(6);
        }
    }
    if (SKM || (a.ksplit > 1 && (a.Cout & 3) == 0)) {
        partial_pass();
        if (!ES_OWN) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
    } else if (VST && a.vec_store && a.bn && a.act == Y2H_ACT_LEAKY) {        // (every conv of the target cfgs but the last)
        epilogue_pass(std::integral_constant<int, 1>{}, std::true_type{});
        if (!ES_OWN) {            // the scratch lies in the buffer the next K-step stores its slice into
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
    } else {
        if (a.bn && a.act == Y2H_ACT_LEAKY) epilogue_pass(std::integral_constant<int, 1>{}, std::false_type{});
        else if (a.bn && a.act == Y2H_ACT_LINEAR) epilogue_pass(std::integral_constant<int, 2>{}, std::false_type{});     // resnet's 1x1 expansions
        else epilogue_pass(std::integral_constant<int, 0>{}, std::false_type{});
    }
    F32_STAMP(2);
    }   // tile loop
#ifdef Y2_F32_STAMPS
    if (lane == 0 && a.stamps)
        for (int k = 0; k < 7; ++k) a.stamps[((size_t)blockIdx.x * (NT / 64) + wv) * 7 + k] = (k == 0) ? st_bar : st[k];      // [0]: barrier wait (setup dropped)
#endif
}

// second pass of a split-K convolution: y[p][co] = epilogue(sum_s ws[s][p][co]), s ascending.  With the fused 2x2
// maxpool the GEMM rows are in pool-major order (row 4q + t = corner t of window q): the pooled value is the max over
// the four epilogue results, exactly what the unsplit kernel's epilogue computes.
__global__ __launch_bounds__(256) void splitk_reduce_kernel(ConvK a)
{
    const long rows = a.pool ? (long)a.npix >> 2 : (long)a.npix;
    const long total = rows * a.Cout, slab = (long)a.npix * a.Cout;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const int co = (int)(idx % a.Cout);
        const long p = idx / a.Cout;
        float mean = 0.f, scale = 1.f;
        double rinv = 1.0;
        if (a.bn) { mean = a.mean[co]; rinv = a.rinv[co]; scale = a.scale[co]; }
        const float bias = a.bias[co];
        if (a.pool) {
            float sums[4];
            for (int t = 0; t < 4; ++t) {
                const long off = (4 * p + t) * a.Cout + co;
                float sum = a.ws[off];
                for (int s = 1; s < a.ksplit; ++s) sum += a.ws[(size_t)s * slab + off];
                sums[t] = sum;
            }
            a.y[(size_t)p * a.ldy + co] = epilogue_f32(pool_pick(sums[0], sums[1], sums[2], sums[3], !a.bn || scale >= 0.f), a.bn, mean,
                                                      rinv, scale, bias, a.act);
            continue;
        }
        float sum = a.ws[idx];
        for (int s = 1; s < a.ksplit; ++s) sum += a.ws[(size_t)s * slab + idx];
        a.y[(size_t)p * a.ldy + co] = epilogue_f32(sum, a.bn, mean, rinv, scale, bias, a.act);
    }
}

// second pass of a stream-K convolution (conv_mfma_kernel<..., SKM>): y[p][co .. co+3] = epilogue(sum of the pieces of the
// output's tile, ascending K = ascending workgroup).  A piece slot holds a whole tile in tile-local layout [bm][bn]; the
// pieces of tile t are those of the workgroups whose share [w * I / G, (w + 1) * I / G) reaches into [t * nk, (t + 1) * nk):
// slot 2 w + (1 if the share began in the previous tile).  Cout and ldy multiples of 4 (the host checks).
__global__ __launch_bounds__(256) void sk_reduce_kernel(ConvK a, int bm, int bn, int nk)
{
    const int c4 = a.Cout >> 2;
    const long rows = a.pool ? (long)a.npix >> 2 : (long)a.npix;
    const long total = rows * c4, I = (long)a.sk_tiles * nk, G = a.sk_wgs;
    const int nt = a.pool ? 4 : 1;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const int q = (int)(idx % c4), co = q * 4;
        const long p = idx / c4;
        const long g0 = a.pool ? 4 * p : p;                     // first GEMM row of this output (a pooling window's four rows share a tile: bm % 4 == 0)
        const int tile = (int)(g0 / bm) * a.tiles_n + co / bn;
        const int r0 = (int)(g0 % bm), c = co % bn;
        const long tb = (long)tile * nk, te = tb + nk;
        long w = tb * G / I;
        while (w > 0 && w * I / G > tb) --w;
        while (w + 1 < G && (w + 1) * I / G <= tb) ++w;
        f32x4 sums[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) sums[t] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (; w < G; ++w) {
            const long lo = w * I / G, hi = (w + 1) * I / G;
            if (lo >= te) break;
            if (hi <= lo) continue;
            const float *slot = a.ws + (size_t)(2 * w + (lo < tb ? 1 : 0)) * ((size_t)bm * bn);
            for (int t = 0; t < nt; ++t) sums[t] = sums[t] + *(const f32x4 *)&slot[(size_t)(r0 + t) * bn + c];
        }
        f32x4 out;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float mean = 0.f, scale = 1.f;
            double rinv = 1.0;
            if (a.bn) { mean = a.mean[co + k]; rinv = a.rinv[co + k]; scale = a.scale[co + k]; }
            const float v = a.pool ? pool_pick(sums[0][k], sums[1][k], sums[2][k], sums[3][k], !a.bn || scale >= 0.f) : sums[0][k];
            out[k] = epilogue_f32(v, a.bn, mean, rinv, scale, a.bias[co + k], a.act);
        }
        *(f32x4 *)&a.y[(size_t)p * a.ldy + co] = out;
    }
}

// ---------------------------------------------------------------------------
// First layer (3 input channels, 3x3, stride 1, pad 1, <= 32*NT filters).
//
// K = 27 is too short for the LDS-staged kernel and the layer is HBM bound
// (0.64 GFLOP against 52 MB per 608x608 image), so it gets its own shape:
// no LDS at all.  The input is kept with a one-pixel zero halo
// ([batch][H+2][W+2][ldx]), so no tap needs a bounds test.  A wave takes
// tiles of 32 consecutive output pixels; lane (i, half) loads A[i][k] for
// k = 2t + half, t = 0..13 (k = 27 is a dummy with a zero weight) with plain
// dword loads -- 32 neighbouring pixels are 384 contiguous bytes per tap row --
// holds the 14 matching filter taps of its filter column in registers for the
// whole kernel, and issues 14 v_mfma_f32_32x32x2_f32 per tile.  The next
// tile's loads are in flight while the current one is multiplied.
// ---------------------------------------------------------------------------
#ifndef Y2_FIRST_MINB
#define Y2_FIRST_MINB 1
#endif
template <int NT>
__global__ __launch_bounds__(256, Y2_FIRST_MINB) void conv_first_kernel(ConvK a)
{
    const int lane = threadIdx.x & 63, li = lane & 31, lh = lane >> 5;
    const long wave = ((long)blockIdx.x * 256 + threadIdx.x) >> 6;
    const long nwaves = (long)gridDim.x * 4;
    const long ntiles = ((long)a.npix + 31) / 32;
    const int W2 = a.W + 2, H2 = a.H + 2;
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void *)a.x, 0, a.xbytes, 0x00020000);

    float bw[NT][14];
    unsigned delta[14], tapbit[14];
#pragma unroll
    for (int t = 0; t < 14; ++t) {
        const int k = 2 * t + lh;
        const int kk = k < 27 ? k : 26;
        const int tap = kk / 3, ci = kk - tap * 3;
        const int kh = tap / 3, kw = tap - kh * 3;
        delta[t] = (unsigned)(((kh * W2 + kw) * a.ldx + ci) * 4);
        // a.nchw: x is the network input itself, fp32 planes [batch][3][H][W], no halo: the tap's offset from the lane's
        // pixel in plane 0 (may be negative: unsigned wrap-around), and the tap's bit in the lane's validity mask
        if (a.nchw) delta[t] = (unsigned)(((ci * a.H + kh - 1) * a.W + (kw - 1)) * 4);
        tapbit[t] = 1u << tap;
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const int co = j * 32 + li;
            bw[j][t] = (co < a.Cout && k < 27) ? a.w[(size_t)co * 27 + k] : 0.f;
        }
    }
    float mean[NT], scale[NT], bias[NT];
    double rinv[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int co = j * 32 + li;
        mean[j] = 0.f; scale[j] = 1.f; bias[j] = 0.f; rinv[j] = 1.0;
        if (co < a.Cout) {
            bias[j] = a.bias[co];
            if (a.bn) { mean[j] = a.mean[co]; rinv[j] = a.rinv[co]; scale[j] = a.scale[co]; }
        }
    }

    // Each wave owns a CONTIGUOUS run of tiles, so a lane's image coordinates advance by a constant step from
    // tile to tile: a carry update instead of five integer divisions per tile (which made this HBM-bound kernel
    // VALU bound).  Unit grid: pooling windows (8 per tile, lane = window li/4, corner li%4) with the fused
    // pool, pixels (32 per tile) without.
    const long chunk = (ntiles + nwaves - 1) / nwaves;
    const long t_begin = wave * chunk, t_end = (t_begin + chunk < ntiles) ? t_begin + chunk : ntiles;
    const int Wu = a.pool ? a.W >> 1 : a.W, Hu = a.pool ? a.H >> 1 : a.H;
    const int ustep = a.pool ? 8 : 32;
    const long nunits = (long)a.batch * Hu * Wu;
    long unit = t_begin * ustep + (a.pool ? (li >> 2) : li);      // unit of the NEXT tile to load
    int cn, cy, cx;
    {
        const long uu = unit < nunits ? unit : 0;
        cn = (int)(uu / ((long)Hu * Wu));
        const int rem = (int)(uu - (long)cn * Hu * Wu);
        cy = rem / Wu; cx = rem - cy * Wu;
    }
    auto load_tile = [&](float (&av)[14]) {
        const int py = a.pool ? 2 * cy + ((li >> 1) & 1) : cy, px = a.pool ? 2 * cx + (li & 1) : cx;
        const bool live = unit < nunits;
        const int cn0 = cn;
        unit += ustep;
        cx += ustep;
        while (cx >= Wu) { cx -= Wu; if (++cy >= Hu) { cy = 0; ++cn; } }
        if (a.nchw) {
            // No halo to lean on: taps outside the image must read as zero.  A tile whose 32 pixels all lie in the interior
            // (nine of ten) needs no per-tap test; otherwise tap (kh, kw) is valid iff row bit kh and column bit kw are set.
            const unsigned pbase = live ? (((unsigned)cn0 * 3u * (unsigned)a.H + (unsigned)py) * (unsigned)a.W + (unsigned)px) * 4u : a.xbytes;
            const bool inner = !live || (py > 0 && py < a.H - 1 && px > 0 && px < a.W - 1);
            if (__builtin_amdgcn_ballot_w64(!inner) == 0) {
#pragma unroll
                for (int t = 0; t < 14; ++t)
                    av[t] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xr, live ? pbase + delta[t] : a.xbytes, 0, 0));
            } else {
                const unsigned rm = (py > 0 ? 1u : 0u) | 2u | (py < a.H - 1 ? 4u : 0u);
                const unsigned cm = (px > 0 ? 1u : 0u) | 2u | (px < a.W - 1 ? 4u : 0u);
                // bit 3 kh + kw = row bit kh & column bit kw
                const unsigned tm = live ? ((rm & 1u ? cm : 0u) | (rm & 2u ? cm << 3 : 0u) | (rm & 4u ? cm << 6 : 0u)) : 0u;
#pragma unroll
                for (int t = 0; t < 14; ++t)
                    av[t] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xr, (tm & tapbit[t]) ? pbase + delta[t] : a.xbytes, 0, 0));
            }
            return;
        }
        // rows past the end read out of range: zeros, no traffic (their results are never stored)
        const unsigned base = live ? ((unsigned)(cn0 * H2 + py) * (unsigned)W2 + (unsigned)px) * (unsigned)a.ldx * 4u : a.xbytes;
#pragma unroll
        for (int t = 0; t < 14; ++t)
            av[t] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xr, base == a.xbytes ? base : base + delta[t], 0, 0));
    };
    auto compute_tile = [&](long tile, const float (&av)[14]) {
        f32x16 acc[NT];
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
#pragma unroll
        for (int t = 0; t < 14; ++t)
#pragma unroll
            for (int j = 0; j < NT; ++j)
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[t], bw[j][t], acc[j], 0, 0, 0);
        const long prow = tile * 32 + 4 * lh;
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const int co = j * 32 + li;
            if (a.pool) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const long r0 = prow + 8 * g;
                    const float m = epilogue_f32(pool_pick(acc[j][4 * g], acc[j][4 * g + 1], acc[j][4 * g + 2], acc[j][4 * g + 3], !a.bn || scale[j] >= 0.f),
                                                 a.bn, mean[j], rinv[j], scale[j], bias[j], a.act);
                    if (co < a.Cout && r0 < a.npix) {
                        if (a.y_f16) ((_Float16 *)a.y)[(size_t)(r0 >> 2) * a.ldy + co] = (_Float16)m;
                        else a.y[(size_t)(r0 >> 2) * a.ldy + co] = m;
                    }
                }
                continue;
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const long p = prow + (r & 3) + 8 * (r >> 2);
                if (co < a.Cout && p < a.npix) {
                    const float v = epilogue_f32(acc[j][r], a.bn, mean[j], rinv[j], scale[j], bias[j], a.act);
                    if (a.y_f16) ((_Float16 *)a.y)[(size_t)p * a.ldy + co] = (_Float16)v;
                    else a.y[(size_t)p * a.ldy + co] = v;
                }
            }
        }
    };

    // Software pipeline inside the wave: the 14 MFMAs of tile t+1 are interleaved (in program order -- a wave
    // issues in order) with the epilogue of tile t, whose VALU work is several times the MFMA issue time of a tile;
    // back to back they left the matrix pipe idle during every epilogue (27 % MFMA-busy, 1.4 TB/s).  The interleave
    // only exists inside ONE basic block, so the pipelined form is compiled for the case that matters -- batch-norm +
    // leaky, the four combinations of (fused pool, half output) as compile-time constants -- and out-of-range rows and
    // filters are dropped by the buffer range check of the stores instead of by branches.
    const size_t ybytes_full = a.pool ? (size_t)(a.npix >> 2) * a.ldy : (size_t)a.npix * a.ldy;
    const unsigned esz = a.y_f16 ? 2u : 4u;
    if (NT > 1 || ybytes_full * esz >= 4294967000.0 || t_begin >= t_end || !a.bn || a.act != Y2H_ACT_LEAKY) {
        // (two accumulator sets of NT = 2 would leave one wave per SIMD; outputs beyond the 32-bit buffer range
        // cannot use the range-checked stores; other epilogues: plain per-tile path)
        float a0[14], a1[14];
        long tile = t_begin;
        if (tile < t_end) load_tile(a0);
        for (; tile < t_end; tile += 2) {
            if (tile + 1 < t_end) load_tile(a1);
            compute_tile(tile, a0);
            if (tile + 2 < t_end) load_tile(a0);
            if (tile + 1 < t_end) compute_tile(tile + 1, a1);
        }
        return;
    }
    const __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc((void *)a.y, 0, (unsigned)(ybytes_full * esz), 0x00020000);
    auto pipelined = [&](auto POOLC, auto F16C) {
        constexpr bool POOL = decltype(POOLC)::value, F16 = decltype(F16C)::value;
        auto mfma_tile = [&](const float (&av)[14], f32x16 (&acc)[NT]) {
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
#pragma unroll
            for (int t = 0; t < 14; ++t)
#pragma unroll
                for (int j = 0; j < NT; ++j)
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[t], bw[j][t], acc[j], 0, 0, 0);
        };
        auto put = [&](unsigned long row, int co, float v, bool ok) {
            const unsigned long off = (row * (unsigned long)a.ldy + (unsigned long)co) * (F16 ? 2u : 4u);
            const unsigned o = ok ? (unsigned)off : 0xffffffffu;            // out of range: the store is dropped
            if (F16) __builtin_amdgcn_raw_buffer_store_b16(__builtin_bit_cast(unsigned short, (_Float16)v), yr, o, 0, 0);
            else __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), yr, o, 0, 0);
        };
        auto finish_tile = [&](long tile, const f32x16 (&acc)[NT]) {
            const long prow = tile * 32 + 4 * lh;
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const int co = j * 32 + li;
                if (POOL) {
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const long r0 = prow + 8 * g;
                        const float m = epilogue_f32(pool_pick(acc[j][4 * g], acc[j][4 * g + 1], acc[j][4 * g + 2], acc[j][4 * g + 3], scale[j] >= 0.f),
                                                     true, mean[j], rinv[j], scale[j], bias[j], Y2H_ACT_LEAKY);
                        put((unsigned long)(r0 >> 2), co, m, co < a.Cout && r0 < a.npix);
                    }
                } else {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const long p = prow + (r & 3) + 8 * (r >> 2);
                        put((unsigned long)p, co, epilogue_f32(acc[j][r], true, mean[j], rinv[j], scale[j], bias[j], Y2H_ACT_LEAKY),
                            co < a.Cout && p < a.npix);
                    }
                }
            }
        };
        auto overlap = [&](const float (&av)[14], f32x16 (&acc_next)[NT], long tile_prev, const f32x16 (&acc_prev)[NT]) {
            mfma_tile(av, acc_next);
            finish_tile(tile_prev, acc_prev);
#pragma unroll
            for (int t = 0; t < 14 * NT; ++t) {          // one MFMA, then a share of the epilogue's VALU work and a store
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, POOL ? 7 : 13, 0);
                __builtin_amdgcn_sched_group_barrier(0x040, 1, 0);
            }
        };
        float a0[14], a1[14];
        f32x16 accA[NT], accB[NT];
        long cur = t_begin;
        load_tile(a0);
        if (cur + 1 < t_end) load_tile(a1);
        mfma_tile(a0, accA);
        for (;;) {
            // accA holds tile cur, a1 the inputs of cur + 1
            if (cur + 1 >= t_end) { finish_tile(cur, accA); break; }
            if (cur + 2 < t_end) load_tile(a0);
            overlap(a1, accB, cur, accA);
            ++cur;
            // accB holds tile cur, a0 the inputs of cur + 1
            if (cur + 1 >= t_end) { finish_tile(cur, accB); break; }
            if (cur + 2 < t_end) load_tile(a1);
            overlap(a0, accA, cur, accB);
            ++cur;
        }
    };
    if (a.pool) { if (a.y_f16) pipelined(std::true_type{}, std::true_type{}); else pipelined(std::true_type{}, std::false_type{}); }
    else { if (a.y_f16) pipelined(std::false_type{}, std::true_type{}); else pipelined(std::false_type{}, std::false_type{}); }
}

// ---------------------------------------------------------------------------
// Stem convolutions on a few-channel image: any size / stride with Cin <= 4 (7x7/2 of cfg/yolov1/yolo.cfg,
// resnet50.cfg, extraction.cfg; 11x11/4 of alexnet.cfg; 3x3 stems the first-layer kernel above does not take).
// Same shape of kernel as conv_first_kernel -- no im2col, no activation staging: the input carries a zero halo of
// `pad` pixels ([batch][H+2p][W+2p][ldx]) so no tap needs a bounds test, a wave takes tiles of 32 consecutive output
// pixels and lane (i, half) loads A[i][k], k = 2t + half, with plain dword loads -- but K = size*size*Cin is too long
// to keep a filter column in registers, so the packed weights [2t+half][filter] and the per-k input offsets sit in
// LDS (filled once per workgroup) and one ds_read feeds each MFMA.  K is padded to a multiple of 16 with zero weights.
// ---------------------------------------------------------------------------
template <int NT>
__global__ __launch_bounds__(256) void conv_stem_kernel(ConvK a)
{
    extern __shared__ __attribute__((aligned(16))) float stem_smem[];
    constexpr int NC = NT * 32;
    constexpr int U = 8;                                        // k-pairs per unrolled chunk
    const int lane = threadIdx.x & 63, li = lane & 31, lh = lane >> 5;
    const int K = a.size * a.size * a.Cin;
    const int Tp = ((K + 1) / 2 + U - 1) / U * U;               // k-pairs, padded
    unsigned *dl = (unsigned *)stem_smem;                       // [2][Tp]: byte offset of tap k = 2t + half from the window origin
    float *wl = stem_smem + 2 * Tp;                             // [2*Tp][NC]
    const int W2 = a.W + 2 * a.pad, H2 = a.H + 2 * a.pad;
    for (int k = threadIdx.x; k < 2 * Tp; k += 256) {
        const int kk = k < K ? k : K - 1;                       // padding taps re-read the last one (their weights are zero)
        const int kh = kk / (a.size * a.Cin), rem = kk - kh * a.size * a.Cin;
        const int kw = rem / a.Cin, ci = rem - kw * a.Cin;
        dl[(k & 1) * Tp + (k >> 1)] = (unsigned)(((kh * W2 + kw) * a.ldx + ci) * 4);
    }
    for (int idx = threadIdx.x; idx < 2 * Tp * NC; idx += 256) {
        const int k = idx / NC, co = idx - k * NC;
        wl[idx] = (k < K && co < a.Cout) ? a.w[(size_t)co * K + k] : 0.f;
    }
    __syncthreads();

    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void *)a.x, 0, a.xbytes, 0x00020000);
    float mean[NT], scale[NT], bias[NT];
    double rinv[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int co = j * 32 + li;
        mean[j] = 0.f; scale[j] = 1.f; bias[j] = 0.f; rinv[j] = 1.0;
        if (co < a.Cout) {
            bias[j] = a.bias[co];
            if (a.bn) { mean[j] = a.mean[co]; rinv[j] = a.rinv[co]; scale[j] = a.scale[co]; }
        }
    }
    // contiguous run of tiles per wave: output coordinates advance by a carry update (as in conv_first_kernel)
    const long wave = ((long)blockIdx.x * 256 + threadIdx.x) >> 6;
    const long nwaves = (long)gridDim.x * 4;
    const long ntiles = ((long)a.npix + 31) / 32;
    const long chunk = (ntiles + nwaves - 1) / nwaves;
    const long t_begin = wave * chunk, t_end = (t_begin + chunk < ntiles) ? t_begin + chunk : ntiles;
    long unit = t_begin * 32 + li;
    int cn, cy, cx;
    {
        const long uu = unit < a.npix ? unit : 0;
        cn = (int)(uu / ((long)a.out_h * a.out_w));
        const int rem = (int)(uu - (long)cn * a.out_h * a.out_w);
        cy = rem / a.out_w; cx = rem - cy * a.out_w;
    }
    const unsigned *dlh = dl + lh * Tp;
    const float *wlh = wl + lh * NC + li;
    for (long tile = t_begin; tile < t_end; ++tile) {
        const bool live = unit < a.npix;
        const unsigned base = ((unsigned)(cn * H2 + cy * a.stride) * (unsigned)W2 + (unsigned)(cx * a.stride)) * (unsigned)a.ldx * 4u;
        unit += 32;
        cx += 32;
        while (cx >= a.out_w) { cx -= a.out_w; if (++cy >= a.out_h) { cy = 0; ++cn; } }
        f32x16 acc[NT];
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
        for (int t0 = 0; t0 < Tp; t0 += U) {
            float av[U];
#pragma unroll
            for (int u = 0; u < U; ++u)     // rows past the end read out of range: zeros, no traffic
                av[u] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xr, live ? base + dlh[t0 + u] : a.xbytes, 0, 0));
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int j = 0; j < NT; ++j)
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u], wlh[(size_t)(t0 + u) * 2 * NC + j * 32], acc[j], 0, 0, 0);
        }
        const long prow = tile * 32 + 4 * lh;
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const int co = j * 32 + li;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const long p = prow + (r & 3) + 8 * (r >> 2);
                if (co < a.Cout && p < a.npix)
                    a.y[(size_t)p * a.ldy + co] = epilogue_f32(acc[j][r], a.bn, mean[j], rinv[j], scale[j], bias[j], a.act);
            }
        }
    }
}

// ---------------------------------------------------------------------------
// 3x3 convolution with 32 input channels and <= 64 filters in fp32 (yolo.cfg layer 2: 304x304 32 -> 64 + pool at 608x608,
// 0.91 ms of the 15.6 ms step at 0.76 of the matrix peak on the 64x64 tile, whose nine K-steps of 32 pay a barrier, a
// staging round and a prologue / epilogue share each), weights stationary -- the plan of the fp16 conv_c32 / conv_c64
// kernels carried over to v_mfma_f32_32x32x2_f32:
//   * a wave owns ONE 32-filter tile: the lane's B operands W[filter li][tap][8 g + 4 h + s] (9 taps x 4 groups x 4 steps)
//     are 144 VGPRs, loaded once per kernel; the eight waves of the one workgroup per CU are (filter tile fq, strip pair sg);
//   * a workgroup walks 16x16-pixel output tiles; the 18x18-pixel input patch of the NEXT tile (128 B per pixel) is fetched
//     into registers during the current tile's MFMAs and written to the other LDS buffer afterwards: one barrier per tile,
//     none in the K loop; every input pixel reaches the CU once per tile (1.27x with the halo) instead of nine times;
//   * an MFMA row tile is a 2 x 16 pixel strip in pool-major order; lane (pixel li, half h) takes channels 8 g + 4 h .. + 3 of
//     a tap with ONE ds_read_b128, which feeds four MFMAs (the k pairing of conv_mfma_kernel).  Pixel pitch 144 B, row pitch
//     2688 B: the 16-lane groups of a ds_read_b128 hit sixteen distinct 16-byte slots;
//   * the epilogue is the reference's arithmetic (epilogue_f32 / pool_pick), outputs leave as 16-byte stores through a
//     wave-private LDS transpose.
// K order = (tap, channel): any order gives the same fp32 sum up to association; exact on integer data like every tile.
// ---------------------------------------------------------------------------
template <bool POOL, bool FAST>
__global__ __launch_bounds__(512, 2) void conv_c32_f32_kernel(ConvK a)
{
    constexpr int PW = 18, PIX_B = 144, ROW_B = 2688, BUF_B = PW * ROW_B;
    constexpr int NCH = PW * PW * 8;                 // 16-byte chunks of one patch
    constexpr int NP = (NCH + 511) / 512;            // staging passes
    constexpr int ES_B = 144;                        // scratch row: 32 filters x 4 B + 16 B
    extern __shared__ __attribute__((aligned(16))) unsigned char c32_smem[];
    const int t = threadIdx.x, lane = t & 63, li = lane & 31, lh = lane >> 5;
    const int wv = __builtin_amdgcn_readfirstlane(t >> 6);
    const int fq = wv & 1, sg = wv >> 1;
    const bool BN_ = FAST ? true : (bool)a.bn;
    const int ACT_ = FAST ? (int)Y2H_ACT_LEAKY : a.act;
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void *)a.x, 0, a.xbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc((void *)a.w, 0, a.wbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc((void *)a.y, 0, a.ybytes, 0x00020000);

    f32x4 bw[36];                                    // [tap * 4 + g]: W[co][tap][8 g + 4 lh .. + 3]
    const int co = 32 * fq + li;
    const bool cok = co < a.Cout;
    float mean = 0.f, scale = 1.f, bias = 0.f;
    double rinv = 1.0;
    if (cok) {
        bias = a.bias[co];
        if (BN_) { mean = a.mean[co]; rinv = a.rinv[co]; scale = a.scale[co]; }
    }
#pragma unroll
    for (int kk = 0; kk < 36; ++kk) {
        const unsigned off = cok ? (unsigned)((co * 288 + (kk >> 2) * 32 + (kk & 3) * 8 + 4 * lh) * 4) : a.wbytes;
        bw[kk] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(wr, off, 0, 0));
    }

    const int ldxB = a.ldx * 4;
    int s_lds[NP], s_rel[NP], s_yx[NP];
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        const int q = t + 512 * p;
        const int pixel = q >> 3, part = q & 7;
        const int py = pixel / PW, px = pixel - py * PW;
        s_lds[p] = py * ROW_B + px * PIX_B + part * 16;
        s_rel[p] = ((py - 1) * a.W + (px - 1)) * ldxB + part * 16;
        s_yx[p] = q < NCH ? (py << 8) | px : -1;
    }
    const int tiles_x = a.W >> 4, tpi = (a.H >> 4) * tiles_x;
    const int ntiles = a.batch * tpi;
    u32x4 sreg[NP];
    auto load_tile = [&](int tile) {
        const int n = tile / tpi, rem = tile - n * tpi;
        const int oy0 = (rem / tiles_x) << 4, ox0 = (rem - (rem / tiles_x) * tiles_x) << 4;
        const int base = ((n * a.H + oy0) * a.W + ox0) * ldxB;
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            const int iy = oy0 - 1 + (s_yx[p] >> 8), ix = ox0 - 1 + (s_yx[p] & 255);
            const bool ok = s_yx[p] >= 0 && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
            sreg[p] = __builtin_amdgcn_raw_buffer_load_b128(xr, ok ? (unsigned)(base + s_rel[p]) : a.xbytes, 0, 0);
        }
    };
    auto store_tile = [&](int buf) {
#pragma unroll
        for (int p = 0; p < NP; ++p)
            if (s_yx[p] >= 0) *(u32x4 *)(c32_smem + buf * BUF_B + s_lds[p]) = sreg[p];
    };

    const int a_off = ((li >> 1) & 1) * ROW_B + (2 * (li >> 2) + (li & 1)) * PIX_B + lh * 16;
    const int Hp = a.H >> 1, Wp = a.W >> 1;
    float *es = (float *)(c32_smem + 2 * BUF_B + wv * 32 * ES_B);        // [32 rows][32 filters + pad], wave-private
    const int cbase = 32 * fq + (lane & 7) * 4;                          // first of the 4 filters this lane stores
    const bool fok = cbase < a.Cout;

    int tile = blockIdx.x, cur = 0;
    if (tile < ntiles) { load_tile(tile); store_tile(0); }
    __syncthreads();
    for (; tile < ntiles; tile += gridDim.x, cur ^= 1) {
        const int next = tile + gridDim.x;
        if (next < ntiles) load_tile(next);
        const int n = tile / tpi, rem = tile - n * tpi;
        const int oy0 = (rem / tiles_x) << 4, ox0 = (rem - (rem / tiles_x) * tiles_x) << 4;
#pragma unroll
        for (int rpi = 0; rpi < 2; ++rpi) {
            const int rp = 2 * sg + rpi;                       // strip = output rows oy0 + 2 rp, + 1
            const unsigned char *ap = c32_smem + cur * BUF_B + 2 * rp * ROW_B + a_off;
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
            for (int kh = 0; kh < 3; ++kh)
#pragma unroll
                for (int kw = 0; kw < 3; ++kw)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const f32x4 af = *(const f32x4 *)(ap + kh * ROW_B + kw * PIX_B + g * 32);
                        const f32x4 bf = bw[(kh * 3 + kw) * 4 + g];
#pragma unroll
                        for (int s4 = 0; s4 < 4; ++s4) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af[s4], bf[s4], acc, 0, 0, 0);
                    }
            if (POOL) {
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    es[(2 * g + lh) * (ES_B / 4) + li] = epilogue_f32(pool_pick(acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3],
                                                                                !BN_ || scale >= 0.f), BN_, mean, rinv, scale, bias, ACT_);
                const u32x4 v = *(const u32x4 *)&es[(lane >> 3) * (ES_B / 4) + (lane & 7) * 4];
                const unsigned prow = (unsigned)((n * Hp + (oy0 >> 1) + rp) * Wp + (ox0 >> 1) + (lane >> 3));
                const unsigned off = fok ? (prow * (unsigned)a.ldy + (unsigned)cbase) * 4u : 0xffffffffu;
                __builtin_amdgcn_raw_buffer_store_b128(v, yr, off, 0, 0);
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int rr = (r & 3) + 8 * (r >> 2) + 4 * lh;            // GEMM row = 4 * window + corner
                    es[rr * (ES_B / 4) + li] = epilogue_f32(acc[r], BN_, mean, rinv, scale, bias, ACT_);
                }
#pragma unroll
                for (int p4 = 0; p4 < 4; ++p4) {
                    const int rr = (lane >> 3) + 8 * p4;
                    const u32x4 v = *(const u32x4 *)&es[rr * (ES_B / 4) + (lane & 7) * 4];
                    const int oy = oy0 + 2 * rp + ((rr >> 1) & 1), ox = ox0 + 2 * (rr >> 2) + (rr & 1);
                    const unsigned off = fok ? ((unsigned)((n * a.H + oy) * a.W + ox) * (unsigned)a.ldy + (unsigned)cbase) * 4u : 0xffffffffu;
                    __builtin_amdgcn_raw_buffer_store_b128(v, yr, off, 0, 0);
                }
            }
        }
        if (next < ntiles) store_tile(cur ^ 1);
        __syncthreads();
    }
}

static bool c32_f32_ok(const y2h_conv *d)
{
    if (d->x_f16 || d->y_f16 || d->x_halo || d->x_nchw || getenv("Y2_NO_C32F")) return false;
    if (d->size != 3 || d->stride != 1 || d->pad != 1 || d->c != 32 || d->n > 64 || d->n % 4 != 0) return false;
    if (d->out_h != d->h || d->out_w != d->w || (d->h & 15) || (d->w & 15) || d->ldx % 4 != 0 || d->ldy % 4 != 0) return false;
    if (((uintptr_t)d->x | (uintptr_t)d->w_packed | (uintptr_t)d->y) % 16 != 0) return false;
    if (d->tile_bm || d->ksplit > 1) return false;                         // a measured / forced tile choice wins
    if (getenv("Y2_CONV_TILE")) return false;
    {
        long min_tiles = 512;                                              // a plan for big batches: one workgroup per CU, >= 2 tiles each
        if (const char *m = getenv("Y2_C32F_MIN_TILES")) min_tiles = atol(m);
        if ((long)d->batch * (d->h >> 4) * (d->w >> 4) < min_tiles) return false;
    }
    const double xbytes = (double)d->batch * d->h * d->w * d->ldx * 4.0;
    const double ybytes = (double)d->batch * d->h * d->w * (d->fuse_maxpool2 ? 0.25 : 1.0) * d->ldy * 4.0;
    return xbytes < 4294967000.0 && ybytes < 4294967000.0 && d->w_packed != nullptr;
}

static int c32_f32_launch(const y2h_conv *d, ConvK &a, y2h_stream s)
{
    a.w = d->w_packed;
    a.npix = d->batch * d->h * d->w;
    a.xbytes = (unsigned)((size_t)d->batch * d->h * d->w * d->ldx * 4);
    a.wbytes = (unsigned)((size_t)d->n * 288 * 4);
    a.ybytes = (unsigned)((size_t)(d->fuse_maxpool2 ? a.npix / 4 : a.npix) * d->ldy * 4);
    const bool fast = d->batch_normalize && d->activation == Y2H_ACT_LEAKY;
    void (*fn)(ConvK) = a.pool ? (fast ? conv_c32_f32_kernel<true, true> : conv_c32_f32_kernel<true, false>)
                               : (fast ? conv_c32_f32_kernel<false, true> : conv_c32_f32_kernel<false, false>);
    const size_t lds = (size_t)2 * 18 * 2688 + 8 * 32 * 144;
    {
        static bool attr_set[16][4] = {{false}};
        const int which = (a.pool ? 2 : 0) + (fast ? 1 : 0);
        int dev = 0;
        Y2H_CHECK(hipGetDevice(&dev));
        if (dev < 0 || dev >= 16 || !attr_set[dev][which]) {
            Y2H_CHECK(hipFuncSetAttribute((const void *)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            if (dev >= 0 && dev < 16) attr_set[dev][which] = true;
        }
    }
    long tiles = (long)d->batch * (d->h >> 4) * (d->w >> 4);
    long grid = tiles < 256 ? tiles : 256;
    if (const char *g = getenv("Y2_CONV_GRID")) { if (atol(g) > 0 && atol(g) < grid) grid = atol(g); }
    hipLaunchKernelGGL(fn, dim3((unsigned)grid), dim3(512), lds, S(s), a);
    Y2H_LAUNCH_CHECK();
    return Y2H_OK;
}

// ---------------------------------------------------------------------------
// direct kernel (reference accumulation order; bit-identical to the CPU path)
// weights in the reference's [n][c][kh][kw] layout
// ---------------------------------------------------------------------------
template <bool XH>
__global__ __launch_bounds__(256) void conv_direct_kernel(ConvK a)
{
    const long total = (long)a.batch * a.out_h * a.out_w * a.Cout;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const int co = (int)(idx % a.Cout);
        const long op = idx / a.Cout;
        const int ox = (int)(op % a.out_w);
        const int oy = (int)((op / a.out_w) % a.out_h);
        const int n = (int)(op / ((long)a.out_w * a.out_h));
        const float *wrow = a.w + (size_t)co * a.K;
        float sum = 0.f;
        for (int c = 0; c < a.Cin; ++c)
            for (int kh = 0; kh < a.size; ++kh) {
                const int iy = oy * a.stride + kh - a.pad;
                for (int kw = 0; kw < a.size; ++kw) {
                    const int ix = ox * a.stride + kw - a.pad;
                    float xv = 0.f;
                    if (iy >= 0 && iy < a.H && ix >= 0 && ix < a.W) {
                        const size_t xi = ((size_t)(n * a.H + iy) * a.W + ix) * a.ldx + c;
                        xv = XH ? (float)((const _Float16 *)a.x)[xi] : a.x[xi];
                    }
                    const float prod = wrow[(c * a.size + kh) * a.size + kw] * xv;
                    sum = sum + prod;
                }
            }
        float mean = 0.f, scale = 1.f;
        double rinv = 1.0;
        if (a.bn) { mean = a.mean[co]; rinv = a.rinv[co]; scale = a.scale[co]; }
        const float v = epilogue(sum, a.bn, mean, rinv, scale, a.bias[co], a.act);
        if (a.y_f16) ((_Float16 *)a.y)[(size_t)op * a.ldy + co] = (_Float16)v;
        else a.y[(size_t)op * a.ldy + co] = v;
    }
}

// ---------------------------------------------------------------------------
// dispatch
// ---------------------------------------------------------------------------
struct Variant {
    const char *name;
    int bm, bn, bk, ks;
    void (*fn)(ConvK);
    size_t lds;
    int threads;
    bool attr_set[16];     // per device
    void (*fn_xo)(ConvK);  // the same tile with the XCD-grouped tile order (ConvK.xcd_order), where instantiated
    bool attr_set_xo[16];
    void (*fn_sk)(ConvK);  // the same tile with stream-K work items (ConvK.sk_tiles / sk_wgs), where instantiated
    bool attr_set_sk[16];
    void (*fn_skh)(ConvK); // the same tile, hybrid: whole tiles + the last partial round as stream-K pieces finished in the launch
    bool attr_set_skh[16];
};

#define VAR(BM, BN, BK, KS, WM, WN)                                                             \
    { "conv_mfma_f32_" #BM "x" #BN "x" #BK "_k" #KS, BM, BN, BK, KS, conv_mfma_kernel<BM, BN, BK, KS, WM, WN, Y2_PIPE>, \
      (size_t)2 * (BM + BN) * (BK + 4) * sizeof(float) + (WM * WN == 8 ? (size_t)8 * 16 * 36 * sizeof(float) : 0), WM * WN * 64, {false}, nullptr, {false}, nullptr, {false}, nullptr, {false} }
#define VARSK(BM, BN, BK, KS, WM, WN)                                                           \
    { "conv_mfma_f32_" #BM "x" #BN "x" #BK "_k" #KS, BM, BN, BK, KS, conv_mfma_kernel<BM, BN, BK, KS, WM, WN, Y2_PIPE>, \
      (size_t)2 * (BM + BN) * (BK + 4) * sizeof(float) + (WM * WN == 8 ? (size_t)8 * 16 * 36 * sizeof(float) : 0), WM * WN * 64, {false}, \
      nullptr, {false}, conv_mfma_kernel<BM, BN, BK, KS, WM, WN, Y2_PIPE, false, 1>, {false}, conv_mfma_kernel<BM, BN, BK, KS, WM, WN, Y2_PIPE, false, 2>, {false} }
#define VARXO(BM, BN, BK, KS, WM, WN)                                                           \
    { "conv_mfma_f32_" #BM "x" #BN "x" #BK "_k" #KS, BM, BN, BK, KS, conv_mfma_kernel<BM, BN, BK, KS, WM, WN, Y2_PIPE>, \
      (size_t)2 * (BM + BN) * (BK + 4) * sizeof(float) + (WM * WN == 8 ? (size_t)8 * 16 * 36 * sizeof(float) : 0), WM * WN * 64, {false}, \
      conv_mfma_kernel<BM, BN, BK, KS, WM, WN, Y2_PIPE, true>, {false}, nullptr, {false}, nullptr, {false} }

#ifndef Y2_PIPE
#define Y2_PIPE true
#endif
static Variant g_variants[] = {
    // 8 waves (2 per SIMD), ONE workgroup per CU: the co-resident partner wave that hides LDS/barrier
    // stalls comes from the same workgroup, so a CU never ends up with a lone half-speed pair in the tail
    VARSK(192, 256, 32, 3, 2, 4), VARXO(192, 256, 32, 1, 2, 4),
    VAR(256, 128, 32, 3, 4, 2), VAR(256, 128, 32, 1, 4, 2),
    VAR(256, 64, 32, 3, 8, 1),  VAR(256, 64, 32, 1, 8, 1),
    // 4 waves, two workgroups per CU
    VARSK(128, 128, 32, 3, 2, 2), VARXO(128, 128, 32, 1, 2, 2),
    VAR(128, 64, 32, 3, 2, 2),  VAR(128, 64, 32, 1, 2, 2),
    VAR(64, 64, 32, 3, 2, 2),   VARXO(64, 64, 32, 1, 2, 2),
    VAR(128, 32, 32, 3, 4, 1),  VAR(128, 32, 32, 1, 4, 1),
    VAR(128, 128, 16, 3, 2, 2), VAR(128, 128, 16, 1, 2, 2),
    VAR(128, 64, 16, 3, 2, 2),  VAR(128, 64, 16, 1, 2, 2),
    VAR(64, 64, 16, 3, 2, 2),   VAR(64, 64, 16, 1, 2, 2),
    VAR(128, 32, 16, 3, 4, 1),  VAR(128, 32, 16, 1, 4, 1),
    // 5x5 (alexnet.cfg)
    VAR(128, 128, 32, 5, 2, 2), VAR(64, 64, 32, 5, 2, 2),
    VAR(128, 128, 16, 5, 2, 2), VAR(64, 64, 16, 5, 2, 2),
};

static unsigned long g_xcd_order_launches = 0;
extern "C" unsigned long y2h_xcd_order_launches(void) { return g_xcd_order_launches; }

#ifdef Y2_F32_STAMPS
static void f32_stamps_report(const Variant *v, const y2h_conv *d, long grid, unsigned long long *d_st, y2h_stream s)
{
    if (!getenv("Y2_F32_STAMPS")) return;
    static unsigned long long h[1024 * 8 * 7];
    if (hipStreamSynchronize(S(s)) != hipSuccess || hipMemcpy(h, d_st, sizeof h, hipMemcpyDeviceToHost) != hipSuccess) return;
    double tot[7] = {0, 0, 0, 0, 0, 0, 0};
    const int waves = v->threads / 64;
    for (long b = 0; b < grid; ++b) for (int w = 0; w < waves; ++w) for (int k = 0; k < 7; ++k) tot[k] += (double)h[(b * waves + w) * 7 + k];
    const double tiles = tot[4] > 0 ? tot[4] : 1, steps = tot[3] > 0 ? tot[3] : 1, nw = (double)grid * waves;
    fprintf(stderr, "f32 stamps %s %dx%d c%d n%d: per wave: K-step %.0f cycles of which %.0f at the barrier (%.1f steps per work item), epilogue %.0f per work item; "
            "per wave and launch: K loop %.0f, epilogue %.0f, publish %.0f, wait+gather %.0f kcycles\n",
            v->name, d->h, d->w, d->c, d->n, tot[1] / steps, tot[0] / steps, steps / tiles, tot[2] / tiles, tot[1] / nw / 1e3, tot[2] / nw / 1e3,
            tot[5] / nw / 1e3, tot[6] / nw / 1e3);
}
#endif

static bool mfma_ok(const y2h_conv *d)
{
    if (d->x_f16 || d->y_f16) return false;       // the fp32 matrix-core kernel reads and writes fp32 only
    if (!(d->size == 1 || d->size == 3 || d->size == 5)) return false;
    if (d->stride < 1 || d->pad != d->size / 2) return false;
    if (d->c % 16 != 0 || d->ldx % 4 != 0) return false;
    if (d->out_h != (d->h + 2 * d->pad - d->size) / d->stride + 1 || d->out_w != (d->w + 2 * d->pad - d->size) / d->stride + 1) return false;
    if (d->fuse_maxpool2 && d->stride != 1) return false;
    if (((uintptr_t)d->x | (uintptr_t)d->w_packed) % 16 != 0) return false;
    const double xbytes = (double)d->batch * d->h * d->w * d->ldx * 4.0;
    const double wbytes = (double)d->n * d->size * d->size * d->c * 4.0;
    const double ybytes = (double)d->batch * d->out_h * d->out_w * (d->fuse_maxpool2 ? 0.25 : 1.0) * d->ldy * 4.0;
    if (xbytes >= 4294967000.0 || wbytes >= 4294967000.0 || ybytes >= 4294967000.0) return false;   // 32-bit buffer offsets
    return d->w_packed != nullptr;
}

// Tile choice.  The kernels are MFMA bound, so a CU finishes work at one rate however many
// workgroups share it; what differs between tile shapes is how evenly the grid divides over the
// 256 CUs.  Estimated time = (largest number of tiles any CU ends up with) x (tile area).  A kernel
// whose CU holds `bpc` workgroups is refilled bpc at a time once the first wave of workgroups
// retires (both partners finish together), so its tail is counted in whole groups of bpc -- this is
// what made 728 tiles of 128x128 on 19x19x1024 layers run at 62 % MFMA utilisation.  Ties go to the
// larger tile (fewer L2 bytes per flop).  Y2_CONV_TILE=BMxBN forces a shape (for experiments).
static int variant_bpc(const Variant &v)               // workgroups co-resident on one CU
{
    int bpc = (int)(160 * 1024 / v.lds);
    // two waves per SIMD; the 64x64 tile needs <= 64 VGPRs and fits four
    const int by_waves = (v.bm * v.bn <= 64 * 64 ? 16 : 8) / (v.threads / 64);
    if (bpc > by_waves) bpc = by_waves;
    return bpc < 1 ? 1 : bpc;
}

// Split-K: when even the best tile shape leaves most CUs idle (13x13 grids at small batch, batch-1
// inference), each output tile is cut into `ksplit` K ranges computed by different workgroups; the
// partial sums go through an fp32 workspace and splitk_reduce_kernel.  Chosen together with the tile.
static unsigned long g_skf_launches = 0;
extern "C" unsigned long y2h_f32_stream_k_launches(void) { return g_skf_launches; }

// sk_wgs_out (optional): > 0 when the choice is the stream-K form of the tile (conv_mfma_kernel<..., SKM>) on that many
// workgroups -- then *ksplit_out is 1.  Env: Y2_SKF=0 off; Y2_SKF_WGS=n forces stream-K on n workgroups for the (forced) tile.
// Hybrid stream-K plan (conv_mfma_kernel<..., 2>): of `tiles` output tiles of nk K-steps on `slots` co-resident workgroups, how
// many tiles of the last, partial round are cut along K over all the workgroups (0 = none).  In microseconds of one CU:
// the partial round costs a whole tile time whatever it holds; cut, every workgroup gets R / slots of a tile plus a second
// work item's fixed cost, one 16-byte-store pass of its accumulators, one read pass per earlier piece of the tile it
// finishes, and the launch pays a flag memset.  OFF by default: measured in the benchmark's own context (yolo.cfg 608 b32, four
// interleaved runs, profiles/r03_notes.md section 9) the hybrid instantiation's K loop comes out of the compiler 2.5 % of a
// step slower than the plain one (the kernel sits at the 256-register ceiling; the extra scalar state moves the fragment
// reads of the software pipeline) and cutting the partial rounds wins back 1.2 %.  Env Y2_SKH=1 whenever a partial round
// exists (tests), 2 by the cost model below.
static unsigned long g_skh_launches = 0;
extern "C" unsigned long y2h_f32_hybrid_stream_k_launches(void) { return g_skh_launches; }
extern "C" int y2h_f32_stream_k_timeouts(void)
{
    int n = -1;
    if (hipMemcpyFromSymbol(&n, HIP_SYMBOL(g_skh_timeouts), sizeof n) != hipSuccess) return -1;
    return n;
}
static int skh_plan(long tiles, int nk, long slots, int bm, int bn, int bk, int bpc)
{
    int mode = 0;
    if (const char *f = getenv("Y2_SKH")) mode = atoi(f);
    if (mode <= 0 || tiles < 1 || slots < 2) return 0;
    const long R = tiles < slots ? tiles : tiles % slots;
    if (R == 0) return 0;
    const long share = R * nk / slots;
    if (share < 4) return 0;
    if (mode == 1) return (int)R;
    const double kstep_us = (double)bm * bn * bk / 128.0 * bpc / 2400.0;
    const double piece_us = (double)bm * bn * 4.0 / 65e3;                       // one slot at ~65 GB/s per workgroup (cross-XCD hand-off read)
    const double gain = (1.0 - (double)R / slots) * (nk + 1.2) * kstep_us;
    const double cost = 1.6 * kstep_us + piece_us + piece_us * (double)((slots + R - 1) / R) + 4.0;
    return gain > 1.3 * cost ? (int)R : 0;
}

static long skh_slots(const Variant &v)      // workgroups of a hybrid stream-K launch: all that are co-resident (Y2_SKH_WGS=n: tests, small shapes)
{
    if (const char *f = getenv("Y2_SKH_WGS")) { if (atol(f) >= 2 && atol(f) <= 256L * variant_bpc(v)) return atol(f); }
    return 256L * variant_bpc(v);
}
static size_t skh_ws_bytes(const Variant &v)
{
    const size_t wgs = (size_t)skh_slots(v);
    return wgs * v.bm * v.bn * sizeof(float) + wgs * sizeof(int);
}

static Variant *pick_variant(const y2h_conv *d, int *ksplit_out = nullptr, int *sk_wgs_out = nullptr, int *skh_tiles_out = nullptr)
{
    const int bk = (d->c % 32 == 0) ? 32 : 16;
    const long npix = (long)d->batch * d->out_h * d->out_w;
    const int nk = d->size * d->size * (d->c / bk);
    const int CUS = 256;
    int force_bm = d->tile_bm, force_bn = d->tile_bn, force_split = d->ksplit;      // a tuned descriptor (y2_set_autotune)
    if (const char *f = getenv("Y2_CONV_TILE")) sscanf(f, "%dx%d", &force_bm, &force_bn);
    if (const char *f = getenv("Y2_CONV_KSPLIT")) force_split = atoi(f);
    Variant *best = nullptr;
    double best_cost = 0;
    int best_split = 1, best_sk = 0;
    bool sk_on = true;
    long sk_force = 0;
    if (const char *f = getenv("Y2_SKF")) sk_on = atoi(f) != 0;
    if (const char *f = getenv("Y2_SKF_WGS")) sk_force = atol(f);
    for (Variant &v : g_variants) {
        if (v.bk != bk || v.ks != d->size) continue;
        if (force_bm && (v.bm != force_bm || v.bn != force_bn)) continue;
        const long tiles = ((npix + v.bm - 1) / v.bm) * ((d->n + v.bn - 1) / v.bn);
        const int bpc = variant_bpc(v);
        // split only grids that cannot give every CU one workgroup, keep >= 8 slices per range
        int ksplit = 1;
        if (tiles < CUS && nk >= 16) {
            ksplit = (int)((long)CUS * bpc / tiles);
            // (>= 8 K-steps per range; 1x1 layers >= 4: their ranges are short anyway -- K = 512-1024 is 16-32 steps -- and the
            // batch-1 autotune logs preferred twice the split on every one of them: yolo 416 b1 device part 0.656 -> 0.648 ms)
            { const int per = (d->size == 1 && !getenv("Y2_MODEL_R1")) ? 4 : 8; if (ksplit > nk / per) ksplit = nk / per; }
            if (ksplit > 32) ksplit = 32;
            if (ksplit < 1) ksplit = 1;
        }
        if (force_split > 0) ksplit = force_split <= nk ? force_split : nk;
        const long blocks = tiles * ksplit;
        // work items of the busiest CU.  The grids are persistent (workgroup b walks items b, b + G, ...) and workgroups are
        // placed round-robin, so the CU that hosts workgroups c, c + 256, ... gets items c, c + 256, c + 512, ...: ceil(items / CUs)
        // whatever bpc is.  (Round 1's non-persistent grids refilled a CU bpc workgroups at a time, and the model counted the
        // tail in whole groups of bpc: that over-charged the small tiles on grids of 1.3-2.6 rounds -- autotune logs of yolo 416
        // b8 / yolo9000 544 b8, profiles/r03_notes.md section 11.)
        long per_cu = (blocks + CUS - 1) / CUS;
        if (getenv("Y2_MODEL_R1")) {
            if (blocks <= (long)CUS * bpc) per_cu = (blocks + CUS - 1) / CUS;
            else per_cu = (long)bpc * ((blocks + (long)CUS * bpc - 1) / ((long)CUS * bpc));
        }
        // measured in-tile efficiency relative to the 192x256 tile (yolo.cfg 608x608 b32 sweep,
        // profiles/r01_tile_sweep.txt): bigger wave tiles re-read less LDS per MFMA; the small
        // 64x64 tile wins on short-K 1x1 layers, where prologue/epilogue dominate and four
        // co-resident workgroups overlap them
        double eff = 0.6;
        if (v.bm == 192 && v.bn == 256) eff = 1.0;
        else if (v.bm == 128 && v.bn == 128) eff = 0.96;
        else if (v.bm == 256 && v.bn == 128) eff = 0.945;
        else if (v.bm == 64 && v.bn == 64) eff = (d->size == 1) ? 1.0 : 0.90;
        else if (v.bm == 128 && v.bn == 64) eff = (d->size == 3 && !getenv("Y2_MODEL_R1")) ? 0.92 : 0.86;     // r3: 0.94-0.95 against the 64x64 tile's 0.90 on grids of 3-5 tiles per CU (52x52 / 68x68 128->256 at batch 8)
        else if (v.bm == 256 && v.bn == 64) eff = 0.82;
        else if (v.bm == 128 && v.bn == 32) eff = (d->size == 1) ? 0.95 : 0.7;
        // in CU cycles: one K-step of a tile = bm*bn*bk*2 flop at 256 flop/clk; ~5 K-steps of fixed cost
        // per work item (measured); a split pays the workspace round trip (~4 TB/s) and a launch
        double cost = (double)per_cu * v.bm * v.bn * v.bk / 128.0 * ((double)nk / ksplit + 5.0) / eff;
        if (ksplit > 1) cost += (double)npix * d->n * 4.0 * (ksplit + 1) * 5.75e-4 + 5000.0;
        if (!best || cost < best_cost * 0.999 || (cost <= best_cost * 1.001 && v.bm * v.bn > best->bm * best->bn)) {
            best = &v;
            best_cost = cost;
            best_split = ksplit;
            best_sk = 0;
        }
        // Stream-K candidate of the same tile (grids smaller than the machine): every workgroup gets tiles * nk / wgs K-steps,
        // whatever the tile count -- the integer split above leaves CUs idle whenever tiles * ksplit is not a multiple of the
        // slots (113 tiles x 2 = 226 of 256) and cannot cut a tile into 2.5.  Costs: up to two work items per workgroup (their
        // fixed cost once and a half), pieces = wgs + tiles slots of bm x bn floats written and read once, one more launch.
        if (sk_wgs_out && v.fn_sk && d->n % 4 == 0 && d->ldy % 4 == 0 && force_split <= 0 && d->tile_bm == 0 && nk >= 16 && sk_on) {
            long wgs = (long)CUS * bpc;
            if (wgs > tiles * nk / 8) wgs = tiles * nk / 8;             // shares of >= 8 K-steps
            if (sk_force > 0) wgs = sk_force;
            if (tiles <= wgs && wgs >= 2 && (sk_force > 0 || tiles * 2 < (long)CUS * bpc * 2)) {
                const double share = (double)tiles * nk / wgs;
                const long cu_load = (wgs + CUS - 1) / CUS;                // workgroups a CU hosts
                double c_sk = (double)cu_load * v.bm * v.bn * v.bk / 128.0 * (share + 7.5) / eff;
                c_sk += (double)(wgs + tiles) * v.bm * v.bn * 4.0 * 2.0 * 5.75e-4 + 5000.0;
                if (sk_force > 0 || c_sk < best_cost * 0.97) {
                    best = &v;
                    best_cost = sk_force > 0 ? 0.0 : c_sk;
                    best_split = 1;
                    best_sk = (int)wgs;
                }
            }
        }
    }
    if (ksplit_out) *ksplit_out = best_split;
    if (sk_wgs_out) *sk_wgs_out = best_sk;
    if (skh_tiles_out) {
        *skh_tiles_out = 0;
        if (best && best->fn_skh && best_sk == 0 && best_split == 1 && !getenv("Y2_CONV_GRID")) {
            const long tiles = ((npix + best->bm - 1) / best->bm) * ((d->n + best->bn - 1) / best->bn);
            const int bpc = variant_bpc(*best);
            *skh_tiles_out = skh_plan(tiles, nk, skh_slots(*best), best->bm, best->bn, best->bk, bpc);
        }
    }
    return best;
}

// ---------------------------------------------------------------------------
// Tile autotuning: the cost model above ranks tile shapes from grid arithmetic; on small grids (batch 1..8) its
// error is 10-40 % of a layer (profiles/r02_notes.md), so a plan may instead MEASURE each shape once.  The engine times
// the candidates inside whole forward passes (y2_engine.c autotune_layers): timed in isolation, back to back, a layer
// finds its own weights in the Infinity Cache and small tiles look better than they are in the real sequence.
// ---------------------------------------------------------------------------
// the (tile, K-split) combinations worth measuring for a descriptor; entry 0 is the cost model's own choice
extern "C" int y2h_conv_candidates(const y2h_conv *d, int *bm, int *bn, int *ks, int max)
{
    if (!d || !bm || !bn || !ks || max < 1) return Y2H_EINVAL;
    if (d->x_f16 || d->x_halo != 0 || !mfma_ok(d)) return 0;
    const int bk = (d->c % 32 == 0) ? 32 : 16;
    const int nk = d->size * d->size * (d->c / bk);
    y2h_conv t = *d;
    t.tile_bm = t.tile_bn = t.ksplit = 0;
    int model_split = 1, n = 0;
    Variant *mv = pick_variant(&t, &model_split);
    if (!mv) return 0;
    bm[n] = mv->bm; bn[n] = mv->bn; ks[n] = model_split; ++n;
    for (Variant &v : g_variants) {
        if (v.bk != bk || v.ks != d->size) continue;
        y2h_conv q = t; q.tile_bm = v.bm; q.tile_bn = v.bn;
        int own = 1;
        if (!pick_variant(&q, &own)) continue;
        const int splits[5] = {1, model_split, model_split * 2, model_split / 2, own};
        for (int a = 0; a < 5; ++a) {
            const int k = splits[a];
            bool dup = k < 1 || k > nk;
            for (int b = 0; b < n && !dup; ++b) dup = bm[b] == v.bm && bn[b] == v.bn && ks[b] == k;
            if (dup || n >= max) continue;
            if (k > 1 && (size_t)k * d->batch * d->out_h * d->out_w * d->n * sizeof(float) > ((size_t)256 << 20)) continue;
            bm[n] = v.bm; bn[n] = v.bn; ks[n] = k; ++n;
        }
    }
    return n;
}

extern "C" size_t y2h_conv_workspace_bytes(const y2h_conv *d)
{
    int ksplit = 1;
    if (d->x_f16 && !d->x_halo) return y2_f16_conv_workspace_bytes(d);      // stream-K piece slots of the fp16 256x256 kernel
    if (c32_f32_ok(d)) return 0;
    int sk_wgs = 0, skh_tiles = 0;
    Variant *v = (d->x_halo || d->x_f16 || !mfma_ok(d)) ? nullptr : pick_variant(d, &ksplit, &sk_wgs, &skh_tiles);
    if (!v) return 0;
    if (sk_wgs > 0) return (size_t)2 * sk_wgs * v->bm * v->bn * sizeof(float);          // stream-K piece slots
    if (skh_tiles > 0) return skh_ws_bytes(*v);                                          // one slot and one flag per workgroup
    if (ksplit <= 1) return 0;
    return (size_t)ksplit * d->batch * d->out_h * d->out_w * d->n * sizeof(float);
}

// first-layer kernel: 3 channels, 3x3/1 pad 1, <= 64 filters, input stored with a 1-pixel zero halo
static bool first_ok(const y2h_conv *d)
{
    if (d->x_f16) return false;                   // reads the fp32 network input (writes fp32 or half)
    if (d->c != 3 || d->size != 3 || d->stride != 1 || d->pad != 1 || d->n > 64) return false;
    if (d->out_h != d->h || d->out_w != d->w || d->x_halo != 1) return false;
    const double xbytes = (double)d->batch * (d->h + 2) * (d->w + 2) * d->ldx * 4.0;
    return xbytes < 4294967000.0 && d->w_packed != nullptr;
}

// the same kernel reading the fp32 NCHW network input itself (x_nchw = 1): no halo, border taps masked per tile
static bool first_nchw_ok(const y2h_conv *d)
{
    if (!d->x_nchw || d->x_f16 || d->x_halo) return false;
    if (d->c != 3 || d->size != 3 || d->stride != 1 || d->pad != 1 || d->n > 64) return false;
    if (d->out_h != d->h || d->out_w != d->w || getenv("Y2_NO_FIRST_NCHW")) return false;
    const double xbytes = (double)d->batch * 3.0 * d->h * d->w * 4.0;
    return xbytes < 4294967000.0 && d->w_packed != nullptr;
}

// stem kernel: Cin <= 4, any size / stride, the halo equal to the padding, <= 128 filters, tables within the LDS
static size_t stem_lds_bytes(const y2h_conv *d)
{
    const int K = d->size * d->size * d->c;
    const int Tp = ((K + 1) / 2 + 7) / 8 * 8;
    const int NT = (d->n + 31) / 32;
    return ((size_t)2 * Tp + (size_t)2 * Tp * NT * 32) * sizeof(float);
}

static bool stem_ok(const y2h_conv *d)
{
    if (d->x_f16 || d->y_f16 || d->fuse_maxpool2) return false;
    if (d->c > 4 || d->n > 128 || d->size < 1 || d->stride < 1 || d->pad < 0 || d->x_halo != d->pad) return false;
    if (d->out_h != (d->h + 2 * d->pad - d->size) / d->stride + 1 || d->out_w != (d->w + 2 * d->pad - d->size) / d->stride + 1) return false;
    if (stem_lds_bytes(d) > 160 * 1024) return false;
    const double xbytes = (double)d->batch * (d->h + 2 * d->pad) * (d->w + 2 * d->pad) * d->ldx * 4.0;
    return xbytes < 4294967000.0 && d->w_packed != nullptr;
}

extern "C" int y2h_conv_stem_halo(const y2h_conv *d)
{
    y2h_conv t = *d;
    t.x_halo = t.pad;
    if (!t.w_packed) t.w_packed = (const float *)(uintptr_t)256;
    return stem_ok(&t) ? t.pad : -1;
}

extern "C" int y2h_conv_first_layer_ok(const y2h_conv *d)
{
    y2h_conv t = *d;
    t.x_halo = 1;
    return first_ok(&t) ? 1 : 0;
}

extern "C" int y2h_conv_first_layer_f16_ok(const y2h_conv *d)
{
    y2h_conv t = *d;
    t.x_halo = 1; t.x_f16 = 1; t.ldx = 4;
    if (!t.x) t.x = (const float *)(uintptr_t)256;
    return y2_f16_first_ok(&t) ? 1 : 0;
}

extern "C" int y2h_conv_first_layer_nchw_ok(const y2h_conv *d)
{
    y2h_conv t = *d;
    t.x_halo = 0; t.x_f16 = 0; t.x_nchw = 1;
    if (!t.x) t.x = (const float *)(uintptr_t)256;
    if (!t.w_packed) t.w_packed = (const float *)(uintptr_t)256;
    return (t.y_f16 ? y2_f16_first_nchw_ok(&t) || first_nchw_ok(&t) : first_nchw_ok(&t)) ? 1 : 0;
}

extern "C" int y2h_conv_uses_mfma(const y2h_conv *d)
{
    if (d->x_nchw) return (y2_f16_first_nchw_ok(d) || first_nchw_ok(d)) ? 1 : 0;
    if (d->x_f16) return (y2_f16_first_ok(d) || y2_f16_conv_ok(d)) ? 1 : 0;
    return first_ok(d) || (d->x_halo == 0 && mfma_ok(d) && pick_variant(d)) || (d->c <= 4 && stem_ok(d)) ? 1 : 0;
}

extern "C" const char *y2h_conv_variant(const y2h_conv *d, int strict)
{
    if (d->x_nchw) {
        if (!strict && y2_f16_first_nchw_ok(d)) return d->n <= 32 ? "conv_first_mfma_f16_nchw_c3_n32" : "conv_first_mfma_f16_nchw_c3_n64";
        if (!strict && first_nchw_ok(d)) return d->n <= 32 ? "conv_first_mfma_f32_nchw_c3_n32" : "conv_first_mfma_f32_nchw_c3_n64";
        return nullptr;
    }
    if (!strict && first_ok(d)) return d->n <= 32 ? "conv_first_mfma_f32_c3_n32" : "conv_first_mfma_f32_c3_n64";
    if (d->x_f16) {
        if (!strict && y2_f16_first_ok(d)) return d->n <= 32 ? "conv_first_mfma_f16_c3_n32" : "conv_first_mfma_f16_c3_n64";
        const char *nm = strict ? nullptr : y2_f16_conv_variant(d);
        return nm ? nm : "conv_direct_f16";
    }
    if (!strict && c32_f32_ok(d)) return "conv_c32_f32_16x16";
    if (!strict && d->x_halo == 0 && mfma_ok(d)) {
        Variant *v = pick_variant(d);
        if (v) return v->name;
    }
    if (!strict && stem_ok(d)) return "conv_stem_mfma_f32";
    return "conv_direct_f32";
}

extern "C" int y2h_conv_forward(const y2h_conv *d, int strict, y2h_stream s)
{
    if (!d || !d->x || !d->y || !d->bias) return Y2H_EINVAL;
    if (d->batch <= 0 || d->h <= 0 || d->w <= 0 || d->c <= 0 || d->n <= 0 || d->ldx < d->c || d->ldy < d->n) return Y2H_EINVAL;
    if (d->batch_normalize && (!d->mean || !d->rinv || !d->scale)) return Y2H_EINVAL;
    if (d->out_h != (d->h + 2 * d->pad - d->size) / d->stride + 1) return Y2H_EINVAL;
    if (d->out_w != (d->w + 2 * d->pad - d->size) / d->stride + 1) return Y2H_EINVAL;

    ConvK a;
    memset(&a, 0, sizeof a);
    a.x = d->x; a.y = d->y;
    a.mean = d->mean; a.rinv = d->rinv; a.scale = d->scale; a.bias = d->bias;
    a.H = d->h; a.W = d->w; a.Cin = d->c; a.ldx = d->ldx; a.Cout = d->n; a.ldy = d->ldy;
    a.K = d->size * d->size * d->c;
    a.bn = d->batch_normalize; a.act = d->activation;
    a.size = d->size; a.stride = d->stride; a.pad = d->pad; a.out_h = d->out_h; a.out_w = d->out_w; a.batch = d->batch;
    a.y_f16 = d->y_f16;

    if (d->x_nchw) {
        // the network input itself (fp32 planes): only the fused fp16 first-layer kernel reads that layout
        if (strict || !(y2_f16_first_nchw_ok(d) || first_nchw_ok(d))) return Y2H_EINVAL;
        if (d->fuse_maxpool2 && ((d->h | d->w) & 1)) return Y2H_EINVAL;
        a.pool = d->fuse_maxpool2 ? 1 : 0;
        if (y2_f16_first_nchw_ok(d)) return y2_f16_first_nchw_launch(d, a, s);
        a.nchw = 1;
        a.w = d->w_packed;
        a.npix = d->batch * d->h * d->w;
        a.xbytes = (unsigned)((size_t)d->batch * 3 * d->h * d->w * 4);
        const long ntiles = ((long)a.npix + 31) / 32;
        long blocks = (ntiles + 3) / 4;
        void (*fn)(ConvK) = d->n <= 32 ? conv_first_kernel<1> : conv_first_kernel<2>;
        const long res = resident_blocks((const void *)fn, 256, 0, 3);
        if (blocks > res) blocks = res;              // tiles are grid-strided
        hipLaunchKernelGGL(fn, dim3((unsigned)blocks), dim3(256), 0, S(s), a);
        Y2H_LAUNCH_CHECK();
        return Y2H_OK;
    }
    if (d->fuse_maxpool2) {
        // only the matrix-core kernels pool in their epilogue, and 2x2/2 windows need even dims
        if (strict || (d->h & 1) || (d->w & 1) ||
            !(first_ok(d) || y2_f16_first_ok(d) || y2_f16_conv_ok(d) || (d->x_halo == 0 && mfma_ok(d) && pick_variant(d))))
            return Y2H_EINVAL;
        a.pool = 1;
    }
    if (!strict && y2_f16_first_ok(d)) return y2_f16_first_launch(d, a, s);
    if (!strict && first_ok(d)) {
        a.w = d->w_packed;
        a.npix = d->batch * d->h * d->w;
        a.xbytes = (unsigned)((size_t)d->batch * (d->h + 2) * (d->w + 2) * d->ldx * 4);
        const long ntiles = ((long)a.npix + 31) / 32;
        long blocks = (ntiles + 3) / 4;
        void (*fn)(ConvK) = d->n <= 32 ? conv_first_kernel<1> : conv_first_kernel<2>;
        const long res = resident_blocks((const void *)fn, 256, 0, 3);
        if (blocks > res) blocks = res;              // tiles are grid-strided
        hipLaunchKernelGGL(fn, dim3((unsigned)blocks), dim3(256), 0, S(s), a);
        Y2H_LAUNCH_CHECK();
        return Y2H_OK;
    }
    if (!strict && y2_f16_conv_ok(d)) return y2_f16_conv_launch(d, a, s);
    if (!strict && c32_f32_ok(d)) return c32_f32_launch(d, a, s);
    int ksplit = 1, sk_wgs = 0, skh_tiles = 0;
    Variant *v = (!strict && d->x_halo == 0 && mfma_ok(d)) ? pick_variant(d, &ksplit, &sk_wgs, &skh_tiles) : nullptr;
    if (v && skh_tiles > 0 && (!d->ws || d->ws_bytes < skh_ws_bytes(*v) || ((uintptr_t)d->ws % 16) != 0)) skh_tiles = 0;      // no room: every tile whole
    if (v && sk_wgs > 0 && (!d->ws || d->ws_bytes < (size_t)2 * sk_wgs * v->bm * v->bn * sizeof(float) || ((uintptr_t)d->y % 16) != 0 ||
                            ((uintptr_t)d->ws % 16) != 0)) {
        sk_wgs = 0;                                       // no room for the piece slots: the integer split of the same descriptor
        v = pick_variant(d, &ksplit);
    }
    if (!v && !strict && stem_ok(d)) {
        a.w = d->w_packed;
        a.npix = d->batch * d->out_h * d->out_w;
        a.xbytes = (unsigned)((size_t)d->batch * (d->h + 2 * d->pad) * (d->w + 2 * d->pad) * d->ldx * 4);
        const size_t lds = stem_lds_bytes(d);
        const int nt = (d->n + 31) / 32;
        void (*fn)(ConvK) = nt == 1 ? conv_stem_kernel<1> : nt == 2 ? conv_stem_kernel<2> : nt == 3 ? conv_stem_kernel<3> : conv_stem_kernel<4>;
        {
            static size_t attr_lds[16][4] = {{0}};       // per device and instantiation: largest size requested so far
            int dev = 0;
            Y2H_CHECK(hipGetDevice(&dev));
            if (dev < 0 || dev >= 16 || attr_lds[dev][nt - 1] < lds) {
                Y2H_CHECK(hipFuncSetAttribute((const void *)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                if (dev >= 0 && dev < 16) attr_lds[dev][nt - 1] = lds;
            }
        }
        const long ntiles = ((long)a.npix + 31) / 32;
        int bpc = (int)(160 * 1024 / lds);
        if (bpc > 4) bpc = 4;
        long blocks = (ntiles + 3) / 4;
        if (blocks > 256L * bpc) blocks = 256L * bpc;
        hipLaunchKernelGGL(fn, dim3((unsigned)blocks), dim3(256), lds, S(s), a);
        Y2H_LAUNCH_CHECK();
        return Y2H_OK;
    }
    if (d->x_halo != 0) return Y2H_EINVAL;       // only the first-layer and stem kernels read a haloed input
    if (v) {
        a.ksplit = ksplit;
        if (ksplit > 1) {
            if (!d->ws || d->ws_bytes < (size_t)ksplit * d->batch * d->out_h * d->out_w * d->n * sizeof(float)) return Y2H_EINVAL;
            a.ws = d->ws;
        }
        a.w = d->w_packed;
        a.npix = d->batch * d->out_h * d->out_w;
        a.xbytes = (unsigned)((size_t)d->batch * d->h * d->w * d->ldx * 4);
        a.wbytes = (unsigned)((size_t)d->n * a.K * 4);
        // 16-byte output stores (8-wave tiles): a pixel's filters must start on 16 bytes
        {
            a.ybytes = (unsigned)((size_t)(d->fuse_maxpool2 ? a.npix / 4 : a.npix) * d->ldy * 4);        // < 4 GB: mfma_ok
            a.vec_store = d->ldy % 4 == 0 && d->n % 4 == 0 && ((uintptr_t)d->y % 16) == 0 && ksplit == 1 && !getenv("Y2_F32_SCALAR_STORES");
        }
        a.tiles_n = (d->n + v->bn - 1) / v->bn;
        const long tiles_m = ((long)a.npix + v->bm - 1) / v->bm;
        int dev = 0;
        Y2H_CHECK(hipGetDevice(&dev));
        if (dev < 0 || dev >= 16 || !v->attr_set[dev]) {
            Y2H_CHECK(hipFuncSetAttribute((const void *)v->fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)v->lds));
            if (dev >= 0 && dev < 16) v->attr_set[dev] = true;
        }
        a.ntiles = (int)(tiles_m * a.tiles_n) * ksplit;
        if (getenv("Y2_CONV_XCD_REMAP")) a.dbg |= 64;      // A/B switch; measured 0.2-0.5 % slower in the pipelined step (r02 notes)
        long grid = 256L * variant_bpc(*v);          // persistent: at most what is co-resident
        if (const char *g = getenv("Y2_CONV_GRID")) { if (atol(g) > 0 && atol(g) < grid) grid = atol(g); }   // tests: many tiles per workgroup on small shapes
        if (grid > a.ntiles) grid = a.ntiles;
        {
            // Wide heads: when the weights are the large operand and there are many filter tiles, filter-tile-fastest numbering
            // makes every round of workgroups stream the whole weight matrix again (yolo9000 544 b8, final 1x1: 4.3 GB fetched
            // for 125 MB of operands, profiles/r02_9k544b8_pmc_summary.txt).  See ConvK.xcd_order.  Y2_XCD_ORDER=0/1 forces.
            const double wb = (double)d->n * a.K * 4.0, xb = (double)a.npix * d->c * 4.0;
            bool on = v->fn_xo && ksplit == 1 && grid >= 8 && a.tiles_n >= 16 && wb > 2.0 * xb && wb > 16e6;
            if (const char *f = getenv("Y2_XCD_ORDER")) on = v->fn_xo && atoi(f) != 0 && ksplit == 1 && grid >= 8;
            if (on) {
                a.xcd_order = 1;
                a.tiles_m = (int)tiles_m;
                const double tile_b = (double)v->bm * d->c * d->size * d->size * 4.0;      // input bytes one pixel tile touches (upper bound)
                long pb = (long)(3.0e6 / tile_b);
                if (const char *f = getenv("Y2_XCD_PBLK")) pb = atol(f);
                if (pb < 1) pb = 1;
                if (pb > tiles_m) pb = tiles_m;
                a.pblk = (int)pb;
                grid -= grid % 8;
            }
        }
#ifdef Y2_F32_STAMPS
        static unsigned long long *d_st = nullptr;
        if (!d_st) Y2H_CHECK(hipMalloc((void **)&d_st, 1024 * 8 * 7 * sizeof(unsigned long long)));
        Y2H_CHECK(hipMemsetAsync(d_st, 0, 1024 * 8 * 7 * sizeof(unsigned long long), S(s)));
        a.stamps = d_st;
#endif
        if (sk_wgs > 0) {
            // stream-K: all output tiles' K loops in equal shares over sk_wgs workgroups, then the piece reduction
            a.sk_tiles = (int)(tiles_m * a.tiles_n);
            a.sk_wgs = sk_wgs;
            a.ws = d->ws;
            a.ksplit = 1;
            a.xcd_order = 0;
            if (dev < 0 || dev >= 16 || !v->attr_set_sk[dev]) {
                Y2H_CHECK(hipFuncSetAttribute((const void *)v->fn_sk, hipFuncAttributeMaxDynamicSharedMemorySize, (int)v->lds));
                if (dev >= 0 && dev < 16) v->attr_set_sk[dev] = true;
            }
            hipLaunchKernelGGL(v->fn_sk, dim3((unsigned)sk_wgs), dim3(v->threads), v->lds, S(s), a);
            Y2H_LAUNCH_CHECK();
            const long outs4 = (long)(d->fuse_maxpool2 ? a.npix / 4 : a.npix) * (a.Cout / 4);
            hipLaunchKernelGGL(sk_reduce_kernel, dim3(y2h_grid(outs4, 256)), dim3(256), 0, S(s), a, v->bm, v->bn,
                               d->size * d->size * (d->c / v->bk));
            Y2H_LAUNCH_CHECK();
            ++g_skf_launches;
            return Y2H_OK;
        }
        if (skh_tiles > 0 && !a.xcd_order) {
            // hybrid stream-K: the partial last round's K loops in equal shares over ALL co-resident workgroups, finished in-launch
            const long wgs = skh_slots(*v);
            a.sk_tiles = getenv("Y2_SKH_NOSPLIT") ? 0 : skh_tiles;      // (diagnostic: the hybrid instantiation walking every tile whole)
            a.sk_wgs = (int)wgs;
            a.ws = d->ws;
            a.sk_flags = (int *)(d->ws + (size_t)wgs * v->bm * v->bn);
            Y2H_CHECK(hipMemsetAsync(a.sk_flags, 0, (size_t)wgs * sizeof(int), S(s)));
            if (dev < 0 || dev >= 16 || !v->attr_set_skh[dev]) {
                Y2H_CHECK(hipFuncSetAttribute((const void *)v->fn_skh, hipFuncAttributeMaxDynamicSharedMemorySize, (int)v->lds));
                if (dev >= 0 && dev < 16) v->attr_set_skh[dev] = true;
            }
            hipLaunchKernelGGL(v->fn_skh, dim3((unsigned)wgs), dim3(v->threads), v->lds, S(s), a);
            Y2H_LAUNCH_CHECK();
            ++g_skh_launches;
#ifdef Y2_F32_STAMPS
            f32_stamps_report(v, d, wgs, d_st, s);
#endif
            return Y2H_OK;
        }
        if (a.xcd_order) {
            ++g_xcd_order_launches;
            if (dev < 0 || dev >= 16 || !v->attr_set_xo[dev]) {
                Y2H_CHECK(hipFuncSetAttribute((const void *)v->fn_xo, hipFuncAttributeMaxDynamicSharedMemorySize, (int)v->lds));
                if (dev >= 0 && dev < 16) v->attr_set_xo[dev] = true;
            }
            hipLaunchKernelGGL(v->fn_xo, dim3((unsigned)grid), dim3(v->threads), v->lds, S(s), a);
        } else {
            hipLaunchKernelGGL(v->fn, dim3((unsigned)grid), dim3(v->threads), v->lds, S(s), a);
        }
        Y2H_LAUNCH_CHECK();
#ifdef Y2_F32_STAMPS
        f32_stamps_report(v, d, grid, d_st, s);
#endif
        if (ksplit > 1) {
            hipLaunchKernelGGL(splitk_reduce_kernel, dim3(y2h_grid((long)a.npix * a.Cout, 256)), dim3(256), 0, S(s), a);
            Y2H_LAUNCH_CHECK();
        }
        return Y2H_OK;
    }
    if (!d->w_ref) return Y2H_EINVAL;     // direct kernel needs the reference-layout weights
    a.w = d->w_ref;
    const long total = (long)d->batch * d->out_h * d->out_w * d->n;
    if (d->x_f16) hipLaunchKernelGGL(conv_direct_kernel<true>, dim3(y2h_grid(total, 256, 256 * 32)), dim3(256), 0, S(s), a);
    else hipLaunchKernelGGL(conv_direct_kernel<false>, dim3(y2h_grid(total, 256, 256 * 32)), dim3(256), 0, S(s), a);
    Y2H_LAUNCH_CHECK();
    return Y2H_OK;
}

// ---------------------------------------------------------------------------
// [connected] in the reference's own accumulation order (connected_layer.c:141-176: gemm(0,1,..) = gemm_nt,
// gemm.c:90-106: sum over k ascending of separately rounded products, then C += sum), for the strict mode and for
// shapes the matrix-core kernel does not take.  The input vector of the reference is the producer's output flattened
// as [c][y][x]; here the producer's activations are NHWC, so element k = c*HW + p is read at pixel p, channel c.
// Epilogue = the convolution's (normalize / scale / bias / activation on a 1x1 "image").
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void connected_ref_kernel(const float *__restrict__ x, long x_batch_stride, int ld, int HW, int C,
                                                            const float *__restrict__ w, float *__restrict__ y, int outputs,
                                                            int batch, ConvK a)
{
    const long total = (long)batch * outputs;
    const int K = HW * C;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const int o = (int)(idx % outputs);
        const long b = idx / outputs;
        const float *xb = x + b * x_batch_stride;
        const float *wr = w + (size_t)o * K;
        float sum = 0.f;
        for (int c = 0; c < C; ++c)
            for (int p = 0; p < HW; ++p) {
                const float prod = xb[(size_t)p * ld + c] * wr[(size_t)c * HW + p];
                sum = sum + prod;
            }
        sum = 0.f + sum;
        float mean = 0.f, scale = 1.f;
        double rinv = 1.0;
        if (a.bn) { mean = a.mean[o]; rinv = a.rinv[o]; scale = a.scale[o]; }
        y[idx] = epilogue(sum, a.bn, mean, rinv, scale, a.bias[o], a.act);
    }
}

extern "C" int y2h_connected_ref(const float *x, long x_batch_stride, int ld, int hw, int c, const float *w_ref, float *y,
                                 int outputs, int batch, int batch_normalize, int activation, const float *mean,
                                 const double *rinv, const float *scale, const float *bias, y2h_stream s)
{
    if (!x || !w_ref || !y || !bias || hw <= 0 || c <= 0 || outputs <= 0 || batch <= 0 || ld < c) return Y2H_EINVAL;
    if (batch_normalize && (!mean || !rinv || !scale)) return Y2H_EINVAL;
    ConvK a;
    memset(&a, 0, sizeof a);
    a.mean = mean; a.rinv = rinv; a.scale = scale; a.bias = bias; a.bn = batch_normalize; a.act = activation;
    hipLaunchKernelGGL(connected_ref_kernel, dim3(y2h_grid((long)batch * outputs, 256)), dim3(256), 0, S(s), x, x_batch_stride,
                       ld, hw, c, w_ref, y, outputs, batch, a);
    Y2H_LAUNCH_CHECK();
    return Y2H_OK;
}
