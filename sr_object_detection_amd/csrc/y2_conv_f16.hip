// fp16-storage convolution for gfx950 (BASELINE configs[4]: darknet19_448 classifier, fp16 MFMA path).
//
// The reference has no half-precision path (note 16); this is an engine
// extension selected with y2_set_half().  Activations and packed weights are IEEE half in HBM, the
// contraction runs on v_mfma_f32_32x32x16_f16 with fp32 accumulation, batch-norm is folded into one
// fp32 fma per output (alpha, beta computed on the host in double), the activation is evaluated in
// fp32 and the result is rounded once to half.  Parity is defined against the fp32 CPU path with the
// relaxed tolerance SURVEY 8(d) gives for this config (top-5 identity, 1e-2 on probabilities).
//
// Kernel shape: the same implicit GEMM as conv_mfma_kernel (y2_conv.hip) -- GEMM-M = output pixels
// (NHWC), GEMM-N = filters, K = (kh*3+kw)*Cin + ci, no im2col buffer, per K-step a [BM pixels][BK
// channels] slice of one filter tap and a [BN filters][BK] weight slice staged through LDS with
// buffer loads (padding taps = out-of-range offsets = zeros), double buffered, persistent workgroups
// staging one slice ahead across tile boundaries -- re-tiled for a matrix pipe that is 16x faster
// per flop: 256x256 tiles so that the L2->LDS traffic per MFMA stays under what a CU can pull
// ((BM+BN)*BK*2 bytes per BM*BN*BK*2 flops), BK = 64 halves = one 128-byte line per pixel and tap,
// LDS rows padded by 16 bytes so the ds_read_b128 operand reads (8 halves = the k-fragment of one
// lane of a 32x32x16 MFMA) are bank-conflict free.
#include "y2_conv_shared.hpp"
#include <type_traits>

// Tuning ablations (profiles/r01_notes.md): build with -DY2_F16_ABLATE and set Y2_DBG to a mask of
// 1 no LDS staging writes, 2 no global loads, 4 no output stores, 8 scalar output stores, 16 no barrier,
// 32 no fragment reads.  Results are garbage; compiled out by default so the K loop carries no extra branches.
#ifdef Y2_F16_ABLATE
#define ABL(bit) (a.dbg & (bit))
#else
#define ABL(bit) false
#endif

template <int BM, int BN, int BK, int KS, int WM, int WN, int MINB, bool DB, bool M16>
__global__ __launch_bounds__(WM *WN * 64, MINB) void conv_mfma_f16_kernel(ConvK a)
{
    constexpr int NT = WM * WN * 64;
    constexpr int LS = BK + 8;            // LDS row stride in halves: 2*BK+16 bytes, (bytes/16) odd
    constexpr int CH = BK / 8;            // 16-byte chunks per staged row
    constexpr int RP = NT / CH;           // rows staged per pass
    constexpr int PA = (BM + RP - 1) / RP, PB = (BN + RP - 1) / RP;
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
    constexpr int NG = BK / 16;           // MFMA k-steps per staged slice
    // M16: the contraction runs on v_mfma_f32_16x16x32_f16 (same FLOP per cycle; under a dense matrix load the chip
    // holds a higher clock with this shape than with 32x32x16 -- MI355X_MICROARCH.md clock notes); the wave tile is
    // then 2*TM x 2*TN tiles of 16x16 and a lane's four accumulator registers are four consecutive GEMM rows
    constexpr int TM6 = 2 * TM, TN6 = 2 * TN, HM = TM;      // HM = 16-row tiles per half sub-group
    static_assert(TM >= 1 && TN >= 1, "wave tile must hold at least one 32x32 MFMA tile");
    static_assert(NG >= 2, "the pipelined K-step needs at least two k-groups per slice");

    extern __shared__ __attribute__((aligned(16))) _Float16 smem_h[];
    constexpr int BUF = (BM + BN) * LS;   // halves per buffer

    const int t = threadIdx.x;
    const int lane = t & 63, wv = t >> 6;
    const int wm = wv / WN, wn = wv % WN;
    const int li = lane & 31, lh = lane >> 5;

    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void *)a.x, 0, a.xbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc((void *)a.w, 0, a.wbytes, 0x00020000);

    const int sc = t % CH, sr = t / CH;
    unsigned a_off[PA], a_msk[PA], b_off[PB];
    const int nk = KS * KS * (a.Cin / BK);
    int tap = 0, c0 = 0;
    auto setup_tile = [&](int tile) {
        const bool live = tile < a.ntiles;
        c0 = 0;
        tap = 0;
        const int p0 = a.row0 + (tile / a.tiles_n) * BM, n0 = (tile % a.tiles_n) * BN;
        // Row geometry with ONE coordinate decode per thread and tile: this thread's rows are RP apart, so the
        // image coordinates of the following rows come from a carry update.  (With a matrix pipe this fast the
        // per-row divisions and nine tap tests of the fp32 kernel's setup were ~10 % of a tile.)  Unit grid:
        // pooling windows (row r = 4*window + corner) with the fused pool, pixels without.
        const int Wu = a.pool ? a.W >> 1 : a.W, Hu = a.pool ? a.H >> 1 : a.H;
        const int r0 = p0 + sr;
        const int u0 = a.pool ? r0 >> 2 : r0, tc = r0 & 3;      // RP is a multiple of 4: the corner is the same for all q
        int cn = u0 / (Hu * Wu);
        int cy = (u0 - cn * Hu * Wu) / Wu, cx = u0 - cn * Hu * Wu - cy * Wu;
        constexpr int USTEP_P = RP / 4, USTEP_N = RP;
#pragma unroll
        for (int q = 0; q < PA; ++q) {
            const int r = r0 + q * RP;
            const int py = a.pool ? 2 * cy + (tc >> 1) : cy, px = a.pool ? 2 * cx + (tc & 1) : cx;
            a_off[q] = ((unsigned)((cn * a.H + py) * a.W + px) * (unsigned)a.ldx + (unsigned)sc * 8u) * 2u;
            unsigned m = 0;
            if (live && r < a.npix && (BM % RP == 0 || sr + q * RP < BM)) {
                if (KS == 1) m = 1u;
                else {
                    // bit kh*3+kw = tap inside the image: (column pattern) x (row pattern spread 3 bits apart), no carries
                    const unsigned xm = (px > 0 ? 1u : 0u) | 2u | (px < a.W - 1 ? 4u : 0u);
                    const unsigned ym = (py > 0 ? 1u : 0u) | 8u | (py < a.H - 1 ? 64u : 0u);
                    m = xm * ym;
                }
            }
            a_msk[q] = m;
            cx += a.pool ? USTEP_P : USTEP_N;
            while (cx >= Wu) { cx -= Wu; if (++cy >= Hu) { cy = 0; ++cn; } }
        }
#pragma unroll
        for (int q = 0; q < PB; ++q) {
            const unsigned co = (unsigned)(n0 + sr + q * RP);
            b_off[q] = (live && co < (unsigned)a.Cout && sr + q * RP < BN) ? (co * (unsigned)a.K + (unsigned)sc * 8u) * 2u : a.wbytes;
        }
    };

    f32x16 acc[M16 ? 1 : TM][M16 ? 1 : TN];
    f32x4 acc6[M16 ? TM6 : 1][M16 ? TN6 : 1];
    f32x4 ra[PA], rb[PB];

    auto tile_at = [&](int i) -> int {
        const long tl = (long)blockIdx.x + (long)i * gridDim.x;
        return tl < a.ntiles ? (int)tl : a.ntiles;
    };
    // The staging side runs TWO slices ahead of the matrix side: the slice fetched during K-step k sits
    // in registers for the whole step, is written to LDS at the start of step k+1 and multiplied in step
    // k+2.  With a matrix pipe this fast one K-step is shorter than an L2 round trip, so a one-step
    // distance (as in the fp32 kernel) would stall every step on the loads.
    int lti = 0, stage_k = 0;          // staging cursor: tile index in this workgroup's sequence, slices loaded of it
    setup_tile(tile_at(0));
    auto load_slice = [&]() {
        int delta = 0;
        if (KS == 3) {
            const int kh = tap / 3, kw = tap - kh * 3;
            delta = ((kh - 1) * a.W + (kw - 1)) * a.ldx;
        }
        const unsigned add = (unsigned)((delta + c0) * 2);
#pragma unroll
        for (int q = 0; q < PA; ++q) {
            const bool ok = (a_msk[q] >> tap) & 1u;
            const unsigned off = ok ? a_off[q] + add : a.xbytes;
            ra[q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xr, off, 0, 0));
        }
        const unsigned kadd = (unsigned)((tap * a.Cin + c0) * 2);
#pragma unroll
        for (int q = 0; q < PB; ++q) {
            const unsigned off = (b_off[q] == a.wbytes) ? a.wbytes : b_off[q] + kadd;
            rb[q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(wr, off, 0, 0));
        }
        if (++tap == KS * KS) { tap = 0; c0 += BK; }
        ++stage_k;
    };
    auto store_slice = [&](int buf) {
        _Float16 *As = smem_h + buf * BUF;
        _Float16 *Bs = As + BM * LS;
#pragma unroll
        for (int q = 0; q < PA; ++q)
            if (BM % RP == 0 || sr + q * RP < BM) *(f32x4 *)&As[(sr + q * RP) * LS + sc * 8] = ra[q];
#pragma unroll
        for (int q = 0; q < PB; ++q)
            if (BN % RP == 0 || sr + q * RP < BN) *(f32x4 *)&Bs[(sr + q * RP) * LS + sc * 8] = rb[q];
    };

    load_slice();
    store_slice(0);
    if (stage_k == nk) { setup_tile(tile_at(++lti)); stage_k = 0; }
    load_slice();                      // slice 1 stays in registers
    __syncthreads();

    int cur = 0;
    for (int cti = 0;; ++cti) {
        const int ct = tile_at(cti);
        if (ct >= a.ntiles) break;
        const int p0 = a.row0 + (ct / a.tiles_n) * BM, n0 = (ct % a.tiles_n) * BN;
        if constexpr (M16) {
#pragma unroll
            for (int i = 0; i < TM6; ++i)
#pragma unroll
                for (int j = 0; j < TN6; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc6[i][j][r] = 0.f;
        } else {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        }
        for (int kt = 0; kt < nk; ++kt) {
            // hop of the staging cursor to the workgroup's next tile: here, outside the K-step body, so that
            // the body stays one scheduling region
            if (stage_k == nk) { setup_tile(tile_at(++lti)); stage_k = 0; }
            if constexpr (M16) {
                // lane (row l16, quarter lq) of a 16x16x32 operand holds k = 8*lq .. 8*lq+7 of its row.  A K-step is
                // 2*(BK/32) sub-groups: the B fragments of a 32-deep group are read once, the A row-tiles in two halves
                const int l16 = lane & 15, lq = lane >> 4;
                const _Float16 *As6 = smem_h + cur * BUF + (wm * (BM / WM) + l16) * LS + lq * 8;
                const _Float16 *Bs6 = smem_h + cur * BUF + BM * LS + (wn * (BN / WN) + l16) * LS + lq * 8;
                f16x8 af6[HM], bf6[TN6];
#pragma unroll
                for (int kg = 0; kg < BK / 32; ++kg)
#pragma unroll
                    for (int hf = 0; hf < 2; ++hf) {
                        if (hf == 0) {
#pragma unroll
                            for (int j = 0; j < TN6; ++j) bf6[j] = *(const f16x8 *)&Bs6[j * 16 * LS + kg * 32];
                        }
#pragma unroll
                        for (int i = 0; i < HM; ++i) af6[i] = *(const f16x8 *)&As6[(hf * HM + i) * 16 * LS + kg * 32];
                        if (kg == 0 && hf == 0) store_slice(cur ^ 1);     // the slice loaded one step ago
                        if (kg == 0 && hf == 1) load_slice();             // two slices ahead
                        __builtin_amdgcn_s_setprio(2);
#pragma unroll
                        for (int i = 0; i < HM; ++i)
#pragma unroll
                            for (int j = 0; j < TN6; ++j)
                                acc6[hf * HM + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af6[i], bf6[j], acc6[hf * HM + i][j], 0, 0, 0);
                        __builtin_amdgcn_s_setprio(0);
                        __builtin_amdgcn_sched_barrier(0);
                    }
            } else {
            // lane (row li, half lh) of a 32x32x16 operand holds k = 8*lh .. 8*lh+7 of its row
            const _Float16 *As = smem_h + cur * BUF + (wm * (BM / WM) + li) * LS + lh * 8;
            const _Float16 *Bs = smem_h + cur * BUF + BM * LS + (wn * (BN / WN) + li) * LS + lh * 8;
            // DB: the operand fragments of group g+1 are read while group g multiplies (two register sets);
            // !DB (tiles whose accumulators leave no room): one set, the co-resident wave covers the LDS latency
            f16x8 af[DB ? 2 : 1][TM], bf[DB ? 2 : 1][TN];
            if (DB) {
#pragma unroll
                for (int i = 0; i < TM; ++i) af[0][i] = *(const f16x8 *)&As[i * 32 * LS];
#pragma unroll
                for (int j = 0; j < TN; ++j) bf[0][j] = *(const f16x8 *)&Bs[j * 32 * LS];
            }
#pragma unroll
            for (int kg = 0; kg < NG; ++kg) {
                const int c = DB ? (kg & 1) : 0, n = DB ? (c ^ 1) : 0;
                if (!DB) {
                    if (!ABL(32) || kt == 0) {
#pragma unroll
                    for (int i = 0; i < TM; ++i) af[0][i] = *(const f16x8 *)&As[i * 32 * LS + kg * 16];
#pragma unroll
                    for (int j = 0; j < TN; ++j) bf[0][j] = *(const f16x8 *)&Bs[j * 32 * LS + kg * 16];
                    }
                } else if (kg + 1 < NG) {
#pragma unroll
                    for (int i = 0; i < TM; ++i) af[n][i] = *(const f16x8 *)&As[i * 32 * LS + (kg + 1) * 16];
#pragma unroll
                    for (int j = 0; j < TN; ++j) bf[n][j] = *(const f16x8 *)&Bs[j * 32 * LS + (kg + 1) * 16];
                }
                if (kg == 0 && !ABL(1)) store_slice(cur ^ 1);     // the slice loaded one step ago
                if (kg == 1 && !ABL(2)) load_slice();             // two slices ahead
                if (kg == 1 && ABL(2)) ++stage_k;
                if (!DB) __builtin_amdgcn_s_setprio(2);     // matrix issue ahead of the partner wave's staging traffic
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[c][i], bf[c][j], acc[i][j], 0, 0, 0);
                if (!DB) __builtin_amdgcn_s_setprio(0);
                // issue order inside the group: one MFMA first, the fragment reads of the next group, then
                // the staging work spread one piece per MFMA
                if (DB) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                if (kg + 1 < NG) __builtin_amdgcn_sched_group_barrier(0x100, TM + TN, 0);
                if (kg == 1) {
#pragma unroll
                    for (int q = 0; q < PA + PB; ++q) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x002, 6, 0);
                        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                    }
                }
                if (kg == 0) {
#pragma unroll
                    for (int q = 0; q < PA + PB; ++q) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
                    }
                }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            }   // !M16
            // Raw barrier: wait for this wave's LDS traffic only.  __syncthreads() would also drain vmcnt(0), i.e.
            // wait at every K-step for the global loads of the slice two steps ahead that were just issued --
            // the latency the two-slice distance exists to hide.
#ifdef Y2_F16_SYNCTHREADS
            __syncthreads();
#else
            if (!ABL(16)) {
                __builtin_amdgcn_sched_barrier(0);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
            }
#endif
            cur ^= 1;
        }

        // (the activation is uniform, but tested per output value it is a real branch -- 1844 s_cbranch in the 256x256
        // instantiation; the leaky case is compiled with it as a constant)
        auto epilogue_pass = [&](auto LEAKYC) {
            const int ACT_ = decltype(LEAKYC)::value ? (int)Y2H_ACT_LEAKY : a.act;
            _Float16 *yh = (_Float16 *)a.y;
            if constexpr (M16) {
                // 16x16 tiles: lane (l16, lq) holds filter l16 and GEMM rows 4*lq .. 4*lq+3 -- with the fused pool exactly
                // one pooling window.  Two filter tiles (32 filters = 64 bytes per pixel) go through the LDS scratch
                // together and leave as 16-byte stores (the host selects this variant only when that is possible).
                constexpr int ES = 40;
                static_assert(BUF >= WM * WN * 32 * ES, "epilogue scratch must fit in one staging buffer");
                _Float16 *es = smem_h + (cur ^ 1) * BUF + wv * 32 * ES;
                const int l16 = lane & 15, lq = lane >> 4;
                const int rrow = lane >> 2, rchunk = (lane & 3) * 8;
                if (a.pool) {
#pragma unroll
                    for (int jp = 0; jp < TN; ++jp) {
                        const int cb = n0 + wn * (BN / WN) + jp * 32;
                        const int c0f = cb + l16, c1f = cb + 16 + l16;
                        const float al0 = c0f < a.Cout ? a.alpha[c0f] : 0.f, be0 = c0f < a.Cout ? a.beta[c0f] : 0.f;
                        const float al1 = c1f < a.Cout ? a.alpha[c1f] : 0.f, be1 = c1f < a.Cout ? a.beta[c1f] : 0.f;
#pragma unroll
                        for (int ip = 0; ip < TM; ++ip) {                 // two 16-row tiles = 8 pooled rows
                            const int pb = p0 + wm * (BM / WM) + ip * 32;
#pragma unroll
                            for (int ti = 0; ti < 2; ++ti) {
                                const f32x4 q0 = acc6[2 * ip + ti][2 * jp], q1 = acc6[2 * ip + ti][2 * jp + 1];
                                float m0 = epilogue_fast(q0[0], al0, be0, ACT_), m1 = epilogue_fast(q1[0], al1, be1, ACT_);
#pragma unroll
                                for (int u = 1; u < 4; ++u) {
                                    m0 = __builtin_fmaxf(m0, epilogue_fast(q0[u], al0, be0, ACT_));
                                    m1 = __builtin_fmaxf(m1, epilogue_fast(q1[u], al1, be1, ACT_));
                                }
                                es[(ti * 4 + lq) * ES + l16] = (_Float16)m0;
                                es[(ti * 4 + lq) * ES + 16 + l16] = (_Float16)m1;
                            }
                            const f32x4 v = *(const f32x4 *)&es[rrow * ES + rchunk];
                            const int prow = (pb >> 2) + rrow;
                            if (lane < 32 && 4 * prow < a.npix && cb + rchunk < a.Cout) *(f32x4 *)&yh[(size_t)prow * a.ldy + cb + rchunk] = v;
                        }
                    }
                } else {
#pragma unroll
                    for (int jp = 0; jp < TN; ++jp) {
                        const int cb = n0 + wn * (BN / WN) + jp * 32;
                        const int c0f = cb + l16, c1f = cb + 16 + l16;
                        const float al0 = c0f < a.Cout ? a.alpha[c0f] : 0.f, be0 = c0f < a.Cout ? a.beta[c0f] : 0.f;
                        const float al1 = c1f < a.Cout ? a.alpha[c1f] : 0.f, be1 = c1f < a.Cout ? a.beta[c1f] : 0.f;
#pragma unroll
                        for (int i6 = 0; i6 < TM6; ++i6) {
                            const int pb = p0 + wm * (BM / WM) + i6 * 16;
                            const f32x4 q0 = acc6[i6][2 * jp], q1 = acc6[i6][2 * jp + 1];
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                es[(lq * 4 + r) * ES + l16] = (_Float16)epilogue_fast(q0[r], al0, be0, ACT_);
                                es[(lq * 4 + r) * ES + 16 + l16] = (_Float16)epilogue_fast(q1[r], al1, be1, ACT_);
                            }
                            const f32x4 v = *(const f32x4 *)&es[rrow * ES + rchunk];
                            const int p = pb + rrow;
                            if (p < a.npix && cb + rchunk < a.Cout) *(f32x4 *)&yh[(size_t)p * a.ldy + cb + rchunk] = v;
                        }
                    }
                }
                __syncthreads();
                return;
            }
            // epilogue: lane holds filter li of each 32x32 tile and 16 pixels
            if (a.vec_store) {
                // Half outputs go out as 16-byte stores: a lane of the accumulator layout owns ONE filter of 16
                // pixels, which would be sixteen 2-byte stores per 32x32 tile (measured: 38 % of the kernel).  Each
                // wave transposes its tiles through 2.5 KB of the LDS buffer the K loop has just released (the other
                // buffer already holds the next tile's first slice): 16 ds_write_b16, 2 ds_read_b128, 2 stores of
                // 8 consecutive filters per lane.  LDS operations of one wave execute in order, so the wave-private
                // scratch needs no barrier; the workgroup barrier below keeps the next K-step's staging writes out.
                constexpr int ES = 40;                 // scratch row stride in halves (32 filters + 16 bytes)
                static_assert(BUF >= WM * WN * 32 * ES, "epilogue scratch must fit in one staging buffer");
                _Float16 *es = smem_h + (cur ^ 1) * BUF + wv * 32 * ES;
                const int rrow = lane >> 2, rchunk = (lane & 3) * 8;
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const int cb = n0 + wn * (BN / WN) + j * 32;       // first filter of this 32-wide tile
                    const int co = cb + li;
                    const bool cok = co < a.Cout;
                    const float alpha = cok ? a.alpha[co] : 0.f, beta = cok ? a.beta[co] : 0.f;
#pragma unroll
                    for (int i = 0; i < TM; ++i) {
                        const int pb = p0 + wm * (BM / WM) + i * 32;   // first GEMM row of this tile
                        if (a.pool) {
#pragma unroll
                            for (int g = 0; g < 4; ++g) {
                                float m = epilogue_fast(acc[i][j][4 * g], alpha, beta, ACT_);
#pragma unroll
                                for (int u = 1; u < 4; ++u) {
                                    const float v = epilogue_fast(acc[i][j][4 * g + u], alpha, beta, ACT_);
                                    m = (v > m) ? v : m;
                                }
                                es[(2 * g + lh) * ES + li] = (_Float16)m;       // pooled row (pb + 8g + 4lh) / 4 - pb / 4
                            }
                            const f32x4 v = *(const f32x4 *)&es[rrow * ES + rchunk];
                            const int prow = (pb >> 2) + rrow;
                            if (lane < 32 && 4 * prow < a.npix && cb + rchunk < a.Cout && !ABL(4))
                                *(f32x4 *)&yh[(size_t)prow * a.ldy + cb + rchunk] = v;
                        } else {
#pragma unroll
                            for (int r = 0; r < 16; ++r)
                                es[((r & 3) + 8 * (r >> 2) + 4 * lh) * ES + li] = (_Float16)epilogue_fast(acc[i][j][r], alpha, beta, ACT_);
#pragma unroll
                            for (int h2 = 0; h2 < 2; ++h2) {
                                const f32x4 v = *(const f32x4 *)&es[(rrow + 16 * h2) * ES + rchunk];
                                const int p = pb + rrow + 16 * h2;
                                if (p < a.npix && cb + rchunk < a.Cout && !ABL(4)) *(f32x4 *)&yh[(size_t)p * a.ldy + cb + rchunk] = v;
                            }
                        }
                    }
                }
                __syncthreads();
                return;
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int co = n0 + wn * (BN / WN) + j * 32 + li;
                const bool cok = co < a.Cout && !ABL(4);
                const float alpha = cok ? a.alpha[co] : 0.f, beta = cok ? a.beta[co] : 0.f;
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    const int prow = p0 + wm * (BM / WM) + i * 32 + 4 * lh;
                    if (a.pool) {
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            const int r0 = prow + 8 * g;
                            float m = epilogue_fast(acc[i][j][4 * g], alpha, beta, ACT_);
#pragma unroll
                            for (int u = 1; u < 4; ++u) {
                                const float v = epilogue_fast(acc[i][j][4 * g + u], alpha, beta, ACT_);
                                m = (v > m) ? v : m;
                            }
                            if (cok && r0 < a.npix) {
                                const size_t o = (size_t)(r0 >> 2) * a.ldy + co;
                                if (a.y_f16) yh[o] = (_Float16)m; else a.y[o] = m;
                            }
                        }
                    } else {
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int p = prow + (r & 3) + 8 * (r >> 2);
                            if (cok && p < a.npix) {
                                const float v = epilogue_fast(acc[i][j][r], alpha, beta, ACT_);
                                const size_t o = (size_t)p * a.ldy + co;
                                if (a.y_f16) yh[o] = (_Float16)v; else a.y[o] = v;
                            }
                        }
                    }
                }
            }
        };
        if (a.act == Y2H_ACT_LEAKY) epilogue_pass(std::true_type{});
        else epilogue_pass(std::false_type{});
    }
}

// ---------------------------------------------------------------------------
// 256x256x64 tile, LDS-DMA staging, two 32-MFMA phases per K-tile (the dominant fp16 kernel).
//
// The register-staged kernel above spends its K loop at about twice its pure-MFMA time: per K-tile every thread
// issues 8 buffer loads, holds them in 32 VGPRs, writes them to LDS with 8 ds_write_b128 and the whole workgroup meets
// at one barrier, with both waves of a SIMD reaching their matrix work, their LDS bursts and that barrier together
// (profiles/r01_notes.md).  This kernel is built around `buffer_load_dwordx4 ... lds` instead:
//
//   * staging is LDS-DMA: no staging VGPRs, no ds_write; a wave-instruction moves 8 rows x 128 B (one 64-channel
//     chunk of one filter tap of 8 pixels, or of 8 filters).  The DMA destination is lane-linear (wave base + lane*16),
//     so rows cannot be padded; bank conflicts of the ds_read_b128 fragment reads are removed by an XOR swizzle of the
//     16-byte chunk index with (row >> 1) & 7, applied to the per-lane SOURCE address of the DMA and to the read
//     address (both sides or neither).  Padding taps and rows past the end are out-of-range buffer offsets: the DMA
//     writes zeros (tools/probes/glds_oob.hip).
//   * a K-tile (64 channels of one tap) is four half-tiles of 16 KB -- A-h0 (GEMM rows 0..127), B-h0 (filters 0..127),
//     B-h1, A-h1 -- in one of two 64 KB buffers.  A wave (wm, wn) owns rows {64 wm .. +64} of EACH A half and filters
//     {32 wn .. +32} of EACH B half, so its 128 x 64 output is four quadrants (A half, B half) of 16
//     v_mfma_f32_16x16x32_f16 each, and a K-tile is two phases: A-h0 against both B halves, then A-h1 against both.
//   * phase = { fragment reads, DMA of the next K-tile's half-tiles, counted s_waitcnt vmcnt } s_barrier
//     { lgkmcnt(0), 32 MFMA } s_barrier (see the `phase` lambda for the counts).  The waits never drain (raw s_barrier:
//     __syncthreads() would add vmcnt(0)); staging runs across K-tile and tile boundaries (persistent workgroups).
//     Waves 4..7 run one barrier behind waves 0..3, so on every SIMD one wave is in its matrix segment while its
//     partner reads fragments and issues DMA (without the skew the kernel is 20 % slower).
//     Hazards, in barriers: a half-tile is read >= 1 barrier after the wait + barrier that retire it for every wave
//     (read-after-write), and a slot is re-filled >= 4 barriers after its last fragment read (write-after-read; one
//     barrier of that is eaten by the skew between the wave groups).
//   * the epilogue is the 16x16-tile one of the register-staged kernel (folded batch-norm, activation, optional 2x2
//     maxpool over the four accumulator registers of a lane, 16-byte stores through a wave-private LDS transpose) with
//     its own 20 KB of LDS, so it never touches a buffer a DMA may be writing.
// ---------------------------------------------------------------------------
typedef __attribute__((address_space(3))) void lds_void;
// Diagnostic build (-DP8_STAMPS, tools/build_variant.sh): every wave accumulates s_memtime deltas per phase segment and
// writes them to a.ws at the end; y2_f16_conv_launch prints the averages.  Stamps cost ~10 % and serialise the fragment
// reads in front of the first barrier; never compiled into the product.
#ifdef P8_STAMPS
#define P8_STAMP(k) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); st_acc[k] += t_ - st_prev; st_prev = t_; } while (0)
#else
#define P8_STAMP(k) do { } while (0)
#endif

// One quadrant (A half x, B half y) of a wave's 128 x 64 share of a 256x256 tile: four 16-row tiles x two 16-filter tiles,
// lane (l16, lq) holds filter l16 and GEMM rows 4 lq .. 4 lq + 3 of each (with the fused pool: one pooling window).
// Folded batch-norm + activation, optional 2x2 maxpool, half outputs as 16-byte stores through the wave-private LDS
// scratch `es` (LDS operations of one wave execute in order: no barrier).  pq = first GEMM row, cb = first filter.
// Shared by conv_p8_f16_kernel and the stream-K fix-up kernel, so a fixed-up tile takes the same arithmetic.
__device__ __forceinline__ void p8_epilogue_quadrant(const ConvK &a, const int ACT_, const f32x4 (&q)[4][2], const int pq, const int cb,
                                                     const float al0, const float be0, const float al1, const float be1,
                                                     _Float16 *es, const int lane)
{
    constexpr int ES = 40;
    _Float16 *yh = (_Float16 *)a.y;
    const int l16 = lane & 15, lq = lane >> 4;
    const int rrow = lane >> 2, rchunk = (lane & 3) * 8;
    if (a.pool) {
#pragma unroll
        for (int ip = 0; ip < 2; ++ip) {            // two 16-row tiles = 8 pooled rows
            const int pb = pq + ip * 32;
#pragma unroll
            for (int ti = 0; ti < 2; ++ti) {
                const f32x4 q0 = q[2 * ip + ti][0], q1 = q[2 * ip + ti][1];
                float m0 = epilogue_fast(q0[0], al0, be0, ACT_), m1 = epilogue_fast(q1[0], al1, be1, ACT_);
#pragma unroll
                for (int u = 1; u < 4; ++u) {
                    m0 = __builtin_fmaxf(m0, epilogue_fast(q0[u], al0, be0, ACT_));
                    m1 = __builtin_fmaxf(m1, epilogue_fast(q1[u], al1, be1, ACT_));
                }
                es[(ti * 4 + lq) * ES + l16] = (_Float16)m0;
                es[(ti * 4 + lq) * ES + 16 + l16] = (_Float16)m1;
            }
            const f32x4 v = *(const f32x4 *)&es[rrow * ES + rchunk];
            const int prow = (pb >> 2) + rrow;
            if (lane < 32 && 4 * prow < a.npix && cb + rchunk < a.Cout) *(f32x4 *)&yh[(size_t)prow * a.ldy + cb + rchunk] = v;
        }
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int pb = pq + i * 16;
            const f32x4 q0 = q[i][0], q1 = q[i][1];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                es[(lq * 4 + r) * ES + l16] = (_Float16)epilogue_fast(q0[r], al0, be0, ACT_);
                es[(lq * 4 + r) * ES + 16 + l16] = (_Float16)epilogue_fast(q1[r], al1, be1, ACT_);
            }
            const f32x4 v = *(const f32x4 *)&es[rrow * ES + rchunk];
            const int p = pb + rrow;
            if (p < a.npix && cb + rchunk < a.Cout) *(f32x4 *)&yh[(size_t)p * a.ldy + cb + rchunk] = v;
        }
    }
}

// Stream-K share of workgroup w: K-tile iterations [w * I / G, (w + 1) * I / G) of the I = sk_tiles * nk iterations of the
// tail tiles (the kernel and the fix-up launch must agree on this arithmetic)
__device__ __host__ static inline long sk_share_begin(long w, long I, long G) { return w * I / G; }

// Second launch of a stream-K convolution: block (tail tile ts, wave wv of the producing workgroups), its four waves take
// the four quadrants of that wave's share.  A piece slot holds the accumulators as the producing thread held them:
// [slot][e = ((x*2+y)*4+i)*2+j][thread 0..511] f32x4, so both sides move whole 1 KB lines.  Pieces are added in ascending
// K order (= ascending workgroup number), a fixed order: results are reproducible run to run.
__global__ __launch_bounds__(256) void conv_p8_fixup_kernel(ConvK a, int nk)
{
    __shared__ __attribute__((aligned(16))) _Float16 es_all[4 * 32 * 40];
    const int ts = blockIdx.x >> 3, wv = blockIdx.x & 7;
    const int qd = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int x = qd >> 1, y = qd & 1, wm = wv >> 2, wn = wv & 3;
    const long I = (long)a.sk_tiles * nk, G = a.sk_wgs;
    const long tb = (long)ts * nk, te = tb + nk;
    long w = tb * G / I;
    while (w > 0 && sk_share_begin(w, I, G) > tb) --w;
    while (w + 1 < G && sk_share_begin(w + 1, I, G) <= tb) ++w;        // first share that reaches into this tile
    f32x4 q[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) q[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (; w < G; ++w) {
        const long lo = sk_share_begin(w, I, G), hi = sk_share_begin(w + 1, I, G);
        if (lo >= te) break;
        if (hi <= lo) continue;
        const int slot = 2 * (int)w + (lo < tb ? 1 : 0);       // a share that began in the previous tile: its second piece
        const f32x4 *src = (const f32x4 *)a.ws + ((size_t)slot * 32 + (size_t)(x * 2 + y) * 8) * 512 + wv * 64 + lane;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const f32x4 v = src[(size_t)(i * 2 + j) * 512];
                q[i][j] = q[i][j] + v;
            }
    }
    const int tile = a.ntiles - a.sk_tiles + ts;
    const int p0 = (tile / a.tiles_n) * 256, n0 = (tile % a.tiles_n) * 256;
    const int cb = n0 + y * 128 + wn * 32, l16 = lane & 15;
    const int c0f = cb + l16, c1f = cb + 16 + l16;
    const float al0 = c0f < a.Cout ? a.alpha[c0f] : 0.f, be0 = c0f < a.Cout ? a.beta[c0f] : 0.f;
    const float al1 = c1f < a.Cout ? a.alpha[c1f] : 0.f, be1 = c1f < a.Cout ? a.beta[c1f] : 0.f;
    p8_epilogue_quadrant(a, a.act, q, p0 + x * 128 + wm * 64, cb, al0, be0, al1, be1, es_all + qd * 32 * 40, lane);
}

template <int KS>
__global__ __launch_bounds__(512, 1) void conv_p8_f16_kernel(ConvK a)
{
#ifdef P8_STAMPS
    unsigned long long st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, st_prev = 0;
#endif
    constexpr int BM = 256, BN = 256, BK = 64;
    constexpr int HALF_B = 128 * 128;           // bytes of a half-tile: 128 rows x 64 halves
    constexpr int BUF_B = 4 * HALF_B;           // A-h0 | A-h1 | B-h0 | B-h1
    constexpr int ES = 40;                      // epilogue scratch row stride in halves (32 filters + 16 B)
    extern __shared__ __attribute__((aligned(16))) unsigned char p8_smem[];     // [2][BUF_B] + [8 waves][32][ES] halves

    const int t = threadIdx.x, lane = t & 63;
    const int wv = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wm = wv >> 2, wn = wv & 3;
    const int l16 = lane & 15, lq = lane >> 4;

    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void *)a.x, 0, a.xbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc((void *)a.w, 0, a.wbytes, 0x00020000);

    // ---- staging role: 16-byte chunk `sc` of rows sr + 64 q (q = 0..3) of the A tile and of the B tile ----
    const int sc = t & 7, sr = t >> 3;
    const unsigned scs = (unsigned)(sc ^ ((sr >> 1) & 7)) * 16u;      // swizzled source chunk (bytes); (row >> 1) & 7 == (sr >> 1) & 7
    unsigned a_off[4], a_msk[4], b_off[4];
    const int nk = KS * KS * (a.Cin / BK);
    // Optional XCD-aware placement (env Y2_P8_REMAP): workgroups are dealt round-robin over the 8 XCDs (b and b + 8 share
    // one L2); remapped, each XCD owns a contiguous run of tile numbers in every round.  Measured neutral to slightly
    // negative on darknet19_448 b128 (the kernel is bound by the CU's own L2 -> LDS path, not by L2 misses), so it is off.
    int wgid = blockIdx.x;
    if (a.dbg & 64) {
        const int nwg = gridDim.x, xcd = wgid & 7, q = nwg >> 3, r = nwg & 7;
        wgid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (wgid >> 3);
    }
    // Work items of this workgroup: first its stream-K share of the tail tiles (0, 1 or 2 pieces = (tile, K range), raw sums
    // to piece slot 2*wg + piece), then whole tiles wg, wg + G, ... of the first ntiles - sk_tiles.  The pieces come FIRST, so
    // their 256 KB stores drain under the whole tiles that follow and every workgroup ends on a full-tile epilogue.
    const int ndp = a.ntiles - a.sk_tiles;
    int n_sk = 0, sk_t0 = 0, sk_kb0 = 0, sk_ke0 = 0, sk_ke1 = 0;
    if (a.sk_tiles > 0 && wgid < a.sk_wgs) {
        const long I = (long)a.sk_tiles * nk;
        const long lo = sk_share_begin(wgid, I, a.sk_wgs), hi = sk_share_begin(wgid + 1, I, a.sk_wgs);
        if (hi > lo) {
            const int t0 = (int)(lo / nk), k0 = (int)(lo - (long)t0 * nk), len = (int)(hi - lo);      // len <= nk: sk_tiles <= sk_wgs
            sk_t0 = ndp + t0; sk_kb0 = k0; sk_ke0 = k0 + len < nk ? k0 + len : nk; n_sk = 1;
            if (k0 + len > nk) { sk_ke1 = k0 + len - nk; n_sk = 2; }
        }
    }
    struct Item { int tile, kb, ke, slot; };
    auto item_at = [&](int i) -> Item {
        if (i < n_sk) return i == 0 ? Item{sk_t0, sk_kb0, sk_ke0, 2 * wgid} : Item{sk_t0 + 1, 0, sk_ke1, 2 * wgid + 1};
        const long tl = (long)wgid + (long)(i - n_sk) * gridDim.x;
        if (tl < ndp) return Item{(int)tl, 0, nk, -1};
        return Item{a.ntiles, 0, 0x40000000, -1};               // past the end: everything masked, never hop again
    };
    auto setup_tile = [&](int tile) {
        const bool live = tile < a.ntiles;
        const int p0 = (tile / a.tiles_n) * BM, n0 = (tile % a.tiles_n) * BN;
        const int Wu = a.pool ? a.W >> 1 : a.W, Hu = a.pool ? a.H >> 1 : a.H;
        const int r0 = p0 + sr;
        const int u0 = a.pool ? r0 >> 2 : r0, tc = r0 & 3;       // rows of one thread are 64 apart: same pooling-window corner
        int cn = u0 / (Hu * Wu);
        int cy = (u0 - cn * Hu * Wu) / Wu, cx = u0 - cn * Hu * Wu - cy * Wu;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int r = r0 + q * 64;
            const int py = a.pool ? 2 * cy + (tc >> 1) : cy, px = a.pool ? 2 * cx + (tc & 1) : cx;
            a_off[q] = (unsigned)((cn * a.H + py) * a.W + px) * (unsigned)a.ldx * 2u + scs;
            unsigned m = 0;
            if (live && r < a.npix) {
                if (KS == 1) m = 1u;
                else {
                    const unsigned xm = (px > 0 ? 1u : 0u) | 2u | (px < a.W - 1 ? 4u : 0u);
                    const unsigned ym = (py > 0 ? 1u : 0u) | 8u | (py < a.H - 1 ? 64u : 0u);
                    m = xm * ym;
                }
            }
            a_msk[q] = m;
            cx += a.pool ? 16 : 64;
            while (cx >= Wu) { cx -= Wu; if (++cy >= Hu) { cy = 0; ++cn; } }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const unsigned co = (unsigned)(n0 + sr + q * 64);
            b_off[q] = (live && co < (unsigned)a.Cout) ? co * (unsigned)a.K * 2u + scs : a.wbytes;
        }
    };
    // cursor of the staging side: K-tile (s_tap, s_c0) of work item s_lti (s_rem K-tiles of it still to stage), in LDS buffer s_buf
    int s_tap = 0, s_c0 = 0, s_rem = 0, s_lti = 0, s_buf = 0;
    auto begin_item = [&](const Item &it) {
        setup_tile(it.tile);
        s_c0 = (it.kb / (KS * KS)) * BK;
        s_tap = it.kb % (KS * KS);
        s_rem = it.ke - it.kb;
    };
    begin_item(item_at(0));
    // one half-tile = two DMA instructions per thread (U = 0, 1): J = 0 A-h0, 1 B-h0, 2 B-h1, 3 A-h1
    auto stage_one = [&](auto JC, auto UC) {
        constexpr int J = decltype(JC)::value, U = decltype(UC)::value;
        constexpr bool IS_A = (J == 0 || J == 3);
        constexpr int Q = ((J == 0 || J == 1) ? 0 : 2) + U;
        constexpr int SLOT = J == 0 ? 0 : J == 3 ? 1 : J == 1 ? 2 : 3;
        unsigned char *dst = p8_smem + s_buf * BUF_B + SLOT * HALF_B + wv * (8 * 128) + U * (64 * 128);      // + lane * 16 by the DMA
        if (IS_A) {
            int delta = 0;
            if (KS == 3) {
                const int kh = s_tap / 3, kw = s_tap - kh * 3;
                delta = ((kh - 1) * a.W + (kw - 1)) * a.ldx;
            }
            const unsigned add = (unsigned)((delta + s_c0) * 2);
            bool ok = (a_msk[Q] >> s_tap) & 1u;
#ifdef P8_ABLATE
            // timing ablation (wrong results; tools/build_variant.sh ablate -DP8_ABLATE, Y2_DBG bits): 128 = the A half-tiles of every
            // tap but the centre one are fetched out of range (zeros, no memory traffic; same instruction count and waits):
            // what re-using ONE staged input patch for the nine taps would save on the L2 -> LDS path; 256 = the same for B
            if ((a.dbg & 128) && KS == 3 && s_tap != 4) ok = false;
            if ((a.dbg & 512) && KS == 3 && s_tap != 4) return;      // 512 = those DMA instructions are not issued at all (the counted waits then pass early)
#endif
            const unsigned off = ok ? a_off[Q] + add : a.xbytes;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(xr, (lds_void *)dst, 16, off, 0, 0, 0);
        } else {
            const unsigned kadd = (unsigned)((s_tap * a.Cin + s_c0) * 2);
            unsigned off = (b_off[Q] == a.wbytes) ? a.wbytes : b_off[Q] + kadd;
#ifdef P8_ABLATE
            if (a.dbg & 256) off = a.wbytes;
#endif
            __builtin_amdgcn_raw_ptr_buffer_load_lds(wr, (lds_void *)dst, 16, off, 0, 0, 0);
        }
    };
    auto stage = [&](auto JC) {
        stage_one(JC, std::integral_constant<int, 0>{});
        stage_one(JC, std::integral_constant<int, 1>{});
    };
    auto advance = [&]() {        // the cursor moves to the next K-tile (of the next work item after the last one)
        s_buf ^= 1;
        if (++s_tap == KS * KS) { s_tap = 0; s_c0 += BK; }
        if (--s_rem == 0) begin_item(item_at(++s_lti));
    };

    // ---- matrix side: fragment addresses.  Lane (l16, lq) of a 16x16x32 operand holds k = 32 kk + 8 lq .. +7 of row l16 ----
    const unsigned swz = (unsigned)((l16 >> 1) & 7);
    const unsigned fo0 = (unsigned)l16 * 128u + (((unsigned)lq) ^ swz) * 16u;             // kk = 0
    const unsigned fo1 = (unsigned)l16 * 128u + (((unsigned)(4 + lq)) ^ swz) * 16u;       // kk = 1
    const unsigned a_base = (unsigned)wm * (64u * 128u), b_base = 2u * HALF_B + (unsigned)wn * (32u * 128u);

    f32x4 acc[2][2][4][2];            // [A half][B half][row tile][filter tile]
    f16x8 af[4][2], bf0[2][2], bf1[2][2];

    // prologue: K-tile 0 whole and the first half-tiles of K-tile 1; the first two retired before the first phase
    stage(std::integral_constant<int, 0>{});
    stage(std::integral_constant<int, 1>{});
    stage(std::integral_constant<int, 2>{});
    stage(std::integral_constant<int, 3>{});
    advance();
    asm volatile("s_waitcnt vmcnt(2)" ::: "memory");      // A-h0, B-h0, B-h1 of K-tile 0 are in; A-h1 may still fly
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    if (wm == 1) __builtin_amdgcn_s_barrier();        // waves 4..7 run one barrier behind (without the skew: 20 % slower)
    __builtin_amdgcn_sched_barrier(0);

    // One phase = one A half of the K-tile against both B halves: 32 MFMAs between two barriers (r2 first had four phases
    // of 16 MFMAs per K-tile; halving the barrier round trips per MFMA was worth 5-12 % on the 3x3 layers, while a third
    // fewer LDS-DMA instructions -- a timing ablation -- was worth nothing: profiles/r02_notes.md).
    //   X = 0: fragments of A-h0, B-h0, B-h1 (16 ds_read_b128); DMA of A-h0, B-h0, B-h1 of the NEXT K-tile, whose slots were
    //          last read two phases ago; vmcnt(6) retires A-h1 of this K-tile (issued one phase ago); quadrants (A0,B0) (A0,B1)
    //   X = 1: fragments of A-h1 (8); DMA of A-h1 of the next K-tile, cursor advance; vmcnt(2) retires the six DMAs of the
    //          X = 0 phase, i.e. everything the next phase reads; quadrants (A1,B1) (A1,B0) with the B fragments still in registers
    // Stamps (-DP8_STAMPS): [0] fragment reads issued, [1] DMA issue, [2] wait for DMA data, [3] first barrier, [4] matrix
    // segment, [5] second barrier, [6] epilogue.
    auto phase = [&](auto XC, const unsigned char *buf) {
        constexpr int X = decltype(XC)::value;
        P8_STAMP(5);
        if (X == 0) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                bf0[j][0] = *(const f16x8 *)(buf + b_base + 0 * HALF_B + j * 2048 + fo0);
                bf0[j][1] = *(const f16x8 *)(buf + b_base + 0 * HALF_B + j * 2048 + fo1);
                bf1[j][0] = *(const f16x8 *)(buf + b_base + 1 * HALF_B + j * 2048 + fo0);
                bf1[j][1] = *(const f16x8 *)(buf + b_base + 1 * HALF_B + j * 2048 + fo1);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            af[i][0] = *(const f16x8 *)(buf + X * HALF_B + a_base + i * 2048 + fo0);
            af[i][1] = *(const f16x8 *)(buf + X * HALF_B + a_base + i * 2048 + fo1);
        }
        P8_STAMP(0);
        if (X == 0) {
            stage(std::integral_constant<int, 0>{});
            stage(std::integral_constant<int, 1>{});
            stage(std::integral_constant<int, 2>{});
            P8_STAMP(1);
            asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        } else {
            stage(std::integral_constant<int, 3>{});
            advance();
            P8_STAMP(1);
            asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        }
        P8_STAMP(2);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        P8_STAMP(3);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(1);
        // first the B half whose fragments are older: (A0,B0) (A0,B1) / (A1,B1) (A1,B0)
#pragma unroll
        for (int yy = 0; yy < 2; ++yy) {
            const int Y = X ? 1 - yy : yy;
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[X][Y][i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[i][kk], Y ? bf1[j][kk] : bf0[j][kk], acc[X][Y][i][j], 0, 0, 0);
        }
        __builtin_amdgcn_s_setprio(0);
        P8_STAMP(4);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    };
    typedef std::integral_constant<int, 0> I0;
    typedef std::integral_constant<int, 1> I1;
    typedef std::integral_constant<int, 2> I2;
    typedef std::integral_constant<int, 3> I3;
    typedef std::true_type T_;
    typedef std::false_type F_;

#ifdef P8_STAMPS
    st_prev = __builtin_amdgcn_s_memtime();
    const unsigned long long st_real0 = __builtin_amdgcn_s_memrealtime();     // 100 MHz: in-kernel clock = cycles / (ticks * 10 ns)
#endif
    int cbuf = 0;
    for (int cti = 0;; ++cti) {
        const Item it = item_at(cti);
        const int ct = it.tile;
        if (ct >= a.ntiles) break;
        const int p0 = (ct / a.tiles_n) * BM, n0 = (ct % a.tiles_n) * BN;
#pragma unroll
        for (int x = 0; x < 2; ++x)
#pragma unroll
            for (int y = 0; y < 2; ++y)
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
#pragma unroll
                        for (int r = 0; r < 4; ++r) acc[x][y][i][j][r] = 0.f;
        for (int kt = it.kb; kt < it.ke; ++kt) {
            const unsigned char *buf = p8_smem + cbuf * BUF_B;
            // quadrants (A0,B0) (A0,B1) (A1,B1) (A1,B0); half-tile staged: the one first read D half-tiles later
            phase(I0{}, buf);
            phase(I1{}, buf);
            cbuf ^= 1;
        }

        // ---- stream-K piece: the raw sums leave as they stand, 32 coalesced 16-byte stores per lane (see conv_p8_fixup_kernel) ----
        if (it.slot >= 0) {
            f32x4 *wsp = (f32x4 *)a.ws + (size_t)it.slot * (32 * 512) + t;
#pragma unroll
            for (int x = 0; x < 2; ++x)
#pragma unroll
                for (int y = 0; y < 2; ++y)
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int j = 0; j < 2; ++j) wsp[(size_t)((((x * 2 + y) * 4 + i) * 2 + j)) * 512] = acc[x][y][i][j];
            P8_STAMP(6);
            continue;
        }
        // ---- epilogue (no barrier inside: the wave groups keep their one-barrier skew across tiles) ----
        auto epilogue_pass = [&](auto LEAKYC) {
            const int ACT_ = decltype(LEAKYC)::value ? (int)Y2H_ACT_LEAKY : a.act;
            _Float16 *es = (_Float16 *)(p8_smem + 2 * BUF_B) + wv * 32 * ES;
#pragma unroll
            for (int y = 0; y < 2; ++y) {
                const int cb = n0 + y * 128 + wn * 32;             // 32 filters: two 16-wide MFMA tiles
                const int c0f = cb + l16, c1f = cb + 16 + l16;
                const float al0 = c0f < a.Cout ? a.alpha[c0f] : 0.f, be0 = c0f < a.Cout ? a.beta[c0f] : 0.f;
                const float al1 = c1f < a.Cout ? a.alpha[c1f] : 0.f, be1 = c1f < a.Cout ? a.beta[c1f] : 0.f;
#pragma unroll
                for (int x = 0; x < 2; ++x)
                    p8_epilogue_quadrant(a, ACT_, acc[x][y], p0 + x * 128 + wm * 64, cb, al0, be0, al1, be1, es, lane);
            }
        };
        if (a.act == Y2H_ACT_LEAKY) epilogue_pass(std::true_type{});
        else epilogue_pass(std::false_type{});
        P8_STAMP(6);                                       // epilogue
    }
#ifdef P8_STAMPS
    st_acc[7] = __builtin_amdgcn_s_memrealtime() - st_real0;
    if (lane == 0 && a.stamps)
        for (int k = 0; k < 8; ++k) a.stamps[((size_t)blockIdx.x * 8 + wv) * 8 + k] = st_acc[k];
#endif
    __builtin_amdgcn_sched_barrier(0);
    if (wm == 0) __builtin_amdgcn_s_barrier();        // pairs with the extra barrier waves 4..7 took at the start
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the look-ahead DMAs of the (masked) tiles past the end
}

// ---------------------------------------------------------------------------
// dispatch
// ---------------------------------------------------------------------------
struct VariantH {
    const char *name;
    int bm, bn, bk, ks;
    void (*fn)(ConvK);
    size_t lds;
    int threads;
    int minb;
    bool m16;          // 16x16x32 MFMA variant: needs the 16-byte output stores (vec_store)
    bool p8;           // the LDS-DMA / eight-phase kernel (conv_p8_f16_kernel)
    bool attr_set[16];
};

// MINB = workgroups per CU the register budget is sized for: 4-wave kernels with MINB 1 get the whole
// 512-entry register file of their SIMD (one wave per SIMD), everything else 256 registers per lane
#define VARH(BM, BN, BK, KS, WM, WN, MINB, DB, M16)                                                  \
    { "conv_mfma_f16_" #BM "x" #BN "x" #BK "_k" #KS, BM, BN, BK, KS, conv_mfma_f16_kernel<BM, BN, BK, KS, WM, WN, MINB, DB, M16>, \
      (size_t)2 * (BM + BN) * (BK + 8) * sizeof(_Float16), WM * WN * 64, MINB, M16, false, {false} }
// same tile and name (tests and profiles address kernels by tile), different structure: see conv_p8_f16_kernel
#define VARP8(KS)                                                                                    \
    { "conv_mfma_f16_256x256x64_k" #KS, 256, 256, 64, KS, conv_p8_f16_kernel<KS>,                      \
      (size_t)2 * 4 * 128 * 128 + (size_t)8 * 32 * 40 * sizeof(_Float16), 512, 1, true, true, {false} }

static VariantH g_variants_h[] = {
    // LDS-DMA staging, 8 phases per two K-tiles (first in the table: wins the tie against the register-staged 256x256)
    VARP8(3), VARP8(1),
    // 256x256: eight waves of 128x64 (eight accumulator tiles each), two waves per SIMD, ONE workgroup per CU
    VARH(256, 256, 64, 3, 2, 4, 1, false, true), VARH(256, 256, 64, 1, 2, 4, 1, false, true),
    VARH(256, 256, 32, 3, 2, 4, 1, false, true), VARH(256, 256, 32, 1, 2, 4, 1, false, true),
    VARH(256, 256, 64, 3, 2, 4, 1, false, false), VARH(256, 256, 64, 1, 2, 4, 1, false, false),
    VARH(256, 256, 32, 3, 2, 4, 1, false, false), VARH(256, 256, 32, 1, 2, 4, 1, false, false),
    VARH(256, 128, 64, 3, 4, 2, 1, true, false), VARH(256, 128, 64, 1, 4, 2, 1, true, false),
    VARH(256, 128, 32, 3, 4, 2, 1, true, false), VARH(256, 128, 32, 1, 4, 2, 1, true, false),
    VARH(256, 64, 64, 3, 4, 2, 1, true, false),  VARH(256, 64, 64, 1, 4, 2, 1, true, false),
    VARH(256, 64, 32, 3, 4, 2, 1, true, false),  VARH(256, 64, 32, 1, 4, 2, 1, true, false),
    VARH(128, 128, 64, 3, 2, 2, 2, true, false), VARH(128, 128, 64, 1, 2, 2, 2, true, false),
    VARH(128, 128, 32, 3, 2, 2, 2, true, false), VARH(128, 128, 32, 1, 2, 2, 2, true, false),
    VARH(128, 64, 64, 3, 2, 2, 2, true, false),  VARH(128, 64, 64, 1, 2, 2, 2, true, false),
    VARH(128, 64, 32, 3, 2, 2, 2, true, false),  VARH(128, 64, 32, 1, 2, 2, 2, true, false),
    VARH(64, 64, 64, 3, 2, 2, 2, true, false),   VARH(64, 64, 64, 1, 2, 2, 2, true, false),
    VARH(64, 64, 32, 3, 2, 2, 2, true, false),   VARH(64, 64, 32, 1, 2, 2, 2, true, false),
};

// ---------------------------------------------------------------------------
// 3x3 convolution with 32 input channels (the 32 -> 64 layer of every Darknet-19 trunk), weights stationary.
//
// K = 288 is nine slices of the generic kernel, each with a barrier and a staging round for two MFMA steps of work:
// that layer ran at 340 TFLOP/s (13 % of the darknet19_448 b128 step).  Here nothing is staged inside the K loop:
//   * the filter fragments of this lane -- [filter tile][tap*2 + k-step], 18 x 16 bytes per 32 filters -- are loaded
//     once per kernel and stay in registers (144 VGPRs for 64 filters);
//   * a workgroup walks 16x16-pixel output tiles; the 18x18-pixel input patch of the NEXT tile (64 bytes per pixel) is
//     fetched into registers while the current one multiplies and written to the other LDS buffer afterwards: one
//     barrier per tile of 144 MFMAs per wave;
//   * an MFMA row tile is a 2 x 16 pixel strip in pool-major order (row r = 4*window + corner), read straight from the
//     patch with one ds_read_b128 per tap and k-step, shared by both filter tiles: 0.5 LDS reads per MFMA (the generic
//     kernel needs 0.75).  Pixel pitch 80 B and row pitch 1664 B make those reads bank-conflict free.
// ---------------------------------------------------------------------------
template <int NF, bool POOL, bool LEAKY>
__global__ __launch_bounds__(256, 2) void conv_c32_f16_kernel(ConvK a)
{
    const int ACT_ = LEAKY ? (int)Y2H_ACT_LEAKY : a.act;     // a constant in the common case: no per-value branches
    constexpr int PW = 18, PIX_B = 80, ROW_B = 1664, BUF_B = PW * ROW_B;
    constexpr int NCH = PW * PW * 4;                 // 16-byte chunks of one patch
    constexpr int NP = (NCH + 255) / 256;            // staging passes
    extern __shared__ __attribute__((aligned(16))) unsigned char c32_smem[];
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6, li = lane & 31, lh = lane >> 5;
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void *)a.x, 0, a.xbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc((void *)a.w, 0, a.wbytes, 0x00020000);

    f16x8 bw[NF][18];
    float alpha[NF], beta[NF];
#pragma unroll
    for (int j = 0; j < NF; ++j) {
        const int co = 32 * j + li;
        const bool cok = co < a.Cout;
        alpha[j] = cok ? a.alpha[co] : 0.f;
        beta[j] = cok ? a.beta[co] : 0.f;
#pragma unroll
        for (int kk = 0; kk < 18; ++kk) {
            const unsigned off = cok ? (unsigned)((co * 288 + (kk >> 1) * 32 + (kk & 1) * 16 + 8 * lh) * 2) : a.wbytes;
            bw[j][kk] = __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(wr, off, 0, 0));
        }
    }

    // staging role: chunk q = t + 256*p of the patch -> patch pixel (py, px), 16-byte part
    const int ldxB = a.ldx * 2;
    int s_lds[NP], s_rel[NP], s_yx[NP];
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        const int q = t + 256 * p;
        const int pixel = q >> 2, part = q & 3;
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

    // this lane's pixel inside a 2 x 16 strip: window li>>2, corner li&3
    const int a_off = ((li >> 1) & 1) * ROW_B + (2 * (li >> 2) + (li & 1)) * PIX_B + lh * 16;
    const int Hp = a.H >> 1, Wp = a.W >> 1;
    _Float16 *yh = (_Float16 *)a.y;
    // pooled outputs leave as 16-byte stores: a strip's 8 windows x 64 filters are transposed through a wave-private
    // 1.1 KB scratch behind the two patch buffers (LDS operations of one wave execute in order: no barrier)
    constexpr int ES_B = 144;                        // scratch row: 64 filters x 2 B + 16 B
    unsigned char *es = c32_smem + 2 * BUF_B + wv * 8 * ES_B;

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
            const int rp = 2 * wv + rpi;                       // strip = output rows oy0 + 2rp, +1
            const unsigned char *ap = c32_smem + cur * BUF_B + 2 * rp * ROW_B + a_off;
            f32x16 acc[NF];
#pragma unroll
            for (int j = 0; j < NF; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
#pragma unroll
            for (int kh = 0; kh < 3; ++kh)
#pragma unroll
                for (int kw = 0; kw < 3; ++kw)
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks) {
                        const f16x8 af = *(const f16x8 *)(ap + kh * ROW_B + kw * PIX_B + ks * 32);
#pragma unroll
                        for (int j = 0; j < NF; ++j)
                            acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af, bw[j][(kh * 3 + kw) * 2 + ks], acc[j], 0, 0, 0);
                    }
            if (POOL && a.vec_store) {
#pragma unroll
                for (int j = 0; j < NF; ++j)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        float m = epilogue_fast(acc[j][4 * g], alpha[j], beta[j], ACT_);
#pragma unroll
                        for (int u = 1; u < 4; ++u) m = __builtin_fmaxf(m, epilogue_fast(acc[j][4 * g + u], alpha[j], beta[j], ACT_));
                        *(_Float16 *)(es + (2 * g + lh) * ES_B + (32 * j + li) * 2) = (_Float16)m;
                    }
                const u32x4 v = *(const u32x4 *)(es + (lane >> 3) * ES_B + (lane & 7) * 16);
                const size_t prow = ((size_t)n * Hp + (oy0 >> 1) + rp) * Wp + (ox0 >> 1) + (lane >> 3);
                if ((lane & 7) * 8 < a.Cout) *(u32x4 *)&yh[prow * a.ldy + (lane & 7) * 8] = v;
                continue;
            }
#pragma unroll
            for (int j = 0; j < NF; ++j) {
                const int co = 32 * j + li;
                if (POOL) {
                    const size_t prow = ((size_t)n * Hp + (oy0 >> 1) + rp) * Wp + (ox0 >> 1);
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        float m = epilogue_fast(acc[j][4 * g], alpha[j], beta[j], ACT_);
#pragma unroll
                        for (int u = 1; u < 4; ++u) m = __builtin_fmaxf(m, epilogue_fast(acc[j][4 * g + u], alpha[j], beta[j], ACT_));
                        if (co < a.Cout) yh[(prow + 2 * g + lh) * a.ldy + co] = (_Float16)m;
                    }
                } else {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int rr = (r & 3) + 8 * (r >> 2) + 4 * lh;            // GEMM row = 4*window + corner
                        const int oy = oy0 + 2 * rp + ((rr >> 1) & 1), ox = ox0 + 2 * (rr >> 2) + (rr & 1);
                        if (co < a.Cout)
                            yh[(((size_t)n * a.H + oy) * a.W + ox) * a.ldy + co] = (_Float16)epilogue_fast(acc[j][r], alpha[j], beta[j], ACT_);
                    }
                }
            }
        }
        if (next < ntiles) store_tile(cur ^ 1);
        __syncthreads();
    }
}

static bool c32_ok(const y2h_conv *d)
{
    if (!d->x_f16 || !d->y_f16 || d->x_halo || getenv("Y2_NO_C32")) return false;
    if (d->size != 3 || d->stride != 1 || d->pad != 1 || d->c != 32 || d->n > 64) return false;
    if (d->out_h != d->h || d->out_w != d->w || (d->h & 15) || (d->w & 15) || d->ldx % 8 != 0) return false;
    if (((uintptr_t)d->x | (uintptr_t)d->w_packed) % 16 != 0) return false;
    const double xbytes = (double)d->batch * d->h * d->w * d->ldx * 2.0;
    return xbytes < 2147483000.0 && d->w_packed != nullptr;
}

static int c32_launch(const y2h_conv *d, ConvK &a, y2h_stream s)
{
    if (!d->alpha || !d->beta) return Y2H_EINVAL;
    a.w = d->w_packed;
    a.alpha = d->alpha; a.beta = d->beta;
    a.npix = d->batch * d->h * d->w;
    a.xbytes = (unsigned)((size_t)d->batch * d->h * d->w * d->ldx * 2);
    a.wbytes = (unsigned)((size_t)d->n * 288 * 2);
    const int nf = d->n <= 32 ? 1 : 2;
    const bool lk = d->activation == Y2H_ACT_LEAKY;
    void (*fn)(ConvK) = nf == 1 ? (a.pool ? (lk ? conv_c32_f16_kernel<1, true, true> : conv_c32_f16_kernel<1, true, false>)
                                          : (lk ? conv_c32_f16_kernel<1, false, true> : conv_c32_f16_kernel<1, false, false>))
                                : (a.pool ? (lk ? conv_c32_f16_kernel<2, true, true> : conv_c32_f16_kernel<2, true, false>)
                                          : (lk ? conv_c32_f16_kernel<2, false, true> : conv_c32_f16_kernel<2, false, false>));
    const size_t lds = (size_t)2 * 18 * 1664 + 4 * 8 * 144;
    a.vec_store = d->ldy % 8 == 0 && d->n % 8 == 0 && ((uintptr_t)d->y % 16) == 0 && !getenv("Y2_C32_SCALAR");
    {
        static bool attr_set[16][8] = {{false}};         // per device and instantiation: the attribute call is not free
        const int which = ((nf - 1) * 2 + (a.pool ? 1 : 0)) * 2 + (lk ? 1 : 0);
        int dev = 0;
        Y2H_CHECK(hipGetDevice(&dev));
        if (dev < 0 || dev >= 16 || !attr_set[dev][which]) {
            Y2H_CHECK(hipFuncSetAttribute((const void *)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            if (dev >= 0 && dev < 16) attr_set[dev][which] = true;
        }
    }
    long tiles = (long)d->batch * (d->h >> 4) * (d->w >> 4);
    long grid = tiles < 512 ? tiles : 512;               // two workgroups per CU, tiles are grid-strided
    hipLaunchKernelGGL(fn, dim3((unsigned)grid), dim3(256), lds, S(s), a);
    Y2H_LAUNCH_CHECK();
    return Y2H_OK;
}

// ---------------------------------------------------------------------------
// 3x3 convolution with 64 input channels and <= 128 filters (the 64 -> 128 layers at 112x112 of every Darknet-19 trunk:
// 17 % of the darknet19_448 b128 step at 0.28-0.30 of the fp16 peak on the generic 256x128 tile, whose K loop stages every
// input pixel nine times -- once per tap -- for only 128 filters of work).  Same plan as conv_c32_f16_kernel:
//   * weights stationary in registers: a wave owns ONE 32-filter tile, 36 k-fragments (9 taps x 4 k-steps of 16 channels)
//     = 144 VGPRs, two waves per SIMD; the eight waves of the one workgroup per CU are (filter tile fq, strip half sh):
//     wave (fq, sh) computes filters 32 fq .. +31 of strips 4 sh .. 4 sh + 3.  (A first form with two filter tiles per wave
//     -- 288 registers, one wave per SIMD, half the LDS reads -- ran at 34 % matrix-busy: with nobody to hide them every
//     operand read was exposed, and at the 256-VGPR ceiling the compiler sinks each read next to its MFMAs whatever the
//     source order says; profiles/r03_notes.md);
//   * a workgroup walks 16x16-pixel output tiles; the 18x18-pixel input patch of the NEXT tile (128 B per pixel) is fetched
//     into registers during the current tile's 144 MFMAs per wave and written to the other LDS buffer afterwards: every
//     input pixel reaches the CU once per tile (1.27x with the halo) instead of nine times, one barrier per tile;
//   * an MFMA row tile is a 2 x 16 pixel strip in pool-major order read straight from the patch with one ds_read_b128 per
//     tap and k-step.  Pixel pitch 144 B and row pitch 2688 B (= 128 mod 256) make the 16-lane groups of a ds_read_b128
//     hit sixteen distinct 16-byte slots: conflict free;
//   * outputs leave as 16-byte stores through a wave-private LDS transpose, pooled (8 windows x 32 filters) or not
//     (32 pixels x 32 filters).
// ---------------------------------------------------------------------------
template <bool POOL, bool LEAKY>
__global__ __launch_bounds__(512, 2) void conv_c64_f16_kernel(ConvK a)
{
    const int ACT_ = LEAKY ? (int)Y2H_ACT_LEAKY : a.act;
    constexpr int PW = 18, PIX_B = 144, ROW_B = 2688, BUF_B = PW * ROW_B;
    constexpr int NCH = PW * PW * 8;                 // 16-byte chunks of one patch
    constexpr int NP = (NCH + 511) / 512;            // staging passes
    constexpr int ES_B = 80;                         // scratch row: 32 filters x 2 B + 16 B
    extern __shared__ __attribute__((aligned(16))) unsigned char c64_smem[];
    const int t = threadIdx.x, lane = t & 63, li = lane & 31, lh = lane >> 5;
    const int wv = __builtin_amdgcn_readfirstlane(t >> 6);
    const int fq = wv & 3, sh = wv >> 2;
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void *)a.x, 0, a.xbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc((void *)a.w, 0, a.wbytes, 0x00020000);

    f16x8 bw[36];
    const int co = 32 * fq + li;
    const bool cok = co < a.Cout;
    const float alpha = cok ? a.alpha[co] : 0.f, beta = cok ? a.beta[co] : 0.f;
#pragma unroll
    for (int kk = 0; kk < 36; ++kk) {            // fragment kk = tap * 4 + k-step: K index tap * 64 + ks * 16 + 8 lh .. + 7
        const unsigned off = cok ? (unsigned)((co * 576 + (kk >> 2) * 64 + (kk & 3) * 16 + 8 * lh) * 2) : a.wbytes;
        bw[kk] = __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(wr, off, 0, 0));
    }

    // staging role: chunk q = t + 512 p of the patch -> patch pixel (py, px), 16-byte part
    const int ldxB = a.ldx * 2;
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
            if (s_yx[p] >= 0) *(u32x4 *)(c64_smem + buf * BUF_B + s_lds[p]) = sreg[p];
    };

    // this lane's pixel inside a 2 x 16 strip: window li>>2, corner li&3
    const int a_off = ((li >> 1) & 1) * ROW_B + (2 * (li >> 2) + (li & 1)) * PIX_B + lh * 16;
    const int Hp = a.H >> 1, Wp = a.W >> 1;
    _Float16 *yh = (_Float16 *)a.y;
    unsigned char *es = c64_smem + 2 * BUF_B + wv * 32 * ES_B;          // [32 rows][32 filters + pad], wave-private
    const int cbase = 32 * fq + (lane & 3) * 8;                          // first of the 8 filters this lane stores

    int tile = blockIdx.x, cur = 0;
    if (tile < ntiles) { load_tile(tile); store_tile(0); }
    __syncthreads();
    for (; tile < ntiles; tile += gridDim.x, cur ^= 1) {
        const int next = tile + gridDim.x;
        if (next < ntiles) load_tile(next);
        const int n = tile / tpi, rem = tile - n * tpi;
        const int oy0 = (rem / tiles_x) << 4, ox0 = (rem - (rem / tiles_x) * tiles_x) << 4;
#pragma unroll
        for (int rpi = 0; rpi < 4; ++rpi) {
            const int rp = 4 * sh + rpi;                       // strip = output rows oy0 + 2 rp, + 1
            const unsigned char *ap = c64_smem + cur * BUF_B + 2 * rp * ROW_B + a_off;
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
            for (int kh = 0; kh < 3; ++kh)
#pragma unroll
                for (int kw = 0; kw < 3; ++kw)
#pragma unroll
                    for (int ks = 0; ks < 4; ++ks) {
                        const f16x8 af = *(const f16x8 *)(ap + kh * ROW_B + kw * PIX_B + ks * 32);
                        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(af, bw[(kh * 3 + kw) * 4 + ks], acc, 0, 0, 0);
                    }
            if (POOL) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    float m = epilogue_fast(acc[4 * g], alpha, beta, ACT_);
#pragma unroll
                    for (int u = 1; u < 4; ++u) m = __builtin_fmaxf(m, epilogue_fast(acc[4 * g + u], alpha, beta, ACT_));
                    *(_Float16 *)(es + (2 * g + lh) * ES_B + li * 2) = (_Float16)m;
                }
                const u32x4 v = *(const u32x4 *)(es + (lane >> 2) * ES_B + (lane & 3) * 16);
                const size_t prow = ((size_t)n * Hp + (oy0 >> 1) + rp) * Wp + (ox0 >> 1) + (lane >> 2);
                if (lane < 32 && cbase < a.Cout) *(u32x4 *)&yh[prow * a.ldy + cbase] = v;
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int rr = (r & 3) + 8 * (r >> 2) + 4 * lh;            // GEMM row = 4 * window + corner
                    *(_Float16 *)(es + rr * ES_B + li * 2) = (_Float16)epilogue_fast(acc[r], alpha, beta, ACT_);
                }
#pragma unroll
                for (int p2 = 0; p2 < 2; ++p2) {
                    const int rr = (lane >> 2) + 16 * p2;
                    const u32x4 v = *(const u32x4 *)(es + rr * ES_B + (lane & 3) * 16);
                    const int oy = oy0 + 2 * rp + ((rr >> 1) & 1), ox = ox0 + 2 * (rr >> 2) + (rr & 1);
                    if (cbase < a.Cout) *(u32x4 *)&yh[(((size_t)n * a.H + oy) * a.W + ox) * a.ldy + cbase] = v;
                }
            }
        }
        if (next < ntiles) store_tile(cur ^ 1);
        __syncthreads();
    }
}

static bool c64_ok(const y2h_conv *d)
{
    if (!d->x_f16 || !d->y_f16 || d->x_halo || getenv("Y2_NO_C64")) return false;
    if (d->size != 3 || d->stride != 1 || d->pad != 1 || d->c != 64 || d->n > 128 || d->n <= 64) return false;
    if (d->out_h != d->h || d->out_w != d->w || (d->h & 15) || (d->w & 15) || d->ldx % 8 != 0) return false;
    if (d->ldy % 8 != 0 || d->n % 8 != 0 || ((uintptr_t)d->y % 16) != 0) return false;            // 16-byte output stores only
    if (((uintptr_t)d->x | (uintptr_t)d->w_packed) % 16 != 0) return false;
    {
        long min_tiles = 512;                                  // a plan for big batches: one workgroup per CU, >= 2 tiles each
        if (const char *m = getenv("Y2_C64_MIN_TILES")) min_tiles = atol(m);
        if ((long)d->batch * (d->h >> 4) * (d->w >> 4) < min_tiles) return false;
    }
    const double xbytes = (double)d->batch * d->h * d->w * d->ldx * 2.0;
    return xbytes < 2147483000.0 && d->w_packed != nullptr;
}

static int c64_launch(const y2h_conv *d, ConvK &a, y2h_stream s)
{
    if (!d->alpha || !d->beta) return Y2H_EINVAL;
    a.w = d->w_packed;
    a.alpha = d->alpha; a.beta = d->beta;
    a.npix = d->batch * d->h * d->w;
    a.xbytes = (unsigned)((size_t)d->batch * d->h * d->w * d->ldx * 2);
    a.wbytes = (unsigned)((size_t)d->n * 576 * 2);
    const bool lk = d->activation == Y2H_ACT_LEAKY;
    void (*fn)(ConvK) = a.pool ? (lk ? conv_c64_f16_kernel<true, true> : conv_c64_f16_kernel<true, false>)
                               : (lk ? conv_c64_f16_kernel<false, true> : conv_c64_f16_kernel<false, false>);
    const size_t lds = (size_t)2 * 18 * 2688 + 8 * 32 * 80;
    a.vec_store = 1;
    {
        static bool attr_set[16][4] = {{false}};
        const int which = (a.pool ? 2 : 0) + (lk ? 1 : 0);
        int dev = 0;
        Y2H_CHECK(hipGetDevice(&dev));
        if (dev < 0 || dev >= 16 || !attr_set[dev][which]) {
            Y2H_CHECK(hipFuncSetAttribute((const void *)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            if (dev >= 0 && dev < 16) attr_set[dev][which] = true;
        }
    }
    long tiles = (long)d->batch * (d->h >> 4) * (d->w >> 4);
    long grid = tiles < 256 ? tiles : 256;               // one workgroup per CU, tiles are grid-strided
    if (const char *g = getenv("Y2_CONV_GRID")) { if (atol(g) > 0 && atol(g) < grid) grid = atol(g); }
    hipLaunchKernelGGL(fn, dim3((unsigned)grid), dim3(512), lds, S(s), a);
    Y2H_LAUNCH_CHECK();
    return Y2H_OK;
}

bool y2_f16_conv_ok(const y2h_conv *d)
{
    if (!d->x_f16) return false;
    if (!(d->size == 1 || d->size == 3)) return false;
    if (d->stride != 1 || d->pad != d->size / 2) return false;
    if (d->c % 32 != 0 || d->ldx % 8 != 0) return false;
    if (d->out_h != d->h || d->out_w != d->w || d->x_halo) return false;
    if (((uintptr_t)d->x | (uintptr_t)d->w_packed) % 16 != 0) return false;
    const double xbytes = (double)d->batch * d->h * d->w * d->ldx * 2.0;
    const double wbytes = (double)d->n * d->size * d->size * d->c * 2.0;
    if (xbytes >= 4294967000.0 || wbytes >= 4294967000.0) return false;
    return d->w_packed != nullptr;
}

static int bpc_h(const VariantH &v)
{
    int bpc = (int)(160 * 1024 / v.lds);
    int by_waves = 8 / (v.threads / 64);           // two waves per SIMD (<= 256 VGPRs each)
    if (v.threads == 256 && v.minb == 1) by_waves = 1;   // 512-register kernel: one wave per SIMD
    if (bpc > by_waves) bpc = by_waves;
    return bpc < 1 ? 1 : bpc;
}

// Tile choice.  Per 16-deep MFMA step a CU spends max(matrix time, L2->LDS staging time): BM*BN/128 cycles on
// its four matrix pipes against (BM+BN)*32 bytes at the ~40 B/clk a CU sustains from L2, so small tiles are
// staging bound; the grid quantisation over 256 CUs is counted as in the fp32 picker.  Y2_CONV_TILE forces.
static VariantH *pick_h(const y2h_conv *d)
{
    const int bk = (d->c % 64 == 0) ? 64 : 32;
    const long npix = (long)d->batch * d->h * d->w;
    int force_bm = 0, force_bn = 0;
    if (const char *f = getenv("Y2_CONV_TILE")) sscanf(f, "%dx%d", &force_bm, &force_bn);
    const bool vec_ok = d->y_f16 && d->ldy % 8 == 0 && d->n % 8 == 0 && ((uintptr_t)d->y % 16) == 0;
    const bool no_m16 = getenv("Y2_NO_M16") != nullptr;      // A/B switch: keep to the 32x32x16 variants
    const bool no_p8 = getenv("Y2_F16_NO_P8") != nullptr;    // A/B switch: the register-staged 256x256 kernel
    VariantH *best = nullptr;
    double best_cost = 0;
    for (VariantH &v : g_variants_h) {
        if (v.bk != bk || v.ks != d->size) continue;
        if (force_bm && (v.bm != force_bm || v.bn != force_bn)) continue;
        if (v.m16 && (!vec_ok || no_m16)) continue;
        if (v.p8 && no_p8) continue;
        const long tiles = ((npix + v.bm - 1) / v.bm) * ((d->n + v.bn - 1) / v.bn);
        const int bpc = bpc_h(v);
        long per_cu;
        if (tiles <= 256L * bpc) per_cu = (tiles + 255) / 256;
        else per_cu = (long)bpc * ((tiles + 256L * bpc - 1) / (256L * bpc));
        const double mfma = (double)v.bm * v.bn / 128.0, stage = (double)(v.bm + v.bn) * 0.8;
        const double cost = (double)per_cu * (mfma > stage ? mfma : stage);
        if (!best || cost < best_cost * 0.999 || (cost <= best_cost * 1.001 && v.bm * v.bn > best->bm * best->bn)) {
            best = &v;
            best_cost = cost;
        }
    }
    return best;
}

// ---------------------------------------------------------------------------
// Stream-K plan of the 256x256 LDS-DMA kernel.  A persistent grid of G workgroups walks ntiles tiles in ceil(ntiles / G)
// rounds; the last round holds only R = ntiles mod G tiles (darknet19_448 b128: 392 / 784 / 1568 tiles on 256 CUs = 77 / 77 /
// 88 % occupancy).  With stream-K the R tail tiles are cut along K into equal shares for `wgs` workgroups (pieces of >= MINK
// K-tiles), at the price of the piece traffic (256 KB per piece, written once and read once) and a second launch.
// Estimated in units of one K-tile of one workgroup (~1.7 us): a tile costs nk + C_TILE, a piece its K-tiles + C_TILE,
// the fix-up launch FIX_LAUNCH + pieces * FIX_PIECE.  Env: Y2_SK=0 off, Y2_SK_TILES / Y2_SK_WGS force (tests, sweeps).
// ---------------------------------------------------------------------------
static unsigned long g_sk_launches = 0;
extern "C" unsigned long y2h_stream_k_launches(void) { return g_sk_launches; }

extern "C" int y2h_p8_stream_k_plan(long ntiles, int nk, long grid, int *sk_tiles, int *sk_wgs)
{
    *sk_tiles = 0; *sk_wgs = 0;
    if (ntiles <= 0 || nk <= 0 || grid <= 1) return 0;
    const char *on = getenv("Y2_SK");
    if (on && atoi(on) == 0) return 0;
    const char *ft = getenv("Y2_SK_TILES"), *fw = getenv("Y2_SK_WGS");
    if (ft && atol(ft) > 0) {
        long t = atol(ft), w = fw && atol(fw) > 0 ? atol(fw) : grid;
        if (t > ntiles) t = ntiles;
        if (w > grid) w = grid;
        if (w < t) w = t;                                   // a share never exceeds one tile
        if (w > grid) return 0;
        *sk_tiles = (int)t; *sk_wgs = (int)w;
        return 1;
    }
    const long R = ntiles % grid;
    if (R == 0) return 0;
    const double C_TILE = 5.0, FIX_LAUNCH = 3.0, FIX_PIECE = 0.04;
    int MINK = 4;
    double margin = 0.97;                                   // sweeps: Y2_SK_MARGIN=2 splits whenever a split is possible
    if (const char *m = getenv("Y2_SK_MARGIN")) margin = atof(m);
    if (const char *m = getenv("Y2_SK_MINK")) { if (atoi(m) > 0) MINK = atoi(m); }
    const long rounds = ntiles / grid;
    const double t_plain = (double)(rounds + 1) * (nk + C_TILE);
    long wgs = (R * nk) / MINK;
    if (wgs > grid) wgs = grid;
    if (wgs <= R) return 0;                                 // shares of a whole tile or more: nothing to gain
    const double share = (double)R * nk / wgs;
    const double pieces = (double)wgs + (double)R;          // every share is one piece, plus one more per tile boundary it crosses
    const double t_sk = (double)rounds * (nk + C_TILE) + share + C_TILE * (1.0 + (double)R / wgs) + FIX_LAUNCH + pieces * FIX_PIECE;
    if (t_sk >= margin * t_plain) return 0;
    *sk_tiles = (int)R; *sk_wgs = (int)wgs;
    return 1;
}

// ---------------------------------------------------------------------------
// The other way to finish a partial last round: a second launch that covers the tail tiles' GEMM rows with a SMALLER tile
// of the register-staged kernel (ConvK.row0), so that the rows spread over all 256 CUs again.  No partial sums, no
// workspace: worth it when the tail is a small part of a round (784 tiles = 3 rounds + 16 tiles: those 4096 rows x 512
// filters are 256 tiles of 64x64... one per CU) -- stream-K moves 256 KB per piece and pays a burst of piece stores.  A
// small tile runs at a fraction of the 256x256 kernel's efficiency (REL, measured: profiles/r03_notes.md), so a tail of
// half a round or more is better off with stream-K or as it is.
// Tile numbers are filter-tile fastest, and the whole-tile part is a multiple of the grid (256) and therefore of tiles_n
// (1, 2 or 4): the tail is the row range [row0, npix) x all filters.
// Env: Y2_TAIL=0 off; Y2_TAIL_TILE=BMxBN forces the tail tile (and the tail itself whenever there is a partial round),
// Y2_TAIL_TILES=n the number of 256x256 tiles handed to it (tests).
// ---------------------------------------------------------------------------
static VariantH *find_h(int bm, int bn, int bk, int ks, bool vec_ok)
{
    for (VariantH &v : g_variants_h)
        if (!v.p8 && v.bm == bm && v.bn == bn && v.bk == bk && v.ks == ks && (!v.m16 || vec_ok)) return &v;
    return nullptr;
}

static unsigned long g_tail_launches = 0;
extern "C" unsigned long y2h_tail_launches(void) { return g_tail_launches; }

// estimated time of `rows` GEMM rows x n filters on variant v, in units of one K-tile of the 256x256 kernel
static double tail_cost(const VariantH &v, long rows, int n, int nk)
{
    double rel = 0.25;                                      // efficiency relative to the 256x256 LDS-DMA kernel
    if (v.bm == 256 && v.bn == 128) rel = 0.62;
    else if (v.bm == 256 && v.bn == 64) rel = 0.45;
    else if (v.bm == 128 && v.bn == 128) rel = 0.50;
    else if (v.bm == 128 && v.bn == 64) rel = 0.40;
    else if (v.bm == 64 && v.bn == 64) rel = 0.30;
    const long tiles = ((rows + v.bm - 1) / v.bm) * ((n + v.bn - 1) / v.bn);
    const int bpc = bpc_h(v);
    long per_cu;
    if (tiles <= 256L * bpc) per_cu = (tiles + 255) / 256;
    else per_cu = (long)bpc * ((tiles + 256L * bpc - 1) / (256L * bpc));
    return 1.5 + (double)per_cu * ((double)v.bm * v.bn / 65536.0) / rel * (nk + 6.0);
}

struct P8Plan {
    int sk_tiles, sk_wgs;          // stream-K (0: none)
    int tail_tiles;                // 256x256 tiles handed to the tail launch (0: none)
    VariantH *tail;
};

static P8Plan p8_plan(const y2h_conv *d, long ntiles, int tiles_n, int nk, long grid)
{
    P8Plan pl = {0, 0, 0, nullptr};
    const bool vec_ok = d->y_f16 && d->ldy % 8 == 0 && d->n % 8 == 0 && ((uintptr_t)d->y % 16) == 0;
    const char *toff = getenv("Y2_TAIL");
    const bool tail_on = !(toff && atoi(toff) == 0);
    int fbm = 0, fbn = 0;
    if (const char *f = getenv("Y2_TAIL_TILE")) sscanf(f, "%dx%d", &fbm, &fbn);
    const char *ftn = getenv("Y2_TAIL_TILES");
    const long npix = (long)d->batch * d->h * d->w;
    if (tail_on && fbm && ((ftn && atol(ftn) > 0) || ntiles % grid)) {          // forced
        VariantH *tv = find_h(fbm, fbn, 64, d->size, vec_ok);
        long t = (ftn && atol(ftn) > 0) ? atol(ftn) : ntiles % grid;
        if (t > ntiles) t = ntiles;
        while ((ntiles - t) % tiles_n) ++t;                 // the whole-tile part ends on a row boundary
        if (tv && t <= ntiles) { pl.tail_tiles = (int)t; pl.tail = tv; return pl; }
    }
    if (y2h_p8_stream_k_plan(ntiles, nk, grid, &pl.sk_tiles, &pl.sk_wgs) && (getenv("Y2_SK_TILES") || !tail_on)) return pl;
    const long R = ntiles % grid;
    if (R == 0 || !tail_on || (ntiles - R) % tiles_n) return pl;
    // three candidates for the last round: as it is, stream-K (estimate inside y2h_p8_stream_k_plan, recomputed here), tail
    const double C_TILE = 5.0;
    const double t_round = nk + C_TILE;
    double t_sk = 1e30;
    if (pl.sk_tiles) {
        const double share = (double)R * nk / pl.sk_wgs, pieces = (double)pl.sk_wgs + (double)R;
        t_sk = share + C_TILE * (1.0 + (double)R / pl.sk_wgs) + 3.0 + pieces * 0.04;
    }
    const long rows = npix - ((ntiles - R) / tiles_n) * 256;
    VariantH *best = nullptr;
    double t_tail = 1e30;
    for (VariantH &tv : g_variants_h) {
        if (tv.p8 || tv.bk != 64 || tv.ks != d->size || (tv.m16 && !vec_ok) || (tv.bm == 256 && tv.bn == 256)) continue;
        const double c = tail_cost(tv, rows, d->n, nk);
        if (c < t_tail) { t_tail = c; best = &tv; }
    }
    if (best && t_tail < 0.97 * t_round && t_tail < t_sk) { pl.sk_tiles = pl.sk_wgs = 0; pl.tail_tiles = (int)R; pl.tail = best; }
    return pl;
}

// piece slots the plan may need for this descriptor (bytes of y2h_conv.ws)
size_t y2_f16_conv_workspace_bytes(const y2h_conv *d);

const char *y2_f16_conv_variant(const y2h_conv *d)
{
    if (c32_ok(d)) return "conv_c32_f16_16x16";
    if (c64_ok(d)) return "conv_c64_f16_16x16";
    VariantH *v = y2_f16_conv_ok(d) ? pick_h(d) : nullptr;
    return v ? v->name : nullptr;
}

int y2_f16_conv_launch(const y2h_conv *d, ConvK &a, y2h_stream s)
{
    if (c32_ok(d)) return c32_launch(d, a, s);
    if (c64_ok(d)) return c64_launch(d, a, s);
    VariantH *v = pick_h(d);
    if (!v || !d->alpha || !d->beta) return Y2H_EINVAL;
    a.w = d->w_packed;
    a.alpha = d->alpha; a.beta = d->beta;
    a.npix = d->batch * d->h * d->w;
    a.xbytes = (unsigned)((size_t)d->batch * d->h * d->w * d->ldx * 2);
    a.wbytes = (unsigned)((size_t)d->n * a.K * 2);
    a.tiles_n = (d->n + v->bn - 1) / v->bn;
    a.ksplit = 1;
    if (const char *dbg = getenv("Y2_DBG")) a.dbg = atoi(dbg);
    a.vec_store = d->y_f16 && d->ldy % 8 == 0 && d->n % 8 == 0 && ((uintptr_t)d->y % 16) == 0 && !ABL(8);
    const long tiles_m = ((long)a.npix + v->bm - 1) / v->bm;
    int dev = 0;
    Y2H_CHECK(hipGetDevice(&dev));
    if (dev < 0 || dev >= 16 || !v->attr_set[dev]) {
        Y2H_CHECK(hipFuncSetAttribute((const void *)v->fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)v->lds));
        if (dev >= 0 && dev < 16) v->attr_set[dev] = true;
    }
    a.ntiles = (int)(tiles_m * a.tiles_n);
    if (v->p8 && getenv("Y2_P8_REMAP")) a.dbg |= 64;            // A/B: XCD-contiguous placement of workgroups
    long grid = 256L * bpc_h(*v);
    if (const char *g = getenv("Y2_CONV_GRID")) { if (atol(g) > 0 && atol(g) < grid) grid = atol(g); }   // tests: many tiles per workgroup on small shapes
    bool sk = false;
    ConvK tl;                       // the tail launch, if the plan has one
    VariantH *tailv = nullptr;
    if (v->p8) {
        const P8Plan pl = p8_plan(d, a.ntiles, a.tiles_n, d->size * d->size * (d->c / 64), grid);
        if (pl.sk_tiles && d->ws && d->ws_bytes >= (size_t)2 * pl.sk_wgs * 256 * 256 * sizeof(float)) {
            a.sk_tiles = pl.sk_tiles; a.sk_wgs = pl.sk_wgs; a.ws = d->ws;
            sk = true;
        } else if (pl.tail_tiles) {
            tailv = pl.tail;
            tl = a;
            a.ntiles -= pl.tail_tiles;
            tl.row0 = (a.ntiles / a.tiles_n) * 256;
            tl.tiles_n = (d->n + tailv->bn - 1) / tailv->bn;
            tl.ntiles = (int)((((long)a.npix - tl.row0 + tailv->bm - 1) / tailv->bm) * tl.tiles_n);
        }
    }
    if (!sk && grid > a.ntiles) grid = a.ntiles;
    if (sk) {            // whole tiles need min(grid, ntiles - sk_tiles) workgroups, the shares sk_wgs
        long need = a.ntiles - a.sk_tiles;
        if (need < a.sk_wgs) need = a.sk_wgs;
        if (grid > need) grid = need;
    }
#ifdef P8_STAMPS
    static unsigned long long *d_st = nullptr;
    if (v->p8) {
        if (!d_st) Y2H_CHECK(hipMalloc((void **)&d_st, 256 * 8 * 8 * sizeof(unsigned long long)));
        Y2H_CHECK(hipMemsetAsync(d_st, 0, 256 * 8 * 8 * sizeof(unsigned long long), S(s)));
        a.stamps = d_st;
    }
#endif
    if (grid > 0) {
        hipLaunchKernelGGL(v->fn, dim3((unsigned)grid), dim3(v->threads), v->lds, S(s), a);
        Y2H_LAUNCH_CHECK();
    }
    if (tailv) {
        if (dev < 0 || dev >= 16 || !tailv->attr_set[dev]) {
            Y2H_CHECK(hipFuncSetAttribute((const void *)tailv->fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)tailv->lds));
            if (dev >= 0 && dev < 16) tailv->attr_set[dev] = true;
        }
        long g2 = 256L * bpc_h(*tailv);
        if (g2 > tl.ntiles) g2 = tl.ntiles;
        hipLaunchKernelGGL(tailv->fn, dim3((unsigned)g2), dim3(tailv->threads), tailv->lds, S(s), tl);
        Y2H_LAUNCH_CHECK();
        ++g_tail_launches;
    }
    if (sk) {
        hipLaunchKernelGGL(conv_p8_fixup_kernel, dim3((unsigned)a.sk_tiles * 8), dim3(256), 0, S(s), a, d->size * d->size * (d->c / 64));
        Y2H_LAUNCH_CHECK();
        ++g_sk_launches;
    }
#ifdef P8_STAMPS
    if (v->p8 && getenv("Y2_P8_STAMPS")) {
        static unsigned long long h[256 * 8 * 8];
        Y2H_CHECK(hipStreamSynchronize(S(s)));
        Y2H_CHECK(hipMemcpy(h, d_st, sizeof h, hipMemcpyDeviceToHost));
        const char *names[8] = {"frag reads", "dma issue", "vmcnt wait", "barrier1", "mfma issue", "barrier2", "epilogue", "(realtime)"};
        const int nk = d->size * d->size * (d->c / 64);
        for (int g = 0; g < 2; ++g) {
            double tot[8] = {0}, phases = 0;
            for (long b = 0; b < grid; ++b) {
                const long tiles_b = (a.ntiles - b + grid - 1) / grid;
                for (int w = 4 * g; w < 4 * g + 4; ++w) {
                    phases += 2.0 * nk * tiles_b;
                    for (int k = 0; k < 8; ++k) tot[k] += (double)h[(b * 8 + w) * 8 + k];
                }
            }
            fprintf(stderr, "p8 stamps %dx%d c%d n%d waves %d-%d (shader cycles per phase):", d->h, d->w, d->c, d->n, 4 * g, 4 * g + 3);
            double sum = 0;
            for (int k = 0; k < 7; ++k) { fprintf(stderr, " %s %.0f", names[k], tot[k] / phases); sum += tot[k]; }
            fprintf(stderr, " | total %.0f | in-kernel clock %.3f GHz (cycles / 100 MHz ticks)\n", sum / phases, tot[7] > 0 ? sum / tot[7] * 0.1 : 0.0);
        }
    }
#endif
    return Y2H_OK;
}

size_t y2_f16_conv_workspace_bytes(const y2h_conv *d)
{
    if (c32_ok(d) || c64_ok(d) || !y2_f16_conv_ok(d)) return 0;
    VariantH *v = pick_h(d);
    if (!v || !v->p8) return 0;
    const long npix = (long)d->batch * d->h * d->w;
    const long ntiles = ((npix + 255) / 256) * ((d->n + 255) / 256);
    long grid = 256L * bpc_h(*v);
    if (const char *g = getenv("Y2_CONV_GRID")) { if (atol(g) > 0 && atol(g) < grid) grid = atol(g); }
    const P8Plan pl = p8_plan(d, ntiles, (d->n + 255) / 256, d->size * d->size * (d->c / 64), grid);
    return (size_t)2 * pl.sk_wgs * 256 * 256 * sizeof(float);
}

// ---------------------------------------------------------------------------
// First layer in fp16 mode (3 input channels, 3x3/1 pad 1, <= 64 filters).
//
// The layer is HBM bound (K = 27), so everything is about bytes and instruction count.  The network
// input is converted once to half with FOUR channels per pixel (the fourth is zero) and a one-pixel
// zero halo -- [batch][H+2][W+2][4], 8 bytes per pixel (y2h_nchw_to_nhwc4_halo_f16) -- so that a filter
// tap of a pixel is one aligned 8-byte load and no tap needs a bounds test.  K is laid out as
// k = tap*4 + ci (taps 9..11 and ci = 3 carry zero weights): 48 = three v_mfma_f32_32x32x16_f16 steps
// per 32 pixels instead of the fourteen fp32 32x32x2 steps of conv_first_kernel.  Lane (pixel r, half h)
// of step s holds taps 4s+2h and 4s+2h+1: five 8-byte loads per lane and tile.  The filter fragments
// (converted from the fp32 packed weights) and the folded batch-norm constants live in registers for
// the whole kernel; no LDS.
// ---------------------------------------------------------------------------
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

template <int NT>
__global__ __launch_bounds__(256) void conv_first_f16_kernel(ConvK a)
{
    const int lane = threadIdx.x & 63, li = lane & 31, lh = lane >> 5;
    const long wave = ((long)blockIdx.x * 256 + threadIdx.x) >> 6;
    const long nwaves = (long)gridDim.x * 4;
    const long ntiles = ((long)a.npix + 31) / 32;
    const int W2 = a.W + 2, H2 = a.H + 2, HW = a.H * a.W;
    __shared__ __attribute__((aligned(16))) unsigned char first_es[4 * 8 * 144];
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void *)a.x, 0, a.xbytes, 0x00020000);

    // taps of this lane: step s -> taps 4s+2lh, 4s+2lh+1 (>= 9: none)
    unsigned delta[3][2];
    bool live[3][2];
    f16x8 bw[NT][3];
#pragma unroll
    for (int s = 0; s < 3; ++s)
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int tap = 4 * s + 2 * lh + u;
            live[s][u] = tap < 9;
            const int tt = live[s][u] ? tap : 0;
            const int kh = tt / 3, kw = tt - kh * 3;
            delta[s][u] = (unsigned)((kh * W2 + kw) * 8);                 // 4 halves per pixel
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const int co = j * 32 + li;
#pragma unroll
                for (int ci = 0; ci < 4; ++ci) {
                    float w = 0.f;
                    if (live[s][u] && ci < 3 && co < a.Cout) w = a.w[(size_t)co * 27 + tap * 3 + ci];
                    bw[j][s][u * 4 + ci] = (_Float16)w;
                }
            }
        }
    float alpha[NT], beta[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int co = j * 32 + li;
        alpha[j] = 0.f; beta[j] = 0.f;
        if (co < a.Cout) {
            double al = 1.0, be = a.bias[co];
            if (a.bn) { al = (double)a.scale[co] * a.rinv[co]; be = (double)a.bias[co] - (double)a.mean[co] * al; }
            alpha[j] = (float)al; beta[j] = (float)be;
        }
    }

    // Each wave owns a CONTIGUOUS run of tiles, so the image coordinates of a lane's pixel advance by a
    // constant step from tile to tile (carry logic instead of five integer divisions per tile, which
    // would make this HBM-bound kernel VALU bound).  Unit grid: pooling windows (H/2 x W/2, 8 per tile,
    // lane = window li/4, corner li%4) with the fused pool, pixels (H x W, 32 per tile) without.
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
    auto load_tile = [&](u32x2 (&av)[3][2]) {
        const int py = a.pool ? 2 * cy + ((li >> 1) & 1) : cy, px = a.pool ? 2 * cx + (li & 1) : cx;
        const unsigned base = (unit < nunits) ? ((unsigned)(cn * H2 + py) * (unsigned)W2 + (unsigned)px) * 8u : a.xbytes;
        unit += ustep;
        cx += ustep;
        while (cx >= Wu) { cx -= Wu; if (++cy >= Hu) { cy = 0; ++cn; } }
#pragma unroll
        for (int s = 0; s < 3; ++s)
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                // dead taps read out of range: zeros, no traffic (their weights are zero as well)
                const unsigned off = (live[s][u] && base != a.xbytes) ? base + delta[s][u] : a.xbytes;
                av[s][u] = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(xr, off, 0, 0));
            }
    };
    // The per-output tests of `act` (and of pool / store form) are uniform but real branches; left at run time they cut the
    // epilogue into hundreds of basic blocks (856 s_cbranch in this kernel).  The loop is therefore compiled twice: once
    // with the case that matters -- leaky, fused pool, 16-byte half stores -- as constants, once generic.
    auto run = [&](auto FASTC) {
        constexpr bool FAST = decltype(FASTC)::value;
        const int ACT_ = FAST ? (int)Y2H_ACT_LEAKY : a.act;
        auto compute_tile = [&](long tile, const u32x2 (&av)[3][2]) {
            f32x16 acc[NT];
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
#pragma unroll
            for (int s = 0; s < 3; ++s) {
                u32x4 q;
                q[0] = av[s][0][0]; q[1] = av[s][0][1]; q[2] = av[s][1][0]; q[3] = av[s][1][1];
                const f16x8 af = __builtin_bit_cast(f16x8, q);
#pragma unroll
                for (int j = 0; j < NT; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af, bw[j][s], acc[j], 0, 0, 0);
            }
            const long prow = tile * 32 + 4 * lh;
            _Float16 *yh = (_Float16 *)a.y;
            if (FAST || (a.pool && a.vec_store)) {
                // half outputs as 16-byte stores: the tile's 8 pooled pixels x 32*NT filters go through a wave-private LDS
                // scratch (2-byte stores cost several MFMA times each; LDS operations of one wave execute in order)
                unsigned char *es = first_es + (threadIdx.x >> 6) * 8 * 144;
#pragma unroll
                for (int j = 0; j < NT; ++j)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        float m = epilogue_fast(acc[j][4 * g], alpha[j], beta[j], ACT_);
#pragma unroll
                        for (int t = 1; t < 4; ++t) m = __builtin_fmaxf(m, epilogue_fast(acc[j][4 * g + t], alpha[j], beta[j], ACT_));
                        *(_Float16 *)(es + (2 * g + lh) * 144 + (32 * j + li) * 2) = (_Float16)m;
                    }
                const u32x4 v = *(const u32x4 *)(es + (lane >> 3) * 144 + (lane & 7) * 16);
                const long pr = tile * 8 + (lane >> 3);
                if ((lane & 7) * 8 < a.Cout && pr * 4 < a.npix) *(u32x4 *)&yh[(size_t)pr * a.ldy + (lane & 7) * 8] = v;
                return;
            }
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const int co = j * 32 + li;
                if (a.pool) {
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const long r0 = prow + 8 * g;
                        float m = epilogue_fast(acc[j][4 * g], alpha[j], beta[j], ACT_);
#pragma unroll
                        for (int t = 1; t < 4; ++t) {
                            const float v = epilogue_fast(acc[j][4 * g + t], alpha[j], beta[j], ACT_);
                            m = (v > m) ? v : m;
                        }
                        if (co < a.Cout && r0 < a.npix) {
                            const size_t o = (size_t)(r0 >> 2) * a.ldy + co;
                            if (a.y_f16) yh[o] = (_Float16)m; else a.y[o] = m;
                        }
                    }
                    continue;
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const long p = prow + (r & 3) + 8 * (r >> 2);
                    if (co < a.Cout && p < a.npix) {
                        const float v = epilogue_fast(acc[j][r], alpha[j], beta[j], ACT_);
                        const size_t o = (size_t)p * a.ldy + co;
                        if (a.y_f16) yh[o] = (_Float16)v; else a.y[o] = v;
                    }
                }
            }
        };

        u32x2 a0[3][2], a1[3][2];
        long tile = t_begin;
        if (tile < t_end) load_tile(a0);
        for (; tile < t_end; tile += 2) {
            if (tile + 1 < t_end) load_tile(a1);
            compute_tile(tile, a0);
            if (tile + 2 < t_end) load_tile(a0);
            if (tile + 1 < t_end) compute_tile(tile + 1, a1);
        }
    };
    if (a.act == Y2H_ACT_LEAKY && a.pool && a.vec_store) run(std::true_type{});
    else run(std::false_type{});
}

// ---------------------------------------------------------------------------
// The same layer straight from the network input as the API hands it over: fp32 NCHW planes (x_nchw = 1).
//
// conv_first_f16_kernel needs the half NHWC4 haloed copy, i.e. one more kernel that reads 12 and writes 8 bytes per
// pixel, and then reads every pixel through L1 nine times as 8-byte pieces.  Here a workgroup owns BANDS of eight image
// rows: it reads the band's three fp32 planes (+ one row above and below) with 16-byte loads, coalesced along x,
// converts to half and writes [row][x][4 halves] with a zero halo into LDS -- the layout the MFMA operand wants --
// and its four waves then walk the band's 32-pixel tiles with the operand coming from LDS (ds_read_b64 per tap pair;
// the row stride is padded to 16 mod 32 pixels so the two image rows of a pooled tile fall into different bank halves).
// HBM sees the input once (9/8 with the band halo, which L2 absorbs) and the half output once.  Same K layout, filter
// fragments, epilogue and stores as conv_first_f16_kernel.  Workgroups are persistent (setup once, bands with stride
// gridDim.x); LDS per workgroup = 10 rows x 8 B x stride + the epilogue scratch, so 3-4 workgroups share a CU and one's
// band load overlaps the others' matrix work.
// ---------------------------------------------------------------------------
__host__ __device__ static inline int first_nchw_stride(int W)
{
    const int s = W + 3;                                  // slot 0 unused (keeps pixel 0 16-byte aligned), 1 = left halo, W + 2 = right halo
    return s + ((16 - s % 32) + 32) % 32;                 // == 16 (mod 32) pixels, i.e. 32 (mod 64) LDS banks
}

template <int NT>
__global__ __launch_bounds__(256) void conv_first_nchw_f16_kernel(ConvK a)
{
    constexpr int R = 8;                                  // image rows per band (even: pooling windows never straddle bands)
    extern __shared__ __attribute__((aligned(16))) unsigned char fb_smem[];
    const int t = threadIdx.x, lane = t & 63, li = lane & 31, lh = lane >> 5;
    const int wv = __builtin_amdgcn_readfirstlane(t >> 6);
    const int W = a.W, H = a.H;
    const int stride = first_nchw_stride(W);
    unsigned char *first_es = fb_smem + (size_t)(R + 2) * stride * 8;

    unsigned delta[3][2];
    bool live[3][2];
    f16x8 bw[NT][3];
#pragma unroll
    for (int s = 0; s < 3; ++s)
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int tap = 4 * s + 2 * lh + u;
            live[s][u] = tap < 9;
            const int tt = live[s][u] ? tap : 0;
            const int kh = tt / 3, kw = tt - kh * 3;
            delta[s][u] = (unsigned)((kh * stride + kw) * 8);
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const int co = j * 32 + li;
#pragma unroll
                for (int ci = 0; ci < 4; ++ci) {
                    float w = 0.f;
                    if (live[s][u] && ci < 3 && co < a.Cout) w = a.w[(size_t)co * 27 + tap * 3 + ci];
                    bw[j][s][u * 4 + ci] = (_Float16)w;
                }
            }
        }
    float alpha[NT], beta[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int co = j * 32 + li;
        alpha[j] = 0.f; beta[j] = 0.f;
        if (co < a.Cout) {
            double al = 1.0, be = a.bias[co];
            if (a.bn) { al = (double)a.scale[co] * a.rinv[co]; be = (double)a.bias[co] - (double)a.mean[co] * al; }
            alpha[j] = (float)al; beta[j] = (float)be;
        }
    }

    const int bands_per_img = (H + R - 1) / R;
    const int nbands = a.batch * bands_per_img;
    const int Wu = a.pool ? W >> 1 : W, Hu = a.pool ? H >> 1 : H;
    const int tpr = a.pool ? (Wu + 7) >> 3 : (W + 31) >> 5;           // tiles per row of units
    const size_t plane = (size_t)H * W;
    const int wq = W >> 2;
    _Float16 *yh = (_Float16 *)a.y;

    auto pack2 = [](float lo, float hi) -> unsigned {
        return (unsigned)__builtin_bit_cast(unsigned short, (_Float16)lo) | ((unsigned)__builtin_bit_cast(unsigned short, (_Float16)hi) << 16);
    };

    auto run = [&](auto FASTC) {
        constexpr bool FAST = decltype(FASTC)::value;
        const int ACT_ = FAST ? (int)Y2H_ACT_LEAKY : a.act;
        for (int band = blockIdx.x; band < nbands; band += gridDim.x) {
            const int n = band / bands_per_img, y0 = (band - n * bands_per_img) * R;
            const int rows = (H - y0 < R) ? H - y0 : R;
            __syncthreads();                                          // the previous band's operand reads are done
            // ---- stage: rows y0-1 .. y0+R of the three planes -> half [r][slot][4]; rows outside the image are zeros ----
            const float *img = a.x + (size_t)n * 3 * plane;
#pragma unroll
            for (int rr = 0; rr < 3; ++rr) {
                const int r = wv + 4 * rr;
                if (r < R + 2) {
                    const int y = y0 - 1 + r;
                    const bool yin = y >= 0 && y < H;
                    unsigned char *row = fb_smem + (size_t)r * stride * 8;
                    for (int g0 = 0; g0 < wq; g0 += 128) {
                        f32x4 c[2][3];
#pragma unroll
                        for (int h = 0; h < 2; ++h) {
                            const int g = g0 + 64 * h + lane;
#pragma unroll
                            for (int ch = 0; ch < 3; ++ch) {
                                c[h][ch] = f32x4{0.f, 0.f, 0.f, 0.f};
                                if (yin && g < wq) c[h][ch] = *(const f32x4 *)(img + ch * plane + (size_t)y * W + 4 * g);
                            }
                        }
#pragma unroll
                        for (int h = 0; h < 2; ++h) {
                            const int g = g0 + 64 * h + lane;
                            if (g < wq) {
                                u32x4 v0, v1;
                                v0[0] = pack2(c[h][0][0], c[h][1][0]); v0[1] = pack2(c[h][2][0], 0.f);
                                v0[2] = pack2(c[h][0][1], c[h][1][1]); v0[3] = pack2(c[h][2][1], 0.f);
                                v1[0] = pack2(c[h][0][2], c[h][1][2]); v1[1] = pack2(c[h][2][2], 0.f);
                                v1[2] = pack2(c[h][0][3], c[h][1][3]); v1[3] = pack2(c[h][2][3], 0.f);
                                *(u32x4 *)(row + (size_t)(2 + 4 * g) * 8) = v0;
                                *(u32x4 *)(row + (size_t)(2 + 4 * g) * 8 + 16) = v1;
                            }
                        }
                    }
                    if (lane < 2) *(u32x2 *)(row + (size_t)(lane ? W + 2 : 1) * 8) = u32x2{0u, 0u};
                }
            }
            __syncthreads();

            // ---- tiles of the band: wave wv takes tiles wv, wv + 4, ... (row-major over the band's unit rows) ----
            const int urows = a.pool ? rows >> 1 : rows;
            const int ntile = urows * tpr;
            auto load_tile = [&](int trow, int tcol, u32x2 (&av)[3][2]) {
                const int ux = a.pool ? tcol * 8 + (li >> 2) : tcol * 32 + li;
                const int px = a.pool ? 2 * ux + (li & 1) : ux, pyr = a.pool ? 2 * trow + ((li >> 1) & 1) : trow;
                const unsigned base = ((unsigned)(pyr * stride) + (unsigned)(ux < Wu ? px : 0) + 1u) * 8u;
#pragma unroll
                for (int s = 0; s < 3; ++s)
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        const u32x2 v = *(const u32x2 *)(fb_smem + base + delta[s][u]);
                        av[s][u] = live[s][u] ? v : u32x2{0u, 0u};         // taps 9..11 do not exist
                    }
            };
            auto compute_tile = [&](int trow, int tcol, const u32x2 (&av)[3][2]) {
                f32x16 acc[NT];
#pragma unroll
                for (int j = 0; j < NT; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
#pragma unroll
                for (int s = 0; s < 3; ++s) {
                    u32x4 q;
                    q[0] = av[s][0][0]; q[1] = av[s][0][1]; q[2] = av[s][1][0]; q[3] = av[s][1][1];
                    const f16x8 af = __builtin_bit_cast(f16x8, q);
#pragma unroll
                    for (int j = 0; j < NT; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af, bw[j][s], acc[j], 0, 0, 0);
                }
                // first unit (pooled pixel / pixel) of the tile in the output, and how many of its units exist
                const int u0 = a.pool ? tcol * 8 : tcol * 32;
                const size_t obase = ((size_t)n * Hu + (size_t)((a.pool ? y0 >> 1 : y0) + trow)) * Wu + u0;
                const int nu = Wu - u0;                                    // units of this tile inside the row (>= 1)
                if (FAST || (a.pool && a.vec_store)) {
                    unsigned char *es = first_es + wv * 8 * 144;
#pragma unroll
                    for (int j = 0; j < NT; ++j)
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            float m = epilogue_fast(acc[j][4 * g], alpha[j], beta[j], ACT_);
#pragma unroll
                            for (int tt = 1; tt < 4; ++tt) m = __builtin_fmaxf(m, epilogue_fast(acc[j][4 * g + tt], alpha[j], beta[j], ACT_));
                            *(_Float16 *)(es + (2 * g + lh) * 144 + (32 * j + li) * 2) = (_Float16)m;
                        }
                    const u32x4 v = *(const u32x4 *)(es + (lane >> 3) * 144 + (lane & 7) * 16);
                    if ((lane & 7) * 8 < a.Cout && (lane >> 3) < nu) *(u32x4 *)&yh[(obase + (lane >> 3)) * a.ldy + (lane & 7) * 8] = v;
                    return;
                }
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    const int co = j * 32 + li;
                    if (a.pool) {
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            float m = epilogue_fast(acc[j][4 * g], alpha[j], beta[j], ACT_);
#pragma unroll
                            for (int tt = 1; tt < 4; ++tt) {
                                const float v = epilogue_fast(acc[j][4 * g + tt], alpha[j], beta[j], ACT_);
                                m = (v > m) ? v : m;
                            }
                            const int ui = lh + 2 * g;
                            if (co < a.Cout && ui < nu) yh[(obase + ui) * a.ldy + co] = (_Float16)m;
                        }
                        continue;
                    }
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int ui = 4 * lh + (r & 3) + 8 * (r >> 2);
                        if (co < a.Cout && ui < nu) yh[(obase + ui) * a.ldy + co] = (_Float16)epilogue_fast(acc[j][r], alpha[j], beta[j], ACT_);
                    }
                }
            };
            u32x2 a0[3][2], a1[3][2];
            int lr = wv / tpr, lc = wv - lr * tpr;                        // cursor of the next tile to load
            auto step4 = [&](int &r_, int &c_) { c_ += 4; while (c_ >= tpr) { c_ -= tpr; ++r_; } };
            int ti = wv;
            if (ti < ntile) load_tile(lr, lc, a0);
            for (; ti < ntile; ti += 8) {
                const int r0 = lr, c0 = lc;
                step4(lr, lc);
                if (ti + 4 < ntile) load_tile(lr, lc, a1);
                compute_tile(r0, c0, a0);
                const int r1 = lr, c1 = lc;
                step4(lr, lc);
                if (ti + 8 < ntile) load_tile(lr, lc, a0);
                if (ti + 4 < ntile) compute_tile(r1, c1, a1);
            }
        }
    };
    if (a.act == Y2H_ACT_LEAKY && a.pool && a.vec_store) run(std::true_type{});
    else run(std::false_type{});
}

static size_t first_nchw_lds(int W) { return (size_t)10 * first_nchw_stride(W) * 8 + 4 * 8 * 144; }

// x is the fp32 NCHW network input (x_nchw = 1, x_f16 = 0), the output is half; weights are the fp32 packed [n][27]
bool y2_f16_first_nchw_ok(const y2h_conv *d)
{
    if (!d->x_nchw || d->x_f16 || !d->y_f16 || d->x_halo) return false;
    if (d->c != 3 || d->size != 3 || d->stride != 1 || d->pad != 1 || d->n > 64) return false;
    if (d->out_h != d->h || d->out_w != d->w || (d->w & 3)) return false;
    if (d->fuse_maxpool2 && ((d->h | d->w) & 1)) return false;
    if (first_nchw_lds(d->w) > 64 * 1024 || getenv("Y2_NO_FIRST_NCHW")) return false;
    return d->w_packed != nullptr && ((uintptr_t)d->x % 16) == 0;
}

int y2_f16_first_nchw_launch(const y2h_conv *d, ConvK &a, y2h_stream s)
{
    a.w = d->w_packed;
    a.npix = d->batch * d->h * d->w;
    a.vec_store = d->ldy % 8 == 0 && d->n % 8 == 0 && ((uintptr_t)d->y % 16) == 0 && !getenv("Y2_C32_SCALAR");
    const size_t lds = first_nchw_lds(d->w);
    const long nbands = (long)d->batch * ((d->h + 7) / 8);
    int per_cu = (int)((160 * 1024) / lds);
    if (per_cu > 4) per_cu = 4;
    long blocks = 256L * per_cu;
    if (blocks > nbands) blocks = nbands;
    void (*fn)(ConvK) = d->n <= 32 ? conv_first_nchw_f16_kernel<1> : conv_first_nchw_f16_kernel<2>;
    static bool attr[2];
    if (!attr[d->n > 32]) {
        if (hipFuncSetAttribute((const void *)fn, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024) != hipSuccess) return Y2H_EHIP;
        attr[d->n > 32] = true;
    }
    hipLaunchKernelGGL(fn, dim3((unsigned)blocks), dim3(256), lds, S(s), a);
    Y2H_LAUNCH_CHECK();
    return Y2H_OK;
}

// x is the half NHWC4 haloed input (x_f16 = 1, x_halo = 1, ldx = 4), weights are the fp32 packed [n][27]
bool y2_f16_first_ok(const y2h_conv *d)
{
    if (!d->x_f16 || d->x_halo != 1 || d->ldx != 4) return false;
    if (d->c != 3 || d->size != 3 || d->stride != 1 || d->pad != 1 || d->n > 64) return false;
    if (d->out_h != d->h || d->out_w != d->w) return false;
    const double xbytes = (double)d->batch * (d->h + 2) * (d->w + 2) * 8.0;
    return xbytes < 4294967000.0 && d->w_packed != nullptr && ((uintptr_t)d->x % 8) == 0;
}

int y2_f16_first_launch(const y2h_conv *d, ConvK &a, y2h_stream s)
{
    a.w = d->w_packed;
    a.npix = d->batch * d->h * d->w;
    a.xbytes = (unsigned)((size_t)d->batch * (d->h + 2) * (d->w + 2) * 8);
    const long ntiles = ((long)a.npix + 31) / 32;
    a.vec_store = d->y_f16 && d->ldy % 8 == 0 && d->n % 8 == 0 && ((uintptr_t)d->y % 16) == 0 && !getenv("Y2_C32_SCALAR");
    long blocks = (ntiles + 3) / 4;
    void (*fn)(ConvK) = d->n <= 32 ? conv_first_f16_kernel<1> : conv_first_f16_kernel<2>;
    const long res = resident_blocks((const void *)fn, 256, 0, 3);
    if (blocks > res) blocks = res;                  // tiles are grid-strided
    hipLaunchKernelGGL(fn, dim3((unsigned)blocks), dim3(256), 0, S(s), a);
    Y2H_LAUNCH_CHECK();
    return Y2H_OK;
}
