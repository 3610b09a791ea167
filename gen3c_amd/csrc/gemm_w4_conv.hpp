// Included by gemm.hip behind gemm_w4.hpp (same anonymous namespace).
//
// gemm_bf16_nt_w4_conv_kernel: the tokenizer's CausalConv3d as an implicit GEMM on the ONE-wave-per-SIMD structure of gemm_w4.hpp (128 x 128 per
// wave, all 256 AGPRs accumulators, every operand fragment feeds four MFMAs, one barrier per K tile, LDS-DMA pieces spread behind the MFMAs).
// What changes against the plain GEMM is only where a K tile's operands come from:
//   * K tile t = (tap, channel tile kc), t = tap * (K / 64) + kc. Weight rows: the tap's [N][K] slab - a different wave-uniform base per tile
//     (W + n0 rows + tap * w_tap_stride + kc * 64), the per-lane offsets stay loop invariant.
//   * token rows are GATHERED: tile row r is output position m0 + r = (to, yo, xo); tap (dt, dy, dx) reads input position
//     (max(to st + ot + dt, 0), yo sh + oh + dy, xo sw + ow + dx), or zeros outside the frame. The LDS-DMA piece takes a per-lane 64-bit
//     address (global_load_lds_dwordx4 v[a:a+1], off): 16 address registers per lane, advanced by 128 bytes per channel tile and recomputed
//     when the tap changes (one v_mad_u64_u32 + two selects per piece from per-lane row bases and validity bit masks computed once at kernel
//     start). Padded taps read the library's zero page (g3_zero_page, 8 KiB: K <= 4096 per tap).
// Epilogues: EPI_NONE / EPI_BIAS / EPI_BIAS_RESIDUAL as the plain kernel (same LDS transpose, residual rows requested before the last K tile),
// plus - optional - the GroupNorm statistics of the OUTPUT: per frame (gn_rows consecutive output rows) sum and sum of squares of the stored
// bf16 values, accumulated per wave in fp32 and added to p.gn_stats (double [frames][2]) with one atomic pair per wave and frame: the
// consumer's CausalNormalize (tokenizer/modules/utils.py:58-83) then needs no statistics pass over the tensor.
// K accumulation order per output element is that of the other conv kernels (taps outer, channels inner): bitwise equal outputs (tested).

constexpr int G3_ZERO_PAGE_BYTES = 8192;

template <int EPI>
__global__ __launch_bounds__(GW4_THREADS, 1) void gemm_bf16_nt_w4_conv_kernel(GemmParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31;
    const int g = lane >> 5;

    // XCD-aware tile order (as gemm_bf16_nt_w4_kernel)
    const int nblk = p.tiles_m * p.tiles_n;
    int bid = blockIdx.x;
    {
        const int q = nblk >> 3, r = nblk & 7;
        const int xcd = bid & 7, slot = bid >> 3;
        const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
        bid = base + slot;
    }
    int tile_m, tile_n;
    {
        const int GM = 4;
        const int per_group = GM * p.tiles_n;
        const int grp = bid / per_group;
        const int within = bid - grp * per_group;
        const int gm = min(GM, p.tiles_m - grp * GM);
        tile_n = within / gm;
        tile_m = grp * GM + (within - tile_n * gm);
    }
    const int m0 = tile_m * BM;
    const int n0 = tile_n * BN;
    const int wn = wave & 1;
    const int wm = wave >> 1;

    // ---- weight pieces: per-lane byte offsets from the tile's first weight row (rows clamped to N)
    uint32_t vo_w[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int r = wave * 64 + 8 * q + (lane >> 3);
        const int chunk = (lane & 7) ^ ((r >> 1) & 7);
        const int nrow = min(n0 + r, p.N - 1) - n0;
        vo_w[q] = (uint32_t)((int64_t)nrow * p.ldw * 2 + chunk * 16);
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) asm volatile("" : "+v"(vo_w[q]));
    const char* w_tile = reinterpret_cast<const char*>(p.W + (int64_t)n0 * p.ldw);
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem_raw;
    const uint32_t m0_w = lds0 + (uint32_t)wave * 8192u, m0_t = lds0 + GW4_T_OFF + (uint32_t)wave * 8192u;

    // ---- token pieces: per piece the input coordinates of its output position and the validity mask of the spatial taps
    //   bt[q]    = to st + ot                          (frame of temporal tap 0; tap dt reads frame max(bt + dt, 0), valid while < Ti)
    //   yx[q]    = (yo sh + oh) * Wi + (xo sw + ow)     (may be negative at the top / left border: only ever used for valid taps)
    //   smask[q] = bit (dy * kw + dx): the tap's (yi, xi) lies inside the frame
    int bt[8], yx[8];
    uint32_t smask[8];
    const uint64_t a_base = (uint64_t)(uintptr_t)p.A;
    uint64_t zero_base = (uint64_t)(uintptr_t)g3_zero_page;
    asm volatile("" : "+s"(zero_base));  // resident in scalar registers: re-materialised inside the K loop it is a scalar LOAD (+ lgkmcnt wait) per tap change
    // 16-byte chunk of the 128-byte K tile this lane fetches for piece q: (lane & 7) ^ ((row >> 1) & 7), row = 64 wave + 8 q + (lane >> 3)
    const uint32_t chunk_even = (uint32_t)(((lane & 7) ^ ((lane >> 4) & 7)) * 16), chunk_odd = (uint32_t)(((lane & 7) ^ ((4 + (lane >> 4)) & 7)) * 16);
    {
        // one division pair per lane (piece 0), the other pieces are 8, 16, .. rows further: carried through (xo, yo, to)
        const int hw = p.cv.Ho * p.cv.Wo;
        const int r0 = m0 + wave * 64 + (lane >> 3);
        int to0 = r0 / hw;
        const int rem0 = r0 - to0 * hw;
        int yo0 = rem0 / p.cv.Wo;
        int xo0 = rem0 - yo0 * p.cv.Wo;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int r = wave * 64 + 8 * q + (lane >> 3);
            int to = to0, yo = yo0, xo = xo0;
            if (m0 + r >= p.M) {  // M tail: any valid position (the row is never stored)
                to = p.cv.To - 1; yo = p.cv.Ho - 1; xo = p.cv.Wo - 1;
            }
            const int by = yo * p.cv.sh + p.cv.oh, bx = xo * p.cv.sw + p.cv.ow;
            uint32_t sm = 0;
            for (int dy = 0; dy < p.cv.kh; ++dy)
                for (int dx = 0; dx < p.cv.kw; ++dx) {
                    const int yi = by + dy, xi = bx + dx;
                    if (yi >= 0 && yi < p.cv.Hi && xi >= 0 && xi < p.cv.Wi) sm |= 1u << (dy * p.cv.kw + dx);
                }
            bt[q] = to * p.cv.st + p.cv.ot;
            yx[q] = by * p.cv.Wi + bx;
            smask[q] = sm;
            xo0 += 8;  // next piece: 8 rows on
            while (xo0 >= p.cv.Wo) { xo0 -= p.cv.Wo; ++yo0; }
            while (yo0 >= p.cv.Ho) { yo0 -= p.cv.Ho; ++to0; }
        }
    }
    const int nkc = p.K / BK;
    const uint32_t lda2 = (uint32_t)(p.lda * 2);
    uint64_t ta[8];  // per-lane source address of token piece q for the tile whose pieces are issued next
    // (dt, spatial tap index sp = dy * kw + dx, dyx = dy * Wi + dx) of a tap: wave-uniform
    const int frame_rows = p.cv.Hi * p.cv.Wi;
    auto set_tap = [&](int dt, int sp, int dyx) {
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            int ti = bt[q] + dt;
            ti = ti < 0 ? 0 : ti;  // causal: the first frame is replicated in front
            const bool ok = ((smask[q] >> sp) & 1u) && ti < p.cv.Ti;
            const uint32_t row = (uint32_t)(ti * frame_rows + yx[q] + dyx);  // >= 0 whenever ok
            const uint32_t co = (q & 1) ? chunk_odd : chunk_even;
            const uint64_t a = a_base + (uint64_t)row * lda2 + co;
            ta[q] = ok ? a : zero_base + co;
        }
    };
    auto advance_kc = [&]() {
#pragma unroll
        for (int q = 0; q < 8; ++q) ta[q] += 128;  // next 64 channels (a padded tap walks through the zero page: K * 2 <= 8 KiB)
    };
    // odometer of the NEXT tile whose token pieces get issued: (kc, dt, dy, dx)
    int o_kc = 0, o_dt = 0, o_dy = 0, o_dx = 0;
    auto step_tokens = [&]() {  // ta <- addresses of the following tile; the + 128 bytes of a channel-tile step were applied inside K step 2 (gw4_kstep2_bar_advance)
        if (++o_kc < nkc) return;
        o_kc = 0;
        if (++o_dx == p.cv.kw) {
            o_dx = 0;
            if (++o_dy == p.cv.kh) {
                o_dy = 0;
                ++o_dt;
            }
        }
        set_tap(o_dt, o_dy * p.cv.kw + o_dx, o_dy * p.cv.Wi + o_dx);
    };
    // Tap changes inside the K loop (round 5): where the 32-bit form applies - input rows < 2^24, activation tensor < 4 GiB, which every tokenizer
    // layer satisfies - the next tap's addresses are computed in the MFMA gaps of the barrier K step (gw4_kstep2_bar_settap) instead of by set_tap
    // between two statements: at 128 channels a tap lasts two K tiles, and the ~110 exposed instructions came to a tenth of the K loop.
    const bool tap_in_gaps = p.cv.tap_gaps && (int64_t)p.cv.Ti * frame_rows < (1ll << 24) && (int64_t)p.cv.Ti * frame_rows * lda2 + 128 < (1ll << 32) && lda2 < (1u << 24);
    const uint64_t z_even = zero_base + chunk_even, z_odd = zero_base + chunk_odd;
    const uint32_t zel = (uint32_t)z_even, zeh = (uint32_t)(z_even >> 32), zol = (uint32_t)z_odd, zoh = (uint32_t)(z_odd >> 32);
    const uint32_t a_hi_v = (uint32_t)(a_base >> 32);
    // weight source of a K tile: RUNNING pointers for tile t + 1 and tile t + 2 (+ 128 bytes per channel tile, a jump to the next tap's slab behind a tap's
    // last one) - a handful of scalar instructions per K tile. (Round 5: recomputing base + tap * slab + kc * 128 with 64-bit multiplies for both
    // pointers cost ~37 scalar instructions in front of every K tile's first step: compiler code between the asm statements is exposed at one wave per SIMD.)
    const int64_t w_tap_bytes = p.cv.w_tap_stride * 2;
    const uint32_t w_jump = (uint32_t)(w_tap_bytes - (int64_t)(nkc - 1) * 128);  // last channel tile of a tap -> first of the next (host: slab < 4 GiB)
    const char* w1p = w_tile;  // tile t + 1
    const char* w2p = w_tile;  // tile t + 2
    int w1_kc = 0, w2_kc = 0;
    auto wadvance = [&](const char*& wp, int& kc) {
        const bool wrap = kc + 1 == nkc;
        wp += wrap ? w_jump : 128u;
        kc = wrap ? 0 : kc + 1;
    };

    // ---- fragment read addresses (as gemm_bf16_nt_w4_kernel)
    uint32_t adw[2][4], adt[2][4];
    {
        const uint32_t c0 = (uint32_t)((g ^ ((l31 >> 1) & 7)) << 4);
#pragma unroll
        for (int st = 0; st < 2; ++st)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                adw[st][ks] = ((lds0 + (uint32_t)((wn * 128 + l31) * 128) + c0) ^ (uint32_t)(ks << 5)) + (uint32_t)(st * GW4_STAGE_BYTES);
                adt[st][ks] = ((lds0 + (uint32_t)((wm * 128 + l31) * 128) + c0) ^ (uint32_t)(ks << 5)) + (uint32_t)(st * GW4_STAGE_BYTES);
            }
    }
    if (lds0 & 127u) __builtin_trap();

    const int nk = nkc * p.cv.ntaps;  // >= 2 (host)
    // ---- prologue: tile 0 complete, weight pieces 0..5 of tile 1 in flight, first fragments of tile 0 into buffer 0
    {
        set_tap(0, 0, 0);
#pragma unroll
        for (int q = 0; q < 8; ++q)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(w_tile + vo_w[q]),
                                             (__attribute__((address_space(3))) void*)(uintptr_t)(m0_w + 1024u * q), 16, 0, 0);
#pragma unroll
        for (int q = 0; q < 8; ++q)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(uintptr_t)ta[q],
                                             (__attribute__((address_space(3))) void*)(uintptr_t)(m0_t + 1024u * q), 16, 0, 0);
        wadvance(w1p, w1_kc);  // tile 1
        w2p = w1p; w2_kc = w1_kc;
        wadvance(w2p, w2_kc);  // tile 2
        const char* w1 = w1p;
#pragma unroll
        for (int q = 0; q < 6; ++q)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(w1 + vo_w[q]),
                                             (__attribute__((address_space(3))) void*)(uintptr_t)(m0_w + GW4_STAGE_BYTES + 1024u * q), 16, 0, 0);
        advance_kc();   // (inside the K loop this + 128 happens in K step 2 of the previous tile)
        step_tokens();  // ta: tile 1
        static_for<0, 256>([&](auto rc) { gw4_acc_zero<decltype(rc)::value>(); });
        asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        __syncthreads();
        asm volatile("ds_read_b128 v[192:195], %0\n\tds_read_b128 v[196:199], %0 offset:4096\n\tds_read_b128 v[200:203], %0 offset:8192\n\t"
                     "ds_read_b128 v[204:207], %0 offset:12288\n\tds_read_b128 v[208:211], %1 offset:32768\n\tds_read_b128 v[212:215], %1 offset:36864\n\t"
                     "ds_read_b128 v[216:219], %1 offset:40960\n\tds_read_b128 v[220:223], %1 offset:45056"
                     ::"v"(adw[0][0]), "v"(adt[0][0]) : GW4_OWNED, "memory");
    }

    // K tile t in stage S (see gemm_bf16_nt_w4_kernel for the piece schedule). On entry: ta = token addresses of tile t + 1, (w1_*) = tile t + 1,
    // (w2_*) = tile t + 2; on exit all three moved on by one tile.
    auto ktile = [&](auto sc, auto dma_n_c, auto dma_w_c, auto next_c, int t) {
        constexpr int S = decltype(sc)::value;
        constexpr bool DMA_N = decltype(dma_n_c)::value, DMA_W = decltype(dma_w_c)::value, NEXT = decltype(next_c)::value;
        constexpr int SO = S * GW4_STAGE_BYTES, SN = (S ^ 1) * GW4_STAGE_BYTES;
        G3_JITTER(wave + blockIdx.x, t);
        GW4Pieces p0{}, p1{}, p3{};
        if constexpr (DMA_N) {
            const char* wn1 = w1p;
#pragma unroll
            for (int q = 0; q < 2; ++q) p0.m[q] = m0_w + SN + 1024u * (6 + q), p0.vo[q] = vo_w[6 + q], p0.sb[q] = wn1;
#pragma unroll
            for (int q = 0; q < 3; ++q) p0.m[2 + q] = m0_t + SN + 1024u * q, p0.va[2 + q] = ta[q];
#pragma unroll
            for (int q = 0; q < 5; ++q) p1.m[q] = m0_t + SN + 1024u * (3 + q), p1.va[q] = ta[3 + q];
        }
        if constexpr (DMA_W) {
            const char* wn2 = w2p;
#pragma unroll
            for (int q = 0; q < 6; ++q) p3.m[q] = m0_w + SO + 1024u * q, p3.vo[q] = vo_w[q], p3.sb[q] = wn2;
        }
        gw4_kstep<0, true, DMA_N ? 5 : 0, false, DMA_N ? 1 : 0>(adw[S][1], adt[S][1], p0);
        gw4_kstep<1, true, DMA_N ? 5 : 0, false, DMA_N ? 2 : 0>(adw[S][2], adt[S][2], p1);
        bool tap_done = false;  // this tile's barrier step already produced the addresses of tile t + 2
        if constexpr (NEXT) {
            if constexpr (DMA_N) {  // (K steps 0, 1 above issued this tile's token pieces: ta is free to move on)
                if (tap_in_gaps && t + 2 < nk && o_kc + 1 == nkc) {  // tile t + 2 opens a new tap
                    o_kc = 0;
                    if (++o_dx == p.cv.kw) {
                        o_dx = 0;
                        if (++o_dy == p.cv.kh) {
                            o_dy = 0;
                            ++o_dt;
                        }
                    }
                    GW4Tap tp;
                    tp.dt = o_dt; tp.sp = o_dy * p.cv.kw + o_dx; tp.dyx = o_dy * p.cv.Wi + o_dx; tp.Ti = p.cv.Ti; tp.frame_rows = frame_rows;
                    tp.lda2 = lda2; tp.a_lo = (uint32_t)a_base;
                    uint32_t tl[8], th[8];
                    gw4_kstep2_bar_settap(adw[S][3], adt[S][3], tl, th, bt, yx, smask, chunk_even, chunk_odd, zel, zeh, zol, zoh, a_hi_v, tp);
#pragma unroll
                    for (int q = 0; q < 8; ++q) ta[q] = ((uint64_t)th[q] << 32) | tl[q];
                    tap_done = true;
                } else {
                    gw4_kstep2_bar_advance(adw[S][3], adt[S][3], ta, 128ull);
                }
            } else gw4_kstep<2, true, 0, true>(adw[S][3], adt[S][3], p3);
            gw4_kstep<3, true, DMA_W ? 6 : 0, false>(adw[S ^ 1][0], adt[S ^ 1][0], p3);
        } else {
            gw4_kstep<2, true, 0, false>(adw[S][3], adt[S][3], p3);
            gw4_kstep<3, false, 0, false>(0u, 0u, p3);
        }
        if constexpr (DMA_N) {
            wadvance(w1p, w1_kc);
            if (t + 2 < nk && !tap_done) step_tokens();  // (the odometer must not run past the last tap: set_tap would index outside the masks)
        }
        if constexpr (DMA_W) wadvance(w2p, w2_kc);
    };
    constexpr bool HAS_RES = (EPI == EPI_BIAS_RESIDUAL);
    const int rsub = lane >> 4, c2 = lane & 15;
    bf16x8 rpre[HAS_RES ? 4 : 1][HAS_RES ? 8 : 1];
    // residual rows of token blocks J0..J1-1: blocks 0, 1 are requested before the last K tile (64 VGPRs that are idle by then: nothing else
    // covers their latency at one wave per SIMD), blocks 2, 3 behind the K loop - their latency hides under the epilogue of blocks 0, 1. (All
    // four before the last tile, as the plain kernel does, does not fit beside the gather state: 51 spills.)
    auto prefetch_residual = [&](auto j0c, auto j1c) {
        if constexpr (HAS_RES) {
            const int n = n0 + wn * 128 + 8 * c2;
            static_for<decltype(j0c)::value, decltype(j1c)::value>([&](auto jc) {
                constexpr int J = decltype(jc)::value;
#pragma unroll
                for (int s8 = 0; s8 < 8; ++s8) {
                    const int m = m0 + wm * 128 + 32 * J + 4 * s8 + rsub;
                    rpre[J][s8] = (m < p.M && n < p.N) ? load_bf16x8(p.R + (int64_t)m * p.ldr + n) : zero_bf16x8();
                }
            });
        }
    };
    using I0 = std::integral_constant<int, 0>;
    using I2 = std::integral_constant<int, 2>;
    using I4 = std::integral_constant<int, 4>;
    using T_ = std::integral_constant<bool, true>;
    using F_ = std::integral_constant<bool, false>;
    using S0 = std::integral_constant<int, 0>;
    using S1 = std::integral_constant<int, 1>;
    int t = 0;
    for (; t + 3 < nk; t += 2) {
        ktile(S0{}, T_{}, T_{}, T_{}, t);
        ktile(S1{}, T_{}, T_{}, T_{}, t + 1);
    }
    if (nk - t == 3) {
        ktile(S0{}, T_{}, T_{}, T_{}, t);
        ktile(S1{}, T_{}, F_{}, T_{}, t + 1);
        prefetch_residual(I0{}, I2{});
        ktile(S0{}, F_{}, F_{}, F_{}, t + 2);
    } else {  // 2 tiles left
        ktile(S0{}, T_{}, F_{}, T_{}, t);
        prefetch_residual(I0{}, I2{});
        ktile(S1{}, F_{}, F_{}, F_{}, t + 1);
    }
    prefetch_residual(I2{}, I4{});

    // ---- epilogue (gemm_bf16_nt_w4_kernel's, bias / residual forms) + optional GroupNorm statistics of the stored values
    asm volatile("s_nop 7\n\ts_nop 3" ::: GW4_OWNED);
    __syncthreads();
    {
        char* stage = smem_raw + wave * 16384;
        const int n = n0 + wn * 128 + 8 * c2;
        bf16x8 gv1 = zero_bf16x8();
        if (EPI != EPI_NONE && n < p.N) gv1 = load_bf16x8(p.gate + n);
        // statistics: rows of this wave's quadrant lie in frame f0 or f0 + 1 (host: gn_rows >= 128)
        const int mq0 = m0 + wm * 128;
        const int f0 = p.gn_stats ? mq0 / p.gn_rows : 0;
        const int m_split = (f0 + 1) * p.gn_rows;  // first row of frame f0 + 1
        float gs[2] = {0.f, 0.f}, gq[2] = {0.f, 0.f};
        static_for<0, 4>([&](auto jc) {
            constexpr int J = decltype(jc)::value;
            f32x16 acc[4];
            static_for<0, 64>([&](auto rc) {
                constexpr int R = decltype(rc)::value;
                acc[R >> 4][R & 15] = gw4_acc_read<64 * J + R>();
            });
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int q4 = 0; q4 < 4; ++q4) {
                    f32x4 v;
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = acc[i][4 * q4 + e];
                    *reinterpret_cast<f32x4*>(stage + l31 * 512 + (((8 * i + 2 * q4 + g) ^ l31) << 4)) = v;
                }
#pragma unroll
            for (int s8 = 0; s8 < 8; ++s8) {
                const int row = 4 * s8 + rsub;
                const int m = mq0 + 32 * J + row;
                const f32x4 lo = *reinterpret_cast<const f32x4*>(stage + row * 512 + (((2 * c2) ^ row) << 4));
                const f32x4 hi = *reinterpret_cast<const f32x4*>(stage + row * 512 + (((2 * c2 + 1) ^ row) << 4));
                if (m >= p.M || n >= p.N) continue;
                float v[8];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    v[e] = lo[e];
                    v[4 + e] = hi[e];
                }
                if (EPI == EPI_BIAS) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] += (float)gv1[e];
                } else if (EPI == EPI_BIAS_RESIDUAL) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = (v[e] + (float)gv1[e]) + (float)rpre[J][s8][e];
                }
                bf16x8 o;
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = f32_to_bf16(v[e]);
                store_bf16x8(p.C + (int64_t)m * p.ldc + n, o);
                if (p.gn_stats) {
                    float s = 0.f, q = 0.f;
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const float f = (float)o[e];
                        s += f;
                        q += f * f;
                    }
                    const int b = m >= m_split ? 1 : 0;
                    gs[b] += s;
                    gq[b] += q;
                }
            }
        });
        if (p.gn_stats) {
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                double ds = gs[b], dq = gq[b];
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) {
                    ds += __shfl_xor(ds, o, 64);
                    dq += __shfl_xor(dq, o, 64);
                }
                const int mfirst = b == 0 ? mq0 : m_split;
                if (lane == 0 && mfirst < p.M && (b == 0 || m_split < mq0 + 128)) {
                    atomicAdd(p.gn_stats + 2 * (f0 + b) + 0, ds);
                    atomicAdd(p.gn_stats + 2 * (f0 + b) + 1, dq);
                }
            }
        }
    }
}

template <int EPI>
int launch_w4_conv(const GemmParams& p, hipStream_t stream, const char* what) {
    const size_t smem = 2 * GW4_STAGE_BYTES;
    static bool attr_set[64] = {};
    static std::mutex attr_mu;
    int dev_id = 0;
    if (hipGetDevice(&dev_id) != hipSuccess || dev_id < 0 || dev_id >= 64) return g3_set_error(G3_ERR_LAUNCH, "conv: hipGetDevice failed");
    {
        std::lock_guard<std::mutex> lock(attr_mu);
        if (!attr_set[dev_id]) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_bf16_nt_w4_conv_kernel<EPI>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
            if (e != hipSuccess) return g3_set_error(G3_ERR_LAUNCH, "conv: hipFuncSetAttribute: %s", hipGetErrorString(e));
            attr_set[dev_id] = true;
        }
    }
    hipLaunchKernelGGL((gemm_bf16_nt_w4_conv_kernel<EPI>), dim3(p.tiles_m * p.tiles_n), dim3(GW4_THREADS), smem, stream, p);
    return g3_check_launch(what);
}
