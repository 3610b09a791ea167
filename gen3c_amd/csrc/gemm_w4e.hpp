// Included by gemm.hip behind gemm_w4.hpp (same anonymous namespace).
//
// gemm_bf16_nt_w4e_kernel: the one-wave-per-SIMD block GEMM of gemm_w4.hpp as a PERSISTENT tile loop whose epilogue is DEFERRED: the finished
// tile leaves the accumulators as packed bf16 (the rounding nn.Linear itself applies, attention.py:61-62 / blocks.py:455-471), and its transpose,
// GELU / gate * x + residual arithmetic and stores are issued as single instructions in the MFMA gaps of the NEXT tile's K loop. What stays exposed
// between two K loops is the drain alone (256 v_accvgpr_read + 128 v_cvt_pk, ~0.8 us) instead of the whole epilogue (3.3 / 8 / 13 us per 256 x 256
// tile for the plain / gated-residual / GELU classes, profiles/r2_gemm_w4_ablation.txt): round 3's persistent loop had shown that the dead time
// between two K loops IS the epilogue (profiles/r3_gemm_persistent_ab.txt).
//   * one K stream per workgroup: the last two K tiles of an output tile fetch the first two of the next one (same pieces, other base pointers), the
//     first K step of a tile multiplies into C = 0 (no accumulator clear), so the matrix pipe only stops for the drain;
//   * the instruction streams (which epilogue instruction sits in which MFMA gap) are generated: tools/gen_gemm_w4e.py -> gemm_w4e_gen.hpp, whose
//     head describes the schedule: K tile 0 = preamble (unit 0 back from LDS in row layout), then 8 periods of 4 K tiles, one per 32 x 64 unit;
//   * LDS: the two 64 KiB operand stages + per wave a 4 KiB transpose slice X and a 4 KiB residual slice Y (LDS-DMA destination) = all 160 KiB;
//   * the workgroup's LAST tile has no K loop to ride in: the same instruction sequences run bare (gw4e_*_flush_*), so a tile's values do not depend
//     on its position in the workgroup's tile list.
// Applies to (host: w4e_applies): M, N multiples of 256, K of 128, >= 38 K tiles (34 carry the epilogue), epilogues NONE / GELU / GATED_RESIDUAL with
// gate_rows in {1, 2, 4}, full-line alignment. Everything else runs on gemm_bf16_nt_w4_kernel. Off: g3_set_option("gemm_deferred", 0).
// Arithmetic: bf16(acc) first, then the epilogue in fp32 with the operation order of store_tile_lds (which rounds the same way since round 5): bitwise
// equal outputs across all GEMM kernels (tests/test_kernels_gpu.py).

#include "gemm_w4e_gen.hpp"

constexpr int GW4E_LDS_BYTES = 2 * GW4_STAGE_BYTES + 32768;  // 160 KiB
#ifndef G3_GW4E_PFD
#define G3_GW4E_PFD 4  // L2 prefetch distance in K tiles (A/B builds: -DG3_GW4E_PFD=<n>)
#endif
constexpr int GW4E_MIN_NK = 38;                               // K tile 0 preamble + 32 period tiles + tiles 33 (last store), 34; the gate vectors load at K tile nk - 3 >= 35

// TF (token pieces first): the order in which a K tile's 16 LDS-DMA pieces are requested. false (gemm_w4.hpp's order): weight rows 0..5 two K tiles ahead (K step 3),
// weight rows 6, 7 + token rows one tile ahead (K steps 0, 1); true: the same with the operands' roles swapped - the token panel gets the longer lead. The token
// operand of MLP-down is a 3.7 GB stream from HBM (K = 16 384), its weights come from the L2 / Infinity Cache: host heuristic in launch_w4e (N <= 4096).
template <int EPI, bool TF>
__global__ __launch_bounds__(GW4_THREADS, 1) void gemm_bf16_nt_w4e_kernel(GemmParams p) {
    static_assert(EPI == EPI_NONE || EPI == EPI_GELU || EPI == EPI_GATED_RESIDUAL, "gemm_bf16_nt_w4e_kernel: epilogue class without a deferred form");
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];  // [stage 2][W tile | T tile][X 4 waves x 4 KiB][Y 4 waves x 4 KiB]
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31;
    const int g = lane >> 5;
    const int wn = wave & 1;   // feature half of the block tile
    const int wm = wave >> 1;  // token half
    const int nblk = p.tiles_m * p.tiles_n;
    const int nk = p.K / BK;

    auto tile_origin = [&](int bid, int& m0_out, int& n0_out) {  // XCD-aware order of gemm_bf16_nt_w4_kernel
        {
            const int q = nblk >> 3, r = nblk & 7;
            const int xcd = bid & 7, slot = bid >> 3;
            const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
            bid = base + slot;
        }
        int tile_m, tile_n;
        if (p.tile_order_rowmajor == 1) {
            tile_m = bid / p.tiles_n;
            tile_n = bid - tile_m * p.tiles_n;
        } else {
            const int GM = p.tile_order_rowmajor >= 2 ? p.tile_order_rowmajor : 4;
            const int per_group = GM * p.tiles_n;
            const int grp = bid / per_group;
            const int within = bid - grp * per_group;
            const int gm = min(GM, p.tiles_m - grp * GM);
            tile_n = within / gm;
            tile_m = grp * GM + (within - tile_n * gm);
        }
        m0_out = tile_m * BM;
        n0_out = tile_n * BN;
    };

    // ---- operand LDS-DMA: per-lane offsets from a tile's first row (every tile is full here, so they are the same for all tiles)
    uint32_t vo_w[8], vo_t[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int r = wave * 64 + 8 * q + (lane >> 3);
        const int chunk = (lane & 7) ^ ((r >> 1) & 7);
        vo_w[q] = (uint32_t)((int64_t)r * p.ldw * 2 + chunk * 16);
        vo_t[q] = (uint32_t)((int64_t)r * p.lda * 2 + chunk * 16);
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) asm volatile("" : "+v"(vo_w[q]), "+v"(vo_t[q]));
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem_raw;
    if (lds0 & 127u) __builtin_trap();
    const uint32_t m0_w = lds0 + (uint32_t)wave * 8192u, m0_t = lds0 + GW4_T_OFF + (uint32_t)wave * 8192u;
    // "first" / "second" operand of the piece order (see TF)
    const uint32_t (&vo_f)[8] = TF ? vo_t : vo_w;
    const uint32_t (&vo_s)[8] = TF ? vo_w : vo_t;
    const uint32_t m0_f = TF ? m0_t : m0_w, m0_s = TF ? m0_w : m0_t;
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
    // ---- deferred epilogue: per-lane addresses (tools/gen_gemm_w4e.py: layout_model)
    const uint32_t lds_x = lds0 + 2u * GW4_STAGE_BYTES + (uint32_t)wave * 4096u;
    const uint32_t lds_y = lds0 + 2u * GW4_STAGE_BYTES + 16384u + (uint32_t)wave * 4096u;
    const int rr = lane >> 3, cc = lane & 7;
    GW4EOps E{};  // the fields every statement of the kernel shares
    E.xw = lds_x + (uint32_t)(l31 * 128 + 8 * g + ((l31 & 7) << 4));
    E.xr = lds_x + (uint32_t)(rr * 128 + ((cc ^ rr) << 4));
    E.yb = lds_y + (uint32_t)lane * 16u;
    E.coff = (uint32_t)(((int64_t)rr * p.ldc + 8 * cc) * 2);
    E.roff = (uint32_t)(((int64_t)rr * p.ldr + 8 * cc) * 2);
    E.goff = (uint32_t)(((int64_t)(rr % p.gate_rows) * p.ldg + 8 * cc) * 2);
    E.c0 = 0.3275911f * 0.70710678118654752440f;
    // L2 prefetch (gw4e_*ks2*: one 4-byte load per lane = one 128-byte line per row of the 256-row slice of K tile t + D)
    E.pfo = (uint32_t)((int64_t)(wave * 64 + lane) * p.lda * 2);
    E.pfwo = (uint32_t)((int64_t)(wave * 64 + lane) * p.ldw * 2);

    int L = blockIdx.x;
    int m0, n0;
    tile_origin(L, m0, n0);
    const char* w_tile = reinterpret_cast<const char*>(p.W + (int64_t)n0 * p.ldw);
    const char* t_tile = reinterpret_cast<const char*>(p.A + (int64_t)m0 * p.lda);

    // ---- prologue: K tile 0 complete (stage 0), the first operand's rows 0..5 of K tile 1 (stage 1), first fragments
#pragma unroll
    for (int q = 0; q < 8; ++q)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(w_tile + vo_w[q]),
                                         (__attribute__((address_space(3))) void*)(uintptr_t)(m0_w + 1024u * q), 16, 0, 0);
#pragma unroll
    for (int q = 0; q < 8; ++q)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(t_tile + vo_t[q]),
                                         (__attribute__((address_space(3))) void*)(uintptr_t)(m0_t + 1024u * q), 16, 0, 0);
#pragma unroll
    for (int q = 0; q < 6; ++q)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)((TF ? t_tile : w_tile) + 128 + vo_f[q]),
                                         (__attribute__((address_space(3))) void*)(uintptr_t)(m0_f + GW4_STAGE_BYTES + 1024u * q), 16, 0, 0);
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    __syncthreads();
    asm volatile("ds_read_b128 v[192:195], %0\n\tds_read_b128 v[196:199], %0 offset:4096\n\tds_read_b128 v[200:203], %0 offset:8192\n\t"
                 "ds_read_b128 v[204:207], %0 offset:12288\n\tds_read_b128 v[208:211], %1 offset:32768\n\tds_read_b128 v[212:215], %1 offset:36864\n\t"
                 "ds_read_b128 v[216:219], %1 offset:40960\n\tds_read_b128 v[220:223], %1 offset:45056"
                 ::"v"(adw[0][0]), "v"(adt[0][0]) : GW4E_OWNED, "memory");

    using S0 = std::integral_constant<int, 0>;
    using S1 = std::integral_constant<int, 1>;
    bool carry = false;  // a finished tile's epilogue is waiting in P
    int m0e = 0, n0e = 0;
    for (;;) {
        const int Lnext = L + (int)gridDim.x;
        const bool has_next = Lnext < nblk;
        int m0n = m0, n0n = n0;
        if (has_next) tile_origin(Lnext, m0n, n0n);
        const char* w_next = reinterpret_cast<const char*>(p.W + (int64_t)n0n * p.ldw);
        const char* t_next = reinterpret_cast<const char*>(p.A + (int64_t)m0n * p.lda);
        const char* f_tile = TF ? t_tile : w_tile;
        const char* s_tile = TF ? w_tile : t_tile;
        const char* f_next = TF ? t_next : w_next;
        const char* s_next = TF ? w_next : t_next;
        // (the sources of K tiles t + 1 / t + 2 are given explicitly: inside an output tile they are plain strides of its panels - no per-step select;
        //  only the last two K tiles of an output tile reach into the next one)
        auto kops_src = [&](auto sc, auto ksc, const char* f1, const char* s1, const char* f2) -> GW4EOps {
            constexpr int S = decltype(sc)::value, KS = decltype(ksc)::value;
            constexpr uint32_t SO = S * GW4_STAGE_BYTES, SN = (S ^ 1) * GW4_STAGE_BYTES;
            GW4EOps o = E;
            if constexpr (KS < 3) {
                o.adw = adw[S][KS + 1];
                o.adt = adt[S][KS + 1];
            } else {
                o.adw = adw[S ^ 1][0];
                o.adt = adt[S ^ 1][0];
            }
            if constexpr (KS == 0) {
#pragma unroll
                for (int q = 0; q < 2; ++q) o.m[q] = m0_f + SN + 1024u * (6 + q), o.vo[q] = vo_f[6 + q], o.sb[q] = f1;
#pragma unroll
                for (int q = 0; q < 3; ++q) o.m[2 + q] = m0_s + SN + 1024u * q, o.vo[2 + q] = vo_s[q], o.sb[2 + q] = s1;
            } else if constexpr (KS == 1) {
#pragma unroll
                for (int q = 0; q < 5; ++q) o.m[q] = m0_s + SN + 1024u * (3 + q), o.vo[q] = vo_s[3 + q], o.sb[q] = s1;
            } else if constexpr (KS == 3) {
#pragma unroll
                for (int q = 0; q < 6; ++q) o.m[q] = m0_f + SO + 1024u * q, o.vo[q] = vo_f[q], o.sb[q] = f2;
            }
            return o;
        };
        // the slices the barrier step prefetches into the L2: K tile t + D of this output tile, or of the next one behind its end
        auto with_prefetch = [&](GW4EOps o, int t) -> GW4EOps {
#if G3_AB_GW4E_PF  // (A/B build only: the shipped kernel does not prefetch, and computing the operands would cost ~25 scalar instructions per K tile)
            const int tp = t + G3_GW4E_PFD;
            const bool in_tile = tp < nk;
            const int64_t off = (int64_t)(in_tile ? tp : tp - nk) * 128;
            o.pfb = (in_tile ? t_tile : t_next) + off;
            // leader: of the workgroups an XCD runs side by side (4 token tiles x 8 feature tiles of the XCD-aware order) the one with feature tile % 8 == 0
            const uint32_t la = (uint32_t)__builtin_amdgcn_readfirstlane((((in_tile ? n0 : n0n) / BN) & 7) == 0 ? -1 : 0);
            o.pfxa = ((uint64_t)la << 32) | la;
#else
            (void)t;
#endif
            return o;
        };
        auto kops = [&](auto sc, auto ksc, int t) -> GW4EOps {  // K tile t with t + 2 < nk
            GW4EOps o = kops_src(sc, ksc, f_tile + (int64_t)(t + 1) * 128, s_tile + (int64_t)(t + 1) * 128, f_tile + (int64_t)(t + 2) * 128);
            if constexpr (decltype(ksc)::value == 2) o = with_prefetch(o, t);
            return o;
        };
        using K0 = std::integral_constant<int, 0>;
        using K1 = std::integral_constant<int, 1>;
        using K2 = std::integral_constant<int, 2>;
        using K3 = std::integral_constant<int, 3>;
        auto plain_tile = [&](auto sc, int t) {
            G3_JITTER(wave + blockIdx.x, t);
            gw4e_ks0(kops(sc, K0{}, t));
            gw4e_ks1(kops(sc, K1{}, t));
            gw4e_ks2_bar(kops(sc, K2{}, t));
            gw4e_ks3(kops(sc, K3{}, t));
        };
        auto xwrite = [&](int u) {  // unit u's registers -> X (the only statements that name a unit's registers)
            switch (u) {
                case 0: gw4e_xwrite_0(E); break;
                case 1: gw4e_xwrite_1(E); break;
                case 2: gw4e_xwrite_2(E); break;
                case 3: gw4e_xwrite_3(E); break;
                case 4: gw4e_xwrite_4(E); break;
                case 5: gw4e_xwrite_5(E); break;
                case 6: gw4e_xwrite_6(E); break;
                default: gw4e_xwrite_7(E); break;
            }
        };
        // the finished tile's unit u: first output / residual byte of its chunk 0 (rows 8 ch further per chunk)
        auto unit_c = [&](int u) -> char* {
            return reinterpret_cast<char*>(p.C) + (((int64_t)(m0e + wm * 128 + 32 * (u >> 1))) * p.ldc + (n0e + wn * 128 + 64 * (u & 1))) * 2;
        };
        auto unit_r = [&](int u) -> const char* {
            return reinterpret_cast<const char*>(p.R) + (((int64_t)(m0e + wm * 128 + 32 * (u >> 1))) * p.ldr + (n0e + wn * 128 + 64 * (u & 1))) * 2;
        };
        const int64_t c_chunk = (int64_t)8 * p.ldc * 2, r_chunk = (int64_t)8 * p.ldr * 2;

        int t = 0;
        if (carry) {
            // ---- K tile 0: preamble - unit 0 (written to X behind the drain) comes back in row layout
            gw4e_ks0_init(kops(S0{}, K0{}, 0));
            gw4e_ks1_pre(kops(S0{}, K1{}, 0));
            gw4e_ks2_bar(kops(S0{}, K2{}, 0));
            gw4e_ks3(kops(S0{}, K3{}, 0));
            // ---- 8 periods of 4 K tiles: the arithmetic and stores of unit u, the transposition of unit u + 1
            for (int u = 0; u < 8; ++u) {
                char* cb = unit_c(u);
                char* cb_prev3 = unit_c(u > 0 ? u - 1 : 0) + 3 * c_chunk;
                const char* rb_u = (EPI == EPI_GATED_RESIDUAL) ? unit_r(u) : nullptr;
                const char* rb_n = (EPI == EPI_GATED_RESIDUAL) ? unit_r(min(u + 1, 7)) : nullptr;
                if (EPI == EPI_GATED_RESIDUAL && u > 0) gw4e_gate_swap();
                static_for<0, 4>([&](auto tc) {
                    constexpr int TAU = decltype(tc)::value;
                    using SC = std::integral_constant<int, (1 + TAU) & 1>;
                    const int tt = 1 + 4 * u + TAU;
                    G3_JITTER(wave + blockIdx.x, tt);
                    auto step = [&](auto ksc) {
                        constexpr int KS = decltype(ksc)::value, KAPPA = 4 * TAU + KS;
                        GW4EOps o = kops(SC{}, ksc, tt);
                        if constexpr (KS == 2) {
                            // the barrier step stores the chunk finished one K tile earlier: chunk 3 of the PREVIOUS unit at kappa = 2 (unit 0 has
                            // none: the statement without a store), chunks 0, 1, 2 of this unit after
                            o.cb = TAU == 0 ? cb_prev3 : cb + (int64_t)(TAU - 1) * c_chunk;
                            if constexpr (EPI == EPI_GATED_RESIDUAL) {  // residual pieces (u,2) (u,3) (u+1,0) (u+1,1) -> Y slot = chunk
                                constexpr int SLOT = (TAU + 2) & 3;
                                o.rb = (TAU < 2 ? rb_u : rb_n) + (int64_t)SLOT * r_chunk;
                                o.ym = lds_y + 1024u * SLOT;
                            }
                        }
                        if constexpr (EPI == EPI_NONE) {
                            if constexpr (KAPPA == 0) gw4e_none_k0(o); else if constexpr (KAPPA == 1) gw4e_none_k1(o); else if constexpr (KAPPA == 2) { if (u == 0) gw4e_none_k2f(o); else gw4e_none_k2(o); }
                            else if constexpr (KAPPA == 3) gw4e_none_k3(o); else if constexpr (KAPPA == 4) gw4e_none_k4(o); else if constexpr (KAPPA == 5) gw4e_none_k5(o);
                            else if constexpr (KAPPA == 6) gw4e_none_k6(o); else if constexpr (KAPPA == 7) gw4e_none_k7(o); else if constexpr (KAPPA == 8) gw4e_none_k8(o);
                            else if constexpr (KAPPA == 9) gw4e_none_k9(o); else if constexpr (KAPPA == 10) gw4e_none_k10(o); else if constexpr (KAPPA == 11) gw4e_none_k11(o);
                            else if constexpr (KAPPA == 12) gw4e_none_k12(o); else if constexpr (KAPPA == 13) gw4e_none_k13(o); else if constexpr (KAPPA == 14) gw4e_none_k14(o);
                            else gw4e_none_k15(o);
                        } else if constexpr (EPI == EPI_GELU) {
                            if constexpr (KAPPA == 0) gw4e_gelu_k0(o); else if constexpr (KAPPA == 1) gw4e_gelu_k1(o); else if constexpr (KAPPA == 2) { if (u == 0) gw4e_gelu_k2f(o); else gw4e_gelu_k2(o); }
                            else if constexpr (KAPPA == 3) gw4e_gelu_k3(o); else if constexpr (KAPPA == 4) gw4e_gelu_k4(o); else if constexpr (KAPPA == 5) gw4e_gelu_k5(o);
                            else if constexpr (KAPPA == 6) gw4e_gelu_k6(o); else if constexpr (KAPPA == 7) gw4e_gelu_k7(o); else if constexpr (KAPPA == 8) gw4e_gelu_k8(o);
                            else if constexpr (KAPPA == 9) gw4e_gelu_k9(o); else if constexpr (KAPPA == 10) gw4e_gelu_k10(o); else if constexpr (KAPPA == 11) gw4e_gelu_k11(o);
                            else if constexpr (KAPPA == 12) gw4e_gelu_k12(o); else if constexpr (KAPPA == 13) gw4e_gelu_k13(o); else if constexpr (KAPPA == 14) gw4e_gelu_k14(o);
                            else gw4e_gelu_k15(o);
                        } else {
                            if constexpr (KAPPA == 0) gw4e_gated_k0(o); else if constexpr (KAPPA == 1) gw4e_gated_k1(o); else if constexpr (KAPPA == 2) { if (u == 0) gw4e_gated_k2f(o); else gw4e_gated_k2(o); }
                            else if constexpr (KAPPA == 3) gw4e_gated_k3(o); else if constexpr (KAPPA == 4) gw4e_gated_k4(o); else if constexpr (KAPPA == 5) gw4e_gated_k5(o);
                            else if constexpr (KAPPA == 6) gw4e_gated_k6(o); else if constexpr (KAPPA == 7) gw4e_gated_k7(o); else if constexpr (KAPPA == 8) gw4e_gated_k8(o);
                            else if constexpr (KAPPA == 9) gw4e_gated_k9(o); else if constexpr (KAPPA == 10) gw4e_gated_k10(o); else if constexpr (KAPPA == 11) gw4e_gated_k11(o);
                            else if constexpr (KAPPA == 12) gw4e_gated_k12(o); else if constexpr (KAPPA == 13) gw4e_gated_k13(o); else if constexpr (KAPPA == 14) gw4e_gated_k14(o);
                            else gw4e_gated_k15(o);
                        }
                    };
                    step(K0{});
                    step(K1{});
                    step(K2{});
                    step(K3{});
                    // behind the period's first K tile: X is free (unit u's chunk 3 was read in kappa = 3) -> unit u + 1 goes in
                    if constexpr (TAU == 0)
                        if (u < 7) xwrite(u + 1);
                });
            }
            // ---- K tile 33: its barrier step stores the last unit's chunk 3; K tile 34 brings the plain loop back to an odd tile index
            {
                gw4e_ks0(kops(S1{}, K0{}, 33));
                gw4e_ks1(kops(S1{}, K1{}, 33));
                GW4EOps o = kops(S1{}, K2{}, 33);
                o.cb = unit_c(7) + 3 * c_chunk;
                gw4e_ks2_bar_store3(o);
                gw4e_ks3(kops(S1{}, K3{}, 33));
                plain_tile(S0{}, 34);
            }
            t = 35;
        } else {
            gw4e_ks0_init(kops(S0{}, K0{}, 0));
            gw4e_ks1(kops(S0{}, K1{}, 0));
            gw4e_ks2_bar(kops(S0{}, K2{}, 0));
            gw4e_ks3(kops(S0{}, K3{}, 0));
            t = 1;
        }
        // ---- the rest of the K loop (t is odd here, nk even): pairs of plain tiles up to K tile nk - 4
        for (; t + 3 < nk; t += 2) {
            plain_tile(S1{}, t);
            plain_tile(S0{}, t + 1);
        }
        // K tile nk - 3: + the finished-to-be tile's gate vectors into v[56:63] (the previous tile's epilogue is done with them: t >= 33)
        {
            gw4e_ks0(kops(S1{}, K0{}, t));
            gw4e_ks1(kops(S1{}, K1{}, t));
            gw4e_ks2_bar(kops(S1{}, K2{}, t));
            GW4EOps o = kops(S1{}, K3{}, t);
            if constexpr (EPI == EPI_GATED_RESIDUAL) {
                o.gb0 = reinterpret_cast<const char*>(p.gate) + (int64_t)(n0 + wn * 128) * 2;
                o.gb1 = o.gb0 + 128;
                gw4e_ks3_gate(o);
            } else {
                gw4e_ks3(o);
            }
            ++t;
        }
        if (has_next) {  // K tiles nk - 2, nk - 1 fetch the next output tile's K tiles 0, 1
            const char* fl = f_tile + (int64_t)(nk - 1) * 128;
            const char* sl = s_tile + (int64_t)(nk - 1) * 128;
            gw4e_ks0(kops_src(S0{}, K0{}, fl, sl, f_next));
            gw4e_ks1(kops_src(S0{}, K1{}, fl, sl, f_next));
            gw4e_ks2_bar(with_prefetch(kops_src(S0{}, K2{}, fl, sl, f_next), nk - 2));
            gw4e_ks3(kops_src(S0{}, K3{}, fl, sl, f_next));
            gw4e_ks0(kops_src(S1{}, K0{}, f_next, s_next, f_next + 128));
            gw4e_ks1(kops_src(S1{}, K1{}, f_next, s_next, f_next + 128));
            gw4e_ks2_bar(with_prefetch(kops_src(S1{}, K2{}, f_next, s_next, f_next + 128), nk - 1));
            gw4e_ks3(kops_src(S1{}, K3{}, f_next, s_next, f_next + 128));
        } else {  // the workgroup's last tile: nothing further to fetch
            gw4e_ks0(kops(S0{}, K0{}, t));
            gw4e_ks1(kops(S0{}, K1{}, t));
            gw4e_ks2_bar(kops(S0{}, K2{}, t));
            gw4e_ks3_nodma(kops(S0{}, K3{}, t));
            gw4e_ks0_nodma(kops(S1{}, K0{}, t + 1));
            gw4e_ks1_nodma(kops(S1{}, K1{}, t + 1));
            gw4e_ks2_nobar(kops(S1{}, K2{}, t + 1));
            gw4e_ks3_last(kops(S1{}, K3{}, t + 1));
        }

        // ---- drain (the only dead time between two K loops): accumulators -> P
        if (!(G3_AB_GW4E_ABLATE & 8)) gw4e_drain();  // (bit 8: timing ablation)
        m0e = m0;
        n0e = n0;
        if (!has_next) {
            // ---- flush: the same instruction sequences, bare
            for (int u = 0; u < 8; ++u) {
                if (EPI == EPI_GATED_RESIDUAL && u > 0) gw4e_gate_swap();
                xwrite(u);
                GW4EOps o = E;
                if constexpr (EPI == EPI_GATED_RESIDUAL) {
                    const char* rb = unit_r(u);
#pragma unroll
                    for (int ch = 0; ch < 4; ++ch) {
                        o.rb = rb + (int64_t)ch * r_chunk;
                        o.ym = lds_y + 1024u * ch;
                        gw4e_resid_piece(o);
                    }
                }
                char* cb = unit_c(u);
                if constexpr (EPI == EPI_NONE) {
                    gw4e_none_flush_load(o);
                    o.cb = cb; gw4e_none_flush_c0(o);
                    o.cb = cb + c_chunk; gw4e_none_flush_c1(o);
                    o.cb = cb + 2 * c_chunk; gw4e_none_flush_c2(o);
                    o.cb = cb + 3 * c_chunk; gw4e_none_flush_c3(o);
                } else if constexpr (EPI == EPI_GELU) {
                    gw4e_gelu_flush_load(o);
                    o.cb = cb; gw4e_gelu_flush_c0(o);
                    o.cb = cb + c_chunk; gw4e_gelu_flush_c1(o);
                    o.cb = cb + 2 * c_chunk; gw4e_gelu_flush_c2(o);
                    o.cb = cb + 3 * c_chunk; gw4e_gelu_flush_c3(o);
                } else {
                    gw4e_gated_flush_load(o);
                    o.cb = cb; gw4e_gated_flush_c0(o);
                    o.cb = cb + c_chunk; gw4e_gated_flush_c1(o);
                    o.cb = cb + 2 * c_chunk; gw4e_gated_flush_c2(o);
                    o.cb = cb + 3 * c_chunk; gw4e_gated_flush_c3(o);
                }
            }
            break;
        }
        // ---- hand the finished tile to the next K loop: unit 0 -> X, its first two residual pieces -> Y slots 0, 1
        xwrite(0);
        if constexpr (EPI == EPI_GATED_RESIDUAL) {
            GW4EOps o = E;
            const char* rb = unit_r(0);
            o.rb = rb; o.ym = lds_y;
            gw4e_resid_piece(o);
            o.rb = rb + r_chunk; o.ym = lds_y + 1024u;
            gw4e_resid_piece(o);
        }
        carry = true;
        L = Lnext;
        m0 = m0n;
        n0 = n0n;
        w_tile = w_next;
        t_tile = t_next;
    }
}

// gate_rows in {1, 2, 4}: the gate row of an output row m is m % gate_rows = (8 ch + rr) % gate_rows = rr % gate_rows (tile origins are multiples of 32)
static bool w4e_applies(const GemmParams& p, int epi, int n_cu) {
    const int nk = p.K / BK;
    const int grid = (n_cu / 8) * 8;
    if (!g3_opt_gemm_deferred || grid < 8) return false;
    if ((p.M % BM) || (p.N % BN) || (p.K % (2 * BK)) || nk < GW4E_MIN_NK || !p.wide_store) return false;
    if (p.tiles_m * p.tiles_n < 2 * grid) return false;  // a workgroup with one tile has nothing to defer into
    if (epi == EPI_GATED_RESIDUAL && !(p.gate_rows == 1 || p.gate_rows == 2 || p.gate_rows == 4)) return false;
    if ((int64_t)8 * p.ldc * 2 >= (1ll << 31) || (int64_t)8 * p.ldr * 2 >= (1ll << 31) || (int64_t)4 * p.ldg * 2 >= (1ll << 31)) return false;
    return true;
}

template <int EPI, bool TF>
int launch_w4e_tf(const GemmParams& p, hipStream_t stream, const char* what, int n_cu) {
    static bool attr_set[64] = {};
    static std::mutex attr_mu;
    int dev_id = 0;
    if (hipGetDevice(&dev_id) != hipSuccess || dev_id < 0 || dev_id >= 64) return g3_set_error(G3_ERR_LAUNCH, "gemm: hipGetDevice failed");
    {
        std::lock_guard<std::mutex> lock(attr_mu);
        if (!attr_set[dev_id]) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_bf16_nt_w4e_kernel<EPI, TF>), hipFuncAttributeMaxDynamicSharedMemorySize, GW4E_LDS_BYTES);
            if (e != hipSuccess) return g3_set_error(G3_ERR_LAUNCH, "gemm: hipFuncSetAttribute(w4e): %s", hipGetErrorString(e));
            attr_set[dev_id] = true;
        }
    }
    int grid = (n_cu / 8) * 8;
    // (tests: g3_set_option("gemm_deferred_grid", g) runs the tile loop on g workgroups - any multiple of 8 that leaves every workgroup a tile - to walk long and
    // uneven tile lists; 0 = one workgroup per CU)
    if (g3_opt_gemm_deferred_grid >= 8 && (g3_opt_gemm_deferred_grid % 8) == 0 && g3_opt_gemm_deferred_grid <= p.tiles_m * p.tiles_n) grid = g3_opt_gemm_deferred_grid;
    hipLaunchKernelGGL((gemm_bf16_nt_w4e_kernel<EPI, TF>), dim3(grid), dim3(GW4_THREADS), GW4E_LDS_BYTES, stream, p);
    return g3_check_launch(what);
}

// piece order by shape: g3_set_option("gemm_tokens_first", 0 / 1 / 2): never / always / (default) where few feature tiles share a token panel (N <= 4096: its
// rows are fetched from HBM by the first of at most 16 workgroups and everybody waits for them) - measured (profiles/r5_gemm_tokens_first_ab.txt): out-projection +1 %,
// MLP-down +1.5 %; QKV (N = 12 288) -3.5 %, MLP-up (N = 16 384) -1 % -> those keep the weights-first order
template <int EPI>
int launch_w4e(const GemmParams& p, hipStream_t stream, const char* what, int n_cu) {
    const bool tf = g3_opt_gemm_tokens_first == 1 || (g3_opt_gemm_tokens_first == 2 && p.N <= 4096);
    return tf ? launch_w4e_tf<EPI, true>(p, stream, what, n_cu) : launch_w4e_tf<EPI, false>(p, stream, what, n_cu);
}
