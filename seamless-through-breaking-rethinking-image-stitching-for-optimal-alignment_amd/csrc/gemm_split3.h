// Exact-split fp32 contraction on the bf16 matrix cores ("split3"), included by gemm.hip (it shares the epilogue).
//
//   x = hi + mid + lo EXACTLY, three bf16 (8 + 8 + 8 significand bits cover fp32's 24; st_split3 below), so
//   a . b = hi.hi + hi.mid + mid.hi + mid.mid + hi.lo + lo.hi  (+ three dropped terms <= 2^-23 |a||b| together):
//   six v_mfma_f32_32x32x16_bf16 (exact bf16 products, fp32 accumulation) = 6 x 32 matrix cycles per 16 k against
//   8 x 64 for v_mfma_f32_32x32x2_f32: 2.67x fewer.  Round 1 split the operands in registers inside the K loop and was
//   VALU-bound (e270afa, 9944b5e); here NOTHING is split in the loop:
//     - both operands arrive as three bf16 PLANES, blocked by 32-channel chunks: plane p = [C/32][rows][32] bf16, so the
//       32-deep K step of 16 consecutive rows is ONE contiguous KiB -- one LDS-DMA instruction; weights are split once
//       at pack time, activations by the epilogue of the kernel that produced them (or st_split3_pack);
//     - 8 waves: waves 4-7 are LOADERS (nothing but buffer_load ... lds into a 3-stage ring + a counted s_waitcnt +
//       one s_barrier per K step), waves 0-3 CONSUMERS (ds_read_b128 + MFMA only: per 16 k and 64x64 wave tile twelve
//       fragment reads feed 24 MFMAs).  A loader's DMA issue stalls (60-185 cycles per KiB piece, MI355X_MICROARCH.md)
//       therefore never sit between two MFMAs, which at 32 cycles each could not hide them.
//   LDS image of a stage: [plane][A rows | B rows][64 B], 16-B slot c of row r holds k-chunk c ^ ((r >> 2) & 3): the DMA
//   still writes 1 KiB contiguously and every ds_read_b128 of a 16-lane group covers all 64 banks.
//   Accumulator layout = that of the fp32 32x32 MFMA, so gemm_epilogue_* is shared unchanged; the K chain is cut every
//   256 k exactly like the fp32 kernels'.
//
// Matches: /root/reference/core/FlowFormer/PerCostFormer3/gru.py:44-59,246-254 (SepConvGRU, motion encoder convs),
// gma.py:102-115 (aggregate), encoder.py:359-369 (all-pairs correlation).
#pragma once


// fp32 [rows, ldx] (C columns, C % 32 == 0) -> three blocked planes.  One thread = 8 channels of one row.
__global__ __launch_bounds__(256) void split3_pack_kernel(const float* __restrict__ x, __bf16* __restrict__ planes, long long rows, int C,
                                                          long long ldx, long long plane_stride, long long chunk_rows) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    const int groups = C >> 3;
    if (idx >= rows * groups) return;
    // consecutive threads walk the 4 groups of a chunk row, then the rows of the chunk: 64-B runs on both sides
    const int g4 = (int)(idx & 3);
    const long long t = idx >> 2;
    const long long row = t % rows;
    const int chunk = (int)(t / rows);
    const float4 v0 = *reinterpret_cast<const float4*>(x + row * ldx + chunk * 32 + g4 * 8);
    const float4 v1 = *reinterpret_cast<const float4*>(x + row * ldx + chunk * 32 + g4 * 8 + 4);
    const float v[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
    bf16x8 h, m, l;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        __bf16 a, b, c;
        st_split3(v[e], a, b, c);
        h[e] = a; m[e] = b; l[e] = c;
    }
    const long long o = ((long long)chunk * chunk_rows + row) * 32 + g4 * 8;
    *reinterpret_cast<bf16x8*>(planes + o) = h;
    *reinterpret_cast<bf16x8*>(planes + plane_stride + o) = m;
    *reinterpret_cast<bf16x8*>(planes + 2 * plane_stride + o) = l;
}

// DIAG (timing experiments only, results are garbage): 1 = consumers skip the MFMAs (ingest alone), 2 = loaders skip the DMA (reads + MFMAs alone)
// KPAR = 2: EIGHT consumer waves (768-thread workgroups, three waves per SIMD): waves 0-3 take the first 16-k half of every 32-k step, waves 4-7 the
// second half, on the same 32 x 32 (x TM x TN) sub-tiles; the second group's accumulators cross LDS once, behind the K loop, and the first group
// runs the epilogue.  A single consumer wave per SIMD issues one v_mfma_f32_32x32x16_bf16 per 32 cycles at best (tools/probes/mfma_bf16_chain.hip)
// and, with 128 x 64 tiles at one workgroup per CU, that is what bounds the loop (MFMA-only 35 us against DMA-only 29 us on the 1x5 conv); the
// decoder's M = 8 192 shapes have no second tile to give the CU, so the K dimension is what two waves of a SIMD can share.
template <int WM, int WN, int TM, int TN, int STAGES, int DIAG = 0, int KPAR = 1>
__device__ __forceinline__ void conv_gemm_split3_body(const st_gemm_desc& d, const int block_id) {
    static_assert(WM * WN == 4, "4 consumer waves per K half (+ 4 loader waves)");
    static_assert(KPAR == 1 || KPAR == 2, "one or two consumer groups");
    constexpr int BM = WM * TM * 32, BN = WN * TN * 32, ROWS = BM + BN;
    constexpr int PLANE_B = ROWS * 64, STAGE_B = 3 * PLANE_B;       // bytes
    constexpr int GA = BM / 16, GB = BN / 16;                       // 16-row groups (1-KiB pieces) per plane
    static_assert((3 * GA) % 4 == 0 && (3 * GB) % 4 == 0, "pieces divide among the 4 loader waves");
    constexpr int PA = 3 * GA / 4, PB = 3 * GB / 4, PPW = PA + PB;  // pieces per loader wave and K step
    static_assert((STAGES - 1) * PPW <= 63 && STAGES >= 3 && STAGES <= 4, "vmcnt immediate");
    extern __shared__ __attribute__((aligned(1024))) float smem[];
    char* const sm = reinterpret_cast<char*>(smem);

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int split = d.split_k > 1 ? d.split_k : 1;
    const int bz = split > 1 ? 0 : blockIdx.z;
    const int kz = split > 1 ? blockIdx.z : 0;

    const int ntm = (d.M + BM - 1) / BM, ntn = (d.N + BN - 1) / BN;
    const int nwg = ntm * ntn;
    int bid = block_id;
    {
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int tile_n = bid % ntn, tile_m = bid / ntn;
    const int m0 = tile_m * BM, n0 = tile_n * BN;

    const int nkt_all = d.K / 32;
    const int per = (nkt_all + split - 1) / split;
    const int kt0 = kz * per;
    const int ntiles = min(nkt_all, kt0 + per) - kt0;       // K steps of this workgroup (> 0: the host never over-splits)

    unsigned long long sr_entry = 0;
    if (DIAG == 4) sr_entry = __builtin_amdgcn_s_memrealtime();
    if (wave >= 4 * KPAR) {
        // ------------------------------------------------------------------ loader waves
        const int lw = wave - 4 * KPAR;
        const char* A = reinterpret_cast<const char*>(d.a) + (size_t)bz * d.batch_stride_a * 2;
        const char* A2 = d.a2 ? reinterpret_cast<const char*>(d.a2) + (size_t)bz * d.batch_stride_a * 2 : A;
        const char* Wt = reinterpret_cast<const char*>(d.w) + (size_t)bz * d.batch_stride_w * 2;
        const i32x4 rsrcA = make_rsrc(A, d.a_bytes), rsrcA2 = make_rsrc(A2, d.a_bytes), rsrcW = make_rsrc(Wt, d.w_bytes);
        const int a2c = d.a2 ? d.a2_channels : 0;
        const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)sm;
        const unsigned a_plane_b = (unsigned)(d.a_plane_stride * 2), w_plane_b = (unsigned)(d.w_plane_stride * 2);
        const unsigned a_chunk_b = (unsigned)(d.a_rows * 64), w_chunk_b = (unsigned)(d.w_rows * 64);

        int a_row[PA], a_iy0[PA], a_ix0[PA];
        unsigned a_c[PA], voffA[PA], voffB[PB], ldsA[PA], ldsB[PB];
#pragma unroll
        for (int i = 0; i < PA; ++i) {
            const int q = lw * PA + i, plane = q / GA, rg = q % GA;
            const int r = 16 * rg + (lane >> 2), c = (lane & 3) ^ ((r >> 2) & 3);
            const int m = min(m0 + r, d.M - 1);                                  // rows past M: any valid row (never stored)
            const int hw = d.Ho * d.Wo;
            const int b = m / hw, rr = m - b * hw;
            const int oy = rr / d.Wo, ox = rr - oy * d.Wo;
            a_iy0[i] = oy * d.sh - d.ph; a_ix0[i] = ox * d.sw - d.pw;
            a_row[i] = (b * d.H + a_iy0[i]) * d.W + a_ix0[i];
            a_c[i] = (unsigned)plane * a_plane_b + (unsigned)c * 16u;
            ldsA[i] = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)(plane * PLANE_B + rg * 1024));
        }
#pragma unroll
        for (int i = 0; i < PB; ++i) {
            const int q = lw * PB + i, plane = q / GB, rg = q % GB;
            const int r = 16 * rg + (lane >> 2), c = (lane & 3) ^ ((r >> 2) & 3);
            voffB[i] = (unsigned)plane * w_plane_b + (unsigned)min(n0 + r, d.N - 1) * 64u + (unsigned)c * 16u;
            ldsB[i] = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)(plane * PLANE_B + BM * 64 + rg * 1024));
        }
        // (K steps ordered (tap, chunk) like the fp32 kernels.  The other order -- a chunk of A re-read for all its taps back to back, hoping for
        // L1 / L2 hits -- was measured and is SLOWER: 1x5 conv 41.4 -> 48.4 us, 3x3 28.3 -> 34.3 us, tools/split3_probe.py of round 6.)
        int i_c0, i_ky, i_kx;
        {
            const int k = kt0 * 32, tap = k / d.Cin;
            i_c0 = k - tap * d.Cin; i_ky = tap / d.kw; i_kx = tap - i_ky * d.kw;
        }
        unsigned soffA = (unsigned)(i_c0 >> 5) * a_chunk_b, soffB = (unsigned)kt0 * w_chunk_b;
        auto set_tap = [&]() {
            const int ty = i_ky * (d.dh > 1 ? d.dh : 1), tx = i_kx * (d.dw > 1 ? d.dw : 1);
            const int tapoff = ty * d.W + tx;
#pragma unroll
            for (int i = 0; i < PA; ++i) {
                const bool ok = (unsigned)(a_iy0[i] + ty) < (unsigned)d.H && (unsigned)(a_ix0[i] + tx) < (unsigned)d.W;
                voffA[i] = ok ? a_c[i] + (unsigned)(a_row[i] + tapoff) * 64u : ST_OOB;
            }
        };
        set_tap();
        auto issue_tile = [&](int stage) {
            const unsigned so = (unsigned)(stage * STAGE_B);
            const bool second = i_c0 < a2c;
#pragma unroll
            for (int i = 0; i < PA; ++i) if (DIAG != 2 && DIAG != 3) lds_dma16(second ? rsrcA2 : rsrcA, ldsA[i] + so, voffA[i], soffA);
#pragma unroll
            for (int i = 0; i < PB; ++i) if (DIAG != 2 && DIAG != 3) lds_dma16(rsrcW, ldsB[i] + so, voffB[i], soffB);
            soffB += w_chunk_b; soffA += a_chunk_b;
            i_c0 += 32;
            if (i_c0 >= d.Cin) {
                i_c0 = 0; soffA = 0;
                if (++i_kx == d.kw) { i_kx = 0; ++i_ky; }
                set_tap();
            }
        };
        auto wait_tiles = [&](int tiles) {         // until at most `tiles` whole tiles of this wave's pieces are in flight
            if (tiles >= 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * PPW) : "memory");
            else if (tiles == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PPW) : "memory");
            else if (tiles == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PPW) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        };
        for (int t = 0; t < STAGES && t < ntiles; ++t) issue_tile(t);
        wait_tiles(min(STAGES - 1, ntiles - 1));    // tile 0 has landed (this wave's pieces) ...
        if (DIAG != 3) asm volatile("s_barrier" ::: "memory");     // ... and everybody else's
        int stage = 0;
        for (int t = 0; t + 1 < ntiles; ++t) {
            wait_tiles(min(STAGES - 2, ntiles - 2 - t));   // tile t+1 landed; tiles t+2 .. t+STAGES-1 may still fly
            if (DIAG != 3) asm volatile("s_barrier" ::: "memory"); // consumers have every fragment of tile t in registers: its stage is free
            if (t + STAGES < ntiles) issue_tile(stage);
            stage = stage == STAGES - 1 ? 0 : stage + 1;
        }
        return;
    }

    // ---------------------------------------------------------------------- consumer waves
    float* __restrict__ C = d.c + (size_t)bz * d.batch_stride_c;
    const int grp = KPAR == 2 ? wave >> 2 : 0, cw = KPAR == 2 ? wave & 3 : wave;
    const int wm = cw / WN, wn = cw % WN;
    const int li = lane & 31, lh = lane >> 5;

    f32x16 acc[TM][TN], tot[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc[i][j][r] = 0.f; tot[i][j][r] = 0.f; }
    constexpr int KBLK = 8;                          // K steps per accumulation block (256 k, as the fp32 kernels)

    // per-lane fragment addresses of the two 16-k steps of a tile (the XOR swizzle is lane dependent)
    const char* pa[2];
    const char* pb[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int ch = (2 * j + lh) ^ ((li >> 2) & 3);
        pa[j] = sm + (wm * TM * 32 + li) * 64 + ch * 16;
        pb[j] = sm + (BM + wn * TN * 32 + li) * 64 + ch * 16;
    }
    if constexpr (KPAR == 2) {
        // ---- two consumer groups: this wave owns 16-k half `grp` of every K step.  Fragments of step t + 1 are requested under the first three
        // products of step t (one whole step ahead of their use); the barrier of step t + 1 comes once they have all returned.
        bf16x8 ga[2][3][TM], gb[2][3][TN];
        const int chg = (2 * grp + lh) ^ ((li >> 2) & 3);        // (computed here, not selected from pa[] / pb[]: a run-time select loses the LDS address space -> flat loads)
        const char* const qa = sm + (wm * TM * 32 + li) * 64 + chg * 16;
        const char* const qb = sm + (BM + wn * TN * 32 + li) * 64 + chg * 16;
        auto kread = [&](int nbuf, int nstage, int set) {
            const int ra = set == 0 ? 2 : set == 1 ? 0 : 1, rb = set == 0 ? 0 : set == 1 ? 2 : 1;
#pragma unroll
            for (int r = 0; r < TM; ++r) ga[nbuf][ra][r] = *reinterpret_cast<const bf16x8*>(qa + nstage * STAGE_B + ra * PLANE_B + r * 2048);
#pragma unroll
            for (int r = 0; r < TN; ++r) gb[nbuf][rb][r] = *reinterpret_cast<const bf16x8*>(qb + nstage * STAGE_B + rb * PLANE_B + r * 2048);
        };
        auto kprod = [&](int buf, int nbuf, int p, int q, int nstage, int set) {
            const int ra = set == 0 ? 2 : set == 1 ? 0 : 1, rb = set == 0 ? 0 : set == 1 ? 2 : 1;
            constexpr int NM = TM * TN, NR = TM + TN;
            int issued = 0;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int jn = 0; jn < TN; ++jn) {
                    acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ga[buf][p][i], gb[buf][q][jn], acc[i][jn], 0, 0, 0);
                    if (set >= 0) {
                        const int upto = ((i * TN + jn + 1) * NR + NM - 1) / NM;
#pragma unroll
                        for (int r = 0; r < NR; ++r)
                            if (r >= issued && r < upto) {
                                if (r < TM) ga[nbuf][ra][r] = *reinterpret_cast<const bf16x8*>(qa + nstage * STAGE_B + ra * PLANE_B + r * 2048);
                                else gb[nbuf][rb][r - TM] = *reinterpret_cast<const bf16x8*>(qb + nstage * STAGE_B + rb * PLANE_B + (r - TM) * 2048);
                            }
                        issued = upto;
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
        };
        constexpr bool EPI_AHEAD2 = TM * TN == 1;
        EpiOperands<TM, TN> eop2;
        if (EPI_AHEAD2 && grp == 0) {
            gemm_epilogue_consts<TM, TN>(d, eop2, n0, wn, li, split);
            gemm_epilogue_load<TM, TN>(d, eop2, m0, n0, wm, wn, li, lh, split);
            __builtin_amdgcn_sched_barrier(0);
        }
        asm volatile("s_barrier" ::: "memory");                 // tile 0 is in LDS
        kread(0, 0, 0); kread(0, 0, 1); kread(0, 0, 2);
        __builtin_amdgcn_sched_barrier(0);
        auto kstep = [&](auto more_c, int buf, int nbuf, int s, int t, bool more_rt) {
            constexpr bool MORE = decltype(more_c)::value;
            if (MORE || more_rt) {
                // the fragments of step t have returned (lgkmcnt(0)): nobody reads stage s any more; step t + 1 has landed everywhere
                asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
                kprod(buf, nbuf, 2, 0, (s + 1) % STAGES, 0); kprod(buf, nbuf, 0, 2, (s + 1) % STAGES, 1); kprod(buf, nbuf, 1, 1, (s + 1) % STAGES, 2);
            } else {
                kprod(buf, nbuf, 2, 0, 0, -1); kprod(buf, nbuf, 0, 2, 0, -1); kprod(buf, nbuf, 1, 1, 0, -1);
            }
            kprod(buf, nbuf, 1, 0, 0, -1); kprod(buf, nbuf, 0, 1, 0, -1); kprod(buf, nbuf, 0, 0, 0, -1);
            if ((t & (KBLK - 1)) == KBLK - 1 && (MORE || more_rt)) {
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        tot[i][j] = tot[i][j] + acc[i][j];
#pragma unroll
                        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
                    }
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        static_assert(STAGES % 2 == 0, "the fragment buffers alternate with the steps: an even ring keeps (stage, buffer) literal in the unrolled turn");
        int tb = 0;
        for (; tb + STAGES < ntiles; tb += STAGES) {
#pragma unroll
            for (int s = 0; s < STAGES; ++s) kstep(st_true{}, s & 1, (s & 1) ^ 1, s, tb + s, true);
        }
#pragma unroll
        for (int s = 0; s < STAGES; ++s)
            if (tb + s < ntiles) kstep(st_false{}, s & 1, (s & 1) ^ 1, s, tb + s, tb + s + 1 < ntiles);
        if (ntiles > KBLK) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = acc[i][j] + tot[i][j];
        }
        // the ring is dead (every DMA has landed, every fragment read has returned before the last barrier): the second group's partial tiles cross it
        float* const red = reinterpret_cast<float*>(sm) + (size_t)cw * (TM * TN * 16 * 64) + lane;
        if (grp == 1) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) red[((i * TN + j) * 16 + r) * 64] = acc[i][j][r];
        }
        __syncthreads();                                        // (the loader waves have left: the barrier counts the waves that are still alive)
        if (grp == 1) return;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] += red[((i * TN + j) * 16 + r) * 64];
        if (EPI_AHEAD2) gemm_epilogue_store<TM, TN, false, false>(d, C, acc, eop2, m0, n0, wm, wn, li, lh, split, kz);
        else gemm_tile_epilogue<TM, TN>(d, C, acc, m0, n0, wm, wn, li, lh, split, kz);
        return;
    }
    // (Measured and dropped, round 6, 1x5 conv 8192 x 256 x 1920 on 64x64 tiles: a third fragment buffer so that both 16-k steps of the next
    // tile are requested right behind the barrier -- 44.5 us either way; two alternating accumulators per sub-tile -- 33.1 vs 33.6 us MFMA-only;
    // no barriers at all, MFMA-only -- 33.6 us: the consumer side is MFMA-bound at ~90 % while its workgroup runs, tools/split3_clock.py.)
    bf16x8 fa[2][3][TM], fb[2][3][TN];
    // One product = TM x TN MFMAs on one (A plane, B plane) pair.  The fragment reads of the NEXT 16-k step ride in the gaps between the
    // MFMAs of the first three products, one or two ds_read_b128 per gap (a burst of 12 reads between two MFMAs drains the matrix pipe:
    // MFMA-only runs of the first version, tools/split3_probe.py ST_SPLIT3_DIAG=2, took 34 us where the MFMAs need 21), in the order the
    // next step's products need them: set 0 = (A lo, B hi), set 1 = (A hi, B lo), set 2 = (A mid, B mid).  Each set is thus requested one
    // whole 16-k step (6 TM TN MFMAs) before its first use.
    auto prod = [&](int buf, int p, int q, int nbuf, int nstage, int nj, int set) {
        const int ra = set == 0 ? 2 : set == 1 ? 0 : 1, rb = set == 0 ? 0 : set == 1 ? 2 : 1;
        constexpr int NM = TM * TN, NR = TM + TN;
        int issued = 0;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int jn = 0; jn < TN; ++jn) {
                if (DIAG == 1) asm volatile("" ::"v"(fa[buf][p][i]), "v"(fb[buf][q][jn]));       // keep the fragment reads alive
                else acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[buf][p][i], fb[buf][q][jn], acc[i][jn], 0, 0, 0);
                if (set >= 0) {
                    const int upto = ((i * TN + jn + 1) * NR + NM - 1) / NM;           // reads due after this MFMA
#pragma unroll
                    for (int r = 0; r < NR; ++r)
                        if (r >= issued && r < upto) {
                            if (r < TM) fa[nbuf][ra][r] = *reinterpret_cast<const bf16x8*>(pa[nj] + nstage * STAGE_B + ra * PLANE_B + r * 2048);
                            else fb[nbuf][rb][r - TM] = *reinterpret_cast<const bf16x8*>(pb[nj] + nstage * STAGE_B + rb * PLANE_B + (r - TM) * 2048);
                        }
                    issued = upto;
                }
                __builtin_amdgcn_sched_barrier(0);
            }
    };
    auto read_set = [&](int nbuf, int nstage, int nj, int set) {
        const int ra = set == 0 ? 2 : set == 1 ? 0 : 1, rb = set == 0 ? 0 : set == 1 ? 2 : 1;
#pragma unroll
        for (int r = 0; r < TM; ++r) fa[nbuf][ra][r] = *reinterpret_cast<const bf16x8*>(pa[nj] + nstage * STAGE_B + ra * PLANE_B + r * 2048);
#pragma unroll
        for (int r = 0; r < TN; ++r) fb[nbuf][rb][r] = *reinterpret_cast<const bf16x8*>(pb[nj] + nstage * STAGE_B + rb * PLANE_B + r * 2048);
    };

    // Epilogue operands: fetched AHEAD of the K loop when the wave owns one sub-tile (48 registers; issued after it they are two or three
    // dependent L2 round trips with nothing to hide them: all workgroups of a launch finish together); for larger wave tiles AFTER it
    // (3 x 64 operand registers would have to live beside 128 accumulator + 96 fragment registers at two waves per SIMD).
    constexpr bool EPI_AHEAD = TM * TN == 1;
    EpiOperands<TM, TN> eop;
    if (EPI_AHEAD) {
        gemm_epilogue_consts<TM, TN>(d, eop, n0, wn, li, split);
        gemm_epilogue_load<TM, TN>(d, eop, m0, n0, wm, wn, li, lh, split);
        __builtin_amdgcn_sched_barrier(0);
    }

    if (DIAG != 3) asm volatile("s_barrier" ::: "memory");         // tile 0 is in LDS
    read_set(0, 0, 0, 0); read_set(0, 0, 0, 1); read_set(0, 0, 0, 2);
    __builtin_amdgcn_sched_barrier(0);
    // MORE (a literal in the steady state): tile t+1 exists.  Kept out of a run-time branch there: behind a branch that only sometimes
    // issues the next fragment reads, hipcc must wait for ALL outstanding LDS reads at the merge point.
    // Product order: smallest terms first (lo.hi, hi.lo, mid.mid), the leading product last.
    auto tile_body = [&](auto more_c, int s, int t, bool more_rt) {
        constexpr bool MORE = decltype(more_c)::value;
        // 16-k step 0 (buffer 0); the j = 1 fragments of this tile are requested under its first three products
        prod(0, 2, 0, 1, s, 1, 0); prod(0, 0, 2, 1, s, 1, 1); prod(0, 1, 1, 1, s, 1, 2);
        prod(0, 1, 0, 0, 0, 0, -1); prod(0, 0, 1, 0, 0, 0, -1); prod(0, 0, 0, 0, 0, 0, -1);
        // 16-k step 1 (buffer 1)
        if (MORE || more_rt) {
            // every read of stage s has returned (lgkmcnt(0)); tile t+1 has landed everywhere; the loaders may now refill stage s
            if (DIAG != 3) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            prod(1, 2, 0, 0, (s + 1) % STAGES, 0, 0); prod(1, 0, 2, 0, (s + 1) % STAGES, 0, 1); prod(1, 1, 1, 0, (s + 1) % STAGES, 0, 2);
        } else {
            prod(1, 2, 0, 0, 0, 0, -1); prod(1, 0, 2, 0, 0, 0, -1); prod(1, 1, 1, 0, 0, 0, -1);
        }
        prod(1, 1, 0, 0, 0, 0, -1); prod(1, 0, 1, 0, 0, 0, -1); prod(1, 0, 0, 0, 0, 0, -1);
        if ((t & (KBLK - 1)) == KBLK - 1 && (MORE || more_rt)) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    tot[i][j] = tot[i][j] + acc[i][j];
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
                }
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    unsigned long long st0 = 0, sr0 = 0;
    if (DIAG == 4) { st0 = __builtin_amdgcn_s_memtime(); sr0 = __builtin_amdgcn_s_memrealtime(); }
    int tb = 0;
    for (; tb + STAGES < ntiles; tb += STAGES) {     // every tile of the turn has a successor
#pragma unroll
        for (int s = 0; s < STAGES; ++s) tile_body(st_true{}, s, tb + s, true);
    }
#pragma unroll
    for (int s = 0; s < STAGES; ++s)
        if (tb + s < ntiles) tile_body(st_false{}, s, tb + s, tb + s + 1 < ntiles);
    if (DIAG == 4 && wave == 0 && lane == 0 && d.workspace) {      // in-kernel clock of the K loop: shader cycles per 100 MHz tick (diagnostic build only)
        unsigned long long* o = reinterpret_cast<unsigned long long*>(d.workspace) + (size_t)(blockIdx.x + gridDim.x * blockIdx.z) * 8;
        const unsigned long long sr1 = __builtin_amdgcn_s_memrealtime();
        o[0] = __builtin_amdgcn_s_memtime() - st0; o[1] = sr1 - sr0; o[2] = sr_entry; o[3] = sr0; o[4] = sr1;
    }
    if (ntiles > KBLK) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j] = acc[i][j] + tot[i][j];
    }
    if (EPI_AHEAD) gemm_epilogue_store<TM, TN, false, false>(d, C, acc, eop, m0, n0, wm, wn, li, lh, split, kz);
    else gemm_tile_epilogue<TM, TN>(d, C, acc, m0, n0, wm, wn, li, lh, split, kz);
    if (DIAG == 4 && wave == 0 && lane == 0 && d.workspace) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        reinterpret_cast<unsigned long long*>(d.workspace)[(size_t)(blockIdx.x + gridDim.x * blockIdx.z) * 8 + 5] = __builtin_amdgcn_s_memrealtime();
    }
}

// PERSISTENT walk (64x64 tiles, 3-stage ring): a workgroup walks output tiles w, w + G, w + 2G, ... of its batch with ONE continuous DMA ring,
// so the operand K steps of the next tile stream in under the epilogue of the current one.  For shapes with many tiles and a short K (the
// all-pairs volume: K = 256 = 8 K steps per tile, 4 096 tiles per sample; PatchEmbed's third convolution: 36 K steps, 8 192 tiles) the
// single-tile launch is all ring fill and epilogue: measured on the volume, DMA-only 600 us against MFMA-only 291 us (tools/split3_probe.py).
// Same roles, same barrier protocol as conv_gemm_split3_body -- a tile boundary is just another K step with new addresses for the loaders, and
// "fold, epilogue, clear" between two K steps for the consumers.  CT: the transposed second store of st_corr_volume_both.
template <bool CT>
__device__ __forceinline__ void conv_gemm_split3_persist_body(const st_gemm_desc& d) {
    constexpr int BM = 64, BN = 64, ROWS = 128, STAGES = 3, WN = 2;
    constexpr int PLANE_B = ROWS * 64, STAGE_B = 3 * PLANE_B, GA = 4, GB = 4, PA = 3, PB = 3, PPW = 6;
    extern __shared__ __attribute__((aligned(1024))) float smem[];
    char* const sm = reinterpret_cast<char*>(smem);
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int bz = blockIdx.z;
    const int ntm = (d.M + BM - 1) / BM, ntn = (d.N + BN - 1) / BN, ntl = ntm * ntn;
    const int G = (int)gridDim.x, w = (int)blockIdx.x;
    const int nmine = (ntl - w + G - 1) / G;                 // G <= ntl (host): every workgroup owns at least one tile
    const int nkt = d.K / 32;
    const int total = nmine * nkt;                           // flat (tile, K step) sequence of this workgroup

    if (wave >= 4) {
        // ------------------------------------------------------------------ loader waves
        const int lw = wave - 4;
        const char* A = reinterpret_cast<const char*>(d.a) + (size_t)bz * d.batch_stride_a * 2;
        const char* Wt = reinterpret_cast<const char*>(d.w) + (size_t)bz * d.batch_stride_w * 2;
        const i32x4 rsrcA = make_rsrc(A, d.a_bytes), rsrcW = make_rsrc(Wt, d.w_bytes);
        const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)sm;
        const unsigned a_plane_b = (unsigned)(d.a_plane_stride * 2), w_plane_b = (unsigned)(d.w_plane_stride * 2);
        const unsigned a_chunk_b = (unsigned)(d.a_rows * 64), w_chunk_b = (unsigned)(d.w_rows * 64);
        int a_row[PA], a_iy0[PA], a_ix0[PA];
        unsigned a_c[PA], voffA[PA], voffB[PB], ldsA[PA], ldsB[PB];
#pragma unroll
        for (int i = 0; i < PA; ++i) {
            const int q = lw * PA + i, plane = q / GA, rg = q % GA;
            ldsA[i] = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)(plane * PLANE_B + rg * 1024));
        }
#pragma unroll
        for (int i = 0; i < PB; ++i) {
            const int q = lw * PB + i, plane = q / GB, rg = q % GB;
            ldsB[i] = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)(plane * PLANE_B + BM * 64 + rg * 1024));
        }
        int i_c0 = 0, i_ky = 0, i_kx = 0, i_kt = 0, i_tile = w;
        unsigned soffA = 0, soffB = 0;
        auto set_tap = [&]() {
            const int ty = i_ky * (d.dh > 1 ? d.dh : 1), tx = i_kx * (d.dw > 1 ? d.dw : 1);
            const int tapoff = ty * d.W + tx;
#pragma unroll
            for (int i = 0; i < PA; ++i) {
                const bool ok = (unsigned)(a_iy0[i] + ty) < (unsigned)d.H && (unsigned)(a_ix0[i] + tx) < (unsigned)d.W;
                voffA[i] = ok ? a_c[i] + (unsigned)(a_row[i] + tapoff) * 64u : ST_OOB;
            }
        };
        auto set_tile = [&](int idx) {                      // per-lane addresses of the tile's A rows (tap 0) and B rows
            const int m0 = (idx / ntn) * BM, n0 = (idx % ntn) * BN;
#pragma unroll
            for (int i = 0; i < PA; ++i) {
                const int q = lw * PA + i, plane = q / GA, rg = q % GA;
                const int r = 16 * rg + (lane >> 2), c = (lane & 3) ^ ((r >> 2) & 3);
                const int m = min(m0 + r, d.M - 1);
                const int hw = d.Ho * d.Wo;
                const int b = m / hw, rr = m - b * hw;
                const int oy = rr / d.Wo, ox = rr - oy * d.Wo;
                a_iy0[i] = oy * d.sh - d.ph; a_ix0[i] = ox * d.sw - d.pw;
                a_row[i] = (b * d.H + a_iy0[i]) * d.W + a_ix0[i];
                a_c[i] = (unsigned)plane * a_plane_b + (unsigned)c * 16u;
            }
#pragma unroll
            for (int i = 0; i < PB; ++i) {
                const int q = lw * PB + i, plane = q / GB, rg = q % GB;
                const int r = 16 * rg + (lane >> 2), c = (lane & 3) ^ ((r >> 2) & 3);
                voffB[i] = (unsigned)plane * w_plane_b + (unsigned)min(n0 + r, d.N - 1) * 64u + (unsigned)c * 16u;
            }
            i_c0 = 0; i_ky = 0; i_kx = 0; i_kt = 0; soffA = 0; soffB = 0;
            set_tap();
        };
        set_tile(i_tile);
        auto issue_step = [&](int stage) {
            const unsigned so = (unsigned)(stage * STAGE_B);
            const unsigned sa = __builtin_amdgcn_readfirstlane(soffA), sb = __builtin_amdgcn_readfirstlane(soffB);     // (wave-uniform by construction)
#pragma unroll
            for (int i = 0; i < PA; ++i) lds_dma16(rsrcA, ldsA[i] + so, voffA[i], sa);
#pragma unroll
            for (int i = 0; i < PB; ++i) lds_dma16(rsrcW, ldsB[i] + so, voffB[i], sb);
            if (++i_kt == nkt) { i_tile += G; if (i_tile < ntl) set_tile(i_tile); return; }
            soffB += w_chunk_b; soffA += a_chunk_b;
            i_c0 += 32;
            if (i_c0 >= d.Cin) {
                i_c0 = 0; soffA = 0;
                if (++i_kx == d.kw) { i_kx = 0; ++i_ky; }
                set_tap();
            }
        };
        auto wait_tiles = [&](int tiles) {
            if (tiles >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PPW) : "memory");
            else if (tiles == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PPW) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        };
        for (int t = 0; t < STAGES && t < total; ++t) issue_step(t);
        wait_tiles(min(STAGES - 1, total - 1));
        asm volatile("s_barrier" ::: "memory");
        int stage = 0;
        for (int t = 0; t + 1 < total; ++t) {
            wait_tiles(min(STAGES - 2, total - 2 - t));
            asm volatile("s_barrier" ::: "memory");
            if (t + STAGES < total) issue_step(stage);
            stage = stage == STAGES - 1 ? 0 : stage + 1;
        }
        return;
    }

    // ---------------------------------------------------------------------- consumer waves
    float* __restrict__ C = d.c + (size_t)bz * d.batch_stride_c;
    const int wm = wave / WN, wn = wave % WN;
    const int li = lane & 31, lh = lane >> 5;
    f32x16 acc[1][1], tot;
    constexpr int KBLK = 8;
    unsigned offA[2], offB[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int ch = (2 * j + lh) ^ ((li >> 2) & 3);
        offA[j] = (unsigned)((wm * 32 + li) * 64 + ch * 16);
        offB[j] = (unsigned)((BM + wn * 32 + li) * 64 + ch * 16);
    }
    bf16x8 fa[2][3], fb[2][3];
    auto read_set = [&](int nbuf, int so, int nj, int set) {
        const int ra = set == 0 ? 2 : set == 1 ? 0 : 1, rb = set == 0 ? 0 : set == 1 ? 2 : 1;
        fa[nbuf][ra] = *reinterpret_cast<const bf16x8*>(sm + so + ra * PLANE_B + offA[nj]);
        fb[nbuf][rb] = *reinterpret_cast<const bf16x8*>(sm + so + rb * PLANE_B + offB[nj]);
    };
    auto prod = [&](int buf, int p, int q, int nbuf, int so, int nj, int set) {
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[buf][p], fb[buf][q], acc[0][0], 0, 0, 0);
        if (set >= 0) read_set(nbuf, so, nj, set);
        __builtin_amdgcn_sched_barrier(0);
    };
    asm volatile("s_barrier" ::: "memory");         // step 0 is in LDS
    read_set(0, 0, 0, 0); read_set(0, 0, 0, 1); read_set(0, 0, 0, 2);
    __builtin_amdgcn_sched_barrier(0);
    auto step_body = [&](auto more_c, int stage, int kt) {       // one K step; stage = (global step) % STAGES, a run-time value here
        constexpr bool MORE = decltype(more_c)::value;
        const int so = stage * STAGE_B, sn = (stage == STAGES - 1 ? 0 : stage + 1) * STAGE_B;
        prod(0, 2, 0, 1, so, 1, 0); prod(0, 0, 2, 1, so, 1, 1); prod(0, 1, 1, 1, so, 1, 2);
        prod(0, 1, 0, 0, 0, 0, -1); prod(0, 0, 1, 0, 0, 0, -1); prod(0, 0, 0, 0, 0, 0, -1);
        if (MORE) {
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            prod(1, 2, 0, 0, sn, 0, 0); prod(1, 0, 2, 0, sn, 0, 1); prod(1, 1, 1, 0, sn, 0, 2);
        } else {
            prod(1, 2, 0, 0, 0, 0, -1); prod(1, 0, 2, 0, 0, 0, -1); prod(1, 1, 1, 0, 0, 0, -1);
        }
        prod(1, 1, 0, 0, 0, 0, -1); prod(1, 0, 1, 0, 0, 0, -1); prod(1, 0, 0, 0, 0, 0, -1);
        if ((kt & (KBLK - 1)) == KBLK - 1 && kt + 1 < nkt) {
            tot = tot + acc[0][0];
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[0][0][r] = 0.f;
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    int stage = 0;
    for (int ti = 0; ti < nmine; ++ti) {
        const int idx = w + ti * G;
        const int m0 = (idx / ntn) * BM, n0 = (idx % ntn) * BN;
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc[0][0][r] = 0.f; tot[r] = 0.f; }
        const int nsteps = ti == nmine - 1 ? nkt - 1 : nkt;          // the very last K step of the walk has no successor: peeled
        for (int kt = 0; kt < nsteps; ++kt) {
            step_body(st_true{}, stage, kt);
            stage = stage == STAGES - 1 ? 0 : stage + 1;
        }
        if (ti == nmine - 1) step_body(st_false{}, stage, nkt - 1);
        if (nkt > KBLK) acc[0][0] = acc[0][0] + tot;
        {
            // (non-temporal stores for the raw volume -- 64 MiB per sample written once -- were measured: 589 -> 987 us on 8 x 4096 x 4096 x 256;
            // the default write-back path, which lets the L2 merge the 128-byte row pieces of neighbouring waves, stays)
            EpiOperands<1, 1> eop;
            gemm_epilogue_consts<1, 1>(d, eop, n0, wn, li, 1);
            gemm_epilogue_load<1, 1>(d, eop, m0, n0, wm, wn, li, lh, 1);
            gemm_epilogue_store<1, 1, false, CT>(d, C, acc, eop, m0, n0, wm, wn, li, lh, 1, 0);
        }
    }
}

template <bool CT>
__global__ __launch_bounds__(512, 4) void conv_gemm_split3_persist_kernel(const st_gemm_desc d) {
    conv_gemm_split3_persist_body<CT>(d);
}

template <int WM, int WN, int TM, int TN, int STAGES, int DIAG = 0>
__global__ __launch_bounds__(512) void conv_gemm_split3_kernel(const st_gemm_desc d) {
    conv_gemm_split3_body<WM, WN, TM, TN, STAGES, DIAG>(d, (int)blockIdx.x);
}
// the 64x64 configuration lives on TWO workgroups per CU (72 KB of LDS each): 4 waves per SIMD, so at most 128 registers per wave
template <int WM, int WN, int TM, int TN, int STAGES>
__global__ __launch_bounds__(768, 1) void conv_gemm_split3_kpar_kernel(const st_gemm_desc d) {
    conv_gemm_split3_body<WM, WN, TM, TN, STAGES, 0, 2>(d, (int)blockIdx.x);
}

template <int STAGES, int DIAG = 0>
__global__ __launch_bounds__(512, 4) void conv_gemm_split3_kernel64(const st_gemm_desc d) {
    conv_gemm_split3_body<2, 2, 1, 1, STAGES, DIAG>(d, (int)blockIdx.x);
}

// two independent split3 contractions in one launch (st_conv_gemm_pair with split3 descriptors): workgroups [0, tiles0) run d[0], the rest d[1]
template <int WM, int WN, int TM, int TN, int STAGES>
__global__ __launch_bounds__(512, 4) void conv_gemm_split3_pair_kernel(const st_gemm_pair_args g) {
    const bool second = (int)blockIdx.x >= g.tiles0;             // workgroup-uniform
    conv_gemm_split3_body<WM, WN, TM, TN, STAGES, 0>(second ? g.d[1] : g.d[0], second ? (int)blockIdx.x - g.tiles0 : (int)blockIdx.x);
}
