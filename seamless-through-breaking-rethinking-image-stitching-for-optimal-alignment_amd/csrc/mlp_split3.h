// ---------------------------------------------------------------------------------------------
// The C = 128 block tail (st_mlp128: [projection + residual ->] LayerNorm -> fc1 + GELU -> fc2 + residual(s); twins.py:622-623, 785-790)
// on the exact-split contraction of csrc/gemm_split3.h: every fp32 product of the three GEMMs is the sum of six
// v_mfma_f32_32x32x16_bf16 products of the operands' bf16 planes (hi / mid / lo, x == hi + mid + lo), accumulated in fp32.
// Included by gemm.hip behind rowmlp128_kernel, whose structure it keeps:
//   * a wave owns a 32-row block; the block's rows live in REGISTERS in the MFMA operand layout -- lane (li, lh) holds row li's
//     features 8 m + 4 lh + t (m = 0..15, t = 0..3) -- and every product is the TRANSPOSED one (weight fragment first), whose 32 x 32
//     accumulator tile is again that layout: nothing crosses LDS between the layers;
//   * what is new: the activations are split ONCE where they are produced (the loaded / projected rows after LayerNorm: 64 values per
//     lane; the hidden chunk after GELU: 16 values per lane; 11 VALU instructions per pair of values: three v_cvt_pk_bf16_f32, four
//     exact subtractions, four shifts / masks) and the weights arrive pre-split: st_mlp128_split3_pack writes, per step of the walk,
//     the LDS IMAGE of that step -- [W1 chunk: 3 planes x 32 rows x 256 B | W2 slice: 3 planes x 128 rows x 64 B | 32 bias floats] =
//     49 KiB, XOR-swizzled 16-byte slots and K order already those of the fragment reads -- so a step's weights are 49 linear
//     1-KiB `buffer_load_dwordx4 ... lds` copies and the loop holds no global load besides them;
//   * a 3-stage ring (147 KB of LDS: one workgroup of four waves per CU, one wave per SIMD with up to 512 registers) because the
//     hidden walk is SOFTWARE-PIPELINED inside the wave: while chunk c's 16 hidden values per lane go through bias + GELU + split on
//     the VALU, the matrix pipe runs chunk c + 1's fc1 product (48 MFMAs); then chunk c's fc2 products (48 MFMAs) run with the
//     fragment reads of the next group between them.  One barrier per step, the DMA of step q + 2 issued right behind it.
//   * what bounds it (tools/probes/mfma_bf16_chain.hip, in-kernel stamps ST_MLP3_DIAG=1, profiles/r6_mlp_split3_*): a wave issues one
//     v_mfma_f32_32x32x16_bf16 per 32 cycles whatever it accumulates into, and each one keeps the SIMD's VALU issue busy for ~23 of
//     them -- also for the OTHER wave of the SIMD: an eight-wave variant (two per SIMD, the two halves of the workgroup half a step
//     apart so that one's GELU phase meets the other's MFMA phase) measured 4 500 cycles per block and chunk against 4 700 here, with
//     half the chip idle at M = 32 768; it was dropped.  So a chunk costs 96 x ~23 + 4 x (VALU instructions): the GELU + split count
//     per hidden value is what is left to tune.
// Per 32-row block: 16 x 96 (+ 192 with the projection) MFMAs (the fp32 kernel: 2 304 of twice the matrix cycles).
// Accuracy: that of the split3 GEMM (each product exact, fp32 accumulation; measured 0.8x the fp32-MFMA chain's error against
// fp64).  Non-finite activations: the in-register split has no special case -- an inf / NaN value of x or of the hidden layer
// gives NaN in its row (the fp32 kernel gives inf or NaN there).
// K order inside a 16-k MFMA step: lane half lh holds k = 16 ks + 8 qq + 4 lh + t for element e = 4 qq + t of its 8 -- the order in
// which the accumulator layout hands the values over; the packed weights follow it.
#define MS3_W1_B 24576            // 3 planes x 32 rows x 256 B
#define MS3_W2_B 24576            // 3 planes x 128 rows x 64 B
#define MS3_BIAS_B 1024           // 32 floats (128 B) padded to one DMA piece
#define MS3_STAGE_B (MS3_W1_B + MS3_W2_B + MS3_BIAS_B)
#define MS3_PIECES (MS3_STAGE_B / 1024)

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));

__device__ __forceinline__ unsigned ms3_pk(float a, float b) {
    const f32x2_t v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_t));      // v_cvt_pk_bf16_f32 (round to nearest even)
}
// two fp32 values -> three dwords of packed bf16 pairs (hi, mid, lo), x == hi + mid + lo exactly for finite x
__device__ __forceinline__ void ms3_split_pair(float x0, float x1, unsigned& h, unsigned& m, unsigned& l) {
    h = ms3_pk(x0, x1);
    const float r0 = x0 - __uint_as_float(h << 16), r1 = x1 - __uint_as_float(h & 0xffff0000u);
    m = ms3_pk(r0, r1);
    l = ms3_pk(r0 - __uint_as_float(m << 16), r1 - __uint_as_float(m & 0xffff0000u));
}

// image of the walk: step s < npre: [Wp chunk s | unused | bp chunk]; step npre + hc: [W1 chunk hc | W2 slice hc | b1 chunk].
// grid (5, steps) x 256 threads: blocks 0-1 the 512 slots of the W1 part, 2-3 those of the W2 part, 4 the bias floats.
__global__ __launch_bounds__(256) void mlp_split3_pack_kernel(const float* __restrict__ w1, const float* __restrict__ b1, const float* __restrict__ w2,
                                                              const float* __restrict__ wp, const float* __restrict__ bp, int hidden,
                                                              unsigned char* __restrict__ image) {
    const int step = blockIdx.y, npre = wp ? 4 : 0;
    const bool pre = step < npre;
    const int c = pre ? step : step - npre;
    unsigned char* st = image + (size_t)step * MS3_STAGE_B;
    const int b = blockIdx.x, t = threadIdx.x;
    if (b < 2) {
        const int idx = b * 256 + t, r = idx >> 4, sl = idx & 15, ks = sl >> 1, lh = sl & 1;
        const float* W = (pre ? wp : w1) + (size_t)(32 * c + r) * 128;
        bf16x8 h, m, l;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            __bf16 a0, a1, a2;
            st_split3(W[16 * ks + 8 * (e >> 2) + 4 * lh + (e & 3)], a0, a1, a2);
            h[e] = a0; m[e] = a1; l[e] = a2;
        }
        unsigned char* o = st + r * 256 + ((sl ^ (r & 15)) << 4);
        *reinterpret_cast<bf16x8*>(o) = h;
        *reinterpret_cast<bf16x8*>(o + 8192) = m;
        *reinterpret_cast<bf16x8*>(o + 16384) = l;
    } else if (b < 4) {
        const int idx = (b - 2) * 256 + t, r = idx >> 2, sl = idx & 3, ks = sl >> 1, lh = sl & 1;
        bf16x8 h, m, l;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            __bf16 a0, a1, a2;
            const float v = pre ? 0.f : w2[(size_t)r * hidden + 32 * c + 16 * ks + 8 * (e >> 2) + 4 * lh + (e & 3)];
            st_split3(v, a0, a1, a2);
            h[e] = a0; m[e] = a1; l[e] = a2;
        }
        unsigned char* o = st + MS3_W1_B + r * 64 + ((sl ^ ((r >> 2) & 3)) << 4);
        *reinterpret_cast<bf16x8*>(o) = h;
        *reinterpret_cast<bf16x8*>(o + 8192) = m;
        *reinterpret_cast<bf16x8*>(o + 16384) = l;
    } else {
        float v = 0.f;
        if (t < 32) v = pre ? (bp ? bp[32 * c + t] : 0.f) : b1[32 * c + t];
        reinterpret_cast<float*>(st + MS3_W1_B + MS3_W2_B)[t] = v;
    }
}

// scheduling pattern of a pipelined K step: six MFMAs, each followed by a slice of the VALU work
#define MS3_SCHED_MFMA_VALU(NV)                                \
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);         \
    __builtin_amdgcn_sched_group_barrier(0x002, NV, 0);

#define MS3_NWAVES 4

// GELU for the in-register path: st_gelu's erfc form (Abramowitz-Stegun 7.1.26, |error| < 1.5e-7) with the constants folded -- u = |x| sqrt(log2 e / 2),
// so that exp(-x^2 / 2) = exp2(-u^2); the 0.5 inside the polynomial; x Phi(x) = max(x, 0) - |x| y with y = erfc(|x| / sqrt 2) / 2: 15 VALU
// instructions per value (st_gelu: 19).  Against st_gelu: the same function to ~1e-7 absolute, not the same bits.
__device__ __forceinline__ float ms3_gelu(float x) {
    const float u = fabsf(x) * 0.84932180028801904272f;                      // sqrt(log2(e) / 2)
    const float t = __builtin_amdgcn_rcpf(fmaf(0.2727374808792225f, u, 1.0f));    // 0.3275911 / sqrt(log2 e): 1 + p |x| / sqrt 2 expressed in u
    float y = fmaf(0.5307027145f, t, -0.7265760135f);
    y = fmaf(y, t, 0.7107068705f);
    y = fmaf(y, t, -0.142248368f);
    y = fmaf(y, t, 0.127414796f);
    y = (y * t) * __builtin_amdgcn_exp2f(-(u * u));
    return fmaf(-fabsf(x), y, fmaxf(x, 0.f));
}

// DIAG: s_memtime sums per phase of wave 0 -> diag[8 * blockIdx.x ..]: 0 syncs (wait + barrier), 1 row load + split, 2 projection steps,
// 3 LayerNorm + split, 4 first fc1 product, 5 phase 1, 6 phase 2, 7 last chunk + epilogue (ST_MLP3_DIAG=1, tools/mlp_split3_diag.py)
template <bool PROJ, bool DIAG = false>
__global__ __launch_bounds__(256, 1) void rowmlp128_split3_kernel(const st_mlp_desc d, const unsigned char* __restrict__ image, const unsigned image_bytes,
                                                                  unsigned long long* __restrict__ diag) {
    unsigned long long tsum[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tmark = 0;
#define MS3_T0() if (DIAG) tmark = __builtin_amdgcn_s_memtime();
#define MS3_T1(i) if (DIAG) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); tsum[i] += t_ - tmark; tmark = t_; }
    extern __shared__ __attribute__((aligned(1024))) unsigned char sm3[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, lh = lane >> 5;
    const int nblk = (d.M + 31) >> 5;
    const int G = (int)gridDim.x;
    const int blk0 = (int)blockIdx.x * MS3_NWAVES;
    const int rounds = blk0 < nblk ? (nblk - blk0 + G * MS3_NWAVES - 1) / (G * MS3_NWAVES) : 0;
    constexpr int npre = PROJ ? 4 : 0;
    const int nhc = d.hidden >> 5, spr = npre + nhc, total = rounds * spr;
    if (total == 0) return;
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)sm3;

    // Weights of a step of the walk: image step s_dma -> ring stage g_dma by LDS-DMA (`buffer_load_dwordx4 ... lds`, 1 KiB per instruction).  Every
    // M0 write (the LDS address of a DMA instruction) costs the issuing wave 100-175 cycles (measured in three placements: all pieces behind the
    // barrier, spread between the fc2 MFMA groups, or as plain loads + ds_write_b128 through registers -- 1 300 to 2 300 cycles per step each way), so a
    // wave copies CONTIGUOUS runs: the instruction offset (imm12) advances the global and the LDS address alike, four 1-KiB pieces per M0 value.
    // Hidden step: wave w copies bytes [12 288 w, 12 288 (w + 1)) of the 49-KiB image (three runs of four), wave 0 also the bias KiB; projection
    // step (W1 part + bias only): wave w bytes [6 144 w, 6 144 (w + 1)) (a run of four + a run of two).
    int s_dma = 0, g_dma = 0, q_dma = 0;
    const unsigned vlin = (unsigned)lane << 4;
    const i32x4 rsd = make_rsrc(image, image_bytes);
    auto dma_run4 = [&](unsigned lds, unsigned soff) {
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\t"
                     "buffer_load_dwordx4 %1, %2, %3 offen lds\n\t"
                     "buffer_load_dwordx4 %1, %2, %3 offen offset:1024 lds\n\t"
                     "buffer_load_dwordx4 %1, %2, %3 offen offset:2048 lds\n\t"
                     "buffer_load_dwordx4 %1, %2, %3 offen offset:3072 lds"
                     : : "s"(lds), "v"(vlin), "s"(rsd), "s"(soff) : "memory");
    };
    auto dma_run2 = [&](unsigned lds, unsigned soff) {
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\t"
                     "buffer_load_dwordx4 %1, %2, %3 offen lds\n\t"
                     "buffer_load_dwordx4 %1, %2, %3 offen offset:1024 lds"
                     : : "s"(lds), "v"(vlin), "s"(rsd), "s"(soff) : "memory");
    };
    auto dma_step = [&]() {
        const unsigned dst = lds0 + (unsigned)(g_dma * MS3_STAGE_B), src = (unsigned)(s_dma * MS3_STAGE_B);
        if (PROJ && s_dma < npre) {
            const unsigned o = (unsigned)wave * 6144u;
            dma_run4(dst + o, src + o);
            dma_run2(dst + o + 4096u, src + o + 4096u);
        } else {
            const unsigned o = (unsigned)wave * 12288u;
            dma_run4(dst + o, src + o);
            dma_run4(dst + o + 4096u, src + o + 4096u);
            dma_run4(dst + o + 8192u, src + o + 8192u);
        }
        if (wave == 0) lds_dma16(rsd, dst + (unsigned)(MS3_W1_B + MS3_W2_B), vlin, src + (unsigned)(MS3_W1_B + MS3_W2_B));
        ++q_dma;
        s_dma = s_dma + 1 == spr ? 0 : s_dma + 1;
        g_dma = g_dma + 1 == 3 ? 0 : g_dma + 1;
    };
    dma_step();
    if (total > 1) dma_step();
    // sync of step q: the pieces of steps <= q + 1 have landed in every wave's view, everyone is past step q - 1, whose stage takes step q + 2
    auto step_sync = [&]() {
        MS3_T0()
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (q_dma < total) dma_step();
        MS3_T1(0)
    };

    // fragment offsets: W1 part, row li (256 B), 16-B slot (2 ks + lh) ^ (li & 15); W2 part, row 32 t + li (64 B), slot (2 ks + lh) ^ ((li >> 2) & 3)
    int fo1[8], fo2[2];
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) fo1[ks] = li * 256 + (((2 * ks + lh) ^ (li & 15)) << 4);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) fo2[ks] = MS3_W1_B + li * 64 + (((2 * ks + lh) ^ ((li >> 2) & 3)) << 4);

    u32x4 xp[3][8];                                              // the block's rows as three planes: xp[p][ks] = the 8 k of MFMA step ks
    auto split_rows = [&](const float4 (&x)[16]) {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            unsigned h0, m0, l0, h1, m1, l1;
            ms3_split_pair(x[j].x, x[j].y, h0, m0, l0);
            ms3_split_pair(x[j].z, x[j].w, h1, m1, l1);
            xp[0][j >> 1][2 * (j & 1)] = h0; xp[0][j >> 1][2 * (j & 1) + 1] = h1;
            xp[1][j >> 1][2 * (j & 1)] = m0; xp[1][j >> 1][2 * (j & 1) + 1] = m1;
            xp[2][j >> 1][2 * (j & 1)] = l0; xp[2][j >> 1][2 * (j & 1) + 1] = l1;
        }
    };
#define MS3_BF(v) __builtin_bit_cast(bf16x8, v)
    // six products of one 16-k step, smallest terms first (lo.hi, hi.lo, mid.mid, mid.hi, hi.mid, hi.hi); f = weight planes, x = row planes
#define MS3_SIX(acc, f, x0, x1, x2)                                                              \
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(MS3_BF(f[2]), MS3_BF(x0), acc, 0, 0, 0);       \
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(MS3_BF(f[0]), MS3_BF(x2), acc, 0, 0, 0);       \
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(MS3_BF(f[1]), MS3_BF(x1), acc, 0, 0, 0);       \
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(MS3_BF(f[1]), MS3_BF(x0), acc, 0, 0, 0);       \
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(MS3_BF(f[0]), MS3_BF(x1), acc, 0, 0, 0);       \
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(MS3_BF(f[0]), MS3_BF(x0), acc, 0, 0, 0);

    // K = 128 product of the block with the 32-row weight chunk of ring stage g (not pipelined: projection steps, first chunk of a round)
#define MS3_CHUNK128(acc, g, DMA)                                                                                           \
    {                                                                                                                       \
        const unsigned char* ws_ = sm3 + (g) * MS3_STAGE_B;                                                                 \
        _Pragma("unroll") for (int r = 0; r < 16; ++r) acc[r] = 0.f;                                                        \
        u32x4 f[3], fn[3];                                                                                                  \
        _Pragma("unroll") for (int p = 0; p < 3; ++p) f[p] = *reinterpret_cast<const u32x4*>(ws_ + p * 8192 + fo1[0]);      \
        _Pragma("unroll") for (int ks = 0; ks < 8; ++ks) {                                                                  \
            const int kn = ks + 1 < 8 ? ks + 1 : ks;                                                                        \
            _Pragma("unroll") for (int p = 0; p < 3; ++p) fn[p] = *reinterpret_cast<const u32x4*>(ws_ + p * 8192 + fo1[kn]); \
            MS3_SIX(acc, f, xp[0][ks], xp[1][ks], xp[2][ks])                                                                \
            _Pragma("unroll") for (int p = 0; p < 3; ++p) f[p] = fn[p];                                                     \
            __builtin_amdgcn_sched_barrier(0);                                                                              \
        }                                                                                                                   \
    }
    // bias + GELU + split of the hidden values 2 ks, 2 ks + 1 of the lane (accumulator registers r = 2 ks, 2 ks + 1: features
    // 8 (r >> 2) + 4 lh + (r & 3) of the chunk), bias floats from the stage image
#define MS3_GELU_PAIR(ks, acc, bs)                                                                                          \
    {                                                                                                                       \
        const float2 bq = *reinterpret_cast<const float2*>(bs + 8 * ((2 * (ks)) >> 2) + 4 * lh + ((2 * (ks)) & 3));           \
        unsigned h_, m_, l_;                                                                                                \
        ms3_split_pair(ms3_gelu(acc[2 * (ks)] + bq.x), ms3_gelu(acc[2 * (ks) + 1] + bq.y), h_, m_, l_);                     \
        hp[0][(ks) >> 2][(ks) & 3] = h_; hp[1][(ks) >> 2][(ks) & 3] = m_; hp[2][(ks) >> 2][(ks) & 3] = l_;                  \
    }
    // phase 2 of a chunk: the chunk is the k slice [32 hc, 32 hc + 32) of fc2: four accumulator tiles x two 16-k steps
#define MS3_PHASE2(wc)                                                                                                      \
    {                                                                                                                       \
        u32x4 f[3], fn[3];                                                                                                  \
        _Pragma("unroll") for (int p = 0; p < 3; ++p) f[p] = *reinterpret_cast<const u32x4*>(wc + p * 8192 + fo2[0]);       \
        _Pragma("unroll") for (int u = 0; u < 8; ++u) {                                                                     \
            const int oc = u >> 1, ks = u & 1, un = u + 1 < 8 ? u + 1 : u;                                                  \
            _Pragma("unroll") for (int p = 0; p < 3; ++p)                                                                   \
                fn[p] = *reinterpret_cast<const u32x4*>(wc + p * 8192 + (un >> 1) * 2048 + fo2[un & 1]);                    \
            MS3_SIX(o[oc], f, hp[0][ks], hp[1][ks], hp[2][ks])                                                              \
            _Pragma("unroll") for (int p = 0; p < 3; ++p) f[p] = fn[p];                                                     \
            __builtin_amdgcn_sched_barrier(0);                                                                              \
        }                                                                                                                   \
    }

    int g = 0;                                                   // ring stage of the current step of the walk (step % 3)
    auto next_stage = [](int s) { return s + 1 == 3 ? 0 : s + 1; };
    for (int rd = 0; rd < rounds; ++rd) {
        const int blk = blk0 + wave + rd * G * MS3_NWAVES;
        if (blk >= nblk) {                                      // wave-uniform: an idle wave still copies weights and meets the barriers
            for (int s = 0; s < spr; ++s) { step_sync(); g = next_stage(g); }
            continue;
        }
        const int row = blk * 32 + li;
        const bool rok = row < d.M;
        const size_t rowc = (size_t)(rok ? row : d.M - 1);      // rows past M (last block only) read a valid row and are never stored
        float4 xf[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) xf[j] = *reinterpret_cast<const float4*>(d.a + rowc * d.lda + 8 * j + 4 * lh);
        MS3_T0()
        if (PROJ) {
            // x = a . wp^T + bp + res0, 32 features per step; x stays in registers (xf), a's planes make way for x's; the residual rows are requested up front
            float4 ev[16];
            if (d.res0) {                                       // (one scalar branch around all sixteen loads: a branch per load makes hipcc wait for each on the spot)
#pragma unroll
                for (int j = 0; j < 16; ++j) ev[j] = *reinterpret_cast<const float4*>(d.res0 + rowc * d.ld_res0 + 8 * j + 4 * lh);
            } else {
#pragma unroll
                for (int j = 0; j < 16; ++j) ev[j] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
            split_rows(xf);
            MS3_T1(1)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                step_sync();
                f32x16 acc;
                MS3_CHUNK128(acc, g, true)
                const float* bs = reinterpret_cast<const float*>(sm3 + g * MS3_STAGE_B + MS3_W1_B + MS3_W2_B);
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
                    const float4 bv = *reinterpret_cast<const float4*>(bs + 8 * jj + 4 * lh);
                    const float4 e = ev[4 * c + jj];
                    // (acc + bias) + residual: the unfused epilogue's order
                    xf[4 * c + jj] = make_float4((acc[4 * jj] + bv.x) + e.x, (acc[4 * jj + 1] + bv.y) + e.y, (acc[4 * jj + 2] + bv.z) + e.z, (acc[4 * jj + 3] + bv.w) + e.w);
                }
                g = next_stage(g);
                MS3_T1(2)
            }
        }
        {
            float4 xn[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) xn[j] = xf[j];
            if (d.ln) {                                         // LayerNorm without affine (gamma / beta are folded into w1 / b1): as rowmlp128_kernel
                float s = 0.f;
#pragma unroll
                for (int j = 0; j < 16; ++j) s += (xn[j].x + xn[j].y) + (xn[j].z + xn[j].w);
                s += __shfl_xor(s, 32, 64);
                const float mean = s * (1.0f / 128.0f);
                float v = 0.f;
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    xn[j].x -= mean; xn[j].y -= mean; xn[j].z -= mean; xn[j].w -= mean;
                    v += (xn[j].x * xn[j].x + xn[j].y * xn[j].y) + (xn[j].z * xn[j].z + xn[j].w * xn[j].w);
                }
                v += __shfl_xor(v, 32, 64);
                const float rstd = 1.0f / sqrtf(v * (1.0f / 128.0f) + d.ln_eps);
#pragma unroll
                for (int j = 0; j < 16; ++j) { xn[j].x *= rstd; xn[j].y *= rstd; xn[j].z *= rstd; xn[j].w *= rstd; }
            }
            split_rows(xn);
        }
        MS3_T1(3)
        __builtin_amdgcn_sched_barrier(0);
        f32x16 o[4];
#pragma unroll
        for (int oc = 0; oc < 4; ++oc)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[oc][r] = 0.f;
        // ---- hidden walk, software-pipelined: acc holds chunk hc's fc1 product when iteration hc starts
        step_sync();
        f32x16 acc;
        MS3_CHUNK128(acc, g, false)
        MS3_T1(4)
        u32x4 hp[3][2];                                         // a chunk's hidden values as planes: two 16-k MFMA steps of fc2
#pragma unroll 1
        for (int hc = 0; hc + 1 < nhc; ++hc) {
            if (hc > 0) step_sync();
            const int gn = next_stage(g);
            const unsigned char* wc = sm3 + g * MS3_STAGE_B;    // this chunk: bias + W2 slice
            const unsigned char* wn = sm3 + gn * MS3_STAGE_B;   // next chunk: W1 chunk
            const float* bs = reinterpret_cast<const float*>(wc + MS3_W1_B + MS3_W2_B);
            f32x16 an;
#pragma unroll
            for (int r = 0; r < 16; ++r) an[r] = 0.f;
            {
                // ---- phase 1: fc1 of chunk hc + 1 on the matrix pipe under bias + GELU + split of chunk hc on the VALU
                u32x4 f[3], fn[3];
#pragma unroll
                for (int p = 0; p < 3; ++p) f[p] = *reinterpret_cast<const u32x4*>(wn + p * 8192 + fo1[0]);
#pragma unroll
                for (int ks = 0; ks < 8; ++ks) {
                    const int kn = ks + 1 < 8 ? ks + 1 : ks;
#pragma unroll
                    for (int p = 0; p < 3; ++p) fn[p] = *reinterpret_cast<const u32x4*>(wn + p * 8192 + fo1[kn]);
                    MS3_SIX(an, f, xp[0][ks], xp[1][ks], xp[2][ks])
                    MS3_GELU_PAIR(ks, acc, bs)
#pragma unroll
                    for (int p = 0; p < 3; ++p) f[p] = fn[p];
                    MS3_SCHED_MFMA_VALU(7) MS3_SCHED_MFMA_VALU(7) MS3_SCHED_MFMA_VALU(7) MS3_SCHED_MFMA_VALU(7) MS3_SCHED_MFMA_VALU(7) MS3_SCHED_MFMA_VALU(7)
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            MS3_T1(5)
            MS3_PHASE2(wc)
            MS3_T1(6)
            acc = an;
            g = gn;
        }
        {
            // last chunk of the block: nothing left to run beside its GELU
            if (nhc > 1) step_sync();
            const unsigned char* wc = sm3 + g * MS3_STAGE_B;
            const float* bs = reinterpret_cast<const float*>(wc + MS3_W1_B + MS3_W2_B);
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) MS3_GELU_PAIR(ks, acc, bs)
            __builtin_amdgcn_sched_barrier(0);
            MS3_PHASE2(wc)
            g = next_stage(g);
        }
        if (rok) {
            // out = (fc2 + b2) + x [+ res]: the unfused epilogue's order; a tile's operands are requested together.  (The b2 pointer is laundered per
            // round: see rowmlp128_kernel.)
            const float* b2p = d.b2;
            asm volatile("" : "+s"(b2p));
#pragma unroll
            for (int oc = 0; oc < 4; ++oc) {
                float4 bb[4], e[4];
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) bb[jj] = *reinterpret_cast<const float4*>(b2p + oc * 32 + 8 * jj + 4 * lh);
                if (d.res) {
#pragma unroll
                    for (int jj = 0; jj < 4; ++jj) e[jj] = *reinterpret_cast<const float4*>(d.res + rowc * d.ld_res + oc * 32 + 8 * jj + 4 * lh);
                } else {
#pragma unroll
                    for (int jj = 0; jj < 4; ++jj) e[jj] = make_float4(0.f, 0.f, 0.f, 0.f);
                }
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
                    const int col = oc * 32 + 8 * jj + 4 * lh;
                    const float4 x = xf[4 * oc + jj];
                    float4 v = make_float4((o[oc][4 * jj] + bb[jj].x) + x.x, (o[oc][4 * jj + 1] + bb[jj].y) + x.y, (o[oc][4 * jj + 2] + bb[jj].z) + x.z,
                                           (o[oc][4 * jj + 3] + bb[jj].w) + x.w);
                    if (d.res) { v.x += e[jj].x; v.y += e[jj].y; v.z += e[jj].z; v.w += e[jj].w; }
                    *reinterpret_cast<float4*>(d.out + rowc * d.ldo + col) = v;
                }
            }
        }
        MS3_T1(7)
    }
    if (DIAG && tid == 0)
        for (int i = 0; i < 8; ++i) diag[8 * blockIdx.x + i] = tsum[i];
}

// ---------------------------------------------------------------------------------------------
// LayerNorm -> Linear(128 -> N) + bias over 128-wide rows (the q | k | v projections behind norm1: twins.py:598-600, encoder.py:156-160; the operator of
// rowstream_gemm_kernel with a_ln) on the same machinery: a wave's 32-row block as three bf16 planes in registers, the weights' 32-feature chunks as
// pre-split LDS images ([W chunk 3 x 32 x 256 B | 32 bias floats] = 25 KiB per step) through a 3-stage DMA ring, 48 MFMAs per chunk, the chunk's 32
// output features stored straight from the accumulator layout (16-byte runs).  75 KB of LDS and ~150 registers: two workgroups per CU, so one wave's
// stores and LayerNorm meet the other's MFMAs.
#define LS3_STAGE_B (MS3_W1_B + MS3_BIAS_B)

// grid (3, N / 32) x 256 threads: blocks 0-1 the 512 slots of the chunk, 2 the bias floats
__global__ __launch_bounds__(256) void rowlin_split3_pack_kernel(const float* __restrict__ w, const float* __restrict__ b, unsigned char* __restrict__ image) {
    const int c = blockIdx.y, t = threadIdx.x;
    unsigned char* st = image + (size_t)c * LS3_STAGE_B;
    if (blockIdx.x < 2) {
        const int idx = blockIdx.x * 256 + t, r = idx >> 4, sl = idx & 15, ks = sl >> 1, lh = sl & 1;
        const float* W = w + (size_t)(32 * c + r) * 128;
        bf16x8 h, m, l;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            __bf16 a0, a1, a2;
            st_split3(W[16 * ks + 8 * (e >> 2) + 4 * lh + (e & 3)], a0, a1, a2);
            h[e] = a0; m[e] = a1; l[e] = a2;
        }
        unsigned char* o = st + r * 256 + ((sl ^ (r & 15)) << 4);
        *reinterpret_cast<bf16x8*>(o) = h;
        *reinterpret_cast<bf16x8*>(o + 8192) = m;
        *reinterpret_cast<bf16x8*>(o + 16384) = l;
    } else {
        reinterpret_cast<float*>(st + MS3_W1_B)[t] = (t < 32 && b) ? b[32 * c + t] : 0.f;
    }
}

// AUX: + aux[row / row_div, col] (a per-position table shared by row_div consecutive rows: the context / position part of q | k, encoder.py:95-110)
template <bool AUX>
__global__ __launch_bounds__(256, 2) void rowlin128_split3_kernel(const float* __restrict__ a, const int lda, float* __restrict__ out, const int ldo, const int M,
                                                                  const int N, const int ln, const float ln_eps, const unsigned char* __restrict__ image,
                                                                  const unsigned image_bytes, const float* __restrict__ aux, const int ld_aux, const int row_div) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char sm3[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, lh = lane >> 5;
    const int nblk = (M + 31) >> 5;
    const int G = (int)gridDim.x;
    const int blk0 = (int)blockIdx.x * 4;
    const int rounds = blk0 < nblk ? (nblk - blk0 + G * 4 - 1) / (G * 4) : 0;
    const int spr = N >> 5, total = rounds * spr;
    if (total == 0) return;
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)sm3;
    const i32x4 rsd = make_rsrc(image, image_bytes);
    const unsigned vlin = (unsigned)lane << 4;
    int s_dma = 0, g_dma = 0, q_dma = 0;
    auto dma_step = [&]() {                                      // wave w: bytes [6 144 w, 6 144 (w + 1)) of the chunk as a run of four + a run of two; wave 0 the bias KiB
        const unsigned dst = lds0 + (unsigned)(g_dma * LS3_STAGE_B), src = (unsigned)(s_dma * LS3_STAGE_B), o = (unsigned)wave * 6144u;
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\t"
                     "buffer_load_dwordx4 %1, %2, %3 offen lds\n\t"
                     "buffer_load_dwordx4 %1, %2, %3 offen offset:1024 lds\n\t"
                     "buffer_load_dwordx4 %1, %2, %3 offen offset:2048 lds\n\t"
                     "buffer_load_dwordx4 %1, %2, %3 offen offset:3072 lds"
                     : : "s"(dst + o), "v"(vlin), "s"(rsd), "s"(src + o) : "memory");
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\t"
                     "buffer_load_dwordx4 %1, %2, %3 offen lds\n\t"
                     "buffer_load_dwordx4 %1, %2, %3 offen offset:1024 lds"
                     : : "s"(dst + o + 4096u), "v"(vlin), "s"(rsd), "s"(src + o + 4096u) : "memory");
        if (wave == 0) lds_dma16(rsd, dst + (unsigned)MS3_W1_B, vlin, src + (unsigned)MS3_W1_B);
        ++q_dma;
        s_dma = s_dma + 1 == spr ? 0 : s_dma + 1;
        g_dma = g_dma + 1 == 3 ? 0 : g_dma + 1;
    };
    dma_step();
    if (total > 1) dma_step();
    int fo1[8];
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) fo1[ks] = li * 256 + (((2 * ks + lh) ^ (li & 15)) << 4);
    int g = 0;
    for (int rd = 0; rd < rounds; ++rd) {
        const int blk = blk0 + wave + rd * G * 4;
        const bool active = blk < nblk;                         // wave-uniform; an idle wave still copies weights and meets the barriers
        const int row = blk * 32 + li;
        const bool rok = active && row < M;
        const size_t rowc = (size_t)(active && row < M ? row : M - 1);
        u32x4 xp[3][8];
        if (active) {
            float4 x[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) x[j] = *reinterpret_cast<const float4*>(a + rowc * lda + 8 * j + 4 * lh);
            if (ln) {
                float s = 0.f;
#pragma unroll
                for (int j = 0; j < 16; ++j) s += (x[j].x + x[j].y) + (x[j].z + x[j].w);
                s += __shfl_xor(s, 32, 64);
                const float mean = s * (1.0f / 128.0f);
                float v = 0.f;
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    x[j].x -= mean; x[j].y -= mean; x[j].z -= mean; x[j].w -= mean;
                    v += (x[j].x * x[j].x + x[j].y * x[j].y) + (x[j].z * x[j].z + x[j].w * x[j].w);
                }
                v += __shfl_xor(v, 32, 64);
                const float rstd = 1.0f / sqrtf(v * (1.0f / 128.0f) + ln_eps);
#pragma unroll
                for (int j = 0; j < 16; ++j) { x[j].x *= rstd; x[j].y *= rstd; x[j].z *= rstd; x[j].w *= rstd; }
            }
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                unsigned h0, m0, l0, h1, m1, l1;
                ms3_split_pair(x[j].x, x[j].y, h0, m0, l0);
                ms3_split_pair(x[j].z, x[j].w, h1, m1, l1);
                xp[0][j >> 1][2 * (j & 1)] = h0; xp[0][j >> 1][2 * (j & 1) + 1] = h1;
                xp[1][j >> 1][2 * (j & 1)] = m0; xp[1][j >> 1][2 * (j & 1) + 1] = m1;
                xp[2][j >> 1][2 * (j & 1)] = l0; xp[2][j >> 1][2 * (j & 1) + 1] = l1;
            }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll 1
        for (int c = 0; c < spr; ++c) {
            // this wave's DMA pieces of the step have landed; the four stores of its previous chunk (younger than those pieces; an active wave always
            // stores: its first row exists) may still be in flight
            if (c > 0 && active) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (q_dma < total) dma_step();
            if (active) {
                const unsigned char* ws_ = sm3 + g * LS3_STAGE_B;
                float4 ev[4];
                if (AUX) {                                      // the chunk's table values: requested before the MFMAs, used after them
#pragma unroll
                    for (int jj = 0; jj < 4; ++jj) ev[jj] = *reinterpret_cast<const float4*>(aux + (rowc / (size_t)row_div) * ld_aux + (c << 5) + 8 * jj + 4 * lh);
                }
                f32x16 acc;
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] = 0.f;
                u32x4 f[3], fn[3];
#pragma unroll
                for (int p = 0; p < 3; ++p) f[p] = *reinterpret_cast<const u32x4*>(ws_ + p * 8192 + fo1[0]);
#pragma unroll
                for (int ks = 0; ks < 8; ++ks) {
                    const int kn = ks + 1 < 8 ? ks + 1 : ks;
#pragma unroll
                    for (int p = 0; p < 3; ++p) fn[p] = *reinterpret_cast<const u32x4*>(ws_ + p * 8192 + fo1[kn]);
                    MS3_SIX(acc, f, xp[0][ks], xp[1][ks], xp[2][ks])
#pragma unroll
                    for (int p = 0; p < 3; ++p) f[p] = fn[p];
                    __builtin_amdgcn_sched_barrier(0);
                }
                const float* bs = reinterpret_cast<const float*>(ws_ + MS3_W1_B);
                float4 v[4];
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
                    const float4 bv = *reinterpret_cast<const float4*>(bs + 8 * jj + 4 * lh);
                    v[jj] = make_float4(acc[4 * jj] + bv.x, acc[4 * jj + 1] + bv.y, acc[4 * jj + 2] + bv.z, acc[4 * jj + 3] + bv.w);     // (acc + bias) + table: the fp32 epilogue's order
                    if (AUX) { v[jj].x += ev[jj].x; v[jj].y += ev[jj].y; v[jj].z += ev[jj].z; v[jj].w += ev[jj].w; }
                }
                if (rok) {
#pragma unroll
                    for (int jj = 0; jj < 4; ++jj) *reinterpret_cast<float4*>(out + rowc * ldo + (c << 5) + 8 * jj + 4 * lh) = v[jj];
                }
            }
            g = g + 1 == 3 ? 0 : g + 1;
        }
    }
}


// ---------------------------------------------------------------------------------------------
// PatchEmbed's tail (encoder.py:77-95): tokens = LayerNorm( ffn_with_coord.2( ReLU( ffn_with_coord.0([x | pe(pos)]) ) ) ) over the M * P rows of the 64-channel
// patch features -- three HBM-bound launches over 524 288 rows per pair (64 -> 128 with the position table, 128 -> 128, LayerNorm: 0.36 ms, the
// [M P, 128] intermediate written and read twice) as ONE: both weight matrices fit the LDS as pre-split images (4 x 12 KiB + 4 x 24 KiB = 144 KB), so
// there is no ring and NO barrier after the load: a wave walks its 32-row blocks alone -- rows split into planes in registers, fc1 in four 32-feature
// chunks (K = 64: 24 MFMAs each, the next chunk's under the table add + ReLU + split of the current one), fc2 on four accumulator tiles (48 MFMAs per
// chunk), bias + LayerNorm (affine) in registers, the next block's rows and table values requested one block ahead.
#define PT3_W1_B 12288            // 3 planes x 32 rows x 128 B (K = 64)
#define PT3_IMAGE_B (4 * PT3_W1_B + 4 * MS3_W2_B)
#define PT3_VEC_B 1536            // b2 | gamma | beta

// grid (3, 4) x 256 threads: block 0 the 256 slots of W1 chunk c (slot = 16-k step and lane half of a row, XOR (row >> 1) & 7), 1-2 the 512 slots of W2 slice c
__global__ __launch_bounds__(256) void pe_tail_split3_pack_kernel(const float* __restrict__ w1, const int ld1, const float* __restrict__ w2, unsigned char* __restrict__ image) {
    const int c = blockIdx.y, t = threadIdx.x;
    bf16x8 h, m, l;
    unsigned char* o;
    if (blockIdx.x == 0) {
        const int r = t >> 3, sl = t & 7, ks = sl >> 1, lh = sl & 1;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            __bf16 a0, a1, a2;
            st_split3(w1[(size_t)(32 * c + r) * ld1 + 16 * ks + 8 * (e >> 2) + 4 * lh + (e & 3)], a0, a1, a2);
            h[e] = a0; m[e] = a1; l[e] = a2;
        }
        o = image + c * PT3_W1_B + r * 128 + ((sl ^ ((r >> 1) & 7)) << 4);
        *reinterpret_cast<bf16x8*>(o) = h;
        *reinterpret_cast<bf16x8*>(o + 4096) = m;
        *reinterpret_cast<bf16x8*>(o + 8192) = l;
    } else {
        const int idx = (blockIdx.x - 1) * 256 + t, r = idx >> 2, sl = idx & 3, ks = sl >> 1, lh = sl & 1;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            __bf16 a0, a1, a2;
            st_split3(w2[(size_t)r * 128 + 32 * c + 16 * ks + 8 * (e >> 2) + 4 * lh + (e & 3)], a0, a1, a2);
            h[e] = a0; m[e] = a1; l[e] = a2;
        }
        o = image + 4 * PT3_W1_B + c * MS3_W2_B + r * 64 + ((sl ^ ((r >> 2) & 3)) << 4);
        *reinterpret_cast<bf16x8*>(o) = h;
        *reinterpret_cast<bf16x8*>(o + 8192) = m;
        *reinterpret_cast<bf16x8*>(o + 16384) = l;
    }
}

// TABINV: the table rows of a wave's blocks are the same for every block (the host picks a grid with (gridDim.x * 128) % P == 0): loaded once, and the loop
// holds only the eight row loads of the next block and the sixteen stores of the current one
template <bool TABINV>
__global__ __launch_bounds__(256, 1) void pe_tail_split3_kernel(const float* __restrict__ x, const float* __restrict__ tab, const unsigned char* __restrict__ image,
                                                                const float* __restrict__ b2, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                const float eps, float* __restrict__ out, const int R, const int P) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char sm3[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, lh = lane >> 5;
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)sm3;
    {
        // the whole image once: wave w copies bytes [36 864 w, 36 864 (w + 1)) in nine runs of four 1-KiB pieces
        const i32x4 rsd = make_rsrc(image, (unsigned)PT3_IMAGE_B);
        const unsigned vlin = (unsigned)lane << 4;
#pragma unroll
        for (int u = 0; u < 9; ++u) {
            const unsigned o = (unsigned)wave * 36864u + (unsigned)u * 4096u;
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\t"
                         "buffer_load_dwordx4 %1, %2, %3 offen lds\n\t"
                         "buffer_load_dwordx4 %1, %2, %3 offen offset:1024 lds\n\t"
                         "buffer_load_dwordx4 %1, %2, %3 offen offset:2048 lds\n\t"
                         "buffer_load_dwordx4 %1, %2, %3 offen offset:3072 lds"
                         : : "s"(lds0 + o), "v"(vlin), "s"(rsd), "s"(o) : "memory");
        }
        float* vec = reinterpret_cast<float*>(sm3 + PT3_IMAGE_B);
        if (tid < 128) { vec[tid] = b2[tid]; vec[128 + tid] = gamma[tid]; vec[256 + tid] = beta[tid]; }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    const float* vec = reinterpret_cast<const float*>(sm3 + PT3_IMAGE_B) + 4 * lh;
    const int fo1_row = li * 128, fo1_x = (li >> 1) & 7;
    int fo2[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) fo2[ks] = 4 * PT3_W1_B + li * 64 + (((2 * ks + lh) ^ ((li >> 2) & 3)) << 4);
    const int nblk = (R + 31) >> 5, stride = (int)gridDim.x * 4;
    int blk = (int)blockIdx.x * 4 + wave;
    if (blk >= nblk) return;

    float4 xa[8], tv[16];
    auto load_rows = [&](int b, float4 (&xo)[8]) {
        const int row = b * 32 + li;
        const float* xr = x + (size_t)(row < R ? row : R - 1) * 64 + 4 * lh;
#pragma unroll
        for (int j = 0; j < 8; ++j) xo[j] = *reinterpret_cast<const float4*>(xr + 8 * j);
    };
    auto load_table = [&](int b, float4 (&to)[16]) {
        const int row = b * 32 + li;            // (rows past R: the table row of their own index -- the same row in every block of the wave, never stored)
        const float* tr = tab + (size_t)(row % P) * 128 + 4 * lh;
#pragma unroll
        for (int j = 0; j < 16; ++j) to[j] = *reinterpret_cast<const float4*>(tr + 8 * j);
    };
    load_rows(blk, xa);
    load_table(blk, tv);
#define PT3_BF(v) __builtin_bit_cast(bf16x8, v)
#pragma unroll 1
    for (; blk < nblk; blk += stride) {
        const int row = blk * 32 + li;
        const bool rok = row < R;
        const size_t rowc = (size_t)(rok ? row : R - 1);
        u32x4 xp[3][4];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            unsigned h0, m0, l0, h1, m1, l1;
            ms3_split_pair(xa[j].x, xa[j].y, h0, m0, l0);
            ms3_split_pair(xa[j].z, xa[j].w, h1, m1, l1);
            xp[0][j >> 1][2 * (j & 1)] = h0; xp[0][j >> 1][2 * (j & 1) + 1] = h1;
            xp[1][j >> 1][2 * (j & 1)] = m0; xp[1][j >> 1][2 * (j & 1) + 1] = m1;
            xp[2][j >> 1][2 * (j & 1)] = l0; xp[2][j >> 1][2 * (j & 1) + 1] = l1;
        }
        float4 tc[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) tc[j] = tv[j];
        // the next block's rows, one block ahead -- UNCONDITIONALLY (the last block re-reads itself): behind a branch the compiler has two paths to merge
        // and waits for vmcnt(0) at the loop top, i.e. for the sixteen stores of the previous block as well
        const int nb = blk + stride;
        load_rows(nb < nblk ? nb : blk, xa);
        if (!TABINV) load_table(nb < nblk ? nb : blk, tv);
        __builtin_amdgcn_sched_barrier(0);
        f32x16 o[4];
#pragma unroll
        for (int oc = 0; oc < 4; ++oc)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[oc][r] = 0.f;
        // fc1 chunk 0 (K = 64: four 16-k steps)
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        {
            const unsigned char* w1 = sm3 + fo1_row;
            u32x4 f[3], fn[3];
#pragma unroll
            for (int p = 0; p < 3; ++p) f[p] = *reinterpret_cast<const u32x4*>(w1 + p * 4096 + (((0 + lh) ^ fo1_x) << 4));
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const int kn = ks + 1 < 4 ? ks + 1 : ks;
#pragma unroll
                for (int p = 0; p < 3; ++p) fn[p] = *reinterpret_cast<const u32x4*>(w1 + p * 4096 + (((2 * kn + lh) ^ fo1_x) << 4));
                MS3_SIX(acc, f, xp[0][ks], xp[1][ks], xp[2][ks])
#pragma unroll
                for (int p = 0; p < 3; ++p) f[p] = fn[p];
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            u32x4 hp[3][2];
            f32x16 an;
#pragma unroll
            for (int r = 0; r < 16; ++r) an[r] = 0.f;
            // table add + ReLU + split of chunk c [under fc1 of chunk c + 1, its fragments read one 16-k step ahead]
            u32x4 g1[3], g1n[3];
            if (c < 3) {
#pragma unroll
                for (int p = 0; p < 3; ++p) g1[p] = *reinterpret_cast<const u32x4*>(sm3 + (c + 1) * PT3_W1_B + fo1_row + p * 4096 + (((0 + lh) ^ fo1_x) << 4));
            }
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                if (c < 3) {
                    const unsigned char* w1 = sm3 + (c + 1) * PT3_W1_B + fo1_row;
                    const int kn = ks + 1 < 4 ? ks + 1 : ks;
#pragma unroll
                    for (int p = 0; p < 3; ++p) g1n[p] = *reinterpret_cast<const u32x4*>(w1 + p * 4096 + (((2 * kn + lh) ^ fo1_x) << 4));
                    MS3_SIX(an, g1, xp[0][ks], xp[1][ks], xp[2][ks])
#pragma unroll
                    for (int p = 0; p < 3; ++p) g1[p] = g1n[p];
                }
                {
                    const float4 t4 = tc[4 * c + ks];            // hidden features 32 c + 8 ks + 4 lh + (0..3): accumulator registers 4 ks .. 4 ks + 3
                    unsigned h0, m0, l0, h1, m1, l1;
                    ms3_split_pair(fmaxf(acc[4 * ks] + t4.x, 0.f), fmaxf(acc[4 * ks + 1] + t4.y, 0.f), h0, m0, l0);
                    ms3_split_pair(fmaxf(acc[4 * ks + 2] + t4.z, 0.f), fmaxf(acc[4 * ks + 3] + t4.w, 0.f), h1, m1, l1);
                    hp[0][ks >> 1][2 * (ks & 1)] = h0; hp[0][ks >> 1][2 * (ks & 1) + 1] = h1;
                    hp[1][ks >> 1][2 * (ks & 1)] = m0; hp[1][ks >> 1][2 * (ks & 1) + 1] = m1;
                    hp[2][ks >> 1][2 * (ks & 1)] = l0; hp[2][ks >> 1][2 * (ks & 1) + 1] = l1;
                }
                if (c < 3) { MS3_SCHED_MFMA_VALU(5) MS3_SCHED_MFMA_VALU(5) MS3_SCHED_MFMA_VALU(5) MS3_SCHED_MFMA_VALU(5) MS3_SCHED_MFMA_VALU(5) MS3_SCHED_MFMA_VALU(5) }
                __builtin_amdgcn_sched_barrier(0);
            }
            // fc2: the chunk is the k slice [32 c, 32 c + 32): four accumulator tiles x two 16-k steps
            {
                const unsigned char* wc = sm3 + c * MS3_W2_B;
                u32x4 f[3], fn[3];
#pragma unroll
                for (int p = 0; p < 3; ++p) f[p] = *reinterpret_cast<const u32x4*>(wc + p * 8192 + fo2[0]);
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int oc = u >> 1, ks = u & 1, un = u + 1 < 8 ? u + 1 : u;
#pragma unroll
                    for (int p = 0; p < 3; ++p) fn[p] = *reinterpret_cast<const u32x4*>(wc + p * 8192 + (un >> 1) * 2048 + fo2[un & 1]);
                    MS3_SIX(o[oc], f, hp[0][ks], hp[1][ks], hp[2][ks])
#pragma unroll
                    for (int p = 0; p < 3; ++p) f[p] = fn[p];
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            acc = an;
        }
        // + b2, LayerNorm over the row's 128 values (lanes li and li + 32 hold them), affine, store
        float s = 0.f;
#pragma unroll
        for (int oc = 0; oc < 4; ++oc)
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                const float4 bb = *reinterpret_cast<const float4*>(vec + oc * 32 + 8 * jj);
                o[oc][4 * jj] += bb.x; o[oc][4 * jj + 1] += bb.y; o[oc][4 * jj + 2] += bb.z; o[oc][4 * jj + 3] += bb.w;
                s += (o[oc][4 * jj] + o[oc][4 * jj + 1]) + (o[oc][4 * jj + 2] + o[oc][4 * jj + 3]);
            }
        s += __shfl_xor(s, 32, 64);
        const float mean = s * (1.0f / 128.0f);
        float v = 0.f;
#pragma unroll
        for (int oc = 0; oc < 4; ++oc)
#pragma unroll
            for (int r = 0; r < 16; ++r) { o[oc][r] -= mean; v += o[oc][r] * o[oc][r]; }
        v += __shfl_xor(v, 32, 64);
        const float rstd = 1.0f / sqrtf(v * (1.0f / 128.0f) + eps);
        if (rok) {
#pragma unroll
            for (int oc = 0; oc < 4; ++oc)
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
                    const float4 gg = *reinterpret_cast<const float4*>(vec + 128 + oc * 32 + 8 * jj), be = *reinterpret_cast<const float4*>(vec + 256 + oc * 32 + 8 * jj);
                    *reinterpret_cast<float4*>(out + rowc * 128 + oc * 32 + 8 * jj + 4 * lh) =
                        make_float4(o[oc][4 * jj] * rstd * gg.x + be.x, o[oc][4 * jj + 1] * rstd * gg.y + be.y, o[oc][4 * jj + 2] * rstd * gg.z + be.z,
                                    o[oc][4 * jj + 3] * rstd * gg.w + be.w);
                }
        }
    }
#undef PT3_BF
}

#undef MS3_T0
#undef MS3_T1
#undef MS3_PHASE2
#undef MS3_GELU_PAIR
#undef MS3_CHUNK128
#undef MS3_SIX
#undef MS3_BF
