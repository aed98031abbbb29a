// bl_mlp.hip -- the leaf-evaluation network's body + head Linears (boardlaw/networks.py:10-40) as ONE gfx950 kernel.
//
// PyTorch runs this fp16-autocast MLP as 6 GEMM launches + elementwise launches (about 95 us per 4096-row batch on an
// MI355X, launch- and epilogue-bound at this size).  Here a workgroup of 4-8 waves takes 32 rows through every layer:
// activations live in LDS (the residual stream x, rectified as the next GEMM reads it), weights stream from L2 straight into MFMA fragments
// (v_mfma_f32_32x32x16_f16), and the ReZero tail x + alpha*y / relu are the epilogue.  Rounding points are torch's
// (Linear output, alpha*y, x + ., each rounded to f16); only the K-summation order inside a GEMM differs, so results
// agree with the autocast module to f16 rounding (tests/test_gpu_parity.py::test_fused_mlp_matches_autocast).
// The heads' nonlinearities (masked log-softmax, tanh) stay in bl_sim_finish.
#include "bl_gemm.h"

namespace blmlp {

#ifdef BL_MLP_CLK
__device__ long long g_debug_clk[64];
#define CLK(i) if (blockIdx.x == 0 && threadIdx.x == 0) g_debug_clk[i] = clock64();
#else
#define CLK(i)
#endif

struct Params {
    const uint16_t* obs;      // (M, K0) f16
    const uint16_t* w0;       // (W, K0pad) f16, zero padded in K; this and the other matrices are fragment-major packed
    const uint16_t* b0;       // (W)
    const uint16_t* wb;       // (D, W, W)
    const uint16_t* bb;       // (D, W)
    const float* alphas;      // (D) f32
    const uint16_t* wh;       // (NHpad, W): rows 0..NH-2 policy, row NH-1 value, rest zero
    const uint16_t* bh;       // (NHpad)
    uint16_t* policy;         // (M, NH-1)
    uint16_t* value;          // (M)
    int M, K0, K0pad, W, D, NH, NHpad;
    int xcd_rows;             // 1: tile i takes rows 256*(i/8) + 8*r + i%8 (rows whose index is i mod 8), else rows 32*i + r
    const int32_t* n_active;  // device scalar or null: rows >= *n_active are treated like rows >= M (bl_search_t.n_active)
    int bias_off;             // set by mlp_launch: where, in halves from the LDS base, the (D + 1) x W biases are staged
};

// What bl_sim_finish does for a leaf (heads, store, backup, next q range), as this kernel's epilogue: bl_sim_infer_finish.
// Field meanings as in bl_search_t; qrange points at the row the NEXT descent reads.
struct FinArgs {
    uint16_t* logits; uint16_t* v; uint16_t* w; int16_t* n; const uint16_t* rewards; const uint8_t* terminal;
    const int16_t* path; uint32_t* qrange; const int16_t* leaves; const int32_t* leaf_seats; const uint8_t* valid;
    int T, A, Wsm, iters;
    float* cpi; uint32_t* cca; int16_t* nk; const float* exp_table;      // compacted policy rows (bl_device.h), or cpi == null
};
#ifndef BLM_EPI_AHEAD
#define BLM_EPI_AHEAD 4     // layer epilogue: groups whose LDS reads are in flight ahead of the arithmetic (0 = group by group, as through round 6's first half)
#endif
#ifndef BLM_RD64
#define BLM_RD64 3          // weight-ring depth of the 512-wide, 64-row instantiation (measurement switch)
#endif
#ifndef BLM_RD32
#define BLM_RD32 3          // ... of the 512-wide, 32-row instantiation
#endif
#define BLM_QSLOTS 64
#define BLM_QSTRIDE 64
__device__ __forceinline__ uint32_t enc(float f) {      // order-preserving float -> u32 (as in bl_device.h)
    const uint32_t b = __builtin_bit_cast(uint32_t, f);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
template <int CTRL, int RM>
__device__ __forceinline__ int dpp_i(int old, int v) { return __builtin_amdgcn_update_dpp(old, v, CTRL, RM, 0xf, false); }
__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v) {      // row_shr 1,2,4,8, row_bcast 15/31: lane 63 ends with the max
    v = max(v, (uint32_t)dpp_i<0x111, 0xf>(0, (int)v)); v = max(v, (uint32_t)dpp_i<0x112, 0xf>(0, (int)v));
    v = max(v, (uint32_t)dpp_i<0x114, 0xf>(0, (int)v)); v = max(v, (uint32_t)dpp_i<0x118, 0xf>(0, (int)v));
    v = max(v, (uint32_t)dpp_i<0x142, 0xa>(0, (int)v)); v = max(v, (uint32_t)dpp_i<0x143, 0xc>(0, (int)v));
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}
// x[lane ^ off] for the softmax butterflies, off a power of two: DPP within a row (1, 2, 8), ds_swizzle within 32 lanes (4, 16),
// v_permlane32_swap across the halves -- the same values __shfl_xor's ds_bpermute returns, without its address arithmetic and
// LDS round trip (a butterfly of 6 steps over four envs: 1.5k -> 0.4k cycles)
template <int OFF>
__device__ __forceinline__ float xor_lane(float x) {
    const int v = __builtin_bit_cast(int, x);
    int r;
    if constexpr (OFF == 1) r = __builtin_amdgcn_update_dpp(v, v, 0xB1, 0xf, 0xf, false);            // quad_perm [1,0,3,2]
    else if constexpr (OFF == 2) r = __builtin_amdgcn_update_dpp(v, v, 0x4E, 0xf, 0xf, false);       // quad_perm [2,3,0,1]
    else if constexpr (OFF == 4) r = __builtin_amdgcn_ds_swizzle(v, (4 << 10) | 0x1f);
    else if constexpr (OFF == 8) r = __builtin_amdgcn_update_dpp(v, v, 0x128, 0xf, 0xf, false);      // row_ror:8
    else if constexpr (OFF == 16) r = __builtin_amdgcn_ds_swizzle(v, (16 << 10) | 0x1f);
    else {
        const auto sw = __builtin_amdgcn_permlane32_swap((unsigned)v, (unsigned)v, false, false);   // {[lo,lo], [hi,hi]}
        r = (int)((threadIdx.x & 32) ? sw[0] : sw[1]);
    }
    return __builtin_bit_cast(float, r);
}

__device__ __forceinline__ uint32_t wave_or_u32(uint32_t v) {       // as wave_max_u32, with OR
    v |= (uint32_t)dpp_i<0x111, 0xf>(0, (int)v); v |= (uint32_t)dpp_i<0x112, 0xf>(0, (int)v);
    v |= (uint32_t)dpp_i<0x114, 0xf>(0, (int)v); v |= (uint32_t)dpp_i<0x118, 0xf>(0, (int)v);
    v |= (uint32_t)dpp_i<0x142, 0xa>(0, (int)v); v |= (uint32_t)dpp_i<0x143, 0xc>(0, (int)v);
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}
__device__ __forceinline__ float dpp_next_lane(float beyond, float x) {   // lane j <- x[j + 1]; lane 63 <- `beyond`
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, beyond), __builtin_bit_cast(int, x), 0x130, 0xf, 0xf, false));
}
__device__ __forceinline__ float readlane_f(float v, int lane) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), lane));
}

// WAVES x PASSES x NT x 32 == W: every wave owns PASSES groups of NT 32-column tiles of a body layer's output and works
// through them one group at a time (accumulators and weight ring sized for NT tiles; W = 1024 would not fit otherwise).
// RG: row groups of 32 rows per workgroup (1, or 2 for launches of more than one workgroup per CU; see gemm_run).
template <int NT, int PASSES, int WAVES, bool FINISH, int RD, int RG = 1>
__global__ void __launch_bounds__(WAVES * 64) mlp_kernel(Params p, FinArgs f) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int ROWS = 32 * RG;
    // widths 256 and 512: the body layers' weight stream on hand-placed waits (ring_load_asm)
#ifdef BL_MLP_NO_HANDW
    constexpr bool HANDW = false;
#else
    constexpr bool HANDW = PASSES == 1 && (WAVES * NT * 32 == 512 || WAVES * NT * 32 == 256);
#endif
    const int W = p.W, ld = W + 8;                  // +8 halves: rows 16 B apart in bank space, ds_read_b128 conflict-free
    // Two activation buffers R(0), R(1) of [ROWS][ld] f16 holding the residual stream x: layer l reads x_l from R((l + par0) & 1)
    // (its GEMM rectifies the fragments on the way, its epilogue re-reads its own slice) and writes x_{l+1} to the other one, so
    // one barrier per layer suffices; the last layer's output is the neck the heads read, and
    // par0 is chosen so that the neck lands in R(0), leaving everything from R(1) on to the heads' staging.  (Offsets
    // from the one LDS base, not an array of pointers: the latter decays to generic pointers and turns every LDS
    // access into a flat_load/flat_store.)
    uint16_t* R0 = (uint16_t*)smem;
    const int par0 = (p.D + 1) & 1;
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;      // wave: provably uniform (scalar address arithmetic)
    // Which 32 batch rows this workgroup takes.  Workgroup i runs on XCD i % 8 and so does bl_sim_expand's workgroup for env
    // b = i mod 8 (one workgroup per env, placed the same way): with xcd_rows a tile is made of envs of its own XCD, so what
    // the search kernel just wrote for them (observation, path, leaf) and what this kernel writes for the next descent
    // (logits, compacted row, w, n) stay within one XCD's L2 instead of crossing the fabric.  Placement is a speed matter
    // only: any mapping gives the same results.
    const int tile_j = blockIdx.x >> 3, tile_x = blockIdx.x & 7;
    auto grow = [&](int r) { return p.xcd_rows ? 8 * ROWS * tile_j + 8 * r + tile_x : (int)blockIdx.x * ROWS + r; };
    // rows that exist: M, or fewer when the search runs with only its first *n_active envs (bl_search_t.n_active)
    int Mrows = p.M;
    if (p.n_active) { const int na = __builtin_amdgcn_readfirstlane(*p.n_active); Mrows = na < Mrows ? na : Mrows; }
    if (grow(0) >= Mrows) return;                   // nothing in this tile (its smallest row index is row 0's)
    constexpr int NTHREADS = WAVES * 64;
    const int brow = lane & 31, hf = lane >> 5;     // this lane's batch row within a row group of the tile, and its feature half

    CLK(0)
    // FINISH: this wave finishes the tile's envs 4*wave .. 4*wave+3 after the heads (WAVES == 8).  Everything that step
    // reads from the tree is requested after the staging barrier (paths, valid masks) and after the second layer (the
    // nodes those paths name), so that the round trips run under the GEMMs without holding up the staging (vmcnt
    // retires in order): lane j holds the j-th node of the env's recorded descent (bl_sim_expand's path) and,
    // separately, node slot `lane` of the env (T <= 64) for the q range.
#ifndef BLM_FIN_PASSES
#define BLM_FIN_PASSES 2    // 64-row tiles: a wave's eight envs as two passes of four (2) or phase by phase all at once (1) -- measurement switch
#endif
    constexpr int FPASS = RG > 1 ? BLM_FIN_PASSES : 1;
    constexpr int EPA = 4 * RG;                   // envs a wave finishes in all: tile rows EPA * wave .. EPA * wave + EPA - 1 ...
    constexpr int EPW = EPA / FPASS;              // ... EPW of them at a time
    // Held across the GEMMs, where registers are scarce (ring 96 + accumulators 32 + x 16 + ...): the per-env scalars are wave
    // uniform (SGPRs), and the envs' valid bits share one register.
    int fb[EPA], fleaf[EPA], fmover[EPA], flen[EPA];
    int fnode[EPA];                               // lane j's node of env e's path (left untouched until after the last GEMM:
                                                  // any use would wait for the load, and with it for the weights in flight)
    uint32_t fvbits = 0;                          // bit 2e + k: valid[lane + 64 k] of env e
    int fterm[EPA], fn[EPA], fallN[EPA];
    uint32_t frew[EPA], fw[EPA], fallW[EPA];
    float16v acc[RG * NT];
    Ring<NT, RD> rg;
    // stage the observation tile as 32-bit words (K0 is even: 2 planes per cell), zero-padding K0 -> K0pad and rows >= M.
    // The observation loads are issued BEFORE the first weight fragments: vmcnt retires in order, so the other way round
    // the staging barrier would also wait for the (cold, just evicted by the search kernel) weights.
    {
        const int wpr = p.K0pad >> 1, wvalid = p.K0 >> 1;           // words per staged row / per real row
        const uint32_t* src = (const uint32_t*)p.obs;               // row r starts at word r * K0 / 2 (K0 even)
        uint32_t* dst = (uint32_t*)(R0 + ROWS * ld * par0);
        constexpr int RPT = ROWS / (NTHREADS / 32);                 // rows per thread: a thread owns column words w, w+32, ...
        constexpr int WMAX = 4;                                     // ... up to 4 of them in registers (K0pad <= 256)
        const int c = tid & 31, r_first = tid >> 5;
        if (wpr <= 32 * WMAX) {
            uint32_t st[RPT][WMAX];
#pragma unroll
            for (int i = 0; i < RPT; i++) {
                const int r = r_first + i * (NTHREADS / 32);
#pragma unroll
                for (int k = 0; k < WMAX; k++) {
                    const int w = c + 32 * k;
                    st[i][k] = (w < wvalid && grow(r) < Mrows) ? src[(long)grow(r) * wvalid + w] : 0u;
                }
            }
            CLK(50)
            gemm_prefetch<NT, RD>(rg, p.w0, p.K0pad, wave * PASSES * NT, NT);
            CLK(51)
#pragma unroll
            for (int i = 0; i < RPT; i++) {
                const int r = r_first + i * (NTHREADS / 32);
#pragma unroll
                for (int k = 0; k < WMAX; k++) { const int w = c + 32 * k; if (w < wpr) dst[r * (ld >> 1) + w] = st[i][k]; }
            }
        } else {
            gemm_prefetch<NT, RD>(rg, p.w0, p.K0pad, wave * PASSES * NT, NT);
            for (int r = r_first; r < ROWS; r += NTHREADS / 32)
                for (int w = c; w < wpr; w += 32)
                    dst[r * (ld >> 1) + w] = (w < wvalid && grow(r) < Mrows) ? src[(long)grow(r) * wvalid + w] : 0u;
        }
    }
    // every layer's bias: (D + 1) x W halves behind the two activation buffers, read back by the epilogues
    uint16_t* BiasL = R0 + p.bias_off;
    {
        const uint32_t* b0w = (const uint32_t*)p.b0; const uint32_t* bbw = (const uint32_t*)p.bb;
        uint32_t* dstw = (uint32_t*)BiasL;
        const int hw = W >> 1, nb = (p.D + 1) * hw;
        // four words per thread and trip, all four loads in flight before the first store (as `for (i = tid; i < nb; i += NTHREADS)
        // dstw[i] = ...` every trip was load, s_waitcnt vmcnt(0), write -- for 512x4 three memory round trips in a row ahead of the
        // staging barrier, each also waiting for the weight fragments requested before it)
        for (int i0 = tid; i0 < nb; i0 += 4 * NTHREADS) {
            uint32_t bw[4];
#pragma unroll
            for (int j = 0; j < 4; j++) { const int i = i0 + j * NTHREADS; bw[j] = i < nb ? (i < hw ? b0w[i] : bbw[i - hw]) : 0u; }
#pragma unroll
            for (int j = 0; j < 4; j++) { const int i = i0 + j * NTHREADS; if (i < nb) dstw[i] = bw[j]; }
        }
    }
    CLK(52)
    __syncthreads();
    CLK(53)
    // a zero in a VGPR the compiler cannot see through: added to wave-uniform addresses below so that the loaded values stay vector
    // loads (with a provably uniform address it would move each loaded value to an SGPR at once, and the s_waitcnt for that also
    // waits for whatever was requested before)
    int zero;
    asm volatile("v_mov_b32 %0, 0" : "=v"(zero));
    if constexpr (FINISH) {
#pragma unroll
        for (int e = 0; e < EPA; e++) {
            const int b = grow(EPA * wave + e);
            fb[e] = b < Mrows ? b : -1;
            const long bb = (b < Mrows ? b : 0) + zero;
            const int16_t* path = f.path + bb * (f.T + 2);
            flen[e] = path[0];
            fnode[e] = lane < f.T ? (int)path[1 + lane] : 0;
            // valid(a) = both observation planes of cell a empty (hex/__init__.py:154-159), read from the tile staged above
            // (word a of a row = the two f16 planes of cell a; layer 1's epilogue is the first to overwrite this buffer).
            // Loading f.valid here instead cost 4.5k cycles: the compiler tests the byte at once, and the s_waitcnt vmcnt(0)
            // it needs for that also waits for the cold weight fragments requested just before.
            const uint32_t* stg = (const uint32_t*)(R0 + ROWS * ld * par0) + (EPA * wave + e) * (ld >> 1);
            if (lane < f.A && stg[lane] == 0u) fvbits |= 1u << (2 * e);
            if (lane + 64 < f.A && stg[lane + 64] == 0u) fvbits |= 2u << (2 * e);
        }
    }
    CLK(1)

    // intake Linear, then the ReZero blocks (networks.py:17-18).  A wave owns the same columns of the same rows in every
    // layer.  (Through round 5 its slice of the residual stream stayed in registers and only relu(x) went through LDS; since
    // round 6 x itself lives in LDS -- rectified as the next GEMM reads it, re-read by the owner's epilogue -- which frees 16 RG
    // registers per wave across the GEMMs for the weight ring.)
    // a wave's 64 output features are exactly one k block of the next layer: it can start on it before the layer barrier
#ifdef BL_MLP_NO_OWN_FIRST
    constexpr bool OWNC = false;
#else
    constexpr bool OWNC = PASSES == 1 && NT == 2;                // then WAVES * 64 == W
#endif
    const bool own = OWNC;
    // heads' Linears on the un-rectified neck.  The NHpad/32 output tiles are few (3 for 9x9), so each tile's K range is split
    // over two waves (2t: the even k blocks, 2t+1: the odd ones); the odd wave's partial sums go through LDS (R(1) is free
    // by then) to the even one, which adds them in a fixed order and stores.  Wave w's set contains k block w -- the 64
    // features it wrote itself in the last layer -- so with `own` it starts there, like the layers do, ahead of the barrier.
    const int htiles = p.NHpad / 32, KBh = W >> 6, nbh = KBh >> 1;
    auto hblk = [&](int j) {                                      // j-th k block of this wave's unit
        const int j0 = own ? (wave >> 1) : 0;
        const int jj = j0 + j < nbh ? j0 + j : j0 + j - nbh;
        return (wave & 1) + 2 * jj;
    };
    auto heads_prefetch = [&](half8 (&be)[4], half8 (&bo)[4]) {
        if (wave < 2 * htiles) {
            const uint16_t* bt = p.wh + (long)(wave >> 1) * KBh * 2048 + lane * 8;
#pragma unroll
            for (int s2 = 0; s2 < 4; s2++) be[s2] = *(const half8*)(bt + hblk(0) * 2048 + s2 * 512);
            if (nbh > 1) {
#pragma unroll
                for (int s2 = 0; s2 < 4; s2++) bo[s2] = *(const half8*)(bt + hblk(1) * 2048 + s2 * 512);
            }
        }
    };
    // What the finish step reads from the tree for env e of this wave (its path's nodes fnode[e] were requested after the staging
    // and have long landed).  Leaf, mover and the path's length are direct loads by env: requested here with the rest (held across
    // the GEMMs they were 3 x EPA more registers).  Lanes beyond the path's length hold whatever the path row held before --
    // clamped to a slot of the env, read, and never used.
    auto finish_request = [&](const int e) __attribute__((always_inline)) {
            const long envbase = (long)(fb[e] < 0 ? 0 : fb[e]) * f.T;
            const long bb = (fb[e] < 0 ? 0 : fb[e]) + zero;
            fleaf[e] = f.leaves[bb]; fmover[e] = f.leaf_seats[bb]; flen[e] = f.path[bb * (f.T + 2)];
            const int nd = fnode[e] < 0 ? 0 : (fnode[e] < f.T ? fnode[e] : f.T - 1);
            const long i = envbase + nd;
            fterm[e] = f.terminal[i]; fn[e] = f.n[i];
            frew[e] = *(const uint32_t*)(f.rewards + i * 2); fw[e] = *(const uint32_t*)(f.w + i * 2);
            const long t = envbase + (lane < f.T ? lane : 0);
            fallW[e] = *(const uint32_t*)(f.w + t * 2); fallN[e] = f.n[t];
    };
    auto layer = [&](const int l, auto first_c, auto last_c) __attribute__((always_inline)) {
        constexpr bool FIRST = decltype(first_c)::value, LAST = decltype(last_c)::value;
        constexpr int WC = WAVES * PASSES * NT * 32;                  // == W (mlp_launch picks the instantiation by it)
        const uint16_t* Wl = l == 0 ? p.w0 : p.wb + (long)(l - 1) * W * W;
        const int Kl = l == 0 ? p.K0pad : W;
        half2v al2 = {(f16)0.f, (f16)0.f};
        // (read through the scalar cache: as a vector load its s_waitcnt vmcnt(0) at the top of every layer also waited for the
        // whole weight prefetch instead of its first block)
        if (l > 0) { const f16 a = (f16)((const __attribute__((address_space(4))) float*)p.alphas)[l - 1]; al2[0] = a; al2[1] = a; }   // torch casts the f32 0-dim parameter to f16
        const uint16_t* Rin = R0 + ROWS * ld * ((l + par0) & 1);
        uint16_t* Rn = R0 + ROWS * ld * ((l + 1 + par0) & 1);
#pragma unroll
        for (int ps = 0; ps < PASSES; ps++) {
            const int tile0 = (wave * PASSES + ps) * NT, n0 = tile0 * 32;
            if constexpr (HANDW) {
                // the intake's K is the board's (a run-time loop on compiler-placed waits); the body layers' block loop is
                // straight-line code on the hand-placed weight stream
                if constexpr (FIRST) gemm_run<NT, RD, 0, false, RG, true>(rg, Rin, ld, Wl, Kl, tile0, NT, acc, 0, false, false);
                else gemm_run<NT, RD, WC / 64, OWNC, RG, true, true>(rg, Rin, ld, Wl, Kl, tile0, NT, acc, own ? wave : 0, false, true);
            } else
                gemm_run<NT, RD, 0, false, RG, true>(rg, Rin, ld, Wl, Kl, tile0, NT, acc, (own && l > 0) ? wave : 0, own && l > 0, l > 0);
            CLK(2 + 3 * l)
            if constexpr (FINISH && LAST) {
                // what the finish step reads from the tree (the paths requested after the staging have long landed) is requested
                // here, after the last GEMM: the last epilogue and the heads hide the trip, and the GEMMs above do not carry
                // these 24 registers
#pragma unroll
                for (int e = 0; e < EPW; e++) finish_request(e);      // the first pass's envs; the second pass's after this epilogue (registers)
            }
            // next weights in flight before the epilogue: this layer's next pass, or the next layer's first pass
            if (ps + 1 < PASSES) gemm_prefetch<NT, RD>(rg, Wl, Kl, tile0 + NT, NT);
            else if constexpr (!LAST) gemm_prefetch<NT, RD, HANDW>(rg, p.wb + (long)l * W * W, W, wave * PASSES * NT, NT, own ? wave : 0);
            else heads_prefetch(rg.b[0][0], rg.b[1][0]);        // the heads' first two k blocks travel under the last epilogue
            // The layer's bias and this wave's slice of the residual stream come from LDS (round 6; they used to be held in 16 + 16 RG
            // registers across the GEMM): x_l sits in the buffer the GEMM has just read, x_{l+1} goes to the other.  The RG * NT * 4
            // groups of 4 features are software-pipelined BLM_EPI_AHEAD deep: group i+AHEAD's two ds_reads go out before group i's
            // arithmetic and store.  Written group by group the compiler serialises them -- it cannot tell the buffers apart, and the
            // registers it reads into are the accumulators it has just consumed -- and every group pays an LDS round trip: 8 x 270 cycles
            // per layer against 2 x 270 + the arithmetic.
            {
                constexpr int NG = RG * NT * 4, AH = BLM_EPI_AHEAD;
                auto grp_f0 = [&](int i) { const int t = (i >> 2) % NT, g = i & 3; return n0 + 32 * t + 8 * g + 4 * hf; };   // 4 consecutive features of batch row ...
                auto grp_row = [&](int i) { return 32 * (i / (4 * NT)) + brow; };                                              // ... `32 rgi + brow`
                uint2 xq[AH + 1], bq[AH + 1];
#pragma unroll
                for (int i = 0; i < AH && i < NG; i++) {
                    xq[i] = l == 0 ? make_uint2(0, 0) : *(const uint2*)(Rin + grp_row(i) * ld + grp_f0(i));
                    bq[i] = *(const uint2*)(BiasL + l * W + grp_f0(i));
                }
#pragma unroll
                for (int i = 0; i < NG; i++) {
                    if (i + AH < NG) {
                        xq[(i + AH) % (AH + 1)] = l == 0 ? make_uint2(0, 0) : *(const uint2*)(Rin + grp_row(i + AH) * ld + grp_f0(i + AH));
                        bq[(i + AH) % (AH + 1)] = *(const uint2*)(BiasL + l * W + grp_f0(i + AH));
                    }
                    const int ai = i >> 2, g = i & 3;             // accumulator tile rgi * NT + t, its g-th group of four
                    const float a4[4] = {acc[ai][4 * g], acc[ai][4 * g + 1], acc[ai][4 * g + 2], acc[ai][4 * g + 3]};
                    uint2 xo, ro;
                    rezero4(a4, bq[i % (AH + 1)], xq[i % (AH + 1)], al2, l == 0, xo, ro);
                    *(uint2*)(Rn + grp_row(i) * ld + grp_f0(i)) = xo;          // x itself: the next GEMM rectifies as it reads, the heads read the neck
                }
            }
        }
        CLK(3 + 3 * l)
        if (!own) __syncthreads();                       // else the barrier sits inside the next layer's GEMM (or the heads'), after its first block
        CLK(4 + 3 * l)
    };
    // the last layer is its own copy of the body: it prefetches the heads' weights instead of a next layer's (one loop body
    // doing either keeps both sets of registers alive around the back edge)
    if constexpr (HANDW) {
        if (p.D == 0) layer(0, std::true_type{}, std::true_type{});
        else {
            layer(0, std::true_type{}, std::false_type{});
            for (int l = 1; l < p.D; l++) layer(l, std::false_type{}, std::false_type{});
            layer(p.D, std::false_type{}, std::true_type{});
        }
    } else {
        for (int l = 0; l < p.D; l++) layer(l, std::false_type{}, std::false_type{});
        layer(p.D, std::false_type{}, std::true_type{});
    }
    if constexpr (FINISH && RG > 1) {
#pragma unroll
        for (int e = EPW; e < EPA; e++) finish_request(e);            // consumed after the first pass of the finish
    }
    const uint16_t* X = R0;                                        // the neck (par0 makes the last layer write R(0))
    float* Part = (float*)(R0 + ROWS * ld);                       // [row group][unit][16][64] f32, 4 KiB per unit and row group
    for (int round = 0; round * WAVES < 2 * htiles; round++) {    // same trip count for every wave: uniform barriers
        const int u = round * WAVES + wave;
        const bool active = u < 2 * htiles;
        const int t0 = active ? (u >> 1) : 0, khalf = u & 1;
        const bool first = round == 0;                            // round 0's first two blocks are in flight already
        float16v hacc[RG];
#pragma unroll
        for (int g = 0; g < RG; g++) for (int i = 0; i < 16; i++) hacc[g][i] = 0.f;
        const uint16_t* arow = X + brow * ld + 32 * hf;
        const uint16_t* bt = p.wh + (long)t0 * KBh * 2048 + lane * 8;
        auto loadb = [&](half8 (&b)[4], int kb) {
#pragma unroll
            for (int s2 = 0; s2 < 4; s2++) b[s2] = *(const half8*)(bt + kb * 2048 + s2 * 512);
        };
        auto step = [&](half8 (&b)[4], int kb) {
            half8 a[RG][4];
#pragma unroll
            for (int g = 0; g < RG; g++) {
#pragma unroll
                for (int s2 = 0; s2 < 4; s2++) a[g][s2] = *(const half8*)(arow + g * 32 * ld + kb * 64 + 8 * s2);
            }
#pragma unroll
            for (int s2 = 0; s2 < 4; s2++) {
#pragma unroll
                for (int g = 0; g < RG; g++) hacc[g] = __builtin_amdgcn_mfma_f32_32x32x16_f16(b[s2], a[g][s2], hacc[g], 0, 0, 0);
            }
        };
        half8 (&be)[4] = rg.b[0][0], (&bo)[4] = rg.b[1][0];           // two k blocks in flight, alternating (the layers' ring is free)
        if (active) {
            if (!first) { loadb(be, hblk(0)); if (nbh > 1) loadb(bo, hblk(1)); }
            step(be, hblk(0));
            if (2 < nbh) loadb(be, hblk(2));
        }
        if (first && own) __syncthreads();                        // the last layer's barrier, behind the wave's own block
        if (active) {
            for (int j = 1; j < nbh; j += 2) {
                step(bo, hblk(j));
                if (j + 2 < nbh) loadb(bo, hblk(j + 2));
                if (j + 1 < nbh) {
                    step(be, hblk(j + 1));
                    if (j + 3 < nbh) loadb(be, hblk(j + 3));
                }
            }
        }
        // no barrier here: Part lies in R(1), which nobody has read since the last layer's barrier (the heads read the neck in R(0))
        if (active && khalf == 1) {
#pragma unroll
            for (int rgi = 0; rgi < RG; rgi++) {
#pragma unroll
                for (int i = 0; i < 16; i++) Part[((rgi * htiles + t0) * 16 + i) * 64 + lane] = hacc[rgi][i];
            }
        }
        __syncthreads();
        uint16_t* Out = (uint16_t*)(Part + RG * htiles * 16 * 64);  // [ROWS][NHpad] f16 staging for coalesced stores
        if (active && khalf == 0) {
#pragma unroll
            for (int rgi = 0; rgi < RG; rgi++) {
#pragma unroll
            for (int g = 0; g < 4; g++) {
                const int f0 = 32 * t0 + 8 * g + 4 * hf;              // 4 consecutive output features of row `32 rgi + brow`
                uint16_t o[4];
#pragma unroll
                for (int j = 0; j < 4; j++) o[j] = f2h(hacc[rgi][4 * g + j] + Part[((rgi * htiles + t0) * 16 + 4 * g + j) * 64 + lane] + h2f(p.bh[f0 + j]));
                *(uint2*)(Out + (32 * rgi + brow) * p.NHpad + f0) = make_uint2(o[0] | ((uint32_t)o[1] << 16), o[2] | ((uint32_t)o[3] << 16));
            }
            }
        }
        __syncthreads();
    }
    const uint16_t* Out = (const uint16_t*)(Part + RG * htiles * 16 * 64);
    CLK(34)
    if constexpr (!FINISH) {
        // coalesced stores: a wave writes one row's NH-1 policy outputs as consecutive halves
        for (int r = wave; r < ROWS; r += WAVES) {
            if (grow(r) < Mrows) {
                for (int fi = lane; fi < p.NH - 1; fi += 64) p.policy[(long)grow(r) * (p.NH - 1) + fi] = Out[r * p.NHpad + fi];
            }
        }
        if (tid < ROWS && grow(tid) < Mrows) p.value[grow(tid)] = Out[tid * p.NHpad + p.NH - 1];
    } else {
        // ---- bl_sim_finish's work for this wave's four envs, operation for operation as in bl_sim.hip:
        // sim_finish_kernel (heads with torch's order; backup cuda.cu:205-236; transition_q's range), but PHASE by phase
        // across the four envs so that their cross-lane exchanges and LDS trips overlap instead of queueing.
        const int A = f.A, T = f.T, Wsm = f.Wsm, iters = f.iters;
        const bool two = iters > 1;
        uint32_t nmin = 0, vmax = 0;
        // the wave's envs EPW at a time (RG passes): what a pass holds in registers is what the RG = 1 kernel holds
        auto finish_envs = [&](auto h_c) __attribute__((always_inline)) {
        constexpr int E0 = EPW * decltype(h_c)::value;
#pragma unroll
        for (int e = 0; e < EPW; e++) flen[E0 + e] = __builtin_amdgcn_readfirstlane(flen[E0 + e]);
        long envbase[EPW]; int leaf[EPW];
        float ev[EPW][2], mx[EPW], sum[EPW];
        // policy head: lane l < Wsm holds actions l + it * Wsm; A <= 128 here, so that is register `it` of the valid bytes
        // the heads' outputs of the four envs: all twelve LDS reads in flight at once, selected afterwards (as `cond ? h2f(Out[..]) : -inf`
        // under `if (lane < A)` every read sat behind its own EXEC branch with a wait: eight LDS round trips in a row; a read past a
        // row's NH entries -- lane + Wsm up to 127 -- stays inside Out + the scratch behind it and is discarded)
        uint16_t o0[EPW], o1[EPW], ov[EPW];
#pragma unroll
        for (int e = 0; e < EPW; e++) {
            const int r = EPA * wave + E0 + e;
            o0[e] = Out[r * p.NHpad + lane]; o1[e] = Out[r * p.NHpad + lane + Wsm]; ov[e] = Out[r * p.NHpad + p.NH - 1];
        }
#pragma unroll
        for (int e = 0; e < EPW; e++) {
            envbase[e] = (long)(fb[E0 + e] < 0 ? 0 : fb[E0 + e]) * T;
            leaf[e] = __builtin_amdgcn_readfirstlane(fleaf[E0 + e]);
            const bool k0 = lane < Wsm && lane < A && ((fvbits >> (2 * (E0 + e))) & 1u), k1 = lane < Wsm && two && lane + Wsm < A && ((fvbits >> (2 * (E0 + e) + 1)) & 1u);
            ev[e][0] = k0 ? h2f(o0[e]) : -INFINITY; ev[e][1] = k1 ? h2f(o1[e]) : -INFINITY;
            mx[e] = two ? ((ev[e][0] > ev[e][1]) ? ev[e][0] : ev[e][1]) : ev[e][0];
        }
        CLK(E0 ? 41 : 54)
#define BLM_MAXSTEP(OFF) if (Wsm > OFF) { _Pragma("unroll") for (int e = 0; e < EPW; e++) { const float o = xor_lane<OFF>(mx[e]); mx[e] = (mx[e] < o) ? o : mx[e]; } }
        BLM_MAXSTEP(32) BLM_MAXSTEP(16) BLM_MAXSTEP(8) BLM_MAXSTEP(4) BLM_MAXSTEP(2) BLM_MAXSTEP(1)
#undef BLM_MAXSTEP
        CLK(E0 ? 42 : 55)
#pragma unroll
        for (int e = 0; e < EPW; e++) { sum[e] = 0.f; sum[e] += expf(ev[e][0] - mx[e]); if (two) sum[e] += expf(ev[e][1] - mx[e]); }
        CLK(E0 ? 43 : 56)
#define BLM_SUMSTEP(OFF) if (Wsm > OFF) { _Pragma("unroll") for (int e = 0; e < EPW; e++) sum[e] = sum[e] + xor_lane<OFF>(sum[e]); }
        BLM_SUMSTEP(32) BLM_SUMSTEP(16) BLM_SUMSTEP(8) BLM_SUMSTEP(4) BLM_SUMSTEP(2) BLM_SUMSTEP(1)
#undef BLM_SUMSTEP
        CLK(E0 ? 44 : 57)
        uint16_t vb0[EPW], vb1[EPW];
        uint16_t lb[EPW][2];
#pragma unroll
        for (int e = 0; e < EPW; e++) {
            const float lsum = logf(sum[e]);
            lb[e][0] = f2h(ev[e][0] - mx[e] - lsum); lb[e][1] = f2h(ev[e][1] - mx[e] - lsum);
        }
        // the leaf's compacted policy row (bl_device.h: compact_store): pi = exp_table[logit bits] of the kept actions.  The
        // gathers go out FIRST -- ahead of the stores below, so that waiting for them later does not also wait for the
        // stores' acknowledgements (vmcnt retires in order) -- and are consumed after the backup scan, which hides their trip.
        CLK(E0 ? 45 : 58)
        float pi[EPW][2];
        bool in[EPW][2];
#pragma unroll
        for (int e = 0; e < EPW; e++) {
            in[e][0] = f.cpi && fb[E0 + e] >= 0 && lane < Wsm && lane < A; in[e][1] = f.cpi && fb[E0 + e] >= 0 && two && lane < Wsm && lane + Wsm < A;
            pi[e][0] = in[e][0] ? f.exp_table[lb[e][0]] : 0.f; pi[e][1] = in[e][1] ? f.exp_table[lb[e][1]] : 0.f;
        }
        CLK(E0 ? 46 : 59)
#pragma unroll
        for (int e = 0; e < EPW; e++) {
            const int r = EPA * wave + E0 + e;
            if (fb[E0 + e] >= 0 && lane < Wsm) {
                uint16_t* dst = f.logits + (envbase[e] + leaf[e]) * A;
                if (lane < A) dst[lane] = lb[e][0];
                if (two && lane + Wsm < A) dst[lane + Wsm] = lb[e][1];
            }
            // value head
            const uint16_t tv = f2h(tanhf(h2f(ov[e])));
            const int mover = __builtin_amdgcn_readfirstlane(fmover[E0 + e]);
            vb0[e] = (mover == 0) ? tv : (uint16_t)(tv ^ 0x8000u); vb1[e] = (uint16_t)(vb0[e] ^ 0x8000u);
            if (fb[E0 + e] >= 0 && lane == 0) { f.v[(envbase[e] + leaf[e]) * 2] = vb0[e]; f.v[(envbase[e] + leaf[e]) * 2 + 1] = vb1[e]; }
        }
        CLK(E0 ? 20 : 35)
        CLK(E0 ? 21 : 36)
        // backup (cuda.cu:205-236), leaf -> root: node j's value is v_j = (terminal_j ? 0 : v_{j+1}) + r_j with v_len the
        // leaf evaluation.  Every lane applies that step to its right neighbour's current value at once; after k rounds
        // the last k nodes of the path are final (each re-evaluation reads a final neighbour and recomputes the same
        // sum), so maxlen rounds finish all four envs' paths -- two DPP instructions per round and env instead of a
        // scalar walk.  w_j = rn16(w_j + rn16(v_j)) then needs no order at all.
        float x0[EPW], x1[EPW], r0[EPW], r1[EPW];
        int maxlen = 0;
#pragma unroll
        for (int e = 0; e < EPW; e++) {
            x0[e] = h2f(vb0[e]); x1[e] = h2f(vb1[e]);                          // lanes >= len keep the leaf evaluation
            r0[e] = h2f((uint16_t)frew[E0 + e]); r1[e] = h2f((uint16_t)(frew[E0 + e] >> 16));
            if (fb[E0 + e] < 0) flen[E0 + e] = 0;
            maxlen = flen[E0 + e] > maxlen ? flen[E0 + e] : maxlen;
        }
        // In Hex a reward and `terminal` only ever sit on the LAST node of a path (a descent stops at a terminal node), so every
        // interior node just passes its successor's value on, plus its own reward +0.0 (which turns a -0 into +0, once):
        // v_j = v_leaf' + 0.0f for j < len - 1, v_leaf' = (terminal ? 0 : v) + r at the last node.  That is two instructions
        // instead of `len` rounds (25 rounds on the deepest paths); the general scan remains for paths that do carry
        // something on an interior node.
        bool plain = true;
#pragma unroll
        for (int e = 0; e < EPW; e++) plain = plain && !__any(lane < flen[E0 + e] - 1 && (fterm[E0 + e] != 0 || frew[E0 + e] != 0u));
        if (plain) {
#pragma unroll
            for (int e = 0; e < EPW; e++) {
                if (flen[E0 + e] > 0) {
                    const float l0 = (fterm[E0 + e] ? 0.f : h2f(vb0[e])) + r0[e], l1 = (fterm[E0 + e] ? 0.f : h2f(vb1[e])) + r1[e];   // right in lane len - 1
                    const float b0 = readlane_f(l0, flen[E0 + e] - 1), b1 = readlane_f(l1, flen[E0 + e] - 1);
                    if (lane < flen[E0 + e] - 1) { x0[e] = b0 + 0.f; x1[e] = b1 + 0.f; }
                    else if (lane == flen[E0 + e] - 1) { x0[e] = b0; x1[e] = b1; }
                }
            }
        } else {
            for (int k = 0; k < maxlen; k++) {
#pragma unroll
                for (int e = 0; e < EPW; e++) {
                    const float n0 = dpp_next_lane(h2f(vb0[e]), x0[e]), n1 = dpp_next_lane(h2f(vb1[e]), x1[e]);
                    if (lane < flen[E0 + e]) { x0[e] = (fterm[E0 + e] ? 0.f : n0) + r0[e]; x1[e] = (fterm[E0 + e] ? 0.f : n1) + r1[e]; }
                }
            }
        }
        float w0[EPW], w1[EPW];
#pragma unroll
        for (int e = 0; e < EPW; e++) {
            w0[e] = h2f(f2h(h2f((uint16_t)fw[E0 + e]) + h2f(f2h(x0[e]))));
            w1[e] = h2f(f2h(h2f((uint16_t)(fw[E0 + e] >> 16)) + h2f(f2h(x1[e]))));
        }
        if (f.cpi) {
#pragma unroll
            for (int e = 0; e < EPW; e++) {
                const long rowbase = (envbase[e] + leaf[e]) * A;
                const bool k0 = in[e][0] && pi[e][0] != 0.f, k1 = in[e][1] && pi[e][1] != 0.f;
                const unsigned long long m0 = __ballot(k0), m1 = __ballot(k1);
                const unsigned long long below = (1ull << lane) - 1ull;
                const int c0 = __builtin_popcountll(m0);
                if (k0) { const int j = __builtin_popcountll(m0 & below); f.cpi[rowbase + j] = pi[e][0]; f.cca[rowbase + j] = 0xffff0000u | (uint32_t)lane; }
                if (k1) { const int j = c0 + __builtin_popcountll(m1 & below); f.cpi[rowbase + j] = pi[e][1]; f.cca[rowbase + j] = 0xffff0000u | (uint32_t)(lane + Wsm); }
                if (fb[E0 + e] >= 0 && lane == 0) f.nk[envbase[e] + leaf[e]] = (int16_t)(c0 + __builtin_popcountll(m1));
            }
        }
        CLK(E0 ? 22 : 37)
        // stores, and the q range over all T slots of each env with the path's nodes replaced by their new statistics.  Lane t
        // still holds slot t's old (w, n) and lane j the j-th path node's new ones: a slot's old q counts unless the slot is on
        // the path (a 64-bit mask, OR-reduced over the lanes with DPP), a path node's new q always does.  No LDS involved.
        uint32_t wnew[EPW]; int nnew[EPW];
#pragma unroll
        for (int e = 0; e < EPW; e++) {
            wnew[e] = (uint32_t)f2h(w0[e]) | ((uint32_t)f2h(w1[e]) << 16);
            nnew[e] = (int)(int16_t)(fn[E0 + e] + 2);                            // n += 1 once per seat (cuda.cu:230), int16 wrap kept
            const bool onp = lane < flen[E0 + e];
            const int fnode_e = fnode[E0 + e];
            if (onp) {
                const long i = envbase[e] + fnode_e;
                *(uint32_t*)(f.w + i * 2) = wnew[e];
                f.n[i] = (int16_t)nnew[e];
            }
            const uint32_t lo = (onp && fnode_e < 32) ? (1u << fnode_e) : 0u, hi = (onp && fnode_e >= 32) ? (1u << (fnode_e - 32)) : 0u;
            const uint32_t mlo = wave_or_u32(lo), mhi = wave_or_u32(hi);
            const bool replaced = ((lane < 32 ? mlo >> lane : mhi >> (lane - 32)) & 1u) != 0;
            if (fb[E0 + e] >= 0) {
                if (lane < T && !replaced) {
                    const float den = (float)fallN[E0 + e] + 1.e-4f;
                    const uint32_t e0 = enc(h2f((uint16_t)fallW[E0 + e]) / den), e1 = enc(h2f((uint16_t)(fallW[E0 + e] >> 16)) / den);
                    nmin = max(nmin, max(~e0, ~e1)); vmax = max(vmax, max(e0, e1));
                }
                if (onp) {
                    const float den = (float)nnew[e] + 1.e-4f;
                    const uint32_t e0 = enc(w0[e] / den), e1 = enc(w1[e] / den);
                    nmin = max(nmin, max(~e0, ~e1)); vmax = max(vmax, max(e0, e1));
                }
            }
        }
        CLK(E0 ? 23 : 38)
        };
        finish_envs(std::integral_constant<int, 0>{});
        if constexpr (FPASS > 1) finish_envs(std::integral_constant<int, 1>{});
        static_assert(RG <= 2, "finish_envs is instantiated for two passes");
        // max is associative: one reduction and one conditional atomic pair for all of the wave's envs
        nmin = wave_max_u32(nmin); vmax = wave_max_u32(vmax);
        if (lane == 0 && fb[0] >= 0) {
            uint32_t* q = f.qrange + BLM_QSTRIDE * ((blockIdx.x * WAVES + wave) % BLM_QSLOTS);
            // unconditional: with one pair per wave (1024 per launch over 64 slots) the atomics are cheap, and a checking
            // load first would put a round trip at the very end of every workgroup
            // words in memory = the unsigned codes XOR 0x80000000, compared SIGNED (bl_device.h: BL_QBIAS; include/boardlaw_amd.h)
            atomicMax((int*)q, (int)(nmin ^ 0x80000000u));
            atomicMax((int*)(q + 1), (int)(vmax ^ 0x80000000u));
        }
    }
    CLK(40)
}


}  // namespace blmlp

#ifdef BL_MLP_CLK
extern "C" int bl_mlp_debug_clk(long long* out) { return hipMemcpyFromSymbol(out, HIP_SYMBOL(blmlp::g_debug_clk), 64 * 8) == hipSuccess ? 0 : -3; }
#endif
// rows: rows per workgroup -- 32, 64 (widths 256 and 512 only: two 64-row activation buffers of a wider network do not fit the
// LDS), or 0 = by the batch: a workgroup's time is its weight stream through the CU's L1, the same for 32 rows and for 64, so 64
// pay as soon as the 32-row tiles outnumber the chip's 256 CUs (every CU then takes several, one after the other: 32768 rows of
// 512x4 146 -> ... us, profiles/r06_mlp_rows.txt), and lose below that (half of the CUs would idle).  Same bits either way.
static int mlp_launch(const blmlp::Params& p, const blmlp::FinArgs* fin, bl_stream_t stream, int rows = 0) {
    using namespace blmlp;
    const int W = p.W, NHpad = p.NHpad, M = p.M;
    if (rows != 0 && rows != 32 && rows != 64) return BL_EINVAL;
    const bool can64 = W == 256 || W == 512;
    if (rows == 64 && !can64) return BL_ETOOBIG;
    // two activation buffers; the heads keep the neck in the first and stage split-K partials + outputs after it
    // (+ 2 KiB of slack per wave behind them: the finish epilogue's unconditional reads run past a row's end)
    // ... and behind both every layer's bias, (D + 1) x W halves
    auto acts_for = [&](int r) {
        const size_t buf = (size_t)r * (W + 8) * 2;
        const size_t staging = (size_t)(r / 32) * (NHpad / 32) * 16 * 64 * 4 + (size_t)r * NHpad * 2 + (fin ? 8 * 4 * 128 * 4 : 0);
        return buf + (staging > buf ? staging : buf);
    };
    const size_t biases = (size_t)(p.D + 1) * W * 2;
    if (rows == 0) rows = (can64 && (M + 31) / 32 > 256 && acts_for(64) + biases <= 160 * 1024) ? 64 : 32;
    const int RGn = rows / 32;
    const size_t lds = acts_for(rows) + biases;
    if (lds > 160 * 1024) return BL_ETOOBIG;
    Params pp = p;
    pp.bias_off = (int)(acts_for(rows) / 2);
    const dim3 grid(p.xcd_rows ? 8 * ((M + 8 * rows - 1) / (8 * rows)) : (M + rows - 1) / rows);
    hipStream_t hs = (hipStream_t)stream;
    const FinArgs f = fin ? *fin : FinArgs{};
    // above the 64 KiB default the limit has to be raised per kernel (gfx950 has 160 KiB per CU)
#define BL_MLP_LAUNCH1(NT, PASSES, WAVES, FIN, RD, RG)                                                                     \
    {                                                                                                                  \
        static size_t raised[64];                                                                                      \
        if (!bl_raise_lds_limit((const void*)mlp_kernel<NT, PASSES, WAVES, FIN, RD, RG>, lds, raised)) return BL_ELAUNCH;  \
        hipLaunchKernelGGL((mlp_kernel<NT, PASSES, WAVES, FIN, RD, RG>), grid, dim3(WAVES * 64), lds, hs, pp, f);          \
    }
#define BL_MLP_LAUNCH(NT, PASSES, WAVES, RD, RG) { if (fin) BL_MLP_LAUNCH1(NT, PASSES, WAVES, true, RD, RG) else BL_MLP_LAUNCH1(NT, PASSES, WAVES, false, RD, RG) }
    // 8 waves (two per SIMD) from W = 256 up: while one wave waits for its weight fragments the other issues MFMAs
    switch (W / 128) {
        case 1: if (fin) return BL_ETOOBIG; BL_MLP_LAUNCH1(1, 1, 4, false, 3, 1) break;      // the epilogue assumes 8 waves
        case 2: if (RGn == 2) BL_MLP_LAUNCH(1, 1, 8, 3, 2) else BL_MLP_LAUNCH(1, 1, 8, 3, 1) break;
        // 64 rows: accumulators 64 + residual 32 + activations 32 registers; the weight ring keeps two k blocks (one in flight per
        // wave is enough now that every block feeds twice the MFMAs)
        case 4: if (RGn == 2) BL_MLP_LAUNCH(2, 1, 8, BLM_RD64, 2) else BL_MLP_LAUNCH(2, 1, 8, BLM_RD32, 1) break;
        case 6: BL_MLP_LAUNCH(1, 3, 8, 3, 1) break;
        case 8: BL_MLP_LAUNCH(2, 2, 8, 3, 1) break;
        default: return BL_ETOOBIG;
    }
#undef BL_MLP_LAUNCH
#undef BL_MLP_LAUNCH1
    return hipGetLastError() == hipSuccess ? BL_OK : BL_ELAUNCH;
}

extern "C" int bl_mlp_forward_f16(const void* obs, int M, int K0, const void* w0, const void* b0, const void* wb,
                                  const void* bb, const float* alphas, const void* wh, const void* bh, int W, int D,
                                  int K0pad, int NH, int NHpad, void* policy_out, void* value_out, bl_stream_t stream) {
    using namespace blmlp;
    if (!policy_out || !value_out) return BL_EINVAL;
    if (int rc = mlp_check(obs, M, K0, w0, b0, wb, bb, alphas, wh, bh, W, D, K0pad, NH, NHpad)) return rc;
    Params p{(const uint16_t*)obs, (const uint16_t*)w0, (const uint16_t*)b0, (const uint16_t*)wb, (const uint16_t*)bb, alphas,
             (const uint16_t*)wh, (const uint16_t*)bh, (uint16_t*)policy_out, (uint16_t*)value_out, M, K0, K0pad, W, D, NH, NHpad, 0, nullptr};
    return mlp_launch(p, nullptr, stream);
}

extern "C" int bl_sim_infer_finish(const bl_search_t* s, int sim, const int16_t* leaves, const void* obs, const uint8_t* valid,
                                   const int32_t* leaf_seats, const void* w0, const void* b0, const void* wb, const void* bb,
                                   const float* alphas, const void* wh, const void* bh, int W, int D, int K0pad, int NHpad,
                                   bl_stream_t stream) {
    using namespace blmlp;
    if (!s || !s->logits || !s->v || !s->w || !s->n || !s->rewards || !s->terminal || !s->qrange || !s->path || !leaves ||
        !valid || !leaf_seats || s->B <= 0 || s->T <= 0 || s->boardsize <= 0 || sim < 1 || sim >= s->T) return BL_EINVAL;
    const int A = s->boardsize * s->boardsize, M = s->B, K0 = 2 * A, NH = A + 1;
    if (s->T > 64 || A > 128 || W < 256) return BL_ETOOBIG;     // the epilogue keeps a whole env in one wave's registers
    if (int rc = mlp_check(obs, M, K0, w0, b0, wb, bb, alphas, wh, bh, W, D, K0pad, NH, NHpad)) return rc;
    const int xcd_rows = s->tune.mlp_no_xcd ? 0 : 1;     // tiles of same-XCD envs (see mlp_kernel)
    Params p{(const uint16_t*)obs, (const uint16_t*)w0, (const uint16_t*)b0, (const uint16_t*)wb, (const uint16_t*)bb, alphas,
             (const uint16_t*)wh, (const uint16_t*)bh, nullptr, nullptr, M, K0, K0pad, W, D, NH, NHpad, xcd_rows, s->n_active};
    int np2 = 1; while (np2 < A) np2 *= 2;
    const int Wsm = np2 < 64 ? np2 : 64;
    FinArgs f{(uint16_t*)s->logits, (uint16_t*)s->v, (uint16_t*)s->w, s->n, (const uint16_t*)s->rewards, s->terminal, s->path,
              s->qrange + (long)BLM_QSLOTS * BLM_QSTRIDE * (sim + 1), leaves, leaf_seats, valid, s->T, A, Wsm, np2 / Wsm,
              (s->cpi && s->cca && s->nk) ? s->cpi : nullptr, s->cca, s->nk, s->exp_table};
    return mlp_launch(p, &f, stream, s->tune.mlp_rows);
}

