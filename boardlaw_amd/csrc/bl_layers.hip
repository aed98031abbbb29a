// bl_layers.hip -- the leaf-evaluation network with a launch per Linear (bl_mlp_layers_f16: wide networks on small batches, BASELINE
// config 4's per-GPU shape) and its two one-launch forms with in-kernel hand-offs (bl_mlp_layers_persist_f16, bl_mlp_layers_xcd_f16:
// bit-identical, measured, not faster).  Split out of bl_mlp.hip in round 6; the GEMM pieces are bl_gemm.h's.
#include "bl_gemm.h"

namespace blmlp {


// ------------------------------------------------------------------------------------------------------------------
// One Linear (+ ReZero tail) per launch, every layer split over the whole chip: the plan for networks whose weights are too
// many to stream through each workgroup's L1 (mlp_kernel's time is 2 * weights bytes / 64 B/clk per 32-row workgroup: 1024x8 is
// 17.9 MB = 130 us however small the batch), i.e. wide networks on small batches -- 13x13 / 1024 envs / 1024x8 (BASELINE
// config 4's per-GPU shape).  A workgroup of 4 waves takes 32 rows x 128 output features: the input rows (relu applied on the
// way in, as the next block's relu) are staged in LDS once, every wave streams the fragment-major weights of its 32 features
// (the same packing as above) into v_mfma_f32_32x32x16_f16, and the epilogue is the fused kernel's (rezero4: torch's
// rounding points).  The residual stream x lives in global memory between launches (two buffers, ping-pong: a workgroup
// writes features other workgroups still read as inputs).  Per launch a CU moves 64 KiB of activations + 256 KiB of weights
// instead of the whole network.
// ------------------------------------------------------------------------------------------------------------------
struct LayerArgs {
    const uint16_t* X; int ldx, Kvalid, Kpad; int relu_in;        // input rows (M, Kvalid) f16, row stride ldx; zero padded to Kpad
    const uint16_t* Wp; const uint16_t* bias; int N;              // packed (N/32 tiles x Kpad/64 blocks), bias (N); N % 32 == 0
    const uint16_t* Xres; const float* alpha;                     // residual rows (M, N) stride N and its ReZero alpha, or null: y itself
    uint16_t* Y; int ldy;                                         // body: x' (M, N)
    uint16_t* policy; uint16_t* value; int NH;                    // heads (Y == null): features 0..NH-2 -> policy (M, NH-1), NH-1 -> value (M)
    int M;
    int ncol;                                                     // column groups of 128 features per row tile (set by the launcher)
    int xcd;                                                      // 1: row tile r on XCD r % 8 (set by the launcher)
    int sc1_out;                                                  // 1: Y is read by other workgroups of the SAME launch (layers_persist_kernel)
};

struct NoWait { static constexpr bool early = false; __device__ __forceinline__ void operator()() const {} };

// 8-byte relaxed agent-scope atomics = `global_load/store_dwordx2 ... sc1`: the loads bypass the CU's L1, the stores go through to
// memory -- a valid payload form for a cross-workgroup hand-off WITHOUT fences (MI355X_MICROARCH.md, inter-workgroup visibility:
// "8-B agent atomics both sides"); an agent-scope release/acquire pair instead writes back / invalidates whole caches (3.4-8 us per
// hand-off, and polling with acquire loads cuts the chip's bandwidth -- the first version of the kernel below: 447 us per forward).
__device__ __forceinline__ uint2 ld_sc1(const void* p) {
    const unsigned long long v = __hip_atomic_load((const unsigned long long*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return make_uint2((uint32_t)v, (uint32_t)(v >> 32));
}
__device__ __forceinline__ void st_sc1(void* p, uint2 v) {
    __hip_atomic_store((unsigned long long*)p, (unsigned long long)v.x | ((unsigned long long)v.y << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// One workgroup's share of one Linear: rows 32 * rowtile .., features 128 * colgroup ..  Contains one workgroup barrier; waves
// without a tile and lanes whose row is beyond M leave after it.
// LRG: row groups of 32 rows per workgroup (every weight fragment then feeds LRG MFMAs: see gemm_run).
template <int RD, int KBC, typename PRE, int LRG = 1>
__device__ __forceinline__ void layer_body(const LayerArgs& a, const int rowtile, const int colgroup, uint16_t* R, PRE pre) {
    const int ld = a.Kpad + 8;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int row0 = rowtile * 32 * LRG, tile = colgroup * 4 + wave, ntiles = a.N >> 5;
    const int brow = lane & 31, hf = lane >> 5;
    Ring<1, RD> rg;
    float16v acc[LRG];
    const bool active = tile < ntiles;
    // EARLY (the persistent kernel): the first weight fragments are requested BEFORE `pre` waits for the previous layer -- weights
    // do not depend on it -- and the input rows after
    constexpr bool EARLY = PRE::early;      // also: this layer's rows come from / go to other workgroups of THIS launch (sc1 payload)
    if (EARLY) { if (active) gemm_prefetch<1, RD>(rg, a.Wp, a.Kpad, tile, 1); pre(); }
    // the input rows first, then the first weight fragments (vmcnt retires in order: see mlp_kernel)
    const half2v z2 = {(f16)0.f, (f16)0.f};
    if ((a.ldx & 7) == 0 && (a.Kvalid & 7) == 0) {
        // 8 threads per row, 16 bytes each: one 128-byte line per row and step; all of a thread's chunks in flight at once
        // (Kpad <= 1024: at most 16), the weight prefetch right behind them
        const int j = tid & 7;
        constexpr int NB = 16;
        uint4 v[LRG][NB];
#pragma unroll
        for (int rr = 0; rr < LRG; rr++) {
            const int r = (tid >> 3) + 32 * rr;
            const bool rok = row0 + r < a.M;
            const uint16_t* src = a.X + (long)(row0 + r) * a.ldx + 8 * j;
#pragma unroll
            for (int i = 0; i < NB; i++) {
                v[rr][i] = make_uint4(0, 0, 0, 0);
                if (rok && 64 * i + 8 * j < a.Kvalid) {
                    if constexpr (EARLY) { const uint2 lo = ld_sc1(src + 64 * i), hi = ld_sc1(src + 64 * i + 4); v[rr][i] = make_uint4(lo.x, lo.y, hi.x, hi.y); }
                    else v[rr][i] = *(const uint4*)(src + 64 * i);
                }
            }
        }
        if (!EARLY && active) gemm_prefetch<1, RD>(rg, a.Wp, a.Kpad, tile, 1);
#pragma unroll
        for (int rr = 0; rr < LRG; rr++) {
            uint16_t* dst = R + ((tid >> 3) + 32 * rr) * ld + 8 * j;
#pragma unroll
            for (int i = 0; i < NB; i++) {
                if (64 * i < a.Kpad) {
                    uint4 w = v[rr][i];
                    if (a.relu_in) {
                        w.x = __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(half2v, w.x), z2));
                        w.y = __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(half2v, w.y), z2));
                        w.z = __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(half2v, w.z), z2));
                        w.w = __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(half2v, w.w), z2));
                    }
                    *(uint4*)(dst + 64 * i) = w;
                }
            }
        }
    } else {
        // rows that are only 4-byte aligned (the observation: 2 planes per cell): 32-bit words.  Wave w stages rows 8w .. 8w+7,
        // lane l their words l, l + 64, ...; a row's loads all in flight at once (Kpad <= 1024: at most 8 per lane and row)
        const int wpr = a.Kpad >> 1, wvalid = a.Kvalid >> 1;
        constexpr int WMAX = 8;
        uint32_t v[8 * LRG][WMAX];
#pragma unroll
        for (int i = 0; i < 8 * LRG; i++) {
            const int r = wave * 8 * LRG + i;
            const bool rok = row0 + r < a.M;
            const uint32_t* src = (const uint32_t*)a.X + ((long)(row0 + r) * a.ldx >> 1);
#pragma unroll
            for (int k = 0; k < WMAX; k++) {
                const int w = lane + 64 * k;
                v[i][k] = 0u;
                if (64 * k < wvalid) { if (rok && w < wvalid) v[i][k] = src[w]; }
            }
        }
        if (!EARLY && active) gemm_prefetch<1, RD>(rg, a.Wp, a.Kpad, tile, 1);      // behind the rows' loads: vmcnt retires in order
#pragma unroll
        for (int i = 0; i < 8 * LRG; i++) {
            const int r = wave * 8 * LRG + i;
#pragma unroll
            for (int k = 0; k < WMAX; k++) {
                const int w = lane + 64 * k;
                if (w < wpr) {
                    uint32_t x = v[i][k];
                    if (a.relu_in) x = __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(half2v, x), z2));
                    ((uint32_t*)R)[r * (ld >> 1) + w] = x;
                }
            }
        }
    }
    __syncthreads();
    if (!active) return;
    const int n0 = tile * 32;
    uint2 biasr[4], xold[LRG][4];
#pragma unroll
    for (int g = 0; g < 4; g++) biasr[g] = *(const uint2*)(a.bias + n0 + 8 * g + 4 * hf);
#pragma unroll
    for (int rgi = 0; rgi < LRG; rgi++) {
        const long grow = row0 + 32 * rgi + brow;
#pragma unroll
        for (int g = 0; g < 4; g++) {
            xold[rgi][g] = make_uint2(0, 0);
            if (a.Xres && grow < a.M) { if constexpr (EARLY) xold[rgi][g] = ld_sc1(a.Xres + grow * a.N + n0 + 8 * g + 4 * hf); else xold[rgi][g] = *(const uint2*)(a.Xres + grow * a.N + n0 + 8 * g + 4 * hf); }
        }
    }
    half2v al2 = {(f16)0.f, (f16)0.f};
    if (a.Xres) { const f16 al = (f16)((const __attribute__((address_space(4))) float*)a.alpha)[0]; al2[0] = al; al2[1] = al; }
    gemm_run<1, RD, KBC, false, LRG>(rg, R, ld, a.Wp, a.Kpad, tile, 1, acc);
#pragma unroll
    for (int rgi = 0; rgi < LRG; rgi++) {
        const long grow = row0 + 32 * rgi + brow;
        if (grow >= a.M) continue;
#pragma unroll
        for (int g = 0; g < 4; g++) {
            const int f0 = n0 + 8 * g + 4 * hf;
            const float a4[4] = {acc[rgi][4 * g], acc[rgi][4 * g + 1], acc[rgi][4 * g + 2], acc[rgi][4 * g + 3]};
            uint2 xo, ro;
            rezero4(a4, biasr[g], xold[rgi][g], al2, a.Xres == nullptr, xo, ro);
            if (a.Y) { if (a.sc1_out) st_sc1(a.Y + grow * a.ldy + f0, xo); else *(uint2*)(a.Y + grow * a.ldy + f0) = xo; }
            else {
                const uint16_t o[4] = {(uint16_t)xo.x, (uint16_t)(xo.x >> 16), (uint16_t)xo.y, (uint16_t)(xo.y >> 16)};
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    if (f0 + j < a.NH - 1) a.policy[grow * (a.NH - 1) + f0 + j] = o[j];
                    else if (f0 + j == a.NH - 1) a.value[grow] = o[j];
                }
            }
        }
    }
}

// RD - 1 k blocks of 4 KiB in flight per wave; KBC = Kpad / 64 when it is one of the body widths' (the block loop is then
// straight-line code and every MFMA waits for exactly its fragment -- with ONE wave per SIMD there is nobody to hide a
// drained weight stream behind, unlike in mlp_kernel), 0 = any (the intake).
#ifndef BLM_LAYER_RG
#define BLM_LAYER_RG 1          // row groups per workgroup of layer_kernel (measurement switch: 2 = 64-row tiles)
#endif
template <int RD, int KBC>
__global__ void __launch_bounds__(256) layer_kernel(LayerArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    uint16_t* R = (uint16_t*)smem;
    // Which 32 rows x 128 features this workgroup takes.  Workgroup i runs on XCD i % 8.  All column groups of a row tile are placed
    // on ONE XCD (row tile r on XCD r % 8), and the same way in every layer's launch: the rows a workgroup stages were written by
    // workgroups of its own XCD in the previous launch and are still in that XCD's L2, instead of coming from the seven others
    // through the fabric.  Placement is a speed matter only.
    // Measured (13x13, 1024x8, tools/ab_layers.sh): 1024 rows 94.6 -> 80.5 us per forward; 256 rows 71.3 -> 78.5 us -- there
    // every XCD then streams ALL the weights for its one row tile, where the plain mapping (column group x on XCD x, any row
    // tile) lets an XCD fetch only its eighth of them: the launcher picks by the row count.
    int rowtile, colgroup;
    if (a.xcd) {
        const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
        rowtile = xcd + 8 * (slot / a.ncol); colgroup = slot % a.ncol;
        if (rowtile * 32 * BLM_LAYER_RG >= a.M) return;
    } else {
        rowtile = blockIdx.x / a.ncol; colgroup = blockIdx.x % a.ncol;
    }
    layer_body<RD, KBC, NoWait, BLM_LAYER_RG>(a, rowtile, colgroup, R, NoWait{});
}

// ------------------------------------------------------------------------------------------------------------------
// All Linears of the forward in ONE launch (round 4): the launch-per-Linear plan above sits on a floor of ~7.5 us per layer
// whatever the batch (a dependent launch boundary, a cold trip for the rows, the first weight fragments' trip), ten times per
// forward at 1024x8.  But a row tile's next layer depends on THAT ROW TILE's previous layer only: the eight workgroups that own
// its column groups (all on one XCD, see layer_kernel).  So a workgroup keeps its (row tile, column group) through all layers,
// and between two layers it
//   * requests the next layer's first weight fragments (they depend on nothing),
//   * publishes its rows -- every wave a release fence at agent scope, a workgroup barrier, one atomic add on the row tile's
//     counter for that layer -- and waits until the counter says all column groups of the row tile have published (one lane
//     polls with acquire loads, then the workgroup barrier: the CU's vector L1 is invalidated by the acquire, plain loads of
//     the rows follow; /opt/skills/guides recipe G16),
//   * stages the rows and runs the layer as before.
// Waiting only for one's own row tile is also what makes buffer re-use safe: layer l + 1 overwrites the rows layer l read, and
// only after every reader of that row tile has finished layer l.
// Co-residency: a workgroup spins for peers of its row tile, so those must get CUs.  The grid is at most 256 workgroups of 4
// waves and 66 KiB of LDS (two fit a CU), the peers of a row tile lie within 64 consecutive workgroup indices, and every spin is
// BOUNDED: if a wait runs out (another process hogging the chip) the kernel raises `error` in the counter block and finishes
// with whatever it has -- wrong results the host can see (networks.Inference checks the word), never a hung GPU.
// The last workgroup to finish zeroes the counters, so a captured forward replays without a reset launch.
// ------------------------------------------------------------------------------------------------------------------
#define BLM_MAX_LAYERS 12
#define BLM_SPIN_LIMIT (1 << 21)
struct PersistArgs {
    LayerArgs layer[BLM_MAX_LAYERS];
    int nlayers, rowtiles, ncol;      // ncol: column groups of the body layers = the grid's (the heads use fewer)
    int* counters;                    // [rowtiles][nlayers] arrivals + one word: workgroups finished
    int* error;                       // set to 1 when a bounded wait ran out
    int xcd;
    int local;                        // 1: the XCD-local protocol (layers_persist_kernel, round 5): counters = [rowtiles][nlayers][8] flags +
                                      // 8 tickets + one word: workgroups finished
};

__global__ void __launch_bounds__(256) zero_words_kernel(int* p, int n) {
    for (int i = threadIdx.x; i < n; i += 256) p[i] = 0;
}

// layers_persist_kernel's wait between two layers: all column groups of the row tile have published the previous layer.  ONE lane
// polls with relaxed loads (acquire loads in a poll loop invalidate the L1 every time), then the workgroup barrier.
struct RowTileWait {
    static constexpr bool early = true;
    const int* counter; int want; int* error;
    __device__ __forceinline__ void operator()() const {
        if (threadIdx.x == 0) {
            int polls = 0;
            while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
                if (++polls > BLM_SPIN_LIMIT) { __hip_atomic_store(error, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
                __builtin_amdgcn_s_sleep(1);
            }
        }
        __syncthreads();
    }
};

// The XCD-local protocol's wait: lanes 0 .. want-1 of wave 0 each poll ONE flag -- the word column group c of this row tile stored
// (plain, after its rows) when it had finished the previous layer -- with L1-bypassing loads: producer and poller share an XCD, so
// the word and the rows behind it are in the L2 both sides use.
struct RowTileFlags {
    static constexpr bool early = true;
    const int* flags; int want; int* error;
    __device__ __forceinline__ void operator()() const {
        if ((int)threadIdx.x < want) {
            int polls = 0;
            while (__hip_atomic_load(flags + threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) {
                if (++polls > BLM_SPIN_LIMIT) { __hip_atomic_store(error, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
                __builtin_amdgcn_s_sleep(1);
            }
        }
        __syncthreads();
    }
};

// Round 5, `p.local`: the hand-off between two layers stays inside ONE XCD's L2.  Round 4's protocol above is placement-independent:
// rows as write-through (`sc1`) stores, a device-scope counter per row tile and layer, L1-bypassing loads -- which drops every row
// from the producer's L2 and brings it back at the cross-XCD rate even when, as always, the eight workgroups of a row tile DO share
// an XCD (91.9 us per forward against 80.8 for a launch per Linear).  Here a workgroup ASKS where it runs (s_getreg HW_REG_XCC_ID),
// takes a ticket from that XCD's counter and works on row tile xcd + 8 (ticket / ncol), column group ticket % ncol: the peers of
// a row tile share an L2 by construction, not by an assumed dispatch order.  So the rows are plain stores (they stay in that L2),
// `s_waitcnt vmcnt(0)` = acknowledged by it, a barrier, then the workgroup's flag word as a plain store; the peers poll the flags
// and read the rows with L1-bypassing loads, served by the same L2.  No device-scope atomic and no write-through on the path.
// What it needs of the dispatcher: every XCD receives (its row tiles) x ncol workgroups of the grid -- the hardware deals a grid's
// workgroups to the XCDs in turn, and the grid is 8 x ceil(rowtiles / 8) x ncol.  An XCD that received more leaves the surplus
// idle and one that received fewer cannot finish a row tile: its peers' bounded waits run out and raise `error` (wrong results
// the host sees, never a hang), exactly as when a peer is kept off the chip.
template <int RDB, int KBCB>
__global__ void __launch_bounds__(256) layers_persist_kernel(const PersistArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    uint16_t* R = (uint16_t*)smem;
    const int tid = threadIdx.x;
    int rowtile, colgroup;
    if (p.local) {
        __shared__ int ticket[2];
        if (tid == 0) {
            int xcc;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
            xcc &= 7;
            ticket[0] = xcc;
            ticket[1] = __hip_atomic_fetch_add(p.counters + (long)p.rowtiles * p.nlayers * 8 + xcc, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
        rowtile = ticket[0] + 8 * (ticket[1] / p.ncol); colgroup = ticket[1] % p.ncol;
    } else if (p.xcd) {
        const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
        rowtile = xcd + 8 * (slot / p.ncol); colgroup = slot % p.ncol;
    } else {
        rowtile = blockIdx.x / p.ncol; colgroup = blockIdx.x % p.ncol;
    }
    int* done = p.counters + (p.local ? (long)p.rowtiles * p.nlayers * 8 + 8 : (long)p.rowtiles * p.nlayers);
    if (p.local) {
        if (rowtile < p.rowtiles) {
            int* mine = p.counters + (long)rowtile * p.nlayers * 8;
            for (int l = 0; l < p.nlayers; l++) {
                const LayerArgs& a = p.layer[l];
                const RowTileFlags wait{mine + (l > 0 ? l - 1 : 0) * 8, l > 0 ? p.layer[l - 1].ncol : 0, p.error};
                if (colgroup < a.ncol) {
                    if (l == 0) layer_body<3, 0>(a, rowtile, colgroup, R, NoWait{});
                    else layer_body<RDB, KBCB>(a, rowtile, colgroup, R, wait);
                }
                if (l + 1 < p.nlayers) {
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's rows are in the XCD's L2 ...
                    __syncthreads();                                      // ... and so are the other waves' (and nobody still reads R)
                    if (tid == 0 && colgroup < a.ncol) __hip_atomic_store(mine + l * 8 + colgroup, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
            }
        }
    } else if (rowtile < p.rowtiles) {
        int* mine = p.counters + (long)rowtile * p.nlayers;
        for (int l = 0; l < p.nlayers; l++) {
            const LayerArgs& a = p.layer[l];
            const RowTileWait wait{mine + (l > 0 ? l - 1 : 0), l > 0 ? p.layer[l - 1].ncol : 0, p.error};
            if (colgroup < a.ncol) {
                if (l == 0) layer_body<3, 0>(a, rowtile, colgroup, R, NoWait{});
                else layer_body<RDB, KBCB>(a, rowtile, colgroup, R, wait);
            }                                                         // (a workgroup beyond the layer's column groups -- the heads' -- has nothing to do)
            if (l + 1 < p.nlayers) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's sc1 stores have been acknowledged ...
                __syncthreads();                                      // ... and so have the other waves' (and nobody still reads R)
                if (tid == 0 && colgroup < a.ncol) __hip_atomic_fetch_add(mine + l, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
    // the last workgroup out zeroes the counters for the next forward (write-through stores: the words were last written by other
    // XCDs and are next read by them; all 256 threads: a write-through dword is one fabric write each)
    __shared__ int last;
    __syncthreads();
    if (tid == 0) last = __hip_atomic_fetch_add(done, 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == (int)gridDim.x - 1;
    __syncthreads();
    if (last) {
        const long words = p.local ? (long)p.rowtiles * p.nlayers * 8 + 8 : (long)p.rowtiles * p.nlayers;
        for (long i = tid; i < words; i += 256) __hip_atomic_store(p.counters + i, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) __hip_atomic_store(done, 0, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
}

}  // namespace blmlp

extern "C" int bl_mlp_layers_f16(const void* obs, int M, int K0, const void* w0, const void* b0, const void* wb,
                                 const void* bb, const float* alphas, const void* wh, const void* bh, int W, int D,
                                 int K0pad, int NH, int NHpad, void* scratch, void* policy_out, void* value_out, bl_stream_t stream) {
    using namespace blmlp;
    if (!policy_out || !value_out || !scratch) return BL_EINVAL;
    if (int rc = mlp_check(obs, M, K0, w0, b0, wb, bb, alphas, wh, bh, W, D, K0pad, NH, NHpad)) return rc;
    if ((K0 & 1) != 0) return BL_EINVAL;
    hipStream_t hs = (hipStream_t)stream;
    uint16_t* buf[2] = {(uint16_t*)scratch, (uint16_t*)scratch + (size_t)M * W};
    const dim3 rows((M + 32 * BLM_LAYER_RG - 1) / (32 * BLM_LAYER_RG));
    int rc = BL_OK;
    auto launch = [&](LayerArgs a, int Kpad) {
        a.ncol = (a.N / 32 + 3) / 4;
        a.xcd = M >= 512;               // row tiles pinned to XCDs once there are enough of them (see layer_kernel)
        const dim3 grid(a.xcd ? 8 * ((rows.x + 7) / 8) * a.ncol : rows.x * a.ncol);
        const size_t l = (size_t)32 * BLM_LAYER_RG * (Kpad + 8) * 2;
#define BL_LAYER_LAUNCH(RD, KBC)                                                                                                  \
        {                                                                                                                         \
            static size_t raised[64];                                                                                             \
            if (!bl_raise_lds_limit((const void*)layer_kernel<RD, KBC>, l, raised)) { rc = BL_ELAUNCH; return; }                   \
            hipLaunchKernelGGL((layer_kernel<RD, KBC>), grid, dim3(256), l, hs, a);                                               \
        }
        switch (Kpad) {
            case 1024: BL_LAYER_LAUNCH(8, 16) break;
            case 768: BL_LAYER_LAUNCH(8, 12) break;
            case 512: BL_LAYER_LAUNCH(6, 8) break;
            case 256: BL_LAYER_LAUNCH(4, 4) break;
            case 384: BL_LAYER_LAUNCH(6, 6) break;      // 13x13's intake
            case 192: BL_LAYER_LAUNCH(3, 3) break;      // 9x9's intake
            default: BL_LAYER_LAUNCH(3, 0) break;
        }
#undef BL_LAYER_LAUNCH
    };
    // intake: x = Linear(obs)
    launch(LayerArgs{(const uint16_t*)obs, K0, K0, K0pad, 0, (const uint16_t*)w0, (const uint16_t*)b0, W, nullptr, nullptr,
                     buf[0], W, nullptr, nullptr, 0, M}, K0pad);
    // ReZero blocks: x' = x + alpha * Linear(relu(x))
    for (int l = 0; l < D; l++)
        launch(LayerArgs{buf[l & 1], W, W, W, 1, (const uint16_t*)wb + (size_t)l * W * W, (const uint16_t*)bb + (size_t)l * W, W,
                         buf[l & 1], alphas + l, buf[(l + 1) & 1], W, nullptr, nullptr, 0, M}, W);
    // heads on the un-rectified neck
    launch(LayerArgs{buf[D & 1], W, W, W, 0, (const uint16_t*)wh, (const uint16_t*)bh, NHpad, nullptr, nullptr, nullptr, 0,
                     (uint16_t*)policy_out, (uint16_t*)value_out, NH, M}, W);
    if (rc != BL_OK) return rc;
    return hipGetLastError() == hipSuccess ? BL_OK : BL_ELAUNCH;
}

static int layers_persist_launch(const void* obs, int M, int K0, const void* w0, const void* b0, const void* wb,
                                 const void* bb, const float* alphas, const void* wh, const void* bh, int W, int D,
                                 int K0pad, int NH, int NHpad, void* scratch, int* counters, int zero_first, int* error,
                                 void* policy_out, void* value_out, int local, bl_stream_t stream) {
    using namespace blmlp;
    if (!policy_out || !value_out || !scratch || !counters || !error) return BL_EINVAL;
    if (int rc = mlp_check(obs, M, K0, w0, b0, wb, bb, alphas, wh, bh, W, D, K0pad, NH, NHpad)) return rc;
    if ((K0 & 1) != 0) return BL_EINVAL;
    if (D + 2 > BLM_MAX_LAYERS || (W != 256 && W != 512 && W != 768 && W != 1024)) return BL_ETOOBIG;
    PersistArgs p;
    p.nlayers = D + 2; p.rowtiles = (M + 31) / 32; p.ncol = W / 128; p.counters = counters; p.error = error; p.xcd = local || M >= 512; p.local = local;
    const unsigned grid = p.xcd ? 8u * ((p.rowtiles + 7) / 8) * p.ncol : (unsigned)p.rowtiles * p.ncol;
    if (grid > 256) return BL_ETOOBIG;          // every workgroup must find a CU while its row tile's peers run: see layers_persist_kernel
    uint16_t* buf[2] = {(uint16_t*)scratch, (uint16_t*)scratch + (size_t)M * W};
    auto set = [&](int i, LayerArgs a) { a.ncol = (a.N / 32 + 3) / 4; a.xcd = p.xcd; a.sc1_out = !local && a.Y != nullptr; p.layer[i] = a; };
    set(0, LayerArgs{(const uint16_t*)obs, K0, K0, K0pad, 0, (const uint16_t*)w0, (const uint16_t*)b0, W, nullptr, nullptr, buf[0], W, nullptr, nullptr, 0, M});
    for (int l = 0; l < D; l++)
        set(1 + l, LayerArgs{buf[l & 1], W, W, W, 1, (const uint16_t*)wb + (size_t)l * W * W, (const uint16_t*)bb + (size_t)l * W, W,
                             buf[l & 1], alphas + l, buf[(l + 1) & 1], W, nullptr, nullptr, 0, M});
    set(D + 1, LayerArgs{buf[D & 1], W, W, W, 0, (const uint16_t*)wh, (const uint16_t*)bh, NHpad, nullptr, nullptr, nullptr, 0,
                         (uint16_t*)policy_out, (uint16_t*)value_out, NH, M});
    const int kmax = K0pad > W ? K0pad : W;
    const size_t lds = (size_t)32 * (kmax + 8) * 2;
    hipStream_t hs = (hipStream_t)stream;
    // fresh memory: the counters are zeroed by a launch of their own (a kernel, not a memset node: those replay only once in a captured
    // graph on this ROCm); the kernel leaves them zero, so a caller that keeps the block passes zero_first = 0 from the second call on
    if (zero_first) hipLaunchKernelGGL(zero_words_kernel, dim3(1), dim3(256), 0, hs, counters, local ? p.rowtiles * p.nlayers * 8 + 9 : p.rowtiles * p.nlayers + 1);
#define BL_PERSIST_LAUNCH(RD, KBC)                                                                                               \
    {                                                                                                                            \
        static size_t raised[64];                                                                                                \
        if (!bl_raise_lds_limit((const void*)layers_persist_kernel<RD, KBC>, lds, raised)) return BL_ELAUNCH;                     \
        hipLaunchKernelGGL((layers_persist_kernel<RD, KBC>), dim3(grid), dim3(256), lds, hs, p);                                 \
    }
    switch (W) {
        case 1024: BL_PERSIST_LAUNCH(8, 16) break;
        case 768: BL_PERSIST_LAUNCH(8, 12) break;
        case 512: BL_PERSIST_LAUNCH(6, 8) break;
        default: BL_PERSIST_LAUNCH(4, 4) break;
    }
#undef BL_PERSIST_LAUNCH
    return hipGetLastError() == hipSuccess ? BL_OK : BL_ELAUNCH;
}

extern "C" int bl_mlp_layers_persist_f16(const void* obs, int M, int K0, const void* w0, const void* b0, const void* wb,
                                         const void* bb, const float* alphas, const void* wh, const void* bh, int W, int D,
                                         int K0pad, int NH, int NHpad, void* scratch, int* counters, int zero_first, int* error,
                                         void* policy_out, void* value_out, bl_stream_t stream) {
    return layers_persist_launch(obs, M, K0, w0, b0, wb, bb, alphas, wh, bh, W, D, K0pad, NH, NHpad, scratch, counters, zero_first, error,
                                 policy_out, value_out, 0, stream);
}

extern "C" int bl_mlp_layers_xcd_f16(const void* obs, int M, int K0, const void* w0, const void* b0, const void* wb,
                                     const void* bb, const float* alphas, const void* wh, const void* bh, int W, int D,
                                     int K0pad, int NH, int NHpad, void* scratch, int* counters, int zero_first, int* error,
                                     void* policy_out, void* value_out, bl_stream_t stream) {
    return layers_persist_launch(obs, M, K0, w0, b0, wb, bb, alphas, wh, bh, W, D, K0pad, NH, NHpad, scratch, counters, zero_first, error,
                                 policy_out, value_out, 1, stream);
}

