// bl_hex.hip -- the Hex board kernels behind bl_hex_step / bl_hex_world_step / bl_hex_observe(_valid) (boardlaw/hex/cpp/cuda.cu:76-217,
// hex/__init__.py:148-195), in two forms: lanes per env (a group of 16-64 lanes per board; small batches) and tiles of 64 consecutive
// envs staged through LDS with the flood as a bit-board fill.  Split out of bl_kernels.hip in round 6.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <math.h>
#include <string.h>
#include <stdlib.h>
#include "../../include/boardlaw_amd.h"
#include "bl_device.h"
#include "bl_dispatch.h"

#pragma clang fp contract(off)

namespace bl {

// ------------------------------------------------------------------------------------------------------------------
// Hex.  Cell codes and rules: boardlaw/hex/cpp/cuda.cu:8-16,76-137; flood cuda.cu:18-74.
// ------------------------------------------------------------------------------------------------------------------

template <int G>
__global__ void __launch_bounds__(BL_WAVE) hex_step_kernel(uint8_t* board, const int32_t* seats, const int32_t* actions,
                                                           float* rewards, int B, int S) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int A = S * S, grp = threadIdx.x / G, gl = threadIdx.x % G;
    const int b = blockIdx.x * (BL_WAVE / G) + grp;
    const bool go = b < B;
    uint8_t* cells = (uint8_t*)smem + (size_t)grp * ((A + 15) & ~15);
    uint8_t* src = board + (long)b * A;
    if (go) for (int a = gl; a < A; a += G) cells[a] = src[a];
    __syncthreads();
    const int win = hex_step_group<G>(cells, S, go ? seats[b] : 0, go ? actions[b] : 0, go, gl);
    if (go) {
        for (int a = gl; a < A; a += G) src[a] = cells[a];
        if (gl == 0) { rewards[2 * b] = (float)win; rewards[2 * b + 1] = (float)(-win); }
    }
}

// Hex.step as one launch (hex/__init__.py:161-195 with reset=True): clone the board, step it, terminal = any reward > 0,
// wipe finished boards, pass the move to the other seat (seat 0 after a finished game).
template <int G>
__global__ void __launch_bounds__(BL_WAVE) hex_world_step_kernel(const uint8_t* board_in, const int32_t* seats_in, const void* actions,
                                                                 int actions_i64, uint8_t* board_out, int32_t* seats_out,
                                                                 float* rewards, uint8_t* terminal, int B, int S) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int A = S * S, grp = threadIdx.x / G, gl = threadIdx.x % G;
    const int b = blockIdx.x * (BL_WAVE / G) + grp;
    const bool go = b < B;
    uint8_t* cells = (uint8_t*)smem + (size_t)grp * ((A + 15) & ~15);
    const uint8_t* src = board_in + (long)b * A;
    if (go) for (int a = gl; a < A; a += G) cells[a] = src[a];
    int seat = 0, action = 0;
    if (go) { seat = seats_in[b]; action = actions_i64 ? (int)((const long long*)actions)[b] : ((const int32_t*)actions)[b]; }
    __syncthreads();
    const int win = hex_step_group<G>(cells, S, seat, action, go, gl);
    if (go) {
        uint8_t* dst = board_out + (long)b * A;
        for (int a = gl; a < A; a += G) dst[a] = win ? (uint8_t)0 : cells[a];
        if (gl == 0) {
            rewards[2 * b] = (float)win; rewards[2 * b + 1] = (float)(-win);
            terminal[b] = win != 0;
            seats_out[b] = win ? 0 : 1 - seat;
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// Round 5: the board kernels as HBM streams.  The kernels above give every env a group of lanes that fetches its own board byte by
// byte -- fine inside a search (one launch per move, 4096 boards) and 9-13 % of the HBM roofline on a million boards
// (bench.py `hex_kernels`).  Here a workgroup of 256 threads takes E = 256 / LPE CONSECUTIVE envs: their boards are one
// contiguous, 16-byte-aligned run of E * A bytes (E * A is a multiple of 16 for E = 64 and 16), which goes into LDS as 16-byte
// loads, is stepped there by LPE lanes per env (hex_step_group, wave-scope synchronisation: an env's lanes share a wave and the
// waves of the workgroup never wait for each other inside the flood), and leaves as 16-byte stores.  Same cell arithmetic, same
// results; `world` adds Hex.step's tail (hex/__init__.py:183-190: wipe finished boards, pass the move on).
// ------------------------------------------------------------------------------------------------------------------
// Column masks of an S x S board as bit sets over the cells (bit a = cell a, row-major): cells that have a left / a right neighbour.
struct HexMasks { uint32_t not_first[8], not_last[8]; };

// Hex's step on a board in LDS by FOUR lanes, the flood as a bit-board fill.  The sweeps of hex_step_group cost a pass over the board
// per propagation step with LDS round trips in it; here the four lanes of an env collect the cells of the mover's plain colour as a
// bit set (NW 32-bit words, lane l the cells 4 i + l, OR-ed over the quad with two DPP moves), grow the component from the new stone
// with shifts -- the six neighbours of cell a are a -+ S, a -+ 1 and a -+ (S - 1), the latter four behind the column masks -- until
// it stops growing, and write the label into its cells.  Same component (cuda.cu:18-74 relabels the 6-connected plain cells reachable
// from the new stone; the net effect is order-independent), same bytes.  A <= 32 NW.
template <int NW>
__device__ __forceinline__ int hex_step_quad(uint8_t* cells, int S, int seat, int action, bool go, int gl, const HexMasks& hm) {
    const int A = S * S;
    const float invS = 1.0f / (float)S;
    int label = 0, win = 0, start = 0, plain = 0;
    if (go && gl == 0) {
        const int qd = (int)(((float)action + 0.5f) * invS), rm = action - qd * S;
        const int row = seat == 0 ? qd : rm, col = seat == 0 ? rm : qd;   // white plays transposed, cuda.cu:88-91
        unsigned adj = 0;
        const int dr[6] = {-1, -1, 0, 0, +1, +1}, dc[6] = {0, +1, -1, +1, -1, 0};
#pragma unroll
        for (int k = 0; k < 6; k++) {
            const int r = row + dr[k], c = col + dc[k];
            int code;
            if (r < 0) code = TOP; else if (r >= S) code = BOT; else if (c < 0) code = LEFT; else if (c >= S) code = RIGHT;
            else code = cells[r * S + c];
            adj |= 1u << code;
        }
        const bool aT = adj & (1u << TOP), aB = adj & (1u << BOT), aL = adj & (1u << LEFT), aR = adj & (1u << RIGHT);
        if (seat) { if (aL && aR) win = -1; label = aL ? LEFT : (aR ? RIGHT : WHITE); plain = WHITE; }
        else      { if (aT && aB) win = +1; label = aT ? TOP : (aB ? BOT : BLACK); plain = BLACK; }
        start = row * S + col;
        if (label < TOP) cells[start] = (uint8_t)plain;                  // no flood: the plain colour (cuda.cu:134)
    }
    // lane 0 of the quad -> all four (quad_perm [0,0,0,0])
    label = dpp_i<0x00, 0xf>(label, label); win = dpp_i<0x00, 0xf>(win, win);
    plain = dpp_i<0x00, 0xf>(plain, plain); start = dpp_i<0x00, 0xf>(start, start);
    const bool flooding = go && label >= TOP;
    if (__any(flooding)) {
        uint32_t P[NW], M[NW];
#pragma unroll
        for (int w = 0; w < NW; w++) {
            // branch-free: all eight reads of a word in flight at once (a read beyond the board -- up to 32 NW - A bytes, into the
            // following envs' cells or the pad the launcher allocates behind the last one -- is masked out by `a < A`); as `if (...) bits |= ...` the compiler put every
            // cell behind its own EXEC branch with a wait per read: ten instructions and an LDS round trip per cell
            uint32_t bits = 0;
            uint8_t c[8];
#pragma unroll
            for (int i = 0; i < 8; i++) c[i] = cells[32 * w + 4 * i + gl];
#pragma unroll
            for (int i = 0; i < 8; i++) {
                const int a = 32 * w + 4 * i + gl;
                bits |= (uint32_t)((c[i] == (uint8_t)plain) & (a < A)) << (4 * i + gl);
            }
            if (!flooding) bits = 0;
            bits |= (uint32_t)dpp_i<0xB1, 0xf>(0, (int)bits);            // quad_perm [1,0,3,2]
            bits |= (uint32_t)dpp_i<0x4E, 0xf>(0, (int)bits);            // quad_perm [2,3,0,1]
            P[w] = bits;
            M[w] = (flooding && (start >> 5) == w) ? 1u << (start & 31) : 0u;
        }
        // shl / shr of an NW-word bit set by k in 1..31 (k = 0 only ever meets an empty set: S = 1)
        auto shl = [&](const uint32_t (&x)[NW], int k, uint32_t (&y)[NW]) {
#pragma unroll
            for (int w = 0; w < NW; w++) y[w] = __builtin_amdgcn_alignbit(x[w], w ? x[w - 1] : 0u, 32 - k);
        };
        auto shr = [&](const uint32_t (&x)[NW], int k, uint32_t (&y)[NW]) {
#pragma unroll
            for (int w = 0; w < NW; w++) y[w] = __builtin_amdgcn_alignbit(w + 1 < NW ? x[w + 1] : 0u, x[w], k);
        };
        for (int it = 0; it < A; it++) {
            // neighbours of the set M: L = (M with a right neighbour) << 1, R = (M with a left neighbour) >> 1, and the rows above and
            // below as ONE shift each -- cell a - S and a - S + 1 are (M | L) >> S, cell a + S and a + S - 1 are (M | R) << S
            uint32_t L[NW], R[NW], U[NW], D[NW], t[NW];
#pragma unroll
            for (int w = 0; w < NW; w++) { U[w] = M[w] & hm.not_last[w]; D[w] = M[w] & hm.not_first[w]; }
            shl(U, 1, L);
            shr(D, 1, R);
#pragma unroll
            for (int w = 0; w < NW; w++) { U[w] = M[w] | L[w]; D[w] = M[w] | R[w]; }
            shr(U, S, t);
            shl(D, S, U);
            uint32_t grew = 0;
#pragma unroll
            for (int w = 0; w < NW; w++) { const uint32_t nw = (L[w] | R[w] | t[w] | U[w]) & P[w] & ~M[w]; M[w] |= nw; grew |= nw; }
            if (!__any(grew != 0)) break;
        }
#pragma unroll
        for (int w = 0; w < NW; w++) {
            if (32 * (w + 1) <= A) {
                // a word that lies inside the board: every lane rewrites its eight cells, select(label, old value) -- two instructions a
                // cell and no EXEC juggling (a lane's cells are its own: nobody else writes them; envs that do not flood have M = 0)
                uint8_t c[8];
#pragma unroll
                for (int i = 0; i < 8; i++) c[i] = cells[32 * w + 4 * i + gl];
#pragma unroll
                for (int i = 0; i < 8; i++) cells[32 * w + 4 * i + gl] = ((M[w] >> (4 * i + gl)) & 1u) ? (uint8_t)label : c[i];
            } else if (32 * w < A) {
#pragma unroll
                for (int i = 0; i < 8; i++) {
                    const int a = 32 * w + 4 * i + gl;
                    if (a < A && ((M[w] >> (4 * i + gl)) & 1u)) cells[a] = (uint8_t)label;
                }
            }
        }
    }
    return win;
}

template <int NW>
__global__ void __launch_bounds__(256) hex_step_tile_kernel(const uint8_t* board_in, uint8_t* board_out, const int32_t* seats_in,
                                                            const void* actions, int actions_i64, int32_t* seats_out, float* rewards,
                                                            uint8_t* terminal, int B, int S, int world, const HexMasks hm) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int LPE = 4, E = 256 / LPE;
    const int A = S * S, tid = threadIdx.x;
    const long e0 = (long)blockIdx.x * E;
    const int nE = (int)((long)B - e0 < (long)E ? (long)B - e0 : (long)E);
    const long start = e0 * A;
    const int bytes = nE * A, n16 = bytes >> 4;
    uint8_t* all = (uint8_t*)smem;
    {
        const uint4* src = (const uint4*)(board_in + start);
        for (int i = tid; i < n16; i += 256) ((uint4*)all)[i] = src[i];
        for (int i = (n16 << 4) + tid; i < bytes; i += 256) all[i] = board_in[start + i];
    }
    const int env = tid / LPE, gl = tid % LPE;
    const bool go = env < nE;
    const long b = e0 + env;
    int seat = 0, action = 0;
    if (go) { seat = seats_in[b]; action = actions_i64 ? (int)((const long long*)actions)[b] : ((const int32_t*)actions)[b]; }
    __syncthreads();
    uint8_t* cells = all + (size_t)env * A;
    const int win = hex_step_quad<NW>(cells, S, seat, action, go, gl, hm);
    if (go) {
        if (world && win) for (int a = gl; a < A; a += LPE) cells[a] = 0;
        if (gl == 0) {
            rewards[2 * b] = (float)win; rewards[2 * b + 1] = (float)(-win);
            if (world) { terminal[b] = win != 0; seats_out[b] = win ? 0 : 1 - seat; }
        }
    }
    __syncthreads();
    {
        uint4* dst = (uint4*)(board_out + start);
        for (int i = tid; i < n16; i += 256) dst[i] = ((const uint4*)all)[i];
        for (int i = (n16 << 4) + tid; i < bytes; i += 256) board_out[start + i] = all[i];
    }
}

// observe (+ Hex.valid) the same way: 64 consecutive envs per workgroup, boards staged in LDS with 16-byte loads (the transposed
// read of a white mover's board is then an LDS gather, not an HBM one), every thread four consecutive cells at a time: two 16-byte
// stores of f32 planes and one 4-byte store of the mask (a 64-env run starts at a multiple of 64 cells: both are aligned).
__global__ void __launch_bounds__(256) hex_observe_tile_kernel(const uint8_t* board, const int32_t* seats, float* obs, uint8_t* valid, int B, int S) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int E = 64;
    const int A = S * S, tid = threadIdx.x;
    const long e0 = (long)blockIdx.x * E;
    const int nE = (int)((long)B - e0 < (long)E ? (long)B - e0 : (long)E);
    const long start = e0 * A;
    const int bytes = nE * A, n16 = bytes >> 4;
    uint8_t* all = (uint8_t*)smem;
    uint8_t* flips = all + (((size_t)E * A + 15) & ~(size_t)15);
    {
        const uint4* src = (const uint4*)(board + start);
        for (int i = tid; i < n16; i += 256) ((uint4*)all)[i] = src[i];
        for (int i = (n16 << 4) + tid; i < bytes; i += 256) all[i] = board[start + i];
        if (tid < nE) flips[tid] = seats[e0 + tid] == 1;
    }
    __syncthreads();
    const float invA = 1.0f / (float)A, invS = 1.0f / (float)S;
    const int quads = (bytes + 3) >> 2;
    float* obase = obs + start * 2;
    for (int q = tid; q < quads; q += 256) {
        float o[8];
        uint32_t vm = 0;
        // the quad's first cell by division (idx < 64 * 1024: exact in f32), the other three by stepping (cell, row, column) with wrap-around
        int idx = 4 * q;
        int e = (int)(((float)idx + 0.5f) * invA), a = idx - e * A;
        int i = (int)(((float)a + 0.5f) * invS), j = a - i * S;
        bool flip = flips[e] != 0;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            o[2 * k] = 0.f; o[2 * k + 1] = 0.f;
            if (idx < bytes) {
                const int c = all[e * A + (flip ? j * S + i : a)];
                const int color = c < 7 ? (0x1412 >> (2 * c)) & 3 : 2;  // color_of as a table: codes 1,3,4 -> 0; 2,5,6 -> 1; anything else -> 2
                if (color < 2) { if ((flip ? 1 - color : color) == 0) o[2 * k] = 1.f; else o[2 * k + 1] = 1.f; }
                else vm |= 1u << (8 * k);
            }
            idx++; a++; j++;
            if (j == S) { j = 0; i++; }
            if (a == A) { a = 0; i = 0; j = 0; e++; flip = (e < nE) && flips[e] != 0; }
        }
        if (4 * q + 3 < bytes) {
            ((float4*)obase)[2 * q] = make_float4(o[0], o[1], o[2], o[3]);
            ((float4*)obase)[2 * q + 1] = make_float4(o[4], o[5], o[6], o[7]);
            if (valid) *(uint32_t*)(valid + start + 4 * q) = vm;
        } else {
            for (int k = 0; k < 4 && 4 * q + k < bytes; k++) {
                obase[2 * (4 * q + k)] = o[2 * k]; obase[2 * (4 * q + k) + 1] = o[2 * k + 1];
                if (valid) valid[start + 4 * q + k] = (vm >> (8 * k)) & 1;
            }
        }
    }
}

// observe, cuda.cu:154-195: mover sees itself in channel 0, playing top-to-bottom.

__global__ void __launch_bounds__(256) hex_observe_kernel(const uint8_t* board, const int32_t* seats, float2* obs, long cells, int S) {
    const int A = S * S;
    const float invS = 1.0f / (float)S;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < cells; idx += (long)gridDim.x * blockDim.x) {
        const long b = idx / A;
        const int a = (int)(idx - b * A);
        const int i = (int)(((float)a + 0.5f) * invS), j = a - i * S;
        const bool flip = seats[b] == 1;
        const int color = color_of(board[b * A + (flip ? j * S + i : a)]);
        float2 o = make_float2(0.f, 0.f);
        if (color < 2) { if ((flip ? 1 - color : color) == 0) o.x = 1.f; else o.y = 1.f; }
        obs[idx] = o;
    }
}

// observe + Hex.valid (hex/__init__.py:154-159: (obs == 0).all(-1)) in one pass
__global__ void __launch_bounds__(256) hex_observe_valid_kernel(const uint8_t* board, const int32_t* seats, float2* obs, uint8_t* valid, long cells, int S) {
    const int A = S * S;
    const float invS = 1.0f / (float)S;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < cells; idx += (long)gridDim.x * blockDim.x) {
        const long b = idx / A;
        const int a = (int)(idx - b * A);
        const int i = (int)(((float)a + 0.5f) * invS), j = a - i * S;
        const bool flip = seats[b] == 1;
        const int color = color_of(board[b * A + (flip ? j * S + i : a)]);
        float2 o = make_float2(0.f, 0.f);
        if (color < 2) { if ((flip ? 1 - color : color) == 0) o.x = 1.f; else o.y = 1.f; }
        obs[idx] = o;
        valid[idx] = color == 2;
    }
}


}  // namespace bl

using namespace bl;

extern "C" {

// From how many envs the board kernels run as LDS-staged tiles of 64 consecutive envs (tools/hex_tile_ab.py, profiles/r05_hex_tiles.txt,
// 11x11, us per call lanes-per-env / tiled): step 7.4 / 6.1 at 1024 envs, 250 / 72 at 2^20 -- always; observe + valid 3.1 / 12.7 at
// 4096, 19.6 / 20.5 at 65536, 119 / 70 at 262144, 518 / 258 at 2^20 -- a tile's 30 dependent quads per thread are its latency floor,
// so only from 2^17 envs on.
#define BL_HEX_TILE_MIN_ENVS 1
#define BL_HEX_OBSERVE_TILE_MIN_ENVS (1 << 17)
static void hex_tile_launch(const uint8_t* board_in, uint8_t* board_out, const int32_t* seats_in, const void* actions, int actions_i64,
                            int32_t* seats_out, float* rewards, uint8_t* terminal, int B, int S, int world, hipStream_t stream) {
    const int A = S * S;          // <= 256: the callers check S <= 16
    bl::HexMasks hm{};
    for (int a = 0; a < A; a++) {
        if (a % S > 0) hm.not_first[a >> 5] |= 1u << (a & 31);
        if (a % S < S - 1) hm.not_last[a >> 5] |= 1u << (a & 31);
    }
    const dim3 grid((unsigned)((B + 63) / 64));
    // + the scan's over-read behind the last env: it reads all 32 NW bytes of an env's words whatever A is (up to 32 NW - A bytes past
    // the board; round-5 advisor: the old pad of 32 only held because of LDS allocation granularity)
    const size_t lds = (size_t)((64 * A + 15) & ~15) + (A <= 128 ? 128 : 256);
    if (A <= 128)
        hipLaunchKernelGGL((hex_step_tile_kernel<4>), grid, dim3(256), lds, stream, board_in, board_out, seats_in, actions, actions_i64, seats_out,
                           rewards, terminal, B, S, world, hm);
    else
        hipLaunchKernelGGL((hex_step_tile_kernel<8>), grid, dim3(256), lds, stream, board_in, board_out, seats_in, actions, actions_i64, seats_out,
                           rewards, terminal, B, S, world, hm);
}

int bl_hex_step_tiled(uint8_t* board, const int32_t* seats, const int32_t* actions, float* rewards, int B, int S, bl_stream_t stream) {
    if (!board || !seats || !actions || !rewards || B <= 0 || S <= 0 || ((uintptr_t)board & 15) != 0) return BL_EINVAL;
    if (S > 16) return BL_ETOOBIG;
    hex_tile_launch(board, board, seats, actions, 0, nullptr, rewards, nullptr, B, S, 0, (hipStream_t)stream);
    return check_launch();
}

int bl_hex_step(uint8_t* board, const int32_t* seats, const int32_t* actions, float* rewards, int B, int S, bl_stream_t stream) {
    if (!board || !seats || !actions || !rewards || B <= 0 || S <= 0) return BL_EINVAL;
    if (S > 32) return BL_ETOOBIG;
    if (B >= BL_HEX_TILE_MIN_ENVS && S <= 16 && ((uintptr_t)board & 15) == 0) return bl_hex_step_tiled(board, seats, actions, rewards, B, S, stream);
    constexpr int G = 16;
    const int blocks = (B + 64 / G - 1) / (64 / G);
    hipLaunchKernelGGL((hex_step_kernel<G>), dim3(blocks), dim3(64), (size_t)((S * S + 15) & ~15) * (64 / G),
                       (hipStream_t)stream, board, seats, actions, rewards, B, S);
    return check_launch();
}

int bl_hex_world_step_tiled(const uint8_t* board_in, const int32_t* seats_in, const void* actions, int actions_i64,
                            uint8_t* board_out, int32_t* seats_out, float* rewards, uint8_t* terminal, int B, int S, bl_stream_t stream) {
    if (!board_in || !seats_in || !actions || !board_out || !seats_out || !rewards || !terminal || B <= 0 || S <= 0 ||
        (((uintptr_t)board_in | (uintptr_t)board_out) & 15) != 0) return BL_EINVAL;
    if (S > 16) return BL_ETOOBIG;
    hex_tile_launch(board_in, board_out, seats_in, actions, actions_i64, seats_out, rewards, terminal, B, S, 1, (hipStream_t)stream);
    return check_launch();
}

int bl_hex_world_step(const uint8_t* board_in, const int32_t* seats_in, const void* actions, int actions_i64,
                      uint8_t* board_out, int32_t* seats_out, float* rewards, uint8_t* terminal, int B, int S,
                      bl_stream_t stream) {
    if (!board_in || !seats_in || !actions || !board_out || !seats_out || !rewards || !terminal || B <= 0 || S <= 0) return BL_EINVAL;
    if (S > 32) return BL_ETOOBIG;
    if (B >= BL_HEX_TILE_MIN_ENVS && S <= 16 && (((uintptr_t)board_in | (uintptr_t)board_out) & 15) == 0)
        return bl_hex_world_step_tiled(board_in, seats_in, actions, actions_i64, board_out, seats_out, rewards, terminal, B, S, stream);
    const int A = S * S, G = pick_group(B, A);
    const int blocks = (B + 64 / G - 1) / (64 / G);
    const size_t lds = (size_t)((A + 15) & ~15) * (64 / G);
#define CALL(g) hipLaunchKernelGGL((hex_world_step_kernel<g>), dim3(blocks), dim3(64), lds, (hipStream_t)stream, board_in, seats_in, \
                                   actions, actions_i64, board_out, seats_out, rewards, terminal, B, S)
    switch (G) { case 8: CALL(8); break; case 16: CALL(16); break; case 32: CALL(32); break; default: CALL(64); break; }
#undef CALL
    return check_launch();
}

/* valid may be null (observe only).  Boards up to 16x16 (64 boards of a workgroup within the default 64 KiB of LDS). */
int bl_hex_observe_valid_tiled(const uint8_t* board, const int32_t* seats, float* obs, uint8_t* valid, int B, int S, bl_stream_t stream) {
    if (!board || !seats || !obs || B <= 0 || S <= 0 || (((uintptr_t)board | (uintptr_t)obs | (uintptr_t)valid) & 15) != 0) return BL_EINVAL;
    if (S > 16) return BL_ETOOBIG;
    hipLaunchKernelGGL(hex_observe_tile_kernel, dim3((unsigned)((B + 63) / 64)), dim3(256), (size_t)((64 * S * S + 15) & ~15) + 64, (hipStream_t)stream,
                       board, seats, obs, valid, B, S);
    return check_launch();
}

int bl_hex_observe(const uint8_t* board, const int32_t* seats, float* obs, int B, int S, bl_stream_t stream) {
    if (!board || !seats || !obs || B <= 0 || S <= 0) return BL_EINVAL;
    if (S > 32) return BL_ETOOBIG;
    if (B >= BL_HEX_OBSERVE_TILE_MIN_ENVS && S <= 16 && (((uintptr_t)board | (uintptr_t)obs) & 15) == 0)
        return bl_hex_observe_valid_tiled(board, seats, obs, nullptr, B, S, stream);
    const long cells = (long)B * S * S;
    long blocks = (cells + 255) / 256; if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(hex_observe_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, board, seats,
                       (float2*)obs, cells, S);
    return check_launch();
}

int bl_hex_observe_valid(const uint8_t* board, const int32_t* seats, float* obs, uint8_t* valid, int B, int S, bl_stream_t stream) {
    if (!board || !seats || !obs || !valid || B <= 0 || S <= 0) return BL_EINVAL;
    if (S > 32) return BL_ETOOBIG;
    if (B >= BL_HEX_OBSERVE_TILE_MIN_ENVS && S <= 16 && (((uintptr_t)board | (uintptr_t)obs | (uintptr_t)valid) & 15) == 0)
        return bl_hex_observe_valid_tiled(board, seats, obs, valid, B, S, stream);
    const long cells = (long)B * S * S;
    long blocks = (cells + 255) / 256; if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(hex_observe_valid_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, board, seats,
                       (float2*)obs, valid, cells, S);
    return check_launch();
}


}  // extern "C"
