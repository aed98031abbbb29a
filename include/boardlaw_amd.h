/* boardlaw_amd.h -- C ABI of libboardlaw_amd.so: the MI355X (gfx950) drop-in for boardlaw's two native modules.
 *
 * The reference has no C ABI: its boundary is two pybind11/ATen modules (`mctscuda`, `hexcuda`) JIT-built by
 * boardlaw/cuda.py:48-63 and wrapped by boardlaw/mcts/cuda.py and boardlaw/hex/cuda.py.  Each entry point below
 * names the reference binding it replaces; INTEGRATION.md shows the ctypes stub a boardlaw maintainer would put in
 * boardlaw/mcts/cuda.py / boardlaw/hex/cuda.py to call it.
 *
 * Conventions (all functions):
 *   - plain pointers and sizes, no torch types; every pointer is a DEVICE pointer unless its name says host_;
 *   - all buffers (inputs, outputs, scratch) are owned by the caller, contiguous row-major, dtypes as listed:
 *       f16  = IEEE binary16 (torch.half)   i16 = int16_t (torch.short)   u8 = uint8_t (torch.bool / torch.uint8)
 *   - work is enqueued on `stream` (a hipStream_t passed as void*) and NOT synchronised, like the reference's
 *     launches on the current torch stream (boardlaw/cpp/kernels.cu:8-10); safe inside hipGraph capture
 *     (no allocation, no synchronisation, no host-side state);
 *   - return 0 on success, a negative BL_E* code otherwise (never throws); bl_strerror() describes it;
 *   - re-entrant and GIL-free: the library keeps no global mutable state and reads no environment variables; every
 *     tuning choice is an explicit field of bl_tune_t, set by the caller (zero-initialised = the defaults).
 *
 * Shapes: B envs, T node slots per env, A actions (= board cells), S seats.
 */
#ifndef BOARDLAW_AMD_H
#define BOARDLAW_AMD_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BL_OK 0
#define BL_EINVAL (-1)     /* bad size / null pointer */
#define BL_ETOOBIG (-2)    /* A, T or S beyond what the kernels support (A <= 1024, T <= 32767, S <= 8, board <= 32) */
#define BL_ELAUNCH (-3)    /* hipGetLastError() != hipSuccess after the launch (reference: C10_CUDA_CHECK) */

#define BL_QRANGE_WORDS 4096 /* 32-bit words in one q-range state: 64 slots, one per 256 B, words 0/1 = {~enc(min), enc(max)} ^ 0x80000000 */

typedef void* bl_stream_t; /* hipStream_t */

/* Explicit tuning choices (all 0 = the library's defaults).  Results never depend on them -- only which of the
 * parity-tested kernel variants runs. */
typedef struct {
    int fold_fast;     /* 1: bl_sim_expand pads its dependent DPP fold steps with ONE wait state instead of the ISA's two (30 % faster).
                          Set it only for a device on which bl_selftest() returned 0; 0: the ISA-padded fold */
    int expand_waves;  /* waves per env in bl_sim_expand: 0 = default (4 up to 1024 envs, 2 below 16384, else 1); 1, 2, 4, 21 = two nodes per wave, or 16 = four envs per wave (bl_rows.hip; measured slower) */
    int expand_deep;   /* speculative guesses only from this descent level on (default 0) */
    int expand_legacy; /* 1: bl_sim_expand runs the general kernel on logits/children instead of the compacted rows */
    int group;         /* lanes per env in the general kernels: 0 = heuristic (64), or 8 / 16 / 32 / 64 */
    int mlp_no_xcd;    /* 1: bl_sim_infer_finish forms its 32-row tiles from consecutive envs instead of same-XCD envs */
    int lazy_init;     /* 1: bl_sim_init resets only what a search reads before writing (the (B,T) arrays, node 0's rows); the big
                          (B,T,A) API arrays get their reset values slot by slot from bl_sim_expand #sim (children[b,sim,:] = -1;
                          if the simulation creates no node also logits[b,sim,:] = NaN and the root board in boards[b,sim]).  After
                          all T-1 simulations every array equals the eager reset's; before that, slots > sim are undefined: for
                          callers that always run a whole search (MCTSAgent) */
    int expand_envs;   /* reserved (ABI 3 layout kept): round 4's shared-workgroup bl_sim_expand -- 2 / 4 envs per workgroup, the waves
                          of finished descents helping the ones still going; bit-exact, slower at every batch size -- was removed in
                          round 5.  0 or 1; anything else makes bl_sim_expand return BL_EINVAL */
    int mlp_rows;      /* rows per workgroup in bl_sim_infer_finish: 0 = by the batch (32; 64 once 32-row tiles outnumber the CUs and
                          the width allows), or 32 / 64.  (This slot was round 4's reserved `expand_help`: ABI 4 layout kept.) */
    int powf_libm;     /* NOT a tuning choice but a second parity target: 1 = the Newton derivative term divides by glibc's
                          powf(bot, 2) (what the reference's own JIT build computes: no -O flag, boardlaw/cuda.py:29-45,
                          boardlaw/mcts/cpp/cpu.cpp:60) instead of bot * bot (what g++ -O1 and up make of it; the default).  The two
                          differ on 0.036 % of floats; results then match oracle/liboracle_powf.so / _ref/mctscuda_O0.so */
} bl_tune_t;

int bl_abi_version(void);

/* powf(x, 2.0f) as glibc 2.35 on x86-64 with FMA computes it (csrc/bl_powf.h), on the device: out[i] = powf(x[i], 2).  What
 * bl_tune_t.powf_libm puts under the Newton derivative term; exported so that tests can pin it to the host libm. */
int bl_powf2(const float* x, float* out, long n, bl_stream_t stream);
const char* bl_strerror(int code);

/* pi = expf(logit) for every binary16 bit pattern, computed by the host libm -- the function the reference's CPU
 * path calls at boardlaw/mcts/cpp/cpu.cpp:86,90.  The caller uploads the 65536 floats once per device and passes the
 * device copy as `exp_table` below; this keeps the GPU's pi bit-identical to the reference CPU path's. */
int bl_exp_table_host(float* host_table /* 65536 floats, HOST memory */);

/* ---- transition_q's batch-global range (boardlaw/mcts/cpp/cuda.cu:101-105) --------------------------------------
 * qrange_state: BL_QRANGE_WORDS x 32-bit words, device: 64 slots of {max of ~enc(q), max of enc(q)} over the slot's share of
 * the B*T*S values q = f32(w)/(f32(n)+1e-4f); enc = the order-preserving float->u32 map, and a word IN MEMORY is its code
 * XOR 0x80000000, i.e. order-preserving as SIGNED int32 (the identity of the MAX is the word 0x80000000; ABI 4).  One slot
 * per 256 B: atomics on one cache line serialise in L2 (about 12 ns each, measured: 4096 waves x 2 atomics on 4 lines cost
 * ~25 us), on 64 lines they do not; consumers max-reduce the 64 slots with one wave-wide load.  bl_mcts_qrange resets the
 * state itself (a kernel, not a memset node) and reduces into it.  Shards that want the reference's *global* normalisation
 * all-reduce(MAX) the words across ranks AS int32, in place -- one collective, no conversion (that is what the signed
 * representation is for: boardlaw_amd/parallel.py: allreduce_qrange).  bl_qrange_decode turns a HOST copy into {min,max}. */
int bl_mcts_qrange(const void* w /*f16 (B,T,S)*/, const int16_t* n /*(B,T)*/, int B, int T, int S,
                   uint32_t* qrange_state, bl_stream_t stream);
int bl_qrange_decode(const uint32_t host_state[BL_QRANGE_WORDS], float host_minmax[2]);

/* ---- mctscuda.descend(m) -> Descent{parents, actions}   (boardlaw/mcts/cpp/wrappers.cpp:24-30, cuda.cu:138-203) --
 * `rands` is the (B,T) f16 tensor the reference draws internally with at::rand_like (cuda.cu:191); the host wrapper
 * draws it with torch so the generator is consumed identically.  seats are i16 (B,T) as in the reference's MCTS
 * struct (mcts/cpp/common.h:25-33). */
int bl_mcts_descend(const void* logits /*f16 (B,T,A)*/, const void* w /*f16 (B,T,S)*/, const int16_t* n /*(B,T)*/,
                    const void* c_puct /*f16 (B)*/, const int16_t* seats /*(B,T)*/, const uint8_t* terminal /*(B,T)*/,
                    const int16_t* children /*(B,T,A)*/, const void* rands /*f16 (B,T)*/,
                    const uint32_t* qrange_state, const float* exp_table,
                    int B, int T, int A, int S,
                    int16_t* parents_out /*(B)*/, int16_t* actions_out /*(B)*/, bl_stream_t stream);

/* bl_mcts_descend with explicit tuning (`tune->group`: the narrower lanes-per-env kernels, kept for parity tests). */
int bl_mcts_descend_tuned(const bl_tune_t* tune, const void* logits, const void* w, const int16_t* n, const void* c_puct,
                          const int16_t* seats, const uint8_t* terminal, const int16_t* children, const void* rands,
                          const uint32_t* qrange_state, const float* exp_table, int B, int T, int A, int S,
                          int16_t* parents_out, int16_t* actions_out, bl_stream_t stream);

/* ---- mctscuda.root(m) -> (B,A) f16 probabilities   (wrappers.cpp:32-38, cuda.cu:107-136) ------------------------- */
int bl_mcts_root(const void* logits, const void* w, const int16_t* n, const void* c_puct, const int16_t* seats,
                 const uint8_t* terminal, const int16_t* children,
                 const uint32_t* qrange_state, const float* exp_table,
                 int B, int T, int A, int S, void* probs_out /*f16 (B,A)*/, bl_stream_t stream);

int bl_mcts_root_tuned(const bl_tune_t* tune, const void* logits, const void* w, const int16_t* n, const void* c_puct,
                       const int16_t* seats, const uint8_t* terminal, const int16_t* children, const uint32_t* qrange_state,
                       const float* exp_table, int B, int T, int A, int S, void* probs_out, bl_stream_t stream);

/* ---- mctscuda.backup(bk, leaves)   (wrappers.cpp:40-46, cuda.cu:205-248): mutates w and n in place -------------- */
int bl_mcts_backup(const void* v /*f16 (B,T,S)*/, void* w /*f16 (B,T,S)*/, int16_t* n /*(B,T)*/,
                   const void* rewards /*f16 (B,T,S)*/, const int16_t* parents /*(B,T)*/,
                   const uint8_t* terminal /*(B,T)*/, const int16_t* leaves /*(B)*/,
                   int B, int T, int S, bl_stream_t stream);

/* ---- hexcuda.step(board, seats, actions) -> (B,2) f32   (hex/cpp/wrappers.cpp:20-26, cuda.cu:76-152) -------------
 * board (B,S,S) u8 is mutated in place (the caller clones, hex/__init__.py:181); rewards_out is fully written
 * (zeros unless the move wins).  No legality check, like the reference. */
int bl_hex_step(uint8_t* board, const int32_t* seats, const int32_t* actions, float* rewards_out /*(B,2)*/,
                int B, int boardsize, bl_stream_t stream);

/* Hex.step(actions) with reset=True as one launch (boardlaw/hex/__init__.py:161-195): board_out = step(clone(board_in));
 * rewards (B,2) f32; terminal = any reward > 0; finished boards are wiped and handed to seat 0, the others to the other
 * seat.  actions: i32 (actions_i64 = 0) or i64 (1) flat cell indices in the mover's frame.  No validity checks (the
 * Python caller keeps the reference's asserts when asked to). */
int bl_hex_world_step(const uint8_t* board_in, const int32_t* seats_in, const void* actions, int actions_i64,
                      uint8_t* board_out, int32_t* seats_out, float* rewards_out, uint8_t* terminal_out, int B, int S,
                      bl_stream_t stream);

/* ---- hexcuda.observe(board, seats) -> (B,S,S,2) f32   (hex/cpp/wrappers.cpp:28-34, cuda.cu:154-217) -------------
 * Leading dims are flattened by the caller.  obs_out is fully written. */
int bl_hex_observe(const uint8_t* board, const int32_t* seats, float* obs_out, int B, int boardsize, bl_stream_t stream);

/* observe plus Hex.valid = (obs == 0).all(-1) (boardlaw/hex/__init__.py:148-159) in one launch; valid_out u8 (B, S*S). */
int bl_hex_observe_valid(const uint8_t* board, const int32_t* seats, float* obs_out, uint8_t* valid_out, int B, int boardsize,
                         bl_stream_t stream);

/* The three board kernels as HBM streams (round 5), for boards up to 16x16: a workgroup takes 64 CONSECUTIVE envs -- one contiguous
 * 16-byte-aligned run of boards -- through LDS with 16-byte loads and stores; four lanes step an env, the flood as a bit-board fill
 * (the component grows by shifts of a cell bit set instead of sweeps over the board); observe writes its f32 planes as 16-byte and
 * its mask as 4-byte stores.  Same results as the functions above, which call these whenever the boards qualify (step, world_step) resp.
 * from 2^17 envs on (observe: below that a group of lanes per env fills more of the chip and a launch is shorter).  BL_EINVAL if board / obs / valid pointers are not
 * 16-byte aligned, BL_ETOOBIG beyond 16x16; bl_hex_observe_valid_tiled: valid_out may be null. */
int bl_hex_step_tiled(uint8_t* board, const int32_t* seats, const int32_t* actions, float* rewards_out, int B, int boardsize, bl_stream_t stream);
int bl_hex_world_step_tiled(const uint8_t* board_in, const int32_t* seats_in, const void* actions, int actions_i64,
                            uint8_t* board_out, int32_t* seats_out, float* rewards_out, uint8_t* terminal_out, int B, int S,
                            bl_stream_t stream);
int bl_hex_observe_valid_tiled(const uint8_t* board, const int32_t* seats, float* obs_out, uint8_t* valid_out, int B, int boardsize,
                               bl_stream_t stream);

/* ================= fused search step for Hex (no reference counterpart) ==================
 * One simulation of boardlaw/mcts/__init__.py:108-140 is  descend -> expand -> world.step -> observe -> network ->
 * store -> backup.  The reference runs ~25 torch ops and 4 host syncs around its three kernels; here everything
 * before the network is bl_sim_expand and everything after it is bl_sim_backup, operating directly on the search's
 * SoA arrays (same names/layouts as MCTS.tree/stats/decisions/transitions/worlds). */
typedef struct {
    /* MCTS.decisions / stats / tree / transitions / worlds, all (B,T,...) */
    void* logits;        /* f16 (B,T,A) */
    void* v;             /* f16 (B,T,2) */
    void* w;             /* f16 (B,T,2) */
    int16_t* n;          /* (B,T) */
    int16_t* children;   /* (B,T,A) */
    int16_t* parents;    /* (B,T) */
    int16_t* relation;   /* (B,T) */
    void* rewards;       /* f16 (B,T,2) */
    uint8_t* terminal;   /* (B,T) */
    uint8_t* boards;     /* (B,T,S,S) */
    int32_t* seats;      /* (B,T) */
    const void* c_puct;  /* f16 (B) */
    uint32_t* qrange;    /* (T+1, BL_QRANGE_WORDS) u32: row s is the state descend #s reads; bl_sim_init resets it */
    const float* exp_table;
    int B, T, boardsize;
    int obs_f16;         /* 0: bl_sim_expand writes obs as f32 (the reference's layout); 1: as f16 (what fp16 autocast feeds
                            the first Linear anyway; exact, the planes are 0/1) */
    int16_t* path;       /* (B,T+2) i16 scratch or NULL: bl_sim_expand records each descent as [len, root, ..., leaf] so
                            that bl_sim_finish can back up without chasing parents[]; required by bl_sim_finish */
    const int32_t* order; /* (B) i32 or NULL: launch slot -> env for bl_sim_expand (a permutation of 0..B-1).  Only the
                            dispatch order changes (e.g. deepest trees first), never a result */
    int prio_thresh;     /* > 0: bl_sim_expand raises the wave priority of envs whose previous descent (path[0]) was at
                            least this long; 0: off */
    /* Compacted policy rows, written by whoever stores a node's logits (bl_sim_finish, bl_sim_infer_finish, bl_sim_backup,
     * bl_sim_plant_root, bl_sim_compact) and read by bl_sim_expand instead of logits[b,t,:] / children[b,t,:]: the actions
     * with expf(logit) != 0 in ascending order (the others add +-0 to the Newton sums of cuda.cu:35-68 and are never drawn).
     * All three NULL: bl_sim_expand runs its general kernel on logits/children. */
    float* cpi;          /* f32 (B,T,A): cpi[b,t,j] = expf(logit of the j-th kept action) (host libm's, via exp_table) */
    uint32_t* cca;       /* u32 (B,T,A): child << 16 | action; child 0xffff = not expanded yet (bl_sim_expand fills it in) */
    int16_t* nk;         /* i16 (B,T): kept actions per node; bl_sim_init zeroes it */
    int16_t* fav;        /* i16 (B,T) scratch or NULL: each node's most visited child (-1: none), maintained by bl_sim_expand as
                            the guess for its speculative batches (several levels of a deep descent evaluated at once, one per
                            wave); a hint only -- results never depend on it.  NULL: one level at a time */
    bl_tune_t tune;      /* explicit tuning choices, zero = defaults */
    const int32_t* n_active; /* DEVICE scalar or NULL (= B): only envs 0 .. *n_active-1 take part in a simulation -- bl_sim_expand,
                            bl_sim_finish, bl_sim_infer_finish, bl_sim_backup and bl_sim_compact skip the others, which then add
                            nothing to the q-range either.  What lets ONE captured move (arrays sized for B envs) serve the arena's
                            masked calls of any size <= B (boardlaw/arena/common.py:88-93): the caller packs the live envs first
                            and rewrites the scalar before each replay; results for the active envs equal a B = *n_active search */
} bl_search_t;

/* mcts/__init__.py:113-129 + hex/__init__.py:148-195 for simulation number `sim` (1..T-1):
 * reads qrange slot `sim`, descends, creates/looks up the leaf, steps the parent's board, stores the leaf world and
 * its transition, and emits what the network needs for the leaf worlds. */
int bl_sim_expand(const bl_search_t* s, int sim, const void* rands /*f16 (B,T)*/,
                  int16_t* leaves_out /*(B)*/, void* obs_out /*(B,S,S,2) f32 or f16, see obs_f16*/, uint8_t* valid_out /*(B,A)*/,
                  int32_t* leaf_seats_out /*(B)*/, bl_stream_t stream);

/* mcts/__init__.py:135-140: stores the network's outputs for the leaves (rounding to f16 exactly like `.half()`),
 * runs the backup walk, and reduces the q-range for descend #(sim+1) into qrange slot sim+1.
 * logits_dtype / v_dtype: 0 = f32, 1 = f16. */
int bl_sim_backup(const bl_search_t* s, int sim, const int16_t* leaves /*(B)*/,
                  const void* leaf_logits /*(B,A)*/, int logits_dtype, const void* leaf_v /*(B,2)*/, int v_dtype,
                  bl_stream_t stream);

/* The same step when the network hands over its PRE-HEAD outputs: policy_raw (B,A) f16 = the policy Linear's output,
 * value_raw (B) f16 = the value Linear's output (both under fp16 autocast).  Computes the heads of boardlaw/heads.py --
 * masked log-softmax in f32 (heads.py:101-104) rounded to f16 like `.half()`, tanh + seat scatter (heads.py:122-142) --
 * with torch's own operation order, then does what bl_sim_backup does. */
int bl_sim_finish(const bl_search_t* s, int sim, const int16_t* leaves /*(B)*/, const void* policy_raw, const void* value_raw,
                  const uint8_t* valid /*(B,A)*/, const int32_t* leaf_seats /*(B)*/, bl_stream_t stream);

/* bl_sim_finish for pre-head outputs in FP32: policy_raw (B,A) f32, value_raw (B) f32 -- the leaf evaluation of the reference's
 * CPU configuration, where `torch.cuda.amp.autocast` (boardlaw/mcts/__init__.py:131-134) is a no-op, the network and its heads
 * run in f32 and only the stores round (`decisions.logits.half()`, `decisions.v.half()`, :135-136).  Same heads, store, backup
 * and q-range as bl_sim_finish; the masked log-softmax and tanh take f32 inputs.  With the Linears from bl_root_mlp_f32 this is
 * the exact-mode leaf evaluation (networks.Inference(precision='fp32')): a seeded search then stores the reference's own f16
 * logits wherever the two f32 GEMM summation orders round to the same binary16. */
int bl_sim_finish_f32(const bl_search_t* s, int sim, const int16_t* leaves /*(B)*/, const float* policy_raw, const float* value_raw,
                      const uint8_t* valid /*(B,A)*/, const int32_t* leaf_seats /*(B)*/, bl_stream_t stream);

/* ReZero residual block tail under fp16 autocast (boardlaw/networks.py:17-18), fused: x_out = x + alpha*y (rounded
 * where torch rounds) and relu_out = relu(x_out); n f16 elements, alpha one f32 on the device. */
int bl_rezero_relu_f16(const void* x, const void* y, const float* alpha, void* x_out, void* relu_out, long n,
                       bl_stream_t stream);

/* The leaf-evaluation network's Linears (boardlaw/networks.py:10-40, heads.py:47-52,95,132) under fp16 autocast as one
 * kernel: intake, D ReZero blocks, and the policy+value head Linears; the heads' nonlinearities are bl_sim_finish's.
 * Matrices are f16 (out, in): w0 (W,K0pad) zero-padded from (W,K0); wb (D,W,W); wh (NHpad,W) with rows 0..NH-2 the
 * policy Linear, row NH-1 the value Linear, zero rows after -- each PACKED fragment-major so that a wave's MFMA operand
 * loads are contiguous:  packed[n/32][k/64][s][lane][j] = M[32*(n/32) + (lane&31)][64*(k/64) + 32*(lane>>5) + 8*s + j],
 * s < 4, lane < 64, j < 8 (boardlaw_amd/networks.py: pack_fragment_major).  W in {128, 256, 512, 768, 1024}, K0pad % 64 == 0,
 * K0pad <= W, NHpad % 32 == 0; BL_ETOOBIG otherwise, or when the two activation buffers + head staging exceed 160 KiB of LDS.
 * Writes policy_out (M,NH-1) f16 and value_out (M) f16.  Same rounding points as torch; GEMM summation order differs. */
int bl_mlp_forward_f16(const void* obs /*f16 (M,K0)*/, int M, int K0, const void* w0, const void* b0, const void* wb,
                       const void* bb, const float* alphas /*(D) f32*/, const void* wh, const void* bh, int W, int D,
                       int K0pad, int NH, int NHpad, void* policy_out, void* value_out, bl_stream_t stream);

/* The ROOT evaluation's Linears in fp32 (MCTS.initialize calls the network outside autocast, mcts/__init__.py:72-76;
 * networks.py:10-40) as one kernel on v_mfma_f32_16x16x4_f32: intake, D ReZero blocks, policy+value head Linears, with
 * torch's fp32 rounding points (bias add, alpha*y, x + ., relu); the heads' nonlinearities are bl_sim_plant_root's.
 * Matrices are f32 (out, in) -- w0 (W,K0pad) zero-padded from (W,K0); wb (D,W,W); wh (NHpad,W) with rows 0..NH-2 the policy
 * Linear, row NH-1 the value Linear, zero rows after -- each PACKED fragment-major:
 *     packed[n/16][k/16][lane][s] = M[16*(n/16) + (lane&15)][16*(k/16) + 4*(lane>>4) + s],  lane < 64, s < 4
 * (boardlaw_amd/networks.py: pack_fragment_major_f32).  W % 128 == 0, W <= 1024, K0pad % 64 == 0, K0pad <= W,
 * NHpad % 16 == 0; BL_ETOOBIG otherwise.  Writes policy_out (M,NH-1) f32 and value_out (M) f32; GEMM summation order is
 * the kernel's own (agrees with the module's library GEMMs to fp32 rounding). */
int bl_root_mlp_f32(const float* obs /*(M,K0)*/, int M, int K0, const float* w0, const float* b0, const float* wb,
                    const float* bb, const float* alphas /*(D)*/, const float* wh, const float* bh, int W, int D, int K0pad,
                    int NH, int NHpad, float* policy_out, float* value_out, bl_stream_t stream);

/* The same network as bl_mlp_forward_f16 (same packed weights, same rounding points), one launch per Linear with every layer
 * split over the whole chip: 32 rows x 128 output features per workgroup, the ReZero tail fused into the epilogue, the residual
 * stream kept in `scratch` (2*M*W f16, caller-owned) between launches.  The plan for wide networks on small batches, where
 * bl_mlp_forward_f16's per-workgroup weight stream (2 bytes x all weights through one CU's L1) is the bound: 1024x8 on 1024 rows.
 * K0 even.  Replaces, like bl_mlp_forward_f16, the reference's autocast FCModel forward (boardlaw/networks.py:10-40). */
int bl_mlp_layers_f16(const void* obs /*f16 (M,K0)*/, int M, int K0, const void* w0, const void* b0, const void* wb,
                      const void* bb, const float* alphas, const void* wh, const void* bh, int W, int D, int K0pad, int NH,
                      int NHpad, void* scratch /*f16 (2,M,W)*/, void* policy_out /*f16 (M,NH-1)*/, void* value_out /*f16 (M)*/,
                      bl_stream_t stream);

/* The same forward as bl_mlp_layers_f16 in ONE launch (round 4): a workgroup keeps its (32 rows, 128 features) share through
 * all layers and waits, between two layers, only for the workgroups that own the other column groups of ITS row tile (release /
 * acquire on a counter per row tile and layer in `counters`).  Bit-identical to bl_mlp_layers_f16.  For grids of at most 256
 * workgroups (1024 rows at W = 1024, 2048 at W = 512, ...): BL_ETOOBIG otherwise -- the caller then launches per Linear.
 * counters: int32 device array of ceil(M/32) * (D + 2) + 1 words that must be ZERO when the kernel starts and is zero again when it
 * ends; zero_first = 1 zeroes it with a launch of its own first (for fresh memory: e.g. a block allocated per call).
 * error: one int32 device word the caller zeroes once; the kernel sets it to 1 when a bounded wait ran out (another process kept the
 * workgroups' peers off the chip): the outputs of that call are invalid. */
int bl_mlp_layers_persist_f16(const void* obs /*f16 (M,K0)*/, int M, int K0, const void* w0, const void* b0, const void* wb,
                              const void* bb, const float* alphas, const void* wh, const void* bh, int W, int D, int K0pad,
                              int NH, int NHpad, void* scratch /*f16 (2,M,W)*/, int32_t* counters, int zero_first, int32_t* error,
                              void* policy_out, void* value_out, bl_stream_t stream);

/* The same forward in one launch with the hand-off between two layers kept inside ONE XCD's L2 (round 5): every workgroup reads the
 * XCD it runs on (HW_REG_XCC_ID), takes a ticket there and works on row tile xcd + 8 * (ticket / column groups) -- the workgroups
 * of a row tile share an L2 by construction -- so the rows are plain stores, the flags plain words, the loads L1-bypassing; no
 * device-scope atomic and no write-through store on the path.  Bit-identical to bl_mlp_layers_f16.  Same arguments and limits as
 * bl_mlp_layers_persist_f16 except: counters = ceil(M/32) * (D + 2) * 8 + 9 words (zero at start, zero again at the end).  Needs
 * the dispatcher to deal the grid's workgroups evenly over the 8 XCDs (it does: in turn); if an XCD receives too few, the bounded
 * waits of an unfinished row tile raise `error` (outputs invalid, no hang). */
int bl_mlp_layers_xcd_f16(const void* obs /*f16 (M,K0)*/, int M, int K0, const void* w0, const void* b0, const void* wb,
                          const void* bb, const float* alphas, const void* wh, const void* bh, int W, int D, int K0pad,
                          int NH, int NHpad, void* scratch /*f16 (2,M,W)*/, int32_t* counters, int zero_first, int32_t* error,
                          void* policy_out, void* value_out, bl_stream_t stream);

/* bl_mlp_forward_f16 followed by bl_sim_finish as ONE launch: the workgroup that took 32 leaves through the network also
 * applies the heads to them, stores logits/v, backs up along the recorded paths and publishes the next q range; what
 * that step reads from the tree is requested at the start of the kernel and arrives under the GEMMs.  Same results as
 * the two calls, bit for bit.  obs is the f16 observation bl_sim_expand wrote (s->obs_f16 == 1); M = s->B, K0 = 2A,
 * NH = A + 1.  BL_ETOOBIG (use the two calls) unless s->T <= 64, A <= 128 and W >= 256. */
int bl_sim_infer_finish(const bl_search_t* s, int sim, const int16_t* leaves /*(B)*/, const void* obs /*f16 (B,2A)*/,
                        const uint8_t* valid /*(B,A)*/, const int32_t* leaf_seats /*(B)*/, const void* w0, const void* b0,
                        const void* wb, const void* bb, const float* alphas, const void* wh, const void* bh, int W, int D,
                        int K0pad, int NHpad, bl_stream_t stream);

/* The ReZero tail in fp32 for the root evaluation (which runs outside autocast): x_out = x + alpha*y, product and sum
 * rounded separately like torch's mul and add; relu_out = relu(x_out). */
int bl_rezero_relu_f32(const float* x, const float* y, const float* alpha, float* x_out, float* relu_out, long n,
                       bl_stream_t stream);

/* MCTS.initialize after the root network's Linears (mcts/__init__.py:72-80, 13-24; heads.py:101-104,122-142) in one
 * launch: masked log-softmax of policy_raw (B,A) f32, dirichlet noise from `draw` (B,A) f32 -- zeroed where invalid,
 * renormalised, mixed in as log(exp(l)(1-eps) + eps*draw); draw may be NULL: no noise step at all -- tanh(value_raw)
 * scattered by seat, both rounded to f16 into node 0 of s->logits / s->v.  The caller then sets its sim counter to 1. */
int bl_sim_plant_root(const bl_search_t* s, const float* policy_raw, const float* value_raw /*(B)*/, const uint8_t* valid,
                      const int32_t* seats /*(B)*/, const float* draw, float eps, bl_stream_t stream);

/* bl_sim_plant_root fed torch's standard-gamma variates (B,A) f32 instead of a finished Dirichlet sample: the kernel first does
 * what at::_sample_dirichlet does after its gamma kernel -- gamma / gamma.sum(-1), clamped to [FLT_MIN, 1 - FLT_EPSILON] -- with
 * that sum in torch's own order (its reduce kernel's lane layout and tree on this build), so node 0's row carries the bits of
 * the reference's `torch.distributions.Dirichlet(alpha).sample()` path (boardlaw/mcts/__init__.py:13-24) with two launches fewer
 * per move.  A < 128 (beyond, torch's sum is address-alignment dependent): BL_ETOOBIG, the caller then draws the Dirichlet
 * with torch and calls bl_sim_plant_root. */
int bl_sim_plant_root_gamma(const bl_search_t* s, const float* policy_raw, const float* value_raw /*(B)*/, const uint8_t* valid,
                            const int32_t* seats /*(B)*/, const float* gamma /*(B,A)*/, float eps, bl_stream_t stream);

/* MCTSAgent's action draw exactly as torch makes it (boardlaw/mcts/__init__.py:221: Categorical(logits = root logits.float())
 * .sample()), in one launch: actions[b] = argmax_a softmax(x - logsumexp(x))[a] / q[b,a] with x = f32(logits[b,:]) and q the
 * Exponential(1) variates torch.multinomial draws (the caller makes them with torch's own exponential_ kernel, so the generator
 * is consumed as by the reference's call) -- every intermediate rounded where torch's ~12 launches round it, sums in torch's
 * order, the lower index on ties.  logits f16 (B,A), q f32 (B,A), actions i64 (B).  A < 128: BL_ETOOBIG otherwise. */
int bl_categorical(const void* logits, const float* q, long long* actions_out, int B, int A, bl_stream_t stream);

/* Builds the compacted rows (cpi, cca, nk) of node leaves[b] of every env -- node 0 when leaves is NULL -- from
 * logits[b,node,:]; for logits stored without one of the calls above (MCTS.plant_root's tensor assignment). */
int bl_sim_compact(const bl_search_t* s, const int16_t* leaves /*(B) or NULL*/, bl_stream_t stream);

/* Device self-test of the one-wait-state fold: call once per DEVICE (with that device current), outside any stream capture
 * (synchronises `stream`).  bl_sim_expand's serial folds pad every dependent DPP step with the 2 wait states the ISA asks for
 * (`s_nop 1`); on gfx950 one (`s_nop 0`) is measured to be enough and 30 % faster.  This runs both variants on 4096 waves x
 * 108 random chains against a serial sum.  Returns the number of wrong totals of the one-wait-state variant -- 0: the caller
 * may set bl_tune_t.fold_fast = 1 for searches on this device -- or a BL_E* code (also when the ISA-padded fold itself is
 * wrong).  The library stores nothing. */
int bl_selftest(bl_stream_t stream);

/* MCTSAgent's action draw (mcts/__init__.py:221: Categorical(logits = log of the root distribution).sample()) by inverse CDF:
 * actions[b] = first action whose running total of probs[b,:] (f16, ascending, summed in f32) reaches uniforms[b] * total,
 * among the actions with positive probability.  One launch instead of torch.multinomial's ~12; the caller supplies the
 * uniforms (torch.rand from its generator).  Same distribution as the reference's draw, its own use of the generator. */
int bl_draw_actions(const void* probs /*f16 (B,A)*/, const float* uniforms /*(B)*/, long long* actions_out /*i64 (B)*/,
                    int B, int A, bl_stream_t stream);

/* The T-1 descend uniforms of one move as ONE launch, bit for bit what n_calls consecutive `at::rand_like` calls on a
 * (numel,) f16 tensor draw from a torch generator on this device (boardlaw/mcts/cpp/cuda.cu:191; torch's kernel:
 * ATen/native/cuda/DistributionTemplates.h, distribution_elementwise_grid_stride_kernel + uniform_kernel).  Call c, thread
 * idx < threads, loop l < loops evaluates Philox4x32-10 with key = seed and counter = {offset/4 + c*loops + l, idx}; component j
 * goes to element idx + threads*(4l + j) of call c as f16(2^-32 + float(u)*2^-32), 1.0 mapped to 0.  `threads` and `loops` are
 * torch's launch geometry for numel elements on the device: threads = 256 * min(CUs * (max threads per CU / 256),
 * ceil(numel / 256)), loops = (numel - 1) / (4 * threads) + 1 (BL_EINVAL if inconsistent); the caller advances the generator by
 * n_calls * 4 * loops.  captured != 0: seed_or_ptr / offset_or_ptr are DEVICE POINTERS to int64 (what torch's generator hands
 * kernels during HIP-graph capture) and offset_intragraph is added to the loaded offset; else they are the values. */
int bl_rand_block(void* out /*f16 (n_calls, numel)*/, int n_calls, long numel, long threads, int loops,
                  unsigned long long seed_or_ptr, unsigned long long offset_or_ptr, unsigned int offset_intragraph, int captured,
                  int only_slots_upto_call /* 0: every element; T > 0: the tensors are (B,T) rows and call c writes only the
                  elements (b, t <= c) -- all descend #c+1 can read (nodes 0..c exist then); the rest stays unwritten.  Halves
                  the Philox work of a move; the written elements are the same bits */,
                  bl_stream_t stream);

/* Up to BL_COPY_MAX device-to-device copies in ONE launch (`items` is HOST memory, read before the call returns).  Copy k
 * moves `rows` rows of `row_bytes` bytes; row r starts at src + r*src_pitch and dst + r*dst_pitch (rows == 1: a flat copy,
 * pitches ignored).  What replaces the reference's `decisions.clone()` / `arrdict.clone()` (mcts/__init__.py:229,
 * rebar/arrdict.py) -- a launch per tensor, ten per move -- when a captured move's outputs are handed to the caller. */
#define BL_COPY_MAX 24
typedef struct {
    const void* src; void* dst;
    unsigned long long row_bytes, rows, src_pitch, dst_pitch;
} bl_copy_t;
int bl_copy_many(const bl_copy_t* items, int n, bl_stream_t stream);

/* MCTS.n_leaves (mcts/__init__.py:151-152): per env, nodes with parents != -1 that no node names as its parent. */
int bl_sim_n_leaves(const bl_search_t* s, long long* out /*i64 (B)*/, bl_stream_t stream);

/* root distribution from qrange slot `sim` (call with sim = number of filled slots, i.e. MCTS.sim).  logits_out (f16 (B,A),
 * may be NULL) = log_table[bits of probs_out]: MCTS.root's `r.log()` (mcts/__init__.py:147) through the caller's 65536-entry
 * f16 -> f16 table (the host's r.float().log().half() for every bit pattern). */
int bl_sim_root(const bl_search_t* s, int sim, void* probs_out /*f16 (B,A)*/, const void* log_table, void* logits_out,
                bl_stream_t stream);

/* MCTS.__init__ (mcts/__init__.py:43-67): children/parents/relation = -1, logits/v = NaN, w/n/rewards/terminal = 0,
 * every slot's board/seat = the root world's, qrange slots zeroed. */
int bl_sim_init(const bl_search_t* s, const uint8_t* root_board /*(B,S,S)*/, const int32_t* root_seats /*(B)*/,
                bl_stream_t stream);

/* Diagnostics for the roofline model: bl_sim_expand that also accumulates, per env, into `counters`
 * ((B,12) u64, device, caller-zeroed): [0] policy evaluations (levels), [1] Newton iterations, [2] most iterations in
 * one level, [3] expanded-child look-ups, then shader-clock totals [4] loads, [5] term evaluation, [6] serial folds,
 * [7] alpha update, [8] whole descent, [9] expansion (step+flood+observe+stores); [10..11] unused. */
int bl_sim_expand_counted(const bl_search_t* s, int sim, const void* rands, int16_t* leaves_out, void* obs_out,
                          uint8_t* valid_out, int32_t* leaf_seats_out, unsigned long long* counters,
                          bl_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif
