// bl_dispatch.h -- host-side launch helpers shared by the translation units that hold the general (lanes-per-env) kernels:
// the choice of lanes per env, the template dispatch over (group, k), and the post-launch error check.
#pragma once
#include <hip/hip_runtime.h>
#include "../../include/boardlaw_amd.h"

// Lanes per env in the general kernels.  `forced` (bl_tune_t.group: 8|16|32|64, 0 = none) overrides the heuristic below.
static inline int pick_group(int B, int A, int forced = 0) {
    const int f = forced;
    if ((f == 8 || f == 16 || f == 32 || f == 64) && (A + f - 1) / f <= 16) return f;
    // One wave per env whenever the action count allows (A <= 64 x 16): that is the DPP path (no LDS, no barriers,
    // serial folds across lanes).  Measured on MI355X at 9x9 it beats the narrower LDS-fold groups at every batch size
    // tried (4096 ... 32768 envs: 1.2-1.6x), because a descent is one long dependent chain and what hides its latency
    // is other waves, not busier lanes.  The narrower groups remain for BL_FORCE_GROUP experiments and for parity tests.
    int G = 64;
    (void)B;
    return G;
}

static inline int pick_k(int A, int G) {
    const int need = (A + G - 1) / G;
    const int ks[7] = {2, 3, 4, 6, 8, 12, 16};
    for (int i = 0; i < 7; i++) if (need <= ks[i]) return ks[i];
    return -1;
}

static inline int check_launch() { return hipGetLastError() == hipSuccess ? BL_OK : BL_ELAUNCH; }

#define BL_DISPATCH_K(g, K, CALL)                                                                    \
    switch (K) {                                                                                     \
        case 2: { CALL(g, 2); } break;   case 3: { CALL(g, 3); } break;   case 4: { CALL(g, 4); } break; \
        case 6: { CALL(g, 6); } break;   case 8: { CALL(g, 8); } break;   case 12: { CALL(g, 12); } break; \
        case 16: { CALL(g, 16); } break; default: return BL_ETOOBIG;                                 \
    }
#define BL_DISPATCH_GK(G, K, CALL)                                                                   \
    switch (G) {                                                                                     \
        case 8: BL_DISPATCH_K(8, K, CALL) break;    case 16: BL_DISPATCH_K(16, K, CALL) break;        \
        case 32: BL_DISPATCH_K(32, K, CALL) break;  case 64: BL_DISPATCH_K(64, K, CALL) break;        \
        default: return BL_ETOOBIG;                                                                  \
    }

