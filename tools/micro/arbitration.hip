// How does a SIMD share its VALU between co-resident single-wave workgroups, and does s_setprio change it?
// 4096 one-wave blocks (4 per SIMD) run the fold's dependent DPP chain; every 16th block is "deep" (5x the work).
// Prints when the shallow and the deep waves finish (us after the first wave started) for:
//   mode 0: no priorities     mode 1: deep waves s_setprio 3     mode 2: deep waves alone (shallow ones exit at once)
// and the spread of finish times among equal-work waves (age-ordered arbitration shows up as quartiles).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#define R4(x) x x x x
#define R16(x) R4(R4(x))
#define WS "wave_shr:1 row_mask:0xf bank_mask:0xf"

__global__ void work(float* out, long long* t_start, long long* t_end, unsigned* hwid, int steps16, int mode) {
    const bool deep = ((blockIdx.x * 2654435761u) >> 28) == 0;      // 1 in 16, spread over XCDs
    if (mode == 1 && deep) __builtin_amdgcn_s_setprio(3);
    long long t0 = wall_clock64();
    float x = threadIdx.x, y = 1.f, ts = 1.0f, tg = 0.5f;
    int n = deep ? 5 * steps16 : (mode == 2 ? 0 : steps16);
    for (int j = 0; j < n; j++)
        asm volatile(R16("v_add_f32_dpp %0, %0, %2 " WS "\n v_add_f32_dpp %1, %1, %3 " WS "\n") : "+v"(x), "+v"(y) : "v"(ts), "v"(tg));
    out[blockIdx.x * 64 + threadIdx.x] = x + y;
    if (threadIdx.x == 0) {
        t_start[blockIdx.x] = t0; t_end[blockIdx.x] = wall_clock64();
        unsigned id, xcc; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(id));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        hwid[blockIdx.x] = (id & 0xffffu) | ((xcc & 0xf) << 16);
    }
}

int main() {
    const int W = 4096, steps16 = 400;     // shallow: 6400 steps ~ 63k cycles alone
    float* out; long long *ts, *te; unsigned* id;
    hipMalloc(&out, W * 64 * 4); hipMalloc(&ts, W * 8); hipMalloc(&te, W * 8); hipMalloc(&id, W * 4);
    std::vector<long long> hs(W), he(W); std::vector<unsigned> hid(W);
    for (int mode = 0; mode < 3; mode++) {
        for (int rep = 0; rep < 2; rep++) hipLaunchKernelGGL(work, dim3(W), dim3(64), 0, 0, out, ts, te, id, steps16, mode);
        hipDeviceSynchronize();
        hipMemcpy(hs.data(), ts, W * 8, hipMemcpyDeviceToHost); hipMemcpy(he.data(), te, W * 8, hipMemcpyDeviceToHost);
        hipMemcpy(hid.data(), id, W * 4, hipMemcpyDeviceToHost);
        long long t0 = *std::min_element(hs.begin(), hs.end());
        std::vector<double> sh, dp, st;
        auto isdeep = [](unsigned b) { return ((b * 2654435761u) >> 28) == 0; };
        for (int b = 0; b < W; b++) { (isdeep(b) ? dp : sh).push_back((he[b] - t0) / 100.0); st.push_back((hs[b] - t0) / 100.0); }
        {   // deep waves that are the only deep wave on their SIMD, by age rank among the SIMD's waves (0 = started first)
            std::vector<double> byrank[8];
            for (int b = 0; b < W; b++) if (isdeep(b)) {
                const unsigned key = hid[b] & 0xffff0u & ~0xfu;      // everything but the wave slot
                int rank = 0, deeps = 0, total = 0;
                for (int c = 0; c < W; c++) if ((hid[c] & 0xffff0u) == (hid[b] & 0xffff0u)) {
                    total++; if (isdeep(c)) deeps++;
                    if (hs[c] < hs[b] || (hs[c] == hs[b] && c < b)) rank++;
                }
                (void)key;
                if (deeps == 1 && rank < 8) byrank[rank].push_back((he[b] - t0) / 100.0);
            }
            printf("  lone deep waves by age rank on their SIMD (count, mean end us):");
            for (int r = 0; r < 8; r++) if (!byrank[r].empty()) { double m = 0; for (double v : byrank[r]) m += v; printf("  r%d: %zu, %.1f", r, byrank[r].size(), m / byrank[r].size()); }
            printf("\n");
        }
        std::sort(sh.begin(), sh.end()); std::sort(dp.begin(), dp.end()); std::sort(st.begin(), st.end());
        auto q = [](std::vector<double>& v, double f) { return v[(size_t)(f * (v.size() - 1))]; };
        printf("mode %d: starts p50 %.1f max %.1f us | shallow end p5 %.1f p25 %.1f p50 %.1f p75 %.1f p95 %.1f max %.1f | deep end p5 %.1f p50 %.1f p95 %.1f max %.1f\n",
               mode, q(st, .5), q(st, 1), q(sh, .05), q(sh, .25), q(sh, .5), q(sh, .75), q(sh, .95), q(sh, 1), q(dp, .05), q(dp, .5), q(dp, .95), q(dp, 1));
        if (mode == 0) {
            // where do consecutive blocks land?  HW_ID: wave_id [3:0], simd_id [5:4], cu_id [11:8], sh_id [12], se_id [15:13] (gfx9 layout)
            printf("  block -> (se, cu, simd): ");
            for (int b = 0; b < 24; b++) printf("%d:(x%u,%u,%u,%u,w%u) ", b, hid[b] >> 16, (hid[b] >> 13) & 7, (hid[b] >> 8) & 15, (hid[b] >> 4) & 3, hid[b] & 15);
            printf("\n");
        }
    }
    return 0;
}
