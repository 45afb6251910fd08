// render_sweep_dev.h -- device body of the sweeping observation render (battle-shaped float32 / bf16-cell observations at scale)
#pragma once
#include "kernels_dev.h"

namespace magent_amd {

// ---- the battle-shaped float32 observation at scale: ONE four-wave workgroup per CU, the whole launch sweeping the output together.
// Measured on MI355X (profiles/r03_render_experiments.md): HBM takes stores fastest when few waves per CU write, in lock step, into
// one narrow moving window -- a device memset's geometry -- and worst from 20-32 independent waves per CU writing 56 KB apart,
// which is what k_render / k_render_fast need to cover their latencies.  Here the launch is 256 persistent workgroups; round r of
// workgroup b is the 4 x SU consecutive steps starting at (r * 256 + b) * 4 * SU, so everything in flight lies within ~3.6 MB.
// What lets four waves per CU keep up: every load is unconditional (clamped address, result selected) and requested DV rounds
// ahead -- x / y one round further -- in a ring of register slots that is never copied (a copy of a register with a load in
// flight waits for the load); the only branches are wave-uniform; a wave carries SU steps through SU LDS strips at once so that
// their ds_write -> ds_read -> store round trips overlap.  (SU = 2, DV = 2 measured best; SU = 3 / 4 and two workgroups per CU lose.)
// (MINI: the game has minimap channels -- 7 floats per cell [wall | has, hp, minimap | has, hp, minimap]; without -- round 5 -- 5: [wall | has, hp | has, hp],
// the shape of the reference's pursuit-like 1M harness)
// (the body is shared: render.hip's k_render_sweep2 -- one group per launch, 256 workgroups -- and pipe.hip's k_pipe_render_sweep -- the groups
// of MANY environments per launch, `sweep` workgroups per segment; bx / nb: the workgroup's position within its segment, of sweep + feature workgroups)
template <bool CELLS16, int DV, int SU, bool MINI = true>
__device__ __forceinline__ void render_sweep2_body(const RenderWorld &W, const RenderArgs &R, const RenderPlan &P, int sweep, unsigned bx, unsigned nb) {
    constexpr int C = MINI ? 7 : 5;                   // floats per window cell
    constexpr int Q2 = 16 * C - 64;                   // a strip is 16 * C float4: 64 in a first store instruction, Q2 in a second
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int N = DV + 2;                         // ring slots: rounds r .. r + DV + 1
    const int VHW = R.VH * R.VW;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if ((int)bx >= sweep) {     // trailing workgroups: the feature rows (measured: better here than at the end of the sweeping
        features_body<true>(W, R, P, bx - sweep, nb - sweep);   // workgroups, where four waves per CU crawl through them)
        return;
    }
    float *strips = (float *)smem + (size_t)wave * (SU * 64 * C);
    RenderFastPos *wtab = (RenderFastPos *)((float *)smem + RENDER_WAVES * SU * 64 * C);
    const GroupDev Gd = W.grp;
    const TypeDev T = W.type;
    const unsigned total_cells = (unsigned)R.n * (unsigned)VHW;
    const unsigned total_steps = (total_cells + 63u) / 64u;
    const size_t total_floats = (size_t)total_cells * C;
    const unsigned *vc = (const unsigned *)W.viewcell;
    const unsigned g = (unsigned)R.g;
    const unsigned ncell_map = (unsigned)W.w * (unsigned)W.h;
    for (int c = threadIdx.x; c < VHW; c += 64 * RENDER_WAVES) {
        const int vy = fdiv_u32(c, P.div_vw), vx = c - vy * R.VW;
        RenderFastPos e;
        e.dxy = ((T.view_y1 + vy) << 16) | ((T.view_x1 + vx) & 0xFFFF);
        e.m0 = MINI ? R.mini[(int)g * VHW + c] : 0.0f;
        e.m1 = MINI ? R.mini[(1 - (int)g) * VHW + c] : 0.0f;
        e.mask = W.mask[T.mask_off + c];
        wtab[c] = e;
    }
    __syncthreads();
    // ring state per slot and step: agent, window cell, x, y (requested DV + 1 rounds ahead), view cell (DV rounds ahead)
    int ia[N][SU], ic[N][SU], x[N][SU], y[N][SU];
    unsigned v[N][SU], in[N][SU];
    // (P.xcd_chunk < 0, tuning: workgroup b -- which runs on XCD b % 8 -- takes slot (b % 8) * (sweep / 8) + b / 8 of the round, so that an XCD's
    // workgroups write one contiguous eighth of the window)
    const unsigned slot = (P.xcd_chunk < 0 && (sweep & 7) == 0) ? (bx & 7u) * ((unsigned)sweep >> 3) + (bx >> 3) : bx;
    auto first_step = [&](unsigned round) { return ((round * (unsigned)sweep + slot) * RENDER_WAVES + wave) * SU; };
    auto index = [&](unsigned round, int slot) {
#pragma unroll
        for (int u = 0; u < SU; u++) {
            const unsigned kk = min((first_step(round) + u) * 64u + lane, total_cells - 1u);
            ia[slot][u] = (int)fdiv_u32(kk, P.div_vhw);
            ic[slot][u] = (int)(kk - (unsigned)ia[slot][u] * VHW);
            x[slot][u] = Gd.x[ia[slot][u]]; y[slot][u] = Gd.y[ia[slot][u]];
        }
    };
    auto request = [&](int slot) {
#pragma unroll
        for (int u = 0; u < SU; u++) {
            const RenderFastPos wt = wtab[ic[slot][u]];
            const int mx = x[slot][u] + ((wt.dxy << 16) >> 16), my = y[slot][u] + (wt.dxy >> 16);
            const bool inside = wt.mask && (unsigned)mx < (unsigned)W.w && (unsigned)my < (unsigned)W.h;
            in[slot][u] = inside ? 1u : 0u;
            v[slot][u] = vc[min((unsigned)(my * W.w + mx), ncell_map - 1u)];     // always in bounds; dropped below when outside
        }
    };
#pragma unroll
    for (int j = 0; j < N; j++) index(j, j);
#pragma unroll
    for (int j = 0; j < DV; j++) request(j);
    bool running = true;
    for (unsigned round = 0; running; round += N) {
#pragma unroll
        for (int s = 0; s < N; s++) {
            if (!running) break;
            const unsigned step0 = first_step(round + s);
            if (step0 >= total_steps) { running = false; break; }
            request((s + DV) % N);
            float cs[SU][C];
#pragma unroll
            for (int u = 0; u < SU; u++) {
                const int cell = ic[s][u];
                const RenderFastPos wt = wtab[cell];
                const unsigned v0 = in[s][u] ? v[s][u] : VC_EMPTY;
                const unsigned top = v0 >> 30;
                const float hp = __uint_as_float(v0 & 0x3FFFFFFFu);
                const bool mine = top == g, theirs = top == 1u - g;
                cs[u][0] = v0 == VC_WALL ? 1.0f : 0.0f; cs[u][1] = mine ? 1.0f : 0.0f; cs[u][2] = mine ? hp : 0.0f;
                if (MINI) {
                    const bool self = cell == (int)(fdiv_u32(y[s][u], P.div_scale_h) * R.VW + fdiv_u32(x[s][u], P.div_scale_w));
                    const float m0 = (self && wt.m0 == wt.m0) ? wt.m0 + 1.0f : wt.m0;
                    const float m1 = (self && wt.m1 == wt.m1) ? wt.m1 + 1.0f : wt.m1;
                    cs[u][3] = m0;
                    cs[u][C - 3] = theirs ? 1.0f : 0.0f; cs[u][C - 2] = theirs ? hp : 0.0f; cs[u][C - 1] = m1;
                } else {
                    cs[u][3] = theirs ? 1.0f : 0.0f; cs[u][4] = theirs ? hp : 0.0f;
                }
            }
            const unsigned k_grp = step0 * 64u;
            if (CELLS16) {
#pragma unroll
                for (int u = 0; u < SU; u++) {
                    const unsigned k = k_grp + 64u * u + lane;
                    cell16_t o;
#pragma unroll
                    for (int e = 0; e < 7; e++) o[e] = (__bf16)(e < C ? cs[u][e < C ? e : 0] : 0.0f);
                    o[7] = (__bf16)1.0f;
                    if (k < total_cells) __builtin_nontemporal_store(o, (cell16_t *)R.view + k);
                }
            } else {
#pragma unroll
                for (int u = 0; u < SU; u++) {
                    float *dst = strips + u * (64 * C) + lane * C;
#pragma unroll
                    for (int e = 0; e < C; e++) dst[e] = cs[u][e];
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                const size_t f_grp = (size_t)k_grp * C;
                if (total_floats - f_grp >= (size_t)(SU * 64 * C)) {
                    v4f q0[SU], q1[SU];
#pragma unroll
                    for (int u = 0; u < SU; u++) {
                        const v4f *src4 = (const v4f *)(strips + u * (64 * C));
                        q0[u] = src4[lane];
                        q1[u] = src4[lane + (lane < Q2 ? 64 : 0)];   // (lanes Q2..63 re-read a vector they do not store: no branch around the read)
                    }
#pragma unroll
                    for (int u = 0; u < SU; u++) {
                        v4f *out4 = (v4f *)(R.view + f_grp + (size_t)u * (64 * C));
                        __builtin_nontemporal_store(q0[u], out4 + lane);
                        if (lane < Q2) __builtin_nontemporal_store(q1[u], out4 + lane + 64);
                    }
                } else {
#pragma unroll
                    for (int u = 0; u < SU; u++) {
                        const size_t f0 = f_grp + (size_t)u * (64 * C);
                        if (f0 >= total_floats) break;
                        const size_t remain = total_floats - f0;
                        const float *strip_u = strips + u * (64 * C);
                        const int nq = remain >= (size_t)(64 * C) ? 16 * C : (int)(remain >> 2);
                        for (int q = lane; q < nq; q += 64) __builtin_nontemporal_store(((const v4f *)strip_u)[q], (v4f *)(R.view + f0) + q);
                        if (remain < (size_t)(64 * C))
                            for (int e = (nq << 2) + lane; e < (int)remain; e += 64) R.view[f0 + e] = strip_u[e];
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
            }
            index(round + s + N, s);                   // the slot is free: round r + N moves in
        }
    }
}

}  // namespace magent_amd
