// render.hip -- the observation kernels of the grid-world engine for gfx950: painted map, minimap, feature rows, the three render kernels, their launchers
// (device bodies shared with the other kernel translation units: kernels_dev.h)
#include "render_sweep_dev.h"

namespace magent_amd {

template <bool PACKED>
__global__ void __launch_bounds__(256) k_paint(WorldView W, const GroupDev *gtab, const TypeDev *ttab) {
    const int ncell = W.w * W.h;
    for (int c = blockIdx.x * blockDim.x + threadIdx.x; c < ncell; c += gridDim.x * blockDim.x) {
        int o = W.occ[c];
        int2 rec = make_int2(o, 0);
        if (o >= 0) {
            int g = ref_group(o), i = ref_index(o);
            rec.x = g;
            rec.y = __float_as_int(__fdiv_rn(gtab[g].hp[i], ttab[g].hp));
        }
        if (PACKED) {
            unsigned v = o == OCC_EMPTY ? VC_EMPTY : o == OCC_WALL ? VC_WALL : o == OCC_FOOD ? VC_FOOD : (((unsigned)rec.x << 30) | (unsigned)rec.y);
            if (o >= 0 && ((unsigned)rec.y >> 30)) W.counters[CTR_PACK_OVERFLOW] = 1;   // ratio outside [0, 2): never expected
            ((unsigned *)W.viewcell)[c] = v;
        } else {
            W.viewcell[c] = rec;
        }
    }
}

// ------------------------------------------------------------------------------------------------ minimap histogram
// counts[j][cell] = number of agents of group j whose (x / scale_w, y / scale_h) is cell (GridWorld.cc:341-352;
// dead-but-not-cleared agents are counted, as in the reference).  LDS int atomics per block, then one global
// atomic per non-empty bin.  blockIdx.y = group.
// `skip`: the observing type is can_absorb -- absorbed agents are left out and counted in left_out[j], which the
// normalisation takes off the divisor (GridWorld.cc:343-347: the OBSERVING group's type decides).
__global__ void __launch_bounds__(256) k_minimap(WorldView W, RenderArgs R, int *counts, int *left_out, int skip) {
    extern __shared__ int s_hist[];
    const int VHW = R.VH * R.VW, j = blockIdx.y;
    const GroupDev G = W.grp[j];
    if ((int)(blockIdx.x * blockDim.x) >= G.n) return;
    for (int k = threadIdx.x; k < VHW; k += blockDim.x) s_hist[k] = 0;
    __syncthreads();
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < G.n; i += gridDim.x * blockDim.x) {
        if (skip && G.absorbed[i]) { atomicAdd(&left_out[j], 1); continue; }
        int cx = G.x[i] / R.scale_w, cy = G.y[i] / R.scale_h;
        atomicAdd(&s_hist[cy * R.VW + cx], 1);
    }
    __syncthreads();
    for (int k = threadIdx.x; k < VHW; k += blockDim.x)
        if (s_hist[k]) atomicAdd(&counts[j * VHW + k], s_hist[k]);
}

// ------------------------------------------------------------------------------------------------ minimap normalise
// mini[j][cell] = float(count) / float(total_j) exactly as the reference (GridWorld.cc:350,356): float ++ saturates
// at 2^24; an empty group divides 0 by 0 and the x86 default NaN the reference then holds is 0xFFC00000.
__global__ void __launch_bounds__(256) k_minimap_norm(RenderArgs R, int G, int *counts, float *mini, const int *left_out, int skip) {
    const int VHW = R.VH * R.VW;
    int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= G * VHW) return;
    int tot = R.totals[k / VHW] - (skip ? left_out[k / VHW] : 0);
    mini[k] = tot == 0 ? __int_as_float(0xFFC00000) : __fdiv_rn((float)min(counts[k], 1 << 24), (float)(unsigned)tot);
    counts[k] = 0;   // the next histogram starts from zero (the buffer is zeroed when it is allocated)
}

// stand-alone launch, used when the feature pointer is not 16-byte aligned while the view pointer is (or vice versa)
template <bool VEC4>
__global__ void __launch_bounds__(256) k_features(WorldView W, RenderArgs R, RenderPlan P) {
    features_body<VEC4>(render_world(W, R.g), R, P, blockIdx.x, gridDim.x);
}

template <bool VEC4, bool NT, int U, bool PACKED, bool TURN>
__global__ void __launch_bounds__(64 * RENDER_WAVES) k_render(WorldView W, RenderArgs R, RenderPlan P) {
    render_block<VEC4, NT, U, PACKED, TURN>(render_world(W, R.g), R, P, blockIdx.x, gridDim.x);
}
template <bool PACKED, bool TURN>
__global__ void __launch_bounds__(64 * RENDER_WAVES) k_render_cells16(WorldView W, RenderArgs R, RenderPlan P) {
    render_block<true, true, 1, PACKED, TURN, true>(render_world(W, R.g), R, P, blockIdx.x, gridDim.x);
}

template <bool CELLS16>
__global__ void __launch_bounds__(64 * RENDER_WAVES) k_render_fast(RenderWorld W, RenderArgs R, RenderPlan P) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int VHW = R.VH * R.VW;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if ((int)blockIdx.x >= P.spans) {   // the trailing workgroups write the group's feature rows
        features_body<true>(W, R, P, blockIdx.x - P.spans, gridDim.x - P.spans);
        return;
    }
    float *strip = (float *)smem + (size_t)wave * (64 * 7);
    RenderFastPos *wtab = (RenderFastPos *)((float *)smem + RENDER_WAVES * 64 * 7);
    int4 *atab = (int4 *)(wtab + VHW);
    int span = blockIdx.x;
    if (P.xcd_chunk > 0 && span < P.xcd_chunk * 8) span = (span & 7) * P.xcd_chunk + (span >> 3);
    const GroupDev Gd = W.grp;
    const TypeDev T = W.type;
    const unsigned total_cells = (unsigned)R.n * (unsigned)VHW;
    const size_t total_floats = (size_t)total_cells * 7;
    const unsigned *vc = (const unsigned *)W.viewcell;
    const unsigned g = (unsigned)R.g;

    for (int c = threadIdx.x; c < VHW; c += 64 * RENDER_WAVES) {
        const int vy = fdiv_u32(c, P.div_vw), vx = c - vy * R.VW;
        RenderFastPos e;
        e.dxy = ((T.view_y1 + vy) << 16) | ((T.view_x1 + vx) & 0xFFFF);
        e.m0 = R.mini[(int)g * VHW + c];
        e.m1 = R.mini[(1 - (int)g) * VHW + c];
        e.mask = W.mask[T.mask_off + c];
        wtab[c] = e;
    }
    const int jump_a = (64 * RENDER_WAVES) / VHW, jump_c = (64 * RENDER_WAVES) - jump_a * VHW;   // a wave's next step is 256 cells on

    for (int b0 = 0; b0 < P.steps_per_span; b0 += RF_BLOCK_STEPS) {
        const unsigned step_b0 = (unsigned)span * P.steps_per_span + b0;
        const unsigned k_b0 = step_b0 * 64u;
        if (k_b0 >= total_cells) break;
        const int nsteps = min(RF_BLOCK_STEPS, P.steps_per_span - b0);
        const unsigned k_end = min(k_b0 + (unsigned)nsteps * 64u, total_cells);
        const int a0 = fdiv_u32(k_b0, P.div_vhw), a1 = fdiv_u32(k_end - 1u, P.div_vhw);
        __syncthreads();                                   // (the previous block's readers of atab are done)
        for (int i = threadIdx.x; i <= a1 - a0; i += 64 * RENDER_WAVES) {
            const int x = Gd.x[a0 + i], y = Gd.y[a0 + i];
            atab[i] = make_int4(x, y, (int)(fdiv_u32(y, P.div_scale_h) * R.VW + fdiv_u32(x, P.div_scale_w)), 0);
        }
        __syncthreads();

        // ---- the wave's first step of this block: index by division, view cell requested
        int it = wave;
        unsigned k = (step_b0 + it) * 64u + lane;
        int a = fdiv_u32(min(k, total_cells - 1u), P.div_vhw);
        int cell = (int)(min(k, total_cells - 1u) - (unsigned)a * VHW);
        RenderFastPos wt = wtab[cell];
        int4 at = atab[a - a0];
        unsigned v = VC_EMPTY;
        {
            const int mx = at.x + ((wt.dxy << 16) >> 16), my = at.y + (wt.dxy >> 16);
            if (it < nsteps && k < total_cells && wt.mask && (unsigned)mx < (unsigned)W.w && (unsigned)my < (unsigned)W.h) v = vc[my * W.w + mx];
        }
        for (; it < nsteps; it += RENDER_WAVES) {
            const unsigned k0 = (step_b0 + it) * 64u;
            if (k0 >= total_cells) break;
            // ---- next step: indices and the view-cell request (in flight while this step is expanded and stored)
            int a_n = a + jump_a, cell_n = cell + jump_c;
            if (cell_n >= VHW) { cell_n -= VHW; a_n++; }
            const unsigned k_n = k + 64u * RENDER_WAVES;
            const bool more = it + RENDER_WAVES < nsteps && k_n < total_cells;
            RenderFastPos wt_n = wt;
            int4 at_n = at;
            unsigned v_n = VC_EMPTY;
            if (more) {
                wt_n = wtab[cell_n];
                at_n = atab[a_n - a0];
                const int mx = at_n.x + ((wt_n.dxy << 16) >> 16), my = at_n.y + (wt_n.dxy >> 16);
                if (wt_n.mask && (unsigned)mx < (unsigned)W.w && (unsigned)my < (unsigned)W.h) v_n = vc[my * W.w + mx];
            }
            // ---- expand this step's cell: [wall | has, hp, minimap of the observing group | has, hp, minimap of the other]
            const unsigned top = v >> 30;
            const float hp = __uint_as_float(v & 0x3FFFFFFFu);
            const bool mine = top == g, theirs = top == 1u - g;
            float m0 = wt.m0, m1 = wt.m1;
            if (cell == at.z) { if (m0 == m0) m0 += 1.0f; if (m1 == m1) m1 += 1.0f; }   // self marker; NaN stays the same NaN
            const float c0 = v == VC_WALL ? 1.0f : 0.0f, c1 = mine ? 1.0f : 0.0f, c2 = mine ? hp : 0.0f;
            const float c4 = theirs ? 1.0f : 0.0f, c5 = theirs ? hp : 0.0f;
            if (CELLS16) {
                cell16_t o;
                o[0] = (__bf16)c0; o[1] = (__bf16)c1; o[2] = (__bf16)c2; o[3] = (__bf16)m0; o[4] = (__bf16)c4; o[5] = (__bf16)c5; o[6] = (__bf16)m1;
                o[7] = (__bf16)1.0f;
                if (k < total_cells) __builtin_nontemporal_store(o, (cell16_t *)R.view + k);
            } else {
                float *dst = strip + lane * 7;
                dst[0] = c0; dst[1] = c1; dst[2] = c2; dst[3] = m0; dst[4] = c4; dst[5] = c5; dst[6] = m1;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                const size_t f0 = (size_t)k0 * 7;
                const size_t remain = total_floats - f0;
                const int nq = remain >= (size_t)(64 * 7) ? 16 * 7 : (int)(remain >> 2);
                v4f *out4 = (v4f *)(R.view + f0);
                const v4f *src4 = (const v4f *)strip;
                for (int q = lane; q < nq; q += 64) __builtin_nontemporal_store(src4[q], out4 + q);
                if (remain < (size_t)(64 * 7))
                    for (int e = (nq << 2) + lane; e < (int)remain; e += 64) R.view[f0 + e] = strip[e];
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
            }
            a = a_n; cell = cell_n; k = k_n; wt = wt_n; at = at_n; v = v_n;
        }
    }
}
template <bool CELLS16, int DV, int SU, bool MINI = true>
__global__ void __launch_bounds__(64 * RENDER_WAVES) k_render_sweep2(RenderWorld W, RenderArgs R, RenderPlan P, int sweep) {
    render_sweep2_body<CELLS16, DV, SU, MINI>(W, R, P, sweep, blockIdx.x, gridDim.x);
}
// the shapes k_render_fast takes
bool render_sweep_mini_ok(const WorldView &W, const RenderArgs &R) {     // (launch.h: the float32 battle shape the sweeping kernel takes)
    const int VHW = R.VH * R.VW;
    return W.G == 2 && R.minimap && !R.food && R.C == 7 && W.vc_packed && !R.turn && !R.cells16 && VHW >= 16 && VHW <= 1024;
}
static bool render_fast_ok(const WorldView &W, const RenderArgs &R, const RenderPlan &P, bool vec4) {
    const int VHW = R.VH * R.VW;
    return vec4 && W.G == 2 && R.minimap && !R.food && R.C == 7 && W.vc_packed && !R.turn && VHW >= 16 && VHW <= 1024 &&
           render_fast_lds(VHW) <= 48 * 1024;
}

// several groups of a small world in one launch (blockIdx.y = slot): small worlds are bound by the number of launches
template <bool PACKED>
__global__ void __launch_bounds__(64 * RENDER_WAVES) k_render_multi(WorldView W, RenderMulti M) {
    const int k = blockIdx.y;
    if ((int)blockIdx.x >= M.blocks[k]) return;
    if (W.turn_mode) render_block<true, true, 1, PACKED, true>(render_world(W, M.R[k].g), M.R[k], M.P[k], blockIdx.x, M.blocks[k]);
    else render_block<true, true, 1, PACKED, false>(render_world(W, M.R[k].g), M.R[k], M.P[k], blockIdx.x, M.blocks[k]);
}

// the painted map streamed through the caches ahead of the renders of a map that does not fit the L2s (Env::observe_device: when, and why)
__global__ void __launch_bounds__(256) k_touch(const uint4 *p, size_t n16, unsigned *sink) {
    unsigned acc = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) { const uint4 v = p[i]; acc ^= v.x ^ v.y ^ v.z ^ v.w; }
    if (acc == 0x9E3779B9u) *sink = acc;      // (never, in practice: keeps the loads)
}
void launch_touch_map(hipStream_t s, const WorldView &W) {
    const size_t bytes = (size_t)W.w * W.h * (W.vc_packed ? 4 : 8);
    hipLaunchKernelGGL(k_touch, dim3(2048), dim3(256), 0, s, (const uint4 *)W.viewcell, bytes / 16, (unsigned *)W.counters + CTR_GOALS_ACT);
}

void launch_paint(hipStream_t s, const WorldView &W, const GroupDev *gtab, const TypeDev *ttab) {
    int ncell = W.w * W.h;
    int blocks = (ncell + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    if (W.vc_packed) hipLaunchKernelGGL(k_paint<true>, dim3(blocks), dim3(256), 0, s, W, gtab, ttab);
    else hipLaunchKernelGGL(k_paint<false>, dim3(blocks), dim3(256), 0, s, W, gtab, ttab);
}

void launch_minimap(hipStream_t s, const WorldView &W, const RenderArgs &R, int *counts, float *mini) {
    int VHW = R.VH * R.VW;
    const int skip = W.type[R.g].can_absorb;
    int *left_out = counts;      // the first MAXG ints of the buffer; the histogram follows
    counts += MAXG;
    if (skip) (void)hipMemsetAsync(left_out, 0, sizeof(int) * MAXG, s);
    int mx = 1;
    for (int g = 0; g < W.G; g++) mx = W.grp[g].n > mx ? W.grp[g].n : mx;
    int bx = (mx + 255) / 256;
    if (bx > 128) bx = 128;      // every block ends with one global atomic per non-empty bin: few, fat blocks
    // (folding the normalisation into the histogram's last block -- ticket counter -- was measured: +12 us, the tickets serialise)
    hipLaunchKernelGGL(k_minimap, dim3(bx, W.G), dim3(256), VHW * sizeof(int), s, W, R, counts, left_out, skip);
    hipLaunchKernelGGL(k_minimap_norm, dim3((W.G * VHW + 255) / 256), dim3(256), 0, s, R, W.G, counts, mini, left_out, skip);
}

// returns the kernel taken: 0 k_render / k_render_cells16 (every game), 1 k_render_fast, 4 k_render_sweep2
int launch_render(hipStream_t s, const WorldView &W, const RenderArgs &R, const RenderPlan &P, bool vec4, bool nt) {
    if (R.n <= 0) return 0;
    size_t lds = (size_t)RENDER_WAVES * P.strip_floats * sizeof(float);
    dim3 grid(P.spans + P.feat_blocks), block(64 * RENDER_WAVES);
    const bool packed = W.vc_packed != 0;   // must match launch_paint
    // the sweeping kernel takes the two-group shapes with packed view cells: [wall | has, hp, minimap | ...] (battle, gather: also as bf16 cells and
    // through k_render_fast) and, since round 5, [wall | has, hp | has, hp] (no minimap channels: the reference's pursuit-like 1M harness)
    const bool fast_ok = render_fast_ok(W, R, P, vec4 && nt);
    const bool sweep5_ok = vec4 && nt && W.G == 2 && !R.minimap && !R.food && R.C == 5 && W.vc_packed && !R.turn && !R.cells16 && R.VH * R.VW >= 16 && R.VH * R.VW <= 1024;
    if (fast_ok || sweep5_ok) {
        // MAGENT_TUNE render: 0 generic kernels | 1 k_render_fast | 4 k_render_sweep2 | unset: bf16 cells -> 1; float32 -> 4 at scale, else generic
        static const int forced = tune("render", -1);
        static const int sweep_fixed = tune("render_sweep", 0);   // (tests: few workgroups, many rounds)
        static const int su_env = tune("render_su", 2);
        static const int dv_env = tune("render_depth", 2);
        const long long steps = ((long long)R.n * R.VH * R.VW + 63) / 64;
        const int VHW = R.VH * R.VW;
        const int Cc = fast_ok ? 7 : 5;
        int mode = forced;
        if (mode < 0) mode = R.cells16 ? 1 : (steps >= 256ll * RENDER_WAVES * 2 * 8 ? 4 : 0);   // (a sweep wants >= 8 rounds of 256 workgroups)
        if (mode == 1 && !fast_ok) mode = 0;
        if (mode == 4) {
            const int SUv = su_env >= 3 ? 3 : su_env >= 2 ? 2 : 1;
            const int sweep = (int)std::min<long long>(sweep_fixed > 0 ? sweep_fixed : 256, (steps + RENDER_WAVES * SUv - 1) / (RENDER_WAVES * SUv));
            dim3 sgrid(sweep + P.feat_blocks);
            const size_t sl = (size_t)RENDER_WAVES * SUv * 64 * Cc * sizeof(float) + (size_t)VHW * sizeof(RenderFastPos);
            const RenderPlan &Ps = P;
#define SW2(C16, DVV, SUV, MI) hipLaunchKernelGGL((k_render_sweep2<C16, DVV, SUV, MI>), sgrid, block, sl, s, render_world(W, R.g), R, Ps, sweep)
#define SW2D(C16, SUV, MI) do { if (dv_env <= 1) SW2(C16, 1, SUV, MI); else if (dv_env == 2) SW2(C16, 2, SUV, MI); else SW2(C16, 3, SUV, MI); } while (0)
            if (!fast_ok) { if (SUv == 3) SW2D(false, 3, false); else if (SUv == 2) SW2D(false, 2, false); else SW2D(false, 1, false); }
            else if (R.cells16) { if (SUv == 3) SW2D(true, 3, true); else if (SUv == 2) SW2D(true, 2, true); else SW2D(true, 1, true); }
            else { if (SUv == 3) SW2D(false, 3, true); else if (SUv == 2) SW2D(false, 2, true); else SW2D(false, 1, true); }
#undef SW2D
#undef SW2
            return 4;
        }
        if (mode == 1) {
            const size_t fl = render_fast_lds(VHW);
            if (R.cells16) hipLaunchKernelGGL((k_render_fast<true>), grid, block, fl, s, render_world(W, R.g), R, P);
            else hipLaunchKernelGGL((k_render_fast<false>), grid, block, fl, s, render_world(W, R.g), R, P);
            return 1;
        }
    }
    if (R.cells16) {
        if (R.turn) { if (packed) hipLaunchKernelGGL((k_render_cells16<true, true>), grid, block, lds, s, W, R, P); else hipLaunchKernelGGL((k_render_cells16<false, true>), grid, block, lds, s, W, R, P); }
        else { if (packed) hipLaunchKernelGGL((k_render_cells16<true, false>), grid, block, lds, s, W, R, P); else hipLaunchKernelGGL((k_render_cells16<false, false>), grid, block, lds, s, W, R, P); }
        return 0;
    }
#define RENDER_LAUNCH(V, N, UU, PK) hipLaunchKernelGGL((k_render<V, N, UU, PK, false>), grid, block, lds, s, W, R, P)
#define RENDER_PK(V, N, UU) do { if (packed) RENDER_LAUNCH(V, N, UU, true); else RENDER_LAUNCH(V, N, UU, false); } while (0)
    if (R.turn) {      // turn_mode: one step per wave iteration, scalar or 16-byte stores
        if (vec4 && packed) hipLaunchKernelGGL((k_render<true, true, 1, true, true>), grid, block, lds, s, W, R, P);
        else if (vec4) hipLaunchKernelGGL((k_render<true, true, 1, false, true>), grid, block, lds, s, W, R, P);
        else if (packed) hipLaunchKernelGGL((k_render<false, false, 1, true, true>), grid, block, lds, s, W, R, P);
        else hipLaunchKernelGGL((k_render<false, false, 1, false, true>), grid, block, lds, s, W, R, P);
    } else if (!vec4) RENDER_PK(false, false, 1);
    else RENDER_PK(true, true, 1);       // (16-byte nontemporal stores; two / four steps in flight per wave and plain stores measured no better: removed)
    (void)nt;
#undef RENDER_PK
#undef RENDER_LAUNCH
    return 0;
}

void launch_render_multi(hipStream_t s, const WorldView &W, const RenderMulti &M) {
    size_t lds = 0;
    int mx = 0;
    for (int k = 0; k < M.n; k++) { lds = std::max(lds, (size_t)RENDER_WAVES * M.P[k].strip_floats * sizeof(float)); mx = std::max(mx, M.blocks[k]); }
    if (M.n <= 0 || mx <= 0) return;
    dim3 grid(mx, M.n), block(64 * RENDER_WAVES);
    if (W.vc_packed) hipLaunchKernelGGL((k_render_multi<true>), grid, block, lds, s, W, M);
    else hipLaunchKernelGGL((k_render_multi<false>), grid, block, lds, s, W, M);
}

void launch_features(hipStream_t s, const WorldView &W, const RenderArgs &R, const RenderPlan &P, bool vec4) {
    if (R.n <= 0) return;
    unsigned total = (unsigned)R.n * (unsigned)R.F;
    int fb = (int)std::min<unsigned>((total / 4 + 255) / 256 + 1, 16384);   // ~1 float4 per thread: latency-bound gathers
    if (vec4) hipLaunchKernelGGL((k_features<true>), dim3(fb), dim3(256), 0, s, W, R, P);
    else hipLaunchKernelGGL((k_features<false>), dim3(fb), dim3(256), 0, s, W, R, P);
}
size_t render_strip_lds(const RenderPlan &P) { return (size_t)RENDER_WAVES * P.strip_floats * sizeof(float); }

}  // namespace magent_amd
