// policy.hip -- the reference's deep Q network (python/magent/builtin/tf_model/dqn.py:151-189), inference only, as two
// hand-written bf16 MFMA kernels for gfx950.  This is the caller on the far side of the hot path (SURVEY.md 8f rank 1:
// BASELINE config 5 puts a policy between get_observation and set_action); with the PyTorch / MIOpen network a 2 x 400k
// self-play step is 42-48 ms of which the engine is 1.5; with these kernels 5.2 (profiles/r02_policy.txt).
//
//   network:  view [n][H][W][C] f32 (or the engine's bf16 cells, env_get_observation_device_bf16) -> conv3x3(32, valid) relu -> conv3x3(32, valid) relu -> flatten (NHWC) -> dense 256 relu
//             feature [n][F] f32 -> dense 256 relu;  concat 512 -> advantage (n_action, no bias) and value (1);
//             Q = value + advantage - mean(advantage)
//   numerics: inputs, weights and the activations between layers are rounded to bf16 (round to nearest even), every product
//             is accumulated in f32 by v_mfma_f32_32x32x16_bf16, biases are added in f32 (conv1's rides in the MFMA as the
//             weight of a constant-1 channel: bf16).  tests/test_policy.py compares with
//             a PyTorch f32 computation that rounds at the same points.
//
// k_dqn_conv : conv1 + conv2 fused.  A workgroup takes TA agents at a time: their views go to LDS as bf16 with the channels
//   padded to 8 (one window cell = one 16-byte MFMA operand), conv1's output stays in LDS, conv2's goes to HBM as bf16.
//   Convolutions are implicit GEMMs with M = output positions, N = 32 channels, K = taps x channels; the 32 x 32 weight
//   tiles of both layers live in REGISTERS for the life of the wave (92 VGPRs), so an MFMA costs one 16-byte LDS read per
//   lane.  The weights are the first MFMA operand: the result tile then has the output channels down the registers and the
//   positions across the lanes, i.e. a lane owns 16 channels of ONE position and stores them as two 16-byte vectors.
//   (Which 16: ch_of() below.  Activations are kept in that "slot" order; the next layer's weights are permuted to match when
//   they are packed -- magent_amd/builtin/torch_model/hip_policy.py.)
// k_dqn_head : dense 2592 -> 256 as a GEMM over 128 agents per workgroup of 8 waves (activations through LDS, packed weights
//   straight from L2 in fragment order, both streams four K-chunks ahead in registers), the feature embedding, the dueling head
//   and the argmax, fused.
//
// Weight layouts ("fragment order"): for every k-step s (16 values of K) and 32-wide output tile, 64 lanes x 8 bf16 --
// lane l holds W[out = l & 31][k = 16 s + 8 (l >> 5) + 0..7], exactly the first operand of v_mfma_f32_32x32x16_bf16
// (lane map verified on the hardware by tools/probe/mfma_layout.hip).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <type_traits>
#include <vector>

#include "../../include/magent_policy.h"
#include "tune.h"

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

// result register r of lane group g (= lane >> 5) is row (r & 3) + 8 (r >> 2) + 4 g of the 32 x 32 tile
__device__ __forceinline__ int ch_of(int g, int r) { return (r & 3) + 8 * (r >> 2) + 4 * g; }

// conv2's output, dense_view's input: [group of ACT_GROUP agents][K-chunk of 64 values = two positions][agent][64 values].  A k_dqn_head
// workgroup (ACT_GROUP agents) then reads one contiguous 16 KB block per K-chunk, and its blocks follow each other -- round 2 kept
// [agent][K]: 128-byte pieces 5 KB apart, and the head waited for them (without its activation loads the head ran 0.064 ms faster of
// 0.205 -- the pieces came at the rate HBM serves scattered lines, not at its streaming rate).
constexpr int ACT_GROUP = 128;
__device__ __forceinline__ size_t act_at(int agent, int pos, int n_pos) {       // index (in bf16 values) of slot 0 of `pos` of `agent`
    const int chunks = (n_pos + 1) >> 1;
    return ((size_t)(agent / ACT_GROUP) * chunks + (pos >> 1)) * (ACT_GROUP * 64) + (size_t)(agent % ACT_GROUP) * 64 + (pos & 1) * 32;
}
constexpr int CONV_THREADS = 256, CONV_TA = 4, CONV_CELLS = 4;   // CONV_TA agents per tile; CONV_CELLS: window cells of a tile per thread (register prefetch)

struct ConvArgs {
    const float *view;     // [n][H][W][C] f32, or (CELLS16) [n][H][W][8] bf16 cells as env_get_observation_device_bf16 writes them
    __bf16 *act;           // [n / 128][K / 64][128 agents][64 values] (ACT_GROUP): what one k_dqn_head workgroup reads per K-chunk is one 16 KB block
    const bf16x8 *w1, *w2; // fragment order: [5][64], [18][64]; conv1's bias sits in w1 at (tap 0, channel 7)
    const float *b2;       // [2][16]: conv2's bias of the channel in slot 16 g + r
    int n, H, W, C, VP, AP, n_tiles;
    bf16x8 *dump;          // 2 KB behind the workspace: where lanes without a conv2 position store
    long long *stamps;     // STAMPS instantiation (MAGENT_TUNE policy_stamps=1): per workgroup, cycles spent in each phase
};

// relu and round two f32 to a bf16 pair: v_cvt_pk_bf16_f32, then v_pk_max_i16 against 0 -- a negative bf16 is a negative int16, and
// rounding never changes a sign (-0 becomes +0), so max(int16(round(x)), 0) == round(max(x, 0)) bit for bit.  Two instructions per
// pair; `(__bf16)fmaxf(x, 0)` is five (two v_max_f32 each -- one quiets NaNs -- and the conversion).
__device__ __forceinline__ unsigned relu_bf16x2(float a, float b) {
    typedef __attribute__((ext_vector_type(2))) float f32x2;
    typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
    typedef __attribute__((ext_vector_type(2))) short s16x2;
    const f32x2 v = {a, b};
    s16x2 t = __builtin_bit_cast(s16x2, __builtin_convertvector(v, bf16x2));
    t = __builtin_elementwise_max(t, (s16x2)(0));
    return __builtin_bit_cast(unsigned, t);
}
__device__ __forceinline__ bf16x8 relu_bf16x8(const f32x16 &acc, int base) {
    typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
    u32x4 u;
#pragma unroll
    for (int r = 0; r < 4; r++) u[r] = relu_bf16x2(acc[base + 2 * r], acc[base + 2 * r + 1]);
    return __builtin_bit_cast(bf16x8, u);
}

// Round 3 layout.  The kernel was bound by instruction issue, not by the matrix pipe (per-phase cycle stamps: 1430 instructions per
// wave and tile around 106 MFMAs -- 15 vector instructions per MFMA in conv1, 7 in conv2: look-up tables, swizzled addresses for every
// tap, a five-instruction ReLU).  Now:
//   * the 32 lanes of an MFMA tile are 8 CONSECUTIVE POSITIONS x 4 AGENTS (position-major, agent-minor: lane index E -> position
//     E >> 2, agent E & 3).  With the agent pitches of both LDS images == 4 (mod 16) sixteen-byte slots, every 16-lane service group of a
//     ds_read_b128 (MI355X_MICROARCH.md, LDS) lands on 16 different slots WITHOUT any swizzle -- agents contribute 0, 4, 8, 12, the
//     group's four positions four different residues mod 4 (a row wrap of conv2's 9-wide rows in the 13-wide image adds W - W2 = 4) --
//     so a tap is a constant offset from a per-lane base: an immediate of the ds_read when the view is the 13 x 13 one (F13).
//   s_view [4][VP] cells of 8 bf16: channels 0..C-1, zeros, 1.0 in channel 7 (conv1's bias is the weight of that constant); VP >= H W + 3,
//          the cells past H W are zero (the padding tap and the garbage columns below read them).  conv1 is evaluated over FULL rows
//          of W positions (the last two of a row are garbage that lands in columns conv2 never reads): a position's index IS its
//          top-left cell's index, and its index in conv1's output image.
//   s_c1   [4 planes][4 agents][AP positions] x 8 bf16: plane c holds slots 8 c .. 8 c + 7 of every position (conv1's result lane
//          (position, g) writes planes 2 g and 2 g + 1; conv2's operand for (tap, half h) of lane group g is plane 2 h + g).
//   * conv1's ten taps (nine and a zero-weight one) are paired (0|3) (1|4) (2|5) (6|7) (8|pad): lane group 1 reads W cells (first three
//     k-steps) or 1 cell (last two) after lane group 0 -- two bases per tile, the rest immediates.  The padding tap reads cell + 2W + 3:
//     a real or a zeroed cell, never uninitialised LDS (NaN x 0).
//   * no look-up tables: positions are shifts and one division by a constant.
template <bool CELLS16, bool F13, bool STAMPS = false>     // CELLS16: the views arrive as bf16 cells -- conv1's operands as they are.  F13: 13 x 13 views
__global__ void __launch_bounds__(CONV_THREADS) k_dqn_conv(ConvArgs A) {
    extern __shared__ __attribute__((aligned(16))) unsigned char s_raw[];
    constexpr int TA = CONV_TA, NW = CONV_THREADS / 64;
    const int H = F13 ? 13 : A.H, W = F13 ? 13 : A.W, VP = F13 ? 180 : A.VP, AP = F13 ? 148 : A.AP;
    const int H1 = H - 2, H2 = H - 4, W2 = W - 4, HW = H * W, NP2 = H2 * W2;
    const int E1 = TA * H1 * W, P2 = TA * NP2, PL = TA * AP;
    bf16x8 *s_view = (bf16x8 *)s_raw;                     // [TA][VP]
    bf16x8 *s_c1 = s_view + TA * VP;                      // [4][TA][AP]
    float *s_bias = (float *)(s_c1 + 4 * PL);             // [2][16]

    const int tid = threadIdx.x, l = tid & 63, w = tid >> 6, g = l >> 5, r32 = l & 31;
    bf16x8 wf1[5], wf2[18];
#pragma unroll
    for (int s = 0; s < 5; s++) wf1[s] = A.w1[s * 64 + l];
#pragma unroll
    for (int s = 0; s < 18; s++) wf2[s] = A.w2[s * 64 + l];
    for (int k = tid; k < 32; k += CONV_THREADS) s_bias[k] = A.b2[k];
    for (int c = tid; c < TA * VP; c += CONV_THREADS) s_view[c] = bf16x8{0};      // (the cells behind every agent's H W stay zero)

    // A tile's window cells are fetched a whole tile AHEAD, into registers: the loads of tile t + 1 are issued before the
    // convolutions of tile t and waited for after them, so their HBM latency hides behind the MFMA work.  A thread owns whole
    // cells: C floats (or one bf16 cell) in, one 16-byte LDS store out.  Cell c of a tile (agent-major, as the views lie in memory)
    // goes to s_view[(c / HW) * VP + c % HW].
    float nv[CONV_CELLS][7];
    // (the loads are unconditional, from clamped addresses: a select on a loaded value would make the wave wait for it at once)
    // (seven channels -- every reference game with a minimap and two groups -- are fetched as one 16-byte and one 12-byte load per
    // cell: a wave's lanes are 28 bytes apart, so every load instruction walks 14 cache lines whatever its width, and seven
    // 4-byte loads per cell kept the CU's one address pipeline busy for a quarter of the kernel)
    typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));
    typedef float f32x3u __attribute__((ext_vector_type(3), aligned(4)));
    bf16x8 nc[CONV_CELLS];
    int sdst[CONV_CELLS];
#pragma unroll
    for (int k = 0; k < CONV_CELLS; k++) { const int c = k * CONV_THREADS + tid, a = c / HW; sdst[k] = a < TA ? a * VP + (c - a * HW) : -1; }
    auto fetch = [&](int tile) {
        const int live = min(TA, A.n - tile * TA) * HW;
        if (CELLS16) {
            const bf16x8 *src16 = (const bf16x8 *)A.view + (size_t)tile * TA * HW;
#pragma unroll
            for (int k = 0; k < CONV_CELLS; k++) nc[k] = src16[min(k * CONV_THREADS + tid, live - 1)];
            return;
        }
        const float *src = A.view + (size_t)tile * TA * HW * A.C;
#pragma unroll
        for (int k = 0; k < CONV_CELLS; k++) {
            const float *p = src + (size_t)min(k * CONV_THREADS + tid, live - 1) * A.C;
            if (A.C == 7) {
                const f32x4u lo = *(const f32x4u *)p;
                const f32x3u hi = *(const f32x3u *)(p + 4);
                nv[k][0] = lo[0]; nv[k][1] = lo[1]; nv[k][2] = lo[2]; nv[k][3] = lo[3]; nv[k][4] = hi[0]; nv[k][5] = hi[1]; nv[k][6] = hi[2];
            } else {
#pragma unroll
                for (int e = 0; e < 7; e++) nv[k][e] = p[min(e, A.C - 1)];
            }
        }
    };
    // the window cells in registers (a tile's, fetched earlier) -> s_view as conv1's operands
    auto stage = [&](int tile) {
        const int live = min(TA, A.n - tile * TA) * HW;
#pragma unroll
        for (int k = 0; k < CONV_CELLS; k++) {
            const bool have = k * CONV_THREADS + tid < live;
            bf16x8 v;
            if (CELLS16) {
                v = nc[k];
                if (!have) { v = bf16x8{0}; v[7] = (__bf16)1.0f; }
            } else {
#pragma unroll
                for (int e = 0; e < 7; e++) v[e] = (__bf16)((have && e < A.C) ? nv[k][e] : 0.0f);
                v[7] = (__bf16)1.0f;
            }
            if (sdst[k] >= 0) s_view[sdst[k]] = v;
        }
    };
    // Order of a tile's memory traffic.  Loads and stores share one counter on this part and complete out of order with respect
    // to each other, so WAITING FOR A LOAD ALSO WAITS FOR EVERY STORE IN FLIGHT (profiles/r03_render_experiments.md).  The views of tile
    // t + 1 go to LDS in the MIDDLE of tile t: conv1 has finished with s_view, the loads were issued a conv2 + a conv1 ago, and the only
    // stores in flight are the previous tile's, a conv1 old; the stores of tile t then have the whole conv1 of tile t + 1 to land
    // before anybody waits again.
    // (The compiler's wait-count pass merges control-flow paths pessimistically.  Every wait for the weight loads above sat inside some
    // branch, so there was a path on which they were still in flight at the head of the tile loop -- and the loop then waited for
    // vmcnt(4) / vmcnt(0), i.e. for the view prefetch issued a moment earlier and for the previous tile's stores, in the middle of
    // EVERY conv1 and conv2.  An explicit wait here retires the weights on every path; it costs one round trip per workgroup.)
    __builtin_amdgcn_s_waitcnt(0x0F70);      // vmcnt(0): the weights are in their registers, on every path into the loop
    __syncthreads();                         // (the zero fill of s_view is complete before the first cells land)
    fetch(blockIdx.x); stage(blockIdx.x);
    if ((int)(blockIdx.x + gridDim.x) < A.n_tiles) fetch(blockIdx.x + gridDim.x);

    const int T1 = (E1 + 31) / 32, T2 = (P2 + 31) / 32;
    long long t_bar = 0, t_c1 = 0, t_stage = 0, t_c2 = 0, t_a = 0, t_b = 0;      // (STAMPS: wave 0's cycle counter around the phases)
    const long long t_begin = STAMPS ? clock64() : 0;
    for (int tile = blockIdx.x; tile < A.n_tiles; tile += gridDim.x) {
        const int a0 = tile * TA;
        if (STAMPS) t_a = clock64();
        __syncthreads();     // s_view holds this tile (staged in the middle of the tile before); the previous conv2's readers of s_c1 are done
        if (STAMPS) { t_b = clock64(); t_bar += t_b - t_a; }
        // ---- conv1: [E1 positions, full rows] x [32 channels], K = 10 taps x 8 channels (bias: the constant channel of tap 0).
        // A wave runs TWO position tiles at a time (t and t + 4): two independent accumulator chains keep the matrix pipe busy
        // while the other chain's operands are on their way from LDS.
        auto conv1 = [&](int t, auto two_c) __attribute__((always_inline)) {
            constexpr bool two = decltype(two_c)::value;
            const int Ea = min(t * 32 + r32, E1 - 1), Eb = min((t + NW) * 32 + r32, E1 - 1);
            const int pa = Ea >> 2, pb = Eb >> 2, aa = Ea & 3, ab = Eb & 3;         // position (= top-left cell = c1 position), agent
            const bf16x8 *va = s_view + aa * VP + pa + g * W, *vb = s_view + ab * VP + pb + g * W;        // taps (0|3) (1|4) (2|5)
            const bf16x8 *ua = s_view + aa * VP + pa + g + 2 * W, *ub = s_view + ab * VP + pb + g + 2 * W;  // taps (6|7) (8|pad)
            bf16x8 xa[5], xb[5];
            xa[0] = va[0]; xa[1] = va[1]; xa[2] = va[2]; xa[3] = ua[0]; xa[4] = ua[2];
            if (two) { xb[0] = vb[0]; xb[1] = vb[1]; xb[2] = vb[2]; xb[3] = ub[0]; xb[4] = ub[2]; }
            f32x16 acca = {0}, accb = {0};
#pragma unroll
            for (int s = 0; s < 5; s++) {
                acca = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf1[s], xa[s], acca, 0, 0, 0);
                if (two) accb = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf1[s], xb[s], accb, 0, 0, 0);
            }
            bf16x8 *da = s_c1 + (2 * g) * PL + aa * AP + pa;       // (lanes past E1 repeat the last position: same values, harmless)
            da[0] = relu_bf16x8(acca, 0); da[PL] = relu_bf16x8(acca, 8);
            if (two) {
                bf16x8 *db = s_c1 + (2 * g) * PL + ab * AP + pb;
                db[0] = relu_bf16x8(accb, 0); db[PL] = relu_bf16x8(accb, 8);
            }
        };
        {
            int t = w;
            for (; t + NW < T1; t += 2 * NW) conv1(t, std::true_type{});
            if (t < T1) conv1(t, std::false_type{});
        }
        if (STAMPS) { t_a = clock64(); t_c1 += t_a - t_b; }
        __syncthreads();
        if (STAMPS) { t_b = clock64(); t_bar += t_b - t_a; }
        // ---- conv1 is done with s_view: the next tile's views move in, the one after is requested (see above)
        if (tile + (int)gridDim.x < A.n_tiles) {
            stage(tile + gridDim.x);
            if (tile + 2 * (int)gridDim.x < A.n_tiles) fetch(tile + 2 * gridDim.x);
        }
        if (STAMPS) { t_a = clock64(); t_stage += t_a - t_b; }
        // ---- conv2: [P2 positions] x [32 channels], K = 9 taps x 32 slots; starts from the bias, result straight to HBM.
        // Two tiles per wave at a time here too; the four operands of tap + 1 are read while the MFMAs of tap run (the scheduling
        // barriers keep the compiler from sinking the reads to where their values are used).
        auto conv2 = [&](int t, auto two_c) __attribute__((always_inline)) {
            constexpr bool two = decltype(two_c)::value;
            const int Qa = t * 32 + r32, Qb = (t + NW) * 32 + r32;
            const int Qca = min(Qa, P2 - 1), Qcb = min(Qb, P2 - 1);
            const int qa = Qca >> 2, qb = Qcb >> 2, aa = Qca & 3, ab = Qcb & 3;
            const int ya = qa / W2, yb = qb / W2;
            const bf16x8 *ca = s_c1 + g * PL + aa * AP + ya * W + (qa - ya * W2);      // plane g (and g + 2) of the top-left c1 position
            const bf16x8 *cb = s_c1 + g * PL + ab * AP + yb * W + (qb - yb * W2);
            f32x16 acca, accb;
            {
                const f32x4 *bp = (const f32x4 *)(s_bias + g * 16);
#pragma unroll
                for (int q4 = 0; q4 < 4; q4++) {
                    const f32x4 bq = bp[q4];
#pragma unroll
                    for (int i = 0; i < 4; i++) { acca[4 * q4 + i] = bq[i]; accb[4 * q4 + i] = bq[i]; }
                }
            }
            bf16x8 x[2][4];
            auto xread = [&](int tap, bf16x8 (&dst)[4]) {
                const int off = (tap / 3) * W + tap % 3;
                dst[0] = ca[off]; dst[1] = ca[off + 2 * PL];
                if (two) { dst[2] = cb[off]; dst[3] = cb[off + 2 * PL]; }
            };
            xread(0, x[0]);
#pragma unroll
            for (int tap = 0; tap < 9; tap++) {
                if (tap < 8) xread(tap + 1, x[(tap + 1) & 1]);
                __builtin_amdgcn_sched_barrier(0);
                acca = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf2[2 * tap], x[tap & 1][0], acca, 0, 0, 0);
                if (two) accb = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf2[2 * tap], x[tap & 1][2], accb, 0, 0, 0);
                acca = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf2[2 * tap + 1], x[tap & 1][1], acca, 0, 0, 0);
                if (two) accb = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf2[2 * tap + 1], x[tap & 1][3], accb, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            {
                bf16x8 *dst = (Qa < P2 && a0 + aa < A.n) ? (bf16x8 *)(A.act + act_at(a0 + aa, qa, NP2) + 16 * g) : A.dump + 2 * l;
                dst[0] = relu_bf16x8(acca, 0); dst[1] = relu_bf16x8(acca, 8);
            }
            if (two) {
                bf16x8 *dst = (Qb < P2 && a0 + ab < A.n) ? (bf16x8 *)(A.act + act_at(a0 + ab, qb, NP2) + 16 * g) : A.dump + 2 * l;
                dst[0] = relu_bf16x8(accb, 0); dst[1] = relu_bf16x8(accb, 8);
            }
        };
        {
            int t = w;
            for (; t + NW < T2; t += 2 * NW) conv2(t, std::true_type{});
            if (t < T2) conv2(t, std::false_type{});
        }
        if (STAMPS) t_c2 += clock64() - t_a;
    }
    if (STAMPS && tid == 0) {
        long long *o = A.stamps + (size_t)blockIdx.x * 8;
        o[0] = t_bar; o[1] = t_c1; o[2] = t_stage; o[3] = t_c2; o[4] = clock64() - t_begin; o[5] = t_begin;
    }
}

// ---------------------------------------------------------------------------------------------------- dense + head
constexpr int HEAD_THREADS = 512, HEAD_M = 128, HEAD_KC = 64;   // 128 agents per workgroup of 8 waves; K staged 64 at a time
#ifndef HEAD_RING
#define HEAD_RING 4
#endif
constexpr int HEAD_ABUF = HEAD_M * (HEAD_KC / 8);                  // 16-byte units of one activation buffer (128 agents x 64 K values)
constexpr size_t HEAD_LDS = (3 * HEAD_ABUF + HEAD_M * 32 + 32 * 64) * 16 + 512 * 4;   // 3 x 16 KB activations + 64 KB hidden + 32 KB head weights + 2 KB biases = 146 KB

struct HeadArgs {
    const __bf16 *act;        // [n / 128][K / 64][128][64] (act_at; K = H2 * W2 * 32, slot order)
    const float *feat;        // [n][F]
    const bf16x8 *wv;         // dense_view, fragment order [K / 16][8 tiles][64]
    const bf16x8 *we;         // dense_emb,  fragment order [FK / 16][8 tiles][64]   (FK = F rounded up to 16)
    const bf16x8 *wh;         // head, fragment order [32][64]: K = 512 hidden slots, outputs 0..n_action-1 advantage, n_action value
    const float *bv, *be;     // [8 tiles][2][16]: bias of the output in slot 16 g + r of its tile
    float value_bias;
    int n, K, F, FK, n_action;
    int *actions;             // [n] argmax_a Q
    float *q;                 // [n][n_action] or null
    long long *stamps;        // STAMPS instantiation: per workgroup, the cycle counter at the phase boundaries
};

// one half of the hidden layer (256 values: relu(dense_view), later relu(dense_emb)) of the 128 agents: [agent][32 chunks of
// 8 bf16], chunk index xor-ed with the agent's low bits (a 16-lane group of ds_read_b128 then covers all 16 slots of the LDS).
// The head is accumulated half by half, so one buffer does for both.
__device__ __forceinline__ int hid_at(int agent, int chunk) { return agent * 32 + (chunk ^ (agent & 15)); }

// Wave w owns output tile w (32 of the 256 outputs) for all four agent tiles: per k-step ONE weight fragment from L2 and four
// activation operands from LDS feed four MFMAs.  Both global streams run ahead of the MFMAs through rings of four register
// sets: the wave's weight fragments (L2: ~0.5 us away) and the workgroup's activations (HBM: 1.5-2 us), each FOUR 64-wide
// K-chunks ahead; a chunk is 0.26 us of MFMA work per wave.
// (History per 131072 agents, 64 agents / 4 waves per workgroup: operands loaded at the k-step that uses them 0.45 ms, 79 % of
// the wave cycles waiting; one chunk ahead 0.31 ms -- two weight fragments per k-step left room for one chunk of look-ahead only.)
template <bool STAMPS>
__global__ void __launch_bounds__(HEAD_THREADS) k_dqn_head(HeadArgs A) {
    const long long t_begin = STAMPS ? clock64() : 0;
    extern __shared__ __attribute__((aligned(16))) unsigned char s_raw[];
    bf16x8 *s_act = (bf16x8 *)s_raw;                                                           // [3][agent][8 chunks], swizzled
    bf16x8 *s_hid = s_act + 3 * HEAD_ABUF;                                                     // [agent][32 chunks], swizzled
    bf16x8 *s_wh = s_hid + HEAD_M * 32;                                                        // the head's weights, fragment order [32][64]
    const int tid = threadIdx.x, l = tid & 63, w = tid >> 6, g = l >> 5, r32 = l & 31;
    const int a0 = blockIdx.x * HEAD_M;

    const int srow = tid >> 2, spiece = tid & 3;       // staging: thread t -> row t / 4, 32-byte piece t % 4 of a 128-byte chunk row
    const int sdst = srow * 8, ssw = (srow >> 1) & 7, rsw = (r32 >> 1) & 7;
    const bf16x8 *wbase = A.wv + (size_t)w * 64 + l;   // fragment (k-step s, tile w) = wbase[s * 8 * 64]
    const int n_steps = A.K / 16;                      // k-steps in all; the last chunk may be half (K is a multiple of 32)
    const int total = (n_steps + 3) / 4;
    static_assert(HEAD_M == ACT_GROUP && HEAD_KC == 64 && HEAD_THREADS == 512, "one workgroup reads one block of the activation layout per chunk");
    // activation block of a chunk: 1024 sixteen-byte units (unit u: agent row u >> 3, piece u & 7).  Thread t moves units t and 512 + t:
    // every load instruction of a wave covers 1 KB of consecutive memory (eight whole cache lines), every LDS store a whole swizzled row
    // per eight lanes.
    const int arow0 = tid >> 3, apiece = tid & 7;
    const bf16x8 *ablock = (const bf16x8 *)A.act + (size_t)blockIdx.x * total * HEAD_ABUF;
    auto aload = [&](int c, bf16x8 (&dst)[2]) {        // (a half chunk re-reads its first piece: clamped, never out of range)
        const int valid = min(8, (n_steps - c * 4) * 2);
        const int u = arow0 * 8 + (apiece < valid ? apiece : 0);
        // (non-temporal: the activations pass through once; 0.526 -> 0.513 ms for both kernels.  With the [agent][K] rows of round 2 the
        // same hint lost)
        dst[0] = __builtin_nontemporal_load(&ablock[(size_t)c * HEAD_ABUF + u]);
        dst[1] = __builtin_nontemporal_load(&ablock[(size_t)c * HEAD_ABUF + 512 + u]);
    };
    auto astore = [&](int off, const bf16x8 (&src)[2]) {       // (off: the buffer's offset in s_act)
        const int slot = arow0 * 8 + (apiece ^ ((arow0 >> 1) & 7));       // (rows r and 64 + r share (r >> 1) & 7)
        s_act[off + slot] = src[0];
        s_act[off + 512 + slot] = src[1];
    };
    auto wload1 = [&](int c, int ks) { return wbase[(size_t)min(c * 4 + ks, n_steps - 1) * 8 * 64]; };
    auto wload = [&](int c, bf16x8 (&dst)[4]) {
#pragma unroll
        for (int ks = 0; ks < 4; ks++) dst[ks] = wload1(c, ks);
    };

    f32x16 acc[4];        // [agent tile]
#pragma unroll
    for (int j = 0; j < 4; j++) acc[j] = f32x16{0};
    // what the phases behind the main loop need from L2 is fetched now (measured: loaded at their point of use -- 32 dependent
    // round trips for the head's weights alone -- those phases were 100 of the kernel's 280 us): the head's weights and the biases of
    // both hidden halves go to LDS (the biases sat in 32 registers through the main loop in round 2; the activation ring has them now)
#pragma unroll
    for (int k = 0; k < 4; k++) s_wh[k * HEAD_THREADS + tid] = A.wh[k * HEAD_THREADS + tid];
    float *s_bias = (float *)(s_wh + 32 * 64);            // [2][8 tiles][2][16]
    s_bias[tid] = tid < 256 ? A.bv[tid] : A.be[tid - 256];

    // The main loop (round 3).  THREE activation buffers: chunk kc + 2 is written while chunk kc is multiplied, so the barrier that ends
    // chunk kc publishes data nobody reads before chunk kc + 1 is over -- and the first operands of chunk kc + 1 (published a barrier
    // earlier) are read BEFORE that barrier, during the last k-step of chunk kc.  The matrix pipe then runs straight through the
    // barriers (with two buffers every chunk began with all eight waves waiting for their first LDS reads: 2100 cycles per chunk for
    // 1024 of MFMAs, by the cycle stamps).  Within a chunk the four operands of k-step ks + 1 are read while the MFMAs of k-step ks
    // run; the chunk's LDS stores and global requests sit behind its second k-step, their latencies behind the other two.  The
    // scheduling barriers pin that order (left alone the compiler sinks every read to its use).
    // The loop over chunks is written WITHOUT a branch inside a group of four: the compiler's wait-count pass merges control-flow
    // paths pessimistically, and with `if (kc >= total) break` after every chunk there was a path from the end of chunk q = 0 straight
    // to the loop latch and back to the top on which only three loads follow the one a k-step waits for -- the top of every group
    // waited for vmcnt(3), (2), (1), (0): the whole ring of 24 loads drained once per four chunks.  Full groups are straight-line code
    // (stores and requests past the end are clamped and harmless); the chunks left over, one of which may be short, follow the loop.
    // (Four chunks ahead for both streams: loads return in order on one counter, so the shallower ring sets the distance of both, and
    // eight slots of each do not fit the register file.  The waves still wait for activations a quarter of the loop -- storing data
    // nobody waits for instead ran 0.053 ms faster of 0.23, profiles/r03_policy.txt.)
    constexpr int RING = HEAD_RING;
    bf16x8 ar[RING][2], wr[RING][4];       // ring slot q = kc % RING: activations of chunk kc + 2 (on their way to LDS), weights of chunk kc
    aload(0, ar[0]);
    aload(min(1, total - 1), ar[1]);
#pragma unroll
    for (int q = 0; q < RING; q++) wload(min(q, total - 1), wr[q]);
    astore(0, ar[0]);
    astore(HEAD_ABUF, ar[1]);
#pragma unroll
    for (int q = 0; q < RING; q++) aload(min(2 + q, total - 1), ar[q]);
    int boff[4];
#pragma unroll
    for (int ks = 0; ks < 4; ks++) boff[ks] = r32 * 8 + ((2 * ks + g) ^ rsw);
    bf16x8 b[2][4];
    auto bread = [&](int off, int ks, bf16x8 (&dst)[4]) {
#pragma unroll
        for (int j = 0; j < 4; j++) dst[j] = s_act[off + 32 * 8 * j + boff[ks]];
    };
    __syncthreads();
    int o_cur = 0, o_nxt = HEAD_ABUF, o_wr = 2 * HEAD_ABUF;      // buffers of chunk kc, kc + 1, kc + 2
    bread(o_cur, 0, b[0]);
    const long long t_loop = STAMPS ? clock64() : 0;
    auto chunk = [&](int kc, int q, bool full) __attribute__((always_inline)) {
        const int steps = full ? 4 : min(4, n_steps - kc * 4);
#pragma unroll
        for (int ks = 0; ks < 4; ks++) {
            bread(ks < 3 ? o_cur : o_nxt, (ks + 1) & 3, b[(ks + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);
            if (full || ks < steps) {
#pragma unroll
                for (int j = 0; j < 4; j++) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wr[q][ks], b[ks & 1][j], acc[j], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            wr[q][ks] = wload1(min(kc + RING, total - 1), ks);                // this ring slot is chunk kc + RING's now
            if (ks == 1) {
                astore(o_wr, ar[q]);                                      // chunk kc + 2, requested four chunks ago
                aload(min(kc + 2 + RING, total - 1), ar[q]);
            }
        }
        __syncthreads();
        const int o = o_cur; o_cur = o_nxt; o_nxt = o_wr; o_wr = o;
    };
    int kc0 = 0;
    for (; (kc0 + RING) * 4 <= n_steps; kc0 += RING) {
#pragma unroll
        for (int q = 0; q < RING; q++) chunk(kc0 + q, q, true);
    }
#pragma unroll
    for (int q = 0; q < RING; q++)
        if (kc0 + q < total) chunk(kc0 + q, q, false);
    const long long t_tail = STAMPS ? clock64() : 0;
    // relu(dense_view) -> hidden slots: output tile w holds chunks 4 w + 2 g, 4 w + 2 g + 1 of every agent
    auto hidden_out = [&](int half) {
        float bias[16];
        const f32x4 *bp = (const f32x4 *)(s_bias + half * 256 + w * 32 + g * 16);
#pragma unroll
        for (int q4 = 0; q4 < 4; q4++) { const f32x4 bq = bp[q4]; bias[4 * q4] = bq[0]; bias[4 * q4 + 1] = bq[1]; bias[4 * q4 + 2] = bq[2]; bias[4 * q4 + 3] = bq[3]; }
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int agent = 32 * j + r32;
            f32x16 v;
#pragma unroll
            for (int r = 0; r < 16; r++) v[r] = acc[j][r] + bias[r];
            s_hid[hid_at(agent, 4 * w + 2 * g)] = relu_bf16x8(v, 0);
            s_hid[hid_at(agent, 4 * w + 2 * g + 1)] = relu_bf16x8(v, 8);
        }
    };
    // what the embedding needs from memory is requested before the hidden layer is written out: the features of this thread's agent
    // (two chunks of 8 at most: FK <= 64) and the wave's embedding weights fly while hidden_out and the first half of the head run
    // (they were requested behind the head's first half and waited for on the spot: 2 of the tail's 14 thousand cycles)
    bf16x8 wemb[4];
#pragma unroll
    for (int s = 0; s < 4; s++) wemb[s] = A.we[((size_t)min(s, A.FK / 16 - 1) * 8 + w) * 64 + l];
    float fv[2][8];
    {
        const float *frow = A.feat + (size_t)min(a0 + srow, A.n - 1) * A.F;
#pragma unroll
        for (int i = 0; i < 2; i++)
#pragma unroll
            for (int e = 0; e < 8; e++) fv[i][e] = frow[min((spiece + 4 * i) * 8 + e, A.F - 1)];       // (clamped: selected below)
    }
    hidden_out(0);
    // ---- the dueling head: [32 outputs] x [128 agents], K = 512 hidden slots in two halves; waves 0..3 take 32 agents each
    f32x16 h = {0};
    const int hagent = 32 * (w & 3) + r32;
    __syncthreads();
    // ---- the feature embedding: K = FK (features as bf16 through the staging buffer, [agent][FK / 8 chunks] <= 8 chunks)
#pragma unroll
    for (int i = 0; i < 2; i++) {
        const int c = spiece + 4 * i;
        bf16x8 v;
#pragma unroll
        for (int e = 0; e < 8; e++) v[e] = (__bf16)(c * 8 + e < A.F ? fv[i][e] : 0.0f);
        if (c < A.FK / 8) s_act[sdst + (c ^ ssw)] = v;
    }
    if (w < 4) {
#pragma unroll 4
        for (int s = 0; s < 16; s++) h = __builtin_amdgcn_mfma_f32_32x32x16_bf16(s_wh[s * 64 + l], s_hid[hid_at(hagent, 2 * s + g)], h, 0, 0, 0);
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 4; j++) acc[j] = f32x16{0};
#pragma unroll
    for (int s = 0; s < 4; s++) {
        if (s < A.FK / 16) {
            const int c = (2 * s + g) ^ rsw;
#pragma unroll
            for (int j = 0; j < 4; j++) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wemb[s], s_act[(32 * j + r32) * 8 + c], acc[j], 0, 0, 0);
        }
    }
    __syncthreads();     // the first half of the head has read relu(dense_view)
    hidden_out(1);
    __syncthreads();
    if (w < 4) {
#pragma unroll 4
        for (int s = 16; s < 32; s++) h = __builtin_amdgcn_mfma_f32_32x32x16_bf16(s_wh[s * 64 + l], s_hid[hid_at(hagent, 2 * (s - 16) + g)], h, 0, 0, 0);
        const int agent = hagent;
        // lane (agent, g) holds outputs ch_of(g, r); its partner lane ^ 32 the other sixteen
        float best = -INFINITY, sum = 0.0f, value = 0.0f;
        int arg = 0x7FFFFFFF;
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int o = ch_of(g, r);
            if (o < A.n_action) { sum += h[r]; if (h[r] > best || (h[r] == best && o < arg)) { best = h[r]; arg = o; } }
            if (o == A.n_action) value = h[r];
        }
        const float obest = __shfl_xor(best, 32);
        const int oarg = __shfl_xor(arg, 32);
        sum += __shfl_xor(sum, 32);
        value += __shfl_xor(value, 32);
        if (obest > best || (obest == best && oarg < arg)) { best = obest; arg = oarg; }
        if (a0 + agent < A.n) {
            if (g == 0) A.actions[a0 + agent] = arg;      // argmax Q = argmax advantage: value and mean are per-agent constants
            if (A.q) {
                const float shift = value + A.value_bias - sum / (float)A.n_action;
#pragma unroll
                for (int r = 0; r < 16; r++) { const int o = ch_of(g, r); if (o < A.n_action) A.q[(size_t)(a0 + agent) * A.n_action + o] = h[r] + shift; }
            }
        }
    }
    if (STAMPS && tid == 0) {
        long long *o = A.stamps + (size_t)blockIdx.x * 8;
        o[0] = t_loop - t_begin; o[1] = t_tail - t_loop; o[2] = clock64() - t_tail; o[3] = t_begin;
    }
}

}  // namespace

extern "C" {

static size_t act_bytes(const PolicyDqnShape *s, int n) {      // whole groups of ACT_GROUP agents, whole K-chunks of 64
    const size_t n_pos = (size_t)(s->view_h - 4) * (s->view_w - 4);
    return (size_t)((n + ACT_GROUP - 1) / ACT_GROUP) * ((n_pos + 1) / 2) * ACT_GROUP * 64 * 2;
}
int policy_dqn_act_bytes(const PolicyDqnShape *s, int n, size_t *bytes) {
    *bytes = act_bytes(s, n) + 2048;       // (+ the dump line of k_dqn_conv)
    return 0;
}

int policy_dqn_supported(const PolicyDqnShape *s) {
    return s->view_c >= 1 && s->view_c <= 7 && s->view_h >= 5 && s->view_w >= 5 && s->view_h * s->view_w * CONV_TA <= CONV_CELLS * CONV_THREADS && s->feat >= 1 && s->feat <= 64 &&
           s->n_action >= 1 && s->n_action <= 31;
}

static int dqn_infer(const PolicyDqnShape *s, const PolicyDqnWeights *w, const void *view_any, bool cells16, const float *feat, int n,
                     void *act_workspace, int *actions, float *q, void *stream) {
    const float *view = (const float *)view_any;
    if (!policy_dqn_supported(s)) return 1;
    if (n <= 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    const int H = s->view_h, W = s->view_w, H1 = H - 2, H2 = H - 4, W2 = W - 4;
    // LDS images of k_dqn_conv: agent pitches == 4 (mod 16) sixteen-byte slots (conflict-free operand reads without a swizzle)
    auto pitch = [](int least) { return least + ((4 - least % 16) % 16 + 16) % 16; };
    const int VP = pitch(H * W + 3), AP = pitch(H1 * W);
    const size_t lds = ((size_t)CONV_TA * VP + (size_t)4 * CONV_TA * AP) * 16 + 32 * 4;
    if (lds > 150 * 1024) return 1;
    constexpr int conv_wpc = 2;      // workgroups of k_dqn_conv per CU (1 and 3 measured slower: profiles/r03_policy.txt)
    const bool f13 = H == 13 && W == 13;
    // the stream's device is made current (launches and function attributes are per device), and the dynamic-LDS allowance is
    // granted once per DEVICE, not once per process
    // (the caller's current device is put back when the call returns: a C-ABI call must not leave a side effect in a multi-GPU process)
    int dev = 0, caller_dev = -1;
    if (hipGetDevice(&caller_dev) != hipSuccess) return 2;
    if (st) { if (hipStreamGetDevice(st, &dev) != hipSuccess || hipSetDevice(dev) != hipSuccess) return 2; }
    else dev = caller_dev;
    struct Restore { int d, cur; ~Restore() { if (d != cur) (void)hipSetDevice(d); } } restore{caller_dev, dev};
    constexpr int MAX_DEV = 64;
    if (dev < 0 || dev >= MAX_DEV) return 2;
    static bool lds_ok_dev[MAX_DEV] = {};
    bool &lds_ok = lds_ok_dev[dev];
    if (!lds_ok) {
        const void *convs[5] = {reinterpret_cast<const void *>(k_dqn_conv<false, false>), reinterpret_cast<const void *>(k_dqn_conv<true, false>),
                                reinterpret_cast<const void *>(k_dqn_conv<false, true>), reinterpret_cast<const void *>(k_dqn_conv<true, true>),
                                reinterpret_cast<const void *>(k_dqn_conv<true, true, true>)};
        for (const void *f : convs)
            if (hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024) != hipSuccess) return 2;
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(k_dqn_head<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)HEAD_LDS) != hipSuccess) return 2;
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(k_dqn_head<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)HEAD_LDS) != hipSuccess) return 2;
        lds_ok = true;
    }
    ConvArgs C{};
    C.view = view; C.act = (__bf16 *)act_workspace; C.w1 = (const bf16x8 *)w->conv1; C.w2 = (const bf16x8 *)w->conv2; C.b2 = w->conv2_bias;
    C.n = n; C.H = H; C.W = W; C.C = s->view_c; C.VP = VP; C.AP = AP; C.n_tiles = (n + CONV_TA - 1) / CONV_TA; C.dump = (bf16x8 *)((char *)act_workspace + act_bytes(s, n));
    static const int grid_cap = magent_amd::tune("policy_grid", 256 * conv_wpc);   // (tests: a few workgroups walk many tiles)
    const int grid = C.n_tiles < grid_cap ? C.n_tiles : grid_cap < 1 ? 1 : grid_cap;     // persistent (2 per CU): weights are fetched once per wave
    // development (MAGENT_TUNE policy_stamps=1, bf16-cell views of the battle shape only): wave 0 of every workgroup reads the cycle counter at
    // its phase boundaries; the launch is waited for and the averages go to stderr.  A separate instantiation: the product kernels
    // carry none of it.
    static const bool stamps_on = magent_amd::tune("policy_stamps", 0) != 0;
    const int head_grid = (n + HEAD_M - 1) / HEAD_M;
    long long *d_stamps = nullptr;
    const bool stamp_conv = stamps_on && cells16 && f13;
    if (stamps_on) {
        if (hipMalloc((void **)&d_stamps, (size_t)(grid + head_grid) * 64) != hipSuccess) return 2;
        (void)hipMemsetAsync(d_stamps, 0, (size_t)(grid + head_grid) * 64, st);
        C.stamps = d_stamps;
    }
#define CONV_LAUNCH(...) hipLaunchKernelGGL((k_dqn_conv<__VA_ARGS__>), dim3(grid), dim3(CONV_THREADS), lds, st, C)
    if (stamp_conv) CONV_LAUNCH(true, true, true);
    else if (f13) { if (cells16) CONV_LAUNCH(true, true); else CONV_LAUNCH(false, true); }
    else { if (cells16) CONV_LAUNCH(true, false); else CONV_LAUNCH(false, false); }
#undef CONV_LAUNCH
    HeadArgs Hd{};
    Hd.act = (const __bf16 *)act_workspace; Hd.feat = feat; Hd.wv = (const bf16x8 *)w->dense_view; Hd.we = (const bf16x8 *)w->dense_emb; Hd.wh = (const bf16x8 *)w->head;
    Hd.bv = w->dense_view_bias; Hd.be = w->dense_emb_bias; Hd.value_bias = w->value_bias;
    Hd.n = n; Hd.K = H2 * W2 * 32; Hd.F = s->feat; Hd.FK = (s->feat + 15) / 16 * 16; Hd.n_action = s->n_action; Hd.actions = actions; Hd.q = q;
    if (stamps_on) {
        Hd.stamps = d_stamps + (size_t)grid * 8;
        hipLaunchKernelGGL(k_dqn_head<true>, dim3(head_grid), dim3(HEAD_THREADS), HEAD_LDS, st, Hd);
        std::vector<long long> h((size_t)(grid + head_grid) * 8);
        if (hipStreamSynchronize(st) != hipSuccess || hipMemcpy(h.data(), d_stamps, h.size() * 8, hipMemcpyDeviceToHost) != hipSuccess) return 3;
        (void)hipFree(d_stamps);
        auto mean = [&](int first, int count, int k) { double a = 0; for (int i = 0; i < count; i++) a += (double)h[(size_t)(first + i) * 8 + k]; return a / count; };
        auto span = [&](int first, int count, int k_begin, int k_total) {      // first start .. last end, cycles
            long long lo = h[(size_t)first * 8 + k_begin], hi = lo;
            for (int i = 0; i < count; i++) { const long long b = h[(size_t)(first + i) * 8 + k_begin]; lo = b < lo ? b : lo; hi = b + k_total > hi ? b + k_total : hi; }
            return (double)(hi - lo);
        };
        if (stamp_conv)
            fprintf(stderr, "[policy stamps] conv: %d workgroups x %.1f tiles; cycles per tile: barriers %.0f conv1 %.0f stage %.0f conv2 %.0f; per workgroup %.0f\n", grid,
                    (double)C.n_tiles / grid, mean(0, grid, 0) * grid / C.n_tiles, mean(0, grid, 1) * grid / C.n_tiles, mean(0, grid, 2) * grid / C.n_tiles,
                    mean(0, grid, 3) * grid / C.n_tiles, mean(0, grid, 4));
        fprintf(stderr, "[policy stamps] head: %d workgroups; cycles: prologue %.0f main loop %.0f tail %.0f\n", head_grid, mean(grid, head_grid, 0),
                mean(grid, head_grid, 1), mean(grid, head_grid, 2));
        (void)span;
    } else
        hipLaunchKernelGGL(k_dqn_head<false>, dim3(head_grid), dim3(HEAD_THREADS), HEAD_LDS, st, Hd);
    return hipGetLastError() == hipSuccess ? 0 : 3;
}

int policy_dqn_infer(const PolicyDqnShape *s, const PolicyDqnWeights *w, const float *view, const float *feat, int n, void *act_workspace,
                     int *actions, float *q, void *stream) {
    return dqn_infer(s, w, view, false, feat, n, act_workspace, actions, q, stream);
}
int policy_dqn_infer_bf16(const PolicyDqnShape *s, const PolicyDqnWeights *w, const void *view_cells, const float *feat, int n,
                          void *act_workspace, int *actions, float *q, void *stream) {
    return dqn_infer(s, w, view_cells, true, feat, n, act_workspace, actions, q, stream);
}

}  // extern "C"
