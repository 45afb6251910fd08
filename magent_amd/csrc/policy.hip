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

#include <vector>

#include "../../include/magent_policy.h"

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

// result register r of lane group g (= lane >> 5) is row (r & 3) + 8 (r >> 2) + 4 g of the 32 x 32 tile
__device__ __forceinline__ int ch_of(int g, int r) { return (r & 3) + 8 * (r >> 2) + 4 * g; }

constexpr int CONV_THREADS = 256, CONV_CELLS = 4, CONV_C2_ITERS = 4;   // CONV_C2_ITERS x 8 tiles of 32 conv2 positions per pass, at most   // CONV_CELLS: window cells of a tile per thread (register prefetch)

struct ConvArgs {
    const float *view;     // [n][H][W][C] f32, or (CELLS16) [n][H][W][8] bf16 cells as env_get_observation_device_bf16 writes them
    __bf16 *act;           // [n][H2 * W2][32 slots]
    const bf16x8 *w1, *w2; // fragment order: [5][64], [18][64]; conv1's bias sits in w1 at (tap 0, channel 7)
    const float *b2;       // [2][16]: conv2's bias of the channel in slot 16 g + r
    int n, H, W, C, TA, AP, n_tiles;
    bf16x8 *dump;          // 2 KB behind the workspace: where lanes without a conv2 position store
    long long *stamps;     // STAMPS instantiation (MAGENT_POLICY_STAMPS): per workgroup, cycles spent in each phase
};

// LDS images (the layouts make every ds_read_b128 of an MFMA operand conflict-free: a 16-lane service group of the instruction
// must hit 16 different 16-byte slots of the 256-byte LDS row -- checked offline by simulating the group lists of
// MI355X_MICROARCH.md, LDS section):
//   s_view [TA * H * W + 2] cells of 8 bf16: channels 0..C-1, zeros, and 1.0 in channel 7 -- conv1's bias is the weight of that
//          constant, so the MFMA adds it.  conv1 is evaluated over FULL rows of W positions (the last two of a row are garbage
//          that lands in the padding columns of s_c1): consecutive lanes then read consecutive cells.
//   s_c1   [TA][AP = H1 * W + pad positions][4 chunks of 8 slots]: row pitch W (== W2 mod 4) and agent pitch AP (== H2 * W2
//          mod 4) make a position's index congruent mod 4 to u = its rank in conv2's own enumeration; the chunk index is xor-ed
//          with (u >> 2) & 3.  The 16 lanes of a group have 16 consecutive-modulo-16 ranks, for every tap: 16 distinct slots.
template <int C2I, bool CELLS16, bool STAMPS = false>   // C2I: passes of conv2 per tile = ceil(tiles of 32 positions / 8), a compile-time count (see the
                                   // stores below).  CELLS16: the views arrive as bf16 cells -- conv1's operands as they are
__global__ void __launch_bounds__(CONV_THREADS) k_dqn_conv(ConvArgs A) {
    extern __shared__ __attribute__((aligned(16))) unsigned char s_raw[];
    const int H1 = A.H - 2, H2 = A.H - 4, W2 = A.W - 4;
    const int cells = A.TA * A.H * A.W, E1 = A.TA * H1 * A.W, P2 = A.TA * H2 * W2;
    bf16x8 *s_view = (bf16x8 *)s_raw;                                  // [cells + 2]
    bf16x8 *s_c1 = s_view + cells + 2;                                 // [TA * AP][4]
    unsigned *s_lut2 = (unsigned *)(s_c1 + (size_t)A.TA * A.AP * 4);   // conv2 position -> its top-left c1 position | 9 taps x 2 swizzle bits << 12
    unsigned short *s_lut1 = (unsigned short *)(s_lut2 + P2);          // conv1 position (full rows) -> c1 position | swizzle << 14
    unsigned short *s_lutc = s_lut1 + E1;                              // conv1 position -> its top-left window cell
    float *s_bias = (float *)(s_lutc + E1);                            // [2][16] (4-byte aligned: 2 * E1 ushorts lie before it)

    const int tid = threadIdx.x, l = tid & 63, w = tid >> 6, g = l >> 5, r32 = l & 31;
    bf16x8 wf1[5], wf2[18];
#pragma unroll
    for (int s = 0; s < 5; s++) wf1[s] = A.w1[s * 64 + l];
#pragma unroll
    for (int s = 0; s < 18; s++) wf2[s] = A.w2[s * 64 + l];
    for (int k = tid; k < 32; k += CONV_THREADS) s_bias[k] = A.b2[k];
    for (int E = tid; E < E1; E += CONV_THREADS) {
        const int a = E / (H1 * A.W), rem = E - a * H1 * A.W, y = rem / A.W, x = rem - y * A.W;
        const int u = (a * H2 * W2 + y * W2 + x) & 15;                 // (rank of c1 position (a, y, x) in conv2's enumeration, mod 16)
        s_lut1[E] = (unsigned short)((a * A.AP + y * A.W + x) | (((u >> 2) & 3) << 14));
        s_lutc[E] = (unsigned short)(a * A.H * A.W + y * A.W + x);
    }
    for (int Q = tid; Q < P2; Q += CONV_THREADS) {
        const int a = Q / (H2 * W2), rem = Q - a * H2 * W2, y = rem / W2, x = rem - y * W2;
        unsigned q = 0;
        for (int tap = 0; tap < 9; tap++) q |= (unsigned)((((Q + (tap / 3) * W2 + tap % 3) & 15) >> 2) & 3) << (2 * tap);
        s_lut2[Q] = (unsigned)(a * A.AP + y * A.W + x) | (q << 12);
    }
    if (tid < 2) s_view[cells + tid] = bf16x8{0};
    // conv1: k-step s covers taps 2 s and 2 s + 1 (lane group g takes tap 2 s + g; tap 9 is padding: zero weights)
    int off1[5];
#pragma unroll
    for (int s = 0; s < 5; s++) { const int tap = 2 * s + g; off1[s] = tap < 9 ? (tap / 3) * A.W + tap % 3 : 0; }

    // A tile's window cells are fetched a whole tile AHEAD, into registers: the loads of tile t + 1 are issued before the
    // convolutions of tile t and waited for after them, so their HBM latency hides behind ~2 us of MFMA work (the loop was bound
    // by it).  A thread owns whole cells: C floats in, one 16-byte LDS store out.
    float nv[CONV_CELLS][7];
    // (the loads are unconditional, from clamped addresses: a select on a loaded value would make the wave wait for it at once)
    // (seven channels -- every reference game with a minimap and two groups -- are fetched as one 16-byte and one 12-byte load per
    // cell: a wave's lanes are 28 bytes apart, so every load instruction walks 14 cache lines whatever its width, and seven
    // 4-byte loads per cell kept the CU's one address pipeline busy for a quarter of the kernel)
    typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));
    typedef float f32x3u __attribute__((ext_vector_type(3), aligned(4)));
    bf16x8 nc[CONV_CELLS];
    auto fetch = [&](int tile) {
        const int live = min(A.TA, A.n - tile * A.TA) * A.H * A.W;
        if (CELLS16) {
            const bf16x8 *src16 = (const bf16x8 *)A.view + (size_t)tile * cells;
#pragma unroll
            for (int k = 0; k < CONV_CELLS; k++) nc[k] = src16[min(k * CONV_THREADS + tid, live - 1)];
            return;
        }
        const float *src = A.view + (size_t)tile * cells * A.C;
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
        const int na_t = min(A.TA, A.n - tile * A.TA);
#pragma unroll
        for (int k = 0; k < CONV_CELLS; k++) {
            const int c = k * CONV_THREADS + tid;
            const bool have = c < na_t * A.H * A.W;
            bf16x8 v;
            if (CELLS16) {
                v = nc[k];
                if (!have) { v = bf16x8{0}; v[7] = (__bf16)1.0f; }
            } else {
#pragma unroll
                for (int e = 0; e < 7; e++) v[e] = (__bf16)((have && e < A.C) ? nv[k][e] : 0.0f);
                v[7] = (__bf16)1.0f;
            }
            if (c < cells) s_view[c] = v;
        }
    };
    // Order of a tile's memory traffic (round 3).  Loads and stores share one counter on this part and complete out of order with respect
    // to each other, so WAITING FOR A LOAD ALSO WAITS FOR EVERY STORE IN FLIGHT (profiles/r03_render_experiments.md).  Round 2 staged the
    // next tile's views at the top of the loop -- right behind the conv2 stores of the tile before, whose whole round trip the wave then sat
    // out (the "stores 0.11 ms" of profiles/r02_policy.txt).  Now the views of tile t + 1 go to LDS in the MIDDLE of tile t: conv1 has
    // finished with s_view, the loads were issued a conv2 + a conv1 ago, and the only stores in flight are the previous tile's, a conv1 old;
    // the stores of tile t then have the whole conv1 of tile t + 1 to land before anybody waits again.
    // (The compiler's wait-count pass merges control-flow paths pessimistically.  Every wait for the weight loads above sat inside some
    // branch, so there was a path on which they were still in flight at the head of the tile loop -- and the loop then waited for
    // vmcnt(4) / vmcnt(0), i.e. for the view prefetch issued a moment earlier and for the previous tile's stores, in the middle of
    // EVERY conv1 and conv2.  An explicit wait here retires the weights on every path; it costs one round trip per workgroup.)
    __builtin_amdgcn_s_waitcnt(0x0F70);      // vmcnt(0): the weights are in their registers, on every path into the loop
    fetch(blockIdx.x); __syncthreads(); stage(blockIdx.x);
    if ((int)(blockIdx.x + gridDim.x) < A.n_tiles) fetch(blockIdx.x + gridDim.x);

    long long t_bar = 0, t_c1 = 0, t_stage = 0, t_c2 = 0, t_a = 0, t_b = 0;      // (STAMPS: wave 0's cycle counter around the phases)
    const long long t_begin = STAMPS ? clock64() : 0;
    for (int tile = blockIdx.x; tile < A.n_tiles; tile += gridDim.x) {
        const int a0 = tile * A.TA, na = min(A.TA, A.n - a0);
        if (STAMPS) t_a = clock64();
        __syncthreads();     // s_view holds this tile (staged in the middle of the tile before); the previous conv2's readers of s_c1 are done
        // ---- conv1: [E1 positions, full rows] x [32 channels], K = 10 taps x 8 channels (bias: the constant channel of tap 0).
        // A wave runs TWO position tiles at a time (t and t + 4): two independent accumulator chains keep the matrix pipe busy
        // while the other chain's operands are on their way from LDS.
        if (STAMPS) { t_b = clock64(); t_bar += t_b - t_a; }
        const int T1 = (E1 + 31) / 32, NW = CONV_THREADS / 64;
        for (int t = w; t < T1; t += 2 * NW) {
            const bool two = t + NW < T1;
            const int Ea = min(t * 32 + r32, E1 - 1), Eb = min((t + NW) * 32 + r32, E1 - 1);
            const int ba = s_lutc[Ea], bb = s_lutc[Eb];
            const unsigned ea = s_lut1[Ea], eb = s_lut1[Eb];
            bf16x8 xa[5], xb[5];
#pragma unroll
            for (int s = 0; s < 5; s++) { xa[s] = s_view[ba + off1[s]]; xb[s] = s_view[bb + off1[s]]; }
            f32x16 acca = {0}, accb = {0};
#pragma unroll
            for (int s = 0; s < 5; s++) {
                acca = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf1[s], xa[s], acca, 0, 0, 0);
                accb = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf1[s], xb[s], accb, 0, 0, 0);
            }
            bf16x8 o0, o1;
#pragma unroll
            for (int r = 0; r < 8; r++) { o0[r] = (__bf16)fmaxf(acca[r], 0.0f); o1[r] = (__bf16)fmaxf(acca[8 + r], 0.0f); }
            s_c1[(ea & 0x3FFF) * 4 + ((2 * g) ^ (ea >> 14))] = o0;       // (lanes past E1 repeat the last position: same values, harmless)
            s_c1[(ea & 0x3FFF) * 4 + ((2 * g + 1) ^ (ea >> 14))] = o1;
            if (two) {
#pragma unroll
                for (int r = 0; r < 8; r++) { o0[r] = (__bf16)fmaxf(accb[r], 0.0f); o1[r] = (__bf16)fmaxf(accb[8 + r], 0.0f); }
                s_c1[(eb & 0x3FFF) * 4 + ((2 * g) ^ (eb >> 14))] = o0;
                s_c1[(eb & 0x3FFF) * 4 + ((2 * g + 1) ^ (eb >> 14))] = o1;
            }
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
        // Two tiles per wave at a time here too; the operands of the next tap are read while the current one runs.
        const unsigned char *c1b = (const unsigned char *)s_c1;
        // (a fixed number of passes, every one issuing its four stores -- lanes without a position write to a dump line behind the
        // workspace: the number of stores outstanding when the next tile's views are waited for is then known at compile time,
        // and nothing sits in a run-time loop that would make the compiler drain the prefetch before entering it.  Measured:
        // neutral -- the conv2 stores cost their 0.1 ms per 131072 agents either way, profiles/r02_policy.txt.)
#pragma unroll
        for (int it = 0; it < C2I; it++) {
            const int t = w + it * 2 * NW;
            const int Qa = t * 32 + r32, Qb = (t + NW) * 32 + r32;
            const unsigned ea = s_lut2[min(Qa, P2 - 1)], eb = s_lut2[min(Qb, P2 - 1)];
            const unsigned basea = (ea & 0xFFFu) << 6, qa = ea >> 12, baseb = (eb & 0xFFFu) << 6, qb = eb >> 12;
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
            unsigned ada = basea + (((qa & 3) ^ g) << 4), adb = baseb + (((qb & 3) ^ g) << 4);
            bf16x8 xa0 = *(const bf16x8 *)(c1b + ada), xa1 = *(const bf16x8 *)(c1b + (ada ^ 32));
            bf16x8 xb0 = *(const bf16x8 *)(c1b + adb), xb1 = *(const bf16x8 *)(c1b + (adb ^ 32));
#pragma unroll
            for (int tap = 0; tap < 9; tap++) {
                bf16x8 nxa0 = xa0, nxa1 = xa1, nxb0 = xb0, nxb1 = xb1;
                if (tap < 8) {
                    const int nt = tap + 1;
                    const unsigned toff = (unsigned)(((nt / 3) * A.W + nt % 3) * 64);
                    ada = basea + toff + ((((qa >> (2 * nt)) & 3) ^ g) << 4);
                    adb = baseb + toff + ((((qb >> (2 * nt)) & 3) ^ g) << 4);
                    nxa0 = *(const bf16x8 *)(c1b + ada); nxa1 = *(const bf16x8 *)(c1b + (ada ^ 32));
                    nxb0 = *(const bf16x8 *)(c1b + adb); nxb1 = *(const bf16x8 *)(c1b + (adb ^ 32));
                }
                acca = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf2[2 * tap], xa0, acca, 0, 0, 0);
                accb = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf2[2 * tap], xb0, accb, 0, 0, 0);
                acca = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf2[2 * tap + 1], xa1, acca, 0, 0, 0);
                accb = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf2[2 * tap + 1], xb1, accb, 0, 0, 0);
                xa0 = nxa0; xa1 = nxa1; xb0 = nxb0; xb1 = nxb1;
            }
            const int lim = na * H2 * W2;
            bf16x8 o0, o1;
            {
#pragma unroll
                for (int r = 0; r < 8; r++) { o0[r] = (__bf16)fmaxf(acca[r], 0.0f); o1[r] = (__bf16)fmaxf(acca[8 + r], 0.0f); }
                bf16x8 *dst = Qa < lim ? (bf16x8 *)(A.act + ((size_t)a0 * H2 * W2 + Qa) * 32 + 16 * g) : A.dump + 2 * l;
                dst[0] = o0; dst[1] = o1;
            }
            {
#pragma unroll
                for (int r = 0; r < 8; r++) { o0[r] = (__bf16)fmaxf(accb[r], 0.0f); o1[r] = (__bf16)fmaxf(accb[8 + r], 0.0f); }
                bf16x8 *dst = Qb < lim ? (bf16x8 *)(A.act + ((size_t)a0 * H2 * W2 + Qb) * 32 + 16 * g) : A.dump + 2 * l;
                dst[0] = o0; dst[1] = o1;
            }
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
constexpr size_t HEAD_LDS = (2 * HEAD_M * (HEAD_KC / 8) + HEAD_M * 32 + 32 * 64) * 16;   // 2 x 16 KB activations + 64 KB hidden + 32 KB head weights = 128 KB

struct HeadArgs {
    const __bf16 *act;        // [n][K] (K = H2 * W2 * 32, slot order)
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
    bf16x8 (*s_act)[HEAD_M * (HEAD_KC / 8)] = (bf16x8 (*)[HEAD_M * (HEAD_KC / 8)])s_raw;      // [2][agent][8 chunks], swizzled
    bf16x8 *s_hid = (bf16x8 *)s_raw + 2 * HEAD_M * (HEAD_KC / 8);                              // [agent][32 chunks], swizzled
    bf16x8 *s_wh = s_hid + HEAD_M * 32;                                                        // the head's weights, fragment order [32][64]
    const int tid = threadIdx.x, l = tid & 63, w = tid >> 6, g = l >> 5, r32 = l & 31;
    const int a0 = blockIdx.x * HEAD_M;

    const int srow = tid >> 2, spiece = tid & 3;       // staging: thread t -> row t / 4, 32-byte piece t % 4 of a 128-byte chunk row
    const bf16x8 *arow = (const bf16x8 *)(A.act + (size_t)min(a0 + srow, A.n - 1) * A.K);
    const int sdst = srow * 8, ssw = (srow >> 1) & 7, rsw = (r32 >> 1) & 7;
    const bf16x8 *wbase = A.wv + (size_t)w * 64 + l;   // fragment (k-step s, tile w) = wbase[s * 8 * 64]
    const int n_steps = A.K / 16;                      // k-steps in all; the last chunk may be half (K is a multiple of 32)
    const int total = (n_steps + 3) / 4;
    auto aload = [&](int c, bf16x8 (&dst)[2]) {        // (a half chunk re-reads its first piece: clamped, never out of range)
        const int valid = min(8, (n_steps - c * 4) * 2);
#pragma unroll
        for (int i = 0; i < 2; i++) { const int ch = spiece * 2 + i; dst[i] = arow[(size_t)c * 8 + (ch < valid ? ch : 0)]; }
    };
    auto astore = [&](int buf, const bf16x8 (&src)[2]) {
        s_act[buf][sdst + ((spiece * 2) ^ ssw)] = src[0];
        s_act[buf][sdst + ((spiece * 2 + 1) ^ ssw)] = src[1];
    };
    auto wload = [&](int c, bf16x8 (&dst)[4]) {
#pragma unroll
        for (int ks = 0; ks < 4; ks++) dst[ks] = wbase[(size_t)min(c * 4 + ks, n_steps - 1) * 8 * 64];
    };

    f32x16 acc[4];        // [agent tile]
#pragma unroll
    for (int j = 0; j < 4; j++) acc[j] = f32x16{0};
    // what the phases behind the main loop need from L2 is fetched now (measured: loaded at their point of use -- 32 dependent
    // round trips for the head's weights alone -- those phases were 100 of the kernel's 280 us): the head's weights go to LDS,
    // this lane's biases to registers
#pragma unroll
    for (int k = 0; k < 4; k++) s_wh[k * HEAD_THREADS + tid] = A.wh[k * HEAD_THREADS + tid];
    float bias_v[16], bias_e[16];
#pragma unroll
    for (int r = 0; r < 16; r++) { bias_v[r] = A.bv[w * 32 + g * 16 + r]; bias_e[r] = A.be[w * 32 + g * 16 + r]; }

    bf16x8 ar[4][2], wr[4][4];
    aload(0, ar[0]);
#pragma unroll
    for (int q = 0; q < 4; q++) wload(min(q, total - 1), wr[q]);
    astore(0, ar[0]);
#pragma unroll
    for (int q = 0; q < 4; q++) aload(min(1 + q, total - 1), ar[q]);
    __syncthreads();
    const long long t_loop = STAMPS ? clock64() : 0;
    // One 64-wide K-chunk: its MFMAs, then the ring slot q = kc & 3 moves on (chunk kc + 1 to LDS, chunk kc + 5's activations and
    // chunk kc + 4's weights requested).  The loop over chunks is written WITHOUT a branch inside a group of four (round 3).  The
    // compiler's wait-count pass merges control-flow paths pessimistically: with `if (kc >= total) break` after every chunk there is
    // a path from the end of chunk q = 0 straight to the loop latch and back to the top, on which only three loads follow the one a
    // k-step waits for -- so the top of every group waited for vmcnt(3), (2), (1), (0): the whole ring of 24 loads drained once per
    // four chunks, and a k-step in its own basic block (`if (ks < steps)`) read its LDS operands right before the MFMAs that use them.
    // Full groups are now straight-line code (the stores and requests of a chunk past the end are clamped and harmless: the buffer
    // they land in was last read before the previous barrier); the chunks left over, one of which may be short, follow the loop.
    auto chunk = [&](int kc, int q, bool full) __attribute__((always_inline)) {
        const int buf = kc & 1;
        const int steps = full ? 4 : min(4, n_steps - kc * 4);
        // (the four activation operands of k-step ks + 1 are read while the MFMAs of k-step ks run; the scheduling barriers keep the
        // compiler from sinking the reads to where their values are used, which is what it does to save registers)
        bf16x8 b[2][4];
        auto bread = [&](int ks, bf16x8 (&dst)[4]) {
            const int c = (2 * ks + g) ^ rsw;
#pragma unroll
            for (int j = 0; j < 4; j++) dst[j] = s_act[buf][(32 * j + r32) * 8 + c];
        };
        bread(0, b[0]);
#pragma unroll
        for (int ks = 0; ks < 4; ks++) {
            if (ks < 3 && (full || ks + 1 < steps)) bread(ks + 1, b[(ks + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);
            if (full || ks < steps) {
#pragma unroll
                for (int j = 0; j < 4; j++) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wr[q][ks], b[ks & 1][j], acc[j], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        astore(buf ^ 1, ar[q]);                                       // chunk kc + 1, requested four chunks ago
        aload(min(kc + 5, total - 1), ar[q]);
        wload(min(kc + 4, total - 1), wr[q]);                         // this ring slot is chunk kc + 4's now
        __syncthreads();
    };
    int kc0 = 0;
    for (; (kc0 + 4) * 4 <= n_steps; kc0 += 4) {
#pragma unroll
        for (int q = 0; q < 4; q++) chunk(kc0 + q, q, true);
    }
#pragma unroll
    for (int q = 0; q < 4; q++)
        if (kc0 + q < total) chunk(kc0 + q, q, false);
    const long long t_tail = STAMPS ? clock64() : 0;
    // relu(dense_view) -> hidden slots: output tile w holds chunks 4 w + 2 g, 4 w + 2 g + 1 of every agent
    auto hidden_out = [&](const float (&bias)[16]) {
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int agent = 32 * j + r32;
            bf16x8 o0, o1;
#pragma unroll
            for (int r = 0; r < 8; r++) {
                o0[r] = (__bf16)fmaxf(acc[j][r] + bias[r], 0.0f);
                o1[r] = (__bf16)fmaxf(acc[j][8 + r] + bias[8 + r], 0.0f);
            }
            s_hid[hid_at(agent, 4 * w + 2 * g)] = o0;
            s_hid[hid_at(agent, 4 * w + 2 * g + 1)] = o1;
        }
    };
    hidden_out(bias_v);
    // ---- the dueling head: [32 outputs] x [128 agents], K = 512 hidden slots in two halves; waves 0..3 take 32 agents each
    f32x16 h = {0};
    const int hagent = 32 * (w & 3) + r32;
    __syncthreads();
    if (w < 4) {
#pragma unroll 4
        for (int s = 0; s < 16; s++) h = __builtin_amdgcn_mfma_f32_32x32x16_bf16(s_wh[s * 64 + l], s_hid[hid_at(hagent, 2 * s + g)], h, 0, 0, 0);
    }
    // ---- the feature embedding: K = FK (features as bf16 through the staging buffer, [agent][FK / 8 chunks] <= 8 chunks)
    bf16x8 wemb[4];
#pragma unroll
    for (int s = 0; s < 4; s++) wemb[s] = A.we[((size_t)min(s, A.FK / 16 - 1) * 8 + w) * 64 + l];
    {
        const int agent = min(a0 + srow, A.n - 1);
        for (int c = spiece; c < A.FK / 8; c += 4) {
            bf16x8 v;
#pragma unroll
            for (int e = 0; e < 8; e++) { const int k = c * 8 + e; v[e] = (__bf16)(k < A.F ? A.feat[(size_t)agent * A.F + k] : 0.0f); }
            s_act[0][sdst + (c ^ ssw)] = v;
        }
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 4; j++) acc[j] = f32x16{0};
#pragma unroll
    for (int s = 0; s < 4; s++) {
        if (s < A.FK / 16) {
            const int c = (2 * s + g) ^ rsw;
#pragma unroll
            for (int j = 0; j < 4; j++) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wemb[s], s_act[0][(32 * j + r32) * 8 + c], acc[j], 0, 0, 0);
        }
    }
    __syncthreads();     // the first half of the head has read relu(dense_view)
    hidden_out(bias_e);
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

static size_t act_bytes(const PolicyDqnShape *s, int n) { return (size_t)n * (s->view_h - 4) * (s->view_w - 4) * 32 * 2; }
int policy_dqn_act_bytes(const PolicyDqnShape *s, int n, size_t *bytes) {
    *bytes = act_bytes(s, n) + 2048;       // (+ the dump line of k_dqn_conv)
    return 0;
}

int policy_dqn_supported(const PolicyDqnShape *s) {
    return s->view_c >= 1 && s->view_c <= 7 && s->view_h >= 5 && s->view_w >= 5 && s->view_h * s->view_w <= 1024 && s->feat >= 1 && s->feat <= 64 &&
           s->n_action >= 1 && s->n_action <= 31;
}

static int dqn_infer(const PolicyDqnShape *s, const PolicyDqnWeights *w, const void *view_any, bool cells16, const float *feat, int n,
                     void *act_workspace, int *actions, float *q, void *stream) {
    const float *view = (const float *)view_any;
    if (!policy_dqn_supported(s)) return 1;
    if (n <= 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    const int H = s->view_h, W = s->view_w, H1 = H - 2, H2 = H - 4, W2 = W - 4;
    // agents per workgroup pass: as many as leave two workgroups per CU their LDS (views 16 B / cell, conv1 64 B / position)
    const int AP = H1 * W + ((H2 * W2 - H1 * W) % 4 + 4) % 4;      // agent pitch of conv1's LDS image: == H2 * W2 (mod 4)
    static const int ta_cap = getenv("MAGENT_POLICY_TA") ? atoi(getenv("MAGENT_POLICY_TA")) : 8;       // (tuning: fewer agents per pass = more workgroups per CU)
    static const int conv_wpc = getenv("MAGENT_POLICY_WPC") ? atoi(getenv("MAGENT_POLICY_WPC")) : 2;
    int TA = ta_cap < 1 ? 1 : ta_cap > 8 ? 8 : ta_cap;
    size_t lds = 0;
    for (; TA >= 1; TA--) {
        const size_t cells = (size_t)TA * H * W, E1 = (size_t)TA * H1 * W, P2 = (size_t)TA * H2 * W2;
        lds = (cells + 2) * 16 + (size_t)TA * AP * 64 + P2 * 4 + E1 * 4 + 32 * 4;
        if (lds <= 78 * 1024 && (size_t)TA * AP < 4096 && cells <= (size_t)CONV_CELLS * CONV_THREADS && P2 <= (size_t)CONV_C2_ITERS * 256) break;
    }
    if (TA < 1) return 1;
    const int T2 = (TA * H2 * W2 + 31) / 32, c2i = (T2 + 7) / 8;
    // the stream's device is made current (launches and function attributes are per device), and the dynamic-LDS allowance is
    // granted once per DEVICE, not once per process
    int dev = 0;
    if (st) { if (hipStreamGetDevice(st, &dev) != hipSuccess || hipSetDevice(dev) != hipSuccess) return 2; }
    else if (hipGetDevice(&dev) != hipSuccess) return 2;
    static bool lds_ok_dev[64] = {};
    bool &lds_ok = lds_ok_dev[dev & 63];
    if (!lds_ok) {
        const void *convs[8] = {reinterpret_cast<const void *>(k_dqn_conv<1, false>), reinterpret_cast<const void *>(k_dqn_conv<2, false>),
                                reinterpret_cast<const void *>(k_dqn_conv<3, false>), reinterpret_cast<const void *>(k_dqn_conv<4, false>),
                                reinterpret_cast<const void *>(k_dqn_conv<1, true>), reinterpret_cast<const void *>(k_dqn_conv<2, true>),
                                reinterpret_cast<const void *>(k_dqn_conv<3, true>), reinterpret_cast<const void *>(k_dqn_conv<4, true>)};
        for (const void *f : convs)
            if (hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024) != hipSuccess) return 2;
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(k_dqn_conv<2, true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024) != hipSuccess) return 2;
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(k_dqn_head<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)HEAD_LDS) != hipSuccess) return 2;
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(k_dqn_head<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)HEAD_LDS) != hipSuccess) return 2;
        lds_ok = true;
    }
    ConvArgs C{};
    C.view = view; C.act = (__bf16 *)act_workspace; C.w1 = (const bf16x8 *)w->conv1; C.w2 = (const bf16x8 *)w->conv2; C.b2 = w->conv2_bias;
    C.n = n; C.H = H; C.W = W; C.C = s->view_c; C.TA = TA; C.AP = AP; C.n_tiles = (n + TA - 1) / TA; C.dump = (bf16x8 *)((char *)act_workspace + act_bytes(s, n));
    static const int grid_cap = getenv("MAGENT_POLICY_GRID") ? atoi(getenv("MAGENT_POLICY_GRID")) : 256 * conv_wpc;   // (tests: a few workgroups walk many tiles)
    const int grid = C.n_tiles < grid_cap ? C.n_tiles : grid_cap < 1 ? 1 : grid_cap;     // persistent (2 per CU): weights are fetched once per wave
    // development (MAGENT_POLICY_STAMPS=1, bf16-cell views of the battle shape only): wave 0 of every workgroup reads the cycle counter at
    // its phase boundaries; the launch is waited for and the averages go to stderr.  A separate instantiation: the product kernels
    // carry none of it.
    static const bool stamps_on = getenv("MAGENT_POLICY_STAMPS") && atoi(getenv("MAGENT_POLICY_STAMPS")) != 0;
    const int head_grid = (n + HEAD_M - 1) / HEAD_M;
    long long *d_stamps = nullptr;
    const bool stamp_conv = stamps_on && cells16 && c2i == 2;
    if (stamps_on) {
        if (hipMalloc((void **)&d_stamps, (size_t)(grid + head_grid) * 64) != hipSuccess) return 2;
        (void)hipMemsetAsync(d_stamps, 0, (size_t)(grid + head_grid) * 64, st);
        C.stamps = d_stamps;
    }
#define CONV_LAUNCH(I, B) hipLaunchKernelGGL((k_dqn_conv<I, B>), dim3(grid), dim3(CONV_THREADS), lds, st, C)
    if (stamp_conv) hipLaunchKernelGGL((k_dqn_conv<2, true, true>), dim3(grid), dim3(CONV_THREADS), lds, st, C);
    else switch (c2i * 2 + (cells16 ? 1 : 0)) {
        case 2: CONV_LAUNCH(1, false); break;
        case 3: CONV_LAUNCH(1, true); break;
        case 4: CONV_LAUNCH(2, false); break;
        case 5: CONV_LAUNCH(2, true); break;
        case 6: CONV_LAUNCH(3, false); break;
        case 7: CONV_LAUNCH(3, true); break;
        case 8: CONV_LAUNCH(4, false); break;
        default: CONV_LAUNCH(4, true); break;
    }
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
