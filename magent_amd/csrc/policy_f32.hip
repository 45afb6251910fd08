// policy_f32.hip -- the reference's deep Q network (python/magent/builtin/tf_model/dqn.py:151-189), inference only, in the REFERENCE'S OWN
// ARITHMETIC: float32 inputs, weights, activations and accumulation, on the f32 matrix instruction of gfx950.
//
// BASELINE config 5 puts this network between get_observation and set_action.  With the PyTorch / MIOpen float32 network a 2 x 500k-agent
// step is 79 ms of inference around 2.3 ms of engine (profiles/r05_summary.md: ~43 TFLOP/s, 27 % of the f32 matrix peak); policy.hip's
// kernels are bf16 -- narrower than the reference's arithmetic.  Here every product is an exact f32 multiply-add:
// v_mfma_f32_32x32x2_f32 is, bit for bit, a k-ordered chain of fmaf (cdna_hip_programming.md "FP32-input MFMA"), 64 FLOP / clk / SIMD =
// 157 TFLOP/s on the chip -- a sixteenth of the bf16 rate, so unlike policy.hip's kernels these are bound by the matrix pipe itself and
// everything else (LDS reads, epilogues, the staging of views and activations) hides behind 64-cycle instructions.
//
//   network:  view [n][H][W][C] f32 -> conv3x3(32, valid) relu -> conv3x3(32, valid) relu -> flatten (NHWC) -> dense 256 relu
//             feature [n][F] f32 -> dense 256 relu;  concat 512 -> advantage (n_action, no bias) and value (1);
//             Q = value + advantage - mean(advantage)
//
// Operands.  A = weights (lane l: output l & 31, k = l >> 5), B = activations (lane l: position or agent l & 31, k = l >> 5); the result
// register r of lane (column, g = l >> 5) is output (r & 3) + 8 (r >> 2) + 4 g: a lane owns outputs 8 q + 4 g + 0..3 for q = 0..3 -- four
// contiguous quads, stored as four 16-byte vectors in NATURAL channel order (no slot permutation as in policy.hip).
// One 16-byte LDS read feeds FOUR k-steps: the reduction index of every layer is cut into groups of 8 values; lane group g reads values
// 4 g .. 4 g + 3 of the group as a float4 and k-step j of the group multiplies element j of the weight float4 with element j of the
// activation float4 -- between them the two lane groups cover the 8 values.  Weights are packed to match ("f32 fragment order"):
//   [K / 8 groups][N / 32 tiles][64 lanes][4]: lane l of (group m, tile T) holds W[32 T + (l & 31)][8 m + 4 (l >> 5) + 0..3]
// (magent_amd/builtin/torch_model/hip_policy.py: fragment_order_f32).  The reduction index of each layer:
//   conv1      : tap * 8 + channel (channels padded to 8; channel 7 is a constant 1.0 whose weight at tap 0 is the bias)     9 groups
//   conv2      : tap * 32 + channel                                                                                           36 groups
//   dense_view : position (y * (W - 4) + x) * 32 + channel                                                                    K / 8 groups
//   dense_emb  : feature index, padded to a multiple of 8
//   head       : hidden unit (dense_view's 256, then dense_emb's 256); outputs 0..n_action-1 advantage, n_action value         64 groups
//
// k_dqn_conv_f32 : conv1 + conv2 fused, a persistent workgroup of 8 waves per CU walking tiles of 4 agents.  The views go to LDS as two
//   planes of float4 (channels 0-3 | 4-7 of every window cell), conv1's output stays in LDS as eight planes (channels 4 p .. 4 p + 3 of
//   every position), both layers' packed weights sit in LDS for the life of the workgroup (46 KB), conv2's output goes to HBM in the
//   order the head's workgroups read it.  Lanes of a 32-wide tile are 8 consecutive positions x 4 agents (position-major): with agent
//   pitches == 4 (mod 16) sixteen-byte units every 16-lane group of a ds_read_b128 covers all 16 LDS columns, a tap is a constant offset.
// k_dqn_head_f32 : dense K -> 256 as a GEMM over 128 agents per workgroup of 8 waves (wave w owns outputs 32 w .. 32 w + 31 for all four
//   agent tiles: per group of 8 K-values ONE weight float4 from L2 and four activation float4 from LDS feed 16 MFMAs), activations double
//   buffered through LDS a 64-value chunk at a time, the feature embedding, the dueling head and the argmax, fused.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/magent_policy.h"
#include "tune.h"

namespace {

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

__device__ __forceinline__ f32x16 mfma4(const f32x4 &w, const f32x4 &x, f32x16 acc) {     // the four k-steps of one group of 8 K-values
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w[0], x[0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w[1], x[1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w[2], x[2], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w[3], x[3], acc, 0, 0, 0);
    return acc;
}
__device__ __forceinline__ f32x4 relu4(const f32x16 &acc, int q) {
    return f32x4{fmaxf(acc[4 * q], 0.0f), fmaxf(acc[4 * q + 1], 0.0f), fmaxf(acc[4 * q + 2], 0.0f), fmaxf(acc[4 * q + 3], 0.0f)};
}

// conv2's output, dense_view's input: [group of 128 agents][K-chunk of 64 values = two positions][agent][64 values] (as policy.hip's
// act_at, in floats): one k_dqn_head_f32 workgroup reads one contiguous 32 KB block per K-chunk
constexpr int ACT_GROUP = 128;
__device__ __forceinline__ size_t act_at(int agent, int pos, int n_pos) {
    const int chunks = (n_pos + 1) >> 1;
    return ((size_t)(agent / ACT_GROUP) * chunks + (pos >> 1)) * (ACT_GROUP * 64) + (size_t)(agent % ACT_GROUP) * 64 + (pos & 1) * 32;
}

constexpr int CONV_THREADS = 512, CONV_CELLS = 2;      // window cells of a tile per thread (register prefetch); a tile is TA = 4 agents (2 for views whose LDS images would not fit)
struct ConvArgs {
    const float *view;      // [n][H][W][C]
    float *act;             // act_at order
    const f32x4 *w1, *w2;   // f32 fragment order: [9][64], [36][64]
    const float *b2;        // [32] conv2's bias, natural channel order
    int n, H, W, C, VP, AP, n_tiles;
    f32x4 *dump;            // 4 KB behind the workspace: where lanes without a conv2 position store
};

template <int TA>
__global__ void __launch_bounds__(CONV_THREADS) k_dqn_conv_f32(ConvArgs A) {
    extern __shared__ __attribute__((aligned(16))) unsigned char s_raw[];
    constexpr int NW = CONV_THREADS / 64, TS = TA == 4 ? 2 : 1;      // (TS: log2 TA)
    const int H = A.H, W = A.W, VP = A.VP, AP = A.AP;
    const int H1 = H - 2, H2 = H - 4, W2 = W - 4, HW = H * W, NP2 = H2 * W2;
    const int W1 = W - 2, E1 = TA * H1 * W1, P2 = TA * NP2, PLV = TA * VP, PL1 = TA * AP;
    f32x4 *s_view = (f32x4 *)s_raw;                 // [2 planes][TA][VP]
    f32x4 *s_c1 = s_view + 2 * PLV;                 // [8 planes][TA][AP]
    f32x4 *s_w1 = s_c1 + 8 * PL1;                   // [9][64]
    f32x4 *s_w2 = s_w1 + 9 * 64;                    // [36][64]
    float *s_bias = (float *)(s_w2 + 36 * 64);      // [32]

    const int tid = threadIdx.x, l = tid & 63, w = tid >> 6, g = l >> 5, r32 = l & 31;
    for (int k = tid; k < 9 * 64; k += CONV_THREADS) s_w1[k] = A.w1[k];
    for (int k = tid; k < 36 * 64; k += CONV_THREADS) s_w2[k] = A.w2[k];
    if (tid < 32) s_bias[tid] = A.b2[tid];
    // plane 0: channels 0..3; plane 1: channels 4..6 and the constant 1.0 of conv1's bias (the cells behind every agent's H W stay zero)
    for (int c = tid; c < 2 * PLV; c += CONV_THREADS) s_view[c] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};

    // a tile's window cells are fetched a whole tile AHEAD into registers (policy.hip: k_dqn_conv)
    float nv[CONV_CELLS][7];
    int sdst[CONV_CELLS];
#pragma unroll
    for (int k = 0; k < CONV_CELLS; k++) { const int c = k * CONV_THREADS + tid, a = c / HW; sdst[k] = a < TA ? a * VP + (c - a * HW) : -1; }
    typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));
    typedef float f32x3u __attribute__((ext_vector_type(3), aligned(4)));
    auto fetch = [&](int tile) {
        const int live = min(TA, A.n - tile * TA) * HW;
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
    auto stage = [&](int tile) {
        const int live = min(TA, A.n - tile * TA) * HW;
#pragma unroll
        for (int k = 0; k < CONV_CELLS; k++) {
            const bool have = k * CONV_THREADS + tid < live;
            float v[7];
#pragma unroll
            for (int e = 0; e < 7; e++) v[e] = (have && e < A.C) ? nv[k][e] : 0.0f;
            if (sdst[k] >= 0) {
                s_view[sdst[k]] = f32x4{v[0], v[1], v[2], v[3]};
                s_view[PLV + sdst[k]] = f32x4{v[4], v[5], v[6], 1.0f};
            }
        }
    };
    __syncthreads();                         // (weights and the zero fill are in place before the first cells land)
    fetch(blockIdx.x); stage(blockIdx.x);
    if ((int)(blockIdx.x + gridDim.x) < A.n_tiles) fetch(blockIdx.x + gridDim.x);

    const int T1 = (E1 + 31) / 32, T2 = (P2 + 31) / 32;
    for (int tile = blockIdx.x; tile < A.n_tiles; tile += gridDim.x) {
        const int a0 = tile * TA;
        __syncthreads();     // s_view holds this tile; the previous conv2's readers of s_c1 are done
        // ---- conv1: [E1 = TA x H1 x W1 valid positions] x [32 channels], K = 9 taps x 8 channels.  (Only the valid columns: 4 x 121 positions
        // are 16 tiles -- four per SIMD; over full rows of 13, as policy.hip has them, they were 18: five on two of the four SIMDs.)
        for (int t = w; t < T1; t += NW) {
            const int E = min(t * 32 + r32, E1 - 1);
            const int p1 = E >> TS, ag = E & (TA - 1);           // valid position, agent
            const int y1 = p1 / W1, pos = y1 * W + (p1 - y1 * W1);      // its top-left window cell = its place in conv1's image (rows of W)
            const f32x4 *vb = s_view + g * PLV + ag * VP + pos;
            f32x16 acc = {0};
            f32x4 x[2];
            x[0] = vb[0];
#pragma unroll
            for (int tap = 0; tap < 9; tap++) {
                if (tap < 8) x[(tap + 1) & 1] = vb[((tap + 1) / 3) * W + (tap + 1) % 3];
                acc = mfma4(s_w1[tap * 64 + l], x[tap & 1], acc);
            }
            f32x4 *d = s_c1 + g * PL1 + ag * AP + pos;           // (lanes past E1 repeat the last position: same values, harmless)
#pragma unroll
            for (int q = 0; q < 4; q++) d[2 * q * PL1] = relu4(acc, q);      // channels 8 q + 4 g + 0..3: plane 2 q + g
        }
        __syncthreads();
        // ---- conv1 is done with s_view: the next tile's views move in, the one after is requested
        if (tile + (int)gridDim.x < A.n_tiles) {
            stage(tile + gridDim.x);
            if (tile + 2 * (int)gridDim.x < A.n_tiles) fetch(tile + 2 * gridDim.x);
        }
        // ---- conv2: [P2 positions] x [32 channels], K = 9 taps x 32 channels; starts from the bias, result straight to HBM
        for (int t = w; t < T2; t += NW) {
            const int Q = t * 32 + r32, Qc = min(Q, P2 - 1);
            const int q = Qc >> TS, ag = Qc & (TA - 1);
            const int y = q / W2;
            const f32x4 *cb = s_c1 + g * PL1 + ag * AP + y * W + (q - y * W2);       // plane g (+ 2 m) of the top-left c1 position
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; r++) acc[r] = s_bias[(r & 3) + 8 * (r >> 2) + 4 * g];
            f32x4 x[2][4];
            auto xread = [&](int tap, f32x4 (&dst)[4]) {
                const int off = (tap / 3) * W + tap % 3;
#pragma unroll
                for (int m = 0; m < 4; m++) dst[m] = cb[off + 2 * m * PL1];
            };
            xread(0, x[0]);
#pragma unroll
            for (int tap = 0; tap < 9; tap++) {
                if (tap < 8) xread(tap + 1, x[(tap + 1) & 1]);
#pragma unroll
                for (int m = 0; m < 4; m++) acc = mfma4(s_w2[(tap * 4 + m) * 64 + l], x[tap & 1][m], acc);
            }
            f32x4 *dst = (Q < P2 && a0 + ag < A.n) ? (f32x4 *)(A.act + act_at(a0 + ag, q, NP2)) + g : A.dump + 4 * l;
            const int step = (Q < P2 && a0 + ag < A.n) ? 2 : 1;
#pragma unroll
            for (int qq = 0; qq < 4; qq++) dst[qq * step] = relu4(acc, qq);          // channels 8 qq + 4 g + 0..3
        }
    }
}

// ---------------------------------------------------------------------------------------------------- dense + head
constexpr int HEAD_THREADS = 512, HEAD_M = 128, HEAD_KC = 64;      // 128 agents per workgroup of 8 waves; K staged 64 values at a time
constexpr int HEAD_ABUF = HEAD_M * (HEAD_KC / 4);                   // float4 units of one activation buffer: 128 agents x 16 = 32 KB
constexpr int HEAD_FMAX = 56;                                       // most features (padded to 8) the embedding's LDS image holds
constexpr size_t HEAD_LDS = ((size_t)HEAD_M * 64 + (size_t)HEAD_M * (HEAD_FMAX / 4)) * 16;      // hidden half 128 KB (the loop's two 32 KB buffers lie inside) + features 28 KB

struct HeadArgs {
    const float *act;         // act_at order, K = H2 * W2 * 32
    const float *feat;        // [n][F]
    const f32x4 *wv;          // dense_view, f32 fragment order [K / 8][8 tiles][64]
    const f32x4 *we;          // dense_emb,  [FK / 8][8 tiles][64]      (FK = F rounded up to 8)
    const f32x4 *wh;          // head, [64][64]: K = 512 hidden units, outputs 0..n_action-1 advantage, n_action value
    const float *bv, *be;     // [256] biases, natural order
    float value_bias;
    int n, K, F, FK, n_action;
    int *actions;             // [n] argmax_a Q
    float *q;                 // [n][n_action] or null
};

// LDS images: rows of 16 float4 (activation chunk) / 64 float4 (hidden half), the unit index xor-ed with the row's low bits so that the 16
// lanes of a ds_read_b128 service group (16 consecutive agents, one unit) cover all 16 columns
__device__ __forceinline__ int act_slot(int row, int unit) { return row * 16 + (unit ^ (row & 15)); }
__device__ __forceinline__ int hid_slot(int row, int unit) { return row * 64 + (unit ^ (row & 15)); }

__global__ void __launch_bounds__(HEAD_THREADS) k_dqn_head_f32(HeadArgs A) {
    extern __shared__ __attribute__((aligned(16))) unsigned char s_raw[];
    f32x4 *s_act = (f32x4 *)s_raw;                         // [2][128 agents][16 units], swizzled -- and, behind the main loop,
    f32x4 *s_hid = (f32x4 *)s_raw;                         // [128 agents][64 units]: one half of the hidden layer
    f32x4 *s_feat = s_hid + HEAD_M * 64;                   // [128 agents][FK / 4 units]
    const int tid = threadIdx.x, l = tid & 63, w = tid >> 6, g = l >> 5, r32 = l & 31;
    const int a0 = blockIdx.x * HEAD_M;
    const int n_groups = A.K / 8;                          // groups of 8 K-values; 8 per chunk, the last chunk may be half (K is a multiple of 32)
    const int total = (n_groups + 7) / 8;
    const f32x4 *ablock = (const f32x4 *)A.act + (size_t)blockIdx.x * total * HEAD_ABUF;
    const f32x4 *wbase = A.wv + (size_t)w * 64 + l;        // fragment (group m, tile w) = wbase[m * 8 * 64]

    // staging: a chunk's block is 2048 float4 units (unit u: agent row u >> 4, piece u & 15); thread t moves units t + 512 i
    f32x4 ar[4];
    auto aload = [&](int c) {
        const int valid = min(16, (n_groups - c * 8) * 2);         // units of a row that exist in this chunk (a half chunk: 8)
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int u = tid + 512 * i, piece = u & 15;
            ar[i] = __builtin_nontemporal_load(&ablock[(size_t)c * HEAD_ABUF + (piece < valid ? u : (u & ~15))]);
        }
    };
    auto astore = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 4; i++) { const int u = tid + 512 * i; s_act[buf * HEAD_ABUF + act_slot(u >> 4, u & 15)] = ar[i]; }
    };
    f32x4 wr[2][8];          // the wave's weight fragments: this chunk's and the next one's
    auto wload = [&](int c, f32x4 (&dst)[8]) {
#pragma unroll
        for (int m = 0; m < 8; m++) dst[m] = wbase[(size_t)min(c * 8 + m, n_groups - 1) * 8 * 64];
    };
    f32x16 acc[4];
#pragma unroll
    for (int j = 0; j < 4; j++) acc[j] = f32x16{0};

    aload(0);
    wload(0, wr[0]);
    astore(0);
    if (total > 1) aload(1);
    __syncthreads();
    auto chunk = [&](int c, f32x4 (&wc)[8], f32x4 (&wn)[8]) __attribute__((always_inline)) {
        const int buf = c & 1;
        const int groups = min(8, n_groups - c * 8);
        if (c + 1 < total) wload(c + 1, wn);                 // a chunk (16 x 8 MFMAs per wave) ahead of its use
        f32x4 x[2][4];
        auto xread = [&](int m, f32x4 (&dst)[4]) {
#pragma unroll
            for (int j = 0; j < 4; j++) dst[j] = s_act[buf * HEAD_ABUF + act_slot(32 * j + r32, 2 * m + g)];
        };
        xread(0, x[0]);
#pragma unroll
        for (int m = 0; m < 8; m++) {
            if (m < 7) xread(m + 1, x[(m + 1) & 1]);
            if (m < groups) {
#pragma unroll
                for (int j = 0; j < 4; j++) acc[j] = mfma4(wc[m], x[m & 1][j], acc[j]);
            }
            if (m == 1 && c + 1 < total) astore(buf ^ 1);    // the next chunk, requested a chunk ago (its buffer was last read two barriers back)
        }
        if (c + 2 < total) aload(c + 2);
        __syncthreads();
    };
    for (int c = 0; c < total; c += 2) {
        chunk(c, wr[0], wr[1]);
        if (c + 1 < total) chunk(c + 1, wr[1], wr[0]);
    }
    // ---- behind the main loop.  The features of the workgroup's agents go to LDS (FK / 8 groups of the embedding's reduction)
    for (int k = tid; k < HEAD_M * A.FK; k += HEAD_THREADS) {
        const int row = k / A.FK, f = k - row * A.FK;
        ((float *)s_feat)[row * A.FK + f] = (f < A.F && a0 + row < A.n) ? A.feat[(size_t)(a0 + row) * A.F + f] : 0.0f;
    }
    // relu(acc + bias) -> one half of the hidden layer: lane (agent, g) of output tile w holds units 32 w + 8 q + 4 g + 0..3
    auto hidden_out = [&](const float *bias) {
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const f32x4 b = *(const f32x4 *)(bias + 32 * w + 8 * q + 4 * g);
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const f32x4 v = {fmaxf(acc[j][4 * q] + b[0], 0.0f), fmaxf(acc[j][4 * q + 1] + b[1], 0.0f), fmaxf(acc[j][4 * q + 2] + b[2], 0.0f),
                                 fmaxf(acc[j][4 * q + 3] + b[3], 0.0f)};
                s_hid[hid_slot(32 * j + r32, 8 * w + 2 * q + g)] = v;
            }
        }
    };
    // the dueling head: [32 outputs] x [128 agents], K = 512 hidden units in two halves; waves 0..3 take 32 agents each
    f32x16 h = {0};
    const int hagent = 32 * (w & 3) + r32;
    auto head_half = [&](int half) {
        if (w < 4) {
            f32x4 hw[2], hx[2];
            hw[0] = A.wh[(half * 32) * 64 + l]; hx[0] = s_hid[hid_slot(hagent, g)];
            for (int m = 0; m < 32; m++) {
                if (m < 31) { hw[(m + 1) & 1] = A.wh[(half * 32 + m + 1) * 64 + l]; hx[(m + 1) & 1] = s_hid[hid_slot(hagent, 2 * (m + 1) + g)]; }
                h = mfma4(hw[m & 1], hx[m & 1], h);
            }
        }
    };
    hidden_out(A.bv);          // (the main loop's last barrier is behind us: nobody reads the activation buffers any more)
    __syncthreads();
    head_half(0);
    // the feature embedding: K = FK, all eight waves (output tile w, four agent tiles)
#pragma unroll
    for (int j = 0; j < 4; j++) acc[j] = f32x16{0};
    for (int m = 0; m < A.FK / 8; m++) {
        const f32x4 we = A.we[((size_t)m * 8 + w) * 64 + l];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const f32x4 x = *(const f32x4 *)((const float *)s_feat + (32 * j + r32) * A.FK + 8 * m + 4 * g);
            acc[j] = mfma4(we, x, acc[j]);
        }
    }
    __syncthreads();           // the first half of the head has read relu(dense_view)
    hidden_out(A.be);
    __syncthreads();
    head_half(1);
    if (w < 4) {
        // lane (agent, g) holds outputs (r & 3) + 8 (r >> 2) + 4 g; its partner lane ^ 32 the other sixteen
        float best = -INFINITY, sum = 0.0f, value = 0.0f;
        int arg = 0x7FFFFFFF;
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int o = (r & 3) + 8 * (r >> 2) + 4 * g;
            if (o < A.n_action) { sum += h[r]; if (h[r] > best || (h[r] == best && o < arg)) { best = h[r]; arg = o; } }
            if (o == A.n_action) value = h[r];
        }
        const float obest = __shfl_xor(best, 32);
        const int oarg = __shfl_xor(arg, 32);
        sum += __shfl_xor(sum, 32);
        value += __shfl_xor(value, 32);
        if (obest > best || (obest == best && oarg < arg)) { best = obest; arg = oarg; }
        if (a0 + hagent < A.n) {
            if (g == 0) A.actions[a0 + hagent] = arg;      // argmax Q = argmax advantage: value and mean are per-agent constants
            if (A.q) {
                const float shift = value + A.value_bias - sum / (float)A.n_action;
#pragma unroll
                for (int r = 0; r < 16; r++) { const int o = (r & 3) + 8 * (r >> 2) + 4 * g; if (o < A.n_action) A.q[(size_t)(a0 + hagent) * A.n_action + o] = h[r] + shift; }
            }
        }
    }
}

static size_t act_bytes_f32(const PolicyDqnShape *s, int n) {      // whole groups of ACT_GROUP agents, whole K-chunks of 64
    const size_t n_pos = (size_t)(s->view_h - 4) * (s->view_w - 4);
    return (size_t)((n + ACT_GROUP - 1) / ACT_GROUP) * ((n_pos + 1) / 2) * ACT_GROUP * 64 * 4;
}
// agent pitch of an LDS image: == 16 / TA (mod 16) sixteen-byte units -- the 16 lanes of a ds_read_b128 service group (16 / TA consecutive
// positions x TA agents) then cover all 16 columns
static int pitch_for(int least, int ta) { const int want = 16 / ta; return least + ((want - least % 16) % 16 + 16) % 16; }
static size_t conv_lds(const PolicyDqnShape *s, int ta) {
    const int VP = pitch_for(s->view_h * s->view_w + 2, ta), AP = pitch_for((s->view_h - 2) * s->view_w, ta);
    return ((size_t)2 * ta * VP + (size_t)8 * ta * AP + 9 * 64 + 36 * 64) * 16 + 32 * 4;
}
static int conv_ta(const PolicyDqnShape *s) { return conv_lds(s, 4) <= 160 * 1024 ? 4 : 2; }

}  // namespace

extern "C" {

int policy_dqn_f32_supported(const PolicyDqnShape *s) {
    return s->view_c >= 1 && s->view_c <= 7 && s->view_h >= 5 && s->view_w >= 5 && s->view_h * s->view_w * conv_ta(s) <= CONV_CELLS * CONV_THREADS && s->feat >= 1 &&
           (s->feat + 7) / 8 * 8 <= HEAD_FMAX && s->n_action >= 1 && s->n_action <= 31 && conv_lds(s, conv_ta(s)) <= 160 * 1024;
}
int policy_dqn_f32_act_bytes(const PolicyDqnShape *s, int n, size_t *bytes) {
    *bytes = act_bytes_f32(s, n) + 4096;       // (+ the dump lines of k_dqn_conv_f32)
    return 0;
}
int policy_dqn_infer_f32(const PolicyDqnShape *s, const PolicyDqnWeightsF32 *w, const float *view, const float *feat, int n, void *act_workspace,
                         int *actions, float *q, void *stream) {
    if (!policy_dqn_f32_supported(s)) return 1;
    if (n <= 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    const int H = s->view_h, W = s->view_w, H2 = H - 4, W2 = W - 4;
    const int TA = conv_ta(s);
    const size_t lds = conv_lds(s, TA);
    int dev = 0, caller_dev = -1;
    if (hipGetDevice(&caller_dev) != hipSuccess) return 2;
    if (st) { if (hipStreamGetDevice(st, &dev) != hipSuccess || hipSetDevice(dev) != hipSuccess) return 2; }
    else dev = caller_dev;
    struct Restore { int d, cur; ~Restore() { if (d != cur) (void)hipSetDevice(d); } } restore{caller_dev, dev};
    constexpr int MAX_DEV = 64;
    if (dev < 0 || dev >= MAX_DEV) return 2;
    static bool lds_ok_dev[MAX_DEV] = {};
    if (!lds_ok_dev[dev]) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(k_dqn_conv_f32<4>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return 2;
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(k_dqn_conv_f32<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return 2;
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(k_dqn_head_f32), hipFuncAttributeMaxDynamicSharedMemorySize, (int)HEAD_LDS) != hipSuccess) return 2;
        lds_ok_dev[dev] = true;
    }
    ConvArgs C{};
    C.view = view; C.act = (float *)act_workspace; C.w1 = (const f32x4 *)w->conv1; C.w2 = (const f32x4 *)w->conv2; C.b2 = w->conv2_bias;
    C.n = n; C.H = H; C.W = W; C.C = s->view_c; C.VP = pitch_for(H * W + 2, TA); C.AP = pitch_for((H - 2) * W, TA); C.n_tiles = (n + TA - 1) / TA;
    C.dump = (f32x4 *)((char *)act_workspace + act_bytes_f32(s, n));
    static const int grid_cap = magent_amd::tune("policy_grid", 256);      // persistent: one workgroup of 8 waves per CU (tests: a few walk many tiles)
    const int grid = C.n_tiles < grid_cap ? C.n_tiles : grid_cap < 1 ? 1 : grid_cap;
    if (TA == 4) hipLaunchKernelGGL(k_dqn_conv_f32<4>, dim3(grid), dim3(CONV_THREADS), lds, st, C);
    else hipLaunchKernelGGL(k_dqn_conv_f32<2>, dim3(grid), dim3(CONV_THREADS), lds, st, C);
    HeadArgs Hd{};
    Hd.act = (const float *)act_workspace; Hd.feat = feat; Hd.wv = (const f32x4 *)w->dense_view; Hd.we = (const f32x4 *)w->dense_emb; Hd.wh = (const f32x4 *)w->head;
    Hd.bv = w->dense_view_bias; Hd.be = w->dense_emb_bias; Hd.value_bias = w->value_bias;
    Hd.n = n; Hd.K = H2 * W2 * 32; Hd.F = s->feat; Hd.FK = (s->feat + 7) / 8 * 8; Hd.n_action = s->n_action; Hd.actions = actions; Hd.q = q;
    hipLaunchKernelGGL(k_dqn_head_f32, dim3((n + HEAD_M - 1) / HEAD_M), dim3(HEAD_THREADS), HEAD_LDS, st, Hd);
    return hipGetLastError() == hipSuccess ? 0 : 3;
}

}  // extern "C"
