/*
 * magent_policy.h -- C-ABI of the inference path of the reference's deep Q network on MI355X (libmagent.so).
 *
 * What it replaces: the forward pass of python/magent/builtin/tf_model/dqn.py:151-189 (2 x conv3x3(32, valid, relu) ->
 * dense 256 || dense 256 -> dueling head) as DeepQNetwork.infer_action calls it (dqn.py:191-228) between
 * GridWorld.get_observation and GridWorld.set_action -- BASELINE config 5's policy step.  Additive: the reference has no
 * C entry point here (its model is a TensorFlow graph); the Python binding is magent_amd/builtin/torch_model/hip_policy.py.
 *
 * All pointers are DEVICE pointers; the call enqueues two kernels on `stream` and returns.  Inputs are the engine's own
 * observation tensors (env_get_observation_device): view float[n][view_h][view_w][view_c], feature float[n][feat].
 * Numerics: inputs, weights and inter-layer activations are rounded to bf16, products accumulate in f32 (MFMA).
 */
#ifndef MAGENT_AMD_POLICY_H
#define MAGENT_AMD_POLICY_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct {
    int view_h, view_w, view_c;   /* view_c <= 7 (the eighth channel of a window cell carries conv1's bias); view_h * view_w <= 256 */
    int feat;                     /* <= 64 */
    int n_action;                 /* <= 31 */
} PolicyDqnShape;

/* Weights in "fragment order" (bf16, 16-byte units of 8 values): for k-step s (16 values of the reduction dimension) and
 * 32-wide output tile T, lane l (0..63) holds W[out = 32 T + (l & 31)][k = 16 s + 8 (l >> 5) + 0..7].
 * The reduction index k of each layer:
 *   conv1      : j * 8 + channel, j = 0..9 standing for tap (ky * 3 + kx) 0, 3, 1, 4, 2, 5, 6, 7, 8, padding -- the two taps of
 *                a k-step lie a constant number of window cells apart (channels padded to 8: 5 k-steps)       [5][64][8]
 *                its bias is the weight of (tap 0, channel 7): the kernel feeds a constant 1.0 there
 *   conv2      : tap * 32 + slot                          (18 k-steps)                                         [18][64][8]
 *   dense_view : position (y * (view_w - 4) + x) * 32 + slot                                                    [K/16][8][64][8]
 *   dense_emb  : feature index (padded to a multiple of 16)                                                     [FK/16][8][64][8]
 *   head       : hidden slot (tile T' of dense_view / 256 + tile T' of dense_emb: 32 T' + slot); outputs 0..n_action-1 =
 *                advantage, output n_action = value, the rest zero                                              [32][64][8]
 * A "slot" s of a 32-wide tile stands for its channel / output (s & 3) + 8 ((s & 15) >> 2) + 4 (s >> 4): the order in which
 * a lane of the MFMA result holds them (magent_amd/csrc/policy.hip: ch_of).  Biases are float[tiles][32] in slot order. */
typedef struct {
    const void *conv1, *conv2, *dense_view, *dense_emb, *head;
    const float *conv2_bias, *dense_view_bias, *dense_emb_bias;
    float value_bias;
} PolicyDqnWeights;

/* 1 if the kernels take this shape */
int policy_dqn_supported(const PolicyDqnShape *shape);
/* size of the activation workspace (conv2's output, bf16) for n agents */
int policy_dqn_act_bytes(const PolicyDqnShape *shape, int n, size_t *bytes);
/* actions[i] = argmax_a Q(view[i], feature[i]); q (optional, may be NULL) = float[n][n_action].  Returns 0, or non-zero if the
 * shape is not supported / a launch failed (nothing is written then). */
int policy_dqn_infer(const PolicyDqnShape *shape, const PolicyDqnWeights *weights, const float *view, const float *feature, int n,
                     void *act_workspace, int *actions, float *q, void *stream);

/* the same with the views as the engine's bf16 cells (env_get_observation_device_bf16: [n][view_h][view_w][8], channel 7 = 1.0):
 * they ARE conv1's operands -- nothing is converted, and the kernel reads 16 bytes per window cell instead of 4 * view_c */
int policy_dqn_infer_bf16(const PolicyDqnShape *shape, const PolicyDqnWeights *weights, const void *view_cells, const float *feature, int n,
                          void *act_workspace, int *actions, float *q, void *stream);

/* ---- the same network in the reference's own arithmetic: float32 inputs, weights, activations and accumulation, on the f32 matrix
 * instruction (v_mfma_f32_32x32x2_f32: bit for bit a k-ordered chain of fmaf) -- magent_amd/csrc/policy_f32.hip.
 * Weights in "f32 fragment order" (float, 16-byte units of 4 values): the reduction index is cut into groups of 8; for group m and 32-wide
 * output tile T, lane l (0..63) holds W[out = 32 T + (l & 31)][k = 8 m + 4 (l >> 5) + 0..3].  The reduction index k of each layer:
 *   conv1      : tap (ky * 3 + kx) * 8 + channel (channels padded to 8); the bias is the weight of (tap 0, channel 7): the kernel
 *                feeds a constant 1.0 there                                                                     [9][64][4]
 *   conv2      : tap * 32 + channel                                                                              [36][64][4]
 *   dense_view : position (y * (view_w - 4) + x) * 32 + channel                                                  [K/8][8][64][4]
 *   dense_emb  : feature index (padded to a multiple of 8)                                                       [FK/8][8][64][4]
 *   head       : hidden unit (dense_view's 256, then dense_emb's 256); outputs 0..n_action-1 = advantage, output n_action = value,
 *                the rest zero                                                                                   [64][64][4]
 * Channels, hidden units and biases are in their natural order (float[32], float[256], float[256]). */
typedef struct {
    const void *conv1, *conv2, *dense_view, *dense_emb, *head;
    const float *conv2_bias, *dense_view_bias, *dense_emb_bias;
    float value_bias;
} PolicyDqnWeightsF32;

/* 1 if the f32 kernels take this shape (view_c <= 7, feat <= 56, n_action <= 31, views up to ~19 x 19: the conv kernel's LDS images of
 * two agents must fit 160 KB) */
int policy_dqn_f32_supported(const PolicyDqnShape *shape);
/* size of the activation workspace (conv2's output, float32) for n agents */
int policy_dqn_f32_act_bytes(const PolicyDqnShape *shape, int n, size_t *bytes);
/* as policy_dqn_infer: view float[n][view_h][view_w][view_c], feature float[n][feat] as env_get_observation_device writes them */
int policy_dqn_infer_f32(const PolicyDqnShape *shape, const PolicyDqnWeightsF32 *weights, const float *view, const float *feature, int n,
                         void *act_workspace, int *actions, float *q, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* MAGENT_AMD_POLICY_H */
