// mlp_layout.h — packed parameter layout of the actor-critic network (reference ppo.py:10-102)
// shared by the MFMA kernels (mlp_mfma.hip) and mirrored by fly_bproject_amd/policy.py.
//
// One flat fp32 buffer `P` (MLP_PACKED_FLOATS) holds every layer as a row-major [N][K] matrix
// (torch's nn.Linear weight orientation), K padded to a multiple of 8 so that each lane's
// MFMA B-fragment is four consecutive floats (one 16-byte load):
//
//   L1  shared_net.0   W1 [256][80]   cols 0..72 = weight [256][73], cols 73..79 = 0      b1 [256]
//   L2  shared_net.2   W2 [128][256]                                                       b2 [128]
//   L3  to_mean.0 | to_value.0 stacked: W3 [128][128], rows 0..63 actor, 64..127 critic    b3 [128]
//   L4  to_mean.2 / to_value.2 as one [32][128] matrix over the concatenated (a1 | c1):
//         rows 0..17  = [ to_mean.2.weight [18][64] | 0 ]
//         row  18     = [ 0 | to_value.2.weight [1][64] ]
//         rows 19..31 = 0                                                                   b4 [32]
//   The torch-visible parameters are strided views into this buffer, so there is one copy of truth.
//   Structural zeros stay zero because their gradients are masked (policy.py).
//
// The kernels do not stream `P` itself: a wave's MFMA B-fragment is (32 output columns) x (4
// consecutive k) per lane-half, and fetching that from a row-major matrix touches 64 different
// cache lines per wave-instruction.  Two derived buffers hold the same weights in FRAGMENT ORDER,
// so that every wave-instruction reads one contiguous 1 KiB block:
//   PF  (MLP_FRAG_FLOATS)   forward operands  W_l   [N][K]
//   PTF (MLP_FRAG_T_FLOATS) backward operands W_l^T [K][N]  (layers 2..4; layer 1 needs no dX)
// Fragment order of a [N][K] operand (N outputs, K reduced): element (n, k) lives at
//     (((n/32) * (K/8) + kq) * 64 + (h*32 + n%32)) * 4 + q,   h = k / (K/2), kk = k % (K/2),
//     kq = kk / 4, q = kk % 4
// i.e. [column tile][k-quad][lane][4].  Layer 2 forward is stored as TWO such operands with K = 128
// (k in [0,128) at MLP_OFF_F2, k in [128,256) at MLP_OFF_F2 + 128*128): the forward kernel keeps only
// one 128-column half of H1 in LDS at a time and accumulates layer 2 over the two k ranges.
// Layer 4 forward (split-K over the four waves) uses
//     ((w*4 + kq) * 64 + (h*32 + n)) * 4 + q,   w = k/32, h = (k%32)/16, kq = (k%16)/4, q = k%4.
// fly_bproject_amd/policy.py builds the index maps; mlp_adam_step scatters every updated weight
// into PF/PTF, so the three buffers never diverge.
#ifndef MLP_LAYOUT_H
#define MLP_LAYOUT_H

#define MLP_IN 73
#define MLP_IN_PAD 80
#define MLP_H1 256
#define MLP_H2 128
#define MLP_H3 128      /* 64 actor + 64 critic */
#define MLP_OUT 32      /* 18 means + 1 value + 13 pad */
#define MLP_NACT 18

#define MLP_OFF_W1 0
#define MLP_OFF_B1 (MLP_OFF_W1 + MLP_H1 * MLP_IN_PAD)      /* 20480 */
#define MLP_OFF_W2 (MLP_OFF_B1 + MLP_H1)                   /* 20736 */
#define MLP_OFF_B2 (MLP_OFF_W2 + MLP_H2 * MLP_H1)          /* 53504 */
#define MLP_OFF_W3 (MLP_OFF_B2 + MLP_H2)                   /* 53632 */
#define MLP_OFF_B3 (MLP_OFF_W3 + MLP_H3 * MLP_H2)          /* 70016 */
#define MLP_OFF_W4 (MLP_OFF_B3 + MLP_H3)                   /* 70144 */
#define MLP_OFF_B4 (MLP_OFF_W4 + MLP_OUT * MLP_H3)         /* 74240 */
#define MLP_PACKED_FLOATS (MLP_OFF_B4 + MLP_OUT)           /* 74272 */

#define MLP_OFF_F1 0
#define MLP_OFF_F2 (MLP_OFF_F1 + MLP_H1 * MLP_IN_PAD)      /* 20480 */
#define MLP_OFF_F3 (MLP_OFF_F2 + MLP_H2 * MLP_H1)          /* 53248 */
#define MLP_OFF_F4 (MLP_OFF_F3 + MLP_H3 * MLP_H2)          /* 69632 */
#define MLP_FRAG_FLOATS (MLP_OFF_F4 + MLP_OUT * MLP_H3)    /* 73728 */

#define MLP_OFF_TF2 0                                      /* W2^T: 256 outputs, 128 reduced */
#define MLP_OFF_TF3 (MLP_OFF_TF2 + MLP_H1 * MLP_H2)        /* 32768: W3^T 128 x 128 */
#define MLP_OFF_TF4 (MLP_OFF_TF3 + MLP_H2 * MLP_H3)        /* 49152: W4^T 128 outputs, 32 reduced */
#define MLP_FRAG_T_FLOATS (MLP_OFF_TF4 + MLP_H3 * MLP_OUT) /* 53248 */


/* bf16x3 operand planes (v_mfma_f32_32x32x16_bf16): every fp32 weight w is also kept as three bf16
 * terms w = w0 + w1 + w2 (w0 = bf16(w), w1 = bf16(w - w0), w2 = bf16(w - w0 - w1); the sum is exact)
 * in the order the 32x32x16 MFMA consumes them: element (n, k) of an operand with N outputs and K
 * reduced, term p, lives at (16-bit units)
 *     (((n/32) * (K/16) + k/16) * 3 + p) * 512 + (((k%16)/8) * 32 + n%32) * 8 + k%8
 * i.e. [column tile][16-wide k block][term][lane][8]: one wave-load = one contiguous KiB.
 * PB  forward operands  W1 [256][80] | W2 as two K = 128 operands (k < 128, k >= 128) | W3 | W4
 * PTB backward operands W2^T [256][128] | W3^T [128][128] | W4^T [128][32]
 * tools/bf16x3_gemm.hip: six product terms (w0x0 in one accumulator, w0x1 w1x0 w1x1 w0x2 w2x0 in a
 * second, summed at the end) are MORE accurate than v_mfma_f32_32x32x2_f32 on the same data. */
#define MLP_OFF_PB1 0
#define MLP_OFF_PB2 (MLP_OFF_PB1 + 3 * MLP_H1 * MLP_IN_PAD)   /*  61440 */
#define MLP_OFF_PB3 (MLP_OFF_PB2 + 3 * MLP_H2 * MLP_H1)       /* 159744 */
#define MLP_OFF_PB4 (MLP_OFF_PB3 + 3 * MLP_H3 * MLP_H2)       /* 208896 */
#define MLP_PB_HALVES (MLP_OFF_PB4 + 3 * MLP_OUT * MLP_H3)    /* 221184 16-bit words */
#define MLP_OFF_PTB2 0
#define MLP_OFF_PTB3 (MLP_OFF_PTB2 + 3 * MLP_H1 * MLP_H2)     /*  98304 */
#define MLP_OFF_PTB4 (MLP_OFF_PTB3 + 3 * MLP_H2 * MLP_H3)     /* 147456 */
#define MLP_PTB_HALVES (MLP_OFF_PTB4 + 3 * MLP_H3 * MLP_OUT)  /* 159744 16-bit words */

#endif
