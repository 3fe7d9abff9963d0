// dqn_layout.h — packed parameter layout of the DQN variant's Q-network (reference
// UselessFiles/dqn.py:17-29: Linear(num_obs,256) LeakyReLU Linear(256,256) LeakyReLU Linear(256,18)),
// mirrored by fly_bproject_amd/dqn.py.  Same conventions as mlp_layout.h: one flat fp32 buffer with every
// layer as a row-major [N][K] matrix (K padded to a multiple of 8, N of the last layer padded to 32),
// plus two derived copies in MFMA fragment order.
//   L1  net.0  W1 [256][80]  (cols 73..79 = 0)   b1 [256]
//   L2  net.2  W2 [256][256]                     b2 [256]
//   L3  net.4  W3 [32][256]  (rows 18..31 = 0)   b3 [32]
// QF  forward operands: W1, W2 in the generic fragment order of mlp_layout.h
//       (((n/32)*(K/8) + kq)*64 + (h*32 + n%32))*4 + q, h = k/(K/2), kk = k%(K/2), kq = kk/4, q = kk%4;
//     W3 split-K over the four waves (64 k each):
//       ((w*8 + kq)*64 + (h*32 + n))*4 + q,   w = k/64, h = (k%64)/32, kq = (k%32)/4, q = k%4.
// QT  backward operands W3^T [256 outputs][32 reduced] and W2^T [256][256], generic fragment order.
#ifndef DQN_LAYOUT_H
#define DQN_LAYOUT_H

#define DQN_IN 73
#define DQN_IN_PAD 80
#define DQN_H 256
#define DQN_OUT 32
#define DQN_NACT 18

#define DQN_OFF_W1 0
#define DQN_OFF_B1 (DQN_OFF_W1 + DQN_H * DQN_IN_PAD)     /* 20480 */
#define DQN_OFF_W2 (DQN_OFF_B1 + DQN_H)                  /* 20736 */
#define DQN_OFF_B2 (DQN_OFF_W2 + DQN_H * DQN_H)          /* 86272 */
#define DQN_OFF_W3 (DQN_OFF_B2 + DQN_H)                  /* 86528 */
#define DQN_OFF_B3 (DQN_OFF_W3 + DQN_OUT * DQN_H)        /* 94720 */
#define DQN_PACKED_FLOATS (DQN_OFF_B3 + DQN_OUT)         /* 94752 */

#define DQN_OFF_F1 0
#define DQN_OFF_F2 (DQN_OFF_F1 + DQN_H * DQN_IN_PAD)     /* 20480 */
#define DQN_OFF_F3 (DQN_OFF_F2 + DQN_H * DQN_H)          /* 86016 */
#define DQN_FRAG_FLOATS (DQN_OFF_F3 + DQN_OUT * DQN_H)   /* 94208 */

#define DQN_OFF_T3 0                                     /* W3^T: 256 outputs, 32 reduced */
#define DQN_OFF_T2 (DQN_OFF_T3 + DQN_H * DQN_OUT)        /* 8192: W2^T 256 x 256 */
#define DQN_FRAG_T_FLOATS (DQN_OFF_T2 + DQN_H * DQN_H)   /* 73728 */


/* bf16x3 operand planes of the same weights (the layout of mlp_layout.h: element (n, k), term p of an operand with N outputs
 * and K reduced at (((n/32) * (K/16) + k/16) * 3 + p) * 512 + (((k%16)/8) * 32 + n%32) * 8 + k%8 16-bit words):
 *   QB  forward operands  W1 [256][80] | W2 [256][256] | W3 [32][256]   (the fused update's chain reads W3's k range of a wave
 *       as the sub-operand starting at k-step 4 w: split-K needs no layout of its own here)
 *   QTB backward operands W3^T [256 outputs][32 reduced] | W2^T [256][256]
 * dqn_adam_soft_update maintains QB / QTB of the online network and QB of the target network. */
#define DQN_OFF_QB1 0
#define DQN_OFF_QB2 (DQN_OFF_QB1 + 3 * DQN_H * DQN_IN_PAD)    /*  61440 */
#define DQN_OFF_QB3 (DQN_OFF_QB2 + 3 * DQN_H * DQN_H)         /* 258048 */
#define DQN_QB_HALVES (DQN_OFF_QB3 + 3 * DQN_OUT * DQN_H)     /* 282624 16-bit words */
#define DQN_OFF_QTB3 0
#define DQN_OFF_QTB2 (DQN_OFF_QTB3 + 3 * DQN_H * DQN_OUT)     /*  24576 */
#define DQN_QTB_HALVES (DQN_OFF_QTB2 + 3 * DQN_H * DQN_H)     /* 221184 16-bit words */

#endif
