/*
 * oracle/sepconv_ref.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Plain-C restatement of the reference's four sepconv CUDA kernels, one loop nest per kernel,
 * following the kernel text line by line (same index decomposition, same loop order, same
 * float accumulator, same multiplication order in * v * h):
 *
 *   sepconv_ref_forward          <- kernel_Sepconv_updateOutput          sepconv/sepconv_op/sepconv.py:12-29
 *   sepconv_ref_grad_vertical    <- kernel_Sepconv_updateGradVertical    sepconv/sepconv_op/sepconv.py:145-162
 *   sepconv_ref_grad_horizontal  <- kernel_Sepconv_updateGradHorizontal  sepconv/sepconv_op/sepconv.py:172-189
 *   sepconv_ref_grad_input       <- kernel_Sepconv_updateGradInput       sepconv/sepconv_op/sepconv.py:39-62
 *                                   with the bounds test corrected to `>= max` (the reference's
 *                                   `> max`, :51,54, reads one row/column past the end); the
 *                                   literal variant is kept as sepconv_ref_grad_input_literal for
 *                                   documentation and is only safe on padded buffers.
 *
 * The reference has no CPU implementation of this op (both branches raise NotImplementedError,
 * sepconv.py:293-294,373-374) and its kernels cannot be run here (CUDA + cupy), so this file is
 * pinned only by (a) being a transcription of the kernel text, (b) agreeing with an independent
 * PyTorch restatement checked by fp64 gradcheck (oracle/torch_ops.py, tests/test_oracle_golden.py).
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.
 *
 * Layout: contiguous fp32 NCHW.  `n` threads of the CUDA grid become the `idx` loop (OpenMP).
 */
#include <stddef.h>

#define IDX4(s1, s2, s3, a, b, c, d) ((((size_t)(a) * (s1) + (b)) * (s2) + (c)) * (s3) + (d))

void sepconv_ref_forward(const float* input, const float* vertical, const float* horizontal, float* output,
                         int B, int C, int Ho, int Wo, int K) {
  const int Hi = Ho + K - 1, Wi = Wo + K - 1;
  const long n = (long)B * C * Ho * Wo;
#pragma omp parallel for schedule(static)
  for (long idx = 0; idx < n; ++idx) {
    float acc = 0.0f;
    const int s = (int)((idx / Wo / Ho / C) % B);
    const int d = (int)((idx / Wo / Ho) % C);
    const int y = (int)((idx / Wo) % Ho);
    const int x = (int)(idx % Wo);
    for (int fy = 0; fy < K; ++fy) {
      for (int fx = 0; fx < K; ++fx) {
        acc += input[IDX4(C, Hi, Wi, s, d, y + fy, x + fx)] * vertical[IDX4(K, Ho, Wo, s, fy, y, x)] *
               horizontal[IDX4(K, Ho, Wo, s, fx, y, x)];
      }
    }
    output[idx] = acc;
  }
}

void sepconv_ref_grad_vertical(const float* input, const float* horizontal, const float* gradOutput,
                               float* gradVertical, int B, int C, int Ho, int Wo, int K) {
  const int Hi = Ho + K - 1, Wi = Wo + K - 1;
  const long n = (long)B * K * Ho * Wo;
#pragma omp parallel for schedule(static)
  for (long idx = 0; idx < n; ++idx) {
    float acc = 0.0f;
    const int s = (int)((idx / Wo / Ho / K) % B);
    const int fy = (int)((idx / Wo / Ho) % K);
    const int y = (int)((idx / Wo) % Ho);
    const int x = (int)(idx % Wo);
    for (int d = 0; d < C; ++d) {
      for (int fx = 0; fx < K; ++fx) {
        acc += gradOutput[IDX4(C, Ho, Wo, s, d, y, x)] * input[IDX4(C, Hi, Wi, s, d, y + fy, x + fx)] *
               horizontal[IDX4(K, Ho, Wo, s, fx, y, x)];
      }
    }
    gradVertical[idx] = acc;
  }
}

void sepconv_ref_grad_horizontal(const float* input, const float* vertical, const float* gradOutput,
                                 float* gradHorizontal, int B, int C, int Ho, int Wo, int K) {
  const int Hi = Ho + K - 1, Wi = Wo + K - 1;
  const long n = (long)B * K * Ho * Wo;
#pragma omp parallel for schedule(static)
  for (long idx = 0; idx < n; ++idx) {
    float acc = 0.0f;
    const int s = (int)((idx / Wo / Ho / K) % B);
    const int fx = (int)((idx / Wo / Ho) % K);
    const int y = (int)((idx / Wo) % Ho);
    const int x = (int)(idx % Wo);
    for (int d = 0; d < C; ++d) {
      for (int fy = 0; fy < K; ++fy) {
        acc += gradOutput[IDX4(C, Ho, Wo, s, d, y, x)] * input[IDX4(C, Hi, Wi, s, d, y + fy, x + fx)] *
               vertical[IDX4(K, Ho, Wo, s, fy, y, x)];
      }
    }
    gradHorizontal[idx] = acc;
  }
}

static void grad_input_impl(const float* vertical, const float* horizontal, const float* gradOutput,
                            float* gradInput, int B, int C, int Ho, int Wo, int K, int literal) {
  const int Hi = Ho + K - 1, Wi = Wo + K - 1;
  const long n = (long)B * C * Hi * Wi;
  const int maxOutY = Ho, maxOutX = Wo;
#pragma omp parallel for schedule(static)
  for (long idx = 0; idx < n; ++idx) {
    float acc = 0.0f;
    const int s = (int)((idx / Wi / Hi / C) % B);
    const int d = (int)((idx / Wi / Hi) % C);
    const int Y = (int)((idx / Wi) % Hi);
    const int X = (int)(idx % Wi);
    for (int fy = 0; fy < K; ++fy) {
      if (Y - fy < 0) break;
      if (literal ? (Y - fy > maxOutY) : (Y - fy >= maxOutY)) continue;
      for (int fx = 0; fx < K; ++fx) {
        if (X - fx < 0) break;
        if (literal ? (X - fx > maxOutX) : (X - fx >= maxOutX)) continue;
        acc += gradOutput[IDX4(C, Ho, Wo, s, d, Y - fy, X - fx)] *
               vertical[IDX4(K, Ho, Wo, s, fy, Y - fy, X - fx)] *
               horizontal[IDX4(K, Ho, Wo, s, fx, Y - fy, X - fx)];
      }
    }
    gradInput[idx] = acc;
  }
}

void sepconv_ref_grad_input(const float* vertical, const float* horizontal, const float* gradOutput,
                            float* gradInput, int B, int C, int Ho, int Wo, int K) {
  grad_input_impl(vertical, horizontal, gradOutput, gradInput, B, C, Ho, Wo, K, 0);
}

/* Literal bounds of the reference (out-of-bounds reads unless the buffers are padded by the caller). */
void sepconv_ref_grad_input_literal(const float* vertical, const float* horizontal, const float* gradOutput,
                                    float* gradInput, int B, int C, int Ho, int Wo, int K) {
  grad_input_impl(vertical, horizontal, gradOutput, gradInput, B, C, Ho, Wo, K, 1);
}
