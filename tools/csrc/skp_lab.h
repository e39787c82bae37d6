/* tools/csrc/skp_lab.h -- C ABI of libskp_lab.so: measurement aids and experiments kept OUT of the product library
 * (libskp_hip.so, include/skp.h).  Same conventions: extern "C", caller-owned device buffers, caller's stream, int return
 * (0, a positive hipError_t, or a negative SKP_E_* argument error). */
#ifndef SKP_LAB_H
#define SKP_LAB_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* EXPERIMENT, not the path of record (bench line dtype "f32-emulated (bf16x3)"): fp32 GEMM of the frozen nn.Linear layers on
 * the bf16 matrix cores, every fp32 operand split into three bf16 terms (h + m + l, round to nearest), six products accumulated
 * in fp32 (tools/csrc/skp_gemm_x3.hip).
 *   skp_gemm_x3_split_f32: planes [3][rows][cols] bf16 of w (transpose = 0: w is [rows, cols]; 1: w is [cols, rows]);
 *   skp_gemm_x3_nt_f32:    C[M,N] = A[M,K] . B[N,K]^T (+ bias[N], may be NULL), A fp32 row-major (lda), B = planes [3][N][K],
 *                          C fp32 (ldc).  K % 32 == 0, lda % 4 == 0, A and planes 16-byte aligned, else SKP_E_RANGE. */
int skp_gemm_x3_split_f32(const void* w, void* planes, int rows, int cols, int transpose, void* stream);
int skp_gemm_x3_nt_f32(const void* a, const void* b_planes, const void* bias, void* c, int M, int N, int K, int64_t lda,
                       int64_t ldc, void* stream);

/* Measurement aid (bench.py): sustained rate, in TFLOP/s, of back-to-back independent v_mfma_f32_16x16x4_f32 with
 * `waves_per_simd` (1..4) resident waves on every SIMD.  Synchronises `stream`.  scratch: >= 256*4*256 floats. */
int skp_probe_mfma_f32(int waves_per_simd, int iters, float* scratch, float* tflops, void* stream);

#ifdef __cplusplus
}
#endif
#endif
